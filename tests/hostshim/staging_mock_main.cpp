// TEST INFRASTRUCTURE: the HOST side of libbee2hip.so -- bee2_amd/csrc/staging.hpp (error record, scratch pool, per-thread
// staging, host fallback, the duplex pipeline of large in-place batches) and multi.hip (the persistent worker pool behind the
// *_multi entries) -- compiled with g++ against tests/hostshim/mockhip (streams = threads, device memory = heap) so that it
// runs under -fsanitize=thread and -fsanitize=address,undefined on a box without a GPU.  The kernels are stand-ins (a byte
// transform queued on the stream); what is under test is the library's own synchronisation: every result is compared with
// the same transform applied directly, every failure path must leave the contract of *done_units intact, and the sanitizer
// must stay silent.  Built and run by tests/test_host_sanitizers.py.  Nothing here ships.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <new>
#include "../../bee2_amd/csrc/staging.hpp"

namespace bee2hip {
err_t upload_beltH(const uint8_t *) { return ERR_OK; }
err_t upload_beltH_bign(const uint8_t *) { return ERR_OK; }
}  // namespace bee2hip

// ---- the single-device entries multi.hip dispatches to: CPU stand-ins with the real contracts -------------------------
static inline octet tf(octet x, size_t unit) { return (octet)(x * 5 + 1 + (octet)(unit * 7)); }      // "kernel": depends on the unit's index
static std::atomic<int> g_calls{0};
extern "C" {
err_t bee2hip_set_device(int d) { return hipSetDevice(d) == hipSuccess ? ERR_OK : ERR_BEE2HIP_DEVICE; }
err_t bee2hip_bashF_batch(octet *states, size_t n)
{
    g_calls.fetch_add(1);
    for (size_t i = 0; i < n; ++i) hostp::bashF(states + 192 * i);
    return ERR_OK;
}
struct ctr_st { u32 key[8]; u32 ctr[4]; octet block[16]; size_t reserved; };
err_t bee2hip_beltCTR_bulk(void *buf, size_t count, void *state)
{
    g_calls.fetch_add(1);
    ctr_st *st = (ctr_st *)state;
    hostp::ctr_blocks(hostT(), (octet *)buf, count, st->key, st->ctr, st->block, &st->reserved);
    return ERR_OK;
}
err_t bee2hip_bashF_batch_dev(void *d_states, size_t n, void *stream)
{
    octet *p = (octet *)d_states;
    mockhipLaunch(as_stream(stream), [p, n] { for (size_t i = 0; i < n; ++i) hostp::bashF(p + 192 * i); });
    return n == 12345 ? ERR_BAD_INPUT : ERR_OK;          // (a launcher that queues work and THEN reports an error: drain() must still wait)
}
#define STUB(name, ...) err_t name(__VA_ARGS__) { return ERR_OK; }
STUB(bee2hip_beltCTR_blocks_dev, void *, size_t, const u32 *, const u32 *, uint64_t, void *)
STUB(bee2hip_bignVerify_batch, const bign_params *, const octet *, size_t, const octet *, const octet *, const octet *, size_t, err_t *)
STUB(bee2hip_bignVerify_keyed_batch, const bign_params *, const octet *, size_t, const octet *, const octet *, const octet *, size_t, const u32 *, size_t, err_t *)
STUB(bee2hip_bignVerify_onekey_batch, const bign_params *, const octet *, size_t, const octet *, const octet *, const octet *, size_t, err_t *)
STUB(bee2hip_bignSign2_batch, const bign_params *, const octet *, size_t, const octet *, const octet *, const void *, size_t, size_t, octet *, err_t *)
STUB(bee2hip_bashHash_beltMAC_batch, const octet *, size_t, size_t, size_t, const octet *, size_t, octet *, octet *)
STUB(bee2hip_hash_ragged, size_t, const octet *, const uint64_t *, size_t, octet *)
STUB(bee2hip_bignVerifyL_batch_dev, size_t, const octet *, size_t, const void *, const void *, const void *, size_t, void *, void *)
STUB(bee2hip_bignVerifyL_onekey_batch_dev, size_t, const octet *, size_t, const void *, const void *, const octet *, size_t, void *, void *)
STUB(bee2hip_bignVerifyL_keyed_batch_dev, size_t, const octet *, size_t, const void *, const void *, const octet *, size_t, const void *, size_t, void *, void *)
STUB(bee2hip_bashHash_beltMAC_batch_dev, const void *, size_t, size_t, size_t, const octet *, size_t, void *, void *, void *)
}
#include "../../bee2_amd/csrc/multi.hip"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

// the "kernel" of the duplex tests: transforms units [first, first + cnt) of the device chunk on the stream
static err_t launch_tf(octet *d, size_t unit_bytes, size_t first, size_t cnt, hipStream_t st)
{
    mockhipLaunch(st, [=] { for (size_t u = 0; u < cnt; ++u) for (size_t b = 0; b < unit_bytes; ++b) d[u * unit_bytes + b] = tf(d[u * unit_bytes + b], first + u); });
    return ERR_OK;
}
static void fill(std::vector<octet> &v, unsigned seed) { for (size_t i = 0; i < v.size(); ++i) v[i] = (octet)((i * 2654435761u + seed) >> 13); }

static void duplex_ok(size_t unit_bytes, size_t units, size_t chunk, int ramp)
{
    if (ramp >= 0) g_duplex_ramp = ramp;                  // (-1: leave the knob alone -- the concurrent callers below must not write it)
    std::vector<octet> host(unit_bytes * units), want;
    fill(host, (unsigned)(units + chunk));
    want = host;
    for (size_t u = 0; u < units; ++u) for (size_t b = 0; b < unit_bytes; ++b) want[u * unit_bytes + b] = tf(want[u * unit_bytes + b], u);
    void *dev = nullptr;
    CHECK(hipMalloc(&dev, host.size()) == hipSuccess);
    size_t done = 777;
    const err_t code = duplex_inplace(host.data(), (octet *)dev, unit_bytes, units, chunk,
                                      [=](octet *d, size_t first, size_t cnt, hipStream_t st) { return launch_tf(d, unit_bytes, first, cnt, st); }, &done);
    CHECK(code == ERR_OK && done == units && host == want);
    hipFree(dev);
}

// mode 1: the launcher reports an error at chunk `bad`; 2: it throws std::bad_alloc there; 3: it throws something else
static void duplex_fails(size_t units, size_t chunk, size_t bad, int mode)
{
    g_duplex_ramp = 0;
    const size_t ub = 16;
    std::vector<octet> host(ub * units), orig;
    fill(host, 99);
    orig = host;
    void *dev = nullptr;
    CHECK(hipMalloc(&dev, host.size()) == hipSuccess);
    size_t done = 777, seen = 0;
    const err_t code = duplex_inplace(host.data(), (octet *)dev, ub, units, chunk, [&](octet *d, size_t first, size_t cnt, hipStream_t st) -> err_t {
        if (seen++ == bad) {
            if (mode == 2) throw std::bad_alloc();
            if (mode == 3) throw 42;
            return ERR_BEE2HIP_DEVICE;
        }
        return launch_tf(d, ub, first, cnt, st);
    }, &done);
    CHECK(code == (mode == 2 ? ERR_OUTOFMEMORY : ERR_BEE2HIP_DEVICE));       // a code, never an exception
    CHECK(done <= bad * chunk && done % chunk == 0);
    // the contract: the leading `done` units are transformed in the caller's buffer, everything behind them is untouched
    for (size_t u = 0; u < units; ++u)
        for (size_t b = 0; b < ub; ++b) CHECK(host[u * ub + b] == (u < done ? tf(orig[u * ub + b], u) : orig[u * ub + b]));
    hipFree(dev);
}

static void pool_and_fallback()
{
    // scratch_for_stream from many threads on the NULL stream (keyed by thread) and on private streams; blocks grow; threads exit
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int t = 0; t < 8; ++t)
        th.emplace_back([t, &bad] {
            hipStream_t st = nullptr;
            if (t & 1) { if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { bad++; return; } }
            for (size_t sz = 64; sz <= 65536; sz *= 4) {
                void *p = nullptr;
                if (scratch_for_stream(st, t & 3, sz, &p) != ERR_OK || !p) { bad++; continue; }
                octet *q = (octet *)p;
                mockhipLaunch(st, [q, sz, t] { memset(q, t, sz); });
                if (hipStreamSynchronize(st) != hipSuccess || q[sz - 1] != (octet)t) bad++;
            }
            if (st) { scratch_release_stream(st); hipStreamDestroy(st); }
        });
    for (auto &t : th) t.join();
    CHECK(bad.load() == 0);
    // with_host: a GPU path that fails twice is finished on the host and counted; ERR_BAD_INPUT is reported, not retried
    g_force.store(FORCE_AUTO);
    int host_runs = 0, gpu_runs = 0;
    const unsigned long long fb0 = g_n_fallback.load();
    CHECK(with_host(K_PARALLEL, 1 << 20, "test", [&]() -> err_t { ++gpu_runs; return ERR_BEE2HIP_DEVICE; }, [&] { ++host_runs; }) == ERR_OK);
    CHECK(gpu_runs == 2 && host_runs == 1 && g_n_fallback.load() == fb0 + 1);
    CHECK(with_host(K_PARALLEL, 1 << 20, "test", [&]() -> err_t { throw std::bad_alloc(); }, [&] { ++host_runs; }) == ERR_OK && host_runs == 2);
    CHECK(with_host(K_PARALLEL, 1 << 20, "test", [&]() -> err_t { return ERR_BAD_INPUT; }, [&] { ++host_runs; }) == ERR_BAD_INPUT && host_runs == 2);
    g_force.store(FORCE_GPU);
    CHECK(with_host(K_PRIM, 16, "test", [&]() -> err_t { return ERR_BEE2HIP_DEVICE; }, [&] { ++host_runs; }) == ERR_BEE2HIP_DEVICE && host_runs == 2);
    g_force.store(FORCE_AUTO);
    // a refused device allocation is ERR_OUTOFMEMORY, and the pool entry stays usable
    void *p = nullptr;
    mockhip::g().malloc_fail_in.store(1);
    CHECK(scratch_for_stream(nullptr, 3, 1 << 22, &p) == ERR_OUTOFMEMORY);
    CHECK(scratch_for_stream(nullptr, 3, 1 << 22, &p) == ERR_OK && p);
}

static void worker_pool()
{
    // the *_multi entries from several host threads at once: one batch over all (logical) devices at a time, results exact
    const size_t n = 301;
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int t = 0; t < 4; ++t)
        th.emplace_back([t, n, &bad] {
            std::vector<octet> a(192 * n), want;
            fill(a, 1000 + t);
            want = a;
            for (size_t i = 0; i < n; ++i) hostp::bashF(want.data() + 192 * i);
            if (bee2hip_bashF_batch_multi(a.data(), n, 1 + t) != ERR_OK || a != want) bad++;
            // resident shards: worker i launches on its own stream and drains it -- the data must be final when the call returns
            const int nd = 2;
            std::vector<std::vector<octet>> sh(nd, std::vector<octet>(192 * 40)), w2;
            for (int i = 0; i < nd; ++i) fill(sh[i], 7 * t + i);
            w2 = sh;
            for (int i = 0; i < nd; ++i) for (size_t k = 0; k < 40; ++k) hostp::bashF(w2[i].data() + 192 * k);
            void *ptrs[nd] = {sh[0].data(), sh[1].data()};
            size_t cnt[nd] = {40, 40};
            if (bee2hip_bashF_batch_multi_dev(ptrs, cnt, nd) != ERR_OK || sh != w2) bad++;
        });
    for (auto &t : th) t.join();
    CHECK(bad.load() == 0);
    // a launcher that queued work and then reported an error: the call returns the error AND the queued work is finished
    std::vector<octet> s(192 * 12345), w;
    fill(s, 5);
    w = s;
    for (size_t k = 0; k < 12345; ++k) hostp::bashF(w.data() + 192 * k);
    void *ptrs[1] = {s.data()};
    size_t cnt[1] = {12345};
    CHECK(bee2hip_bashF_batch_multi_dev(ptrs, cnt, 1) == ERR_BAD_INPUT);
    CHECK(s == w);                                      // nothing was still in flight when the call came back
    // CTR over several logical devices leaves the state exactly as ONE serial call does
    ctr_st st0;
    for (int i = 0; i < 8; ++i) st0.key[i] = 0x01020304u * (i + 1);
    for (int i = 0; i < 4; ++i) st0.ctr[i] = 0xFFFFFFF0u + i;
    memset(st0.block, 0, 16); st0.reserved = 0;
    std::vector<octet> m(16 * 1000 + 5), m1;
    fill(m, 3);
    m1 = m;
    ctr_st a = st0, b = st0;
    CHECK(bee2hip_beltCTR_bulk(m1.data(), m1.size(), &a) == ERR_OK);
    CHECK(bee2hip_beltCTR_bulk_multi(m.data(), m.size(), &b, 3) == ERR_OK);
    CHECK(m == m1 && !memcmp(&a, &b, sizeof a));
}

int main()
{
    CHECK(ensure_device() == ERR_OK);
    for (int ramp = 0; ramp < 2; ++ramp) {
        duplex_ok(16, 1, 4, ramp);
        duplex_ok(16, 1000, 64, ramp);
        duplex_ok(192, 777, 100, ramp);
        duplex_ok(16, 64 * 40, 64, ramp);
    }
    for (int mode = 1; mode <= 3; ++mode) {
        duplex_fails(64 * 10, 64, 0, mode);
        duplex_fails(64 * 10, 64, 4, mode);
        duplex_fails(64 * 10 + 3, 64, 9, mode);
    }
    {   // several host threads, each with its own pipeline (streams are per thread)
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t) th.emplace_back([t] { duplex_ok(16, 3000 + 100 * t, 128, -1); });
        for (auto &t : th) t.join();
    }
    pool_and_fallback();
    worker_pool();
    CHECK(hipDeviceSynchronize() == hipSuccess);
    puts("staging mock ok");
    return 0;
}
