// multi.hip -- one host batch over every visible GPU from ONE process (VERDICT r01 "next" 3b).
//
// bench.py and the tests shard with one process per GPU over torch.distributed; a plain C caller of the
// library (bee2cmd-style programs, examples/bsum_hip.c) has no such launcher.  The *_multi entry points give
// him the same partition: the batch is cut into contiguous index ranges, one worker thread per device does
// hipSetDevice + the single-device host entry on its range, nothing crosses devices (the 48 bytes of key +
// counter are arguments of the call).  CTR hands every range its first_block so lanes compute
// ctr0 + first + i + 1 directly (belt_ctr.c:27-35 -- the serial counter walk is not replayed).
//
// BEE2HIP_FAKE_DEVICES=k makes the library pretend there are k devices, mapped round-robin onto the real ones:
// the sharding, the threads and the state hand-over can then be exercised on a one-GPU box
// (tests/test_gpu_multi.py); bee2hip_multi_plan is the pure index arithmetic, callable without a GPU
// (tests/test_multi_plan.py runs it against the oracle).
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "common.hpp"

using namespace bee2hip;

// range i of `parts` over n items: [n i / parts, n (i + 1) / parts) -- sizes differ by at most one; the same cut as
// bee2_amd/shard.py shard_range, which bench.py and the torch.distributed tests use
extern "C" err_t bee2hip_multi_plan(size_t n, int parts, int i, size_t *first, size_t *count)
try {
    if (parts <= 0 || i < 0 || i >= parts || !first || !count) return ERR_BAD_INPUT;
    const unsigned __int128 N = n;
    const size_t lo = (size_t)(N * (unsigned)i / (unsigned)parts), hi = (size_t)(N * ((unsigned)i + 1u) / (unsigned)parts);
    *first = lo;
    *count = hi - lo;
    return ERR_OK;
} B2H_CATCH

static int real_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
extern "C" int bee2hip_device_count(void)
{
    const char *fake = getenv("BEE2HIP_FAKE_DEVICES");
    if (fake && atoi(fake) > 0 && real_device_count() > 0) return atoi(fake);
    return real_device_count();
}

namespace {
static thread_local hipStream_t t_wstream = nullptr;      // a pool worker's own non-blocking stream on its device
static thread_local hipEvent_t t_wevent = nullptr;
// Worker pool: ONE persistent thread per logical device, created on first use and kept for the life of the process
// (fresh std::threads per call threw their per-thread staging -- pinned blocks, device scratch, the NULL-stream
// pool -- away at every return and paid device-synchronising frees for it).  A worker binds its device once; its
// thread-local staging therefore survives between calls.  Calls from several host threads take turns (one batch over
// all devices at a time).  The pool is never destroyed: the workers sleep on a condition variable until process exit.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<err_t()> job;          // set by the dispatcher, cleared by the worker
    bool has_job = false, done = false;
    err_t code = ERR_OK;
    int device = -1;                     // real device this worker is bound to
};
struct Pool {
    std::mutex call_mu;                  // one multi-device batch at a time
    std::vector<Worker *> w;
    Worker *get(size_t i, int real_dev)
    {
        while (w.size() <= i) w.push_back(nullptr);
        if (w[i] && w[i]->device == real_dev) return w[i];
        // (a worker bound to another device -- the visible device set changed: BEE2HIP_FAKE_DEVICES in tests -- is left idle)
        Worker *k = new Worker;
        k->device = real_dev;
        k->th = std::thread([k, real_dev] {
            err_t bind = bee2hip_set_device(real_dev);
            // the worker's own queue (round 5): the _multi_dev jobs launch here, so logical devices that share a card
            // (BEE2HIP_FAKE_DEVICES, or a caller who cuts one card's data into several shards) run side by side instead of
            // taking turns on the card's NULL stream
            if (bind == ERR_OK && (hipStreamCreateWithFlags(&t_wstream, hipStreamNonBlocking) != hipSuccess ||
                                   hipEventCreateWithFlags(&t_wevent, hipEventDisableTiming) != hipSuccess)) {
                bind = hip_fail(hipGetLastError(), "worker stream");
                if (t_wstream) { (void)hipStreamDestroy(t_wstream); t_wstream = nullptr; }     // (the stream went when the event could not be had)
            }
            for (;;) {
                std::unique_lock<std::mutex> lk(k->mu);
                k->cv.wait(lk, [k] { return k->has_job; });
                std::function<err_t()> f = std::move(k->job);
                lk.unlock();
                err_t c = bind;
                if (c == ERR_OK) {
                    // nothing may leave a worker as an exception (std::terminate would take the whole process down) nor
                    // cross the C ABI: an allocation that fails inside a job is the job's ERR_OUTOFMEMORY
                    try { c = f(); }
                    catch (const std::bad_alloc &) { c = ERR_OUTOFMEMORY; }
                    catch (...) { c = hip_fail(hipErrorUnknown, "exception in a multi-device job"); }
                }
                lk.lock();
                k->code = c; k->has_job = false; k->done = true;
                k->cv.notify_all();
            }
        });
        k->th.detach();
        w[i] = k;
        return k;
    }
};
Pool &pool()
{
    static Pool *p = new Pool;           // leaked on purpose: no join at exit
    return *p;
}

// run job(i, parts) on logical device i = 0 .. parts-1, each on its pool worker bound to real device i % real
template <class F>
err_t run_on_devices_unguarded(int ndev, F job);
// the dispatcher allocates too (worker objects, std::function copies): same rule
template <class F>
err_t run_on_devices(int ndev, F job)
{
    try { return run_on_devices_unguarded(ndev, job); }
    catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }
    catch (...) { return hip_fail(hipErrorUnknown, "exception in the multi-device dispatcher"); }
}
template <class F>
err_t run_on_devices_unguarded(int ndev, F job)
{
    const int real = real_device_count();
    if (real <= 0) return hip_fail(hipErrorNoDevice, "hipGetDeviceCount");
    int parts = ndev > 0 ? ndev : bee2hip_device_count();
    if (parts <= 0) return ERR_BAD_INPUT;
    Pool &P = pool();
    std::lock_guard<std::mutex> call(P.call_mu);
    std::vector<Worker *> ws((size_t)parts);
    for (int i = 0; i < parts; ++i) {
        Worker *k = ws[(size_t)i] = P.get((size_t)i, i % real);
        std::lock_guard<std::mutex> lk(k->mu);
        k->job = [job, i, parts]() -> err_t { return job(i, parts); };
        k->done = false;
        k->has_job = true;
        k->cv.notify_all();
    }
    err_t first_bad = ERR_OK;
    for (Worker *k : ws) {
        std::unique_lock<std::mutex> lk(k->mu);
        k->cv.wait(lk, [k] { return k->done; });
        if (k->code != ERR_OK && first_bad == ERR_OK) first_bad = k->code;
    }
    return first_bad;
}
}  // namespace

extern "C" err_t bee2hip_bashF_batch_multi(octet *states, size_t n, int ndev)
try {
    if (n == 0) return ERR_OK;
    if (!states) return ERR_BAD_INPUT;
    return run_on_devices(ndev, [=](int i, int parts) {
        size_t lo, cnt;
        bee2hip_multi_plan(n, parts, i, &lo, &cnt);
        return bee2hip_bashF_batch(states + 192 * lo, cnt);
    });
} B2H_CATCH

// belt_ctr_st as in capi.hip / belt_lcl.h:135-141
struct multi_ctr_st { u32 key[8]; u32 ctr[4]; octet block[16]; size_t reserved; };

extern "C" err_t bee2hip_beltCTR_bulk_multi(void *buf_, size_t count, void *ctr_state, int ndev)
try {
    multi_ctr_st *st = (multi_ctr_st *)ctr_state;
    octet *buf = (octet *)buf_;
    if (!st || (count && !buf)) return ERR_BAD_INPUT;
    // left-over gamma of the previous call and everything shorter than a few blocks: the single-device entry
    if (st->reserved || count < 4096) return bee2hip_beltCTR_bulk(buf, count, ctr_state);
    const size_t full = count / 16, tail = count % 16;
    // whole blocks are cut into ranges; the LAST range also takes the partial tail and leaves the state behind
    const multi_ctr_st start = *st;
    multi_ctr_st last = *st;
    const err_t code = run_on_devices(ndev, [&](int i, int parts) {
        size_t lo, cnt;
        bee2hip_multi_plan(full, parts, i, &lo, &cnt);
        multi_ctr_st mine = start;
        // jump the counter to this range: ctr0 + lo (128-bit little-endian add)
        uint64_t c0 = (uint64_t)mine.ctr[0] | (uint64_t)mine.ctr[1] << 32, c1 = (uint64_t)mine.ctr[2] | (uint64_t)mine.ctr[3] << 32;
        const uint64_t n0 = c0 + (uint64_t)lo;
        c1 += n0 < c0;
        mine.ctr[0] = (u32)n0; mine.ctr[1] = (u32)(n0 >> 32); mine.ctr[2] = (u32)c1; mine.ctr[3] = (u32)(c1 >> 32);
        const bool is_last = i == parts - 1;
        const err_t c = bee2hip_beltCTR_bulk(buf + 16 * lo, 16 * cnt + (is_last ? tail : 0), &mine);
        if (is_last) last = mine;
        return c;
    });
    if (code != ERR_OK) return code;
    *st = last;                                   // counter after all blocks, gamma of the final block, reserved
    return ERR_OK;
} B2H_CATCH

extern "C" err_t bee2hip_bignVerify_batch_multi(const bign_params *params, const octet oid_der[], size_t oid_len,
                                                const octet *hashes, const octet *sigs, const octet *pubkeys,
                                                size_t n, err_t *codes, int ndev)
try {
    // argument checks once, in bignVerify's order, through an empty single-device call
    err_t code = bee2hip_bignVerify_batch(params, oid_der, oid_len, hashes, sigs, pubkeys, 0, codes);
    if (code != ERR_OK) return code;
    if (n && (!hashes || !sigs || !pubkeys || !codes)) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    const size_t no = params->l / 4;
    return run_on_devices(ndev, [=](int i, int parts) {
        size_t lo, cnt;
        bee2hip_multi_plan(n, parts, i, &lo, &cnt);
        return bee2hip_bignVerify_batch(params, oid_der, oid_len, hashes + no * lo, sigs + (no + no / 2) * lo,
                                        pubkeys + 2 * no * lo, cnt, codes + lo);
    });
} B2H_CATCH

// one signer / a few signers over all GPUs: every device builds (and caches) the tables of the keys its part meets
extern "C" err_t bee2hip_bignVerify_keyed_batch_multi(const bign_params *params, const octet oid_der[], size_t oid_len,
                                                      const octet *hashes, const octet *sigs, const octet *pubkeys, size_t nkeys,
                                                      const u32 *key_index, size_t n, err_t *codes, int ndev)
try {
    err_t code = bee2hip_bignVerify_keyed_batch(params, oid_der, oid_len, hashes, sigs, pubkeys, nkeys, key_index, 0, codes);
    if (code != ERR_OK) return code;
    if (n && (!hashes || !sigs || !pubkeys || !nkeys || !key_index || !codes)) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    const size_t no = params->l / 4;
    return run_on_devices(ndev, [=](int i, int parts) {
        size_t lo, cnt;
        bee2hip_multi_plan(n, parts, i, &lo, &cnt);
        return bee2hip_bignVerify_keyed_batch(params, oid_der, oid_len, hashes + no * lo, sigs + (no + no / 2) * lo, pubkeys, nkeys,
                                              key_index + lo, cnt, codes + lo);
    });
} B2H_CATCH
extern "C" err_t bee2hip_bignVerify_onekey_batch_multi(const bign_params *params, const octet oid_der[], size_t oid_len,
                                                       const octet *hashes, const octet *sigs, const octet pubkey[], size_t n,
                                                       err_t *codes, int ndev)
try {
    err_t code = bee2hip_bignVerify_onekey_batch(params, oid_der, oid_len, hashes, sigs, pubkey, 0, codes);
    if (code != ERR_OK) return code;
    if (n && (!hashes || !sigs || !pubkey || !codes)) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    const size_t no = params->l / 4;
    return run_on_devices(ndev, [=](int i, int parts) {
        size_t lo, cnt;
        bee2hip_multi_plan(n, parts, i, &lo, &cnt);
        return bee2hip_bignVerify_onekey_batch(params, oid_der, oid_len, hashes + no * lo, sigs + (no + no / 2) * lo, pubkey, cnt, codes + lo);
    });
} B2H_CATCH

extern "C" err_t bee2hip_bignSign2_batch_multi(const bign_params *params, const octet oid_der[], size_t oid_len,
                                               const octet *hashes, const octet *privkeys, const void *t, size_t t_len,
                                               size_t n, octet *sigs, err_t *codes, int ndev)
try {
    err_t code = bee2hip_bignSign2_batch(params, oid_der, oid_len, hashes, privkeys, t, t_len, 0, sigs, codes);
    if (code != ERR_OK) return code;
    if (n && (!hashes || !privkeys || !sigs || !codes)) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    const size_t no = params->l / 4;
    return run_on_devices(ndev, [=](int i, int parts) {
        size_t lo, cnt;
        bee2hip_multi_plan(n, parts, i, &lo, &cnt);
        return bee2hip_bignSign2_batch(params, oid_der, oid_len, hashes + no * lo, privkeys + no * lo, t, t_len, cnt,
                                       sigs + (no + no / 2) * lo, codes + lo);
    });
} B2H_CATCH

extern "C" err_t bee2hip_bashHash_beltMAC_batch_multi(const octet *msgs, size_t msg_len, size_t n, size_t l,
                                                      const octet key[], size_t key_len, octet *digests, octet *tags,
                                                      int ndev)
try {
    err_t code = bee2hip_bashHash_beltMAC_batch(msgs, msg_len, 0, l, key, key_len, digests, tags);
    if (code != ERR_OK) return code;
    if (n && msg_len && !msgs) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    const size_t dlen = digests ? l / 4 : 0;
    return run_on_devices(ndev, [=](int i, int parts) {
        size_t lo, cnt;
        bee2hip_multi_plan(n, parts, i, &lo, &cnt);
        return bee2hip_bashHash_beltMAC_batch(msgs + msg_len * lo, msg_len, cnt, l, key, key_len,
                                              digests ? digests + dlen * lo : nullptr, tags ? tags + 8 * lo : nullptr);
    });
} B2H_CATCH

// ragged messages: ranges of whole messages with about equal BYTE counts (a range's cost is its bytes)
extern "C" err_t bee2hip_hash_ragged_multi(size_t alg, const octet *data, const uint64_t *offsets, size_t n,
                                           octet *digests, int ndev)
try {
    if (alg != 0 && alg != 128 && alg != 192 && alg != 256) return ERR_BAD_PARAMS;
    if (n == 0) return ERR_OK;
    if (!offsets || !digests) return ERR_BAD_INPUT;
    for (size_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return ERR_BAD_INPUT;
    if (offsets[n] != offsets[0] && !data) return ERR_BAD_INPUT;
    const size_t dlen = alg ? alg / 4 : 32;
    int parts = ndev > 0 ? ndev : bee2hip_device_count();
    if (parts <= 0) return hip_fail(hipErrorNoDevice, "hipGetDeviceCount");
    try {
    // cut[i] = first message of range i: the first index whose start offset reaches i / parts of the bytes
    std::vector<size_t> cut((size_t)parts + 1, n);
    cut[0] = 0;
    const uint64_t base = offsets[0], total = offsets[n] - base;
    size_t m = 0;
    for (int i = 1; i < parts; ++i) {
        const uint64_t want = base + total / (uint64_t)parts * (uint64_t)i;
        while (m < n && offsets[m] < want) ++m;
        cut[(size_t)i] = m;
    }
    return run_on_devices(parts, [&](int i, int) {
        const size_t lo = cut[(size_t)i], hi = cut[(size_t)i + 1];
        if (hi <= lo) return (err_t)ERR_OK;
        std::vector<uint64_t> off(hi - lo + 1);
        for (size_t k = lo; k <= hi; ++k) off[k - lo] = offsets[k] - offsets[lo];
        return bee2hip_hash_ragged(alg, data + offsets[lo], off.data(), hi - lo, digests + dlen * lo);
    });
    } catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }      // (the per-range vectors are guarded in the worker)
} B2H_CATCH

// ---- shards that already live on the devices -------------------------------------------------
// A caller whose data is resident -- shard i in the memory of device i -- must not be pushed through PCIe twice.  These
// take one device pointer and one count per device; worker i (bound to device i) launches the single-device _dev entry on
// its OWN non-blocking stream -- ordered behind whatever the device's NULL stream had seen when the job started -- and drains
// it, so the call returns when every device is done and logical devices that share a card overlap.  Nothing crosses devices; the parameter
// block (key, counter, curve, OID) is an argument.  ndev = entries in the arrays (<= bee2hip_device_count()).
static err_t multi_dev_args(const void *const ptrs, const size_t *counts, int ndev)
{
    if (ndev <= 0 || !ptrs || !counts) return ERR_BAD_INPUT;
    if (ndev > bee2hip_device_count()) return ERR_BAD_INPUT;
    return ERR_OK;
}
// The worker's stream, ordered behind everything the device's NULL stream has seen so far -- the guarantee the NULL-stream launches
// of rounds 3-4 gave a caller who produced his shards on the default stream (or on blocking streams) without synchronising.
static inline err_t worker_stream(hipStream_t *st)
{
    *st = t_wstream;
    if (!t_wstream) return ERR_OK;                          // (not a pool worker: the NULL stream, as before)
    B2H_TRY(hipEventRecord(t_wevent, nullptr));
    B2H_TRY(hipStreamWaitEvent(t_wstream, t_wevent, 0));
    return ERR_OK;
}
static inline err_t drain(err_t code)
{
    // ALWAYS drained: when the launcher reports an error, kernels it had already queued on the worker's private stream may
    // still be reading or writing the caller's shards -- the caller cannot name that stream, so the call must not return before it is idle
    const hipError_t e = hipStreamSynchronize(t_wstream);
    if (code != ERR_OK) { (void)hipGetLastError(); return code; }
    B2H_TRY(e);
    return ERR_OK;
}
#define B2H_WSTREAM(st) hipStream_t st = nullptr; { const err_t c_ = worker_stream(&st); if (c_ != ERR_OK) return c_; }

extern "C" err_t bee2hip_bashF_batch_multi_dev(void *const d_states[], const size_t counts[], int ndev)
try {
    err_t code = multi_dev_args(d_states, counts, ndev);
    if (code != ERR_OK) return code;
    return run_on_devices(ndev, [=](int i, int) -> err_t {
        B2H_WSTREAM(st);
        return drain(bee2hip_bashF_batch_dev(d_states[i], counts[i], st));
    });
} B2H_CATCH

// shard i holds blocks [first_block + sum of nblocks[0 .. i), ...) of ONE stream that started at ctr0 (beltCTRStart's E_K(iv)):
// lanes compute ctr0 + offset + 1 directly, no state passes between devices
extern "C" err_t bee2hip_beltCTR_blocks_multi_dev(void *const d_bufs[], const size_t nblocks[], const u32 key[8],
                                                  const u32 ctr0[4], uint64_t first_block, int ndev)
try {
    err_t code = multi_dev_args(d_bufs, nblocks, ndev);
    if (code != ERR_OK) return code;
    if (!key || !ctr0) return ERR_BAD_INPUT;
    uint64_t first[64];
    if (ndev > 64) return ERR_BAD_INPUT;
    for (int i = 0; i < ndev; ++i) { first[i] = first_block; first_block += (uint64_t)nblocks[i]; }
    return run_on_devices(ndev, [=](int i, int) -> err_t {
        B2H_WSTREAM(st);
        return drain(bee2hip_beltCTR_blocks_dev(d_bufs[i], nblocks[i], key, ctr0, first[i], st));
    });
} B2H_CATCH

extern "C" err_t bee2hip_bignVerifyL_batch_multi_dev(size_t l, const octet oid_der[], size_t oid_len, const void *const d_hashes[],
                                                     const void *const d_sigs[], const void *const d_pubkeys[],
                                                     const size_t counts[], void *const d_codes[], int ndev)
try {
    err_t code = multi_dev_args(d_hashes, counts, ndev);
    if (code != ERR_OK) return code;
    if (!d_sigs || !d_pubkeys || !d_codes) return ERR_BAD_INPUT;
    // argument checks once (level, OID), in the single-device entry's order, through an empty call
    code = bee2hip_bignVerifyL_batch_dev(l, oid_der, oid_len, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
    if (code != ERR_OK) return code;
    return run_on_devices(ndev, [=](int i, int) -> err_t {
        B2H_WSTREAM(st);
        return drain(bee2hip_bignVerifyL_batch_dev(l, oid_der, oid_len, d_hashes[i], d_sigs[i], d_pubkeys[i], counts[i], d_codes[i], st));
    });
} B2H_CATCH

extern "C" err_t bee2hip_bignVerifyL_onekey_batch_multi_dev(size_t l, const octet oid_der[], size_t oid_len, const void *const d_hashes[],
                                                            const void *const d_sigs[], const octet pubkey[], const size_t counts[],
                                                            void *const d_codes[], int ndev)
try {
    err_t code = multi_dev_args(d_hashes, counts, ndev);
    if (code != ERR_OK) return code;
    if (!d_sigs || !pubkey || !d_codes) return ERR_BAD_INPUT;
    code = bee2hip_bignVerifyL_onekey_batch_dev(l, oid_der, oid_len, nullptr, nullptr, pubkey, 0, nullptr, nullptr);
    if (code != ERR_OK) return code;
    return run_on_devices(ndev, [=](int i, int) -> err_t {
        B2H_WSTREAM(st);
        return drain(bee2hip_bignVerifyL_onekey_batch_dev(l, oid_der, oid_len, d_hashes[i], d_sigs[i], pubkey, counts[i], d_codes[i], st));
    });
} B2H_CATCH
extern "C" err_t bee2hip_bignVerifyL_keyed_batch_multi_dev(size_t l, const octet oid_der[], size_t oid_len, const void *const d_hashes[],
                                                           const void *const d_sigs[], const octet pubkeys[], size_t nkeys,
                                                           const void *const d_key_index[], const size_t counts[], void *const d_codes[],
                                                           int ndev)
try {
    err_t code = multi_dev_args(d_hashes, counts, ndev);
    if (code != ERR_OK) return code;
    if (!d_sigs || !pubkeys || !nkeys || !d_key_index || !d_codes) return ERR_BAD_INPUT;
    code = bee2hip_bignVerifyL_keyed_batch_dev(l, oid_der, oid_len, nullptr, nullptr, pubkeys, nkeys, nullptr, 0, nullptr, nullptr);
    if (code != ERR_OK) return code;
    return run_on_devices(ndev, [=](int i, int) -> err_t {
        B2H_WSTREAM(st);
        return drain(bee2hip_bignVerifyL_keyed_batch_dev(l, oid_der, oid_len, d_hashes[i], d_sigs[i], pubkeys, nkeys, d_key_index[i], counts[i],
                                                         d_codes[i], st));
    });
} B2H_CATCH

extern "C" err_t bee2hip_bashHash_beltMAC_batch_multi_dev(const void *const d_msgs[], size_t msg_len, const size_t counts[], size_t l,
                                                          const octet key[], size_t key_len, void *const d_digests[],
                                                          void *const d_tags[], int ndev)
try {
    err_t code = multi_dev_args(d_msgs, counts, ndev);
    if (code != ERR_OK) return code;
    code = bee2hip_bashHash_beltMAC_batch_dev(nullptr, msg_len, 0, l, key, key_len, d_digests ? (void *)1 : nullptr,
                                              d_tags ? (void *)1 : nullptr, nullptr);
    if (code != ERR_OK) return code;
    return run_on_devices(ndev, [=](int i, int) -> err_t {
        B2H_WSTREAM(st);
        return drain(bee2hip_bashHash_beltMAC_batch_dev(d_msgs[i], msg_len, counts[i], l, key, key_len,
                                                        d_digests ? d_digests[i] : nullptr, d_tags ? d_tags[i] : nullptr, st));
    });
} B2H_CATCH
