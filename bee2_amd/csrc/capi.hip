// capi.hip -- the C ABI of libbee2hip.so (declared in include/bee2hip.h).
//
// Host side of the engine: argument checks and state bookkeeping mirror bee2's
// C functions line for line in *behaviour* (same names, same state layouts, same
// error codes).  Every batch / _dev / _multi entry point and every bign operation evaluates its primitives in
// kernels; the bee2 drop-in symbols do so too, except for small single calls, which take the host path of
// host_small.hpp ("host path for small single calls" below says exactly when).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <new>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.hpp"
#include "host_small.hpp"
#include "host_bign.hpp"
#include "host_bign_ct.hpp"
#include "bign_curves.inc"   // (#pragma once: shared with bign_kernels.hip in the unity build)

namespace bee2hip {

// ------------------------------------------------------------------ errors ---
static thread_local char t_err[256] = "";

err_t hip_fail(hipError_t e, const char *what)
{
    snprintf(t_err, sizeof t_err, "%s: %s", what, hipGetErrorString(e));
    // HIP keeps the failure as this thread's "last error"; a later hipGetLastError() (the launchers check
    // it after every kernel launch) would blame an unrelated call for it.  The error has been reported: clear it.
    (void)hipGetLastError();
    return ERR_BEE2HIP_DEVICE;
}
// the same for failures the library maps to a bee2 error code itself (a refused allocation)
static inline err_t out_of_memory()
{
    (void)hipGetLastError();
    return ERR_OUTOFMEMORY;
}

// ---------------------------------------------------------- per-device init ---
// the belt S-box from the standard's own generator (belt_block.c:21-35): an 8-bit
// LFSR stepped 116 times per entry, anchored at H[10] = 0x00, H[11] = 0x8E.
static uint8_t g_H[256];
static std::once_flag g_H_once;
static void gen_H()
{
    g_H[10] = 0x00; g_H[11] = 0x8E;
    for (unsigned x = 12; x < 10 + 256; ++x) {
        unsigned t = g_H[(x - 1) % 256];
        for (int i = 0; i < 116; ++i) t = (t >> 1) | ((unsigned)__builtin_parity(t & 0x63) << 7);
        g_H[x % 256] = (uint8_t)t;
    }
}
const uint8_t *host_beltH()
{
    std::call_once(g_H_once, gen_H);
    return g_H;
}


constexpr int MAX_DEV = 64;
static std::mutex g_dev_mu;
static bool g_dev_ready[MAX_DEV];

// make sure the current device has its constants (S-box, curve tables)
err_t ensure_device()
{
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) return ERR_BAD_INPUT;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_dev_ready[dev]) return ERR_OK;
    err_t code = upload_beltH(host_beltH());
    if (code == ERR_OK) code = upload_beltH_bign(host_beltH());
    if (code != ERR_OK) return code;
    g_dev_ready[dev] = true;
    return ERR_OK;
}

// Device scratch of the launchers (tweak tables, partial sums, the verify pipeline's SoA arrays), keyed by
// (device, stream, slot): work queued on one stream is ordered, so one buffer per stream is enough.  The
// NULL stream is the exception -- every thread of the host-pointer / drop-in API launches on it, and
// thread B's first kernel may run between thread A's first and second -- so there the key also carries
// the calling thread.  (A caller who drives one non-null stream from several threads at once has to
// serialise them himself, as for any stream.)  Stream-keyed buffers live until process exit; the NULL-stream
// buffers of a thread are released when that thread exits (ThreadReaper below) -- a thread-per-request caller
// of the drop-in API must not accumulate device memory.
struct PoolEntry { int dev; hipStream_t st; int slot; unsigned tid; void *p; size_t bytes; };
static std::mutex g_pool_mu;
static std::vector<PoolEntry> g_pool;
static std::atomic<unsigned> g_next_tid{1};
static thread_local unsigned t_tid = 0;

// The thread that loaded the library (normally the main thread) runs its thread_local destructors during process
// teardown, when the HIP runtime may already be unusable: it leaks on purpose.  Every other thread exits while
// the runtime is alive and frees what it owns.
static const std::thread::id g_loader_thread = std::this_thread::get_id();
static bool on_loader_thread() { return std::this_thread::get_id() == g_loader_thread; }
struct ThreadReaper {
    void touch() {}
    ~ThreadReaper()
    {
        if (on_loader_thread() || t_tid == 0) return;
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size();) {
            if (g_pool[i].tid == t_tid) {
                if (g_pool[i].p) (void)hipFree(g_pool[i].p);      // the thread's calls were synchronous: nothing is in flight
                g_pool[i] = g_pool.back();
                g_pool.pop_back();
            } else ++i;
        }
    }
};
static thread_local ThreadReaper t_reaper;

err_t scratch_for_stream(hipStream_t st, int slot, size_t bytes, void **out)
{
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    unsigned tid = 0;
    if (st == nullptr) {
        if (t_tid == 0) { t_tid = g_next_tid.fetch_add(1); t_reaper.touch(); }
        tid = t_tid;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    PoolEntry *e = nullptr;
    for (PoolEntry &x : g_pool)
        if (x.dev == dev && x.st == st && x.slot == slot && x.tid == tid) { e = &x; break; }
    if (!e) {
        g_pool.push_back(PoolEntry{dev, st, slot, tid, nullptr, 0});
        e = &g_pool.back();
    }
    if (e->bytes < bytes) {
        if (e->p) {
            B2H_TRY(hipStreamSynchronize(st));            // earlier batches may still use the old block
            (void)hipFree(e->p);
            e->p = nullptr; e->bytes = 0;
        }
        if (hipMalloc(&e->p, bytes) != hipSuccess) { e->p = nullptr; return out_of_memory(); }
        e->bytes = bytes;
    }
    *out = e->p;
    return ERR_OK;
}

// Entries keyed on a stream the LIBRARY owns (the per-thread duplex streams below) must go when that stream goes: nothing else
// would ever free them (ThreadReaper only knows the NULL-stream entries of its thread), and a later stream that got the same
// handle value would inherit a stale block.  The stream is drained first.
static void scratch_release_stream(hipStream_t st)
{
    if (!st) return;
    (void)hipStreamSynchronize(st);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size();) {
        if (g_pool[i].st == st) {
            if (g_pool[i].p) (void)hipFree(g_pool[i].p);
            g_pool[i] = g_pool.back();
            g_pool.pop_back();
        } else ++i;
    }
}

// scratch device buffer for the host-pointer API, grown on demand, per thread
// Staging for the host-pointer entry points, per thread and slot.  Small requests (<= 64 KiB: every drop-in call on a
// block, a state, a signature ...) are served from a PINNED, device-mapped host buffer: the caller's bytes are copied
// into it by the CPU, the kernels read and write it across PCIe, and the result is copied out after one stream
// synchronise -- no hipMemcpy at all (a hipMemcpy of a few bytes costs ~10 us each way; bashF() went from 33 to
// ~15 us per call).  Larger requests use device memory and hipMemcpy as before.  h2d() / d2h() below pick the path
// from the pointer.
constexpr size_t PINNED_MAX = 64 * 1024;      // size of the pinned buffer
static size_t g_pinned_limit = PINNED_MAX;     // requests up to this size use it (bee2hip_internal_tune(3, bytes): A/B)
struct Scratch {
    void *p = nullptr;          // what the current request uses: pin or devp
    void *pin = nullptr;        // PINNED_MAX bytes of mapped host memory, allocated on first small request
    void *devp = nullptr;
    size_t cap = 0;             // of devp
    int dev = -1;
    // `chain` = the kernel walks the input as one dependent chain on a lane or two (sponge absorption, the belt-hash
    // iteration): there every load is a PCIe round trip on the critical path, and pinned staging only pays below ~2 KiB
    // (tools/pinned_ab.py: belt-hash of 16 KiB 2.49 ms pinned vs 2.23 ms copied; of 1 KiB 186 vs 200 us)
    err_t need(size_t n, bool chain = false)
    {
        int cur = 0;
        B2H_TRY(hipGetDevice(&cur));
        if (n <= (chain && g_pinned_limit > 2048 ? (size_t)2048 : g_pinned_limit)) {
            if (!pin && hipHostMalloc(&pin, PINNED_MAX, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
                pin = nullptr;
                return out_of_memory();
            }
            p = pin;
            return ERR_OK;
        }
        if (devp && (cur != dev || cap < n)) { (void)hipFree(devp); devp = nullptr; cap = 0; }
        if (!devp) {
            if (hipMalloc(&devp, n) != hipSuccess) { devp = nullptr; return out_of_memory(); }
            cap = n; dev = cur;
        }
        p = devp;
        return ERR_OK;
    }
    ~Scratch()
    {
        // thread exit: give the blocks back, except on the loader thread (process teardown, see ThreadReaper)
        if (!on_loader_thread()) {
            if (devp) (void)hipFree(devp);
            if (pin) (void)hipHostFree(pin);
        }
        p = devp = pin = nullptr;
    }
};
static thread_local Scratch t_scr[4];

static inline bool in_pinned(const void *q)
{
    for (const Scratch &sc : t_scr)
        if (sc.pin && (const char *)q >= (const char *)sc.pin && (const char *)q < (const char *)sc.pin + PINNED_MAX) return true;
    return false;
}
// host -> staging.  Pinned: the NULL stream is idle here (every host entry point ends with d2h or a synchronise), and
// a kernel launched afterwards sees what the CPU wrote.
static inline hipError_t h2d(void *d, const void *h, size_t n)
{
    if (in_pinned(d)) { memcpy(d, h, n); return hipSuccess; }
    return hipMemcpy(d, h, n, hipMemcpyHostToDevice);
}
// staging -> host, after everything queued on the NULL stream
static inline hipError_t d2h(void *h, const void *d, size_t n)
{
    if (in_pinned(d)) {
        const hipError_t e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) return e;
        memcpy(h, d, n);
        return hipSuccess;
    }
    return hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
}
static inline hipError_t zero_staging(void *d, size_t n)
{
    if (in_pinned(d)) {
        const hipError_t e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) return e;
        memset(d, 0, n);
        return hipSuccess;
    }
    return hipMemset(d, 0, n);
}


// ------------------------------------------------- host path for small single calls ---
// host_small.hpp has the what and why.  Who runs where:
//   BEE2HIP_FORCE=gpu   every drop-in call evaluates its primitives on the GPU (rounds 1-2 behaviour; a device failure
//                       inside a void function aborts with a message)
//   BEE2HIP_FORCE=cpu   every drop-in call that has a host path takes it, whatever its size (tests run the fixtures so)
//   unset (auto)        by crossover: single primitives (bashF, one block), block-parallel modes below 8 KiB per call
//                       and the serial chains of ONE message (sponge, CBC-MAC, belt-hash, CBC encryption, a belt-sde
//                       sector: one lane of the GPU runs them at 3-7 MB/s, a host core at 60-170 MB/s) on the host;
//                       and ONE signature verification on a standard curve (host_bign.hpp: ~40 us against a 0.4 ms
//                       call through the GPU); everything else, every bign operation that touches a private or
//                       one-time key, and EVERY batch / _dev / _multi entry point on the GPU.
// In every mode the calling thread must have initialised its HIP device first (ensure_device): without a GPU the
// library fails exactly as before.  In auto mode a GPU path that fails twice (once more after hipDeviceSynchronize) is
// finished on the host with a warning on stderr instead of abort() -- bee2's Step functions cannot report errors and a
// long-running service must survive a transient device fault.
enum { FORCE_AUTO = 0, FORCE_GPU = 1, FORCE_CPU = 2 };
enum { K_PRIM = 0, K_PARALLEL = 1, K_SERIAL = 2, K_POLY = 3, K_VERIFY1 = 4, K_SIGN1 = 5 };
static std::atomic<int> g_force{-1};
static std::atomic<unsigned long long> g_n_host{0}, g_n_gpu{0}, g_n_fallback{0};
static std::atomic<int> g_inject_fail{0};                  // tests: make the next n GPU attempts of a drop-in helper fail
static hostp::BeltTables g_hostT;
static std::once_flag g_hostT_once;
static const hostp::BeltTables &hostT()
{
    std::call_once(g_hostT_once, [] { hostp::belt_tables(g_hostT, host_beltH()); });
    return g_hostT;
}
// a call that hashes a SECRET through the drop-in's own streaming functions pins the path to the GPU for its duration
// (ForceScope): the host path's table-driven belt is not constant-time and keeps its temporaries
static thread_local int t_force_scope = -1;
struct ForceScope {
    int old;
    explicit ForceScope(int m) : old(t_force_scope) { t_force_scope = m; }
    ~ForceScope() { t_force_scope = old; }
};
// a wipe the optimiser may not drop (the buffer dies right afterwards)
static inline void wipe_host(void *p, size_t n)
{
    volatile unsigned char *q = (volatile unsigned char *)p;
    while (n--) *q++ = 0;
}
static int force_mode()
{
    if (t_force_scope >= 0) return t_force_scope;
    int m = g_force.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *e = getenv("BEE2HIP_FORCE");
        m = !e ? FORCE_AUTO : !strcmp(e, "gpu") ? FORCE_GPU : !strcmp(e, "cpu") ? FORCE_CPU : FORCE_AUTO;
        g_force.store(m);
    }
    return m;
}
static bool host_wanted(int kind, size_t bytes)
{
    const int m = force_mode();
    if (m == FORCE_GPU) return false;
    if (m == FORCE_CPU) return true;
    switch (kind) {
    case K_PRIM: return bytes <= 1024;          // one permutation / up to 64 blocks: 0.3-0.5 us each vs ~20 us per launch
    case K_PARALLEL: return bytes < 8192;       // INTEGRATION.md crossover table (CTR: 16 KiB 36 us vs 79 us on one core)
    case K_POLY: return bytes <= (hostp::gf_have_clmul() ? (size_t)32768 : (size_t)4096);   // host product: 7 ns per block with PCLMULQDQ (2.2 GB/s), 60 ns by table; a GPU call is ~30 us
    case K_VERIFY1: return true;                // one signature: ~40 us on a core vs ~0.4 ms through the GPU
    case K_SIGN1: return true;                  // one key pair / signature: ~30 us in constant-time host arithmetic (host_bign_ct.hpp) vs ~190 us
    default: return true;                       // K_SERIAL: one message = one dependent chain
    }
}
static thread_local bool t_dev_seen = false;
static inline err_t device_seen()
{
    if (t_dev_seen) return ERR_OK;
    const err_t code = ensure_device();
    if (code == ERR_OK) t_dev_seen = true;
    return code;
}
// run a drop-in helper: `gpu` stages, launches and copies back (returns err_t, leaves the caller's data untouched when it
// fails); `host` does the same work with host_small.hpp
template <class G, class H>
static err_t with_host(int kind, size_t bytes, const char *what, G gpu, H host)
{
    err_t code = device_seen();
    if (code != ERR_OK) return code;            // no usable GPU: an error (void callers: die_on), never a silent CPU run
    if (host_wanted(kind, bytes)) { host(); g_n_host.fetch_add(1, std::memory_order_relaxed); return ERR_OK; }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (g_inject_fail.load(std::memory_order_relaxed) > 0 && g_inject_fail.fetch_sub(1) > 0)
            code = ERR_BEE2HIP_DEVICE;
        else
            code = gpu();
        if (code == ERR_OK) { g_n_gpu.fetch_add(1, std::memory_order_relaxed); return ERR_OK; }
        if (code != ERR_BEE2HIP_DEVICE) return code;         // bad input, out of memory: report, nothing to retry
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
    }
    if (force_mode() == FORCE_GPU) return code;
    fprintf(stderr, "libbee2hip: %s: device path failed twice (%s); finished on the host\n", what, t_err);
    host();
    g_n_fallback.fetch_add(1, std::memory_order_relaxed);
    return ERR_OK;
}

// ---------------------------------------------- duplex staging of large in-place host batches ---
// PCIe is full duplex and this box's two SDMA directions do run side by side -- 53 GiB/s each way alone, 87-90 GiB/s
// together -- but only for copies issued with hipMemcpyAsync on two non-blocking streams, and, the caller's buffers being
// ordinary pageable memory (an async copy of pageable memory holds its calling thread), from two host threads
// (tools/ubench/pcie_duplex.hip, profiles/r03_pcie_duplex.txt: blocking hipMemcpy from two threads serialises, 48 GiB/s).
// A large in-place batch is therefore cut into chunks: the calling thread uploads chunk c and queues its kernel behind
// the copy on the same stream; a helper thread downloads chunk c - 1 on a second stream as soon as its kernel is through.
// launch(dev_chunk, first_unit, units, stream) queues the kernel(s) for `units` units starting at unit `first_unit`.
constexpr size_t DUPLEX_MIN = (size_t)48 << 20;          // below this the two copies cost < 2 ms: not worth a thread
static int g_duplex_log2_states = 16, g_duplex_log2_blocks = 20;   // chunk sizes (bee2hip_internal_tune 6 / 7: sweep)
constexpr size_t VERIFY_PIPE_MIN = (size_t)1 << 19, VERIFY_PIPE_CHUNK = (size_t)1 << 18;   // host-pointer verification batches
static int g_verify_pipe = 1;                                      // (tune 11: A/B)
static int g_duplex_ramp = 0;                                      // quarter / half chunks at both ends (tune 9): measured -2 %, off
#ifdef BEE2HIP_EXPERIMENTS
static std::atomic<int> g_duplex_fail_chunk{0}, g_duplex_fail_times{0};   // tests (tune 14 / 15): the next `times` pipelines fail at chunk `chunk`
#endif
struct DuplexStreams {
    hipStream_t up = nullptr, dn = nullptr;
    int dev = -1;
    err_t get()
    {
        int cur = 0;
        B2H_TRY(hipGetDevice(&cur));
        if (up && cur == dev) return ERR_OK;
        if (up) { drop(); }
        B2H_TRY(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        B2H_TRY(hipStreamCreateWithFlags(&dn, hipStreamNonBlocking));
        dev = cur;
        return ERR_OK;
    }
    // the launchers' scratch keyed on these streams (a 2^18-signature chunk of the verification pipeline: 275-550 MB) goes with them
    void drop()
    {
        scratch_release_stream(up);
        scratch_release_stream(dn);
        if (up) (void)hipStreamDestroy(up);
        if (dn) (void)hipStreamDestroy(dn);
        up = dn = nullptr;
    }
    ~DuplexStreams() { if (up && !on_loader_thread()) drop(); }
};
static thread_local DuplexStreams t_duplex;

// A second queue of the calling thread on the current device, with the two events of a fork / join around it (common.hpp
// side_stream): launchers whose two kernels are independent put the second one there -- launch_hash_ragged's long chains
// (latency-bound, a few wavefronts) and its short messages (throughput-bound) then share the chip instead of queueing.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int dev = -1;
    void drop()
    {
        if (s) { scratch_release_stream(s); (void)hipStreamDestroy(s); }
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        s = nullptr; fork = join = nullptr;
    }
    err_t get()
    {
        int cur = 0;
        B2H_TRY(hipGetDevice(&cur));
        if (s && cur == dev) return ERR_OK;
        drop();
        B2H_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        B2H_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        B2H_TRY(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        dev = cur;
        return ERR_OK;
    }
    ~SideStream() { if (s && !on_loader_thread()) drop(); }
};
static thread_local SideStream t_side;
err_t side_stream(hipStream_t *side, hipEvent_t *fork, hipEvent_t *join)
{
    const err_t code = t_side.get();
    if (code != ERR_OK) { t_side.drop(); return code; }
    *side = t_side.s; *fork = t_side.fork; *join = t_side.join;
    return ERR_OK;
}

// *done_units (may be null) = leading units whose results are back in the caller's buffer when the call returns: all of
// them on success; after a failure the chunks whose download had completed.  A caller that retries or finishes on the host
// MUST skip them -- they have been transformed in place already (a second CTR pass would decrypt them again).
template <class Launch>
static err_t duplex_inplace(octet *host, octet *dev, size_t unit_bytes, size_t units, size_t chunk_units, Launch launch,
                            size_t *done_units = nullptr)
{
    if (done_units) *done_units = 0;
    err_t code = t_duplex.get();
    if (code != ERR_OK) return code;
    // chunk boundaries: full chunks, with a quarter and a half chunk at either end when there are enough of them -- the first
    // upload and the last download are the only transfers with nothing in the other direction beside them (knob 9)
    std::vector<size_t> cut;
    {
        const size_t q = chunk_units / 4, h = chunk_units / 2;
        const bool ramp = g_duplex_ramp && q && units >= 6 * chunk_units;
        size_t pos = 0;
        cut.push_back(0);
        if (ramp) { cut.push_back(pos += q); cut.push_back(pos += h); }
        const size_t tail = ramp ? q + h : 0;
        while (units - pos > chunk_units + tail) cut.push_back(pos += chunk_units);
        if (ramp) {
            const size_t rest = units - pos - tail;       // <= chunk_units, > 0
            cut.push_back(pos += rest);
            cut.push_back(pos += h);
        }
        cut.push_back(units);
    }
    const size_t nch = cut.size() - 1;
    std::vector<hipEvent_t> ev(2 * nch, nullptr);         // [c] kernel of chunk c queued behind its upload; [nch + c] its download queued
    for (size_t c = 0; c < 2 * nch; ++c)
        if (hipEventCreateWithFlags(&ev[c], hipEventDisableTiming) != hipSuccess) {
            for (size_t k = 0; k < c; ++k) (void)hipEventDestroy(ev[k]);
            return hip_fail(hipGetLastError(), "hipEventCreate");
        }
    std::atomic<size_t> queued{0}, dn_queued{0};
    std::atomic<int> failed{0};
    std::atomic<int> first_err{(int)hipSuccess};           // the first failing hipError_t of either thread
    const auto fail = [&](hipError_t e) {
        int ok = (int)hipSuccess;
        first_err.compare_exchange_strong(ok, (int)(e == hipSuccess ? hipErrorUnknown : e));
        failed.store(1);
    };
    int devno = 0;
    (void)hipGetDevice(&devno);
    const hipStream_t sup = t_duplex.up, sdn = t_duplex.dn;
    std::thread down([&] {
        hipError_t e = hipSetDevice(devno);
        if (e != hipSuccess) { fail(e); return; }
        for (size_t c = 0; c < nch; ++c) {
            while (queued.load(std::memory_order_acquire) <= c) {
                if (failed.load()) return;                 // (the caller drains sdn before it returns)
                std::this_thread::yield();
            }
            const size_t first = cut[c], cnt = cut[c + 1] - first;
            if ((e = hipStreamWaitEvent(sdn, ev[c], 0)) != hipSuccess ||
                (e = hipMemcpyAsync(host + first * unit_bytes, dev + first * unit_bytes, cnt * unit_bytes, hipMemcpyDeviceToHost, sdn)) != hipSuccess ||
                (e = hipEventRecord(ev[nch + c], sdn)) != hipSuccess) {
                fail(e);
                return;
            }
            dn_queued.store(c + 1, std::memory_order_release);
        }
    });
#ifdef BEE2HIP_EXPERIMENTS
    const size_t inject_at = g_duplex_fail_times.load() > 0 && g_duplex_fail_times.fetch_sub(1) > 0 ? (size_t)g_duplex_fail_chunk.load() : 0;
#endif
    for (size_t c = 0; c < nch && !failed.load(); ++c) {
        const size_t first = cut[c], cnt = cut[c + 1] - first;
#ifdef BEE2HIP_EXPERIMENTS
        if (inject_at && c + 1 == (inject_at < nch ? inject_at : nch)) { fail(hipErrorUnknown); break; }   // a device fault in mid-pipeline
#endif
        hipError_t e = hipMemcpyAsync(dev + first * unit_bytes, host + first * unit_bytes, cnt * unit_bytes, hipMemcpyHostToDevice, sup);
        if (e != hipSuccess) { fail(e); break; }
        code = launch(dev + first * unit_bytes, first, cnt, sup);
        if (code != ERR_OK) { failed.store(1); break; }
        if ((e = hipEventRecord(ev[c], sup)) != hipSuccess) { fail(e); break; }
        queued.store(c + 1, std::memory_order_release);
    }
    down.join();
    // both streams are drained on EVERY path before the events go and the caller sees its buffer again: no copy into the
    // caller's memory may still be in flight after this function has returned
    hipError_t e = hipStreamSynchronize(sdn);
    if (e != hipSuccess) fail(e);
    e = hipStreamSynchronize(sup);
    if (e != hipSuccess) fail(e);
    size_t done = 0;
    {
        const size_t nq = dn_queued.load(std::memory_order_acquire);
        while (done < nq && hipEventQuery(ev[nch + done]) == hipSuccess) ++done;
    }
    if (done_units) *done_units = failed.load() ? cut[done] : units;
    for (size_t c = 0; c < 2 * nch; ++c) (void)hipEventDestroy(ev[c]);
    if (failed.load()) {
        (void)hipGetLastError();
        return code != ERR_OK ? code : hip_fail((hipError_t)first_err.load(), "duplex staging");
    }
    return ERR_OK;
}

}  // namespace bee2hip

using namespace bee2hip;

// The kernels read blocks / states / field elements as 16-byte vectors: a misaligned device pointer
// would be a GPU memory fault, so the _dev entry points refuse it with ERR_BAD_INPUT instead.
static inline bool misaligned(const void *p, size_t a) { return p && ((uintptr_t)p & (a - 1)) != 0; }

// ============================================================== management ===
extern "C" err_t bee2hip_set_device(int device)
{
    B2H_TRY(hipSetDevice(device));
    return ensure_device();
}
extern "C" err_t bee2hip_sync(void *stream)
{
    B2H_TRY(hipStreamSynchronize(as_stream(stream)));
    return ERR_OK;
}
extern "C" const char *bee2hip_last_error(void) { return t_err; }
extern "C" const char *bee2hip_version(void) { return "bee2hip 0.1 gfx950"; }

// ===================================================== device-pointer batch ===
extern "C" err_t bee2hip_bashF_batch_dev(void *d_states, size_t n, void *stream)
{
    if (misaligned(d_states, 16)) return ERR_BAD_INPUT;
    if (n && !d_states) return ERR_BAD_INPUT;
    return launch_bashF_batch(d_states, n, as_stream(stream));
}

extern "C" err_t bee2hip_beltCTR_blocks_dev(void *d_buf, size_t nblocks, const u32 key[8],
                                            const u32 ctr0[4], uint64_t first_block, void *stream)
{
    if (misaligned(d_buf, 16)) return ERR_BAD_INPUT;
    if ((nblocks && !d_buf) || !key || !ctr0) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_ctr_blocks(d_buf, nblocks, key, ctr0, first_block, nullptr, as_stream(stream));
}

extern "C" err_t bee2hip_beltBlockEncr_dev(void *d_blocks, size_t nblocks, const u32 key[8], void *stream)
{
    if (misaligned(d_blocks, 16)) return ERR_BAD_INPUT;
    if ((nblocks && !d_blocks) || !key) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_encr_blocks(d_blocks, nblocks, key, as_stream(stream));
}

// ======================================================= host-pointer batch ===
extern "C" err_t bee2hip_bashF_batch(octet *states, size_t n)
{
    if (n == 0) return ERR_OK;
    if (!states) return ERR_BAD_INPUT;
    Scratch &s = t_scr[0];
    err_t code = s.need(n * 192);
    if (code != ERR_OK) return code;
    if (n * 192 >= DUPLEX_MIN)              // chunks of 2^16 states = 12 MiB: upload, permute and download overlap
        return duplex_inplace(states, (octet *)s.p, 192, n, (size_t)1 << g_duplex_log2_states,
                              [](octet *d, size_t, size_t cnt, hipStream_t st) { return launch_bashF_batch(d, cnt, st); });
    B2H_TRY(h2d(s.p, states, n * 192));
    code = launch_bashF_batch(s.p, n, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(states, s.p, n * 192));
    return ERR_OK;
}

// E_K over host blocks (n small): the only way the drop-in layer evaluates belt
static err_t encr_host_blocks(uint32_t *blocks, size_t n, const u32 key[8])
{
    return with_host(K_PRIM, n * 16, "belt block encryption", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[1];
        code = s.need(n * 16);
        if (code != ERR_OK) return code;
        B2H_TRY(h2d(s.p, blocks, n * 16));
        code = launch_belt_encr_blocks(s.p, n, key, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(blocks, s.p, n * 16));
        return ERR_OK;
    }, [&] { for (size_t i = 0; i < n; ++i) hostp::belt_encr(hostT(), blocks + 4 * i, key); });
}

// a device failure inside a void bee2 function cannot be reported through the bee2
// signature: fail loudly instead of returning wrong bytes.
static void die_on(err_t code, const char *where)
{
    if (code == ERR_OK) return;
    fprintf(stderr, "libbee2hip: %s failed (err %u): %s\n", where, (unsigned)code, t_err);
    abort();
}

// =================================================================== bash ====
extern "C" const char bash_platform[] = "BASH_HIP_GFX950";

extern "C" void bashF(octet block[192], void *stack)
{
    (void)stack;                                   // bashF_deep() == 0
    die_on(with_host(K_PRIM, 192, "bashF", [&] { return bee2hip_bashF_batch(block, 1); }, [&] { hostp::bashF(block); }), "bashF");
}
extern "C" size_t bashF_deep(void) { return 0; }

// ==================================================================== belt ===
extern "C" const octet *beltH(void) { return host_beltH(); }

static inline u32 load32le(const octet *p)
{
    return (u32)p[0] | (u32)p[1] << 8 | (u32)p[2] << 16 | (u32)p[3] << 24;
}
static inline void store32le(octet *p, u32 v)
{
    p[0] = (octet)v; p[1] = (octet)(v >> 8); p[2] = (octet)(v >> 16); p[3] = (octet)(v >> 24);
}

extern "C" void beltKeyExpand2(u32 key_[8], const octet key[], size_t len)
{
    // pure data formatting, no cipher work (belt_block.c:88-106)
    for (size_t i = 0; i < len / 4; ++i) key_[i] = load32le(key + 4 * i);
    if (len == 16) {
        key_[4] = key_[0]; key_[5] = key_[1]; key_[6] = key_[2]; key_[7] = key_[3];
    } else if (len == 24) {
        key_[6] = key_[0] ^ key_[1] ^ key_[2];
        key_[7] = key_[3] ^ key_[4] ^ key_[5];
    }
}

extern "C" void beltBlockEncr2(u32 block[4], const u32 key[8])
{
    die_on(encr_host_blocks(block, 1, key), "beltBlockEncr2");
}
extern "C" void beltBlockEncr(octet block[16], const u32 key[8])
{
    u32 w[4];
    for (int i = 0; i < 4; ++i) w[i] = load32le(block + 4 * i);
    beltBlockEncr2(w, key);
    for (int i = 0; i < 4; ++i) store32le(block + 4 * i, w[i]);
}
extern "C" void beltBlockEncr3(u32 *a, u32 *b, u32 *c, u32 *d, const u32 key[8])
{
    u32 w[4] = {*a, *b, *c, *d};
    beltBlockEncr2(w, key);
    *a = w[0]; *b = w[1]; *c = w[2]; *d = w[3];
}

// ---- CTR: belt_ctr.c:46-135, state layout belt_lcl.h:135-141 ----
struct belt_ctr_st {
    u32 key[8];
    u32 ctr[4];
    octet block[16];
    size_t reserved;
};

extern "C" size_t beltCTR_keep(void) { return sizeof(belt_ctr_st); }

extern "C" void beltCTRStart(void *state, const octet key[], size_t len, const octet iv[16])
{
    belt_ctr_st *st = (belt_ctr_st *)state;
    beltKeyExpand2(st->key, key, len);
    for (int i = 0; i < 4; ++i) st->ctr[i] = load32le(iv + 4 * i);
    beltBlockEncr2(st->ctr, st->key);              // ctr0 = E_K(iv) on the GPU
    st->reserved = 0;
}

static inline void ctr_add(u32 c[4], uint64_t add)
{
    uint64_t lo = (uint64_t)c[0] | (uint64_t)c[1] << 32, hi = (uint64_t)c[2] | (uint64_t)c[3] << 32;
    const uint64_t nlo = lo + add;
    hi += nlo < lo;
    c[0] = (u32)nlo; c[1] = (u32)(nlo >> 32); c[2] = (u32)hi; c[3] = (u32)(hi >> 32);
}

// allow_host: the bee2 drop-ins (beltCTRStepE, beltCTR, beltDWPStepE ...) may finish a small call on the host; the batch
// entry point bee2hip_beltCTR_bulk never does
static err_t ctr_bulk(void *buf_, size_t count, void *ctr_state, bool allow_host)
{
    belt_ctr_st *st = (belt_ctr_st *)ctr_state;
    octet *buf = (octet *)buf_;
    if (!st || (count && !buf)) return ERR_BAD_INPUT;
    // gamma left over from the previous call (belt_ctr.c:70-83)
    if (st->reserved) {
        const size_t take = st->reserved < count ? st->reserved : count;
        const octet *g = st->block + 16 - st->reserved;
        for (size_t i = 0; i < take; ++i) buf[i] ^= g[i];
        st->reserved -= take; buf += take; count -= take;
        if (!count) return ERR_OK;
    }
    // whole blocks plus, if the tail is partial, one more gamma block: all on the GPU.
    // The tail is staged zero-padded to a full block; the kernel also hands back the
    // gamma of the final block, which the streaming state keeps (belt_ctr.c:89-96,101-108).
    const auto gpu = [&]() -> err_t {
        if (count >= DUPLEX_MIN) {
            // all but the last (at most one) chunk through the duplex pipeline, whole blocks; what is left -- with the partial
            // block and the gamma the state keeps -- takes the plain path below, from the advanced counter
            const size_t CH = (size_t)1 << g_duplex_log2_blocks;             // blocks per chunk (2^20 = 16 MiB)
            const size_t pipe_blocks = (count - 1) / (16 * CH) * CH;
            err_t pc = ensure_device();
            if (pc != ERR_OK) return pc;
            Scratch &ps = t_scr[2];
            pc = ps.need(pipe_blocks * 16);
            if (pc != ERR_OK) return pc;
            const u32 *key = st->key, *ctr = st->ctr;
            size_t done = 0;
            pc = duplex_inplace(buf, (octet *)ps.p, 16, pipe_blocks, CH, [key, ctr](octet *d, size_t first, size_t cnt, hipStream_t s2) {
                return launch_belt_ctr_blocks(d, cnt, key, ctr, first, nullptr, s2);
            }, &done);
            // the blocks that came back ARE encrypted in the caller's buffer, also when a later chunk failed: whoever goes on
            // (the retry, the host fallback of with_host) starts behind them, from the advanced counter
            ctr_add(st->ctr, done);
            buf += done * 16;
            count -= done * 16;
            if (pc != ERR_OK) return pc;
        }
        const size_t full = count / 16, tail = count % 16;
        const size_t nblk = full + (tail ? 1 : 0);
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[2];
        code = s.need(nblk * 16 + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)s.p;
        if (tail) B2H_TRY(zero_staging(d + full * 16, 16));
        B2H_TRY(h2d(d, buf, count));
        // first_block = 0: the offset is relative to the state's *current* counter
        code = launch_belt_ctr_blocks(d, nblk, st->key, st->ctr, 0, d + nblk * 16, nullptr);
        if (code != ERR_OK) return code;
        octet last[16];
        B2H_TRY(d2h(last, d + nblk * 16, 16));
        B2H_TRY(d2h(buf, d, count));
        memcpy(st->block, last, 16);
        ctr_add(st->ctr, nblk);                        // what nblk beltBlockIncU32 calls leave
        st->reserved = tail ? 16 - tail : 0;
        return ERR_OK;
    };
    if (!allow_host) return gpu();
    return with_host(K_PARALLEL, count, "beltCTRStepE", gpu,
                     [&] { hostp::ctr_blocks(hostT(), buf, count, st->key, st->ctr, st->block, &st->reserved); });
}
extern "C" err_t bee2hip_beltCTR_bulk(void *buf, size_t count, void *ctr_state) { return ctr_bulk(buf, count, ctr_state, false); }

extern "C" void beltCTRStepE(void *buf, size_t count, void *state)
{
    die_on(ctr_bulk(buf, count, state, true), "beltCTRStepE");
}

extern "C" err_t beltCTR(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16])
{
    if ((len != 16 && len != 24 && len != 32) || (count && (!src || !dest)) || !key || !iv)
        return ERR_BAD_INPUT;
    belt_ctr_st *st = new (std::nothrow) belt_ctr_st;
    if (!st) return ERR_OUTOFMEMORY;
    beltCTRStart(st, key, len, iv);
    if (dest != src) memmove(dest, src, count);
    err_t code = ctr_bulk(dest, count, st, true);
    delete st;
    return code;
}

// ==================================================================== bign ===
// STB 34.101.45 annex B parameter sets: k_bign{128,192,256}_{p,a,b,q,yG,seed} come from
// bign_curves.inc (generated from the reference's bignParamsStd, bign_params.c:34-230)
// DER of the pre-hash OIDs the level-fixed facades use (bign128.c:151-153, bign192.c:151-153,
// bign256.c:151-153): belt-hash, bash384, bash512
static const octet k_oid_belt_hash[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x1F, 0x51};
static const octet k_oid_bash384[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x4D, 0x0C};
static const octet k_oid_bash512[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x4D, 0x0D};

struct StdCurve { size_t l; const char *name; const octet *p, *a, *b, *q, *yG, *seed; };
static const StdCurve k_curves[3] = {
    {128, "1.2.112.0.2.0.34.101.45.3.1", k_bign128_p, k_bign128_a, k_bign128_b, k_bign128_q, k_bign128_yG, k_bign128_seed},
    {192, "1.2.112.0.2.0.34.101.45.3.2", k_bign192_p, k_bign192_a, k_bign192_b, k_bign192_q, k_bign192_yG, k_bign192_seed},
    {256, "1.2.112.0.2.0.34.101.45.3.3", k_bign256_p, k_bign256_a, k_bign256_b, k_bign256_q, k_bign256_yG, k_bign256_seed},
};

extern "C" err_t bignParamsStd(bign_params *params, const char *name)
{
    if (!params || !name) return ERR_BAD_INPUT;
    memset(params, 0, sizeof *params);
    for (const StdCurve &c : k_curves) {
        if (strcmp(name, c.name) == 0) {
            const size_t no = c.l / 4;
            params->l = c.l;
            memcpy(params->p, c.p, no);
            memcpy(params->a, c.a, no);
            memcpy(params->seed, c.seed, 8);
            memcpy(params->b, c.b, no);
            memcpy(params->q, c.q, no);
            memcpy(params->yG, c.yG, no);
            return ERR_OK;
        }
    }
    return ERR_FILE_NOT_FOUND;
}

static bool all_zero(const octet *p, size_t n)
{
    octet acc = 0;
    for (size_t i = 0; i < n; ++i) acc |= p[i];
    return acc == 0;
}

// bignParamsCheck (bign_params.c:244-280).  *standard = one of the three parameter sets of STB 34.101.45 annex B, which
// have their own kernels; anything else that passes goes to the general-curve kernels where they exist (verification,
// public-key validation) and is ERR_NOT_IMPLEMENTED elsewhere (the constant-time signing path).
static err_t params_check2(const bign_params *params, bool *standard)
{
    *standard = false;
    if (!params) return ERR_BAD_INPUT;
    if (2 * params->l % 64) return ERR_NOT_IMPLEMENTED;
    const size_t no = 2 * params->l / 8;
    if (no == 0 || no > 64) return ERR_BAD_PARAMS;
    const bool ok = params->p[0] % 4 == 3 && params->q[0] % 2 == 1 && params->p[no - 1] >= 128 &&
                    params->q[no - 1] >= 128 && all_zero(params->p + no, 64 - no) &&
                    !all_zero(params->a, no) && !all_zero(params->b, no) &&
                    all_zero(params->a + no, 64 - no) && all_zero(params->b + no, 64 - no) &&
                    all_zero(params->q + no, 64 - no) && all_zero(params->yG + no, 64 - no);
    if (!ok) return ERR_BAD_PARAMS;
    if (params->l % 64) return ERR_NOT_IMPLEMENTED;
    if (params->l != 128 && params->l != 192 && params->l != 256) return ERR_BAD_PARAMS;
    for (const StdCurve &c : k_curves) {
        if (c.l != params->l) continue;
        *standard = !(memcmp(params->p, c.p, no) || memcmp(params->a, c.a, no) || memcmp(params->b, c.b, no) ||
                      memcmp(params->q, c.q, no) || memcmp(params->yG, c.yG, no));
        break;
    }
    return ERR_OK;
}
// for the entry points that serve the standard curves only
// the signing side (bignPubkeyCalc, bignKeypairGen, bignSign*): every set bignParamsCheck + bignEcCreate accept;
// *standard tells which kernels serve it (the table-driven ones of bign_sign_kernels.hip, or the general-curve
// constant-time ladder of bign_generic_kernels.hip)
static err_t params_check_sign(const bign_params *params, bool *standard)
{
    const err_t code = params_check2(params, standard);
    if (code != ERR_OK) return code;
    return *standard ? ERR_OK : bign_generic_check(params);
}
static err_t params_check(const bign_params *params)
{
    bool standard;
    return params_check_sign(params, &standard);
}
static err_t pubkey_calc_any(const bign_params *params, bool keygen, const void *d_privkeys, size_t n, void *d_pubkeys, void *d_codes)
{
    bool standard;
    const err_t code = params_check_sign(params, &standard);
    if (code != ERR_OK) return code;
    return standard ? launch_bign_pubkey_calc(params->l, keygen, d_privkeys, n, d_pubkeys, d_codes, nullptr)
                    : launch_bign_pubkey_calc_generic(params, keygen, d_privkeys, n, d_pubkeys, d_codes, nullptr);
}

// oidFromDER(0, der, count) != SIZE_MAX  (src/core/oid.c:94-101, src/core/der.c:114-258,921-975):
// tag 0x06, definite minimal length covering the whole buffer, sub-identifiers without a
// leading 0x80 octet and below 2^32.
static bool oid_der_valid(const octet *der, size_t count)
{
    if (!der || count < 2 || count == (size_t)-1) return false;
    if (der[0] != 0x06) return false;
    size_t len, hdr;
    if (der[1] < 128) { len = der[1]; hdr = 2; }
    else {
        const size_t r = der[1] - 128;
        if (der[1] == 128 || der[1] == 255 || r > sizeof(size_t) || count < 2 + r) return false;
        if (der[2] == 0 || (r == 1 && der[2] < 128)) return false;
        len = 0;
        for (size_t i = 0; i < r; ++i) len = (len << 8) | der[2 + i];
        hdr = 2 + r;
    }
    if (hdr + len != count) return false;
    const octet *v = der + hdr;
    u32 val = 0;
    for (size_t pos = 0; pos < len; ++pos) {
        if (val & 0xFE000000u) return false;
        if (val == 0 && v[pos] == 128) return false;
        val = (val << 7) | (v[pos] & 127u);
        if ((v[pos] & 128) == 0) val = 0;
    }
    return true;
}

extern "C" err_t bee2hip_bignVerifyL_batch_dev(size_t l, const octet oid_der[], size_t oid_len,
                                               const void *d_hashes, const void *d_sigs,
                                               const void *d_pubkeys, size_t n, void *d_codes,
                                               void *stream)
{
    if (misaligned(d_hashes, 16) || misaligned(d_sigs, 16) || misaligned(d_pubkeys, 16) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_sigs || !d_pubkeys || !d_codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_verify(l, oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, as_stream(stream));
}

extern "C" err_t bee2hip_bignVerify_batch_dev(const octet oid_der[], size_t oid_len,
                                              const void *d_hashes, const void *d_sigs,
                                              const void *d_pubkeys, size_t n, void *d_codes,
                                              void *stream)
{
    return bee2hip_bignVerifyL_batch_dev(128, oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, stream);
}

extern "C" err_t bee2hip_bign128Verify_batch_dev(const void *d_hashes, const void *d_sigs,
                                                 const void *d_pubkeys, size_t n, void *d_codes,
                                                 void *stream)
{
    return bee2hip_bignVerifyL_batch_dev(128, k_oid_belt_hash, sizeof k_oid_belt_hash, d_hashes, d_sigs,
                                         d_pubkeys, n, d_codes, stream);
}

extern "C" err_t bee2hip_bignVerify_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                                          const octet *hashes, const octet *sigs, const octet *pubkeys,
                                          size_t n, err_t *codes)
{
    // order of checks as bignVerify: params first (bign_sign.c:355-356), then inputs, then OID
    bool standard;
    err_t code = params_check2(params, &standard);
    if (code != ERR_OK) return code;
    if (!standard) {                              // bignEcCreate judges the parameters next (bign_sign.c:357-358)
        code = bign_generic_check(params);
        if (code != ERR_OK) return code;
    }
    if (n && (!hashes || !sigs || !pubkeys || !codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4;                 // octets per field element
    const size_t hb = no * n, sb = (no + no / 2) * n, pb = 2 * no * n;
    const size_t so = (hb + 15) & ~(size_t)15, po = (so + sb + 15) & ~(size_t)15, co = (po + pb + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + n * 4);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    if (standard && n >= VERIFY_PIPE_MIN && g_verify_pipe) {
        // big host batch: chunk c + 1 is uploaded (its own stream) while the kernels of chunk c run -- the 144 n octets
        // of input are a quarter of the time of the whole call otherwise (profiles/r03_verify_hostpipe.txt)
        code = t_duplex.get();
        if (code != ERR_OK) return code;
        const hipStream_t up = t_duplex.up, st = t_duplex.dn;
        const size_t CH = VERIFY_PIPE_CHUNK, nch = (n + CH - 1) / CH, sg = no + no / 2;
        std::vector<hipEvent_t> ev(nch, nullptr);
        hipError_t he = hipSuccess;
        for (size_t c = 0; c < nch && he == hipSuccess; ++c) he = hipEventCreateWithFlags(&ev[c], hipEventDisableTiming);
        for (size_t c = 0; c < nch && he == hipSuccess && code == ERR_OK; ++c) {
            const size_t first = c * CH, cnt = std::min(CH, n - first);
            he = hipMemcpyAsync(d + first * no, hashes + first * no, cnt * no, hipMemcpyHostToDevice, up);
            if (he == hipSuccess) he = hipMemcpyAsync(d + so + first * sg, sigs + first * sg, cnt * sg, hipMemcpyHostToDevice, up);
            if (he == hipSuccess) he = hipMemcpyAsync(d + po + first * 2 * no, pubkeys + first * 2 * no, cnt * 2 * no, hipMemcpyHostToDevice, up);
            if (he == hipSuccess) he = hipEventRecord(ev[c], up);
            if (he == hipSuccess) he = hipStreamWaitEvent(st, ev[c], 0);
            if (he == hipSuccess)
                code = launch_bign_verify(params->l, oid_der, oid_len, d + first * no, d + so + first * sg, d + po + first * 2 * no,
                                          cnt, d + co + 4 * first, st);
        }
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        (void)hipStreamSynchronize(up);
        for (size_t c = 0; c < nch; ++c) if (ev[c]) (void)hipEventDestroy(ev[c]);
        if (code != ERR_OK) return code;
        B2H_TRY(he);
        B2H_TRY(hipMemcpy(codes, d + co, 4 * n, hipMemcpyDeviceToHost));
        return ERR_OK;
    }
    B2H_TRY(h2d(d, hashes, hb));
    B2H_TRY(h2d(d + so, sigs, sb));
    B2H_TRY(h2d(d + po, pubkeys, pb));
    code = standard ? launch_bign_verify(params->l, oid_der, oid_len, d, d + so, d + po, n, d + co, nullptr)
                    : launch_bign_verify_generic(params, oid_der, oid_len, d, d + so, d + po, n, d + co, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(codes, d + co, 4 * n));
    return ERR_OK;
}

// the three standard curves for host_bign.hpp: c, q, y_G and the fixed tables of G, built at first use
template <int N>
static const hostb::Curve<N> &host_curve(int which, uint64_t c)
{
    static hostb::Curve<N> E;
    static std::once_flag once;
    std::call_once(once, [&] { E.init(c, k_curves[which].q, k_curves[which].yG); });
    return E;
}
static err_t verify_one_host(size_t l, const octet oid_der[], size_t oid_len, const octet hash[], const octet sig[],
                             const octet pubkey[])
{
    const hostp::BeltTables &T = hostT();
    if (l == 128) return hostb::verify<4>(host_curve<4>(0, BIGN128_CRANDALL_C), T, host_beltH(), oid_der, oid_len, hash, sig, pubkey);
    if (l == 192) return hostb::verify<6>(host_curve<6>(1, BIGN192_CRANDALL_C), T, host_beltH(), oid_der, oid_len, hash, sig, pubkey);
    return hostb::verify<8>(host_curve<8>(2, BIGN256_CRANDALL_C), T, host_beltH(), oid_der, oid_len, hash, sig, pubkey);
}

// ONE key pair / public key / signature on a standard curve, on the calling core in constant-time arithmetic
// (host_bign_ct.hpp; its header says what that covers).  BEE2HIP_FORCE=gpu keeps every secret in the GPU kernels.
template <int N>
static const hostct::SignCurve<N> &host_sign_curve(int which, uint64_t c)
{
    static hostct::SignCurve<N> S;
    static std::once_flag once;
    std::call_once(once, [&] { S.init(host_curve<N>(which, c)); });
    return S;
}
static bool sign_on_host(const bign_params *params)
{
    bool standard;
    return params_check2(params, &standard) == ERR_OK && standard && host_wanted(K_SIGN1, 1);
}
// -> an error code of bee2, or ERR_OUTOFMEMORY when the window table could not be built
static err_t pubkey_calc_one_host(size_t l, bool keygen, const octet *privkey, octet *pubkey)
{
    if (l == 128) { const auto &S = host_sign_curve<4>(0, BIGN128_CRANDALL_C); return S.ready ? hostct::pubkey_calc<4>(S, keygen, privkey, pubkey) : ERR_OUTOFMEMORY; }
    if (l == 192) { const auto &S = host_sign_curve<6>(1, BIGN192_CRANDALL_C); return S.ready ? hostct::pubkey_calc<6>(S, keygen, privkey, pubkey) : ERR_OUTOFMEMORY; }
    const auto &S = host_sign_curve<8>(2, BIGN256_CRANDALL_C);
    return S.ready ? hostct::pubkey_calc<8>(S, keygen, privkey, pubkey) : ERR_OUTOFMEMORY;
}
static err_t sign_one_host(size_t l, const octet oid_der[], size_t oid_len, const octet *hash, const octet *privkey, const octet *k,
                           const void *t, size_t t_len, octet *sig)
{
    const hostp::BeltTables &T = hostT();
    const octet *H = host_beltH();
    if (l == 128) { const auto &S = host_sign_curve<4>(0, BIGN128_CRANDALL_C); return S.ready ? hostct::sign<4>(S, T, H, oid_der, oid_len, hash, privkey, k, t, t_len, sig) : ERR_OUTOFMEMORY; }
    if (l == 192) { const auto &S = host_sign_curve<6>(1, BIGN192_CRANDALL_C); return S.ready ? hostct::sign<6>(S, T, H, oid_der, oid_len, hash, privkey, k, t, t_len, sig) : ERR_OUTOFMEMORY; }
    const auto &S = host_sign_curve<8>(2, BIGN256_CRANDALL_C);
    return S.ready ? hostct::sign<8>(S, T, H, oid_der, oid_len, hash, privkey, k, t, t_len, sig) : ERR_OUTOFMEMORY;
}

// ---- n signatures under ONE public key (bign_kernels.hip "one signer") ----
// device-resident hashes / signatures, the key on the host; standard curve
static err_t verify_onekey_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes, const void *d_sigs,
                               const octet pubkey[], size_t n, void *d_codes, hipStream_t st)
{
    const size_t no = l / 4;
    err_t code = launch_bign_verify_onekey(l, oid_der, oid_len, d_hashes, d_sigs, pubkey, n, d_codes, st);
    if (code != ERR_KEY_NOT_ON_CURVE) return code;
    // a key off the curve (or with a coordinate >= p): bee2 does not check (bign_sign.c:306-311), and the comb table of such a
    // point proves nothing about the reference's walk -- the general path with the key n times gives the reference's codes
    void *rep = nullptr;
    code = scratch_for_stream(st, 7, 2 * no * (n + 1), &rep);
    if (code != ERR_OK) return code;
    octet *d_key = (octet *)rep + 2 * no * n;
    B2H_TRY(hipMemcpyAsync(d_key, pubkey, 2 * no, hipMemcpyHostToDevice, st));
    B2H_TRY(hipStreamSynchronize(st));                 // (pubkey is the caller's, pageable)
    code = launch_replicate_key(d_key, 2 * no, n, rep, st);
    if (code != ERR_OK) return code;
    return launch_bign_verify(l, oid_der, oid_len, d_hashes, d_sigs, rep, n, d_codes, st);
}

extern "C" err_t bee2hip_bignVerifyL_onekey_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                                      const void *d_sigs, const octet pubkey[], size_t n, void *d_codes, void *stream)
{
    if (misaligned(d_hashes, 16) || misaligned(d_sigs, 16) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_sigs || !pubkey || !d_codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return verify_onekey_dev(l, oid_der, oid_len, d_hashes, d_sigs, pubkey, n, d_codes, as_stream(stream));
}

extern "C" err_t bee2hip_bignVerify_onekey_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                                                 const octet *hashes, const octet *sigs, const octet pubkey[], size_t n, err_t *codes)
{
    // order of checks as bignVerify / bee2hip_bignVerify_batch
    bool standard;
    err_t code = params_check2(params, &standard);
    if (code != ERR_OK) return code;
    if (!standard) {
        code = bign_generic_check(params);
        if (code != ERR_OK) return code;
    }
    if (n && (!hashes || !sigs || !pubkey || !codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    const size_t no = params->l / 4;
    if (!standard) {                                   // general-curve kernels: the key n times through the batch entry
        try {
            std::vector<octet> rep(2 * no * n);
            for (size_t i = 0; i < n; ++i) memcpy(rep.data() + 2 * no * i, pubkey, 2 * no);
            return bee2hip_bignVerify_batch(params, oid_der, oid_len, hashes, sigs, rep.data(), n, codes);
        } catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }
    }
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t hb = no * n, sb = (no + no / 2) * n;
    const size_t so = (hb + 15) & ~(size_t)15, co = (so + sb + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + n * 4);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    B2H_TRY(h2d(d, hashes, hb));
    B2H_TRY(h2d(d + so, sigs, sb));
    code = verify_onekey_dev(params->l, oid_der, oid_len, d, d + so, pubkey, n, d + co, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(codes, d + co, 4 * n));
    return ERR_OK;
}

// ---- n signatures of K signers: key_index[i] < nkeys says whose signature i is ----
extern "C" err_t bee2hip_bignVerifyL_keyed_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                                     const void *d_sigs, const octet pubkeys[], size_t nkeys,
                                                     const void *d_key_index, size_t n, void *d_codes, void *stream)
{
    if (misaligned(d_hashes, 16) || misaligned(d_sigs, 16) || misaligned(d_codes, 4) || misaligned(d_key_index, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_sigs || !pubkeys || !nkeys || !d_key_index || !d_codes)) return ERR_BAD_INPUT;
    if (nkeys > 4096) return ERR_BAD_INPUT;             // (more signers than that: the general entry)
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_verify_keyed(l, oid_der, oid_len, d_hashes, d_sigs, pubkeys, nkeys, d_key_index, n, d_codes, as_stream(stream));
}

extern "C" err_t bee2hip_bignVerify_keyed_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                                                const octet *hashes, const octet *sigs, const octet *pubkeys, size_t nkeys,
                                                const u32 *key_index, size_t n, err_t *codes)
{
    bool standard;
    err_t code = params_check2(params, &standard);
    if (code != ERR_OK) return code;
    if (!standard) {
        code = bign_generic_check(params);
        if (code != ERR_OK) return code;
    }
    if (n && (!hashes || !sigs || !pubkeys || !nkeys || !key_index || !codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    const size_t no = params->l / 4;
    if (!standard || nkeys > 4096) {                   // general-curve kernels / a crowd of signers: every signature with its key, the general entry
        try {
            std::vector<octet> rep(2 * no * n);
            std::vector<size_t> bad;
            for (size_t i = 0; i < n; ++i) {
                if (key_index[i] >= nkeys) { bad.push_back(i); memcpy(rep.data() + 2 * no * i, pubkeys, 2 * no); }
                else memcpy(rep.data() + 2 * no * i, pubkeys + 2 * no * key_index[i], 2 * no);
            }
            code = bee2hip_bignVerify_batch(params, oid_der, oid_len, hashes, sigs, rep.data(), n, codes);
            if (code == ERR_OK) for (size_t i : bad) codes[i] = ERR_BAD_INPUT;
            return code;
        } catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }
    }
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t hb = no * n, sb = (no + no / 2) * n;
    const size_t so = (hb + 15) & ~(size_t)15, io = (so + sb + 15) & ~(size_t)15, co = (io + 4 * n + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + n * 4);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    B2H_TRY(h2d(d, hashes, hb));
    B2H_TRY(h2d(d + so, sigs, sb));
    B2H_TRY(h2d(d + io, key_index, 4 * n));
    code = launch_bign_verify_keyed(params->l, oid_der, oid_len, d, d + so, pubkeys, nkeys, d + io, n, d + co, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(codes, d + co, 4 * n));
    return ERR_OK;
}

extern "C" err_t bignVerify(const bign_params *params, const octet oid_der[], size_t oid_len,
                            const octet hash[], const octet sig[], const octet pubkey[])
{
    err_t one = ERR_BAD_SIG;
    bool standard;
    err_t pc = params_check2(params, &standard);
    if (!hash || !sig || !pubkey) {
        if (pc == ERR_OK && !standard) pc = bign_generic_check(params);
        return pc != ERR_OK ? pc : ERR_BAD_INPUT;
    }
    // ONE signature on a standard curve: the calling core (host_bign.hpp) unless BEE2HIP_FORCE=gpu; same order of checks
    // as the batch entry (parameters, inputs, OID), same requirement of a usable device
    if (pc == ERR_OK && standard && host_wanted(K_VERIFY1, 1) && oid_der_valid(oid_der, oid_len)) {
        const err_t code = device_seen();
        if (code != ERR_OK) return code;
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        return verify_one_host(params->l, oid_der, oid_len, hash, sig, pubkey);
    }
    const err_t code = bee2hip_bignVerify_batch(params, oid_der, oid_len, hash, sig, pubkey, 1, &one);
    if (code == ERR_BEE2HIP_DEVICE && pc == ERR_OK && standard && force_mode() != FORCE_GPU) {
        // a device fault under a single verification: finished on the host like the void drop-ins (with_host)
        fprintf(stderr, "libbee2hip: bignVerify: device path failed (%s); finished on the host\n", t_err);
        g_n_fallback.fetch_add(1, std::memory_order_relaxed);
        return verify_one_host(params->l, oid_der, oid_len, hash, sig, pubkey);
    }
    return code != ERR_OK ? code : one;
}

static err_t level_verify(int which, const octet *oid, const octet *hash, const octet *sig, const octet *pubkey)
{
    bign_params params;
    bignParamsStd(&params, k_curves[which].name);
    return bignVerify(&params, oid, 11, hash, sig, pubkey);
}
extern "C" err_t bign128Verify(const octet hash[32], const octet sig[48], const octet pubkey[64])
{
    return level_verify(0, k_oid_belt_hash, hash, sig, pubkey);
}
extern "C" err_t bign192Verify(const octet hash[48], const octet sig[72], const octet pubkey[96])
{
    return level_verify(1, k_oid_bash384, hash, sig, pubkey);
}
extern "C" err_t bign256Verify(const octet hash[64], const octet sig[96], const octet pubkey[128])
{
    return level_verify(2, k_oid_bash512, hash, sig, pubkey);
}

// ---- public-key validation (bign_misc.c:319-365) ----
extern "C" err_t bee2hip_bignPubkeyValL_batch_dev(size_t l, const void *d_pubkeys, size_t n, void *d_codes,
                                                  void *stream)
{
    if (misaligned(d_pubkeys, 16) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_pubkeys || !d_codes)) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_pubkey_val(l, d_pubkeys, n, d_codes, as_stream(stream));
}

extern "C" err_t bee2hip_bignPubkeyVal_batch(const bign_params *params, const octet *pubkeys, size_t n,
                                             err_t *codes)
{
    // bignPubkeyVal: params first (bign_misc.c:358-361), then the key
    bool standard;
    err_t code = params_check2(params, &standard);
    if (code != ERR_OK) return code;
    if (!standard) {
        code = bign_generic_check(params);
        if (code != ERR_OK) return code;
    }
    if (n && (!pubkeys || !codes)) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t pb = params->l / 2 * n, co = (pb + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + n * 4);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    B2H_TRY(h2d(d, pubkeys, pb));
    code = standard ? launch_bign_pubkey_val(params->l, d, n, d + co, nullptr)
                    : launch_bign_pubkey_val_generic(params, d, n, d + co, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(codes, d + co, 4 * n));
    return ERR_OK;
}

extern "C" err_t bignPubkeyVal(const bign_params *params, const octet pubkey[])
{
    err_t one = ERR_BAD_PUBKEY;
    if (!pubkey) {
        bool standard;
        err_t pc = params_check2(params, &standard);
        if (pc == ERR_OK && !standard) pc = bign_generic_check(params);
        return pc != ERR_OK ? pc : ERR_BAD_INPUT;
    }
    {   // ONE key on a standard curve: the calling core (host_bign.hpp) unless BEE2HIP_FORCE=gpu
        bool standard;
        if (params_check2(params, &standard) == ERR_OK && standard && host_wanted(K_VERIFY1, 1)) {
            const err_t code = device_seen();
            if (code != ERR_OK) return code;
            g_n_host.fetch_add(1, std::memory_order_relaxed);
            const size_t l = params->l;
            if (l == 128) return hostb::pubkey_val<4>(host_curve<4>(0, BIGN128_CRANDALL_C), k_curves[0].b, pubkey);
            if (l == 192) return hostb::pubkey_val<6>(host_curve<6>(1, BIGN192_CRANDALL_C), k_curves[1].b, pubkey);
            return hostb::pubkey_val<8>(host_curve<8>(2, BIGN256_CRANDALL_C), k_curves[2].b, pubkey);
        }
    }
    const err_t code = bee2hip_bignPubkeyVal_batch(params, pubkey, 1, &one);
    return code != ERR_OK ? code : one;
}
static err_t level_pubkey_val(int which, const octet *pubkey)
{
    bign_params params;
    bignParamsStd(&params, k_curves[which].name);
    return bignPubkeyVal(&params, pubkey);
}
extern "C" err_t bign128PubkeyVal(const octet pubkey[64]) { return level_pubkey_val(0, pubkey); }
extern "C" err_t bign192PubkeyVal(const octet pubkey[96]) { return level_pubkey_val(1, pubkey); }
extern "C" err_t bign256PubkeyVal(const octet pubkey[128]) { return level_pubkey_val(2, pubkey); }

// ---- 8f-4 tail: public key from private key, key generation, signing (bign_misc.c:182-229,373-417,
// bign_sign.c:32-245).  Secrets cross the staging buffer t_scr[3]; it is overwritten with zeros before return.
static void wipe_dev(void *p, size_t n) { if (p && n) (void)zero_staging(p, n); }
// staged secrets are zeroed on EVERY way out of a host entry point (early error returns, an allocation that throws)
struct WipeGuard {
    void *p;
    size_t n;
    ~WipeGuard() { wipe_dev(p, n); }
};

extern "C" err_t bee2hip_bignPubkeyCalcL_batch_dev(size_t l, const void *d_privkeys, size_t n, void *d_pubkeys,
                                                   void *d_codes, void *stream)
{
    if (misaligned(d_privkeys, 4) || misaligned(d_pubkeys, 4) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_privkeys || !d_pubkeys || !d_codes)) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_pubkey_calc(l, false, d_privkeys, n, d_pubkeys, d_codes, as_stream(stream));
}
extern "C" err_t bee2hip_bignSign2L_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                              const void *d_privkeys, const void *d_t, size_t t_len, int t_shared,
                                              size_t n, void *d_sigs, void *d_codes, void *stream)
{
    if (misaligned(d_hashes, 16) || misaligned(d_privkeys, 4) || misaligned(d_sigs, 4) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_privkeys || !d_sigs || !d_codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (!d_t) t_len = 0;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_sign(l, 0, oid_der, oid_len, d_hashes, d_privkeys, t_len ? d_t : nullptr, t_len, t_shared, n, d_sigs,
                            d_codes, as_stream(stream));
}
extern "C" err_t bee2hip_bignSignKL_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                              const void *d_privkeys, const void *d_ks, size_t n, void *d_sigs,
                                              void *d_codes, void *stream)
{
    if (misaligned(d_hashes, 16) || misaligned(d_privkeys, 4) || misaligned(d_ks, 4) || misaligned(d_sigs, 4) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_privkeys || !d_ks || !d_sigs || !d_codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_sign(l, 1, oid_der, oid_len, d_hashes, d_privkeys, d_ks, 0, 0, n, d_sigs, d_codes, as_stream(stream));
}

extern "C" err_t bee2hip_bignPubkeyCalc_batch(const bign_params *params, const octet *privkeys, size_t n,
                                              octet *pubkeys, err_t *codes)
{
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    if (n && (!privkeys || !pubkeys || !codes)) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4;
    const size_t db = no * n, po = (db + 15) & ~(size_t)15, co = (po + 2 * no * n + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + 4 * n);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    const WipeGuard wipe{d, db};
    B2H_TRY(h2d(d, privkeys, db));
    code = pubkey_calc_any(params, false, d, n, d + po, d + co);
    if (code == ERR_OK) {
        hipError_t e = d2h(codes, d + co, 4 * n);
        // bee2 leaves the output alone when it fails: copy the keys of the good items only
        octet *tmp = new (std::nothrow) octet[2 * no * n];
        if (!tmp) return ERR_OUTOFMEMORY;                       // (the guard above wipes the staged keys)
        if (e == hipSuccess) e = d2h(tmp, d + po, 2 * no * n);
        if (e != hipSuccess) code = hip_fail(e, "bignPubkeyCalc copy");
        else for (size_t i = 0; i < n; ++i) if (codes[i] == ERR_OK) memcpy(pubkeys + 2 * no * i, tmp + 2 * no * i, 2 * no);
        delete[] tmp;
    }
    return code;
}

// mode 0: t (shared by the batch, may be null) -- bignSign2; mode 1: aux = one-time keys k[n][no] -- bignSign after its rng
static err_t sign_batch_host(int mode, const bign_params *params, const octet oid_der[], size_t oid_len, const octet *hashes,
                             const octet *privkeys, const octet *aux, size_t t_len, size_t n, octet *sigs, err_t *codes)
{
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    if (n && (!hashes || !privkeys || !sigs || !codes || (mode == 1 && !aux))) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    if (mode == 0 && !aux) t_len = 0;
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4, sg = no + no / 2;
    // theta for additional input beyond what the nonce kernel assembles itself: the library's streaming belt-hash
    // (beltHashStart / StepH / StepG below run on the device), one signature at a time
    std::vector<octet> theta;
    int dev_mode = mode;
    size_t ab = mode == 1 ? no * n : t_len;
    if (mode == 0 && t_len > 64) {
        theta.resize(32 * n);
        std::vector<octet> st(beltHash_keep());
        const ForceScope on_device(FORCE_GPU);                 // the private key is hashed by the kernels, in every mode
        for (size_t i = 0; i < n; ++i) {
            beltHashStart(st.data());
            beltHashStepH(oid_der, oid_len, st.data());
            beltHashStepH(privkeys + no * i, no, st.data());
            beltHashStepH(aux, t_len, st.data());
            beltHashStepG(theta.data() + 32 * i, st.data());
        }
        wipe_host(st.data(), st.size());
        // the streaming belt-hash staged the private keys and -- behind the data, at offset nblocks * 32 -- its chaining state
        // through t_scr[2] (pinned or device memory): wipe everything a call of this size can have touched
        if (t_scr[2].p) {
            const size_t cap = t_scr[2].p == t_scr[2].pin ? PINNED_MAX : t_scr[2].cap;
            (void)zero_staging(t_scr[2].p, std::min<size_t>(cap, ((std::max<size_t>(t_len, std::max<size_t>(oid_len, no)) + 31) & ~(size_t)31) + 128));
        }
        aux = theta.data();
        ab = 32 * n;
        dev_mode = 2;
    }
    const size_t hb = no * n;
    const size_t o_d = (hb + 15) & ~(size_t)15, o_a = (o_d + hb + 15) & ~(size_t)15, o_s = (o_a + ab + 15) & ~(size_t)15,
                 o_c = (o_s + sg * n + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(o_c + 4 * n);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    const WipeGuard wipe{d + o_d, o_s - o_d};              // private keys and one-time keys / t / theta
    B2H_TRY(h2d(d, hashes, hb));
    B2H_TRY(h2d(d + o_d, privkeys, hb));
    if (ab) B2H_TRY(h2d(d + o_a, aux, ab));
    {
        bool standard;
        code = params_check_sign(params, &standard);
        if (code == ERR_OK)
            code = standard ? launch_bign_sign(params->l, dev_mode, oid_der, oid_len, d, d + o_d, ab ? d + o_a : nullptr, t_len, 1, n,
                                               d + o_s, d + o_c, nullptr)
                            : launch_bign_sign_generic(params, dev_mode, oid_der, oid_len, d, d + o_d, ab ? d + o_a : nullptr, t_len, 1,
                                                       n, d + o_s, d + o_c, nullptr);
    }
    if (code == ERR_OK) {
        hipError_t e = d2h(codes, d + o_c, 4 * n);
        std::vector<octet> tmp(sg * n);
        if (e == hipSuccess) e = d2h(tmp.data(), d + o_s, sg * n);
        if (e != hipSuccess) code = hip_fail(e, "bignSign copy");
        else for (size_t i = 0; i < n; ++i) if (codes[i] == ERR_OK) memcpy(sigs + sg * i, tmp.data() + sg * i, sg);
    }
    if (!theta.empty()) wipe_host(theta.data(), theta.size());
    return code;
}
extern "C" err_t bee2hip_bignSign2_batch(const bign_params *params, const octet oid_der[], size_t oid_len, const octet *hashes,
                                         const octet *privkeys, const void *t, size_t t_len, size_t n, octet *sigs, err_t *codes)
{
    try {
        return sign_batch_host(0, params, oid_der, oid_len, hashes, privkeys, (const octet *)t, t_len, n, sigs, codes);
    } catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }      // nothing may unwind through the C ABI
}
extern "C" err_t bee2hip_bignSignK_batch(const bign_params *params, const octet oid_der[], size_t oid_len, const octet *hashes,
                                         const octet *privkeys, const octet *ks, size_t n, octet *sigs, err_t *codes)
{
    try {
        return sign_batch_host(1, params, oid_der, oid_len, hashes, privkeys, ks, 0, n, sigs, codes);
    } catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }
}

// ---- drop-ins.  Order of checks as the reference: parameters (bignParamsCheck), pointers, OID, private key.
extern "C" err_t bignPubkeyCalc(octet pubkey[], const bign_params *params, const octet privkey[])
{
    err_t one = ERR_BAD_PRIVKEY;
    if (!pubkey || !privkey) {
        const err_t pc = params_check(params);
        return pc != ERR_OK ? pc : ERR_BAD_INPUT;
    }
    if (sign_on_host(params)) {
        const err_t dc = device_seen();            // the library still needs its GPU (no GPU-less operation)
        if (dc != ERR_OK) return dc;
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        return pubkey_calc_one_host(params->l, false, privkey, pubkey);
    }
    const err_t code = bee2hip_bignPubkeyCalc_batch(params, privkey, 1, pubkey, &one);
    return code != ERR_OK ? code : one;
}
// zzRandNZMod (zz_mod.c:463-485) on the host, exactly as bee2 calls the caller's generator: draws of no octets
// until 0 < a < mod, at most B_PER_IMPOSSIBLE + 1 = 65 of them.  The comparison is the only arithmetic involved.
static bool rand_nz_mod(octet *a, const octet *mod, size_t no, gen_i rng, void *rng_state)
{
    for (int tries = 0; tries <= 64; ++tries) {
        rng(a, no, rng_state);
        bool zero = true, less = false;
        for (size_t i = 0; i < no; ++i) zero = zero && a[i] == 0;
        for (size_t i = no; i-- > 0;) {
            if (a[i] != mod[i]) { less = a[i] < mod[i]; break; }
        }
        if (!zero && less) return true;
    }
    return false;
}
extern "C" err_t bignKeypairGen(octet privkey[], octet pubkey[], const bign_params *params, gen_i rng, void *rng_state)
{
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    if (!privkey || !pubkey) return ERR_BAD_INPUT;
    if (!rng) return ERR_BAD_RNG;
    const size_t no = params->l / 4;
    octet d[64];
    // bignKeypairGenEc draws d below the FIELD modulus p (bign_misc.c:209), not below q
    if (!rand_nz_mod(d, params->p, no, rng, rng_state)) return ERR_BAD_RNG;
    // any d below 2^(2l) is multiplied, as bignMulBase does (no range check against q here)
    code = ensure_device();
    if (code == ERR_OK && sign_on_host(params)) {
        octet q2[128];
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        code = pubkey_calc_one_host(params->l, true, d, q2);      // ERR_BAD_PARAMS when d G = O (bign_misc.c:214-218)
        if (code == ERR_OK) { memcpy(privkey, d, no); memcpy(pubkey, q2, 2 * no); }
        wipe_host(d, sizeof d);
        return code;
    }
    if (code == ERR_OK) {
        Scratch &s = t_scr[3];
        code = s.need(64 + 128 + 16);
        if (code == ERR_OK) {
            octet *dd = (octet *)s.p;
            hipError_t e = h2d(dd, d, no);
            if (e == hipSuccess) {
                code = pubkey_calc_any(params, true, dd, 1, dd + 64, dd + 192);
                octet q[128];
                err_t one = ERR_BAD_PARAMS;
                if (code == ERR_OK) e = d2h(q, dd + 64, 2 * no);
                if (code == ERR_OK && e == hipSuccess) e = d2h(&one, dd + 192, 4);
                if (code == ERR_OK && e == hipSuccess) {
                    code = one;                                  // ERR_BAD_PARAMS when d G = O (bign_misc.c:214-218)
                    if (one == ERR_OK) { memcpy(privkey, d, no); memcpy(pubkey, q, 2 * no); }
                }
            }
            if (e != hipSuccess) code = hip_fail(e, "bignKeypairGen copy");
            wipe_dev(dd, 64);
        }
    }
    memset(d, 0, sizeof d);
    return code;
}
extern "C" err_t bignSign(octet sig[], const bign_params *params, const octet oid_der[], size_t oid_len, const octet hash[],
                          const octet privkey[], gen_i rng, void *rng_state)
{
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4;
    if (!hash || !privkey || !sig || (hash < sig + no + no / 2 && sig < hash + no)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (!rng) return ERR_BAD_RNG;
    // d first (bign_sign.c:62-68): a bad key must not consume the generator
    {
        bool zero = true, less = false;
        for (size_t i = 0; i < no; ++i) zero = zero && privkey[i] == 0;
        for (size_t i = no; i-- > 0;) if (privkey[i] != params->q[i]) { less = privkey[i] < params->q[i]; break; }
        if (zero || !less) return ERR_BAD_PRIVKEY;
    }
    octet k[64];
    if (!rand_nz_mod(k, params->q, no, rng, rng_state)) return ERR_BAD_RNG;
    err_t one = ERR_BAD_PRIVKEY;
    if (sign_on_host(params)) {
        code = device_seen();
        if (code == ERR_OK) {
            g_n_host.fetch_add(1, std::memory_order_relaxed);
            octet out[96];
            code = sign_one_host(params->l, oid_der, oid_len, hash, privkey, k, nullptr, 0, out);
            if (code == ERR_OK) memcpy(sig, out, no + no / 2);
        }
        wipe_host(k, sizeof k);
        return code;
    }
    code = bee2hip_bignSignK_batch(params, oid_der, oid_len, hash, privkey, k, 1, sig, &one);
    wipe_host(k, sizeof k);
    return code != ERR_OK ? code : one;
}
extern "C" err_t bignSign2(octet sig[], const bign_params *params, const octet oid_der[], size_t oid_len, const octet hash[],
                           const octet privkey[], const void *t, size_t t_len)
{
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4;
    if (!hash || !privkey || !sig || (hash < sig + no + no / 2 && sig < hash + no)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    err_t one = ERR_BAD_PRIVKEY;
    if (sign_on_host(params)) {
        code = device_seen();
        if (code != ERR_OK) return code;
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        octet out[96];                                   // sig may alias nothing else, but is written only on success
        code = sign_one_host(params->l, oid_der, oid_len, hash, privkey, nullptr, t, t ? t_len : 0, out);
        if (code == ERR_OK) memcpy(sig, out, no + no / 2);
        return code;
    }
    code = bee2hip_bignSign2_batch(params, oid_der, oid_len, hash, privkey, t, t_len, 1, sig, &one);
    return code != ERR_OK ? code : one;
}
#define B2H_LEVEL_FACADE(L, IDX, OID, NO)                                                                          \
    extern "C" err_t bign##L##PubkeyCalc(octet pubkey[2 * NO], const octet privkey[NO])                             \
    { bign_params p; bignParamsStd(&p, k_curves[IDX].name); return bignPubkeyCalc(pubkey, &p, privkey); }            \
    extern "C" err_t bign##L##KeypairGen(octet privkey[NO], octet pubkey[2 * NO], gen_i rng, void *rng_state)       \
    { bign_params p; bignParamsStd(&p, k_curves[IDX].name); return bignKeypairGen(privkey, pubkey, &p, rng, rng_state); } \
    extern "C" err_t bign##L##Sign(octet sig[NO + NO / 2], const octet hash[NO], const octet privkey[NO], gen_i rng, void *rng_state) \
    { bign_params p; bignParamsStd(&p, k_curves[IDX].name); return bignSign(sig, &p, OID, 11, hash, privkey, rng, rng_state); } \
    extern "C" err_t bign##L##Sign2(octet sig[NO + NO / 2], const octet hash[NO], const octet privkey[NO], const void *t, size_t t_len) \
    { bign_params p; bignParamsStd(&p, k_curves[IDX].name); return bignSign2(sig, &p, OID, 11, hash, privkey, t, t_len); }
B2H_LEVEL_FACADE(128, 0, k_oid_belt_hash, 32)
B2H_LEVEL_FACADE(192, 1, k_oid_bash384, 48)
B2H_LEVEL_FACADE(256, 2, k_oid_bash512, 64)
#undef B2H_LEVEL_FACADE

#ifdef BEE2HIP_EXPERIMENTS
extern "C" err_t bee2hip_debug_fe(int op, const void *d_a, const void *d_b, void *d_out, size_t n, void *stream)
{
    return launch_bign_debug_fe(128, op, d_a, d_b, d_out, n, as_stream(stream));
}
extern "C" err_t bee2hip_debug_feL(size_t l, int op, const void *d_a, const void *d_b, void *d_out, size_t n,
                                   void *stream)
{
    return launch_bign_debug_fe(l, op, d_a, d_b, d_out, n, as_stream(stream));
}
#endif

// ============================================================= bash hashing ===
// bash_hash_st / belt_mac_st (bee2 layouts) are defined in mixed_kernels.hip
extern "C" size_t bashHash_keep(void) { return sizeof(bash_hash_st); }   // + bashF_deep() == 0

extern "C" void bashHashStart(void *state, size_t l)
{
    bash_hash_st *st = (bash_hash_st *)state;
    memset(st->s, 0, sizeof st->s);
    st->s[192 - 8] = (octet)(l / 4);
    st->buf_len = 192 - l / 2;
    st->pos = 0;
}

// run the device sponge over `count` host bytes for one state
static err_t sponge_gpu(bash_hash_st *st, const octet *buf, size_t count)
{
    Scratch &s = t_scr[0];
    err_t code = s.need(sizeof(bash_hash_st) + count + 16, true);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    B2H_TRY(h2d(d, st, sizeof *st));
    B2H_TRY(h2d(d + sizeof *st, buf, count));
    const octet *dd = d + sizeof *st;
    // large chunk: byte-wise up to the next block boundary, whole rate blocks with 8 lanes (a 3x shorter chain,
    // DESIGN.md 4.7), the remainder byte-wise again
    const size_t head = st->pos ? st->buf_len - st->pos : 0;
    if (count >= 4096 + head) {
        const size_t blocks = (count - head) / st->buf_len, tail = count - head - blocks * st->buf_len;
        if (head) { code = launch_bash_sponge(d, dd, 0, head, 1, 0, nullptr); if (code != ERR_OK) return code; }
        code = launch_bash_sponge_cols(d, dd + head, blocks, nullptr);
        if (code != ERR_OK) return code;
        if (tail) { code = launch_bash_sponge(d, dd + head + blocks * st->buf_len, 0, tail, 1, 0, nullptr); if (code != ERR_OK) return code; }
    } else {
        code = launch_bash_sponge(d, dd, 0, count, 1, 0, nullptr);
        if (code != ERR_OK) return code;
    }
    B2H_TRY(d2h(st, d, sizeof *st));
    return ERR_OK;
}

static err_t sponge_host(bash_hash_st *st, const octet *buf, size_t count)
{
    return with_host(K_SERIAL, count, "bashHashStepH", [&] { return sponge_gpu(st, buf, count); },
                     [&] { hostp::sponge_absorb(st->s, st->buf_len, &st->pos, buf, count); });
}

extern "C" void bashHashStepH(const void *buf, size_t count, void *state)
{
    bash_hash_st *st = (bash_hash_st *)state;
    // not a full rate block yet: buffering only, no permutation (bash_hash.c:57-62)
    if (count < st->buf_len - st->pos) {
        memcpy(st->s + st->pos, buf, count);
        st->pos += count;
        return;
    }
    die_on(sponge_host(st, (const octet *)buf, count), "bashHashStepH");
}

static void hash_final(bash_hash_st *st)
{
    // s1 = s, pad with 0x40 0.. (bash_hash.c:86-100), one more bashF -- on the GPU
    memcpy(st->s1, st->s, 192);
    memset(st->s1 + st->pos, 0, st->buf_len - st->pos);
    st->s1[st->pos] = 0x40;
    die_on(with_host(K_PRIM, 192, "bashHashStepG", [&] { return bee2hip_bashF_batch(st->s1, 1); }, [&] { hostp::bashF(st->s1); }),
           "bashHashStepG");
}

extern "C" void bashHashStepG(octet hash[], size_t hash_len, void *state)
{
    bash_hash_st *st = (bash_hash_st *)state;
    hash_final(st);
    memmove(hash, st->s1, hash_len);
}

extern "C" bool_t bashHashStepV(const octet hash[], size_t hash_len, void *state)
{
    bash_hash_st *st = (bash_hash_st *)state;
    hash_final(st);
    return memcmp(hash, st->s1, hash_len) == 0;
}

extern "C" err_t bashHash(octet hash[], size_t l, const void *src, size_t count)
{
    if (l == 0 || l % 16 != 0 || l > 256) return ERR_BAD_PARAMS;
    if ((count && !src) || !hash) return ERR_BAD_INPUT;
    bash_hash_st *st = new (std::nothrow) bash_hash_st;
    if (!st) return ERR_OUTOFMEMORY;
    bashHashStart(st, l);
    bashHashStepH(src, count, st);
    bashHashStepG(hash, l / 4, st);
    delete st;
    return ERR_OK;
}

// ================================================================ belt MAC ===
extern "C" size_t beltMAC_keep(void) { return sizeof(belt_mac_st); }

static err_t mac_host(belt_mac_st *st, const octet *buf, size_t count, int mode)
{
    return with_host(K_SERIAL, count, "beltMAC", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[1];
        code = s.need(sizeof(belt_mac_st) + 8 + count + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)s.p;
        const size_t off = (sizeof(belt_mac_st) + 15) & ~(size_t)15;
        B2H_TRY(h2d(d, st, sizeof *st));
        if (count) B2H_TRY(h2d(d + off, buf, count));
        code = launch_belt_mac(d, d + off, 0, count, 1, mode, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(st, d, sizeof *st));
        return ERR_OK;
    }, [&] { hostp::mac_step(hostT(), st->key, st->s, st->r, st->mac, st->block, &st->filled, buf, count, mode); });
}

extern "C" void beltMACStart(void *state, const octet key[], size_t len)
{
    belt_mac_st *st = (belt_mac_st *)state;
    beltKeyExpand2(st->key, key, len);
    die_on(mac_host(st, nullptr, 0, 1), "beltMACStart");     // s = 0, r = E_K(0), filled = 0
}

extern "C" void beltMACStepA(const void *buf, size_t count, void *state)
{
    belt_mac_st *st = (belt_mac_st *)state;
    // still filling the look-ahead block: no cipher work (belt_mac.c:63-70)
    if (st->filled < 16 && count <= 16 - st->filled) {
        memcpy(st->block + st->filled, buf, count);
        st->filled += count;
        return;
    }
    die_on(mac_host(st, (const octet *)buf, count, 2), "beltMACStepA");
}

extern "C" void beltMACStepG2(octet mac[], size_t mac_len, void *state)
{
    belt_mac_st *st = (belt_mac_st *)state;
    die_on(mac_host(st, nullptr, 0, 4), "beltMACStepG");
    octet full[8];
    store32le(full, st->mac[0]);
    store32le(full + 4, st->mac[1]);
    memcpy(mac, full, mac_len);
}
extern "C" void beltMACStepG(octet mac[8], void *state) { beltMACStepG2(mac, 8, state); }

extern "C" bool_t beltMACStepV2(const octet mac[], size_t mac_len, void *state)
{
    octet full[8];
    beltMACStepG2(full, 8, state);
    return memcmp(mac, full, mac_len) == 0;
}
extern "C" bool_t beltMACStepV(const octet mac[8], void *state) { return beltMACStepV2(mac, 8, state); }

extern "C" err_t beltMAC(octet mac[8], const void *src, size_t count, const octet key[], size_t len)
{
    if ((len != 16 && len != 24 && len != 32) || (count && !src) || !key || !mac) return ERR_BAD_INPUT;
    belt_mac_st *st = new (std::nothrow) belt_mac_st;
    if (!st) return ERR_OUTOFMEMORY;
    beltMACStart(st, key, len);
    beltMACStepA(src, count, st);
    beltMACStepG(mac, st);
    delete st;
    return ERR_OK;
}

// ====================================================== mixed batch (H4) ===
extern "C" err_t bee2hip_bashHash_beltMAC_batch_dev(const void *d_msgs, size_t msg_len, size_t n, size_t l,
                                                    const octet key[], size_t key_len,
                                                    void *d_digests, void *d_tags, void *stream)
{
    if (misaligned(d_msgs, 16)) return ERR_BAD_INPUT;
    const bool do_hash = d_digests != nullptr, do_mac = d_tags != nullptr;
    if (do_hash && (l == 0 || l % 16 != 0 || l > 256)) return ERR_BAD_PARAMS;      // bash_hash.c:122-123
    if (do_mac && ((key_len != 16 && key_len != 24 && key_len != 32) || !key)) return ERR_BAD_INPUT;
    if (n && msg_len && !d_msgs) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    u32 kw[8] = {0};
    if (do_mac) beltKeyExpand2(kw, key, key_len);
    return launch_bashHash_beltMAC(d_msgs, msg_len, n, l, kw, do_hash, do_mac, d_digests, d_tags,
                                   as_stream(stream));
}

extern "C" err_t bee2hip_bashHash_beltMAC_batch(const octet *msgs, size_t msg_len, size_t n, size_t l,
                                                const octet key[], size_t key_len,
                                                octet *digests, octet *tags)
{
    if (digests && (l == 0 || l % 16 != 0 || l > 256)) return ERR_BAD_PARAMS;
    if (tags && ((key_len != 16 && key_len != 24 && key_len != 32) || !key)) return ERR_BAD_INPUT;
    if (n && msg_len && !msgs) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    const size_t dlen = digests ? l / 4 : 0;
    const size_t in_b = (n * msg_len + 15) & ~(size_t)15, dg_b = (n * dlen + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    err_t code = s.need(in_b + dg_b + n * 8 + 16);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    if (n * msg_len) B2H_TRY(h2d(d, msgs, n * msg_len));
    code = bee2hip_bashHash_beltMAC_batch_dev(d, msg_len, n, l, key, key_len, digests ? d + in_b : nullptr,
                                              tags ? d + in_b + dg_b : nullptr, nullptr);
    if (code != ERR_OK) return code;
    if (digests) B2H_TRY(d2h(digests, d + in_b, n * dlen));
    if (tags) B2H_TRY(d2h(tags, d + in_b + dg_b, n * 8));
    return ERR_OK;
}

// ============================================================ path policy (product ABI) ===
extern "C" int bee2hip_path_policy(int mode)
{
    const int was = bee2hip::force_mode();
    if (mode >= 0) bee2hip::g_force.store(mode == 1 ? bee2hip::FORCE_GPU : mode == 2 ? bee2hip::FORCE_CPU : bee2hip::FORCE_AUTO);
    return was;
}
// drop-in helper calls so far: which = 0 host path (by size or by BEE2HIP_FORCE=cpu), 1 GPU path, 2 finished on the host
// after the GPU path failed twice
extern "C" unsigned long long bee2hip_path_count(int which)
{
    return which == 0 ? bee2hip::g_n_host.load() : which == 1 ? bee2hip::g_n_gpu.load() : bee2hip::g_n_fallback.load();
}

#ifdef BEE2HIP_EXPERIMENTS      // everything from here to the end of the kernel-timing hook: libbee2hip_exp.so only
// ============================================================ internal tuning hook ===
// A/B switch for experiment builds (tools/bashf_ab.py); not part of the product ABI (BEE2HIP_INTERNAL).
namespace bee2hip { void set_bashF_variant(int v); void set_ctr_variant(int v); void set_verify_path(int v); void set_verify_split(int v); void set_sign_coop(int v); void set_sign_wg(int v); void set_fused_tab(int v); void set_long_hash_form(int v); void set_ragged_fork(int v); void set_verify_pairs(int v); void set_onekey_tab16(int v); void set_onekey_slots(int v); void set_onekey_quads(int v); void set_inv_lanes(int v); }
extern "C" err_t bee2hip_internal_tune(int key, int value)
{
    switch (key) {
    case 0: bee2hip::set_bashF_variant(value); return ERR_OK;
    case 1: bee2hip::set_ctr_variant(value); return ERR_OK;
    case 2: bee2hip::set_verify_path(value); return ERR_OK;
    case 3: bee2hip::g_pinned_limit = value < 0 ? 0 : (size_t)value > bee2hip::PINNED_MAX ? bee2hip::PINNED_MAX : (size_t)value; return ERR_OK;
    case 4: bee2hip::g_force.store(value == 1 ? bee2hip::FORCE_GPU : value == 2 ? bee2hip::FORCE_CPU : bee2hip::FORCE_AUTO); return ERR_OK;   // as BEE2HIP_FORCE
    case 5: bee2hip::g_inject_fail.store(value); return ERR_OK;       // tests: the next `value` GPU attempts of drop-in helpers fail
    case 8: bee2hip::set_verify_split(value); return ERR_OK;          // parts of a big verification batch (0 by size, 1 never, 2..4)
    case 6: bee2hip::g_duplex_log2_states = value; return ERR_OK;     // chunk of the duplex host pipeline, bashF states (log2)
    case 7: bee2hip::g_duplex_log2_blocks = value; return ERR_OK;     //                                   belt blocks (log2)
    case 12: bee2hip::set_sign_wg(value); return ERR_OK;              // largest workgroup of the signing side's hashing kernels
    case 11: bee2hip::g_verify_pipe = value; return ERR_OK;           // chunked upload of big host-pointer verification batches
    case 10: bee2hip::set_sign_coop(value); return ERR_OK;            // lanes per scalar of k G, signing side (0 = by batch size)
    case 9: bee2hip::g_duplex_ramp = value; return ERR_OK;            // ramped chunk sizes at the ends of the pipeline
    case 13: bee2hip::set_fused_tab(value); return ERR_OK;            // belt table of the fused bash + belt-mac kernel (A/B)
    case 16: bee2hip::set_long_hash_form(value); return ERR_OK;        // table / workgroup of the long belt-hash kernel (A/B)
    case 19: bee2hip::set_verify_pairs(value); return ERR_OK;          // verification main kernel: multiply-adds in pairs (-1 by size, 0 never, else always)
    case 23: bee2hip::set_inv_lanes(value); return ERR_OK;             // lanes of the shared-inversion kernel of verification (log2; 0 = by curve)
    case 22: bee2hip::set_onekey_quads(value); return ERR_OK;          // one-signer verification: four lanes per signature (-1 by size, 0 never, 1 always)
    case 21: bee2hip::set_onekey_slots(value); return ERR_OK;          // one-signer verification: keys the table cache keeps (tests: evictions under load)
    case 20: bee2hip::set_onekey_tab16(value); return ERR_OK;          // one-signer verification: log2 of the signatures after which a key gets its 16-bit table (-1 by curve, 63 never)
    case 17: bee2hip::set_ragged_fork(value); return ERR_OK;           // ragged hashing: long chains and short messages on two queues (1) or one (0)
    case 14: bee2hip::g_duplex_fail_chunk.store(value); return ERR_OK;   // tests: the duplex host pipeline fails at this chunk (1-based) ...
    case 15: bee2hip::g_duplex_fail_times.store(value); return ERR_OK;   // ... in the next `value` pipelines
    default: return ERR_BAD_INPUT;
    }
}

// drop-in helper calls so far: which = 0 host path (by size or by BEE2HIP_FORCE=cpu), 1 GPU path, 2 finished on the host
// after the GPU path failed twice
extern "C" unsigned long long bee2hip_internal_stat(int which)
{
    if (which == 3) return bee2hip::bign_onekey_table_builds();       // key tables built so far (one-signer / few-signers verification)
    return which == 0 ? bee2hip::g_n_host.load() : which == 1 ? bee2hip::g_n_gpu.load() : bee2hip::g_n_fallback.load();
}

// shader-clock probe: one wavefront spins for `us` microseconds of s_memrealtime (100 MHz) and reports how many
// shader cycles (s_memtime) went by -- launched on a second stream beside the kernels under test, it gives the
// clock the chip actually sustained under that load (DVFS: MI355X_MICROARCH.md "DVFS give-back")
__global__ void clock_probe_kernel(unsigned long long *out, unsigned long long ticks)
{
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    do {
        __builtin_amdgcn_s_sleep(32);
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    } while (r1 - r0 < ticks);
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
extern "C" err_t bee2hip_internal_clock_probe(void *d_out16, unsigned us, void *stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), (unsigned long long *)d_out16,
                       (unsigned long long)us * 100ull);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// ============================================================ kernel timing ===
extern "C" err_t bee2hip_time_kernel(int which, int reps, void *d_a, void *d_b, void *d_c, void *d_d,
                                     size_t n, size_t aux, void *stream, float *ms)
{
    if (reps <= 0 || !ms) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    hipStream_t st = as_stream(stream);
    const octet *H = host_beltH();
    u32 kw[8], c0[4];
    beltKeyExpand2(kw, H + 128, 32);
    for (int i = 0; i < 4; ++i) c0[i] = load32le(H + 192 + 4 * i);
    struct Events {                       // destroyed on every return path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    B2H_TRY(hipEventCreate(&ev.e0));
    B2H_TRY(hipEventCreate(&ev.e1));
    hipEvent_t e0 = ev.e0, e1 = ev.e1;
    B2H_TRY(hipEventRecord(e0, st));
    for (int r = 0; r < reps && code == ERR_OK; ++r) {
        switch (which) {
        case 0: code = launch_bashF_batch(d_a, n, st); break;
        case 1: code = launch_belt_ctr_blocks(d_a, n, kw, c0, 0, nullptr, st); break;
        case 2: code = launch_bign_verify(128, k_oid_belt_hash, sizeof k_oid_belt_hash, d_a, d_b, d_c, n, d_d, st); break;
        case 3: code = launch_bashHash_beltMAC(d_a, aux, n, 256, kw, d_b != nullptr, d_c != nullptr, d_b, d_c, st); break;
        default: code = ERR_BAD_INPUT;
        }
    }
    B2H_TRY(hipEventRecord(e1, st));
    B2H_TRY(hipEventSynchronize(e1));
    float total = 0;
    B2H_TRY(hipEventElapsedTime(&total, e0, e1));
    *ms = total / (float)reps;
    return code;
}
#endif   // BEE2HIP_EXPERIMENTS

// ============================================ 8f-1: block decrypt, ECB, CBC ===
static err_t decr_host_blocks(uint32_t *blocks, size_t n, const u32 key[8])
{
    return with_host(K_PRIM, n * 16, "belt block decryption", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[1];
        code = s.need(n * 16);
        if (code != ERR_OK) return code;
        B2H_TRY(h2d(s.p, blocks, n * 16));
        code = launch_belt_decr_blocks(s.p, n, key, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(blocks, s.p, n * 16));
        return ERR_OK;
    }, [&] { for (size_t i = 0; i < n; ++i) hostp::belt_decr(hostT(), blocks + 4 * i, key); });
}

extern "C" void beltBlockDecr2(u32 block[4], const u32 key[8])
{
    die_on(decr_host_blocks(block, 1, key), "beltBlockDecr2");
}
extern "C" void beltBlockDecr(octet block[16], const u32 key[8])
{
    u32 w[4];
    for (int i = 0; i < 4; ++i) w[i] = load32le(block + 4 * i);
    beltBlockDecr2(w, key);
    for (int i = 0; i < 4; ++i) store32le(block + 4 * i, w[i]);
}
extern "C" void beltBlockDecr3(u32 *a, u32 *b, u32 *c, u32 *d, const u32 key[8])
{
    u32 w[4] = {*a, *b, *c, *d};
    beltBlockDecr2(w, key);
    *a = w[0]; *b = w[1]; *c = w[2]; *d = w[3];
}

extern "C" err_t bee2hip_beltModes_blocks_dev(int mode, const void *d_src, void *d_dst, size_t nblocks,
                                              const u32 key[8], const u32 iv[4], void *stream)
{
    if (misaligned(d_src, 16) || misaligned(d_dst, 16)) return ERR_BAD_INPUT;
    if ((nblocks && (!d_src || !d_dst)) || !key || (mode == 2 && !iv)) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_modes(mode, d_src, d_dst, nblocks, key, iv, as_stream(stream));
}

extern "C" err_t bee2hip_beltCBCEncr_batch_dev(void *d_msgs, size_t nblk, size_t n, const u32 key[8],
                                               void *d_ivs, void *stream)
{
    if (misaligned(d_msgs, 16) || misaligned(d_ivs, 16)) return ERR_BAD_INPUT;
    if ((n && (!d_msgs || !d_ivs)) || !key) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_cbc_encr(d_msgs, nblk, n, key, d_ivs, as_stream(stream));
}

// whole blocks of a host buffer through one of the block-parallel modes
static err_t modes_host(int mode, octet *buf, size_t nblocks, const u32 key[8], const octet chain[16])
{
    if (nblocks == 0) return ERR_OK;
    const size_t bytes = nblocks * 16;
    u32 iv[4] = {0, 0, 0, 0};
    if (chain) for (int i = 0; i < 4; ++i) iv[i] = load32le(chain + 4 * i);
    return with_host(K_PARALLEL, bytes, "belt ECB / CBC blocks", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[2];
        code = s.need(2 * bytes);
        if (code != ERR_OK) return code;
        octet *d = (octet *)s.p;
        B2H_TRY(h2d(d, buf, bytes));
        code = launch_belt_modes(mode, d, d + bytes, nblocks, key, iv, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(buf, d + bytes, bytes));
        return ERR_OK;
    }, [&] { hostp::modes_blocks(hostT(), mode, buf, nblocks, key, iv); });
}

struct belt_ecb_st {          // belt_ecb.c:42-46
    u32 key[8];
    octet block[16];
};
extern "C" size_t beltECB_keep(void) { return sizeof(belt_ecb_st); }
extern "C" void beltECBStart(void *state, const octet key[], size_t len)
{
    beltKeyExpand2(((belt_ecb_st *)state)->key, key, len);
}

static void ecb_step(void *buf_, size_t count, belt_ecb_st *st, int decr)
{
    octet *buf = (octet *)buf_;
    const size_t full = count / 16, tail = count % 16;
    die_on(modes_host(decr ? 1 : 0, buf, full, st->key, nullptr), decr ? "beltECBStepD" : "beltECBStepE");
    if (tail) {
        // ciphertext stealing (belt_ecb.c:74-83,97-106): data shuffling on the host, the block on the GPU
        octet *p = buf + full * 16;
        memcpy(st->block, p, tail);
        memcpy(st->block + tail, p - 16 + tail, 16 - tail);
        if (decr) beltBlockDecr(st->block, st->key); else beltBlockEncr(st->block, st->key);
        memcpy(p, p - 16, tail);
        memcpy(p - 16, st->block, 16);
    }
}
extern "C" void beltECBStepE(void *buf, size_t count, void *state) { ecb_step(buf, count, (belt_ecb_st *)state, 0); }
extern "C" void beltECBStepD(void *buf, size_t count, void *state) { ecb_step(buf, count, (belt_ecb_st *)state, 1); }

static err_t ecb_oneshot(void *dest, const void *src, size_t count, const octet key[], size_t len, int decr)
{
    if (count < 16 || (len != 16 && len != 24 && len != 32) || !src || !dest || !key) return ERR_BAD_INPUT;
    belt_ecb_st st;
    beltECBStart(&st, key, len);
    memmove(dest, src, count);
    ecb_step(dest, count, &st, decr);
    return ERR_OK;
}
extern "C" err_t beltECBEncr(void *dest, const void *src, size_t count, const octet key[], size_t len)
{
    return ecb_oneshot(dest, src, count, key, len, 0);
}
extern "C" err_t beltECBDecr(void *dest, const void *src, size_t count, const octet key[], size_t len)
{
    return ecb_oneshot(dest, src, count, key, len, 1);
}

// ------------------------------------------------------------------ belt-dwp ---
struct belt_dwp_st {          // belt_dwp.c:27-37 (own layout: no beltPolyMul stack)
    belt_ctr_st ctr;
    u32 r[4];
    u32 t[4];
    uint64_t bits_open, bits_crit;
    octet block[16];
    size_t filled;
};
extern "C" size_t beltDWP_keep(void) { return sizeof(belt_dwp_st); }
extern "C" void beltDWPStart(void *state, const octet key[], size_t len, const octet iv[16])
{
    belt_dwp_st *st = (belt_dwp_st *)state;
    beltCTRStart(&st->ctr, key, len, iv);                       // ctr = E_K(iv)
    for (int i = 0; i < 4; ++i) st->r[i] = st->ctr.ctr[i];
    beltBlockEncr2(st->r, st->ctr.key);                         // r = E_K(ctr)   (belt_dwp.c:52-54)
    const octet *H = beltH();
    for (int i = 0; i < 4; ++i) st->t[i] = load32le(H + 4 * i); // t = H[0..16)   (:59)
    st->bits_open = st->bits_crit = 0;
    st->filled = 0;
}
extern "C" void beltDWPStepE(void *buf, size_t count, void *state) { beltCTRStepE(buf, count, &((belt_dwp_st *)state)->ctr); }
extern "C" void beltDWPStepD(void *buf, size_t count, void *state) { beltCTRStepE(buf, count, &((belt_dwp_st *)state)->ctr); }

extern "C" err_t bee2hip_beltDWP_absorb_dev(const void *d_data, size_t nbytes, const u32 r[4], const u32 t[4],
                                            void *d_t_out, void *stream)
{
    if (misaligned(d_data, 16) || misaligned(d_t_out, 4)) return ERR_BAD_INPUT;
    if ((nbytes && !d_data) || !r || !t || !d_t_out) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_polyhash(d_data, nbytes, r, t, d_t_out, as_stream(stream));
}
// t_out <- t after absorbing `nbytes` of host data (zero-padded to whole blocks), on the GPU
static err_t dwp_absorb_host(u32 t_out[4], const u32 t[4], const u32 r[4], const octet *data, size_t nbytes)
{
    return with_host(K_POLY, nbytes, "belt-dwp authentication", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        const size_t off = (nbytes + 15) & ~(size_t)15;
        code = sc.need(off + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        if (nbytes) B2H_TRY(h2d(d, data, nbytes));
        code = launch_belt_polyhash(d, nbytes, r, t, d + off, nullptr);
        if (code != ERR_OK) return code;
        octet out[16];
        B2H_TRY(d2h(out, d + off, 16));
        for (int i = 0; i < 4; ++i) t_out[i] = load32le(out + 4 * i);
        return ERR_OK;
    }, [&] {
        u32 acc[4] = {t[0], t[1], t[2], t[3]};
        hostp::polyhash(acc, r, data, nbytes);
        for (int i = 0; i < 4; ++i) t_out[i] = acc[i];
    });
}
// buffered absorb shared by StepI / StepA (belt_dwp.c:79-106,128-154): whole blocks go to the GPU in one call
static void dwp_feed(belt_dwp_st *st, const octet *p, size_t count, const char *who)
{
    if (st->filled) {
        size_t take = 16 - st->filled;
        if (take > count) take = count;
        memcpy(st->block + st->filled, p, take);
        st->filled += take; p += take; count -= take;
        if (st->filled < 16) return;
        die_on(dwp_absorb_host(st->t, st->t, st->r, st->block, 16), who);
        st->filled = 0;
    }
    const size_t full = count & ~(size_t)15;
    if (full) die_on(dwp_absorb_host(st->t, st->t, st->r, p, full), who);
    if (count - full) { memcpy(st->block, p + full, count - full); st->filled = count - full; }
}
extern "C" void beltDWPStepI(const void *buf, size_t count, void *state)
{
    belt_dwp_st *st = (belt_dwp_st *)state;
    st->bits_open += (uint64_t)count * 8;
    dwp_feed(st, (const octet *)buf, count, "beltDWPStepI");
}
extern "C" void beltDWPStepA(const void *buf, size_t count, void *state)
{
    belt_dwp_st *st = (belt_dwp_st *)state;
    if (count && st->bits_crit == 0 && st->filled) {            // the open data ends here: pad it (belt_dwp.c:115-122)
        die_on(dwp_absorb_host(st->t, st->t, st->r, st->block, st->filled), "beltDWPStepA");
        st->filled = 0;
    }
    st->bits_crit += (uint64_t)count * 8;
    dwp_feed(st, (const octet *)buf, count, "beltDWPStepA");
}
// the tag of everything absorbed so far; the state is not disturbed (belt_dwp.c:162-189)
static void dwp_tag(octet mac[8], const belt_dwp_st *st, const char *who)
{
    octet tail[32];
    size_t n = 0;
    if (st->filled) { memset(tail, 0, 16); memcpy(tail, st->block, st->filled); n = 16; }
    for (int i = 0; i < 8; ++i) {
        tail[n + i] = (octet)(st->bits_open >> (8 * i));
        tail[n + 8 + i] = (octet)(st->bits_crit >> (8 * i));
    }
    u32 t1[4];
    die_on(dwp_absorb_host(t1, st->t, st->r, tail, n + 16), who);
    beltBlockEncr2(t1, st->ctr.key);
    octet out[16];
    for (int i = 0; i < 4; ++i) store32le(out + 4 * i, t1[i]);
    memcpy(mac, out, 8);
}
extern "C" void beltDWPStepG(octet mac[8], void *state) { dwp_tag(mac, (const belt_dwp_st *)state, "beltDWPStepG"); }
extern "C" bool_t beltDWPStepV(const octet mac[8], void *state)
{
    octet m[8];
    dwp_tag(m, (const belt_dwp_st *)state, "beltDWPStepV");
    return memcmp(m, mac, 8) == 0;
}
extern "C" err_t beltDWPWrap(void *dest, octet mac[8], const void *src1, size_t count1, const void *src2,
                             size_t count2, const octet key[], size_t len, const octet iv[16])
{
    if ((len != 16 && len != 24 && len != 32) || (count1 && (!src1 || !dest)) || (count2 && !src2) || !key || !iv || !mac)
        return ERR_BAD_INPUT;
    belt_dwp_st st;
    beltDWPStart(&st, key, len, iv);
    beltDWPStepI(src2, count2, &st);                            // I before E: src2 may overlap dest (belt_dwp.c:218)
    if (count1) memmove(dest, src1, count1);
    beltDWPStepE(dest, count1, &st);
    beltDWPStepA(dest, count1, &st);
    beltDWPStepG(mac, &st);
    return ERR_OK;
}
extern "C" err_t beltDWPUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                               const octet mac[8], const octet key[], size_t len, const octet iv[16])
{
    if ((len != 16 && len != 24 && len != 32) || (count1 && (!src1 || !dest)) || (count2 && !src2) || !key || !iv || !mac)
        return ERR_BAD_INPUT;
    belt_dwp_st st;
    beltDWPStart(&st, key, len, iv);
    beltDWPStepI(src2, count2, &st);
    beltDWPStepA(src1, count1, &st);
    if (!beltDWPStepV(mac, &st)) return ERR_BAD_MAC;            // nothing is decrypted (belt_dwp.c:258-262)
    if (count1) memmove(dest, src1, count1);
    beltDWPStepD(dest, count1, &st);
    return ERR_OK;
}

// ----------------------------------------------------------------- belt-hash ---
struct belt_hash_st {         // belt_hash.c:28-36 (own layout: h || s contiguous for the kernel)
    u32 hs[12];               // h[8] || s[4]
    uint64_t bits_lo, bits_hi;
    octet block[32];
    size_t filled;
};
extern "C" size_t beltHash_keep(void) { return sizeof(belt_hash_st); }
extern "C" void beltHashStart(void *state)
{
    belt_hash_st *st = (belt_hash_st *)state;
    const octet *H = beltH();
    for (int i = 0; i < 8; ++i) st->hs[i] = load32le(H + 4 * i);     // h = H[0..32)  (belt_hash.c:52)
    for (int i = 8; i < 12; ++i) st->hs[i] = 0;
    st->bits_lo = st->bits_hi = 0;
    st->filled = 0;
}
// hs <- hs after nblocks 32-byte blocks of host data (+ the final length block when fin); on the GPU
static err_t hash_stream_host(u32 hs[12], const octet *data, size_t nblocks, int fin, uint64_t lo, uint64_t hi)
{
    return with_host(K_SERIAL, nblocks * 32, "beltHash", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        const size_t bytes = nblocks * 32;
        code = sc.need(bytes + 64, true);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        if (bytes) B2H_TRY(h2d(d, data, bytes));
        B2H_TRY(h2d(d + bytes, hs, 48));
        code = launch_belt_hash_stream(d + bytes, d, nblocks, fin, lo, hi, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(hs, d + bytes, 48));
        return ERR_OK;
    }, [&] { hostp::hash_stream(hostT(), hs, data, nblocks, fin, lo, hi); });
}
extern "C" void beltHashStepH(const void *buf, size_t count, void *state)
{
    belt_hash_st *st = (belt_hash_st *)state;
    const octet *p = (const octet *)buf;
    const uint64_t add = (uint64_t)count << 3;                        // 128-bit bit counter (belt_lcl.c:25-51)
    st->bits_lo += add;
    st->bits_hi += ((uint64_t)count >> 61) + (st->bits_lo < add);
    if (st->filled) {
        size_t take = 32 - st->filled;
        if (take > count) take = count;
        memcpy(st->block + st->filled, p, take);
        st->filled += take; p += take; count -= take;
        if (st->filled < 32) return;
        die_on(hash_stream_host(st->hs, st->block, 1, 0, 0, 0), "beltHashStepH");
        st->filled = 0;
    }
    const size_t full = count / 32;
    if (full) die_on(hash_stream_host(st->hs, p, full, 0, 0, 0), "beltHashStepH");
    if (count % 32) { memcpy(st->block, p + 32 * full, count % 32); st->filled = count % 32; }
}
static void hash_digest(octet out[32], const belt_hash_st *st, const char *who)
{
    u32 hs[12];
    memcpy(hs, st->hs, sizeof hs);                                    // the state is not disturbed (belt_hash.c:108-135)
    octet tail[32];
    size_t n = 0;
    if (st->filled) { memset(tail, 0, 32); memcpy(tail, st->block, st->filled); n = 1; }
    die_on(hash_stream_host(hs, tail, n, 1, st->bits_lo, st->bits_hi), who);
    for (int i = 0; i < 8; ++i) store32le(out + 4 * i, hs[i]);
}
extern "C" void beltHashStepG(octet hash[32], void *state) { hash_digest(hash, (const belt_hash_st *)state, "beltHashStepG"); }
extern "C" void beltHashStepG2(octet hash[], size_t hash_len, void *state)
{
    octet d[32];
    hash_digest(d, (const belt_hash_st *)state, "beltHashStepG2");
    memcpy(hash, d, hash_len < 32 ? hash_len : 32);
}
extern "C" bool_t beltHashStepV(const octet hash[32], void *state)
{
    octet d[32];
    hash_digest(d, (const belt_hash_st *)state, "beltHashStepV");
    return memcmp(d, hash, 32) == 0;
}
extern "C" bool_t beltHashStepV2(const octet hash[], size_t hash_len, void *state)
{
    octet d[32];
    hash_digest(d, (const belt_hash_st *)state, "beltHashStepV2");
    return memcmp(d, hash, hash_len < 32 ? hash_len : 32) == 0;
}
extern "C" err_t beltHash(octet hash[32], const void *src, size_t count)
{
    if (!hash || (count && !src)) return ERR_BAD_INPUT;              // belt_hash.c:177-179
    belt_hash_st st;
    beltHashStart(&st);
    beltHashStepH(src, count, &st);
    beltHashStepG(hash, &st);
    return ERR_OK;
}

// ------------------------------------------------------------------ belt-sde ---
struct belt_wbl_st {          // belt_lcl.h:143-149
    u32 key[8];
    octet block[16];
    octet sum[16];
    uint64_t round;           // `word` on this ABI
};
struct belt_sde_st {          // belt_sde.c:26-30
    belt_wbl_st wbl[1];
    octet s[16];
};
extern "C" size_t beltSDE_keep(void) { return sizeof(belt_sde_st); }
extern "C" void beltSDEStart(void *state, const octet key[], size_t len)
{
    belt_sde_st *st = (belt_sde_st *)state;
    beltKeyExpand2(st->wbl->key, key, len);
    st->wbl->round = 0;
}
extern "C" err_t bee2hip_beltSDE_sectors_dev(int decr, void *d_sectors, size_t sector_bytes, size_t nsectors,
                                             const u32 key[8], const void *d_ivs, void *stream)
{
    if (misaligned(d_sectors, 16) || misaligned(d_ivs, 16)) return ERR_BAD_INPUT;
    if ((decr != 0 && decr != 1) || !key || (nsectors && (!d_sectors || !d_ivs))) return ERR_BAD_INPUT;
    if (sector_bytes % 16 != 0 || sector_bytes < 32) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_sde(decr, d_sectors, sector_bytes / 16, nsectors, key, d_ivs, as_stream(stream));
}
static err_t sde_host(int decr, octet *buf, size_t count, const octet iv[16], belt_sde_st *st)
{
    const err_t rc = with_host(K_SERIAL, count, "beltSDE", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        code = sc.need(count + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        B2H_TRY(h2d(d, buf, count));
        B2H_TRY(h2d(d + count, iv, 16));
        code = launch_belt_sde(decr, d, count / 16, 1, st->wbl->key, d + count, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(buf, d, count));
        return ERR_OK;
    }, [&] { hostp::sde_sector(hostT(), decr, buf, count, iv, st->wbl->key); });
    if (rc == ERR_OK) st->wbl->round = decr ? 0 : 2 * (uint64_t)(count / 16);     // where the reference's loops stop (belt_wbl.c)
    return rc;
}
extern "C" void beltSDEStepE(void *buf, size_t count, const octet iv[16], void *state)
{
    die_on(sde_host(0, (octet *)buf, count, iv, (belt_sde_st *)state), "beltSDEStepE");
}
extern "C" void beltSDEStepD(void *buf, size_t count, const octet iv[16], void *state)
{
    die_on(sde_host(1, (octet *)buf, count, iv, (belt_sde_st *)state), "beltSDEStepD");
}
static err_t sde_oneshot(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16], int decr)
{
    // belt_sde.c:79-86
    if (count % 16 != 0 || count < 32 || (len != 16 && len != 24 && len != 32) || !src || !dest || !key || !iv)
        return ERR_BAD_INPUT;
    belt_sde_st st;
    beltSDEStart(&st, key, len);
    memmove(dest, src, count);
    return sde_host(decr, (octet *)dest, count, iv, &st);
}
extern "C" err_t beltSDEEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
{
    return sde_oneshot(dest, src, count, key, len, iv, 0);
}
extern "C" err_t beltSDEDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
{
    return sde_oneshot(dest, src, count, key, len, iv, 1);
}

// ------------------------------------------------------------------ belt-che ---
struct belt_che_st {          // belt_che.c:27-41 (own layout).  mac.ctr.key = K, mac.r = E_K(iv); mac.ctr's
    belt_dwp_st mac;          // counter fields are unused
    u32 s[4];
    octet gamma[16];
    size_t reserved;
};
extern "C" size_t beltCHE_keep(void) { return sizeof(belt_che_st); }
extern "C" void beltCHEStart(void *state, const octet key[], size_t len, const octet iv[16])
{
    belt_che_st *st = (belt_che_st *)state;
    memset(st, 0, sizeof *st);
    beltKeyExpand2(st->mac.ctr.key, key, len);
    for (int i = 0; i < 4; ++i) st->mac.r[i] = load32le(iv + 4 * i);
    beltBlockEncr2(st->mac.r, st->mac.ctr.key);                 // r = E_K(iv)  (belt_che.c:54-56)
    for (int i = 0; i < 4; ++i) st->s[i] = st->mac.r[i];        // s = r
    const octet *H = beltH();
    for (int i = 0; i < 4; ++i) st->mac.t[i] = load32le(H + 4 * i);
}
extern "C" err_t bee2hip_beltCHE_blocks_dev(const void *d_src, void *d_dst, size_t nblocks, const u32 key[8],
                                            const u32 s[4], uint64_t first_block, void *d_s_out, void *stream)
{
    if (misaligned(d_src, 16) || misaligned(d_dst, 16)) return ERR_BAD_INPUT;
    if ((nblocks && (!d_src || !d_dst)) || !key || !s) return ERR_BAD_INPUT;
    if (first_block + nblocks < first_block || first_block + nblocks == ~(uint64_t)0) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_che(d_src, d_dst, nblocks, key, s, first_block, d_s_out, as_stream(stream));
}
static err_t che_blocks_host(octet *buf, size_t nblocks, belt_che_st *st)
{
    if (nblocks == 0) return ERR_OK;
    const size_t bytes = nblocks * 16;
    return with_host(K_PARALLEL, bytes, "beltCHEStepE", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        code = sc.need(bytes + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        B2H_TRY(h2d(d, buf, bytes));
        code = launch_belt_che(d, d, nblocks, st->mac.ctr.key, st->s, 0, d + bytes, nullptr);
        if (code != ERR_OK) return code;
        octet snew[16];
        B2H_TRY(d2h(snew, d + bytes, 16));
        B2H_TRY(d2h(buf, d, bytes));
        for (int i = 0; i < 4; ++i) st->s[i] = load32le(snew + 4 * i);
        return ERR_OK;
    }, [&] { hostp::che_blocks(hostT(), buf, nblocks, st->mac.ctr.key, st->s); });
}
extern "C" void beltCHEStepE(void *buf_, size_t count, void *state)
{
    belt_che_st *st = (belt_che_st *)state;
    octet *buf = (octet *)buf_;
    if (st->reserved) {                                         // gamma left from the previous call (belt_che.c:69-83)
        const size_t take = st->reserved < count ? st->reserved : count;
        for (size_t i = 0; i < take; ++i) buf[i] ^= st->gamma[16 - st->reserved + i];
        st->reserved -= take; buf += take; count -= take;
    }
    die_on(che_blocks_host(buf, count / 16, st), "beltCHEStepE");
    buf += count / 16 * 16;
    count %= 16;
    if (count) {                                                // partial block: advance s (bookkeeping, like the CTR
        const u32 out = st->s[3] >> 31;                         // counter increment), gamma = E_K(s) on the GPU
        for (int i = 3; i > 0; --i) st->s[i] = (st->s[i] << 1) | (st->s[i - 1] >> 31);
        st->s[0] = (st->s[0] << 1) ^ (out ? 0x87u : 0u) ^ 1u;
        u32 g[4] = {st->s[0], st->s[1], st->s[2], st->s[3]};
        beltBlockEncr2(g, st->mac.ctr.key);
        for (int i = 0; i < 4; ++i) store32le(st->gamma + 4 * i, g[i]);
        for (size_t i = 0; i < count; ++i) buf[i] ^= st->gamma[i];
        st->reserved = 16 - count;
    }
}
extern "C" void beltCHEStepD(void *buf, size_t count, void *state) { beltCHEStepE(buf, count, state); }
extern "C" void beltCHEStepI(const void *buf, size_t count, void *state) { beltDWPStepI(buf, count, &((belt_che_st *)state)->mac); }
extern "C" void beltCHEStepA(const void *buf, size_t count, void *state) { beltDWPStepA(buf, count, &((belt_che_st *)state)->mac); }
extern "C" void beltCHEStepG(octet mac[8], void *state) { dwp_tag(mac, &((const belt_che_st *)state)->mac, "beltCHEStepG"); }
extern "C" bool_t beltCHEStepV(const octet mac[8], void *state)
{
    octet m[8];
    dwp_tag(m, &((const belt_che_st *)state)->mac, "beltCHEStepV");
    return memcmp(m, mac, 8) == 0;
}
extern "C" err_t beltCHEWrap(void *dest, octet mac[8], const void *src1, size_t count1, const void *src2,
                             size_t count2, const octet key[], size_t len, const octet iv[16])
{
    if ((len != 16 && len != 24 && len != 32) || (count1 && (!src1 || !dest)) || (count2 && !src2) || !key || !iv || !mac)
        return ERR_BAD_INPUT;
    belt_che_st st;
    beltCHEStart(&st, key, len, iv);
    beltCHEStepI(src2, count2, &st);
    if (count1) memmove(dest, src1, count1);
    beltCHEStepE(dest, count1, &st);
    beltCHEStepA(dest, count1, &st);
    beltCHEStepG(mac, &st);
    return ERR_OK;
}
extern "C" err_t beltCHEUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                               const octet mac[8], const octet key[], size_t len, const octet iv[16])
{
    if ((len != 16 && len != 24 && len != 32) || (count1 && (!src1 || !dest)) || (count2 && !src2) || !key || !iv || !mac)
        return ERR_BAD_INPUT;
    belt_che_st st;
    beltCHEStart(&st, key, len, iv);
    beltCHEStepI(src2, count2, &st);
    beltCHEStepA(src1, count1, &st);
    if (!beltCHEStepV(mac, &st)) return ERR_BAD_MAC;
    if (count1) memmove(dest, src1, count1);
    beltCHEStepD(dest, count1, &st);
    return ERR_OK;
}

// ------------------------------------------------------------------ belt-bde ---
struct belt_bde_st {          // belt_bde.c:26-32
    u32 key[8];
    u32 s[4];
    octet block[16];
    octet block1[16];
};
extern "C" size_t beltBDE_keep(void) { return sizeof(belt_bde_st); }
extern "C" void beltBDEStart(void *state, const octet key[], size_t len, const octet iv[16])
{
    belt_bde_st *st = (belt_bde_st *)state;
    beltKeyExpand2(st->key, key, len);
    for (int i = 0; i < 4; ++i) st->s[i] = load32le(iv + 4 * i);
    beltBlockEncr2(st->s, st->key);                 // s = E_K(iv), on the GPU
}

extern "C" err_t bee2hip_beltBDE_blocks_dev(int decr, const void *d_src, void *d_dst, size_t nblocks,
                                            const u32 key[8], const u32 s[4], uint64_t first_block,
                                            void *d_s_out, void *stream)
{
    if (misaligned(d_src, 16) || misaligned(d_dst, 16)) return ERR_BAD_INPUT;
    if ((nblocks && (!d_src || !d_dst)) || !key || !s || (decr != 0 && decr != 1)) return ERR_BAD_INPUT;
    if (first_block + nblocks < first_block || first_block + nblocks == ~(uint64_t)0) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_bde(decr, d_src, d_dst, nblocks, key, s, first_block, d_s_out, as_stream(stream));
}

// whole blocks of a host buffer; advances st->s exactly as the reference's loop does
static err_t bde_host(int decr, octet *buf, size_t nblocks, belt_bde_st *st)
{
    if (nblocks == 0) return ERR_OK;
    const size_t bytes = nblocks * 16;
    const err_t rc = with_host(K_PARALLEL, bytes, "beltBDE", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        code = sc.need(bytes + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        B2H_TRY(h2d(d, buf, bytes));
        code = launch_belt_bde(decr, d, d, nblocks, st->key, st->s, 0, d + bytes, nullptr);
        if (code != ERR_OK) return code;
        octet snew[16];
        B2H_TRY(d2h(snew, d + bytes, 16));
        B2H_TRY(d2h(buf, d, bytes));
        for (int i = 0; i < 4; ++i) st->s[i] = load32le(snew + 4 * i);
        return ERR_OK;
    }, [&] { hostp::bde_blocks(hostT(), decr, buf, nblocks, st->key, st->s); });
    if (rc != ERR_OK) return rc;
    // what the reference's last iteration leaves behind (belt_bde.c:56-63): s, block = <s>, block1 = Y ^ <s>
    octet snew[16];
    for (int i = 0; i < 4; ++i) store32le(snew + 4 * i, st->s[i]);
    memcpy(st->block, snew, 16);
    for (int i = 0; i < 16; ++i) st->block1[i] = buf[bytes - 16 + i] ^ snew[i];
    return ERR_OK;
}
extern "C" void beltBDEStepE(void *buf, size_t count, void *state)
{
    die_on(bde_host(0, (octet *)buf, count / 16, (belt_bde_st *)state), "beltBDEStepE");
}
extern "C" void beltBDEStepD(void *buf, size_t count, void *state)
{
    die_on(bde_host(1, (octet *)buf, count / 16, (belt_bde_st *)state), "beltBDEStepD");
}
static err_t bde_oneshot(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16], int decr)
{
    // belt_bde.c:93-100, 118-125
    if (count % 16 != 0 || count < 16 || (len != 16 && len != 24 && len != 32) || !src || !dest || !key || !iv)
        return ERR_BAD_INPUT;
    belt_bde_st st;
    beltBDEStart(&st, key, len, iv);
    memmove(dest, src, count);
    return bde_host(decr, (octet *)dest, count / 16, &st);
}
extern "C" err_t beltBDEEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
{
    return bde_oneshot(dest, src, count, key, len, iv, 0);
}
extern "C" err_t beltBDEDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
{
    return bde_oneshot(dest, src, count, key, len, iv, 1);
}

struct belt_cbc_st {          // belt_cbc.c:63-68
    u32 key[8];
    octet block[16];
    octet block1[16];
};
extern "C" size_t beltCBC_keep(void) { return sizeof(belt_cbc_st); }
extern "C" void beltCBCStart(void *state, const octet key[], size_t len, const octet iv[16])
{
    belt_cbc_st *st = (belt_cbc_st *)state;
    beltKeyExpand2(st->key, key, len);
    memcpy(st->block, iv, 16);
}

extern "C" void beltCBCStepE(void *buf_, size_t count, void *state)
{
    belt_cbc_st *st = (belt_cbc_st *)state;
    octet *buf = (octet *)buf_;
    const size_t full = count / 16, tail = count % 16;
    if (full) {
        // the serial chain runs on one lane of the per-message kernel (n = 1), or on the host
        die_on(with_host(K_SERIAL, full * 16, "beltCBCStepE", [&]() -> err_t {
            err_t code = ensure_device();
            if (code != ERR_OK) return code;
            Scratch &s = t_scr[2];
            code = s.need(full * 16 + 16);
            if (code != ERR_OK) return code;
            octet *d = (octet *)s.p;
            B2H_TRY(h2d(d, buf, full * 16));
            B2H_TRY(h2d(d + full * 16, st->block, 16));
            code = launch_belt_cbc_encr(d, full, 1, st->key, d + full * 16, nullptr);
            if (code != ERR_OK) return code;
            octet chain[16];
            B2H_TRY(d2h(chain, d + full * 16, 16));
            B2H_TRY(d2h(buf, d, full * 16));
            memcpy(st->block, chain, 16);
            return ERR_OK;
        }, [&] { hostp::cbc_encr_blocks(hostT(), buf, full, st->key, st->block); }), "beltCBCStepE");
    }
    if (tail) {                                   // stealing, belt_cbc.c:86-93
        octet *p = buf + full * 16;
        for (size_t i = 0; i < tail; ++i) st->block1[i] = p[i] ^ st->block[i];
        memcpy(st->block1 + tail, p - 16 + tail, 16 - tail);
        beltBlockEncr(st->block1, st->key);
        memcpy(p, p - 16, tail);
        memcpy(p - 16, st->block1, 16);
    }
}

extern "C" void beltCBCStepD(void *buf_, size_t count, void *state)
{
    belt_cbc_st *st = (belt_cbc_st *)state;
    octet *buf = (octet *)buf_;
    // whole blocks handled by the parallel kernel: all of them, or all but the last full one
    // when a partial tail follows (belt_cbc.c:101-116: "while (count >= 32 || count == 16)")
    const size_t tail = count % 16;
    const size_t par = tail ? count / 16 - 1 : count / 16;
    if (par) {
        octet last[16];
        memcpy(last, buf + (par - 1) * 16, 16);               // becomes the next chaining value
        die_on(modes_host(2, buf, par, st->key, st->block), "beltCBCStepD");
        memcpy(st->block, last, 16);
    }
    if (tail) {                                   // 16 < rest < 32, belt_cbc.c:118-130
        octet *p = buf + par * 16;
        const size_t r = tail;
        memcpy(st->block1, p, 16);
        beltBlockDecr(st->block1, st->key);
        for (size_t i = 0; i < r; ++i) { octet x = st->block1[i]; st->block1[i] = p[16 + i]; p[16 + i] = x; }
        for (size_t i = 0; i < r; ++i) p[16 + i] ^= st->block1[i];
        beltBlockDecr(st->block1, st->key);
        for (int i = 0; i < 16; ++i) p[i] = st->block1[i] ^ st->block[i];
    }
}

static err_t cbc_oneshot(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16], int decr)
{
    if (count < 16 || (len != 16 && len != 24 && len != 32) || !src || !dest || !key || !iv) return ERR_BAD_INPUT;
    belt_cbc_st st;
    beltCBCStart(&st, key, len, iv);
    memmove(dest, src, count);
    if (decr) beltCBCStepD(dest, count, &st); else beltCBCStepE(dest, count, &st);
    return ERR_OK;
}
extern "C" err_t beltCBCEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
{
    return cbc_oneshot(dest, src, count, key, len, iv, 0);
}
extern "C" err_t beltCBCDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
{
    return cbc_oneshot(dest, src, count, key, len, iv, 1);
}

// ================================================= 8f-3: ragged hash batches ===
extern "C" err_t bee2hip_hash_ragged_ordered_dev(size_t alg, const void *d_data, const void *d_offsets,
                                                 const void *d_order, size_t n, void *d_digests, void *stream)
{
    if (misaligned(d_offsets, 8) || misaligned(d_order, 4) || misaligned(d_digests, 4)) return ERR_BAD_INPUT;
    if (alg != 0 && alg != 128 && alg != 192 && alg != 256) return ERR_BAD_PARAMS;
    if (n && (!d_offsets || !d_digests)) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_hash_ragged(alg, d_data, d_offsets, d_order, n, d_digests, as_stream(stream));
}

extern "C" err_t bee2hip_hash_ragged_dev(size_t alg, const void *d_data, const void *d_offsets, size_t n,
                                         void *d_digests, void *stream)
{
    return bee2hip_hash_ragged_ordered_dev(alg, d_data, d_offsets, nullptr, n, d_digests, stream);
}

static err_t hash_ragged_host(size_t alg, const octet *data, const uint64_t *offsets, size_t n, octet *digests);
extern "C" err_t bee2hip_hash_ragged(size_t alg, const octet *data, const uint64_t *offsets, size_t n,
                                     octet *digests)
{
    try { return hash_ragged_host(alg, data, offsets, n, digests); }
    catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }             // nothing may unwind through the C ABI
    catch (...) { return hip_fail(hipErrorUnknown, "bee2hip_hash_ragged: exception"); }
}
static err_t hash_ragged_host(size_t alg, const octet *data, const uint64_t *offsets, size_t n, octet *digests)
{
    if (alg != 0 && alg != 128 && alg != 192 && alg != 256) return ERR_BAD_PARAMS;
    if (n == 0) return ERR_OK;
    if (!offsets || !digests) return ERR_BAD_INPUT;
    for (size_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return ERR_BAD_INPUT;
    const size_t total = (size_t)offsets[n] , dlen = alg ? alg / 4 : 32;
    if (total && !data) return ERR_BAD_INPUT;
    if (n > 0xffffffffull) return ERR_BAD_INPUT;
    // longest first: the 64 lanes of a wavefront then hold messages of similar length and the long
    // serial chains start at once (bench.py "hash_ragged": +20 % belt-hash, +57 % bash256)
    std::vector<uint32_t> ord(n);
    for (size_t i = 0; i < n; ++i) ord[i] = (uint32_t)i;
    std::stable_sort(ord.begin(), ord.end(), [offsets](uint32_t a, uint32_t b) {
        return offsets[a + 1] - offsets[a] > offsets[b + 1] - offsets[b];
    });
    // A message is ONE dependent chain: a GPU lane (pair) walks it at ~0.12-0.15 us per octet, a host core at ~0.008.  When a
    // few messages are far longer than the rest the batch would wait for their chains (a 256 KiB message: 26-30 ms; a 1 GiB
    // file: minutes) with the device otherwise idle, so this HOST-pointer entry -- the data is in host memory anyway --
    // hands the K longest messages to host threads (host_small.hpp, as the drop-in beltHash / bashHash of one message
    // does) while the GPU takes the rest.  K balances the two sides: it grows while the host threads would finish before
    // the GPU's longest remaining chain.  The device-pointer entries never do this.  BEE2HIP_FORCE=gpu: K = 0.
    size_t K = 0;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t T = std::min<size_t>(hw ? hw : 1, 16);
    if (force_mode() != FORCE_GPU && n >= 2) {
        const double c_host = (alg ? 5.0e-9 : 8.4e-9) / (double)T, c_gpu = alg ? 4.0e-8 : 1.2e-7;    // seconds per octet (bench.py ragged leg)
        double host_s = 0;
        while (K + 1 < n) {
            const double len_k = (double)(offsets[ord[K] + 1] - offsets[ord[K]]), len_next = (double)(offsets[ord[K + 1] + 1] - offsets[ord[K + 1]]);
            if (len_k < 65536.0) break;                                   // chains under ~8 ms are the GPU's
            if (host_s + len_k * c_host > len_k * c_gpu) break;           // the host side would become the longer one
            host_s += len_k * c_host;
            ++K;
            if (len_next * c_gpu <= host_s) break;                         // the GPU's longest remaining chain is already shorter
        }
    }
    std::vector<octet> hdig(K * dlen);
    std::vector<std::thread> workers;
    std::atomic<size_t> next{0};
    const auto host_job = [&] {
        const hostp::BeltTables &HT = hostT();
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= K) return;
            const size_t i = ord[k];
            const octet *m = data + offsets[i];
            const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
            octet *out = hdig.data() + k * dlen;
            if (alg == 0) {
                hostb::BeltHashPieces bh(HT, host_beltH());
                bh.absorb(m, len);
                bh.digest(out);
            } else {
                octet st[192];
                memset(st, 0, sizeof st);
                st[192 - 8] = (octet)(alg / 4);                             // bashHashStart (bash_hash.c:38-48)
                const size_t rate = 192 - alg / 2;
                size_t pos = 0;
                hostp::sponge_absorb(st, rate, &pos, m, len);
                memset(st + pos, 0, rate - pos);                            // bashHashStepG (bash_hash.c:84-102)
                st[pos] = 0x40;
                hostp::bashF(st);
                memcpy(out, st, dlen);
            }
        }
    };
    struct Joiner {                                                        // joined on every way out
        std::vector<std::thread> &w;
        ~Joiner() { for (std::thread &t : w) if (t.joinable()) t.join(); }
    } joiner{workers};
    if (K) {
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        for (size_t t = 0; t < std::min(T, K); ++t) workers.emplace_back(host_job);
    }
    const size_t ng = n - K;                                               // slots of the GPU launch: ord[K .. n)
    const size_t ob = (n + 1) * 8, oo = (total + 15) & ~(size_t)15, ro = (oo + ob + 15) & ~(size_t)15,
                 go = (ro + n * 4 + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    err_t code = s.need(go + n * dlen + 16);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    if (total) B2H_TRY(h2d(d, data, total));
    B2H_TRY(h2d(d + oo, offsets, ob));
    B2H_TRY(h2d(d + ro, ord.data() + K, ng * 4));
    code = bee2hip_hash_ragged_ordered_dev(alg, d, d + oo, d + ro, ng, d + go, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(digests, d + go, n * dlen));
    for (std::thread &t : workers) t.join();
    for (size_t k = 0; k < K; ++k) memcpy(digests + (size_t)ord[k] * dlen, hdig.data() + k * dlen, dlen);
    return ERR_OK;
}
