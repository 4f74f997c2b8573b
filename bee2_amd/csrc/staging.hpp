// staging.hpp -- host side of libbee2hip.so shared by the capi_*.hip files (one translation unit: bee2hip_tu_belt.hip): error
// record, per-device constants, the scratch pool, pinned / device staging of the host-pointer entry points, the host path of small
// single calls, the duplex pipeline of large in-place batches.  Split out of capi.hip in round 5, no behaviour change.
#pragma once
#include <algorithm>
#include <atomic>
#include <exception>
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

// ------------------------------------------------- nothing unwinds through the C ABI ---
// Every extern "C" entry point of the library is a FUNCTION-TRY-BLOCK that ends in one of the macros below (tests/
// test_capi_exports.py reads the sources and refuses an entry without one): whatever a container, a std::thread or the
// runtime throws below it -- std::bad_alloc from a staging vector, std::system_error from a thread that cannot be started --
// becomes the entry's error code.  Helpers that own a thread or a secret add their own try / catch to join / wipe, then rethrow.
//   err_t entries   bad_alloc -> ERR_OUTOFMEMORY; anything else -> ERR_BEE2HIP_DEVICE with the message in bee2hip_last_error()
//   void entries    (bee2's Step functions cannot report) message on stderr + abort, as die_on() does for a device failure;
//                   with_host() below turns what the GPU path throws into a code first, so in the default mode these finish on the host
//   bool_t entries  (StepV) message on stderr, FALSE: a check that could not be made does not pass
static err_t caught() noexcept
{
    try { throw; }
    catch (const std::bad_alloc &) { (void)hipGetLastError(); snprintf(t_err, sizeof t_err, "out of host memory"); return ERR_OUTOFMEMORY; }
    catch (const std::exception &e) { snprintf(t_err, sizeof t_err, "exception: %s", e.what()); }
    catch (...) { snprintf(t_err, sizeof t_err, "unknown exception"); }
    (void)hipGetLastError();
    return ERR_BEE2HIP_DEVICE;
}
[[noreturn]] static void caught_void(const char *where) noexcept
{
    const err_t code = caught();
    fprintf(stderr, "libbee2hip: %s failed (err %u): %s\n", where, (unsigned)code, t_err);
    abort();
}
static bool_t caught_false(const char *where) noexcept
{
    const err_t code = caught();
    fprintf(stderr, "libbee2hip: %s failed (err %u): %s; reporting FALSE\n", where, (unsigned)code, t_err);
    return 0;
}
#define B2H_CATCH catch (...) { return ::bee2hip::caught(); }
#define B2H_CATCH_VOID(name) catch (...) { ::bee2hip::caught_void(name); }
#define B2H_CATCH_FALSE(name) catch (...) { return ::bee2hip::caught_false(name); }

#ifdef BEE2HIP_EXPERIMENTS
// tests (bee2hip_internal_tune 24): the n-th allocation through operator new from now on, by any thread of THIS library, fails
// (0 = off).  The replacement operators below are hidden symbols of libbee2hip_exp.so (-fvisibility=hidden, -Bsymbolic): only
// the library's own code -- its std::vector / std::thread / std::string instantiations included -- allocates through them.
static std::atomic<long> g_new_fail_in{0};
static std::atomic<unsigned long long> g_new_calls{0};
#endif

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
    // (tools/ab/pinned_ab.py: belt-hash of 16 KiB 2.49 ms pinned vs 2.23 ms copied; of 1 KiB 186 vs 200 us)
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
        else {
            try { code = gpu(); }
            catch (...) { code = caught(); }                  // a staging container that could not grow, a thread that could not start
        }
        if (code == ERR_OK) { g_n_gpu.fetch_add(1, std::memory_order_relaxed); return ERR_OK; }
        if (code == ERR_OUTOFMEMORY && force_mode() != FORCE_GPU) break;   // nothing to retry; the host path needs no memory
        if (code != ERR_BEE2HIP_DEVICE) return code;         // bad input: report, nothing to retry
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
    }
    if (force_mode() == FORCE_GPU) return code;
    fprintf(stderr, "libbee2hip: %s: device path %s (%s); finished on the host\n", what,
            code == ERR_OUTOFMEMORY ? "ran out of memory" : "failed twice", t_err);
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
    struct EvGuard {                                       // the events go on every exit, also when the thread below cannot be started
        std::vector<hipEvent_t> &ev; bool armed = true;
        ~EvGuard() { if (armed) for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } ev_guard{ev};
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
    // (a launcher may throw -- its scratch bookkeeping allocates: the helper thread is told, joined and both streams are drained
    //  as on every other path, *done_units is set, and the exception becomes the error code the entry point would have made of it)
    std::exception_ptr thrown;
    try {
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
    } catch (...) { thrown = std::current_exception(); failed.store(1); }
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
    ev_guard.armed = false;
    if (thrown) {
        //  the exception does NOT go on past the done_units contract: callers apply *done_units (advance the counter, skip
        // the chunks that are back) on an error CODE; an exception would skip that and a host fallback would transform them twice
        try { std::rethrow_exception(thrown); } catch (...) { return caught(); }
    }
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
