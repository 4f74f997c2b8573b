// TEST INFRASTRUCTURE: a CPU stand-in for the slice of the HIP runtime that the HOST side of libbee2hip.so uses (staging.hpp,
// multi.hip): streams are worker threads with a task queue, events are markers in those queues, device memory is the heap.
// It exists so that the library's own thread / stream / event logic -- the duplex pipeline, the scratch pool, the worker pool,
// the host fallback -- can run under ThreadSanitizer and AddressSanitizer on a box without a GPU
// (tests/test_host_sanitizers.py builds tests/hostshim/staging_mock_main.cpp against it).  Asynchrony is REAL (copies and
// "kernels" run on the stream's thread, later than the call), so a missing wait in the library is a data race TSan sees; "device"
// buffers are heap blocks, so an overrun is a heap-buffer-overflow ASan sees.  Never compiled into the product.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600,
             hipErrorStreamCaptureUnsupported = 900, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum : unsigned { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocPortable = 1, hipHostMallocMapped = 2 };

namespace mockhip {
struct Stream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    void run()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return stop || !q.empty(); });
            if (q.empty()) return;
            std::function<void()> f = std::move(q.front());
            q.pop_front();
            busy = true;
            lk.unlock();
            f();
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void push(std::function<void()> f)
    {
        std::lock_guard<std::mutex> lk(mu);
        q.push_back(std::move(f));
        cv.notify_all();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return q.empty() && !busy; });
    }
    ~Stream()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
        th.join();
    }
};
struct Event {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;      // generations: a wait or query refers to the record that was last when it was issued
    std::atomic<int> refs{1};                  // the handle + every queued task that names the event (hipEventDestroy defers, as HIP does)
    void ref() { refs.fetch_add(1); }
    void unref() { if (refs.fetch_sub(1) == 1) delete this; }
};
struct Global {
    std::mutex mu;
    std::vector<Stream *> streams;
    std::atomic<long> malloc_fail_in{0};       // the n-th device allocation from now fails (0 = off): tests of the out-of-memory paths
    std::atomic<long> live_device_blocks{0};
    int ndev = 2;
    Global() { if (const char *e = getenv("MOCKHIP_DEVICES")) ndev = atoi(e); }
};
inline Global &g() { static Global *x = new Global; return *x; }
inline Stream *null_stream()
{
    static Stream *s = [] { Stream *x = new Stream; std::lock_guard<std::mutex> lk(g().mu); g().streams.push_back(x); return x; }();
    return s;
}
inline thread_local int t_dev = 0;
}  // namespace mockhip

typedef mockhip::Stream *hipStream_t;
typedef mockhip::Event *hipEvent_t;

static inline mockhip::Stream *mock_q(hipStream_t s) { return s ? s : mockhip::null_stream(); }
// what a kernel launch is here: work queued on the stream, run later by the stream's thread
static inline void mockhipLaunch(hipStream_t s, std::function<void()> f) { mock_q(s)->push(std::move(f)); }

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "mock hip error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = mockhip::t_dev; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = mockhip::g().ndev; return *n > 0 ? hipSuccess : hipErrorNoDevice; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= mockhip::g().ndev) return hipErrorUnknown; mockhip::t_dev = d; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n)
{
    if (mockhip::g().malloc_fail_in.load() > 0 && mockhip::g().malloc_fail_in.fetch_sub(1) == 1) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    mockhip::g().live_device_blocks.fetch_add(1);
    return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize()
{
    std::vector<mockhip::Stream *> all;
    { std::lock_guard<std::mutex> lk(mockhip::g().mu); all = mockhip::g().streams; }
    for (mockhip::Stream *s : all) s->drain();
    return hipSuccess;
}
static inline hipError_t hipFree(void *p) { if (p) { hipDeviceSynchronize(); free(p); mockhip::g().live_device_blocks.fetch_sub(1); } return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    *s = new mockhip::Stream;
    std::lock_guard<std::mutex> lk(mockhip::g().mu);
    mockhip::g().streams.push_back(*s);
    return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t s) { mock_q(s)->drain(); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s) return hipErrorUnknown;
    s->drain();
    {
        std::lock_guard<std::mutex> lk(mockhip::g().mu);
        auto &v = mockhip::g().streams;
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == s) { v[i] = v.back(); v.pop_back(); break; }
    }
    delete s;
    return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st)
{
    mock_q(st)->push([d, s, n] { memcpy(d, s, n); });
    return hipSuccess;
}
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
    mockhip::null_stream()->drain();          // a blocking copy runs behind everything the NULL stream holds
    memcpy(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { mockhip::null_stream()->drain(); memset(d, v, n); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new mockhip::Event; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { e->unref(); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t st)
{
    uint64_t gen;
    { std::lock_guard<std::mutex> lk(e->mu); gen = ++e->recorded; }
    e->ref();
    mock_q(st)->push([e, gen] { { std::lock_guard<std::mutex> lk(e->mu); if (e->completed < gen) e->completed = gen; e->cv.notify_all(); } e->unref(); });
    return hipSuccess;
}
static inline hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned)
{
    uint64_t gen;
    { std::lock_guard<std::mutex> lk(e->mu); gen = e->recorded; }
    if (gen) {
        e->ref();
        mock_q(st)->push([e, gen] { { std::unique_lock<std::mutex> lk(e->mu); e->cv.wait(lk, [&] { return e->completed >= gen; }); } e->unref(); });
    }
    return hipSuccess;
}
static inline hipError_t hipEventQuery(hipEvent_t e)
{
    std::lock_guard<std::mutex> lk(e->mu);
    return e->completed >= e->recorded ? hipSuccess : hipErrorNotReady;
}
