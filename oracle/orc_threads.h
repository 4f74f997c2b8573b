/*
 * orc_threads.h -- TEST INFRASTRUCTURE (see oracle.h).
 * Minimal pthread "parallel for" so the CPU baseline can be timed on 1 thread
 * and on all host cores over disjoint slices of the same batch.
 */
#ifndef BEE2_AMD_ORC_THREADS_H
#define BEE2_AMD_ORC_THREADS_H
#include <pthread.h>
#include <stddef.h>

typedef void (*orc_range_fn)(void *ctx, size_t lo, size_t hi);

typedef struct {
    orc_range_fn fn;
    void *ctx;
    size_t lo, hi;
} orc_slice;

static void *orc_slice_main(void *p)
{
    orc_slice *s = (orc_slice *)p;
    s->fn(s->ctx, s->lo, s->hi);
    return 0;
}

static inline void orc_parallel_for(size_t n, int nthreads, orc_range_fn fn, void *ctx)
{
    if (nthreads <= 1 || n < 2) { fn(ctx, 0, n); return; }
    if (nthreads > 256) nthreads = 256;
    pthread_t tid[256];
    orc_slice sl[256];
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
        sl[t].fn = fn; sl[t].ctx = ctx;
        sl[t].lo = n * (size_t)t / (size_t)nthreads;
        sl[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
        if (pthread_create(&tid[t], 0, orc_slice_main, &sl[t]) != 0) {
            /* could not spawn: run the slice inline */
            fn(ctx, sl[t].lo, sl[t].hi);
            tid[t] = 0;
            continue;
        }
        ++started;
    }
    (void)started;
    for (int t = 0; t < nthreads; ++t)
        if (tid[t]) pthread_join(tid[t], 0);
}
#endif
