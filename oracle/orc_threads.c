/*
 * orc_threads.c -- TEST INFRASTRUCTURE (see oracle.h): the persistent thread pool behind orc_parallel_for.
 * Workers are created once (on the first call that asks for them, never destroyed: the process exits under them) and sleep
 * on a condition variable between jobs; a job is one (fn, ctx, n, nthreads) tuple, worker t takes slice t + 1, the caller
 * slice 0.  Nothing in the product library uses this.
 */
#define _GNU_SOURCE
#include "orc_threads.h"
#include <pthread.h>
#include <stdint.h>
#include <time.h>

#define ORC_MAX_THREADS 1024

static pthread_mutex_t g_call_mu = PTHREAD_MUTEX_INITIALIZER;   /* one job at a time */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_go = PTHREAD_COND_INITIALIZER, g_done = PTHREAD_COND_INITIALIZER;
static int g_workers = 0;                /* threads alive */
static unsigned long g_gen = 0;          /* job generation */
static int g_pending = 0;                /* slices of the current job not finished yet */
static struct { orc_range_fn fn; void *ctx; size_t n; int nthreads; } g_job;

static int g_slice_reps = 1;             /* every slice is run this many times (orc_set_slice_reps) */

void orc_set_slice_reps(int reps) { g_slice_reps = reps < 1 ? 1 : reps; }

static void run_slice(int t)
{
    const size_t n = g_job.n, nt = (size_t)g_job.nthreads;
    for (int r = 0; r < g_slice_reps; ++r)
        g_job.fn(g_job.ctx, n * (size_t)t / nt, n * ((size_t)t + 1) / nt);
}

static void *worker_main(void *arg)
{
    const int me = (int)(intptr_t)arg;   /* slice index of this worker: 1 .. */
    unsigned long seen = 0;
    pthread_mutex_lock(&g_mu);
    for (;;) {
        while (g_gen == seen) pthread_cond_wait(&g_go, &g_mu);
        seen = g_gen;
        if (me < g_job.nthreads) {
            pthread_mutex_unlock(&g_mu);
            run_slice(me);
            pthread_mutex_lock(&g_mu);
            if (--g_pending == 0) pthread_cond_signal(&g_done);
        }
    }
    return 0;
}

void orc_parallel_for(size_t n, int nthreads, orc_range_fn fn, void *ctx)
{
    if (nthreads <= 1 || n < 2) {
        for (int r = 0; r < g_slice_reps; ++r) fn(ctx, 0, n);
        return;
    }
    if (nthreads > ORC_MAX_THREADS) nthreads = ORC_MAX_THREADS;
    pthread_mutex_lock(&g_call_mu);
    pthread_mutex_lock(&g_mu);
    while (g_workers < nthreads - 1) {   /* grow the pool; a thread that cannot be started shrinks the job instead */
        pthread_t tid;
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        const int rc = pthread_create(&tid, &at, worker_main, (void *)(intptr_t)(g_workers + 1));
        pthread_attr_destroy(&at);
        if (rc != 0) break;
        ++g_workers;
    }
    if (nthreads > g_workers + 1) nthreads = g_workers + 1;
    g_job.fn = fn; g_job.ctx = ctx; g_job.n = n; g_job.nthreads = nthreads;
    g_pending = nthreads - 1;
    ++g_gen;
    pthread_cond_broadcast(&g_go);
    pthread_mutex_unlock(&g_mu);
    run_slice(0);
    pthread_mutex_lock(&g_mu);
    while (g_pending != 0) pthread_cond_wait(&g_done, &g_mu);
    pthread_mutex_unlock(&g_mu);
    pthread_mutex_unlock(&g_call_mu);
}

/* ---- spin calibration ---- */
typedef struct { double seconds; uint64_t iters[ORC_MAX_THREADS]; } spin_job;
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void spin_range(void *ctx, size_t lo, size_t hi)
{
    spin_job *j = (spin_job *)ctx;
    for (size_t t = lo; t < hi; ++t) {
        uint64_t x = 0x9E3779B97F4A7C15ull + t, it = 0;
        const double end = now_s() + j->seconds;
        do {
            for (int k = 0; k < 4096; ++k) x = x * 6364136223846793005ull + (x >> 29);   /* one dependent chain */
            it += 4096;
        } while (now_s() < end);
        j->iters[t] = it + (x == 42);
    }
}
double orc_spin_rate(int nthreads, double seconds)
{
    static spin_job j;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > ORC_MAX_THREADS) nthreads = ORC_MAX_THREADS;
    j.seconds = seconds;
    for (int t = 0; t < nthreads; ++t) j.iters[t] = 0;
    const int reps_was = g_slice_reps;
    g_slice_reps = 1;
    const double t0 = now_s();
    orc_parallel_for((size_t)nthreads, nthreads, spin_range, &j);
    const double dt = now_s() - t0;
    g_slice_reps = reps_was;
    uint64_t total = 0;
    for (int t = 0; t < nthreads; ++t) total += j.iters[t];
    return (double)total / dt;
}
