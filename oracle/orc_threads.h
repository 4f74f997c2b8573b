/*
 * orc_threads.h -- TEST INFRASTRUCTURE (see oracle.h).
 * "parallel for" over a PERSISTENT pool of pthreads (orc_threads.c), so that the CPU baseline can be timed on 1 thread and
 * on all the host cores this process may use, over disjoint slices of the same batch, without paying a
 * pthread_create / join per thread per pass (256 spawns around 1.4 ms of work each).
 */
#ifndef BEE2_AMD_ORC_THREADS_H
#define BEE2_AMD_ORC_THREADS_H
#include <pthread.h>
#include <stddef.h>

typedef void (*orc_range_fn)(void *ctx, size_t lo, size_t hi);

/* fn(ctx, lo, hi) over [0, n) cut into `nthreads` contiguous slices, one per pool thread (the caller runs slice 0 itself);
   returns when every slice is done.  nthreads <= 1: the caller runs [0, n) inline.  Not re-entrant: one job at a time
   (concurrent callers are serialised). */
void orc_parallel_for(size_t n, int nthreads, orc_range_fn fn, void *ctx);

/* bench only: from now on every slice of a job is run `reps` times back to back by its thread (default 1), so that a timed
   pass gives each thread >= tens of milliseconds of work whatever the batch size (in-place work simply iterates) */
void orc_set_slice_reps(int reps);

/* how many threads' worth of cycles this process gets: `nthreads` pool threads each run a dependent integer chain for about
   `seconds`; returns the total iterations per second (compare with the value for nthreads = 1) */
double orc_spin_rate(int nthreads, double seconds);
#endif
