/* TEST INFRASTRUCTURE: ThreadSanitizer driver for the oracle's persistent thread pool (oracle/orc_threads.c) -- built by
 * tests/test_host_sanitizers.py together with the oracle's sources under -fsanitize=thread.  Two caller threads submit
 * jobs at once (orc_parallel_for serialises them), thread counts change between jobs, slices are repeated, the spin
 * calibration runs: every result must equal the single-thread one and TSan must stay silent.  Nothing here ships. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/oracle.h"
#include "../../oracle/orc_threads.h"

enum { N = 1 << 10 };
static unsigned char want[192 * N];

static void *caller(void *arg)
{
    const int seed = (int)(size_t)arg;
    unsigned char *buf = malloc(192 * N);
    for (int round = 0; round < 6; ++round) {
        for (size_t i = 0; i < 192 * N; ++i) buf[i] = (unsigned char)(i * 131 + 7);
        orc_bashF_batch(buf, N, 2 + (round + seed) % 7);
        if (memcmp(buf, want, 192 * N)) { fprintf(stderr, "pool result differs (caller %d round %d)\n", seed, round); exit(2); }
    }
    free(buf);
    return 0;
}

int main(void)
{
    for (size_t i = 0; i < 192 * N; ++i) want[i] = (unsigned char)(i * 131 + 7);
    orc_bashF_batch(want, N, 1);
    pthread_t t[2];
    for (size_t k = 0; k < 2; ++k) pthread_create(&t[k], 0, caller, (void *)k);
    for (size_t k = 0; k < 2; ++k) pthread_join(t[k], 0);
    orc_set_slice_reps(2);                /* in-place work simply iterates: two passes = bashF applied twice */
    unsigned char *twice = malloc(192 * N), *got = malloc(192 * N);
    memcpy(twice, want, 192 * N);
    orc_set_slice_reps(1);
    orc_bashF_batch(twice, N, 1);
    for (size_t i = 0; i < 192 * N; ++i) got[i] = (unsigned char)(i * 131 + 7);
    orc_set_slice_reps(2);
    orc_bashF_batch(got, N, 5);
    orc_set_slice_reps(1);
    if (memcmp(got, twice, 192 * N)) { fprintf(stderr, "slice repetition differs\n"); return 3; }
    if (!(orc_spin_rate(4, 0.05) > 0) || !(orc_spin_rate(1, 0.05) > 0)) return 4;
    puts("tsan pool ok");
    return 0;
}
