/* bee2hip_internal.h -- test / bench / experiment hooks of libbee2hip.so.
 *
 * NOT part of the product ABI and NOT in the product library: these functions exist only in libbee2hip_exp.so, the
 * -DBEE2HIP_EXPERIMENTS build of the same sources (bee2_amd/csrc/Makefile, target exp).  Nothing here replaces a bee2
 * interface, a bee2 caller never needs it, and the names may change between builds.  The product ABI is include/bee2hip.h (bee2's own
 * symbols plus the bee2hip_* batch entry points).  tests/, bench.py and tools/ use these hooks to
 * reach device code that has no entry point of its own (the field arithmetic) and to switch
 * experiment variants of a kernel inside one process.
 */
#ifndef BEE2HIP_INTERNAL_H
#define BEE2HIP_INTERNAL_H
#include "bee2hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* time `reps` launches of one kernel with hipEvents on `stream`; returns the
   average milliseconds per launch in *ms (used by bench.py's roofline object).
   which: 0 bashF_batch, 1 beltCTR_blocks, 2 bign128Verify_batch, 3 bashHash_beltMAC */
err_t bee2hip_time_kernel(int which, int reps, void *d_a, void *d_b, void *d_c, void *d_d,
                          size_t n, size_t aux, void *stream, float *ms);

/* self-test hook: element-wise GF(2^256-189) ops on device arrays of 8 x u32 limbs
   (op: 0 mul, 1 sqr, 2 add, 3 sub, 4 inv, 5 3*mul, 6 8*sqr, 7 canon, 8 x(2P)) */
err_t bee2hip_debug_fe(int op, const void *d_a, const void *d_b, void *d_out, size_t n, void *stream);
/* same over GF(2^(2l) - c) for l in {128, 192, 256} (8 / 12 / 16 limbs per element) */
err_t bee2hip_debug_feL(size_t l, int op, const void *d_a, const void *d_b, void *d_out, size_t n,
                        void *stream);

/* one wavefront spins for `us` microseconds on `stream` and writes {shader cycles, 100 MHz ticks} to d_out16:
   the clock the chip sustains under whatever runs beside it */
err_t bee2hip_internal_clock_probe(void *d_out16, unsigned us, void *stream);
/* experiment switch (process-wide, not thread-safe): key 0 = bashF batch kernel variant (tools/ab/bashf_ab.py; -1 = product),
   key 1 = beltCTR kernel variant (tools/ab/belt_ab.py; 0 = product), key 2 = kernels of the 256-bit verification
   (0 = by batch size, 1 = 32-bit limbs, 2 = 29-bit limbs, 3 = one signature per quad / pair by size, 0x43 = quads,
   0x23 = pairs, 0x83 = quad + helper quad; tests force each) */
err_t bee2hip_internal_tune(int key, int value);
/* (key 3 = size limit of the pinned staging buffer; key 4 = path of the drop-in layer's small calls, as the environment
   variable BEE2HIP_FORCE: 0 auto (by size), 1 gpu, 2 cpu -- bee2_amd/csrc/host_small.hpp; key 5 = fault injection: the
   next `value` GPU attempts of drop-in helpers report a device failure; keys 6 / 7 = log2 of the chunk of the duplex host
   pipeline in bashF states / belt blocks; key 8 = parts a big verification batch is split into (0 by size, 1 never);
   key 9 = quarter and half chunks at both ends of the duplex pipeline (0 = product: measured -2 %); key 10 = lanes per scalar of k G on the signing side
   (0 = by batch size: 64 / 16 / 4 / 1, and 7 from 2^18 scalars on the 256-bit curve; 1, 4, 16, 64 forced; 7 = one lane, signed 7-bit windows looked up in LDS (256-bit curve; elsewhere as 1); 101 = one lane on the 4-bit windows of round 2, 102 = one lane, signed 6-bit windows, complete additions); key 11 = chunked upload of host-pointer
   verification batches of 2^19 signatures and more (1 = product); key 12 = largest workgroup of the signing side's hashing
   kernels for batches of 2^16 and more (0 = 1024, the product; 256 = round 2); key 13 = belt table of the fused kernel;
   keys 14 / 15 = fault injection into the duplex host pipeline: the next `15` pipelines fail when they reach chunk `14`) */
/* drop-in helper calls so far: which = 0 taken on the host path, 1 on the GPU, 2 finished on the host after the GPU path
   failed twice */
unsigned long long bee2hip_internal_stat(int which);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BEE2HIP_INTERNAL_H */
