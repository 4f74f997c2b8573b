// gfx950: how long must a wavefront stay in one VALU class (v_alignbit_b32 = half rate, v_xor_b32 = full rate)
// before half-rate and full-rate work of different wavefronts on one SIMD overlap as they do when every
// wavefront is pure (tools/ubench/valu_overlap.hip, "split")?  Every wavefront alternates runs of R alignbit and
// R * 7 / 4 xor (the bash-f ratio 64 : 112); the second wavefront of each SIMD pair starts with the other class.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define A(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[(i) & 15]) : "v"(b));
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(e[(i) & 15]) : "v"(b));
__device__ __forceinline__ void runA16(uint32_t (&a)[16], uint32_t b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) A(i)
}
__device__ __forceinline__ void runX28(uint32_t (&e)[16], uint32_t b) {
#pragma unroll
    for (int i = 0; i < 28; ++i) X(i)
}
// REPS: number of 16 / 28 blocks per run; ANTI: second wavefront of a pair starts with xor
template <int REPS, bool ANTI> __global__ __launch_bounds__(512) void k(uint32_t *out, uint32_t seed, int iters, unsigned long long *cyc)
{
    unsigned long long t0, t1;
    uint32_t a[16], e[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed * (i + 1) + threadIdx.x; e[i] = a[i] ^ 0x1234567u; }
    uint32_t b = seed | 1;
    const int half = threadIdx.x >> 8;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (ANTI && half) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll 1
            for (int r = 0; r < REPS; ++r) runX28(e, b);
#pragma unroll 1
            for (int r = 0; r < REPS; ++r) runA16(a, b);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll 1
            for (int r = 0; r < REPS; ++r) runA16(a, b);
#pragma unroll 1
            for (int r = 0; r < REPS; ++r) runX28(e, b);
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
    uint32_t r = 0; for (int i = 0; i < 16; ++i) r ^= a[i] ^ e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int g_wps = 8;
template <int REPS, bool ANTI> int run()
{
    uint32_t *d; unsigned long long *c; CHK(hipMalloc(&d, 1024 * 1024 * 16)); CHK(hipMalloc(&c, 8));
    const int iters = 16384 / REPS;
    int blocks = 256 * g_wps / 2, threads = 512;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<REPS, ANTI><<<blocks, threads>>>(d, 12345, iters, c); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); k<REPS, ANTI><<<blocks, threads>>>(d, 12345, iters, c); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double instr = (double)g_wps * iters * REPS * 44;
    unsigned long long cy; CHK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
    printf("[wps=%d] runs of %5d H / %5d F, %s  %7.3f ms | wave 0: %10llu cycles (%.2f GHz) = %5.2f cyc/instr/SIMD\n", g_wps, REPS * 16, REPS * 28,
           ANTI ? "anti-phase start" : "same phase      ", ms, cy, cy / (ms * 1e6), cy / instr);
    hipFree(d); return 0;
}
int main()
{
    for (int w : {4, 8}) {
        g_wps = w;
        run<1, false>(); run<1, true>(); run<4, false>(); run<4, true>(); run<16, false>(); run<16, true>();
        run<64, false>(); run<64, true>(); run<256, false>(); run<256, true>(); run<1024, false>(); run<1024, true>();
        run<4096, false>(); run<4096, true>();
    }
    return 0;
}
