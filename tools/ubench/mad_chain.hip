// Does the s_nop the compiler puts after EVERY v_mad_u64_u32 and v_addc_co_u32 of the GF(p) column sums (the gfx940 "VALU writes
// SGPR -> VALU reads it" wait state; 3 878 of the 9 900 instructions of bign_main_kernel<8>) cost throughput?  Form A: one column
// chain as compiled (mad, nop, addc, nop).  Form B: two independent chains interleaved with their own carry registers, no nop
// (the other chain's instruction is the wait state).  Form C: form A without the nops (timing only: the hazard is not covered).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mad_chain.hip -o mad_chain && ./mad_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 2048
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template <int FORM> __global__ void k(uint32_t *out, uint32_t seed)
{
    uint64_t accA = seed * 0x9E3779B97F4A7C15ull + threadIdx.x, accB = accA ^ 0x1234567;
    uint32_t cA = 0, cB = 0, a = seed | 1, b = (seed ^ 0x5bd1e995) + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (FORM == 0)
                asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n s_nop 0\n v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n s_nop 0\n"
                             "v_mad_u64_u32 %0, s[10:11], %3, %2, %0\n s_nop 0\n v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n s_nop 0"
                             : "+v"(accA), "+v"(cA) : "v"(a), "v"(b) : "s10", "s11");
            if (FORM == 1)
                asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %2, s[12:13], %5, %4, %2\n"
                             "v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n v_addc_co_u32 %3, s[12:13], 0, %3, s[12:13]"
                             : "+v"(accA), "+v"(cA), "+v"(accB), "+v"(cB) : "v"(a), "v"(b) : "s10", "s11", "s12", "s13");
            if (FORM == 3)        // D: ONE column (same accumulator, same counter), the two carries in their own registers, no s_nop
                asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n v_mad_u64_u32 %0, s[12:13], %3, %2, %0\n"
                             "v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n v_addc_co_u32 %1, s[12:13], 0, %1, s[12:13]"
                             : "+v"(accA), "+v"(cA) : "v"(a), "v"(b) : "s10", "s11", "s12", "s13");
            if (FORM == 2)
                asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n"
                             "v_mad_u64_u32 %0, s[10:11], %3, %2, %0\n v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]"
                             : "+v"(accA), "+v"(cA) : "v"(a), "v"(b) : "s10", "s11");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)accA ^ (uint32_t)(accA >> 32) ^ cA ^ (uint32_t)accB ^ (uint32_t)(accB >> 32) ^ cB;
}

template <int FORM> static int run(const char *name, int wps, uint32_t *d)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 4 * wps;                    // one 64-lane workgroup per wavefront slot
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(64), 0, 0, d, 12345u);
    CHK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(64), 0, 0, d, 12345u + r);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double mads = (double)blocks * 64 * ITERS * 16;          // 16 multiply-adds per unrolled step in every form
    printf("[wps=%d] %-44s %8.3f ms  %6.2f T mad-lanes/s\n", wps, name, best, mads / best / 1e9);
    return 0;
}
int main()
{
    uint32_t *d;
    CHK(hipMalloc(&d, 256 * 4 * 8 * 64 * 4));
    {   // same arithmetic in A, C and D: the outputs must agree if the hazard is respected (C: not guaranteed)
        uint32_t h[3][64];
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, 777u); CHK(hipMemcpy(h[0], d, 256, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, 777u); CHK(hipMemcpy(h[1], d, 256, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, 777u); CHK(hipMemcpy(h[2], d, 256, hipMemcpyDeviceToHost));
        int badC = 0, badD = 0;
        for (int i = 0; i < 64; ++i) { badC += h[0][i] != h[1][i]; badD += h[0][i] != h[2][i]; }
        printf("results: C differs from A in %d of 64 lanes, D in %d\n", badC, badD);
    }
    for (int wps : {1, 2, 3, 4, 8}) {
        if (run<0>("A: one chain, s_nop as compiled", wps, d)) return 1;
        if (run<1>("B: two chains interleaved, no s_nop", wps, d)) return 1;
        if (run<2>("C: one chain, no s_nop (timing only)", wps, d)) return 1;
        if (run<3>("D: one chain, carries in two registers, no nop", wps, d)) return 1;
    }
    return 0;
}
