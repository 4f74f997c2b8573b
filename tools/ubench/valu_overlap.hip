// gfx950: can half-rate (v_alignbit_b32) and full-rate (v_xor_b32) VALU work of DIFFERENT wavefronts on one SIMD
// overlap, and what keeps wavefronts that run the same code from doing so?  Time is read with s_memtime by
// wavefront 0 of workgroup 0 (shader cycles); every kernel fills the chip with exactly `wps` resident
// wavefronts per SIMD (512-lane workgroups: wavefronts w and w + 4 of a workgroup share a SIMD).
// The "bash" pattern is one bash-f round in the staged order: F80 H48 F32 H16 (F = xor, H = alignbit).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define A(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[(i) & 15]) : "v"(b));
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(e[(i) & 15]) : "v"(b));
template <int N> __device__ __forceinline__ void runA(uint32_t (&a)[16], uint32_t b) {
#pragma unroll
    for (int i = 0; i < N; ++i) A(i)
}
template <int N> __device__ __forceinline__ void runX(uint32_t (&e)[16], uint32_t b) {
#pragma unroll
    for (int i = 0; i < N; ++i) X(i)
}
enum { PURE_X, PURE_A, SPLIT_16_16, SPLIT_16_28, BASH, BASH_PRIO, BASH_OFFSET, BASH_BARRIER, BASH_BARRIER_BAL, FINE_1_2, FINE_2_4, FINE_4_7,
       BASH_SLEEP, BASH_W4, BASH_W4_OFFSET, BASH_W4_BARRIER, BASH_DYNPRIO_H, BASH_DYNPRIO_F, BASH_W4_DYNPRIO_H, FINE_DYNPRIO_H, SPLIT_YOUNG_H };

template <int MODE> __global__ __launch_bounds__(512) void k(uint32_t *out, uint32_t seed, int iters, long long *cyc)
{
    uint32_t a[16], e[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed * (i + 1) + threadIdx.x; e[i] = a[i] ^ 0x1234567u; }
    uint32_t b = seed | 1;
    const int wave = threadIdx.x >> 6, half = wave >> 2;          // half 0 / 1: the two wavefronts of a SIMD pair
    if (MODE == BASH_PRIO) { if (half) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
    if (MODE == BASH_SLEEP) { if (half) __builtin_amdgcn_s_sleep(40); }
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    if (MODE == PURE_X) for (int it = 0; it < iters; ++it) runX<16>(e, b);
    if (MODE == PURE_A) for (int it = 0; it < iters; ++it) runA<16>(a, b);
    if (MODE == SPLIT_16_16) { if (half) for (int it = 0; it < iters; ++it) runX<16>(e, b); else for (int it = 0; it < iters; ++it) runA<16>(a, b); }
    if (MODE == SPLIT_16_28) { if (half) for (int it = 0; it < iters; ++it) runX<28>(e, b); else for (int it = 0; it < iters; ++it) runA<16>(a, b); }
    if (MODE == BASH || MODE == BASH_PRIO || MODE == BASH_SLEEP)
        for (int it = 0; it < iters; ++it) { runX<80>(e, b); runA<48>(a, b); runX<32>(e, b); runA<16>(a, b); }
    if (MODE == BASH_OFFSET) {          // second wavefront of each pair starts one segment later in the cycle
        if (half) { for (int it = 0; it < iters; ++it) { runA<48>(a, b); runX<32>(e, b); runA<16>(a, b); runX<80>(e, b); } }
        else      { for (int it = 0; it < iters; ++it) { runX<80>(e, b); runA<48>(a, b); runX<32>(e, b); runA<16>(a, b); } }
    }
    if (MODE == BASH_BARRIER) {         // the same, segment boundaries synchronised in the workgroup
        if (half) { for (int it = 0; it < iters; ++it) { runA<48>(a, b); __builtin_amdgcn_s_barrier(); runX<32>(e, b); __builtin_amdgcn_s_barrier(); runA<16>(a, b); __builtin_amdgcn_s_barrier(); runX<80>(e, b); __builtin_amdgcn_s_barrier(); } }
        else      { for (int it = 0; it < iters; ++it) { runX<80>(e, b); __builtin_amdgcn_s_barrier(); runA<48>(a, b); __builtin_amdgcn_s_barrier(); runX<32>(e, b); __builtin_amdgcn_s_barrier(); runA<16>(a, b); __builtin_amdgcn_s_barrier(); } }
    }
    if (MODE == BASH_BARRIER_BAL) {     // balanced segments: F56 H32 F56 H32 against H32 F56 H32 F56
        if (half) { for (int it = 0; it < iters; ++it) { runA<32>(a, b); __builtin_amdgcn_s_barrier(); runX<56>(e, b); __builtin_amdgcn_s_barrier(); runA<32>(a, b); __builtin_amdgcn_s_barrier(); runX<56>(e, b); __builtin_amdgcn_s_barrier(); } }
        else      { for (int it = 0; it < iters; ++it) { runX<56>(e, b); __builtin_amdgcn_s_barrier(); runA<32>(a, b); __builtin_amdgcn_s_barrier(); runX<56>(e, b); __builtin_amdgcn_s_barrier(); runA<32>(a, b); __builtin_amdgcn_s_barrier(); } }
    }
    if (MODE == FINE_1_2) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { A(i) X(2 * i) X(2 * i + 1) } }
    if (MODE == FINE_2_4) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { A(2 * i) A(2 * i + 1) X(4 * i) X(4 * i + 1) X(4 * i + 2) X(4 * i + 3) } }
    if (MODE == FINE_4_7) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { A(4 * i) A(4 * i + 1) A(4 * i + 2) A(4 * i + 3) X(7 * i) X(7 * i + 1) X(7 * i + 2) X(7 * i + 3) X(7 * i + 4) X(7 * i + 5) X(7 * i + 6) } }
    // priority follows the instruction class: raised for the half-rate runs (DYNPRIO_H) or for the full-rate runs (DYNPRIO_F)
    if (MODE == BASH_DYNPRIO_H) for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_setprio(0); runX<80>(e, b); __builtin_amdgcn_s_setprio(3); runA<48>(a, b);
        __builtin_amdgcn_s_setprio(0); runX<32>(e, b); __builtin_amdgcn_s_setprio(3); runA<16>(a, b); }
    if (MODE == BASH_DYNPRIO_F) for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_setprio(3); runX<80>(e, b); __builtin_amdgcn_s_setprio(0); runA<48>(a, b);
        __builtin_amdgcn_s_setprio(3); runX<32>(e, b); __builtin_amdgcn_s_setprio(0); runA<16>(a, b); }
    if (MODE == BASH_W4_DYNPRIO_H) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_s_setprio(0); runX<40>(e, b); __builtin_amdgcn_s_setprio(3); runA<24>(a, b);
            __builtin_amdgcn_s_setprio(0); runX<16>(e, b); __builtin_amdgcn_s_setprio(3); runA<8>(a, b); } }
    if (MODE == FINE_DYNPRIO_H) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { __builtin_amdgcn_s_setprio(3); A(4 * i) A(4 * i + 1) A(4 * i + 2) A(4 * i + 3) __builtin_amdgcn_s_setprio(0); X(7 * i) X(7 * i + 1) X(7 * i + 2) X(7 * i + 3) X(7 * i + 4) X(7 * i + 5) X(7 * i + 6) } }
    if (MODE == SPLIT_YOUNG_H) { if (!half) for (int it = 0; it < iters; ++it) runX<28>(e, b); else for (int it = 0; it < iters; ++it) runA<16>(a, b); }
    // W = 4 staging: two half-rounds F40 H24 F16 H8
    if (MODE == BASH_W4) for (int it = 0; it < iters; ++it) { runX<40>(e, b); runA<24>(a, b); runX<16>(e, b); runA<8>(a, b); runX<40>(e, b); runA<24>(a, b); runX<16>(e, b); runA<8>(a, b); }
    if (MODE == BASH_W4_OFFSET) {
        if (half) for (int it = 0; it < iters; ++it) { runA<24>(a, b); runX<16>(e, b); runA<8>(a, b); runX<40>(e, b); runA<24>(a, b); runX<16>(e, b); runA<8>(a, b); runX<40>(e, b); }
        else      for (int it = 0; it < iters; ++it) { runX<40>(e, b); runA<24>(a, b); runX<16>(e, b); runA<8>(a, b); runX<40>(e, b); runA<24>(a, b); runX<16>(e, b); runA<8>(a, b); }
    }
    if (MODE == BASH_W4_BARRIER) {
#define BAR __builtin_amdgcn_s_barrier();
        if (half) for (int it = 0; it < iters; ++it) { runA<24>(a, b); BAR runX<16>(e, b); BAR runA<8>(a, b); BAR runX<40>(e, b); BAR runA<24>(a, b); BAR runX<16>(e, b); BAR runA<8>(a, b); BAR runX<40>(e, b); BAR }
        else      for (int it = 0; it < iters; ++it) { runX<40>(e, b); BAR runA<24>(a, b); BAR runX<16>(e, b); BAR runA<8>(a, b); BAR runX<40>(e, b); BAR runA<24>(a, b); BAR runX<16>(e, b); BAR runA<8>(a, b); BAR }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    // the LAST workgroup's wavefronts are the youngest: under oldest-first arbitration they finish last
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
    uint32_t r = 0; for (int i = 0; i < 16; ++i) r ^= a[i] ^ e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int g_wps = 4;
template <int MODE> int run(const char *name, double nA, double nX)     // nA, nX: per iteration, averaged over the wavefronts
{
    uint32_t *d; long long *c; CHK(hipMalloc(&d, 1024 * 1024 * 16)); CHK(hipMalloc(&c, 16));
    const int iters = ((MODE <= SPLIT_16_28 || MODE == SPLIT_YOUNG_H) ? 16 : MODE == FINE_DYNPRIO_H ? 6 : MODE >= FINE_1_2 && MODE <= FINE_4_7 ? 6 : 2) * 4096;
    int blocks = 256 * g_wps / 2, threads = 512;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<MODE><<<blocks, threads>>>(d, 12345, iters, c); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); k<MODE><<<blocks, threads>>>(d, 12345, iters, c); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    long long cy2[2]; CHK(hipMemcpy(cy2, c, 16, hipMemcpyDeviceToHost)); long long cyc = cy2[0]; double ghz = cy2[0] / (cy2[1] * 10.0);
    double per_simd_instr = (double)g_wps * iters * (nA + nX);
    // wall time -> cycles with the clock that wavefront 0 saw: cyc ticks during (its share of) the kernel
    printf("[wps=%d] %-58s wall %7.3f ms | youngest wave: %9lld cycles (%4.0f%% of wall) at %.3f GHz | %5.2f cyc/instr/SIMD  (serial 4H+2F %4.2f)\n", g_wps, name,
           ms, cyc, 100.0 * cyc / (ms * 1e6 * ghz), ghz, ms * 1e6 * ghz / per_simd_instr, (4 * nA + 2 * nX) / (nA + nX));
    hipFree(d); hipFree(c); return 0;
}
int main()
{
    for (int w : {4, 6, 8}) {
        g_wps = w;
        run<PURE_X>("xor only", 0, 16);
        run<PURE_A>("alignbit only", 16, 0);
        run<SPLIT_16_16>("split: half the wavefronts 16 H, the others 16 F", 8, 8);
        run<SPLIT_16_28>("split: 16 H | 28 F", 8, 14);
        run<BASH>("bash round in every wavefront: F80 H48 F32 H16", 64, 112);
        run<BASH_PRIO>("  + s_setprio 2 on the second wavefront of each pair", 64, 112);
        run<BASH_SLEEP>("  + second wavefront starts after s_sleep 40", 64, 112);
        run<BASH_OFFSET>("  second wavefront starts one segment later", 64, 112);
        run<BASH_BARRIER>("  one segment later + s_barrier per segment", 64, 112);
        run<BASH_BARRIER_BAL>("  balanced F56 H32 F56 H32 vs H32 F56 H32 F56 + s_barrier", 64, 112);
        run<BASH_W4>("W=4 round: (F40 H24 F16 H8) x 2", 64, 112);
        run<BASH_W4_OFFSET>("  second wavefront one segment later", 64, 112);
        run<BASH_W4_BARRIER>("  one segment later + s_barrier", 64, 112);
        run<BASH_DYNPRIO_H>("bash round, s_setprio 3 during H runs, 0 during F runs", 64, 112);
        run<BASH_DYNPRIO_F>("bash round, s_setprio 3 during F runs, 0 during H runs", 64, 112);
        run<BASH_W4_DYNPRIO_H>("W=4 round, s_setprio 3 during H runs", 64, 112);
        run<FINE_DYNPRIO_H>("fine (H4 F7) x 4, s_setprio 3 during H", 16, 28);
        run<SPLIT_YOUNG_H>("split: 28 F (older wavefronts) | 16 H (younger)", 8, 14);
        run<FINE_1_2>("fine: (H F F) x 16", 16, 32);
        run<FINE_2_4>("fine: (H H F F F F) x 8", 16, 32);
        run<FINE_4_7>("fine: (H4 F7) x 4", 16, 28);
    }
    return 0;
}
