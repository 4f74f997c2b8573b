// gfx950: which VALU instruction classes of DIFFERENT wavefronts can run side by side on one SIMD?
// 512-lane workgroups, wavefronts w and w + 4 share a SIMD; the first four ("older") run class P at
// s_setprio 3, the other four class Q at priority 0.  Reported: real shader cycles per instruction per SIMD
// (s_memtime against s_memrealtime), next to the two classes alone and their serial sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
enum { XOR, ALIGN, MAD, ADDC, MULLO, ADDCO, BITOP, LDSR, MADADDC, NOP };
template <int C> __device__ __forceinline__ void op16(uint32_t (&a)[16], uint64_t (&w)[8], uint32_t b, uint32_t addr)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (C == XOR)   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (C == ALIGN) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
        if (C == MAD)   asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i & 7]) : "v"(a[i]), "v"(b) : "s10", "s11");
        if (C == ADDC)  asm volatile("v_addc_co_u32 %0, s[10:11], %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11");
        if (C == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (C == ADDCO) asm volatile("v_add_co_u32 %0, s[10:11], %0, %1" : "+v"(a[i]) : "v"(b) : "s10", "s11");
        if (C == BITOP) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a[i]) : "v"(b));
        if (C == LDSR)  asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(addr) : "memory");
        if (C == MADADDC) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n s_nop 0\n v_addc_co_u32 %3, s[10:11], %3, %2, s[10:11]" : "+v"(w[i & 7]), "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 15]) : "s10", "s11");
    }
    if (C == LDSR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int P, int Q> __global__ __launch_bounds__(512) void k(uint32_t *out, uint32_t seed, int iters, unsigned long long *cyc)
{
    __shared__ uint32_t lds[4096];
    uint32_t a[16]; uint64_t w[8];
    for (int i = 0; i < 16; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    for (int i = 0; i < 8; ++i) w[i] = a[i] * 0x9E3779B97F4A7C15ull;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 4;
    __syncthreads();
    uint32_t b = seed | 1, addr = (threadIdx.x & 31) * 4;
    const int half = threadIdx.x >> 8;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    if (half == 0) { if (P != NOP) { __builtin_amdgcn_s_setprio(3); for (int it = 0; it < iters; ++it) op16<P>(a, w, b, addr); } }
    else           { if (Q != NOP) { for (int it = 0; it < iters; ++it) op16<Q>(a, w, b, addr); } }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    if (blockIdx.x == gridDim.x - 1 && (threadIdx.x == 0 || threadIdx.x == 256)) { cyc[2 * half] = t1 - t0; cyc[2 * half + 1] = r1 - r0; }
    uint32_t r = 0; for (int i = 0; i < 16; ++i) r ^= a[i]; for (int i = 0; i < 8; ++i) r ^= (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static double g_alone[16];
template <int P, int Q> int run(const char *pn, const char *qn, double ip, double iq)   // ip, iq: instructions per op16
{
    uint32_t *d; unsigned long long *c; CHK(hipMalloc(&d, 1024 * 1024 * 16)); CHK(hipMalloc(&c, 32)); CHK(hipMemset(c, 0, 32));
    const int iters = 32768, wps = 8;
    int blocks = 256 * wps / 2, threads = 512;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<P, Q><<<blocks, threads>>>(d, 12345, iters, c); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); k<P, Q><<<blocks, threads>>>(d, 12345, iters, c); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cy[4]; CHK(hipMemcpy(cy, c, 32, hipMemcpyDeviceToHost));
    double ghz = (cy[0] + cy[2]) / ((cy[1] + cy[3]) * 10.0);
    double cycles = ms * 1e6 * ghz;                         // kernel wall time in shader cycles
    double np = P == NOP ? 0 : 4.0 * iters * ip, nq = Q == NOP ? 0 : 4.0 * iters * iq;   // instructions per SIMD
    if (Q == NOP) g_alone[P] = cycles / np;
    if (P == NOP) { if (g_alone[Q] == 0) g_alone[Q] = cycles / nq; }
    double serial = np * g_alone[P] + nq * g_alone[Q];
    printf("%-10s (prio 3) | %-10s  %7.3f ms at %.3f GHz  %9.0f kcycles", pn, qn, ms, ghz, cycles / 1e3);
    if (P != NOP && Q != NOP) printf("   serial sum %9.0f k  -> %.2f of serial", serial / 1e3, cycles / serial);
    else printf("   %.2f cycles / instr (4 wavefronts/SIMD)", cycles / (np + nq));
    printf("\n");
    hipFree(d); hipFree(c); return 0;
}
int main()
{
    run<XOR, NOP>("xor", "-", 16, 0); run<ALIGN, NOP>("alignbit", "-", 16, 0); run<MAD, NOP>("mad_u64", "-", 16, 0); run<ADDC, NOP>("addc", "-", 16, 0);
    run<MULLO, NOP>("mul_lo", "-", 16, 0); run<ADDCO, NOP>("add_co", "-", 16, 0); run<BITOP, NOP>("bitop3", "-", 16, 0); run<LDSR, NOP>("ds_read_b32", "-", 16, 0);
    run<MADADDC, NOP>("mad+addc", "-", 32, 0);
    run<ALIGN, XOR>("alignbit", "xor", 16, 16); run<XOR, ALIGN>("xor", "alignbit", 16, 16);
    run<MAD, XOR>("mad_u64", "xor", 16, 16);    run<XOR, MAD>("xor", "mad_u64", 16, 16);
    run<MAD, ADDC>("mad_u64", "addc", 16, 16);  run<ADDC, MAD>("addc", "mad_u64", 16, 16);
    run<MAD, ALIGN>("mad_u64", "alignbit", 16, 16); run<ALIGN, MAD>("alignbit", "mad_u64", 16, 16);
    run<ADDC, XOR>("addc", "xor", 16, 16);      run<ADDC, ALIGN>("addc", "alignbit", 16, 16);
    run<MAD, MULLO>("mad_u64", "mul_lo", 16, 16);
    run<LDSR, XOR>("ds_read_b32", "xor", 16, 16); run<XOR, LDSR>("xor", "ds_read_b32", 16, 16);
    run<LDSR, ALIGN>("ds_read_b32", "alignbit", 16, 16); run<ALIGN, LDSR>("alignbit", "ds_read_b32", 16, 16);
    run<MADADDC, XOR>("mad+addc", "xor", 32, 16); run<MADADDC, MADADDC>("mad+addc", "mad+addc", 32, 32);
    return 0;
}
