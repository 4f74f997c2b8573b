// Where does the VALU class-mixing penalty of gfx950 come from?  v_alignbit_b32 (half rate, 4 cycles) and
// v_xor_b32 (full rate, 2 cycles) streams, (a) separated by wavefront (even wavefronts only alignbit, odd only
// xor), (b) mixed inside every wavefront in runs of G.  All chains independent.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define A(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(e[i]) : "v"(b));

// MODE 0: every wavefront NA alignbit then NX xor per iteration, in runs of G (G = run length of each class,
//         classes alternate; totals per iteration NA, NX)
// MODE 1: wavefronts with (wave id & 1) == 0 run only alignbit (NA per iteration), the others only xor (NX)
template <int MODE, int NA, int NX, int G> __global__ void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[16], e[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed * (i + 1) + threadIdx.x; e[i] = a[i] ^ 0x1234567u; }
    uint32_t b = seed | 1;
    const int wave = threadIdx.x >> 8;   // 512-lane workgroups: wavefronts w and w + 4 share a SIMD, so every SIMD gets both kinds
    if (MODE == 1) {
        if (wave & 1) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < NX; ++i) X(i & 15)
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < NA; ++i) A(i & 15)
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            constexpr int RUNS = NA / G;                   // NA == NX * (NA / NX); run of G alignbit, then G * NX / NA xor
#pragma unroll
            for (int r = 0; r < RUNS; ++r) {
#pragma unroll
                for (int i = 0; i < G; ++i) A((r * G + i) & 15)
#pragma unroll
                for (int i = 0; i < G * NX / NA; ++i) X((r * G + i) & 15)
            }
        }
    }
    uint32_t r = 0; for (int i = 0; i < 16; ++i) r ^= a[i] ^ e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int g_wps = 4;
template <int MODE, int NA, int NX, int G> int run(const char *name)
{
    uint32_t *d; CHK(hipMalloc(&d, 1024 * 1024 * 16));
    const int iters = 16384;
    int blocks = 256 * g_wps / 2, threads = 512;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<MODE, NA, NX, G><<<blocks, threads>>>(d, 12345, iters); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); k<MODE, NA, NX, G><<<blocks, threads>>>(d, 12345, iters); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double waves = (double)blocks * threads / 64;
    // instructions issued per SIMD: MODE 0: waves/1024 * iters * (NA + NX); MODE 1: half the waves each
    double na = MODE ? waves / 2 * iters * NA : waves * iters * NA, nx = MODE ? waves / 2 * iters * NX : waves * iters * NX;
    double cyc = ms * 1e-3 * 2.4e9 * 1024;
    printf("[wps=%d] %-46s %7.3f ms  %5.2f cyc/instr  (ideal 4A+2X: %5.2f)\n", g_wps, name, ms, cyc / (na + nx), (4 * na + 2 * nx) / (na + nx));
    hipFree(d); return 0;
}
int main()
{
    for (int w : {4, 8}) {
        g_wps = w;
        run<0, 16, 0, 16>("alignbit only");
        run<1, 0, 16, 16>("xor only, odd wavefronts (half the SIMD's waves)");
        run<1, 16, 0, 16>("alignbit only, even wavefronts");
        run<1, 16, 16, 16>("split by wavefront: 16 A | 16 X");
        run<1, 16, 32, 16>("split by wavefront: 16 A | 32 X");
        run<0, 16, 16, 1>("mixed in every wavefront, runs of 1");
        run<0, 16, 16, 4>("mixed, runs of 4");
        run<0, 16, 16, 16>("mixed, runs of 16");
        run<0, 64, 64, 64>("mixed, runs of 64");
        run<0, 16, 32, 16>("mixed, 16 A + 32 X, runs of 16 / 32");
        run<0, 64, 128, 64>("mixed, 64 A + 128 X, runs of 64 / 128");
    }
    return 0;
}
