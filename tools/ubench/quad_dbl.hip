// Micro-benchmark: the dependent doubling / addition chain of v Q (4 doublings + 1 addition per digit) with one
// lane per point (jac_dbl / jac_madd of bign_dev.hpp) against four lanes per point (quad_dbl / quad_add of
// bign_quad.hpp), on wavefronts that are ALONE on their SIMD -- the regime of a small verification batch.
// Build: hipcc --offload-arch=gfx950 -O3 -I bee2_amd/csrc -I include tools/ubench/quad_dbl.hip -o tools/ubench/quad_dbl
// Run on the GPU: ./quad_dbl   (prints shader cycles per digit and checks that both chains reach the same point)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bign_quad.hpp"
#include "bign_curves.inc"
using namespace bee2hip;

__constant__ uint32_t c_yG[8] = BIGN128_YG_LIMBS;

__device__ __forceinline__ uint64_t now()
{
    uint64_t t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// x = X / Z^2 compared across the two chains without inverting: X_a Z_b^2 == X_b Z_a^2
__global__ __launch_bounds__(64) void serial_kernel(int digits, uint32_t *out, uint64_t *cycles, uint32_t zero)
{
    affT<8> G;
    fe_set_zero(G.x);
#pragma unroll
    for (int i = 0; i < 8; ++i) G.y.v[i] = c_yG[i];
    G.x.v[0] = zero;                            // keep the compiler from folding the chain
    jacT<8> T;
    T.X = G.x; T.Y = G.y; fe_set_one(T.Z);
    jac_dbl(T);
    bool ok = true;
    const uint64_t t0 = now();
#pragma unroll 1
    for (int d = 0; d < digits; ++d) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) jac_dbl(T);
        ok &= jac_madd(T, G);
    }
    const uint64_t t1 = now();
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    fe_canon(T.X, T.X); fe_canon(T.Z, T.Z);
    for (int i = 0; i < 8; ++i) { out[idx * 17 + i] = T.X.v[i]; out[idx * 17 + 8 + i] = T.Z.v[i]; }
    out[idx * 17 + 16] = ok;
}

__global__ __launch_bounds__(64) void quad_kernel(int digits, uint32_t *out, uint64_t *cycles, uint32_t zero)
{
    const uint32_t q = threadIdx.x & 3u;
    qentT<8> E;
    fe_set_zero(E.X);
#pragma unroll
    for (int i = 0; i < 8; ++i) E.Y.v[i] = c_yG[i];
    E.X.v[0] = zero;
    fe_set_one(E.Z); fe_set_one(E.ZZ); fe_set_one(E.ZZZ);
    qjacT<8> T;
    T.X = E.X; T.Y = E.Y; fe_set_one(T.Z); fe_set_one(T.D);
    quad_dbl(T, q);
    bool ok = true;
    const uint64_t t0 = now();
#pragma unroll 1
    for (int d = 0; d < digits; ++d) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) quad_dbl(T, q);
        ok &= quad_add(T, E, q);
    }
    const uint64_t t1 = now();
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    fe_canon(T.X, T.X); fe_canon(T.Z, T.Z);
    for (int i = 0; i < 8; ++i) { out[idx * 17 + i] = T.X.v[i]; out[idx * 17 + 8 + i] = T.Z.v[i]; }
    out[idx * 17 + 16] = ok;
}

// host check of X_a Z_b^2 == X_b Z_a^2 mod p with __int128-free Python-ish big arithmetic: done on the device instead
__global__ void compare_kernel(const uint32_t *a, const uint32_t *b, size_t n, uint32_t *bad)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    feT<8> Xa, Za, Xb, Zb, l, r;
    for (int k = 0; k < 8; ++k) { Xa.v[k] = a[i * 17 + k]; Za.v[k] = a[i * 17 + 8 + k]; Xb.v[k] = b[i * 17 + k]; Zb.v[k] = b[i * 17 + 8 + k]; }
    fe_sqr(l, Zb); fe_mul(l, l, Xa);
    fe_sqr(r, Za); fe_mul(r, r, Xb);
    fe_sub(l, l, r);
    if (!fe_is_zero(l) || fe_is_zero(Za) || fe_is_zero(Zb) || !a[i * 17 + 16] || !b[i * 17 + 16]) atomicAdd(bad, 1u);
}

int main(int argc, char **argv)
{
    const int digits = argc > 1 ? atoi(argv[1]) : 32;
    for (int blocks : {256, 1024, 2048, 4096}) {
        const size_t n = (size_t)blocks * 64;
        uint32_t *oa, *ob, *bad;
        uint64_t *ca, *cb;
        hipMalloc(&oa, n * 17 * 4); hipMalloc(&ob, n * 17 * 4); hipMalloc(&bad, 4);
        hipMalloc(&ca, blocks * 8); hipMalloc(&cb, blocks * 8);
        hipMemset(bad, 0, 4);
        hipEvent_t e0, e1, e2;
        hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(serial_kernel, dim3(blocks), dim3(64), 0, 0, digits, oa, ca, 0u);
            hipEventRecord(e1);
            hipLaunchKernelGGL(quad_kernel, dim3(blocks), dim3(64), 0, 0, digits, ob, cb, 0u);
            hipEventRecord(e2);
            hipDeviceSynchronize();
        }
        float ms_s, ms_q;
        hipEventElapsedTime(&ms_s, e0, e1); hipEventElapsedTime(&ms_q, e1, e2);
        hipLaunchKernelGGL(compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, oa, ob, n, bad);
        uint32_t hbad = 0;
        hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        std::vector<uint64_t> ha(blocks), hb(blocks);
        hipMemcpy(ha.data(), ca, blocks * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), cb, blocks * 8, hipMemcpyDeviceToHost);
        double sa = 0, sb = 0;
        for (int i = 0; i < blocks; ++i) { sa += ha[i]; sb += hb[i]; }
        // s_memtime counts at 100 MHz on gfx950: report the kernel times as well
        printf("%5d wavefronts (%.2f per SIMD): serial %.1f us, quad %.1f us (x%.2f); memtime ticks per digit %.1f / %.1f; mismatches %u of %zu\n",
               blocks, blocks / 1024.0, ms_s * 1e3, ms_q * 1e3, ms_s / ms_q, sa / blocks / digits, sb / blocks / digits, hbad, n);
        hipFree(oa); hipFree(ob); hipFree(bad); hipFree(ca); hipFree(cb);
    }
    return 0;
}
