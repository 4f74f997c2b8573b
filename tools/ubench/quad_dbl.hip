// Micro-benchmark: the dependent doubling / addition chain of v Q (4 doublings + 1 addition per digit) with one
// lane per point (jac_dbl / jac_madd of bign_dev.hpp) against four lanes per point (quad_dbl / quad_add of
// bign_quad32.hpp beside this file), on wavefronts that are ALONE on their SIMD -- the regime of a small verification batch.
// Build: hipcc --offload-arch=gfx950 -O3 -I bee2_amd/csrc -I include -I tools/ubench tools/ubench/quad_dbl.hip -o tools/ubench/quad_dbl
// Run on the GPU: ./quad_dbl   (prints shader cycles per digit and checks that both chains reach the same point)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bign_quad32.hpp"
#include "bign_fe29.hpp"
#include "bign_quad29.hpp"
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

// call-based variant: every multiplication / squaring is a real function (code footprint ~6 KB instead of ~26 KB)
template <int K> __device__ __noinline__ fe29 c_mul(fe29 a, fe29 b) { fe29 r; f29_mul<K>(r, a, b); return r; }
template <int K> __device__ __noinline__ fe29 c_sqr(fe29 a) { fe29 r; f29_sqr<K>(r, a); return r; }
__device__ __forceinline__ void jac29_dbl_c(jac29 &T)
{
    fe29 delta, gamma, beta4, alpha, t0, t1;
    delta = c_sqr<1>(T.Z);
    gamma = c_sqr<1>(T.Y);
    beta4 = c_mul<4>(T.X, gamma);
    f29_sub(t0, T.X, delta);
    f29_add(t1, T.X, delta);
    alpha = c_mul<3>(t0, t1);
    T.Z = c_mul<2>(T.Y, T.Z);
    t0 = c_sqr<1>(alpha);
    f29_add(t1, beta4, beta4);
    f29_sub(T.X, t0, t1);
    f29_carry(T.X);
    t1 = c_sqr<8>(gamma);
    f29_sub(t0, beta4, T.X);
    t0 = c_mul<1>(alpha, t0);
    f29_sub(T.Y, t0, t1);
}
__device__ __forceinline__ void jac29_madd_c(jac29 &T, const aff29 &E)
{
    fe29 Z1Z1, U2, S2, H, HH, HHH, r, V, t;
    Z1Z1 = c_sqr<1>(T.Z);
    U2 = c_mul<1>(E.x, Z1Z1);
    t = c_mul<1>(T.Z, Z1Z1);
    S2 = c_mul<1>(E.y, t);
    f29_sub(H, U2, T.X);
    f29_sub(r, S2, T.Y);
    f29_carry(r);
    HH = c_sqr<1>(H);
    HHH = c_mul<1>(H, HH);
    V = c_mul<1>(T.X, HH);
    T.Z = c_mul<1>(T.Z, H);
    t = c_sqr<1>(r);
    f29_sub(t, t, HHH);
    f29_sub(t, t, V);
    f29_sub(T.X, t, V);
    f29_carry(T.X);
    f29_sub(t, V, T.X);
    t = c_mul<1>(r, t);
    S2 = c_mul<1>(T.Y, HHH);
    f29_sub(T.Y, t, S2);
}
template <bool CALLS> __device__ __forceinline__ void dbl29(jac29 &T) { if (CALLS) jac29_dbl_c(T); else jac29_dbl(T); }
template <bool CALLS> __device__ __forceinline__ void madd29(jac29 &T, const aff29 &E) { if (CALLS) jac29_madd_c(T, E); else jac29_madd(T, E); }

// the same chain on the signed 29-bit limbs of bign_fe29.hpp (one lane per point)
template <bool CALLS>
__global__ __launch_bounds__(64) void serial29_kernel(int digits, uint32_t *out, uint64_t *cycles, uint32_t zero)
{
    feT<8> gx, gy;
    fe_set_zero(gx);
#pragma unroll
    for (int i = 0; i < 8; ++i) gy.v[i] = c_yG[i];
    gx.v[0] = out[(size_t)blockIdx.x * 64 * 17 + threadIdx.x] & zero;   // per-lane (divergent) zero: keeps the chain off the scalar unit
    aff29 G;
    f29_from_words(G.x, gx);
    f29_from_words(G.y, gy);
    jac29 T;
    T.X = G.x; T.Y = G.y;
    for (int i = 0; i < 9; ++i) T.Z.l[i] = i == 0;
    jac29_dbl(T);
    const uint64_t t0 = now();
#pragma unroll 1
    for (int d = 0; d < digits; ++d) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) dbl29<CALLS>(T);
        madd29<CALLS>(T, G);
    }
    const uint64_t t1 = now();
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    feT<8> X, Z;
    f29_to_words(X, T.X); f29_to_words(Z, T.Z);
    fe_canon(X, X); fe_canon(Z, Z);
    for (int i = 0; i < 8; ++i) { out[idx * 17 + i] = X.v[i]; out[idx * 17 + 8 + i] = Z.v[i]; }
    out[idx * 17 + 16] = 1;
}

// four lanes per point on the 29-bit limbs (bign_quad29.hpp)
__global__ __launch_bounds__(64) void quad29_kernel(int digits, uint32_t *out, uint64_t *cycles, uint32_t zero)
{
    const uint32_t q = threadIdx.x & 3u;
    feT<8> gx, gy;
    fe_set_zero(gx);
#pragma unroll
    for (int i = 0; i < 8; ++i) gy.v[i] = c_yG[i];
    gx.v[0] = out[(size_t)blockIdx.x * 64 * 17 + threadIdx.x] & zero;
    qent29 E;
    f29_from_words(E.X, gx);
    f29_from_words(E.Y, gy);
    for (int i = 0; i < 9; ++i) E.Z.l[i] = E.ZZ.l[i] = i == 0;
    qjac29 T;
    T.X = E.X; T.Y = E.Y; T.Z = E.Z; T.D = E.Z;
    quad29_dbl(T, q);
    const uint64_t t0 = now();
#pragma unroll 1
    for (int d = 0; d < digits; ++d) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) quad29_dbl(T, q);
        quad29_add(T, E, q);
    }
    const uint64_t t1 = now();
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    feT<8> X, Z;
    f29_to_words(X, T.X); f29_to_words(Z, T.Z);
    fe_canon(X, X); fe_canon(Z, Z);
    for (int i = 0; i < 8; ++i) { out[idx * 17 + i] = X.v[i]; out[idx * 17 + 8 + i] = Z.v[i]; }
    out[idx * 17 + 16] = 1;
}

// two lanes per point on the 29-bit limbs
__global__ __launch_bounds__(64) void pair29_kernel(int digits, uint32_t *out, uint64_t *cycles, uint32_t zero)
{
    const uint32_t q = threadIdx.x & 1u;
    feT<8> gx, gy;
    fe_set_zero(gx);
#pragma unroll
    for (int i = 0; i < 8; ++i) gy.v[i] = c_yG[i];
    gx.v[0] = out[(size_t)blockIdx.x * 64 * 17 + threadIdx.x] & zero;
    qent29 E;
    f29_from_words(E.X, gx);
    f29_from_words(E.Y, gy);
    for (int i = 0; i < 9; ++i) E.Z.l[i] = E.ZZ.l[i] = i == 0;
    qjac29 T;
    T.X = E.X; T.Y = E.Y; T.Z = E.Z; T.D = E.Z;
    pair29_dbl(T, q);
    const uint64_t t0 = now();
#pragma unroll 1
    for (int d = 0; d < digits; ++d) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) pair29_dbl(T, q);
        pair29_add(T, E, q);
    }
    const uint64_t t1 = now();
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    feT<8> X, Z;
    f29_to_words(X, T.X); f29_to_words(Z, T.Z);
    fe_canon(X, X); fe_canon(Z, Z);
    for (int i = 0; i < 8; ++i) { out[idx * 17 + i] = X.v[i]; out[idx * 17 + 8 + i] = Z.v[i]; }
    out[idx * 17 + 16] = 1;
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
        uint32_t *oa, *ob, *oc, *od, *oe, *bad;
        uint64_t *ca, *cb;
        hipMalloc(&oa, n * 17 * 4); hipMalloc(&ob, n * 17 * 4); hipMalloc(&oc, n * 17 * 4); hipMalloc(&od, n * 17 * 4); hipMalloc(&oe, n * 17 * 4); hipMalloc(&bad, 4);
        hipMalloc(&ca, blocks * 8); hipMalloc(&cb, blocks * 8);
        hipMemset(bad, 0, 4);
        hipEvent_t e0, e1, e2, e3, e4, e5;
        hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3); hipEventCreate(&e4); hipEventCreate(&e5);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(serial_kernel, dim3(blocks), dim3(64), 0, 0, digits, oa, ca, 0u);
            hipEventRecord(e1);
            hipLaunchKernelGGL(quad_kernel, dim3(blocks), dim3(64), 0, 0, digits, ob, cb, 0u);
            hipEventRecord(e2);
            hipLaunchKernelGGL(serial29_kernel<false>, dim3(blocks), dim3(64), 0, 0, digits, oc, cb, 0u);
            hipEventRecord(e3);
            hipLaunchKernelGGL(quad29_kernel, dim3(blocks), dim3(64), 0, 0, digits, od, cb, 0u);
            hipEventRecord(e4);
            hipLaunchKernelGGL(pair29_kernel, dim3(blocks), dim3(64), 0, 0, digits, oe, cb, 0u);
            hipEventRecord(e5);
            hipDeviceSynchronize();
        }
        float ms_s, ms_q, ms_29, ms_29c, ms_p;
        hipEventElapsedTime(&ms_s, e0, e1); hipEventElapsedTime(&ms_q, e1, e2); hipEventElapsedTime(&ms_29, e2, e3); hipEventElapsedTime(&ms_29c, e3, e4); hipEventElapsedTime(&ms_p, e4, e5);
        hipLaunchKernelGGL(compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, oa, ob, n, bad);
        hipLaunchKernelGGL(compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, oa, oc, n, bad);
        hipLaunchKernelGGL(compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, oa, od, n, bad);
        hipLaunchKernelGGL(compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, oa, oe, n, bad);
        uint32_t hbad = 0;
        hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        std::vector<uint64_t> ha(blocks), hb(blocks);
        hipMemcpy(ha.data(), ca, blocks * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), cb, blocks * 8, hipMemcpyDeviceToHost);
        double sa = 0, sb = 0;
        for (int i = 0; i < blocks; ++i) { sa += ha[i]; sb += hb[i]; }
        // s_memtime counts at 100 MHz on gfx950: report the kernel times as well
        printf("%5d wavefronts (%.2f per SIMD): serial %.1f us, quad %.1f us (x%.2f), serial 29-bit limbs %.1f us (x%.2f), quad 29-bit limbs %.1f us (x%.2f), pair 29-bit limbs %.1f us (x%.2f); mismatches %u of %zu\n",
               blocks, blocks / 1024.0, ms_s * 1e3, ms_q * 1e3, ms_s / ms_q, ms_29 * 1e3, ms_s / ms_29, ms_29c * 1e3, ms_s / ms_29c, ms_p * 1e3, ms_s / ms_p, hbad, 4 * n);
        hipFree(oa); hipFree(ob); hipFree(bad); hipFree(ca); hipFree(cb);
    }
    return 0;
}
