// diagnostic: quad29_dbl / quad29_add against jac29_dbl / jac29_madd / 32-bit jac_dbl on the same inputs (residues must agree)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "bign_quad29.hpp"
#include "bign_curves.inc"
using namespace bee2hip;
__constant__ uint32_t c_yG[8] = BIGN128_YG_LIMBS;

__device__ void put(uint32_t *o, const fe29 &a) { feT<8> w; f29_to_words(w, a); fe_canon(w, w); for (int i = 0; i < 8; ++i) o[i] = w.v[i]; }
__device__ void put32(uint32_t *o, feT<8> a) { fe_canon(a, a); for (int i = 0; i < 8; ++i) o[i] = a.v[i]; }

__global__ void check(uint32_t *out, const uint32_t *zin)
{
    const uint32_t q = threadIdx.x & 3u;
    feT<8> gx, gy;
    fe_set_zero(gx);
    for (int i = 0; i < 8; ++i) gy.v[i] = c_yG[i];
    gx.v[0] = zin[threadIdx.x];
    // 32-bit reference: 2G, 4G, 4G + G
    jacT<8> R; R.X = gx; R.Y = gy; fe_set_one(R.Z);
    affT<8> A; A.x = gx; A.y = gy;
    jac_dbl(R);
    uint32_t *o = out + threadIdx.x * 200;
    put32(o + 0, R.X); put32(o + 8, R.Y); put32(o + 16, R.Z);
    jac_dbl(R);
    put32(o + 24, R.X); put32(o + 32, R.Y); put32(o + 40, R.Z);
    jac_madd(R, A);
    put32(o + 48, R.X); put32(o + 56, R.Y); put32(o + 64, R.Z);
    // quad29
    qent29 E;
    f29_from_words(E.X, gx); f29_from_words(E.Y, gy);
    for (int i = 0; i < 9; ++i) E.Z.l[i] = E.ZZ.l[i] = i == 0;
    qjac29 T; T.X = E.X; T.Y = E.Y; T.Z = E.Z; T.D = E.Z;
    quad29_dbl(T, q);
    put(o + 72, T.X); put(o + 80, T.Y); put(o + 88, T.Z);
    quad29_dbl(T, q);
    put(o + 96, T.X); put(o + 104, T.Y); put(o + 112, T.Z);
    quad29_add(T, E, q);
    put(o + 120, T.X); put(o + 128, T.Y); put(o + 136, T.Z); put(o + 144, T.D);
    {   // the pair (two lanes per point) forms
        const uint32_t pp = threadIdx.x & 1u;
        qjac29 P; P.X = E.X; P.Y = E.Y; P.Z = E.Z; P.D = E.Z;
        pair29_dbl(P, pp);
        uint32_t *o2 = out + 64 * 200 + threadIdx.x * 80;
        put(o2 + 0, P.X); put(o2 + 8, P.Y); put(o2 + 16, P.Z);
        pair29_dbl(P, pp);
        put(o2 + 24, P.X); put(o2 + 32, P.Y); put(o2 + 40, P.Z);
        pair29_add(P, E, pp);
        put(o2 + 48, P.X); put(o2 + 56, P.Y); put(o2 + 64, P.Z);
    }
    {   // pair vs quad from a general point (5G in Jacobian form), and the pair's first doubling step by step
        qjac29 A = T, B = T;
        quad29_dbl(A, q);
        pair29_dbl(B, threadIdx.x & 1u);
        uint32_t *o3 = out + 64 * 280 + threadIdx.x * 64;
        put(o3 + 0, A.X); put(o3 + 8, A.Y); put(o3 + 16, A.Z); put(o3 + 24, A.D);
        put(o3 + 32, B.X); put(o3 + 40, B.Y); put(o3 + 48, B.Z); put(o3 + 56, B.D);
    }
    {   // level C of the first doubling, by hand, with dumps
        qjac29 U; U.X = E.X; U.Y = E.Y; U.Z = E.Z; U.D = E.Z;
        const bool q0 = q == 0, q1 = q == 1, q2 = q == 2, lo = q < 2;
        fe29 a, b, r, gamma, alpha, b4, t;
        q29_pick(a, q2, U.X, U.D); q29_pick(a, lo, U.Y, a); q29_pick(b, q1, U.Z, a);
        f29_mul_k(r, a, b, q0 ? 1 : q1 ? 2 : 3);
        q29_bcast<0>(gamma, r); q29_bcast<1>(U.Z, r); q29_bcast<2>(alpha, r); q29_bcast<3>(t, r);
        f29_sub(alpha, alpha, t);
        put(o + 184, gamma); put(o + 192, alpha);
    }
    // f29_mul_k against f29_mul<K>
    fe29 a = T.X, b = T.Y, r1, r2;
    f29_mul_k(r1, a, b, 8); f29_mul<8>(r2, a, b);
    put(o + 152, r1); put(o + 160, r2);
    f29_mul_k(r1, a, b, 3); f29_mul<3>(r2, a, b);
    put(o + 168, r1); put(o + 176, r2);
}
int main()
{
    uint32_t *d, *z; static uint32_t h[64 * 344];
    hipMalloc(&d, sizeof h); hipMalloc(&z, 256); hipMemset(z, 0, 256);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d, z);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"2G.X", "2G.Y", "2G.Z", "4G.X", "4G.Y", "4G.Z", "5G.X", "5G.Y", "5G.Z"};
    for (int lane = 0; lane < 8; ++lane) {
        const uint32_t *o = h + lane * 200;
        printf("lane %d:", lane);
        for (int k = 0; k < 9; ++k) {
            bool eq = true;
            for (int i = 0; i < 8; ++i) eq &= o[8 * k + i] == o[72 + 8 * k + i];
            printf(" %s %s", names[k], eq ? "ok" : "DIFF");
        }
        bool e1 = true, e2 = true;
        for (int i = 0; i < 8; ++i) { e1 &= o[152 + i] == o[160 + i]; e2 &= o[168 + i] == o[176 + i]; }
        printf(" mul_k8 %s mul_k3 %s\n", e1 ? "ok" : "DIFF", e2 ? "ok" : "DIFF");
    }
    for (int lane = 0; lane < 4; ++lane) {
        const uint32_t *o = h + lane * 200, *o2 = h + 64 * 200 + lane * 80;
        printf("pair lane %d:", lane);
        for (int k = 0; k < 9; ++k) {
            bool eq = true;
            for (int i = 0; i < 8; ++i) eq &= o[8 * k + i] == o2[8 * k + i];
            printf(" %s %s", names[k], eq ? "ok" : "DIFF");
        }
        printf("\n");
    }
    for (int lane = 0; lane < 2; ++lane) {
        const uint32_t *o3 = h + 64 * 280 + lane * 64;
        const char *nm[] = {"X", "Y", "Z", "D"};
        printf("pair vs quad doubling of 5G, lane %d:", lane);
        for (int k = 0; k < 4; ++k) {
            bool eq = true;
            for (int i = 0; i < 8; ++i) eq &= o3[8 * k + i] == o3[32 + 8 * k + i];
            printf(" %s %s", nm[k], eq ? "ok" : "DIFF");
        }
        printf("\n");
    }
    const uint32_t *o = h;
    auto val = [&](int off) { printf("0x"); for (int i = 7; i >= 0; --i) printf("%08x", o[off + i]); printf("\n"); };
    printf("ref 2G.X "); val(0); printf("ref 2G.Y "); val(8); printf("ref 2G.Z "); val(16);
    { const uint32_t *o2 = h + 64 * 200; printf("pair 2G.Y 0x"); for (int i = 7; i >= 0; --i) printf("%08x", o2[8 + i]); printf("\n"); }
    printf("gamma    "); val(184); printf("alpha    "); val(192);
    printf("q29 2G.X "); val(72); printf("q29 2G.Y "); val(80); printf("q29 2G.Z "); val(88);
    return 0;
}
