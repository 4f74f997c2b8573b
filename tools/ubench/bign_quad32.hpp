// bign_quad32.hpp -- one signature on FOUR adjacent lanes (a DPP quad): the point operations of bign_dev.hpp with the
// field multiplications of one dependency level spread over the lanes of the quad.
//
// Why (DESIGN.md 4.3 "latency floor"): v Q needs 32N dependent doublings, and with one lane per
// signature a batch of <= 2^16 signatures is a single wavefront per SIMD walking that chain -- the time of a batch
// does not depend on its size below that.  The doubling's 8 multiplications have dependency depth 3, the addition's
// 16 depth 4; a quad runs each level's multiplications side by side (lane k of the quad = role k), exchanges the
// products with v_mov_b32 quad_perm broadcasts and evaluates the cheap additions redundantly in all four lanes, so
// the state (X, Y, Z, Z^2) stays replicated.  Same formulas as jac_dbl / jac_add (dbl-2001-b, add-1998-cmo-2), i.e.
// the same group elements and the same exceptional cases, which go to bign_slow_kernel as before.
//
// MEASUREMENT ONLY (tools/ubench/quad_dbl.hip: x1.6 on a lone wavefront): the kernels use the 29-bit-limb form of the
// same idea, bign_quad29.hpp (x2.5), and its single-block asm broadcasts.  The __builtin_amdgcn_mov_dpp broadcasts
// below are safe HERE only because every consumer is a carry chain (v_sub_co / v_subb), which LLVM does not fold a
// DPP operand into -- see the note in bign_quad29.hpp before reusing them.
#pragma once
#include "bign_dev.hpp"

namespace bee2hip {

// a from lane K of the quad
template <int K, int N>
__device__ __forceinline__ void quad_bcast(feT<N> &r, const feT<N> &a)
{
#pragma unroll
    for (int i = 0; i < N; ++i)
        r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], K * 0x55, 0xF, 0xF, true);
}
template <int N>
__device__ __forceinline__ void fe_pick(feT<N> &r, bool p, const feT<N> &a, const feT<N> &b)
{
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = p ? a.v[i] : b.v[i];
}

// r = K * a * b with a per-lane K (1, 2, 3, 4, 8): fe_reduce with the two scale factors in registers
template <int N>
__device__ __forceinline__ void fe_mul_k(feT<N> &r, const feT<N> &a, const feT<N> &b, uint32_t K)
{
    constexpr uint32_t C = CurveC<N>::C;
    uint32_t w[2 * N];
    uint64_t acc = 0;
    uint32_t c2 = 0;
    static_for<0, 2 * N - 1>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;
        static_for<i0, (k < N ? k : N - 1) + 1>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            mac_col<i == i0>(acc, c2, a.v[i], b.v[k - i]);
        });
        w[k] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32);
    });
    w[2 * N - 1] = (uint32_t)acc;
    const uint32_t KC = K * C;
    uint32_t t[N];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t s = (uint64_t)w[N + i] * KC + c;
        s += (uint64_t)w[i] * K;
        t[i] = (uint32_t)s; c = s >> 32;
    }
    const uint32_t cc = (uint32_t)c * C;
    uint64_t cy;
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(r.v[0]), "=s"(cy) : "v"(t[0]), "v"(cc));
#pragma unroll
    for (int i = 1; i < N; ++i)
        asm("v_addc_co_u32 %0, %1, 0, %2, %1" : "=v"(r.v[i]), "+s"(cy) : "v"(t[i]));
    uint32_t top;
    asm("v_addc_co_u32 %0, %1, 0, 0, %1" : "=v"(top), "+s"(cy));
    r.v[0] += (0u - top) & C;
}

// (X : Y : Z) with D = Z^2 carried along, replicated in the four lanes of the quad
template <int N> struct qjacT { feT<N> X, Y, Z, D; };
// table entry for the general addition: Jacobian point with Z^2 and Z^3 (affine: Z = ZZ = ZZZ = 1)
template <int N> struct qentT { feT<N> X, Y, Z, ZZ, ZZZ; };

// T <- 2T, a = -3: three levels of multiplications
//   A: Y^2 | 2YZ | 3(X-D)(X+D)      B: 4X g | 8X g | Z3^2 | alpha^2      C: alpha (4b - X3) | 8 g^2
template <int N>
__device__ __forceinline__ void quad_dbl(qjacT<N> &T, uint32_t q)
{
    const bool q0 = q == 0, q1 = q == 1, q2 = q == 2;
    feT<N> a, b, r, t2, t3, gamma, alpha, b4, b8, A2;
    fe_sub(t2, T.X, T.D);
    fe_add(t3, T.X, T.D);
    // level A
    fe_pick(a, q2, t2, T.Y);
    fe_pick(b, q2, t3, T.Z);
    fe_pick(b, q0, T.Y, b);
    fe_mul_k(r, a, b, q0 ? 1u : q1 ? 2u : 3u);
    quad_bcast<0>(gamma, r);
    quad_bcast<1>(T.Z, r);                  // Z3 = 2 Y Z
    quad_bcast<2>(alpha, r);
    // level B
    fe_pick(a, q2, T.Z, alpha);             // q2: Z3, q3: alpha
    fe_pick(a, q0 || q1, T.X, a);
    fe_pick(b, q0 || q1, gamma, a);
    fe_mul_k(r, a, b, q0 ? 4u : q1 ? 8u : 1u);
    quad_bcast<0>(b4, r);
    quad_bcast<1>(b8, r);
    quad_bcast<2>(T.D, r);                  // D3 = Z3^2
    quad_bcast<3>(A2, r);
    fe_sub(T.X, A2, b8);                    // X3 = alpha^2 - 8 beta
    fe_sub(t2, b4, T.X);
    // level C
    fe_pick(a, q0, alpha, gamma);
    fe_pick(b, q0, t2, gamma);
    fe_mul_k(r, a, b, q0 ? 1u : 8u);
    quad_bcast<0>(t2, r);
    quad_bcast<1>(t3, r);
    fe_sub(T.Y, t2, t3);                    // Y3 = alpha (4 beta - X3) - 8 Y^4
}

// T <- T + E (general addition, 12M + 4S + Z3^2 in four levels); false when the generic formula does not apply
//   1: X1 ZZ2 | X2 D | Z1 D | Y1 ZZZ2    2: Y2 t | H^2 | Z1 Z2    3: H HH | U1 HH | ZZ H | r^2    4: r (V - X3) | S1 H^3 | Z3^2
template <int N>
__device__ __forceinline__ bool quad_add(qjacT<N> &T, const qentT<N> &E, uint32_t q)
{
    const bool q0 = q == 0, q1 = q == 1, q2 = q == 2;
    const bool bad_in = fe_is_zero(T.Z) || fe_is_zero(E.Z);
    feT<N> a, b, r, U1, S1, H, HH, rr, V, t;
    // level 1
    fe_pick(a, q2, T.Z, T.Y);
    fe_pick(a, q1, E.X, a);
    fe_pick(a, q0, T.X, a);
    fe_pick(b, q0, E.ZZ, E.ZZZ);
    fe_pick(b, q1 || q2, T.D, b);
    fe_mul_k(r, a, b, 1u);
    quad_bcast<0>(U1, r);
    quad_bcast<1>(H, r);                    // U2
    quad_bcast<2>(t, r);                    // Z1^3
    quad_bcast<3>(S1, r);
    fe_sub(H, H, U1);
    const bool bad = bad_in || fe_is_zero(H);
    // level 2
    fe_pick(a, q0, E.Y, T.Z);
    fe_pick(a, q1, H, a);
    fe_pick(b, q0, t, E.Z);
    fe_pick(b, q1, H, b);
    fe_mul_k(r, a, b, 1u);
    quad_bcast<0>(rr, r);                   // S2
    quad_bcast<1>(HH, r);
    quad_bcast<2>(t, r);                    // Z1 Z2
    fe_sub(rr, rr, S1);                     // r = S2 - S1
    // level 3
    fe_pick(a, q2, t, rr);
    fe_pick(a, q1, U1, a);
    fe_pick(a, q0, H, a);
    fe_pick(b, q2, H, rr);
    fe_pick(b, q0 || q1, HH, b);
    fe_mul_k(r, a, b, 1u);
    quad_bcast<0>(H, r);                    // H^3
    quad_bcast<1>(V, r);
    quad_bcast<2>(T.Z, r);                  // Z3 = Z1 Z2 H
    quad_bcast<3>(t, r);                    // r^2
    fe_sub(t, t, H);
    fe_sub(t, t, V);
    fe_sub(T.X, t, V);                      // X3 = r^2 - H^3 - 2V
    fe_sub(t, V, T.X);
    // level 4
    fe_pick(a, q0, rr, S1);
    fe_pick(a, q2, T.Z, a);
    fe_pick(b, q0, t, H);
    fe_pick(b, q2, T.Z, b);
    fe_mul_k(r, a, b, 1u);
    quad_bcast<0>(t, r);
    quad_bcast<1>(V, r);
    quad_bcast<2>(T.D, r);                  // D3
    fe_sub(T.Y, t, V);                      // Y3 = r (V - X3) - S1 H^3
    return !bad;
}

}  // namespace bee2hip
