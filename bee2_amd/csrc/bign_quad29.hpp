// bign_quad29.hpp -- one signature on FOUR adjacent lanes (a DPP quad) or on a PAIR of lanes, field elements on the
// signed 29-bit limbs of bign_fe29.hpp: the point operations of the small verification batches (quads up to 2^14
// signatures, pairs up to 2^15; bign_quad29_kernel in bign_kernels.hip).
//
// v Q is 32N dependent doublings; with one lane per signature a batch below 2^16 signatures is one wavefront per
// SIMD walking that chain, so its time does not depend on its size (the latency floor).  A doubling's
// multiplications have dependency depth 3, a general addition's depth 4: lane k of a quad takes the k-th
// multiplication of a level, the products travel with v_mov_b32 quad_perm broadcasts, the limb-wise additions are
// evaluated by all four lanes, so the state (X, Y, Z, Z^2) stays replicated.  The spare lanes buy the formulas'
// small multiples as separately scaled products (4 X g AND 8 X g, 3 X^2 AND 3 Z^4), which also keeps every sum
// inside the bounds of the carry-free limbs: a doubling needs no carry pass at all, an addition one.
// Same group law as jac_dbl / jac_add (bign_dev.hpp), hence the same exceptional cases; each of them zeroes Z3 and
// every Z after it, and the caller tests the final Z once.
// tools/ubench/quad_dbl.hip on a lone wavefront: quads x2.5, pairs x1.9, quads on 32-bit limbs (tools/ubench/bign_quad32.hpp) x1.6.
#pragma once
#include "bign_fe29.hpp"

namespace bee2hip {

// a from lane K of the quad.  ONE asm block per field element, on purpose:
//  * with __builtin_amdgcn_mov_dpp LLVM folds the broadcast into the consuming v_sub_u32 (v_subrev_u32_dpp ... quad_perm)
//    and the differences of two broadcasts came out wrong on gfx950 (tools/ubench/quad29_check.hip);
//  * with one asm statement per limb the scheduler is free to place the VALU instruction that produces limb i right
//    in front of the v_mov_b32_dpp that reads it, and the hazard "VALU writes a VGPR -> DPP reads it within two
//    instructions" is nobody's job inside inline asm (LLVM's hazard recogniser does not look into it): results then
//    depended on the schedule (the same source was right in the kernel and wrong in the micro-benchmark).
// A block of nine moves behind one `s_nop 1` cannot be split, and its outputs are early-clobber because the inputs
// are still being read while the first outputs are written.
#define Q29_BLK7(R, A, O, CTRL) \
    asm volatile("s_nop 1\n\t" "v_mov_b32_dpp %0, %7 " CTRL "\n\t" "v_mov_b32_dpp %1, %8 " CTRL "\n\t" "v_mov_b32_dpp %2, %9 " CTRL "\n\t" "v_mov_b32_dpp %3, %10 " CTRL "\n\t" "v_mov_b32_dpp %4, %11 " CTRL "\n\t" "v_mov_b32_dpp %5, %12 " CTRL "\n\t" "v_mov_b32_dpp %6, %13 " CTRL "\n\t" "" : "=&v"((R)[(O) + 0]), "=&v"((R)[(O) + 1]), "=&v"((R)[(O) + 2]), "=&v"((R)[(O) + 3]), "=&v"((R)[(O) + 4]), "=&v"((R)[(O) + 5]), "=&v"((R)[(O) + 6]) : "v"((A)[(O) + 0]), "v"((A)[(O) + 1]), "v"((A)[(O) + 2]), "v"((A)[(O) + 3]), "v"((A)[(O) + 4]), "v"((A)[(O) + 5]), "v"((A)[(O) + 6]))
#define Q29_BLK9(R, A, O, CTRL) \
    asm volatile("s_nop 1\n\t" "v_mov_b32_dpp %0, %9 " CTRL "\n\t" "v_mov_b32_dpp %1, %10 " CTRL "\n\t" "v_mov_b32_dpp %2, %11 " CTRL "\n\t" "v_mov_b32_dpp %3, %12 " CTRL "\n\t" "v_mov_b32_dpp %4, %13 " CTRL "\n\t" "v_mov_b32_dpp %5, %14 " CTRL "\n\t" "v_mov_b32_dpp %6, %15 " CTRL "\n\t" "v_mov_b32_dpp %7, %16 " CTRL "\n\t" "v_mov_b32_dpp %8, %17 " CTRL "\n\t" "" : "=&v"((R)[(O) + 0]), "=&v"((R)[(O) + 1]), "=&v"((R)[(O) + 2]), "=&v"((R)[(O) + 3]), "=&v"((R)[(O) + 4]), "=&v"((R)[(O) + 5]), "=&v"((R)[(O) + 6]), "=&v"((R)[(O) + 7]), "=&v"((R)[(O) + 8]) : "v"((A)[(O) + 0]), "v"((A)[(O) + 1]), "v"((A)[(O) + 2]), "v"((A)[(O) + 3]), "v"((A)[(O) + 4]), "v"((A)[(O) + 5]), "v"((A)[(O) + 6]), "v"((A)[(O) + 7]), "v"((A)[(O) + 8]))
#define Q29_BLK10(R, A, O, CTRL) \
    asm volatile("s_nop 1\n\t" "v_mov_b32_dpp %0, %10 " CTRL "\n\t" "v_mov_b32_dpp %1, %11 " CTRL "\n\t" "v_mov_b32_dpp %2, %12 " CTRL "\n\t" "v_mov_b32_dpp %3, %13 " CTRL "\n\t" "v_mov_b32_dpp %4, %14 " CTRL "\n\t" "v_mov_b32_dpp %5, %15 " CTRL "\n\t" "v_mov_b32_dpp %6, %16 " CTRL "\n\t" "v_mov_b32_dpp %7, %17 " CTRL "\n\t" "v_mov_b32_dpp %8, %18 " CTRL "\n\t" "v_mov_b32_dpp %9, %19 " CTRL "\n\t" "" : "=&v"((R)[(O) + 0]), "=&v"((R)[(O) + 1]), "=&v"((R)[(O) + 2]), "=&v"((R)[(O) + 3]), "=&v"((R)[(O) + 4]), "=&v"((R)[(O) + 5]), "=&v"((R)[(O) + 6]), "=&v"((R)[(O) + 7]), "=&v"((R)[(O) + 8]), "=&v"((R)[(O) + 9]) : "v"((A)[(O) + 0]), "v"((A)[(O) + 1]), "v"((A)[(O) + 2]), "v"((A)[(O) + 3]), "v"((A)[(O) + 4]), "v"((A)[(O) + 5]), "v"((A)[(O) + 6]), "v"((A)[(O) + 7]), "v"((A)[(O) + 8]), "v"((A)[(O) + 9]))
// L limbs through blocks of at most ten moves (an asm statement takes at most 30 operands): 9 = 9, 14 = 7 + 7, 19 = 10 + 9
#define Q29_DPP_ALL(CTRL)                                                                       \
    do {                                                                                        \
        if constexpr (LZ<N>::L == 9) { Q29_BLK9(r.l, a.l, 0, CTRL); }                           \
        else if constexpr (LZ<N>::L == 14) { Q29_BLK7(r.l, a.l, 0, CTRL); Q29_BLK7(r.l, a.l, 7, CTRL); } \
        else { Q29_BLK10(r.l, a.l, 0, CTRL); Q29_BLK9(r.l, a.l, 10, CTRL); }                    \
    } while (0)
template <int K, int N>
__device__ __forceinline__ void q29_bcast(lzT<N> &r, const lzT<N> &a)
{
    static_assert(K >= 0 && K < 4, "lane of the quad");
    static_assert(LZ<N>::L == 9 || LZ<N>::L == 14 || LZ<N>::L == 19, "limb counts the blocks are written for");
    if constexpr (K == 0) Q29_DPP_ALL("quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1");
    else if constexpr (K == 1) Q29_DPP_ALL("quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1");
    else if constexpr (K == 2) Q29_DPP_ALL("quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1");
    else Q29_DPP_ALL("quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1");
}
// a from the lane four places up (the helper quad of an 8-lane group hands its point to the main quad)
template <int N>
__device__ __forceinline__ void q29_from_next_quad(lzT<N> &r, const lzT<N> &a)
{
    Q29_DPP_ALL("row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
}
template <int N>
__device__ __forceinline__ void q29_pick(lzT<N> &r, bool p, const lzT<N> &a, const lzT<N> &b)
{
#pragma unroll
    for (int i = 0; i < LZ<N>::L; ++i) r.l[i] = p ? a.l[i] : b.l[i];
}

template <int N> struct lqjacT { lzT<N> X, Y, Z, D; };    // D = Z^2; X, Y: L1; Z, D: N; replicated in the quad
template <int N> struct lqentT { lzT<N> X, Y, Z, ZZ; };   // X, Y: L1; Z, ZZ = Z^2: N (affine: 1, 1)
typedef lqjacT<8> qjac29;
typedef lqentT<8> qent29;

// T <- 2T, a = -3, three levels.  Bounds in units of u = 2^29 (operand bounds of a product must multiply to <= 3):
//   A: Y^2 | 2 Y Z | 3 X^2 | 3 D^2        (1 x 1 each)      alpha = 3 X^2 - 3 Z^4                      L1
//   B: 4 X g | 8 X g | Z3^2 | alpha^2      (1 x 1 each)      X3 = alpha^2 - 8 X g                       L1
//   C: alpha (4 X g - X3) | 8 g^2          (1 x 2, 1 x 1)    Y3 = ... - 8 g^2                           L1
template <int N>
__device__ __forceinline__ void quad29_dbl(lqjacT<N> &T, uint32_t q)
{
    const bool q0 = q == 0, q1 = q == 1, q2 = q == 2, lo = q < 2;
    lzT<N> a, b, r, gamma, alpha, b4, t;
    // level A
    q29_pick(a, q2, T.X, T.D);
    q29_pick(a, lo, T.Y, a);
    q29_pick(b, q1, T.Z, a);
    f29_mul_k(r, a, b, q0 ? 1 : q1 ? 2 : 3);
    q29_bcast<0>(gamma, r);
    q29_bcast<1>(T.Z, r);                           // Z3 = 2 Y Z
    q29_bcast<2>(alpha, r);
    q29_bcast<3>(t, r);
    f29_sub(alpha, alpha, t);
    // level B
    q29_pick(a, q2, T.Z, alpha);
    q29_pick(a, lo, T.X, a);
    q29_pick(b, lo, gamma, a);
    f29_mul_k(r, a, b, q0 ? 4 : q1 ? 8 : 1);
    q29_bcast<0>(b4, r);
    q29_bcast<1>(t, r);                             // 8 X g
    q29_bcast<2>(T.D, r);                           // D3 = Z3^2
    q29_bcast<3>(T.X, r);                           // alpha^2
    f29_sub(T.X, T.X, t);                           // X3
    f29_sub(t, b4, T.X);                            // [-1, 2]
    // level C
    q29_pick(a, q0, alpha, gamma);
    q29_pick(b, q0, t, gamma);
    f29_mul_k(r, a, b, q1 ? 8 : 1);
    q29_bcast<0>(T.Y, r);
    q29_bcast<1>(t, r);
    f29_sub(T.Y, T.Y, t);                           // Y3
}

// T <- T + E, general addition (add-1998-cmo-2 with Z1^2 carried and Z2^2 tabulated) in four levels:
//   1: X1 ZZ2 | X2 D | Z1 D | Y1 ZZ2             H = U2 - U1 (L1)
//   2: Y2 Z1^3 | H^2 | Z1 Z2 | (Y1 ZZ2) Z2       r = S2 - S1 (L1)
//   3: H H^2 | U1 H^2 | (Z1 Z2) H | r^2          X3 = r^2 - H^3 - 2V  ([-3, 1] -> carry -> N)
//   4: r (V - X3) | S1 H^3 | Z3^2                Y3 (L1), D3
template <int N>
__device__ __forceinline__ void quad29_add(lqjacT<N> &T, const lqentT<N> &E, uint32_t q)
{
    const bool q0 = q == 0, q1 = q == 1, q2 = q == 2;
    lzT<N> a, b, r, U1, S1, H, HH, rr, V, t;
    // level 1
    q29_pick(a, q2, T.Z, T.Y);
    q29_pick(a, q1, E.X, a);
    q29_pick(a, q0, T.X, a);
    q29_pick(b, q1 || q2, T.D, E.ZZ);
    f29_mul(r, a, b);
    q29_bcast<0>(U1, r);
    q29_bcast<1>(H, r);
    q29_bcast<2>(t, r);                             // Z1^3
    f29_sub(H, H, U1);
    // level 2 (lane 3 multiplies its own Y1 ZZ2 by Z2)
    q29_pick(a, q0, E.Y, T.Z);
    q29_pick(a, q1, H, a);
    q29_pick(a, q == 3, r, a);
    q29_pick(b, q0, t, E.Z);
    q29_pick(b, q1, H, b);
    f29_mul(r, a, b);
    q29_bcast<0>(rr, r);
    q29_bcast<1>(HH, r);
    q29_bcast<2>(t, r);                             // Z1 Z2
    q29_bcast<3>(S1, r);
    f29_sub(rr, rr, S1);
    // level 3
    q29_pick(a, q2, t, rr);
    q29_pick(a, q1, U1, a);
    q29_pick(a, q0, H, a);
    q29_pick(b, q2, H, rr);
    q29_pick(b, q0 || q1, HH, b);
    f29_mul(r, a, b);
    q29_bcast<0>(H, r);                             // H^3
    q29_bcast<1>(V, r);
    q29_bcast<2>(T.Z, r);                           // Z3
    q29_bcast<3>(t, r);                             // r^2
    f29_sub(t, t, H);
    f29_sub(t, t, V);
    f29_sub(T.X, t, V);
    f29_carry(T.X);                                 // X3, N
    f29_sub(t, V, T.X);
    // level 4
    q29_pick(a, q0, rr, S1);
    q29_pick(a, q2, T.Z, a);
    q29_pick(b, q0, t, H);
    q29_pick(b, q2, T.Z, b);
    f29_mul(r, a, b);
    q29_bcast<0>(t, r);
    q29_bcast<1>(V, r);
    q29_bcast<2>(T.D, r);                           // D3
    f29_sub(T.Y, t, V);                             // Y3
}

// ------------------------------------------------------------------ two lanes per signature ---
// Between 2^14 and 2^15 signatures quads would already share SIMDs (two wavefronts each); a PAIR of lanes per
// signature keeps one wavefront per SIMD there.  Same formulas and bounds, two multiplications per level: a doubling
// is five levels (two of them pure squarings), a general addition eight.
template <int K, int N>
__device__ __forceinline__ void p29_bcast(lzT<N> &r, const lzT<N> &a)      // a from lane K of the pair
{
    static_assert(K == 0 || K == 1, "lane of the pair");
    if constexpr (K == 0) Q29_DPP_ALL("quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1");
    else Q29_DPP_ALL("quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1");
}

//   1: Y^2 | 2 Y Z      2: 3 X^2 | 3 D^2      3: 4 X g | 8 X g      4: alpha^2 | Z3^2      5: alpha (4 X g - X3) | 8 g^2
template <int N>
__device__ __forceinline__ void pair29_dbl(lqjacT<N> &T, uint32_t p)
{
    const bool p0 = p == 0;
    lzT<N> a, b, r, gamma, alpha, b4, t;
    q29_pick(b, p0, T.Y, T.Z);
    f29_mul_k(r, T.Y, b, p0 ? 1 : 2);
    p29_bcast<0>(gamma, r);
    q29_pick(a, p0, T.X, T.D);
    p29_bcast<1>(T.Z, r);                           // Z3 = 2 Y Z (after T.Z was read)
    f29_sqr<3>(r, a);
    p29_bcast<0>(alpha, r);
    p29_bcast<1>(t, r);
    f29_sub(alpha, alpha, t);                       // 3 X^2 - 3 Z^4
    f29_mul_k(r, T.X, gamma, p0 ? 4 : 8);
    p29_bcast<0>(b4, r);
    p29_bcast<1>(t, r);                             // 8 X g
    q29_pick(a, p0, alpha, T.Z);
    f29_sqr(r, a);
    p29_bcast<0>(T.X, r);
    p29_bcast<1>(T.D, r);                           // D3 = Z3^2
    f29_sub(T.X, T.X, t);                           // X3
    f29_sub(t, b4, T.X);                            // [-1, 2]
    q29_pick(a, p0, alpha, gamma);
    q29_pick(b, p0, t, gamma);
    f29_mul_k(r, a, b, p0 ? 1 : 8);
    p29_bcast<0>(T.Y, r);
    p29_bcast<1>(t, r);
    f29_sub(T.Y, T.Y, t);                           // Y3
}

//   1: X1 ZZ2 | X2 D      2: Z1 D | Y1 ZZ2      3: Y2 Z1^3 | (Y1 ZZ2) Z2      4: H^2 | Z1 Z2
//   5: H H^2 | U1 H^2     6: r^2 | (Z1 Z2) H    7: r (V - X3) | S1 H^3        8: Z3^2 (both lanes)
template <int N>
__device__ __forceinline__ void pair29_add(lqjacT<N> &T, const lqentT<N> &E, uint32_t p)
{
    const bool p0 = p == 0;
    lzT<N> a, b, r, U1, S1, H, HH, rr, V, t, w;
    q29_pick(a, p0, T.X, E.X);
    q29_pick(b, p0, E.ZZ, T.D);
    f29_mul(r, a, b);
    p29_bcast<0>(U1, r);
    p29_bcast<1>(H, r);
    f29_sub(H, H, U1);                              // U2 - U1
    q29_pick(a, p0, T.Z, T.Y);
    q29_pick(b, p0, T.D, E.ZZ);
    f29_mul(r, a, b);                               // Z1^3 | Y1 ZZ2
    q29_pick(a, p0, E.Y, r);
    p29_bcast<0>(t, r);
    q29_pick(b, p0, t, E.Z);
    f29_mul(r, a, b);
    p29_bcast<0>(rr, r);
    p29_bcast<1>(S1, r);
    f29_sub(rr, rr, S1);                            // r = S2 - S1
    q29_pick(a, p0, H, T.Z);
    q29_pick(b, p0, H, E.Z);
    f29_mul(r, a, b);
    p29_bcast<0>(HH, r);
    p29_bcast<1>(w, r);                             // Z1 Z2
    q29_pick(a, p0, H, U1);
    f29_mul(r, a, HH);
    p29_bcast<0>(t, r);                             // H^3
    p29_bcast<1>(V, r);
    q29_pick(a, p0, rr, w);
    q29_pick(b, p0, rr, H);
    f29_mul(r, a, b);
    p29_bcast<1>(T.Z, r);                           // Z3
    p29_bcast<0>(a, r);                             // r^2
    f29_sub(a, a, t);
    f29_sub(a, a, V);
    f29_sub(T.X, a, V);
    f29_carry(T.X);                                 // X3, N
    f29_sub(b, V, T.X);
    q29_pick(a, p0, rr, S1);
    q29_pick(b, p0, b, t);
    f29_mul(r, a, b);
    p29_bcast<0>(a, r);
    p29_bcast<1>(b, r);
    f29_sub(T.Y, a, b);                             // Y3
    f29_sqr(T.D, T.Z);                              // D3, the same in both lanes
}

}  // namespace bee2hip
