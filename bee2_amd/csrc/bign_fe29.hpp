// bign_fe29.hpp -- GF(2^256 - 189) on nine SIGNED 29-bit limbs, for wavefronts that are alone on their SIMD.
//
// Why a second field representation (round 2, DESIGN.md 4.3 "small batches"): a wavefront that has its SIMD to
// itself issues one instruction every ~5 cycles WHATEVER the instruction is (tools/ubench/valu_rates.hip section w),
// so below ~2^16 signatures the time of a verification batch is its instruction COUNT, not its multiplier work.  The
// product-scanning 8 x 32-bit multiplication of bign_dev.hpp is 64 v_mad_u64_u32 + 64 v_addc_co_u32 + 64 s_nop
// (SGPR-carry wait state) + reduction = 232 instructions; with 29-bit limbs a column of nine products fits a 64-bit
// accumulator without carries: 81 + 9 multiply-adds + ~70 full-rate shifts / masks = 160, a squaring 123 against
// 180, an addition 9 against 20.  tools/ubench/fe29.hip: x1.40-1.42 per multiplication at <= 1 wavefront per SIMD,
// x0.98-0.99 at 2 and 4 (there the half-rate instructions decide and the two forms tie) -- so this form serves the
// small batches only (bign_main29_kernel) and the 32-bit form stays the throughput path.
//
// Representation: value = sum l[i] 2^(29 i), i = 0..8, limbs int32, any value (positive or negative) congruent to
// the residue.  2^261 = 2^5 * 2^256 = 6048 (mod p).  "u" below = 2^29.
//   N  (normalised) : output of f29_mul / f29_sqr / f29_carry: l[2..8] in [0, u), l[1] within 2 and l[0] within 2^16 of [0, u)
//   L1 (lazy)       : |l[i]| <= u + 2^16: a difference of two N values, or the negation of one
// Additions and subtractions are limb-wise with no carries at all; a multiplication accepts operands whose limb
// bounds A u and B u satisfy A * B <= 3 (nine products of 2^58 A B plus the carry-in stay below 2^63); the point
// formulas below place a carry pass (f29_carry, 29 instructions) exactly where a value would exceed that
// (tools/fe29_bounds.py propagates the intervals through both formulas and checks every accumulator).
#pragma once
#include "bign_dev.hpp"

namespace bee2hip {

struct fe29 { int32_t l[9]; };
constexpr int32_t F29_M = (1 << 29) - 1;
constexpr int32_t F29_FOLD = 189 * 32;                  // 2^261 mod p

// 8 x 32-bit words (any value < 2^256) -> N
__device__ __forceinline__ void f29_from_words(fe29 &r, const feT<8> &a)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        const uint32_t lo = a.v[w], hi = w + 1 < 8 ? a.v[w + 1] : 0u;
        const uint32_t x = sh ? __builtin_amdgcn_alignbit(hi, lo, sh) : lo;
        r.l[i] = (int32_t)(x & (uint32_t)F29_M);
    }
}

// one floor-carry pass, the carry out of limb 8 folded back into limb 0: |l[i]| < 4 u in, N out
__device__ __forceinline__ void f29_carry(fe29 &a)
{
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t t = a.l[i] + c;
        a.l[i] = t & F29_M;
        c = t >> 29;
    }
    a.l[0] += c * F29_FOLD;
}

__device__ __forceinline__ void f29_add(fe29 &r, const fe29 &a, const fe29 &b)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
}
__device__ __forceinline__ void f29_sub(fe29 &r, const fe29 &a, const fe29 &b)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i];
}
__device__ __forceinline__ void f29_neg(fe29 &r, const fe29 &a)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = -a.l[i];
}

// c[0..17] (c[k] in [0, u) for k < 17, c[17] signed) -> r = K (lo + 2^261 hi), N
template <int K>
__device__ __forceinline__ void f29_fold(fe29 &r, const int32_t (&c)[18])
{
    int64_t cy = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        int64_t t = (int64_t)c[9 + j] * (int64_t)(F29_FOLD * K) + cy;
        if (K == 1) t += c[j];
        else t += (int64_t)c[j] * K;
        r.l[j] = (int32_t)t & F29_M;
        cy = t >> 29;
    }
    // |cy| < 2^17: its weight is 2^261 again
    const int32_t t0 = r.l[0] + (int32_t)cy * F29_FOLD;
    r.l[0] = t0 & F29_M;
    r.l[1] += t0 >> 29;
}

template <int K = 1>
__device__ __forceinline__ void f29_mul(fe29 &r, const fe29 &a, const fe29 &b)
{
    int32_t c[18];
    int64_t acc = 0;
    static_for<0, 17>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        static_for<(k > 8 ? k - 8 : 0), (k < 8 ? k : 8) + 1>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            acc += (int64_t)a.l[i] * b.l[k - i];
        });
        c[k] = (int32_t)acc & F29_M;
        acc >>= 29;
    });
    c[17] = (int32_t)acc;
    f29_fold<K>(r, c);
}

template <int K = 1>
__device__ __forceinline__ void f29_sqr(fe29 &r, const fe29 &a)
{
    int32_t c[18], d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.l[i] * 2;
    int64_t acc = 0;
    static_for<0, 17>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        static_for<(k > 8 ? k - 8 : 0), (k < 8 ? k : 8) + 1>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int j = k - i;
            if constexpr (i < j) acc += (int64_t)a.l[i] * d[j];
            else if constexpr (i == j) acc += (int64_t)a.l[i] * a.l[i];
        });
        c[k] = (int32_t)acc & F29_M;
        acc >>= 29;
    });
    c[17] = (int32_t)acc;
    f29_fold<K>(r, c);
}

// any |l[i]| < 4 u -> 8 x 32-bit words, weakly reduced (a value in [0, 2^256) congruent to the residue), exactly
__device__ __forceinline__ void f29_to_words(feT<8> &r, fe29 a)
{
    f29_carry(a);                                   // N: value in (-2^16, 2^261 + 2^17)
    // + p (limbs of 2^256 - 189) makes the value positive; carry without wrap (limb 8 keeps what is above 2^232)
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t pi = i == 0 ? F29_M + 1 - 189 : i < 8 ? F29_M : (1 << 24) - 1;
        const int32_t t = a.l[i] + pi + c;
        if (i < 8) { a.l[i] = t & F29_M; c = t >> 29; }
        else a.l[i] = t;                            // in [0, 2^29 + 2^24]
    }
    // fold the bits from 2^256 up (limb 8 holds bits 232..): top < 64
    int32_t top = a.l[8] >> 24;
    a.l[8] &= (1 << 24) - 1;
    c = top * 189;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t t = a.l[i] + c;
        if (i < 8) { a.l[i] = t & F29_M; c = t >> 29; }
        else a.l[i] = t;
    }
    // a second wrap leaves a value below 64 * 189 + 189: no further carry
    top = a.l[8] >> 24;
    a.l[8] &= (1 << 24) - 1;
    a.l[0] += top * 189;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int lo = 32 * w / 29, sh = 32 * w % 29;
        uint64_t v = (uint64_t)(uint32_t)a.l[lo] >> sh;
        v |= (uint64_t)(uint32_t)a.l[lo + 1] << (29 - sh);
        if (lo + 2 < 9 && 58 - sh < 32) v |= (uint64_t)(uint32_t)a.l[lo + 2] << (58 - sh);
        r.v[w] = (uint32_t)v;
    }
}

struct jac29 { fe29 X, Y, Z; };                    // X, Z: N; Y: L1
struct aff29 { fe29 x, y; };                       // x: N; y: N or L1 (negated table entry)

// T <- 2T (jac_dbl of bign_dev.hpp, same formulas).  Bounds in units of u:
__device__ __forceinline__ void jac29_dbl(jac29 &T)
{
    fe29 delta, gamma, beta4, alpha, t0, t1;
    f29_sqr(delta, T.Z);                            // N
    f29_sqr(gamma, T.Y);                            // 1 x 1
    f29_mul<4>(beta4, T.X, gamma);                  // 4 X Y^2
    f29_sub(t0, T.X, delta);                        // [-1, 1]
    f29_add(t1, T.X, delta);                        // [0, 2]
    f29_mul<3>(alpha, t0, t1);                      // 1 x 2
    f29_mul<2>(T.Z, T.Y, T.Z);                      // Z3 = 2 Y Z
    f29_sqr(t0, alpha);
    f29_add(t1, beta4, beta4);
    f29_sub(T.X, t0, t1);                           // [-2, 1]
    f29_carry(T.X);                                 // X3 = alpha^2 - 8 beta, N
    f29_sqr<8>(t1, gamma);                          // 8 Y^4
    f29_sub(t0, beta4, T.X);                        // [-1, 1]
    f29_mul(t0, alpha, t0);
    f29_sub(T.Y, t0, t1);                           // Y3, L1
}

// T <- T + E, E affine (jac_madd).  Exceptional cases (T = O, T = +-E) are NOT flagged here: each of them makes
// Z3 = Z1 H = 0, every later Z is a multiple of it, and the caller tests the final Z once.
__device__ __forceinline__ void jac29_madd(jac29 &T, const aff29 &E)
{
    fe29 Z1Z1, U2, S2, H, HH, HHH, r, V, t;
    f29_sqr(Z1Z1, T.Z);
    f29_mul(U2, E.x, Z1Z1);
    f29_mul(t, T.Z, Z1Z1);
    f29_mul(S2, E.y, t);                            // 1 x 1
    f29_sub(H, U2, T.X);                            // [-1, 1]
    f29_sub(r, S2, T.Y);                            // [-1, 2]
    f29_carry(r);                                   // N
    f29_sqr(HH, H);
    f29_mul(HHH, H, HH);
    f29_mul(V, T.X, HH);
    f29_mul(T.Z, T.Z, H);                           // Z3 = Z1 H
    f29_sqr(t, r);
    f29_sub(t, t, HHH);
    f29_sub(t, t, V);
    f29_sub(T.X, t, V);                             // [-3, 1]
    f29_carry(T.X);                                 // X3 = r^2 - H^3 - 2V, N
    f29_sub(t, V, T.X);                             // [-1, 1]
    f29_mul(t, r, t);
    f29_mul(S2, T.Y, HHH);                          // 1 x 1
    f29_sub(T.Y, t, S2);                            // Y3, L1
}

}  // namespace bee2hip
