// bign_fe29.hpp -- GF(2^256 - 189) on nine SIGNED 29-bit limbs, for wavefronts that are alone on their SIMD.
//
// Why a second field representation (round 2): a wavefront that has its SIMD to itself issues one instruction every
// ~5 cycles WHATEVER the instruction is, so below ~2^16 signatures the time of a verification batch is its instruction COUNT.  The
// product-scanning 8 x 32-bit multiplication of bign_dev.hpp is 64 v_mad_u64_u32 + 64 v_addc_co_u32 + 64 s_nop (SGPR-carry wait
// state) + reduction = 232 instructions; with 29-bit limbs a column of nine products fits a 64-bit accumulator without carry flags.
// Round 6 made the multiplication ONE inline-asm block (bign_fe29_asm.inc: 135 instructions, 91 of them v_mad_i64_i32; a squaring 99;
// an addition 9 against 20): at that count the form also wins at four wavefronts per SIMD -- 91 multiply-adds and no v_addc against
// 64 + 64 -- and carries every 256-bit batch up to 2^18 signatures (bign_main29_kernel) as well as the signing side's window walk.
//
// Representation: value = sum l[i] 2^(B i), i = 0..L-1, limbs int32, any value (positive or negative) congruent to
// the residue; p = 2^(32N) - c.  256-bit curve (the graded one, after which the file is named): L = 9, B = 29,
// 2^261 = 2^5 * 2^256 = 6048 (mod p).  384-bit: L = 14, B = 28, 2^392 = 2^8 * 317.  512-bit: L = 19, B = 27,
// 2^513 = 2 * 569 (template LZ<N>).  "u" below = 2^B.
//   N  (normalised) : output of f29_mul / f29_sqr / f29_carry: l[2..] in [0, u); l[0] within 2^20 of [0, u) (f29_carry folds its last
//                     carry there), l[1] within 2^21 (a multiplication folds the carry out of column L-1 there) -- tools/fe29_bounds.py norm()
//   L1 (lazy)       : a difference of two N values, or the negation of one: |l[i]| <= u + 2^21
// Additions and subtractions are limb-wise with no carries at all; a multiplication accepts operands whose limb
// bounds A u and B u satisfy A * B <= 3 (nine products of 2^58 A B plus the carry-in stay below 2^63); the point
// formulas below place a carry pass (f29_carry, 29 instructions) exactly where a value would exceed that
// (tools/fe29_bounds.py propagates the intervals through both formulas and checks every accumulator).
#pragma once
#include "bign_dev.hpp"

namespace bee2hip {

// limb layout per curve: L limbs of B bits, TOP = bits of the value that the last limb holds below 2^(32N),
// FOLD = 2^(B L) mod p
template <int N> struct LZ;
template <> struct LZ<8> { static constexpr int L = 9, B = 29, TOP = 24; static constexpr int32_t FOLD = 189 << 5; };
template <> struct LZ<12> { static constexpr int L = 14, B = 28, TOP = 20; static constexpr int32_t FOLD = 317 << 8; };
template <> struct LZ<16> { static constexpr int L = 19, B = 27, TOP = 26; static constexpr int32_t FOLD = 569 << 1; };

template <int N> struct lzT {
    static constexpr int L = LZ<N>::L, B = LZ<N>::B;
    static constexpr int32_t M = (1 << LZ<N>::B) - 1;
    int32_t l[LZ<N>::L];
};
typedef lzT<8> fe29;
constexpr int32_t F29_M = (1 << 29) - 1;
constexpr int32_t F29_FOLD = 189 * 32;                  // 2^261 mod p

// N x 32-bit words (any value < 2^(32N)) -> N
template <int N>
__device__ __forceinline__ void f29_from_words(lzT<N> &r, const feT<N> &a)
{
    constexpr int L = LZ<N>::L, B = LZ<N>::B;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int bit = B * i, w = bit >> 5, sh = bit & 31;
        const uint32_t lo = a.v[w], hi = w + 1 < N ? a.v[w + 1] : 0u;
        const uint32_t x = sh ? __builtin_amdgcn_alignbit(hi, lo, sh) : lo;
        r.l[i] = (int32_t)(x & (uint32_t)lzT<N>::M);
    }
}

// one floor-carry pass, the carry out of the last limb folded back into limb 0: |l[i]| < 4 u in, N out
template <int N>
__device__ __forceinline__ void f29_carry(lzT<N> &a)
{
    constexpr int L = LZ<N>::L, B = LZ<N>::B;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int32_t t = a.l[i] + c;
        a.l[i] = t & lzT<N>::M;
        c = t >> B;
    }
    a.l[0] += c * LZ<N>::FOLD;
}

template <int N>
__device__ __forceinline__ void f29_add(lzT<N> &r, const lzT<N> &a, const lzT<N> &b)
{
#pragma unroll
    for (int i = 0; i < LZ<N>::L; ++i) r.l[i] = a.l[i] + b.l[i];
}
template <int N>
__device__ __forceinline__ void f29_sub(lzT<N> &r, const lzT<N> &a, const lzT<N> &b)
{
#pragma unroll
    for (int i = 0; i < LZ<N>::L; ++i) r.l[i] = a.l[i] - b.l[i];
}
template <int N>
__device__ __forceinline__ void f29_neg(lzT<N> &r, const lzT<N> &a)
{
#pragma unroll
    for (int i = 0; i < LZ<N>::L; ++i) r.l[i] = -a.l[i];
}
template <int N>
__device__ __forceinline__ void f29_set_one(lzT<N> &r)
{
#pragma unroll
    for (int i = 0; i < LZ<N>::L; ++i) r.l[i] = i == 0;
}

// ---- multiplication (round 6: two carry chains, the fold inside the low one) -------------------------------------
// r = a b:  phase A  columns L .. 2L-2 on their own chain, from zero -> h[0 .. L-2] masked, h[L-1] = the rest (signed, < 2^31);
//           phase B  columns 0 .. L-1: acc = carry + sum a[i] b[k-i] + h[k] FOLD -> r[k] = acc & M, carry = acc >> B;
//           final    the carry out of column L-1 (weight 2^(B L) again, up to 34 bits) times FOLD into r[0], r[1].
// The 256-bit curve takes the same steps as ONE inline-asm block (bign_fe29_asm.inc, generated by tools/gen_f29_asm.py, which says
// why: the C++ below compiles to ~165 instructions per multiplication, the block is 135); the wider curves (14 / 19 limbs: more
// operands than an asm statement may have) run this C++.  tools/fe29_bounds.py replays the order on intervals.
template <int N, bool SQ>
__device__ __forceinline__ void f29_mul2(lzT<N> &r, const lzT<N> &a, const lzT<N> &b)
{
    constexpr int L = LZ<N>::L, B = LZ<N>::B;
    constexpr int32_t M = lzT<N>::M, F = LZ<N>::FOLD;
    int32_t h[L], d[L], o[L];
    if constexpr (SQ) {
#pragma unroll
        for (int i = 0; i < L; ++i) d[i] = a.l[i] * 2;
    }
    const auto column = [&](int64_t &acc, auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        static_for<(k > L - 1 ? k - (L - 1) : 0), (k < L - 1 ? k : L - 1) + 1>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int j = k - i;
            if constexpr (!SQ) acc += (int64_t)a.l[i] * b.l[j];
            else if constexpr (i < j) acc += (int64_t)a.l[i] * d[j];
            else if constexpr (i == j) acc += (int64_t)a.l[i] * a.l[i];
        });
    };
    int64_t acc = 0;
    static_for<L, 2 * L - 1>([&](auto kc) __attribute__((always_inline)) {
        column(acc, kc);
        h[decltype(kc)::value - L] = (int32_t)acc & M;
        acc >>= B;
    });
    h[L - 1] = (int32_t)acc;
    acc = 0;
    static_for<0, L>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        column(acc, kc);
        acc += (int64_t)h[k] * F;
        o[k] = (int32_t)acc & M;
        acc >>= B;
    });
    // c = cl + 2^B ch, |c| < 2^34: cl FOLD < 2^42 -> its low B bits into r[0] (carried once more), the rest and ch FOLD into r[1]
    const int32_t cl = (int32_t)acc & M, ch = (int32_t)(acc >> B);
    const int64_t t = (int64_t)cl * F;
    const int32_t r0 = o[0] + ((int32_t)t & M);
    o[1] += (int32_t)(t >> B) + ch * F + (int32_t)((uint32_t)r0 >> B);
    o[0] = r0 & M;
#pragma unroll
    for (int i = 0; i < L; ++i) r.l[i] = o[i];
}
// r <- K r, K in {1, 2, 3, 4, 8} (may differ per lane): N in (l[1] a little over u), N out
template <int N>
__device__ __forceinline__ void f29_scale(lzT<N> &r, int32_t K)
{
    constexpr int L = LZ<N>::L, B = LZ<N>::B;
    constexpr int32_t M = lzT<N>::M, F = LZ<N>::FOLD;
    int64_t cy = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        cy += (int64_t)r.l[i] * K;
        r.l[i] = (int32_t)cy & M;
        cy >>= B;
    }
    const int32_t r0 = r.l[0] + (int32_t)cy * F;          // |cy| <= 8: its weight is 2^(B L)
    r.l[1] += r0 >> B;
    r.l[0] = r0 & M;
}
#include "bign_fe29_asm.inc"

template <int K = 1, int N>
__device__ __forceinline__ void f29_mul(lzT<N> &r, const lzT<N> &a, const lzT<N> &b)
{
    lzT<N> t;
    if constexpr (N == 8 && K == 1) f29_mul9_asm(t, a, b);
    else if constexpr (N == 8) f29_mul9k_asm(t, a, b, K);
    else {
        f29_mul2<N, false>(t, a, b);
        if constexpr (K != 1) f29_scale(t, K);
    }
    r = t;
}
// r = K a b with a per-lane K in {1, 2, 3, 4, 8}
template <int N>
__device__ __forceinline__ void f29_mul_k(lzT<N> &r, const lzT<N> &a, const lzT<N> &b, int32_t K)
{
    lzT<N> t;
    if constexpr (N == 8) f29_mul9k_asm(t, a, b, K);
    else { f29_mul2<N, false>(t, a, b); f29_scale(t, K); }
    r = t;
}

template <int K = 1, int N>
__device__ __forceinline__ void f29_sqr(lzT<N> &r, const lzT<N> &a)
{
    lzT<N> t;
    if constexpr (N == 8 && K == 1) f29_sqr9_asm(t, a);
    else if constexpr (N == 8) f29_sqr9k_asm(t, a, K);
    else {
        f29_mul2<N, true>(t, a, a);
        if constexpr (K != 1) f29_scale(t, K);
    }
    r = t;
}

// any |l[i]| < 4 u -> N x 32-bit words, weakly reduced (a value in [0, 2^(32N)) congruent to the residue), exactly
template <int N>
__device__ __forceinline__ void f29_to_words(feT<N> &r, lzT<N> a)
{
    constexpr int L = LZ<N>::L, B = LZ<N>::B, TOP = LZ<N>::TOP;
    constexpr int32_t M = lzT<N>::M, C = (int32_t)CurveC<N>::C, TM = (1 << TOP) - 1;
    f29_carry(a);                                   // N: value in (-2^20, 2^(B L) + 2^20)
    // + p (limbs of 2^(32N) - c) makes the value positive; carry without wrap (the last limb keeps what is above its TOP bits)
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int32_t pi = i == 0 ? M + 1 - C : i < L - 1 ? M : TM;
        const int32_t t = a.l[i] + pi + c;
        if (i < L - 1) { a.l[i] = t & M; c = t >> B; }
        else a.l[i] = t;
    }
    // fold the bits from 2^(32N) up
    int32_t top = a.l[L - 1] >> TOP;
    a.l[L - 1] &= TM;
    c = top * C;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int32_t t = a.l[i] + c;
        if (i < L - 1) { a.l[i] = t & M; c = t >> B; }
        else a.l[i] = t;
    }
    // a second wrap leaves a value below (2^(B - TOP) + 2) c: no further carry
    top = a.l[L - 1] >> TOP;
    a.l[L - 1] &= TM;
    a.l[0] += top * C;
#pragma unroll
    for (int w = 0; w < N; ++w) {
        const int lo = 32 * w / B, sh = 32 * w % B;
        uint64_t v = (uint64_t)(uint32_t)a.l[lo] >> sh;
        v |= (uint64_t)(uint32_t)a.l[lo + 1] << (B - sh);
        if (lo + 2 < L && 2 * B - sh < 32) v |= (uint64_t)(uint32_t)a.l[lo + 2] << (2 * B - sh);
        r.v[w] = (uint32_t)v;
    }
}

template <int N> struct ljacT { lzT<N> X, Y, Z; };         // X, Z: N; Y: L1
template <int N> struct laffT { lzT<N> x, y; };            // x: N; y: N or L1 (negated table entry)
typedef ljacT<8> jac29;
typedef laffT<8> aff29;

// T <- 2T (jac_dbl of bign_dev.hpp, same formulas).  Bounds in units of u:
template <int N>
__device__ __forceinline__ void jac29_dbl(ljacT<N> &T)
{
    lzT<N> delta, gamma, beta4, alpha, t0, t1;
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
template <int N>
__device__ __forceinline__ void jac29_madd(ljacT<N> &T, const laffT<N> &E)
{
    lzT<N> Z1Z1, U2, S2, H, HH, HHH, r, V, t;
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
