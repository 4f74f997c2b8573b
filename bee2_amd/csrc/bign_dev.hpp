// bign_dev.hpp -- GF(p) and Jacobian point arithmetic for bign-curve256v1 on one lane.
//
// Replaces, for p = 2^256 - 189 only:
//   zzMul / zzSqr / zzAddMulW        src/math/zz/zz_mul.c:45-154
//   zzRedCrand                       src/math/zz/zz_red.c:71-105
//   zmMulCrand / zmSqrCrand          src/math/zm.c:214-263
//   zzAddMod / zzSubMod              src/math/zz/zz_mod.c:42-133
//   gfpInv (a^(p-2))                 src/math/gfp.c:33-44
//   ecpDblJA3 / ecpAddJ / ecpAddAJ / ecpToAJ   src/math/ecp/ecp_j.c:104-133,241-299,397-590
//
// MI355X mapping.  One lane owns one signature.  A field element is 8 x 32-bit limbs
// in VGPRs (bee2 uses 4 x 64-bit words + 128-bit products; CDNA4's widest multiplier
// is v_mad_u64_u32, 32x32+64).  Measured on gfx950 (tools/ubench/valu_rates.hip):
// v_mad_u64_u32 and every carry-producing/consuming add issue at HALF the rate of a
// plain v_add_u32, so the multiplier is product-scanning with a 64-bit column
// accumulator: one v_mad_u64_u32 (carry-out to an SGPR pair) + one v_addc_co_u32 per
// 32x32 product, no other carry traffic.  Elements are kept only WEAKLY reduced
// (any value < 2^256 congruent to the residue): Crandall folding 2^256 = 189 makes a
// conditional subtraction of p unnecessary until a value is compared or exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bee2hip {

struct fe { uint32_t v[8]; };
struct jac { fe X, Y, Z; };          // O <=> Z == 0 (mod p)
struct aff { fe x, y; };

constexpr uint32_t CRANDALL_C = 189u;          // p = 2^256 - 189 (bign_params.c:36-41)
constexpr uint32_t P_LIMB0 = 0xFFFFFF43u;      // low limb of p; all other limbs 0xFFFFFFFF

// ------------------------------------------------------------------ helpers ---
// acc(64) += a*b ; c2 += carry-out.  Exactly one v_mad_u64_u32 + one v_addc_co_u32.
__device__ __forceinline__ void mac(uint64_t &acc, uint32_t &c2, uint32_t a, uint32_t b)
{
    uint64_t cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b));
    asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(c2), "+s"(cy));
}
__device__ __forceinline__ void fe_set_zero(fe &r)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
}
__device__ __forceinline__ void fe_set_one(fe &r)
{
    fe_set_zero(r);
    r.v[0] = 1;
}
__device__ __forceinline__ uint32_t fe_or_all(const fe &a)
{
    return (a.v[0] | a.v[1]) | (a.v[2] | a.v[3]) | (a.v[4] | a.v[5]) | (a.v[6] | a.v[7]);
}
// a == 0 (mod p) for a weakly reduced a: a is 0 or p
__device__ __forceinline__ bool fe_is_zero(const fe &a)
{
    const uint32_t z = fe_or_all(a);
    const uint32_t hi = (a.v[1] & a.v[2]) & (a.v[3] & a.v[4]) & (a.v[5] & a.v[6]) & a.v[7];
    return z == 0 || (hi == 0xFFFFFFFFu && a.v[0] == P_LIMB0);
}
// 256-bit compare a >= b on raw limbs
__device__ __forceinline__ bool u256_ge(const uint32_t (&a)[8], const uint32_t (&b)[8])
{
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t d = (uint64_t)a[i] - b[i] - borrow;
        borrow = (d >> 32) & 1u;
    }
    return borrow == 0;
}

// --------------------------------------------------------------- add / sub ---
// r = a + b (mod p), weakly reduced.  Carry out of 2^256 folds back as +189; a second
// carry can only come from inputs in the 189-value zone [p, 2^256) and then the wrapped
// value is < 189, so a final +189 on limb 0 alone is exact.
__device__ __forceinline__ void fe_add(fe &r, const fe &a, const fe &b)
{
    uint64_t c = 0;
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c += (uint64_t)a.v[i] + b.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    uint64_t f = (uint64_t)t[0] + ((uint32_t)c ? CRANDALL_C : 0u);
    r.v[0] = (uint32_t)f; f >>= 32;
#pragma unroll
    for (int i = 1; i < 8; ++i) { f += t[i]; r.v[i] = (uint32_t)f; f >>= 32; }
    r.v[0] += (uint32_t)f ? CRANDALL_C : 0u;
}
// r = a - b (mod p), weakly reduced (borrow folds back as -189, mirrored reasoning)
__device__ __forceinline__ void fe_sub(fe &r, const fe &a, const fe &b)
{
    uint32_t t[8];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t d = (uint64_t)a.v[i] - b.v[i] - borrow;
        t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
    }
    uint32_t sub = borrow ? CRANDALL_C : 0u;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t d = (uint64_t)t[i] - (i == 0 ? sub : 0u) - bw;
        r.v[i] = (uint32_t)d; bw = (uint32_t)(d >> 32) & 1u;
    }
    r.v[0] -= bw ? CRANDALL_C : 0u;
}
__device__ __forceinline__ void fe_dbl(fe &r, const fe &a) { fe_add(r, a, a); }
__device__ __forceinline__ void fe_neg(fe &r, const fe &a)
{
    fe z; fe_set_zero(z);
    fe_sub(r, z, a);
}
// the unique representative in [0, p)
__device__ __forceinline__ void fe_canon(fe &r, const fe &a)
{
    // a >= p  <=>  a + 189 carries out of 2^256; then a - p = a + 189 - 2^256
    uint64_t c = CRANDALL_C;
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c += a.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    const bool ge = (uint32_t)c != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = ge ? t[i] : a.v[i];
}
__device__ __forceinline__ void fe_select(fe &r, bool take_b, const fe &a, const fe &b)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = take_b ? b.v[i] : a.v[i];
}

// ---------------------------------------------------------------- reduction ---
// w[0..15] -> r = (lo + K*189*hi) mod p, weakly reduced.  K is a small compile-time
// post-scale (1, 2, 3, 4, 8): r = K * (a*b).  K*189 <= 1512, so lo*K + hi*K*189 needs
// 16 multiplies instead of 8 when K > 1 -- still far cheaper than K-1 modular additions.
template <uint32_t K>
__device__ __forceinline__ void fe_reduce(fe &r, const uint32_t (&w)[16])
{
    uint32_t t[8];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // c < 2^44 always: K*189*(2^32-1) + K*(2^32-1) + c fits easily in 64 bits
        uint64_t s = (uint64_t)w[8 + i] * (K * CRANDALL_C) + c;
        if (K == 1) s += w[i];
        else s += (uint64_t)w[i] * K;
        t[i] = (uint32_t)s; c = s >> 32;
    }
    // fold the top (< 2^12) once more, then the at-most-one final carry
    uint64_t f = (uint64_t)t[0] + (uint64_t)(uint32_t)c * CRANDALL_C;
    r.v[0] = (uint32_t)f; f >>= 32;
#pragma unroll
    for (int i = 1; i < 8; ++i) { f += t[i]; r.v[i] = (uint32_t)f; f >>= 32; }
    r.v[0] += (uint32_t)f ? CRANDALL_C : 0u;
}

// ------------------------------------------------------------- mul / sqr ---
// product scanning: column k sums a[i]*b[k-i] into (c2 : acc64)
template <uint32_t K = 1>
__device__ __forceinline__ void fe_mul(fe &r, const fe &a, const fe &b)
{
    uint32_t w[16];
    uint64_t acc = 0;
    uint32_t c2 = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = k - i;
            if (j >= 0 && j < 8) mac(acc, c2, a.v[i], b.v[j]);
        }
        w[k] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32);
        c2 = 0;
    }
    w[15] = (uint32_t)acc;
    fe_reduce<K>(r, w);
}

// Squaring with 44 multiplications instead of 64: row i multiplies a_i by the vector
// (a_i, e_{i+1}, d_{i+2}, .., d_8) where d = limbs of 2a (9 limbs, d_8 = top bit) and
// e_j = a_j << 1 is the doubled limb WITHOUT the bit shifted in from a_{j-1} (that bit belongs
// to the part of 2a below position i+1, which row i does not use).  Same product-scanning
// accumulator as fe_mul (cf. zzSqr, src/math/zz/zz_mul.c:112-154, which doubles afterwards).
template <uint32_t K = 1>
__device__ __forceinline__ void fe_sqr(fe &r, const fe &a)
{
    uint32_t d[9], e[8];
#pragma unroll
    for (int j = 1; j < 8; ++j) {
        e[j] = a.v[j] << 1;
        d[j] = __builtin_amdgcn_alignbit(a.v[j], a.v[j - 1], 31);       // (a_j << 1) | (a_{j-1} >> 31)
    }
    d[8] = a.v[7] >> 31;
    uint32_t w[16];
    uint64_t acc = 0;
    uint32_t c2 = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = k - i;                     // position inside row i's vector
            if (j == i) mac(acc, c2, a.v[i], a.v[i]);
            else if (j == i + 1 && j < 8) mac(acc, c2, a.v[i], e[j]);
            else if (j >= i + 2 && j <= 8) mac(acc, c2, a.v[i], d[j]);
        }
        w[k] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32);
        c2 = 0;
    }
    fe_reduce<K>(r, w);
}

__device__ __forceinline__ void fe_sqr_n(fe &r, const fe &a, int n)
{
    r = a;
#pragma unroll 1
    for (int i = 0; i < n; ++i) fe_sqr(r, r);
}

// a^(p-2): p - 2 = (2^248 - 1) * 2^8 + 0b01000001.  255 S + 13 M.  By value on purpose:
// a by-reference noinline call would pin the caller's elements in scratch memory.
__device__ __noinline__ fe fe_inv(fe x)
{
    fe x2, x4, x8, x16, x32, x64, t;
    fe_sqr(t, x);           fe_mul(x2, t, x);
    fe_sqr_n(t, x2, 2);     fe_mul(x4, t, x2);
    fe_sqr_n(t, x4, 4);     fe_mul(x8, t, x4);
    fe_sqr_n(t, x8, 8);     fe_mul(x16, t, x8);
    fe_sqr_n(t, x16, 16);   fe_mul(x32, t, x16);
    fe_sqr_n(t, x32, 32);   fe_mul(x64, t, x32);
    fe_sqr_n(t, x64, 64);   fe_mul(t, t, x64);        // 2^128 - 1
    fe_sqr_n(t, t, 64);     fe_mul(t, t, x64);        // 2^192 - 1
    fe_sqr_n(t, t, 32);     fe_mul(t, t, x32);        // 2^224 - 1
    fe_sqr_n(t, t, 16);     fe_mul(t, t, x16);        // 2^240 - 1
    fe_sqr_n(t, t, 8);      fe_mul(t, t, x8);         // 2^248 - 1
    fe_sqr_n(t, t, 2);      fe_mul(t, t, x);          // ..01
    fe_sqr_n(t, t, 6);      fe_mul(t, t, x);          // ..01000001
    return t;
}

// ----------------------------------------------------------------- points ---
// T <- 2T, a = -3.  4M + 4S + 6 add/sub.  Z3 = 2YZ, so Y = 0 or Z = 0 gives O as in
// ecp_j.c:258-263.  (dbl-2001-b with the small multiples moved into the reductions.)
__device__ __forceinline__ void jac_dbl(jac &T)
{
    fe delta, gamma, beta4, alpha, t0, t1;
    fe_sqr(delta, T.Z);
    fe_sqr(gamma, T.Y);
    fe_mul<4>(beta4, T.X, gamma);            // 4 X Y^2
    fe_sub(t0, T.X, delta);
    fe_add(t1, T.X, delta);
    fe_mul<3>(alpha, t0, t1);                // 3 (X - Z^2)(X + Z^2)
    fe_mul<2>(T.Z, T.Y, T.Z);                // Z3 = 2 Y Z
    fe_sqr(t0, alpha);
    fe_dbl(t1, beta4);
    fe_sub(T.X, t0, t1);                     // X3 = alpha^2 - 8 beta
    fe_sqr<8>(t1, gamma);                    // 8 Y^4
    fe_sub(t0, beta4, T.X);
    fe_mul(t0, alpha, t0);
    fe_sub(T.Y, t0, t1);                     // Y3 = alpha (4 beta - X3) - 8 Y^4
}

// T <- T + E for Jacobian E (add-1998-cmo-2, 12M + 4S).  Returns false when the generic
// formula does not apply (either operand O, or T = +-E): the caller then marks the
// signature for the complete slow path (ecp_j.c:416-427,455-464 handle these inline).
__device__ __forceinline__ bool jac_add(jac &T, const jac &E)
{
    fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, HH, HHH, r, V, t;
    const bool bad_in = fe_is_zero(T.Z) || fe_is_zero(E.Z);
    fe_sqr(Z1Z1, T.Z);
    fe_sqr(Z2Z2, E.Z);
    fe_mul(U1, T.X, Z2Z2);
    fe_mul(U2, E.X, Z1Z1);
    fe_mul(t, E.Z, Z2Z2);  fe_mul(S1, T.Y, t);
    fe_mul(t, T.Z, Z1Z1);  fe_mul(S2, E.Y, t);
    fe_sub(H, U2, U1);
    const bool bad = bad_in || fe_is_zero(H);
    fe_sub(r, S2, S1);
    fe_sqr(HH, H);
    fe_mul(HHH, H, HH);
    fe_mul(V, U1, HH);
    fe_mul(t, T.Z, E.Z);   fe_mul(T.Z, t, H);           // Z3 = Z1 Z2 H
    fe_sqr(t, r);
    fe_sub(t, t, HHH);
    fe_dbl(U2, V);
    fe_sub(T.X, t, U2);                                 // X3 = r^2 - H^3 - 2V
    fe_sub(t, V, T.X);
    fe_mul(t, r, t);
    fe_mul(S2, S1, HHH);
    fe_sub(T.Y, t, S2);                                 // Y3 = r (V - X3) - S1 H^3
    return !bad;
}

// T <- T + E for affine E (madd, 8M + 3S); same contract as jac_add
__device__ __forceinline__ bool jac_madd(jac &T, const aff &E)
{
    fe Z1Z1, U2, S2, H, HH, HHH, r, V, t;
    const bool bad_in = fe_is_zero(T.Z);
    fe_sqr(Z1Z1, T.Z);
    fe_mul(U2, E.x, Z1Z1);
    fe_mul(t, T.Z, Z1Z1);  fe_mul(S2, E.y, t);
    fe_sub(H, U2, T.X);
    const bool bad = bad_in || fe_is_zero(H);
    fe_sub(r, S2, T.Y);
    fe_sqr(HH, H);
    fe_mul(HHH, H, HH);
    fe_mul(V, T.X, HH);
    fe_mul(T.Z, T.Z, H);                                // Z3 = Z1 H
    fe_sqr(t, r);
    fe_sub(t, t, HHH);
    fe_dbl(U2, V);
    fe_sub(T.X, t, U2);
    fe_sub(t, V, T.X);
    fe_mul(t, r, t);
    fe_mul(S2, T.Y, HHH);
    fe_sub(T.Y, t, S2);
    return !bad;
}

// complete addition (all exceptional cases, as ecpAddJ ecp_j.c:397-497): slow path only
__device__ __forceinline__ void jac_add_complete(jac &T, const jac &E)
{
    if (fe_is_zero(E.Z)) return;
    if (fe_is_zero(T.Z)) { T = E; return; }
    fe Z1Z1, Z2Z2, U1, U2, S1, S2, t;
    fe_sqr(Z1Z1, T.Z);
    fe_sqr(Z2Z2, E.Z);
    fe_mul(U1, T.X, Z2Z2);
    fe_mul(U2, E.X, Z1Z1);
    fe_mul(t, E.Z, Z2Z2);  fe_mul(S1, T.Y, t);
    fe_mul(t, T.Z, Z1Z1);  fe_mul(S2, E.Y, t);
    fe_sub(t, U2, U1);
    if (fe_is_zero(t)) {
        fe_sub(t, S2, S1);
        if (fe_is_zero(t)) jac_dbl(T);                  // T == E
        else fe_set_zero(T.Z);                          // T == -E
        return;
    }
    (void)jac_add(T, E);
}

}  // namespace bee2hip
