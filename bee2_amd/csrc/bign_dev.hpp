// bign_dev.hpp -- GF(p) and Jacobian point arithmetic for the bign curves on one lane.
//
// Replaces, for p = 2^(32 N) - c (N = 8, 12, 16 limbs; c = 189, 317, 569 -- the three
// parameter sets of STB 34.101.45 annex B, src/crypto/bign/bign_params.c:34-178):
//   zzMul / zzSqr / zzAddMulW        src/math/zz/zz_mul.c:45-154
//   zzRedCrand                       src/math/zz/zz_red.c:71-105
//   zmMulCrand / zmSqrCrand          src/math/zm.c:214-263
//   zzAddMod / zzSubMod              src/math/zz/zz_mod.c:42-133
//   gfpInv (a^(p-2))                 src/math/gfp.c:33-44
//   ecpDblJA3 / ecpAddJ / ecpAddAJ / ecpToAJ   src/math/ecp/ecp_j.c:104-133,241-299,397-590
//
// MI355X mapping.  One lane owns one signature.  A field element is N x 32-bit limbs
// in VGPRs (bee2 uses 64-bit words + 128-bit products; CDNA4's widest multiplier
// is v_mad_u64_u32, 32x32+64).  Measured on gfx950 (tools/ubench/valu_rates.hip):
// v_mad_u64_u32 and every carry-producing/consuming add issue at HALF the rate of a
// plain v_add_u32, so the multiplier is product-scanning with a 64-bit column
// accumulator: one v_mad_u64_u32 (carry-out to an SGPR pair) + one v_addc_co_u32 per
// 32x32 product, no other carry traffic.  Elements are kept only WEAKLY reduced
// (any value < 2^(32N) congruent to the residue): Crandall folding 2^(32N) = c makes a
// conditional subtraction of p unnecessary until a value is compared or exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bash_dev.hpp"        // bitop3

namespace bee2hip {

// compile-time loop: f(integral_constant<int, I>) for I = Begin .. End-1, in order
template <int I> struct IntC { static constexpr int value = I; };
template <int Begin, int End, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (Begin < End) {
        f(IntC<Begin>{});
        static_for<Begin + 1, End>(f);
    }
}

template <int N> struct CurveC;
template <> struct CurveC<8> { static constexpr uint32_t C = 189u; static constexpr int LOWBITS = 8; };
template <> struct CurveC<12> { static constexpr uint32_t C = 317u; static constexpr int LOWBITS = 9; };
template <> struct CurveC<16> { static constexpr uint32_t C = 569u; static constexpr int LOWBITS = 10; };

template <int N> struct feT { uint32_t v[N]; };

// Two flavours of the field operations.  CtOps (the default): straight-line code only, what the signing kernels
// need (their operands are secret).  VtOps: VERIFICATION ONLY (all operands public): the second carry pass of an
// addition / subtraction / Crandall fold -- needed with probability 2^-24 (a + b wraps past 2^(32N) AND limb 0 then
// overflows on +c) resp. 2^-40 per lane -- is skipped by a wavefront-uniform branch on the carry mask the first
// pass leaves in an SGPR pair, and the carry chains are written as v_add(c)_co_u32 chains instead of 64-bit adds of
// zero-extended limbs.  Results are identical, bit for bit; only the instruction count depends on the data.
struct CtOps { static constexpr bool VT = false, PAIRS = false; };
struct VtOps { static constexpr bool VT = true, PAIRS = false; };
// VtOps with the two multiply-adds of a column pair issued back to back (mac2): for kernels that run at ONE wavefront per SIMD
struct VtOpsP { static constexpr bool VT = true, PAIRS = true; };
template <int N> struct jacT { feT<N> X, Y, Z; };          // O <=> Z == 0 (mod p)
template <int N> struct affT { feT<N> x, y; };
typedef feT<8> fe;
typedef jacT<8> jac;
typedef affT<8> aff;

// ------------------------------------------------------------------ helpers ---
// acc(64) += a*b ; c2 += carry-out.  Exactly one v_mad_u64_u32 + one v_addc_co_u32.
__device__ __forceinline__ void mac(uint64_t &acc, uint32_t &c2, uint32_t a, uint32_t b)
{
    uint64_t cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b));
    asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(c2), "+s"(cy));
}
// The first product of a column WRITES c2 (= 0 + 0 + carry) instead of accumulating into it, so
// the counter needs no zero-initialising v_mov per column.
// NOCARRY: the sum cannot leave 64 bits -- column 0 (one product into an empty accumulator) and the last column (the whole
// product fits 2N limbs) -- so the carry is not collected (round 3: 2 of the 64 / 44 v_addc_co_u32 of a product)
template <bool FIRST, bool NOCARRY = false>
__device__ __forceinline__ void mac_col(uint64_t &acc, uint32_t &c2, uint32_t a, uint32_t b)
{
    if constexpr (NOCARRY) {
        uint64_t cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b));
        if constexpr (FIRST) c2 = 0;
    } else if constexpr (FIRST) {
        uint64_t cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b));
        asm("v_addc_co_u32 %0, %1, 0, 0, %1" : "=v"(c2), "+s"(cy));
    } else {
        mac(acc, c2, a, b);
    }
}

// TWO products of one column: mad, mad, addc, addc, each carry in its own SGPR pair.  gfx940+ wants a wait state between a VALU
// that writes an SGPR and the VALU that reads it; alone, every v_mad_u64_u32 / v_addc_co_u32 pair gets an s_nop from the
// compiler on both sides (3 878 of the 9 900 instructions of bign_main_kernel<8>).  Here the second multiply-add IS the wait
// state of the first carry: same arithmetic, no s_nop -- +8.6 % multiply-add throughput at 4 wavefronts per SIMD, +21 % at 3,
// +67 % on a lone wavefront in tools/ubench/mad_chain.hip (form D).  MEASURED IN THE KERNELS AND NOT TAKEN: bign_main_kernel<8>
// drops from 9 932 to 7 974 instructions (s_nop 3 878 -> 1 900) and 2^18 verifications take the same 2.17 ms (120.4 against
// 120.0-120.6 M/s), signing +1 % -- the s_nop of a wavefront are issue slots other wavefronts fill (profiles/r03_mad_pairs.txt).
// BIGN_MAC_PAIRS=1 builds the paired form (all 232 bign GPU tests pass with it); 0, the product, is the audited order.
// (round 4, second session) TAKEN where it pays: the ops class VtOpsP (PAIRS) builds the paired form.  What round 3 measured was the
// main kernel of the 256-bit curve at FOUR wavefronts per SIMD, where other wavefronts fill the wait states.  At ONE wavefront per
// SIMD a wait state is an issue slot lost (a lone wavefront gets one slot per 4 cycles, s_nop included: tools/ubench/lone_chain.hip):
// the 384- / 512-bit pipelines at 2^16 signatures 2.83 -> 2.12 ms and 6.16 -> 4.67 ms; at two wavefronts and more the paired form
// LOSES 1-6 % (two quarter-rate instructions back to back).  launch_bign_verify_t picks by batch size; profiles/r04_mad_pairs_vt.txt.
#ifndef BIGN_MAC_PAIRS
#define BIGN_MAC_PAIRS 0
#endif
template <bool FIRST, bool PAIRS = false>
__device__ __forceinline__ void mac2(uint64_t &acc, uint32_t &c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1)
{
  if constexpr (BIGN_MAC_PAIRS || PAIRS) {
    uint64_t cy0, cy1;
    if constexpr (FIRST)
        asm("v_mad_u64_u32 %0, %2, %4, %5, %0\n\tv_mad_u64_u32 %0, %3, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, %2, 0, 0, %2\n\tv_addc_co_u32 %1, %3, 0, %1, %3"
            : "+v"(acc), "=&v"(c2), "=&s"(cy0), "=&s"(cy1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
    else
        asm("v_mad_u64_u32 %0, %2, %4, %5, %0\n\tv_mad_u64_u32 %0, %3, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, %2, 0, %1, %2\n\tv_addc_co_u32 %1, %3, 0, %1, %3"
            : "+v"(acc), "+v"(c2), "=&s"(cy0), "=&s"(cy1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
  } else {
    mac_col<FIRST>(acc, c2, a0, b0);
    mac_col<false>(acc, c2, a1, b1);
  }
}

template <int N>
__device__ __forceinline__ void fe_set_zero(feT<N> &r)
{
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = 0;
}
template <int N>
__device__ __forceinline__ void fe_set_one(feT<N> &r)
{
    fe_set_zero(r);
    r.v[0] = 1;
}
// a == 0 (mod p) for a weakly reduced a: a is 0 or p
template <int N>
__device__ __forceinline__ bool fe_is_zero(const feT<N> &a)
{
    uint32_t z = a.v[0], hi = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 1; i < N; ++i) { z |= a.v[i]; hi &= a.v[i]; }
    return z == 0 || (hi == 0xFFFFFFFFu && a.v[0] == 0u - CurveC<N>::C);
}
// multi-limb compare a >= b on raw limbs
template <int N>
__device__ __forceinline__ bool limbs_ge(const uint32_t (&a)[N], const uint32_t (&b)[N])
{
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)a[i] - b[i] - borrow;
        borrow = (d >> 32) & 1u;
    }
    return borrow == 0;
}

// --------------------------------------------------------------- add / sub ---
// r = a + b (mod p), weakly reduced.  Carry out of 2^(32N) folds back as +c; a second
// carry can only come from inputs in the c-value zone [p, 2^(32N)) and then the wrapped
// value is < c, so a final +c on limb 0 alone is exact.
template <class P = CtOps, int N>
__device__ __forceinline__ void fe_add(feT<N> &r, const feT<N> &a, const feT<N> &b)
{
    constexpr uint32_t C = CurveC<N>::C;
    uint32_t t[N], k;
    uint64_t cy, cy2;
    const uint32_t cv = C;
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(t[0]), "=s"(cy) : "v"(a.v[0]), "v"(b.v[0]));
#pragma unroll
    for (int i = 1; i < N; ++i)
        asm("v_addc_co_u32 %0, %1, %2, %3, %1" : "=v"(t[i]), "+s"(cy) : "v"(a.v[i]), "v"(b.v[i]));
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(k) : "v"(cv), "s"(cy));           // wrapped past 2^(32N): + c
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(t[0]), "=s"(cy2) : "v"(t[0]), "v"(k));
    // limb 0 overflowed in some lane (probability 2^-24 on random data): ripple.  The verification flavour skips the
    // pass with a wavefront-uniform branch; the constant-time flavour always runs it (round 3: the same carry chains
    // instead of 64-bit adds of zero-extended limbs: 20 half-rate instructions instead of ~50 mixed ones).
    if (!P::VT || __builtin_expect(cy2 != 0, 0)) {
#pragma unroll
        for (int i = 1; i < N; ++i) asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(t[i]), "+s"(cy2));
        uint32_t top;
        asm("v_addc_co_u32 %0, %1, 0, 0, %1" : "=v"(top), "+s"(cy2));
        t[0] += (0u - top) & C;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
}
// r = a - b (mod p), weakly reduced (borrow folds back as -c, mirrored reasoning)
template <class P = CtOps, int N>
__device__ __forceinline__ void fe_sub(feT<N> &r, const feT<N> &a, const feT<N> &b)
{
    constexpr uint32_t C = CurveC<N>::C;
    uint32_t t[N], k;
    uint64_t bw, bw2;
    const uint32_t cv = C;
    asm("v_sub_co_u32 %0, %1, %2, %3" : "=v"(t[0]), "=s"(bw) : "v"(a.v[0]), "v"(b.v[0]));
#pragma unroll
    for (int i = 1; i < N; ++i)
        asm("v_subb_co_u32 %0, %1, %2, %3, %1" : "=v"(t[i]), "+s"(bw) : "v"(a.v[i]), "v"(b.v[i]));
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(k) : "v"(cv), "s"(bw));            // borrowed from 2^(32N): - c
    asm("v_sub_co_u32 %0, %1, %2, %3" : "=v"(t[0]), "=s"(bw2) : "v"(t[0]), "v"(k));
    if (!P::VT || __builtin_expect(bw2 != 0, 0)) {                                   // as fe_add
#pragma unroll
        for (int i = 1; i < N; ++i) asm("v_subb_co_u32 %0, %1, %0, 0, %1" : "+v"(t[i]), "+s"(bw2));
        uint32_t top;                                                               // 0 or 0xFFFFFFFF
        asm("v_subb_co_u32 %0, %1, 0, 0, %1" : "=v"(top), "+s"(bw2));
        t[0] -= top & C;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
}
template <class P = CtOps, int N>
__device__ __forceinline__ void fe_dbl(feT<N> &r, const feT<N> &a) { fe_add<P>(r, a, a); }
template <class P = CtOps, int N>
__device__ __forceinline__ void fe_neg(feT<N> &r, const feT<N> &a)
{
    feT<N> z; fe_set_zero(z);
    fe_sub<P>(r, z, a);
}
// the unique representative in [0, p)
template <int N>
__device__ __forceinline__ void fe_canon(feT<N> &r, const feT<N> &a)
{
    // a >= p  <=>  a + c carries out of 2^(32N); then a - p = a + c - 2^(32N)
    uint64_t c = CurveC<N>::C;
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { c += a.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    const bool ge = (uint32_t)c != 0;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = ge ? t[i] : a.v[i];
}

// ---------------------------------------------------------------- reduction ---
// w[0..2N) -> r = K * (lo + c*hi) mod p, weakly reduced.  K is a small compile-time
// post-scale (1, 2, 3, 4, 8): r = K * (a*b).  K*c <= 4552, so lo*K + hi*K*c needs 2N
// multiplies instead of N when K > 1 -- still far cheaper than K-1 modular additions.
template <uint32_t K, class P = CtOps, int N>
__device__ __forceinline__ void fe_reduce(feT<N> &r, const uint32_t (&w)[2 * N])
{
    constexpr uint32_t C = CurveC<N>::C;
    uint32_t t[N];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // s = w[N+i] * (K C) + w[i] * K + c as TWO multiply-adds into one 64-bit accumulator (c < 2^14, so nothing overflows).
        // Left to the compiler, "w[i] * K" became v_lshlrev_b64 + two v_and + v_lshl_add_u64 for K = 2, 4, 8 and "+ w[i]" a
        // zero-extension plus v_lshl_add_u64 for K = 1 (13 / 7 issue cycles per limb where the multiplier needs 5).
        if constexpr (N > 12 || !P::VT) {   // 512-bit curve and the signing kernels: the compiler's form is 1.5 % faster there (register pressure)
            uint64_t s0 = (uint64_t)w[N + i] * (K * C) + c;
            if (K == 1) s0 += w[i];
            else s0 += (uint64_t)w[i] * K;
            t[i] = (uint32_t)s0; c = s0 >> 32;
            continue;
        }
        uint64_t s, unused;
        const uint32_t kc = K * C;
        // (the carry is a SOURCE pair of the first multiply-add, not its destination: (hi(s), 0) then costs one v_mov, the zero
        // register being shared; as "+v" the compiler copied the pair first, v_mov_b32 + v_mov_b64 per limb)
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(s), "=s"(unused) : "v"(w[N + i]), "s"(kc), "v"(c));
        asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(s), "=s"(unused) : "v"(w[i]), "n"(K));
        t[i] = (uint32_t)s; c = s >> 32;
    }
    // fold the top (c <= K*(C+1) < 2^14, so c*C < 2^24) once more, then the at-most-one final carry.
    // A plain 32-bit carry chain: one half-rate op per limb.  (Written as 64-bit adds the compiler
    // zero-extends every limb into a register pair: 2 v_mov + 1 v_lshl_add_u64 per limb.)
    const uint32_t cc = (uint32_t)c * C;
    uint64_t cy;
    if constexpr (P::VT) {
        // cc < 2^24: the carry leaves limb 1 with probability 2^-40 per lane -- two limbs, then a wavefront-uniform test
        asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(t[0]), "=s"(cy) : "v"(t[0]), "v"(cc));
        asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(t[1]), "+s"(cy));
        if (__builtin_expect(cy != 0, 0)) {
#pragma unroll
            for (int i = 2; i < N; ++i) asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(t[i]), "+s"(cy));
            uint32_t top;
            asm("v_addc_co_u32 %0, %1, 0, 0, %1" : "=v"(top), "+s"(cy));
            t[0] += (0u - top) & C;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = t[i];
        return;
    }
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(r.v[0]), "=s"(cy) : "v"(t[0]), "v"(cc));
#pragma unroll
    for (int i = 1; i < N; ++i)
        asm("v_addc_co_u32 %0, %1, 0, %2, %1" : "=v"(r.v[i]), "+s"(cy) : "v"(t[i]));
    // carry out of the top limb: the value wrapped past 2^(32N) (what is left is < 2^24), add C once
    uint32_t top;
    asm("v_addc_co_u32 %0, %1, 0, 0, %1" : "=v"(top), "+s"(cy));
    r.v[0] += (0u - top) & C;
}

// ------------------------------------------------------------- mul / sqr ---
// product scanning: column k sums a[i]*b[k-i] into (c2 : acc64)
template <uint32_t K, class P = CtOps, int N>
__device__ __forceinline__ void fe_mul_body(feT<N> &r, const feT<N> &a, const feT<N> &b)
{
    uint32_t w[2 * N];
    uint64_t acc = 0;
    uint32_t c2 = 0;
    static_for<0, 2 * N - 1>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;          // first row with a product in column k
        constexpr int cnt = (k < N ? k : N - 1) + 1 - i0;  // products in column k, taken two at a time (mac2)
        static_for<0, cnt / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int i = i0 + 2 * decltype(pc)::value;
            mac2<i == i0, P::PAIRS>(acc, c2, a.v[i], b.v[k - i], a.v[i + 1], b.v[k - i - 1]);
        });
        if constexpr (cnt % 2) mac_col<cnt == 1, k == 0 || k == 2 * N - 2>(acc, c2, a.v[i0 + cnt - 1], b.v[k - i0 - cnt + 1]);
        w[k] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32);
    });
    w[2 * N - 1] = (uint32_t)acc;
    fe_reduce<K, P>(r, w);
}

// Squaring with N(N+3)/2 multiplications instead of N^2: row i multiplies a_i by the vector
// (a_i, e_{i+1}, d_{i+2}, .., d_N) where d = limbs of 2a (N+1 limbs, d_N = top bit) and
// e_j = a_j << 1 is the doubled limb WITHOUT the bit shifted in from a_{j-1} (that bit belongs
// to the part of 2a below position i+1, which row i does not use).  Same product-scanning
// accumulator as fe_mul (cf. zzSqr, src/math/zz/zz_mul.c:112-154, which doubles afterwards).
template <uint32_t K, class P = CtOps, int N>
__device__ __forceinline__ void fe_sqr_body(feT<N> &r, const feT<N> &a)
{
    uint32_t d[N + 1], e[N];
#pragma unroll
    for (int j = 1; j < N; ++j) {
        e[j] = a.v[j] << 1;
        d[j] = __builtin_amdgcn_alignbit(a.v[j], a.v[j - 1], 31);       // (a_j << 1) | (a_{j-1} >> 31)
    }
    d[N] = a.v[N - 1] >> 31;
    uint32_t w[2 * N];
    uint64_t acc = 0;
    uint32_t c2 = 0;
    // Column k sums row i's element at position j = k - i.  The (k, i) pairs are enumerated at
    // compile time (static_for + if constexpr), not with `#pragma unroll` over the 2N x N square:
    // for N = 16 that is 512 bodies, the unroller gives up, and what is left evaluates the three
    // index conditions at run time -- 1 456 scalar instructions and 472 branches per squaring
    // (profiles/r01_bign512_sqr_fix.txt: SQ_INSTS_SALU > SQ_INSTS_VALU in bign_main_kernel<16>).
    static_for<0, 2 * N>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        // rows i0 .. k / 2 have a product in column k (none in column 2N - 1): row i meets position j = k - i of its vector,
        // a_i itself for j = i, e_j for j = i + 1, d_j beyond; two products at a time (mac2)
        constexpr int i0 = k < N ? 0 : k - N;
        constexpr int cnt = k <= 2 * N - 2 ? k / 2 - i0 + 1 : 0;
        auto other = [&](auto ic) __attribute__((always_inline)) -> uint32_t {
            constexpr int i = decltype(ic)::value, j = k - i;
            if constexpr (j == i) return a.v[i];
            else if constexpr (j == i + 1) return e[j];
            else return d[j];
        };
        static_for<0, cnt / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int i = i0 + 2 * decltype(pc)::value;
            mac2<i == i0, P::PAIRS>(acc, c2, a.v[i], other(IntC<i>{}), a.v[i + 1], other(IntC<i + 1>{}));
        });
        if constexpr (cnt % 2) {
            constexpr int i = i0 + cnt - 1;
            mac_col<cnt == 1, k == 0 || k == 2 * N - 2>(acc, c2, a.v[i], other(IntC<i>{}));
        }
        w[k] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32);
        c2 = 0;
    });
    fe_reduce<K, P>(r, w);
}

// The 256-bit curve (the graded path) inlines every multiplication.  For N = 12 / 16 the fully
// unrolled bodies (144 / 256 multiply-adds) are compiled ONCE per (K, N) as real functions taking
// and returning elements by value (in VGPRs): it keeps the build in seconds instead of minutes and
// relieves the register allocator of the callers.
template <uint32_t K, class P, int N>
__device__ __noinline__ feT<N> fe_mul_call(feT<N> a, feT<N> b)
{
    feT<N> r;
    fe_mul_body<K, P>(r, a, b);
    return r;
}
template <uint32_t K, class P, int N>
__device__ __noinline__ feT<N> fe_sqr_call(feT<N> a)
{
    feT<N> r;
    fe_sqr_body<K, P>(r, a);
    return r;
}
// Round 3: the verification kernels of the wider curves inline their multiplications too (BIGN_INLINE_WIDE = 1): the
// out-of-line calls cost by-value register shuffles and spills -- bign_main_kernel<12> 191 VGPRs + 64 B scratch -> 180 and
// none, <16> 251 + 80 B -> 240 and none; 2^18 signatures +21 % / +16 % (profiles/r03_verify_wide.txt), and the build takes
// the same three minutes.  The constant-time (signing) kernels keep the calls.
#ifndef BIGN_INLINE_WIDE
#define BIGN_INLINE_WIDE 1
#endif
template <uint32_t K = 1, class P = CtOps, int N>
__device__ __forceinline__ void fe_mul(feT<N> &r, const feT<N> &a, const feT<N> &b)
{
    if (N == 8 || (BIGN_INLINE_WIDE && P::VT)) fe_mul_body<K, P>(r, a, b);
    else r = fe_mul_call<K, P, N>(a, b);
}
template <uint32_t K = 1, class P = CtOps, int N>
__device__ __forceinline__ void fe_sqr(feT<N> &r, const feT<N> &a)
{
    if (N == 8 || (BIGN_INLINE_WIDE && P::VT)) fe_sqr_body<K, P>(r, a);
    else r = fe_sqr_call<K, P, N>(a);
}

template <int N>
__device__ __forceinline__ void fe_sqr_n(feT<N> &r, const feT<N> &a, int n)
{
    r = a;
#pragma unroll 1
    for (int i = 0; i < n; ++i) fe_sqr(r, r);
}

// a^(p-2).  p - 2 = (2^m - 1) * 2^k + low with k = LOWBITS, m = 32N - k, low = 2^k - (c + 2)
// (N = 8: 248 ones then 01000001).  x^(2^m - 1) by the doubling chain over the bits of m,
// then k squarings interleaved with multiplications by x for the set bits of `low`.
// By value on purpose: a by-reference noinline call would pin the caller's elements in scratch.
template <int N>
__device__ __noinline__ feT<N> fe_inv(feT<N> x)
{
    constexpr int k = CurveC<N>::LOWBITS;
    constexpr int m = 32 * N - k;
    constexpr uint32_t low = (1u << k) - (CurveC<N>::C + 2u);
    // ones(a) := x^(2^a - 1).  ones(2a) = ones(a)^(2^a) * ones(a); ones(a+1) = ones(a)^2 * x
    feT<N> r = x, t;
    int a = 1;
    int top = 31;
    while (!((m >> top) & 1)) --top;
#pragma unroll 1
    for (int bit = top - 1; bit >= 0; --bit) {
        fe_sqr_n(t, r, a);
        fe_mul(r, t, r);
        a *= 2;
        if ((m >> bit) & 1) {
            fe_sqr(t, r);
            fe_mul(r, t, x);
            a += 1;
        }
    }
#pragma unroll 1
    for (int bit = k - 1; bit >= 0; --bit) {
        fe_sqr(r, r);
        if ((low >> bit) & 1u) fe_mul(r, r, x);
    }
    return r;
}

// ------------------------------------------------- inversion by division steps ---
// a^-1 mod p with the Bernstein-Yang "safegcd" recurrence (half-delta variant) instead of a^(p-2): about
// 2.3 * 32N division steps of a dozen full-rate VALU ops each, applied to the full-width numbers in batches
// of 30 through a 2x2 transition matrix -- ~11 k instructions on the 256-bit curve against ~62 k for the
// Fermat chain (268 dependent field multiplications).  Numbers are L signed limbs of 30 bits
// (f, g in (-p, p); d, e in (-2p, p)).  Branch-free inside an iteration; the loop ends when g = 0 in every
// lane of the wavefront (further steps leave f and d untouched).  bee2 itself inverts by a^(p-2)
// (gfpInv, gfp.c:33-44): the result is the same canonical residue either way, and fe_inv_checked below
// verifies a * r = 1 and falls back to the Fermat chain otherwise.
template <int N> struct SafeGcd {
    static constexpr int L = (32 * N + 2 + 29) / 30;           // 9, 13, 18
    static constexpr int FULL = 32 * N / 30, REM = 32 * N % 30; // p = 2^(30 FULL + REM) - c
    static_assert(REM != 0 && L == FULL + 1, "limb layout of p");
    static constexpr int32_t M30 = 0x3FFFFFFF;
    static constexpr int32_t mod_limb(int i)
    {
        return i == 0 ? (int32_t)((1u << 30) - CurveC<N>::C) : i < FULL ? M30 : (int32_t)((1u << REM) - 1u);
    }
    static constexpr uint32_t inv30()                           // p^-1 mod 2^30 (Newton iteration mod 2^32)
    {
        const uint32_t a = 0u - CurveC<N>::C;                   // p mod 2^32
        uint32_t x = 1;
        for (int i = 0; i < 6; ++i) x *= 2u - a * x;
        return x & 0x3FFFFFFFu;
    }
    static constexpr int MAX_ITER = (32 * N * 24 / 10 + 60) / 30;   // generous; the g == 0 test ends the loop
    // the constant-time callers run a FIXED number of iterations: the proven bound of the half-delta variant for b-bit inputs,
    // (45907 b + 26313) / 19929 division steps (592 / 886 / 1181 rounded up), in batches of 30: 20 / 30 / 40 iterations
    // (round 4: was MAX_ITER = 22 / 32 / 42; the Z Z^-1 = 1 check of the callers would report a shortfall loudly)
    static constexpr int CT_ITER = ((45907 * 32 * N + 26313 + 19928) / 19929 + 29) / 30;
    static_assert(CT_ITER <= MAX_ITER && CT_ITER * 30 >= (45907 * 32 * N + 26313) / 19929 + 1, "division-step bound");
};

// 30 division steps on the low limbs; returns the new zeta = -(delta + 1/2), matrix t = (u v; q r) with
// 2^30 (f', g') = t (f, g).  Entries are signed, |.| <= 2^30, carried as uint32 mod 2^32.
__device__ __forceinline__ int32_t sg_divsteps_30(int32_t zeta, uint32_t f, uint32_t g, uint32_t &u, uint32_t &v,
                                                  uint32_t &q, uint32_t &r)
{
    u = 1; v = 0; q = 0; r = 1;
#pragma unroll 1
    for (int i = 0; i < 30; ++i) {
        uint32_t m1 = (uint32_t)(zeta >> 31);                   // delta > 0
        const uint32_t m2 = 0u - (g & 1u);                      // g odd
        const uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;
        g += x & m2; q += y & m2; r += z & m2;
        m1 &= m2;                                               // swap step
        zeta = (int32_t)((uint32_t)zeta ^ m1) - 1;
        f += g & m1; u += q & m1; v += r & m1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    return zeta;
}

// a^-1 mod p for canonical a (0 -> 0); no fallback.  CT = true: no early exit -- all MAX_ITER iterations run whatever
// the value (the bound of the half-delta variant, (45907 b + 26313) / 19929 division steps for b-bit inputs: 590 / 886 /
// 1181 for the three curves, is below 30 CT_ITER = 600 / 900 / 1200), and nothing else in the function depends on
// the data (masks only): the form the signing kernels use on secret Z coordinates.
template <int N, bool CT = false>
__device__ __noinline__ feT<N> fe_inv_safegcd(feT<N> a)
{
    using SG = SafeGcd<N>;
    constexpr int L = SG::L;
    constexpr int32_t M30 = SG::M30;
    int32_t d[L], e[L], f[L], g[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint32_t lo = w < N ? a.v[w] : 0u, hi = w + 1 < N ? a.v[w + 1] : 0u;
        g[i] = (int32_t)((uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & (uint32_t)M30);
        f[i] = SG::mod_limb(i);
        d[i] = 0; e[i] = i == 0 ? 1 : 0;
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < (CT ? SG::CT_ITER : SG::MAX_ITER); ++it) {
        if (!CT) {
            int32_t nz = 0;
#pragma unroll
            for (int i = 0; i < L; ++i) nz |= g[i];
            if (!__any(nz != 0)) break;
        }
        uint32_t uu, vv, qq, rr;
        zeta = sg_divsteps_30(zeta, (uint32_t)f[0], (uint32_t)g[0], uu, vv, qq, rr);
        const int32_t u = (int32_t)uu, v = (int32_t)vv, q = (int32_t)qq, r = (int32_t)rr;
        // (d, e) <- t (d, e) / 2^30 mod p: multiples md, me of p make the low 30 bits vanish
        {
            const int32_t sd = d[L - 1] >> 31, se = e[L - 1] >> 31;
            int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
            int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0];
            int64_t ce = (int64_t)q * d[0] + (int64_t)r * e[0];
            md -= (int32_t)((SG::inv30() * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
            me -= (int32_t)((SG::inv30() * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd += (int64_t)SG::mod_limb(0) * md;
            ce += (int64_t)SG::mod_limb(0) * me;
            cd >>= 30; ce >>= 30;
#pragma unroll
            for (int i = 1; i < L; ++i) {
                cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)SG::mod_limb(i) * md;
                ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)SG::mod_limb(i) * me;
                d[i - 1] = (int32_t)cd & M30; cd >>= 30;
                e[i - 1] = (int32_t)ce & M30; ce >>= 30;
            }
            d[L - 1] = (int32_t)cd; e[L - 1] = (int32_t)ce;
        }
        // (f, g) <- t (f, g) / 2^30 (exact)
        {
            int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0];
            int64_t cg = (int64_t)q * f[0] + (int64_t)r * g[0];
            cf >>= 30; cg >>= 30;
#pragma unroll
            for (int i = 1; i < L; ++i) {
                cf += (int64_t)u * f[i] + (int64_t)v * g[i];
                cg += (int64_t)q * f[i] + (int64_t)r * g[i];
                f[i - 1] = (int32_t)cf & M30; cf >>= 30;
                g[i - 1] = (int32_t)cg & M30; cg >>= 30;
            }
            f[L - 1] = (int32_t)cf; g[L - 1] = (int32_t)cg;
        }
    }
    // f = +-1 and d = +-a^-1 with the same sign: bring d from (-2p, p) to [0, p)
    {
        int32_t ca = d[L - 1] >> 31;
#pragma unroll
        for (int i = 0; i < L; ++i) d[i] += SG::mod_limb(i) & ca;
        const int32_t cn = f[L - 1] >> 31;
#pragma unroll
        for (int i = 0; i < L; ++i) d[i] = (d[i] ^ cn) - cn;
#pragma unroll
        for (int i = 0; i < L - 1; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
        ca = d[L - 1] >> 31;
#pragma unroll
        for (int i = 0; i < L; ++i) d[i] += SG::mod_limb(i) & ca;
#pragma unroll
        for (int i = 0; i < L - 1; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    }
    feT<N> out;
#pragma unroll
    for (int w = 0; w < N; ++w) {                               // 32-bit word w = bits [32w, 32w+32)
        const int lo = 32 * w / 30, sh = 32 * w % 30;           // starts in limb lo at bit sh
        uint64_t acc = (uint64_t)(uint32_t)d[lo] >> sh;
        if (lo + 1 < L) acc |= (uint64_t)(uint32_t)d[lo + 1] << (30 - sh);
        if (lo + 2 < L && 60 - sh < 32) acc |= (uint64_t)(uint32_t)d[lo + 2] << (60 - sh);
        out.v[w] = (uint32_t)acc;
    }
    return out;
}

// the inversion the kernels call: division steps, verified, Fermat chain if the check ever fails
template <int N>
__device__ __forceinline__ feT<N> fe_inv_checked(const feT<N> &x)
{
    feT<N> a, r, t, one;
    fe_canon(a, x);
    r = fe_inv_safegcd(a);
    fe_mul(t, a, r);
    fe_canon(t, t);
    fe_set_one(one);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) ok = ok && (t.v[i] == one.v[i]);
    if (!ok && !fe_is_zero(a)) r = fe_inv(a);                   // a = 0 gives 0 either way, as 0^(p-2)
    return r;
}

// ----------------------------------------------------------------- points ---
// T <- 2T, a = -3.  4M + 4S + 6 add/sub.  Z3 = 2YZ, so Y = 0 or Z = 0 gives O as in
// ecp_j.c:258-263.  (dbl-2001-b with the small multiples moved into the reductions.)
template <int N, class P = CtOps>
__device__ __forceinline__ void jac_dbl(jacT<N> &T)
{
    feT<N> delta, gamma, beta4, alpha, t0, t1;
    fe_sqr<1, P>(delta, T.Z);
    fe_sqr<1, P>(gamma, T.Y);
    fe_mul<4, P>(beta4, T.X, gamma);            // 4 X Y^2
    fe_sub<P>(t0, T.X, delta);
    fe_add<P>(t1, T.X, delta);
    fe_mul<3, P>(alpha, t0, t1);                // 3 (X - Z^2)(X + Z^2)
#ifdef BIGN_DBL_2001B
    // experiment: dbl-2001-b proper, Z3 = (Y + Z)^2 - gamma - delta: a squaring and three
    // additions instead of the multiplication 2 Y Z (3M + 5S)
    fe_add<P>(t0, T.Y, T.Z);
    fe_sqr<1, P>(t0, t0);
    fe_sub<P>(t0, t0, gamma);
    fe_sub<P>(T.Z, t0, delta);
#else
    fe_mul<2, P>(T.Z, T.Y, T.Z);                // Z3 = 2 Y Z
#endif
    fe_sqr<1, P>(t0, alpha);
    fe_dbl<P>(t1, beta4);
    fe_sub<P>(T.X, t0, t1);                     // X3 = alpha^2 - 8 beta
    fe_sqr<8, P>(t1, gamma);                    // 8 Y^4
    fe_sub<P>(t0, beta4, T.X);
    fe_mul<1, P>(t0, alpha, t0);
    fe_sub<P>(T.Y, t0, t1);                     // Y3 = alpha (4 beta - X3) - 8 Y^4
}

// T <- T + E for Jacobian E (add-1998-cmo-2, 12M + 4S).  Returns false when the generic
// formula does not apply (either operand O, or T = +-E): the caller then marks the
// signature for the complete slow path (ecp_j.c:416-427,455-464 handle these inline).
template <int N, class P = CtOps>
__device__ __forceinline__ bool jac_add(jacT<N> &T, const jacT<N> &E)
{
    feT<N> Z1Z1, Z2Z2, U1, U2, S1, S2, H, HH, HHH, r, V, t;
    const bool bad_in = fe_is_zero(T.Z) || fe_is_zero(E.Z);
    fe_sqr<1, P>(Z1Z1, T.Z);
    fe_sqr<1, P>(Z2Z2, E.Z);
    fe_mul<1, P>(U1, T.X, Z2Z2);
    fe_mul<1, P>(U2, E.X, Z1Z1);
    fe_mul<1, P>(t, E.Z, Z2Z2);  fe_mul<1, P>(S1, T.Y, t);
    fe_mul<1, P>(t, T.Z, Z1Z1);  fe_mul<1, P>(S2, E.Y, t);
    fe_sub<P>(H, U2, U1);
    const bool bad = bad_in || fe_is_zero(H);
    fe_sub<P>(r, S2, S1);
    fe_sqr<1, P>(HH, H);
    fe_mul<1, P>(HHH, H, HH);
    fe_mul<1, P>(V, U1, HH);
    fe_mul<1, P>(t, T.Z, E.Z);   fe_mul<1, P>(T.Z, t, H);           // Z3 = Z1 Z2 H
    fe_sqr<1, P>(t, r);
    fe_sub<P>(t, t, HHH);
    fe_dbl<P>(U2, V);
    fe_sub<P>(T.X, t, U2);                                 // X3 = r^2 - H^3 - 2V
    fe_sub<P>(t, V, T.X);
    fe_mul<1, P>(t, r, t);
    fe_mul<1, P>(S2, S1, HHH);
    fe_sub<P>(T.Y, t, S2);                                 // Y3 = r (V - X3) - S1 H^3
    return !bad;
}

// T <- T + E for affine E (madd, 8M + 3S); same contract as jac_add
template <int N, class P = CtOps>
__device__ __forceinline__ bool jac_madd(jacT<N> &T, const affT<N> &E)
{
    feT<N> Z1Z1, U2, S2, H, HH, HHH, r, V, t;
    const bool bad_in = fe_is_zero(T.Z);
    fe_sqr<1, P>(Z1Z1, T.Z);
    fe_mul<1, P>(U2, E.x, Z1Z1);
    fe_mul<1, P>(t, T.Z, Z1Z1);  fe_mul<1, P>(S2, E.y, t);
    fe_sub<P>(H, U2, T.X);
    const bool bad = bad_in || fe_is_zero(H);
    fe_sub<P>(r, S2, T.Y);
    fe_sqr<1, P>(HH, H);
    fe_mul<1, P>(HHH, H, HH);
    fe_mul<1, P>(V, T.X, HH);
    fe_mul<1, P>(T.Z, T.Z, H);                                // Z3 = Z1 H
    fe_sqr<1, P>(t, r);
    fe_sub<P>(t, t, HHH);
    fe_dbl<P>(U2, V);
    fe_sub<P>(T.X, t, U2);
    fe_sub<P>(t, V, T.X);
    fe_mul<1, P>(t, r, t);
    fe_mul<1, P>(S2, T.Y, HHH);
    fe_sub<P>(T.Y, t, S2);
    return !bad;
}

// complete addition (all exceptional cases, as ecpAddJ ecp_j.c:397-497): slow path only
template <int N>
__device__ __forceinline__ void jac_add_complete(jacT<N> &T, const jacT<N> &E)
{
    if (fe_is_zero(E.Z)) return;
    if (fe_is_zero(T.Z)) { T = E; return; }
    feT<N> Z1Z1, Z2Z2, U1, U2, S1, S2, t;
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
