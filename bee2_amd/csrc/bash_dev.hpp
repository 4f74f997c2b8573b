// bash_dev.hpp -- bash-f (STB 34.101.77) for one CDNA4 lane.
//
// Replaces bee2's bashF (include/bee2/crypto/bash.h:136; bodies
// src/crypto/bash/bash_f64.c:142-187, bash_favx512.c:236-249).
//
// MI355X mapping: one wavefront lane owns one 1536-bit state = 24 x u64 held as
// 48 x 32-bit VGPRs (lo/hi halves).  gfx950 has no 64-bit rotate, so rotl64 by a
// constant is two v_alignbit_b32; the three-input S-box logic maps to
// v_bitop3_b32 / v_xor3_b32.  The word permutation of each round is pure
// register renaming: rounds are unrolled in groups of 6 (the permutation has
// order 6), so after each group the words are back in canonical slots and no
// v_mov is ever issued.  Round constants come from the LFSR recurrence
// (bash_f64.c:50-59) evaluated on the scalar unit (wave-uniform).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bee2hip {

struct u64x2 { uint32_t lo, hi; };

__device__ __forceinline__ u64x2 rotl64c(u64x2 x, int n)
{
    // n is a compile-time constant in 1..63, never 32 on this path
    u64x2 r;
    if (n < 32) {
        r.lo = __builtin_amdgcn_alignbit(x.lo, x.hi, 32 - n);   // (lo << n) | (hi >> (32-n))
        r.hi = __builtin_amdgcn_alignbit(x.hi, x.lo, 32 - n);
    } else if (n == 32) {
        r.lo = x.hi; r.hi = x.lo;
    } else {
        r.lo = __builtin_amdgcn_alignbit(x.hi, x.lo, 64 - n);
        r.hi = __builtin_amdgcn_alignbit(x.lo, x.hi, 64 - n);
    }
    return r;
}

// 3-input boolean function by truth table (v_bitop3_b32); table = f(0xF0, 0xCC, 0xAA)
template <int TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
constexpr int TT_XOR3 = 0xF0 ^ 0xCC ^ 0xAA;                       // a ^ b ^ c
constexpr int TT_S0 = 0xF0 ^ ((~0xAA & 0xFF) | 0xCC);             // a ^ (~c | b)
constexpr int TT_S1 = 0xCC ^ (0xF0 | 0xAA);                       // b ^ (a | c)
constexpr int TT_S2 = 0xAA ^ (0xF0 & 0xCC);                       // c ^ (a & b)

// column S-box, bash_f64.c:32-44 / bash_favx512.c:124-132.
// 8 v_alignbit + 4 v_xor + 10 v_bitop3 = 22 VALU ops per column.
template <int M1, int N1, int M2, int N2>
__device__ __forceinline__ void bash_s(u64x2 &w0, u64x2 &w1, u64x2 &w2)
{
    u64x2 u0, t, u1, u2, r, r2;
    u0.lo = bitop3<TT_XOR3>(w0.lo, w1.lo, w2.lo);
    u0.hi = bitop3<TT_XOR3>(w0.hi, w1.hi, w2.hi);
    r = rotl64c(u0, N1);  t.lo = w1.lo ^ r.lo;  t.hi = w1.hi ^ r.hi;
    r = rotl64c(w0, M1);  u1.lo = t.lo ^ r.lo;  u1.hi = t.hi ^ r.hi;
    r = rotl64c(w2, M2);
    r2 = rotl64c(t, N2);
    u2.lo = bitop3<TT_XOR3>(w2.lo, r.lo, r2.lo);
    u2.hi = bitop3<TT_XOR3>(w2.hi, r.hi, r2.hi);
    w0.lo = bitop3<TT_S0>(u0.lo, u1.lo, u2.lo);  w0.hi = bitop3<TT_S0>(u0.hi, u1.hi, u2.hi);
    w1.lo = bitop3<TT_S1>(u0.lo, u1.lo, u2.lo);  w1.hi = bitop3<TT_S1>(u0.hi, u1.hi, u2.hi);
    w2.lo = bitop3<TT_S2>(u0.lo, u1.lo, u2.lo);  w2.hi = bitop3<TT_S2>(u0.hi, u1.hi, u2.hi);
}

// S-layer over the 8 columns; row r of the 3x8 matrix is s[8r .. 8r+7]
__device__ __forceinline__ void bash_s_layer(u64x2 (&a)[24], const int (&ix)[24])
{
    // ix[k] = register slot currently holding logical word k
    bash_s< 8, 53, 14,  1>(a[ix[0]], a[ix[ 8]], a[ix[16]]);
    bash_s<56, 51, 34,  7>(a[ix[1]], a[ix[ 9]], a[ix[17]]);
    bash_s< 8, 37, 46, 49>(a[ix[2]], a[ix[10]], a[ix[18]]);
    bash_s<56,  3,  2, 23>(a[ix[3]], a[ix[11]], a[ix[19]]);
    bash_s< 8, 21, 14, 33>(a[ix[4]], a[ix[12]], a[ix[20]]);
    bash_s<56, 19, 34, 39>(a[ix[5]], a[ix[13]], a[ix[21]]);
    bash_s< 8,  5, 46, 17>(a[ix[6]], a[ix[14]], a[ix[22]]);
    bash_s<56, 35,  2, 55>(a[ix[7]], a[ix[15]], a[ix[23]]);
}

// ---- staged S-layer -------------------------------------------------------------------------
// The same 22 ops per column as bash_s, but issued stage by stage across all 8 columns, every op a
// volatile asm so the issue order is exactly the written one: per round X16 A48 X32 A16 X64
// (X = full-rate v_bitop3/v_xor, A = half-rate v_alignbit).  Left to itself the compiler
// interleaves the two classes almost one for one (290 class switches per 6 rounds; staged: 24) and
// puts consumers right behind their producers; gfx950 issues that order 3 % slower on the whole
// bashF kernel (profiles/r01_valu_rates_ubench.txt "class-switch cost"; A/B log profiles/r01_bashF_ab_staged.txt).
// Costs ~40 more live VGPRs (113 in bashF_batch_kernel).  The fused hash+MAC kernel, capped at 128
// VGPRs by its 1024-lane workgroups, spills 48-112 B per lane with it and is still 6 % faster
// (profiles/r01_fused_ab_staged.txt: the spills are outside the round loop); with 768-lane
// workgroups (no spills, 3 wavefronts per SIMD instead of 4) it loses 1 %.
template <int TT> __device__ __forceinline__ uint32_t vbitop3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(r) : "v"(a), "v"(b), "v"(c), "n"(TT));
    return r;
}
__device__ __forceinline__ uint32_t vxor(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int SH> __device__ __forceinline__ uint32_t valign(uint32_t hi, uint32_t lo)
{
    uint32_t r;
    asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "n"(SH));
    return r;
}
template <int N> __device__ __forceinline__ u64x2 vrotl64(u64x2 x)
{
    static_assert(N > 0 && N < 64 && N != 32, "rotation amount");
    u64x2 r;
    if constexpr (N < 32) { r.lo = valign<32 - N>(x.lo, x.hi); r.hi = valign<32 - N>(x.hi, x.lo); }
    else                  { r.lo = valign<64 - N>(x.hi, x.lo); r.hi = valign<64 - N>(x.lo, x.hi); }
    return r;
}
template <int C> struct BashRotC;
template <> struct BashRotC<0> { static constexpr int m1 =  8, n1 = 53, m2 = 14, n2 =  1; };
template <> struct BashRotC<1> { static constexpr int m1 = 56, n1 = 51, m2 = 34, n2 =  7; };
template <> struct BashRotC<2> { static constexpr int m1 =  8, n1 = 37, m2 = 46, n2 = 49; };
template <> struct BashRotC<3> { static constexpr int m1 = 56, n1 =  3, m2 =  2, n2 = 23; };
template <> struct BashRotC<4> { static constexpr int m1 =  8, n1 = 21, m2 = 14, n2 = 33; };
template <> struct BashRotC<5> { static constexpr int m1 = 56, n1 = 19, m2 = 34, n2 = 39; };
template <> struct BashRotC<6> { static constexpr int m1 =  8, n1 =  5, m2 = 46, n2 = 17; };
template <> struct BashRotC<7> { static constexpr int m1 = 56, n1 = 35, m2 =  2, n2 = 55; };

// ---- issue priority follows the instruction class (r02) ---------------------------------------------
// gfx950 arbitrates VALU issue oldest-wavefront-first.  A half-rate op (v_alignbit_b32) holds its unit for 4
// cycles but needs one issue slot; full-rate ops of OTHER wavefronts can issue while it runs -- if the half-rate
// op gets its slot in time.  With every wavefront at the same priority the older wavefronts' full-rate runs
// take the slots, the half-rate unit idles, and mixed code runs 15 % slower than the two classes one after the
// other.  s_setprio 3 for the half-rate runs and 0 for the full-rate runs: the bash round pattern
// F80 H48 F32 H16 goes from 3.63 to 2.18 cycles per instruction per SIMD (tools/ubench/valu_overlap.hip,
// profiles/r02_valu_overlap.txt; raising the priority of the FULL-rate runs instead: 3.69).
__device__ __forceinline__ void bash_prio_half() { __builtin_amdgcn_s_setprio(3); }
__device__ __forceinline__ void bash_prio_full() { __builtin_amdgcn_s_setprio(0); }

struct BashStage { u64x2 u0[8], ra[8], rb[8], rc[8], t[8], u1[8], r2[8]; };
template <int C> __device__ __forceinline__ void st_rot3(BashStage &q, u64x2 (&a)[24], const int (&ix)[24])
{
    q.ra[C] = vrotl64<BashRotC<C>::n1>(q.u0[C]);
    q.rb[C] = vrotl64<BashRotC<C>::m1>(a[ix[C]]);
    q.rc[C] = vrotl64<BashRotC<C>::m2>(a[ix[16 + C]]);
}
template <int C> __device__ __forceinline__ void st_rot1(BashStage &q) { q.r2[C] = vrotl64<BashRotC<C>::n2>(q.t[C]); }

// PRIO: raise the wavefront's issue priority for the half-rate runs (see bash_prio_half / bash_prio_full below)
template <bool PRIO = false>
__device__ __forceinline__ void bash_s_layer_staged(u64x2 (&a)[24], const int (&ix)[24])
{
    BashStage q;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        q.u0[c].lo = vbitop3<TT_XOR3>(a[ix[c]].lo, a[ix[8 + c]].lo, a[ix[16 + c]].lo);
        q.u0[c].hi = vbitop3<TT_XOR3>(a[ix[c]].hi, a[ix[8 + c]].hi, a[ix[16 + c]].hi);
    }
    if constexpr (PRIO) bash_prio_half();
    st_rot3<0>(q, a, ix); st_rot3<1>(q, a, ix); st_rot3<2>(q, a, ix); st_rot3<3>(q, a, ix);
    st_rot3<4>(q, a, ix); st_rot3<5>(q, a, ix); st_rot3<6>(q, a, ix); st_rot3<7>(q, a, ix);
    if constexpr (PRIO) bash_prio_full();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        q.t[c].lo = vxor(a[ix[8 + c]].lo, q.ra[c].lo);  q.t[c].hi = vxor(a[ix[8 + c]].hi, q.ra[c].hi);
        q.u1[c].lo = vxor(q.t[c].lo, q.rb[c].lo);       q.u1[c].hi = vxor(q.t[c].hi, q.rb[c].hi);
    }
    if constexpr (PRIO) bash_prio_half();
    st_rot1<0>(q); st_rot1<1>(q); st_rot1<2>(q); st_rot1<3>(q); st_rot1<4>(q); st_rot1<5>(q); st_rot1<6>(q); st_rot1<7>(q);
    if constexpr (PRIO) bash_prio_full();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        u64x2 u2;
        u2.lo = vbitop3<TT_XOR3>(a[ix[16 + c]].lo, q.rc[c].lo, q.r2[c].lo);
        u2.hi = vbitop3<TT_XOR3>(a[ix[16 + c]].hi, q.rc[c].hi, q.r2[c].hi);
        a[ix[c]].lo = vbitop3<TT_S0>(q.u0[c].lo, q.u1[c].lo, u2.lo);       a[ix[c]].hi = vbitop3<TT_S0>(q.u0[c].hi, q.u1[c].hi, u2.hi);
        a[ix[8 + c]].lo = vbitop3<TT_S1>(q.u0[c].lo, q.u1[c].lo, u2.lo);   a[ix[8 + c]].hi = vbitop3<TT_S1>(q.u0[c].hi, q.u1[c].hi, u2.hi);
        a[ix[16 + c]].lo = vbitop3<TT_S2>(q.u0[c].lo, q.u1[c].lo, u2.lo);  a[ix[16 + c]].hi = vbitop3<TT_S2>(q.u0[c].hi, q.u1[c].hi, u2.hi);
    }
}

// ---- staged S-layer, second form (r02) ----------------------------------------------------------
// Same stages, but (i) inside a stage no instruction reads the result of its predecessor (the first form
// computed t and u1 = t ^ rb back to back, and u2 right before the three S-box outputs that need it: hipcc
// pads every such asm -> asm dependency with an s_nop, 81 of them per 6 rounds, and the wavefront stalls on
// the VALU latency anyway), and (ii) the stage width W is a parameter: W = 8 is one pass over all columns
// (temporaries for 8 columns live), W = 4 two passes over 4 columns each (half the temporaries, shorter
// runs of one instruction class).
template <int C0, int W, int K = 0>
__device__ __forceinline__ void st2_rot3(u64x2 (&ra)[8], u64x2 (&rb)[8], u64x2 (&rc)[8], const u64x2 (&u0)[8],
                                         u64x2 (&a)[24], const int (&ix)[24])
{
    if constexpr (K < W) {
        constexpr int C = C0 + K;
        ra[C] = vrotl64<BashRotC<C>::n1>(u0[C]);
        rb[C] = vrotl64<BashRotC<C>::m1>(a[ix[C]]);
        rc[C] = vrotl64<BashRotC<C>::m2>(a[ix[16 + C]]);
        st2_rot3<C0, W, K + 1>(ra, rb, rc, u0, a, ix);
    }
}
template <int C0, int W, int K = 0>
__device__ __forceinline__ void st2_rot1(u64x2 (&r2)[8], const u64x2 (&t)[8])
{
    if constexpr (K < W) {
        r2[C0 + K] = vrotl64<BashRotC<C0 + K>::n2>(t[C0 + K]);
        st2_rot1<C0, W, K + 1>(r2, t);
    }
}
template <int C0, int W, bool PRIO>
__device__ __forceinline__ void bash_s_cols_staged2(u64x2 (&a)[24], const int (&ix)[24])
{
    u64x2 u0[8], ra[8], rb[8], rc[8], t[8], u1[8], r2[8], u2[8];
#pragma unroll
    for (int c = C0; c < C0 + W; ++c) {
        u0[c].lo = vbitop3<TT_XOR3>(a[ix[c]].lo, a[ix[8 + c]].lo, a[ix[16 + c]].lo);
        u0[c].hi = vbitop3<TT_XOR3>(a[ix[c]].hi, a[ix[8 + c]].hi, a[ix[16 + c]].hi);
    }
    if constexpr (PRIO) bash_prio_half();
    st2_rot3<C0, W>(ra, rb, rc, u0, a, ix);
    if constexpr (PRIO) bash_prio_full();
#pragma unroll
    for (int c = C0; c < C0 + W; ++c) { t[c].lo = vxor(a[ix[8 + c]].lo, ra[c].lo); t[c].hi = vxor(a[ix[8 + c]].hi, ra[c].hi); }
#pragma unroll
    for (int c = C0; c < C0 + W; ++c) { u1[c].lo = vxor(t[c].lo, rb[c].lo); u1[c].hi = vxor(t[c].hi, rb[c].hi); }
    if constexpr (PRIO) bash_prio_half();
    st2_rot1<C0, W>(r2, t);
    if constexpr (PRIO) bash_prio_full();
#pragma unroll
    for (int c = C0; c < C0 + W; ++c) {
        u2[c].lo = vbitop3<TT_XOR3>(a[ix[16 + c]].lo, rc[c].lo, r2[c].lo);
        u2[c].hi = vbitop3<TT_XOR3>(a[ix[16 + c]].hi, rc[c].hi, r2[c].hi);
    }
#pragma unroll
    for (int c = C0; c < C0 + W; ++c) {
        a[ix[c]].lo = vbitop3<TT_S0>(u0[c].lo, u1[c].lo, u2[c].lo);       a[ix[c]].hi = vbitop3<TT_S0>(u0[c].hi, u1[c].hi, u2[c].hi);
        a[ix[8 + c]].lo = vbitop3<TT_S1>(u0[c].lo, u1[c].lo, u2[c].lo);   a[ix[8 + c]].hi = vbitop3<TT_S1>(u0[c].hi, u1[c].hi, u2[c].hi);
        a[ix[16 + c]].lo = vbitop3<TT_S2>(u0[c].lo, u1[c].lo, u2[c].lo);  a[ix[16 + c]].hi = vbitop3<TT_S2>(u0[c].hi, u1[c].hi, u2[c].hi);
    }
}
template <int W, bool PRIO = false>
__device__ __forceinline__ void bash_s_layer_staged2(u64x2 (&a)[24], const int (&ix)[24])
{
    if constexpr (W == 8) bash_s_cols_staged2<0, 8, PRIO>(a, ix);
    else if constexpr (W == 4) { bash_s_cols_staged2<0, 4, PRIO>(a, ix); bash_s_cols_staged2<4, 4, PRIO>(a, ix); }
    else { bash_s_cols_staged2<0, 2, PRIO>(a, ix); bash_s_cols_staged2<2, 2, PRIO>(a, ix); bash_s_cols_staged2<4, 2, PRIO>(a, ix); bash_s_cols_staged2<6, 2, PRIO>(a, ix); }
}

// logical word k of the next round = logical word BASH_PERM[k] of this round:
// new_row0 = pi1(row1), new_row1 = pi2(row2), new_row2 = pi0(row0)
// (bash_f64.c:100-134 "P1", explicit in bash_favx512.c:140-171)
struct BashPerm {
    int v[24];
    constexpr BashPerm() : v{} {
        constexpr int pi0[8] = {6, 3, 0, 5, 2, 7, 4, 1};
        constexpr int pi1[8] = {7, 2, 1, 4, 3, 6, 5, 0};
        constexpr int pi2[8] = {1, 0, 3, 2, 5, 4, 7, 6};
        for (int k = 0; k < 8; ++k) {
            v[k] = 8 + pi1[k];
            v[8 + k] = 16 + pi2[k];
            v[16 + k] = pi0[k];
        }
    }
};

// slot map after r rounds (r = 0..6); map[6] is the identity again
struct BashSlots {
    int m[7][24];
    constexpr BashSlots() : m{} {
        constexpr BashPerm P{};
        for (int k = 0; k < 24; ++k) m[0][k] = k;
        for (int r = 1; r < 7; ++r)
            for (int k = 0; k < 24; ++k) m[r][k] = m[r - 1][P.v[k]];
    }
};

__device__ __forceinline__ uint64_t bash_next_const(uint64_t c)
{
    return (c >> 1) ^ (0xDC2BE1997FE0D8AEull & (0ull - (c & 1ull)));
}

// ORDER: 0 = compiler's order ("compact"), 1 = staged (r01), 28 / 24 / 22 = staged, second form, W = 8 / 4 / 2;
// +100: issue priority follows the instruction class
template <int R, int ORDER>
__device__ __forceinline__ void bash_round(u64x2 (&a)[24], uint64_t &c)
{
    constexpr BashSlots S{};
    if constexpr (ORDER == 1)        bash_s_layer_staged<false>(a, S.m[R]);
    else if constexpr (ORDER == 101) bash_s_layer_staged<true>(a, S.m[R]);
    else if constexpr (ORDER == 28)  bash_s_layer_staged2<8>(a, S.m[R]);
    else if constexpr (ORDER == 24)  bash_s_layer_staged2<4>(a, S.m[R]);
    else if constexpr (ORDER == 22)  bash_s_layer_staged2<2>(a, S.m[R]);
    else if constexpr (ORDER == 128) bash_s_layer_staged2<8, true>(a, S.m[R]);
    else if constexpr (ORDER == 124) bash_s_layer_staged2<4, true>(a, S.m[R]);
    else if constexpr (ORDER == 122) bash_s_layer_staged2<2, true>(a, S.m[R]);
    else                            bash_s_layer(a, S.m[R]);
    // after the word permutation the constant lands on logical word 23 of the next round
    constexpr int slot = S.m[R + 1][23];
    a[slot].lo ^= (uint32_t)c;
    a[slot].hi ^= (uint32_t)(c >> 32);
    c = bash_next_const(c);
}

// the permutation: 4 x 6 rounds.  STAGED picks the issue order of the S-layer (see above).
template <int STAGED = 0>
__device__ __forceinline__ void bash_f(u64x2 (&a)[24])
{
    uint64_t c = 0x3BF5080AC8BA94B1ull;
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        bash_round<0, STAGED>(a, c);
        bash_round<1, STAGED>(a, c);
        bash_round<2, STAGED>(a, c);
        bash_round<3, STAGED>(a, c);
        bash_round<4, STAGED>(a, c);
        bash_round<5, STAGED>(a, c);
    }
}

// ---- one state across 8 lanes, one column each ------------------------------------------------
// For a LONG message the sponge is a serial chain of permutations and one lane per state runs it at a
// lone wavefront's issue rate.  Here lane j = lane & 7 of a group of 8 holds column j
// (w0, w1, w2 = words j, 8 + j, 16 + j): the S-box is column-local, and the word permutation of a round
// (bash_favx512.c:140-171) is an exchange inside the group: new w0 <- w1 of lane pi1[j], new w1 <- w2
// of lane pi2[j], new w2 <- w0 of lane pi0[j] (six ds_bpermute_b32).  The rotation amounts differ per
// lane; a rotation by n >= 32 is a swap of the halves followed by a rotation by n - 32, so every
// rotation is two v_bitop3 selects with a per-lane mask and two v_alignbit with a per-lane shift
// (no amount is a multiple of 32).  36 instructions per lane per round instead of 22: worse for
// throughput, ~4x shorter for one message.
struct BashCol {
    uint32_t sh[4], mk[4];            // for m1, n1, m2, n2: alignbit shift 32 - (n & 31), all-ones if n >= 32
    int a0, a1, a2;                   // ds_bpermute byte addresses: lanes pi0[j], pi1[j], pi2[j] of my group
    uint32_t last;                    // all-ones in lane 7 of the group: word 23 takes the round constant
};
__device__ inline BashCol bash_col_setup(unsigned lane)
{
    // column j: (8, 53, 14, 1) * 7^j mod 64  (bash_f64.c:126-133)
    const unsigned j = lane & 7u;
    unsigned m = 1;
    for (unsigned k = 0; k < j; ++k) m = (m * 7u) & 63u;
    const unsigned par[4] = {(8u * m) & 63u, (53u * m) & 63u, (14u * m) & 63u, (1u * m) & 63u};
    BashCol c;
    for (int k = 0; k < 4; ++k) { c.sh[k] = 32u - (par[k] & 31u); c.mk[k] = par[k] >= 32u ? ~0u : 0u; }
    const unsigned pi0 = (0x14725036u >> (4 * j)) & 7u;       // (6,3,0,5,2,7,4,1), digit j
    const unsigned pi1 = (0x05634127u >> (4 * j)) & 7u;       // (7,2,1,4,3,6,5,0)
    const unsigned pi2 = (0x67452301u >> (4 * j)) & 7u;       // (1,0,3,2,5,4,7,6)
    const unsigned base = lane & ~7u;
    c.a0 = (int)((base | pi0) << 2); c.a1 = (int)((base | pi1) << 2); c.a2 = (int)((base | pi2) << 2);
    c.last = j == 7u ? ~0u : 0u;
    return c;
}
constexpr int TT_SEL = 0xD8;                                   // c ? b : a
template <int P>
__device__ __forceinline__ u64x2 bash_col_rot(const u64x2 x, const BashCol &c)
{
    const uint32_t a = bitop3<TT_SEL>(x.lo, x.hi, c.mk[P]), b = bitop3<TT_SEL>(x.hi, x.lo, c.mk[P]);
    u64x2 r;
    r.lo = __builtin_amdgcn_alignbit(a, b, c.sh[P]);
    r.hi = __builtin_amdgcn_alignbit(b, a, c.sh[P]);
    return r;
}
// (round 4, second session) A chain of permutations is walked by a LONE wavefront (one issue slot per 4 cycles, LDS round trip ~56
// cycles: tools/ubench/lone_chain.hip), so the round is written for its slot count and ONE exposed round trip: the 24 rounds are
// unrolled with their constants as literals (the rolled loop spent 13 of its 55 slots on the LFSR, its copy into lane 7 and the loop
// itself), and the ds_bpermute of a round are issued back to back (the rolled loop waited for the first three before it sent the
// others: two round trips per round).
constexpr uint64_t bash_round_const(int r)
{
    uint64_t c = 0x3BF5080AC8BA94B1ull;
    for (int i = 0; i < r; ++i) c = (c >> 1) ^ (0xDC2BE1997FE0D8AEull & (0ull - (c & 1ull)));
    return c;
}
template <int R>
__device__ __forceinline__ void bash_round_cols(u64x2 &w0, u64x2 &w1, u64x2 &w2, const BashCol &c)
{
    constexpr uint64_t rc = bash_round_const(R);
    u64x2 u0, t, u1, u2, r, r2;
    u0.lo = bitop3<TT_XOR3>(w0.lo, w1.lo, w2.lo); u0.hi = bitop3<TT_XOR3>(w0.hi, w1.hi, w2.hi);
    r = bash_col_rot<1>(u0, c);  t.lo = w1.lo ^ r.lo;  t.hi = w1.hi ^ r.hi;
    r = bash_col_rot<0>(w0, c);  u1.lo = t.lo ^ r.lo;  u1.hi = t.hi ^ r.hi;
    r = bash_col_rot<2>(w2, c);
    r2 = bash_col_rot<3>(t, c);
    u2.lo = bitop3<TT_XOR3>(w2.lo, r.lo, r2.lo); u2.hi = bitop3<TT_XOR3>(w2.hi, r.hi, r2.hi);
    const uint32_t s0l = bitop3<TT_S0>(u0.lo, u1.lo, u2.lo), s0h = bitop3<TT_S0>(u0.hi, u1.hi, u2.hi);
    const uint32_t s1l = bitop3<TT_S1>(u0.lo, u1.lo, u2.lo), s1h = bitop3<TT_S1>(u0.hi, u1.hi, u2.hi);
    const uint32_t p1l = (uint32_t)__builtin_amdgcn_ds_bpermute(c.a1, (int)s1l), p1h = (uint32_t)__builtin_amdgcn_ds_bpermute(c.a1, (int)s1h);
    const uint32_t p0l = (uint32_t)__builtin_amdgcn_ds_bpermute(c.a0, (int)s0l), p0h = (uint32_t)__builtin_amdgcn_ds_bpermute(c.a0, (int)s0h);
    // pi2 = (1,0,3,2,5,4,7,6) is a DPP quad_perm: computed and moved WHILE the four ds_bpermute are on their round trip (its wait
    // states cost nothing there), which also takes two of six operations off the LDS crossbar
    const uint32_t s2l = bitop3<TT_S2>(u0.lo, u1.lo, u2.lo), s2h = bitop3<TT_S2>(u0.hi, u1.hi, u2.hi);
    w1.lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s2l, 0xB1, 0xF, 0xF, true);
    w1.hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s2h, 0xB1, 0xF, 0xF, true);
    w0.lo = p1l; w0.hi = p1h;
    w2.lo = p0l ^ ((uint32_t)rc & c.last);
    w2.hi = p0h ^ ((uint32_t)(rc >> 32) & c.last);
}
template <int R = 0>
__device__ __forceinline__ void bash_f_cols(u64x2 &w0, u64x2 &w1, u64x2 &w2, const BashCol &c)
{
    bash_round_cols<R>(w0, w1, w2, c);
    if constexpr (R + 1 < 24) bash_f_cols<R + 1>(w0, w1, w2, c);
}

}  // namespace bee2hip
