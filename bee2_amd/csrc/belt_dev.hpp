// belt_dev.hpp -- belt block encryption (STB 34.101.31) for one CDNA4 lane.
//
// Replaces bee2's G-boxes / round macro / E macro and beltBlockEncr2
// (src/crypto/belt/belt_block.c:121-269, 323-327) and the S-box H (:43-60).
//
// MI355X mapping.  A G-box is four byte-indexed table lookups; on a CPU those are
// L1 hits, on CDNA4 they are LDS reads and the LDS -- not HBM, not the VALU -- is
// the binding unit (224 ds_read_b32 per 16-byte block).  Two table layouts:
//
//   BeltTabWide   4 pre-rotated tables (rotl 5/13/21/29 of the S-box byte, as in
//                 belt_block.c:121-195) x 256 entries x 32 bank-private copies
//                 = 128 KiB of the CU's 160 KiB LDS.  Lane l only ever touches bank
//                 (l & 31), and ds_read_b32 services lanes 0-31 / 32-63 as separate
//                 groups, so every lookup is conflict-free: 2 LDS cycles per
//                 wave-instruction instead of ~7 with a shared 1 KiB table
//                 (SQ_LDS_BANK_CONFLICT = 0 measured, profiles/r01_pmc_summary.json).
//   BeltTabSmall  the same 4 tables without replication (4 KiB) for kernels where
//                 belt is a sliver of the work (bign verify tail, single blocks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bee2hip {

// the belt S-box, generated at library load by the standard's LFSR recipe
// (belt_block.c:21-35) -- see capi.cpp -- and uploaded once per device.  The library is two translation units without
// relocatable device code (bee2hip_tu_belt.hip, bee2hip_tu_bign.hip: they compile side by side), so each holds its own
// 256-byte copy: upload_beltH() fills the one of the TU it is compiled in, upload_beltH_bign() the other.
static __constant__ uint8_t c_beltH[256];

__device__ __forceinline__ uint32_t rotl32c(uint32_t x, int r)
{
    return __builtin_amdgcn_alignbit(x, x, 32 - r);
}

// (a & b) | c in one full-rate op
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, (0xF0 & 0xCC) | 0xAA);
}
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xF0 ^ 0xCC ^ 0xAA);
}

// opaque shifts: written as asm so that LLVM's demanded-bits logic cannot rewrite
// "(x >> 7) & 0x1FE00" into v_bfe_u32 + shift or an SDWA form (both half rate)
template <int N>
__device__ __forceinline__ uint32_t shl_c(uint32_t x)
{
    uint32_t r;
    asm("v_lshlrev_b32 %0, %1, %2" : "=v"(r) : "n"(N), "v"(x));
    return r;
}
template <int N>
__device__ __forceinline__ uint32_t shr_c(uint32_t x)
{
    uint32_t r;
    asm("v_lshrrev_b32 %0, %1, %2" : "=v"(r) : "n"(N), "v"(x));
    return r;
}

// the four table values of one G-box; the consumer folds them with v_bitop3 (xor3)
struct GParts { uint32_t p, q; };          // G = p ^ q, p already = t0 ^ t1 ^ t2

// experiment hook (r02): raise the wavefront's priority while it issues the four table reads of a G-box
#ifdef BELT_LDS_PRIO
__device__ __forceinline__ void belt_prio_lds() { __builtin_amdgcn_s_setprio(BELT_LDS_PRIO); }
__device__ __forceinline__ void belt_prio_valu() { __builtin_amdgcn_s_setprio(0); }
#else
__device__ __forceinline__ void belt_prio_lds() {}
__device__ __forceinline__ void belt_prio_valu() {}
#endif

struct BeltTabWide {
    // dword index = byte*128 + R*32 + bank: the table selector R sits in the instruction's
    // immediate offset (R*128 bytes), the byte in address bits [16:9], the bank-private copy
    // in bits [6:2].  Address = (x_shifted & 0x1FE00) | bank4 is ONE v_bitop3 (full rate) after
    // ONE shift, instead of v_bfe + v_lshl_add (both half rate on gfx950).
    static constexpr int kBytes = 256 * 4 * 32 * 4;        // 131072
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    uint32_t base;          // LDS byte address of the table + (lane & 31) * 4
    // fill from every thread of the workgroup; caller must __syncthreads() afterwards
    __device__ static void fill(uint8_t *lds, int tid, int nthreads)
    {
        uint32_t *t = reinterpret_cast<uint32_t *>(lds);
        for (int i = tid; i < 256 * 4 * 32; i += nthreads) {
            const int r = (i >> 5) & 3, idx = i >> 7;
            t[i] = rotl32c((uint32_t)c_beltH[idx], 5 + 8 * r);
        }
    }
    __device__ explicit BeltTabWide(const uint8_t *l)
    {
        const uint32_t tab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)l;
        // the OR-composed address needs the table 128 KiB aligned in LDS, i.e. at LDS address 0
        // (dynamic LDS with no static __shared__ in the kernel)
        if (tab & (kBytes - 1)) __builtin_trap();
        base = tab + ((threadIdx.x & 31) << 2);
    }
    template <int R0>
    __device__ __forceinline__ GParts g(uint32_t x) const
    {
        constexpr uint32_t M = 0x1FE00u;
        const uint32_t a0 = and_or(shl_c<9>(x), M, base), a1 = and_or(shl_c<1>(x), M, base);
        const uint32_t a2 = and_or(shr_c<7>(x), M, base), a3 = and_or(shr_c<15>(x), M, base);
        belt_prio_lds();
        const uint32_t t0 = *(lds_u32 *)(uintptr_t)(a0 + ((R0 + 0) & 3) * 128);
        const uint32_t t1 = *(lds_u32 *)(uintptr_t)(a1 + ((R0 + 1) & 3) * 128);
        const uint32_t t2 = *(lds_u32 *)(uintptr_t)(a2 + ((R0 + 2) & 3) * 128);
        const uint32_t t3 = *(lds_u32 *)(uintptr_t)(a3 + ((R0 + 3) & 3) * 128);
        belt_prio_valu();
        GParts r;
        r.p = xor3(t0, t1, t2);
        r.q = t3;
        return r;
    }
};

// Two-table variant, 64 KiB: T5[b] = S(b) << 5 and T29[b] = rotl(S(b), 29).  S(b) << 5
// spans bits 5..12, so the rotations by 13 and 21 are plain left shifts of the same entry
// (no wrap-around); only the byte that lands on bits 29..36 needs the wrapped table:
//   G5  = T5[b0] ^ T5[b1]<<8  ^ T5[b2]<<16 ^ T29[b3]
//   G13 = T5[b0]<<8 ^ T5[b1]<<16 ^ T29[b2] ^ T5[b3]
//   G21 = T5[b0]<<16 ^ T29[b1] ^ T5[b2]    ^ T5[b3]<<8
// Two extra full-rate shifts per G-box buy a 64 KiB footprint, i.e. TWO 1024-thread
// workgroups (8 wavefronts per SIMD) per CU -- a single wavefront can issue a VALU op only
// every ~6 cycles on gfx950 (tools/ubench), so occupancy, not instruction count, was what
// capped the 128 KiB variant.
struct BeltTabTwo {
    // dword index = byte*64 + t*32 + bank  (t = 0: T5, t = 1: T29): byte in address bits
    // [15:8], table select in the immediate offset (t*128), bank copy in bits [6:2]
    static constexpr int kBytes = 256 * 2 * 32 * 4;        // 65536
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    uint32_t base;
    __device__ static void fill(uint8_t *lds, int tid, int nthreads)
    {
        uint32_t *t = reinterpret_cast<uint32_t *>(lds);
        for (int i = tid; i < 256 * 2 * 32; i += nthreads) {
            const int sel = (i >> 5) & 1, idx = i >> 6;
            const uint32_t sv = c_beltH[idx];
            t[i] = sel ? rotl32c(sv, 29) : (sv << 5);
        }
    }
    __device__ explicit BeltTabTwo(const uint8_t *l)
    {
        const uint32_t tab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)l;
        if (tab & (kBytes - 1)) __builtin_trap();          // OR-composed addresses need 64 KiB alignment
        base = tab + ((threadIdx.x & 31) << 2);
    }
    template <int R0>
    __device__ __forceinline__ GParts g(uint32_t x) const
    {
        constexpr uint32_t M = 0xFF00u;
        const uint32_t a0 = and_or(shl_c<8>(x), M, base), a1 = and_or(x, M, base);
        const uint32_t a2 = and_or(shr_c<8>(x), M, base), a3 = and_or(shr_c<16>(x), M, base);
        // byte k gets rotation 5 + 8*((R0 + k) & 3): 29 -> wrapped table, else T5 << 8*((R0+k)&3)
        constexpr int r0 = (R0 + 0) & 3, r1 = (R0 + 1) & 3, r2 = (R0 + 2) & 3, r3 = (R0 + 3) & 3;
        belt_prio_lds();
        uint32_t t0 = *(lds_u32 *)(uintptr_t)(a0 + (r0 == 3 ? 128 : 0));
        uint32_t t1 = *(lds_u32 *)(uintptr_t)(a1 + (r1 == 3 ? 128 : 0));
        uint32_t t2 = *(lds_u32 *)(uintptr_t)(a2 + (r2 == 3 ? 128 : 0));
        uint32_t t3 = *(lds_u32 *)(uintptr_t)(a3 + (r3 == 3 ? 128 : 0));
        belt_prio_valu();
        if (r0 == 1 || r0 == 2) t0 <<= 8 * r0;
        if (r1 == 1 || r1 == 2) t1 <<= 8 * r1;
        if (r2 == 1 || r2 == 2) t2 <<= 8 * r2;
        if (r3 == 1 || r3 == 2) t3 <<= 8 * r3;
        GParts r;
        r.p = xor3(t0, t1, t2);
        r.q = t3;
        return r;
    }
};

// EXPERIMENT (round 3; tools/ab/belt_ab.py variants 10 / 11, profiles/r03_belt_hybrid.txt): the two-table
// LDS layout for most G-boxes, and every G-box for which `via_l1<R0, SLOT>()` says so looked up through the vector L1
// instead -- four rotated tables of 256 dwords in global memory (4 KiB, L1-resident), one global_load_dword per byte
// with the table selector in the immediate offset -- so that TA/TCP cycles run beside the LDS pipe.  Not the product.
extern __device__ uint32_t d_beltT4[1024];
template <int ROUNDS>                  // bit i-1 set: the last G-box of round i goes through the L1
struct BeltTabHyb : BeltTabTwo {
    __device__ explicit BeltTabHyb(const uint8_t *l) : BeltTabTwo(l) {}
    template <int R0>
    __device__ __forceinline__ GParts g_l1(uint32_t x) const
    {
        const uint32_t o0 = shl_c<2>(x) & 0x3FCu, o1 = shr_c<6>(x) & 0x3FCu, o2 = shr_c<14>(x) & 0x3FCu, o3 = shr_c<22>(x) & 0x3FCu;
        const uint8_t *t = reinterpret_cast<const uint8_t *>(d_beltT4);
        const uint32_t t0 = *reinterpret_cast<const uint32_t *>(t + ((R0 + 0) & 3) * 1024 + o0);
        const uint32_t t1 = *reinterpret_cast<const uint32_t *>(t + ((R0 + 1) & 3) * 1024 + o1);
        const uint32_t t2 = *reinterpret_cast<const uint32_t *>(t + ((R0 + 2) & 3) * 1024 + o2);
        const uint32_t t3 = *reinterpret_cast<const uint32_t *>(t + ((R0 + 3) & 3) * 1024 + o3);
        GParts r;
        r.p = xor3(t0, t1, t2);
        r.q = t3;
        return r;
    }
    // SLOT = position of the G-box inside its round (0..6), I = round (1..8)
    template <int R0, int SLOT, int I>
    __device__ __forceinline__ GParts gs(uint32_t x) const
    {
        if constexpr (SLOT == 6 && ((ROUNDS >> (I - 1)) & 1)) return g_l1<R0>(x);
        else return BeltTabTwo::template g<R0>(x);
    }
};

// EXPERIMENT (round 3, tools/ab/belt_ab.py variant 15, profiles/r03_belt_sdwa_ab.txt): the two-table layout with every LDS address
// made by ONE instruction.  The address is (byte << 8) | lane_base with lane_base < 128: byte 1 of a register whose other
// bytes hold lane_base for good.  v_mov_b32_sdwa with dst_sel:BYTE_1 and dst_unused:UNUSED_PRESERVE drops byte k of x there
// (half rate: the cycles of the shift + v_bitop3 pair it replaces, one issue slot instead of two).  Each G-box position of
// the round owns its three address registers so that the two G-boxes the compiler keeps in flight do not meet in one.
struct BeltTabTwoS : BeltTabTwo {
    mutable uint32_t ar[7][3];
    __device__ explicit BeltTabTwoS(const uint8_t *l) : BeltTabTwo(l)
    {
#pragma unroll
        for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int k = 0; k < 3; ++k) ar[s][k] = base;
    }
    template <int R0, int SLOT, int I>
    __device__ __forceinline__ GParts gs(uint32_t x) const
    {
        uint32_t &a0 = ar[SLOT][0], &a2 = ar[SLOT][1], &a3 = ar[SLOT][2];
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(a0) : "v"(x));
        const uint32_t a1 = and_or(x, 0xFF00u, base);
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a2) : "v"(x));
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(a3) : "v"(x));
        constexpr int r0 = (R0 + 0) & 3, r1 = (R0 + 1) & 3, r2 = (R0 + 2) & 3, r3 = (R0 + 3) & 3;
        uint32_t t0 = *(lds_u32 *)(uintptr_t)(a0 + (r0 == 3 ? 128 : 0));
        uint32_t t1 = *(lds_u32 *)(uintptr_t)(a1 + (r1 == 3 ? 128 : 0));
        uint32_t t2 = *(lds_u32 *)(uintptr_t)(a2 + (r2 == 3 ? 128 : 0));
        uint32_t t3 = *(lds_u32 *)(uintptr_t)(a3 + (r3 == 3 ? 128 : 0));
        if (r0 == 1 || r0 == 2) t0 <<= 8 * r0;
        if (r1 == 1 || r1 == 2) t1 <<= 8 * r1;
        if (r2 == 1 || r2 == 2) t2 <<= 8 * r2;
        if (r3 == 1 || r3 == 2) t3 <<= 8 * r3;
        GParts r;
        r.p = xor3(t0, t1, t2);
        r.q = t3;
        return r;
    }
};

// EXPERIMENT (variant 16): BeltTabTwoS with the two post-shifts and the first xor folded into two v_lshl_or_b32 (the three T5
// entries of a G-box occupy disjoint bit ranges once shifted): ((t_<<16 << 8) | t_<<8) << 8 | t_<<0 -- 8 VALU instructions per
// G-box instead of 9, two of them half rate.
struct BeltTabTwoL : BeltTabTwoS {
    __device__ explicit BeltTabTwoL(const uint8_t *l) : BeltTabTwoS(l) {}
    template <int R0, int SLOT, int I>
    __device__ __forceinline__ GParts gs(uint32_t x) const
    {
        uint32_t &a0 = ar[SLOT][0], &a2 = ar[SLOT][1], &a3 = ar[SLOT][2];
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(a0) : "v"(x));
        const uint32_t a1 = and_or(x, 0xFF00u, base);
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a2) : "v"(x));
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(a3) : "v"(x));
        const uint32_t a[4] = {a0, a1, a2, a3};
        uint32_t t[4];                               // t[r]: the entry whose rotation is 5 + 8 r
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (R0 + k) & 3;
            t[r] = *(lds_u32 *)(uintptr_t)(a[k] + (r == 3 ? 128 : 0));
        }
        uint32_t u, v;
        asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(u) : "v"(t[2]), "v"(t[1]));
        asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(v) : "v"(u), "v"(t[0]));
        GParts g;
        g.p = v;
        g.q = t[3];
        return g;
    }
};

// The round-3 product table of the bank-private kernels: BeltTabTwoL with THREE address-register sets instead of seven
// (G-boxes that can be in flight together never share one: slots 0..6 of a round use sets 0 1 0 1 0 1 2, and slot 6 never
// meets slot 0 of the next round in the same set), so that the kernels built around 64 VGPRs keep their occupancy.
template <bool ONEWAIT>
struct BeltTabTwoPT : BeltTabTwo {
    mutable uint32_t ar[3][3];
    __device__ explicit BeltTabTwoPT(const uint8_t *l) : BeltTabTwo(l)
    {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 3; ++k) ar[s][k] = base;
    }
    template <int R0, int SLOT, int I>
    __device__ __forceinline__ GParts gs(uint32_t x) const
    {
        constexpr int SET = SLOT == 6 ? 2 : (SLOT & 1);
        uint32_t &a0 = ar[SET][0], &a2 = ar[SET][1], &a3 = ar[SET][2];
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(a0) : "v"(x));
        const uint32_t a1 = and_or(x, 0xFF00u, base);
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a2) : "v"(x));
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(a3) : "v"(x));
        const uint32_t a[4] = {a0, a1, a2, a3};
        uint32_t t[4];                               // t[r]: the entry whose rotation is 5 + 8 r
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (R0 + k) & 3;
            t[r] = *(lds_u32 *)(uintptr_t)(a[k] + (r == 3 ? 128 : 0));
        }
        // ONEWAIT (experiment, variant 22): all four entries asked for at one point, so that ONE s_waitcnt serves the G-box
        // instead of one per first use (the entries come back in order, a few cycles apart)
        if constexpr (ONEWAIT) asm("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
        uint32_t u, v;
        asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(u) : "v"(t[2]), "v"(t[1]));
        asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(v) : "v"(u), "v"(t[0]));
        GParts g;
        g.p = v;
        g.q = t[3];
        return g;
    }
};
typedef BeltTabTwoPT<false> BeltTabTwoP;
typedef BeltTabTwoPT<true> BeltTabTwoQ;

struct BeltTabSmall {
    static constexpr int kBytes = 4 * 256 * 4;             // 4096
    const uint8_t *lds;
    __device__ static void fill(uint8_t *lds, int tid, int nthreads)
    {
        uint32_t *t = reinterpret_cast<uint32_t *>(lds);
        for (int i = tid; i < 4 * 256; i += nthreads)
            t[i] = rotl32c((uint32_t)c_beltH[i & 255], 5 + 8 * (i >> 8));
    }
    __device__ explicit BeltTabSmall(const uint8_t *l) : lds(l) {}
    template <int R0>
    __device__ __forceinline__ GParts g(uint32_t x) const
    {
        const uint32_t b0 = x & 255u, b1 = (x >> 8) & 255u, b2 = (x >> 16) & 255u, b3 = x >> 24;
        const uint32_t t0 = *reinterpret_cast<const uint32_t *>(lds + ((R0 + 0) & 3) * 1024 + (b0 << 2));
        const uint32_t t1 = *reinterpret_cast<const uint32_t *>(lds + ((R0 + 1) & 3) * 1024 + (b1 << 2));
        const uint32_t t2 = *reinterpret_cast<const uint32_t *>(lds + ((R0 + 2) & 3) * 1024 + (b2 << 2));
        const uint32_t t3 = *reinterpret_cast<const uint32_t *>(lds + ((R0 + 3) & 3) * 1024 + (b3 << 2));
        GParts r;
        r.p = xor3(t0, t1, t2);
        r.q = t3;
        return r;
    }
};

// EXPERIMENT (round 4, tools/ab/long_hash_ab.py form 4): BeltTabSmall with each look-up address made by ONE instruction,
// v_lshlrev_b32_sdwa (byte k of x, shifted by 2), instead of extract + shift -- one dependent instruction less on a chain that is
// bound by dependent latency (profiles/r04_long_hash_ab.txt).  Same table, same bank behaviour.
struct BeltTabSmallS : BeltTabSmall {
    __device__ explicit BeltTabSmallS(const uint8_t *l) : BeltTabSmall(l) {}
    template <int R0>
    __device__ __forceinline__ GParts g(uint32_t x) const
    {
        uint32_t a0, a1, a2, a3;
        const uint32_t two = 2u;
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a0) : "v"(two), "v"(x));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a1) : "v"(two), "v"(x));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a2) : "v"(two), "v"(x));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a3) : "v"(two), "v"(x));
        const uint32_t t0 = *reinterpret_cast<const uint32_t *>(lds + ((R0 + 0) & 3) * 1024 + a0);
        const uint32_t t1 = *reinterpret_cast<const uint32_t *>(lds + ((R0 + 1) & 3) * 1024 + a1);
        const uint32_t t2 = *reinterpret_cast<const uint32_t *>(lds + ((R0 + 2) & 3) * 1024 + a2);
        const uint32_t t3 = *reinterpret_cast<const uint32_t *>(lds + ((R0 + 3) & 3) * 1024 + a3);
        GParts r;
        r.p = xor3(t0, t1, t2);
        r.q = t3;
        return r;
    }
};

// G_r(x), r = 5 + 8*R0 (belt_block.c:210-215): table (R0 + k) & 3 serves byte k of x.
// G5 = g<0>, G13 = g<1>, G21 = g<2>.

// G-box number SLOT of a round: tables with a slot-aware accessor (BeltTabHyb) choose their path by it
template <int R0, int SLOT, int I, class Tab>
__device__ __forceinline__ auto gbox(const Tab &T, uint32_t x, int) -> decltype(T.template gs<R0, SLOT, I>(x))
{
    return T.template gs<R0, SLOT, I>(x);
}
template <int R0, int SLOT, int I, class Tab>
__device__ __forceinline__ GParts gbox(const Tab &T, uint32_t x, long)
{
    return T.template g<R0>(x);
}

// one round, steps 2.1-2.9 of the standard (belt_block.c:231-240); I = round number,
// key index (7 I - 7 + j) mod 8 (subkey_e, :242).  Every "x ^= G" is one more xor3.
template <int I, class Tab>
__device__ __forceinline__ void belt_round(const Tab &T, uint32_t &a, uint32_t &b, uint32_t &c,
                                           uint32_t &d, const uint32_t (&K)[8])
{
    constexpr int o = 7 * I - 7;
    GParts g;
    g = gbox<0, 0, I>(T, a + K[(o + 0) & 7], 0);   b = xor3(b, g.p, g.q);                 // b ^= G5(a + k)
    g = gbox<2, 1, I>(T, d + K[(o + 1) & 7], 0);   c = xor3(c, g.p, g.q);                 // c ^= G21(d + k)
    g = gbox<1, 2, I>(T, b + K[(o + 2) & 7], 0);   a -= g.p ^ g.q;                        // a -= G13(b + k)
    g = gbox<2, 3, I>(T, b + c + K[(o + 3) & 7], 0);
    const uint32_t e = xor3(g.p, g.q, (uint32_t)I);                                   // G21(b + c + k) ^ i
    b += e;
    c -= e;
    g = gbox<1, 4, I>(T, c + K[(o + 4) & 7], 0);   d += g.p ^ g.q;                        // d += G13(c + k)
    g = gbox<2, 5, I>(T, a + K[(o + 5) & 7], 0);   b = xor3(b, g.p, g.q);                 // b ^= G21(a + k)
    g = gbox<0, 6, I>(T, d + K[(o + 6) & 7], 0);   c = xor3(c, g.p, g.q);                 // c ^= G5(d + k)
}

// E_K on (x0..x3): eight rounds with the (a,b,c,d) <- (b,d,a,c) role change realised by
// argument order (no moves), then the output order (b,d,a,c) (belt_block.c:258-269).
template <class Tab>
__device__ __forceinline__ void belt_encr(const Tab &T, uint32_t (&x)[4], const uint32_t (&K)[8])
{
    uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
    belt_round<1>(T, a, b, c, d, K);
    belt_round<2>(T, b, d, a, c, K);
    belt_round<3>(T, d, c, b, a, K);
    belt_round<4>(T, c, a, d, b, K);
    belt_round<5>(T, a, b, c, d, K);
    belt_round<6>(T, b, d, a, c, K);
    belt_round<7>(T, d, c, b, a, K);
    belt_round<8>(T, c, a, d, b, K);
    x[0] = b; x[1] = d; x[2] = a; x[3] = c;
}

// D_K: the same round with the subkeys taken in reverse (subkey_d, belt_block.c:243), rounds
// 8..1, role change (a,b,c,d) <- (c,a,d,b) by argument order, output order (c,a,d,b)
// (belt_block.c:286-295).
template <int I, class Tab>
__device__ __forceinline__ void belt_round_d(const Tab &T, uint32_t &a, uint32_t &b, uint32_t &c,
                                             uint32_t &d, const uint32_t (&K)[8])
{
    constexpr int o = 7 * I - 1;
    GParts g;
    g = gbox<0, 0, I>(T, a + K[(o - 0) & 7], 0);   b = xor3(b, g.p, g.q);
    g = gbox<2, 1, I>(T, d + K[(o - 1) & 7], 0);   c = xor3(c, g.p, g.q);
    g = gbox<1, 2, I>(T, b + K[(o - 2) & 7], 0);   a -= g.p ^ g.q;
    g = gbox<2, 3, I>(T, b + c + K[(o - 3) & 7], 0);
    const uint32_t e = xor3(g.p, g.q, (uint32_t)I);
    b += e;
    c -= e;
    g = gbox<1, 4, I>(T, c + K[(o - 4) & 7], 0);   d += g.p ^ g.q;
    g = gbox<2, 5, I>(T, a + K[(o - 5) & 7], 0);   b = xor3(b, g.p, g.q);
    g = gbox<0, 6, I>(T, d + K[(o - 6) & 7], 0);   c = xor3(c, g.p, g.q);
}

template <class Tab>
__device__ __forceinline__ void belt_decr(const Tab &T, uint32_t (&x)[4], const uint32_t (&K)[8])
{
    uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
    belt_round_d<8>(T, a, b, c, d, K);
    belt_round_d<7>(T, c, a, d, b, K);
    belt_round_d<6>(T, d, c, b, a, K);
    belt_round_d<5>(T, b, d, a, c, K);
    belt_round_d<4>(T, a, b, c, d, K);
    belt_round_d<3>(T, c, a, d, b, K);
    belt_round_d<2>(T, d, c, b, a, K);
    belt_round_d<1>(T, b, d, a, c, K);
    x[0] = c; x[1] = a; x[2] = d; x[3] = b;
}

// N independent blocks in lockstep: each G-box step is issued for all N blocks before the
// next step, so 4N LDS reads are in flight per wave (the LDS round trip, not the VALU, is
// what a single E_K chain waits on).
template <int N, int I, class Tab>
__device__ __forceinline__ void belt_round_n(const Tab &T, uint32_t (&a)[N], uint32_t (&b)[N],
                                             uint32_t (&c)[N], uint32_t (&d)[N], const uint32_t (&K)[8])
{
    constexpr int o = 7 * I - 7;
    GParts g[N];
#pragma unroll
    for (int u = 0; u < N; ++u) g[u] = gbox<0, 0, I>(T, a[u] + K[(o + 0) & 7], 0);
#pragma unroll
    for (int u = 0; u < N; ++u) b[u] = xor3(b[u], g[u].p, g[u].q);
#pragma unroll
    for (int u = 0; u < N; ++u) g[u] = gbox<2, 1, I>(T, d[u] + K[(o + 1) & 7], 0);
#pragma unroll
    for (int u = 0; u < N; ++u) c[u] = xor3(c[u], g[u].p, g[u].q);
#pragma unroll
    for (int u = 0; u < N; ++u) g[u] = gbox<1, 2, I>(T, b[u] + K[(o + 2) & 7], 0);
#pragma unroll
    for (int u = 0; u < N; ++u) a[u] -= g[u].p ^ g[u].q;
#pragma unroll
    for (int u = 0; u < N; ++u) g[u] = gbox<2, 3, I>(T, b[u] + c[u] + K[(o + 3) & 7], 0);
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const uint32_t e = xor3(g[u].p, g[u].q, (uint32_t)I);
        b[u] += e;
        c[u] -= e;
    }
#pragma unroll
    for (int u = 0; u < N; ++u) g[u] = gbox<1, 4, I>(T, c[u] + K[(o + 4) & 7], 0);
#pragma unroll
    for (int u = 0; u < N; ++u) d[u] += g[u].p ^ g[u].q;
#pragma unroll
    for (int u = 0; u < N; ++u) g[u] = gbox<2, 5, I>(T, a[u] + K[(o + 5) & 7], 0);
#pragma unroll
    for (int u = 0; u < N; ++u) b[u] = xor3(b[u], g[u].p, g[u].q);
#pragma unroll
    for (int u = 0; u < N; ++u) g[u] = gbox<0, 6, I>(T, d[u] + K[(o + 6) & 7], 0);
#pragma unroll
    for (int u = 0; u < N; ++u) c[u] = xor3(c[u], g[u].p, g[u].q);
}

template <int N, class Tab>
__device__ __forceinline__ void belt_encr_n(const Tab &T, uint32_t (&x)[N][4], const uint32_t (&K)[8])
{
    uint32_t a[N], b[N], c[N], d[N];
#pragma unroll
    for (int u = 0; u < N; ++u) { a[u] = x[u][0]; b[u] = x[u][1]; c[u] = x[u][2]; d[u] = x[u][3]; }
    belt_round_n<N, 1>(T, a, b, c, d, K);
    belt_round_n<N, 2>(T, b, d, a, c, K);
    belt_round_n<N, 3>(T, d, c, b, a, K);
    belt_round_n<N, 4>(T, c, a, d, b, K);
    belt_round_n<N, 5>(T, a, b, c, d, K);
    belt_round_n<N, 6>(T, b, d, a, c, K);
    belt_round_n<N, 7>(T, d, c, b, a, K);
    belt_round_n<N, 8>(T, c, a, d, b, K);
#pragma unroll
    for (int u = 0; u < N; ++u) { x[u][0] = b[u]; x[u][1] = d[u]; x[u][2] = a[u]; x[u][3] = c[u]; }
}

// E_K for a COUNTER stream.  Words 2 and 3 of the block (bits 64..127 of the counter) are the same for every block of a
// launch unless the low 64 bits wrap inside it, so the second G-box of round 1 -- c ^= G21(d + k_2) -- does not depend on
// the block: pre_c = c ^ G21(d + k_2) is computed once per thread and 55 G-boxes per block are left of 56 (round 3,
// profiles/r03_belt_mem_ab.txt).  The caller guarantees the precondition (launch_ctr_t checks the range on the host).
template <int N, class Tab>
__device__ __forceinline__ void belt_encr_n_ctr(const Tab &T, uint32_t (&x)[N][4], const uint32_t (&K)[8], uint32_t pre_c)
{
    uint32_t a[N], b[N], c[N], d[N];
    GParts g[N];
#pragma unroll
    for (int u = 0; u < N; ++u) { a[u] = x[u][0]; b[u] = x[u][1]; c[u] = pre_c; d[u] = x[u][3]; }
    {   // round 1, steps 1 and 3..7 (belt_round_n<N, 1> without its second G-box)
        constexpr int I = 1;
#pragma unroll
        for (int u = 0; u < N; ++u) g[u] = gbox<0, 0, I>(T, a[u] + K[0], 0);
#pragma unroll
        for (int u = 0; u < N; ++u) b[u] = xor3(b[u], g[u].p, g[u].q);
#pragma unroll
        for (int u = 0; u < N; ++u) g[u] = gbox<1, 2, I>(T, b[u] + K[2], 0);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] -= g[u].p ^ g[u].q;
#pragma unroll
        for (int u = 0; u < N; ++u) g[u] = gbox<2, 3, I>(T, b[u] + c[u] + K[3], 0);
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const uint32_t e = xor3(g[u].p, g[u].q, (uint32_t)I);
            b[u] += e;
            c[u] -= e;
        }
#pragma unroll
        for (int u = 0; u < N; ++u) g[u] = gbox<1, 4, I>(T, c[u] + K[4], 0);
#pragma unroll
        for (int u = 0; u < N; ++u) d[u] += g[u].p ^ g[u].q;
#pragma unroll
        for (int u = 0; u < N; ++u) g[u] = gbox<2, 5, I>(T, a[u] + K[5], 0);
#pragma unroll
        for (int u = 0; u < N; ++u) b[u] = xor3(b[u], g[u].p, g[u].q);
#pragma unroll
        for (int u = 0; u < N; ++u) g[u] = gbox<0, 6, I>(T, d[u] + K[6], 0);
#pragma unroll
        for (int u = 0; u < N; ++u) c[u] = xor3(c[u], g[u].p, g[u].q);
    }
    belt_round_n<N, 2>(T, b, d, a, c, K);
    belt_round_n<N, 3>(T, d, c, b, a, K);
    belt_round_n<N, 4>(T, c, a, d, b, K);
    belt_round_n<N, 5>(T, a, b, c, d, K);
    belt_round_n<N, 6>(T, b, d, a, c, K);
    belt_round_n<N, 7>(T, d, c, b, a, K);
    belt_round_n<N, 8>(T, c, a, d, b, K);
#pragma unroll
    for (int u = 0; u < N; ++u) { x[u][0] = b[u]; x[u][1] = d[u]; x[u][2] = a[u]; x[u][3] = c[u]; }
}

// sigma1/sigma2 of belt-compress (src/crypto/belt/belt_compr.c:27-87):
//   s1 = E_X(h0 ^ h1) ^ h0 ^ h1 ; h0' = E_{s1 || h1}(X0) ^ X0 ; h1' = E_{~s1 || h0}(X1) ^ X1
template <class Tab>
__device__ __forceinline__ void belt_compress(const Tab &T, uint32_t (&s1)[4], uint32_t (&h)[8],
                                              const uint32_t (&X)[8])
{
    uint32_t u[4], k1[8], k2[8], y0[4], y1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i] = h[i] ^ h[4 + i]; s1[i] = u[i]; }
    belt_encr(T, s1, X);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s1[i] ^= u[i];
        k1[i] = s1[i]; k1[4 + i] = h[4 + i];
        k2[i] = ~s1[i]; k2[4 + i] = h[i];
        y0[i] = X[i]; y1[i] = X[4 + i];
    }
    belt_encr(T, y0, k1);
    belt_encr(T, y1, k2);
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = y0[i] ^ X[i]; h[4 + i] = y1[i] ^ X[4 + i]; }
}

// The same compression by a PAIR of lanes (even / odd) that both hold h and X: the first encryption is
// done by both, then the two remaining ones -- which do not depend on each other -- one each, and the
// halves are swapped with a one-lane shuffle.  The chain step of a long belt-hash is 2 E instead of 3.
template <class Tab>
__device__ __forceinline__ void belt_compress_pair(const Tab &T, uint32_t (&s1)[4], uint32_t (&h)[8],
                                                   const uint32_t (&X)[8], uint32_t odd /* all-ones in the odd lane */)
{
    uint32_t u[4], key[8], y[4], xs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i] = h[i] ^ h[4 + i]; s1[i] = u[i]; }
    belt_encr(T, s1, X);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s1[i] ^= u[i];
        key[i] = s1[i] ^ odd;                                   // s1 | ~s1
        key[4 + i] = (h[4 + i] & ~odd) | (h[i] & odd);          // h1 | h0
        xs[i] = (X[i] & ~odd) | (X[4 + i] & odd);               // X0 | X1
        y[i] = xs[i];
    }
    belt_encr(T, y, key);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t mine = y[i] ^ xs[i];                     // h0' in the even lane, h1' in the odd one
        const uint32_t other = (uint32_t)__shfl_xor((int)mine, 1, 64);
        h[i] = (mine & ~odd) | (other & odd);
        h[4 + i] = (other & ~odd) | (mine & odd);
    }
}

// ---- E_K walked by a PAIR of lanes (round 4: long belt-hash chains) -------------------------------------------------------
// A lone chain is bound by the latency of its dependent instructions and LDS round trips (profiles/r04_long_hash_ab.txt), and
// the seven G-boxes of a round have dependency depth FOUR:
//     level 1   b ^= G5(a + k0)        ||  c ^= G21(d + k1)
//     level 2   a -= G13(b + k2)       ||  e = G21(b + c + k3) ^ i;  b += e;  c -= e
//     level 3   d += G13(c + k4)       ||  b ^= G21(a + k5)
//     level 4   c ^= G5(d + k6)
// Lane P (rQ = 0) takes the left column, lane Q (rQ = all-ones) the right one; each G-box result goes to the partner by DPP
// (quad_perm 1,0,3,2), folded into the consuming instruction where the compiler can.  So that both lanes run the SAME
// instructions on the same registers, Q keeps the state mirrored -- (w, u, v, z) = (a, b, c, d) in P, (d, c, b, a) in Q -- its
// round keys shifted by one (Ks[j] = K[j + 1]), and the rotation its G-box needs on top of the table's (G21 from the G5 / G13
// look-up: 16 / 8 more) is one v_alignbit with a per-lane amount.  Both lanes end with the whole block.
__device__ __forceinline__ uint32_t dpp_swap1(uint32_t x)      // the value of the other lane of the pair
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);    // quad_perm [1, 0, 3, 2]
}
template <int I, class Tab>
__device__ __forceinline__ void belt_round_split(const Tab &T, uint32_t &w, uint32_t &u, uint32_t &v, uint32_t &z,
                                                 const uint32_t (&K)[8], const uint32_t (&Ks)[8], uint32_t rQ, uint32_t sh16,
                                                 uint32_t sh8)
{
    constexpr int o = 7 * I - 7;
    const uint32_t nQ = ~rQ;
    GParts g;
    uint32_t gg, og, t;
    // level 1
    g = T.template g<0>(w + Ks[(o + 0) & 7]);
    gg = __builtin_amdgcn_alignbit(g.p ^ g.q, g.p ^ g.q, sh16);
    u ^= gg;
    v ^= dpp_swap1(gg);
    // level 2
    g = T.template g<1>(u + (v & rQ) + Ks[(o + 2) & 7]);
    t = xor3(g.p, g.q, ((uint32_t)I << 24) & rQ);               // e = G21(..) ^ i: i goes in BEFORE the 8 extra bits of rotation
    gg = __builtin_amdgcn_alignbit(t, t, sh8);
    og = dpp_swap1(gg);
    t = (og & nQ) | ((0u - gg) & rQ);                           // P: e from Q;  Q: -e (its own)
    u += t;
    v -= t;
    w -= gg & nQ;                                               // P: a -= G13(b + k2)
    z -= og & rQ;                                               // Q: the same a, from P
    // level 3
    g = T.template g<1>(((v & nQ) | (z & rQ)) + Ks[(o + 4) & 7]);
    gg = __builtin_amdgcn_alignbit(g.p ^ g.q, g.p ^ g.q, sh8);
    og = dpp_swap1(gg);
    z += gg & nQ;                                               // P: d += G13(c + k4)
    v ^= gg & rQ;                                               // Q: b ^= G21(a + k5)
    u ^= og & nQ;                                               // P: b ^= (from Q)
    w += og & rQ;                                               // Q: d += (from P)
    // level 4: both lanes the same G-box
    g = T.template g<0>(((z & nQ) | (w & rQ)) + K[(o + 6) & 7]);
    t = g.p ^ g.q;
    v ^= t & nQ;
    u ^= t & rQ;
}
template <class Tab>
__device__ __forceinline__ void belt_encr_split(const Tab &T, uint32_t (&x)[4], const uint32_t (&K)[8], uint32_t rQ)
{
    const uint32_t nQ = ~rQ;
    const uint32_t sh16 = rQ & 16u, sh8 = rQ & 24u;             // v_alignbit amounts: rotate left by 16 / 8 in lane Q, by 0 in lane P
    uint32_t Ks[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) Ks[j] = (K[j] & nQ) | (K[(j + 1) & 7] & rQ);
    uint32_t w = (x[0] & nQ) | (x[3] & rQ), u = (x[1] & nQ) | (x[2] & rQ), v = (x[2] & nQ) | (x[1] & rQ), z = (x[3] & nQ) | (x[0] & rQ);
    belt_round_split<1>(T, w, u, v, z, K, Ks, rQ, sh16, sh8);
    belt_round_split<2>(T, u, z, w, v, K, Ks, rQ, sh16, sh8);
    belt_round_split<3>(T, z, v, u, w, K, Ks, rQ, sh16, sh8);
    belt_round_split<4>(T, v, w, z, u, K, Ks, rQ, sh16, sh8);
    belt_round_split<5>(T, w, u, v, z, K, Ks, rQ, sh16, sh8);
    belt_round_split<6>(T, u, z, w, v, K, Ks, rQ, sh16, sh8);
    belt_round_split<7>(T, z, v, u, w, K, Ks, rQ, sh16, sh8);
    belt_round_split<8>(T, v, w, z, u, K, Ks, rQ, sh16, sh8);
    // logical (a, b, c, d) = (w, u, v, z) in P, (z, v, u, w) in Q; the block is (b, d, a, c)
    x[0] = (u & nQ) | (v & rQ);
    x[1] = (z & nQ) | (w & rQ);
    x[2] = (w & nQ) | (z & rQ);
    x[3] = (v & nQ) | (u & rQ);
}

// The compression by a QUAD of lanes that all hold h and X: lanes {0, 1} and {2, 3} are two such pairs; the first encryption
// is walked by both pairs (the same values), the two independent ones of the second stage by one pair each (`odd` = all-ones
// in lanes 2, 3), and the halves are swapped across the pairs.
template <class Tab>
__device__ __forceinline__ void belt_compress_quad(const Tab &T, uint32_t (&s1)[4], uint32_t (&h)[8], const uint32_t (&X)[8],
                                                   uint32_t rQ /* lane bit 0 */, uint32_t odd /* lane bit 1 */)
{
    uint32_t uu[4], key[8], y[4], xs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { uu[i] = h[i] ^ h[4 + i]; s1[i] = uu[i]; }
    belt_encr_split(T, s1, X, rQ);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s1[i] ^= uu[i];
        key[i] = s1[i] ^ odd;                                   // s1 | ~s1
        key[4 + i] = (h[4 + i] & ~odd) | (h[i] & odd);          // h1 | h0
        xs[i] = (X[i] & ~odd) | (X[4 + i] & odd);               // X0 | X1
        y[i] = xs[i];
    }
    belt_encr_split(T, y, key, rQ);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t mine = y[i] ^ xs[i];                     // h0' in lanes 0, 1; h1' in lanes 2, 3
        const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x4E, 0xF, 0xF, false);   // quad_perm [2, 3, 0, 1]
        h[i] = (mine & ~odd) | (other & odd);
        h[4 + i] = (other & ~odd) | (mine & odd);
    }
}

// ---- E_K with every G-box shared by a QUAD of lanes, one S-box byte each (round 4) -----------------------------------------
// The level-split above leaves the chain as slow as it was (profiles/r04_long_hash_ab.txt: same instruction count per lane --
// a lone wavefront is bound by the instructions it issues, ~5 cycles each, with the LDS round trip mostly hidden).  This form
// cuts the instructions per G-box instead: all four lanes of a quad hold the same (a, b, c, d) and compute x = a + k, lane j
// extracts byte j (v_bfe with its own shift), looks up ONE entry -- table (R0 + j) mod 4, a per-lane offset -- and two
// v_xor_b32_dpp (quad_perm 1,0,3,2 then 2,3,0,1) leave G(x) in all four: add, bfe, shift-add, ds_read, wait, xor, xor, apply =
// 8 instructions where one lane alone needs 13.
struct BeltQuadLane {
    uint32_t sh;            // 8 j
    uint32_t off[3];        // byte offsets of table (R0 + j) mod 4 for R0 = 0 (G5), 1 (G13), 2 (G21)
    __device__ explicit BeltQuadLane(unsigned j)
    {
        sh = 8u * j;
#pragma unroll
        for (int r = 0; r < 3; ++r) off[r] = ((r + j) & 3u) * 1024u;
    }
};
template <int R0>
__device__ __forceinline__ uint32_t gbox_quad(const uint8_t *lds, const BeltQuadLane &Q, uint32_t x)
{
    const uint32_t b = __builtin_amdgcn_ubfe(x, Q.sh, 8u);
    uint32_t t = *reinterpret_cast<const uint32_t *>(lds + ((b << 2) + Q.off[R0]));
    t ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xF, 0xF, false);      // quad_perm [1, 0, 3, 2]
    t ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xF, 0xF, false);      // quad_perm [2, 3, 0, 1]
    return t;
}
template <int I>
__device__ __forceinline__ void belt_round_quad(const uint8_t *lds, const BeltQuadLane &Q, uint32_t &a, uint32_t &b, uint32_t &c,
                                                uint32_t &d, const uint32_t (&K)[8])
{
    constexpr int o = 7 * I - 7;
    b ^= gbox_quad<0>(lds, Q, a + K[(o + 0) & 7]);
    c ^= gbox_quad<2>(lds, Q, d + K[(o + 1) & 7]);
    a -= gbox_quad<1>(lds, Q, b + K[(o + 2) & 7]);
    const uint32_t e = gbox_quad<2>(lds, Q, b + c + K[(o + 3) & 7]) ^ (uint32_t)I;
    b += e;
    c -= e;
    d += gbox_quad<1>(lds, Q, c + K[(o + 4) & 7]);
    b ^= gbox_quad<2>(lds, Q, a + K[(o + 5) & 7]);
    c ^= gbox_quad<0>(lds, Q, d + K[(o + 6) & 7]);
}
// lds: a BeltTabSmall image (4 tables of 256 dwords, rotations 5, 13, 21, 29)
__device__ __forceinline__ void belt_encr_quad(const uint8_t *lds, const BeltQuadLane &Q, uint32_t (&x)[4], const uint32_t (&K)[8])
{
    uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
    belt_round_quad<1>(lds, Q, a, b, c, d, K);
    belt_round_quad<2>(lds, Q, b, d, a, c, K);
    belt_round_quad<3>(lds, Q, d, c, b, a, K);
    belt_round_quad<4>(lds, Q, c, a, d, b, K);
    belt_round_quad<5>(lds, Q, a, b, c, d, K);
    belt_round_quad<6>(lds, Q, b, d, a, c, K);
    belt_round_quad<7>(lds, Q, d, c, b, a, K);
    belt_round_quad<8>(lds, Q, c, a, d, b, K);
    x[0] = b; x[1] = d; x[2] = a; x[3] = c;
}
// the compression by EIGHT lanes that all hold h and X: two such quads; the first encryption by both (the same values), the two
// independent ones of the second stage by one quad each (`odd` = all-ones in lanes 4-7), halves swapped across the quads
__device__ __forceinline__ void belt_compress_oct(const uint8_t *lds, const BeltQuadLane &Q, uint32_t (&s1)[4], uint32_t (&h)[8],
                                                  const uint32_t (&X)[8], uint32_t odd)
{
    uint32_t uu[4], key[8], y[4], xs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { uu[i] = h[i] ^ h[4 + i]; s1[i] = uu[i]; }
    belt_encr_quad(lds, Q, s1, X);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s1[i] ^= uu[i];
        key[i] = s1[i] ^ odd;
        key[4 + i] = (h[4 + i] & ~odd) | (h[i] & odd);
        xs[i] = (X[i] & ~odd) | (X[4 + i] & odd);
        y[i] = xs[i];
    }
    belt_encr_quad(lds, Q, y, key);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t mine = y[i] ^ xs[i];
        const uint32_t other = (uint32_t)__shfl_xor((int)mine, 4, 64);
        h[i] = (mine & ~odd) | (other & odd);
        h[4 + i] = (other & ~odd) | (mine & odd);
    }
}

// ---------------------------------------------------------------- GF(2^128) ---
// belt-bde tweaks (belt_bde.c:58, belt_lcl.c:99-108): elements of GF(2)[x] / (x^128 + x^7 + x^2 + x + 1),
// bit i of the 128-bit little-endian block = coefficient of x^i.  Only multiplications by powers
// of x are needed on the data path; the general product serves the jump-ahead of the tweak kernel.
struct Gf128 { uint64_t lo, hi; };

// r ^= over * (x^7 + x^2 + x + 1), i.e. `over` x^128 reduced; over < 2^64 so the product is < 2^71
__device__ __forceinline__ Gf128 gf_fold(Gf128 r, uint64_t over)
{
    r.lo ^= over ^ (over << 1) ^ (over << 2) ^ (over << 7);
    r.hi ^= (over >> 63) ^ (over >> 62) ^ (over >> 57);
    return r;
}
// s * x^k, 0 <= k < 64 (k may differ per lane)
__device__ __forceinline__ Gf128 gf_mul_xk(Gf128 s, unsigned k)
{
    if (k == 0) return s;
    Gf128 r;
    r.hi = (s.hi << k) | (s.lo >> (64 - k));
    r.lo = s.lo << k;
    return gf_fold(r, s.hi >> (64 - k));
}
// s * x^64: the halves move up, the old high half comes back folded
__device__ __forceinline__ Gf128 gf_mul_x64(Gf128 s)
{
    Gf128 r;
    r.hi = s.lo;
    r.lo = 0;
    return gf_fold(r, s.hi);
}
// general product (shift-and-add over the bits of b); jump-ahead only
__device__ inline Gf128 gf_mul(Gf128 a, Gf128 b)
{
    Gf128 r = {0, 0};
#pragma unroll 1
    for (int i = 0; i < 128; ++i) {
        const uint64_t bit = ((i < 64 ? b.lo >> i : b.hi >> (i - 64)) & 1ull);
        const uint64_t m = 0ull - bit;
        r.lo ^= a.lo & m;
        r.hi ^= a.hi & m;
        a = gf_mul_xk(a, 1);
    }
    return r;
}
// s * x^e for any 64-bit e
__device__ inline Gf128 gf_mul_xpow(Gf128 s, uint64_t e)
{
    Gf128 base = {2, 0};                 // x
#pragma unroll 1
    while (e) {
        if (e & 1) s = gf_mul(s, base);
        e >>= 1;
        if (e) base = gf_mul(base, base);
    }
    return s;
}

// base^e by square-and-multiply (jump-ahead of the polynomial MAC)
__device__ inline Gf128 gf_pow(Gf128 base, uint64_t e)
{
    Gf128 r = {1, 0};
#pragma unroll 1
    while (e) {
        if (e & 1) r = gf_mul(r, base);
        e >>= 1;
        if (e) base = gf_mul(base, base);
    }
    return r;
}
__device__ __forceinline__ Gf128 gf_from(const uint4 v)
{
    Gf128 a;
    a.lo = (uint64_t)v.x | (uint64_t)v.y << 32;
    a.hi = (uint64_t)v.z | (uint64_t)v.w << 32;
    return a;
}
__device__ __forceinline__ uint4 gf_to(const Gf128 a)
{
    return make_uint4((uint32_t)a.lo, (uint32_t)(a.lo >> 32), (uint32_t)a.hi, (uint32_t)(a.hi >> 32));
}

// Multiplication by one FIXED element R through 4-bit windows: a * R = XOR_p T[p][nibble_p(a)] with
// T[p][v] = (v * x^(4p)) * R, 32 x 16 entries of 16 bytes = 8 KiB of LDS.  All lanes read row p in
// the same instruction and entry v occupies banks 4v..4v+3, so distinct nibbles never conflict and
// equal nibbles broadcast.  (beltPolyMul with one operand fixed, belt_lcl.c:119-132.)
struct GfMulTab {
    static constexpr int kBytes = 32 * 16 * 16;
    const uint4 *t;
    __device__ explicit GfMulTab(const uint8_t *lds) : t(reinterpret_cast<const uint4 *>(lds)) {}
    static __device__ void fill(uint8_t *lds, Gf128 R, int tid, int nthreads)
    {
        uint4 *t = reinterpret_cast<uint4 *>(lds);
        for (int idx = tid; idx < 512; idx += nthreads) {
            const int p = idx >> 4, v = idx & 15;
            Gf128 b = R;                                   // R * x^(4p)
            unsigned k = 4u * p;
            if (k >= 64) { b = gf_mul_x64(b); k -= 64; }
            b = gf_mul_xk(b, k);
            Gf128 e = {0, 0};
#pragma unroll
            for (int bit = 0; bit < 4; ++bit) {
                if ((v >> bit) & 1) { e.lo ^= b.lo; e.hi ^= b.hi; }
                b = gf_mul_xk(b, 1);
            }
            t[idx] = gf_to(e);
        }
    }
    __device__ __forceinline__ Gf128 mul(const Gf128 a) const
    {
        const uint32_t w[4] = {(uint32_t)a.lo, (uint32_t)(a.lo >> 32), (uint32_t)a.hi, (uint32_t)(a.hi >> 32)};
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const uint32_t v = (w[p >> 3] >> (4 * (p & 7))) & 15u;
            const uint4 e = t[p * 16 + v];
            acc.x ^= e.x; acc.y ^= e.y; acc.z ^= e.z; acc.w ^= e.w;
        }
        return gf_from(acc);
    }
};

}  // namespace bee2hip
