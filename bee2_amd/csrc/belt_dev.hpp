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
//                 wave-instruction instead of ~7 with a shared 1 KiB table.
//   BeltTabSmall  the same 4 tables without replication (4 KiB) for kernels where
//                 belt is a sliver of the work (bign verify tail, single blocks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bee2hip {

// the belt S-box, generated at library load by the standard's LFSR recipe
// (belt_block.c:21-35) -- see capi.cpp -- and uploaded once per device.
extern __constant__ uint8_t c_beltH[256];

__device__ __forceinline__ uint32_t rotl32c(uint32_t x, int r)
{
    return __builtin_amdgcn_alignbit(x, x, 32 - r);
}

struct BeltTabWide {
    static constexpr int kBytes = 4 * 256 * 32 * 4;        // 131072
    const uint8_t *base;                                    // LDS base + (lane & 31) * 4
    // fill from every thread of the workgroup; caller must __syncthreads() afterwards
    __device__ static void fill(uint8_t *lds, int tid, int nthreads)
    {
        uint32_t *t = reinterpret_cast<uint32_t *>(lds);
        for (int i = tid; i < 4 * 256 * 32; i += nthreads) {
            const int e = i >> 5;                           // (r, idx); low 5 bits = bank copy
            const int r = e >> 8, idx = e & 255;
            t[i] = rotl32c((uint32_t)c_beltH[idx], 5 + 8 * r);
        }
    }
    __device__ explicit BeltTabWide(const uint8_t *lds) : base(lds + ((threadIdx.x & 31) << 2)) {}
    template <int R>   // R = 0..3  ->  rotl by 5, 13, 21, 29
    __device__ __forceinline__ uint32_t get(uint32_t byte) const
    {
        return *reinterpret_cast<const uint32_t *>(base + R * 32768 + (byte << 7));
    }
};

struct BeltTabSmall {
    static constexpr int kBytes = 4 * 256 * 4;             // 4096
    const uint8_t *base;
    __device__ static void fill(uint8_t *lds, int tid, int nthreads)
    {
        uint32_t *t = reinterpret_cast<uint32_t *>(lds);
        for (int i = tid; i < 4 * 256; i += nthreads)
            t[i] = rotl32c((uint32_t)c_beltH[i & 255], 5 + 8 * (i >> 8));
    }
    __device__ explicit BeltTabSmall(const uint8_t *lds) : base(lds) {}
    template <int R>
    __device__ __forceinline__ uint32_t get(uint32_t byte) const
    {
        return *reinterpret_cast<const uint32_t *>(base + R * 1024 + (byte << 2));
    }
};

// G_r(x), r = 5 + 8*R0: belt_block.c:210-215.  Table (R0 + k) & 3 serves byte k.
template <int R0, class Tab>
__device__ __forceinline__ uint32_t belt_G(const Tab &T, uint32_t x)
{
    const uint32_t b0 = x & 255u, b1 = (x >> 8) & 255u, b2 = (x >> 16) & 255u, b3 = x >> 24;
    return T.template get<(R0 + 0) & 3>(b0) ^ T.template get<(R0 + 1) & 3>(b1) ^
           T.template get<(R0 + 2) & 3>(b2) ^ T.template get<(R0 + 3) & 3>(b3);
}
#define BELT_G5(T, x)  belt_G<0>(T, x)
#define BELT_G13(T, x) belt_G<1>(T, x)
#define BELT_G21(T, x) belt_G<2>(T, x)

// one round, steps 2.1-2.9 of the standard (belt_block.c:231-240); I = round number,
// key index (7 I - 7 + j) mod 8 (subkey_e, :242)
template <int I, class Tab>
__device__ __forceinline__ void belt_round(const Tab &T, uint32_t &a, uint32_t &b, uint32_t &c,
                                           uint32_t &d, const uint32_t (&K)[8])
{
    constexpr int o = 7 * I - 7;
    b ^= BELT_G5(T, a + K[(o + 0) & 7]);
    c ^= BELT_G21(T, d + K[(o + 1) & 7]);
    a -= BELT_G13(T, b + K[(o + 2) & 7]);
    const uint32_t e = BELT_G21(T, b + c + K[(o + 3) & 7]) ^ (uint32_t)I;
    b += e;
    c -= e;
    d += BELT_G13(T, c + K[(o + 4) & 7]);
    b ^= BELT_G21(T, a + K[(o + 5) & 7]);
    c ^= BELT_G5(T, d + K[(o + 6) & 7]);
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

}  // namespace bee2hip
