// belt_kernels.hip -- belt CTR bulk encryption and plain block encryption on gfx950.
//
// beltCTR_blocks_kernel : H2 of SURVEY.md 8a.  Replaces the hot loop of
//   beltCTRStepE (src/crypto/belt/belt_ctr.c:85-97): block i of the stream is
//   X_i ^ E_K(ctr0 + first + i + 1), the counter being a 128-bit little-endian
//   integer (beltBlockIncU32, belt_ctr.c:27-35).  Blocks are independent, so one
//   lane takes one 16-byte block per step; a wavefront reads/writes 1 KiB per
//   global_load/store_dwordx4 (perfectly coalesced).  Algorithmic traffic:
//   32 B per block (16 read + 16 written), in place.
//
//   Launch shape: persistent, two 1024-thread workgroups per CU (512 total): the
//   bank-private two-table S-box layout takes 64 KiB of LDS (belt_dev.hpp, BeltTabTwo),
//   so 32 wavefronts per CU (8 per SIMD) hide the ~56 dependent LDS round trips of E_K;
//   each lane carries CTR_ILP independent blocks.  Measured cost model on MI355X
//   (tools/ubench): per SIMD a block costs ~2.5 cycles per VALU op plus
//   ~3.7 issue cycles per ds_read_b32 -- VALU and LDS issue do not overlap within a SIMD.
//
// belt_encr_blocks_kernel : E_K over n blocks in place (ctr0 = E_K(iv) of
//   beltCTRStart, belt_ctr.c:55-64; r = E_K(0) of beltMACStart, belt_mac.c:47-56;
//   the drop-in beltBlockEncr*).
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>
#include "belt_dev.hpp"
#include "common.hpp"

namespace bee2hip {

__device__ uint32_t d_beltT4[1024];       // experiment only (BeltTabHyb): rotl(S, 5 / 13 / 21 / 29), 4 x 256 dwords

constexpr int CTR_WG = 1024;
// 64 KiB, two workgroups (32 wavefronts) per CU; LDS addresses by one SDWA move each and the post-shifts folded into two
// v_lshl_or_b32: 8 VALU instructions per G-box instead of 12 (round 3: beltCTR 835 -> 965 GiB/s on 16 GiB, profiles/r03_belt_sdwa_ab.txt)
typedef BeltTabTwoP CtrTab;
constexpr int CTR_ILP = 1;          // independent blocks per lane per step (r02 A/B: 1 beats 2 by 1.3-1.5 %, profiles/r02_belt_variants.txt)

typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));      // what the non-temporal builtins accept
__device__ __forceinline__ uint4 nt_load16(const uint4 *p)
{
    const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store16(uint4 *p, const uint4 o)
{
    v4u_t v = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4u_t *>(p));
}

struct BeltKey { uint32_t k[8]; };
struct BeltCtr { uint32_t c[4]; };

__device__ __forceinline__ void ctr_at(uint32_t (&x)[4], const BeltCtr &c0, uint64_t add)
{
    // (c0 + add) mod 2^128, 64-bit offset
    const uint64_t lo = ((uint64_t)c0.c[1] << 32) | c0.c[0];
    const uint64_t hi = ((uint64_t)c0.c[3] << 32) | c0.c[2];
    const uint64_t nlo = lo + add;
    const uint64_t nhi = hi + (nlo < lo ? 1u : 0u);
    x[0] = (uint32_t)nlo; x[1] = (uint32_t)(nlo >> 32);
    x[2] = (uint32_t)nhi; x[3] = (uint32_t)(nhi >> 32);
}

// Tab / ILP are template parameters so that the round-2 A/B (profiles/r02_belt_variants.txt)
// runs the very same body: the product is <CtrTab, CTR_ILP>; the others are reachable only through
// bee2hip_internal_tune(1, v).
// MEM bit 2 (round 3): the counter's upper half does not change inside the launch (checked by launch_ctr_t): the second
// G-box of round 1 is hoisted out of the block loop (belt_encr_n_ctr).
// MEM (round-3 A/B; profiles/r03_belt_mem_ab.txt): bit 0 = non-temporal loads and stores of the
// stream (read once, written once), bit 1 = every workgroup walks ONE contiguous range of tiles instead of taking every
// gridDim-th tile (a new 2 MiB page per tile and workgroup).
template <class Tab, int ILP, int MEM = 0>
__global__ __launch_bounds__(CTR_WG, (BeltTabWide::kBytes / Tab::kBytes) * (CTR_WG / 256))
void beltCTR_blocks_kernel(uint4 *__restrict__ buf, size_t nblocks, BeltKey key, BeltCtr ctr0,
                           uint64_t first, uint4 *__restrict__ last_gamma)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Tab::fill(smem, threadIdx.x, CTR_WG);
    __syncthreads();
    const Tab T(smem);

    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];

    // tile = CTR_WG * ILP consecutive blocks; tiles are dealt round-robin to workgroups
    uint32_t pre_c = 0;
    if constexpr ((MEM & 4) != 0) {
        uint32_t x0[4];
        ctr_at(x0, ctr0, first + 1);
        const GParts g0 = T.template g<2>(x0[3] + K[1]);
        pre_c = xor3(x0[2], g0.p, g0.q);
    }
    const size_t tile = (size_t)CTR_WG * ILP;
    const size_t ntiles = (nblocks + tile - 1) / tile;
    const size_t per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const size_t t_begin = (MEM & 2) ? (size_t)blockIdx.x * per_wg * tile : (size_t)blockIdx.x * tile;
    const size_t t_end = (MEM & 2) ? ((size_t)blockIdx.x + 1) * per_wg * tile : nblocks;
    const size_t t_step = (MEM & 2) ? tile : (size_t)gridDim.x * tile;
    for (size_t t0 = t_begin; t0 < nblocks && t0 < t_end; t0 += t_step) {
        uint4 data[ILP];
        uint32_t g[ILP][4];
        bool live[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            const size_t i = t0 + (size_t)u * CTR_WG + threadIdx.x;
            live[u] = i < nblocks;
            if (live[u]) data[u] = (MEM & 1) ? nt_load16(&buf[i]) : buf[i];
            ctr_at(g[u], ctr0, first + i + 1);
        }
        if constexpr ((MEM & 4) != 0) belt_encr_n_ctr<ILP>(T, g, K, pre_c);
        else belt_encr_n<ILP>(T, g, K);
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            const size_t i = t0 + (size_t)u * CTR_WG + threadIdx.x;
            if (live[u]) {
                uint4 o;
                o.x = data[u].x ^ g[u][0]; o.y = data[u].y ^ g[u][1];
                o.z = data[u].z ^ g[u][2]; o.w = data[u].w ^ g[u][3];
                if (MEM & 1) nt_store16(&buf[i], o); else buf[i] = o;
                // streaming state needs the gamma of the final block (belt_ctr.c:89-96)
                if (last_gamma && i == nblocks - 1)
                    *last_gamma = make_uint4(g[u][0], g[u][1], g[u][2], g[u][3]);
            }
        }
    }
}

__global__ __launch_bounds__(64)
void belt_encr_blocks_kernel(uint4 *__restrict__ blocks, size_t nblocks, BeltKey key)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[BeltTabSmall::kBytes];
    BeltTabSmall::fill(smem, threadIdx.x, 64);
    __syncthreads();
    const BeltTabSmall T(smem);
    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= nblocks) return;
    const uint4 v = blocks[i];
    uint32_t x[4] = {v.x, v.y, v.z, v.w};
    belt_encr(T, x, K);
    blocks[i] = make_uint4(x[0], x[1], x[2], x[3]);
}

// ---- block-parallel modes on the same block function -------------------
// mode 0: ECB encrypt   dst[i] = E(src[i])                      (belt_ecb.c:63-84)
// mode 1: ECB decrypt   dst[i] = D(src[i])                      (belt_ecb.c:86-107)
// mode 2: CBC decrypt   dst[i] = D(src[i]) ^ (i ? src[i-1] : iv)  (belt_cbc.c:101-116); src != dst
template <int MODE>
__global__ __launch_bounds__(CTR_WG)
void belt_modes_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t nblocks,
                       BeltKey key, BeltCtr iv)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CtrTab::fill(smem, threadIdx.x, CTR_WG);
    __syncthreads();
    const CtrTab T(smem);
    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];
    for (size_t i = (size_t)blockIdx.x * CTR_WG + threadIdx.x; i < nblocks; i += (size_t)gridDim.x * CTR_WG) {
        const uint4 v = src[i];
        uint32_t x[4] = {v.x, v.y, v.z, v.w};
        if (MODE == 0) belt_encr(T, x, K);
        else belt_decr(T, x, K);
        if (MODE == 2) {
            uint4 p = make_uint4(iv.c[0], iv.c[1], iv.c[2], iv.c[3]);
            if (i) p = src[i - 1];
            x[0] ^= p.x; x[1] ^= p.y; x[2] ^= p.z; x[3] ^= p.w;
        }
        dst[i] = make_uint4(x[0], x[1], x[2], x[3]);
    }
}

// ---------------------------------------------------------------- belt-bde ---
// Block i (0-based, counted from the Start of the stream) uses the tweak s * x^(i+1) and
// Y = E_K(X ^ t) ^ t (belt_bde.c:51-85).  A wavefront owns one contiguous chunk of the stream;
// lane l starts at block c0 + l with t = B * x^l, where B = s * x^(c0+1) comes from
// bde_tweak_kernel, and steps 64 blocks at a time: t <- t * x^64, a swap of halves plus one fold.
// Every global access is 64 consecutive blocks = 1 KiB per wave-instruction.
//
// tweaks[w] = s * x^(first + w*chunk + 1) for w < nwaves; tweaks[nwaves] = s * x^(first + nblocks),
// the value beltBDEStepE leaves in belt_bde_st.s.
__global__ __launch_bounds__(64)
void bde_tweak_kernel(BeltCtr s0, uint64_t first, uint64_t chunk, uint64_t nblocks, unsigned nwaves,
                      uint4 *__restrict__ tweaks)
{
    const unsigned w = blockIdx.x * 64 + threadIdx.x;
    if (w > nwaves) return;
    Gf128 s;
    s.lo = (uint64_t)s0.c[0] | (uint64_t)s0.c[1] << 32;
    s.hi = (uint64_t)s0.c[2] | (uint64_t)s0.c[3] << 32;
    const uint64_t e = w < nwaves ? first + (uint64_t)w * chunk + 1 : first + nblocks;
    s = gf_mul_xpow(s, e);
    tweaks[w] = make_uint4((uint32_t)s.lo, (uint32_t)(s.lo >> 32), (uint32_t)s.hi, (uint32_t)(s.hi >> 32));
}

template <int DECR>
__global__ __launch_bounds__(CTR_WG)
void belt_bde_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t nblocks, uint64_t chunk,
                     BeltKey key, const uint4 *__restrict__ tweaks)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CtrTab::fill(smem, threadIdx.x, CTR_WG);
    __syncthreads();
    const CtrTab T(smem);
    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t w = (uint64_t)blockIdx.x * (CTR_WG / 64) + (threadIdx.x >> 6);     // global wavefront
    const uint64_t c0 = w * chunk;
    if (c0 >= nblocks) return;
    const uint64_t end = c0 + chunk < nblocks ? c0 + chunk : nblocks;
    const uint4 b = tweaks[w];
    Gf128 t;
    t.lo = (uint64_t)b.x | (uint64_t)b.y << 32;
    t.hi = (uint64_t)b.z | (uint64_t)b.w << 32;
    t = gf_mul_xk(t, lane);
    for (uint64_t i = c0 + lane; i < end; i += 64) {
        const uint4 v = src[i];
        const uint32_t t0 = (uint32_t)t.lo, t1 = (uint32_t)(t.lo >> 32), t2 = (uint32_t)t.hi, t3 = (uint32_t)(t.hi >> 32);
        uint32_t x[4] = {v.x ^ t0, v.y ^ t1, v.z ^ t2, v.w ^ t3};
        if (DECR) belt_decr(T, x, K); else belt_encr(T, x, K);
        dst[i] = make_uint4(x[0] ^ t0, x[1] ^ t1, x[2] ^ t2, x[3] ^ t3);
        t = gf_mul_x64(t);
    }
}

// ---------------------------------------------------------------- belt-sde ---
// belt-sde = XEX around belt-wbl (belt_sde.c:47-71): the first block is XORed with E_K(iv) before and
// after the wide-block transform of the whole sector.  belt-wbl on n blocks is 2n *sequential* rounds
// (belt_wbl.c:58-152): s = r_1 ^ .. ^ r_{n-1}; (r_1..r_n) <- (r_2, .., r_{n-1}, r_n ^ E_K(s) ^ <i>, s).
// A sector is a serial chain, sectors are independent: one lane per sector, the sector stays in HBM.
// Rolling form with a cyclic head h (logical r_1 lives at position h): the block a round needs as
// "r_n" is the `s` written by the previous round, so it is carried in registers and a round costs
// one 16-byte load (a[h]) and one store; after 2n rounds h is back at 0.  Decryption is the same
// walk backwards.  (Derived from the definition and checked against it in tests; not the
// reference's beltWBLStepEOpt.)
template <int DECR>
__global__ __launch_bounds__(CTR_WG)
void belt_sde_kernel(uint4 *__restrict__ sectors, uint32_t n, uint64_t nsectors, BeltKey key,
                     const uint4 *__restrict__ ivs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CtrTab::fill(smem, threadIdx.x, CTR_WG);
    __syncthreads();
    const CtrTab T(smem);
    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];
    const uint64_t idx = (uint64_t)blockIdx.x * CTR_WG + threadIdx.x;
    if (idx >= nsectors) return;
    uint4 *a = sectors + idx * n;
    const uint4 ivv = ivs[idx];
    uint32_t tw[4] = {ivv.x, ivv.y, ivv.z, ivv.w};
    belt_encr(T, tw, K);                                                  // E_K(iv)
    auto ld = [&](uint32_t q) { return a[q]; };
    {   // a[0] ^= E_K(iv)
        uint4 v = a[0];
        a[0] = make_uint4(v.x ^ tw[0], v.y ^ tw[1], v.z ^ tw[2], v.w ^ tw[3]);
    }
    uint32_t cur[4] = {0, 0, 0, 0};                                       // XOR of positions 0 .. n-2
    for (uint32_t q = 0; q + 1 < n; ++q) {
        const uint4 v = ld(q);
        cur[0] ^= v.x; cur[1] ^= v.y; cur[2] ^= v.z; cur[3] ^= v.w;
    }
    const uint64_t rounds = 2ull * n;
    if (!DECR) {
        const uint4 l = ld(n - 1);
        uint32_t prev[4] = {l.x, l.y, l.z, l.w};                          // logical r_n
        uint32_t h = 0;
        for (uint64_t i = 1; i <= rounds; ++i) {
            uint32_t e[4] = {cur[0], cur[1], cur[2], cur[3]};
            belt_encr(T, e, K);
            e[0] ^= (uint32_t)i; e[1] ^= (uint32_t)(i >> 32);
            const uint32_t x0 = prev[0] ^ e[0], x1 = prev[1] ^ e[1], x2 = prev[2] ^ e[2], x3 = prev[3] ^ e[3];
            const uint4 ah = a[h];
            a[h ? h - 1 : n - 1] = make_uint4(x0, x1, x2, x3);
#pragma unroll
            for (int k = 0; k < 4; ++k) prev[k] = cur[k];
            cur[0] ^= ah.x ^ x0; cur[1] ^= ah.y ^ x1; cur[2] ^= ah.z ^ x2; cur[3] ^= ah.w ^ x3;
            h = h + 1 == n ? 0 : h + 1;
        }
        a[n - 1] = make_uint4(prev[0], prev[1], prev[2], prev[3]);        // h == 0 again
    } else {
        const uint4 l = ld(n - 1);
        uint32_t sv[4] = {l.x, l.y, l.z, l.w};                            // s = logical r_n
        uint32_t h = n - 1;
        for (uint64_t i = rounds; i >= 1; --i) {
            uint32_t e[4] = {sv[0], sv[1], sv[2], sv[3]};
            belt_encr(T, e, K);
            e[0] ^= (uint32_t)i; e[1] ^= (uint32_t)(i >> 32);
            const uint4 x = a[h ? h - 1 : n - 1];
            a[h] = make_uint4(cur[0] ^ sv[0] ^ x.x, cur[1] ^ sv[1] ^ x.y, cur[2] ^ sv[2] ^ x.z, cur[3] ^ sv[3] ^ x.w);
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = sv[k];
            sv[0] = x.x ^ e[0]; sv[1] = x.y ^ e[1]; sv[2] = x.z ^ e[2]; sv[3] = x.w ^ e[3];
            h = h ? h - 1 : n - 1;
        }
        a[h] = make_uint4(sv[0], sv[1], sv[2], sv[3]);                    // h == n - 1 again
    }
    {
        uint4 v = a[0];
        a[0] = make_uint4(v.x ^ tw[0], v.y ^ tw[1], v.z ^ tw[2], v.w ^ tw[3]);
    }
}

// belt-sde with per-lane LINE buffering.  The rolling walk of belt_sde_kernel touches one 16-byte block
// per round and lane, and the 64 lanes of a wavefront sit in 64 different sectors: every access drags a
// whole memory line along and the resident working set (32 wavefronts x 64 sectors per CU) is far beyond
// L2 -- rocprof counted 2.75 GiB fetched + 2.5 GiB written for 0.5 GiB of sectors.  The walk is sequential
// though, so a lane can fetch LINE consecutive blocks at once every LINE-th round and collect its LINE
// outputs in registers to store them as one line: 3 line reads + 2 line writes per line of the sector
// and LINE times fewer memory instructions.  Encryption stores trail the loads by one position (the
// block of position h - 1 is produced in the round at h), so a line is completed by the first round of
// the next group; decryption stores fill exactly one line per group and its loads reach one block into
// the line below.  Both orders were checked against the definition in a plain-Python model first.
// Requires n % LINE == 0 and n >= 2 * LINE; other sector shapes use belt_sde_kernel.
template <int DECR, int LINE>
__global__ __launch_bounds__(CTR_WG)
void belt_sde_lines_kernel(uint4 *__restrict__ sectors, uint32_t n, uint64_t nsectors, BeltKey key,
                           const uint4 *__restrict__ ivs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CtrTab::fill(smem, threadIdx.x, CTR_WG);
    __syncthreads();
    const CtrTab T(smem);
    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];
    const uint64_t idx = (uint64_t)blockIdx.x * CTR_WG + threadIdx.x;
    if (idx >= nsectors) return;
    uint4 *a = sectors + idx * n;
    const uint4 ivv = ivs[idx];
    uint32_t tw[4] = {ivv.x, ivv.y, ivv.z, ivv.w};
    belt_encr(T, tw, K);                                                  // E_K(iv)
    const uint32_t nl = n / LINE;
    uint4 rbuf[LINE], wcur[LINE], wprev[LINE];
    // first pass: cur = XOR of positions 0 .. n-2 with the tweak on position 0 (XORed on the fly: position 0 is
    // rewritten by the walk anyway), and the block of position n-1
    uint32_t cur[4] = {tw[0], tw[1], tw[2], tw[3]};
    uint4 last = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < nl; ++k) {
#pragma unroll
        for (int j = 0; j < LINE; ++j) rbuf[j] = a[(size_t)LINE * k + j];
#pragma unroll
        for (int j = 0; j < LINE; ++j) {
            if (k + 1 < nl || j + 1 < LINE) { cur[0] ^= rbuf[j].x; cur[1] ^= rbuf[j].y; cur[2] ^= rbuf[j].z; cur[3] ^= rbuf[j].w; }
            else last = rbuf[j];
        }
    }
    const uint64_t rounds = 2ull * n;
    if (!DECR) {
        uint32_t prev[4] = {last.x, last.y, last.z, last.w};
        uint64_t i = 0;
        for (uint32_t g = 0; g < 2 * nl; ++g) {
            const uint32_t k = g < nl ? g : g - nl;
#pragma unroll
            for (int r = 0; r < LINE; ++r) {
                ++i;
                uint32_t e[4] = {cur[0], cur[1], cur[2], cur[3]};
                belt_encr(T, e, K);
                e[0] ^= (uint32_t)i; e[1] ^= (uint32_t)(i >> 32);
                const uint4 x = make_uint4(prev[0] ^ e[0], prev[1] ^ e[1], prev[2] ^ e[2], prev[3] ^ e[3]);
                if (r == 0) {
                    if (g == 0) a[n - 1] = x;                             // the rest of that line is still the input
                    else {
                        wprev[LINE - 1] = x;
                        const uint32_t kp = k ? k - 1 : nl - 1;
#pragma unroll
                        for (int j = 0; j < LINE; ++j) a[(size_t)LINE * kp + j] = wprev[j];
                    }
#pragma unroll
                    for (int j = 0; j < LINE; ++j) rbuf[j] = a[(size_t)LINE * k + j];
                    if (g == 0) { rbuf[0].x ^= tw[0]; rbuf[0].y ^= tw[1]; rbuf[0].z ^= tw[2]; rbuf[0].w ^= tw[3]; }
                } else {
                    wcur[r - 1] = x;
                }
                const uint4 ah = rbuf[r];
#pragma unroll
                for (int q = 0; q < 4; ++q) prev[q] = cur[q];
                cur[0] ^= ah.x ^ x.x; cur[1] ^= ah.y ^ x.y; cur[2] ^= ah.z ^ x.z; cur[3] ^= ah.w ^ x.w;
            }
#pragma unroll
            for (int j = 0; j < LINE - 1; ++j) wprev[j] = wcur[j];
        }
        wprev[LINE - 1] = make_uint4(prev[0], prev[1], prev[2], prev[3]);
#pragma unroll
        for (int j = 0; j < LINE; ++j) a[(size_t)LINE * (nl - 1) + j] = wprev[j];
        // the tweak on the first block once more (belt_sde.c:59)
        uint4 v = a[0];
        a[0] = make_uint4(v.x ^ tw[0], v.y ^ tw[1], v.z ^ tw[2], v.w ^ tw[3]);
    } else {
        uint32_t sv[4] = {last.x, last.y, last.z, last.w};               // rbuf holds the top line
        uint64_t i = rounds;
        for (uint32_t g = 0; g < 2 * nl; ++g) {
            const uint32_t k = g < nl ? nl - 1 - g : 2 * nl - 1 - g;
            uint4 nxt[LINE];
#pragma unroll
            for (int r = 0; r < LINE; ++r) {
                uint32_t e[4] = {sv[0], sv[1], sv[2], sv[3]};
                belt_encr(T, e, K);
                e[0] ^= (uint32_t)i; e[1] ^= (uint32_t)(i >> 32);
                uint4 x;
                if (r < LINE - 1) x = rbuf[LINE - 2 - r];
                else {
                    const uint32_t kb = k ? k - 1 : nl - 1;
#pragma unroll
                    for (int j = 0; j < LINE; ++j) nxt[j] = a[(size_t)LINE * kb + j];
                    // line 0 read in the first lap is still the input: its first block carries the tweak
                    if (g < nl && kb == 0) { nxt[0].x ^= tw[0]; nxt[0].y ^= tw[1]; nxt[0].z ^= tw[2]; nxt[0].w ^= tw[3]; }
                    x = nxt[LINE - 1];
                }
                wcur[LINE - 1 - r] = make_uint4(cur[0] ^ sv[0] ^ x.x, cur[1] ^ sv[1] ^ x.y, cur[2] ^ sv[2] ^ x.z,
                                                cur[3] ^ sv[3] ^ x.w);
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[q] = sv[q];
                sv[0] = x.x ^ e[0]; sv[1] = x.y ^ e[1]; sv[2] = x.z ^ e[2]; sv[3] = x.w ^ e[3];
                --i;
            }
#pragma unroll
            for (int j = 0; j < LINE; ++j) a[(size_t)LINE * k + j] = wcur[j];
#pragma unroll
            for (int j = 0; j < LINE; ++j) rbuf[j] = nxt[j];
        }
        a[n - 1] = make_uint4(sv[0], sv[1], sv[2], sv[3]);
        uint4 v = a[0];
        a[0] = make_uint4(v.x ^ tw[0], v.y ^ tw[1], v.z ^ tw[2], v.w ^ tw[3]);
    }
}

// ---------------------------------------------------------------- belt-che ---
// Keystream of belt-che (belt_che.c:86-98): S_0 = E_K(iv), S_j = S_{j-1} * x ^ 1, block i (0-based
// from the Start of the stream) is XORed with E_K(S_{i+1}).  Closed form S_j = S_0 x^j ^ (x^j ^ 1) q
// with q = (x + 1)^-1 = 0xFF..F82 (f(1) = 1, so q = (f + 1) / (x + 1)); for l <= 64:
// S_{j+l} = S_j x^l ^ (2^l - 1), whose constant never reaches x^128 and needs no reduction.
// Same shape as belt-bde: a wavefront owns a contiguous chunk, lane l starts from B x^l ^ (2^l - 1)
// and steps 64 blocks at a time with S <- S x^64 ^ (2^64 - 1).
//
// states[w] = S_{first + w*chunk + 1} for w < nwaves; states[nwaves] = S_{first + nblocks}
__global__ __launch_bounds__(64)
void che_state_kernel(BeltCtr s0, uint64_t first, uint64_t chunk, uint64_t nblocks, unsigned nwaves,
                      uint4 *__restrict__ states)
{
    const unsigned w = blockIdx.x * 64 + threadIdx.x;
    if (w > nwaves) return;
    const uint64_t j = w < nwaves ? first + (uint64_t)w * chunk + 1 : first + nblocks;
    const Gf128 one = {1, 0}, q = {0xFFFFFFFFFFFFFF82ull, 0xFFFFFFFFFFFFFFFFull};
    const Gf128 P = gf_mul_xpow(one, j);                                   // x^j
    const Gf128 a = gf_mul(gf_from(make_uint4(s0.c[0], s0.c[1], s0.c[2], s0.c[3])), P);
    Gf128 Pm = P;
    Pm.lo ^= 1;
    const Gf128 b = gf_mul(Pm, q);
    Gf128 r;
    r.lo = a.lo ^ b.lo; r.hi = a.hi ^ b.hi;
    states[w] = gf_to(r);
}

__global__ __launch_bounds__(CTR_WG)
void belt_che_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t nblocks, uint64_t chunk,
                     BeltKey key, const uint4 *__restrict__ states)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CtrTab::fill(smem, threadIdx.x, CTR_WG);
    __syncthreads();
    const CtrTab T(smem);
    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t w = (uint64_t)blockIdx.x * (CTR_WG / 64) + (threadIdx.x >> 6);
    const uint64_t c0 = w * chunk;
    if (c0 >= nblocks) return;
    const uint64_t end = c0 + chunk < nblocks ? c0 + chunk : nblocks;
    Gf128 t = gf_mul_xk(gf_from(states[w]), lane);
    t.lo ^= (1ull << lane) - 1ull;                                         // lane < 64
    for (uint64_t i = c0 + lane; i < end; i += 64) {
        uint32_t g[4] = {(uint32_t)t.lo, (uint32_t)(t.lo >> 32), (uint32_t)t.hi, (uint32_t)(t.hi >> 32)};
        belt_encr(T, g, K);
        const uint4 v = src[i];
        dst[i] = make_uint4(v.x ^ g[0], v.y ^ g[1], v.z ^ g[2], v.w ^ g[3]);
        t = gf_mul_x64(t);
        t.lo = ~t.lo;                                                      // ^ (2^64 - 1)
    }
}

// ------------------------------------------------- belt-dwp: polynomial MAC ---
// T = t * r^n ^ XOR_{i=1..n} X_i * r^(n-i+1)  -- the value of `t` after absorbing n blocks
// (belt_dwp.c:96-101: t <- (t ^ X) * r).  A wavefront owns one contiguous chunk; inside it the 64
// lanes take interleaved blocks (coalesced 1 KiB loads) and run Horner with the single fixed
// multiplier R = r^64 (GfMulTab); at the end lane l scales by r^(63-l) and the wavefront XOR-reduces
// to the chunk polynomial P_w = XOR_j X_j r^(end_w-1-j).  A chunk that is not a multiple of 64 blocks
// is treated as left-padded with zero blocks.  polyhash_finish_kernel then weights every P_w with
// r^(n - end_w + 1) and XORs everything, plus t * r^n, into the 16-byte result.
//
// consts[l] = r^(63-l) for l < 64, consts[64] = r^64.
__global__ __launch_bounds__(128)
void polyhash_prep_kernel(BeltCtr r, uint4 *__restrict__ consts)
{
    const unsigned l = threadIdx.x;
    if (l > 64) return;
    const Gf128 rr = gf_from(make_uint4(r.c[0], r.c[1], r.c[2], r.c[3]));
    consts[l] = gf_to(gf_pow(rr, l < 64 ? 63 - l : 64));
}

constexpr int PH_WG = 256;
// nbytes: length of the data; the last block may be partial and counts as zero-padded (belt_dwp.c:118-119)
__global__ __launch_bounds__(PH_WG)
void polyhash_kernel(const uint8_t *__restrict__ data, uint64_t nbytes, uint64_t chunk,
                     const uint4 *__restrict__ consts, uint4 *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[GfMulTab::kBytes];
    GfMulTab::fill(smem, gf_from(consts[64]), threadIdx.x, PH_WG);
    __syncthreads();
    const GfMulTab M(smem);
    const uint64_t nblocks = (nbytes + 15) / 16;
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t w = (uint64_t)blockIdx.x * (PH_WG / 64) + (threadIdx.x >> 6);
    const uint64_t c0 = w * chunk;
    if (c0 >= nblocks) return;
    const uint64_t nw = c0 + chunk <= nblocks ? chunk : nblocks - c0;      // blocks in this chunk
    const uint64_t steps = (nw + 63) / 64;
    const uint64_t pad = steps * 64 - nw;                                  // zero blocks in front
    const uint4 *blk = reinterpret_cast<const uint4 *>(data);
    Gf128 a = {0, 0};
    for (uint64_t k = 0; k < steps; ++k) {
        const uint64_t p = 64 * k + lane;
        a = M.mul(a);
        if (p >= pad) {
            const uint64_t j = c0 + p - pad;
            uint4 v;
            if (16 * j + 16 <= nbytes) v = blk[j];
            else {                                                        // the ragged last block
                uint32_t q[4] = {0, 0, 0, 0};
                for (uint64_t b = 16 * j; b < nbytes; ++b) q[(b & 15) >> 2] |= (uint32_t)data[b] << (8 * (b & 3));
                v = make_uint4(q[0], q[1], q[2], q[3]);
            }
            a.lo ^= (uint64_t)v.x | (uint64_t)v.y << 32;
            a.hi ^= (uint64_t)v.z | (uint64_t)v.w << 32;
        }
    }
    a = gf_mul(a, gf_from(consts[lane]));                                  // * r^(63 - lane)
    uint32_t x0 = (uint32_t)a.lo, x1 = (uint32_t)(a.lo >> 32), x2 = (uint32_t)a.hi, x3 = (uint32_t)(a.hi >> 32);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        x0 ^= __shfl_xor(x0, m, 64); x1 ^= __shfl_xor(x1, m, 64);
        x2 ^= __shfl_xor(x2, m, 64); x3 ^= __shfl_xor(x3, m, 64);
    }
    if (lane == 0) partial[w] = make_uint4(x0, x1, x2, x3);
}

// out (zeroed before) ^= P_w * r^(n - end_w + 1) for every chunk, and ^= t * r^n
__global__ __launch_bounds__(64)
void polyhash_finish_kernel(const uint4 *__restrict__ partial, unsigned nwaves, uint64_t chunk, uint64_t nblocks,
                            BeltCtr r, BeltCtr t_in, uint32_t *__restrict__ out)
{
    const unsigned w = blockIdx.x * 64 + threadIdx.x;
    if (w > nwaves) return;
    const Gf128 rr = gf_from(make_uint4(r.c[0], r.c[1], r.c[2], r.c[3]));
    Gf128 v;
    if (w < nwaves) {
        const uint64_t c0 = (uint64_t)w * chunk;
        if (c0 >= nblocks) return;
        const uint64_t end = c0 + chunk <= nblocks ? c0 + chunk : nblocks;
        v = gf_mul(gf_from(partial[w]), gf_pow(rr, nblocks - end + 1));
    } else {
        v = gf_mul(gf_from(make_uint4(t_in.c[0], t_in.c[1], t_in.c[2], t_in.c[3])), gf_pow(rr, nblocks));
    }
    atomicXor(out + 0, (uint32_t)v.lo); atomicXor(out + 1, (uint32_t)(v.lo >> 32));
    atomicXor(out + 2, (uint32_t)v.hi); atomicXor(out + 3, (uint32_t)(v.hi >> 32));
}

// CBC encryption is a serial chain per message (belt_cbc.c:63-84): one lane per message,
// n messages of nblk full blocks each, per-message iv, in place.  ivs[m] receives the last
// ciphertext block (the chaining value bee2 keeps in belt_cbc_st.block).
__global__ __launch_bounds__(CTR_WG)
void belt_cbc_encr_kernel(uint4 *__restrict__ msgs, size_t nblk, size_t n, BeltKey key, uint4 *__restrict__ ivs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CtrTab::fill(smem, threadIdx.x, CTR_WG);
    __syncthreads();
    const CtrTab T(smem);
    const size_t m = (size_t)blockIdx.x * CTR_WG + threadIdx.x;
    if (m >= n) return;
    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];
    uint4 c = ivs[m];
    uint32_t x[4] = {c.x, c.y, c.z, c.w};
    uint4 *p = msgs + m * nblk;
    // 8 blocks = one 128-byte line per lane and memory instruction: the lanes of a wavefront sit in 64 different
    // messages, a 16-byte access would move a whole line for a quarter of it (cf. belt_sde_lines_kernel)
    size_t i = 0;
    for (; i + 8 <= nblk; i += 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[i + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[0] ^= v[j].x; x[1] ^= v[j].y; x[2] ^= v[j].z; x[3] ^= v[j].w;
            belt_encr(T, x, K);
            v[j] = make_uint4(x[0], x[1], x[2], x[3]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) p[i + j] = v[j];
    }
    for (; i < nblk; ++i) {
        const uint4 v = p[i];
        x[0] ^= v.x; x[1] ^= v.y; x[2] ^= v.z; x[3] ^= v.w;
        belt_encr(T, x, K);
        p[i] = make_uint4(x[0], x[1], x[2], x[3]);
    }
    ivs[m] = make_uint4(x[0], x[1], x[2], x[3]);
}

__global__ __launch_bounds__(64)
void belt_decr_blocks_kernel(uint4 *__restrict__ blocks, size_t nblocks, BeltKey key)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[BeltTabSmall::kBytes];
    BeltTabSmall::fill(smem, threadIdx.x, 64);
    __syncthreads();
    const BeltTabSmall T(smem);
    uint32_t K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = key.k[i];
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= nblocks) return;
    const uint4 v = blocks[i];
    uint32_t x[4] = {v.x, v.y, v.z, v.w};
    belt_decr(T, x, K);
    blocks[i] = make_uint4(x[0], x[1], x[2], x[3]);
}

// per-device launch facts (several devices may be driven from one process)
static int g_num_cus[64];
int cur_dev()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return dev;
}
int num_cus()
{
    const int dev = cur_dev();
    if (!g_num_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_num_cus[dev] = n;
        else
            g_num_cus[dev] = 256;
    }
    return g_num_cus[dev];
}

// hipFuncAttributeMaxDynamicSharedMemorySize, once per (device, kernel) -- again only when a launch asks for more than the
// kernel was granted so far -- and safe from any thread (the flags used to be plain static bools per call site: a benign
// race, but a race)
hipError_t dyn_lds_once(const void *kern, size_t bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, const void *>, size_t>> done;
    const std::pair<int, const void *> key(cur_dev(), kern);
    std::lock_guard<std::mutex> lk(mu);
    auto it = std::find_if(done.begin(), done.end(), [&](const auto &e) { return e.first == key; });
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    if (it != done.end()) it->second = bytes;
    else done.push_back({key, bytes});
    return hipSuccess;
}

err_t upload_beltH(const uint8_t *H)
{
    B2H_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_beltH), H, 256));
    uint32_t t4[1024];
    for (int i = 0; i < 1024; ++i) {
        const uint32_t v = H[i & 255];
        const int r = 5 + 8 * (i >> 8);
        t4[i] = (v << r) | (v >> (32 - r));
    }
    B2H_TRY(hipMemcpyToSymbol(HIP_SYMBOL(d_beltT4), t4, sizeof t4));
    return ERR_OK;
}

static int g_ctr_variant = 0;
void set_ctr_variant(int v) { g_ctr_variant = v; }

template <class Tab, int ILP, int MEM = 0>
static err_t launch_ctr_t(void *d_buf, size_t nblocks, const BeltKey &k, const BeltCtr &c, uint64_t first,
                          void *d_last_gamma, hipStream_t st)
{
    auto kern = beltCTR_blocks_kernel<Tab, ILP, MEM>;
    B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(kern), Tab::kBytes));
    const size_t tile = (size_t)CTR_WG * ILP;
    size_t grid = (nblocks + tile - 1) / tile;
    const size_t cap = (size_t)num_cus() * (BeltTabWide::kBytes / Tab::kBytes);
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CTR_WG), Tab::kBytes, st, (uint4 *)d_buf, nblocks, k, c, first,
                       (uint4 *)d_last_gamma);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

err_t launch_belt_ctr_blocks(void *d_buf, size_t nblocks, const uint32_t key[8],
                             const uint32_t ctr0[4], uint64_t first, void *d_last_gamma,
                             hipStream_t st)
{
    if (nblocks == 0) return ERR_OK;
    BeltKey k; BeltCtr c;
    for (int i = 0; i < 8; ++i) k.k[i] = key[i];
    for (int i = 0; i < 4; ++i) c.c[i] = ctr0[i];
#ifdef BEE2HIP_EXPERIMENTS      // the A/B record (tools/ab/belt_ab.py; built by tools/ab/ab_lib.sh only)
    switch (g_ctr_variant) {          // A/B only (bee2hip_internal_tune(1, v)); 0 = the product
    case 1: return launch_ctr_t<BeltTabTwo, 2>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 2: return launch_ctr_t<BeltTabTwo, 3>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 3: return launch_ctr_t<BeltTabTwo, 4>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 4: return launch_ctr_t<BeltTabWide, 2>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 5: return launch_ctr_t<BeltTabWide, 3>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 6: return launch_ctr_t<BeltTabWide, 4>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 7: return launch_ctr_t<BeltTabTwo, CTR_ILP, 1>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 8: return launch_ctr_t<BeltTabTwo, CTR_ILP, 2>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 9: return launch_ctr_t<BeltTabTwo, CTR_ILP, 3>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    case 10: return launch_ctr_t<BeltTabHyb<0xFF>, 1>(d_buf, nblocks, k, c, first, d_last_gamma, st);   // 32 of 224 lookups via L1
    case 11: return launch_ctr_t<BeltTabHyb<0x55>, 1>(d_buf, nblocks, k, c, first, d_last_gamma, st);   // 16
    case 12: return launch_ctr_t<BeltTabHyb<0x11>, 1>(d_buf, nblocks, k, c, first, d_last_gamma, st);   // 8
    case 13: return launch_ctr_t<BeltTabTwo, CTR_ILP, 0>(d_buf, nblocks, k, c, first, d_last_gamma, st);   // the round-2 product
    case 14: return launch_ctr_t<BeltTabTwo, CTR_ILP, 3>(d_buf, nblocks, k, c, first, d_last_gamma, st);   // round 3 without the hoisted G-box
    case 15: return launch_ctr_t<BeltTabTwoS, 1, 3>(d_buf, nblocks, k, c, first, d_last_gamma, st);    // one-instruction (SDWA) LDS addresses
    case 16: return launch_ctr_t<BeltTabTwoL, 1, 3>(d_buf, nblocks, k, c, first, d_last_gamma, st);    // + v_lshl_or_b32 combine
    case 17: return launch_ctr_t<BeltTabTwoS, 1, 7>(d_buf, nblocks, k, c, first, d_last_gamma, st);    // 15 + hoisted round-1 G-box
    case 18: return launch_ctr_t<BeltTabTwoL, 1, 7>(d_buf, nblocks, k, c, first, d_last_gamma, st);    // 16 + hoisted round-1 G-box
    case 19: return launch_ctr_t<BeltTabTwoS, 2, 7>(d_buf, nblocks, k, c, first, d_last_gamma, st);    // 17 with 2 blocks per lane
    case 21: return launch_ctr_t<BeltTabTwo, 1, 7>(d_buf, nblocks, k, c, first, d_last_gamma, st);     // the product up to 4ab23f3
    case 22: return launch_ctr_t<BeltTabTwoQ, 1, 7>(d_buf, nblocks, k, c, first, d_last_gamma, st);    // 20 with one s_waitcnt per G-box
    case 20: return launch_ctr_t<BeltTabTwoP, 1, 7>(d_buf, nblocks, k, c, first, d_last_gamma, st);    // 18 with three address-register sets
    default: break;
    }
#endif
    // product (round 3): contiguous tile ranges + non-temporal stream accesses, +1 % (profiles/r03_belt_mem_ab.txt); the table
    // type CtrTab = BeltTabTwoP (one-instruction LDS addresses), +15 % (profiles/r03_belt_sdwa_ab.txt)
    {
        // block i uses ctr0 + ((first + 1 + i) mod 2^64): the upper 64 bits of that sum stay put over the launch unless the
        // lower 64 bits wrap inside it or the 64-bit offset itself does (then the carry into the upper half goes away again)
        const uint64_t lo = ((uint64_t)c.c[1] << 32) | c.c[0];
        const uint64_t a0 = first + 1, a1 = a0 + (nblocks - 1);
        const uint64_t s0 = lo + a0, e0 = s0 + (nblocks - 1);
        if (e0 >= s0 && a1 >= a0) return launch_ctr_t<CtrTab, CTR_ILP, 7>(d_buf, nblocks, k, c, first, d_last_gamma, st);
        return launch_ctr_t<CtrTab, CTR_ILP, 3>(d_buf, nblocks, k, c, first, d_last_gamma, st);
    }
}

template <int MODE>
static err_t launch_modes_t(const void *d_src, void *d_dst, size_t nblocks, const BeltKey &k, const BeltCtr &iv,
                            hipStream_t st)
{
    auto kern = belt_modes_kernel<MODE>;
    B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(kern), CtrTab::kBytes));
    size_t grid = (nblocks + CTR_WG - 1) / CTR_WG;
    const size_t cap = (size_t)num_cus() * (BeltTabWide::kBytes / CtrTab::kBytes);
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CTR_WG), CtrTab::kBytes, st, (const uint4 *)d_src,
                       (uint4 *)d_dst, nblocks, k, iv);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// mode: 0 ECB encrypt, 1 ECB decrypt (src may equal dst), 2 CBC decrypt (src != dst)
err_t launch_belt_modes(int mode, const void *d_src, void *d_dst, size_t nblocks, const uint32_t key[8],
                        const uint32_t iv[4], hipStream_t st)
{
    if (nblocks == 0) return ERR_OK;
    BeltKey k; BeltCtr c;
    for (int i = 0; i < 8; ++i) k.k[i] = key[i];
    for (int i = 0; i < 4; ++i) c.c[i] = iv ? iv[i] : 0u;
    if (mode == 0) return launch_modes_t<0>(d_src, d_dst, nblocks, k, c, st);
    if (mode == 1) return launch_modes_t<1>(d_src, d_dst, nblocks, k, c, st);
    if (mode == 2) { if (d_src == d_dst) return ERR_BAD_INPUT; return launch_modes_t<2>(d_src, d_dst, nblocks, k, c, st); }
    return ERR_BAD_INPUT;
}

template <int DECR>
static err_t launch_bde_t(const void *d_src, void *d_dst, uint64_t nblocks, uint64_t chunk, unsigned grid,
                          const BeltKey &k, const uint4 *tweaks, hipStream_t st)
{
    auto kern = belt_bde_kernel<DECR>;
    B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(kern), CtrTab::kBytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CTR_WG), CtrTab::kBytes, st, (const uint4 *)d_src, (uint4 *)d_dst,
                       nblocks, chunk, k, tweaks);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// belt-bde over nblocks whole blocks that start `first` blocks after beltBDEStart (s = E_K(iv), as
// words); d_src may equal d_dst.  d_s_out (may be null) receives s * x^(first + nblocks), 16 bytes.
err_t launch_belt_bde(int decr, const void *d_src, void *d_dst, size_t nblocks, const uint32_t key[8],
                      const uint32_t s[4], uint64_t first, void *d_s_out, hipStream_t st)
{
    if (nblocks == 0 && !d_s_out) return ERR_OK;
    BeltKey k; BeltCtr s0;
    for (int i = 0; i < 8; ++i) k.k[i] = key[i];
    for (int i = 0; i < 4; ++i) s0.c[i] = s[i];
    // persistent grid as for CTR; every wavefront gets one contiguous chunk (a multiple of 64 blocks)
    const size_t wg_waves = CTR_WG / 64;
    size_t grid = (nblocks + CTR_WG - 1) / CTR_WG;
    const size_t cap = (size_t)num_cus() * (BeltTabWide::kBytes / CtrTab::kBytes);
    if (grid > cap) grid = cap;
    if (grid == 0) grid = 1;
    const uint64_t nwaves = grid * wg_waves;
    uint64_t chunk = (nblocks + nwaves - 1) / nwaves;
    chunk = (chunk + 63) / 64 * 64;
    if (chunk == 0) chunk = 64;
    void *tw = nullptr;
    err_t code = scratch_for_stream(st, 8, (nwaves + 1) * 16, &tw);
    if (code != ERR_OK) return code;
    hipLaunchKernelGGL(bde_tweak_kernel, dim3((unsigned)((nwaves + 1 + 63) / 64)), dim3(64), 0, st, s0,
                       (uint64_t)first, chunk, (uint64_t)nblocks, (unsigned)nwaves, (uint4 *)tw);
    B2H_TRY(hipGetLastError());
    if (nblocks) {
        code = decr ? launch_bde_t<1>(d_src, d_dst, nblocks, chunk, (unsigned)grid, k, (const uint4 *)tw, st)
                    : launch_bde_t<0>(d_src, d_dst, nblocks, chunk, (unsigned)grid, k, (const uint4 *)tw, st);
        if (code != ERR_OK) return code;
    }
    if (d_s_out)
        B2H_TRY(hipMemcpyAsync(d_s_out, (const uint4 *)tw + nwaves, 16, hipMemcpyDeviceToDevice, st));
    return ERR_OK;
}

// nsectors sectors of nblk >= 2 whole blocks each, contiguous, transformed in place; d_ivs = nsectors x 16 bytes
err_t launch_belt_sde(int decr, void *d_sectors, size_t nblk, size_t nsectors, const uint32_t key[8],
                      const void *d_ivs, hipStream_t st)
{
    if (nsectors == 0) return ERR_OK;
    if (nblk < 2 || nblk > 0x7fffffffull) return ERR_BAD_INPUT;
    const void *kern = decr ? reinterpret_cast<const void *>(belt_sde_kernel<1>)
                            : reinterpret_cast<const void *>(belt_sde_kernel<0>);
    B2H_TRY(dyn_lds_once(kern, CtrTab::kBytes));
    BeltKey k;
    for (int i = 0; i < 8; ++i) k.k[i] = key[i];
    const unsigned grid = (unsigned)((nsectors + CTR_WG - 1) / CTR_WG);
    constexpr int SDE_LINE = 8;                                    // 128-byte lines: 4-block lines measured slower, esp. decryption
    if (nblk % SDE_LINE == 0 && nblk >= 2 * SDE_LINE) {             // whole-line accesses per lane (see the kernel)
        const void *lk = decr ? reinterpret_cast<const void *>(belt_sde_lines_kernel<1, SDE_LINE>)
                              : reinterpret_cast<const void *>(belt_sde_lines_kernel<0, SDE_LINE>);
        B2H_TRY(dyn_lds_once(lk, CtrTab::kBytes));
        if (decr)
            hipLaunchKernelGGL((belt_sde_lines_kernel<1, SDE_LINE>), dim3(grid), dim3(CTR_WG), CtrTab::kBytes, st,
                               (uint4 *)d_sectors, (uint32_t)nblk, (uint64_t)nsectors, k, (const uint4 *)d_ivs);
        else
            hipLaunchKernelGGL((belt_sde_lines_kernel<0, SDE_LINE>), dim3(grid), dim3(CTR_WG), CtrTab::kBytes, st,
                               (uint4 *)d_sectors, (uint32_t)nblk, (uint64_t)nsectors, k, (const uint4 *)d_ivs);
    } else if (decr)
        hipLaunchKernelGGL(belt_sde_kernel<1>, dim3(grid), dim3(CTR_WG), CtrTab::kBytes, st, (uint4 *)d_sectors,
                           (uint32_t)nblk, (uint64_t)nsectors, k, (const uint4 *)d_ivs);
    else
        hipLaunchKernelGGL(belt_sde_kernel<0>, dim3(grid), dim3(CTR_WG), CtrTab::kBytes, st, (uint4 *)d_sectors,
                           (uint32_t)nblk, (uint64_t)nsectors, k, (const uint4 *)d_ivs);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// belt-che keystream over nblocks whole blocks that start `first` blocks after beltCHEStart
// (s = the state's s as words); d_src may equal d_dst.  d_s_out (may be null) receives the state
// after the piece, 16 bytes.
err_t launch_belt_che(const void *d_src, void *d_dst, size_t nblocks, const uint32_t key[8], const uint32_t s[4],
                      uint64_t first, void *d_s_out, hipStream_t st)
{
    if (nblocks == 0 && !d_s_out) return ERR_OK;
    B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(belt_che_kernel), CtrTab::kBytes));
    BeltKey k; BeltCtr s0;
    for (int i = 0; i < 8; ++i) k.k[i] = key[i];
    for (int i = 0; i < 4; ++i) s0.c[i] = s[i];
    const size_t wg_waves = CTR_WG / 64;
    size_t grid = (nblocks + CTR_WG - 1) / CTR_WG;
    const size_t cap = (size_t)num_cus() * (BeltTabWide::kBytes / CtrTab::kBytes);
    if (grid > cap) grid = cap;
    if (grid == 0) grid = 1;
    const uint64_t nwaves = grid * wg_waves;
    uint64_t chunk = (nblocks + nwaves - 1) / nwaves;
    chunk = (chunk + 63) / 64 * 64;
    if (chunk == 0) chunk = 64;
    void *sv = nullptr;
    err_t code = scratch_for_stream(st, 10, (nwaves + 1) * 16, &sv);
    if (code != ERR_OK) return code;
    hipLaunchKernelGGL(che_state_kernel, dim3((unsigned)((nwaves + 1 + 63) / 64)), dim3(64), 0, st, s0,
                       (uint64_t)first, chunk, (uint64_t)nblocks, (unsigned)nwaves, (uint4 *)sv);
    if (nblocks)
        hipLaunchKernelGGL(belt_che_kernel, dim3((unsigned)grid), dim3(CTR_WG), CtrTab::kBytes, st,
                           (const uint4 *)d_src, (uint4 *)d_dst, (uint64_t)nblocks, chunk, k, (const uint4 *)sv);
    B2H_TRY(hipGetLastError());
    if (d_s_out)
        B2H_TRY(hipMemcpyAsync(d_s_out, (const uint4 *)sv + nwaves, 16, hipMemcpyDeviceToDevice, st));
    return ERR_OK;
}

// d_t_out (16 bytes, device) <- t after absorbing the nbytes at d_data as 16-byte blocks, the last
// one zero-padded; r, t as u32 words.  nbytes == 0 gives t itself.
err_t launch_belt_polyhash(const void *d_data, size_t nbytes, const uint32_t r[4], const uint32_t t[4],
                           void *d_t_out, hipStream_t st)
{
    BeltCtr rr, tt;
    for (int i = 0; i < 4; ++i) { rr.c[i] = r[i]; tt.c[i] = t[i]; }
    const uint64_t nblocks = ((uint64_t)nbytes + 15) / 16;
    // enough wavefronts to fill the chip, chunks of at least 64 x 16 blocks so that the per-chunk
    // constants (table build, one general product per lane) stay below ~5 % of the Horner loop
    size_t nwaves = (size_t)num_cus() * 32;
    uint64_t chunk = (nblocks + nwaves - 1) / (nwaves ? nwaves : 1);
    if (chunk < 1024) chunk = 1024;
    chunk = (chunk + 63) / 64 * 64;
    nwaves = (size_t)((nblocks + chunk - 1) / chunk);
    void *scr = nullptr;
    err_t code = scratch_for_stream(st, 9, (65 + nwaves + 1) * 16, &scr);
    if (code != ERR_OK) return code;
    uint4 *consts = (uint4 *)scr, *partial = consts + 65;
    B2H_TRY(hipMemsetAsync(d_t_out, 0, 16, st));
    if (nwaves) {
        hipLaunchKernelGGL(polyhash_prep_kernel, dim3(1), dim3(128), 0, st, rr, consts);
        const size_t wg_waves = PH_WG / 64;
        hipLaunchKernelGGL(polyhash_kernel, dim3((unsigned)((nwaves + wg_waves - 1) / wg_waves)), dim3(PH_WG), 0, st,
                           (const uint8_t *)d_data, (uint64_t)nbytes, chunk, (const uint4 *)consts, partial);
    }
    hipLaunchKernelGGL(polyhash_finish_kernel, dim3((unsigned)((nwaves + 1 + 63) / 64)), dim3(64), 0, st,
                       (const uint4 *)partial, (unsigned)nwaves, chunk, nblocks, rr, tt, (uint32_t *)d_t_out);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

err_t launch_belt_cbc_encr(void *d_msgs, size_t nblk, size_t n, const uint32_t key[8], void *d_ivs, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    BeltKey k;
    for (int i = 0; i < 8; ++i) k.k[i] = key[i];
    B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(belt_cbc_encr_kernel), CtrTab::kBytes));
    hipLaunchKernelGGL(belt_cbc_encr_kernel, dim3((unsigned)((n + CTR_WG - 1) / CTR_WG)), dim3(CTR_WG), CtrTab::kBytes, st,
                       (uint4 *)d_msgs, nblk, n, k, (uint4 *)d_ivs);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

err_t launch_belt_decr_blocks(void *d_blocks, size_t nblocks, const uint32_t key[8], hipStream_t st)
{
    if (nblocks == 0) return ERR_OK;
    BeltKey k;
    for (int i = 0; i < 8; ++i) k.k[i] = key[i];
    hipLaunchKernelGGL(belt_decr_blocks_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, st,
                       (uint4 *)d_blocks, nblocks, k);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

err_t launch_belt_encr_blocks(void *d_blocks, size_t nblocks, const uint32_t key[8], hipStream_t st)
{
    if (nblocks == 0) return ERR_OK;
    BeltKey k;
    for (int i = 0; i < 8; ++i) k.k[i] = key[i];
    const size_t grid = (nblocks + 63) / 64;
    if (grid > 0x7fffffffull) return ERR_BAD_INPUT;
    hipLaunchKernelGGL(belt_encr_blocks_kernel, dim3((unsigned)grid), dim3(64), 0, st,
                       (uint4 *)d_blocks, nblocks, k);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

}  // namespace bee2hip
