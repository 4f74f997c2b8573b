// mixed_kernels.hip -- bash sponge hashing and belt MAC over batches of messages (gfx950).
//
// H4 of SURVEY.md 8a: per message a bashHash(l) digest (src/crypto/bash/bash_hash.c:38-137)
// and a beltMAC tag (src/crypto/belt/belt_mac.c:47-203).  Both are strictly serial per
// message (sponge chain / CBC-MAC chain), so the parallel axis is the message: one lane
// per message, 64 messages per wavefront.
//
//  hash_mac_fused_kernel<RW,..> : the batch kernel.  Message length a multiple of 16,
//      level l in {128,192,256} (bash256/384/512: rate RW = 16/12/8 u64 words).  Each lane
//      streams its own message with 16-byte loads, prefetching the next rate block while
//      the current one is absorbed.  Per-lane strided loads are fine here: the path does
//      ~145 integer ops per byte, so the TA cost of 64 distinct lines per load is noise
//      and every fetched line is consumed in full by its lane (lines stay in L2 between
//      the two halves).  The MAC's belt tables are the bank-private 128 KiB LDS layout
//      (belt_dev.hpp): with 257 E_K per 4 KiB message the LDS lookup pipe is ~40 % busy
//      even conflict-free, so conflicts would make it the bottleneck.
//      Algorithmic HBM bytes: msg_len read once + l/4 + 8 written per message.
//
//  bash_sponge_kernel / belt_mac_kernel : generic lane-per-state forms (any level, any
//      byte count, state resident in memory in bee2's own layout).  They back the drop-in
//      streaming API (bashHashStepH / beltMACStepA / StepG) and batches whose shape the
//      fused kernel does not take.
#include "bash_dev.hpp"
#include "belt_dev.hpp"
#include "common.hpp"

namespace bee2hip {

struct MacKey { uint32_t k[8]; };

// phi1 / phi2 tweaks of the last block (belt_mac.c:112-115,129-132)
__device__ __forceinline__ void mac_phi1(uint32_t (&m)[4], const uint32_t (&r)[4])
{
    m[0] ^= r[1]; m[1] ^= r[2]; m[2] ^= r[3]; m[3] ^= r[0] ^ r[1];
}
__device__ __forceinline__ void mac_phi2(uint32_t (&m)[4], const uint32_t (&r)[4])
{
    m[0] ^= r[0] ^ r[3]; m[1] ^= r[0]; m[2] ^= r[1]; m[3] ^= r[2];
}

constexpr int FUSED_WG = 1024;
// issue order of bash-f inside the fused kernel (bash_dev.hpp bash_round): A/B builds override it
#ifndef BASH_FUSED_ORDER
#define BASH_FUSED_ORDER 101
#endif

template <int RW, bool HASH, bool MAC, class Tab = BeltTabWide>
__global__ __launch_bounds__(FUSED_WG)
void hash_mac_fused_kernel(const uint4 *__restrict__ msgs, size_t msg_len, size_t n, uint32_t level,
                           MacKey key, uint8_t *__restrict__ digests, uint8_t *__restrict__ tags)
{
    constexpr int RB = RW / 2;                   // 16-byte blocks per rate block
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (MAC) {
        Tab::fill(smem, threadIdx.x, FUSED_WG);
        __syncthreads();
    }
    const Tab T(smem);
    const size_t idx = (size_t)blockIdx.x * FUSED_WG + threadIdx.x;
    if (idx >= n) return;

    const size_t nb16 = msg_len / 16;            // 16-byte blocks per message
    const uint4 *m = msgs + idx * nb16;
    const size_t nchunks = nb16 / RB;            // full rate blocks
    const int tail = (int)(nb16 % RB);           // leftover 16-byte blocks

    // ---- states
    u64x2 a[24];
    if (HASH) {
#pragma unroll
        for (int i = 0; i < 24; ++i) a[i].lo = a[i].hi = 0;
        a[23].lo = level / 4;                    // s[192 - 8] = l / 4 (bash_hash.c:43-45)
    }
    uint32_t K[8], s[4] = {0, 0, 0, 0}, r[4] = {0, 0, 0, 0}, last[4] = {0, 0, 0, 0};
    if (MAC) {
#pragma unroll
        for (int i = 0; i < 8; ++i) K[i] = key.k[i];
        belt_encr(T, r, K);                      // r = E_K(0) (belt_mac.c:52-54)
    }

    // ---- full rate blocks, software-prefetched one block ahead
    uint4 nxt[RB];
    if (nchunks) {
#pragma unroll
        for (int j = 0; j < RB; ++j) nxt[j] = m[j];
    }
#pragma unroll 1
    for (size_t c = 0; c < nchunks; ++c) {
        uint4 x[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) x[j] = nxt[j];
        if (c + 1 < nchunks) {
#pragma unroll
            for (int j = 0; j < RB; ++j) nxt[j] = m[(c + 1) * RB + j];
        }
        if (MAC) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const size_t bi = c * RB + j;
                if (bi + 1 < nb16) {             // every block but the last: s = E_K(s ^ X)
                    s[0] ^= x[j].x; s[1] ^= x[j].y; s[2] ^= x[j].z; s[3] ^= x[j].w;
                    belt_encr(T, s, K);
                } else {
                    last[0] = x[j].x; last[1] = x[j].y; last[2] = x[j].z; last[3] = x[j].w;
                }
            }
        }
        if (HASH) {
            // absorb = overwrite the first RW words (bash_hash.c:69-75)
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                a[2 * j].lo = x[j].x; a[2 * j].hi = x[j].y;
                a[2 * j + 1].lo = x[j].z; a[2 * j + 1].hi = x[j].w;
            }
            bash_f<BASH_FUSED_ORDER>(a);   // staged order + class-following priority (bash_dev.hpp)
        }
    }
    // ---- tail blocks + padding (uniform across the batch: msg_len is)
    {
        uint4 x[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            x[j] = make_uint4(0, 0, 0, 0);
            if (j < tail) x[j] = m[nchunks * RB + j];
        }
        if (MAC) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                if (j < tail) {
                    const size_t bi = nchunks * RB + j;
                    if (bi + 1 < nb16) {
                        s[0] ^= x[j].x; s[1] ^= x[j].y; s[2] ^= x[j].z; s[3] ^= x[j].w;
                        belt_encr(T, s, K);
                    } else {
                        last[0] = x[j].x; last[1] = x[j].y; last[2] = x[j].z; last[3] = x[j].w;
                    }
                }
            }
            uint32_t mac[4];
            if (nb16) {                          // full last block (belt_mac.c:105-119)
#pragma unroll
                for (int i = 0; i < 4; ++i) mac[i] = s[i] ^ last[i];
                mac_phi1(mac, r);
            } else {                             // empty message: block 80 00.. (belt_mac.c:121-136)
                mac[0] = s[0] ^ 0x80u; mac[1] = s[1]; mac[2] = s[2]; mac[3] = s[3];
                mac_phi2(mac, r);
            }
            belt_encr(T, mac, K);
            *reinterpret_cast<uint2 *>(tags + 8 * idx) = make_uint2(mac[0], mac[1]);
        }
        if (HASH) {
            // last block: tail || 0x40 || 0..  (bash_hash.c:81-102); pad word index = 2*tail
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                if (j == tail) x[j].x = 0x40u;
                a[2 * j].lo = x[j].x; a[2 * j].hi = x[j].y;
                a[2 * j + 1].lo = x[j].z; a[2 * j + 1].hi = x[j].w;
            }
            bash_f<BASH_FUSED_ORDER>(a);
            const int nw = (int)(level / 32);    // digest = l/4 bytes = l/32 words
            uint64_t *d = reinterpret_cast<uint64_t *>(digests + (size_t)(level / 4) * idx);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < nw) d[i] = ((uint64_t)a[i].hi << 32) | a[i].lo;
        }
    }
}

// ----------------------------------------------------------------- generic ---
// bash hash state in bee2's layout (bash_hash.c:25-31): s[192], s1[192], buf_len, pos
struct bash_hash_st {
    uint8_t s[192];
    uint8_t s1[192];
    size_t buf_len;
    size_t pos;
};

__device__ __forceinline__ void bashF_mem(uint8_t *s)
{
    u64x2 a[24];
    const uint32_t *w = reinterpret_cast<const uint32_t *>(s);
#pragma unroll
    for (int i = 0; i < 24; ++i) { a[i].lo = w[2 * i]; a[i].hi = w[2 * i + 1]; }
    bash_f(a);
    uint32_t *o = reinterpret_cast<uint32_t *>(s);
#pragma unroll
    for (int i = 0; i < 24; ++i) { o[2 * i] = a[i].lo; o[2 * i + 1] = a[i].hi; }
}

// item i absorbs `count` bytes at data + i*stride into states[i] (bashHashStepH,
// bash_hash.c:52-79).  fin != 0: also run the final padded permutation on the s1 copy
// (bashHashStepG_internal, bash_hash.c:81-102).
__global__ __launch_bounds__(64)
void bash_sponge_kernel(bash_hash_st *__restrict__ states, const uint8_t *__restrict__ data,
                        size_t stride, size_t count, size_t n, int fin)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    bash_hash_st *st = states + i;
    const uint8_t *p = data + i * stride;
    size_t pos = st->pos;
    const size_t buf_len = st->buf_len;
    for (size_t k = 0; k < count; ++k) {
        st->s[pos++] = p[k];
        if (pos == buf_len) { bashF_mem(st->s); pos = 0; }
    }
    st->pos = pos;
    if (fin) {
        for (int k = 0; k < 192; ++k) st->s1[k] = st->s[k];
        for (size_t k = pos; k < buf_len; ++k) st->s1[k] = 0;
        st->s1[pos] = 0x40;
        bashF_mem(st->s1);
    }
}

// Little-endian loads from any address (message starts are not aligned): aligned words and a funnel shift by
// the misalignment.  The extra word is read only when the address is misaligned, and then it holds at least
// one of the requested bytes -- an aligned word with a valid byte in it cannot fault.  Branch-free on purpose:
// the lanes of a wavefront hold messages of every alignment, and a wavefront that splits into an aligned and a
// misaligned path pays for both (ragged belt-hash 33 -> 23 GiB/s when tried).
__device__ __forceinline__ uint32_t load32_any(const uint8_t *p)
{
    const uint32_t *b = reinterpret_cast<const uint32_t *>((uintptr_t)p & ~(uintptr_t)3);
    const uint32_t sh = ((uint32_t)(uintptr_t)p & 3u) * 8u;
    const uint32_t lo = b[0], hi = sh ? b[1] : 0u;
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}
__device__ __forceinline__ uint64_t load64_any(const uint8_t *p)
{
    const uint32_t *b = reinterpret_cast<const uint32_t *>((uintptr_t)p & ~(uintptr_t)3);
    const uint32_t sh = ((uint32_t)(uintptr_t)p & 3u) * 8u;
    const uint32_t w0 = b[0], w1 = b[1], w2 = sh ? b[2] : 0u;
    return ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
}
// the first `valid` (1..3) bytes at p, zero above; never touches a word without a valid byte
__device__ __forceinline__ uint32_t load32_head(const uint8_t *p, uint32_t valid)
{
    const uint32_t *b = reinterpret_cast<const uint32_t *>((uintptr_t)p & ~(uintptr_t)3);
    const uint32_t mis = (uint32_t)(uintptr_t)p & 3u, sh = mis * 8u;
    const uint32_t lo = b[0], hi = (mis + valid > 4u) ? b[1] : 0u;
    return __builtin_amdgcn_alignbit(hi, lo, sh) & ((1u << (8u * valid)) - 1u);
}

// Whole rate blocks of ONE state with 8 lanes, one bash-f column each (bash_dev.hpp bash_f_cols): the bulk
// of a large bashHashStepH.  Precondition: st->pos == 0 (the host aligns to a block boundary with the
// byte-wise kernel first).  rate = st->buf_len bytes = RW words, RW <= 23; lane j absorbs words j, 8 + j
// and 16 + j that lie inside the rate.
__global__ __launch_bounds__(64)
void bash_sponge_cols_kernel(bash_hash_st *__restrict__ st, const uint8_t *__restrict__ data, size_t nblocks)
{
    if (threadIdx.x >= 8) return;
    const unsigned j = threadIdx.x;
    const unsigned rate = (unsigned)st->buf_len, rw = rate / 8;
    const BashCol C = bash_col_setup(threadIdx.x);
    u64x2 w[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint64_t v = load64_any(st->s + 8 * (8 * r + j));
        w[r].lo = (uint32_t)v; w[r].hi = (uint32_t)(v >> 32);
    }
    const uint8_t *p = data;
    for (size_t b = 0; b < nblocks; ++b, p += rate) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const unsigned k = 8u * r + j;
            if (k < rw) {
                const uint64_t v = load64_any(p + 8 * k);
                w[r].lo = (uint32_t)v; w[r].hi = (uint32_t)(v >> 32);
            }
        }
        bash_f_cols(w[0], w[1], w[2], C);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        uint8_t *d = st->s + 8 * (8 * r + j);
        const uint64_t v = ((uint64_t)w[r].hi << 32) | w[r].lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (uint8_t)(v >> (8 * k));
    }
}
err_t launch_bash_sponge_cols(void *d_state, const void *d_data, size_t nblocks, hipStream_t st)
{
    if (nblocks == 0) return ERR_OK;
    hipLaunchKernelGGL(bash_sponge_cols_kernel, dim3(1), dim3(64), 0, st, (bash_hash_st *)d_state,
                       (const uint8_t *)d_data, nblocks);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// belt MAC state in bee2's layout (belt_mac.c:32-40)
struct belt_mac_st {
    uint32_t key[8];
    uint32_t s[4];
    uint32_t r[4];
    uint32_t mac[4];
    uint8_t block[16];
    size_t filled;
};

// mode bit 0: start (s = 0, r = E_K(0), filled = 0; key already expanded in the state)
// mode bit 1: absorb `count` bytes (beltMACStepA, belt_mac.c:58-99)
// mode bit 2: finalise into st->mac without disturbing s/block (beltMACStepG_internal, :101-138)
__global__ __launch_bounds__(64)
void belt_mac_kernel(belt_mac_st *__restrict__ states, const uint8_t *__restrict__ data,
                     size_t stride, size_t count, size_t n, int mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[BeltTabSmall::kBytes];
    BeltTabSmall::fill(smem, threadIdx.x, 64);
    __syncthreads();
    const BeltTabSmall T(smem);
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    belt_mac_st *st = states + i;
    uint32_t K[8], s[4], r[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) K[k] = st->key[k];
    if (mode & 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] = 0; r[k] = 0; }
        belt_encr(T, r, K);
#pragma unroll
        for (int k = 0; k < 4; ++k) { st->s[k] = 0; st->r[k] = r[k]; }
        st->filled = 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = st->s[k]; r[k] = st->r[k]; }
    size_t filled = st->filled;
    if (mode & 2) {
        const uint8_t *p = data + i * stride;
        for (size_t k = 0; k < count; ++k) {
            if (filled == 16) {                  // absorb the buffered block only when more data follows
                const uint32_t *b = reinterpret_cast<const uint32_t *>(st->block);
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] ^= b[q];
                belt_encr(T, s, K);
                filled = 0;
            }
            st->block[filled++] = p[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) st->s[k] = s[k];
        st->filled = filled;
    }
    if (mode & 4) {
        uint32_t mac[4];
        if (filled == 16) {
            const uint32_t *b = reinterpret_cast<const uint32_t *>(st->block);
#pragma unroll
            for (int q = 0; q < 4; ++q) mac[q] = s[q] ^ b[q];
            mac_phi1(mac, r);
        } else {
            // pad 80 00.. in place, as bee2 does (belt_mac.c:123-124)
            st->block[filled] = 0x80;
            for (size_t k = filled + 1; k < 16; ++k) st->block[k] = 0;
            const uint32_t *b = reinterpret_cast<const uint32_t *>(st->block);
#pragma unroll
            for (int q = 0; q < 4; ++q) mac[q] = s[q] ^ b[q];
            mac_phi2(mac, r);
        }
        belt_encr(T, mac, K);
#pragma unroll
        for (int q = 0; q < 4; ++q) st->mac[q] = mac[q];
    }
}

// gather digests / tags of the generic batch path out of the per-item states
__global__ void gather_results_kernel(const bash_hash_st *hs, const belt_mac_st *ms, size_t n,
                                      size_t dlen, uint8_t *digests, uint8_t *tags)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (hs && digests)
        for (size_t k = 0; k < dlen; ++k) digests[i * dlen + k] = hs[i].s1[k];
    if (ms && tags)
        for (int k = 0; k < 8; ++k) tags[i * 8 + k] = (uint8_t)(ms[i].mac[k >> 2] >> (8 * (k & 3)));
}
__global__ void init_states_kernel(bash_hash_st *hs, belt_mac_st *ms, size_t n, uint32_t level, MacKey key)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (hs) {
        for (int k = 0; k < 192; ++k) hs[i].s[k] = 0;
        hs[i].s[184] = (uint8_t)(level / 4);
        hs[i].buf_len = 192 - level / 2;
        hs[i].pos = 0;
    }
    if (ms)
        for (int k = 0; k < 8; ++k) ms[i].key[k] = key.k[k];
}

// ------------------------------------------------ 8f-3: ragged batches of messages ---
// n messages of different lengths packed back to back: message i = data[off[i] .. off[i+1]).
// One lane per message (lanes of a wavefront finish at different times; that is the nature of
// ragged input).  This is the device side of a `bee2cmd bsum`-style front-end that hashes many
// files per launch instead of one file per bashHashStepH loop (cmd/bsum/bsum.c:133-221).

// ALG = 8 / 12 / 16: bash512 / bash384 / bash256 (rate in u64 words); digests are l/4 bytes each.
// (round 4, second session) ONE absorb-and-permute site: a block is read as the aligned 32-bit words that hold its octets (never a
// word without an octet of the message), shifted into place by v_alignbit, and the LAST block -- shorter than the rate, possibly empty
// -- gets its zeros and the 0x40 by masks on the same words, so the loop body serves every block and bash-f is in the kernel once,
// in the staged order with the class-following priority of the fused kernel.  Before: two copies of the compact bash-f, octet loads
// for the tail, 256-276 VGPRs with 40 of them spilled (two wavefronts per SIMD); now 4 wavefronts per SIMD without scratch.
#ifndef BASH_RAGGED_WAVES
#define BASH_RAGGED_WAVES 4
#endif
template <int RW>
__global__ __launch_bounds__(64, BASH_RAGGED_WAVES)
void bash_ragged_kernel(const uint8_t *__restrict__ data, const uint64_t *__restrict__ off,
                        const uint32_t *__restrict__ order, size_t n,
                        uint32_t level, uint8_t *__restrict__ digests, uint64_t long_from)
{
    const size_t slot = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (slot >= n) return;
    const size_t i = order ? order[slot] : slot;      // lane `slot` hashes message order[slot]
    const uint8_t *p = data + off[i];
    size_t left = (size_t)(off[i + 1] - off[i]);
    if (left >= long_from) return;                    // long messages: bash_long_kernel, 8 lanes each
    constexpr uint32_t RATE = 8 * RW, NW = 2 * RW;
    u64x2 a[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) a[k].lo = a[k].hi = 0;
    a[23].lo = level / 4;
    // 16-octet loads: a lane's block sits in its own cache lines, so every load instruction of the wavefront is 64 separate line
    // accesses whatever its width -- 9 of them per 128-octet block instead of 33 dword loads (the texture path, not the VALU, bounded the
    // dword form: 718 GiB/s on 1000-octet messages where bash-f alone allows ~1.2 TiB/s).  The block starts mis16 = p mod 16 octets into
    // the first quad: a dword shift d = mis16 / 4 (two rounds of selects) and a bit shift (v_alignbit) bring it into place.
    const uint4 *qp = reinterpret_cast<const uint4 *>((uintptr_t)p & ~(uintptr_t)15);
    const uint32_t mis16 = (uint32_t)(uintptr_t)p & 15u, sh = (mis16 & 3u) * 8u;
    const uint32_t m1 = (mis16 & 4u) ? ~0u : 0u, m2 = (mis16 & 8u) ? ~0u : 0u;
    constexpr uint32_t NQ = NW / 4 + 1;                                 // quads that can hold octets of one block
    for (;;) {
        const uint32_t cnt = left < RATE ? (uint32_t)left : RATE;       // octets of this block (the last one: 0 .. RATE - 1)
        const uint32_t span = mis16 + cnt;                              // aligned quad j holds octets of the block iff 16 j < span
        uint32_t W[4 * NQ + 4];
#pragma unroll
        for (uint32_t j = 0; j < NQ; ++j) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (16u * j < span) v = qp[j];
            W[4 * j] = v.x; W[4 * j + 1] = v.y; W[4 * j + 2] = v.z; W[4 * j + 3] = v.w;
        }
#pragma unroll
        for (uint32_t j = 4 * NQ; j < 4 * NQ + 4; ++j) W[j] = 0;
        // (selects by MASK, one v_bitop3 each: written as `d ? W[j + 1] : W[j]` the compiler selects the ADDRESS and parks W in scratch)
        uint32_t A[NW + 3], B[NW + 1];
#pragma unroll
        for (uint32_t j = 0; j < NW + 3; ++j) A[j] = __builtin_amdgcn_bitop3_b32(W[j], W[j + 1], m1, 0xD8);       // m1 ? W[j + 1] : W[j]
#pragma unroll
        for (uint32_t j = 0; j <= NW; ++j) B[j] = __builtin_amdgcn_bitop3_b32(A[j], A[j + 2], m2, 0xD8);
        uint32_t x[NW];
#pragma unroll
        for (uint32_t j = 0; j < NW; ++j) x[j] = __builtin_amdgcn_alignbit(B[j + 1], B[j], sh);
        if (cnt < RATE) {
            // tail || 0x40 || 0.. (bash_hash.c:84-102): word j keeps its first r = cnt - 4 j octets, 0x40 follows the last octet
#pragma unroll
            for (uint32_t j = 0; j < NW; ++j) {
                const int32_t r = (int32_t)cnt - (int32_t)(4u * j);
                const uint32_t keep = r >= 4 ? ~0u : r <= 0 ? 0u : (1u << (8 * r)) - 1u;
                const uint32_t pad = (r >= 0 && r < 4) ? 0x40u << (8 * r) : 0u;
                x[j] = (x[j] & keep) | pad;
            }
        }
        // absorb = overwrite the first RW words (bash_hash.c:69-75)
#pragma unroll
        for (int w = 0; w < RW; ++w) { a[w].lo = x[2 * w]; a[w].hi = x[2 * w + 1]; }
        bash_f<BASH_FUSED_ORDER>(a);
        if (cnt < RATE) break;
        qp += NW / 4; left -= RATE;
    }
    uint8_t *d = digests + (size_t)(level / 4) * i;
    const int nw = (int)(level / 32);
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        if (w < nw) {
            const uint64_t v = ((uint64_t)a[w].hi << 32) | a[w].lo;
#pragma unroll
            for (int k = 0; k < 8; ++k) d[8 * w + k] = (uint8_t)(v >> (8 * k));
        }
    }
}

// Long messages (>= long_from bytes): 8 lanes per message, one bash-f column each (bash_dev.hpp
// bash_f_cols).  Launched over ALL n messages; groups whose message is short leave at once.
// RW = rate in 64-bit words (16 / 12 / 8 for bash256 / 384 / 512): lane j absorbs words j and 8 + j.
template <int RW>
__global__ __launch_bounds__(64)
void bash_long_kernel(const uint8_t *__restrict__ data, const uint64_t *__restrict__ off,
                      const uint32_t *__restrict__ order, size_t n, uint32_t level,
                      uint8_t *__restrict__ digests, uint64_t long_from)
{
    const size_t slot = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 3;
    const unsigned j = threadIdx.x & 7u;
    if (slot >= n) return;
    // a wavefront runs as long as the longest of its 8 messages: with a longest-first order the groups of
    // a wavefront hold similar lengths (and wavefronts of short messages leave as a whole)
    const size_t g = order ? order[slot] : slot;
    const uint8_t *p = data + off[g];
    size_t left = (size_t)(off[g + 1] - off[g]);
    if (left < long_from) return;
    constexpr int RATE = 8 * RW;
    const BashCol C = bash_col_setup(threadIdx.x);
    u64x2 w0 = {0, 0}, w1 = {0, 0}, w2 = {0, 0};
    if (j == 7) w2.lo = level / 4;                     // s[184] = l / 4 (bash_hash.c:43-45): word 23
    // the words of the next block are fetched before the permutation of the current one: a block is a
    // step of the serial chain and its load latency would otherwise sit on it
    const bool row1 = 8 + j < (unsigned)RW;            // row 0 is always inside the rate (RW >= 8)
    uint64_t n0 = 0, n1 = 0;
    if (left >= (size_t)RATE) { n0 = load64_any(p + 8 * j); if (row1) n1 = load64_any(p + 8 * (8 + j)); }
    while (left >= (size_t)RATE) {
        w0.lo = (uint32_t)n0; w0.hi = (uint32_t)(n0 >> 32);
        if (row1) { w1.lo = (uint32_t)n1; w1.hi = (uint32_t)(n1 >> 32); }
        p += RATE; left -= RATE;
        if (left >= (size_t)RATE) { n0 = load64_any(p + 8 * j); if (row1) n1 = load64_any(p + 8 * (8 + j)); }
        bash_f_cols(w0, w1, w2, C);
    }
    // last block: tail || 0x40 || 0..  (bash_hash.c:81-102)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned k = 8u * r + j;
        if (k < (unsigned)RW) {
            uint64_t v = 0;
#pragma unroll
            for (int b = 7; b >= 0; --b) {
                const size_t pos = (size_t)(8 * k + b);
                const uint32_t x = pos < left ? p[pos] : (pos == left ? 0x40u : 0u);
                v = (v << 8) | x;
            }
            if (r == 0) { w0.lo = (uint32_t)v; w0.hi = (uint32_t)(v >> 32); }
            else        { w1.lo = (uint32_t)v; w1.hi = (uint32_t)(v >> 32); }
        }
    }
    bash_f_cols(w0, w1, w2, C);
    const unsigned nw = level / 32;                    // digest = l / 4 bytes = l / 32 words, all in row 0
    if (j < nw) {
        uint8_t *d = digests + (size_t)(level / 4) * g + 8 * j;
        const uint64_t v = ((uint64_t)w0.hi << 32) | w0.lo;
#pragma unroll
        for (int b = 0; b < 8; ++b) d[b] = (uint8_t)(v >> (8 * b));
    }
}

// belt-hash of ragged messages (src/crypto/belt/belt_hash.c:43-171): 32-byte digests, one lane per message.
// Tab / WG: BeltTabSmall in 64-thread workgroups (4 KiB table, data-dependent bank conflicts) for small batches,
// BeltTabTwo in 1024-thread workgroups (the CTR kernel's conflict-free 64 KiB table) once its fill pays.
template <class Tab, int WG>
__global__ __launch_bounds__(WG)
void belt_hash_ragged_kernel(const uint8_t *__restrict__ data, const uint64_t *__restrict__ off,
                             const uint32_t *__restrict__ order, size_t n, uint8_t *__restrict__ digests,
                             uint64_t long_from)
{
    // (the 4 KiB table in STATIC shared memory: look-up addresses with immediate offsets -- see belt_hash_long_kernel)
    constexpr bool STATIC_TAB = Tab::kBytes <= 4096;
    __shared__ __attribute__((aligned(16))) uint8_t smem_static[STATIC_TAB ? Tab::kBytes : 16];
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_dyn[];
    uint8_t *smem = STATIC_TAB ? smem_static : smem_dyn;
    Tab::fill(smem, threadIdx.x, WG);
    __syncthreads();
    const Tab T(smem);
    const size_t slot = (size_t)blockIdx.x * WG + threadIdx.x;
    if (slot >= n) return;
    const size_t i = order ? order[slot] : slot;
    const uint8_t *p = data + off[i];
    const size_t len = (size_t)(off[i + 1] - off[i]);
    if (len >= long_from) return;                      // long messages: belt_hash_long_kernel, 2 lanes each
    size_t left = len;
    uint32_t h[8], s[4] = {0, 0, 0, 0}, X[8], s1[4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        h[k] = (uint32_t)c_beltH[4 * k] | (uint32_t)c_beltH[4 * k + 1] << 8 |
               (uint32_t)c_beltH[4 * k + 2] << 16 | (uint32_t)c_beltH[4 * k + 3] << 24;
    // (round 4, second session) ONE compression site for the whole blocks, the zero-padded last one and the closing block
    // <bit length> || s (belt_hash.c:115-135): three inlined copies of the three encryptions were 86 KB of code for a 64 KB
    // instruction cache.  A block is read as the aligned 16-octet quads that hold its octets (a lane's load instruction is 64 line
    // accesses whatever its width: 3 instead of 9 per block; never a quad without an octet of the message), brought into place by
    // a dword shift (mask selects) and v_alignbit -- as bash_ragged_kernel.
    const uint4 *qp = reinterpret_cast<const uint4 *>((uintptr_t)p & ~(uintptr_t)15);
    const uint32_t mis16 = (uint32_t)(uintptr_t)p & 15u, sh = (mis16 & 3u) * 8u;
    const uint32_t m1 = (mis16 & 4u) ? ~0u : 0u, m2 = (mis16 & 8u) ? ~0u : 0u;
    const size_t nblk = (len + 31) / 32;               // data blocks, the partial one included
#pragma unroll 1
    for (size_t blk = 0; blk <= nblk; ++blk) {
        const bool is_data = blk < nblk;
        if (is_data) {
            const uint32_t cnt = left < 32 ? (uint32_t)left : 32u;
            const uint32_t span = mis16 + cnt;          // aligned quad j holds octets of the block iff 16 j < span
            uint32_t W[16];
#pragma unroll
            for (uint32_t j = 0; j < 3; ++j) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (16u * j < span) v = qp[j];
                W[4 * j] = v.x; W[4 * j + 1] = v.y; W[4 * j + 2] = v.z; W[4 * j + 3] = v.w;
            }
            W[12] = W[13] = W[14] = W[15] = 0;
            uint32_t A[11], B[9];
#pragma unroll
            for (uint32_t j = 0; j < 11; ++j) A[j] = __builtin_amdgcn_bitop3_b32(W[j], W[j + 1], m1, 0xD8);     // m1 ? W[j + 1] : W[j]
#pragma unroll
            for (uint32_t j = 0; j < 9; ++j) B[j] = __builtin_amdgcn_bitop3_b32(A[j], A[j + 2], m2, 0xD8);
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const int32_t r = (int32_t)cnt - (int32_t)(4u * j);                 // octets of word j that belong to the message
                const uint32_t keep = r >= 4 ? ~0u : r <= 0 ? 0u : (1u << (8 * r)) - 1u;
                X[j] = __builtin_amdgcn_alignbit(B[j + 1], B[j], sh) & keep;
            }
            qp += 2; left -= cnt;
        } else {
            const uint64_t bits_lo = (uint64_t)len << 3, bits_hi = (uint64_t)len >> 61;
            X[0] = (uint32_t)bits_lo; X[1] = (uint32_t)(bits_lo >> 32); X[2] = (uint32_t)bits_hi; X[3] = 0;
            X[4] = s[0]; X[5] = s[1]; X[6] = s[2]; X[7] = s[3];
        }
        belt_compress(T, s1, h, X);
        if (is_data) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] ^= s1[k];
        }
    }
    uint32_t *d = reinterpret_cast<uint32_t *>(digests + 32 * i);      // 32-byte slots of an aligned buffer
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = h[k];
}

// Streaming belt-hash for the drop-in beltHashStep* (belt_hash.c:43-171): one serial chain, run by a pair of
// lanes (belt_compress_pair: 2 E per block on the chain instead of 3).
// hs = h[8] || s[4] (in / out); nblocks whole 32-byte blocks at data; with fin != 0 the block
// <bit length>_128 || s is compressed as well (belt_hash.c:120-135) and hs[0..8) is the digest.
__global__ __launch_bounds__(64)
void belt_hash_stream_kernel(uint32_t *__restrict__ hs, const uint8_t *__restrict__ data, size_t nblocks, int fin,
                             uint64_t bits_lo, uint64_t bits_hi)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[BeltTabSmall::kBytes];
    BeltTabSmall::fill(smem, threadIdx.x, 64);
    __syncthreads();
    const BeltTabSmall T(smem);
    if (threadIdx.x >= 2 || blockIdx.x != 0) return;                    // a pair of lanes: belt_compress_pair
    const uint32_t odd = threadIdx.x ? ~0u : 0u;
    uint32_t h[8], s[4], X[8], s1[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = hs[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = hs[8 + k];
    const uint32_t *w = reinterpret_cast<const uint32_t *>(data);           // scratch buffer: 4-byte aligned
    uint32_t Xn[8];
    if (nblocks) {
#pragma unroll
        for (int k = 0; k < 8; ++k) X[k] = w[k];
    }
#pragma unroll 1
    for (size_t b = 0; b < nblocks; ++b) {
        if (b + 1 < nblocks) {                      // the next block is in flight while this one is compressed
#pragma unroll
            for (int k = 0; k < 8; ++k) Xn[k] = w[8 * (b + 1) + k];
        }
        belt_compress_pair(T, s1, h, X, odd);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] ^= s1[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) X[k] = Xn[k];
    }
    if (fin) {
        X[0] = (uint32_t)bits_lo; X[1] = (uint32_t)(bits_lo >> 32); X[2] = (uint32_t)bits_hi; X[3] = (uint32_t)(bits_hi >> 32);
        X[4] = s[0]; X[5] = s[1]; X[6] = s[2]; X[7] = s[3];
        belt_compress_pair(T, s1, h, X, odd);
    }
    if (odd) return;                                                    // both lanes hold the result; lane 0 stores it
#pragma unroll
    for (int k = 0; k < 8; ++k) hs[k] = h[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) hs[8 + k] = s[k];
}
err_t launch_belt_hash_stream(void *d_hs, const void *d_data, size_t nblocks, int fin, uint64_t bits_lo,
                              uint64_t bits_hi, hipStream_t st)
{
    hipLaunchKernelGGL(belt_hash_stream_kernel, dim3(1), dim3(64), 0, st, (uint32_t *)d_hs, (const uint8_t *)d_data,
                       nblocks, fin, bits_lo, bits_hi);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// alg: 0 = belt-hash; 128 / 192 / 256 = bash256 / bash384 / bash512
// d_order (may be null): a permutation of 0..n-1; lane k hashes message d_order[k].  Lanes of a
// wavefront run until the longest of their 64 messages is done, so callers pass the messages
// sorted by decreasing length (the host entry point does) -- digests still land at index i.
// Long belt-hash messages (>= long_from bytes): a PAIR of lanes per message (belt_compress_pair): of the
// three encryptions of a block the last two are independent, so the chain step is 2 E instead of 3.
// Launched over all n messages (2 lanes each); pairs whose message is short leave at once.
// (round 4) templated on the table and the workgroup for the A/B of profiles/r04_long_hash_ab.txt; a workgroup none of whose
// pairs has a long message leaves before it fills the table (the caller sorts long messages to the front).
// LANES = 2: a pair per message (belt_compress_pair); LANES = 4 (round 4): a quad -- each encryption walked by two lanes that
// split the rounds' G-boxes (belt_encr_split: 4 levels of a round instead of 7 steps), the two second-stage encryptions by
// the two pairs of the quad.  LANES = 8: every G-box shared by a quad, one S-box byte per lane (belt_encr_quad), two quads for the
// second stage.
template <class Tab, int LONG_WG, int LANES = 2>
__global__ __launch_bounds__(LONG_WG)
void belt_hash_long_kernel(const uint8_t *__restrict__ data, const uint64_t *__restrict__ off,
                           const uint32_t *__restrict__ order, size_t n, uint8_t *__restrict__ digests,
                           uint64_t long_from)
{
    // the 4 KiB table of the product sits in STATIC shared memory: its LDS address is then a compile-time constant and every
    // look-up address is "byte * 4 + immediate"; through the dynamic segment the base is a run-time value and each look-up pays
    // one more dependent add -- +16 % on a chain that is bound by exactly that latency (45.3 against 39.0 ms for a 256 KiB
    // message, profiles/r04_long_hash_ab.txt).  The 64 KiB tables of the A/B forms need the dynamic segment (LDS address 0).
    constexpr bool STATIC_TAB = Tab::kBytes <= 4096;
    __shared__ __attribute__((aligned(16))) uint8_t smem_static[STATIC_TAB ? Tab::kBytes : 16];
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_dyn[];
    uint8_t *smem = STATIC_TAB ? smem_static : smem_dyn;
    static_assert(LANES == 2 || LANES == 4 || LANES == 8, "lanes per message");
    const size_t slot = ((size_t)blockIdx.x * LONG_WG + threadIdx.x) / LANES;
    const uint32_t odd = (threadIdx.x & (unsigned)(LANES / 2)) ? ~0u : 0u;        // which second-stage encryption the lane (pair / quad) walks
    const uint32_t rQ = (threadIdx.x & 1u) ? ~0u : 0u;                           // LANES == 4: the lane's column of the split rounds
    const BeltQuadLane QL(threadIdx.x & 3u);                                     // LANES == 8: the lane's S-box byte
    (void)rQ; (void)QL;
    const auto compress = [&](uint32_t (&s1)[4], uint32_t (&h)[8], const uint32_t (&X)[8], const Tab &T) {
        if constexpr (LANES == 8) belt_compress_oct(T.lds, QL, s1, h, X, odd);     // (a BeltTabSmall image)
        else if constexpr (LANES == 4) belt_compress_quad(T, s1, h, X, rQ, odd);
        else belt_compress_pair(T, s1, h, X, odd);
    };
    size_t i = 0, len = 0;
    if (slot < n) {
        i = order ? order[slot] : slot;
        len = (size_t)(off[i + 1] - off[i]);
    }
    const bool mine = slot < n && len >= long_from;
    if constexpr (!STATIC_TAB) {
        // "is anything long here?" through the first word of the (dynamic) table area itself: a static __shared__ flag -- what
        // __syncthreads_or allocates -- would push the table off LDS address 0, which its OR-composed addresses need
        volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(smem);
        if (threadIdx.x == 0) *flag = 0;
        __syncthreads();
        if (mine) *flag = 1;
        __syncthreads();
        const uint32_t any = *flag;
        __syncthreads();                                // everyone has read it before the fill overwrites it
        if (!any) return;                               // nothing long in this workgroup: no table, no work
    }
    Tab::fill(smem, threadIdx.x, LONG_WG);
    __syncthreads();
    const Tab T(smem);
    if (!mine) return;
    const uint8_t *p = data + off[i];
    size_t left = len;
    uint32_t h[8], s[4] = {0, 0, 0, 0}, X[8], s1[4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        h[k] = (uint32_t)c_beltH[4 * k] | (uint32_t)c_beltH[4 * k + 1] << 8 |
               (uint32_t)c_beltH[4 * k + 2] << 16 | (uint32_t)c_beltH[4 * k + 3] << 24;
    // The chain step is two dependent encryptions (~3.5 us on a lone pair of lanes); a global load issued when its
    // block is needed would add its ~1 us of latency to EVERY step, so the nine aligned words that hold block b + 1
    // are requested before block b is compressed (round 2: the ragged batch, whose time is its longest chain).
    {
        const uint32_t *wp = reinterpret_cast<const uint32_t *>((uintptr_t)p & ~(uintptr_t)3);
        const uint32_t sh = ((uint32_t)(uintptr_t)p & 3u) * 8u;
        uint32_t W[9], Wn[9];
        if (left >= 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) W[k] = wp[k];
            W[8] = sh ? wp[8] : 0u;
        }
        while (left >= 32) {
            if (left >= 64) {
#pragma unroll
                for (int k = 0; k < 8; ++k) Wn[k] = wp[8 + k];
                Wn[8] = sh ? wp[16] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) X[k] = __builtin_amdgcn_alignbit(W[k + 1], W[k], sh);
            compress(s1, h, X, T);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] ^= s1[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) W[k] = Wn[k];
            wp += 8; p += 32; left -= 32;
        }
    }
    if (left) {                                        // the last, zero-padded block (belt_hash.c:115-119)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                const size_t pos = (size_t)(4 * k + b);
                v = (v << 8) | (pos < left ? p[pos] : 0u);
            }
            X[k] = v;
        }
        compress(s1, h, X, T);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] ^= s1[k];
    }
    const uint64_t bits_lo = (uint64_t)len << 3, bits_hi = (uint64_t)len >> 61;
    X[0] = (uint32_t)bits_lo; X[1] = (uint32_t)(bits_lo >> 32); X[2] = (uint32_t)bits_hi; X[3] = 0;
    X[4] = s[0]; X[5] = s[1]; X[6] = s[2]; X[7] = s[3];
    compress(s1, h, X, T);
    if (!odd) {
        uint8_t *d = digests + 32 * i;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[4 * k + b] = (uint8_t)(h[k] >> (8 * b));
    }
}

// Launch order for a ragged batch when the caller gives none: messages bucketed by the power of two of
// their length, buckets in descending order (a counting sort: histogram, prefix, scatter).  Inside a
// wavefront lengths then differ by less than 2x; the order inside a bucket is whatever the atomics
// give -- it only decides which lane hashes which message, every digest lands at its own index.
// work[0..63] = histogram, work[64..127] = bucket start, work[128..191] = cursor
__device__ __forceinline__ unsigned ragged_bucket(uint64_t len) { return len ? 63u - (unsigned)__builtin_clzll(len) + 1u : 0u; }
// (global atomics of 65 536 threads on ~19 addresses serialise -- 160 us per kernel; a workgroup therefore
// counts in LDS first and touches each global bucket once)
__global__ __launch_bounds__(256)
void ragged_hist_kernel(const uint64_t *__restrict__ off, size_t n, unsigned *__restrict__ work)
{
    __shared__ unsigned h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&h[ragged_bucket(off[i + 1] - off[i])], 1u);
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&work[threadIdx.x], h[threadIdx.x]);
}
__global__ void ragged_scan_kernel(unsigned *__restrict__ work)
{
    if (threadIdx.x || blockIdx.x) return;
    unsigned acc = 0;
    for (int b = 63; b >= 0; --b) { work[64 + b] = acc; work[128 + b] = 0; acc += work[b]; }
}
__global__ __launch_bounds__(256)
void ragged_scatter_kernel(const uint64_t *__restrict__ off, size_t n, unsigned *__restrict__ work,
                           uint32_t *__restrict__ order)
{
    __shared__ unsigned cnt[64], base[64];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned b = 0, rank = 0;
    if (i < n) { b = ragged_bucket(off[i + 1] - off[i]); rank = atomicAdd(&cnt[b], 1u); }
    __syncthreads();
    if (threadIdx.x < 64 && cnt[threadIdx.x])                         // this workgroup's slice of every bucket
        base[threadIdx.x] = work[64 + threadIdx.x] + atomicAdd(&work[128 + threadIdx.x], cnt[threadIdx.x]);
    __syncthreads();
    if (i < n) order[base[b] + rank] = (uint32_t)i;
}

#ifdef BEE2HIP_EXPERIMENTS
static int g_long_hash_form = 0;
void set_long_hash_form(int v) { g_long_hash_form = v; }
#endif
static int g_ragged_fork = 1;               // long chains and short messages on two queues (0: one queue, the A/B of tools/ab/long_hash_ab.py)
#ifdef BEE2HIP_EXPERIMENTS
void set_ragged_fork(int v) { g_ragged_fork = v; }
#endif
constexpr uint64_t RAGGED_LONG = 4096;     // bytes; see bench.py --only ragged and DESIGN.md 4.7
// secret (belt-hash only): the messages hold private keys (theta of bignSign2 with long additional input) -- every message, whatever
// its length, goes through the kernel with the BANK-PRIVATE S-box copies (lane l only touches bank l & 31: the LDS cycles of a look-up
// do not depend on its index), never through the 4 KiB table whose bank conflicts follow the data
err_t launch_hash_ragged(size_t alg, const void *d_data, const void *d_off, const void *d_order, size_t n,
                         void *d_digests, hipStream_t st, bool secret)
{
    if (n == 0) return ERR_OK;
    if (secret) {
        if (alg != 0) return ERR_BAD_INPUT;
        const void *kern = reinterpret_cast<const void *>(belt_hash_ragged_kernel<BeltTabTwoP, 256>);
        B2H_TRY(dyn_lds_once(kern, BeltTabTwo::kBytes));
        hipLaunchKernelGGL((belt_hash_ragged_kernel<BeltTabTwoP, 256>), dim3((unsigned)((n + 255) / 256)), dim3(256), BeltTabTwo::kBytes, st,
                           (const uint8_t *)d_data, (const uint64_t *)d_off, (const uint32_t *)nullptr, n, (uint8_t *)d_digests, ~(size_t)0);
        B2H_TRY(hipGetLastError());
        return ERR_OK;
    }
    if (n > 0xffffffffull) return ERR_BAD_INPUT;
    const uint32_t *ord = (const uint32_t *)d_order;
    const dim3 g((unsigned)((n + 63) / 64)), t(64);
    const uint8_t *data = (const uint8_t *)d_data;
    const uint64_t *off = (const uint64_t *)d_off;
    if (!ord && n >= 128) {                          // no order given: bucket the lengths on the device
        void *scr = nullptr;
        err_t code = scratch_for_stream(st, 11, 192 * 4 + n * 4, &scr);
        if (code != ERR_OK) return code;
        unsigned *work = (unsigned *)scr;
        uint32_t *order = (uint32_t *)(work + 192);
        B2H_TRY(hipMemsetAsync(work, 0, 192 * 4, st));
        const dim3 gn((unsigned)((n + 255) / 256)), tn(256);
        hipLaunchKernelGGL(ragged_hist_kernel, gn, tn, 0, st, off, n, work);
        hipLaunchKernelGGL(ragged_scan_kernel, dim3(1), dim3(64), 0, st, work);
        hipLaunchKernelGGL(ragged_scatter_kernel, gn, tn, 0, st, off, n, work, order);
        ord = order;
    }
    uint8_t *dig = (uint8_t *)d_digests;
    // bash: messages of >= RAGGED_LONG bytes go to the 8-lanes-per-message kernel (a 4x shorter serial chain),
    // the rest stay one lane each; both launches cover all n messages and each skips what is not its own
    const dim3 gl((unsigned)((n * 8 + 63) / 64));
    // The two launches are independent (disjoint messages, disjoint digests): the long chains are a few latency-bound wavefronts
    // that leave most of the chip idle for tens of milliseconds, the short messages a throughput kernel.  Queued on ONE stream the
    // second waits for the first; forked onto the thread's side stream (joined before returning to the caller's stream order) they
    // share the chip (round 4: the bench's ragged batch 32.4 -> 31.0 ms, profiles/r04_long_hash_ab.txt).  Small batches stay on one queue.
    hipStream_t st2 = st;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    const bool forked = n >= 1024 && g_ragged_fork && side_stream(&st2, &ev_fork, &ev_join) == ERR_OK;
    if (forked) {
        B2H_TRY(hipEventRecord(ev_fork, st));
        B2H_TRY(hipStreamWaitEvent(st2, ev_fork, 0));
    } else
        st2 = st;
    if (alg == 0) {
#ifdef BEE2HIP_EXPERIMENTS      // A/B (tune 16, tools/ab/long_hash_ab.py, profiles/r04_long_hash_ab.txt): the SDWA table in one-wavefront / four-wavefront workgroups
        if (g_long_hash_form == 1) {      // the pair form (round 3's product) at every size
            hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabSmall, 64, 2>), dim3((unsigned)((n * 2 + 63) / 64)), dim3(64), 0, st, data, off, ord, n,
                               dig, RAGGED_LONG);
        } else if (g_long_hash_form == 7) {      // eight lanes per message: each G-box shared by a quad (one byte look-up per lane)
            hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabSmall, 64, 8>), dim3((unsigned)((n * 8 + 63) / 64)), dim3(64), 0, st, data, off, ord,
                               n, dig, RAGGED_LONG);
        } else if (g_long_hash_form == 5 || g_long_hash_form == 6) {      // a quad per message, rounds split over two lanes (4 KiB table / + SDWA addresses)
            const dim3 gq((unsigned)((n * 4 + 63) / 64));
            if (g_long_hash_form == 5)
                hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabSmall, 64, 4>), gq, dim3(64), 0, st, data, off, ord, n, dig, RAGGED_LONG);
            else
                hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabSmallS, 64, 4>), gq, dim3(64), 0, st, data, off, ord, n, dig, RAGGED_LONG);
        } else if (g_long_hash_form == 4) {
            hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabSmallS, 64>), dim3((unsigned)((n * 2 + 63) / 64)), dim3(64), 0, st, data, off, ord, n,
                               dig, RAGGED_LONG);
        } else if (g_long_hash_form == 2 || g_long_hash_form == 3) {
            const int wg = g_long_hash_form == 2 ? 64 : 256;
            const void *kern = wg == 64 ? reinterpret_cast<const void *>(belt_hash_long_kernel<BeltTabTwoP, 64>)
                                        : reinterpret_cast<const void *>(belt_hash_long_kernel<BeltTabTwoP, 256>);
            B2H_TRY(dyn_lds_once(kern, BeltTabTwo::kBytes));
            if (wg == 64)
                hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabTwoP, 64>), dim3((unsigned)((n * 2 + 63) / 64)), dim3(64), BeltTabTwo::kBytes, st,
                                   data, off, ord, n, dig, RAGGED_LONG);
            else
                hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabTwoP, 256>), dim3((unsigned)((n * 2 + 255) / 256)), dim3(256), BeltTabTwo::kBytes,
                                   st, data, off, ord, n, dig, RAGGED_LONG);
        } else
#endif
        // product (round 4): EIGHT lanes per long message -- every G-box shared by a quad, one S-box byte per lane, the two second-stage
        // encryptions on two quads (belt_compress_oct: 11.6 instead of 17 instructions per G-box and lane; a lone wavefront's chain
        // goes with the instructions it issues): 256 KiB in 26.6 ms against 32.5 ms for the pair form (round 3: 39).  Four times
        // the lanes, though: batches of more than 2^16 messages (where long ones could fill the SIMDs many times over) keep the pair
        // form.  4 KiB table in STATIC shared memory either way.  profiles/r04_long_hash_ab.txt has the A/B of all forms, incl.
        // the SDWA-address table (1.9x slower on a chain) and the rounds split over two lanes (no gain).
        {
        if (n <= ((size_t)1 << 16))
            hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabSmall, 64, 8>), dim3((unsigned)((n * 8 + 63) / 64)), dim3(64), 0, st, data, off, ord, n,
                               dig, RAGGED_LONG);
        else
            hipLaunchKernelGGL((belt_hash_long_kernel<BeltTabSmall, 64, 2>), dim3((unsigned)((n * 2 + 63) / 64)), dim3(64), 0, st, data, off, ord, n,
                               dig, RAGGED_LONG);
        }
    if (n >= 32768) {
            // big table; 256-thread workgroups until there are enough messages to fill 1024-thread ones on every CU
            const bool wide = n >= (size_t)num_cus() * 1024;
            const void *kern = wide ? reinterpret_cast<const void *>(belt_hash_ragged_kernel<BeltTabTwoP, 1024>)
                                    : reinterpret_cast<const void *>(belt_hash_ragged_kernel<BeltTabTwoP, 256>);
            B2H_TRY(dyn_lds_once(kern, BeltTabTwo::kBytes));
            if (wide)
                hipLaunchKernelGGL((belt_hash_ragged_kernel<BeltTabTwoP, 1024>), dim3((unsigned)((n + 1023) / 1024)),
                                   dim3(1024), BeltTabTwo::kBytes, st2, data, off, ord, n, dig, RAGGED_LONG);
            else
                hipLaunchKernelGGL((belt_hash_ragged_kernel<BeltTabTwoP, 256>), dim3((unsigned)((n + 255) / 256)),
                                   dim3(256), BeltTabTwo::kBytes, st2, data, off, ord, n, dig, RAGGED_LONG);
        } else {
            hipLaunchKernelGGL((belt_hash_ragged_kernel<BeltTabSmall, 64>), g, t, 0, st2, data, off, ord, n, dig, RAGGED_LONG);
        }
    }
    else if (alg == 256) {
        hipLaunchKernelGGL(bash_long_kernel<8>, gl, t, 0, st, data, off, ord, n, 256u, dig, RAGGED_LONG);
        hipLaunchKernelGGL(bash_ragged_kernel<8>, g, t, 0, st2, data, off, ord, n, 256u, dig, RAGGED_LONG);
    } else if (alg == 192) {
        hipLaunchKernelGGL(bash_long_kernel<12>, gl, t, 0, st, data, off, ord, n, 192u, dig, RAGGED_LONG);
        hipLaunchKernelGGL(bash_ragged_kernel<12>, g, t, 0, st2, data, off, ord, n, 192u, dig, RAGGED_LONG);
    } else if (alg == 128) {
        hipLaunchKernelGGL(bash_long_kernel<16>, gl, t, 0, st, data, off, ord, n, 128u, dig, RAGGED_LONG);
        hipLaunchKernelGGL(bash_ragged_kernel<16>, g, t, 0, st2, data, off, ord, n, 128u, dig, RAGGED_LONG);
    }
    else return ERR_NOT_IMPLEMENTED;
    B2H_TRY(hipGetLastError());
    if (forked) {
        B2H_TRY(hipEventRecord(ev_join, st2));
        B2H_TRY(hipStreamWaitEvent(st, ev_join, 0));
    }
    return ERR_OK;
}

// ------------------------------------------------------------------ launchers ---
err_t launch_bash_sponge(void *d_states, const void *d_data, size_t stride, size_t count, size_t n,
                         int fin, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    hipLaunchKernelGGL(bash_sponge_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st,
                       (bash_hash_st *)d_states, (const uint8_t *)d_data, stride, count, n, fin);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}
err_t launch_belt_mac(void *d_states, const void *d_data, size_t stride, size_t count, size_t n,
                      int mode, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    hipLaunchKernelGGL(belt_mac_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st,
                       (belt_mac_st *)d_states, (const uint8_t *)d_data, stride, count, n, mode);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// table of the MAC half: 0 = the product, 1 = BeltTabWide (four tables, 128 KiB: rounds 1-2), 2 = BeltTabTwoP (A/B: tune 13)
static int g_fused_tab = 0;
void set_fused_tab(int v) { g_fused_tab = v; }
template <int RW, bool HASH, bool MAC, class Tab>
static err_t launch_fused_tt(const void *d_msgs, size_t msg_len, size_t n, size_t l, const MacKey &key,
                             void *d_digests, void *d_tags, hipStream_t st);
template <int RW, bool HASH, bool MAC>
static err_t launch_fused_t(const void *d_msgs, size_t msg_len, size_t n, size_t l, const MacKey &key,
                            void *d_digests, void *d_tags, hipStream_t st)
{
#ifdef BEE2HIP_EXPERIMENTS      // A/B only (tools/ab/fused_tab_ab.py): +0.5 %, bash-f's VALU work bounds the kernel
    if (MAC && g_fused_tab == 2) return launch_fused_tt<RW, HASH, MAC, BeltTabTwoP>(d_msgs, msg_len, n, l, key, d_digests, d_tags, st);
#endif
    return launch_fused_tt<RW, HASH, MAC, BeltTabWide>(d_msgs, msg_len, n, l, key, d_digests, d_tags, st);
}
template <int RW, bool HASH, bool MAC, class Tab>
static err_t launch_fused_tt(const void *d_msgs, size_t msg_len, size_t n, size_t l, const MacKey &key,
                             void *d_digests, void *d_tags, hipStream_t st)
{
    auto kern = hash_mac_fused_kernel<RW, HASH, MAC, Tab>;
    const size_t lds = MAC ? (size_t)Tab::kBytes : 0;
    if (MAC) B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(kern), lds));
    const size_t grid = (n + FUSED_WG - 1) / FUSED_WG;
    if (grid > 0x7fffffffull) return ERR_BAD_INPUT;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FUSED_WG), lds, st, (const uint4 *)d_msgs, msg_len, n,
                       (uint32_t)l, key, (uint8_t *)d_digests, (uint8_t *)d_tags);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

template <int RW>
static err_t launch_fused_rw(const void *d_msgs, size_t msg_len, size_t n, size_t l, const MacKey &key,
                             bool do_hash, bool do_mac, void *d_digests, void *d_tags, hipStream_t st)
{
    if (do_hash && do_mac) return launch_fused_t<RW, true, true>(d_msgs, msg_len, n, l, key, d_digests, d_tags, st);
    if (do_hash) return launch_fused_t<RW, true, false>(d_msgs, msg_len, n, l, key, d_digests, d_tags, st);
    // MAC alone does not depend on the rate: launch_bashHash_beltMAC sends it to RW = 8 only (one instantiation)
    if constexpr (RW == 8) return launch_fused_t<RW, false, true>(d_msgs, msg_len, n, l, key, d_digests, d_tags, st);
    else return ERR_BAD_INPUT;
}

err_t launch_bashHash_beltMAC(const void *d_msgs, size_t msg_len, size_t n, size_t l,
                              const uint32_t key[8], bool do_hash, bool do_mac,
                              void *d_digests, void *d_tags, hipStream_t st)
{
    if (n == 0 || (!do_hash && !do_mac)) return ERR_OK;
    MacKey k;
    for (int i = 0; i < 8; ++i) k.k[i] = do_mac ? key[i] : 0u;
    const bool aligned = msg_len % 16 == 0 && ((uintptr_t)d_msgs % 16) == 0 &&
                         (!do_hash || ((uintptr_t)d_digests % 8) == 0) && (!do_mac || ((uintptr_t)d_tags % 8) == 0);
    if (aligned && (!do_hash || l == 128 || l == 192 || l == 256)) {
        if (!do_hash || l == 256) return launch_fused_rw<8>(d_msgs, msg_len, n, 256, k, do_hash, do_mac, d_digests, d_tags, st);
        if (l == 192) return launch_fused_rw<12>(d_msgs, msg_len, n, l, k, do_hash, do_mac, d_digests, d_tags, st);
        return launch_fused_rw<16>(d_msgs, msg_len, n, l, k, do_hash, do_mac, d_digests, d_tags, st);
    }
    // generic shapes: per-item states in scratch, byte-granular kernels
    const size_t need = n * (sizeof(bash_hash_st) + sizeof(belt_mac_st));
    void *base = nullptr;
    {
        const err_t sc = scratch_for_stream(st, 1, need, &base);
        if (sc != ERR_OK) return sc;
    }
    bash_hash_st *hs = (bash_hash_st *)base;
    belt_mac_st *ms = (belt_mac_st *)(hs + n);
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(init_states_kernel, dim3(g), dim3(256), 0, st, do_hash ? hs : nullptr,
                       do_mac ? ms : nullptr, n, (uint32_t)l, k);
    err_t code = ERR_OK;
    if (do_hash) code = launch_bash_sponge(hs, d_msgs, msg_len, msg_len, n, 1, st);
    if (code == ERR_OK && do_mac) code = launch_belt_mac(ms, d_msgs, msg_len, msg_len, n, 1 | 2 | 4, st);
    if (code != ERR_OK) return code;
    hipLaunchKernelGGL(gather_results_kernel, dim3(g), dim3(256), 0, st, do_hash ? hs : nullptr,
                       do_mac ? ms : nullptr, n, l / 4, (uint8_t *)d_digests, (uint8_t *)d_tags);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

}  // namespace bee2hip
