// bash_kernels.hip -- batched bash-f and the lane-per-message bash sponge on gfx950.
//
// bashF_batch_kernel  : H1 of SURVEY.md 8a -- n independent 192-byte states,
//                       one lane per state (replaces n calls of bashF,
//                       include/bee2/crypto/bash.h:136).
//
// HBM layout: states are contiguous 192-byte records, exactly bee2's layout.
// A wavefront owns 64 consecutive states = 12 KiB.  It reads them with 12 coalesced
// global_load_dwordx4 (consecutive lanes take consecutive 16-byte pieces), transposes
// through LDS in two 96-byte halves (record stride padded to 112 bytes so the per-lane
// ds_read_b128 of a record is bank-conflict free), permutes in registers and writes back
// the same way.  Algorithmic traffic: 384 B/state.
#include "bash_dev.hpp"
#include "common.hpp"

namespace bee2hip {

static_assert(BashSlots{}.m[6][0] == 0 && BashSlots{}.m[6][13] == 13 && BashSlots{}.m[6][23] == 23,
              "bash word permutation must have order 6");

constexpr int BASHF_WG = 256;                 // 4 wavefronts
constexpr int BASHF_REC = 192;                // bytes per state
// The transposition goes through LDS 64 / PASSES records at a time; every global access is
// a fully contiguous 1 KiB per wave-instruction.  Stride 208 B = 52 dwords keeps the per-lane
// ds_read_b128 / ds_write_b128 of a whole record bank-conflict free.
constexpr int BASHF_PAD = 208;

template <int PASSES>
__global__ __launch_bounds__(BASHF_WG)
void bashF_batch_kernel(uint8_t *__restrict__ states, size_t n)
{
    constexpr int RECS = 64 / PASSES;                 // records staged per pass
    constexpr int SLAB = RECS * BASHF_REC;            // bytes per pass (12 KiB or 6 KiB)
    constexpr int BASHF_WAVE_LDS = RECS * BASHF_PAD;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint8_t *wl = smem + wave * BASHF_WAVE_LDS;
    const size_t first = ((size_t)blockIdx.x * (BASHF_WG / 64) + wave) * 64;   // first state of this wave
    if (first >= n) return;                                                     // whole wave idle (wave-uniform)
    const size_t left = n - first;
    const int cnt = left < 64 ? (int)left : 64;                                 // states owned by this wave
    const int bytes = cnt * BASHF_REC;
    uint8_t *g = states + first * BASHF_REC;
    const int half = lane / RECS;                                               // which pass holds my record
    const int lrec = lane % RECS;

    u64x2 a[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) a[j].lo = a[j].hi = 0;
    // ---- load: pass h moves records [32h, 32h+32) = 6 KiB: 6 coalesced 1 KiB loads -> LDS
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
#pragma unroll
        for (int k = 0; k < SLAB / 1024; ++k) {
            const int o = k * 1024 + lane * 16;                                 // offset inside the slab
            if (h * SLAB + o < bytes) {
                const uint4 v = *reinterpret_cast<const uint4 *>(g + h * SLAB + o);
                const int rec = o / BASHF_REC, off = o % BASHF_REC;
                *reinterpret_cast<uint4 *>(wl + rec * BASHF_PAD + off) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (half == h && lane < cnt) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const uint4 v = *reinterpret_cast<const uint4 *>(wl + lrec * BASHF_PAD + 16 * j);
                a[2 * j].lo = v.x; a[2 * j].hi = v.y; a[2 * j + 1].lo = v.z; a[2 * j + 1].hi = v.w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    bash_f<true>(a);          // staged issue order: +3 % here (bash_dev.hpp)

    // ---- store: mirror image
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
        if (half == h && lane < cnt) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                uint4 v;
                v.x = a[2 * j].lo; v.y = a[2 * j].hi; v.z = a[2 * j + 1].lo; v.w = a[2 * j + 1].hi;
                *reinterpret_cast<uint4 *>(wl + lrec * BASHF_PAD + 16 * j) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int k = 0; k < SLAB / 1024; ++k) {
            const int o = k * 1024 + lane * 16;
            if (h * SLAB + o < bytes) {
                const int rec = o / BASHF_REC, off = o % BASHF_REC;
                const uint4 v = *reinterpret_cast<const uint4 *>(wl + rec * BASHF_PAD + off);
                *reinterpret_cast<uint4 *>(g + h * SLAB + o) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}


// ---------------------------------------------------------------------------------------------
// r02 variants of the batch kernel (profiles/r02_bashF_variants.txt has the A/B numbers).
//
// LDS-DMA load: global_load_lds_dwordx4 writes lane i's 16 bytes to <wave-uniform base> + 16 i, so the
// LDS image of one instruction is 64 consecutive 16-byte slots and the record stride cannot be padded
// by the *destination*.  The padded layout (13 slots per record, slot 12 unused) is produced by the
// *source* instead: 13 instructions cover the 832 slots, lane i of instruction k fills slot
// s = 64 k + i with chunk min(s mod 13, 11) of record s div 13 (the pad slot re-reads chunk 11 of the
// same record: same memory line, no extra traffic).  No VGPR round trip, no ds_write_b128.
__device__ __forceinline__ void bashF_dma_tile(const uint8_t *g, uint8_t *wl, int lane, int cnt)
{
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const unsigned s = 64u * k + lane;
        unsigned r = (s * 5042u) >> 16;                    // s div 13 for s < 832
        unsigned j = s - 13u * r;
        j = j > 11u ? 11u : j;
        r = r < (unsigned)cnt ? r : (unsigned)cnt - 1u;    // ragged tile: stay inside the batch
        const uint8_t *src = g + r * BASHF_REC + j * 16u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(wl + 1024 * k), 16, 0, 0);
    }
}
__device__ __forceinline__ void bashF_read_slab(u64x2 (&a)[24], const uint8_t *wl, int lane)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const uint4 v = *reinterpret_cast<const uint4 *>(wl + lane * BASHF_PAD + 16 * j);
        a[2 * j].lo = v.x; a[2 * j].hi = v.y; a[2 * j + 1].lo = v.z; a[2 * j + 1].hi = v.w;
    }
}
// Ordering of one wavefront's own LDS traffic.  A workgroup-scope release fence also waits for the
// wavefront's outstanding *global* stores (vmcnt(0)) -- in a walking wavefront that exposed the store
// latency of every tile (r02: 171 us instead of 118), and at the end of a one-tile kernel it keeps the
// slab allocated until the stores are acknowledged.  DS operations of one wavefront execute in order,
// so all that is needed is that the compiler keeps the order and that returned data has arrived.
__device__ __forceinline__ void bashF_wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// registers -> slab (transposed) -> coalesced 1 KiB global stores
__device__ __forceinline__ void bashF_store_via_slab(const u64x2 (&a)[24], uint8_t *g, uint8_t *wl, int lane, int cnt)
{
    if (lane < cnt) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            uint4 v;
            v.x = a[2 * j].lo; v.y = a[2 * j].hi; v.z = a[2 * j + 1].lo; v.w = a[2 * j + 1].hi;
            *reinterpret_cast<uint4 *>(wl + lane * BASHF_PAD + 16 * j) = v;
        }
    }
    bashF_wave_sync();
    const int bytes = cnt * BASHF_REC;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int o = k * 1024 + lane * 16;
        if (o < bytes) {
            const int rec = o / BASHF_REC, off = o % BASHF_REC;
            const uint4 v = *reinterpret_cast<const uint4 *>(wl + rec * BASHF_PAD + off);
            *reinterpret_cast<uint4 *>(g + o) = v;
        }
    }
}
// registers -> global, every lane its own record (16-byte pieces at stride 192)
__device__ __forceinline__ void bashF_store_direct(const u64x2 (&a)[24], uint8_t *g, int lane, int cnt)
{
    if (lane < cnt) {
        uint4 *p = reinterpret_cast<uint4 *>(g + lane * BASHF_REC);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            uint4 v;
            v.x = a[2 * j].lo; v.y = a[2 * j].hi; v.z = a[2 * j + 1].lo; v.w = a[2 * j + 1].hi;
            p[j] = v;
        }
    }
}

// Generic one-tile-per-wavefront kernel.  LOAD / STORE: 0 = direct (every lane its own record, 16-byte
// pieces at stride 192), 1 = through a 64-record slab (13 KiB per wavefront; load by LDS-DMA), 2 = through a
// 32-record half slab in two passes (7 KiB per wavefront).  ORDER: issue order of the S-layer (bash_dev.hpp).
template <int LOAD, int STORE>
constexpr int bashF_tile_lds()
{
    // STORE = P in {2, 4, 8}: P passes of 64 / P records through a slab of that many padded records
    return (LOAD == 1 || STORE == 1) ? 64 * BASHF_PAD : LOAD == 2 ? 7168 : (STORE == 2 || STORE == 4 || STORE == 8) ? (64 / STORE) * BASHF_PAD : 0;
}
typedef uint32_t bash_v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 bashF_ld16(const uint4 *p)
{
    if constexpr (NT) {
        const bash_v4u v = __builtin_nontemporal_load(reinterpret_cast<const bash_v4u *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else return *p;
}
template <bool NT>
__device__ __forceinline__ void bashF_st16(uint4 *p, const uint4 o)
{
    if constexpr (NT) {
        bash_v4u v = {o.x, o.y, o.z, o.w};
        __builtin_nontemporal_store(v, reinterpret_cast<bash_v4u *>(p));
    } else *p = o;
}

// FLAGS: 1 = priority 3 until the loads are out, 2 = priority 3 for the store phase, 4 = non-temporal loads,
// 8 = non-temporal stores (round-3 A/B, profiles/r03_bashF_nt_ab.txt)
template <int LOAD, int STORE, int ORDER, int MINW, int FLAGS = 0>
__global__ __launch_bounds__(BASHF_WG, MINW)
void bashF_tile_kernel(uint8_t *__restrict__ states, size_t n)
{
    // FLAGS & 1: the new wavefront is the youngest on its SIMD; without help its address arithmetic and load issue
    // wait behind every older wavefront's VALU work.  Priority 3 until the loads are out.
    if constexpr (FLAGS & 1) __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *wl = smem + wave * bashF_tile_lds<LOAD, STORE>();
    const size_t first = ((size_t)blockIdx.x * (BASHF_WG / 64) + wave) * 64;
    if (first >= n) return;
    const size_t left = n - first;
    const int cnt = left < 64 ? (int)left : 64;
    uint8_t *g = states + first * BASHF_REC;
    u64x2 a[24];
    if constexpr (LOAD == 9) {                     // ablation: no memory phase
#pragma unroll
        for (int j = 0; j < 24; ++j) { a[j].lo = lane * (2 * j + 1) + (uint32_t)first; a[j].hi = lane ^ (j * 0x9E3779B9u); }
    } else if constexpr (LOAD == 0) {
        const int r = lane < cnt ? lane : cnt - 1;
        const uint4 *p = reinterpret_cast<const uint4 *>(g + r * BASHF_REC);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const uint4 v = bashF_ld16<(FLAGS & 4) != 0>(p + j);
            a[2 * j].lo = v.x; a[2 * j].hi = v.y; a[2 * j + 1].lo = v.z; a[2 * j + 1].hi = v.w;
        }
    } else if constexpr (LOAD == 1) {
        bashF_dma_tile(g, wl, lane, cnt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bashF_read_slab(a, wl, lane);
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // records [32h, 32h + 32): 416 slots of the padded layout, 7 instructions (the last 32 lanes of
            // the seventh land in the spare KiB of the 7 KiB half slab)
            if (h) bashF_wave_sync();
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const unsigned s = 64u * k + lane;
                unsigned r = (s * 5042u) >> 16;
                unsigned j = s - 13u * r;
                j = j > 11u ? 11u : j;
                r += 32u * h;
                r = r < (unsigned)cnt ? r : (unsigned)cnt - 1u;
                const uint8_t *src = g + r * BASHF_REC + j * 16u;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(wl + 1024 * k), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((lane >> 5) == h) bashF_read_slab(a, wl, lane & 31);
        }
    }

    if constexpr (FLAGS & 1) __builtin_amdgcn_s_setprio(0);
    if constexpr (ORDER >= 0) bash_f<ORDER>(a);
    if constexpr (FLAGS & 2) __builtin_amdgcn_s_setprio(3);          // drain the store phase ahead of others' arithmetic

    if constexpr (STORE == 9) {                    // ablation: keep the result alive with one dword per lane
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 24; ++j) x ^= a[j].lo ^ a[j].hi;
        if (x == 0x12345678u) *reinterpret_cast<uint32_t *>(g + lane * 4) = x;
    } else if constexpr (STORE == 0) bashF_store_direct(a, g, lane, cnt);
    else if constexpr (STORE == 1) { bashF_wave_sync(); bashF_store_via_slab(a, g, wl, lane, cnt); }
    else {
        constexpr int P = STORE, RECS = 64 / P, SLAB = RECS * BASHF_REC;      // SLAB bytes of states per pass
#pragma unroll
        for (int h = 0; h < P; ++h) {
            bashF_wave_sync();
            if (lane / RECS == h && lane < cnt) {
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    uint4 v;
                    v.x = a[2 * j].lo; v.y = a[2 * j].hi; v.z = a[2 * j + 1].lo; v.w = a[2 * j + 1].hi;
                    *reinterpret_cast<uint4 *>(wl + (lane % RECS) * BASHF_PAD + 16 * j) = v;
                }
            }
            bashF_wave_sync();
            const int bytes = cnt * BASHF_REC - h * SLAB;
#pragma unroll
            for (int k = 0; k < SLAB / 1024; ++k) {
                const int o = k * 1024 + lane * 16;
                if (o < bytes) {
                    const int rec = o / BASHF_REC, off = o % BASHF_REC;
                    const uint4 v = *reinterpret_cast<const uint4 *>(wl + rec * BASHF_PAD + off);
                    bashF_st16<(FLAGS & 8) != 0>(reinterpret_cast<uint4 *>(g + h * SLAB + o), v);
                }
            }
        }
    }
}

// V4 / V5: persistent wavefronts.  A wavefront walks tiles t, t + nwaves, ...; the LDS-DMA of the next
// tile is in flight during the 24 rounds of the current one (it costs LDS, not VGPRs), so the memory
// phase of a wavefront overlaps its own arithmetic.  V4 stores through the slab: after the rounds the
// next tile moves slab -> second register set, the results go registers -> slab -> memory, and only
// then is the slab handed to the DMA of the tile after next.  V5 stores directly (slab is free as soon
// as it has been read).
template <bool DIRECT_STORE, int STAGED>
__global__ __launch_bounds__(BASHF_WG, 3)
void bashF_walk_kernel(uint8_t *__restrict__ states, size_t n, unsigned nwaves)
{
    // 256-lane workgroups only so that the four wavefronts land on the four SIMDs (one-wavefront
    // workgroups were placed unevenly: 172 us); the wavefronts never synchronise with each other.
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *wl = smem + wave * (64 * BASHF_PAD);
    const int lane = threadIdx.x & 63;
    const size_t ntiles = (n + 63) / 64;
    size_t tile = (size_t)blockIdx.x * (BASHF_WG / 64) + wave;
    if (tile >= ntiles) return;
    auto count = [&](size_t t) { const size_t left = n - t * 64; return left < 64 ? (int)left : 64; };
    u64x2 a[24];
    bashF_dma_tile(states + tile * 64 * BASHF_REC, wl, lane, count(tile));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bashF_read_slab(a, wl, lane);
    for (;;) {
        const size_t next = tile + nwaves;
        const bool more = next < ntiles;
        uint8_t *g = states + tile * 64 * BASHF_REC;
        const int cnt = count(tile);
        bashF_wave_sync();                                  // slab reads done before the DMA may overwrite it
        int ln = lane;
        asm volatile("" : "+v"(ln));                        // keep the 13 source offsets out of the loop-invariant set
        if (more) bashF_dma_tile(states + next * 64 * BASHF_REC, wl, ln, count(next));
        bash_f<STAGED>(a);
        if constexpr (DIRECT_STORE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the DMA landed long ago; waiting *before* the
            bashF_store_direct(a, g, ln, cnt);                   // stores keeps their latency off this wavefront
            if (!more) break;
            bashF_read_slab(a, wl, ln);
        } else {
            u64x2 b[24];
            int l2 = lane;
            asm volatile("" : "+v"(l2));
            if (more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bashF_read_slab(b, wl, l2);
            }
            bashF_wave_sync();
            bashF_store_via_slab(a, g, wl, l2, cnt);
            if (!more) break;
#pragma unroll
            for (int j = 0; j < 24; ++j) a[j] = b[j];
        }
        tile = next;
    }
}

template <int LOAD, int STORE, int ORDER, int MINW, int FLAGS = 0>
static void launch_bashF_tile(unsigned grid, uint8_t *p, size_t n, hipStream_t st)
{
    constexpr int lds = 4 * bashF_tile_lds<LOAD, STORE>();
    auto k = bashF_tile_kernel<LOAD, STORE, ORDER, MINW, FLAGS>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(BASHF_WG), lds, st, p, n);
}

static int g_bashF_variant = -1;
void set_bashF_variant(int v) { g_bashF_variant = v; }

err_t launch_bashF_batch(void *d_states, size_t n, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    const size_t per_wg = BASHF_WG;           // one state per lane
    const size_t grid = (n + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffull) return ERR_BAD_INPUT;
    uint8_t *p = (uint8_t *)d_states;
    const size_t lds4 = (BASHF_WG / 64) * 64 * BASHF_PAD;
    const int v = g_bashF_variant;
    const size_t ntiles = (n + 63) / 64;
    const unsigned nwg = (unsigned)((ntiles + 3) / 4 < 768 ? (ntiles + 3) / 4 : 768);   // 256 CUs x 3 workgroups (12 slabs of 13 KiB)
    const unsigned nwaves = nwg * 4;
#define TILE(L, S, O, W) launch_bashF_tile<L, S, O, W>((unsigned)grid, p, n, st)
#define TILEF(L, S, O, W, F) launch_bashF_tile<L, S, O, W, F>((unsigned)grid, p, n, st)
    // Variants kept for the A/B record (profiles/r02_bashF_variants.txt lists every one that was measured; the
    // numbers there name them as below).  Product = default.
#ifdef BEE2HIP_EXPERIMENTS      // the A/B record only (tools/ab/ab_lib.sh builds it); the product library holds ONE instantiation
    switch (v) {
    case 0: hipLaunchKernelGGL(bashF_batch_kernel<1>, dim3((unsigned)grid), dim3(BASHF_WG), lds4, st, p, n); break;   // r01 product
    case 1: TILE(1, 1, 1, 3); break;         // LDS-DMA load, slab store, r01 staged order
    case 3: TILE(0, 0, 1, 4); break;         // no LDS, r01 staged order
    case 4: hipLaunchKernelGGL((bashF_walk_kernel<false, 1>), dim3(nwg), dim3(BASHF_WG), lds4, st, p, n, nwaves); break;
    case 11: TILE(0, 0, 24, 6); break;       // no LDS, staged2 W = 4, no priority
    case 31: TILE(0, 0, 128, 4); break;      // class-following priority: no LDS, staged2 W = 8
    case 36: TILE(0, 2, 128, 4); break;      // direct load, half-slab store, W = 8
    case 53: TILEF(0, 4, 122, 8, 1); break;  // W = 2, quarter-slab store, 8 wavefronts/SIMD, load priority
    case 40: TILE(9, 9, 128, 4); break;      // ablation: rounds only, priority
    case 41: TILE(9, 9, 28, 4); break;       //           rounds only, no priority
    case 42: TILE(0, 2, -1, 4); break;       //           memory only: direct load, half-slab store
    case 70: TILEF(0, 2, 124, 6, 3 | 4); break;      // product + non-temporal loads
    case 71: TILEF(0, 2, 124, 6, 3 | 8); break;      // product + non-temporal stores
    case 72: TILEF(0, 2, 124, 6, 3 | 12); break;     // product + both
    default: TILEF(0, 2, 124, 6, 3);         // product (v61): W = 4, priority, direct load, half-slab store
    }
    (void)lds4; (void)nwg; (void)nwaves;
#else
    (void)v; (void)lds4; (void)nwg; (void)nwaves;
    TILEF(0, 2, 124, 6, 3);                  // W = 4, priority, direct load, half-slab store
#endif
#undef TILE
#undef TILEF
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

}  // namespace bee2hip
