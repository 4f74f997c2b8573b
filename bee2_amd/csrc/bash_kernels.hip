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

err_t launch_bashF_batch(void *d_states, size_t n, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    const size_t per_wg = BASHF_WG;           // one state per lane
    const size_t grid = (n + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffull) return ERR_BAD_INPUT;
    // PASSES = 1: 64 records staged at once (13 KiB LDS per wavefront, 3 wavefronts/SIMD).
    // PASSES = 2 (6.5 KiB, 5-6 wavefronts/SIMD) measured the same within noise on MI355X:
    // the kernel is VALU-issue bound, not latency bound.
    hipLaunchKernelGGL(bashF_batch_kernel<1>, dim3((unsigned)grid), dim3(BASHF_WG),
                       (BASHF_WG / 64) * 64 * BASHF_PAD, st, (uint8_t *)d_states, n);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

}  // namespace bee2hip
