// bash_kernels.hip -- batched bash-f and the lane-per-message bash sponge on gfx950.
//
// bashF_batch_kernel  : H1 of SURVEY.md 8a -- n independent 192-byte states,
//                       one lane per state (replaces n calls of bashF,
//                       include/bee2/crypto/bash.h:136).
//
// HBM layout: states are contiguous 192-byte records, exactly bee2's layout.
// A wavefront owns 64 consecutive states = 12 KiB.  It reads them with 12 fully
// coalesced global_load_dwordx4 (lane i takes bytes [16 i, 16 i + 16) of each
// 1 KiB slab), transposes through LDS (record stride padded 192 -> 208 bytes so
// the per-lane ds_read_b128 of a whole record is bank-conflict free), permutes
// in registers and writes back the same way.  Algorithmic traffic: 384 B/state.
#include "bash_dev.hpp"
#include "common.hpp"

namespace bee2hip {

static_assert(BashSlots{}.m[6][0] == 0 && BashSlots{}.m[6][13] == 13 && BashSlots{}.m[6][23] == 23,
              "bash word permutation must have order 6");

constexpr int BASHF_WG = 256;                 // 4 wavefronts
constexpr int BASHF_REC = 192;                // bytes per state
constexpr int BASHF_PAD = 208;                // LDS record stride (16-B aligned, conflict-free b128)
constexpr int BASHF_WAVE_LDS = 64 * BASHF_PAD;

__global__ __launch_bounds__(BASHF_WG)
void bashF_batch_kernel(uint8_t *__restrict__ states, size_t n)
{
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

    // ---- coalesced load -> LDS (record-padded) ----
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int o = k * 1024 + lane * 16;
        if (o < bytes) {
            const uint4 v = *reinterpret_cast<const uint4 *>(g + o);
            const int rec = o / BASHF_REC, off = o % BASHF_REC;
            *reinterpret_cast<uint4 *>(wl + rec * BASHF_PAD + off) = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    u64x2 a[24];
    if (lane < cnt) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const uint4 v = *reinterpret_cast<const uint4 *>(wl + lane * BASHF_PAD + 16 * j);
            a[2 * j].lo = v.x; a[2 * j].hi = v.y; a[2 * j + 1].lo = v.z; a[2 * j + 1].hi = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 24; ++j) a[j].lo = a[j].hi = 0;
    }

    bash_f(a);

    if (lane < cnt) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            uint4 v;
            v.x = a[2 * j].lo; v.y = a[2 * j].hi; v.z = a[2 * j + 1].lo; v.w = a[2 * j + 1].hi;
            *reinterpret_cast<uint4 *>(wl + lane * BASHF_PAD + 16 * j) = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- LDS -> coalesced store ----
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

err_t launch_bashF_batch(void *d_states, size_t n, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    const size_t per_wg = BASHF_WG;           // one state per lane
    const size_t grid = (n + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffull) return ERR_BAD_INPUT;
    hipLaunchKernelGGL(bashF_batch_kernel, dim3((unsigned)grid), dim3(BASHF_WG),
                       (BASHF_WG / 64) * BASHF_WAVE_LDS, st, (uint8_t *)d_states, n);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

}  // namespace bee2hip
