// Micro-benchmark: one belt E_K chain (x <- E_K(x), the critical path of a long belt-hash) on ONE lane with table
// lookups for all four bytes of a G-box (belt_dev.hpp, BeltTabSmall) against FOUR lanes with one byte each and a
// two-step DPP xor reduction (belt_dev.hpp, BeltQuad), on a wavefront that is alone on its SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -I bee2_amd/csrc -I include tools/ubench/belt_quad.hip -o tools/ubench/belt_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "belt_dev.hpp"
namespace bee2hip { __constant__ uint8_t c_beltH[256]; }
using namespace bee2hip;

// A G-box on FOUR lanes (a DPP quad that holds x replicated): lane q looks up byte q of x in the table that byte
// needs, and two v_xor_b32_dpp steps give every lane the xor of the four values -- 6 instructions per G-box on the
// critical path instead of ~19.  The hope: for ONE dependent chain on a wavefront that is alone on its SIMD (the
// long message of a ragged belt-hash batch, the drop-in beltHash) a kernel costs its instruction count.  Measured:
// x1.11 only (1.78 vs 1.99 us per E_K) -- this chain is ~34 dependent LDS round trips, not an instruction stream;
// with the asm block volatile (no overlap of independent G-boxes) it is even slower, x0.83.  Not in the product.  The two DPP steps form one asm block: the second
// reads what the first wrote (2 wait states), and nothing inside inline asm inserts them.
struct BeltQuad {
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    uint32_t base[3];       // LDS byte address of the table of byte q for G5, G13, G21: table (R0 + q) & 3
    uint32_t sh;            // 8 q
    __device__ BeltQuad(const uint8_t *l, uint32_t q)       // l: a filled BeltTabSmall
    {
        const uint32_t tab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)l;
#pragma unroll
        for (int r0 = 0; r0 < 3; ++r0) base[r0] = tab + ((r0 + q) & 3u) * 1024u;
        sh = 8u * q;
    }
    template <int R0>
    __device__ __forceinline__ uint32_t g(uint32_t x) const
    {
        const uint32_t byte = __builtin_amdgcn_ubfe(x, sh, 8u);
        const uint32_t t = *(lds_u32 *)(uintptr_t)(base[R0] + (byte << 2));
        uint32_t o;
        asm("v_xor_b32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_xor_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
            : "=&v"(o) : "v"(t));
        return o;
    }
};
template <int I>
__device__ __forceinline__ void belt_round_quad(const BeltQuad &T, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d,
                                                const uint32_t (&K)[8])
{
    constexpr int o = 7 * I - 7;
    b ^= T.template g<0>(a + K[(o + 0) & 7]);
    c ^= T.template g<2>(d + K[(o + 1) & 7]);
    a -= T.template g<1>(b + K[(o + 2) & 7]);
    const uint32_t e = T.template g<2>(b + c + K[(o + 3) & 7]) ^ (uint32_t)I;
    b += e;
    c -= e;
    d += T.template g<1>(c + K[(o + 4) & 7]);
    b ^= T.template g<2>(a + K[(o + 5) & 7]);
    c ^= T.template g<0>(d + K[(o + 6) & 7]);
}
// E_K by a quad; x and K replicated in its four lanes, the result too
__device__ __forceinline__ void belt_encr_quad(const BeltQuad &T, uint32_t (&x)[4], const uint32_t (&K)[8])
{
    uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
    belt_round_quad<1>(T, a, b, c, d, K);
    belt_round_quad<2>(T, b, d, a, c, K);
    belt_round_quad<3>(T, d, c, b, a, K);
    belt_round_quad<4>(T, c, a, d, b, K);
    belt_round_quad<5>(T, a, b, c, d, K);
    belt_round_quad<6>(T, b, d, a, c, K);
    belt_round_quad<7>(T, d, c, b, a, K);
    belt_round_quad<8>(T, c, a, d, b, K);
    x[0] = b; x[1] = d; x[2] = a; x[3] = c;
}


template <int MODE>
__global__ __launch_bounds__(64) void chain(const uint32_t *in, uint32_t *out, int reps)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[BeltTabSmall::kBytes];
    BeltTabSmall::fill(smem, threadIdx.x, 64);
    __syncthreads();
    uint32_t x[4], K[8];
    const int src = MODE == 0 ? threadIdx.x : (threadIdx.x >> 2);        // quads share their input
    for (int i = 0; i < 4; ++i) x[i] = in[12 * src + i];
    for (int i = 0; i < 8; ++i) K[i] = in[12 * src + 4 + i];
    if (MODE == 0) {
        const BeltTabSmall T(smem);
#pragma unroll 1
        for (int r = 0; r < reps; ++r) belt_encr(T, x, K);
    } else {
        const BeltQuad T(smem, threadIdx.x & 3u);
#pragma unroll 1
        for (int r = 0; r < reps; ++r) belt_encr_quad(T, x, K);
    }
    for (int i = 0; i < 4; ++i) out[4 * threadIdx.x + i] = x[i];
}

int main()
{
    uint8_t H[256];
    {   // the belt S-box by the standard's recipe is not needed for timing: any permutation of bytes will do, but keep it
        // deterministic and non-trivial
        for (int i = 0; i < 256; ++i) H[i] = (uint8_t)(i * 167 + 13);
    }
    hipMemcpyToSymbol(HIP_SYMBOL(bee2hip::c_beltH), H, 256);
    uint32_t hin[64 * 12], ha[64 * 4], hb[64 * 4];
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 64 * 12; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hin[i] = (uint32_t)(s >> 16); }
    uint32_t *din, *dout;
    hipMalloc(&din, sizeof hin); hipMalloc(&dout, sizeof ha);
    hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    const int reps = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, din, dout, reps);
            else hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, din, dout, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
        hipMemcpy(mode ? hb : ha, dout, sizeof ha, hipMemcpyDeviceToHost);
    }
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i) bad += hb[4 * lane + i] != ha[4 * (lane >> 2) + i];
    printf("one lane per chain: %.3f us per E_K; four lanes per chain: %.3f us per E_K (x%.2f); %s\n", ms[0] * 1e3 / reps, ms[1] * 1e3 / reps,
           ms[0] / ms[1], bad ? "RESULTS DIFFER" : "same results");
    return bad != 0;
}
