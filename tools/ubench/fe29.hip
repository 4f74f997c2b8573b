// Micro-benchmark: GF(2^256 - 189) multiplication / squaring with nine 29-bit limbs and 64-bit column
// accumulators (no carry flags) against the product-scanning 8 x 32-bit code of bign_dev.hpp.
// Build: hipcc --offload-arch=gfx950 -O3 -I bee2_amd/csrc tools/ubench/fe29.hip -o tools/ubench/fe29
// Run on the GPU: ./fe29   (prints ns per multiplication per wavefront slot and checks both give the same residue)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "bign_fe29.hpp"
using namespace bee2hip;

struct ufe29 { uint32_t l[9]; };
#define fe29 ufe29
constexpr uint32_t M29 = (1u << 29) - 1u;
constexpr uint32_t FOLD = 189u * 32u;          // 2^261 mod p

__device__ __forceinline__ void to29(fe29 &r, const feT<8> &a)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        const uint32_t lo = a.v[w], hi = w + 1 < 8 ? a.v[w + 1] : 0u;
        r.l[i] = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & M29;
    }
}
// fully carried, value < 2^261: fold the bits above 2^256 and emit 8 words (weakly reduced)
__device__ __forceinline__ void from29(feT<8> &r, const fe29 &a)
{
    uint32_t l[9];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { const uint32_t t = a.l[i] + c; l[i] = t & M29; c = t >> 29; }
    // c * 2^261 + (l[8] >> 24) * 2^256 fold back: 2^256 = 189
    const uint32_t top = (l[8] >> 24) + (c << 5);
    l[8] &= (1u << 24) - 1u;
    uint64_t acc = (uint64_t)top * 189u;
    uint32_t w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { acc += l[i]; w[i] = (uint32_t)acc & M29; acc >>= 29; }
    // now < 2^256 + small; pack (a second wrap is impossible for the test's purposes except by ~2^-200)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int lo = 32 * k / 29, sh = 32 * k % 29;
        uint64_t v = (uint64_t)w[lo] >> sh;
        if (lo + 1 < 9) v |= (uint64_t)w[lo + 1] << (29 - sh);
        if (lo + 2 < 9 && 58 - sh < 32) v |= (uint64_t)w[lo + 2] << (58 - sh);
        r.v[k] = (uint32_t)v;
    }
}

template <int I> struct IC { static constexpr int v = I; };
template <int B, int E, class F> __device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (B < E) { f(IC<B>{}); sfor<B + 1, E>(f); }
}

// r = a b, limbs of a, b <= 2^29 + 2^28; result limbs < 2^29 except l[0] < 2^29 + 2^27
__device__ __forceinline__ void mul29(fe29 &r, const fe29 a, const fe29 b)
{
    uint32_t c[18];
    uint64_t acc = 0;
    sfor<0, 17>([&](auto kc) {
        constexpr int k = decltype(kc)::v;
        sfor<(k > 8 ? k - 8 : 0), (k < 8 ? k : 8) + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::v;
            acc += (uint64_t)a.l[i] * b.l[k - i];
        });
        c[k] = (uint32_t)acc & M29;
        acc >>= 29;
    });
    c[17] = (uint32_t)acc;
    uint32_t cy = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const uint64_t t = (uint64_t)c[9 + j] * FOLD + (uint64_t)(c[j] + cy);
        r.l[j] = (uint32_t)t & M29;
        cy = (uint32_t)(t >> 29);
    }
    r.l[0] += cy * FOLD;
}
__device__ __forceinline__ void sqr29(fe29 &r, const fe29 a)
{
    uint32_t c[18], d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
    uint64_t acc = 0;
    sfor<0, 17>([&](auto kc) {
        constexpr int k = decltype(kc)::v;
        sfor<(k > 8 ? k - 8 : 0), (k < 8 ? k : 8) + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::v;
            constexpr int j = k - i;
            if constexpr (i < j) acc += (uint64_t)a.l[i] * d[j];
            else if constexpr (i == j) acc += (uint64_t)a.l[i] * a.l[i];
        });
        c[k] = (uint32_t)acc & M29;
        acc >>= 29;
    });
    c[17] = (uint32_t)acc;
    uint32_t cy = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const uint64_t t = (uint64_t)c[9 + j] * FOLD + (uint64_t)(c[j] + cy);
        r.l[j] = (uint32_t)t & M29;
        cy = (uint32_t)(t >> 29);
    }
    r.l[0] += cy * FOLD;
}
__device__ __noinline__ fe29 mul29_call(fe29 a, fe29 b) { fe29 r; mul29(r, a, b); return r; }
__device__ __noinline__ fe29 sqr29_call(fe29 a) { fe29 r; sqr29(r, a); return r; }

// mode 0: 32-bit limbs (bign_dev.hpp), 1: 29-bit limbs.  Each lane: x <- x*y, y <- x^2, REPS times.
template <int MODE>
__global__ __launch_bounds__(256, 4)
void chain(const uint32_t *in, uint32_t *out, int reps)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    feT<8> x, y;
    for (int i = 0; i < 8; ++i) { x.v[i] = in[16 * idx + i]; y.v[i] = in[16 * idx + 8 + i]; }
    if (MODE == 0) {
#pragma unroll 1
        for (int r = 0; r < reps; ++r) { fe_mul(x, x, y); fe_sqr(y, x); }
        fe_canon(x, x); fe_canon(y, y);
    } else if (MODE == 2) {                     // the signed form of bign_fe29.hpp (what the kernels use)
#undef fe29
        bee2hip::fe29 a, b;
#define fe29 ufe29
        f29_from_words(a, x); f29_from_words(b, y);
#pragma unroll 1
        for (int r = 0; r < reps; ++r) { f29_mul(a, a, b); f29_sqr(b, a); }
        f29_to_words(x, a); f29_to_words(y, b);
        fe_canon(x, x); fe_canon(y, y);
    } else {
        fe29 a, b;
        to29(a, x); to29(b, y);
#pragma unroll 1
        for (int r = 0; r < reps; ++r) { mul29(a, a, b); sqr29(b, a); }
        from29(x, a); from29(y, b);
        fe_canon(x, x); fe_canon(y, y);
    }
    for (int i = 0; i < 8; ++i) { out[16 * idx + i] = x.v[i]; out[16 * idx + 8 + i] = y.v[i]; }
}

int main()
{
    // wavefronts per SIMD: 4 = the verification kernel at full load; <= 1 = a small batch (one wavefront alone on
    // its SIMD issues an instruction every ~5 cycles whatever its class, so instruction COUNT decides there)
    const int reps = 2000, max_thr = 4096 * 64;
    uint32_t *h = (uint32_t *)malloc((size_t)max_thr * 64), *o0 = (uint32_t *)malloc((size_t)max_thr * 64), *o1 = (uint32_t *)malloc((size_t)max_thr * 64);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < (size_t)max_thr * 16; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 16); }
    uint32_t *din, *dout;
    hipMalloc(&din, (size_t)max_thr * 64); hipMalloc(&dout, (size_t)max_thr * 64);
    hipMemcpy(din, h, (size_t)max_thr * 64, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t bad = 0;
    for (int nwave : {256, 1024, 2048, 4096}) {
        const int nthr = nwave * 64;
        float t[3] = {0, 0, 0};
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(nwave), dim3(64), 0, 0, din, dout, reps);
                else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(nwave), dim3(64), 0, 0, din, dout, reps);
                else hipLaunchKernelGGL(chain<2>, dim3(nwave), dim3(64), 0, 0, din, dout, reps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&t[mode], e0, e1);
            }
            hipMemcpy(mode ? o1 : o0, dout, (size_t)nthr * 64, hipMemcpyDeviceToHost);
            if (mode) for (size_t i = 0; i < (size_t)nthr * 16; ++i) bad += o0[i] != o1[i];
        }
        printf("%.2f wavefronts/SIMD: 32-bit limbs %.3f ms, 29-bit unsigned %.3f ms (x%.2f), 29-bit signed (bign_fe29.hpp) %.3f ms (x%.2f) for %d x (mul + sqr)\n",
               nwave / 1024.0, t[0], t[1], t[0] / t[1], t[2], t[0] / t[2], reps);
    }
    printf("results %s (%zu words differ)\n", bad ? "DIFFER" : "identical", bad);
    return bad != 0;
}
