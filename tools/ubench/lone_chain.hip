// lone_chain.hip -- what does ONE wavefront alone on its SIMD pay per instruction of a dependent chain?  (round 4: the long belt-hash
// chain, profiles/r04_long_hash_ab.txt.)  Part 1: shader cycles (s_memtime) per instruction for chains of one instruction class.
// Part 2: one belt encryption by chain forms -- one lane (belt_encr), byte-per-lane quad (belt_encr_quad, the product of the long
// chain), the quad with the XOR G-boxes' reduction folded into the update, the quad with ONE v_perm_b32 per look-up address (64 KiB
// row table at LDS address 0) -- cycles per encryption, results compared.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/lone_chain.hip -o /tmp/lone_chain && /tmp/lone_chain
#include "../../bee2_amd/csrc/belt_dev.hpp"
#include <stdio.h>
using namespace bee2hip;
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define R4(s) s "\n" s "\n" s "\n" s "\n"
#define R16(s) R4(s) R4(s) R4(s) R4(s)
#define R64(s) R16(s) R16(s) R16(s) R16(s)

__device__ __forceinline__ unsigned long long cyc()
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <int P> __global__ __launch_bounds__(64) void k_class(unsigned long long *out, uint32_t seed)
{
    __shared__ uint32_t tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 2654435761u >> 7) & 0x3fcu;     // a pointer-chase table of dword offsets
    __syncthreads();
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = a ^ 0x55, d = a + 7, e = 0x0c0c0400u;
    uint32_t base = (uint32_t)(uintptr_t)tab;
    uint32_t sa = seed, sb = 12345;
    const unsigned long long t0 = cyc();
    for (int it = 0; it < 64; ++it) {
        if (P == 0) asm volatile(R64("v_add_u32 %0, %0, %1") : "+v"(a) : "v"(b));
        if (P == 1) asm volatile(R16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if (P == 2) asm volatile(R64("v_add_u32 %0, %0, %1\n s_nop 1") : "+v"(a) : "v"(b));
        if (P == 3) asm volatile(R64("v_xor_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1") : "+v"(a));
        if (P == 4) asm volatile(R64("v_xor_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") : "+v"(a));
        if (P == 5) asm volatile(R64("v_bfe_u32 %0, %0, 2, 8\n v_lshl_add_u32 %0, %0, 2, %1\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)") : "+v"(a) : "v"(base));
        if (P == 6) asm volatile(R64("v_perm_b32 %0, %0, %1, %2") : "+v"(a) : "v"(b), "v"(e));
        if (P == 7) asm volatile(R64("v_mov_b32_sdwa %0, %0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0") : "+v"(a));
        if (P == 8) asm volatile(R64("v_alignbit_b32 %0, %0, %0, 5") : "+v"(a));
        if (P == 9) asm volatile(R64("s_add_u32 %0, %0, %1") : "+s"(sa) : "s"(sb) : "scc");
        if (P == 10) asm volatile(R64("v_readlane_b32 s20, %0, 3\n s_nop 1\n v_add_u32 %0, s20, %0") : "+v"(a) :: "s20");
        if (P == 11) asm volatile(R64("v_and_b32 %0, 0x3fc, %0\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)") : "+v"(a));
        if (P == 12) asm volatile(R64("v_add_u32 %0, %0, %1\n s_nop 0") : "+v"(a) : "v"(b));
        if (P == 13) asm volatile(R64("v_add3_u32 %0, %0, %1, %1") : "+v"(a) : "v"(b));
        if (P == 14) asm volatile(R64("v_cndmask_b32 %0, %0, %1, vcc") : "+v"(a) : "v"(b));
        if (P == 15) asm volatile(R64("v_bfe_u32 %0, %0, %1, 8") : "+v"(a) : "v"(b));
        if (P == 16) asm volatile(R64("v_add_u32 %0, %0, %1\n v_bfe_u32 %0, %0, 2, 8\n v_lshl_add_u32 %0, %0, 2, %2\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n"
                                      "v_xor_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                                      "v_xor_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_xor_b32 %3, %3, %0") : "+v"(a) : "v"(b), "v"(base), "v"(c));
        if (P == 17) asm volatile(R64("v_add_u32 %0, %0, %1\n v_and_b32 %0, 0x3fc, %0\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n"
                                      "v_xor_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                                      "v_xor_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_xor_b32 %3, %3, %0") : "+v"(a) : "v"(b), "v"(base), "v"(c));
    }
    const unsigned long long t1 = cyc();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a ^ c ^ d ^ e ^ sa; }
}

// the quad form with the XOR G-boxes' reduction folded into the update: u = t ^ dpp(t); b ^= u; b ^= dpp'(u) -- the plain xor is the
// first wait state of the second DPP read, so the critical path loses one slot where the result is XORed in (4 of 7 G-boxes)
template <int R0> __device__ __forceinline__ void gq_xor(const uint8_t *lds, const BeltQuadLane &Q, uint32_t x, uint32_t &dst)
{
    const uint32_t b = __builtin_amdgcn_ubfe(x, Q.sh, 8u);
    uint32_t t = *reinterpret_cast<const uint32_t *>(lds + ((b << 2) + Q.off[R0]));
    asm volatile("v_xor_b32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_xor_b32 %0, %0, %1\n"
                 "s_nop 0\n"
                 "v_xor_b32_dpp %0, %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(dst), "+v"(t));
}
template <int I> __device__ __forceinline__ void round_fold(const uint8_t *lds, const BeltQuadLane &Q, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, const uint32_t (&K)[8])
{
    constexpr int o = 7 * I - 7;
    gq_xor<0>(lds, Q, a + K[(o + 0) & 7], b);
    gq_xor<2>(lds, Q, d + K[(o + 1) & 7], c);
    a -= gbox_quad<1>(lds, Q, b + K[(o + 2) & 7]);
    const uint32_t e = gbox_quad<2>(lds, Q, b + c + K[(o + 3) & 7]) ^ (uint32_t)I;
    b += e;
    c -= e;
    d += gbox_quad<1>(lds, Q, c + K[(o + 4) & 7]);
    gq_xor<2>(lds, Q, a + K[(o + 5) & 7], b);
    gq_xor<0>(lds, Q, d + K[(o + 6) & 7], c);
}
__device__ __forceinline__ void encr_quad_fold(const uint8_t *lds, const BeltQuadLane &Q, uint32_t (&x)[4], const uint32_t (&K)[8])
{
    uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
    round_fold<1>(lds, Q, a, b, c, d, K); round_fold<2>(lds, Q, b, d, a, c, K); round_fold<3>(lds, Q, d, c, b, a, K); round_fold<4>(lds, Q, c, a, d, b, K);
    round_fold<5>(lds, Q, a, b, c, d, K); round_fold<6>(lds, Q, b, d, a, c, K); round_fold<7>(lds, Q, d, c, b, a, K); round_fold<8>(lds, Q, c, a, d, b, K);
    x[0] = b; x[1] = d; x[2] = a; x[3] = c;
}

constexpr int ENC = 512;
// form 0: byte-per-lane quad (belt_encr_quad); 1: the same with the XOR boxes' reduction folded into the update; 2: plain lane
template <int FORM> __global__ __launch_bounds__(64) void k_encr(unsigned long long *out, uint32_t *res, const uint32_t *key)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[4096];
    BeltTabSmall::fill(smem, threadIdx.x, 64);
    __syncthreads();
    const BeltQuadLane QL(threadIdx.x & 3u);
    const BeltTabSmall T(smem);
    uint32_t x[4] = {1, 2, 3, 4}, K[8];
    for (int i = 0; i < 8; ++i) K[i] = key[i];
    const unsigned long long t0 = cyc();
    for (int it = 0; it < ENC; ++it) {
        if (FORM == 0) belt_encr_quad(smem, QL, x, K);
        if (FORM == 1) encr_quad_fold(smem, QL, x, K);
        if (FORM == 2) belt_encr(T, x, K);
    }
    const unsigned long long t1 = cyc();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (threadIdx.x < 16) for (int i = 0; i < 4; ++i) res[threadIdx.x * 4 + i] = x[i];
}

// form 3: the quad with the look-up address made by ONE v_perm_b32: 64 KiB table at LDS address 0, row b (256 bytes) = 64 dwords,
// dword s = rotl(H[b], 5 + 8 (s & 3)): lane l reads dword ((R0 + j) & 3) + 4 (l >> 2) of row "byte j of x": its own bank
template <int R0> __device__ __forceinline__ uint32_t gbox_perm(uint32_t x, uint32_t sel, const uint32_t (&off)[3])
{
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    const uint32_t ad = __builtin_amdgcn_perm(x, off[R0], sel);
    uint32_t t = *reinterpret_cast<const uint32_t *>(dyn + ad);
    t ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xF, 0xF, false);
    t ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xF, 0xF, false);
    return t;
}
template <int I> __device__ __forceinline__ void round_perm(uint32_t sel, const uint32_t (&off)[3], uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, const uint32_t (&K)[8])
{
    constexpr int o = 7 * I - 7;
    b ^= gbox_perm<0>(a + K[(o + 0) & 7], sel, off);
    c ^= gbox_perm<2>(d + K[(o + 1) & 7], sel, off);
    a -= gbox_perm<1>(b + K[(o + 2) & 7], sel, off);
    const uint32_t e = gbox_perm<2>(b + c + K[(o + 3) & 7], sel, off) ^ (uint32_t)I;
    b += e;
    c -= e;
    d += gbox_perm<1>(c + K[(o + 4) & 7], sel, off);
    b ^= gbox_perm<2>(a + K[(o + 5) & 7], sel, off);
    c ^= gbox_perm<0>(d + K[(o + 6) & 7], sel, off);
}
__global__ __launch_bounds__(64) void k_encr_perm(unsigned long long *out, uint32_t *res, const uint32_t *key)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    uint32_t *t = reinterpret_cast<uint32_t *>(dyn);
    for (int i = threadIdx.x; i < 256 * 64; i += 64) t[i] = rotl32c((uint32_t)c_beltH[i >> 6], 5 + 8 * (i & 3));
    __syncthreads();
    const unsigned j = threadIdx.x & 3u;
    const uint32_t sel = 0x0c0c0000u | ((4u + j) << 8);
    uint32_t off[3];
    for (int r = 0; r < 3; ++r) off[r] = 4u * (((r + j) & 3u) + 4u * (threadIdx.x >> 2));
    uint32_t x[4] = {1, 2, 3, 4}, K[8];
    for (int i = 0; i < 8; ++i) K[i] = key[i];
    const unsigned long long t0 = cyc();
    for (int it = 0; it < ENC; ++it) {
        uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
        round_perm<1>(sel, off, a, b, c, d, K); round_perm<2>(sel, off, b, d, a, c, K); round_perm<3>(sel, off, d, c, b, a, K); round_perm<4>(sel, off, c, a, d, b, K);
        round_perm<5>(sel, off, a, b, c, d, K); round_perm<6>(sel, off, b, d, a, c, K); round_perm<7>(sel, off, d, c, b, a, K); round_perm<8>(sel, off, c, a, d, b, K);
        x[0] = b; x[1] = d; x[2] = a; x[3] = c;
    }
    const unsigned long long t1 = cyc();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (threadIdx.x < 16) for (int i = 0; i < 4; ++i) res[threadIdx.x * 4 + i] = x[i];
}

int main()
{
    unsigned long long *d_out, h_out[2];
    uint32_t *d_res, *d_key, h_res[4][64], h_key[8];
    CHK(hipMalloc(&d_out, 16)); CHK(hipMalloc(&d_res, 256)); CHK(hipMalloc(&d_key, 32));
    uint8_t H[256];
    for (int i = 0; i < 256; ++i) H[i] = (uint8_t)(i * 167 + 13);            // a permutation of 0..255 (167 is odd): timing + cross-check only
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_beltH), H, 256));
    for (int i = 0; i < 8; ++i) h_key[i] = 0x9E3779B9u * (i + 1);
    CHK(hipMemcpy(d_key, h_key, 32, hipMemcpyHostToDevice));
    static const char *names[] = {"v_add_u32 dependent", "v_add_u32 x4 independent (per instr)", "v_add_u32 + s_nop 1 (per pair)", "v_xor_b32_dpp dependent + s_nop 1 (per pair)",
        "v_xor_b32_dpp dependent, no nop (timing only)", "bfe, lshl_add, ds_read, wait (per look-up)", "v_perm_b32 dependent", "v_mov_b32_sdwa dependent", "v_alignbit_b32 dependent",
        "s_add_u32 dependent", "v_readlane, s_nop 1, v_add (per triple)", "v_and, ds_read, wait (per look-up)", "v_add_u32 + s_nop 0 (per pair)", "v_add3_u32 dependent", "v_cndmask_b32 dependent",
        "v_bfe_u32 dependent", "G-box of the quad form, 9 instr (per G-box)", "G-box with a 1-instr address, 8 instr (per G-box)"};
#define RUNC(P) do { for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_class<P>, dim3(1), dim3(64), 0, 0, d_out, 77u + r); CHK(hipDeviceSynchronize()); } \
        CHK(hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost)); printf("%-58s %7.2f cycles\n", names[P], (double)h_out[0] / 4096.0 / (P == 1 ? 1.0 : 1.0)); } while (0)
    RUNC(0); RUNC(1); RUNC(2); RUNC(12); RUNC(3); RUNC(4); RUNC(5); RUNC(11); RUNC(6); RUNC(7); RUNC(8); RUNC(9); RUNC(10); RUNC(13); RUNC(14); RUNC(15); RUNC(16); RUNC(17);
#define RUNE(F, idx, name) do { for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_encr<F>, dim3(1), dim3(64), 0, 0, d_out, d_res, d_key); CHK(hipDeviceSynchronize()); } \
        CHK(hipMemcpy(h_out, d_out, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h_res[idx], d_res, 256, hipMemcpyDeviceToHost)); \
        printf("%-58s %7.0f cycles per encryption\n", name, (double)h_out[0] / ENC); } while (0)
    RUNE(2, 0, "belt_encr, one lane (4 look-ups per G-box)");
    RUNE(0, 1, "belt_encr_quad (product of the long chain)");
    RUNE(1, 2, "quad, XOR boxes' reduction folded into the update");
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_encr_perm), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_encr_perm, dim3(1), dim3(64), 65536, 0, d_out, d_res, d_key); CHK(hipDeviceSynchronize()); }
    CHK(hipMemcpy(h_out, d_out, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h_res[3], d_res, 256, hipMemcpyDeviceToHost));
    printf("%-58s %7.0f cycles per encryption\n", "quad, v_perm_b32 address, 64 KiB row table", (double)h_out[0] / ENC);
    int bad = 0;
    for (int f = 1; f < 4; ++f) for (int i = 0; i < 64; ++i) if (h_res[f][i] != h_res[0][i & 3]) { if (!bad) printf("form %d differs at %d: %08x vs %08x\n", f, i, h_res[f][i], h_res[0][i & 3]); bad++; }
    printf(bad ? "MISMATCH (%d)\n" : "all forms agree\n", bad);
    return bad != 0;
}
