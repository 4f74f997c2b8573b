// Does the VGPR file of gfx950 have operand-bank conflicts?  Each kernel issues 16 independent VALU ops per
// iteration on FIXED registers (named in the asm text, reserved through the clobber list) so that the source
// operands of every instruction sit at chosen distances d1, d2 from each other:
//     v_bitop3_b32 v[D], v[A], v[A + d1], v[A + d2]      (D = A: accumulate in place)
// If registers are banked by (index mod 4), d = 4 puts two sources on one bank and d1 = 4, d2 = 8 all three.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 8192
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

// registers v64..v127 are ours inside the asm block
#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
             "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
             "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
             "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

#define STR2(x) #x
#define STR(x) STR2(x)
// one op: destination/source A = 64 + i, sources at +D1, +D2 (wrapping inside 64..127 is the caller's job)
#define B3(i, D1, D2) "v_bitop3_b32 v" STR(i) ", v" STR(i) ", v[" STR(i) "+" STR(D1) "], v[" STR(i) "+" STR(D2) "] bitop3:0x96\n"
#define X2(i, D1)     "v_xor_b32 v" STR(i) ", v" STR(i) ", v[" STR(i) "+" STR(D1) "]\n"
#define AL(i, D1)     "v_alignbit_b32 v" STR(i) ", v" STR(i) ", v[" STR(i) "+" STR(D1) "], 7\n"
#define AD(i, D1)     "v_add_u32 v" STR(i) ", v" STR(i) ", v[" STR(i) "+" STR(D1) "]\n"
#define SH(i)         "v_lshrrev_b32 v" STR(i) ", 7, v" STR(i) "\n"
#define B3D(i, DD, D1, D2) "v_bitop3_b32 v[" STR(i) "+" STR(DD) "], v" STR(i) ", v[" STR(i) "+" STR(D1) "], v[" STR(i) "+" STR(D2) "] bitop3:0x96\n"
#define REP16(M, ...) M(64, __VA_ARGS__) M(65, __VA_ARGS__) M(66, __VA_ARGS__) M(67, __VA_ARGS__) M(68, __VA_ARGS__) M(69, __VA_ARGS__) M(70, __VA_ARGS__) M(71, __VA_ARGS__) \
                      M(72, __VA_ARGS__) M(73, __VA_ARGS__) M(74, __VA_ARGS__) M(75, __VA_ARGS__) M(76, __VA_ARGS__) M(77, __VA_ARGS__) M(78, __VA_ARGS__) M(79, __VA_ARGS__)
#define REP16_0(M) M(64) M(65) M(66) M(67) M(68) M(69) M(70) M(71) M(72) M(73) M(74) M(75) M(76) M(77) M(78) M(79)

template <int K> __global__ void kern(uint32_t *out, uint32_t seed)
{
    // initialise v64..v127 from lane data
    asm volatile("v_mov_b32 v64, %0\n v_mov_b32 v65, %0\n v_mov_b32 v66, %0\n v_mov_b32 v67, %0\n"
                 "v_mov_b32 v68, %0\n v_mov_b32 v69, %0\n v_mov_b32 v70, %0\n v_mov_b32 v71, %0\n"
                 "v_mov_b32 v72, %0\n v_mov_b32 v73, %0\n v_mov_b32 v74, %0\n v_mov_b32 v75, %0\n"
                 "v_mov_b32 v76, %0\n v_mov_b32 v77, %0\n v_mov_b32 v78, %0\n v_mov_b32 v79, %0\n"
                 "v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %0\n v_mov_b32 v83, %0\n"
                 "v_mov_b32 v84, %0\n v_mov_b32 v85, %0\n v_mov_b32 v86, %0\n v_mov_b32 v87, %0\n"
                 "v_mov_b32 v88, %0\n v_mov_b32 v89, %0\n v_mov_b32 v90, %0\n v_mov_b32 v91, %0\n"
                 "v_mov_b32 v92, %0\n v_mov_b32 v93, %0\n v_mov_b32 v94, %0\n v_mov_b32 v95, %0\n"
                 "v_mov_b32 v96, %0\n v_mov_b32 v97, %0\n v_mov_b32 v98, %0\n v_mov_b32 v99, %0\n"
                 "v_mov_b32 v100, %0\n v_mov_b32 v101, %0\n v_mov_b32 v102, %0\n v_mov_b32 v103, %0\n"
                 "v_mov_b32 v104, %0\n v_mov_b32 v105, %0\n v_mov_b32 v106, %0\n v_mov_b32 v107, %0\n"
                 "v_mov_b32 v108, %0\n v_mov_b32 v109, %0\n v_mov_b32 v110, %0\n v_mov_b32 v111, %0\n"
                 :: "v"(seed + threadIdx.x) : CLOB);
    for (int it = 0; it < ITERS; ++it) {
        if (K == 0)  asm volatile(REP16(B3, 17, 34) ::: CLOB);      // 3 sources, banks (i, i+1, i+2) mod 4
        if (K == 1)  asm volatile(REP16(B3, 16, 33) ::: CLOB);      // A and B on one bank (d1 = 16)
        if (K == 2)  asm volatile(REP16(B3, 16, 32) ::: CLOB);      // all three on one bank
        if (K == 3)  asm volatile(REP16(B3, 4, 8) ::: CLOB);        // all three, small distances
        if (K == 4)  asm volatile(REP16(B3, 1, 2) ::: CLOB);        // consecutive registers
        if (K == 5)  asm volatile(REP16(B3, 2, 4) ::: CLOB);        // d = 2, 4  (bank = idx mod 2 ?)
        if (K == 6)  asm volatile(REP16(B3, 8, 16) ::: CLOB);
        if (K == 7)  asm volatile(REP16(X2, 17) ::: CLOB);          // 2 sources, different banks
        if (K == 8)  asm volatile(REP16(X2, 16) ::: CLOB);          // 2 sources, same bank
        if (K == 9)  asm volatile(REP16(AL, 17) ::: CLOB);
        if (K == 10) asm volatile(REP16(AL, 16) ::: CLOB);
        if (K == 11) asm volatile(REP16(AD, 17) ::: CLOB);
        if (K == 12) asm volatile(REP16(AD, 16) ::: CLOB);
        if (K == 13) asm volatile(REP16_0(SH) ::: CLOB);            // one VGPR source
        if (K == 14) asm volatile(REP16(B3D, 17, 18, 35) ::: CLOB); // dst on a 4th register class
        if (K == 15) asm volatile(REP16(B3D, 16, 17, 34) ::: CLOB); // dst on the bank of source A (different register)
        if (K == 16) asm volatile(REP16(B3, 3, 6) ::: CLOB);
        if (K == 17) asm volatile(REP16(B3, 5, 10) ::: CLOB);
        if (K == 18) asm volatile(REP16(B3, 1, 3) ::: CLOB);
        if (K == 19) asm volatile(REP16(B3, 2, 3) ::: CLOB);
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v64, v65\n v_xor_b32 %0, %0, v66\n v_xor_b32 %0, %0, v67\n v_xor_b32 %0, %0, v81\n" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int g_wps = 4;
template <int K> int run(const char *name)
{
    uint32_t *d; CHK(hipMalloc(&d, 1024 * 1024 * 16));
    int blocks = 256 * g_wps, threads = 256;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    kern<K><<<blocks, threads>>>(d, 12345); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); kern<K><<<blocks, threads>>>(d, 12345); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double waves = (double)blocks * threads / 64, instr = waves * ITERS * 16;
    printf("[wps=%d] %-44s %7.3f ms  %5.2f cyc/wave-instr/SIMD (@2.4GHz)\n", g_wps, name, ms, ms * 1e-3 * 2.4e9 * 1024 / instr);
    hipFree(d); return 0;
}
int main()
{
    for (int w : {4, 8}) {
        g_wps = w;
        run<13>("lshrrev (1 VGPR source)");
        run<7>("xor    src d=17 (banks differ if mod 4)");
        run<8>("xor    src d=16 (same bank if mod 4)");
        run<11>("add    src d=17");
        run<12>("add    src d=16");
        run<9>("alignbit src d=17");
        run<10>("alignbit src d=16");
        run<0>("bitop3 src d=17,34 (3 banks)");
        run<1>("bitop3 src d=16,33 (2 on one bank)");
        run<2>("bitop3 src d=16,32 (3 on one bank)");
        run<3>("bitop3 src d=4,8");
        run<6>("bitop3 src d=8,16");
        run<4>("bitop3 src d=1,2");
        run<5>("bitop3 src d=2,4");
        run<16>("bitop3 src d=3,6");
        run<17>("bitop3 src d=5,10");
        run<18>("bitop3 src d=1,3");
        run<19>("bitop3 src d=2,3");
        run<14>("bitop3 dst d=17, src 0,18,35");
        run<15>("bitop3 dst d=16 (bank of A), src 0,17,34");
    }
    return 0;
}
