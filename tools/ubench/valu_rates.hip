// VALU throughput micro-benchmark for gfx950: which integer-multiply flavour should the
// GF(p) code be built on?  Each kernel runs a long chain of independent ops per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template<int OP> __global__ void k(uint32_t* out, uint32_t seed){
    uint32_t a[8]; uint64_t w[8]; double d[8];
    for(int i=0;i<8;i++){ a[i]=seed*(i+1)+threadIdx.x; w[i]=a[i]*0x9E3779B97F4A7C15ull; d[i]=(double)a[i]; }
    uint32_t b=seed|1, c=seed^0x5bd1e995;
    for(int it=0; it<ITERS; ++it){
#pragma unroll
        for(int i=0;i<8;i++){
            if(OP==0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
            if(OP==1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==3) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==5) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i+1)&7]));
            if(OP==6) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(c) : "v"(b), "v"(b) : "vcc");
            if(OP==7) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==8) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
            if(OP==9) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==10) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "s10","s11");
            if(OP==11) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i+1)&7]));
            if(OP==12) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==13) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==14) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==15) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==16) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            if(OP==17) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if(OP==18) asm volatile("v_add_co_u32 %0, s[10:11], %0, %1" : "+v"(a[i]) : "v"(b) : "s10","s11");
            if(OP==19) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==20) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
            if(OP==21) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==22) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==23) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(a[i]) : "v"(b));
            if(OP==24) asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(w[i]));
            if(OP==25) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(a[i]));
            if(OP==26) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==27) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            if(OP==28) asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(a[i]) : "v"(b));
            if(OP==29) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
            if(OP==30) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==31) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(b));
            if(OP==32) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==33) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==34) asm volatile("v_addc_co_u32 %0, s[10:11], %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10","s11");
            if(OP==35) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if(OP==36) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==37) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if(OP==38) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(w[(i+1)&7]));
        }
    }
    if(OP==99){ long long t0=clock64(), w0=wall_clock64(); for(int it=0;it<ITERS*16;++it){ asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); }
        long long t1=clock64(), w1=wall_clock64(); if(threadIdx.x==0&&blockIdx.x==0){ out[1<<20]=(uint32_t)(t1-t0); out[(1<<20)+1]=(uint32_t)(w1-w0);} }
    uint32_t r=c; for(int i=0;i<8;i++){ r^=a[i]^(uint32_t)w[i]^(uint32_t)(w[i]>>32)^(uint32_t)d[i]; }
    out[blockIdx.x*blockDim.x+threadIdx.x]=r;
}
static int g_wps=8;
template<int OP> int run(const char* name, int ops_per){
    uint32_t* d; CHK(hipMalloc(&d, 1024*1024*16));
    int blocks=256*g_wps, threads=256;   // g_wps WG/CU x 4 waves = g_wps waves/SIMD
    hipEvent_t e0,e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<OP><<<blocks,threads>>>(d,12345); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); k<OP><<<blocks,threads>>>(d,12345); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms,e0,e1));
    double waves=(double)blocks*threads/64, instr=waves*ITERS*8*ops_per;
    // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
    double cyc = ms*1e-3*2.4e9*1024/instr; if(g_wps!=8) { printf("[wps=%d] ", g_wps); }
    printf("%-22s %8.3f ms  %6.2f cyc/wave-instr/SIMD (@2.4GHz)  %.2f Tinstr-lanes/s\n", name, ms, cyc, instr*64/ms*1e-9);
    hipFree(d); return 0;
}
// Mixed-op kernels: do two instruction classes share one issue pipe (time = sum) or overlap
// (time = max)?  Each iteration issues NA ops of class A on a[] and NB ops of class B on e[],
// all chains independent.  MODE 0: alignbit + xor, 1: mad_u64_u32 + xor, 2: addc_co + xor,
// 3: ds_read_b32 + xor, 4: ds_read_b32 + alignbit.
template<int MODE, int NA, int NB> __global__ void kmix(uint32_t* out, uint32_t seed){
    __shared__ uint32_t lds[4096];
    uint32_t a[8], e[8]; uint64_t w[8];
    for(int i=0;i<8;i++){ a[i]=seed*(i+1)+threadIdx.x; e[i]=a[i]^0x1234567u; w[i]=a[i]*0x9E3779B97F4A7C15ull; }
    for(int i=threadIdx.x;i<4096;i+=blockDim.x) lds[i]=i*4;   // every loaded value is again a valid byte address
    __syncthreads();
    uint32_t b=seed|1;
    uint32_t addr=(threadIdx.x&31)*4;                       // bank-private column: conflict free
    for(int it=0; it<ITERS; ++it){
#pragma unroll
        for(int i=0;i<8;i++){
            if(i<NA){
                if(MODE==0) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
                if(MODE==1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
                if(MODE==2) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if(MODE==3||MODE==4) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i]) : "v"(addr), "n"(0) : "memory");
            }
#pragma unroll
            for(int r=0;r<NB/8;r++){
                if(MODE==4) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(e[i]) : "v"(b));
                else        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(e[i]) : "v"(b));
            }
        }
        if(MODE==3||MODE==4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    uint32_t r=0; for(int i=0;i<8;i++){ r^=a[i]^e[i]^(uint32_t)w[i]^(uint32_t)(w[i]>>32); }
    out[blockIdx.x*blockDim.x+threadIdx.x]=r;
}
template<int MODE, int NA, int NB> int runmix(const char* name){
    uint32_t* d; CHK(hipMalloc(&d, 1024*1024*16));
    int blocks=256*g_wps, threads=256;
    hipEvent_t e0,e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    kmix<MODE,NA,NB><<<blocks,threads>>>(d,12345); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); kmix<MODE,NA,NB><<<blocks,threads>>>(d,12345); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms,e0,e1));
    double waves=(double)blocks*threads/64;
    double cyc = ms*1e-3*2.4e9*1024/(waves*ITERS);        // cycles per loop iteration per SIMD-resident wave slot
    printf("mix %-34s A=%d B=%-2d %8.3f ms  %6.2f cyc/iteration/SIMD (@2.4GHz)\n", name, NA, NB, ms, cyc);
    hipFree(d); return 0;
}
// Class-switch cost: 8 alignbit + 8 xor per iteration, issued in runs of G of each class
// (G=1: a x a x ..., G=8: aaaaaaaa xxxxxxxx).  Same instruction totals, only the order changes.
template<int G> __global__ void kgrp(uint32_t* out, uint32_t seed){
    uint32_t a[8], e[8];
    for(int i=0;i<8;i++){ a[i]=seed*(i+1)+threadIdx.x; e[i]=a[i]^0x1234567u; }
    uint32_t b=seed|1;
    for(int it=0; it<ITERS; ++it){
#pragma unroll
        for(int g=0; g<8; g+=G){
#pragma unroll
            for(int i=g;i<g+G;i++) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
#pragma unroll
            for(int i=g;i<g+G;i++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(e[i]) : "v"(b));
        }
    }
    uint32_t r=0; for(int i=0;i<8;i++) r^=a[i]^e[i];
    out[blockIdx.x*blockDim.x+threadIdx.x]=r;
}
template<int G> int rungrp(){
    uint32_t* d; CHK(hipMalloc(&d, 1024*1024*16));
    int blocks=256*g_wps, threads=256;
    hipEvent_t e0,e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    kgrp<G><<<blocks,threads>>>(d,12345); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); kgrp<G><<<blocks,threads>>>(d,12345); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms,e0,e1));
    double waves=(double)blocks*threads/64;
    printf("[wps=%d] 8 alignbit + 8 xor in runs of %d: %8.3f ms  %6.2f cyc/iteration/SIMD (@2.4GHz)\n", g_wps, G, ms, ms*1e-3*2.4e9*1024/(waves*ITERS));
    hipFree(d); return 0;
}
int main(int argc,char**argv){
    if(argc>1 && argv[1][0]=='m'){   // pipe-sharing test: A alone, B alone, A+B
        runmix<0,8,0>("alignbit");  runmix<0,0,8>("xor");  runmix<0,0,16>("xor"); runmix<0,8,8>("alignbit + xor"); runmix<0,8,16>("alignbit + xor");
        runmix<1,8,0>("mad_u64_u32"); runmix<1,8,8>("mad_u64_u32 + xor"); runmix<1,8,16>("mad_u64_u32 + xor");
        runmix<2,8,0>("addc_co"); runmix<2,8,8>("addc_co + xor");
        runmix<3,8,0>("ds_read_b32"); runmix<3,8,8>("ds_read_b32 + xor"); runmix<3,8,16>("ds_read_b32 + xor");
        runmix<4,0,8>("alignbit"); runmix<4,8,8>("ds_read_b32 + alignbit");
        return 0; }
    if(argc>1 && argv[1][0]=='g'){ for(int w : {3,8}){ g_wps=w; rungrp<1>(); rungrp<2>(); rungrp<4>(); rungrp<8>(); } return 0; }
    if(argc>1){ for(int w : {1,2,4,8}){ g_wps=w; run<4>("v_add_u32",1); run<9>("v_bitop3_b32",1); run<8>("v_alignbit_b32",1); run<0>("v_mad_u64_u32",1);} return 0; }
    run<4>("v_add_u32",1); run<8>("v_alignbit_b32",1); run<9>("v_bitop3_b32",1);
    run<0>("v_mad_u64_u32 (vcc)",1); run<10>("v_mad_u64_u32 (sgpr)",1);
    run<1>("v_mul_lo_u32",1); run<2>("v_mul_hi_u32",1);
    run<3>("v_mad_u32_u24",1); run<12>("v_mad_i32_i24",1); run<7>("v_mul_hi_u32_u24",1);
    run<5>("v_fma_f64",1); run<6>("add_co+addc_co",2); run<11>("v_lshl_add_u64",1);
    run<13>("v_dot4_u32_u8",1); run<14>("v_pk_mul_lo_u16",1); run<15>("v_pk_mad_u16",1);
    run<16>("v_add_co_u32 vcc",1); run<17>("v_addc_co_u32 vcc",1); run<18>("v_add_co_u32 sgpr",1); run<34>("v_addc_co_u32 sgpr",1);
    run<19>("v_add3_u32",1); run<20>("v_lshl_add_u32",1); run<21>("v_and_or_b32",1); run<22>("v_perm_b32",1);
    run<23>("v_alignbyte_b32",1); run<24>("v_lshlrev_b64",1); run<25>("v_lshrrev_b32",1); run<26>("v_xor_b32",1);
    run<27>("v_cndmask_b32",1); run<28>("v_lshl_or_b32",1); run<29>("v_bfe_u32",1); run<30>("v_mov_b32",1);
    run<31>("v_lshlrev_b32_sdwa",1); run<32>("v_pk_add_u16",1); run<33>("v_sub_u32",1); run<35>("v_mul_u32_u24",1);
    run<36>("v_xad_u32",1); run<37>("v_fma_f32",1); run<38>("v_pk_fma_f32",1);
    { uint32_t* d; hipMalloc(&d,(1<<20)*4+64); k<99><<<256*8,256>>>(d,1); hipDeviceSynchronize(); uint32_t h[2]; hipMemcpy(h,d+(1<<20),8,hipMemcpyDeviceToHost);
      printf("clock64 delta %u, wall_clock64 delta %u (100 MHz) -> shader clock %.3f GHz under load\n", h[0], h[1], h[0]/(h[1]*10.0)); }
    return 0;
}
