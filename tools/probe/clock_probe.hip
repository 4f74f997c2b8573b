// clock_probe.hip -- bench-side diagnostic, NOT part of libbee2hip.so: one wavefront spins for `us` microseconds of
// s_memrealtime (100 MHz) on `stream` and reports how many shader cycles (s_memtime) went by.  Launched on a second stream
// beside the kernels under test it gives the clock the chip sustained under that load (DVFS: MI355X_MICROARCH.md).
// Built by __graft_entry__.build() into bee2_amd/lib/libb2hprobe.so; bench.py and tools/ load it with ctypes.
#include <hip/hip_runtime.h>

__global__ void clock_probe_kernel(unsigned long long *out, unsigned long long ticks)
{
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    do {
        __builtin_amdgcn_s_sleep(32);
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    } while (r1 - r0 < ticks);
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

// d_out16: two u64 {shader cycles, 100 MHz ticks}; returns the hipError_t of the launch
extern "C" __attribute__((visibility("default"))) int b2h_clock_probe(void *d_out16, unsigned us, void *stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long *)d_out16,
                       (unsigned long long)us * 100ull);
    return (int)hipGetLastError();
}
