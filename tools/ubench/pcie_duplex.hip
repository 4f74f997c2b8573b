// pcie_duplex.hip -- can this box move bytes host->device and device->host AT THE SAME TIME, and by which engines?
// (VERDICT r02 item 8 wants a double-buffered duplex pipeline behind the host-pointer API; profiles/r01_pcie_duplex.txt
// found no overlap for two hipMemcpy issued from two Python threads.)  One process, pinned host buffers, 1 GiB each way:
//   a  H2D alone (hipMemcpyAsync, SDMA)              b  D2H alone (SDMA)
//   c  H2D (SDMA, stream 1) || D2H (SDMA, stream 2)
//   d  H2D (SDMA) || D2H by a copy kernel writing mapped host memory
//   e  H2D by a copy kernel reading mapped host memory || D2H (SDMA)
//   f  both by copy kernels on two streams              g  ONE kernel: reads mapped host src, writes mapped host dst
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pcie_duplex tools/ubench/pcie_duplex.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main()
{
    const size_t N = (size_t)1 << 30, n16 = N / 16;
    void *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc(&h_in, N, hipHostMallocMapped | hipHostMallocPortable));
    CK(hipHostMalloc(&h_out, N, hipHostMallocMapped | hipHostMallocPortable));
    memset(h_in, 1, N); memset(h_out, 0, N);
    CK(hipMalloc(&d_a, N)); CK(hipMalloc(&d_b, N));
    CK(hipMemset(d_b, 2, N));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const dim3 grid(2048), block(256);
    auto run = [&](const char *name, int up, int down) -> int {      // up/down: 0 none, 1 SDMA, 2 kernel
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            if (up == 1) CK(hipMemcpyAsync(d_a, h_in, N, hipMemcpyHostToDevice, s1));
            if (up == 2) hipLaunchKernelGGL(copy_kernel, grid, block, 0, s1, (const uint4 *)h_in, (uint4 *)d_a, n16);
            if (down == 1) CK(hipMemcpyAsync(h_out, d_b, N, hipMemcpyDeviceToHost, s2));
            if (down == 2) hipLaunchKernelGGL(copy_kernel, grid, block, 0, s2, (const uint4 *)d_b, (uint4 *)h_out, n16);
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < best) best = dt;
        }
        const double gib = ((up ? 1 : 0) + (down ? 1 : 0)) * (double)N / (1 << 30);
        printf("%-64s %7.2f ms  %6.1f GiB/s aggregate\n", name, best * 1e3, gib / best);
        return 0;
    };
    if (run("a  H2D alone, SDMA", 1, 0)) return 1;
    if (run("b  D2H alone, SDMA", 0, 1)) return 1;
    if (run("a' H2D alone, copy kernel reading mapped host memory", 2, 0)) return 1;
    if (run("b' D2H alone, copy kernel writing mapped host memory", 0, 2)) return 1;
    if (run("c  H2D SDMA || D2H SDMA (two streams)", 1, 1)) return 1;
    if (run("d  H2D SDMA || D2H copy kernel", 1, 2)) return 1;
    if (run("e  H2D copy kernel || D2H SDMA", 2, 1)) return 1;
    if (run("f  H2D copy kernel || D2H copy kernel (two streams)", 2, 2)) return 1;
    {   // g: one kernel, host -> host through the GPU (what a zero-copy batch kernel would do)
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(copy_kernel, grid, block, 0, s1, (const uint4 *)h_in, (uint4 *)h_out, n16);
            CK(hipStreamSynchronize(s1));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < best) best = dt;
        }
        printf("%-64s %7.2f ms  %6.1f GiB/s aggregate\n", "g  one kernel: mapped host src -> mapped host dst", best * 1e3, 2.0 / best);
    }
    // pageable source, for reference (what a C caller of the host-pointer API hands over)
    {
        char *p = (char *)malloc(N); memset(p, 3, N);
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipMemcpy(d_a, p, N, hipMemcpyHostToDevice));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < best) best = dt;
        }
        printf("%-64s %7.2f ms  %6.1f GiB/s\n", "h  H2D alone, hipMemcpy from PAGEABLE memory", best * 1e3, 1.0 / best);
        const auto t0 = std::chrono::steady_clock::now();
        memcpy(h_in, p, N);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%-64s %7.2f ms  %6.1f GiB/s\n", "i  CPU memcpy pageable -> pinned, one thread", dt * 1e3, 1.0 / dt);
        // j: two host threads, blocking hipMemcpy of PAGEABLE memory each way at the same time (what a pipeline behind the
        //    host-pointer API would have to rely on, the caller's buffers being ordinary memory)
        char *q = (char *)malloc(N); memset(q, 0, N);
        for (int mode = 0; mode < 2; ++mode) {
            double bestj = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                const auto t1 = std::chrono::steady_clock::now();
                std::thread up([&] { if (mode == 0) (void)hipMemcpy(d_a, p, N, hipMemcpyHostToDevice);
                                     else { (void)hipMemcpyAsync(d_a, p, N, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); } });
                std::thread dn([&] { if (mode == 0) (void)hipMemcpy(q, d_b, N, hipMemcpyDeviceToHost);
                                     else { (void)hipMemcpyAsync(q, d_b, N, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2); } });
                up.join(); dn.join();
                const double dtj = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
                if (dtj < bestj) bestj = dtj;
            }
            printf("%-64s %7.2f ms  %6.1f GiB/s aggregate\n", mode == 0 ? "j  two threads: hipMemcpy PAGEABLE H2D || hipMemcpy PAGEABLE D2H"
                                                                         : "k  two threads: hipMemcpyAsync PAGEABLE on two streams + sync", bestj * 1e3, 2.0 / bestj);
        }
        // l: chunked pageable pipeline in miniature: 16 MiB chunks, thread A copies chunk i up while thread B copies chunk i-1 down
        {
            const size_t CH = (size_t)16 << 20, nch = N / CH;
            double bestl = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                const auto t1 = std::chrono::steady_clock::now();
                std::thread up([&] { for (size_t c = 0; c < nch; ++c) (void)hipMemcpyAsync((char *)d_a + c * CH, p + c * CH, CH, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
                std::thread dn([&] { for (size_t c = 0; c < nch; ++c) (void)hipMemcpyAsync(q + c * CH, (char *)d_b + c * CH, CH, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2); });
                up.join(); dn.join();
                const double dtl = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
                if (dtl < bestl) bestl = dtl;
            }
            printf("%-64s %7.2f ms  %6.1f GiB/s aggregate\n", "l  two threads, 16 MiB chunks, PAGEABLE, hipMemcpyAsync", bestl * 1e3, 2.0 / bestl);
        }
        free(q);
        free(p);
    }
    return 0;
}
