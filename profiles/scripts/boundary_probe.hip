// What stretches a dependent kernel boundary under load?  One chain of dependent launch pairs (a small kernel, then a one-workgroup kernel that reads its output) on a stream of
// its own, timed alone and beside background work on three other streams: (1) arithmetic only, (2) streaming WRITES, (3) streaming READS, (4) short kernels that write a little.
// The release at a kernel's end writes back every dirty line of the L2s, whoever wrote it; the acquire at the next kernel's start invalidates them.
// build: hipcc -O3 --offload-arch=gfx950 -pthread profiles/scripts/boundary_probe.hip -o build/exp/boundary_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void __launch_bounds__(512) step_like(unsigned* out) {
    __shared__ unsigned s[6000];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = s[threadIdx.x * 3] + out[(blockIdx.x * 64 + threadIdx.x + 7) % (gridDim.x * 64)];
}
__global__ void __launch_bounds__(1024) control_like(unsigned* out, int n) {
    __shared__ unsigned s[64];
    if (threadIdx.x < 64) s[threadIdx.x] = 0;
    __syncthreads();
    unsigned a = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += out[i];
    atomicAdd(&s[threadIdx.x % 64], a);
    __syncthreads();
    if (threadIdx.x < 64) out[threadIdx.x] = s[threadIdx.x];
}
__global__ void __launch_bounds__(256) bg_alu(unsigned* sink, int iters) {
    unsigned a = threadIdx.x + blockIdx.x, b = 0x9E3779B9u;
    for (int i = 0; i < iters; ++i) { a = a * 1664525u + 1013904223u; b ^= a >> 7; b += a; }
    if (a + b == 0x12345u) sink[0] = a;
}
__global__ void __launch_bounds__(256) bg_write(uint4* dst, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4(seed, (unsigned)i, seed ^ 7u, 1u);
}
__global__ void __launch_bounds__(256) bg_read(const uint4* src, size_t n, unsigned* sink) {
    unsigned a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; a += v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345u) sink[0] = a;
}
int main() {
    hipStream_t chain_st; (void)hipStreamCreateWithFlags(&chain_st, hipStreamNonBlocking);
    int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipStream_t chain_hi; (void)hipStreamCreateWithPriority(&chain_hi, hipStreamNonBlocking, prio_hi);
    printf("stream priorities: lowest %d, highest %d\n", prio_lo, prio_hi); fflush(stdout);
    unsigned* out; (void)hipMalloc(&out, 1 << 20); (void)hipMemset(out, 0, 1 << 20);
    const int NB = 3; std::vector<hipStream_t> bst(NB); std::vector<uint4*> buf(NB); const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    unsigned* sink; (void)hipMalloc(&sink, 64);
    for (int b = 0; b < NB; ++b) { (void)hipStreamCreateWithFlags(&bst[b], hipStreamNonBlocking); (void)hipMalloc(&buf[b], bytes); (void)hipMemset(buf[b], 1, bytes); }
    // the same background on streams that may only use three quarters of the CUs (every fourth block of eight mask bits off: a quarter of every XCD whichever way the bits are dealt out)
    std::vector<hipStream_t> mst(NB); uint32_t mask[8]; for (int w = 0; w < 8; ++w) mask[w] = 0x00FFFFFFu;
    for (int b = 0; b < NB; ++b) { const hipError_t e = hipExtStreamCreateWithCUMask(&mst[b], 8, mask); if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 1; } }
    (void)hipDeviceSynchronize();
    const char* names[] = { "alone", "beside arithmetic-only kernels (3 streams, device full)", "beside streaming WRITES (3 streams, 1 GB each per kernel)", "beside streaming READS (3 streams)",
                            "beside short kernels that write 4 MB each (3 streams)", "beside arithmetic-only kernels of 64 workgroups (3 streams: a quarter of the CUs)",
                            "beside arithmetic-only kernels (device full) on streams MASKED to 3/4 of the CUs", "beside streaming WRITES on streams MASKED to 3/4 of the CUs",
                            "on a HIGH-PRIORITY stream beside arithmetic-only kernels (device full)", "on a HIGH-PRIORITY stream beside streaming WRITES" };
    for (int mode : {0, 8, 9, 7, 2, 3, 4, 5}) {
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        if (mode > 0) for (int b = 0; b < NB; ++b) th.emplace_back([&, b]() {
            (void)hipSetDevice(0);
            unsigned k = 0;
            while (!stop.load()) {
                for (int r = 0; r < 4; ++r) {
                    if (mode == 1 || mode == 8) hipLaunchKernelGGL(bg_alu, dim3(256 * 64), dim3(256), 0, bst[b], sink, 3000);
                    else if (mode == 2 || mode == 9) hipLaunchKernelGGL(bg_write, dim3(256 * 8), dim3(256), 0, bst[b], buf[b], n4, ++k);
                    else if (mode == 3) hipLaunchKernelGGL(bg_read, dim3(256 * 8), dim3(256), 0, bst[b], (const uint4*)buf[b], n4, sink);
                    else if (mode == 4) hipLaunchKernelGGL(bg_write, dim3(256), dim3(256), 0, bst[b], buf[b], (size_t)(4 << 20) / 16, ++k);
                    else if (mode == 5) hipLaunchKernelGGL(bg_alu, dim3(64), dim3(256), 0, bst[b], sink, 100000);
                    else if (mode == 6) hipLaunchKernelGGL(bg_alu, dim3(256 * 64), dim3(256), 0, mst[b], sink, 3000);
                    else hipLaunchKernelGGL(bg_write, dim3(256 * 8), dim3(256), 0, mst[b], buf[b], n4, ++k);
                }
                (void)hipStreamSynchronize((mode == 6 || mode == 7) ? mst[b] : bst[b]);
            }
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        for (int blocks : {13, 271}) {
            hipStream_t cs = mode >= 8 ? chain_hi : chain_st;
            const int iters = (mode == 1 || mode == 8 || mode == 6) ? 40 : 600;
            for (int i = 0; i < ((mode == 1 || mode == 8 || mode == 6) ? 2 : 50); ++i) { hipLaunchKernelGGL(step_like, dim3(blocks), dim3(512), 0, cs, out); hipLaunchKernelGGL(control_like, dim3(1), dim3(1024), 0, cs, out, blocks * 64); }
            (void)hipStreamSynchronize(cs);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; ++i) { hipLaunchKernelGGL(step_like, dim3(blocks), dim3(512), 0, cs, out); hipLaunchKernelGGL(control_like, dim3(1), dim3(1024), 0, cs, out, blocks * 64); }
            (void)hipStreamSynchronize(cs);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            printf("chain of dependent pairs (%3d + 1 workgroups) %-80s %7.2f us per pair\n", blocks, names[mode], us); fflush(stdout);
        }
        stop.store(true);
        for (auto& t : th) t.join();
        (void)hipDeviceSynchronize();
    }
    return 0;
}
