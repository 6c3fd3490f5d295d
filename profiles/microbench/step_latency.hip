// step_latency.hip -- what a "one dependent launch per consensus base" loop costs on MI355X, bottom up.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/step_latency profiles/microbench/step_latency.hip && /tmp/step_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#include <vector>

__global__ void k_empty(int* p, int t) { if (t < 0) p[0] = 1; }

// every wave reads and rewrites a 256-byte state row
__global__ void k_state(int* __restrict__ H, int n, int t) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    int h = H[(size_t)r * 64 + lane];
    H[(size_t)r * 64 + lane] = h + (t & 1);
}

// + a dependent second load (the read word at the tip) and 80 counter words read by everybody
__global__ void k_chain(int* __restrict__ H, const uint32_t* __restrict__ words, const uint32_t* __restrict__ votes, int n, int t) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    uint32_t acc = 0;
    for (int i = 0; i < 80; ++i) acc += votes[(size_t)(t & 1023) * 128 + i];
    int h = H[(size_t)r * 64 + lane];
    uint32_t w = words[(size_t)r * 256 + ((h + (int)acc) & 255)];
    H[(size_t)r * 64 + lane] = h + (int)(w & 1) + (t & 1);
}

// + LDS vote accumulation, two barriers and a handful of global atomics per block
__global__ void k_votes(int* __restrict__ H, const uint32_t* __restrict__ words, uint32_t* __restrict__ votes, int n, int t) {
    __shared__ uint32_t lv[16];
    const int lane = threadIdx.x & 63, r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (threadIdx.x < 16) lv[threadIdx.x] = 0;
    uint32_t acc = 0;
    for (int i = 0; i < 80; ++i) acc += votes[(size_t)(t & 1023) * 128 + i];
    __syncthreads();
    if (r < n) {
        int h = H[(size_t)r * 64 + lane];
        uint32_t w = words[(size_t)r * 256 + ((h + (int)(acc & 1)) & 255)];
        H[(size_t)r * 64 + lane] = h + (int)(w & 1) + (t & 1);
        if (lane == 0) atomicAdd(&lv[w & 3], 12u);
    }
    __syncthreads();
    if (threadIdx.x < 5 && lv[threadIdx.x]) atomicAdd(&votes[(size_t)((t + 1) & 1023) * 128 + (blockIdx.x & 7) * 8 + threadIdx.x], lv[threadIdx.x]);
}

template <class F> static double loop(const char* name, int launches, F f) {
    for (int t = 0; t < 64; ++t) f(t);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < launches; ++t) f(t);
    hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
    printf("%-34s %7.2f us per launch\n", name, us);
    return us;
}

int main() {
    const int launches = 4000;
    int* H; uint32_t* words; uint32_t* votes; int* p;
    const int nmax = 16384;
    hipMalloc(&H, (size_t)nmax * 64 * 4); hipMalloc(&words, (size_t)nmax * 256 * 4); hipMalloc(&votes, 1024 * 128 * 4); hipMalloc(&p, 64);
    hipMemset(H, 0, (size_t)nmax * 64 * 4); hipMemset(words, 0x5a, (size_t)nmax * 256 * 4); hipMemset(votes, 0, 1024 * 128 * 4);
    for (int n : {64, 2048, 10240}) {
        for (int waves : {4, 8, 16}) {
            const dim3 grid((n + waves - 1) / waves), block(waves * 64);
            printf("-- %d reads, %d waves per workgroup (%u workgroups)\n", n, waves, grid.x);
            loop("empty", launches, [&](int t) { hipLaunchKernelGGL(k_empty, grid, block, 0, 0, p, t); });
            loop("state row read+write", launches, [&](int t) { hipLaunchKernelGGL(k_state, grid, block, 0, 0, H, n, t); });
            loop("+ counters + dependent read", launches, [&](int t) { hipLaunchKernelGGL(k_chain, grid, block, 0, 0, H, words, votes, n, t); });
            loop("+ LDS votes, barriers, atomics", launches, [&](int t) { hipLaunchKernelGGL(k_votes, grid, block, 0, 0, H, words, votes, n, t); });
        }
    }
    return 0;
}
