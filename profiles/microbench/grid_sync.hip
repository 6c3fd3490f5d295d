// grid_sync.hip -- what a grid-wide barrier per consensus column would cost on MI355X (cooperative launch), against the
// ~10 us of one dependent launch per column.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/grid_sync profiles/microbench/grid_sync.hip && timeout 60 /tmp/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <chrono>
namespace cg = cooperative_groups;

// every column: a handful of atomics per workgroup, the barrier, every wave reads the counters back
__global__ void k_columns(unsigned* __restrict__ votes, int columns, unsigned* __restrict__ sink) {
    cg::grid_group grid = cg::this_grid();
    unsigned acc = 0;
    for (int t = 0; t < columns; ++t) {
        if (threadIdx.x < 4) atomicAdd(&votes[(size_t)(t & 1023) * 256 + (blockIdx.x & 7) * 32 + threadIdx.x], 12u);
        grid.sync();
        if ((threadIdx.x & 63) < 16) acc += votes[(size_t)(t & 1023) * 256 + (threadIdx.x & 7) * 32 + ((threadIdx.x >> 3) & 1)];
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// the same barrier by hand: one arrival counter per column, a wave per workgroup spins on it
__global__ void k_columns_manual(unsigned* __restrict__ votes, unsigned* __restrict__ arrive, int columns, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (int t = 0; t < columns; ++t) {
        if (threadIdx.x < 4) atomicAdd(&votes[(size_t)(t & 1023) * 256 + (blockIdx.x & 7) * 32 + threadIdx.x], 12u);
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(&arrive[t], 1u);
            while (__hip_atomic_load(&arrive[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if ((threadIdx.x & 63) < 16) acc += __hip_atomic_load(&votes[(size_t)(t & 1023) * 256 + (threadIdx.x & 7) * 32 + ((threadIdx.x >> 3) & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main() {
    unsigned *votes, *sink, *arrive;
    hipMalloc(&votes, 1024 * 256 * 4); hipMalloc(&sink, 64); hipMalloc(&arrive, 8192 * 4);
    hipMemset(votes, 0, 1024 * 256 * 4);
    int columns = 2000;
    for (int threads : {1024, 512}) {
        // the hand-made barrier spins: every workgroup of the launch has to be resident, so the grid is sized by the
        // smaller of the two kernels' own occupancies (a grid sized for k_columns alone could leave k_columns_manual waiting
        // for workgroups that can never start)
        int per_cu = 0, per_cu_manual = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_columns, threads, 0);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_manual, k_columns_manual, threads, 0);
        if (per_cu_manual < per_cu) per_cu = per_cu_manual;
        hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
        for (int blocks : {64, 256, per_cu * prop.multiProcessorCount}) {
            void* args[] = { &votes, &columns, &sink };
            hipError_t e = hipLaunchCooperativeKernel((const void*)k_columns, dim3(blocks), dim3(threads), args, 0, 0);
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            e = hipLaunchCooperativeKernel((const void*)k_columns, dim3(blocks), dim3(threads), args, 0, 0);
            hipError_t e2 = hipDeviceSynchronize();
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / columns;
            printf("cg grid.sync   %4d threads x %4d workgroups (%d per CU max): %6.2f us per column  (%s / %s)\n", threads, blocks, per_cu, us, hipGetErrorString(e), hipGetErrorString(e2));
            hipMemset(arrive, 0, 8192 * 4);
            hipLaunchKernelGGL(k_columns_manual, dim3(blocks), dim3(threads), 0, 0, votes, arrive, columns, sink);
            hipDeviceSynchronize();
            hipMemset(arrive, 0, 8192 * 4);
            t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_columns_manual, dim3(blocks), dim3(threads), 0, 0, votes, arrive, columns, sink);
            e2 = hipDeviceSynchronize();
            us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / columns;
            printf("manual barrier %4d threads x %4d workgroups: %6.2f us per column  (%s)\n", threads, blocks, us, hipGetErrorString(e2));
        }
    }
    return 0;
}
