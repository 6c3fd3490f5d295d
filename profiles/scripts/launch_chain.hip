// Cost of a chain of dependent small launches: plain stream launches against the same kernels captured in a graph (N nodes per graph launch).
// build: hipcc -O3 --offload-arch=gfx950 profiles/scripts/launch_chain.hip -o build/exp/launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { int v[200]; };                                   // an 800-byte kernel argument like the consensus batch descriptor
__global__ void __launch_bounds__(512) step_like(Big b, unsigned* out) {
    __shared__ unsigned s[6000];
    s[threadIdx.x] = threadIdx.x + b.v[threadIdx.x % 200];
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = s[threadIdx.x * 3] + out[(blockIdx.x * 64 + threadIdx.x + 7) % (gridDim.x * 64)];
}
__global__ void __launch_bounds__(1024) control_like(Big b, unsigned* out, int n) {
    __shared__ unsigned s[20000];
    for (int i = threadIdx.x; i < 20000; i += blockDim.x) s[i] = 0;
    __syncthreads();
    unsigned a = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += out[i];
    atomicAdd(&s[threadIdx.x % 64], a);
    __syncthreads();
    if (threadIdx.x < 64) out[threadIdx.x] = s[threadIdx.x] + b.v[3];
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned* out; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20);
    Big b{}; const int iters = 3000;
    for (int blocks : {13, 105}) {
        auto run_plain = [&](int n) { for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(step_like, dim3(blocks), dim3(512), 0, st, b, out); hipLaunchKernelGGL(control_like, dim3(1), dim3(1024), 0, st, b, out, blocks * 64); } };
        run_plain(50); hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        run_plain(iters); hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("blocks %3d plain stream launches : %.2f us per pair\n", blocks, us / iters);
        for (int per : {4, 8, 32}) {
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            run_plain(per);
            hipStreamEndCapture(st, &g);
            auto ti = std::chrono::steady_clock::now();
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            double inst = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ti).count();
            for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters / per; ++i) hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("blocks %3d graph of %2d pairs        : %.2f us per pair (instantiate %.0f us)\n", blocks, per, us / (iters / per * per), inst);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
