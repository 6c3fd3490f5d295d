// Chains of dependent small launches on several streams at once (one host thread each): how much does a chain slow down beside seven others, as
// plain stream launches and as graph launches of 8 pairs?   build: hipcc -O3 --offload-arch=gfx950 -pthread profiles/scripts/launch_chain_mt.hip -o build/exp/launch_chain_mt
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Big { int v[200]; };
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
static double chain(hipStream_t st, unsigned* out, int iters, int blocks, hipGraphExec_t ge, int per) {
    Big b{};
    auto t0 = std::chrono::steady_clock::now();
    if (ge) for (int i = 0; i < iters / per; ++i) (void)hipGraphLaunch(ge, st);
    else for (int i = 0; i < iters; ++i) { hipLaunchKernelGGL(step_like, dim3(blocks), dim3(512), 0, st, b, out); hipLaunchKernelGGL(control_like, dim3(1), dim3(1024), 0, st, b, out, blocks * 64); }
    (void)hipStreamSynchronize(st);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
}
int main() {
    const int iters = 2000, blocks = 13, per = 8, T = 8;
    std::vector<hipStream_t> st(T); std::vector<unsigned*> out(T); std::vector<hipGraphExec_t> ge(T);
    for (int t = 0; t < T; ++t) {
        (void)hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking); (void)hipMalloc(&out[t], 1 << 20); (void)hipMemset(out[t], 0, 1 << 20);
        hipGraph_t g; Big b{};
        (void)hipStreamBeginCapture(st[t], hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < per; ++i) { hipLaunchKernelGGL(step_like, dim3(blocks), dim3(512), 0, st[t], b, out[t]); hipLaunchKernelGGL(control_like, dim3(1), dim3(1024), 0, st[t], b, out[t], blocks * 64); }
        (void)hipStreamEndCapture(st[t], &g); (void)hipGraphInstantiate(&ge[t], g, nullptr, nullptr, 0);
        chain(st[t], out[t], 50, blocks, nullptr, per); chain(st[t], out[t], 48, blocks, ge[t], per);
    }
    for (int mode = 0; mode < 2; ++mode)
        for (int n : {1, 2, 4, 8}) {
            std::vector<double> us(n); std::vector<std::thread> th;
            for (int t = 0; t < n; ++t) th.emplace_back([&, t]() { us[t] = chain(st[t], out[t], iters, blocks, mode ? ge[t] : nullptr, per); });
            for (auto& x : th) x.join();
            double worst = 0; for (double u : us) worst = u > worst ? u : worst;
            printf("%s, %d chain(s) at once: %.2f us per dependent pair (slowest chain)\n", mode ? "graphs of 8 pairs   " : "plain stream launches", n, worst);
        }
    return 0;
}
