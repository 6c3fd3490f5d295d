// sp_microbench.hip -- measurement aids behind the ABI (bench.py's roofline peaks, measured in the same run as the kernels):
// the sustained integer VALU issue rate of the chip and its streaming HBM copy rate.  Not on the product path.
#include "sp_internal.h"
#include <cstring>

namespace {

constexpr int MB_ITER = 4096;

// eight independent v_add_u32 chains per lane: 8 VALU wave-instructions per iteration and wavefront
__global__ __launch_bounds__(256) void mb_valu_kernel(unsigned* __restrict__ out, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x55u, a3 = a0 + 7u, a4 = a1 + 5u, a5 = a2 + 9u, a6 = a3 ^ a1, a7 = a0 + 11u;
    for (int i = 0; i < MB_ITER; ++i) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// the 16-base compare of the WFA cell (two LDS word pairs, two v_alignbit, xor / or / and, v_ffbl, min): 17 VALU + 2 LDS per iteration
__global__ __launch_bounds__(256) void mb_match_kernel(unsigned* __restrict__ out, unsigned seed) {
    __shared__ unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    unsigned p = threadIdx.x * 7u + seed, acc = 0;
    for (int i = 0; i < MB_ITER; ++i) {
        const unsigned wa = (p >> 5) & 1022u, wb = ((p + 77u) >> 5) & 1022u;
        const unsigned x = __builtin_amdgcn_alignbit(lds[wa + 1], lds[wa], p) ^ __builtin_amdgcn_alignbit(lds[1024 + wb + 1], lds[1024 + wb], p + 77u);
        const unsigned mm = (x | (x >> 1)) & 0x55555555u;
        const unsigned f = mm ? (unsigned)__builtin_ctz(mm) : 32u;
        acc += f; p += (f < 32u ? f : 32u) + 2u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc ^ p;
}

__global__ __launch_bounds__(256) void mb_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

} // namespace

// what: "valu_int" -> wave-instructions / s of v_add_u32 (the plain integer VALU rate of the whole chip)
//       "match16"  -> wave-instructions / s of the WFA cell's 16-base compare mix (VALU instructions only; its LDS reads ride along)
//       "hbm_copy" -> bytes / s moved (read + written) by a streaming copy of 2 x 1 GiB
extern "C" int32_t sp_microbench(sp_ctx* ctx, const char* what, double* rate) {
    if (!ctx || !what || !rate) return SP_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    *rate = 0.0;
    hipEvent_t e0, e1;
    SP_HIP_CHECK(ctx, hipEventCreate(&e0)); SP_HIP_CHECK(ctx, hipEventCreate(&e1));
    float ms = 0.f; double units = 0.0;
    if (std::strcmp(what, "valu_int") == 0 || std::strcmp(what, "match16") == 0) {
        const bool plain = what[0] == 'v';
        const int blocks = ctx->num_cus * 32;                      // 8 waves per SIMD worth of workgroups, several rounds
        unsigned* d = (unsigned*)sp_pool(ctx, "mb_out", (size_t)blocks * 256 * 4);
        if (!d) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_microbench");
        for (int rep = 0; rep < 2; ++rep) {                        // the first launch warms up
            (void)hipEventRecord(e0, ctx->stream);
            if (plain) hipLaunchKernelGGL(mb_valu_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d, (unsigned)rep);
            else hipLaunchKernelGGL(mb_match_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d, (unsigned)rep);
            (void)hipEventRecord(e1, ctx->stream);
            SP_HIP_CHECK(ctx, hipEventSynchronize(e1));
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        units = (double)blocks * 4.0 * MB_ITER * (plain ? 8.0 : 17.0);
    } else if (std::strcmp(what, "hbm_copy") == 0) {
        const size_t bytes = (size_t)1 << 30;
        uint4* src = (uint4*)sp_pool(ctx, "mb_src", bytes); uint4* dst = (uint4*)sp_pool(ctx, "mb_dst", bytes);
        if (!src || !dst) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_microbench");
        (void)hipMemsetAsync(src, 1, bytes, ctx->stream);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, ctx->stream);
            hipLaunchKernelGGL(mb_copy_kernel, dim3(ctx->num_cus * 16), dim3(256), 0, ctx->stream, src, dst, bytes / 16);
            (void)hipEventRecord(e1, ctx->stream);
            SP_HIP_CHECK(ctx, hipEventSynchronize(e1));
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        units = 2.0 * (double)bytes;
    } else return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_microbench: unknown measurement");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ms > 0.f) *rate = units / ((double)ms * 1e-3);
    return SP_OK;
}
