// sp_microbench.hip -- measurement aids behind the ABI (bench.py's roofline peaks, measured in the same run as the kernels):
// the sustained integer VALU issue rate of the chip and its streaming HBM copy rate.  Not on the product path.
#include "sp_internal.h"
#include <cstring>
#include <cstdlib>

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

// streaming copy: every workgroup owns contiguous 16 KiB tiles (256 lanes x 16 B x 4 loads in flight per lane before the first store), non-temporal
// both ways (nothing is read twice: the lines need not stay in L2 / MALL).  One load per lane and iteration with a grid-wide stride reached 4.6 TB/s;
// VARIANT selects what bench.py's "hbm_copy" runs (profiles/r04/hbm_copy_variants.txt)
typedef unsigned mb_u4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void mb_copy_kernel(const mb_u4* __restrict__ src, mb_u4* __restrict__ dst, size_t n) {
    const size_t tile = (size_t)256 * UNROLL, n_tiles = n / tile;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const mb_u4* s = src + t * tile + threadIdx.x; mb_u4* d = dst + t * tile + threadIdx.x;
        mb_u4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(s + u * 256) : s[u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + u * 256); else d[u * 256] = v[u]; }
    }
    for (size_t i = n_tiles * tile + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
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
        mb_u4* src = (mb_u4*)sp_pool(ctx, "mb_src", bytes); mb_u4* dst = (mb_u4*)sp_pool(ctx, "mb_dst", bytes);
        if (!src || !dst) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_microbench");
        (void)hipMemsetAsync(src, 1, bytes, ctx->stream);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, ctx->stream);
            const char* var = std::getenv("SP_MB_COPY_VARIANT");
            const int v = var ? std::atoi(var) : 3;
            const dim3 grid(ctx->num_cus * (v >= 10 ? 32 : 16));
            switch (v % 10) {
                case 0: hipLaunchKernelGGL((mb_copy_kernel<1, false>), grid, dim3(256), 0, ctx->stream, src, dst, bytes / 16); break;
                case 1: hipLaunchKernelGGL((mb_copy_kernel<4, false>), grid, dim3(256), 0, ctx->stream, src, dst, bytes / 16); break;
                case 2: hipLaunchKernelGGL((mb_copy_kernel<1, true>), grid, dim3(256), 0, ctx->stream, src, dst, bytes / 16); break;
                case 3: hipLaunchKernelGGL((mb_copy_kernel<4, true>), grid, dim3(256), 0, ctx->stream, src, dst, bytes / 16); break;
                default: hipLaunchKernelGGL((mb_copy_kernel<8, true>), grid, dim3(256), 0, ctx->stream, src, dst, bytes / 16); break;
            }
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
