// Microbenchmark (measurement aid, not product): sustained issue rate of the integer VALU / SALU / DPP / LDS
// instructions the WFA cell is made of, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 valu_int_rate.hip -o valu_int_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x55u, a3 = a0 + 7u, a4 = a1 + 5u, a5 = a2 + 9u, a6 = a3 ^ a1, a7 = a0 + 11u;
    __shared__ unsigned lds[1024];
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1; lds[threadIdx.x + 512] = a2; lds[threadIdx.x + 768] = a3;
    __syncthreads();
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 0) {        // v_add_u32 (8 independent chains)
            a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
        } else if (MODE == 1) { // v_xor / v_and / shifts
            a0 ^= a1 >> 1; a1 = (a1 | a2) & 0x55555555u; a2 ^= a3 << 2; a3 = (a3 ^ a4) >> 1; a4 ^= a5; a5 |= a6 >> 3; a6 &= a7 | 1u; a7 ^= a0 << 1;
        } else if (MODE == 2) { // v_alignbit
            a0 = __builtin_amdgcn_alignbit(a1, a0, a2 & 31); a1 = __builtin_amdgcn_alignbit(a2, a1, a3 & 31);
            a2 = __builtin_amdgcn_alignbit(a3, a2, a4 & 31); a3 = __builtin_amdgcn_alignbit(a4, a3, a5 & 31);
            a4 = __builtin_amdgcn_alignbit(a5, a4, a6 & 31); a5 = __builtin_amdgcn_alignbit(a6, a5, a7 & 31);
            a6 = __builtin_amdgcn_alignbit(a7, a6, a0 & 31); a7 = __builtin_amdgcn_alignbit(a0, a7, a1 & 31);
        } else if (MODE == 3) { // v_cmp + v_cndmask pairs
            a0 = a0 > a1 ? a2 : a3; a1 = a1 > a2 ? a3 : a4; a2 = a2 > a3 ? a4 : a5; a3 = a3 > a4 ? a5 : a6;
            a4 = a4 > a5 ? a6 : a7; a5 = a5 > a6 ? a7 : a0; a6 = a6 > a7 ? a0 : a1; a7 = a7 > a0 ? a1 : a2;
        } else if (MODE == 4) { // DPP wave_shr moves
            a0 = __builtin_amdgcn_update_dpp(a0, a1, 0x138, 0xf, 0xf, false); a1 = __builtin_amdgcn_update_dpp(a1, a2, 0x130, 0xf, 0xf, false);
            a2 = __builtin_amdgcn_update_dpp(a2, a3, 0x138, 0xf, 0xf, false); a3 = __builtin_amdgcn_update_dpp(a3, a4, 0x130, 0xf, 0xf, false);
            a4 = __builtin_amdgcn_update_dpp(a4, a5, 0x138, 0xf, 0xf, false); a5 = __builtin_amdgcn_update_dpp(a5, a6, 0x130, 0xf, 0xf, false);
            a6 = __builtin_amdgcn_update_dpp(a6, a7, 0x138, 0xf, 0xf, false); a7 = __builtin_amdgcn_update_dpp(a7, a0, 0x130, 0xf, 0xf, false);
        } else if (MODE == 5) { // v_fma_f32 reference (floats)
            float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
            float f4 = __uint_as_float(a4), f5 = __uint_as_float(a5), f6 = __uint_as_float(a6), f7 = __uint_as_float(a7);
            f0 = __builtin_fmaf(f0, f1, f2); f1 = __builtin_fmaf(f1, f2, f3); f2 = __builtin_fmaf(f2, f3, f4); f3 = __builtin_fmaf(f3, f4, f5);
            f4 = __builtin_fmaf(f4, f5, f6); f5 = __builtin_fmaf(f5, f6, f7); f6 = __builtin_fmaf(f6, f7, f0); f7 = __builtin_fmaf(f7, f0, f1);
            a0 = __float_as_uint(f0); a1 = __float_as_uint(f1); a2 = __float_as_uint(f2); a3 = __float_as_uint(f3);
            a4 = __float_as_uint(f4); a5 = __float_as_uint(f5); a6 = __float_as_uint(f6); a7 = __float_as_uint(f7);
        } else if (MODE == 6) { // ds_read2_b32 + alignbit (the match16 core)
            unsigned p = (a0 & 1023u) >> 1;
            unsigned lo = lds[p & 1022u], hi = lds[(p & 1022u) + 1];
            a0 = __builtin_amdgcn_alignbit(hi, lo, a1 & 31) + a0;
            a1 += 3;
        } else if (MODE == 7) { // ffbl + min + bitop mix (rest of match16)
            unsigned x = a0 ^ a1; unsigned mm = (x | (x >> 1)) & 0x55555555u;
            a0 += mm ? (__builtin_ffs((int)mm) - 1) >> 1 : 16; a1 = a1 * 1u + a2; a2 ^= a0;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int MODE> void run(const char* name, int ops_per_iter, unsigned* d_out, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * ITER * ops_per_iter;
    // per SIMD per cycle at 2.4 GHz nominal, 1024 SIMDs
    printf("%-28s %8.3f ms  %.3e wave-instr/s  = %.3f wave-instr/cycle/SIMD @2.4GHz (blocks=%d)\n", name, ms, wave_instr / (ms * 1e-3),
           wave_instr / (ms * 1e-3) / 2.4e9 / 1024.0, blocks);
}

int main() {
    unsigned* d_out; hipMalloc(&d_out, 256 * 8192 * 4);
    for (int blocks : {256 * 2, 256 * 8}) {
        run<0>("v_add_u32", 8, d_out, blocks);
        run<1>("xor/and/or/shift mix", 13, d_out, blocks);
        run<2>("v_alignbit (+and)", 16, d_out, blocks);
        run<3>("v_cmp+v_cndmask", 16, d_out, blocks);
        run<4>("v_mov_dpp wave_shr/shl", 8, d_out, blocks);
        run<5>("v_fma_f32", 8, d_out, blocks);
        run<6>("ds_read2+alignbit chain", 6, d_out, blocks);
        run<7>("xor/ffbl/min mix", 12, d_out, blocks);
    }
    return 0;
}
