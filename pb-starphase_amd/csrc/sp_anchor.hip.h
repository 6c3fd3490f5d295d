// sp_anchor.hip.h -- the two halves of the k-mer vote anchor that every way of producing the votes shares (sp_anchor_kernel, sp_device.hip;
// k2_anchor_dict_kernel, sp_hla.hip): adding a wave's runs of equal single votes, and reading the peaks out of the histogram.
// Bins are u16 counters packed two per dword in LDS; bin = b_pos - a_pos + len(A).
#pragma once
#include "sp_internal.h"
#include "sp_wfa.hip.h"

// a read that crosses the gene puts thousands of votes on one diagonal, and neighbouring lanes hold neighbouring k-mers of it: lanes
// whose single vote goes to the bin of the lane before them hand it to the first lane of their run, which adds the run's count once
// (64 atomics on one LDS address would be carried out one after the other).  bin1 < 0: this lane has no single vote.
__device__ __forceinline__ void sp_anchor_vote_run(uint32_t* lds, int bin1) {
    const int prev = __shfl_up(bin1, 1);
    const int lane = threadIdx.x & 63;
    const bool follows = bin1 >= 0 && lane > 0 && prev == bin1;
    const unsigned long long F = __ballot(follows);
    if (bin1 >= 0 && !follows) {
        const unsigned long long rest = lane == 63 ? 0ull : (F >> (lane + 1));
        const uint32_t count = 1u + (uint32_t)__builtin_ctzll(~rest);
        atomicAdd(&lds[bin1 >> 1], (bin1 & 1) ? count << 16 : count);
    }
}

// clears the first nb32 dwords of the histogram, 16 bytes per store
template <int THREADS>
__device__ __forceinline__ void sp_anchor_clear(uint32_t* lds, int nb32) {
    const int nq = nb32 >> 2;
    for (int i = threadIdx.x; i < nq; i += THREADS) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = 4 * nq + threadIdx.x; i < nb32; i += THREADS) lds[i] = 0;
}

// top-K peaks of the finished histogram (every thread of the workgroup calls it after the barrier that ends the voting): argmax votes
// (ties -> smallest diagonal), the reported diagonal is the midpoint of the outermost diagonals within +-SP_PEAK_SPREAD of the peak
// that still hold >= max(2, peak/8) votes (a long indel splits the votes over two diagonals), then every bin within
// +-SP_PEAK_SUPPRESS of the peak is cleared before the next one is taken.  red: THREADS / 64 words, spread: 2 ints, both in LDS.
template <int THREADS>
__device__ __forceinline__ void sp_anchor_peaks(uint32_t* lds, int nbins, int m, int topk, uint64_t p, int32_t* __restrict__ diag_out, int32_t* __restrict__ votes_out,
                                                unsigned long long* red, int* spread) {
    const int tid = threadIdx.x;
    const int nb32 = (nbins + 1) >> 1;
    for (int round = 0; round < topk; ++round) {
        // every lane keeps the first of its heaviest bins (it visits its bins in ascending order); eight bins per LDS read
        int bv = 0, bb = 0;
        auto take = [&](uint32_t w, int bin0) {
            const int v0 = (int)(w & 0xFFFFu), v1 = (int)(w >> 16);
            if (v0 > bv) { bv = v0; bb = bin0; }
            if (v1 > bv && bin0 + 1 < nbins) { bv = v1; bb = bin0 + 1; }
        };
        const int nq = nb32 >> 2;
        for (int i = tid; i < nq; i += THREADS) {
            const uint4 q = reinterpret_cast<const uint4*>(lds)[i];
            take(q.x, 8 * i); take(q.y, 8 * i + 2); take(q.z, 8 * i + 4); take(q.w, 8 * i + 6);
        }
        for (int i = 4 * nq + tid; i < nb32; i += THREADS) take(lds[i], 2 * i);
        // heaviest count of the wave, then the smallest bin that holds it: two DPP reductions (no LDS traffic)
        const int vmax = spw::wave_max(bv);
        const int kb = spw::wave_max(bv == vmax ? 0x7FFFFFFF - bb : -1);
        unsigned long long best = ((unsigned long long)(uint32_t)vmax << 32) | (uint32_t)kb;
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        best = red[0];
        for (int w = 1; w < THREADS / 64; ++w) best = red[w] > best ? red[w] : best;
        const int v = (int)(best >> 32);
        const int bin = 0x7FFFFFFF - (int)(best & 0xFFFFFFFFu);
        if (tid == 0) { spread[0] = bin; spread[1] = bin; }
        __syncthreads();
        if (v > 0 && tid <= 2 * SP_PEAK_SPREAD) {
            const int b2 = bin - SP_PEAK_SPREAD + tid;
            const int thr = v / 8 > 2 ? v / 8 : 2;
            if (b2 >= 0 && b2 < nbins && (int)((lds[b2 >> 1] >> ((b2 & 1) << 4)) & 0xFFFFu) >= thr) { atomicMin(&spread[0], b2); atomicMax(&spread[1], b2); }
        }
        __syncthreads();
        if (tid == 0) { diag_out[p * topk + round] = v > 0 ? ((spread[0] + spread[1]) >> 1) - m : 0; votes_out[p * topk + round] = v; }
        __syncthreads();
        if (v == 0) {                                   // nothing left: remaining slots are empty
            if (tid == 0) for (int k2 = round + 1; k2 < topk; ++k2) { diag_out[p * topk + k2] = 0; votes_out[p * topk + k2] = 0; }
            break;
        }
        if (round + 1 < topk) {
            int lo = bin - SP_PEAK_SUPPRESS, hi = bin + SP_PEAK_SUPPRESS;
            if (lo < 0) lo = 0;
            if (hi > nbins - 1) hi = nbins - 1;
            // bins are packed two per dword: clear them one lane per bin with a masked atomic AND
            for (int b2 = lo + tid; b2 <= hi; b2 += THREADS) atomicAnd(&lds[b2 >> 1], (b2 & 1) ? 0x0000FFFFu : 0xFFFF0000u);
            __syncthreads();
        }
    }
}
