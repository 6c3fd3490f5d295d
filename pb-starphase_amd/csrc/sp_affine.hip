// sp_affine.hip -- the two-piece affine re-score of an alignment the library found: the numbers the reference reports.
//
// Every (nm, start, end) of the reference is minimap2's (`standard_hifi_aligner`, src/util/mapping.rs:8-14: map-hifi, match 1 -- 5 in score_read,
// src/hla/caller.rs:1370-1379 --, mismatch 4, gaps min(6 + 2 l, 26 + l), ambiguous bases -1): an alignment through the chain's seeds, global between
// them and extended from the outermost ones to the best-scoring cell, i.e. the best LOCAL alignment through its seeds.  The library's own cell (anchor +
// unit-cost wavefront, sp_wfa.hip.h) decides which pairs align and on which diagonal; this kernel then re-scores a pair the reference's way: the banded
// Smith-Waterman optimum under those scores on the 64 or 256 diagonals around the cell's diagonal, with the forward decisions and end rules of
// oracle/affine.c (the CPU statement this kernel is bit-exact against; oracle/mm2.c is the minimap2 restatement it is measured against:
// tests/test_oracle_affine.py -- identical (nm, spans) on every audited K1 / K2 pair, 98.7 % of the K3 hits inside the 5 % filter).
//
// One wavefront per pair, lane l holds DPL consecutive diagonals (1: 64 diagonals, 4: 256), one target row per step.  Per cell three states come from the
// row before (H on the same diagonal, H / E / E2 on the next one: one DPP move each) and the two insertion states run ALONG the row: F(j) = max over
// j' < j of H(j') - q - (j - j') e is a max-plus prefix scan over the lanes -- the gfx9 DPP scan (row_shr 1, 2, 4, 8, row_bcast 15 / 31) on a packed
// key (score + position * e, ties to the nearest opening, as the sequential recurrence decides them) with the path's counters riding along.  Every state
// carries the mismatch + gap + ambiguous bases of its path and the cell it began in: the result needs no traceback and no memory beyond the two packed
// sequences in LDS.
#include "sp_internal.h"
#include "sp_wfa.hip.h"
#include <mutex>
#include <set>
#include <string>

namespace {

constexpr int AF_NEG = -(1 << 28);
constexpr uint32_t AF_BIAS = 1u << 22;

struct AfState { int s; uint32_t m0, m1; };                 // score; start cell (i << 16 | j); mismatch + gap + ambiguous bases of the path
struct AfKey { uint32_t k, m0, m1; };                       // scan element: (score + pos * e + bias) << 8 | pos, and the counters of that cell's path

__device__ __forceinline__ AfState af_none() { AfState x; x.s = AF_NEG; x.m0 = 0; x.m1 = 0; return x; }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ AfKey af_dpp(const AfKey& v) {
    AfKey r;
    r.k = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.k, CTRL, ROW_MASK, 0xf, false);
    r.m0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.m0, CTRL, ROW_MASK, 0xf, false);
    r.m1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.m1, CTRL, ROW_MASK, 0xf, false);
    return r;
}
__device__ __forceinline__ AfKey af_max(const AfKey& a, const AfKey& b) { return b.k > a.k ? b : a; }
// inclusive prefix maximum over the 64 lanes (keys are unique per position, 0 = nothing)
__device__ __forceinline__ AfKey af_scan(AfKey v) {
    v = af_max(v, af_dpp<0x111, 0xf>(v));                   // row_shr:1
    v = af_max(v, af_dpp<0x112, 0xf>(v));                   // row_shr:2
    v = af_max(v, af_dpp<0x114, 0xf>(v));                   // row_shr:4
    v = af_max(v, af_dpp<0x118, 0xf>(v));                   // row_shr:8
    v = af_max(v, af_dpp<0x142, 0xa>(v));                   // row_bcast:15 into rows 1 and 3
    v = af_max(v, af_dpp<0x143, 0xc>(v));                   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ AfKey af_from_lower(const AfKey& v) {          // lane l - 1's value, nothing for lane 0
    AfKey r;
    r.k = (uint32_t)spw::from_lower((int)v.k, 0); r.m0 = (uint32_t)spw::from_lower((int)v.m0, 0); r.m1 = (uint32_t)spw::from_lower((int)v.m1, 0);
    return r;
}
__device__ __forceinline__ AfState af_from_upper(const AfState& v) {      // lane l + 1's value, nothing for lane 63
    AfState r;
    r.s = spw::from_upper(v.s, AF_NEG); r.m0 = (uint32_t)spw::from_upper((int)v.m0, 0); r.m1 = (uint32_t)spw::from_upper((int)v.m1, 0);
    return r;
}

struct AfPair { uint32_t a, b; int32_t diag, pad; };        // a = query (set A), b = target (set B), diag = b_pos - a_pos (the library's convention)
// The rows of a pair the DP has to run (sp_rescore_mappings, below): the stretches of the alignment whose edits do not stand alone.  r0 >= 0: the DP starts at target
// row r0 with ONE live cell in the row before it, diagonal index idx_in, holding the alignment's closed-form score up to there (s_in, nm_in, start_cell); r1 >= 0:
// when row r1 is done and the cell on diagonal idx_out is the one best cell of the row, the rest of the alignment is added in closed form (d_score, d_nm, its end
// end_t / end_q) and the DP stops; otherwise it runs on to the last row.  Between two stretches (n_mid of them, AfMid): when row r_exit is done and the cell on idx_out
// is the one best cell of the row, the DP goes on at row r_entry from that cell alone, d_score / d_nm later, on diagonal idx_in; otherwise it runs through.
struct AfWin { int32_t r0, idx_in, s_in, nm_in; uint32_t start_cell; int32_t r1, idx_out, d_score, d_nm, end_t, end_q, n_mid; };
struct AfMid { int32_t r_exit, idx_out, d_score, d_nm, r_entry, idx_in; };
constexpr int AF_MAXMID = 15;

template <int DPL, bool HASN>
__global__ __launch_bounds__(64) void affine_kernel(SeqSetView A, SeqSetView B, const AfPair* __restrict__ pairs, uint32_t n_pairs, const uint32_t* __restrict__ n_live, sp_affine_opts o,
                                                     sp_affine_aln* __restrict__ out, int t_words_max, const AfWin* __restrict__ wins, const AfMid* __restrict__ mids) {
    extern __shared__ uint32_t lds[];
    const uint32_t p = blockIdx.x;
    if (p >= n_pairs || (n_live && p >= *n_live)) return;              // (n_live: the number of pairs a kernel before this one left in the list)
    const int lane = threadIdx.x;
    const AfPair pr = pairs[p];
    const int tlen = B.len[pr.b], qlen = A.len[pr.a];
    sp_affine_aln res; res.score = 0; res.nm = 0; res.a_start = res.a_end = res.b_start = res.b_end = 0;
    constexpr int BAND = 64 * DPL;
    const int klo = -pr.diag - BAND / 2;                    // diagonal k = q_pos - t_pos of lane 0's first cell
    int i_lo = -(klo + BAND - 1); if (i_lo < 0) i_lo = 0;
    int i_hi = qlen - 1 - klo; if (i_hi > tlen - 1) i_hi = tlen - 1;
    if (pr.pad < 0 || tlen <= 0 || qlen <= 0 || i_lo > i_hi) { if (lane == 0) out[p] = res; return; }      // (a pair marked "skip" by the library's own callers: max_ed < 0)
    // the rows of the target and the query bases they can meet, packed as they are in memory (2 bits per base, + the N plane when the set has one)
    constexpr bool hasn = HASN;
    const int tw0 = i_lo >> 4, tw1 = (i_hi >> 4) + 1;                                   // target words [tw0, tw1)
    int q_lo = i_lo + klo; if (q_lo < 0) q_lo = 0;
    int q_hi = i_hi + klo + BAND - 1; if (q_hi > qlen - 1) q_hi = qlen - 1;
    const int qw0 = q_lo >> 4, qw1 = (q_hi >> 4) + 1;
    uint32_t* LT = lds; uint32_t* LQ = LT + t_words_max; uint32_t* NT = LQ + t_words_max + 2 * BAND / 16 + 8; uint32_t* NQ = NT + t_words_max;
    {
        const uint32_t* tw = B.words + B.word_off[pr.b]; const uint32_t* qw = A.words + A.word_off[pr.a];
        const uint32_t* tn = B.nplane ? B.nplane + B.word_off[pr.b] : nullptr; const uint32_t* qn = A.nplane ? A.nplane + A.word_off[pr.a] : nullptr;
        for (int w = lane; w < tw1 - tw0; w += SP_WAVE) { LT[w] = tw[tw0 + w]; if (hasn) NT[w] = tn ? tn[tw0 + w] : 0u; }
        for (int w = lane; w < qw1 - qw0; w += SP_WAVE) { LQ[w] = qw[qw0 + w]; if (hasn) NQ[w] = qn ? qn[qw0 + w] : 0u; }
    }
    spw::wave_lds_sync();
    auto base_of = [&](const uint32_t* W, const uint32_t* N, int pos, int w0) {
        const int w = (pos >> 4) - w0; const uint32_t sh = (uint32_t)(pos & 15) << 1;
        if (hasn && ((N[w] >> sh) & 1u)) return 4;
        return (int)((W[w] >> sh) & 3u);
    };
    AfState H[DPL], E1[DPL], E2[DPL];
#pragma unroll
    for (int c = 0; c < DPL; ++c) H[c] = E1[c] = E2[c] = af_none();
    int bs = 0, bi = -1, bj = -1; uint32_t bm0 = 0, bm1 = 0;
    const int q1 = o.q, e1 = o.e, q2 = o.q2, e2 = o.e2;
    AfWin win; win.r0 = -1; win.r1 = -1; win.n_mid = 0;
    if (wins) win = wins[p];
    int i_first = i_lo;
    if (win.r0 > i_lo && win.r0 <= i_hi && (unsigned)win.idx_in < (unsigned)BAND) {
        i_first = win.r0;
#pragma unroll
        for (int c = 0; c < DPL; ++c)
            if (lane * DPL + c == win.idx_in) {
                H[c].s = win.s_in; H[c].m0 = win.start_cell; H[c].m1 = (uint32_t)win.nm_in;
                bs = win.s_in; bi = win.r0 - 1; bj = win.r0 - 1 + klo + win.idx_in; bm0 = win.start_cell; bm1 = (uint32_t)win.nm_in;
            }
    }
    const int i_tail = (win.r1 >= i_first && win.r1 < i_hi && (unsigned)win.idx_out < (unsigned)BAND) ? win.r1 : -1;
    bool tail_ok = false; int tail_s = 0; uint32_t tail_m0 = 0, tail_m1 = 0;
    int mid_k = 0; AfMid mid; mid.r_exit = -1;
    if (win.n_mid > 0) mid = mids[(size_t)p * AF_MAXMID];
    int i_check = mid_k < win.n_mid ? mid.r_exit : i_tail, chk_idx = mid_k < win.n_mid ? mid.idx_out : win.idx_out;
    for (int i = i_first; i <= i_hi; ++i) {
        const int ct = base_of(LT, NT, i, tw0);
        // the diagonal above lane's last cell: lane l + 1's first cell of the row before
        const AfState upH = af_from_upper(H[0]), upE1 = af_from_upper(E1[0]), upE2 = af_from_upper(E2[0]);
        AfState hA[DPL], e1n[DPL], e2n[DPL]; AfKey k1[DPL], k2[DPL]; bool valid[DPL];
#pragma unroll
        for (int c = 0; c < DPL; ++c) {
            const int idx = lane * DPL + c, j = i + klo + idx;
            valid[c] = (unsigned)j < (unsigned)qlen;
            const AfState hu = c + 1 < DPL ? H[c + 1] : upH, eu = c + 1 < DPL ? E1[c + 1] : upE1, eu2 = c + 1 < DPL ? E2[c + 1] : upE2;
            AfState a1, a2;
            { const int eo = hu.s - q1; if (eu.s > eo) { a1 = eu; a1.s = eu.s - e1; } else { a1 = hu; a1.s = eo - e1; } a1.m1 += 1; }
            { const int eo = hu.s - q2; if (eu2.s > eo) { a2 = eu2; a2.s = eu2.s - e2; } else { a2 = hu; a2.s = eo - e2; } a2.m1 += 1; }
            if (a1.s < AF_NEG) a1.s = AF_NEG;
            if (a2.s < AF_NEG) a2.s = AF_NEG;
            AfState h = H[c];
            const int cq = valid[c] ? base_of(LQ, NQ, j, qw0) : 4;
            const bool ambi = ct > 3 || cq > 3;
            const int sub = ambi ? -o.sc_ambi : (ct == cq ? o.a : -o.b);
            if (h.s <= 0) { h.s = 0; h.m1 = 0; h.m0 = ((uint32_t)i << 16) | (uint32_t)(j & 0xFFFF); }
            h.s += sub; h.m1 += (ambi || ct != cq) ? 1u : 0u;
            if (a1.s > h.s) h = a1;                         // (F comes between E and E2 in the order of ties: below)
            if (!valid[c]) { h = af_none(); a1 = af_none(); a2 = af_none(); }
            hA[c] = h; e1n[c] = a1; e2n[c] = a2;
            // what this cell offers to the cells to its right as the opening of a gap: the best of its non-F states (an opening behind an F is never better
            // than that F continued), keyed so that the maximum over the cells to the left is the sequential recurrence's choice
            AfState src = h; if (a2.s > src.s) src = a2;
            if (src.s <= 0) { src.s = 0; }                 // (a cell nothing ends in: opening a gap from it scores below zero and never wins)
            const bool offer = valid[c] && src.s > 0;
            k1[c].k = offer ? (((uint32_t)(src.s + idx * e1) + AF_BIAS) << 8 | (uint32_t)idx) : 0u; k1[c].m0 = src.m0; k1[c].m1 = src.m1;
            k2[c].k = offer ? (((uint32_t)(src.s + idx * e2) + AF_BIAS) << 8 | (uint32_t)idx) : 0u; k2[c].m0 = src.m0; k2[c].m1 = src.m1;
        }
        // exclusive prefix maxima over the diagonals to the left: across the lanes by DPP, inside a lane cell by cell
        AfKey in1 = k1[0], in2 = k2[0];
#pragma unroll
        for (int c = 1; c < DPL; ++c) { in1 = af_max(in1, k1[c]); in2 = af_max(in2, k2[c]); }
        in1 = af_from_lower(af_scan(in1)); in2 = af_from_lower(af_scan(in2));
#pragma unroll
        for (int c = 0; c < DPL; ++c) {
            const int idx = lane * DPL + c, j = i + klo + idx;
            AfState f1 = af_none(), f2 = af_none();
            if (in1.k) { const int src_idx = (int)(in1.k & 0xFFu), v = (int)((in1.k >> 8) - AF_BIAS); f1.s = v - q1 - idx * e1; f1.m0 = in1.m0; f1.m1 = in1.m1 + (uint32_t)(idx - src_idx); }
            if (in2.k) { const int src_idx = (int)(in2.k & 0xFFu), v = (int)((in2.k >> 8) - AF_BIAS); f2.s = v - q2 - idx * e2; f2.m0 = in2.m0; f2.m1 = in2.m1 + (uint32_t)(idx - src_idx); }
            AfState h = hA[c];
            if (valid[c]) {
                if (f1.s > h.s) h = f1;
                if (e2n[c].s > h.s) h = e2n[c];
                if (f2.s > h.s) h = f2;
                if (h.s <= 0) { h.s = 0; h.m1 = 0; h.m0 = ((uint32_t)i << 16) | (uint32_t)(j & 0xFFFF); }
                const bool better = h.s > bs || (h.s == bs && h.s > 0 && (i + j < bi + bj || (i + j == bi + bj && i < bi)));
                bs = better ? h.s : bs; bi = better ? i : bi; bj = better ? j : bj; bm0 = better ? h.m0 : bm0; bm1 = better ? h.m1 : bm1;
            }
            H[c] = h; E1[c] = e1n[c]; E2[c] = e2n[c];
            in1 = af_max(in1, k1[c]); in2 = af_max(in2, k2[c]);
        }
        if (i == i_check) {
            // the row behind the last cluster: is the alignment where the library's own one runs, alone at the top of the row?
            int mine = AF_NEG, others = AF_NEG; uint32_t m0 = 0, m1 = 0;
#pragma unroll
            for (int c = 0; c < DPL; ++c) {
                if (lane * DPL + c == chk_idx) { mine = H[c].s; m0 = H[c].m0; m1 = H[c].m1; }
                else if (H[c].s > others) others = H[c].s;
            }
            const int src = chk_idx / DPL;
            const int s_exit = __shfl(mine, src);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { const int v = __shfl_xor(others, d); others = v > others ? v : others; }
            const bool alone = s_exit > 0 && others < s_exit;
            m0 = (uint32_t)__shfl((int)m0, src); m1 = (uint32_t)__shfl((int)m1, src);
            if (mid_k < win.n_mid) {
                if (alone && mid.r_entry > i && mid.r_entry <= i_hi && (unsigned)mid.idx_in < (unsigned)BAND) {
#pragma unroll
                    for (int c = 0; c < DPL; ++c) {
                        H[c] = E1[c] = E2[c] = af_none();
                        if (lane * DPL + c == mid.idx_in) {
                            const int s = s_exit + mid.d_score;
                            H[c].s = s; H[c].m0 = m0; H[c].m1 = m1 + (uint32_t)mid.d_nm;
                            if (s > bs) { bs = s; bi = mid.r_entry - 1; bj = mid.r_entry - 1 + klo + mid.idx_in; bm0 = m0; bm1 = H[c].m1; }
                        }
                    }
                    i = mid.r_entry - 1;
                }
                ++mid_k;
                if (mid_k < win.n_mid) mid = mids[(size_t)p * AF_MAXMID + mid_k];
                i_check = mid_k < win.n_mid ? mid.r_exit : i_tail; chk_idx = mid_k < win.n_mid ? mid.idx_out : win.idx_out;
            } else if (alone) {
                tail_ok = true; tail_s = s_exit + win.d_score; tail_m0 = m0; tail_m1 = m1 + (uint32_t)win.d_nm;
                break;
            } else i_check = -1;
        }
    }
    // the best cell of the wave: highest score, then the smallest anti-diagonal, then the smallest row
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int os = __shfl_xor(bs, d), oi = __shfl_xor(bi, d), oj = __shfl_xor(bj, d);
        const uint32_t om0 = (uint32_t)__shfl_xor((int)bm0, d), om1 = (uint32_t)__shfl_xor((int)bm1, d);
        if (os > bs || (os == bs && os > 0 && (oi + oj < bi + bj || (oi + oj == bi + bj && oi < bi)))) { bs = os; bi = oi; bj = oj; bm0 = om0; bm1 = om1; }
    }
    if (tail_ok && tail_s > bs) { bs = tail_s; bi = win.end_t - 1; bj = win.end_q - 1; bm0 = tail_m0; bm1 = tail_m1; }      // (the end of the alignment lies behind every cell of the rows run: it wins only with the higher score)
    if (lane == 0) {
        if (bs > 0) { res.score = bs; res.nm = (int32_t)bm1; res.b_start = (int32_t)(bm0 >> 16); res.b_end = bi + 1; res.a_start = (int32_t)(bm0 & 0xFFFFu); res.a_end = bj + 1; }
        out[p] = res;
    }
}

} // namespace

// device-side entry for the library's own callers: pairs and results in device memory
int sp_launch_affine(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, const void* d_pairs, uint64_t n_pairs, const sp_affine_opts& o, int band, sp_affine_aln* d_out,
                     const char* prof_name, const uint32_t* d_n_live, const void* d_wins, const void* d_mids) {
    if (n_pairs == 0) return SP_OK;
    if (band != 64 && band != 256) return sp_fail(ctx, SP_ERR_INVALID_ARG, "affine: band must be 64 or 256");
    if (B->max_len > 65535 || A->max_len > 65535) return sp_fail(ctx, SP_ERR_TOO_LONG, "affine: sequences of up to 65,535 bases");
    const int t_words_max = (B->max_len >> 4) + 4;
    const size_t lds_bytes = sizeof(uint32_t) * (size_t)(4 * t_words_max + 2 * (2 * band / 16 + 8));
    ProfScope ps(ctx, prof_name, n_pairs);
    const bool hasn = A->has_n || B->has_n;
#define SP_AF_LAUNCH(D, N) do { \
        (void)hipFuncSetAttribute((const void*)affine_kernel<D, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL((affine_kernel<D, N>), dim3((uint32_t)n_pairs), dim3(64), lds_bytes, ctx->stream, A->view(), B->view(), (const AfPair*)d_pairs, (uint32_t)n_pairs, d_n_live, o, d_out, t_words_max, (const AfWin*)d_wins, (const AfMid*)d_mids); } while (0)
    if (band == 64) { if (hasn) SP_AF_LAUNCH(1, true); else SP_AF_LAUNCH(1, false); }
    else { if (hasn) SP_AF_LAUNCH(4, true); else SP_AF_LAUNCH(4, false); }
#undef SP_AF_LAUNCH
    if (hipGetLastError() != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, "affine launch failed");
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Re-score of mappings the library already has, cheaply: most of them need no DP.  A mapping whose edits all stand alone -- at least AF_ISOLATED bases from one
// another and from both ends of the alignment on the window sequence, no ambiguous base in either sequence -- has the same optimum under the two-piece affine scores
// as under unit costs (a lone mismatch or one-base gap is spelled the same way by both, and an edit that far from an end is not clipped: -4 or -8 against at
// least +16): its numbers are the unit-cost numbers, its score a * matches - b * mismatches - (q + e) * gap bases.  (Measured on 1,242 K1 pairs: every mapping
// with all distances >= 12 had identical numbers; 16 is used.)  The others -- one in nine of the K1 winners -- go through the DP above, over the rows around their clustered edits (below).
// The cells are run again with their traceback (sp_cells_kernel<TRACE>) for the positions of the edits; a cell whose second run differs from the alignment the
// caller holds takes the DP as well.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int AF_ISOLATED = 16;
constexpr int AF_MARGIN = 24;

// d_ref: the alignment the caller holds for each pair (WFA orientation: a_* on Aw, b_* on Bw); cells[x].max_ed < 0: no mapping (score 0)
__global__ void af_classify_kernel(const CellDesc* __restrict__ cells, const sp_aln* __restrict__ ref, const sp_aln* __restrict__ tr, const uint32_t* __restrict__ ev,
                                   uint32_t stride, uint32_t n, int target_is_a, int has_n, sp_affine_opts o, sp_affine_aln* __restrict__ out,
                                   AfPair* __restrict__ todo, uint32_t* __restrict__ todo_at, uint32_t* __restrict__ n_todo, AfWin* __restrict__ wins, AfMid* __restrict__ mids, int band, int windows, int ends_only) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    sp_affine_aln res; res.score = 0; res.nm = 0; res.a_start = res.a_end = res.b_start = res.b_end = 0;
    const CellDesc c = cells[x];
    if (c.max_ed < 0) { out[x] = res; return; }
    const sp_aln r = ref[x], t = tr[x];
    bool simple = !has_n && r.ok && t.ok && t.nm == r.nm && t.a_start == r.a_start && t.a_end == r.a_end && t.b_start == r.b_start && t.b_end == r.b_end && (uint32_t)r.nm <= stride;
    const bool traced = simple;
    int nx = 0, ngap = 0;
    if (simple) {
        const uint32_t* e = ev + (size_t)x * stride;
        int prev = -(1 << 29);
        for (int k = 0; k < r.nm && simple; ++k) {
            const uint32_t w = e[k]; const int pos = (int)(w & 0x3FFFFFFFu), type = (int)(w >> 30);
            const int gap = pos > prev ? pos - prev : prev - pos;
            if (k > 0 && gap < AF_ISOLATED) simple = false;
            if (pos - r.b_start < AF_ISOLATED || r.b_end - pos < AF_ISOLATED) simple = false;
            if (type == (int)SP_EV_X) ++nx; else ++ngap;
            prev = pos;
        }
    }
    // ends_only: the caller takes the mapping's EXTENT (where minimap2's end clipping leaves its two ends) and nothing else.  Where an alignment ends is decided by its last
    // bases alone -- every candidate end carries the same score of what lies in front of it, and that score is far from the restart at zero --: an end without an edit
    // within AF_ISOLATED bases is not clipped (the rule above), and an end with one takes the DP over the rows from that end to the first stretch of AF_MARGIN + AF_ISOLATED
    // bases without an edit; what lies in between enters in the closed form of lone edits, clustered or not (its count and the score's magnitude are then the unit-cost
    // spelling's: such a caller uses neither).  An HLA read is 40 - 100 clustered edits from the reference: the DP over all of its rows was 17.5 ms per 10,000 reads.
    bool ends_head = false, ends_tail = false;
    if (ends_only > 0 && traced && !simple && r.nm > 0) {
        const uint32_t* e = ev + (size_t)x * stride;
        ends_head = (int)(e[0] & 0x3FFFFFFFu) - r.b_start < AF_ISOLATED; ends_tail = r.b_end - (int)(e[r.nm - 1] & 0x3FFFFFFFu) < AF_ISOLATED;
        if (!ends_head && !ends_tail) { simple = true; nx = 0; ngap = 0; for (int k = 0; k < r.nm; ++k) { if ((int)(e[k] >> 30) == (int)SP_EV_X) ++nx; else ++ngap; } }
    }
    if (simple) {
        // columns: M matches, X mismatches, gap bases on either side; a_span = M + X + (A-only bases), b_span = M + X + (B-only bases), gap bases = ngap
        const int a_span = r.a_end - r.a_start, b_span = r.b_end - r.b_start;
        const int m2 = a_span + b_span - 2 * nx - ngap;                          // = 2 M
        const int M = m2 / 2;
        res.score = o.a * M - o.b * nx - (o.q + o.e) * ngap; res.nm = r.nm;
        if (target_is_a) { res.b_start = r.a_start; res.b_end = r.a_end; res.a_start = r.b_start; res.a_end = r.b_end; }
        else { res.a_start = r.a_start; res.a_end = r.a_end; res.b_start = r.b_start; res.b_end = r.b_end; }
        out[x] = res;
        return;
    }
    // the DP: query / target in the affine kernel's order, on the diagonal the alignment lies on
    const uint32_t at = atomicAdd(n_todo, 1u);
    const int d_mid = ((r.b_start - r.a_start) + (r.b_end - r.a_end)) / 2;         // b_pos - a_pos (WFA orientation)
    AfPair p;
    if (target_is_a) { p.a = c.b; p.b = c.a; p.diag = -d_mid; } else { p.a = c.a; p.b = c.b; p.diag = d_mid; }
    p.pad = 0;
    if (!r.ok) p.diag = target_is_a ? -c.diag : c.diag;
    todo[at] = p; todo_at[at] = x;
    // the rows the DP has to run: for every run of edits that do not stand alone, from AF_MARGIN bases before its first edit to AF_MARGIN behind its last one, both ends
    // moved outwards until no other edit lies within AF_ISOLATED bases of them; what lies before, between and behind these stretches is spelled by both scoring schemes
    // the same way (the argument above) and enters in closed form.  An end that would come within AF_ISOLATED bases of the alignment's own end is left to the DP, two
    // stretches less than AF_ISOLATED apart are one, and so are the stretches beyond the AF_MAXMID + 1 a pair can have.
    AfWin w; w.r0 = -1; w.r1 = -1; w.idx_in = w.idx_out = 0; w.s_in = w.nm_in = w.d_score = w.d_nm = w.end_t = w.end_q = w.n_mid = 0; w.start_cell = 0;
    if (traced && windows && r.nm > 0) {
        const uint32_t* e = ev + (size_t)x * stride;
        auto pos_of = [&](int k) { return (int)(e[k] & 0x3FFFFFFFu); };
        int wF[AF_MAXMID + 1], wL[AF_MAXMID + 1], wIn[AF_MAXMID + 1], wOut[AF_MAXMID + 1], nw = 0;
        if (ends_head || ends_tail) {
            // (ends_only: at most two stretches, one per end that has an edit close to it)
            int kh = -1, kt = r.nm;
            if (ends_head) { kh = 0; while (kh + 1 < r.nm && pos_of(kh + 1) - pos_of(kh) < AF_MARGIN + AF_ISOLATED) ++kh; }
            if (ends_tail) { kt = r.nm - 1; while (kt > 0 && pos_of(kt) - pos_of(kt - 1) < AF_MARGIN + AF_ISOLATED) --kt; }
            if (kh < kt) {
                if (ends_head) { wF[nw] = 0; wL[nw] = kh; wIn[nw] = pos_of(0) - AF_MARGIN; wOut[nw] = pos_of(kh) + AF_MARGIN; ++nw; }
                if (ends_tail) { wF[nw] = kt; wL[nw] = r.nm - 1; wIn[nw] = pos_of(kt) - AF_MARGIN; wOut[nw] = pos_of(r.nm - 1) + AF_MARGIN; ++nw; }
            }                                                                                      // (else the two meet: the DP over all rows)
        } else
        for (int k = 0; k < r.nm; ++k) {
            const int pos = pos_of(k);
            const bool lone = !(pos - r.b_start < AF_ISOLATED || r.b_end - pos < AF_ISOLATED) && !(k > 0 && pos - pos_of(k - 1) < AF_ISOLATED) && !(k + 1 < r.nm && pos_of(k + 1) - pos < AF_ISOLATED);
            if (lone) continue;
            int kF = k, kL = k;
            const int floor_k = nw ? wL[nw - 1] + 1 : 0;
            int entry = pos_of(kF) - AF_MARGIN;
            while (kF > floor_k && pos_of(kF - 1) > entry - AF_ISOLATED) { --kF; entry = pos_of(kF) - AF_MARGIN; }
            int leave = pos_of(kL) + AF_MARGIN;
            while (kL + 1 < r.nm && pos_of(kL + 1) < leave + AF_ISOLATED) { ++kL; leave = pos_of(kL) + AF_MARGIN; }
            if (nw && (entry - wOut[nw - 1] < AF_ISOLATED || nw == AF_MAXMID + 1)) { wL[nw - 1] = kL; wOut[nw - 1] = leave; }
            else { wF[nw] = kF; wL[nw] = kL; wIn[nw] = entry; wOut[nw] = leave; ++nw; }
            k = kL;
        }
        if (nw > 0) {
            const int klo = -p.diag - band / 2;
            // the walk along the alignment: (i, j) = the next bases of the streamed and the window sequence; cx / cd / ci = the lone edits since the last stretch
            int i = r.a_start, j = r.b_start, cx = 0, cd = 0, ci = 0, k = 0;
            auto step_to = [&](int k2) { const int pos = pos_of(k2), type = (int)(e[k2] >> 30); i += pos - j; j = pos; if (type == (int)SP_EV_X) { ++i; ++j; ++cx; } else if (type == (int)SP_EV_D) { ++j; ++cd; } else { ++i; ++ci; } };
            AfMid* mid = mids + (size_t)at * AF_MAXMID;
            int out_t = 0, out_idx = 0, out_b = 0; bool have_out = false;
            for (int v = 0; v < nw; ++v) {
                for (; k < wF[v]; ++k) step_to(k);
                {
                    const int ia = i + (wIn[v] - j), jb = wIn[v];                                   // the first cell the DP scores; the one before it is a match on the same diagonal
                    const int t_in = target_is_a ? ia : jb, q_in = target_is_a ? jb : ia;
                    const int idx = (q_in - t_in) - klo;
                    const bool fits = idx >= 0 && idx < band && t_in >= 1 && q_in >= 1;
                    if (v == 0) {
                        if (wIn[0] - r.b_start >= AF_ISOLATED && fits) {
                            w.r0 = t_in; w.idx_in = idx;
                            w.s_in = o.a * ((wIn[0] - r.b_start) - cx - cd) - o.b * cx - (o.q + o.e) * (cd + ci); w.nm_in = cx + cd + ci;
                            const int ts = target_is_a ? r.a_start : r.b_start, qs = target_is_a ? r.b_start : r.a_start;
                            w.start_cell = ((uint32_t)ts << 16) | (uint32_t)(qs & 0xFFFF);
                        }
                    } else if (have_out && fits && t_in > out_t + 1) {
                        AfMid m; m.r_exit = out_t; m.idx_out = out_idx; m.r_entry = t_in; m.idx_in = idx;
                        m.d_score = o.a * ((wIn[v] - out_b - 1) - cx - cd) - o.b * cx - (o.q + o.e) * (cd + ci); m.d_nm = cx + cd + ci;
                        mid[w.n_mid++] = m;
                    }
                }
                for (; k <= wL[v]; ++k) step_to(k);
                cx = cd = ci = 0;
                {
                    const int ia = i + (wOut[v] - j), jb = wOut[v];                                 // a matching cell AF_MARGIN behind the last edit of the stretch
                    const int t_out = target_is_a ? ia : jb, q_out = target_is_a ? jb : ia;
                    const int idx = (q_out - t_out) - klo;
                    have_out = idx >= 0 && idx < band && wOut[v] >= j;
                    out_t = t_out; out_idx = idx; out_b = wOut[v];
                }
            }
            if (have_out && r.b_end - out_b >= AF_ISOLATED) {
                for (; k < r.nm; ++k) step_to(k);
                w.r1 = out_t; w.idx_out = out_idx;
                w.d_score = o.a * ((r.b_end - out_b - 1) - cx - cd) - o.b * cx - (o.q + o.e) * (cd + ci); w.d_nm = cx + cd + ci;
                w.end_t = target_is_a ? r.a_end : r.b_end; w.end_q = target_is_a ? r.b_end : r.a_end;
            }
        }
    }
    wins[at] = w;
}
__global__ void af_scatter_kernel(const sp_affine_aln* __restrict__ part, const uint32_t* __restrict__ todo_at, const uint32_t* __restrict__ n_todo, sp_affine_aln* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < *n_todo) out[todo_at[k]] = part[k];
}

// the mappings of cells (WFA orientation: Aw streamed, Bw window; d_ref = the alignments the caller holds) re-scored into d_out (a_* on minimap2's query, b_* on its target:
// target_is_a tells which of the two sets is the target); everything stays on the device
int sp_rescore_mappings(sp_ctx* ctx, const sp_seqset* Aw, const sp_seqset* Bw, const CellDesc* d_cells, const sp_aln* d_ref, uint64_t n, bool target_is_a,
                        const sp_affine_opts& o, int band, sp_affine_aln* d_out, const char* prefix, uint32_t stride, int trace_retry_wide, const sp_aln* d_tr_in, const uint32_t* d_ev_in, int ends_only) {
    if (n == 0) return SP_OK;
    const std::string pre(prefix);
    static std::mutex names_lock; static std::set<std::string> names;                 // (the profiler keeps the pointers it is given)
    auto stable = [&](const std::string& n2) { std::lock_guard<std::mutex> g(names_lock); return names.insert(n2).first->c_str(); };
    // (d_tr_in / d_ev_in: the caller ran its cells with their traceback itself -- d_ref's own alignments and their edit events, `stride` words each: no second run of the cells)
    const sp_aln* d_tr = d_tr_in ? d_tr_in : (const sp_aln*)sp_pool(ctx, (pre + "_tr").c_str(), n * sizeof(sp_aln));
    const uint32_t* d_ev = d_ev_in ? d_ev_in : (const uint32_t*)sp_pool(ctx, (pre + "_ev").c_str(), n * (size_t)stride * 4);
    AfPair* d_todo = (AfPair*)sp_pool(ctx, (pre + "_todo").c_str(), n * sizeof(AfPair));
    uint32_t* d_at = (uint32_t*)sp_pool(ctx, (pre + "_at").c_str(), n * 4 + 64);
    sp_affine_aln* d_part = (sp_affine_aln*)sp_pool(ctx, (pre + "_part").c_str(), n * sizeof(sp_affine_aln));
    AfWin* d_win = (AfWin*)sp_pool(ctx, (pre + "_win").c_str(), n * sizeof(AfWin));
    AfMid* d_mid = (AfMid*)sp_pool(ctx, (pre + "_mid").c_str(), n * sizeof(AfMid) * AF_MAXMID);
    if (!d_tr || !d_ev || !d_todo || !d_at || !d_part || !d_win || !d_mid) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "rescore buffers");
    uint32_t* d_n = d_at + n;
    (void)hipMemsetAsync(d_n, 0, 4, ctx->stream);
    int rc = SP_OK;
    if (!d_tr_in) rc = sp_launch_cells(ctx, Aw, Bw, d_cells, n, const_cast<sp_aln*>(d_tr), const_cast<uint32_t*>(d_ev), stride, stable(pre + "_trace"), trace_retry_wide);
    if (rc != SP_OK) return rc;
    hipLaunchKernelGGL(af_classify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_cells, d_ref, d_tr, d_ev, stride, (uint32_t)n, target_is_a ? 1 : 0,
                       (Aw->has_n || Bw->has_n) ? 1 : 0, o, d_out, d_todo, d_at, d_n, d_win, d_mid, band, ctx->mm2_rescore == 2 ? 0 : 1, ends_only);
    // the DP over the list the classification left: launched for every pair, the workgroups behind the list's end return at once (no host round trip for the count)
    rc = sp_launch_affine(ctx, target_is_a ? Bw : Aw, target_is_a ? Aw : Bw, d_todo, n, o, band, d_part, stable(pre + "_dp"), d_n, d_win, d_mid);
    if (rc != SP_OK) return rc;
    hipLaunchKernelGGL(af_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_part, d_at, d_n, d_out);
    return SP_OK;
}

extern "C" int32_t sp_affine_rescore_batch(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, const sp_pair* pairs, uint64_t n_pairs, const sp_affine_opts* opts,
                                           int32_t band, sp_affine_aln* out) {
    if (!ctx || !A || !B || !opts || (n_pairs && (!pairs || !out))) return SP_ERR_INVALID_ARG;
    if (n_pairs == 0) return SP_OK;
    if (n_pairs > 0xFFFFFFFFull) return sp_fail(ctx, SP_ERR_INVALID_ARG, "affine: too many pairs");
    (void)hipSetDevice(ctx->device);
    for (uint64_t i = 0; i < n_pairs; ++i) if (pairs[i].a >= A->n || pairs[i].b >= B->n) return sp_fail(ctx, SP_ERR_INVALID_ARG, "affine: index out of range");
    static_assert(sizeof(sp_pair) == sizeof(AfPair), "pair layout");
    void* d_pairs = sp_pool(ctx, "affine_pairs", n_pairs * sizeof(sp_pair));
    sp_affine_aln* d_out = (sp_affine_aln*)sp_pool(ctx, "affine_out", n_pairs * sizeof(sp_affine_aln));
    if (!d_pairs || !d_out) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "affine buffers");
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_pairs, pairs, n_pairs * sizeof(sp_pair), hipMemcpyHostToDevice, ctx->stream));
    const int rc = sp_launch_affine(ctx, A, B, d_pairs, n_pairs, *opts, band, d_out, "affine_rescore", nullptr, nullptr, nullptr);
    if (rc != SP_OK) return rc;
    SP_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, n_pairs * sizeof(sp_affine_aln), hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}
