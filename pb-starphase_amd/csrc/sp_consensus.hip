// sp_consensus.hip -- K8: read consensus by dynamic wavefront alignment (gfx950).
//
// Serves the waffle_con call sites of the reference: DualConsensusDWFA in run_dual_consensus_with_offsets
// (src/hla/caller.rs:1103-1219) and the per-group ConsensusDWFA (src/hla/caller.rs:706-747), configured as
// dwfa_config_from_cli does (src/hla/caller.rs:1103-1116).  waffle_con itself (v0.4.4) is a third-party crate that is not
// on disk, so the contract is the one stated in DESIGN.md section 9 and restated for the CPU in oracle/consensus.c; the
// two agree bit for bit (consensus strings, read assignment, per-read edit counts).
//
// Mapping: ONE wavefront per read, one lane per diagonal (64-diagonal band, lane l <-> consensus pos - read pos = l - 32),
// the per-read state is one VGPR per lane (furthest read position at the current edit count).
//
// The decision for a column needs the votes of every read, so a column is a global step.  One launch per column (round 1) is bound
// by launch latency (~10 us per base).  This version pushes a WINDOW of up to CW bases per launch:
//   * the first base of a window is exact (decided from the complete votes of the verified state);
//   * the others are speculated from the reads' own continuation ("lookahead" votes: every read names the bases that follow its
//     tips) -- at HiFi error rates the majority continuation is the consensus except at real differences;
//   * every wave pushes its read through the window and records the exact votes of every column it passes (exact GIVEN the
//     speculated prefix), the new lookahead votes and the new state into the other state buffer;
//   * a one-workgroup control kernel per problem sums the workgroups' vote words, re-derives the decisions in order and keeps the
//     longest prefix on which decision == speculation with no structural event (stop, split).  A fully accepted window commits the
//     new state buffer; a partly accepted one is re-pushed from the kept state with the verified bases only (no speculation),
//     which costs one extra window.  Events only ever happen at the first base of a window.
// Results are therefore exactly those of the one-base-per-step contract; only the number of launches changes (about two per
// accepted window instead of one per base).  There is no host round trip inside the loop.
#include "sp_internal.h"
#include "sp_wfa.cuh"
#include <algorithm>
#include <cstring>
#include <map>
#include <string>

namespace {

constexpr int CB = 64;          // band
constexpr int CH = 32;          // lane of diagonal 0
constexpr int CWAVES = 16;      // waves per workgroup
constexpr int CW = 32;          // bases per window
constexpr int CWIN = 512;       // consensus bases in front of the window kept in LDS (offset_window + slack)
constexpr int RWORDS = 24;      // packed read words a wave keeps in LDS (384 bases around its tips)
enum { F_ACTIVE = 1, F_FINISHED = 2, F_LOST = 4 };

struct ConsCtrl {               // state between two windows (written by the control kernel, read by the step kernel)
    int32_t T;                  // column of the state in state_buf: consensus [0, T) is final
    int32_t n;                  // bases the next window pushes (0: init or done)
    int32_t replay, init;       // replay: the n bases are verified already (re-push after a cut window); init: build the initial state
    int32_t state_buf;
    int32_t dual, split_at, split_now;   // split_now: the first base of the window splits consensus 2 off (state clone, base spec[1][0])
    int32_t done, pad0;
    int32_t stopped[2], len[2], go[2];
    int32_t windows, cut_windows;        // statistics
    long long best_w2, best_total;
    long long split_w2, split_total;     // the votes of the column at which the second consensus was split off
    uint8_t spec[2][CW];
};
struct ConsMeta { int32_t e, c0, flags, pad; };

// One consensus problem of a batch.  All problems of a batch advance window by window in the same launches (each at its own
// column); a workgroup belongs to exactly one problem (each problem's reads are padded to whole workgroups in the flattened read
// order).  The descriptors travel in the kernel argument block (scalar loads) for small batches.
constexpr int CMAXP = 32;
struct ConsParams {
    int n, first, first_block, n_blocks, rpw;   // reads; flattened index of local read 0; first workgroup; workgroups; reads per wave
    int min_count, delta, et, allow_dual, window, cmp_len; double min_af;
    uint8_t* C; int cap;        // [2][cap] base codes; consensus 2 shares [0, split_at) with consensus 1
    ConsCtrl* ctrl;
};
struct ReadInfo { const uint32_t* w; const uint32_t* np; int n, off; long long pad; };
template <int MAXP> struct ConsBatchT {
    ConsParams p[MAXP]; int n_prob;
    const ReadInfo* info;       // [total]
    uint16_t* H;                // [2 buffers][2 consensuses][total][64] furthest read position per diagonal (0xFFFF = none)
    ConsMeta* meta;             // [2][2][total]
    unsigned long long* PV;     // [blocks][2][CW + 1] exact votes per workgroup: four 16-bit fields (A, C, G, T) in 12ths of a read
    uint32_t* PE;               // [blocks][2][CW + 1] "the read ends here" votes
    unsigned long long* PL;     // [blocks][2][CW]     lookahead votes (one per read and tip)
    int total;
};
// large batches (a cohort): the descriptors live in device memory and a table maps every workgroup to its problem
template <> struct ConsBatchT<0> {
    const ConsParams* p; const int* block_prob; int n_prob;
    const ReadInfo* info; uint16_t* H; ConsMeta* meta; unsigned long long* PV; uint32_t* PE; unsigned long long* PL; int total;
};
struct ConsSetup { SeqSetView reads; const uint32_t* idx; const int32_t* offsets; int n, first; };

struct ReadView { const uint32_t* w; const uint32_t* np; int n; };

__device__ __forceinline__ int h_load(const uint16_t* H, size_t at) { const uint16_t v = H[at]; return v == 0xFFFF ? SP_NEG : (int)v; }
__device__ __forceinline__ void h_store(uint16_t* H, size_t at, int h) { H[at] = h < 0 ? (uint16_t)0xFFFF : (uint16_t)h; }

__device__ __forceinline__ int read_base(const ReadView& rv, int h) {
    const uint32_t sh = (uint32_t)(h & 15) << 1;
    if (rv.np && ((rv.np[h >> 4] >> sh) & 1u)) return 4;
    return (int)((rv.w[h >> 4] >> sh) & 3u);
}

// 12 / d for d = 1..4 distinct tip bases without an integer division
__device__ __forceinline__ uint32_t vote_units(int d) { return d == 1 ? 12u : d == 2 ? 6u : d == 3 ? 4u : 3u; }

struct Dwfa { int H, e, c0, flags; };

// the consensus now has T bases after c0; `nb` is its newest base.  rb(h) = read base at h, ca(pos) = consensus base at pos.
template <class RB, class CA>
__device__ __forceinline__ void dwfa_push_t(Dwfa& d, int n, RB rb, CA ca, int T, int nb, int et, int lane) {
    const int k = lane - CH;
    if (d.H >= 0 && d.H + k == T - 1 && d.H < n && rb(d.H) == nb) d.H += 1;                       // only the old tips can move
    while (!__ballot(d.H >= 0 && d.H + k == T)) {
        const int c = d.H, up = spw::from_lower(d.H, SP_NEG), dn = spw::from_upper(d.H, SP_NEG);
        int best = SP_NEG;
        if (c >= 0 && c < n && c + k < T) best = c + 1;
        if (up >= 0 && up + k <= T && up + k >= 0 && up > best) best = up;
        if (dn >= 0 && dn < n && dn + 1 + k >= 0 && dn + 1 > best) best = dn + 1;
        if (!__ballot(best >= 0)) { d.flags |= F_LOST; return; }
        d.H = best; d.e += 1;
        for (;;) {
            bool go = d.H >= 0 && d.H < n && d.H + k < T;
            if (go) { const int x = rb(d.H); go = x < 4 && x == ca(d.c0 + d.H + k); }
            if (!__ballot(go)) break;
            if (go) d.H += 1;
        }
    }
    if (et && __ballot(d.H == n)) d.flags |= F_FINISHED;
}

// A freshly placed read catches up with `span` consensus bases at once.  Pushing them one by one (dwfa_push_t for T = 1 .. span)
// leaves, after the last push, the wavefront of the smallest edit count at which some diagonal reaches consensus column `span`,
// every diagonal extended as far as it goes inside those columns: while a wavefront still has a tip the pushes only extend
// tips, and a new wavefront is only built when none is left, i.e. from diagonals that are all parked on a mismatch (or the read
// end), where the column bound of that moment excludes nothing.  So the same state is reached by building wavefront after
// wavefront with the extension bounded by `span` alone -- one pass over the span instead of one full push per base.
// Early termination makes the per-base order observable (a read that ends inside the span freezes at that column): callers keep
// the per-base loop for reads that could end inside the span.
template <class EXT>
__device__ __forceinline__ void dwfa_catchup_t(Dwfa& d, int n, EXT extend, int span, int lane) {
    const int k = lane - CH;
    extend();                                            // every diagonal as far as it matches inside the span
    while (!__ballot(d.H >= 0 && d.H + k == span)) {
        const int c = d.H, up = spw::from_lower(d.H, SP_NEG), dn = spw::from_upper(d.H, SP_NEG);
        int best = SP_NEG;
        if (c >= 0 && c < n && c + k < span) best = c + 1;
        if (up >= 0 && up + k <= span && up + k >= 0 && up > best) best = up;
        if (dn >= 0 && dn < n && dn + 1 + k >= 0 && dn + 1 > best) best = dn + 1;
        if (!__ballot(best >= 0)) { d.flags |= F_LOST; break; }
        d.H = best; d.e += 1;
        extend();
    }
}

// placement of a late read: Sellers' search of its first L bases in the last W consensus bases, one Myers bit-vector scan per
// lane over the end positions it owns (an occurrence of an L-base pattern with <= L edits spans <= 2L text bases)
template <class RB, class CA>
__device__ __forceinline__ int find_start(int rn, RB rb, CA ca, int off, int W, int L, int lane) {
    const int ws = off - W > 0 ? off - W : 0, M = off - ws;
    if (L > rn) L = rn;
    if (M <= 0 || L <= 0) return off;
    unsigned long long peq[4];
    {
        const int code = lane < L ? rb(L - 1 - lane) : 7;
#pragma unroll
        for (int b = 0; b < 4; ++b) peq[b] = __ballot(code == b);
    }
    const int q = (M + CB - 1) / CB;
    const int jlo = lane * q + 1, jhi = min(M, jlo + q - 1);
    unsigned long long key = ~0ull;
    if (jlo <= M) {
        const unsigned long long ones = L == 64 ? ~0ull : ((1ull << L) - 1), top = 1ull << (L - 1);
        unsigned long long Pv = ones, Mv = 0;
        int score = L;
        const int centre = off - W / 2;
        for (int j = max(1, jlo - 2 * L); j <= jhi; ++j) {
            const int x = ca(off - j);
            const unsigned long long Eq = x == 0 ? peq[0] : x == 1 ? peq[1] : x == 2 ? peq[2] : peq[3];
            const unsigned long long Xv = Eq | Mv;
            const unsigned long long Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            unsigned long long Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
            if (Ph & top) ++score; else if (Mh & top) --score;
            Ph <<= 1; Mh <<= 1;
            Pv = (Mh | ~(Xv | Ph)) & ones; Mv = Ph & Xv & ones;
            if (j >= jlo) {
                const int p = off - j, dist = p > centre ? p - centre : centre - p;
                const unsigned long long kk = ((unsigned long long)score << 44) | ((unsigned long long)dist << 22) | (unsigned long long)p;
                key = kk < key ? kk : key;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other < key ? other : key; }
    return (int)(key & ((1ull << 22) - 1));
}

constexpr int ACT_CONS = 512;   // consensus bases a wave packs while a late read catches up (offset_window + slack)
constexpr int ACT_READ = 640;   // read bases it keeps (catch-up length + band + edits)
struct ActScratch {             // per wave
    uint8_t rcache[ACT_READ];
    uint32_t cpack[ACT_CONS / 16 + 2];
    uint32_t rpack[ACT_READ / 16 + 2];
};

// heaviest base of a packed vote word, ties to the lower code; second = heaviest of the others
struct ColVotes { uint32_t w[4], end; };
__device__ __forceinline__ void top2(const ColVotes& v, int& b1, uint32_t& w1, int& b2, uint32_t& w2, uint32_t& total) {
    b1 = 0; w1 = v.w[0];
#pragma unroll
    for (int b = 1; b < 4; ++b) if (v.w[b] > w1) { b1 = b; w1 = v.w[b]; }
    b2 = -1; w2 = 0; bool have2 = false;
#pragma unroll
    for (int b = 0; b < 4; ++b) if (b != b1 && (!have2 || v.w[b] > w2)) { b2 = b; w2 = v.w[b]; have2 = true; }
    total = v.w[0] + v.w[1] + v.w[2] + v.w[3];
}

template <int MAXP> __device__ __forceinline__ int block_problem(const ConsBatchT<MAXP>& B) {
    int pi = 0;
    if constexpr (MAXP == 0) pi = B.block_prob[blockIdx.x];
    else {
#pragma unroll
        for (int i = 1; i < MAXP; ++i) if (i < B.n_prob && (int)blockIdx.x >= B.p[i].first_block) pi = i;
    }
    return pi;
}

// ------------------------------------------------------------------------------------------------------------------------------
// the window: every wave pushes its read(s) through the n bases the control kernel set up
// ------------------------------------------------------------------------------------------------------------------------------
template <int MAXP>
__global__ void __launch_bounds__(CWAVES * SP_WAVE) cons_step_kernel(ConsBatchT<MAXP> B) {
    __shared__ unsigned long long lv[2][CW + 1];          // exact votes after j pushes (column T + j)
    __shared__ uint32_t le[2][CW + 1];
    __shared__ unsigned long long ll[2][CW];              // lookahead: ll[i][x] predicts column T + n + 1 + x
    __shared__ uint8_t cwin[2][CWIN + CW];                // consensus bases [T - CWIN, T + n)
    __shared__ uint32_t rwin[CWAVES][2][RWORDS + 2];      // packed read window of the wave (+ N plane)
    __shared__ ActScratch act[CWAVES];
    const int pi = block_problem<MAXP>(B);
    const ConsParams P = B.p[pi];
    const ConsCtrl c = *P.ctrl;
    if (c.done || (c.n == 0 && !c.init)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = c.T, n = c.n;
    const int dual_in = c.dual;                           // the kept state has two consensuses
    const int dual = c.dual || c.split_now;               // the window runs two
    const int split_at = c.split_now ? T : c.split_at;
    for (int x = threadIdx.x; x < 2 * (CW + 1); x += blockDim.x) { (&lv[0][0])[x] = 0; (&le[0][0])[x] = 0; }
    for (int x = threadIdx.x; x < 2 * CW; x += blockDim.x) (&ll[0][0])[x] = 0;
    // the consensus in front of the window and the window itself
    const int w0 = T - CWIN;
    for (int x = threadIdx.x; x < 2 * (CWIN + CW); x += blockDim.x) {
        const int i = x / (CWIN + CW), y = x % (CWIN + CW), pos = w0 + y;
        uint8_t v = 0;
        if (i == 0 || dual) {
            if (pos >= T) v = (pos - T < n) ? P.ctrl->spec[i][pos - T] : 0;
            else if (pos >= 0) v = (i == 1 && pos < split_at) ? P.C[pos] : P.C[(size_t)i * P.cap + pos];
        }
        cwin[i][y] = v;
    }
    __syncthreads();
    const size_t in_buf = (size_t)c.state_buf, out_buf = in_buf ^ 1;
    const size_t plane = (size_t)B.total;                 // state layout [buffer][consensus][flattened read]
    for (int rr = 0; rr < P.rpw; ++rr) {
        const int r = (((int)blockIdx.x - P.first_block) * CWAVES + wave) * P.rpw + rr;
        if (r >= P.n) break;
        const size_t g = (size_t)P.first + r;
        const ReadInfo ri = B.info[g];
        ReadView rv; rv.w = ri.w; rv.np = ri.np; rv.n = ri.n;
        Dwfa d[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { d[i].H = SP_NEG; d[i].e = 0; d[i].c0 = 0; d[i].flags = 0; }
        if (c.init) {
            if (ri.off < 0) { d[0].flags = F_ACTIVE | ((P.et && rv.n == 0) ? F_FINISHED : 0); d[0].H = lane == CH ? 0 : SP_NEG; }
        } else {
            const ConsMeta m0 = B.meta[(in_buf * 2 + 0) * plane + g];
            d[0].H = h_load(B.H, ((in_buf * 2 + 0) * plane + g) * CB + lane); d[0].e = m0.e; d[0].c0 = m0.c0; d[0].flags = m0.flags;
            if (c.split_now) d[1] = d[0];
            else if (dual_in) {
                const ConsMeta m1 = B.meta[(in_buf * 2 + 1) * plane + g];
                d[1].H = h_load(B.H, ((in_buf * 2 + 1) * plane + g) * CB + lane); d[1].e = m1.e; d[1].c0 = m1.c0; d[1].flags = m1.flags;
            }
        }
        // the stretch of the read around its tips: read position of diagonal 0 at column T minus the band, 384 bases from there
        int rbase = 0;
        {
            int lo = 0x7FFFFFFF;
#pragma unroll
            for (int i = 0; i < 2; ++i) if ((i == 0 || dual) && (d[i].flags & F_ACTIVE)) { const int x = T - d[i].c0 - CB; lo = x < lo ? x : lo; }
            if (lo == 0x7FFFFFFF || lo < 0) lo = 0;
            rbase = (lo >> 4) << 4;
            const int w_first = rbase >> 4, w_last = (rv.n + 15) >> 4;          // words [w_first, w_last] exist (guard words follow the sequence)
            if (lane < RWORDS + 2) {
                const int w = w_first + lane;
                rwin[wave][0][lane] = w <= w_last + 1 ? rv.w[w] : 0u;
                rwin[wave][1][lane] = (rv.np && w <= w_last + 1) ? rv.np[w] : 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        auto rb = [&](int h) -> int {
            const int x = h - rbase;
            if ((unsigned)x < (unsigned)(RWORDS * 16)) {
                const uint32_t sh = (uint32_t)(x & 15) << 1;
                if ((rwin[wave][1][x >> 4] >> sh) & 1u) return 4;
                return (int)((rwin[wave][0][x >> 4] >> sh) & 3u);
            }
            return read_base(rv, h);
        };
        auto ca_of = [&](int i) {
            return [&, i](int pos) -> int {
                const int y = pos - w0;
                if ((unsigned)y < (unsigned)(CWIN + CW)) return (int)cwin[i][y];
                return (int)((i == 1 && pos < split_at) ? P.C[pos] : P.C[(size_t)i * P.cap + pos]);
            };
        };
        // votes of the state for the column it stands at (after j pushes), into the workgroup's tallies
        auto vote = [&](int j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !dual) continue;
                if (!c.init && !c.go[i]) continue;
                if (!(d[i].flags & F_ACTIVE) || (d[i].flags & (F_FINISHED | F_LOST))) continue;
                if (dual) { const Dwfa& o = d[1 - i]; if ((o.flags & F_ACTIVE) && !(o.flags & F_LOST) && o.e < d[i].e) continue; }
                const int Tl = T + j - d[i].c0, k = lane - CH;
                const bool tip = d[i].H >= 0 && d[i].H + k == Tl;
                const int code = (tip && d[i].H < rv.n) ? rb(d[i].H) : 5;
                int dc = 0; unsigned long long word = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) { const bool s = __ballot(code == b) != 0; dc += s; word |= s ? (1ull << (16 * b)) : 0ull; }
                const bool ended = __ballot(tip) != 0 && __ballot(code == 4) == 0;      // every tip sits at the end of the read
                if (lane == 0) {
                    if (dc) atomicAdd(&lv[i][j], word * vote_units(dc));
                    else if (ended) atomicAdd(&le[i][j], 12u);
                }
            }
        };
        if (c.init) vote(0);
        for (int j = 0; j < n; ++j) {
            const int len = T + j + 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !dual) continue;
                if (!c.go[i]) continue;
                const int nb = cwin[i][CWIN + j];
                if (d[i].flags & F_ACTIVE) {
                    if (!(d[i].flags & (F_FINISHED | F_LOST))) dwfa_push_t(d[i], rv.n, rb, ca_of(i), len - d[i].c0, nb, P.et, lane);
                } else if (ri.off == len) {
                    // a late read (add_sequence_offset): start search in the window before the offset, then catch up
                    auto ca = ca_of(i);
                    ActScratch& A = act[wave];
                    const int ws = ri.off - P.window > 0 ? ri.off - P.window : 0;
                    for (int x = lane; x < rv.n && x < ACT_READ; x += SP_WAVE) A.rcache[x] = (uint8_t)read_base(rv, x);
                    spw::wave_lds_sync();
                    auto rbc = [&](int h) { return h < ACT_READ ? (int)A.rcache[h] : read_base(rv, h); };
                    d[i].c0 = find_start(rv.n, rbc, ca, ri.off, P.window, P.cmp_len, lane);
                    d[i].H = lane == CH ? 0 : SP_NEG; d[i].e = 0; d[i].flags = F_ACTIVE | ((P.et && rv.n == 0) ? F_FINISHED : 0);
                    const int c0 = d[i].c0, span = len - c0, cwinlen = len - ws;
                    const bool packed = rv.np == nullptr && cwinlen <= ACT_CONS;
                    if (span > 0 && !(d[i].flags & F_FINISHED) && !(P.et && rv.n <= span + CB) && packed) {
                        // 16 bases per step out of 2-bit packed copies of the two windows
                        for (int w = lane; w < ACT_CONS / 16 + 2; w += SP_WAVE) {
                            uint32_t word = 0;
                            for (int b = 0; b < 16; ++b) { const int x = w * 16 + b; if (x < cwinlen) word |= (uint32_t)(ca(ws + x) & 3) << (b << 1); }
                            A.cpack[w] = word;
                        }
                        const int rwords = ((rv.n < ACT_READ ? rv.n : ACT_READ) + 15) >> 4;
                        for (int w = lane; w < ACT_READ / 16 + 2; w += SP_WAVE) A.rpack[w] = w < rwords ? rv.w[w] : 0u;
                        spw::wave_lds_sync();
                        const int kk = lane - CH, cbase = c0 - ws;
                        dwfa_catchup_t(d[i], rv.n, [&]() {
                            for (;;) {
                                Dwfa& q = d[i];
                                int left = rv.n - q.H; { const int l2 = span - (q.H + kk); left = l2 < left ? l2 : left; }
                                const bool go = q.H >= 0 && left > 0;
                                int nm = 0;
                                if (go) {
                                    const int pr = q.H, pc = cbase + q.H + kk;
                                    const uint32_t a = __builtin_amdgcn_alignbit(A.rpack[(pr >> 4) + 1], A.rpack[pr >> 4], (uint32_t)(pr & 15) << 1);
                                    const uint32_t b = __builtin_amdgcn_alignbit(A.cpack[(pc >> 4) + 1], A.cpack[pc >> 4], (uint32_t)(pc & 15) << 1);
                                    const uint32_t x = a ^ b, mm = (x | (x >> 1)) & 0x55555555u;
                                    nm = mm ? (__builtin_ctz(mm) >> 1) : 16;
                                    nm = nm < left ? nm : left;
                                    q.H += nm;
                                }
                                if (!__ballot(go && nm == 16 && left > 16)) break;
                            }
                        }, span, lane);
                    } else if (span > 0 && !(d[i].flags & F_FINISHED) && !(P.et && rv.n <= span + CB)) {
                        const int kk = lane - CH;
                        dwfa_catchup_t(d[i], rv.n, [&]() {
                            for (;;) {
                                Dwfa& q = d[i];
                                bool go = q.H >= 0 && q.H < rv.n && q.H + kk < span;
                                if (go) { const int x = rbc(q.H); go = x < 4 && x == ca(q.c0 + q.H + kk); }
                                if (!__ballot(go)) break;
                                if (go) q.H += 1;
                            }
                        }, span, lane);
                    } else {
                        for (int Tl = 1; Tl <= span; ++Tl) {
                            if (d[i].flags & (F_FINISHED | F_LOST)) break;
                            dwfa_push_t(d[i], rv.n, rbc, ca, Tl, ca(c0 + Tl - 1), P.et, lane);
                        }
                    }
                    spw::wave_lds_sync();
                }
            }
            if (dual) {
                const int both = (d[0].flags & F_ACTIVE) && (d[1].flags & F_ACTIVE) && !(d[0].flags & F_LOST) && !(d[1].flags & F_LOST);
                if (both) {
                    if (d[0].e > d[1].e + P.delta) d[0].flags |= F_LOST;
                    else if (d[1].e > d[0].e + P.delta) d[1].flags |= F_LOST;
                }
            }
            vote(j + 1);
        }
        // lookahead: the bases behind every tip (at most two tips per consensus speak) predict the columns after the window
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !dual) continue;
            if (!c.init && !c.go[i]) continue;
            if (!(d[i].flags & F_ACTIVE) || (d[i].flags & (F_FINISHED | F_LOST))) continue;
            if (dual) { const Dwfa& o = d[1 - i]; if ((o.flags & F_ACTIVE) && !(o.flags & F_LOST) && o.e < d[i].e) continue; }
            const int Tl = T + n - d[i].c0, k = lane - CH;
            unsigned long long tips = __ballot(d[i].H >= 0 && d[i].H + k == Tl && d[i].H < rv.n);
            for (int cnt = 0; tips && cnt < 2; ++cnt) {
                const int tl = __builtin_ctzll(tips); tips &= tips - 1;
                const int h = __builtin_amdgcn_readlane(d[i].H, tl);
                if (lane < CW - 1 && h + 1 + lane < rv.n) {
                    const int b = rb(h + 1 + lane);
                    if (b < 4) atomicAdd(&ll[i][lane], 1ull << (16 * b));
                }
            }
        }
        // the new state goes to the other buffer
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !dual) continue;
            if (lane == 0) { ConsMeta m; m.e = d[i].e; m.c0 = d[i].c0; m.flags = d[i].flags; m.pad = 0; B.meta[(out_buf * 2 + i) * plane + g] = m; }
            h_store(B.H, ((out_buf * 2 + i) * plane + g) * CB + lane, d[i].H);
        }
        spw::wave_lds_sync();
    }
    __syncthreads();
    for (int x = threadIdx.x; x < 2 * (CW + 1); x += blockDim.x) {
        B.PV[(size_t)blockIdx.x * 2 * (CW + 1) + x] = (&lv[0][0])[x];
        B.PE[(size_t)blockIdx.x * 2 * (CW + 1) + x] = (&le[0][0])[x];
    }
    for (int x = threadIdx.x; x < 2 * CW; x += blockDim.x) B.PL[(size_t)blockIdx.x * 2 * CW + x] = (&ll[0][0])[x];
}

// ------------------------------------------------------------------------------------------------------------------------------
// the control step: one workgroup per problem sums the vote words of the problem's workgroups, verifies the window and sets up
// the next one
// ------------------------------------------------------------------------------------------------------------------------------
template <int MAXP>
__global__ void __launch_bounds__(1024) cons_control_kernel(ConsBatchT<MAXP> B) {
    __shared__ uint32_t sv[2][CW + 1][5];                 // summed exact votes: w[4], end
    __shared__ uint32_t sl[2][CW][4];                     // summed lookahead votes
    __shared__ ConsCtrl cs;
    const int pi = blockIdx.x;
    const ConsParams P = B.p[pi];
    if (threadIdx.x == 0) cs = *P.ctrl;
    for (int x = threadIdx.x; x < 2 * (CW + 1) * 5; x += blockDim.x) (&sv[0][0][0])[x] = 0;
    for (int x = threadIdx.x; x < 2 * CW * 4; x += blockDim.x) (&sl[0][0][0])[x] = 0;
    __syncthreads();
    if (cs.done || (cs.n == 0 && !cs.init)) return;
    const int n = cs.n;
    {
        // element e of a workgroup's partial block: e < 2 (CW + 1): exact votes (V + E); then 2 CW lookahead words
        const int EV = 2 * (CW + 1), EL = 2 * CW, E = EV + EL;
        const int per = blockDim.x / E > 0 ? blockDim.x / E : 1;
        const int e = threadIdx.x % E, sub = threadIdx.x / E;
        if (sub < per) {
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
            const bool is_v = e < EV;
            const int j = is_v ? e % (CW + 1) : (e - EV) % CW;
            const bool wanted = is_v ? j <= n : true;
            if (wanted) for (int b = sub; b < P.n_blocks; b += per) {
                const size_t blk = (size_t)P.first_block + b;
                if (is_v) {
                    const unsigned long long v = B.PV[blk * EV + e];
                    a0 += (uint32_t)(v & 0xFFFF); a1 += (uint32_t)((v >> 16) & 0xFFFF); a2 += (uint32_t)((v >> 32) & 0xFFFF); a3 += (uint32_t)(v >> 48);
                    a4 += B.PE[blk * EV + e];
                } else {
                    const unsigned long long v = B.PL[blk * EL + (e - EV)];
                    a0 += (uint32_t)(v & 0xFFFF); a1 += (uint32_t)((v >> 16) & 0xFFFF); a2 += (uint32_t)((v >> 32) & 0xFFFF); a3 += (uint32_t)(v >> 48);
                }
            }
            if (is_v) { uint32_t* d = &sv[e / (CW + 1)][j][0]; atomicAdd(d, a0); atomicAdd(d + 1, a1); atomicAdd(d + 2, a2); atomicAdd(d + 3, a3); atomicAdd(d + 4, a4); }
            else { uint32_t* d = &sl[(e - EV) / CW][j][0]; atomicAdd(d, a0); atomicAdd(d + 1, a1); atomicAdd(d + 2, a2); atomicAdd(d + 3, a3); }
        }
    }
    __syncthreads();
    if (threadIdx.x >= SP_WAVE) return;
    const int lane = threadIdx.x;
    const int dual = cs.dual || cs.split_now;
    const int T = cs.T;
    // the decision the complete votes make for the column after j pushes, per consensus
    auto decide = [&](int i, int j, int& base, int& b2, uint32_t& w2, uint32_t& total) -> bool {
        ColVotes v; for (int b = 0; b < 4; ++b) v.w[b] = sv[i][j][b]; v.end = sv[i][j][4];
        int b1; uint32_t w1; top2(v, b1, w1, b2, w2, total);
        base = b1;
        if (T + j >= P.cap) return false;                                               // out of room: the consensus is cut at cap
        return P.et ? w1 > 0 : (total > v.end && w1 > 0);
    };
    // 1. how much of the window stands: lane j checks the base pushed as number j (1 <= j < n) against the votes after j pushes
    int a = n;
    uint32_t my_w2 = 0, my_total = 1; bool my_cand = false;
    if (!cs.replay && n > 1) {
        bool ok = true;
        if (lane >= 1 && lane < n) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !dual) continue;
                if (!cs.go[i]) continue;
                int base, b2; uint32_t w2, total;
                const bool go = decide(i, lane, base, b2, w2, total);
                if (!go || base != cs.spec[i][lane]) ok = false;
                if (!dual && i == 0 && go && w2 >= 12u * (uint32_t)P.min_count) {
                    my_cand = true; my_w2 = w2; my_total = total;
                    if (P.allow_dual && (double)w2 >= P.min_af * (double)total) ok = false;       // a split is an event: it opens a window
                }
            }
        }
        const unsigned long long bad = __ballot(!ok);
        if (bad) a = __builtin_ctzll(bad);
    }
    // strongest second-base column among the accepted ones, in column order (a strictly better ratio replaces)
    if (!dual && !cs.replay) {
        unsigned long long cand = __ballot(my_cand && lane < a);
        while (cand) {
            const int l = __builtin_ctzll(cand); cand &= cand - 1;
            const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)my_w2, l), tot = (uint32_t)__builtin_amdgcn_readlane((int)my_total, l);
            if (lane == 0 && (unsigned long long)w2 * (unsigned long long)cs.best_total > (unsigned long long)cs.best_w2 * (unsigned long long)tot) { cs.best_w2 = w2; cs.best_total = tot; }
        }
    }
    // 2. commit the accepted bases
    if (lane < a) {
#pragma unroll
        for (int i = 0; i < 2; ++i) if ((i == 0 || dual) && cs.go[i]) P.C[(size_t)i * P.cap + T + lane] = cs.spec[i][lane];
    }
    if (lane != 0) return;
    cs.windows += 1;
    for (int i = 0; i < 2; ++i) if ((i == 0 || dual) && cs.go[i] && a > 0) cs.len[i] = T + a;
    if (a < n) {                                          // cut window: push the verified bases again, from the kept state
        cs.n = a; cs.replay = 1; cs.cut_windows += 1;
        *P.ctrl = cs;
        return;
    }
    // 3. the whole window stands: the new state is the other buffer; decide the column it stands at and speculate on
    cs.T = T + n; cs.state_buf ^= 1; cs.replay = 0; cs.init = 0;
    if (cs.split_now) { cs.dual = 1; cs.split_at = T; cs.split_now = 0; }
    const int nd = cs.dual ? 2 : 1;
    int going = 0;
    for (int i = 0; i < 2; ++i) cs.go[i] = 0;
    for (int i = 0; i < nd; ++i) {
        if (cs.stopped[i]) continue;
        int base, b2; uint32_t w2, total;
        const bool go = decide(i, n, base, b2, w2, total);
        if (!go) { cs.stopped[i] = 1; continue; }
        cs.go[i] = 1; cs.spec[i][0] = (uint8_t)base; going += 1;
        if (!cs.dual && w2 >= 12u * (uint32_t)P.min_count) {
            if ((unsigned long long)w2 * (unsigned long long)cs.best_total > (unsigned long long)cs.best_w2 * (unsigned long long)total) { cs.best_w2 = w2; cs.best_total = total; }
            if (P.allow_dual && (double)w2 >= P.min_af * (double)total) {
                cs.split_now = 1; cs.go[1] = 1; cs.stopped[1] = 0; cs.spec[1][0] = (uint8_t)b2; cs.split_w2 = w2; cs.split_total = total;
            }
        }
    }
    if (!going) { cs.done = 1; cs.n = 0; *P.ctrl = cs; return; }
    int nn = 1;
    if (!cs.split_now) {
        for (; nn < CW && cs.T + nn < P.cap; ++nn) {
            bool have = true;
            for (int i = 0; i < nd && have; ++i) {
                if (!cs.go[i]) continue;
                const uint32_t* w = sl[i][nn - 1];
                int b1 = 0; uint32_t w1 = w[0];
                for (int b = 1; b < 4; ++b) if (w[b] > w1) { b1 = b; w1 = w[b]; }
                if (w1 == 0) have = false; else cs.spec[i][nn] = (uint8_t)b1;
            }
            if (!have) break;
        }
    }
    cs.n = nn;
    *P.ctrl = cs;
}

// gathers the per-read constants of one problem into the flattened ReadInfo array (once per batch)
__global__ void cons_setup_kernel(ConsSetup S, ReadInfo* __restrict__ info) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= S.n) return;
    const uint32_t rid = S.idx ? S.idx[r] : (uint32_t)r;
    ReadInfo ri;
    ri.w = S.reads.words + S.reads.word_off[rid];
    ri.np = S.reads.nplane ? S.reads.nplane + S.reads.word_off[rid] : nullptr;
    ri.n = S.reads.len[rid]; ri.off = S.offsets ? S.offsets[r] : -1; ri.pad = 0;
    info[S.first + r] = ri;
}

template <int MAXP>
__global__ void __launch_bounds__(CWAVES * SP_WAVE) cons_finalize_kernel(ConsBatchT<MAXP> B, uint8_t* is_cons1, int32_t* score1, int32_t* score2) {
    const int pi = block_problem<MAXP>(B);
    const ConsParams P = B.p[pi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const ConsCtrl c = *P.ctrl;
    const size_t buf = (size_t)c.state_buf, plane = (size_t)B.total;
    for (int rr = 0; rr < P.rpw; ++rr) {
        const int r = (((int)blockIdx.x - P.first_block) * CWAVES + wave) * P.rpw + rr;
        if (r >= P.n) break;
        const size_t g = (size_t)P.first + r;
        const int n = B.info[g].n;
        int sc[2] = { -1, -1 };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !c.dual) continue;
            const ConsMeta m = B.meta[(buf * 2 + i) * plane + g];
            if (!(m.flags & F_ACTIVE) || (m.flags & F_LOST)) continue;
            int e = m.e;
            if (!P.et) {
                const int h = h_load(B.H, ((buf * 2 + i) * plane + g) * CB + lane), k = lane - CH;
                int rest = (h >= 0 && h + k == c.len[i] - m.c0) ? n - h : (1 << 30);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(rest, o); rest = other < rest ? other : rest; }
                if (rest < (1 << 30)) e += rest;
            }
            sc[i] = e;
        }
        if (lane == 0) {
            score1[g] = sc[0]; score2[g] = sc[1];
            is_cons1[g] = !(sc[1] >= 0 && (sc[0] < 0 || sc[1] < sc[0]));
        }
    }
}

} // namespace

// host side of a batch of at most CMAXP problems (or any number with the descriptors in device memory): all of them advance one
// window per launch pair until every one has stopped
template <int MAXP>
static int32_t run_chunk(sp_ctx* ctx, uint32_t n_prob, const sp_cons_problem* probs, sp_cons_output* outs) {
    hipStream_t st = ctx->stream;
    ConsBatchT<MAXP> B; std::memset(&B, 0, sizeof B);
    B.n_prob = (int)n_prob;
    std::vector<ConsParams> hp(n_prob);                      // the descriptors; they end up in the kernel arguments or, for MAXP == 0, in device memory
    std::vector<int> block_prob;
    std::vector<ConsSetup> setup(n_prob);
    std::vector<uint32_t> h_idx; std::vector<int32_t> h_off;
    std::vector<size_t> idx_at(n_prob), off_at(n_prob), c_at(n_prob);
    size_t total = 0, c_bytes = 0; int max_cap = 0, n_blocks = 0;
    for (uint32_t p = 0; p < n_prob; ++p) {
        const sp_cons_problem& q = probs[p];
        const uint32_t n = q.read_idx ? q.n : q.reads->n;
        ConsParams& P = hp[p];
        P.n = (int)n; P.cap = (int)outs[p].cap - 1;                                   // one byte of the caller's buffer is the NUL
        P.first = (int)total; P.first_block = n_blocks;
        P.min_count = q.cfg.min_count; P.delta = q.cfg.dual_max_ed_delta; P.et = q.cfg.allow_early_termination != 0; P.allow_dual = q.cfg.allow_dual != 0;
        P.window = q.cfg.offset_window; P.cmp_len = q.cfg.offset_compare_length; P.min_af = q.cfg.min_af;
        P.rpw = n > 16384 ? 4 : n > 4096 ? 2 : 1;                                     // large problems: several reads per wave (fewer vote words to sum)
        const uint32_t per_block = (uint32_t)(CWAVES * P.rpw);
        const uint32_t nb = (n + per_block - 1) / per_block;
        P.n_blocks = (int)nb;
        n_blocks += (int)nb;
        if (MAXP == 0) block_prob.insert(block_prob.end(), nb, (int)p);
        setup[p].reads = q.reads->view(); setup[p].n = (int)n; setup[p].first = (int)total;
        total += (size_t)nb * per_block;
        idx_at[p] = h_idx.size(); if (q.read_idx) h_idx.insert(h_idx.end(), q.read_idx, q.read_idx + n);
        off_at[p] = h_off.size(); if (q.offsets) h_off.insert(h_off.end(), q.offsets, q.offsets + n);
        c_at[p] = c_bytes; c_bytes += 2 * (size_t)std::max(P.cap, 1);
        max_cap = std::max(max_cap, P.cap);
    }
    if (n_blocks == 0) return SP_OK;
    uint32_t* d_idx = (uint32_t*)sp_pool(ctx, "cons_idx", sizeof(uint32_t) * std::max<size_t>(1, h_idx.size()));
    int32_t* d_off = (int32_t*)sp_pool(ctx, "cons_off", sizeof(int32_t) * std::max<size_t>(1, h_off.size()));
    uint8_t* d_C = (uint8_t*)sp_pool(ctx, "cons_C", c_bytes);
    ConsCtrl* d_ctrl = (ConsCtrl*)sp_pool(ctx, "cons_ctrl", sizeof(ConsCtrl) * n_prob);
    ReadInfo* d_info = (ReadInfo*)sp_pool(ctx, "cons_info", sizeof(ReadInfo) * total);
    B.info = d_info; B.total = (int)total;
    B.H = (uint16_t*)sp_pool(ctx, "cons_H", sizeof(uint16_t) * 4 * total * CB);
    B.meta = (ConsMeta*)sp_pool(ctx, "cons_meta", sizeof(ConsMeta) * 4 * total);
    B.PV = (unsigned long long*)sp_pool(ctx, "cons_pv", sizeof(unsigned long long) * (size_t)n_blocks * 2 * (CW + 1));
    B.PE = (uint32_t*)sp_pool(ctx, "cons_pe", sizeof(uint32_t) * (size_t)n_blocks * 2 * (CW + 1));
    B.PL = (unsigned long long*)sp_pool(ctx, "cons_pl", sizeof(unsigned long long) * (size_t)n_blocks * 2 * CW);
    uint8_t* d_is1 = (uint8_t*)sp_pool(ctx, "cons_is1", total);
    int32_t* d_sc = (int32_t*)sp_pool(ctx, "cons_scores", sizeof(int32_t) * 2 * total);
    ConsCtrl* h_ctrl = (ConsCtrl*)sp_host_pool(ctx, "cons_ctrl", sizeof(ConsCtrl) * n_prob);
    if (!d_idx || !d_off || !d_C || !d_ctrl || !d_info || !B.H || !B.meta || !B.PV || !B.PE || !B.PL || !d_is1 || !d_sc || !h_ctrl)
        return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus buffers");
    for (uint32_t p = 0; p < n_prob; ++p) {
        setup[p].idx = probs[p].read_idx ? d_idx + idx_at[p] : nullptr;
        setup[p].offsets = probs[p].offsets ? d_off + off_at[p] : nullptr;
        hp[p].C = d_C + c_at[p]; hp[p].ctrl = d_ctrl + p;
    }
    if constexpr (MAXP == 0) {
        ConsParams* d_probs = (ConsParams*)sp_pool(ctx, "cons_probs", sizeof(ConsParams) * n_prob);
        int* d_block_prob = (int*)sp_pool(ctx, "cons_block_prob", sizeof(int) * block_prob.size());
        if (!d_probs || !d_block_prob) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus descriptors");
        SP_HIP_CHECK(ctx, hipMemcpyAsync(d_probs, hp.data(), sizeof(ConsParams) * n_prob, hipMemcpyHostToDevice, st));
        SP_HIP_CHECK(ctx, hipMemcpyAsync(d_block_prob, block_prob.data(), sizeof(int) * block_prob.size(), hipMemcpyHostToDevice, st));
        B.p = d_probs; B.block_prob = d_block_prob;
    } else {
        for (uint32_t p = 0; p < n_prob; ++p) B.p[p] = hp[p];
    }
    ConsCtrl c0; std::memset(&c0, 0, sizeof c0); c0.init = 1; c0.split_at = -1; c0.stopped[1] = 1; c0.best_total = 1; c0.split_total = 1;
    for (uint32_t p = 0; p < n_prob; ++p) h_ctrl[p] = c0;
    if (!h_idx.empty()) SP_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, h_idx.data(), sizeof(uint32_t) * h_idx.size(), hipMemcpyHostToDevice, st));
    if (!h_off.empty()) SP_HIP_CHECK(ctx, hipMemcpyAsync(d_off, h_off.data(), sizeof(int32_t) * h_off.size(), hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_ctrl, h_ctrl, sizeof(ConsCtrl) * n_prob, hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemsetAsync(B.meta, 0, sizeof(ConsMeta) * 4 * total, st));
    SP_HIP_CHECK(ctx, hipMemsetAsync(d_info, 0, sizeof(ReadInfo) * total, st));
    for (uint32_t p = 0; p < n_prob; ++p)
        if (setup[p].n) hipLaunchKernelGGL(cons_setup_kernel, dim3((setup[p].n + 255) / 256), dim3(256), 0, st, setup[p], d_info);
    SP_HIP_CHECK(ctx, hipStreamSynchronize(st));      // the pageable host sources above must stay valid until copied

    const dim3 grid((uint32_t)n_blocks), block(CWAVES * SP_WAVE);
    uint64_t windows = 0;
    {
        ProfScope ps(ctx, "cons_steps", total);
        // the first poll comes when a consensus of max_cap bases can be through if (nearly) every window stands; then every 16 windows.
        // A window pair whose problems are all done is a pair of empty launches.
        int until_poll = max_cap / CW + 8;
        const uint64_t limit = (uint64_t)4 * (uint64_t)(max_cap + 2) + 64;           // every column costs at most a cut and a replay
        for (;;) {
            hipLaunchKernelGGL(cons_step_kernel<MAXP>, grid, block, 0, st, B);
            hipLaunchKernelGGL(cons_control_kernel<MAXP>, dim3(n_prob), dim3(1024), 0, st, B);
            ++windows;
            if (--until_poll <= 0 || windows >= limit) {
                SP_HIP_CHECK(ctx, hipMemcpyAsync(h_ctrl, d_ctrl, sizeof(ConsCtrl) * n_prob, hipMemcpyDeviceToHost, st));
                SP_HIP_CHECK(ctx, hipStreamSynchronize(st));
                bool all = true; int left = 0;
                for (uint32_t p = 0; p < n_prob; ++p) { all = all && h_ctrl[p].done; if (!h_ctrl[p].done) left = std::max(left, hp[p].cap - h_ctrl[p].T); }
                if (all) break;
                if (windows >= limit) return sp_fail(ctx, SP_ERR_HIP, "sp_consensus: the window loop did not finish");
                until_poll = std::max(4, std::min(64, left / CW + 2));
            }
        }
    }
    SP_HIP_CHECK(ctx, hipGetLastError());
    if (ctx->profiling) {
        unsigned long long* cnt = sp_counters(ctx);
        (void)cnt;
    }
    hipLaunchKernelGGL(cons_finalize_kernel<MAXP>, grid, block, 0, st, B, d_is1, d_sc, d_sc + total);
    std::vector<uint8_t> hc(c_bytes), h_is1(total);
    std::vector<int32_t> h_sc(2 * total);
    SP_HIP_CHECK(ctx, hipMemcpyAsync(hc.data(), d_C, c_bytes, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(h_is1.data(), d_is1, total, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(h_sc.data(), d_sc, sizeof(int32_t) * 2 * total, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(st));
    SP_HIP_CHECK(ctx, hipGetLastError());
    {   // launch statistics of the batch (sp_profile_get "cons_windows" / "cons_cut_windows": cells = count)
        uint64_t w = 0, cut = 0, cols = 0;
        for (uint32_t p = 0; p < n_prob; ++p) { w = std::max<uint64_t>(w, (uint64_t)h_ctrl[p].windows); cut += (uint64_t)h_ctrl[p].cut_windows; cols = std::max<uint64_t>(cols, (uint64_t)h_ctrl[p].T); }
        ctx->prof["cons_windows"].cells += windows; ctx->prof["cons_windows"].launches += 2 * windows;
        ctx->prof["cons_cut_windows"].cells += cut; ctx->prof["cons_columns"].cells += cols;
    }
    static const char dec[4] = { 'A', 'C', 'G', 'T' };
    int32_t rc = SP_OK;
    for (uint32_t p = 0; p < n_prob; ++p) {
        const ConsParams& P = hp[p]; sp_cons_output& o = outs[p];
        const ConsCtrl& cur = h_ctrl[p];
        const uint8_t* c = hc.data() + c_at[p];
        const int len1 = cur.len[0], len2 = cur.dual ? cur.len[1] : 0;
        for (int x = 0; x < len1; ++x) o.cons1[x] = dec[c[x] & 3];
        o.cons1[len1] = '\0';
        for (int x = 0; x < len2; ++x) o.cons2[x] = dec[(x < cur.split_at ? c[x] : c[(size_t)P.cap + x]) & 3];
        o.cons2[len2] = '\0';
        for (int r = 0; r < P.n; ++r) { o.is_cons1[r] = h_is1[P.first + r]; o.score1[r] = h_sc[P.first + r]; o.score2[r] = h_sc[total + P.first + r]; }
        o.result.is_dual = cur.dual; o.result.len1 = len1; o.result.len2 = len2; o.result.split_at = cur.split_at;
        o.result.best_w2 = cur.best_w2; o.result.best_total = cur.best_total;
        o.result.split_w2 = cur.split_w2; o.result.split_total = cur.split_total;
        // a consensus that filled its buffer was still growing: the caller sized cap too small
        if (len1 >= P.cap || len2 >= P.cap) { o.status = SP_ERR_CAPACITY; rc = SP_ERR_CAPACITY; }
    }
    if (rc != SP_OK) sp_fail(ctx, rc, "sp_consensus: a consensus reached cap");
    return rc;
}

static int32_t run_batch(sp_ctx* ctx, uint32_t n_prob, const sp_cons_problem* probs, sp_cons_output* outs) {
    for (uint32_t p = 0; p < n_prob; ++p) {
        const sp_cons_problem& q = probs[p]; sp_cons_output& o = outs[p];
        if (!q.reads || !o.cons1 || !o.cons2 || o.cap == 0 || !o.is_cons1 || !o.score1 || !o.score2) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: null argument");
        if (q.cfg.offset_compare_length > 64 || q.cfg.offset_compare_length < 0 || q.cfg.offset_window < 0 || q.cfg.min_count < 0)
            return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: offset_compare_length must be in [0, 64]");
        if (o.cap >= (1u << 22)) return sp_fail(ctx, SP_ERR_TOO_LONG, "sp_consensus: cap must be below 4,194,304");
        if (q.reads->max_len >= 65535) return sp_fail(ctx, SP_ERR_TOO_LONG, "sp_consensus: sequences must be shorter than 65,535 bases");
        std::memset(&o.result, 0, sizeof o.result); o.result.split_at = -1; o.result.best_total = 1; o.status = SP_OK;
        o.cons1[0] = o.cons2[0] = '\0';
        const uint32_t n = q.read_idx ? q.n : q.reads->n;
        if (q.read_idx) for (uint32_t i = 0; i < n; ++i) if (q.read_idx[i] >= q.reads->n) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: read index out of range");
    }
    SP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int32_t rc = SP_OK;
    const uint32_t big = 4096;                               // problems per launch sequence once the descriptors live in device memory
    for (uint32_t at = 0; at < n_prob; at += big) {
        const uint32_t k = std::min<uint32_t>(big, n_prob - at);
        const int32_t e = k <= 4 ? run_chunk<4>(ctx, k, probs + at, outs + at) : k <= 8 ? run_chunk<8>(ctx, k, probs + at, outs + at)
                        : k <= CMAXP ? run_chunk<CMAXP>(ctx, k, probs + at, outs + at) : run_chunk<0>(ctx, k, probs + at, outs + at);
        if (e != SP_OK && e != SP_ERR_CAPACITY) return e;
        if (e != SP_OK) rc = e;
    }
    return rc;
}

extern "C" {

int32_t sp_consensus_batch(sp_ctx* ctx, uint32_t n_problems, const sp_cons_problem* problems, sp_cons_output* outputs) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (n_problems == 0) return SP_OK;
    if (!problems || !outputs) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_batch: null argument");
    return run_batch(ctx, n_problems, problems, outputs);
}

int32_t sp_consensus_dual_batch(sp_ctx* ctx, uint32_t n_problems, const sp_cons_problem* problems, sp_cons_output* outputs) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (n_problems == 0) return SP_OK;
    if (!problems || !outputs) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_dual_batch: null argument");
    // The two passes of the policy run side by side: next to the pass that never splits (it finds the strongest second-base
    // column) a speculative pass splits at the first column that reaches cfg.min_af.  No earlier column can reach the final
    // threshold X >= cfg.min_af either, so if that column also reaches X the speculative pass IS the second pass; only otherwise
    // the second pass is run on its own.
    std::vector<sp_cons_problem> pr(2 * (size_t)n_problems); std::vector<sp_cons_output> out(2 * (size_t)n_problems);
    std::vector<std::vector<char>> text(n_problems); std::vector<std::vector<uint8_t>> fl(n_problems); std::vector<std::vector<int32_t>> sc(n_problems);
    for (uint32_t p = 0; p < n_problems; ++p) {
        const uint32_t n = problems[p].read_idx ? problems[p].n : (problems[p].reads ? problems[p].reads->n : 0);
        text[p].assign((size_t)2 * std::max<uint32_t>(outputs[p].cap, 1), 0); fl[p].assign(std::max<uint32_t>(n, 1), 0); sc[p].assign((size_t)2 * std::max<uint32_t>(n, 1), 0);
        pr[2 * p] = problems[p]; pr[2 * p].cfg.allow_dual = 0;
        out[2 * p] = outputs[p];
        out[2 * p].cons1 = text[p].data(); out[2 * p].cons2 = text[p].data() + outputs[p].cap; out[2 * p].is_cons1 = fl[p].data();
        out[2 * p].score1 = sc[p].data(); out[2 * p].score2 = sc[p].data() + std::max<uint32_t>(n, 1);
        pr[2 * p + 1] = problems[p]; pr[2 * p + 1].cfg.allow_dual = 1;
        out[2 * p + 1] = outputs[p];
    }
    int32_t rc = run_batch(ctx, 2 * n_problems, pr.data(), out.data());
    if (rc != SP_OK && rc != SP_ERR_CAPACITY) return rc;
    std::vector<sp_cons_problem> again; std::vector<sp_cons_output> outs2; std::vector<uint32_t> who;
    for (uint32_t p = 0; p < n_problems; ++p) {
        const sp_cons_result& single = out[2 * p].result; const sp_cons_result& spec = out[2 * p + 1].result;
        outputs[p] = out[2 * p + 1];
        if (single.best_w2 == 0 || !spec.is_dual) continue;            // nothing to split, or no column reaches even cfg.min_af: the speculative pass never split
        const double strongest = 0.5 * (double)single.best_w2 / (double)single.best_total;
        const double x = problems[p].cfg.min_af > strongest ? problems[p].cfg.min_af : strongest;
        if ((double)spec.split_w2 >= x * (double)spec.split_total) continue;
        sp_cons_problem q = problems[p]; q.cfg.allow_dual = 1; q.cfg.min_af = x;
        again.push_back(q); outs2.push_back(outputs[p]); who.push_back(p);
    }
    if (!again.empty()) {
        const int32_t rc2 = run_batch(ctx, (uint32_t)again.size(), again.data(), outs2.data());
        if (rc2 != SP_OK && rc2 != SP_ERR_CAPACITY) return rc2;
        for (size_t k = 0; k < who.size(); ++k) outputs[who[k]] = outs2[k];
    }
    rc = SP_OK;
    for (uint32_t p = 0; p < n_problems; ++p) if (outputs[p].status != SP_OK) rc = outputs[p].status;
    return rc;
}

int32_t sp_consensus(sp_ctx* ctx, const sp_seqset* reads, const uint32_t* read_idx, uint32_t n, const int32_t* offsets,
                     const sp_cons_config* cfg, char* cons1, char* cons2, uint32_t cap,
                     uint8_t* is_cons1, int32_t* score1, int32_t* score2, sp_cons_result* result) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!reads || !cfg || !result) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: null argument");
    sp_cons_problem q; q.reads = reads; q.read_idx = read_idx; q.n = n; q.offsets = offsets; q.cfg = *cfg;
    sp_cons_output o; std::memset(&o, 0, sizeof o);
    o.cons1 = cons1; o.cons2 = cons2; o.cap = cap; o.is_cons1 = is_cons1; o.score1 = score1; o.score2 = score2;
    const int32_t rc = run_batch(ctx, 1, &q, &o);
    *result = o.result;
    return rc;
}

int32_t sp_consensus_dual(sp_ctx* ctx, const sp_seqset* reads, const uint32_t* read_idx, uint32_t n, const int32_t* offsets,
                          const sp_cons_config* cfg, char* cons1, char* cons2, uint32_t cap,
                          uint8_t* is_cons1, int32_t* score1, int32_t* score2, sp_cons_result* result) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!reads || !cfg || !result) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_dual: null argument");
    sp_cons_problem q; q.reads = reads; q.read_idx = read_idx; q.n = n; q.offsets = offsets; q.cfg = *cfg;
    sp_cons_output o; std::memset(&o, 0, sizeof o);
    o.cons1 = cons1; o.cons2 = cons2; o.cap = cap; o.is_cons1 = is_cons1; o.score1 = score1; o.score2 = score2;
    const int32_t rc = sp_consensus_dual_batch(ctx, 1, &q, &o);
    *result = o.result;
    return rc;
}

int32_t sp_consensus_priority(sp_ctx* ctx, const sp_priority_problem* pr, uint32_t max_groups, uint32_t cap,
                              uint32_t* n_groups, int32_t* group_of, char* cons) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!pr || !n_groups || !group_of || !cons || !pr->levels || pr->n_levels == 0 || cap < 2) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_priority: null argument");
    for (uint32_t l = 0; l < pr->n_levels; ++l) if (!pr->levels[l] || pr->levels[l]->n != pr->n) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_priority: every level needs one sequence per read");
    *n_groups = 0;
    const uint32_t n = pr->n, NL = pr->n_levels;
    if (n == 0) return SP_OK;
    const int half = pr->cfg.offset_window / 2;
    struct Item { std::vector<uint32_t> members; uint32_t level; std::string key; };
    // initial groups: unseeded reads first, then the seeds in ascending order
    std::vector<Item> work, done;
    {
        std::map<int32_t, std::vector<uint32_t>> by_seed;
        for (uint32_t r = 0; r < n; ++r) by_seed[pr->seeds ? (pr->seeds[r] < 0 ? -1 : pr->seeds[r]) : -1].push_back(r);
        uint32_t ord = 0;
        for (auto& kv : by_seed) { Item it; it.members = kv.second; it.level = 0; it.key = std::string(1, (char)('a' + std::min<uint32_t>(ord, 25))) + std::to_string(ord); ++ord; work.push_back(std::move(it)); }
    }
    auto rebased = [&](const std::vector<uint32_t>& m, uint32_t level, std::vector<int32_t>& out) -> bool {
        const int32_t* src = pr->offsets ? pr->offsets[level] : nullptr;
        if (!src) return false;
        int64_t mn = INT64_MAX;
        for (uint32_t r : m) mn = std::min<int64_t>(mn, src[r] < 0 ? 0 : src[r]);
        out.resize(m.size());
        for (size_t i = 0; i < m.size(); ++i) { const int64_t o = src[m[i]] < 0 ? 0 : src[m[i]]; out[i] = o == mn ? -1 : (int32_t)(o - mn + (mn == 0 ? 0 : half)); }
        return true;
    };
    while (!work.empty()) {
        // one round: every open group as one two-way problem, all in lockstep
        const size_t k = work.size();
        std::vector<sp_cons_problem> P(k); std::vector<sp_cons_output> O(k);
        std::vector<std::vector<int32_t>> offs(k), s1(k), s2(k); std::vector<std::vector<uint8_t>> is1(k); std::vector<std::vector<char>> text(k);
        for (size_t x = 0; x < k; ++x) {
            const Item& it = work[x];
            const sp_seqset* set = pr->levels[it.level];
            int32_t longest = 0; for (uint32_t r : it.members) longest = std::max(longest, set->h_len[r]);
            const bool has_off = rebased(it.members, it.level, offs[x]);
            int32_t far = 0; if (has_off) for (int32_t o : offs[x]) far = std::max(far, o);
            const uint32_t c = (uint32_t)longest + (uint32_t)far + 66;
            s1[x].resize(it.members.size()); s2[x].resize(it.members.size()); is1[x].resize(it.members.size()); text[x].assign((size_t)2 * c, 0);
            P[x].reads = set; P[x].read_idx = it.members.data(); P[x].n = (uint32_t)it.members.size(); P[x].offsets = has_off ? offs[x].data() : nullptr;
            P[x].cfg = pr->cfg; P[x].cfg.allow_dual = 1;
            std::memset(&O[x], 0, sizeof O[x]);
            O[x].cons1 = text[x].data(); O[x].cons2 = text[x].data() + c; O[x].cap = c; O[x].is_cons1 = is1[x].data(); O[x].score1 = s1[x].data(); O[x].score2 = s2[x].data();
        }
        const int32_t rc = sp_consensus_dual_batch(ctx, (uint32_t)k, P.data(), O.data());
        if (rc != SP_OK) return rc;
        std::vector<Item> next;
        for (size_t x = 0; x < k; ++x) {
            Item& it = work[x];
            std::vector<uint32_t> g1, g2;
            for (size_t i = 0; i < it.members.size(); ++i) (is1[x][i] ? g1 : g2).push_back(it.members[i]);
            if (O[x].result.is_dual && !g1.empty() && !g2.empty()) {
                Item a; a.members = std::move(g1); a.level = it.level; a.key = it.key + "0"; next.push_back(std::move(a));
                Item b; b.members = std::move(g2); b.level = it.level; b.key = it.key + "1"; next.push_back(std::move(b));
            } else if (it.level + 1 < NL) { it.level += 1; it.key += "_"; next.push_back(std::move(it)); }
            else done.push_back(std::move(it));
        }
        work.swap(next);
        if (done.size() + work.size() > (size_t)n) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_priority: more groups than reads");
    }
    std::sort(done.begin(), done.end(), [](const Item& a, const Item& b) { return a.key < b.key; });
    *n_groups = (uint32_t)done.size();
    for (size_t g = 0; g < done.size(); ++g) for (uint32_t r : done[g].members) group_of[r] = (int32_t)g;
    if (done.size() > max_groups) return sp_fail(ctx, SP_ERR_CAPACITY, "sp_consensus_priority: more groups than max_groups");
    // one consensus per emitted group and level
    const size_t k = done.size() * NL;
    std::vector<sp_cons_problem> P(k); std::vector<sp_cons_output> O(k);
    std::vector<std::vector<int32_t>> offs(k), s1(k), s2(k); std::vector<std::vector<uint8_t>> is1(k); std::vector<std::vector<char>> spare(k);
    for (size_t g = 0; g < done.size(); ++g) for (uint32_t l = 0; l < NL; ++l) {
        const size_t x = g * NL + l; const Item& it = done[g];
        const bool has_off = rebased(it.members, l, offs[x]);
        s1[x].resize(it.members.size()); s2[x].resize(it.members.size()); is1[x].resize(it.members.size()); spare[x].assign(cap, 0);
        P[x].reads = pr->levels[l]; P[x].read_idx = it.members.data(); P[x].n = (uint32_t)it.members.size(); P[x].offsets = has_off ? offs[x].data() : nullptr;
        P[x].cfg = pr->cfg; P[x].cfg.allow_dual = 0;
        std::memset(&O[x], 0, sizeof O[x]);
        O[x].cons1 = cons + x * (size_t)cap; O[x].cons2 = spare[x].data(); O[x].cap = cap; O[x].is_cons1 = is1[x].data(); O[x].score1 = s1[x].data(); O[x].score2 = s2[x].data();
    }
    return sp_consensus_batch(ctx, (uint32_t)k, P.data(), O.data());
}

} // extern "C"
