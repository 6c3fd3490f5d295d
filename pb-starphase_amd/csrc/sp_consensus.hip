// sp_consensus.hip -- K8: read consensus by dynamic wavefront alignment (gfx950).
//
// Serves the waffle_con call sites of the reference: DualConsensusDWFA in run_dual_consensus_with_offsets
// (src/hla/caller.rs:1103-1219) and the per-group ConsensusDWFA (src/hla/caller.rs:706-747), configured as
// dwfa_config_from_cli does (src/hla/caller.rs:1103-1116).  waffle_con itself (v0.4.4) is a third-party crate that is not
// on disk, so the contract is the one stated in DESIGN.md section 9 and restated for the CPU in oracle/consensus.c; the
// two agree bit for bit (consensus strings, read assignment, per-read edit counts).
//
// Mapping: ONE wavefront per read, one lane per diagonal (64-diagonal band, lane l <-> consensus pos - read pos = l - 32),
// the per-read state is one VGPR per lane (furthest read position at the current edit count).  The consensus grows by one
// base per kernel launch: every wave first re-derives the (wave-uniform) decision for position t from the vote counters
// the previous launch accumulated, pushes that base into its read's wavefront(s), and adds the read's vote for position
// t+1.  There is no host round trip inside the loop: the launches are enqueued back to back and become no-ops once the
// consensus has stopped; the control block is double buffered by launch parity so a launch never reads what it writes.
#include "sp_internal.h"
#include "sp_wfa.cuh"
#include <algorithm>
#include <cstring>
#include <map>
#include <string>

namespace {

constexpr int CB = 64;          // band
constexpr int CH = 32;          // lane of diagonal 0
constexpr int CWAVES = 16;      // reads per workgroup
constexpr int CSLOTS = 8;       // vote counters are spread over this many slots (blockIdx & 7), one 128-byte line each:
                                // same-line device atomics serialise (profiles/microbench/step_latency.hip)
constexpr int CSTRIDE = 32;     // words per slot
enum { F_ACTIVE = 1, F_FINISHED = 2, F_LOST = 4 };

struct ConsCtrl {               // state before a step
    int32_t dual, split_at, stopped[2], len[2], done, pad;
    long long best_w2, best_total;
    long long split_w2, split_total;     // the votes of the column at which the second consensus was split off
};
struct ConsMeta { int32_t e, c0, flags, pad; };

// One consensus problem of a batch.  All problems of a batch advance in lockstep, one base per launch; a workgroup belongs to
// exactly one problem (each problem's reads are padded to a multiple of CWAVES in the flattened read order).  The descriptors
// travel in the kernel argument block (scalar loads), per-read constants are gathered once into ReadInfo: a step is a chain
// of only two dependent memory round trips (state + control, then the read words at the wavefront tips).
constexpr int CMAXP = 32;       // problems per launch sequence: their descriptors travel in the kernel argument block (32 x 80 bytes);
                                // batches of up to 8 use an 8-entry block, whose launches are ~2 us shorter
struct ConsParams {
    int n, first, first_block;  // reads; flattened index of local read 0; first workgroup of the problem
    int min_count, delta, et, allow_dual, window, cmp_len; double min_af;
    uint8_t* C; int cap;        // [2][cap] base codes; consensus 2 shares [0, split_at) with consensus 1
    uint32_t* votes;            // [2][cap+1][CSLOTS][8] : w[4], end
    ConsCtrl* ctrl;             // [2]
};
struct ReadInfo { const uint32_t* w; const uint32_t* np; int n, off; long long pad; };
template <int MAXP> struct ConsBatchT {
    ConsParams p[MAXP]; int n_prob;
    const ReadInfo* info;       // [total]
    uint16_t* H;                // [2][total][64] furthest read position per diagonal (0xFFFF = none): 128 bytes per read and consensus
    ConsMeta* meta;             // [2][total]
    int total;
};
// large batches (a cohort): the descriptors live in device memory and a table maps every workgroup to its problem
template <> struct ConsBatchT<0> {
    const ConsParams* p; const int* block_prob; int n_prob;
    const ReadInfo* info; uint16_t* H; ConsMeta* meta; int total;
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

struct Decision { int go[2]; int base[2]; int split; long long best_w2, best_total, split_w2, split_total; };

// Every wave re-derives the decision for position t.  Lanes 0-15 fetch the 2 x CSLOTS vote slots (one 128-byte line each) and the
// sums are formed with three shuffle steps, so the whole decision costs one memory round trip and a handful of registers.
__device__ __forceinline__ Decision decide(const ConsParams& P, const ConsCtrl& c, int t, int lane) {
    Decision d; d.go[0] = d.go[1] = 0; d.base[0] = d.base[1] = 0; d.split = 0; d.best_w2 = c.best_w2; d.best_total = c.best_total;
    d.split_w2 = c.split_w2; d.split_total = c.split_total;
    if (t >= P.cap) return d;                                                      // out of room: the consensus is cut at cap
    uint32_t x[5] = { 0, 0, 0, 0, 0 };
    if (lane < 2 * CSLOTS) {
        const uint32_t* v = P.votes + (((size_t)(lane >> 3) * (P.cap + 1) + t) * CSLOTS + (lane & (CSLOTS - 1))) * CSTRIDE;
        const uint4 q = *reinterpret_cast<const uint4*>(v);
        x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w; x[4] = v[4];
    }
    // sums over each group of 8 lanes with DPP (quad swaps, then the half-row mirror): no LDS traffic, three adds per counter
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        x[j] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[j], 0xB1, 0xf, 0xf, true);      // quad_perm:[1,0,3,2]
        x[j] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[j], 0x4E, 0xf, 0xf, true);      // quad_perm:[2,3,0,1]
        x[j] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[j], 0x141, 0xf, 0xf, true);     // row_half_mirror
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && !c.dual) continue;
        if (c.stopped[i]) continue;
        // (32-bit counters: a column holds at most 12 units per read; only the cross products below need 64 bits, and the scalar
        // unit has no ordered 64-bit compare -- with wider types every comparison here becomes a vector instruction)
        uint32_t w[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) w[b] = (uint32_t)__builtin_amdgcn_readlane((int)x[b], i * CSLOTS);
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)x[4], i * CSLOTS);
        const uint32_t total = w[0] + w[1] + w[2] + w[3];
        int b1 = 0; uint32_t w1 = w[0];                                            // heaviest base, ties to the lower code
#pragma unroll
        for (int b = 1; b < 4; ++b) if (w[b] > w1) { b1 = b; w1 = w[b]; }
        int b2 = -1; uint32_t w2 = 0; bool have2 = false;                          // heaviest of the others, ties to the lower code
#pragma unroll
        for (int b = 0; b < 4; ++b) if (b != b1 && (!have2 || w[b] > w2)) { b2 = b; w2 = w[b]; have2 = true; }
        const bool go = P.et ? w1 > 0 : (total > end && w1 > 0);
        if (!go) continue;
        d.go[i] = 1; d.base[i] = b1;
        if (!c.dual && w2 >= 12u * (uint32_t)P.min_count) {
            if ((unsigned long long)w2 * (unsigned long long)d.best_total > (unsigned long long)d.best_w2 * (unsigned long long)total) { d.best_w2 = w2; d.best_total = total; }
            if (P.allow_dual && (double)w2 >= P.min_af * (double)total) { d.split = 1; d.go[1] = 1; d.base[1] = b2; d.split_w2 = w2; d.split_total = total; }
        }
    }
    return d;
}

// consensus i at absolute position pos as seen by launch t (position t itself is this launch's decision, not yet in memory)
struct ConsView {
    const uint8_t* C; int cap, split_at, t; int base0, base1;
    __device__ __forceinline__ int at(int i, int pos) const {
        if (pos == t) return i ? base1 : base0;
        return (i == 1 && pos < split_at) ? C[pos] : C[(size_t)i * cap + pos];
    }
};

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

__device__ __forceinline__ void dwfa_push(Dwfa& d, const ReadView& rv, const ConsView& cv, int i, int T, int nb, int et, int lane) {
    dwfa_push_t(d, rv.n, [&](int h) { return read_base(rv, h); }, [&](int pos) { return cv.at(i, pos); }, T, nb, et, lane);
}

// placement of a late read: Sellers' search of its first L bases in the last W consensus bases, one Myers bit-vector scan per
// lane over the end positions it owns (an occurrence of an L-base pattern with <= L edits spans <= 2L text bases)
template <class CA>
__device__ __forceinline__ int find_start(const ReadView& rv, CA ca, int off, int W, int L, int lane) {
    const int ws = off - W > 0 ? off - W : 0, M = off - ws;
    if (L > rv.n) L = rv.n;
    if (M <= 0 || L <= 0) return off;
    unsigned long long peq[4];
    {
        const int code = lane < L ? read_base(rv, L - 1 - lane) : 7;
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

template <int MAXP>
__global__ void __launch_bounds__(CWAVES * SP_WAVE) cons_step_kernel(ConsBatchT<MAXP> B, int t) {
    __shared__ uint32_t lv[2][8];
    int pi = 0;
    if constexpr (MAXP == 0) pi = B.block_prob[blockIdx.x];
    else {
#pragma unroll
        for (int i = 1; i < MAXP; ++i) if (i < B.n_prob && (int)blockIdx.x >= B.p[i].first_block) pi = i;
    }
    const ConsParams P = B.p[pi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lead = (int)blockIdx.x == P.first_block;
    const int r = ((int)blockIdx.x - P.first_block) * CWAVES + wave;
    const size_t g = (size_t)P.first + r;                     // slot in the flattened state arrays (padding slots exist in memory)
    // everything this wave will need from memory that does not depend on the decision is requested up front
    const ReadInfo ri = B.info[g];
    const bool second = P.allow_dual != 0;                   // problems that cannot split never touch the second state
    const ConsMeta m0 = B.meta[g];
    ConsMeta m1; m1.e = 0; m1.c0 = 0; m1.flags = 0; m1.pad = 0;
    if (second) m1 = B.meta[(size_t)B.total + g];
    const int h0 = h_load(B.H, g * CB + lane), h1 = second ? h_load(B.H, ((size_t)B.total + g) * CB + lane) : SP_NEG;
    ConsCtrl cin;
    Decision dec;
    if (t >= 0) {
        cin = P.ctrl[t & 1];
        if (cin.done) { if (lead && threadIdx.x == 0) P.ctrl[(t + 1) & 1] = cin; return; }
        dec = decide(P, cin, t, lane);
    } else {
        cin = P.ctrl[0];
        dec.go[0] = dec.go[1] = 0; dec.base[0] = dec.base[1] = 0; dec.split = 0; dec.best_w2 = 0; dec.best_total = 1; dec.split_w2 = 0; dec.split_total = 1;
    }
    const int dual = (t >= 0) && (cin.dual || dec.split);
    const int split_at = dec.split ? t : cin.split_at;
    if (threadIdx.x < 16) lv[threadIdx.x >> 3][threadIdx.x & 7] = 0;
    if (t >= 0 && lead && threadIdx.x == 0) {
        ConsCtrl co = cin;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (dec.go[i]) { P.C[(size_t)i * P.cap + t] = (uint8_t)dec.base[i]; co.len[i] = t + 1; co.stopped[i] = 0; }
            else if (i == 0 || cin.dual) co.stopped[i] = 1;
        }
        co.dual = dual; co.split_at = split_at; co.best_w2 = dec.best_w2; co.best_total = dec.best_total;
        co.split_w2 = dec.split_w2; co.split_total = dec.split_total;
        co.done = !(dec.go[0] || dec.go[1]);
        P.ctrl[(t + 1) & 1] = co;
    }
    __syncthreads();
    if (r < P.n) {
        ReadView rv; rv.w = ri.w; rv.np = ri.np; rv.n = ri.n;
        ConsView cv; cv.C = P.C; cv.cap = P.cap; cv.split_at = split_at; cv.t = t; cv.base0 = dec.base[0]; cv.base1 = dec.base[1];
        Dwfa d[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { d[i].H = SP_NEG; d[i].e = 0; d[i].c0 = 0; d[i].flags = 0; }
        if (t < 0) {
            if (ri.off < 0) { d[0].flags = F_ACTIVE | ((P.et && rv.n == 0) ? F_FINISHED : 0); d[0].H = lane == CH ? 0 : SP_NEG; }
        } else {
            d[0].H = h0; d[0].e = m0.e; d[0].c0 = m0.c0; d[0].flags = m0.flags;
            if (dec.split) d[1] = d[0];
            else if (cin.dual) { d[1].H = h1; d[1].e = m1.e; d[1].c0 = m1.c0; d[1].flags = m1.flags; }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !dual) continue;
                if (!dec.go[i]) continue;
                const int len = t + 1;
                // (a read whose offset equals len is placed by cons_activate_kernel, launched right after this step)
                if ((d[i].flags & F_ACTIVE) && !(d[i].flags & (F_FINISHED | F_LOST))) dwfa_push(d[i], rv, cv, i, len - d[i].c0, dec.base[i], P.et, lane);
            }
            if (dual) {
                const int both = (d[0].flags & F_ACTIVE) && (d[1].flags & F_ACTIVE) && !(d[0].flags & F_LOST) && !(d[1].flags & F_LOST);
                if (both) {
                    if (d[0].e > d[1].e + P.delta) d[0].flags |= F_LOST;
                    else if (d[1].e > d[0].e + P.delta) d[1].flags |= F_LOST;
                }
            }
        }
        // votes for position t+1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !dual) continue;
            if (t >= 0 && !dec.go[i]) continue;
            if (!(d[i].flags & F_ACTIVE) || (d[i].flags & (F_FINISHED | F_LOST))) continue;
            if (dual) { const Dwfa& o = d[1 - i]; if ((o.flags & F_ACTIVE) && !(o.flags & F_LOST) && o.e < d[i].e) continue; }
            const int T = t + 1 - d[i].c0, k = lane - CH;
            const bool tip = d[i].H >= 0 && d[i].H + k == T;
            const int code = (tip && d[i].H < rv.n) ? read_base(rv, d[i].H) : 5;
            int seen[4], dc = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) { seen[b] = __ballot(code == b) != 0; dc += seen[b]; }
            const bool ended = __ballot(tip) != 0 && __ballot(code == 4) == 0;      // every tip sits at the end of the read
            if (lane == 0) {
                if (dc) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) if (seen[b]) atomicAdd(&lv[i][b], vote_units(dc));
                }
                else if (ended) atomicAdd(&lv[i][4], 12u);
            }
        }
        // store
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !dual) continue;
            if (lane == 0) { ConsMeta m; m.e = d[i].e; m.c0 = d[i].c0; m.flags = d[i].flags; m.pad = 0; B.meta[(size_t)i * B.total + g] = m; }
            h_store(B.H, ((size_t)i * B.total + g) * CB + lane, d[i].H);
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int i = threadIdx.x >> 3, j = threadIdx.x & 7;
        const uint32_t v = lv[i][j];
        if (v && t + 1 <= P.cap) atomicAdd(P.votes + (((size_t)i * (P.cap + 1) + (t + 1)) * CSLOTS + (blockIdx.x & (CSLOTS - 1))) * CSTRIDE + j, v);
    }
}

// Late reads (add_sequence_offset): the reads whose offset equals the length the consensus reached in step t are placed by
// this kernel, launched right after that step (the host knows the offsets, so it knows when to launch it): start search in the
// window before the offset, catch-up pushes, dual bookkeeping and the read's vote for position t+1.
struct ActItem { int prob, r; };

constexpr int ACT_CONS = 512;   // consensus bases a wave keeps in LDS while a late read catches up (offset_window + slack)
constexpr int ACT_READ = 640;   // read bases it keeps (catch-up length + band + edits)

template <int MAXP>
__global__ void __launch_bounds__(4 * SP_WAVE) cons_activate_kernel(ConsBatchT<MAXP> B, int t, const ActItem* __restrict__ items, int n_items) {
    __shared__ uint8_t ccache[4][ACT_CONS];
    __shared__ uint8_t rcache[4][ACT_READ];
    // the same two windows 2 bits per base (16 bases per dword, two guard words): the catch-up compares 16 bases per step out of them
    __shared__ uint32_t cpack[4][ACT_CONS / 16 + 2];
    __shared__ uint32_t rpack[4][ACT_READ / 16 + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, it = blockIdx.x * 4 + wv;
    if (it >= n_items) return;
    const ActItem item = items[it];
    int pi = 0;
    if constexpr (MAXP == 0) pi = item.prob;
    else {
#pragma unroll
        for (int i = 1; i < MAXP; ++i) if (i == item.prob) pi = i;
    }
    const ConsParams P = B.p[pi];
    const size_t g = (size_t)P.first + item.r;
    const ReadInfo ri = B.info[g];
    const ConsCtrl c = P.ctrl[(t + 1) & 1];                    // state after step t
    ReadView rv; rv.w = ri.w; rv.np = ri.np; rv.n = ri.n;
    ConsView cv; cv.C = P.C; cv.cap = P.cap; cv.split_at = c.split_at; cv.t = -1; cv.base0 = cv.base1 = 0;   // position t is in memory by now
    const int len = t + 1;
    Dwfa d[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const ConsMeta m = B.meta[(size_t)i * B.total + g];
        d[i].H = h_load(B.H, ((size_t)i * B.total + g) * CB + lane); d[i].e = m.e; d[i].c0 = m.c0; d[i].flags = (i == 1 && !c.dual) ? 0 : m.flags;
    }
    int placed[2] = { 0, 0 };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && !c.dual) continue;
        if (c.len[i] != len || c.stopped[i] || (d[i].flags & F_ACTIVE) || ri.off != len) continue;
        placed[i] = 1;
        // the window in front of the offset and the head of the read go to LDS once: both the start search and the catch-up run out of it
        const int ws = ri.off - P.window > 0 ? ri.off - P.window : 0;
        for (int x = lane; x < len - ws && x < ACT_CONS; x += SP_WAVE) ccache[wv][x] = (uint8_t)cv.at(i, ws + x);
        for (int x = lane; x < rv.n && x < ACT_READ; x += SP_WAVE) rcache[wv][x] = (uint8_t)read_base(rv, x);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // packed copies: the consensus window from its byte codes (a lane packs 16 of them), the read words straight from memory
        const int cwin = len - ws;
        const bool packed = rv.np == nullptr && cwin <= ACT_CONS;
        if (packed) {
            for (int w = lane; w < ACT_CONS / 16 + 2; w += SP_WAVE) {
                uint32_t word = 0;
                for (int b = 0; b < 16; ++b) { const int x = w * 16 + b; if (x < cwin) word |= (uint32_t)(ccache[wv][x] & 3u) << (b << 1); }
                cpack[wv][w] = word;
            }
            const int rwords = ((rv.n < ACT_READ ? rv.n : ACT_READ) + 15) >> 4;
            for (int w = lane; w < ACT_READ / 16 + 2; w += SP_WAVE) rpack[wv][w] = w < rwords ? rv.w[w] : 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        auto rb = [&](int h) { return h < ACT_READ ? (int)rcache[wv][h] : read_base(rv, h); };
        auto ca = [&](int pos) { const int x = pos - ws; return (x >= 0 && x < ACT_CONS) ? (int)ccache[wv][x] : cv.at(i, pos); };
        ReadView rvc = rv;
        d[i].c0 = find_start(rvc, ca, ri.off, P.window, P.cmp_len, lane);
        d[i].H = lane == CH ? 0 : SP_NEG; d[i].e = 0; d[i].flags = F_ACTIVE | ((P.et && rv.n == 0) ? F_FINISHED : 0);
        const int c0 = d[i].c0, span = len - c0;
        if (span > 0 && !(d[i].flags & F_FINISHED) && !(P.et && rv.n <= span + CB)) {
            const int kk = lane - CH;
            if (packed) {
                // 16 bases per step: xor of the two funnel-shifted words, first differing base, clamped by what is left of the read and
                // of the span (H <= span + 32 < ACT_READ - 16 and the consensus window starts at or before c0: every word is cached)
                const int cbase = c0 - ws;
                dwfa_catchup_t(d[i], rv.n, [&]() {
                    for (;;) {
                        Dwfa& q = d[i];
                        int left = rv.n - q.H; { const int l2 = span - (q.H + kk); left = l2 < left ? l2 : left; }
                        const bool go = q.H >= 0 && left > 0;
                        int nm = 0;
                        if (go) {
                            const int pr = q.H, pc = cbase + q.H + kk;
                            const uint32_t a = __builtin_amdgcn_alignbit(rpack[wv][(pr >> 4) + 1], rpack[wv][pr >> 4], (uint32_t)(pr & 15) << 1);
                            const uint32_t b = __builtin_amdgcn_alignbit(cpack[wv][(pc >> 4) + 1], cpack[wv][pc >> 4], (uint32_t)(pc & 15) << 1);
                            const uint32_t x = a ^ b, mm = (x | (x >> 1)) & 0x55555555u;
                            nm = mm ? (__builtin_ctz(mm) >> 1) : 16;
                            nm = nm < left ? nm : left;
                            q.H += nm;
                        }
                        if (!__ballot(go && nm == 16 && left > 16)) break;
                    }
                }, span, lane);
            } else {
                dwfa_catchup_t(d[i], rv.n, [&]() {
                    for (;;) {
                        Dwfa& q = d[i];
                        bool go = q.H >= 0 && q.H < rv.n && q.H + kk < span;
                        if (go) { const int x = rb(q.H); go = x < 4 && x == ca(q.c0 + q.H + kk); }
                        if (!__ballot(go)) break;
                        if (go) q.H += 1;
                    }
                }, span, lane);
            }
        } else {
            for (int T = 1; T <= span; ++T) {
                if (d[i].flags & (F_FINISHED | F_LOST)) break;
                dwfa_push_t(d[i], rv.n, rb, ca, T, ca(c0 + T - 1), P.et, lane);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!placed[0] && !placed[1]) return;
    if (c.dual) {
        const int both = (d[0].flags & F_ACTIVE) && (d[1].flags & F_ACTIVE) && !(d[0].flags & F_LOST) && !(d[1].flags & F_LOST);
        if (both) {
            if (d[0].e > d[1].e + P.delta) d[0].flags |= F_LOST;
            else if (d[1].e > d[0].e + P.delta) d[1].flags |= F_LOST;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!placed[i]) continue;
        if (lane == 0) { ConsMeta m; m.e = d[i].e; m.c0 = d[i].c0; m.flags = d[i].flags; m.pad = 0; B.meta[(size_t)i * B.total + g] = m; }
        h_store(B.H, ((size_t)i * B.total + g) * CB + lane, d[i].H);
        if (d[i].flags & (F_FINISHED | F_LOST)) continue;
        if (c.dual) { const Dwfa& o = d[1 - i]; if ((o.flags & F_ACTIVE) && !(o.flags & F_LOST) && o.e < d[i].e) continue; }
        const int T = len - d[i].c0, k = lane - CH;
        const bool tip = d[i].H >= 0 && d[i].H + k == T;
        const int code = (tip && d[i].H < rv.n) ? read_base(rv, d[i].H) : 5;
        int seen[4], dc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) { seen[b] = __ballot(code == b) != 0; dc += seen[b]; }
        const bool ended = __ballot(tip) != 0 && __ballot(code == 4) == 0;
        if (lane == 0 && len <= P.cap) {
            uint32_t* v = P.votes + (((size_t)i * (P.cap + 1) + len) * CSLOTS + (blockIdx.x & (CSLOTS - 1))) * CSTRIDE;
            if (dc) {
#pragma unroll
                for (int b = 0; b < 4; ++b) if (seen[b]) atomicAdd(v + b, vote_units(dc));
            } else if (ended) atomicAdd(v + 4, 12u);
        }
    }
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
__global__ void __launch_bounds__(CWAVES * SP_WAVE) cons_finalize_kernel(ConsBatchT<MAXP> B, int which, uint8_t* is_cons1, int32_t* score1, int32_t* score2) {
    int pi = 0;
    if constexpr (MAXP == 0) pi = B.block_prob[blockIdx.x];
    else {
#pragma unroll
        for (int i = 1; i < MAXP; ++i) if (i < B.n_prob && (int)blockIdx.x >= B.p[i].first_block) pi = i;
    }
    const ConsParams P = B.p[pi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = ((int)blockIdx.x - P.first_block) * CWAVES + wave;
    if (r >= P.n) return;
    const size_t g = (size_t)P.first + r;
    const ConsCtrl c = P.ctrl[which];
    const int n = B.info[g].n;
    int sc[2] = { -1, -1 };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && !c.dual) continue;
        const ConsMeta m = B.meta[(size_t)i * B.total + g];
        if (!(m.flags & F_ACTIVE) || (m.flags & F_LOST)) continue;
        int e = m.e;
        if (!P.et) {
            const int h = h_load(B.H, ((size_t)i * B.total + g) * CB + lane), k = lane - CH;
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

} // namespace

// host side of a batch of at most CMAXP problems: all of them advance one base per launch until every one has stopped
template <int MAXP>
static int32_t run_chunk(sp_ctx* ctx, uint32_t n_prob, const sp_cons_problem* probs, sp_cons_output* outs) {
    hipStream_t st = ctx->stream;
    ConsBatchT<MAXP> B; std::memset(&B, 0, sizeof B);
    B.n_prob = (int)n_prob;
    std::vector<ConsParams> hp(n_prob);                      // the descriptors; they end up in the kernel arguments or, for MAXP == 0, in device memory
    std::vector<int> block_prob;
    std::vector<ConsSetup> setup(n_prob);
    std::vector<uint32_t> h_idx; std::vector<int32_t> h_off;
    std::vector<size_t> idx_at(n_prob), off_at(n_prob), c_at(n_prob), v_at(n_prob);
    size_t total = 0, c_bytes = 0, v_words = 0; int max_cap = 0, n_blocks = 0;
    for (uint32_t p = 0; p < n_prob; ++p) {
        const sp_cons_problem& q = probs[p];
        const uint32_t n = q.read_idx ? q.n : q.reads->n;
        ConsParams& P = hp[p];
        P.n = (int)n; P.cap = (int)outs[p].cap - 1;                                   // one byte of the caller's buffer is the NUL
        P.first = (int)total; P.first_block = n_blocks;
        P.min_count = q.cfg.min_count; P.delta = q.cfg.dual_max_ed_delta; P.et = q.cfg.allow_early_termination != 0; P.allow_dual = q.cfg.allow_dual != 0;
        P.window = q.cfg.offset_window; P.cmp_len = q.cfg.offset_compare_length; P.min_af = q.cfg.min_af;
        const uint32_t nb = (n + CWAVES - 1) / CWAVES;
        n_blocks += (int)nb;
        if (MAXP == 0) block_prob.insert(block_prob.end(), nb, (int)p);
        setup[p].reads = q.reads->view(); setup[p].n = (int)n; setup[p].first = (int)total;
        total += (size_t)nb * CWAVES;
        idx_at[p] = h_idx.size(); if (q.read_idx) h_idx.insert(h_idx.end(), q.read_idx, q.read_idx + n);
        off_at[p] = h_off.size(); if (q.offsets) h_off.insert(h_off.end(), q.offsets, q.offsets + n);
        c_at[p] = c_bytes; c_bytes += 2 * (size_t)std::max(P.cap, 1);
        v_at[p] = v_words; v_words += (size_t)2 * (P.cap + 1) * CSLOTS * CSTRIDE;
        max_cap = std::max(max_cap, P.cap);
    }
    if (n_blocks == 0) return SP_OK;
    // late reads, ordered by the step that places them
    std::vector<std::pair<int, ActItem>> late;
    for (uint32_t p = 0; p < n_prob; ++p) if (probs[p].offsets)
        for (int r = 0; r < hp[p].n; ++r) if (probs[p].offsets[r] >= 1) late.push_back({ probs[p].offsets[r], ActItem{ (int)p, r } });
    std::stable_sort(late.begin(), late.end(), [](const std::pair<int, ActItem>& a, const std::pair<int, ActItem>& b) { return a.first < b.first; });
    std::vector<ActItem> h_act(late.size());
    for (size_t i = 0; i < late.size(); ++i) h_act[i] = late[i].second;
    ActItem* d_act = (ActItem*)sp_pool(ctx, "cons_act", sizeof(ActItem) * std::max<size_t>(1, h_act.size()));
    if (!d_act) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus buffers");
    uint32_t* d_idx = (uint32_t*)sp_pool(ctx, "cons_idx", sizeof(uint32_t) * std::max<size_t>(1, h_idx.size()));
    int32_t* d_off = (int32_t*)sp_pool(ctx, "cons_off", sizeof(int32_t) * std::max<size_t>(1, h_off.size()));
    uint8_t* d_C = (uint8_t*)sp_pool(ctx, "cons_C", c_bytes);
    uint32_t* d_votes = (uint32_t*)sp_pool(ctx, "cons_votes", sizeof(uint32_t) * v_words);
    ConsCtrl* d_ctrl = (ConsCtrl*)sp_pool(ctx, "cons_ctrl", sizeof(ConsCtrl) * 2 * n_prob);
    ReadInfo* d_info = (ReadInfo*)sp_pool(ctx, "cons_info", sizeof(ReadInfo) * total);
    B.info = d_info; B.total = (int)total;
    B.H = (uint16_t*)sp_pool(ctx, "cons_H", sizeof(uint16_t) * 2 * total * CB);
    B.meta = (ConsMeta*)sp_pool(ctx, "cons_meta", sizeof(ConsMeta) * 2 * total);
    uint8_t* d_is1 = (uint8_t*)sp_pool(ctx, "cons_is1", total);
    int32_t* d_sc = (int32_t*)sp_pool(ctx, "cons_scores", sizeof(int32_t) * 2 * total);
    if (!d_idx || !d_off || !d_C || !d_votes || !d_ctrl || !d_info || !B.H || !B.meta || !d_is1 || !d_sc)
        return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus buffers");
    for (uint32_t p = 0; p < n_prob; ++p) {
        setup[p].idx = probs[p].read_idx ? d_idx + idx_at[p] : nullptr;
        setup[p].offsets = probs[p].offsets ? d_off + off_at[p] : nullptr;
        hp[p].C = d_C + c_at[p]; hp[p].votes = d_votes + v_at[p]; hp[p].ctrl = d_ctrl + 2 * p;
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
    ConsCtrl c0; std::memset(&c0, 0, sizeof c0); c0.split_at = -1; c0.stopped[1] = 1; c0.best_total = 1; c0.split_total = 1;
    std::vector<ConsCtrl> h_ctrl(2 * (size_t)n_prob, c0);
    if (!h_idx.empty()) SP_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, h_idx.data(), sizeof(uint32_t) * h_idx.size(), hipMemcpyHostToDevice, st));
    if (!h_off.empty()) SP_HIP_CHECK(ctx, hipMemcpyAsync(d_off, h_off.data(), sizeof(int32_t) * h_off.size(), hipMemcpyHostToDevice, st));
    if (!h_act.empty()) SP_HIP_CHECK(ctx, hipMemcpyAsync(d_act, h_act.data(), sizeof(ActItem) * h_act.size(), hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_ctrl, h_ctrl.data(), sizeof(ConsCtrl) * h_ctrl.size(), hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemsetAsync(d_votes, 0, sizeof(uint32_t) * v_words, st));
    SP_HIP_CHECK(ctx, hipMemsetAsync(B.meta, 0, sizeof(ConsMeta) * 2 * total, st));
    SP_HIP_CHECK(ctx, hipMemsetAsync(d_info, 0, sizeof(ReadInfo) * total, st));
    for (uint32_t p = 0; p < n_prob; ++p)
        if (setup[p].n) hipLaunchKernelGGL(cons_setup_kernel, dim3((setup[p].n + 255) / 256), dim3(256), 0, st, setup[p], d_info);
    SP_HIP_CHECK(ctx, hipStreamSynchronize(st));      // the pageable host sources above must stay valid until copied

    const dim3 grid((uint32_t)n_blocks), block(CWAVES * SP_WAVE);
    int last = -1;
    size_t act_at = 0;
    {
        ProfScope ps(ctx, "cons_steps", total);
        hipLaunchKernelGGL(cons_step_kernel<MAXP>, grid, block, 0, st, B, -1);
        for (int t = 0; t <= max_cap; ++t) {                 // launch t = max_cap only records the stop of a consensus that filled its room
            hipLaunchKernelGGL(cons_step_kernel<MAXP>, grid, block, 0, st, B, t);
            last = t;
            size_t hi = act_at;
            while (hi < late.size() && late[hi].first == t + 1) ++hi;
            if (hi > act_at) {
                const int cnt = (int)(hi - act_at);
                hipLaunchKernelGGL(cons_activate_kernel<MAXP>, dim3((cnt + 3) / 4), dim3(4 * SP_WAVE), 0, st, B, t, d_act + act_at, cnt);
                act_at = hi;
            }
            if ((t & 255) == 255 || t == max_cap) {
                SP_HIP_CHECK(ctx, hipMemcpyAsync(h_ctrl.data(), d_ctrl, sizeof(ConsCtrl) * h_ctrl.size(), hipMemcpyDeviceToHost, st));
                SP_HIP_CHECK(ctx, hipStreamSynchronize(st));
                bool all = true;
                for (uint32_t p = 0; p < n_prob; ++p) all = all && h_ctrl[2 * p + ((t + 1) & 1)].done;
                if (all) break;
            }
        }
    }
    SP_HIP_CHECK(ctx, hipGetLastError());
    const int which = (last + 1) & 1;
    hipLaunchKernelGGL(cons_finalize_kernel<MAXP>, grid, block, 0, st, B, which, d_is1, d_sc, d_sc + total);
    std::vector<uint8_t> hc(c_bytes), h_is1(total);
    std::vector<int32_t> h_sc(2 * total);
    SP_HIP_CHECK(ctx, hipMemcpyAsync(hc.data(), d_C, c_bytes, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(h_is1.data(), d_is1, total, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(h_sc.data(), d_sc, sizeof(int32_t) * 2 * total, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(h_ctrl.data(), d_ctrl, sizeof(ConsCtrl) * h_ctrl.size(), hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(st));
    SP_HIP_CHECK(ctx, hipGetLastError());
    static const char dec[4] = { 'A', 'C', 'G', 'T' };
    int32_t rc = SP_OK;
    for (uint32_t p = 0; p < n_prob; ++p) {
        const ConsParams& P = hp[p]; sp_cons_output& o = outs[p];
        const ConsCtrl& cur = h_ctrl[2 * p + which];
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
