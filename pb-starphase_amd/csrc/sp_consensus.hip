// sp_consensus.hip -- K8: read consensus by dynamic wavefront alignment (gfx950).
//
// Serves the waffle_con call sites of the reference: DualConsensusDWFA in run_dual_consensus_with_offsets
// (src/hla/caller.rs:1103-1219), the per-group ConsensusDWFA (src/hla/caller.rs:706-747) and, through the multi-way driver,
// PriorityConsensusDWFA (src/cyp2d6/caller.rs:145-270), configured as dwfa_config_from_cli does (src/hla/caller.rs:1103-1116).
// waffle_con itself (v0.4.4) is a third-party crate that is not on disk, so the contract is the one stated in DESIGN.md section 9
// and restated for the CPU in oracle/consensus.c -- a best-first search over consensus extensions, lowest total edit distance
// first, bounded as the reference configures it (max_queue_size 20, max_capacity_per_size 10); the two agree bit for bit
// (consensus strings, read assignment, per-read edit counts, nodes expanded).
//
// Mapping: ONE wavefront per read, one lane per diagonal (64-diagonal band, lane l <-> consensus pos - read pos = l - 32), the
// per-read state is one VGPR per lane (furthest read position at the current edit count).  A search node owns two state slots in
// HBM (the state at its column, and the state a window leaves behind) and one consensus buffer.
//
// The decision for a column needs the votes of every read, so a column is a global step; and the search decides after every
// column which node goes on.  Both are taken off the critical path:
//   * STEP kernel, window mode: the node the search has chosen is pushed through a WINDOW of up to CW bases in one launch.  The first
//     base is exact (it comes from the complete votes of the state), the others are speculated from the reads' own continuation
//     ("lookahead" votes).  Every wave records the exact votes of every column it passes (exact GIVEN the speculated prefix), the
//     change of the node's cost, the new lookahead votes, and leaves the new state in the node's other slot.
//   * CONTROL kernel (one workgroup per problem): sums the workgroups' vote words, keeps the longest prefix of the window on which
//     "one candidate, equal to the speculated base" holds -- the node's verified TAPE -- and then plays the search forward over the
//     tapes exactly as the one-column-per-step search would run: take the best node out, do the bookkeeping of the bounds, let it
//     consume one column of its tape, again.  Only when the best node stands at the end of its tape does it ask for the next launch:
//     another window, a replay of a cut window (verified bases only, from the kept state), or
//   * STEP kernel, expand mode: a node with several candidates gets one child per candidate (and per pair, when a second consensus
//     may be split off): every wave pushes its read once per child into the child's slot.
// Results are those of the plain search; only the number of launches changes (about two per window instead of one per column).
#include "sp_internal.h"
#include <atomic>
#include <cstdlib>
#include "sp_wfa.hip.h"
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <map>
#include <string>

namespace {

constexpr int CB = 64;          // band
constexpr int CH = 32;          // lane of diagonal 0
#ifndef SP_K8_WAVES
#define SP_K8_WAVES 8
#endif
#ifndef SP_K8_MIN_WAVES
#define SP_K8_MIN_WAVES 4      // two workgroups of eight waves per CU: the step kernel then fits 128 VGPRs (16 bytes of scratch per lane); at 134 VGPRs and three waves the 10,000-read sample took 3.4 ms longer
#endif
constexpr int CWAVES = SP_K8_WAVES;      // waves per workgroup
#ifndef SP_K8_DBG_READS
#define SP_K8_DBG_READS 16384          // SP_K8_TIMING builds: reads per launch the per-read record array holds (4 Mi records in all)
#endif
#define SP_K8_DBG_LAUNCHES (4194304 / SP_K8_DBG_READS)
#ifndef SP_K8_CW
#define SP_K8_CW 256
#endif
constexpr int CW = SP_K8_CW;    // bases per window (a multiple of 64: the control kernel walks a window 64 columns -- one per lane -- at a time)
static_assert(CW % 64 == 0 && CW >= 64 && CW <= 512, "window length");
constexpr int CWIN = 512;       // consensus bases in front of the window kept in LDS (offset_window + slack)
constexpr int RWORDS = (2 * CW + 320) / 16;   // packed read words a wave keeps in LDS: band + window + lookahead + slack around its tips
constexpr int NQ = 56;          // search nodes per problem: max_queue_size waiting + the children of one expansion + the complete one + the children of the expansions made ahead (side orders)
constexpr int NWORK = 4;        // work orders per problem and step: the search's own (order 0) and up to NWORK - 1 SIDE ORDERS -- the window or the expansion another waiting node will need when
                                // its turn comes, made in the same launch (the step kernel's grid has one row of workgroups per order)
constexpr int MAXKIDS = 16;     // children of one expansion
constexpr int CLUSTER = 64;     // workgroups whose vote words the reduce kernel sums into one set of cluster sums
constexpr int RSLICES = 16;     // reduce-kernel workgroups per cluster: each sums one slice of the cluster's words
constexpr int RGROUP = 16;      // members of a cluster one thread loads side by side
constexpr int QSV = 2 * (CW + 1) * 5, QSL = 2 * CW * 4, QE = QSV + QSL + 2 * (CW + 1);   // one cluster's sums: exact votes (w[4], end), lookahead votes, cost growth, final-cost extra
enum { F_ACTIVE = 1, F_FINISHED = 2, F_LOST = 4 };
enum { M_NONE = 0, M_INIT = 1, M_WINDOW = 2, M_EXPAND = 3 };

struct CWork {                  // what the next step launch does for this problem (written by the control kernel)
    int32_t mode, done;
    int32_t node, in_slot;      // window / init: the node and the slot that holds its state at column T; expand: the parent
    int32_t T, n, replay;       // window: n bases from column T on; replay: they are verified already
    int32_t dual, split_at;     // of the state at T
    int32_t go[2];              // window: which consensuses grow
    int32_t n_kids, pad;
    int32_t pre_go[2];          // expand behind a cut window (n > 0): which consensuses the n verified bases in front of the branch grow
    int32_t kid_node[MAXKIDS];
    int8_t  kid_base[MAXKIDS][2];      // -1: that consensus does not grow
    int8_t  kid_split[MAXKIDS];        // the child starts consensus 2 as a copy of consensus 1
    uint8_t spec[2][CW];
};
struct CNode {
    // the head (160 bytes): every node's head travels between memory and LDS in every control step
    int32_t used, id, complete; // used: 0 free, 1 in the search, 2 a child made ahead of its parent's turn (pex_kid of that parent)
    int32_t T, cur;             // column of the state in slot `cur`
    int32_t dual, split_at, stopped[2], len[2];
    int32_t n, a, q;            // the tape: n bases were pushed from T, the first a are verified, q are consumed
    int16_t have_out, la_valid; // the other slot holds the state at T + n; lookahead votes exist for that state / the state at T
    int32_t wcap;               // longest window the node may ask for: a child of an expansion starts with WRAMP0 columns and doubles with every window that stood
    long long cost0;            // cost of the state at T
    long long rest;             // what the unfinished reads add to the final cost (no early termination), for the state at T
    long long rest_out;         // the same for the state at T + n
    uint32_t ev[2][5];          // the complete votes at column T + a
    // an expansion made ahead of the node's turn (a side order): its children wait in node slots marked used = 2, unseen by the search, until the node is taken out at that
    // column and adopts them without a launch (or is dropped, and they with it)
    int32_t pex_n, pex_L;       // children made ahead (0: none); the column that branched
    int8_t  pex_kid[MAXKIDS];
    int32_t pad_head[2];
    // the tape (1.5 KB at 256-column windows): only the part a node with a tape uses travels (a branching search holds twenty nodes that have none)
    int32_t dc[CW + 1];         // cost after j pushes from T, minus cost0 (dc[0] = 0)
    uint8_t spec[2][CW];
    int32_t pad_[3];
    __device__ __forceinline__ long long cost_at(int j) const { return cost0 + (long long)dc[j]; }
};
constexpr int NODE_HEAD_WORDS = 40, NODE_WORDS = (int)(sizeof(CNode) / 4), NODE_TAPE_WORDS = NODE_WORDS - NODE_HEAD_WORDS;
static_assert(offsetof(CNode, dc) == 4 * NODE_HEAD_WORDS && sizeof(CNode) % 16 == 0, "CNode layout");
struct CSearch {
    int32_t threshold, farthest, next_id, best_node, inflight, max_queue, per_size, wo_constraint;
    int32_t windows, cut_windows, expansions, pops_mod;   // pops_mod = pops % wo_constraint, kept beside pops (a 64-bit remainder in every pass of the search was half a microsecond)
    long long pops, best_final;
    long long ticks[4];         // control kernel, 100 MHz wall clock: load + vote reduction / result of the step / search / tail
    long long step_ticks, gap_ticks, last_end;   // profiling contexts: the step kernel from its first workgroup's start to its last one's end, summed over the steps; what lies
                                                 // between the kernels of the chain (control end -> step start, step end -> control start); the wall clock at the last control end
    int32_t compound, compound_ok;               // windows ordered with the children of the branch they were foreseen to end in; those whose children were taken
    long long gap_ctl_ticks;                     // the part of gap_ticks in front of the control steps (step end -> control start)
    int32_t steps, side_windows, side_expansions, adopted;   // control steps that ordered a launch; side orders made; expansions adopted from a side order (no launch of their own)
#ifdef SP_K8_PF_PROBE
    int32_t pf_win, pf_replay, pf_exp, pf_pad;   // probe build: orders whose node stood idle at the end of its tape when the order before was made (a launch that one could have carried)
    unsigned long long idle_mask;
#endif
};
struct ConsMeta { int32_t e, c0, flags, pad; };
struct ConsRes { int32_t best, dual, split_at, len1, len2, pad; };   // what the host needs of the winning node

// One consensus problem of a batch.  All problems of a batch advance in the same launches (each with its own work order); a
// workgroup belongs to exactly one problem.  The descriptors travel in the kernel argument block for small batches.
constexpr int CMAXP = 24;
struct ConsParams {
    int n, first, first_block, n_blocks, rpw;   // reads; flattened index of local read 0; first workgroup; workgroups; reads per wave
    int first_cluster, n_clusters;              // clusters of CLUSTER consecutive workgroups (the last one may be smaller)
    int acc_block;                              // problems of <= DIRECT_BLOCKS workgroups: the first of the NWORK word blocks (one per order) their workgroups ADD their words to (atomics), read and cleared by the control step
    int min_count, delta, et, allow_dual, window, cmp_len; double min_af;
    int cap, cs;                // longest consensus; bytes between the two consensuses of a node (cap rounded up to 16: children copy their parent 16 bytes at a time)
    uint8_t* C;                 // [NQ][2][cs] base codes per node; consensus 2 shares [0, split_at) with consensus 1
    CWork* work; CSearch* srch; CNode* nodes;   // work: [NWORK] orders
    uint32_t* la;               // [NQ][2][CW][4] lookahead votes per node
    uint8_t* processed;         // [cap + 2] nodes expanded per length
    uint8_t* out_cons;          // [2][cap] the consensus bytes of the node the search ended with (written by the finalize kernel)
    struct ConsRes* out_res; CSearch* out_srch;
};
// off: the consensus length at which a late read is placed = its offset + the bases the placement compares (min(offset_compare_length, n)); -1: from the start.  off0: the offset itself
struct ReadInfo { const uint32_t* w; const uint32_t* np; int n, off; int off0, pad; };
// The last placement search of a read on consensus 1 / 2 and its answer.  Every node of a search that reaches the read's offset places it again,
// mostly in front of the very same bases (siblings differ at one column): 2.7 searches per read in a CYP2D6 region batch, 20 us each.
struct PlaceMemo { int valid, M, off, c0; uint32_t text[512 / 16 + 2]; };
template <int MAXP> struct ConsBatchT {
    ConsParams p[MAXP]; int n_prob;
    const ReadInfo* info;       // [total]
    PlaceMemo* memo;            // [NWORK][total][2] (a set per order row: two rows of a launch may place the same read at once)
    uint16_t* H;                // [node][slot][consensus][total][64] furthest read position per diagonal (0xFFFF = none)
    ConsMeta* meta;             // [node][slot][consensus][total]
    unsigned long long* PV;     // [blocks][2][CW + 1] exact votes per workgroup: four 16-bit fields (A, C, G, T) in 12ths of a read
    uint32_t* PE;               // [blocks][2][CW + 1] "the read ends here" votes
    unsigned long long* PL;     // [blocks][2][CW]     lookahead votes (one per read and tip)
    uint32_t* PC;               // [blocks][CW + 1]    growth of the node's cost at push j (expand: cost the child adds)
    uint32_t* PR;               // [blocks][CW + 1]    what unfinished reads add to a final cost (index n / child)
    uint32_t* Q;                // [clusters][QE]      the words above summed over a cluster of workgroups, one u32 per field
    unsigned long long* dbg;    // SP_K8_TIMING builds: per launch index [4096][4] = slowest wave, slowest wave that placed no read, sum of waves, waves (ticks)
    uint32_t* prog;             // host memory the device writes: per problem { control steps made, search ended }: the host enqueues a few launches ahead of it
    unsigned long long* step_t; // profiling contexts (else nullptr): per problem { earliest workgroup start, latest workgroup end } of the step kernel, reset by the control step
    uint32_t* sync;             // persistent mode (else nullptr): per problem { steps the control workgroup has answered, arrivals of step workgroups, -, - }, then one abort word for the batch
    uint32_t* ready;            // persistent mode: host memory, counts the control workgroups that have started (the step workgroups are launched behind them)
    uint32_t step_cap;          // persistent mode: a search that has not ended after this many steps ends the batch (the launch-pair loop's own bound)
    int total;
    int nside;                  // side orders per problem and step (0 .. NWORK - 1): the step kernel's grid has 1 + nside rows
    int k8_compound;            // windows may be ordered with the children of the branch the lookahead votes foresee at their end
};
template <> struct ConsBatchT<0> {
    const ConsParams* p; const int* block_prob; int n_prob;
    const ReadInfo* info; PlaceMemo* memo; uint16_t* H; ConsMeta* meta; unsigned long long* PV; uint32_t* PE; unsigned long long* PL; uint32_t* PC; uint32_t* PR; uint32_t* Q; unsigned long long* dbg; uint32_t* prog; unsigned long long* step_t; uint32_t* sync; uint32_t* ready; uint32_t step_cap; const int* cluster_prob; int total; int nside; int k8_compound;
};
struct ConsSetup { SeqSetView reads; const uint32_t* idx; const int32_t* offsets; int n, first, cmp_len, pad_; };

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
// extend(d, T): every diagonal of d as far as its read keeps matching the consensus, inside the read and the T columns
template <class RB, class EXT>
__device__ __forceinline__ void dwfa_push_t(Dwfa& d, int n, RB rb, EXT extend, int T, int nb, int et, int lane) {
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
        extend(d, T);
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

// the same search for windows that do not fit the wave's scratch (offset_window > 512): Sellers' search, one Myers bit-vector scan per
// lane over the end positions it owns (an occurrence of an L-base pattern with <= L edits spans <= 2L text bases)
template <class RB, class CA>
__device__ __noinline__ int find_start_scan(int rn, RB rb, CA ca, int off, int end, int W, int L, int lane) {
    const int ws = off - W > 0 ? off - W : 0, M = end - ws;
    if (L > rn) L = rn;
    if (M <= 0 || L <= 0 || off <= ws) return off < end ? off : end;
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
            const int x = ca(end - j);
            const unsigned long long Eq = x == 0 ? peq[0] : x == 1 ? peq[1] : x == 2 ? peq[2] : peq[3];
            const unsigned long long Xv = Eq | Mv;
            const unsigned long long Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            unsigned long long Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
            if (Ph & top) ++score; else if (Mh & top) --score;
            Ph <<= 1; Mh <<= 1;
            Pv = (Mh | ~(Xv | Ph)) & ones; Mv = Ph & Xv & ones;
            if (j >= jlo && j >= end - off) {
                const int p = end - j, dist = p > centre ? p - centre : centre - p;
                const unsigned long long kk = ((unsigned long long)score << 44) | ((unsigned long long)dist << 22) | (unsigned long long)p;
                key = kk < key ? kk : key;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other < key ? other : key; }
    return (int)(key & ((1ull << 22) - 1));
}

// placement of a late read: Sellers' search of its first L bases (read backwards) in the last W consensus bases (backwards from `off`): the
// distance matrix D[i][j] (i bases of the pattern against a stretch that ends at text position j; D[0][j] = 0, D[i][0] = i) is filled one
// anti-diagonal per step, lane l holding row l + 1: a cell needs the lane's own last value and the last two of the lane below (two DPP moves).
// The last row names, for every start position p = off - j, the edits of the best occurrence (kept per column in `score`); the smallest
// (edits, distance from the middle of the window, p) wins.  L + M short steps; the per-lane Myers scan this replaces (every lane its own 2 L
// warm-up columns and its own 64-bit bit-vector) took 28 us per state: more than half of a CYP2D6 window launch that places a read.
// pc: the lane's pattern base (read base L - 1 - lane; 7 behind the pattern, 4 for an N: neither matches anything)
// tr: the text, backwards, 2 bits per base (text position j = 1 .. M at bit 2 (j - 1)); score: M bytes.  Out of line: its loop wants few registers.
__device__ __noinline__ int find_start_diag(int pc, int L_, int M_, int off_, int end_, int W_, int lane, const uint32_t* tr_, uint8_t* score_) {
    // (arguments of an out-of-line function arrive in vector registers: the uniform ones go back to scalars, the two LDS pointers to LDS addresses --
    //  through generic pointers every access is a flat load the loop has to wait for)
    const int L = __builtin_amdgcn_readfirstlane(L_), M = __builtin_amdgcn_readfirstlane(M_), off = __builtin_amdgcn_readfirstlane(off_), W = __builtin_amdgcn_readfirstlane(W_), end = __builtin_amdgcn_readfirstlane(end_);
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    spw::lds_cu32* tr = (spw::lds_cu32*)(uintptr_t)spw::lds_addr(tr_);
    lds_u8* score = (lds_u8*)(uintptr_t)spw::lds_addr(reinterpret_cast<const uint32_t*>(score_));
    const int i = lane + 1;                                                  // this lane's row
    int prev1 = i, prev2 = i;                                                // the lane's value one and two steps ago
    uint32_t tw = tr[0], tw_next = tw;
    const bool last_row = lane == L - 1;
    for (int d = 1; d <= L + M; ++d) {
        const int j = d - i;                                                 // the lane's column in this step
        const int up = spw::from_lower(prev1, 0), diag = spw::from_lower(prev2, 0);   // row i - 1 at columns j and j - 1 (row 0 is all zeros)
        if (j >= 1 && ((j - 1) & 15) == 0) tw = tw_next;
        int cur = i;                                                         // column 0 (and the steps before the lane starts)
        if (j >= 1 && j <= M) {
            const int tc = (int)((tw >> (((j - 1) & 15) << 1)) & 3u);
            const int a = diag + (pc != tc ? 1 : 0), bmin = (up < prev1 ? up : prev1) + 1;
            cur = a < bmin ? a : bmin;
            if (last_row) score[j - 1] = (uint8_t)cur;
        }
        if (j >= 0 && (j & 15) == 0) tw_next = tr[j >> 4];                   // the word the lane starts in the next step
        prev2 = prev1; prev1 = cur;
    }
    spw::wave_lds_sync();
    const int centre = off - W / 2;
    unsigned long long key = ~0ull;
    for (int j = end - off + lane; j <= M; j += SP_WAVE) {                 // starts off, off - 1, ..., the window's first base (text position j <-> start end - j)
        if (j < 1) continue;
        const int p = end - j, dist = p > centre ? p - centre : centre - p;
        const unsigned long long kk = ((unsigned long long)score[j - 1] << 44) | ((unsigned long long)dist << 22) | (unsigned long long)p;
        key = kk < key ? kk : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other < key ? other : key; }
    return (int)(key & ((1ull << 22) - 1));
}

// the same search for patterns of 65 .. 128 bases (the reference's CYP2D6 caller compares 100): two rows per lane, lane l holding rows 2l + 1 and 2l + 2.
// The upper row takes its neighbours from the lane's own lower row (one and two steps ago), the lower row from the lane below's upper row (DPP); the two
// rows of a lane stand one text column apart, so the upper row's text base is the lower row's of the step before.  pa / pb: the pattern bases of the two rows.
__device__ __noinline__ int find_start_diag2(int pa, int pb, int L_, int M_, int off_, int end_, int W_, int lane, const uint32_t* tr_, uint8_t* score_) {
    const int L = __builtin_amdgcn_readfirstlane(L_), M = __builtin_amdgcn_readfirstlane(M_), off = __builtin_amdgcn_readfirstlane(off_), W = __builtin_amdgcn_readfirstlane(W_), end = __builtin_amdgcn_readfirstlane(end_);
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    spw::lds_cu32* tr = (spw::lds_cu32*)(uintptr_t)spw::lds_addr(tr_);
    lds_u8* score = (lds_u8*)(uintptr_t)spw::lds_addr(reinterpret_cast<const uint32_t*>(score_));
    const int ia = 2 * lane + 1, ib = ia + 1;                                // this lane's rows
    int a1 = ia, a2 = ia, b1 = ib, b2 = ib;                                  // their values one and two steps ago
    uint32_t tw = tr[0], tw_next = tw;
    int t_prev = 0;                                                          // the text base of row ia's column one step ago = row ib's column now
    for (int d = 1; d <= L + M; ++d) {
        const int ja = d - ia, jb = ja - 1;                                  // the columns of the two rows in this step
        const int up_a = spw::from_lower(b1, 0), diag_a = spw::from_lower(b2, 0);     // row ia - 1 (the lane below's upper row; row 0 is all zeros) at columns ja and ja - 1
        if (ja >= 1 && ((ja - 1) & 15) == 0) tw = tw_next;
        int cur_a = ia, cur_b = ib, tc = 0;
        if (ja >= 1 && ja <= M) {
            tc = (int)((tw >> (((ja - 1) & 15) << 1)) & 3u);
            const int x = diag_a + (pa != tc ? 1 : 0), y = (up_a < a1 ? up_a : a1) + 1;
            cur_a = x < y ? x : y;
            if (ia == L) score[ja - 1] = (uint8_t)cur_a;
        }
        if (jb >= 1 && jb <= M) {
            const int x = a2 + (pb != t_prev ? 1 : 0), y = (a1 < b1 ? a1 : b1) + 1;       // row ia at columns jb - 1 and jb: two steps and one step ago
            cur_b = x < y ? x : y;
            if (ib == L) score[jb - 1] = (uint8_t)cur_b;
        }
        if (ja >= 0 && (ja & 15) == 0) tw_next = tr[ja >> 4];                // the word row ia starts in the next step
        t_prev = tc;
        a2 = a1; a1 = cur_a; b2 = b1; b1 = cur_b;
    }
    spw::wave_lds_sync();
    const int centre = off - W / 2;
    unsigned long long key = ~0ull;
    for (int j = end - off + lane; j <= M; j += SP_WAVE) {                 // starts off, off - 1, ..., the window's first base (text position j <-> start end - j)
        if (j < 1) continue;
        const int p = end - j, dist = p > centre ? p - centre : centre - p;
        const unsigned long long kk = ((unsigned long long)score[j - 1] << 44) | ((unsigned long long)dist << 22) | (unsigned long long)p;
        key = kk < key ? kk : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other < key ? other : key; }
    return (int)(key & ((1ull << 22) - 1));
}

constexpr int ACT_CONS = 512;   // consensus bases a wave packs while a late read catches up (offset_window + slack)
constexpr int ACT_READ = 2 * (CW + 1) > 640 ? ((2 * (CW + 1) + 63) / 64) * 64 : 640;   // read bases it keeps (catch-up length + band + edits; the edit-count profile of a window shares the bytes)
struct ActScratch {             // per wave
    uint8_t rcache[ACT_READ];
    uint32_t cpack[ACT_CONS / 16 + 2];
    uint32_t rpack[ACT_READ / 16 + 2];
    uint32_t npack[ACT_READ / 16 + 2];
    uint8_t score[ACT_CONS];    // placement search: edits of the best occurrence per start position
    const uint32_t* staged;     // the read whose first bases rcache / rpack / npack hold (both states of a read, and every child of an expansion, place the same read)
#ifdef SP_K8_TIMING
    long long tk[4];            // ticks (100 MHz) spent in a placement: read staging, start search, consensus packing, catch-up
#endif
};


// consensus i of a node as a step kernel sees it: the LDS copy of [w0, w0 + CWIN + CW), one overriding base (the base a child
// appends), the node's committed bases elsewhere
// What the control step and the step workgroups of a problem hand each other in PERSISTENT mode crosses no kernel boundary, so it crosses the caches explicitly: the
// writer stores it write-through (agent-scope relaxed atomic stores: `sc1`, the line leaves the XCD's L2), waits for its stores (s_waitcnt vmcnt(0)) and only then raises
// the problem's word; the reader loads it past its CU's L1 (`sc1` loads).  No L2 write-back and no invalidate per step (an agent-scope release / acquire pair cost 7 us
// of every hand-over).  In launch-pair mode the same accessors are harmless: an `sc1` load is served by L2 like a plain one.
template <class T> __device__ __forceinline__ T coh_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void coh_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void put(T* p, T v, bool coh) { if (coh) coh_store(p, v); else *p = v; }       // (a write-through store drops the line from L2: only where it is needed)

struct ConsAccess {
    const uint8_t* win; int w0; const uint8_t* C; int cap, split_at, i, ov_pos, ov_base;
    const uint32_t* pk;         // the same window 2 bits a base (word k = window bases 16 k .. 16 k + 15; zero where the window has no base)
    // the 16 bases from `pos` on as one packed word, where they all lie in the staged window and none of them is the overriding base
    __device__ __forceinline__ bool word16(int pos, uint32_t& out) const {
        const int y = pos - w0;
        if (y < 0 || y + 16 > CWIN + CW || (ov_pos >= pos && ov_pos < pos + 16)) return false;
        out = __builtin_amdgcn_alignbit(pk[(y >> 4) + 1], pk[y >> 4], (uint32_t)(y & 15) << 1);
        return true;
    }
    __device__ __forceinline__ int at(int pos) const {
        if (pos == ov_pos) return ov_base;
        const int y = pos - w0;
        if ((unsigned)y < (unsigned)(CWIN + CW)) return (int)win[y];
        return (int)coh_load((i == 1 && pos < split_at) ? &C[pos] : &C[(size_t)i * cap + pos]);      // (bases the control step wrote: rarely outside the staged window)
    }
};

// Placement of a late read (add_sequence_offset) when the consensus reaches length `len` == its offset: start search in the
// offset_window bases before it, then the catch-up pushes.  Kept out of line: it is rare and needs twice the registers of
// the window loop.
__device__ __noinline__ Dwfa activate_late(ReadView rv, ConsAccess cacc, ActScratch* Ap, PlaceMemo* memo, int off, int len, int window, int cmp_len, int et, int lane) {
    ActScratch& A = *Ap;
    // the scratch arrays through LDS addresses (Ap arrives as a generic pointer: every access through it is a flat load or store that the loops below wait for)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    lds_u32* const L_rpack = (lds_u32*)(uintptr_t)spw::lds_addr(A.rpack);
    lds_u32* const L_npack = (lds_u32*)(uintptr_t)spw::lds_addr(A.npack);
    lds_u32* const L_cpack = (lds_u32*)(uintptr_t)spw::lds_addr(A.cpack);
    lds_u32* const L_score32 = (lds_u32*)(uintptr_t)spw::lds_addr(reinterpret_cast<const uint32_t*>(A.score));
    lds_u8* const L_rcache = (lds_u8*)(uintptr_t)spw::lds_addr(reinterpret_cast<const uint32_t*>(A.rcache));
    Dwfa d;
#ifdef SP_K8_TIMING
    long long tq = wall_clock64();
#define ACT_T(k) do { const long long _n = wall_clock64(); if (lane == 0) A.tk[k] += _n - tq; tq = _n; } while (0)
#else
#define ACT_T(k) do { } while (0)
#endif
    auto ca = [&](int pos) { return cacc.at(pos); };
    const int ws = off - window > 0 ? off - window : 0;
    // the read's first 640 bases: one packed word (and N word) per lane from memory, the byte per base the search and the slow catch-up read
    // unpacked out of LDS (a base at a time from memory was ten dependent round trips)
    if (A.staged != rv.w) {
        const int rwords0 = ((rv.n < ACT_READ ? rv.n : ACT_READ) + 15) >> 4;
        for (int w = lane; w < ACT_READ / 16 + 2; w += SP_WAVE) { L_rpack[w] = w < rwords0 ? rv.w[w] : 0u; L_npack[w] = (rv.np && w < rwords0) ? rv.np[w] : 0u; }
        spw::wave_lds_sync();
        for (int x = lane; x < rv.n && x < ACT_READ; x += SP_WAVE) {
            const uint32_t sh = (uint32_t)(x & 15) << 1;
            L_rcache[x] = ((L_npack[x >> 4] >> sh) & 1u) ? (uint8_t)4 : (uint8_t)((L_rpack[x >> 4] >> sh) & 3u);
        }
        if (lane == 0) A.staged = rv.w;
    }
    spw::wave_lds_sync();
    auto rbc = [&](int h) { return h < ACT_READ ? (int)L_rcache[h] : read_base(rv, h); };
    ACT_T(0);
    {
        // the search (oracle/consensus.c find_start): start positions [off - window, off], the read's first L bases against any prefix of the consensus behind the
        // start; the consensus has just reached off + L, so the text is its last M = len - ws bases, read backwards from `len`
        const int ws0 = off - window > 0 ? off - window : 0, M = len - ws0, L = cmp_len < rv.n ? cmp_len : rv.n;
        if (M <= 0 || L <= 0 || off <= ws0) d.c0 = off < len ? off : len;
        else if (M > ACT_CONS) d.c0 = find_start_scan(rv.n, rbc, ca, off, len, window, cmp_len, lane);
        else {
            for (int w = lane; w < ((M + 15) >> 4) + 1; w += SP_WAVE) {      // the text backwards, 2 bits per base (cpack is rebuilt for the catch-up below)
                // (out of the window's packed copy where the 16 bases lie in it: two loads and a digit reversal; base by base through cacc.at() -- a flat load and three
                //  branches each -- the two packings of a placement were 8 of the 33 us of a wave that places a read)
                uint32_t word = 0, fw;
                const int left = M - w * 16;                                   // text positions of this word that exist
                if (left <= 0) word = 0;
                else if (cacc.word16(len - w * 16 - 16, fw)) {
                    fw = ((fw >> 2) & 0x33333333u) | ((fw & 0x33333333u) << 2);
                    fw = ((fw >> 4) & 0x0F0F0F0Fu) | ((fw & 0x0F0F0F0Fu) << 4);
                    word = __builtin_bswap32(fw);
                    if (left < 16) word &= (1u << (left << 1)) - 1u;
                }
                else for (int b = 0; b < 16; ++b) { const int j = w * 16 + b + 1; if (j <= M) word |= (uint32_t)(ca(len - j) & 3) << (b << 1); }
                L_cpack[w] = word;
            }
            spw::wave_lds_sync();
            const int nw = ((M + 15) >> 4) + 1;                                // (<= 33 words: a lane each)
            PlaceMemo* mm = memo + cacc.i;
            // (loads that go past this CU's L1: the wave may have written the entry a moment ago, for another child of the same expansion; the text
            //  is read before it is known whether the entry holds anything: one round trip)
            auto peek = [](const int* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            const uint32_t seen = lane < nw ? (uint32_t)peek(reinterpret_cast<const int*>(&mm->text[lane])) : 0u;
            // (the entry's four words asked for together: `valid && M == .. && off == ..` on the loads themselves asked for one after the other -- an atomic load is not moved
            //  across the && in front of it --, three round trips to memory in front of every placement: 4 - 5 us of the slowest wave of a window launch)
            const int m_c0 = peek(&mm->c0), m_valid = peek(&mm->valid), m_M = peek(&mm->M), m_off = peek(&mm->off);
            bool known = m_valid && m_M == M && m_off == off;
            if (known) known = __ballot(lane < nw && seen != L_cpack[lane]) == 0;
            ACT_T(0);
            // The search by diagonal transition first (Landau-Vishkin for Sellers' matrix): for e = 0, 1, 2, ... the furthest pattern row every diagonal of the matrix
            // reaches with at most e edits -- a slide along the equal bases (word compares on the two packed sequences) after the step from the neighbouring diagonals'
            // reach with e - 1 -- until a diagonal that ends at a start position of the window reaches the pattern's last row: the occurrences with e edits, and e is the
            // smallest number any has; among them the start closest to the middle of the window, then the leftmost, exactly what the matrix's last row says.  A read's
            // first bases hold a handful of errors at most: one to four rounds of a few word compares instead of the L x M matrix (89 us of the slowest wave of a CYP2D6
            // window launch with the reference's 100-base comparison -- two thirds of the slowest-wave time of a sample's first batch).  Both sequences are packed
            // backwards here; a pattern with an N, more than LV_MAX edits or more diagonals than a lane holds eight of go to the matrix.
            int exact_c0 = -1;
            constexpr int LV_MAX = 12;
            const int lv_nd = M - L + 1 + 2 * LV_MAX;                                    // diagonals d = x - LV_MAX (text column - pattern row), x = c * 64 + lane
            if (!known && rv.np == nullptr && M >= L && lv_nd <= 8 * SP_WAVE) {
                lds_u32* const ppack = L_score32;                                        // (the matrix's score bytes are not in use yet)
                const int npw = (L + 15) >> 4;
                if (lane < 10) {
                    uint32_t pw = 0;
                    if (lane < npw) for (int b = 0; b < 16; ++b) { const int u = lane * 16 + b; if (u < L) pw |= (uint32_t)(rbc(L - 1 - u) & 3) << (b << 1); }
                    ppack[lane] = pw;
                }
                spw::wave_lds_sync();
                const int npl = (lv_nd + SP_WAVE - 1) / SP_WAVE, centre = off - window / 2;
                // slide: from pattern row r on diagonal dd along the equal bases (pattern base r against text base r + dd, both 0-based in the backward packing)
                auto slide = [&](int r, int dd) {
                    const int lim = L < M - dd ? L : M - dd;
                    while (r < lim) {
                        const int t = r + dd;
                        const uint32_t pwd = __builtin_amdgcn_alignbit(ppack[(r >> 4) + 1], ppack[r >> 4], (uint32_t)(r & 15) << 1);
                        const uint32_t twd = __builtin_amdgcn_alignbit(L_cpack[(t >> 4) + 1], L_cpack[t >> 4], (uint32_t)(t & 15) << 1);
                        const uint32_t diff = pwd ^ twd;
                        const int same = diff ? (__builtin_ctz(diff) >> 1) : 16;
                        r += same;
                        if (same < 16) break;
                    }
                    return r < lim ? r : lim;
                };
                int reach[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int x = c * SP_WAVE + lane, dd = x - LV_MAX;
                    reach[c] = (c < npl && x < lv_nd && dd >= 0 && dd <= M) ? slide(0, dd) : -1;
                }
                for (int e = 0; e <= LV_MAX; ++e) {
                    unsigned long long key = ~0ull;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int x = c * SP_WAVE + lane, dd = x - LV_MAX;
                        if (c < npl && x < lv_nd && dd >= 0 && dd + L >= len - off && dd <= M - L && reach[c] >= L) {      // ends at text position dd + L <-> start len - dd - L (the starts off, off - 1, ... of the window)
                            const int p = len - dd - L, dist = p > centre ? p - centre : centre - p;
                            const unsigned long long kk = ((unsigned long long)dist << 22) | (unsigned long long)p;
                            key = kk < key ? kk : key;
                        }
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other < key ? other : key; }
                    if (key != ~0ull) { exact_c0 = (int)(key & ((1ull << 22) - 1)); break; }
                    if (e == LV_MAX) break;
                    int nxt[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        nxt[c] = -1;
                        if (c < npl) {
                            const int x = c * SP_WAVE + lane, dd = x - LV_MAX;
                            int left = __shfl_up(reach[c], 1), up = __shfl_down(reach[c], 1);           // diagonals dd - 1 and dd + 1
                            const int wrap_l = c > 0 ? __shfl(reach[c > 0 ? c - 1 : 0], SP_WAVE - 1) : -1, wrap_u = c + 1 < 8 ? __shfl(reach[c + 1 < 8 ? c + 1 : 7], 0) : -1;
                            if (lane == 0) left = wrap_l;
                            if (lane == SP_WAVE - 1) up = (c + 1 < npl) ? wrap_u : -1;
                            int best = reach[c] >= 0 ? reach[c] + 1 : -1;                               // a mismatch on the diagonal
                            if (up >= 0 && up + 1 > best) best = up + 1;                                // a pattern base without a text base (from dd + 1, one row down)
                            if (left > best) best = left;                                               // a text base without a pattern base (from dd - 1, same row)
                            if (x < lv_nd && dd <= M && best >= 0) {
                                const int lim = L < M - dd ? L : M - dd;
                                nxt[c] = slide(best < lim ? best : lim, dd);
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) reach[c] = nxt[c];
                }
                spw::wave_lds_sync();
            }
            if (known) d.c0 = m_c0;
            else if (exact_c0 >= 0) {
                d.c0 = exact_c0;
                if (lane < nw) mm->text[lane] = L_cpack[lane];
                if (lane == 0) { mm->valid = 1; mm->M = M; mm->off = off; mm->c0 = d.c0; }
            }
            else {
                if (L <= SP_WAVE) d.c0 = find_start_diag(lane < L ? rbc(L - 1 - lane) : 7, L, M, off, len, window, lane, A.cpack, A.score);
                else d.c0 = find_start_diag2(2 * lane + 1 <= L ? rbc(L - 2 * lane - 1) : 7, 2 * lane + 2 <= L ? rbc(L - 2 * lane - 2) : 7, L, M, off, len, window, lane, A.cpack, A.score);
                if (lane < nw) mm->text[lane] = L_cpack[lane];
                if (lane == 0) { mm->valid = 1; mm->M = M; mm->off = off; mm->c0 = d.c0; }
            }
            spw::wave_lds_sync();
        }
    }
    ACT_T(1);
    d.H = lane == CH ? 0 : SP_NEG; d.e = 0; d.flags = F_ACTIVE | ((et && rv.n == 0) ? F_FINISHED : 0);
    const int c0 = d.c0, span = len - c0, cwinlen = len - ws;
    const bool packed = rv.np == nullptr && cwinlen <= ACT_CONS;
    if (span > 0 && !(d.flags & F_FINISHED) && !(et && rv.n <= span + CB) && packed) {
        // 16 bases per step out of 2-bit packed copies of the two windows
        for (int w = lane; w < ACT_CONS / 16 + 2; w += SP_WAVE) {
            uint32_t word = 0;
            const int left = cwinlen - w * 16;
            if (left <= 0) word = 0;
            else if (cacc.word16(ws + w * 16, word)) { if (left < 16) word &= (1u << (left << 1)) - 1u; }
            else { word = 0; for (int b = 0; b < 16; ++b) { const int x = w * 16 + b; if (x < cwinlen) word |= (uint32_t)(ca(ws + x) & 3) << (b << 1); } }
            L_cpack[w] = word;
        }
        spw::wave_lds_sync();                                  // (rpack holds the read's packed words since the start)
        ACT_T(2);
        const int kk = lane - CH, cbase = c0 - ws;
        dwfa_catchup_t(d, rv.n, [&]() {
            for (;;) {
                int left = rv.n - d.H; { const int l2 = span - (d.H + kk); left = l2 < left ? l2 : left; }
                const bool go = d.H >= 0 && left > 0;
                int nm = 0;
                if (go) {
                    const int pr = d.H, pc = cbase + d.H + kk;
                    const uint32_t a = __builtin_amdgcn_alignbit(L_rpack[(pr >> 4) + 1], L_rpack[pr >> 4], (uint32_t)(pr & 15) << 1);
                    const uint32_t b = __builtin_amdgcn_alignbit(L_cpack[(pc >> 4) + 1], L_cpack[pc >> 4], (uint32_t)(pc & 15) << 1);
                    const uint32_t x = a ^ b, mm = (x | (x >> 1)) & 0x55555555u;
                    nm = mm ? (__builtin_ctz(mm) >> 1) : 16;
                    nm = nm < left ? nm : left;
                    d.H += nm;
                }
                if (!__ballot(go && nm == 16 && left > 16)) break;
            }
        }, span, lane);
    } else if (span > 0 && !(d.flags & F_FINISHED) && !(et && rv.n <= span + CB)) {
        const int kk = lane - CH;
        dwfa_catchup_t(d, rv.n, [&]() {
            for (;;) {
                bool go = d.H >= 0 && d.H < rv.n && d.H + kk < span;
                if (go) { const int x = rbc(d.H); go = x < 4 && x == ca(d.c0 + d.H + kk); }
                if (!__ballot(go)) break;
                if (go) d.H += 1;
            }
        }, span, lane);
    } else {
        for (int Tl = 1; Tl <= span; ++Tl) {
            if (d.flags & (F_FINISHED | F_LOST)) break;
            dwfa_push_t(d, rv.n, rbc, [&](Dwfa& x, int Tc) {
                const int kk = lane - CH;
                for (;;) {
                    bool go = x.H >= 0 && x.H < rv.n && x.H + kk < Tc;
                    if (go) { const int b = rbc(x.H); go = b < 4 && b == ca(x.c0 + x.H + kk); }
                    if (!__ballot(go)) break;
                    if (go) x.H += 1;
                }
            }, Tl, ca(c0 + Tl - 1), et, lane);
        }
    }
    ACT_T(3);
    return d;
}

// The words a step launch leaves behind that anybody reads: the slots 0 .. used - 1 of the exact votes, the cost growth and the final-cost extra
// (window mode: one per push + the state's own; expand mode: one per child; init: slot 0) and, outside expand mode, the lookahead votes.  The
// reduce and the control kernel walk them through a compact index; an expansion of a 100-read group moves 0.3 KB per workgroup instead of 12.
// An expansion of up to KID_LA_KIDS children collects their lookahead votes as well -- KID_LA columns each, the lookahead words of a launch shared out
// [child][consensus][column] -- so that a child's first window is a speculated one (a child without them first needs a launch of one column to have any: every
// expansion cost its surviving child a launch pair more).
constexpr int KID_LA_KIDS = 4, KID_LA = CW / KID_LA_KIDS;
#ifndef SP_K8_KID_W0
#define SP_K8_KID_W0 32
#endif
constexpr int KID_W0 = SP_K8_KID_W0 < KID_LA ? SP_K8_KID_W0 : KID_LA;      // first window of such a child in a crowded search (most children are dropped after a few columns)
#ifndef SP_K8_RAMP
#define SP_K8_RAMP 2            // a node's window grows by this factor with every window of its that stood
#endif
#ifndef SP_K8_KID_CALM
#define SP_K8_KID_CALM 6
#endif
constexpr int KID_CALM = SP_K8_KID_CALM;                                   // up to this many other nodes waiting: the child's first window is KID_LA columns
struct UsedWords {
    int used, has_la, a_end, b_end, total;
    __device__ __forceinline__ UsedWords(int mode, int n, int n_kids) {
        used = mode == M_EXPAND ? n_kids : (mode == M_WINDOW ? n + 1 + n_kids : 1);      // (a window that branches at its end: the children's words lie behind its columns')
        has_la = mode != M_EXPAND || n_kids <= KID_LA_KIDS;
        a_end = 2 * used * 5; b_end = a_end + (has_la ? QSL : 0); total = b_end + 2 * used;
    }
    // compact index -> position in the QE-word layout of the cluster sums
    __device__ __forceinline__ int at(int c) const {
        if (c < a_end) { const int i = c / (used * 5), rem = c % (used * 5); return (i * (CW + 1) + rem / 5) * 5 + rem % 5; }
        if (c < b_end) return QSV + (c - a_end);
        const int r = c - b_end;
        return r < used ? QSV + QSL + r : QSV + QSL + (CW + 1) + (r - used);
    }
};
// word o (QE layout) of workgroup blk
template <class BT> __device__ __forceinline__ uint32_t block_word(const BT& B, size_t blk, int o) {
    constexpr int EV = 2 * (CW + 1), EL = 2 * CW, EC = CW + 1;
    if (o < QSV) { const int e = o / 5, f = o % 5; return f < 4 ? (uint32_t)((B.PV[blk * EV + e] >> (16 * f)) & 0xFFFFull) : B.PE[blk * EV + e]; }
    if (o < QSV + QSL) { const int e = (o - QSV) / 4, f = (o - QSV) % 4; return (uint32_t)((B.PL[blk * EL + e] >> (16 * f)) & 0xFFFFull); }
    if (o < QSV + QSL + EC) return B.PC[blk * EC + (o - QSV - QSL)];
    return B.PR[blk * EC + (o - QSV - QSL - EC)];
}
#ifndef SP_K8_WRAMP0
#define SP_K8_WRAMP0 64
#endif
constexpr int WRAMP0 = SP_K8_WRAMP0;           // first window of a node born in an expansion (most such nodes are dropped after a few columns: a 256-column window costs its slowest read 4 x as long)
constexpr int BULK_MARGIN_PLACED = 2;  // the same right behind the column that placed the read (its states have only the catch-up's edits yet)
constexpr int BULK_MARGIN = 8;         // edits a read's worse state must be behind the better one to go through a window ahead of it (a state that close may draw level)
#ifndef SP_K8_AUTO_BATCHES
#define SP_K8_AUTO_BATCHES 1
#endif
constexpr int PERSIST_AUTO_BATCHES = SP_K8_AUTO_BATCHES;
#ifndef SP_K8_LIGHT
#define SP_K8_LIGHT 64
#endif
constexpr int PERSIST_LIGHT = SP_K8_LIGHT;          // resident workgroups (step + 2 per control workgroup) up to which a batch's footprint counts as light
#ifndef SP_K8_DIRECT_BLOCKS
#define SP_K8_DIRECT_BLOCKS (680 * 8 / SP_K8_WAVES)
#endif
// Workgroups of a problem up to which its workgroups ADD their vote words to one block per problem (memory-side atomics) and the control kernel reads that block: a batch of such
// problems has no reduce launch -- two launches per step instead of three.  The bound is the 16-bit vote fields: 12 units per read and column, 5,461 reads = 682 workgroups at one
// read per wave (128 until round 5: an HLA gene of a 10,000-read sample has 625 workgroups and paid the third launch and its boundary in every step).
constexpr int DIRECT_BLOCKS = SP_K8_DIRECT_BLOCKS;
constexpr int32_t SP_K8_AGAIN = -777;    // run_chunk to its caller: run the batch again (never leaves the library)
constexpr int PERSIST_BLOCKS = 128 * 8 / CWAVES;     // workgroups of a problem up to which a batch may run as persistent kernels (all workgroups of the batch resident together)

template <int MAXP> __device__ __forceinline__ int block_problem(const ConsBatchT<MAXP>& B, int block) {
    int pi = 0;
    if constexpr (MAXP == 0) pi = B.block_prob[block];
    else {
#pragma unroll
        for (int i = 1; i < MAXP; ++i) if (i < B.n_prob && block >= B.p[i].first_block) pi = i;
    }
    return pi;
}
template <int MAXP> __device__ __forceinline__ int block_problem(const ConsBatchT<MAXP>& B) { return block_problem<MAXP>(B, (int)blockIdx.x); }

__device__ __forceinline__ size_t state_plane(int node, int slot, int cons) { return (size_t)((node * 2 + slot) * 2 + cons); }

// the smaller edit count of the placed states of a read (a state that is no longer tracked keeps its count)
__device__ __forceinline__ int read_cost(const Dwfa& a0, const Dwfa& a1, bool dualrun) {
    int c = -1;
    if (a0.flags & F_ACTIVE) c = a0.e;
    if (dualrun && (a1.flags & F_ACTIVE) && (c < 0 || a1.e < c)) c = a1.e;
    return c < 0 ? 0 : c;
}

// ------------------------------------------------------------------------------------------------------------------------------
// the step: window mode pushes the chosen node through n bases, expand mode makes the children of a node, init builds the root
// ------------------------------------------------------------------------------------------------------------------------------
template <int MAXP>
__device__ __forceinline__ void cons_step_body(const ConsBatchT<MAXP>& B, const int wrow) {       // wrow: the order of the problem this workgroup works on (0: the search's own)
    __shared__ unsigned long long lv[2][CW + 1];          // exact votes after j pushes (column T + j) / of child j
    __shared__ uint32_t le[2][CW + 1];
    __shared__ unsigned long long ll[2][CW];              // lookahead: ll[i][x] predicts column T + n + 1 + x
    __shared__ uint32_t lc[CW + 1], lr[CW + 1];           // cost growth at push j / of child j; final-cost extra of the end state / child
    __shared__ uint8_t cwin[2][CWIN + CW];                // consensus bases [T - CWIN, T + n)
    __shared__ uint32_t rwin[CWAVES][2][RWORDS + 2];      // packed read window of the wave (+ N plane)
    __shared__ uint32_t cpk[2][(CWIN + CW) / 16 + 2];     // the same bases, 2 bits each (runs are compared 16 bases at a time); the window starts at word CWIN / 16
    __shared__ ActScratch act[CWAVES];
    __shared__ CWork wk_s;                                // persistent mode: the work order as the control step published it
    const int pi = block_problem<MAXP>(B);
    const ConsParams P = B.p[pi];
    const bool coh = B.sync != nullptr;                   // persistent mode: no kernel boundary between the control step's stores and these loads
    if (coh) {
        for (int x = threadIdx.x; x < (int)(sizeof(CWork) / 4); x += blockDim.x) ((uint32_t*)&wk_s)[x] = coh_load(((const uint32_t*)P.work) + x);
        __syncthreads();
    }
    const CWork* Wp = coh ? &wk_s : P.work + wrow;
    const int mode = Wp->mode;
    if (P.work->done || mode == M_NONE) return;
    if (wrow > 0 && P.n_blocks > DIRECT_BLOCKS) return;   // (side orders exist only where the workgroups add their words to a block per order)
    if (B.step_t && threadIdx.x == 0) atomicMin(&B.step_t[2 * pi], (unsigned long long)wall_clock64());
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // window mode: n bases from column T on.  Expand mode: the parent's kept state stands at column T as well, `pre` verified bases of a window that was cut lie between it
    // and the column that branches (the replay launch a cut window used to need runs inside the expansion), the children are born at column T + pre + 1
    const int T = Wp->T, n = (mode == M_WINDOW || mode == M_EXPAND) ? Wp->n : 0;
    const int pre = mode == M_EXPAND ? n : 0;
    const int node = Wp->node, in_slot = Wp->in_slot;
    const int dual_in = Wp->dual, split_at = Wp->split_at;
    const int go0 = Wp->go[0], go1 = Wp->go[1];
    const int n_kids = (mode == M_EXPAND || mode == M_WINDOW) ? Wp->n_kids : 0;      // (window mode: the children of the expansion the window ends in, DESIGN.md section 9 "a window that branches")
    const uint8_t* Cn = P.C + (size_t)node * 2 * P.cs;
    for (int x = threadIdx.x; x < 2 * (CW + 1); x += blockDim.x) { (&lv[0][0])[x] = 0; (&le[0][0])[x] = 0; }
    for (int x = threadIdx.x; x < 2 * CW; x += blockDim.x) (&ll[0][0])[x] = 0;
    for (int x = threadIdx.x; x < CW + 1; x += blockDim.x) { lc[x] = 0; lr[x] = 0; }
    // the consensus in front of column T (a single node shows its one consensus on both sides: a child may split it) and the window
    const int w0 = T - CWIN;
    for (int x = threadIdx.x; x < 2 * (CWIN + CW); x += blockDim.x) {
        const int i = x / (CWIN + CW), y = x % (CWIN + CW), pos = w0 + y;
        uint8_t v = 0;
        if (pos >= T) v = (pos - T < n) ? Wp->spec[dual_in ? i : 0][pos - T] : 0;       // (a single node shows its one consensus on both sides here too: a child that splits it behind a cut window reads the verified bases on its second side)
        else if (pos >= 0) v = coh_load((i == 1 && dual_in && pos >= split_at) ? &Cn[(size_t)P.cs + pos] : &Cn[pos]);
        cwin[i][y] = v;
    }
    __syncthreads();
    constexpr int CPW = (CWIN + CW) / 16 + 2;
    for (int x = threadIdx.x; x < 2 * CPW; x += blockDim.x) {
        const int i = x / CPW, w = x % CPW;
        uint32_t word = 0;
        for (int b = 0; b < 16; ++b) { const int y = w * 16 + b; if (y < CWIN + n) word |= (uint32_t)(cwin[i][y] & 3u) << (b << 1); }
        cpk[i][w] = word;
    }
    __syncthreads();
    const size_t plane = (size_t)B.total;                 // reads per state plane
    for (int rr = 0; rr < P.rpw; ++rr) {
        const int r = (((int)blockIdx.x - P.first_block) * CWAVES + wave) * P.rpw + rr;
        if (r >= P.n) break;
        const size_t g = (size_t)P.first + r;
#ifdef SP_K8_TIMING
        const long long wt0 = wall_clock64();
        if (lane == 0) { act[wave].tk[0] = act[wave].tk[1] = act[wave].tk[2] = act[wave].tk[3] = 0; }
#endif
        const ReadInfo ri = B.info[g];
        ReadView rv; rv.w = ri.w; rv.np = ri.np; rv.n = ri.n;
        if (lane == 0) act[wave].staged = nullptr;                   // (the placement scratch holds no read yet; the worse-state profile shares its bytes)
        Dwfa d0, d1;
        d0.H = SP_NEG; d0.e = 0; d0.c0 = 0; d0.flags = 0; d1 = d0;
        if (mode == M_INIT) {
            if (ri.off < 0) { d0.flags = F_ACTIVE | ((P.et && rv.n == 0) ? F_FINISHED : 0); d0.H = lane == CH ? 0 : SP_NEG; }
        } else {
            const size_t p0 = state_plane(node, in_slot, 0) * plane + g;
            const ConsMeta m0 = B.meta[p0];
            d0.H = h_load(B.H, p0 * CB + lane); d0.e = m0.e; d0.c0 = m0.c0; d0.flags = m0.flags;
            if (dual_in) {
                const size_t p1 = state_plane(node, in_slot, 1) * plane + g;
                const ConsMeta m1 = B.meta[p1];
                d1.H = h_load(B.H, p1 * CB + lane); d1.e = m1.e; d1.c0 = m1.c0; d1.flags = m1.flags;
            }
        }
        // the stretch of the read around its tips: read position of diagonal 0 at column T minus the band, 384 bases from there
        int rbase = 0;
        {
            int lo = 0x7FFFFFFF;
            if (d0.flags & F_ACTIVE) { const int x = T - d0.c0 - CB; lo = x < lo ? x : lo; }
            if (dual_in && (d1.flags & F_ACTIVE)) { const int x = T - d1.c0 - CB; lo = x < lo ? x : lo; }
            if (lo == 0x7FFFFFFF || lo < 0) lo = 0;
            rbase = (lo >> 4) << 4;
            const int w_first = rbase >> 4, w_last = (rv.n + 15) >> 4;          // words [w_first, w_last + 1] exist (guard words follow the sequence)
            if (lane < RWORDS + 2) {
                const int w = w_first + lane;
                rwin[wave][0][lane] = w <= w_last + 1 ? rv.w[w] : 0u;
                rwin[wave][1][lane] = (rv.np && w <= w_last + 1) ? rv.np[w] : 0u;
            }
            spw::wave_lds_sync();
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
        // votes of state a (the other state of the read: o) for column `col`, into tally slot `slot` of consensus i
        auto vote = [&](const Dwfa& a, const Dwfa& o, bool dualrun, int i, int col, int slot) {
            if (!(a.flags & F_ACTIVE) || (a.flags & (F_FINISHED | F_LOST))) return;
            if (dualrun && (o.flags & F_ACTIVE) && !(o.flags & F_LOST) && o.e < a.e) return;      // the read follows its better consensus
            const int Tl = col - a.c0, k = lane - CH;
            const bool tip = a.H >= 0 && a.H + k == Tl;
            const int code = (tip && a.H < rv.n) ? rb(a.H) : 5;
            int dc = 0; unsigned long long word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) { const bool s = __ballot(code == b) != 0; dc += s; word |= s ? (1ull << (16 * b)) : 0ull; }
            const bool ended = __ballot(tip) != 0 && __ballot(code == 4) == 0;      // every tip sits at the end of the read
            if (lane == 0) {
                if (dc) atomicAdd(&lv[i][slot], word * vote_units(dc));
                else if (ended) atomicAdd(&le[i][slot], 12u);
            }
        };
        // one column the slow way: consensus i grows by nb_i when g_i; a late read is placed; the two states are compared
        // the extension of a new wavefront: 16 bases per step out of the packed windows where both lie inside them (window mode; the base
        // a child appends in expand mode is not in the window), base by base elsewhere
        bool use_pk = mode == M_WINDOW;                  // the packed compares may read the window (expand mode: only while the verified bases in front of the branch are replayed)
        auto extender = [&](const ConsAccess& cacc, int i) {
            return [&cacc, i, &rb, wave, rbase, w0, lane, &use_pk, &rv](Dwfa& d, int Tc) {
                const int k = lane - CH;
                for (;;) {
                    int left = rv.n - d.H; { const int l2 = Tc - (d.H + k); left = l2 < left ? l2 : left; }
                    const bool go = d.H >= 0 && left > 0;
                    int nm = 0; bool more = false;
                    if (go) {
                        const int xr = d.H - rbase, yc = d.c0 + d.H + k - w0;
                        if (use_pk && xr >= 0 && xr < RWORDS * 16 && yc >= 0 && yc < CWIN + CW) {
                            const uint32_t a = __builtin_amdgcn_alignbit(rwin[wave][0][(xr >> 4) + 1], rwin[wave][0][xr >> 4], (uint32_t)(xr & 15) << 1);
                            const uint32_t nn = __builtin_amdgcn_alignbit(rwin[wave][1][(xr >> 4) + 1], rwin[wave][1][xr >> 4], (uint32_t)(xr & 15) << 1);
                            const uint32_t b = __builtin_amdgcn_alignbit(cpk[i][(yc >> 4) + 1], cpk[i][yc >> 4], (uint32_t)(yc & 15) << 1);
                            const uint32_t xo = a ^ b, mm = ((xo | (xo >> 1)) | nn) & 0x55555555u;
                            nm = mm ? (__builtin_ctz(mm) >> 1) : 16;
                            nm = nm < left ? nm : left;
                            more = nm == 16 && left > 16;
                        } else {
                            const int x = rb(d.H);
                            nm = (x < 4 && x == cacc.at(d.c0 + d.H + k)) ? 1 : 0;
                            more = nm == 1 && left > 1;
                        }
                        d.H += nm;
                    }
                    if (!__ballot(more)) break;
                }
            };
        };
        auto column = [&](Dwfa& a0, Dwfa& a1, bool dualrun, int g0, int g1, int nb0, int nb1, int len, const ConsAccess& c0a, const ConsAccess& c1a) {
            if (g0) {
                if (a0.flags & F_ACTIVE) { if (!(a0.flags & (F_FINISHED | F_LOST))) dwfa_push_t(a0, rv.n, rb, extender(c0a, c0a.i), len - a0.c0, nb0, P.et, lane); }
                else if (ri.off == len) { a0 = activate_late(rv, c0a, &act[wave], B.memo + 2 * ((size_t)wrow * plane + g), ri.off0, len, P.window, P.cmp_len, P.et, lane); spw::wave_lds_sync(); }
            }
            if (dualrun && g1) {
                if (a1.flags & F_ACTIVE) { if (!(a1.flags & (F_FINISHED | F_LOST))) dwfa_push_t(a1, rv.n, rb, extender(c1a, c1a.i), len - a1.c0, nb1, P.et, lane); }
                else if (ri.off == len) { a1 = activate_late(rv, c1a, &act[wave], B.memo + 2 * ((size_t)wrow * plane + g), ri.off0, len, P.window, P.cmp_len, P.et, lane); spw::wave_lds_sync(); }
            }
            if (dualrun) {
                const int both = (a0.flags & F_ACTIVE) && (a1.flags & F_ACTIVE) && !(a0.flags & F_LOST) && !(a1.flags & F_LOST);
                if (both) {
                    if (a0.e > a1.e + P.delta) a0.flags |= F_LOST;
                    else if (a1.e > a0.e + P.delta) a1.flags |= F_LOST;
                }
            }
        };
        // what the read adds to a FINAL cost over its running cost when its consensuses end at len0 / len1 (no early termination)
        auto final_extra = [&](const Dwfa& a0, const Dwfa& a1, bool dualrun, int len0, int len1) -> int {
            if (P.et) return 0;
            int best = -1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !dualrun) continue;
                const Dwfa& a = i ? a1 : a0;
                if (!(a.flags & F_ACTIVE)) continue;
                int s = a.e;
                if (!(a.flags & F_LOST)) {
                    const int k = lane - CH, Tl = (i ? len1 : len0) - a.c0;
                    int rest = (a.H >= 0 && a.H + k == Tl) ? rv.n - a.H : (1 << 30);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(rest, o); rest = other < rest ? other : rest; }
                    if (rest < (1 << 30)) s += rest;
                }
                if (best < 0 || s < best) best = s;
            }
            return best < 0 ? 0 : best - read_cost(a0, a1, dualrun);
        };
        auto store = [&](const Dwfa& a, int nd, int slot, int i) {
            const size_t p = state_plane(nd, slot, i) * plane + g;
            if (lane == 0) { ConsMeta m; m.e = a.e; m.c0 = a.c0; m.flags = a.flags; m.pad = 0; B.meta[p] = m; }
            h_store(B.H, p * CB + lane, a.H);
        };
        ConsAccess ca0, ca1;
        ca0.win = &cwin[0][0]; ca0.w0 = w0; ca0.C = Cn; ca0.cap = P.cs; ca0.split_at = split_at; ca0.i = 0; ca0.ov_pos = -1; ca0.ov_base = 0; ca0.pk = &cpk[0][0];
        ca1 = ca0; ca1.win = &cwin[1][0]; ca1.i = 1; ca1.pk = &cpk[1][0];

        // one pass of a read through the n bases of the window behind column T.  quiet: the pass only moves the state (the verified bases of a cut window in front of an
        // expansion: their votes and costs are on the parent's tape already)
        auto win_pass = [&](Dwfa& d0, Dwfa& d1, const bool dualrun, const int go0, const int go1, const int n, const bool quiet) {
        if (!quiet && mode == M_INIT) { vote(d0, d1, false, 0, 0, 0); }
        // number of leading bases on which read[x0 ..] (position relative to the staged window) and the window's bases [j ..] agree
        // The blocks of 16 bases are compared side by side, one lane each (a window has at most CW / 16 + 1 of them): one round of LDS reads
        // instead of one per block.
        auto match_run = [&](int x0, int i, int j0, int len) -> int {
            if (x0 < 0 || x0 + len + 16 > RWORDS * 16) return 0;
            const int nblk = (len + 15) >> 4;                                  // <= CW / 16
            const int blk = lane < nblk ? lane : 0;
            const int x = x0 + (blk << 4), jj = j0 + (blk << 4);
            const uint32_t a = __builtin_amdgcn_alignbit(rwin[wave][0][(x >> 4) + 1], rwin[wave][0][x >> 4], (uint32_t)(x & 15) << 1);
            const uint32_t nn = __builtin_amdgcn_alignbit(rwin[wave][1][(x >> 4) + 1], rwin[wave][1][x >> 4], (uint32_t)(x & 15) << 1);
            const uint32_t b = __builtin_amdgcn_alignbit(cpk[i][CWIN / 16 + (jj >> 4) + 1], cpk[i][CWIN / 16 + (jj >> 4)], (uint32_t)(jj & 15) << 1);
            const uint32_t xr = a ^ b, mm = ((xr | (xr >> 1)) | nn) & 0x55555555u;
            int run = mm ? (__builtin_ctz(mm) >> 1) : 16;
            const int room = len - (blk << 4);
            run = run < room ? run : room;
            const unsigned long long stop = __ballot(lane < nblk && run < 16 && run < room);
            if (!stop) return len;
            const int t = __builtin_ctzll(stop);
            return (t << 4) + __builtin_amdgcn_readlane(run, t);
        };
        // The worse state of a read that follows two consensuses goes through the whole window FIRST, wavefront by wavefront (as a late read
        // catches up: dwfa_catchup_t), not column by column.  While it has more edits than the other state it has no say in any column (vote),
        // so all the columns need of it is its edit count -- for the comparison that stops tracking it (dual_max_ed_delta) and for the moment
        // it would draw level -- and a wavefront with e edits that reaches column c names that count for every column up to c.  A state 3 %
        // off its consensus (the other gene copy, the other haplotype) costs ten wavefronts per window this way instead of a hundred and
        // fifty single-column pushes, and no longer cuts the clean runs of the good state short.  If the other state catches up with it inside the
        // window, the state is rebuilt at that column from its copy and the two go on column by column.
        // (its edit count after j pushes, 16 bits each, lies in the wave's placement scratch: a read with both states placed has no use for that)
        static_assert(sizeof(((ActScratch*)nullptr)->rcache) >= 2 * (CW + 1), "edit-count profile");
        uint16_t* const ewp = reinterpret_cast<uint16_t*>(act[wave].rcache);
        auto bulk_push = [&](Dwfa& d, int i, const ConsAccess& cacc, int Tl0, int cols, int jbase) -> bool {   // jbase >= 0: the counts go to ewp[jbase ..]
            auto ext = extender(cacc, i);
            const int k = lane - CH, span = Tl0 + cols;
            int reached = Tl0;
            const bool record = jbase >= 0;
            if (record && lane == 0) ewp[jbase] = (uint16_t)d.e;
            for (;;) {
                ext(d, span);
                int far = spw::wave_max(d.H >= 0 ? d.H + k : -1);
                far = far < span ? far : span;
                if (record) for (int c = reached + 1 + lane; c <= far; c += SP_WAVE) ewp[jbase + c - Tl0] = (uint16_t)d.e;
                reached = far > reached ? far : reached;
                if (reached >= span) break;
                const int c = d.H, up = spw::from_lower(d.H, SP_NEG), dn = spw::from_upper(d.H, SP_NEG);
                int best = SP_NEG;
                if (c >= 0 && c < rv.n && c + k < span) best = c + 1;
                if (up >= 0 && up + k <= span && up + k >= 0 && up > best) best = up;
                if (dn >= 0 && dn < rv.n && dn + 1 + k >= 0 && dn + 1 > best) best = dn + 1;
                if (!__ballot(best >= 0)) return false;
                d.H = best; d.e += 1;
            }
            spw::wave_lds_sync();
            return true;
        };
        int bulk = -1;                                                       // the state that went ahead (0 / 1), none: -1
        int wsave_H = 0, wsave_e = 0, wsave_j = 0;                           // its state at the column it left from (c0 and flags do not change on the way)
        int want_bulk = BULK_MARGIN;                                         // > 0: at the head of the loop, send the worse state ahead if it is that many edits behind
        int j = 0;
#ifdef SP_K8_TIMING
        int slow_cols = 0, multi_tip = 0, zero_run = 0, fast_iters = 0;
        long long tb_fast = 0, tb_col = 0, tb_vote = 0, tb_mark = 0, tb_bulk = 0;
#ifdef SP_K8_DBG_PARTS
        const long long tb_pre = wall_clock64() - wt0;                       // (parts build: everything on the 100 MHz wall clock)
#define K8_T0() tb_mark = wall_clock64()
#define K8_T(acc) do { const long long _n = wall_clock64(); acc += _n - tb_mark; tb_mark = _n; } while (0)
#else
#define K8_T0() tb_mark = clock64()
#define K8_T(acc) do { const long long _n = clock64(); acc += _n - tb_mark; tb_mark = _n; } while (0)
#endif
#else
#define K8_T0() do { } while (0)
#define K8_T(acc) do { } while (0)
#endif
        while (j < n) {
            // (one place for it: at the start of the window, and behind the column that placed the read -- its wrong state would otherwise cost a slow
            //  column every few bases until the window ends)
            K8_T0();
            if (want_bulk > 0) {
                const int margin = want_bulk; want_bulk = 0;
                if (!quiet && mode == M_WINDOW && dualrun && go0 && go1 && n - j >= 8 && (d0.flags & F_ACTIVE) && (d1.flags & F_ACTIVE) &&
                    !((d0.flags | d1.flags) & (F_FINISHED | F_LOST)) && (d0.e >= d1.e + margin || d1.e >= d0.e + margin)) {
                    const int wi = d0.e > d1.e ? 0 : 1;
                    Dwfa& w = wi ? d1 : d0;
                    const int Tl0 = T + j - w.c0;
                    if ((!P.et || rv.n > Tl0 + (n - j) + CB) && w.e < 60000) {      // (early termination: a read that could end inside the window freezes at that column)
                        wsave_H = w.H; wsave_e = w.e; wsave_j = j;
                        if (bulk_push(w, wi, wi ? ca1 : ca0, Tl0, n - j, j)) { bulk = wi; w.e = wsave_e; }   // w.e follows the columns: its count after j pushes
                        else { w.H = wsave_H; w.e = wsave_e; }
                    }
                }
            }
            K8_T(tb_bulk);
            // A consensus whose state has ONE tip that keeps matching moves nothing but that tip: such a clean run is applied in one go
            // (the tip's position grows by m, the votes of the m columns are the m read bases behind it, one lane per column).
            int m = n - j;
            int tl0 = -1, th0 = 0, tl1 = -1, th1 = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !dualrun) continue;
                if (!(i ? go1 : go0)) continue;
                if (i == bulk) continue;
                if (m == 0) continue;                                       // (the other state has already ruled a clean run out: the column goes the slow way whatever this one would say)
                const Dwfa& a = i ? d1 : d0;
                if (a.flags & F_ACTIVE) {
                    if (a.flags & (F_FINISHED | F_LOST)) continue;
                    const int Tl = T + j - a.c0, k = lane - CH;
                    const unsigned long long tm = __ballot(a.H >= 0 && a.H + k == Tl);
                    if (__builtin_popcountll(tm) != 1) {
#ifdef SP_K8_TIMING
                        multi_tip += 1;
#endif
                        m = 0; continue; }
                    const int tl = __builtin_ctzll(tm), h = __builtin_amdgcn_readlane(a.H, tl);
                    int room = rv.n - h; room = room < m ? room : m;
                    const int run = room > 0 ? match_run(h - rbase, i, j, room) : 0;
                    m = run < m ? run : m;
                    if (i) { tl1 = tl; th1 = h; } else { tl0 = tl; th0 = h; }
                } else if (ri.off > T + j && ri.off <= T + n) {
                    const int before = ri.off - T - 1 - j;                  // pushes before the one that places the read
                    m = before < m ? before : m;
                }
            }
            if (m > 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int tl = i ? tl1 : tl0, th = i ? th1 : th0;
                    if (tl < 0) continue;
                    Dwfa& a = i ? d1 : d0; const Dwfa& o = i ? d0 : d1;
                    bool speaks = true;
                    if (dualrun && (o.flags & F_ACTIVE) && !(o.flags & F_LOST) && o.e < a.e) speaks = false;
                    if (speaks && !quiet) for (int x = lane + 1; x <= m; x += SP_WAVE) {   // lane l names the votes after l + 1, l + 65, ... of the m pushes
                        const int pos = th + x;
                        if (pos < rv.n) { const int code = rb(pos); if (code < 4) atomicAdd(&lv[i][j + x], 12ull << (16 * code)); }
                        else if (!P.et) atomicAdd(&le[i][j + x], 12u);       // (with early termination the read is finished by that push)
                    }
                    if (lane == tl) a.H += m;
                    if (P.et && th + m == rv.n) a.flags |= F_FINISHED;
                }
                if (bulk >= 0) {
                    // the state that went ahead: is it dropped in one of these columns?  (the other state's count does not move in a clean run)
                    Dwfa& w = bulk ? d1 : d0; const Dwfa& b = bulk ? d0 : d1;
                    if (!(w.flags & F_LOST)) {
                        int hit = -1;
                        if (!(b.flags & F_LOST)) {
                            const int thr = b.e + P.delta;
                            for (int base = 0; base < m; base += SP_WAVE) {
                                const int x = j + 1 + base + lane;
                                const unsigned long long over = __ballot(x <= j + m && (int)ewp[x] > thr);
                                if (over) { hit = j + 1 + base + __builtin_ctzll(over); break; }
                            }
                        }
                        if (hit >= 0) { w.flags |= F_LOST; w.e = (int)ewp[hit]; } else w.e = (int)ewp[j + m];
                    }
                }
                j += m;
                K8_T(tb_fast);
#ifdef SP_K8_TIMING
                fast_iters += 1;
#endif
                continue;
            }
            K8_T(tb_fast);
#ifdef SP_K8_TIMING
            slow_cols += 1;
#endif
            const int before = read_cost(d0, d1, dualrun);
            const int active_before = (d0.flags & F_ACTIVE) + (d1.flags & F_ACTIVE);
            if (bulk >= 0) {
                // only the other state is pushed; the one that went ahead shows its count at the new column to the comparison in column()
                Dwfa& w = bulk ? d1 : d0; const Dwfa& b = bulk ? d0 : d1;
                if (!(w.flags & F_LOST)) w.e = (int)ewp[j + 1];
                column(d0, d1, dualrun, bulk != 0, bulk != 1, cwin[0][CWIN + j], cwin[1][CWIN + j], T + j + 1, ca0, ca1);
                if (!(w.flags & F_LOST) && ((b.flags & F_LOST) || w.e <= b.e)) {
                    // the other state has caught up (or is no longer tracked): from this column on this one has a say.  The state as it stands at
                    // this column, from its copy
                    w.H = wsave_H; w.e = wsave_e;
                    (void)bulk_push(w, bulk, bulk ? ca1 : ca0, T + wsave_j - w.c0, j + 1 - wsave_j, -1);
                    bulk = -1;
                }
            } else
                column(d0, d1, dualrun, go0, go1, cwin[0][CWIN + j], cwin[1][CWIN + j], T + j + 1, ca0, ca1);
            const int grow = read_cost(d0, d1, dualrun) - before;
            if (grow && lane == 0 && !quiet) atomicAdd(&lc[j + 1], (uint32_t)grow);
            K8_T(tb_col);
            if (!quiet && go0 && bulk != 0) vote(d0, d1, dualrun, 0, T + j + 1, j + 1);
            if (!quiet && dualrun && go1 && bulk != 1) vote(d1, d0, dualrun, 1, T + j + 1, j + 1);
            K8_T(tb_vote);
            j += 1;
            if (bulk < 0 && (d0.flags & F_ACTIVE) + (d1.flags & F_ACTIVE) != active_before) want_bulk = BULK_MARGIN_PLACED;   // a read placed in this column
        }
        // lookahead: the bases behind every tip (at most two tips per consensus speak) predict the columns after the window
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (quiet) continue;
            if (mode == M_WINDOW && n_kids > 0) continue;                        // (the lookahead words belong to the children of the expansion behind the window)
            if (i == 1 && !dualrun) continue;
            if (mode != M_INIT && !(i ? go1 : go0)) continue;
            const Dwfa& a = i ? d1 : d0; const Dwfa& o = i ? d0 : d1;
            if (!(a.flags & F_ACTIVE) || (a.flags & (F_FINISHED | F_LOST))) continue;
            if (dualrun && (o.flags & F_ACTIVE) && !(o.flags & F_LOST) && o.e < a.e) continue;
            const int Tl = T + n - a.c0, k = lane - CH;
            unsigned long long tips = __ballot(a.H >= 0 && a.H + k == Tl && a.H < rv.n);
            for (int cnt = 0; tips && cnt < 2; ++cnt) {
                const int tl = __builtin_ctzll(tips); tips &= tips - 1;
                const int h = __builtin_amdgcn_readlane(a.H, tl);
                for (int x = lane; x < CW - 1 && h + 1 + x < rv.n; x += SP_WAVE) {
                    const int b = rb(h + 1 + x);
                    if (b < 4) atomicAdd(&ll[i][x], 1ull << (16 * b));
                }
            }
        }
        {
            const int len0 = go0 || mode == M_INIT ? T + n : coh_load(&P.nodes[node].len[0]), len1 = go1 ? T + n : coh_load(&P.nodes[node].len[1]);
            const int extra = final_extra(d0, d1, dualrun, len0, len1);
            if (extra && lane == 0 && !quiet) atomicAdd(&lr[n], (uint32_t)extra);
        }
        // the new state goes to the node's other slot (the root's first state: its slot 0)
        if (!quiet) {
            const int out_slot = mode == M_INIT ? in_slot : in_slot ^ 1;
            store(d0, node, out_slot, 0);
            if (dualrun) store(d1, node, out_slot, 1);
        }
        spw::wave_lds_sync();
#ifdef SP_K8_TIMING
#ifdef SP_K8_DBG_PARTS
        if (!quiet && B.dbg && lane == 0 && Wp->pad < SP_K8_DBG_LAUNCHES / 2 && g < SP_K8_DBG_READS) {
#else
        if (!quiet && B.dbg && lane == 0 && Wp->pad < SP_K8_DBG_LAUNCHES && g < SP_K8_DBG_READS) {
#endif
            const unsigned long long dt = (unsigned long long)(wall_clock64() - wt0);
            const unsigned long long placed = ri.off > T && ri.off <= T + n;
            // [launch][read]: ticks (24 bits, 100 MHz) | slow columns (9) | multi-tip events (9) | a late read was placed (1) | mode (2) | window bases (9) | column pushes, clock64 / 1024 (10)
            auto cl = [](long long v, long long cap) { return (unsigned long long)(v > cap ? cap : v); };
#ifdef SP_K8_DBG_PARTS
            // (variant: two words per read.  The second: where the wave's time went, 20 ns units -- loads in front of the loop (10 bits) / clean runs (12) / column pushes (12) /
            //  votes (12) / the worse state sent ahead (12) / clean-run iterations (6))
            B.dbg[((size_t)Wp->pad * SP_K8_DBG_READS + g) * 2] = cl((long long)dt, 0xFFFFFF) | (cl(slow_cols, 511) << 24) | (cl(multi_tip, 511) << 33) | (placed << 42) | ((unsigned long long)mode << 43) |
                                                            (cl(n, 511) << 45);
            B.dbg[((size_t)Wp->pad * SP_K8_DBG_READS + g) * 2 + 1] = cl(tb_pre >> 1, 1023) | (cl(tb_fast >> 1, 4095) << 10) | (cl(tb_col >> 1, 4095) << 22) | (cl(tb_vote >> 1, 4095) << 34) |
                                                            (cl(tb_bulk >> 1, 4095) << 46) | (cl(fast_iters, 63) << 58);
            (void)tb_pre;
#elif defined(SP_K8_DBG_EDITS)
            // (variant: the edit counts of the two states instead of the multi-tip events and the column-push clocks)
            // (variant: where a placement's time goes instead of the multi-tip events and the column-push clocks: start search and catch-up, 0.32 us units;
            //  staging + packing in the slow-column field)
            B.dbg[(size_t)Wp->pad * SP_K8_DBG_READS + g] = cl((long long)dt, 0xFFFFFF) | (cl((act[wave].tk[0] + act[wave].tk[2]) >> 5, 511) << 24) | (cl(act[wave].tk[1] >> 5, 511) << 33) | (placed << 42) |
                                                            ((unsigned long long)mode << 43) | (cl(n, 511) << 45) | (cl(act[wave].tk[3] >> 5, 1023) << 54);
#else
            B.dbg[(size_t)Wp->pad * SP_K8_DBG_READS + g] = cl((long long)dt, 0xFFFFFF) | (cl(slow_cols, 511) << 24) | (cl(multi_tip, 511) << 33) | (placed << 42) | ((unsigned long long)mode << 43) |
                                                            (cl(n, 511) << 45) | (cl(tb_col >> 10, 1023) << 54);
#endif
            (void)tb_fast; (void)tb_vote; (void)tb_bulk; (void)fast_iters;
        }
#endif
        };
        // one push per child, every child from the parent's state (d0, d1) at column TB; child k's words go to slot s0 + k
        auto expand_kids = [&](const Dwfa& d0, const Dwfa& d1, const int TB, const int s0) {
            const int base_cost = read_cost(d0, d1, dual_in != 0);
            for (int k = 0; k < n_kids; ++k) {
                Dwfa e0 = d0, e1 = d1;
                const int kb0 = Wp->kid_base[k][0], kb1 = Wp->kid_base[k][1], ksplit = Wp->kid_split[k];
                if (ksplit) e1 = e0;                                         // consensus 2 starts as a copy of consensus 1
                const bool kdual = dual_in || ksplit;
                ConsAccess k0 = ca0, k1 = ca1;
                k0.ov_pos = kb0 >= 0 ? TB : -1; k0.ov_base = kb0;
                k1.ov_pos = kb1 >= 0 ? TB : -1; k1.ov_base = kb1;
                if (ksplit) { k1.i = 0; }                                      // (its prefix is consensus 1's; only the new base differs)
                column(e0, e1, kdual, kb0 >= 0, kb1 >= 0, kb0, kb1, TB + 1, k0, k1);
                if (kb0 >= 0) vote(e0, e1, kdual, 0, TB + 1, s0 + k);
                if (kdual && kb1 >= 0) vote(e1, e0, kdual, 1, TB + 1, s0 + k);
                if (n_kids <= KID_LA_KIDS && (P.n < 256 || (r & 3) == 0)) {
                    // the child's lookahead votes (as behind a window: the bases behind every tip, at most two tips per consensus speak), KID_LA columns of them; of a
                    // large problem every fourth read speaks -- a speculated base only has to be the likely one, the exact votes of the window decide
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i == 1 && !kdual) continue;
                        if ((i ? kb1 : kb0) < 0) continue;
                        const Dwfa& a = i ? e1 : e0; const Dwfa& o = i ? e0 : e1;
                        if (!(a.flags & F_ACTIVE) || (a.flags & (F_FINISHED | F_LOST))) continue;
                        if (kdual && (o.flags & F_ACTIVE) && !(o.flags & F_LOST) && o.e < a.e) continue;
                        const int Tl = TB + 1 - a.c0, kd = lane - CH;
                        unsigned long long tips = __ballot(a.H >= 0 && a.H + kd == Tl && a.H < rv.n);
                        unsigned long long* const kl = &ll[0][0] + (size_t)(k * 2 + i) * KID_LA;
                        for (int cnt = 0; tips && cnt < 2; ++cnt) {
                            const int tl = __builtin_ctzll(tips); tips &= tips - 1;
                            const int h = __builtin_amdgcn_readlane(a.H, tl);
                            for (int x = lane; x < KID_LA - 1 && h + 1 + x < rv.n; x += SP_WAVE) {
                                const int b = rb(h + 1 + x);
                                if (b < 4) atomicAdd(&kl[x], 1ull << (16 * b));
                            }
                        }
                    }
                }
                const int grow = read_cost(e0, e1, kdual) - base_cost;
                const int len0 = kb0 >= 0 ? TB + 1 : (go0 ? TB : coh_load(&P.nodes[node].len[0]));
                const int len1 = kb1 >= 0 ? TB + 1 : (go1 ? TB : coh_load(&P.nodes[node].len[1]));
                const int extra = final_extra(e0, e1, kdual, len0, len1);
                if (lane == 0) { if (grow) atomicAdd(&lc[s0 + k], (uint32_t)grow); if (extra) atomicAdd(&lr[s0 + k], (uint32_t)extra); }
                const int kn = Wp->kid_node[k];
                store(e0, kn, 0, 0);
                if (kdual) store(e1, kn, 0, 1);
            }
            spw::wave_lds_sync();
        };
        if (mode == M_EXPAND) {
            // behind a cut window: first the verified bases in front of the branch (no votes: they are on the parent's tape), then the children
            if (pre > 0) { use_pk = true; win_pass(d0, d1, dual_in != 0, Wp->pre_go[0], Wp->pre_go[1], pre, true); use_pk = false; }
            expand_kids(d0, d1, T + pre, 0);
#ifdef SP_K8_DBG_PARTS
            if (B.dbg && lane == 0 && Wp->pad < SP_K8_DBG_LAUNCHES / 2 && g < SP_K8_DBG_READS) {
                const unsigned long long dt = (unsigned long long)(wall_clock64() - wt0);
                B.dbg[((size_t)Wp->pad * SP_K8_DBG_READS + g) * 2] = (dt > 0xFFFFFF ? 0xFFFFFFull : dt) | ((unsigned long long)(n_kids > 511 ? 511 : n_kids) << 24) | (3ull << 43) | ((unsigned long long)(pre > 511 ? 511 : pre) << 45);
                B.dbg[((size_t)Wp->pad * SP_K8_DBG_READS + g) * 2 + 1] = 0;
            }
#endif
            continue;
        }

        win_pass(d0, d1, dual_in != 0, go0, go1, n, false);
        // a window that branches at its end (the control step foresaw the branch from the lookahead votes): the children of that expansion in the same launch, from the state
        // the window has just left at column T + n -- if the window stands and the exact votes of that column name these very children, no launch is needed for them
        if (n_kids > 0) {
            use_pk = false;
            if (lane == 0) act[wave].staged = nullptr;                        // (the window pass may have written a worse state's edit-count profile over the staged read: a child that places the read stages it again)
            spw::wave_lds_sync();
            expand_kids(d0, d1, T + n, n + 1);
        }
    }
    __syncthreads();
    const UsedWords uw(mode, n, n_kids);
    if (P.n_blocks <= DIRECT_BLOCKS) {
        // a problem of few workgroups: every workgroup adds its words to the problem's one block (fire-and-forget atomics, zeros skipped; a 16-bit field of
        // 128 workgroups cannot carry into the next); the control step then reads one block instead of up to 128 -- its sums were half of its time
        const size_t blk = (size_t)P.acc_block + (size_t)wrow;
        for (int x = threadIdx.x; x < 2 * uw.used; x += blockDim.x) {
            const int i = x / uw.used, e = i * (CW + 1) + x % uw.used;
            const unsigned long long v = (&lv[0][0])[e]; const uint32_t w = (&le[0][0])[e];
            if (v) atomicAdd(&B.PV[blk * 2 * (CW + 1) + e], v);
            if (w) atomicAdd(&B.PE[blk * 2 * (CW + 1) + e], w);
        }
        if (uw.has_la) for (int x = threadIdx.x; x < 2 * CW; x += blockDim.x) { const unsigned long long v = (&ll[0][0])[x]; if (v) atomicAdd(&B.PL[blk * 2 * CW + x], v); }
        for (int x = threadIdx.x; x < uw.used; x += blockDim.x) {
            if (lc[x]) atomicAdd(&B.PC[blk * (CW + 1) + x], lc[x]);
            if (lr[x]) atomicAdd(&B.PR[blk * (CW + 1) + x], lr[x]);
        }
        if (B.step_t && threadIdx.x == 0) atomicMax(&B.step_t[2 * pi + 1], (unsigned long long)wall_clock64());
        return;
    }
    for (int x = threadIdx.x; x < 2 * uw.used; x += blockDim.x) {
        const int i = x / uw.used, e = i * (CW + 1) + x % uw.used;
        B.PV[(size_t)blockIdx.x * 2 * (CW + 1) + e] = (&lv[0][0])[e];
        B.PE[(size_t)blockIdx.x * 2 * (CW + 1) + e] = (&le[0][0])[e];
    }
    if (uw.has_la) for (int x = threadIdx.x; x < 2 * CW; x += blockDim.x) B.PL[(size_t)blockIdx.x * 2 * CW + x] = (&ll[0][0])[x];
    for (int x = threadIdx.x; x < uw.used; x += blockDim.x) { B.PC[(size_t)blockIdx.x * (CW + 1) + x] = lc[x]; B.PR[(size_t)blockIdx.x * (CW + 1) + x] = lr[x]; }
    if (B.step_t && threadIdx.x == 0) atomicMax(&B.step_t[2 * pi + 1], (unsigned long long)wall_clock64());
}

template <int MAXP>
__global__ void __launch_bounds__(CWAVES * SP_WAVE, SP_K8_MIN_WAVES) cons_step_kernel(ConsBatchT<MAXP> B) { cons_step_body<MAXP>(B, (int)blockIdx.y); }
// The same body at two waves per SIMD (one workgroup per CU): 240 registers and no spill code, where the first instantiation keeps 128 and spills 73 to scratch -- 11 MB of dirty
// scratch lines per launch that the launch's end writes back (profiles/r05/counters_cons_step.json).  For a batch whose workgroups fit the device in one round either way (a
// single sample's searches: <= one workgroup per CU) it is the faster one (`*4+*68/*1` alone 165 -> 155 ms; a rank's 32-sample share 0.138 -> 0.133 s); a batch with more workgroups
// than CUs (an HLA gene of 5,000 reads: 625) needs the two workgroups per CU of the first (its step 111 -> 133 us with this one).
template <int MAXP>
__global__ void __launch_bounds__(CWAVES * SP_WAVE, 2) cons_step_wide_kernel(ConsBatchT<MAXP> B) { cons_step_body<MAXP>(B, (int)blockIdx.y); }

// sums the vote words of CLUSTER consecutive workgroups of a problem (several hundred workgroups would otherwise be summed by the one
// workgroup of the control kernel, word by word from memory).  RSLICES workgroups per cluster, each a slice of the words: with 256-column
// windows a workgroup leaves 12 KB of words behind, 7.7 MB per launch of a 5,000-read problem, and ten workgroups would be alone with them.
// A thread sums one word over RGROUP members (all their loads issued before the first is used), the CLUSTER / RGROUP threads of a word sit in
// neighbouring lanes and add up through DPP.
template <int MAXP>
__global__ void __launch_bounds__(512) cons_reduce_kernel(ConsBatchT<MAXP> B) {
    const int cluster = (int)blockIdx.x / RSLICES, slice = (int)blockIdx.x % RSLICES;
    int pi = 0;
    if constexpr (MAXP == 0) pi = B.cluster_prob[cluster];
    else {
#pragma unroll
        for (int i = 1; i < MAXP; ++i) if (i < B.n_prob && cluster >= B.p[i].first_cluster) pi = i;
    }
    const ConsParams P = B.p[pi];
    const int mode = P.work->mode;
    if (P.work->done || mode == M_NONE) return;
    const UsedWords uw(mode, mode == M_WINDOW ? P.work->n : 0, mode == M_EXPAND ? P.work->n_kids : 0);
    if (P.n_blocks <= DIRECT_BLOCKS) return;                                 // the control kernel sums the words of so few workgroups itself
    const int cl = cluster - P.first_cluster;
    const int members = P.n_blocks - cl * CLUSTER < CLUSTER ? P.n_blocks - cl * CLUSTER : CLUSTER;
    const size_t blk0 = (size_t)P.first_block + (size_t)cl * CLUSTER;
    constexpr int NG = CLUSTER / RGROUP;
    const int per = (uw.total + RSLICES - 1) / RSLICES;
    const int c_lo = slice * per, c_hi = c_lo + per < uw.total ? c_lo + per : uw.total;
    for (int idx = threadIdx.x; idx < (c_hi - c_lo) * NG; idx += blockDim.x) {
        const int o = uw.at(c_lo + idx / NG), m0 = (idx % NG) * RGROUP;
        uint32_t v[RGROUP];
#pragma unroll
        for (int m = 0; m < RGROUP; ++m) v[m] = m0 + m < members ? block_word(B, blk0 + m0 + m, o) : 0u;
        uint32_t sum = 0;
#pragma unroll
        for (int m = 0; m < RGROUP; ++m) sum += v[m];
        // the NG threads of a word are neighbouring lanes (NG divides 64 and the block size: they never straddle a wavefront)
#pragma unroll
        for (int d = 1; d < NG; d <<= 1) sum += __shfl_xor(sum, d);
        if (idx % NG == 0) B.Q[(size_t)cluster * QE + o] = sum;
    }
}

// heaviest base of a vote column first (ties to the lower code)
struct ColVotes { uint32_t w[4], end; };

// candidates of a column (oracle/consensus.c: candidates): 0 = the consensus stops there
__device__ __forceinline__ int col_candidates(const uint32_t* w5, int col, int cap, int et, int min_count, double min_af, int out[4]) {
    if (col >= cap) return 0;                                                           // out of room: the consensus is cut at cap
    int order[4] = { 0, 1, 2, 3 };
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) if (w5[order[b]] > w5[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
    const uint32_t w1 = w5[order[0]], total = w5[0] + w5[1] + w5[2] + w5[3];
    const bool go = et ? w1 > 0 : (total > w5[4] && w1 > 0);
    if (!go) return 0;
    uint32_t need = 12u * (uint32_t)min_count; if (w1 < need) need = w1;
    int n = 0;
    out[n++] = order[0];
#pragma unroll
    for (int a = 1; a < 4; ++a) { const uint32_t w = w5[order[a]]; if (w > 0 && w >= need && (double)w >= min_af * (double)total) out[n++] = order[a]; }
    return n;
}

// the same list without an array (an `out[n++]` lives in scratch memory: a round trip to memory per access in the one lane that decides): the bases 2 bits each, heaviest
// first, in `packed`; the weights are ordered by the very compare-and-swap sequence of the loops above
__device__ __forceinline__ int col_candidates_packed(const uint32_t* w5, int col, int cap, int et, int min_count, double min_af, uint32_t& packed) {
    packed = 0;
    if (col >= cap) return 0;
    uint32_t wa = w5[0], wb = w5[1], wc = w5[2], wd = w5[3]; uint32_t ca = 0, cb = 1, cc = 2, cd = 3;
#define SP_CSW(x, y, cx, cy) if (y > x) { const uint32_t tw_ = x; x = y; y = tw_; const uint32_t tc_ = cx; cx = cy; cy = tc_; }
    SP_CSW(wa, wb, ca, cb) SP_CSW(wa, wc, ca, cc) SP_CSW(wa, wd, ca, cd) SP_CSW(wb, wc, cb, cc) SP_CSW(wb, wd, cb, cd) SP_CSW(wc, wd, cc, cd)
#undef SP_CSW
    const uint32_t total = wa + wb + wc + wd;
    const bool go = et ? wa > 0 : (total > w5[4] && wa > 0);
    if (!go) return 0;
    uint32_t need = 12u * (uint32_t)min_count; if (wa < need) need = wa;
    int n = 1; packed = ca;
    const double floor_w = min_af * (double)total;
    if (wb > 0 && wb >= need && (double)wb >= floor_w) { packed |= cb << (2 * n); ++n; }
    if (wc > 0 && wc >= need && (double)wc >= floor_w) { packed |= cc << (2 * n); ++n; }
    if (wd > 0 && wd >= need && (double)wd >= floor_w) { packed |= cd << (2 * n); ++n; }
    return n;
}

// ------------------------------------------------------------------------------------------------------------------------------
// the control step: one workgroup per problem sums the vote words of the problem's workgroups, takes the result of the last step
// into the node table and plays the search forward until the best node needs the next launch
// ------------------------------------------------------------------------------------------------------------------------------
template <int MAXP>
__device__ __forceinline__ void cons_control_body(const ConsBatchT<MAXP>& B) {
    extern __shared__ uint8_t proc[];                     // nodes expanded per length (cap + 2 bytes, padded to 16)
    __shared__ uint32_t acc[QE];                          // the sums over all workgroups of the problem for ONE order, in the order of the cluster sums:
    uint32_t (*sv)[CW + 1][5] = reinterpret_cast<uint32_t (*)[CW + 1][5]>(acc);              // exact votes: w[4], end
    uint32_t (*sl)[CW][4] = reinterpret_cast<uint32_t (*)[CW][4]>(acc + QSV);                // lookahead votes
    uint32_t* sc = acc + QSV + QSL; uint32_t* sr = sc + (CW + 1);                              // cost growth / final-cost extra
    __shared__ CNode nh[NQ];
    __shared__ CWork wks[NWORK];                          // the orders of the last step on the way in, those of the next step on the way out; wks[0] is the search's own
    __shared__ CSearch ss;
    __shared__ int copy_from[NWORK], copy_len[NWORK], need_la[NWORK], replay_of[NWORK];
    // the children a window that branched at its end brought with it (take_result): their consensus rows are written at the end of the pass
    struct KidCopy { int from, T, n, nk, dual; int8_t node[4], b0[4], b1[4], sp[4]; uint8_t spec[2][CW]; };
    static_assert(KID_LA_KIDS <= 4, "KidCopy");
    __shared__ KidCopy kcopy[NWORK];
    const int pi = blockIdx.x;
    const ConsParams P = B.p[pi];
    const int tid = threadIdx.x;
    const bool coh = B.sync != nullptr;                   // persistent mode: what the step workgroups read next is stored write-through, their words are fetched where the atomics ran
    // side orders: only where every order has a word block of its own (problems whose workgroups add their words up), and not between persistent kernels (one row of resident workgroups)
    const int nside = (coh || P.n_blocks > DIRECT_BLOCKS) ? 0 : (B.nside < NWORK - 1 ? B.nside : NWORK - 1);
    const int n_orders = 1 + nside;
    CWork& wk = wks[0];
    const long long tk0 = wall_clock64();
    // first round of loads, all independent: the heads of the nodes, the work orders, the search state, the per-length counters
    // (only the nodes in use travel whole between memory and LDS: a node is 1.7 KB at 256-column windows; a linear search holds one or two)
    for (int x = tid; x < NQ * NODE_HEAD_WORDS; x += blockDim.x)
        ((uint32_t*)&nh[x / NODE_HEAD_WORDS])[x % NODE_HEAD_WORDS] = ((const uint32_t*)&P.nodes[x / NODE_HEAD_WORDS])[x % NODE_HEAD_WORDS];
    for (int x = tid; x < n_orders * (int)(sizeof(CWork) / 4); x += blockDim.x) ((uint32_t*)wks)[x] = ((const uint32_t*)P.work)[x];
    for (int x = tid; x < (int)(sizeof(CSearch) / 4); x += blockDim.x) ((uint32_t*)&ss)[x] = ((const uint32_t*)P.srch)[x];
    const int proc_words = (P.cap + 2 + 3) / 4;
    for (int x = tid; x < proc_words; x += blockDim.x) ((uint32_t*)proc)[x] = ((const uint32_t*)P.processed)[x];
    for (int x = tid; x < QE; x += blockDim.x) acc[x] = 0;
    if (tid < NWORK) { copy_from[tid] = -1; copy_len[tid] = 0; need_la[tid] = -1; replay_of[tid] = -1; kcopy[tid].nk = 0; if (tid >= n_orders) wks[tid].mode = M_NONE; }
    __syncthreads();
    if (wk.done) return;
    // second round: the tapes of the nodes that have one and the vote words of the step
    // the tapes of the nodes that have one: dc[0 .. n] and spec[i][0 .. n) (one flat loop: every load is in flight at once)
    auto tape_word_used = [](const CNode& x, int w) {                       // word w of the tape area
        if (!x.used) return false;
        if (w == 0) return true;                                            // dc[0] (= 0) is read through cost_at(0) whether or not the node has a tape
        if (x.n <= 0) return false;
        if (w < CW + 1) return w <= x.n;
        const int sw = w - (CW + 1);                                        // spec words: [i][CW / 4]
        return sw < 2 * (CW / 4) && (sw % (CW / 4)) * 4 < x.n;
    };
    for (int x = tid; x < NQ * NODE_TAPE_WORDS; x += blockDim.x) {
        const int k = x / NODE_TAPE_WORDS, w = x % NODE_TAPE_WORDS;
        if (tape_word_used(nh[k], w)) ((uint32_t*)&nh[k])[NODE_HEAD_WORDS + w] = ((const uint32_t*)&P.nodes[k])[NODE_HEAD_WORDS + w];
    }
    // the words of one order of the step that are in use: the cluster sums (a few per problem) or, for a problem of at most DIRECT_BLOCKS workgroups, the block its
    // workgroups added their words to (no reduce launch at all for a batch of such problems)
    auto load_votes = [&](const int w) {
        const int mode_in = wks[w].mode, n_in = (mode_in == M_WINDOW || mode_in == M_EXPAND) ? wks[w].n : 0;
        if (mode_in == M_NONE) return;
        const UsedWords uw(mode_in, n_in, wks[w].n_kids);
        if (P.n_blocks <= DIRECT_BLOCKS) {
            // the order's one block of words (the workgroups added theirs to it): read and cleared for the next step.  The exact and the lookahead
            // votes are four 16-bit fields per 64-bit word
            const int u = uw.used, ev = 2 * u, el = uw.has_la ? 2 * CW : 0, E = ev + ev + el + u + u;
            const size_t blk = (size_t)P.acc_block + (size_t)w;
            for (int c = tid; c < E; c += blockDim.x) {
                if (c < ev || (c >= 2 * ev && c < 2 * ev + el)) {
                    const bool exact = c < ev;
                    const int e = exact ? (c / u) * (CW + 1) + c % u : c - 2 * ev;
                    unsigned long long* src = exact ? B.PV + blk * 2 * (CW + 1) + e : B.PL + blk * 2 * CW + e;
                    const unsigned long long sum = coh ? __hip_atomic_exchange(src, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
                    if (sum) {
                        if (!coh) *src = 0ull;
                        uint32_t* dst = exact ? acc + e * 5 : acc + QSV + e * 4;
#pragma unroll
                        for (int f = 0; f < 4; ++f) dst[f] = (uint32_t)((sum >> (16 * f)) & 0xFFFFull);
                    }
                } else {
                    uint32_t* src; uint32_t* dst;
                    if (c < 2 * ev) { const int cc = c - ev, e = (cc / u) * (CW + 1) + cc % u; src = B.PE + blk * 2 * (CW + 1) + e; dst = acc + e * 5 + 4; }
                    else if (c < 2 * ev + el + u) { const int e = c - 2 * ev - el; src = B.PC + blk * (CW + 1) + e; dst = acc + QSV + QSL + e; }
                    else { const int e = c - 2 * ev - el - u; src = B.PR + blk * (CW + 1) + e; dst = acc + QSV + QSL + (CW + 1) + e; }
                    const uint32_t sum = coh ? __hip_atomic_exchange(src, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
                    if (sum) { if (!coh) *src = 0u; *dst = sum; }
                }
            }
        } else {
            const int parts = P.n_clusters;
            for (int c = tid; c < uw.total; c += blockDim.x) {
                const int o = uw.at(c);
                uint32_t sum = 0;
                for (int c0 = 0; c0 < parts; c0 += 8) {
                    uint32_t v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = c0 + q < parts ? B.Q[(size_t)(P.first_cluster + c0 + q) * QE + o] : 0u;
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum += v[q];
                }
                acc[o] = sum;
            }
        }
    };
    // ---- what the search wave works with (the lambdas below run on the workgroup's first wave only: every lane reads the same LDS words, so control flow is uniform;
    //      lane 0 writes, and a wave-level fence separates its writes from the reads that follow)
    const int lane = tid & (SP_WAVE - 1);
    auto cands = [&](const uint32_t* w5, int col, int out[4]) { return col_candidates(w5, col, P.cap, P.et, P.min_count, P.min_af, out); };
    // a node leaves the table -- and with it the children an expansion made ahead of its turn
    auto node_free = [&](int k) {
        const int np = nh[k].pex_n;
        for (int j = 0; j < np; ++j) { CNode& c = nh[nh[k].pex_kid[j]]; c.used = 0; c.complete = 0; }
        nh[k].pex_n = 0; nh[k].used = 0; nh[k].complete = 0;
    };
    uint8_t* Cb = P.C;
    int la_fresh = -1;                                    // the node whose lookahead votes are the sums of the search's own order of this very step (still in LDS)
    // More than max_queue_size nodes wait: the length threshold rises until at most that many stand at or above it -- the SHORTEST nodes go and the search is
    // pushed forwards (CdwfaConfig::max_queue_size; oracle/consensus.c).  Lane k looks at node k; nodes under the threshold are freed at once (they would be
    // dropped at their pop: their slots are needed)
    auto trim_queue = [&]() {
        for (;;) {
            const bool waits = lane < NQ && nh[lane].used == 1 && !nh[lane].complete;
            const int wlen = waits ? nh[lane].T + nh[lane].q : 0;
            const bool live = waits && wlen >= ss.threshold;
            if (waits && !live) node_free(lane);
            if (__builtin_popcountll(__ballot(live)) <= ss.max_queue) break;
            int shortest = live ? wlen : 0x7FFFFFFF;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(shortest, o); shortest = other < shortest ? other : shortest; }
            if (lane == 0) ss.threshold = shortest + 1;
            spw::wave_lds_sync();
        }
    };
    // The children of an expansion join the search: their ids, the length of their first window by how crowded the search is -- in a search that branches a little (two
    // haplotypes: a handful of nodes) a child's first window may be long; where the queue is full most children are dropped within a few columns --, the parent leaves.
    // kid(k): the node slot of child k.  (The children's other fields were filled when the expansion's launch came back: kids_from_sums.)
    auto adopt = [&](const int parent, const int n_kids, const bool kids_wait, auto kid) {      // kids_wait: the children's slots are marked as the search's already
        const int others = __builtin_popcountll(__ballot(lane < NQ && nh[lane].used == 1 && !nh[lane].complete)) - 1 - (kids_wait ? n_kids : 0);
        const int kid_w0 = others <= KID_CALM ? KID_LA : KID_W0;
        if (lane < n_kids) {
            CNode& c = nh[kid(lane)];
            c.used = 1; c.id = ss.next_id + lane;
            c.wcap = n_kids <= KID_LA_KIDS ? kid_w0 : WRAMP0;
        }
        spw::wave_lds_sync();
        if (lane == 0) {
            ss.next_id += n_kids;
            nh[parent].pex_n = 0;                         // (its children are the search's now)
            node_free(parent);
            ss.expansions += 1;
        }
        spw::wave_lds_sync();
        trim_queue();
    };
    struct Plan { int kind, nk, nc0, nc1, cut, L; uint32_t cd0, cd1; bool ends; };   // kind 0: no candidates at all (complete), 1: goes on alone, 2: branches
    auto cands_p = [&](const uint32_t* w5, int col, uint32_t& packed) { return col_candidates_packed(w5, col, P.cap, P.et, P.min_count, P.min_af, packed); };
    auto plan_of = [&](const CNode& x) -> Plan {
        Plan p;
        const bool has = x.q > 0 || x.n > 0, stood = !has || (x.a == x.n && x.have_out);
        p.cut = stood ? -1 : x.a;                                             // >= 0: the node stands behind that many verified bases of a window that was cut
        p.L = (has && stood ? x.T + x.n : x.T) + (p.cut > 0 ? p.cut : 0);     // the column of the decision (a window that stood: the other slot holds the state there)
        p.nc0 = p.nc1 = 0; p.cd0 = p.cd1 = 0; p.ends = false;
        if (!x.stopped[0]) { p.nc0 = cands_p(x.ev[0], p.L, p.cd0); p.ends = p.ends || p.nc0 == 0; }
        if (x.dual && !x.stopped[1]) { p.nc1 = cands_p(x.ev[1], p.L, p.cd1); p.ends = p.ends || p.nc1 == 0; }
        if (!x.dual) p.nk = p.nc0 + (P.allow_dual ? p.nc0 * (p.nc0 - 1) / 2 : 0);
        else p.nk = (p.nc0 || p.nc1) ? (p.nc0 ? p.nc0 : 1) * (p.nc1 ? p.nc1 : 1) : 0;
        p.kind = p.nk == 0 ? 0 : p.nk == 1 ? 1 : 2;
        return p;
    };
    // child k of a decision in the oracle's order: a single node's candidates, then (a second consensus allowed) its pairs; a dual node's combinations
    auto kid_of = [&](const Plan& pl, const int dual, const int k, int& b0, int& b1, int& sp) {
        auto code = [](uint32_t packed, int j) { return (int)((packed >> (2 * j)) & 3u); };
        sp = 0;
        if (!dual) {
            if (k < pl.nc0) { b0 = code(pl.cd0, k); b1 = -1; }
            else { int pa = 0, pb = 1; for (int j = pl.nc0; j < k; ++j) if (++pb >= pl.nc0) { ++pa; pb = pa + 1; } b0 = code(pl.cd0, pa); b1 = code(pl.cd0, pb); sp = 1; }
        } else { const int n1 = pl.nc1 ? pl.nc1 : 1, a2 = k / n1, b2 = k - a2 * n1; b0 = pl.nc0 ? code(pl.cd0, a2) : -1; b1 = pl.nc1 ? code(pl.cd1, b2) : -1; }
    };
    // the sums of an expansion's launch into its children's node slots (one lane per child); `mark`: 1 = the children join the search at once (adopt follows), 2 = made ahead.
    // L: the column that branched; par_cost: the parent's cost there; my_j: which of the order's children this lane's child is (-1: none) -- the order of an expansion lists
    // them in the search's own order (my_j = lane), a window that branches at its end lists the children it FORESAW and the exact votes say which of them there are, in
    // which order; s0: where the children's words start among the launch's sums
    auto kids_from_sums = [&](const CWork& w, const int mark, const int L, const long long par_cost, const int my_j, const int s0) {
        if (my_j >= 0) {
            const CNode& par = nh[w.node];
            const int k = my_j;
            CNode& c = nh[w.kid_node[k]];
            c.used = mark; c.complete = 0; c.id = 0; c.T = L + 1; c.cur = 0; c.pex_n = 0; c.pex_L = 0;
            c.dual = par.dual || w.kid_split[k]; c.split_at = w.kid_split[k] ? L : par.split_at;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool was_going = (i == 0 || par.dual) && !par.stopped[i];
                int st = par.stopped[i], ln = par.len[i];
                if (w.kid_base[k][i] >= 0) { st = 0; ln = L + 1; }
                else if (was_going) { st = 1; ln = L; }
                c.stopped[i] = st; c.len[i] = ln;
            }
            if (!c.dual) c.stopped[1] = 1;
            c.n = c.a = c.q = 0; c.have_out = 0; c.la_valid = w.n_kids <= KID_LA_KIDS ? 1 : 0; c.wcap = WRAMP0;
            c.cost0 = par_cost + (long long)sc[s0 + k]; c.dc[0] = 0; c.rest = sr[s0 + k]; c.rest_out = 0;
            for (int i = 0; i < 2; ++i) for (int bq = 0; bq < 5; ++bq) c.ev[i][bq] = sv[i][s0 + k][bq];
        }
        if (w.n_kids <= KID_LA_KIDS) {
            // the children's lookahead votes ([child][consensus][KID_LA columns] in the launch's lookahead words) into the children's rows; the columns behind them
            // are empty: a child's first window ends there at the latest.  (The rows are read back below, if a child is the next to go: a fence and the wave's
            // own order make the stores visible to its loads.)
            const uint32_t* flat = &sl[0][0][0];
            for (int y = lane; y < w.n_kids * 2 * KID_LA * 4; y += SP_WAVE) {                // (column KID_LA - 1 of a child has no votes: its window ends there, what lies behind is never read)
                const int k = y / (2 * KID_LA * 4), i = (y / (KID_LA * 4)) & 1, xb = y % (KID_LA * 4);
                P.la[(size_t)w.kid_node[k] * 2 * CW * 4 + (size_t)i * CW * 4 + xb] = flat[y];
            }
            __threadfence_block();
        }
        spw::wave_lds_sync();
    };
    // ---------------------------------------------------------------- 1. the result of the last step, order by order (`acc` holds the order's sums)
    auto take_result = [&](const int w) {
        const CWork& wo = wks[w];
        const int mode_in = wo.mode, n_in = (mode_in == M_WINDOW || mode_in == M_EXPAND) ? wo.n : 0;
        const bool side = w > 0;
        if (mode_in == M_INIT) {
            la_fresh = wo.node;
            CNode& x = nh[wo.node];
            if (lane == 0) {
                x.used = 1; x.id = ss.next_id++; x.complete = 0; x.T = 0; x.cur = wo.in_slot; x.dual = 0; x.split_at = -1;
                x.stopped[0] = 0; x.stopped[1] = 1; x.len[0] = x.len[1] = 0; x.n = x.a = x.q = 0; x.have_out = 0; x.la_valid = 1; x.wcap = CW;
                x.cost0 = 0; x.dc[0] = 0; x.rest = sr[0]; x.rest_out = 0; x.pex_n = 0; x.pex_L = 0;
            }
            if (lane < 10) x.ev[lane / 5][lane % 5] = sv[lane / 5][0][lane % 5];
            for (int y = lane; y < 2 * CW * 4; y += SP_WAVE) P.la[(size_t)wo.node * 2 * CW * 4 + y] = (&sl[0][0][0])[y];
        } else if (mode_in == M_WINDOW) {
            CNode& x = nh[wo.node];
            const int n = n_in, T = wo.T;
            int a = n;
            if (!wo.replay && n > 1) {
                for (int base = 0; base < n; base += SP_WAVE) {                  // 64 columns of the window at a time, one per lane
                    const int col = base + lane;
                    bool ok = true;
                    if (col >= 1 && col < n) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (i == 1 && !x.dual) continue;
                            if (!wo.go[i]) continue;
                            int c4[4];
                            const int nc = cands(sv[i][col], T + col, c4);
                            if (nc != 1 || c4[0] != wo.spec[i][col]) ok = false;   // a stop, a second candidate or another base: the tape ends there
                        }
                    }
                    const unsigned long long bad = __ballot(!ok);
                    if (bad) { a = base + __builtin_ctzll(bad); break; }
                }
            }
            for (int col = lane; col < a; col += SP_WAVE) {
#pragma unroll
                for (int i = 0; i < 2; ++i) if ((i == 0 || x.dual) && wo.go[i]) put(&Cb[((size_t)wo.node * 2 + i) * P.cs + T + col], wo.spec[i][col], coh);
            }
            // a replay re-reads columns whose costs are on the tape already: it only brings the state (and the votes) of column T + a
            if (!wo.replay) {
                long long carry = 0;
                for (int base = 0; base < a; base += SP_WAVE) {
                    const int col = base + lane;
                    long long c = col < a ? (long long)sc[col + 1] : 0;                   // push col + 1; inclusive prefix sum over the lanes
#pragma unroll
                    for (int o = 1; o < SP_WAVE; o <<= 1) { const long long up = __shfl_up(c, o); if (lane >= o) c += up; }
                    if (col < a) x.dc[col + 1] = (int32_t)(carry + c);
                    carry += __shfl(c, SP_WAVE - 1);
                }
                for (int col = lane; col < n; col += SP_WAVE) { x.spec[0][col] = wo.spec[0][col]; x.spec[1][col] = wo.spec[1][col]; }
            }
            if (lane < 10) x.ev[lane / 5][lane % 5] = sv[lane / 5][a][lane % 5];
            if (lane == 0) {
                ss.windows += 1; if (a < n) ss.cut_windows += 1;
                x.n = n; x.a = wo.replay ? n : a;
                x.have_out = (a == n) ? (wo.n_kids > 0 ? 2 : 1) : 0;        // 2: the state is there, lookahead votes for it are not (the launch's lookahead words were the children's)
                if (a == n) x.rest_out = sr[n];
            }
            if (wo.n_kids > 0) {
                // the window was ordered with the children of the branch the lookahead votes foresaw behind its last base.  They are taken when the window stood and the
                // exact votes of that column name no other children: then the node carries them as an expansion made ahead (pex) and adopts them at its turn without a launch
                spw::wave_lds_sync();
                bool take = a == n && B.k8_compound != 2;                        // (k8_compound 2, an experiment switch: the children are ordered and never taken)
                int my_j = -1, nk = 0;
                if (take) {
                    Plan pl;                                                        // what the node's decision at column T + n will be: its votes are the sums of this launch
                    pl.nc0 = pl.nc1 = 0; pl.cd0 = pl.cd1 = 0; bool ends = false;
                    if (!x.stopped[0]) { pl.nc0 = cands_p(sv[0][n], T + n, pl.cd0); ends = ends || pl.nc0 == 0; }
                    if (x.dual && !x.stopped[1]) { pl.nc1 = cands_p(sv[1][n], T + n, pl.cd1); ends = ends || pl.nc1 == 0; }
                    if (!x.dual) nk = pl.nc0 + (P.allow_dual ? pl.nc0 * (pl.nc0 - 1) / 2 : 0);
                    else nk = (pl.nc0 || pl.nc1) ? (pl.nc0 ? pl.nc0 : 1) * (pl.nc1 ? pl.nc1 : 1) : 0;
                    take = !ends && nk >= 2 && nk <= wo.n_kids;
                    if (take) {
                        bool found = true;
                        if (lane < nk) {                                            // lane k: child k in the search's own order, and which of the foreseen children it is
                            int b0, b1, sp;
                            kid_of(pl, x.dual, lane, b0, b1, sp);
                            for (int j = 0; j < wo.n_kids; ++j) if (wo.kid_base[j][0] == b0 && wo.kid_base[j][1] == b1 && wo.kid_split[j] == sp) my_j = j;
                            found = my_j >= 0;
                        }
                        take = __ballot(!found) == 0;
                    }
                }
                if (take) {
                    kids_from_sums(wo, 2, T + n, x.cost_at(n), lane < nk ? my_j : -1, n + 1);
                    // the node carries them in the search's order; foreseen children the exact votes do not name go back to the free slots
                    for (int k = 0; k < nk; ++k) { const int j = __builtin_amdgcn_readlane(my_j, k); if (lane == 0) x.pex_kid[k] = (int8_t)wo.kid_node[j]; }
                    if (lane < wo.n_kids) { bool mine = false; for (int k = 0; k < nk; ++k) mine = mine || __builtin_amdgcn_readlane(my_j, k) == lane; if (!mine) { nh[wo.kid_node[lane]].used = 0; nh[wo.kid_node[lane]].complete = 0; } }
                    if (lane == 0) {
                        x.pex_n = nk; x.pex_L = T + n; ss.compound_ok += 1;
                        // the children's consensus rows: the parent's bases in front of the window, the window's own (they reach the parent's row in this pass), their base
                        KidCopy& kc = kcopy[w];
                        kc.from = wo.node; kc.T = T; kc.n = n; kc.nk = nk; kc.dual = x.dual;
                    }
                    if (lane < nk) { KidCopy& kc = kcopy[w]; kc.node[lane] = (int8_t)wo.kid_node[my_j]; kc.b0[lane] = wo.kid_base[my_j][0]; kc.b1[lane] = wo.kid_base[my_j][1]; kc.sp[lane] = wo.kid_split[my_j]; }
                    for (int col = lane; col < n; col += SP_WAVE) { kcopy[w].spec[0][col] = wo.spec[0][col]; kcopy[w].spec[1][col] = wo.spec[1][col]; }
                } else if (lane < wo.n_kids) { nh[wo.kid_node[lane]].used = 0; nh[wo.kid_node[lane]].complete = 0; }
                spw::wave_lds_sync();
            } else if (a == n) {
                if (!side) la_fresh = wo.node;
                for (int y = lane; y < 2 * CW * 4; y += SP_WAVE) P.la[(size_t)wo.node * 2 * CW * 4 + y] = (&sl[0][0][0])[y];
                if (side) __threadfence_block();          // (a side order's rows may be read back from memory further down in this very pass)
            }
        } else if (mode_in == M_EXPAND) {
#ifdef SP_K8_TRACE
            if (lane == 0 && pi == 0) { printf("EXP%s L %d pre %d kids %d parcost %lld :", side ? " (side)" : "", wo.T + n_in, n_in, wo.n_kids, nh[wo.node].cost_at(nh[wo.node].q)); for (int k = 0; k < wo.n_kids; ++k) printf(" [%d/%d sc %u sr %u ev %u %u %u %u %u]", wo.kid_base[k][0], wo.kid_base[k][1], sc[k], sr[k], sv[0][k][0], sv[0][k][1], sv[0][k][2], sv[0][k][3], sv[0][k][4]); printf("\n"); }
#endif
            const long long par_cost = nh[wo.node].cost_at(nh[wo.node].q);
            if (!side) {
                kids_from_sums(wo, 1, wo.T + n_in, par_cost, lane < wo.n_kids ? lane : -1, 0);
                adopt(wo.node, wo.n_kids, true, [&](int k) { return (int)wo.kid_node[k]; });
            } else {
                // made ahead of the parent's turn: the children wait unseen (used = 2) until the search takes the parent out at that column
                kids_from_sums(wo, 2, wo.T + n_in, par_cost, lane < wo.n_kids ? lane : -1, 0);
                if (lane == 0) {
                    CNode& par = nh[wo.node];
                    par.pex_n = wo.n_kids; par.pex_L = wo.T + n_in;
                    for (int k = 0; k < wo.n_kids; ++k) par.pex_kid[k] = (int8_t)wo.kid_node[k];
                }
            }
        }
        spw::wave_lds_sync();
    };
    // the children a window brought with it get their consensus rows right behind the window's result, while the parent is still in the table (the search below may take it out,
    // adopt the children and hand its slot to another node's child in this very pass): the parent's row up to the window (whole 16-byte words from memory), then byte by byte
    // the rest of the last word, the window's own bases (kept in LDS: they reached the parent's row in this very pass) and the child's base
    auto copy_window_kids = [&](const int w) {
        const KidCopy& kc = kcopy[w];
        if (kc.nk <= 0) return;
        const int whole = kc.T >> 4, L = kc.T + kc.n;
        for (int k = 0; k < kc.nk; ++k) {
            const bool split = kc.sp[k] != 0;
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !(kc.dual || split)) continue;
                const int si = (i == 1 && split) ? 0 : i;
                const uint8_t* srow = P.C + ((size_t)kc.from * 2 + si) * P.cs;
                uint8_t* dst = P.C + ((size_t)kc.node[k] * 2 + i) * P.cs;
                for (int y = tid; y < whole; y += blockDim.x) reinterpret_cast<uint4*>(dst)[y] = reinterpret_cast<const uint4*>(srow)[y];
                for (int y = (whole << 4) + tid; y <= L; y += blockDim.x) {
                    int v;
                    if (y < kc.T) v = srow[y];
                    else if (y < L) v = kc.spec[kc.dual ? si : 0][y - kc.T];
                    else v = i ? kc.b1[k] : kc.b0[k];
                    if (v >= 0) dst[y] = (uint8_t)v;
                }
            }
        }
    };
    // side orders first, the search's own last: its sums stay in LDS for the search (la_fresh)
    for (int w = n_orders - 1; w >= 0; --w) {
        if (wks[w].mode == M_NONE) continue;
        load_votes(w);
        __syncthreads();
        if (tid < SP_WAVE) take_result(w);
        __syncthreads();
        copy_window_kids(w);
        if (w > 0) {                                       // clear the sums for the next order
            for (int x = tid; x < QE; x += blockDim.x) acc[x] = 0;
            __syncthreads();
        }
    }
    const long long tk1 = wall_clock64();
    long long tk2 = tk1, tk3 = tk1;
#ifdef SP_K8_SEARCH_TICKS
    long long ts_pick = 0, ts_block = 0, ts_iters = 0;
#endif
    if (tid < SP_WAVE) {
        // ---------------------------------------------------------------- 2. the search, played forward over the tapes
        tk2 = wall_clock64();
        if (lane == 0) { const int steps_made = wk.pad + 1; for (int w = 0; w < n_orders; ++w) { wks[w].mode = M_NONE; wks[w].n = 0; wks[w].replay = 0; wks[w].n_kids = 0; } wk.pad = steps_made; }
        spw::wave_lds_sync();
        // The decision of a node that stands at the end of its tape, written as work order `w` (lane 0 only).  w == 0: the search's own -- the node has just been taken out at
        // that column.  w > 0: a side order, made AHEAD of the node's turn for a node that waits -- nothing the search can observe changes: the node keeps its place, cost
        // and id; its window comes back as a tape it consumes when its turn comes, the children of its expansion wait unseen until then.
        // -> 0: no launch for this node (it completed, or -- side orders -- it is left for its turn), 1: the order was written
        // The decision of a node that stands at the end of its tape, in two parts: what it will be (plan_of: reads the node, changes nothing) and the work order that carries it
        // out (act).  The search's own order (w = 0) is planned and written by lane 0 for the node that has just been taken out at that column.  Side orders (w > 0) are made
        // AHEAD of a waiting node's turn -- every lane plans the node of its number, the chosen ones write their orders side by side -- and change nothing the search can observe:
        // the node keeps its place, cost and id; its window comes back as a tape it consumes when its turn comes, the children of its expansion wait unseen until then.
        // -> 0: no launch for this node (it completed), 1: the order was written.  free_bits: the free node slots the children of an expansion are taken from, lowest first
        auto act = [&](const int xi, const int w, const Plan& pl, unsigned long long& free_bits) -> int {
            CNode& x = nh[xi];
            CWork& W = wks[w];
            const bool side = w > 0;
            const int cut = pl.cut;
            // its state at that column has to be there
            if ((x.q > 0 || x.n > 0) && cut < 0) {                                 // the window stood: the other slot is the state
                const int xn = x.n;
                if (xn > 1) x.wcap = SP_K8_RAMP * x.wcap < CW ? SP_K8_RAMP * x.wcap : CW;
                const long long c_end = x.cost_at(xn);
                const int la_ok = x.have_out == 1 ? 1 : 0;
                x.T += xn; x.cur ^= 1; x.cost0 = c_end; x.dc[0] = 0; x.rest = x.rest_out; x.n = x.a = x.q = 0; x.have_out = 0; x.la_valid = la_ok;
            }
            // (which consensuses grew in that window: as they stand BEFORE this column's decision, which may end one)
            const int dual = x.dual, cur = x.cur, split_at = x.split_at, xT = x.T;
            const int pre_go0 = !x.stopped[0] ? 1 : 0, pre_go1 = (dual && !x.stopped[1]) ? 1 : 0;
            auto order_replay = [&]() {
                // push the verified bases again from the kept state (nothing is speculated)
                W.mode = M_WINDOW; W.node = xi; W.in_slot = cur; W.T = xT; W.n = x.a; W.replay = 1;
                W.dual = dual; W.split_at = split_at; W.go[0] = pre_go0; W.go[1] = pre_go1;
                replay_of[w] = xi;                                              // (the bases are copied below, a lane each)
                ss.inflight = xi;
            };
            // (a cut window: the votes of the column behind the verified bases are there -- x.ev -- and say whether the node branches; if it does, the expansion
            //  launch pushes the verified bases itself in front of the children's: the replay launch is only made where the node goes on alone)
#ifdef SP_K8_NO_PRE
            if (!side && cut >= 0 && (SP_K8_NO_PRE == 0 || (SP_K8_NO_PRE == 1 && dual) || (SP_K8_NO_PRE == 2 && !dual))) { order_replay(); return 1; }
#endif
            const int L = pl.L;
            if (!side) {                                                          // a consensus without candidates ends here
                if (!x.stopped[0] && pl.nc0 == 0) { x.stopped[0] = 1; x.len[0] = L; }
                if (dual && !x.stopped[1] && pl.nc1 == 0) { x.stopped[1] = 1; x.len[1] = L; }
            }
            if (pl.nk == 0) {
                if (cut >= 0) { order_replay(); return 1; }                       // (its final cost needs the state: the replay brings it)
                const long long fc = x.cost0 + (P.et ? 0 : x.rest);               // complete
                if (ss.best_node < 0 || fc < ss.best_final) { if (ss.best_node >= 0) node_free(ss.best_node); ss.best_node = xi; ss.best_final = fc; x.complete = 1; }
                else node_free(xi);
                return 0;
            }
            // child k in the oracle's order: a single node's candidates, then (a second consensus allowed) its pairs; a dual node's combinations
            const int nc0 = pl.nc0, nc1 = pl.nc1;
            auto code = [](uint32_t packed, int j) { return (int)((packed >> (2 * j)) & 3u); };
            if (pl.nk == 1 && cut >= 0) { order_replay(); return 1; }
            if (pl.nk == 1) {
                // one child: the node itself goes on, through a window whose first base is this decision
                const int b0 = nc0 ? code(pl.cd0, 0) : -1, b1 = (dual && nc1) ? code(pl.cd1, 0) : -1;
                W.mode = M_WINDOW; W.node = xi; W.in_slot = cur; W.T = L; W.replay = 0; W.dual = dual; W.split_at = split_at;
                W.go[0] = b0 >= 0; W.spec[0][0] = (uint8_t)(b0 >= 0 ? b0 : 0); W.go[1] = b1 >= 0; W.spec[1][0] = (uint8_t)(b1 >= 0 ? b1 : 0);
                need_la[w] = x.la_valid ? xi : -1;
                W.n = 1;
                x.n = 0; x.a = 0; x.q = 0;
                if (!side) ss.inflight = xi;                                     // the pop is under way: when the window is back the node consumes its first column
                return 1;
            }
            // several children: one push each into fresh nodes
            W.mode = M_EXPAND; W.node = xi; W.in_slot = cur; W.T = xT; W.n = cut > 0 ? cut : 0; W.dual = dual; W.split_at = split_at; W.replay = 0;
            W.go[0] = nc0 > 0; W.go[1] = nc1 > 0; W.pre_go[0] = pre_go0; W.pre_go[1] = pre_go1;
            if (cut > 0) replay_of[w] = xi;                                     // (the verified bases go into the work order below, a lane each)
            int made = 0;
            for (int k = 0; k < pl.nk && made < MAXKIDS; ++k) {
                if (!free_bits) break;                                          // (the table holds the queue plus one expansion: not reached)
                int b0, b1, sp;
                kid_of(pl, dual, k, b0, b1, sp);
                const int kn = __builtin_ctzll(free_bits); free_bits &= free_bits - 1;
                nh[kn].used = side ? 2 : 1; nh[kn].complete = 0; nh[kn].pex_n = 0;
                W.kid_node[made] = kn; W.kid_base[made][0] = (int8_t)b0; W.kid_base[made][1] = (int8_t)b1; W.kid_split[made] = (int8_t)sp;
                ++made;
            }
            W.n_kids = made;
            copy_from[w] = xi; copy_len[w] = L;
            return 1;
        };
        unsigned long long main_kids = 0;                                       // the node slots the search's own order gave to the children of its expansion
        for (int guard = 0; ; ++guard) {
            if (guard > 100000) { if (lane == 0) wk.done = 1; break; }
#ifdef SP_K8_SEARCH_TICKS
            const long long ts0 = wall_clock64(); ts_iters += 1;
#endif
            int xi = ss.inflight;
            if (xi < 0) {
                // the best and the second best of the nodes that wait: lane k looks at node k
                long long kc = 0x7FFFFFFFFFFFFFFFll; int kt = -1, kid = 0x7FFFFFFF, kx = -1;
                if (lane < NQ && nh[lane].used == 1 && !nh[lane].complete) { const CNode& p = nh[lane]; kc = p.cost_at(p.q); kt = p.T + p.q; kid = p.id; kx = lane; }
                auto less = [](long long c1, int t1, int i1, long long c2, int t2, int i2) { return c1 < c2 || (c1 == c2 && (t1 > t2 || (t1 == t2 && i1 < i2))); };
                long long bc = kc; int bt = kt, bid = kid, bx = kx;
                long long sc2 = 0x7FFFFFFFFFFFFFFFll; int st2 = -1, sid2 = 0x7FFFFFFF, sx2 = -1;
                const unsigned long long waiting = __ballot(kx >= 0);
                if (__builtin_popcountll(waiting) == 1) {
                    // one node waits (a linear search: most steps of most searches): it is the best, there is no second best -- no reductions
                    const int only = __builtin_ctzll(waiting);                      // (its lane holds what was just read of it)
                    bc = ((long long)__builtin_amdgcn_readlane((int)(kc >> 32), only) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)kc, only);
                    bt = __builtin_amdgcn_readlane(kt, only); bid = __builtin_amdgcn_readlane(kid, only); bx = only;
                    xi = only;
                } else {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const long long oc = __shfl_xor(bc, o); const int ot = __shfl_xor(bt, o), oi = __shfl_xor(bid, o), ox = __shfl_xor(bx, o);
                        if (ox >= 0 && (bx < 0 || less(oc, ot, oi, bc, bt, bid))) { bc = oc; bt = ot; bid = oi; bx = ox; }
                    }
                    xi = bx;
                    if (xi < 0) { if (lane == 0) wk.done = 1; break; }                          // nothing waits
                    sc2 = (kx == xi) ? 0x7FFFFFFFFFFFFFFFll : kc; st2 = kt; sid2 = kid; sx2 = (kx == xi) ? -1 : kx;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const long long oc = __shfl_xor(sc2, o); const int ot = __shfl_xor(st2, o), oi = __shfl_xor(sid2, o), ox = __shfl_xor(sx2, o);
                        if (ox >= 0 && (sx2 < 0 || less(oc, ot, oi, sc2, st2, sid2))) { sc2 = oc; st2 = ot; sid2 = oi; sx2 = ox; }
                    }
                }
                CNode& x = nh[xi];
                if (ss.best_node >= 0 && bc >= ss.best_final) { if (lane == 0) wk.done = 1; break; }   // nothing that waits can beat (or precede) the complete node
                // pops this node makes in a row: the first is decided; the following ones need the node to stay ahead of the second best,
                // its lengths to be open (threshold, capacity per length) and stop at a pop that moves the threshold.  A tape longer than the wave is
                // consumed 64 columns at a time without looking for the best node again: the others have not moved, and the comparison a new look
                // would make for the node's next pop is the one every further pop of a row has to pass anyway
                const int linear0 = x.a - x.q;
                bool again = false, freed = false;
                int Lc = bt;
                for (;;) {
                    const int L = Lc, q = x.q, a = x.a;
                    const int wo = ss.wo_constraint;
                    const int jc = wo - 1 - ss.pops_mod;                                  // the pop with this index (0-based) makes pops a multiple of wo
                    const int linear = a - q;                                            // pops that only consume the tape
                    const int want = linear > 0 ? (linear < SP_WAVE ? linear : SP_WAVE) : 1;
                    bool can = lane < want && L + lane >= ss.threshold && proc[L + lane] < ss.per_size && lane <= jc;
                    if (can && (lane > 0 || again)) {
                        const long long cj = x.cost_at(q + lane);
                        can = (sx2 < 0 || less(cj, L + lane, ss.next_id + lane - 1, sc2, st2, sid2)) && !(ss.best_node >= 0 && cj >= ss.best_final);
                    }
                    const unsigned long long no = ~__ballot(can);
                    const int m = no ? __builtin_ctzll(no) : SP_WAVE;
                    if (m == 0) { if (!again) { if (lane == 0) node_free(xi); freed = true; } break; }   // shorter than the threshold / its length is full (a further chunk: just look again)
                    if (lane < m) proc[L + lane] += 1;
                    if (lane == 0) {
                        ss.pops += m;
                        { int pm = ss.pops_mod + m; if (pm >= wo) pm %= wo; ss.pops_mod = pm; }
                        if (L + m - 1 > ss.farthest) ss.farthest = L + m - 1;
                        if (jc < m && ss.farthest > ss.threshold) ss.threshold = ss.farthest;
                        if (linear > 0) { x.q = q + m; x.id = ss.next_id + m - 1; ss.next_id += m; }
                    }
                    spw::wave_lds_sync();
                    if (!(linear > 0 && m == SP_WAVE && q + m < a)) break;
                    again = true; Lc = L + m;
                }
                if (freed) { spw::wave_lds_sync(); continue; }
                if (linear0 > 0) continue;                                               // verified columns of its tape: the node moved on (its one child each)
            } else if (lane == 0) ss.inflight = -1;
            spw::wave_lds_sync();
            // the node stands at the end of its tape and its pop is accounted for: the decision of that column
            int stop = 0;
#ifdef SP_K8_SEARCH_TICKS
            const long long ts1 = wall_clock64(); ts_pick += ts1 - ts0;
#endif
            {
                const CNode& x = nh[xi];
                if (x.q < x.a) {                                                         // (a window that was under way came back: its first column is this pop's child)
                    if (lane == 0) { nh[xi].q += 1; nh[xi].id = ss.next_id++; }
                    spw::wave_lds_sync();
                    continue;
                }
                if (x.pex_n > 0) {
                    // its expansion was made ahead of its turn (a side order): the children join the search here, where the launch would have been ordered
                    const int np = x.pex_n;
                    if (lane == 0) ss.adopted += 1;
                    adopt(xi, np, false, [&](int k) { return (int)nh[xi].pex_kid[k]; });
                    continue;
                }
            }
            const unsigned long long free_before = __ballot(lane < NQ && !nh[lane].used);
            unsigned long long free_nodes = free_before;
            if (lane == 0) { const Plan pl = plan_of(nh[xi]); stop = act(xi, 0, pl, free_nodes); }
            stop = __builtin_amdgcn_readfirstlane(stop);
            free_nodes = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(free_nodes >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)free_nodes);
            main_kids = free_before & ~free_nodes;
            spw::wave_lds_sync();
#ifdef SP_K8_SEARCH_TICKS
            ts_block += wall_clock64() - ts1;
#endif
            if (stop) break;
        }
        spw::wave_lds_sync();
        // ---------------------------------------------------------------- 3. side orders: what the nodes that wait beside the search's own order will need at their turn
        if (lane == 0 && !wk.done && wk.mode != M_NONE) ss.steps += 1;
        if (nside > 0 && !wk.done && wk.mode != M_NONE && __builtin_popcountll(__ballot(lane < NQ && nh[lane].used == 1 && !nh[lane].complete)) > 1) {
            // lane k plans node k: does it stand at the end of its tape, would it be taken out at all, and what would its decision be
            Plan pl; pl.kind = 0; pl.nk = 0;
            bool cand = false;
            long long kc = 0x7FFFFFFFFFFFFFFFll; int kt = -1, kid = 0x7FFFFFFF;
            if (lane < NQ && nh[lane].used == 1 && !nh[lane].complete && lane != wk.node && !((main_kids >> lane) & 1ull) && nh[lane].q >= nh[lane].a && nh[lane].pex_n == 0) {
                const CNode& x = nh[lane];
                const int Lq = x.T + x.q;
                if (Lq >= ss.threshold && Lq < P.cap + 2 && proc[Lq] < ss.per_size) {
                    pl = plan_of(x);
                    // a window needs the state at that column (a cut window is replayed at the node's turn); a column that ends a consensus is left for the node's turn as well
                    cand = !pl.ends && ((pl.kind == 1 && pl.cut < 0) || (pl.kind == 2 && pl.nk <= MAXKIDS));
                    if (cand) { kc = x.cost_at(x.q); kt = Lq; kid = x.id; }
                }
            }
            const unsigned long long cm = __ballot(cand);
            if (cm) {
                // the candidates in the search's own order: the rank of a lane = how many of the others go before it (they are few: a scalar pass over them, no reductions)
                auto less = [](long long c1, int t1, int i1, long long c2, int t2, int i2) { return c1 < c2 || (c1 == c2 && (t1 > t2 || (t1 == t2 && i1 < i2))); };
                int my_rank = 0;
                for (unsigned long long m = cm; m; m &= m - 1) {
                    const int b = __builtin_ctzll(m);
                    const long long oc = ((long long)__builtin_amdgcn_readlane((int)(kc >> 32), b) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)kc, b);
                    const int ot = __builtin_amdgcn_readlane(kt, b), oi = __builtin_amdgcn_readlane(kid, b);
                    if (b != lane && less(oc, ot, oi, kc, kt, kid)) ++my_rank;
                }
                const unsigned long long free_nodes = __ballot(lane < NQ && !nh[lane].used);
                const int hidden = __builtin_popcountll(__ballot(lane < NQ && nh[lane].used == 2)), n_free = __builtin_popcountll(free_nodes);
                // the node table keeps room for the queue, the complete node and the children of the search's own next expansion whatever is made ahead
                const int room = NQ - (ss.max_queue + 1) - MAXKIDS;
                const int n_cand = __builtin_popcountll(cm);
                int my_w = 0, my_skip = 0, accepted = 0, taken = 0, n_sw = 0, n_se = 0;
                for (int r = 0; r < n_cand && accepted < nside; ++r) {
                    const unsigned long long who = __ballot(cand && my_rank == r);
                    if (!who) continue;
                    const int b = __builtin_ctzll(who);
                    const int bkind = __builtin_amdgcn_readlane(pl.kind, b), bnk = __builtin_amdgcn_readlane(pl.nk, b);
                    if (bkind == 2 && (hidden + taken + bnk > room || n_free - taken < bnk + MAXKIDS)) continue;
                    if (lane == b) { my_w = 1 + accepted; my_skip = taken; }
                    ++accepted;
                    if (bkind == 2) { taken += bnk; ++n_se; } else ++n_sw;
                }
                if (lane == 0) { ss.side_windows += n_sw; ss.side_expansions += n_se; }
                if (my_w > 0) {                                                      // the chosen lanes write their orders side by side
                    unsigned long long mine = free_nodes;
                    for (int j = 0; j < my_skip; ++j) mine &= mine - 1;              // (the slots the orders in front of this one take)
                    (void)act(lane, my_w, pl, mine);
                }
            }
        }
        spw::wave_lds_sync();
        for (int w = 0; w < n_orders; ++w) {
            const int ro = replay_of[w];
            if (ro >= 0) {
                const CNode& x = nh[ro];
                for (int jq = lane; jq < x.a; jq += SP_WAVE) { wks[w].spec[0][jq] = x.spec[0][jq]; wks[w].spec[1][jq] = x.spec[1][jq]; }
            }
        }
#ifdef SP_K8_PF_PROBE
        if (wk.mode == M_WINDOW || wk.mode == M_EXPAND) {
            bool idle = lane < NQ && nh[lane].used == 1 && !nh[lane].complete && lane != wk.node && nh[lane].q >= nh[lane].a;
            if (wk.mode == M_EXPAND) for (int k = 0; k < wk.n_kids; ++k) if (wk.kid_node[k] == lane) idle = false;
            const unsigned long long mask = __ballot(idle);
            if (lane == 0) {
                if ((ss.idle_mask >> wk.node) & 1ull) { if (wk.mode == M_EXPAND) ss.pf_exp += 1; else if (wk.replay) ss.pf_replay += 1; else ss.pf_win += 1; }
                ss.idle_mask = mask;
            }
        }
#endif
        // the speculated part of a new window: lane j takes the heaviest lookahead vote for push j of every consensus that grows; the
        // window ends where a consensus has no lookahead votes left (or at cap)
        spw::wave_lds_sync();
        tk3 = wall_clock64();
        for (int w = 0; w < n_orders; ++w) {
            const int nl = need_la[w];
            if (nl < 0) continue;
            CWork& W = wks[w];
            const uint32_t* la = (w == 0 && nl == la_fresh) ? &sl[0][0][0] : P.la + (size_t)nl * 2 * CW * 4;
            const int lim = nh[nl].wcap;
            // A WINDOW THAT BRANCHES AT ITS END.  The lookahead votes also say where the node will probably branch: the first column at which a second base has a fair share
            // of them.  The window is then ordered up to that column WITH the children of the foreseen expansion (the step kernel makes them from the state the window leaves
            // there, their words behind the window's): if the window stands and the exact votes of that column name no other children, the node adopts them at its turn
            // without a launch of its own -- the expansion launch that follows nearly every cut window of a branching search.  Needs free node slots like a side expansion.
            const int hidden = __builtin_popcountll(__ballot(lane < NQ && nh[lane].used == 2)), n_free = __builtin_popcountll(__ballot(lane < NQ && !nh[lane].used));
            const int n_waiting = __builtin_popcountll(__ballot(lane < NQ && nh[lane].used == 1 && !nh[lane].complete));
            // (only in a search that is branching -- other nodes wait beside this one, or its windows have not grown to full length since it was born in an expansion --: in a
            //  linear search a branch foreseen in vain cuts a 256-column window short, 18 of 92 windows of the HLA sample for 2 children taken)
            const bool may_branch = B.k8_compound && !coh && P.n_blocks <= DIRECT_BLOCKS && (lim < CW || n_waiting > 1 || B.k8_compound == 3) && hidden + KID_LA_KIDS <= NQ - (ss.max_queue + 1) - MAXKIDS && n_free >= KID_LA_KIDS + MAXKIDS;
            int nn = lim, bcol = -1, bn0 = 0, bn1 = 0; uint32_t bc0 = 0, bc1 = 0;
            for (int base = 0; base < lim; base += SP_WAVE) {
                const int col = base + lane;
                bool have = col >= 1 && col < lim && W.T + col < P.cap;
                int pick[2] = { 0, 0 };
                int n2[2] = { 0, 0 }; uint32_t pc[2] = { 0, 0 };                // the bases with a fair share of the column's lookahead votes, heaviest first
                if (have) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (!W.go[i]) continue;
                        const uint4 wv = *reinterpret_cast<const uint4*>(la + ((size_t)i * CW + (col - 1)) * 4);
                        uint32_t wa = wv.x, wb = wv.y, wc = wv.z, wd = wv.w, ca = 0, cb = 1, cc = 2, cd = 3;
#define SP_CSW(x, y, cx, cy) if (y > x) { const uint32_t tw_ = x; x = y; y = tw_; const uint32_t tc_ = cx; cx = cy; cy = tc_; }
                        SP_CSW(wa, wb, ca, cb) SP_CSW(wa, wc, ca, cc) SP_CSW(wa, wd, ca, cd) SP_CSW(wb, wc, cb, cc) SP_CSW(wb, wd, cb, cd) SP_CSW(wc, wd, cc, cd)
#undef SP_CSW
                        if (wa == 0) have = false;
                        pick[i] = (int)ca;
                        // (a foreseen branch that does not come costs a window cut short; one that comes saves a launch: only the clear cases -- a second base with SHARE
                        //  per cent of the column's lookahead votes, k8_compound >= 10 names another share)
                        const double share = B.k8_compound >= 10 ? 0.01 * (double)B.k8_compound : 0.25;
                        const double fair = (share > P.min_af ? share : P.min_af) * (double)(wa + wb + wc + wd);
                        int m = 1; uint32_t pk = ca;
                        if (wb >= 2 && (double)wb >= fair) { pk |= cb << (2 * m); ++m; }
                        if (wc >= 2 && (double)wc >= fair) { pk |= cc << (2 * m); ++m; }
                        if (wd >= 2 && (double)wd >= fair) { pk |= cd << (2 * m); ++m; }
                        n2[i] = m; pc[i] = pk;
                    }
                }
                unsigned long long miss = __ballot(!have);
                if (base == 0) miss &= ~1ull;                                  // column 0 is the exact base
                int first = miss ? __builtin_ctzll(miss) : SP_WAVE;
                const unsigned long long brm = may_branch ? __ballot(have && col >= 1 && col <= CW - KID_LA_KIDS - 1 && (n2[0] > 1 || n2[1] > 1)) : 0ull;
                const int fb = brm ? __builtin_ctzll(brm) : SP_WAVE;
                if (fb < first) {
                    // how many children would that be?  (the node's candidates there, as plan_of counts them)
                    const int c0 = W.go[0] ? __builtin_amdgcn_readlane(n2[0], fb) : 0, c1 = (W.dual && W.go[1]) ? __builtin_amdgcn_readlane(n2[1], fb) : 0;
                    const int kids = !W.dual ? c0 + (P.allow_dual ? c0 * (c0 - 1) / 2 : 0) : (c0 ? c0 : 1) * (c1 ? c1 : 1);
                    if (kids >= 2 && kids <= KID_LA_KIDS) {
                        bcol = base + fb; bn0 = c0; bn1 = c1;
                        bc0 = (uint32_t)__builtin_amdgcn_readlane((int)pc[0], fb); bc1 = (uint32_t)__builtin_amdgcn_readlane((int)pc[1], fb);
                        first = fb;
                    }
                }
                if (col >= 1 && lane < first) { W.spec[0][col] = (uint8_t)pick[0]; W.spec[1][col] = (uint8_t)pick[1]; }
                if (miss || bcol >= 0) { nn = base + first; break; }
            }
            if (lane == 0) {
                W.n = nn < lim ? nn : lim;
            }
            if (bcol >= 1) {
                const unsigned long long free_mask = __ballot(lane < NQ && !nh[lane].used);
                if (lane == 0 && W.n == bcol) {
                    Plan fp; fp.nc0 = bn0; fp.nc1 = bn1; fp.cd0 = bc0; fp.cd1 = bc1;
                    const int kids = !W.dual ? bn0 + (P.allow_dual ? bn0 * (bn0 - 1) / 2 : 0) : (bn0 ? bn0 : 1) * (bn1 ? bn1 : 1);
                    unsigned long long fm = free_mask;
                    int made = 0;
                    for (int k = 0; k < kids && fm; ++k) {
                        int b0, b1, sp;
                        kid_of(fp, W.dual, k, b0, b1, sp);
                        const int kn = __builtin_ctzll(fm); fm &= fm - 1;
                        nh[kn].used = 2; nh[kn].complete = 0; nh[kn].pex_n = 0;
                        W.kid_node[made] = kn; W.kid_base[made][0] = (int8_t)b0; W.kid_base[made][1] = (int8_t)b1; W.kid_split[made] = (int8_t)sp;
                        ++made;
                    }
                    W.n_kids = made;
                    ss.compound += 1;
                }
                spw::wave_lds_sync();
            }
        }
    }
    __syncthreads();
    // children get their parent's consensus (and their own base behind it)
    for (int w = 0; w < n_orders; ++w) {
        if (copy_from[w] < 0) continue;
        const CWork& W = wks[w];
        const int cfrom = copy_from[w], clen = copy_len[w];
        for (int k = 0; k < W.n_kids; ++k) {
            const int kn = W.kid_node[k];
            const bool split = W.kid_split[k] != 0;
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !(W.dual || split)) continue;
                const uint4* src = reinterpret_cast<const uint4*>(P.C + ((size_t)cfrom * 2 + ((i == 1 && split) ? 0 : i)) * P.cs);
                uint8_t* dst = P.C + ((size_t)kn * 2 + i) * P.cs;
                const int whole = clen >> 4, tail = clen & 15;                  // (the rows are 16-byte aligned)
                for (int y = tid; y < whole + (tail ? 1 : 0); y += blockDim.x) {
                    uint4 v;
                    if (coh) {                                                          // (the parent's bytes were stored write-through; 8 bytes per access both ways)
                        const unsigned long long* s8 = reinterpret_cast<const unsigned long long*>(src + y);
                        const unsigned long long lo = coh_load(s8), hi = coh_load(s8 + 1);
                        v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32); v.z = (uint32_t)hi; v.w = (uint32_t)(hi >> 32);
                    } else v = src[y];
                    if (y == whole && W.kid_base[k][i] >= 0) reinterpret_cast<uint8_t*>(&v)[tail] = (uint8_t)W.kid_base[k][i];   // the child's own base lies in the last word
                    if (coh) {
                        unsigned long long* d8 = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint4*>(dst) + y);
                        coh_store(d8, (unsigned long long)v.x | ((unsigned long long)v.y << 32)); coh_store(d8 + 1, (unsigned long long)v.z | ((unsigned long long)v.w << 32));
                    } else reinterpret_cast<uint4*>(dst)[y] = v;
                }
                if (tid == 0 && tail == 0 && W.kid_base[k][i] >= 0) put(&dst[clen], (uint8_t)W.kid_base[k][i], coh);
            }
        }
    }
    if (tid == 0) {
        const long long tk4 = wall_clock64(); ss.ticks[0] += tk1 - tk0; ss.ticks[1] += tk2 - tk1; ss.ticks[2] += tk3 - tk2; ss.ticks[3] += tk4 - tk3;
        if (B.step_t) {
            const long long st0 = (long long)coh_load(&B.step_t[2 * pi]), st1 = (long long)coh_load(&B.step_t[2 * pi + 1]);
            if (st1 > 0 && st0 <= st1) {
                ss.step_ticks += st1 - st0;
                if (ss.last_end > 0 && st0 >= ss.last_end) ss.gap_ticks += st0 - ss.last_end;
                if (tk0 >= st1) { ss.gap_ticks += tk0 - st1; ss.gap_ctl_ticks += tk0 - st1; }
            }
            put(&B.step_t[2 * pi], ~0ull, coh); put(&B.step_t[2 * pi + 1], 0ull, coh);
            ss.last_end = tk4;
        }
    }
#ifdef SP_K8_SEARCH_TICKS
    // (variant: the search split into picking the node + tape consumption / the decision block / loop iterations x 100, in place of result, tail and load)
    if (tid == 0) { ss.ticks[0] += 100 * ts_iters - (tk1 - tk0); ss.ticks[1] += ts_pick - (tk2 - tk1); ss.ticks[3] += ts_block - (wall_clock64() - tk3); }
#endif
    __syncthreads();
    for (int x = tid; x < NQ * NODE_HEAD_WORDS; x += blockDim.x)          // (the step workgroups read the nodes' lengths: write-through)
        put(&((uint32_t*)&P.nodes[x / NODE_HEAD_WORDS])[x % NODE_HEAD_WORDS], ((const uint32_t*)&nh[x / NODE_HEAD_WORDS])[x % NODE_HEAD_WORDS], coh);
    for (int x = tid; x < NQ * NODE_TAPE_WORDS; x += blockDim.x) {
        const int k = x / NODE_TAPE_WORDS, w = x % NODE_TAPE_WORDS;
        if (tape_word_used(nh[k], w)) ((uint32_t*)&P.nodes[k])[NODE_HEAD_WORDS + w] = ((const uint32_t*)&nh[k])[NODE_HEAD_WORDS + w];
    }
    for (int x = tid; x < n_orders * (int)(sizeof(CWork) / 4); x += blockDim.x) put(&((uint32_t*)P.work)[x], ((const uint32_t*)wks)[x], coh);
    for (int x = tid; x < (int)(sizeof(CSearch) / 4); x += blockDim.x) ((uint32_t*)P.srch)[x] = ((const uint32_t*)&ss)[x];
    for (int x = tid; x < proc_words; x += blockDim.x) ((uint32_t*)P.processed)[x] = ((const uint32_t*)proc)[x];
    if (tid == 0 && B.prog) {
        __hip_atomic_store(B.prog + 2 * pi, (uint32_t)wk.pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (wk.done) __hip_atomic_store(B.prog + 2 * pi + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int MAXP>
__global__ void __launch_bounds__(1024) cons_control_kernel(ConsBatchT<MAXP> B) { cons_control_body<MAXP>(B); }

// ------------------------------------------------------------------------------------------------------------------------------
// Persistent mode (batches whose problems all have <= PERSIST_BLOCKS workgroups and whose workgroups fit the device together): the same two bodies, each in a
// loop of its own kernel, started ONCE per batch on two streams.  A step workgroup runs step k when the problem's control workgroup has answered step k - 1,
// then reports in; the control workgroup runs control step k when all workgroups of its problem have reported step k.  No kernel boundary, no dispatch, and no
// lockstep between the problems of the batch: each search runs at the pace of its own chain.  (A launch pair per step cost a CYP2D6 sample 34 of its 105 us per
// step between the kernels while K1's grids ran beside it: bench.py critical_path.)
// THE HAND-OVER (gfx950; measured, DESIGN.md section 9 -- not a release / acquire pair of the memory model, whose agent-scope fences cost an L2 write-back and an
// invalidate per step, 14.8 us):  control -> step: everything the step workgroups read next (work order, node lengths, consensus bytes) is stored write-through (sc1,
// relaxed agent-scope atomic stores), every thread drains its stores (s_waitcnt vmcnt(0)), a workgroup barrier, then ONE relaxed store raises the problem's word; the
// step workgroups poll the word and read those data with sc1 loads (past their L1).  step -> control: the vote words are memory-side atomics, drained (s_waitcnt vmcnt(0))
// before the workgroup's relaxed fetch_add reports in; the control workgroup fetches and clears them where the atomics ran and invalidates its own L1 once per step for
// its private state.  This relies on gfx950's write-through L1 / memory-side atomics and is guarded by the architecture check of sp_ctx_create (gfx950 only).
// All workgroups have to be resident together: the host admits a batch only within a budget of CUs (run_chunk), launches the control workgroups first and the step
// workgroups only when all of them have started; if they have not within PERSIST_READY_S the batch is aborted and run launch by launch.  A wait that lasts longer than
// PERSIST_TIMEOUT ticks of the 100 MHz clock, or a search that passes step_cap steps, raises the batch's abort word: every loop ends and the host reports the failure.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr long long PERSIST_TIMEOUT = 400000000ll;                 // 4 s
constexpr double PERSIST_READY_S = 0.5;                            // how long the host waits for every control workgroup of a batch to start
__device__ __forceinline__ bool persist_wait(const uint32_t* word, uint32_t at_least, uint32_t* abort_word) {
    const long long t0 = wall_clock64();
    for (uint32_t spins = 1;; ++spins) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= at_least) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63u) == 0) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (wall_clock64() - t0 > PERSIST_TIMEOUT) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
}

template <int MAXP>
__global__ void __launch_bounds__(CWAVES * SP_WAVE, SP_K8_MIN_WAVES) cons_step_persist_kernel(ConsBatchT<MAXP> B) {
    __shared__ int go_on;
    const int pi = block_problem<MAXP>(B);
    uint32_t* sy = B.sync + 4 * pi; uint32_t* abort_word = B.sync + 4 * B.n_prob;
    for (uint32_t k = 1;; ++k) {
        if (threadIdx.x == 0) go_on = persist_wait(sy + 0, k - 1, abort_word) ? 1 : 0;
        __syncthreads();
        if (!go_on) break;
        // (what the control step wrote -- work order, node lengths, consensus bytes -- is loaded past the L1, the vote words it cleared were cleared where the atomics run:
        //  no acquire; a wave's states are its own from step to step)
        if (coh_load(&B.p[pi].work->done)) break;
        cons_step_body<MAXP>(B, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // this wave's atomic adds have been acknowledged before the workgroup reports in
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(sy + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MAXP>
__global__ void __launch_bounds__(1024) cons_control_persist_kernel(ConsBatchT<MAXP> B) {
    __shared__ int go_on;
    const int pi = blockIdx.x;
    const uint32_t nb = (uint32_t)B.p[pi].n_blocks;
    uint32_t* sy = B.sync + 4 * pi; uint32_t* abort_word = B.sync + 4 * B.n_prob;
    // a control workgroup takes a whole CU (16 waves at 128 registers): they are started first and say so, the step workgroups are launched once all of them
    // have a CU (with a step workgroup on every CU and spinning, no control workgroup would ever start)
    if (threadIdx.x == 0 && B.ready) __hip_atomic_fetch_add(B.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (uint32_t k = 1;; ++k) {
        if (threadIdx.x == 0) {
            go_on = persist_wait(sy + 1, nb * k, abort_word) ? 1 : 0;
            if (go_on && k > B.step_cap) { __hip_atomic_store(abort_word, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); go_on = 0; }      // the search did not finish
        }
        __syncthreads();
        if (!go_on) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                      // (this CU's L1 only: the workgroup's own state of the step before comes from L2, not from stale lines)
        cons_control_body<MAXP>(B);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // every thread's write-through stores have been acknowledged ...
        __syncthreads();                                                        // ... before one of them raises the word
        const bool done = coh_load(&B.p[pi].work->done) != 0;
        if (threadIdx.x == 0) __hip_atomic_store(sy + 0, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done) break;
    }
    if (threadIdx.x == 0 && B.prog) {
        const uint32_t ab = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ab) __hip_atomic_store(B.prog + 2 * pi + 1, 1u + ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // "ended" with 2: a wait timed out (or the host gave the batch up); 3: a search passed the step cap
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
    ri.n = S.reads.len[rid]; ri.off0 = S.offsets ? S.offsets[r] : -1; ri.off = ri.off0 < 0 ? -1 : ri.off0 + max(0, min(S.cmp_len, ri.n)); ri.pad = 0;
    info[S.first + r] = ri;
}

// the same for ALL the problems of a batch in one launch (a cohort's lockstep round holds hundreds of problems: a launch each was 2 ms of a round): thread t is
// read t of the flattened array; its problem is the last one that starts at or before t
__global__ void cons_setup_many_kernel(const ConsSetup* __restrict__ S, int n_prob, int total, ReadInfo* __restrict__ info) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int lo = 0, hi = n_prob - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (S[mid].first <= t) lo = mid; else hi = mid - 1; }
    while (lo + 1 < n_prob && S[lo].n == 0 && S[lo + 1].first <= t) ++lo;          // (problems without reads share their start with their successor)
    const ConsSetup& q = S[lo];
    const int r = t - q.first;
    if (r < 0 || r >= q.n) return;
    const uint32_t rid = q.idx ? q.idx[r] : (uint32_t)r;
    ReadInfo ri;
    ri.w = q.reads.words + q.reads.word_off[rid];
    ri.np = q.reads.nplane ? q.reads.nplane + q.reads.word_off[rid] : nullptr;
    ri.n = q.reads.len[rid]; ri.off0 = q.offsets ? q.offsets[r] : -1; ri.off = ri.off0 < 0 ? -1 : ri.off0 + max(0, min(q.cmp_len, ri.n)); ri.pad = 0;
    info[t] = ri;
}

// scores and assignment of the reads on the complete node the search ended with
// One wavefront per workgroup (workgroup b stands for wave b % CWAVES of the step kernel's workgroup b / CWAVES): beside K1's grid of single-wave workgroups, which refill
// every wave slot that frees up, an 8-wave workgroup can wait for two free slots on each SIMD of one CU until that grid is through -- this 6 us kernel took 8-10 ms twice
// in four steps of a trace (profiles/r04/rocprof_r04_kernel_stats.csv), at the end of a consensus batch of the other lane.
template <int MAXP>
__global__ void __launch_bounds__(SP_WAVE) cons_finalize_kernel(ConsBatchT<MAXP> B, uint8_t* is_cons1, int32_t* score1, int32_t* score2) {
    const int vblock = (int)blockIdx.x / CWAVES, wave = (int)blockIdx.x % CWAVES;
    const int pi = block_problem<MAXP>(B, vblock);
    const ConsParams P = B.p[pi];
    const int lane = threadIdx.x;
    const int best = P.srch->best_node;
    const size_t plane = (size_t)B.total;
    if (vblock == P.first_block && wave == 0) {
        // everything the host reads of the problem goes to one output region: the search record, the winning node's shape, its bases
        for (int x = threadIdx.x; x < (int)(sizeof(CSearch) / 4); x += blockDim.x) ((uint32_t*)P.out_srch)[x] = ((const uint32_t*)P.srch)[x];
        if (threadIdx.x == 0) {
            ConsRes r; r.best = best; r.dual = 0; r.split_at = -1; r.len1 = r.len2 = 0; r.pad = 0;
            if (best >= 0) { const CNode* x = P.nodes + best; r.dual = x->dual; r.split_at = x->split_at; r.len1 = x->len[0]; r.len2 = x->dual ? x->len[1] : 0; }
            *P.out_res = r;
        }
        if (best >= 0) {
            // (four bases per load: behind persistent kernels the winner's bytes come from memory, not from L2 -- they were stored write-through --, and a byte per load
            //  with a division per byte made this the slowest part of the batch's tail, 0.4 ms)
            const uint8_t* src = P.C + (size_t)best * 2 * P.cs;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + (size_t)i * P.cs);       // (the rows are 16-byte aligned and cs >= cap rounded up to 16)
                uint8_t* dst = P.out_cons + (size_t)i * P.cap;
                for (int x = (int)threadIdx.x * 4; x < P.cap; x += (int)blockDim.x * 4) {
                    const uint32_t w = s4[x >> 2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (x + q < P.cap) dst[x + q] = (uint8_t)(w >> (8 * q));
                }
            }
        }
    }
    for (int rr = 0; rr < P.rpw; ++rr) {
        const int r = ((vblock - P.first_block) * CWAVES + wave) * P.rpw + rr;
        if (r >= P.n) break;
        const size_t g = (size_t)P.first + r;
        int sc[2] = { -1, -1 };
        if (best >= 0) {
            const CNode* x = P.nodes + best;
            const int n = B.info[g].n, cur = x->cur, dual = x->dual;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !dual) continue;
                const size_t p = state_plane(best, cur, i) * plane + g;
                const ConsMeta m = B.meta[p];
                if (!(m.flags & F_ACTIVE) || (m.flags & F_LOST)) continue;
                int e = m.e;
                if (!P.et) {
                    const int h = h_load(B.H, p * CB + lane), k = lane - CH;
                    int rest = (h >= 0 && h + k == x->len[i] - m.c0) ? n - h : (1 << 30);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(rest, o); rest = other < rest ? other : rest; }
                    if (rest < (1 << 30)) e += rest;
                }
                sc[i] = e;
            }
        }
        if (lane == 0) {
            score1[g] = sc[0]; score2[g] = sc[1];
            is_cons1[g] = !(sc[1] >= 0 && (sc[0] < 0 || sc[1] < sc[0]));
        }
    }
}

} // namespace

// CUs (in halves: a step workgroup takes half a CU, a control workgroup a whole one) the persistent batches of this process hold on each device
// and how many persistent batches run at once: each holds two hardware queues for its whole length (its step kernel and its control kernel wait for each other), and two
// batches whose kernels queue up behind one another on shared queues wait for ever (seen with eight cohort streams, each with a batch of its own, beside the HLA half's
// streams on the 16 queues of a process: the four-second time-out).  PERSIST_MAX_BATCHES at once, the others go the launch-pair way.
constexpr int PERSIST_MAX_BATCHES = 4;
struct PersistLease {
    int device = -1, taken = 0;
    static std::atomic<int>& used(int device) { static std::atomic<int> u[64]; return u[device & 63]; }
    static std::atomic<int>& batches(int device) { static std::atomic<int> b[64]; return b[device & 63]; }
    bool take(int dev, int budget, int want, int max_batches) {
        if (batches(dev).fetch_add(1) >= max_batches) { batches(dev).fetch_sub(1); return false; }
        std::atomic<int>& u = used(dev);
        int cur = u.load();
        while (cur + want <= budget) if (u.compare_exchange_weak(cur, cur + want)) { device = dev; taken = want; return true; }
        batches(dev).fetch_sub(1);
        return false;
    }
    void give_back() { if (taken) { used(device).fetch_sub(taken); batches(device).fetch_sub(1); taken = 0; } }
    ~PersistLease() { give_back(); }
};

// host side of a batch of at most CMAXP problems (or any number with the descriptors in device memory): all of them advance one
// step + control launch pair at a time until every search has ended
template <int MAXP>
static int32_t run_chunk(sp_ctx* ctx, uint32_t n_prob, const sp_cons_problem* probs, sp_cons_output* outs) {
    hipStream_t st = ctx->stream;
    HostMarks hm(ctx);
    ConsBatchT<MAXP> B; std::memset(&B, 0, sizeof B);
    B.n_prob = (int)n_prob;
    std::vector<ConsParams> hp(n_prob);                      // the descriptors; they end up in the kernel arguments or, for MAXP == 0, in device memory
    std::vector<int> block_prob, cluster_prob;
    std::vector<ConsSetup> setup(n_prob);
    std::vector<uint32_t> h_idx; std::vector<int32_t> h_off;
    std::vector<size_t> idx_at(n_prob), off_at(n_prob), c_at(n_prob), proc_at(n_prob);
    size_t total = 0, c_bytes = 0, proc_bytes = 0; int max_cap = 0, n_blocks = 0, n_clusters = 0;
    std::vector<int> expect(n_prob, 0);
    // Persistent mode (see cons_step_persist_kernel): every problem small enough to do without a reduce launch, and all workgroups of the batch -- step workgroups at
    // two per CU, a control workgroup per problem and CU -- resident together within the device's budget of CUs (32 stay free for everybody else; a process-wide
    // count per device, taken for the length of the batch).  A batch that is too wide gives its waves up to four reads each; one that still does not fit, or finds the
    // budget taken by other batches, runs launch by launch as before.
    PersistLease lease;
    int persist_rpw = 0;
    // side orders per step (sp_ctx_set_option "k8_side_orders", default NWORK - 1): one more row of workgroups in every step launch each.  A batch of very many workgroups (a
    // cohort's first lockstep rounds) keeps to the search's own order: its launches are wide already and rows of workgroups that mostly find no order still have to be dealt out
    int nside = std::max(0, std::min(ctx->k8_side_orders, NWORK - 1));
    {
        uint64_t nb_all = 0;
        for (uint32_t p = 0; p < n_prob; ++p) nb_all += ((probs[p].read_idx ? probs[p].n : probs[p].reads->n) + CWAVES - 1) / CWAVES;
        if (nb_all * CWAVES > (uint64_t)ctx->k8_side_max_blocks * 8) nside = 0;      // (the option counts workgroups of eight waves)
    }
    // Consensus batches of this process under way on the device, this one included from here on; and, for the library's own choice of mode, when two batches that could
    // run as persistent kernels (single samples' searches) last ran side by side: a host that keeps several samples in flight on one device
    struct InFlight {
        int dev; bool cand; int others = 0;
        static std::atomic<int>& count(int d) { static std::atomic<int> c[64]; return c[d & 63]; }
        static std::atomic<int>& cands(int d) { static std::atomic<int> c[64]; return c[d & 63]; }
        static std::atomic<long long>& overlap_ns(int d) { static std::atomic<long long> t[64]; return t[d & 63]; }
        static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
        bool recent_overlap = false;
        InFlight(int d, bool candidate) : dev(d), cand(candidate) {
            others = count(dev).fetch_add(1);
            const long long t = now_ns();
            if (cand && cands(dev).fetch_add(1) > 0) overlap_ns(dev).store(t);
            const long long last = overlap_ns(dev).load();
            recent_overlap = last != 0 && t - last < 1000000000ll;
        }
        ~InFlight() { count(dev).fetch_sub(1); if (cand) cands(dev).fetch_sub(1); }
    } in_flight(ctx->device, n_prob > 0 && n_prob <= 8 && [&] { uint64_t f = 2 * (uint64_t)n_prob; for (uint32_t p = 0; p < n_prob; ++p) f += ((probs[p].read_idx ? probs[p].n : probs[p].reads->n) + CWAVES - 1) / CWAVES; return f > (uint64_t)PERSIST_LIGHT; }());
    // (k8_persistent 2 = the library decides.  A batch of at most eight problems, when the streams have hardware queues of their own and the mode has not just failed here, and
    //  -- a batch of a LIGHT footprint, at most PERSIST_LIGHT resident workgroups (the late levels of a cohort call's lockstep searches: a few problems of a hundred reads):
    //     whenever no two heavy batches ran side by side within the last second (a host with several samples in flight: there the light batches of one sample run beside the
    //     other samples' chains and cost them 5 %); up to PERSIST_MAX_BATCHES of them at once.  A 256-sample cohort 330 -> 349-362 samples/s, a rank's share of 32 samples
    //     0.169-0.184 -> 0.138-0.145 s against launch pairs;
    //  -- a heavier one (a single 2,000-read sample: 250 step workgroups, half of the device's register files for the length of the batch): only when no other consensus batch
    //     of the process is under way on the device AND no two such batches ran side by side within the last second, one at a time.  Resident workgroups are a third faster for a
    //     chain that has the device to itself and take the CUs from everything that runs beside them: with four CYP2D6 samples in flight a launch pair per step made 301-324k
    //     reads/s of bench.py's stream, persistent kernels whenever a batch happened to start alone 282-303k, one persistent batch at a time 272k)
    uint64_t footprint = 2 * (uint64_t)n_prob;
    for (uint32_t p = 0; p < n_prob; ++p) footprint += ((probs[p].read_idx ? probs[p].n : probs[p].reads->n) + CWAVES - 1) / CWAVES;
    const bool light = footprint <= (uint64_t)PERSIST_LIGHT;
    const bool persist_wanted = ctx->k8_persistent == 1 || (ctx->k8_persistent == 2 && n_prob <= 8 && ctx->hw_queues_effective >= 16 && ctx->k8_persist_failures < 3 &&
                                                            !in_flight.recent_overlap && light);      // (round 6: a heavy batch -- a single large sample -- runs faster as launch pairs
                                                                                                               //  now that a launch carries side orders and branching windows, which the resident
                                                                                                               //  workgroups do not: one CYP2D6 lane beside the HLA lane 183k -> 199k reads/s)
    if (ctx->k8_persist_backoff > 0) --ctx->k8_persist_backoff;
    else if (persist_wanted && n_prob > 0) {
        uint64_t blocks1 = 0; bool small = true;
        for (uint32_t p = 0; p < n_prob; ++p) {
            const uint32_t n = probs[p].read_idx ? probs[p].n : probs[p].reads->n;
            const uint32_t nb = (n + CWAVES - 1) / CWAVES;
            small = small && nb <= (uint32_t)PERSIST_BLOCKS;
            blocks1 += nb;
        }
        constexpr int WG_PER_CU = SP_K8_MIN_WAVES * 4 / CWAVES;             // step workgroups a CU holds at the kernel's register budget
        int budget = WG_PER_CU * std::max(0, ctx->num_cus - 32); const int ctl = 2 * (int)n_prob;
        { const char* e = std::getenv("SP_K8_BUDGET"); if (e && *e) budget = std::atoi(e); }
        if (small && ctl < budget) {
            for (int scale = 1; scale <= 4 && !persist_rpw; ++scale) {
                uint64_t nb = 0;
                for (uint32_t p = 0; p < n_prob; ++p) { const uint32_t n = probs[p].read_idx ? probs[p].n : probs[p].reads->n; nb += (n + CWAVES * scale - 1) / (CWAVES * scale); }
                // (the library's own choice, k8_persistent 2: one persistent batch per device at a time -- the resident workgroups of two would take the CUs the other's and
                //  everybody else's kernels need; a second batch in flight runs as launch pairs in the gaps of the first)
                if ((int64_t)nb + ctl <= budget && lease.take(ctx->device, budget, (int)nb + ctl, (ctx->k8_persistent == 2 && !light) ? PERSIST_AUTO_BATCHES : PERSIST_MAX_BATCHES)) persist_rpw = scale;
                else if ((int64_t)nb + ctl <= budget) break;                       // it would fit, but the budget is taken right now
            }
        }
    }
    for (uint32_t p = 0; p < n_prob; ++p) {
        const sp_cons_problem& q = probs[p];
        const uint32_t n = q.read_idx ? q.n : q.reads->n;
        ConsParams& P = hp[p];
        P.n = (int)n; P.cap = (int)outs[p].cap - 1;                                   // one byte of the caller's buffer is the NUL
        P.cs = (std::max(P.cap, 1) + 1 + 15) & ~15;
        P.first = (int)total; P.first_block = n_blocks;
        P.min_count = q.cfg.min_count; P.delta = q.cfg.dual_max_ed_delta; P.et = q.cfg.allow_early_termination != 0; P.allow_dual = q.cfg.allow_dual != 0;
        P.window = q.cfg.offset_window; P.cmp_len = q.cfg.offset_compare_length; P.min_af = q.cfg.min_af;
        P.rpw = persist_rpw ? persist_rpw : n > 65536 ? 4 : n > 32768 ? 2 : 1;       // one read per wave while that stays below 4,096 workgroups
        const uint32_t per_block = (uint32_t)(CWAVES * P.rpw);
        const uint32_t nb = (n + per_block - 1) / per_block;
        P.n_blocks = (int)nb;
        P.first_cluster = n_clusters; P.n_clusters = (int)((nb + CLUSTER - 1) / CLUSTER);
        n_clusters += P.n_clusters;
        n_blocks += (int)nb;
        if (MAXP == 0) { block_prob.insert(block_prob.end(), nb, (int)p); cluster_prob.insert(cluster_prob.end(), (size_t)P.n_clusters, (int)p); }
        setup[p].reads = q.reads->view(); setup[p].n = (int)n; setup[p].first = (int)total;
        total += (size_t)nb * per_block;
        idx_at[p] = h_idx.size(); if (q.read_idx) h_idx.insert(h_idx.end(), q.read_idx, q.read_idx + n);
        off_at[p] = h_off.size(); if (q.offsets) h_off.insert(h_off.end(), q.offsets, q.offsets + n);
        c_at[p] = c_bytes; c_bytes += (size_t)NQ * 2 * (size_t)P.cs;
        proc_at[p] = proc_bytes; proc_bytes += ((size_t)std::max(P.cap, 1) + 2 + 15) & ~(size_t)15;
        max_cap = std::max(max_cap, P.cap);
        // how long the consensus can get: the furthest any read of the problem reaches (its placement offset + its length), within cap
        int reach = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t r = q.read_idx ? q.read_idx[i] : i;
            const int o = q.offsets && q.offsets[i] > 0 ? q.offsets[i] : 0;
            reach = std::max(reach, o + (r < q.reads->h_len.size() ? q.reads->h_len[r] : P.cap));
        }
        expect[p] = std::min(P.cap, reach);
    }
    if (n_blocks == 0) return SP_OK;
    const size_t planes = (size_t)NQ * 4;                   // [node][slot][consensus]
    // one staging region each way (pinned host side): the inputs go up in one copy, the results come down in one
    size_t in_bytes = 0, out_bytes = 0, zero_bytes = 0;
    auto place = [](size_t& total_bytes, size_t bytes) { const size_t at = (total_bytes + 15) & ~(size_t)15; total_bytes = at + bytes; return at; };
    const size_t in_idx = place(in_bytes, sizeof(uint32_t) * h_idx.size()), in_off = place(in_bytes, sizeof(int32_t) * h_off.size());
    const size_t in_work = place(in_bytes, sizeof(CWork) * NWORK * n_prob), in_srch = place(in_bytes, sizeof(CSearch) * n_prob);
    const size_t in_probs = place(in_bytes, MAXP == 0 ? sizeof(ConsParams) * n_prob : 0), in_bp = place(in_bytes, sizeof(int) * block_prob.size());
    const size_t in_cp = place(in_bytes, sizeof(int) * cluster_prob.size());
    const size_t in_setup = place(in_bytes, n_prob > 8 ? sizeof(ConsSetup) * n_prob : 0);
    const size_t out_srch = place(out_bytes, sizeof(CSearch) * n_prob), out_res = place(out_bytes, sizeof(ConsRes) * n_prob);
    const size_t out_is1 = place(out_bytes, total), out_sc = place(out_bytes, sizeof(int32_t) * 2 * total);
    std::vector<size_t> out_cons(n_prob);
    for (uint32_t p = 0; p < n_prob; ++p) out_cons[p] = place(out_bytes, (size_t)2 * std::max(hp[p].cap, 1));
    const size_t zero_nodes = place(zero_bytes, sizeof(CNode) * NQ * n_prob), zero_proc = place(zero_bytes, proc_bytes), zero_info = place(zero_bytes, sizeof(ReadInfo) * total), zero_memo = place(zero_bytes, sizeof(PlaceMemo) * 2 * total * (size_t)(1 + nside));
    const size_t zero_sync = place(zero_bytes, sizeof(uint32_t) * (4 * (size_t)n_prob + 4));
    uint8_t* d_in = (uint8_t*)sp_pool(ctx, "cons_in", in_bytes + 16); uint8_t* h_in = (uint8_t*)sp_host_pool(ctx, "cons_in", in_bytes + 16);
    uint8_t* d_out = (uint8_t*)sp_pool(ctx, "cons_out", out_bytes + 16); uint8_t* h_out = (uint8_t*)sp_host_pool(ctx, "cons_out", out_bytes + 16);
    uint8_t* d_zero = (uint8_t*)sp_pool(ctx, "cons_zero", zero_bytes + 16);
    if (!d_in || !h_in || !d_out || !h_out || !d_zero) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus staging");
    uint32_t* d_idx = (uint32_t*)(d_in + in_idx); int32_t* d_off = (int32_t*)(d_in + in_off);
    uint8_t* d_C = (uint8_t*)sp_pool(ctx, "cons_C", c_bytes);
    CWork* d_work = (CWork*)(d_in + in_work);
    CSearch* d_srch = (CSearch*)(d_in + in_srch);
    CNode* d_nodes = (CNode*)(d_zero + zero_nodes);
    uint32_t* d_la = (uint32_t*)sp_pool(ctx, "cons_la", sizeof(uint32_t) * (size_t)NQ * 2 * CW * 4 * n_prob);
    uint8_t* d_proc = d_zero + zero_proc;
    ReadInfo* d_info = (ReadInfo*)(d_zero + zero_info);
    B.memo = (PlaceMemo*)(d_zero + zero_memo);
    B.sync = persist_rpw ? (uint32_t*)(d_zero + zero_sync) : nullptr;
    B.info = d_info; B.total = (int)total;
    B.H = (uint16_t*)sp_pool(ctx, "cons_H", sizeof(uint16_t) * planes * total * CB);
    B.meta = (ConsMeta*)sp_pool(ctx, "cons_meta", sizeof(ConsMeta) * planes * total);
    const size_t wblocks = (size_t)n_blocks + (size_t)NWORK * n_prob;         // the workgroups' word blocks + one block per problem and order its workgroups add to (acc_block + order)
    B.PV = (unsigned long long*)sp_pool(ctx, "cons_pv", sizeof(unsigned long long) * wblocks * 2 * (CW + 1));
    B.PE = (uint32_t*)sp_pool(ctx, "cons_pe", sizeof(uint32_t) * wblocks * 2 * (CW + 1));
    B.PL = (unsigned long long*)sp_pool(ctx, "cons_pl", sizeof(unsigned long long) * wblocks * 2 * CW);
    B.PC = (uint32_t*)sp_pool(ctx, "cons_pc", sizeof(uint32_t) * wblocks * (CW + 1));
    B.PR = (uint32_t*)sp_pool(ctx, "cons_pr", sizeof(uint32_t) * wblocks * (CW + 1));
    B.Q = (uint32_t*)sp_pool(ctx, "cons_q", sizeof(uint32_t) * (size_t)n_clusters * QE);
#ifdef SP_K8_TIMING
    B.dbg = (unsigned long long*)sp_pool(ctx, "cons_dbg", (size_t)SP_K8_DBG_LAUNCHES * SP_K8_DBG_READS * 8);
    (void)hipMemsetAsync(B.dbg, 0, (size_t)SP_K8_DBG_LAUNCHES * SP_K8_DBG_READS * 8, st);
#endif
    uint8_t* d_is1 = d_out + out_is1;
    int32_t* d_sc = (int32_t*)(d_out + out_sc);
    B.step_t = nullptr;
    if (ctx->profiling) {
        B.step_t = (unsigned long long*)sp_pool(ctx, "cons_step_t", sizeof(unsigned long long) * 2 * n_prob);
        if (B.step_t) {
            std::vector<unsigned long long> init((size_t)2 * n_prob);
            for (uint32_t p = 0; p < n_prob; ++p) { init[2 * p] = ~0ull; init[2 * p + 1] = 0ull; }
            unsigned long long* h_st = (unsigned long long*)sp_host_pool(ctx, "cons_step_t", sizeof(unsigned long long) * 2 * n_prob);
            if (h_st) { std::memcpy(h_st, init.data(), init.size() * 8); (void)hipMemcpyAsync(B.step_t, h_st, init.size() * 8, hipMemcpyHostToDevice, st); } else B.step_t = nullptr;
        }
    }
    uint32_t* h_prog = (uint32_t*)sp_host_pool(ctx, "cons_prog", sizeof(uint32_t) * 2 * n_prob);
    if (!h_prog) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus progress words");
    std::memset(h_prog, 0, sizeof(uint32_t) * 2 * n_prob);
    { void* dp = nullptr; SP_HIP_CHECK(ctx, hipHostGetDevicePointer(&dp, h_prog, 0)); B.prog = (uint32_t*)dp; }
    CWork* h_work = (CWork*)sp_host_pool(ctx, "cons_work", sizeof(CWork) * n_prob);
    CWork* h_work0 = (CWork*)(h_in + in_work); CSearch* h_srch0 = (CSearch*)(h_in + in_srch);
    const CSearch* h_srch = (const CSearch*)(h_out + out_srch);
    if (!d_idx || !d_off || !d_C || !d_work || !d_srch || !d_nodes || !d_la || !d_proc || !d_info || !B.H || !B.meta || !B.PV || !B.PE || !B.PL || !B.PC || !B.PR || !B.Q ||
        !h_work)
        return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus buffers");
    for (uint32_t p = 0; p < n_prob; ++p) {
        setup[p].idx = probs[p].read_idx ? d_idx + idx_at[p] : nullptr;
        setup[p].offsets = probs[p].offsets ? d_off + off_at[p] : nullptr; setup[p].cmp_len = probs[p].cfg.offset_compare_length; setup[p].pad_ = 0;
        hp[p].C = d_C + c_at[p]; hp[p].work = d_work + (size_t)p * NWORK; hp[p].srch = d_srch + p; hp[p].nodes = d_nodes + (size_t)p * NQ;
        hp[p].la = d_la + (size_t)p * NQ * 2 * CW * 4; hp[p].processed = d_proc + proc_at[p];
        hp[p].acc_block = n_blocks + (int)p * NWORK;
        hp[p].out_cons = d_out + out_cons[p]; hp[p].out_res = (ConsRes*)(d_out + out_res) + p; hp[p].out_srch = (CSearch*)(d_out + out_srch) + p;
    }
    if constexpr (MAXP == 0) {
        std::memcpy(h_in + in_probs, hp.data(), sizeof(ConsParams) * n_prob);
        std::memcpy(h_in + in_bp, block_prob.data(), sizeof(int) * block_prob.size());
        std::memcpy(h_in + in_cp, cluster_prob.data(), sizeof(int) * cluster_prob.size());
        B.p = (const ConsParams*)(d_in + in_probs); B.block_prob = (const int*)(d_in + in_bp); B.cluster_prob = (const int*)(d_in + in_cp);
    } else {
        for (uint32_t p = 0; p < n_prob; ++p) B.p[p] = hp[p];
    }
    for (uint32_t p = 0; p < n_prob; ++p) {
        std::memset(&h_work0[(size_t)p * NWORK], 0, sizeof(CWork) * NWORK);       // (the side orders of the first step: none)
        CWork& w = h_work0[(size_t)p * NWORK]; w.mode = M_INIT; w.node = 0; w.in_slot = 0; w.split_at = -1;
        CSearch& s = h_srch0[p]; std::memset(&s, 0, sizeof s); s.best_node = -1; s.inflight = -1;
        const sp_cons_config& cf = probs[p].cfg;
        s.max_queue = cf.max_queue_size > 0 ? std::min(cf.max_queue_size, NQ - MAXKIDS - 2) : 20;
        s.per_size = cf.max_capacity_per_size > 0 ? std::min(cf.max_capacity_per_size, 255) : 10;
        s.wo_constraint = cf.max_nodes_wo_constraint > 0 ? cf.max_nodes_wo_constraint : 1000;
    }
    if (!h_idx.empty()) std::memcpy(h_in + in_idx, h_idx.data(), sizeof(uint32_t) * h_idx.size());
    if (!h_off.empty()) std::memcpy(h_in + in_off, h_off.data(), sizeof(int32_t) * h_off.size());
    if (n_prob > 8) std::memcpy(h_in + in_setup, setup.data(), sizeof(ConsSetup) * n_prob);
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemsetAsync(d_zero, 0, zero_bytes, st));
    {   // the blocks the workgroups of small problems add their words to start at zero (the control step clears what it reads)
        const size_t nb = (size_t)n_blocks, np = (size_t)n_prob * NWORK;
        SP_HIP_CHECK(ctx, hipMemsetAsync(B.PV + nb * 2 * (CW + 1), 0, sizeof(unsigned long long) * np * 2 * (CW + 1), st));
        SP_HIP_CHECK(ctx, hipMemsetAsync(B.PE + nb * 2 * (CW + 1), 0, sizeof(uint32_t) * np * 2 * (CW + 1), st));
        SP_HIP_CHECK(ctx, hipMemsetAsync(B.PL + nb * 2 * CW, 0, sizeof(unsigned long long) * np * 2 * CW, st));
        SP_HIP_CHECK(ctx, hipMemsetAsync(B.PC + nb * (CW + 1), 0, sizeof(uint32_t) * np * (CW + 1), st));
        SP_HIP_CHECK(ctx, hipMemsetAsync(B.PR + nb * (CW + 1), 0, sizeof(uint32_t) * np * (CW + 1), st));
    }
    if (n_prob > 8) hipLaunchKernelGGL(cons_setup_many_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const ConsSetup*)(d_in + in_setup), (int)n_prob, (int)total, d_info);
    else for (uint32_t p = 0; p < n_prob; ++p)
        if (setup[p].n) hipLaunchKernelGGL(cons_setup_kernel, dim3((setup[p].n + 255) / 256), dim3(256), 0, st, setup[p], d_info);

    B.nside = persist_rpw ? 0 : nside;
    B.k8_compound = ctx->k8_compound;
    const dim3 grid((uint32_t)n_blocks, (uint32_t)(1 + B.nside)), block(CWAVES * SP_WAVE);
    // (the step kernel's instantiation for batches that fit the device at one workgroup per CU, give or take a few: cons_step_wide_kernel.  A 2,000-read CYP2D6 sample's first batch
    //  has 271 workgroups -- its 2,162 region segments -- and is still faster there)
    const bool one_round = n_blocks <= (ctx->num_cus + ctx->num_cus / 4) * (8 / CWAVES);
    size_t proc_lds = ((size_t)max_cap + 2 + 15) & ~(size_t)15;
    {   // the control kernel's node table and vote sums are static LDS; the per-length counters come on top (160 KiB per workgroup on gfx950)
        hipFuncAttributes fa;
        SP_HIP_CHECK(ctx, hipFuncGetAttributes(&fa, (const void*)cons_control_kernel<MAXP>));
        if (fa.sharedSizeBytes + proc_lds > 160 * 1024) return sp_fail(ctx, SP_ERR_TOO_LONG, "sp_consensus: cap must stay below ~65,000");
    }
    SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)cons_control_kernel<MAXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)proc_lds));
    uint64_t pairs = 0;
    bool persist_ran = false;                                           // the batch's two persistent kernels were launched (and may be running)
    hm.mark("host:k8_prologue");
    double t_launch = 0.0;                                              // host time inside the launch calls (sp_profile_get "host:k8_launch": several host threads share the runtime's launch path)
    {
        ProfScope ps(ctx, "cons_steps", total);
        // The host stays a few launch triples ahead of the device and never waits for it: the control kernel of a problem writes the number
        // of steps it has made and, at the end, "done" into host memory (fine-grained pinned: the write is seen here without a copy or a
        // stream synchronisation), and the host reads those words between launches.  A search that ends leaves at most `ahead` empty triples
        // behind (~4 us per empty launch).  Before: a device-to-host copy + stream synchronisation every few triples, ~40 us of idle device
        // each, 35 of them per 10,000-read sample.
        const uint64_t limit = (uint64_t)64 * (uint64_t)(max_cap + 2) + 1024;
        const uint32_t ahead = 3;
        bool need_reduce = false;
        for (uint32_t p = 0; p < n_prob; ++p) need_reduce = need_reduce || hp[p].n_blocks > DIRECT_BLOCKS;
        volatile uint32_t* prog = h_prog;
        bool finished = false;
        if (persist_rpw) {
            // two kernels for the whole batch: the control workgroups on the context's control stream, the step workgroups on its own stream; the control stream
            // joins in behind the set-up work and hands back when its kernel has ended
            if (!ctx->ctl_stream && hipStreamCreateWithFlags(&ctx->ctl_stream, hipStreamNonBlocking) != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, "sp_consensus: control stream");
            if (!ctx->ev_fork && (hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess))
                return sp_fail(ctx, SP_ERR_HIP, "sp_consensus: events");
            SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)cons_control_persist_kernel<MAXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)proc_lds));
            SP_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, st));
            SP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->ctl_stream, ctx->ev_fork, 0));
            uint32_t* h_ready = (uint32_t*)sp_host_pool(ctx, "cons_ready", 64);
            if (!h_ready) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus ready word");
            *(volatile uint32_t*)h_ready = 0;
            { void* dp = nullptr; SP_HIP_CHECK(ctx, hipHostGetDevicePointer(&dp, h_ready, 0)); B.ready = (uint32_t*)dp; }
            B.step_cap = (uint32_t)std::min<uint64_t>(limit, 0xFFFFFFFFull);
            // from here on two kernels may be running: whatever goes wrong, both streams are waited for before this function returns (the lease and the pooled buffers
            // go back to other batches when it does)
            auto drain = [&]() { (void)hipStreamSynchronize(ctx->ctl_stream); (void)hipStreamSynchronize(st); };
            hipLaunchKernelGGL(cons_control_persist_kernel<MAXP>, dim3(n_prob), dim3(1024), proc_lds, ctx->ctl_stream, B);
            bool all_ready = true;
            {   // a control workgroup takes a whole CU: on a device whose CUs are held by others (another process's batches, which this process's lease cannot see) some
                // of them never start.  Then the batch is given up HERE -- the abort word ends the ones that did start -- and run launch by launch below, and the context
                // keeps away from the mode for a while
                const auto t_wait = std::chrono::steady_clock::now();
                const bool forced = std::getenv("SP_K8_FORCE_READY_TIMEOUT") != nullptr;      // (tests: take the time-out's path whatever the device does)
                while (forced || *(volatile uint32_t*)h_ready < n_prob) {
                    __builtin_ia32_pause();
                    if (forced || std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count() > PERSIST_READY_S) { all_ready = false; break; }
                }
            }
            if (all_ready) {
                hipLaunchKernelGGL(cons_step_persist_kernel<MAXP>, dim3((uint32_t)n_blocks), block, 0, st, B);
                const hipError_t e1 = hipGetLastError();
                const hipError_t e2 = e1 == hipSuccess ? hipEventRecord(ctx->ev_join, ctx->ctl_stream) : e1;
                const hipError_t e3 = e2 == hipSuccess ? hipStreamWaitEvent(st, ctx->ev_join, 0) : e2;
                if (e3 != hipSuccess) {
                    const uint32_t one = 1;                    // end the control workgroups (and any step workgroup that started), then wait for both streams
                    if (!ctx->copy_stream) (void)hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
                    if (ctx->copy_stream) { (void)hipMemcpyAsync(B.sync + 4 * n_prob, &one, 4, hipMemcpyHostToDevice, ctx->copy_stream); (void)hipStreamSynchronize(ctx->copy_stream); }
                    drain();
                    return sp_fail(ctx, SP_ERR_HIP, std::string("sp_consensus: persistent launch: ") + hipGetErrorString(e3));
                }
                finished = true; persist_ran = true;
            } else {
                const uint32_t one = 1;
                if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { drain(); return sp_fail(ctx, SP_ERR_HIP, "sp_consensus: copy stream"); }
                (void)hipMemcpyAsync(B.sync + 4 * n_prob, &one, 4, hipMemcpyHostToDevice, ctx->copy_stream);
                (void)hipStreamSynchronize(ctx->copy_stream);
                (void)hipStreamSynchronize(ctx->ctl_stream);                       // the control workgroups that had started have seen the word (they poll it) and ended
                lease.give_back();
                ctx->k8_persist_failures += 1; ctx->k8_persist_backoff = 64;
                ctx->warning = "persistent consensus kernels: the control workgroups of a batch found no free CUs within " + std::to_string(PERSIST_READY_S) +
                               " s (another process on this device?): the batch ran launch by launch, and so will the next 64";
                // the same batch as a launch pair per step: nothing has run yet but `ready` counts and abort marks
                (void)hipMemsetAsync(B.sync, 0, sizeof(uint32_t) * (4 * (size_t)n_prob + 1), st);
                std::memset(h_prog, 0, sizeof(uint32_t) * 2 * n_prob);
                B.sync = nullptr; B.ready = nullptr;
                persist_rpw = 0;
            }
        }
        while (!finished) {
            const auto tl0 = std::chrono::steady_clock::now();
            if (one_round) hipLaunchKernelGGL(cons_step_wide_kernel<MAXP>, grid, block, 0, st, B);
            else hipLaunchKernelGGL(cons_step_kernel<MAXP>, grid, block, 0, st, B);
            if (need_reduce) hipLaunchKernelGGL(cons_reduce_kernel<MAXP>, dim3((uint32_t)n_clusters * RSLICES), dim3(512), 0, st, B);
            hipLaunchKernelGGL(cons_control_kernel<MAXP>, dim3(n_prob), dim3(1024), proc_lds, st, B);
            t_launch += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count();
            ++pairs;
            uint64_t spins = 0;
            for (;;) {
                bool all = true; uint32_t slowest = 0xFFFFFFFFu;
                for (uint32_t p = 0; p < n_prob; ++p) if (!prog[2 * p + 1]) { all = false; const uint32_t v = prog[2 * p]; slowest = v < slowest ? v : slowest; }
                if (all) { finished = true; break; }
                if (pairs - slowest <= ahead) break;
                __builtin_ia32_pause();
                if ((++spins & 0xFFFFF) == 0) {                          // now and then: is the stream still alive?
                    const hipError_t q = hipStreamQuery(st);
                    if (q != hipSuccess && q != hipErrorNotReady) return sp_fail(ctx, SP_ERR_HIP, std::string("sp_consensus: ") + hipGetErrorString(q));
                    if (q == hipSuccess) {                                // everything enqueued has run and the words did not move
                        bool all2 = true; uint32_t slow2 = 0xFFFFFFFFu;
                        for (uint32_t p = 0; p < n_prob; ++p) if (!prog[2 * p + 1]) { all2 = false; const uint32_t v = prog[2 * p]; slow2 = v < slow2 ? v : slow2; }
                        if (all2) { finished = true; break; }
                        if (pairs - slow2 <= ahead) break;
                        return sp_fail(ctx, SP_ERR_HIP, "sp_consensus: the device stopped reporting progress");
                    }
                }
            }
            if (!finished && pairs >= limit) return sp_fail(ctx, SP_ERR_HIP, "sp_consensus: the search did not finish");
        }
    }
    // (with persistent kernels under way an error below still waits for both streams before it returns: the lease and the pooled buffers are other batches' the moment it does)
#define SP_K8_CHECK(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) { if (persist_ran) { (void)hipStreamSynchronize(ctx->ctl_stream); (void)hipStreamSynchronize(st); } \
        return sp_fail(ctx, SP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    SP_K8_CHECK(hipGetLastError());
    hm.mark("host:k8_loop");
    if (ctx->profiling) { auto& e = ctx->prof["host:k8_launch"]; e.ms += t_launch; e.launches += (uint32_t)(2 * pairs); }
    hipLaunchKernelGGL(cons_finalize_kernel<MAXP>, dim3((uint32_t)n_blocks * CWAVES), dim3(SP_WAVE), 0, st, B, d_is1, d_sc, d_sc + total);
    SP_K8_CHECK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    const uint8_t* h_is1 = h_out + out_is1; const int32_t* h_sc = (const int32_t*)(h_out + out_sc); const ConsRes* h_res = (const ConsRes*)(h_out + out_res);
    SP_K8_CHECK(hipStreamSynchronize(st));
    SP_K8_CHECK(hipGetLastError());
#undef SP_K8_CHECK
    lease.give_back();
    if (persist_ran) {
        for (uint32_t p = 0; p < n_prob; ++p) {
            if (h_prog[2 * p + 1] == 2u) {
                // a wait of the batch's kernels for each other timed out (other processes' resident workgroups on the device, which this process's lease cannot see): both kernels have
                // ended (the stream is synchronised).  Where the mode was the library's own choice the batch is run again from its start as launch pairs (the caller below), the
                // context stays away from the mode for a while and says so; a caller that asked for the mode gets the error
                ctx->k8_persist_failures += 1; ctx->k8_persist_backoff = 64;
                if (ctx->k8_persistent == 2) {
                    ctx->warning = "persistent consensus kernels: the kernels of a batch waited for each other for more than four seconds (another process on this device?): the batch ran "
                                   "again launch by launch, and so will the next 64";
                    return SP_K8_AGAIN;
                }
                return sp_fail(ctx, SP_ERR_HIP, "sp_consensus: the persistent kernels of a batch waited for each other for more than four seconds and gave up");
            }
            if (h_prog[2 * p + 1] == 3u) return sp_fail(ctx, SP_ERR_HIP, "sp_consensus: the search did not finish");
            pairs = std::max<uint64_t>(pairs, h_prog[2 * p]);
        }
        ctx->prof["cons_persistent_batches"].cells += 1;
    }
    hm.mark("host:k8_result_wait");
#ifdef SP_K8_TIMING
    if (std::getenv("SP_K8_DUMP")) {
        std::vector<unsigned long long> h((size_t)SP_K8_DBG_LAUNCHES * SP_K8_DBG_READS);
        (void)hipMemcpy(h.data(), B.dbg, h.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = std::fopen(std::getenv("SP_K8_DUMP"), "ab");
        if (f) { const unsigned long long tot = total; std::fwrite(&tot, 8, 1, f); std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
    }
#endif
    {   // launch statistics of the batch (sp_profile_get: cells = count)
        uint64_t cut = 0, ex = 0, pops = 0;
        for (uint32_t p = 0; p < n_prob; ++p) { cut += (uint64_t)h_srch[p].cut_windows; ex += (uint64_t)h_srch[p].expansions; pops += (uint64_t)h_srch[p].pops; }
        ctx->prof["cons_windows"].cells += pairs; ctx->prof["cons_windows"].launches += 3 * pairs;           // (two per step for a batch of small problems)
        ctx->prof["cons_cut_windows"].cells += cut; ctx->prof["cons_expansions"].cells += ex; ctx->prof["cons_columns"].cells += pops;
        for (uint32_t p = 0; p < n_prob; ++p) {
            ctx->prof["cons_side_windows"].cells += (uint64_t)h_srch[p].side_windows; ctx->prof["cons_side_expansions"].cells += (uint64_t)h_srch[p].side_expansions;
            ctx->prof["cons_adopted"].cells += (uint64_t)h_srch[p].adopted;
            ctx->prof["cons_compound"].cells += (uint64_t)h_srch[p].compound; ctx->prof["cons_compound_ok"].cells += (uint64_t)h_srch[p].compound_ok;
        }
#ifdef SP_K8_PF_PROBE
        {   // the batch's chain is its slowest problem's: steps as they are / without the window orders a launch before could have carried / without any such order
            uint64_t c0 = 0, c1 = 0, c2 = 0;
            for (uint32_t p = 0; p < n_prob; ++p) {
                const uint64_t st = (uint64_t)h_srch[p].windows + (uint64_t)h_srch[p].expansions, w = (uint64_t)h_srch[p].pf_win + (uint64_t)h_srch[p].pf_replay, e = (uint64_t)h_srch[p].pf_exp;
                c0 = std::max(c0, st); c1 = std::max(c1, st - w); c2 = std::max(c2, st - w - e);
                ctx->prof["cons_pf_win"].cells += (uint64_t)h_srch[p].pf_win; ctx->prof["cons_pf_replay"].cells += (uint64_t)h_srch[p].pf_replay; ctx->prof["cons_pf_exp"].cells += e;
                ctx->prof["cons_all_windows"].cells += (uint64_t)h_srch[p].windows;
            }
            ctx->prof["cons_pf_chain"].cells += c0; ctx->prof["cons_pf_chain_win"].cells += c1; ctx->prof["cons_pf_chain_all"].cells += c2;
        }
#endif
        // where the control kernel's time goes: ticks of the 100 MHz wall clock, the slowest problem of the batch (they run side by side)
        static const char* tick_names[4] = { "cons_ticks_reduce", "cons_ticks_result", "cons_ticks_search", "cons_ticks_tail" };
        for (int k = 0; k < 4; ++k) { long long m = 0; for (uint32_t p = 0; p < n_prob; ++p) m = std::max(m, h_srch[p].ticks[k]); ctx->prof[tick_names[k]].cells += (uint64_t)m; }
        // the critical path of the batch: the problem with the largest step + control + gap total (they run side by side; its chain is the batch's)
        if (B.step_t) {
            long long best = -1; uint32_t bp = 0;
            for (uint32_t p = 0; p < n_prob; ++p) {
                const long long tot = h_srch[p].step_ticks + h_srch[p].gap_ticks + h_srch[p].ticks[0] + h_srch[p].ticks[1] + h_srch[p].ticks[2] + h_srch[p].ticks[3];
                if (tot > best) { best = tot; bp = p; }
            }
            ctx->prof["cons_path_step_ticks"].cells += (uint64_t)h_srch[bp].step_ticks; ctx->prof["cons_path_gap_ticks"].cells += (uint64_t)h_srch[bp].gap_ticks; ctx->prof["cons_path_gap_ctl_ticks"].cells += (uint64_t)h_srch[bp].gap_ctl_ticks;
            ctx->prof["cons_path_control_ticks"].cells += (uint64_t)(h_srch[bp].ticks[0] + h_srch[bp].ticks[1] + h_srch[bp].ticks[2] + h_srch[bp].ticks[3]);
            ctx->prof["cons_path_steps"].cells += (uint64_t)h_srch[bp].steps;
        }
    }
    static const char dec[4] = { 'A', 'C', 'G', 'T' };
    int32_t rc = SP_OK;
    for (uint32_t p = 0; p < n_prob; ++p) {
        const ConsParams& P = hp[p]; sp_cons_output& o = outs[p];
        const ConsRes& res = h_res[p];
        const int best = res.best, len1 = res.len1, len2 = res.len2, dual = res.dual, split_at = res.split_at;
        if (best >= 0) {
            const uint8_t* c = h_out + out_cons[p];
            for (int x2 = 0; x2 < len1; ++x2) o.cons1[x2] = dec[c[x2] & 3];
            for (int x2 = 0; x2 < len2; ++x2) o.cons2[x2] = dec[(x2 < split_at ? c[x2] : c[(size_t)P.cap + x2]) & 3];
        }
        o.cons1[len1] = '\0'; o.cons2[len2] = '\0';
        for (int r = 0; r < P.n; ++r) { o.is_cons1[r] = h_is1[P.first + r]; o.score1[r] = h_sc[P.first + r]; o.score2[r] = h_sc[total + P.first + r]; }
        o.result.is_dual = dual; o.result.len1 = len1; o.result.len2 = len2; o.result.split_at = split_at;
        o.result.gave_up = best < 0 ? 1 : 0; o.result.best_total = 1; o.result.split_w2 = 0; o.result.split_total = 1;
        o.result.nodes_expanded = h_srch[p].pops;
        // a consensus that filled its buffer was still growing: the caller sized cap too small
        if (len1 >= P.cap || len2 >= P.cap) { o.status = SP_ERR_CAPACITY; rc = SP_ERR_CAPACITY; }
    }
    if (rc != SP_OK) sp_fail(ctx, rc, "sp_consensus: a consensus reached cap");
    hm.mark("host:k8_epilogue");
    return rc;
}

static int32_t run_batch(sp_ctx* ctx, uint32_t n_prob, const sp_cons_problem* probs, sp_cons_output* outs) {
    for (uint32_t p = 0; p < n_prob; ++p) {
        const sp_cons_problem& q = probs[p]; sp_cons_output& o = outs[p];
        if (!q.reads || !o.cons1 || !o.cons2 || o.cap == 0 || !o.is_cons1 || !o.score1 || !o.score2) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: null argument");
        if (q.cfg.offset_compare_length > 128 || q.cfg.offset_compare_length < 0 || q.cfg.offset_window < 0 || q.cfg.min_count < 0 ||
            (q.cfg.offset_compare_length > 64 && q.cfg.offset_window + q.cfg.offset_compare_length > ACT_CONS))
            return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: offset_compare_length must be in [0, 128] (in [0, 64] when offset_window + offset_compare_length exceeds 512)");
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
        auto run = [&]() { return k <= 4 ? run_chunk<4>(ctx, k, probs + at, outs + at) : k <= 8 ? run_chunk<8>(ctx, k, probs + at, outs + at)
                                  : k <= CMAXP ? run_chunk<CMAXP>(ctx, k, probs + at, outs + at) : run_chunk<0>(ctx, k, probs + at, outs + at); };
        int32_t e = run();
        if (e == SP_K8_AGAIN) e = run();                    // (the library's own persistent batch timed out: once more, as launch pairs -- the context's back-off sees to that)
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
    // DualConsensusDWFA: the same search with a second consensus allowed (round 1 ran a two-pass split policy here; the best-first
    // search decides by cost where to split)
    std::vector<sp_cons_problem> pr(problems, problems + n_problems);
    for (auto& q : pr) q.cfg.allow_dual = 1;
    return run_batch(ctx, n_problems, pr.data(), outputs);
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

// The multi-way consensus of SEVERAL independent problems (the samples of a cohort) in lockstep: every round all open groups of all problems
// are one batch of two-way searches -- the launches a round costs are those of its slowest search, not their sum over the problems.  Each job
// is what sp_consensus_priority does for it alone; a job that fails (more groups than max_groups, ...) carries its own status.
int32_t sp_consensus_priority_many(sp_ctx* ctx, uint32_t n_jobs, sp_priority_job* jobs) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (n_jobs && !jobs) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_priority_many: null argument");
    // kept[l]: the consensus of level l when the two-way search of exactly these members ended with ONE consensus at the configured fraction: it is the group's
    // consensus at that level, and only the levels without one are solved again at the end (a third of a CYP2D6 sample's launches were those repeats)
    struct Item { std::vector<uint32_t> members; uint32_t level; std::string key; int retry = 0; std::vector<std::string> kept; std::vector<uint8_t> have; };
    struct JobState { std::vector<Item> work, done; int half = 0; bool live = false; };
    // A search that gave up (no complete node: a mixture of more classes than a search holds consensuses can exhaust the queue and capacity
    // bounds) is run again with only the stronger differences as candidates; the split it finds is the split, the groups it leaves are solved
    // with the configured fraction again.
    static const double retry_min_af[4] = { 0.15, 0.20, 0.30, 0.40 };
    std::vector<JobState> S(n_jobs);
    for (uint32_t j = 0; j < n_jobs; ++j) {
        sp_priority_job& J = jobs[j];
        J.status = SP_OK; J.gave_up = 0;
        const sp_priority_problem* pr = J.problem;
        if (!pr || !J.n_groups || !J.group_of || !J.cons || !pr->levels || pr->n_levels == 0 || J.cap < 2) { J.status = SP_ERR_INVALID_ARG; continue; }
        bool ok = true;
        for (uint32_t l = 0; l < pr->n_levels; ++l) if (!pr->levels[l] || pr->levels[l]->n != pr->n) ok = false;
        if (!ok) { J.status = SP_ERR_INVALID_ARG; continue; }
        *J.n_groups = 0;
        if (pr->n == 0) continue;
        S[j].half = pr->cfg.offset_window / 2; S[j].live = true;
        // initial groups: unseeded reads first, then the seeds in ascending order
        std::map<int32_t, std::vector<uint32_t>> by_seed;
        for (uint32_t r = 0; r < pr->n; ++r) by_seed[pr->seeds ? (pr->seeds[r] < 0 ? -1 : pr->seeds[r]) : -1].push_back(r);
        uint32_t ord = 0;
        for (auto& kv : by_seed) { Item it; it.members = kv.second; it.level = 0; it.key = std::string(1, (char)('a' + std::min<uint32_t>(ord, 25))) + std::to_string(ord); ++ord; S[j].work.push_back(std::move(it)); }
    }
    auto rebased = [&](uint32_t j, const std::vector<uint32_t>& m, uint32_t level, std::vector<int32_t>& out) -> bool {
        const sp_priority_problem* pr = jobs[j].problem;
        const int32_t* src = pr->offsets ? pr->offsets[level] : nullptr;
        if (!src) return false;
        int64_t mn = INT64_MAX;
        for (uint32_t r : m) mn = std::min<int64_t>(mn, src[r] < 0 ? 0 : src[r]);
        out.resize(m.size());
        for (size_t i = 0; i < m.size(); ++i) { const int64_t o = src[m[i]] < 0 ? 0 : src[m[i]]; out[i] = o == mn ? -1 : (int32_t)(o - mn + (mn == 0 ? 0 : S[j].half)); }
        return true;
    };
    for (;;) {
        // one round: every open group of every job as one two-way problem, all in lockstep
        std::vector<std::pair<uint32_t, size_t>> at;                           // (job, index in its work list)
        for (uint32_t j = 0; j < n_jobs; ++j) if (S[j].live) for (size_t x = 0; x < S[j].work.size(); ++x) at.push_back({ j, x });
        const size_t k = at.size();
        if (k == 0) break;
        std::vector<sp_cons_problem> P(k); std::vector<sp_cons_output> O(k);
        std::vector<std::vector<int32_t>> offs(k), s1(k), s2(k); std::vector<std::vector<uint8_t>> is1(k); std::vector<std::vector<char>> text(k);
        for (size_t x = 0; x < k; ++x) {
            const uint32_t j = at[x].first; const Item& it = S[j].work[at[x].second];
            const sp_priority_problem* pr = jobs[j].problem;
            const sp_seqset* set = pr->levels[it.level];
            int32_t longest = 0; for (uint32_t r : it.members) longest = std::max(longest, set->h_len[r]);
            const bool has_off = rebased(j, it.members, it.level, offs[x]);
            int32_t far = 0; if (has_off) for (int32_t o : offs[x]) far = std::max(far, o);
            const uint32_t c = (uint32_t)longest + (uint32_t)far + 66;
            s1[x].resize(it.members.size()); s2[x].resize(it.members.size()); is1[x].resize(it.members.size()); text[x].assign((size_t)2 * c, 0);
            P[x].reads = set; P[x].read_idx = it.members.data(); P[x].n = (uint32_t)it.members.size(); P[x].offsets = has_off ? offs[x].data() : nullptr;
            P[x].cfg = pr->cfg; P[x].cfg.allow_dual = 1;
            if (it.retry > 0) P[x].cfg.min_af = retry_min_af[it.retry - 1];
            std::memset(&O[x], 0, sizeof O[x]);
            O[x].cons1 = text[x].data(); O[x].cons2 = text[x].data() + c; O[x].cap = c; O[x].is_cons1 = is1[x].data(); O[x].score1 = s1[x].data(); O[x].score2 = s2[x].data();
        }
        const int32_t rc = sp_consensus_dual_batch(ctx, (uint32_t)k, P.data(), O.data());
        if (rc != SP_OK) return rc;
        std::vector<std::vector<Item>> next(n_jobs);
        for (size_t x = 0; x < k; ++x) {
            const uint32_t j = at[x].first; Item& it = S[j].work[at[x].second];
            const sp_priority_problem* pr = jobs[j].problem;
            const uint32_t NL = pr->n_levels;
            // gave up: once more with the next stricter fraction of the ladder that is above the configured one
            if (O[x].result.gave_up && !pr->cfg.no_retry_ladder && !it.members.empty()) {
                int step = it.retry;
                while (step < 4 && retry_min_af[step] <= pr->cfg.min_af) ++step;
                if (step < 4) { it.retry = step + 1; next[j].push_back(std::move(it)); continue; }
            }
            if (O[x].result.gave_up) jobs[j].gave_up += 1;
            std::vector<uint32_t> g1, g2;
            for (size_t i = 0; i < it.members.size(); ++i) (is1[x][i] ? g1 : g2).push_back(it.members[i]);
            if (O[x].result.is_dual && !g1.empty() && !g2.empty()) {
                Item a; a.members = std::move(g1); a.level = it.level; a.key = it.key + "0"; next[j].push_back(std::move(a));
                Item b; b.members = std::move(g2); b.level = it.level; b.key = it.key + "1"; next[j].push_back(std::move(b));
            } else {
                if (it.retry == 0 && !O[x].result.is_dual && !O[x].result.gave_up) {
                    it.kept.resize(NL); it.have.resize(NL, 0);
                    it.kept[it.level] = std::string(text[x].data()); it.have[it.level] = 1;
                }
                if (it.level + 1 < NL) { it.level += 1; it.key += "_"; it.retry = 0; next[j].push_back(std::move(it)); }
                else S[j].done.push_back(std::move(it));
            }
        }
        for (uint32_t j = 0; j < n_jobs; ++j) if (S[j].live) {
            S[j].work.swap(next[j]);
            if (S[j].done.size() + S[j].work.size() > (size_t)jobs[j].problem->n) { jobs[j].status = SP_ERR_INVALID_ARG; S[j].live = false; S[j].work.clear(); }
        }
    }
    // one consensus per emitted group and level: the one its own two-way search left when that ended with one consensus, else a search of its own -- those of all
    // jobs in one batch
    std::vector<sp_cons_problem> P; std::vector<sp_cons_output> O;
    std::vector<std::vector<int32_t>> offs, s1, s2; std::vector<std::vector<uint8_t>> is1; std::vector<std::vector<char>> spare;
    size_t k = 0;
    auto has_kept = [](const Item& it, uint32_t l) { return l < it.have.size() && it.have[l]; };
    for (uint32_t j = 0; j < n_jobs; ++j) {
        if (!S[j].live) continue;
        sp_priority_job& J = jobs[j];
        auto& done = S[j].done;
        std::sort(done.begin(), done.end(), [](const Item& a, const Item& b) { return a.key < b.key; });
        *J.n_groups = (uint32_t)done.size();
        for (size_t g = 0; g < done.size(); ++g) for (uint32_t r : done[g].members) J.group_of[r] = (int32_t)g;
        if (done.size() > J.max_groups) { J.status = SP_ERR_CAPACITY; S[j].live = false; continue; }
        const uint32_t NL = J.problem->n_levels;
        for (size_t g = 0; g < done.size() && S[j].live; ++g) for (uint32_t l = 0; l < NL; ++l) {
            if (!has_kept(done[g], l)) { ++k; continue; }
            const std::string& c = done[g].kept[l];
            if (c.size() + 1 > (size_t)J.cap) { J.status = SP_ERR_CAPACITY; S[j].live = false; break; }
            std::memcpy(J.cons + (g * NL + l) * (size_t)J.cap, c.c_str(), c.size() + 1);
        }
    }
    k = 0;
    for (uint32_t j = 0; j < n_jobs; ++j) if (S[j].live) for (const Item& it : S[j].done) for (uint32_t l = 0; l < jobs[j].problem->n_levels; ++l) if (!has_kept(it, l)) ++k;
    P.resize(k); O.resize(k); offs.resize(k); s1.resize(k); s2.resize(k); is1.resize(k); spare.resize(k);
    size_t x = 0;
    for (uint32_t j = 0; j < n_jobs; ++j) {
        if (!S[j].live) continue;
        sp_priority_job& J = jobs[j]; const sp_priority_problem* pr = J.problem; const uint32_t NL = pr->n_levels;
        for (size_t g = 0; g < S[j].done.size(); ++g) for (uint32_t l = 0; l < NL; ++l) {
            const Item& it = S[j].done[g];
            if (has_kept(it, l)) continue;
            const bool has_off = rebased(j, it.members, l, offs[x]);
            s1[x].resize(it.members.size()); s2[x].resize(it.members.size()); is1[x].resize(it.members.size()); spare[x].assign(J.cap, 0);
            P[x].reads = pr->levels[l]; P[x].read_idx = it.members.data(); P[x].n = (uint32_t)it.members.size(); P[x].offsets = has_off ? offs[x].data() : nullptr;
            P[x].cfg = pr->cfg; P[x].cfg.allow_dual = 0;
            std::memset(&O[x], 0, sizeof O[x]);
            O[x].cons1 = J.cons + (g * NL + l) * (size_t)J.cap; O[x].cons2 = spare[x].data(); O[x].cap = J.cap; O[x].is_cons1 = is1[x].data(); O[x].score1 = s1[x].data(); O[x].score2 = s2[x].data();
            ++x;
        }
    }
    if (k == 0) return SP_OK;
    return sp_consensus_batch(ctx, (uint32_t)k, P.data(), O.data());
}

int32_t sp_consensus_priority(sp_ctx* ctx, const sp_priority_problem* pr, uint32_t max_groups, uint32_t cap,
                              uint32_t* n_groups, int32_t* group_of, char* cons) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!pr || !n_groups || !group_of || !cons || !pr->levels || pr->n_levels == 0 || cap < 2) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_priority: null argument");
    for (uint32_t l = 0; l < pr->n_levels; ++l) if (!pr->levels[l] || pr->levels[l]->n != pr->n) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_priority: every level needs one sequence per read");
    sp_priority_job J; J.problem = pr; J.max_groups = max_groups; J.cap = cap; J.n_groups = n_groups; J.group_of = group_of; J.cons = cons; J.status = SP_OK; J.gave_up = 0;
    const int32_t rc = sp_consensus_priority_many(ctx, 1, &J);
    if (rc != SP_OK) return rc;
    if (J.status == SP_ERR_CAPACITY) return sp_fail(ctx, SP_ERR_CAPACITY, "sp_consensus_priority: more groups than max_groups");
    if (J.status == SP_ERR_INVALID_ARG) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_priority: more groups than reads");
    return J.status;
}

} // extern "C"
