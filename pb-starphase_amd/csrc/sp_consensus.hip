// sp_consensus.hip -- K8: read consensus by dynamic wavefront alignment (gfx950).
//
// Serves the waffle_con call sites of the reference: DualConsensusDWFA in run_dual_consensus_with_offsets
// (src/hla/caller.rs:1103-1219) and the per-group ConsensusDWFA (src/hla/caller.rs:706-747), configured as
// dwfa_config_from_cli does (src/hla/caller.rs:1103-1116).  waffle_con itself (v0.4.4) is a third-party crate that is not
// on disk, so the contract is the one stated in DESIGN.md section 12 and restated for the CPU in oracle/consensus.c; the
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

namespace {

constexpr int CB = 64;          // band
constexpr int CH = 32;          // lane of diagonal 0
constexpr int CWAVES = 4;       // reads per workgroup
enum { F_ACTIVE = 1, F_FINISHED = 2, F_LOST = 4 };

struct ConsCtrl {               // state before a step
    int32_t dual, split_at, stopped[2], len[2], done, pad;
    long long best_w2, best_total;
};
struct ConsMeta { int32_t e, c0, flags, pad; };

struct ConsParams {
    SeqSetView reads; const uint32_t* idx; int n; const int32_t* offsets;
    int min_count, delta, et, allow_dual, window, cmp_len; double min_af;
    uint8_t* C; int cap;        // [2][cap] base codes; consensus 2 shares [0, split_at) with consensus 1
    uint32_t* votes;            // [2][cap+1][8] : w[4], end
    int32_t* H;                 // [2][n][64]
    ConsMeta* meta;             // [2][n]
    ConsCtrl* ctrl;             // [2]
};

struct ReadView { const uint32_t* w; const uint32_t* np; int n; };

__device__ __forceinline__ int read_base(const ReadView& rv, int h) {
    const uint32_t sh = (uint32_t)(h & 15) << 1;
    if (rv.np && ((rv.np[h >> 4] >> sh) & 1u)) return 4;
    return (int)((rv.w[h >> 4] >> sh) & 3u);
}

struct Decision { int go[2]; int base[2]; int split; long long best_w2, best_total; };

__device__ __forceinline__ Decision decide(const ConsParams& P, const ConsCtrl& c, int t) {
    Decision d; d.go[0] = d.go[1] = 0; d.base[0] = d.base[1] = 0; d.split = 0; d.best_w2 = c.best_w2; d.best_total = c.best_total;
    const int ncons = c.dual ? 2 : 1;
    for (int i = 0; i < ncons; ++i) {
        if (c.stopped[i]) continue;
        const uint32_t* v = P.votes + ((size_t)i * (P.cap + 1) + t) * 8;
        long long w[4] = { v[0], v[1], v[2], v[3] };
        const long long end = v[4], total = w[0] + w[1] + w[2] + w[3];
        int b1 = 0, b2 = -1;
        for (int b = 1; b < 4; ++b) if (w[b] > w[b1]) b1 = b;
        for (int b = 0; b < 4; ++b) if (b != b1 && (b2 < 0 || w[b] > w[b2])) b2 = b;
        const bool go = P.et ? w[b1] > 0 : (total > end && w[b1] > 0);
        if (!go) continue;
        d.go[i] = 1; d.base[i] = b1;
        if (!c.dual && w[b2] >= 12ll * P.min_count) {
            if (w[b2] * d.best_total > d.best_w2 * total) { d.best_w2 = w[b2]; d.best_total = total; }
            if (P.allow_dual && (double)w[b2] >= P.min_af * (double)total) { d.split = 1; d.go[1] = 1; d.base[1] = b2; }
        }
    }
    return d;
}

// consensus i at absolute position pos as seen by launch t (position t itself is this launch's decision, not yet in memory)
struct ConsView {
    const uint8_t* C; int cap, split_at, t; int base_t[2];
    __device__ __forceinline__ int at(int i, int pos) const {
        if (pos == t) return base_t[i];
        return (i == 1 && pos < split_at) ? C[pos] : C[(size_t)i * cap + pos];
    }
};

struct Dwfa { int H, e, c0, flags; };

// the consensus (cons i) now has T bases after c0; `nb` is its newest base
__device__ __forceinline__ void dwfa_push(Dwfa& d, const ReadView& rv, const ConsView& cv, int i, int T, int nb, int et, int lane) {
    const int k = lane - CH;
    if (d.H >= 0 && d.H + k == T - 1 && d.H < rv.n && read_base(rv, d.H) == nb) d.H += 1;      // only the old tips can move
    while (!__ballot(d.H >= 0 && d.H + k == T)) {
        const int c = d.H, up = spw::from_lower(d.H, SP_NEG), dn = spw::from_upper(d.H, SP_NEG);
        int best = SP_NEG;
        if (c >= 0 && c < rv.n && c + k < T) best = c + 1;
        if (up >= 0 && up + k <= T && up + k >= 0 && up > best) best = up;
        if (dn >= 0 && dn < rv.n && dn + 1 + k >= 0 && dn + 1 > best) best = dn + 1;
        if (!__ballot(best >= 0)) { d.flags |= F_LOST; return; }
        d.H = best; d.e += 1;
        for (;;) {
            bool go = d.H >= 0 && d.H < rv.n && d.H + k < T;
            if (go) { const int rb = read_base(rv, d.H); go = rb < 4 && rb == cv.at(i, d.c0 + d.H + k); }
            if (!__ballot(go)) break;
            if (go) d.H += 1;
        }
    }
    if (et && __ballot(d.H == rv.n)) d.flags |= F_FINISHED;
}

// placement of a late read: Sellers' search of its first L bases in the last W consensus bases, one Myers bit-vector scan per
// lane over the end positions it owns (an occurrence of an L-base pattern with <= L edits spans <= 2L text bases)
__device__ __forceinline__ int find_start(const ReadView& rv, const ConsView& cv, int i, int off, int W, int L, int lane) {
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
            const int x = cv.at(i, off - j);
            const unsigned long long Eq = peq[x & 3];
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

__global__ void __launch_bounds__(CWAVES * SP_WAVE) cons_step_kernel(ConsParams P, int t) {
    __shared__ uint32_t lv[2][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * CWAVES + wave;
    ConsCtrl cin;
    Decision dec;
    if (t >= 0) {
        cin = P.ctrl[t & 1];
        if (cin.done) { if (blockIdx.x == 0 && threadIdx.x == 0) P.ctrl[(t + 1) & 1] = cin; return; }
        dec = decide(P, cin, t);
    } else {
        cin = P.ctrl[0];
        dec.go[0] = dec.go[1] = 0; dec.base[0] = dec.base[1] = 0; dec.split = 0; dec.best_w2 = 0; dec.best_total = 1;
    }
    const int dual = (t >= 0) && (cin.dual || dec.split);
    const int split_at = dec.split ? t : cin.split_at;
    if (threadIdx.x < 16) lv[threadIdx.x >> 3][threadIdx.x & 7] = 0;
    if (t >= 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        ConsCtrl co = cin;
        for (int i = 0; i < 2; ++i) {
            if (dec.go[i]) { P.C[(size_t)i * P.cap + t] = (uint8_t)dec.base[i]; co.len[i] = t + 1; co.stopped[i] = 0; }
            else if (i == 0 || cin.dual) co.stopped[i] = 1;
        }
        co.dual = dual; co.split_at = split_at; co.best_w2 = dec.best_w2; co.best_total = dec.best_total;
        co.done = !(dec.go[0] || dec.go[1]);
        P.ctrl[(t + 1) & 1] = co;
    }
    __syncthreads();
    if (r < P.n) {
        const uint32_t rid = P.idx ? P.idx[r] : (uint32_t)r;
        ReadView rv; rv.w = P.reads.words + P.reads.word_off[rid]; rv.np = P.reads.nplane ? P.reads.nplane + P.reads.word_off[rid] : nullptr;
        rv.n = P.reads.len[rid];
        const int off = P.offsets ? P.offsets[r] : -1;
        ConsView cv; cv.C = P.C; cv.cap = P.cap; cv.split_at = split_at; cv.t = t; cv.base_t[0] = dec.base[0]; cv.base_t[1] = dec.base[1];
        Dwfa d[2];
        for (int i = 0; i < 2; ++i) { d[i].H = SP_NEG; d[i].e = 0; d[i].c0 = 0; d[i].flags = 0; }
        if (t < 0) {
            if (off < 0) { d[0].flags = F_ACTIVE | ((P.et && rv.n == 0) ? F_FINISHED : 0); d[0].H = lane == CH ? 0 : SP_NEG; }
        } else {
            const ConsMeta m0 = P.meta[r];
            d[0].H = P.H[(size_t)r * CB + lane]; d[0].e = m0.e; d[0].c0 = m0.c0; d[0].flags = m0.flags;
            if (dec.split) d[1] = d[0];
            else if (cin.dual) {
                const ConsMeta m1 = P.meta[(size_t)P.n + r];
                d[1].H = P.H[((size_t)P.n + r) * CB + lane]; d[1].e = m1.e; d[1].c0 = m1.c0; d[1].flags = m1.flags;
            }
            for (int i = 0; i < (dual ? 2 : 1); ++i) {
                if (!dec.go[i]) continue;
                const int len = t + 1;
                if (d[i].flags & F_ACTIVE) {
                    if (!(d[i].flags & (F_FINISHED | F_LOST))) dwfa_push(d[i], rv, cv, i, len - d[i].c0, dec.base[i], P.et, lane);
                } else if (off == len) {
                    d[i].c0 = find_start(rv, cv, i, off, P.window, P.cmp_len, lane);
                    d[i].H = lane == CH ? 0 : SP_NEG; d[i].e = 0; d[i].flags = F_ACTIVE | ((P.et && rv.n == 0) ? F_FINISHED : 0);
                    for (int T = 1; T <= len - d[i].c0; ++T) {
                        if (d[i].flags & (F_FINISHED | F_LOST)) break;
                        dwfa_push(d[i], rv, cv, i, T, cv.at(i, d[i].c0 + T - 1), P.et, lane);
                    }
                }
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
        for (int i = 0; i < (dual ? 2 : 1); ++i) {
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
                if (dc) { for (int b = 0; b < 4; ++b) if (seen[b]) atomicAdd(&lv[i][b], 12u / dc); }
                else if (ended) atomicAdd(&lv[i][4], 12u);
            }
        }
        // store
        for (int i = 0; i < (dual ? 2 : 1); ++i) {
            if (lane == 0) { ConsMeta m; m.e = d[i].e; m.c0 = d[i].c0; m.flags = d[i].flags; m.pad = 0; P.meta[(size_t)i * P.n + r] = m; }
            P.H[((size_t)i * P.n + r) * CB + lane] = d[i].H;
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int i = threadIdx.x >> 3, j = threadIdx.x & 7;
        const uint32_t v = lv[i][j];
        if (v) atomicAdd(P.votes + ((size_t)i * (P.cap + 1) + (t + 1)) * 8 + j, v);
    }
}

__global__ void __launch_bounds__(CWAVES * SP_WAVE) cons_finalize_kernel(ConsParams P, int which, uint8_t* is_cons1, int32_t* score1, int32_t* score2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * CWAVES + wave;
    if (r >= P.n) return;
    const ConsCtrl c = P.ctrl[which];
    const uint32_t rid = P.idx ? P.idx[r] : (uint32_t)r;
    const int n = P.reads.len[rid];
    int sc[2] = { -1, -1 };
    for (int i = 0; i < (c.dual ? 2 : 1); ++i) {
        const ConsMeta m = P.meta[(size_t)i * P.n + r];
        if (!(m.flags & F_ACTIVE) || (m.flags & F_LOST)) continue;
        int e = m.e;
        if (!P.et) {
            const int h = P.H[((size_t)i * P.n + r) * CB + lane], k = lane - CH;
            int rest = (h >= 0 && h + k == c.len[i] - m.c0) ? n - h : (1 << 30);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(rest, o); rest = other < rest ? other : rest; }
            if (rest < (1 << 30)) e += rest;
        }
        sc[i] = e;
    }
    if (lane == 0) {
        score1[r] = sc[0]; score2[r] = sc[1];
        is_cons1[r] = !(sc[1] >= 0 && (sc[0] < 0 || sc[1] < sc[0]));
    }
}

} // namespace

extern "C" {

int32_t sp_consensus(sp_ctx* ctx, const sp_seqset* reads, const uint32_t* read_idx, uint32_t n, const int32_t* offsets,
                     const sp_cons_config* cfg, char* cons1, char* cons2, uint32_t cap,
                     uint8_t* is_cons1, int32_t* score1, int32_t* score2, sp_cons_result* result) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!reads || !cfg || !cons1 || !cons2 || !is_cons1 || !score1 || !score2 || !result || cap == 0)
        return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: null argument");
    if (cfg->offset_compare_length > 64 || cfg->offset_compare_length < 0 || cfg->offset_window < 0 || cfg->min_count < 0)
        return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: offset_compare_length must be <= 64");
    if (cap >= (1u << 22)) return sp_fail(ctx, SP_ERR_TOO_LONG, "sp_consensus: cap must be below 4,194,304");
    if (!read_idx) n = reads->n;
    std::memset(result, 0, sizeof *result);
    result->split_at = -1; result->best_total = 1;
    cons1[0] = cons2[0] = '\0';
    if (n == 0) return SP_OK;
    if (read_idx) for (uint32_t i = 0; i < n; ++i) if (read_idx[i] >= reads->n) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus: read index out of range");
    SP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    ConsParams P;
    P.reads = reads->view(); P.n = (int)n; P.cap = (int)cap;
    P.min_count = cfg->min_count; P.delta = cfg->dual_max_ed_delta; P.et = cfg->allow_early_termination != 0; P.allow_dual = cfg->allow_dual != 0;
    P.window = cfg->offset_window; P.cmp_len = cfg->offset_compare_length; P.min_af = cfg->min_af;
    uint32_t* d_idx = nullptr; int32_t* d_off = nullptr;
    if (read_idx) { d_idx = (uint32_t*)sp_pool(ctx, "cons_idx", sizeof(uint32_t) * n); if (!d_idx) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "cons_idx"); }
    if (offsets)  { d_off = (int32_t*)sp_pool(ctx, "cons_off", sizeof(int32_t) * n); if (!d_off) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "cons_off"); }
    P.idx = d_idx; P.offsets = d_off;
    const size_t votes_bytes = sizeof(uint32_t) * 8 * 2 * ((size_t)cap + 1);
    P.C = (uint8_t*)sp_pool(ctx, "cons_C", 2 * (size_t)cap);
    P.votes = (uint32_t*)sp_pool(ctx, "cons_votes", votes_bytes);
    P.H = (int32_t*)sp_pool(ctx, "cons_H", sizeof(int32_t) * 2 * (size_t)n * CB);
    P.meta = (ConsMeta*)sp_pool(ctx, "cons_meta", sizeof(ConsMeta) * 2 * (size_t)n);
    P.ctrl = (ConsCtrl*)sp_pool(ctx, "cons_ctrl", sizeof(ConsCtrl) * 2);
    uint8_t* d_is1 = (uint8_t*)sp_pool(ctx, "cons_is1", n);
    int32_t* d_sc = (int32_t*)sp_pool(ctx, "cons_scores", sizeof(int32_t) * 2 * (size_t)n);
    if (!P.C || !P.votes || !P.H || !P.meta || !P.ctrl || !d_is1 || !d_sc) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_consensus buffers");
    if (read_idx) SP_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, read_idx, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
    if (offsets)  SP_HIP_CHECK(ctx, hipMemcpyAsync(d_off, offsets, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemsetAsync(P.votes, 0, votes_bytes, st));
    SP_HIP_CHECK(ctx, hipMemsetAsync(P.meta, 0, sizeof(ConsMeta) * 2 * (size_t)n, st));
    ConsCtrl c0; std::memset(&c0, 0, sizeof c0); c0.split_at = -1; c0.stopped[1] = 1; c0.best_total = 1;
    ConsCtrl init[2] = { c0, c0 };
    SP_HIP_CHECK(ctx, hipMemcpyAsync(P.ctrl, init, sizeof init, hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(st));      // the pageable host sources above must stay valid until copied

    const dim3 grid((n + CWAVES - 1) / CWAVES), block(CWAVES * SP_WAVE);
    ConsCtrl cur = c0;
    int last = -1;
    {
        ProfScope ps(ctx, "cons_steps", n);
        hipLaunchKernelGGL(cons_step_kernel, grid, block, 0, st, P, -1);
        for (int t = 0; t < (int)cap; ++t) {
            hipLaunchKernelGGL(cons_step_kernel, grid, block, 0, st, P, t);
            last = t;
            if ((t & 255) == 255 || t + 1 == (int)cap) {
                SP_HIP_CHECK(ctx, hipMemcpyAsync(&cur, P.ctrl + ((t + 1) & 1), sizeof cur, hipMemcpyDeviceToHost, st));
                SP_HIP_CHECK(ctx, hipStreamSynchronize(st));
                if (cur.done) break;
            }
        }
    }
    SP_HIP_CHECK(ctx, hipGetLastError());
    const int which = (last + 1) & 1;
    hipLaunchKernelGGL(cons_finalize_kernel, grid, block, 0, st, P, which, d_is1, d_sc, d_sc + n);
    std::vector<uint8_t> hc(2 * (size_t)cap);
    SP_HIP_CHECK(ctx, hipMemcpyAsync(hc.data(), P.C, hc.size(), hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(is_cons1, d_is1, n, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(score1, d_sc, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(score2, d_sc + n, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(&cur, P.ctrl + which, sizeof cur, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(st));
    SP_HIP_CHECK(ctx, hipGetLastError());
    // a consensus that is still growing when cap is reached is truncated there (len == cap): the caller sized cap too small
    static const char dec[4] = { 'A', 'C', 'G', 'T' };
    const int len1 = cur.len[0], len2 = cur.dual ? cur.len[1] : 0;
    if ((uint32_t)len1 >= cap || (uint32_t)len2 >= cap) return sp_fail(ctx, SP_ERR_CAPACITY, "sp_consensus: consensus reached cap");
    for (int p = 0; p < len1; ++p) cons1[p] = dec[hc[p] & 3];
    cons1[len1] = '\0';
    for (int p = 0; p < len2; ++p) cons2[p] = dec[(p < cur.split_at ? hc[p] : hc[(size_t)cap + p]) & 3];
    cons2[len2] = '\0';
    result->is_dual = cur.dual; result->len1 = len1; result->len2 = len2; result->split_at = cur.split_at;
    result->best_w2 = cur.best_w2; result->best_total = cur.best_total;
    return SP_OK;
}

int32_t sp_consensus_dual(sp_ctx* ctx, const sp_seqset* reads, const uint32_t* read_idx, uint32_t n, const int32_t* offsets,
                          const sp_cons_config* cfg, char* cons1, char* cons2, uint32_t cap,
                          uint8_t* is_cons1, int32_t* score1, int32_t* score2, sp_cons_result* result) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!cfg) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_consensus_dual: null config");
    sp_cons_config pass = *cfg;
    pass.allow_dual = 0;
    int32_t rc = sp_consensus(ctx, reads, read_idx, n, offsets, &pass, cons1, cons2, cap, is_cons1, score1, score2, result);
    if (rc != SP_OK || result->best_w2 == 0) return rc;
    pass.allow_dual = 1;
    const double strongest = 0.5 * (double)result->best_w2 / (double)result->best_total;
    pass.min_af = cfg->min_af > strongest ? cfg->min_af : strongest;
    return sp_consensus(ctx, reads, read_idx, n, offsets, &pass, cons1, cons2, cap, is_cons1, score1, score2, result);
}

} // extern "C"
