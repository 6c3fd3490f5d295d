/*
 * consensus.c -- CPU ORACLE (test infrastructure only): read consensus by dynamic wavefront alignment.
 *
 * The reference delegates consensus to the third-party crate waffle_con v0.4.4 (Cargo.lock:2246-2248), whose sources are
 * not under /root/reference; no reference test runs a consensus.  PARITY UNPINNED: this file states the
 * contract the HIP path implements, built on waffle_con's published idea -- every read keeps an
 * edit-distance wavefront against the growing consensus, reads vote for the next base, a second consensus is split off
 * when a second base has enough support -- with every rule made explicit and deterministic:
 *   call sites it serves      src/hla/caller.rs:1103-1219 (dual, HPC then DNA, offsets, early termination)
 *                             src/hla/caller.rs:706-747   (one consensus per read group)
 *   configuration mirrored    dwfa_config_from_cli, src/hla/caller.rs:1103-1116 (min_count, min_af, dual_max_ed_delta,
 *                             allow_early_termination, offset_window 400, offset_compare_length 50)
 *
 * Per read and consensus: band of 64 diagonals (lane l <-> k = l - 32, k = consensus position - read position, both counted
 * from the read's start on the consensus), H[l] = furthest read position with e edits.  When the consensus grows by one base
 * every lane extends; if no lane consumes the whole consensus, e increases by one (next wavefront, priority X > D > I as in
 * align.c).  Lanes that have consumed the whole consensus ("tips") name the read's next base: the read gives 12/d vote
 * units to each of its d distinct tip bases.  The next base is the heaviest one (ties: A < C < G < T).
 */
#include "sp_oracle.h"
#include "consensus_oracle.h"
#include <stdlib.h>
#include <string.h>

#define NEG OSP_NEG
#define BAND 64
#define HALF 32

typedef struct {
    int active, finished, tracked;
    int e, c0;
    int H[BAND];
} dwfa;

static void dwfa_reset(dwfa* d) { memset(d, 0, sizeof *d); d->tracked = 1; for (int l = 0; l < BAND; ++l) d->H[l] = NEG; }

/* extend every lane against consensus C[c0 .. c0+T) */
static void dwfa_extend(dwfa* d, const uint8_t* S, int n, const uint8_t* C, int T) {
    for (int l = 0; l < BAND; ++l) {
        int h = d->H[l]; if (h < 0) continue;
        const int k = l - HALF;
        while (h < n && h + k < T && S[h] < 4 && S[h] == C[d->c0 + h + k]) ++h;
        d->H[l] = h;
    }
}
static int dwfa_has_tip(const dwfa* d, int T) {
    for (int l = 0; l < BAND; ++l) if (d->H[l] >= 0 && d->H[l] + (l - HALF) == T) return 1;
    return 0;
}
static int dwfa_read_done(const dwfa* d, int n) {
    for (int l = 0; l < BAND; ++l) if (d->H[l] == n) return 1;
    return 0;
}
/* the consensus now has T bases after c0 */
static void dwfa_push(dwfa* d, const uint8_t* S, int n, const uint8_t* C, int T, int early_termination) {
    dwfa_extend(d, S, n, C, T);
    while (!dwfa_has_tip(d, T)) {
        int nx[BAND];
        for (int l = 0; l < BAND; ++l) {
            const int k = l - HALF;
            int best = NEG;
            const int c = d->H[l], up = l > 0 ? d->H[l - 1] : NEG, dn = l < BAND - 1 ? d->H[l + 1] : NEG;
            if (c >= 0 && c < n && c + k < T) best = c + 1;                                   /* X */
            if (up >= 0 && up + k <= T && up + k >= 0 && up > best) best = up;                /* D: consensus base only (from k-1) */
            if (dn >= 0 && dn < n && dn + 1 + k >= 0 && dn + 1 > best) best = dn + 1;         /* I: read base only (from k+1) */
            nx[l] = best;
        }
        int alive = 0;
        for (int l = 0; l < BAND; ++l) alive |= nx[l] >= 0;
        if (!alive) { d->tracked = 0; return; }                                               /* the band lost the read: its score is None */
        memcpy(d->H, nx, sizeof nx);
        d->e += 1;
        dwfa_extend(d, S, n, C, T);
    }
    if (early_termination && dwfa_read_done(d, n)) d->finished = 1;
}

/* start of a late read on the consensus: Sellers' search of the read's first L bases in the last W consensus bases */
static int find_start(const uint8_t* S, int n, const uint8_t* C, int off, int W, int L) {
    const int ws = off - W > 0 ? off - W : 0, M = off - ws;
    if (L > n) L = n;
    if (M <= 0 || L <= 0) return off;
    int* prev = (int*)malloc(sizeof(int) * (size_t)(M + 1)), *cur = (int*)malloc(sizeof(int) * (size_t)(M + 1));
    for (int j = 0; j <= M; ++j) prev[j] = 0;                                                 /* free start in the (reversed) text */
    for (int i = 1; i <= L; ++i) {
        const uint8_t p = S[L - i];
        cur[0] = i;
        for (int j = 1; j <= M; ++j) {
            const uint8_t x = C[off - j];
            int v = prev[j - 1] + ((p < 4 && p == x) ? 0 : 1);
            if (prev[j] + 1 < v) v = prev[j] + 1;
            if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
            cur[j] = v;
        }
        int* t = prev; prev = cur; cur = t;
    }
    const int centre = off - W / 2;
    int best_p = off, best_d = 1 << 30, best_c = 1 << 30;
    for (int j = 1; j <= M; ++j) {
        const int p = off - j, dist = p > centre ? p - centre : centre - p;
        if (prev[j] < best_d || (prev[j] == best_d && (dist < best_c || (dist == best_c && p < best_p)))) { best_d = prev[j]; best_c = dist; best_p = p; }
    }
    free(prev); free(cur);
    return best_p;
}

static void activate(dwfa* d, const uint8_t* S, int n, const uint8_t* C, int len, int off, const osp_cons_config* cfg) {
    dwfa_reset(d);
    d->active = 1;
    d->c0 = off < 0 ? 0 : find_start(S, n, C, off, cfg->offset_window, cfg->offset_compare_length);
    d->H[HALF] = 0;
    for (int T = 0; T <= len - d->c0; ++T) {
        if (T == 0) { if (n == 0 && cfg->allow_early_termination) d->finished = 1; continue; }
        if (d->finished || !d->tracked) break;
        dwfa_push(d, S, n, C, T, cfg->allow_early_termination);
    }
}

typedef struct { int64_t w[4], end, total; } votes;

static void add_votes(votes* v, const dwfa* d, const uint8_t* S, int n, int T) {
    int seen[5] = {0, 0, 0, 0, 0}, any_tip = 0;
    for (int l = 0; l < BAND; ++l) {
        const int h = d->H[l];
        if (h < 0 || h + (l - HALF) != T) continue;
        any_tip = 1;
        if (h < n) seen[S[h] < 4 ? S[h] : 4] = 1;
    }
    const int dcount = seen[0] + seen[1] + seen[2] + seen[3];
    if (dcount) { for (int b = 0; b < 4; ++b) if (seen[b]) { v->w[b] += 12 / dcount; v->total += 12 / dcount; } }
    else if (any_tip && !seen[4]) v->end += 12;                                                /* every tip is at the end of the read */
}

int osp_consensus(int n_reads, const uint8_t* const* seqs, const int32_t* lens, const int32_t* offsets, const osp_cons_config* cfg,
                  uint8_t* cons1, uint8_t* cons2, int cap, uint8_t* is_cons1, int32_t* score1, int32_t* score2, osp_cons_result* res) {
    dwfa* st[2];
    st[0] = (dwfa*)malloc(sizeof(dwfa) * (size_t)(n_reads + 1)); st[1] = (dwfa*)malloc(sizeof(dwfa) * (size_t)(n_reads + 1));
    uint8_t* C[2] = { cons1, cons2 };
    int stopped[2] = { 0, 1 }, len[2] = { 0, 0 }, dual = 0, split_at = -1;
    int64_t best_w2 = 0, best_total = 1;
    for (int r = 0; r < n_reads; ++r) { dwfa_reset(&st[0][r]); dwfa_reset(&st[1][r]); if (offsets[r] < 0) activate(&st[0][r], seqs[r], lens[r], C[0], 0, -1, cfg); }
    for (int t = 0; t < cap; ++t) {
        int appended[2] = { 0, 0 };
        const int ncons = dual ? 2 : 1;
        int do_split = 0; uint8_t split_base = 0;
        for (int i = 0; i < ncons; ++i) {
            if (stopped[i]) continue;
            votes v; memset(&v, 0, sizeof v);
            for (int r = 0; r < n_reads; ++r) {
                const dwfa* d = &st[i][r];
                if (!d->active || d->finished || !d->tracked) continue;
                if (dual) { const dwfa* o = &st[1 - i][r]; if (o->active && o->tracked && o->e < d->e) continue; }   /* the read follows its better consensus */
                add_votes(&v, d, seqs[r], lens[r], t - d->c0);
            }
            int b1 = 0, b2 = -1;
            for (int b = 1; b < 4; ++b) if (v.w[b] > v.w[b1]) b1 = b;
            for (int b = 0; b < 4; ++b) if (b != b1 && (b2 < 0 || v.w[b] > v.w[b2])) b2 = b;
            /* with early termination the consensus follows the reads that are left; otherwise it ends where most reads end */
            const int go = cfg->allow_early_termination ? v.w[b1] > 0 : (v.total > v.end && v.w[b1] > 0);
            if (!go) { stopped[i] = 1; continue; }
            C[i][t] = (uint8_t)b1; appended[i] = 1;
            if (!dual && v.w[b2] >= 12 * (int64_t)cfg->min_count && v.w[b2] * best_total > best_w2 * v.total) { best_w2 = v.w[b2]; best_total = v.total; }
            if (!dual && cfg->allow_dual && v.w[b2] >= 12 * (int64_t)cfg->min_count && (double)v.w[b2] >= cfg->min_af * (double)v.total) { do_split = 1; split_base = (uint8_t)b2; }
        }
        if (do_split) {
            dual = 1; split_at = t; stopped[1] = 0; appended[1] = 1;
            memcpy(C[1], C[0], (size_t)t); C[1][t] = split_base;
            memcpy(st[1], st[0], sizeof(dwfa) * (size_t)n_reads);
        }
        if (!appended[0] && !appended[1]) break;
        for (int i = 0; i < 2; ++i) {
            if (!appended[i]) continue;
            len[i] = t + 1;
            for (int r = 0; r < n_reads; ++r) {
                dwfa* d = &st[i][r];
                if (d->active) { if (!d->finished && d->tracked) dwfa_push(d, seqs[r], lens[r], C[i], len[i] - d->c0, cfg->allow_early_termination); }
                else if (offsets[r] == len[i]) activate(d, seqs[r], lens[r], C[i], len[i], offsets[r], cfg);
            }
        }
        if (dual) for (int r = 0; r < n_reads; ++r) {
            dwfa* a = &st[0][r], *b = &st[1][r];
            if (!(a->active && b->active && a->tracked && b->tracked)) continue;
            if (a->e > b->e + cfg->dual_max_ed_delta) a->tracked = 0;
            else if (b->e > a->e + cfg->dual_max_ed_delta) b->tracked = 0;
        }
    }
    /* scores */
    for (int r = 0; r < n_reads; ++r) {
        int sc[2] = { -1, -1 };
        for (int i = 0; i < (dual ? 2 : 1); ++i) {
            const dwfa* d = &st[i][r];
            if (!d->active || !d->tracked) continue;
            int e = d->e;
            if (!cfg->allow_early_termination) {                                                /* the rest of the read is unmatched */
                int rest = 1 << 30;
                for (int l = 0; l < BAND; ++l) if (d->H[l] >= 0 && d->H[l] + (l - HALF) == len[i] - d->c0 && lens[r] - d->H[l] < rest) rest = lens[r] - d->H[l];
                if (rest < (1 << 30)) e += rest;
            }
            sc[i] = e;
        }
        score1[r] = sc[0]; score2[r] = sc[1];
        is_cons1[r] = !(sc[1] >= 0 && (sc[0] < 0 || sc[1] < sc[0]));
    }
    res->is_dual = dual; res->len1 = len[0]; res->len2 = dual ? len[1] : 0; res->split_at = split_at;
    res->best_w2 = best_w2; res->best_total = best_total;
    free(st[0]); free(st[1]);
    return 0;
}
