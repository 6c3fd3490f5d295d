/*
 * consensus.c -- CPU ORACLE (test infrastructure only): read consensus by dynamic wavefront alignment.
 *
 * The reference delegates consensus to the third-party crate waffle_con v0.4.4 (Cargo.lock:2246-2248), whose sources are
 * not under /root/reference; no reference test runs a consensus.  PARITY UNPINNED: this file states the
 * contract the HIP path implements.  It follows waffle_con's published algorithm as far as it can be
 * restated without the sources: every read keeps an edit-distance wavefront against the growing consensus and votes for the
 * next base; candidate extensions are explored BEST FIRST -- lowest total edit distance (ConsensusCost::L1Distance), then the
 * longest consensus -- under the bounds dwfa_config_from_cli sets (src/hla/caller.rs:1103-1116: max_queue_size 20,
 * max_capacity_per_size 10; max_nodes_wo_constraint keeps waffle_con's default of 1,000); a dual node keeps two consensuses and
 * every read counts with the one it is closer to.  Every rule is made explicit and deterministic here:
 *   call sites it serves      src/hla/caller.rs:1103-1219 (dual, HPC then DNA, offsets, early termination)
 *                             src/hla/caller.rs:706-747   (one consensus per read group)
 *                             src/cyp2d6/caller.rs:145-270 (through the multi-way driver)
 *
 * Per read and consensus: band of 64 diagonals (lane l <-> k = l - 32, k = consensus position - read position, both counted
 * from the read's start on the consensus), H[l] = furthest read position with e edits.  When the consensus grows by one base
 * every lane extends; if no lane consumes the whole consensus, e increases by one (next wavefront, priority X > D > I as in
 * align.c).  Lanes that have consumed the whole consensus ("tips") name the read's next base: the read gives 12/d vote
 * units to each of its d distinct tip bases.
 *
 * The search.  A node = the consensus(es) so far + the state of every read.  cost(node) = sum over reads of the smaller edit count
 * of its placed states (a state that is no longer tracked keeps its last count: costs never decrease).  Nodes wait in a queue ordered
 * by (cost, longer first, older first).  The best node is taken out and, unless the bounds drop it, expanded by one column:
 *   candidates of a consensus: its heaviest base always; every other base with >= min(min_count, heaviest) reads AND >= min_af of
 *   the votes of that column.  No votes at all (with early termination) / the end votes outweigh the base votes (without): the
 *   consensus stops there.
 *   single node: one child per candidate; when a second consensus is allowed, one dual child per pair of candidates (the heavier
 *   base continues consensus 1, the other starts consensus 2 as a copy).  dual node: one child per combination of the two
 *   consensuses' candidates (a stopped consensus stays stopped).  A node whose consensuses have all stopped is complete; the
 *   complete node of lowest final cost wins, the first one found on ties ("Found multiple solutions, selecting first").
 *   bounds: a node is dropped when it is shorter than the threshold or when max_capacity_per_size nodes of its length have been
 *   expanded already; while more than max_queue_size nodes wait, the one the search would take last is dropped; the threshold jumps
 *   to the longest length reached after every max_nodes_wo_constraint expansions.
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

/* Start of a late read on the consensus (add_sequence_offset).  The reference's callers say what the crate searches: start positions in the
 * offset_window bases BEFORE the offset ("the config only lets us look before; so we have to shift things", src/cyp2d6/caller.rs:145-146), comparing
 * the read's first offset_compare_length bases (100 for CYP2D6 in a window of 2 x 50, :144-159; 50 in 400 for HLA, src/hla/caller.rs:1113-1114).  A
 * 100-base comparison from a start 0..100 bases before the offset needs the consensus up to offset + 100, so a read is placed when the consensus
 * reaches length off + L (L = min(offset_compare_length, read length)): every candidate start p in [off - W, off] then has at least L consensus
 * bases behind it.  ed(p) = the fewest edits between the read's first L bases and ANY prefix of C[p .. len) (Sellers: the end on the consensus is
 * free); the smallest ed wins, ties go to the start closest to the middle of the window (off - W / 2: the expected start), then to the leftmost.
 * The DP runs backwards (pattern and text reversed, free start in the reversed text): its last row holds ed(p) for p = len - j. */
static int find_start(const uint8_t* S, int n, const uint8_t* C, int len, int off, int W, int L) {
    const int ws = off - W > 0 ? off - W : 0, M = len - ws;
    if (L > n) L = n;
    if (M <= 0 || L <= 0 || off <= ws) return off < len ? off : len;
    int* prev = (int*)malloc(sizeof(int) * (size_t)(M + 1)), *cur = (int*)malloc(sizeof(int) * (size_t)(M + 1));
    for (int j = 0; j <= M; ++j) prev[j] = 0;                                                 /* free start in the (reversed) text */
    for (int i = 1; i <= L; ++i) {
        const uint8_t p = S[L - i];
        cur[0] = i;
        for (int j = 1; j <= M; ++j) {
            const uint8_t x = C[len - j];
            int v = prev[j - 1] + ((p < 4 && p == x) ? 0 : 1);
            if (prev[j] + 1 < v) v = prev[j] + 1;
            if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
            cur[j] = v;
        }
        int* t = prev; prev = cur; cur = t;
    }
    const int centre = off - W / 2;
    int best_p = off, best_d = 1 << 30, best_c = 1 << 30;
    for (int j = len - off; j <= M; ++j) {                                                    /* starts off, off - 1, ..., ws */
        if (j < 1) continue;
        const int p = len - j, dist = p > centre ? p - centre : centre - p;
        if (prev[j] < best_d || (prev[j] == best_d && (dist < best_c || (dist == best_c && p < best_p)))) { best_d = prev[j]; best_c = dist; best_p = p; }
    }
    free(prev); free(cur);
    return best_p;
}

/* the consensus length at which read r is placed: its offset + the bases the placement compares */
static int activation_length(int off, int n, const osp_cons_config* cfg) {
    const int L = cfg->offset_compare_length < n ? cfg->offset_compare_length : n;
    return off + (L > 0 ? L : 0);
}

static void activate(dwfa* d, const uint8_t* S, int n, const uint8_t* C, int len, int off, const osp_cons_config* cfg) {
    dwfa_reset(d);
    d->active = 1;
    d->c0 = off < 0 ? 0 : find_start(S, n, C, len, off, cfg->offset_window, cfg->offset_compare_length);
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


/* ---------------------------------------------------------------- the search */
typedef struct {
    dwfa* st[2];
    uint8_t* C[2];
    int len[2], stopped[2];
    int dual, split_at, t;          /* t = columns decided so far */
    int64_t cost;
    int id;
} node;

typedef struct { int n_reads; const uint8_t* const* seqs; const int32_t* lens; const int32_t* offsets; const osp_cons_config* cfg; int cap; } problem;

static node* node_new(const problem* P) {
    node* x = (node*)calloc(1, sizeof(node));
    for (int i = 0; i < 2; ++i) {
        x->st[i] = (dwfa*)malloc(sizeof(dwfa) * (size_t)(P->n_reads + 1));
        x->C[i] = (uint8_t*)malloc((size_t)P->cap + 1);
    }
    return x;
}
static void node_free(node* x) { if (!x) return; for (int i = 0; i < 2; ++i) { free(x->st[i]); free(x->C[i]); } free(x); }
static node* node_clone(const problem* P, const node* s) {
    node* x = node_new(P);
    dwfa* st[2] = { x->st[0], x->st[1] }; uint8_t* C[2] = { x->C[0], x->C[1] };
    *x = *s; x->st[0] = st[0]; x->st[1] = st[1]; x->C[0] = C[0]; x->C[1] = C[1];
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && !s->dual) continue;
        memcpy(x->st[i], s->st[i], sizeof(dwfa) * (size_t)P->n_reads);
        memcpy(x->C[i], s->C[i], (size_t)s->len[i]);
    }
    return x;
}

/* the smaller edit count of the placed states of every read (a lost state keeps the count it had) */
static int64_t node_cost(const problem* P, const node* x) {
    int64_t c = 0;
    for (int r = 0; r < P->n_reads; ++r) {
        int best = -1;
        for (int i = 0; i < (x->dual ? 2 : 1); ++i) { const dwfa* d = &x->st[i][r]; if (d->active && (best < 0 || d->e < best)) best = d->e; }
        if (best > 0) c += best;
    }
    return c;
}

/* votes of consensus i of node x for its next column */
static void node_votes(const problem* P, const node* x, int i, votes* v) {
    memset(v, 0, sizeof *v);
    for (int r = 0; r < P->n_reads; ++r) {
        const dwfa* d = &x->st[i][r];
        if (!d->active || d->finished || !d->tracked) continue;
        if (x->dual) { const dwfa* o = &x->st[1 - i][r]; if (o->active && o->tracked && o->e < d->e) continue; }   /* the read follows its better consensus */
        add_votes(v, d, P->seqs[r], P->lens[r], x->t - d->c0);
    }
}

/* candidates of a column, heaviest first (ties to the lower code); 0 = the consensus stops */
static int candidates(const problem* P, const votes* v, int t, int out[4]) {
    if (t >= P->cap) return 0;                                  /* out of room: the consensus is cut at cap */
    int order[4] = { 0, 1, 2, 3 };
    for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) if (v->w[order[b]] > v->w[order[a]]) { int tmp = order[a]; order[a] = order[b]; order[b] = tmp; }
    const int64_t w1 = v->w[order[0]];
    const int go = P->cfg->allow_early_termination ? w1 > 0 : (v->total > v->end && w1 > 0);
    if (!go) return 0;
    int64_t need = 12 * (int64_t)P->cfg->min_count; if (w1 < need) need = w1;
    int n = 0;
    out[n++] = order[0];
    for (int a = 1; a < 4; ++a) { const int64_t w = v->w[order[a]]; if (w > 0 && w >= need && (double)w >= P->cfg->min_af * (double)v->total) out[n++] = order[a]; }
    return n;
}

/* one column: consensus i of x grows by base b[i] (b[i] < 0: it does not), late reads are placed, the two states of a read are
 * compared (dual_max_ed_delta) */
static void node_push(const problem* P, node* x, const int b[2]) {
    const osp_cons_config* cfg = P->cfg;
    for (int i = 0; i < (x->dual ? 2 : 1); ++i) {
        if (b[i] < 0) continue;
        x->C[i][x->t] = (uint8_t)b[i]; x->len[i] = x->t + 1;
        for (int r = 0; r < P->n_reads; ++r) {
            dwfa* d = &x->st[i][r];
            if (d->active) { if (!d->finished && d->tracked) dwfa_push(d, P->seqs[r], P->lens[r], x->C[i], x->len[i] - d->c0, cfg->allow_early_termination); }
            else if (P->offsets[r] >= 0 && activation_length(P->offsets[r], P->lens[r], cfg) == x->len[i]) activate(d, P->seqs[r], P->lens[r], x->C[i], x->len[i], P->offsets[r], cfg);
        }
    }
    if (x->dual) for (int r = 0; r < P->n_reads; ++r) {
        dwfa* a = &x->st[0][r], *c = &x->st[1][r];
        if (!(a->active && c->active && a->tracked && c->tracked)) continue;
        if (a->e > c->e + cfg->dual_max_ed_delta) a->tracked = 0;
        else if (c->e > a->e + cfg->dual_max_ed_delta) c->tracked = 0;
    }
    x->t += 1;
    x->cost = node_cost(P, x);
}

/* what is left of the reads when a consensus without early termination ends: added to the scores and to the final cost */
static int read_score(const problem* P, const node* x, int i, int r) {
    const dwfa* d = &x->st[i][r];
    if (!d->active || !d->tracked) return -1;
    int e = d->e;
    if (!P->cfg->allow_early_termination) {
        int rest = 1 << 30;
        for (int l = 0; l < BAND; ++l) if (d->H[l] >= 0 && d->H[l] + (l - HALF) == x->len[i] - d->c0 && P->lens[r] - d->H[l] < rest) rest = P->lens[r] - d->H[l];
        if (rest < (1 << 30)) e += rest;
    }
    return e;
}
static int64_t final_cost(const problem* P, const node* x) {
    if (P->cfg->allow_early_termination) return x->cost;
    int64_t c = 0;
    for (int r = 0; r < P->n_reads; ++r) {
        int best = -1;
        for (int i = 0; i < (x->dual ? 2 : 1); ++i) {
            const dwfa* d = &x->st[i][r];
            if (!d->active) continue;
            int s = d->tracked ? read_score(P, x, i, r) : d->e;
            if (best < 0 || s < best) best = s;
        }
        if (best > 0) c += best;
    }
    return c;
}

#define QCAP 64
int osp_consensus(int n_reads, const uint8_t* const* seqs, const int32_t* lens, const int32_t* offsets, const osp_cons_config* cfg,
                  uint8_t* cons1, uint8_t* cons2, int cap, uint8_t* is_cons1, int32_t* score1, int32_t* score2, osp_cons_result* res) {
    problem P = { n_reads, seqs, lens, offsets, cfg, cap };
    const int max_queue = cfg->max_queue_size > 0 ? cfg->max_queue_size : 20;
    const int per_size = cfg->max_capacity_per_size > 0 ? cfg->max_capacity_per_size : 10;
    const int wo_constraint = cfg->max_nodes_wo_constraint > 0 ? cfg->max_nodes_wo_constraint : 1000;
    node* queue[QCAP]; int nq = 0, next_id = 0;
    int* processed = (int*)calloc((size_t)cap + 2, sizeof(int));
    int threshold = 0, farthest = 0; int64_t pops = 0;
    node* best = NULL; int64_t best_final = 0;
    {
        node* root = node_new(&P);
        root->split_at = -1; root->stopped[1] = 1; root->id = next_id++;
        for (int r = 0; r < n_reads; ++r) { dwfa_reset(&root->st[0][r]); dwfa_reset(&root->st[1][r]); if (offsets[r] < 0) activate(&root->st[0][r], seqs[r], lens[r], root->C[0], 0, -1, cfg); }
        root->cost = 0;
        queue[nq++] = root;
    }
    while (nq > 0) {
        int bi = 0;
        for (int q = 1; q < nq; ++q) {
            const node* a = queue[q], *b = queue[bi];
            if (a->cost < b->cost || (a->cost == b->cost && (a->t > b->t || (a->t == b->t && a->id < b->id)))) bi = q;
        }
        node* x = queue[bi];
        if (best && x->cost >= best_final) break;                 /* nothing that waits can beat (or precede) the complete node */
        queue[bi] = queue[--nq];
        if (x->t < threshold || processed[x->t] >= per_size) { node_free(x); continue; }
        processed[x->t] += 1; ++pops;
        if (x->t > farthest) farthest = x->t;
        if (pops % wo_constraint == 0 && farthest > threshold) threshold = farthest;
        /* candidates of every consensus that is still going */
        int nc[2] = { 0, 0 }, cand[2][4];
        for (int i = 0; i < (x->dual ? 2 : 1); ++i) {
            if (x->stopped[i]) continue;
            votes v; node_votes(&P, x, i, &v);
            nc[i] = candidates(&P, &v, x->t, cand[i]);
            if (nc[i] == 0) x->stopped[i] = 1;
        }
        if (nc[0] == 0 && nc[1] == 0) {                           /* complete */
            const int64_t fc = final_cost(&P, x);
            if (!best || fc < best_final) { node_free(best); best = x; best_final = fc; } else node_free(x);
            continue;
        }
        /* children, in a fixed order (it decides ties through the ids) */
        int kids[32][3], nk = 0;                                   /* base of consensus 1, base of consensus 2, splits? */
        if (!x->dual) {
            for (int a = 0; a < nc[0]; ++a) { kids[nk][0] = cand[0][a]; kids[nk][1] = -1; kids[nk][2] = 0; ++nk; }
            if (cfg->allow_dual) for (int a = 0; a < nc[0]; ++a) for (int b = a + 1; b < nc[0]; ++b) { kids[nk][0] = cand[0][a]; kids[nk][1] = cand[0][b]; kids[nk][2] = 1; ++nk; }
        } else {
            const int n0 = nc[0] ? nc[0] : 1, n1 = nc[1] ? nc[1] : 1;
            for (int a = 0; a < n0; ++a) for (int b = 0; b < n1; ++b) { kids[nk][0] = nc[0] ? cand[0][a] : -1; kids[nk][1] = nc[1] ? cand[1][b] : -1; kids[nk][2] = 0; ++nk; }
        }
        for (int k = 0; k < nk; ++k) {
            node* c = (k + 1 == nk) ? x : node_clone(&P, x);      /* the last child takes the node over */
            if (kids[k][2]) {                                     /* consensus 2 starts as a copy of consensus 1 */
                c->dual = 1; c->split_at = c->t; c->stopped[1] = 0;
                memcpy(c->C[1], c->C[0], (size_t)c->t); c->len[1] = c->len[0];
                memcpy(c->st[1], c->st[0], sizeof(dwfa) * (size_t)n_reads);
            }
            const int b[2] = { kids[k][0], kids[k][1] };
            node_push(&P, c, b);
            c->id = next_id++;
            if (nq >= QCAP) { node_free(c); continue; }            /* (never reached with max_queue_size <= 32: the threshold below keeps the queue short) */
            queue[nq++] = c;
        }
        /* too many nodes wait: the length threshold rises until at most max_queue_size nodes stand at or above it -- the SHORTEST nodes go, the search is
         * pushed forwards (CdwfaConfig::max_queue_size: "if the queue exceeds this size, the threshold for ignoring nodes is increased") */
        for (;;) {
            int live = 0, shortest = 1 << 30;
            for (int q = 0; q < nq; ++q) if (queue[q]->t >= threshold) { ++live; if (queue[q]->t < shortest) shortest = queue[q]->t; }
            if (live <= max_queue) break;
            threshold = shortest + 1;
        }
        for (int q = 0; q < nq;) { if (queue[q]->t < threshold) { node_free(queue[q]); queue[q] = queue[--nq]; } else ++q; }
    }
    for (int q = 0; q < nq; ++q) node_free(queue[q]);
    free(processed);
    memset(res, 0, sizeof *res); res->split_at = -1; res->best_total = 1;
    res->nodes_expanded = pops;
    if (!best) { res->gave_up = 1; for (int r = 0; r < n_reads; ++r) { score1[r] = score2[r] = -1; is_cons1[r] = 1; } return 0; }
    memcpy(cons1, best->C[0], (size_t)best->len[0]);
    if (best->dual) memcpy(cons2, best->C[1], (size_t)best->len[1]);
    for (int r = 0; r < n_reads; ++r) {
        int sc[2] = { -1, -1 };
        for (int i = 0; i < (best->dual ? 2 : 1); ++i) sc[i] = read_score(&P, best, i, r);
        score1[r] = sc[0]; score2[r] = sc[1];
        is_cons1[r] = !(sc[1] >= 0 && (sc[0] < 0 || sc[1] < sc[0]));
    }
    res->is_dual = best->dual; res->len1 = best->len[0]; res->len2 = best->dual ? best->len[1] : 0; res->split_at = best->split_at;
    res->nodes_expanded = pops;
    node_free(best);
    return 0;
}
