/*
 * oracle/mm2.c -- CPU ORACLE (test infrastructure, never shipped, never linked by the product).
 *
 * A plain scalar restatement of minimap2's published mapping algorithm at the `map-hifi` preset, the
 * configuration the reference fixes in `standard_hifi_aligner` (src/util/mapping.rs:8-14) and uses at every
 * `Aligner::map` call site listed in SURVEY.md section 8(a).  See mm2_oracle.h for what is restated and why.
 * minimap2's sources are not on disk (crate minimap2 0.1.23+minimap2.2.28, Cargo.lock:1137-1152): nothing here
 * is checked against minimap2 itself, only against the cases the reference's tests pin and a brute-force DP.
 */
#include "mm2_oracle.h"
#include "sp_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define NEGINF (-(1 << 29))

void omm_default_opts(omm_opts* o) {
    memset(o, 0, sizeof(*o));
    o->k = 19; o->w = 19;
    o->a = 1; o->b = 4; o->q = 6; o->e = 2; o->q2 = 26; o->e2 = 1; o->sc_ambi = 1;
    o->zdrop = 400; o->zdrop_inv = 200; o->end_bonus = -1;
    o->bw = 500; o->max_gap = 10000;
    o->min_cnt = 3; o->min_chain_score = 40; o->max_chain_skip = 25; o->max_chain_iter = 5000;
    o->chain_gap_scale = 0.8f; o->mask_level = 0.5f; o->pri_ratio = 0.8f; o->best_n = 5;
    o->min_dp_max = 200; o->min_ksw_len = 200;
    o->min_mid_occ = 50; o->max_mid_occ = 500; o->max_max_occ = 4095; o->occ_dist = 500; o->mid_occ_frac = 2e-4f;
    o->forward_only = 0;
}

static inline int gapcost(const omm_opts* o, int l) {
    const int c1 = o->q + l * o->e, c2 = o->q2 + l * o->e2;
    return c1 < c2 ? c1 : c2;
}
static inline int subst(const omm_opts* o, int ct, int cq) {
    if (ct > 3 || cq > 3) return -o->sc_ambi;
    return ct == cq ? o->a : -o->b;
}

/* ------------------------------------------------------------------ minimizers */

typedef struct { uint64_t x, y; } mz_t;      /* x = hash << 8 | span ; y = rid << 32 | end_pos << 1 | strand */
typedef struct { mz_t* a; int64_t n, m; } mzv_t;

static void mzv_push(mzv_t* v, mz_t z) {
    if (v->n == v->m) { v->m = v->m ? v->m * 2 : 256; v->a = (mz_t*)realloc(v->a, sizeof(mz_t) * (size_t)v->m); }
    v->a[v->n++] = z;
}

/* the invertible integer hash of the sketch (Thomas Wang's 64-bit mix restricted to 2k bits) */
static inline uint64_t mix64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

/* (w,k)-minimizers: in every window of w consecutive k-mers of an N-free stretch the k-mer(s) with the smallest
 * hash of the lexicographically smaller strand; a stretch shorter than a window contributes its smallest k-mer */
static void sketch(const uint8_t* s, int len, int w, int k, uint32_t rid, mzv_t* out) {
    const uint64_t mask = (1ULL << 2 * k) - 1, shift1 = 2 * (uint64_t)(k - 1);
    mz_t* ring = (mz_t*)malloc(sizeof(mz_t) * (size_t)w);
    uint64_t fw = 0, rv = 0;
    int l = 0;                                  /* valid bases in the current stretch */
    int64_t last_emitted_y_pos = -1;
    int n_in_stretch = 0;                       /* k-mers seen in the current stretch */
    int emitted_in_stretch = 0;
    mz_t stretch_min = { UINT64_MAX, UINT64_MAX };
    for (int i = 0; i <= len; ++i) {
        const int c = i < len ? s[i] : 4;
        if (c > 3) {                            /* stretch ends */
            if (n_in_stretch > 0 && !emitted_in_stretch && stretch_min.x != UINT64_MAX) mzv_push(out, stretch_min);
            l = 0; fw = rv = 0; n_in_stretch = 0; emitted_in_stretch = 0;
            stretch_min.x = stretch_min.y = UINT64_MAX;
            continue;
        }
        fw = (fw << 2 | (uint64_t)c) & mask;
        rv = (rv >> 2) | (3ULL ^ (uint64_t)c) << shift1;
        if (++l < k) continue;
        mz_t info = { UINT64_MAX, UINT64_MAX };
        if (fw != rv) {
            const int z = fw < rv ? 0 : 1;
            info.x = mix64(z ? rv : fw, mask) << 8 | (uint64_t)k;
            info.y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)z;
        }
        ring[n_in_stretch % w] = info;
        ++n_in_stretch;
        if (info.x <= stretch_min.x && info.x != UINT64_MAX) stretch_min = info;
        if (n_in_stretch >= w) {                /* a full window ends here */
            uint64_t mn = UINT64_MAX;
            for (int j = 0; j < w; ++j) if (ring[j].x < mn) mn = ring[j].x;
            if (mn == UINT64_MAX) continue;
            /* emit in position order the tied minima not yet emitted */
            for (int back = w - 1; back >= 0; --back) {
                const mz_t* c2 = &ring[(n_in_stretch - 1 - back) % w];
                if (c2->x != mn) continue;
                const int64_t pos = (int64_t)((uint32_t)c2->y >> 1);
                if (pos > last_emitted_y_pos) { mzv_push(out, *c2); last_emitted_y_pos = pos; emitted_in_stretch = 1; }
            }
        }
    }
    free(ring);
}

/* ------------------------------------------------------------------ index */

struct omm_index {
    int32_t n_seqs, k, w;
    const uint8_t* codes;       /* borrowed */
    int64_t* off;               /* n_seqs + 1 */
    mz_t* mz;                   /* sorted by hash, then y */
    int64_t n_mz;
    uint64_t* keys;             /* distinct hashes (x >> 8), ascending */
    int64_t* start;             /* n_keys + 1 */
    int64_t n_keys;
    int32_t mid_occ;
};

static int mz_cmp(const void* a, const void* b) {
    const mz_t* x = (const mz_t*)a; const mz_t* y = (const mz_t*)b;
    if (x->x >> 8 != y->x >> 8) return x->x >> 8 < y->x >> 8 ? -1 : 1;
    return x->y < y->y ? -1 : (x->y > y->y);
}
static int u32_cmp(const void* a, const void* b) { const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }

omm_index* omm_index_build(const uint8_t* codes, const int64_t* offsets, int32_t n_seqs, const omm_opts* o) {
    omm_index* idx = (omm_index*)calloc(1, sizeof(*idx));
    idx->n_seqs = n_seqs; idx->k = o->k; idx->w = o->w; idx->codes = codes;
    idx->off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_seqs + 1));
    memcpy(idx->off, offsets, sizeof(int64_t) * (size_t)(n_seqs + 1));
    mzv_t v = { 0, 0, 0 };
    for (int32_t r = 0; r < n_seqs; ++r) sketch(codes + offsets[r], (int)(offsets[r + 1] - offsets[r]), o->w, o->k, (uint32_t)r, &v);
    qsort(v.a, (size_t)v.n, sizeof(mz_t), mz_cmp);
    idx->mz = v.a; idx->n_mz = v.n;
    int64_t nk = 0;
    for (int64_t i = 0; i < v.n; ++i) if (i == 0 || v.a[i].x >> 8 != v.a[i - 1].x >> 8) ++nk;
    idx->keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(nk + 1));
    idx->start = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nk + 1));
    nk = 0;
    for (int64_t i = 0; i < v.n; ++i) if (i == 0 || v.a[i].x >> 8 != v.a[i - 1].x >> 8) { idx->keys[nk] = v.a[i].x >> 8; idx->start[nk] = i; ++nk; }
    idx->start[nk] = v.n; idx->n_keys = nk;
    /* occurrence threshold: the count at the top mid_occ_frac of the distinct minimizers, + 1, clamped */
    int32_t mid = 0x7fffffff;
    if (o->mid_occ_frac > 0.f && nk > 0) {
        uint32_t* cnt = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nk);
        for (int64_t i = 0; i < nk; ++i) cnt[i] = (uint32_t)(idx->start[i + 1] - idx->start[i]);
        qsort(cnt, (size_t)nk, sizeof(uint32_t), u32_cmp);
        int64_t kth = (int64_t)((1.0 - (double)o->mid_occ_frac) * (double)nk);
        if (kth >= nk) kth = nk - 1;
        mid = (int32_t)cnt[kth] + 1;
        free(cnt);
    }
    if (mid < o->min_mid_occ) mid = o->min_mid_occ;
    if (o->max_mid_occ > o->min_mid_occ && mid > o->max_mid_occ) mid = o->max_mid_occ;
    idx->mid_occ = mid;
    return idx;
}
void omm_index_free(omm_index* idx) {
    if (!idx) return;
    free(idx->off); free(idx->mz); free(idx->keys); free(idx->start); free(idx);
}
int32_t omm_index_mid_occ(const omm_index* idx) { return idx->mid_occ; }
int64_t omm_index_n_minimizers(const omm_index* idx) { return idx->n_mz; }

static int64_t idx_lookup(const omm_index* idx, uint64_t key, int64_t* n) {
    int64_t lo = 0, hi = idx->n_keys;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (idx->keys[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo == idx->n_keys || idx->keys[lo] != key) { *n = 0; return 0; }
    *n = idx->start[lo + 1] - idx->start[lo];
    return idx->start[lo];
}

/* ------------------------------------------------------------------ two-piece affine DP */

typedef struct {
    int score, max, max_t, max_q, mqe, mqe_t, zdropped, reach_end;
    int t_end, q_end;                 /* exclusive ends of the reported alignment */
    uint32_t* cigar; int n_cigar, m_cigar;
} ez_t;

static void ez_push(ez_t* ez, int op, int len) {
    if (len <= 0) return;
    if (ez->n_cigar > 0 && (int)(ez->cigar[ez->n_cigar - 1] & 15u) == op) { ez->cigar[ez->n_cigar - 1] += (uint32_t)len << 4; return; }
    if (ez->n_cigar == ez->m_cigar) { ez->m_cigar = ez->m_cigar ? ez->m_cigar * 2 : 16; ez->cigar = (uint32_t*)realloc(ez->cigar, sizeof(uint32_t) * (size_t)ez->m_cigar); }
    ez->cigar[ez->n_cigar++] = (uint32_t)len << 4 | (uint32_t)op;
}

/* rows = target, columns = query.  States H, E / E2 (deletion: consumes target), F / F2 (insertion: consumes query).
 * A gap of l bases costs min(q + l e, q2 + l e2).  mode 0: global; mode 1: extension from the origin.
 * zdrop < 0: off.  right: ties prefer gaps and continuations (used on reversed sequences so that gaps stay left-aligned
 * in forward coordinates).  rev_cigar: leave the operations in traceback order (end -> origin). */
static void dp_align(const uint8_t* t, int tlen, const uint8_t* q, int qlen, const omm_opts* o, int w, int mode,
                     int zdrop, int end_bonus, int right, int rev_cigar, ez_t* ez) {
    ez->score = NEGINF; ez->max = 0; ez->max_t = ez->max_q = -1; ez->mqe = NEGINF; ez->mqe_t = -1;
    ez->zdropped = 0; ez->reach_end = 0; ez->n_cigar = 0; ez->t_end = ez->q_end = 0;
    if (tlen <= 0 || qlen <= 0) {
        if (mode == 0) {              /* one side empty: a single gap */
            if (tlen > 0) { ez_push(ez, 2, tlen); ez->score = -gapcost(o, tlen); ez->t_end = tlen; }
            else if (qlen > 0) { ez_push(ez, 1, qlen); ez->score = -gapcost(o, qlen); ez->q_end = qlen; }
            else ez->score = 0;
        }
        return;
    }
    const int lmax = tlen > qlen ? tlen : qlen;
    if (w > lmax) w = lmax;
    if (mode == 0) { const int dl = tlen > qlen ? tlen - qlen : qlen - tlen; if (w < dl) w = dl; }
    const int bwid = 2 * w + 1;
    int32_t* Hm = (int32_t*)malloc(sizeof(int32_t) * (size_t)tlen * (size_t)bwid);
    uint8_t* Dm = (uint8_t*)malloc((size_t)tlen * (size_t)bwid);
    int32_t* Hp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 2));   /* H(i-1, j) at [j+1] */
    int32_t* Hc = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 2));
    int32_t* E1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 2));   /* E(i-1, j) at [j+1] */
    int32_t* E2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 2));
    for (int j = 0; j <= qlen + 1; ++j) { Hp[j] = Hc[j] = E1[j] = E2[j] = NEGINF; }
    Hp[0] = 0;
    for (int j = 0; j < qlen && j + 1 <= w; ++j) Hp[j + 1] = -gapcost(o, j + 1);
    const int q1 = o->q, e1 = o->e, q2 = o->q2, e2 = o->e2;
    for (int i = 0; i < tlen; ++i) {
        int jlo = i - w, jhi = i + w;
        if (jlo < 0) jlo = 0;
        if (jhi > qlen - 1) jhi = qlen - 1;
        Hc[0] = (i + 1 <= w) ? -gapcost(o, i + 1) : NEGINF;
        if (jlo > 0) Hc[jlo] = NEGINF;                 /* cell jlo-1 is outside this row's band */
        int32_t F1 = NEGINF, F2 = NEGINF;
        int32_t* hrow = Hm + (size_t)i * bwid; uint8_t* drow = Dm + (size_t)i * bwid;
        for (int x = 0; x < bwid; ++x) { hrow[x] = NEGINF; drow[x] = 0; }
        const int ct = t[i];
        for (int j = jlo; j <= jhi; ++j) {
            uint8_t d = 0;
            /* E(i,j) from the cell above */
            int32_t eo = Hp[j + 1] - q1, ee = E1[j + 1], ev;
            if (right ? ee >= eo : ee > eo) { ev = ee - e1; d |= 0x08; } else ev = eo - e1;
            int32_t eo2 = Hp[j + 1] - q2, ee2 = E2[j + 1], ev2;
            if (right ? ee2 >= eo2 : ee2 > eo2) { ev2 = ee2 - e2; d |= 0x20; } else ev2 = eo2 - e2;
            if (ev < NEGINF) ev = NEGINF;
            if (ev2 < NEGINF) ev2 = NEGINF;
            /* F(i,j) from the cell to the left */
            int32_t fo = Hc[j] - q1, fv;
            if (right ? F1 >= fo : F1 > fo) { fv = F1 - e1; d |= 0x10; } else fv = fo - e1;
            int32_t fo2 = Hc[j] - q2, fv2;
            if (right ? F2 >= fo2 : F2 > fo2) { fv2 = F2 - e2; d |= 0x40; } else fv2 = fo2 - e2;
            if (fv < NEGINF) fv = NEGINF;
            if (fv2 < NEGINF) fv2 = NEGINF;
            int32_t h = Hp[j] + subst(o, ct, q[j]);
            int st = 0;
            if (right) {
                if (ev >= h) { h = ev; st = 1; }
                if (fv >= h) { h = fv; st = 2; }
                if (ev2 >= h) { h = ev2; st = 3; }
                if (fv2 >= h) { h = fv2; st = 4; }
            } else {
                if (ev > h) { h = ev; st = 1; }
                if (fv > h) { h = fv; st = 2; }
                if (ev2 > h) { h = ev2; st = 3; }
                if (fv2 > h) { h = fv2; st = 4; }
            }
            if (h < NEGINF) h = NEGINF;
            d |= (uint8_t)st;
            Hc[j + 1] = h; E1[j + 1] = ev; E2[j + 1] = ev2; F1 = fv; F2 = fv2;
            hrow[j - (i - w)] = h; drow[j - (i - w)] = d;
        }
        if (jhi + 2 <= qlen + 1) { Hc[jhi + 2] = NEGINF; E1[jhi + 2] = NEGINF; E2[jhi + 2] = NEGINF; }
        int32_t* tmp = Hp; Hp = Hc; Hc = tmp;
    }
#define HM(i, j) Hm[(size_t)(i) * bwid + ((j) - ((i) - w))]
#define DM(i, j) Dm[(size_t)(i) * bwid + ((j) - ((i) - w))]
    if (tlen - 1 - (qlen - 1) <= w && (qlen - 1) - (tlen - 1) <= w) ez->score = HM(tlen - 1, qlen - 1);
    /* anti-diagonal sweep: running maximum, best at the query end, z-drop */
    if (mode == 1 || zdrop >= 0) {
        for (int r = 0; r <= tlen + qlen - 2; ++r) {
            int st = r - qlen + 1, en = r;
            if (st < 0) st = 0;
            if (en > tlen - 1) en = tlen - 1;
            { int s2 = r - w; s2 = s2 >= 0 ? (s2 + 1) >> 1 : -((-s2) >> 1); if (st < s2) st = s2; }
            { int e2b = (r + w) >> 1; if (en > e2b) en = e2b; }
            if (st > en) continue;
            int32_t mh = NEGINF; int mt = -1;
            for (int i = st; i <= en; ++i) { const int32_t h = HM(i, r - i); if (h > mh) { mh = h; mt = i; } }
            if (r - st == qlen - 1) { const int32_t h = HM(st, qlen - 1); if (h > ez->mqe) { ez->mqe = h; ez->mqe_t = st; } }
            if (mh > ez->max) { ez->max = mh; ez->max_t = mt; ez->max_q = r - mt; }
            else if (mt >= ez->max_t && r - mt >= ez->max_q) {
                const int tl = mt - ez->max_t, ql = (r - mt) - ez->max_q, l = tl > ql ? tl - ql : ql - tl;
                if (zdrop >= 0 && ez->max - mh > zdrop + l * e2) { ez->zdropped = 1; break; }
            }
        }
    }
    int bi = -1, bj = -1;
    if (!ez->zdropped && mode == 0) { bi = tlen - 1; bj = qlen - 1; }
    else if (!ez->zdropped && mode == 1 && ez->mqe + end_bonus > ez->max) { ez->reach_end = 1; bi = ez->mqe_t; bj = qlen - 1; }
    else if (ez->max_t >= 0 && ez->max_q >= 0) { bi = ez->max_t; bj = ez->max_q; }
    if (bi >= 0) {
        ez->t_end = bi + 1; ez->q_end = bj + 1;
        int i = bi, j = bj, state = 0;
        while (i >= 0 && j >= 0) {
            const uint8_t d = DM(i, j);
            if (state == 0) state = d & 7;
            if (state == 0) { ez_push(ez, 0, 1); --i; --j; }
            else if (state == 1) { ez_push(ez, 2, 1); state = (d & 0x08) ? 1 : 0; --i; }
            else if (state == 3) { ez_push(ez, 2, 1); state = (d & 0x20) ? 3 : 0; --i; }
            else if (state == 2) { ez_push(ez, 1, 1); state = (d & 0x10) ? 2 : 0; --j; }
            else { ez_push(ez, 1, 1); state = (d & 0x40) ? 4 : 0; --j; }
        }
        if (i >= 0) ez_push(ez, 2, i + 1);
        if (j >= 0) ez_push(ez, 1, j + 1);
        if (!rev_cigar) for (int x = 0; x < ez->n_cigar / 2; ++x) { const uint32_t c = ez->cigar[x]; ez->cigar[x] = ez->cigar[ez->n_cigar - 1 - x]; ez->cigar[ez->n_cigar - 1 - x] = c; }
    }
#undef HM
#undef DM
    free(Hm); free(Dm); free(Hp); free(Hc); free(E1); free(E2);
}

/* global alignment of a stretch between seeds: a narrow band first; it is exact when its score beats every path that
 * leaves it (such a path pays two extra gaps of the margin's length) */
static void dp_global(const uint8_t* t, int tlen, const uint8_t* q, int qlen, const omm_opts* o, int w, int zdrop, ez_t* ez) {
    const int dl = tlen > qlen ? tlen - qlen : qlen - tlen, margin = 32;
    const int lmin = tlen < qlen ? tlen : qlen;
    if (dl + margin < w && lmin > 0) {
        dp_align(t, tlen, q, qlen, o, dl + margin, 0, zdrop, -1, 0, 0, ez);
        if (!ez->zdropped && ez->score > o->a * lmin - 2 * gapcost(o, margin + 1)) return;
    }
    dp_align(t, tlen, q, qlen, o, w, 0, zdrop, -1, 0, 0, ez);
}

void omm_dp(const uint8_t* t, int32_t tlen, const uint8_t* q, int32_t qlen, const omm_opts* o, int32_t band,
            int32_t mode, int32_t right_align, int32_t* out8, uint32_t* cigar, int32_t cap, int32_t* n_cigar) {
    ez_t ez; memset(&ez, 0, sizeof(ez));
    dp_align(t, tlen, q, qlen, o, band, mode, mode == 1 ? o->zdrop : -1, o->end_bonus, right_align, 0, &ez);
    out8[0] = ez.score; out8[1] = ez.max; out8[2] = ez.max_t; out8[3] = ez.max_q; out8[4] = ez.zdropped;
    out8[5] = ez.reach_end; out8[6] = ez.t_end; out8[7] = ez.q_end;
    int n = ez.n_cigar < cap ? ez.n_cigar : cap;
    if (cigar) memcpy(cigar, ez.cigar, sizeof(uint32_t) * (size_t)n);
    if (n_cigar) *n_cigar = ez.n_cigar;
    free(ez.cigar);
}

int32_t omm_global_score_bruteforce(const uint8_t* t, int32_t tlen, const uint8_t* q, int32_t qlen, const omm_opts* o) {
    /* H(i,j) = best of: diagonal, or ANY single gap of length l ending here (cost min(q+le, q2+le2)): O(mn(m+n)) */
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(tlen + 1) * (size_t)(qlen + 1));
#define HH(i, j) H[(size_t)(i) * (qlen + 1) + (j)]
    for (int i = 0; i <= tlen; ++i) for (int j = 0; j <= qlen; ++j) {
        if (i == 0 && j == 0) { HH(0, 0) = 0; continue; }
        int32_t best = NEGINF;
        if (i > 0 && j > 0) best = HH(i - 1, j - 1) + subst(o, t[i - 1], q[j - 1]);
        for (int l = 1; l <= i; ++l) { const int32_t v = HH(i - l, j) - gapcost(o, l); if (v > best) best = v; }
        for (int l = 1; l <= j; ++l) { const int32_t v = HH(i, j - l) - gapcost(o, l); if (v > best) best = v; }
        HH(i, j) = best;
    }
    const int32_t r = HH(tlen, qlen);
#undef HH
    free(H);
    return r;
}

/* ------------------------------------------------------------------ seeding, chaining */

typedef struct { uint64_t x, y; } an_t;      /* x = rev << 63 | rid << 32 | t_end_pos ; y = span << 32 | q_end_pos (strand coords) */

static int an_cmp(const void* a, const void* b) {
    const an_t* x = (const an_t*)a; const an_t* y = (const an_t*)b;
    if (x->x != y->x) return x->x < y->x ? -1 : 1;
    return x->y < y->y ? -1 : (x->y > y->y);
}

typedef struct { int64_t st; int64_t n; int32_t q_pos, q_span, strand, flt; } seed_t;

static an_t* collect_anchors(const omm_index* idx, const uint8_t* q, int qlen, const omm_opts* o, int64_t* n_out) {
    mzv_t mv = { 0, 0, 0 };
    sketch(q, qlen, o->w, o->k, 0, &mv);
    seed_t* sd = (seed_t*)malloc(sizeof(seed_t) * (size_t)(mv.n + 1));
    int64_t ns = 0;
    for (int64_t i = 0; i < mv.n; ++i) {
        int64_t n; const int64_t st = idx_lookup(idx, mv.a[i].x >> 8, &n);
        if (n == 0) continue;
        seed_t s; s.st = st; s.n = n; s.q_pos = (int32_t)((uint32_t)mv.a[i].y >> 1); s.q_span = (int32_t)(mv.a[i].x & 0xff);
        s.strand = (int32_t)(mv.a[i].y & 1); s.flt = 0;
        sd[ns++] = s;
    }
    /* occurrence filter with the rescue of long high-occurrence streaks */
    const int max_occ = idx->mid_occ;
    if (o->occ_dist > 0 && o->max_max_occ > max_occ) {
        int64_t last0 = -1;
        for (int64_t i = 0; i <= ns; ++i) {
            if (i == ns || sd[i].n <= max_occ) {
                if (i - last0 > 1) {
                    const int ps = last0 < 0 ? 0 : sd[last0].q_pos, pe = i == ns ? qlen : sd[i].q_pos;
                    int keep = (int)((double)(pe - ps) / o->occ_dist + .499);
                    if (keep > 128) keep = 128;
                    for (int64_t j = last0 + 1; j < i; ++j) sd[j].flt = 1;
                    for (int kk = 0; kk < keep; ++kk) {         /* the `keep` lowest-occurrence seeds of the streak */
                        int64_t bj = -1;
                        for (int64_t j = last0 + 1; j < i; ++j) if (sd[j].flt == 1 && (bj < 0 || sd[j].n < sd[bj].n)) bj = j;
                        if (bj < 0) break;
                        sd[bj].flt = 2;
                    }
                    for (int64_t j = last0 + 1; j < i; ++j) sd[j].flt = (sd[j].flt == 2 && sd[j].n <= o->max_max_occ) ? 0 : 1;
                }
                last0 = i;
            }
        }
    } else for (int64_t i = 0; i < ns; ++i) if (sd[i].n > max_occ) sd[i].flt = 1;
    int64_t na = 0;
    for (int64_t i = 0; i < ns; ++i) if (!sd[i].flt) na += sd[i].n;
    an_t* a = (an_t*)malloc(sizeof(an_t) * (size_t)(na + 1));
    na = 0;
    for (int64_t i = 0; i < ns; ++i) {
        if (sd[i].flt) continue;
        for (int64_t h = 0; h < sd[i].n; ++h) {
            const mz_t* m = &idx->mz[sd[i].st + h];
            const uint64_t rid = m->y >> 32, rpos = (uint32_t)m->y >> 1;
            const int same = (int)(m->y & 1) == sd[i].strand;
            an_t z;
            if (same) { z.x = rid << 32 | rpos; z.y = (uint64_t)sd[i].q_span << 32 | (uint64_t)(uint32_t)sd[i].q_pos; }
            else {
                if (o->forward_only) continue;
                z.x = 1ULL << 63 | rid << 32 | rpos;
                z.y = (uint64_t)sd[i].q_span << 32 | (uint64_t)(uint32_t)(qlen - (sd[i].q_pos + 1 - sd[i].q_span) - 1);
            }
            a[na++] = z;
        }
    }
    qsort(a, (size_t)na, sizeof(an_t), an_cmp);
    free(sd); free(mv.a);
    *n_out = na;
    return a;
}

static inline int32_t chain_sc(const an_t* ai, const an_t* aj, int max_dist, int bw, float pen_gap) {
    const int32_t dq = (int32_t)ai->y - (int32_t)aj->y;
    if (dq <= 0 || dq > max_dist) return INT32_MIN;
    const int32_t dr = (int32_t)(ai->x - aj->x);
    if (dr == 0 || dq > max_dist) return INT32_MIN;
    const int32_t dd = dr > dq ? dr - dq : dq - dr;
    if (dd > bw) return INT32_MIN;
    const int32_t dg = dr < dq ? dr : dq, q_span = (int32_t)(aj->y >> 32 & 0xff);
    int32_t sc = q_span < dg ? q_span : dg;
    if (dd || dg > q_span) {
        const float lin = pen_gap * (float)dd;
        const float lg = dd >= 1 ? log2f((float)(dd + 1)) : 0.0f;
        sc -= (int32_t)(lin + .5f * lg);
    }
    return sc;
}

typedef struct {
    int64_t as; int32_t cnt;           /* seeds v[as .. as+cnt) in chain order (ascending) */
    int32_t score, rev, rid;
    int32_t qs, qe, rs, re;            /* query in FORWARD coordinates */
    int32_t parent, id;
} reg_t;

typedef struct { int32_t f; int64_t i; } fz_t;
static int fz_cmp(const void* a, const void* b) { const fz_t* x = (const fz_t*)a; const fz_t* y = (const fz_t*)b; if (x->f != y->f) return x->f < y->f ? -1 : 1; return x->i < y->i ? -1 : (x->i > y->i); }

/* chains of the sorted anchors; returns regs (sorted by score, descending) and the chained anchors in `*v_out` */
static reg_t* chain_anchors(const an_t* a, int64_t n, int qlen, const omm_opts* o, an_t** v_out, int* n_regs) {
    *n_regs = 0; *v_out = NULL;
    if (n == 0) return NULL;
    int32_t* f = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int64_t* p = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* t = (int64_t*)calloc((size_t)n, sizeof(int64_t));
    const float pen_gap = 0.01f * o->chain_gap_scale * (float)o->k;
    int64_t st = 0, max_ii = -1;
    for (int64_t i = 0; i < n; ++i) {
        int64_t max_j = -1; int32_t max_f = (int32_t)(a[i].y >> 32 & 0xff); int n_skip = 0;
        while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)o->max_gap)) ++st;
        if (i - st > o->max_chain_iter) st = i - o->max_chain_iter;
        int64_t j;
        for (j = i - 1; j >= st; --j) {
            int32_t sc = chain_sc(&a[i], &a[j], o->max_gap, o->bw, pen_gap);
            if (sc == INT32_MIN) continue;
            sc += f[j];
            if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
            else if (t[j] == i + 1) { if (++n_skip > o->max_chain_skip) break; }
            if (p[j] >= 0) t[p[j]] = i + 1;
        }
        const int64_t end_j = j;
        if (max_ii < 0 || a[i].x - a[max_ii].x > (uint64_t)o->max_gap || a[i].x >> 32 != a[max_ii].x >> 32) {
            int32_t mx = INT32_MIN; max_ii = -1;
            for (j = i - 1; j >= st; --j) if (mx < f[j]) { mx = f[j]; max_ii = j; }
        }
        if (max_ii >= 0 && max_ii < end_j) {
            const int32_t tmp = chain_sc(&a[i], &a[max_ii], o->max_gap, o->bw, pen_gap);
            if (tmp != INT32_MIN && max_f < tmp + f[max_ii]) { max_f = tmp + f[max_ii]; max_j = max_ii; }
        }
        f[i] = max_f; p[i] = max_j;
        if (max_ii < 0 || (a[i].x - a[max_ii].x <= (uint64_t)o->max_gap && f[max_ii] < f[i])) max_ii = i;
    }
    /* backtrack, best end first; a chain is cut where its score has fallen by more than the band before recovering */
    fz_t* z = (fz_t*)malloc(sizeof(fz_t) * (size_t)n); int64_t nz = 0;
    for (int64_t i = 0; i < n; ++i) if (f[i] >= o->min_chain_score) { z[nz].f = f[i]; z[nz].i = i; ++nz; }
    qsort(z, (size_t)nz, sizeof(fz_t), fz_cmp);
    memset(t, 0, sizeof(int64_t) * (size_t)n);
    an_t* v = (an_t*)malloc(sizeof(an_t) * (size_t)(n + 1)); int64_t nv = 0;
    reg_t* regs = (reg_t*)malloc(sizeof(reg_t) * (size_t)(nz + 1)); int nr = 0;
    for (int64_t k = nz - 1; k >= 0; --k) {
        if (t[z[k].i] != 0) continue;
        int64_t i = z[k].i, end_i = -1, max_i = i; int32_t max_s = 0;
        do {
            t[i] = 2; end_i = i = p[i];
            const int32_t s = i < 0 ? z[k].f : z[k].f - f[i];
            if (s > max_s) { max_s = s; max_i = i; }
            else if (max_s - s > o->bw) break;
        } while (i >= 0 && t[i] == 0);
        for (i = z[k].i; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
        end_i = max_i;
        const int64_t nv0 = nv;
        for (i = z[k].i; i != end_i; i = p[i]) { v[nv++] = a[i]; t[i] = 1; }
        const int32_t sc = i < 0 ? z[k].f : z[k].f - f[i];
        if (sc >= o->min_chain_score && nv - nv0 >= o->min_cnt) {
            for (int64_t x = 0; x < (nv - nv0) / 2; ++x) { const an_t c = v[nv0 + x]; v[nv0 + x] = v[nv - 1 - x]; v[nv - 1 - x] = c; }
            reg_t* r = &regs[nr];
            r->as = nv0; r->cnt = (int32_t)(nv - nv0); r->score = sc; r->rev = (int32_t)(v[nv0].x >> 63); r->rid = (int32_t)(v[nv0].x << 1 >> 33);
            const an_t* first = &v[nv0]; const an_t* last = &v[nv - 1];
            r->rs = (int32_t)first->x + 1 - (int32_t)(first->y >> 32 & 0xff); r->re = (int32_t)last->x + 1;
            const int32_t qs = (int32_t)first->y + 1 - (int32_t)(first->y >> 32 & 0xff), qe = (int32_t)last->y + 1;
            if (r->rev) { r->qs = qlen - qe; r->qe = qlen - qs; } else { r->qs = qs; r->qe = qe; }
            r->parent = nr; r->id = nr;
            ++nr;
        } else nv = nv0;
    }
    /* by score, descending (stable on creation order) */
    for (int i = 1; i < nr; ++i) { reg_t c = regs[i]; int j = i - 1; while (j >= 0 && regs[j].score < c.score) { regs[j + 1] = regs[j]; --j; } regs[j + 1] = c; }
    free(f); free(p); free(t); free(z);
    *v_out = v; *n_regs = nr;
    return regs;
}

/* a region overlapping a better one on the query by more than mask_level of the shorter is its secondary */
static void set_parent(reg_t* r, int n, float mask_level) {
    for (int i = 0; i < n; ++i) {
        r[i].parent = i;
        for (int j = 0; j < i; ++j) {
            if (r[j].parent != j) continue;
            const int sj = r[j].qs, ej = r[j].qe, si = r[i].qs, ei = r[i].qe;
            const int mn = (ej - sj) < (ei - si) ? (ej - sj) : (ei - si);
            const int ol = (ei < ej ? ei : ej) - (si > sj ? si : sj);
            if (ol > 0 && (float)ol > mask_level * (float)mn) { r[i].parent = j; break; }
        }
    }
}
static int select_sub(reg_t* r, int n, float pri_ratio, int min_diff, int best_n) {
    int k = 0, n2 = 0;
    int* newpos = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) {
        const int p = r[i].parent; newpos[i] = -1;
        if (p == i) { newpos[i] = k; r[k++] = r[i]; }
        else if (((float)r[i].score >= (float)r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n2 < best_n) {
            if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) { newpos[i] = k; r[k++] = r[i]; ++n2; }
        }
    }
    for (int i = 0; i < k; ++i) { const int p = r[i].parent; r[i].parent = (p >= 0 && p < n && newpos[p] >= 0) ? newpos[p] : i; }
    free(newpos);
    return k;
}

/* ------------------------------------------------------------------ base-level alignment of a chain */

typedef struct { uint32_t* c; int n, m; } cig_t;
static void cig_push(cig_t* c, int op, int len) {
    if (len <= 0) return;
    if (c->n > 0 && (int)(c->c[c->n - 1] & 15u) == op) { c->c[c->n - 1] += (uint32_t)len << 4; return; }
    if (c->n == c->m) { c->m = c->m ? c->m * 2 : 32; c->c = (uint32_t*)realloc(c->c, sizeof(uint32_t) * (size_t)c->m); }
    c->c[c->n++] = (uint32_t)len << 4 | (uint32_t)op;
}
static void cig_append(cig_t* c, const ez_t* ez) { for (int i = 0; i < ez->n_cigar; ++i) cig_push(c, (int)(ez->cigar[i] & 15u), (int)(ez->cigar[i] >> 4)); }

/* the drop test along a finished global alignment: largest fall of the running score below an earlier peak, corrected for the
 * diagonal shift (as the second, z-drop-enabled pass is only made when this says it would trigger) */
static int path_zdrop(const uint8_t* t, const uint8_t* q, const ez_t* ez, const omm_opts* o) {
    int i = 0, j = 0, score = 0, mx = 0, mi = -1, mj = -1, worst = 0;
    for (int k = 0; k < ez->n_cigar; ++k) {
        const int op = (int)(ez->cigar[k] & 15u), len = (int)(ez->cigar[k] >> 4);
        if (op == 0) {
            for (int l = 0; l < len; ++l) {
                score += subst(o, t[i + l], q[j + l]);
                if (score > mx) { mx = score; mi = i + l; mj = j + l; }
                else { const int li = i + l - mi, lj = j + l - mj, d = li > lj ? li - lj : lj - li; const int z = mx - score - d * o->e; if (z > worst) worst = z; }
            }
            i += len; j += len;
        } else {
            score -= o->q + o->e * len;
            if (op == 1) j += len; else i += len;
            { const int li = i - mi, lj = j - mj, d = li > lj ? li - lj : lj - li; const int z = mx - score - d * o->e; if (z > worst) worst = z; }
        }
    }
    return worst > o->zdrop;
}

typedef struct {
    int ok, rev, rid, qs, qe, rs, re;          /* query in strand coordinates here */
    int dp_score, n_seeds, chain_score;
    cig_t cig;
} aln_t;

/* aligns seeds v[0..cnt) of one chain; on a z-drop inside the chain the rest of the seeds are returned through *rest_from */
static void align_chain(const omm_index* idx, const uint8_t* qs_codes, int qlen, const an_t* v, int cnt, const omm_opts* o,
                        aln_t* out, int* rest_from) {
    memset(out, 0, sizeof(*out)); *rest_from = -1;
    const int rid = (int)(v[0].x << 1 >> 33);
    const uint8_t* tseq = idx->codes + idx->off[rid];
    const int tlen = (int)(idx->off[rid + 1] - idx->off[rid]);
    const int bw = (int)(o->bw * 1.5 + 1.);
    ez_t ez; memset(&ez, 0, sizeof(ez));
    int rs = (int)(uint32_t)v[0].x + 1 - (int)(v[0].y >> 32 & 0xff), qs = (int)(uint32_t)v[0].y + 1 - (int)(v[0].y >> 32 & 0xff);
    int re0, qe0, rs0, qs0;
    {   /* how far the extensions may look */
        int l = qs < o->max_gap ? qs : o->max_gap;
        qs0 = qs - l;
        l += l * o->a > o->q ? (l * o->a - o->q) / o->e : 0;
        if (l > o->max_gap) l = o->max_gap;
        if (l > rs) l = rs;
        rs0 = rs - l;
        const int qe_last = (int)(uint32_t)v[cnt - 1].y + 1, re_last = (int)(uint32_t)v[cnt - 1].x + 1;
        l = qlen - qe_last < o->max_gap ? qlen - qe_last : o->max_gap;
        qe0 = qe_last + l;
        l += l * o->a > o->q ? (l * o->a - o->q) / o->e : 0;
        if (l > o->max_gap) l = o->max_gap;
        if (l > tlen - re_last) l = tlen - re_last;
        re0 = re_last + l;
    }
    int rs1 = rs, qs1 = qs, re1 = rs, qe1 = qs, dp_score = 0, dropped = 0;
    out->rev = (int)(v[0].x >> 63); out->rid = rid; out->n_seeds = cnt;
    if (qs > qs0 && rs > rs0) {                 /* left extension on the reversed prefixes */
        const int ql = qs - qs0, tl = rs - rs0;
        uint8_t* qr = (uint8_t*)malloc((size_t)ql); uint8_t* tr = (uint8_t*)malloc((size_t)tl);
        for (int i = 0; i < ql; ++i) qr[i] = qs_codes[qs - 1 - i];
        for (int i = 0; i < tl; ++i) tr[i] = tseq[rs - 1 - i];
        dp_align(tr, tl, qr, ql, o, bw, 1, o->zdrop, o->end_bonus, 1, 1, &ez);
        if (ez.n_cigar > 0) {
            cig_append(&out->cig, &ez);
            dp_score += ez.reach_end ? ez.mqe : ez.max;
            rs1 = rs - ez.t_end; qs1 = qs - ez.q_end;
        }
        free(qr); free(tr);
    }
    for (int i = 1; i < cnt; ++i) {             /* between the seeds, in stretches of at least min_ksw_len */
        const int re = (int)(uint32_t)v[i].x + 1, qe = (int)(uint32_t)v[i].y + 1;
        re1 = re; qe1 = qe;
        if (i == cnt - 1 || (qe - qs >= o->min_ksw_len && re - rs >= o->min_ksw_len)) {
            dp_global(tseq + rs, re - rs, qs_codes + qs, qe - qs, o, bw, -1, &ez);
            if (path_zdrop(tseq + rs, qs_codes + qs, &ez, o)) dp_align(tseq + rs, re - rs, qs_codes + qs, qe - qs, o, bw, 0, o->zdrop, -1, 0, 0, &ez);
            cig_append(&out->cig, &ez);
            if (ez.zdropped) {                  /* the alignment ends at the best cell; the remaining seeds form a new region */
                int j; for (j = i - 1; j >= 0; --j) if ((int)(uint32_t)v[j].x <= rs + ez.max_t) break;
                dropped = 1; if (j < 0) j = 0;
                dp_score += ez.max; re1 = rs + ez.max_t + 1; qe1 = qs + ez.max_q + 1;
                if (cnt - (j + 1) >= o->min_cnt) *rest_from = j + 1;
                break;
            }
            dp_score += ez.score;
            rs = re; qs = qe;
        }
    }
    if (cnt == 1) { re1 = (int)(uint32_t)v[0].x + 1; qe1 = (int)(uint32_t)v[0].y + 1; cig_push(&out->cig, 0, re1 - rs); dp_score += o->a * (re1 - rs); }
    if (!dropped) {
        const int re = re1, qe = qe1;
        if (qe < qe0 && re < re0) {             /* right extension */
            dp_align(tseq + re, re0 - re, qs_codes + qe, qe0 - qe, o, bw, 1, o->zdrop, o->end_bonus, 0, 0, &ez);
            if (ez.n_cigar > 0) {
                cig_append(&out->cig, &ez);
                dp_score += ez.reach_end ? ez.mqe : ez.max;
                re1 = re + ez.t_end; qe1 = qe + ez.q_end;
            }
        }
    }
    free(ez.cigar);
    out->ok = out->cig.n > 0; out->rs = rs1; out->re = re1; out->qs = qs1; out->qe = qe1; out->dp_score = dp_score;
}

/* '=' / 'X' operations, NM and the peak score of the finished alignment */
static void finish_hit(const omm_index* idx, const uint8_t* qc, const aln_t* al, const omm_opts* o, int qlen, omm_hit* h,
                       uint32_t* pool, int cap, int* pool_n) {
    const uint8_t* tseq = idx->codes + idx->off[al->rid];
    memset(h, 0, sizeof(*h));
    h->rid = al->rid; h->rev = al->rev; h->t_start = al->rs; h->t_end = al->re; h->t_len = (int32_t)(idx->off[al->rid + 1] - idx->off[al->rid]);
    h->q_len = qlen;
    if (al->rev) { h->q_start = qlen - al->qe; h->q_end = qlen - al->qs; } else { h->q_start = al->qs; h->q_end = al->qe; }
    h->dp_score = al->dp_score; h->chain_score = al->chain_score; h->n_seeds = al->n_seeds;
    cig_t eq = { 0, 0, 0 };
    int i = al->rs, j = al->qs, s = 0, mx = 0;
    for (int k = 0; k < al->cig.n; ++k) {
        const int op = (int)(al->cig.c[k] & 15u), len = (int)(al->cig.c[k] >> 4);
        if (op == 0) {
            for (int l = 0; l < len; ++l) {
                const int ct = tseq[i + l], cq = qc[j + l];
                if (ct > 3 || cq > 3) { ++h->n_ambi; cig_push(&eq, 8, 1); }
                else if (ct == cq) { ++h->mlen; ++h->blen; cig_push(&eq, 7, 1); }
                else { ++h->blen; cig_push(&eq, 8, 1); }
                s += subst(o, ct, cq);
                if (s < 0) s = 0; else if (s > mx) mx = s;
            }
            i += len; j += len;
        } else {
            h->blen += len;
            s -= gapcost(o, len); if (s < 0) s = 0;
            cig_push(&eq, op, len);
            if (op == 1) j += len; else i += len;
        }
    }
    h->nm = h->blen - h->mlen + h->n_ambi; h->dp_max = mx;
    h->cigar_off = *pool_n; h->n_cigar = 0;
    if (pool && *pool_n + eq.n <= cap) { memcpy(pool + *pool_n, eq.c, sizeof(uint32_t) * (size_t)eq.n); h->n_cigar = eq.n; *pool_n += eq.n; }
    free(eq.c);
}

int32_t omm_map(const omm_index* idx, const uint8_t* q, int32_t qlen, const omm_opts* o,
                omm_hit* hits, int32_t max_hits, uint32_t* cigar_pool, int32_t cigar_cap) {
    if (qlen <= 0 || max_hits <= 0) return 0;
    int64_t na = 0;
    an_t* a = collect_anchors(idx, q, qlen, o, &na);
    an_t* v = NULL; int nr = 0;
    reg_t* regs = chain_anchors(a, na, qlen, o, &v, &nr);
    free(a);
    if (nr == 0) { free(regs); free(v); return 0; }
    set_parent(regs, nr, o->mask_level);
    nr = select_sub(regs, nr, o->pri_ratio, o->k * 2, o->best_n);
    uint8_t* qrev = NULL;
    omm_hit* tmp = (omm_hit*)malloc(sizeof(omm_hit) * (size_t)(2 * nr + 4));
    reg_t* hr = (reg_t*)malloc(sizeof(reg_t) * (size_t)(2 * nr + 4));
    int nh = 0, pool_n = 0;
    for (int r = 0; r < nr; ++r) {
        const uint8_t* qc = q;
        if (regs[r].rev) {
            if (!qrev) { qrev = (uint8_t*)malloc((size_t)qlen); for (int i = 0; i < qlen; ++i) { const uint8_t c = q[qlen - 1 - i]; qrev[i] = c > 3 ? c : (uint8_t)(3 - c); } }
            qc = qrev;
        }
        int from = 0, cnt = regs[r].cnt, pieces = 0;
        while (from >= 0 && cnt - from >= 1 && pieces < 2) {           /* a z-drop splits the region once */
            aln_t al; int rest = -1;
            align_chain(idx, qc, qlen, v + regs[r].as + from, cnt - from, o, &al, &rest);
            al.chain_score = regs[r].score;
            if (al.ok && nh < 2 * nr + 4) {
                finish_hit(idx, qc, &al, o, qlen, &tmp[nh], cigar_pool, cigar_cap, &pool_n);
                /* mm_filter_regs: too few seeds / matching bases, or a peak score below min_dp_max */
                if (tmp[nh].n_seeds >= o->min_cnt && tmp[nh].mlen >= o->min_chain_score && tmp[nh].dp_max >= o->min_dp_max) {
                    hr[nh] = regs[r]; hr[nh].qs = tmp[nh].q_start; hr[nh].qe = tmp[nh].q_end; hr[nh].rs = tmp[nh].t_start; hr[nh].re = tmp[nh].t_end; hr[nh].id = nh;
                    ++nh;
                }
            }
            free(al.cig.c);
            if (rest < 0) break;
            from += rest; ++pieces;
        }
    }
    /* output order: by peak DP score, descending; parents again on the aligned intervals; secondaries by chain score ratio */
    int* ord = (int*)malloc(sizeof(int) * (size_t)(nh + 1));
    for (int i = 0; i < nh; ++i) ord[i] = i;
    for (int i = 1; i < nh; ++i) { const int c = ord[i]; int j = i - 1; while (j >= 0 && tmp[ord[j]].dp_max < tmp[c].dp_max) { ord[j + 1] = ord[j]; --j; } ord[j + 1] = c; }
    reg_t* hs = (reg_t*)malloc(sizeof(reg_t) * (size_t)(nh + 1));
    for (int i = 0; i < nh; ++i) { hs[i] = hr[ord[i]]; }
    set_parent(hs, nh, o->mask_level);
    const int nk = select_sub(hs, nh, o->pri_ratio, o->k * 2, o->best_n);
    int n_out = 0;
    for (int i = 0; i < nk && n_out < max_hits; ++i) { hits[n_out] = tmp[hs[i].id]; hits[n_out].primary = hs[i].parent == i; ++n_out; }
    free(ord); free(hs); free(hr); free(tmp); free(qrev); free(regs); free(v);
    return n_out;
}

int32_t omm_map_pair(const uint8_t* target, int32_t tlen, const uint8_t* q, int32_t qlen, const omm_opts* o,
                     omm_hit* hits, int32_t max_hits, uint32_t* cigar_pool, int32_t cigar_cap) {
    const int64_t off[2] = { 0, tlen };
    omm_index* idx = omm_index_build(target, off, 1, o);
    const int32_t n = omm_map(idx, q, qlen, o, hits, max_hits, cigar_pool, cigar_cap);
    omm_index_free(idx);
    return n;
}

/* ------------------------------------------------------------------ score_read on the restatement's mappings */

/* score_read (src/hla/caller.rs:1411-1510): every allele of the gene mapped to the consensus at cDNA and DNA level (the caller passes
 * opts with a = 5), Forward mappings only (:1443-1445), select_best_mapping query based and penalised (:1450), HlaProcessedMatch::add_mapping
 * (src/hla/processed_match.rs:53-100) and the running best by is_better_match (:103-184).  Alleles in database order; returns the best index or
 * -1; stats (may be NULL) = [(allele * 2 + level) * 3 + {len, nm, unmapped}], -1 where a level has no mapping. */
int32_t omm_hla_score_read(const uint8_t* cons_cdna, int32_t cdna_len, const uint8_t* cons_dna, int32_t dna_len, int32_t n_alleles,
                           const uint8_t* const* cdna, const int32_t* cdna_lens, const uint8_t* const* dna, const int32_t* dna_lens,
                           const omm_opts* o, int64_t* stats) {
    const uint8_t* cons[2] = { cons_cdna, cons_dna };
    const int32_t clen[2] = { cdna_len, dna_len };
    omm_index* idx[2] = { NULL, NULL };
    for (int lv = 0; lv < 2; ++lv) if (cons[lv] && clen[lv] > 0) { const int64_t off[2] = { 0, clen[lv] }; idx[lv] = omm_index_build(cons[lv], off, 1, o); }
    osp_hla_level best[2]; memset(best, 0, sizeof(best));
    uint64_t* best_pc[2] = { NULL, NULL };
    int best_idx = -1;
    enum { MAXH = 8 };
    for (int a = 0; a < n_alleles; ++a) {
        osp_hla_level cur[2]; uint64_t* cur_pc[2] = { NULL, NULL };
        memset(cur, 0, sizeof(cur));
        for (int lv = 0; lv < 2; ++lv) {
            const uint8_t* seq = lv ? (dna ? dna[a] : NULL) : (cdna ? cdna[a] : NULL);
            const int32_t slen = lv ? (dna_lens ? dna_lens[a] : 0) : (cdna_lens ? cdna_lens[a] : 0);
            if (stats) { int64_t* st = stats + ((size_t)a * 2 + (size_t)lv) * 3; st[0] = st[1] = st[2] = -1; }
            if (!seq || slen <= 0 || !idx[lv]) continue;
            omm_hit hits[MAXH];
            const int cap = 4 * (slen + clen[lv]) + 64;
            uint32_t* pool = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap);
            const int nh = omm_map(idx[lv], seq, slen, o, hits, MAXH, pool, cap);
            int pick = -1; double ps = 1.0;
            for (int h = 0; h < nh; ++h) {
                if (hits[h].rev) continue;
                const int um = hits[h].q_len - (hits[h].q_end - hits[h].q_start);
                double v = (double)(hits[h].nm + um); if (v < 0.1) v = 0.1;
                v /= (double)hits[h].q_len;
                if (v < ps) { ps = v; pick = h; }
            }
            if (pick >= 0) {
                const omm_hit* H = &hits[pick];
                uint32_t* cl = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(H->n_cigar + 1));
                uint8_t* co = (uint8_t*)malloc((size_t)(H->n_cigar + 1));
                for (int c = 0; c < H->n_cigar; ++c) { cl[c] = pool[H->cigar_off + c] >> 4; co[c] = (uint8_t)(pool[H->cigar_off + c] & 15u); }
                uint64_t* pc = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(clen[lv] + 1));
                const uint64_t cs = (uint64_t)H->q_start, ce = (uint64_t)(H->q_len - H->q_end);
                if (osp_process_mm_cigar(cl, co, H->n_cigar, (uint64_t)H->t_start, (uint64_t)clen[lv], cs, ce, pc) == 0) {
                    cur[lv].present = 1;
                    cur[lv].range_start = H->t_start - (int)cs > 0 ? H->t_start - (int)cs : 0;
                    { const int room = clen[lv] - H->t_end; cur[lv].range_end = H->t_end + ((int)ce < room ? (int)ce : room); }
                    cur[lv].len = H->q_len; cur[lv].nm = H->nm; cur[lv].unmapped = H->q_len - (H->q_end - H->q_start);
                    cur[lv].pc = pc; cur_pc[lv] = pc;
                    if (stats) { int64_t* st = stats + ((size_t)a * 2 + (size_t)lv) * 3; st[0] = cur[lv].len; st[1] = cur[lv].nm; st[2] = cur[lv].unmapped; }
                } else free(pc);
                free(cl); free(co);
            }
            free(pool);
        }
        if (osp_is_better_match(cur, best)) {
            for (int lv = 0; lv < 2; ++lv) { free(best_pc[lv]); best_pc[lv] = cur_pc[lv]; best[lv] = cur[lv]; }
            best_idx = a;
        } else {
            for (int lv = 0; lv < 2; ++lv) free(cur_pc[lv]);
        }
    }
    for (int lv = 0; lv < 2; ++lv) { free(best_pc[lv]); omm_index_free(idx[lv]); }
    return best_idx;
}

/* ------------------------------------------------------------------ the stages of omm_map, one by one (parity tests of the seeded K1 path) */

int32_t omm_sketch(const uint8_t* s, int32_t len, const omm_opts* o, uint64_t* hash, int32_t* end_pos, uint8_t* strand, int32_t cap) {
    mzv_t mv = { 0, 0, 0 };
    sketch(s, len, o->w, o->k, 0, &mv);
    const int32_t n = (int32_t)mv.n;
    for (int32_t i = 0; i < n && i < cap; ++i) {
        if (hash) hash[i] = mv.a[i].x >> 8;
        if (end_pos) end_pos[i] = (int32_t)((uint32_t)mv.a[i].y >> 1);
        if (strand) strand[i] = (uint8_t)(mv.a[i].y & 1);
    }
    free(mv.a);
    return n;
}

/* seeding + chaining + selection of omm_map without the base-level alignment.  regs: [cap][10] = {rid, rev, chain score, seeds, qs, qe (forward
 * coordinates), rs, re, parent (index into this list after selection), kept (1 = survives set_parent + select_sub, in the order align_chain sees them)};
 * the list is every chain in chain_anchors' order (score descending, stable); stats[8] = {minimizers of the query, seeds found in the index, seeds kept
 * by the occurrence filter, anchors, distinct (strand, target) pairs among the anchors, chains, chains selected, mid_occ}.  Returns the number of chains. */
int32_t omm_chain_stage(const omm_index* idx, const uint8_t* q, int32_t qlen, const omm_opts* o, int32_t* regs_out, int32_t cap, int64_t* stats) {
    int64_t na = 0;
    if (stats) {
        mzv_t mv = { 0, 0, 0 };
        sketch(q, qlen, o->w, o->k, 0, &mv);
        stats[0] = mv.n; stats[1] = 0;
        for (int64_t i = 0; i < mv.n; ++i) { int64_t n; idx_lookup(idx, mv.a[i].x >> 8, &n); if (n) ++stats[1]; }
        free(mv.a);
        stats[7] = idx->mid_occ;
    }
    an_t* a = collect_anchors(idx, q, qlen, o, &na);
    if (stats) {
        stats[3] = na; stats[4] = 0;
        for (int64_t i = 0; i < na; ++i) if (i == 0 || a[i].x >> 32 != a[i - 1].x >> 32) ++stats[4];
        /* kept seeds = distinct query positions among the anchors */
        int64_t kept = 0;
        uint8_t* seen = (uint8_t*)calloc((size_t)qlen + 1, 1);
        for (int64_t i = 0; i < na; ++i) {
            int32_t qp = (int32_t)a[i].y;
            if (a[i].x >> 63) qp = qlen - 1 - qp + (int32_t)(a[i].y >> 32 & 0xff) - 1;     /* back to the forward end position */
            if (qp >= 0 && qp <= qlen && !seen[qp]) { seen[qp] = 1; ++kept; }
        }
        free(seen);
        stats[2] = kept;
    }
    an_t* v = NULL; int nr = 0;
    reg_t* regs = chain_anchors(a, na, qlen, o, &v, &nr);
    free(a);
    if (stats) { stats[5] = nr; stats[6] = 0; }
    if (nr == 0) { free(regs); free(v); return 0; }
    reg_t* sel = (reg_t*)malloc(sizeof(reg_t) * (size_t)nr);
    memcpy(sel, regs, sizeof(reg_t) * (size_t)nr);
    for (int i = 0; i < nr; ++i) sel[i].id = i;
    set_parent(sel, nr, o->mask_level);
    const int nk = select_sub(sel, nr, o->pri_ratio, o->k * 2, o->best_n);
    if (stats) stats[6] = nk;
    for (int i = 0; i < nr && i < cap; ++i) {
        int32_t* r = regs_out + (size_t)i * 10;
        r[0] = regs[i].rid; r[1] = regs[i].rev; r[2] = regs[i].score; r[3] = regs[i].cnt;
        r[4] = regs[i].qs; r[5] = regs[i].qe; r[6] = regs[i].rs; r[7] = regs[i].re; r[8] = -1; r[9] = 0;
    }
    for (int i = 0; i < nk; ++i) if (sel[i].id < cap) { int32_t* r = regs_out + (size_t)sel[i].id * 10; r[8] = sel[sel[i].parent].id; r[9] = i + 1; }
    free(sel); free(regs); free(v);
    return nr;
}

/* the sorted anchors of omm_map (collect_anchors): x = rev << 63 | rid << 32 | target end position, y = span << 32 | query end position on the mapped
 * strand; returns their number (may exceed cap) */
int64_t omm_anchors(const omm_index* idx, const uint8_t* q, int32_t qlen, const omm_opts* o, uint64_t* x, uint64_t* y, int64_t cap) {
    int64_t na = 0;
    an_t* a = collect_anchors(idx, q, qlen, o, &na);
    for (int64_t i = 0; i < na && i < cap; ++i) { if (x) x[i] = a[i].x; if (y) y[i] = a[i].y; }
    free(a);
    return na;
}

/* ------------------------------------------------------------------ realign_record's seeded map as the library runs it (K1, `k1_best_n` > 0)
 *
 * The statement the HIP path of sp_hla_realign_reads is held to, bit for bit, in seeded mode: seeding, chaining and the selection of the chains that get a
 * base-level alignment are omm_map's (collect_anchors, chain_anchors, set_parent, select_sub above: minimap2's published algorithm at the reference's
 * settings, src/util/mapping.rs:8-14); the base-level alignment of a selected chain is the library's: the unit-cost cell on the 64 diagonals around the
 * chain (256 when that finds nothing: osp_wfa_retry) and its two-piece affine re-score (osp_affine_local on the 64 diagonals around
 * the cell's alignment), which stands in for align_chain + finish_hit; the filters, the output order (peak score, descending), the
 * second set_parent / select_sub and the acceptance loop of realign_record (src/hla/realigner.rs:124-146) follow as in omm_map / the reference.
 * At most OMM_SEED_SEL selected chains are aligned (the best ranked; minimap2 has no such bound: a read with more than ten primary chains is not a read of one locus). */
int32_t omm_hla_k1_seeded(const omm_index* idx, const uint8_t* q, int32_t qlen, const omm_opts* o, omm_seed_hit* hits, int32_t* n_hits, int32_t* n_chains) {
    *n_hits = 0; if (n_chains) *n_chains = 0;
    if (qlen <= 0) return -1;
    int64_t na = 0;
    an_t* a = collect_anchors(idx, q, qlen, o, &na);
    an_t* v = NULL; int nr = 0;
    reg_t* regs = chain_anchors(a, na, qlen, o, &v, &nr);
    free(a);
    if (n_chains) *n_chains = nr;
    if (nr == 0) { free(regs); free(v); return -1; }
    set_parent(regs, nr, o->mask_level);
    nr = select_sub(regs, nr, o->pri_ratio, o->k * 2, o->best_n);
    if (nr > OMM_SEED_SEL) nr = OMM_SEED_SEL;
    uint8_t* qrev = NULL;
    omm_seed_hit tmp[OMM_SEED_SEL]; reg_t hr[OMM_SEED_SEL];
    const osp_affine_opts ao = { o->a, o->b, o->q, o->e, o->q2, o->e2, o->sc_ambi };
    int nh = 0;
    for (int r = 0; r < nr; ++r) {
        const uint8_t* qc = q;
        if (regs[r].rev) {
            if (!qrev) { qrev = (uint8_t*)malloc((size_t)qlen); for (int i = 0; i < qlen; ++i) { const uint8_t c = q[qlen - 1 - i]; qrev[i] = c > 3 ? c : (uint8_t)(3 - c); } }
            qc = qrev;
        }
        const an_t* first = &v[regs[r].as]; const an_t* last = &v[regs[r].as + regs[r].cnt - 1];
        const int d0 = (int32_t)(uint32_t)first->y - (int32_t)(uint32_t)first->x, d1 = (int32_t)(uint32_t)last->y - (int32_t)(uint32_t)last->x;   /* query position - target position of the outermost seeds */
        const int rid = regs[r].rid;
        const uint8_t* tseq = idx->codes + idx->off[rid];
        const int tlen = (int)(idx->off[rid + 1] - idx->off[rid]);
        omm_seed_hit h; memset(&h, 0, sizeof(h));
        h.rid = rid; h.rev = regs[r].rev; h.chain_score = regs[r].score; h.n_seeds = regs[r].cnt; h.t_len = tlen; h.sel_rank = r;
        h.diag = (d0 + d1) >> 1;
        int cap = (int)(0.03 * (double)tlen) + 1; if (cap > OSP_MAX_ED) cap = OSP_MAX_ED;          /* nm <= 0.03 * span <= 0.03 * allele length (realigner.rs:138-141) */
        osp_aln al;
        h.ok = osp_wfa_retry(tseq, tlen, qc, qlen, h.diag, cap, &al, NULL, NULL);
        if (h.ok) {
            h.cell_nm = al.nm; h.a_start = al.a_start; h.a_end = al.a_end; h.b_start = al.b_start; h.b_end = al.b_end;
            const int twice = (al.b_start - al.a_start) + (al.b_end - al.a_end);
            /* an alignment whose ends lie more than 32 diagonals apart crosses a long insertion / deletion (the cell found it on the wide band): its re-score runs on 256
             * diagonals -- minimap2 chains and aligns across such gaps (bw 500, max_gap 10000 above), 64 diagonals would clip the alignment at the gap */
            const int shift = (al.b_end - al.a_end) - (al.b_start - al.a_start);
            osp_affine_out af;
            osp_affine_local(tseq, tlen, qc, qlen, twice / 2, (shift > 32 || shift < -32) ? 256 : 64, &ao, &af);
            h.dp_max = af.score; h.nm = af.nm; h.t_start = af.t_start; h.t_end = af.t_end;
            if (h.rev) { h.q_start = qlen - af.q_end; h.q_end = qlen - af.q_start; } else { h.q_start = af.q_start; h.q_end = af.q_end; }
        }
        /* mm_filter_regs: too few seeds (never: chains have >= min_cnt), too few matching bases (implied by the peak score at a = 1), or a peak score below min_dp_max */
        if (h.ok && h.dp_max >= o->min_dp_max && h.dp_max > 0) {
            hr[nh] = regs[r]; hr[nh].qs = h.q_start; hr[nh].qe = h.q_end; hr[nh].rs = h.t_start; hr[nh].re = h.t_end; hr[nh].id = nh;
            tmp[nh++] = h;
        }
    }
    /* output order: by peak score, descending (stable); parents again on the aligned intervals; secondaries by chain score */
    int ord[OMM_SEED_SEL];
    for (int i = 0; i < nh; ++i) ord[i] = i;
    for (int i = 1; i < nh; ++i) { const int c = ord[i]; int j = i - 1; while (j >= 0 && tmp[ord[j]].dp_max < tmp[c].dp_max) { ord[j + 1] = ord[j]; --j; } ord[j + 1] = c; }
    reg_t hs[OMM_SEED_SEL];
    for (int i = 0; i < nh; ++i) hs[i] = hr[ord[i]];
    set_parent(hs, nh, o->mask_level);
    const int nk = nh ? select_sub(hs, nh, o->pri_ratio, o->k * 2, o->best_n) : 0;
    /* the acceptance loop of realign_record (src/hla/realigner.rs:124-146) over the mappings in output order */
    int pick = -1; double best = 1.0;                           /* custom_score(false) of MappingStats(read_len, read_len, 0) */
    for (int i = 0; i < nk; ++i) {
        hits[i] = tmp[hs[i].id]; hits[i].primary = hs[i].parent == i;
        const omm_seed_hit* h = &hits[i];
        const uint64_t tl = (uint64_t)h->t_len, um = tl - (uint64_t)(h->t_end - h->t_start), nm = (uint64_t)h->nm;
        if (osp_custom_score(tl, nm, um, 1) <= 0.5 && osp_custom_score(tl, nm, um, 0) <= 0.03 && osp_custom_score(tl, nm, um, 0) < best) { best = osp_custom_score(tl, nm, um, 0); pick = i; }
    }
    *n_hits = nk;
    free(qrev); free(regs); free(v);
    return pick;
}
