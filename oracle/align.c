/*
 * oracle/align.c -- CPU ORACLE (test infrastructure, never shipped, never linked by the product).
 *
 * The reference delegates every base-level alignment to minimap2 (src/util/mapping.rs:8-14 and the
 * call sites listed in SURVEY.md 8(a)); minimap2 is not under /root/reference, so this file restates the
 * alignment CONTRACT the build defines in DESIGN.md section 3 (parity unpinned beyond the reference's
 * pinned cases, see sp_oracle.h):
 *
 *   osp_anchor : exact 16-mer diagonal voting (the seed step)
 *   osp_wfa    : edit-distance wavefront alignment (Marco-Sola et al. 2021, edit-distance case) over a
 *                fixed band of 64 diagonals, ends-free at both ends, with deterministic tie rules.
 *
 * The code is deliberately scalar and literal; the HIP kernels must reproduce it bit for bit.
 */
#include "sp_oracle.h"
#include <stdlib.h>
#include <string.h>

void osp_encode(const char* ascii, size_t n, uint8_t* codes) {
    for (size_t i = 0; i < n; ++i) {
        switch (ascii[i]) {
            case 'A': case 'a': codes[i] = 0; break;
            case 'C': case 'c': codes[i] = 1; break;
            case 'G': case 'g': codes[i] = 2; break;
            case 'T': case 't': codes[i] = 3; break;
            default: codes[i] = 4; break;
        }
    }
}

/* ---------------------------------------------------------------- anchor */

typedef struct { uint32_t code; int32_t pos; } kmer_t;

static int kmer_cmp(const void* x, const void* y) {
    const kmer_t* a = (const kmer_t*)x; const kmer_t* b = (const kmer_t*)y;
    if (a->code != b->code) return a->code < b->code ? -1 : 1;
    return a->pos < b->pos ? -1 : (a->pos > b->pos);
}

/* all valid (N-free) k-mers of s, in position order; returns count */
static int list_kmers(const uint8_t* s, int n, kmer_t* out) {
    int cnt = 0, valid = 0; uint32_t code = 0;
    for (int i = 0; i < n; ++i) {
        if (s[i] > 3) { valid = 0; code = 0; continue; }
        code = (code << 2) | s[i];             /* k = 16 -> exactly 32 bits, natural wrap */
        if (++valid >= OSP_KMER) { out[cnt].code = code; out[cnt].pos = i - OSP_KMER + 1; ++cnt; }
    }
    return cnt;
}

/* top-K anchors: repeatedly take the diagonal with most votes (ties: smallest d), then suppress every bin within
 * +-OSP_PEAK_SUPPRESS of it.  Returns the number of peaks with >= 1 vote (at most k). */
int osp_anchor_topk(const uint8_t* A, int m, const uint8_t* B, int n, int k, int* diags, int* votes_out) {
    for (int i = 0; i < k; ++i) { diags[i] = 0; votes_out[i] = 0; }
    if (m < OSP_KMER || n < OSP_KMER) return 0;
    kmer_t* ta = (kmer_t*)malloc(sizeof(kmer_t) * (size_t)m);
    kmer_t* tb = (kmer_t*)malloc(sizeof(kmer_t) * (size_t)n);
    int na = list_kmers(A, m, ta);
    int nb = list_kmers(B, n, tb);
    qsort(ta, (size_t)na, sizeof(kmer_t), kmer_cmp);
    /* votes indexed by d + m (d = j - p in (-m, n)) */
    int nbins = m + n + 1;
    int32_t* votes = (int32_t*)calloc((size_t)nbins, sizeof(int32_t));
    for (int x = 0; x < nb; ++x) {
        uint32_t code = tb[x].code;
        int lo = 0, hi = na;                     /* first index with ta.code >= code */
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ta[mid].code < code) lo = mid + 1; else hi = mid; }
        int e = lo; while (e < na && ta[e].code == code) ++e;
        int occ = e - lo;
        if (occ == 0 || occ > OSP_MAXOCC) continue;
        for (int y = lo; y < e; ++y) votes[tb[x].pos - ta[y].pos + m]++;
    }
    int found = 0;
    for (int round = 0; round < k; ++round) {
        int best = 0, bestb = -1;
        for (int b = 0; b < nbins; ++b) if (votes[b] > best) { best = votes[b]; bestb = b; }  /* ties: smallest d */
        if (bestb < 0) break;
        /* a long indel splits the votes over two diagonals: centre the band between the outermost diagonals within
         * +-OSP_PEAK_SPREAD of the peak that still hold >= max(2, peak/8) votes */
        int thr = best / 8 > 2 ? best / 8 : 2;
        int lo = bestb, hi = bestb;
        for (int b = bestb - OSP_PEAK_SPREAD; b <= bestb + OSP_PEAK_SPREAD; ++b) {
            if (b < 0 || b >= nbins || votes[b] < thr) continue;
            if (b < lo) lo = b;
            if (b > hi) hi = b;
        }
        diags[found] = ((lo + hi) >> 1) - m; votes_out[found] = best; ++found;
        int slo = bestb - OSP_PEAK_SUPPRESS, shi = bestb + OSP_PEAK_SUPPRESS;
        if (slo < 0) slo = 0;
        if (shi > nbins - 1) shi = nbins - 1;
        for (int b = slo; b <= shi; ++b) votes[b] = 0;
    }
    free(ta); free(tb); free(votes);
    return found;
}

int osp_anchor(const uint8_t* A, int m, const uint8_t* B, int n, int* diag) {
    int v = 0;
    osp_anchor_topk(A, m, B, n, 1, diag, &v);
    return v;
}

/* ---------------------------------------------------------------- wavefront alignment */

static inline int extend(const uint8_t* A, int m, const uint8_t* B, int n, int i, int k) {
    int j = i + k;
    while (i < m && j < n && A[i] == B[j] && A[i] < 4) { ++i; ++j; }
    return i;
}

/* candidate evaluation shared by the forward pass and the traceback: priority X > D > I on equal reach */
static inline int pick(const int32_t* H, int l, int band, int* src, int* type) {
    int best = OSP_NEG; *src = l; *type = (int)OSP_EV_X;
    if (H[l] >= 0) best = H[l] + 1;
    if (l > 0 && H[l - 1] >= 0 && H[l - 1] > best) { best = H[l - 1]; *src = l - 1; *type = (int)OSP_EV_D; }
    if (l < band - 1 && H[l + 1] >= 0 && H[l + 1] + 1 > best) { best = H[l + 1] + 1; *src = l + 1; *type = (int)OSP_EV_I; }
    return best;
}

/* the cell on `band` diagonals (64: the contract every cell runs on; 256: the retry of a cell that found no alignment, see osp_wfa_retry) */
int osp_wfa_band(const uint8_t* A, int m, const uint8_t* B, int n, int diag, int max_ed, int band,
                 osp_aln* out, uint32_t* events, int* n_events) {
    memset(out, 0, sizeof(*out));
    out->a_len = m; out->b_len = n;
    if (n_events) *n_events = 0;
    if (m <= 0 || n <= 0 || max_ed < 0) return 0;
    const int kbase = diag - band / 2;                     /* lane l <-> diagonal kbase + l */
    int32_t* hist = (int32_t*)malloc(sizeof(int32_t) * (size_t)band * (size_t)(max_ed + 1));
    int32_t* H = hist;
    for (int l = 0; l < band; ++l) {
        int k = kbase + l;
        int i0 = k < 0 ? -k : 0, j0 = i0 + k;
        H[l] = (i0 < m && j0 < n) ? extend(A, m, B, n, i0, k) : OSP_NEG;
    }
    int s = 0, end_lane = -1;
    for (;;) {
        /* termination: any lane on the last row / last column. longest path, then most central, then lowest lane */
        int64_t best_key = -1;
        for (int l = 0; l < band; ++l) {
            if (H[l] < 0) continue;
            int i = H[l], j = i + kbase + l;
            if (i == m || j == n) {
                int c = l - band / 2; if (c < 0) c = -c;
                int64_t key = (int64_t)(i + j) * (2 * (int64_t)band * band) + (int64_t)(band - c) * band + (band - 1 - l);
                if (key > best_key) { best_key = key; end_lane = l; }
            }
        }
        if (end_lane >= 0) break;
        if (s == max_ed) { free(hist); return 0; }
        int32_t* Hn = hist + (size_t)(s + 1) * band;
        for (int l = 0; l < band; ++l) {
            int src, type; int b = pick(H, l, band, &src, &type);
            Hn[l] = b >= 0 ? extend(A, m, B, n, b, kbase + l) : OSP_NEG;
        }
        H = Hn; ++s;
    }
    /* traceback */
    int l = end_lane, i = H[l];
    out->ok = 1; out->nm = s;
    out->a_end = i; out->b_end = i + kbase + l;
    int ne = 0;
    for (int t = s; t > 0; --t) {
        const int32_t* Hp = hist + (size_t)(t - 1) * band;
        int src, type; int b = pick(Hp, l, band, &src, &type);
        int k = kbase + l; uint32_t bpos;
        if (type == (int)OSP_EV_X)      { bpos = (uint32_t)(b - 1 + k); i = b - 1; }
        else if (type == (int)OSP_EV_D) { bpos = (uint32_t)(b + k - 1); i = b; }
        else                            { bpos = (uint32_t)(b + k);     i = b - 1; }
        if (events) events[ne] = ((uint32_t)type << 30) | bpos;
        ++ne; l = src;
    }
    {
        int k = kbase + l; int i0 = k < 0 ? -k : 0;
        out->a_start = i0; out->b_start = i0 + k;
    }
    if (events) for (int x = 0; x < ne / 2; ++x) { uint32_t t = events[x]; events[x] = events[ne - 1 - x]; events[ne - 1 - x] = t; }
    if (n_events) *n_events = ne;
    free(hist);
    return 1;
}

int osp_wfa(const uint8_t* A, int m, const uint8_t* B, int n, int diag, int max_ed,
            osp_aln* out, uint32_t* events, int* n_events) {
    return osp_wfa_band(A, m, B, n, diag, max_ed, OSP_BAND, out, events, n_events);
}

/* what the library's generic cell launcher does everywhere but in K1 and K3: a cell that finds no alignment on 64 diagonals within
 * max_ed is run again on 256 diagonals around the same anchor (an insertion / deletion of 40-120 bases next to the anchor) */
int osp_wfa_retry(const uint8_t* A, int m, const uint8_t* B, int n, int diag, int max_ed,
                  osp_aln* out, uint32_t* events, int* n_events) {
    if (osp_wfa_band(A, m, B, n, diag, max_ed, OSP_BAND, out, events, n_events)) return 1;
    return osp_wfa_band(A, m, B, n, diag, max_ed, OSP_WIDE_BAND, out, events, n_events);
}

/* the stricter rule of the launcher's few-cell callers (sp_align_batch, allele / consensus placements): the 256-diagonal run is also
 * made when the 64-diagonal alignment needed more than half a band of edits (a path through a long insertion / deletion may be
 * cheaper than the mismatches the narrow band paid); the wide result is taken when it has strictly fewer edits */
int osp_wfa_retry2(const uint8_t* A, int m, const uint8_t* B, int n, int diag, int max_ed,
                   osp_aln* out, uint32_t* events, int* n_events) {
    const int ok = osp_wfa_band(A, m, B, n, diag, max_ed, OSP_BAND, out, events, n_events);
    if (ok && out->nm <= OSP_BAND / 2) return 1;
    osp_aln wide; int ne = 0;
    uint32_t* ev = events ? (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(max_ed + 1)) : NULL;
    const int wok = osp_wfa_band(A, m, B, n, diag, max_ed, OSP_WIDE_BAND, &wide, ev, &ne);
    if (wok && (!ok || wide.nm < out->nm)) {
        *out = wide;
        if (events) memcpy(events, ev, sizeof(uint32_t) * (size_t)ne);
        if (n_events) *n_events = ne;
        free(ev);
        return 1;
    }
    free(ev);
    return ok;
}

int osp_events_to_cigar(const osp_aln* aln, const uint32_t* events, int n_events, uint32_t* cigar, int cap) {
    int nc = 0; int j = aln->b_start;
#define PUSH(len, op) do { if ((len) > 0) { if (nc > 0 && (cigar[nc-1] & 15u) == (uint32_t)(op)) cigar[nc-1] += (uint32_t)(len) << 4; \
        else { if (nc >= cap) return -1; cigar[nc++] = ((uint32_t)(len) << 4) | (uint32_t)(op); } } } while (0)
    for (int e = 0; e < n_events; ++e) {
        uint32_t type = events[e] >> 30; int bpos = (int)(events[e] & 0x3FFFFFFFu);
        PUSH(bpos - j, 7); j = bpos;
        if (type == OSP_EV_X) { PUSH(1, 8); j += 1; }
        else if (type == OSP_EV_D) { PUSH(1, 2); j += 1; }
        else { PUSH(1, 1); }
    }
    PUSH(aln->b_end - j, 7);
#undef PUSH
    return nc;
}
