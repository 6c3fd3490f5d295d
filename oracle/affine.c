/*
 * affine.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY): the two-piece affine re-score of an alignment the library found.
 *
 * The reference's (nm, start, end) are minimap2's (src/util/mapping.rs:8-14): an alignment through the chain's seeds that is global between them
 * and extended from the outermost ones to the best-scoring cell, under match a, mismatch b, gaps min(q + l e, q2 + l e2), ambiguous bases -sc_ambi
 * (oracle/mm2.c restates it).  Such an alignment is the best LOCAL alignment among those through its seeds.  The library's anchor + unit-cost cell
 *  finds the diagonal and the extent of the pair; this routine re-scores it the reference's way: the banded Smith-Waterman optimum
 * under the two-piece affine scores on the 64 (or 256: pairs with long gaps, the other paralog of a CYP2D6 template) diagonals around the cell's own diagonal, with the forward decisions of the restatement's DP
 * (oracle/mm2.c dp_align: a gap is continued only when that is strictly better than opening one; H takes the diagonal, then E, F, E2, F2 on ties) and
 * its end rules (the first best-scoring cell by anti-diagonal, then by row, ends the alignment; a prefix that does not pay is not part of it: the
 * shortest extension among equal ones).  Every state carries the mismatch + gap + ambiguous bases of its path and where that path began, so the
 * numbers come out of one forward pass, without a traceback.  What it cannot see: minimap2's choice of seeds and chains, its z-drop (400) and
 * gaps longer than the band.  tests/test_oracle_affine.py measures how often it equals omm_map_pair on the audited pair classes.
 *
 * The HIP kernel (sp_affine.hip) is bit-exact against THIS statement.
 */
#include <stdlib.h>
#include <string.h>
#include "sp_oracle.h"

#define AF_MAXBAND 256
#define AF_NEG (-(1 << 28))

typedef struct { int32_t s, nm, si, sj; } af_state;

void osp_affine_local(const uint8_t* T, int tlen, const uint8_t* Q, int qlen, int k0, int band, const osp_affine_opts* o, osp_affine_out* out) {
    memset(out, 0, sizeof(*out));
    if (tlen <= 0 || qlen <= 0) return;
    const int AF_BAND = band == 256 ? 256 : 64;              /* diagonals k0 - band / 2 .. k0 + band / 2 - 1 */
    const int klo = k0 - AF_BAND / 2;                       /* lane l holds diagonal k = j - i = klo + l */
    af_state H[AF_MAXBAND + 2], E1[AF_MAXBAND + 2], E2[AF_MAXBAND + 2], Hn[AF_MAXBAND + 2], E1n[AF_MAXBAND + 2], E2n[AF_MAXBAND + 2];
    const af_state none = { AF_NEG, 0, 0, 0 };
    for (int l = 0; l < AF_BAND + 2; ++l) H[l] = E1[l] = E2[l] = none;
    int best = 0, bi = -1, bj = -1, bnm = 0, bsi = 0, bsj = 0;
    /* rows that can hold a cell of the band */
    int i_lo = -(klo + AF_BAND - 1); if (i_lo < 0) i_lo = 0;
    int i_hi = qlen - 1 - klo; if (i_hi > tlen - 1) i_hi = tlen - 1;
    for (int i = i_lo; i <= i_hi; ++i) {
        af_state F1 = none, F2 = none, left = none;         /* F of the cell to the left, H of the cell to the left */
        const int ct = T[i];
        for (int l = 0; l < AF_BAND; ++l) {
            const int j = i + klo + l;
            Hn[l] = E1n[l] = E2n[l] = none;
            if (j < 0 || j >= qlen) { F1 = F2 = left = none; continue; }
            /* E: from the cell above = lane l + 1 of the row before */
            af_state e1 = none, e2 = none;
            if (l + 1 < AF_BAND) {
                const af_state hu = H[l + 1], eu = E1[l + 1], eu2 = E2[l + 1];
                if (hu.s > AF_NEG || eu.s > AF_NEG) {
                    const int eo = hu.s > AF_NEG ? hu.s - o->q : AF_NEG;
                    if (eu.s > eo) { e1 = eu; e1.s = eu.s - o->e; } else { e1 = hu; e1.s = eo - o->e; }
                    e1.nm += 1;
                }
                if (hu.s > AF_NEG || eu2.s > AF_NEG) {
                    const int eo2 = hu.s > AF_NEG ? hu.s - o->q2 : AF_NEG;
                    if (eu2.s > eo2) { e2 = eu2; e2.s = eu2.s - o->e2; } else { e2 = hu; e2.s = eo2 - o->e2; }
                    e2.nm += 1;
                }
            }
            /* F: from the cell to the left = lane l - 1 of this row */
            af_state f1 = none, f2 = none;
            if (left.s > AF_NEG || F1.s > AF_NEG) {
                const int fo = left.s > AF_NEG ? left.s - o->q : AF_NEG;
                if (F1.s > fo) { f1 = F1; f1.s = F1.s - o->e; } else { f1 = left; f1.s = fo - o->e; }
                f1.nm += 1;
            }
            if (left.s > AF_NEG || F2.s > AF_NEG) {
                const int fo2 = left.s > AF_NEG ? left.s - o->q2 : AF_NEG;
                if (F2.s > fo2) { f2 = F2; f2.s = F2.s - o->e2; } else { f2 = left; f2.s = fo2 - o->e2; }
                f2.nm += 1;
            }
            /* the diagonal: lane l of the row before; a cell nothing worth keeping leads to starts an alignment of its own */
            af_state h = H[l];
            const int cq = Q[j];
            const int ambi = ct > 3 || cq > 3;
            const int sub = ambi ? -o->sc_ambi : (ct == cq ? o->a : -o->b);
            if (h.s <= 0) { h.s = 0; h.nm = 0; h.si = i; h.sj = j; }
            h.s += sub; h.nm += (ambi || ct != cq) ? 1 : 0;
            if (e1.s > h.s) h = e1;
            if (f1.s > h.s) h = f1;
            if (e2.s > h.s) h = e2;
            if (f2.s > h.s) h = f2;
            if (h.s <= 0) { h.s = 0; h.nm = 0; h.si = i; h.sj = j; }      /* (nothing ends here; a successor starts afresh) */
            Hn[l] = h; E1n[l] = e1; E2n[l] = e2;
            F1 = f1; F2 = f2; left = h;
            if (h.s > best || (h.s == best && h.s > 0 && (i + j < bi + bj || (i + j == bi + bj && i < bi)))) { best = h.s; bi = i; bj = j; bnm = h.nm; bsi = h.si; bsj = h.sj; }
        }
        memcpy(H, Hn, sizeof H); memcpy(E1, E1n, sizeof E1); memcpy(E2, E2n, sizeof E2);
    }
    if (best <= 0) return;
    out->score = best; out->nm = bnm; out->t_start = bsi; out->t_end = bi + 1; out->q_start = bsj; out->q_end = bj + 1;
}
