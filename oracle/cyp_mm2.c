/*
 * cyp_mm2.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY): the CYP2D6 alignment call sites of the reference in their own CALL PATTERN, on the
 * minimap2 restatement (oracle/mm2.c) instead of the library's alignment contract (oracle/align.c).
 *
 *   omm_cyp_find_base_type    find_base_type_in_sequence (src/cyp2d6/haplotyper.rs:142-315): the read is indexed once
 *                             (`standard_hifi_aligner().with_seq(search_sequence)`, :155-156), every template is mapped onto it in
 *                             full_allele() order (:175-183,193-249), EVERY mapping minimap2 returns is a candidate (best_n 5), reverse
 *                             mappings and those over max_ed_frac are dropped, then sort / collapse / filter as osp_cyp_find_base_type
 *   omm_cyp_weight_sequence   weight_sequence (src/cyp2d6/chaining.rs:28-103): the read segment is indexed, every allowed consensus mapped
 *   omm_cyp_place             the backbone placement of assign_haplotype (src/cyp2d6/haplotyper.rs:391-412): the backbone is indexed, the
 *                             sequence mapped; with several mappings the one with the longest block length
 *
 * Used by tests/cpu_port_cyp.py (the reference-call-pattern CPU port the GPU path is held to in tests/test_gpu_concordance.py, and bench.py's
 * cpu_baseline leg).  Like mm2.c it restates minimap2's published algorithm, not its binary.
 */
#include <stdlib.h>
#include <string.h>
#include "sp_oracle.h"
#include "cyp_oracle.h"
#include "mm2_oracle.h"

#define CM_MAX_HITS 16

static int cm_is_penalized(int t) { return t == OSP_DELETION || t == OSP_REP6 || t == OSP_REP7; }                 /* haplotyper.rs:185-191 */
static double cm_hit_score(const osp_region_hit* h, int penalize) { return osp_custom_score((uint64_t)h->seq_len, (uint64_t)h->nm, (uint64_t)h->unmapped, penalize); }

int omm_cyp_find_base_type(const uint8_t* seq, int seq_len, int n_templates, const uint8_t* const* tmpl, const int32_t* tmpl_len,
                           const int32_t* tmpl_type, double max_missing_frac, const omm_opts* o, osp_region_hit* out, int cap) {
    if (seq_len == 0) return 0;
    const double max_ed_frac = 0.05;
    const int64_t off[2] = { 0, seq_len };
    omm_index* idx = omm_index_build(seq, off, 1, o);
    if (!idx) return 0;
    int n_un = 0;
    osp_region_hit* un = (osp_region_hit*)malloc(sizeof(osp_region_hit) * (size_t)(n_templates * CM_MAX_HITS + 1));
    for (int t = 0; t < n_templates; ++t) {
        omm_hit hits[CM_MAX_HITS];
        const int n = omm_map(idx, tmpl[t], tmpl_len[t], o, hits, CM_MAX_HITS, NULL, 0);
        for (int k = 0; k < n; ++k) {
            osp_region_hit h;
            h.template_idx = t; h.start = hits[k].t_start; h.end = hits[k].t_end;
            h.seq_len = tmpl_len[t]; h.nm = hits[k].nm; h.unmapped = tmpl_len[t] - (hits[k].q_end - hits[k].q_start);
            h.clip_start = hits[k].q_start; h.clip_end = tmpl_len[t] - hits[k].q_end;
            if (cm_hit_score(&h, cm_is_penalized(tmpl_type[t])) > max_ed_frac) continue;             /* :228-232 */
            if (hits[k].rev) continue;                                                                 /* :233-237 */
            un[n_un++] = h;
        }
    }
    omm_index_free(idx);
    for (int i = 1; i < n_un; ++i) {                                                                  /* stable sort by (start, end), :252-255 */
        osp_region_hit x = un[i]; int j = i - 1;
        while (j >= 0 && (un[j].start > x.start || (un[j].start == x.start && un[j].end > x.end))) { un[j + 1] = un[j]; --j; }
        un[j + 1] = x;
    }
    int n_out = 0, have_cur = 0, n_coll = 0; osp_region_hit cur; memset(&cur, 0, sizeof cur);
    osp_region_hit* coll = (osp_region_hit*)malloc(sizeof(osp_region_hit) * (size_t)(n_un + 1));
    for (int i = 0; i < n_un; ++i) {                                                                  /* :260-296 */
        if (!have_cur) { cur = un[i]; have_cur = 1; continue; }
        if (osp_cyp_overlap_score(un[i].start, un[i].end, cur.start, cur.end) > 0.9) {
            const int star5 = cm_is_penalized(tmpl_type[un[i].template_idx]) || cm_is_penalized(tmpl_type[cur.template_idx]);
            const int up = tmpl_type[un[i].template_idx] == OSP_DELETION ? 1 : 0, cp = tmpl_type[cur.template_idx] == OSP_DELETION ? 1 : 0;   /* :897-902 */
            if ((cm_hit_score(&un[i], star5) < cm_hit_score(&cur, star5) && up >= cp) || up > cp) cur = un[i];
        } else { coll[n_coll++] = cur; cur = un[i]; }
    }
    if (have_cur) coll[n_coll++] = cur;
    for (int i = 0; i < n_coll; ++i) {
        if (cm_hit_score(&coll[i], 1) > max_missing_frac) continue;                                   /* :303-306 */
        if (n_out < cap) out[n_out] = coll[i];
        ++n_out;
    }
    free(un); free(coll);
    return n_out;
}

int omm_cyp_weight_sequence(const uint8_t* seq, int seq_len, int n_cons, const uint8_t* const* cons, const int32_t* cons_len,
                            const uint8_t* allowed, const omm_opts* o, uint64_t* out_ed, double* out_ov) {
    const double maximum_allowed_ed = 0.05;
    double min_ed_frac = 1.0;
    for (int c = 0; c < n_cons; ++c) { out_ed[c] = (uint64_t)seq_len; out_ov[c] = 0.0; }
    if (seq_len == 0) return 0;
    const int64_t off[2] = { 0, seq_len };
    omm_index* idx = omm_index_build(seq, off, 1, o);
    if (!idx) return 0;
    for (int c = 0; c < n_cons; ++c) {
        if (!allowed[c] || cons_len[c] == 0) continue;                                                /* :52-55 */
        omm_hit hits[CM_MAX_HITS];
        const int n = omm_map(idx, cons[c], cons_len[c], o, hits, CM_MAX_HITS, NULL, 0);
        for (int k = 0; k < n; ++k) {                                                                 /* (no strand test in the reference, :62-93) */
            const uint64_t nm = (uint64_t)hits[k].nm, unmapped = (uint64_t)(seq_len - (hits[k].t_end - hits[k].t_start));
            const uint64_t clipped_start = (uint64_t)hits[k].q_start, clipped_end = (uint64_t)(cons_len[c] - hits[k].q_end);
            const uint64_t match_score = nm + unmapped;
            const double overlap_score = 1.0 - (double)(clipped_start + clipped_end) / (double)cons_len[c];
            if (match_score < out_ed[c] || (match_score == out_ed[c] && overlap_score > out_ov[c])) {
                out_ed[c] = match_score; out_ov[c] = overlap_score;
                const double sc = osp_custom_score((uint64_t)seq_len, nm, unmapped, 1);
                if (sc < min_ed_frac) min_ed_frac = sc;
            }
        }
    }
    omm_index_free(idx);
    return min_ed_frac <= maximum_allowed_ed;
}

/* out6 = { q_start, q_end (on the sequence), t_start, t_end (on the backbone), nm, centre of the drift band (insertions +1, deletions -1 along the
 * alignment: (min + max) / 2, the quantity osp_cyp_variant_states derives from its own placement) }; returns 0 when nothing maps forward */
int omm_cyp_place(const uint8_t* seq, int seq_len, const uint8_t* backbone, int backbone_len, const omm_opts* o, int32_t* out6) {
    memset(out6, 0, sizeof(int32_t) * 6);
    if (seq_len == 0 || backbone_len == 0) return 0;
    omm_hit hits[CM_MAX_HITS];
    const int cap = 4 * (seq_len + backbone_len) + 64;
    uint32_t* pool = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap * CM_MAX_HITS);
    const int n = omm_map_pair(backbone, backbone_len, seq, seq_len, o, hits, CM_MAX_HITS, pool, cap * CM_MAX_HITS);
    int best = -1;
    for (int k = 0; k < n; ++k) if (best < 0 || hits[k].blen >= hits[best].blen) best = k;           /* (block_len, index).max(): the LAST of the longest (:404-410) */
    if (best < 0 || hits[best].rev) { free(pool); return 0; }
    int drift = 0, dmin = 0, dmax = 0;
    for (int x = 0; x < hits[best].n_cigar; ++x) {
        const uint32_t c = pool[hits[best].cigar_off + x]; const int op = (int)(c & 15u), len = (int)(c >> 4);
        if (op == 1) drift += len; else if (op == 2) drift -= len;
        if (drift < dmin) dmin = drift;
        if (drift > dmax) dmax = drift;
    }
    out6[0] = hits[best].q_start; out6[1] = hits[best].q_end; out6[2] = hits[best].t_start; out6[3] = hits[best].t_end; out6[4] = hits[best].nm;
    out6[5] = (dmin + dmax) / 2;
    free(pool);
    return 1;
}
