/*
 * oracle/hla.c -- CPU ORACLE (test infrastructure only): the HLA scoring logic of the reference,
 * restated function by function.  Citations are /root/reference paths.
 */
#include "sp_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

/* src/data_types/mapping.rs:191-195  MappingScore::score_value */
double osp_score_value(uint64_t len, uint64_t nm, uint64_t unmapped) {
    double numerator = (double)(nm + unmapped);
    if (numerator < 0.1) numerator = 0.1;
    return numerator / (double)len;
}

/* src/data_types/mapping.rs:60-84  MappingStats::custom_score */
double osp_custom_score(uint64_t seq_len, uint64_t nm, uint64_t unmapped, int penalize_unmapped) {
    if (penalize_unmapped) return osp_score_value(seq_len, nm, unmapped);
    return osp_score_value(seq_len - unmapped, nm, 0);
}

/* src/util/mapping.rs:22-57 */
int osp_select_best_mapping(const osp_mapping* maps, int n, int unmapped_from_target, int penalize_unmapped,
                            int64_t base_length_override, uint64_t stats3[3]) {
    uint64_t b_len = base_length_override >= 0 ? (uint64_t)base_length_override : 1;
    uint64_t best[3] = { b_len, b_len, 0 };
    int best_idx = -1;
    for (int x = 0; x < n; ++x) {
        const osp_mapping* m = &maps[x];
        uint64_t bl, um;
        if (unmapped_from_target) {
            bl = base_length_override >= 0 ? (uint64_t)base_length_override : (uint64_t)m->target_len;
            um = bl - (uint64_t)(m->target_end - m->target_start);
        } else {
            bl = base_length_override >= 0 ? (uint64_t)base_length_override : (uint64_t)m->query_len;
            um = bl - (uint64_t)(m->query_end - m->query_start);
        }
        if (osp_custom_score(bl, (uint64_t)m->nm, um, penalize_unmapped) <
            osp_custom_score(best[0], best[1], best[2], penalize_unmapped)) {
            best[0] = bl; best[1] = (uint64_t)m->nm; best[2] = um; best_idx = x;
        }
    }
    stats3[0] = best[0]; stats3[1] = best[1]; stats3[2] = best[2];
    return best_idx;
}

/* src/hla/processed_match.rs:210-263 */
int osp_process_mm_cigar(const uint32_t* cigar_len, const uint8_t* cigar_op, int n_ops,
                         uint64_t target_offset, uint64_t target_len, uint64_t clip_start, uint64_t clip_end,
                         uint64_t* out) {
    uint64_t zero_padding = target_offset > clip_start ? target_offset - clip_start : 0;  /* saturating_sub */
    uint64_t nm_padding = target_offset - zero_padding;
    uint64_t len = 0, current_nm = 0;
    for (uint64_t i = 0; i < zero_padding + 1; ++i) out[len++] = 0;
    for (uint64_t i = 0; i < nm_padding; ++i) { current_nm += 1; out[len++] = current_nm; }
    for (int c = 0; c < n_ops; ++c) {
        uint32_t length = cigar_len[c];
        switch (cigar_op[c]) {
            case 1: current_nm += length; break;                                   /* I */
            case 2: case 8:                                                        /* D | X */
                for (uint32_t i = 0; i < length; ++i) { current_nm += 1; if (len > target_len) return -2; out[len++] = current_nm; }
                break;
            case 7:                                                                /* = */
                for (uint32_t i = 0; i < length; ++i) { if (len > target_len) return -2; out[len++] = current_nm; }
                break;
            default: return -1;
        }
    }
    uint64_t missing_values = target_len + 1 - len;
    uint64_t nm_extension = clip_end < missing_values ? clip_end : missing_values;
    for (uint64_t i = 0; i < nm_extension; ++i) { current_nm += 1; out[len++] = current_nm; }
    uint64_t zp = missing_values - nm_extension;
    for (uint64_t i = 0; i < zp; ++i) out[len++] = current_nm;
    return 0;
}

/* src/hla/mapping.rs:44-61 + derive(PartialOrd) :111 -- lexicographic (cdna, dna), absent level = 1.0 */
static int hla_score_less(const osp_hla_level l[2], const osp_hla_level r[2]) {
    double ls[2], rs[2];
    for (int i = 0; i < 2; ++i) {
        ls[i] = l[i].present ? osp_custom_score((uint64_t)l[i].len, (uint64_t)l[i].nm, (uint64_t)l[i].unmapped, 1) : 1.0;
        rs[i] = r[i].present ? osp_custom_score((uint64_t)r[i].len, (uint64_t)r[i].nm, (uint64_t)r[i].unmapped, 1) : 1.0;
    }
    if (ls[0] < rs[0]) return 1;
    if (ls[0] > rs[0]) return 0;
    return ls[1] < rs[1];
}

/* src/hla/processed_match.rs:103-184 */
int osp_is_better_match(const osp_hla_level lhs[2], const osp_hla_level rhs[2]) {
    for (int i = 0; i < 2; ++i) {
        if (lhs[i].present && rhs[i].present) {
            int os = lhs[i].range_start > rhs[i].range_start ? lhs[i].range_start : rhs[i].range_start;
            int oe = lhs[i].range_end < rhs[i].range_end ? lhs[i].range_end : rhs[i].range_end;
            uint64_t lnm = 0, rnm = 0;
            if (os < oe) { lnm = lhs[i].pc[oe] - lhs[i].pc[os]; rnm = rhs[i].pc[oe] - rhs[i].pc[os]; }
            if (lnm < rnm) return 1;
            if (lnm > rnm) return 0;
        } else if (!lhs[i].present && !rhs[i].present) {
            /* both absent, iterate */
        } else if (lhs[i].present) {
            return 1;
        } else {
            return 0;
        }
    }
    return hla_score_less(lhs, rhs);
}

/* One (allele, level) cell of score_read: minimap2 call + Forward filter + select_best_mapping(query-based,
 * penalised) + HlaProcessedMatch::add_mapping, on top of the alignment contract.
 * src/hla/caller.rs:1433-1462, src/hla/processed_match.rs:53-100 */
static void score_level(const uint8_t* allele, int alen, const uint8_t* cons, int clen, int diag, int max_ed,
                        osp_hla_level* lv, uint64_t** pc_store, osp_aln* aln_out) {
    memset(lv, 0, sizeof(*lv));
    *pc_store = NULL;
    osp_aln aln; memset(&aln, 0, sizeof(aln));
    if (aln_out) *aln_out = aln;
    if (!allele || alen <= 0 || diag == INT_MIN) return;
    uint32_t* ev = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(max_ed + 1));
    int ne = 0;
    if (!osp_wfa_retry(allele, alen, cons, clen, diag, max_ed, &aln, ev, &ne)) { free(ev); return; }
    /* select_best_mapping with a single candidate: must beat the (1,1,0) default => score < 1.0 */
    uint64_t unmapped = (uint64_t)(alen - (aln.a_end - aln.a_start));
    if (!(osp_custom_score((uint64_t)alen, (uint64_t)aln.nm, unmapped, 1) < osp_custom_score(1, 1, 0, 1))) { free(ev); return; }
    if (aln_out) *aln_out = aln;
    int cap = 2 * ne + 2;
    uint32_t* cg = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap);
    int nc = osp_events_to_cigar(&aln, ev, ne, cg, cap);
    uint32_t* cl = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(nc + 1));
    uint8_t* co = (uint8_t*)malloc((size_t)(nc + 1));
    for (int i = 0; i < nc; ++i) { cl[i] = cg[i] >> 4; co[i] = (uint8_t)(cg[i] & 15u); }
    uint64_t clip_start = (uint64_t)aln.a_start, clip_end = (uint64_t)(alen - aln.a_end);
    uint64_t* pc = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(clen + 1));
    osp_process_mm_cigar(cl, co, nc, (uint64_t)aln.b_start, (uint64_t)clen, clip_start, clip_end, pc);
    uint64_t t_off = (uint64_t)aln.b_start;
    uint64_t pc_start = t_off > clip_start ? t_off - clip_start : 0;
    uint64_t rem = (uint64_t)(clen - aln.b_end);
    uint64_t pc_end = (uint64_t)aln.b_end + (clip_end < rem ? clip_end : rem);
    lv->present = 1; lv->range_start = (int32_t)pc_start; lv->range_end = (int32_t)pc_end;
    lv->len = alen; lv->nm = aln.nm; lv->unmapped = (int32_t)unmapped; lv->pc = pc;
    *pc_store = pc;
    free(ev); free(cg); free(cl); free(co);
}

/* src/hla/caller.rs:1411-1510 */
int osp_hla_score_read(const osp_hla_score_problem* p, int64_t* stats, osp_aln* alns) {
    osp_hla_level best[2]; memset(best, 0, sizeof(best));       /* worst_match: no level present, ranges 0..0 */
    uint64_t* best_pc[2] = { NULL, NULL };
    int best_idx = -1;
    for (int a = 0; a < p->n_alleles; ++a) {
        osp_hla_level cur[2]; uint64_t* cur_pc[2];
        for (int lv = 0; lv < 2; ++lv) {
            const uint8_t* seq = p->seq[lv] ? p->seq[lv][a] : NULL;
            int slen = p->seq_len[lv] ? p->seq_len[lv][a] : 0;
            int dg = p->diag[lv] ? p->diag[lv][a] : INT_MIN;
            score_level(seq, slen, p->cons[lv], p->cons_len[lv], dg, p->max_ed, &cur[lv], &cur_pc[lv],
                        alns ? &alns[a * 2 + lv] : NULL);
            int64_t* st = stats + ((size_t)a * 2 + (size_t)lv) * 3;
            if (cur[lv].present) { st[0] = cur[lv].len; st[1] = cur[lv].nm; st[2] = cur[lv].unmapped; }
            else { st[0] = -1; st[1] = -1; st[2] = -1; }
        }
        if (osp_is_better_match(cur, best)) {
            for (int lv = 0; lv < 2; ++lv) { free(best_pc[lv]); best_pc[lv] = cur_pc[lv]; best[lv] = cur[lv]; }
            best_idx = a;
        } else {
            for (int lv = 0; lv < 2; ++lv) free(cur_pc[lv]);
        }
    }
    for (int lv = 0; lv < 2; ++lv) free(best_pc[lv]);
    return best_idx;
}

/* src/hla/realigner.rs:124-146 : read (B) against every allele (A = minimap2 target) */
int osp_hla_pick_allele(const osp_aln* alns, int n, int read_len) {
    uint64_t best[3] = { (uint64_t)read_len, (uint64_t)read_len, 0 };
    int best_idx = -1;
    const double max_unmapped_frac = 0.5, max_ed_frac = 0.03;
    for (int a = 0; a < n; ++a) {
        if (!alns[a].ok) continue;
        uint64_t tlen = (uint64_t)alns[a].a_len;
        uint64_t unmapped = tlen - (uint64_t)(alns[a].a_end - alns[a].a_start);
        uint64_t nm = (uint64_t)alns[a].nm;
        if (osp_custom_score(tlen, nm, unmapped, 1) <= max_unmapped_frac &&
            osp_custom_score(tlen, nm, unmapped, 0) <= max_ed_frac &&
            osp_custom_score(tlen, nm, unmapped, 0) < osp_custom_score(best[0], best[1], best[2], 0)) {
            best[0] = tlen; best[1] = nm; best[2] = unmapped; best_idx = a;
        }
    }
    return best_idx;
}

/* src/hla/caller.rs:1225-1247 */
int osp_is_passing_dual(uint64_t counts1, uint64_t counts2, double min_consensus_fraction, double expected_maf,
                        double min_cdf, double* maf, double* cdf) {
    uint64_t total = counts1 + counts2;
    uint64_t minor = counts1 < counts2 ? counts1 : counts2;
    double m = (double)minor / (double)total;
    double c = osp_binomial_cdf(expected_maf, total, minor);
    if (maf) *maf = m;
    if (cdf) *cdf = c;
    return m >= min_consensus_fraction && c >= min_cdf;
}

/* src/hla/caller.rs:1583-1653 */
int osp_is_hemizygous_better(const int64_t* s1, const int64_t* s2, const uint8_t* is_c1, int n, int is_dual,
                             uint64_t dual_max_ed_delta, int has_norm, double normalized_coverage,
                             double* haploid_cost_out, double* diploid_cost_out) {
    uint64_t read_count = (uint64_t)n;
    uint64_t min_ed = 0;
    if (is_dual) {
        uint64_t c1_cost = 0, c2_cost = 0;
        for (int i = 0; i < n; ++i) {
            uint64_t o1 = s1[i] >= 0 ? (uint64_t)s1[i] : ((s2[i] >= 0 ? (uint64_t)s2[i] : 0) + dual_max_ed_delta);
            uint64_t o2 = s2[i] >= 0 ? (uint64_t)s2[i] : ((s1[i] >= 0 ? (uint64_t)s1[i] : 0) + dual_max_ed_delta);
            uint64_t mn = o1 < o2 ? o1 : o2;
            c1_cost += o1 - mn; c2_cost += o2 - mn;
        }
        min_ed = c1_cost < c2_cost ? c1_cost : c2_cost;
    }
    const double ln_ed_penalty = 2.0;
    double haploid_ed_cost = ln_ed_penalty * (double)min_ed;
    double nc_hap = has_norm ? normalized_coverage : (double)read_count;
    double nc_dev = nc_hap * 0.1;
    double haploid_norm_cost = fabs(osp_normal_ln_pdf(nc_hap, nc_dev, (double)read_count));
    double haploid_cost = haploid_ed_cost + haploid_norm_cost;

    uint64_t obs1 = 0;
    for (int i = 0; i < n; ++i) if (is_c1[i]) ++obs1;
    const double diploid_balance_penalty = 2.0;
    double diploid_balance_cost = is_dual ? diploid_balance_penalty * fabs(osp_binomial_ln_pmf(0.5, read_count, obs1)) : 0.0;
    double nc_dip = 2.0 * (has_norm ? normalized_coverage : (double)read_count);
    double diploid_norm_cost = fabs(osp_normal_ln_pdf(nc_dip, nc_dev, (double)read_count));
    double diploid_cost = diploid_balance_cost + diploid_norm_cost;
    if (haploid_cost_out) *haploid_cost_out = haploid_cost;
    if (diploid_cost_out) *diploid_cost_out = diploid_cost;
    return haploid_cost < diploid_cost;
}

/* Whole K1 search of one read on the CPU: anchors against every gene reference, the gene filter, one banded
 * alignment per candidate allele and the acceptance loop (src/hla/realigner.rs:116-146 on top of the alignment
 * contract).  Used by the tests and as bench.py's cpu_baseline ("port").  off[a] = allele_pos - ref_pos or INT_MIN. */
int osp_hla_k1_read(const uint8_t* read, int rlen, int n_genes, const uint8_t* const* refs, const int32_t* ref_len,
                    int n_alleles, const uint8_t* const* alleles, const int32_t* allele_len, const int32_t* gene_of,
                    const int32_t* off, uint32_t* cells, int64_t* n_cells_run) {
    int32_t* d_rg = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_genes);
    int32_t* v_rg = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_genes);
    int vmax = 0;
    for (int g = 0; g < n_genes; ++g) {
        int d = 0; v_rg[g] = osp_anchor(refs[g], ref_len[g], read, rlen, &d); d_rg[g] = d;
        if (v_rg[g] > vmax) vmax = v_rg[g];
    }
    int vmin = vmax / 10 > 16 ? vmax / 10 : 16;
    osp_aln* alns = (osp_aln*)calloc((size_t)n_alleles, sizeof(osp_aln));
    int64_t run = 0;
    for (int a = 0; a < n_alleles; ++a) {
        if (cells) cells[a] = 0xFFFFFFFFu;
        if (allele_len[a] <= 0 || off[a] == INT_MIN) continue;
        int g = gene_of[a];
        if (v_rg[g] < vmin) continue;
        int cap = (int)(0.03 * (double)allele_len[a]) + 1; if (cap > OSP_MAX_ED) cap = OSP_MAX_ED;
        osp_wfa(alleles[a], allele_len[a], read, rlen, d_rg[g] - off[a], cap, &alns[a], NULL, NULL);
        ++run;
        if (alns[a].ok && cells) cells[a] = ((uint32_t)alns[a].nm << 16) | (uint32_t)(alns[a].a_end - alns[a].a_start);
    }
    int best = osp_hla_pick_allele(alns, n_alleles, rlen);
    if (n_cells_run) *n_cells_run = run;
    free(alns); free(d_rg); free(v_rg);
    return best;
}

/* splice_read (src/hla/caller.rs:1518-1576): exon bases of a read through its aligned pairs.
 * pos = 0-based reference start of the record; cigar words are BAM style (len << 4 | op; M=0 I=1 D=2 N=3 S=4 H=5 ==7 X=8);
 * exons are half-open reference ranges in the same coordinate system as pos, in hg38 order.
 * Outputs the read ranges to concatenate and the offset of missing prefix bases. */
int osp_splice_read(int64_t pos, const uint32_t* cigar, int n_cigar, const int64_t* exon_start, const int64_t* exon_end, int n_exons,
                    int32_t* seg_start, int32_t* seg_end, int* n_seg, int64_t* offset_out) {
    int64_t ref_span = 0;
    for (int c = 0; c < n_cigar; ++c) { uint32_t op = cigar[c] & 15u, len = cigar[c] >> 4; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_span += len; }
    int32_t* lookup = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ref_span + 1));
    for (int64_t i = 0; i <= ref_span; ++i) lookup[i] = -1;
    int64_t rp = 0; int32_t qp = 0;
    for (int c = 0; c < n_cigar; ++c) {                       /* aligned_pairs: only M / = / X columns */
        uint32_t op = cigar[c] & 15u, len = cigar[c] >> 4;
        if (op == 0 || op == 7 || op == 8) { for (uint32_t i = 0; i < len; ++i) lookup[rp++] = qp++; }
        else if (op == 1 || op == 4) qp += (int32_t)len;
        else if (op == 2 || op == 3) rp += len;
    }
#define HAS(x) ((x) >= pos && (x) < pos + ref_span && lookup[(x) - pos] >= 0)
    int64_t offset = 0; int ns = 0;
    for (int e = 0; e < n_exons; ++e) {
        int64_t first = exon_start[e], last = exon_end[e] - 1;
        while (!HAS(first) && first <= last) first += 1;
        while (!HAS(last) && first <= last) last -= 1;
        if (ns == 0) offset += first - exon_start[e];
        if (first <= last) { seg_start[ns] = lookup[first - pos]; seg_end[ns] = lookup[last - pos] + 1; ++ns; }
    }
#undef HAS
    free(lookup);
    *n_seg = ns; if (offset_out) *offset_out = offset;
    return 0;
}
