/*
 * oracle/variant.c -- CPU ORACLE (test infrastructure only): variant-gene diplotype search.
 *   NormalizedVariant::new / parse_sequence      src/data_types/normalized_variant.rs:43-170,235-254
 *   NormalizedPgxHaplotype::quant_match          src/data_types/normalized_variant.rs:431-479
 *   find_best_inexact_matches                    src/diplotyper.rs:1411-1509
 *   solve_diplotype                              src/diplotyper.rs:1211-1371
 * The search works on integer ids: the string normalisation (above) and the VCF / database loading are host prep that the
 * tests do in Python (tests/variant_glue.py) exactly as load_database_haplotypes / load_vcf_variants do.
 * Pinned by the reference's tests: src/data_types/normalized_variant.rs:527-1027 and src/diplotyper.rs:1652-2076.
 */
#include "variant_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <ctype.h>

/* ---------------------------------------------------------------- normalisation */
/* parse_sequence (normalized_variant.rs:235-254): "ACGT(8)" repeats, "delins"/"ins" prefixes are skipped, "del" = empty */
static int parse_sequence(const char* s, char* out, size_t cap) {
    size_t n = strlen(s);
    /* ^[A-Z]+\([0-9]+\)$ */
    size_t i = 0; while (i < n && s[i] >= 'A' && s[i] <= 'Z') ++i;
    if (i > 0 && i < n && s[i] == '(' && s[n - 1] == ')' && n - 1 > i + 1) {
        int digits = 1; for (size_t j = i + 1; j + 1 < n; ++j) if (!isdigit((unsigned char)s[j])) digits = 0;
        if (digits) {
            long count = strtol(s + i + 1, NULL, 10);
            size_t len = 0;
            for (long c = 0; c < count; ++c) { if (len + i >= cap) return -1; memcpy(out + len, s, i); len += i; }
            out[len] = '\0'; return (int)len;
        }
    }
    const char* p = s;
    if (strncmp(s, "delins", 6) == 0) p = s + 6;
    else if (strncmp(s, "ins", 3) == 0) p = s + 3;
    else if (strncmp(s, "del", 3) == 0) p = s + n;
    size_t len = strlen(p);
    if (len >= cap) return -1;
    memcpy(out, p, len + 1);
    return (int)len;
}

#define FAIL(msg) do { if (err) snprintf(err, errcap, "%s", msg); return -1; } while (0)

/* NormalizedVariant::new (normalized_variant.rs:43-170).  chrom_seq may be NULL (no reference genome). */
int osp_normalize_variant(const char* chrom, int64_t position, const char* ref_in, const char* alt_in,
                          const char* chrom_seq, int64_t chrom_len, osp_norm_variant* out, char* err, size_t errcap) {
    if (ref_in[0] == '\0') FAIL("ref_allele cannot be empty");
    if (strcmp(ref_in, "del") == 0 && strncmp(alt_in, "ins", 3) != 0) FAIL("Unexpected non-ins alt sequence with a del reference");
    static char r[OSP_VAR_MAXLEN * 2], a[OSP_VAR_MAXLEN * 2];
    int rl = parse_sequence(ref_in, r, OSP_VAR_MAXLEN), al = parse_sequence(alt_in, a, OSP_VAR_MAXLEN);
    if (rl < 0 || al < 0) FAIL("allele too long");
    int64_t pos = position;
    if (chrom_seq) {
        if (pos < 0 || pos + rl > chrom_len) FAIL("position outside the contig");
        if (memcmp(r, chrom_seq + pos, (size_t)rl) != 0) FAIL("provided reference allele differs from the reference genome");
    }
#define PREPEND(c) do { memmove(r + 1, r, (size_t)rl + 1); r[0] = (c); ++rl; memmove(a + 1, a, (size_t)al + 1); a[0] = (c); ++al; } while (0)
    if (rl == 0 && al == 0) FAIL("ref_allele and alt_allele cannot both be empty");
    else if (rl == 0) { if (chrom_seq) PREPEND(chrom_seq[pos]); }
    else if (al == 0) {
        if (pos == 0) FAIL("alt_allele is empty at position 0");
        if (chrom_seq) { pos -= 1; PREPEND(chrom_seq[pos]); }
    }
    while (rl > 1 && al > 1 && r[rl - 1] == a[al - 1]) { r[--rl] = '\0'; a[--al] = '\0'; }                  /* trim shared suffix */
    while (rl > 1 && al > 1 && r[0] == a[0]) { pos += 1; memmove(r, r + 1, (size_t)rl); --rl; memmove(a, a + 1, (size_t)al); --al; }
    if (rl == 0 || al == 0) FAIL("empty allele without a reference genome");                                    /* Rust would panic indexing [len-1] */
    while (r[rl - 1] == a[al - 1]) {                                                                         /* left shift */
        if (pos == 0) break;
        else if (chrom_seq) { pos -= 1; PREPEND(chrom_seq[pos]); }
        else break;
        r[--rl] = '\0'; a[--al] = '\0';
    }
#undef PREPEND
    for (int i = 0; i < rl; ++i) if (!strchr("ACGT", r[i])) FAIL("ACGT alleles only");
    for (int i = 0; i < al; ++i) if (!strchr("ACGT", a[i])) FAIL("ACGT alleles only");
    if (rl >= OSP_VAR_MAXLEN || al >= OSP_VAR_MAXLEN) FAIL("allele too long");
    snprintf(out->chrom, sizeof(out->chrom), "%s", chrom);
    out->position = pos; memcpy(out->ref, r, (size_t)rl + 1); memcpy(out->alt, a, (size_t)al + 1);
    return 0;
}

/* ---------------------------------------------------------------- quant_match (normalized_variant.rs:431-479) */
/* obs: ordered variant ids of the scored haplotype.  Outputs are variant ids in the reference's push order. */
void osp_quant_match(const osp_variant_problem* p, int h, const int32_t* obs, int n_obs,
                     int32_t* matching, int* n_match, int32_t* missing, int* n_missing, int32_t* extra, int* n_extra) {
    const int s0 = p->slot_off[h], s1 = p->slot_off[h + 1];
    uint8_t* matched = (uint8_t*)calloc((size_t)(s1 - s0) + 1, 1);
    *n_match = *n_missing = *n_extra = 0;
    for (int o = 0; o < n_obs; ++o) {
        int mi = -1;
        for (int s = s0; s < s1 && mi < 0; ++s)
            for (int x = p->alt_off[s]; x < p->alt_off[s + 1]; ++x) if (p->alt_var[x] == obs[o]) { mi = s - s0; break; }
        if (mi >= 0) { if (matched[mi]) extra[(*n_extra)++] = obs[o]; else { matched[mi] = 1; matching[(*n_match)++] = obs[o]; } }
        else extra[(*n_extra)++] = obs[o];
    }
    for (int s = s0; s < s1; ++s) {
        int has_none = 0, first_some = -1;
        for (int x = p->alt_off[s]; x < p->alt_off[s + 1]; ++x) { if (p->alt_var[x] < 0) has_none = 1; else if (first_some < 0) first_some = p->alt_var[x]; }
        if (!(matched[s - s0] || has_none)) missing[(*n_missing)++] = first_some;
    }
    free(matched);
}

/* ---------------------------------------------------------------- find_best_inexact_matches (diplotyper.rs:1411-1509) */
void osp_find_best_inexact(const osp_variant_problem* p, const int32_t* obs, const int32_t* obs_sv_label, int n_obs, osp_inexact* out) {
    out->n_best = 0; out->is_sv = 0; out->n_sv_extra = 0;
    /* SV short-circuit (:1414-1431): the first SV label names the haplotype, the others are "extra" */
    int n_sv = 0;
    for (int o = 0; o < n_obs; ++o) if (obs_sv_label[o] >= 0) { if (n_sv == 0) out->sv_label = obs_sv_label[o]; else if (out->n_sv_extra < OSP_VAR_MAXTIES) out->sv_extra[out->n_sv_extra++] = obs_sv_label[o]; ++n_sv; }
    if (n_sv > 0) {
        /* BTreeSet<RegionVariant> of the remaining labels: duplicates collapse */
        int uniq = 0;
        for (int i = 0; i < out->n_sv_extra; ++i) { int dup = 0; for (int j = 0; j < i; ++j) if (out->sv_extra[j] == out->sv_extra[i]) dup = 1; if (!dup) ++uniq; }
        out->is_sv = 1; out->score[0] = 0; out->score[1] = uniq; out->score[2] = 0; out->score[3] = 0;
        return;
    }
    int64_t best[4] = { 1, INT64_MAX, INT64_MAX, INT64_MAX };            /* max_missing_variants = 1 */
    int32_t* m = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_obs + 1));
    int32_t* e = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_obs + 1));
    int n_tie = 0; int32_t* tie = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p->n_haps + 1));
    for (int h = 0; h < p->n_haps; ++h) {
        if (p->hap_is_sv[h]) continue;
        int ns = p->slot_off[h + 1] - p->slot_off[h];
        int32_t* miss = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ns + 1));
        int nm, nmiss, ne;
        osp_quant_match(p, h, obs, n_obs, m, &nm, miss, &nmiss, e, &ne);
        int64_t sc[4] = { 0, 0, 0, 0 };
        for (int i = 0; i < nmiss; ++i) if (p->var_is_core[miss[i]]) sc[0]++; else sc[2]++;
        for (int i = 0; i < ne; ++i) if (p->var_is_core[e[i]]) sc[1]++; else sc[3]++;
        free(miss);
        int cmp = 0; for (int k = 0; k < 4 && !cmp; ++k) cmp = sc[k] < best[k] ? -1 : (sc[k] > best[k]);
        if (cmp < 0) { memcpy(best, sc, sizeof(best)); n_tie = 0; tie[n_tie++] = h; }
        else if (cmp == 0) tie[n_tie++] = h;
    }
    /* sub-alleles shadow core alleles when both tie (:1488-1499) */
    int any_sub = 0; for (int i = 0; i < n_tie; ++i) if (!p->hap_is_core[tie[i]]) any_sub = 1;
    for (int i = 0; i < n_tie; ++i) if ((!p->hap_is_core[tie[i]]) == any_sub && out->n_best < OSP_VAR_MAXTIES) out->best[out->n_best++] = tie[i];
    for (int k = 0; k < 4; ++k) out->score[k] = best[k];
    free(m); free(e); free(tie);
}

/* ---------------------------------------------------------------- solve_diplotype (diplotyper.rs:1211-1371) */
static void push_pairs(osp_variant_result* res, const osp_inexact* a, const osp_inexact* b, int combination) {
    int na = a->is_sv ? 1 : a->n_best, nb = b->is_sv ? 1 : b->n_best;
    for (int i = 0; i < na; ++i) for (int j = 0; j < nb; ++j) {
        if (res->n_dip >= OSP_VAR_MAXDIP) { res->overflow = 1; return; }
        res->dip[res->n_dip][0] = a->is_sv ? -(a->sv_label + 2) : a->best[i];
        res->dip[res->n_dip][1] = b->is_sv ? -(b->sv_label + 2) : b->best[j];
        res->dip_comb[res->n_dip] = combination;
        res->n_dip++;
    }
}

int osp_solve_diplotype(const osp_variant_problem* p, osp_variant_result* res) {
    memset(res, 0, sizeof(*res));
    int32_t* base = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p->n_obs + 1)); int n_base = 0;
    int32_t* base_sv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p->n_obs + 1));
    int32_t* het = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p->n_obs + 1)); int n_het = 0;   /* indices into obs */
    int null_groups = 0; int64_t* ps_seen = (int64_t*)malloc(sizeof(int64_t) * (size_t)(p->n_obs + 1)); int n_ps = 0;
    for (int o = 0; o < p->n_obs; ++o) {
        if (p->obs_gt[o] == OSP_GT_HOM_ALT) { base[n_base] = p->obs_var[o]; base_sv[n_base] = p->obs_sv_label[o]; ++n_base; }
        else if (p->obs_gt[o] == OSP_GT_HOM_REF) { free(base); free(base_sv); free(het); free(ps_seen); return -1; }   /* panics in the reference */
        else {
            het[n_het++] = o;
            if (p->obs_ps[o] >= 0) { int f = 0; for (int i = 0; i < n_ps; ++i) if (ps_seen[i] == p->obs_ps[o]) f = 1; if (!f) ps_seen[n_ps++] = p->obs_ps[o]; }
            else null_groups++;
        }
    }
    if (n_het == 0) {
        osp_inexact b; osp_find_best_inexact(p, base, base_sv, n_base, &b);
        for (int k = 0; k < 4; ++k) res->score[k] = b.score[k];
        int nb = b.is_sv ? 1 : b.n_best;
        for (int i = 0; i < nb; ++i) { int v = b.is_sv ? -(b.sv_label + 2) : b.best[i]; res->dip[res->n_dip][0] = v; res->dip[res->n_dip][1] = v; res->dip_comb[res->n_dip] = 0; res->n_dip++; }
    } else {
        const int total_groups = null_groups + n_ps;
        const uint64_t max_comb = 1ull << (total_groups - 1);
        int64_t best[4] = { INT64_MAX, INT64_MAX, INT64_MAX, INT64_MAX };
        int32_t* h1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p->n_obs + 1)); int32_t* h1s = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p->n_obs + 1));
        int32_t* h2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p->n_obs + 1)); int32_t* h2s = (int32_t*)malloc(sizeof(int32_t) * (size_t)(p->n_obs + 1));
        int64_t* ps_key = (int64_t*)malloc(sizeof(int64_t) * (size_t)(p->n_obs + 1)); uint8_t* ps_val = (uint8_t*)malloc((size_t)p->n_obs + 1);
        for (uint64_t comb = 0; comb < max_comb; ++comb) {
            int n1 = n_base, n2 = n_base;
            memcpy(h1, base, sizeof(int32_t) * (size_t)n_base); memcpy(h2, base, sizeof(int32_t) * (size_t)n_base);
            memcpy(h1s, base_sv, sizeof(int32_t) * (size_t)n_base); memcpy(h2s, base_sv, sizeof(int32_t) * (size_t)n_base);
            int combo_index = 0, n_lookup = 0;
            for (int x = 0; x < n_het; ++x) {
                int o = het[x]; int is_h1;
                if (p->obs_ps[o] >= 0) {
                    int f = -1; for (int i = 0; i < n_lookup; ++i) if (ps_key[i] == p->obs_ps[o]) f = i;
                    if (f >= 0) is_h1 = ps_val[f];
                    else { is_h1 = (int)((comb >> combo_index) & 1); ps_key[n_lookup] = p->obs_ps[o]; ps_val[n_lookup] = (uint8_t)is_h1; ++n_lookup; ++combo_index; }
                } else { is_h1 = (int)((comb >> combo_index) & 1); ++combo_index; }
                int orientation01 = p->obs_gt[o] != OSP_GT_HET_FLIP;
                if (is_h1 == orientation01) { h1[n1] = p->obs_var[o]; h1s[n1] = p->obs_sv_label[o]; ++n1; }
                else { h2[n2] = p->obs_var[o]; h2s[n2] = p->obs_sv_label[o]; ++n2; }
            }
            osp_inexact b1, b2;
            osp_find_best_inexact(p, h1, h1s, n1, &b1); osp_find_best_inexact(p, h2, h2s, n2, &b2);
            int64_t tot[4]; int cmp = 0;
            for (int k = 0; k < 4; ++k) {                       /* usize additions: MAX + x would overflow-panic in debug, wrap in release */
                tot[k] = (b1.score[k] == INT64_MAX || b2.score[k] == INT64_MAX) ? INT64_MAX : b1.score[k] + b2.score[k];
            }
            for (int k = 0; k < 4 && !cmp; ++k) cmp = tot[k] < best[k] ? -1 : (tot[k] > best[k]);
            if (cmp < 0) { memcpy(best, tot, sizeof(best)); res->n_dip = 0; }
            if (cmp <= 0) push_pairs(res, &b1, &b2, (int)comb);
        }
        for (int k = 0; k < 4; ++k) res->score[k] = best[k];
        free(h1); free(h1s); free(h2); free(h2s); free(ps_key); free(ps_val);
    }
    free(base); free(base_sv); free(het); free(ps_seen);
    return 0;
}

/* ---- is_deletion (src/diplotyper.rs:1020-1026): full-gene deletions first, then partial ones ---- */

/* is_full_deletion (src/diplotyper.rs:1034-1089) */
static int full_deletion(const osp_sv_definitions* d, uint64_t start, uint64_t end, int32_t* index) {
    uint8_t* deleted = (uint8_t*)calloc((size_t)d->n_genes + 1, 1);
    /* :1036-1046 every gene named by a definition needs a gene definition */
    for (int i = 0; i < d->full_off[d->n_full]; ++i) if (d->full_gene[i] < 0 || d->full_gene[i] >= d->n_genes) { free(deleted); return -1; }
    /* :1049-1064 deletable genes that lie FULLY inside the region */
    for (int i = 0; i < d->full_off[d->n_full]; ++i) {
        int g = d->full_gene[i];
        if ((uint64_t)d->gene_start[g] >= start && (uint64_t)d->gene_end[g] <= end) deleted[g] = 1;
    }
    int n_deleted = 0; for (int g = 0; g < d->n_genes; ++g) n_deleted += deleted[g];
    *index = -1;
    for (int k = 0; k < d->n_full; ++k) {                                   /* :1068-1086, BTreeMap (label) order */
        int lo = d->full_off[k], hi = d->full_off[k + 1], all_in = 1, uniq = 0;
        for (int i = lo; i < hi; ++i) {
            if (!deleted[d->full_gene[i]]) all_in = 0;
            int dup = 0; for (int j = lo; j < i; ++j) if (d->full_gene[j] == d->full_gene[i]) dup = 1;
            uniq += !dup;
        }
        if (d->full_generic[k]) { if (all_in) *index = k; }                 /* superset: keep looking for a specific one */
        else if (all_in && uniq == n_deleted) { *index = k; break; }        /* set equality */
    }
    free(deleted);
    return 0;
}

/* is_partial_deletion (src/diplotyper.rs:1098-1174) */
static int partial_deletion(const osp_sv_definitions* d, uint64_t start, uint64_t end, int32_t* index) {
    int n = d->n_genes;
    uint8_t* deletable = (uint8_t*)calloc((size_t)n + 1, 1);
    int32_t* first = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* last = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int rc = 0;
    for (int i = 0; i < d->partial_off[d->n_partial]; ++i) {               /* :1100-1110 */
        if (d->partial_gene[i] < 0 || d->partial_gene[i] >= n) { rc = -1; goto done; }
        deletable[d->partial_gene[i]] = 1;
    }
    for (int g = 0; g < n; ++g) {                                           /* :1113-1148 first..last exon fully inside the region */
        first[g] = last[g] = -1;
        if (!deletable[g]) continue;
        int ne = d->exon_off[g + 1] - d->exon_off[g];
        for (int x = 0; x < ne; ++x) {
            uint64_t es = (uint64_t)d->exon_start[d->exon_off[g] + x], ee = (uint64_t)d->exon_end[d->exon_off[g] + x];
            if (es >= start && ee <= end) { if (first[g] < 0) first[g] = x; last[g] = x; }
        }
        if (!d->gene_forward[g] && first[g] >= 0) {                         /* exons are stored in reference orientation (:1131-1139) */
            int f = ne - 1 - last[g], l = ne - 1 - first[g];
            first[g] = f; last[g] = l;
        }
    }
    int n_deleted = 0; for (int g = 0; g < n; ++g) n_deleted += first[g] >= 0;
    *index = -1;
    for (int k = 0; k < d->n_partial; ++k) {                                /* :1153-1171 */
        int lo = d->partial_off[k], hi = d->partial_off[k + 1], keys_in = 1, same = 1;
        for (int i = lo; i < hi; ++i) {
            int g = d->partial_gene[i];
            if (first[g] < 0) { keys_in = 0; same = 0; continue; }
            if (first[g] != d->partial_first[i] || last[g] + 1 != d->partial_end[i]) same = 0;
        }
        if (d->partial_generic[k]) { if (keys_in) *index = k; }
        else if (same && hi - lo == n_deleted) { *index = k; break; }       /* map equality: same genes, same ranges */
    }
done:
    free(deletable); free(first); free(last);
    return rc;
}

int osp_is_deletion(const osp_sv_definitions* d, uint64_t start, uint64_t end, int32_t* kind, int32_t* index) {
    *kind = 0; *index = -1;
    if (full_deletion(d, start, end, index) != 0) return -1;
    if (*index >= 0) { *kind = 1; return 0; }
    if (partial_deletion(d, start, end, index) != 0) return -1;
    if (*index >= 0) *kind = 2;
    return 0;
}
