/*
 * oracle/cyp.c -- CPU ORACLE (test infrastructure only): CYP2D6 chain grammar and the chain-pair likelihood search,
 * restated from /root/reference/src/cyp2d6/{region_label,chaining}.rs and src/cyp2d6/caller.rs:907-957.
 * Pinned by the reference's own tests (src/cyp2d6/chaining.rs:950-1195, src/cyp2d6/caller.rs:972-1006), ported in
 * tests/test_oracle_cyp.py.
 */
#include "sp_oracle.h"
#include "cyp_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <float.h>

/* ---------------------------------------------------------------- region labels (src/cyp2d6/region_label.rs) */
static int is_cyp2d(int t) { return t == OSP_CYP2D6 || t == OSP_CYP2D7 || t == OSP_DELETION || t == OSP_HYBRID; }   /* :27-42 */
static int is_rep(int t) { return t == OSP_REP6 || t == OSP_REP7; }                                                  /* :44-46 */
static int is_reported(int t) { return t == OSP_CYP2D6 || t == OSP_DELETION || t == OSP_HYBRID; }                    /* :48-54 */
static int is_allowed_label(int t) { return !(t == OSP_UNKNOWN || t == OSP_FALSE_ALLELE); }                          /* :173-175 */

static const char* type_name(int t) {
    switch (t) {
        case OSP_UNKNOWN: return "UNKNOWN"; case OSP_REP6: return "REP6"; case OSP_CYP2D6: return "CYP2D6";
        case OSP_LINK: return "link_region"; case OSP_REP7: return "REP7"; case OSP_SPACER: return "spacer";
        case OSP_CYP2D7: return "CYP2D7"; case OSP_DELETION: return "CYP2D6*5"; case OSP_HYBRID: return "Hybrid";
        default: return "FalseAllele";
    }
}

/* full_allele (:131-170) */
void osp_cyp_full_allele(int type, const char* subtype, char* out, size_t cap) {
    switch (type) {
        case OSP_CYP2D6: if (subtype) snprintf(out, cap, "CYP2D6*%s", subtype); else snprintf(out, cap, "CYP2D6"); break;
        case OSP_HYBRID: if (subtype) snprintf(out, cap, "%s", subtype); else snprintf(out, cap, "Hybrid"); break;
        case OSP_FALSE_ALLELE: if (subtype) snprintf(out, cap, "FalseAllele_%s", subtype); else snprintf(out, cap, "FalseAllele"); break;
        default: snprintf(out, cap, "%s", type_name(type));
    }
}

/* simplify_allele (:77-128) */
void osp_cyp_simplify_allele(int type, const char* subtype, int detailed, const osp_cyp_config* cfg, char* out, size_t cap) {
    if (type == OSP_CYP2D6 || type == OSP_HYBRID) {
        if (subtype) {
            for (int i = 0; i < cfg->n_translate; ++i)
                if (strcmp(cfg->tr_key[i], subtype) == 0) { snprintf(out, cap, "*%s", cfg->tr_val[i]); return; }
            if (detailed) { snprintf(out, cap, "*%s", subtype); return; }
            char* endp = NULL;
            double v = strtod(subtype, &endp);
            /* Rust's str::parse::<f64> takes the whole string, no leading blanks */
            if (subtype[0] != '\0' && subtype[0] != ' ' && endp && *endp == '\0' && !(subtype[0] == '0' && (subtype[1] == 'x' || subtype[1] == 'X'))) {
                snprintf(out, cap, "*%lld", (long long)floor(v));
            } else snprintf(out, cap, "*%s", subtype);
            return;
        }
        osp_cyp_full_allele(type, subtype, out, cap);
        return;
    }
    if (type == OSP_DELETION) { snprintf(out, cap, "*5"); return; }
    osp_cyp_full_allele(type, subtype, out, cap);
}

/* is_allowed_label_pair (:178-222) */
int osp_cyp_is_allowed_label_pair(int type1, int type2) {
    int double_star5 = type1 == OSP_DELETION && type2 == OSP_DELETION;
    int unexpected_order =
        type2 == OSP_REP6 ||
        (is_cyp2d(type1) && type1 != OSP_DELETION && type2 != OSP_LINK) ||
        (type2 == OSP_LINK && !is_cyp2d(type1)) ||
        (type1 == OSP_LINK && !is_rep(type2)) ||
        (is_rep(type2) && type1 != OSP_LINK) ||
        (is_rep(type1) && !(type2 == OSP_SPACER || is_cyp2d(type2))) ||
        (type2 == OSP_SPACER && !(is_rep(type1) || type1 == OSP_DELETION)) ||
        (type1 == OSP_SPACER && !is_cyp2d(type2)) ||
        (type2 == OSP_CYP2D7 && type1 != OSP_SPACER) ||
        type1 == OSP_CYP2D7;
    return !double_star5 && !unexpected_order;
}

static int is_normalizing(int t, int normalize_all) { return normalize_all ? is_cyp2d(t) : t == OSP_CYP2D6; }         /* :253-261 */
static int is_chain_head(int t, int normalize_all) {                                                                   /* :227-246 */
    if (t == OSP_REP6 || t == OSP_DELETION) return 1;
    if (t == OSP_CYP2D6 || t == OSP_HYBRID) return is_normalizing(t, normalize_all);
    return 0;
}

/* convert_chain_to_hap (src/cyp2d6/caller.rs:907-957); detail: 0 core, 1 sub-alleles (deep labels are debug only) */
void osp_cyp_convert_chain_to_hap(const int32_t* chain, int n, const int32_t* type, const char* const* subtype, int detail,
                                  const osp_cyp_config* cfg, char* out, size_t cap) {
    out[0] = '\0';
    int num_non_deletion = 0;
    for (int x = n - 1; x >= 0; --x) {
        int t = type[chain[x]];
        if (is_cyp2d(t) && t != OSP_CYP2D7 && t != OSP_DELETION) num_non_deletion++;
    }
    char prev[256] = ""; int run = 0; int first = 1; size_t len = 0;
    char cur[256];
#define FLUSH() do { if (run > 0) { len += (size_t)snprintf(out + len, len < cap ? cap - len : 0, "%s%s", first ? "" : " + ", prev); first = 0; \
        if (run > 1) len += (size_t)snprintf(out + len, len < cap ? cap - len : 0, "x%d", run); } } while (0)
    for (int x = n - 1; x >= 0; --x) {
        int h = chain[x], t = type[h];
        if (!(is_cyp2d(t) && t != OSP_CYP2D7)) continue;
        if (t == OSP_DELETION && num_non_deletion > 0) continue;
        osp_cyp_simplify_allele(t, subtype[h], detail == 1, cfg, cur, sizeof(cur));
        if (run > 0 && strcmp(cur, prev) == 0) { run++; }
        else { FLUSH(); strcpy(prev, cur); run = 1; }
    }
    FLUSH();
#undef FLUSH
}

/* ---------------------------------------------------------------- chain pair search (src/cyp2d6/chaining.rs:223-903) */
typedef struct { int32_t* items; int32_t len; } chain_t;

static int conn_contains(const osp_cyp_config* cfg, const char* a, const char* b) {
    for (int i = 0; i < cfg->n_conn; ++i) if (strcmp(cfg->conn_a[i], a) == 0 && strcmp(cfg->conn_b[i], b) == 0) return 1;
    return 0;
}

/* check_chain_inferrences (:603-674) */
static void check_chain_inferrences(const osp_chain_problem* p, const int32_t* chain, int n, const uint8_t* inferred,
                                    int* allowed_inferrence, int* allowed_candidate) {
    const int H = p->n_haps;
    int last = chain[n - 1];
    int last_is_cyp2d = is_cyp2d(p->type[last]);
    int opt_index = -1;
    for (int ci = n - 2; ci >= 0; --ci) if (is_cyp2d(p->type[chain[ci]])) { opt_index = ci; break; }
    int inferrence_detected = 0;
    for (int w = (opt_index < 0 ? 0 : opt_index); w + 1 < n; ++w) if (inferred[chain[w] * H + chain[w + 1]]) inferrence_detected = 1;
    if (inferrence_detected) {
        if (last_is_cyp2d) {
            if (opt_index >= 0) {
                int prev = chain[opt_index];
                char h1[256], h2[256];
                osp_cyp_simplify_allele(p->type[prev], p->subtype[prev], 0, &p->cfg, h1, sizeof(h1));
                osp_cyp_simplify_allele(p->type[last], p->subtype[last], 0, &p->cfg, h2, sizeof(h2));
                int connected = prev != last && conn_contains(&p->cfg, h1, h2);
                int d7_tail = p->type[last] == OSP_CYP2D7 && p->type[prev] != OSP_CYP2D7 && is_cyp2d(p->type[prev]);
                int allowed = connected || d7_tail;
                *allowed_inferrence = allowed; *allowed_candidate = allowed;
            } else { *allowed_inferrence = 1; *allowed_candidate = 1; }
        } else { *allowed_inferrence = 1; *allowed_candidate = 0; }
    } else { *allowed_inferrence = 1; *allowed_candidate = 1; }
}

/* containment_score (:683-731); adds split_frac * overlap of every best window into hap_weights */
static uint64_t containment_and_weights(const int32_t* c1, int n1, const int32_t* c2, int n2, const uint64_t* w_ed, const double* w_ov,
                                        int weight_len, int H, double* hap_weights) {
    uint64_t optimum = 0, worst = 0;
    for (int x = 0; x < weight_len; ++x) {
        uint64_t mn = UINT64_MAX, mx = 0;
        for (int h = 0; h < H; ++h) { uint64_t v = w_ed[(size_t)x * H + h]; if (v < mn) mn = v; if (v > mx) mx = v; }
        optimum += mn; worst += mx;
    }
    uint64_t best_score = 2 * worst;
    /* first pass: best score; second pass: the windows that reach it, in the reference's push order */
    for (int pass = 0; pass < 2; ++pass) {
        size_t n_best = 0;
        if (pass == 1) {
            for (int which = 0; which < 2; ++which) {
                const int32_t* other = which == 0 ? c1 : c2; int on = which == 0 ? n1 : n2;
                if (on < weight_len) continue;
                for (int s = 0; s + weight_len <= on; ++s) {
                    uint64_t tot = 0;
                    for (int x = 0; x < weight_len; ++x) tot += w_ed[(size_t)x * H + other[s + x]];
                    if (tot == best_score) ++n_best;
                }
            }
        }
        double split_frac = pass == 1 ? 1.0 / (double)n_best : 0.0;
        for (int which = 0; which < 2; ++which) {
            const int32_t* other = which == 0 ? c1 : c2; int on = which == 0 ? n1 : n2;
            if (on < weight_len) continue;
            for (int s = 0; s + weight_len <= on; ++s) {
                uint64_t tot = 0;
                for (int x = 0; x < weight_len; ++x) tot += w_ed[(size_t)x * H + other[s + x]];
                if (pass == 0) { if (tot < best_score) best_score = tot; }
                else if (tot == best_score) {
                    for (int x = 0; x < weight_len; ++x) {
                        int con = other[s + x];
                        hap_weights[con] += split_frac * w_ov[(size_t)x * H + con];
                    }
                }
            }
        }
    }
    return best_score - optimum;
}

/* unexpected_count (:739-775) */
static uint32_t unexpected_count(const osp_chain_problem* p, const int32_t* chain, int n) {
    char (*red)[256] = (char (*)[256])malloc(sizeof(char[256]) * (size_t)(n > 0 ? n : 1));
    int nr = 0;
    for (int x = 0; x < n; ++x) {
        int t = p->type[chain[x]];
        if (is_cyp2d(t) && t != OSP_CYP2D7) osp_cyp_simplify_allele(t, p->subtype[chain[x]], 0, &p->cfg, red[nr++], 256);
    }
    uint32_t errors = 0;
    if (nr == 0 || red[0][0] != '*') errors += 1;
    if (nr == 1) for (int i = 0; i < p->cfg.n_single; ++i) if (strcmp(p->cfg.singles[i], red[0]) == 0) { errors += 1; break; }
    for (int x = 0; x + 1 < nr; ++x) if (!conn_contains(&p->cfg, red[x], red[x + 1])) errors += 1;
    free(red);
    return errors;
}

static int is_sub(const int32_t* hay, int hn, const int32_t* needle, int nn) {                                        /* :782-784 */
    /* slice::windows(0) panics in Rust; the reference never passes an empty chain */
    for (int s = 0; s + nn <= hn; ++s) if (memcmp(hay + s, needle, sizeof(int32_t) * (size_t)nn) == 0) return 1;
    return 0;
}

int osp_cyp_find_best_chain_pair(const osp_chain_problem* p, osp_chain_result* res) {
    memset(res, 0, sizeof(*res));
    const int H = p->n_haps;
    if (p->lasso < 0.0) { res->status = OSP_CHAIN_BAD_ARG; return res->status; }
    uint8_t* downstream = (uint8_t*)calloc((size_t)H * H + 1, 1);
    uint8_t* inferred = (uint8_t*)calloc((size_t)H * H + 1, 1);
    /* observed connections (:245-264) */
    for (int r = 0; r < p->n_reads; ++r)
        for (int c = p->read_chain_off[r]; c < p->read_chain_off[r + 1]; ++c) {
            const int32_t* ch = p->chain_items + p->chain_off[c]; int n = p->chain_off[c + 1] - p->chain_off[c];
            for (int i = 1; i < n; ++i) {
                int up = ch[i - 1], down = ch[i];
                if (is_allowed_label(p->type[up]) && is_allowed_label(p->type[down]) &&
                    (p->ignore_chain_label_limits || osp_cyp_is_allowed_label_pair(p->type[up], p->type[down]))) downstream[up * H + down] = 1;
            }
        }
    /* inferred connections (:267-305) */
    if (p->infer_connections) {
        for (int i = 0; i < H; ++i) {
            int downstream_no_link = 1; for (int j = 0; j < H; ++j) if (downstream[i * H + j]) downstream_no_link = 0;
            for (int j = 0; j < H; ++j) {
                int upstream_no_link = 1; for (int v = 0; v < H; ++v) if (downstream[v * H + j]) upstream_no_link = 0;
                if ((downstream_no_link || upstream_no_link) && !downstream[i * H + j] && is_allowed_label(p->type[i]) &&
                    is_allowed_label(p->type[j]) && osp_cyp_is_allowed_label_pair(p->type[i], p->type[j])) inferred[i * H + j] = 1;
            }
        }
    }
    /* heads (:308-323) */
    int n_heads = 0; int32_t* heads = (int32_t*)malloc(sizeof(int32_t) * (size_t)(H + 1));
    for (int i = 0; i < H; ++i) if (p->ignore_chain_label_limits || is_chain_head(p->type[i], p->normalize_all_alleles)) heads[n_heads++] = i;
    if (n_heads == 0) { res->status = OSP_CHAIN_NO_HEAD; free(downstream); free(inferred); free(heads); return res->status; }

    /* enumeration: remaining_chains is a LIFO stack (:326-391) */
    size_t stack_cap = 64, stack_n = 0; chain_t* stack = (chain_t*)malloc(sizeof(chain_t) * stack_cap);
    size_t pos_cap = 64, P = 0; chain_t* possible = (chain_t*)malloc(sizeof(chain_t) * pos_cap);
#define PUSH(vec, n_, cap_, item) do { if ((n_) == (cap_)) { (cap_) *= 2; (vec) = (chain_t*)realloc((vec), sizeof(chain_t) * (cap_)); } (vec)[(n_)++] = (item); } while (0)
    for (int i = 0; i < n_heads; ++i) { chain_t c; c.len = 1; c.items = (int32_t*)malloc(sizeof(int32_t)); c.items[0] = heads[i]; PUSH(stack, stack_n, stack_cap, c); }
    const int max_copy_number = 3;
    int overflow = 0;
    while (stack_n > 0) {
        chain_t cur = stack[--stack_n];
        int allowed_inf, allowed_cand;
        check_chain_inferrences(p, cur.items, cur.len, inferred, &allowed_inf, &allowed_cand);
        if (!allowed_inf) { free(cur.items); continue; }
        int nonempty = 0;                                      /* convert_chain_to_hap(..SubAlleles..) is non-empty */
        for (int x = 0; x < cur.len; ++x) { int t = p->type[cur.items[x]]; if (is_cyp2d(t) && t != OSP_CYP2D7) nonempty = 1; }
        if (p->ignore_chain_label_limits || (nonempty && allowed_cand)) {
            chain_t keep; keep.len = cur.len; keep.items = (int32_t*)malloc(sizeof(int32_t) * (size_t)cur.len);
            memcpy(keep.items, cur.items, sizeof(int32_t) * (size_t)cur.len);
            PUSH(possible, P, pos_cap, keep);
        }
        int cur_index = cur.items[cur.len - 1];
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1 && !p->infer_connections) break;
            const uint8_t* table = rep == 0 ? downstream : inferred;
            for (int ext = 0; ext < H; ++ext) {
                if (!table[cur_index * H + ext]) continue;
                int count = 0; for (int x = 0; x < cur.len; ++x) if (cur.items[x] == ext) ++count;
                if (count >= max_copy_number) continue;
                if (cur.len + 1 > OSP_MAX_CHAIN) { overflow = 1; continue; }
                chain_t nc; nc.len = cur.len + 1; nc.items = (int32_t*)malloc(sizeof(int32_t) * (size_t)nc.len);
                memcpy(nc.items, cur.items, sizeof(int32_t) * (size_t)cur.len); nc.items[cur.len] = ext;
                PUSH(stack, stack_n, stack_cap, nc);
            }
        }
        free(cur.items);
    }
#undef PUSH
    free(stack); free(heads);
    res->n_possible = (int32_t)P;
    if (overflow) res->status = OSP_CHAIN_TOO_LONG;
    if (P == 0 && !overflow) res->status = OSP_CHAIN_NO_CHAINS;
    if (res->status) { for (size_t i = 0; i < P; ++i) free(possible[i].items); free(possible); free(downstream); free(inferred); return res->status; }

    /* per-chain quantities that do not depend on the partner */
    uint32_t* unexp = (uint32_t*)calloc(P, sizeof(uint32_t));
    if (!p->ignore_chain_label_limits) for (size_t i = 0; i < P; ++i) unexp[i] = unexpected_count(p, possible[i].items, possible[i].len);

    /* pair loop (:409-534).  The reference keeps a 10-entry max-heap and skips pairs whose cheap partial cost cannot enter
     * it; neither changes the winner = min (primary_score, i, j) over the valid pairs, which is what is computed here. */
    int32_t* hap_counts = (int32_t*)malloc(sizeof(int32_t) * (size_t)H);
    double* hap_weights = (double*)malloc(sizeof(double) * (size_t)H);
    double* probs = (double*)malloc(sizeof(double) * (size_t)H);
    uint64_t* cover = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)H);
    int have = 0; double best_score = 0; size_t bi = 0, bj = 0;
    osp_chain_result best; memset(&best, 0, sizeof(best));
    for (size_t i = 0; i < P; ++i) for (size_t j = i; j < P; ++j) {
        const chain_t* ci = &possible[i]; const chain_t* cj = &possible[j];
        memset(hap_counts, 0, sizeof(int32_t) * (size_t)H);
        for (int x = 0; x < ci->len; ++x) hap_counts[ci->items[x]]++;
        for (int x = 0; x < cj->len; ++x) hap_counts[cj->items[x]]++;
        /* count_unexpected_alleles (:794-819) */
        int32_t unexpected_alleles = 0;
        for (int h = 0; h < H; ++h) {
            int t = p->type[h];
            if (is_allowed_label(t) && (p->ignore_chain_label_limits || is_normalizing(t, p->normalize_all_alleles) || is_reported(t)) && hap_counts[h] > 0)
                unexpected_alleles += hap_counts[h] - 1;
        }
        double allele_expected_penalty = p->lasso * (double)unexpected_alleles;
        uint64_t unmet = 0;                                    /* (:427-440) debug only */
        for (int r = 0; r < p->n_reads; ++r) {
            int supported = 0;
            for (int c = p->read_chain_off[r]; c < p->read_chain_off[r + 1] && !supported; ++c) {
                const int32_t* ch = p->chain_items + p->chain_off[c]; int n = p->chain_off[c + 1] - p->chain_off[c];
                if (is_sub(ci->items, ci->len, ch, n) || is_sub(cj->items, cj->len, ch, n)) supported = 1;
            }
            if (!supported) unmet++;
        }
        uint32_t mismatch = p->ignore_chain_label_limits ? 0 : unexp[i] + unexp[j];
        double unexpected_chain_penalty = (double)mismatch * p->unexpected;
        uint32_t n_inf = 0;                                    /* count_inferred_edges (:828-840) */
        if (p->infer_connections) {
            for (int x = 0; x + 1 < ci->len; ++x) if (inferred[ci->items[x] * H + ci->items[x + 1]]) n_inf++;
            for (int x = 0; x + 1 < cj->len; ++x) if (inferred[cj->items[x] * H + cj->items[x + 1]]) n_inf++;
        }
        double inferred_chain_penalty = (double)n_inf * p->inferred;
        /* read-level part (:466-485) */
        uint64_t read_combined_ed = 0;
        for (int h = 0; h < H; ++h) hap_weights[h] = 0.0;
        for (int r = 0; r < p->n_reads; ++r) {
            int wl = p->read_w_off[r + 1] - p->read_w_off[r];
            uint64_t sc = containment_and_weights(ci->items, ci->len, cj->items, cj->len, p->w_ed + (size_t)p->read_w_off[r] * H,
                                                  p->w_ov + (size_t)p->read_w_off[r] * H, wl, H, hap_weights);
            uint64_t sum = read_combined_ed + sc; read_combined_ed = sum < read_combined_ed ? UINT64_MAX : sum;   /* saturating_add */
        }
        double ln_ed_penalty = (double)read_combined_ed * p->ln_ed;
        /* get_multinomial_score (:854-903) */
        int nr = 0; int32_t total = 0; uint64_t cov_sum = 0;
        for (int h = 0; h < H; ++h) {
            if (hap_counts[h] > 0 && (p->ignore_chain_label_limits || is_normalizing(p->type[h], p->normalize_all_alleles))) {
                probs[nr] = (double)hap_counts[h]; total += hap_counts[h];
                cover[nr] = (uint64_t)round(hap_weights[h]); cov_sum += cover[nr]; nr++;
            }
        }
        for (int x = 0; x < nr; ++x) probs[x] = probs[x] / (double)total;
        double mn;
        if (nr == 0 || cov_sum == 0) {
            int d1 = 0, d2 = 0;
            for (int x = 0; x < ci->len; ++x) if (p->type[ci->items[x]] == OSP_DELETION) d1 = 1;
            for (int x = 0; x < cj->len; ++x) if (p->type[cj->items[x]] == OSP_DELETION) d2 = 1;
            if (!p->normalize_all_alleles && d1 && d2) mn = 0.0; else continue;      /* invalid pair, skipped */
        } else mn = fabs(osp_multinomial_ln_pmf(probs, cover, nr));
        double primary = ln_ed_penalty + mn + allele_expected_penalty + unexpected_chain_penalty + inferred_chain_penalty;   /* (:172-174) */
        if (!have || primary < best_score) {                   /* (i, j) ascending => first minimum = min (score, i, j) */
            have = 1; best_score = primary; bi = i; bj = j;
            best.score = primary; best.ln_ed_penalty = ln_ed_penalty; best.mn_llh_penalty = mn; best.allele_expected_penalty = allele_expected_penalty;
            best.unexpected_chain_penalty = unexpected_chain_penalty; best.inferred_chain_penalty = inferred_chain_penalty;
            best.edit_distance = read_combined_ed; best.unmet_observations = unmet;
        }
    }
    if (!have) res->status = OSP_CHAIN_NO_PAIRS;
    else {
        *res = best; res->status = OSP_CHAIN_OK; res->n_possible = (int32_t)P; res->index1 = (int32_t)bi; res->index2 = (int32_t)bj;
        const chain_t* a = &possible[bi]; const chain_t* b = &possible[bj];
        /* best_chain_pair.sort() : lexicographic Vec<usize> order (:568-573) */
        int cmp = 0; for (int x = 0; x < a->len && x < b->len && !cmp; ++x) cmp = a->items[x] < b->items[x] ? -1 : (a->items[x] > b->items[x]);
        if (!cmp) cmp = a->len < b->len ? -1 : (a->len > b->len);
        if (cmp > 0) { const chain_t* t = a; a = b; b = t; }
        res->n1 = a->len; memcpy(res->chain1, a->items, sizeof(int32_t) * (size_t)a->len);
        res->n2 = b->len; memcpy(res->chain2, b->items, sizeof(int32_t) * (size_t)b->len);
    }
    free(hap_counts); free(hap_weights); free(probs); free(cover); free(unexp);
    for (size_t i = 0; i < P; ++i) free(possible[i].items);
    free(possible); free(downstream); free(inferred);
    return res->status;
}

/* ---------------------------------------------------------------- weight_sequence (src/cyp2d6/chaining.rs:28-103)
 * The read segment is the minimap2 target, every allowed consensus a query.  On the alignment contract each consensus
 * yields at most one mapping (anchor with >= OSP_CYP_MIN_VOTES votes, then one banded cell).
 * out_ed/out_ov: n_cons entries (defaults (seq_len, 0.0)); returns 1 when the weights are kept, 0 when the reference
 * returns the empty vec (best penalised fraction > 0.05). */
int osp_cyp_weight_sequence(const uint8_t* seq, int seq_len, int n_cons, const uint8_t* const* cons, const int32_t* cons_len,
                            const uint8_t* allowed, uint64_t* out_ed, double* out_ov) {
    const double maximum_allowed_ed = 0.05;
    double min_ed_frac = 1.0;
    for (int c = 0; c < n_cons; ++c) { out_ed[c] = (uint64_t)seq_len; out_ov[c] = 0.0; }
    /* the placements of the segment on every consensus that counts (the library's alignment contract), then -- the rule of sp_cyp.hip cyp_pick_near_min -- those
     * within OSP_CYP_K4_NEAR of the smallest edits + unmapped bases re-scored the reference's way: minimap2's two-piece affine scores on the 256 diagonals around
     * the placement (osp_affine_local: the segment is minimap2's target, the consensus its query), i.e. its NM, its clipping, its spans */
    osp_aln* als = (osp_aln*)calloc((size_t)(n_cons > 0 ? n_cons : 1), sizeof(osp_aln));
    int64_t best = INT64_MAX;
    for (int c = 0; c < n_cons; ++c) {
        if (!allowed[c]) continue;                                      /* Unknown / FalseAllele labels are skipped (:52-55) */
        int diag = 0, votes = osp_anchor(cons[c], cons_len[c], seq, seq_len, &diag);      /* seq_pos - cons_pos */
        if (votes < OSP_CYP_MIN_VOTES) continue;
        if (!osp_wfa_retry(cons[c], cons_len[c], seq, seq_len, diag, OSP_MAX_ED, &als[c], NULL, NULL)) { als[c].ok = 0; continue; }
        const int64_t ms = (int64_t)als[c].nm + (seq_len - (als[c].b_end - als[c].b_start));
        if (ms < best) best = ms;
    }
    for (int c = 0; c < n_cons; ++c) {
        if (!allowed[c] || !als[c].ok) continue;
        osp_aln al = als[c];
        if ((int64_t)al.nm + (seq_len - (al.b_end - al.b_start)) <= best + OSP_CYP_K4_NEAR) {
            const osp_affine_opts ao = { 1, 4, 6, 2, 26, 1, 1 };
            osp_affine_out r;
            const int d_mid = ((al.b_start - al.a_start) + (al.b_end - al.a_end)) / 2;                /* seq_pos - cons_pos along the placement */
            osp_affine_local(seq, seq_len, cons[c], cons_len[c], -d_mid, OSP_WIDE_BAND, &ao, &r);
            if (r.score > 0) { al.nm = r.nm; al.a_start = r.q_start; al.a_end = r.q_end; al.b_start = r.t_start; al.b_end = r.t_end; }
        }
        uint64_t nm = (uint64_t)al.nm, unmapped = (uint64_t)(seq_len - (al.b_end - al.b_start));
        uint64_t clipped_start = (uint64_t)al.a_start, clipped_end = (uint64_t)(cons_len[c] - al.a_end);
        uint64_t match_score = nm + unmapped;
        double overlap_score = 1.0 - (double)(clipped_start + clipped_end) / (double)cons_len[c];
        if (match_score < out_ed[c] || (match_score == out_ed[c] && overlap_score > out_ov[c])) {
            out_ed[c] = match_score; out_ov[c] = overlap_score;
            double sc = osp_custom_score((uint64_t)seq_len, nm, unmapped, 1);
            if (sc < min_ed_frac) min_ed_frac = sc;
        }
    }
    free(als);
    return min_ed_frac <= maximum_allowed_ed;
}

/* ---------------------------------------------------------------- find_base_type_in_sequence (src/cyp2d6/haplotyper.rs:142-315)
 * templates in key order (sorted by full_allele, :175-183); up to OSP_CYP_TOPK placements per template. */
static double hit_score(const osp_region_hit* h, int penalize) {
    return osp_custom_score((uint64_t)h->seq_len, (uint64_t)h->nm, (uint64_t)h->unmapped, penalize);
}
static int is_penalized_type(int t) { return t == OSP_DELETION || t == OSP_REP6 || t == OSP_REP7; }              /* :185-191 */
double osp_cyp_overlap_score(int s1, int e1, int s2, int e2) {                                                /* :877-892 */
    int min_end = e1 < e2 ? e1 : e2, max_start = s1 > s2 ? s1 : s2;
    if (max_start >= min_end) return 0.0;
    double l1 = (double)(e1 - s1), l2 = (double)(e2 - s2), shared = (double)(min_end - max_start);
    return shared / (l1 < l2 ? l1 : l2);
}

int osp_cyp_find_base_type(const uint8_t* seq, int seq_len, int n_templates, const uint8_t* const* tmpl, const int32_t* tmpl_len,
                           const int32_t* tmpl_type, double max_missing_frac, osp_region_hit* out, int cap) {
    return osp_cyp_find_base_type_ex(seq, seq_len, n_templates, tmpl, tmpl_len, tmpl_type, max_missing_frac, 0, out, cap);
}
/* rescore: the numbers of a hit are the reference's -- the template is minimap2's query, the sequence its target: two-piece affine gaps and end clipping (osp_affine_local,
 * map-hifi scores) on the 256 diagonals around the placement's middle diagonal; start / end / nm / unmapped / clips of a hit are those numbers (what minimap2 reports for the
 * mapping; what the segments are cut from, what the missing-fraction filter sees).  (The library's sp_cyp_find_regions under its context option "mm2_rescore", the default.)
 * Round 6: the decisions in front of the hit list -- the max_ed_frac filter (:228-232) and the collapse of overlapping placements (:260-296) -- see those numbers too wherever they
 * can decide: a placement (with a score of at most OSP_K3_CAP_HI by the contract's count: cells give up one edit past the 0.05 cap) is re-scored BEFORE the filter and the
 * collapse when it is CRITICAL:
 *       no other placement of the same sequence that overlaps it by more than OSP_K3_OVL (the collapse asks for 0.9) clearly beats it -- "clearly": the rival's score, taken
 *       the way the collapse would compare the two, is lower by more than a quarter of itself plus 0.001 (a handful of edits: less than that, end clipping or an affine gap can
 *       turn the pair round);
 * every other placement keeps the contract's own counts through the filter and the collapse (the other gene copy's templates, hundreds of clustered edits behind a rival),
 * is re-scored if it survives them and then has to pass the max_ed_frac filter on its re-scored numbers once more. */
#define OSP_K3_CAP_HI 0.056
#define OSP_K3_OVL 0.85
#define OSP_K3_RETRY_VOTES 128
#define OSP_K3_RETRY_HOLE 500
static void k3_rescore_hit(const uint8_t* seq, int seq_len, const uint8_t* const* tmpl, const int32_t* tmpl_len, osp_region_hit* h) {
    const int t = h->template_idx;
    const osp_affine_opts ao = { 1, 4, 6, 2, 26, 1, 1 };
    /* (sequence position - template position) at both ends of the placement, as the contract reports them */
    const int twice = (h->start - h->clip_start) + (h->end - (tmpl_len[t] - h->clip_end));
    osp_affine_out af;
    osp_affine_local(seq, seq_len, tmpl[t], tmpl_len[t], -(twice / 2), 256, &ao, &af);
    if (af.score <= 0) return;                                                                  /* nothing aligns the reference's way: the hit keeps the contract's counts */
    h->start = af.t_start; h->end = af.t_end; h->nm = af.nm;
    h->unmapped = tmpl_len[t] - (af.q_end - af.q_start); h->clip_start = af.q_start; h->clip_end = tmpl_len[t] - af.q_end;
}
int osp_cyp_find_base_type_ex(const uint8_t* seq, int seq_len, int n_templates, const uint8_t* const* tmpl, const int32_t* tmpl_len,
                              const int32_t* tmpl_type, double max_missing_frac, int rescore, osp_region_hit* out, int cap) {
    if (seq_len == 0) return 0;
    const double max_ed_frac = 0.05;
    int n_un = 0, un_cap = n_templates * OSP_CYP_TOPK + 1;
    osp_region_hit* un = (osp_region_hit*)malloc(sizeof(osp_region_hit) * (size_t)un_cap);
    uint8_t* done = (uint8_t*)calloc((size_t)un_cap, 1);                                       /* the placement carries its re-scored numbers */
    /* every placement of every template (top-k anchors, 64 diagonals, the edit cap of max_ed_frac) */
    osp_aln* al = (osp_aln*)calloc((size_t)n_templates * OSP_CYP_TOPK, sizeof(osp_aln));
    uint8_t* al_ok = (uint8_t*)calloc((size_t)n_templates * OSP_CYP_TOPK, 1);
    int* d0 = (int*)calloc((size_t)n_templates, sizeof(int)); int* v0 = (int*)calloc((size_t)n_templates, sizeof(int));
    for (int t = 0; t < n_templates; ++t) {
        int diags[OSP_CYP_TOPK], votes[OSP_CYP_TOPK];
        int np = osp_anchor_topk(tmpl[t], tmpl_len[t], seq, seq_len, OSP_CYP_TOPK, diags, votes);      /* seq_pos - template_pos */
        if (np > 0) { d0[t] = diags[0]; v0[t] = votes[0]; }
        for (int k = 0; k < np; ++k) {
            if (votes[k] < OSP_CYP_MIN_VOTES) continue;
            /* nm <= max_ed_frac * aligned span <= 0.05 * template length (haplotyper.rs:160,228-232): beyond that the hit is dropped anyway */
            int cap = (int)(0.05 * (double)tmpl_len[t]) + 1; if (cap > OSP_MAX_ED) cap = OSP_MAX_ED;
            if (osp_wfa(tmpl[t], tmpl_len[t], seq, seq_len, diags[k], cap, &al[t * OSP_CYP_TOPK + k], NULL, NULL)) al_ok[t * OSP_CYP_TOPK + k] = 1;
        }
    }
    if (rescore) {
        /* the wide-band retry (round 6): minimap2 chains a template across a 40 - 120 base insertion or deletion in the sequence; the 64-diagonal cell leaves its band there and
         * is lost, and so is every other template over that stretch.  Every template that is lost altogether although it anchors strongly (>= OSP_K3_RETRY_VOTES votes on its best
         * diagonal) and whose expected span has >= OSP_K3_RETRY_HOLE bases that NO placement covers runs once more on 256 diagonals around that anchor, same edit cap */
        int n_iv = 0, any_lost = 0;
        int* iv = (int*)malloc(sizeof(int) * 2 * (size_t)(n_templates * OSP_CYP_TOPK + 1));
        for (int t = 0; t < n_templates; ++t) {
            int placed = 0;
            for (int k = 0; k < OSP_CYP_TOPK; ++k) if (al_ok[t * OSP_CYP_TOPK + k]) { placed = 1; iv[2 * n_iv] = al[t * OSP_CYP_TOPK + k].b_start; iv[2 * n_iv + 1] = al[t * OSP_CYP_TOPK + k].b_end; ++n_iv; }
            if (!placed && v0[t] >= OSP_K3_RETRY_VOTES) any_lost = 1;
        }
        if (any_lost) {
            for (int i = 1; i < n_iv; ++i) {                                                      /* by (start, end) */
                const int s0 = iv[2 * i], e0 = iv[2 * i + 1]; int j = i - 1;
                while (j >= 0 && (iv[2 * j] > s0 || (iv[2 * j] == s0 && iv[2 * j + 1] > e0))) { iv[2 * j + 2] = iv[2 * j]; iv[2 * j + 3] = iv[2 * j + 1]; --j; }
                iv[2 * j + 2] = s0; iv[2 * j + 3] = e0;
            }
            for (int t = 0; t < n_templates; ++t) {
                int placed = 0;
                for (int k = 0; k < OSP_CYP_TOPK; ++k) placed |= al_ok[t * OSP_CYP_TOPK + k];
                if (placed || v0[t] < OSP_K3_RETRY_VOTES) continue;
                const int s0 = d0[t] > 0 ? d0[t] : 0, e0 = d0[t] + tmpl_len[t] < seq_len ? d0[t] + tmpl_len[t] : seq_len;
                int uncovered = 0, at = s0;
                for (int i = 0; i < n_iv; ++i) { if (iv[2 * i + 1] <= at) continue; if (iv[2 * i] >= e0) break; if (iv[2 * i] > at) uncovered += iv[2 * i] - at; if (iv[2 * i + 1] > at) at = iv[2 * i + 1]; if (at >= e0) break; }
                if (at < e0) uncovered += e0 - at;
                if (uncovered < OSP_K3_RETRY_HOLE) continue;
                int cap = (int)(0.05 * (double)tmpl_len[t]) + 1; if (cap > OSP_MAX_ED) cap = OSP_MAX_ED;
                if (osp_wfa_band(tmpl[t], tmpl_len[t], seq, seq_len, d0[t], cap, OSP_WIDE_BAND, &al[t * OSP_CYP_TOPK], NULL, NULL)) al_ok[t * OSP_CYP_TOPK] = 1;
            }
        }
        free(iv);
    }
    for (int t = 0; t < n_templates; ++t) for (int k = 0; k < OSP_CYP_TOPK; ++k) {
        if (!al_ok[t * OSP_CYP_TOPK + k]) continue;
        const osp_aln* a = &al[t * OSP_CYP_TOPK + k];
        osp_region_hit h;
        h.template_idx = t; h.start = a->b_start; h.end = a->b_end;
        h.seq_len = tmpl_len[t]; h.nm = a->nm; h.unmapped = tmpl_len[t] - (a->a_end - a->a_start);
        h.clip_start = a->a_start; h.clip_end = tmpl_len[t] - a->a_end;
        if (hit_score(&h, is_penalized_type(tmpl_type[t])) > (rescore ? OSP_K3_CAP_HI : max_ed_frac)) continue;           /* :228-232 ; Forward only */
        un[n_un++] = h;
    }
    free(al); free(al_ok); free(d0); free(v0);
    if (rescore) {
        /* the critical placements, by the contract's own counts: near the cap, or not clearly beaten by a rival that overlaps them */
        uint8_t* crit = (uint8_t*)calloc((size_t)n_un + 1, 1);
        for (int i = 0; i < n_un; ++i) crit[i] = 1;
        for (int i = 0; i < n_un; ++i) for (int j = i + 1; j < n_un; ++j) {
            if (!(osp_cyp_overlap_score(un[i].start, un[i].end, un[j].start, un[j].end) > OSP_K3_OVL)) continue;
            const int pen = is_penalized_type(tmpl_type[un[i].template_idx]) || is_penalized_type(tmpl_type[un[j].template_idx]);
            const double a = hit_score(&un[i], pen), b = hit_score(&un[j], pen);
            if (a > 1.25 * b + 0.001) crit[i] = 0;                                           /* i is clearly beaten by j */
            if (b > 1.25 * a + 0.001) crit[j] = 0;
        }
        for (int i = 0; i < n_un; ++i) if (crit[i]) { k3_rescore_hit(seq, seq_len, tmpl, tmpl_len, &un[i]); done[i] = 1; }
        free(crit);
        /* the filter, on what every placement carries now */
        int m = 0;
        for (int i = 0; i < n_un; ++i) if (!(hit_score(&un[i], is_penalized_type(tmpl_type[un[i].template_idx])) > max_ed_frac)) { un[m] = un[i]; done[m] = done[i]; ++m; }
        n_un = m;
    }
    /* stable sort by (start, end) (:252-255) */
    for (int i = 1; i < n_un; ++i) {
        osp_region_hit x = un[i]; const uint8_t dx = done[i]; int j = i - 1;
        while (j >= 0 && (un[j].start > x.start || (un[j].start == x.start && un[j].end > x.end))) { un[j + 1] = un[j]; done[j + 1] = done[j]; --j; }
        un[j + 1] = x; done[j + 1] = dx;
    }
    /* collapse overlapping hits (:260-296) */
    int n_out = 0, have_cur = 0; osp_region_hit cur; memset(&cur, 0, sizeof(cur)); uint8_t cur_done = 0;
    osp_region_hit* coll = (osp_region_hit*)malloc(sizeof(osp_region_hit) * (size_t)(n_un + 1));
    uint8_t* coll_done = (uint8_t*)calloc((size_t)n_un + 1, 1);
    int n_coll = 0;
    for (int i = 0; i < n_un; ++i) {
        if (!have_cur) { cur = un[i]; cur_done = done[i]; have_cur = 1; continue; }
        if (osp_cyp_overlap_score(un[i].start, un[i].end, cur.start, cur.end) > 0.9) {
            int star5_pairing = is_penalized_type(tmpl_type[un[i].template_idx]) || is_penalized_type(tmpl_type[cur.template_idx]);
            int penalized_scoring = star5_pairing ? 1 : 0;
            int up = tmpl_type[un[i].template_idx] == OSP_DELETION ? 1 : 0, cp = tmpl_type[cur.template_idx] == OSP_DELETION ? 1 : 0;   /* :897-902 */
            if ((hit_score(&un[i], penalized_scoring) < hit_score(&cur, penalized_scoring) && up >= cp) || up > cp) { cur = un[i]; cur_done = done[i]; }
        } else { coll_done[n_coll] = cur_done; coll[n_coll++] = cur; cur = un[i]; cur_done = done[i]; }
    }
    if (have_cur) { coll_done[n_coll] = cur_done; coll[n_coll++] = cur; }
    if (rescore) {
        int m = 0;
        for (int i = 0; i < n_coll; ++i) {
            if (!coll_done[i]) {
                k3_rescore_hit(seq, seq_len, tmpl, tmpl_len, &coll[i]);
                if (hit_score(&coll[i], is_penalized_type(tmpl_type[coll[i].template_idx])) > max_ed_frac) continue;       /* (the filter once more, on the re-scored numbers) */
            }
            coll[m++] = coll[i];
        }
        n_coll = m;
    }
    for (int i = 0; i < n_coll; ++i) {
        if (hit_score(&coll[i], 1) > max_missing_frac) continue;                                   /* :303-306 */
        if (n_out < cap) out[n_out] = coll[i];
        ++n_out;
    }
    free(un); free(coll); free(done); free(coll_done);
    return n_out;
}

/* assign_haplotype scoring loop (src/cyp2d6/haplotyper.rs:470-524): best (vi_match, all_match) starting from the (0,0)
 * of the Unknown label; tie[a] = 1 for the alleles in best_id_set */
void osp_cyp_score_alleles(int n_variants, int n_alleles, const uint8_t* hap_matrix, const uint8_t* is_vi, const uint8_t* states,
                           uint32_t* best_vi, uint32_t* best_all, uint8_t* tie) {
    uint32_t bv = 0, ba = 0;
    for (int a = 0; a < n_alleles; ++a) tie[a] = 0;
    for (int a = 0; a < n_alleles; ++a) {
        uint32_t vi_match = 0, all_match = 0;
        for (int i = 0; i < n_variants; ++i) {
            uint8_t seq_value = states[i], hap_value = hap_matrix[(size_t)a * n_variants + i];
            int is_match = (seq_value == 0 || seq_value == 1) ? hap_value == seq_value : seq_value == 2;
            if (is_match) { all_match += 1; if (is_vi[i]) vi_match += 1; }
        }
        if (vi_match > bv || (vi_match == bv && all_match > ba)) {          /* Ordering::Greater: clear + insert */
            for (int x = 0; x < n_alleles; ++x) tie[x] = 0;
            tie[a] = 1; bv = vi_match; ba = all_match;
        } else if (vi_match == bv && all_match == ba) tie[a] = 1;           /* Ordering::Equal: insert (joins Unknown at (0,0)) */
    }
    *best_vi = bv; *best_all = ba;
}


/* ------------------------------------------------------------------ chain building (src/cyp2d6/caller.rs:429-583)
 * Follows the reference loop literally: putative chains are extended segment by segment with every consensus at the
 * minimum edit distance (:461-484); a unique minimum adds one to that consensus' count per chain being extended (:478-481);
 * reads that end with no chain are not recorded (:494-517); chains touching a consensus with no unique support are
 * removed afterwards (:521-538); consensuses with no unique support become FalseAllele (:574-583). */
typedef struct { uint32_t n, cap; uint32_t* len; uint32_t** v; } chainset;
static void cs_push(chainset* c, const uint32_t* base, uint32_t n, uint32_t extra) {
    if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 8; c->len = realloc(c->len, c->cap * sizeof *c->len); c->v = realloc(c->v, c->cap * sizeof *c->v); }
    c->v[c->n] = malloc((n + 1) * sizeof(uint32_t));
    if (n) memcpy(c->v[c->n], base, n * sizeof(uint32_t));
    c->v[c->n][n] = extra; c->len[c->n] = n + 1; c->n++;
}
static void cs_free(chainset* c) { for (uint32_t i = 0; i < c->n; ++i) free(c->v[i]); free(c->len); free(c->v); memset(c, 0, sizeof *c); }

int osp_cyp_build_chains(int n_haps, const int32_t* hap_type, int n_reads, const uint32_t* read_seg_off, const uint64_t* ed,
                         const uint8_t* kept, uint32_t* read_index, uint32_t* n_kept, uint32_t* read_chain_off, uint32_t* chain_off,
                         uint32_t chain_cap, uint32_t* chain_items, uint32_t item_cap, uint32_t* read_w_off, uint32_t* w_seg,
                         uint64_t* unique_counts, uint8_t* false_allele) {
    chainset* per_read = calloc((size_t)n_reads, sizeof *per_read);
    uint32_t nk = 0, nrow = 0;
    int rc = 0;
    for (int h = 0; h < n_haps; ++h) unique_counts[h] = 0;
    read_w_off[0] = 0;
    for (int r = 0; r < n_reads; ++r) {
        if (read_seg_off[r] == read_seg_off[r + 1]) continue;                       /* :437-439 */
        chainset cur = {0};
        uint32_t none = 0;
        cs_push(&cur, &none, 0, 0); cur.len[0] = 0;                                  /* vec![vec![]] */
        const uint32_t row0 = nrow;
        for (uint32_t sgm = read_seg_off[r]; sgm < read_seg_off[r + 1]; ++sgm) {
            if (!kept[sgm]) continue;                                                /* :451-460 */
            const uint64_t* w = ed + (size_t)sgm * n_haps;
            uint64_t mn = w[0]; int nmin = 0;
            for (int c = 1; c < n_haps; ++c) if (w[c] < mn) mn = w[c];
            for (int c = 0; c < n_haps; ++c) nmin += w[c] == mn;
            chainset nxt = {0};
            for (uint32_t pc = 0; pc < cur.n; ++pc)
                for (int c = 0; c < n_haps; ++c)
                    if (w[c] == mn) {
                        cs_push(&nxt, cur.v[pc], cur.len[pc], (uint32_t)c);
                        if (nmin == 1) unique_counts[c] += 1;
                    }
            cs_free(&cur); cur = nxt;
            w_seg[nrow++] = sgm;
        }
        if (cur.n == 0 || (cur.n == 1 && cur.len[0] == 0)) { cs_free(&cur); nrow = row0; continue; }   /* :494-496: nothing recorded */
        per_read[r] = cur;
        read_index[nk++] = (uint32_t)r;
        read_w_off[nk] = nrow;
    }
    /* unique-support filter (:521-538) and flattening */
    uint32_t nc = 0, ni = 0;
    read_chain_off[0] = 0; chain_off[0] = 0;
    for (uint32_t k = 0; k < nk && rc == 0; ++k) {
        chainset* c = &per_read[read_index[k]];
        uint32_t kept_chains = 0;
        for (uint32_t i = 0; i < c->n && rc == 0; ++i) {
            int ok = 1;
            for (uint32_t x = 0; x < c->len[i]; ++x) if (unique_counts[c->v[i][x]] == 0) ok = 0;
            if (!ok) continue;
            if (nc + 1 > chain_cap || ni + c->len[i] > item_cap) { rc = 1; break; }
            memcpy(chain_items + ni, c->v[i], c->len[i] * sizeof(uint32_t));
            ni += c->len[i]; chain_off[++nc] = ni; ++kept_chains;
        }
        if (rc == 0 && kept_chains == 0) rc = 2;                                      /* panic!("chain collapse") */
        read_chain_off[k + 1] = nc;
    }
    for (int h = 0; h < n_haps; ++h)
        false_allele[h] = unique_counts[h] == 0 && hap_type[h] != OSP_UNKNOWN && hap_type[h] != OSP_FALSE_ALLELE;
    for (int r = 0; r < n_reads; ++r) cs_free(&per_read[r]);
    free(per_read);
    *n_kept = nk;
    return rc;
}


/* ------------------------------------------------------------------ variant states of a sequence on the backbone (see cyp_oracle.h) */
static int code_of(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

/* The variant graph of the aligned backbone part [gs, ge): reference stretches and SITES.  A site is a maximal run of variants whose
 * reference spans overlap; its alternatives are the subsets of its variants that do not overlap one another (the empty subset is the
 * reference), each one the site's span with those variants applied. */
#define K9_SITE_MAX 8            /* variants per site */
#define K9_DIAGS 256             /* diagonals of the alignment (read offset - graph offset), centred on the drift of the linear placement */
#define K9_INF 30000
typedef struct { int s, e, nv, var[K9_SITE_MAX]; } k9_site;
typedef struct { int site, mask, len; uint8_t* seq; } k9_alt;

/* one base of the graph: col is indexed by diagonal d <-> k = k0 + d, read position i = off + k (off = graph offset before the base) */
static void k9_base(int16_t* col, int x, const uint8_t* seq, int L, int off, int k0) {
    int16_t nw[K9_DIAGS];
    for (int d = 0; d < K9_DIAGS; ++d) {
        const int i = off + k0 + d;                       /* read position before the base on this diagonal */
        int v = K9_INF;
        if (i >= 0 && i < L && col[d] < K9_INF) v = col[d] + ((seq[i] < 4 && seq[i] == x) ? 0 : 1);          /* the base faces read base i */
        if (d + 1 < K9_DIAGS && col[d + 1] < K9_INF && i + 1 >= 0 && i + 1 <= L) { const int w = col[d + 1] + 1; if (w < v) v = w; }   /* the base faces nothing */
        nw[d] = (int16_t)v;
    }
    /* read bases that face nothing: along the column, from lower to higher diagonals */
    for (int d = 1; d < K9_DIAGS; ++d) { const int i = off + 1 + k0 + d; if (i >= 0 && i <= L && nw[d - 1] + 1 < nw[d]) nw[d] = (int16_t)(nw[d - 1] + 1); }
    memcpy(col, nw, sizeof nw);
}

/* forward pass over the graph: exit[a] = the column at the site's end after alternative a (real diagonals), entry[site] = the column
 * in front of the site; returns the final column in col */
static void k9_forward(const uint8_t* bb, int gs, int ge, const k9_site* sites, int n_sites, const k9_alt* alts, const int* alt_first,
                       const uint8_t* seq, int L, int k0, int16_t* col, int16_t* entry, int16_t* exit_cols) {
    for (int d = 0; d < K9_DIAGS; ++d) { const int i = k0 + d; col[d] = (int16_t)((i >= 0 && i <= L) ? i : K9_INF); }
    int b = gs;
    for (int si = 0; si <= n_sites; ++si) {
        const int stop = si < n_sites ? sites[si].s : ge;
        for (; b < stop; ++b) k9_base(col, bb[b], seq, L, b - gs, k0);
        if (si == n_sites) break;
        memcpy(entry + (size_t)si * K9_DIAGS, col, sizeof(int16_t) * K9_DIAGS);
        int16_t acc[K9_DIAGS];
        for (int d = 0; d < K9_DIAGS; ++d) acc[d] = K9_INF;
        const int lr = sites[si].e - sites[si].s;
        for (int a = alt_first[si]; a < alt_first[si + 1]; ++a) {
            int16_t w[K9_DIAGS];
            memcpy(w, entry + (size_t)si * K9_DIAGS, sizeof w);
            for (int x = 0; x < alts[a].len; ++x) k9_base(w, alts[a].seq[x], seq, L, b - gs + x, k0);
            /* the alternative is alts[a].len bases where the reference has lr: its diagonals shift by len - lr at the site's end */
            const int delta = alts[a].len - lr;
            int16_t* out = exit_cols + (size_t)a * K9_DIAGS;
            for (int d = 0; d < K9_DIAGS; ++d) { const int src = d - delta; out[d] = (src >= 0 && src < K9_DIAGS) ? w[src] : K9_INF; }
            for (int d = 0; d < K9_DIAGS; ++d) if (out[d] < acc[d]) acc[d] = out[d];
        }
        memcpy(col, acc, sizeof acc);
        b = sites[si].e;
    }
}

int osp_cyp_variant_states(const uint8_t* seq, int seq_len, const uint8_t* backbone, int backbone_len, int n_variants, const int32_t* var_pos,
                           const char* const* var_ref, const char* const* var_alt, uint8_t* states, int32_t* aln_out) {
    for (int v = 0; v < n_variants; ++v) states[v] = 3;
    if (aln_out) memset(aln_out, 0, sizeof(int32_t) * 5);
    int diag = 0;
    if (osp_anchor(seq, seq_len, backbone, backbone_len, &diag) < OSP_CYP_MIN_VOTES) return 0;      /* backbone_pos - seq_pos */
    osp_aln al; uint32_t ev[OSP_MAX_ED + 1]; int nev = 0;
    if (!osp_wfa_retry2(seq, seq_len, backbone, backbone_len, diag, OSP_MAX_ED, &al, ev, &nev)) return 0;
    if (aln_out) { aln_out[0] = al.a_start; aln_out[1] = al.a_end; aln_out[2] = al.b_start; aln_out[3] = al.b_end; aln_out[4] = al.nm; }
    /* the band: centred on the middle of the drift the linear placement shows (insertions push the read on, deletions pull it back) */
    int drift = 0, dmin = 0, dmax = 0;
    for (int e = 0; e < nev; ++e) {
        const uint32_t type = ev[e] >> 30;
        if (type == OSP_EV_I) ++drift; else if (type == OSP_EV_D) --drift;
        if (drift < dmin) dmin = drift;
        if (drift > dmax) dmax = drift;
    }
    return osp_cyp_variant_states_at(seq, seq_len, backbone, backbone_len, n_variants, var_pos, var_ref, var_alt, al.a_start, al.a_end, al.b_start, al.b_end,
                                     (dmin + dmax) / 2, states);
}

/* the graph half of the above for a GIVEN placement (sequence [a_start, a_end) on backbone [b_start, b_end), band centred on drift_centre): what
 * tests/cpu_port_cyp.py calls with the placement of the minimap2 restatement (omm_cyp_place), as assign_haplotype does with minimap2's */
int osp_cyp_variant_states_at(const uint8_t* seq, int seq_len, const uint8_t* backbone, int backbone_len, int n_variants, const int32_t* var_pos,
                              const char* const* var_ref, const char* const* var_alt, int a_start, int a_end, int b_start, int b_end, int drift_centre,
                              uint8_t* states) {
    (void)seq_len;
    for (int v = 0; v < n_variants; ++v) states[v] = 3;
    const int gs = b_start, ge = b_end, L = a_end - a_start;
    const uint8_t* S = seq + a_start;
    const int k0 = drift_centre - K9_DIAGS / 2;
    /* sites */
    int* order = (int*)malloc(sizeof(int) * (size_t)(n_variants + 1)); int no = 0;
    for (int v = 0; v < n_variants; ++v) { const int rl = (int)strlen(var_ref[v]); if (var_pos[v] >= gs && var_pos[v] + rl <= ge) order[no++] = v; }
    for (int a = 1; a < no; ++a) { const int v = order[a]; int b = a - 1; while (b >= 0 && (var_pos[order[b]] > var_pos[v] || (var_pos[order[b]] == var_pos[v] && order[b] > v))) { order[b + 1] = order[b]; --b; } order[b + 1] = v; }
    k9_site* sites = (k9_site*)malloc(sizeof(k9_site) * (size_t)(no + 1)); int n_sites = 0;
    for (int a = 0; a < no;) {
        k9_site st; st.s = var_pos[order[a]]; st.e = st.s + (int)strlen(var_ref[order[a]]); st.nv = 0;
        while (a < no && var_pos[order[a]] < st.e) {
            const int end = var_pos[order[a]] + (int)strlen(var_ref[order[a]]);
            if (st.nv < K9_SITE_MAX) { if (end > st.e) st.e = end; st.var[st.nv++] = order[a]; }      /* (a ninth overlapping variant stays undecided) */
            ++a;
        }
        sites[n_sites++] = st;
    }
    /* alternatives */
    int* alt_first = (int*)malloc(sizeof(int) * (size_t)(n_sites + 1));
    int cap_alts = 0; for (int si = 0; si < n_sites; ++si) cap_alts += 1 << sites[si].nv;
    k9_alt* alts = (k9_alt*)malloc(sizeof(k9_alt) * (size_t)(cap_alts + 1)); int na = 0;
    for (int si = 0; si < n_sites; ++si) {
        alt_first[si] = na;
        const k9_site* st = &sites[si];
        for (int mask = 0; mask < (1 << st->nv); ++mask) {
            /* the chosen variants must not overlap one another (they are in position order) */
            int ok = 1, last_end = -1;
            for (int x = 0; x < st->nv && ok; ++x) if (mask >> x & 1) { const int v = st->var[x]; if (var_pos[v] < last_end) ok = 0; last_end = var_pos[v] + (int)strlen(var_ref[v]); }
            if (!ok) continue;
            k9_alt A; A.site = si; A.mask = mask; A.seq = (uint8_t*)malloc((size_t)(st->e - st->s) + 64 * (size_t)st->nv + 64); A.len = 0;
            int b = st->s;
            for (int x = 0; x < st->nv; ++x) if (mask >> x & 1) {
                const int v = st->var[x];
                while (b < var_pos[v]) A.seq[A.len++] = backbone[b++];
                for (const char* q = var_alt[v]; *q; ++q) A.seq[A.len++] = (uint8_t)code_of(*q);
                b = var_pos[v] + (int)strlen(var_ref[v]);
            }
            while (b < st->e) A.seq[A.len++] = backbone[b++];
            alts[na++] = A;
        }
    }
    alt_first[n_sites] = na;
    /* forward over the graph, forward over the mirrored graph (= backward) */
    int16_t col[K9_DIAGS];
    int16_t* entry = (int16_t*)malloc(sizeof(int16_t) * K9_DIAGS * (size_t)(n_sites + 1));
    int16_t* exits = (int16_t*)malloc(sizeof(int16_t) * K9_DIAGS * (size_t)(na + 1));
    k9_forward(backbone, gs, ge, sites, n_sites, alts, alt_first, S, L, k0, col, entry, exits);
    const int kL = L - (ge - gs) - k0;                                          /* diagonal of the end point */
    const int opt = (kL >= 0 && kL < K9_DIAGS) ? col[kL] : K9_INF;
    if (opt >= K9_INF) { /* the band lost the path: nothing is decided */
        for (int a = 0; a < na; ++a) free(alts[a].seq);
        free(order); free(sites); free(alt_first); free(alts); free(entry); free(exits);
        return 1;
    }
    /* mirror: graph offset g' = (ge - gs) - g, read position i' = L - i, so diagonal k' = i' - g' = kL_abs - k with kL_abs = L - (ge - gs) */
    const int G = ge - gs, kabs = L - G;
    uint8_t* rbb = (uint8_t*)malloc((size_t)backbone_len + 1), *rS = (uint8_t*)malloc((size_t)L + 1);
    for (int i = 0; i < L; ++i) rS[i] = S[L - 1 - i];
    for (int g = 0; g < G; ++g) rbb[g] = backbone[ge - 1 - g];
    k9_site* rsites = (k9_site*)malloc(sizeof(k9_site) * (size_t)(n_sites + 1));
    k9_alt* ralts = (k9_alt*)malloc(sizeof(k9_alt) * (size_t)(na + 1));
    int* ralt_first = (int*)malloc(sizeof(int) * (size_t)(n_sites + 1));
    int rna = 0;
    for (int si = 0; si < n_sites; ++si) {
        const k9_site* st = &sites[n_sites - 1 - si];
        rsites[si] = *st; rsites[si].s = ge - st->e; rsites[si].e = ge - st->s;       /* offsets in the mirrored backbone (which starts at 0) */
        ralt_first[si] = rna;
        for (int a = alt_first[n_sites - 1 - si]; a < alt_first[n_sites - si]; ++a) {
            k9_alt A = alts[a]; A.seq = (uint8_t*)malloc((size_t)A.len + 1);
            for (int x = 0; x < A.len; ++x) A.seq[x] = alts[a].seq[A.len - 1 - x];
            ralts[rna++] = A;
        }
    }
    ralt_first[n_sites] = rna;
    /* the mirrored band: diagonal index d' of k' = kabs - k; k = k0 + d  =>  k' = kabs - k0 - d: choose k0' so that d' = K9_DIAGS - 1 - d */
    const int rk0 = kabs - k0 - (K9_DIAGS - 1);
    int16_t rcol[K9_DIAGS];
    int16_t* rentry = (int16_t*)malloc(sizeof(int16_t) * K9_DIAGS * (size_t)(n_sites + 1));
    int16_t* rexits = (int16_t*)malloc(sizeof(int16_t) * K9_DIAGS * (size_t)(na + 1));
    k9_forward(rbb, 0, G, rsites, n_sites, ralts, ralt_first, rS, L, rk0, rcol, rentry, rexits);
    /* alternative a of site si lies on an optimal path iff some diagonal has exit_a + (cost of the rest from the site's end) == opt; the
     * rest from the end of site si is the mirrored pass's column in FRONT of mirrored site n_sites - 1 - si */
    for (int si = 0; si < n_sites; ++si) {
        const int16_t* back = rentry + (size_t)(n_sites - 1 - si) * K9_DIAGS;
        int seen1[K9_SITE_MAX] = {0}, seen0[K9_SITE_MAX] = {0};
        for (int a = alt_first[si]; a < alt_first[si + 1]; ++a) {
            const int16_t* ex = exits + (size_t)a * K9_DIAGS;
            int on = 0;
            for (int d = 0; d < K9_DIAGS && !on; ++d) { const int bd = K9_DIAGS - 1 - d; if (ex[d] < K9_INF && back[bd] < K9_INF && ex[d] + back[bd] == opt) on = 1; }
            if (!on) continue;
            for (int x = 0; x < sites[si].nv; ++x) { if (alts[a].mask >> x & 1) seen1[x] = 1; else seen0[x] = 1; }
        }
        for (int x = 0; x < sites[si].nv; ++x) states[sites[si].var[x]] = (seen1[x] && seen0[x]) ? 2 : seen1[x] ? 1 : seen0[x] ? 0 : 3;
    }
    for (int a = 0; a < na; ++a) { free(alts[a].seq); free(ralts[a].seq); }
    free(order); free(sites); free(alt_first); free(alts); free(entry); free(exits);
    free(rbb); free(rS); free(rsites); free(ralts); free(ralt_first); free(rentry); free(rexits);
    return 1;
}
