/* cyp_oracle.h -- CPU ORACLE (test infrastructure only): CYP2D6 chain search interface (see oracle/cyp.c). */
#ifndef CYP_ORACLE_H
#define CYP_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Cyp2d6RegionType (src/cyp2d6/region_label.rs:7-24) */
enum { OSP_UNKNOWN = 0, OSP_REP6 = 1, OSP_CYP2D6 = 2, OSP_LINK = 3, OSP_REP7 = 4, OSP_SPACER = 5, OSP_CYP2D7 = 6,
       OSP_DELETION = 7, OSP_HYBRID = 8, OSP_FALSE_ALLELE = 9 };
enum { OSP_CHAIN_OK = 0, OSP_CHAIN_BAD_ARG = 1, OSP_CHAIN_TOO_LONG = 5,
       OSP_CHAIN_NO_HEAD = 16, OSP_CHAIN_NO_CHAINS = 17, OSP_CHAIN_NO_PAIRS = 18 };   /* CallerError (src/cyp2d6/errors.rs:4-11) */
#define OSP_MAX_CHAIN 64

typedef struct {                                   /* the three tables of Cyp2d6Config the search reads (definitions.rs:242-301) */
    int32_t n_translate; const char* const* tr_key; const char* const* tr_val;      /* cyp_translate */
    int32_t n_conn; const char* const* conn_a; const char* const* conn_b;           /* inferred_connections */
    int32_t n_single; const char* const* singles;                                    /* unexpected_singletons */
} osp_cyp_config;

typedef struct {
    int32_t n_haps;
    const int32_t* type;                           /* region type per consensus region */
    const char* const* subtype;                    /* subtype label or NULL */
    osp_cyp_config cfg;
    int32_t n_reads;                               /* reads in BTreeMap (qname) order */
    const int32_t* read_chain_off;                 /* n_reads+1 : chains of read r = [off[r], off[r+1]) */
    const int32_t* chain_off;                      /* n_chains+1 into chain_items */
    const int32_t* chain_items;
    const int32_t* read_w_off;                     /* n_reads+1 : weight rows of read r */
    const uint64_t* w_ed;                          /* [row][n_haps] edit distance */
    const double* w_ov;                            /* [row][n_haps] overlap score */
    int32_t infer_connections, normalize_all_alleles, ignore_chain_label_limits;
    double lasso, ln_ed, unexpected, inferred;     /* ChainPenalties (chaining.rs:107-139) */
} osp_chain_problem;

typedef struct {
    int32_t status;
    int32_t n_possible;                            /* number of enumerated chains */
    int32_t index1, index2;                        /* winning pair (i <= j) in enumeration order */
    int32_t n1, n2;
    int32_t chain1[OSP_MAX_CHAIN], chain2[OSP_MAX_CHAIN];   /* sorted pair */
    double score, ln_ed_penalty, mn_llh_penalty, allele_expected_penalty, unexpected_chain_penalty, inferred_chain_penalty;
    uint64_t edit_distance, unmet_observations;
} osp_chain_result;

void osp_cyp_full_allele(int type, const char* subtype, char* out, size_t cap);
void osp_cyp_simplify_allele(int type, const char* subtype, int detailed, const osp_cyp_config* cfg, char* out, size_t cap);
int  osp_cyp_is_allowed_label_pair(int type1, int type2);
void osp_cyp_convert_chain_to_hap(const int32_t* chain, int n, const int32_t* type, const char* const* subtype, int detail,
                                  const osp_cyp_config* cfg, char* out, size_t cap);
int  osp_cyp_find_best_chain_pair(const osp_chain_problem* p, osp_chain_result* res);

#define OSP_CYP_TOPK      4          /* placements tried per (template, read) */
#define OSP_CYP_MIN_VOTES 4          /* an anchor needs this many 16-mer votes */
#define OSP_CYP_K4_NEAR 16           /* weight_sequence: placements within this many (edits + unmapped bases) of the segment's smallest are re-scored the reference's way */
typedef struct { int32_t template_idx, start, end, seq_len, nm, unmapped, clip_start, clip_end; } osp_region_hit;   /* AlleleMapping */
int  osp_cyp_weight_sequence(const uint8_t* seq, int seq_len, int n_cons, const uint8_t* const* cons, const int32_t* cons_len,
                             const uint8_t* allowed, uint64_t* out_ed, double* out_ov);
int  osp_cyp_find_base_type(const uint8_t* seq, int seq_len, int n_templates, const uint8_t* const* tmpl, const int32_t* tmpl_len,
                            const int32_t* tmpl_type, double max_missing_frac, osp_region_hit* out, int cap);
int  osp_cyp_find_base_type_ex(const uint8_t* seq, int seq_len, int n_templates, const uint8_t* const* tmpl, const int32_t* tmpl_len,
                               const int32_t* tmpl_type, double max_missing_frac, int rescore, osp_region_hit* out, int cap);   /* rescore: the hits carry minimap2's numbers (two-piece affine re-score) */

double osp_cyp_overlap_score(int s1, int e1, int s2, int e2);          /* overlap_score (src/cyp2d6/haplotyper.rs:877-892) */
void osp_cyp_score_alleles(int n_variants, int n_alleles, const uint8_t* hap_matrix, const uint8_t* is_vi, const uint8_t* states,
                           uint32_t* best_vi, uint32_t* best_all, uint8_t* tie);

/* per-variant state of a sequence against the CYP2D6 backbone: the role of WFAGraph::edit_distance_with_pruning + traversed nodes in
 * assign_haplotype (src/cyp2d6/haplotyper.rs:371-468; hiphase v1.2.1 is not on disk -- PARITY UNPINNED, contract in DESIGN.md 10).
 * The sequence is placed on the backbone (anchor + banded alignment with traceback); for every variant whose reference span lies in
 * the aligned part, the sequence window that faces [p - 24, p + |ref| + 24) is compared (global edit distance) with that backbone
 * window carrying the reference allele and carrying the alternate allele: closer to ref = 0, closer to alt = 1, equal = 2;
 * variants outside the aligned part stay 3.  var_pos is 0-based on the backbone; alleles are ACGT strings (normalised, with anchor
 * base for indels).  Returns 1 when the sequence aligned (aln_out filled), 0 otherwise (all states 3). */
#define OSP_K9_FLANK 24
int osp_cyp_variant_states(const uint8_t* seq, int seq_len, const uint8_t* backbone, int backbone_len, int n_variants, const int32_t* var_pos,
                           const char* const* var_ref, const char* const* var_alt, uint8_t* states, int32_t* aln_out /* a_start, a_end, b_start, b_end, nm */);

int osp_cyp_variant_states_at(const uint8_t* seq, int seq_len, const uint8_t* backbone, int backbone_len, int n_variants, const int32_t* var_pos,
                              const char* const* var_ref, const char* const* var_alt, int a_start, int a_end, int b_start, int b_end, int drift_centre,
                              uint8_t* states);            /* the graph half of osp_cyp_variant_states for a given placement */

/* chain building (src/cyp2d6/caller.rs:429-583).  Segments of read r = [read_seg_off[r], read_seg_off[r+1]) in region order;
 * ed = [segment][n_haps] from weight_sequence, kept[segment] = 0 when weight_sequence returned the empty vector.
 * Outputs (all caller-sized; return 0 ok, 1 capacity, 2 "chain collapse" panic):
 *   read_index[k]   the reads that enter qname_chains, in input (BTreeMap) order; *n_kept of them
 *   read_chain_off / chain_off / chain_items   their chain sets after the unique-support filter
 *   read_w_off / w_seg   rows of qname_chain_scores as segment indices
 *   unique_counts[h]     best_allele_mapping_counts;  false_allele[h] = 1 when the region gets mark_false_allele() */
int osp_cyp_build_chains(int n_haps, const int32_t* hap_type, int n_reads, const uint32_t* read_seg_off, const uint64_t* ed,
                         const uint8_t* kept, uint32_t* read_index, uint32_t* n_kept, uint32_t* read_chain_off, uint32_t* chain_off,
                         uint32_t chain_cap, uint32_t* chain_items, uint32_t item_cap, uint32_t* read_w_off, uint32_t* w_seg,
                         uint64_t* unique_counts, uint8_t* false_allele);

#ifdef __cplusplus
}
#endif
#endif
