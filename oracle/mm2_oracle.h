/*
 * mm2_oracle.h -- CPU ORACLE, second statement of the base-level aligner.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference takes every (nm, start, end, strand, cigar) from minimap2 2.28 through the crate
 * `minimap2 0.1.23+minimap2.2.28` (Cargo.lock:1137-1152); `standard_hifi_aligner`
 * (src/util/mapping.rs:8-14) fixes the configuration: preset `map-hifi`, CIGAR on, `best_n = 5`;
 * `score_read` raises the match score to 5 (src/hla/caller.rs:1370-1379).  minimap2's sources are not under
 * /root/reference and cannot be fetched; this file restates its PUBLISHED algorithm (Li 2018, Bioinformatics
 * 34:3094; Li 2021, Bioinformatics 37:4572; Suzuki & Kasahara 2018 for the difference-recurrence DP whose plain
 * form is used here) with the documented `map-hifi` parameters:
 *
 *   seeding   (k,w) = (19,19) minimizers of the invertible 64-bit integer hash, both strands, occurrence
 *             filter mid_occ in [50,500] from the top 2e-4 fraction, rescue of high-occurrence streaks
 *             (<= 4095 occurrences, one seed per 500 query bases)
 *   chaining  f(i) = max_j f(j) + min(span, gap) - (0.01*0.8*k*|dd| + 0.5*log2(|dd|+1)); max gap 10,000;
 *             band 500; skip 25; at least 3 seeds and score 40; chains cut where the score falls by > band
 *   selection chains overlapping >= 0.5 of the shorter on the QUERY are secondary to the better one;
 *             secondaries within 0.8 of their primary, at most best_n, are kept and base-aligned
 *   alignment two-piece affine gaps  min(q + k*e, q2 + k*e2) = min(6+2k, 26+k), match a = 1 (5 in score_read),
 *             mismatch b = 4, ambiguous base -1; global DP between seeds in stretches >= 200 bases, extension
 *             from the outermost seeds with z-drop 400 ending at the best-scoring cell (end bonus off): a mismatch
 *             within b/a bases of an end is clipped; NM = mismatches + gap bases + ambiguous bases;
 *             alignments whose peak DP score is below 200 are dropped
 *
 * It is a "second opinion": parity of the library is defined against osp_wfa / osp_anchor;
 * this file measures how often that contract and minimap2's algorithm disagree (profiles/r03/aligner_divergence.*)
 * and provides the reference's CALL PATTERN for the CPU baseline (one seeded map of a read against the whole allele
 * index, base-level alignment of the best chains only).  It is pinned on the same reference-held cases as osp_wfa
 * (tests/test_oracle_mm2.py) and against a brute-force Gotoh DP; beyond them it is, like every statement of
 * minimap2 made without its sources, unpinned.
 */
#ifndef MM2_ORACLE_H
#define MM2_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t k, w;                          /* 19, 19 */
    int32_t a, b, q, e, q2, e2, sc_ambi;   /* 1, 4, 6, 2, 26, 1, 1 */
    int32_t zdrop, zdrop_inv, end_bonus;   /* 400, 200, -1 */
    int32_t bw, max_gap;                   /* 500, 10000 */
    int32_t min_cnt, min_chain_score;      /* 3, 40 */
    int32_t max_chain_skip, max_chain_iter;/* 25, 5000 */
    float   chain_gap_scale;               /* 0.8 */
    float   mask_level, pri_ratio;         /* 0.5, 0.8 */
    int32_t best_n;                        /* 5 */
    int32_t min_dp_max, min_ksw_len;       /* 200, 200 */
    int32_t min_mid_occ, max_mid_occ, max_max_occ, occ_dist;   /* 50, 500, 4095, 500 */
    float   mid_occ_frac;                  /* 2e-4 */
    int32_t forward_only;                  /* 0: both strands (minimap2); 1: skip the reverse strand (faster audits) */
} omm_opts;

typedef struct {
    int32_t rid, rev;                      /* target sequence; 1 = query's reverse complement aligned */
    int32_t q_start, q_end, q_len;         /* on the query as given (forward coordinates, as minimap2 reports) */
    int32_t t_start, t_end, t_len;
    int32_t nm, mlen, blen, n_ambi;        /* NM = blen - mlen + n_ambi */
    int32_t dp_score, dp_max, chain_score, n_seeds;
    int32_t primary;                       /* 1 = primary, 0 = secondary */
    int32_t n_cigar, cigar_off;            /* ops (len<<4|op; 7 '=', 8 'X', 1 'I', 2 'D') at cigar_pool[cigar_off..] */
} omm_hit;

typedef struct omm_index omm_index;

void omm_default_opts(omm_opts* o);        /* map-hifi + with_cigar + best_n 5 */

/* codes: A=0 C=1 G=2 T=3, anything else 4; offsets[n_seqs+1] into codes */
omm_index* omm_index_build(const uint8_t* codes, const int64_t* offsets, int32_t n_seqs, const omm_opts* o);
void       omm_index_free(omm_index* idx);
int32_t    omm_index_mid_occ(const omm_index* idx);
int64_t    omm_index_n_minimizers(const omm_index* idx);

/* Aligner::map: hits in minimap2's output order (primary first, then by peak DP score); returns the number written
 * (<= max_hits); cigar_pool may be NULL */
int32_t omm_map(const omm_index* idx, const uint8_t* q, int32_t qlen, const omm_opts* o,
                omm_hit* hits, int32_t max_hits, uint32_t* cigar_pool, int32_t cigar_cap);

/* `aligner.with_seq(target)` + `.map(query)` in one call */
int32_t omm_map_pair(const uint8_t* target, int32_t tlen, const uint8_t* q, int32_t qlen, const omm_opts* o,
                     omm_hit* hits, int32_t max_hits, uint32_t* cigar_pool, int32_t cigar_cap);

/* the stages of omm_map one by one (parity tests of the seeded K1 path): the (w,k)-minimizers of a sequence in position order (hash, end position of the
 * k-mer, strand of the smaller k-mer); returns their number (may exceed cap) */
int32_t omm_sketch(const uint8_t* s, int32_t len, const omm_opts* o, uint64_t* hash, int32_t* end_pos, uint8_t* strand, int32_t cap);
/* seeding + chaining + selection without the base-level alignment: regs [cap][10] = {rid, rev, chain score, seeds, qs, qe, rs, re, parent, kept rank (1-based
 * position in the list align_chain sees, 0 = not selected)} in chain_anchors' order; stats[8] = {minimizers, seeds in the index, seeds kept, anchors, distinct
 * (strand, target) pairs, chains, chains selected, mid_occ}; returns the number of chains */
int32_t omm_chain_stage(const omm_index* idx, const uint8_t* q, int32_t qlen, const omm_opts* o, int32_t* regs_out, int32_t cap, int64_t* stats);

/* the sorted anchors of omm_map: x = rev << 63 | rid << 32 | target end position, y = span << 32 | query end position on the mapped strand; returns their
 * number (may exceed cap) */
int64_t omm_anchors(const omm_index* idx, const uint8_t* q, int32_t qlen, const omm_opts* o, uint64_t* x, uint64_t* y, int64_t cap);

/* realign_record's seeded map as the library runs it in seeded mode (sp_hla_realign_reads with `k1_best_n` > 0): minimap2's seeding, chaining and selection
 * (this file), the library's unit-cost cell + two-piece affine re-score for the base-level alignment of the selected chains (at most OMM_SEED_SEL), then the
 * output order, second selection and acceptance loop (src/hla/realigner.rs:124-146).  hits: the mappings in output order; returns the accepted one or -1. */
#define OMM_SEED_SEL 16
typedef struct {
    int32_t rid, rev, chain_score, n_seeds, t_len;
    int32_t sel_rank;                        /* position among the selected chains (the order align_chain sees them) */
    int32_t diag;                            /* the cell's diagonal: query position - target position, midway between the outermost seeds */
    int32_t ok, cell_nm, a_start, a_end, b_start, b_end;   /* the unit-cost cell: A = target allele, B = query on the mapped strand */
    int32_t dp_max, nm, t_start, t_end, q_start, q_end;    /* the re-scored mapping (query in forward coordinates, as minimap2 reports) */
    int32_t primary;
} omm_seed_hit;
int32_t omm_hla_k1_seeded(const omm_index* idx, const uint8_t* q, int32_t qlen, const omm_opts* o, omm_seed_hit* hits /* OMM_SEED_SEL */, int32_t* n_hits, int32_t* n_chains);

/* the DP alone (tests): mode 0 = global, 1 = extension from (0,0) ending at the best cell.  out8 = {score, max,
 * max_t, max_q, zdropped, reach_end, t_end, q_end}; cigar ops written to cigar (cap entries), *n_cigar set */
void omm_dp(const uint8_t* t, int32_t tlen, const uint8_t* q, int32_t qlen, const omm_opts* o, int32_t band,
            int32_t mode, int32_t right_align, int32_t* out8, uint32_t* cigar, int32_t cap, int32_t* n_cigar);

/* brute-force two-piece affine global score (O(mn) Gotoh with five states), for the tests */
int32_t omm_global_score_bruteforce(const uint8_t* t, int32_t tlen, const uint8_t* q, int32_t qlen, const omm_opts* o);

/* score_read (src/hla/caller.rs:1411-1510) on this file's mappings: alleles in database order, level sequences as codes (NULL / 0 = absent);
 * pass opts with a = 5 (src/hla/caller.rs:1370-1379).  Returns the best index or -1; stats may be NULL ([(allele * 2 + level) * 3]). */
int32_t omm_hla_score_read(const uint8_t* cons_cdna, int32_t cdna_len, const uint8_t* cons_dna, int32_t dna_len, int32_t n_alleles,
                           const uint8_t* const* cdna, const int32_t* cdna_lens, const uint8_t* const* dna, const int32_t* dna_lens,
                           const omm_opts* o, int64_t* stats);

/* the CYP2D6 call sites in the reference's call pattern on this file's mappings (oracle/cyp_mm2.c; osp_region_hit is cyp_oracle.h's) */
int omm_cyp_weight_sequence(const uint8_t* seq, int seq_len, int n_cons, const uint8_t* const* cons, const int32_t* cons_len,
                            const uint8_t* allowed, const omm_opts* o, uint64_t* out_ed, double* out_ov);
int omm_cyp_place(const uint8_t* seq, int seq_len, const uint8_t* backbone, int backbone_len, const omm_opts* o, int32_t* out6);

#ifdef __cplusplus
}
#endif
#endif
