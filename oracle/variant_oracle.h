/* variant_oracle.h -- CPU ORACLE (test infrastructure only): variant-gene diplotype search interface (oracle/variant.c). */
#ifndef VARIANT_ORACLE_H
#define VARIANT_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OSP_VAR_MAXLEN  4096
#define OSP_VAR_MAXTIES 64
#define OSP_VAR_MAXDIP  4096
/* Genotype (src/data_types/normalized_variant.rs:303-315) */
enum { OSP_GT_HOM_REF = 0, OSP_GT_HET_UNPHASED = 1, OSP_GT_HET_PHASED = 2, OSP_GT_HET_FLIP = 3, OSP_GT_HOM_ALT = 4 };

typedef struct { char chrom[64]; int64_t position; char ref[OSP_VAR_MAXLEN]; char alt[OSP_VAR_MAXLEN]; } osp_norm_variant;

int osp_normalize_variant(const char* chrom, int64_t position, const char* ref_allele, const char* alt_allele,
                          const char* chrom_seq, int64_t chrom_len, osp_norm_variant* out, char* err, size_t errcap);

typedef struct {
    int32_t n_haps;                    /* haplotypes in defined_haplotypes (BTreeMap name) order */
    const uint8_t* hap_is_sv;
    const uint8_t* hap_is_core;        /* core_allele().is_none() */
    const int32_t* slot_off;           /* n_haps+1 : AND-slots of haplotype h */
    const int32_t* alt_off;            /* n_slots+1 : OR-alternatives of a slot */
    const int32_t* alt_var;            /* variant id, or -1 for None */
    int32_t n_vars;
    const uint8_t* var_is_core;        /* variant_hash[..].is_core_variant */
    int32_t n_obs;                     /* observed variants in BTreeMap<NormalizedVariant> order */
    const int32_t* obs_var;
    const int32_t* obs_gt;
    const int64_t* obs_ps;             /* phase set or -1 */
    const int32_t* obs_sv_label;       /* -1, or the id of the SV haplotype label */
} osp_variant_problem;

typedef struct {
    int64_t score[4];                  /* core missing, core extra, sub missing, sub extra */
    int32_t is_sv, sv_label, n_sv_extra, sv_extra[OSP_VAR_MAXTIES];
    int32_t n_best, best[OSP_VAR_MAXTIES];
} osp_inexact;

typedef struct {
    int64_t score[4];
    int32_t n_dip, overflow;
    int32_t dip[OSP_VAR_MAXDIP][2];    /* haplotype index, or -(label+2) for an SV label */
    int32_t dip_comb[OSP_VAR_MAXDIP];  /* het assignment (combination) the pair came from */
} osp_variant_result;

void osp_quant_match(const osp_variant_problem* p, int h, const int32_t* obs, int n_obs,
                     int32_t* matching, int* n_match, int32_t* missing, int* n_missing, int32_t* extra, int* n_extra);
void osp_find_best_inexact(const osp_variant_problem* p, const int32_t* obs, const int32_t* obs_sv_label, int n_obs, osp_inexact* out);
int  osp_solve_diplotype(const osp_variant_problem* p, osp_variant_result* res);

/* gene collection + structural variant definitions of one gene entry, flattened (same layout as sp_sv_definitions) */
typedef struct {
    int32_t n_genes;
    const int64_t* gene_start; const int64_t* gene_end; const uint8_t* gene_forward;
    const int32_t* exon_off; const int64_t* exon_start; const int64_t* exon_end;
    int32_t n_full; const uint8_t* full_generic; const int32_t* full_off; const int32_t* full_gene;
    int32_t n_partial; const uint8_t* partial_generic; const int32_t* partial_off; const int32_t* partial_gene;
    const int32_t* partial_first; const int32_t* partial_end;
} osp_sv_definitions;
/* 0 ok (kind 0 none / 1 full / 2 partial, index within the class), -1 = a definition names a gene without a gene definition */
int osp_is_deletion(const osp_sv_definitions* d, uint64_t start, uint64_t end, int32_t* kind, int32_t* index);

#ifdef __cplusplus
}
#endif
#endif
