/* consensus_oracle.h -- CPU ORACLE (test infrastructure only): read consensus by dynamic wavefront alignment (see consensus.c). */
#ifndef CONSENSUS_ORACLE_H
#define CONSENSUS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t min_count;                 /* CdwfaConfig::min_count (3) */
    int32_t dual_max_ed_delta;         /* 100 */
    int32_t allow_early_termination;   /* reads may stop before the consensus does */
    int32_t allow_dual;                /* 0: ConsensusDWFA, 1: DualConsensusDWFA */
    int32_t offset_window;             /* 400 */
    int32_t offset_compare_length;     /* 50 */
    double  min_af;                    /* 0.10 */
    int32_t max_queue_size;            /* 20   (<= 0: these three take the values dwfa_config_from_cli / waffle_con's defaults give) */
    int32_t max_capacity_per_size;     /* 10 */
    int32_t max_nodes_wo_constraint;   /* 1000 */
    int32_t pad;
} osp_cons_config;

typedef struct {
    int32_t is_dual, len1, len2, split_at;
    int64_t gave_up, best_total;       /* gave_up = 1: no complete node was found (best_total is unused since the best-first search replaced the two-pass split policy) */
    int64_t nodes_expanded;
} osp_cons_result;

/* seqs: base codes 0..3 (4 = N).  offsets[r] = -1 (read starts with the consensus) or the consensus length at which the read is
 * placed (its start is searched in the offset_window bases before it).  cons1 / cons2: cap codes each.  score = -1 means None. */
int osp_consensus(int n_reads, const uint8_t* const* seqs, const int32_t* lens, const int32_t* offsets, const osp_cons_config* cfg,
                  uint8_t* cons1, uint8_t* cons2, int cap, uint8_t* is_cons1, int32_t* score1, int32_t* score2, osp_cons_result* res);

#ifdef __cplusplus
}
#endif
#endif
