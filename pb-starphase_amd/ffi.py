"""ctypes binding of include/starphase_hip.h (one-to-one; no logic lives here)."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    """the in-tree library; SP_LIB_PATH names another build of it (kernel-variant experiments)"""
    return os.environ.get("SP_LIB_PATH") or os.path.join(_HERE, "libstarphase_hip.so")


class StarphaseError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"starphase_hip error {code}: {msg}")
        self.code = code


SP_OK = 0
SP_ERR_NO_DEVICE = 2
SP_NO_DIAG = -(2 ** 31)
SP_MAX_ED = 511


class sp_pair(C.Structure):
    _fields_ = [("a", C.c_uint32), ("b", C.c_uint32), ("diag", C.c_int32), ("max_ed", C.c_int32)]


class sp_aln(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("ok", "nm", "a_start", "a_end", "b_start", "b_end", "a_len", "b_len")]


class sp_hla_db_desc(C.Structure):
    _fields_ = [
        ("n_alleles", C.c_uint32), ("n_genes", C.c_uint32),
        ("gene_of", C.c_void_p),
        ("dna", C.c_char_p), ("dna_off", C.c_void_p),
        ("cdna", C.c_char_p), ("cdna_off", C.c_void_p),
        ("gene_ref", C.c_char_p), ("gene_ref_off", C.c_void_p),
        ("gene_fwd", C.c_void_p),
        ("exon_off", C.c_void_p), ("exon_start", C.c_void_p), ("exon_end", C.c_void_p),
        ("ref_buffer", C.c_int32),
    ]


class sp_hla_realign(C.Structure):
    _fields_ = [("status", C.c_int32), ("best_allele", C.c_int32), ("gene", C.c_int32),
                ("nm", C.c_int32), ("target_len", C.c_int32), ("unmapped", C.c_int32),
                ("aln", sp_aln),
                ("seg_start", C.c_int32), ("seg_end", C.c_int32),
                ("dna_offset", C.c_int32), ("hpc_offset", C.c_int32),
                ("mm2_score", C.c_int32), ("mm2_nm", C.c_int32), ("mm2_t_start", C.c_int32), ("mm2_t_end", C.c_int32), ("mm2_q_start", C.c_int32), ("mm2_q_end", C.c_int32),
                ("k1_chains", C.c_int32), ("k1_mappings", C.c_int32), ("k1_chain_score", C.c_int32), ("reserved_", C.c_int32)]


SP_ABI_VERSION = 2          # include/starphase_hip.h
SP_K1_SEL = 16
K1_HIT_FIELDS = ("allele", "rev", "chain_score", "n_seeds", "t_len", "sel_rank", "diag", "ok", "cell_nm", "a_start", "a_end", "b_start", "b_end",
                 "dp_max", "nm", "t_start", "t_end", "q_start", "q_end", "primary")
SP_MAX_CHAIN = 64


class sp_chain_problem(C.Structure):
    _fields_ = [("n_haps", C.c_uint32), ("hap_type", C.c_void_p), ("hap_subtype", C.POINTER(C.c_char_p)),
                ("n_translate", C.c_uint32), ("translate_key", C.POINTER(C.c_char_p)), ("translate_val", C.POINTER(C.c_char_p)),
                ("n_connections", C.c_uint32), ("connection_a", C.POINTER(C.c_char_p)), ("connection_b", C.POINTER(C.c_char_p)),
                ("n_singletons", C.c_uint32), ("singletons", C.POINTER(C.c_char_p)),
                ("n_reads", C.c_uint32), ("read_chain_off", C.c_void_p), ("chain_off", C.c_void_p), ("chain_items", C.c_void_p),
                ("read_w_off", C.c_void_p), ("w_ed", C.c_void_p), ("w_ov", C.c_void_p),
                ("infer_connections", C.c_int32), ("normalize_all_alleles", C.c_int32), ("ignore_chain_label_limits", C.c_int32),
                ("lasso_penalty", C.c_double), ("ln_ed_penalty", C.c_double), ("unexpected_chain_penalty", C.c_double),
                ("inferred_edge_penalty", C.c_double)]


class sp_chain_result(C.Structure):
    _fields_ = [("n_possible", C.c_int32), ("index1", C.c_int32), ("index2", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32),
                ("chain1", C.c_int32 * SP_MAX_CHAIN), ("chain2", C.c_int32 * SP_MAX_CHAIN),
                ("score", C.c_double), ("ln_ed_penalty", C.c_double), ("mn_llh_penalty", C.c_double),
                ("allele_expected_penalty", C.c_double), ("unexpected_chain_penalty", C.c_double), ("inferred_chain_penalty", C.c_double),
                ("edit_distance", C.c_uint64), ("n_pairs_scored", C.c_uint64)]


REGION_HIT_DTYPE = np.dtype([(n, np.int32) for n in ("read", "template_idx", "start", "end", "seq_len", "nm", "unmapped", "clip_start", "clip_end",
                                                     "mm2_score", "mm2_nm", "mm2_start", "mm2_end", "mm2_q_start", "mm2_q_end")])


SP_VAR_MAXDIP = 4096


class sp_variant_problem(C.Structure):
    _fields_ = [("n_haps", C.c_int32), ("hap_is_sv", C.c_void_p), ("hap_is_core", C.c_void_p), ("slot_off", C.c_void_p),
                ("alt_off", C.c_void_p), ("alt_var", C.c_void_p), ("n_vars", C.c_int32), ("var_is_core", C.c_void_p),
                ("n_obs", C.c_int32), ("obs_var", C.c_void_p), ("obs_gt", C.c_void_p), ("obs_ps", C.c_void_p), ("obs_sv_label", C.c_void_p)]


class sp_variant_result(C.Structure):
    _fields_ = [("score", C.c_int64 * 4), ("n_dip", C.c_int32), ("overflow", C.c_int32), ("dip", (C.c_int32 * 2) * SP_VAR_MAXDIP),
                ("dip_comb", C.c_int32 * SP_VAR_MAXDIP)]


class sp_sv_definitions(C.Structure):
    _fields_ = [("n_genes", C.c_int32), ("gene_start", C.c_void_p), ("gene_end", C.c_void_p), ("gene_forward", C.c_void_p),
                ("exon_off", C.c_void_p), ("exon_start", C.c_void_p), ("exon_end", C.c_void_p),
                ("n_full", C.c_int32), ("full_generic", C.c_void_p), ("full_off", C.c_void_p), ("full_gene", C.c_void_p),
                ("n_partial", C.c_int32), ("partial_generic", C.c_void_p), ("partial_off", C.c_void_p), ("partial_gene", C.c_void_p),
                ("partial_first", C.c_void_p), ("partial_end", C.c_void_p)]


class sp_cons_config(C.Structure):
    _fields_ = [("min_count", C.c_int32), ("dual_max_ed_delta", C.c_int32), ("allow_early_termination", C.c_int32), ("allow_dual", C.c_int32),
                ("offset_window", C.c_int32), ("offset_compare_length", C.c_int32), ("min_af", C.c_double),
                ("max_queue_size", C.c_int32), ("max_capacity_per_size", C.c_int32), ("max_nodes_wo_constraint", C.c_int32), ("no_retry_ladder", C.c_int32)]


class sp_cons_result(C.Structure):
    _fields_ = [("is_dual", C.c_int32), ("len1", C.c_int32), ("len2", C.c_int32), ("split_at", C.c_int32), ("gave_up", C.c_int64), ("best_total", C.c_int64),
                ("split_w2", C.c_int64), ("split_total", C.c_int64), ("nodes_expanded", C.c_int64)]


class sp_cons_problem(C.Structure):
    _fields_ = [("reads", C.c_void_p), ("read_idx", C.c_void_p), ("n", C.c_uint32), ("offsets", C.c_void_p), ("cfg", sp_cons_config)]


class sp_cons_output(C.Structure):
    _fields_ = [("cons1", C.c_void_p), ("cons2", C.c_void_p), ("cap", C.c_uint32), ("is_cons1", C.c_void_p), ("score1", C.c_void_p),
                ("score2", C.c_void_p), ("result", sp_cons_result), ("status", C.c_int32)]


class sp_priority_problem(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n", C.c_uint32), ("levels", C.POINTER(C.c_void_p)), ("offsets", C.POINTER(C.c_void_p)), ("seeds", C.c_void_p),
                ("cfg", sp_cons_config)]


class sp_priority_job(C.Structure):
    _fields_ = [("problem", C.POINTER(sp_priority_problem)), ("max_groups", C.c_uint32), ("cap", C.c_uint32), ("n_groups", C.POINTER(C.c_uint32)),
                ("group_of", C.c_void_p), ("cons", C.c_void_p), ("status", C.c_int32), ("gave_up", C.c_int32)]


class sp_cyp_problem(C.Structure):
    _fields_ = [("templates", C.c_void_p), ("template_type", C.c_void_p), ("template_subtype", C.POINTER(C.c_char_p)), ("template_deep", C.c_void_p),
                ("backbone", C.c_char_p), ("backbone_len", C.c_uint32),
                ("n_variants", C.c_uint32), ("var_pos", C.c_void_p), ("var_ref", C.POINTER(C.c_char_p)), ("var_alt", C.POINTER(C.c_char_p)), ("var_is_vi", C.c_void_p),
                ("n_alleles", C.c_uint32), ("allele_subtype", C.POINTER(C.c_char_p)), ("hap_matrix", C.c_void_p),
                ("n_translate", C.c_uint32), ("translate_key", C.POINTER(C.c_char_p)), ("translate_val", C.POINTER(C.c_char_p)),
                ("n_connections", C.c_uint32), ("connection_a", C.POINTER(C.c_char_p)), ("connection_b", C.POINTER(C.c_char_p)),
                ("n_singletons", C.c_uint32), ("singletons", C.POINTER(C.c_char_p)),
                ("min_consensus_count", C.c_int32), ("dual_max_ed_delta", C.c_int32), ("min_consensus_fraction", C.c_double),
                ("infer_connections", C.c_int32), ("normalize_d6_only", C.c_int32), ("var_label", C.POINTER(C.c_char_p))]


class sp_cyp_locus(C.Structure):
    _fields_ = [("chrom_name", C.c_char_p), ("chrom_seq", C.c_char_p), ("window_start", C.c_uint64), ("window_len", C.c_uint64)] + \
               [(n, C.c_uint64) for n in ("d6_start", "d6_end", "d7_start", "d7_end", "rep6_start", "rep6_end", "rep7_start", "rep7_end",
                                          "spacer_start", "spacer_end", "link_start", "link_end", "backbone_start", "backbone_end", "star5_start", "star5_end")] + \
               [(n, C.c_uint64 * 9) for n in ("d6_exon_start", "d6_exon_end", "d7_exon_start", "d7_exon_end")]


class sp_cyp_gene_def(C.Structure):
    _fields_ = [("n_alleles", C.c_uint32), ("star_allele", C.POINTER(C.c_char_p)), ("var_off", C.c_void_p), ("var_pos", C.c_void_p),
                ("var_ref", C.POINTER(C.c_char_p)), ("var_alt", C.POINTER(C.c_char_p)), ("var_id", C.POINTER(C.c_char_p)), ("var_vi", C.POINTER(C.c_char_p))]


class sp_cyp_config(C.Structure):
    _fields_ = [("n_translate", C.c_uint32), ("translate_key", C.POINTER(C.c_char_p)), ("translate_val", C.POINTER(C.c_char_p)),
                ("n_connections", C.c_uint32), ("connection_a", C.POINTER(C.c_char_p)), ("connection_b", C.POINTER(C.c_char_p)),
                ("n_singletons", C.c_uint32), ("singletons", C.POINTER(C.c_char_p))]


class sp_cyp_db_stats(C.Structure):
    _fields_ = [("n_templates", C.c_uint32), ("n_variants", C.c_uint32), ("n_vi", C.c_uint32), ("n_alleles", C.c_uint32), ("backbone_len", C.c_uint32),
                ("first_variant_pos", C.c_int64), ("last_variant_pos", C.c_int64)]


SP_CYP_MAXCONS = 64


class sp_cyp_call(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_consensus", C.c_int32), ("cons_type", C.c_int32 * SP_CYP_MAXCONS), ("cons_subtype", (C.c_char * 48) * SP_CYP_MAXCONS),
                ("n1", C.c_int32), ("n2", C.c_int32), ("chain1", C.c_int32 * 64), ("chain2", C.c_int32 * 64), ("score", C.c_double),
                ("hap1", C.c_char * 256), ("hap2", C.c_char * 256), ("core1", C.c_char * 256), ("core2", C.c_char * 256),
                ("deep1", C.c_char * 2048), ("deep2", C.c_char * 2048), ("searches_gave_up", C.c_int32), ("reserved_", C.c_int32)]


class sp_cyp_region_variants(C.Structure):
    _fields_ = [("has_variants", C.c_uint8 * SP_CYP_MAXCONS), ("state", C.c_void_p)]


class sp_hla_call_config(C.Structure):
    _fields_ = [("min_consensus_count", C.c_int32), ("dual_max_ed_delta", C.c_int32), ("min_consensus_fraction", C.c_double),
                ("expected_maf", C.c_double), ("min_cdf", C.c_double), ("require_dna", C.c_int32), ("disable_cdna", C.c_int32),
                ("absent_capable", C.c_int32), ("normalized_coverage", C.c_double)]


class sp_hla_call(C.Structure):
    _fields_ = [("status", C.c_int32), ("allele1", C.c_int32), ("allele2", C.c_int32), ("typed1", C.c_int32), ("typed2", C.c_int32),
                ("n_reads", C.c_int32), ("counts1", C.c_int32), ("counts2", C.c_int32), ("is_dual", C.c_int32), ("dual_passed", C.c_int32),
                ("is_hemizygous", C.c_int32), ("used_dna_dual", C.c_int32), ("cons1_len", C.c_int32), ("cons2_len", C.c_int32),
                ("maf", C.c_double), ("cdf", C.c_double)]


def hla_call_config(min_consensus_count=3, dual_max_ed_delta=100, min_consensus_fraction=0.10, expected_maf=0.45, min_cdf=0.001,
                    require_dna=False, disable_cdna=False, absent_capable=False, normalized_coverage=-1.0):
    """defaults of the `diplotype` CLI (src/cli/diplotype.rs:110-190)"""
    return sp_hla_call_config(min_consensus_count, dual_max_ed_delta, min_consensus_fraction, expected_maf, min_cdf, int(require_dna),
                              int(disable_cdna), int(absent_capable), normalized_coverage)


class sp_hla_best(C.Structure):
    _fields_ = [("best_allele", C.c_int32), ("n_scored", C.c_int32), ("mm2_stats", C.c_int32 * 6)]


ALN_DTYPE = np.dtype([(n, np.int32) for n in ("ok", "nm", "a_start", "a_end", "b_start", "b_end", "a_len", "b_len")])
REALIGN_DTYPE = np.dtype([("status", np.int32), ("best_allele", np.int32), ("gene", np.int32),
                          ("nm", np.int32), ("target_len", np.int32), ("unmapped", np.int32),
                          ("aln", ALN_DTYPE),
                          ("seg_start", np.int32), ("seg_end", np.int32),
                          ("dna_offset", np.int32), ("hpc_offset", np.int32),
                          ("mm2_score", np.int32), ("mm2_nm", np.int32), ("mm2_t_start", np.int32), ("mm2_t_end", np.int32), ("mm2_q_start", np.int32), ("mm2_q_end", np.int32),
                          ("k1_chains", np.int32), ("k1_mappings", np.int32), ("k1_chain_score", np.int32), ("reserved_", np.int32)])
K1_HIT_DTYPE = np.dtype([(n, np.int32) for n in K1_HIT_FIELDS])

_lib = None


def lib():
    """Load libstarphase_hip.so (in-tree build).  Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP extension is the product; there is no fallback)")
    L = C.CDLL(path)
    vp, u32, i32, u64 = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64
    sigs = {
        "sp_abi_version": (i32, []),
        "sp_struct_size": (i32, [C.c_char_p]),
        "sp_hla_seed_index_info": (i32, [vp, vp, vp]),
        "sp_seqset_sketch": (i32, [vp, vp, u32, vp, vp, vp, u32, C.POINTER(u32)]),
        "sp_hla_realign_seeded_audit": (i32, [vp, vp, vp, u32, vp, u32, C.POINTER(u32), vp, C.POINTER(u32), C.POINTER(i32), vp]),
        "sp_device_count": (i32, [C.POINTER(i32)]),
        "sp_ctx_create": (i32, [i32, vp, C.POINTER(vp)]),
        "sp_ctx_destroy": (None, [vp]),
        "sp_affine_rescore_batch": (i32, [vp, vp, vp, vp, u64, C.POINTER(sp_affine_opts), i32, vp]),
        "sp_ctx_get_info": (i32, [vp, C.POINTER(sp_ctx_info)]),
        "sp_last_error": (C.c_char_p, [vp]),
        "sp_ctx_synchronize": (i32, [vp]),
        "sp_ctx_set_option": (i32, [vp, C.c_char_p, C.c_int64]),
        "sp_seqset_upload": (i32, [vp, C.c_char_p, vp, u32, C.POINTER(vp)]),
        "sp_seqset_upload_format": (i32, [vp, i32, vp, vp, vp, u32, C.POINTER(vp)]),
        "sp_seqset_upload_async": (i32, [vp, i32, vp, vp, vp, u32, C.POINTER(vp)]),
        "sp_seqset_wait": (i32, [vp]),
        "sp_seqset_skipped": (i32, [vp, C.POINTER(u32)]),
        "sp_seqset_free": (None, [vp]),
        "sp_seqset_count": (i32, [vp, C.POINTER(u32)]),
        "sp_seqset_length": (i32, [vp, u32, C.POINTER(u32)]),
        "sp_anchor_batch": (i32, [vp, vp, vp, vp, vp, u64, vp, vp]),
        "sp_align_batch": (i32, [vp, vp, vp, vp, u64, vp, vp, u32]),
        "sp_hla_db_create": (i32, [vp, C.POINTER(sp_hla_db_desc), C.POINTER(vp)]),
        "sp_hla_db_free": (None, [vp]),
        "sp_hla_realign_reads": (i32, [vp, vp, vp, vp, vp]),
        "sp_hla_score_consensus": (i32, [vp, vp, u32, C.c_char_p, u32, C.c_char_p, u32, i32, i32, C.POINTER(sp_hla_best), vp]),
        "sp_hla_score_consensus_batch": (i32, [vp, vp, u32, vp, C.POINTER(C.c_char_p), vp, C.POINTER(C.c_char_p), vp, i32, i32, vp]),
        "sp_hla_type_consensus_batch": (i32, [vp, vp, u32, vp, C.POINTER(C.c_char_p), vp, i32, i32, vp]),
        "sp_hla_type_consensus": (i32, [vp, vp, u32, C.c_char_p, u32, i32, i32, C.POINTER(sp_hla_best), vp, C.c_char_p, u32, C.POINTER(u32)]),
        "sp_cyp_best_chain_pair": (i32, [vp, C.POINTER(sp_chain_problem), C.POINTER(sp_chain_result)]),
        "sp_anchor_batch_topk": (i32, [vp, vp, vp, vp, vp, u64, i32, vp, vp]),
        "sp_cyp_find_regions": (i32, [vp, vp, vp, vp, C.c_double, vp, u64, C.POINTER(u64)]),
        "sp_cyp_weight_segments": (i32, [vp, vp, vp, vp, vp, vp, vp]),
        "sp_cyp_score_alleles": (i32, [vp, u32, u32, vp, vp, u32, vp, vp, vp, vp]),
        "sp_variant_solve": (i32, [vp, C.POINTER(sp_variant_problem), C.POINTER(sp_variant_result)]),
        "sp_variant_solve_batch": (i32, [vp, u32, C.POINTER(C.POINTER(sp_variant_problem)), C.POINTER(sp_variant_result), C.POINTER(i32)]),
        "sp_variant_is_deletion": (i32, [C.POINTER(sp_sv_definitions), u64, u64, C.POINTER(i32), C.POINTER(i32)]),
        "sp_hla_is_passing_dual": (i32, [u64, u64, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "sp_hla_is_hemizygous_better": (i32, [vp, vp, vp, u32, i32, u64, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "sp_hla_normalized_coverage": (i32, [vp, u32, vp, u32, C.POINTER(C.c_double)]),
        "sp_hpc_pos": (u64, [C.c_char_p, u64, u64]),
        "sp_hpc": (u64, [C.c_char_p, u64, C.c_char_p]),
        "sp_cyp_chain_to_hap": (u32, [vp, u32, vp, C.POINTER(C.c_char_p), u32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), i32, C.c_char_p, u32]),
        "sp_cyp_build_chains": (i32, [u32, vp, u32, vp, vp, vp, vp, vp, vp, u32, vp, u32, vp, vp, vp, vp, vp]),
        "sp_variant_normalize": (i32, [C.c_char_p, u64, u64, C.c_char_p, C.c_char_p, C.POINTER(u64), C.c_char_p, C.c_char_p, u32]),
        "sp_variant_multi_normalize": (i32, [C.c_char_p, u64, u64, C.c_char_p, C.c_char_p, u32, C.POINTER(u32), vp, vp, C.c_char_p, C.c_char_p, u32]),
        "sp_consensus": (i32, [vp, vp, vp, u32, vp, C.POINTER(sp_cons_config), C.c_char_p, C.c_char_p, u32, vp, vp, vp, C.POINTER(sp_cons_result)]),
        "sp_consensus_dual": (i32, [vp, vp, vp, u32, vp, C.POINTER(sp_cons_config), C.c_char_p, C.c_char_p, u32, vp, vp, vp, C.POINTER(sp_cons_result)]),
        "sp_consensus_batch": (i32, [vp, u32, C.POINTER(sp_cons_problem), C.POINTER(sp_cons_output)]),
        "sp_consensus_dual_batch": (i32, [vp, u32, C.POINTER(sp_cons_problem), C.POINTER(sp_cons_output)]),
        "sp_cyp_variant_states": (i32, [vp, vp, C.c_char_p, u32, u32, vp, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), vp, vp]),
        "sp_cyp_diplotype": (i32, [vp, C.POINTER(sp_cyp_problem), vp, C.POINTER(sp_cyp_call), C.c_char_p, u32]),
        "sp_cyp_diplotype_cohort": (i32, [vp, C.POINTER(sp_cyp_problem), u32, C.POINTER(vp), C.POINTER(sp_cyp_call), C.c_char_p, u32, C.POINTER(i32)]),
        "sp_cyp_diplotype_detailed": (i32, [vp, C.POINTER(sp_cyp_problem), vp, C.POINTER(sp_cyp_call), C.c_char_p, u32, C.POINTER(sp_cyp_region_variants)]),
        "sp_cyp_alleles_json": (i32, [C.POINTER(sp_cyp_problem), C.POINTER(sp_cyp_call), C.POINTER(sp_cyp_region_variants), C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]),
        "sp_consensus_priority": (i32, [vp, C.POINTER(sp_priority_problem), u32, u32, C.POINTER(u32), vp, C.c_char_p]),
        "sp_consensus_priority_many": (i32, [vp, u32, C.POINTER(sp_priority_job)]),
        "sp_hla_diplotype_gene": (i32, [vp, vp, u32, vp, vp, C.POINTER(sp_hla_call_config), C.POINTER(sp_hla_call), C.c_char_p, C.c_char_p, u32, vp]),
        "sp_hla_diplotype_genes": (i32, [vp, vp, u32, vp, vp, vp, C.POINTER(sp_hla_call_config), C.POINTER(sp_hla_call), C.c_char_p, u32, vp]),
        "sp_hla_diplotype_cohort": (i32, [vp, vp, u32, vp, u32, vp, vp, vp, C.POINTER(sp_hla_call_config), C.POINTER(sp_hla_call), C.c_char_p, u32, vp]),
        "sp_diplotype_string": (u32, [C.c_char_p, C.c_char_p, i32, C.c_char_p, u32]),
        "sp_inexact_haplotype": (u32, [C.c_char_p, u32, C.POINTER(C.c_char_p), vp, vp, C.POINTER(i32), C.c_char_p, u32]),
        "sp_cyp_db_create": (i32, [vp, C.POINTER(sp_cyp_locus), C.POINTER(sp_cyp_gene_def), C.POINTER(sp_cyp_config), C.POINTER(vp)]),
        "sp_cyp_db_free": (None, [vp]),
        "sp_cyp_db_info": (i32, [vp, C.POINTER(sp_cyp_db_stats)]),
        "sp_cyp_db_template": (i32, [vp, u32, C.POINTER(i32), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(u32), C.POINTER(i32)]),
        "sp_cyp_db_variant": (i32, [vp, u32, C.POINTER(C.c_int64), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(i32)]),
        "sp_cyp_db_index_label": (i32, [vp, C.c_char_p, C.POINTER(u32)]),
        "sp_cyp_db_index_variant": (i32, [vp, u64, C.c_char_p, C.c_char_p, C.POINTER(u32)]),
        "sp_cyp_db_allele": (i32, [vp, u32, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_uint8))]),
        "sp_cyp_db_problem": (i32, [vp, C.POINTER(sp_cyp_problem)]),
        "sp_group_unique_id": (i32, [vp]),
        "sp_group_create": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "sp_group_free": (None, [vp]),
        "sp_group_size": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "sp_gather_results": (i32, [vp, vp, u64, vp]),
        "sp_profile_reset": (i32, [vp]),
        "sp_microbench": (i32, [vp, C.c_char_p, C.POINTER(C.c_double)]),
        "sp_profile_get": (i32, [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)           # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    # a library built from another header would fill records of another size into the arrays allocated here: refuse it
    if L.sp_abi_version() != SP_ABI_VERSION:
        raise ImportError(f"{path}: ABI version {L.sp_abi_version()}, this binding was written for {SP_ABI_VERSION} (rebuild: __graft_entry__.build())")
    for name, size in (("sp_hla_realign", REALIGN_DTYPE.itemsize), ("sp_aln", ALN_DTYPE.itemsize), ("sp_k1_seed_hit", K1_HIT_DTYPE.itemsize),
                       ("sp_hla_best", C.sizeof(sp_hla_best)), ("sp_hla_realign", C.sizeof(sp_hla_realign))):
        if L.sp_struct_size(name.encode()) != size:
            raise ImportError(f"{path}: {name} is {L.sp_struct_size(name.encode())} bytes in the library, {size} in this binding")
    _lib = L
    return L


def _cstr(buf, off, cap):
    """the NUL-terminated text at buf[off : off + cap] (buf.raw would copy the whole buffer for every string: 67 MB per consensus of a 256-sample cohort call)"""
    return bytes(memoryview(buf)[off:off + cap]).split(b"\0", 1)[0].decode()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _concat(seqs):
    offs = np.zeros(len(seqs) + 1, np.uint64)
    if len(seqs):
        offs[1:] = np.cumsum([len(s) for s in seqs])
    blob = "".join(seqs).encode("ascii") if len(seqs) and isinstance(seqs[0], str) else b"".join(seqs)
    return blob, offs


class sp_affine_opts(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("a", "b", "q", "e", "q2", "e2", "sc_ambi")]


PAIR_DTYPE = np.dtype([("a", np.uint32), ("b", np.uint32), ("diag", np.int32), ("max_ed", np.int32)])                 # sp_pair
AFFINE_DTYPE = np.dtype([(k, np.int32) for k in ("score", "nm", "a_start", "a_end", "b_start", "b_end")])


class sp_ctx_info(C.Structure):
    _fields_ = [("device", C.c_int32), ("num_cus", C.c_int32), ("hw_queues", C.c_int32), ("hw_queues_set_by_library", C.c_int32), ("warning", C.c_char * 256)]


class Context:
    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        rc = lib().sp_ctx_create(int(device), stream, C.byref(self._h))
        if rc != SP_OK:
            raise StarphaseError(rc, "sp_ctx_create failed (no gfx950 device?)")

    def check(self, rc):
        if rc != SP_OK:
            raise StarphaseError(rc, lib().sp_last_error(self._h).decode())

    def affine_rescore(self, A, B, pairs, a=1, band=64):
        """sp_affine_rescore_batch: pairs = [(a index, b index, diag = b_pos - a_pos)] -> structured array (score, nm, a_start, a_end, b_start, b_end)"""
        rows = np.zeros(len(pairs), PAIR_DTYPE)
        for i, (x, y, d) in enumerate(pairs):
            rows[i] = (x, y, d, 0)
        out = np.zeros(len(pairs), AFFINE_DTYPE)
        op = sp_affine_opts(a, 4, 6, 2, 26, 1, 1)
        self.check(lib().sp_affine_rescore_batch(self._h, A._h, B._h, _ptr(rows), len(pairs), C.byref(op), int(band), _ptr(out)))
        return out

    def info(self):
        """sp_ctx_get_info -> dict(device, num_cus, hw_queues, hw_queues_set_by_library, warning)"""
        st = sp_ctx_info()
        self.check(lib().sp_ctx_get_info(self._h, C.byref(st)))
        return dict(device=st.device, num_cus=st.num_cus, hw_queues=st.hw_queues, hw_queues_set_by_library=bool(st.hw_queues_set_by_library), warning=st.warning.decode())

    def close(self):
        if self._h:
            lib().sp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self.check(lib().sp_ctx_synchronize(self._h))

    def set_option(self, name, value):
        self.check(lib().sp_ctx_set_option(self._h, name.encode(), int(value)))

    def upload(self, seqs):
        return SeqSet(self, seqs)

    def upload_format(self, fmt, blob, offsets, lengths=None, wait=True):
        """sp_seqset_upload_format / sp_seqset_upload_async: bytes already in an upload format (SP_SEQ_ASCII / SP_SEQ_BAM4 / SP_SEQ_PACKED2)"""
        return SeqSet(self, None, blob=blob, offsets=offsets, lengths=lengths, fmt=fmt, wait=wait)

    def anchor_batch(self, A, B, a_idx, b_idx):
        a_idx = np.ascontiguousarray(a_idx, np.uint32)
        b_idx = np.ascontiguousarray(b_idx, np.uint32)
        n = len(a_idx)
        diag = np.zeros(n, np.int32)
        votes = np.zeros(n, np.int32)
        self.check(lib().sp_anchor_batch(self._h, A._h, B._h, _ptr(a_idx), _ptr(b_idx), n, _ptr(diag), _ptr(votes)))
        return diag, votes

    def align_batch(self, A, B, a_idx, b_idx, diag, max_ed, events=False):
        n = len(a_idx)
        pairs = np.zeros(n, np.dtype([("a", np.uint32), ("b", np.uint32), ("diag", np.int32), ("max_ed", np.int32)]))
        pairs["a"] = a_idx
        pairs["b"] = b_idx
        pairs["diag"] = diag
        pairs["max_ed"] = max_ed
        out = np.zeros(n, ALN_DTYPE)
        ev = None
        stride = 0
        if events:
            stride = int(np.max(pairs["max_ed"])) if n else 1
            stride = max(1, stride)
            ev = np.zeros((n, stride), np.uint32)
        self.check(lib().sp_align_batch(self._h, A._h, B._h, _ptr(pairs), n, _ptr(out), _ptr(ev), stride))
        return (out, ev) if events else out

    def cyp_best_chain_pair(self, hap_type, hap_subtype, translate, connections, singletons, read_chain_off, chain_off, chain_items,
                            read_w_off, w_ed, w_ov, infer, normalize_all, ignore_limits, penalties):
        """sp_cyp_best_chain_pair; returns (status, sp_chain_result).  Expected failures (16/17/18) are returned, not raised."""
        def strs(items):
            arr = (C.c_char_p * max(1, len(items)))()
            for i, s in enumerate(items):
                arr[i] = s.encode() if s is not None else None
            return arr
        p = sp_chain_problem()
        ht = np.ascontiguousarray(hap_type, np.int32)
        st = strs(list(hap_subtype))
        tk, tv = strs([a for a, _ in translate]), strs([b for _, b in translate])
        ca, cb = strs([a for a, _ in connections]), strs([b for _, b in connections])
        sg = strs(list(singletons))
        rco = np.ascontiguousarray(read_chain_off, np.uint32)
        co = np.ascontiguousarray(chain_off, np.uint32)
        ci = np.ascontiguousarray(chain_items, np.uint32)
        rwo = np.ascontiguousarray(read_w_off, np.uint32)
        ed = np.ascontiguousarray(w_ed, np.uint64)
        ov = np.ascontiguousarray(w_ov, np.float64)
        p.n_haps, p.hap_type, p.hap_subtype = len(ht), ht.ctypes.data, st
        p.n_translate, p.translate_key, p.translate_val = len(translate), tk, tv
        p.n_connections, p.connection_a, p.connection_b = len(connections), ca, cb
        p.n_singletons, p.singletons = len(singletons), sg
        p.n_reads = len(rco) - 1
        p.read_chain_off, p.chain_off, p.chain_items = rco.ctypes.data, co.ctypes.data, ci.ctypes.data
        p.read_w_off, p.w_ed, p.w_ov = rwo.ctypes.data, ed.ctypes.data, ov.ctypes.data
        p.infer_connections, p.normalize_all_alleles, p.ignore_chain_label_limits = int(infer), int(normalize_all), int(ignore_limits)
        p.lasso_penalty, p.ln_ed_penalty, p.unexpected_chain_penalty, p.inferred_edge_penalty = penalties
        res = sp_chain_result()
        rc = lib().sp_cyp_best_chain_pair(self._h, C.byref(p), C.byref(res))
        if rc not in (0, 16, 17, 18):
            self.check(rc)
        return rc, res

    def anchor_batch_topk(self, A, B, a_idx, b_idx, topk):
        a_idx = np.ascontiguousarray(a_idx, np.uint32)
        b_idx = np.ascontiguousarray(b_idx, np.uint32)
        n = len(a_idx)
        diag = np.zeros((n, topk), np.int32)
        votes = np.zeros((n, topk), np.int32)
        self.check(lib().sp_anchor_batch_topk(self._h, A._h, B._h, _ptr(a_idx), _ptr(b_idx), n, int(topk), _ptr(diag), _ptr(votes)))
        return diag, votes

    def cyp_find_regions(self, templates, template_type, reads, max_missing_frac, cap=None):
        tt = np.ascontiguousarray(template_type, np.int32)
        cap = cap or reads.n * 16 + 16
        hits = np.zeros(cap, REGION_HIT_DTYPE)
        n = C.c_uint64(0)
        self.check(lib().sp_cyp_find_regions(self._h, templates._h, _ptr(tt), reads._h, float(max_missing_frac), _ptr(hits), cap, C.byref(n)))
        if n.value > cap:
            return self.cyp_find_regions(templates, template_type, reads, max_missing_frac, cap=n.value)
        return hits[:n.value]

    def cyp_weight_segments(self, consensus, allowed, segments):
        al = np.ascontiguousarray(allowed, np.uint8)
        ed = np.zeros((segments.n, consensus.n), np.uint64)
        ov = np.zeros((segments.n, consensus.n), np.float64)
        kept = np.zeros(segments.n, np.uint8)
        self.check(lib().sp_cyp_weight_segments(self._h, consensus._h, _ptr(al), segments._h, _ptr(ed), _ptr(ov), _ptr(kept)))
        return ed, ov, kept

    def cyp_score_alleles(self, hap_matrix, is_vi, states):
        hm = np.ascontiguousarray(hap_matrix, np.uint8)
        vi = np.ascontiguousarray(is_vi, np.uint8)
        st = np.ascontiguousarray(states, np.uint8)
        n_alleles, n_variants = hm.shape
        n_seqs = st.shape[0]
        bv = np.zeros(n_seqs, np.uint32)
        ba = np.zeros(n_seqs, np.uint32)
        tie = np.zeros((n_seqs, n_alleles), np.uint8)
        self.check(lib().sp_cyp_score_alleles(self._h, n_variants, n_alleles, _ptr(hm), _ptr(vi), n_seqs, _ptr(st), _ptr(bv), _ptr(ba), _ptr(tie)))
        return bv, ba, tie

    def variant_solve(self, problem):
        """problem: an sp_variant_problem (caller keeps the arrays alive).  Returns (score tuple, [(h1, h2, combination)])"""
        res = sp_variant_result()
        self.check(lib().sp_variant_solve(self._h, C.byref(problem), C.byref(res)))
        if res.overflow:
            raise StarphaseError(5, "more than SP_VAR_MAXDIP tied diplotypes")
        return tuple(res.score), [(res.dip[i][0], res.dip[i][1], res.dip_comb[i]) for i in range(res.n_dip)]

    def variant_solve_batch(self, problems):
        """sp_variant_solve_batch -> [(score tuple, [(h1, h2, combination)])] per problem"""
        n = len(problems)
        ptrs = (C.POINTER(sp_variant_problem) * max(1, n))(*[C.pointer(p) for p in problems])
        res = (sp_variant_result * max(1, n))()
        rcs = (C.c_int32 * max(1, n))()
        self.check(lib().sp_variant_solve_batch(self._h, n, ptrs, res, rcs))
        return [(tuple(res[i].score), [(res[i].dip[k][0], res[i].dip[k][1], res[i].dip_comb[k]) for k in range(res[i].n_dip)]) for i in range(n)]

    def consensus(self, reads, cfg, offsets=None, read_idx=None, cap=None, two_pass=False):
        """sp_consensus / sp_consensus_dual (two_pass) on a SeqSet -> dict like the oracle harness returns"""
        n = len(read_idx) if read_idx is not None else reads.n
        idx = np.ascontiguousarray(read_idx, np.uint32) if read_idx is not None else None
        offs = None
        if offsets is not None:
            offs = np.array([-1 if o is None else int(o) for o in offsets], np.int32)
        lens = reads.lengths if read_idx is None else reads.lengths[np.asarray(read_idx, np.int64)]
        if cap is None:
            cap = int(lens.max() if n else 0) + int(max(0, offs.max()) if offs is not None and n else 0) + 64
        c1, c2 = C.create_string_buffer(cap + 1), C.create_string_buffer(cap + 1)
        is1, s1, s2 = np.zeros(max(1, n), np.uint8), np.zeros(max(1, n), np.int32), np.zeros(max(1, n), np.int32)
        res = sp_cons_result()
        fn = lib().sp_consensus_dual if two_pass else lib().sp_consensus
        self.check(fn(self._h, reads._h, _ptr(idx), n, _ptr(offs), C.byref(cfg), c1, c2, cap + 1, _ptr(is1), _ptr(s1), _ptr(s2), C.byref(res)))
        return dict(cons=[c1.value.decode(), c2.value.decode() if res.is_dual else None], is_dual=bool(res.is_dual), is_cons1=is1[:n].astype(bool),
                    score1=s1[:n].copy(), score2=s2[:n].copy(), split_at=res.split_at, nodes_expanded=res.nodes_expanded, gave_up=bool(res.gave_up))

    def consensus_batch(self, problems, two_pass=False):
        """problems: list of dict(reads=SeqSet, cfg=sp_cons_config, offsets=None|list, read_idx=None|array, cap=None).
        sp_consensus_batch / sp_consensus_dual_batch -> list of result dicts (as consensus())"""
        k = len(problems)
        P, O, keep = (sp_cons_problem * k)(), (sp_cons_output * k)(), []
        for j, q in enumerate(problems):
            reads, idx = q["reads"], q.get("read_idx")
            n = len(idx) if idx is not None else reads.n
            idx = np.ascontiguousarray(idx, np.uint32) if idx is not None else None
            offs = np.array([-1 if o is None else int(o) for o in q["offsets"]], np.int32) if q.get("offsets") is not None else None
            lens = reads.lengths if idx is None else reads.lengths[idx.astype(np.int64)]
            cap = q.get("cap") or (int(lens.max() if n else 0) + int(max(0, offs.max()) if offs is not None and n else 0) + 64)
            c1, c2 = C.create_string_buffer(cap + 1), C.create_string_buffer(cap + 1)
            is1, s1, s2 = np.zeros(max(1, n), np.uint8), np.zeros(max(1, n), np.int32), np.zeros(max(1, n), np.int32)
            keep.append((idx, offs, c1, c2, is1, s1, s2, n))
            P[j].reads, P[j].read_idx, P[j].n, P[j].offsets, P[j].cfg = reads._h, (idx.ctypes.data if idx is not None else None), n, (offs.ctypes.data if offs is not None else None), q["cfg"]
            O[j].cons1, O[j].cons2, O[j].cap = C.cast(c1, C.c_void_p), C.cast(c2, C.c_void_p), cap + 1
            O[j].is_cons1, O[j].score1, O[j].score2 = is1.ctypes.data, s1.ctypes.data, s2.ctypes.data
        fn = lib().sp_consensus_dual_batch if two_pass else lib().sp_consensus_batch
        self.check(fn(self._h, k, P, O))
        out = []
        for j, (idx, offs, c1, c2, is1, s1, s2, n) in enumerate(keep):
            res = O[j].result
            out.append(dict(cons=[c1.value.decode(), c2.value.decode() if res.is_dual else None], is_dual=bool(res.is_dual), is_cons1=is1[:n].astype(bool),
                            score1=s1[:n].copy(), score2=s2[:n].copy(), split_at=res.split_at, nodes_expanded=res.nodes_expanded, gave_up=bool(res.gave_up)))
        return out

    def consensus_priority(self, levels, cfg, offsets=None, seeds=None, max_groups=64, cap=None):
        """sp_consensus_priority: levels = list of SeqSet (one sequence per read each) -> (group_of, [[consensus per level] per group])"""
        nl, n = len(levels), levels[0].n
        cap = cap or int(max(int(L.lengths.max()) if L.n else 0 for L in levels)) + 1024
        lv = (C.c_void_p * nl)(*[L._h for L in levels])
        keep = []
        of = (C.c_void_p * nl)()
        for l in range(nl):
            if offsets is not None and offsets[l] is not None:
                a = np.array([-1 if o is None else int(o) for o in offsets[l]], np.int32)
                keep.append(a)
                of[l] = a.ctypes.data
            else:
                of[l] = None
        sd = np.array([-1 if x is None else int(x) for x in seeds], np.int32) if seeds is not None else None
        pr = sp_priority_problem(nl, n, lv, of, (sd.ctypes.data if sd is not None else None), cfg)
        ng = C.c_uint32(0)
        group_of = np.zeros(max(1, n), np.int32)
        buf = C.create_string_buffer(max_groups * nl * cap)
        self.check(lib().sp_consensus_priority(self._h, C.byref(pr), max_groups, cap, C.byref(ng), _ptr(group_of), buf))
        text = lambda j: _cstr(buf, j * cap, cap)
        return group_of[:n].copy(), [[text(g * nl + l) for l in range(nl)] for g in range(ng.value)]

    def consensus_priority_many(self, problems, max_groups=64):
        """sp_consensus_priority_many: problems = [(levels, cfg, offsets, seeds)] in lockstep -> [(status, group_of, [[consensus per level] per group])]"""
        keep, jobs = [], (sp_priority_job * max(1, len(problems)))()
        for q, (levels, cfg, offsets, seeds) in enumerate(problems):
            nl, n = len(levels), levels[0].n
            cap = int(max(int(L.lengths.max()) if L.n else 0 for L in levels)) + 1024
            lv = (C.c_void_p * nl)(*[L._h for L in levels])
            of = (C.c_void_p * nl)()
            arrs = []
            for l in range(nl):
                if offsets is not None and offsets[l] is not None:
                    a = np.array([-1 if o is None else int(o) for o in offsets[l]], np.int32); arrs.append(a); of[l] = a.ctypes.data
                else:
                    of[l] = None
            sd = np.array([-1 if x is None else int(x) for x in seeds], np.int32) if seeds is not None else None
            pr = sp_priority_problem(nl, n, lv, of, (sd.ctypes.data if sd is not None else None), cfg)
            ng = C.c_uint32(0); group_of = np.zeros(max(1, n), np.int32); buf = C.create_string_buffer(max_groups * nl * cap)
            keep.append((lv, of, arrs, sd, pr, ng, group_of, buf, nl, n, cap))
            jobs[q] = sp_priority_job(C.pointer(pr), max_groups, cap, C.pointer(ng), group_of.ctypes.data, C.cast(buf, C.c_void_p), 0)
        self.check(lib().sp_consensus_priority_many(self._h, len(problems), jobs))
        out = []
        for q, (lv, of, arrs, sd, pr, ng, group_of, buf, nl, n, cap) in enumerate(keep):
            out.append((jobs[q].status, group_of[:n].copy(), [[_cstr(buf, (g * nl + l) * cap, cap) for l in range(nl)] for g in range(ng.value)]))
        return out

    def cyp_variant_states(self, seqs, backbone, var_pos, var_ref, var_alt):
        """sp_cyp_variant_states on a SeqSet -> (states [n_seqs][n_variants] uint8, placements sp_aln array)"""
        nv = len(var_pos)
        pos = np.ascontiguousarray(var_pos, np.int32)
        refs = (C.c_char_p * max(1, nv))(*[r.encode() for r in var_ref])
        alts = (C.c_char_p * max(1, nv))(*[a.encode() for a in var_alt])
        states = np.full((seqs.n, nv), 3, np.uint8)
        alns = np.zeros(max(1, seqs.n), ALN_DTYPE)
        self.check(lib().sp_cyp_variant_states(self._h, seqs._h, backbone.encode(), len(backbone), nv, _ptr(pos), refs, alts, _ptr(states), _ptr(alns)))
        return states, alns[:seqs.n]

    def cyp_diplotype(self, templates, template_type, template_subtype, template_deep, backbone, variants, is_vi, allele_subtype, hap_matrix,
                      cfg, reads, min_count=3, min_af=0.10, delta=100, infer=False, normalize_d6_only=False, cons_cap=16384, var_labels=None):
        """sp_cyp_diplotype.  templates: SeqSet; variants: [(pos, ref, alt)]; cfg: dict(translate, connections, singletons).
        Returns (sp_cyp_call, [consensus strings], [(type, subtype|None)])"""
        def strs(items):
            arr = (C.c_char_p * max(1, len(items)))()
            for i, x in enumerate(items):
                arr[i] = x.encode() if x is not None else None
            return arr
        tt = np.ascontiguousarray(template_type, np.int32)
        deep = np.ascontiguousarray(template_deep, np.uint8)
        pos = np.ascontiguousarray([v[0] for v in variants], np.int32)
        vi = np.ascontiguousarray(is_vi, np.uint8)
        hm = np.ascontiguousarray(hap_matrix, np.uint8)
        keep = [strs(list(template_subtype)), strs([v[1] for v in variants]), strs([v[2] for v in variants]), strs(list(allele_subtype)),
                strs([a for a, _ in cfg["translate"]]), strs([b for _, b in cfg["translate"]]),
                strs([a for a, _ in cfg["connections"]]), strs([b for _, b in cfg["connections"]]), strs(list(cfg["singletons"]))]
        pr = sp_cyp_problem(templates._h, tt.ctypes.data, keep[0], deep.ctypes.data, backbone.encode(), len(backbone),
                            len(variants), pos.ctypes.data, keep[1], keep[2], vi.ctypes.data, len(allele_subtype), keep[3], hm.ctypes.data,
                            len(cfg["translate"]), keep[4], keep[5], len(cfg["connections"]), keep[6], keep[7], len(cfg["singletons"]), keep[8],
                            min_count, delta, min_af, int(infer), int(normalize_d6_only), strs(list(var_labels)) if var_labels is not None else None)
        call = sp_cyp_call()
        buf = C.create_string_buffer(SP_CYP_MAXCONS * cons_cap)
        self.check(lib().sp_cyp_diplotype(self._h, C.byref(pr), reads._h, C.byref(call), buf, cons_cap))
        cons = [_cstr(buf, i * cons_cap, cons_cap) for i in range(call.n_consensus)]
        labels = [(int(call.cons_type[i]), (call.cons_subtype[i].value.decode() or None)) for i in range(call.n_consensus)]
        return call, cons, labels

    def profile_reset(self):
        self.check(lib().sp_profile_reset(self._h))

    def microbench(self, what):
        r = C.c_double(0)
        self.check(lib().sp_microbench(self._h, what.encode(), C.byref(r)))
        return r.value

    def counter(self, name):
        return self.profile_get("count:" + name)[2]

    def profile_get(self, name):
        ms, launches, cells = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        self.check(lib().sp_profile_get(self._h, name.encode(), C.byref(ms), C.byref(launches), C.byref(cells)))
        return ms.value, launches.value, cells.value


def normalize_variant(chrom_seq, position, ref, alt, cap=4096):
    """sp_variant_normalize -> (position, ref, alt); raises StarphaseError(8) where NormalizedVariant::new bails"""
    pos = C.c_uint64(0)
    r, a = C.create_string_buffer(cap), C.create_string_buffer(cap)
    seq = chrom_seq.encode() if isinstance(chrom_seq, str) else chrom_seq
    rc = lib().sp_variant_normalize(seq, len(seq) if seq is not None else 0, int(position), ref.encode(), alt.encode(), C.byref(pos), r, a, cap)
    if rc != SP_OK:
        raise StarphaseError(rc, "sp_variant_normalize")
    return int(pos.value), r.value.decode(), a.value.decode()


class SvDefinitions:
    """GeneCollection (src/data_types/gene_definition.rs) + one gene entry's PgxStructuralVariants
    (src/database/pgx_structural_variants.rs), flattened for sp_variant_is_deletion.  gene_dict: name -> {"coordinates": {start, end},
    "exons": [{start, end}], "is_forward_strand"}; structural_variants: {"full_gene_deletions": {label: {"is_generic",
    "full_genes_deleted": [gene]}}, "partial_gene_deletions": {label: {"is_generic", "exons_deleted": {gene: {start, end}}}}}.
    Definitions are kept in label order (the reference walks BTreeMaps)."""

    def __init__(self, gene_dict, structural_variants):
        names = sorted(gene_dict)
        gid = {g: i for i, g in enumerate(names)}
        self.gene_start = np.array([gene_dict[g]["coordinates"]["start"] for g in names] or [0], np.int64)
        self.gene_end = np.array([gene_dict[g]["coordinates"]["end"] for g in names] or [0], np.int64)
        self.gene_forward = np.array([1 if gene_dict[g]["is_forward_strand"] else 0 for g in names] or [0], np.uint8)
        exon_off, es, ee = [0], [], []
        for g in names:
            for x in gene_dict[g]["exons"]:
                es.append(x["start"]); ee.append(x["end"])
            exon_off.append(len(es))
        self.exon_off = np.array(exon_off, np.int32)
        self.exon_start, self.exon_end = np.array(es or [0], np.int64), np.array(ee or [0], np.int64)
        full = structural_variants.get("full_gene_deletions", {})
        part = structural_variants.get("partial_gene_deletions", {})
        self.full_labels, self.partial_labels = sorted(full), sorted(part)
        fo, fg = [0], []
        for k in self.full_labels:
            fg += [gid.get(g, -1) for g in sorted(full[k]["full_genes_deleted"])]
            fo.append(len(fg))
        po, pg, pf, pe = [0], [], [], []
        for k in self.partial_labels:
            for g in sorted(part[k]["exons_deleted"]):
                pg.append(gid.get(g, -1)); pf.append(part[k]["exons_deleted"][g]["start"]); pe.append(part[k]["exons_deleted"][g]["end"])
            po.append(len(pg))
        self.full_generic = np.array([1 if full[k]["is_generic"] else 0 for k in self.full_labels] or [0], np.uint8)
        self.partial_generic = np.array([1 if part[k]["is_generic"] else 0 for k in self.partial_labels] or [0], np.uint8)
        self.full_off, self.full_gene = np.array(fo, np.int32), np.array(fg or [0], np.int32)
        self.partial_off, self.partial_gene = np.array(po, np.int32), np.array(pg or [0], np.int32)
        self.partial_first, self.partial_end = np.array(pf or [0], np.int32), np.array(pe or [0], np.int32)
        self.n_genes = len(names)

    def struct(self, cls=None):
        d = (cls or sp_sv_definitions)()
        d.n_genes, d.n_full, d.n_partial = self.n_genes, len(self.full_labels), len(self.partial_labels)
        for f in ("gene_start", "gene_end", "gene_forward", "exon_off", "exon_start", "exon_end", "full_generic", "full_off", "full_gene",
                  "partial_generic", "partial_off", "partial_gene", "partial_first", "partial_end"):
            setattr(d, f, getattr(self, f).ctypes.data)
        return d

    def label(self, kind, index):
        return None if kind == 0 else (self.full_labels if kind == 1 else self.partial_labels)[index]

    def is_deletion(self, start, end):
        """is_deletion (src/diplotyper.rs:1020): the label of the defined deletion covering [start, end), or None"""
        kind, index = C.c_int32(0), C.c_int32(-1)
        d = self.struct()
        rc = lib().sp_variant_is_deletion(C.byref(d), int(start), int(end), C.byref(kind), C.byref(index))
        if rc != SP_OK:
            raise StarphaseError(rc, "sp_variant_is_deletion")
        return self.label(kind.value, index.value)


def multi_normalize_variant(chrom_seq, position, ref, alt, cap=4096, max_out=16):
    """sp_variant_multi_normalize -> list of None | (position, ref, alt)"""
    n = C.c_uint32(0)
    none = np.zeros(max_out, np.uint8)
    pos = np.zeros(max_out, np.uint64)
    r, a = C.create_string_buffer(cap * max_out), C.create_string_buffer(cap * max_out)
    seq = chrom_seq.encode() if isinstance(chrom_seq, str) else chrom_seq
    rc = lib().sp_variant_multi_normalize(seq, len(seq) if seq is not None else 0, int(position), ref.encode(), alt.encode(), max_out, C.byref(n),
                                          _ptr(none), _ptr(pos), r, a, cap)
    if rc != SP_OK:
        raise StarphaseError(rc, "sp_variant_multi_normalize")
    out = []
    for i in range(n.value):
        if none[i]:
            out.append(None)
        else:
            out.append((int(pos[i]), r.raw[i * cap:(i + 1) * cap].split(b"\0", 1)[0].decode(), a.raw[i * cap:(i + 1) * cap].split(b"\0", 1)[0].decode()))
    return out


def build_chains(hap_type, read_seg_off, ed, kept):
    """sp_cyp_build_chains (host routine).  ed: [segments][n_haps] uint64, kept: [segments] uint8.
    Returns dict(read_index, chains (list per recorded read of lists), w_rows (list per read of segment indices),
    unique_counts, false_allele); raises StarphaseError(7) on the reference's 'chain collapse' panic."""
    hap_type = np.ascontiguousarray(hap_type, np.int32)
    read_seg_off = np.ascontiguousarray(read_seg_off, np.uint32)
    ed = np.ascontiguousarray(ed, np.uint64)
    kept = np.ascontiguousarray(kept, np.uint8)
    n_haps, n_reads, n_seg = len(hap_type), len(read_seg_off) - 1, len(kept)
    chain_cap, item_cap = max(16, 4 * n_reads), max(64, 8 * max(1, n_seg))
    info = (C.c_uint32 * 4)()
    while True:
        read_index = np.zeros(max(1, n_reads), np.uint32)
        rco, rwo = np.zeros(n_reads + 1, np.uint32), np.zeros(n_reads + 1, np.uint32)
        co, items = np.zeros(chain_cap + 1, np.uint32), np.zeros(item_cap, np.uint32)
        w_seg = np.zeros(max(1, n_seg), np.uint32)
        uniq, false_allele = np.zeros(n_haps, np.uint64), np.zeros(n_haps, np.uint8)
        rc = lib().sp_cyp_build_chains(n_haps, _ptr(hap_type), n_reads, _ptr(read_seg_off), _ptr(ed), _ptr(kept), _ptr(read_index), _ptr(rco),
                                       _ptr(co), chain_cap, _ptr(items), item_cap, _ptr(rwo), _ptr(w_seg), _ptr(uniq), _ptr(false_allele),
                                       C.cast(info, C.c_void_p))
        if rc == 6 and (info[1] > chain_cap or info[2] > item_cap):
            chain_cap, item_cap = max(chain_cap, info[1]), max(item_cap, info[2])
            continue
        break
    if rc != SP_OK:
        raise StarphaseError(rc, "sp_cyp_build_chains")
    nk = info[0]
    chains = [[[int(x) for x in items[co[c]:co[c + 1]]] for c in range(rco[k], rco[k + 1])] for k in range(nk)]
    rows = [[int(x) for x in w_seg[rwo[k]:rwo[k + 1]]] for k in range(nk)]
    return dict(read_index=[int(x) for x in read_index[:nk]], chains=chains, w_rows=rows, unique_counts=uniq, false_allele=false_allele)


def _strs(items):
    arr = (C.c_char_p * max(1, len(items)))()
    for i, x in enumerate(items):
        arr[i] = x.encode() if x is not None else None
    return arr


class CypDb:
    """sp_cyp_db: the CYP2D6 templates and typing tables built from the database's own JSON objects.
    cyp2d6_config / cyp2d6_gene_def: the objects of the same name in the database JSON;
    chrom_seq: bases of chromosome window [window_start, window_start + len(chrom_seq)).  ctx=None builds host tables only."""

    def __init__(self, ctx, cyp2d6_config, cyp2d6_gene_def, chrom_seq, window_start):
        self.ctx = ctx
        cc, reg = cyp2d6_config["cyp_coordinates"], cyp2d6_config["cyp_regions"]
        L = sp_cyp_locus()
        chrom = cc["CYP2D6"]["chrom"]
        self._keep = [chrom.encode(), chrom_seq.encode()]
        L.chrom_name, L.chrom_seq, L.window_start, L.window_len = self._keep[0], self._keep[1], int(window_start), len(chrom_seq)
        for name, key in (("d6", "CYP2D6"), ("d7", "CYP2D7"), ("rep6", "REP6"), ("rep7", "REP7"), ("spacer", "spacer"), ("link", "link_region"),
                          ("backbone", "CYP2D6_wfa_backbone")):
            setattr(L, name + "_start", int(cc[key]["start"])); setattr(L, name + "_end", int(cc[key]["end"]))
        L.star5_start, L.star5_end = int(cyp2d6_config["cyp2d6_star5_del"]["start"]), int(cyp2d6_config["cyp2d6_star5_del"]["end"])
        for x in range(9):
            L.d6_exon_start[x], L.d6_exon_end[x] = int(reg["CYP2D6"][f"exon{x + 1}"]["start"]), int(reg["CYP2D6"][f"exon{x + 1}"]["end"])
            L.d7_exon_start[x], L.d7_exon_end[x] = int(reg["CYP2D7"][f"exon{x + 1}"]["start"]), int(reg["CYP2D7"][f"exon{x + 1}"]["end"])
        keys = sorted(cyp2d6_gene_def, key=lambda k: k.encode())           # BTreeMap<String, _> order
        star, off, pos, ref, alt, vid, vvi = [], [0], [], [], [], [], []
        for k in keys:
            d = cyp2d6_gene_def[k]
            star.append(d["star_allele"])
            for v in d["variants"]:
                pos.append(int(v["position"])); ref.append(v["reference"]); alt.append(v["alternate"]); vid.append(v.get("id"))
                vvi.append(v.get("extras", {}).get("VI"))
            off.append(len(pos))
        G = sp_cyp_gene_def()
        a_off, a_pos = np.array(off, np.uint32), np.array(pos or [0], np.uint64)
        self._keep += [a_off, a_pos, _strs(star), _strs(ref), _strs(alt), _strs(vid), _strs(vvi)]
        G.n_alleles, G.star_allele, G.var_off, G.var_pos = len(keys), self._keep[4], a_off.ctypes.data, a_pos.ctypes.data
        G.var_ref, G.var_alt, G.var_id, G.var_vi = self._keep[5], self._keep[6], self._keep[7], self._keep[8]
        tr = sorted(cyp2d6_config.get("cyp_translate", {}).items())
        con = sorted(tuple(x) for x in cyp2d6_config.get("inferred_connections", []))
        sg = sorted(cyp2d6_config.get("unexpected_singletons", []))
        self.cfg = dict(translate=tr, connections=con, singletons=sg)
        K = sp_cyp_config()
        self._keep += [_strs([a for a, _ in tr]), _strs([b for _, b in tr]), _strs([a for a, _ in con]), _strs([b for _, b in con]), _strs(sg)]
        K.n_translate, K.translate_key, K.translate_val = len(tr), self._keep[9], self._keep[10]
        K.n_connections, K.connection_a, K.connection_b = len(con), self._keep[11], self._keep[12]
        K.n_singletons, K.singletons = len(sg), self._keep[13]
        self._create(L, G, K)

    def _create(self, L, G, K):
        ctx = self.ctx
        self._h = C.c_void_p()
        rc = lib().sp_cyp_db_create(ctx._h if ctx is not None else None, C.byref(L), C.byref(G), C.byref(K), C.byref(self._h))
        if rc != SP_OK:
            raise StarphaseError(rc, lib().sp_last_error(ctx._h).decode() if ctx is not None else "sp_cyp_db_create")
        st = sp_cyp_db_stats()
        lib().sp_cyp_db_info(self._h, C.byref(st))
        self.stats = st

    @classmethod
    def from_structs(cls, ctx, L, G, K, keep=None):
        """locus / gene definition / configuration filled elsewhere (sp_database_cyp_flatten)"""
        self = cls.__new__(cls)
        self.ctx, self._keep = ctx, keep
        self.cfg = dict(translate=[(K.translate_key[i].decode(), K.translate_val[i].decode()) for i in range(K.n_translate)],
                        connections=[(K.connection_a[i].decode(), K.connection_b[i].decode()) for i in range(K.n_connections)],
                        singletons=[K.singletons[i].decode() for i in range(K.n_singletons)])
        self._create(L, G, K)
        return self

    def close(self):
        if self._h:
            lib().sp_cyp_db_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def templates(self):
        """[(type, subtype|None, full_allele, sequence, deep)] in visiting order"""
        out = []
        for i in range(self.stats.n_templates):
            t, d, ln = C.c_int32(0), C.c_int32(0), C.c_uint32(0)
            sub, full, seq = C.c_char_p(), C.c_char_p(), C.c_char_p()
            lib().sp_cyp_db_template(self._h, i, C.byref(t), C.byref(sub), C.byref(full), C.byref(seq), C.byref(ln), C.byref(d))
            out.append((t.value, sub.value.decode() if sub.value is not None else None, full.value.decode(), seq.value.decode(), bool(d.value)))
        return out

    def variants(self):
        """[(chrom_pos, ref, alt, label, is_vi)] in LoadedVariants::ordered_variants order"""
        out = []
        for i in range(self.stats.n_variants):
            p, vi = C.c_int64(0), C.c_int32(0)
            r, a, lab = C.c_char_p(), C.c_char_p(), C.c_char_p()
            lib().sp_cyp_db_variant(self._h, i, C.byref(p), C.byref(r), C.byref(a), C.byref(lab), C.byref(vi))
            out.append((p.value, r.value.decode(), a.value.decode(), lab.value.decode(), bool(vi.value)))
        return out

    def index_label(self, label):
        idx = C.c_uint32(0)
        rc = lib().sp_cyp_db_index_label(self._h, label.encode(), C.byref(idx))
        if rc != SP_OK:
            raise KeyError(label)
        return idx.value

    def index_variant(self, pos, ref, alt):
        idx = C.c_uint32(0)
        rc = lib().sp_cyp_db_index_variant(self._h, int(pos), ref.encode(), alt.encode(), C.byref(idx))
        if rc != SP_OK:
            raise KeyError((pos, ref, alt))
        return idx.value

    def alleles(self):
        """([star_allele], hap_matrix [n_alleles][n_variants]) in haplotype_lookup order"""
        names, rows = [], []
        for a in range(self.stats.n_alleles):
            sub, row = C.c_char_p(), C.POINTER(C.c_uint8)()
            lib().sp_cyp_db_allele(self._h, a, C.byref(sub), C.byref(row))
            names.append(sub.value.decode())
            rows.append(np.ctypeslib.as_array(row, shape=(self.stats.n_variants,)).copy() if self.stats.n_variants else np.zeros(0, np.uint8))
        return names, np.array(rows, np.uint8).reshape(len(names), self.stats.n_variants)

    def problem(self, **overrides):
        pr = sp_cyp_problem()
        self.ctx.check(lib().sp_cyp_db_problem(self._h, C.byref(pr)))
        for k, v in overrides.items():
            setattr(pr, k, v)
        return pr

    def diplotype(self, reads, cons_cap=16384, **overrides):
        """sp_cyp_diplotype on this database -> (sp_cyp_call, [consensus strings], [(type, subtype|None)])"""
        pr = self.problem(**overrides)
        call = sp_cyp_call()
        buf = C.create_string_buffer(SP_CYP_MAXCONS * cons_cap)
        self.ctx.check(lib().sp_cyp_diplotype(self.ctx._h, C.byref(pr), reads._h, C.byref(call), buf, cons_cap))
        cons = [_cstr(buf, i * cons_cap, cons_cap) for i in range(call.n_consensus)]
        labels = [(int(call.cons_type[i]), (call.cons_subtype[i].value.decode() or None)) for i in range(call.n_consensus)]
        return call, cons, labels

    def diplotype_cohort(self, read_sets, cons_cap=16384, **overrides):
        """sp_cyp_diplotype_cohort -> [(sp_cyp_call, [consensus strings], status)] per sample"""
        pr = self.problem(**overrides)
        n = len(read_sets)
        calls = (sp_cyp_call * max(1, n))()
        handles = (C.c_void_p * max(1, n))(*[r._h.value if isinstance(r._h, C.c_void_p) else r._h for r in read_sets])
        buf = C.create_string_buffer(max(1, n) * SP_CYP_MAXCONS * cons_cap)
        rcs = (C.c_int32 * max(1, n))()
        self.ctx.check(lib().sp_cyp_diplotype_cohort(self.ctx._h, C.byref(pr), n, handles, calls, buf, cons_cap, rcs))
        out = []
        for i in range(n):
            base = i * SP_CYP_MAXCONS * cons_cap
            cons = [_cstr(buf, base + k * cons_cap, cons_cap) for k in range(calls[i].n_consensus)]
            out.append((calls[i], cons, rcs[i]))
        return out

    def diplotype_detailed(self, reads, cons_cap=16384, **overrides):
        """sp_cyp_diplotype_detailed + sp_cyp_alleles_json -> (sp_cyp_call, {region: [relationship code per variant]}, cyp2d6_alleles.json text)"""
        pr = self.problem(**overrides)
        call = sp_cyp_call()
        buf = C.create_string_buffer(SP_CYP_MAXCONS * cons_cap)
        state = np.full((SP_CYP_MAXCONS, max(1, pr.n_variants)), 254, np.uint8)
        rv = sp_cyp_region_variants()
        rv.state = state.ctypes.data
        self.ctx.check(lib().sp_cyp_diplotype_detailed(self.ctx._h, C.byref(pr), reads._h, C.byref(call), buf, cons_cap, C.byref(rv)))
        need = C.c_uint64(0)
        self.ctx.check(lib().sp_cyp_alleles_json(C.byref(pr), C.byref(call), C.byref(rv), None, 0, C.byref(need)))
        text = C.create_string_buffer(need.value)
        self.ctx.check(lib().sp_cyp_alleles_json(C.byref(pr), C.byref(call), C.byref(rv), text, need.value, None))
        regions = {h: state[h, :pr.n_variants].copy() for h in range(call.n_consensus) if rv.has_variants[h]}
        return call, regions, text.value.decode()


SP_SEQ_ASCII, SP_SEQ_BAM4, SP_SEQ_PACKED2 = 0, 1, 2
_BAM_CODE = np.full(256, 15, np.uint8)
for _i, _c in enumerate("=ACMGRSVTWYHKDBN"):
    _BAM_CODE[ord(_c)] = _i
    _BAM_CODE[ord(_c.lower())] = _i


def encode_bam4(seqs):
    """the SEQ fields a BAM file would hold for these reads: (bytes, byte offsets[n + 1], lengths[n])"""
    lens = np.array([len(s) for s in seqs], np.uint32)
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum((lens.astype(np.uint64) + 1) // 2)
    out = np.zeros(int(offs[-1]), np.uint8)
    for i, s in enumerate(seqs):
        c = _BAM_CODE[np.frombuffer(s.encode(), np.uint8)]
        if len(c) & 1:
            c = np.append(c, 0)
        out[int(offs[i]):int(offs[i + 1])] = (c[0::2] << 4) | c[1::2]
    return out, offs, lens


def encode_packed2(seqs):
    """2 bits per base, four per byte, base b in bits 2 (b & 3): (bytes, byte offsets[n + 1], lengths[n]); A C G T only"""
    code = np.zeros(256, np.uint8)
    for i, c in enumerate("ACGT"):
        code[ord(c)] = i
        code[ord(c.lower())] = i
    lens = np.array([len(s) for s in seqs], np.uint32)
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum((lens.astype(np.uint64) + 3) // 4)
    out = np.zeros(int(offs[-1]), np.uint8)
    for i, s in enumerate(seqs):
        c = code[np.frombuffer(s.encode(), np.uint8)]
        pad = (-len(c)) % 4
        if pad:
            c = np.append(c, np.zeros(pad, np.uint8))
        out[int(offs[i]):int(offs[i + 1])] = c[0::4] | (c[1::4] << 2) | (c[2::4] << 4) | (c[3::4] << 6)
    return out, offs, lens


SP_GROUP_ID_BYTES = 128


def group_unique_id():
    """sp_group_unique_id: the 128-byte id rank 0 makes and hands to the other ranks"""
    buf = np.zeros(SP_GROUP_ID_BYTES, np.uint8)
    rc = lib().sp_group_unique_id(_ptr(buf))
    if rc != SP_OK:
        raise StarphaseError(rc, "sp_group_unique_id (librccl missing?)")
    return buf


class Group:
    """sp_group: the ranks of a node, one context each; gather() is the path's one exchange step (ncclAllGather over RCCL / xGMI)"""

    def __init__(self, ctx, unique_id, rank, n_ranks):
        self.ctx, self.rank, self.n_ranks = ctx, rank, n_ranks
        uid = np.ascontiguousarray(unique_id, np.uint8)
        self._h = C.c_void_p()
        ctx.check(lib().sp_group_create(ctx._h, _ptr(uid), int(rank), int(n_ranks), C.byref(self._h)))

    def gather(self, records):
        """records: a numpy array (any dtype, the same shape on every rank) -> array of shape (n_ranks,) + records.shape"""
        rec = np.ascontiguousarray(records)
        out = np.zeros((self.n_ranks,) + rec.shape, rec.dtype)
        self.ctx.check(lib().sp_gather_results(self._h, _ptr(rec), rec.nbytes, _ptr(out)))
        return out

    def close(self):
        if self._h:
            lib().sp_group_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SeqSet:
    def __init__(self, ctx, seqs=None, blob=None, offsets=None, lengths=None, fmt=SP_SEQ_ASCII, wait=True):
        """seqs: Python strings (sent as ASCII); or blob / offsets / lengths already in one of the upload formats (numpy arrays or bytes).
        wait=False: sp_seqset_upload_async -- call .wait() before the set is used; the buffers are kept alive by this object until then"""
        self.ctx = ctx
        if seqs is not None:
            blob, offsets = _concat(list(seqs))
            self.lengths = np.array([len(s) for s in seqs], np.int64)
            lengths = None
        else:
            self.lengths = np.asarray(lengths if lengths is not None else np.diff(np.asarray(offsets)), np.int64)
        self.n = len(self.lengths)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        ln = None if lengths is None else np.ascontiguousarray(lengths, np.uint32)
        data = blob if isinstance(blob, (bytes, bytearray)) else np.ascontiguousarray(blob)
        dptr = C.cast(C.c_char_p(data), C.c_void_p) if isinstance(data, (bytes, bytearray)) else _ptr(data)
        self._keep = (data, offsets, ln)
        self._h = C.c_void_p()
        fn = lib().sp_seqset_upload_format if wait else lib().sp_seqset_upload_async
        ctx.check(fn(ctx._h, int(fmt), dptr, _ptr(offsets), _ptr(ln), self.n, C.byref(self._h)))
        if wait:
            self._keep = None

    def wait(self):
        self.ctx.check(lib().sp_seqset_wait(self._h))
        self._keep = None
        return self

    def sketch(self, idx):
        """sp_seqset_sketch: the (19,19)-minimizers of sequence idx in position order -> (hash u64[], end position i32[], strand u8[])"""
        cap = 1 << 16
        h, p, st = np.zeros(cap, np.uint64), np.zeros(cap, np.int32), np.zeros(cap, np.uint8)
        n = C.c_uint32(0)
        self.ctx.check(lib().sp_seqset_sketch(self.ctx._h, self._h, int(idx), _ptr(h), _ptr(p), _ptr(st), cap, C.byref(n)))
        return h[:n.value].copy(), p[:n.value].copy(), st[:n.value].copy()

    @property
    def skipped(self):
        k = C.c_uint32(0)
        self.ctx.check(lib().sp_seqset_skipped(self._h, C.byref(k)))
        return k.value

    def close(self):
        if self._h:
            lib().sp_seqset_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HlaDb:
    """sp_hla_db: alleles in database-key order + gene references (see include/starphase_hip.h)."""

    def __init__(self, ctx, gene_of, dna, cdna, gene_ref, gene_fwd, exons, ref_buffer=100):
        self.ctx = ctx
        self.n_alleles = len(dna)
        self.n_genes = len(gene_ref)
        gene_of = np.ascontiguousarray(gene_of, np.uint32)
        dblob, doff = _concat(list(dna))
        cblob, coff = _concat(list(cdna))
        rblob, roff = _concat(list(gene_ref))
        gfwd = np.ascontiguousarray(gene_fwd, np.uint8)
        eoff = np.zeros(self.n_genes + 1, np.uint32)
        es, ee = [], []
        for g, ex in enumerate(exons):
            eoff[g + 1] = eoff[g] + len(ex)
            es += [e[0] for e in ex]
            ee += [e[1] for e in ex]
        es = np.ascontiguousarray(es, np.int32)
        ee = np.ascontiguousarray(ee, np.int32)
        d = sp_hla_db_desc()
        d.n_alleles, d.n_genes = self.n_alleles, self.n_genes
        d.gene_of = _ptr(gene_of)
        d.dna, d.dna_off = dblob, _ptr(doff)
        d.cdna, d.cdna_off = cblob, _ptr(coff)
        d.gene_ref, d.gene_ref_off = rblob, _ptr(roff)
        d.gene_fwd = _ptr(gfwd)
        d.exon_off, d.exon_start, d.exon_end = _ptr(eoff), _ptr(es), _ptr(ee)
        d.ref_buffer = ref_buffer
        self._keep = (gene_of, dblob, doff, cblob, coff, rblob, roff, gfwd, eoff, es, ee)
        self._h = C.c_void_p()
        ctx.check(lib().sp_hla_db_create(ctx._h, C.byref(d), C.byref(self._h)))

    @classmethod
    def from_desc(cls, ctx, desc):
        """an sp_hla_db_desc filled elsewhere (sp_database_hla_flatten)"""
        self = cls.__new__(cls)
        self.ctx, self.n_alleles, self.n_genes, self._keep = ctx, desc.n_alleles, desc.n_genes, None
        self._h = C.c_void_p()
        ctx.check(lib().sp_hla_db_create(ctx._h, C.byref(desc), C.byref(self._h)))
        return self

    def close(self):
        if self._h:
            lib().sp_hla_db_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def realign_reads(self, reads, cells=False):
        out = np.zeros(reads.n, REALIGN_DTYPE)
        cell = np.zeros((reads.n, self.n_alleles), np.uint32) if cells else None
        self.ctx.check(lib().sp_hla_realign_reads(self.ctx._h, self._h, reads._h, _ptr(out), _ptr(cell)))
        return (out, cell) if cells else out

    def seed_index_info(self):
        """sp_hla_seed_index_info -> dict(minimizers, distinct, mid_occ, sequences) of the minimizer index of the DNA alleles (built on first use)"""
        out = np.zeros(4, np.int64)
        self.ctx.check(lib().sp_hla_seed_index_info(self.ctx._h, self._h, _ptr(out)))
        return dict(minimizers=int(out[0]), distinct=int(out[1]), mid_occ=int(out[2]), sequences=int(out[3]))

    def realign_seeded_audit(self, reads, read, chain_cap=65536):
        """sp_hla_realign_seeded_audit for one read of the set -> dict(chains int32[n][10] in rank order, hits (output order), pick, counters)"""
        chains = np.zeros((chain_cap, 10), np.int32)
        hits = np.zeros(SP_K1_SEL, K1_HIT_DTYPE)
        nc, nh, pick = C.c_uint32(0), C.c_uint32(0), C.c_int32(-1)
        ctr = np.zeros(4, np.uint64)
        self.ctx.check(lib().sp_hla_realign_seeded_audit(self.ctx._h, self._h, reads._h, int(read), _ptr(chains), chain_cap, C.byref(nc), _ptr(hits), C.byref(nh), C.byref(pick), _ptr(ctr)))
        return dict(chains=chains[:min(nc.value, chain_cap)].copy(), n_chains=nc.value, hits=hits[:nh.value].copy(), pick=pick.value,
                    counters=dict(seeds=int(ctr[0]), anchors=int(ctr[1]), max_anchors=int(ctr[2]), capacity_hits=int(ctr[3])))

    def diplotype_gene(self, gene, reads, realign, cfg=None, cap=65536):
        """sp_hla_diplotype_gene -> (sp_hla_call, consensus1, consensus2, is_consensus1 of the gene's realigned reads)"""
        cfg = cfg or hla_call_config()
        call = sp_hla_call()
        c1, c2 = C.create_string_buffer(cap), C.create_string_buffer(cap)
        is1 = np.zeros(max(1, reads.n), np.uint8)
        realign = np.ascontiguousarray(realign)
        self.ctx.check(lib().sp_hla_diplotype_gene(self.ctx._h, self._h, int(gene), reads._h, _ptr(realign), C.byref(cfg), C.byref(call), c1, c2, cap, _ptr(is1)))
        return call, c1.value.decode(), c2.value.decode(), is1[(realign["status"] == 0) & (realign["gene"] == gene)].astype(bool)

    def diplotype_genes(self, genes, reads, realign, cfgs=None, cap=65536):
        """sp_hla_diplotype_genes -> list of (sp_hla_call, consensus1, consensus2) per gene, and is_consensus1 per read"""
        k = len(genes)
        g = np.ascontiguousarray(genes, np.uint32)
        cf = (sp_hla_call_config * k)(*[(cfgs[i] if cfgs else hla_call_config()) for i in range(k)])
        calls = (sp_hla_call * k)()
        buf = C.create_string_buffer(2 * k * cap)
        is1 = np.zeros(max(1, reads.n), np.uint8)
        realign = np.ascontiguousarray(realign)
        self.ctx.check(lib().sp_hla_diplotype_genes(self.ctx._h, self._h, k, _ptr(g), reads._h, _ptr(realign), cf, calls, buf, cap, _ptr(is1)))
        text = lambda j: _cstr(buf, j * cap, cap)
        return [(calls[i], text(2 * i), text(2 * i + 1)) for i in range(k)], is1[:reads.n].astype(bool)

    def diplotype_cohort(self, n_samples, read_sample, genes, reads, realign, cfgs=None, cap=16384):
        """sp_hla_diplotype_cohort -> calls[sample][gene] = (sp_hla_call, consensus1, consensus2), and is_consensus1 per read"""
        k = len(genes)
        g = np.ascontiguousarray(genes, np.uint32)
        rs = np.ascontiguousarray(read_sample, np.uint32)
        cf = (sp_hla_call_config * k)(*[(cfgs[i] if cfgs else hla_call_config()) for i in range(k)])
        calls = (sp_hla_call * (k * n_samples))()
        buf = C.create_string_buffer(2 * k * n_samples * cap)
        is1 = np.zeros(max(1, reads.n), np.uint8)
        realign = np.ascontiguousarray(realign)
        self.ctx.check(lib().sp_hla_diplotype_cohort(self.ctx._h, self._h, n_samples, _ptr(rs), k, _ptr(g), reads._h, _ptr(realign), cf, calls, buf, cap, _ptr(is1)))
        text = lambda j: _cstr(buf, j * cap, cap)
        return [[(calls[s * k + i], text(2 * (s * k + i)), text(2 * (s * k + i) + 1)) for i in range(k)] for s in range(n_samples)], is1[:reads.n].astype(bool)

    def type_consensus(self, gene, consensus_fwd, require_dna=False, disable_cdna=False, stats=True):
        """score_consensus of the reference: hg38-forward consensus in, best allele + spliced gene-strand cDNA out"""
        best = sp_hla_best()
        st = np.full((self.n_alleles, 6), -2, np.int32) if stats else None
        buf = C.create_string_buffer(len(consensus_fwd) + 16)
        n = C.c_uint32(0)
        self.ctx.check(lib().sp_hla_type_consensus(self.ctx._h, self._h, int(gene), consensus_fwd.encode(), len(consensus_fwd),
                                                   int(require_dna), int(disable_cdna), C.byref(best), _ptr(st), buf, len(buf), C.byref(n)))
        self.last_mm2_stats = [int(x) for x in best.mm2_stats]                 # sp_hla_best.mm2_stats of this call
        return best.best_allele, best.n_scored, st, buf.raw[:n.value].decode()

    def score_consensus_batch(self, items, require_dna=False, disable_cdna=False):
        """items: [(gene, cons_dna, cons_cdna)] -> list of (best allele, n_scored)"""
        n = len(items)
        g = np.array([it[0] for it in items], np.uint32)
        d = (C.c_char_p * max(1, n))(*[it[1].encode() for it in items]); dl = np.array([len(it[1]) for it in items], np.uint32)
        c = (C.c_char_p * max(1, n))(*[it[2].encode() for it in items]); cl = np.array([len(it[2]) for it in items], np.uint32)
        best = (sp_hla_best * max(1, n))()
        self.ctx.check(lib().sp_hla_score_consensus_batch(self.ctx._h, self._h, n, _ptr(g), d, _ptr(dl), c, _ptr(cl), int(require_dna), int(disable_cdna), best))
        return [(best[i].best_allele, best[i].n_scored) for i in range(n)]

    def type_consensus_batch(self, items, require_dna=False, disable_cdna=False):
        """items: [(gene, consensus_fwd)] -> list of (best allele, n_scored)"""
        n = len(items)
        g = np.array([it[0] for it in items], np.uint32)
        d = (C.c_char_p * max(1, n))(*[it[1].encode() for it in items]); dl = np.array([len(it[1]) for it in items], np.uint32)
        best = (sp_hla_best * max(1, n))()
        self.ctx.check(lib().sp_hla_type_consensus_batch(self.ctx._h, self._h, n, _ptr(g), d, _ptr(dl), int(require_dna), int(disable_cdna), best))
        return [(best[i].best_allele, best[i].n_scored) for i in range(n)]

    def score_consensus(self, gene, cons_dna, cons_cdna, require_dna=False, disable_cdna=False, stats=True):
        best = sp_hla_best()
        st = np.full((self.n_alleles, 6), -2, np.int32) if stats else None
        self.ctx.check(lib().sp_hla_score_consensus(
            self.ctx._h, self._h, int(gene), cons_dna.encode(), len(cons_dna), cons_cdna.encode(), len(cons_cdna),
            int(require_dna), int(disable_cdna), C.byref(best), _ptr(st)))
        self.last_mm2_stats = [int(x) for x in best.mm2_stats]
        return best.best_allele, best.n_scored, st
