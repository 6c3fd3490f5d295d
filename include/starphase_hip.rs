// Generated from include/starphase_hip.h by profiles/scripts/rust_externs.py -- do not edit (tests/test_abi.py checks it is up to date).
// Handles are opaque; the structs a host fills or reads are `#[repr(C)]` mirrors of the header's, field for field.
#![allow(non_camel_case_types, non_snake_case, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const SP_ABI_VERSION: i64 = 2;
pub const SP_SEQ_ASCII: i64 = 0;
pub const SP_SEQ_BAM4: i64 = 1;
pub const SP_SEQ_PACKED2: i64 = 2;
pub const SP_BAND: i64 = 64;
pub const SP_MAX_ED: i64 = 511;
pub const SP_KMER: i64 = 16;
pub const SP_NO_DIAG: i64 = -2147483648;
pub const SP_EV_X: i64 = 0;
pub const SP_EV_D: i64 = 1;
pub const SP_EV_I: i64 = 2;
pub const SP_K1_SEL: i64 = 16;
pub const SP_MAX_CHAIN: i64 = 64;
pub const SP_CYP_MAXCONS: i64 = 64;
pub const SP_VAR_MAXDIP: i64 = 4096;
pub const SP_GROUP_ID_BYTES: i64 = 128;

#[repr(C)]
pub struct sp_ctx_info {
    pub device: i32,
    pub num_cus: i32,
    pub hw_queues: i32,
    pub hw_queues_set_by_library: i32,
    pub warning: [c_char; 256],
}
#[repr(C)]
pub struct sp_pair {
    pub a: u32,
    pub b: u32,
    pub diag: i32,
    pub max_ed: i32,
}
#[repr(C)]
pub struct sp_aln {
    pub ok: i32,
    pub nm: i32,
    pub a_start: i32,
    pub a_end: i32,
    pub b_start: i32,
    pub b_end: i32,
    pub a_len: i32,
    pub b_len: i32,
}
#[repr(C)]
pub struct sp_affine_opts {
    pub a: i32,
    pub b: i32,
    pub q: i32,
    pub e: i32,
    pub q2: i32,
    pub e2: i32,
    pub sc_ambi: i32,
}
#[repr(C)]
pub struct sp_affine_aln {
    pub score: i32,
    pub nm: i32,
    pub a_start: i32,
    pub a_end: i32,
    pub b_start: i32,
    pub b_end: i32,
}
#[repr(C)]
pub struct sp_hla_db_desc {
    pub n_alleles: u32,
    pub n_genes: u32,
    pub gene_of: *const u32,
    pub dna: *const c_char,
    pub dna_off: *const u64,
    pub cdna: *const c_char,
    pub cdna_off: *const u64,
    pub gene_ref: *const c_char,
    pub gene_ref_off: *const u64,
    pub gene_fwd: *const u8,
    pub exon_off: *const u32,
    pub exon_start: *const i32,
    pub exon_end: *const i32,
    pub ref_buffer: i32,
}
#[repr(C)]
pub struct sp_hla_realign {
    pub status: i32,
    pub best_allele: i32,
    pub gene: i32,
    pub nm: i32,
    pub target_len: i32,
    pub unmapped: i32,
    pub aln: sp_aln,
    pub seg_start: i32,
    pub seg_end: i32,
    pub dna_offset: i32,
    pub hpc_offset: i32,
    pub mm2_score: i32,
    pub mm2_nm: i32,
    pub mm2_t_start: i32,
    pub mm2_t_end: i32,
    pub mm2_q_start: i32,
    pub mm2_q_end: i32,
    pub k1_chains: i32,
    pub k1_mappings: i32,
    pub k1_chain_score: i32,
    pub reserved_: i32,
}
#[repr(C)]
pub struct sp_k1_seed_info {
    pub n_chains: i32,
    pub n_selected: i32,
    pub n_mappings: i32,
    pub pick: i32,
    pub chain_score: i32,
    pub rev: i32,
}
#[repr(C)]
pub struct sp_k1_seed_hit {
    pub allele: i32,
    pub rev: i32,
    pub chain_score: i32,
    pub n_seeds: i32,
    pub t_len: i32,
    pub sel_rank: i32,
    pub diag: i32,
    pub ok: i32,
    pub cell_nm: i32,
    pub a_start: i32,
    pub a_end: i32,
    pub b_start: i32,
    pub b_end: i32,
    pub dp_max: i32,
    pub nm: i32,
    pub t_start: i32,
    pub t_end: i32,
    pub q_start: i32,
    pub q_end: i32,
    pub primary: i32,
}
#[repr(C)]
pub struct sp_hla_best {
    pub best_allele: i32,
    pub n_scored: i32,
    pub mm2_stats: [i32; 6],
}
#[repr(C)]
pub struct sp_chain_problem {
    pub n_haps: u32,
    pub hap_type: *const i32,
    pub hap_subtype: *const *const c_char,
    pub n_translate: u32,
    pub translate_key: *const *const c_char,
    pub translate_val: *const *const c_char,
    pub n_connections: u32,
    pub connection_a: *const *const c_char,
    pub connection_b: *const *const c_char,
    pub n_singletons: u32,
    pub singletons: *const *const c_char,
    pub n_reads: u32,
    pub read_chain_off: *const u32,
    pub chain_off: *const u32,
    pub chain_items: *const u32,
    pub read_w_off: *const u32,
    pub w_ed: *const u64,
    pub w_ov: *const f64,
    pub infer_connections: i32,
    pub normalize_all_alleles: i32,
    pub ignore_chain_label_limits: i32,
    pub lasso_penalty: f64,
    pub ln_ed_penalty: f64,
    pub unexpected_chain_penalty: f64,
    pub inferred_edge_penalty: f64,
}
#[repr(C)]
pub struct sp_chain_result {
    pub n_possible: i32,
    pub index1: i32,
    pub index2: i32,
    pub n1: i32,
    pub n2: i32,
    pub chain1: [i32; SP_MAX_CHAIN as usize],
    pub chain2: [i32; SP_MAX_CHAIN as usize],
    pub score: f64,
    pub ln_ed_penalty: f64,
    pub mn_llh_penalty: f64,
    pub allele_expected_penalty: f64,
    pub unexpected_chain_penalty: f64,
    pub inferred_chain_penalty: f64,
    pub edit_distance: u64,
    pub n_pairs_scored: u64,
}
#[repr(C)]
pub struct sp_region_hit {
    pub read: i32,
    pub template_idx: i32,
    pub start: i32,
    pub end: i32,
    pub seq_len: i32,
    pub nm: i32,
    pub unmapped: i32,
    pub clip_start: i32,
    pub clip_end: i32,
    pub mm2_score: i32,
    pub mm2_nm: i32,
    pub mm2_start: i32,
    pub mm2_end: i32,
    pub mm2_q_start: i32,
    pub mm2_q_end: i32,
}
#[repr(C)]
pub struct sp_cyp_problem {
    pub templates: *const sp_seqset,
    pub template_type: *const i32,
    pub template_subtype: *const *const c_char,
    pub template_deep: *const u8,
    pub backbone: *const c_char,
    pub backbone_len: u32,
    pub n_variants: u32,
    pub var_pos: *const i32,
    pub var_ref: *const *const c_char,
    pub var_alt: *const *const c_char,
    pub var_is_vi: *const u8,
    pub n_alleles: u32,
    pub allele_subtype: *const *const c_char,
    pub hap_matrix: *const u8,
    pub n_translate: u32,
    pub translate_key: *const *const c_char,
    pub translate_val: *const *const c_char,
    pub n_connections: u32,
    pub connection_a: *const *const c_char,
    pub connection_b: *const *const c_char,
    pub n_singletons: u32,
    pub singletons: *const *const c_char,
    pub min_consensus_count: i32,
    pub dual_max_ed_delta: i32,
    pub min_consensus_fraction: f64,
    pub infer_connections: i32,
    pub normalize_d6_only: i32,
    pub var_label: *const *const c_char,
}
#[repr(C)]
pub struct sp_cyp_call {
    pub status: i32,
    pub n_consensus: i32,
    pub cons_type: [i32; SP_CYP_MAXCONS as usize],
    pub cons_subtype: [[c_char; 48]; SP_CYP_MAXCONS as usize],
    pub n1: i32,
    pub n2: i32,
    pub chain1: [i32; SP_MAX_CHAIN as usize],
    pub chain2: [i32; SP_MAX_CHAIN as usize],
    pub score: f64,
    pub hap1: [c_char; 256],
    pub hap2: [c_char; 256],
    pub core1: [c_char; 256],
    pub core2: [c_char; 256],
    pub deep1: [c_char; 2048],
    pub deep2: [c_char; 2048],
    pub searches_gave_up: i32,
    pub reserved_: i32,
}
#[repr(C)]
pub struct sp_cyp_region_variants {
    pub has_variants: [u8; SP_CYP_MAXCONS as usize],
    pub state: *mut u8,
}
#[repr(C)]
pub struct sp_cyp_locus {
    pub chrom_name: *const c_char,
    pub chrom_seq: *const c_char,
    pub window_start: u64,
    pub window_len: u64,
    pub d6_start: u64,
    pub d6_end: u64,
    pub d7_start: u64,
    pub d7_end: u64,
    pub rep6_start: u64,
    pub rep6_end: u64,
    pub rep7_start: u64,
    pub rep7_end: u64,
    pub spacer_start: u64,
    pub spacer_end: u64,
    pub link_start: u64,
    pub link_end: u64,
    pub backbone_start: u64,
    pub backbone_end: u64,
    pub star5_start: u64,
    pub star5_end: u64,
    pub d6_exon_start: [u64; 9],
    pub d6_exon_end: [u64; 9],
    pub d7_exon_start: [u64; 9],
    pub d7_exon_end: [u64; 9],
}
#[repr(C)]
pub struct sp_cyp_gene_def {
    pub n_alleles: u32,
    pub star_allele: *const *const c_char,
    pub var_off: *const u32,
    pub var_pos: *const u64,
    pub var_ref: *const *const c_char,
    pub var_alt: *const *const c_char,
    pub var_id: *const *const c_char,
    pub var_vi: *const *const c_char,
}
#[repr(C)]
pub struct sp_cyp_config {
    pub n_translate: u32,
    pub translate_key: *const *const c_char,
    pub translate_val: *const *const c_char,
    pub n_connections: u32,
    pub connection_a: *const *const c_char,
    pub connection_b: *const *const c_char,
    pub n_singletons: u32,
    pub singletons: *const *const c_char,
}
#[repr(C)]
pub struct sp_cyp_db_stats {
    pub n_templates: u32,
    pub n_variants: u32,
    pub n_vi: u32,
    pub n_alleles: u32,
    pub backbone_len: u32,
    pub first_variant_pos: i64,
    pub last_variant_pos: i64,
}
#[repr(C)]
pub struct sp_variant_problem {
    pub n_haps: i32,
    pub hap_is_sv: *const u8,
    pub hap_is_core: *const u8,
    pub slot_off: *const i32,
    pub alt_off: *const i32,
    pub alt_var: *const i32,
    pub n_vars: i32,
    pub var_is_core: *const u8,
    pub n_obs: i32,
    pub obs_var: *const i32,
    pub obs_gt: *const i32,
    pub obs_ps: *const i64,
    pub obs_sv_label: *const i32,
}
#[repr(C)]
pub struct sp_variant_result {
    pub score: [i64; 4],
    pub n_dip: i32,
    pub overflow: i32,
    pub dip: [[i32; 2]; SP_VAR_MAXDIP as usize],
    pub dip_comb: [i32; SP_VAR_MAXDIP as usize],
}
#[repr(C)]
pub struct sp_sv_definitions {
    pub n_genes: i32,
    pub gene_start: *const i64,
    pub gene_end: *const i64,
    pub gene_forward: *const u8,
    pub exon_off: *const i32,
    pub exon_start: *const i64,
    pub exon_end: *const i64,
    pub n_full: i32,
    pub full_generic: *const u8,
    pub full_off: *const i32,
    pub full_gene: *const i32,
    pub n_partial: i32,
    pub partial_generic: *const u8,
    pub partial_off: *const i32,
    pub partial_gene: *const i32,
    pub partial_first: *const i32,
    pub partial_end: *const i32,
}
#[repr(C)]
pub struct sp_cons_config {
    pub min_count: i32,
    pub dual_max_ed_delta: i32,
    pub allow_early_termination: i32,
    pub allow_dual: i32,
    pub offset_window: i32,
    pub offset_compare_length: i32,
    pub min_af: f64,
    pub max_queue_size: i32,
    pub max_capacity_per_size: i32,
    pub max_nodes_wo_constraint: i32,
    pub no_retry_ladder: i32,
}
#[repr(C)]
pub struct sp_cons_result {
    pub is_dual: i32,
    pub len1: i32,
    pub len2: i32,
    pub split_at: i32,
    pub gave_up: i64,
    pub best_total: i64,
    pub split_w2: i64,
    pub split_total: i64,
    pub nodes_expanded: i64,
}
#[repr(C)]
pub struct sp_cons_problem {
    pub reads: *const sp_seqset,
    pub read_idx: *const u32,
    pub n: u32,
    pub offsets: *const i32,
    pub cfg: sp_cons_config,
}
#[repr(C)]
pub struct sp_cons_output {
    pub cons1: *mut c_char,
    pub cons2: *mut c_char,
    pub cap: u32,
    pub is_cons1: *mut u8,
    pub score1: *mut i32,
    pub score2: *mut i32,
    pub result: sp_cons_result,
    pub status: i32,
}
#[repr(C)]
pub struct sp_priority_problem {
    pub n_levels: u32,
    pub n: u32,
    pub levels: *const *const sp_seqset,
    pub offsets: *const *const i32,
    pub seeds: *const i32,
    pub cfg: sp_cons_config,
}
#[repr(C)]
pub struct sp_priority_job {
    pub problem: *const sp_priority_problem,
    pub max_groups: u32,
    pub cap: u32,
    pub n_groups: *mut u32,
    pub group_of: *mut i32,
    pub cons: *mut c_char,
    pub status: i32,
    pub gave_up: i32,
}
#[repr(C)]
pub struct sp_hla_call_config {
    pub min_consensus_count: i32,
    pub dual_max_ed_delta: i32,
    pub min_consensus_fraction: f64,
    pub expected_maf: f64,
    pub min_cdf: f64,
    pub require_dna: i32,
    pub disable_cdna: i32,
    pub absent_capable: i32,
    pub normalized_coverage: f64,
}
#[repr(C)]
pub struct sp_hla_call {
    pub status: i32,
    pub allele1: i32,
    pub allele2: i32,
    pub typed1: i32,
    pub typed2: i32,
    pub n_reads: i32,
    pub counts1: i32,
    pub counts2: i32,
    pub is_dual: i32,
    pub dual_passed: i32,
    pub is_hemizygous: i32,
    pub used_dna_dual: i32,
    pub cons1_len: i32,
    pub cons2_len: i32,
    pub maf: f64,
    pub cdf: f64,
}
#[repr(C)]
pub struct sp_chain_build_info {
    pub n_reads: u32,
    pub n_chains: u32,
    pub n_items: u32,
    pub n_rows: u32,
}
#[repr(C)]
pub struct sp_database_metadata {
    pub pbstarphase_version: *const c_char,
    pub cpic_version: *const c_char,
    pub hla_version: *const c_char,
    pub pharmvar_version: *const c_char,
    pub build_time: *const c_char,
}
#[repr(C)]
pub struct sp_database_stats {
    pub n_gene_entries: u32,
    pub n_hla_sequences: u32,
    pub n_hla_genes: u32,
    pub n_cyp2d6_alleles: u32,
    pub n_collection_genes: u32,
    pub has_hla_config: i32,
    pub has_cyp2d6_config: i32,
    pub reserved: i32,
}
#[repr(C)]
pub struct sp_gene_region {
    pub name: *const c_char,
    pub chrom: *const c_char,
    pub start: u64,
    pub end: u64,
    pub is_forward_strand: i32,
    pub is_absent_capable: i32,
    pub n_exons: u32,
    pub reserved: u32,
    pub exon_start: *const u64,
    pub exon_end: *const u64,
}
#[repr(C)]
pub struct sp_variant_gene_stats {
    pub n_haplotypes: u32,
    pub n_variants: u32,
    pub n_skipped_haplotypes: u32,
    pub n_full_deletions: u32,
    pub n_partial_deletions: u32,
    pub reserved: u32,
}
#[repr(C)]
pub struct sp_vcf_allele {
    pub position: u64,
    pub ref_: *const c_char,
    pub alt: *const c_char,
    pub gt: i32,
    pub reserved: i32,
    pub ps: i64,
}
#[repr(C)]
pub struct sp_vcf_deletion {
    pub start: u64,
    pub end: u64,
    pub gt: i32,
    pub reserved: i32,
    pub ps: i64,
}
#[repr(C)]
pub struct sp_variant_detail {
    pub variant_id: u64,
    pub variant_name: *const c_char,
    pub dbsnp: *const c_char,
    pub chrom: *const c_char,
    pub position: u64,
    pub reference: *const c_char,
    pub alternate: *const c_char,
    pub sv_label: *const c_char,
    pub sv_start: u64,
    pub sv_end: u64,
    pub genotype: i32,
    pub is_core_variant: i32,
    pub phase_set: i64,
}
#[repr(C)]
pub struct sp_mapping_stats {
    pub present: i32,
    pub has_clips: i32,
    pub seq_len: u64,
    pub nm: u64,
    pub unmapped: u64,
    pub clipped_start: u64,
    pub clipped_end: u64,
}
#[repr(C)]
pub struct sp_detailed_mapping {
    pub present: i32,
    pub reserved: i32,
    pub query_len: u64,
    pub target_len: u64,
    pub match_len: u64,
    pub nm: u64,
    pub query_unmapped: u64,
    pub target_unmapped: u64,
    pub cigar: *const c_char,
    pub md: *const c_char,
}
#[repr(C)]
pub struct sp_bam_read {
    pub qname: *const c_char,
    pub flag: u32,
    pub mapq: u32,
    pub ref_id: i32,
    pub reserved: i32,
    pub pos: i64,
    pub end: i64,
    pub l_seq: u32,
    pub n_cigar: u32,
    pub cigar: *const u32,
}

#[repr(C)] pub struct sp_bam { _private: [u8; 0] }
#[repr(C)] pub struct sp_ctx { _private: [u8; 0] }
#[repr(C)] pub struct sp_cyp_db { _private: [u8; 0] }
#[repr(C)] pub struct sp_database { _private: [u8; 0] }
#[repr(C)] pub struct sp_fasta { _private: [u8; 0] }
#[repr(C)] pub struct sp_gene_details { _private: [u8; 0] }
#[repr(C)] pub struct sp_group { _private: [u8; 0] }
#[repr(C)] pub struct sp_hla_db { _private: [u8; 0] }
#[repr(C)] pub struct sp_hla_debug { _private: [u8; 0] }
#[repr(C)] pub struct sp_result { _private: [u8; 0] }
#[repr(C)] pub struct sp_seqset { _private: [u8; 0] }
#[repr(C)] pub struct sp_variant_gene { _private: [u8; 0] }
#[repr(C)] pub struct sp_vcf { _private: [u8; 0] }

#[link(name = "starphase_hip")]
extern "C" {
    pub fn sp_abi_version() -> i32;
    pub fn sp_struct_size(name: *const c_char) -> i32;
    pub fn sp_device_count(count: *mut i32) -> i32;
    pub fn sp_ctx_create(device: i32, stream: *mut c_void, out: *mut *mut sp_ctx) -> i32;
    pub fn sp_ctx_get_info(ctx: *const sp_ctx, out: *mut sp_ctx_info) -> i32;
    pub fn sp_ctx_destroy(ctx: *mut sp_ctx);
    pub fn sp_last_error(ctx: *const sp_ctx) -> *const c_char;
    pub fn sp_ctx_synchronize(ctx: *mut sp_ctx) -> i32;
    pub fn sp_ctx_set_option(ctx: *mut sp_ctx, name: *const c_char, value: i64) -> i32;
    pub fn sp_seqset_upload(ctx: *mut sp_ctx, bases: *const c_char, offsets: *const u64, n: u32, out: *mut *mut sp_seqset) -> i32;
    pub fn sp_seqset_upload_format(ctx: *mut sp_ctx, format: i32, data: *const c_void, offsets: *const u64, lengths: *const u32, n: u32, out: *mut *mut sp_seqset) -> i32;
    pub fn sp_seqset_upload_async(ctx: *mut sp_ctx, format: i32, data: *const c_void, offsets: *const u64, lengths: *const u32, n: u32, out: *mut *mut sp_seqset) -> i32;
    pub fn sp_seqset_wait(set: *mut sp_seqset) -> i32;
    pub fn sp_seqset_skipped(set: *const sp_seqset, n_skipped: *mut u32) -> i32;
    pub fn sp_seqset_free(set: *mut sp_seqset);
    pub fn sp_seqset_count(set: *const sp_seqset, n: *mut u32) -> i32;
    pub fn sp_seqset_length(set: *const sp_seqset, idx: u32, len: *mut u32) -> i32;
    pub fn sp_anchor_batch(ctx: *mut sp_ctx, A: *const sp_seqset, B: *const sp_seqset, a_idx: *const u32, b_idx: *const u32, n_pairs: u64, diag_out: *mut i32, votes_out: *mut i32) -> i32;
    pub fn sp_anchor_batch_topk(ctx: *mut sp_ctx, A: *const sp_seqset, B: *const sp_seqset, a_idx: *const u32, b_idx: *const u32, n_pairs: u64, topk: i32, diag_out: *mut i32, votes_out: *mut i32) -> i32;
    pub fn sp_align_batch(ctx: *mut sp_ctx, A: *const sp_seqset, B: *const sp_seqset, pairs: *const sp_pair, n_pairs: u64, out: *mut sp_aln, events: *mut u32, events_stride: u32) -> i32;
    pub fn sp_affine_rescore_batch(ctx: *mut sp_ctx, A: *const sp_seqset, B: *const sp_seqset, pairs: *const sp_pair, n_pairs: u64, opts: *const sp_affine_opts, band: i32, out: *mut sp_affine_aln) -> i32;
    pub fn sp_hla_db_create(ctx: *mut sp_ctx, desc: *const sp_hla_db_desc, out: *mut *mut sp_hla_db) -> i32;
    pub fn sp_hla_db_free(db: *mut sp_hla_db);
    pub fn sp_hla_seed_index_info(ctx: *mut sp_ctx, db: *const sp_hla_db, out: *mut i64) -> i32;
    pub fn sp_seqset_sketch(ctx: *mut sp_ctx, set: *const sp_seqset, idx: u32, hash: *mut u64, end_pos: *mut i32, strand: *mut u8, cap: u32, n_out: *mut u32) -> i32;
    pub fn sp_hla_realign_seeded_audit(ctx: *mut sp_ctx, db: *const sp_hla_db, reads: *const sp_seqset, read: u32, chains: *mut i32, chain_cap: u32, n_chains: *mut u32, hits: *mut sp_k1_seed_hit, n_hits: *mut u32, pick: *mut i32, counters: *mut u64) -> i32;
    pub fn sp_hla_realign_reads(ctx: *mut sp_ctx, db: *const sp_hla_db, reads: *const sp_seqset, out: *mut sp_hla_realign, cell_out: *mut u32) -> i32;
    pub fn sp_hla_score_consensus(ctx: *mut sp_ctx, db: *const sp_hla_db, gene: u32, cons_dna: *const c_char, cons_dna_len: u32, cons_cdna: *const c_char, cons_cdna_len: u32, require_dna: i32, disable_cdna: i32, best: *mut sp_hla_best, stats: *mut i32) -> i32;
    pub fn sp_hla_score_consensus_batch(ctx: *mut sp_ctx, db: *const sp_hla_db, n: u32, genes: *const u32, cons_dna: *const *const c_char, cons_dna_len: *const u32, cons_cdna: *const *const c_char, cons_cdna_len: *const u32, require_dna: i32, disable_cdna: i32, best: *mut sp_hla_best) -> i32;
    pub fn sp_hla_type_consensus(ctx: *mut sp_ctx, db: *const sp_hla_db, gene: u32, consensus_fwd: *const c_char, consensus_len: u32, require_dna: i32, disable_cdna: i32, best: *mut sp_hla_best, stats: *mut i32, cdna_out: *mut c_char, cdna_cap: u32, cdna_len: *mut u32) -> i32;
    pub fn sp_hla_type_consensus_batch(ctx: *mut sp_ctx, db: *const sp_hla_db, n: u32, genes: *const u32, consensus_fwd: *const *const c_char, consensus_len: *const u32, require_dna: i32, disable_cdna: i32, best: *mut sp_hla_best) -> i32;
    pub fn sp_cyp_best_chain_pair(ctx: *mut sp_ctx, problem: *const sp_chain_problem, result: *mut sp_chain_result) -> i32;
    pub fn sp_cyp_find_regions(ctx: *mut sp_ctx, templates: *const sp_seqset, template_type: *const i32, reads: *const sp_seqset, max_missing_frac: f64, hits: *mut sp_region_hit, hits_cap: u64, n_hits: *mut u64) -> i32;
    pub fn sp_cyp_weight_segments(ctx: *mut sp_ctx, consensus: *const sp_seqset, allowed: *const u8, segments: *const sp_seqset, ed: *mut u64, ov: *mut f64, kept: *mut u8) -> i32;
    pub fn sp_cyp_score_alleles(ctx: *mut sp_ctx, n_variants: u32, n_alleles: u32, hap_matrix: *const u8, is_vi: *const u8, n_seqs: u32, states: *const u8, best_vi: *mut u32, best_all: *mut u32, tie_mask: *mut u8) -> i32;
    pub fn sp_cyp_variant_states(ctx: *mut sp_ctx, seqs: *const sp_seqset, backbone: *const c_char, backbone_len: u32, n_variants: u32, var_pos: *const i32, var_ref: *const *const c_char, var_alt: *const *const c_char, states: *mut u8, alns: *mut sp_aln) -> i32;
    pub fn sp_cyp_diplotype(ctx: *mut sp_ctx, problem: *const sp_cyp_problem, reads: *const sp_seqset, call: *mut sp_cyp_call, consensus: *mut c_char, cons_cap: u32) -> i32;
    pub fn sp_cyp_diplotype_cohort(ctx: *mut sp_ctx, problem: *const sp_cyp_problem, n_samples: u32, reads: *const *const sp_seqset, calls: *mut sp_cyp_call, consensus: *mut c_char, cons_cap: u32, sample_rc: *mut i32) -> i32;
    pub fn sp_cyp_diplotype_detailed(ctx: *mut sp_ctx, problem: *const sp_cyp_problem, reads: *const sp_seqset, call: *mut sp_cyp_call, consensus: *mut c_char, cons_cap: u32, region_variants: *mut sp_cyp_region_variants) -> i32;
    pub fn sp_cyp_alleles_json(problem: *const sp_cyp_problem, call: *const sp_cyp_call, region_variants: *const sp_cyp_region_variants, out: *mut c_char, cap: u64, needed: *mut u64) -> i32;
    pub fn sp_cyp_db_create(ctx: *mut sp_ctx, locus: *const sp_cyp_locus, gene_def: *const sp_cyp_gene_def, config: *const sp_cyp_config, out: *mut *mut sp_cyp_db) -> i32;
    pub fn sp_cyp_db_free(db: *mut sp_cyp_db);
    pub fn sp_cyp_db_info(db: *const sp_cyp_db, stats: *mut sp_cyp_db_stats) -> i32;
    pub fn sp_cyp_db_template(db: *const sp_cyp_db, i: u32, type_: *mut i32, subtype: *mut *const c_char, full_allele: *mut *const c_char, seq: *mut *const c_char, len: *mut u32, deep: *mut i32) -> i32;
    pub fn sp_cyp_db_variant(db: *const sp_cyp_db, i: u32, chrom_pos: *mut i64, ref_: *mut *const c_char, alt: *mut *const c_char, label: *mut *const c_char, is_vi: *mut i32) -> i32;
    pub fn sp_cyp_db_index_label(db: *const sp_cyp_db, label: *const c_char, idx: *mut u32) -> i32;
    pub fn sp_cyp_db_index_variant(db: *const sp_cyp_db, position: u64, ref_: *const c_char, alt: *const c_char, idx: *mut u32) -> i32;
    pub fn sp_cyp_db_allele(db: *const sp_cyp_db, a: u32, subtype: *mut *const c_char, row: *mut *const u8) -> i32;
    pub fn sp_cyp_db_problem(db: *const sp_cyp_db, problem: *mut sp_cyp_problem) -> i32;
    pub fn sp_variant_solve(ctx: *mut sp_ctx, problem: *const sp_variant_problem, result: *mut sp_variant_result) -> i32;
    pub fn sp_variant_solve_batch(ctx: *mut sp_ctx, n: u32, problems: *const *const sp_variant_problem, results: *mut sp_variant_result, problem_rc: *mut i32) -> i32;
    pub fn sp_variant_is_deletion(defs: *const sp_sv_definitions, start: u64, end: u64, kind: *mut i32, index: *mut i32) -> i32;
    pub fn sp_consensus_batch(ctx: *mut sp_ctx, n_problems: u32, problems: *const sp_cons_problem, outputs: *mut sp_cons_output) -> i32;
    pub fn sp_consensus_dual_batch(ctx: *mut sp_ctx, n_problems: u32, problems: *const sp_cons_problem, outputs: *mut sp_cons_output) -> i32;
    pub fn sp_consensus(ctx: *mut sp_ctx, reads: *const sp_seqset, read_idx: *const u32, n: u32, offsets: *const i32, cfg: *const sp_cons_config, cons1: *mut c_char, cons2: *mut c_char, cap: u32, is_cons1: *mut u8, score1: *mut i32, score2: *mut i32, result: *mut sp_cons_result) -> i32;
    pub fn sp_consensus_dual(ctx: *mut sp_ctx, reads: *const sp_seqset, read_idx: *const u32, n: u32, offsets: *const i32, cfg: *const sp_cons_config, cons1: *mut c_char, cons2: *mut c_char, cap: u32, is_cons1: *mut u8, score1: *mut i32, score2: *mut i32, result: *mut sp_cons_result) -> i32;
    pub fn sp_consensus_priority(ctx: *mut sp_ctx, problem: *const sp_priority_problem, max_groups: u32, cap: u32, n_groups: *mut u32, group_of: *mut i32, cons: *mut c_char) -> i32;
    pub fn sp_consensus_priority_many(ctx: *mut sp_ctx, n_jobs: u32, jobs: *mut sp_priority_job) -> i32;
    pub fn sp_hla_diplotype_gene(ctx: *mut sp_ctx, db: *const sp_hla_db, gene: u32, reads: *const sp_seqset, realign: *const sp_hla_realign, cfg: *const sp_hla_call_config, call: *mut sp_hla_call, cons1: *mut c_char, cons2: *mut c_char, cap: u32, is_cons1: *mut u8) -> i32;
    pub fn sp_hla_diplotype_genes(ctx: *mut sp_ctx, db: *const sp_hla_db, n_genes: u32, genes: *const u32, reads: *const sp_seqset, realign: *const sp_hla_realign, cfgs: *const sp_hla_call_config, calls: *mut sp_hla_call, cons: *mut c_char, cap: u32, is_cons1: *mut u8) -> i32;
    pub fn sp_hla_diplotype_cohort(ctx: *mut sp_ctx, db: *const sp_hla_db, n_samples: u32, read_sample: *const u32, n_genes: u32, genes: *const u32, reads: *const sp_seqset, realign: *const sp_hla_realign, cfgs: *const sp_hla_call_config, calls: *mut sp_hla_call, cons: *mut c_char, cap: u32, is_cons1: *mut u8) -> i32;
    pub fn sp_hla_is_passing_dual(counts1: u64, counts2: u64, min_consensus_fraction: f64, expected_maf: f64, min_cdf: f64, maf_out: *mut f64, cdf_out: *mut f64) -> i32;
    pub fn sp_hla_is_hemizygous_better(scores1: *const i64, scores2: *const i64, is_consensus1: *const u8, n_reads: u32, is_dual: i32, dual_max_ed_delta: u64, normalized_coverage: f64, haploid_cost: *mut f64, diploid_cost: *mut f64) -> i32;
    pub fn sp_hla_normalized_coverage(realign: *const sp_hla_realign, n_reads: u32, normalizing_genes: *const u32, n_normalizing: u32, normalized_coverage: *mut f64) -> i32;
    pub fn sp_hpc_pos(seq: *const c_char, len: u64, position: u64) -> u64;
    pub fn sp_hpc(seq: *const c_char, len: u64, out: *mut c_char) -> u64;
    pub fn sp_cyp_chain_to_hap(chain: *const i32, n: u32, hap_type: *const i32, hap_subtype: *const *const c_char, n_translate: u32, translate_key: *const *const c_char, translate_val: *const *const c_char, detail: i32, out: *mut c_char, cap: u32) -> u32;
    pub fn sp_variant_normalize(chrom_seq: *const c_char, chrom_len: u64, position: u64, ref_allele: *const c_char, alt_allele: *const c_char, out_position: *mut u64, out_ref: *mut c_char, out_alt: *mut c_char, cap: u32) -> i32;
    pub fn sp_variant_multi_normalize(chrom_seq: *const c_char, chrom_len: u64, position: u64, ref_allele: *const c_char, alt_allele: *const c_char, max_out: u32, n_out: *mut u32, is_none: *mut u8, out_position: *mut u64, out_ref: *mut c_char, out_alt: *mut c_char, cap: u32) -> i32;
    pub fn sp_cyp_build_chains(n_haps: u32, hap_type: *const i32, n_reads: u32, read_seg_off: *const u32, ed: *const u64, kept: *const u8, read_index: *mut u32, read_chain_off: *mut u32, chain_off: *mut u32, chain_cap: u32, chain_items: *mut u32, item_cap: u32, read_w_off: *mut u32, w_seg: *mut u32, unique_counts: *mut u64, false_allele: *mut u8, info: *mut sp_chain_build_info) -> i32;
    pub fn sp_diplotype_string(hap1: *const c_char, hap2: *const c_char, pharmcat: i32, out: *mut c_char, cap: u32) -> u32;
    pub fn sp_inexact_haplotype(base_haplotype: *const c_char, n_variants: u32, labels: *const *const c_char, is_vi: *const u8, states: *const i32, match_type: *mut i32, out: *mut c_char, cap: u32) -> u32;
    pub fn sp_database_load(path: *const c_char, out: *mut *mut sp_database, err: *mut c_char, err_cap: u32) -> i32;
    pub fn sp_database_parse(text: *const c_char, len: u64, out: *mut *mut sp_database, err: *mut c_char, err_cap: u32) -> i32;
    pub fn sp_database_free(db: *mut sp_database);
    pub fn sp_database_last_error(db: *const sp_database) -> *const c_char;
    pub fn sp_database_get_metadata(db: *const sp_database, out: *mut sp_database_metadata) -> i32;
    pub fn sp_database_info(db: *const sp_database, out: *mut sp_database_stats) -> i32;
    pub fn sp_database_hla_gene(db: *const sp_database, g: u32, out: *mut sp_gene_region) -> i32;
    pub fn sp_database_gene_entry(db: *const sp_database, i: u32, gene_name: *mut *const c_char, chromosome: *mut *const c_char) -> i32;
    pub fn sp_database_hla_flatten(db: *mut sp_database, n_genes: u32, gene_names: *const *const c_char, gene_ref: *const *const c_char, ref_buffer: i32, desc: *mut sp_hla_db_desc) -> i32;
    pub fn sp_database_hla_allele(db: *const sp_database, i: u32, hla_id: *mut *const c_char, gene_name: *mut *const c_char, star_allele: *mut *const c_char) -> i32;
    pub fn sp_database_cyp_window(db: *const sp_database, chrom: *mut *const c_char, start: *mut u64, end: *mut u64) -> i32;
    pub fn sp_database_cyp_flatten(db: *mut sp_database, chrom_seq: *const c_char, window_start: u64, window_len: u64, locus: *mut sp_cyp_locus, gene_def: *mut sp_cyp_gene_def, config: *mut sp_cyp_config) -> i32;
    pub fn sp_variant_gene_create(db: *mut sp_database, gene_name: *const c_char, chrom_seq: *const c_char, chrom_len: u64, out: *mut *mut sp_variant_gene) -> i32;
    pub fn sp_variant_gene_free(gene: *mut sp_variant_gene);
    pub fn sp_variant_gene_info(gene: *const sp_variant_gene, out: *mut sp_variant_gene_stats) -> i32;
    pub fn sp_variant_gene_haplotype(gene: *const sp_variant_gene, h: u32, name: *mut *const c_char, core_allele: *mut *const c_char) -> i32;
    pub fn sp_variant_gene_variant(gene: *const sp_variant_gene, v: u32, position: *mut u64, ref_: *mut *const c_char, alt: *mut *const c_char, name: *mut *const c_char, dbsnp_id: *mut *const c_char, variant_id: *mut i64, is_core: *mut i32) -> i32;
    pub fn sp_variant_gene_sv_definitions(gene: *const sp_variant_gene, out: *mut sp_sv_definitions) -> i32;
    pub fn sp_variant_gene_sv_label(gene: *const sp_variant_gene, kind: i32, index: i32, label: *mut *const c_char) -> i32;
    pub fn sp_variant_gene_problem(gene: *mut sp_variant_gene, n_alleles: u32, alleles: *const sp_vcf_allele, n_deletions: u32, deletions: *const sp_vcf_deletion, max_sv_length: u64, problem: *mut sp_variant_problem) -> i32;
    pub fn sp_variant_gene_problem_variant(gene: *const sp_variant_gene, id: i32, db_variant: *mut i32, sv_label: *mut *const c_char, sv_start: *mut u64, sv_end: *mut u64) -> i32;
    pub fn sp_variant_gene_problem_sv_label(gene: *const sp_variant_gene, label_id: i32, label: *mut *const c_char) -> i32;
    pub fn sp_variant_gene_last_error(gene: *const sp_variant_gene) -> *const c_char;
    pub fn sp_result_create(db: *const sp_database, pbstarphase_version: *const c_char, out: *mut *mut sp_result) -> i32;
    pub fn sp_result_free(result: *mut sp_result);
    pub fn sp_result_last_error(result: *const sp_result) -> *const c_char;
    pub fn sp_gene_details_create(out: *mut *mut sp_gene_details) -> i32;
    pub fn sp_gene_details_free(details: *mut sp_gene_details);
    pub fn sp_gene_details_add_diplotype(d: *mut sp_gene_details, hap1: *const c_char, hap2: *const c_char) -> i32;
    pub fn sp_gene_details_add_simple_diplotype(d: *mut sp_gene_details, hap1: *const c_char, hap2: *const c_char) -> i32;
    pub fn sp_gene_details_set_simple_diplotypes(d: *mut sp_gene_details, some: i32) -> i32;
    pub fn sp_gene_details_add_inexact_diplotype(d: *mut sp_gene_details, base1: *const c_char, n1: u32, labels1: *const *const c_char, is_vi1: *const u8, states1: *const i32, base2: *const c_char, n2: u32, labels2: *const *const c_char, is_vi2: *const u8, states2: *const i32) -> i32;
    pub fn sp_gene_details_add_diplotype_only(d: *mut sp_gene_details, hap1: *const c_char, hap2: *const c_char) -> i32;
    pub fn sp_gene_details_add_variant(d: *mut sp_gene_details, v: *const sp_variant_detail) -> i32;
    pub fn sp_gene_details_add_mapping(d: *mut sp_gene_details, read_qname: *const c_char, best_hla_id: *const c_char, best_star_allele: *const c_char, cdna: *const sp_mapping_stats, dna: *const sp_mapping_stats, is_ignored: i32) -> i32;
    pub fn sp_gene_details_add_multi_mapping(d: *mut sp_gene_details, read_qname: *const c_char, read_start: u64, read_end: u64, consensus_id: u64, consensus_star_allele: *const c_char) -> i32;
    pub fn sp_result_insert(result: *mut sp_result, gene: *const c_char, details: *const sp_gene_details, constructor: i32) -> i32;
    pub fn sp_result_json(result: *mut sp_result, text: *mut *const c_char, len: *mut u64) -> i32;
    pub fn sp_result_save(result: *mut sp_result, path: *const c_char) -> i32;
    pub fn sp_result_pharmcat_tsv(result: *mut sp_result, text: *mut *const c_char, len: *mut u64) -> i32;
    pub fn sp_result_save_pharmcat_tsv(result: *mut sp_result, path: *const c_char) -> i32;
    pub fn sp_aln_strings(aln: *const sp_aln, events: *const u32, target: *const c_char, target_len: u64, cigar: *mut c_char, cigar_cap: u32, md: *mut c_char, md_cap: u32, match_len: *mut u64) -> i32;
    pub fn sp_hla_debug_create(out: *mut *mut sp_hla_debug) -> i32;
    pub fn sp_hla_debug_free(debug: *mut sp_hla_debug);
    pub fn sp_hla_debug_last_error(debug: *const sp_hla_debug) -> *const c_char;
    pub fn sp_hla_debug_add_read(debug: *mut sp_hla_debug, gene: *const c_char, qname: *const c_char, best_match_id: *const c_char, best_match_star: *const c_char) -> i32;
    pub fn sp_hla_debug_add_mapping(debug: *mut sp_hla_debug, gene: *const c_char, qname: *const c_char, hla_id: *const c_char, cdna: *const sp_detailed_mapping, dna: *const sp_detailed_mapping) -> i32;
    pub fn sp_hla_debug_add_dual_stats(debug: *mut sp_hla_debug, gene: *const c_char, call: *const sp_hla_call) -> i32;
    pub fn sp_hla_debug_json(debug: *mut sp_hla_debug, text: *mut *const c_char, len: *mut u64) -> i32;
    pub fn sp_hla_debug_save(debug: *mut sp_hla_debug, path: *const c_char) -> i32;
    pub fn sp_group_unique_id(id: *mut u8) -> i32;
    pub fn sp_group_create(ctx: *mut sp_ctx, id: *const u8, rank: i32, n_ranks: i32, out: *mut *mut sp_group) -> i32;
    pub fn sp_group_free(group: *mut sp_group);
    pub fn sp_group_size(group: *const sp_group, rank: *mut i32, n_ranks: *mut i32) -> i32;
    pub fn sp_gather_results(group: *mut sp_group, records: *const c_void, bytes_per_rank: u64, all_records: *mut c_void) -> i32;
    pub fn sp_bam_open(path: *const c_char, out: *mut *mut sp_bam, err: *mut c_char, err_cap: u32) -> i32;
    pub fn sp_bam_free(bam: *mut sp_bam);
    pub fn sp_bam_last_error(bam: *const sp_bam) -> *const c_char;
    pub fn sp_bam_references(bam: *const sp_bam, n: *mut u32, names: *mut *const *const c_char, lengths: *mut *const u64) -> i32;
    pub fn sp_bam_fetch(bam: *mut sp_bam, chrom: *const c_char, start: u64, end: u64, exclude_flags: u32, dedupe: i32, reads: *mut *const sp_bam_read, n: *mut u32, bases: *mut *const c_char, offsets: *mut *const u64) -> i32;
    pub fn sp_bam_forget(bam: *mut sp_bam) -> i32;
    pub fn sp_bam_last_seq4(bam: *const sp_bam, seq4: *mut *const u8, byte_offsets: *mut *const u64, lengths: *mut *const u32, n: *mut u32) -> i32;
    pub fn sp_vcf_open(path: *const c_char, out: *mut *mut sp_vcf, err: *mut c_char, err_cap: u32) -> i32;
    pub fn sp_vcf_free(vcf: *mut sp_vcf);
    pub fn sp_vcf_index_info(vcf: *const sp_vcf, indexed: *mut i32, lines_parsed_by_fetches: *mut u64) -> i32;
    pub fn sp_vcf_last_error(vcf: *const sp_vcf) -> *const c_char;
    pub fn sp_vcf_samples(vcf: *const sp_vcf, n: *mut u32, names: *mut *const *const c_char) -> i32;
    pub fn sp_vcf_alleles(vcf: *mut sp_vcf, sample: *const c_char, chrom: *const c_char, start: u64, end: u64, out: *mut *const sp_vcf_allele, n: *mut u32) -> i32;
    pub fn sp_vcf_deletions(vcf: *mut sp_vcf, sample: *const c_char, chrom: *const c_char, start: u64, end: u64, out: *mut *const sp_vcf_deletion, n: *mut u32) -> i32;
    pub fn sp_profile_reset(ctx: *mut sp_ctx) -> i32;
    pub fn sp_profile_get(ctx: *mut sp_ctx, kernel: *const c_char, total_ms: *mut f64, launches: *mut u64, cells: *mut u64) -> i32;
    pub fn sp_microbench(ctx: *mut sp_ctx, what: *const c_char, rate: *mut f64) -> i32;
    pub fn sp_fasta_open(path: *const c_char, out: *mut *mut sp_fasta, err: *mut c_char, err_cap: u32) -> i32;
    pub fn sp_fasta_free(fasta: *mut sp_fasta);
    pub fn sp_fasta_last_error(fasta: *const sp_fasta) -> *const c_char;
    pub fn sp_fasta_sequences(fasta: *mut sp_fasta, n: *mut u32, names: *mut *const *const c_char, lengths: *mut *const u64) -> i32;
    pub fn sp_fasta_fetch(fasta: *mut sp_fasta, chrom: *const c_char, start: u64, end: u64, bases: *mut *const c_char, len: *mut u64) -> i32;
}
