/*
 * starphase_hip.h -- C ABI of the MI355X (gfx950) StarPhase hot-path library (libstarphase_hip.so).
 *
 * The reference (PacificBiosciences/pb-StarPhase v2.0.1) exposes no FFI of its own for this path; its hot
 * path sits behind ordinary Rust calls whose arithmetic is delegated to minimap2 through
 * minimap2::Aligner::{with_seq,with_index,map}.  The boundary is therefore cut at those call sites
 * (SURVEY.md 8(b)).  Every entry point below names the reference interface it replaces (paths relative
 * to the reference checkout).  INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Conventions
 *   - plain C types only; caller owns every host buffer (sizes passed explicitly); the library owns device
 *     memory behind opaque handles; no callbacks.
 *   - every call returns an int32 status (SP_OK = 0); nothing aborts or throws across the boundary;
 *     sp_last_error() returns a NUL-terminated message for the last failing call on that context.
 *   - one sp_ctx = one GPU + one HIP stream; a context is used from one host thread at a time
 *     (the reference is single-threaded: src/cli/diplotype.rs:185-191); contexts are independent.
 *   - calls are stream-ordered and synchronous on return.
 *   - results never depend on launch geometry; ties go to the lowest index in the documented order.
 *   - there is NO CPU fallback: without a usable HIP device sp_ctx_create fails with SP_ERR_NO_DEVICE.
 */
#ifndef STARPHASE_HIP_H
#define STARPHASE_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_ABI_VERSION 2      /* 2 (round 5): sp_hla_realign grew (k1_* fields); the records that grew in round 4 (mm2_* fields of sp_hla_realign, sp_hla_best, sp_region_hit;
                               * sp_cyp_call.searches_gave_up; sp_priority_job.gave_up) are covered by the same step.  A caller checks sp_abi_version() == SP_ABI_VERSION and
                               * sp_struct_size() of every record it allocates before its first call */

/* status codes: "expected" outcomes the reference downgrades (CallerError -> NO_MATCH,
 * src/diplotyper.rs:316-327; src/cyp2d6/errors.rs:4-11) are distinct from fatal ones (src/main.rs:181-185). */
enum {
    SP_OK                  = 0,
    SP_ERR_INVALID_ARG     = 1,
    SP_ERR_NO_DEVICE       = 2,
    SP_ERR_HIP             = 3,
    SP_ERR_OUT_OF_MEMORY   = 4,
    SP_ERR_TOO_LONG        = 5,   /* sequence longer than the kernels support (65,535 bases per window) */
    SP_ERR_CAPACITY        = 6,   /* an output array is too small: call again with more room */
    SP_ERR_CHAIN_COLLAPSE  = 7,   /* the reference panics here ("chain collapse", src/cyp2d6/caller.rs:531-533) */
    SP_ERR_BAD_VARIANT     = 8,   /* NormalizedVariant::new bails (src/data_types/normalized_variant.rs:45-52,66-84,160-168) */
    SP_ERR_NO_CHAINING_HEAD = 16, /* CallerError::NoChainingHead  (src/cyp2d6/chaining.rs:321-323) */
    SP_ERR_NO_CHAINS_FOUND  = 17, /* CallerError::NoChainsFound   (src/cyp2d6/chaining.rs:393-396) */
    SP_ERR_NO_SCORE_PAIRS   = 18  /* CallerError::NoScorePairs    (src/cyp2d6/chaining.rs:559-562) */
};

typedef struct sp_ctx    sp_ctx;
typedef struct sp_seqset sp_seqset;
typedef struct sp_hla_db sp_hla_db;

/* ------------------------------------------------------------------ context */
int32_t sp_abi_version(void);
/* sizeof of a record of this header as THIS build of the library lays it out ("sp_hla_realign", "sp_region_hit", ...; every struct typedef'd here); -1 for an unknown name.
 * A binding checks the records it allocates and the library fills (a record that grew since the binding was generated would overflow the caller's array). */
int32_t sp_struct_size(const char* name);
int32_t sp_device_count(int32_t* count);
/* device: HIP ordinal.  stream: a hipStream_t to run on (e.g. torch's current stream) or NULL to create one. */
int32_t sp_ctx_create(int32_t device, void* stream, sp_ctx** out);
/* What a context found when it was made.  hw_queues: the number of hardware queues the HIP runtime maps this process's streams onto (its GPU_MAX_HW_QUEUES
 * setting; the runtime's own default is 4).  The library runs the loci / genes / samples of a call on streams of their own and wants >= 16: with fewer, two chains
 * of small dependent launches can share a queue with another sample's large grids (measured: 230k instead of 300k reads/s with three samples in flight).  The
 * first sp_ctx_create of a process therefore sets GPU_MAX_HW_QUEUES=16 when the variable is not set (hw_queues_set_by_library = 1) -- which only takes effect
 * if the HIP runtime has not been initialised yet (a host that initialises HIP first should export the variable itself); a value below 16 found in the
 * environment is left alone and reported in `warning` (and once through sp_last_error of that context, status SP_OK). */
typedef struct {
    int32_t device, num_cus;
    int32_t hw_queues;                 /* GPU_MAX_HW_QUEUES as seen at sp_ctx_create (after the library's own default, if it set one) */
    int32_t hw_queues_set_by_library;  /* 1: the variable was not set and the library set it to 16 */
    char warning[256];                 /* "" or what the host should change */
} sp_ctx_info;
int32_t sp_ctx_get_info(const sp_ctx* ctx, sp_ctx_info* out);
void    sp_ctx_destroy(sp_ctx* ctx);
const char* sp_last_error(const sp_ctx* ctx);
int32_t sp_ctx_synchronize(sp_ctx* ctx);
/* Tuning switches of a context.  "hla_split_genes" (default 1): sp_hla_diplotype_genes / sp_hla_diplotype_cohort solve the units of a
 * call (the genes of a sample, the (sample, gene) pairs of a cohort) with >= 1,000 realigned reads side by side on up to
 * "hla_split_streams" (1..4, default 3) streams -- helper streams the context owns, one host thread each for the length of the call:
 * lowest latency for one call; set hla_split_genes to 0 when several samples are in flight on contexts of their own, where the streams
 * of the other samples already fill the gaps (sp_cyp_diplotype follows the same switch: with 1 it places the regions of interest on the consensuses
 * for its weights on a helper stream while it types the consensuses).  The calls are the same either way.  "cons_retry_ladder" (default 0: the reference has no such rule): 1 makes sp_cyp_diplotype* run
 * its multi-way consensus with the retry of searches that give up (sp_cons_config.no_retry_ladder = 0, see sp_consensus_priority); sp_cyp_call.gave_up says whether a search of the call gave up.
 * "k1_best_n" (0..8, default 5 = the reference's `best_n`, src/util/mapping.rs:12): sp_hla_realign_reads maps a read the way realign_record does -- minimizer seeding, chaining and
 * selection of the chains that get a base-level alignment as minimap2's `map-hifi` does them, at most k1_best_n secondary chains per read (sp_hla_seed.hip; statement: oracle/mm2.c
 * omm_hla_k1_seeded) -- and accepts among those mappings only; 0 = the exhaustive search of every allele of every anchored gene (the exact argmin the reference approximates).
 * "mm2_rescore" (default 1): sp_hla_realign_reads, sp_hla_type_consensus / sp_hla_score_consensus and sp_cyp_find_regions also report every mapping they return re-scored
 * the reference's way (fields mm2_*; sp_affine_rescore_batch), and the weights of sp_cyp_weight_segments / sp_cyp_diplotype* are taken from the re-scored placement wherever it is
 * within 16 (edits + unmapped bases) of its segment's smallest; 0 leaves the fields zero, the weights on unit-cost counts, and saves the extra launches; 2 = as 1, but every mapping with edits that do not stand alone
 * takes the DP over all of its rows instead of over the rows around those edits (a check of the shortcut, an order of magnitude slower).
 * "k8_persistent" (0 | 1 | 2, default 2; also the environment variable SP_K8_PERSISTENT): consensus batches whose problems have at most 1,024 reads each can run as two
 * persistent kernels (step workgroups and one control workgroup per problem on a second stream, resident for the length of the batch; the two sides hand over through one word
 * each in memory: write-through stores, `sc1` loads and memory-side atomics, no L2 fences -- gfx950 behaviour) instead of a launch pair per step -- the
 * same search, bit for bit.  0 never, 1 whenever a batch qualifies, 2 = the library decides: batches of at most eight problems, when the HIP runtime came up with >= 16 hardware
 * queues (a batch's two kernels wait for each other: GPU_MAX_HW_QUEUES, sp_ctx_get_info), LIGHT batches only (at most 64 resident workgroups: a cohort call's late levels, a
 * small sample) and not while single-sample batches ran side by side within the last second; a heavy batch (a single large sample) runs as launch pairs since round 6 -- a launch
 * carries side orders and branching windows, the resident workgroups do not.  The budget of CUs is counted per
 * process (several processes on one device: set 0).  A batch whose control workgroups have not all started within half a second runs as launch pairs instead, by itself (the
 * context then stays away from the mode for 64 batches and says so in sp_ctx_get_info().warning); a search that exceeds the launch-pair loop's own step bound ends the batch
 * with the same error in both modes; no path returns while one of the two kernels is still running.
 * "k8_side_orders" (0..3, default 1; also SP_K8_SIDE_ORDERS) and "k8_compound" (0 | 1, default 1; also SP_K8_COMPOUND): how many work orders a consensus step launch carries
 * per problem beside the search's own -- the window or the expansion another WAITING node of the best-first search will need when its turn comes, made in the same launch
 * (a second row of workgroups); the children of an expansion made ahead wait unseen until the search takes their parent out at that column and are adopted without a launch --
 * and whether a window may be ordered WITH the children of the branch its lookahead votes foresee at its end (taken when the window stands and the exact votes of that column
 * name no other children).  Neither changes what the search does -- strings, read assignment, per-read scores and the number of nodes expanded are those of the one-order search,
 * bit for bit --, only how many dependent launches it takes (a 2,000-read CYP2D6 sample 423 -> 285 launches, the branching `*4+*68/*1` 2,485 -> 1,417).
 * "k8_side_max_blocks" (default 4096): batches with more step workgroups than this keep to the search's own order.  Persistent batches keep to the search's own order as well.
 * "cyp_cohort_streams" (1..8, default 8): streams sp_cyp_diplotype_cohort spreads its groups of samples over (one host thread each).
 * "k5_block_pairs" (0..1048576, default 4096): sp_cyp_best_chain_pair scores up to this many chain pairs with one workgroup per pair (the few pairs
 * of an ordinary sample: the reads of a pair are shared out over the workgroup), more with one thread per pair; the results are the same.
 * Unknown names: SP_ERR_INVALID_ARG. */
int32_t sp_ctx_set_option(sp_ctx* ctx, const char* name, int64_t value);

/* ------------------------------------------------------------------ sequences
 * Replaces the targets handed to minimap2 via Aligner::with_seq / with_index
 * (src/hla/realigner.rs:56-60,78-79; src/hla/caller.rs:1370-1379; src/cyp2d6/haplotyper.rs:155;
 * src/cyp2d6/chaining.rs:29-30).  bases: concatenated ASCII; offsets[n+1]: start of each sequence.
 * Packed on upload to 2 bits/base + an N plane; anything outside ACGT is 'N' and never matches. */
int32_t sp_seqset_upload(sp_ctx* ctx, const char* bases, const uint64_t* offsets, uint32_t n, sp_seqset** out);
/* The reads of a sample are new for every sample (the read loop of diplotype_hla_batch, src/hla/caller.rs:544-596, and of
 * diplotype_cyp2d6, src/cyp2d6/caller.rs:96-139, hands every record's SEQ to the aligner): their way to the device is part of the
 * path.  An upload stages the caller's bytes through pinned memory in chunks on a copy stream of the context (the copy of chunk k + 1
 * into the staging ring runs under the DMA of chunk k) and packs them on the device.
 *   format SP_SEQ_ASCII    one byte per base (any case; everything but ACGT is 'N'); offsets count bases = bytes
 *          SP_SEQ_BAM4     BAM's SEQ field as stored (SAMv1 4.2: 4 bits per base, high nibble first, "=ACMGRSVTWYHKDBN", each read starting
 *                          on a byte boundary): half the bytes of ASCII over PCIe; offsets[n + 1] count BYTES, lengths[n] bases
 *          SP_SEQ_PACKED2  2 bits per base, four per byte (base b in bits 2 (b & 3) of byte b >> 2, A C G T = 0 1 2 3, no N), each
 *                          sequence starting on a byte boundary: a quarter of the bytes; offsets count BYTES, lengths[n] bases
 * lengths may be NULL for SP_SEQ_ASCII.  A sequence longer than 65,534 bases does not fail the call (one ultra-long read must not cost
 * the sample): it enters the set with length 0, is never aligned, and sp_seqset_skipped counts it.
 * sp_seqset_upload_async returns as soon as the set's tables exist; a worker thread of the library moves the bytes.  The caller's
 * buffers must stay untouched and the set must not be used until sp_seqset_wait has returned SP_OK (it reports what the upload hit).
 * One upload per context is in flight at a time (a second one waits for the first).  While it runs, the context's other calls
 * proceed: the reads of sample i + 1 travel under the kernels of sample i. */
#define SP_SEQ_ASCII   0
#define SP_SEQ_BAM4    1
#define SP_SEQ_PACKED2 2
int32_t sp_seqset_upload_format(sp_ctx* ctx, int32_t format, const void* data, const uint64_t* offsets, const uint32_t* lengths, uint32_t n, sp_seqset** out);
int32_t sp_seqset_upload_async(sp_ctx* ctx, int32_t format, const void* data, const uint64_t* offsets, const uint32_t* lengths, uint32_t n, sp_seqset** out);
int32_t sp_seqset_wait(sp_seqset* set);
int32_t sp_seqset_skipped(const sp_seqset* set, uint32_t* n_skipped);       /* sequences dropped for their length */
void    sp_seqset_free(sp_seqset* set);                                      /* the set's device buffers go back to its context for the next upload (no hipFree, which
                                                                             * would wait for every stream of the device); the context frees them when it is destroyed.
                                                                             * Call it when every call that was given the set has returned */
int32_t sp_seqset_count(const sp_seqset* set, uint32_t* n);
int32_t sp_seqset_length(const sp_seqset* set, uint32_t idx, uint32_t* len);

/* ------------------------------------------------------------------ alignment primitives
 * One cell = one (A, B) pair = one 64-lane wavefront; lane = diagonal.  The alignment contract (banded
 * ends-free edit alignment, deterministic tie rules) is specified in DESIGN.md section 3. */
typedef struct {
    uint32_t a;        /* index into set A (streamed side; minimap2 "query" at most call sites)     */
    uint32_t b;        /* index into set B (window side; minimap2 "target")                          */
    int32_t  diag;     /* anchor diagonal = (b position) - (a position)                              */
    int32_t  max_ed;   /* give up beyond this many edits (0..SP_MAX_ED)                              */
} sp_pair;

typedef struct {
    int32_t ok;        /* 1 = alignment found within max_ed                                          */
    int32_t nm;        /* #X + #I bases + #D bases  (minimap2 Alignment.nm)                          */
    int32_t a_start, a_end;   /* half-open span on A                                                 */
    int32_t b_start, b_end;   /* half-open span on B                                                 */
    int32_t a_len, b_len;
} sp_aln;              /* 32 bytes */

#define SP_BAND    64
#define SP_MAX_ED  511
#define SP_KMER    16
#define SP_NO_DIAG INT32_MIN
/* event word: (type << 30) | b_pos ; b_pos = B bases consumed before the edit */
#define SP_EV_X 0u     /* mismatch                      (cigar 'X', op 8) */
#define SP_EV_D 1u     /* consumes one B base only      (cigar 'D', op 2 when B is the target) */
#define SP_EV_I 2u     /* consumes one A base only      (cigar 'I', op 1 when B is the target) */

/* k-mer vote anchor for each pair (a indexes set A = the k-mer indexed side).  Replaces minimap2's
 * seed+chain stage inside Aligner::map.  diag_out[i] = b_pos - a_pos, votes_out[i] = #votes (0: no anchor). */
int32_t sp_anchor_batch(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                        const uint32_t* a_idx, const uint32_t* b_idx, uint64_t n_pairs,
                        int32_t* diag_out, int32_t* votes_out);

/* Same, up to topk (1..8) placements per pair: peaks of the vote histogram taken best first, every bin within +-128
 * diagonals of a chosen peak cleared before the next one is taken (multi-copy targets, e.g. CYP2D6 and CYP2D7 in one read).
 * diag_out / votes_out hold n_pairs * topk entries; unused slots have votes 0. */
int32_t sp_anchor_batch_topk(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                             const uint32_t* a_idx, const uint32_t* b_idx, uint64_t n_pairs, int32_t topk,
                             int32_t* diag_out, int32_t* votes_out);

/* Align every pair.  Replaces every `aligner.map(...)` on the hot path: src/hla/realigner.rs:116,231,290;
 * src/hla/caller.rs:1277,1436; src/cyp2d6/haplotyper.rs:198,395; src/cyp2d6/chaining.rs:58.
 * events (optional, may be NULL): n_pairs * events_stride words, the first nm of each row are valid. */
int32_t sp_align_batch(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                       const sp_pair* pairs, uint64_t n_pairs,
                       sp_aln* out, uint32_t* events, uint32_t events_stride);

/* Two-piece affine re-score of alignments the library found: the numbers the reference reports.  Every (nm, start, end) of the reference is minimap2's
 * (standard_hifi_aligner, src/util/mapping.rs:8-14: match a = 1 -- 5 in score_read, src/hla/caller.rs:1370-1379 --, mismatch 4, gaps min(6 + 2 l, 26 + l),
 * ambiguous bases -1): the best local alignment through the chain's seeds.  For each pair (a = minimap2's query in set A, b = its target in set B, diag = the
 * diagonal b_pos - a_pos the library's own cell aligned it on) the banded Smith-Waterman optimum under those scores on `band` (64 or 256) diagonals around diag,
 * with the forward decisions and end rules of the restatement's DP (oracle/affine.c is the CPU statement): score, NM = mismatches + gap bases + ambiguous bases,
 * and the half-open spans on both sequences; score 0 = nothing aligns.  On the audited pair classes the numbers equal the minimap2 restatement's
 * (tests/test_oracle_affine.py).  sp_hla_realign_reads, sp_hla_score_consensus / sp_hla_type_consensus, and sp_cyp_find_regions report
 * them beside the library's own counts (fields mm2_*); sp_cyp_weight_segments takes its weights from them near a segment's minimum. */
typedef struct { int32_t a, b, q, e, q2, e2, sc_ambi; } sp_affine_opts;           /* map-hifi: 1, 4, 6, 2, 26, 1, 1 */
typedef struct { int32_t score, nm, a_start, a_end, b_start, b_end; } sp_affine_aln;
int32_t sp_affine_rescore_batch(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, const sp_pair* pairs, uint64_t n_pairs, const sp_affine_opts* opts,
                                int32_t band, sp_affine_aln* out);

/* ------------------------------------------------------------------ HLA database
 * Replaces HlaRealigner::new + create_hla_fasta (src/hla/realigner.rs:42-91,497-526) and the per-call
 * one-sequence indexes of score_read (src/hla/caller.rs:1370-1379).
 * Alleles must be given in database key order ("HLA:HLA00001" ascending = BTreeMap order,
 * src/hla/caller.rs:1413) because ties go to the lowest index.
 *   gene_of[i]      gene index of allele i
 *   dna / cdna      gene-strand sequences as stored in the database (dna length 0 = no DNA sequence)
 *   gene_ref        hg38-forward reference sequence of each gene region incl. the +-100 bp buffer
 *                   (src/hla/realigner.rs:74-81)
 *   gene_fwd[g]     is_forward_strand
 *   exon_start/end  exon coordinates of gene g relative to gene_ref[g] start (hg38 order),
 *                   exon_off[g]..exon_off[g+1]  
 */
typedef struct {
    uint32_t n_alleles, n_genes;
    const uint32_t* gene_of;
    const char* dna;   const uint64_t* dna_off;     /* n_alleles+1 */
    const char* cdna;  const uint64_t* cdna_off;    /* n_alleles+1 */
    const char* gene_ref; const uint64_t* gene_ref_off;   /* n_genes+1 */
    const uint8_t* gene_fwd;
    const uint32_t* exon_off;                        /* n_genes+1 */
    const int32_t* exon_start; const int32_t* exon_end;
    int32_t ref_buffer;                              /* bases of buffer on each side of the gene inside gene_ref (100) */
} sp_hla_db_desc;

int32_t sp_hla_db_create(sp_ctx* ctx, const sp_hla_db_desc* desc, sp_hla_db** out);
void    sp_hla_db_free(sp_hla_db* db);

/* ------------------------------------------------------------------ K1: read -> best database allele
 * Replaces HlaRealigner::realign_record (src/hla/realigner.rs:98-350) for a batch of reads.
 * reads: hg38-forward read sequences already on the device. For read r:
 *   best_allele  index of the accepted allele (lowest nm/(target_len-unmapped) subject to the 0.5 / 0.03
 *                cut-offs, :137-141) or -1;  gene = gene_of[best_allele]
 *   aln          alignment of the read (B) against that allele in hg38 orientation (A): bm.query_* = b_*,
 *                bm.target_* = a_*
 *   seg_start/end  optimal_segment_start..end on the read (:266-267), dna_offset / hpc_offset (:270-325);
 *                status 0 = realigned, 1 = no acceptable allele, 2 = best mapping not Forward (:178-193), 3 = segment failed to map to the gene reference.
 * Seeded mode (context option "k1_best_n" > 0, the default): the mappings a read is judged on are the ones minimap2 would return -- (19,19)-minimizer seeds through the occurrence
 * filter, chains per (strand, allele), the primary chains and the k1_best_n best secondaries within 0.8 of their primary -- base-aligned by the library's cell (64 diagonals around the
 * chain, 256 when that finds nothing) and re-scored with the two-piece affine scores; the acceptance loop runs over them in minimap2's output order on the re-scored numbers
 * (aln = the accepted mapping's cell, mm2_* = its re-score); status 2 = the accepted mapping is on the reverse strand.  k1_chains / k1_mappings / k1_chain_score: chains found for
 * the read, mappings that came back, chain score of the accepted one.  Exhaustive mode ("k1_best_n" = 0; always when cell_out is given): every allele of every gene the read anchors
 * in (>= 16 16-mer votes and >= 1/10 of the best gene's) is a cell, the lowest index wins ties, the counts are the unit-cost ones and mm2_* a report beside them; status 2 is decided at the
 * seeds (the read's best anchor on the reverse-complemented gene references has more 16-mer votes than its best forward anchor -- only reads with fewer than 512 forward votes are
 * anchored a second time, and a read that realigned acceptably forwards is only dropped when the reverse anchor has at least twice the votes; the other fields then hold what the
 * forward search found); k1_* are 0.
 */
typedef struct {
    int32_t status;
    int32_t best_allele;
    int32_t gene;
    int32_t nm, target_len, unmapped;         /* MappingStats of the best allele (:132-133) */
    sp_aln  aln;
    int32_t seg_start, seg_end;
    int32_t dna_offset, hpc_offset;
    /* the same mapping re-scored the reference's way (two-piece affine gaps, end clipping: sp_affine_rescore_batch on the 64 diagonals around aln): what minimap2
     * reports for this read and allele -- NM, the allele span (target_start / target_end), the read span (query_start / query_end); mm2_score 0 = not re-scored
     * (no best allele).  Exhaustive mode: a report beside the counts above, which decide; left zero with context option "mm2_rescore" 0.
     * Seeded mode: these ARE the numbers the acceptance loop ran on (<= 0.03 edit fraction, <= 0.5 penalised: mm2_nm over the re-scored extent) and they are written whatever
     * "mm2_rescore" says; nm / unmapped / aln above stay the unit-cost cell's numbers of the same mapping, so a record judged by nm / target_len alone may look as if it
     * missed or passed a cut-off it did not.  status 2 in seeded mode: best_allele is -1 and mm2_* describe the reverse-strand mapping that was accepted. */
    int32_t mm2_score, mm2_nm;
    int32_t mm2_t_start, mm2_t_end, mm2_q_start, mm2_q_end;
    int32_t k1_chains, k1_mappings, k1_chain_score;      /* seeded mode: chains of the read, mappings returned, chain score of the accepted mapping */
    int32_t reserved_;
} sp_hla_realign;

/* what the seeded map of a read came to (the audit entry point below; sp_hla_realign carries the same counts) */
#define SP_K1_SEL 16                      /* selected chains per read that are base-aligned (primaries + k1_best_n secondaries; the best ranked when a read has more) */
typedef struct { int32_t n_chains, n_selected, n_mappings, pick, chain_score, rev; } sp_k1_seed_info;
typedef struct {
    int32_t allele, rev, chain_score, n_seeds, t_len;
    int32_t sel_rank;                        /* position among the selected chains */
    int32_t diag;                            /* the cell's diagonal: read position - allele position, midway between the chain's outermost seeds */
    int32_t ok, cell_nm, a_start, a_end, b_start, b_end;   /* the unit-cost cell: A = allele, B = read on the mapped strand */
    int32_t dp_max, nm, t_start, t_end, q_start, q_end;    /* the re-scored mapping (read in forward coordinates, as minimap2 reports) */
    int32_t primary;
} sp_k1_seed_hit;
/* Audit of the seeded stage (tests hold every stage to oracle/mm2.c):
 *   sp_hla_seed_index_info   out[4] = minimizers in the index of the database's DNA alleles, distinct minimizers, the occurrence threshold mid_occ, indexed alleles (built on first use)
 *   sp_seqset_sketch         the (19,19)-minimizers of sequence idx of a set in position order: hash, end position of the k-mer, strand of the smaller k-mer; *n_out may exceed cap
 *   sp_hla_realign_seeded_audit  for read `read` of the set: its chains in rank order (chains[i*10..] = {indexed allele number, rev, chain score, seeds, query start, query end (forward
 *                            coordinates), target start, target end, -, selected}), its mappings in output order (hits, SP_K1_SEL entries), the accepted one (*pick or -1) and
 *                            counters[4] = {kept seeds, anchors, largest anchor count of a read, reads that hit a capacity} of the whole batch */
int32_t sp_hla_seed_index_info(sp_ctx* ctx, const sp_hla_db* db, int64_t* out /* 4 */);
int32_t sp_seqset_sketch(sp_ctx* ctx, const sp_seqset* set, uint32_t idx, uint64_t* hash, int32_t* end_pos, uint8_t* strand, uint32_t cap, uint32_t* n_out);
int32_t sp_hla_realign_seeded_audit(sp_ctx* ctx, const sp_hla_db* db, const sp_seqset* reads, uint32_t read, int32_t* chains, uint32_t chain_cap, uint32_t* n_chains,
                                    sp_k1_seed_hit* hits /* SP_K1_SEL */, uint32_t* n_hits, int32_t* pick, uint64_t* counters /* 4 */);

int32_t sp_hla_realign_reads(sp_ctx* ctx, const sp_hla_db* db, const sp_seqset* reads,
                             sp_hla_realign* out /* n_reads */,
                             uint32_t* cell_out /* optional n_reads*n_alleles: (nm<<16 | span) or 0xFFFFFFFF */);

/* ------------------------------------------------------------------ K2: consensus -> every allele of a gene
 * Replaces score_read's allele loop + HlaProcessedMatch (src/hla/caller.rs:1411-1510,
 * src/hla/processed_match.rs:53-263).  cons_dna / cons_cdna: gene-strand consensus and its spliced cDNA
 * (ASCII).  require_dna mirrors --hla-require-dna, disable_cdna mirrors --disable-cdna-scoring.
 *   best_allele  running-best winner in database order, -1 if nothing maps (best id stays empty, :1503-1509)
 *   stats        optional n_alleles*6: cdna (len, nm, unmapped), dna (len, nm, unmapped); -1,-1,-1 = None;
 *                rows of alleles of other genes are left at -2.
 */
typedef struct {
    int32_t best_allele;
    int32_t n_scored;            /* alleles visited (gene match and allowed) */
    /* HlaMappingStats of the best allele re-scored the reference's way (a = 5, two-piece affine gaps: sp_affine_rescore_batch): cDNA (len, nm, unmapped), DNA (len, nm,
     * unmapped); -1, -1, -1 = level absent or not re-scored (context option "mm2_rescore" 0).  The running-best scan itself uses the library's own counts. */
    int32_t mm2_stats[6];
} sp_hla_best;

int32_t sp_hla_score_consensus(sp_ctx* ctx, const sp_hla_db* db, uint32_t gene,
                               const char* cons_dna, uint32_t cons_dna_len,
                               const char* cons_cdna, uint32_t cons_cdna_len,
                               int32_t require_dna, int32_t disable_cdna,
                               sp_hla_best* best, int32_t* stats);

/* the same for n consensuses at once (different genes allowed): every (consensus, allele) pair of a level is one cell of one launch,
 * one scan workgroup per consensus.  No per-allele statistics in the batched form. */
int32_t sp_hla_score_consensus_batch(sp_ctx* ctx, const sp_hla_db* db, uint32_t n, const uint32_t* genes,
                                     const char* const* cons_dna, const uint32_t* cons_dna_len,
                                     const char* const* cons_cdna, const uint32_t* cons_cdna_len,
                                     int32_t require_dna, int32_t disable_cdna, sp_hla_best* best /* n */);

/* Replaces score_consensus (src/hla/caller.rs:1258-1319) + splice_read (:1518-1576): the hg38-forward consensus is
 * placed on the un-buffered gene reference (GPU alignment with traceback), its exon bases are spliced out through the
 * aligned pairs, both sequences are put on the gene strand and handed to the K2 scoring above.
 * An empty consensus or one that does not align to the reference gives best_allele = -1 and n_scored = 0
 * (caller.rs:1263-1267,1282-1287).  cdna_out (optional, cdna_cap bytes) receives the spliced gene-strand cDNA. */
int32_t sp_hla_type_consensus(sp_ctx* ctx, const sp_hla_db* db, uint32_t gene,
                              const char* consensus_fwd, uint32_t consensus_len,
                              int32_t require_dna, int32_t disable_cdna,
                              sp_hla_best* best, int32_t* stats,
                              char* cdna_out, uint32_t cdna_cap, uint32_t* cdna_len);

/* n hg38-forward consensuses at once: one placement launch, host splicing, one batched K2 (the gene drivers type all the
 * consensuses of a sample / cohort this way). */
int32_t sp_hla_type_consensus_batch(sp_ctx* ctx, const sp_hla_db* db, uint32_t n, const uint32_t* genes,
                                    const char* const* consensus_fwd, const uint32_t* consensus_len,
                                    int32_t require_dna, int32_t disable_cdna, sp_hla_best* best /* n */);

/* ------------------------------------------------------------------ K5: CYP2D6 chain-pair likelihood search
 * Replaces find_best_chain_pair (src/cyp2d6/chaining.rs:223-592) with containment_score (:683-731),
 * get_multinomial_score (:854-903), count_unexpected_alleles (:794-819), unexpected_count (:739-775),
 * count_inferred_edges (:828-840), check_chain_inferrences (:603-674) and multinomial_ln_pmf (src/util/stats.rs:11-37).
 * The grammar / enumeration runs on the host exactly as in the reference (LIFO DFS, copy number <= 3); every unordered
 * chain pair is then scored by one GPU thread in f64 with the reference's operation order, and the winner is
 * min (primary_score, i, j) -- what the reference's top-10 heap returns (chaining.rs:188-196,565-573).
 *   hap_type       Cyp2d6RegionType per consensus region (src/cyp2d6/region_label.rs:7-24), SP_CYP_* below
 *   hap_subtype    subtype label or NULL
 *   translate / connections / singletons   the three Cyp2d6Config tables (src/cyp2d6/definitions.rs:242-301)
 *   reads          in BTreeMap (qname) order: read r owns chains [read_chain_off[r], read_chain_off[r+1]) (obs_chains) and
 *                  weight rows [read_w_off[r], read_w_off[r+1]) (chain_scores); a row holds n_haps (edit distance, overlap) pairs
 * Returns SP_OK, or SP_ERR_NO_CHAINING_HEAD / SP_ERR_NO_CHAINS_FOUND / SP_ERR_NO_SCORE_PAIRS (CallerError, expected failures).
 */
enum { SP_CYP_UNKNOWN = 0, SP_CYP_REP6 = 1, SP_CYP_CYP2D6 = 2, SP_CYP_LINK_REGION = 3, SP_CYP_REP7 = 4, SP_CYP_SPACER = 5,
       SP_CYP_CYP2D7 = 6, SP_CYP_DELETION = 7, SP_CYP_HYBRID = 8, SP_CYP_FALSE_ALLELE = 9 };
#define SP_MAX_CHAIN 64

typedef struct {
    uint32_t n_haps;
    const int32_t* hap_type;
    const char* const* hap_subtype;
    uint32_t n_translate; const char* const* translate_key; const char* const* translate_val;
    uint32_t n_connections; const char* const* connection_a; const char* const* connection_b;
    uint32_t n_singletons; const char* const* singletons;
    uint32_t n_reads;
    const uint32_t* read_chain_off;     /* n_reads+1 */
    const uint32_t* chain_off;          /* n_chains+1 */
    const uint32_t* chain_items;
    const uint32_t* read_w_off;         /* n_reads+1 */
    const uint64_t* w_ed;               /* [row][n_haps] */
    const double*   w_ov;               /* [row][n_haps] */
    int32_t infer_connections, normalize_all_alleles, ignore_chain_label_limits;
    double lasso_penalty, ln_ed_penalty, unexpected_chain_penalty, inferred_edge_penalty;   /* ChainPenalties, chaining.rs:107-139 */
} sp_chain_problem;

typedef struct {
    int32_t n_possible;                 /* enumerated chains */
    int32_t index1, index2;             /* winning pair (i <= j) in enumeration order */
    int32_t n1, n2;
    int32_t chain1[SP_MAX_CHAIN], chain2[SP_MAX_CHAIN];    /* the pair, sorted (chaining.rs:568-573) */
    double score, ln_ed_penalty, mn_llh_penalty, allele_expected_penalty, unexpected_chain_penalty, inferred_chain_penalty;
    uint64_t edit_distance;
    uint64_t n_pairs_scored;            /* pairs whose read-level terms were evaluated on the GPU */
} sp_chain_result;

int32_t sp_cyp_best_chain_pair(sp_ctx* ctx, const sp_chain_problem* problem, sp_chain_result* result);

/* ------------------------------------------------------------------ K3: CYP2D6 template search in reads
 * Replaces Cyp2d6Extractor::find_base_type_in_sequence (src/cyp2d6/haplotyper.rs:142-315) for a batch of reads:
 * every template (D6, D7, *5 signature, hybrids, REP6/REP7, spacer, link_region -- given in the reference's key order,
 * i.e. sorted by full_allele(), :175-183) is placed on every read (up to 4 placements), hits with more than 5 % edits
 * are dropped (un-mapped template bases count only for *5 / REP6 / REP7, :185-191,228-232), hits are sorted by
 * (start, end), overlapping hits (> 0.9 of the shorter) collapse to the better one with *5 priority (:260-296), and hits
 * whose penalised fraction exceeds max_missing_frac are dropped (:303-306).
 * hits: grouped by read, in output order; returns the total in n_hits (may exceed hits_cap: call again with more room). */
typedef struct {
    int32_t read, template_idx;
    int32_t start, end;                 /* region on the read */
    int32_t seq_len, nm, unmapped;      /* MappingStats of the template */
    int32_t clip_start, clip_end;
    /* the hit re-scored the reference's way (two-piece affine gaps, end clipping; 256 diagonals: sp_affine_rescore_batch): minimap2's NM, region on the read and
     * template span for this mapping; mm2_score 0 = not re-scored (context option "mm2_rescore" 0).  The filters above use the library's own counts. */
    int32_t mm2_score, mm2_nm;
    int32_t mm2_start, mm2_end;         /* region on the read */
    int32_t mm2_q_start, mm2_q_end;     /* span on the template: unmapped = seq_len - (mm2_q_end - mm2_q_start), clips = mm2_q_start, seq_len - mm2_q_end */
} sp_region_hit;

int32_t sp_cyp_find_regions(sp_ctx* ctx, const sp_seqset* templates, const int32_t* template_type, const sp_seqset* reads,
                            double max_missing_frac, sp_region_hit* hits, uint64_t hits_cap, uint64_t* n_hits);

/* ------------------------------------------------------------------ K4: consensus weights of read segments
 * Replaces weight_sequence (src/cyp2d6/chaining.rs:28-103) for a batch of read segments: every allowed consensus
 * (allowed[c] = label.is_allowed_label()) is placed on every segment; ed[s][c] = nm + un-mapped segment bases,
 * ov[s][c] = 1 - clipped consensus fraction, defaults (segment length, 0.0); kept[s] = 0 when the reference returns the
 * empty vector (best penalised fraction > 0.05).  With the context option "mm2_rescore" (default 1) a placement whose nm + un-mapped bases are within 16 of the
 * segment's smallest over the allowed consensuses enters with the reference's numbers -- re-scored with minimap2's two-piece affine scores on 256 diagonals
 * (sp_affine_rescore_batch): its NM, spans and clips --, the others with the unit-cost count (a lower bound of the other). */
int32_t sp_cyp_weight_segments(sp_ctx* ctx, const sp_seqset* consensus, const uint8_t* allowed, const sp_seqset* segments,
                               uint64_t* ed, double* ov, uint8_t* kept);

/* ------------------------------------------------------------------ K7: star-allele vector scoring
 * Replaces the scoring loop of Cyp2d6Extractor::assign_haplotype (src/cyp2d6/haplotyper.rs:470-524): for each typed
 * sequence, its per-variant state vector (0 ref, 1 alt, 2 ambiguous, 3 unset) is compared with the 0/1 definition of
 * every star allele; score = (matches at VI variants, matches at all variants), 2 always matches, 3 never does.
 *   hap_matrix   n_alleles x n_variants (0/1), alleles in haplotype_lookup (BTreeMap) order
 *   is_vi        n_variants (LoadedVariants::is_vi)
 *   states       n_seqs x n_variants
 * Outputs per sequence: best (vi_match, all_match) (starting from the (0,0) of the Unknown label, :472-474) and, in
 * tie_mask (n_seqs x n_alleles bytes), which alleles reach it; no allele beats or ties (0,0) => all zero.
 * The caller resolves ties by sorting full_allele() strings (:531-547). */
int32_t sp_cyp_score_alleles(sp_ctx* ctx, uint32_t n_variants, uint32_t n_alleles, const uint8_t* hap_matrix, const uint8_t* is_vi,
                             uint32_t n_seqs, const uint8_t* states, uint32_t* best_vi, uint32_t* best_all, uint8_t* tie_mask);

/* ------------------------------------------------------------------ K9: per-variant state of typed sequences
 * The role of WFAGraph::from_reference_variants + edit_distance_with_pruning + the traversed-node loop of assign_haplotype
 * (src/cyp2d6/haplotyper.rs:371-468): which allele of every database variant a consensus carries.  hiphase v1.2.1 is not on disk;
 * the contract: the sequence is placed on the backbone (GPU alignment with traceback); for every variant whose
 * reference span lies inside the aligned part, the sequence window facing [p - 24, p + |ref| + 24) is compared (global edit
 * distance, one GPU thread per pair) with that backbone window carrying the reference allele and carrying the alternate allele:
 * 0 = closer to the reference, 1 = closer to the alternate, 2 = equally close (ambiguous), 3 = not covered / not aligned.
 *   backbone    the CYP2D6_wfa_backbone slice of the reference; var_pos 0-based on it; var_ref / var_alt normalised ACGT alleles
 *   states      sp_seqset_count(seqs) * n_variants, ready for sp_cyp_score_alleles;  alns (optional) placement of every sequence */
int32_t sp_cyp_variant_states(sp_ctx* ctx, const sp_seqset* seqs, const char* backbone, uint32_t backbone_len, uint32_t n_variants,
                              const int32_t* var_pos, const char* const* var_ref, const char* const* var_alt,
                              uint8_t* states, sp_aln* alns);

/* ------------------------------------------------------------------ CYP2D6, reads to diplotype
 * diplotype_cyp2d6 (src/cyp2d6/caller.rs:39-741) on top of K3 / K8 / K9 / K7 / K4 / K5, without I/O and debug artefacts:
 * regions of interest (:126-139) -> consensus inputs with offsets and seeds (:176-245) -> multi-way consensus (:268) ->
 * merge_consensus_results (:750-898) -> typing of every consensus with find_full_type_in_sequence / assign_haplotype
 * (src/cyp2d6/haplotyper.rs:326-602) and duplicate detection (:331-375) -> weights, chains, best chain pair (:429-640) ->
 * haplotype strings (:660-690).
 *   templates          Cyp2d6Extractor::hybrid_sequences in full_allele() order (haplotyper.rs:175-183) with type, subtype label and
 *                      template_deep[t] = 1 when the label is in mapped_hybrids (typed further by its variants)
 *   backbone, var_*    the CYP2D6_wfa_backbone slice and LoadedVariants::ordered_variants on it (0-based), is_vi flags
 *   allele_subtype / hap_matrix   haplotype_lookup in BTreeMap order: star-allele label ("4.001") and its 0/1 vector
 *   translate / connections / singletons   the three Cyp2d6Config tables
 * call->status: 0 called, 1 = no reads (NO_READS), 7 = chain collapse, 16/17/18 = CallerError (NO_MATCH).  hap1/hap2 carry the
 * sub-allele strings, core1/core2 the core-allele strings (Cyp2d6DetailLevel::SubAlleles / CoreAlleles). */
typedef struct {
    const sp_seqset* templates; const int32_t* template_type; const char* const* template_subtype; const uint8_t* template_deep;
    const char* backbone; uint32_t backbone_len;
    uint32_t n_variants; const int32_t* var_pos; const char* const* var_ref; const char* const* var_alt; const uint8_t* var_is_vi;
    uint32_t n_alleles; const char* const* allele_subtype; const uint8_t* hap_matrix;
    uint32_t n_translate; const char* const* translate_key; const char* const* translate_val;
    uint32_t n_connections; const char* const* connection_a; const char* const* connection_b;
    uint32_t n_singletons; const char* const* singletons;
    int32_t min_consensus_count, dual_max_ed_delta; double min_consensus_fraction;
    int32_t infer_connections, normalize_d6_only;
    const char* const* var_label;                 /* LoadedVariants::variant_label per variant, for the deep strings of the call; NULL: none are listed */
} sp_cyp_problem;

#define SP_CYP_MAXCONS 64
typedef struct {
    int32_t status;
    int32_t n_consensus;                          /* final consensus regions (hap_regions) */
    int32_t cons_type[SP_CYP_MAXCONS];            /* SP_CYP_* after typing, duplicate and false-allele marking */
    char    cons_subtype[SP_CYP_MAXCONS][48];     /* "" = none */
    int32_t n1, n2; int32_t chain1[SP_MAX_CHAIN], chain2[SP_MAX_CHAIN];
    double  score;
    char hap1[256], hap2[256], core1[256], core2[256];
    /* Cyp2d6DetailLevel::DeepAlleles (src/cyp2d6/caller.rs:907-957, Cyp2d6Region::deep_label, src/cyp2d6/region.rs:47-91): every reported
     * region as "(<index>_<full allele>[ +label | -label | ?label ...])" with the variants that differ from the assigned star allele --
     * the haplotypes of the InexactDiplotype the reference stores for CYP2D6 (:711-716,737); truncated to the buffer */
    char deep1[2048], deep2[2048];
    int32_t searches_gave_up;                     /* two-way searches of the multi-way consensus that ended without a complete node: their groups stayed whole */
    int32_t reserved_;
} sp_cyp_call;

int32_t sp_cyp_diplotype(sp_ctx* ctx, const sp_cyp_problem* problem, const sp_seqset* reads, sp_cyp_call* call,
                         char* consensus /* optional: SP_CYP_MAXCONS * cons_cap bytes */, uint32_t cons_cap);

/* One GPU's share of a cohort: the call of sp_cyp_diplotype for each of n_samples read sets against the same problem.  The samples are
 * spread over the context's streams (sp_ctx_set_option), one host thread per stream for the length of the call; calls[i], the
 * consensus block of sample i (optional: n_samples * SP_CYP_MAXCONS * cons_cap bytes) and sample_rc[i] (optional) are what the single
 * call would give.  Returns the first status that is not SP_OK. */
int32_t sp_cyp_diplotype_cohort(sp_ctx* ctx, const sp_cyp_problem* problem, uint32_t n_samples, const sp_seqset* const* reads, sp_cyp_call* calls,
                                char* consensus, uint32_t cons_cap, int32_t* sample_rc);

/* The same call, also handing out Cyp2d6Region::variants of every final consensus region (assign_haplotype,
 * src/cyp2d6/haplotyper.rs:546-595): has_variants[h] = 1 when region h carries a list (typed as a CYP2D6 star allele), and
 * state[h * n_variants + v] is the VariantAlleleRelationship of variant v against the assigned allele in the codes of
 * sp_inexact_haplotype (1 Match .. 7 UnknownMissing), 255 = not listed (reference where the allele has the reference).
 * state: caller's buffer of SP_CYP_MAXCONS * n_variants bytes (NULL: only has_variants is filled).
 * sp_cyp_alleles_json writes the `cyp2d6_alleles.json` debug file (DeeplotypeDebug, src/cyp2d6/debug.rs:10-70; caller.rs:703-707) from
 * the call and these lists: the two haplotypes in deep / sub-allele / core form and {index_label: [RegionVariant]} in key order.
 * out / cap: the text is copied NUL-terminated; *needed = bytes needed incl. the NUL; SP_ERR_CAPACITY when it did not fit. */
typedef struct { uint8_t has_variants[SP_CYP_MAXCONS]; uint8_t* state; } sp_cyp_region_variants;
int32_t sp_cyp_diplotype_detailed(sp_ctx* ctx, const sp_cyp_problem* problem, const sp_seqset* reads, sp_cyp_call* call,
                                  char* consensus, uint32_t cons_cap, sp_cyp_region_variants* region_variants /* optional */);
int32_t sp_cyp_alleles_json(const sp_cyp_problem* problem, const sp_cyp_call* call, const sp_cyp_region_variants* region_variants,
                            char* out, uint64_t cap, uint64_t* needed);

/* ------------------------------------------------------------------ CYP2D6 templates and typing tables
 * Replaces generate_cyp_hybrids (src/cyp2d6/definitions.rs:346-464), LoadedVariants::load_variant_database
 * (src/cyp2d6/haplotyper.rs:650-773) and the table building of Cyp2d6Extractor::new (src/cyp2d6/haplotyper.rs:45-132).
 *   sp_cyp_locus     Cyp2d6Config::{cyp_coordinates, cyp_regions, cyp2d6_star5_del} (definitions.rs:17-30) as 0-based half-open
 *                    chromosome coordinates, plus the chromosome bases of a window that contains all of them (the reference reads
 *                    them with ReferenceGenome::get_slice); exon arrays hold exon1..exon9 of CYP2D6 / CYP2D7
 *   sp_cyp_gene_def  PgxDatabase::cyp2d6_gene_def (src/database/pgx_database.rs:41; AlleleDefinition / VariantDefinition,
 *                    src/data_types/alleles.rs:7-81) flattened in key (BTreeMap) order: allele a owns variants
 *                    [var_off[a], var_off[a+1]); var_id[x] == NULL means Option::None (the label becomes variant_string()),
 *                    var_vi[x] == NULL means no "VI" entry in extras
 *   sp_cyp_config    Cyp2d6Config::{cyp_translate, inferred_connections, unexpected_singletons} (definitions.rs:242-301)
 * Built: the 39 search templates in the visiting order of find_base_type_in_sequence (sorted by full_allele(),
 * haplotyper.rs:175-183) with type, subtype label and deep-typing flag (mapped_hybrids, :117-123); the ordered variant table
 * (first occurrence per (position, ref, alt), stable sort by position), labels, VI flags, label and variant look-ups; one 0/1 row per
 * star allele in haplotype_lookup (BTreeMap<Cyp2d6RegionLabel, _>) order.  ctx may be NULL: the tables are then host only
 * (sp_cyp_db_problem needs a context: it hands out the templates as a packed set in HBM).
 * Errors: SP_ERR_INVALID_ARG when a coordinate leaves the window, the CYP2D6 region does not contain every variant (the reference
 * asserts, haplotyper.rs:110-111) or a variant leaves the backbone. */
typedef struct {
    const char* chrom_name;                       /* "chr22" (used for variant_string() labels); NULL = "chr22" */
    const char* chrom_seq; uint64_t window_start, window_len;
    uint64_t d6_start, d6_end, d7_start, d7_end, rep6_start, rep6_end, rep7_start, rep7_end;
    uint64_t spacer_start, spacer_end, link_start, link_end, backbone_start, backbone_end, star5_start, star5_end;
    uint64_t d6_exon_start[9], d6_exon_end[9], d7_exon_start[9], d7_exon_end[9];
} sp_cyp_locus;
typedef struct {
    uint32_t n_alleles;
    const char* const* star_allele;               /* "4.001" */
    const uint32_t* var_off;                      /* n_alleles + 1 */
    const uint64_t* var_pos;                      /* 0-based on the chromosome */
    const char* const* var_ref; const char* const* var_alt;
    const char* const* var_id; const char* const* var_vi;     /* either array may be NULL altogether */
} sp_cyp_gene_def;
typedef struct {
    uint32_t n_translate; const char* const* translate_key; const char* const* translate_val;
    uint32_t n_connections; const char* const* connection_a; const char* const* connection_b;
    uint32_t n_singletons; const char* const* singletons;
} sp_cyp_config;
typedef struct {
    uint32_t n_templates, n_variants, n_vi, n_alleles, backbone_len;
    int64_t first_variant_pos, last_variant_pos;  /* LoadedVariants::{first,last}_variant_pos, -1 when there is none */
} sp_cyp_db_stats;
typedef struct sp_cyp_db sp_cyp_db;
int32_t sp_cyp_db_create(sp_ctx* ctx /* may be NULL */, const sp_cyp_locus* locus, const sp_cyp_gene_def* gene_def, const sp_cyp_config* config /* may be NULL */,
                         sp_cyp_db** out);
void    sp_cyp_db_free(sp_cyp_db* db);
int32_t sp_cyp_db_info(const sp_cyp_db* db, sp_cyp_db_stats* stats);
/* the pointers stay valid for the life of the database; subtype is NULL for a label without one */
int32_t sp_cyp_db_template(const sp_cyp_db* db, uint32_t i, int32_t* type, const char** subtype, const char** full_allele, const char** seq, uint32_t* len, int32_t* deep);
int32_t sp_cyp_db_variant(const sp_cyp_db* db, uint32_t i, int64_t* chrom_pos, const char** ref, const char** alt, const char** label, int32_t* is_vi);
int32_t sp_cyp_db_index_label(const sp_cyp_db* db, const char* label, uint32_t* idx);                                  /* LoadedVariants::index_label */
int32_t sp_cyp_db_index_variant(const sp_cyp_db* db, uint64_t position, const char* ref, const char* alt, uint32_t* idx);   /* LoadedVariants::index_variant */
int32_t sp_cyp_db_allele(const sp_cyp_db* db, uint32_t a, const char** subtype, const uint8_t** row /* n_variants 0/1 */);
/* everything sp_cyp_diplotype needs, pointing into the database (valid for its life); run parameters get the CLI defaults
 * (min_consensus_count 3, min_consensus_fraction 0.10, dual_max_ed_delta 100, no inferred connections, all alleles normalise) */
int32_t sp_cyp_db_problem(const sp_cyp_db* db, sp_cyp_problem* problem);

/* ------------------------------------------------------------------ K6: variant-gene diplotype search
 * Replaces solve_diplotype (src/diplotyper.rs:1211-1371) with find_best_inexact_matches (:1411-1509) and
 * NormalizedPgxHaplotype::quant_match (src/data_types/normalized_variant.rs:431-479) on integer ids.  The caller keeps the
 * string work (normalisation, names): variants are ids 0..n_vars-1, haplotypes are listed in defined_haplotypes (BTreeMap)
 * order, observed variants in BTreeMap<NormalizedVariant> order.
 *   haplotype h  = AND of slots [slot_off[h], slot_off[h+1]); slot s = OR of alt_var[alt_off[s] .. alt_off[s+1]) (-1 = None)
 *   obs_gt       SP_GT_* ; obs_ps = phase set or -1 ; obs_sv_label = -1 or the id of the SV haplotype label it carries
 * Every (het assignment, haplotype side, database haplotype) cell is scored on the GPU (one thread per cell); the host
 * combines the sides exactly as the reference does (ascending combinations, ties kept, sub-alleles shadow core alleles).
 * dip[i] = the two haplotype indices of diplotype i (an SV label l is encoded as -(l + 2)); dip_comb[i] = the het
 * assignment it came from (needed to derive the inexact haplotypes, :1516-1550). */
enum { SP_GT_HOM_REF = 0, SP_GT_HET_UNPHASED = 1, SP_GT_HET_PHASED = 2, SP_GT_HET_FLIP = 3, SP_GT_HOM_ALT = 4 };
#define SP_VAR_MAXDIP 4096

typedef struct {
    int32_t n_haps;
    const uint8_t* hap_is_sv;
    const uint8_t* hap_is_core;
    const int32_t* slot_off;
    const int32_t* alt_off;
    const int32_t* alt_var;
    int32_t n_vars;
    const uint8_t* var_is_core;
    int32_t n_obs;
    const int32_t* obs_var;
    const int32_t* obs_gt;
    const int64_t* obs_ps;
    const int32_t* obs_sv_label;
} sp_variant_problem;

typedef struct {
    int64_t score[4];                   /* core missing, core extra, sub missing, sub extra (INT64_MAX = unset) */
    int32_t n_dip, overflow;
    int32_t dip[SP_VAR_MAXDIP][2];
    int32_t dip_comb[SP_VAR_MAXDIP];
} sp_variant_result;

int32_t sp_variant_solve(sp_ctx* ctx, const sp_variant_problem* problem, sp_variant_result* result);
/* the solves of a panel / of one GPU's share of a cohort in one call, handed out to the context's streams (sp_ctx_set_option):
 * results[i] and problem_rc[i] (optional) are what sp_variant_solve gives for problems[i]; returns the first status that is not SP_OK */
int32_t sp_variant_solve_batch(sp_ctx* ctx, uint32_t n, const sp_variant_problem* const* problems, sp_variant_result* results, int32_t* problem_rc);

/* is_deletion (src/diplotyper.rs:1020-1174): which defined deletion haplotype, if any, a deleted region [start, end) (0-based,
 * end exclusive) is; its label is what load_sv_vcf_variants (:739-857) attaches to the observed SV (obs_sv_label above).
 * Full-gene deletions are tried first (is_full_deletion, :1034-1089: a gene counts when its coordinates lie inside the region),
 * then partial ones (is_partial_deletion, :1098-1174: the first..last exon inside the region, indices mirrored on the reverse
 * strand).  Within a class the definitions are walked in the order given -- the caller lists them in label (BTreeMap) order
 * -- a generic definition matches when its genes are all among the deleted ones and is kept until a later one matches, a
 * specific definition needs the exact set (genes, and exon ranges for partial ones) and ends the walk.
 *   genes g = 0..n_genes-1: gene_start/gene_end, gene_forward, exons [exon_off[g], exon_off[g+1]) in reference order
 *   full definition d: genes full_gene[full_off[d] .. full_off[d+1])
 *   partial definition d: entries [partial_off[d], partial_off[d+1]) = (partial_gene, exons partial_first .. partial_end exclusive)
 *   a gene id of -1 = a gene without a definition in the gene collection -> SP_ERR_BAD_ARG (the reference bails, :1042-1046)
 * kind: 0 no match, 1 full deletion, 2 partial deletion; index: the matching definition within its class. */
typedef struct {
    int32_t n_genes;
    const int64_t* gene_start;
    const int64_t* gene_end;
    const uint8_t* gene_forward;
    const int32_t* exon_off;
    const int64_t* exon_start;
    const int64_t* exon_end;
    int32_t n_full;
    const uint8_t* full_generic;
    const int32_t* full_off;
    const int32_t* full_gene;
    int32_t n_partial;
    const uint8_t* partial_generic;
    const int32_t* partial_off;
    const int32_t* partial_gene;
    const int32_t* partial_first;
    const int32_t* partial_end;
} sp_sv_definitions;

int32_t sp_variant_is_deletion(const sp_sv_definitions* defs, uint64_t start, uint64_t end, int32_t* kind, int32_t* index);

/* ------------------------------------------------------------------ K8: read consensus by dynamic wavefront alignment
 * Serves the waffle_con calls of the reference: DualConsensusDWFA::{add_sequence_offset, consensus} in
 * run_dual_consensus_with_offsets (src/hla/caller.rs:1103-1219) and ConsensusDWFA per read group (src/hla/caller.rs:706-747),
 * with the fields of dwfa_config_from_cli (src/hla/caller.rs:1103-1116).  waffle_con v0.4.4 is not on disk: the contract is the
 * one in DESIGN.md section 9 / oracle/consensus.c (every read keeps a 64-diagonal edit wavefront against the growing consensus
 * and votes for the next base; extensions are explored best first -- lowest total edit distance, then longest -- under the bounds
 * max_queue_size / max_capacity_per_size / max_nodes_wo_constraint; a candidate needs min_count reads and min_af of the votes;
 * a pair of candidates may start a second consensus).
 *   reads / read_idx   the sequences (read_idx == NULL: all n = sp_seqset_count(reads) of them, in order)
 *   offsets            NULL, or per sequence -1 (None: starts with the consensus) or the consensus length at which the sequence is
 *                      placed; its start is searched in the offset_window bases before that point (add_sequence_offset)
 *   cons1 / cons2      cap bytes each, NUL terminated ASCII; cons2 is empty unless result->is_dual
 *   is_cons1, score1, score2   DualConsensus::{is_consensus1, scores1, scores2}; a score of -1 is None
 * sp_consensus runs the search with cfg->allow_dual as given (ConsensusDWFA / DualConsensusDWFA); sp_consensus_dual is the same
 * with a second consensus allowed. */
typedef struct {
    int32_t min_count;                 /* 3 */
    int32_t dual_max_ed_delta;         /* 100 */
    int32_t allow_early_termination;
    int32_t allow_dual;
    int32_t offset_window;             /* 400 */
    int32_t offset_compare_length;     /* 50 for HLA (src/hla/caller.rs:1114), 100 for CYP2D6 (src/cyp2d6/caller.rs:144).  At most 128; at most 64 when offset_window + offset_compare_length exceeds 512.
                                        * A read with an offset is placed when the consensus reaches offset + min(offset_compare_length, its length): its start is searched in
                                        * [offset - offset_window, offset], its first bases compared with the consensus behind each start (free end on the consensus) */
    double  min_af;                    /* 0.10 */
    int32_t max_queue_size;            /* 20    CdwfaConfig::max_queue_size, set by dwfa_config_from_cli (src/hla/caller.rs:1110) */
    int32_t max_capacity_per_size;     /* 10    CdwfaConfig::max_capacity_per_size (:1111) */
    int32_t max_nodes_wo_constraint;   /* 1000  waffle_con's default; the reference does not set it */
    int32_t no_retry_ladder;           /* (<= 0 in any of the three above: the value named there)  sp_consensus_priority only: 1 = a two-way search that
                                        * gives up is NOT run again with stricter fractions (see there); sp_cyp_diplotype* set it unless the context's "cons_retry_ladder" is 1 */
} sp_cons_config;

typedef struct {
    int32_t is_dual, len1, len2, split_at;
    int64_t gave_up, best_total;       /* gave_up = 1: the search ended without a complete node (its bounds exhausted): no consensus, every read in group 1 */
    int64_t split_w2, split_total;     /* (best_total and this pair are unused since the best-first search replaced the two-pass split policy) */
    int64_t nodes_expanded;            /* nodes the search took out of its queue and expanded */
} sp_cons_result;

/* batched form: independent problems advance in lockstep (one base per kernel launch for all of them), so a batch costs as many
 * launches as its longest consensus.  outputs[p].status is SP_OK or SP_ERR_CAPACITY (cap too small for that consensus). */
typedef struct {
    const sp_seqset* reads; const uint32_t* read_idx; uint32_t n; const int32_t* offsets;
    sp_cons_config cfg;
} sp_cons_problem;
typedef struct {
    char* cons1; char* cons2; uint32_t cap;
    uint8_t* is_cons1; int32_t* score1; int32_t* score2;
    sp_cons_result result; int32_t status;
} sp_cons_output;
int32_t sp_consensus_batch(sp_ctx* ctx, uint32_t n_problems, const sp_cons_problem* problems, sp_cons_output* outputs);
int32_t sp_consensus_dual_batch(sp_ctx* ctx, uint32_t n_problems, const sp_cons_problem* problems, sp_cons_output* outputs);

int32_t sp_consensus(sp_ctx* ctx, const sp_seqset* reads, const uint32_t* read_idx, uint32_t n, const int32_t* offsets,
                     const sp_cons_config* cfg, char* cons1, char* cons2, uint32_t cap,
                     uint8_t* is_cons1, int32_t* score1, int32_t* score2, sp_cons_result* result);
int32_t sp_consensus_dual(sp_ctx* ctx, const sp_seqset* reads, const uint32_t* read_idx, uint32_t n, const int32_t* offsets,
                          const sp_cons_config* cfg, char* cons1, char* cons2, uint32_t cap,
                          uint8_t* is_cons1, int32_t* score1, int32_t* score2, sp_cons_result* result);

/* multi-way consensus by repeated two-way splits: the role of PriorityConsensusDWFA::{add_seeded_sequence_chain, consensus} in the
 * CYP2D6 caller (src/cyp2d6/caller.rs:162-270).  Every read contributes one sequence per level (the reference chains the
 * homopolymer-compressed segment, then the raw segment) with its own offset; seeds force an initial grouping (:231-238).
 * Contract: solve(group, level) = run sp_consensus_dual on the group's level sequences; if it splits, solve(consensus-1 reads,
 * level) followed by solve(the others, level); else solve(group, level + 1), or emit the group after the last level.  Initial
 * groups: the unseeded reads, then each seed in ascending order.  Offsets inside a group are re-based on the group's smallest
 * one (that read starts the consensus, the others keep their distance to it plus half the window).  Finally every emitted group
 * gets one consensus per level: the consensus its own two-way search at that level left when that search ended with ONE consensus at the configured
 * fraction (the search that found the group indivisible), otherwise (a group born from a split at a later level, a search that gave up or was retried) a
 * search of its own (sp_consensus).  All problems of a round run in lockstep on the GPU.  A two-way search that gives up (no
 * complete node: a group of more classes than a search holds consensuses can exhaust the queue / capacity bounds at high depth) is run
 * again with min_af 0.15, 0.20, 0.30, 0.40 (the first above the configured one that completes); the groups it leaves are solved with the
 * configured fraction again.  This retry is a rule of this library, not of waffle_con: cfg.no_retry_ladder = 1 switches it off (the group then
 * stays whole, as a search that gives up leaves it).  "Gave up" is sp_cons_result.gave_up of the two-way search, not an empty string.
 *   levels[l]   the sequences of level l (n each, same read order); offsets[l] = NULL or n entries (-1 = None); seeds = NULL or n (-1 = None)
 *   group_of    n entries: index of the emitted group of every read (MultiConsensus::sequence_indices)
 *   cons        max_groups * n_levels * cap bytes: consensus of group g at level l at cons + (g * n_levels + l) * cap
 * Returns SP_ERR_CAPACITY when there are more than max_groups groups (n_groups then holds the number needed). */
typedef struct {
    uint32_t n_levels, n;
    const sp_seqset* const* levels;
    const int32_t* const* offsets;
    const int32_t* seeds;
    sp_cons_config cfg;
} sp_priority_problem;
int32_t sp_consensus_priority(sp_ctx* ctx, const sp_priority_problem* problem, uint32_t max_groups, uint32_t cap,
                              uint32_t* n_groups, int32_t* group_of, char* cons);

/* Several multi-way consensus problems (the samples of a cohort) in lockstep: what sp_consensus_priority does for each, with all open groups of all
 * problems of a round in the same launches.  status: SP_OK, or why this job alone could not be solved (SP_ERR_CAPACITY: more groups than max_groups). */
typedef struct sp_priority_job {
    const sp_priority_problem* problem;
    uint32_t max_groups, cap;
    uint32_t* n_groups; int32_t* group_of; char* cons;      /* as the arguments of sp_consensus_priority */
    int32_t status;
    int32_t gave_up;                                        /* out: two-way searches of this job that ended without a complete node (sp_cons_result.gave_up) and were not retried */
} sp_priority_job;
int32_t sp_consensus_priority_many(sp_ctx* ctx, uint32_t n_jobs, sp_priority_job* jobs);

/* ------------------------------------------------------------------ one HLA gene, reads to diplotype
 * The gene loop of diplotype_hla_batch (src/hla/caller.rs:642-1040) on top of K1 / K8 / K2, without I/O and debug artefacts:
 * the realigned segments of `gene` (realign[r].status == 0, in input order = qname order) are cut out of the packed reads and
 * homopolymer-compressed on the device; run_dual_consensus_with_offsets (:1118-1219: HPC first, DNA if is_passing_dual fails);
 * is_hemizygous_better when the gene is absent-capable (:676-684); one consensus per read group (:706-747); score_consensus on
 * each (:756,829); heterozygous if the dual passes, else homozygous for the larger group (:889-912).
 *   realign      the K1 output for `reads` (sp_hla_realign_reads)
 *   cons1/cons2  cap bytes each: the hg38-forward consensus of each group (empty = none / failed)
 *   is_cons1     optional, one entry per read of `reads`: DualConsensus::is_consensus1 (0 for reads not realigned to the gene)
 * call->status: 0 = called, 1 = no realigned reads (NO_READS / NO_CALL, :662-668).  allele1/allele2 are database indices,
 * -1 = unknown (no allele typed), -2 = the absent haplotype of a hemizygous call (:919-923). */
typedef struct {
    int32_t min_consensus_count;       /* --min-consensus-count (3) */
    int32_t dual_max_ed_delta;         /* --dual-max-ed-delta (100) */
    double  min_consensus_fraction;    /* --min-consensus-fraction (0.10) */
    double  expected_maf, min_cdf;     /* is_passing_dual (src/hla/caller.rs:1225-1247) */
    int32_t require_dna, disable_cdna; /* --hla-require-dna, --disable-cdna-scoring */
    int32_t absent_capable;            /* gene_def.is_absent_capable() */
    double  normalized_coverage;       /* < 0: unknown */
} sp_hla_call_config;

typedef struct {
    int32_t status;
    int32_t allele1, allele2;
    int32_t typed1, typed2;            /* best database allele of each consensus (-1 = none) */
    int32_t n_reads, counts1, counts2;
    int32_t is_dual, dual_passed, is_hemizygous, used_dna_dual;
    int32_t cons1_len, cons2_len;
    double  maf, cdf;
} sp_hla_call;

int32_t sp_hla_diplotype_gene(sp_ctx* ctx, const sp_hla_db* db, uint32_t gene, const sp_seqset* reads, const sp_hla_realign* realign,
                              const sp_hla_call_config* cfg, sp_hla_call* call, char* cons1, char* cons2, uint32_t cap, uint8_t* is_cons1);
/* several genes at once: their consensus problems advance in lockstep on the GPU (the gene buckets are independent,
 * src/hla/caller.rs:642), so the whole sample costs as many launches as its longest gene.  cfgs / calls: n_genes entries;
 * cons: n_genes * 2 * cap bytes, consensus c of gene k at cons + (2k + c) * cap. */
int32_t sp_hla_diplotype_genes(sp_ctx* ctx, const sp_hla_db* db, uint32_t n_genes, const uint32_t* genes, const sp_seqset* reads,
                               const sp_hla_realign* realign, const sp_hla_call_config* cfgs, sp_hla_call* calls,
                               char* cons, uint32_t cap, uint8_t* is_cons1);
/* a cohort at once: `reads` holds the reads of n_samples samples (read_sample[r] = sample of read r; one sp_hla_realign_reads call
 * serves them all), and the consensus problems of every (sample, gene) advance in lockstep -- for WGS-sized samples a launch costs
 * the same whether it serves one sample or dozens.  cfgs: n_genes entries; calls: n_samples * n_genes, sample-major;
 * cons: n_samples * n_genes * 2 * cap bytes, laid out like calls. */
int32_t sp_hla_diplotype_cohort(sp_ctx* ctx, const sp_hla_db* db, uint32_t n_samples, const uint32_t* read_sample, uint32_t n_genes, const uint32_t* genes,
                                const sp_seqset* reads, const sp_hla_realign* realign, const sp_hla_call_config* cfgs, sp_hla_call* calls,
                                char* cons, uint32_t cap, uint8_t* is_cons1);

/* ------------------------------------------------------------------ host-side decisions of the path (no device work)
 * Small scalar routines the reference evaluates between the kernels; kept behind the same ABI so a host can drop the whole
 * path in.  statrs 0.16 formulas (Binomial::cdf / ln_pmf, Normal::ln_pdf, ln_factorial). */
/* is_passing_dual (src/hla/caller.rs:1225-1247): returns 1 when maf >= min_consensus_fraction and Binomial(expected_maf, n).cdf(minor) >= min_cdf */
int32_t sp_hla_is_passing_dual(uint64_t counts1, uint64_t counts2, double min_consensus_fraction, double expected_maf, double min_cdf,
                               double* maf_out, double* cdf_out);
/* is_hemizygous_better (src/hla/caller.rs:1583-1653): scores1/scores2 < 0 mean None; normalized_coverage < 0 means unknown */
int32_t sp_hla_is_hemizygous_better(const int64_t* scores1, const int64_t* scores2, const uint8_t* is_consensus1, uint32_t n_reads,
                                    int32_t is_dual, uint64_t dual_max_ed_delta, double normalized_coverage,
                                    double* haploid_cost, double* diploid_cost);
/* the coverage normalisation of diplotype_hla_batch (src/hla/caller.rs:598-617): reads realigned to the normalising genes (NORMALIZING_HLA_GENES,
 * src/hla/alleles.rs:49-59: HLA-DRB1) over two haplotypes for each of those genes that has reads; *normalized_coverage = -1 (None) when none has --
 * the value sp_hla_call_config.normalized_coverage takes for the absent-capable genes (DRB3 / 4 / 5) */
int32_t sp_hla_normalized_coverage(const sp_hla_realign* realign, uint32_t n_reads, const uint32_t* normalizing_genes, uint32_t n_normalizing,
                                   double* normalized_coverage);
/* hpc_pos / hpc_bytes (src/util/homopolymers.rs:18-42) */
uint64_t sp_hpc_pos(const char* seq, uint64_t len, uint64_t position);
uint64_t sp_hpc(const char* seq, uint64_t len, char* out);
/* convert_chain_to_hap (src/cyp2d6/caller.rs:907-957); detail 0 = core alleles, 1 = sub-alleles; returns the string length
 * (the output is truncated to cap-1 characters) */
uint32_t sp_cyp_chain_to_hap(const int32_t* chain, uint32_t n, const int32_t* hap_type, const char* const* hap_subtype,
                             uint32_t n_translate, const char* const* translate_key, const char* const* translate_val,
                             int32_t detail, char* out, uint32_t cap);

/* NormalizedVariant::new (src/data_types/normalized_variant.rs:43-170) with parse_sequence (:262-279): CPIC allele syntax
 * ("del", "ins..", "delins..", "XYZ(n)"), anchor base, suffix / prefix trimming, left shift along the reference.
 * chrom_seq = the contig (NULL = no reference genome: no anchoring / shifting, as in the reference); position is 0-based.
 * out_ref / out_alt hold cap bytes each.  SP_ERR_BAD_VARIANT for every condition on which the reference bails. */
int32_t sp_variant_normalize(const char* chrom_seq, uint64_t chrom_len, uint64_t position, const char* ref_allele, const char* alt_allele,
                             uint64_t* out_position, char* out_ref, char* out_alt, uint32_t cap);
/* NormalizedVariant::multi_new (:174-214): IUPAC codes and "; " lists expand to several alternatives; is_none[i] = 1 where the
 * alternative equals the reference allele (Option::None).  Outputs are strided by cap; returns the count in n_out. */
int32_t sp_variant_multi_normalize(const char* chrom_seq, uint64_t chrom_len, uint64_t position, const char* ref_allele, const char* alt_allele,
                                   uint32_t max_out, uint32_t* n_out, uint8_t* is_none, uint64_t* out_position, char* out_ref, char* out_alt, uint32_t cap);

/* chain building between K4 and K5 (src/cyp2d6/caller.rs:429-583): per read, the cartesian product of the minimum-edit
 * consensuses of its kept segments (:461-491); best_allele_mapping_counts (:478-481); reads without a chain are dropped
 * (:494-517); chains that use a consensus nothing maps to uniquely are removed (:521-538) and those consensuses are flagged
 * for mark_false_allele() (:574-583).  Segments of read r = [read_seg_off[r], read_seg_off[r+1]); ed / kept as written by
 * sp_cyp_weight_segments.  Outputs feed sp_chain_problem directly:
 *   read_index[k] (k < info->n_reads)  input index of the k-th recorded read (input order = BTreeMap order of the caller)
 *   read_chain_off[n_reads+1], chain_off[chain_cap+1], chain_items[item_cap]   obs_chains
 *   read_w_off[n_reads+1], w_seg[n_segments]   chain_scores rows as segment indices into ed / ov
 *   unique_counts[n_haps], false_allele[n_haps]
 * Returns SP_OK, SP_ERR_CAPACITY (info holds the sizes needed) or SP_ERR_CHAIN_COLLAPSE. */
typedef struct { uint32_t n_reads, n_chains, n_items, n_rows; } sp_chain_build_info;
int32_t sp_cyp_build_chains(uint32_t n_haps, const int32_t* hap_type, uint32_t n_reads, const uint32_t* read_seg_off,
                            const uint64_t* ed, const uint8_t* kept,
                            uint32_t* read_index, uint32_t* read_chain_off, uint32_t* chain_off, uint32_t chain_cap,
                            uint32_t* chain_items, uint32_t item_cap, uint32_t* read_w_off, uint32_t* w_seg,
                            uint64_t* unique_counts, uint8_t* false_allele, sp_chain_build_info* info);

/* result strings (src/data_types/pgx_diplotype.rs, src/data_types/region_variants.rs) */
/* Diplotype::diplotype ("h1/h2", :13-20) or, pharmcat != 0, Diplotype::pharmcat_diplotype (haplotypes containing '+' in brackets, :51-64);
 * returns the string length (the output is truncated to cap-1 characters) */
uint32_t sp_diplotype_string(const char* hap1, const char* hap2, int32_t pharmcat, char* out, uint32_t cap);
/* InexactHaplotype::new + full_haplotype (:138-196): variants = RegionVariant{label, is_vi, state}, taken in BTreeSet order
 * (label, is_vi, state; duplicates collapse); every variant that is not a Match is appended as " <sign><label>" with the sign of
 * RegionVariant's Display ('+' Unexpected, '-' Missing, '?' the ambiguous / unknown states, region_variants.rs:44-60) and the
 * whole is put in parentheses; match_type: sub-allele match when all match, core match when every VI variant matches. */
enum { SP_REL_UNKNOWN = 0, SP_REL_MATCH = 1, SP_REL_UNEXPECTED = 2, SP_REL_MISSING = 3, SP_REL_AMBIGUOUS_UNEXPECTED = 4,
       SP_REL_AMBIGUOUS_MISSING = 5, SP_REL_UNKNOWN_UNEXPECTED = 6, SP_REL_UNKNOWN_MISSING = 7 };     /* VariantAlleleRelationship */
enum { SP_INEXACT_UNKNOWN = 0, SP_INEXACT_NO_MATCH = 1, SP_INEXACT_CORE_MATCH = 2, SP_INEXACT_SUBALLELE_MATCH = 3 };   /* InexactMatchType */
uint32_t sp_inexact_haplotype(const char* base_haplotype, uint32_t n_variants, const char* const* labels, const uint8_t* is_vi,
                              const int32_t* states, int32_t* match_type, char* out, uint32_t cap);

/* ------------------------------------------------------------------ f3: the database file and the result file (host only)
 * sp_database_* replaces the serde loading of PgxDatabase (src/database/pgx_database.rs:23-41; load_json, src/util/file_io.rs:16-28
 * -- gzip is recognised by its magic number here, not by the ".gz" extension) and the three flattenings the kernels' inputs need:
 * hla_sequences + hla_config -> sp_hla_db_desc (src/hla/alleles.rs:332-344; src/hla/realigner.rs:42-91), cyp2d6_config +
 * cyp2d6_gene_def -> sp_cyp_locus / sp_cyp_gene_def / sp_cyp_config, and one gene entry (PgxGene / PgxVariant / PgxHaplotype,
 * src/database/pgx_database.rs:373-840) -> sp_variant_gene -> sp_variant_problem.  Files written by older releases load too: a missing
 * "cyp2d6_config" / "hla_config" gets the reference's defaults, "hla_config" is read in its gene_collection form and in the older
 * hla_coordinates / hla_exons / hla_is_forward_strand form, is_core_variant / is_core_haplotype default to true.
 * Every pointer handed out belongs to the object it came from and stays valid until that object is freed (flatten results: until the
 * next flatten call of the same kind).  Errors: SP_ERR_INVALID_ARG, with the text in err / sp_database_last_error. */
typedef struct sp_database sp_database;
int32_t sp_database_load(const char* path, sp_database** out, char* err, uint32_t err_cap);                 /* .json or .json.gz */
int32_t sp_database_parse(const char* text, uint64_t len, sp_database** out, char* err, uint32_t err_cap);  /* JSON text or gzip bytes */
void    sp_database_free(sp_database* db);
const char* sp_database_last_error(const sp_database* db);
typedef struct {                                  /* PgxMetadata (src/database/pgx_database.rs:358-371) */
    const char* pbstarphase_version; const char* cpic_version; const char* hla_version; const char* pharmvar_version; const char* build_time;
} sp_database_metadata;
int32_t sp_database_get_metadata(const sp_database* db, sp_database_metadata* out);
typedef struct {
    uint32_t n_gene_entries, n_hla_sequences, n_hla_genes, n_cyp2d6_alleles, n_collection_genes;
    int32_t has_hla_config, has_cyp2d6_config, reserved;
} sp_database_stats;
int32_t sp_database_info(const sp_database* db, sp_database_stats* out);
typedef struct {                                  /* GeneDefinition (src/database/gene_definition.rs:18-34); 0-based half-open */
    const char* name; const char* chrom; uint64_t start, end;
    int32_t is_forward_strand, is_absent_capable; uint32_t n_exons, reserved;
    const uint64_t* exon_start; const uint64_t* exon_end;
} sp_gene_region;
int32_t sp_database_hla_gene(const sp_database* db, uint32_t g, sp_gene_region* out);         /* g < n_hla_genes, gene-name order */
int32_t sp_database_gene_entry(const sp_database* db, uint32_t i, const char** gene_name, const char** chromosome);   /* i < n_gene_entries */
/* the HLA part as sp_hla_db_create wants it.  gene_names: the genes to type, NULL = every gene of hla_config; gene_ref[g] = the hg38
 * bases of [start - ref_buffer, end + ref_buffer) of gene g; alleles of other genes are left out, the others keep database-key order */
int32_t sp_database_hla_flatten(sp_database* db, uint32_t n_genes, const char* const* gene_names, const char* const* gene_ref, int32_t ref_buffer,
                                sp_hla_db_desc* desc);
/* allele i of the last sp_database_hla_flatten: "HLA:HLA00001", "HLA-A", "01:01:01:01" */
int32_t sp_database_hla_allele(const sp_database* db, uint32_t i, const char** hla_id, const char** gene_name, const char** star_allele);
/* the smallest chromosome window holding every CYP2D6 coordinate of the configuration */
int32_t sp_database_cyp_window(const sp_database* db, const char** chrom, uint64_t* start, uint64_t* end);
/* the CYP2D6 part as sp_cyp_db_create wants it; chrom_seq = the bases of [window_start, window_start + window_len) */
int32_t sp_database_cyp_flatten(sp_database* db, const char* chrom_seq, uint64_t window_start, uint64_t window_len,
                                sp_cyp_locus* locus, sp_cyp_gene_def* gene_def, sp_cyp_config* config);

/* One variant-typed gene: load_database_haplotypes (src/diplotyper.rs:437-538) -- every defined haplotype normalised against the
 * chromosome (chrom_seq may be NULL: no reference, as NormalizedVariant::new without a genome), haplotypes with a variant that does
 * not normalise are dropped (n_skipped_haplotypes), the variant table is the BTreeMap<NormalizedVariant, VariantMeta> in key order --
 * and the gene's structural-variant definitions (PgxStructuralVariants + the database's gene_collection) for is_deletion. */
typedef struct sp_variant_gene sp_variant_gene;
int32_t sp_variant_gene_create(sp_database* db, const char* gene_name, const char* chrom_seq, uint64_t chrom_len, sp_variant_gene** out);
void    sp_variant_gene_free(sp_variant_gene* gene);
typedef struct { uint32_t n_haplotypes, n_variants, n_skipped_haplotypes, n_full_deletions, n_partial_deletions, reserved; } sp_variant_gene_stats;
int32_t sp_variant_gene_info(const sp_variant_gene* gene, sp_variant_gene_stats* out);
int32_t sp_variant_gene_haplotype(const sp_variant_gene* gene, uint32_t h, const char** name, const char** core_allele /* NULL: is one */);
int32_t sp_variant_gene_variant(const sp_variant_gene* gene, uint32_t v, uint64_t* position, const char** ref, const char** alt,
                                const char** name, const char** dbsnp_id /* may come back NULL */, int64_t* variant_id, int32_t* is_core);
int32_t sp_variant_gene_sv_definitions(const sp_variant_gene* gene, sp_sv_definitions* out);      /* for sp_variant_is_deletion */
int32_t sp_variant_gene_sv_label(const sp_variant_gene* gene, int32_t kind, int32_t index, const char** label);
/* load_vcf_variants + load_sv_vcf_variants (src/diplotyper.rs:551-857) after the file decoding: the caller hands over the decoded
 * records -- one sp_vcf_allele per ALT allele of a small-variant record (0-based position, REF, that ALT; gt = SP_GT_* of THIS allele
 * in the sample's GT, SP_GT_HOM_REF when the allele is not called; ps = phase set or -1), one sp_vcf_deletion per SVTYPE=DEL record
 * (0-based start, INFO/END).  The library normalises, matches the alleles to the gene's variants (+-50 bp window, the last matching
 * record wins), decides the deletions with is_deletion (gene span, max_sv_length; 0 = 1,000,000), and fills *problem for
 * sp_variant_solve.  Variant ids of the problem: the gene's variants and the observed deletions merged in NormalizedVariant order;
 * sp_variant_gene_problem_variant maps an id back (db_variant = index for sp_variant_gene_variant, or -1 and the deletion label).
 * Errors (SP_ERR_INVALID_ARG): a homozygous allele with a phase set, two records for the same deletion, a deleted gene without a
 * definition in the gene collection. */
typedef struct { uint64_t position; const char* ref; const char* alt; int32_t gt, reserved; int64_t ps; } sp_vcf_allele;
typedef struct { uint64_t start, end; int32_t gt, reserved; int64_t ps; } sp_vcf_deletion;
int32_t sp_variant_gene_problem(sp_variant_gene* gene, uint32_t n_alleles, const sp_vcf_allele* alleles, uint32_t n_deletions,
                                const sp_vcf_deletion* deletions, uint64_t max_sv_length, sp_variant_problem* problem);
int32_t sp_variant_gene_problem_variant(const sp_variant_gene* gene, int32_t id, int32_t* db_variant, const char** sv_label, uint64_t* sv_start,
                                        uint64_t* sv_end);
int32_t sp_variant_gene_problem_sv_label(const sp_variant_gene* gene, int32_t label_id, const char** label);   /* obs_sv_label ids */
const char* sp_variant_gene_last_error(const sp_variant_gene* gene);

/* The result file: StarphaseJson / PgxGeneDetails (src/data_types/starphase_json.rs:11-326) written as serde_json::to_writer_pretty
 * does (save_json, src/util/file_io.rs:37-52; two-space indent, struct fields in declaration order, gene_details in key order,
 * Option::None as null).  A sp_gene_details collects the parts; sp_result_insert applies one of the reference's constructors to it --
 * with that constructor's checks ("diplotypes and simple_diplotypes must be the same length", "diplotypes and inexact_diplotypes
 * must be the same length") and StarphaseJson::insert's ("Entry for <gene> is already occupied.") as SP_ERR_INVALID_ARG. */
typedef struct sp_result sp_result;
typedef struct sp_gene_details sp_gene_details;
int32_t sp_result_create(const sp_database* db /* NULL: PgxMetadata::default() */, const char* pbstarphase_version, sp_result** out);
void    sp_result_free(sp_result* result);
const char* sp_result_last_error(const sp_result* result);
int32_t sp_gene_details_create(sp_gene_details** out);
void    sp_gene_details_free(sp_gene_details* details);
int32_t sp_gene_details_add_diplotype(sp_gene_details* d, const char* hap1, const char* hap2);             /* Diplotype::new */
int32_t sp_gene_details_add_simple_diplotype(sp_gene_details* d, const char* hap1, const char* hap2);      /* simple_diplotypes: Some(..) */
int32_t sp_gene_details_set_simple_diplotypes(sp_gene_details* d, int32_t some);                           /* Some(vec![]) / None */
/* InexactDiplotype::new(InexactHaplotype::new(base, variants), ..): variants as for sp_inexact_haplotype */
int32_t sp_gene_details_add_inexact_diplotype(sp_gene_details* d,
                                              const char* base1, uint32_t n1, const char* const* labels1, const uint8_t* is_vi1, const int32_t* states1,
                                              const char* base2, uint32_t n2, const char* const* labels2, const uint8_t* is_vi2, const int32_t* states2);
int32_t sp_gene_details_add_diplotype_only(sp_gene_details* d, const char* hap1, const char* hap2);        /* InexactDiplotype::new_diplotype_only */
typedef struct {                                  /* PgxVariantDetails (:247-262); sv_label NULL = not a structural variant */
    uint64_t variant_id; const char* variant_name; const char* dbsnp;
    const char* chrom; uint64_t position; const char* reference; const char* alternate;
    const char* sv_label; uint64_t sv_start, sv_end;
    int32_t genotype /* SP_GT_* */, is_core_variant; int64_t phase_set /* -1 = None */;
} sp_variant_detail;
int32_t sp_gene_details_add_variant(sp_gene_details* d, const sp_variant_detail* v);
typedef struct { int32_t present, has_clips; uint64_t seq_len, nm, unmapped, clipped_start, clipped_end; } sp_mapping_stats;   /* Option<MappingStats> */
int32_t sp_gene_details_add_mapping(sp_gene_details* d, const char* read_qname, const char* best_hla_id, const char* best_star_allele,
                                    const sp_mapping_stats* cdna, const sp_mapping_stats* dna, int32_t is_ignored);       /* PgxMappingDetails */
int32_t sp_gene_details_add_multi_mapping(sp_gene_details* d, const char* read_qname, uint64_t read_start, uint64_t read_end,
                                          uint64_t consensus_id, const char* consensus_star_allele);                      /* PgxMultiMappingDetails */
enum { SP_DETAILS_SUBALLELE_MATCH = 0, SP_DETAILS_CORE_MATCH = 1, SP_DETAILS_INEXACT_DIPLOTYPES = 2, SP_DETAILS_FROM_MAPPINGS = 3,
       SP_DETAILS_FROM_MULTI_MAPPINGS = 4, SP_DETAILS_NO_MATCH = 5 };
int32_t sp_result_insert(sp_result* result, const char* gene, const sp_gene_details* details, int32_t constructor);
int32_t sp_result_json(sp_result* result, const char** text, uint64_t* len);
int32_t sp_result_save(sp_result* result, const char* path);                    /* gzip when the name ends in ".gz" */
/* save_pharmcat_tsv (src/main.rs:190-241): "#gene<TAB>diplotype" and one row per gene in key order -- the de-duplicated simple
 * diplotypes of the gene (PgxGeneDetails::dedup_simple_diplotypes: equal up to the order of the two haplotypes) give "Multiple/Multiple"
 * when more than one is left, else Diplotype::pharmcat_diplotype (a haplotype containing '+' in brackets); MT-RNR1 is written as its
 * single haplotype, "Unknown" when the two differ. */
int32_t sp_result_pharmcat_tsv(sp_result* result, const char** text, uint64_t* len);
int32_t sp_result_save_pharmcat_tsv(sp_result* result, const char* path);

/* ------------------------------------------------------------------ f4: the debug files (host only)
 * `hla_debug.json` (HlaDebug, src/hla/debug.rs:7-221; written by diplotype_hla_batch when a debug folder is given,
 * src/hla/caller.rs:1042-1048): per gene and read (or "consensus1" / "consensus2") the best match and the detailed mappings against the
 * alleles compared, and per gene the DualPassingStats of is_passing_dual (src/hla/caller.rs:1225-1247).  Errors as the reference's:
 * "Entry <x> is already occupied" (SP_ERR_INVALID_ARG, text from sp_hla_debug_last_error).
 * sp_aln_strings turns an alignment of this library (sp_align_batch with events; A = query, B = target) into what
 * DetailedMappingStats::from_mapping copies from minimap2 (:148-172): the CIGAR string (M / I / D, no --eqx), the MD tag and
 * match_len (matching bases); query_unmapped / target_unmapped are a_len - (a_end - a_start) and b_len - (b_end - b_start).
 * `cyp2d6_alleles.json` is written by sp_cyp_alleles_json above. */
typedef struct { int32_t present, reserved; uint64_t query_len, target_len, match_len, nm, query_unmapped, target_unmapped;
                 const char* cigar; const char* md; } sp_detailed_mapping;                    /* Option<DetailedMappingStats> */
int32_t sp_aln_strings(const sp_aln* aln, const uint32_t* events, const char* target, uint64_t target_len,
                       char* cigar, uint32_t cigar_cap, char* md, uint32_t md_cap, uint64_t* match_len);
typedef struct sp_hla_debug sp_hla_debug;
int32_t sp_hla_debug_create(sp_hla_debug** out);
void    sp_hla_debug_free(sp_hla_debug* debug);
const char* sp_hla_debug_last_error(const sp_hla_debug* debug);
int32_t sp_hla_debug_add_read(sp_hla_debug* debug, const char* gene, const char* qname,
                              const char* best_match_id /* NULL: None */, const char* best_match_star);
int32_t sp_hla_debug_add_mapping(sp_hla_debug* debug, const char* gene, const char* qname, const char* hla_id,
                                 const sp_detailed_mapping* cdna, const sp_detailed_mapping* dna);
int32_t sp_hla_debug_add_dual_stats(sp_hla_debug* debug, const char* gene, const sp_hla_call* call);
int32_t sp_hla_debug_json(sp_hla_debug* debug, const char** text, uint64_t* len);
int32_t sp_hla_debug_save(sp_hla_debug* debug, const char* path);             /* gzip when the name ends in ".gz" */

/* ------------------------------------------------------------------ e: gathering the ranks' results (the path's one exchange step)
 * The reference has no counterpart: it is single-threaded and a cohort is N process runs (src/cli/diplotype.rs:185-191).  BASELINE.json's
 * north_star shards independent units (samples, genes) over the GPUs of a node, one process and one sp_ctx per GPU, with "RCCL over xGMI used
 * only to gather per-gene results" (SURVEY.md 8(b) sp_gather_results, 8(e)).  Rank 0 makes a 128-byte id (ncclGetUniqueId) and the host hands it
 * to the other ranks over whatever channel it has (a file, a socket, MPI, a torch store); every rank then joins the group on its context's device.
 * sp_gather_results: every rank contributes bytes_per_rank bytes of packed records (host memory) and receives all ranks' records in rank order
 * (n_ranks * bytes_per_rank bytes) -- one ncclAllGather on the context's stream, synchronous on return.  librccl is opened at run time; without it
 * sp_group_* fail with SP_ERR_NO_DEVICE and nothing else in the library is affected. */
typedef struct sp_group sp_group;
#define SP_GROUP_ID_BYTES 128
int32_t sp_group_unique_id(uint8_t* id /* SP_GROUP_ID_BYTES */);
int32_t sp_group_create(sp_ctx* ctx, const uint8_t* id, int32_t rank, int32_t n_ranks, sp_group** out);
void    sp_group_free(sp_group* group);
int32_t sp_group_size(const sp_group* group, int32_t* rank, int32_t* n_ranks);
int32_t sp_gather_results(sp_group* group, const void* records, uint64_t bytes_per_rank, void* all_records);

/* ------------------------------------------------------------------ f2: decoding the input files (host only)
 * What the reference gets from rust-htslib: the records of an indexed BAM that overlap a region (diplotype_hla_batch,
 * src/hla/caller.rs:523-596; the CYP2D6 read collection, src/cyp2d6/caller.rs:96-139) and the records of a VCF around a position
 * (load_vcf_variants / load_sv_vcf_variants, src/diplotyper.rs:551-857).  BGZF is read with zlib; a BAM is read through its .bai when
 * "<path>.bai" (or "<path minus .bam>.bai") exists and by a linear scan otherwise; a VCF (plain, gzip or bgzip) is read into memory
 * once -- no tabix index is needed for files of one sample's PGx-gene size, and none is used.  CRAM is not read.
 * Pointers handed out belong to the reader and stay valid until its next fetch call (or its free). */
typedef struct sp_bam sp_bam;
int32_t sp_bam_open(const char* path, sp_bam** out, char* err, uint32_t err_cap);
void    sp_bam_free(sp_bam* bam);
const char* sp_bam_last_error(const sp_bam* bam);
int32_t sp_bam_references(const sp_bam* bam, uint32_t* n, const char* const** names, const uint64_t** lengths);   /* @SQ of the header */
typedef struct {                                  /* one alignment record */
    const char* qname; uint32_t flag, mapq; int32_t ref_id, reserved; int64_t pos, end;   /* 0-based [pos, end) on the reference (end from the CIGAR) */
    uint32_t l_seq, n_cigar; const uint32_t* cigar;                                         /* BAM encoding: len << 4 | op */
} sp_bam_read;
/* the records overlapping [start, end) of chrom in file order; exclude_flags drops records with any of these FLAG bits; dedupe != 0
 * drops a record whose QNAME was handed out since sp_bam_open / sp_bam_forget (the reference's qnames_checked set, :532,565-570).
 * bases / offsets: the SEQ fields as stored (reference-forward ASCII, "=ACMGRSVTWYHKDBN"), concatenated, n + 1 offsets -- the
 * arguments of sp_seqset_upload. */
int32_t sp_bam_fetch(sp_bam* bam, const char* chrom, uint64_t start, uint64_t end, uint32_t exclude_flags, int32_t dedupe,
                     const sp_bam_read** reads, uint32_t* n, const char** bases, const uint64_t** offsets);
int32_t sp_bam_forget(sp_bam* bam);               /* empties the set of QNAMEs seen */
/* the SEQ fields of the records of the last fetch exactly as the file stores them (4 bits per base, high nibble first, every read on a
 * byte boundary): the arguments of sp_seqset_upload_format(ctx, SP_SEQ_BAM4, ...) -- half of what the ASCII form sends over PCIe */
int32_t sp_bam_last_seq4(const sp_bam* bam, const uint8_t** seq4, const uint64_t** byte_offsets, const uint32_t** lengths, uint32_t* n);

typedef struct sp_vcf sp_vcf;
int32_t sp_vcf_open(const char* path, sp_vcf** out, char* err, uint32_t err_cap);
void    sp_vcf_free(sp_vcf* vcf);
/* A BGZF-compressed VCF with a tabix (<path>.tbi) or CSI (<path>.csi) index beside it is opened by its header alone; every query below then reads only the chunks
 * the index names for its region (bcf::IndexedReader::fetch, src/diplotyper.rs:569-575,800) -- a 5 M-record WGS VCF costs a query a few blocks, not a scan.  Without
 * an index (or with one that cannot be read) the file is read once as a whole.  indexed: which of the two; lines_parsed_by_fetches: record lines the queries so far
 * have parsed (either may be NULL). */
int32_t sp_vcf_index_info(const sp_vcf* vcf, int32_t* indexed, uint64_t* lines_parsed_by_fetches);
const char* sp_vcf_last_error(const sp_vcf* vcf);
int32_t sp_vcf_samples(const sp_vcf* vcf, uint32_t* n, const char* const** names);
/* small variants: one sp_vcf_allele per ALT allele of every record of chrom that overlaps [start, end), for one sample (NULL = the
 * first); gt is the state of THAT allele in the sample's GT -- SP_GT_HOM_ALT a/a, SP_GT_HET_FLIP a|x, SP_GT_HET_PHASED x|a,
 * SP_GT_HET_UNPHASED a/x or x/a, SP_GT_HOM_REF when the allele is not called; ps = FORMAT/PS of a phased GT or -1.  Records whose GT
 * is missing or not diploid are skipped (src/diplotyper.rs:606-640).  The rows are the input of sp_variant_gene_problem. */
int32_t sp_vcf_alleles(sp_vcf* vcf, const char* sample, const char* chrom, uint64_t start, uint64_t end, const sp_vcf_allele** out, uint32_t* n);
/* structural variants (src/diplotyper.rs:739-857): single-ALT records of chrom with INFO/SVTYPE=DEL that overlap [start, end); the
 * deleted region is [POS - 1, INFO/END).  SP_ERR_INVALID_ARG where the reference bails: a record without SVTYPE or a DEL without END. */
int32_t sp_vcf_deletions(sp_vcf* vcf, const char* sample, const char* chrom, uint64_t start, uint64_t end, const sp_vcf_deletion** out, uint32_t* n);

/* ------------------------------------------------------------------ profiling hooks (bench.py)
 * HIP-event timing of the dominant kernel on the context's own stream. */
int32_t sp_profile_reset(sp_ctx* ctx);
int32_t sp_profile_get(sp_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches, uint64_t* cells);
/* device-side counters since the last sp_profile_reset come back in *cells under the names "count:k1_cells_active" (cells of the
 * first K1 pass whose gene the read anchors in), "count:k1_cells_executed" (those that ran the DP; the others were settled by prefix
 * sharing), "count:k1_cells_resumed" (executed cells that started from their predecessor's snapshot), "count:k1_cells_bytes"
 * (algorithmic bytes of the executed cells), "count:cons_launches", "count:cons_columns" (K8). */
/* roofline peaks measured on this device in this run: what = "valu_int" (v_add_u32 wave-instructions / s), "match16" (VALU
 * wave-instructions / s of the WFA cell's 16-base compare), "hbm_copy" (bytes / s, read + written, of a 2 x 1 GiB streaming copy) */
int32_t sp_microbench(sp_ctx* ctx, const char* what, double* rate);

/* The reference FASTA (ReferenceGenome::from_fasta / get_slice of rust-lib-reference-genome, loaded once in src/cli/diplotype.rs): the
 * slices sp_database_hla_gene / sp_database_cyp_window / sp_variant_gene_create ask for.  Plain files are read through "<path>.fai"
 * when it exists (only the requested bases are read), otherwise -- and for gzip / BGZF files -- the file is read into memory once.
 * Coordinates are 0-based half-open; bases come back upper-cased and stay valid until the next fetch on the handle. */
typedef struct sp_fasta sp_fasta;
int32_t sp_fasta_open(const char* path, sp_fasta** out, char* err, uint32_t err_cap);
void    sp_fasta_free(sp_fasta* fasta);
const char* sp_fasta_last_error(const sp_fasta* fasta);
int32_t sp_fasta_sequences(sp_fasta* fasta, uint32_t* n, const char* const** names, const uint64_t** lengths);
int32_t sp_fasta_fetch(sp_fasta* fasta, const char* chrom, uint64_t start, uint64_t end, const char** bases, uint64_t* len);

#ifdef __cplusplus
}
#endif
#endif
