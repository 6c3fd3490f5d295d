// sp_internal.h -- shared host/device declarations of libstarphase_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <chrono>
#include <set>
#include <mutex>
#include <map>
#include "../../include/starphase_hip.h"

#define SP_NEG        (-(1 << 28))
#define SP_SEQ_PAD_WORDS 512     // zero words behind the last sequence of a set: fixed-shape prefetches (k1_cells) may read past a sequence
#define SP_LDS_TAIL    288       // bytes behind the last LDS window: the cooperative extension of wfa_core reads up to 65 words past a stretch
#define SP_WAVE       64
#define SP_MAXOCC     4
#define SP_CELL_NONE  0xFFFFFFFFu
#define SP_PEAK_SPREAD   48      // anchor = midpoint of the strongly voted diagonals within +-48 of the peak
#define SP_PEAK_SUPPRESS 128     // top-K anchors: bins within +-128 diagonals of a chosen peak are cleared

// ---------------------------------------------------------------- device views
// Packed sequence set in HBM.  2 bits/base, 16 bases per 32-bit word, base b of a sequence sits in bits
// [2*(b&15), 2*(b&15)+1] of word (b>>4).  Every sequence starts on a 16-byte boundary (4 words) so a wavefront
// stages it with coalesced dword loads, and is followed by >= 2 zero guard words.  nplane has the same
// layout with 01 at every non-ACGT base (only allocated when the set contains such bases).
struct SeqSetView {
    const uint32_t* words;
    const uint32_t* nplane;      // nullptr when the set has no N
    const uint64_t* word_off;    // n+1
    const int32_t*  len;         // n
    uint32_t n;
};

// sorted k-mer table of the sequences of an indexed set (the "A" side of sp_anchor_batch)
struct KmerIndexView {
    const uint32_t* code;        // sorted per sequence
    const int32_t*  pos;
    const uint64_t* off;         // n+1 entry offsets
};

struct sp_seqset {
    sp_ctx* ctx = nullptr;
    uint32_t n = 0;
    bool has_n = false;
    std::vector<int32_t>  h_len;
    std::vector<uint64_t> h_word_off;
    std::vector<uint32_t> h_words;     // kept for index building / views
    uint32_t* d_words = nullptr;
    uint32_t* d_nplane = nullptr;
    uint64_t* d_word_off = nullptr;
    int32_t*  d_len = nullptr;
    int32_t   max_len = 0;
    uint32_t  n_skipped = 0;           // sequences longer than 65,534 bases: kept as empty entries
    struct sp_upload* up = nullptr;    // an upload that is still under way (sp_seqset_upload_async); sp_seqset_wait ends it
    int32_t up_rc = 0; std::string up_err;   // how it ended
    // lazily built k-mer index
    bool has_index = false;
    uint32_t* d_kcode = nullptr; int32_t* d_kpos = nullptr; uint64_t* d_koff = nullptr;
    SeqSetView view() const { return SeqSetView{d_words, d_nplane, d_word_off, d_len, n}; }
    KmerIndexView kview() const { return KmerIndexView{d_kcode, d_kpos, d_koff}; }
};

struct ProfileEntry { double ms = 0; uint64_t launches = 0; uint64_t cells = 0; };

struct sp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    bool profiling = true;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, ProfileEntry> prof;
    // timed regions whose events have been recorded but not read yet: reading them is put off until somebody asks for the numbers
    // (or the list gets long), so that timing a kernel does not make the host wait for it
    struct PendingProf { hipEvent_t e0, e1; const char* name; uint64_t cells; };
    std::vector<PendingProf> prof_pending;
    std::vector<hipEvent_t> prof_free;
    // reusable scratch
    void* scratch = nullptr; size_t scratch_bytes = 0;
    std::map<std::string, std::pair<void*, size_t>> pool;   // named grow-only device buffers (no malloc/free per call)
    std::map<std::string, std::pair<void*, size_t>> host_pool;   // named grow-only pinned host buffers (results leave the device through them)
    // device buffers of read sets that were freed, kept for the next upload (sp_dev_alloc / sp_dev_release, sp_api.hip): hipFree waits for EVERY stream of the device --
    // a lane that closed its sample's read set stood still until the other lane's kernels had ended, 1.2 ms per sample -- and an upload began with three hipMalloc
    std::multimap<size_t, void*> dev_cache; std::map<void*, size_t> dev_cap; size_t dev_cache_bytes = 0; std::mutex dev_cache_mu;
    std::set<sp_seqset*> live_sets;   // the read sets that will hand their buffers back (a context destroyed before them lets go of them: they free their buffers themselves)
    int num_cus = 256;
    int hw_queues = 4; bool hw_queues_by_library = false; std::string warning;     // sp_ctx_get_info
    bool split_genes = true;         // sp_ctx_set_option "hla_split_genes"
    int split_streams = 3;           // sp_ctx_set_option "hla_split_streams": streams the units of a call are spread over (1..4; a 32-sample cohort call: 71.5 / 58.4 / 54.9 / 68.6 ms)
    bool cons_retry_ladder = false;  // sp_ctx_set_option "cons_retry_ladder": sp_consensus_priority's retry of searches that give up (the drivers pass it on)
    hipStream_t copy_stream = nullptr;   // uploads travel on a stream of their own, beside the kernels of ctx->stream
    hipStream_t ctl_stream = nullptr;    // the control workgroups of a persistent consensus batch run here, beside the step workgroups on ctx->stream (made on first use)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int k1_best_n = 5;                   // sp_ctx_set_option "k1_best_n": K1 base-aligns the chains minimap2's seeding and chaining select (best_n secondaries per read, the reference's 5);
                                         // 0 = every allele of every anchored gene (the exhaustive argmin of rounds 1-4)
    int mm2_rescore = 1;                 // sp_ctx_set_option "mm2_rescore": the entry points that return mappings also report them re-scored with the reference's affine scores (mm2_* fields)
    int k8_persistent = 2;               // sp_ctx_set_option "k8_persistent" (or SP_K8_PERSISTENT): small consensus batches as two persistent kernels instead of a launch pair per step.
                                         // 0 never, 1 whenever a batch fits, 2 (default) the library decides: a single sample's batches (<= 8 problems) when the process's streams
                                         // have hardware queues of their own (hw_queues_effective >= 16) and the mode has not just failed on this context
    int hw_queues_effective = 0;         // what the HIP runtime was initialised with: GPU_MAX_HW_QUEUES as it stood when HIP came up (0: HIP was up before the library could look, and the variable was not set)
    int k8_side_orders = 1;              // sp_ctx_set_option "k8_side_orders" (or SP_K8_SIDE_ORDERS): work orders per consensus problem and step beside the search's own (0 .. 3): the window or
                                         // expansion another waiting node will need at its turn, made in the same launch.  Results do not depend on it.  One:
                                         // a second and third row of workgroups save 5 % more launches and cost every launch more than that (profiles/r06/k8_side_orders.txt)
    int k8_compound = 1;                 // sp_ctx_set_option "k8_compound" (or SP_K8_COMPOUND): a consensus window may be ordered with the children of the branch foreseen at its end
    int k8_side_max_blocks = 4096;       // batches with more step workgroups than this keep to the search's own order
    int k8_persist_backoff = 0;          // batches that still go the launch-pair way after the control workgroups of a persistent batch found no CUs; k8_persist_failures counts those events
    int k8_persist_failures = 0;
    sp_seqset* uploading = nullptr;      // the one upload a context has in flight (the staging buffers are the context's)
    int k5_block_pairs = 4096;       // sp_ctx_set_option "k5_block_pairs": up to this many chain pairs K5 runs one workgroup per pair (0: always one thread per pair)
    int cyp_cohort_min_group = 12;   // sp_ctx_set_option "cyp_cohort_min_group": a stream is only added when it leaves this many samples per stream
    int cyp_cohort_streams = 8;      // sp_ctx_set_option "cyp_cohort_streams": samples of sp_cyp_diplotype_cohort in flight (1..8; WGS-sized samples are chains of tiny launches)
    sp_ctx* helper[7] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };   // further contexts on the same device (own stream, pools, events) for work that runs beside this one's; made on first use
};

// cell descriptor consumed by the generic WFA kernel
struct CellDesc {
    uint32_t a, b;
    int32_t  diag;          // b_pos - a_pos (in full-sequence coordinates of a and b views)
    int32_t  max_ed;
    int32_t  b_lo, b_hi;    // sub-range of B that is the "real" sequence for this cell (segment views); b_hi<0 => whole
};

// ---------------------------------------------------------------- 16-mer dictionary of an allele set (sp_hla_dict.hip)
struct K2Dict {
    bool built = false, failed = false;
    uint32_t* d_ids = nullptr;              // every allele, position by position: number of its 16-mer in its gene's part of the dictionary (0xFFFFFFFF: holds an N)
    uint64_t* d_id_off = nullptr;           // n_alleles + 1: first id slot of each allele
    uint32_t* d_code = nullptr;             // the dictionary: the distinct 16-mers of each gene in the order of their first appearance
    uint32_t* d_dict_off = nullptr;         // n_genes + 1: each gene's part of the dictionary
    std::vector<uint32_t> dict_off;
    uint32_t n_dict = 0, max_dict = 0;
};
void* sp_dev_alloc(sp_ctx* ctx, size_t bytes);    // a device buffer of >= bytes: one a freed read set left behind when one fits, hipMalloc otherwise; nullptr: out of memory
void sp_dev_release(sp_ctx* ctx, void* p);        // hands it back (nullptr is fine); buffers beyond the cache's bounds are freed
int  sp_k2_dict_build(sp_ctx* ctx, const sp_seqset* set, const uint32_t* d_gene_of, uint32_t n_genes, K2Dict* out);
void sp_k2_dict_free(K2Dict* d);

// ---------------------------------------------------------------- K1 in the reference's call pattern: minimizer index, chains, best_n (sp_hla_seed.hip)
struct K1Seed;
struct K1SeedDebug {                        // the chain list, selection and mappings of ONE read of the batch (sp_hla_realign_seeded_audit)
    uint32_t read;
    int32_t* chains; uint32_t chain_cap; uint32_t* n_chains;      // [cap][10] = {rid, rev, score, seeds, qs, qe, rs, re, parent, selected} in rank order
    sp_k1_seed_hit* hits; uint32_t* n_hits;                       // SP_K1_SEL entries, output order
    uint64_t* counters;                                           // 4: kept seeds, anchors, largest anchor count of a read, reads that hit a capacity
};
int  sp_k1_seed_build(sp_ctx* ctx, const sp_seqset* alleles, K1Seed** out);
void sp_k1_seed_free(K1Seed* s);
void sp_k1_seed_stats(const K1Seed* s, int64_t out[4]);            // minimizers, distinct minimizers, mid_occ, indexed sequences
int  sp_k1_seed_sketch(sp_ctx* ctx, const sp_seqset* set, uint32_t idx, uint64_t* hash, int32_t* end_pos, uint8_t* strand, uint32_t cap, uint32_t* n_out);
int  sp_k1_seed_map(sp_ctx* ctx, const K1Seed* idx, const sp_seqset* alleles, const sp_seqset* reads, int best_n, int32_t* d_best, sp_k1_seed_info* d_info,
                    sp_aln* d_win_aln, sp_affine_aln* d_win_af, const K1SeedDebug* dbg);

// ---------------------------------------------------------------- launchers (sp_device.hip)
sp_ctx* sp_ctx_helper(sp_ctx* ctx, int i = 0);                           // helper i (0..6); nullptr when it cannot be made
void sp_profile_merge(sp_ctx* into, sp_ctx* from);                       // adds the timings `from` collected to `into` and clears them
int sp_launch_anchor(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                     const uint32_t* d_a_idx, const uint32_t* d_b_idx, uint64_t n_pairs,
                     int32_t* d_diag, int32_t* d_votes, int topk = 1, const char* prof_name = "anchor", uint32_t a_period = 1);
                     // a_period: the pair list repeats its A indices with this period (pair p has A = p % a_period): the grid is made a multiple of it, so that a
                     // workgroup striding over the list keeps ONE table of A in LDS instead of loading another for every pair
int sp_launch_cells(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                    const CellDesc* d_cells, uint64_t n_cells,
                    sp_aln* d_out, uint32_t* d_events, uint32_t events_stride, const char* prof_name, int retry_wide = 0);   // 0 never, 1 lost cells, 2 lost cells and cells with > 32 edits (few-cell callers)

int sp_launch_affine(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, const void* d_pairs /* sp_pair rows */, uint64_t n_pairs, const sp_affine_opts& o, int band,
                     sp_affine_aln* d_out, const char* prof_name, const uint32_t* d_n_live = nullptr, const void* d_wins = nullptr, const void* d_mids = nullptr);   // sp_affine.hip: two-piece affine re-score, pairs and results in device memory
int sp_rescore_mappings(sp_ctx* ctx, const sp_seqset* Aw, const sp_seqset* Bw, const CellDesc* d_cells, const sp_aln* d_ref, uint64_t n, bool target_is_a,
                        const sp_affine_opts& o, int band, sp_affine_aln* d_out, const char* prefix, uint32_t stride, int trace_retry_wide = 0, const sp_aln* d_tr_in = nullptr, const uint32_t* d_ev_in = nullptr, int ends_only = 0);   // sp_affine.hip: mappings the caller holds, re-scored (no DP for isolated edits); trace_retry_wide: the trace of a mapping the 64-diagonal cell loses runs on the wide band; ends_only > 0: the caller takes the EXTENT of the mapping only -- edits that far from both ends of the alignment never take the DP (af_classify_kernel)
constexpr uint32_t SP_ANCHOR_SKIP = 0xFFFFFFFFu;     // b index of a pair of a device-made pair list that sp_launch_anchor is to skip (votes 0, diagonal 0)
int sp_launch_cells_wide(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, const CellDesc* d_cells, uint64_t n_cells, sp_aln* d_out);
int sp_launch_pack_on(hipStream_t stream, int num_cus, int format, const void* d_src, const uint64_t* d_off, const uint64_t* d_word_off, const int32_t* d_len, uint32_t n,
                      uint32_t* d_words, uint32_t* d_nplane, uint32_t* d_flag);   // format: SP_SEQ_ASCII / SP_SEQ_BAM4 / SP_SEQ_PACKED2
int sp_launch_pack(sp_ctx* ctx, const char* d_ascii, const uint64_t* d_off, const uint64_t* d_word_off, const int32_t* d_len, uint32_t n,
                   uint32_t* d_words, uint32_t* d_nplane, uint32_t* d_flag);
int sp_make_segments(sp_ctx* ctx, const sp_seqset* reads, const std::vector<uint32_t>& idx, const std::vector<int32_t>& start,
                     const std::vector<int32_t>& len, const char* prefix, sp_seqset* seg, sp_seqset* hpc);   // sp_hla_call.hip
int sp_seqset_make_small(sp_ctx* ctx, const char* prefix, const char* bases, const uint64_t* offsets, uint32_t n, bool with_index, sp_seqset* out);   // pooled, never freed
int sp_seqset_fetch_host(sp_ctx* ctx, sp_seqset* s);                      // packed words (and N plane) of a set on the host, fetched once
std::string sp_seqset_decode(sp_ctx* ctx, const sp_seqset* s, uint32_t i); // ASCII of sequence i
// device-side measurement counters of a context (zeroed by sp_profile_reset, read by sp_profile_get under the names below)
enum { SPC_K1_ACTIVE = 0, SPC_K1_EXECUTED, SPC_K1_RESUMED, SPC_K1_BYTES, SPC_CONS_LAUNCHES, SPC_CONS_COLUMNS, SPC_N = 16 };
unsigned long long* sp_counters(sp_ctx* ctx);
void* sp_scratch(sp_ctx* ctx, size_t bytes);
void* sp_pool(sp_ctx* ctx, const char* name, size_t bytes);
void* sp_host_pool(sp_ctx* ctx, const char* name, size_t bytes);
int   sp_fail(sp_ctx* ctx, int code, const std::string& msg);
#define SP_HIP_CHECK(ctx, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
    return sp_fail((ctx), SP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

void sp_profile_flush(sp_ctx* ctx);                                           // sp_api.hip
// HostMarks: the same, stretch after stretch: mark(name) charges the time since the previous mark to `name`
struct HostMarks {
    sp_ctx* ctx; std::chrono::steady_clock::time_point t;
    explicit HostMarks(sp_ctx* c) : ctx(c), t(std::chrono::steady_clock::now()) {}
    void mark(const char* name) { const auto now = std::chrono::steady_clock::now(); if (ctx->profiling) { auto& e = ctx->prof[name]; e.ms += std::chrono::duration<double, std::milli>(now - t).count(); e.launches += 1; } t = now; }
};
// wall-clock time of a stretch of HOST code (sp_profile_get("host:...")): where a step waits for the CPU
struct HostScope {
    sp_ctx* ctx; const char* name; std::chrono::steady_clock::time_point t0;
    HostScope(sp_ctx* c, const char* n) : ctx(c), name(n), t0(std::chrono::steady_clock::now()) {}
    ~HostScope() { if (ctx->profiling) { auto& e = ctx->prof[name]; e.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); e.launches += 1; } }
};
struct ProfScope {
    sp_ctx* ctx; const char* name; uint64_t cells; hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(sp_ctx* c, const char* n, uint64_t cells_) : ctx(c), name(n), cells(cells_) {
        if (!(ctx->profiling && name)) return;
        auto take = [&]() { hipEvent_t e = nullptr; if (!ctx->prof_free.empty()) { e = ctx->prof_free.back(); ctx->prof_free.pop_back(); } else if (hipEventCreate(&e) != hipSuccess) e = nullptr; return e; };
        e0 = take(); e1 = take();
        if (e0 && e1) hipEventRecord(e0, ctx->stream);
    }
    ~ProfScope() {
        if (!(e0 && e1)) { if (e0) ctx->prof_free.push_back(e0); if (e1) ctx->prof_free.push_back(e1); return; }
        hipEventRecord(e1, ctx->stream);
        ctx->prof_pending.push_back({e0, e1, name, cells});
        if (ctx->prof_pending.size() >= 2048) sp_profile_flush(ctx);
    }
};

