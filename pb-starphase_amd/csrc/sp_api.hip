// sp_api.hip -- C ABI of libstarphase_hip: context, sequence sets, generic anchor/align entry points.
// (HLA entry points live in sp_hla.hip.)  Declared in include/starphase_hip.h.
#include "sp_internal.h"
#include <algorithm>
#include <thread>
#include <atomic>
#include <cstring>
#include <new>
#include <unistd.h>

extern "C" {

int32_t sp_abi_version(void) { return SP_ABI_VERSION; }

int32_t sp_device_count(int32_t* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? SP_OK : SP_ERR_NO_DEVICE;
}

int32_t sp_ctx_create(int32_t device, void* stream, sp_ctx** out) {
    if (!out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    // hardware queues: asked for before the first HIP call of this function (see sp_ctx_info in the header).  The variable only counts if the HIP runtime reads it when it
    // comes up: a host that initialised HIP before this call (torch, say) has the runtime's own default of 4 unless it exported the variable itself -- told apart by whether
    // this process already holds the compute driver's device file
    static std::atomic<int> effective{-1};              // GPU_MAX_HW_QUEUES as HIP met it (-1: not known yet)
    bool by_library = false;
    if (effective.load() < 0) {
        bool hip_up = false;
        char link[64];
        for (int fd = 0; fd < 1024 && !hip_up; ++fd) {
            char path[64]; std::snprintf(path, sizeof path, "/proc/self/fd/%d", fd);
            const ssize_t k = readlink(path, link, sizeof link - 1);
            if (k > 0) { link[k] = 0; hip_up = std::strcmp(link, "/dev/kfd") == 0; }
        }
        const char* had = std::getenv("GPU_MAX_HW_QUEUES");
        if (!had && !hip_up) { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); by_library = true; }
        // (a variable that is there while HIP is already up is taken at its word: a host may have set it in-process BEFORE its first HIP call -- bench.py does, through
        //  os.environ ahead of `import torch` -- and /proc/self/environ, the environment at exec time, cannot tell that from one set too late.  If it was too late, the
        //  persistent batches time out: three failures switch the library's own choice off for the context until sp_ctx_set_option("k8_persistent") is called again)
        const char* now = std::getenv("GPU_MAX_HW_QUEUES");
        effective.store(now ? std::atoi(now) : (hip_up ? 0 : 4));
    }
    const int hw_queues = effective.load() > 0 ? effective.load() : 4;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SP_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SP_ERR_NO_DEVICE;
    // gfx950 only: this library carries exactly one code object
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SP_ERR_NO_DEVICE;
    sp_ctx* ctx = new (std::nothrow) sp_ctx();
    if (!ctx) return SP_ERR_OUT_OF_MEMORY;
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    static bool set_here = false;                     // (a later context of the process finds the variable the first one set)
    if (by_library) set_here = true;
    ctx->hw_queues = hw_queues; ctx->hw_queues_by_library = set_here;
    ctx->hw_queues_effective = effective.load();
    { const char* kc = std::getenv("SP_K8_COMPOUND"); if (kc && *kc) ctx->k8_compound = std::atoi(kc); }
    { const char* ks = std::getenv("SP_K8_SIDE_ORDERS"); if (ks && *ks) ctx->k8_side_orders = std::max(0, std::min(3, std::atoi(ks))); }
    { const char* kp = std::getenv("SP_K8_PERSISTENT"); if (kp && *kp) ctx->k8_persistent = std::atoi(kp); }      // (several processes on one device cannot see each other's persistent batches: they switch the mode off)
    if (effective.load() == 0)
        ctx->warning = "the HIP runtime was initialised before the library's first context and GPU_MAX_HW_QUEUES was not set: the process's streams share the runtime's 4 hardware queues; "
                       "export GPU_MAX_HW_QUEUES=16 (or more) before the process initialises HIP";
    else if (hw_queues < 16)
        ctx->warning = "GPU_MAX_HW_QUEUES=" + std::to_string(hw_queues) + ": the library's streams share " + std::to_string(hw_queues) +
                       " hardware queues; export GPU_MAX_HW_QUEUES=16 (or more) before the process initialises HIP";
    if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return SP_ERR_HIP; }
        ctx->own_stream = true;
    }
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { delete ctx; return SP_ERR_HIP; }
    *out = ctx;
    return SP_OK;
}

int32_t sp_ctx_get_info(const sp_ctx* ctx, sp_ctx_info* out) {
    if (!ctx || !out) return SP_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    out->device = ctx->device; out->num_cus = ctx->num_cus; out->hw_queues = ctx->hw_queues; out->hw_queues_set_by_library = ctx->hw_queues_by_library ? 1 : 0;
    std::snprintf(out->warning, sizeof out->warning, "%s", ctx->warning.c_str());
    return SP_OK;
}

static int32_t upload_finish(sp_seqset* s);
void sp_ctx_destroy(sp_ctx* ctx) {
    if (!ctx) return;
    for (sp_ctx*& h : ctx->helper) if (h) { sp_ctx_destroy(h); h = nullptr; }
    hipSetDevice(ctx->device);
    if (ctx->uploading) (void)upload_finish(ctx->uploading);
    if (ctx->copy_stream) { hipStreamSynchronize(ctx->copy_stream); hipStreamDestroy(ctx->copy_stream); ctx->copy_stream = nullptr; }
    if (ctx->ctl_stream) { hipStreamSynchronize(ctx->ctl_stream); hipStreamDestroy(ctx->ctl_stream); ctx->ctl_stream = nullptr; }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    hipStreamSynchronize(ctx->stream);
    for (sp_seqset* s : ctx->live_sets) {                        // read sets that outlive the context keep their buffers and free them themselves
        for (void* p : { (void*)s->d_words, (void*)s->d_nplane, (void*)s->d_word_off, (void*)s->d_len, (void*)s->d_kcode, (void*)s->d_kpos, (void*)s->d_koff }) if (p) ctx->dev_cap.erase(p);
        s->ctx = nullptr;
    }
    for (auto& kv : ctx->dev_cap) (void)hipFree(kv.first);       // the idle ones
    if (ctx->scratch) hipFree(ctx->scratch);
    for (auto& kv : ctx->pool) if (kv.second.first) hipFree(kv.second.first);
    for (auto& kv : ctx->host_pool) if (kv.second.first) hipHostFree(kv.second.first);
    sp_profile_flush(ctx);
    for (hipEvent_t e : ctx->prof_free) hipEventDestroy(e);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

} // extern "C"
sp_ctx* sp_ctx_helper(sp_ctx* ctx, int i) {
    if (i < 0 || i > 6) return nullptr;
    if (!ctx->helper[i] && sp_ctx_create(ctx->device, nullptr, &ctx->helper[i]) != SP_OK) ctx->helper[i] = nullptr;
    if (ctx->helper[i]) { ctx->helper[i]->profiling = ctx->profiling; ctx->helper[i]->k5_block_pairs = ctx->k5_block_pairs; ctx->helper[i]->k8_persistent = ctx->k8_persistent; ctx->helper[i]->k8_side_orders = ctx->k8_side_orders; ctx->helper[i]->k8_compound = ctx->k8_compound; ctx->helper[i]->k8_side_max_blocks = ctx->k8_side_max_blocks; ctx->helper[i]->mm2_rescore = ctx->mm2_rescore; ctx->helper[i]->k1_best_n = ctx->k1_best_n; }
    return ctx->helper[i];
}
void sp_profile_merge(sp_ctx* into, sp_ctx* from) {
    sp_profile_flush(from);
    for (auto& kv : from->prof) { ProfileEntry& e = into->prof[kv.first]; e.ms += kv.second.ms; e.launches += kv.second.launches; e.cells += kv.second.cells; }
    from->prof.clear();
}
extern "C" {

int32_t sp_struct_size(const char* name) {
    if (!name) return -1;
#define SP_SZ(T) if (std::strcmp(name, #T) == 0) return (int32_t)sizeof(T);
    SP_SZ(sp_ctx_info)
    SP_SZ(sp_pair)
    SP_SZ(sp_aln)
    SP_SZ(sp_hla_db_desc)
    SP_SZ(sp_hla_realign)
    SP_SZ(sp_k1_seed_hit)
    SP_SZ(sp_hla_best)
    SP_SZ(sp_chain_problem)
    SP_SZ(sp_chain_result)
    SP_SZ(sp_region_hit)
    SP_SZ(sp_cyp_problem)
    SP_SZ(sp_cyp_call)
    SP_SZ(sp_cyp_locus)
    SP_SZ(sp_cyp_gene_def)
    SP_SZ(sp_cyp_config)
    SP_SZ(sp_cyp_db_stats)
    SP_SZ(sp_variant_problem)
    SP_SZ(sp_variant_result)
    SP_SZ(sp_sv_definitions)
    SP_SZ(sp_cons_config)
    SP_SZ(sp_cons_result)
    SP_SZ(sp_cons_problem)
    SP_SZ(sp_cons_output)
    SP_SZ(sp_priority_problem)
    SP_SZ(sp_priority_job)
    SP_SZ(sp_hla_call_config)
    SP_SZ(sp_hla_call)
    SP_SZ(sp_database_metadata)
    SP_SZ(sp_database_stats)
    SP_SZ(sp_gene_region)
    SP_SZ(sp_variant_detail)
    SP_SZ(sp_bam_read)
    SP_SZ(sp_affine_opts)
    SP_SZ(sp_affine_aln)
    SP_SZ(sp_k1_seed_info)
    SP_SZ(sp_cyp_region_variants)
    SP_SZ(sp_chain_build_info)
    SP_SZ(sp_variant_gene_stats)
    SP_SZ(sp_vcf_allele)
    SP_SZ(sp_vcf_deletion)
    SP_SZ(sp_mapping_stats)
#undef SP_SZ
    return -1;
}

const char* sp_last_error(const sp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int32_t sp_ctx_synchronize(sp_ctx* ctx) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int32_t sp_ctx_set_option(sp_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return SP_ERR_INVALID_ARG;
    if (std::strcmp(name, "hla_split_genes") == 0) { ctx->split_genes = value != 0; return SP_OK; }
    if (std::strcmp(name, "k1_best_n") == 0) { if (value < 0 || value > 8) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_ctx_set_option: k1_best_n is 0..8"); ctx->k1_best_n = (int)value; for (sp_ctx* h : ctx->helper) if (h) h->k1_best_n = ctx->k1_best_n; return SP_OK; }
    if (std::strcmp(name, "mm2_rescore") == 0) { ctx->mm2_rescore = (int)value; for (sp_ctx* h : ctx->helper) if (h) h->mm2_rescore = ctx->mm2_rescore; return SP_OK; }
    if (std::strcmp(name, "k8_side_orders") == 0) { if (value < 0 || value > 3) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_ctx_set_option: k8_side_orders is 0 .. 3"); ctx->k8_side_orders = (int)value; for (sp_ctx* h : ctx->helper) if (h) h->k8_side_orders = ctx->k8_side_orders; return SP_OK; }
    if (std::strcmp(name, "k8_compound") == 0) { ctx->k8_compound = (int)value; for (sp_ctx* h : ctx->helper) if (h) h->k8_compound = ctx->k8_compound; return SP_OK; }
    if (std::strcmp(name, "k8_side_max_blocks") == 0) { if (value < 0) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_ctx_set_option: k8_side_max_blocks >= 0"); ctx->k8_side_max_blocks = (int)value; for (sp_ctx* h : ctx->helper) if (h) h->k8_side_max_blocks = ctx->k8_side_max_blocks; return SP_OK; }
    if (std::strcmp(name, "k8_persistent") == 0) { if (value < 0 || value > 2) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_ctx_set_option: k8_persistent is 0 (never), 1 (whenever a batch fits) or 2 (the library decides)"); ctx->k8_persistent = (int)value; ctx->k8_persist_backoff = 0; ctx->k8_persist_failures = 0; for (sp_ctx* h : ctx->helper) if (h) h->k8_persistent = ctx->k8_persistent; return SP_OK; }
    if (std::strcmp(name, "cons_retry_ladder") == 0) { ctx->cons_retry_ladder = value != 0; for (sp_ctx* h : ctx->helper) if (h) h->cons_retry_ladder = ctx->cons_retry_ladder; return SP_OK; }
    if (std::strcmp(name, "k5_block_pairs") == 0) { if (value < 0 || value > (1 << 20)) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_ctx_set_option: k5_block_pairs is 0..1048576"); ctx->k5_block_pairs = (int)value; for (sp_ctx* h : ctx->helper) if (h) h->k5_block_pairs = ctx->k5_block_pairs; return SP_OK; }
    if (std::strcmp(name, "cyp_cohort_min_group") == 0) { if (value < 1 || value > 64) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_ctx_set_option: cyp_cohort_min_group is 1..64"); ctx->cyp_cohort_min_group = (int)value; return SP_OK; }
    if (std::strcmp(name, "cyp_cohort_streams") == 0) { if (value < 1 || value > 8) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_ctx_set_option: cyp_cohort_streams is 1..8"); ctx->cyp_cohort_streams = (int)value; return SP_OK; }
    if (std::strcmp(name, "hla_split_streams") == 0) { if (value < 1 || value > 4) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_ctx_set_option: hla_split_streams is 1..4"); ctx->split_streams = (int)value; return SP_OK; }
    return sp_fail(ctx, SP_ERR_INVALID_ARG, std::string("sp_ctx_set_option: unknown option ") + name);
}

int32_t sp_profile_reset(sp_ctx* ctx) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    for (sp_ctx* h : ctx->helper) if (h) sp_profile_reset(h);
    sp_profile_flush(ctx); ctx->prof.clear();
    unsigned long long* c = sp_counters(ctx);
    if (c) { hipSetDevice(ctx->device); hipMemsetAsync(c, 0, SPC_N * sizeof(unsigned long long), ctx->stream); }
    return SP_OK;
}

int32_t sp_profile_get(sp_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches, uint64_t* cells) {
    if (!ctx || !kernel) return SP_ERR_INVALID_ARG;
    sp_profile_flush(ctx);
    for (sp_ctx* h : ctx->helper) if (h) sp_profile_merge(ctx, h);  // what ran beside this context on its helpers counts as this context's
    // device counters: "count:<name>" returns the value in *cells (ms and launches are 0)
    static const struct { const char* name; int idx; } counters[] = {
        {"count:k1_cells_active", SPC_K1_ACTIVE}, {"count:k1_cells_executed", SPC_K1_EXECUTED}, {"count:k1_cells_resumed", SPC_K1_RESUMED},
        {"count:k1_cells_bytes", SPC_K1_BYTES}, {"count:cons_launches", SPC_CONS_LAUNCHES}, {"count:cons_columns", SPC_CONS_COLUMNS},
        // slots of the timing builds (profiles/scripts): zero in the production library
        {"count:dbg0", 6}, {"count:dbg1", 7}, {"count:dbg2", 8}, {"count:dbg3", 9}, {"count:dbg4", 10}, {"count:dbg5", 11}, {"count:dbg6", 12}, {"count:dbg7", 13} };
    for (const auto& c : counters) if (std::strcmp(kernel, c.name) == 0) {
        unsigned long long v = 0; unsigned long long* d = sp_counters(ctx);
        if (d) { hipSetDevice(ctx->device); if (hipMemcpyAsync(&v, d + c.idx, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, "sp_profile_get: counter read failed"); }
        for (sp_ctx* h : ctx->helper) if (h) {                       // what the helpers counted is this context's too
            unsigned long long hv = 0; unsigned long long* hd = sp_counters(h);
            if (hd && hipMemcpyAsync(&hv, hd + c.idx, 8, hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess) v += hv;
        }
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = 0;
        if (cells) *cells = v;
        return SP_OK;
    }
    auto it = ctx->prof.find(kernel);
    ProfileEntry e; if (it != ctx->prof.end()) e = it->second;
    if (total_ms) *total_ms = e.ms;
    if (launches) *launches = e.launches;
    if (cells) *cells = e.cells;
    return SP_OK;
}

// ------------------------------------------------------------------ sequence sets

// An upload under way: everything the worker thread needs, owned by the set until sp_seqset_wait joins the thread
struct sp_upload {
    std::thread th;
    int32_t rc = SP_OK; std::string err;
    int format = SP_SEQ_ASCII;
    const uint8_t* src = nullptr; uint64_t src_bytes = 0;
    std::vector<uint64_t> rel;                  // n + 1 offsets into the staged bytes
    uint8_t* d_stage = nullptr; uint64_t* d_off = nullptr; uint32_t* d_flag = nullptr;   // device staging (the context's pools)
    uint8_t* ring = nullptr; size_t chunk = 0; int slots = 0;                             // pinned staging ring (the context's)
    hipEvent_t slot_ev[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t wbytes = 0;
};

static constexpr size_t UPLOAD_CHUNK = (size_t)8 << 20;
static constexpr int UPLOAD_SLOTS = 4;

// the worker: tables, then the payload chunk by chunk through the pinned ring, then the packing kernel(s); ends when the set is complete
static void upload_run(sp_seqset* s) {
    sp_upload* u = s->up; sp_ctx* ctx = s->ctx;
    hipStream_t st = ctx->copy_stream;
    auto bad = [&](hipError_t e, const char* what) { if (e != hipSuccess && u->rc == SP_OK) { u->rc = SP_ERR_HIP; u->err = std::string(what) + ": " + hipGetErrorString(e); } return e != hipSuccess; };
    if (bad(hipSetDevice(ctx->device), "hipSetDevice")) return;
    const uint32_t n = s->n;
    bad(hipMemcpyAsync(s->d_word_off, s->h_word_off.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st), "offsets");
    if (n) bad(hipMemcpyAsync(s->d_len, s->h_len.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st), "lengths");
    bad(hipMemsetAsync(s->d_words, 0, u->wbytes, st), "clear");
    if (u->src_bytes && u->rc == SP_OK) {
        bad(hipMemcpyAsync(u->d_off, u->rel.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st), "staged offsets");
        bad(hipMemsetAsync(u->d_flag, 0, 4, st), "flag");
        int k = 0;
        for (uint64_t at = 0; at < u->src_bytes && u->rc == SP_OK; at += u->chunk, ++k) {
            const int slot = k % u->slots;
            const size_t len = (size_t)std::min<uint64_t>(u->chunk, u->src_bytes - at);
            if (k >= u->slots && bad(hipEventSynchronize(u->slot_ev[slot]), "staging slot")) break;     // the DMA that last read this slot
            std::memcpy(u->ring + (size_t)slot * u->chunk, u->src + at, len);
            if (bad(hipMemcpyAsync(u->d_stage + at, u->ring + (size_t)slot * u->chunk, len, hipMemcpyHostToDevice, st), "payload")) break;
            bad(hipEventRecord(u->slot_ev[slot], st), "staging event");
        }
        if (u->rc == SP_OK && sp_launch_pack_on(st, ctx->num_cus, u->format, u->d_stage, u->d_off, s->d_word_off, s->d_len, n, s->d_words, nullptr, u->d_flag) != SP_OK) { u->rc = SP_ERR_HIP; u->err = "pack kernel launch"; }
        uint32_t flag = 0;
        if (u->rc == SP_OK && u->format != SP_SEQ_PACKED2) {
            bad(hipMemcpyAsync(&flag, u->d_flag, 4, hipMemcpyDeviceToHost, st), "flag read");
            bad(hipStreamSynchronize(st), "upload");
            if (u->rc == SP_OK && flag) {
                s->has_n = true;
                if (!(s->d_nplane = (uint32_t*)sp_dev_alloc(ctx, u->wbytes))) { u->rc = SP_ERR_OUT_OF_MEMORY; u->err = "seqset nplane"; }
                else {
                    bad(hipMemsetAsync(s->d_nplane, 0, u->wbytes, st), "clear N plane");
                    if (sp_launch_pack_on(st, ctx->num_cus, u->format, u->d_stage, u->d_off, s->d_word_off, s->d_len, n, nullptr, s->d_nplane, u->d_flag) != SP_OK) { u->rc = SP_ERR_HIP; u->err = "pack kernel launch"; }
                }
            }
        }
    }
    bad(hipStreamSynchronize(st), "upload");
}

static int32_t upload_finish(sp_seqset* s) {
    sp_upload* u = s->up;
    if (!u) return s->up_rc == SP_OK ? SP_OK : sp_fail(s->ctx, s->up_rc, "seqset upload: " + s->up_err);     // (ended earlier: a later upload of the context waited for it)
    if (u->th.joinable()) u->th.join();
    sp_ctx* ctx = s->ctx;
    if (ctx->uploading == s) ctx->uploading = nullptr;
    s->up_rc = u->rc; s->up_err = u->err;
    for (hipEvent_t e : u->slot_ev) if (e) hipEventDestroy(e);
    delete u; s->up = nullptr;
    return s->up_rc == SP_OK ? SP_OK : sp_fail(ctx, s->up_rc, "seqset upload: " + s->up_err);
}

static int32_t upload_start(sp_ctx* ctx, int32_t format, const void* data, const uint64_t* offsets, const uint32_t* lengths, uint32_t n, sp_seqset** out) {
    if (!ctx || !out || (n && (!data || !offsets)) || format < SP_SEQ_ASCII || format > SP_SEQ_PACKED2 || (format != SP_SEQ_ASCII && n && !lengths)) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    hipSetDevice(ctx->device);
    if (ctx->uploading) (void)upload_finish(ctx->uploading);      // (how it ended stays in that set: its own sp_seqset_wait reports it)
    sp_seqset* s = new (std::nothrow) sp_seqset();
    if (!s) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seqset");
    s->ctx = ctx; s->n = n;
    s->h_len.resize(n); s->h_word_off.resize((size_t)n + 1);
    uint64_t total_words = 0;
    const uint64_t per_byte = format == SP_SEQ_ASCII ? 1 : format == SP_SEQ_BAM4 ? 2 : 4;
    for (uint32_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) { delete s; return sp_fail(ctx, SP_ERR_INVALID_ARG, "seqset: offsets must not decrease"); }
        const uint64_t bytes = offsets[i + 1] - offsets[i];
        uint64_t len = lengths ? lengths[i] : bytes;
        if (len > bytes * per_byte) { delete s; return sp_fail(ctx, SP_ERR_INVALID_ARG, "seqset: a sequence is longer than its bytes"); }
        if (len > 65534) { len = 0; s->n_skipped += 1; }          // an over-long sequence is left out of every alignment instead of failing the sample
        s->h_len[i] = (int32_t)len;
        s->max_len = std::max<int32_t>(s->max_len, (int32_t)len);
        s->h_word_off[i] = total_words;
        const uint64_t w = (len + 15) / 16 + 2;     // data + 2 guard words
        total_words += (w + 3) & ~3ull;             // 16-byte aligned starts
    }
    s->h_word_off[n] = total_words;
    { std::lock_guard<std::mutex> g(ctx->dev_cache_mu); ctx->live_sets.insert(s); }      // (from here on the set leaves through sp_seqset_free, which takes it off the list)
    auto fail = [&](const char* what) { sp_seqset_free(s); return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, what); };
    const size_t wbytes = (total_words + SP_SEQ_PAD_WORDS) * sizeof(uint32_t);
    if (!(s->d_words = (uint32_t*)sp_dev_alloc(ctx, wbytes))) return fail("seqset words");
    if (!(s->d_word_off = (uint64_t*)sp_dev_alloc(ctx, ((size_t)n + 1) * sizeof(uint64_t)))) return fail("seqset offsets");
    if (!(s->d_len = (int32_t*)sp_dev_alloc(ctx, std::max<size_t>(1, n) * sizeof(int32_t)))) return fail("seqset lengths");
    if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { sp_seqset_free(s); return sp_fail(ctx, SP_ERR_HIP, "copy stream"); }
    sp_upload* u = new (std::nothrow) sp_upload();
    if (!u) return fail("seqset upload");
    s->up = u; u->format = format; u->wbytes = wbytes;
    const uint64_t src_bytes = n ? offsets[n] - offsets[0] : 0;
    u->src = (const uint8_t*)data + (n ? offsets[0] : 0); u->src_bytes = src_bytes;
    if (src_bytes) {
        u->rel.resize((size_t)n + 1);
        for (uint32_t i = 0; i <= n; ++i) u->rel[i] = offsets[i] - offsets[0];
        u->chunk = std::min<size_t>(UPLOAD_CHUNK, (size_t)((src_bytes + 4095) & ~4095ull));
        u->slots = UPLOAD_SLOTS;
        u->d_stage = (uint8_t*)sp_pool(ctx, "upload_ascii", src_bytes);
        u->d_off = (uint64_t*)sp_pool(ctx, "upload_off", ((size_t)n + 1) * 8);
        u->d_flag = (uint32_t*)sp_pool(ctx, "upload_flag", 4);
        u->ring = (uint8_t*)sp_host_pool(ctx, "upload_ring", UPLOAD_CHUNK * UPLOAD_SLOTS);
        if (!u->d_stage || !u->d_off || !u->d_flag || !u->ring) return fail("seqset staging");
        for (int k = 0; k < u->slots; ++k) if (hipEventCreateWithFlags(&u->slot_ev[k], hipEventDisableTiming) != hipSuccess) { sp_seqset_free(s); return sp_fail(ctx, SP_ERR_HIP, "staging events"); }
    }
    ctx->uploading = s;
    try { u->th = std::thread(upload_run, s); }
    catch (...) { ctx->uploading = nullptr; sp_seqset_free(s); return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seqset upload thread"); }
    *out = s;
    return SP_OK;
}

int32_t sp_seqset_upload_async(sp_ctx* ctx, int32_t format, const void* data, const uint64_t* offsets, const uint32_t* lengths, uint32_t n, sp_seqset** out) {
    return upload_start(ctx, format, data, offsets, lengths, n, out);
}

int32_t sp_seqset_wait(sp_seqset* s) {
    if (!s) return SP_ERR_INVALID_ARG;
    return upload_finish(s);
}

int32_t sp_seqset_upload_format(sp_ctx* ctx, int32_t format, const void* data, const uint64_t* offsets, const uint32_t* lengths, uint32_t n, sp_seqset** out) {
    const int32_t rc = upload_start(ctx, format, data, offsets, lengths, n, out);
    if (rc != SP_OK) return rc;
    const int32_t rc2 = upload_finish(*out);
    if (rc2 != SP_OK) { sp_seqset_free(*out); *out = nullptr; }
    return rc2;
}

int32_t sp_seqset_upload(sp_ctx* ctx, const char* bases, const uint64_t* offsets, uint32_t n, sp_seqset** out) {
    return sp_seqset_upload_format(ctx, SP_SEQ_ASCII, bases, offsets, nullptr, n, out);
}

} // extern "C"
// ---- the cache of read-set buffers (sp_ctx::dev_cache)
static size_t dev_round(size_t bytes) {           // sizes that differ by a few per cent share a class: the samples of a run are about the same size
    size_t g = 4096; while (g * 16 < bytes) g <<= 1;               // granule: 1/16 .. 1/8 of the size
    return (std::max<size_t>(bytes, 1) + g - 1) / g * g;
}
void* sp_dev_alloc(sp_ctx* ctx, size_t bytes) {
    const size_t want = dev_round(bytes);
    {
        std::lock_guard<std::mutex> g(ctx->dev_cache_mu);
        auto it = ctx->dev_cache.lower_bound(want);
        if (it != ctx->dev_cache.end() && it->first <= 2 * want) { void* p = it->second; ctx->dev_cache_bytes -= it->first; ctx->dev_cache.erase(it); return p; }
    }
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        // out of memory with buffers lying idle: give them back and try once more
        std::vector<void*> idle;
        { std::lock_guard<std::mutex> g(ctx->dev_cache_mu); for (auto& kv : ctx->dev_cache) { idle.push_back(kv.second); ctx->dev_cap.erase(kv.second); } ctx->dev_cache.clear(); ctx->dev_cache_bytes = 0; }
        for (void* q : idle) (void)hipFree(q);
        if (hipMalloc(&p, want) != hipSuccess) return nullptr;
    }
    std::lock_guard<std::mutex> g(ctx->dev_cache_mu);
    ctx->dev_cap[p] = want;
    return p;
}
void sp_dev_release(sp_ctx* ctx, void* p) {
    if (!p) return;
    constexpr size_t MAX_BYTES = (size_t)2 << 30, MAX_BUFFERS = 4096;      // (a cohort pass frees a thousand small buffers at once: the bytes are the bound that matters)
    {
        std::lock_guard<std::mutex> g(ctx->dev_cache_mu);
        auto it = ctx->dev_cap.find(p);
        if (it == ctx->dev_cap.end()) { (void)hipFree(p); return; }        // (not one of ours)
        if (ctx->dev_cache.size() < MAX_BUFFERS && ctx->dev_cache_bytes + it->second <= MAX_BYTES) { ctx->dev_cache.emplace(it->second, p); ctx->dev_cache_bytes += it->second; return; }
        ctx->dev_cap.erase(it);
    }
    (void)hipFree(p);
}
extern "C" {

int32_t sp_seqset_skipped(const sp_seqset* s, uint32_t* n_skipped) { if (!s || !n_skipped) return SP_ERR_INVALID_ARG; *n_skipped = s->n_skipped; return SP_OK; }

void sp_seqset_free(sp_seqset* s) {
    if (!s) return;
    if (s->ctx) hipSetDevice(s->ctx->device);
    if (s->up) { (void)upload_finish(s); }                        // an upload still under way ends first
    if (s->ctx) {
        // the buffers go back to the context for the next read set (no hipFree: that waits for every stream of the device).  Every call that was given the set has
        // returned, i.e. has waited for its own streams; the context's stream is waited for once more here, which costs nothing when it is idle
        (void)hipStreamSynchronize(s->ctx->stream);
        { std::lock_guard<std::mutex> g(s->ctx->dev_cache_mu); s->ctx->live_sets.erase(s); }
        sp_dev_release(s->ctx, s->d_words); sp_dev_release(s->ctx, s->d_nplane); sp_dev_release(s->ctx, s->d_word_off); sp_dev_release(s->ctx, s->d_len);
        sp_dev_release(s->ctx, s->d_kcode); sp_dev_release(s->ctx, s->d_kpos); sp_dev_release(s->ctx, s->d_koff);
    } else {
        if (s->d_words) hipFree(s->d_words);
        if (s->d_nplane) hipFree(s->d_nplane);
        if (s->d_word_off) hipFree(s->d_word_off);
        if (s->d_len) hipFree(s->d_len);
        if (s->d_kcode) hipFree(s->d_kcode);
        if (s->d_kpos) hipFree(s->d_kpos);
        if (s->d_koff) hipFree(s->d_koff);
    }
    delete s;
}

int32_t sp_seqset_count(const sp_seqset* s, uint32_t* n) { if (!s || !n) return SP_ERR_INVALID_ARG; *n = s->n; return SP_OK; }
int32_t sp_seqset_length(const sp_seqset* s, uint32_t idx, uint32_t* len) {
    if (!s || !len || idx >= s->n) return SP_ERR_INVALID_ARG;
    *len = (uint32_t)s->h_len[idx]; return SP_OK;
}

} // extern "C"

void sp_profile_flush(sp_ctx* ctx) {
    for (auto& p : ctx->prof_pending) {
        hipEventSynchronize(p.e1);
        float ms = 0; hipEventElapsedTime(&ms, p.e0, p.e1);
        auto& e = ctx->prof[p.name]; e.ms += ms; e.launches += 1; e.cells += p.cells;
        ctx->prof_free.push_back(p.e0); ctx->prof_free.push_back(p.e1);
    }
    ctx->prof_pending.clear();
}

int sp_seqset_build_index(sp_ctx* ctx, sp_seqset* s);
static void kmer_tables(const sp_seqset* s, std::vector<uint64_t>& koff, std::vector<uint32_t>& kcode, std::vector<int32_t>& kpos);

int sp_seqset_fetch_host(sp_ctx* ctx, sp_seqset* s) {
    if (!s->h_words.empty()) return SP_OK;
    hipSetDevice(ctx->device);
    const size_t plane_words = (size_t)s->h_word_off[s->n] + SP_SEQ_PAD_WORDS;
    s->h_words.assign(plane_words * (s->has_n ? 2 : 1), 0);
    if (hipMemcpy(s->h_words.data(), s->d_words, plane_words * 4, hipMemcpyDeviceToHost) != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, "fetch packed words");
    if (s->has_n && hipMemcpy(s->h_words.data() + plane_words, s->d_nplane, plane_words * 4, hipMemcpyDeviceToHost) != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, "fetch N plane");
    return SP_OK;
}

// A short-lived set of a few sequences (a consensus, a backbone) without a single allocation: packed on the host, device buffers
// from the pool "<prefix>_*", k-mer index built straight from the host words.  The set is a value: never sp_seqset_free it; it
// stays valid until the next call with the same prefix.
int sp_seqset_make_small(sp_ctx* ctx, const char* prefix, const char* bases, const uint64_t* offsets, uint32_t n, bool with_index, sp_seqset* out) {
    hipSetDevice(ctx->device);
    *out = sp_seqset();
    sp_seqset& s = *out;
    s.ctx = ctx; s.n = n;
    s.h_len.resize(n); s.h_word_off.resize((size_t)n + 1);
    uint64_t total_words = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t len = offsets[i + 1] - offsets[i];
        if (offsets[i + 1] < offsets[i] || len > 65534) return sp_fail(ctx, SP_ERR_TOO_LONG, "seqset: sequence longer than 65,534 bases");
        s.h_len[i] = (int32_t)len; s.max_len = std::max<int32_t>(s.max_len, (int32_t)len);
        s.h_word_off[i] = total_words;
        total_words += (((len + 15) / 16 + 2) + 3) & ~3ull;
    }
    s.h_word_off[n] = total_words;
    const size_t plane_words = (size_t)total_words + SP_SEQ_PAD_WORDS;
    std::vector<uint32_t> words(plane_words, 0), nplane(plane_words, 0);
    std::atomic<bool> any_n(false);
    auto pack_one = [&](uint32_t i) {
        const char* src = bases + offsets[i];
        uint32_t* w = words.data() + s.h_word_off[i]; uint32_t* np = nplane.data() + s.h_word_off[i];
        for (int b = 0; b < s.h_len[i]; ++b) {
            uint32_t c = 0;
            switch (src[b]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break;
                              default: np[b >> 4] |= 1u << ((b & 15) << 1); any_n.store(true, std::memory_order_relaxed); }
            w[b >> 4] |= c << ((b & 15) << 1);
        }
    };
    // (the sequences own disjoint words: sets of many -- the consensuses of a cohort -- are packed by a few threads)
    const uint32_t pack_threads = n >= 32 ? std::min<uint32_t>(8, std::max<uint32_t>(1, std::thread::hardware_concurrency())) : 1;
    if (pack_threads > 1) {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < pack_threads; ++t) pool.emplace_back([&, t]() { for (uint32_t i = t; i < n; i += pack_threads) pack_one(i); });
        for (auto& th : pool) th.join();
    } else for (uint32_t i = 0; i < n; ++i) pack_one(i);
    s.has_n = any_n.load();
    s.h_words = words;
    if (s.has_n) s.h_words.insert(s.h_words.end(), nplane.begin(), nplane.end());
    // every array of the set goes into ONE device buffer through ONE pinned staging buffer: a single DMA instead of seven pageable
    // copies (each of those waits for the runtime's bounce buffer: ~0.1 ms per set, as much as the typing launches they feed)
    std::vector<uint64_t> koff; std::vector<uint32_t> kcode; std::vector<int32_t> kpos;
    if (with_index) kmer_tables(&s, koff, kcode, kpos);
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t b_words = plane_words * 4, b_npl = s.has_n ? plane_words * 4 : 0, b_off = ((size_t)n + 1) * 8, b_len = std::max<size_t>(1, n) * 4;
    const size_t b_kc = with_index ? std::max<size_t>(1, kcode.size()) * 4 : 0, b_kp = with_index ? std::max<size_t>(1, kpos.size()) * 4 : 0;
    const size_t b_ko = with_index ? koff.size() * 8 : 0;
    const size_t o_words = 0, o_npl = o_words + up16(b_words), o_off = o_npl + up16(b_npl), o_len = o_off + up16(b_off);
    const size_t o_kc = o_len + up16(b_len), o_kp = o_kc + up16(b_kc), o_ko = o_kp + up16(b_kp), total = o_ko + up16(b_ko);
    const std::string px(prefix);
    char* dev = (char*)sp_pool(ctx, (px + "_all").c_str(), total);
    char* stage = (char*)sp_host_pool(ctx, (px + "_stage").c_str(), total);
    if (!dev || !stage) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "small seqset");
    std::memcpy(stage + o_words, words.data(), b_words);
    if (s.has_n) std::memcpy(stage + o_npl, nplane.data(), b_npl);
    std::memcpy(stage + o_off, s.h_word_off.data(), b_off);
    if (n) std::memcpy(stage + o_len, s.h_len.data(), (size_t)n * 4);
    if (with_index) {
        if (!kcode.empty()) { std::memcpy(stage + o_kc, kcode.data(), kcode.size() * 4); std::memcpy(stage + o_kp, kpos.data(), kpos.size() * 4); }
        std::memcpy(stage + o_ko, koff.data(), b_ko);
    }
    s.d_words = (uint32_t*)(dev + o_words);
    s.d_nplane = s.has_n ? (uint32_t*)(dev + o_npl) : nullptr;
    s.d_word_off = (uint64_t*)(dev + o_off);
    s.d_len = (int32_t*)(dev + o_len);
    if (with_index) { s.d_kcode = (uint32_t*)(dev + o_kc); s.d_kpos = (int32_t*)(dev + o_kp); s.d_koff = (uint64_t*)(dev + o_ko); s.has_index = true; }
    hipMemcpyAsync(dev, stage, total, hipMemcpyHostToDevice, ctx->stream);
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // the staging buffer is reused by the next set with this prefix
    return SP_OK;
}

std::string sp_seqset_decode(sp_ctx* ctx, const sp_seqset* s, uint32_t i) {
    if (sp_seqset_fetch_host(ctx, const_cast<sp_seqset*>(s)) != SP_OK || i >= s->n) return std::string();
    const size_t plane_words = (size_t)s->h_word_off[s->n] + SP_SEQ_PAD_WORDS;
    const uint32_t* w = s->h_words.data() + s->h_word_off[i];
    const uint32_t* np = s->has_n ? s->h_words.data() + plane_words + s->h_word_off[i] : nullptr;
    std::string out((size_t)s->h_len[i], 'N');
    for (int h = 0; h < s->h_len[i]; ++h) {
        const uint32_t sh = (uint32_t)(h & 15) << 1;
        if (np && ((np[h >> 4] >> sh) & 1u)) continue;
        out[h] = "ACGT"[(w[h >> 4] >> sh) & 3u];
    }
    return out;
}

// sorted 16-mer table of every sequence of a set whose packed words are on the host
static void kmer_tables(const sp_seqset* s, std::vector<uint64_t>& koff, std::vector<uint32_t>& kcode, std::vector<int32_t>& kpos) {
    const size_t plane_words = (size_t)s->h_word_off[s->n] + SP_SEQ_PAD_WORDS;
    koff.assign((size_t)s->n + 1, 0); kcode.clear(); kpos.clear();
    // (code, position) pairs in position order, then a stable LSD radix sort on the code (3 passes of 11 bits): the table is sorted
    // by (code, position) like a comparison sort of the pairs would leave it, in a fraction of the time -- this runs on the host
    // between two launches every time a consensus is typed.  The sequences are independent: sets of many (the consensuses of a cohort)
    // are spread over a few threads.
    std::vector<std::vector<uint32_t>> codes(s->n); std::vector<std::vector<int32_t>> poss(s->n);
    auto one = [&](uint32_t i) {
        std::vector<uint32_t>& c0 = codes[i]; std::vector<int32_t>& p0 = poss[i];
        std::vector<uint32_t> c1; std::vector<int32_t> p1;
        const uint32_t* w = s->h_words.data() + s->h_word_off[i];
        const uint32_t* np = s->has_n ? s->h_words.data() + plane_words + s->h_word_off[i] : nullptr;
        const int len = s->h_len[i];
        for (int j = 0; j + SP_KMER <= len; ++j) {
            const int wi = j >> 4; const int sh = (j & 15) << 1;
            auto fetch = [&](const uint32_t* p) -> uint32_t {
                uint64_t v = ((uint64_t)p[wi + 1] << 32) | p[wi];
                return (uint32_t)(v >> sh);
            };
            if (np && fetch(np)) continue;
            c0.push_back(fetch(w)); p0.push_back(j);
        }
        const size_t m = c0.size();
        c1.resize(m); p1.resize(m);
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = pass * 11; const uint32_t mask = pass == 2 ? 0x3FFu : 0x7FFu;
            uint32_t count[2049] = {0};
            for (size_t x = 0; x < m; ++x) ++count[((c0[x] >> shift) & mask) + 1];
            for (int b = 0; b < 2048; ++b) count[b + 1] += count[b];
            for (size_t x = 0; x < m; ++x) { const uint32_t at = count[(c0[x] >> shift) & mask]++; c1[at] = c0[x]; p1[at] = p0[x]; }
            c0.swap(c1); p0.swap(p1);
        }
    };
    const uint32_t n_threads = s->n >= 32 ? std::min<uint32_t>(8, std::max<uint32_t>(1, std::thread::hardware_concurrency())) : 1;
    if (n_threads > 1) {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < n_threads; ++t) pool.emplace_back([&, t]() { for (uint32_t i = t; i < s->n; i += n_threads) one(i); });
        for (auto& th : pool) th.join();
    } else for (uint32_t i = 0; i < s->n; ++i) one(i);
    size_t total = 0;
    for (uint32_t i = 0; i < s->n; ++i) { koff[i] = total; total += codes[i].size(); }
    koff[s->n] = total;
    kcode.resize(total); kpos.resize(total);
    for (uint32_t i = 0; i < s->n; ++i) if (!codes[i].empty()) {
        std::memcpy(kcode.data() + koff[i], codes[i].data(), codes[i].size() * 4); std::memcpy(kpos.data() + koff[i], poss[i].data(), poss[i].size() * 4);
    }
}

// sorted 16-mer table of every sequence of the set (device code order: base t of the k-mer in bits 2t..2t+1)
int sp_seqset_build_index(sp_ctx* ctx, sp_seqset* s) {
    if (s->has_index) return SP_OK;
    hipSetDevice(ctx->device);
    const size_t plane_words = (size_t)s->h_word_off[s->n] + SP_SEQ_PAD_WORDS;
    if (s->h_words.empty()) {                              // packed on the device: fetch the packed words (and N plane) once
        s->h_words.assign(plane_words * (s->has_n ? 2 : 1), 0);
        hipMemcpy(s->h_words.data(), s->d_words, plane_words * 4, hipMemcpyDeviceToHost);
        if (s->has_n) hipMemcpy(s->h_words.data() + plane_words, s->d_nplane, plane_words * 4, hipMemcpyDeviceToHost);
    }
    std::vector<uint64_t> koff; std::vector<uint32_t> kcode; std::vector<int32_t> kpos;
    kmer_tables(s, koff, kcode, kpos);
    size_t ne = std::max<size_t>(1, kcode.size());
    // (through the owning context's cache when the set has one: sp_seqset_free hands them back there)
    sp_ctx* owner = s->ctx ? s->ctx : ctx;
    s->d_kcode = (uint32_t*)sp_dev_alloc(owner, ne * 4); s->d_kpos = (int32_t*)sp_dev_alloc(owner, ne * 4); s->d_koff = (uint64_t*)sp_dev_alloc(owner, koff.size() * 8);
    if (!s->d_kcode || !s->d_kpos || !s->d_koff)
        return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "k-mer index");
    if (!kcode.empty()) {
        hipMemcpy(s->d_kcode, kcode.data(), kcode.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(s->d_kpos, kpos.data(), kpos.size() * 4, hipMemcpyHostToDevice);
    }
    hipMemcpy(s->d_koff, koff.data(), koff.size() * 8, hipMemcpyHostToDevice);
    s->has_index = true;
    return SP_OK;
}

extern "C" {

int32_t sp_anchor_batch_topk(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                             const uint32_t* a_idx, const uint32_t* b_idx, uint64_t n_pairs, int32_t topk,
                             int32_t* diag_out, int32_t* votes_out) {
    if (!ctx || !A || !B || (n_pairs && (!a_idx || !b_idx || !diag_out || !votes_out))) return SP_ERR_INVALID_ARG;
    if (topk < 1 || topk > 8) return sp_fail(ctx, SP_ERR_INVALID_ARG, "anchor: topk must be 1..8");
    if (n_pairs == 0) return SP_OK;
    hipSetDevice(ctx->device);
    for (uint64_t i = 0; i < n_pairs; ++i) if (a_idx[i] >= A->n || b_idx[i] >= B->n) return sp_fail(ctx, SP_ERR_INVALID_ARG, "anchor: index out of range");
    int rc = sp_seqset_build_index(ctx, const_cast<sp_seqset*>(A));
    if (rc) return rc;
    const size_t nb = n_pairs * 4, nbk = nb * (size_t)topk;
    uint32_t* d_a = (uint32_t*)sp_pool(ctx, "anchor_a", nb); uint32_t* d_b = (uint32_t*)sp_pool(ctx, "anchor_b", nb);
    int32_t* d_d = (int32_t*)sp_pool(ctx, "anchor_d", nbk); int32_t* d_v = (int32_t*)sp_pool(ctx, "anchor_v", nbk);
    if (!d_a || !d_b || !d_d || !d_v) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "anchor buffers");
    hipMemcpyAsync(d_a, a_idx, nb, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(d_b, b_idx, nb, hipMemcpyHostToDevice, ctx->stream);
    rc = sp_launch_anchor(ctx, A, B, d_a, d_b, n_pairs, d_d, d_v, topk);
    if (rc == SP_OK) {
        hipMemcpyAsync(diag_out, d_d, nbk, hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(votes_out, d_v, nbk, hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, "anchor: stream sync failed");
    }
    return rc;
}

int32_t sp_anchor_batch(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                        const uint32_t* a_idx, const uint32_t* b_idx, uint64_t n_pairs,
                        int32_t* diag_out, int32_t* votes_out) {
    return sp_anchor_batch_topk(ctx, A, B, a_idx, b_idx, n_pairs, 1, diag_out, votes_out);
}

int32_t sp_align_batch(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                       const sp_pair* pairs, uint64_t n_pairs,
                       sp_aln* out, uint32_t* events, uint32_t events_stride) {
    if (!ctx || !A || !B || (n_pairs && (!pairs || !out))) return SP_ERR_INVALID_ARG;
    if (events && (events_stride == 0 || events_stride > SP_MAX_ED)) return sp_fail(ctx, SP_ERR_INVALID_ARG, "align: events_stride must be 1..SP_MAX_ED");
    if (n_pairs == 0) return SP_OK;
    hipSetDevice(ctx->device);
    std::vector<CellDesc> cells(n_pairs);
    for (uint64_t i = 0; i < n_pairs; ++i) {
        if (pairs[i].a >= A->n || pairs[i].b >= B->n) return sp_fail(ctx, SP_ERR_INVALID_ARG, "align: index out of range");
        if (pairs[i].max_ed < 0 || pairs[i].max_ed > SP_MAX_ED) return sp_fail(ctx, SP_ERR_INVALID_ARG, "align: max_ed must be 0..SP_MAX_ED");
        cells[i] = CellDesc{pairs[i].a, pairs[i].b, pairs[i].diag, pairs[i].max_ed, 0, -1};
    }
    // (the context's pooled buffers: three allocations and three frees per call -- a free waits for the device -- were most of the 4 ms this call took for the
    //  three consensus sequences K9 places on the backbone)
    CellDesc* d_cells = (CellDesc*)sp_pool(ctx, "align_cells", n_pairs * sizeof(CellDesc)); sp_aln* d_out = (sp_aln*)sp_pool(ctx, "align_out", n_pairs * sizeof(sp_aln));
    uint32_t* d_ev = events ? (uint32_t*)sp_pool(ctx, "align_ev", n_pairs * (size_t)events_stride * 4) : nullptr;
    int rc = SP_OK;
    if (!d_cells || !d_out || (events && !d_ev)) {
        rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "align buffers");
    } else {
        hipMemcpyAsync(d_cells, cells.data(), n_pairs * sizeof(CellDesc), hipMemcpyHostToDevice, ctx->stream);
        if (events) hipMemsetAsync(d_ev, 0, n_pairs * (size_t)events_stride * 4, ctx->stream);
        rc = sp_launch_cells(ctx, A, B, d_cells, n_pairs, d_out, d_ev, events_stride, events ? "align_trace" : "align", 2);
        if (rc == SP_OK) {
            hipMemcpyAsync(out, d_out, n_pairs * sizeof(sp_aln), hipMemcpyDeviceToHost, ctx->stream);
            if (events) hipMemcpyAsync(events, d_ev, n_pairs * (size_t)events_stride * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, std::string("align: ") + hipGetErrorString(hipGetLastError()));
        }
    }
    return rc;
}

} // extern "C"
