// sp_group.hip -- the one exchange step of the path: gathering the call records of the ranks of a node (SURVEY.md 8(b) sp_gather_results, 8(e)).
//
// The reference has no communication at all (a cohort is N independent process runs, src/cli/diplotype.rs:185-191); BASELINE.json's north_star
// shards the samples of a cohort over the GPUs of a node and names the one collective: "RCCL over xGMI used only to gather per-gene results".
// One process per GPU, one sp_ctx per process; the records (a few hundred bytes per rank and batch) are gathered with ONE ncclAllGather on
// the context's stream.  librccl is opened at run time (dlopen): a single-GPU caller never needs it, and a host process that already
// carries RCCL (torch) shares that copy.
#include "sp_internal.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

namespace {

struct NcclId { char internal[SP_GROUP_ID_BYTES]; };
typedef void* ncclComm_t;
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

void rccl_load(Rccl& R);
Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;                      // (two host threads on different contexts may come here together: the table is filled exactly once, before anyone reads it)
    std::call_once(once, []() { rccl_load(R); });
    return &R;
}
void rccl_load(Rccl& R) {
    // The RCCL to use is the one that sits on the HIP runtime THIS library is bound to: a Python process with torch carries a second HIP runtime
    // and a second RCCL (torch/lib), and a stream of one runtime means nothing to the other.  Where our own hipMalloc lives tells which runtime
    // that is; its directory holds the matching librccl.  RTLD_DEEPBIND makes that RCCL resolve its HIP calls through its own dependencies
    // (the same file, hence the same loaded runtime) whatever else is in the global scope.  SP_RCCL_PATH names a particular file instead.
    const char* env = std::getenv("SP_RCCL_PATH");
    if (env && *env) R.h = dlopen(env, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (!R.h) {
        Dl_info info;
        hipError_t (*own_malloc)(void**, size_t) = &hipMalloc;          // (the plain overload: where it lives is the runtime this library calls)
        if (dladdr(reinterpret_cast<void*>(own_malloc), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t cut = dir.rfind('/');
            if (cut != std::string::npos) {
                dir.resize(cut + 1);
                for (const char* n : { "librccl.so.1", "librccl.so" }) { if (R.h) break; R.h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND); }
            }
        }
    }
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) { if (R.h) break; R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND); }
    if (!R.h) { R.err = "librccl.so not found"; return; }
    R.GetUniqueId = (int (*)(NcclId*))dlsym(R.h, "ncclGetUniqueId");
    R.CommInitRank = (int (*)(ncclComm_t*, int, NcclId, int))dlsym(R.h, "ncclCommInitRank");
    R.CommDestroy = (int (*)(ncclComm_t))dlsym(R.h, "ncclCommDestroy");
    R.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))dlsym(R.h, "ncclAllGather");
    R.GetErrorString = (const char* (*)(int))dlsym(R.h, "ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather) { R.err = "librccl.so lacks the collective entry points"; dlclose(R.h); R.h = nullptr; }
}

} // namespace

struct sp_group {
    sp_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
};

extern "C" {

int32_t sp_group_unique_id(uint8_t* id) {
    if (!id) return SP_ERR_INVALID_ARG;
    Rccl* R = rccl();
    if (!R->h) return SP_ERR_NO_DEVICE;
    NcclId u;
    if (R->GetUniqueId(&u) != 0) return SP_ERR_HIP;
    std::memcpy(id, u.internal, SP_GROUP_ID_BYTES);
    return SP_OK;
}

int32_t sp_group_create(sp_ctx* ctx, const uint8_t* id, int32_t rank, int32_t n_ranks, sp_group** out) {
    if (!ctx || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    Rccl* R = rccl();
    if (!R->h) return sp_fail(ctx, SP_ERR_NO_DEVICE, "sp_group_create: " + R->err);
    SP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    sp_group* g = new (std::nothrow) sp_group();
    if (!g) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_group");
    g->ctx = ctx; g->rank = rank; g->n_ranks = n_ranks;
    NcclId u; std::memcpy(u.internal, id, SP_GROUP_ID_BYTES);
    const int rc = R->CommInitRank(&g->comm, n_ranks, u, rank);
    if (rc != 0) { delete g; return sp_fail(ctx, SP_ERR_HIP, std::string("ncclCommInitRank: ") + (R->GetErrorString ? R->GetErrorString(rc) : "failed")); }
    *out = g;
    return SP_OK;
}

void sp_group_free(sp_group* g) {
    if (!g) return;
    Rccl* R = rccl();
    if (g->comm && R->h) { hipSetDevice(g->ctx->device); hipStreamSynchronize(g->ctx->stream); R->CommDestroy(g->comm); }
    delete g;
}

int32_t sp_group_size(const sp_group* g, int32_t* rank, int32_t* n_ranks) {
    if (!g) return SP_ERR_INVALID_ARG;
    if (rank) *rank = g->rank;
    if (n_ranks) *n_ranks = g->n_ranks;
    return SP_OK;
}

int32_t sp_gather_results(sp_group* g, const void* records, uint64_t bytes_per_rank, void* all_records) {
    if (!g || (bytes_per_rank && (!records || !all_records))) return SP_ERR_INVALID_ARG;
    if (bytes_per_rank == 0) return SP_OK;
    sp_ctx* ctx = g->ctx;
    Rccl* R = rccl();
    SP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t total = (size_t)bytes_per_rank * (size_t)g->n_ranks;
    // pinned staging both ways (the records are host data of the caller), device buffers from the context's pools
    uint8_t* h_send = (uint8_t*)sp_host_pool(ctx, "gather_send", bytes_per_rank);
    uint8_t* h_recv = (uint8_t*)sp_host_pool(ctx, "gather_recv", total);
    uint8_t* d_send = (uint8_t*)sp_pool(ctx, "gather_send", bytes_per_rank);
    uint8_t* d_recv = (uint8_t*)sp_pool(ctx, "gather_recv", total);
    if (!h_send || !h_recv || !d_send || !d_recv) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_gather_results staging");
    std::memcpy(h_send, records, bytes_per_rank);
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_send, h_send, bytes_per_rank, hipMemcpyHostToDevice, ctx->stream));
    const int rc = R->AllGather(d_send, d_recv, (size_t)bytes_per_rank, /* ncclInt8 */ 0, g->comm, ctx->stream);
    if (rc != 0) return sp_fail(ctx, SP_ERR_HIP, std::string("ncclAllGather: ") + (R->GetErrorString ? R->GetErrorString(rc) : "failed"));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(h_recv, d_recv, total, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(all_records, h_recv, total);
    return SP_OK;
}

} // extern "C"
