// sp_hla_dict.hip -- the 16-mer dictionary of an allele set (gfx950): K2 compares every consensus of a gene with every allele of that
// gene (src/hla/caller.rs:1411-1510), and the alleles of a gene share almost all of their 16-mers.  Built once per database and level
// (cDNA, DNA), on the device:
//   codes  the distinct 16-mers of each gene's alleles, numbered in the order of their FIRST appearance (allele by allele, position by
//          position), genes one after the other
//   ids    every allele again, position by position, as the number of its 16-mer in its gene's part of the dictionary -- the alleles
//          of a gene are near-copies of each other, so the ids of neighbouring positions are almost always neighbouring numbers and a
//          wavefront's table reads by id fall into a few cache lines
// so that the anchor of a (consensus, allele) pair needs one table look-up per DISTINCT 16-mer and consensus (k2_hits_kernel) and a
// table read per vote (k2_anchor_dict_kernel, sp_hla.hip) instead of a binary search per vote.  The votes are the same votes.
#include <cstring>
#include "sp_internal.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>

namespace {

constexpr unsigned long long NO_KEY = 0xFFFFFFFFFFFFFFFFull;

// one workgroup per allele: key of the 16-mer at every position (NO_KEY when it holds a base outside ACGT) and the slot itself as value
__global__ __launch_bounds__(256) void dict_keys_kernel(SeqSetView S, const uint32_t* __restrict__ gene_of, const uint64_t* __restrict__ id_off,
                                                        unsigned long long* __restrict__ keys, uint32_t* __restrict__ slots) {
    const uint32_t a = blockIdx.x;
    const int len = S.len[a];
    const uint32_t* w = S.words + S.word_off[a];
    const uint32_t* np = S.nplane ? S.nplane + S.word_off[a] : nullptr;
    const unsigned long long g = (unsigned long long)gene_of[a] << 32;
    for (int j = threadIdx.x; j + SP_KMER <= len; j += blockDim.x) {
        const int wi = j >> 4; const uint32_t sh = (uint32_t)((j & 15) << 1);
        const bool has_n = np && __builtin_amdgcn_alignbit(np[wi + 1], np[wi], sh) != 0;
        keys[id_off[a] + j] = has_n ? NO_KEY : (g | __builtin_amdgcn_alignbit(w[wi + 1], w[wi], sh));
        if (slots) slots[id_off[a] + j] = (uint32_t)(id_off[a] + j);
    }
}

// distinct key u (ascending key order) first appears in slot first[u]: (gene << 32 | first slot) orders the dictionary of a gene by first appearance
__global__ __launch_bounds__(256) void dict_order_keys_kernel(const unsigned long long* __restrict__ uk, const uint32_t* __restrict__ first, uint32_t n,
                                                              unsigned long long* __restrict__ k2, uint32_t* __restrict__ v2) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    k2[u] = (uk[u] & 0xFFFFFFFF00000000ull) | first[u]; v2[u] = u;
}

// entry i of the ordered dictionary is distinct key order[i]: its number inside its gene's part, and its code
__global__ __launch_bounds__(256) void dict_number_kernel(const unsigned long long* __restrict__ k2s, const uint32_t* __restrict__ order, const unsigned long long* __restrict__ uk, uint32_t n,
                                                          const uint32_t* __restrict__ dict_off, uint32_t* __restrict__ number_of, uint32_t* __restrict__ code) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = order[i];
    number_of[u] = i - dict_off[(uint32_t)(k2s[i] >> 32)];
    code[i] = (uint32_t)uk[u];
}

// number of every slot's 16-mer: its key is found among the distinct keys, whose numbers are known
__global__ __launch_bounds__(256) void dict_ids_kernel(const unsigned long long* __restrict__ keys, uint64_t n, const unsigned long long* __restrict__ uk, uint32_t n_dict,
                                                       const uint32_t* __restrict__ number_of, uint32_t* __restrict__ ids) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    if (k == NO_KEY) { ids[i] = 0xFFFFFFFFu; return; }
    uint32_t lo = 0, hi = n_dict;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (uk[mid] < k) lo = mid + 1; else hi = mid; }
    ids[i] = number_of[lo];
}

} // namespace

void sp_k2_dict_free(K2Dict* d) {
    if (!d) return;
    if (d->d_ids) (void)hipFree(d->d_ids);
    if (d->d_id_off) (void)hipFree(d->d_id_off);
    if (d->d_code) (void)hipFree(d->d_code);
    if (d->d_dict_off) (void)hipFree(d->d_dict_off);
    *d = K2Dict();
}

int sp_k2_dict_build(sp_ctx* ctx, const sp_seqset* set, const uint32_t* d_gene_of, uint32_t n_genes, K2Dict* out) {
    *out = K2Dict();
    const uint32_t n = set->n;
    std::vector<uint64_t> id_off((size_t)n + 1, 0);
    for (uint32_t a = 0; a < n; ++a) id_off[a + 1] = id_off[a] + (uint64_t)std::max(0, set->h_len[a] - SP_KMER + 1);
    const uint64_t total = id_off[n];
    if (total == 0 || total >= (1ull << 32)) { out->failed = true; return SP_OK; }       // nothing to index (or more slots than a 32-bit slot number holds)
    std::vector<void*> temps;                                                             // everything but the tables that stay
    auto grab = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; temps.push_back(q); return q; };
    auto drop = [&]() { for (void* q : temps) (void)hipFree(q); temps.clear(); };
    auto fail = [&](const char* what) { drop(); sp_k2_dict_free(out); out->failed = true; return sp_fail(ctx, SP_ERR_HIP, what); };
    hipStream_t st = ctx->stream;
    auto* keys = (unsigned long long*)grab(total * 8); auto* sk = (unsigned long long*)grab(total * 8);
    auto* slots = (uint32_t*)grab(total * 4); auto* ss = (uint32_t*)grab(total * 4);
    auto* uk = (unsigned long long*)grab(total * 8); auto* first = (uint32_t*)grab(total * 4);
    auto* d_count = (size_t*)grab(sizeof(size_t));
    if (!keys || !sk || !slots || !ss || !uk || !first || !d_count || hipMalloc(&out->d_id_off, ((size_t)n + 1) * 8) != hipSuccess) return fail("k-mer dictionary: buffers");
    if (hipMemcpyAsync(out->d_id_off, id_off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail("k-mer dictionary: offsets");
    // (slots a sequence shorter than 16 bases would own do not exist; every other slot is written by the kernel)
    hipLaunchKernelGGL(dict_keys_kernel, dim3(n), dim3(256), 0, st, set->view(), d_gene_of, out->d_id_off, keys, slots);
    // stable sort by key: the slots of equal keys stay in ascending order, so the first of a run is the key's first appearance
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, keys, sk, slots, ss, (size_t)total, 0, 64, st) != hipSuccess) return fail("k-mer dictionary: sort size");
    void* ws = grab(bytes);
    if (!ws || rocprim::radix_sort_pairs(ws, bytes, keys, sk, slots, ss, (size_t)total, 0, 64, st) != hipSuccess) return fail("k-mer dictionary: sort");
    size_t b2 = 0;
    if (rocprim::unique_by_key(nullptr, b2, sk, ss, uk, first, d_count, (size_t)total, rocprim::equal_to<unsigned long long>(), st) != hipSuccess) return fail("k-mer dictionary: unique size");
    void* ws2 = grab(b2);
    if (!ws2 || rocprim::unique_by_key(ws2, b2, sk, ss, uk, first, d_count, (size_t)total, rocprim::equal_to<unsigned long long>(), st) != hipSuccess) return fail("k-mer dictionary: unique");
    size_t n_uniq = 0;
    if (hipMemcpyAsync(&n_uniq, d_count, sizeof(size_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("k-mer dictionary: count");
    std::vector<unsigned long long> h_uk(n_uniq);
    if (n_uniq && hipMemcpy(h_uk.data(), uk, n_uniq * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail("k-mer dictionary: keys");
    if (n_uniq && h_uk.back() == NO_KEY) { h_uk.pop_back(); --n_uniq; }                       // the 16-mers that hold an N sort last
    if (n_uniq == 0) { drop(); sp_k2_dict_free(out); out->failed = true; return SP_OK; }
    out->dict_off.assign((size_t)n_genes + 1, 0);
    for (uint32_t g = 0; g <= n_genes; ++g)
        out->dict_off[g] = (uint32_t)(std::lower_bound(h_uk.begin(), h_uk.end(), (unsigned long long)g << 32) - h_uk.begin());
    out->n_dict = (uint32_t)n_uniq; out->max_dict = 0;
    for (uint32_t g = 0; g < n_genes; ++g) out->max_dict = std::max(out->max_dict, out->dict_off[g + 1] - out->dict_off[g]);
    // the dictionary of every gene in the order of first appearance
    auto* k2 = (unsigned long long*)grab(n_uniq * 8); auto* k2s = (unsigned long long*)grab(n_uniq * 8);
    auto* v2 = (uint32_t*)grab(n_uniq * 4); auto* order = (uint32_t*)grab(n_uniq * 4); auto* number_of = (uint32_t*)grab(n_uniq * 4);
    if (!k2 || !k2s || !v2 || !order || !number_of || hipMalloc(&out->d_code, n_uniq * 4) != hipSuccess || hipMalloc(&out->d_dict_off, ((size_t)n_genes + 1) * 4) != hipSuccess ||
        hipMalloc(&out->d_ids, total * 4) != hipSuccess)
        return fail("k-mer dictionary: tables");
    (void)hipMemcpyAsync(out->d_dict_off, out->dict_off.data(), ((size_t)n_genes + 1) * 4, hipMemcpyHostToDevice, st);
    const unsigned nb = (unsigned)((n_uniq + 255) / 256);
    hipLaunchKernelGGL(dict_order_keys_kernel, dim3(nb), dim3(256), 0, st, uk, first, (uint32_t)n_uniq, k2, v2);
    size_t b3 = 0;
    if (rocprim::radix_sort_pairs(nullptr, b3, k2, k2s, v2, order, n_uniq, 0, 64, st) != hipSuccess) return fail("k-mer dictionary: order size");
    void* ws3 = grab(b3);
    if (!ws3 || rocprim::radix_sort_pairs(ws3, b3, k2, k2s, v2, order, n_uniq, 0, 64, st) != hipSuccess) return fail("k-mer dictionary: order");
    hipLaunchKernelGGL(dict_number_kernel, dim3(nb), dim3(256), 0, st, k2s, order, uk, (uint32_t)n_uniq, out->d_dict_off, number_of, out->d_code);
    // the keys were sorted out of place: `keys` still holds every slot's key
    hipLaunchKernelGGL(dict_ids_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, keys, total, uk, (uint32_t)n_uniq, number_of, out->d_ids);
    const hipError_t e = hipStreamSynchronize(st);
    drop();
    if (e != hipSuccess) { sp_k2_dict_free(out); out->failed = true; return sp_fail(ctx, SP_ERR_HIP, "k-mer dictionary: ids"); }
    out->built = true;
    return SP_OK;
}
