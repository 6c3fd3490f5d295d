// sp_hla_call.hip -- the per-gene HLA solve that ties K1, K8 and K2 together (gfx950).
//
// Host-side counterpart of the gene loop of diplotype_hla_batch (src/hla/caller.rs:642-1040) without its I/O and debug
// artefacts: realigned segments of the gene -> dual consensus on homopolymer-compressed segments, DNA segments if that
// does not pass (run_dual_consensus_with_offsets, :1118-1219) -> optional hemizygosity test (:676-684) -> one consensus per
// read group on the DNA segments (:706-747) -> typing of each consensus (score_consensus, :756,829) -> heterozygous /
// homozygous decision (:889-912).  Segments never leave the GPU: they are cut out of the packed reads and
// homopolymer-compressed by the two kernels below.
#include "sp_internal.h"
#include <algorithm>
#include <cstring>

namespace {

constexpr int SEG_WAVES = 4;
constexpr int HPC_WAVES = 2;
constexpr int HPC_MAX = 32768;          // bases per segment the compression kernel stages in LDS

// segment s = bases [start[s], start[s] + len[s]) of read idx[s]; output in the packed layout of sp_seqset
__global__ void __launch_bounds__(SEG_WAVES * SP_WAVE) seg_slice_kernel(SeqSetView reads, const uint32_t* __restrict__ idx, const int32_t* __restrict__ start,
                                                                        const int32_t* __restrict__ len, const uint64_t* __restrict__ word_off, int n,
                                                                        uint32_t* __restrict__ out_words, uint32_t* __restrict__ out_nplane) {
    const int lane = threadIdx.x & 63, s = blockIdx.x * SEG_WAVES + (threadIdx.x >> 6);
    if (s >= n) return;
    const uint32_t r = idx[s];
    const uint32_t* src = reads.words + reads.word_off[r];
    const uint32_t* nsrc = reads.nplane ? reads.nplane + reads.word_off[r] : nullptr;
    const int p0 = start[s], L = len[s];
    const int nw = (int)(word_off[s + 1] - word_off[s]);
    const int w0 = p0 >> 4; const uint32_t sh = (uint32_t)(p0 & 15) << 1;
    for (int j = lane; j < nw; j += SP_WAVE) {
        uint32_t v = 0, nv = 0;
        const int first = j << 4;
        if (first < L) {
            v = __builtin_amdgcn_alignbit(src[w0 + j + 1], src[w0 + j], sh);
            if (nsrc) nv = __builtin_amdgcn_alignbit(nsrc[w0 + j + 1], nsrc[w0 + j], sh);
            const int rem = L - first;
            if (rem < 16) { const uint32_t m = (1u << (rem << 1)) - 1; v &= m; nv &= m; }
        }
        out_words[word_off[s] + j] = v;
        if (out_nplane) out_nplane[word_off[s] + j] = nv;
    }
}

// hpc_bytes (src/util/homopolymers.rs:18-23) of every sequence of a packed set; the output set shares the input's word offsets
__global__ void __launch_bounds__(HPC_WAVES * SP_WAVE) hpc_kernel(const uint32_t* __restrict__ words, const uint32_t* __restrict__ nplane,
                                                                  const uint64_t* __restrict__ word_off, const int32_t* __restrict__ len, int n,
                                                                  uint32_t* __restrict__ out_words, uint32_t* __restrict__ out_nplane, int32_t* __restrict__ out_len) {
    __shared__ uint8_t codes[HPC_WAVES][HPC_MAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, s = blockIdx.x * HPC_WAVES + wv;
    if (s >= n) return;
    const uint32_t* src = words + word_off[s];
    const uint32_t* nsrc = nplane ? nplane + word_off[s] : nullptr;
    const int L = len[s];
    auto base = [&](int i) -> int {
        const uint32_t sh = (uint32_t)(i & 15) << 1;
        if (nsrc && ((nsrc[i >> 4] >> sh) & 1u)) return 4;
        return (int)((src[i >> 4] >> sh) & 3u);
    };
    int kept = 0;
    for (int b0 = 0; b0 < L; b0 += SP_WAVE) {
        const int i = b0 + lane;
        const bool valid = i < L;
        const int code = valid ? base(i) : 7;
        const int prev = (valid && i > 0) ? base(i - 1) : 9;
        const bool keep = valid && code != prev;
        const unsigned long long m = __ballot(keep);
        if (keep) codes[wv][kept + __popcll(m & ((1ull << lane) - 1))] = (uint8_t)code;
        kept += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nw = (int)(word_off[s + 1] - word_off[s]);
    for (int j = lane; j < nw; j += SP_WAVE) {
        uint32_t v = 0, nv = 0;
        for (int x = 0; x < 16; ++x) {
            const int p = (j << 4) + x;
            if (p < kept) { const uint32_t c = codes[wv][p]; if (c == 4) nv |= 1u << (x << 1); else v |= c << (x << 1); }
        }
        out_words[word_off[s] + j] = v;
        if (out_nplane) out_nplane[word_off[s] + j] = nv;
    }
    if (lane == 0) out_len[s] = kept;
}

// a library-owned packed set living in pooled buffers
struct PooledSet {
    sp_seqset set;
    uint64_t* d_word_off = nullptr;
};

} // namespace

extern "C" {

int32_t sp_hla_diplotype_gene(sp_ctx* ctx, const sp_hla_db* db, uint32_t gene, const sp_seqset* reads, const sp_hla_realign* realign,
                              const sp_hla_call_config* cfg, sp_hla_call* call, char* cons1, char* cons2, uint32_t cap, uint8_t* is_cons1_out) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!db || !reads || !realign || !cfg || !call || !cons1 || !cons2 || cap == 0) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_hla_diplotype_gene: null argument");
    std::memset(call, 0, sizeof *call);
    call->allele1 = call->allele2 = call->typed1 = call->typed2 = -1;
    cons1[0] = cons2[0] = '\0';
    // realigned_records of this gene, in input (qname) order
    std::vector<uint32_t> sel;
    for (uint32_t r = 0; r < reads->n; ++r) if (realign[r].status == 0 && realign[r].gene == (int32_t)gene) sel.push_back(r);
    const uint32_t n = (uint32_t)sel.size();
    call->n_reads = (int32_t)n;
    if (n == 0) { call->status = 1; return SP_OK; }                                 // NO_READS / NO_CALL (caller.rs:662-668)
    SP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    // ---- segments on the device
    std::vector<int32_t> h_start(n), h_len(n);
    std::vector<uint64_t> h_woff(n + 1, 0);
    int32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const sp_hla_realign& q = realign[sel[i]];
        h_start[i] = q.seg_start; h_len[i] = q.seg_end - q.seg_start;
        if (h_len[i] <= 0 || q.seg_end > reads->h_len[sel[i]]) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_hla_diplotype_gene: segment outside its read");
        if (h_len[i] > HPC_MAX) return sp_fail(ctx, SP_ERR_TOO_LONG, "sp_hla_diplotype_gene: segment longer than 32,768 bases");
        max_len = std::max(max_len, h_len[i]);
        h_woff[i + 1] = h_woff[i] + (uint64_t)((((h_len[i] + 15) >> 4) + 2 + 3) & ~3);
    }
    const uint64_t total_words = h_woff[n];
    uint32_t* d_idx = (uint32_t*)sp_pool(ctx, "hc_idx", sizeof(uint32_t) * n);
    int32_t* d_start = (int32_t*)sp_pool(ctx, "hc_start", sizeof(int32_t) * n);
    int32_t* d_len = (int32_t*)sp_pool(ctx, "hc_len", sizeof(int32_t) * n);
    int32_t* d_hlen = (int32_t*)sp_pool(ctx, "hc_hlen", sizeof(int32_t) * n);
    uint64_t* d_woff = (uint64_t*)sp_pool(ctx, "hc_woff", sizeof(uint64_t) * (n + 1));
    uint32_t* d_seg = (uint32_t*)sp_pool(ctx, "hc_seg", sizeof(uint32_t) * total_words);
    uint32_t* d_hpc = (uint32_t*)sp_pool(ctx, "hc_hpc", sizeof(uint32_t) * total_words);
    uint32_t* d_segn = reads->has_n ? (uint32_t*)sp_pool(ctx, "hc_segn", sizeof(uint32_t) * total_words) : nullptr;
    uint32_t* d_hpcn = reads->has_n ? (uint32_t*)sp_pool(ctx, "hc_hpcn", sizeof(uint32_t) * total_words) : nullptr;
    if (!d_idx || !d_start || !d_len || !d_hlen || !d_woff || !d_seg || !d_hpc || (reads->has_n && (!d_segn || !d_hpcn)))
        return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sp_hla_diplotype_gene buffers");
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, sel.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_start, h_start.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_len, h_len.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_woff, h_woff.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, st));
    {
        ProfScope ps(ctx, "hla_segments", n);
        hipLaunchKernelGGL(seg_slice_kernel, dim3((n + SEG_WAVES - 1) / SEG_WAVES), dim3(SEG_WAVES * SP_WAVE), 0, st,
                           reads->view(), d_idx, d_start, d_len, d_woff, (int)n, d_seg, d_segn);
        hipLaunchKernelGGL(hpc_kernel, dim3((n + HPC_WAVES - 1) / HPC_WAVES), dim3(HPC_WAVES * SP_WAVE), 0, st,
                           d_seg, d_segn, d_woff, d_len, (int)n, d_hpc, d_hpcn, d_hlen);
    }
    SP_HIP_CHECK(ctx, hipStreamSynchronize(st));
    SP_HIP_CHECK(ctx, hipGetLastError());
    sp_seqset seg, hpc;
    seg.ctx = ctx; seg.n = n; seg.has_n = reads->has_n; seg.d_words = d_seg; seg.d_nplane = d_segn; seg.d_word_off = d_woff; seg.d_len = d_len; seg.max_len = max_len;
    hpc = seg; hpc.d_words = d_hpc; hpc.d_nplane = d_hpcn; hpc.d_len = d_hlen;

    // ---- run_dual_consensus_with_offsets (caller.rs:1118-1219)
    sp_cons_config cc;
    cc.min_count = cfg->min_consensus_count; cc.min_af = cfg->min_consensus_fraction; cc.dual_max_ed_delta = cfg->dual_max_ed_delta;
    cc.allow_early_termination = 1; cc.allow_dual = 1; cc.offset_window = 400; cc.offset_compare_length = 50;      // dwfa_config_from_cli (:1103-1116)
    const int half_window = cc.offset_window / 2;
    std::vector<int32_t> off(n), sc1(n), sc2(n);
    std::vector<uint8_t> is1(n);
    const uint32_t ccap = (uint32_t)max_len + 512 + (uint32_t)cc.offset_window;
    std::vector<char> t1(ccap + 1), t2(ccap + 1);
    sp_cons_result cres;
    auto offsets_of = [&](bool use_hpc, const uint8_t* group, int which) {
        int32_t mn = INT32_MAX;
        for (uint32_t i = 0; i < n; ++i) if (!group || group[i] == which) mn = std::min(mn, use_hpc ? realign[sel[i]].hpc_offset : realign[sel[i]].dna_offset);
        for (uint32_t i = 0; i < n; ++i) { const int32_t o = use_hpc ? realign[sel[i]].hpc_offset : realign[sel[i]].dna_offset; off[i] = o == mn ? -1 : o - mn + half_window; }
    };
    auto passing = [&](int32_t* c1, int32_t* c2, double* maf, double* cdf) {
        *c1 = 0; for (uint32_t i = 0; i < n; ++i) *c1 += is1[i]; *c2 = (int32_t)n - *c1;
        if (!cres.is_dual) { *maf = 0.0; *cdf = 0.0; return 0; }                     // DualPassingStats::new_non_dual
        return sp_hla_is_passing_dual((uint64_t)*c1, (uint64_t)*c2, cfg->min_consensus_fraction, cfg->expected_maf, cfg->min_cdf, maf, cdf);
    };
    offsets_of(true, nullptr, 0);
    int32_t rc = sp_consensus_dual(ctx, &hpc, nullptr, n, off.data(), &cc, t1.data(), t2.data(), ccap, is1.data(), sc1.data(), sc2.data(), &cres);
    if (rc != SP_OK) return rc;
    int32_t c1 = 0, c2 = 0; double maf = 0, cdf = 0;
    int pass = passing(&c1, &c2, &maf, &cdf);
    call->used_dna_dual = 0;
    if (!pass) {                                                                     // HPC did not separate the reads: full-length DNA (:1180-1218)
        offsets_of(false, nullptr, 0);
        rc = sp_consensus_dual(ctx, &seg, nullptr, n, off.data(), &cc, t1.data(), t2.data(), ccap, is1.data(), sc1.data(), sc2.data(), &cres);
        if (rc != SP_OK) return rc;
        pass = passing(&c1, &c2, &maf, &cdf);
        call->used_dna_dual = 1;
    }
    int is_dual = cres.is_dual;
    // ---- hemizygosity (caller.rs:676-684)
    int hemi = 0;
    if (cfg->absent_capable) {
        std::vector<int64_t> s1(n), s2(n);
        for (uint32_t i = 0; i < n; ++i) { s1[i] = sc1[i]; s2[i] = sc2[i]; }
        double hc = 0, dc = 0;
        hemi = sp_hla_is_hemizygous_better(s1.data(), s2.data(), is1.data(), n, is_dual, (uint64_t)cfg->dual_max_ed_delta, cfg->normalized_coverage, &hc, &dc);
        if (hemi) { is_dual = 0; std::fill(is1.begin(), is1.end(), (uint8_t)1); }   // boiler-plate non-dual consensus (:687-701)
    }
    // ---- one consensus per group on the DNA segments (caller.rs:706-747)
    sp_cons_config single = cc; single.allow_dual = 0;
    std::vector<uint32_t> grp; std::vector<int32_t> goff, gs1, gs2; std::vector<uint8_t> gis;
    auto group_consensus = [&](int which, char* out, int32_t* out_len) -> int32_t {
        grp.clear(); goff.clear();
        offsets_of(false, is1.data(), which);
        for (uint32_t i = 0; i < n; ++i) if (is1[i] == which) { grp.push_back(i); goff.push_back(off[i]); }
        out[0] = '\0'; *out_len = 0;
        if (grp.empty()) return SP_OK;
        gs1.resize(grp.size()); gs2.resize(grp.size()); gis.resize(grp.size());
        sp_cons_result r2;
        const int32_t e = sp_consensus(ctx, &seg, grp.data(), (uint32_t)grp.size(), goff.data(), &single, t1.data(), t2.data(), ccap, gis.data(), gs1.data(), gs2.data(), &r2);
        if (e == SP_ERR_CAPACITY) return SP_OK;                                    // "Failed to generate a consensus" => empty string => unknown (:741-755)
        if (e != SP_OK) return e;
        if ((uint32_t)r2.len1 + 1 > cap) return sp_fail(ctx, SP_ERR_CAPACITY, "sp_hla_diplotype_gene: consensus buffer too small");
        std::memcpy(out, t1.data(), (size_t)r2.len1 + 1); *out_len = r2.len1;
        return SP_OK;
    };
    rc = group_consensus(1, cons1, &call->cons1_len);
    if (rc != SP_OK) return rc;
    sp_hla_best b1; std::memset(&b1, 0, sizeof b1); b1.best_allele = -1;
    rc = sp_hla_type_consensus(ctx, db, gene, cons1, (uint32_t)call->cons1_len, cfg->require_dna, cfg->disable_cdna, &b1, nullptr, nullptr, 0, nullptr);
    if (rc != SP_OK) return rc;
    call->typed1 = b1.best_allele;
    call->is_dual = is_dual; call->is_hemizygous = hemi; call->counts1 = c1; call->counts2 = c2; call->maf = maf; call->cdf = cdf;
    if (is_dual) {
        rc = group_consensus(0, cons2, &call->cons2_len);
        if (rc != SP_OK) return rc;
        sp_hla_best b2; std::memset(&b2, 0, sizeof b2); b2.best_allele = -1;
        rc = sp_hla_type_consensus(ctx, db, gene, cons2, (uint32_t)call->cons2_len, cfg->require_dna, cfg->disable_cdna, &b2, nullptr, nullptr, 0, nullptr);
        if (rc != SP_OK) return rc;
        call->typed2 = b2.best_allele;
        call->dual_passed = pass;
        if (pass) { call->allele1 = b1.best_allele; call->allele2 = b2.best_allele; }                       // heterozygous (:893-895)
        else if (c1 > c2) call->allele1 = call->allele2 = b1.best_allele;                                   // homozygous for the dominant allele (:896-903)
        else call->allele1 = call->allele2 = b2.best_allele;
    } else {
        call->dual_passed = 0;
        call->allele1 = call->allele2 = b1.best_allele;                                                      // :905-912
        if (hemi) call->allele1 = -2;                                                                        // (NO_CALL_HAP, allele) (:919-923)
    }
    if (is_cons1_out) for (uint32_t i = 0; i < n; ++i) is_cons1_out[i] = is1[i];
    return SP_OK;
}

} // extern "C"
