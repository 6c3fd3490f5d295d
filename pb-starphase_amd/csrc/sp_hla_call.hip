// sp_hla_call.hip -- the per-gene HLA solve that ties K1, K8 and K2 together (gfx950).
//
// Host-side counterpart of the gene loop of diplotype_hla_batch (src/hla/caller.rs:642-1040) without its I/O and debug
// artefacts: realigned segments of the gene -> dual consensus on homopolymer-compressed segments, DNA segments if that
// does not pass (run_dual_consensus_with_offsets, :1118-1219) -> optional hemizygosity test (:676-684) -> one consensus per
// read group on the DNA segments (:706-747) -> typing of each consensus (score_consensus, :756,829) -> heterozygous /
// homozygous decision (:889-912).  Segments never leave the GPU: they are cut out of the packed reads and
// homopolymer-compressed by the two kernels below.
#include "sp_internal.h"
#include <thread>
#include <system_error>
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

} // namespace

// segments [start, start + len) of reads[idx] as a packed set in pooled buffers "<prefix>_*" (and, optionally, their
// homopolymer-compressed form): the data never leaves the GPU.  The sets stay valid until the next call with the same prefix.
int sp_make_segments(sp_ctx* ctx, const sp_seqset* reads, const std::vector<uint32_t>& idx, const std::vector<int32_t>& start,
                     const std::vector<int32_t>& len, const char* prefix, sp_seqset* seg, sp_seqset* hpc) {
    const uint32_t n = (uint32_t)idx.size();
    hipStream_t st = ctx->stream;
    std::vector<uint64_t> h_woff((size_t)n + 1, 0);
    int32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (idx[i] >= reads->n || len[i] <= 0 || start[i] < 0 || start[i] + len[i] > reads->h_len[idx[i]]) return sp_fail(ctx, SP_ERR_INVALID_ARG, "segment outside its read");
        if (hpc && len[i] > HPC_MAX) return sp_fail(ctx, SP_ERR_TOO_LONG, "segment longer than 32,768 bases");
        max_len = std::max(max_len, len[i]);
        h_woff[i + 1] = h_woff[i] + (uint64_t)((((len[i] + 15) >> 4) + 2 + 3) & ~3);
    }
    const uint64_t total_words = h_woff[n] + 4;
    const std::string px(prefix);
    auto pool = [&](const char* what, size_t bytes) { return sp_pool(ctx, (px + what).c_str(), std::max<size_t>(bytes, 16)); };
    // idx / start / len / word offsets go up in one copy out of a pinned staging buffer (four pageable copies cost ~75 us each)
    const size_t at_woff = 0, at_idx = (sizeof(uint64_t) * ((size_t)n + 1) + 15) & ~(size_t)15, at_start = at_idx + ((sizeof(uint32_t) * n + 15) & ~(size_t)15),
                 at_len = at_start + ((sizeof(int32_t) * n + 15) & ~(size_t)15), in_bytes = at_len + ((sizeof(int32_t) * n + 15) & ~(size_t)15);
    uint8_t* d_in = (uint8_t*)pool("_in", in_bytes);
    uint8_t* h_in = (uint8_t*)sp_host_pool(ctx, (px + "_in").c_str(), std::max<size_t>(in_bytes, 16));
    int32_t* h_hl = (int32_t*)sp_host_pool(ctx, (px + "_hlen").c_str(), std::max<size_t>(sizeof(int32_t) * n, 16));
    if (!d_in || !h_in || !h_hl) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "segment buffers");
    uint64_t* d_woff = (uint64_t*)(d_in + at_woff); uint32_t* d_idx = (uint32_t*)(d_in + at_idx); int32_t* d_start = (int32_t*)(d_in + at_start); int32_t* d_len = (int32_t*)(d_in + at_len);
    int32_t* d_hlen = hpc ? (int32_t*)pool("_hlen", sizeof(int32_t) * n) : nullptr;
    uint32_t* d_seg = (uint32_t*)pool("_seg", sizeof(uint32_t) * total_words);
    uint32_t* d_hpc = hpc ? (uint32_t*)pool("_hpc", sizeof(uint32_t) * total_words) : nullptr;
    uint32_t* d_segn = reads->has_n ? (uint32_t*)pool("_segn", sizeof(uint32_t) * total_words) : nullptr;
    uint32_t* d_hpcn = (hpc && reads->has_n) ? (uint32_t*)pool("_hpcn", sizeof(uint32_t) * total_words) : nullptr;
    if (!d_idx || !d_start || !d_len || !d_woff || !d_seg || (hpc && (!d_hlen || !d_hpc)) || (reads->has_n && (!d_segn || (hpc && !d_hpcn))))
        return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "segment buffers");
    std::memcpy(h_in + at_woff, h_woff.data(), sizeof(uint64_t) * ((size_t)n + 1));
    if (n) { std::memcpy(h_in + at_idx, idx.data(), sizeof(uint32_t) * n); std::memcpy(h_in + at_start, start.data(), sizeof(int32_t) * n); std::memcpy(h_in + at_len, len.data(), sizeof(int32_t) * n); }
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, st));
    std::vector<int32_t> h_hlen(n, 0);
    if (n) {
        ProfScope ps(ctx, "segments", n);
        hipLaunchKernelGGL(seg_slice_kernel, dim3((n + SEG_WAVES - 1) / SEG_WAVES), dim3(SEG_WAVES * SP_WAVE), 0, st,
                           reads->view(), d_idx, d_start, d_len, d_woff, (int)n, d_seg, d_segn);
        if (hpc) hipLaunchKernelGGL(hpc_kernel, dim3((n + HPC_WAVES - 1) / HPC_WAVES), dim3(HPC_WAVES * SP_WAVE), 0, st,
                                    d_seg, d_segn, d_woff, d_len, (int)n, d_hpc, d_hpcn, d_hlen);
    }
    if (n && hpc) SP_HIP_CHECK(ctx, hipMemcpyAsync(h_hl, d_hlen, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(st));
    SP_HIP_CHECK(ctx, hipGetLastError());
    if (n && hpc) std::memcpy(h_hlen.data(), h_hl, sizeof(int32_t) * n);
    *seg = sp_seqset();
    seg->ctx = ctx; seg->n = n; seg->has_n = reads->has_n; seg->d_words = d_seg; seg->d_nplane = d_segn; seg->d_word_off = d_woff; seg->d_len = d_len; seg->max_len = max_len;
    seg->h_len = len; seg->h_word_off = h_woff;
    if (hpc) {
        *hpc = *seg;
        hpc->d_words = d_hpc; hpc->d_nplane = d_hpcn; hpc->d_len = d_hlen; hpc->h_len = h_hlen;
    }
    return SP_OK;
}

// the solve over units = (sample, gene) pairs; read_sample == nullptr: one sample, units are genes
static int32_t hla_solve_units(sp_ctx* ctx, const sp_hla_db* db, uint32_t n_genes, const uint32_t* genes, const uint32_t* unit_sample,
                               const uint32_t* read_sample, const sp_seqset* reads,
                               const sp_hla_realign* realign, const sp_hla_call_config* cfgs, sp_hla_call* calls,
                               char* cons, uint32_t cap, uint8_t* is_cons1_out, bool clear_is_cons1 = true) {
    if (is_cons1_out && clear_is_cons1) std::memset(is_cons1_out, 0, reads->n);
    HostMarks hm(ctx);
    struct Gene { uint32_t first = 0, n = 0; int is_dual = 0, pass = 0, hemi = 0, used_dna = 0; int32_t c1 = 0, c2 = 0; double maf = 0, cdf = 0; };
    std::vector<Gene> G(n_genes);
    // realigned_records of every gene, in input (qname) order, flattened gene after gene
    std::vector<uint32_t> sel;
    for (uint32_t k = 0; k < n_genes; ++k) {
        sp_hla_call& call = calls[k];
        std::memset(&call, 0, sizeof call);
        call.allele1 = call.allele2 = call.typed1 = call.typed2 = -1;
        cons[(size_t)(2 * k) * cap] = cons[(size_t)(2 * k + 1) * cap] = '\0';
        G[k].first = (uint32_t)sel.size();
        for (uint32_t r = 0; r < reads->n; ++r)
            if (realign[r].status == 0 && realign[r].gene == (int32_t)genes[k] && (!read_sample || read_sample[r] == unit_sample[k])) sel.push_back(r);
        G[k].n = (uint32_t)sel.size() - G[k].first;
        call.n_reads = (int32_t)G[k].n;
        if (G[k].n == 0) call.status = 1;                                          // NO_READS / NO_CALL (caller.rs:662-668)
    }
    const uint32_t n = (uint32_t)sel.size();
    if (n == 0) return SP_OK;
    SP_HIP_CHECK(ctx, hipSetDevice(ctx->device));

    // ---- segments of all genes on the device
    std::vector<int32_t> h_start(n), h_len(n);
    int32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const sp_hla_realign& q = realign[sel[i]];
        h_start[i] = q.seg_start; h_len[i] = q.seg_end - q.seg_start;
        max_len = std::max(max_len, h_len[i]);
    }
    sp_seqset seg, hpc;
    hm.mark("host:hla_select");
    {
        const int32_t e = sp_make_segments(ctx, reads, sel, h_start, h_len, "hc", &seg, &hpc);
        if (e != SP_OK) return e;
    }
    hm.mark("host:hla_segments");

    // ---- run_dual_consensus_with_offsets (caller.rs:1118-1219), all genes in lockstep
    const int half_window = 200;                                                   // offset_window 400 (dwfa_config_from_cli, :1103-1116)
    const uint32_t ccap = (uint32_t)max_len + 512 + 400 + 1;
    std::vector<uint32_t> ident(n);
    for (uint32_t i = 0; i < n; ++i) ident[i] = i;
    std::vector<int32_t> off(n), sc1(n), sc2(n);
    std::vector<uint8_t> is1(n);
    std::vector<char> text((size_t)2 * n_genes * ccap);
    auto cons_config = [&](uint32_t k, int dual) {
        sp_cons_config cc{};
        cc.min_count = cfgs[k].min_consensus_count; cc.min_af = cfgs[k].min_consensus_fraction; cc.dual_max_ed_delta = cfgs[k].dual_max_ed_delta;
        cc.allow_early_termination = 1; cc.allow_dual = dual; cc.offset_window = 400; cc.offset_compare_length = 50;
        return cc;
    };
    // offsets of the members (group < 0: every read of the gene) relative to their minimum (caller.rs:1127-1141,709-735)
    auto set_offsets = [&](uint32_t k, bool use_hpc, int group) {
        int32_t mn = INT32_MAX;
        for (uint32_t i = G[k].first; i < G[k].first + G[k].n; ++i) if (group < 0 || is1[i] == group) mn = std::min(mn, use_hpc ? realign[sel[i]].hpc_offset : realign[sel[i]].dna_offset);
        for (uint32_t i = G[k].first; i < G[k].first + G[k].n; ++i) if (group < 0 || is1[i] == group) {
            const int32_t o = use_hpc ? realign[sel[i]].hpc_offset : realign[sel[i]].dna_offset;
            off[i] = o == mn ? -1 : o - mn + half_window;
        }
    };
    auto dual_batch = [&](const std::vector<uint32_t>& which, bool use_hpc) -> int32_t {
        std::vector<sp_cons_problem> P(which.size()); std::vector<sp_cons_output> O(which.size());
        for (size_t x = 0; x < which.size(); ++x) {
            const uint32_t k = which[x];
            set_offsets(k, use_hpc, -1);
            P[x].reads = use_hpc ? &hpc : &seg; P[x].read_idx = ident.data() + G[k].first; P[x].n = G[k].n; P[x].offsets = off.data() + G[k].first; P[x].cfg = cons_config(k, 1);
            std::memset(&O[x], 0, sizeof O[x]);
            O[x].cons1 = text.data() + (size_t)(2 * k) * ccap; O[x].cons2 = text.data() + (size_t)(2 * k + 1) * ccap; O[x].cap = ccap;
            O[x].is_cons1 = is1.data() + G[k].first; O[x].score1 = sc1.data() + G[k].first; O[x].score2 = sc2.data() + G[k].first;
        }
        const int32_t rc = sp_consensus_dual_batch(ctx, (uint32_t)P.size(), P.data(), O.data());
        if (rc != SP_OK) return rc;
        for (size_t x = 0; x < which.size(); ++x) {
            Gene& g = G[which[x]];
            g.is_dual = O[x].result.is_dual;
            g.c1 = 0; for (uint32_t i = g.first; i < g.first + g.n; ++i) g.c1 += is1[i];
            g.c2 = (int32_t)g.n - g.c1;
            if (!g.is_dual) { g.pass = 0; g.maf = g.cdf = 0.0; }                  // DualPassingStats::new_non_dual
            else g.pass = sp_hla_is_passing_dual((uint64_t)g.c1, (uint64_t)g.c2, cfgs[which[x]].min_consensus_fraction, cfgs[which[x]].expected_maf, cfgs[which[x]].min_cdf, &g.maf, &g.cdf);
        }
        return SP_OK;
    };
    std::vector<uint32_t> todo;
    for (uint32_t k = 0; k < n_genes; ++k) if (G[k].n) todo.push_back(k);
    hm.mark("host:hla_setup");
    int32_t rc = dual_batch(todo, true);
    if (rc != SP_OK) return rc;
    hm.mark("host:hla_dual_hpc");
    std::vector<uint32_t> retry;
    for (uint32_t k : todo) if (!G[k].pass) retry.push_back(k);                    // HPC did not separate the reads: full-length DNA (:1180-1218)
    if (!retry.empty()) {
        rc = dual_batch(retry, false);
        if (rc != SP_OK) return rc;
        for (uint32_t k : retry) G[k].used_dna = 1;
    }
    hm.mark("host:hla_dual_dna");
    // ---- hemizygosity (caller.rs:676-684)
    for (uint32_t k : todo) if (cfgs[k].absent_capable) {
        Gene& g = G[k];
        std::vector<int64_t> s1(g.n), s2(g.n);
        for (uint32_t i = 0; i < g.n; ++i) { s1[i] = sc1[g.first + i]; s2[i] = sc2[g.first + i]; }
        double hc = 0, dc = 0;
        g.hemi = sp_hla_is_hemizygous_better(s1.data(), s2.data(), is1.data() + g.first, g.n, g.is_dual, (uint64_t)cfgs[k].dual_max_ed_delta, cfgs[k].normalized_coverage, &hc, &dc);
        if (g.hemi) { g.is_dual = 0; std::fill(is1.begin() + g.first, is1.begin() + g.first + g.n, (uint8_t)1); }   // boiler-plate non-dual consensus (:687-701)
    }
    // ---- one consensus per read group on the DNA segments (caller.rs:706-747), every group of every gene in lockstep
    struct Grp { uint32_t k; int which; size_t at, n; };
    std::vector<Grp> groups; std::vector<uint32_t> gidx; std::vector<int32_t> goff;
    for (uint32_t k : todo) for (int which = 1; which >= (G[k].is_dual ? 0 : 1); --which) {
        set_offsets(k, false, which);
        Grp g{ k, which, gidx.size(), 0 };
        for (uint32_t i = G[k].first; i < G[k].first + G[k].n; ++i) if (is1[i] == which) { gidx.push_back(i); goff.push_back(off[i]); }
        g.n = gidx.size() - g.at;
        if (g.n) groups.push_back(g);
    }
    {
        std::vector<sp_cons_problem> P(groups.size()); std::vector<sp_cons_output> O(groups.size());
        std::vector<uint8_t> gis(gidx.size()); std::vector<int32_t> gs1(gidx.size()), gs2(gidx.size());
        std::vector<char> spare((size_t)groups.size() * ccap);
        for (size_t x = 0; x < groups.size(); ++x) {
            const Grp& g = groups[x];
            P[x].reads = &seg; P[x].read_idx = gidx.data() + g.at; P[x].n = (uint32_t)g.n; P[x].offsets = goff.data() + g.at; P[x].cfg = cons_config(g.k, 0);
            std::memset(&O[x], 0, sizeof O[x]);
            O[x].cons1 = text.data() + (size_t)(2 * g.k + (g.which ? 0 : 1)) * ccap; O[x].cons2 = spare.data() + x * ccap; O[x].cap = ccap;
            O[x].is_cons1 = gis.data() + g.at; O[x].score1 = gs1.data() + g.at; O[x].score2 = gs2.data() + g.at;
        }
        for (uint32_t k : todo) text[(size_t)(2 * k) * ccap] = text[(size_t)(2 * k + 1) * ccap] = '\0';
        rc = sp_consensus_batch(ctx, (uint32_t)P.size(), P.data(), O.data());
        if (rc != SP_OK && rc != SP_ERR_CAPACITY) return rc;
        for (size_t x = 0; x < groups.size(); ++x) if (O[x].status != SP_OK) O[x].cons1[0] = '\0';   // "Failed to generate a consensus" => empty => unknown (:741-755)
    }
    hm.mark("host:hla_groups");
    // ---- typing (score_consensus, caller.rs:756,829): every consensus of every unit in one batch
    std::vector<uint32_t> tg; std::vector<const char*> tp; std::vector<uint32_t> tl; std::vector<std::pair<uint32_t, int>> tw;
    for (uint32_t k : todo) {
        const Gene& g = G[k];
        const char* t1 = text.data() + (size_t)(2 * k) * ccap; const char* t2 = text.data() + (size_t)(2 * k + 1) * ccap;
        const size_t l1 = std::strlen(t1), l2 = g.is_dual ? std::strlen(t2) : 0;
        if (l1 + 1 > cap || l2 + 1 > cap) return sp_fail(ctx, SP_ERR_CAPACITY, "sp_hla_diplotype_genes: consensus buffer too small");
        std::memcpy(cons + (size_t)(2 * k) * cap, t1, l1 + 1); calls[k].cons1_len = (int32_t)l1;
        tg.push_back(genes[k]); tp.push_back(cons + (size_t)(2 * k) * cap); tl.push_back((uint32_t)l1); tw.push_back({ k, 0 });
        if (g.is_dual) {
            std::memcpy(cons + (size_t)(2 * k + 1) * cap, t2, l2 + 1); calls[k].cons2_len = (int32_t)l2;
            tg.push_back(genes[k]); tp.push_back(cons + (size_t)(2 * k + 1) * cap); tl.push_back((uint32_t)l2); tw.push_back({ k, 1 });
        }
    }
    // one batch per (require_dna, disable_cdna) setting in use (normally one: they are run-wide switches of the CLI)
    for (int combo = 0; combo < 4; ++combo) {
        const int rd = combo & 1, dc = combo >> 1;
        std::vector<uint32_t> g2; std::vector<const char*> p2; std::vector<uint32_t> l2v; std::vector<size_t> at;
        for (size_t x = 0; x < tw.size(); ++x) if ((cfgs[tw[x].first].require_dna != 0) == rd && (cfgs[tw[x].first].disable_cdna != 0) == dc) { g2.push_back(tg[x]); p2.push_back(tp[x]); l2v.push_back(tl[x]); at.push_back(x); }
        if (g2.empty()) continue;
        std::vector<sp_hla_best> tb(g2.size());
        rc = sp_hla_type_consensus_batch(ctx, db, (uint32_t)g2.size(), g2.data(), p2.data(), l2v.data(), rd, dc, tb.data());
        if (rc != SP_OK) return rc;
        for (size_t y = 0; y < at.size(); ++y) (tw[at[y]].second ? calls[tw[at[y]].first].typed2 : calls[tw[at[y]].first].typed1) = tb[y].best_allele;
    }
    hm.mark("host:hla_typing");
    // ---- the call (caller.rs:889-923)
    for (uint32_t k : todo) {
        Gene& g = G[k]; sp_hla_call& call = calls[k];
        call.is_dual = g.is_dual; call.is_hemizygous = g.hemi; call.counts1 = g.c1; call.counts2 = g.c2; call.maf = g.maf; call.cdf = g.cdf; call.used_dna_dual = g.used_dna;
        if (g.is_dual) {
            call.dual_passed = g.pass;
            if (g.pass) { call.allele1 = call.typed1; call.allele2 = call.typed2; }                            // heterozygous (:893-895)
            else if (g.c1 > g.c2) call.allele1 = call.allele2 = call.typed1;                                    // homozygous for the dominant allele (:896-903)
            else call.allele1 = call.allele2 = call.typed2;
        } else {
            call.dual_passed = 0;
            call.allele1 = call.allele2 = call.typed1;                                                          // :905-912
            if (g.hemi) call.allele1 = -2;                                                                       // (NO_CALL_HAP, allele) (:919-923)
        }
        if (is_cons1_out) for (uint32_t i = g.first; i < g.first + g.n; ++i) is_cons1_out[sel[i]] = is1[i];
    }
    return SP_OK;
}

#ifndef SP_HLA_SPLIT_MIN_READS
#define SP_HLA_SPLIT_MIN_READS 1000     // realigned reads of a sample from which its genes are solved side by side on two streams
#endif

// The units (genes of a sample, or (sample, gene) pairs of a cohort) are independent of each other, and each is a chain of launches that
// wait for one another (consensus windows, typing levels) without filling the device: with enough reads to make it worth a thread, every
// other unit runs on the context's helper (its own stream, pools and events) beside the rest.  The calls are the calls of the single run:
// nothing of one unit's solve reads anything of another's.
static int32_t hla_solve_side_by_side(sp_ctx* ctx, const sp_hla_db* db, uint32_t n_units, const uint32_t* genes, const uint32_t* unit_sample,
                                      const uint32_t* read_sample, const sp_seqset* reads, const sp_hla_realign* realign, const sp_hla_call_config* cfgs,
                                      sp_hla_call* calls, char* cons, uint32_t cap, uint8_t* is_cons1_out) {
    HostScope whole(ctx, "host:hla_genes_total");
    uint64_t n_realigned = 0;
    for (uint32_t r = 0; r < reads->n; ++r) n_realigned += realign[r].status == 0;
    int n_parts = (ctx->split_genes && n_realigned >= SP_HLA_SPLIT_MIN_READS) ? (int)std::min<uint32_t>(n_units, (uint32_t)ctx->split_streams) : 1;
    sp_ctx* on[4] = { ctx, nullptr, nullptr, nullptr };
    for (int x = 1; x < n_parts; ++x) { on[x] = sp_ctx_helper(ctx, x - 1); if (!on[x]) { n_parts = x; break; } }
    if (n_parts < 2) return hla_solve_units(ctx, db, n_units, genes, unit_sample, read_sample, reads, realign, cfgs, calls, cons, cap, is_cons1_out);
    if (is_cons1_out) std::memset(is_cons1_out, 0, reads->n);
    struct Part { std::vector<uint32_t> at, genes, samples; std::vector<sp_hla_call_config> cfgs; std::vector<sp_hla_call> calls; std::vector<char> cons; int32_t rc = SP_OK; };
    Part part[4];
    for (uint32_t k = 0; k < n_units; ++k) { Part& q = part[k % (uint32_t)n_parts]; q.at.push_back(k); q.genes.push_back(genes[k]); if (unit_sample) q.samples.push_back(unit_sample[k]); q.cfgs.push_back(cfgs[k]); }
    for (int x = 0; x < n_parts; ++x) { part[x].calls.resize(part[x].at.size()); part[x].cons.assign((size_t)2 * part[x].at.size() * cap, '\0'); }
    auto solve = [&](sp_ctx* c, Part& q) {                             // (never lets an exception out: a thread that is still joinable when the stack unwinds ends the process)
        (void)hipSetDevice(c->device);                                 // (a helper thread starts on device 0)
        try {
            q.rc = hla_solve_units(c, db, (uint32_t)q.at.size(), q.genes.data(), unit_sample ? q.samples.data() : nullptr, read_sample, reads, realign, q.cfgs.data(), q.calls.data(),
                                   q.cons.data(), cap, is_cons1_out, false);
        } catch (const std::bad_alloc&) { q.rc = SP_ERR_OUT_OF_MEMORY; c->err = "sp_hla_diplotype: out of host memory"; }
        catch (const std::exception& e) { q.rc = SP_ERR_INVALID_ARG; c->err = std::string("sp_hla_diplotype: ") + e.what(); }
    };
    HostMarks hm(ctx);
    std::thread beside[4]; bool started[4] = { false, false, false, false };
    for (int x = 1; x < n_parts; ++x) {
        try { beside[x] = std::thread([&, x]() { solve(on[x], part[x]); }); started[x] = true; }
        catch (const std::system_error&) { }                               // no thread to be had: this one does that part too
    }
    hm.mark("host:hla_split_spawn");
    solve(ctx, part[0]);
    hm.mark("host:hla_split_own");
    for (int x = 1; x < n_parts; ++x) { if (started[x]) beside[x].join(); else solve(ctx, part[x]); }   // (the helpers' timings are added to this context's when somebody asks: sp_profile_get)
    hm.mark("host:hla_split_join");
    int32_t rc = SP_OK;
    for (int x = 0; x < n_parts; ++x) {
        Part& q = part[x];
        for (size_t y = 0; y < q.at.size(); ++y) {
            calls[q.at[y]] = q.calls[y];
            std::memcpy(cons + (size_t)(2 * q.at[y]) * cap, q.cons.data() + (size_t)(2 * y) * cap, (size_t)2 * cap);
        }
        if (q.rc != SP_OK && rc == SP_OK) { rc = q.rc; if (x > 0 && started[x]) ctx->err = on[x]->err; }
    }
    return rc;
}

extern "C" {

int32_t sp_hla_diplotype_genes(sp_ctx* ctx, const sp_hla_db* db, uint32_t n_genes, const uint32_t* genes, const sp_seqset* reads,
                               const sp_hla_realign* realign, const sp_hla_call_config* cfgs, sp_hla_call* calls,
                               char* cons, uint32_t cap, uint8_t* is_cons1_out) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!db || !reads || !realign || !cfgs || !calls || !cons || !genes || cap == 0) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_hla_diplotype_genes: null argument");
    return hla_solve_side_by_side(ctx, db, n_genes, genes, nullptr, nullptr, reads, realign, cfgs, calls, cons, cap, is_cons1_out);
}

int32_t sp_hla_diplotype_cohort(sp_ctx* ctx, const sp_hla_db* db, uint32_t n_samples, const uint32_t* read_sample, uint32_t n_genes, const uint32_t* genes,
                                const sp_seqset* reads, const sp_hla_realign* realign, const sp_hla_call_config* cfgs, sp_hla_call* calls,
                                char* cons, uint32_t cap, uint8_t* is_cons1_out) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!db || !reads || !realign || !cfgs || !calls || !cons || !genes || !read_sample || cap == 0) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_hla_diplotype_cohort: null argument");
    for (uint32_t r = 0; r < reads->n; ++r) if (read_sample[r] >= n_samples) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_hla_diplotype_cohort: sample index out of range");
    const uint32_t units = n_samples * n_genes;
    std::vector<uint32_t> ug(units), us(units); std::vector<sp_hla_call_config> uc(units);
    for (uint32_t s = 0; s < n_samples; ++s) for (uint32_t g = 0; g < n_genes; ++g) { ug[s * n_genes + g] = genes[g]; us[s * n_genes + g] = s; uc[s * n_genes + g] = cfgs[g]; }
    return hla_solve_side_by_side(ctx, db, units, ug.data(), us.data(), read_sample, reads, realign, uc.data(), calls, cons, cap, is_cons1_out);
}

int32_t sp_hla_diplotype_gene(sp_ctx* ctx, const sp_hla_db* db, uint32_t gene, const sp_seqset* reads, const sp_hla_realign* realign,
                              const sp_hla_call_config* cfg, sp_hla_call* call, char* cons1, char* cons2, uint32_t cap, uint8_t* is_cons1_out) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!cons1 || !cons2 || cap == 0) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_hla_diplotype_gene: null argument");
    std::vector<char> both((size_t)2 * cap);
    const int32_t rc = sp_hla_diplotype_genes(ctx, db, 1, &gene, reads, realign, cfg, call, both.data(), cap, is_cons1_out);
    if (rc != SP_OK) { cons1[0] = cons2[0] = '\0'; return rc; }
    std::memcpy(cons1, both.data(), std::strlen(both.data()) + 1);
    std::memcpy(cons2, both.data() + cap, std::strlen(both.data() + cap) + 1);
    return SP_OK;
}

} // extern "C"
