// sp_variant.hip -- K6: variant-gene diplotype search on gfx950 (src/diplotyper.rs:1211-1509,
// src/data_types/normalized_variant.rs:431-479).  Pure integer set logic: one thread per
// (het assignment, haplotype side, database haplotype) cell, scores packed into one u64 so the lexicographic
// (core missing, core extra, sub missing, sub extra) order is an integer order.
#include "sp_internal.h"
#include <system_error>
#include <thread>
#include <atomic>
#include <algorithm>
#include <climits>
#include <cstring>
#include <string>
#include <vector>

#define K6_MAX_OBS   64
#define K6_MAX_SLOTS 64
#define K6_NOKEY     0xFFFFFFFFFFFFFFFFull

// quant_match + core/sub split for one (observed subset, haplotype) cell.
//   match_slot[h][o] : first slot of haplotype h that lists observed variant o, or -1   (normalized_variant.rs:443-447)
// The problems of a batch (a panel, one GPU's share of a cohort: 18-23 genes x 32 samples) are ONE launch: their tables lie back to back, a
// thread finds its problem by a binary search over the problems' first cells.
struct K6Desc { int n_haps, n_obs, n_sides, pad; unsigned long long n_comb, cell_base; uint32_t off_ms, off_h, off_o, pad2; };

__global__ __launch_bounds__(256) void k6_cells_kernel(uint32_t n_prob, unsigned long long total, const K6Desc* __restrict__ desc,
                                                       const signed char* __restrict__ match_slot_all, const unsigned long long* __restrict__ slot_need_all,
                                                       const unsigned long long* __restrict__ slot_core_all, const unsigned char* __restrict__ hap_skip_all,
                                                       const unsigned char* __restrict__ obs_core_all, const unsigned char* __restrict__ obs_het_all,
                                                       const unsigned char* __restrict__ obs_group_all, const unsigned char* __restrict__ obs_orient_all,
                                                       unsigned long long* __restrict__ keys /* per problem [comb][side][hap], problems back to back */) {
    const unsigned long long gcell = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gcell >= total) return;
    uint32_t lo = 0, hi = n_prob;                                    // last problem whose first cell is <= gcell
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (desc[mid].cell_base <= gcell) lo = mid; else hi = mid; }
    const K6Desc D = desc[lo];
    const int n_haps = D.n_haps, n_obs = D.n_obs, n_sides = D.n_sides;
    const signed char* match_slot = match_slot_all + D.off_ms;
    const unsigned long long* slot_need = slot_need_all + D.off_h; const unsigned long long* slot_core = slot_core_all + D.off_h;
    const unsigned char* hap_skip = hap_skip_all + D.off_h;
    const unsigned char* obs_core = obs_core_all + D.off_o; const unsigned char* obs_het = obs_het_all + D.off_o;
    const unsigned char* obs_group = obs_group_all + D.off_o; const unsigned char* obs_orient01 = obs_orient_all + D.off_o;
    const unsigned long long cell = gcell - D.cell_base;
    const int h = (int)(cell % (unsigned long long)n_haps);
    const unsigned long long cs = cell / (unsigned long long)n_haps;
    const int side = (int)(cs % (unsigned long long)n_sides);
    const unsigned long long comb = cs / (unsigned long long)n_sides;
    if (hap_skip[h]) { keys[gcell] = K6_NOKEY; return; }                      // SV haplotypes are not quantified (:1443-1446)
    unsigned long long matched = 0;
    unsigned ev_core = 0, ev_sub = 0;
    // a side's list is the homozygous variants first (base_haplotype, diplotyper.rs:1219-1222,1269-1270) and then the heterozygous ones that
    // go to it, each in call order: when two of them compete for one slot (alternatives of an OR-group), the one that comes first takes it
    for (int pass = 0; pass < 2; ++pass)
        for (int o = 0; o < n_obs; ++o) {
            if ((obs_het[o] != 0) != (pass == 1)) continue;
            bool in_set = true;
            if (obs_het[o]) {                                                    // het assignment (diplotyper.rs:1270-1317)
                const bool is_h1 = ((comb >> obs_group[o]) & 1ull) != 0;
                const bool to_h1 = is_h1 == (obs_orient01[o] != 0);
                in_set = side == 0 ? to_h1 : !to_h1;
            }
            if (!in_set) continue;
            const int mi = match_slot[(size_t)h * n_obs + o];
            if (mi >= 0 && !((matched >> mi) & 1ull)) matched |= 1ull << mi;
            else { if (obs_core[o]) ++ev_core; else ++ev_sub; }
        }
    const unsigned long long missing = slot_need[h] & ~matched;                   // unmatched slots without a None alternative
    const unsigned mv_core = (unsigned)__popcll(missing & slot_core[h]), mv_sub = (unsigned)__popcll(missing & ~slot_core[h]);
    keys[gcell] = ((unsigned long long)mv_core << 48) | ((unsigned long long)ev_core << 32) | ((unsigned long long)mv_sub << 16) | (unsigned long long)ev_sub;
}

namespace {

// the host tables of one problem (what the kernel reads, and what the combination of the sides needs)
struct K6Prep {
    int H = 0, NO = 0, n_sides = 1, n_het = 0;
    unsigned long long n_comb = 1, n_cells = 0;
    std::vector<uint8_t> obs_het, obs_group, obs_orient, obs_core, hap_skip;
    std::vector<int8_t> match_slot;
    std::vector<unsigned long long> slot_need, slot_core;
};

int32_t k6_prep(sp_ctx* ctx, const sp_variant_problem* p, K6Prep& q) {
    const int H = p->n_haps, NO = p->n_obs;
    q.H = H; q.NO = NO;
    if (NO > K6_MAX_OBS) return sp_fail(ctx, SP_ERR_TOO_LONG, "variant solve: more than 64 observed variants");
    // which observed variants are hom / het, het groups in first-seen order (diplotyper.rs:1216-1240,1270-1300)
    q.obs_het.assign(std::max(1, NO), 0); q.obs_group.assign(std::max(1, NO), 0); q.obs_orient.assign(std::max(1, NO), 1); q.obs_core.assign(std::max(1, NO), 1);
    std::vector<std::pair<int64_t, int>> ps_bit;             // phase set -> bit of the het assignment, in first-seen order
    int groups = 0, n_het = 0;
    for (int o = 0; o < NO; ++o) {
        if (p->obs_var[o] < 0 || p->obs_var[o] >= p->n_vars) return sp_fail(ctx, SP_ERR_INVALID_ARG, "variant solve: variant id out of range");
        q.obs_core[o] = p->var_is_core[p->obs_var[o]];
        const int gt = p->obs_gt[o];
        if (gt == SP_GT_HOM_REF) return sp_fail(ctx, SP_ERR_INVALID_ARG, "variant solve: homozygous reference calls must not be passed");
        if (gt == SP_GT_HOM_ALT) continue;
        q.obs_het[o] = 1; ++n_het; q.obs_orient[o] = gt != SP_GT_HET_FLIP;
        int bit = -1;
        if (p->obs_ps[o] >= 0) {
            for (auto& kv : ps_bit) if (kv.first == p->obs_ps[o]) bit = kv.second;
            if (bit < 0) { bit = groups++; ps_bit.emplace_back(p->obs_ps[o], bit); }
        } else bit = groups++;                                  // an unphased het is its own group
        q.obs_group[o] = (uint8_t)bit;
    }
    if (groups > 24) return sp_fail(ctx, SP_ERR_TOO_LONG, "variant solve: more than 24 independent heterozygous groups");
    q.n_het = n_het;
    q.n_comb = n_het ? (1ull << (groups - 1)) : 1ull;
    q.n_sides = n_het ? 2 : 1;
    // per haplotype tables
    q.match_slot.assign((size_t)std::max(1, H) * std::max(1, NO), -1);
    q.slot_need.assign(std::max(1, H), 0); q.slot_core.assign(std::max(1, H), 0); q.hap_skip.assign(std::max(1, H), 0);
    for (int h = 0; h < H; ++h) {
        q.hap_skip[h] = p->hap_is_sv[h] ? 1 : 0;
        const int s0 = p->slot_off[h], s1 = p->slot_off[h + 1];
        if (s1 - s0 > K6_MAX_SLOTS) return sp_fail(ctx, SP_ERR_TOO_LONG, "variant solve: haplotype with more than 64 variants");
        for (int s = s0; s < s1; ++s) {
            bool has_none = false; int first_some = -1;
            for (int x = p->alt_off[s]; x < p->alt_off[s + 1]; ++x) { if (p->alt_var[x] < 0) has_none = true; else if (first_some < 0) first_some = p->alt_var[x]; }
            if (!has_none) q.slot_need[h] |= 1ull << (s - s0);
            if (first_some >= 0 && p->var_is_core[first_some]) q.slot_core[h] |= 1ull << (s - s0);
        }
        for (int o = 0; o < NO; ++o)
            for (int s = s0; s < s1 && q.match_slot[(size_t)h * NO + o] < 0; ++s)
                for (int x = p->alt_off[s]; x < p->alt_off[s + 1]; ++x) if (p->alt_var[x] == p->obs_var[o]) { q.match_slot[(size_t)h * NO + o] = (int8_t)(s - s0); break; }
    }
    q.n_cells = q.n_comb * (unsigned long long)q.n_sides * (unsigned long long)std::max(1, H);
    return SP_OK;
}

// per side: best tuple under the (1, MAX, MAX, MAX) bound, ties, sub-allele shadowing; then the sides of every het assignment (diplotyper.rs:1246-1371,1433-1509)
void k6_combine(const sp_variant_problem* p, const K6Prep& q, const unsigned long long* keys, sp_variant_result* res) {
    const int H = q.H, NO = q.NO, n_sides = q.n_sides;
    // SV short-circuit is decided per side on the host (diplotyper.rs:1414-1431): collect labels of a side
    auto side_members = [&](unsigned long long comb, int side, std::vector<int>& out) {
        out.clear();
        for (int pass = 0; pass < 2; ++pass)                  // the homozygous variants first, then the side's heterozygous ones (diplotyper.rs:1269-1317)
            for (int o = 0; o < NO; ++o) {
                if ((q.obs_het[o] != 0) != (pass == 1)) continue;
                bool in_set = true;
                if (q.obs_het[o]) { const bool is_h1 = ((comb >> q.obs_group[o]) & 1ull) != 0; const bool to_h1 = is_h1 == (q.obs_orient[o] != 0); in_set = side == 0 ? to_h1 : !to_h1; }
                if (in_set) out.push_back(o);
            }
    };
    struct Side { int64_t score[4]; bool is_sv; int sv_label; std::vector<int> best; };
    auto solve_side = [&](unsigned long long comb, int side, Side& out) {
        std::vector<int> mem; side_members(comb, side, mem);
        out.best.clear(); out.is_sv = false; out.sv_label = -1;
        std::vector<int> labels;
        for (int o : mem) if (p->obs_sv_label[o] >= 0) labels.push_back(p->obs_sv_label[o]);
        if (!labels.empty()) {
            std::vector<int> rest(labels.begin() + 1, labels.end());
            std::sort(rest.begin(), rest.end()); rest.erase(std::unique(rest.begin(), rest.end()), rest.end());
            out.is_sv = true; out.sv_label = labels[0];
            out.score[0] = 0; out.score[1] = (int64_t)rest.size(); out.score[2] = 0; out.score[3] = 0;
            return;
        }
        unsigned long long best = K6_NOKEY;
        const unsigned long long* k = keys + (comb * (unsigned long long)n_sides + (unsigned long long)side) * (unsigned long long)std::max(1, H);
        for (int h = 0; h < H; ++h) if (k[h] != K6_NOKEY && (k[h] >> 48) <= 1 && k[h] < best) best = k[h];
        if (best == K6_NOKEY) { out.score[0] = 1; out.score[1] = out.score[2] = out.score[3] = INT64_MAX; return; }
        bool any_sub = false;
        for (int h = 0; h < H; ++h) if (k[h] == best && !p->hap_is_core[h]) any_sub = true;
        for (int h = 0; h < H; ++h) if (k[h] == best && (!p->hap_is_core[h]) == any_sub) out.best.push_back(h);
        out.score[0] = (int64_t)(best >> 48); out.score[1] = (int64_t)((best >> 32) & 0xFFFF); out.score[2] = (int64_t)((best >> 16) & 0xFFFF); out.score[3] = (int64_t)(best & 0xFFFF);
    };
    auto push_pairs = [&](const Side& a, const Side& b, int comb) {
        const size_t na = a.is_sv ? 1 : a.best.size(), nb = b.is_sv ? 1 : b.best.size();
        for (size_t i = 0; i < na; ++i) for (size_t j = 0; j < nb; ++j) {
            if (res->n_dip >= SP_VAR_MAXDIP) { res->overflow = 1; return; }
            res->dip[res->n_dip][0] = a.is_sv ? -(a.sv_label + 2) : a.best[i];
            res->dip[res->n_dip][1] = b.is_sv ? -(b.sv_label + 2) : b.best[j];
            res->dip_comb[res->n_dip] = comb; res->n_dip++;
        }
    };
    if (!q.n_het) {
        Side s; solve_side(0, 0, s);
        for (int k = 0; k < 4; ++k) res->score[k] = s.score[k];
        push_pairs(s, s, 0);
        // a homozygous call pairs every haplotype with itself only (diplotyper.rs:1246-1256)
        int w = 0; for (int i = 0; i < res->n_dip; ++i) if (res->dip[i][0] == res->dip[i][1]) { res->dip[w][0] = res->dip[i][0]; res->dip[w][1] = res->dip[i][1]; res->dip_comb[w] = 0; ++w; }
        res->n_dip = w;
        return;
    }
    int64_t best[4] = {INT64_MAX, INT64_MAX, INT64_MAX, INT64_MAX};
    for (unsigned long long comb = 0; comb < q.n_comb; ++comb) {
        Side a, b; solve_side(comb, 0, a); solve_side(comb, 1, b);
        int64_t tot[4]; int cmp = 0;
        for (int k = 0; k < 4; ++k) tot[k] = (a.score[k] == INT64_MAX || b.score[k] == INT64_MAX) ? INT64_MAX : a.score[k] + b.score[k];
        for (int k = 0; k < 4 && !cmp; ++k) cmp = tot[k] < best[k] ? -1 : (tot[k] > best[k]);
        if (cmp < 0) { std::memcpy(best, tot, sizeof(best)); res->n_dip = 0; }
        if (cmp <= 0) push_pairs(a, b, (int)comb);
    }
    for (int k = 0; k < 4; ++k) res->score[k] = best[k];
}

template <class T> void append(std::vector<uint8_t>& blob, const std::vector<T>& v) { const uint8_t* p = reinterpret_cast<const uint8_t*>(v.data()); blob.insert(blob.end(), p, p + v.size() * sizeof(T)); }

} // namespace

// The solves of a panel or of one GPU's share of a cohort (23 genes x 32 samples) in one launch: the tables of all problems go up in one
// copy, every (het assignment, side, haplotype) cell of every problem is one thread of one kernel, the keys come back in one copy and the
// sides are combined on the host (round 2 ran one launch + one wait per problem on three streams: ~1 ms each).  results[i] / problem_rc[i]
// are what sp_variant_solve gives for problems[i]; a problem that fails its checks is reported in problem_rc and takes no part in the launch.
extern "C" int32_t sp_variant_solve_batch(sp_ctx* ctx, uint32_t n, const sp_variant_problem* const* problems, sp_variant_result* results, int32_t* problem_rc) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (n && (!problems || !results)) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_variant_solve_batch: null argument");
    for (uint32_t i = 0; i < n; ++i) if (!problems[i]) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_variant_solve_batch: null problem");
    if (n == 0) return SP_OK;
    (void)hipSetDevice(ctx->device);
    std::vector<K6Prep> prep(n);
    std::vector<int32_t> rcs(n, SP_OK);
    std::string first_err; int32_t rc = SP_OK;
    std::vector<K6Desc> desc; std::vector<uint32_t> which;
    std::vector<uint8_t> ms, need, core, skip, oc, oh, og, oo;
    unsigned long long total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        std::memset(&results[i], 0, sizeof(results[i]));
        rcs[i] = k6_prep(ctx, problems[i], prep[i]);
        if (rcs[i] != SP_OK) { if (rc == SP_OK) { rc = rcs[i]; first_err = ctx->err; } continue; }
        const K6Prep& q = prep[i];
        if (q.H <= 0) continue;                                   // no haplotypes: every key stays unset, the combination runs on the host alone
        K6Desc d; std::memset(&d, 0, sizeof d);
        d.n_haps = q.H; d.n_obs = q.NO; d.n_sides = q.n_sides; d.n_comb = q.n_comb; d.cell_base = total;
        d.off_ms = (uint32_t)ms.size(); d.off_h = (uint32_t)skip.size(); d.off_o = (uint32_t)oc.size();
        append(ms, q.match_slot); append(need, q.slot_need); append(core, q.slot_core); append(skip, q.hap_skip);
        append(oc, q.obs_core); append(oh, q.obs_het); append(og, q.obs_group); append(oo, q.obs_orient);
        desc.push_back(d); which.push_back(i);
        total += q.n_cells;
    }
    std::vector<unsigned long long> keys;
    if (total > 0) {
        // one staging blob up (pinned), one key array down
        size_t bytes = 0;
        auto place = [&](size_t b) { const size_t at = (bytes + 15) & ~(size_t)15; bytes = at + b; return at; };
        const size_t a_desc = place(desc.size() * sizeof(K6Desc)), a_ms = place(ms.size()), a_need = place(need.size()), a_core = place(core.size()), a_skip = place(skip.size());
        const size_t a_oc = place(oc.size()), a_oh = place(oh.size()), a_og = place(og.size()), a_oo = place(oo.size());
        uint8_t* h_in = (uint8_t*)sp_host_pool(ctx, "k6_in", bytes + 16); uint8_t* d_in = (uint8_t*)sp_pool(ctx, "k6_in", bytes + 16);
        unsigned long long* d_keys = (unsigned long long*)sp_pool(ctx, "k6_keys", total * 8);
        unsigned long long* h_keys = (unsigned long long*)sp_host_pool(ctx, "k6_keys", total * 8);
        if (!h_in || !d_in || !d_keys || !h_keys) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "variant solve buffers");
        std::memcpy(h_in + a_desc, desc.data(), desc.size() * sizeof(K6Desc));
        std::memcpy(h_in + a_ms, ms.data(), ms.size()); std::memcpy(h_in + a_need, need.data(), need.size()); std::memcpy(h_in + a_core, core.data(), core.size());
        std::memcpy(h_in + a_skip, skip.data(), skip.size()); std::memcpy(h_in + a_oc, oc.data(), oc.size()); std::memcpy(h_in + a_oh, oh.data(), oh.size());
        std::memcpy(h_in + a_og, og.data(), og.size()); std::memcpy(h_in + a_oo, oo.data(), oo.size());
        SP_HIP_CHECK(ctx, hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, ctx->stream));
        {
            ProfScope ps(ctx, "k6_cells", total);
            hipLaunchKernelGGL(k6_cells_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)desc.size(), total,
                               (const K6Desc*)(d_in + a_desc), (const signed char*)(d_in + a_ms), (const unsigned long long*)(d_in + a_need),
                               (const unsigned long long*)(d_in + a_core), (const unsigned char*)(d_in + a_skip), (const unsigned char*)(d_in + a_oc),
                               (const unsigned char*)(d_in + a_oh), (const unsigned char*)(d_in + a_og), (const unsigned char*)(d_in + a_oo), d_keys);
        }
        SP_HIP_CHECK(ctx, hipGetLastError());
        SP_HIP_CHECK(ctx, hipMemcpyAsync(h_keys, d_keys, total * 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        keys.assign(h_keys, h_keys + total);
    }
    std::vector<unsigned long long> none;
    size_t at = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (rcs[i] != SP_OK) continue;
        const K6Prep& q = prep[i];
        if (q.H <= 0) { none.assign((size_t)q.n_cells, K6_NOKEY); k6_combine(problems[i], q, none.data(), &results[i]); continue; }
        k6_combine(problems[i], q, keys.data() + desc[at].cell_base, &results[i]);
        ++at;
    }
    if (problem_rc) for (uint32_t i = 0; i < n; ++i) problem_rc[i] = rcs[i];
    if (rc != SP_OK) ctx->err = first_err;
    return rc;
}

extern "C" int32_t sp_variant_solve(sp_ctx* ctx, const sp_variant_problem* p, sp_variant_result* res) {
    if (!ctx || !p || !res) return SP_ERR_INVALID_ARG;
    return sp_variant_solve_batch(ctx, 1, &p, res, nullptr);
}
