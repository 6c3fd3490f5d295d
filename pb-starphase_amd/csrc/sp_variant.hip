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
#include <vector>

#define K6_MAX_OBS   64
#define K6_MAX_SLOTS 64
#define K6_NOKEY     0xFFFFFFFFFFFFFFFFull

// quant_match + core/sub split for one (observed subset, haplotype) cell.
//   match_slot[h][o] : first slot of haplotype h that lists observed variant o, or -1   (normalized_variant.rs:443-447)
__global__ __launch_bounds__(256) void k6_cells_kernel(int n_haps, int n_obs, unsigned long long n_comb, int n_sides,
                                                       const signed char* __restrict__ match_slot, const unsigned long long* __restrict__ slot_need,
                                                       const unsigned long long* __restrict__ slot_core, const unsigned char* __restrict__ hap_skip,
                                                       const unsigned char* __restrict__ obs_core, const unsigned char* __restrict__ obs_het,
                                                       const unsigned char* __restrict__ obs_group, const unsigned char* __restrict__ obs_orient01,
                                                       unsigned long long* __restrict__ keys /* [comb][side][hap] */) {
    const unsigned long long cell = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long total = n_comb * (unsigned long long)n_sides * (unsigned long long)n_haps;
    if (cell >= total) return;
    const int h = (int)(cell % (unsigned long long)n_haps);
    const unsigned long long cs = cell / (unsigned long long)n_haps;
    const int side = (int)(cs % (unsigned long long)n_sides);
    const unsigned long long comb = cs / (unsigned long long)n_sides;
    if (hap_skip[h]) { keys[cell] = K6_NOKEY; return; }                       // SV haplotypes are not quantified (:1443-1446)
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
    keys[cell] = ((unsigned long long)mv_core << 48) | ((unsigned long long)ev_core << 32) | ((unsigned long long)mv_sub << 16) | (unsigned long long)ev_sub;
}

extern "C" int32_t sp_variant_solve(sp_ctx* ctx, const sp_variant_problem* p, sp_variant_result* res) {
    if (!ctx || !p || !res) return SP_ERR_INVALID_ARG;
    std::memset(res, 0, sizeof(*res));
    (void)hipSetDevice(ctx->device);
    const int H = p->n_haps, NO = p->n_obs;
    if (NO > K6_MAX_OBS) return sp_fail(ctx, SP_ERR_TOO_LONG, "variant solve: more than 64 observed variants");
    // host prep: which observed variants are hom / het, het groups in first-seen order (diplotyper.rs:1216-1240,1270-1300)
    std::vector<uint8_t> obs_het(std::max(1, NO), 0), obs_group(std::max(1, NO), 0), obs_orient(std::max(1, NO), 1), obs_core(std::max(1, NO), 1);
    std::vector<std::pair<int64_t, int>> ps_bit;             // phase set -> bit of the het assignment, in first-seen order
    int groups = 0, n_het = 0;
    for (int o = 0; o < NO; ++o) {
        if (p->obs_var[o] < 0 || p->obs_var[o] >= p->n_vars) return sp_fail(ctx, SP_ERR_INVALID_ARG, "variant solve: variant id out of range");
        obs_core[o] = p->var_is_core[p->obs_var[o]];
        const int gt = p->obs_gt[o];
        if (gt == SP_GT_HOM_REF) return sp_fail(ctx, SP_ERR_INVALID_ARG, "variant solve: homozygous reference calls must not be passed");
        if (gt == SP_GT_HOM_ALT) continue;
        obs_het[o] = 1; ++n_het; obs_orient[o] = gt != SP_GT_HET_FLIP;
        int bit = -1;
        if (p->obs_ps[o] >= 0) {
            for (auto& kv : ps_bit) if (kv.first == p->obs_ps[o]) bit = kv.second;
            if (bit < 0) { bit = groups++; ps_bit.emplace_back(p->obs_ps[o], bit); }
        } else bit = groups++;                                  // an unphased het is its own group
        obs_group[o] = (uint8_t)bit;
    }
    if (groups > 24) return sp_fail(ctx, SP_ERR_TOO_LONG, "variant solve: more than 24 independent heterozygous groups");
    const unsigned long long n_comb = n_het ? (1ull << (groups - 1)) : 1ull;
    const int n_sides = n_het ? 2 : 1;

    // per haplotype tables
    std::vector<int8_t> match_slot((size_t)std::max(1, H) * std::max(1, NO), -1);
    std::vector<unsigned long long> slot_need(std::max(1, H), 0), slot_core(std::max(1, H), 0);
    std::vector<uint8_t> hap_skip(std::max(1, H), 0);
    for (int h = 0; h < H; ++h) {
        hap_skip[h] = p->hap_is_sv[h] ? 1 : 0;
        const int s0 = p->slot_off[h], s1 = p->slot_off[h + 1];
        if (s1 - s0 > K6_MAX_SLOTS) return sp_fail(ctx, SP_ERR_TOO_LONG, "variant solve: haplotype with more than 64 variants");
        for (int s = s0; s < s1; ++s) {
            bool has_none = false; int first_some = -1;
            for (int x = p->alt_off[s]; x < p->alt_off[s + 1]; ++x) { if (p->alt_var[x] < 0) has_none = true; else if (first_some < 0) first_some = p->alt_var[x]; }
            if (!has_none) slot_need[h] |= 1ull << (s - s0);
            if (first_some >= 0 && p->var_is_core[first_some]) slot_core[h] |= 1ull << (s - s0);
        }
        for (int o = 0; o < NO; ++o)
            for (int s = s0; s < s1 && match_slot[(size_t)h * NO + o] < 0; ++s)
                for (int x = p->alt_off[s]; x < p->alt_off[s + 1]; ++x) if (p->alt_var[x] == p->obs_var[o]) { match_slot[(size_t)h * NO + o] = (int8_t)(s - s0); break; }
    }
    // SV short-circuit is decided per side on the host (diplotyper.rs:1414-1431): collect labels of a side
    auto side_members = [&](unsigned long long comb, int side, std::vector<int>& out) {
        out.clear();
        for (int pass = 0; pass < 2; ++pass)                  // the homozygous variants first, then the side's heterozygous ones (diplotyper.rs:1269-1317)
            for (int o = 0; o < NO; ++o) {
                if ((obs_het[o] != 0) != (pass == 1)) continue;
                bool in_set = true;
                if (obs_het[o]) { const bool is_h1 = ((comb >> obs_group[o]) & 1ull) != 0; const bool to_h1 = is_h1 == (obs_orient[o] != 0); in_set = side == 0 ? to_h1 : !to_h1; }
                if (in_set) out.push_back(o);
            }
    };

    // device scoring
    const unsigned long long n_cells = n_comb * (unsigned long long)n_sides * (unsigned long long)std::max(1, H);
    std::vector<unsigned long long> keys(n_cells, K6_NOKEY);
    if (H > 0) {
        auto up = [&](const char* name, const void* src, size_t bytes) -> void* {
            void* d = sp_pool(ctx, name, std::max<size_t>(1, bytes));
            if (d && bytes) (void)hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream);
            return d;
        };
        signed char* d_ms = (signed char*)up("k6_ms", match_slot.data(), match_slot.size());
        unsigned long long* d_need = (unsigned long long*)up("k6_need", slot_need.data(), slot_need.size() * 8);
        unsigned long long* d_core = (unsigned long long*)up("k6_core", slot_core.data(), slot_core.size() * 8);
        unsigned char* d_skip = (unsigned char*)up("k6_skip", hap_skip.data(), hap_skip.size());
        unsigned char* d_oc = (unsigned char*)up("k6_oc", obs_core.data(), obs_core.size());
        unsigned char* d_oh = (unsigned char*)up("k6_oh", obs_het.data(), obs_het.size());
        unsigned char* d_og = (unsigned char*)up("k6_og", obs_group.data(), obs_group.size());
        unsigned char* d_oo = (unsigned char*)up("k6_oo", obs_orient.data(), obs_orient.size());
        unsigned long long* d_keys = (unsigned long long*)sp_pool(ctx, "k6_keys", n_cells * 8);
        if (!d_ms || !d_need || !d_core || !d_skip || !d_oc || !d_oh || !d_og || !d_oo || !d_keys) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "variant solve buffers");
        {
            ProfScope ps(ctx, "k6_cells", n_cells);
            hipLaunchKernelGGL(k6_cells_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, ctx->stream, H, NO, n_comb, n_sides,
                               d_ms, d_need, d_core, d_skip, d_oc, d_oh, d_og, d_oo, d_keys);
        }
        (void)hipMemcpyAsync(keys.data(), d_keys, n_cells * 8, hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, std::string("variant solve: ") + hipGetErrorString(e));
    }

    // per side: best tuple under the (1, MAX, MAX, MAX) bound, ties, sub-allele shadowing (diplotyper.rs:1433-1509)
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
        const unsigned long long* k = keys.data() + (comb * (unsigned long long)n_sides + (unsigned long long)side) * (unsigned long long)std::max(1, H);
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
    if (!n_het) {
        Side s; solve_side(0, 0, s);
        for (int k = 0; k < 4; ++k) res->score[k] = s.score[k];
        push_pairs(s, s, 0);
        // a homozygous call pairs every haplotype with itself only (diplotyper.rs:1246-1256)
        int w = 0; for (int i = 0; i < res->n_dip; ++i) if (res->dip[i][0] == res->dip[i][1]) { res->dip[w][0] = res->dip[i][0]; res->dip[w][1] = res->dip[i][1]; res->dip_comb[w] = 0; ++w; }
        res->n_dip = w;
        return SP_OK;
    }
    int64_t best[4] = {INT64_MAX, INT64_MAX, INT64_MAX, INT64_MAX};
    for (unsigned long long comb = 0; comb < n_comb; ++comb) {
        Side a, b; solve_side(comb, 0, a); solve_side(comb, 1, b);
        int64_t tot[4]; int cmp = 0;
        for (int k = 0; k < 4; ++k) tot[k] = (a.score[k] == INT64_MAX || b.score[k] == INT64_MAX) ? INT64_MAX : a.score[k] + b.score[k];
        for (int k = 0; k < 4 && !cmp; ++k) cmp = tot[k] < best[k] ? -1 : (tot[k] > best[k]);
        if (cmp < 0) { std::memcpy(best, tot, sizeof(best)); res->n_dip = 0; }
        if (cmp <= 0) push_pairs(a, b, (int)comb);
    }
    for (int k = 0; k < 4; ++k) res->score[k] = best[k];
    return SP_OK;
}

// The solves of a panel or of one GPU's share of a cohort (23 genes x 32 samples): they are independent and each is a handful of small
// copies, one launch and a wait, so they are handed out to the context's streams (sp_ctx_set_option), one host thread per stream for the
// length of the call.  results[i] / problem_rc[i] are what sp_variant_solve gives for problems[i].
extern "C" int32_t sp_variant_solve_batch(sp_ctx* ctx, uint32_t n, const sp_variant_problem* const* problems, sp_variant_result* results, int32_t* problem_rc) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (n && (!problems || !results)) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_variant_solve_batch: null argument");
    for (uint32_t i = 0; i < n; ++i) if (!problems[i]) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_variant_solve_batch: null problem");
    if (n == 0) return SP_OK;
    int n_parts = ctx->split_genes ? (int)std::min<uint32_t>(n, (uint32_t)ctx->split_streams) : 1;
    sp_ctx* on[4] = { ctx, nullptr, nullptr, nullptr };
    for (int x = 1; x < n_parts; ++x) { on[x] = sp_ctx_helper(ctx, x - 1); if (!on[x]) { n_parts = x; break; } }
    std::vector<int32_t> rcs(n, SP_OK); std::vector<int> where(n, 0);
    std::atomic<uint32_t> next(0);
    auto work = [&](int x) {
        for (;;) {
            const uint32_t i = next.fetch_add(1);
            if (i >= n) break;
            where[i] = x;
            rcs[i] = sp_variant_solve(on[x], problems[i], &results[i]);
        }
    };
    std::thread beside[4]; bool started[4] = { false, false, false, false };
    for (int x = 1; x < n_parts; ++x) {
        try { beside[x] = std::thread(work, x); started[x] = true; }
        catch (const std::system_error&) { }
    }
    work(0);
    for (int x = 1; x < n_parts; ++x) if (started[x]) beside[x].join();
    int32_t rc = SP_OK;
    for (uint32_t i = 0; i < n; ++i) {
        if (problem_rc) problem_rc[i] = rcs[i];
        if (rcs[i] != SP_OK && rc == SP_OK) { rc = rcs[i]; if (where[i] > 0) ctx->err = on[where[i]]->err; }
    }
    return rc;
}
