// sp_cyp.hip -- K5: CYP2D6 chain-pair likelihood search on gfx950.
// Replaces find_best_chain_pair and its helpers (src/cyp2d6/chaining.rs:223-903).  Host: label grammar, edge tables,
// LIFO chain enumeration (sequential by nature, tiny).  Device: one thread per unordered chain pair; the per-read
// containment scores, the haplotype weights and the multinomial ln-likelihood are accumulated in f64 in exactly the
// reference's operation order (no FMA contraction, table look-ups for ln n! and ln p), so the scores are bit-identical
// to the CPU path and the (score, i, j) winner is deterministic.
#include "sp_internal.h"
#include <system_error>
#include <thread>
#include <atomic>
#include <memory>
#include <map>
#include "sp_json.h"
#include "sp_wfa.hip.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace {

// ---- region label helpers (src/cyp2d6/region_label.rs) ----
inline bool t_is_cyp2d(int t) { return t == SP_CYP_CYP2D6 || t == SP_CYP_CYP2D7 || t == SP_CYP_DELETION || t == SP_CYP_HYBRID; }
inline bool t_is_rep(int t) { return t == SP_CYP_REP6 || t == SP_CYP_REP7; }
inline bool t_is_reported(int t) { return t == SP_CYP_CYP2D6 || t == SP_CYP_DELETION || t == SP_CYP_HYBRID; }
inline bool t_allowed(int t) { return t != SP_CYP_UNKNOWN && t != SP_CYP_FALSE_ALLELE; }
inline bool t_normalizing(int t, bool all) { return all ? t_is_cyp2d(t) : t == SP_CYP_CYP2D6; }
inline bool t_head(int t, bool all) {
    if (t == SP_CYP_REP6 || t == SP_CYP_DELETION) return true;
    if (t == SP_CYP_CYP2D6 || t == SP_CYP_HYBRID) return t_normalizing(t, all);
    return false;
}
// is_allowed_label_pair (region_label.rs:178-222)
bool label_pair_allowed(int a, int b) {
    const bool double_star5 = a == SP_CYP_DELETION && b == SP_CYP_DELETION;
    const bool bad_order = b == SP_CYP_REP6 || (t_is_cyp2d(a) && a != SP_CYP_DELETION && b != SP_CYP_LINK_REGION) ||
        (b == SP_CYP_LINK_REGION && !t_is_cyp2d(a)) || (a == SP_CYP_LINK_REGION && !t_is_rep(b)) || (t_is_rep(b) && a != SP_CYP_LINK_REGION) ||
        (t_is_rep(a) && !(b == SP_CYP_SPACER || t_is_cyp2d(b))) || (b == SP_CYP_SPACER && !(t_is_rep(a) || a == SP_CYP_DELETION)) ||
        (a == SP_CYP_SPACER && !t_is_cyp2d(b)) || (b == SP_CYP_CYP2D7 && a != SP_CYP_SPACER) || a == SP_CYP_CYP2D7;
    return !double_star5 && !bad_order;
}
const char* type_str(int t) {
    static const char* names[] = {"UNKNOWN", "REP6", "CYP2D6", "link_region", "REP7", "spacer", "CYP2D7", "CYP2D6*5", "Hybrid", "FalseAllele"};
    return (t >= 0 && t <= 9) ? names[t] : "UNKNOWN";
}
std::string full_allele(int t, const char* sub) {                               // region_label.rs:131-170
    if (t == SP_CYP_CYP2D6) return sub ? std::string("CYP2D6*") + sub : std::string("CYP2D6");
    if (t == SP_CYP_HYBRID) return sub ? std::string(sub) : std::string("Hybrid");
    if (t == SP_CYP_FALSE_ALLELE) return sub ? std::string("FalseAllele_") + sub : std::string("FalseAllele");
    return type_str(t);
}
// simplify_allele(detailed = false) (region_label.rs:77-128)
std::string simple_allele(const sp_chain_problem* p, uint32_t h) {
    const int t = p->hap_type[h]; const char* sub = p->hap_subtype ? p->hap_subtype[h] : nullptr;
    if (t == SP_CYP_CYP2D6 || t == SP_CYP_HYBRID) {
        if (!sub) return full_allele(t, sub);
        for (uint32_t i = 0; i < p->n_translate; ++i) if (std::strcmp(p->translate_key[i], sub) == 0) return std::string("*") + p->translate_val[i];
        char* endp = nullptr;
        const double v = std::strtod(sub, &endp);
        const bool parsed = sub[0] != '\0' && sub[0] != ' ' && endp && *endp == '\0' && !(sub[0] == '0' && (sub[1] == 'x' || sub[1] == 'X'));
        if (parsed) return std::string("*") + std::to_string((long long)std::floor(v));
        return std::string("*") + sub;
    }
    if (t == SP_CYP_DELETION) return "*5";
    return full_allele(t, sub);
}

struct PairConsts {
    int H, P, maxlen, R;
    int ignore_limits, normalize_all, infer;
    double lasso, ln_ed, unexpected, inferred;
    uint64_t n_pairs;
};

} // namespace

// =============================================================================================
// device
// =============================================================================================
#define K5_MAXH 64

__device__ __forceinline__ void k5_pair_from_id(uint64_t pid, int P, int& i, int& j) {
    // rows i = 0..P-1 hold P - i pairs (j = i..P-1); offset(i) = i*P - i*(i-1)/2
    double Pd = (double)P;
    double disc = (2.0 * Pd + 1.0) * (2.0 * Pd + 1.0) - 8.0 * (double)pid;
    long long ii = (long long)(((2.0 * Pd + 1.0) - sqrt(disc)) * 0.5);
    if (ii < 0) ii = 0;
    if (ii > P - 1) ii = P - 1;
    auto off = [&](long long x) -> uint64_t { return (uint64_t)x * (uint64_t)P - (uint64_t)(x * (x - 1) / 2); };
    while (ii > 0 && off(ii) > pid) --ii;
    while (ii + 1 < P && off(ii + 1) <= pid) ++ii;
    i = (int)ii; j = (int)(pid - off(ii)) + i;
}

// per (chain, read): the best window total of the read's weight rows along the chain and which starts reach it (bit s of the mask).
// These do not depend on the partner chain, so the pair kernel below looks them up instead of sliding every read along both chains
// of every pair (containment_score, chaining.rs:683-731).  Tables are read-major ([read][chain]): a wave of consecutive pairs reads them
// coalesced.
__global__ __launch_bounds__(256) void k5_chain_read_kernel(int P, int R, int H, int maxlen, const uint8_t* __restrict__ chains, const int32_t* __restrict__ chain_len,
                                                            const int32_t* __restrict__ read_w_off, const uint32_t* __restrict__ w_ed,
                                                            unsigned long long* __restrict__ tab_best, unsigned long long* __restrict__ tab_mask) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)P * R) return;
    const int c = (int)(id % P), r = (int)(id / P);
    const uint8_t* o = chains + (size_t)c * maxlen; const int on = chain_len[c];
    const int row0 = read_w_off[r], wl = read_w_off[r + 1] - row0;
    const uint32_t* ed = w_ed + (size_t)row0 * H;
    unsigned long long best = ~0ull, mask = 0;
    for (int s = 0; s + wl <= on; ++s) {
        unsigned long long tot = 0;
        for (int x = 0; x < wl; ++x) tot += ed[(size_t)x * H + o[s + x]];
        if (tot < best) { best = tot; mask = 1ull << s; } else if (tot == best) mask |= 1ull << s;
    }
    tab_best[id] = best; tab_mask[id] = mask;
}

// get_multinomial_score (chaining.rs:854-903) + multinomial_ln_pmf (util/stats.rs:11-37) of a pair: cnt(h) = copies of region h in the two chains, hw(h) = the
// coverage the reads gave it
template <typename Cnt, typename Hw>
__device__ __forceinline__ double k5_multinomial(const PairConsts& c, Cnt cnt, Hw hw, const uint8_t* __restrict__ hap_norm, const double* __restrict__ ln_fact, int ln_fact_n,
                                                 const double* __restrict__ ln_p, int ln_p_stride, bool both_have_deletion, bool& valid) {
    const int H = c.H;
    int nr = 0, total = 0; unsigned long long cov_sum = 0;
    for (int h = 0; h < H; ++h) if (cnt(h) > 0 && hap_norm[h]) { ++nr; total += cnt(h); cov_sum += (unsigned long long)round(hw(h)); }
    valid = true;
    if (nr == 0 || cov_sum == 0) { valid = !c.normalize_all && both_have_deletion; return 0.0; }
    double coeff = cov_sum < (unsigned long long)ln_fact_n ? ln_fact[cov_sum] : ln_fact[ln_fact_n - 1];
    for (int h = 0; h < H; ++h) if (cnt(h) > 0 && hap_norm[h]) coeff -= ln_fact[(unsigned long long)round(hw(h))];
    double acc = 0.0;
    for (int h = 0; h < H; ++h) if (cnt(h) > 0 && hap_norm[h]) acc = acc + (double)((unsigned long long)round(hw(h))) * ln_p[(size_t)cnt(h) * ln_p_stride + total];
    return fabs(coeff + acc);
}

// one thread = one unordered pair (i <= j)
__global__ __launch_bounds__(256) void k5_pair_kernel(PairConsts c,
                                                      const uint8_t* __restrict__ chains, const int32_t* __restrict__ chain_len,   // [P][maxlen]
                                                      const uint32_t* __restrict__ chain_unexp, const uint32_t* __restrict__ chain_inf,
                                                      const uint8_t* __restrict__ chain_has_del,
                                                      const uint8_t* __restrict__ hap_lasso, const uint8_t* __restrict__ hap_norm,
                                                      const int32_t* __restrict__ read_w_off, const uint32_t* __restrict__ w_ed, const double* __restrict__ w_ov,
                                                      const unsigned long long* __restrict__ tab_best, const unsigned long long* __restrict__ tab_mask,
                                                      const uint64_t* __restrict__ read_optimum, const uint64_t* __restrict__ read_worst,
                                                      const double* __restrict__ ln_fact, int ln_fact_n,
                                                      const double* __restrict__ ln_p, int ln_p_stride,      // ln_p[c * stride + total]
                                                      unsigned long long* __restrict__ global_best,
                                                      double* __restrict__ blk_score, unsigned long long* __restrict__ blk_pid,
                                                      double* __restrict__ blk_comp, unsigned long long* __restrict__ blk_ed,
                                                      unsigned long long* __restrict__ n_scored) {
    const uint64_t pid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int H = c.H;
    bool have = false;
    int scored = 0;
    double primary = 0.0, comp_lned = 0.0, comp_mn = 0.0, comp_exp = 0.0, comp_unexp = 0.0, comp_inf = 0.0;
    unsigned long long ed_out = 0;
    if (pid < c.n_pairs) {
        int i, j; k5_pair_from_id(pid, c.P, i, j);
        const uint8_t* ci = chains + (size_t)i * c.maxlen; const uint8_t* cj = chains + (size_t)j * c.maxlen;
        const int ni = chain_len[i], nj = chain_len[j];
        unsigned char cnt[K5_MAXH];
        for (int h = 0; h < H; ++h) cnt[h] = 0;
        for (int x = 0; x < ni; ++x) cnt[ci[x]]++;
        for (int x = 0; x < nj; ++x) cnt[cj[x]]++;
        int unexpected_alleles = 0;                                           // count_unexpected_alleles (chaining.rs:794-819)
        for (int h = 0; h < H; ++h) if (hap_lasso[h] && cnt[h] > 0) unexpected_alleles += cnt[h] - 1;
        const double allele_expected_penalty = c.lasso * (double)unexpected_alleles;
        const unsigned mismatch = c.ignore_limits ? 0u : chain_unexp[i] + chain_unexp[j];
        const double unexpected_chain_penalty = (double)mismatch * c.unexpected;
        const unsigned n_inf = c.infer ? chain_inf[i] + chain_inf[j] : 0u;
        const double inferred_chain_penalty = (double)n_inf * c.inferred;
        const double partial_cost = allele_expected_penalty + unexpected_chain_penalty + inferred_chain_penalty;
        // exact pruning: a pair whose cheap partial cost already exceeds the best complete score cannot win (chaining.rs:457-464)
        const double gb = __longlong_as_double((long long)__hip_atomic_load(global_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (!(partial_cost > gb)) {
            scored = 1;
            // per-haplotype coverage of this thread's pair, in LDS ([h][thread]: conflict-free): indexed by data, the array lived in scratch memory and
            // every add was a dependent round trip there -- 1.6 us per read for the 210 pairs of a sample
            extern __shared__ double k5_hw[];
            double* const hw = k5_hw + threadIdx.x;
#define HW(h) hw[(size_t)(h) * 256]
            for (int h = 0; h < H; ++h) HW(h) = 0.0;
            unsigned long long read_combined_ed = 0;
            // (the reads are taken in order -- the f64 sums below depend on it --, but what a read needs from memory does not depend on the sums: the
            //  next read's seven words are on their way while this one is added up; a read used to cost four dependent round trips, 1.7 us)
            struct ReadWords { int row0, row1; unsigned long long bi, bj, mi, mj, worst, opt; };
            auto fetch = [&](int r) {
                ReadWords w;
                w.row0 = read_w_off[r]; w.row1 = read_w_off[r + 1];
                w.bi = tab_best[(size_t)r * c.P + i]; w.bj = tab_best[(size_t)r * c.P + j];
                w.mi = tab_mask[(size_t)r * c.P + i]; w.mj = tab_mask[(size_t)r * c.P + j];
                w.worst = read_worst[r]; w.opt = read_optimum[r];
                return w;
            };
            ReadWords nxt = c.R > 0 ? fetch(0) : ReadWords();
            for (int r = 0; r < c.R; ++r) {
                const ReadWords cur = nxt;
                if (r + 1 < c.R) nxt = fetch(r + 1);
                const int row0 = cur.row0, wl = cur.row1 - row0;
                const double* ov = w_ov + (size_t)row0 * H;
                // containment_score (chaining.rs:683-731): best window total over both chains, ties kept in visiting order (chain i's windows,
                // then chain j's) -- from the per-chain tables
                unsigned long long best_score = 2ull * cur.worst;
                int n_best = 0;
                const unsigned long long bi = cur.bi, bj = cur.bj;
                const unsigned long long mi = cur.mi, mj = cur.mj;
                if (bi < best_score) { best_score = bi; n_best = __popcll(mi); } else if (bi == best_score) n_best += __popcll(mi);
                if (bj < best_score) { best_score = bj; n_best = __popcll(mj); } else if (bj == best_score) n_best += __popcll(mj);
                const unsigned long long sc = best_score - cur.opt;
                const unsigned long long sum = read_combined_ed + sc;
                read_combined_ed = sum < read_combined_ed ? 0xFFFFFFFFFFFFFFFFull : sum;        // saturating_add
                const double split_frac = 1.0 / (double)n_best;
                for (int which = 0; which < 2; ++which) {
                    const uint8_t* o = which ? cj : ci;
                    unsigned long long m = (which ? bj : bi) == best_score ? (which ? mj : mi) : 0ull;
                    while (m) {
                        const int s = __builtin_ctzll(m); m &= m - 1;
                        for (int x = 0; x < wl; ++x) { const int con = o[s + x]; HW(con) += split_frac * ov[(size_t)x * H + con]; }
                    }
                }
            }
            const double ln_ed_penalty = (double)read_combined_ed * c.ln_ed;
            bool valid = true;
            const double mn = k5_multinomial(c, [&](int h) { return (int)cnt[h]; }, [&](int h) { return HW(h); }, hap_norm, ln_fact, ln_fact_n, ln_p, ln_p_stride,
                                             chain_has_del[i] && chain_has_del[j], valid);
            if (valid) {
                primary = ln_ed_penalty + mn + allele_expected_penalty + unexpected_chain_penalty + inferred_chain_penalty;    // chaining.rs:172-174
                have = true; comp_lned = ln_ed_penalty; comp_mn = mn; comp_exp = allele_expected_penalty; comp_unexp = unexpected_chain_penalty;
                comp_inf = inferred_chain_penalty; ed_out = read_combined_ed;
                atomicMin(global_best, (unsigned long long)__double_as_longlong(primary));   // scores are >= 0: bit order = numeric order
            }
        }
    }
    {   // pairs whose read-level terms were evaluated: one atomic per wavefront, not per pair
        const unsigned long long m = __ballot(scored != 0);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_scored, (unsigned long long)__popcll(m));
    }
    // block-level lexicographic min (score, pid)
    __shared__ double s_score[256]; __shared__ unsigned long long s_pid[256];
    s_score[threadIdx.x] = have ? primary : 1.0e308 * 10.0; s_pid[threadIdx.x] = have ? pid : 0xFFFFFFFFFFFFFFFFull;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double so = s_score[threadIdx.x + o]; const unsigned long long po = s_pid[threadIdx.x + o];
            if (po != 0xFFFFFFFFFFFFFFFFull && (s_pid[threadIdx.x] == 0xFFFFFFFFFFFFFFFFull || so < s_score[threadIdx.x] ||
                                                (so == s_score[threadIdx.x] && po < s_pid[threadIdx.x]))) { s_score[threadIdx.x] = so; s_pid[threadIdx.x] = po; }
        }
        __syncthreads();
    }
    if (have && s_pid[0] == pid) {
        blk_score[blockIdx.x] = primary; blk_pid[blockIdx.x] = pid; blk_ed[blockIdx.x] = ed_out;
        double* cp = blk_comp + (size_t)blockIdx.x * 5; cp[0] = comp_lned; cp[1] = comp_mn; cp[2] = comp_exp; cp[3] = comp_unexp; cp[4] = comp_inf;
    } else if (threadIdx.x == 0 && s_pid[0] == 0xFFFFFFFFFFFFFFFFull) {
        blk_pid[blockIdx.x] = 0xFFFFFFFFFFFFFFFFull;
    }
}

// The same score with one WORKGROUP per pair, for the few pairs of an ordinary sample (tens to hundreds: a thread per pair walks the reads one after another, 1.1 us
// each, 2.3 ms for 2,000 reads whatever the number of pairs).  What a read adds to the coverage of the regions -- (region, value) addends, in the order the walk
// would add them -- is worked out for 256 reads at a time, a read per thread; then thread h adds up region h's addends read by read: the f64 sums see the same
// operands in the same order.  A read with more than K5_ADDS addends is added up from memory at its turn.
#define K5_TILE 256
#define K5_ADDS 8
__global__ __launch_bounds__(K5_TILE) void k5_pair_block_kernel(PairConsts c,
                                                      const uint8_t* __restrict__ chains, const int32_t* __restrict__ chain_len,
                                                      const uint32_t* __restrict__ chain_unexp, const uint32_t* __restrict__ chain_inf,
                                                      const uint8_t* __restrict__ chain_has_del,
                                                      const uint8_t* __restrict__ hap_lasso, const uint8_t* __restrict__ hap_norm,
                                                      const int32_t* __restrict__ read_w_off, const double* __restrict__ w_ov,
                                                      const unsigned long long* __restrict__ tab_best, const unsigned long long* __restrict__ tab_mask,
                                                      const uint64_t* __restrict__ read_optimum, const uint64_t* __restrict__ read_worst,
                                                      const double* __restrict__ ln_fact, int ln_fact_n,
                                                      const double* __restrict__ ln_p, int ln_p_stride,
                                                      unsigned long long* __restrict__ global_best,
                                                      double* __restrict__ blk_score, unsigned long long* __restrict__ blk_pid,
                                                      double* __restrict__ blk_comp, unsigned long long* __restrict__ blk_ed,
                                                      unsigned long long* __restrict__ n_scored) {
    __shared__ double s_val[K5_TILE * K5_ADDS];
    __shared__ unsigned long long s_con[K5_TILE];                             // the regions of a read's addends, a byte each (255: none)
    __shared__ int s_nadd[K5_TILE];
    __shared__ unsigned long long s_ed[K5_TILE];
    __shared__ double s_hw[K5_MAXH];
    __shared__ unsigned char s_cnt[K5_MAXH];
    __shared__ double s_cost[3];
    __shared__ int s_go;
    const uint64_t pid = blockIdx.x;
    const int H = c.H, tid = (int)threadIdx.x;
    int i, j; k5_pair_from_id(pid, c.P, i, j);
    const uint8_t* ci = chains + (size_t)i * c.maxlen; const uint8_t* cj = chains + (size_t)j * c.maxlen;
    if (tid == 0) {
        const int ni = chain_len[i], nj = chain_len[j];
        for (int h = 0; h < H; ++h) s_cnt[h] = 0;
        for (int x = 0; x < ni; ++x) s_cnt[ci[x]]++;
        for (int x = 0; x < nj; ++x) s_cnt[cj[x]]++;
        int unexpected_alleles = 0;                                           // count_unexpected_alleles (chaining.rs:794-819)
        for (int h = 0; h < H; ++h) if (hap_lasso[h] && s_cnt[h] > 0) unexpected_alleles += s_cnt[h] - 1;
        s_cost[0] = c.lasso * (double)unexpected_alleles;
        s_cost[1] = (double)(c.ignore_limits ? 0u : chain_unexp[i] + chain_unexp[j]) * c.unexpected;
        s_cost[2] = (double)(c.infer ? chain_inf[i] + chain_inf[j] : 0u) * c.inferred;
        const double partial_cost = s_cost[0] + s_cost[1] + s_cost[2];
        const double gb = __longlong_as_double((long long)__hip_atomic_load(global_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        s_go = !(partial_cost > gb);                                          // exact pruning (chaining.rs:457-464)
        if (!s_go) blk_pid[pid] = 0xFFFFFFFFFFFFFFFFull;
    }
    __syncthreads();
    if (!s_go) return;
    struct ReadTerms { int row0, wl; unsigned long long mi, mj, sc; double split_frac; };
    auto terms = [&](int r) {
        ReadTerms t;
        t.row0 = read_w_off[r]; t.wl = read_w_off[r + 1] - t.row0;
        const unsigned long long bi = tab_best[(size_t)r * c.P + i], bj = tab_best[(size_t)r * c.P + j];
        const unsigned long long mi = tab_mask[(size_t)r * c.P + i], mj = tab_mask[(size_t)r * c.P + j];
        unsigned long long best_score = 2ull * read_worst[r];                 // containment_score (chaining.rs:683-731), as in k5_pair_kernel
        int n_best = 0;
        if (bi < best_score) { best_score = bi; n_best = __popcll(mi); } else if (bi == best_score) n_best += __popcll(mi);
        if (bj < best_score) { best_score = bj; n_best = __popcll(mj); } else if (bj == best_score) n_best += __popcll(mj);
        t.sc = best_score - read_optimum[r];
        t.split_frac = 1.0 / (double)n_best;
        t.mi = bi == best_score ? mi : 0ull; t.mj = bj == best_score ? mj : 0ull;
        return t;
    };
    double acc = 0.0;                                                         // thread h: the coverage of region h
    unsigned long long ed_part = 0;
    for (int base = 0; base < c.R; base += K5_TILE) {
        const int r = base + tid;
        int nadd = 0;
        if (r < c.R) {
            const ReadTerms t = terms(r);
            const unsigned long long sum = ed_part + t.sc;
            ed_part = sum < ed_part ? 0xFFFFFFFFFFFFFFFFull : sum;            // saturating_add (the order does not matter: every term is >= 0)
            const int total = (__popcll(t.mi) + __popcll(t.mj)) * t.wl;
            unsigned long long cons = ~0ull;
            if (total > K5_ADDS) nadd = -1;
            else {
                const double* ov = w_ov + (size_t)t.row0 * H;
                for (int which = 0; which < 2; ++which) {
                    const uint8_t* o = which ? cj : ci;
                    unsigned long long m = which ? t.mj : t.mi;
                    while (m) {
                        const int s = __builtin_ctzll(m); m &= m - 1;
                        for (int x = 0; x < t.wl; ++x) {
                            const int con = o[s + x];
                            cons = (cons & ~(0xFFull << (8 * nadd))) | ((unsigned long long)con << (8 * nadd));
                            s_val[tid * K5_ADDS + nadd] = t.split_frac * ov[(size_t)x * H + con]; ++nadd;
                        }
                    }
                }
            }
            s_con[tid] = cons;
        }
        s_nadd[tid] = nadd;
        __syncthreads();
        if (tid < H) {
            const int lim = c.R - base < K5_TILE ? c.R - base : K5_TILE;
#pragma unroll 2
            for (int q = 0; q < lim; ++q) {
                if (s_nadd[q] >= 0) {                                          // (the eight slots are read together; an empty slot names region 255)
                    const unsigned long long cons = s_con[q];
                    double v[K5_ADDS];
#pragma unroll
                    for (int a = 0; a < K5_ADDS; ++a) v[a] = s_val[q * K5_ADDS + a];
#pragma unroll
                    for (int a = 0; a < K5_ADDS; ++a) if ((int)((cons >> (8 * a)) & 0xFF) == tid) acc += v[a];
                } else {
                    const ReadTerms t = terms(base + q);
                    const double* ov = w_ov + (size_t)t.row0 * H;
                    for (int which = 0; which < 2; ++which) {
                        const uint8_t* o = which ? cj : ci;
                        unsigned long long m = which ? t.mj : t.mi;
                        while (m) {
                            const int s = __builtin_ctzll(m); m &= m - 1;
                            for (int x = 0; x < t.wl; ++x) { const int con = o[s + x]; if (con == tid) acc += t.split_frac * ov[(size_t)x * H + con]; }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid < H) s_hw[tid] = acc;
    s_ed[tid] = ed_part;
    __syncthreads();
    if (tid != 0) return;
    unsigned long long read_combined_ed = 0;
    for (int q = 0; q < K5_TILE; ++q) { const unsigned long long sum = read_combined_ed + s_ed[q]; read_combined_ed = sum < read_combined_ed ? 0xFFFFFFFFFFFFFFFFull : sum; }
    const double ln_ed_penalty = (double)read_combined_ed * c.ln_ed;
    bool valid = true;
    const double mn = k5_multinomial(c, [&](int h) { return (int)s_cnt[h]; }, [&](int h) { return s_hw[h]; }, hap_norm, ln_fact, ln_fact_n, ln_p, ln_p_stride,
                                     chain_has_del[i] && chain_has_del[j], valid);
    atomicAdd(n_scored, 1ull);
    if (!valid) { blk_pid[pid] = 0xFFFFFFFFFFFFFFFFull; return; }
    const double primary = ln_ed_penalty + mn + s_cost[0] + s_cost[1] + s_cost[2];                                        // chaining.rs:172-174
    blk_score[pid] = primary; blk_pid[pid] = pid; blk_ed[pid] = read_combined_ed;
    double* cp = blk_comp + (size_t)pid * 5; cp[0] = ln_ed_penalty; cp[1] = mn; cp[2] = s_cost[0]; cp[3] = s_cost[1]; cp[4] = s_cost[2];
    atomicMin(global_best, (unsigned long long)__double_as_longlong(primary));
}

// =============================================================================================
// host
// =============================================================================================
template <typename T> static T* k5_upload(sp_ctx* ctx, const char* name, const std::vector<T>& v) {
    T* d = (T*)sp_pool(ctx, name, std::max<size_t>(1, v.size()) * sizeof(T));
    if (d && !v.empty()) (void)hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
    return d;
}

extern "C" int32_t sp_cyp_best_chain_pair(sp_ctx* ctx, const sp_chain_problem* p, sp_chain_result* res) {
    if (!ctx || !p || !res || (p->n_haps && !p->hap_type)) return SP_ERR_INVALID_ARG;
    std::memset(res, 0, sizeof(*res));
    (void)hipSetDevice(ctx->device);
    const int H = (int)p->n_haps;
    if (H > K5_MAXH) return sp_fail(ctx, SP_ERR_INVALID_ARG, "chain pair: more than 64 consensus regions");
    if (p->lasso_penalty < 0.0) return sp_fail(ctx, SP_ERR_INVALID_ARG, "Lasso penalty must be >= 0.0");      // chaining.rs:233-235
    const bool ignore = p->ignore_chain_label_limits != 0, norm_all = p->normalize_all_alleles != 0, infer = p->infer_connections != 0;
    std::vector<int> type(p->hap_type, p->hap_type + H);
    std::vector<std::string> simple(H);
    for (int h = 0; h < H; ++h) simple[h] = simple_allele(p, (uint32_t)h);
    auto connected = [&](const std::string& a, const std::string& b) {
        for (uint32_t i = 0; i < p->n_connections; ++i) if (a == p->connection_a[i] && b == p->connection_b[i]) return true;
        return false;
    };
    // observed edges (chaining.rs:245-264)
    std::vector<uint8_t> down((size_t)H * H, 0), inferred((size_t)H * H, 0);
    for (uint32_t r = 0; r < p->n_reads; ++r)
        for (uint32_t c = p->read_chain_off[r]; c < p->read_chain_off[r + 1]; ++c)
            for (uint32_t x = p->chain_off[c] + 1; x < p->chain_off[c + 1]; ++x) {
                const uint32_t up = p->chain_items[x - 1], dn = p->chain_items[x];
                if (up >= (uint32_t)H || dn >= (uint32_t)H) return sp_fail(ctx, SP_ERR_INVALID_ARG, "chain pair: consensus index out of range");
                if (t_allowed(type[up]) && t_allowed(type[dn]) && (ignore || label_pair_allowed(type[up], type[dn]))) down[up * H + dn] = 1;
            }
    // inferred edges (chaining.rs:267-305)
    if (infer) for (int i = 0; i < H; ++i) {
        bool out_none = true; for (int j = 0; j < H; ++j) out_none &= !down[i * H + j];
        for (int j = 0; j < H; ++j) {
            bool in_none = true; for (int v = 0; v < H; ++v) in_none &= !down[v * H + j];
            if ((out_none || in_none) && !down[i * H + j] && t_allowed(type[i]) && t_allowed(type[j]) && label_pair_allowed(type[i], type[j])) inferred[i * H + j] = 1;
        }
    }
    // chain heads (chaining.rs:308-323)
    std::vector<std::vector<int>> stack;
    for (int i = 0; i < H; ++i) if (ignore || t_head(type[i], norm_all)) stack.push_back({i});
    if (stack.empty()) return SP_ERR_NO_CHAINING_HEAD;
    // LIFO enumeration, copy number <= 3 (chaining.rs:326-391)
    std::vector<std::vector<int>> possible;
    while (!stack.empty()) {
        std::vector<int> cur = std::move(stack.back()); stack.pop_back();
        // check_chain_inferrences (chaining.rs:603-674)
        bool ok_infer = true, ok_cand = true;
        {
            const int last = cur.back(); const bool last_d = t_is_cyp2d(type[last]);
            int prev_pos = -1;
            for (int x = (int)cur.size() - 2; x >= 0; --x) if (t_is_cyp2d(type[cur[x]])) { prev_pos = x; break; }
            bool seen = false;
            for (size_t w = prev_pos < 0 ? 0 : (size_t)prev_pos; w + 1 < cur.size(); ++w) seen |= inferred[cur[w] * H + cur[w + 1]] != 0;
            if (seen) {
                if (last_d) {
                    if (prev_pos >= 0) {
                        const int pv = cur[prev_pos];
                        const bool conn = pv != last && connected(simple[pv], simple[last]);
                        const bool d7_tail = type[last] == SP_CYP_CYP2D7 && type[pv] != SP_CYP_CYP2D7 && t_is_cyp2d(type[pv]);
                        ok_infer = ok_cand = conn || d7_tail;
                    }
                } else ok_cand = false;
            }
        }
        if (!ok_infer) continue;
        bool reportable = false;                       // convert_chain_to_hap(.., SubAlleles, ..) non-empty (caller.rs:907-957)
        for (int h : cur) reportable |= t_is_cyp2d(type[h]) && type[h] != SP_CYP_CYP2D7;
        if (ignore || (reportable && ok_cand)) possible.push_back(cur);
        const int tail = cur.back();
        for (int rep = 0; rep < (infer ? 2 : 1); ++rep) {
            const std::vector<uint8_t>& tab = rep ? inferred : down;
            for (int e = 0; e < H; ++e) {
                if (!tab[tail * H + e]) continue;
                if (std::count(cur.begin(), cur.end(), e) >= 3) continue;
                if (cur.size() + 1 > SP_MAX_CHAIN) return sp_fail(ctx, SP_ERR_TOO_LONG, "chain pair: chain longer than 64 regions");
                std::vector<int> nxt(cur); nxt.push_back(e); stack.push_back(std::move(nxt));
            }
        }
    }
    if (possible.empty()) return SP_ERR_NO_CHAINS_FOUND;
    const int P = (int)possible.size();
    res->n_possible = P;

    // flatten for the device
    int maxlen = 1; for (auto& c : possible) maxlen = std::max(maxlen, (int)c.size());
    std::vector<uint8_t> chains((size_t)P * maxlen, 0), has_del(P, 0);
    std::vector<int32_t> clen(P);
    std::vector<uint32_t> unexp(P, 0), ninf(P, 0);
    for (int i = 0; i < P; ++i) {
        const auto& c = possible[i];
        clen[i] = (int32_t)c.size();
        for (size_t x = 0; x < c.size(); ++x) { chains[(size_t)i * maxlen + x] = (uint8_t)c[x]; has_del[i] |= type[c[x]] == SP_CYP_DELETION; }
        for (size_t x = 0; x + 1 < c.size(); ++x) ninf[i] += inferred[c[x] * H + c[x + 1]];
        // unexpected_count (chaining.rs:739-775)
        std::vector<const std::string*> red;
        for (int h : c) if (t_is_cyp2d(type[h]) && type[h] != SP_CYP_CYP2D7) red.push_back(&simple[h]);
        uint32_t e = 0;
        if (red.empty() || (*red[0])[0] != '*') e += 1;
        if (red.size() == 1) for (uint32_t s = 0; s < p->n_singletons; ++s) if (*red[0] == p->singletons[s]) { e += 1; break; }
        for (size_t x = 0; x + 1 < red.size(); ++x) if (!connected(*red[x], *red[x + 1])) e += 1;
        unexp[i] = e;
    }
    std::vector<uint8_t> hap_lasso(H), hap_norm(H);
    for (int h = 0; h < H; ++h) {
        hap_lasso[h] = t_allowed(type[h]) && (ignore || t_normalizing(type[h], norm_all) || t_is_reported(type[h]));
        hap_norm[h] = ignore || t_normalizing(type[h], norm_all);
    }
    const int R = (int)p->n_reads;
    const uint32_t n_rows = p->read_w_off ? p->read_w_off[R] : 0;
    std::vector<int32_t> rwo(R + 1, 0);
    for (int r = 0; r <= R; ++r) rwo[r] = p->read_w_off ? (int32_t)p->read_w_off[r] : 0;
    std::vector<uint32_t> ed32((size_t)n_rows * H);
    std::vector<uint64_t> optimum(R, 0), worst(R, 0);
    for (int r = 0; r < R; ++r) for (int row = rwo[r]; row < rwo[r + 1]; ++row) {
        uint64_t mn = UINT64_MAX, mx = 0;
        for (int h = 0; h < H; ++h) {
            const uint64_t v = p->w_ed[(size_t)row * H + h];
            if (v > 0x7FFFFFFFull) return sp_fail(ctx, SP_ERR_INVALID_ARG, "chain pair: edit distance does not fit 31 bits");
            ed32[(size_t)row * H + h] = (uint32_t)v; mn = std::min(mn, v); mx = std::max(mx, v);
        }
        optimum[r] += mn; worst[r] += mx;
    }
    // ln n! (statrs ln_factorial: cached products up to 170!, ln_gamma beyond) and ln(c / total)
    double ov_bound = 0.0;                                   // sum over rows of the largest overlap: upper bound of every rounded weight sum
    for (uint32_t row = 0; row < n_rows; ++row) { double m = 0.0; for (int h = 0; h < H; ++h) m = std::max(m, p->w_ov[(size_t)row * H + h]); ov_bound += m; }
    const int lf_n = (int)std::ceil(ov_bound) + H + 2;
    std::vector<double> ln_fact(lf_n);
    { double f = 1.0; for (int n = 0; n < lf_n; ++n) { if (n >= 2 && n <= 170) f *= (double)n; ln_fact[n] = n <= 170 ? std::log(f) : std::lgamma((double)n + 1.0); } }
    const int max_cnt = 6, max_total = 6 * std::max(1, H);
    std::vector<double> ln_p((size_t)(max_cnt + 1) * (max_total + 1), 0.0);
    for (int cc = 1; cc <= max_cnt; ++cc) for (int t = cc; t <= max_total; ++t) ln_p[(size_t)cc * (max_total + 1) + t] = std::log((double)cc / (double)t);

    const uint64_t n_pairs = (uint64_t)P * ((uint64_t)P + 1) / 2;
    const bool block_per_pair = n_pairs <= (uint64_t)ctx->k5_block_pairs;   // (k5_pair_block_kernel)
    const uint64_t blocks = block_per_pair ? n_pairs : (n_pairs + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return sp_fail(ctx, SP_ERR_TOO_LONG, "chain pair: too many chain pairs for one launch");
    PairConsts pc{H, P, maxlen, R, ignore ? 1 : 0, norm_all ? 1 : 0, infer ? 1 : 0, p->lasso_penalty, p->ln_ed_penalty, p->unexpected_chain_penalty,
                  p->inferred_edge_penalty, n_pairs};
    std::vector<double> ovv(p->w_ov, p->w_ov + (size_t)n_rows * H);
    // the fourteen input arrays go up in ONE copy out of a pinned staging buffer (a copy each was fourteen links in the stream's chain: 0.2 ms of a small sample)
    size_t in_bytes = 0;
    auto place = [&](size_t bytes) { const size_t at = (in_bytes + 15) & ~(size_t)15; in_bytes = at + std::max<size_t>(bytes, 16); return at; };
    auto bytes_of = [](const auto& v) { return v.size() * sizeof(v[0]); };
    const size_t at_chains = place(bytes_of(chains)), at_clen = place(bytes_of(clen)), at_unexp = place(bytes_of(unexp)), at_ninf = place(bytes_of(ninf)), at_del = place(bytes_of(has_del)),
                 at_lasso = place(bytes_of(hap_lasso)), at_norm = place(bytes_of(hap_norm)), at_rwo = place(bytes_of(rwo)), at_ed = place(bytes_of(ed32)), at_ov = place(bytes_of(ovv)),
                 at_opt = place(bytes_of(optimum)), at_worst = place(bytes_of(worst)), at_lf = place(bytes_of(ln_fact)), at_lp = place(bytes_of(ln_p));
    uint8_t* d_in = (uint8_t*)sp_pool(ctx, "k5_in", in_bytes + 16); uint8_t* h_in = (uint8_t*)sp_host_pool(ctx, "k5_in", in_bytes + 16);
    if (!d_in || !h_in) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "chain pair buffers");
    auto put = [&](size_t at, const auto& v) { if (!v.empty()) std::memcpy(h_in + at, v.data(), v.size() * sizeof(v[0])); };
    put(at_chains, chains); put(at_clen, clen); put(at_unexp, unexp); put(at_ninf, ninf); put(at_del, has_del); put(at_lasso, hap_lasso); put(at_norm, hap_norm); put(at_rwo, rwo);
    put(at_ed, ed32); put(at_ov, ovv); put(at_opt, optimum); put(at_worst, worst); put(at_lf, ln_fact); put(at_lp, ln_p);
    (void)hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, ctx->stream);
    uint8_t* d_chains = d_in + at_chains; int32_t* d_clen = (int32_t*)(d_in + at_clen);
    uint32_t* d_unexp = (uint32_t*)(d_in + at_unexp); uint32_t* d_ninf = (uint32_t*)(d_in + at_ninf);
    uint8_t* d_del = d_in + at_del; uint8_t* d_lasso = d_in + at_lasso; uint8_t* d_norm = d_in + at_norm;
    int32_t* d_rwo = (int32_t*)(d_in + at_rwo); uint32_t* d_ed = (uint32_t*)(d_in + at_ed); double* d_ov = (double*)(d_in + at_ov);
    uint64_t* d_opt = (uint64_t*)(d_in + at_opt); uint64_t* d_worst = (uint64_t*)(d_in + at_worst);
    double* d_lf = (double*)(d_in + at_lf); double* d_lp = (double*)(d_in + at_lp);
    // the results likewise come down in one copy: [global best, pairs scored | block scores | block pair ids | block edit distances | block components]
    const size_t out_bytes = 16 + (size_t)blocks * (8 + 8 + 8 + 40);
    uint8_t* d_out = (uint8_t*)sp_pool(ctx, "k5_out", out_bytes); uint8_t* h_out = (uint8_t*)sp_host_pool(ctx, "k5_out", out_bytes);
    unsigned long long* d_gb = (unsigned long long*)d_out;
    double* d_bs = (double*)(d_out + 16); unsigned long long* d_bp = (unsigned long long*)(d_out + 16 + (size_t)blocks * 8);
    unsigned long long* d_be = (unsigned long long*)(d_out + 16 + (size_t)blocks * 16); double* d_bc = (double*)(d_out + 16 + (size_t)blocks * 24);
    if (!d_out || !h_out) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "chain pair buffers");
    if (!d_chains || !d_clen || !d_unexp || !d_ninf || !d_del || !d_lasso || !d_norm || !d_rwo || !d_ed || !d_ov || !d_opt || !d_worst || !d_lf || !d_lp ||
        !d_gb || !d_bs || !d_bp || !d_bc || !d_be) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "chain pair buffers");
    unsigned long long* d_tb = (unsigned long long*)sp_pool(ctx, "k5_tab_best", std::max<size_t>(1, (size_t)P * R) * 8);
    unsigned long long* d_tm = (unsigned long long*)sp_pool(ctx, "k5_tab_mask", std::max<size_t>(1, (size_t)P * R) * 8);
    if (!d_tb || !d_tm) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "chain pair tables");
    if ((size_t)P * R) {
        ProfScope ps(ctx, "k5_chain_reads", (uint64_t)P * R);
        hipLaunchKernelGGL(k5_chain_read_kernel, dim3((unsigned)(((size_t)P * R + 255) / 256)), dim3(256), 0, ctx->stream, P, R, H, maxlen, d_chains, d_clen, d_rwo, d_ed, d_tb, d_tm);
    }
    const unsigned long long init[2] = {0x7FF0000000000000ull /* +inf */, 0ull};
    (void)hipMemcpyAsync(d_gb, init, 16, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemsetAsync(d_bp, 0xFF, blocks * 8, ctx->stream);
    {
        ProfScope ps(ctx, "k5_pairs", n_pairs);
        if (block_per_pair)
            hipLaunchKernelGGL(k5_pair_block_kernel, dim3((unsigned)blocks), dim3(K5_TILE), 0, ctx->stream, pc, d_chains, d_clen, d_unexp, d_ninf, d_del, d_lasso, d_norm,
                               d_rwo, d_ov, d_tb, d_tm, d_opt, d_worst, d_lf, lf_n, d_lp, max_total + 1, d_gb, d_bs, d_bp, d_bc, d_be, d_gb + 1);
        else {
        (void)hipFuncSetAttribute((const void*)k5_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)256 * H * sizeof(double)));
        hipLaunchKernelGGL(k5_pair_kernel, dim3((unsigned)blocks), dim3(256), (size_t)256 * H * sizeof(double), ctx->stream, pc, d_chains, d_clen, d_unexp, d_ninf, d_del, d_lasso, d_norm,
                           d_rwo, d_ed, d_ov, d_tb, d_tm, d_opt, d_worst, d_lf, lf_n, d_lp, max_total + 1, d_gb, d_bs, d_bp, d_bc, d_be, d_gb + 1);
        }
        if (hipGetLastError() != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, "k5 launch failed");
    }
    (void)hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
    const unsigned long long* gb = (const unsigned long long*)h_out;
    const double* bs = (const double*)(h_out + 16); const unsigned long long* bp = (const unsigned long long*)(h_out + 16 + (size_t)blocks * 8);
    const unsigned long long* be = (const unsigned long long*)(h_out + 16 + (size_t)blocks * 16); const double* bc = (const double*)(h_out + 16 + (size_t)blocks * 24);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, std::string("chain pair: ") + hipGetErrorString(e));
    res->n_pairs_scored = gb[1];
    int64_t wb = -1;
    for (uint64_t b = 0; b < blocks; ++b) {
        if (bp[b] == 0xFFFFFFFFFFFFFFFFull) continue;
        if (wb < 0 || bs[b] < bs[wb] || (bs[b] == bs[wb] && bp[b] < bp[wb])) wb = (int64_t)b;
    }
    if (wb < 0) return SP_ERR_NO_SCORE_PAIRS;
    // pair id -> (i, j)
    uint64_t pid = bp[wb]; int wi = 0;
    while (wi + 1 < P && (uint64_t)(wi + 1) * P - (uint64_t)(wi + 1) * wi / 2 <= pid) ++wi;
    const int wj = (int)(pid - ((uint64_t)wi * P - (uint64_t)wi * (wi - 1) / 2)) + wi;
    res->index1 = wi; res->index2 = wj;
    res->score = bs[wb]; res->ln_ed_penalty = bc[wb * 5]; res->mn_llh_penalty = bc[wb * 5 + 1]; res->allele_expected_penalty = bc[wb * 5 + 2];
    res->unexpected_chain_penalty = bc[wb * 5 + 3]; res->inferred_chain_penalty = bc[wb * 5 + 4]; res->edit_distance = be[wb];
    const std::vector<int>* a = &possible[wi]; const std::vector<int>* b = &possible[wj];
    if (*b < *a) std::swap(a, b);                              // best_chain_pair.sort() (chaining.rs:568-573)
    res->n1 = (int32_t)a->size(); res->n2 = (int32_t)b->size();
    for (size_t x = 0; x < a->size(); ++x) res->chain1[x] = (*a)[x];
    for (size_t x = 0; x < b->size(); ++x) res->chain2[x] = (*b)[x];
    return SP_OK;
}

// =============================================================================================
// K3 / K4: every (query, target) placement through the anchor + WFA-cell kernels, reference logic on the host
// =============================================================================================
#define CYP_TOPK      4
#define CYP_MIN_VOTES 4

int sp_seqset_build_index(sp_ctx* ctx, sp_seqset* s);

// frac_cap > 0: a cell gives up beyond floor(frac_cap * |A|) + 1 edits (K3: hits above max_ed_frac = 0.05 of the template are dropped
// anyway, src/cyp2d6/haplotyper.rs:160,228-232, and nm <= 0.05 * aligned span <= 0.05 * |A|); otherwise the library-wide cap
__global__ void cyp_build_cells_kernel(const uint32_t* __restrict__ a_idx, const uint32_t* __restrict__ b_idx,
                                       const int32_t* __restrict__ diag, const int32_t* __restrict__ votes, uint64_t n_pairs, int topk,
                                       int min_votes, const int32_t* __restrict__ a_len, double frac_cap, CellDesc* __restrict__ cells) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs * (uint64_t)topk) return;
    const uint64_t p = i / (uint64_t)topk;
    CellDesc c; c.a = a_idx[p]; c.b = b_idx[p]; c.max_ed = SP_MAX_ED; c.b_lo = 0; c.b_hi = -1;
    if (frac_cap > 0.0) { const int cap = (int)(frac_cap * (double)a_len[c.a]) + 1; c.max_ed = cap < SP_MAX_ED ? cap : SP_MAX_ED; }
    c.diag = votes[i] >= min_votes ? diag[i] : SP_NO_DIAG;       // anchor: b_pos - a_pos with A the indexed side = the cell's A
    cells[i] = c;
}

// placements of a list of (A, B) pairs (A = indexed query side, B = target side), result[pair][k]; a_period as in sp_launch_anchor
static int cyp_align_pairs(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, const std::vector<uint32_t>& ai, const std::vector<uint32_t>& bi, uint32_t a_period,
                           int topk, double frac_cap, int retry_wide, const char* prof, std::vector<sp_aln>& out, std::vector<int32_t>* diag_out = nullptr, std::vector<int32_t>* votes_out = nullptr) {
    const uint64_t n_pairs = ai.size(), n_cells = n_pairs * (uint64_t)topk;
    out.assign(n_cells, sp_aln{});
    if (n_pairs == 0) return SP_OK;
    int rc = sp_seqset_build_index(ctx, const_cast<sp_seqset*>(A));
    if (rc) return rc;
    uint32_t* d_a = (uint32_t*)sp_pool(ctx, "cyp_a", n_pairs * 4); uint32_t* d_b = (uint32_t*)sp_pool(ctx, "cyp_b", n_pairs * 4);
    int32_t* d_d = (int32_t*)sp_pool(ctx, "cyp_d", n_cells * 4); int32_t* d_v = (int32_t*)sp_pool(ctx, "cyp_v", n_cells * 4);
    CellDesc* d_cells = (CellDesc*)sp_pool(ctx, "cyp_cells", n_cells * sizeof(CellDesc));
    sp_aln* d_alns = (sp_aln*)sp_pool(ctx, "cyp_alns", n_cells * sizeof(sp_aln));
    if (!d_a || !d_b || !d_d || !d_v || !d_cells || !d_alns) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "cyp placement buffers");
    (void)hipMemcpyAsync(d_a, ai.data(), n_pairs * 4, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(d_b, bi.data(), n_pairs * 4, hipMemcpyHostToDevice, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    rc = sp_launch_anchor(ctx, A, B, d_a, d_b, n_pairs, d_d, d_v, topk, "anchor", a_period);
    if (rc) return rc;
    hipLaunchKernelGGL(cyp_build_cells_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, ctx->stream, d_a, d_b, d_d, d_v, n_pairs, topk, CYP_MIN_VOTES, A->d_len, frac_cap, d_cells);
    rc = sp_launch_cells(ctx, A, B, d_cells, n_cells, d_alns, nullptr, 0, prof, retry_wide);
    if (rc) return rc;
    (void)hipMemcpyAsync(out.data(), d_alns, n_cells * sizeof(sp_aln), hipMemcpyDeviceToHost, ctx->stream);
    if (diag_out && votes_out) {                          // (the anchors too: the region search looks for reads whose templates were all lost over a stretch, cyp_find_regions)
        diag_out->resize(n_cells); votes_out->resize(n_cells);
        (void)hipMemcpyAsync(diag_out->data(), d_d, n_cells * 4, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipMemcpyAsync(votes_out->data(), d_v, n_cells * 4, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, std::string("cyp placements: ") + hipGetErrorString(e));
    return SP_OK;
}

// the pair list of "every A against every B" made on the device (pair b * nA + a)
__global__ void cyp_all_pairs_kernel(uint32_t nA, uint64_t n_pairs, uint32_t* __restrict__ a_idx, uint32_t* __restrict__ b_idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    a_idx[i] = (uint32_t)(i % nA); b_idx[i] = (uint32_t)(i / nA);
}
// all |A| x |B| placements for the region search, result[b][a][k], in PINNED host memory of the context (valid until the context's next region search): the grid of a
// 2,000-read sample is 12.5 MB of alignments + 2.5 MB of anchors -- into a fresh std::vector that was a zero-fill, a page fault per 4 KB and a device-to-host copy through
// the runtime's bounce buffers, 0.6 - 9 ms of the search depending on what the allocator had at hand (and several samples' searches at once share the process's page tables)
static int cyp_align_all_pinned(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, int topk, double frac_cap, const char* prof, sp_aln** out, int32_t** diag_out, int32_t** votes_out) {
    const uint64_t nA = A->n, n_pairs = nA * (uint64_t)B->n, n_cells = n_pairs * (uint64_t)topk;
    *out = nullptr; *diag_out = nullptr; *votes_out = nullptr;
    if (n_pairs == 0) return SP_OK;
    int rc = sp_seqset_build_index(ctx, const_cast<sp_seqset*>(A));
    if (rc) return rc;
    uint32_t* d_a = (uint32_t*)sp_pool(ctx, "cyp_a", n_pairs * 4); uint32_t* d_b = (uint32_t*)sp_pool(ctx, "cyp_b", n_pairs * 4);
    int32_t* d_d = (int32_t*)sp_pool(ctx, "cyp_d", n_cells * 4); int32_t* d_v = (int32_t*)sp_pool(ctx, "cyp_v", n_cells * 4);
    CellDesc* d_cells = (CellDesc*)sp_pool(ctx, "cyp_cells", n_cells * sizeof(CellDesc));
    sp_aln* d_alns = (sp_aln*)sp_pool(ctx, "cyp_alns", n_cells * sizeof(sp_aln));
    sp_aln* h_alns = (sp_aln*)sp_host_pool(ctx, "k3_alns_host", n_cells * sizeof(sp_aln));
    int32_t* h_d = (int32_t*)sp_host_pool(ctx, "k3_diag_host", n_cells * 4); int32_t* h_v = (int32_t*)sp_host_pool(ctx, "k3_votes_host", n_cells * 4);
    if (!d_a || !d_b || !d_d || !d_v || !d_cells || !d_alns || !h_alns || !h_d || !h_v) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "cyp placement buffers");
    hipLaunchKernelGGL(cyp_all_pairs_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)nA, n_pairs, d_a, d_b);
    rc = sp_launch_anchor(ctx, A, B, d_a, d_b, n_pairs, d_d, d_v, topk, "anchor", (uint32_t)nA);
    if (rc) return rc;
    hipLaunchKernelGGL(cyp_build_cells_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, ctx->stream, d_a, d_b, d_d, d_v, n_pairs, topk, CYP_MIN_VOTES, A->d_len, frac_cap, d_cells);
    rc = sp_launch_cells(ctx, A, B, d_cells, n_cells, d_alns, nullptr, 0, prof, 0);
    if (rc) return rc;
    (void)hipMemcpyAsync(h_alns, d_alns, n_cells * sizeof(sp_aln), hipMemcpyDeviceToHost, ctx->stream);
    (void)hipMemcpyAsync(h_d, d_d, n_cells * 4, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipMemcpyAsync(h_v, d_v, n_cells * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, std::string("cyp placements: ") + hipGetErrorString(e));
    *out = h_alns; *diag_out = h_d; *votes_out = h_v;
    return SP_OK;
}

// all |A| x |B| placements, result[b][a][k]
static int cyp_align_all(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, int topk, double frac_cap, int retry_wide, const char* prof, std::vector<sp_aln>& out,
                         std::vector<int32_t>* diag_out = nullptr, std::vector<int32_t>* votes_out = nullptr) {
    const uint64_t nA = A->n, nB = B->n, n_pairs = nA * nB;
    std::vector<uint32_t> ai(n_pairs), bi(n_pairs);
    for (uint64_t b = 0; b < nB; ++b) for (uint64_t a = 0; a < nA; ++a) { ai[b * nA + a] = (uint32_t)a; bi[b * nA + a] = (uint32_t)b; }
    return cyp_align_pairs(ctx, A, B, ai, bi, (uint32_t)nA, topk, frac_cap, retry_wide, prof, out, diag_out, votes_out);
}

static inline double cyp_score(int seq_len, int nm, int unmapped, bool penalize) {        // MappingStats::custom_score (data_types/mapping.rs:60-84)
    const int len = penalize ? seq_len : seq_len - unmapped;
    double num = (double)(nm + (penalize ? unmapped : 0)); if (num < 0.1) num = 0.1;
    return num / (double)len;
}

static void cyp_weights_from_alns(uint32_t C, const int32_t* cons_len, const uint8_t* allowed, uint32_t n_segments, const int32_t* seg_len, const std::vector<sp_aln>& alns,
                                  uint64_t* ed, double* ov, uint8_t* kept, const sp_affine_aln* af = nullptr);

// The placements that can decide a segment's weights -- those whose edits + unmapped bases are within CYP_K4_NEAR of the segment's smallest over the consensuses that
// count -- carry the reference's numbers: the placement re-scored with minimap2's two-piece affine scores on the 256 diagonals around it (sp_rescore_mappings: the rows
// around the clustered edits only; oracle/cyp.c osp_cyp_weight_sequence states the same rule with the DP over all rows).  A segment against the consensus of another
// gene copy is hundreds of clustered edits away and never near the minimum: those keep the unit-cost count, a lower bound of the other.
constexpr int CYP_K4_NEAR = 16;
struct K4Pick { uint32_t a, b; const sp_aln* al; sp_affine_aln* out; };
// grid alns[s * C + c]: segment s is sequence b0 + s of the segment set, consensus c sequence a_of_c[c] (or a0 + c) of the consensus set; af: the same grid, zeroed here
static void cyp_pick_near_min(uint32_t C, const uint8_t* allowed, uint32_t n_segments, const int32_t* seg_len, const std::vector<sp_aln>& alns, std::vector<sp_affine_aln>& af,
                              uint32_t a0, const int32_t* a_of_c, uint32_t b0, std::vector<K4Pick>& picks) {
    af.assign((size_t)n_segments * C, sp_affine_aln{});
    for (uint32_t s = 0; s < n_segments; ++s) {
        int64_t best = INT64_MAX;
        for (uint32_t c = 0; c < C; ++c) {
            const sp_aln& al = alns[(size_t)s * C + c];
            if (!allowed[c] || !al.ok) continue;
            best = std::min<int64_t>(best, (int64_t)al.nm + (seg_len[s] - (al.b_end - al.b_start)));
        }
        if (best == INT64_MAX) continue;
        for (uint32_t c = 0; c < C; ++c) {
            const sp_aln& al = alns[(size_t)s * C + c];
            if (!allowed[c] || !al.ok || (int64_t)al.nm + (seg_len[s] - (al.b_end - al.b_start)) > best + CYP_K4_NEAR) continue;
            picks.push_back(K4Pick{ a_of_c ? (uint32_t)a_of_c[c] : a0 + c, b0 + s, &al, &af[(size_t)s * C + c] });
        }
    }
}
static int cyp_rescore_picks(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, const std::vector<K4Pick>& picks) {
    const uint64_t n = picks.size();
    if (n == 0 || !ctx->mm2_rescore) return SP_OK;
    std::vector<CellDesc> cells(n); std::vector<sp_aln> ref(n); std::vector<sp_affine_aln> out(n);
    for (uint64_t x = 0; x < n; ++x) {
        const sp_aln& al = *picks[x].al;
        cells[x] = CellDesc{ picks[x].a, picks[x].b, ((al.b_start - al.a_start) + (al.b_end - al.a_end)) / 2, 320, 0, -1 };
        ref[x] = al;
    }
    CellDesc* d_cells = (CellDesc*)sp_pool(ctx, "k4_af_cells", n * sizeof(CellDesc));
    sp_aln* d_ref = (sp_aln*)sp_pool(ctx, "k4_af_ref", n * sizeof(sp_aln));
    sp_affine_aln* d_af = (sp_affine_aln*)sp_pool(ctx, "k4_af_out", n * sizeof(sp_affine_aln));
    if (!d_cells || !d_ref || !d_af) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "weights: re-score buffers");
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_cells, cells.data(), n * sizeof(CellDesc), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_ref, ref.data(), n * sizeof(sp_aln), hipMemcpyHostToDevice, ctx->stream));
    const sp_affine_opts ao = { 1, 4, 6, 2, 26, 1, 1 };
    const int rc = sp_rescore_mappings(ctx, A, B, d_cells, d_ref, n, false, ao, 256, d_af, "k4_af", 320);
    if (rc != SP_OK) return rc;
    SP_HIP_CHECK(ctx, hipMemcpyAsync(out.data(), d_af, n * sizeof(sp_affine_aln), hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t x = 0; x < n; ++x) *picks[x].out = out[x];
    return SP_OK;
}

extern "C" int32_t sp_cyp_weight_segments(sp_ctx* ctx, const sp_seqset* consensus, const uint8_t* allowed, const sp_seqset* segments,
                                          uint64_t* ed, double* ov, uint8_t* kept) {
    if (!ctx || !consensus || !segments || (consensus->n && !allowed) || (segments->n && (!ed || !ov || !kept))) return SP_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    std::vector<sp_aln> alns;
    int rc = cyp_align_all(ctx, consensus, segments, 1, 0.0, 1, "k4_weight_cells", alns);
    if (rc) return rc;
    std::vector<sp_affine_aln> af; std::vector<K4Pick> picks;
    if (ctx->mm2_rescore) {
        cyp_pick_near_min(consensus->n, allowed, segments->n, segments->h_len.data(), alns, af, 0, nullptr, 0, picks);
        rc = cyp_rescore_picks(ctx, consensus, segments, picks);
        if (rc) return rc;
    }
    cyp_weights_from_alns(consensus->n, consensus->h_len.data(), allowed, segments->n, segments->h_len.data(), alns, ed, ov, kept, af.empty() ? nullptr : af.data());
    return SP_OK;
}

// the host half of sp_cyp_weight_segments: alns[segment * C + consensus] are the placements of every segment on every consensus
static void cyp_weights_from_alns(uint32_t C, const int32_t* cons_len, const uint8_t* allowed, uint32_t n_segments, const int32_t* seg_len, const std::vector<sp_aln>& alns,
                                  uint64_t* ed, double* ov, uint8_t* kept, const sp_affine_aln* af) {
    for (uint32_t s = 0; s < n_segments; ++s) {
        const int seq_len = seg_len[s];
        double min_ed_frac = 1.0;
        for (uint32_t c = 0; c < C; ++c) {
            uint64_t& e = ed[(size_t)s * C + c]; double& o = ov[(size_t)s * C + c];
            e = (uint64_t)seq_len; o = 0.0;                                  // "deleted" default (chaining.rs:40-41)
            sp_aln al = alns[(size_t)s * C + c];
            if (!allowed[c] || !al.ok) continue;
            if (af && af[(size_t)s * C + c].score > 0) {                     // the placement with the reference's numbers (cyp_pick_near_min)
                const sp_affine_aln& r = af[(size_t)s * C + c];
                al.nm = r.nm; al.a_start = r.a_start; al.a_end = r.a_end; al.b_start = r.b_start; al.b_end = r.b_end;
            }
            const int con_len = cons_len[c];
            const uint64_t nm = (uint64_t)al.nm, unmapped = (uint64_t)(seq_len - (al.b_end - al.b_start));
            const uint64_t match_score = nm + unmapped;
            const double overlap_score = 1.0 - (double)(al.a_start + (con_len - al.a_end)) / (double)con_len;
            if (match_score < e || (match_score == e && overlap_score > o)) {
                e = match_score; o = overlap_score;
                const double sc = cyp_score(seq_len, (int)nm, (int)unmapped, true);
                if (sc < min_ed_frac) min_ed_frac = sc;
            }
        }
        kept[s] = min_ed_frac <= 0.05 ? 1 : 0;                                // maximum_allowed_ed (chaining.rs:45,96-102)
    }
}

// The host's share of the region search works read by read (which placements exist, which are critical, which survive the collapse): the reads in K3_HOST_THREADS ranges, a
// thread each, every range's output appended in read order -- the lists are the ones one thread would make (4.1 -> 1.5 ms of a 2,000-read sample's 10)
constexpr uint32_t K3_HOST_THREADS = 4, K3_HOST_MIN_READS = 256;
template <class F> static void k3_read_ranges(uint32_t n_reads, F&& fn) {                                   // fn(range index, first read, one past the last)
    const uint32_t nt = n_reads >= K3_HOST_MIN_READS ? K3_HOST_THREADS : 1, per = (n_reads + nt - 1) / nt;
    std::thread th[K3_HOST_THREADS]; bool started[K3_HOST_THREADS] = {};
    for (uint32_t t = 1; t < nt; ++t) {
        const uint32_t lo = std::min(n_reads, t * per), hi = std::min(n_reads, lo + per);
        try { th[t] = std::thread([&fn, t, lo, hi]() { fn(t, lo, hi); }); started[t] = true; } catch (...) { fn(t, lo, hi); }      // (no thread to be had: this one does the range)
    }
    fn(0u, 0u, std::min(n_reads, per));
    for (uint32_t t = 1; t < nt; ++t) if (started[t]) th[t].join();
}
static int32_t cyp_find_regions(sp_ctx* ctx, const sp_seqset* templates, const int32_t* template_type, const sp_seqset* reads,
                                double max_missing_frac, sp_region_hit* hits, uint64_t hits_cap, uint64_t* n_hits, bool rescore);
extern "C" int32_t sp_cyp_find_regions(sp_ctx* ctx, const sp_seqset* templates, const int32_t* template_type, const sp_seqset* reads,
                                       double max_missing_frac, sp_region_hit* hits, uint64_t hits_cap, uint64_t* n_hits) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    return cyp_find_regions(ctx, templates, template_type, reads, max_missing_frac, hits, hits_cap, n_hits, ctx->mm2_rescore);
}
// rescore (context option "mm2_rescore", the default): the numbers of a hit are the reference's -- template (minimap2's query) against read (its target), two-piece affine gaps
// and end clipping on the 256 diagonals around the placement's own (sp_rescore_mappings; placements whose edits all stand alone keep their counts without a DP) -- and start / end /
// nm / unmapped / clips of a hit ARE those numbers: what minimap2 reports for the mapping, what the segments are cut from and what the missing-fraction filter sees
// (oracle/cyp.c osp_cyp_find_base_type_ex states the same with the DP over all rows).
// Round 6: the decisions in front of the hit list see those numbers too wherever they can decide.  A placement is re-scored BEFORE the edit-fraction filter and the collapse when
// it is CRITICAL (among those with a score of at most K3_CAP_HI by the library's count: cells give up one edit past the 0.05 cap): no other placement on the same read that overlaps it by
// more than K3_OVL (the collapse asks for 0.9) clearly beats it -- the rival's score, taken as the collapse would compare the two, lower by more than a quarter of itself plus 0.001
// (less than that, end clipping or an affine gap can turn the pair round).  Every other placement -- the other gene copy's templates, hundreds of clustered edits behind a rival: re-scoring
// ALL 45,365 placements of a 2,000-read sample cost 39 ms (round 5) -- keeps the library's counts through the filter and the collapse, is re-scored if it survives them and then has
// to pass the filter on its re-scored numbers once more.  The drivers inside the library take the same path.
constexpr double K3_CAP_HI = 0.056, K3_OVL = 0.85;
constexpr int K3_RETRY_VOTES = 128, K3_RETRY_HOLE = 500;
static int32_t cyp_find_regions(sp_ctx* ctx, const sp_seqset* templates, const int32_t* template_type, const sp_seqset* reads,
                                double max_missing_frac, sp_region_hit* hits, uint64_t hits_cap, uint64_t* n_hits, bool rescore) {
    if (!ctx || !templates || !reads || !n_hits || (templates->n && !template_type) || (hits_cap && !hits)) return SP_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    *n_hits = 0;
    HostMarks hm(ctx);
    sp_aln* alns = nullptr; int32_t* adiag = nullptr; int32_t* avotes = nullptr;
    int rc = cyp_align_all_pinned(ctx, templates, reads, CYP_TOPK, 0.05, "k3_region_cells", &alns, &adiag, &avotes);
    if (rc) return rc;
    hm.mark("host:k3_cells");
    if (!alns) return SP_OK;                                                                                 // (no templates or no reads: no hits)
    const uint32_t T = templates->n;
    auto penalized_type = [](int t) { return t == SP_CYP_DELETION || t == SP_CYP_REP6 || t == SP_CYP_REP7; };   // haplotyper.rs:185-191
    // THE WIDE-BAND RETRY (round 6).  minimap2 chains a template across a 40 - 120 base insertion or deletion in the read (bw 500, max_gap 10000); the 64-diagonal cell leaves
    // its band there and is lost -- and so is every other template over that stretch of the read.  A cell does not know whether it ran out of band or of edits (17 % of the
    // strongly anchored pairs are lost: the other gene copy's templates, which drift out of the band over a few kb and lose the collapse to the right copy's anyway), but the READ
    // shows it: a stretch of it where a template anchors strongly (>= K3_RETRY_VOTES 16-mer votes on its best diagonal) and NO template was placed.  Every template lost over such
    // a hole (>= K3_RETRY_HOLE uncovered bases of its expected span) runs once more on 256 diagonals around its best anchor, same edit cap; reads without a hole -- all reads
    // of the six scenarios -- cost nothing.  (oracle/cyp.c osp_cyp_find_base_type_ex states the same rule.)
    if (rescore) {
        std::vector<CellDesc> retry; std::vector<uint32_t> retry_at;
        std::vector<CellDesc> retry_of[K3_HOST_THREADS]; std::vector<uint32_t> retry_at_of[K3_HOST_THREADS];
        k3_read_ranges(reads->n, [&](uint32_t part, uint32_t r_lo, uint32_t r_hi) {
        std::vector<CellDesc>& retry = retry_of[part]; std::vector<uint32_t>& retry_at = retry_at_of[part];
        std::vector<std::pair<int, int>> iv;
        for (uint32_t r = r_lo; r < r_hi; ++r) {
            const int rlen = reads->h_len[r];
            if (rlen == 0) continue;
            iv.clear();
            bool any_lost = false;
            for (uint32_t t = 0; t < T; ++t) {
                bool placed = false;
                for (int k = 0; k < CYP_TOPK; ++k) { const sp_aln& al = alns[((size_t)r * T + t) * CYP_TOPK + k]; if (al.ok) { placed = true; iv.push_back({ al.b_start, al.b_end }); } }
                if (!placed && avotes[((size_t)r * T + t) * CYP_TOPK] >= K3_RETRY_VOTES) any_lost = true;
            }
            if (!any_lost) continue;
            std::sort(iv.begin(), iv.end());
            for (uint32_t t = 0; t < T; ++t) {
                const size_t c0 = ((size_t)r * T + t) * CYP_TOPK;
                bool placed = false;
                for (int k = 0; k < CYP_TOPK; ++k) placed = placed || alns[c0 + k].ok;
                if (placed || avotes[c0] < K3_RETRY_VOTES) continue;
                const int tlen = templates->h_len[t], d0 = adiag[c0];                                    // read position - template position
                const int s0 = std::max(0, d0), e0 = std::min(rlen, d0 + tlen);
                int uncovered = 0, at = s0;
                for (const auto& x : iv) { if (x.second <= at) continue; if (x.first >= e0) break; if (x.first > at) uncovered += x.first - at; at = std::max(at, x.second); if (at >= e0) break; }
                if (at < e0) uncovered += e0 - at;
                if (uncovered < K3_RETRY_HOLE) continue;
                int cap = (int)(0.05 * (double)tlen) + 1; if (cap > SP_MAX_ED) cap = SP_MAX_ED;
                retry.push_back(CellDesc{ t, r, d0, cap, 0, -1 }); retry_at.push_back((uint32_t)c0);
            }
        }
        });
        for (uint32_t part = 0; part < K3_HOST_THREADS; ++part) { retry.insert(retry.end(), retry_of[part].begin(), retry_of[part].end()); retry_at.insert(retry_at.end(), retry_at_of[part].begin(), retry_at_of[part].end()); }
        if (!retry.empty()) {
            const size_t nr = retry.size();
            CellDesc* d_rc = (CellDesc*)sp_pool(ctx, "k3_retry_cells", nr * sizeof(CellDesc)); sp_aln* d_ro = (sp_aln*)sp_pool(ctx, "k3_retry_alns", nr * sizeof(sp_aln));
            if (!d_rc || !d_ro) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "find_regions: retry buffers");
            std::vector<sp_aln> got(nr);
            SP_HIP_CHECK(ctx, hipMemcpyAsync(d_rc, retry.data(), nr * sizeof(CellDesc), hipMemcpyHostToDevice, ctx->stream));
            SP_HIP_CHECK(ctx, hipMemsetAsync(d_ro, 0, nr * sizeof(sp_aln), ctx->stream));
            { ProfScope ps(ctx, "k3_retry_wide", nr); rc = sp_launch_cells_wide(ctx, templates, reads, d_rc, nr, d_ro); }
            if (rc != SP_OK) return rc;
            SP_HIP_CHECK(ctx, hipMemcpyAsync(got.data(), d_ro, nr * sizeof(sp_aln), hipMemcpyDeviceToHost, ctx->stream));
            SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            for (size_t x = 0; x < nr; ++x) if (got[x].ok) alns[retry_at[x]] = got[x];
            if (ctx->profiling) ctx->prof["k3_retried_pairs"].cells += nr;
        }
    }
    auto own_score = [&](const sp_region_hit& h) { return cyp_score(h.seq_len, h.nm, h.unmapped, penalized_type(template_type[h.template_idx])); };
    auto overlap = [](const sp_region_hit& u, const sp_region_hit& v) {                                       // overlap_score (haplotyper.rs:877-892)
        const int min_end = std::min(u.end, v.end), max_start = std::max(u.start, v.start);
        return max_start < min_end ? (double)(min_end - max_start) / std::min((double)(u.end - u.start), (double)(v.end - v.start)) : 0.0;
    };
    // a list of placements (hit + which alignment it is) re-scored in place
    struct Pl { sp_region_hit h; uint32_t which; uint8_t done; };
    auto rescore_list = [&](std::vector<Pl*>& list, const char* prof) -> int {
        const uint64_t nc = list.size();
        if (!nc) return SP_OK;
        // (descriptions and results through pinned memory of the context, filled and read back by the host threads in ranges of the list)
        CellDesc* cells = (CellDesc*)sp_host_pool(ctx, "k3_af_cells_host", nc * sizeof(CellDesc));
        sp_aln* ref = (sp_aln*)sp_host_pool(ctx, "k3_af_ref_host", nc * sizeof(sp_aln));
        sp_affine_aln* af = (sp_affine_aln*)sp_host_pool(ctx, "k3_af_out_host", nc * sizeof(sp_affine_aln));
        CellDesc* d_cells = (CellDesc*)sp_pool(ctx, "k3_af_cells", nc * sizeof(CellDesc));
        sp_aln* d_ref = (sp_aln*)sp_pool(ctx, "k3_af_ref", nc * sizeof(sp_aln));
        sp_affine_aln* d_af = (sp_affine_aln*)sp_pool(ctx, "k3_af_out", nc * sizeof(sp_affine_aln));
        if (!cells || !ref || !af || !d_cells || !d_ref || !d_af) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "find_regions: re-score buffers");
        k3_read_ranges((uint32_t)nc, [&](uint32_t, uint32_t lo, uint32_t hi) {
            for (uint64_t x = lo; x < hi; ++x) {
                const sp_region_hit& h = list[x]->h; const sp_aln& al = alns[list[x]->which];
                cells[x] = CellDesc{ (uint32_t)h.template_idx, (uint32_t)h.read, ((al.b_start - al.a_start) + (al.b_end - al.a_end)) / 2, 320, 0, -1 };
                ref[x] = al;
            }
        });
        SP_HIP_CHECK(ctx, hipMemcpyAsync(d_cells, cells, nc * sizeof(CellDesc), hipMemcpyHostToDevice, ctx->stream));
        SP_HIP_CHECK(ctx, hipMemcpyAsync(d_ref, ref, nc * sizeof(sp_aln), hipMemcpyHostToDevice, ctx->stream));
        const sp_affine_opts ao = { 1, 4, 6, 2, 26, 1, 1 };
        const int rc2 = sp_rescore_mappings(ctx, templates, reads, d_cells, d_ref, nc, false, ao, 256, d_af, prof, 320);
        if (rc2 != SP_OK) return rc2;
        SP_HIP_CHECK(ctx, hipMemcpyAsync(af, d_af, nc * sizeof(sp_affine_aln), hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        k3_read_ranges((uint32_t)nc, [&](uint32_t, uint32_t lo, uint32_t hi) {
            for (uint64_t x = lo; x < hi; ++x) {
                sp_region_hit& h = list[x]->h;
                list[x]->done = 1;
                if (af[x].score <= 0) continue;                                                                  // (nothing aligns the reference's way: the hit keeps the library's counts)
                h.mm2_score = af[x].score; h.mm2_nm = af[x].nm; h.mm2_start = af[x].b_start; h.mm2_end = af[x].b_end; h.mm2_q_start = af[x].a_start; h.mm2_q_end = af[x].a_end;
                h.start = af[x].b_start; h.end = af[x].b_end; h.nm = af[x].nm;
                h.unmapped = h.seq_len - (af[x].a_end - af[x].a_start); h.clip_start = af[x].a_start; h.clip_end = h.seq_len - af[x].a_end;
            }
        });
        return SP_OK;
    };
    hm.mark("host:k3_retry");
    // the placements of every read by the library's counts (cells give up one edit past the cap: a little above it they still exist)
    std::vector<Pl> un; std::vector<uint32_t> first(reads->n + 1, 0);
    {
        std::vector<Pl> un_of[K3_HOST_THREADS]; uint32_t lo_of[K3_HOST_THREADS] = {}, hi_of[K3_HOST_THREADS] = {};
        k3_read_ranges(reads->n, [&](uint32_t part, uint32_t r_lo, uint32_t r_hi) {
            std::vector<Pl>& mine = un_of[part]; lo_of[part] = r_lo; hi_of[part] = r_hi;
            for (uint32_t r = r_lo; r < r_hi; ++r) {
                first[r] = (uint32_t)mine.size();                                                               // (within the range: the range's base is added below)
                if (reads->h_len[r] == 0) continue;
                for (uint32_t t = 0; t < T; ++t) for (int k = 0; k < CYP_TOPK; ++k) {
                    const uint32_t which = (uint32_t)(((size_t)r * T + t) * CYP_TOPK + k);
                    const sp_aln& al = alns[which];
                    if (!al.ok) continue;
                    const int tlen = templates->h_len[t];
                    sp_region_hit h{(int32_t)r, (int32_t)t, al.b_start, al.b_end, tlen, al.nm, tlen - (al.a_end - al.a_start), al.a_start, tlen - al.a_end, 0, 0, 0, 0, 0, 0};
                    if (own_score(h) > (rescore ? K3_CAP_HI : 0.05)) continue;                                  // max_ed_frac, :228-232
                    mine.push_back(Pl{ h, which, 0 });
                }
            }
        });
        size_t total = 0; for (uint32_t part = 0; part < K3_HOST_THREADS; ++part) total += un_of[part].size();
        un.reserve(total);
        for (uint32_t part = 0; part < K3_HOST_THREADS; ++part) {
            const uint32_t base = (uint32_t)un.size();
            for (uint32_t r = lo_of[part]; r < hi_of[part]; ++r) first[r] += base;
            un.insert(un.end(), un_of[part].begin(), un_of[part].end());
        }
    }
    first[reads->n] = (uint32_t)un.size();
    hm.mark("host:k3_list");
    if (rescore) {
        std::vector<Pl*> crit;
        std::vector<uint8_t> mark(un.size(), 1);
        // (both scores of every placement once: the pair loop below runs ~ 250 times per read)
        std::vector<double> s_own(un.size()), s_pen(un.size()); std::vector<uint8_t> is_pen(un.size());
        // (start / end / length of every placement side by side: the loop reads three ints per partner; the division of overlap() only where the spans can reach K3_OVL at all)
        std::vector<int> p_s(un.size()), p_e(un.size());
        k3_read_ranges(reads->n, [&](uint32_t, uint32_t r_lo, uint32_t r_hi) {                              // (a read's placements are its own: every range writes its own stretch of the arrays)
        for (size_t i = first[r_lo]; i < first[r_hi]; ++i) {
            const sp_region_hit& h = un[i].h;
            s_own[i] = cyp_score(h.seq_len, h.nm, h.unmapped, false); s_pen[i] = cyp_score(h.seq_len, h.nm, h.unmapped, true); is_pen[i] = penalized_type(template_type[h.template_idx]);
            p_s[i] = h.start; p_e[i] = h.end;
        }
        for (uint32_t r = r_lo; r < r_hi; ++r) for (uint32_t i = first[r]; i < first[r + 1]; ++i) {
            const int us = p_s[i], ue = p_e[i], ul = ue - us;
            for (uint32_t j = i + 1; j < first[r + 1]; ++j) {
                const int shared = std::min(ue, p_e[j]) - std::max(us, p_s[j]);
                if (shared <= 0) continue;
                const int vl = p_e[j] - p_s[j], ml = std::min(ul, vl);
                if ((int64_t)shared * 100 < (int64_t)84 * ml) continue;                        // (well under K3_OVL: no division)
                if (!((double)shared / (double)ml > K3_OVL)) continue;                         // overlap(u, v)
                const bool pen = is_pen[i] || is_pen[j];
                const double a = pen ? s_pen[i] : s_own[i], b = pen ? s_pen[j] : s_own[j];
                if (a > 1.25 * b + 0.001) mark[i] = 0;                                       // i is clearly beaten by j
                if (b > 1.25 * a + 0.001) mark[j] = 0;
            }
        }
        });
        for (size_t i = 0; i < un.size(); ++i) if (mark[i]) crit.push_back(&un[i]);
        hm.mark("host:k3_mark");
        rc = rescore_list(crit, "k3_af_crit");
        if (rc != SP_OK) return rc;
        if (ctx->profiling) ctx->prof["k3_critical_placements"].cells += crit.size();
        hm.mark("host:k3_crit_rescore");
    }
    std::vector<Pl> coll;
    std::vector<Pl> coll_of[K3_HOST_THREADS];
    k3_read_ranges(reads->n, [&](uint32_t part, uint32_t r_lo, uint32_t r_hi) {
    std::vector<Pl>& coll = coll_of[part];
    std::vector<Pl> cur_read;
    for (uint32_t r = r_lo; r < r_hi; ++r) {
        cur_read.clear();
        for (uint32_t i = first[r]; i < first[r + 1]; ++i) if (!(own_score(un[i].h) > 0.05)) cur_read.push_back(un[i]);         // the filter, on what every placement carries now
        std::stable_sort(cur_read.begin(), cur_read.end(), [](const Pl& x, const Pl& y) { return x.h.start != y.h.start ? x.h.start < y.h.start : x.h.end < y.h.end; });
        bool have = false; Pl cur{};
        for (const Pl& u : cur_read) {
            if (!have) { cur = u; have = true; continue; }
            if (overlap(u.h, cur.h) > 0.9) {
                const bool pen = penalized_type(template_type[u.h.template_idx]) || penalized_type(template_type[cur.h.template_idx]);
                const int up = template_type[u.h.template_idx] == SP_CYP_DELETION, cp = template_type[cur.h.template_idx] == SP_CYP_DELETION;
                if ((cyp_score(u.h.seq_len, u.h.nm, u.h.unmapped, pen) < cyp_score(cur.h.seq_len, cur.h.nm, cur.h.unmapped, pen) && up >= cp) || up > cp) cur = u;
            } else { coll.push_back(cur); cur = u; }
        }
        if (have) coll.push_back(cur);
    }
    });
    for (uint32_t part = 0; part < K3_HOST_THREADS; ++part) coll.insert(coll.end(), coll_of[part].begin(), coll_of[part].end());
    hm.mark("host:k3_collapse");
    if (rescore) {
        std::vector<Pl*> rest;
        for (Pl& c : coll) if (!c.done) rest.push_back(&c);
        rc = rescore_list(rest, "k3_af");
        if (rc != SP_OK) return rc;
        hm.mark("host:k3_rest_rescore");
    }
    for (const Pl& c : coll) {
        const sp_region_hit& h = c.h;
        if (rescore && own_score(h) > 0.05) continue;                                                      // (the filter once more, on the re-scored numbers)
        if (cyp_score(h.seq_len, h.nm, h.unmapped, true) > max_missing_frac) continue;                     // :303-306
        if (*n_hits < hits_cap) hits[*n_hits] = h;
        ++*n_hits;
    }
    return SP_OK;
}

// =============================================================================================
// K7: star-allele vector scoring (assign_haplotype, src/cyp2d6/haplotyper.rs:470-524)
// one wavefront per (sequence, allele): lanes stride over the variants, two ballot/popcount reductions
// =============================================================================================
__global__ __launch_bounds__(256) void k7_score_kernel(const uint8_t* __restrict__ hap, const uint8_t* __restrict__ is_vi, const uint8_t* __restrict__ states,
                                                       uint32_t n_variants, uint32_t n_alleles, uint32_t n_seqs, uint32_t* __restrict__ scores /* [seq][allele][2] */) {
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= (uint64_t)n_seqs * n_alleles) return;
    const uint32_t s = (uint32_t)(w / n_alleles), a = (uint32_t)(w % n_alleles);
    const uint8_t* hv = hap + (size_t)a * n_variants; const uint8_t* sv = states + (size_t)s * n_variants;
    uint32_t vi = 0, all = 0;
    for (uint32_t v = lane; v < n_variants; v += 64) {
        const uint8_t seq_value = sv[v], hap_value = hv[v];
        const bool is_match = seq_value <= 1 ? hap_value == seq_value : seq_value == 2;      // 3 (unset) never matches
        all += is_match; vi += is_match && is_vi[v];
    }
    for (int o = 32; o > 0; o >>= 1) { vi += __shfl_xor(vi, o); all += __shfl_xor(all, o); }
    if (lane == 0) { scores[w * 2] = vi; scores[w * 2 + 1] = all; }
}

extern "C" int32_t sp_cyp_score_alleles(sp_ctx* ctx, uint32_t n_variants, uint32_t n_alleles, const uint8_t* hap_matrix, const uint8_t* is_vi,
                                        uint32_t n_seqs, const uint8_t* states, uint32_t* best_vi, uint32_t* best_all, uint8_t* tie_mask) {
    if (!ctx || (n_alleles && n_variants && (!hap_matrix || !is_vi)) || (n_seqs && (!states || !best_vi || !best_all || !tie_mask))) return SP_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    for (uint32_t s = 0; s < n_seqs; ++s) { best_vi[s] = 0; best_all[s] = 0; }
    if (n_seqs) std::memset(tie_mask, 0, (size_t)n_seqs * n_alleles);
    if (!n_seqs || !n_alleles) return SP_OK;
    const size_t hb = (size_t)n_alleles * n_variants, sb = (size_t)n_seqs * n_variants, nw = (size_t)n_seqs * n_alleles;
    uint8_t* d_hap = (uint8_t*)sp_pool(ctx, "k7_hap", hb + 1); uint8_t* d_vi = (uint8_t*)sp_pool(ctx, "k7_vi", n_variants + 1);
    uint8_t* d_st = (uint8_t*)sp_pool(ctx, "k7_states", sb + 1); uint32_t* d_sc = (uint32_t*)sp_pool(ctx, "k7_scores", nw * 8);
    if (!d_hap || !d_vi || !d_st || !d_sc) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "score_alleles buffers");
    (void)hipMemcpyAsync(d_hap, hap_matrix, hb, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(d_vi, is_vi, n_variants, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(d_st, states, sb, hipMemcpyHostToDevice, ctx->stream);
    std::vector<uint32_t> sc(nw * 2);
    {
        ProfScope ps(ctx, "k7_score", nw);
        hipLaunchKernelGGL(k7_score_kernel, dim3((unsigned)((nw * 64 + 255) / 256)), dim3(256), 0, ctx->stream, d_hap, d_vi, d_st, n_variants, n_alleles, n_seqs, d_sc);
    }
    (void)hipMemcpyAsync(sc.data(), d_sc, nw * 8, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, std::string("score_alleles: ") + hipGetErrorString(e));
    for (uint32_t s = 0; s < n_seqs; ++s) {
        // running best from (0,0) held by the Unknown label; Greater clears the set, Equal joins it (haplotyper.rs:505-523)
        uint32_t bv = 0, ba = 0;
        for (uint32_t a = 0; a < n_alleles; ++a) {
            const uint32_t v = sc[((size_t)s * n_alleles + a) * 2], al = sc[((size_t)s * n_alleles + a) * 2 + 1];
            if (v > bv || (v == bv && al > ba)) { bv = v; ba = al; }
        }
        best_vi[s] = bv; best_all[s] = ba;
        for (uint32_t a = 0; a < n_alleles; ++a) {
            const uint32_t v = sc[((size_t)s * n_alleles + a) * 2], al = sc[((size_t)s * n_alleles + a) * 2 + 1];
            tie_mask[(size_t)s * n_alleles + a] = (v == bv && al == ba) ? 1 : 0;
        }
    }
    return SP_OK;
}


// =============================================================================================
// K9: per-variant state of typed sequences on the CYP2D6 backbone -- the role of the graph alignment in assign_haplotype
// (src/cyp2d6/haplotyper.rs:371-468; hiphase's WFAGraph is not on disk, contract in DESIGN.md section 10 / oracle/cyp.c).
// Each sequence is placed on the backbone with traceback (anchor + cell kernels).  The aligned backbone part with the database
// variants inside it is a graph: reference stretches and SITES (maximal runs of variants whose reference spans overlap), a site's
// alternatives being the subsets of its variants that do not overlap one another.  The sequence is aligned to the graph end to end
// (unit costs) on 256 diagonals around the drift of the placement; a second pass over the mirrored graph gives the cost of the rest
// from every site's end, so an alternative lies on an optimal path iff  cost through it + cost of the rest == optimum.  A variant is
// 1 / 0 when every optimal alternative of its site carries / lacks it, 2 when they disagree (conflicting traversals), 3 outside the
// aligned part.  One workgroup of two wavefronts per sequence: wave 0 runs the graph forwards, wave 1 mirrored; lane l owns diagonals
// 4l .. 4l+3; both passes read the same pools through mirrored indices.
// =============================================================================================
#define K9_DIAGS 256
#define K9_INF 30000
#define K9_SITE_MAX 8

__host__ __device__ inline int k9_lds_bytes(int L, int G, int alt_bytes, int n_sites, int n_alts) { return ((L + G + alt_bytes + 3) & ~3) + 4 * (3 * n_sites + 1 + 2 * n_alts); }

struct K9Job {
    uint32_t seq_off; int32_t L;              // the aligned part of the sequence in the byte pool
    uint32_t bb_off; int32_t G;               // the aligned part of the backbone
    int32_t k0;                               // diagonal of lane 0, index 0
    uint32_t site_first, n_sites;             // into site_so / site_eo / alt_first (alt_first has one more entry per job)
    uint32_t alt_base, n_alts;                // the job's alternatives in alt_len / alt_off
    uint32_t alt_bytes_off, alt_bytes;        // their bases: one stretch of the byte pool
    uint32_t exit_off, entry_off;             // columns (256 u16 each) in the scratch arrays
    int32_t opt;                              // (written by the kernel)
};

// r[j]: the sequence base in front of diagonal lane * 4 + j at this graph offset (255: outside the sequence)
__device__ __forceinline__ void k9_step(int (&c)[4], int x, const int (&r)[4], int L, int off, int k0, int lane) {
    // one base of the graph: diagonal d <-> k = k0 + d, read position i = off + k in front of the base
    const int next0 = spw::from_upper(c[0], K9_INF);                      // the lane above's first diagonal
    int nw[4];
    bool need = false;
    int below0;
    if (off + k0 >= 0 && off + k0 + K9_DIAGS + 1 <= L) {
        // every diagonal of the band faces a sequence base here and behind this graph base (all but the first and last columns of a job): the same
        // recurrence without the range tests, and the run up the column without asking whether it is needed -- it nearly always is: a diagonal above the
        // path's own mismatches three steps in four and gets its value back from the one below; the test, its ballot and the branch cost more than they saved
        // (1,000 cycles per step before, the step's chain of dependent instructions cut by half).  Values above K9_INF only arise as K9_INF + a few and are
        // cut back at the end of the step, before anyone compares them.
        const int xm = x < 4 ? x : 99;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int up = j < 3 ? c[j + 1] : next0;
            nw[j] = min(c[j] + (r[j] == xm ? 0 : 1), up + 1);
        }
        nw[1] = min(nw[1], nw[0] + 1); nw[2] = min(nw[2], nw[1] + 1); nw[3] = min(nw[3], nw[2] + 1);
        int f = nw[3] - 4 * lane;                                           // what this lane offers the lanes above, in lane-0 units
        // inclusive prefix minimum over the lanes, the DPP operand riding on v_min_i32 (a lane without a source is not written: it keeps its own value)
        asm("s_nop 4\n\t"
            "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(f));
        const int in = spw::from_lower(f, 1 << 20) + 4 * lane - 3;         // value arriving at this lane's first diagonal from the lanes below
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = min(min(nw[j], in + j), K9_INF);
        return;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = lane * 4 + j, i = off + k0 + d;
            int v = K9_INF;
            if (r[j] != 255 && c[j] < K9_INF) v = c[j] + ((r[j] < 4 && r[j] == x) ? 0 : 1);
            const int up = j < 3 ? c[j + 1] : next0;
            if (up < K9_INF && i + 1 >= 0 && i + 1 <= L && up + 1 < v) v = up + 1;
            nw[j] = v;
        }
        // read bases that face nothing run up the column: only when some diagonal can be improved from the one below it
        below0 = spw::from_lower(nw[3], K9_INF);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = lane * 4 + j, i = off + 1 + k0 + d, prev = j ? nw[j - 1] : below0;
            if (d > 0 && i >= 0 && i <= L && prev + 1 < nw[j]) need = true;
        }
    }
    if (__ballot(need)) {
        // inside the lane, then across lanes (a prefix minimum of value - 4 * lane), then inside again
#pragma unroll
        for (int j = 1; j < 4; ++j) { const int i = off + 1 + k0 + lane * 4 + j; if (i >= 0 && i <= L && nw[j - 1] + 1 < nw[j]) nw[j] = nw[j - 1] + 1; }
        int f = nw[3] < K9_INF ? nw[3] - 4 * lane : (1 << 20);              // what this lane offers the lanes above, in lane-0 units
        // inclusive prefix minimum over the lanes with DPP moves (row_shr 1, 2, 4, 8, then the two row broadcasts): a lane without a
        // source keeps the identity
        {
            const int id = 1 << 20;
            int t;
            t = __builtin_amdgcn_update_dpp(id, f, 0x111, 0xf, 0xf, false); f = t < f ? t : f;
            t = __builtin_amdgcn_update_dpp(id, f, 0x112, 0xf, 0xf, false); f = t < f ? t : f;
            t = __builtin_amdgcn_update_dpp(id, f, 0x114, 0xf, 0xf, false); f = t < f ? t : f;
            t = __builtin_amdgcn_update_dpp(id, f, 0x118, 0xf, 0xf, false); f = t < f ? t : f;
            t = __builtin_amdgcn_update_dpp(id, f, 0x142, 0xa, 0xf, false); f = t < f ? t : f;     // row_bcast:15 into rows 1 and 3
            t = __builtin_amdgcn_update_dpp(id, f, 0x143, 0xc, 0xf, false); f = t < f ? t : f;     // row_bcast:31 into rows 2 and 3
        }
        const int before = spw::from_lower(f, 1 << 20);                    // best offer of the lanes below
        if (lane > 0 && before < (1 << 19)) {
            const int in = before + 4 * lane - 3;                         // value arriving at this lane's first diagonal
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int i = off + 1 + k0 + lane * 4 + j; if (i >= 0 && i <= L && in + j < nw[j]) nw[j] = in + j; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = nw[j] < K9_INF ? nw[j] : K9_INF;
}

template <bool STAGED>
__global__ __launch_bounds__(128) void k9_graph_kernel(const uint8_t* __restrict__ pool, K9Job* __restrict__ jobs, const int32_t* __restrict__ site_so,
                                                       const int32_t* __restrict__ site_eo, const uint32_t* __restrict__ alt_first,
                                                       const int32_t* __restrict__ alt_len, const uint32_t* __restrict__ alt_off,
                                                       uint16_t* __restrict__ exits, uint16_t* __restrict__ entries, int32_t* __restrict__ alt_on, int lds_cap) {
    extern __shared__ __align__(16) uint8_t seq_lds[];    // the aligned stretch of the sequence (both passes read it four bases per lane and step)
    __shared__ int shift[2][K9_DIAGS];
    __shared__ int opt_s;
    K9Job& J = jobs[blockIdx.x];
    // (what is the same for a whole wavefront is said to be: the pass, the job's sizes and the band's position live in scalar registers, the branches on them are
    //  scalar branches, and the staged bases are read with LDS instructions -- through generic pointers every step waited for a flat load)
    const int lane = threadIdx.x & 63, mirror = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint8_t* bb = pool + J.bb_off;
    const int L = __builtin_amdgcn_readfirstlane(J.L), G = __builtin_amdgcn_readfirstlane(J.G), ns = __builtin_amdgcn_readfirstlane((int)J.n_sites);
    const uint8_t* S = pool + J.seq_off;
    // Everything a pass reads step by step lies in LDS: the two stretches, the bases of the job's alternatives and its site / alternative tables (a site's offsets,
    // an alternative's length and bases were dependent loads from memory before: with some 400 sites and 800 alternatives per sequence, half of the pass's time)
    const int na = __builtin_amdgcn_readfirstlane((int)J.n_alts), AB = __builtin_amdgcn_readfirstlane((int)J.alt_bytes);
    const uint32_t ab = __builtin_amdgcn_readfirstlane(J.alt_base);
    const int tab0 = (L + G + AB + 3) & ~3;               // byte offset of the tables: so[ns], eo[ns], first[ns + 1], len[na], off[na]
    constexpr bool staged = STAGED;                       // (a batch with a job too long for that stays in memory: same results, dependent loads on the way)
    int* const T_so = reinterpret_cast<int*>(seq_lds + tab0); int* const T_eo = T_so + ns; int* const T_af = T_eo + ns; int* const T_al = T_af + ns + 1; int* const T_ao = T_al + na;
    if (staged) {
        const uint8_t* ap = pool + J.alt_bytes_off;
        for (int x = threadIdx.x; x < L; x += blockDim.x) seq_lds[x] = S[x];
        for (int x = threadIdx.x; x < G; x += blockDim.x) seq_lds[L + x] = bb[x];
        for (int x = threadIdx.x; x < AB; x += blockDim.x) seq_lds[L + G + x] = ap[x];
        for (int x = threadIdx.x; x < ns; x += blockDim.x) { T_so[x] = site_so[J.site_first + x]; T_eo[x] = site_eo[J.site_first + x]; }
        for (int x = threadIdx.x; x <= ns; x += blockDim.x) T_af[x] = (int)alt_first[J.site_first + blockIdx.x + x];
        for (int x = threadIdx.x; x < na; x += blockDim.x) { T_al[x] = alt_len[ab + x]; T_ao[x] = (int)(alt_off[ab + x] - J.alt_bytes_off); }
        __syncthreads();
    }
    typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
    lds_cu8* const S3 = (lds_cu8*)(uintptr_t)spw::lds_addr(reinterpret_cast<const uint32_t*>(seq_lds));
    lds_cu8* const B3 = S3 + L;
    lds_cu8* const A3 = B3 + G;
    // (what comes out of LDS is the same for the whole wavefront: said so, the loops on it stay scalar loops)
    auto so_at = [&](int si) -> int { return staged ? __builtin_amdgcn_readfirstlane(T_so[si]) : site_so[J.site_first + si]; };
    auto eo_at = [&](int si) -> int { return staged ? __builtin_amdgcn_readfirstlane(T_eo[si]) : site_eo[J.site_first + si]; };
    auto af_at = [&](int si) -> uint32_t { return staged ? (uint32_t)__builtin_amdgcn_readfirstlane(T_af[si]) : alt_first[J.site_first + blockIdx.x + si]; };
    const int k0 = __builtin_amdgcn_readfirstlane(mirror ? (L - G) - J.k0 - (K9_DIAGS - 1) : J.k0);
    auto gbase = [&](int g) -> int { const int x = mirror ? G - 1 - g : g; return staged ? (int)B3[x] : (int)bb[x]; };   // graph base at offset g of this pass
    // The four sequence bases in front of a lane's diagonals move up by one per graph base: they are kept in registers and shifted
    // through the lanes (the top lane reads the one new base), instead of four loads per lane and step.
    auto sbase = [&](int i) -> int { if (i < 0 || i >= L) return 255; const int x = mirror ? L - 1 - i : i; return staged ? (int)S3[x] : (int)S[x]; };
    int r[4];
    auto load_r = [&](int off) {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = sbase(off + k0 + lane * 4 + j);
    };
    // (the one new base of a step -- and the step's graph base -- are asked for BEFORE the step's arithmetic and used behind it: their LDS latency hides under it)
    auto top_base = [&](int new_off) -> int { return sbase(new_off + k0 + 255); };
    auto advance_r = [&](int top) {                                          // r for the next offset; top = top_base(that offset)
        const int from_above = spw::from_upper(r[0], 255);
        r[0] = r[1]; r[1] = r[2]; r[2] = r[3];
        r[3] = lane == 63 ? top : from_above;
    };
    int c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int i = k0 + lane * 4 + j; c[j] = (i >= 0 && i <= L) ? i : K9_INF; }
    int g = 0;
    load_r(0);
    for (int t = 0; t <= ns; ++t) {
        const int si = mirror ? ns - 1 - t : t;                               // the site this pass meets t-th
        const int so = t < ns ? (mirror ? G - eo_at(si) : so_at(si)) : G;
        if (g < so) {
            int xn = gbase(g);
            for (; g < so; ++g) {
                const int x = xn, top = top_base(g + 1);
                xn = g + 1 < G ? gbase(g + 1) : 0;
                k9_step(c, x, r, L, g, k0, lane); advance_r(top);
            }
        }
        if (t == ns) break;
        const int eo = mirror ? G - so_at(si) : eo_at(si);
        if (mirror) {                                                         // the mirrored pass leaves the column in FRONT of every site
            uint16_t* out = entries + ((size_t)J.entry_off + si) * K9_DIAGS;
            reinterpret_cast<uint2*>(out)[lane] = make_uint2((uint32_t)c[0] | ((uint32_t)c[1] << 16), (uint32_t)c[2] | ((uint32_t)c[3] << 16));      // (values <= K9_INF)
        }
        int e0[4], acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { e0[j] = c[j]; acc[j] = K9_INF; }
        const int lr = eo - so;
        const uint32_t a_lo = af_at(si), a_hi = af_at(si + 1);
        for (uint32_t a = a_lo; a < a_hi; ++a) {
            int w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = e0[j];
            const int len = staged ? __builtin_amdgcn_readfirstlane(T_al[a - ab]) : alt_len[a];
            const uint8_t* as = pool + (staged ? 0u : alt_off[a]); lds_cu8* const as3 = A3 + (staged ? __builtin_amdgcn_readfirstlane(T_ao[a - ab]) : 0);
            auto abase = [&](int x) -> int { const int y = mirror ? len - 1 - x : x; return staged ? (int)as3[y] : (int)as[y]; };
            load_r(so);
            if (len > 0) {
                int xn = abase(0);
                for (int x = 0; x < len; ++x) {
                    const int b = xn, top = top_base(so + x + 1);
                    xn = x + 1 < len ? abase(x + 1) : 0;
                    k9_step(w, b, r, L, so + x, k0, lane); advance_r(top);
                }
            }
            // the alternative is len bases where the reference has lr: its diagonals shift by len - lr at the site's end
            const int delta = len - lr;
#pragma unroll
            for (int j = 0; j < 4; ++j) shift[mirror][lane * 4 + j] = w[j];
            spw::wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int src = lane * 4 + j - delta; w[j] = (src >= 0 && src < K9_DIAGS) ? shift[mirror][src] : K9_INF; }
            spw::wave_lds_sync();
            if (!mirror) {
                uint16_t* out = exits + ((size_t)J.exit_off + (a - J.alt_base)) * K9_DIAGS;
                reinterpret_cast<uint2*>(out)[lane] = make_uint2((uint32_t)w[0] | ((uint32_t)w[1] << 16), (uint32_t)w[2] | ((uint32_t)w[3] << 16));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = w[j] < acc[j] ? w[j] : acc[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = acc[j];
        g = eo;
        load_r(g);
    }
    if (!mirror) {
        const int kL = L - G - k0;
        int v = K9_INF;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lane * 4 + j == kL) v = c[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(v, o); v = other < v ? other : v; }
        if (lane == 0) { opt_s = v; J.opt = v; }
    }
    __threadfence();
    __syncthreads();
    // which alternatives lie on an optimal path: cost through it + cost of the rest from the site's end == optimum
    const int opt = opt_s;
    if (opt >= K9_INF) return;
    const int wave = threadIdx.x >> 6;
    for (int si = wave; si < ns; si += 2) {
        const uint16_t* back = entries + ((size_t)J.entry_off + si) * K9_DIAGS;
        const uint32_t a_lo = af_at(si), a_hi = af_at(si + 1);
        for (uint32_t a = a_lo; a < a_hi; ++a) {
            const uint16_t* ex = exits + ((size_t)J.exit_off + (a - J.alt_base)) * K9_DIAGS;
            bool on = false;
            const uint2 e4 = reinterpret_cast<const uint2*>(ex)[lane], b4 = reinterpret_cast<const uint2*>(back)[63 - lane];      // diagonal d of the exit faces K9_DIAGS - 1 - d of the mirrored pass
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = (int)(((j < 2 ? e4.x : e4.y) >> (16 * (j & 1))) & 0xFFFFu), bk = (int)(((j < 2 ? b4.y : b4.x) >> (16 * (1 - (j & 1)))) & 0xFFFFu);
                if (e < K9_INF && bk < K9_INF && e + bk == opt) on = true;
            }
            const bool any = __ballot(on) != 0;
            if (lane == 0) alt_on[a] = any ? 1 : 0;
        }
    }
}

extern "C" int32_t sp_cyp_variant_states(sp_ctx* ctx, const sp_seqset* seqs, const char* backbone, uint32_t backbone_len, uint32_t n_variants,
                                         const int32_t* var_pos, const char* const* var_ref, const char* const* var_alt,
                                         uint8_t* states, sp_aln* alns_out) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!seqs || !backbone || (n_variants && (!var_pos || !var_ref || !var_alt)) || (seqs->n && n_variants && !states)) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_cyp_variant_states: null argument");
    (void)hipSetDevice(ctx->device);
    const uint32_t S = seqs->n;
    if (states) std::memset(states, 3, (size_t)S * n_variants);
    if (S == 0) return SP_OK;
    const bool k9dbg = std::getenv("SP_K9_DEBUG") != nullptr; auto k9t0 = std::chrono::steady_clock::now();
    auto k9mark = [&](const char* what) { if (!k9dbg) return; const auto n = std::chrono::steady_clock::now(); std::fprintf(stderr, "k9 %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - k9t0).count()); k9t0 = n; };
    // place every sequence on the backbone (sequence = indexed / streamed side A, backbone = window side B), with traceback
    sp_seqset bb_pooled; sp_seqset* bbset = &bb_pooled;
    const uint64_t boff[2] = { 0, backbone_len };
    int32_t rc = sp_seqset_make_small(ctx, "k9_backbone", backbone, boff, 1, false, bbset);
    if (rc != SP_OK) return rc;
    std::vector<uint32_t> ai(S), bi(S, 0); for (uint32_t i = 0; i < S; ++i) ai[i] = i;
    std::vector<int32_t> diag(S), votes(S);
    k9mark("backbone set");
    rc = sp_anchor_batch(ctx, seqs, bbset, ai.data(), bi.data(), S, diag.data(), votes.data());
    k9mark("anchor");
    std::vector<sp_pair> pairs; std::vector<uint32_t> who;
    if (rc == SP_OK) for (uint32_t i = 0; i < S; ++i) if (votes[i] >= CYP_MIN_VOTES) { pairs.push_back(sp_pair{ i, 0, diag[i], SP_MAX_ED }); who.push_back(i); }
    std::vector<sp_aln> alns(pairs.size()); std::vector<uint32_t> events(pairs.size() * (size_t)SP_MAX_ED);
    if (rc == SP_OK && !pairs.empty()) rc = sp_align_batch(ctx, seqs, bbset, pairs.data(), pairs.size(), alns.data(), events.data(), SP_MAX_ED);
    if (rc != SP_OK) return rc;
    k9mark("align + traceback");
    if (alns_out) { std::memset(alns_out, 0, sizeof(sp_aln) * S); for (size_t x = 0; x < who.size(); ++x) alns_out[who[x]] = alns[x]; }
    if (n_variants == 0) return SP_OK;
    auto code = [](char c) -> uint8_t { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; };
    std::vector<uint8_t> bcode(backbone_len);
    for (uint32_t i = 0; i < backbone_len; ++i) bcode[i] = code(backbone[i]);
    std::vector<int> rlen(n_variants);
    for (uint32_t v = 0; v < n_variants; ++v) rlen[v] = (int)std::strlen(var_ref[v]);
    // the graphs: per aligned sequence its sites and alternatives (oracle/cyp.c states the same construction)
    const size_t plane_words = (size_t)seqs->h_word_off[seqs->n] + 4;             // the host copy exists: the anchor step indexed the set
    std::vector<uint8_t> pool(bcode);                                               // backbone first, then sequences and alternatives
    std::vector<K9Job> jobs; std::vector<int32_t> site_so, site_eo; std::vector<uint32_t> alt_first; std::vector<int32_t> alt_len; std::vector<uint32_t> alt_off;
    struct SiteVars { int nv; int var[K9_SITE_MAX]; }; std::vector<SiteVars> site_vars; std::vector<int> alt_mask; std::vector<uint32_t> job_seq;
    size_t n_exit = 0, n_entry = 0;
    for (size_t x = 0; x < who.size(); ++x) {
        const sp_aln& al = alns[x];
        if (!al.ok) continue;
        const uint32_t sidx = who[x];
        const uint32_t* w = seqs->h_words.data() + seqs->h_word_off[sidx];
        const uint32_t* np = seqs->has_n ? seqs->h_words.data() + plane_words + seqs->h_word_off[sidx] : nullptr;
        K9Job J; std::memset(&J, 0, sizeof J);
        J.seq_off = (uint32_t)pool.size(); J.L = al.a_end - al.a_start;
        for (int h = al.a_start; h < al.a_end; ++h) { const uint32_t sh = (uint32_t)(h & 15) << 1; pool.push_back((np && ((np[h >> 4] >> sh) & 1u)) ? 4 : (uint8_t)((w[h >> 4] >> sh) & 3u)); }
        const int gs = al.b_start, ge = al.b_end;
        J.bb_off = (uint32_t)gs; J.G = ge - gs;
        int drift = 0, dmin = 0, dmax = 0;
        const uint32_t* ev = events.data() + x * (size_t)SP_MAX_ED;
        for (int e = 0; e < al.nm; ++e) { const uint32_t type = ev[e] >> 30; if (type == SP_EV_I) ++drift; else if (type == SP_EV_D) --drift; dmin = std::min(dmin, drift); dmax = std::max(dmax, drift); }
        J.k0 = (dmin + dmax) / 2 - K9_DIAGS / 2;
        std::vector<int> order;
        for (uint32_t v = 0; v < n_variants; ++v) if (var_pos[v] >= gs && var_pos[v] + rlen[v] <= ge) order.push_back((int)v);
        std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return var_pos[p] < var_pos[q]; });     // (ties keep the variant order)
        J.site_first = (uint32_t)site_so.size(); J.alt_base = (uint32_t)alt_len.size(); J.alt_bytes_off = (uint32_t)pool.size();
        J.exit_off = (uint32_t)n_exit; J.entry_off = (uint32_t)n_entry;
        uint32_t ns = 0;
        for (size_t a = 0; a < order.size();) {
            int s0 = var_pos[order[a]], e0 = s0 + rlen[order[a]]; SiteVars sv; sv.nv = 0;
            while (a < order.size() && var_pos[order[a]] < e0) {
                const int end = var_pos[order[a]] + rlen[order[a]];
                if (sv.nv < K9_SITE_MAX) { e0 = std::max(e0, end); sv.var[sv.nv++] = order[a]; }              // (a ninth overlapping variant stays undecided)
                ++a;
            }
            site_so.push_back(s0 - gs); site_eo.push_back(e0 - gs); site_vars.push_back(sv);
            alt_first.push_back((uint32_t)alt_len.size());
            for (int mask = 0; mask < (1 << sv.nv); ++mask) {
                bool ok = true; int last_end = -1;
                for (int y = 0; y < sv.nv && ok; ++y) if (mask >> y & 1) { const int v = sv.var[y]; if (var_pos[v] < last_end) ok = false; last_end = var_pos[v] + rlen[v]; }
                if (!ok) continue;
                alt_off.push_back((uint32_t)pool.size());
                int b = s0, len = 0;
                for (int y = 0; y < sv.nv; ++y) if (mask >> y & 1) {
                    const int v = sv.var[y];
                    for (; b < var_pos[v]; ++b, ++len) pool.push_back(bcode[b]);
                    for (const char* q = var_alt[v]; *q; ++q, ++len) pool.push_back(code(*q));
                    b = var_pos[v] + rlen[v];
                }
                for (; b < e0; ++b, ++len) pool.push_back(bcode[b]);
                alt_len.push_back(len); alt_mask.push_back(mask);
            }
            ++ns;
        }
        alt_first.push_back((uint32_t)alt_len.size());                              // one closing entry per job (the kernel indexes site_first + job + site)
        J.n_sites = ns; J.n_alts = (uint32_t)alt_len.size() - J.alt_base; J.alt_bytes = (uint32_t)pool.size() - J.alt_bytes_off;
        n_exit += alt_len.size() - J.alt_base; n_entry += ns;
        jobs.push_back(J); job_seq.push_back(sidx);
    }
    if (jobs.empty()) return SP_OK;
    k9mark("graphs on the host");
    uint8_t* d_pool = (uint8_t*)sp_pool(ctx, "k9_pool", pool.size() + 16);
    K9Job* d_jobs = (K9Job*)sp_pool(ctx, "k9_jobs", jobs.size() * sizeof(K9Job));
    int32_t* d_so = (int32_t*)sp_pool(ctx, "k9_so", (site_so.size() + 1) * 4); int32_t* d_eo = (int32_t*)sp_pool(ctx, "k9_eo", (site_eo.size() + 1) * 4);
    uint32_t* d_af = (uint32_t*)sp_pool(ctx, "k9_af", (alt_first.size() + 1) * 4);
    int32_t* d_al = (int32_t*)sp_pool(ctx, "k9_al", (alt_len.size() + 1) * 4); uint32_t* d_ao = (uint32_t*)sp_pool(ctx, "k9_ao", (alt_off.size() + 1) * 4);
    uint16_t* d_exit = (uint16_t*)sp_pool(ctx, "k9_exit", (n_exit + 1) * K9_DIAGS * 2); uint16_t* d_entry = (uint16_t*)sp_pool(ctx, "k9_entry", (n_entry + 1) * K9_DIAGS * 2);
    int32_t* d_on = (int32_t*)sp_pool(ctx, "k9_on", (alt_len.size() + 1) * 4);
    if (!d_pool || !d_jobs || !d_so || !d_eo || !d_af || !d_al || !d_ao || !d_exit || !d_entry || !d_on) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "k9 buffers");
    (void)hipMemcpyAsync(d_pool, pool.data(), pool.size(), hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(K9Job), hipMemcpyHostToDevice, ctx->stream);
    if (!site_so.empty()) { (void)hipMemcpyAsync(d_so, site_so.data(), site_so.size() * 4, hipMemcpyHostToDevice, ctx->stream); (void)hipMemcpyAsync(d_eo, site_eo.data(), site_eo.size() * 4, hipMemcpyHostToDevice, ctx->stream); }
    (void)hipMemcpyAsync(d_af, alt_first.data(), alt_first.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (!alt_len.empty()) { (void)hipMemcpyAsync(d_al, alt_len.data(), alt_len.size() * 4, hipMemcpyHostToDevice, ctx->stream); (void)hipMemcpyAsync(d_ao, alt_off.data(), alt_off.size() * 4, hipMemcpyHostToDevice, ctx->stream); }
    (void)hipMemsetAsync(d_on, 0, (alt_len.size() + 1) * 4, ctx->stream);
    {
        ProfScope ps(ctx, "k9_graph", jobs.size());
        int max_l = 0; for (const K9Job& j : jobs) max_l = std::max(max_l, k9_lds_bytes(j.L, j.G, (int)j.alt_bytes, (int)j.n_sites, (int)j.n_alts));
        const bool fits = max_l <= 60 * 1024;
        const int lds_cap = fits ? max_l : 0;
        if (std::getenv("SP_K9_DEBUG")) for (const K9Job& j : jobs) std::fprintf(stderr, "k9 job L %d G %d sites %u alts %u alt_bytes %u need %d cap %d\n", j.L, j.G, j.n_sites, j.n_alts, j.alt_bytes, k9_lds_bytes(j.L, j.G, (int)j.alt_bytes, (int)j.n_sites, (int)j.n_alts), lds_cap);
        if (fits) {
            (void)hipFuncSetAttribute((const void*)k9_graph_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cap + 16);
            hipLaunchKernelGGL(k9_graph_kernel<true>, dim3((unsigned)jobs.size()), dim3(128), (size_t)lds_cap + 16, ctx->stream, d_pool, d_jobs, d_so, d_eo, d_af, d_al, d_ao, d_exit, d_entry, d_on, lds_cap);
        } else
            hipLaunchKernelGGL(k9_graph_kernel<false>, dim3((unsigned)jobs.size()), dim3(128), 16, ctx->stream, d_pool, d_jobs, d_so, d_eo, d_af, d_al, d_ao, d_exit, d_entry, d_on, 0);
    }
    std::vector<int32_t> on(alt_len.size() + 1);
    (void)hipMemcpyAsync(on.data(), d_on, (alt_len.size() + 1) * 4, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, std::string("k9: ") + hipGetErrorString(e));
    k9mark("uploads + kernel + sync");
    // a variant is 1 / 0 when every optimal alternative of its site carries / lacks it, 2 when they disagree
    for (size_t jx = 0; jx < jobs.size(); ++jx) {
        const K9Job& J = jobs[jx];
        for (uint32_t si = 0; si < J.n_sites; ++si) {
            const SiteVars& sv = site_vars[J.site_first + si];
            bool seen1[K9_SITE_MAX] = {}, seen0[K9_SITE_MAX] = {};
            for (uint32_t a = alt_first[J.site_first + jx + si]; a < alt_first[J.site_first + jx + si + 1]; ++a) {
                if (!on[a]) continue;
                for (int y = 0; y < sv.nv; ++y) { if (alt_mask[a] >> y & 1) seen1[y] = true; else seen0[y] = true; }
            }
            for (int y = 0; y < sv.nv; ++y) states[(size_t)job_seq[jx] * n_variants + sv.var[y]] = (seen1[y] && seen0[y]) ? 2 : seen1[y] ? 1 : seen0[y] ? 0 : 3;
        }
    }
    return SP_OK;
}


// =============================================================================================
// CYP2D6, reads to diplotype: the host-side counterpart of diplotype_cyp2d6 (src/cyp2d6/caller.rs:39-741)
// =============================================================================================
namespace {

struct Label { int type = SP_CYP_UNKNOWN; bool has_sub = false; std::string sub; };
static std::string label_full(const Label& l) { return full_allele(l.type, l.has_sub ? l.sub.c_str() : nullptr); }
static bool label_allowed(const Label& l) { return l.type != SP_CYP_UNKNOWN && l.type != SP_CYP_FALSE_ALLELE; }          // region_label.rs:171-173
// simplify_allele(detailed = true) (region_label.rs:101-128)
static std::string label_reduced(const Label& l, const sp_cyp_problem* pr) {
    if (l.type == SP_CYP_CYP2D6 || l.type == SP_CYP_HYBRID) {
        if (!l.has_sub) return label_full(l);
        for (uint32_t i = 0; i < pr->n_translate; ++i) if (l.sub == pr->translate_key[i]) return std::string("*") + pr->translate_val[i];
        return "*" + l.sub;
    }
    if (l.type == SP_CYP_DELETION) return "*5";
    return label_full(l);
}

// The variants of a typed sequence relative to the star allele it was assigned (assign_haplotype, src/cyp2d6/haplotyper.rs:546-595) in the
// form Cyp2d6Region::deep_label appends them (src/cyp2d6/region.rs:60-91): " +label" unexpected, " -label" missing, " ?label" ambiguous or
// unknown-but-expected; matches and unknown-and-not-expected are not listed.
// VariantAlleleRelationship of every variant of a typed sequence against its assigned allele (haplotyper.rs:551-577) as the code
// sp_inexact_haplotype uses (1 Match, 2 Unexpected, 3 Missing, 4 AmbiguousUnexpected, 5 AmbiguousMissing, 6 UnknownUnexpected,
// 7 UnknownMissing); 255: reference matching reference, which the region does not list (:579-580)
static void region_relationships(const sp_cyp_problem* pr, uint32_t allele, const std::vector<uint8_t>& states, std::vector<uint8_t>& out) {
    out.assign(pr->n_variants, 255);
    if (states.size() != pr->n_variants) return;
    const uint8_t* row = pr->hap_matrix + (size_t)allele * pr->n_variants;
    static const uint8_t expect_ref[4] = { 255, 2, 4, 6 }, expect_alt[4] = { 3, 1, 5, 7 };
    for (uint32_t v = 0; v < pr->n_variants; ++v) out[v] = (row[v] ? expect_alt : expect_ref)[states[v] & 3];
}
struct DeepInfo { std::vector<std::string> suffix; std::vector<std::vector<uint8_t>> rel; std::vector<uint8_t> has; };   // has[i] == 0: Cyp2d6Region::variants is None

static std::string deep_suffix(const sp_cyp_problem* pr, uint32_t allele, const std::vector<uint8_t>& states) {
    std::string out;
    if (states.size() != pr->n_variants || !pr->var_label) return out;
    const uint8_t* row = pr->hap_matrix + (size_t)allele * pr->n_variants;
    for (uint32_t v = 0; v < pr->n_variants; ++v) {
        const int hv = row[v], sv = states[v];
        char sign = 0;
        if (hv == 0) sign = sv == 1 ? '+' : sv == 2 ? '?' : 0;               // 0: match, 3: UnknownUnexpected (not listed)
        else sign = sv == 0 ? '-' : sv == 1 ? 0 : '?';                          // 2: AmbiguousMissing, 3: UnknownMissing
        if (sign) { out += ' '; out += sign; out += pr->var_label[v]; }
    }
    return out;
}

// find_full_type_in_sequence + assign_haplotype for a batch of sequences (src/cyp2d6/haplotyper.rs:326-602).  What the device computes
// for a sequence (its best template, the allele scores over its variant states) does not depend on `force`; the caller types the group
// consensuses and, after merging, the final ones -- mostly the same strings -- so the results are kept per sequence.
struct Typed { int best_template = -1; uint32_t bvi = 0, ball = 0; std::vector<uint8_t> tie, states; };
using TypeCache = std::map<std::string, Typed>;
// the device part: every sequence the cache does not hold yet is placed on the templates and on the backbone, its variant states and its best alleles
// are worked out -- one batch, however many samples the sequences come from (a cohort's group of samples types its consensuses together)
static int32_t type_fresh(sp_ctx* ctx, const sp_cyp_problem* pr, const std::vector<std::string>& seqs, double max_missing, TypeCache& cache) {
    std::vector<const std::string*> fresh;                             // sequences the device has not seen yet (an empty one has no matches: Unknown)
    std::string blob; std::vector<uint64_t> off(1, 0);
    for (const std::string& q : seqs) if (!q.empty() && !cache.count(q)) { cache[q]; fresh.push_back(&q); blob += q; off.push_back(blob.size()); }
    if (!fresh.empty()) {
        sp_seqset pooled; sp_seqset* set = &pooled;                       // pooled buffers: no allocation, no free (a handful of consensuses)
        int32_t rc = sp_seqset_make_small(ctx, "cyp_typed", blob.data(), off.data(), (uint32_t)fresh.size(), true, set);
        if (rc != SP_OK) return rc;
        std::vector<sp_region_hit> hits(fresh.size() * 16 + 16); uint64_t nh = 0;
        rc = cyp_find_regions(ctx, pr->templates, pr->template_type, set, max_missing, hits.data(), hits.size(), &nh, ctx->mm2_rescore != 0);
        if (rc == SP_OK && nh > hits.size()) { hits.resize(nh); rc = cyp_find_regions(ctx, pr->templates, pr->template_type, set, max_missing, hits.data(), hits.size(), &nh, ctx->mm2_rescore != 0); }
        if (rc != SP_OK) return rc;
        // the best match of a sequence: lowest penalised score, first on ties (:344-349)
        std::vector<uint32_t> deep_of;                                   // the sequences whose best match is a CYP2D6 template: only they are looked at variant by variant
        std::string dblob; std::vector<uint64_t> doff(1, 0);             // (assign_haplotype is only called for them, haplotyper.rs:371; a CYP2D7 consensus is 4 % away from the
        for (uint32_t x = 0; x < fresh.size(); ++x) {                    //  backbone: its placement alone was a third of this step)
            Typed& t = cache[*fresh[x]];
            double bs = 0;
            for (uint64_t h = 0; h < nh; ++h) if (hits[h].read == (int32_t)x) {
                const double sc = cyp_score(hits[h].seq_len, hits[h].nm, hits[h].unmapped, true);
                if (t.best_template < 0 || sc < bs) { t.best_template = hits[h].template_idx; bs = sc; }
            }
            t.bvi = 0; t.ball = 0; t.tie.assign(pr->n_alleles, 0); t.states.assign(pr->n_variants, 3);
            if (t.best_template >= 0 && pr->template_deep[t.best_template]) { deep_of.push_back(x); dblob += *fresh[x]; doff.push_back(dblob.size()); }
        }
        if (!deep_of.empty()) {
            const uint32_t D = (uint32_t)deep_of.size();
            sp_seqset dpooled; sp_seqset* dset = &dpooled;
            const bool all = D == fresh.size();
            if (!all) { rc = sp_seqset_make_small(ctx, "cyp_typed_deep", dblob.data(), doff.data(), D, true, dset); if (rc != SP_OK) return rc; }
            std::vector<uint8_t> states((size_t)D * pr->n_variants, 3);
            rc = sp_cyp_variant_states(ctx, all ? set : dset, pr->backbone, pr->backbone_len, pr->n_variants, pr->var_pos, pr->var_ref, pr->var_alt, states.data(), nullptr);
            std::vector<uint32_t> bvi(D), ball(D); std::vector<uint8_t> tie((size_t)D * std::max<uint32_t>(pr->n_alleles, 1));
            if (rc == SP_OK && pr->n_alleles && pr->n_variants)
                rc = sp_cyp_score_alleles(ctx, pr->n_variants, pr->n_alleles, pr->hap_matrix, pr->var_is_vi, D, states.data(), bvi.data(), ball.data(), tie.data());
            if (rc != SP_OK) return rc;
            for (uint32_t y = 0; y < D; ++y) {
                Typed& t = cache[*fresh[deep_of[y]]];
                t.bvi = bvi[y]; t.ball = ball[y];
                t.tie.assign(tie.begin() + (size_t)y * pr->n_alleles, tie.begin() + (size_t)(y + 1) * pr->n_alleles);
                t.states.assign(states.begin() + (size_t)y * pr->n_variants, states.begin() + (size_t)(y + 1) * pr->n_variants);
            }
        }
    }
    return SP_OK;
}

static int32_t type_sequences(sp_ctx* ctx, const sp_cyp_problem* pr, const std::vector<std::string>& seqs, double max_missing, bool force, std::vector<Label>& out,
                              TypeCache& cache, DeepInfo* deep = nullptr) {
    out.assign(seqs.size(), Label());
    if (deep) { deep->suffix.assign(seqs.size(), std::string()); deep->rel.assign(seqs.size(), std::vector<uint8_t>()); deep->has.assign(seqs.size(), 0); }
    const int32_t rc0 = type_fresh(ctx, pr, seqs, max_missing, cache);
    if (rc0 != SP_OK) return rc0;
    for (size_t i = 0; i < seqs.size(); ++i) {
        if (seqs[i].empty()) continue;
        const Typed& ty = cache[seqs[i]];
        Label& lab = out[i];
        if (ty.best_template < 0) continue;                            // "no matches found" => Unknown (caller.rs:350-355)
        const int t = ty.best_template;
        if (!pr->template_deep[t]) { lab.type = pr->template_type[t]; lab.has_sub = pr->template_subtype && pr->template_subtype[t]; if (lab.has_sub) lab.sub = pr->template_subtype[t]; continue; }
        std::vector<Label> cands;
        for (uint32_t a = 0; a < pr->n_alleles; ++a) if (ty.tie[a]) { Label c; c.type = SP_CYP_CYP2D6; c.has_sub = true; c.sub = pr->allele_subtype[a]; cands.push_back(c); }
        if (ty.bvi == 0 && ty.ball == 0) cands.push_back(Label());     // the Unknown label the search starts from (haplotyper.rs:471-473)
        if (cands.empty()) continue;
        if (cands.size() > 1) {
            if (!force) continue;                                       // ambiguous => Unknown (:543-546)
            std::stable_sort(cands.begin(), cands.end(), [](const Label& p, const Label& q) { return label_full(p) < label_full(q); });
        }
        lab = cands[0];
        // the variants that set the sequence apart from the allele it got (Cyp2d6Region::variants; None for Unknown)
        if (deep && lab.type == SP_CYP_CYP2D6 && lab.has_sub)
            for (uint32_t a = 0; a < pr->n_alleles; ++a) if (lab.sub == pr->allele_subtype[a]) {
                deep->suffix[i] = deep_suffix(pr, a, ty.states); region_relationships(pr, a, ty.states, deep->rel[i]); deep->has[i] = 1; break;
            }
    }
    return SP_OK;
}

} // namespace

extern "C" int32_t sp_cyp_diplotype(sp_ctx* ctx, const sp_cyp_problem* pr, const sp_seqset* reads, sp_cyp_call* call, char* consensus, uint32_t cons_cap) {
    return sp_cyp_diplotype_detailed(ctx, pr, reads, call, consensus, cons_cap, nullptr);
}

namespace {

// What a CYP2D6 call holds between its three parts: (a) regions of interest and the consensus inputs, (b) the multi-way consensus -- alone
// (sp_consensus_priority) or in lockstep with the other samples of a cohort (sp_consensus_priority_many) --, (c) merge, typing, weights, chains.
struct CypMid {
    uint32_t R = 0;
    std::vector<sp_region_hit> hits;
    std::vector<uint32_t> c_idx, c_hit; std::vector<int32_t> c_start, c_len, boff, hoff, seeds;
    sp_seqset raw, hpc;                                   // the segments (pooled buffers named after the call's segment prefix)
    sp_cons_config cc{};
    const sp_seqset* levels[2] = { nullptr, nullptr }; const int32_t* offs[2] = { nullptr, nullptr };
    sp_priority_problem pp{};
    uint32_t n_in = 0, cap = 0, n_groups = 0;
    std::vector<int32_t> group_of; std::vector<char> text;
    bool finished = false;                                // the call is complete after part (a) already (no reads)
    // between (c1), the weights and (c2)
    std::vector<std::string> full_cons, final_cons; std::vector<int32_t> final_group; std::vector<Label> labels; DeepInfo deep; std::vector<uint8_t> allowed;
    std::vector<uint32_t> a_idx, seg_off; std::vector<int32_t> a_start, a_len; uint32_t H = 0;
};

int32_t cyp_part_a(sp_ctx* ctx, const sp_cyp_problem* pr, const sp_seqset* reads, sp_cyp_call* call, sp_cyp_region_variants* region_variants, const char* seg_prefix, CypMid& m,
                   const std::vector<sp_region_hit>* found_hits = nullptr) {
    if (region_variants) std::memset(region_variants->has_variants, 0, sizeof region_variants->has_variants);
    if (!pr || !reads || !call || !pr->templates || !pr->template_type || !pr->template_deep || !pr->backbone) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_cyp_diplotype: null argument");
    (void)hipSetDevice(ctx->device);
    std::memset(call, 0, sizeof *call);
    const uint32_t R = m.R = reads->n;
    auto& hits = m.hits; auto& c_idx = m.c_idx; auto& c_hit = m.c_hit; auto& c_start = m.c_start; auto& c_len = m.c_len; auto& boff = m.boff; auto& hoff = m.hoff; auto& seeds = m.seeds;
    sp_seqset& raw = m.raw; sp_seqset& hpc = m.hpc;
    HostMarks hm(ctx);
    // 1. regions of interest (caller.rs:126-139): max_missing_chain_frac = 0.5
    int32_t rc = SP_OK;
    if (found_hits) hits = *found_hits;                                   // (a cohort's group of samples searched its regions in one call: cyp_group_regions)
    else {
        hits.assign((size_t)R * 8 + 16, sp_region_hit{}); uint64_t nh = 0;
        rc = cyp_find_regions(ctx, pr->templates, pr->template_type, reads, 0.5, hits.data(), hits.size(), &nh, ctx->mm2_rescore != 0);
        if (rc == SP_OK && nh > hits.size()) { hits.resize(nh); rc = cyp_find_regions(ctx, pr->templates, pr->template_type, reads, 0.5, hits.data(), hits.size(), &nh, ctx->mm2_rescore != 0); }
        if (rc != SP_OK) return rc;
        hits.resize(nh);
    }
    hm.mark("host:cyp_regions");
    // 2. consensus inputs (caller.rs:176-245): max_missing_consensus_frac = 0.5, offset window +-50
    std::vector<std::string> guides(pr->templates->n);
    std::vector<std::vector<int32_t>> run_of(pr->templates->n);          // sp_hpc_pos of every position of a template, built once (a scan per hit was 1 ms of a 2,000-read sample)
    for (uint32_t h = 0; h < hits.size(); ++h) {
        const sp_region_hit& q = hits[h];
        if (cyp_score(q.seq_len, q.nm, q.unmapped, true) > 0.5) continue;
        c_idx.push_back((uint32_t)q.read); c_start.push_back(q.start); c_len.push_back(q.end - q.start); c_hit.push_back(h);
        boff.push_back(q.clip_start == 0 ? -1 : q.clip_start + 50);
        std::string& guide = guides[(size_t)q.template_idx];                                             // (decoded once per template, not once per hit)
        std::vector<int32_t>& runs = run_of[(size_t)q.template_idx];
        if (guide.empty()) {
            guide = sp_seqset_decode(ctx, pr->templates, (uint32_t)q.template_idx);
            runs.resize(guide.size() + 1);
            int32_t r = -1;
            for (size_t x = 0; x < guide.size(); ++x) { if (x == 0 || guide[x] != guide[x - 1]) ++r; runs[x] = r; }
            runs[guide.size()] = r + 1;                                                                  // (a position behind the end: the number of runs)
        }
        const int32_t hp = q.clip_start < 0 ? 0 : runs[std::min<size_t>((size_t)q.clip_start, guide.size())];   // sp_hpc_pos: hpc_with_guide (homopolymers.rs:53-64)
        hoff.push_back(hp == 0 ? -1 : hp + 50);
        int seed = -1;
        switch (pr->template_type[q.template_idx]) { case SP_CYP_DELETION: seed = 0; break; case SP_CYP_REP6: seed = 1; break; case SP_CYP_REP7: seed = 2; break;
                                                     case SP_CYP_SPACER: seed = 3; break; case SP_CYP_LINK_REGION: seed = 4; break; default: break; }
        seeds.push_back(seed);
    }
    if (c_idx.empty()) { call->status = 1; m.finished = true; return SP_OK; }                               // NO_READS (caller.rs:254-266)
    rc = sp_make_segments(ctx, reads, c_idx, c_start, c_len, seg_prefix, &raw, &hpc);
    if (rc != SP_OK) return rc;
    hm.mark("host:cyp_segments");
    // 3. multi-way consensus, homopolymer-compressed level first (caller.rs:162-270): the problem
    sp_cons_config& cc = m.cc;
    cc.min_count = pr->min_consensus_count; cc.min_af = pr->min_consensus_fraction; cc.dual_max_ed_delta = pr->dual_max_ed_delta;
    cc.no_retry_ladder = ctx->cons_retry_ladder ? 0 : 1;
    cc.allow_early_termination = 1; cc.allow_dual = 1; cc.offset_window = 100; cc.offset_compare_length = 100;      // caller.rs:144-159: compare 100 bases, window 2 x 50
    // (the reference's own numbers: the placement rule -- the read is placed when the consensus reaches offset + compare length, the end of the comparison on the consensus is
    //  free -- makes a 100-base pattern in a window of 100 work as written: sp_consensus.hip activate_late, oracle/consensus.c find_start)
    m.n_in = raw.n;
    m.levels[0] = &hpc; m.levels[1] = &raw; m.offs[0] = hoff.data(); m.offs[1] = boff.data();
    m.pp.n_levels = 2; m.pp.n = m.n_in; m.pp.levels = m.levels; m.pp.offsets = m.offs; m.pp.seeds = seeds.data(); m.pp.cfg = cc;
    m.cap = (uint32_t)raw.max_len + 1024;
    m.group_of.assign(m.n_in, 0); m.n_groups = 0;
    m.text.assign((size_t)SP_CYP_MAXCONS * 2 * m.cap, 0);
    return SP_OK;
}

// The read sets of several samples seen as ONE set -- no bases move: a set addresses its packed words as base + 64-bit word offset, so the offsets of all the
// sets are rewritten relative to the lowest of their bases.  first[i] = number of the first read of sample i in the view.  false when the sets cannot be
// seen as one (a set with an N plane, more reads than one search should take).
#define CYP_GROUP_READS 16384
bool cyp_group_view(sp_ctx* ctx, const sp_seqset* const* reads, uint32_t n, const char* prefix, sp_seqset& all, std::vector<uint32_t>& first, int32_t& rc) {
    rc = SP_OK;
    uint64_t total = 0; const uint32_t* base = nullptr;
    for (uint32_t i = 0; i < n; ++i) {
        if (reads[i]->has_n || reads[i]->d_nplane) return false;
        total += reads[i]->n;
        if (reads[i]->n && (!base || (uintptr_t)reads[i]->d_words < (uintptr_t)base)) base = reads[i]->d_words;
    }
    if (n < 2 || total == 0 || total > CYP_GROUP_READS) return false;
    all = sp_seqset(); all.ctx = ctx; all.n = (uint32_t)total; all.has_n = false;
    all.h_len.reserve(total); all.h_word_off.reserve(total + 1);
    first.assign(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t shift = reads[i]->n ? (uint64_t)(((uintptr_t)reads[i]->d_words - (uintptr_t)base) / sizeof(uint32_t)) : 0;
        for (uint32_t r = 0; r < reads[i]->n; ++r) { all.h_len.push_back(reads[i]->h_len[r]); all.h_word_off.push_back(shift + reads[i]->h_word_off[r]); }
        all.max_len = std::max(all.max_len, reads[i]->max_len);
        first[i + 1] = first[i] + reads[i]->n;
    }
    all.h_word_off.push_back(all.h_word_off.back());
    all.d_words = const_cast<uint32_t*>(base);
    const std::string px(prefix);
    all.d_word_off = (uint64_t*)sp_pool(ctx, (px + "_woff").c_str(), (total + 1) * 8); all.d_len = (int32_t*)sp_pool(ctx, (px + "_len").c_str(), total * 4);
    if (!all.d_word_off || !all.d_len) { rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "cyp group read set"); return true; }
    (void)hipMemcpyAsync(all.d_word_off, all.h_word_off.data(), (total + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(all.d_len, all.h_len.data(), total * 4, hipMemcpyHostToDevice, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    return true;
}

// The regions of interest of several samples in ONE search (the result for a read does not depend on the other reads); the hits go back to their samples with
// the read numbered inside its sample.  false: the caller searches sample by sample.
bool cyp_group_regions(sp_ctx* ctx, const sp_cyp_problem* pr, const sp_seqset& all, const std::vector<uint32_t>& first, uint32_t n, std::vector<std::vector<sp_region_hit>>& per_sample, int32_t& rc) {
    HostScope hs(ctx, "host:cyp_regions");
    std::vector<sp_region_hit> hits((size_t)all.n * 8 + 16); uint64_t nh = 0;
    rc = cyp_find_regions(ctx, pr->templates, pr->template_type, &all, 0.5, hits.data(), hits.size(), &nh, ctx->mm2_rescore != 0);
    if (rc == SP_OK && nh > hits.size()) { hits.resize(nh); rc = cyp_find_regions(ctx, pr->templates, pr->template_type, &all, 0.5, hits.data(), hits.size(), &nh, ctx->mm2_rescore != 0); }
    if (rc != SP_OK) return true;
    per_sample.assign(n, std::vector<sp_region_hit>());
    uint32_t i = 0;
    for (uint64_t h = 0; h < nh; ++h) {                                    // (hits come read by read)
        while ((uint32_t)hits[h].read >= first[i + 1]) ++i;
        sp_region_hit q = hits[h]; q.read -= (int32_t)first[i];
        per_sample[i].push_back(q);
    }
    return true;
}

// Part (c) in three steps, so that a cohort's group of samples can take the middle one together: (c1) merge + typing of the final consensus regions, the weights
// (cyp_weights_one for a sample alone, cyp_weights_group for a group), (c2) chains + best chain pair + the call.
struct Ahead { std::thread th; bool started = false; int32_t rc = -1; sp_seqset all, cons; std::vector<sp_aln> alns; };
struct JoinAhead { Ahead& a; ~JoinAhead() { if (a.started && a.th.joinable()) a.th.join(); } };

int32_t cyp_part_c1(sp_ctx* ctx, const sp_cyp_problem* pr, const sp_seqset* reads, sp_cyp_region_variants* region_variants, CypMid& m, TypeCache* shared_types, Ahead* ahead_p) {
    const uint32_t R = m.R, n_in = m.n_in, cap = m.cap, n_groups = m.n_groups;
    auto& hits = m.hits; auto& boff = m.boff; auto& group_of = m.group_of; auto& text = m.text;
    sp_seqset& raw = m.raw; const sp_cons_config cc = m.cc;
    int32_t rc = SP_OK;
    HostMarks hm(ctx);
    std::vector<std::string> hpc_cons(n_groups); std::vector<std::string>& full_cons = m.full_cons; full_cons.assign(n_groups, std::string());
    for (uint32_t g = 0; g < n_groups; ++g) { hpc_cons[g] = text.data() + (size_t)(2 * g) * cap; full_cons[g] = text.data() + (size_t)(2 * g + 1) * cap; }
    // (step 6's device half ahead of its turn: the placements of every region of interest on every group consensus do not depend on the typing, and unless step 4
    //  merges groups the group consensuses ARE the final ones -- a helper stream places them while this one types, a 2,000-read sample's 3 ms under its 7 ms)
    std::vector<uint32_t>& a_idx = m.a_idx; std::vector<int32_t>& a_start = m.a_start; std::vector<int32_t>& a_len = m.a_len; std::vector<uint32_t>& seg_off = m.seg_off;
    a_idx.assign(hits.size(), 0); a_start.assign(hits.size(), 0); a_len.assign(hits.size(), 0); seg_off.assign(R + 1, 0);
    for (size_t h = 0; h < hits.size(); ++h) { a_idx[h] = (uint32_t)hits[h].read; a_start[h] = hits[h].start; a_len[h] = hits[h].end - hits[h].start; seg_off[hits[h].read + 1] += 1; }
    for (uint32_t r = 0; r < R; ++r) seg_off[r + 1] += seg_off[r];
    sp_ctx* beside = (ahead_p && ctx->split_genes && !shared_types && n_groups && hits.size() * n_groups >= 2048) ? sp_ctx_helper(ctx, 0) : nullptr;     // (a cohort's streams are busy already; so is a GPU with several samples in flight: "hla_split_genes" 0)
    if (beside) {
        Ahead& ahead = *ahead_p;
        try {
            ahead.th = std::thread([&m, &ahead, beside, reads, n_groups]() {      // (outlives this function: nothing of its frame is referred to)
                try {
                    (void)hipSetDevice(beside->device);
                    int32_t r2 = sp_make_segments(beside, reads, m.a_idx, m.a_start, m.a_len, "cypa", &ahead.all, nullptr);
                    std::string blob; std::vector<uint64_t> off(1, 0);
                    for (const std::string& c : m.full_cons) { blob += c; off.push_back(blob.size()); }
                    if (r2 == SP_OK) r2 = sp_seqset_make_small(beside, "cyp_final", blob.data(), off.data(), n_groups, true, &ahead.cons);
                    if (r2 == SP_OK) r2 = cyp_align_all(beside, &ahead.cons, &ahead.all, 1, 0.0, 1, "k4_weight_cells", ahead.alns);
                    ahead.rc = r2;
                } catch (...) { ahead.rc = SP_ERR_OUT_OF_MEMORY; }
            });
            ahead.started = true;
        } catch (const std::system_error&) { }
    }
    // 4. merge_consensus_results (caller.rs:750-898): max_missing_typing_frac = 0.1, no forced assignment
    std::vector<Label> glabel;
    TypeCache own_types;
    TypeCache& typed = shared_types ? *shared_types : own_types;          // (a cohort's group of samples shares what it has typed)
    rc = type_sequences(ctx, pr, full_cons, 0.1, false, glabel, typed);
    if (rc != SP_OK) return rc;
    std::map<std::pair<std::string, std::string>, std::vector<uint32_t>> cset;
    std::map<std::string, std::vector<uint32_t>> uset;
    for (uint32_t g = 0; g < n_groups; ++g) {
        if (!label_allowed(glabel[g])) uset[hpc_cons[g]].push_back(g);
        else cset[{ hpc_cons[g], label_reduced(glabel[g], pr) }].push_back(g);
    }
    std::vector<std::pair<std::string, std::string>> ignored;
    for (auto& u : uset) {
        std::vector<std::pair<std::string, std::string>> others;
        for (auto& kv : cset) if (kv.first.first == u.first) others.push_back(kv.first);
        if (others.size() == 1) { auto& dst = cset[others[0]]; dst.insert(dst.end(), u.second.begin(), u.second.end()); }
        else { if (others.size() > 1) ignored.push_back({ u.first, "UNKNOWN" }); cset[{ u.first, "UNKNOWN" }] = u.second; }
    }
    if (cset.size() > SP_CYP_MAXCONS) return sp_fail(ctx, SP_ERR_CAPACITY, "sp_cyp_diplotype: more than SP_CYP_MAXCONS consensus regions");
    std::vector<std::string>& final_cons = m.final_cons; final_cons.clear(); std::vector<int32_t> seq_idx(n_in, -1);
    std::vector<int32_t>& final_group = m.final_group; final_group.clear();                                     // the group a final consensus is the consensus of (-1: merged from several, or ignored)
    {
        std::vector<sp_cons_problem> P; std::vector<sp_cons_output> O; std::vector<size_t> slot;
        std::vector<std::vector<uint32_t>> members; std::vector<std::vector<int32_t>> moffs, ms1, ms2; std::vector<std::vector<uint8_t>> mis; std::vector<std::vector<char>> mtext;
        sp_cons_config single = cc; single.allow_dual = 0;
        for (auto& kv : cset) {
            const size_t ci = final_cons.size();
            std::vector<uint32_t> mem;
            for (uint32_t s2 = 0; s2 < n_in; ++s2) if (std::find(kv.second.begin(), kv.second.end(), (uint32_t)group_of[s2]) != kv.second.end()) { mem.push_back(s2); seq_idx[s2] = (int32_t)ci; }
            if (std::find(ignored.begin(), ignored.end(), kv.first) != ignored.end()) { final_cons.push_back(std::string()); final_group.push_back(-1); continue; }
            if (kv.second.size() == 1) { final_cons.push_back(full_cons[kv.second[0]]); final_group.push_back((int32_t)kv.second[0]); continue; }
            final_cons.push_back(std::string()); final_group.push_back(-1);                                  // merge: one consensus over all their reads (:852-871)
            int32_t mn = INT32_MAX; for (uint32_t s2 : mem) mn = std::min(mn, boff[s2] < 0 ? 0 : boff[s2]);
            std::vector<int32_t> mo; for (uint32_t s2 : mem) { const int32_t v = boff[s2] < 0 ? 0 : boff[s2]; mo.push_back(v == mn ? -1 : v - mn + (mn == 0 ? 0 : 50)); }
            members.push_back(mem); moffs.push_back(mo); slot.push_back(ci);
        }
        P.resize(members.size()); O.resize(members.size()); ms1.resize(members.size()); ms2.resize(members.size()); mis.resize(members.size()); mtext.resize(members.size());
        for (size_t x = 0; x < members.size(); ++x) {
            ms1[x].resize(members[x].size()); ms2[x].resize(members[x].size()); mis[x].resize(members[x].size()); mtext[x].assign((size_t)2 * cap, 0);
            P[x].reads = &raw; P[x].read_idx = members[x].data(); P[x].n = (uint32_t)members[x].size(); P[x].offsets = moffs[x].data(); P[x].cfg = single;
            std::memset(&O[x], 0, sizeof O[x]);
            O[x].cons1 = mtext[x].data(); O[x].cons2 = mtext[x].data() + cap; O[x].cap = cap; O[x].is_cons1 = mis[x].data(); O[x].score1 = ms1[x].data(); O[x].score2 = ms2[x].data();
        }
        if (!P.empty()) { rc = sp_consensus_batch(ctx, (uint32_t)P.size(), P.data(), O.data()); if (rc != SP_OK) return rc; }
        for (size_t x = 0; x < members.size(); ++x) final_cons[slot[x]] = mtext[x].data();
    }
    hm.mark("host:cyp_merge");
    // 5. typing of the final consensus regions, forced assignment, duplicates become FalseAllele (caller.rs:331-375)
    std::vector<Label>& labels = m.labels;
    DeepInfo& deep = m.deep;
    rc = type_sequences(ctx, pr, final_cons, 0.1, true, labels, typed, &deep);
    if (rc != SP_OK) return rc;
    for (size_t i = 0; i < final_cons.size(); ++i)
        for (size_t j = 0; j < i; ++j) if (final_cons[j] == final_cons[i]) { labels[i].type = SP_CYP_FALSE_ALLELE; break; }
    const uint32_t H = (uint32_t)final_cons.size();
    if (region_variants)
        for (uint32_t h = 0; h < H && h < SP_CYP_MAXCONS; ++h) {
            region_variants->has_variants[h] = deep.has[h];
            if (deep.has[h] && region_variants->state && pr->n_variants) std::memcpy(region_variants->state + (size_t)h * pr->n_variants, deep.rel[h].data(), pr->n_variants);
        }
    hm.mark("host:cyp_typing");
    m.H = H;
    m.allowed.assign(H, 0); for (uint32_t h = 0; h < H; ++h) m.allowed[h] = label_allowed(labels[h]) && !final_cons[h].empty();
    return SP_OK;
}

// 6. weights of every region of interest (caller.rs:429-640), one sample: from the placements a helper stream made ahead, or here
int32_t cyp_weights_one(sp_ctx* ctx, const sp_seqset* reads, CypMid& m, Ahead& ahead, std::vector<uint64_t>& ed, std::vector<double>& ov, std::vector<uint8_t>& kept, sp_seqset& all_here, const sp_seqset*& all_p) {
    const uint32_t H = m.H, n_groups = m.n_groups;
    const std::vector<uint8_t>& allowed = m.allowed; const std::vector<int32_t>& final_group = m.final_group; const std::vector<std::string>& final_cons = m.final_cons;
    const std::vector<uint32_t>& a_idx = m.a_idx; const std::vector<int32_t>& a_start = m.a_start; const std::vector<int32_t>& a_len = m.a_len;
    int32_t rc = SP_OK;
    HostMarks hm(ctx);
    if (ahead.started) ahead.th.join();
    bool from_ahead = ahead.started && ahead.rc == SP_OK;                 // every final consensus that counts is a group's own: its placements are there already
    for (uint32_t h = 0; h < H && from_ahead; ++h) if (allowed[h] && final_group[h] < 0) from_ahead = false;
    sp_seqset& all = from_ahead ? ahead.all : all_here;
    all_p = &all;
    if (from_ahead) {
        ed.resize((size_t)all.n * H); ov.resize((size_t)all.n * H); kept.resize(all.n);
        std::vector<sp_aln> alns((size_t)all.n * H, sp_aln{}); std::vector<int32_t> cons_len(H);
        for (uint32_t h = 0; h < H; ++h) cons_len[h] = (int32_t)final_cons[h].size();
        for (uint32_t sx = 0; sx < all.n; ++sx) for (uint32_t h = 0; h < H; ++h) if (final_group[h] >= 0) alns[(size_t)sx * H + h] = ahead.alns[(size_t)sx * n_groups + final_group[h]];
        std::vector<sp_affine_aln> af; std::vector<K4Pick> picks;
        if (ctx->mm2_rescore) {
            cyp_pick_near_min(H, allowed.data(), all.n, all.h_len.data(), alns, af, 0, final_group.data(), 0, picks);      // (consensus h is sequence final_group[h] of the set placed ahead)
            rc = cyp_rescore_picks(ctx, &ahead.cons, &all, picks);
            if (rc != SP_OK) return rc;
        }
        cyp_weights_from_alns(H, cons_len.data(), allowed.data(), all.n, all.h_len.data(), alns, ed.data(), ov.data(), kept.data(), af.empty() ? nullptr : af.data());
    } else {
        rc = sp_make_segments(ctx, reads, a_idx, a_start, a_len, "cypa", &all, nullptr);
        if (rc != SP_OK) return rc;
        std::string blob; std::vector<uint64_t> off(1, 0);
        for (const std::string& c : final_cons) { blob += c; off.push_back(blob.size()); }
        sp_seqset cons_pooled; sp_seqset* cons_set = &cons_pooled;
        rc = sp_seqset_make_small(ctx, "cyp_final", blob.data(), off.data(), H, true, cons_set);
        if (rc != SP_OK) return rc;
        ed.resize((size_t)all.n * H); ov.resize((size_t)all.n * H); kept.resize(all.n);
        rc = sp_cyp_weight_segments(ctx, cons_set, allowed.data(), &all, ed.data(), ov.data(), kept.data());
        if (rc != SP_OK) return rc;
    }
    hm.mark("host:cyp_weights");
    return SP_OK;
}

// The same weights for a cohort's group of samples in one go: the regions of interest of all its samples are cut out of the group view of their reads as one set,
// their final consensuses are one indexed set, and one list of pairs (every region of a sample x every consensus of that sample that counts) goes through the
// placement kernels; a placement does not depend on what else is in the sets.  ms[k] == nullptr: sample k is not in play.
int32_t cyp_weights_group(sp_ctx* ctx, const sp_seqset& view, const std::vector<uint32_t>& first, const std::vector<CypMid*>& ms,
                          std::vector<std::vector<uint64_t>>& ed, std::vector<std::vector<double>>& ov, std::vector<std::vector<uint8_t>>& kept) {
    HostScope hs(ctx, "host:cyp_weights");
    const uint32_t n = (uint32_t)ms.size();
    std::vector<uint32_t> a_idx; std::vector<int32_t> a_start, a_len;
    std::vector<uint32_t> seg_first(n + 1, 0), cons_first(n + 1, 0);
    std::string blob; std::vector<uint64_t> off(1, 0);
    for (uint32_t k = 0; k < n; ++k) {
        seg_first[k + 1] = seg_first[k]; cons_first[k + 1] = cons_first[k];
        if (!ms[k]) continue;
        const CypMid& m = *ms[k];
        for (size_t x = 0; x < m.a_idx.size(); ++x) { a_idx.push_back(first[k] + m.a_idx[x]); a_start.push_back(m.a_start[x]); a_len.push_back(m.a_len[x]); }
        for (const std::string& c : m.final_cons) { blob += c; off.push_back(blob.size()); }
        seg_first[k + 1] += (uint32_t)m.a_idx.size(); cons_first[k + 1] += m.H;
    }
    ed.assign(n, {}); ov.assign(n, {}); kept.assign(n, {});
    if (a_idx.empty()) return SP_OK;
    sp_seqset all, cons;
    int32_t rc = sp_make_segments(ctx, &view, a_idx, a_start, a_len, "cypga", &all, nullptr);
    if (rc == SP_OK) rc = sp_seqset_make_small(ctx, "cypg_final", blob.data(), off.data(), cons_first[n], true, &cons);
    if (rc != SP_OK) return rc;
    std::vector<uint32_t> ai, bi;
    for (uint32_t k = 0; k < n; ++k) if (ms[k])
        for (uint32_t sx = seg_first[k]; sx < seg_first[k + 1]; ++sx)
            for (uint32_t h = 0; h < ms[k]->H; ++h) if (ms[k]->allowed[h]) { ai.push_back(cons_first[k] + h); bi.push_back(sx); }
    std::vector<sp_aln> alns;
    rc = cyp_align_pairs(ctx, &cons, &all, ai, bi, 1, 1, 0.0, 1, "k4_weight_cells", alns);
    if (rc != SP_OK) return rc;
    size_t p = 0;
    std::vector<std::vector<sp_aln>> mine(n); std::vector<std::vector<sp_affine_aln>> af(n); std::vector<K4Pick> picks;
    for (uint32_t k = 0; k < n; ++k) if (ms[k]) {
        const CypMid& m = *ms[k];
        const uint32_t H = m.H, ns = seg_first[k + 1] - seg_first[k];
        mine[k].assign((size_t)ns * H, sp_aln{});
        for (uint32_t sx = 0; sx < ns; ++sx) for (uint32_t h = 0; h < H; ++h) if (m.allowed[h]) mine[k][(size_t)sx * H + h] = alns[p++];
        if (ctx->mm2_rescore) cyp_pick_near_min(H, m.allowed.data(), ns, all.h_len.data() + seg_first[k], mine[k], af[k], cons_first[k], nullptr, seg_first[k], picks);
    }
    rc = cyp_rescore_picks(ctx, &cons, &all, picks);                        // (the whole group's placements near a minimum in one go)
    if (rc != SP_OK) return rc;
    for (uint32_t k = 0; k < n; ++k) if (ms[k]) {
        const CypMid& m = *ms[k];
        const uint32_t H = m.H, ns = seg_first[k + 1] - seg_first[k];
        std::vector<int32_t> cons_len(H);
        for (uint32_t h = 0; h < H; ++h) cons_len[h] = (int32_t)m.final_cons[h].size();
        ed[k].resize((size_t)ns * H); ov[k].resize((size_t)ns * H); kept[k].resize(ns);
        cyp_weights_from_alns(H, cons_len.data(), m.allowed.data(), ns, all.h_len.data() + seg_first[k], mine[k], ed[k].data(), ov[k].data(), kept[k].data(), af[k].empty() ? nullptr : af[k].data());
    }
    return SP_OK;
}

int32_t cyp_part_c2(sp_ctx* ctx, const sp_cyp_problem* pr, sp_cyp_call* call, char* consensus, uint32_t cons_cap, CypMid& m, const std::vector<uint64_t>& ed, const std::vector<double>& ov,
                    const std::vector<uint8_t>& kept, uint32_t n_all) {
    const uint32_t R = m.R, H = m.H;
    std::vector<Label>& labels = m.labels; const std::vector<std::string>& final_cons = m.final_cons; const std::vector<std::string>& deep_tail = m.deep.suffix;
    const std::vector<uint32_t>& seg_off = m.seg_off;
    int32_t rc = SP_OK;
    HostMarks hm(ctx);
    std::vector<int32_t> types(H); for (uint32_t h = 0; h < H; ++h) types[h] = labels[h].type;
    std::vector<uint32_t> read_index(R + 1), rco(R + 1), rwo(R + 1), w_seg(n_all + 1), chain_off, chain_items;
    std::vector<uint64_t> uniq(H); std::vector<uint8_t> fa(H);
    sp_chain_build_info info; uint32_t chain_cap = 4 * R + 16, item_cap = 8 * n_all + 64;
    for (;;) {
        chain_off.assign(chain_cap + 1, 0); chain_items.assign(item_cap, 0);
        rc = sp_cyp_build_chains(H, types.data(), R, seg_off.data(), ed.data(), kept.data(), read_index.data(), rco.data(), chain_off.data(), chain_cap,
                                 chain_items.data(), item_cap, rwo.data(), w_seg.data(), uniq.data(), fa.data(), &info);
        if (rc == SP_ERR_CAPACITY && (info.n_chains > chain_cap || info.n_items > item_cap)) { chain_cap = std::max(chain_cap, info.n_chains); item_cap = std::max(item_cap, info.n_items); continue; }
        break;
    }
    if (rc == SP_ERR_CHAIN_COLLAPSE) { call->status = SP_ERR_CHAIN_COLLAPSE; return SP_OK; }
    if (rc != SP_OK) return sp_fail(ctx, rc, "sp_cyp_diplotype: chain building");
    for (uint32_t h = 0; h < H; ++h) if (fa[h]) labels[h].type = SP_CYP_FALSE_ALLELE;                       // mark_false_allele (:574-583)
    call->n_consensus = (int32_t)H;
    std::vector<const char*> subs(H);
    for (uint32_t h = 0; h < H; ++h) {
        call->cons_type[h] = labels[h].type;
        std::snprintf(call->cons_subtype[h], sizeof call->cons_subtype[h], "%s", labels[h].has_sub ? labels[h].sub.c_str() : "");
        subs[h] = labels[h].has_sub ? labels[h].sub.c_str() : nullptr; types[h] = labels[h].type;
        if (consensus && cons_cap) std::snprintf(consensus + (size_t)h * cons_cap, cons_cap, "%s", final_cons[h].c_str());
    }
    std::vector<uint64_t> w_ed((size_t)info.n_rows * H); std::vector<double> w_ov((size_t)info.n_rows * H);
    for (uint32_t row = 0; row < info.n_rows; ++row) for (uint32_t h = 0; h < H; ++h) { w_ed[(size_t)row * H + h] = ed[(size_t)w_seg[row] * H + h]; w_ov[(size_t)row * H + h] = ov[(size_t)w_seg[row] * H + h]; }
    sp_chain_problem cp; std::memset(&cp, 0, sizeof cp);
    cp.n_haps = H; cp.hap_type = types.data(); cp.hap_subtype = subs.data();
    cp.n_translate = pr->n_translate; cp.translate_key = pr->translate_key; cp.translate_val = pr->translate_val;
    cp.n_connections = pr->n_connections; cp.connection_a = pr->connection_a; cp.connection_b = pr->connection_b;
    cp.n_singletons = pr->n_singletons; cp.singletons = pr->singletons;
    cp.n_reads = info.n_reads; cp.read_chain_off = rco.data(); cp.chain_off = chain_off.data(); cp.chain_items = chain_items.data();
    cp.read_w_off = rwo.data(); cp.w_ed = w_ed.data(); cp.w_ov = w_ov.data();
    cp.infer_connections = pr->infer_connections; cp.normalize_all_alleles = !pr->normalize_d6_only; cp.ignore_chain_label_limits = 0;
    cp.lasso_penalty = 4.0; cp.ln_ed_penalty = 2.0; cp.unexpected_chain_penalty = 10.0; cp.inferred_edge_penalty = 2.0;      // ChainPenalties::default (chaining.rs:107-139)
    sp_chain_result cr;
    hm.mark("host:cyp_chains");
    rc = sp_cyp_best_chain_pair(ctx, &cp, &cr);
    hm.mark("host:cyp_chain_pair");
    if (rc == SP_ERR_NO_CHAINING_HEAD || rc == SP_ERR_NO_CHAINS_FOUND || rc == SP_ERR_NO_SCORE_PAIRS) { call->status = rc; return SP_OK; }
    if (rc != SP_OK) return rc;
    call->n1 = cr.n1; call->n2 = cr.n2; call->score = cr.score;
    std::memcpy(call->chain1, cr.chain1, sizeof(int32_t) * cr.n1); std::memcpy(call->chain2, cr.chain2, sizeof(int32_t) * cr.n2);
    sp_cyp_chain_to_hap(cr.chain1, cr.n1, types.data(), subs.data(), pr->n_translate, pr->translate_key, pr->translate_val, 1, call->hap1, sizeof call->hap1);
    sp_cyp_chain_to_hap(cr.chain2, cr.n2, types.data(), subs.data(), pr->n_translate, pr->translate_key, pr->translate_val, 1, call->hap2, sizeof call->hap2);
    sp_cyp_chain_to_hap(cr.chain1, cr.n1, types.data(), subs.data(), pr->n_translate, pr->translate_key, pr->translate_val, 0, call->core1, sizeof call->core1);
    sp_cyp_chain_to_hap(cr.chain2, cr.n2, types.data(), subs.data(), pr->n_translate, pr->translate_key, pr->translate_val, 0, call->core2, sizeof call->core2);
    // Cyp2d6DetailLevel::DeepAlleles (caller.rs:907-957): the same walk with "(<index>_<full allele> <variants>)" as the label of a region
    auto deep_hap = [&](const int32_t* chain, uint32_t n, char* out, size_t cap2) {
        auto is_2d = [](int t) { return t == SP_CYP_CYP2D6 || t == SP_CYP_CYP2D7 || t == SP_CYP_DELETION || t == SP_CYP_HYBRID; };
        int non_deletion = 0;
        for (uint32_t x = 0; x < n; ++x) { const int t = types[chain[x]]; if (is_2d(t) && t != SP_CYP_CYP2D7 && t != SP_CYP_DELETION) ++non_deletion; }
        std::string res, prev; int run = 0;
        auto flush = [&]() { if (run > 0) { if (!res.empty()) res += " + "; res += prev; if (run > 1) res += "x" + std::to_string(run); } };
        for (int x = (int)n - 1; x >= 0; --x) {
            const int h = chain[x], t = types[h];
            if (!(is_2d(t) && t != SP_CYP_CYP2D7)) continue;
            if (t == SP_CYP_DELETION && non_deletion > 0) continue;
            const std::string cur = "(" + std::to_string(h) + "_" + label_full(labels[h]) + deep_tail[h] + ")";
            if (run > 0 && cur == prev) ++run; else { flush(); prev = cur; run = 1; }
        }
        flush();
        std::snprintf(out, cap2, "%s", res.c_str());
    };
    deep_hap(cr.chain1, cr.n1, call->deep1, sizeof call->deep1);
    deep_hap(cr.chain2, cr.n2, call->deep2, sizeof call->deep2);
    return SP_OK;
}


int32_t cyp_part_c(sp_ctx* ctx, const sp_cyp_problem* pr, const sp_seqset* reads, sp_cyp_call* call, char* consensus, uint32_t cons_cap, sp_cyp_region_variants* region_variants, CypMid& m,
                   TypeCache* shared_types = nullptr) {
    Ahead ahead; JoinAhead join_ahead{ahead};
    int32_t rc = cyp_part_c1(ctx, pr, reads, region_variants, m, shared_types, &ahead);
    if (rc != SP_OK) return rc;
    std::vector<uint64_t> ed; std::vector<double> ov; std::vector<uint8_t> kept; sp_seqset all_here; const sp_seqset* all = nullptr;
    rc = cyp_weights_one(ctx, reads, m, ahead, ed, ov, kept, all_here, all);
    if (rc != SP_OK) return rc;
    return cyp_part_c2(ctx, pr, call, consensus, cons_cap, m, ed, ov, kept, all->n);
}

} // namespace

extern "C" int32_t sp_cyp_diplotype_detailed(sp_ctx* ctx, const sp_cyp_problem* pr, const sp_seqset* reads, sp_cyp_call* call, char* consensus, uint32_t cons_cap,
                                             sp_cyp_region_variants* region_variants) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    CypMid m;
    int32_t rc = cyp_part_a(ctx, pr, reads, call, region_variants, "cypc", m);
    if (rc != SP_OK || m.finished) return rc;
    {
        HostScope hs(ctx, "host:cyp_consensus");
        sp_priority_job J; J.problem = &m.pp; J.max_groups = SP_CYP_MAXCONS; J.cap = m.cap; J.n_groups = &m.n_groups; J.group_of = m.group_of.data(); J.cons = m.text.data(); J.status = SP_OK; J.gave_up = 0;
        rc = sp_consensus_priority_many(ctx, 1, &J);
        if (rc == SP_OK && J.status == SP_ERR_CAPACITY) rc = sp_fail(ctx, SP_ERR_CAPACITY, "sp_consensus_priority: more groups than max_groups");
        else if (rc == SP_OK && J.status != SP_OK) rc = sp_fail(ctx, J.status, "sp_consensus_priority: more groups than reads");
        call->searches_gave_up = J.gave_up;
    }
    if (rc != SP_OK) return rc;
    return cyp_part_c(ctx, pr, reads, call, consensus, cons_cap, region_variants, m);
}

// The CYP2D6 calls of several samples (one GPU's share of a cohort): each sample alone is a chain of small launches that wait for one another, so
// GROUPS of samples are handed to the context and its helper streams (sp_ctx_set_option "hla_split_genes" / "cyp_cohort_streams"), one host thread
// per stream for the length of the call, and a group goes through every stage that allows it together.  Every call is the call sp_cyp_diplotype makes.
extern "C" int32_t sp_cyp_diplotype_cohort(sp_ctx* ctx, const sp_cyp_problem* pr, uint32_t n_samples, const sp_seqset* const* reads, sp_cyp_call* calls,
                                           char* consensus, uint32_t cons_cap, int32_t* sample_rc) {
    if (!ctx) return SP_ERR_INVALID_ARG;
    if (!pr || (n_samples && (!reads || !calls))) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_cyp_diplotype_cohort: null argument");
    for (uint32_t i = 0; i < n_samples; ++i) if (!reads[i]) return sp_fail(ctx, SP_ERR_INVALID_ARG, "sp_cyp_diplotype_cohort: null read set");
    if (n_samples == 0) return SP_OK;
    // what the first placement would build on the shared template set is built before the streams part
    if (pr->templates) { const int rc0 = sp_seqset_build_index(ctx, const_cast<sp_seqset*>(pr->templates)); if (rc0 != SP_OK) return rc0; }
    // streams: up to cyp_cohort_streams, but no more than leave a dozen samples per stream -- a stream keeps its samples in lockstep through the consensus and types
    // their consensuses in one batch, which pays with the size of the group (48 samples: 6.7 / 4.8 / 4.7 / 4.7 ms per sample on 1 / 2 / 4 / 6 streams)
    // (with the persistent consensus kernels at most four: a persistent batch holds two hardware queues for its whole length, and two of its own streams on ONE queue -- more
    //  streams in the process than queues -- is a batch whose step kernel waits behind its own control kernel for ever: the four-second time-out, seen with eight streams)
    const uint32_t stream_cap = ctx->k8_persistent ? std::min(ctx->cyp_cohort_streams, 4) : ctx->cyp_cohort_streams;
    int n_parts = ctx->split_genes ? (int)std::min<uint32_t>(std::max<uint32_t>(1, n_samples / (uint32_t)ctx->cyp_cohort_min_group), stream_cap) : 1;
    sp_ctx* on[8] = { ctx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    for (int x = 1; x < n_parts; ++x) { on[x] = sp_ctx_helper(ctx, x - 1); if (!on[x]) { n_parts = x; break; } }
    std::vector<int32_t> rcs(n_samples, SP_OK); std::vector<int> where(n_samples, 0);
    std::atomic<uint32_t> next(0);
    // A stream takes a GROUP of samples at a time and keeps them in lockstep through the multi-way consensus (sp_consensus_priority_many): the launches of
    // that stage -- two thirds of a sample's chain -- are then those of the group's slowest search, not their sum over its samples.  The group's read
    // sets are seen as one set (cyp_group_view): its regions of interest are one search, its weights one list of placements; its consensuses are typed in one
    // batch.  Consensus inputs, merge, chains and the best chain pair stay sample by sample.
    const uint32_t group = std::min<uint32_t>(64, std::max<uint32_t>(1, (n_samples + (uint32_t)n_parts - 1) / (uint32_t)n_parts));
    auto work = [&](int x) {
        sp_ctx* c = on[x];
        (void)hipSetDevice(c->device);             // a new host thread starts on device 0: everything below (pools, copies, launches) belongs to the context's device
        for (;;) {
            const uint32_t first = next.fetch_add(group);
            if (first >= n_samples) break;
            const uint32_t n = std::min<uint32_t>(group, n_samples - first);
            std::vector<std::unique_ptr<CypMid>> mids(n);
            std::vector<sp_priority_job> jobs; std::vector<uint32_t> job_of;
            try {
                std::vector<std::vector<sp_region_hit>> group_hits; int32_t rc_regions = SP_OK;
                sp_seqset view; std::vector<uint32_t> view_first;
                const bool viewed = pr->templates && pr->template_type && cyp_group_view(c, reads + first, n, "cypg", view, view_first, rc_regions);
                const bool searched = viewed && (rc_regions != SP_OK || cyp_group_regions(c, pr, view, view_first, n, group_hits, rc_regions));
                for (uint32_t k = 0; k < n; ++k) {
                    const uint32_t i = first + k;
                    where[i] = x;
                    mids[k].reset(new CypMid());
                    const std::string prefix = "cypc" + std::to_string(k);
                    if (searched && rc_regions != SP_OK) { rcs[i] = rc_regions; continue; }
                    rcs[i] = cyp_part_a(c, pr, reads[i], &calls[i], nullptr, prefix.c_str(), *mids[k], searched ? &group_hits[k] : nullptr);
                    if (rcs[i] != SP_OK || mids[k]->finished) continue;
                    CypMid& m = *mids[k];
                    sp_priority_job J; J.problem = &m.pp; J.max_groups = SP_CYP_MAXCONS; J.cap = m.cap; J.n_groups = &m.n_groups; J.group_of = m.group_of.data(); J.cons = m.text.data(); J.status = SP_OK; J.gave_up = 0;
                    jobs.push_back(J); job_of.push_back(k);
                }
                int32_t rc_all = SP_OK;
                if (!jobs.empty()) { HostScope hs(c, "host:cyp_consensus"); rc_all = sp_consensus_priority_many(c, (uint32_t)jobs.size(), jobs.data()); }
                // the consensuses of the whole group are typed in one batch (placement on the templates, graph alignment, allele scores: one launch sequence
                // instead of one per sample; equal sequences -- the common alleles of a cohort -- are typed once)
                TypeCache types;
                if (rc_all == SP_OK) {
                    std::vector<std::string> all;
                    for (size_t q = 0; q < jobs.size(); ++q) if (jobs[q].status == SP_OK) {
                        const CypMid& m = *mids[job_of[q]];
                        for (uint32_t g = 0; g < m.n_groups; ++g) all.emplace_back(m.text.data() + (size_t)(2 * g + 1) * m.cap);
                    }
                    HostScope hs(c, "host:cyp_merge");
                    rc_all = type_fresh(c, pr, all, 0.1, types);
                }
                // (c1) per sample, the weights of the group in one go (or sample by sample when its read sets cannot be seen as one), (c2) per sample
                std::vector<CypMid*> in_play(n, nullptr);
                for (size_t q = 0; q < jobs.size(); ++q) {
                    const uint32_t k = job_of[q], i = first + k;
                    if (rc_all != SP_OK) { rcs[i] = rc_all; continue; }
                    calls[i].searches_gave_up = jobs[q].gave_up;
                    if (jobs[q].status != SP_OK) { rcs[i] = jobs[q].status; c->err = jobs[q].status == SP_ERR_CAPACITY ? "sp_consensus_priority: more groups than max_groups" : "sp_consensus_priority: more groups than reads"; continue; }
                    rcs[i] = cyp_part_c1(c, pr, reads[i], nullptr, *mids[k], &types, nullptr);
                    if (rcs[i] == SP_OK) in_play[k] = mids[k].get();
                }
                std::vector<std::vector<uint64_t>> ed; std::vector<std::vector<double>> ov; std::vector<std::vector<uint8_t>> kept;
                const bool together = viewed && rc_regions == SP_OK;
                if (together) {
                    const int32_t rc_w = cyp_weights_group(c, view, view_first, in_play, ed, ov, kept);
                    if (rc_w != SP_OK) for (uint32_t k = 0; k < n; ++k) if (in_play[k]) { rcs[first + k] = rc_w; in_play[k] = nullptr; }
                }
                for (uint32_t k = 0; k < n; ++k) if (in_play[k]) {
                    const uint32_t i = first + k;
                    char* cons_out = consensus ? consensus + (size_t)i * SP_CYP_MAXCONS * cons_cap : nullptr;
                    if (together) { rcs[i] = cyp_part_c2(c, pr, &calls[i], cons_out, cons_cap, *mids[k], ed[k], ov[k], kept[k], (uint32_t)kept[k].size()); continue; }
                    Ahead none; std::vector<uint64_t> e1; std::vector<double> o1; std::vector<uint8_t> k1; sp_seqset all_here; const sp_seqset* all = nullptr;
                    rcs[i] = cyp_weights_one(c, reads[i], *mids[k], none, e1, o1, k1, all_here, all);
                    if (rcs[i] == SP_OK) rcs[i] = cyp_part_c2(c, pr, &calls[i], cons_out, cons_cap, *mids[k], e1, o1, k1, all->n);
                }
            }
            catch (const std::bad_alloc&) { for (uint32_t k = 0; k < n; ++k) rcs[first + k] = SP_ERR_OUT_OF_MEMORY; c->err = "sp_cyp_diplotype: out of host memory"; }      // (an exception must not leave a thread, nor
            catch (const std::exception& e) { for (uint32_t k = 0; k < n; ++k) rcs[first + k] = SP_ERR_INVALID_ARG; c->err = std::string("sp_cyp_diplotype: ") + e.what(); }   // the caller while threads are joinable)
        }
    };
    std::thread beside[8]; bool started[8] = { false, false, false, false, false, false, false, false };
    for (int x = 1; x < n_parts; ++x) {
        try { beside[x] = std::thread(work, x); started[x] = true; }
        catch (const std::system_error&) { }
    }
    work(0);
    for (int x = 1; x < n_parts; ++x) if (started[x]) beside[x].join();
    int32_t rc = SP_OK;
    for (uint32_t i = 0; i < n_samples; ++i) {
        if (sample_rc) sample_rc[i] = rcs[i];
        if (rcs[i] != SP_OK && rc == SP_OK) { rc = rcs[i]; if (where[i] > 0) ctx->err = on[where[i]]->err; }
    }
    return rc;
}

// cyp2d6_alleles.json: DeeplotypeDebug (src/cyp2d6/debug.rs:10-70) written as save_json writes it (serde_json pretty print)
extern "C" int32_t sp_cyp_alleles_json(const sp_cyp_problem* pr, const sp_cyp_call* call, const sp_cyp_region_variants* rv, char* out, uint64_t cap, uint64_t* needed) {
    if (!pr || !call || !rv) return SP_ERR_INVALID_ARG;
    static const char* names[] = { "Unknown", "Match", "Unexpected", "Missing", "AmbiguousUnexpected", "AmbiguousMissing", "UnknownUnexpected", "UnknownMissing" };
    spj::Value root = spj::object();
    const char* forms[2][3] = { { call->deep1, call->hap1, call->core1 }, { call->deep2, call->hap2, call->core2 } };
    for (int h = 0; h < 2; ++h) {
        spj::Value hap = spj::object();
        hap.obj.emplace_back("deep_form", spj::str(forms[h][0])); hap.obj.emplace_back("suballele_form", spj::str(forms[h][1])); hap.obj.emplace_back("core_form", spj::str(forms[h][2]));
        root.obj.emplace_back(h == 0 ? "hap1" : "hap2", std::move(hap));
    }
    std::map<std::string, spj::Value> alleles;                        // BTreeMap<String, Vec<RegionVariant>> keyed by index_label()
    for (int32_t h = 0; h < call->n_consensus && h < SP_CYP_MAXCONS; ++h) {
        if (!rv->has_variants[h]) continue;
        if (pr->n_variants && (!rv->state || !pr->var_label || !pr->var_is_vi)) return SP_ERR_INVALID_ARG;
        spj::Value list = spj::array();
        for (uint32_t v = 0; v < pr->n_variants; ++v) {
            const uint8_t st = rv->state[(size_t)h * pr->n_variants + v];
            if (st > 7) continue;
            spj::Value e = spj::object();
            e.obj.emplace_back("label", spj::str(pr->var_label[v])); e.obj.emplace_back("is_vi", spj::boolean(pr->var_is_vi[v] != 0)); e.obj.emplace_back("variant_state", spj::str(names[st]));
            list.arr.push_back(std::move(e));
        }
        alleles.emplace(std::to_string(h) + "_" + full_allele(call->cons_type[h], call->cons_subtype[h][0] ? call->cons_subtype[h] : nullptr), std::move(list));
    }
    spj::Value amap = spj::object();
    for (auto& kv : alleles) amap.obj.emplace_back(kv.first, std::move(kv.second));
    root.obj.emplace_back("alleles", std::move(amap));
    std::string text;
    spj::write_pretty(text, root);
    if (needed) *needed = text.size() + 1;
    if (out && cap) { const size_t n = std::min<size_t>(text.size(), (size_t)cap - 1); std::memcpy(out, text.data(), n); out[n] = 0; if (n < text.size()) return SP_ERR_CAPACITY; }
    return SP_OK;
}
