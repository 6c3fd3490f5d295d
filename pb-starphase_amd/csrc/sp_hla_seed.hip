// sp_hla_seed.hip -- K1 in the reference's CALL PATTERN (gfx950): realign_record maps a read against the index of every DNA allele with minimap2
// (`map-hifi`, best_n 5: src/util/mapping.rs:8-14) and only looks at the mappings minimap2 returns (src/hla/realigner.rs:116-146) -- the chains its seeding
// and chaining rank best, not every allele.  This file is that stage on the device:
//   index     (w,k) = (19,19) minimizers of the hg38-strand alleles, one sorted table (hash -> occurrences), the occurrence threshold mid_occ
//   seeds     per read: minimizers -> look-up -> occurrence filter with the rescue of long high-occurrence streaks       k1s_seed_kernel
//   chains    per read: anchors grouped by (strand, allele) -> chaining DP per allele -> chains ranked, primaries /
//             secondaries (mask_level 0.5), the best_n secondaries within pri_ratio 0.8                                   k1s_chain_kernel
//   mappings  the selected chains (<= 16 per read) base-aligned by the library's own cell + two-piece affine re-score (sp_wfa.hip.h, sp_affine.hip),
//             ordered by peak score, selected once more, and put through realign_record's acceptance loop                k1s_pick_kernel
// CPU statement, bit for bit: oracle/mm2.c (collect_anchors, chain_anchors, set_parent, select_sub, omm_hla_k1_seeded).  The exhaustive search of
// sp_hla.hip (every allele of every anchored gene) stays available: context option "k1_best_n" = 0.
#include <cstring>
#include "sp_internal.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>

namespace {

// ---- the reference's settings (map-hifi; oracle/mm2.c omm_default_opts)
constexpr int MZ_K = 19, MZ_W = 19;
constexpr uint64_t MZ_MASK = (1ull << (2 * MZ_K)) - 1;
constexpr int MIN_MID_OCC = 50, MAX_MID_OCC = 500, MAX_MAX_OCC = 4095, OCC_DIST = 500;
constexpr float MID_OCC_FRAC = 2e-4f;
constexpr int CH_MAX_GAP = 10000, CH_BW = 500, CH_MAX_SKIP = 25, CH_MAX_ITER = 5000, CH_MIN_CNT = 3, CH_MIN_SCORE = 40;
constexpr float MASK_LEVEL = 0.5f, PRI_RATIO = 0.8f;
constexpr int MIN_DP_MAX = 200;

constexpr int BUCKET_BITS = 18;                 // look-up: the top bits of the 38-bit hash pick a bucket of the sorted key table
constexpr uint64_t H_PAL = 1ull << 40;          // a k-mer that is its own reverse complement: never a minimizer, larger than every hash
constexpr uint64_t H_NONE = ~0ull;              // no k-mer ends here (sequence boundary, ambiguous base): a stretch ends
constexpr int TILE = 1024, HALO = MZ_W - 1;
constexpr int SEL_CAP = SP_K1_SEL;              // selected chains per read that are base-aligned
constexpr int PRIM_CAP = 64;

struct IndexView {
    const uint64_t* keys; const uint32_t* start; const uint32_t* occ; const uint32_t* bucket;
    uint32_t n_keys; int32_t mid_occ; uint32_t n_seqs;
};

// the invertible integer hash of the sketch (oracle/mm2.c mix64)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t key) {
    key = (~key + (key << 21)) & MZ_MASK;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & MZ_MASK;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & MZ_MASK;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & MZ_MASK;
    return key;
}
// the 32 two-bit digits of v in reverse order
__device__ __forceinline__ uint64_t rev_digits(uint64_t v) {
    v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
    v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(v);
}
// 38 bits starting at base `start` of a packed sequence (first base in the lowest bits); the set's guard words make the third word readable
__device__ __forceinline__ uint64_t bits38(const uint32_t* __restrict__ w, int start) {
    const int wi = start >> 4, sh = (start & 15) << 1;
    const uint64_t lo = (uint64_t)w[wi] | ((uint64_t)w[wi + 1] << 32);
    uint64_t e = lo >> sh;
    if (sh) e |= (uint64_t)w[wi + 2] << (64 - sh);
    return e & MZ_MASK;
}
// hash (or H_PAL / H_NONE) and strand of the k-mer that ENDS at position i
__device__ __forceinline__ uint64_t kmer_hash(const uint32_t* __restrict__ w, const uint32_t* __restrict__ np, int len, int i, uint8_t& z) {
    z = 0;
    if (i < MZ_K - 1 || i >= len) return H_NONE;
    const int start = i - (MZ_K - 1);
    if (np && bits38(np, start)) return H_NONE;
    const uint64_t e = bits38(w, start);
    const uint64_t fw = rev_digits(e) >> (64 - 2 * MZ_K), rv = (~e) & MZ_MASK;
    if (fw == rv) return H_PAL;
    z = fw < rv ? 0 : 1;
    return mix64(z ? rv : fw);
}
// is the k-mer at tile slot c a minimizer?  (w,k)-minimizers with every tied minimum of a window kept; a stretch shorter than a window gives its last smallest k-mer
__device__ __forceinline__ bool is_minimizer(const uint64_t* __restrict__ h, int c) {
    const uint64_t hp = h[c];
    if (hp >= H_PAL) return false;
    int l = 0; bool lb = false;
    for (; l < HALO; ++l) { const uint64_t v = h[c - 1 - l]; if (v == H_NONE) { lb = true; break; } if (v < hp) break; }
    int r = 0; bool rb = false, tie = false;
    for (; r < HALO; ++r) { const uint64_t v = h[c + 1 + r]; if (v == H_NONE) { rb = true; break; } if (v < hp) break; if (v == hp) tie = true; }
    if (l + r >= MZ_W - 1) return true;
    return lb && rb && !tie;
}
// rank of the calling thread among the threads of the block that pass `flag`, in thread order; total = how many do
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* wave_tot, uint32_t& total) {
    const unsigned long long b = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t within = (uint32_t)__builtin_popcountll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wave_tot[wave] = (uint32_t)__builtin_popcountll(b);
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int x = 0; x < nw; ++x) { const uint32_t c = wave_tot[x]; if (x < wave) base += c; tot += c; }
    total = tot;
    return base + within;
}
__device__ __forceinline__ void index_lookup(const IndexView& ix, uint64_t hash, uint32_t& st, uint32_t& n) {
    const uint32_t b = (uint32_t)(hash >> (2 * MZ_K - BUCKET_BITS));
    uint32_t lo = ix.bucket[b], hi = ix.bucket[b + 1];
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ix.keys[mid] < hash) lo = mid + 1; else hi = mid; }
    if (lo < ix.n_keys && lo < ix.bucket[b + 1] && ix.keys[lo] == hash) { st = ix.start[lo]; n = ix.start[lo + 1] - st; } else { st = 0; n = 0; }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// the sketch of every sequence of a set (index building, and the audit entry point): one workgroup per sequence.  FILL = false counts, FILL = true
// writes (hash, rid << 17 | end position << 1 | strand) in position order at off[s]
template <bool FILL>
__global__ __launch_bounds__(256) void mz_sketch_kernel(SeqSetView S, const uint32_t* __restrict__ seq_of, uint32_t n, uint32_t* __restrict__ cnt, const uint64_t* __restrict__ off,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ occ) {
    __shared__ uint64_t h[TILE + 2 * HALO];
    __shared__ uint8_t zs[TILE + 2 * HALO];
    __shared__ uint32_t wave_tot[4];
    const uint32_t rid = blockIdx.x;
    if (rid >= n) return;
    const uint32_t s = seq_of ? seq_of[rid] : rid;
    const int len = S.len[s];
    const uint32_t* w = S.words + S.word_off[s];
    const uint32_t* np = S.nplane ? S.nplane + S.word_off[s] : nullptr;
    uint32_t running = 0;
    for (int lo = 0; lo < len; lo += TILE) {
        __syncthreads();
        for (int c = threadIdx.x; c < TILE + 2 * HALO; c += 256) { uint8_t z; h[c] = kmer_hash(w, np, len, lo - HALO + c, z); zs[c] = z; }
        __syncthreads();
        for (int j = 0; j < TILE; j += 256) {
            const int c = HALO + j + (int)threadIdx.x, pos = lo + j + (int)threadIdx.x;
            const bool flag = pos < len && is_minimizer(h, c);
            uint32_t total;
            const uint32_t rk = block_rank(flag, wave_tot, total);
            if (FILL && flag) {
                keys[off[rid] + running + rk] = h[c];
                occ[off[rid] + running + rk] = rid << 17 | (uint32_t)pos << 1 | zs[c];
            }
            running += total;
        }
    }
    if (!FILL && threadIdx.x == 0) cnt[rid] = running;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// seeds of a read: minimizers that occur in the index, in query order, through the occurrence filter (oracle/mm2.c collect_anchors):
// a seed with <= mid_occ occurrences is kept; of a streak of seeds above it the `(streak length on the query) / 500` ones with the fewest
// occurrences are kept when they have <= 4095.  seeds[]: {query end position << 1 | strand, first occurrence, occurrences, 0}
struct SeedCounters { unsigned long long seeds, anchors; uint32_t max_anchors, max_seeds, overflow_reads, seed_overflow, rev_selected, pad; };

__global__ __launch_bounds__(256) void k1s_seed_kernel(SeqSetView reads, IndexView ix, uint32_t n_reads, int sd_cap, uint4* __restrict__ seeds, uint32_t seed_cap,
                                                       uint2* __restrict__ read_seed, uint32_t* __restrict__ read_anchors, SeedCounters* __restrict__ ctr) {
    extern __shared__ uint32_t lds[];
    uint64_t* h = reinterpret_cast<uint64_t*>(lds);                      // TILE + 2 HALO
    uint32_t* sd_q = lds + 2 * (TILE + 2 * HALO);                        // sd_cap each: query end position << 1 | strand, first occurrence
    uint32_t* sd_st = sd_q + sd_cap;
    uint16_t* sd_n = reinterpret_cast<uint16_t*>(sd_st + sd_cap);       // occurrences, capped at 65535 (only values <= 4095 are ever told apart, see below)
    uint8_t* zs = reinterpret_cast<uint8_t*>(sd_n + sd_cap);            // TILE + 2 HALO
    uint8_t* keep = zs + TILE + 2 * HALO + 8;                            // sd_cap
    __shared__ uint32_t wave_tot[4], s_sum[4];
    __shared__ unsigned long long s_base;
    const uint32_t r = blockIdx.x;
    if (r >= n_reads) return;
    const int len = reads.len[r];
    const uint32_t* w = reads.words + reads.word_off[r];
    const uint32_t* np = reads.nplane ? reads.nplane + reads.word_off[r] : nullptr;
    uint32_t ns = 0; bool over = false;
    for (int lo = 0; lo < len; lo += TILE) {
        __syncthreads();
        for (int c = threadIdx.x; c < TILE + 2 * HALO; c += 256) { uint8_t z; h[c] = kmer_hash(w, np, len, lo - HALO + c, z); zs[c] = z; }
        __syncthreads();
        for (int j = 0; j < TILE; j += 256) {
            const int c = HALO + j + (int)threadIdx.x, pos = lo + j + (int)threadIdx.x;
            bool flag = pos < len && is_minimizer(h, c);
            uint32_t st = 0, n = 0;
            if (flag) { index_lookup(ix, h[c], st, n); flag = n > 0; }
            uint32_t total;
            const uint32_t rk = block_rank(flag, wave_tot, total);
            if (flag && ns + rk < (uint32_t)sd_cap) { sd_q[ns + rk] = (uint32_t)pos << 1 | zs[c]; sd_st[ns + rk] = st; sd_n[ns + rk] = (uint16_t)(n > 0xFFFFu ? 0xFFFFu : n); }
            if (ns + total > (uint32_t)sd_cap) over = true;
            ns += total;
        }
    }
    if (over) ns = (uint32_t)sd_cap;            // (counted below; a read with more seeds than fit is mapped with the first sd_cap of them)
    __syncthreads();
    // the occurrence filter.  A seed's rank inside its streak is by (occurrences, position); only ranks of seeds with <= 4095 occurrences matter (the others are dropped
    // whatever their rank), and every seed ahead of such a one has fewer occurrences still: capping the stored counts at 65535 changes no decision
    const int max_occ = ix.mid_occ;
    for (uint32_t j = threadIdx.x; j < ns; j += 256) {
        const uint32_t nj = sd_n[j];
        uint8_t k = 1;
        if ((int)nj > max_occ) {
            int a = (int)j - 1; while (a >= 0 && (int)sd_n[a] > max_occ) --a;               // last0
            uint32_t b = j + 1; while (b < ns && (int)sd_n[b] > max_occ) ++b;                // the seed that ends the streak
            const int ps = a < 0 ? 0 : (int)(sd_q[a] >> 1), pe = b == ns ? len : (int)(sd_q[b] >> 1);
            int kp = (2 * (pe - ps) + 499) / 1000;                                           // (int)((pe - ps) / 500.0 + .499)
            if (kp > 128) kp = 128;
            k = 0;
            if (kp > 0 && nj <= (uint32_t)MAX_MAX_OCC) {
                int rank = 0;
                for (uint32_t x = (uint32_t)(a + 1); x < b && rank < kp; ++x) { const uint32_t nx = sd_n[x]; if (nx < nj || (nx == nj && x < j)) ++rank; }
                k = rank < kp ? 1 : 0;
            }
        }
        keep[j] = k;
    }
    __syncthreads();
    // the kept seeds in query order, one reservation per read; anchors of the read = the occurrences of its kept seeds
    uint32_t kept = 0, mine = 0;
    for (uint32_t j = threadIdx.x; j < ns; j += 256) if (keep[j]) { ++kept; mine += sd_n[j]; }
    for (int o = 32; o > 0; o >>= 1) { kept += __shfl_xor(kept, o); mine += __shfl_xor(mine, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { wave_tot[threadIdx.x >> 6] = kept; s_sum[threadIdx.x >> 6] = mine; }
    __syncthreads();
    kept = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    const uint32_t tot = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    __syncthreads();
    if (threadIdx.x == 0) s_base = kept ? atomicAdd(&ctr->seeds, (unsigned long long)kept) : 0ull;
    __syncthreads();
    const unsigned long long base = s_base;
    const bool fits = base + kept <= seed_cap;
    uint32_t done = 0;
    for (uint32_t j0 = 0; j0 < ns && fits; j0 += 256) {
        const uint32_t j = j0 + threadIdx.x;
        const bool flag = j < ns && keep[j];
        uint32_t total;
        const uint32_t rk = block_rank(flag, wave_tot, total);
        if (flag) seeds[base + done + rk] = make_uint4(sd_q[j], sd_st[j], sd_n[j], 0u);
        done += total;
    }
    if (threadIdx.x == 0) {
        read_seed[r] = fits ? make_uint2((uint32_t)base, kept) : make_uint2(0u, 0u);
        read_anchors[r] = fits ? tot : 0u;
        if (fits && tot) { atomicAdd(&ctr->anchors, (unsigned long long)tot); atomicMax(&ctr->max_anchors, tot); atomicMax(&ctr->max_seeds, kept); }
        if (over) atomicAdd(&ctr->overflow_reads, 1u);
        if (!fits) atomicAdd(&ctr->seed_overflow, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// chains of a read.  One workgroup per read at a time (the grid strides over the reads), a private scratch region in device memory per workgroup.
// Per window of targets (strand, [rid_lo, rid_lo + W)): count the anchors of every target in an LDS table, lay the targets with >= 3 anchors out one
// after the other (their anchors as target end position << 16 | query end position on the mapped strand: ascending = the order of oracle/mm2.c an_cmp),
// scatter seed by seed so that a colinear target's anchors arrive sorted, then the chaining DP of chain_anchors, one thread per target, out of LDS.
struct SChain { int32_t score, f_end; uint32_t key, end, first, cnt; };       // key = rev << 15 | rid; end / first = target end position << 16 | query end position (strand coordinates)
struct SGroup { uint32_t key, off, cnt; };
struct SSel { int32_t rid, rev, score, cnt, diag, qs, qe, rs, re, n_chains; };        // qs / qe in forward coordinates

constexpr int CH_THREADS = 256;
constexpr int DP_CAP = 3584;                    // anchors a DP round holds in LDS (12 bytes each)
constexpr int WIN_KEYS = 12288;                 // targets per window (4 bytes each in LDS)

struct ChainScratch { uint32_t* anchors; SGroup* groups; uint32_t* perm; SChain* chains; int32_t* parent; uint32_t anchor_cap, group_cap, chain_cap; };

__device__ __forceinline__ int32_t chain_sc(uint32_t ai, uint32_t aj, const int32_t* __restrict__ pen) {
    const int32_t dq = (int32_t)(ai & 0xFFFFu) - (int32_t)(aj & 0xFFFFu);
    if (dq <= 0 || dq > CH_MAX_GAP) return INT32_MIN;
    const int32_t dr = (int32_t)(ai >> 16) - (int32_t)(aj >> 16);
    if (dr == 0) return INT32_MIN;
    const int32_t dd = dr > dq ? dr - dq : dq - dr;
    if (dd > CH_BW) return INT32_MIN;
    const int32_t dg = dr < dq ? dr : dq;
    int32_t sc = MZ_K < dg ? MZ_K : dg;
    if (dd || dg > MZ_K) sc -= pen[dd];               // (int)(0.01 * 0.8 * k * dd + 0.5 * log2(dd + 1)), tabulated on the host
    return sc;
}

// the block's best (largest) 128-bit key and its index; -1 when no thread offers one.  All threads call it.
__device__ __forceinline__ int block_argmax(bool have, unsigned long long k1, unsigned long long k2, int idx, unsigned long long* r1, unsigned long long* r2, int* ri) {
    for (int o = 32; o > 0; o >>= 1) {
        const bool oh = __shfl_xor((int)have, o) != 0;
        const unsigned long long o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o); const int oi = __shfl_xor(idx, o);
        if (oh && (!have || o1 > k1 || (o1 == k1 && o2 > k2))) { have = true; k1 = o1; k2 = o2; idx = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { const int wv = threadIdx.x >> 6; r1[wv] = k1; r2[wv] = k2; ri[wv] = have ? idx : -1; }
    __syncthreads();
    int best = -1; unsigned long long b1 = 0, b2 = 0;
    for (int wv = 0; wv < CH_THREADS / 64; ++wv) if (ri[wv] >= 0 && (best < 0 || r1[wv] > b1 || (r1[wv] == b1 && r2[wv] > b2))) { best = ri[wv]; b1 = r1[wv]; b2 = r2[wv]; }
    return best;
}

__device__ __forceinline__ void chain_interval(const SChain& c, int qlen, int& qs, int& qe, int& rs, int& re) {
    rs = (int)(c.first >> 16) + 1 - MZ_K; re = (int)(c.end >> 16) + 1;
    const int s = (int)(c.first & 0xFFFFu) + 1 - MZ_K, e = (int)(c.end & 0xFFFFu) + 1;
    if (c.key >> 15) { qs = qlen - e; qe = qlen - s; } else { qs = s; qe = e; }
}

__global__ __launch_bounds__(CH_THREADS) void k1s_chain_kernel(SeqSetView reads, IndexView ix, uint32_t n_reads, const uint4* __restrict__ seeds, const uint2* __restrict__ read_seed,
                                                               const int32_t* __restrict__ pen_tab, uint8_t* __restrict__ scratch, size_t scratch_stride, uint32_t anchor_cap,
                                                               uint32_t group_cap, uint32_t chain_cap, SSel* __restrict__ sel, uint32_t* __restrict__ sel_cnt, SeedCounters* __restrict__ ctr,
                                                               int best_n, SChain* __restrict__ dbg_chains, int32_t* __restrict__ dbg_parent, uint32_t* __restrict__ dbg_n, uint32_t dbg_read) {
    extern __shared__ uint32_t lds[];
    // [table / DP arrays (union)] [penalty table 512]
    uint32_t* tab = lds;
    uint32_t* dp_key = lds; int32_t* dp_f = reinterpret_cast<int32_t*>(lds + DP_CAP); int16_t* dp_p = reinterpret_cast<int16_t*>(lds + 2 * DP_CAP); int16_t* dp_t = dp_p + DP_CAP;
    constexpr int UNION_WORDS = WIN_KEYS > 3 * DP_CAP ? WIN_KEYS : 3 * DP_CAP;
    int32_t* pen = reinterpret_cast<int32_t*>(lds + UNION_WORDS);
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_scan[CH_THREADS / 64];
    __shared__ uint32_t s_u[8];
    __shared__ unsigned long long s_r1[CH_THREADS / 64], s_r2[CH_THREADS / 64];
    __shared__ int s_ri[CH_THREADS / 64];
    __shared__ int s_prim[PRIM_CAP];
    __shared__ int s_sel[PRIM_CAP + 8];
    const int tid = threadIdx.x;
    for (int i = tid; i <= CH_BW; i += CH_THREADS) pen[i] = pen_tab[i];
    uint8_t* my = scratch + (size_t)blockIdx.x * scratch_stride;
    uint32_t* sc_anchor = reinterpret_cast<uint32_t*>(my);
    SGroup* sc_group = reinterpret_cast<SGroup*>(sc_anchor + anchor_cap);
    uint32_t* sc_perm = reinterpret_cast<uint32_t*>(sc_group + group_cap);
    SChain* sc_chain = reinterpret_cast<SChain*>(sc_perm + group_cap);
    int32_t* sc_parent = reinterpret_cast<int32_t*>(sc_chain + chain_cap);
    for (uint32_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        __syncthreads();
        const uint2 rs_ = read_seed[r];
        const uint32_t s0 = rs_.x, K = rs_.y;
        const int qlen = reads.len[r];
        if (tid == 0) s_u[0] = 0;                         // chains of this read
        __syncthreads();
        for (int rev = 0; rev < 2 && K; ++rev) {
            for (uint32_t rid_lo = 0; rid_lo < ix.n_seqs; rid_lo += WIN_KEYS) {
                const uint32_t rid_hi = rid_lo + WIN_KEYS < ix.n_seqs ? rid_lo + WIN_KEYS : ix.n_seqs, W = rid_hi - rid_lo;
                __syncthreads();
                for (uint32_t i = tid; i < W; i += CH_THREADS) tab[i] = 0;
                __syncthreads();
                // A. anchors per target
                for (uint32_t s = 0; s < K; ++s) {
                    const uint4 sd = seeds[s0 + s];
                    const uint32_t strand = sd.x & 1u;
                    for (uint32_t k = tid; k < sd.z; k += CH_THREADS) {
                        const uint32_t o = ix.occ[sd.y + k], rid = o >> 17;
                        if ((int)((o & 1u) != strand) == rev && rid >= rid_lo && rid < rid_hi) atomicAdd(&tab[rid - rid_lo], 1u);
                    }
                }
                __syncthreads();
                // B. the targets with >= min_cnt anchors, one after the other: tab[] becomes the running slot of each (0xFFFFFFFF: dropped)
                uint32_t run_a = 0, run_g = 0;            // uniform: anchors / groups laid out so far in this window
                for (uint32_t i0 = 0; i0 < W; i0 += CH_THREADS) {
                    const uint32_t i = i0 + tid;
                    const uint32_t c = i < W ? tab[i] : 0u;
                    const bool kept = c >= (uint32_t)CH_MIN_CNT;
                    // exclusive prefix of the kept counts in thread order
                    uint32_t v = kept ? c : 0u, incl = v;
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((tid & 63) >= o) incl += t; }
                    uint32_t total_g;
                    const uint32_t g_rank = block_rank(kept, s_scan, total_g);
                    __syncthreads();
                    if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
                    __syncthreads();
                    uint32_t base = 0, tot = 0;
                    for (int wv = 0; wv < CH_THREADS / 64; ++wv) { if (wv < (tid >> 6)) base += s_scan[wv]; tot += s_scan[wv]; }
                    const uint32_t off = run_a + base + incl - v;
                    const bool placed = kept && off + c <= anchor_cap && run_g + g_rank < group_cap;
                    if (i < W) tab[i] = placed ? off : 0xFFFFFFFFu;
                    if (placed) { SGroup g; g.key = (uint32_t)rev << 15 | (rid_lo + i); g.off = off; g.cnt = c; sc_group[run_g + g_rank] = g; }
                    run_a += tot; run_g += total_g;
                    __syncthreads();
                }
                if (run_g == 0) continue;
                if (run_a > anchor_cap || run_g > group_cap) { if (tid == 0) atomicAdd(&ctr->overflow_reads, 1u); if (run_g > group_cap) run_g = group_cap; }
                // C. scatter, seed by seed in the order of the query on the mapped strand
                for (uint32_t step = 0; step < K; ++step) {
                    const uint32_t s = rev ? K - 1 - step : step;
                    const uint4 sd = seeds[s0 + s];
                    const uint32_t strand = sd.x & 1u, qpos = sd.x >> 1;
                    const uint32_t qp = rev ? (uint32_t)(qlen - ((int)qpos + 1 - MZ_K) - 1) : qpos;
                    for (uint32_t k = tid; k < sd.z; k += CH_THREADS) {
                        const uint32_t o = ix.occ[sd.y + k], rid = o >> 17;
                        if ((int)((o & 1u) != strand) == rev && rid >= rid_lo && rid < rid_hi && tab[rid - rid_lo] != 0xFFFFFFFFu) {
                            const uint32_t slot = atomicAdd(&tab[rid - rid_lo], 1u);
                            sc_anchor[slot] = ((o >> 1) & 0xFFFFu) << 16 | (qp & 0xFFFFu);
                        }
                    }
                    __syncthreads();
                }
                __threadfence_block();
                __syncthreads();
                // D. the targets by size, largest first (the threads of a wave then run DPs of like length)
                s_hist[tid] = 0;
                __syncthreads();
                for (uint32_t g = tid; g < run_g; g += CH_THREADS) { const uint32_t c = sc_group[g].cnt; atomicAdd(&s_hist[255 - (c > 255 ? 255 : c)], 1u); }
                __syncthreads();
                {
                    uint32_t v = s_hist[tid], incl = v;
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((tid & 63) >= o) incl += t; }
                    if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
                    __syncthreads();
                    uint32_t base = 0; for (int wv = 0; wv < (tid >> 6); ++wv) base += s_scan[wv];
                    __syncthreads();
                    s_hist[tid] = base + incl - v;
                }
                __syncthreads();
                for (uint32_t g = tid; g < run_g; g += CH_THREADS) { const uint32_t c = sc_group[g].cnt; sc_perm[atomicAdd(&s_hist[255 - (c > 255 ? 255 : c)], 1u)] = g; }
                __threadfence_block();
                __syncthreads();
                // rounds of the DP: the next targets of the size order that fit DP_CAP anchors together, one thread each
                uint32_t g0 = 0;
                while (g0 < run_g) {
                    const uint32_t gi = g0 + tid;
                    SGroup grp; grp.key = 0; grp.off = 0; grp.cnt = 0;
                    if (gi < run_g) grp = sc_group[sc_perm[gi]];
                    uint32_t v = grp.cnt, incl = v;
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((tid & 63) >= o) incl += t; }
                    __syncthreads();
                    if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
                    __syncthreads();
                    uint32_t base = 0; for (int wv = 0; wv < (tid >> 6); ++wv) base += s_scan[wv];
                    const uint32_t b0 = base + incl - v;                                 // my slice [b0, b0 + cnt)
                    const bool in_round = gi < run_g && b0 + v <= (uint32_t)DP_CAP;       // (sizes descend: whoever fits is a prefix of the round)
                    uint32_t n_round;
                    (void)block_rank(in_round, s_scan, n_round);
                    if (n_round == 0) {
                        // a target with more anchors than a round holds: not chained (counted); never seen with the reference's settings on an allele set
                        if (tid == 0) atomicAdd(&ctr->overflow_reads, 1u);
                        g0 += 1;
                        continue;
                    }
                    if (in_round) {
                        const int n = (int)grp.cnt;
                        uint32_t* key = dp_key + b0; int32_t* f = dp_f + b0; int16_t* p = dp_p + b0; int16_t* t = dp_t + b0;
                        bool sorted = true;
                        for (int i = 0; i < n; ++i) { key[i] = sc_anchor[grp.off + i]; if (i && key[i] <= key[i - 1]) sorted = false; t[i] = 0; }
                        if (!sorted) for (int i = 1; i < n; ++i) { const uint32_t kx = key[i]; int j = i - 1; while (j >= 0 && key[j] > kx) { key[j + 1] = key[j]; --j; } key[j + 1] = kx; }
                        // chain_anchors (oracle/mm2.c), the anchors of ONE target
                        int st = 0, max_ii = -1;
                        for (int i = 0; i < n; ++i) {
                            const uint32_t ai = key[i]; const int ri = (int)(ai >> 16);
                            int max_j = -1, max_f = MZ_K, n_skip = 0;
                            while (st < i && ri > (int)(key[st] >> 16) + CH_MAX_GAP) ++st;
                            if (i - st > CH_MAX_ITER) st = i - CH_MAX_ITER;
                            int j;
                            for (j = i - 1; j >= st; --j) {
                                int32_t sc = chain_sc(ai, key[j], pen);
                                if (sc == INT32_MIN) continue;
                                sc += f[j];
                                if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
                                else if (t[j] == i + 1) { if (++n_skip > CH_MAX_SKIP) break; }
                                if (p[j] >= 0) t[p[j]] = (int16_t)(i + 1);
                            }
                            const int end_j = j;
                            if (max_ii < 0 || ri - (int)(key[max_ii] >> 16) > CH_MAX_GAP) {
                                int mx = INT32_MIN; max_ii = -1;
                                for (j = i - 1; j >= st; --j) if (mx < f[j]) { mx = f[j]; max_ii = j; }
                            }
                            if (max_ii >= 0 && max_ii < end_j) {
                                const int32_t tmp = chain_sc(ai, key[max_ii], pen);
                                if (tmp != INT32_MIN && max_f < tmp + f[max_ii]) { max_f = tmp + f[max_ii]; max_j = max_ii; }
                            }
                            f[i] = max_f; p[i] = (int16_t)max_j;
                            if (max_ii < 0 || (ri - (int)(key[max_ii] >> 16) <= CH_MAX_GAP && f[max_ii] < f[i])) max_ii = i;
                        }
                        // backtrack, best end first ((f, index) descending); t: 0 free, 1 in a chain, 2 being walked; +4 tried as an end
                        for (int i = 0; i < n; ++i) t[i] = 0;
                        for (;;) {
                            int zi = -1;
                            for (int i = 0; i < n; ++i) if (t[i] == 0 && f[i] >= CH_MIN_SCORE && (zi < 0 || f[i] >= f[zi])) zi = i;
                            if (zi < 0) break;
                            const int zf = f[zi];
                            int i = zi, end_i = -1, max_i = zi, max_s = 0;
                            do {
                                t[i] = 2; end_i = i = p[i];
                                const int s = i < 0 ? zf : zf - f[i];
                                if (s > max_s) { max_s = s; max_i = i; }
                                else if (max_s - s > CH_BW) break;
                            } while (i >= 0 && (t[i] & 3) == 0);
                            for (i = zi; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
                            end_i = max_i;
                            int cnt = 0, first = zi;
                            for (i = zi; i != end_i; i = p[i]) { t[i] = 1; first = i; ++cnt; }
                            const int sc = i < 0 ? zf : zf - f[i];
                            if (cnt == 0) t[zi] = 4;                                     // (an end whose best cut is itself: tried, left free for other chains to run into)
                            if (sc >= CH_MIN_SCORE && cnt >= CH_MIN_CNT) {
                                const uint32_t at = atomicAdd(&s_u[0], 1u);
                                if (at < chain_cap) { SChain c; c.score = sc; c.f_end = zf; c.key = grp.key; c.end = key[zi]; c.first = key[first]; c.cnt = (uint32_t)cnt; sc_chain[at] = c; }
                            }
                        }
                    }
                    g0 += n_round;
                    __syncthreads();
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        // E. rank, parents, selection (set_parent + select_sub of oracle/mm2.c on the ranked list: score, then creation order = (f of the end anchor, anchor index) descending)
        uint32_t nc = s_u[0];
        if (nc > chain_cap) { nc = chain_cap; if (tid == 0) atomicAdd(&ctr->overflow_reads, 1u); }
        auto k1_of = [](const SChain& c) { return (unsigned long long)(uint32_t)c.score << 32 | (uint32_t)c.f_end; };
        auto k2_of = [](const SChain& c) { return (unsigned long long)c.key << 32 | c.end; };
        for (uint32_t i = tid; i < nc; i += CH_THREADS) sc_parent[i] = -1;
        __syncthreads();
        int n_prim = 0;
        while (n_prim < PRIM_CAP) {
            bool have = false; unsigned long long b1 = 0, b2 = 0; int bi = -1;
            for (uint32_t i = tid; i < nc; i += CH_THREADS) if (sc_parent[i] == -1) {
                const SChain c = sc_chain[i]; const unsigned long long a1 = k1_of(c), a2 = k2_of(c);
                if (!have || a1 > b1 || (a1 == b1 && a2 > b2)) { have = true; b1 = a1; b2 = a2; bi = (int)i; }
            }
            const int P = block_argmax(have, b1, b2, bi, s_r1, s_r2, s_ri);
            if (P < 0) break;
            if (tid == 0) { sc_parent[P] = P; s_prim[n_prim] = P; }
            ++n_prim;
            const SChain pc = sc_chain[P];
            int pqs, pqe, prs, pre; chain_interval(pc, qlen, pqs, pqe, prs, pre);
            __syncthreads();
            for (uint32_t i = tid; i < nc; i += CH_THREADS) if (sc_parent[i] == -1) {
                int qs, qe, rs, re; chain_interval(sc_chain[i], qlen, qs, qe, rs, re);
                const int mn = (pqe - pqs) < (qe - qs) ? (pqe - pqs) : (qe - qs);
                const int ol = (qe < pqe ? qe : pqe) - (qs > pqs ? qs : pqs);
                if (ol > 0 && (float)ol > MASK_LEVEL * (float)mn) sc_parent[i] = P;
            }
            __syncthreads();
        }
        // the best_n best-ranked secondaries within pri_ratio of their primary (or min_diff = 2 k of it) that are not its very interval
        int n_sec = 0;
        for (; n_sec < best_n; ++n_sec) {
            bool have = false; unsigned long long b1 = 0, b2 = 0; int bi = -1;
            for (uint32_t i = tid; i < nc; i += CH_THREADS) {
                const int p = sc_parent[i];
                if (p < 0 || p == (int)i) continue;
                const SChain c = sc_chain[i], pc = sc_chain[p];
                if (!((float)c.score >= (float)pc.score * PRI_RATIO || c.score + 2 * MZ_K >= pc.score)) continue;
                int qs, qe, rs, re, pqs, pqe, prs, pre; chain_interval(c, qlen, qs, qe, rs, re); chain_interval(pc, qlen, pqs, pqe, prs, pre);
                if (qs == pqs && qe == pqe && (c.key & 0x7FFFu) == (pc.key & 0x7FFFu) && rs == prs && re == pre) continue;
                const unsigned long long a1 = k1_of(c), a2 = k2_of(c);
                if (!have || a1 > b1 || (a1 == b1 && a2 > b2)) { have = true; b1 = a1; b2 = a2; bi = (int)i; }
            }
            const int S = block_argmax(have, b1, b2, bi, s_r1, s_r2, s_ri);
            if (S < 0) break;
            if (tid == 0) { sc_parent[S] = -2 - sc_parent[S]; s_sel[n_prim + n_sec] = S; }          // (taken: no longer a candidate; the parent stays readable)
            __syncthreads();
        }
        __syncthreads();
        if (dbg_chains && r == dbg_read) {
            for (uint32_t i = tid; i < nc; i += CH_THREADS) { dbg_chains[i] = sc_chain[i]; dbg_parent[i] = sc_parent[i]; }
            if (tid == 0) *dbg_n = nc;
        }
        if (tid == 0) {
            // the selected chains in rank order
            int n = 0;
            for (int i = 0; i < n_prim; ++i) s_sel[n++] = s_prim[i];
            for (int i = 0; i < n_sec; ++i) s_sel[n++] = s_sel[n_prim + i];
            for (int i = 1; i < n; ++i) {
                const int x = s_sel[i]; const SChain cx = sc_chain[x]; int j = i - 1;
                while (j >= 0) { const SChain cj = sc_chain[s_sel[j]]; if (k1_of(cj) > k1_of(cx) || (k1_of(cj) == k1_of(cx) && k2_of(cj) > k2_of(cx))) break; s_sel[j + 1] = s_sel[j]; --j; }
                s_sel[j + 1] = x;
            }
            if (n > SEL_CAP) n = SEL_CAP;
            uint32_t n_rev = 0;
            for (int i = 0; i < n; ++i) {
                const SChain c = sc_chain[s_sel[i]];
                SSel o; o.rid = (int32_t)(c.key & 0x7FFFu); o.rev = (int32_t)(c.key >> 15); o.score = c.score; o.cnt = (int32_t)c.cnt; o.n_chains = (int32_t)nc;
                chain_interval(c, qlen, o.qs, o.qe, o.rs, o.re);
                const int d0 = (int)(c.first & 0xFFFFu) - (int)(c.first >> 16), d1 = (int)(c.end & 0xFFFFu) - (int)(c.end >> 16);
                o.diag = (d0 + d1) >> 1;
                sel[(size_t)r * SEL_CAP + i] = o;
                n_rev += (uint32_t)o.rev;
            }
            sel_cnt[r] = (uint32_t)n;
            if (n_rev) atomicAdd(&ctr->rev_selected, n_rev);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// the selected chains as cells of the library's aligner: A = the allele (hg38 strand), B = the read (or its reverse complement), diagonal = midway between
// the chain's outermost seeds; slots a read does not use (and chains of the other strand) are marked "no cell"
__global__ void k1s_cells_kernel(const SSel* __restrict__ sel, const uint32_t* __restrict__ sel_cnt, uint32_t n_reads, const uint32_t* __restrict__ rid_allele,
                                 const int32_t* __restrict__ allele_len, int rev, CellDesc* __restrict__ cells) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_reads * SEL_CAP) return;
    const uint32_t r = x / SEL_CAP, s = x % SEL_CAP;
    CellDesc c; c.a = 0; c.b = r; c.diag = SP_NO_DIAG; c.max_ed = 0; c.b_lo = 0; c.b_hi = -1;
    if (s < sel_cnt[r]) {
        const SSel q = sel[x];
        if (q.rev == rev) {
            c.a = rid_allele[q.rid]; c.diag = q.diag;
            int cap = (int)(0.03 * (double)allele_len[c.a]) + 1; if (cap > SP_MAX_ED) cap = SP_MAX_ED;      // nm <= 0.03 * span <= 0.03 * allele length (realigner.rs:138-141)
            c.max_ed = cap;
        }
    }
    cells[x] = c;
}
// the cells of their re-score: on the diagonal the alignment lies on; a chain whose cell found nothing is not a mapping
__global__ void k1s_rescore_cells_kernel(const CellDesc* __restrict__ cells, const sp_aln* __restrict__ alns, uint32_t n, CellDesc* __restrict__ out) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    CellDesc c = cells[x];
    const sp_aln a = alns[x];
    if (c.diag == SP_NO_DIAG || !a.ok) { c.diag = 0; c.max_ed = -1; }
    else { c.diag = ((a.b_start - a.a_start) + (a.b_end - a.a_end)) / 2; c.max_ed = 127; }
    out[x] = c;
}
// merges the reverse-strand results into the slots of the forward arrays
__global__ void k1s_merge_rev_kernel(const SSel* __restrict__ sel, const uint32_t* __restrict__ sel_cnt, uint32_t n_reads, const sp_aln* __restrict__ aln_rev,
                                     const sp_affine_aln* __restrict__ af_rev, sp_aln* __restrict__ aln, sp_affine_aln* __restrict__ af) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_reads * SEL_CAP) return;
    const uint32_t r = x / SEL_CAP, s = x % SEL_CAP;
    if (s < sel_cnt[r] && sel[x].rev) { aln[x] = aln_rev[x]; af[x] = af_rev[x]; }
}
// reverse complement of the reads that have a selected chain on the reverse strand (same offsets and lengths as the read set)
__global__ __launch_bounds__(256) void k1s_revcomp_kernel(SeqSetView reads, const SSel* __restrict__ sel, const uint32_t* __restrict__ sel_cnt, uint32_t* __restrict__ words,
                                                          uint32_t* __restrict__ nplane) {
    const uint32_t r = blockIdx.x;
    bool need = false;
    for (uint32_t s = 0; s < sel_cnt[r]; ++s) need |= sel[(size_t)r * SEL_CAP + s].rev != 0;
    const int len = reads.len[r];
    const uint64_t wo = reads.word_off[r];
    const int nw = (len + 15) >> 4;
    if (!need) return;
    const uint32_t* w = reads.words + wo; const uint32_t* np = reads.nplane ? reads.nplane + wo : nullptr;
    for (int x = threadIdx.x; x < nw + 2; x += 256) {
        uint32_t out = 0, outn = 0;
        for (int b = 0; b < 16; ++b) {
            const int pos = x * 16 + b, src = len - 1 - pos;
            if (pos >= len) break;
            const uint32_t base = (w[src >> 4] >> ((src & 15) << 1)) & 3u;
            out |= (3u - base) << (b << 1);
            if (np) outn |= ((np[src >> 4] >> ((src & 15) << 1)) & 1u) << (b << 1);
        }
        if (np) { const uint32_t m = outn | (outn << 1); out &= ~m; }       // (an ambiguous base is stored as 00 in the words)
        words[wo + x] = out;
        if (nplane) nplane[wo + x] = outn;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// the mappings of a read in minimap2's output order and realign_record's acceptance loop over them (oracle/mm2.c omm_hla_k1_seeded; src/hla/realigner.rs:124-146)
__device__ __forceinline__ double seed_score_value(int len, int nm, int unmapped) { double num = (double)(nm + unmapped); if (num < 0.1) num = 0.1; return num / (double)len; }

__global__ void k1s_pick_kernel(SeqSetView reads, const SSel* __restrict__ sel, const uint32_t* __restrict__ sel_cnt, uint32_t n_reads, const sp_aln* __restrict__ alns,
                                const sp_affine_aln* __restrict__ afs, const uint32_t* __restrict__ rid_allele, const int32_t* __restrict__ allele_len, int best_n,
                                int32_t* __restrict__ best_out, sp_k1_seed_info* __restrict__ info, sp_aln* __restrict__ win_aln, sp_affine_aln* __restrict__ win_af,
                                sp_k1_seed_hit* __restrict__ dbg_hits, uint32_t* __restrict__ dbg_n, uint32_t dbg_read) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int qlen = reads.len[r];
    const uint32_t ns = sel_cnt[r];
    struct Hit { int slot, rid, rev, score, dp, nm, qs, qe, rs, re, tl, parent; };
    Hit h[SEL_CAP]; int nh = 0;
    for (uint32_t s = 0; s < ns; ++s) {
        const size_t x = (size_t)r * SEL_CAP + s;
        const SSel q = sel[x]; const sp_aln a = alns[x]; const sp_affine_aln f = afs[x];
        if (!a.ok || f.score < MIN_DP_MAX || f.score <= 0) continue;
        Hit t; t.slot = (int)s; t.rid = q.rid; t.rev = q.rev; t.score = q.score; t.dp = f.score; t.nm = f.nm; t.rs = f.b_start; t.re = f.b_end;
        if (q.rev) { t.qs = qlen - f.a_end; t.qe = qlen - f.a_start; } else { t.qs = f.a_start; t.qe = f.a_end; }
        t.tl = allele_len[rid_allele[q.rid]]; t.parent = 0;
        // by peak score, descending (stable)
        int j = nh - 1;
        while (j >= 0 && h[j].dp < t.dp) { h[j + 1] = h[j]; --j; }
        h[j + 1] = t; ++nh;
    }
    // parents on the aligned intervals, then the secondaries that stay
    for (int i = 0; i < nh; ++i) {
        h[i].parent = i;
        for (int j = 0; j < i; ++j) {
            if (h[j].parent != j) continue;
            const int mn = (h[j].qe - h[j].qs) < (h[i].qe - h[i].qs) ? (h[j].qe - h[j].qs) : (h[i].qe - h[i].qs);
            const int ol = (h[i].qe < h[j].qe ? h[i].qe : h[j].qe) - (h[i].qs > h[j].qs ? h[i].qs : h[j].qs);
            if (ol > 0 && (float)ol > MASK_LEVEL * (float)mn) { h[i].parent = j; break; }
        }
    }
    int pick = -1, n_out = 0, n2 = 0; double best = 1.0;
    for (int i = 0; i < nh; ++i) {
        const int p = h[i].parent;
        bool keep = p == i;
        if (!keep && ((float)h[i].score >= (float)h[p].score * PRI_RATIO || h[i].score + 2 * MZ_K >= h[p].score) && n2 < best_n &&
            !(h[i].qs == h[p].qs && h[i].qe == h[p].qe && h[i].rid == h[p].rid && h[i].rs == h[p].rs && h[i].re == h[p].re)) { keep = true; ++n2; }
        if (!keep) continue;
        const int tl = h[i].tl, um = tl - (h[i].re - h[i].rs), nm = h[i].nm;
        const double pen = seed_score_value(tl, nm, um), ed = seed_score_value(tl - um, nm, 0);
        if (pen <= 0.5 && ed <= 0.03 && ed < best) { best = ed; pick = i; }
        if (dbg_hits && r == dbg_read) {
            sp_k1_seed_hit d; const size_t x = (size_t)r * SEL_CAP + h[i].slot; const SSel q = sel[x]; const sp_aln a = alns[x];
            d.allele = (int32_t)rid_allele[h[i].rid]; d.rev = h[i].rev; d.chain_score = q.score; d.n_seeds = q.cnt; d.t_len = tl; d.sel_rank = h[i].slot; d.diag = q.diag;
            d.ok = a.ok; d.cell_nm = a.nm; d.a_start = a.a_start; d.a_end = a.a_end; d.b_start = a.b_start; d.b_end = a.b_end;
            d.dp_max = h[i].dp; d.nm = nm; d.t_start = h[i].rs; d.t_end = h[i].re; d.q_start = h[i].qs; d.q_end = h[i].qe; d.primary = p == i ? 1 : 0;
            dbg_hits[n_out] = d;
        }
        ++n_out;
    }
    if (dbg_hits && r == dbg_read) *dbg_n = (uint32_t)n_out;
    sp_k1_seed_info o; o.n_chains = ns ? sel[(size_t)r * SEL_CAP].n_chains : 0; o.n_selected = (int32_t)ns; o.n_mappings = n_out; o.pick = pick; o.chain_score = 0; o.rev = 0;
    sp_aln wa; wa.ok = 0; wa.nm = 0; wa.a_start = wa.a_end = wa.b_start = wa.b_end = wa.a_len = wa.b_len = 0;
    sp_affine_aln wf; wf.score = 0; wf.nm = 0; wf.a_start = wf.a_end = wf.b_start = wf.b_end = 0;
    int32_t b = -1;
    if (pick >= 0) {
        const size_t x = (size_t)r * SEL_CAP + h[pick].slot;
        o.chain_score = h[pick].score; o.rev = h[pick].rev;
        // a best mapping on the reverse strand drops the read (src/hla/realigner.rs:178-193)
        if (!h[pick].rev) { b = (int32_t)rid_allele[h[pick].rid]; wa = alns[x]; wf = afs[x]; }
    }
    best_out[r] = b; info[r] = o; win_aln[r] = wa; win_af[r] = wf;
}

} // namespace

// =========================================================================================================================================
// host side
// =========================================================================================================================================
struct K1Seed {
    uint32_t n_seqs = 0, n_keys = 0; uint64_t n_mz = 0; int32_t mid_occ = 0;
    uint32_t* d_rid_allele = nullptr;       // indexed sequence -> allele of the database (the alleles with a DNA sequence, in database order)
    uint64_t* d_keys = nullptr; uint32_t* d_start = nullptr; uint32_t* d_occ = nullptr; uint32_t* d_bucket = nullptr; int32_t* d_pen = nullptr;
    IndexView view() const { return IndexView{ d_keys, d_start, d_occ, d_bucket, n_keys, mid_occ, n_seqs }; }
};

void sp_k1_seed_free(K1Seed* s) {
    if (!s) return;
    (void)hipFree(s->d_rid_allele); (void)hipFree(s->d_keys); (void)hipFree(s->d_start); (void)hipFree(s->d_occ); (void)hipFree(s->d_bucket); (void)hipFree(s->d_pen);
    delete s;
}

void sp_k1_seed_stats(const K1Seed* s, int64_t out[4]) { out[0] = (int64_t)s->n_mz; out[1] = s->n_keys; out[2] = s->mid_occ; out[3] = s->n_seqs; }

template <typename T> static T* seed_dev_copy(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(1, v.size()) * sizeof(T)) != hipSuccess) return nullptr;
    if (!v.empty()) (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

// the minimizer index of the alleles with a DNA sequence (aligner.with_index of HlaRealigner::new, src/hla/realigner.rs:56-60): sketch and sort on the device,
// the distinct keys and the occurrence threshold (oracle/mm2.c omm_index_build) on the host
int sp_k1_seed_build(sp_ctx* ctx, const sp_seqset* alleles, K1Seed** out) {
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    std::vector<uint32_t> rid_allele;
    for (uint32_t a = 0; a < alleles->n; ++a) if (alleles->h_len[a] > 0) rid_allele.push_back(a);
    const uint32_t n = (uint32_t)rid_allele.size();
    if (n == 0) return sp_fail(ctx, SP_ERR_INVALID_ARG, "seeded K1: the database has no DNA allele");
    if (n >= (1u << 15) || alleles->max_len > 65535) return sp_fail(ctx, SP_ERR_TOO_LONG, "seeded K1: at most 32,767 DNA alleles of at most 65,535 bases");
    K1Seed* s = new (std::nothrow) K1Seed();
    if (!s) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: index");
    s->n_seqs = n;
    std::vector<void*> temps;
    auto grab = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; temps.push_back(q); return q; };
    auto fail = [&](const char* what) { for (void* q : temps) (void)hipFree(q); sp_k1_seed_free(s); return sp_fail(ctx, SP_ERR_HIP, what); };
    hipStream_t st = ctx->stream;
    s->d_rid_allele = seed_dev_copy(rid_allele);
    uint32_t* d_cnt = (uint32_t*)grab((size_t)n * 4); uint64_t* d_off = (uint64_t*)grab(((size_t)n + 1) * 8);
    if (!s->d_rid_allele || !d_cnt || !d_off) return fail("seeded K1: index buffers");
    hipLaunchKernelGGL(mz_sketch_kernel<false>, dim3(n), dim3(256), 0, st, alleles->view(), s->d_rid_allele, n, d_cnt, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr);
    std::vector<uint32_t> cnt(n);
    if (hipMemcpyAsync(cnt.data(), d_cnt, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("seeded K1: sketch counts");
    std::vector<uint64_t> off((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) off[i + 1] = off[i] + cnt[i];
    const uint64_t total = off[n];
    if (total == 0 || total >= (1ull << 32)) return fail("seeded K1: nothing to index");
    s->n_mz = total;
    uint64_t* d_k = (uint64_t*)grab(total * 8); uint64_t* d_ks = (uint64_t*)grab(total * 8); uint32_t* d_o = (uint32_t*)grab(total * 4);
    if (!d_k || !d_ks || !d_o || hipMalloc(&s->d_occ, total * 4) != hipSuccess) return fail("seeded K1: index tables");
    if (hipMemcpyAsync(d_off, off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail("seeded K1: offsets");
    hipLaunchKernelGGL(mz_sketch_kernel<true>, dim3(n), dim3(256), 0, st, alleles->view(), s->d_rid_allele, n, (uint32_t*)nullptr, d_off, d_k, d_o);
    // stable sort by hash: the occurrences of a hash stay in (sequence, position) order
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, d_k, d_ks, d_o, s->d_occ, (size_t)total, 0, 2 * MZ_K, st) != hipSuccess) return fail("seeded K1: sort size");
    void* ws = grab(bytes);
    if (!ws || rocprim::radix_sort_pairs(ws, bytes, d_k, d_ks, d_o, s->d_occ, (size_t)total, 0, 2 * MZ_K, st) != hipSuccess) return fail("seeded K1: sort");
    std::vector<uint64_t> ks(total);
    if (hipMemcpyAsync(ks.data(), d_ks, total * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("seeded K1: sorted keys");
    std::vector<uint64_t> keys; std::vector<uint32_t> start;
    for (uint64_t i = 0; i < total; ++i) if (i == 0 || ks[i] != ks[i - 1]) { keys.push_back(ks[i]); start.push_back((uint32_t)i); }
    start.push_back((uint32_t)total);
    const size_t nk = keys.size();
    s->n_keys = (uint32_t)nk;
    // occurrence threshold: the count at the top mid_occ_frac of the distinct minimizers, + 1, clamped (omm_index_build)
    {
        std::vector<uint32_t> c(nk);
        for (size_t i = 0; i < nk; ++i) c[i] = start[i + 1] - start[i];
        std::sort(c.begin(), c.end());
        int64_t kth = (int64_t)((1.0 - (double)MID_OCC_FRAC) * (double)nk);
        if (kth >= (int64_t)nk) kth = (int64_t)nk - 1;
        int32_t mid = (int32_t)c[(size_t)kth] + 1;
        if (mid < MIN_MID_OCC) mid = MIN_MID_OCC;
        if (mid > MAX_MID_OCC) mid = MAX_MID_OCC;
        s->mid_occ = mid;
    }
    std::vector<uint32_t> bucket(((size_t)1 << BUCKET_BITS) + 1, 0);
    {
        size_t k = 0;
        for (size_t b = 0; b <= ((size_t)1 << BUCKET_BITS); ++b) {
            while (k < nk && (keys[k] >> (2 * MZ_K - BUCKET_BITS)) < b) ++k;
            bucket[b] = (uint32_t)k;
        }
    }
    // the gap penalty of the chaining score by |dr - dq| (chain_sc): the very float expression of the statement, evaluated once on the host
    std::vector<int32_t> pen(CH_BW + 1, 0);
    {
        const float pen_gap = 0.01f * 0.8f * (float)MZ_K;
        for (int dd = 1; dd <= CH_BW; ++dd) { const float lin = pen_gap * (float)dd; const float lg = log2f((float)(dd + 1)); pen[dd] = (int32_t)(lin + .5f * lg); }
    }
    s->d_keys = seed_dev_copy(keys); s->d_start = seed_dev_copy(start); s->d_bucket = seed_dev_copy(bucket); s->d_pen = seed_dev_copy(pen);
    for (void* q : temps) (void)hipFree(q);
    temps.clear();
    if (!s->d_keys || !s->d_start || !s->d_bucket || !s->d_pen) { sp_k1_seed_free(s); return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: index upload"); }
    *out = s;
    return SP_OK;
}

// minimizers of sequence `idx` of a set (audit: tests compare them with oracle/mm2.c omm_sketch)
int sp_k1_seed_sketch(sp_ctx* ctx, const sp_seqset* set, uint32_t idx, uint64_t* hash, int32_t* end_pos, uint8_t* strand, uint32_t cap, uint32_t* n_out) {
    (void)hipSetDevice(ctx->device);
    const uint32_t maxn = (uint32_t)std::max(1, set->h_len[idx]);
    uint32_t* d_seq = (uint32_t*)sp_pool(ctx, "k1s_sk_seq", 4); uint32_t* d_cnt = (uint32_t*)sp_pool(ctx, "k1s_sk_cnt", 4); uint64_t* d_off = (uint64_t*)sp_pool(ctx, "k1s_sk_off", 8);
    uint64_t* d_k = (uint64_t*)sp_pool(ctx, "k1s_sk_k", (size_t)maxn * 8); uint32_t* d_o = (uint32_t*)sp_pool(ctx, "k1s_sk_o", (size_t)maxn * 4);
    if (!d_seq || !d_cnt || !d_off || !d_k || !d_o) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "sketch buffers");
    const uint64_t zero = 0;
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_seq, &idx, 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP_CHECK(ctx, hipMemcpyAsync(d_off, &zero, 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(mz_sketch_kernel<false>, dim3(1), dim3(256), 0, ctx->stream, set->view(), d_seq, 1u, d_cnt, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL(mz_sketch_kernel<true>, dim3(1), dim3(256), 0, ctx->stream, set->view(), d_seq, 1u, (uint32_t*)nullptr, d_off, d_k, d_o);
    uint32_t n = 0;
    SP_HIP_CHECK(ctx, hipMemcpyAsync(&n, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    const uint32_t m = std::min(n, cap);
    std::vector<uint64_t> k(m); std::vector<uint32_t> o(m);
    if (m) { SP_HIP_CHECK(ctx, hipMemcpy(k.data(), d_k, (size_t)m * 8, hipMemcpyDeviceToHost)); SP_HIP_CHECK(ctx, hipMemcpy(o.data(), d_o, (size_t)m * 4, hipMemcpyDeviceToHost)); }
    for (uint32_t i = 0; i < m; ++i) { if (hash) hash[i] = k[i]; if (end_pos) end_pos[i] = (int32_t)((o[i] >> 1) & 0xFFFFu); if (strand) strand[i] = (uint8_t)(o[i] & 1u); }
    return SP_OK;
}

// The seeded map of a batch of reads.  On return (stream-ordered, nothing copied back): d_best[r] = accepted allele or -1, d_info[r], d_win_aln[r] / d_win_af[r] = the
// accepted mapping's cell and its re-score.  dbg (optional): the chain list, selection and mappings of one read.
int sp_k1_seed_map(sp_ctx* ctx, const K1Seed* idx, const sp_seqset* alleles, const sp_seqset* reads, int best_n, int32_t* d_best, sp_k1_seed_info* d_info,
                   sp_aln* d_win_aln, sp_affine_aln* d_win_af, const K1SeedDebug* dbg) {
    const uint32_t R = reads->n;
    if (R == 0) return SP_OK;
    (void)hipSetDevice(ctx->device);
    if (reads->max_len > 65535) return sp_fail(ctx, SP_ERR_TOO_LONG, "seeded K1: reads of up to 65,535 bases");
    const IndexView ix = idx->view();
    SeedCounters* d_ctr = (SeedCounters*)sp_pool(ctx, "k1s_ctr", sizeof(SeedCounters));
    uint2* d_read_seed = (uint2*)sp_pool(ctx, "k1s_read_seed", (size_t)R * 8);
    uint32_t* d_read_anchors = (uint32_t*)sp_pool(ctx, "k1s_read_anchors", (size_t)R * 4);
    if (!d_ctr || !d_read_seed || !d_read_anchors) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: buffers");
    // 1. seeds.  A read has about len / 10 minimizers; the LDS list takes len / 6 + 64 of those found in the index, the global list grows when a batch needs more
    const int sd_cap = std::min(13000, reads->max_len / 6 + 64);
    const size_t seed_lds = (size_t)(2 * (TILE + 2 * HALO)) * 4 + (size_t)sd_cap * (4 + 4 + 2 + 1) + (TILE + 2 * HALO) + 64;
    SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k1s_seed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)seed_lds));
    SeedCounters hc;
    uint32_t seed_cap = (uint32_t)std::min<size_t>(0xFFFFFFF0u, (size_t)R * 96 + 4096);
    uint4* d_seeds = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        d_seeds = (uint4*)sp_pool(ctx, "k1s_seeds", (size_t)seed_cap * 16);
        if (!d_seeds) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: seed list");
        SP_HIP_CHECK(ctx, hipMemsetAsync(d_ctr, 0, sizeof(SeedCounters), ctx->stream));
        {
            ProfScope ps(ctx, "k1s_seeds", R);
            hipLaunchKernelGGL(k1s_seed_kernel, dim3(R), dim3(256), seed_lds, ctx->stream, reads->view(), ix, R, sd_cap, d_seeds, seed_cap, d_read_seed, d_read_anchors, d_ctr);
        }
        SP_HIP_CHECK(ctx, hipGetLastError());
        SP_HIP_CHECK(ctx, hipMemcpyAsync(&hc, d_ctr, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (hc.seed_overflow == 0) break;
        if (attempt == 1 || hc.seeds > 0xFFFFFFF0ull) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: seed list overflow");
        seed_cap = (uint32_t)hc.seeds + 1024;
    }
    // 2. chains: a grid of resident workgroups, each with a scratch region sized for the batch's largest read
    uint32_t* d_sel_cnt = (uint32_t*)sp_pool(ctx, "k1s_sel_cnt", (size_t)R * 4);
    SSel* d_sel = (SSel*)sp_pool(ctx, "k1s_sel", (size_t)R * SEL_CAP * sizeof(SSel));
    if (!d_sel_cnt || !d_sel) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: selection");
    const uint32_t anchor_cap = std::max<uint32_t>(hc.max_anchors, 16), group_cap = std::min<uint32_t>(anchor_cap / CH_MIN_CNT + 1, 2 * idx->n_seqs), chain_cap = anchor_cap / CH_MIN_CNT + 1;
    const size_t stride = (((size_t)anchor_cap * 4 + (size_t)group_cap * (sizeof(SGroup) + 4) + (size_t)chain_cap * (sizeof(SChain) + 4)) + 255) & ~(size_t)255;
    const uint32_t grid = std::min<uint32_t>(R, (uint32_t)ctx->num_cus * 3);
    uint8_t* d_scratch = (uint8_t*)sp_pool(ctx, "k1s_scratch", stride * grid);
    if (!d_scratch) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: chain scratch");
    constexpr int UNION_WORDS = WIN_KEYS > 3 * DP_CAP ? WIN_KEYS : 3 * DP_CAP;
    const size_t chain_lds = (size_t)(UNION_WORDS + 512) * 4;
    SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k1s_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
    SChain* dbg_chains = nullptr; int32_t* dbg_parent = nullptr; uint32_t* dbg_n = nullptr;
    if (dbg) {
        dbg_chains = (SChain*)sp_pool(ctx, "k1s_dbg_chains", (size_t)chain_cap * sizeof(SChain)); dbg_parent = (int32_t*)sp_pool(ctx, "k1s_dbg_parent", (size_t)chain_cap * 4);
        dbg_n = (uint32_t*)sp_pool(ctx, "k1s_dbg_n", 8);
        if (!dbg_chains || !dbg_parent || !dbg_n) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: debug buffers");
        SP_HIP_CHECK(ctx, hipMemsetAsync(dbg_n, 0, 8, ctx->stream));
    }
    {
        ProfScope ps(ctx, "k1s_chains", hc.anchors);
        hipLaunchKernelGGL(k1s_chain_kernel, dim3(grid), dim3(CH_THREADS), chain_lds, ctx->stream, reads->view(), ix, R, d_seeds, d_read_seed, idx->d_pen, d_scratch, stride, anchor_cap,
                           group_cap, chain_cap, d_sel, d_sel_cnt, d_ctr, best_n, dbg_chains, dbg_parent, dbg_n, dbg ? dbg->read : 0u);
    }
    SP_HIP_CHECK(ctx, hipGetLastError());
    // 3. the selected chains through the cell and its re-score
    const uint64_t NC = (uint64_t)R * SEL_CAP;
    CellDesc* d_cells = (CellDesc*)sp_pool(ctx, "k1s_cells", NC * sizeof(CellDesc)); CellDesc* d_rc = (CellDesc*)sp_pool(ctx, "k1s_rc", NC * sizeof(CellDesc));
    sp_aln* d_aln = (sp_aln*)sp_pool(ctx, "k1s_aln", NC * sizeof(sp_aln)); sp_affine_aln* d_af = (sp_affine_aln*)sp_pool(ctx, "k1s_af_out", NC * sizeof(sp_affine_aln));
    if (!d_cells || !d_rc || !d_aln || !d_af) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: cells");
    const sp_affine_opts ao = { 1, 4, 6, 2, 26, 1, 1 };
    const unsigned nb = (unsigned)((NC + 255) / 256);
    hipLaunchKernelGGL(k1s_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_sel, d_sel_cnt, R, idx->d_rid_allele, alleles->d_len, 0, d_cells);
    int rc = sp_launch_cells(ctx, alleles, reads, d_cells, NC, d_aln, nullptr, 0, "k1s_cells", 1);
    if (rc != SP_OK) return rc;
    hipLaunchKernelGGL(k1s_rescore_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_cells, d_aln, (uint32_t)NC, d_rc);
    rc = sp_rescore_mappings(ctx, alleles, reads, d_rc, d_aln, NC, true, ao, 64, d_af, "k1s_af", 128);
    if (rc != SP_OK) return rc;
    // chains on the reverse strand (a read from the other strand, the homologous gene on the other strand): the same through the reads' reverse complements
    SP_HIP_CHECK(ctx, hipMemcpyAsync(&hc, d_ctr, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (hc.rev_selected) {
        const size_t words = (size_t)reads->h_word_off[R] + SP_SEQ_PAD_WORDS;
        uint32_t* d_rw = (uint32_t*)sp_pool(ctx, "k1s_rev_words", words * 4);
        uint32_t* d_rn = reads->d_nplane ? (uint32_t*)sp_pool(ctx, "k1s_rev_nplane", words * 4) : nullptr;
        CellDesc* d_cells2 = (CellDesc*)sp_pool(ctx, "k1s_cells_rev", NC * sizeof(CellDesc));
        sp_aln* d_aln2 = (sp_aln*)sp_pool(ctx, "k1s_aln_rev", NC * sizeof(sp_aln)); sp_affine_aln* d_af2 = (sp_affine_aln*)sp_pool(ctx, "k1s_af_rev", NC * sizeof(sp_affine_aln));
        if (!d_rw || (reads->d_nplane && !d_rn) || !d_cells2 || !d_aln2 || !d_af2) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: reverse strand");
        SP_HIP_CHECK(ctx, hipMemsetAsync(d_rw, 0, words * 4, ctx->stream));
        if (d_rn) SP_HIP_CHECK(ctx, hipMemsetAsync(d_rn, 0, words * 4, ctx->stream));
        hipLaunchKernelGGL(k1s_revcomp_kernel, dim3(R), dim3(256), 0, ctx->stream, reads->view(), d_sel, d_sel_cnt, d_rw, d_rn);
        sp_seqset rset;
        rset.ctx = reads->ctx; rset.n = R; rset.has_n = reads->has_n; rset.max_len = reads->max_len; rset.d_words = d_rw; rset.d_nplane = d_rn; rset.d_word_off = reads->d_word_off; rset.d_len = reads->d_len;
        rset.h_len = reads->h_len; rset.h_word_off = reads->h_word_off;
        hipLaunchKernelGGL(k1s_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_sel, d_sel_cnt, R, idx->d_rid_allele, alleles->d_len, 1, d_cells2);
        rc = sp_launch_cells(ctx, alleles, &rset, d_cells2, NC, d_aln2, nullptr, 0, "k1s_cells_rev", 1);
        if (rc != SP_OK) return rc;
        hipLaunchKernelGGL(k1s_rescore_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_cells2, d_aln2, (uint32_t)NC, d_rc);
        rc = sp_rescore_mappings(ctx, alleles, &rset, d_rc, d_aln2, NC, true, ao, 64, d_af2, "k1s_af_rev", 128);
        if (rc != SP_OK) return rc;
        hipLaunchKernelGGL(k1s_merge_rev_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_sel, d_sel_cnt, R, d_aln2, d_af2, d_aln, d_af);
        SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));           // (rset's vectors go out of scope: nothing of it may still be queued)
    }
    // 4. output order, second selection, acceptance
    sp_k1_seed_hit* dbg_hits = nullptr; uint32_t* dbg_nh = nullptr;
    if (dbg) {
        dbg_hits = (sp_k1_seed_hit*)sp_pool(ctx, "k1s_dbg_hits", SEL_CAP * sizeof(sp_k1_seed_hit));
        if (!dbg_hits) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: debug buffers");
        dbg_nh = dbg_n + 1;
    }
    hipLaunchKernelGGL(k1s_pick_kernel, dim3((R + 63) / 64), dim3(64), 0, ctx->stream, reads->view(), d_sel, d_sel_cnt, R, d_aln, d_af, idx->d_rid_allele, alleles->d_len, best_n,
                       d_best, d_info, d_win_aln, d_win_af, dbg_hits, dbg_nh, dbg ? dbg->read : 0u);
    SP_HIP_CHECK(ctx, hipGetLastError());
    if (dbg) {
        uint32_t n2[2] = { 0, 0 };
        SP_HIP_CHECK(ctx, hipMemcpyAsync(n2, dbg_n, 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<SChain> ch(n2[0]); std::vector<int32_t> par(n2[0]);
        if (n2[0]) { SP_HIP_CHECK(ctx, hipMemcpy(ch.data(), dbg_chains, (size_t)n2[0] * sizeof(SChain), hipMemcpyDeviceToHost)); SP_HIP_CHECK(ctx, hipMemcpy(par.data(), dbg_parent, (size_t)n2[0] * 4, hipMemcpyDeviceToHost)); }
        // the chains in rank order, as omm_chain_stage lists them
        std::vector<uint32_t> order(n2[0]);
        for (uint32_t i = 0; i < n2[0]; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const unsigned long long x1 = (unsigned long long)(uint32_t)ch[x].score << 32 | (uint32_t)ch[x].f_end, y1 = (unsigned long long)(uint32_t)ch[y].score << 32 | (uint32_t)ch[y].f_end;
            if (x1 != y1) return x1 > y1;
            const unsigned long long x2 = (unsigned long long)ch[x].key << 32 | ch[x].end, y2 = (unsigned long long)ch[y].key << 32 | ch[y].end;
            return x2 > y2; });
        const int qlen = reads->h_len[dbg->read];
        *dbg->n_chains = n2[0];
        for (uint32_t i = 0; i < n2[0] && i < dbg->chain_cap; ++i) {
            const SChain& c = ch[order[i]]; int32_t* o = dbg->chains + (size_t)i * 10;
            int rs = (int)(c.first >> 16) + 1 - MZ_K, re = (int)(c.end >> 16) + 1, s = (int)(c.first & 0xFFFFu) + 1 - MZ_K, e = (int)(c.end & 0xFFFFu) + 1;
            const int rev = (int)(c.key >> 15);
            o[0] = (int32_t)(c.key & 0x7FFFu); o[1] = rev; o[2] = c.score; o[3] = (int32_t)c.cnt; o[4] = rev ? qlen - e : s; o[5] = rev ? qlen - s : e; o[6] = rs; o[7] = re;
            const int32_t p = par[order[i]];
            o[8] = p; o[9] = (p == (int32_t)order[i] || p <= -2) ? 1 : 0;             // selected: a primary, or a secondary the selection took
        }
        *dbg->n_hits = n2[1];
        if (n2[1]) SP_HIP_CHECK(ctx, hipMemcpy(dbg->hits, dbg_hits, (size_t)std::min<uint32_t>(n2[1], SEL_CAP) * sizeof(sp_k1_seed_hit), hipMemcpyDeviceToHost));
        dbg->counters[0] = hc.seeds; dbg->counters[1] = hc.anchors; dbg->counters[2] = hc.max_anchors; dbg->counters[3] = hc.overflow_reads;
    }
    return SP_OK;
}
