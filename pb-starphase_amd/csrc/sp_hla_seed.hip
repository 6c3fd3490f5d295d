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
#include <cstdio>
#include "sp_internal.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>

#ifndef SP_DP_EXP
#define SP_DP_EXP 0
#endif
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
// is_minimizer for every position of a tile (all threads of the 256-thread workgroup call it, h[] complete and synchronised): mzf[j] = 1 when the k-mer at slot HALO + j is one.
// The scan of a true minimizer reads all 36 neighbours and the lanes of its wave wait for it (two thirds of the seeds kernel's time): a first pass looks MZ_QUICK neighbours
// either way with every lane in step -- a k-mer that meets a smaller one on both sides within that reach is decided there (l + r < w - 1; the stretch rule needs both ends inside
// the reach too) --, the others (a third of the positions) go on a list and the full scans run with the waves packed with them.
#ifndef SP_MZ_QUICK
#define SP_MZ_QUICK 5
#endif
constexpr int MZ_QUICK = SP_MZ_QUICK;
__device__ __forceinline__ void tile_minimizers(const uint64_t* __restrict__ h, int lo, int len, uint8_t* __restrict__ mzf, uint16_t* __restrict__ cand, uint32_t* __restrict__ n_cand) {
    if (threadIdx.x == 0) *n_cand = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < TILE; j += 256) {
        const int c = HALO + j;
        const uint64_t hp = h[c];
        uint8_t f = 0;
        if (lo + j < len && hp < H_PAL) {
            int l = MZ_QUICK, r = MZ_QUICK; bool lb = false, rb = false, tie = false;
#pragma unroll
            for (int x = MZ_QUICK - 1; x >= 0; --x) {                      // (nearest stop wins: walk inwards)
                const uint64_t vl = h[c - 1 - x], vr = h[c + 1 + x];
                if (vl == H_NONE) { l = x; lb = true; } else if (vl < hp) { l = x; lb = false; }
                if (vr == H_NONE) { r = x; rb = true; } else if (vr < hp) { r = x; rb = false; }
            }
            if (l < MZ_QUICK && r < MZ_QUICK) {
                // both scans end inside the reach: l + r < w - 1, so only a stretch shorter than a window can make it one (its last smallest k-mer: no tie to the right)
                if (lb && rb) { for (int x = 0; x < r; ++x) tie |= h[c + 1 + x] == hp; f = tie ? 0 : 1; }
            } else cand[atomicAdd(n_cand, 1u)] = (uint16_t)c;
        }
        mzf[j] = f;
    }
    __syncthreads();
    const uint32_t nc = *n_cand;
    for (uint32_t i = threadIdx.x; i < nc; i += 256) { const int c = cand[i]; mzf[c - HALO] = is_minimizer(h, c) ? 1 : 0; }
    __syncthreads();
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
    __shared__ uint8_t mzf[TILE];
    __shared__ uint16_t cand[TILE];
    __shared__ uint32_t wave_tot[4], n_cand;
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
        tile_minimizers(h, lo, len, mzf, cand, &n_cand);
        for (int j = 0; j < TILE; j += 256) {
            const int c = HALO + j + (int)threadIdx.x, pos = lo + j + (int)threadIdx.x;
            const bool flag = pos < len && mzf[j + (int)threadIdx.x] != 0;
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
struct SeedCounters { unsigned long long seeds, anchors; uint32_t max_anchors, max_seeds, overflow_reads, seed_overflow, rev_selected, wide_cells; unsigned long long t[8]; };
#ifdef SP_K1S_TIMING
#define K1S_T(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&ctr->t[i], now_ - t_last); t_last = now_; } } while (0)
#else
#define K1S_T(i) do { } while (0)
#endif
#ifdef SP_K1S_TIMING
#define K1S_TW(i) do { if (lane == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&ctr->t[i], now_ - t_last); t_last = now_; } } while (0)
#else
#define K1S_TW(i) do { } while (0)
#endif

__global__ __launch_bounds__(256) void k1s_seed_kernel(SeqSetView reads, IndexView ix, uint32_t n_reads, int sd_cap, uint4* __restrict__ seeds, uint32_t seed_cap,
                                                       uint2* __restrict__ read_seed, uint32_t* __restrict__ read_anchors, SeedCounters* __restrict__ ctr) {
    extern __shared__ uint32_t lds[];
    uint64_t* h = reinterpret_cast<uint64_t*>(lds);                      // TILE + 2 HALO
    uint32_t* sd_q = lds + 2 * (TILE + 2 * HALO);                        // sd_cap each: query end position << 1 | strand, first occurrence
    uint32_t* sd_st = sd_q + sd_cap;
    uint16_t* sd_n = reinterpret_cast<uint16_t*>(sd_st + sd_cap);       // occurrences, capped at 65535 (only values <= 4095 are ever told apart, see below)
    uint8_t* zs = reinterpret_cast<uint8_t*>(sd_n + sd_cap);            // TILE + 2 HALO
    uint8_t* keep = zs + TILE + 2 * HALO + 8;                            // sd_cap
    __shared__ uint32_t wave_tot[4], s_sum[4], n_cand;
    __shared__ unsigned long long s_base;
    __shared__ uint8_t mzf[TILE];
    __shared__ uint16_t cand[TILE];
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
        tile_minimizers(h, lo, len, mzf, cand, &n_cand);
        for (int j = 0; j < TILE; j += 256) {
            const int c = HALO + j + (int)threadIdx.x, pos = lo + j + (int)threadIdx.x;
            bool flag = pos < len && mzf[j + (int)threadIdx.x] != 0;
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
// chains of the reads of a batch, in three kernels:
//   k1s_group_kernel   one workgroup per read: per window of targets (strand, [rid_lo, rid_lo + W)) the anchors of every target are counted in an LDS table, the targets
//                      with >= 3 anchors get a stretch of the batch's anchor array (anchor = target end position << 16 | query end position on the mapped strand:
//                      ascending = the order of oracle/mm2.c an_cmp) and a record in the batch's group list, largest first
//   k1s_dp_kernel      the chaining DP of chain_anchors, one THREAD per target, over the whole batch's group list: a wave's 64 targets are neighbours in a read's
//                      size order, so its lanes run DPs of like length, and nothing waits for a read's largest target
//   k1s_select_kernel  one workgroup per read: its chains ranked, primaries / secondaries (set_parent), the best_n secondaries within pri_ratio (select_sub)
struct SChain { int32_t score, f_end; uint32_t key, end, first, cnt; };       // key = rev << 15 | rid; end / first = target end position << 16 | query end position (strand coordinates)
struct SGroup { uint32_t read, key, off, cnt; };                              // off: first anchor in the batch's anchor array
struct SSel { int32_t rid, rev, score, cnt, diag, qs, qe, rs, re, n_chains; };        // qs / qe in forward coordinates
struct SDp { uint32_t key; int32_t f; int16_t p; uint16_t t; };               // one anchor of a DP: 12 bytes of LDS

constexpr int CH_THREADS = 256;
constexpr int DP_CAP = 3584;                    // anchors a DP round of a workgroup holds in LDS
constexpr int DP_CHUNK = 1024;                  // targets a workgroup takes from the list at a time
constexpr int WIN_KEYS = 12288;                 // targets per window (4 bytes each in LDS)

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

// inclusive prefix sum over the block in thread order; total = the block's sum.  All threads call it.
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t* wave_tot, uint32_t& total) {
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((int)(threadIdx.x & 63) >= o) incl += t; }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int wv = 0; wv < CH_THREADS / 64; ++wv) { const uint32_t c = wave_tot[wv]; if (wv < (int)(threadIdx.x >> 6)) base += c; tot += c; }
    total = tot;
    return base + incl;
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

struct BatchCursors { unsigned long long groups_anchors; uint32_t chains, dp_chunk, big_items, big_members, big_chunk, pad; };        // groups << 32 | anchors: one reservation keeps both lists in step
struct SBigItem { SGroup g; uint32_t member_off, n_members, pad0, pad1; };      // a large target that is chained by a wave of its own, and the targets that take its chains (itself among them)
struct SMember { uint32_t read, key, shift, pad; };

__global__ __launch_bounds__(CH_THREADS) void k1s_group_kernel(SeqSetView reads, IndexView ix, uint32_t r0, uint32_t n_batch, const uint4* __restrict__ seeds, const uint2* __restrict__ read_seed,
                                                               uint32_t* __restrict__ anchors, uint32_t anchor_cap, SGroup* __restrict__ groups, uint32_t group_cap,
                                                               uint3* __restrict__ read_chain, BatchCursors* __restrict__ cur, SeedCounters* __restrict__ ctr) {
    extern __shared__ uint32_t tab[];                   // WIN_KEYS
    __shared__ uint32_t s_hist[256], s_hista[256], s_first[256];
    __shared__ uint32_t s_scan[CH_THREADS / 64];
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    const uint32_t r = r0 + blockIdx.x;
    if (blockIdx.x >= n_batch) return;
    const uint2 rs_ = read_seed[r];
    const uint32_t s0 = rs_.x, K = rs_.y;
    const int qlen = reads.len[r];
    uint32_t n_groups_read = 0, n_anchors_read = 0;
    unsigned long long t_last = 0;
#ifdef SP_K1S_TIMING
    t_last = wall_clock64();
#endif
    (void)t_last;
    bool any_rev = false;                     // does the read have an anchor on the reverse strand at all?  (mostly not: those windows are skipped)
    for (int rev = 0; rev < 2 && K; ++rev) {
        if (rev == 1 && !any_rev) break;
        for (uint32_t rid_lo = 0; rid_lo < ix.n_seqs; rid_lo += WIN_KEYS) {
            const uint32_t rid_hi = rid_lo + WIN_KEYS < ix.n_seqs ? rid_lo + WIN_KEYS : ix.n_seqs, W = rid_hi - rid_lo;
            __syncthreads();
            for (uint32_t i = tid; i < W; i += CH_THREADS) tab[i] = 0;
            __syncthreads();
            // A. anchors per target; `other` = anchors of this target range on the other strand
            uint32_t other = 0;
            for (uint32_t s = 0; s < K; ++s) {
                const uint4 sd = seeds[s0 + s];
                const uint32_t strand = sd.x & 1u;
                for (uint32_t k = tid; k < sd.z; k += CH_THREADS) {
                    const uint32_t o = ix.occ[sd.y + k], rid = o >> 17;
                    if (rid < rid_lo || rid >= rid_hi) continue;
                    if ((int)((o & 1u) != strand) == rev) atomicAdd(&tab[rid - rid_lo], 1u); else ++other;
                }
            }
            if (rev == 0) any_rev |= __syncthreads_or(other != 0) != 0; else __syncthreads();
            K1S_T(0);
            // B. the targets with >= min_cnt anchors get their stretch of the anchor array and their group record, LARGEST FIRST: the targets of one size lie side by side
            // (their anchors too), so the 64 targets of a DP wave run loops of like length and read one compact piece of the anchor array
            const uint32_t per = (W + CH_THREADS - 1) / CH_THREADS, lo = (uint32_t)tid * per, hi = lo + per < W ? lo + per : W;
            s_hist[tid] = 0; s_hista[tid] = 0;
            __syncthreads();
            uint32_t my_g = 0;
            for (uint32_t i = lo; i < hi; ++i) { const uint32_t c = tab[i]; if (c >= (uint32_t)CH_MIN_CNT) { ++my_g; const uint32_t k = 255 - (c > 255 ? 255 : c); atomicAdd(&s_hist[k], 1u); atomicAdd(&s_hista[k], c); } }
            __syncthreads();
            uint32_t tot_a, tot_g;
            {
                const uint32_t vg = s_hist[tid], va = s_hista[tid];
                const uint32_t ig = block_scan(vg, s_scan, tot_g), ia = block_scan(va, s_scan, tot_a);
                __syncthreads();
                s_hist[tid] = ig - vg; s_hista[tid] = ia - va; s_first[tid] = ig - vg;          // first group / first anchor of every size class
                __syncthreads();
            }
            (void)my_g;
            if (tot_g == 0) { K1S_T(1); continue; }
            if (tid == 0) s_base = atomicAdd(&cur->groups_anchors, (unsigned long long)tot_g << 32 | tot_a);
            __syncthreads();
            const unsigned long long base = s_base;
            const uint32_t base_g = (uint32_t)(base >> 32), base_a = (uint32_t)base;
            const bool fits = (unsigned long long)base_a + tot_a <= anchor_cap && (unsigned long long)base_g + tot_g <= group_cap;
            if (!fits) { if (tid == 0) atomicAdd(&ctr->overflow_reads, 1u); continue; }
            for (uint32_t i = lo; i < hi; ++i) {
                const uint32_t c = tab[i];
                if (c < (uint32_t)CH_MIN_CNT) { tab[i] = 0xFFFFFFFFu; continue; }
                const uint32_t k = 255 - (c > 255 ? 255 : c);
                const uint32_t slot = atomicAdd(&s_hist[k], 1u);                       // (the class's next group; its anchors: slot * c behind the class's first, or the class's running end)
                const uint32_t a_off = base_a + (c < 255 ? s_hista[k] + (slot - s_first[k]) * c : atomicAdd(&s_hista[k], c));
                SGroup g; g.read = r; g.key = (uint32_t)rev << 15 | (rid_lo + i); g.off = a_off; g.cnt = c;
                groups[base_g + slot] = g;
                tab[i] = a_off;                                                      // the running slot of the target's anchors
            }
            __syncthreads();
            K1S_T(1);
            // C. scatter (any order: the DP sorts its target's anchors)
            for (uint32_t s = 0; s < K; ++s) {
                const uint4 sd = seeds[s0 + s];
                const uint32_t strand = sd.x & 1u, qpos = sd.x >> 1;
                const uint32_t qp = rev ? (uint32_t)(qlen - ((int)qpos + 1 - MZ_K) - 1) : qpos;
                for (uint32_t k = tid; k < sd.z; k += CH_THREADS) {
                    const uint32_t o = ix.occ[sd.y + k], rid = o >> 17;
                    if ((int)((o & 1u) != strand) == rev && rid >= rid_lo && rid < rid_hi && tab[rid - rid_lo] != 0xFFFFFFFFu) {
                        const uint32_t slot = atomicAdd(&tab[rid - rid_lo], 1u);
                        anchors[slot] = ((o >> 1) & 0xFFFFu) << 16 | (qp & 0xFFFFu);
                    }
                }
            }
            __syncthreads();
            K1S_T(2);
            n_groups_read += tot_g; n_anchors_read += tot_a;
            K1S_T(3);
        }
    }
    // the read's stretch of the batch's chain list: a chain has at least min_cnt anchors of its own, so a third of the anchors of the read's targets bounds their number
    // (a target often gives more than one chain: a chimeric read has two on every target)
    if (tid == 0) {
        const uint32_t cap = n_groups_read ? n_anchors_read / CH_MIN_CNT : 0u;
        const uint32_t off = cap ? atomicAdd(&cur->chains, cap) : 0u;
        read_chain[r] = make_uint3(off, cap, 0u);
    }
}

// The chaining DP.  A wave takes 64 neighbouring targets of the group list, one lane each.  For a target with n <= 26 anchors the bounds of chain_anchors cannot act: its inner loop
// looks at < 26 predecessors, so the skip counter never passes max_chain_skip = 25 (no early break, hence no use of the marks t[] and no rescue through max_ii), and max_chain_iter
// = 5000 is out of reach; what is left is the plain recurrence  f[i] = max(k, max_{j < i, r_i - r_j <= max_gap} f[j] + sc(i, j)),  ties to the largest j  -- run here out of REGISTERS,
// fully unrolled (sizes 4 / 8 / 16 / 26 by the wave's largest target), after a sorting network has put the anchors in an_cmp's order.  The packed (f, p) rows then go to LDS for the
// backtrack, which walks data-dependent links.  Targets with more anchors (about one in a hundred) run chain_anchors step for step out of LDS, a few lanes at a time.
constexpr int DP_REG_MAX = 26;
constexpr int DP_BIG_MAX = 4096;                        // anchors of the largest target that is chained (a small database: every minimizer of a read is a seed, and a target holds them all)
constexpr int DP_BIG_WORDS = 3 * DP_BIG_MAX;            // LDS words of the one-wave workgroups of k1s_dp_big_kernel<true>: the 12-byte entries of a target with more than 64 anchors
constexpr int DP_WAVE_WORDS = DP_REG_MAX * 32 * 2;      // LDS words of a wave: [row][lane] of packed rows (4 bytes for up to 8 anchors, 8 bytes for up to 26), or 12-byte SDp entries of the large targets

template <int N> __device__ __forceinline__ void sort_network(uint32_t (&k)[N]) {
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ stride;
                if (l > i) {
                    const bool up = (i & size) == 0;
                    const uint32_t a = k[i], b = k[l];
                    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
                    k[i] = up ? lo : hi; k[l] = up ? hi : lo;
                }
            }
        }
    }
}
template <int N> __device__ __forceinline__ uint32_t pick_key(const uint32_t (&k)[N], int idx) {
    uint32_t v = k[0];
#pragma unroll
    for (int i = 1; i < N; ++i) v = idx == i ? k[i] : v;
    return v;
}

// The chains the lanes of a wave found (have) go to their reads' stretches of the chain list: the lanes that share a read make ONE reservation (the 64 targets of a wave are
// mostly one read's: a lane-by-lane atomic put 64 same-address operations behind one another, 19 M of them per 10,000 reads -- that, not the DP, was the kernel's time).
// Called by all lanes that are active together.
__device__ __forceinline__ void emit_chains(bool have, const SGroup& grp, int sc, int zf, uint32_t end_key, uint32_t first_key, int cnt, SChain* __restrict__ chains, uint32_t chain_cap,
                                            uint3* __restrict__ read_chain, int lane) {
    unsigned long long todo = __ballot(have);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)grp.read, leader);
        const unsigned long long same = __ballot(have && grp.read == r0);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&read_chain[r0].z, (uint32_t)__builtin_popcountll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (have && grp.read == r0) {
            const uint3 rc = read_chain[r0];
            const uint32_t at = base + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
            if (at < rc.y && rc.x + at < chain_cap) { SChain c; c.score = sc; c.f_end = zf; c.key = grp.key; c.end = end_key; c.first = first_key; c.cnt = (uint32_t)cnt; chains[rc.x + at] = c; }
        }
        todo &= ~same;
    }
}

// NS: size of the sorting network (power of two), ND: rows of the DP (<= NS)
template <int NS, int ND>
__device__ __forceinline__ void dp_in_registers(const SGroup& grp, int n, const uint32_t* __restrict__ anchors, const int32_t* __restrict__ pen, uint32_t* __restrict__ wl, int lane,
                                                SChain* __restrict__ chains, uint32_t chain_cap, uint3* __restrict__ read_chain) {
    uint32_t k[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) k[i] = i < n ? anchors[grp.off + i] : 0xFFFFFFFFu;
#if SP_DP_EXP == 1
    if (k[0] == 12345u) wl[lane] = k[NS - 1];
    return;
#endif
    sort_network<NS>(k);
#if SP_DP_EXP == 2
    if (k[0] == 12345u) wl[lane] = k[NS - 1];
    return;
#endif
    int f[ND]; int p[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        int fi = MZ_K, pi = 0xFF;
        const int ri = (int)(k[i] >> 16), qi = (int)(k[i] & 0xFFFFu);
#pragma unroll
        for (int j = i - 1; j >= 0; --j) {
            const int dq = qi - (int)(k[j] & 0xFFFFu), dr = ri - (int)(k[j] >> 16);
            const int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
            const bool ok = dq > 0 && dq <= CH_MAX_GAP && dr != 0 && dr <= CH_MAX_GAP && dd <= CH_BW;
            const int cand = f[j] + (MZ_K < dg ? MZ_K : dg) - pen[dd <= CH_BW ? dd : 0];
            const bool better = ok && cand > fi;
            fi = better ? cand : fi; pi = better ? j : pi;
        }
        f[i] = fi; p[i] = pi;
        if (i < n) wl[i * 64 + lane] = (uint32_t)fi << 8 | (uint32_t)pi;
    }
#if SP_DP_EXP == 3
    return;
#endif
    // backtrack, best end first ((f, index) descending); t (bits 24..26): 0 free, 1 in a chain, 2 being walked, 4 tried as an end whose best cut was itself
    auto F = [&](int i) { return (int)((wl[i * 64 + lane] >> 8) & 0xFFFFu); };
    auto P = [&](int i) { const int v = (int)(wl[i * 64 + lane] & 0xFFu); return v == 0xFF ? -1 : v; };
    auto T = [&](int i) { return (int)(wl[i * 64 + lane] >> 24); };
    auto setT = [&](int i, int t) { wl[i * 64 + lane] = (wl[i * 64 + lane] & 0x00FFFFFFu) | (uint32_t)t << 24; };
    bool active = n > 0;
    while (__ballot(active)) {
        bool have = false; int sc = 0, zf = 0, cnt = 0; uint32_t end_key = 0, first_key = 0;
        if (active) {
            int zi = -1;
            for (int i = 0; i < n; ++i) { const uint32_t v = wl[i * 64 + lane]; const int fi = (int)((v >> 8) & 0xFFFFu); if ((v >> 24) == 0 && fi >= CH_MIN_SCORE && (zi < 0 || fi >= zf)) { zi = i; zf = fi; } }
            if (zi < 0) active = false;
            else {
                int i = zi, end_i = -1, max_i = zi, max_s = 0;
                do {
                    setT(i, 2); end_i = i = P(i);
                    const int s2 = i < 0 ? zf : zf - F(i);
                    if (s2 > max_s) { max_s = s2; max_i = i; }
                    else if (max_s - s2 > CH_BW) break;
                } while (i >= 0 && (T(i) & 3) == 0);
                for (i = zi; i >= 0 && i != end_i; i = P(i)) setT(i, 0);
                end_i = max_i;
                int first = zi;
                for (i = zi; i != end_i; i = P(i)) { setT(i, 1); first = i; ++cnt; }
                sc = i < 0 ? zf : zf - F(i);
                if (cnt == 0) setT(zi, 4);
                have = sc >= CH_MIN_SCORE && cnt >= CH_MIN_CNT;
                if (have) { end_key = pick_key<NS>(k, zi); first_key = pick_key<NS>(k, first); }
            }
        }
        emit_chains(have, grp, sc, zf, end_key, first_key, cnt, chains, chain_cap, read_chain, lane);
    }
}

// Targets of 9 .. 26 anchors: the same recurrence with the rows in LDS ([row][lane], 8 bytes: key | f << 32 | p << 48 | t << 56) and rolled loops -- unrolled over registers the
// 26-row form alone was 65 KB of code, more than the instruction cache the CUs share, and every wave of the kernel waited for instruction fetches.
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void dp_in_lds_rows(const SGroup& grp, int n, int mx, const uint32_t* __restrict__ anchors, const int32_t* __restrict__ pen, unsigned long long* __restrict__ wl, int lane,
                                               SChain* __restrict__ chains, uint32_t chain_cap, uint3* __restrict__ read_chain) {
    uint32_t k[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) k[i] = i < n ? anchors[grp.off + i] : 0xFFFFFFFFu;
    sort_network<32>(k);
    // the two halves of the wave one after the other on the same rows (13 KB of rows per wave left the kernel two workgroups per CU, and its short paths live on occupancy)
    const int n_all = n, col = lane & 31;
    for (int half = 0; half < 2; ++half) {
    n = (lane >> 5) == half ? n_all : 0;
#pragma unroll
    for (int i = 0; i < DP_REG_MAX; ++i) if (i < n) wl[i * 32 + col] = k[i];
    for (int i = 0; i < mx; ++i) {
        if (i >= n) continue;
        const uint32_t ki = (uint32_t)wl[i * 32 + col];
        const int ri = (int)(ki >> 16), qi = (int)(ki & 0xFFFFu);
        int fi = MZ_K, pi = 0xFF;
#pragma unroll 4
        for (int j = i - 1; j >= 0; --j) {
            const unsigned long long e = wl[j * 32 + col];
            const int dq = qi - (int)((uint32_t)e & 0xFFFFu), dr = ri - (int)((uint32_t)e >> 16);
            const int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
            const bool ok = dq > 0 && dq <= CH_MAX_GAP && dr != 0 && dr <= CH_MAX_GAP && dd <= CH_BW;
            const int cand = (int)((e >> 32) & 0xFFFFu) + (MZ_K < dg ? MZ_K : dg) - pen[dd <= CH_BW ? dd : 0];
            const bool better = ok && cand > fi;
            fi = better ? cand : fi; pi = better ? j : pi;
        }
        wl[i * 32 + col] = (unsigned long long)ki | (unsigned long long)fi << 32 | (unsigned long long)pi << 48;
    }
    // backtrack, best end first ((f, index) descending); t: 0 free, 1 in a chain, 2 being walked, 4 tried as an end whose best cut was itself
    auto F = [&](int i) { return (int)((wl[i * 32 + col] >> 32) & 0xFFFFu); };
    auto P = [&](int i) { const int v = (int)((wl[i * 32 + col] >> 48) & 0xFFu); return v == 0xFF ? -1 : v; };
    auto T = [&](int i) { return (int)(wl[i * 32 + col] >> 56); };
    auto setT = [&](int i, int t) { wl[i * 32 + col] = (wl[i * 32 + col] & 0x00FFFFFFFFFFFFFFull) | (unsigned long long)t << 56; };
    bool active = n > 0;
    while (__ballot(active)) {
        bool have = false; int sc = 0, zf = 0, cnt = 0; uint32_t end_key = 0, first_key = 0;
        if (active) {
            int zi = -1;
            for (int i = 0; i < n; ++i) { const unsigned long long v = wl[i * 32 + col]; const int fi = (int)((v >> 32) & 0xFFFFu); if ((v >> 56) == 0 && fi >= CH_MIN_SCORE && (zi < 0 || fi >= zf)) { zi = i; zf = fi; } }
            if (zi < 0) active = false;
            else {
                int i = zi, end_i = -1, max_i = zi, max_s = 0;
                do {
                    setT(i, 2); end_i = i = P(i);
                    const int s2 = i < 0 ? zf : zf - F(i);
                    if (s2 > max_s) { max_s = s2; max_i = i; }
                    else if (max_s - s2 > CH_BW) break;
                } while (i >= 0 && (T(i) & 3) == 0);
                for (i = zi; i >= 0 && i != end_i; i = P(i)) setT(i, 0);
                end_i = max_i;
                int first = zi;
                for (i = zi; i != end_i; i = P(i)) { setT(i, 1); first = i; ++cnt; }
                sc = i < 0 ? zf : zf - F(i);
                if (cnt == 0) setT(zi, 4);
                have = sc >= CH_MIN_SCORE && cnt >= CH_MIN_CNT;
                if (have) { end_key = (uint32_t)wl[zi * 32 + col]; first_key = (uint32_t)wl[first * 32 + col]; }
            }
        }
        emit_chains(have, grp, sc, zf, end_key, first_key, cnt, chains, chain_cap, read_chain, lane);
    }
    wave_sync_lds();
    }
}

// cross-lane moves by DPP (no LDS traffic): src from `shift` lanes below inside a row of 16, from the last lane of the row(s) below (row_bcast), from the lane below across the wave;
// lanes without a source keep `fill`
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_move(int fill, int src) { return __builtin_amdgcn_update_dpp(fill, src, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_prefix_max(int v) {          // inclusive, lanes 0..63
    const int NEG = INT32_MIN;
    int t;
    t = dpp_move<0x111, 0xf>(NEG, v); v = t > v ? t : v;
    t = dpp_move<0x112, 0xf>(NEG, v); v = t > v ? t : v;
    t = dpp_move<0x114, 0xf>(NEG, v); v = t > v ? t : v;
    t = dpp_move<0x118, 0xf>(NEG, v); v = t > v ? t : v;
    t = dpp_move<0x142, 0xa>(NEG, v); v = t > v ? t : v;
    t = dpp_move<0x143, 0xc>(NEG, v); v = t > v ? t : v;
    return v;
}
// inclusive prefix composition of the maps n -> max(n + a, b), the lower lane's map first
__device__ __forceinline__ void wave_prefix_maps(int& a, int& b) {
    const int NEG = -(1 << 20);
#define SP_MAP_STEP(CTRL, MASK) { const int ta = dpp_move<CTRL, MASK>(0, a), tb = dpp_move<CTRL, MASK>(NEG, b); const int nb = tb + a > b ? tb + a : b; a = ta + a; b = nb; }
    SP_MAP_STEP(0x111, 0xf) SP_MAP_STEP(0x112, 0xf) SP_MAP_STEP(0x114, 0xf) SP_MAP_STEP(0x118, 0xf) SP_MAP_STEP(0x142, 0xa) SP_MAP_STEP(0x143, 0xc)
#undef SP_MAP_STEP
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// chain_anchors (oracle/mm2.c) on the anchors of ONE target with more than DP_REG_MAX anchors, by a whole wave: anchor i takes its predecessors 64 at a time, lane l looking at
// j = jb - l, and the loop's running state -- the best score so far, the skip counter with its marks, the early break -- is replayed over the lanes in the order the loop would
// visit them:
//   * a predecessor's mark t[j] == i + 1 is set by the predecessors above it that chain through it (p[j'] == j, j' > j): every valid lane writes its mark, then reads its own
//     (marks of lanes behind a break are never looked at again: the marks of step i only count in step i);
//   * "improves on everything before it" is an exclusive prefix maximum; the skip counter (- 1 but not below 0 at an improvement, + 1 at a marked predecessor that does not improve)
//     is a prefix composition of maps n -> max(n + a, b); the loop breaks at the first lane whose increment takes the counter past max_chain_skip.
// all: the wave's lanes are all here (uniform arguments)
// grp: the target the wave chains (uniform).  Its chains also go to the lanes that hold a target with the very same anchors but for a shift along the target (member; mine =
// the lane's own target, shift = its target positions minus grp's): the alleles nearest a read mostly carry the read's seeds at the same places
__device__ __forceinline__ void dp_by_wave(const SGroup& grp, SDp* __restrict__ a, const uint32_t* __restrict__ anchors, const int32_t* __restrict__ pen,
                                           SChain* __restrict__ chains, uint32_t chain_cap, uint3* __restrict__ read_chain, int lane, SeedCounters* __restrict__ ctr,
                                           bool member, const SGroup& mine, uint32_t shift) {
    const int n = (int)grp.cnt;
    unsigned long long t_last = 0;
#ifdef SP_K1S_TIMING
    t_last = wall_clock64();
#endif
    (void)t_last; (void)ctr;
    for (int i = lane; i < n; i += 64) a[i].key = anchors[grp.off + i];
    wave_sync();
    // an_cmp's order: every key's rank among the target's keys (they are distinct), the keys moved to their ranks through the (p, t) words
    for (int i = lane; i < n; i += 64) { const uint32_t kx = a[i].key; int rk = 0; for (int j = 0; j < n; ++j) rk += a[j].key < kx ? 1 : 0; a[i].f = rk; }
    wave_sync();
    for (int i = lane; i < n; i += 64) { uint32_t* dst = reinterpret_cast<uint32_t*>(&a[a[i].f].p); *dst = a[i].key; }
    wave_sync();
    for (int i = lane; i < n; i += 64) { a[i].key = *reinterpret_cast<uint32_t*>(&a[i].p); }
    wave_sync();
    for (int i = lane; i < n; i += 64) { a[i].t = 0; a[i].p = -1; a[i].f = 0; }
    wave_sync();
    int st = 0, max_ii = -1;
    for (int i = 0; i < n; ++i) {
        const uint32_t ai = a[i].key; const int ri = (int)(ai >> 16);
        while (st < i && ri > (int)(a[st].key >> 16) + CH_MAX_GAP) ++st;
        if (i - st > CH_MAX_ITER) st = i - CH_MAX_ITER;
        int max_f = MZ_K, max_j = -1, n_skip = 0, end_j = st - 1;
        for (int jb = i - 1; jb >= st; jb -= 64) {
            const int j = jb - lane;
            const bool in = j >= st;
            SDp aj; aj.key = 0; aj.f = 0; aj.p = -1; aj.t = 0;
            if (in) aj = a[j];
            int32_t sc = in ? chain_sc(ai, aj.key, pen) : INT32_MIN;
            const bool valid = sc != INT32_MIN;
            const int v = valid ? sc + aj.f : INT32_MIN;
            if (valid && aj.p >= 0) a[aj.p].t = (uint16_t)(i + 1);
            wave_sync();
            const bool marked = in && a[j].t == (uint16_t)(i + 1);
            // exclusive prefix maximum of v over the lanes, behind the incoming best
            int before = dpp_move<0x138, 0xf>(INT32_MIN, wave_prefix_max(v));       // (wave_shr:1)
            before = before > max_f ? before : max_f;
            const bool isnew = valid && v > before;
            // the skip counter after every lane: maps n -> max(n + ca, cb), composed lane by lane
            int ca = isnew ? -1 : (valid && marked ? 1 : 0), cb = isnew ? 0 : -(1 << 20);
            wave_prefix_maps(ca, cb);
            const int n_after = n_skip + ca > cb ? n_skip + ca : cb;
            const unsigned long long brk = __ballot(valid && !isnew && marked && n_after > CH_MAX_SKIP);
            const int lb = brk ? __builtin_ctzll(brk) : 64;                    // the lane at which the loop breaks
            const unsigned long long news = __ballot(isnew) & (lb >= 63 ? ~0ull : ((2ull << lb) - 1ull));
            if (news) { const int ln = 63 - __builtin_clzll(news); max_f = __builtin_amdgcn_readlane(v, ln); max_j = jb - ln; }
            if (brk) { end_j = jb - lb; n_skip = __builtin_amdgcn_readlane(n_after, lb); break; }
            const int last = jb - st < 63 ? jb - st : 63;                       // the chunk's last lane in range
            n_skip = __builtin_amdgcn_readlane(n_after, last);
            end_j = st - 1;
        }
        if (max_ii < 0 || ri - (int)(a[max_ii].key >> 16) > CH_MAX_GAP) {
            // the best f among [st, i): the largest index among equals (the loop walks downwards and takes strict improvements)
            int bf = INT32_MIN, bj = -1;
            for (int j = i - 1 - lane; j >= st; j -= 64) { const int fj = a[j].f; if (fj > bf) { bf = fj; bj = j; } }
            for (int o = 32; o > 0; o >>= 1) { const int of = __shfl_xor(bf, o), oj = __shfl_xor(bj, o); if (of > bf || (of == bf && oj > bj)) { bf = of; bj = oj; } }
            max_ii = bj;
        }
        if (max_ii >= 0 && max_ii < end_j) {
            const int32_t tmp = chain_sc(ai, a[max_ii].key, pen);
            if (tmp != INT32_MIN && max_f < tmp + a[max_ii].f) { max_f = tmp + a[max_ii].f; max_j = max_ii; }
        }
        wave_sync();
        if (lane == 0) { a[i].f = max_f; a[i].p = (int16_t)max_j; }
        wave_sync();
        if (max_ii < 0 || (ri - (int)(a[max_ii].key >> 16) <= CH_MAX_GAP && a[max_ii].f < max_f)) max_ii = i;
    }
    // backtrack, best end first ((f, index) descending); t: 0 free, 1 in a chain, 2 being walked, 4 tried as an end whose best cut was itself.  The walk is one lane's work
    // (data-dependent links); every lane follows it on the same addresses
    for (int i = lane; i < n; i += 64) a[i].t = 0;
    wave_sync();
    for (;;) {
        int zf = INT32_MIN, zi = -1;
        for (int i = lane; i < n; i += 64) { const int fi = a[i].f; if (a[i].t == 0 && fi >= CH_MIN_SCORE && (fi > zf || (fi == zf && i > zi))) { zf = fi; zi = i; } }
        for (int o = 32; o > 0; o >>= 1) { const int of = __shfl_xor(zf, o), oi = __shfl_xor(zi, o); if (of > zf || (of == zf && oi > zi)) { zf = of; zi = oi; } }
        if (zi < 0) break;
        // (every lane walks the same links and writes the same marks: a wave's LDS operations keep their order, no barrier is needed inside the walk)
        int i = zi, end_i = -1, max_i = zi, max_s = 0;
        do {
            a[i].t = 2;
            end_i = i = a[i].p;
            const int s2 = i < 0 ? zf : zf - a[i].f;
            if (s2 > max_s) { max_s = s2; max_i = i; }
            else if (max_s - s2 > CH_BW) break;
        } while (i >= 0 && (a[i].t & 3) == 0);
        for (i = zi; i >= 0 && i != end_i; i = a[i].p) a[i].t = 0;
        end_i = max_i;
        int cnt = 0, first = zi;
        for (i = zi; i != end_i; i = a[i].p) { a[i].t = 1; first = i; ++cnt; }
        const int sc = i < 0 ? zf : zf - a[i].f;
        if (cnt == 0) a[zi].t = 4;
        wave_sync();
        emit_chains(member && sc >= CH_MIN_SCORE && cnt >= CH_MIN_CNT, mine, sc, zf, a[zi].key + (shift << 16), a[first].key + (shift << 16), cnt, chains, chain_cap, read_chain, lane);
    }
    wave_sync();
}

// The same for a target of up to 64 anchors without LDS: anchor j lives in the registers of lane 63 - j (key, f, p, mark), so that the loop's downward walk over the
// predecessors is the upward order of the lanes and the prefix operations run as they stand.  Row i: every lane in [63 - (i - 1), 63 - st] scores its own anchor against
// anchor i (read by v_readlane); the marks t[p[j]] = i + 1 travel as ONE forward permute (a lane that has none sends a zero to lane 63 - i, which is not looked at); the
// backtrack follows the links with v_readlane.  About 0.2 us a row against 1 us out of LDS.
__device__ __forceinline__ int rl(int v, int lane_idx) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane_idx)); }

__device__ __forceinline__ void dp_by_wave_regs(const SGroup& grp, const uint32_t* __restrict__ anchors, const int32_t* __restrict__ pen, SChain* __restrict__ chains, uint32_t chain_cap,
                                                uint3* __restrict__ read_chain, int lane, bool member, const SGroup& mine, uint32_t shift) {
    const int n = (int)grp.cnt;                                     // 27 .. 64
    const int my_j = 63 - lane;
    // the anchors into an_cmp's order: rank of every key among the target's keys (they are distinct), sent to lane 63 - rank
    uint32_t raw = lane < n ? anchors[grp.off + lane] : 0xFFFFFFFFu;
    int rank = 0;
    for (int x = 0; x < n; ++x) { const uint32_t kx = (uint32_t)__builtin_amdgcn_readlane((int)raw, x); rank += kx < raw ? 1 : 0; }
    // (lanes >= n hold 0xFFFFFFFF and rank n: they send their filler to the lanes 63 - n and below, which hold no anchor; several may land on one lane, all the same value)
    const int dst = lane < n ? 63 - rank : lane - n;                // lanes n .. 63 fill lanes 0 .. 63 - n: a bijection together with the ranks
    const uint32_t key = (uint32_t)__builtin_amdgcn_ds_permute(dst << 2, (int)raw);
    int f = 0, p = -1;
    int st = 0, max_ii = -1;
    for (int i = 0; i < n; ++i) {
        const uint32_t ai = (uint32_t)rl((int)key, 63 - i); const int ri = (int)(ai >> 16);
        while (st < i && ri > (int)((uint32_t)rl((int)key, 63 - st) >> 16) + CH_MAX_GAP) ++st;
        int max_f = MZ_K, max_j = -1, end_j = st - 1;
        if (i > st && i <= CH_MAX_SKIP) {
            // fewer than 26 predecessors: the skip counter cannot pass max_chain_skip, the loop sees them all -- the best score, ties to the largest index (the first met walking down)
            const bool in = my_j >= st && my_j < i;
            const int32_t sc = in ? chain_sc(ai, key, pen) : INT32_MIN;
            const int v = sc != INT32_MIN ? sc + f : INT32_MIN;
            const int m = __builtin_amdgcn_readlane(wave_prefix_max(v), 63);
            if (m > max_f) { max_f = m; max_j = 63 - (int)__builtin_ctzll(__ballot(v == m)); }
        } else if (i > st) {
            const bool in = my_j >= st && my_j < i;
            const int32_t sc = in ? chain_sc(ai, key, pen) : INT32_MIN;
            const bool valid = sc != INT32_MIN;
            const int v = valid ? sc + f : INT32_MIN;
            const bool sends = valid && p >= 0;
            const int got = __builtin_amdgcn_ds_permute((sends ? 63 - p : 63 - i) << 2, sends ? 1 : 0);
            const bool marked = in && got == 1;
            int before = dpp_move<0x138, 0xf>(INT32_MIN, wave_prefix_max(v));
            before = before > max_f ? before : max_f;
            const bool isnew = valid && v > before;
            int ca = isnew ? -1 : (valid && marked ? 1 : 0), cb = isnew ? 0 : -(1 << 20);
            wave_prefix_maps(ca, cb);
            const int n_after = ca > cb ? ca : cb;                               // (the counter starts every row at 0)
            const unsigned long long brk = __ballot(valid && !isnew && marked && n_after > CH_MAX_SKIP);
            const int lb = brk ? __builtin_ctzll(brk) : 64;
            const unsigned long long news = __ballot(isnew) & (lb >= 63 ? ~0ull : ((2ull << lb) - 1ull));
            if (news) { const int ln = 63 - __builtin_clzll(news); max_f = rl(v, ln); max_j = 63 - ln; }
            if (brk) end_j = 63 - lb;
        }
        if (max_ii < 0 || ri - (int)((uint32_t)rl((int)key, 63 - max_ii) >> 16) > CH_MAX_GAP) {
            // the best f among [st, i): the largest index among equals
            const bool in = my_j >= st && my_j < i;
            int bk = in ? (f << 8 | my_j) : -1;
            for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(bk, o); bk = t > bk ? t : bk; }
            max_ii = bk < 0 ? -1 : (bk & 0xFF);
        }
        if (max_ii >= 0 && max_ii < end_j) {
            const int32_t tmp = chain_sc(ai, (uint32_t)rl((int)key, 63 - max_ii), pen);
            const int fm = rl(f, 63 - max_ii);
            if (tmp != INT32_MIN && max_f < tmp + fm) { max_f = tmp + fm; max_j = max_ii; }
        }
        if (my_j == i) { f = max_f; p = max_j; }
        if (max_ii < 0 || (ri - (int)((uint32_t)rl((int)key, 63 - max_ii) >> 16) <= CH_MAX_GAP && rl(f, 63 - max_ii) < max_f)) max_ii = i;
    }
    // backtrack, best end first ((f, index) descending); t: 0 free, 1 in a chain, 2 being walked, 4 tried as an end whose best cut was itself
    int t = 0;
    for (;;) {
        int bk = (my_j < n && t == 0 && f >= CH_MIN_SCORE) ? (f << 8 | my_j) : -1;
        for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(bk, o); bk = x > bk ? x : bk; }
        if (bk < 0) break;
        const int zi = bk & 0xFF, zf = bk >> 8;
        int i = zi, end_i = -1, max_i = zi, max_s = 0;
        do {
            if (my_j == i) t = 2;
            end_i = i = rl(p, 63 - i);
            const int s2 = i < 0 ? zf : zf - rl(f, 63 - i);
            if (s2 > max_s) { max_s = s2; max_i = i; }
            else if (max_s - s2 > CH_BW) break;
        } while (i >= 0 && (rl(t, 63 - i) & 3) == 0);
        for (i = zi; i >= 0 && i != end_i; i = rl(p, 63 - i)) { if (my_j == i) t = 0; }
        end_i = max_i;
        int cnt = 0, first = zi;
        for (i = zi; i != end_i; i = rl(p, 63 - i)) { if (my_j == i) t = 1; first = i; ++cnt; }
        const int sc = i < 0 ? zf : zf - rl(f, 63 - i);
        if (cnt == 0 && my_j == zi) t = 4;
        const uint32_t end_key = (uint32_t)rl((int)key, 63 - zi), first_key = (uint32_t)rl((int)key, 63 - first);
        emit_chains(member && sc >= CH_MIN_SCORE && cnt >= CH_MIN_CNT, mine, sc, zf, end_key + (shift << 16), first_key + (shift << 16), cnt, chains, chain_cap, read_chain, lane);
    }
}

__global__ __launch_bounds__(CH_THREADS) void k1s_dp_kernel(const SGroup* __restrict__ groups, const uint32_t* __restrict__ anchors, const int32_t* __restrict__ pen_tab,
                                                            BatchCursors* __restrict__ cur, SChain* __restrict__ chains, uint32_t chain_cap, uint3* __restrict__ read_chain,
                                                            SBigItem* __restrict__ items, SMember* __restrict__ members, uint32_t big_cap, SeedCounters* __restrict__ ctr) {
    extern __shared__ uint32_t lds[];
    int32_t* pen = reinterpret_cast<int32_t*>(lds);                          // 512
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* wl = lds + 512 + wave * DP_WAVE_WORDS;
    for (int i = tid; i <= CH_BW; i += CH_THREADS) pen[i] = pen_tab[i];
    __syncthreads();
    const uint32_t n_groups = (uint32_t)(cur->groups_anchors >> 32);
    unsigned long long t_last = 0;
#ifdef SP_K1S_TIMING
    t_last = wall_clock64();
#endif
    (void)t_last;
    // the tasks (64 neighbouring targets each) go to the waves in a scrambled order: the first task of every read holds its large targets, and a plain stride that divides the
    // reads' period handed all of those to a few hundred waves; a shared counter (one atomic per task) queued 300,000 operations on one address
    const uint32_t n_tasks = (n_groups + 63) / 64;
    uint32_t pow2 = 1; while (pow2 < n_tasks) pow2 <<= 1;
    const uint32_t n_waves = gridDim.x * (CH_THREADS / 64);
    for (uint32_t k = blockIdx.x * (CH_THREADS / 64) + wave; k < pow2; k += n_waves) {
        const uint32_t task = (k * 0x9E3779B1u) & (pow2 - 1);              // (an odd multiplier: a permutation of [0, pow2))
        if (task >= n_tasks) continue;
        const uint32_t g0 = task * 64;
        const uint32_t gi = g0 + lane;
        SGroup grp; grp.read = 0; grp.key = 0; grp.off = 0; grp.cnt = 0;
        if (gi < n_groups) grp = groups[gi];
        const int n = (int)grp.cnt;
        const int small_n = n <= DP_REG_MAX ? n : 0;
        int mx = small_n;
        for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(mx, o); mx = t > mx ? t : mx; }
        K1S_TW(7);
#if SP_DP_EXP == 5
        if (mx > 8) { }
        else
#endif
        if (mx > 8) dp_in_lds_rows(grp, small_n, mx, anchors, pen, reinterpret_cast<unsigned long long*>(wl), lane, chains, chain_cap, read_chain);
        else if (mx > 4) dp_in_registers<8, 8>(grp, small_n, anchors, pen, wl, lane, chains, chain_cap, read_chain);
        else if (mx > 0) dp_in_registers<4, 4>(grp, small_n, anchors, pen, wl, lane, chains, chain_cap, read_chain);
#ifdef SP_K1S_TIMING
        if (lane == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&ctr->t[mx > 8 ? 5 : 4], now_ - t_last); t_last = now_; }
#endif
        // the large targets of the wave (more than DP_REG_MAX anchors: the first tasks of a read hold them).  Their signatures -- the anchors of a target relative to its first
        // target position, as an unordered set (two 64-bit sums of mixed words): targets with the same signature and size chain alike, up to the shift -- and one work item per
        // signature for k1s_dp_big_kernel, which gives every item a wave of its own
#if SP_DP_EXP == 4
        const unsigned long long big = 0;
#else
        const unsigned long long big = __ballot(n > DP_REG_MAX);
#endif
        if (!big) continue;
        unsigned long long h1 = 0, h2 = 0; uint32_t min_r = 0;
        for (unsigned long long todo = big; todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)grp.off, l), cnt = (uint32_t)__builtin_amdgcn_readlane((int)grp.cnt, l);
            uint32_t mn = 0xFFFFu;
            for (uint32_t i = lane; i < cnt; i += 64) { const uint32_t r = anchors[off + i] >> 16; mn = r < mn ? r : mn; }
            for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)mn, o); mn = t < mn ? t : mn; }
            unsigned long long s1 = 0, s2 = 0;
            for (uint32_t i = lane; i < cnt; i += 64) {
                const unsigned long long x = anchors[off + i] - (mn << 16);
                s1 += mix64((x ^ 0x9E3779B97F4A7C15ull) & MZ_MASK) * 0x9E3779B97F4A7C15ull + x;
                unsigned long long y = (x + 0x632BE59BD9B4E019ull) * 0xD6E8FEB86659FD93ull; y ^= y >> 32; y *= 0xD6E8FEB86659FD93ull; y ^= y >> 32;
                s2 += y;
            }
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            if (lane == l) { h1 = s1; h2 = s2; min_r = mn; }
        }
        // the first lane of every signature is chained; the others take its chains
        int rep = -1;
        for (unsigned long long todo = big; todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const unsigned long long b1 = __shfl(h1, l), b2 = __shfl(h2, l);
            const int bn = __builtin_amdgcn_readlane(n, l);
            if (rep < 0 && n > DP_REG_MAX && n == bn && h1 == b1 && h2 == b2) rep = l;
        }
        const unsigned long long reps = __ballot(n > DP_REG_MAX && rep == lane);
        uint32_t item0 = 0, mem0 = 0;
        if (lane == 0) { item0 = atomicAdd(&cur->big_items, (uint32_t)__builtin_popcountll(reps)); mem0 = atomicAdd(&cur->big_members, (uint32_t)__builtin_popcountll(big)); }
        item0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)item0); mem0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mem0);
        uint32_t k_item = 0, k_mem = 0;
        for (unsigned long long todo = reps; todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const unsigned long long mem = __ballot(n > DP_REG_MAX && rep == l);
            const uint32_t rep_min = (uint32_t)__builtin_amdgcn_readlane((int)min_r, l);
            if (item0 + k_item < big_cap && mem0 + k_mem + (uint32_t)__builtin_popcountll(mem) <= big_cap) {
                if (rep == l && n > DP_REG_MAX) {
                    SMember m; m.read = grp.read; m.key = grp.key; m.shift = min_r - rep_min; m.pad = 0;
                    members[mem0 + k_mem + (uint32_t)__builtin_popcountll(mem & ((1ull << lane) - 1ull))] = m;
                }
                if (lane == l) { SBigItem it; it.g = grp; it.member_off = mem0 + k_mem; it.n_members = (uint32_t)__builtin_popcountll(mem); it.pad0 = it.pad1 = 0; items[item0 + k_item] = it; }
            } else if (lane == l) atomicAdd(&ctr->overflow_reads, 1u);
            ++k_item; k_mem += (uint32_t)__builtin_popcountll(mem);
        }
        K1S_TW(6);
    }
}

// every large target's item: a whole wave chains it and hands the chains to the item's members.  Two launches share the list: LONG = false takes the targets of up to 64 anchors
// (registers, no LDS beyond the penalty table: four waves a workgroup), LONG = true the longer ones (one wave a workgroup with 48 KB of LDS rows)
template <bool LONG>
__global__ __launch_bounds__(LONG ? 64 : CH_THREADS) void k1s_dp_big_kernel(const SBigItem* __restrict__ items, const SMember* __restrict__ members, const uint32_t* __restrict__ anchors,
                                                                            const int32_t* __restrict__ pen_tab, BatchCursors* __restrict__ cur, uint32_t big_cap, SChain* __restrict__ chains,
                                                                            uint32_t chain_cap, uint3* __restrict__ read_chain, SeedCounters* __restrict__ ctr) {
    extern __shared__ uint32_t lds[];
    int32_t* pen = reinterpret_cast<int32_t*>(lds);                          // 512
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* wl = lds + 512;
    for (int i = tid; i <= CH_BW; i += (int)blockDim.x) pen[i] = pen_tab[i];
    __syncthreads();
    const uint32_t n_items = cur->big_items < big_cap ? cur->big_items : big_cap;
    uint32_t pow2 = 1; while (pow2 < n_items) pow2 <<= 1;
    const uint32_t n_waves = gridDim.x * (blockDim.x / 64);
    for (uint32_t k = blockIdx.x * (blockDim.x / 64) + wave; k < pow2; k += n_waves) {          // (scrambled order, as in k1s_dp_kernel: neighbouring items are one read's, of like length)
        const uint32_t x = (k * 0x9E3779B1u) & (pow2 - 1);
        if (x >= n_items) continue;
        const SBigItem it = items[x];
        if ((it.g.cnt > 64) != LONG) continue;
        if (LONG && it.g.cnt > (uint32_t)DP_BIG_MAX) { if (lane == 0) atomicAdd(&ctr->overflow_reads, 1u); continue; }      // more anchors than the LDS rows hold: not chained (counted)
        for (uint32_t m0 = 0; m0 < it.n_members; m0 += 64) {              // (more than 64 members: the target is chained once per 64 of them)
            const bool member = m0 + lane < it.n_members;
            SGroup mine = it.g; uint32_t shift = 0;
            if (member) { const SMember m = members[it.member_off + m0 + lane]; mine.read = m.read; mine.key = m.key; shift = m.shift; }
            if (!LONG) dp_by_wave_regs(it.g, anchors, pen, chains, chain_cap, read_chain, lane, member, mine, shift);
            else dp_by_wave(it.g, reinterpret_cast<SDp*>(wl), anchors, pen, chains, chain_cap, read_chain, lane, ctr, member, mine, shift);
        }
    }
}

__global__ __launch_bounds__(CH_THREADS) void k1s_select_kernel(SeqSetView reads, uint32_t r0, uint32_t n_batch, const SChain* __restrict__ chains, int32_t* __restrict__ parent,
                                                                const uint3* __restrict__ read_chain, SSel* __restrict__ sel, uint32_t* __restrict__ sel_cnt,
                                                                SeedCounters* __restrict__ ctr, int best_n, SChain* __restrict__ dbg_chains, int32_t* __restrict__ dbg_parent,
                                                                uint32_t* __restrict__ dbg_n, uint32_t dbg_read) {
    __shared__ unsigned long long s_r1[CH_THREADS / 64], s_r2[CH_THREADS / 64];
    __shared__ int s_ri[CH_THREADS / 64];
    __shared__ int s_prim[PRIM_CAP];
    __shared__ int s_sel[PRIM_CAP + 8];
    const int tid = threadIdx.x;
    if (blockIdx.x >= n_batch) return;
    const uint32_t r = r0 + blockIdx.x;
    const int qlen = reads.len[r];
    const uint3 rc = read_chain[r];
    uint32_t nc = rc.z;
    if (nc > rc.y) { nc = rc.y; if (tid == 0) atomicAdd(&ctr->overflow_reads, 1u); }
    const SChain* sc_chain = chains + rc.x; int32_t* sc_parent = parent + rc.x;
    // rank, parents, selection (set_parent + select_sub of oracle/mm2.c on the ranked list: score, then creation order = (f of the end anchor, anchor index) descending)
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
        // (both caps are far above what `best_n 5` lets through -- a handful of primaries and their secondaries; a read that meets one is counted like every other
        //  truncation: sp_hla_realign_reads reports the count, sp_profile_get "k1s_truncated_reads")
        if (n > SEL_CAP || n_prim >= PRIM_CAP) atomicAdd(&ctr->overflow_reads, 1u);
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
// the cells of the re-score (at the middle diagonal of the found alignment).  An alignment whose ends lie more than K1S_WIDE_SHIFT diagonals apart -- it crosses a long
// insertion / deletion, the cell found it on the wide band -- is re-scored on 256 diagonals in a pass of its own (wide = 1 lists those, wide = 0 the others): minimap2
// chains and aligns across such gaps (bw 500), the 64-diagonal band of the re-score would clip the alignment at the gap
constexpr int K1S_WIDE_SHIFT = 32;
__global__ void k1s_rescore_cells_kernel(const CellDesc* __restrict__ cells, const sp_aln* __restrict__ alns, uint32_t n, CellDesc* __restrict__ out, int wide, SeedCounters* __restrict__ ctr) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    CellDesc c = cells[x];
    const sp_aln a = alns[x];
    bool is_wide = false;
    if (c.diag != SP_NO_DIAG && a.ok) { const int shift = (a.b_end - a.a_end) - (a.b_start - a.a_start); is_wide = shift > K1S_WIDE_SHIFT || shift < -K1S_WIDE_SHIFT; }
    if (c.diag == SP_NO_DIAG || !a.ok || is_wide != (wide != 0)) { c.diag = wide ? SP_NO_DIAG : 0; c.max_ed = -1; }
    else { c.diag = ((a.b_start - a.a_start) + (a.b_end - a.a_end)) / 2; c.max_ed = 127; }
    out[x] = c;
    if (!wide && is_wide && ctr) atomicAdd(&ctr->wide_cells, 1u);
}
__global__ void k1s_merge_wide_kernel(const CellDesc* __restrict__ wide_cells, const sp_affine_aln* __restrict__ af_wide, uint32_t n, sp_affine_aln* __restrict__ af) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n && wide_cells[x].diag != SP_NO_DIAG) af[x] = af_wide[x];
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
    // 2. chains, in batches of reads whose anchors fit the batch buffers (a read of the bundled database has about 20,000 anchors in 2,000 targets)
    uint32_t* d_sel_cnt = (uint32_t*)sp_pool(ctx, "k1s_sel_cnt", (size_t)R * 4);
    SSel* d_sel = (SSel*)sp_pool(ctx, "k1s_sel", (size_t)R * SEL_CAP * sizeof(SSel));
    uint3* d_read_chain = (uint3*)sp_pool(ctx, "k1s_read_chain", (size_t)R * sizeof(uint3));
    BatchCursors* d_cur = (BatchCursors*)sp_pool(ctx, "k1s_cursors", sizeof(BatchCursors));
    uint32_t* h_anchors = (uint32_t*)sp_host_pool(ctx, "k1s_read_anchors", (size_t)R * 4);
    if (!d_sel_cnt || !d_sel || !d_read_chain || !d_cur || !h_anchors) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: selection");
    SP_HIP_CHECK(ctx, hipMemcpyAsync(h_anchors, d_read_anchors, (size_t)R * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t batch_anchors = std::max<uint64_t>(hc.max_anchors, 64ull << 20);            // 256 MB of anchors
    const uint64_t a_cap = std::min<uint64_t>(hc.anchors, batch_anchors + hc.max_anchors), g_cap = a_cap / CH_MIN_CNT + 1;
    if (a_cap >= (1ull << 32)) return sp_fail(ctx, SP_ERR_TOO_LONG, "seeded K1: a read with more than 4 G anchors");
    uint32_t r_batch_max = 0;
    for (uint32_t r0 = 0; r0 < R;) { uint64_t a = 0; uint32_t r1 = r0; while (r1 < R && (r1 == r0 || a + h_anchors[r1] <= batch_anchors)) a += h_anchors[r1++]; r_batch_max = std::max(r_batch_max, r1 - r0); r0 = r1; }
    const uint64_t c_cap = g_cap;                    // (a chain takes at least min_cnt anchors, like a target)
    uint32_t* d_anchors = (uint32_t*)sp_pool(ctx, "k1s_anchors", std::max<uint64_t>(a_cap, 1) * 4);
    SGroup* d_groups = (SGroup*)sp_pool(ctx, "k1s_groups", g_cap * sizeof(SGroup));
    SChain* d_chains = (SChain*)sp_pool(ctx, "k1s_chains", c_cap * sizeof(SChain));
    int32_t* d_parent = (int32_t*)sp_pool(ctx, "k1s_parent", c_cap * 4);
    const uint64_t big_cap = a_cap / (DP_REG_MAX + 1) + 64;
    SBigItem* d_items = (SBigItem*)sp_pool(ctx, "k1s_big_items", big_cap * sizeof(SBigItem));
    SMember* d_members = (SMember*)sp_pool(ctx, "k1s_big_members", big_cap * sizeof(SMember));
    if (!d_anchors || !d_groups || !d_chains || !d_parent || !d_items || !d_members) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: chain buffers");
    const size_t group_lds = (size_t)WIN_KEYS * 4, dp_lds = (size_t)(512 + (CH_THREADS / 64) * DP_WAVE_WORDS) * 4;
    SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k1s_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds));
    SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k1s_dp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds));
    const size_t big_lds = (size_t)(512 + DP_BIG_WORDS) * 4;
    SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k1s_dp_big_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
    SChain* dbg_chains = nullptr; int32_t* dbg_parent = nullptr; uint32_t* dbg_n = nullptr;
    if (dbg) {
        dbg_chains = (SChain*)sp_pool(ctx, "k1s_dbg_chains", g_cap * sizeof(SChain)); dbg_parent = (int32_t*)sp_pool(ctx, "k1s_dbg_parent", g_cap * 4);
        dbg_n = (uint32_t*)sp_pool(ctx, "k1s_dbg_n", 8);
        if (!dbg_chains || !dbg_parent || !dbg_n) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: debug buffers");
        SP_HIP_CHECK(ctx, hipMemsetAsync(dbg_n, 0, 8, ctx->stream));
    }
    for (uint32_t r0 = 0; r0 < R;) {
        uint64_t a = 0; uint32_t r1 = r0;
        while (r1 < R && (r1 == r0 || a + h_anchors[r1] <= batch_anchors)) a += h_anchors[r1++];
        const uint32_t nb_reads = r1 - r0;
        SP_HIP_CHECK(ctx, hipMemsetAsync(d_cur, 0, sizeof(BatchCursors), ctx->stream));
        {
            ProfScope ps(ctx, "k1s_groups", a);
            hipLaunchKernelGGL(k1s_group_kernel, dim3(nb_reads), dim3(CH_THREADS), group_lds, ctx->stream, reads->view(), ix, r0, nb_reads, d_seeds, d_read_seed, d_anchors, (uint32_t)a_cap,
                               d_groups, (uint32_t)g_cap, d_read_chain, d_cur, d_ctr);
        }
        {
            ProfScope ps(ctx, "k1s_dp", a);
            hipLaunchKernelGGL(k1s_dp_kernel, dim3((unsigned)ctx->num_cus * 8), dim3(CH_THREADS), dp_lds, ctx->stream, d_groups, d_anchors, idx->d_pen, d_cur, d_chains, (uint32_t)c_cap, d_read_chain, d_items, d_members, (uint32_t)big_cap, d_ctr);
        }
        {
            ProfScope ps(ctx, "k1s_dp_big", a);
            hipLaunchKernelGGL(k1s_dp_big_kernel<false>, dim3((unsigned)ctx->num_cus * 8), dim3(CH_THREADS), 512 * 4, ctx->stream, d_items, d_members, d_anchors, idx->d_pen, d_cur, (uint32_t)big_cap,
                               d_chains, (uint32_t)c_cap, d_read_chain, d_ctr);
            hipLaunchKernelGGL(k1s_dp_big_kernel<true>, dim3((unsigned)ctx->num_cus * 3), dim3(64), big_lds, ctx->stream, d_items, d_members, d_anchors, idx->d_pen, d_cur, (uint32_t)big_cap,
                               d_chains, (uint32_t)c_cap, d_read_chain, d_ctr);
        }
        {
            ProfScope ps(ctx, "k1s_select", nb_reads);
            hipLaunchKernelGGL(k1s_select_kernel, dim3(nb_reads), dim3(CH_THREADS), 0, ctx->stream, reads->view(), r0, nb_reads, d_chains, d_parent, d_read_chain, d_sel, d_sel_cnt, d_ctr, best_n,
                               dbg_chains, dbg_parent, dbg_n, dbg ? dbg->read : 0xFFFFFFFFu);
        }
        SP_HIP_CHECK(ctx, hipGetLastError());
        r0 = r1;
    }
    // 3. the selected chains through the cell and its re-score
    const uint64_t NC = (uint64_t)R * SEL_CAP;
    CellDesc* d_cells = (CellDesc*)sp_pool(ctx, "k1s_cells", NC * sizeof(CellDesc)); CellDesc* d_rc = (CellDesc*)sp_pool(ctx, "k1s_rc", NC * sizeof(CellDesc));
    sp_aln* d_aln = (sp_aln*)sp_pool(ctx, "k1s_aln", NC * sizeof(sp_aln)); sp_affine_aln* d_af = (sp_affine_aln*)sp_pool(ctx, "k1s_af_out", NC * sizeof(sp_affine_aln));
    if (!d_cells || !d_rc || !d_aln || !d_af) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: cells");
    const sp_affine_opts ao = { 1, 4, 6, 2, 26, 1, 1 };
    const unsigned nb = (unsigned)((NC + 255) / 256);
    hipLaunchKernelGGL(k1s_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_sel, d_sel_cnt, R, idx->d_rid_allele, alleles->d_len, 0, d_cells);
    // the cells run WITH their traceback: the re-score needs the edit positions of exactly these alignments (it used to run every cell a second time for them).  The events of a
    // cell take as many words as its edit cap can reach: 3 % of the longest allele
    const uint32_t ev_stride = (uint32_t)std::min<int64_t>(SP_MAX_ED, (int64_t)(0.03 * (double)alleles->max_len) + 2);
    uint32_t* d_ev = (uint32_t*)sp_pool(ctx, "k1s_ev", NC * (size_t)ev_stride * 4);
    if (!d_ev) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: cell events");
    int rc = sp_launch_cells(ctx, alleles, reads, d_cells, NC, d_aln, d_ev, ev_stride, "k1s_cells", 1);
    if (rc != SP_OK) return rc;
    hipLaunchKernelGGL(k1s_rescore_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_cells, d_aln, (uint32_t)NC, d_rc, 0, d_ctr);
    rc = sp_rescore_mappings(ctx, alleles, reads, d_rc, d_aln, NC, true, ao, 64, d_af, "k1s_af", ev_stride, 0, d_aln, d_ev);
    if (rc != SP_OK) return rc;
    // chains on the reverse strand (a read from the other strand, the homologous gene on the other strand): the same through the reads' reverse complements
    SP_HIP_CHECK(ctx, hipMemcpyAsync(&hc, d_ctr, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
#ifdef SP_K1S_TIMING
    fprintf(stderr, "k1s_group phases (100 MHz ticks summed over workgroups): count %llu scan %llu scatter %llu records %llu; dp kernel waves: task + load %llu, up to 8 anchors %llu, 9..26 %llu, signatures of the large %llu\n", hc.t[0], hc.t[1], hc.t[2], hc.t[3], hc.t[7], hc.t[4], hc.t[5], hc.t[6]);
#endif
    if (hc.overflow_reads) {
        // a capacity of the seeded map was met (seeds per read, the anchor / group batch buffers, more than DP_BIG_MAX anchors on one target, chains per read, PRIM_CAP / SEL_CAP of
        // the selection): the reads concerned were mapped from a truncated list -- minimap2 has no such bounds -- and the caller is told so instead of finding out from a call
        ctx->prof["k1s_truncated_reads"].cells += hc.overflow_reads;
        ctx->warning = "seeded K1: " + std::to_string(hc.overflow_reads) + " capacity events (seed / anchor / chain lists of a read cut short) in the last sp_hla_realign_reads call: "
                       "those reads were mapped from truncated lists; k1_best_n 0 (the exhaustive search) has no such bounds";
    }
    if (hc.wide_cells) {
        // alignments across long insertions / deletions: their re-score on 256 diagonals (the trace on the wide band as well)
        sp_affine_aln* d_afw = (sp_affine_aln*)sp_pool(ctx, "k1s_af_wide", NC * sizeof(sp_affine_aln));
        if (!d_afw) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: wide re-score");
        hipLaunchKernelGGL(k1s_rescore_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_cells, d_aln, (uint32_t)NC, d_rc, 1, (SeedCounters*)nullptr);
        rc = sp_rescore_mappings(ctx, alleles, reads, d_rc, d_aln, NC, true, ao, 256, d_afw, "k1s_afw", 128, 1);
        if (rc != SP_OK) return rc;
        hipLaunchKernelGGL(k1s_merge_wide_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_rc, d_afw, (uint32_t)NC, d_af);
    }
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
        hipLaunchKernelGGL(k1s_rescore_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_cells2, d_aln2, (uint32_t)NC, d_rc, 0, d_ctr);
        rc = sp_rescore_mappings(ctx, alleles, &rset, d_rc, d_aln2, NC, true, ao, 64, d_af2, "k1s_af_rev", 128);
        if (rc != SP_OK) return rc;
        {   // (the reverse strand's alignments across long gaps: whether there are any is known one round trip later; the pass is cheap when there are none)
            sp_affine_aln* d_afw = (sp_affine_aln*)sp_pool(ctx, "k1s_af_wide", NC * sizeof(sp_affine_aln));
            if (!d_afw) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "seeded K1: wide re-score");
            hipLaunchKernelGGL(k1s_rescore_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_cells2, d_aln2, (uint32_t)NC, d_rc, 1, (SeedCounters*)nullptr);
            rc = sp_rescore_mappings(ctx, alleles, &rset, d_rc, d_aln2, NC, true, ao, 256, d_afw, "k1s_afw_rev", 128, 1);
            if (rc != SP_OK) return rc;
            hipLaunchKernelGGL(k1s_merge_wide_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_rc, d_afw, (uint32_t)NC, d_af2);
        }
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
