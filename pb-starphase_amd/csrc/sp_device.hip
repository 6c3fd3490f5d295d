// sp_device.hip -- generic device kernels of libstarphase_hip (gfx950): k-mer vote anchor and the
// one-wavefront-per-cell WFA kernel, plus their launchers.
#include "sp_internal.h"
#include "sp_wfa.hip.h"
#include "sp_anchor.hip.h"
#include <algorithm>

// =============================================================================================
// anchor: one workgroup per (A, B) pair.  A carries a sorted 16-mer table (built on the host at
// first use); every 16-mer of B is looked up and votes for its diagonal in an LDS histogram.
// Replaces minimap2's minimizer seeding + chaining inside Aligner::map (DESIGN.md section 3.1;
// CPU restatement: oracle/align.c osp_anchor).
// =============================================================================================
#ifndef SP_ANCHOR_BUCKET_BITS
#ifndef SP_ANCHOR_BUCKET_BITS
#define SP_ANCHOR_BUCKET_BITS 10
#endif
#endif
#ifndef SP_ANCHOR_ILP
#define SP_ANCHOR_ILP 2          // look-ups a thread runs side by side (anchors of a bench step: 1: 1.83 ms, 2: 1.77, 3: 1.83, 4: 1.89, 8: 2.36)
#endif
#ifndef SP_ANCHOR_THREADS
#define SP_ANCHOR_THREADS 512      // 256: 2.65 ms, 512: 2.14 ms, 1024: 3.65 ms for the 20,000 (read, gene) pairs of the bench step
#endif
__global__ __launch_bounds__(SP_ANCHOR_THREADS) void sp_anchor_kernel(SeqSetView A, KmerIndexView KA, SeqSetView B,
                                                        const uint32_t* __restrict__ a_idx, const uint32_t* __restrict__ b_idx,
                                                        uint64_t n_pairs, int32_t* __restrict__ diag_out, int32_t* __restrict__ votes_out,
                                                        int bins_cap, int topk, int tab_cap, unsigned long long* dbg) {
    extern __shared__ uint32_t lds[];                 // [packed u16 vote bins, 2 per dword][A's sorted 16-mer codes][their positions]
    __shared__ unsigned long long red[SP_ANCHOR_THREADS / 64];
    __shared__ int spread[2];
    // where the codes with each value of the top SP_ANCHOR_BUCKET_BITS bits start in the LDS table: a look-up begins inside its bucket, so
    // the lower bound takes log2(largest bucket) halvings instead of log2(table) (13 -> 4 for a 5 kb gene)
    __shared__ uint16_t bucket[(1 << SP_ANCHOR_BUCKET_BITS) + 1];
    __shared__ int bucket_steps;
    const int tid = threadIdx.x;
    uint32_t* tab_code = lds + ((bins_cap + 1) >> 1);
    int32_t* tab_pos = reinterpret_cast<int32_t*>(tab_code + tab_cap);
    int tab_a = -1;                                   // the A whose table sits in LDS
    // the description of a pair is three dependent global loads deep (pair -> sequence ids -> lengths and offsets -> bases): it is fetched one
    // pair ahead, so the chain runs under the work of the pair before (a workgroup has the CU to itself or shares it with one other)
    struct PairMeta { uint32_t a, b; int m, n; uint64_t woff, k0, k1; };
    auto fetch = [&](uint64_t q) {
        PairMeta x;
        x.a = a_idx[q]; x.b = b_idx[q];
        if (x.b == SP_ANCHOR_SKIP) { x.m = 0; x.n = 0; x.woff = 0; x.k0 = 0; x.k1 = 0; return x; }       // (a pair the caller does not need: no votes, diagonal 0)
        x.m = A.len[x.a]; x.n = B.len[x.b]; x.woff = B.word_off[x.b]; x.k0 = KA.off[x.a]; x.k1 = KA.off[x.a + 1];
        return x;
    };
    PairMeta next = {};
    if (blockIdx.x < n_pairs) next = fetch(blockIdx.x);
    for (uint64_t p = blockIdx.x; p < n_pairs; p += gridDim.x) {
        const PairMeta cur = next;
        if (p + gridDim.x < n_pairs) next = fetch(p + gridDim.x);
        const uint32_t a = cur.a;
        const int m = cur.m, n = cur.n;
        const int nbins = m + n + 1;
        if (m < SP_KMER || n < SP_KMER || nbins > bins_cap) {
            if (tid == 0) for (int k2 = 0; k2 < topk; ++k2) { diag_out[p * topk + k2] = 0; votes_out[p * topk + k2] = 0; }
            continue;
        }
        const int nb32 = (nbins + 1) >> 1;
#ifdef SP_ANCHOR_TIMING
        long long tq[5]; tq[0] = clock64();
#define SP_AT(k) tq[k] = clock64()
#else
#define SP_AT(k)
#endif
        sp_anchor_clear<SP_ANCHOR_THREADS>(lds, nb32);
        const uint32_t* bw = B.words + cur.woff;
        const uint32_t* bn = B.nplane ? B.nplane + cur.woff : nullptr;
        const uint64_t k0 = cur.k0, k1 = cur.k1;
        const int nk = (int)(k1 - k0);
        // A's table moves to LDS once per run of pairs with the same A (a workgroup strides over the pair list, and callers lay
        // pairs out gene-minor, so a workgroup mostly keeps one table): the binary search then runs at LDS latency, not L2 latency
        const bool in_lds = nk <= tab_cap && nk < 65536;
        if (in_lds && (int)a != tab_a) {
            constexpr int NB = 1 << SP_ANCHOR_BUCKET_BITS, SH = 32 - SP_ANCHOR_BUCKET_BITS;
            for (int i = tid; i < nk; i += SP_ANCHOR_THREADS) {
                const uint32_t c = KA.code[k0 + i];
                tab_code[i] = c; tab_pos[i] = KA.pos[k0 + i];
                // entry i opens every bucket after its predecessor's up to its own
                const int from = i > 0 ? (int)(KA.code[k0 + i - 1] >> SH) + 1 : 0, to = (int)(c >> SH);
                for (int q = from; q <= to; ++q) bucket[q] = (uint16_t)i;
            }
            if (tid == 0) { bucket_steps = 0; for (int q = nk > 0 ? (int)(KA.code[k1 - 1] >> SH) + 1 : 0; q <= NB; ++q) bucket[q] = (uint16_t)nk; }
            __syncthreads();
            int widest = 0;
            for (int q = tid; q < NB; q += SP_ANCHOR_THREADS) { const int w = (int)bucket[q + 1] - (int)bucket[q]; widest = w > widest ? w : widest; }
            if (widest > 0) atomicMax(&bucket_steps, 32 - __builtin_clz((unsigned)widest));
        }
        tab_a = in_lds ? (int)a : -1;
        const uint32_t* kc = in_lds ? tab_code : KA.code + k0; const int32_t* kp = in_lds ? tab_pos : KA.pos + k0;
        __syncthreads();
        SP_AT(1);
        // SP_ANCHOR_ILP look-ups per thread run side by side: a lower bound over nk entries takes the same number of halvings for every k-mer,
        // so their dependent load chains overlap instead of queueing behind one another (one wave per SIMD here: nothing else would)
        const int steps = in_lds ? bucket_steps : nk > 0 ? 32 - __builtin_clz((unsigned)nk) : 0;
        for (int j0 = tid; j0 + SP_KMER <= n; j0 += SP_ANCHOR_THREADS * SP_ANCHOR_ILP) {
            uint32_t code[SP_ANCHOR_ILP]; int lo[SP_ANCHOR_ILP], hi[SP_ANCHOR_ILP], jj[SP_ANCHOR_ILP]; bool live[SP_ANCHOR_ILP];
#pragma unroll
            for (int u = 0; u < SP_ANCHOR_ILP; ++u) {
                const int j = j0 + u * SP_ANCHOR_THREADS;
                jj[u] = j;
                bool ok = j + SP_KMER <= n;
                const int w = ok ? j >> 4 : 0; const uint32_t sh = (uint32_t)((j & 15) << 1);
                if (ok && bn && __builtin_amdgcn_alignbit(bn[w + 1], bn[w], sh)) ok = false;
                code[u] = __builtin_amdgcn_alignbit(bw[w + 1], bw[w], sh);
                const int q = (int)(code[u] >> (32 - SP_ANCHOR_BUCKET_BITS));
                lo[u] = in_lds ? (int)bucket[q] : 0; hi[u] = !ok ? lo[u] : in_lds ? (int)bucket[q + 1] : nk; live[u] = ok;
            }
            for (int st = 0; st < steps; ++st) {
#pragma unroll
                for (int u = 0; u < SP_ANCHOR_ILP; ++u) {
                    const bool open = lo[u] < hi[u];
                    int mid = (lo[u] + hi[u]) >> 1; mid = mid < nk ? mid : nk - 1;
                    const bool less = kc[mid] < code[u];
                    lo[u] = open && less ? mid + 1 : lo[u];
                    hi[u] = open && !less ? mid : hi[u];
                }
            }
#pragma unroll
            for (int u = 0; u < SP_ANCHOR_ILP; ++u) {
                const int l = lo[u];
                int e = l;
                if (live[u]) while (e < nk && e - l <= SP_MAXOCC && kc[e] == code[u]) ++e;
                int occ = e - l;
                if (occ > SP_MAXOCC) occ = 0;
                // a read that crosses the gene puts thousands of votes on one diagonal, and neighbouring lanes hold neighbouring k-mers of it:
                // lanes whose single vote goes to the bin of the lane before them hand it to the first lane of their run, which adds the
                // run's count once (64 atomics on one LDS address would be carried out one after the other)
                sp_anchor_vote_run(lds, occ == 1 ? jj[u] - kp[l] + m : -1);
                if (occ > 1)
                    for (int y = l; y < e; ++y) {
                        const int bin = jj[u] - kp[y] + m;
                        atomicAdd(&lds[bin >> 1], (bin & 1) ? 0x10000u : 1u);
                    }
            }
        }
        SP_AT(2);
        __syncthreads();
        SP_AT(3);
        sp_anchor_peaks<SP_ANCHOR_THREADS>(lds, nbins, m, topk, p, diag_out, votes_out, red, spread);
        __syncthreads();
#ifdef SP_ANCHOR_TIMING
        SP_AT(4);
        if (tid == 0) { for (int k2 = 0; k2 < 4; ++k2) atomicAdd(&dbg[6 + k2], (unsigned long long)(tq[k2 + 1] - tq[k2])); atomicAdd(&dbg[10], 1ull); atomicAdd(&dbg[11], (unsigned long long)n); atomicAdd(&dbg[12], (unsigned long long)m); }
#endif
    }
}

// =============================================================================================
// generic WFA cells: wave w of block b takes cells (b*4 + w), (b*4 + w) + 4*gridDim.x, ...
// =============================================================================================
template <bool TRACE, bool HASN>
__global__ __launch_bounds__(256) void sp_cells_kernel(SeqSetView A, SeqSetView B, const CellDesc* __restrict__ cells, uint64_t n_cells,
                                                       sp_aln* __restrict__ out, uint32_t* __restrict__ events, uint32_t ev_stride,
                                                       uint16_t* __restrict__ hist_pool, int hist_rows, int slot_words) {
    extern __shared__ uint32_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* slot = lds + wave * slot_words;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    uint16_t* hist = TRACE ? hist_pool + gw * (uint64_t)hist_rows * SP_WAVE : nullptr;
    for (uint64_t c = gw; c < n_cells; c += nw) {
        const CellDesc cd = cells[c];
        spw::CellIn in;
        in.a_words = A.words + A.word_off[cd.a]; in.a_nplane = A.nplane ? A.nplane + A.word_off[cd.a] : nullptr;
        in.a0 = 0; in.a1 = A.len[cd.a];
        in.b_words = B.words + B.word_off[cd.b]; in.b_nplane = B.nplane ? B.nplane + B.word_off[cd.b] : nullptr;
        const int blen = B.len[cd.b];
        in.b0 = cd.b_hi >= 0 ? cd.b_lo : 0; in.b1 = cd.b_hi >= 0 ? cd.b_hi : blen;
        in.diag = cd.diag;
        in.max_ed = cd.max_ed < hist_rows - 1 ? cd.max_ed : hist_rows - 1;
        spw::CellOut o;
        uint32_t* ev = (TRACE && events) ? events + c * (uint64_t)ev_stride : nullptr;
        if (TRACE && ev && (uint32_t)in.max_ed > ev_stride) in.max_ed = (int)ev_stride;
        if (cd.diag == SP_NO_DIAG) { o.ok = 0; o.nm = 0; o.a_start = o.a_end = o.b_start = o.b_end = 0; }
        else spw::wfa_cell<TRACE, HASN>(in, slot, slot_words, lane, hist, ev, o);
        if (lane == 0) {
            sp_aln r; r.ok = o.ok; r.nm = o.nm; r.a_start = o.a_start; r.a_end = o.a_end;
            r.b_start = o.b_start; r.b_end = o.b_end; r.a_len = in.a1 - in.a0; r.b_len = in.b1 - in.b0;
            out[c] = r;
        }
    }
}


// =============================================================================================
// wide retry: a cell that found no alignment on 64 diagonals within its edit cap is run again on 256 diagonals around the same anchor
// (an insertion / deletion of 40-120 bases next to the anchor; oracle/align.c: osp_wfa_retry).  Same contract as the 64-diagonal core
// (ends-free, unit costs, X > D > I, longest path / most central / lowest diagonal at the end), four diagonals per lane; the windows
// are read from memory (this is the rare path: the launcher applies it where a lost cell matters, not in K1 and K3).
// =============================================================================================
constexpr int SP_WIDE = 256;      // diagonals
constexpr int SP_WPL = 4;         // per lane

__device__ __forceinline__ uint32_t wide_get16(const uint32_t* w, int p) { return __builtin_amdgcn_alignbit(w[(p >> 4) + 1], w[p >> 4], (uint32_t)(p & 15) << 1); }

__device__ __forceinline__ int wide_extend(const uint32_t* aw, const uint32_t* an, int m, const uint32_t* bw, const uint32_t* bn, int b0, int n, int i, int k) {
    while (i < m && i + k < n) {
        const int pa = i, pb = b0 + i + k;
        const uint32_t x = wide_get16(aw, pa) ^ wide_get16(bw, pb);
        uint32_t mm = (x | (x >> 1)) & 0x55555555u;
        if (an) mm |= wide_get16(an, pa) & 0x55555555u;
        if (bn) mm |= wide_get16(bn, pb) & 0x55555555u;
        int run = mm ? (__builtin_ctz(mm) >> 1) : 16;
        const int l1 = m - i, l2 = n - i - k, lim = l1 < l2 ? l1 : l2;
        run = run < lim ? run : lim;
        i += run;
        if (run < 16) break;
    }
    return i;
}

template <bool TRACE>
__global__ __launch_bounds__(256) void sp_cells_wide_kernel(SeqSetView A, SeqSetView B, const CellDesc* __restrict__ cells, uint64_t n_cells,
                                                            sp_aln* __restrict__ out, uint32_t* __restrict__ events, uint32_t ev_stride,
                                                            uint16_t* __restrict__ hist_pool, int hist_rows, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    uint16_t* hist = TRACE ? hist_pool + gw * (uint64_t)hist_rows * SP_WIDE : nullptr;
    for (uint64_t c = gw; c < n_cells; c += nw) {
        const CellDesc cd = cells[c];
        if (cd.diag == SP_NO_DIAG) continue;
        const sp_aln narrow = out[c];
        if (narrow.ok && (mode < 2 || narrow.nm <= SP_BAND / 2)) continue;     // the 64-diagonal run found it (cheaply enough)
        const uint32_t* aw = A.words + A.word_off[cd.a]; const uint32_t* an = A.nplane ? A.nplane + A.word_off[cd.a] : nullptr;
        const uint32_t* bw = B.words + B.word_off[cd.b]; const uint32_t* bn = B.nplane ? B.nplane + B.word_off[cd.b] : nullptr;
        const int m = A.len[cd.a], blen = B.len[cd.b];
        const int b0 = cd.b_hi >= 0 ? cd.b_lo : 0, n = (cd.b_hi >= 0 ? cd.b_hi : blen) - b0;
        int max_ed = cd.max_ed < hist_rows - 1 ? cd.max_ed : hist_rows - 1;
        uint32_t* ev = (TRACE && events) ? events + c * (uint64_t)ev_stride : nullptr;
        if (TRACE && ev && (uint32_t)max_ed > ev_stride) max_ed = (int)ev_stride;
        if (m <= 0 || n <= 0 || max_ed < 0) continue;
        const int kb = cd.diag - SP_WIDE / 2;
        int H[SP_WPL], O[SP_WPL];
#pragma unroll
        for (int j = 0; j < SP_WPL; ++j) {
            const int D = lane * SP_WPL + j, k = kb + D, i0 = k < 0 ? -k : 0;
            H[j] = (i0 < m && i0 + k < n) ? wide_extend(aw, an, m, bw, bn, b0, n, i0, k) : SP_NEG;
            O[j] = D;
            if (TRACE) hist[D] = (uint16_t)(H[j] >= 0 ? H[j] : 0xFFFF);
        }
        int s = 0, end_D = -1;
        for (;;) {
            // termination: a diagonal on the last row / last column; longest path, then most central, then lowest diagonal
            long long key = -1;
#pragma unroll
            for (int j = 0; j < SP_WPL; ++j) {
                const int D = lane * SP_WPL + j, k = kb + D, i = H[j];
                if (i >= 0 && (i == m || i + k == n)) {
                    int cdist = D - SP_WIDE / 2; cdist = cdist < 0 ? -cdist : cdist;
                    const long long kk = (long long)(i + i + k) * (2ll * SP_WIDE * SP_WIDE) + (long long)(SP_WIDE - cdist) * SP_WIDE + (SP_WIDE - 1 - D);
                    key = kk > key ? kk : key;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const long long other = __shfl_xor(key, o); key = other > key ? other : key; }
            if (key >= 0) { end_D = SP_WIDE - 1 - (int)(key % SP_WIDE); break; }
            if (s == max_ed) break;
            // next wavefront: X from the same diagonal, D from the one below, I from the one above (that priority on equal reach)
            const int from_below = spw::from_lower(H[SP_WPL - 1], SP_NEG), ofb = spw::from_lower(O[SP_WPL - 1], 0);
            const int from_above = spw::from_upper(H[0], SP_NEG), ofa = spw::from_upper(O[0], 0);
            int NH[SP_WPL], NO[SP_WPL];
#pragma unroll
            for (int j = 0; j < SP_WPL; ++j) {
                const int up = j > 0 ? H[j - 1] : from_below, oup = j > 0 ? O[j - 1] : ofb;
                const int dn = j < SP_WPL - 1 ? H[j + 1] : from_above, odn = j < SP_WPL - 1 ? O[j + 1] : ofa;
                int best = H[j] >= 0 ? H[j] + 1 : SP_NEG, o = O[j];
                if (up >= 0 && up > best) { best = up; o = oup; }
                if (dn >= 0 && dn + 1 > best) { best = dn + 1; o = odn; }
                NH[j] = best; NO[j] = o;
            }
            ++s;
#pragma unroll
            for (int j = 0; j < SP_WPL; ++j) {
                const int D = lane * SP_WPL + j;
                H[j] = NH[j] >= 0 ? wide_extend(aw, an, m, bw, bn, b0, n, NH[j], kb + D) : SP_NEG;
                O[j] = NO[j];
                if (TRACE) hist[(size_t)s * SP_WIDE + D] = (uint16_t)(H[j] >= 0 ? H[j] : 0xFFFF);
            }
        }
        if (end_D < 0) continue;                                            // still nothing: the cell stays as it was
        if (narrow.ok && s >= narrow.nm) continue;                          // the narrow alignment is as cheap: it stands
        // the end point and the origin diagonal live in the lane that owns end_D
        const int el = end_D / SP_WPL, ej = end_D % SP_WPL;
        int he = 0, oe = 0;
#pragma unroll
        for (int j = 0; j < SP_WPL; ++j) if (j == ej) { he = __builtin_amdgcn_readlane(H[j], el); oe = __builtin_amdgcn_readlane(O[j], el); }
        // (readlane needs a uniform lane index: el is uniform)
        const int ko = kb + oe, i0 = ko < 0 ? -ko : 0;
        if (lane == 0) {
            sp_aln r; r.ok = 1; r.nm = s; r.a_start = i0; r.a_end = he; r.b_start = i0 + ko; r.b_end = he + kb + end_D; r.a_len = m; r.b_len = n;
            out[c] = r;
            if (TRACE && ev) {
                __threadfence();
                int l = end_D;
                for (int t = s; t > 0; --t) {
                    const uint16_t* Hp = hist + (size_t)(t - 1) * SP_WIDE;
                    auto hv = [&](int d) { return (d < 0 || d >= SP_WIDE || Hp[d] == 0xFFFF) ? SP_NEG : (int)Hp[d]; };
                    const int cc = hv(l), uu = hv(l - 1), dd = hv(l + 1);
                    int best = cc >= 0 ? cc + 1 : SP_NEG, src = l; uint32_t type = SP_EV_X;
                    if (uu >= 0 && uu > best) { best = uu; src = l - 1; type = SP_EV_D; }
                    if (dd >= 0 && dd + 1 > best) { best = dd + 1; src = l + 1; type = SP_EV_I; }
                    const int kk = kb + l;
                    const int bpos = type == SP_EV_X ? best - 1 + kk : (type == SP_EV_D ? best + kk - 1 : best + kk);
                    ev[t - 1] = (type << 30) | (uint32_t)bpos;
                    l = src;
                }
            }
        }
        if (TRACE) { __threadfence(); }
    }
}

// =============================================================================================
// 2-bit packing on the device: one thread per output dword (16 bases).  A=0 C=1 G=2 T=3 (either case), anything else is
// 'N': code 0 in the word plane, 01 in the N plane, and the set is flagged as carrying N.
// =============================================================================================
__global__ __launch_bounds__(256) void sp_pack_kernel(const char* __restrict__ ascii, const uint64_t* __restrict__ off, const uint64_t* __restrict__ word_off,
                                                      const int32_t* __restrict__ len, uint32_t n, uint32_t* __restrict__ words, uint32_t* __restrict__ nplane,
                                                      uint32_t* __restrict__ flag) {
    for (uint32_t s = blockIdx.x; s < n; s += gridDim.x) {
        const int L = len[s]; const char* src = ascii + off[s];
        const int nw = (L + 15) >> 4;
        for (int w = threadIdx.x; w < nw; w += blockDim.x) {
            uint32_t word = 0, nw_bits = 0;
            const int b0 = w << 4, b1 = b0 + 16 < L ? b0 + 16 : L;
            for (int b = b0; b < b1; ++b) {
                uint32_t c;
                switch (src[b]) {
                    case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break;
                    case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break;
                    default: c = 0; nw_bits |= 1u << ((b & 15) << 1);
                }
                word |= c << ((b & 15) << 1);
            }
            if (words) words[word_off[s] + w] = word;
            if (nplane) nplane[word_off[s] + w] = nw_bits;
            if (nw_bits && flag) atomicOr(flag, 1u);
        }
    }
}

// BAM's SEQ field as stored (SAMv1 4.2: two bases per byte, high nibble first, "=ACMGRSVTWYHKDBN"; every sequence starts on a byte):
// 1 / 2 / 4 / 8 are A / C / G / T, every other code is 'N'.  One thread per output dword = 8 input bytes.
__global__ __launch_bounds__(256) void sp_pack4_kernel(const uint8_t* __restrict__ seq4, const uint64_t* __restrict__ off, const uint64_t* __restrict__ word_off,
                                                       const int32_t* __restrict__ len, uint32_t n, uint32_t* __restrict__ words, uint32_t* __restrict__ nplane,
                                                       uint32_t* __restrict__ flag) {
    for (uint32_t s = blockIdx.x; s < n; s += gridDim.x) {
        const int L = len[s]; const uint8_t* src = seq4 + off[s];
        const int nw = (L + 15) >> 4;
        const int nbytes = (L + 1) >> 1;
        for (int w = threadIdx.x; w < nw; w += blockDim.x) {
            // the word's eight input bytes in two loads, all sixteen nibbles decoded side by side: a code has ONE bit set (A 1, C 2, G 4, T 8): its 2-bit value is
            // (bit 1 | bit 3, bit 2 | bit 3); any other number of set bits is an N.  (A byte load and a compare chain per base was 2.1 ms per 10,000-read sample.)
            const int y0 = w << 3;
            uint32_t half[2] = { 0u, 0u };
            if (y0 + 8 <= nbytes) __builtin_memcpy(half, src + y0, 8);
            else for (int k = 0; y0 + k < nbytes; ++k) half[k >> 2] |= (uint32_t)src[y0 + k] << ((k & 3) << 3);
            uint32_t word = 0, nw_bits = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t v = half[h], M = 0x11111111u;
                const uint32_t p0 = v & M, p1 = (v >> 1) & M, p2 = (v >> 2) & M, p3 = (v >> 3) & M;
                const uint32_t lo = p1 | p3, hi = p2 | p3, odd = (p0 + p1 + p2 + p3) ^ M;          // odd: a nibble is 0 where exactly one bit was set
                const uint32_t inv = (odd | (odd >> 1) | (odd >> 2)) & M;
                const uint32_t code = (lo | (hi << 1)) & ~(inv * 3u);                                // 2-bit code in the low bits of every nibble, 0 for an N
#pragma unroll
                for (int k = 0; k < 4; ++k) {                                                        // byte k: its high nibble is the earlier base
                    const int at = (h << 4) + (k << 2);
                    word |= ((code >> ((k << 3) + 4)) & 3u) << at | ((code >> (k << 3)) & 3u) << (at + 2);
                    nw_bits |= ((inv >> ((k << 3) + 4)) & 1u) << at | ((inv >> (k << 3)) & 1u) << (at + 2);
                }
            }
            const int left = L - (w << 4);
            if (left < 16) { const uint32_t keep = (1u << (left << 1)) - 1u; word &= keep; nw_bits &= keep; }
            if (words) words[word_off[s] + w] = word;
            if (nplane) nplane[word_off[s] + w] = nw_bits;
            if (nw_bits && flag) atomicOr(flag, 1u);
        }
    }
}

// 2 bits per base already (four bases per byte, base b in bits 2 (b & 3) of byte b >> 2, every sequence starts on a byte, A C G T only):
// a dword of the set is four input bytes, little end first; only the starts move (16-byte aligned in the set) and the tail is masked
__global__ __launch_bounds__(256) void sp_pack2_kernel(const uint8_t* __restrict__ seq2, const uint64_t* __restrict__ off, const uint64_t* __restrict__ word_off,
                                                       const int32_t* __restrict__ len, uint32_t n, uint32_t* __restrict__ words) {
    for (uint32_t s = blockIdx.x; s < n; s += gridDim.x) {
        const int L = len[s]; const uint8_t* src = seq2 + off[s];
        const int nw = (L + 15) >> 4, nbytes = (L + 3) >> 2;
        for (int w = threadIdx.x; w < nw; w += blockDim.x) {
            uint32_t word = 0;
            for (int k = 0; k < 4; ++k) { const int y = 4 * w + k; if (y < nbytes) word |= (uint32_t)src[y] << (8 * k); }
            const int left = L - (w << 4);
            if (left < 16) word &= (1u << (2 * left)) - 1u;
            words[word_off[s] + w] = word;
        }
    }
}

int sp_launch_pack_on(hipStream_t stream, int num_cus, int format, const void* d_src, const uint64_t* d_off, const uint64_t* d_word_off, const int32_t* d_len, uint32_t n,
                      uint32_t* d_words, uint32_t* d_nplane, uint32_t* d_flag) {
    if (n == 0) return SP_OK;
    const unsigned grid = std::min<unsigned>(n, (unsigned)num_cus * 16);
    if (format == SP_SEQ_ASCII) hipLaunchKernelGGL(sp_pack_kernel, dim3(grid), dim3(256), 0, stream, (const char*)d_src, d_off, d_word_off, d_len, n, d_words, d_nplane, d_flag);
    else if (format == SP_SEQ_BAM4) hipLaunchKernelGGL(sp_pack4_kernel, dim3(grid), dim3(256), 0, stream, (const uint8_t*)d_src, d_off, d_word_off, d_len, n, d_words, d_nplane, d_flag);
    else if (d_words) hipLaunchKernelGGL(sp_pack2_kernel, dim3(grid), dim3(256), 0, stream, (const uint8_t*)d_src, d_off, d_word_off, d_len, n, d_words);
    return hipGetLastError() == hipSuccess ? SP_OK : SP_ERR_HIP;
}

int sp_launch_pack(sp_ctx* ctx, const char* d_ascii, const uint64_t* d_off, const uint64_t* d_word_off, const int32_t* d_len, uint32_t n,
                   uint32_t* d_words, uint32_t* d_nplane, uint32_t* d_flag) {
    if (sp_launch_pack_on(ctx->stream, ctx->num_cus, SP_SEQ_ASCII, d_ascii, d_off, d_word_off, d_len, n, d_words, d_nplane, d_flag) != SP_OK)
        return sp_fail(ctx, SP_ERR_HIP, "pack kernel launch");
    return SP_OK;
}

// =============================================================================================
// host side
// =============================================================================================
int sp_fail(sp_ctx* ctx, int code, const std::string& msg) { if (ctx) ctx->err = msg; return code; }

void* sp_scratch(sp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return ctx->scratch;
    (void)hipSetDevice(ctx->device);       // the current device is per host thread: a helper thread's first HIP call may be this allocation
    if (ctx->scratch) { hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    size_t want = bytes + bytes / 4;
    if (hipMalloc(&ctx->scratch, want) != hipSuccess) { ctx->scratch = nullptr; return nullptr; }
    ctx->scratch_bytes = want;
    return ctx->scratch;
}

void* sp_pool(sp_ctx* ctx, const char* name, size_t bytes) {
    auto& e = ctx->pool[name];
    if (bytes <= e.second && e.first) return e.first;
    (void)hipSetDevice(ctx->device);       // (as in sp_scratch: never allocate on whatever device the calling thread happens to have current)
    if (e.first) { (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&e.first, want) != hipSuccess) { e.first = nullptr; return nullptr; }
    e.second = want;
    return e.first;
}

unsigned long long* sp_counters(sp_ctx* ctx) {
    auto it = ctx->pool.find("prof_counters");
    if (it != ctx->pool.end() && it->second.first) return (unsigned long long*)it->second.first;
    unsigned long long* c = (unsigned long long*)sp_pool(ctx, "prof_counters", SPC_N * sizeof(unsigned long long));
    if (c) (void)hipMemsetAsync(c, 0, SPC_N * sizeof(unsigned long long), ctx->stream);
    return c;
}

// pinned host memory for results: a device-to-host copy into pageable caller memory goes through the runtime's bounce buffers in
// pieces (0.25 ms for the 0.8 MB of a K1 batch); into pinned memory it is one DMA, and the memcpy to the caller is ~40 us
void* sp_host_pool(sp_ctx* ctx, const char* name, size_t bytes) {
    auto& e = ctx->host_pool[name];
    if (bytes <= e.second && e.first) return e.first;
    (void)hipSetDevice(ctx->device);
    if (e.first) { (void)hipHostFree(e.first); e.first = nullptr; e.second = 0; }
    size_t want = bytes + bytes / 8 + 256;
    if (hipHostMalloc(&e.first, want, hipHostMallocDefault) != hipSuccess) { e.first = nullptr; return nullptr; }
    e.second = want;
    return e.first;
}

// only the wide-band pass of sp_launch_cells (untraced): every listed cell whose entry of d_out is not found yet runs on 256 diagonals; cells without a diagonal are skipped
int sp_launch_cells_wide(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B, const CellDesc* d_cells, uint64_t n_cells, sp_aln* d_out) {
    if (n_cells == 0) return SP_OK;
    const uint64_t wblocks = std::min<uint64_t>((n_cells + 3) / 4, (uint64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL((sp_cells_wide_kernel<false>), dim3((unsigned)wblocks), dim3(256), 0, ctx->stream, A->view(), B->view(), d_cells, n_cells, d_out, (uint32_t*)nullptr, 0u,
                       (uint16_t*)nullptr, SP_MAX_ED + 1, 1);
    SP_HIP_CHECK(ctx, hipGetLastError());
    return SP_OK;
}

int sp_slot_words(const sp_seqset* A, const sp_seqset* B, bool hasn) {
    int mn = std::min(A->max_len, B->max_len);
    int w = (mn + SP_BAND + 30) / 16 + 3;
    return (hasn ? 4 : 2) * w;
}

int sp_launch_anchor(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                     const uint32_t* d_a_idx, const uint32_t* d_b_idx, uint64_t n_pairs,
                     int32_t* d_diag, int32_t* d_votes, int topk, const char* prof_name, uint32_t a_period) {
    if (topk < 1 || topk > 8) return sp_fail(ctx, SP_ERR_INVALID_ARG, "anchor: topk must be 1..8");
    if (n_pairs == 0) return SP_OK;
    if (!A->has_index) return sp_fail(ctx, SP_ERR_INVALID_ARG, "anchor: set A has no k-mer index");
    int bins_cap = A->max_len + B->max_len + 1;
    size_t lds_bytes = (size_t)((bins_cap + 1) / 2) * 4;
    if (lds_bytes > 160 * 1024 - 64) return sp_fail(ctx, SP_ERR_TOO_LONG, "anchor: sequences too long for the LDS vote histogram");
    // A's 16-mer table rides along in LDS when it fits (8 bytes per k-mer)
    int tab_cap = A->max_len;
    if (lds_bytes + (size_t)tab_cap * 8 > 160 * 1024 - 64) tab_cap = 0;
    lds_bytes += (size_t)tab_cap * 8;
    SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)sp_anchor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    uint64_t grid = std::min<uint64_t>(n_pairs, (uint64_t)ctx->num_cus * 8);       // 1 / 2 / 4 / 8 / 16 per CU: 3.3 / 2.04 / 2.0 / 1.9 / 1.96 ms of anchors per bench step
    if (a_period > 1 && grid > a_period) grid -= grid % a_period;           // (39 templates: a workgroup's stride of 2,048 pairs changed A every time, 72 KB of table per pair)
    ProfScope ps(ctx, prof_name, n_pairs);
    hipLaunchKernelGGL(sp_anchor_kernel, dim3((unsigned)grid), dim3(SP_ANCHOR_THREADS), lds_bytes, ctx->stream,
                       A->view(), A->kview(), B->view(), d_a_idx, d_b_idx, n_pairs, d_diag, d_votes, bins_cap, topk, tab_cap, sp_counters(ctx));
    SP_HIP_CHECK(ctx, hipGetLastError());
    return SP_OK;
}

#ifndef SP_TRACE_BLOCKS_PER_CU
#define SP_TRACE_BLOCKS_PER_CU 16     // traced cells: every wave owns 32 KB of wavefront history in the scratch pool (512 MB at 16 workgroups per CU; 4: 0.93 ms, 8: 0.87 ms, 16: 0.74 ms for the two K2 levels of the bench step)
#endif
int sp_launch_cells(sp_ctx* ctx, const sp_seqset* A, const sp_seqset* B,
                    const CellDesc* d_cells, uint64_t n_cells,
                    sp_aln* d_out, uint32_t* d_events, uint32_t events_stride, const char* prof_name, int retry_wide) {
    if (n_cells == 0) return SP_OK;
    const bool trace = d_events != nullptr;
    const bool hasn = A->has_n || B->has_n;
    const int slot_words = sp_slot_words(A, B, hasn);
    const size_t lds_bytes = (size_t)slot_words * 4 * 4 + SP_LDS_TAIL;
    if (lds_bytes > 160 * 1024 - 64) return sp_fail(ctx, SP_ERR_TOO_LONG, "align: sequences too long for the LDS window");
    uint64_t blocks = (n_cells + 3) / 4;
    const uint64_t max_blocks = trace ? (uint64_t)ctx->num_cus * SP_TRACE_BLOCKS_PER_CU : (uint64_t)ctx->num_cus * 16;
    if (blocks > max_blocks) blocks = max_blocks;
    const int hist_rows = SP_MAX_ED + 1;
    uint16_t* hist = nullptr;
    if (trace) {
        hist = (uint16_t*)sp_scratch(ctx, blocks * 4 * (size_t)hist_rows * SP_WAVE * sizeof(uint16_t));
        if (!hist) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "align: traceback scratch");
    }
    SeqSetView av = A->view(), bv = B->view();
    ProfScope ps(ctx, prof_name, n_cells);
#define SP_LAUNCH(T, N) do { \
        SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)sp_cells_kernel<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        hipLaunchKernelGGL((sp_cells_kernel<T, N>), dim3((unsigned)blocks), dim3(256), lds_bytes, ctx->stream, \
                           av, bv, d_cells, n_cells, d_out, d_events, events_stride, hist, hist_rows, slot_words); } while (0)
    if (trace) { if (hasn) SP_LAUNCH(true, true); else SP_LAUNCH(true, false); }
    else       { if (hasn) SP_LAUNCH(false, true); else SP_LAUNCH(false, false); }
#undef SP_LAUNCH
    SP_HIP_CHECK(ctx, hipGetLastError());
    if (retry_wide) {
        // the cells the 64-diagonal run lost are run again on 256 diagonals (one wave each; cells that were found return at once)
        // (untraced: enough workgroups that a wave meets a lost cell or two, not six -- a lost cell is hundreds of rounds of one wave, and the launch took as long as its
        //  unluckiest wave: 1.7 ms for the weights' placements of a 2,000-read CYP2D6 sample at two workgroups per CU; traced runs keep that size: every wave owns history scratch)
        uint64_t wblocks = std::min<uint64_t>((n_cells + 3) / 4, (uint64_t)ctx->num_cus * (trace ? 2 : 8));
        uint16_t* whist = nullptr;
        if (trace) {
            whist = (uint16_t*)sp_pool(ctx, "wide_hist", wblocks * 4 * (size_t)hist_rows * SP_WIDE * sizeof(uint16_t));
            if (!whist) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "align: wide traceback scratch");
        }
        if (trace) hipLaunchKernelGGL((sp_cells_wide_kernel<true>), dim3((unsigned)wblocks), dim3(256), 0, ctx->stream, av, bv, d_cells, n_cells, d_out, d_events, events_stride, whist, hist_rows, retry_wide);
        else hipLaunchKernelGGL((sp_cells_wide_kernel<false>), dim3((unsigned)wblocks), dim3(256), 0, ctx->stream, av, bv, d_cells, n_cells, d_out, d_events, events_stride, whist, hist_rows, retry_wide);
        SP_HIP_CHECK(ctx, hipGetLastError());
    }
    return SP_OK;
}
