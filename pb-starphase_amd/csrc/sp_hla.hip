// sp_hla.hip -- HLA hot path on gfx950.
//   K1  sp_hla_realign_reads    replaces HlaRealigner::realign_record   (src/hla/realigner.rs:98-350)
//   K2  sp_hla_score_consensus  replaces score_read's allele loop + HlaProcessedMatch
//                               (src/hla/caller.rs:1411-1510, src/hla/processed_match.rs:53-263)
// Decisions use f64 exactly as the reference does (MappingScore::score_value, src/data_types/mapping.rs:191-195);
// alignments come from the one-wavefront-per-cell WFA kernel (sp_wfa.hip.h).
#include "sp_internal.h"
#include <tuple>
#include <mutex>
#include "sp_wfa.hip.h"
#include "sp_anchor.hip.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <new>

int sp_seqset_build_index(sp_ctx* ctx, sp_seqset* s);
int sp_slot_words(const sp_seqset* A, const sp_seqset* B, bool hasn);

#define K1_CHUNK       64       // visiting-order positions of one k1_cells workgroup (= one wavefront; 128 and 256 per workgroup measured slower)
#define K1_MIN_VOTES   16

// the alleles by position of the K1 visiting order (structure of arrays, one coalesced load per field and group of 64 positions)
struct K1Positions {
    const int32_t* alen;      // allele length, 0 = no DNA sequence
    const int32_t* off;       // frame offset (allele_fwd_pos - ref_fwd_pos), SP_NO_DIAG = no anchor
    const uint32_t* gene;
    const uint32_t* woff;     // first packed word of the allele in dna_fwd
    const int32_t* lcp;       // prefix shared with the position before (same gene and frame offset), else 0
};
#define K2_MIN_VOTES   2
#define K2_MAX_ED      255     // edit cap of a K2 / typing cell (allele vs consensus, consensus vs gene reference): 6 % of the longest bundled allele;
                               // it sizes the per-cell event rows, so it stays below the library-wide SP_MAX_ED

struct sp_hla_db {
    sp_ctx* ctx = nullptr;
    uint32_t n_alleles = 0, n_genes = 0;
    int ref_buffer = 100;
    std::vector<uint32_t> gene_of;
    std::vector<uint8_t>  gene_fwd, has_dna;
    std::vector<std::vector<uint32_t>> gene_alleles;      // allele indices per gene (database order)
    std::vector<uint32_t> exon_off; std::vector<int32_t> exon_start, exon_end;   // per gene, relative to the buffered reference
    sp_seqset* dna_gene = nullptr;    // allele DNA as stored (gene strand); len 0 = none
    sp_seqset* cdna_gene = nullptr;   // allele cDNA as stored
    sp_seqset* dna_fwd = nullptr;     // allele DNA in hg38 orientation (create_hla_fasta, realigner.rs:497-526)
    sp_seqset* ref_fwd = nullptr;     // buffered gene references, hg38 forward (realigner.rs:74-81)
    sp_seqset* ref_rev = nullptr;     // their reverse complements: a read that anchors better there has its best mapping on the reverse strand (realigner.rs:178-193)
    uint32_t* d_gene_of = nullptr;
    int32_t*  d_off_fwd = nullptr;    // allele_fwd_pos - ref_fwd_pos (SP_NO_DIAG = no anchor)
    std::vector<uint32_t> h_order;
    // allowed alleles of a gene in database order (is_allowed_allele_def), on the host and on the device, built on first use
    struct GeneList { std::vector<uint32_t> idx; uint32_t* d_idx = nullptr; uint32_t* d_l0 = nullptr; uint32_t* d_l1 = nullptr; };
    mutable std::map<uint32_t, GeneList> gene_lists;       // key = gene * 2 + require_dna
    uint32_t* d_order = nullptr;      // K1 visits the alleles sorted by (gene, frame offset, hg38-strand sequence) ...
    int32_t*  d_lcp = nullptr;        // ... d_lcp[i] = common prefix of order[i-1] and order[i] when they share gene and frame offset, else 0
    int32_t*  d_pos = nullptr;        // 5 x n_alleles: the K1Positions arrays
    int32_t*  d_am = nullptr;         // n_alleles*3: ok, am.query_start, am.target_start (allele -> gene ref, realigner.rs:289-310)
    int32_t*  d_hpc_ref = nullptr;    // hpc_pos(ref_fwd[g], p) for p in 0..len, concatenated
    uint64_t* d_hpc_ref_off = nullptr;
    mutable K1Seed* seed = nullptr;   // minimizer index of dna_fwd for the seeded K1 (sp_hla_seed.hip), built at the first seeded call
    mutable K2Dict kdict[2];          // 16-mer dictionaries of cdna_gene / dna_gene, built at the first K2 call (sp_hla_dict.hip)
    mutable std::mutex lazy;          // guards gene_lists and kdict: two contexts may type consensuses on one database at the same time
};

// ---------------------------------------------------------------------------------------------
// f64 score algebra (src/data_types/mapping.rs:60-84,191-195)
__device__ __forceinline__ double score_value(int len, int nm, int unmapped) {
    double num = (double)(nm + unmapped);
    if (num < 0.1) num = 0.1;
    return num / (double)len;
}

// =============================================================================================
// K1 cells: workgroup = one wavefront = (read r, K1_CHUNK positions of the visiting order).  The read window every cell of the
// chunk can touch is staged once into LDS; the alleles stream through one LDS slot, one WFA cell each.
// cell_out[r * n_alleles + p] = (nm << 16) | aligned allele span, or SP_CELL_NONE, p = position of the allele in the visiting order
// (a workgroup's results are one contiguous stretch of the row; the reduce maps positions back to allele indices).
//
// Exact branch-and-bound (bound != nullptr): bound[r] holds (num10 << 32 | span) of the best ACCEPTABLE cell
// finished so far for read r (num10 = 10*nm, or 1 when nm == 0, i.e. max(nm, 0.1) * 10).  A cell whose edit
// count s already satisfies s / allele_len > best ratio can never win (its span <= allele_len), so its edit cap
// is lowered to floor(num10 * allele_len / (10 * span)).  Cells that tie the best ratio are never cut, so the
// lowest-index tie rule of the acceptance loop (realigner.rs:139-141) is preserved.
// =============================================================================================
#define K1_NO_BOUND 0xFFFFFFFF00000001ull

__device__ __forceinline__ int k1_dyn_cap(unsigned long long b, int alen, int cap) {
    const uint32_t nb10 = (uint32_t)(b >> 32), sb = (uint32_t)b;
    if (nb10 == 0xFFFFFFFFu) return cap;
    // floor(T / D) with T = nb10 * alen < 2^28 and D = 10 * span: float estimate, then an exact +-1 fix-up
    const uint32_t T = nb10 * (uint32_t)alen, D = 10u * sb;
    uint32_t q = (uint32_t)(__fdividef((float)T, (float)D));
    if (q * D > T) --q;
    if ((q + 1) * D <= T) ++q;
    if (q * D > T) --q;
    return q < (uint32_t)cap ? (int)q : cap;
}

#ifndef K1_CAP1
#define K1_CAP1 12       // edit cap of the first deepening pass
#endif
#define K1_PRE_WORDS 320 // words of an allele the register prefetch covers

// DEEP only names the later (deeper, much smaller) passes of the iterative deepening differently, so that profilers list them apart
// One wavefront = one workgroup = (read, 64 consecutive positions of the visiting order).  Lane j IS cell j:
// its metadata stays in that lane's registers, its result too, and whatever is per cell but not part of the DP -- the edit cap under
// the current bound, the scan for the next cell that has to run -- is one vector operation over the group.  Nothing waits for another
// wave (workgroups of 2, 4, 8 waves measured 13.3 / 15.5 / 19.9 ms against 12.6 ms: every wave idles until the slowest is done).
//
// Prefix sharing (exact): the positions follow the database's visiting order, so neighbours mostly start with the same bases.
// A cell that runs out of edits has looked at A[0 .. explored] only (wfa_core); if the next cell's allele shares that prefix
// (same gene, same frame offset => same diagonal, same read window), is allowed no more edits than the run had, and sits right
// behind it in the order, its run would be the same run cut at the same or an earlier step: it fails too and is not executed.
// The chain carries on from a skipped cell with the executed run's extent and the smaller cap (caps along a chain never grow).
// Any bound value ever observed is a valid one (bounds only tighten), so the caps of a group may come from one read of it.
// eight waves per SIMD (63 registers, no scratch) instead of the seven the compiler's own 72 registers allow: the rounds of a cell are chains of dependent DPP / ballot /
// branch steps, and one more wave to switch to is worth 3 % of the launch (9.86 -> 9.53 ms; six and five waves: 10.4 and 11.4)
#ifndef K1_WAVES_PER_EU
#define K1_WAVES_PER_EU 8
#endif
#define K1_OCC __attribute__((amdgpu_waves_per_eu(K1_WAVES_PER_EU, K1_WAVES_PER_EU)))
template <bool HASN, bool DEEP>
__global__ __launch_bounds__(64) K1_OCC void k1_cells_kernel(SeqSetView alleles, SeqSetView reads, K1Positions pos,
                                                      const int32_t* __restrict__ d_rg, const int32_t* __restrict__ votes_rg,
                                                      int n_genes, uint32_t n_alleles, uint32_t n_chunks,
                                                      uint32_t* __restrict__ cell_out, unsigned long long* __restrict__ bound,
                                                      const uint32_t* __restrict__ read_list, uint32_t* __restrict__ read_maxlen,
                                                      unsigned long long* __restrict__ winner, const uint32_t* __restrict__ order,
                                                      int pass_cap, int b_words, int a_words, uint2* __restrict__ wg_stats) {
    extern __shared__ uint32_t lds[];
    // layout: [B window b_words (x2 with N plane)][A slot a_words (x2 with N plane)]
    uint32_t* LB = lds;
    uint32_t* NB = HASN ? LB + b_words : nullptr;
    uint32_t* LA = LB + (HASN ? 2 : 1) * b_words;
    uint32_t* NA = HASN ? LA + a_words : nullptr;
    const int lane = threadIdx.x;
    // deeper passes only visit the reads an earlier (shallower) pass could not settle
    // workgroups are dealt to the 8 XCDs round-robin by index: index % 8 picks the chunk's residue class, so one XCD only ever sees
    // 1/8 of the database positions and keeps those alleles in its own L2 (n_chunks is the padded count, a multiple of 8)
    const uint32_t per_read = n_chunks >> 3, j = blockIdx.x >> 3;
    const uint32_t slot = j / per_read, chunk = (j % per_read) * 8 + (blockIdx.x & 7);
    const uint32_t r = read_list ? read_list[slot] : slot;
    if (chunk * (uint32_t)K1_CHUNK >= n_alleles) return;
    // gene filter: a gene is searched when it has >= K1_MIN_VOTES and >= 1/10 of the read's best gene
    int vmax = 0;
    for (int g = 0; g < n_genes; ++g) { int v = votes_rg[(uint64_t)r * n_genes + g]; vmax = v > vmax ? v : vmax; }
    const int vmin = vmax / 10 > K1_MIN_VOTES ? vmax / 10 : K1_MIN_VOTES;
    const int rlen = reads.len[r];
    const uint32_t* rw = reads.words + reads.word_off[r];
    const uint32_t* rn = reads.nplane ? reads.nplane + reads.word_off[r] : nullptr;
    const uint32_t p_first = chunk * (uint32_t)K1_CHUNK;

    // the cell of a position: active?  diagonal, static cap, packed words, prefix shared with the position before
    struct Cell { int act, alen, kb, cap, lcp; uint32_t woff; };
    auto load_cell = [&](uint32_t p, int& j_min, int& j_max) -> Cell {
        Cell c; c.act = 0; c.alen = 0; c.kb = 0; c.cap = 0; c.lcp = 0; c.woff = 0; j_min = 0x7FFFFFFF; j_max = -1;
        if (p < n_alleles) {
            c.alen = pos.alen[p];
            const int off = pos.off[p];
            const uint32_t g = pos.gene[p];
            if (c.alen > 0 && off != SP_NO_DIAG && votes_rg[(uint64_t)r * n_genes + g] >= vmin) {
                c.kb = d_rg[(uint64_t)r * n_genes + g] - off - SP_BAND / 2;
                int i_min, i_max;
                if (spw::cell_windows(c.alen, rlen, c.kb, i_min, i_max, j_min, j_max)) {
                    // nm <= 0.03 * aligned span <= 0.03 * allele length (realigner.rs:138-141)
                    c.cap = (int)(0.03 * (double)c.alen) + 1; if (c.cap > SP_MAX_ED) c.cap = SP_MAX_ED;
                    if (c.cap > pass_cap) c.cap = pass_cap;
                    c.act = 1; c.woff = pos.woff[p]; c.lcp = pos.lcp[p];
                } else { j_min = 0x7FFFFFFF; j_max = -1; }
            }
        }
        return c;
    };

    // the 64 cells of the workgroup: metadata, union of their read windows, longest allele in use
    const uint32_t p_mine = p_first + lane;
    int w_lo, w_hi;
    const Cell u = load_cell(p_mine, w_lo, w_hi);
    int longest = u.act ? u.alen : 0;
    w_lo = -spw::wave_max(-w_lo); w_hi = spw::wave_max(w_hi); longest = spw::wave_max(longest);
    if (w_hi < 0) {                                       // nothing to run (the positions of another gene, mostly)
        if (cell_out && p_mine < n_alleles) cell_out[(uint64_t)r * n_alleles + p_mine] = SP_CELL_NONE;
        return;
    }
    // Register prefetch of one allele, fixed shape: lane l takes words [4l, 4l+4) and word 256+l (sequences start 16-byte aligned
    // and the set is padded, SP_SEQ_PAD_WORDS), 320 words = 5,088 bases + guard; longer alleles are staged directly.  The
    // running bound of the read comes along: any value it ever had is a valid bound.
    uint4 pre4 = make_uint4(0, 0, 0, 0); uint32_t pre1 = 0; unsigned long long pre_bound = K1_NO_BOUND; int pre_for = -1;
    auto prefetch = [&](int idx, uint32_t woff) {
        const uint32_t* aw = alleles.words + woff;
        pre4 = *reinterpret_cast<const uint4*>(aw + 4 * lane);
        pre1 = aw[256 + lane];
        if (bound) pre_bound = __hip_atomic_load(&bound[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pre_for = idx;
    };
    {
        const int g = 0;
        const unsigned long long active0 = __ballot(u.act != 0);
        // the first allele is on its way while the read window is staged
        { const int j0 = __builtin_ctzll(active0); prefetch(j0, (uint32_t)__builtin_amdgcn_readlane((int)u.woff, j0)); }
        int b_base;
        {
            const int w0 = w_lo >> 4;
            const int nw = ((w_hi + 15) >> 4) - w0 + 2;
            for (int w = lane; w < nw; w += SP_WAVE) { LB[w] = rw[w0 + w]; if (HASN) NB[w] = rn ? rn[w0 + w] : 0u; }
            b_base = w0 << 4;
        }
        // longest allele any cell of this read uses; the word only grows, so a (possibly stale) plain read that already shows the value saves the atomic:
        // one 8-byte atomic per workgroup was half of the kernel's 62.8 MB of HBM writes per launch (the other half: the wg_stats word below)
        if (read_maxlen && lane == 0 && __builtin_nontemporal_load(&read_maxlen[r]) < (uint32_t)longest) atomicMax(&read_maxlen[r], (uint32_t)longest);
        const unsigned long long active = __ballot(u.act != 0);
        uint32_t res = SP_CELL_NONE;
        // lane j: is the cell right before it in the order active too?  (a chain or a saved state never crosses a gap or a group)
        const int u_adjacent = u.act && spw::from_lower(u.act, 0);
        // the state the previous cell left for its successor (wfa_core, Snap)
        int snap_for = -1, snap_s = -1, snap_H = 0;
        // the caps of the group under the bound they were last worked out for (redone only when the bound has moved)
        int u_cap = u.cap; unsigned long long cap_bound = K1_NO_BOUND;
        int jc = active ? __builtin_ctzll(active) : 64;
        // measurement (bench.py's roofline over the cells actually executed): executed / resumed cells of this workgroup and the
        // algorithmic bytes of the executed ones (ceil(Lq/4) + ceil(Lt/4) + 32), all in scalar registers
        uint32_t st_exec = 0, st_resumed = 0, st_bytes = 0;
        while (jc < 64) {
            const int c_key = g * 64 + jc;
            if (pre_for != c_key) prefetch(c_key, (uint32_t)__builtin_amdgcn_readlane((int)u.woff, jc));      // (first of a group, behind a chain)
            if (bound && pre_bound != cap_bound) { u_cap = k1_dyn_cap(pre_bound, u.alen, u.cap); cap_bound = pre_bound; }
            const int c_alen = __builtin_amdgcn_readlane(u.alen, jc), c_kb = __builtin_amdgcn_readlane(u.kb, jc);
            const int c_cap = __builtin_amdgcn_readlane(u_cap, jc);
            const uint32_t c_woff = (uint32_t)__builtin_amdgcn_readlane((int)u.woff, jc);
            if (((c_alen + 15) >> 4) + 2 <= K1_PRE_WORDS) {
                *reinterpret_cast<uint4*>(LA + 4 * lane) = pre4;
                LA[256 + lane] = pre1;
            } else {
                spw::stage(LA, alleles.words + c_woff, 0, c_alen, lane);
            }
            // the next active cell streams into the registers while this one runs (it is the next to run unless this run settles it)
            const unsigned long long later = jc < 63 ? active & (~0ull << (jc + 1)) : 0ull;
            const int jn = later ? __builtin_ctzll(later) : 64;
            if (jn < 64) prefetch(g * 64 + jn, (uint32_t)__builtin_amdgcn_readlane((int)u.woff, jn & 63));
            if (HASN) {
                if (alleles.nplane) spw::stage(NA, alleles.nplane + c_woff, 0, c_alen, lane);
                else for (int w = lane; w < ((c_alen + 15) >> 4) + 2; w += SP_WAVE) NA[w] = 0;
            }
            spw::wave_lds_sync();
            {
                const int lt = c_alen + SP_BAND < rlen ? c_alen + SP_BAND : rlen;
                st_exec += 1; st_bytes += (uint32_t)(((c_alen + 3) >> 2) + ((lt + 3) >> 2) + 32);
            }
            SP_STAT(0, 1); SP_STAT(7, c_cap); SP_STAT(48 + (c_cap < 15 ? c_cap : 15), 1);
            spw::CellOut o; o.ok = 0; o.nm = 0; o.a_start = o.a_end = o.b_start = o.b_end = 0; o.explored = 0x7FFFFFFF;
            int nxt = jn;
            if (c_kb >= 0) {
                // the successor in the order shares its lcp with this allele: leave it the last state that stayed inside those bases,
                // and start from the state the predecessor left when it was made for this cell and is not past this cell's cap
                const bool has_succ = jc < 63 && __builtin_amdgcn_readlane(u_adjacent, (jc + 1) & 63);
                const int thr2 = has_succ ? __builtin_amdgcn_readlane(u.lcp, (jc + 1) & 63) << 1 : 0;
                const bool resume = snap_for == jc && snap_s >= 0 && snap_s <= c_cap;
                SP_STAT(8, resume ? 1 : 0); SP_STAT(9, resume ? snap_s + 1 : 0);
                st_resumed += resume ? 1u : 0u;
                int out_s = -1, out_H = 0;
                spw::wfa_core<false, HASN, false, true>(LA, NA, 0, c_alen, LB, NB, -b_base, rlen, c_kb, c_cap, lane, nullptr, nullptr, o,
                                                        thr2, resume ? snap_s : -1, snap_H, &out_s, &out_H);
                snap_for = jc + 1; snap_s = __builtin_amdgcn_readfirstlane(out_s); snap_H = out_H;
            } else {
                spw::wfa_core<false, HASN, false>(LA, NA, 0, c_alen, LB, NB, -b_base, rlen, c_kb, c_cap, lane, nullptr, nullptr, o);
                snap_for = -1;
            }
            if (o.ok) {
                const int span = o.a_end - o.a_start;
                if (lane == jc) res = ((uint32_t)o.nm << 16) | (uint32_t)span;
                if (bound && lane == 0) {
                    const double pen = score_value(c_alen, o.nm, c_alen - span), ed = score_value(span, o.nm, 0);
                    if (pen <= 0.5 && ed <= 0.03) {
                        const unsigned long long nn10 = o.nm ? 10ull * (unsigned long long)o.nm : 1ull;
                        const unsigned long long cand = (nn10 << 32) | (unsigned long long)span;
                        unsigned long long curb = __hip_atomic_load(&bound[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (;;) {
                            const unsigned long long cn = curb >> 32, cs = curb & 0xFFFFFFFFull;
                            const bool better = (cn == 0xFFFFFFFFull) || (nn10 * cs < cn * (unsigned long long)span);
                            if (!better) break;
                            const unsigned long long prev = atomicCAS(&bound[r], curb, cand);
                            if (prev == curb) break;
                            curb = prev;
                        }
                        if (winner) {
                            // the read's best acceptable cell so far, by the very comparison k1_reduce_kernel makes on the cell
                            // matrix (f64 ratio, ties to the lowest allele index): nm << 40 | span << 24 | allele
                            const uint32_t a = order[p_first + g * 64 + jc];
                            const unsigned long long mine = ((unsigned long long)o.nm << 40) | ((unsigned long long)span << 24) | a;
                            unsigned long long cw = __hip_atomic_load(&winner[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            for (;;) {
                                bool better = cw == ~0ull;
                                if (!better) {
                                    const double ced = score_value((int)((cw >> 24) & 0xFFFFu), (int)(cw >> 40), 0);
                                    better = ed < ced || (ed == ced && a < (uint32_t)(cw & 0xFFFFFFu));
                                }
                                if (!better) break;
                                const unsigned long long prev = atomicCAS(&winner[r], cw, mine);
                                if (prev == cw) break;
                                cw = prev;
                            }
                        }
                    }
                }
            } else if (jc < 63) {
                // the cells behind jc that this failed run settles: adjacent in the order, sharing more than the explored prefix,
                // caps not growing -- the first lane behind jc that breaks the chain ends it; the next active lane from there runs
                const int extent = __builtin_amdgcn_readfirstlane(o.explored);
                const bool carries = u_adjacent && u.lcp > extent && u_cap <= spw::from_lower(u_cap, -1);
                const unsigned long long behind = ~0ull << (jc + 1);
                const unsigned long long stop = __ballot(!carries) & behind;
                const int brk = stop ? __builtin_ctzll(stop) : 64;
                const unsigned long long rest = brk < 64 ? active & (~0ull << brk) : 0ull;
                nxt = rest ? __builtin_ctzll(rest) : 64;
                SP_STAT(1, __builtin_popcountll(active & behind & ~rest));
            }
            spw::wave_lds_sync();
            jc = nxt;
        }
        if (cell_out && p_mine < n_alleles) cell_out[(uint64_t)r * n_alleles + p_mine] = res;
        if (wg_stats && lane == 0) wg_stats[blockIdx.x] = make_uint2(st_exec | (st_resumed << 8) | ((uint32_t)__builtin_popcountll(active) << 16), st_bytes);
    }
}

// sums the per-workgroup words of a cells launch into the context's counters: active / executed / resumed cells, algorithmic bytes
__global__ __launch_bounds__(256) void k1_stats_kernel(const uint2* __restrict__ wg_stats, uint32_t n, unsigned long long* __restrict__ counters) {
    unsigned long long act = 0, ex = 0, res = 0, by = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint2 w = wg_stats[i];
        ex += w.x & 0xFFu; res += (w.x >> 8) & 0xFFu; act += w.x >> 16; by += w.y;
    }
    for (int o = 32; o > 0; o >>= 1) { act += __shfl_xor(act, o); ex += __shfl_xor(ex, o); res += __shfl_xor(res, o); by += __shfl_xor(by, o); }
    if ((threadIdx.x & 63) == 0) {
        if (act) atomicAdd(&counters[SPC_K1_ACTIVE], act);
        if (ex) atomicAdd(&counters[SPC_K1_EXECUTED], ex);
        if (res) atomicAdd(&counters[SPC_K1_RESUMED], res);
        if (by) atomicAdd(&counters[SPC_K1_BYTES], by);
    }
}

// Iterative deepening bookkeeping: after a pass whose cells were capped at pass_cap edits, read r is settled when an
// acceptable cell exists and no unfinished cell (nm >= pass_cap + 1, span <= max_alen) can reach its ratio:
//   (pass_cap + 1) / max_alen > num10 / (10 * span)   <=>   10 * (pass_cap + 1) * span > num10 * max_alen
__global__ void k1_done_kernel(const unsigned long long* __restrict__ bound, uint8_t* __restrict__ done, uint32_t n_reads,
                               int pass_cap, const uint32_t* __restrict__ read_maxlen,
                               uint32_t* __restrict__ n_open, uint32_t* __restrict__ open_list) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads || done[r]) return;
    const unsigned long long b = bound[r];
    const unsigned long long nb10 = b >> 32, sb = b & 0xFFFFFFFFull;
    const unsigned long long max_alen = read_maxlen[r];
    if (max_alen == 0 ||                                                  // no cell at all: nothing deeper can appear
        (nb10 != 0xFFFFFFFFull && 10ull * (unsigned long long)(pass_cap + 1) * sb > nb10 * max_alen)) done[r] = 1;
    else open_list[atomicAdd(n_open, 1u)] = r;
}

// Reverse strand (src/hla/realigner.rs:178-193: a read whose best mapping is not Forward is dropped).  The strand is decided where minimap2
// decides it, at the seeds: a read whose best anchor on the reverse-complemented gene references collects more 16-mer votes than its best
// forward anchor (and at least K1_MIN_VOTES) is dropped with status 2.  Only reads with a WEAK forward anchor (< K1_WEAK_VOTES) can lose that
// comparison -- a read cannot share hundreds of exact 16-mers with both strands of one locus -- so only those are anchored a second time.
constexpr int K1_WEAK_VOTES = 512, K1_MIN_VOTES_REV = 16;
__global__ void k1_weak_kernel(const int32_t* __restrict__ votes, uint32_t n_reads, uint32_t n_genes, uint32_t* __restrict__ n_weak, uint32_t* __restrict__ weak_list) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int best = 0;
    for (uint32_t g = 0; g < n_genes; ++g) { const int v = votes[(size_t)r * n_genes + g]; best = v > best ? v : best; }
    if (best < K1_WEAK_VOTES) weak_list[atomicAdd(n_weak, 1u)] = r;
}
__global__ void k1_weak_pairs_kernel(const uint32_t* __restrict__ weak_list, uint32_t n_weak, uint32_t n_genes, uint32_t* __restrict__ a_idx, uint32_t* __restrict__ b_idx) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_weak * n_genes) { a_idx[p] = p % n_genes; b_idx[p] = weak_list[p / n_genes]; }
}
__global__ void k1_reverse_kernel(const uint32_t* __restrict__ weak_list, uint32_t n_weak, uint32_t n_genes, const int32_t* __restrict__ votes_fwd,
                                  const int32_t* __restrict__ votes_rev, uint8_t* __restrict__ is_reverse) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_weak) return;
    const uint32_t r = weak_list[k];
    int fb = 0, rb = 0;
    for (uint32_t g = 0; g < n_genes; ++g) {
        const int f = votes_fwd[(size_t)r * n_genes + g], v = votes_rev[(size_t)k * n_genes + g];
        fb = f > fb ? f : fb; rb = v > rb ? v : rb;
    }
    is_reverse[r] = (rb >= K1_MIN_VOTES_REV && rb > fb) ? (rb >= 2 * fb ? 2 : 1) : 0;      // 2: a clear margin, 1: out-voted narrowly (only counts against a read that found nothing acceptable forwards)
}

// the cells of the re-score of a K1 batch (sp_rescore_mappings): read r (the window side, minimap2's query) against its best allele (the streamed side, minimap2's
// target) on the diagonal its alignment lies on; reads without a best allele are marked "no mapping"
__global__ void k1_rescore_cells_kernel(const sp_hla_realign* __restrict__ rec, uint32_t n_reads, CellDesc* __restrict__ cells, sp_aln* __restrict__ ref) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const sp_hla_realign& q = rec[r];
    CellDesc c; c.a = 0; c.b = r; c.diag = 0; c.max_ed = -1; c.b_lo = 0; c.b_hi = -1;
    if (q.best_allele >= 0 && q.aln.ok) { c.a = (uint32_t)q.best_allele; c.diag = ((q.aln.b_start - q.aln.a_start) + (q.aln.b_end - q.aln.a_end)) / 2; c.max_ed = 127; }
    cells[r] = c; ref[r] = q.aln;
}
__global__ void k1_affine_store_kernel(sp_hla_realign* __restrict__ rec, uint32_t n_reads, const sp_affine_aln* __restrict__ af) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    sp_hla_realign& q = rec[r];
    const sp_affine_aln a = af[r];
    q.mm2_score = a.score; q.mm2_nm = a.nm; q.mm2_t_start = a.b_start; q.mm2_t_end = a.b_end; q.mm2_q_start = a.a_start; q.mm2_q_end = a.a_end;
}

// seeded mode: the accepted mapping's re-score and the counts of the seeded stage go into the record; a read whose accepted mapping is on the reverse strand is dropped
// (src/hla/realigner.rs:178-193)
__global__ void k1_seed_store_kernel(sp_hla_realign* __restrict__ rec, uint32_t n_reads, const sp_k1_seed_info* __restrict__ info, const sp_affine_aln* __restrict__ af) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    sp_hla_realign& q = rec[r];
    const sp_k1_seed_info i = info[r]; const sp_affine_aln a = af[r];
    q.k1_chains = i.n_chains; q.k1_mappings = i.n_mappings; q.k1_chain_score = i.chain_score;
    if (i.pick >= 0 && i.rev) q.status = 2;
    q.mm2_score = a.score; q.mm2_nm = a.nm; q.mm2_t_start = a.b_start; q.mm2_t_end = a.b_end; q.mm2_q_start = a.a_start; q.mm2_q_end = a.a_end;
}

// K1 reduce: one wavefront per read, exact restatement of the acceptance loop (realigner.rs:124-146)
// seeded mode: the same pair list with every pair but (gene of the read's accepted allele, read) marked for the anchor kernel to skip
__global__ void k1_seed_pairs_kernel(const int32_t* __restrict__ best, const uint32_t* __restrict__ gene_of, uint32_t n_reads, uint32_t n_genes, uint32_t* __restrict__ a_idx, uint32_t* __restrict__ b_idx) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_reads * n_genes) return;
    const uint32_t r = p / n_genes, g = p % n_genes;
    const int b = best[r];
    a_idx[p] = g; b_idx[p] = (b >= 0 && gene_of[b] == g) ? r : SP_ANCHOR_SKIP;
}
// pair p of the K1 anchor is (gene p % G, read p / G); every read starts without a bound
__global__ void k1_init_kernel(uint32_t* __restrict__ a_idx, uint32_t* __restrict__ b_idx, uint32_t n_reads, uint32_t n_genes,
                               unsigned long long* __restrict__ bound) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_reads * n_genes) { a_idx[p] = p % n_genes; b_idx[p] = p / n_genes; }
    if (bound && p < n_reads) bound[p] = K1_NO_BOUND;
}

// production mode keeps no cell matrix: the cells kernels maintain the winner word of every read (see there), this unpacks it
__global__ void k1_winner_kernel(const unsigned long long* __restrict__ winner, uint32_t n_reads, int32_t* __restrict__ best_out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) { const unsigned long long w = winner[r]; best_out[r] = w == ~0ull ? -1 : (int32_t)(w & 0xFFFFFFu); }
}

__global__ __launch_bounds__(256) void k1_reduce_kernel(const uint32_t* __restrict__ cell_out, const int32_t* __restrict__ allele_len,
                                                        const uint32_t* __restrict__ order, uint32_t n_alleles, uint32_t n_reads, int32_t* __restrict__ best_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= n_reads) return;
    double best = 1.0;             // custom_score(false) of MappingStats(read_len, read_len, 0)
    int best_idx = 0x7FFFFFFF;
    for (uint32_t p = lane; p < n_alleles; p += 64) {
        uint32_t c = cell_out[(uint64_t)r * n_alleles + p];
        if (c == SP_CELL_NONE) continue;
        const uint32_t a = order[p];                         // ties go to the lowest allele index, whatever the visiting order
        int nm = (int)(c >> 16), span = (int)(c & 0xFFFFu), tlen = allele_len[a];
        int unmapped = tlen - span;
        double pen = score_value(tlen, nm, unmapped);
        double ed = score_value(tlen - unmapped, nm, 0);
        if (pen <= 0.5 && ed <= 0.03 && (ed < best || (ed == best && (int)a < best_idx))) { best = ed; best_idx = (int)a; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        double ob = __shfl_xor(best, o); int oi = __shfl_xor(best_idx, o);
        if (ob < best || (ob == best && oi < best_idx)) { best = ob; best_idx = oi; }
    }
    if (lane == 0) best_out[r] = best_idx == 0x7FFFFFFF ? -1 : best_idx;
}

// hpc_pos(seq, position) on a packed sequence: number of run boundaries in (0, position]
// (src/util/homopolymers.rs:25-42).  One wavefront.
__device__ __forceinline__ int hpc_pos_wave(const uint32_t* __restrict__ w, int len, int position, int lane) {
    if (len <= 0) return 0;
    int last = position < len - 1 ? position : len - 1;     // boundaries at p in [1, last]
    int cnt = 0;
    for (int base = lane * 16; base <= last; base += 64 * 16) {
        // bases base .. base+15 ; boundary at p when base[p] != base[p-1]
        int wi = base >> 4;
        uint32_t cur = w[wi];
        uint32_t prev = wi > 0 ? w[wi - 1] : 0;
        uint32_t shifted = (cur << 2) | (prev >> 30);          // base p-1 aligned to base p
        uint32_t x = cur ^ shifted; uint32_t diff = (x | (x >> 1)) & 0x55555555u;
        if (base == 0) diff &= ~1u;                              // p = 0 is not a boundary
        int hi = last - base;                                    // keep p <= last
        if (hi < 15) diff &= (hi >= 0) ? ((1u << ((hi + 1) << 1)) - 1u) : 0u;
        cnt += __builtin_popcount(diff);
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    return cnt;
}

// the end of realign_record once the segment is mapped to the gene's reference (realigner.rs:262-331): the segment's bounds and the DNA / HPC offsets.  One wavefront.
__device__ __forceinline__ void k1_finish_segment(sp_hla_realign& res, int rm_ok, int rm_nm, int rm_a_start, int rm_a_end, int rm_b_start, int rm_b_end, int buf_start, int reflen,
                                                  int a, uint32_t g, int db_start, int db_end, int bm_a_start, const SeqSetView& alleles_gene, const int32_t* __restrict__ am,
                                                  const int32_t* __restrict__ hpc_ref, const uint64_t* __restrict__ hpc_ref_off, int lane) {
    // select_best_mapping(target-based, penalised): must beat the 1.0 default (util/mapping.rs:22-57)
    if (!(rm_ok && score_value(reflen, rm_nm, reflen - (rm_a_end - rm_a_start)) < 1.0)) return;
    const int adj_start = buf_start + rm_b_start, adj_end = buf_start + rm_b_end;
    res.seg_start = db_start < adj_start ? db_start : adj_start;
    res.seg_end = db_end > adj_end ? db_end : adj_end;
    const int32_t* hp = hpc_ref + hpc_ref_off[g];
    int d, h;
    if (adj_start < db_start || !am[a * 3 + 0]) {
        d = rm_a_start; h = hp[d];
    } else {
        int added = am[a * 3 + 2] - am[a * 3 + 1]; if (added < 0) added = 0;
        d = added + bm_a_start;
        const uint32_t* gw = alleles_gene.words + alleles_gene.word_off[a];
        h = hp[added < reflen ? added : reflen] + hpc_pos_wave(gw, alleles_gene.len[a], bm_a_start, lane);
    }
    res.dna_offset = d; res.hpc_offset = h;
    res.status = 0;
}

// what realign_record takes from the accepted mapping (bm.query_start / query_end / target_start, realigner.rs:219-221,307): in seeded mode the re-scored mapping's numbers
// (minimap2's end-clipped extent), else the cell's
struct K1Span { int db_start, db_end, t_start; };
__device__ __forceinline__ K1Span k1_span(const sp_aln& cell, const sp_affine_aln* af) {
    K1Span s; s.db_start = cell.b_start; s.db_end = cell.b_end; s.t_start = cell.a_start;
    if (af && af->score > 0) { s.db_start = af->a_start; s.db_end = af->a_end; s.t_start = af->b_start; }
    return s;
}

// Seeded mode (the reference's call pattern): a segment whose 64-diagonal cell against the gene's reference found nothing -- a read with a 40+ base insertion or
// deletion against the reference, which minimap2 chains across -- is run again on the wide band (sp_cells_wide_kernel), like the chains' own cells.
// k1_seg_retry_cells_kernel lists those reads' cells (the others: no cell), k1_seg_retry_finish_kernel completes their records.
__global__ void k1_seg_retry_cells_kernel(const sp_hla_realign* __restrict__ out, uint32_t n_reads, const int32_t* __restrict__ read_len, const int32_t* __restrict__ d_rg, int n_genes,
                                          const sp_affine_aln* __restrict__ af, CellDesc* __restrict__ cells, sp_aln* __restrict__ alns) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    CellDesc c; c.a = 0; c.b = r; c.diag = SP_NO_DIAG; c.max_ed = SP_MAX_ED; c.b_lo = 0; c.b_hi = -1;
    const sp_hla_realign o = out[r];
    if (o.status == 3 && o.best_allele >= 0 && o.aln.ok) {
        const int rlen = read_len[r], buffer = 1000;
        const sp_affine_aln fa = af[r];
        const K1Span sp = k1_span(o.aln, &fa);
        const int buf_start = sp.db_start > buffer ? sp.db_start - buffer : 0;
        const int buf_end = sp.db_end + buffer < rlen ? sp.db_end + buffer : rlen;
        c.a = (uint32_t)o.gene; c.b_lo = buf_start; c.b_hi = buf_end;
        c.diag = d_rg[(uint64_t)r * n_genes + o.gene] - buf_start;
    }
    cells[r] = c;
    sp_aln z; memset(&z, 0, sizeof z);
    alns[r] = z;
}
__global__ __launch_bounds__(256) void k1_seg_retry_finish_kernel(SeqSetView alleles_gene, SeqSetView refs, const int32_t* __restrict__ am, const int32_t* __restrict__ hpc_ref,
                                                                  const uint64_t* __restrict__ hpc_ref_off, const CellDesc* __restrict__ cells, const sp_aln* __restrict__ alns,
                                                                  const sp_affine_aln* __restrict__ af, uint32_t n_reads, sp_hla_realign* __restrict__ out, sp_aln* __restrict__ rm_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= n_reads) return;
    const CellDesc c = cells[r];
    if (c.diag == SP_NO_DIAG) return;
    const sp_aln rm = alns[r];
    if (!rm.ok) return;
    if (rm_out && lane == 0) { sp_aln w = rm; w.a_len = refs.len[c.a]; w.b_len = c.b_hi - c.b_lo; rm_out[r] = w; }
    sp_hla_realign res = out[r];
    const sp_affine_aln fa = af[r];
    const K1Span sp = k1_span(res.aln, &fa);
    k1_finish_segment(res, rm.ok, rm.nm, rm.a_start, rm.a_end, rm.b_start, rm.b_end, c.b_lo, refs.len[c.a], res.best_allele, c.a, sp.db_start, sp.db_end, sp.t_start, alleles_gene, am, hpc_ref, hpc_ref_off, lane);
    if (lane == 0 && res.status == 0) out[r] = res;
}

// Seeded mode, the second stage of realign_record in the reference's numbers (round 6).  The read's segment +- 1,000 bases is mapped to the gene's reference by minimap2
// (gene_aligner.map, src/hla/realigner.rs:231) and the record takes that mapping's query span and target start (:262-283): the library's cell found the placement (rm: ends-free unit
// cost, wide band when needed); its extent as minimap2 reports it -- two-piece affine gaps, END CLIPPING at the ends of the reference -- is the re-score of that placement on the 256
// diagonals around it, segment = query, reference = target (sp_rescore_mappings).  k1_seg_rs_prep_kernel lists the placements and cuts the segments out of the reads into a set of
// their own (the re-score knows whole sequences only: the segment must end where the buffer ends), k1_seg_rs_finish_kernel completes the records from the re-scored extents
// (tests/hla_expected.py K1Tables.record states the same; against the reference-call-pattern port: tests/test_gpu_concordance.py K1_RECORDS_SAME_MIN).
__global__ void k1_seg_rs_prep_kernel(const sp_hla_realign* __restrict__ out, uint32_t n_reads, const int32_t* __restrict__ read_len, const sp_affine_aln* __restrict__ af,
                                      const sp_aln* __restrict__ rm, CellDesc* __restrict__ cells, sp_aln* __restrict__ ref, int32_t* __restrict__ seg_start, int32_t* __restrict__ seg_len) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    CellDesc c; c.a = 0; c.b = r; c.diag = 0; c.max_ed = -1; c.b_lo = 0; c.b_hi = -1;
    sp_aln w = rm[r];
    int s0 = 0, sl = 0;
    const sp_hla_realign o = out[r];
    if (o.best_allele >= 0 && o.aln.ok && w.ok && (o.status == 0 || o.status == 3)) {
        const int rlen = read_len[r], buffer = 1000;
        const sp_affine_aln fa = af[r];
        const K1Span sp = k1_span(o.aln, &fa);
        s0 = sp.db_start > buffer ? sp.db_start - buffer : 0;
        const int buf_end = sp.db_end + buffer < rlen ? sp.db_end + buffer : rlen;
        sl = buf_end - s0;
        c.a = (uint32_t)o.gene; c.diag = ((w.b_start - w.a_start) + (w.b_end - w.a_end)) / 2; c.max_ed = SP_MAX_ED;
    } else w.ok = 0;
    cells[r] = c; ref[r] = w; seg_start[r] = s0; seg_len[r] = sl;
}
// segment r = bases [start[r], start[r] + len[r]) of read r, in the read's own slot of a second word buffer (the set shares the reads' word offsets)
__global__ void __launch_bounds__(256) k1_seg_slice_kernel(SeqSetView reads, const int32_t* __restrict__ start, const int32_t* __restrict__ len, uint32_t n,
                                                           uint32_t* __restrict__ out_words, uint32_t* __restrict__ out_nplane) {
    const int lane = threadIdx.x & 63; const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const uint64_t off = reads.word_off[r];
    const uint32_t* src = reads.words + off;
    const uint32_t* nsrc = reads.nplane ? reads.nplane + off : nullptr;
    const int p0 = start[r], L = len[r];
    const int nw = ((reads.len[r] + 15) >> 4) + 2;                            // (the read's slot holds at least this many words: its bases + the guard words)
    const int w0 = p0 >> 4; const uint32_t sh = (uint32_t)(p0 & 15) << 1;
    for (int j = lane; j < nw; j += 64) {
        uint32_t v = 0, nv = 0;
        const int first = j << 4;
        if (first < L) {
            v = __builtin_amdgcn_alignbit(src[w0 + j + 1], src[w0 + j], sh);
            if (nsrc) nv = __builtin_amdgcn_alignbit(nsrc[w0 + j + 1], nsrc[w0 + j], sh);
            const int rem = L - first;
            if (rem < 16) { const uint32_t m = (1u << (rem << 1)) - 1; v &= m; nv &= m; }
        }
        out_words[off + j] = v;
        if (out_nplane) out_nplane[off + j] = nv;
    }
}
__global__ __launch_bounds__(256) void k1_seg_rs_finish_kernel(SeqSetView alleles_gene, SeqSetView refs, const int32_t* __restrict__ am, const int32_t* __restrict__ hpc_ref,
                                                               const uint64_t* __restrict__ hpc_ref_off, const CellDesc* __restrict__ cells, const int32_t* __restrict__ seg_start,
                                                               const sp_affine_aln* __restrict__ seg_af, const sp_affine_aln* __restrict__ af, uint32_t n_reads, sp_hla_realign* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= n_reads) return;
    const CellDesc c = cells[r];
    if (c.max_ed < 0) return;
    const sp_affine_aln sa = seg_af[r];
    if (sa.score <= 0) return;                                              // (nothing aligns the reference's way: the record keeps the cell's extent)
    sp_hla_realign res = out[r];
    res.status = 3; res.seg_start = 0; res.seg_end = 0; res.dna_offset = 0; res.hpc_offset = 0;
    const sp_affine_aln fa = af[r];
    const K1Span sp = k1_span(res.aln, &fa);
    // the re-scored mapping: a_* on the query (the segment), b_* on the target (the reference)
    k1_finish_segment(res, 1, sa.nm, sa.b_start, sa.b_end, sa.a_start, sa.a_end, seg_start[r], refs.len[c.a], res.best_allele, c.a, sp.db_start, sp.db_end, sp.t_start, alleles_gene, am, hpc_ref, hpc_ref_off, lane);
    if (lane == 0) out[r] = res;
}

// K1 finalize: one wavefront per read: full alignment of the accepted allele, segment +-1000 against the
// gene reference, offsets (realigner.rs:219-331).
template <bool HASN>
__global__ __launch_bounds__(256) void k1_finalize_kernel(SeqSetView alleles_fwd, SeqSetView alleles_gene, SeqSetView refs, SeqSetView reads,
                                                          const uint32_t* __restrict__ gene_of, const int32_t* __restrict__ off_fwd,
                                                          const int32_t* __restrict__ d_rg, int n_genes,
                                                          const int32_t* __restrict__ am, const int32_t* __restrict__ hpc_ref,
                                                          const uint64_t* __restrict__ hpc_ref_off,
                                                          const int32_t* __restrict__ best_in, uint32_t n_reads,
                                                          sp_hla_realign* __restrict__ out, int slot_words, const sp_aln* __restrict__ aln_in,
                                                          const sp_affine_aln* __restrict__ af_in, sp_aln* __restrict__ rm_out) {
    extern __shared__ uint32_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* slot = lds + wave * slot_words;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= n_reads) return;
    sp_hla_realign res;
    memset(&res, 0, sizeof(res));
    res.status = 1; res.best_allele = -1; res.gene = -1;
    const int a = best_in[r];
    const int rlen = reads.len[r];
    if (a >= 0) {
        const uint32_t g = gene_of[a];
        const int alen = alleles_fwd.len[a];
        const uint32_t* rw = reads.words + reads.word_off[r];
        const uint32_t* rn = reads.nplane ? reads.nplane + reads.word_off[r] : nullptr;
        const int drg = d_rg[(uint64_t)r * n_genes + g];
        spw::CellIn in;
        in.a_words = alleles_fwd.words + alleles_fwd.word_off[a];
        in.a_nplane = alleles_fwd.nplane ? alleles_fwd.nplane + alleles_fwd.word_off[a] : nullptr;
        in.a0 = 0; in.a1 = alen; in.b_words = rw; in.b_nplane = rn; in.b0 = 0; in.b1 = rlen;
        in.diag = drg - off_fwd[a];
        int cap = (int)(0.03 * (double)alen) + 1; if (cap > SP_MAX_ED) cap = SP_MAX_ED;
        in.max_ed = cap;
        spw::CellOut bm;
        if (aln_in) {                      // seeded mode: the accepted mapping's cell was run on its chain's diagonal (sp_hla_seed.hip)
            const sp_aln w = aln_in[r];
            bm.ok = w.ok; bm.nm = w.nm; bm.a_start = w.a_start; bm.a_end = w.a_end; bm.b_start = w.b_start; bm.b_end = w.b_end; bm.explored = 0;
        } else spw::wfa_cell<false, HASN>(in, slot, slot_words, lane, nullptr, nullptr, bm);
        res.best_allele = a; res.gene = (int)g;
        res.nm = bm.nm; res.target_len = alen; res.unmapped = alen - (bm.a_end - bm.a_start);
        res.aln.ok = bm.ok; res.aln.nm = bm.nm; res.aln.a_start = bm.a_start; res.aln.a_end = bm.a_end;
        res.aln.b_start = bm.b_start; res.aln.b_end = bm.b_end; res.aln.a_len = alen; res.aln.b_len = rlen;
        res.status = 3;
        if (bm.ok) {
            sp_affine_aln fa; if (af_in) fa = af_in[r];
            const K1Span sp = k1_span(res.aln, af_in ? &fa : nullptr);
            const int db_start = sp.db_start, db_end = sp.db_end;
            const int buffer = 1000;
            const int buf_start = db_start > buffer ? db_start - buffer : 0;
            const int buf_end = db_end + buffer < rlen ? db_end + buffer : rlen;
            const int reflen = refs.len[g];
            spw::CellIn in2;
            in2.a_words = refs.words + refs.word_off[g];
            in2.a_nplane = refs.nplane ? refs.nplane + refs.word_off[g] : nullptr;
            in2.a0 = 0; in2.a1 = reflen; in2.b_words = rw; in2.b_nplane = rn; in2.b0 = buf_start; in2.b1 = buf_end;
            in2.diag = drg - buf_start; in2.max_ed = SP_MAX_ED;
            spw::CellOut rm;
            spw::wfa_cell<false, HASN>(in2, slot, slot_words, lane, nullptr, nullptr, rm);
            k1_finish_segment(res, rm.ok, rm.nm, rm.a_start, rm.a_end, rm.b_start, rm.b_end, buf_start, reflen, a, g, db_start, db_end, sp.t_start, alleles_gene, am, hpc_ref, hpc_ref_off, lane);
            if (rm_out && lane == 0) {                                      // (seeded mode: the segment's mapping on the reference is re-scored the reference's way, k1_seg_rs_*)
                sp_aln w; w.ok = rm.ok; w.nm = rm.nm; w.a_start = rm.a_start; w.a_end = rm.a_end; w.b_start = rm.b_start; w.b_end = rm.b_end; w.a_len = reflen; w.b_len = buf_end - buf_start;
                rm_out[r] = w;
            }
        }
    }
    if (lane == 0) out[r] = res;
}

// =============================================================================================
// K2
// =============================================================================================
struct K2Level {           // one level of an HlaProcessedMatch (processed_match.rs:10-21)
    int32_t present, range_start, range_end, len, nm, unmapped;
};

// build the cell list of one level from the anchor votes (A = consensus set idx level, B = allele)
__global__ void k2_build_cells_kernel(const uint32_t* __restrict__ allele_idx, uint32_t n, const uint32_t* __restrict__ cons_idx,
                                      const int32_t* __restrict__ anchor_diag, const int32_t* __restrict__ anchor_votes,
                                      const int32_t* __restrict__ allele_len, CellDesc* __restrict__ cells) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    CellDesc c;
    c.a = allele_idx[i]; c.b = cons_idx[i]; c.max_ed = K2_MAX_ED; c.b_lo = 0; c.b_hi = -1;
    // anchor diag = allele_pos - cons_pos ; cell diag = cons_pos - allele_pos
    c.diag = (allele_len[c.a] > 0 && anchor_votes[i] >= K2_MIN_VOTES) ? -anchor_diag[i] : SP_NO_DIAG;
    cells[i] = c;
}

// the cells of the re-score of the winners of a K2 batch at one level (sp_rescore_mappings): item k's best allele (streamed side, minimap2's query) against the item's
// consensus (window side, its target) on the diagonal the level's alignment lies on
__global__ void k2_rescore_cells_kernel(const int32_t* __restrict__ best, const uint32_t* __restrict__ seg, uint32_t n_items, const uint32_t* __restrict__ allele_idx,
                                        const uint32_t* __restrict__ cons_idx, const sp_aln* __restrict__ alns, CellDesc* __restrict__ cells, sp_aln* __restrict__ ref) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_items) return;
    CellDesc c; c.a = 0; c.b = 0; c.diag = 0; c.max_ed = -1; c.b_lo = 0; c.b_hi = -1;
    sp_aln al; al.ok = 0; al.nm = 0; al.a_start = al.a_end = al.b_start = al.b_end = al.a_len = al.b_len = 0;
    if (best[k] >= 0) {
        const uint32_t x = seg[k] + (uint32_t)best[k];
        al = alns[x];
        if (al.ok) { c.a = allele_idx[x]; c.b = cons_idx[x]; c.diag = ((al.b_start - al.a_start) + (al.b_end - al.a_end)) / 2; c.max_ed = K2_MAX_ED; }
    }
    cells[k] = c; ref[k] = al;
}

// select_best_mapping(query-based, penalised) + add_mapping bookkeeping (caller.rs:1447-1461, processed_match.rs:53-100)
__global__ void k2_levels_kernel(const sp_aln* __restrict__ alns, uint32_t n, K2Level* __restrict__ lv) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const sp_aln a = alns[i];
    K2Level l; l.present = 0; l.range_start = l.range_end = 0; l.len = l.nm = l.unmapped = -1;
    if (a.ok) {
        int qlen = a.a_len, unmapped = qlen - (a.a_end - a.a_start);
        if (score_value(qlen, a.nm, unmapped) < 1.0) {
            int clip_start = a.a_start, clip_end = qlen - a.a_end, t_off = a.b_start, rem = a.b_len - a.b_end;
            l.present = 1;
            l.range_start = t_off > clip_start ? t_off - clip_start : 0;
            l.range_end = a.b_end + (clip_end < rem ? clip_end : rem);
            l.len = qlen; l.nm = a.nm; l.unmapped = unmapped;
        }
    }
    lv[i] = l;
}

// pc[e] - pc[s] of process_mm_cigar (processed_match.rs:210-263) from the event list:
// every X / D / I event contributes 1 at pc index (b_pos + 1); clip padding contributes 1 per base.
__device__ __forceinline__ int k2_range_edits(const K2Level& l, const sp_aln& a, const uint32_t* __restrict__ ev, int s, int e) {
    // the events are in path order, so their B positions never decrease: two binary searches instead of a pass over all of them
    auto upto = [&](int x) {                              // events whose pc index (b_pos + 1) is <= x
        int lo = 0, hi = a.nm;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)(ev[mid] & 0x3FFFFFFFu) + 1 <= x) lo = mid + 1; else hi = mid; }
        return lo;
    };
    int cnt = upto(e) - upto(s);
    // soft-clip padding (only non-zero when the allele overhangs the consensus end it touches)
    int clip_start = a.a_start, t_off = a.b_start;
    int zero_pad = t_off > clip_start ? t_off - clip_start : 0;
    // indices zero_pad+1 .. t_off each add 1
    { int lo = zero_pad + 1 > s + 1 ? zero_pad + 1 : s + 1, hi = t_off < e ? t_off : e; if (hi >= lo) cnt += hi - lo + 1; }
    int clip_end = a.a_len - a.a_end, rem = a.b_len - a.b_end;
    int ext = clip_end < rem ? clip_end : rem;
    { int lo = a.b_end + 1 > s + 1 ? a.b_end + 1 : s + 1, hi = a.b_end + ext < e ? a.b_end + ext : e; if (hi >= lo) cnt += hi - lo + 1; }
    (void)l;
    return cnt;
}

__device__ __forceinline__ bool k2_score_less(const K2Level* l, const K2Level* r) {
    double ls0 = l[0].present ? score_value(l[0].len, l[0].nm, l[0].unmapped) : 1.0;
    double rs0 = r[0].present ? score_value(r[0].len, r[0].nm, r[0].unmapped) : 1.0;
    if (ls0 < rs0) return true;
    if (ls0 > rs0) return false;
    double ls1 = l[1].present ? score_value(l[1].len, l[1].nm, l[1].unmapped) : 1.0;
    double rs1 = r[1].present ? score_value(r[1].len, r[1].nm, r[1].unmapped) : 1.0;
    return ls1 < rs1;
}

// Sequential running-best scan of score_read (caller.rs:1411-1500) evaluated in parallel: the block looks for
// the first candidate after `pos` that beats the current best, adopts it, and continues behind it.
// lv / alns / ev are laid out [level][i].
// one workgroup per consensus: its candidates are entries [seg_off[b], seg_off[b+1]) of the batch (level arrays laid out [2][total])
__global__ __launch_bounds__(1024) void k2_scan_kernel(const K2Level* __restrict__ lv_all, const sp_aln* __restrict__ alns_all,
                                                       const uint32_t* __restrict__ ev_all, uint32_t ev_stride, uint32_t total,
                                                       const uint32_t* __restrict__ seg_off, int32_t* __restrict__ best_out) {
    const uint32_t seg = seg_off[blockIdx.x], n = seg_off[blockIdx.x + 1] - seg;
    __shared__ K2Level bl[2];
    __shared__ sp_aln ba[2];
    __shared__ uint32_t bev[2][K2_MAX_ED + 1];
    __shared__ int s_first;
    __shared__ int s_best;
    const int tid = threadIdx.x;
    if (tid < 2) { bl[tid].present = 0; bl[tid].range_start = bl[tid].range_end = 0; bl[tid].len = bl[tid].nm = bl[tid].unmapped = -1; ba[tid].nm = 0; }
    if (tid == 0) { s_best = -1; }
    __syncthreads();
    uint32_t pos = 0;
    while (pos < n) {
        if (tid == 0) s_first = 0x7FFFFFFF;
        __syncthreads();
        const uint32_t i = pos + tid;
        bool better = false;
        if (i < n) {
            K2Level cl[2] = { lv_all[seg + i], lv_all[(uint64_t)total + seg + i] };
            bool decided = false;
            for (int L = 0; L < 2 && !decided; ++L) {
                if (cl[L].present && bl[L].present) {
                    int os = cl[L].range_start > bl[L].range_start ? cl[L].range_start : bl[L].range_start;
                    int oe = cl[L].range_end < bl[L].range_end ? cl[L].range_end : bl[L].range_end;
                    int cn = 0, bn = 0;
                    if (os < oe) {
                        const sp_aln ca = alns_all[(uint64_t)L * total + seg + i];
                        cn = k2_range_edits(cl[L], ca, ev_all + ((uint64_t)L * total + seg + i) * ev_stride, os, oe);
                        bn = k2_range_edits(bl[L], ba[L], bev[L], os, oe);
                    }
                    if (cn < bn) { better = true; decided = true; }
                    else if (cn > bn) { better = false; decided = true; }
                } else if (!cl[L].present && !bl[L].present) {
                } else { better = cl[L].present != 0; decided = true; }
            }
            if (!decided) better = k2_score_less(cl, bl);
        }
        if (better) atomicMin(&s_first, (int)i);
        __syncthreads();
        const int first = s_first;
        if (first != 0x7FFFFFFF) {
            if (tid < 2) { bl[tid] = lv_all[(uint64_t)tid * total + seg + first]; ba[tid] = alns_all[(uint64_t)tid * total + seg + first]; }
            for (int x = tid; x < 2 * (K2_MAX_ED + 1); x += 1024) {
                int L = x / (K2_MAX_ED + 1), y = x % (K2_MAX_ED + 1);
                bev[L][y] = (uint32_t)y < ev_stride ? ev_all[((uint64_t)L * total + seg + first) * ev_stride + y] : 0;
            }
            if (tid == 0) s_best = first;
            pos = (uint32_t)first + 1;
        } else {
            pos += 1024;
        }
        __syncthreads();
    }
    if (tid == 0) best_out[blockIdx.x] = s_best;
}

__global__ void k2_stats_kernel(const K2Level* __restrict__ lv, uint32_t total, uint32_t seg, uint32_t n, const uint32_t* __restrict__ allele_idx, int32_t* __restrict__ stats) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t* s = stats + (uint64_t)allele_idx[seg + i] * 6;
    for (int L = 0; L < 2; ++L) {
        K2Level l = lv[(uint64_t)L * total + seg + i];
        s[L * 3 + 0] = l.present ? l.len : -1; s[L * 3 + 1] = l.present ? l.nm : -1; s[L * 3 + 2] = l.present ? l.unmapped : -1;
    }
}

// =============================================================================================
// host side
// =============================================================================================
static std::string revcomp(const char* s, size_t n) {
    std::string r(n, 'N');
    for (size_t i = 0; i < n; ++i) {
        char c = s[n - 1 - i];
        switch (c) { case 'A': case 'a': r[i] = 'T'; break; case 'C': case 'c': r[i] = 'G'; break;
                     case 'G': case 'g': r[i] = 'C'; break; case 'T': case 't': r[i] = 'A'; break; default: r[i] = 'N'; }
    }
    return r;
}

template <typename T> static T* dev_copy(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(1, v.size()) * sizeof(T)) != hipSuccess) return nullptr;
    if (!v.empty()) (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

// ---------------------------------------------------------------------------------------------
// K2 anchors through the allele set's 16-mer dictionary (sp_hla_dict.hip): the votes of a (consensus, allele) pair are the votes
// sp_anchor_kernel collects -- every 16-mer of the allele votes for allele position - consensus position for each of its <= SP_MAXOCC
// occurrences in the consensus -- but the occurrences of a 16-mer in a consensus are looked up once per distinct 16-mer of the gene
// (k2_hits_kernel) instead of once per allele that carries it.
// hits[item][id] = first entry of the 16-mer in the consensus' sorted table | occurrences << 24; 0 = none, or more than SP_MAXOCC
__global__ __launch_bounds__(256) void k2_hits_kernel(KmerIndexView KA, const uint32_t* __restrict__ dict, const uint32_t* __restrict__ dict_off,
                                                      const uint32_t* __restrict__ item_gene, int level, uint32_t max_dict, uint32_t* __restrict__ hits) {
    const uint32_t k = blockIdx.y, g = item_gene[k], e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= dict_off[g + 1] - dict_off[g]) return;
    const uint32_t code = dict[dict_off[g] + e];
    const uint64_t k0 = KA.off[2 * k + level], k1 = KA.off[2 * k + level + 1];
    uint64_t lo = k0, hi = k1;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (KA.code[mid] < code) lo = mid + 1; else hi = mid; }
    uint32_t cnt = 0;
    while (lo + cnt < k1 && cnt <= SP_MAXOCC && KA.code[lo + cnt] == code) ++cnt;
    hits[(size_t)k * max_dict + e] = (cnt == 0 || cnt > SP_MAXOCC) ? 0u : ((uint32_t)(lo - k0) | cnt << 24);
}

#ifndef K2_DICT_THREADS
#define K2_DICT_THREADS 256
#endif
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k2_anchor_dict_kernel(SeqSetView A, KmerIndexView KA, SeqSetView B, const uint32_t* __restrict__ c_idx, const uint32_t* __restrict__ a_idx,
                                                             uint64_t n_pairs, const uint32_t* __restrict__ ids, const uint64_t* __restrict__ id_off,
                                                             const uint32_t* __restrict__ hits, uint32_t max_dict, int32_t* __restrict__ diag_out, int32_t* __restrict__ votes_out,
                                                             int bins_cap, int tab_cap) {
    extern __shared__ uint32_t lds[];                 // [packed u16 vote bins][positions of the consensus' sorted 16-mer table]
    __shared__ unsigned long long red[THREADS / 64];
    __shared__ int spread[2];
    constexpr int AHEAD = 4096 / THREADS;          // rounds of a thread that cover an allele of 4 kb
    const int tid = threadIdx.x;
    int32_t* tab_pos = reinterpret_cast<int32_t*>(lds + ((bins_cap + 1) >> 1));
    int tab_c = -1;
    for (uint64_t p = blockIdx.x; p < n_pairs; p += gridDim.x) {
        const uint32_t c = c_idx[p], a = a_idx[p];
        const int m = A.len[c], n = B.len[a];
        const int nbins = m + n + 1;
        if (m < SP_KMER || n < SP_KMER || nbins > bins_cap) {
            if (tid == 0) { diag_out[p] = 0; votes_out[p] = 0; }
            continue;
        }
        const int nb32 = (nbins + 1) >> 1;
        // the ids of the allele and their hits are two dependent loads from memory: the first AHEAD rounds of a thread are issued
        // together and ahead of the histogram's clearing (an allele of 4 kb is AHEAD rounds of THREADS positions)
        const uint32_t* my_ids = ids + id_off[a];
        const uint32_t* my_hits = hits + (size_t)(c >> 1) * max_dict;
        uint32_t hv[AHEAD];
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) { const int j = tid + u * THREADS; hv[u] = j + SP_KMER <= n ? my_ids[j] : 0xFFFFFFFFu; }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) hv[u] = hv[u] != 0xFFFFFFFFu ? my_hits[hv[u]] : 0u;
        sp_anchor_clear<THREADS>(lds, nb32);
        if ((int)c != tab_c) {
            const uint64_t k0 = KA.off[c]; const int nk = (int)(KA.off[c + 1] - k0);
            for (int i = tid; i < nk && i < tab_cap; i += THREADS) tab_pos[i] = KA.pos[k0 + i];
            tab_c = (int)c;
        }
        __syncthreads();
        auto vote = [&](int j, uint32_t h) {
            const int cnt = (int)(h >> 24), first = (int)(h & 0xFFFFFFu);
            sp_anchor_vote_run(lds, cnt == 1 ? j - tab_pos[first] + m : -1);
            if (cnt > 1)
                for (int y = 0; y < cnt; ++y) {
                    const int bin = j - tab_pos[first + y] + m;
                    atomicAdd(&lds[bin >> 1], (bin & 1) ? 0x10000u : 1u);
                }
        };
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) if (u * THREADS + SP_KMER <= n) vote(tid + u * THREADS, hv[u]);       // (uniform condition: every lane of a round votes, with or without a bin)
        for (int j = tid + AHEAD * THREADS; j + SP_KMER <= n; j += THREADS) {
            const uint32_t id = my_ids[j];
            vote(j, id != 0xFFFFFFFFu ? my_hits[id] : 0u);
        }
        __syncthreads();
        sp_anchor_peaks<THREADS>(lds, nbins, m, 1, p, diag_out, votes_out, red, spread);
        __syncthreads();
    }
}

extern "C" {

void sp_hla_db_free(sp_hla_db* db) {
    if (!db) return;
    if (db->ctx) (void)hipSetDevice(db->ctx->device);
    sp_seqset_free(db->dna_gene); sp_seqset_free(db->cdna_gene); sp_seqset_free(db->dna_fwd); sp_seqset_free(db->ref_fwd); sp_seqset_free(db->ref_rev);
    (void)hipFree(db->d_gene_of); (void)hipFree(db->d_off_fwd); (void)hipFree(db->d_am); (void)hipFree(db->d_order); (void)hipFree(db->d_lcp); (void)hipFree(db->d_pos);
    (void)hipFree(db->d_hpc_ref); (void)hipFree(db->d_hpc_ref_off);
    for (auto& kv : db->gene_lists) { (void)hipFree(kv.second.d_idx); (void)hipFree(kv.second.d_l0); (void)hipFree(kv.second.d_l1); }
    sp_k2_dict_free(&db->kdict[0]); sp_k2_dict_free(&db->kdict[1]);
    sp_k1_seed_free(db->seed);
    delete db;
}

int32_t sp_hla_db_create(sp_ctx* ctx, const sp_hla_db_desc* d, sp_hla_db** out) {
    if (!ctx || !d || !out || !d->gene_of || !d->dna_off || !d->cdna_off || !d->gene_ref_off || !d->gene_fwd) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    sp_hla_db* db = new (std::nothrow) sp_hla_db();
    if (!db) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "hla db");
    db->ctx = ctx; db->n_alleles = d->n_alleles; db->n_genes = d->n_genes; db->ref_buffer = d->ref_buffer;
    db->gene_of.assign(d->gene_of, d->gene_of + d->n_alleles);
    db->gene_fwd.assign(d->gene_fwd, d->gene_fwd + d->n_genes);
    db->gene_alleles.resize(d->n_genes);
    if (d->exon_off && d->exon_start && d->exon_end) {
        db->exon_off.assign(d->exon_off, d->exon_off + d->n_genes + 1);
        db->exon_start.assign(d->exon_start, d->exon_start + db->exon_off[d->n_genes]);
        db->exon_end.assign(d->exon_end, d->exon_end + db->exon_off[d->n_genes]);
    } else db->exon_off.assign(d->n_genes + 1, 0);
    db->has_dna.resize(d->n_alleles);
    for (uint32_t a = 0; a < d->n_alleles; ++a) {
        if (db->gene_of[a] >= d->n_genes) { delete db; return sp_fail(ctx, SP_ERR_INVALID_ARG, "hla db: gene index out of range"); }
        db->gene_alleles[db->gene_of[a]].push_back(a);
        db->has_dna[a] = d->dna_off[a + 1] > d->dna_off[a];
    }
    int rc = sp_seqset_upload(ctx, d->dna, d->dna_off, d->n_alleles, &db->dna_gene);
    if (rc == SP_OK) rc = sp_seqset_upload(ctx, d->cdna, d->cdna_off, d->n_alleles, &db->cdna_gene);
    if (rc == SP_OK) rc = sp_seqset_upload(ctx, d->gene_ref, d->gene_ref_off, d->n_genes, &db->ref_fwd);
    std::string blob; std::vector<uint64_t> off(d->n_alleles + 1, 0);
    if (rc == SP_OK) {
        // hg38-oriented copy of every DNA allele (create_hla_fasta, realigner.rs:497-526)
        blob.reserve(d->dna_off[d->n_alleles]);
        for (uint32_t a = 0; a < d->n_alleles; ++a) {
            const char* s = d->dna + d->dna_off[a]; size_t len = d->dna_off[a + 1] - d->dna_off[a];
            if (db->gene_fwd[db->gene_of[a]]) blob.append(s, len); else blob += revcomp(s, len);
            off[a + 1] = blob.size();
        }
        rc = sp_seqset_upload(ctx, blob.data(), off.data(), d->n_alleles, &db->dna_fwd);
    }
    if (rc == SP_OK) {
        std::string rblob; std::vector<uint64_t> roff(d->n_genes + 1, 0);
        for (uint32_t g = 0; g < d->n_genes; ++g) { rblob += revcomp(d->gene_ref + d->gene_ref_off[g], d->gene_ref_off[g + 1] - d->gene_ref_off[g]); roff[g + 1] = rblob.size(); }
        rc = sp_seqset_upload(ctx, rblob.data(), roff.data(), d->n_genes, &db->ref_rev);
    }
    if (rc != SP_OK) { sp_hla_db_free(db); return rc; }
    if ((rc = sp_seqset_build_index(ctx, db->ref_fwd)) != SP_OK) { sp_hla_db_free(db); return rc; }
    if ((rc = sp_seqset_build_index(ctx, db->ref_rev)) != SP_OK) { sp_hla_db_free(db); return rc; }
    db->d_gene_of = dev_copy(db->gene_of);

    // hpc_pos tables of the gene references (homopolymers.rs:25-42)
    {
        std::vector<int32_t> hp; std::vector<uint64_t> hoff(d->n_genes + 1, 0);
        for (uint32_t g = 0; g < d->n_genes; ++g) {
            const char* s = d->gene_ref + d->gene_ref_off[g]; size_t len = d->gene_ref_off[g + 1] - d->gene_ref_off[g];
            hoff[g] = hp.size();
            int runs = 0;
            for (size_t p = 0; p <= len; ++p) {
                if (p > 0 && p < len && s[p] != s[p - 1]) ++runs;
                if (p == len && len > 0) { hp.push_back(runs + 1); break; }   // position >= len -> number of runs
                hp.push_back(runs);
            }
            if (len == 0) hp.push_back(0);
        }
        hoff[d->n_genes] = hp.size();
        db->d_hpc_ref = dev_copy(hp); db->d_hpc_ref_off = dev_copy(hoff);
    }

    // per-allele frame offset (allele_fwd vs buffered gene reference) and static allele->reference mapping
    const uint32_t n = d->n_alleles;
    std::vector<uint32_t> a_idx(n), b_idx(n);
    for (uint32_t a = 0; a < n; ++a) { a_idx[a] = db->gene_of[a]; b_idx[a] = a; }
    uint32_t* d_a = dev_copy(a_idx); uint32_t* d_b = dev_copy(b_idx);
    int32_t *d_diag = nullptr, *d_votes = nullptr;
    (void)hipMalloc(&d_diag, std::max<size_t>(1, n) * 4); (void)hipMalloc(&d_votes, std::max<size_t>(1, n) * 4);
    rc = sp_launch_anchor(ctx, db->ref_fwd, db->dna_fwd, d_a, d_b, n, d_diag, d_votes);
    std::vector<int32_t> diag(n), votes(n), off_fwd(n, SP_NO_DIAG);
    if (rc == SP_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(diag.data(), d_diag, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(votes.data(), d_votes, n * 4, hipMemcpyDeviceToHost);
        std::vector<CellDesc> cells(n);
        for (uint32_t a = 0; a < n; ++a) {
            bool ok = db->has_dna[a] && votes[a] >= K1_MIN_VOTES;
            if (ok) off_fwd[a] = diag[a];                       // allele_pos - ref_pos
            cells[a] = CellDesc{a, db->gene_of[a], ok ? -diag[a] : SP_NO_DIAG, SP_MAX_ED, 0, -1};   // A = allele (query), B = ref (target)
        }
        CellDesc* d_cells = dev_copy(cells); sp_aln* d_alns = nullptr;
        (void)hipMalloc(&d_alns, std::max<size_t>(1, n) * sizeof(sp_aln));
        rc = sp_launch_cells(ctx, db->dna_fwd, db->ref_fwd, d_cells, n, d_alns, nullptr, 0, "db_allele_ref", 2);
        if (rc == SP_OK) {
            (void)hipStreamSynchronize(ctx->stream);
            std::vector<sp_aln> alns(n);
            (void)hipMemcpy(alns.data(), d_alns, n * sizeof(sp_aln), hipMemcpyDeviceToHost);
            std::vector<int32_t> am(n * 3, 0);
            for (uint32_t a = 0; a < n; ++a) {
                const sp_aln& x = alns[a];
                // Forward-only, query-based penalised best mapping must beat 1.0 (realigner.rs:295-305)
                if (x.ok) {
                    double num = (double)(x.nm + (x.a_len - (x.a_end - x.a_start))); if (num < 0.1) num = 0.1;
                    if (num / (double)x.a_len < 1.0) { am[a * 3] = 1; am[a * 3 + 1] = x.a_start; am[a * 3 + 2] = x.b_start; }
                }
            }
            db->d_am = dev_copy(am);
        }
        (void)hipFree(d_cells); (void)hipFree(d_alns);
    }
    db->d_off_fwd = dev_copy(off_fwd);
    {
        // K1 visiting order: alleles that start alike sit next to each other, so a cell that runs out of edits inside the prefix it
        // shares with its successor settles the successor too (k1_cells_kernel).  Results never depend on the order.
        std::vector<uint32_t> order(n);
        for (uint32_t a = 0; a < n; ++a) order[a] = a;
        auto live = [&](uint32_t a) { return off[a + 1] > off[a] && off_fwd[a] != SP_NO_DIAG; };
        auto text = [&](uint32_t a) { return std::make_pair(blob.data() + off[a], (size_t)(off[a + 1] - off[a])); };
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            if (live(x) != live(y)) return live(x);
            if (!live(x)) return x < y;
            if (db->gene_of[x] != db->gene_of[y]) return db->gene_of[x] < db->gene_of[y];
            if (off_fwd[x] != off_fwd[y]) return off_fwd[x] < off_fwd[y];
            const auto tx = text(x), ty = text(y);
            const int c = std::memcmp(tx.first, ty.first, std::min(tx.second, ty.second));
            if (c != 0) return c < 0;
            if (tx.second != ty.second) return tx.second < ty.second;
            return x < y; });
        std::vector<int32_t> lcp(n, 0);
        for (uint32_t i = 1; i < n; ++i) {
            const uint32_t x = order[i - 1], y = order[i];
            if (!live(x) || !live(y) || db->gene_of[x] != db->gene_of[y] || off_fwd[x] != off_fwd[y]) continue;
            const auto tx = text(x), ty = text(y);
            const size_t m = std::min(tx.second, ty.second); size_t k = 0;
            while (k < m && tx.first[k] == ty.first[k]) ++k;
            lcp[i] = (int32_t)std::min<size_t>(k, 1u << 30);
        }
        db->d_order = dev_copy(order); db->d_lcp = dev_copy(lcp); db->h_order = order;
        // the same facts by position (K1Positions): length, frame offset, gene, first packed word, shared prefix
        std::vector<int32_t> posv((size_t)5 * n, 0);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t a = order[i];
            posv[i] = db->dna_fwd->h_len[a];
            posv[(size_t)n + i] = off_fwd[a];
            posv[(size_t)2 * n + i] = (int32_t)db->gene_of[a];
            posv[(size_t)3 * n + i] = (int32_t)(uint32_t)db->dna_fwd->h_word_off[a];
            posv[(size_t)4 * n + i] = lcp[i];
        }
        db->d_pos = dev_copy(posv);
    }
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_diag); (void)hipFree(d_votes);
    if (rc != SP_OK) { sp_hla_db_free(db); return rc; }
    *out = db;
    return SP_OK;
}

static int32_t k1_realign_chunk(sp_ctx* ctx, const sp_hla_db* db, const sp_seqset* reads, sp_hla_realign* out, uint32_t* cell_out);

static int32_t k1_seed_index(sp_ctx* ctx, const sp_hla_db* db) {
    std::lock_guard<std::mutex> guard(db->lazy);
    if (db->seed) return SP_OK;
    (void)hipSetDevice(ctx->device);
    return sp_k1_seed_build(ctx, db->dna_fwd, &db->seed);
}

int32_t sp_hla_seed_index_info(sp_ctx* ctx, const sp_hla_db* db, int64_t* out) {
    if (!ctx || !db || !out) return SP_ERR_INVALID_ARG;
    const int32_t rc = k1_seed_index(ctx, db);
    if (rc != SP_OK) return rc;
    sp_k1_seed_stats(db->seed, out);
    return SP_OK;
}

int32_t sp_seqset_sketch(sp_ctx* ctx, const sp_seqset* set, uint32_t idx, uint64_t* hash, int32_t* end_pos, uint8_t* strand, uint32_t cap, uint32_t* n_out) {
    if (!ctx || !set || !n_out || idx >= set->n) return SP_ERR_INVALID_ARG;
    return sp_k1_seed_sketch(ctx, set, idx, hash, end_pos, strand, cap, n_out);
}

int32_t sp_hla_realign_seeded_audit(sp_ctx* ctx, const sp_hla_db* db, const sp_seqset* reads, uint32_t read, int32_t* chains, uint32_t chain_cap, uint32_t* n_chains,
                                    sp_k1_seed_hit* hits, uint32_t* n_hits, int32_t* pick, uint64_t* counters) {
    if (!ctx || !db || !reads || read >= reads->n || !chains || !n_chains || !hits || !n_hits || !pick || !counters) return SP_ERR_INVALID_ARG;
    int32_t rc = k1_seed_index(ctx, db);
    if (rc != SP_OK) return rc;
    const uint32_t R = reads->n;
    int32_t* d_best = (int32_t*)sp_pool(ctx, "k1_best", (size_t)R * 4);
    sp_aln* d_win_aln = (sp_aln*)sp_pool(ctx, "k1s_win_aln", (size_t)R * sizeof(sp_aln)); sp_affine_aln* d_win_af = (sp_affine_aln*)sp_pool(ctx, "k1s_win_af", (size_t)R * sizeof(sp_affine_aln));
    sp_k1_seed_info* d_info = (sp_k1_seed_info*)sp_pool(ctx, "k1s_info", (size_t)R * sizeof(sp_k1_seed_info));
    if (!d_best || !d_win_aln || !d_win_af || !d_info) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "audit buffers");
    K1SeedDebug dbg; dbg.read = read; dbg.chains = chains; dbg.chain_cap = chain_cap; dbg.n_chains = n_chains; dbg.hits = hits; dbg.n_hits = n_hits; dbg.counters = counters;
    rc = sp_k1_seed_map(ctx, db->seed, db->dna_fwd, reads, ctx->k1_best_n > 0 ? ctx->k1_best_n : 5, d_best, d_info, d_win_aln, d_win_af, &dbg);
    if (rc != SP_OK) return rc;
    sp_k1_seed_info inf;
    SP_HIP_CHECK(ctx, hipMemcpyAsync(&inf, d_info + read, sizeof(inf), hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *pick = inf.pick;
    return SP_OK;
}

int32_t sp_hla_realign_reads(sp_ctx* ctx, const sp_hla_db* db, const sp_seqset* reads, sp_hla_realign* out, uint32_t* cell_out) {
    if (!ctx || !db || !reads || !out) return SP_ERR_INVALID_ARG;
    HostScope host_total(ctx, "host:k1_total");
    // big batches go through in slices of 65,536 reads (a shallow view of the same packed words): the read x allele matrix of a
    // slice is what bounds the device memory of a call, however many reads the caller hands over
    const char* env = std::getenv("SP_K1_SLICE");              // (tests shrink the slice to exercise this path)
    const uint32_t slice = env && std::atoi(env) > 0 ? (uint32_t)std::atoi(env) : 65536u;
    if (reads->n <= slice) return k1_realign_chunk(ctx, db, reads, out, cell_out);
    for (uint32_t r0 = 0; r0 < reads->n; r0 += slice) {
        const uint32_t k = std::min<uint32_t>(slice, reads->n - r0);
        sp_seqset part;
        part.ctx = reads->ctx; part.n = k; part.has_n = reads->has_n; part.max_len = reads->max_len;
        part.d_words = reads->d_words; part.d_nplane = reads->d_nplane; part.d_word_off = reads->d_word_off + r0; part.d_len = reads->d_len + r0;
        part.h_len.assign(reads->h_len.begin() + r0, reads->h_len.begin() + r0 + k);
        part.h_word_off.assign(reads->h_word_off.begin() + r0, reads->h_word_off.begin() + r0 + k + 1);
        const int32_t rc = k1_realign_chunk(ctx, db, &part, out + r0, cell_out ? cell_out + (size_t)r0 * db->n_alleles : nullptr);
        if (rc != SP_OK) return rc;
    }
    return SP_OK;
}

static int32_t k1_realign_chunk(sp_ctx* ctx, const sp_hla_db* db, const sp_seqset* reads, sp_hla_realign* out, uint32_t* cell_out) {
    const uint32_t R = reads->n, G = db->n_genes, NA = db->n_alleles;
    if (R == 0) return SP_OK;
    (void)hipSetDevice(ctx->device);
    // the reference's call pattern (seeds, chains, best_n: sp_hla_seed.hip) unless the caller wants every cell or has switched it off
    const bool seeded = ctx->k1_best_n > 0 && !cell_out;
    sp_aln* d_win_aln = nullptr; sp_affine_aln* d_win_af = nullptr; sp_k1_seed_info* d_seed_info = nullptr;
    if (seeded) {
        std::lock_guard<std::mutex> guard(db->lazy);
        if (!db->seed) { const int brc = sp_k1_seed_build(ctx, db->dna_fwd, &db->seed); if (brc != SP_OK) return brc; }
    }
    // 1. anchors read x gene
    uint32_t* d_a = (uint32_t*)sp_pool(ctx, "k1_a_idx", (size_t)R * G * 4);
    uint32_t* d_b = (uint32_t*)sp_pool(ctx, "k1_b_idx", (size_t)R * G * 4);
    int32_t* d_rg = (int32_t*)sp_pool(ctx, "k1_rg", (size_t)R * G * 4);
    int32_t* d_votes = (int32_t*)sp_pool(ctx, "k1_votes", (size_t)R * G * 4);
    int32_t* d_best = (int32_t*)sp_pool(ctx, "k1_best", (size_t)R * 4);
    // the read x allele matrix only exists when the caller asked for it (738 MB per 10,000 reads with the bundled database)
    uint32_t* d_cells = cell_out ? (uint32_t*)sp_pool(ctx, "k1_cells", (size_t)R * NA * 4) : nullptr;
    unsigned long long* d_win = cell_out ? nullptr : (unsigned long long*)sp_pool(ctx, "k1_winner", (size_t)R * 8);
    sp_hla_realign* d_out = (sp_hla_realign*)sp_pool(ctx, "k1_out", (size_t)R * sizeof(sp_hla_realign));
    int rc = SP_OK;
    if (!d_a || !d_b || !d_rg || !d_votes || !d_best || (cell_out ? !d_cells : !d_win) || !d_out || NA >= (1u << 24)) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign buffers");
    if (rc == SP_OK && d_win) (void)hipMemsetAsync(d_win, 0xFF, (size_t)R * 8, ctx->stream);
    // the (gene, read) pair list of the anchor and the empty bounds are written on the device: no host vectors, no waiting for copies
    unsigned long long* d_bound = nullptr;
    if (rc == SP_OK && !cell_out && !seeded) {
        d_bound = (unsigned long long*)sp_pool(ctx, "k1_bound", (size_t)R * 8);
        if (!d_bound) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign bound");
    }
    // (seeded mode: the anchors against the genes' references only serve the finalize stage -- the diagonal of the read's segment on the reference of the ACCEPTED allele's gene --
    //  and are made behind the seeded map, for that gene alone)
    if (rc == SP_OK && !seeded) hipLaunchKernelGGL(k1_init_kernel, dim3((R * G + 255) / 256), dim3(256), 0, ctx->stream, d_a, d_b, R, G, d_bound);
    if (rc == SP_OK && !seeded) rc = sp_launch_anchor(ctx, db->ref_fwd, reads, d_a, d_b, (uint64_t)R * G, d_rg, d_votes, 1, "anchor_k1", G);
    // reads with a weak forward anchor are listed now; their count comes back with the first pass's own host synchronisation
    uint32_t* d_weak_n = (uint32_t*)sp_pool(ctx, "k1_weak_n", 4);
    uint32_t* d_weak = (uint32_t*)sp_pool(ctx, "k1_weak", (size_t)R * 4);
    uint8_t* d_isrev = (uint8_t*)sp_pool(ctx, "k1_isrev", R);
    uint32_t n_weak = 0; bool weak_known = false;
    if (rc == SP_OK && (!d_weak_n || !d_weak || !d_isrev)) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign strand buffers");
    if (rc == SP_OK && !seeded) {
        (void)hipMemsetAsync(d_weak_n, 0, 4, ctx->stream); (void)hipMemsetAsync(d_isrev, 0, R, ctx->stream);
        hipLaunchKernelGGL(k1_weak_kernel, dim3((R + 255) / 256), dim3(256), 0, ctx->stream, d_votes, R, G, d_weak_n, d_weak);
    }
    const bool hasn = db->dna_fwd->has_n || reads->has_n || db->ref_fwd->has_n;
    // finalize: private per-wave windows of (allele, read) and (reference, read segment)
    int slot_words = std::max(sp_slot_words(db->dna_fwd, reads, hasn), sp_slot_words(db->ref_fwd, reads, hasn));
    const size_t lds_bytes = (size_t)slot_words * 16 + SP_LDS_TAIL;
    // cells: the read window and one allele slot of the single wavefront of a workgroup
    // (both multiples of 4 words: the allele slots take 16-byte LDS stores; a slot always holds a whole register prefetch)
    const int b_words = (((reads->max_len + 15) / 16 + 4) + 3) & ~3;
    const int a_words = std::max(K1_PRE_WORDS, (((db->dna_fwd->max_len + 15) / 16 + 4) + 3) & ~3);
    const size_t cells_lds = (size_t)((hasn ? 2 : 1) * (b_words + a_words)) * 4 + SP_LDS_TAIL;
    if (rc == SP_OK && (lds_bytes > 160 * 1024 - 64 || cells_lds > 160 * 1024 - 64)) rc = sp_fail(ctx, SP_ERR_TOO_LONG, "realign: window too long");
    // (exact branch-and-bound is switched off when the caller wants the full cell matrix: d_bound stays null)
    // Pruned mode runs the cells as exact iterative deepening: pass 1 caps every cell at 12 edits (plus the running
    // bound); reads whose best acceptable cell cannot be beaten by any unfinished cell are settled; the few others
    // are redone at 40 and then at the full 3 % cap.  The full-matrix mode is one un-pruned pass.
    uint8_t* d_done = nullptr; uint32_t* d_open = nullptr; uint32_t* d_open_list = nullptr; uint32_t* d_maxlen = nullptr;
    if (rc == SP_OK && d_bound) {
        d_done = (uint8_t*)sp_pool(ctx, "k1_done", R); d_open = (uint32_t*)sp_pool(ctx, "k1_open", 4);
        d_open_list = (uint32_t*)sp_pool(ctx, "k1_open_list", (size_t)R * 4); d_maxlen = (uint32_t*)sp_pool(ctx, "k1_maxlen", (size_t)R * 4);
        if (!d_done || !d_open || !d_open_list || !d_maxlen) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign done flags");
        else { (void)hipMemsetAsync(d_done, 0, R, ctx->stream); (void)hipMemsetAsync(d_maxlen, 0, (size_t)R * 4, ctx->stream); }
    }
    const int pass_caps[3] = {K1_CAP1, 40, SP_MAX_ED};
    const int n_pass = seeded ? 0 : d_bound ? 3 : 1;
    uint32_t n_open = R;
    if (rc == SP_OK && seeded) {
        d_win_aln = (sp_aln*)sp_pool(ctx, "k1s_win_aln", (size_t)R * sizeof(sp_aln)); d_win_af = (sp_affine_aln*)sp_pool(ctx, "k1s_win_af", (size_t)R * sizeof(sp_affine_aln));
        d_seed_info = (sp_k1_seed_info*)sp_pool(ctx, "k1s_info", (size_t)R * sizeof(sp_k1_seed_info));
        if (!d_win_aln || !d_win_af || !d_seed_info) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign buffers");
        else rc = sp_k1_seed_map(ctx, db->seed, db->dna_fwd, reads, ctx->k1_best_n, d_best, d_seed_info, d_win_aln, d_win_af, nullptr);
        if (rc == SP_OK) {
            hipLaunchKernelGGL(k1_seed_pairs_kernel, dim3((R * G + 255) / 256), dim3(256), 0, ctx->stream, d_best, db->d_gene_of, R, G, d_a, d_b);
            rc = sp_launch_anchor(ctx, db->ref_fwd, reads, d_a, d_b, (uint64_t)R * G, d_rg, d_votes, 1, "anchor_k1", G);
        }
        weak_known = true;
    }
    for (int pass = 0; pass < n_pass && rc == SP_OK; ++pass) {
        const int pass_cap = d_bound ? pass_caps[pass] : SP_MAX_ED;
        const uint32_t n_chunks = (((NA + K1_CHUNK - 1) / K1_CHUNK) + 7) & ~7u;       // padded to the 8 XCDs (k1_cells_kernel)
        const uint32_t* d_list = pass == 0 ? nullptr : d_open_list;
        // per-workgroup measurement words of the first (dominant) pass, summed into the context's counters after the launch
        uint2* d_wgs = nullptr; unsigned long long* d_cnt = nullptr;
        if (pass == 0 && ctx->profiling) {
            d_wgs = (uint2*)sp_pool(ctx, "k1_wg_stats", (size_t)n_open * n_chunks * sizeof(uint2)); d_cnt = sp_counters(ctx);
            if (d_wgs && d_cnt) (void)hipMemsetAsync(d_wgs, 0, (size_t)n_open * n_chunks * sizeof(uint2), ctx->stream); else d_wgs = nullptr;
        }
        {
            ProfScope ps(ctx, pass == 0 ? "k1_cells" : "k1_cells_deep", (uint64_t)n_open * NA);
            auto go = [&](auto kernel) {
                (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells_lds);
                K1Positions pos;
                pos.alen = db->d_pos; pos.off = db->d_pos + NA; pos.gene = reinterpret_cast<const uint32_t*>(db->d_pos + (size_t)2 * NA);
                pos.woff = reinterpret_cast<const uint32_t*>(db->d_pos + (size_t)3 * NA); pos.lcp = db->d_pos + (size_t)4 * NA;
                hipLaunchKernelGGL(kernel, dim3(n_open * n_chunks), dim3(64), cells_lds, ctx->stream, db->dna_fwd->view(), reads->view(), pos,
                                   d_rg, d_votes, (int)G, NA, n_chunks, d_cells, d_bound, d_list, d_maxlen, d_win, db->d_order, pass_cap, b_words, a_words, d_wgs);
            };
            if (hasn) { if (pass == 0) go(k1_cells_kernel<true, false>); else go(k1_cells_kernel<true, true>); }
            else { if (pass == 0) go(k1_cells_kernel<false, false>); else go(k1_cells_kernel<false, true>); }
            if (hipGetLastError() != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, "k1_cells launch failed");
        }
        if (d_wgs) hipLaunchKernelGGL(k1_stats_kernel, dim3(256), dim3(256), 0, ctx->stream, d_wgs, n_open * n_chunks, d_cnt);
        if (rc == SP_OK && d_bound && pass + 1 < n_pass) {
            (void)hipMemsetAsync(d_open, 0, 4, ctx->stream);
            hipLaunchKernelGGL(k1_done_kernel, dim3((R + 255) / 256), dim3(256), 0, ctx->stream, d_bound, d_done, R, pass_cap, d_maxlen, d_open, d_open_list);
            (void)hipMemcpyAsync(&n_open, d_open, 4, hipMemcpyDeviceToHost, ctx->stream);
            if (!weak_known) (void)hipMemcpyAsync(&n_weak, d_weak_n, 4, hipMemcpyDeviceToHost, ctx->stream);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, "k1 done sync");
            weak_known = true;
            if (n_open == 0) break;
        }
    }
    if (rc == SP_OK && !weak_known) {                          // (the cell-matrix mode has no synchronisation of its own before this point)
        (void)hipMemcpyAsync(&n_weak, d_weak_n, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, "k1 strand sync");
        weak_known = true;
    }
    if (rc == SP_OK && n_weak > 0) {
        const uint64_t np = (uint64_t)n_weak * G;
        uint32_t* d_a2 = (uint32_t*)sp_pool(ctx, "k1_rev_a", np * 4); uint32_t* d_b2 = (uint32_t*)sp_pool(ctx, "k1_rev_b", np * 4);
        int32_t* d_rg2 = (int32_t*)sp_pool(ctx, "k1_rev_rg", np * 4); int32_t* d_votes2 = (int32_t*)sp_pool(ctx, "k1_rev_votes", np * 4);
        if (!d_a2 || !d_b2 || !d_rg2 || !d_votes2) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign strand anchors");
        if (rc == SP_OK) {
            hipLaunchKernelGGL(k1_weak_pairs_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, d_weak, n_weak, G, d_a2, d_b2);
            rc = sp_launch_anchor(ctx, db->ref_rev, reads, d_a2, d_b2, np, d_rg2, d_votes2, 1, "anchor_k1_rev", G);
        }
        if (rc == SP_OK) hipLaunchKernelGGL(k1_reverse_kernel, dim3((n_weak + 255) / 256), dim3(256), 0, ctx->stream, d_weak, n_weak, G, d_votes, d_votes2, d_isrev);
    }
    if (rc == SP_OK && !seeded) {
        ProfScope ps(ctx, "k1_reduce", R);
        if (d_win) hipLaunchKernelGGL(k1_winner_kernel, dim3((R + 255) / 256), dim3(256), 0, ctx->stream, d_win, R, d_best);
        else hipLaunchKernelGGL(k1_reduce_kernel, dim3((R + 3) / 4), dim3(256), 0, ctx->stream, d_cells, db->dna_fwd->d_len, db->d_order, NA, R, d_best);
    }
    sp_aln* d_rm = nullptr;                                    // seeded mode: the placement of every read's segment on its gene's reference (for the second stage's re-score)
    if (rc == SP_OK && seeded) {
        d_rm = (sp_aln*)sp_pool(ctx, "k1_seg_rm", (size_t)R * sizeof(sp_aln));
        if (!d_rm) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign buffers");
        else (void)hipMemsetAsync(d_rm, 0, (size_t)R * sizeof(sp_aln), ctx->stream);
    }
    if (rc == SP_OK) {
        ProfScope ps(ctx, "k1_finalize", R);
        if (hasn) {
            (void)hipFuncSetAttribute((const void*)k1_finalize_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(k1_finalize_kernel<true>, dim3((R + 3) / 4), dim3(256), lds_bytes, ctx->stream, db->dna_fwd->view(), db->dna_gene->view(),
                               db->ref_fwd->view(), reads->view(), db->d_gene_of, db->d_off_fwd, d_rg, (int)G, db->d_am, db->d_hpc_ref, db->d_hpc_ref_off,
                               d_best, R, d_out, slot_words, d_win_aln, seeded ? d_win_af : nullptr, d_rm);
        } else {
            (void)hipFuncSetAttribute((const void*)k1_finalize_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(k1_finalize_kernel<false>, dim3((R + 3) / 4), dim3(256), lds_bytes, ctx->stream, db->dna_fwd->view(), db->dna_gene->view(),
                               db->ref_fwd->view(), reads->view(), db->d_gene_of, db->d_off_fwd, d_rg, (int)G, db->d_am, db->d_hpc_ref, db->d_hpc_ref_off,
                               d_best, R, d_out, slot_words, d_win_aln, seeded ? d_win_af : nullptr, d_rm);
        }
        if (hipGetLastError() != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, "k1_finalize launch failed");
    }
    if (rc == SP_OK && seeded) {
        // segments whose cell against the gene's reference left the 64-diagonal band: once more on the wide band
        ProfScope ps(ctx, "k1_seg_retry", R);
        CellDesc* d_sc = (CellDesc*)sp_pool(ctx, "k1_seg_cells", (size_t)R * sizeof(CellDesc));
        sp_aln* d_sa = (sp_aln*)sp_pool(ctx, "k1_seg_alns", (size_t)R * sizeof(sp_aln));
        if (!d_sc || !d_sa) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign buffers");
        else {
            hipLaunchKernelGGL(k1_seg_retry_cells_kernel, dim3((R + 255) / 256), dim3(256), 0, ctx->stream, d_out, R, reads->d_len, d_rg, (int)G, d_win_af, d_sc, d_sa);
            rc = sp_launch_cells_wide(ctx, db->ref_fwd, reads, d_sc, R, d_sa);
            if (rc == SP_OK) hipLaunchKernelGGL(k1_seg_retry_finish_kernel, dim3((R + 3) / 4), dim3(256), 0, ctx->stream, db->dna_gene->view(), db->ref_fwd->view(), db->d_am, db->d_hpc_ref,
                                                db->d_hpc_ref_off, d_sc, d_sa, d_win_af, R, d_out, d_rm);
            if (rc == SP_OK && hipGetLastError() != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, "k1 segment retry launch failed");
        }
    }
    if (rc == SP_OK && seeded) {
        // the second stage in the reference's numbers: the segments' placements re-scored (segment = query, reference = target), the records completed from those extents
        ProfScope ps(ctx, "k1_seg_rescore", R);
        const size_t words = (size_t)reads->h_word_off[R] + SP_SEQ_PAD_WORDS;
        CellDesc* d_c2 = (CellDesc*)sp_pool(ctx, "k1_segrs_cells", (size_t)R * sizeof(CellDesc));
        sp_aln* d_r2 = (sp_aln*)sp_pool(ctx, "k1_segrs_ref", (size_t)R * sizeof(sp_aln));
        int32_t* d_s0 = (int32_t*)sp_pool(ctx, "k1_segrs_start", (size_t)R * 4); int32_t* d_sl = (int32_t*)sp_pool(ctx, "k1_segrs_len", (size_t)R * 4);
        uint32_t* d_sw = (uint32_t*)sp_pool(ctx, "k1_segrs_words", words * 4);
        uint32_t* d_sn = reads->d_nplane ? (uint32_t*)sp_pool(ctx, "k1_segrs_nplane", words * 4) : nullptr;
        sp_affine_aln* d_saf = (sp_affine_aln*)sp_pool(ctx, "k1_segrs_af", (size_t)R * sizeof(sp_affine_aln));
        if (!d_c2 || !d_r2 || !d_s0 || !d_sl || !d_sw || (reads->d_nplane && !d_sn) || !d_saf) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign buffers");
        else {
            hipLaunchKernelGGL(k1_seg_rs_prep_kernel, dim3((R + 255) / 256), dim3(256), 0, ctx->stream, d_out, R, reads->d_len, d_win_af, d_rm, d_c2, d_r2, d_s0, d_sl);
            hipLaunchKernelGGL(k1_seg_slice_kernel, dim3((R + 3) / 4), dim3(256), 0, ctx->stream, reads->view(), d_s0, d_sl, R, d_sw, d_sn);
            sp_seqset segs;                                    // the segments as a set of their own: the reads' slots in a second buffer, their own lengths
            segs.ctx = ctx; segs.n = R; segs.has_n = reads->has_n; segs.d_words = d_sw; segs.d_nplane = d_sn; segs.d_word_off = reads->d_word_off; segs.d_len = d_sl; segs.max_len = reads->max_len;
            segs.h_len = reads->h_len; segs.h_word_off = reads->h_word_off;
            const sp_affine_opts ao = { 1, 4, 6, 2, 26, 1, 1 };
            // (ends only: the record takes the mapping's query span and target start; an HLA read is 40 - 100 clustered edits from the reference, and the DP over all of its
            //  rows cost 17.5 ms per 10,000 reads)
            constexpr int ends_knob = 64, band_knob = 256;      // bases from either end inside which an edit sends the placement to the DP; diagonals of that DP
            rc = sp_rescore_mappings(ctx, db->ref_fwd, &segs, d_c2, d_r2, R, true, ao, band_knob, d_saf, "k1_segrs", SP_MAX_ED + 1, 1, nullptr, nullptr, ends_knob);
            segs.d_words = nullptr; segs.d_nplane = nullptr; segs.d_word_off = nullptr; segs.d_len = nullptr;       // (pooled buffers: the set owns nothing)
            if (rc == SP_OK) hipLaunchKernelGGL(k1_seg_rs_finish_kernel, dim3((R + 3) / 4), dim3(256), 0, ctx->stream, db->dna_gene->view(), db->ref_fwd->view(), db->d_am, db->d_hpc_ref,
                                                db->d_hpc_ref_off, d_c2, d_s0, d_saf, d_win_af, R, d_out);
            if (rc == SP_OK && hipGetLastError() != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, "k1 segment re-score launch failed");
        }
    }
    // 3b. the winners re-scored the reference's way (two-piece affine gaps, end clipping): the numbers minimap2 reports for the read and its allele
    if (rc == SP_OK && seeded) hipLaunchKernelGGL(k1_seed_store_kernel, dim3((R + 255) / 256), dim3(256), 0, ctx->stream, d_out, R, d_seed_info, d_win_af);
    if (rc == SP_OK && ctx->mm2_rescore && !seeded) {
        CellDesc* d_rc = (CellDesc*)sp_pool(ctx, "k1_af_cells", (size_t)R * sizeof(CellDesc));
        sp_aln* d_ref = (sp_aln*)sp_pool(ctx, "k1_af_ref", (size_t)R * sizeof(sp_aln));
        sp_affine_aln* d_af = (sp_affine_aln*)sp_pool(ctx, "k1_af_out", (size_t)R * sizeof(sp_affine_aln));
        if (!d_rc || !d_ref || !d_af) rc = sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "realign buffers");
        else {
            hipLaunchKernelGGL(k1_rescore_cells_kernel, dim3((R + 255) / 256), dim3(256), 0, ctx->stream, d_out, R, d_rc, d_ref);
            const sp_affine_opts ao = { 1, 4, 6, 2, 26, 1, 1 };
            rc = sp_rescore_mappings(ctx, db->dna_fwd, reads, d_rc, d_ref, R, true, ao, 64, d_af, "k1_af", 128);
            if (rc == SP_OK) hipLaunchKernelGGL(k1_affine_store_kernel, dim3((R + 255) / 256), dim3(256), 0, ctx->stream, d_out, R, d_af);
        }
    }
    if (rc == SP_OK) {
        void* h_out = sp_host_pool(ctx, "k1_out", (size_t)R * sizeof(sp_hla_realign));
        uint8_t* h_isrev = n_weak ? (uint8_t*)sp_host_pool(ctx, "k1_isrev", R) : nullptr;
        if (h_isrev) (void)hipMemcpyAsync(h_isrev, d_isrev, R, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipMemcpyAsync(h_out ? h_out : (void*)out, d_out, (size_t)R * sizeof(sp_hla_realign), hipMemcpyDeviceToHost, ctx->stream);
        if (cell_out) (void)hipMemcpyAsync(cell_out, d_cells, (size_t)R * NA * 4, hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, std::string("realign: ") + hipGetErrorString(e));
        else if (h_out) std::memcpy(out, h_out, (size_t)R * sizeof(sp_hla_realign));
        // a read that anchors better on the reverse strand is dropped (the record keeps what the forward search found) -- unless it realigned acceptably forwards and the
        // reverse anchor leads by less than a factor of two: the reference drops a read only when minimap2's best SCORING mapping is reverse (realigner.rs:178-193), and a
        // read that passes the forward filters with a weak, repeat-ridden anchor is not one of those
        if (rc == SP_OK && h_isrev) for (uint32_t r = 0; r < R; ++r) if (h_isrev[r] == 2 || (h_isrev[r] == 1 && out[r].status != 0)) out[r].status = 2;
        if (rc == SP_OK && cell_out) {                        // the device rows are in visiting order: hand them out by allele index
            std::vector<uint32_t> row(NA);
            for (uint32_t r = 0; r < R; ++r) {
                uint32_t* c = cell_out + (size_t)r * NA;
                for (uint32_t p2 = 0; p2 < NA; ++p2) row[db->h_order[p2]] = c[p2];
                std::memcpy(c, row.data(), (size_t)NA * 4);
            }
        }
    }
    return rc;
}

// K2 for a batch of consensuses: every (consensus, allowed allele) pair of a level is one cell of one launch, one scan workgroup
// per consensus.  items: gene, gene-strand DNA consensus, spliced cDNA consensus.  stats (optional): per item n_alleles * 6.
// pair t of a K2 batch belongs to the item k with seg_off[k] <= t < seg_off[k + 1]: allele = t - seg_off[k] of the item's gene list, consensus
// set entries 2k (cDNA) and 2k + 1 (DNA)
__global__ __launch_bounds__(256) void k2_pairs_kernel(uint32_t T, uint32_t n_items, const uint32_t* __restrict__ seg_off, const uint32_t* const* __restrict__ lists,
                                                       uint32_t* __restrict__ idx, uint32_t* __restrict__ c0, uint32_t* __restrict__ c1) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    uint32_t lo = 0, hi = n_items;                                   // last k with seg_off[k] <= t
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_off[mid] <= t) lo = mid; else hi = mid; }
    idx[t] = lists[lo][t - seg_off[lo]]; c0[t] = 2 * lo; c1[t] = 2 * lo + 1;
}

struct K2Item { uint32_t gene; const char* dna; uint32_t dna_len; const char* cdna; uint32_t cdna_len; };
static int32_t k2_score_batch(sp_ctx* ctx, const sp_hla_db* db, uint32_t n_items, const K2Item* items, int32_t require_dna, int32_t disable_cdna,
                              sp_hla_best* best, int32_t* const* stats) {
    // "If cDNA scoring is disabled, require HLA DNA must be enabled" (caller.rs:517-520)
    if (disable_cdna && !require_dna) return sp_fail(ctx, SP_ERR_INVALID_ARG, "If cDNA scoring is disabled, require HLA DNA must be enabled");
    (void)hipSetDevice(ctx->device);
    // the samples of a cohort carry the same common alleles: consensuses that are the same gene and the same two strings are scored once
    if (n_items > 1 && !stats) {
        std::map<std::tuple<uint32_t, std::string, std::string>, uint32_t> seen;
        std::vector<K2Item> uniq; std::vector<uint32_t> as(n_items);
        for (uint32_t k = 0; k < n_items; ++k) {
            auto key = std::make_tuple(items[k].gene, std::string(items[k].dna, items[k].dna_len), std::string(disable_cdna ? "" : std::string(items[k].cdna, items[k].cdna_len)));
            auto it = seen.find(key);
            if (it == seen.end()) { it = seen.emplace(std::move(key), (uint32_t)uniq.size()).first; uniq.push_back(items[k]); }
            as[k] = it->second;
        }
        if (uniq.size() < n_items) {
            std::vector<sp_hla_best> ub(uniq.size());
            const int32_t urc = k2_score_batch(ctx, db, (uint32_t)uniq.size(), uniq.data(), require_dna, disable_cdna, ub.data(), nullptr);
            if (urc != SP_OK) return urc;
            for (uint32_t k = 0; k < n_items; ++k) best[k] = ub[as[k]];
            return SP_OK;
        }
    }
    HostMarks hm(ctx);
    // allowed alleles of each item's gene, database order (is_allowed_allele_def, caller.rs:1090-1095)
    std::vector<const sp_hla_db::GeneList*> lists(n_items);
    std::vector<uint32_t> seg_off(n_items + 1, 0);
    std::string blob; std::vector<uint64_t> coff(1, 0);
    for (uint32_t k = 0; k < n_items; ++k) {
        if (items[k].gene >= db->n_genes) return sp_fail(ctx, SP_ERR_INVALID_ARG, "score_consensus: gene out of range");
        std::lock_guard<std::mutex> guard(db->lazy);
        sp_hla_db::GeneList& gl = db->gene_lists[items[k].gene * 2 + (require_dna ? 1u : 0u)];
        if (gl.idx.empty() && !gl.d_idx) { for (uint32_t a : db->gene_alleles[items[k].gene]) if (db->has_dna[a] || !require_dna) gl.idx.push_back(a); gl.d_idx = dev_copy(gl.idx); }
        lists[k] = &gl;
        best[k].best_allele = -1; best[k].n_scored = (int32_t)gl.idx.size(); for (int x = 0; x < 6; ++x) best[k].mm2_stats[x] = -1;
        if (stats && stats[k]) for (uint32_t a = 0; a < db->n_alleles; ++a) for (int x = 0; x < 6; ++x) stats[k][(size_t)a * 6 + x] = -2;
        seg_off[k + 1] = seg_off[k] + (uint32_t)gl.idx.size();
        // consensus set: [2k] = cDNA, [2k+1] = DNA
        if (!disable_cdna) blob.append(items[k].cdna, items[k].cdna_len);
        coff.push_back(blob.size());
        blob.append(items[k].dna, items[k].dna_len);
        coff.push_back(blob.size());
    }
    const uint32_t T = seg_off[n_items];
    if (T == 0) return SP_OK;
    sp_seqset cons_set; sp_seqset* cons = &cons_set;       // pooled: no allocation, no free
    int rc = sp_seqset_make_small(ctx, "k2_cons", blob.data(), coff.data(), 2 * n_items, true, cons);
    if (rc != SP_OK) return rc;
    const uint32_t stride = K2_MAX_ED;
    // the pair lists (allele, cDNA consensus, DNA consensus per pair) are written on the device from the genes' allele lists, which already
    // sit there: what travels is one offset and one pointer per item (a 32-sample cohort has 1.2 M pairs: 14 MB of indices otherwise)
    const size_t head_bytes = (size_t)(n_items + 1) * 4 + 8 + (size_t)n_items * 8 + (size_t)n_items * 4;
    uint32_t* d_in = (uint32_t*)sp_pool(ctx, "k2_in", (size_t)3 * T * 4);
    uint8_t* d_head = (uint8_t*)sp_pool(ctx, "k2_head", head_bytes);
    uint8_t* h_head = (uint8_t*)sp_host_pool(ctx, "k2_head_stage", head_bytes);
    uint32_t* d_idx = d_in; uint32_t* d_c0 = d_in ? d_in + T : nullptr; uint32_t* d_c1 = d_in ? d_in + 2 * (size_t)T : nullptr;
    uint32_t* d_seg = (uint32_t*)d_head;
    const size_t lists_at = (((size_t)(n_items + 1) * 4 + 7) / 8) * 8, genes_at = lists_at + (size_t)n_items * 8;
    int32_t* d_diag = (int32_t*)sp_pool(ctx, "k2_diag", (size_t)T * 4); int32_t* d_votes = (int32_t*)sp_pool(ctx, "k2_votes", (size_t)T * 4);
    int32_t* d_best = (int32_t*)sp_pool(ctx, "k2_best", (size_t)n_items * 4);
    CellDesc* d_cells = (CellDesc*)sp_pool(ctx, "k2_cells", (size_t)T * sizeof(CellDesc));
    sp_aln* d_alns = (sp_aln*)sp_pool(ctx, "k2_alns", (size_t)2 * T * sizeof(sp_aln));
    uint32_t* d_ev = (uint32_t*)sp_pool(ctx, "k2_ev", (size_t)2 * T * stride * 4);
    K2Level* d_lv = (K2Level*)sp_pool(ctx, "k2_lv", (size_t)2 * T * sizeof(K2Level));
    if (!d_in || !d_head || !h_head || !d_diag || !d_votes || !d_best || !d_cells || !d_alns || !d_ev || !d_lv)
        return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "score_consensus buffers");
    std::memcpy(h_head, seg_off.data(), (size_t)(n_items + 1) * 4);
    for (uint32_t k = 0; k < n_items; ++k) { const uint32_t* lp = lists[k]->d_idx; std::memcpy(h_head + lists_at + (size_t)k * 8, &lp, 8); std::memcpy(h_head + genes_at + (size_t)k * 4, &items[k].gene, 4); }
    (void)hipMemcpyAsync(d_head, h_head, head_bytes, hipMemcpyHostToDevice, ctx->stream);    // (h_head is reused by the next call: the results' sync below covers it)
    hipLaunchKernelGGL(k2_pairs_kernel, dim3((T + 255) / 256), dim3(256), 0, ctx->stream, T, n_items, d_seg, (const uint32_t* const*)(d_head + lists_at), d_idx, d_c0, d_c1);
    const unsigned tb = 256, nb = (T + tb - 1) / tb;
    hm.mark("host:k2_setup");
    for (int L = 0; L < 2 && rc == SP_OK; ++L) {
        const sp_seqset* aset = L == 0 ? db->cdna_gene : db->dna_gene;
        const uint32_t* d_c = L == 0 ? d_c0 : d_c1;
        // anchors: through the allele set's 16-mer dictionary (built at the first call; the generic anchor kernel when it could not be built)
        K2Dict& kd = db->kdict[L];
        {
            std::lock_guard<std::mutex> guard(db->lazy);
            if (!kd.built && !kd.failed) { rc = sp_k2_dict_build(ctx, aset, db->d_gene_of, db->n_genes, &kd); if (rc != SP_OK) { rc = SP_OK; ctx->err.clear(); } }
        }
        const int bins_cap = cons->max_len + aset->max_len + 1;
        const size_t anchor_lds = (size_t)((bins_cap + 1) / 2) * 4 + (size_t)cons->max_len * 4;
        uint32_t* d_hits = kd.built && anchor_lds <= 160 * 1024 - 256 && n_items <= 65535 /* grid.y of k2_hits_kernel */
                               ? (uint32_t*)sp_pool(ctx, "k2_hits", std::max<size_t>(1, (size_t)n_items * kd.max_dict) * 4) : nullptr;
        if (d_hits && kd.max_dict) {
            ProfScope ps(ctx, "anchor_k2", T);
            hipLaunchKernelGGL(k2_hits_kernel, dim3((kd.max_dict + 255) / 256, n_items), dim3(256), 0, ctx->stream, cons->kview(), kd.d_code, kd.d_dict_off,
                               (const uint32_t*)(d_head + genes_at), L, kd.max_dict, d_hits);
            SP_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k2_anchor_dict_kernel<K2_DICT_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)anchor_lds));
            const uint64_t grid = std::min<uint64_t>(T, (uint64_t)ctx->num_cus * 16);
            hipLaunchKernelGGL(k2_anchor_dict_kernel<K2_DICT_THREADS>, dim3((unsigned)grid), dim3(K2_DICT_THREADS), anchor_lds, ctx->stream, cons->view(), cons->kview(), aset->view(), d_c, d_idx, (uint64_t)T,
                               kd.d_ids, kd.d_id_off, d_hits, kd.max_dict, d_diag, d_votes, bins_cap, cons->max_len);
            SP_HIP_CHECK(ctx, hipGetLastError());
        } else rc = sp_launch_anchor(ctx, cons, aset, d_c, d_idx, T, d_diag, d_votes, 1, "anchor_k2");
        if (rc != SP_OK) break;
        hipLaunchKernelGGL(k2_build_cells_kernel, dim3(nb), dim3(tb), 0, ctx->stream, d_idx, T, d_c, d_diag, d_votes, aset->d_len, d_cells);
        rc = sp_launch_cells(ctx, aset, cons, d_cells, T, d_alns + (size_t)L * T, d_ev + (size_t)L * T * stride, stride, L == 0 ? "k2_cells_cdna" : "k2_cells_dna", 1);
        if (rc != SP_OK) break;
        hipLaunchKernelGGL(k2_levels_kernel, dim3(nb), dim3(tb), 0, ctx->stream, d_alns + (size_t)L * T, T, d_lv + (size_t)L * T);
    }
    hm.mark("host:k2_launch");
    if (rc == SP_OK) {
        ProfScope ps(ctx, "k2_scan", T);
        hipLaunchKernelGGL(k2_scan_kernel, dim3(n_items), dim3(1024), 0, ctx->stream, d_lv, d_alns, d_ev, stride, T, d_seg, d_best);
    }
    if (rc == SP_OK && stats) {
        int32_t* d_stats = (int32_t*)sp_pool(ctx, "k2_stats", (size_t)db->n_alleles * 6 * 4);
        if (!d_stats) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "score_consensus stats");
        for (uint32_t k = 0; k < n_items; ++k) if (stats[k]) {
            const uint32_t n = seg_off[k + 1] - seg_off[k];
            (void)hipMemcpyAsync(d_stats, stats[k], (size_t)db->n_alleles * 6 * 4, hipMemcpyHostToDevice, ctx->stream);
            if (n) hipLaunchKernelGGL(k2_stats_kernel, dim3((n + tb - 1) / tb), dim3(tb), 0, ctx->stream, d_lv, T, seg_off[k], n, d_idx, d_stats);
            (void)hipMemcpyAsync(stats[k], d_stats, (size_t)db->n_alleles * 6 * 4, hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
        }
    }
    // the winners re-scored the reference's way (a = 5): what minimap2 reports for the best allele at both levels
    std::vector<sp_affine_aln> af;
    if (rc == SP_OK && ctx->mm2_rescore) {
        CellDesc* d_rc = (CellDesc*)sp_pool(ctx, "k2_af_cells", (size_t)n_items * sizeof(CellDesc));
        sp_aln* d_ref = (sp_aln*)sp_pool(ctx, "k2_af_ref", (size_t)n_items * sizeof(sp_aln));
        sp_affine_aln* d_af = (sp_affine_aln*)sp_pool(ctx, "k2_af_out", (size_t)2 * n_items * sizeof(sp_affine_aln));
        if (!d_rc || !d_ref || !d_af) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "score_consensus re-score buffers");
        const sp_affine_opts ao = { 5, 4, 6, 2, 26, 1, 1 };
        for (int L = 0; L < 2 && rc == SP_OK; ++L) {
            hipLaunchKernelGGL(k2_rescore_cells_kernel, dim3((n_items + 255) / 256), dim3(256), 0, ctx->stream, d_best, d_seg, n_items, d_idx, L == 0 ? d_c0 : d_c1,
                               d_alns + (size_t)L * T, d_rc, d_ref);
            rc = sp_rescore_mappings(ctx, L == 0 ? db->cdna_gene : db->dna_gene, cons, d_rc, d_ref, n_items, false, ao, 64, d_af + (size_t)L * n_items, L == 0 ? "k2_af0" : "k2_af1", stride);
        }
        if (rc == SP_OK) { af.resize((size_t)2 * n_items); (void)hipMemcpyAsync(af.data(), d_af, af.size() * sizeof(sp_affine_aln), hipMemcpyDeviceToHost, ctx->stream); }
    }
    if (rc == SP_OK) {
        std::vector<int32_t> b(n_items, -1);
        (void)hipMemcpyAsync(b.data(), d_best, (size_t)n_items * 4, hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = sp_fail(ctx, SP_ERR_HIP, std::string("score_consensus: ") + hipGetErrorString(e));
        else for (uint32_t k = 0; k < n_items; ++k) {
            best[k].best_allele = b[k] >= 0 ? (int32_t)lists[k]->idx[b[k]] : -1;
            for (int x = 0; x < 6; ++x) best[k].mm2_stats[x] = -1;
            if (best[k].best_allele >= 0 && !af.empty()) for (int L = 0; L < 2; ++L) {
                const sp_affine_aln& a = af[(size_t)L * n_items + k];
                if (a.score <= 0) continue;
                const int len = (L == 0 ? db->cdna_gene : db->dna_gene)->h_len[best[k].best_allele];
                best[k].mm2_stats[3 * L] = len; best[k].mm2_stats[3 * L + 1] = a.nm; best[k].mm2_stats[3 * L + 2] = len - (a.a_end - a.a_start);
            }
        }
    }
    hm.mark("host:k2_wait");
    return rc;
}

int32_t sp_hla_score_consensus(sp_ctx* ctx, const sp_hla_db* db, uint32_t gene,
                               const char* cons_dna, uint32_t cons_dna_len, const char* cons_cdna, uint32_t cons_cdna_len,
                               int32_t require_dna, int32_t disable_cdna, sp_hla_best* best, int32_t* stats) {
    if (!ctx || !db || !best || gene >= db->n_genes || (cons_dna_len && !cons_dna) || (cons_cdna_len && !cons_cdna)) return SP_ERR_INVALID_ARG;
    const K2Item item{ gene, cons_dna, cons_dna_len, cons_cdna, cons_cdna_len };
    int32_t* st[1] = { stats };
    return k2_score_batch(ctx, db, 1, &item, require_dna, disable_cdna, best, stats ? st : nullptr);
}

int32_t sp_hla_score_consensus_batch(sp_ctx* ctx, const sp_hla_db* db, uint32_t n, const uint32_t* genes,
                                     const char* const* cons_dna, const uint32_t* cons_dna_len, const char* const* cons_cdna, const uint32_t* cons_cdna_len,
                                     int32_t require_dna, int32_t disable_cdna, sp_hla_best* best) {
    if (!ctx || !db || (n && (!genes || !cons_dna || !cons_dna_len || !cons_cdna || !cons_cdna_len || !best))) return SP_ERR_INVALID_ARG;
    std::vector<K2Item> items(n);
    for (uint32_t k = 0; k < n; ++k) items[k] = K2Item{ genes[k], cons_dna[k], cons_dna_len[k], cons_cdna[k], cons_cdna_len[k] };
    return n ? k2_score_batch(ctx, db, n, items.data(), require_dna, disable_cdna, best, nullptr) : SP_OK;
}


// score_consensus + splice_read for a batch of hg38-forward consensuses: all placements on the gene references are one anchor launch
// and one traced cell launch, the splicing is host work on the event lists, and the typing is one k2_score_batch.
struct TypeItem { uint32_t gene; const char* cons; uint32_t len; };
static int32_t type_batch(sp_ctx* ctx, const sp_hla_db* db, uint32_t n_items, const TypeItem* items, int32_t require_dna, int32_t disable_cdna,
                          sp_hla_best* best, int32_t* const* stats, std::vector<std::string>* cdna_out) {
    (void)hipSetDevice(ctx->device);
    for (uint32_t k = 0; k < n_items; ++k) {
        if (items[k].gene >= db->n_genes || (items[k].len && !items[k].cons)) return sp_fail(ctx, SP_ERR_INVALID_ARG, "type_consensus: bad item");
        best[k].best_allele = -1; best[k].n_scored = 0;
        if (stats && stats[k]) for (uint32_t a = 0; a < db->n_alleles; ++a) for (int x = 0; x < 6; ++x) stats[k][(size_t)a * 6 + x] = -2;
    }
    if (cdna_out) cdna_out->assign(n_items, std::string());
    // 1. place every non-empty consensus on its un-buffered gene reference (an empty one is a failed consensus: unknown, caller.rs:1263-1267)
    std::vector<uint32_t> live;
    std::string blob; std::vector<uint64_t> off(1, 0);
    for (uint32_t k = 0; k < n_items; ++k) if (items[k].len) { live.push_back(k); blob.append(items[k].cons, items[k].len); off.push_back(blob.size()); }
    const uint32_t n = (uint32_t)live.size();
    if (n == 0) return SP_OK;
    sp_seqset cons_set; sp_seqset* cons = &cons_set;       // pooled: no allocation, no free
    int rc = sp_seqset_make_small(ctx, "tc_cons", blob.data(), off.data(), n, false, cons);
    if (rc != SP_OK) return rc;
    const int buffer = db->ref_buffer;
    std::vector<uint32_t> a_idx(n), b_idx(n);
    for (uint32_t x = 0; x < n; ++x) { a_idx[x] = items[live[x]].gene; b_idx[x] = x; }
    uint32_t* d_ab = (uint32_t*)sp_pool(ctx, "tc_idx", (size_t)2 * n * 4);
    int32_t* d_dv = (int32_t*)sp_pool(ctx, "tc_dv", (size_t)2 * n * 4);
    CellDesc* d_cell = (CellDesc*)sp_pool(ctx, "tc_cell", (size_t)n * sizeof(CellDesc));
    sp_aln* d_aln = (sp_aln*)sp_pool(ctx, "tc_aln", (size_t)n * sizeof(sp_aln));
    uint32_t* d_ev = (uint32_t*)sp_pool(ctx, "tc_ev", (size_t)n * K2_MAX_ED * 4);
    if (!d_ab || !d_dv || !d_cell || !d_aln || !d_ev) return sp_fail(ctx, SP_ERR_OUT_OF_MEMORY, "type_consensus buffers");
    (void)hipMemcpyAsync(d_ab, a_idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(d_ab + n, b_idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    rc = sp_launch_anchor(ctx, db->ref_fwd, cons, d_ab, d_ab + n, n, d_dv, d_dv + n, 1, "anchor_type");
    if (rc != SP_OK) return rc;
    std::vector<int32_t> dv((size_t)2 * n);
    (void)hipMemcpyAsync(dv.data(), d_dv, (size_t)2 * n * 4, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    std::vector<CellDesc> cells(n);
    for (uint32_t x = 0; x < n; ++x) {
        const uint32_t gene = items[live[x]].gene;
        const int v_lo = buffer, v_hi = db->ref_fwd->h_len[gene] - buffer;          // region_sequence has no buffer (caller.rs:651-654)
        if (v_hi <= v_lo) return sp_fail(ctx, SP_ERR_INVALID_ARG, "type_consensus: reference shorter than its buffer");
        // dv[x] = cons_pos - ref_pos (buffered reference); cell: A = consensus, B = reference view [v_lo, v_hi)
        cells[x] = CellDesc{ x, gene, dv[n + x] >= K2_MIN_VOTES ? -dv[x] - v_lo : SP_NO_DIAG, K2_MAX_ED, v_lo, v_hi };
    }
    (void)hipMemcpyAsync(d_cell, cells.data(), (size_t)n * sizeof(CellDesc), hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemsetAsync(d_ev, 0, (size_t)n * K2_MAX_ED * 4, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    rc = sp_launch_cells(ctx, cons, db->ref_fwd, d_cell, n, d_aln, d_ev, K2_MAX_ED, "type_consensus_ref", 2);
    if (rc != SP_OK) return rc;
    std::vector<sp_aln> alns(n); std::vector<uint32_t> evs((size_t)n * K2_MAX_ED);
    (void)hipMemcpyAsync(alns.data(), d_aln, (size_t)n * sizeof(sp_aln), hipMemcpyDeviceToHost, ctx->stream);
    (void)hipMemcpyAsync(evs.data(), d_ev, (size_t)n * K2_MAX_ED * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return sp_fail(ctx, SP_ERR_HIP, "type_consensus placement");
    // 2. splice, put on the gene strand
    std::vector<std::string> dna_g(n), cdna_g(n); std::vector<K2Item> k2; std::vector<uint32_t> k2_of;
    for (uint32_t x = 0; x < n; ++x) {
        const TypeItem& it = items[live[x]];
        const uint32_t gene = it.gene;
        const sp_aln& aln = alns[x]; const uint32_t* ev = evs.data() + (size_t)x * K2_MAX_ED;
        const int v_lo = buffer, v_hi = db->ref_fwd->h_len[gene] - buffer, tlen = v_hi - v_lo;
        // select_best_mapping(target-based, penalised) must beat the 1.0 default (util/mapping.rs:22-57, caller.rs:1289-1297)
        bool mapped = aln.ok != 0;
        if (mapped) { double num = (double)(aln.nm + (tlen - (aln.b_end - aln.b_start))); if (num < 0.1) num = 0.1; mapped = num / (double)tlen < 1.0; }
        if (!mapped) continue;                                             // "Failed to align consensus to reference genome" (caller.rs:1282-1287)
        // aligned pairs of the consensus record: reference view position -> consensus position (M columns only)
        std::vector<int32_t> lookup((size_t)tlen, -1);
        {
            int i = aln.a_start, j = aln.b_start;
            for (int e = 0; e <= aln.nm; ++e) {
                const int jn = e < aln.nm ? (int)(ev[e] & 0x3FFFFFFFu) : aln.b_end;
                for (; j < jn; ++j, ++i) lookup[j] = i;                   // '=' run
                if (e == aln.nm) break;
                const uint32_t type = ev[e] >> 30;
                if (type == SP_EV_X) { lookup[j] = i; ++i; ++j; }        // mismatch is still an aligned pair (cigar M)
                else if (type == SP_EV_D) { ++j; }                         // reference base without consensus base
                else { ++i; }                                              // inserted consensus base
            }
        }
        // splice_read (caller.rs:1518-1576) with exons relative to the view
        std::string spliced;
        auto has = [&](int64_t p2) { return p2 >= 0 && p2 < tlen && lookup[(size_t)p2] >= 0; };
        for (uint32_t e = db->exon_off[gene]; e < db->exon_off[gene + 1]; ++e) {
            int64_t first = (int64_t)db->exon_start[e] - buffer, last = (int64_t)db->exon_end[e] - buffer - 1;
            while (!has(first) && first <= last) ++first;
            while (!has(last) && first <= last) --last;
            if (first <= last) spliced.append(it.cons + lookup[(size_t)first], (size_t)(lookup[(size_t)last] + 1 - lookup[(size_t)first]));
        }
        // gene strand (caller.rs:1344-1363); no exon bases => cDNA "N"
        const bool fwd = db->gene_fwd[gene] != 0;
        dna_g[x] = fwd ? std::string(it.cons, it.len) : revcomp(it.cons, it.len);
        cdna_g[x] = spliced.empty() ? std::string("N") : (fwd ? spliced : revcomp(spliced.data(), spliced.size()));
        if (cdna_out) (*cdna_out)[live[x]] = cdna_g[x];
        k2.push_back(K2Item{ gene, dna_g[x].data(), (uint32_t)dna_g[x].size(), cdna_g[x].data(), (uint32_t)cdna_g[x].size() });
        k2_of.push_back(live[x]);
    }
    if (k2.empty()) return SP_OK;
    std::vector<sp_hla_best> b2(k2.size()); std::vector<int32_t*> st2(k2.size(), nullptr);
    if (stats) for (size_t y = 0; y < k2.size(); ++y) st2[y] = stats[k2_of[y]];
    rc = k2_score_batch(ctx, db, (uint32_t)k2.size(), k2.data(), require_dna, disable_cdna, b2.data(), stats ? st2.data() : nullptr);
    if (rc != SP_OK) return rc;
    for (size_t y = 0; y < k2.size(); ++y) best[k2_of[y]] = b2[y];
    return SP_OK;
}

int32_t sp_hla_type_consensus(sp_ctx* ctx, const sp_hla_db* db, uint32_t gene,
                              const char* consensus_fwd, uint32_t consensus_len,
                              int32_t require_dna, int32_t disable_cdna,
                              sp_hla_best* best, int32_t* stats,
                              char* cdna_out, uint32_t cdna_cap, uint32_t* cdna_len) {
    if (!ctx || !db || !best || gene >= db->n_genes || (consensus_len && !consensus_fwd)) return SP_ERR_INVALID_ARG;
    if (cdna_len) *cdna_len = 0;
    const TypeItem item{ gene, consensus_fwd, consensus_len };
    int32_t* st[1] = { stats };
    std::vector<std::string> cd;
    const int32_t rc = type_batch(ctx, db, 1, &item, require_dna, disable_cdna, best, stats ? st : nullptr, &cd);
    if (rc == SP_OK && !cd.empty()) {
        if (cdna_len) *cdna_len = (uint32_t)cd[0].size();
        if (cdna_out && cdna_cap) memcpy(cdna_out, cd[0].data(), std::min<size_t>(cdna_cap, cd[0].size()));
    }
    return rc;
}

int32_t sp_hla_type_consensus_batch(sp_ctx* ctx, const sp_hla_db* db, uint32_t n, const uint32_t* genes,
                                    const char* const* consensus_fwd, const uint32_t* consensus_len,
                                    int32_t require_dna, int32_t disable_cdna, sp_hla_best* best) {
    if (!ctx || !db || (n && (!genes || !consensus_fwd || !consensus_len || !best))) return SP_ERR_INVALID_ARG;
    std::vector<TypeItem> items(n);
    for (uint32_t k = 0; k < n; ++k) items[k] = TypeItem{ genes[k], consensus_fwd[k], consensus_len[k] };
    return n ? type_batch(ctx, db, n, items.data(), require_dna, disable_cdna, best, nullptr, nullptr) : SP_OK;
}

} // extern "C"

#ifdef SP_K1_STATS
extern "C" int32_t sp_debug_wfa_stats(uint64_t* out, int32_t reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wfa_stats), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wfa_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
