// sp_wfa.hip.h -- the wavefront-alignment cell for gfx950: ONE 64-lane wavefront per (A, B) cell,
// lane l <-> diagonal (diag - 32 + l).  Device-side mirror of the contract in DESIGN.md section 3
// (restated for the CPU in oracle/align.c; the two must agree bit for bit).
//
// Replaces the base-level alignment inside minimap2::Aligner::map at every hot-path call site of the
// reference (src/hla/realigner.rs:116,231,290; src/hla/caller.rs:1277,1436; src/cyp2d6/haplotyper.rs:198,395;
// src/cyp2d6/chaining.rs:58).
//
// Data movement: both sequences are 2-bit packed in HBM (16 bases per dword).  A wavefront stages only the
// band-reachable window of each sequence into its private LDS slot with coalesced dword loads (64 dwords =
// 1,024 bases per wave instruction), then runs entirely out of LDS + registers: the whole DP state is ONE
// VGPR per lane (furthest-reaching A position on that diagonal) plus one VGPR for the origin diagonal.
#pragma once
#include "sp_internal.h"

#ifdef SP_K1_STATS
// profiling build only (profiles/scripts/k1_stats.sh): wave-level event counters of the DP core
static __device__ unsigned long long g_wfa_stats[64];
#define SP_STAT(i, v) do { if (lane == 0) atomicAdd(&g_wfa_stats[i], (unsigned long long)(v)); } while (0)
#else
#define SP_STAT(i, v) do { } while (0)
#endif

namespace spw {

// Positions inside the core are kept in BIT units (2 x base position): the alignbit shift is then the position itself (the
// instruction reads its low 5 bits), the first set bit of the mismatch mask is the match length, nothing is shifted back and forth.
// The windows are addressed by their LDS byte address held in an SGPR: word address = scalar base + 4 * (p2 >> 5) is one
// shift-add per load (a generic pointer makes the compiler rebuild "lds base + constant + index" with extra adds every time).
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
__device__ __forceinline__ uint32_t lds_addr(const uint32_t* p) {
    uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_cu32*)p);
    asm("" : "+s"(a));        // opaque: otherwise a compile-time-known window offset is re-added as literals at every use
    return a;
}
__device__ __forceinline__ uint32_t load16b(uint32_t base, int p2) {
    // 16 bases starting at bit position p2 of an LDS-resident packed sequence (guard word guaranteed)
    uint32_t addr;                                  // base + 4 * (p2 >> 5): shift, then ONE shift-add (the compiler's canonical
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(addr) : "v"(p2 >> 5), "s"(base));        // form, (p2 >> 3) & ~3 then add, takes three)
    lds_cu32* w = (lds_cu32*)(uintptr_t)addr;
    return __builtin_amdgcn_alignbit(w[1], w[0], (uint32_t)p2);
}

// v_ffbl_b32 as the hardware defines it: index of the lowest set bit, 0xFFFFFFFF for 0 (the C builtins are undefined there and
// cost a compare + select to make defined; the caller clamps with an unsigned min instead)
__device__ __forceinline__ uint32_t ffbl_raw(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }

// matched length in bit units, at most 32 (16 bases) and at most rem2 (> 0)
template <bool HASN>
__device__ __forceinline__ int match16b(uint32_t LA, uint32_t NA, int pa2, uint32_t LB, uint32_t NB, int pb2, int rem2) {
    const uint32_t x = load16b(LA, pa2) ^ load16b(LB, pb2);
    uint32_t mm = (x | (x >> 1)) & 0x55555555u;
    if (HASN) mm |= (load16b(NA, pa2) | load16b(NB, pb2));
    const uint32_t f = ffbl_raw(mm);
    const uint32_t c = f < 32u ? f : 32u;
    return (int)(c < (uint32_t)rem2 ? c : (uint32_t)rem2);
}

struct CellIn {
    const uint32_t* a_words; const uint32_t* a_nplane; int a0, a1;   // A view [a0,a1) in full-sequence bases
    const uint32_t* b_words; const uint32_t* b_nplane; int b0, b1;   // B view
    int diag;      // (b view pos) - (a view pos)
    int max_ed;
};

struct CellOut {
    int ok, nm, a_start, a_end, b_start, b_end;
    int explored;      // ok == 0 (edit cap exhausted): the largest A position any diagonal reached, i.e. the last A base the run looked at
};

// stage bases [lo,hi) (full-sequence coordinates) of a packed sequence into dst; returns the base index of dst[0]
__device__ __forceinline__ int stage(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int lo, int hi, int lane) {
    int w0 = lo >> 4;
    int nw = ((hi + 15) >> 4) - w0 + 2;            // + guard words (the source carries >= 2 zero guard words)
    for (int w = lane; w < nw; w += SP_WAVE) dst[w] = src[w0 + w];
    return w0 << 4;
}

__device__ __forceinline__ int words_needed(int lo, int hi) { return ((hi + 15) >> 4) - (lo >> 4) + 2; }

// neighbour exchange over the whole 64-lane wavefront with DPP (no LDS traffic):
//   from_lower(x, fill): lane l receives lane l-1 (lane 0 receives fill)    -- wave_shr:1
//   from_upper(x, fill): lane l receives lane l+1 (lane 63 receives fill)   -- wave_shl:1
__device__ __forceinline__ int from_lower(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int from_upper(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x130, 0xf, 0xf, false); }

// max(acc, neighbour's x) in ONE instruction: the DPP operand rides on v_max_i32; a lane without that neighbour keeps acc
// (s_nop 4: a DPP read needs two wait states after a VALU write of that VGPR and five after a VALU write of EXEC, and the
// compiler's hazard recogniser does not look inside inline asm -- the nop covers the worst case)
__device__ __forceinline__ int max_from_lower(int x, int acc) {
    asm("s_nop 4\n\tv_max_i32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x));
    return acc;
}
__device__ __forceinline__ int max_from_upper(int x, int acc) {
    asm("s_nop 4\n\tv_max_i32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x));
    return acc;
}

// max over the 64 lanes, wave-uniform result: four DPP butterflies inside each row of 16 (quad swaps, half mirror, mirror --
// every lane has a partner, no fill needed), then the four row results meet on the scalar side
__device__ __forceinline__ int wave_max(int v) {
    int r;
    asm("s_nop 4\n\tv_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(v));
    const int r0 = __builtin_amdgcn_readlane(r, 0), r1 = __builtin_amdgcn_readlane(r, 16);
    const int r2 = __builtin_amdgcn_readlane(r, 32), r3 = __builtin_amdgcn_readlane(r, 48);
    const int x = r0 > r1 ? r0 : r1, y = r2 > r3 ? r2 : r3;
    return x > y ? x : y;
}

// The DP core on two LDS-resident windows.  All 64 lanes call it together.
//   LA/NA, a_sh : packed A window (+ N plane) and the LDS base position of view position 0 (may be negative)
//   m, n        : view lengths; kb = diagonal of lane 0 (view coordinates, b - a)
//   NEED_O      : track the origin diagonal even when it cannot influence a_start (needed for b_start)
// hist (TRACE only): this wave's global scratch, (max_ed+1)*64 uint16.  events (TRACE only): >= max_ed words.
// Snapshot / resume (K1 only): a run on another A that shares its first L bases with this one, on the same B and diagonal, is
// this run step for step while no lane has gone past A[L-1] (a lane parked at H has compared A[H] and nothing beyond).  With
// thr2 = 2L the core hands back the last such state (out_s steps done, out_H per lane; out_s < 0: none), and a run given
// in_s >= 0 starts from (in_s, in_H) instead of from scratch.  Only used when the origin is not tracked.
struct Snap { int thr2, in_s, in_H, out_s, out_H; };

template <bool TRACE, bool HASN, bool NEED_O, bool SNAP = false>
__device__ __forceinline__ void wfa_core(const uint32_t* __restrict__ LA_, const uint32_t* __restrict__ NA_, int a_sh, int m,
                                         const uint32_t* __restrict__ LB_, const uint32_t* __restrict__ NB_, int b_sh, int n,
                                         int kb_, int max_ed_, int lane,
                                         uint16_t* __restrict__ hist, uint32_t* __restrict__ events, CellOut& out,
                                         const int snap_thr2 = 0, const int snap_in_s = -1, const int snap_in_H = 0,
                                         int* __restrict__ snap_out_s = nullptr, int* __restrict__ snap_out_H = nullptr) {
    // every cell parameter is wave-uniform; pin them to SGPRs so the step loop is scalar control flow
    m = __builtin_amdgcn_readfirstlane(m); n = __builtin_amdgcn_readfirstlane(n);
    a_sh = __builtin_amdgcn_readfirstlane(a_sh); b_sh = __builtin_amdgcn_readfirstlane(b_sh);
    const int kb = __builtin_amdgcn_readfirstlane(kb_), max_ed = __builtin_amdgcn_readfirstlane(max_ed_);
    const uint32_t LA = lds_addr(LA_), LB = lds_addr(LB_), NA = HASN ? lds_addr(NA_) : 0u, NB = HASN ? lds_addr(NB_) : 0u;
    const uint32_t lane4 = (uint32_t)lane << 2;
    const int k = kb + lane;
    // (SNAP callers only come with kb >= 0: the origin diagonal is then never needed and the branch on it leaves the step loop)
    const bool track = NEED_O || TRACE || (!SNAP && kb < 0);
    // per-lane constants (bit units): the furthest A position diagonal k can hold -- a live lane sits there exactly when it has
    // reached the last row or column -- and the offsets of the two windows
    const int lim = m < n - k ? m : n - k;
    const int lim2 = lim << 1;
    const int a2 = a_sh << 1, b2 = (k + b_sh) << 1;
    const int NEG2 = SP_NEG * 2;

    // extension of every lane from bit position i (negative = dead lane, left as it is: dead values only ever grow by 2 per
    // step from SP_NEG * 2 and stay far below 0); returns the new furthest position
    // the kept cooperative scan (see extend): diagonal (lane), bit position of its block 0, blocks that hold a stop, stop offsets
    int scan_src = -1, scan_base = 0, scan_fm = 0;
    uint64_t scan_stop = 0;
    auto extend = [&](int i) -> int {
        // every lane compares, dead lanes included (their addresses fall outside the LDS allocation: such reads return 0) and are
        // masked afterwards; a live lane has 0 <= rem, and rem = 0 (parked on the last row / column) clamps its match to 0
        int rem = lim2 - i;
        int nmv = match16b<HASN>(LA, NA, i + a2, LB, NB, i + b2, rem);
        nmv = i >= 0 ? nmv : 0;
        i += nmv;
        bool going = nmv == 32 && rem > 32;
        rem -= nmv;
        // lanes still matching after 16 bases finish cooperatively: all 64 lanes compare 1,024 bases per step.  The stretch is cut
        // into blocks of 16 bases aligned in A (the block the stretch starts in was compared up to the start a moment ago: nothing
        // to mask); lane l takes block l: the word index is (scalar word of block 0) + l and the alignbit shift is scalar, so the
        // per-lane work is two adds, the compare and the clamp.  Lanes past the end of the stretch clamp to 0 (their loads may touch
        // words behind the staged window: still inside this workgroup's LDS, values unused).
        // One scan tells where EVERY block of those 1,024 bases first differs.  It is kept (diagonal, first block, stop mask, offsets
        // per lane), and the next long stretch on that diagonal -- after a substitution the alignment stays on it -- is answered
        // from the kept scan: furthest-reaching points only grow, and whatever lies between a block's start and the new stretch
        // start has just been compared equal, so the kept first stop of a block is its first stop at or behind the new start.
        uint64_t longmask = __ballot(going);
        while (longmask) {
            const int src = __builtin_ctzll(longmask);
            longmask &= longmask - 1;
            SP_STAT(5, 1);
            const int ci = __builtin_amdgcn_readlane(i, src);
            const int crem = __builtin_amdgcn_readlane(rem, src);
            int stop_at = -1;
            if (src == scan_src && (unsigned)(ci - scan_base) < (unsigned)(SP_WAVE * 32)) {
                const int blk = (ci - scan_base) >> 5;
                const uint64_t later = scan_stop >> blk;
                if (later) { const int t = blk + __builtin_ctzll(later); stop_at = scan_base + (t << 5) + __builtin_amdgcn_readlane(scan_fm, t); SP_STAT(13, 1); }
            }
            if (stop_at < 0) {
                const int cb2 = (kb + src + b_sh) << 1;
                int ca = ci & ~31;
                int left = crem + (ci - ca);                                 // bits from block 0 to the end of the diagonal's range
                for (;;) {
                    SP_STAT(6, 1);
                    const int pa = ca + a2, pb = ca + cb2;                   // scalar bit positions of block 0 in the two windows
                    const uint32_t oa = (uint32_t)((pa >> 5) << 2), ob = (uint32_t)((pb >> 5) << 2);      // scalar
                    lds_cu32* wa = (lds_cu32*)(uintptr_t)((LA + oa) + lane4);
                    lds_cu32* wb = (lds_cu32*)(uintptr_t)((LB + ob) + lane4);
                    const uint32_t x = __builtin_amdgcn_alignbit(wa[1], wa[0], (uint32_t)pa) ^ __builtin_amdgcn_alignbit(wb[1], wb[0], (uint32_t)pb);
                    uint32_t mm = (x | (x >> 1)) & 0x55555555u;
                    if (HASN) {
                        lds_cu32* na = (lds_cu32*)(uintptr_t)((NA + oa) + lane4); lds_cu32* nb = (lds_cu32*)(uintptr_t)((NB + ob) + lane4);
                        mm |= __builtin_amdgcn_alignbit(na[1], na[0], (uint32_t)pa) | __builtin_amdgcn_alignbit(nb[1], nb[0], (uint32_t)pb);
                    }
                    int r = left - (lane << 5); r = r > 0 ? r : 0;
                    const uint32_t f = ffbl_raw(mm);
                    const uint32_t c = f < 32u ? f : 32u;
                    const int fm = (int)(c < (uint32_t)r ? c : (uint32_t)r);
                    const uint64_t stop = __ballot(fm < 32);
                    if (stop) {
                        const int t = __builtin_ctzll(stop);
                        stop_at = ca + (t << 5) + __builtin_amdgcn_readlane(fm, t);
                        scan_src = src; scan_base = ca; scan_stop = stop; scan_fm = fm;
                        break;
                    }
                    ca += SP_WAVE * 32; left -= SP_WAVE * 32;
                }
            }
            if (lane == src) i = stop_at;
        }
        return i;
    };

    // s = 0 : every band diagonal may start for free on the first row / first column
    int H, O = lane;
    int s = 0, end_lane = -1;
    bool below = SNAP;
    int keep_s = -1, keep_H = 0;
    if (SNAP && __builtin_amdgcn_readfirstlane(snap_in_s) >= 0) {
        H = snap_in_H; s = __builtin_amdgcn_readfirstlane(snap_in_s);
    } else {
        const int i0 = k < 0 ? -k : 0;
        H = extend(i0 < lim ? i0 << 1 : NEG2);
    }
    if (TRACE) hist[lane] = (uint16_t)(H >= 0 ? H >> 1 : 0xFFFF);
    for (;;) {
        if (SNAP && below) {
            if (__ballot(H >= snap_thr2)) below = false; else { keep_H = H; keep_s = s; }
        }
        const bool reached = H == lim2;                   // dead lanes are negative and lim2 > NEG2 + 2 * SP_MAX_ED
        if (__ballot(reached)) {
            const int Hb = H >> 1, j = Hb + k;
            int cdist = lane - SP_BAND / 2; if (cdist < 0) cdist = -cdist;
            int key = reached ? ((Hb + j) * 8192 + (SP_BAND - cdist) * 64 + (SP_BAND - 1 - lane)) : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { int other = __shfl_xor(key, o); key = other > key ? other : key; }
            end_lane = __builtin_amdgcn_readfirstlane((SP_BAND - 1) - (key & 63));
            break;
        }
        if (s == max_ed) break;
        // a dead lane is hugely negative (+2 keeps it so), so validity is one sign test on the winner
        const int t = H + 2;                               // mismatch on the same diagonal; seen from the diagonal below it is
                                                           // the step "up from the upper neighbour"
        int best;
        if (track) {
            const int up = from_lower(H, NEG2), dn = from_upper(t, NEG2);
            const int oup = from_lower(O, 0), odn = from_upper(O, 0);
            int o = O;
            best = t;
            if (up > best) { best = up; o = oup; }
            if (dn > best) { best = dn; o = odn; }
            O = o;
        } else {
            best = max_from_lower(H, t);
            best = max_from_upper(t, best);
        }
        H = extend(best);
        ++s;
        if (TRACE) hist[s * SP_WAVE + lane] = (uint16_t)(H >= 0 ? H >> 1 : 0xFFFF);
    }
    {
        // When the cap ran out, every live diagonal stopped on a mismatch at A[H]; furthest-reaching points only grow with s, so no
        // base beyond max H was ever compared: a run on another A with the same first max H + 1 bases is this run, step for step.
        out.explored = end_lane < 0 ? (wave_max(H) >> 1) : -1;
    }
    if (SNAP) { *snap_out_s = keep_s; *snap_out_H = keep_H; }
    SP_STAT(3, s); SP_STAT(end_lane < 0 ? 16 + (s < 15 ? s : 15) : 32 + (s < 15 ? s : 15), 1);
    if (end_lane >= 0) { SP_STAT(2, 1); SP_STAT(4, s); }
    // one exit, every field assigned by value (an early return here leaves the struct in scratch memory)
    const bool done = end_lane >= 0;
    const int el = done ? end_lane : 0;
    const int he = __builtin_amdgcn_readlane(H, el) >> 1;
    const int oe = track ? __builtin_amdgcn_readlane(O, el) : el;
    const int ko = kb + oe, i0 = ko < 0 ? -ko : 0;
    out.ok = done ? 1 : 0; out.nm = done ? s : 0;
    out.a_end = done ? he : 0; out.b_end = done ? he + kb + el : 0;
    out.a_start = done ? i0 : 0; out.b_start = done ? i0 + ko : 0;      // b_start is only meaningful when the origin was tracked
    if (TRACE && done) {
        int l = end_lane;
        for (int t = s; t > 0; --t) {
            uint16_t raw = hist[(t - 1) * SP_WAVE + lane];
            int Hp = raw == 0xFFFF ? SP_NEG : (int)raw;
            int cc = __builtin_amdgcn_readlane(Hp, l);
            int uu = l > 0 ? __builtin_amdgcn_readlane(Hp, l - 1) : SP_NEG;
            int dd = l < SP_BAND - 1 ? __builtin_amdgcn_readlane(Hp, l + 1) : SP_NEG;
            int best = cc >= 0 ? cc + 1 : SP_NEG; int src = l; uint32_t type = SP_EV_X;
            if (uu >= 0 && uu > best) { best = uu; src = l - 1; type = SP_EV_D; }
            if (dd >= 0 && dd + 1 > best) { best = dd + 1; src = l + 1; type = SP_EV_I; }
            int kk = kb + l;
            int bpos = type == SP_EV_X ? best - 1 + kk : (type == SP_EV_D ? best + kk - 1 : best + kk);
            if (lane == 0) events[t - 1] = (type << 30) | (uint32_t)bpos;
            l = src;
        }
    }
}

// band-reachable windows of a cell (view coordinates); false when the band misses the rectangle
__device__ __forceinline__ bool cell_windows(int m, int n, int kb, int& i_min, int& i_max, int& j_min, int& j_max) {
    j_min = kb > 0 ? kb : 0;
    j_max = m + kb + SP_BAND - 1; if (j_max > n) j_max = n;
    i_min = -(kb + SP_BAND - 1); if (i_min < 0) i_min = 0;
    i_max = n - kb; if (i_max > m) i_max = m;
    return j_max > j_min && i_max > i_min;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The self-contained cell: stages both windows into this wave's private LDS slot, then runs the core.
template <bool TRACE, bool HASN>
__device__ __forceinline__ void wfa_cell(const CellIn& c, uint32_t* __restrict__ slot, int slot_words, int lane,
                                         uint16_t* __restrict__ hist, uint32_t* __restrict__ events, CellOut& out) {
    out.ok = 0; out.nm = 0; out.a_start = out.a_end = out.b_start = out.b_end = 0;
    const int m = c.a1 - c.a0, n = c.b1 - c.b0;
    if (m <= 0 || n <= 0 || c.max_ed < 0) return;
    const int kb = c.diag - SP_BAND / 2;
    int i_min, i_max, j_min, j_max;
    if (!cell_windows(m, n, kb, i_min, i_max, j_min, j_max)) return;
    const int alo = c.a0 + i_min, ahi = c.a0 + i_max, blo = c.b0 + j_min, bhi = c.b0 + j_max;
    const int wa = words_needed(alo, ahi), wb = words_needed(blo, bhi);
    if ((HASN ? 2 : 1) * (wa + wb) > slot_words) return;        // host sizes the slot; defensive
    uint32_t* LA = slot; uint32_t* LB = slot + wa;
    uint32_t* NA = HASN ? slot + wa + wb : nullptr; uint32_t* NB = HASN ? slot + 2 * wa + wb : nullptr;
    const int a_base = stage(LA, c.a_words, alo, ahi, lane);
    const int b_base = stage(LB, c.b_words, blo, bhi, lane);
    if (HASN) {
        // a set without N has no plane: treat as all zero
        if (c.a_nplane) stage(NA, c.a_nplane, alo, ahi, lane); else for (int w = lane; w < wa; w += SP_WAVE) NA[w] = 0;
        if (c.b_nplane) stage(NB, c.b_nplane, blo, bhi, lane); else for (int w = lane; w < wb; w += SP_WAVE) NB[w] = 0;
    }
    wave_lds_sync();
    wfa_core<TRACE, HASN, true>(LA, NA, c.a0 - a_base, m, LB, NB, c.b0 - b_base, n, kb, c.max_ed, lane, hist, events, out);
    wave_lds_sync();      // the slot is reused by the next cell
}

} // namespace spw
