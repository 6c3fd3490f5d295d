"""GPU parity: sp_anchor_batch / sp_align_batch (HIP, through the C ABI) vs the CPU oracle, bit exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rand_seq(rng, n):
    return "".join(rng.choice(list("ACGT"), n))


def make_cases(rng, pkg):
    from pb_starphase_amd import synth
    A, B = [], []
    # contained: B = flank + mutated(A) + flank
    for ln in (40, 100, 218, 930, 1100, 3200, 6100):
        for (ns, ni, nd) in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (3, 2, 2), (12, 5, 6)):
            if (ns + ni + nd) * 12 + 60 > ln:
                continue
            a = rand_seq(rng, ln)
            core = synth.mutate(rng, a, ns, ni, nd)
            fl, fr = int(rng.integers(0, 300)), int(rng.integers(0, 300))
            A.append(a)
            B.append(rand_seq(rng, fl) + core + rand_seq(rng, fr))
    # dovetails: A overhangs B at the start / at the end, B overhangs, equal
    for ln in (300, 1500, 4000):
        g = rand_seq(rng, ln + 600)
        a = g[100:100 + ln]
        A += [a, a, a, a]
        B += [g[300:], g[:ln - 50], g[100:100 + ln], synth.mutate(rng, g[150:ln + 400], 2, 1, 1)]
    # N handling and tiny sequences
    a = rand_seq(rng, 400)
    A += [a, a[:200] + "N" + a[201:], "ACGT", "ACGTACGTACGTACGTACGT"]
    B += [a[:100] + "N" + a[101:], a, "ACGT", "TTACGTACGTACGTACGTACGTAA"]
    # repeats: the 218-mer of the reference's test_weight_sequence shape (two near-identical halves)
    half = rand_seq(rng, 109)
    A.append(half + half[:84] + "G" + half[85:])
    B.append(half + half[:84] + "C" + half[85:])
    return A, B


def compare(oracle, ctx, A, B, a_idx, b_idx, max_ed=255, shift=0):
    sa, sb = ctx.upload(A), ctx.upload(B)
    # anchors: B side indexed = "A" of sp_anchor_batch is the indexed set; index the B list, look up A
    diag, votes = ctx.anchor_batch(sb, sa, b_idx, a_idx)      # diag = a_pos - b_pos
    exp = [oracle.anchor(B[j], A[i]) for i, j in zip(a_idx, b_idx)]
    assert diag.tolist() == [d if v > 0 else 0 for d, v in exp]
    assert votes.tolist() == [v for _, v in exp]
    cell_diag = (-diag + shift).astype(np.int32)
    out, ev = ctx.align_batch(sa, sb, a_idx, b_idx, cell_diag, np.full(len(a_idx), max_ed, np.int32), events=True)
    out2 = ctx.align_batch(sa, sb, a_idx, b_idx, cell_diag, np.full(len(a_idx), max_ed, np.int32), events=False)
    n_ok = 0
    for x, (i, j) in enumerate(zip(a_idx, b_idx)):
        al, oev = oracle.wfa(A[i], B[j], int(cell_diag[x]), max_ed, retry=2)          # sp_align_batch: lost cells and cells that needed > 32 edits run again on 256 diagonals
        got = out[x]
        want = (al.ok, al.nm, al.a_start, al.a_end, al.b_start, al.b_end, al.a_len, al.b_len) if al.ok else None
        if al.ok:
            assert tuple(int(v) for v in got.tolist()) == want, (x, i, j, got, want)
            assert tuple(int(v) for v in out2[x].tolist()) == want, ("untraced", x, got, want)
            assert ev[x, :al.nm].tolist() == oev.tolist(), (x, ev[x, :al.nm], oev)
            n_ok += 1
        else:
            assert got["ok"] == 0 and out2[x]["ok"] == 0, (x, got)
    return n_ok


def test_align_synthetic(oracle, pkg, gpu_ctx):
    rng = np.random.default_rng(11)
    A, B = make_cases(rng, pkg)
    idx = np.arange(len(A), dtype=np.uint32)
    n_ok = compare(oracle, gpu_ctx, A, B, idx, idx)
    assert n_ok >= len(A) - 4
    # off-centre anchors must give the same contract on both sides
    for shift in (-20, 7, 31):
        compare(oracle, gpu_ctx, A, B, idx, idx, shift=shift)
    # tight edit caps: failures must agree too
    compare(oracle, gpu_ctx, A, B, idx, idx, max_ed=3)
    compare(oracle, gpu_ctx, A, B, idx, idx, max_ed=0)


def test_align_real_alleles(oracle, pkg, gpu_ctx):
    """real IMGT/HLA alleles against the GRCh38 gene regions (nm 80..160, band-limited failures included)"""
    from pb_starphase_amd import synth
    fx = synth.HlaFixture(max_alleles_per_gene=120, seed=5)
    A = [fx.dna_fwd(a) for a in range(len(fx.ids)) if fx.dna[a]]
    gene = [int(fx.gene_of[a]) for a in range(len(fx.ids)) if fx.dna[a]]
    a_idx = np.arange(len(A), dtype=np.uint32)
    b_idx = np.array(gene, np.uint32)
    n_ok = compare(oracle, gpu_ctx, A, fx.gene_ref, a_idx, b_idx)
    assert n_ok > 0.9 * len(A)
    # alleles against each other (cDNA, short)
    C = [fx.cdna[a] for a in range(0, len(fx.ids), 3)][:80]
    ai = np.repeat(np.arange(len(C)), 4).astype(np.uint32)
    bi = ((ai * 7 + np.tile(np.arange(4), len(C))) % len(C)).astype(np.uint32)
    compare(oracle, gpu_ctx, C, C, ai, bi)


def _edit(rng, seq, n_sub, n_ins, n_del):
    """edits anywhere, adjacent ones included (synth.mutate keeps them apart)"""
    s = list(seq)
    for _ in range(n_sub):
        if s:
            p = int(rng.integers(0, len(s))); s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + int(rng.integers(3))) % 4]
    for _ in range(n_ins):
        p = int(rng.integers(0, len(s) + 1))
        s[p:p] = list(rand_seq(rng, int(rng.integers(1, 4))))
    for _ in range(n_del):
        if len(s) > 8:
            p = int(rng.integers(0, len(s) - 3)); del s[p:p + int(rng.integers(1, 4))]
    return "".join(s)


def test_align_fuzz(oracle, pkg, gpu_ctx):
    """seeded random pairs of every shape the core has a special path for: long clean stretches (cooperative scans and the kept
    scan), substitution runs on one diagonal, indel-heavy paths, diagonals that leave the rectangle, tight and exhausted caps"""
    import os
    for seed in [int(x) for x in os.environ.get("SP_FUZZ_SEEDS", "20261003").split(",")]:       # (a list of seeds for a longer hunt)
        _fuzz_once(oracle, gpu_ctx, seed)


def _fuzz_once(oracle, gpu_ctx, seed):
    rng = np.random.default_rng(seed)
    A, B = [], []
    for _ in range(260):
        ln = int(rng.choice([24, 60, 180, 700, 1500, 3300, 5200, 8800]))
        a = rand_seq(rng, ln)
        kind = int(rng.integers(0, 5))
        if kind == 0:      # substitutions only: the alignment stays on one diagonal
            core = _edit(rng, a, int(rng.integers(0, 31)), 0, 0)
        elif kind == 1:    # indel heavy
            core = _edit(rng, a, int(rng.integers(0, 5)), int(rng.integers(0, 13)), int(rng.integers(0, 13)))
        elif kind == 2:    # a long repeat inside: many diagonals match for a while
            unit = rand_seq(rng, int(rng.integers(2, 40)))
            core = a[:ln // 3] + unit * int(rng.integers(3, 30)) + a[ln // 3:]
            a = a[:ln // 3] + unit * int(rng.integers(3, 30)) + a[ln // 3:]
        elif kind == 3:    # partial overlap
            cut = int(rng.integers(0, max(1, ln // 2)))
            core = _edit(rng, a[cut:], int(rng.integers(0, 7)), int(rng.integers(0, 4)), int(rng.integers(0, 4)))
        else:              # unrelated tail
            core = _edit(rng, a[:ln * 2 // 3], 2, 1, 1) + rand_seq(rng, ln // 3)
        fl, fr = int(rng.integers(0, 200)), int(rng.integers(0, 200))
        A.append(a)
        B.append(rand_seq(rng, fl) + core + rand_seq(rng, fr))
    idx = np.arange(len(A), dtype=np.uint32)
    n_ok = compare(oracle, gpu_ctx, A, B, idx, idx)
    assert n_ok > len(A) // 2
    for shift, cap in ((-31, 255), (13, 40), (0, 6), (29, 1)):
        compare(oracle, gpu_ctx, A, B, idx, idx, max_ed=cap, shift=shift)


def test_long_indels_use_the_wide_band(oracle, pkg, gpu_ctx):
    """an insertion / deletion of 40-100 bases inside an otherwise near-identical pair: on 64 diagonals the cell is lost (or, for the shorter
    ones that the midpoint anchor still covers, found); sp_align_batch runs a lost cell again on 256 diagonals.  Bit for bit against
    oracle/align.c (osp_wfa_retry), and the wide run recovers the pair with about `indel` edits."""
    from pb_starphase_amd import synth
    rng = np.random.default_rng(2026)
    A, B, sizes = [], [], []
    for ln in (1500, 3200, 6100):
        for size in (40, 55, 70, 85, 100):
            for kind in ("del", "ins"):
                a = rand_seq(rng, ln)
                core = synth.mutate(rng, a, 3, 1, 1)
                at = int(rng.integers(ln // 3, 2 * ln // 3))
                core = core[:at] + core[at + size:] if kind == "del" else core[:at] + rand_seq(rng, size) + core[at:]
                A.append(a); B.append(rand_seq(rng, 120) + core + rand_seq(rng, 80)); sizes.append(size)
    idx = np.arange(len(A), dtype=np.uint32)
    n_ok = compare(oracle, gpu_ctx, A, B, idx, idx, max_ed=300)
    assert n_ok == len(A)
    sa, sb = gpu_ctx.upload(A), gpu_ctx.upload(B)
    diag, _votes = gpu_ctx.anchor_batch(sb, sa, idx, idx)
    out = gpu_ctx.align_batch(sa, sb, idx, idx, (-diag).astype(np.int32), np.full(len(A), 300, np.int32))
    narrow_lost = 0
    for x, size in enumerate(sizes):
        assert out[x]["ok"] and size <= out[x]["nm"] <= size + 12, (x, size, out[x])
        narrow_lost += not oracle.wfa(A[x], B[x], int(-diag[x]), 300, events=False)[0].ok
    assert narrow_lost >= len(A) // 3                                   # the retry did the work
