"""sp_affine_rescore_batch (the two-piece affine re-score the library reports: sp_affine.hip) == oracle/affine.c, bit for bit: K1 pairs, K3 hits incl. the
other paralog on 256 diagonals, sequences with N, partial overlaps, bands that miss the rectangle, empty sequences."""
import numpy as np
import pytest

import oracle_ffi as of

pytestmark = pytest.mark.gpu


def check(oracle, gpu_ctx, targets, queries, pairs, band, a=1):
    T, Q = gpu_ctx.upload(targets), gpu_ctx.upload(queries)
    got = gpu_ctx.affine_rescore(Q, T, [(q, t, d) for (t, q, d) in pairs], a=a, band=band)          # set A = queries, set B = targets, diag = t_pos - q_pos
    for x, (t, q, d) in enumerate(pairs):
        exp = of.oracle_affine(oracle, targets[t], queries[q], -d, band, a)
        have = (int(got[x]["score"]), int(got[x]["nm"]), int(got[x]["b_start"]), int(got[x]["b_end"]), int(got[x]["a_start"]), int(got[x]["a_end"]))
        assert have == exp, (x, t, q, d, band, a, have, exp)
    return got


def test_k1_pairs(oracle, pkg, gpu_ctx):
    from pb_starphase_amd import synth
    fx = synth.HlaFixture()
    wl = synth.Config2Workload(fx, n_reads=300, seed=1000)
    rng = np.random.default_rng(5)
    reads = [wl.reads[r] for r in rng.choice(len(wl.reads), 60, replace=False)]
    alleles, pairs = [], []
    for q, read in enumerate(reads):
        g, a = wl.read_truth[wl.reads.index(read)]
        others = [x for x in range(len(fx.ids)) if fx.gene_of[x] == g and fx.dna[x]]
        for al in [a] + [int(x) for x in rng.choice(others, 2)]:
            alleles.append(fx.dna_fwd(al))
            d, _v = oracle.anchor(oracle.encode(alleles[-1]), oracle.encode(read))          # read_pos - allele_pos
            pairs.append((len(alleles) - 1, q, -d))
    got = check(oracle, gpu_ctx, alleles, reads, pairs, 64)
    assert (got["score"] > 1000).sum() >= 60
    check(oracle, gpu_ctx, alleles, reads, pairs[:40], 256)
    check(oracle, gpu_ctx, alleles, reads, pairs[:40], 64, a=5)


def test_k3_hits_both_bands(oracle, pkg, gpu_ctx):
    import cyp_cases_real as cr
    import cpu_port_cyp as cpc
    from pb_starphase_amd import synth
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    db, _ccfg = cpc.tables(cfg, gene_def, locus)
    sc = {n: (h, e) for n, h, e in cr.scenarios(locus)}
    reads = locus.sample(np.random.default_rng(7), sc["*4+*68/*1"][0], 60)[:24]
    templates = list(db.seqs)
    pairs = []
    for r, read in enumerate(reads):
        re = oracle.encode(read)
        for t in (0, 1, 7, 19, 36):                                            # CYP2D6, CYP2D7, hybrids, REP: top two placements of each on the read
            L = oracle.L
            import ctypes as C
            diags, votes = (C.c_int32 * 4)(), (C.c_int32 * 4)()
            te = oracle.encode(templates[t])
            n = L.osp_anchor_topk(te.ctypes.data_as(C.c_void_p), len(te), re.ctypes.data_as(C.c_void_p), len(re), 2, diags, votes)      # read_pos - template_pos
            for k in range(n):
                if votes[k] >= 4:
                    pairs.append((r, t, -int(diags[k])))                         # target = read, query = template: diag = t_pos - q_pos = read_pos - template_pos ... negated below
    pairs = [(r, t, -d) for (r, t, d) in pairs]
    assert len(pairs) > 100
    for band in (64, 256):
        check(oracle, gpu_ctx, reads, templates, pairs, band)


def test_random_pairs_with_n_and_edges(oracle, pkg, gpu_ctx):
    from pb_starphase_amd import synth
    rng = np.random.default_rng(11)
    targets, queries, pairs = [], [], []
    for it in range(120):
        L = int(rng.integers(260, 1500))
        base = "".join(rng.choice(list("ACGT"), L))
        q = synth.mutate(rng, base, int(rng.integers(0, 8)), int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        if rng.random() < 0.3:
            q = q[:len(q) // 3] + "N" * int(rng.integers(1, 4)) + q[len(q) // 3:]
        if rng.random() < 0.3:
            base = base[:L // 2] + "N" + base[L // 2 + 1:]
        a, b = int(rng.integers(0, L // 4 + 1)), int(rng.integers(0, L // 4 + 1))
        t = "".join(rng.choice(list("ACGT"), a)) + base                        # the target starts with a stretch the query does not have
        q = q[b:]
        targets.append(t); queries.append(q)
        pairs.append((it, it, (a + b) + int(rng.integers(-6, 7))))               # t_pos - q_pos, a little off
    targets += ["", "ACGT", "ACGTACGTACGTACGTAGCTAGCTAGCTAGCATCGATCGACTAGCTACG"]
    queries += ["ACGTACGT", "", "ACGTACGTACGTACGTAGCTAGCTAGCTAGCATCGATCGACTAGCTACG"]
    pairs += [(120, 120, 0), (121, 121, 0), (122, 122, 0), (122, 122, 900), (122, 122, -900), (0, 122, 3), (122, 0, -2)]
    for band in (64, 256):
        check(oracle, gpu_ctx, targets, queries, pairs, band)
        check(oracle, gpu_ctx, targets, queries, pairs, band, a=5)


def test_rows_around_the_clusters_equal_all_rows(pkg, gpu_ctx):
    """the re-score of the library's own mappings runs the DP over the rows around the edits that do not stand alone (context option mm2_rescore 1, the default);
    2 runs it over all rows of such a mapping: the same numbers for every K1 winner of 3,000 of configs[1]'s reads and every K3 hit of 600 of a configs[2] sample"""
    from pb_starphase_amd import synth
    import cyp_cases_real as cr
    fx = synth.HlaFixture(); db = fx.make_db(pkg, gpu_ctx)
    wl = synth.Config2Workload(fx, n_reads=3000, seed=1000)
    R = gpu_ctx.upload(wl.reads)
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    cdb = pkg.ffi.CypDb(gpu_ctx, cfg, gene_def, locus.sequence, locus.start)
    tm = cdb.templates(); tset = gpu_ctx.upload([t[3] for t in tm]); ttype = np.array([t[0] for t in tm], np.int32)
    sc = {n: (h, e) for n, h, e in cr.scenarios(locus)}
    Rc = gpu_ctx.upload(locus.sample(np.random.default_rng(11), sc["*10+*36/*10"][0], 600))
    got = {}
    try:
        for mode in (1, 2):
            gpu_ctx.set_option("mm2_rescore", mode)
            gpu_ctx.profile_reset()
            got[mode] = (db.realign_reads(R), gpu_ctx.cyp_find_regions(tset, ttype, Rc, 0.5))
            got[mode] += (gpu_ctx.profile_get("k1_af_dp")[0], gpu_ctx.profile_get("k3_af_dp")[0])
    finally:
        gpu_ctx.set_option("mm2_rescore", 1)
    for f in ("mm2_score", "mm2_nm", "mm2_t_start", "mm2_t_end", "mm2_q_start", "mm2_q_end"):
        assert (got[1][0][f] == got[2][0][f]).all(), f
    assert (got[1][0]["mm2_score"] > 0).sum() > 2500
    assert len(got[1][1]) == len(got[2][1]) > 600
    for f in ("mm2_score", "mm2_nm", "mm2_start", "mm2_end", "mm2_q_start", "mm2_q_end"):
        assert (got[1][1][f] == got[2][1][f]).all(), f
    print("DP ms, rows around the clusters / all rows: K1", round(got[1][2], 3), "/", round(got[2][2], 3), " K3", round(got[1][3], 3), "/", round(got[2][3], 3))
