"""Differential hunt on whole calls (run on the GPU box): random CYP2D6 samples (sp_cyp_diplotype) and random HLA samples (sp_hla_realign_reads
+ sp_hla_diplotype_gene) against the same pipelines assembled from the oracle's pieces (tests/cyp_pipeline.py, tests/hla_pipeline.py).
usage: pipeline_fuzz.py <n_cyp> <n_hla> [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
import oracle_ffi as of
import cyp_cases_real as cr
import cyp_pipeline as cp
import hla_expected as hx
import hla_pipeline as hp

n_cyp, n_hla = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
oracle = of.load()
ctx = pkg.Context(0)
rng = np.random.default_rng(seed)
bad = 0

# ---------------------------------------------------------------- CYP2D6
if n_cyp:
    cfg, gene_def = cr.load_db()
    locus = synth.Chr22Locus(cfg, gene_def, seed=3)
    db = pkg.ffi.CypDb(ctx, cfg, gene_def, locus.sequence, locus.start)
    tm, vs = db.templates(), db.variants()
    names, rows = db.alleles()
    bb = cfg["cyp_coordinates"]["CYP2D6_wfa_backbone"]
    odb = cp.Db([t[2] for t in tm], [t[0] for t in tm], [t[1] for t in tm], [t[3] for t in tm], [t[4] for t in tm],
                locus.slice(bb["start"], bb["end"]), [(p - bb["start"], r, a) for p, r, a, _l, _v in vs], [v[4] for v in vs], names, rows,
                var_labels=[v[3] for v in vs])
    stars = sorted({d["star_allele"] for d in gene_def.values()})
    hybrids = [t[2] for t in tm if "::" in (t[2] or "")] or []
    hybrid_names = [n for n in ("CYP2D6::CYP2D7::exon2", "CYP2D6::CYP2D7::exon9", "CYP2D7::CYP2D6::exon9", "CYP2D7::CYP2D6::intron1")]

    def gene_body():
        if rng.random() < 0.15:
            return locus.hybrid(str(rng.choice(hybrid_names))), "hybrid"
        for _ in range(100):
            s = str(rng.choice(stars))
            try:
                return locus.star_allele(s), s
            except AssertionError:                            # (a few database alleles list overlapping variants the simple generator cannot apply)
                continue
        raise RuntimeError("no star allele could be realised")

    def haplotype():
        u = rng.random()
        if u < 0.08:
            return locus.haplotype(None), ["*5"]
        k = 1 if u < 0.7 else 2
        bodies = [gene_body() for _ in range(k)]
        return locus.haplotype([b for b, _ in bodies]), [n for _, n in bodies]

    for it in range(n_cyp):
        (h1, n1), (h2, n2) = haplotype(), haplotype()
        n_reads = int(rng.choice([60, 120, 160, 300]))
        reads = locus.sample(rng, [h1, h2], n_reads)
        t0 = time.time()
        try:
            exp = cp.diplotype(oracle, odb, reads, cfg=db.cfg)
        except Exception as e:                                   # the Python statement itself gave up: report, do not count
            print(f"cyp {it}: oracle pipeline raised {e!r} ({n1} / {n2}, {n_reads} reads)"); continue
        call, cons, labels = db.diplotype(ctx.upload(reads))
        diff = []
        if call.status != exp["status"]:
            diff.append(("status", call.status, exp["status"]))
        elif call.status == 0:
            got = dict(cons=cons, labels=labels, chain1=list(call.chain1[:call.n1]), chain2=list(call.chain2[:call.n2]), score=call.score,
                       hap1=call.hap1.decode(), hap2=call.hap2.decode(), core1=call.core1.decode(), core2=call.core2.decode(),
                       deep1=call.deep1.decode(), deep2=call.deep2.decode())
            want = dict(cons=exp["consensus"], labels=[(int(t), s) for t, s in exp["labels"]], chain1=exp["chain1"], chain2=exp["chain2"], score=exp["score"],
                        hap1=exp["hap1"], hap2=exp["hap2"], core1=exp["core1"], core2=exp["core2"], deep1=exp["deep1"], deep2=exp["deep2"])
            diff = [(k, got[k] if k != "cons" else [len(c) for c in got[k]], want[k] if k != "cons" else [len(c) for c in want[k]]) for k in got if got[k] != want[k]]
        print(f"cyp {it}: {'+'.join(n1)} / {'+'.join(n2)}, {len(reads)} reads -> status {call.status} {call.hap1.decode()} / {call.hap2.decode()}  "
              f"[{time.time() - t0:.1f} s]{'  DIFFERS' if diff else ''}", flush=True)
        for d in diff:
            print("     ", d)
        bad += bool(diff)

# ---------------------------------------------------------------- HLA
if n_hla:
    fx = synth.HlaFixture(max_alleles_per_gene=150, seed=4)
    hdb = fx.make_db(pkg, ctx)
    for it in range(n_hla):
        reads, truth = [], {}
        per_hap = int(rng.choice([4, 8, 14, 25]))
        p = float(rng.choice([0.0004, 0.001, 0.003]))
        for g in range(len(fx.genes)):
            fl = fx.full_length_alleles(g)
            pick = rng.choice(fl, 2, replace=False).tolist()
            if rng.random() < 0.3:
                pick = [pick[0], pick[0]]
            truth[g] = pick
            for a in pick:
                hap, s = fx.haplotype(g, a)
                clean = synth.simulate_reads(rng, hap, s, len(fx.dna[a]), per_hap + int(rng.integers(0, 4)), errors=False)
                reads += [synth.hifi_errors(rng, r, p_sub=p / 2, p_ins=p, p_del=p) for r in clean]
        order = rng.permutation(len(reads))
        reads = [reads[i] for i in order]
        R = ctx.upload(reads)
        k1_gpu = hdb.realign_reads(R)
        k1_exp, _ = hx.k1_expected_seeded(oracle, fx, reads)       # (the library's default since round 5: K1 in the reference's call pattern)
        diff = []
        for g in range(len(fx.genes)):
            call, c1, c2, is1 = hdb.diplotype_gene(g, R, k1_gpu)
            exp = hp.diplotype_gene(oracle, fx, g, reads, k1_exp, synth)
            got = dict(status=call.status, n_reads=call.n_reads, cons1=c1, cons2=c2, is_cons1=is1.tolist(), is_dual=call.is_dual, dual_passed=call.dual_passed,
                       used_dna_dual=call.used_dna_dual, counts1=call.counts1, counts2=call.counts2, typed1=call.typed1, typed2=call.typed2,
                       allele1=call.allele1, allele2=call.allele2) if call.status == 0 else dict(status=call.status)
            want = {k: ([bool(x) for x in exp[k]] if k == "is_cons1" else exp[k]) for k in got}
            diff += [(fx.genes[g], k, got[k] if "cons" not in k else "...", want[k] if "cons" not in k else "...") for k in got if got[k] != want[k]]
        print(f"hla {it}: {per_hap} reads per haplotype, error rate {p}, truth {truth} -> {'DIFFERS' if diff else 'equal'}", flush=True)
        for d in diff:
            print("     ", d)
        bad += bool(diff)

print("samples that differ:", bad)
sys.exit(1 if bad else 0)
