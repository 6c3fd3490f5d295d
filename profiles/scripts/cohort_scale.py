"""Times a WGS-style cohort on one GPU: S samples x 2 genes x ~45 reads each, one read set, one realignment call, one cohort call.
Run on the GPU box:  python profiles/scripts/cohort_scale.py [n_samples]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
from pb_starphase_amd import synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fx = synth.HlaFixture()
ctx = pkg.Context(0)
db = fx.make_db(pkg, ctx)
rng = np.random.default_rng(5)
reads, sample_of, truth = [], [], []
for s in range(S):
    t = {}
    for g in range(len(fx.genes)):
        pick = rng.choice(fx.full_length_alleles(g), 2, replace=False).tolist()
        t[g] = sorted(pick)
        for a in pick:
            hap, st = fx.haplotype(g, a)
            rs = synth.simulate_reads(rng, hap, st, len(fx.dna[a]), 22, mean_len=7000, sd_len=1500, min_overlap=2500)
            reads += rs; sample_of += [s] * len(rs)
    truth.append(t)
R = ctx.upload(reads)
genes = list(range(len(fx.genes)))
same = lambda a, b: a == b or (a >= 0 and b >= 0 and fx.cdna[a] == fx.cdna[b] and fx.dna[a] == fx.dna[b])
for rep in range(3):
    t0 = time.perf_counter(); k1 = db.realign_reads(R); t1 = time.perf_counter()
    cohort, _ = db.diplotype_cohort(S, sample_of, genes, R, k1); t2 = time.perf_counter()
ok = sum(all(same(x, y) for x, y in zip(sorted([cohort[s][g][0].allele1, cohort[s][g][0].allele2]), truth[s][g])) for s in range(S) for g in genes)
print(f"{S} samples, {len(reads)} reads: realign {1e3*(t1-t0):.1f} ms, cohort solve {1e3*(t2-t1):.1f} ms -> {S/(t2-t0):.1f} samples/s, {len(reads)/(t2-t0):.0f} reads/s; calls equal truth: {ok}/{S*len(genes)}")
# one sample at a time, for comparison
mine = [i for i, x in enumerate(sample_of) if x == 0]
R0 = ctx.upload([reads[i] for i in mine])
for rep in range(2):
    t0 = time.perf_counter(); k0 = db.realign_reads(R0); one, _ = db.diplotype_genes(genes, R0, k0); t1 = time.perf_counter()
print(f"one sample alone ({len(mine)} reads): {1e3*(t1-t0):.1f} ms -> {1/(t1-t0):.1f} samples/s")
