"""K1 on a database of the shape of the reference's current one (v2.0.0: 41,374 alleles over 11 genes, 23,124 with DNA, class II alleles up to 15.5 kb;
pb_starphase_amd.synth.SyntheticHlaFixture): time of the cells launch by read count, for the counters of profiles/scripts/k1_scale.sh.
usage: k1_scale.py [reads per haplotype and gene]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import synth
per_hap = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = pkg.Context(0)
fx = synth.SyntheticHlaFixture(scale=1.0, seed=5)
db = fx.make_db(pkg, ctx)
rng = np.random.default_rng(31)
reads = []
for g in range(len(fx.genes)):
    full = fx.full_length_alleles(g)
    for a in sorted(rng.choice(full, 2, replace=False).tolist()):
        hap, s = fx.haplotype(g, a)
        reads += synth.simulate_reads(rng, hap, s, len(fx.dna[a]), per_hap, mean_len=16000, sd_len=2500, min_overlap=min(9000, len(fx.dna[a]) - 200))
R = ctx.upload([reads[i] for i in rng.permutation(len(reads))])
n_dna = sum(1 for d in fx.dna if d)
bases = sum(len(d) for d in fx.dna if d)
print(f"database: {len(fx.ids)} alleles, {n_dna} with DNA, {bases / 1e6:.1f} Mbases of DNA ({bases / 4 / 1e6:.1f} MB packed), longest {max(len(d) for d in fx.dna)}; reads {len(reads)}, mean {np.mean([len(r) for r in reads]):.0f} bases")
out = db.realign_reads(R)
ctx.profile_reset(); ctx.synchronize()
t0 = time.perf_counter(); out = db.realign_reads(R); dt = time.perf_counter() - t0
ok = (out["status"] == 0).mean()
print(f"K1 wall {1e3 * dt:.2f} ms, reads accepted {100 * ok:.1f} %")
for k in ("anchor_k1", "k1_cells", "k1_cells_deep", "k1_finalize"):
    ms, n, cells = ctx.profile_get(k)
    print(f"  {k:16s} {ms:8.3f} ms  launches {n}  cells {cells}")
