"""GPU twin of the reference's real-data aligner pin `test_hlaconfig_new` (/root/reference/src/hla/alleles.rs:512-547; the CPU statement: tests/test_oracle_mm2.py):
the two real IMGT alleles of test_data/HLA-faux against the RefSeq records +- 2,000 bases of the chr6 islands, through the library's own mapping contract -- anchor
(sp_anchor_batch), then the two-piece affine re-score that reports minimap2's numbers (sp_affine_rescore_batch: end-clipped local alignment, NM) -- must extend the coordinates
to HlaConfig::default()'s, with the numbers the minimap2 restatement gives (nm 42 forward for A*01:01:01:01, nm 0 on the reverse strand for B*07:02:01:01)."""
import pytest

import test_oracle_mm2 as tm

pytestmark = pytest.mark.gpu


def test_hlaconfig_new_on_the_gpu(pkg, gpu_ctx):
    def map_best(target, query):
        out = []
        queries = [query, tm.revcomp(query)]
        Q, T = gpu_ctx.upload(queries), gpu_ctx.upload([target])
        diag, votes = gpu_ctx.anchor_batch(Q, T, [0, 1], [0, 0])              # diag = target position - query position
        for rev in (0, 1):
            if votes[rev] <= 0:
                continue
            r = gpu_ctx.affine_rescore(Q, T, [(rev, 0, int(diag[rev]))], a=1, band=256)[0]
            if r["score"] <= 0:
                continue
            qs, qe = int(r["a_start"]), int(r["a_end"])
            if rev:                                                           # minimap2 reports query coordinates on the query's own strand
                qs, qe = len(query) - qe, len(query) - qs
            out.append(dict(rev=rev, nm=int(r["nm"]), q_start=qs, q_end=qe, t_start=int(r["b_start"]), t_end=int(r["b_end"]), score=int(r["score"])))
        return sorted(out, key=lambda m: -m["score"])                        # minimap2's output order: best DP score first
    got = tm.hlaconfig_extend(map_best, tm._islands(), tm._faux_alleles())
    for name, _s, _e, want in tm.HLACONFIG_CASES:
        assert got[name][:2] == want, (name, got[name])
    a, b = got["A*01:01:01:01"][2], got["B*07:02:01:01"][2]
    assert (a["rev"], a["nm"]) == (0, 42) and (b["rev"], b["nm"]) == (1, 0)
    # the same mappings the minimap2 restatement reports (oracle/mm2.c), field by field
    mm = tm.mm2_ffi.Mm2()
    ref = tm.hlaconfig_extend(lambda t, q: mm.map_pair(t, q), tm._islands(), tm._faux_alleles())
    for name in ("A*01:01:01:01", "B*07:02:01:01"):
        for k in ("rev", "nm", "q_start", "q_end", "t_start", "t_end"):
            assert got[name][2][k] == ref[name][2][k], (name, k, got[name][2], ref[name][2])
