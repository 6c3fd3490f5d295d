"""Synthetic workload generator (host side, numpy only) for tests and bench.py.

Inputs are the committed data fixtures under tests/golden/ (IMGT/HLA alleles of the bundled v0.14.1 database and
the two GRCh38 chr6 islands the reference's own tests ship); see tests/golden/make_fixtures.py.
Nothing here is on the product path: it only manufactures reads / consensuses of the shape BASELINE.json names
(config 2: HLA-A/-B, 10k HiFi reads).
"""
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_COMP = str.maketrans("ACGTN", "TGCAN")


def revcomp(s):
    return s.translate(_COMP)[::-1]


class HlaFixture:
    """The HLA part of the database in the shape sp_hla_db_create wants (alleles in database-key order)."""

    def __init__(self, genes=None, max_alleles_per_gene=None, seed=0):
        db = json.load(gzip.open(os.path.join(GOLDEN, "hla_db_v0.14.1.json.gz")))
        isl = json.load(open(os.path.join(GOLDEN, "chr6_hla_islands.json")))["islands"]
        cfg = db["hla_config"]
        self.genes = genes or sorted(cfg["hla_coordinates"].keys())
        self.buffer = 100
        self.gene_fwd, self.gene_ref, self.exons, self.coords, self.island = [], [], [], [], []
        for g in self.genes:
            co = cfg["hla_coordinates"][g]
            I = next(i for i in isl if i["start"] <= co["start"] and co["end"] <= i["end"])
            lo, hi = co["start"] - self.buffer, co["end"] + self.buffer
            self.gene_ref.append(I["sequence"][lo - I["start"]:hi - I["start"]])
            self.gene_fwd.append(1 if cfg["hla_is_forward_strand"][g] else 0)
            self.exons.append([(e["start"] - lo, e["end"] - lo) for e in cfg["hla_exons"][g]])
            self.coords.append((co["start"], co["end"]))
            self.island.append(I)
        rng = np.random.default_rng(seed)
        ids = sorted(k for k, v in db["hla_sequences"].items() if v["gene_name"] in self.genes)
        if max_alleles_per_gene is not None:
            keep = []
            for g in self.genes:
                gi = [k for k in ids if db["hla_sequences"][k]["gene_name"] == g]
                if len(gi) > max_alleles_per_gene:
                    gi = sorted(rng.choice(gi, max_alleles_per_gene, replace=False).tolist())
                keep += gi
            ids = sorted(keep)
        self.ids = ids
        self.gene_of = np.array([self.genes.index(db["hla_sequences"][k]["gene_name"]) for k in ids], np.uint32)
        self.dna = [db["hla_sequences"][k]["dna_sequence"] or "" for k in ids]
        self.cdna = [db["hla_sequences"][k]["cdna_sequence"] for k in ids]
        self.star = [":".join(db["hla_sequences"][k]["star_allele"]) for k in ids]

    def dna_fwd(self, a):
        """allele DNA in hg38 orientation (create_hla_fasta, src/hla/realigner.rs:497-526)"""
        s = self.dna[a]
        return s if self.gene_fwd[self.gene_of[a]] else revcomp(s)

    def make_db(self, pkg, ctx):
        return pkg.HlaDb(ctx, self.gene_of, self.dna, self.cdna, self.gene_ref, self.gene_fwd, self.exons, self.buffer)

    def full_length_alleles(self, g, frac=0.95):
        """alleles whose DNA covers (nearly) the longest span known for the gene"""
        lens = [len(self.dna[a]) for a in range(len(self.ids)) if self.gene_of[a] == g]
        cut = frac * max(lens)
        return [a for a in range(len(self.ids)) if self.gene_of[a] == g and len(self.dna[a]) >= cut]

    def _island_index(self, g):
        if not hasattr(self, "_idx"):
            self._idx = {}
        if g not in self._idx:
            seq = self.island[g]["sequence"]
            d = {}
            for p in range(len(seq) - 15):
                d.setdefault(seq[p:p + 16], []).append(p)
            self._idx[g] = d
        return self._idx[g]

    def _vote(self, g, piece):
        idx, votes = self._island_index(g), {}
        for j in range(len(piece) - 15):
            for p in idx.get(piece[j:j + 16], ()):
                votes[p - j] = votes.get(p - j, 0) + 1
        if not votes:
            raise ValueError("allele does not anchor in the island")
        return max(sorted(votes), key=lambda k: votes[k])

    def haplotype(self, g, a):
        """island with the stretch covered by allele a replaced by the allele (hg38 orientation).
        Returns (sequence, allele_start_in_haplotype)."""
        seq = self.island[g]["sequence"]
        al = self.dna_fwd(a)
        s = self._vote(g, al[:400])                      # island position of allele base 0
        tail = al[-400:]
        e = self._vote(g, tail) + len(tail)              # island position just past the allele
        s = max(0, s)
        e = min(len(seq), max(e, s))
        return seq[:s] + al + seq[e:], s


def hifi_errors(rng, seq, p_sub=0.0002, p_ins=0.0004, p_del=0.0004):
    """HiFi-like error model: rare substitutions, indels biased to homopolymer runs (Q30-ish overall)."""
    n = len(seq)
    arr = np.frombuffer(seq.encode(), np.uint8)
    u = rng.random(n)
    hp = np.zeros(n, bool)
    hp[1:] = arr[1:] == arr[:-1]
    w = np.where(hp, 4.0, 0.5)                         # homopolymer positions 8x more indel-prone
    sub = u < p_sub
    ins = (u >= p_sub) & (u < p_sub + p_ins * w)
    dele = (u >= p_sub + p_ins * w) & (u < p_sub + (p_ins + p_del) * w)
    if not (sub.any() or ins.any() or dele.any()):
        return seq
    out = []
    last = 0
    bases = "ACGT"
    for i in np.flatnonzero(sub | ins | dele):
        out.append(seq[last:i])
        c = seq[i]
        if sub[i]:
            out.append(bases[(bases.index(c) + 1 + int(rng.integers(3))) % 4] if c in bases else c)
        elif ins[i]:
            out.append(c + c)
        last = i + 1
    out.append(seq[last:])
    return "".join(out)


def simulate_reads(rng, hap, gene_start, gene_len, n, mean_len=15000, sd_len=3000, min_overlap=2200, errors=True):
    """n reads drawn from one haplotype string; every read overlaps the gene body by >= min_overlap bases."""
    reads = []
    L = len(hap)
    while len(reads) < n:
        ln = int(min(L, max(3000, rng.normal(mean_len, sd_len))))
        s = int(rng.integers(0, L - ln + 1))
        e = s + ln
        ov = min(e, gene_start + gene_len) - max(s, gene_start)
        if ov < min_overlap:
            continue
        r = hap[s:e]
        reads.append(hifi_errors(rng, r) if errors else r)
    return reads


def mutate(rng, seq, n_sub=0, n_ins=0, n_del=0):
    """exact numbers of isolated edits at distinct, well separated positions (for parity tests)"""
    pos = sorted(rng.choice(np.arange(20, len(seq) - 20, 12), n_sub + n_ins + n_del, replace=False).tolist(), reverse=True)
    kinds = ["s"] * n_sub + ["i"] * n_ins + ["d"] * n_del
    rng.shuffle(kinds)
    s = seq
    for p, k in zip(pos, kinds):
        c = s[p]
        other = "ACGT"[("ACGT".index(c) + 1 + int(rng.integers(3))) % 4] if c in "ACGT" else "A"
        if k == "s":
            s = s[:p] + other + s[p + 1:]
        elif k == "i":
            s = s[:p] + other + s[p:]
        else:
            s = s[:p] + s[p + 1:]
    return s


class Config2Workload:
    """BASELINE.json configs[1]: HLA-A/-B diplotyping on n_reads synthetic HiFi reads vs the bundled IMGT/HLA DB.

    Truth: two distinct full-length DNA alleles per gene; reads split evenly over genes and haplotypes; read length
    ~N(15 kb, 3 kb) clipped to the GRCh38 island the reference's tests ship (7.6 / 8.1 kb), HiFi error model.
    The K2 inputs are synthetic consensuses: the truth haplotype around the allele (gene strand) and its cDNA
    (the DWFA consensus step itself is a 'next' row)."""

    def __init__(self, fx, n_reads=10000, seed=1):
        rng = np.random.default_rng(seed)
        self.fx = fx
        self.truth = []            # (gene, allele index) per haplotype
        self.reads = []
        self.read_truth = []
        self.consensus = []        # (gene, cons_dna_gene_strand, cons_cdna, truth allele)
        G = len(fx.genes)
        per_gene = n_reads // G
        for g in range(G):
            full = fx.full_length_alleles(g)
            pair = rng.choice(full, 2, replace=False).tolist()
            for h, a in enumerate(pair):
                hap, s = fx.haplotype(g, a)
                n = per_gene // 2 + (per_gene % 2 if h == 0 else 0)
                rs = simulate_reads(rng, hap, s, len(fx.dna[a]), n)
                self.reads += rs
                self.read_truth += [(g, a)] * len(rs)
                self.truth.append((g, a))
                lo, hi = max(0, s - 60), min(len(hap), s + len(fx.dna[a]) + 60)
                cons = hap[lo:hi]
                if not fx.gene_fwd[g]:
                    cons = revcomp(cons)
                self.consensus.append((g, cons, fx.cdna[a], a))
        order = rng.permutation(len(self.reads))
        self.reads = [self.reads[i] for i in order]
        self.read_truth = [self.read_truth[i] for i in order]
