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


class CypLocus:
    """Synthetic chr22-like CYP2D6 locus with the region sizes of Cyp2d6Config::default()
    (src/cyp2d6/definitions.rs:128-240): REP6 2,772 / D6 6,165 / link 2,919 / REP7 2,772 / spacer 1,564 / D7 5,938.
    D7 = D6 with ~3 % divergence, REP7 = REP6 with differences near its end (unrelated last 300 bases); templates as the extractor builds them
    (D6, D7, two hybrids, *5 signature, REP6, REP7, spacer, link_region), in full_allele() order."""
    TYPES = {"UNKNOWN": 0, "REP6": 1, "CYP2D6": 2, "link_region": 3, "REP7": 4, "spacer": 5, "CYP2D7": 6, "CYP2D6*5": 7, "Hybrid": 8}

    def __init__(self, seed=3):
        rng = np.random.default_rng(seed)
        rnd = lambda n: "".join(rng.choice(list("ACGT"), n))
        self.rep6 = rnd(2772)
        self.d6 = rnd(6165)
        self.link = rnd(2919)
        self.spacer = rnd(1564)
        d7 = mutate(rng, self.d6, 150, 20, 30)
        self.d7 = d7[:5938] if len(d7) >= 5938 else d7
        # REP7 = REP6 except near its end: two edited stretches and an unrelated 300-base tail.  The deletion allele (*5) removes
        # REP6's tail .. REP7's head (D6 and the link region with them): its REP is REP6 up to the middle of the differing zone and
        # REP7 after it, so the three REP versions are pairwise distinguishable.
        m1, m2, tail = mutate(rng, self.rep6[2200:2336], 4, 0, 1), mutate(rng, self.rep6[2336:2472], 4, 1, 0), rnd(300)
        self.rep7 = self.rep6[:2200] + m1 + m2 + tail
        self.rep_del = self.rep6[:2336] + m2 + tail
        self.left, self.right = rnd(3000), rnd(3000)
        cut = 2000
        hyb67 = self.d6[:cut] + self.d7[cut:]          # CYP2D6::CYP2D7 (starts as D6)
        hyb76 = self.d7[:cut] + self.d6[cut:]          # CYP2D7::CYP2D6
        star5 = self.left[-500:] + self.rep_del + self.spacer[:228]  # deletion signature: 500 bases upstream + 3,000 across the fused REP
        named = [("CYP2D6", "CYP2D6", self.d6), ("CYP2D6*5", "CYP2D6*5", star5), ("Hybrid", "CYP2D6::CYP2D7::exon2", hyb67),
                 ("CYP2D7", "CYP2D7", self.d7), ("Hybrid", "CYP2D7::CYP2D6::exon2", hyb76), ("REP6", "REP6", self.rep6),
                 ("REP7", "REP7", self.rep7), ("link_region", "link_region", self.link), ("spacer", "spacer", self.spacer)]
        named.sort(key=lambda x: x[1])                 # key order = sorted by full_allele() (haplotyper.rs:175-183)
        self.template_names = [n for _, n, _ in named]
        self.template_types = np.array([self.TYPES[t] for t, _, _ in named], np.int32)
        self.templates = [s for _, _, s in named]

    def haplotype(self, kind="normal"):
        if kind == "deletion":                         # *5: D6 gone, one fused REP left
            return self.left + self.rep_del + self.spacer + self.d7 + self.right
        if kind == "dup":
            return self.left + self.rep6 + self.d6 + self.link + self.rep7 + self.d6 + self.link + self.rep7 + self.spacer + self.d7 + self.right
        return self.left + self.rep6 + self.d6 + self.link + self.rep7 + self.spacer + self.d7 + self.right

    def reads(self, rng, n, kind="normal", mean_len=9000, sd_len=3000):
        hap = self.haplotype(kind)
        out = []
        for _ in range(n):
            ln = int(min(len(hap), max(2500, rng.normal(mean_len, sd_len))))
            s = int(rng.integers(0, len(hap) - ln + 1))
            out.append(hifi_errors(rng, hap[s:s + ln]))
        return out


class Chr22Locus:
    """BASELINE.json configs[2]: a synthetic chr22 window laid out with the coordinates of the database's own Cyp2d6Config
    (src/cyp2d6/definitions.rs:128-240: REP6 2,772 / CYP2D6 6,165 / link_region 2,919 / REP7 2,772 / spacer 1,564 / CYP2D7 5,938) so
    that the library builds the real 39 templates from it and the real variant table applies.
      * random ACGT (seed), then the reference allele of every database variant is written at its position (they all come from one
        real sequence, so they agree where they overlap);
      * CYP2D7 = CYP2D6 piece by piece (the 19 exon / intron pieces the hybrid breakpoints cut) with ~3 % substitutions and one
        contiguous insertion / deletion per piece that makes up the piece's real length difference: hybrids are meaningful;
      * REP7 = REP6 except near its end (two edited stretches and an unrelated 300-base tail), as in the real locus.
    Haplotypes are the window with CYP2D6 replaced by a star allele (database variants applied), with the *5 region deleted,
    or with a second gene copy (duplication, or the *68 / *36-style tandem arrangements) inserted upstream of the first."""

    def __init__(self, cyp2d6_config, cyp2d6_gene_def, seed=3, flank=4000):
        rng = np.random.default_rng(seed)
        self.config, self.gene_def = cyp2d6_config, cyp2d6_gene_def
        cc, reg = cyp2d6_config["cyp_coordinates"], cyp2d6_config["cyp_regions"]
        s5 = cyp2d6_config["cyp2d6_star5_del"]
        lo = min([v["start"] for v in cc.values()] + [s5["start"] - 500]) - flank
        hi = max([v["end"] for v in cc.values()] + [s5["end"] + 3000]) + flank
        self.start = lo
        seq = rng.choice(list("ACGT"), hi - lo).tolist()
        put = lambda p, s: seq.__setitem__(slice(p - lo, p - lo + len(s)), list(s))
        get = lambda a, b: "".join(seq[a - lo:b - lo])
        for d in cyp2d6_gene_def.values():
            for v in d["variants"]:
                put(v["position"], v["reference"])
        # D7 from D6, piece by piece
        def cuts(g):
            b = [cc[g]["start"]]
            for x in range(9, 0, -1):
                b += [reg[g][f"exon{x}"]["start"], reg[g][f"exon{x}"]["end"]]
            return b + [cc[g]["end"]]
        b6, b7 = cuts("CYP2D6"), cuts("CYP2D7")
        for i in range(len(b6) - 1):
            piece = list(get(b6[i], b6[i + 1]))
            want = b7[i + 1] - b7[i]
            for p in np.flatnonzero(rng.random(len(piece)) < 0.03):
                piece[p] = "ACGT"[("ACGT".index(piece[p]) + 1 + int(rng.integers(3))) % 4]
            if want < len(piece):
                at = int(rng.integers(10, max(11, want - 10)))
                del piece[at:at + len(piece) - want]
            elif want > len(piece):
                at = int(rng.integers(10, max(11, len(piece) - 10)))
                piece[at:at] = rng.choice(list("ACGT"), want - len(piece)).tolist()
            put(b7[i], "".join(piece))
        # REP7 from REP6
        rep6 = get(cc["REP6"]["start"], cc["REP6"]["end"])
        m1, m2 = mutate(rng, rep6[2200:2336], 4, 0, 1), mutate(rng, rep6[2336:2472], 4, 1, 0)
        rep7 = rep6[:2200] + m1 + m2 + "".join(rng.choice(list("ACGT"), 300))
        assert len(rep7) == cc["REP7"]["end"] - cc["REP7"]["start"]
        put(cc["REP7"]["start"], rep7)
        self.sequence = "".join(seq)
        self.cc, self.s5 = cc, s5

    def slice(self, a, b):
        return self.sequence[a - self.start:b - self.start]

    def star_allele(self, star, drop=(), add=()):
        """the CYP2D6 region carrying the variants of allele `star` ("4.001"); drop: variant ids left out, add: (star, id) pairs of variants
        borrowed from other alleles -- a novel allele for the deep labels"""
        d = next(v for v in self.gene_def.values() if v["star_allele"] == star)
        a, b = self.cc["CYP2D6"]["start"], self.cc["CYP2D6"]["end"]
        s = self.slice(a, b)
        variants = [v for v in d["variants"] if v["id"] not in drop]
        for other, vid in add:
            od = next(v for v in self.gene_def.values() if v["star_allele"] == other)
            variants.append(next(v for v in od["variants"] if v["id"] == vid))
        for v in sorted(variants, key=lambda v: -v["position"]):               # right to left: earlier coordinates stay valid
            p = v["position"] - a
            assert s[p:p + len(v["reference"])] == v["reference"], (star, v)
            s = s[:p] + v["alternate"] + s[p + len(v["reference"]):]
        return s

    def hybrid(self, name):
        """a template-shaped hybrid gene, e.g. "CYP2D6::CYP2D7::exon2" (definitions.rs:395-446)"""
        kind, cut = name.rsplit("::", 1)
        reg = self.config["cyp_regions"]
        side = "end" if cut.startswith("exon") else "start"
        x = cut.replace("exon", "").replace("intron", "")
        bp1, bp2 = reg["CYP2D6"][f"exon{x}"][side], reg["CYP2D7"][f"exon{x}"][side]
        cc = self.cc
        if kind == "CYP2D6::CYP2D7":
            return self.slice(cc["CYP2D7"]["start"], bp2) + self.slice(bp1, cc["CYP2D6"]["end"])
        return self.slice(cc["CYP2D6"]["start"], bp1) + self.slice(bp2, cc["CYP2D7"]["end"])

    def haplotype(self, genes):
        """genes: list of gene bodies in chromosome order (the one next to REP6 first); None = the *5 deletion haplotype.
        One body: the window with CYP2D6 replaced.  More: REP6 body1 link REP7 body2 link REP7 ... spacer CYP2D7."""
        cc = self.cc
        if genes is None:
            return self.slice(self.start, self.s5["start"]) + self.slice(self.s5["end"], self.start + len(self.sequence))
        left = self.slice(self.start, cc["CYP2D6"]["start"])                    # flank + REP6 (+ the short piece up to the gene)
        link = self.slice(cc["CYP2D6"]["end"], cc["REP7"]["end"])               # link_region + REP7
        right = self.slice(cc["REP7"]["end"], self.start + len(self.sequence))  # spacer + CYP2D7 + flank
        return left + link.join(genes) + link + right

    def sample(self, rng, haplotypes, n_reads, lo=3000, hi=8000, errors=True):
        """targeted-style reads: fragments of lo..hi bases drawn from the haplotypes in proportion to their lengths"""
        total = sum(len(h) for h in haplotypes)
        reads = []
        for h in haplotypes:
            for _ in range(int(round(n_reads * len(h) / total))):
                ln = int(rng.integers(lo, hi + 1))
                s = int(rng.integers(0, len(h) - ln + 1))
                r = h[s:s + ln]
                reads.append(hifi_errors(rng, r) if errors else r)
        order = rng.permutation(len(reads))
        return [reads[i] for i in order]


def chain_pair_problem(n_d6, n_reads, rng):
    """A synthetic find_best_chain_pair input (src/cyp2d6/chaining.rs:421-566) whose chain enumeration explodes the way duplication-rich
    samples do: labels REP6, n_d6 CYP2D6 alleles, link_region, REP7, spacer, CYP2D7; every read observes a stretch of REP6 -> D6_a ->
    link -> REP7 -> D6_b -> ... and scores every label per observed segment.  Returns the keyword arguments of
    Context.cyp_best_chain_pair (configuration tables of the bundled database)."""
    cfg = json.load(gzip.open(os.path.join(GOLDEN, "cyp2d6_db_v0.14.1.json.gz")))["cyp2d6_config"]
    codes = {"REP6": 1, "CYP2D6": 2, "link_region": 3, "REP7": 4, "spacer": 5, "CYP2D7": 6}
    labels = [("REP6", None)] + [("CYP2D6", str(k + 1)) for k in range(n_d6)] + [("link_region", None), ("REP7", None), ("spacer", None), ("CYP2D7", None)]
    H = len(labels)
    REP6, LINK, REP7, SP, D7 = 0, n_d6 + 1, n_d6 + 2, n_d6 + 3, n_d6 + 4
    rco, co, items, rwo, ed, ov = [0], [0], [], [0], [], []
    for r in range(n_reads):
        a, b = int(rng.integers(1, n_d6 + 1)), int(rng.integers(1, n_d6 + 1))
        kind = r % 4
        chain = [REP6, a, LINK, REP7][: int(rng.integers(2, 5))] if kind == 0 else [a, LINK, REP7, b, LINK][: int(rng.integers(2, 6))] if kind == 1 \
            else [b, LINK, REP7, SP, D7][int(rng.integers(0, 3)):] if kind == 2 else [LINK, REP7, b, LINK, REP7][: int(rng.integers(2, 6))]
        items += chain
        co.append(len(items)); rco.append(len(co) - 1)
        for h in chain:
            row = [(int(rng.integers(20, 60)), 1.0) if labels[x][0] == labels[h][0] else (int(rng.integers(300, 900)), 0.9) for x in range(H)]
            row[h] = (int(rng.integers(0, 4)), 1.0)
            ed.append([e for e, _ in row]); ov.append([o for _, o in row])
        rwo.append(len(ed))
    return dict(hap_type=np.array([codes[t] for t, _ in labels], np.int32), hap_subtype=[s for _, s in labels],
                translate=sorted(cfg["cyp_translate"].items()), connections=sorted(tuple(x) for x in cfg["inferred_connections"]),
                singletons=sorted(cfg["unexpected_singletons"]), read_chain_off=np.array(rco, np.uint32), chain_off=np.array(co, np.uint32),
                chain_items=np.array(items, np.uint32), read_w_off=np.array(rwo, np.uint32), w_ed=np.array(ed, np.uint64), w_ov=np.array(ov, np.float64),
                infer=False, normalize_all=True, ignore_limits=False, penalties=(4.0, 2.0, 10.0, 2.0))


class SyntheticHlaFixture:
    """An HLA database of the SHAPE of the reference's current release (data/v2.0.0/pbstarphase_20251106.db_stat.txt: 41,374 alleles over 11 genes,
    23,152 with DNA; the blob itself is not shipped): class I genes of ~3.5 kb, class II genes of 11-16 kb, a few dozen lineages per gene ~1 % apart,
    alleles a handful of substitutions away from their lineage (indels only in one intron so that the cDNA stays a coordinate slice), partial DNA
    alleles, three absent-capable genes (DRB3 / 4 / 5) and the normalising gene DRB1 (src/hla/alleles.rs:18-69).  Same interface as HlaFixture.
    scale < 1 shrinks the allele counts (tests); the sequences are random, only the shape is real."""

    GENES = [("HLA-A", 3500, True, 8600, 5300), ("HLA-B", 4080, False, 10700, 6800), ("HLA-C", 4300, False, 9600, 6400),
             ("HLA-DPA1", 9700, False, 800, 350), ("HLA-DPB1", 11500, True, 2600, 900), ("HLA-DQA1", 6500, True, 800, 400),
             ("HLA-DQB1", 7500, False, 2800, 1100), ("HLA-DRB1", 13500, False, 4300, 1000), ("HLA-DRB3", 13200, False, 600, 350),
             ("HLA-DRB4", 15500, False, 300, 250), ("HLA-DRB5", 13300, False, 274, 302)]
    ABSENT_CAPABLE = ("HLA-DRB3", "HLA-DRB4", "HLA-DRB5")
    NORMALIZING = ("HLA-DRB1",)

    def __init__(self, scale=1.0, seed=0):
        rng = np.random.default_rng(seed)
        self.buffer = 100
        self.genes = [g[0] for g in self.GENES]
        self.gene_fwd, self.gene_ref, self.exons, self.island = [], [], [], []
        self.ids, gene_of, self.dna, self.cdna, self.star = [], [], [], [], []
        self.absent_capable = [g in self.ABSENT_CAPABLE for g in self.genes]
        flank = 2500
        comp = bytes.maketrans(b"ACGT", b"TGCA")
        for gi, (name, length, fwd, n_all, n_dna) in enumerate(self.GENES):
            n_all, n_dna = max(6, int(n_all * scale)), max(4, int(min(n_dna, n_all) * scale))
            isl = rng.integers(0, 4, length + 2 * flank).astype(np.uint8)
            base = isl[flank:flank + length]                                    # hg38-forward gene body
            # exons: 6 stretches of 150-280 bases in the first 70 % of the gene (hg38 order), the indel-carrying intron behind them
            cuts = np.sort(rng.choice(np.arange(200, int(0.7 * length), 50), 12, replace=False))
            exons = [(int(cuts[2 * k]), int(min(cuts[2 * k] + rng.integers(150, 280), cuts[2 * k + 1] - 20))) for k in range(6)]
            indel_zone = (int(0.75 * length), int(0.9 * length))
            self.gene_fwd.append(1 if fwd else 0)
            self.gene_ref.append("".join("ACGT"[c] for c in isl[flank - self.buffer:flank + length + self.buffer]))
            self.exons.append([(a + self.buffer, b + self.buffer) for a, b in exons])
            self.island.append({"sequence": "".join("ACGT"[c] for c in isl), "start": 0, "gene_start": flank, "length": length})
            n_lin = max(3, min(40, n_all // 60))
            lineages = []
            for _ in range(n_lin):
                s = base.copy()
                pos = rng.choice(length, max(5, length // 100), replace=False)
                s[pos] = (s[pos] + rng.integers(1, 4, len(pos))) % 4
                lineages.append(s)
            with_dna = set(rng.choice(n_all, n_dna, replace=False).tolist())
            lut = np.frombuffer(b"ACGT", np.uint8)
            for a in range(n_all):
                s = lineages[int(rng.integers(0, n_lin))].copy()
                k = int(rng.integers(1, 9))                                     # (at least one: no two alleles of a lineage are the same sequence)
                pos = rng.choice(length, k, replace=False)
                s[pos] = (s[pos] + rng.integers(1, 4, k)) % 4
                fwd_bytes = lut[s].tobytes()
                cd = b"".join(fwd_bytes[x:y] for x, y in exons)
                if rng.random() < 0.1:                                          # a small indel in the designated intron
                    p = int(rng.integers(*indel_zone))
                    fwd_bytes = fwd_bytes[:p] + (fwd_bytes[p + 2:] if rng.random() < 0.5 else b"AC" + fwd_bytes[p:])
                lo, hi = 0, len(fwd_bytes)
                if rng.random() < 0.2:                                          # a partial allele: the ends are not in the record
                    lo, hi = int(rng.integers(0, 300)), len(fwd_bytes) - int(rng.integers(0, 300))
                dna_fwd = fwd_bytes[lo:hi]
                if not fwd:
                    dna_gene, cd = dna_fwd.translate(comp)[::-1], cd.translate(comp)[::-1]
                else:
                    dna_gene = dna_fwd
                self.ids.append(f"HLA:SYN{gi:02d}{a:05d}")
                gene_of.append(gi)
                self.dna.append(dna_gene.decode() if a in with_dna else "")
                self.cdna.append(cd.decode())
                self.star.append(f"{a // 97 + 1:02d}:{a % 97 + 1:02d}")
        order = sorted(range(len(self.ids)), key=lambda i: self.ids[i])      # (generated in key order already)
        assert order == list(range(len(self.ids)))
        self.gene_of = np.array(gene_of, np.uint32)

    def dna_fwd(self, a):
        s = self.dna[a]
        return s if self.gene_fwd[self.gene_of[a]] else revcomp(s)

    def make_db(self, pkg, ctx):
        return pkg.HlaDb(ctx, self.gene_of, self.dna, self.cdna, self.gene_ref, self.gene_fwd, self.exons, self.buffer)

    def full_length_alleles(self, g, frac=0.97):
        lens = [len(self.dna[a]) for a in range(len(self.ids)) if self.gene_of[a] == g]
        cut = frac * max(lens)
        return [a for a in range(len(self.ids)) if self.gene_of[a] == g and len(self.dna[a]) >= cut]

    def haplotype(self, g, a):
        """the island with the gene body replaced by allele a (hg38 orientation) -> (sequence, allele start)"""
        I = self.island[g]
        al = self.dna_fwd(a)
        s = I["gene_start"]
        return I["sequence"][:s] + al + I["sequence"][s + I["length"]:], s
