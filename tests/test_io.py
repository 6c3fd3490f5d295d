"""SURVEY.md 8(f) row f2 through the C ABI (host only): the BAM / BAI and VCF readers.

VCF: the reference's own test files (bgzip, tests/golden/vcf/) read by the library == the rows decoded with Python's gzip module
(tests/golden/variant_vcfs.json), and every diplotyper scenario again from the FILES (database file + VCF file -> sp_variant_problem ==
the test-side restatement).  BAM: files written here by an independent Python encoder of the published format (BGZF blocks, records, BAI
bins + linear index); region fetches == a Python filter of the same records, with and without the index."""
import gzip
import json
import os
import struct
import zlib

import numpy as np
import pytest

import variant_glue as vg
from test_database import vcf_alleles, vcf_deletions, check_problem
from test_oracle_variant import CASES, SV_CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DECODED = json.load(open(os.path.join(GOLDEN, "variant_vcfs.json")))


@pytest.fixture(scope="module")
def D(pkg):
    return pkg.database


# ------------------------------------------------------------------ VCF
@pytest.mark.parametrize("key", sorted(DECODED))
def test_vcf_file_equals_decoded_rows(D, key):
    vcf = D.Vcf(os.path.join(GOLDEN, "vcf", key))
    want = DECODED[key]
    sample = [c for c in want["columns"] if c not in ("CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT")]
    assert vcf.samples() == sample
    chroms = sorted({r["CHROM"] for r in want["rows"]})
    got = [a for c in chroms for a in vcf.alleles(c)]
    exp = [a for c in chroms for a in vcf_alleles({"columns": want["columns"], "rows": [r for r in want["rows"] if r["CHROM"] == c]})]
    # rows whose GT is missing / haploid are skipped by both; every other ALT allele is a row
    assert got == exp
    if "DPYD-sv-test" in key and "del" in key:
        assert [d for c in chroms for d in vcf.deletions(c)] == vcf_deletions(want) and len(vcf_deletions(want)) >= 1
    # a window: only the records that overlap it
    if want["rows"]:
        r = want["rows"][len(want["rows"]) // 2]
        p0 = int(r["POS"]) - 1
        inside = vcf.alleles(r["CHROM"], p0, p0 + 1)
        assert inside and all(a[0] <= p0 < a[0] + len(a[1]) for a in inside)
        assert vcf.alleles(r["CHROM"], p0 + len(r["REF"]) + 5000, p0 + len(r["REF"]) + 5001) == [a for a in got if a[0] <= p0 + len(r["REF"]) + 5000 < a[0] + len(a[1])]


@pytest.mark.parametrize("key", sorted(DECODED))
def test_vcf_through_the_reference_tabix_index_equals_the_scan(D, tmp_path, key):
    """bcf::IndexedReader::fetch (src/diplotyper.rs:569-575,800): with the .tbi htslib wrote for the reference's test files beside them the library opens a file by its
    header and reads records through the index; the same file without its index is read as a whole.  Same rows either way, region by region."""
    import shutil
    src = os.path.join(GOLDEN, "vcf", key)
    assert os.path.exists(src + ".tbi")
    alone = str(tmp_path / "alone.vcf.gz")
    shutil.copyfile(src, alone)
    ix, scan = D.Vcf(src), D.Vcf(alone)
    assert ix.index_info()[0] is True and scan.index_info()[0] is False
    assert ix.samples() == scan.samples()
    rows = DECODED[key]["rows"]
    for c in sorted({r["CHROM"] for r in rows}) + ["chrNone"]:
        assert ix.alleles(c) == scan.alleles(c)
        pos = sorted({int(r["POS"]) - 1 for r in rows if r["CHROM"] == c})
        for p0 in pos:
            for a, b in ((p0 - 50, p0 + 50), (p0, p0 + 1), (p0 + 1, p0 + 2), (max(0, p0 - 3), p0)):
                a = max(0, a)
                assert ix.alleles(c, a, b) == scan.alleles(c, a, b)
                if "DPYD-sv-test" in key:
                    assert ix.deletions(c, a, b) == scan.deletions(c, a, b)
        if "DPYD-sv-test" in key:
            assert ix.deletions(c) == scan.deletions(c)


def tbx_reg2bin(beg, end, min_shift=14, depth=5):
    """the smallest bin that holds [beg, end) (CSIv1 / tabix specification)"""
    end -= 1
    s, t = min_shift, ((1 << depth * 3) - 1) // 7
    for l in range(depth, 0, -1):
        if beg >> s == end >> s:
            return t + (beg >> s)
        s += 3
        t -= 1 << ((l - 1) * 3)
    return 0


def write_indexed_vcf(path, samples, records, block_bytes, kind):
    """records: sorted [(chrom, pos0, ref, alt, info, fmt, calls)].  A bgzip file of block_bytes blocks (lines cross block borders) and, written here from the published
    formats by an encoder of its own, a tabix (.tbi) or CSI (.csi, min_shift 12 / depth 6) index; a record's extent is INFO/END for a symbolic ALT, else its REF"""
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n"
    stream, spans = bytearray(head.encode()), []
    for (c, p0, ref, alt, info, fmt, calls) in records:
        line = f"{c}\t{p0 + 1}\t.\t{ref}\t{alt}\t.\tPASS\t{info}\t{fmt}\t" + "\t".join(calls) + "\n"
        spans.append((len(stream), len(stream) + len(line)))
        stream += line.encode()
    coff, out = [], bytearray()
    for i in range(0, len(stream), block_bytes):
        coff.append(len(out)); out += bgzf_block(bytes(stream[i:i + block_bytes]))
    coff.append(len(out))
    out += bgzf_block(b"")
    open(path, "wb").write(out)
    voff = lambda u: (coff[u // block_bytes] << 16) | (u % block_bytes)
    min_shift, depth = (14, 5) if kind == "tbi" else (12, 6)
    names = []
    for r in records:
        if r[0] not in names:
            names.append(r[0])
    per_ref = {n: ({}, {}, {}) for n in names}                  # bins -> chunks, linear, loffset
    for r, (u0, u1) in zip(records, spans):
        c, p0, ref, alt, info = r[0], r[1], r[2], r[3], r[4]
        end = p0 + len(ref)
        if alt.startswith("<"):
            for kv in info.split(";"):
                if kv.startswith("END="):
                    end = int(kv[4:])
        bins, linear, loff = per_ref[c]
        b = tbx_reg2bin(p0, end, min_shift, depth)
        chunks = bins.setdefault(b, [])
        if chunks and chunks[-1][1] == voff(u0):
            chunks[-1][1] = voff(u1)
        else:
            chunks.append([voff(u0), voff(u1)])
        loff[b] = min(loff.get(b, 1 << 63), voff(u0))
        for w in range(p0 >> 14, ((end - 1) >> 14) + 1):
            linear[w] = min(linear.get(w, 1 << 63), voff(u0))
    nm = b"".join(n.encode() + b"\0" for n in names)
    tbx_head = struct.pack("<6i", 2, 1, 2, 0, ord("#"), 0) + struct.pack("<i", len(nm)) + nm      # format VCF, col_seq 1, col_beg 2, col_end 0, meta '#', skip 0
    if kind == "tbi":
        body = bytearray(b"TBI\1" + struct.pack("<i", len(names)) + tbx_head)
    else:
        body = bytearray(b"CSI\1" + struct.pack("<3i", min_shift, depth, len(tbx_head)) + tbx_head + struct.pack("<i", len(names)))
    for n in names:
        bins, linear, loff = per_ref[n]
        body += struct.pack("<i", len(bins))
        for b in sorted(bins):
            body += struct.pack("<I", b)
            if kind == "csi":
                body += struct.pack("<Q", loff[b])
            body += struct.pack("<i", len(bins[b])) + b"".join(struct.pack("<QQ", x, y) for x, y in bins[b])
        if kind == "tbi":
            n_intv = (max(linear) + 1) if linear else 0
            body += struct.pack("<i", n_intv)
            last = 0
            for w in range(n_intv):
                last = linear.get(w, last)
                body += struct.pack("<Q", last)
    comp = bytearray()
    for i in range(0, len(body), 60000):
        comp += bgzf_block(bytes(body[i:i + 60000]))
    comp += bgzf_block(b"")
    open(path + "." + kind, "wb").write(comp)


@pytest.mark.parametrize("kind", ["tbi", "csi"])
def test_vcf_region_fetch_on_a_large_file_reads_a_few_blocks(D, tmp_path, kind):
    """a VCF of WGS shape (60,000 records, three chromosomes, multi-allelic records, symbolic deletions with INFO/END that reach far to the right): region fetches through
    an index written by the test's own encoder (tabix, and CSI with another bin scheme) == the scan of the same file without an index, and a fetch parses dozens of lines,
    not the file"""
    import shutil
    rng = np.random.default_rng(4)
    recs = []
    for c, n, length in (("chr1", 30000, 50_000_000), ("chr6", 20000, 30_000_000), ("chr22", 10000, 8_000_000)):
        for p0 in sorted(rng.choice(length, n, replace=False).tolist()):
            u = rng.random()
            gt = str(rng.choice(["0/1", "1/1", "0|1", "1|0", "./.", "1/2"]))
            if u < 0.02:
                end = p0 + int(rng.integers(50, 300000))
                recs.append((c, p0, "N", "<DEL>", f"SVTYPE=DEL;END={end}", "GT", [gt if gt != "1/2" else "0/1"]))
            elif u < 0.1:
                recs.append((c, p0, "ACGT"[int(rng.integers(4))] * int(rng.integers(1, 6)), "A,T", "DP=30", "GT:PS", [gt + ":" + str(p0 - p0 % 1000)]))
            else:
                recs.append((c, p0, "ACGT"[int(rng.integers(4))], "ACGT"[int(rng.integers(4))], "DP=30", "GT:PS", [gt + ":" + str(p0 - p0 % 1000)]))
    path = str(tmp_path / f"big_{kind}.vcf.gz")
    write_indexed_vcf(path, ["S1"], recs, 20000, kind)
    alone = str(tmp_path / "alone.vcf.gz")
    shutil.copyfile(path, alone)
    ix, scan = D.Vcf(path), D.Vcf(alone)
    assert ix.index_info() == (True, 0) and scan.index_info()[0] is False
    small = [r for r in recs if not r[3].startswith("<")]
    for q in range(60):
        r = recs[int(rng.integers(len(recs)))] if q % 2 else small[int(rng.integers(len(small)))]
        a = max(0, r[1] - int(rng.integers(0, 60)))
        b = r[1] + int(rng.integers(1, 120))
        before = ix.index_info()[1]
        got = ix.alleles(r[0], a, b)
        assert got == scan.alleles(r[0], a, b) and (got or r[3].startswith("<") or "." in r[6][0])
        assert ix.index_info()[1] - before < 1500                       # (one or two 20,000-byte blocks of ~40-byte lines, not 60,000 records)
        try:
            want = ("ok", scan.deletions(r[0], a, b))
        except Exception as e:
            want = ("error", str(e))
        try:
            have = ("ok", ix.deletions(r[0], a, b))
        except Exception as e:
            have = ("error", str(e))
        assert have == want
    assert ix.alleles("chr22") == scan.alleles("chr22") and ix.alleles("chrUn", 0, 1000) == []
    dels = [(r[0], r[1]) for r in recs if r[3].startswith("<")]
    c, p0 = dels[len(dels) // 2]
    assert any(d[0] == p0 for d in ix.deletions(c, p0, p0 + 1))


def test_vcf_plain_gzip_and_errors(D, pkg, tmp_path):
    key = "UGT1A1-faux/different_phaseset_001.vcf.gz"
    text = gzip.open(os.path.join(GOLDEN, "vcf", key)).read()
    (tmp_path / "plain.vcf").write_bytes(text)
    (tmp_path / "one_member.vcf.gz").write_bytes(gzip.compress(text))
    ref = D.Vcf(os.path.join(GOLDEN, "vcf", key))
    chrom = DECODED[key]["rows"][0]["CHROM"]
    for name in ("plain.vcf", "one_member.vcf.gz"):
        assert D.Vcf(str(tmp_path / name)).alleles(chrom) == ref.alleles(chrom) != []
    with pytest.raises(pkg.StarphaseError, match="cannot open"):
        D.Vcf(str(tmp_path / "missing.vcf.gz"))
    (tmp_path / "headless.vcf").write_text("chr1\t5\t.\tA\tC\t.\tPASS\t.\tGT\t0/1\n")
    with pytest.raises(pkg.StarphaseError, match="header"):
        D.Vcf(str(tmp_path / "headless.vcf"))
    with pytest.raises(pkg.StarphaseError, match="no sample"):
        ref.alleles(chrom, sample="nobody")
    # load_sv_vcf_variants bails on a record without SVTYPE and on a DEL without END (src/diplotyper.rs:795-812)
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
    (tmp_path / "sv1.vcf").write_text(head + "chr1\t100\t.\tA\t<DEL>\t.\tPASS\tEND=500\tGT\t0/1\n")
    with pytest.raises(pkg.StarphaseError, match="SVTYPE"):
        D.Vcf(str(tmp_path / "sv1.vcf")).deletions("chr1")
    (tmp_path / "sv2.vcf").write_text(head + "chr1\t100\t.\tA\t<DEL>\t.\tPASS\tSVTYPE=DEL\tGT\t0/1\n")
    with pytest.raises(pkg.StarphaseError, match="END"):
        D.Vcf(str(tmp_path / "sv2.vcf")).deletions("chr1")
    (tmp_path / "sv3.vcf").write_text(head + "chr1\t100\t.\tA\t<DEL>\t.\tPASS\tSVTYPE=DEL;END=500\tGT:PS\t1|0:77\n"
                                      + "chr1\t900\t.\tA\t<INS>\t.\tPASS\tSVTYPE=INS;END=900\tGT\t0/1\n" + "chr2\t100\t.\tA\t<DEL>\t.\tPASS\tSVTYPE=DEL;END=300\tGT\t1/1\n")
    v = D.Vcf(str(tmp_path / "sv3.vcf"))
    assert v.deletions("chr1") == [(99, 500, 3, 77)] and v.deletions("chr2") == [(99, 300, 4, None)] and v.deletions("chr1", 600, 700) == []


@pytest.mark.parametrize("case", CASES + [("DPYD-sv-test", "DPYD-sv-test/empty_small.vcf.gz", True, None, None, c[0]) for c in SV_CASES],
                         ids=lambda c: c[1] + ("+" + c[5] if len(c) > 5 else ""))
def test_scenarios_from_the_files(D, oracle, case):
    """database FILE + VCF FILE(s) -> the integer problem, == the restatement fed with the decoded rows"""
    db_name, vcf_key, with_ref = case[0], case[1], case[2]
    sv_key = case[5] if len(case) > 5 else None
    _name, want = vg.load_case(oracle, db_name, vcf_key, with_ref, sv_vcf_key=sv_key)
    db = D.Database(os.path.join(GOLDEN, "variant_dbs", db_name + ".json"))
    gene_name, chrom = db.gene_entries()[0]
    genome = json.load(open(os.path.join(GOLDEN, "test_reference.json"))) if with_ref else None
    gene = db.variant_gene(gene_name, genome[chrom] if genome else None)
    small = D.Vcf(os.path.join(GOLDEN, "vcf", vcf_key)).alleles(chrom)
    dels = D.Vcf(os.path.join(GOLDEN, "vcf", sv_key)).deletions(chrom) if sv_key else []
    check_problem(D, gene, gene.problem(small, dels), want)


# ------------------------------------------------------------------ BAM: an independent encoder of the published format
SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
CIGAR_OPS = "MIDNSHP=X"


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def bam_record(ref_id, pos, qname, flag, mapq, cigar, seq):
    span = sum(n for op, n in cigar if op in "MDN=X") or 1
    name = qname.encode() + b"\0"
    packed = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= SEQ_CODE[c] << (0 if i & 1 else 4)
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(name), mapq, reg2bin(pos, pos + span), len(cigar), flag, len(seq), -1, -1, 0)
    body += name + b"".join(struct.pack("<I", n << 4 | CIGAR_OPS.index(op)) for op, n in cigar) + bytes(packed) + b"\xff" * len(seq)
    body += b"NMC\x00"                                                         # one aux tag behind the qualities
    return struct.pack("<i", len(body)) + body


def bgzf_block(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def write_bam(path, refs, records, block_bytes, index=True):
    """refs: [(name, length)]; records: sorted [(ref_id, pos, qname, flag, mapq, cigar, seq)].  Blocks of block_bytes uncompressed bytes
    (records cross block borders); BAI written beside the file when index."""
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(f"@SQ\tSN:{n}\tLN:{l}\n".encode() for n, l in refs)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for n, l in refs:
        head += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    stream, spans = bytearray(head), []
    for r in records:
        enc = bam_record(*r)
        spans.append((len(stream), len(stream) + len(enc)))
        stream += enc
    blocks = [bytes(stream[i:i + block_bytes]) for i in range(0, len(stream), block_bytes)]
    coff, out = [], bytearray()
    for b in blocks:
        coff.append(len(out)); out += bgzf_block(b)
    coff.append(len(out))
    out += bgzf_block(b"")                                                     # the EOF marker block
    open(path, "wb").write(out)
    voff = lambda u: (coff[u // block_bytes] << 16) | (u % block_bytes)
    if not index:
        return
    bai = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
    for rid in range(len(refs)):
        bins, linear = {}, {}
        for (r, (u0, u1)) in zip(records, spans):
            if r[0] != rid:
                continue
            span = sum(n for op, n in r[5] if op in "MDN=X") or 1
            b = reg2bin(r[1], r[1] + span)
            chunks = bins.setdefault(b, [])
            if chunks and chunks[-1][1] == voff(u0):
                chunks[-1][1] = voff(u1)
            else:
                chunks.append([voff(u0), voff(u1)])
            for w in range(r[1] >> 14, ((r[1] + span - 1) >> 14) + 1):
                linear[w] = min(linear.get(w, 1 << 63), voff(u0))
        bai += struct.pack("<i", len(bins))
        for b in sorted(bins):
            bai += struct.pack("<Ii", b, len(bins[b])) + b"".join(struct.pack("<QQ", c[0], c[1]) for c in bins[b])
        n_intv = (max(linear) + 1) if linear else 0
        bai += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):
            last = linear.get(w, last)
            bai += struct.pack("<Q", last)
    open(path + ".bai", "wb").write(bai)


def make_records(rng, refs, n):
    recs = []
    for i in range(n):
        rid = int(rng.integers(0, len(refs)))
        L = int(rng.integers(30, 1500))
        pos = int(rng.integers(0, refs[rid][1] - 2 * L))
        seq = "".join(rng.choice(list("ACGTN"), L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        kind = i % 4
        if kind == 0:
            cigar = [("M", L)]
        elif kind == 1:
            cigar = [("S", 5), ("M", L - 15), ("I", 4), ("M", 6), ("D", 37)]
        elif kind == 2:
            cigar = [("=", L // 2), ("X", 1), ("N", 2000), ("=", L - L // 2 - 1)]
        else:
            cigar = [("H", 10), ("M", L)]
        flag = [0, 16, 256, 2048, 1024][int(rng.integers(0, 5))]
        recs.append((rid, pos, f"m84/{i % (n - 7)}/ccs", flag, int(rng.integers(0, 61)), cigar, seq))
    recs.sort(key=lambda r: (r[0], r[1]))
    return recs


def expected(records, refs, chrom, start, end, exclude=0):
    rid = [n for n, _ in refs].index(chrom)
    out = []
    for r in records:
        span = sum(n for op, n in r[5] if op in "MDN=X") or 1
        if r[0] == rid and r[1] < end and r[1] + span > start and not (r[3] & exclude):
            out.append(dict(qname=r[2], flag=r[3], mapq=r[4], pos=r[1], end=r[1] + span, cigar=[(CIGAR_OPS.index(op), n) for op, n in r[5]], seq=r[6]))
    return out


@pytest.mark.parametrize("block_bytes", [700, 65280])
def test_bam_fetch_equals_python_filter(D, tmp_path, block_bytes):
    rng = np.random.default_rng(block_bytes)
    refs = [("chr1", 400000), ("chr6", 2500000), ("chr22", 900000)]
    records = make_records(rng, refs, 900)
    path = str(tmp_path / "reads.bam")
    write_bam(path, refs, records, block_bytes, index=True)
    scan_path = str(tmp_path / "scan.bam")
    write_bam(scan_path, refs, records, block_bytes, index=False)
    indexed, scanned = D.Bam(path), D.Bam(scan_path)
    assert indexed.references() == refs == scanned.references()
    regions = [("chr6", 0, 2500000), ("chr1", 1000, 1001), ("chr22", 450000, 470000), ("chr6", 1 << 20, (1 << 20) + 5), ("chr1", 399000, 400000)]
    regions += [("chr6", a, a + int(rng.integers(1, 300000))) for a in rng.integers(0, 2200000, 12).tolist()]
    n_hits = 0
    for chrom, a, b in regions:
        want = expected(records, refs, chrom, a, b)
        assert indexed.fetch(chrom, a, b) == want
        assert scanned.fetch(chrom, a, b) == want
        n_hits += len(want)
    assert n_hits > 400
    # FLAG filter; QNAME dedupe across fetches (qnames_checked, src/hla/caller.rs:532,565-570)
    want = expected(records, refs, "chr6", 0, 2500000, exclude=0x900)
    assert indexed.fetch("chr6", 0, 2500000, exclude_flags=0x900) == want and any(r[3] & 0x900 for r in records)
    first = indexed.fetch("chr6", 0, 1200000, dedupe=True)
    second = indexed.fetch("chr6", 600000, 2500000, dedupe=True)
    names = [r["qname"] for r in first + second]
    assert len(names) == len(set(names)) and set(names) == {r["qname"] for r in expected(records, refs, "chr6", 0, 2500000)}
    indexed.forget()
    assert len(indexed.fetch("chr6", 0, 1200000, dedupe=True)) == len(first)


def test_bam_errors(D, pkg, tmp_path):
    with pytest.raises(pkg.StarphaseError, match="cannot open"):
        D.Bam(str(tmp_path / "none.bam"))
    (tmp_path / "text.bam").write_bytes(b"@HD\tVN:1.6\n" * 10)
    with pytest.raises(pkg.StarphaseError, match="BGZF"):
        D.Bam(str(tmp_path / "text.bam"))
    (tmp_path / "gz.bam").write_bytes(bgzf_block(b"not a bam at all, but long enough"))
    with pytest.raises(pkg.StarphaseError, match="not a BAM"):
        D.Bam(str(tmp_path / "gz.bam"))
    refs = [("chr1", 100000)]
    recs = make_records(np.random.default_rng(1), refs, 50)
    path = str(tmp_path / "ok.bam")
    write_bam(path, refs, recs, 900)
    bam = D.Bam(path)
    with pytest.raises(pkg.StarphaseError, match="no reference"):
        bam.fetch("chrX", 0, 10)
    # a flipped byte inside a block: the CRC (or the inflate) catches it
    data = bytearray(open(path, "rb").read())
    data[len(data) // 2] ^= 0x5A
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(data)
    with pytest.raises(pkg.StarphaseError):
        D.Bam(bad).fetch("chr1", 0, 100000)


def test_bam_reads_go_to_the_library(D, pkg, tmp_path):
    """the fetch hands out exactly what sp_seqset_upload takes (bases + n + 1 offsets); checked on the host side of the call"""
    import ctypes as C
    refs = [("chr6", 50000)]
    recs = make_records(np.random.default_rng(3), refs, 40)
    path = str(tmp_path / "r.bam")
    write_bam(path, refs, recs, 4096)
    bam = D.Bam(path)
    reads, n, bases, offs = C.POINTER(D.sp_bam_read)(), C.c_uint32(), C.c_void_p(), C.POINTER(C.c_uint64)()
    assert D._io().sp_bam_fetch(bam._h, b"chr6", 0, 50000, 0, 0, C.byref(reads), C.byref(n), C.byref(bases), C.byref(offs)) == 0
    assert n.value == 40 and offs[0] == 0 and [offs[i + 1] - offs[i] for i in range(40)] == [len(r[6]) for r in recs]
    assert C.string_at(bases.value, int(offs[40])).decode() == "".join(r[6] for r in recs)


def test_sv_records_outside_the_region_are_never_looked_at(D, pkg, tmp_path):
    """load_sv_vcf_variants reads SVTYPE / END of the records its region fetch returns (src/diplotyper.rs:796-815): a record elsewhere on the
    chromosome that lacks them is no error; one inside the region is"""
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
    rows = ["chr1\t1000\t.\tA\t<DEL>\t.\tPASS\tSVTYPE=DEL;END=5000\tGT\t0/1",
            "chr1\t900000\t.\tC\tT\t.\tPASS\tDP=30\tGT\t0/1"]                       # a plain SNV far away: no SVTYPE
    path = tmp_path / "sv.vcf"
    path.write_text(head + "\n".join(rows) + "\n")
    v = D.Vcf(str(path))
    assert v.deletions("chr1", 0, 10000) == [(999, 5000, 1, None)]
    assert v.deletions("chr1", 6000, 10000) == []
    with pytest.raises(pkg.StarphaseError, match="No INFO:SVTYPE"):
        v.deletions("chr1", 899990, 900010)
    rows[1] = "chr1\t2000\t.\tC\t<DEL>\t.\tPASS\tSVTYPE=DEL\tGT\t0/1"                # inside, a deletion without END
    path.write_text(head + "\n".join(rows) + "\n")
    with pytest.raises(pkg.StarphaseError, match="No INFO:END"):
        D.Vcf(str(path)).deletions("chr1", 0, 10000)


def test_bam_many_blocks_inflate_side_by_side(D, pkg, tmp_path):
    """a region of several hundred BGZF blocks: the reader inflates them in batches on a few threads (sp_io.hip, Bgzf::load); the records
    and the SEQ fields as stored (sp_bam_last_seq4: the input of sp_seqset_upload_format(SP_SEQ_BAM4)) equal the Python filter's"""
    rng = np.random.default_rng(77)
    refs = [("chr6", 3000000)]
    records = make_records(rng, refs, 4000)
    path = str(tmp_path / "big.bam")
    write_bam(path, refs, records, 8000, index=True)                       # ~ 400 blocks
    assert os.path.getsize(path) > 1000000
    bam = D.Bam(path)
    for a, b in ((0, 3000000), (1000000, 1900000)):
        want = expected(records, refs, "chr6", a, b)
        got = bam.fetch("chr6", a, b)
        assert got == want and len(want) > 1000
        blob, offs, lens = bam.last_seq4()
        assert lens.tolist() == [len(r["seq"]) for r in want] and int(offs[-1]) == len(blob)
        eb, eo, el = pkg.ffi.encode_bam4([r["seq"] for r in want])
        assert eo.tolist() == offs.tolist() and bytes(eb) == bytes(blob)
    scan = D.Bam(path)
    os.remove(path + ".bai")
    scan = D.Bam(path)
    assert scan.fetch("chr6", 1000000, 1900000) == expected(records, refs, "chr6", 1000000, 1900000)


def test_bam_damage_behind_the_region_does_not_fail_the_fetch(D, pkg, tmp_path):
    """the reader inflates batches of blocks ahead of the records it hands out: a damaged or truncated block that lies behind everything the region
    needs ends the batch in front of it, and is an error only for a fetch that gets there (sp_io.hip Bgzf::load)"""
    rng = np.random.default_rng(5)
    refs = [("chr6", 3000000)]
    records = make_records(rng, refs, 3000)
    path = str(tmp_path / "big.bam")
    write_bam(path, refs, records, 6000, index=True)
    data = bytearray(open(path, "rb").read())
    early = expected(records, refs, "chr6", 0, 400000)
    assert len(early) > 100
    for name, frac in (("flip16", 0.16), ("flip18", 0.18), ("flip22", 0.22), ("flip30", 0.30), ("flip90", 0.9), ("cut17", 0.17), ("cut25", 0.25), ("cut90", 0.9)):
        d2 = bytearray(data)
        if name.startswith("flip"):
            d2[int(len(d2) * frac)] ^= 0x5A
        else:
            d2 = d2[:int(len(d2) * frac)]
        bad = str(tmp_path / (name + ".bam"))
        open(bad, "wb").write(d2)
        open(bad + ".bai", "wb").write(open(path + ".bai", "rb").read())
        assert D.Bam(bad).fetch("chr6", 0, 400000) == early
        with pytest.raises(pkg.StarphaseError):
            D.Bam(bad).fetch("chr6", 0, 3000000)


# ------------------------------------------------------------------ reference FASTA
def write_fasta(path, seqs, width, newline="\n", lower=False, index=False, describe=False):
    """a FASTA file the way samtools faidx expects it (fixed line width per sequence) and, optionally, its .fai"""
    fai, parts, size = [], [], 0
    for name, bases in seqs.items():
        header = ((f">{name} test sequence" if describe else f">{name}") + newline).encode()
        parts.append(header); size += len(header)
        fai.append((name, len(bases), size, width, width + len(newline)))
        body = (bases.lower() if lower else bases).encode()
        lines = [body[i:i + width] for i in range(0, len(body), width)]
        block = newline.encode().join(lines) + (newline.encode() if lines else b"")
        parts.append(block); size += len(block)
    with open(path, "wb") as f:
        f.write(b"".join(parts))
    if index:
        open(str(path) + ".fai", "w").write("".join("\t".join(map(str, row)) + "\n" for row in fai))


@pytest.mark.parametrize("index", [False, True])
@pytest.mark.parametrize("width,newline", [(10, "\n"), (7, "\n"), (60, "\r\n")])
def test_fasta_slices(D, tmp_path, index, width, newline):
    """ReferenceGenome::get_slice on the reference's own test genome (test_data/test_reference.fa, decoded in tests/golden/test_reference.json)
    plus a long random chromosome: every slice equals the Python slice, with and without the .fai"""
    seqs = dict(json.load(open(os.path.join(GOLDEN, "test_reference.json"))))
    rng = np.random.default_rng(3)
    seqs["chrLong"] = "".join(rng.choice(list("ACGTN"), 5003, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
    path = tmp_path / "ref.fa"
    write_fasta(path, seqs, width, newline, lower=True, index=index, describe=True)
    fa = D.Fasta(str(path))
    assert fa.sequences() == [(k, len(v)) for k, v in seqs.items()]
    for name, bases in seqs.items():
        assert fa.fetch(name, 0, len(bases)) == bases                     # soft-masked (lower-case) input comes back upper-cased
        for _ in range(40):
            a = int(rng.integers(0, len(bases) + 1)); b = int(rng.integers(a, len(bases) + 1))
            assert fa.fetch(name, a, b) == bases[a:b]
    assert fa.fetch("chr1", 5, 5) == ""


def test_fasta_gzip_and_errors(D, pkg, tmp_path):
    seqs = dict(json.load(open(os.path.join(GOLDEN, "test_reference.json"))))
    plain = tmp_path / "ref.fa"
    write_fasta(plain, seqs, 10)
    gz = tmp_path / "ref.fa.gz"
    gzip.open(gz, "wb").write(open(plain, "rb").read())
    fa = D.Fasta(str(gz))
    assert fa.fetch("chr2", 8, 14) == seqs["chr2"][8:14] and fa.fetch("chr3", 0, 20) == seqs["chr3"]
    with pytest.raises(pkg.StarphaseError, match="no sequence chrX"):
        fa.fetch("chrX", 0, 1)
    with pytest.raises(pkg.StarphaseError, match="slice outside of chr1"):
        fa.fetch("chr1", 0, 21)
    with pytest.raises(pkg.StarphaseError, match="slice outside of chr1"):
        fa.fetch("chr1", 5, 4)
    with pytest.raises(pkg.StarphaseError, match="cannot open"):
        D.Fasta(str(tmp_path / "missing.fa"))
    bad = tmp_path / "bad.fa"
    open(bad, "w").write("ACGT\n>chr1\nACGT\n")
    with pytest.raises(pkg.StarphaseError, match="before the first header"):
        D.Fasta(str(bad))
    # an index that does not describe the file is reported, not trusted
    write_fasta(plain, seqs, 10, index=True)
    open(str(plain) + ".fai", "w").write("chr1\t20\t6\t10\t11\nchr2\t20\t5\t4\t5\n")
    fa = D.Fasta(str(plain))
    with pytest.raises(pkg.StarphaseError, match="does not describe"):
        fa.fetch("chr2", 0, 20)


# ------------------------------------------------------------------ damaged input
def damage(rnd, data, k):
    """k random edits of a file's bytes: a byte replaced, a stretch cut out, a stretch of noise put in"""
    b = bytearray(data)
    for _ in range(k):
        op = rnd.random()
        if op < 0.6 and b:
            b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif op < 0.8 and len(b) > 4:
            i = rnd.randrange(len(b)); del b[i:min(len(b), i + rnd.randrange(1, 64))]
        else:
            i = rnd.randrange(len(b) + 1); b[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 32)))
    return bytes(b)


@pytest.mark.parametrize("seed", [1, 2])
def test_damaged_files_are_errors_not_crashes(D, pkg, tmp_path, seed):
    """Every reader of the library is handed files with random damage (database JSON, VCF, FASTA + .fai, BAM + .bai; 4,800 such files were
    tried while writing this, 320 here): it either reads what is there or reports an error -- sizes and offsets taken from a file never
    index past what was read."""
    import random
    rnd = random.Random(seed)
    outcome = {"read": 0, "refused": 0}

    def attempt(fn):
        try:
            fn(); outcome["read"] += 1
        except pkg.StarphaseError:
            outcome["refused"] += 1
        except UnicodeDecodeError:                                # the file was read; its damaged text is not UTF-8 for the Python wrapper
            outcome["read"] += 1

    raw = open(os.path.join(GOLDEN, "hla_faux_database.json"), "rb").read()
    for _ in range(40):
        data = damage(rnd, raw, rnd.randrange(1, 6))
        attempt(lambda: (lambda db: (db.hla_genes(), db.gene_entries()))(D.Database(data)))
    vdir = os.path.join(GOLDEN, "vcf")
    vfiles = sorted(os.path.join(r, f) for r, _d, fs in os.walk(vdir) for f in fs if ".vcf" in f)
    for _ in range(40):
        f = rnd.choice(vfiles)
        text = gzip.open(f, "rb").read() if f.endswith(".gz") else open(f, "rb").read()
        p = tmp_path / "damaged.vcf"
        open(p, "wb").write(damage(rnd, text, rnd.randrange(1, 6)))

        def read_vcf():
            v = D.Vcf(str(p)); v.samples()
            for chrom in ("chr1", "chr10", "chr22", "chrM"):
                v.alleles(chrom); v.deletions(chrom)
        attempt(read_vcf)
    seqs = {"chr1": "ACGT" * 200, "chr2": "GATTACA" * 100}
    for _ in range(40):
        p = tmp_path / "damaged.fa"
        write_fasta(p, seqs, 60, index=True)
        target = str(p) if rnd.random() < 0.5 else str(p) + ".fai"
        data = damage(rnd, open(target, "rb").read(), rnd.randrange(1, 4))
        open(target, "wb").write(data)

        def read_fasta():
            fa = D.Fasta(str(p))
            for name, ln in fa.sequences():
                fa.fetch(name, 0, min(ln, 100)); fa.fetch(name, max(0, ln - 50), ln)
        attempt(read_fasta)
    refs = [("chr6", 200000), ("chr22", 100000)]
    recs = make_records(np.random.default_rng(3), refs, 80)
    for _ in range(40):
        p = str(tmp_path / "damaged.bam")
        write_bam(p, refs, recs, 4000)
        u = rnd.random()
        target = p if u < 0.6 else p + ".bai"
        if u > 0.9:
            os.remove(p + ".bai"); target = p                   # linear scan of a damaged file
        data = damage(rnd, open(target, "rb").read(), rnd.randrange(1, 4))
        open(target, "wb").write(data)

        def read_bam():
            b = D.Bam(p); b.references()
            for chrom, a0, b0 in (("chr6", 0, 200000), ("chr6", 50000, 60000), ("chr22", 0, 100000)):
                b.fetch(chrom, a0, b0, exclude_flags=0x900, dedupe=True)
        attempt(read_bam)
    assert outcome["read"] + outcome["refused"] == 160 and outcome["refused"] >= 40 and outcome["read"] >= 40
