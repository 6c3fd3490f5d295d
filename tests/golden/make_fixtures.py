#!/usr/bin/env python3
"""Generate the committed DATA fixtures under tests/golden/ from the read-only reference checkout.

Run in the authoring container only (needs /root/reference).  Everything written here is data
(sequences, coordinates, VCF rows, database JSON) -- never reference source code.

  hla_db_v0.14.1.json.gz   HLA part (hla_config + hla_sequences + metadata) of
                           data/v0.14.1/pbstarphase_20240826.json.gz, same schema
  chr6_hla_islands.json    the two non-N islands of test_data/refseq_faux/hg38_chr6_masked.fa.gz
                           (real GRCh38 around HLA-A / HLA-B) with their chr6 0-based starts
  hla_faux_database.json   test_data/HLA-faux/database.json (2 real IMGT alleles)
  variant_dbs/*.json       test_data/{CACNA1S,RNR1-faux,UGT1A1-faux,CYP2C8-faux,DPYD-sv-test}/database.json
  variant_vcfs.json        decoded rows of every test VCF (no htslib needed downstream)
  vcf/<set>/*.vcf.gz       the same test VCFs as they are (bgzip data files), and their tabix indices (*.vcf.gz.tbi, as htslib wrote them)
  test_reference.json      test_data/test_reference.fa
  cyp2d6_db_v0.14.1.json.gz  cyp2d6_config + cyp2d6_gene_def of the bundled DB
  cyp2d6_gene_def_v0.9.0.json.gz  cyp2d6_gene_def of data/v0.9.0/cpic_20240404.json.gz, the database test_load_variant_database
                           (src/cyp2d6/haplotyper.rs:918-933) loads
  gene_entries_v0.14.1.json.gz  gene_entries (the CPIC / PharmVar variant genes) of the bundled DB, for the full-panel tests
"""
import gzip, json, os, shutil, sys, glob

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

def dump(obj, name, gz=False):
    path = os.path.join(OUT, name)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()
    if gz:
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)
    print(name, len(data))

def read_fasta(path):
    op = gzip.open if path.endswith(".gz") else open
    seqs, name = {}, None
    with op(path, "rt") as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                name = line[1:].split()[0]; seqs[name] = []
            elif name is not None:
                seqs[name].append(line)
    return {k: "".join(v) for k, v in seqs.items()}

def main():
    db = json.load(gzip.open(f"{REF}/data/v0.14.1/pbstarphase_20240826.json.gz"))
    dump({"database_metadata": db["database_metadata"], "hla_config": db["hla_config"],
          "hla_sequences": db["hla_sequences"]}, "hla_db_v0.14.1.json.gz", gz=True)
    dump({"database_metadata": db["database_metadata"], "cyp2d6_config": db["cyp2d6_config"],
          "cyp2d6_gene_def": db["cyp2d6_gene_def"]}, "cyp2d6_db_v0.14.1.json.gz", gz=True)

    old = json.load(gzip.open(f"{REF}/data/v0.9.0/cpic_20240404.json.gz"))
    dump({"database_metadata": old["database_metadata"], "cyp2d6_gene_def": old["cyp2d6_gene_def"]}, "cyp2d6_gene_def_v0.9.0.json.gz", gz=True)
    dump({"database_metadata": db["database_metadata"], "gene_entries": db["gene_entries"]}, "gene_entries_v0.14.1.json.gz", gz=True)

    chr6 = read_fasta(f"{REF}/test_data/refseq_faux/hg38_chr6_masked.fa.gz")["chr6"]
    islands, i, n = [], 0, len(chr6)
    while i < n:
        if chr6[i] in "Nn":
            i += 1; continue
        j = i
        while j < n and chr6[j] not in "Nn":
            j += 1
        islands.append({"chrom": "chr6", "start": i, "end": j, "sequence": chr6[i:j].upper()})
        i = j
    dump({"chrom_length": n, "islands": islands}, "chr6_hla_islands.json")

    dump(json.load(open(f"{REF}/test_data/HLA-faux/database.json")), "hla_faux_database.json")
    for d in ["CACNA1S", "RNR1-faux", "UGT1A1-faux", "CYP2C8-faux", "DPYD-sv-test"]:
        dump(json.load(open(f"{REF}/test_data/{d}/database.json")), f"variant_dbs/{d}.json")
    vcfs = {}
    for path in sorted(glob.glob(f"{REF}/test_data/*/*.vcf.gz")):
        key = "/".join(path.split("/")[-2:])
        header, rows = [], []
        for line in gzip.open(path, "rt"):
            line = line.rstrip("\n")
            if line.startswith("##"):
                header.append(line)
            elif line.startswith("#"):
                cols = line[1:].split("\t")
            else:
                rows.append(dict(zip(cols, line.split("\t"))))
        vcfs[key] = {"header": header, "columns": cols, "rows": rows}
        # the data file itself (bgzip; the reference's tests read it through htslib): input of the library's VCF reader
        os.makedirs(os.path.join(OUT, "vcf", os.path.dirname(key)), exist_ok=True)
        shutil.copyfile(path, os.path.join(OUT, "vcf", key))
        if os.path.exists(path + ".tbi"):                                  # the tabix index htslib wrote for it (a data file of the reference's tests)
            shutil.copyfile(path + ".tbi", os.path.join(OUT, "vcf", key + ".tbi"))
    dump(vcfs, "variant_vcfs.json")
    dump(read_fasta(f"{REF}/test_data/test_reference.fa"), "test_reference.json")
    for d in ["HLA_configs", "CYP2D6_configs"]:
        for path in sorted(glob.glob(f"{REF}/test_data/{d}/*.json")):
            dump(json.load(open(path)), f"{d}/{os.path.basename(path)}")

if __name__ == "__main__":
    main()
