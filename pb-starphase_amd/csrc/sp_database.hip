// sp_database.hip -- the database file (host only): reading PgxDatabase's JSON and flattening it into the inputs of the kernels.
//
// Replaces the serde side of PgxDatabase (src/database/pgx_database.rs:23-41: database_metadata, gene_collection, gene_entries,
// hla_config, hla_sequences, cyp2d6_config, cyp2d6_gene_def; load_json, src/util/file_io.rs:16-28) and what the callers do with it
// before any compute: the allele list HlaRealigner::new walks (src/hla/realigner.rs:42-91, alleles: src/hla/alleles.rs:332-344), the
// tables Cyp2d6Extractor::new needs (src/cyp2d6/haplotyper.rs:45-132, built by sp_cyp_db_create), load_database_haplotypes and the
// matching half of load_vcf_variants / load_sv_vcf_variants (src/diplotyper.rs:437-857) for the variant-typed genes.
// Map order = BTreeMap order: keys are compared as bytes (std::string), PgxGene's variant / haplotype ids as u64.
#include "sp_internal.h"
#include "sp_json.h"
#include "sp_defaults.h"
#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <set>
#include <tuple>

namespace {

using spj::Value;

bool read_file(const char* path, std::string& out, std::string& err) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    char buf[1 << 16]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    std::fclose(f);
    return true;
}

// MultiGzDecoder: members one after the other
bool gunzip(const std::string& in, std::string& out, std::string& err) {
    z_stream z{};
    if (inflateInit2(&z, 15 + 16) != Z_OK) { err = "inflateInit2 failed"; return false; }
    z.next_in = (Bytef*)in.data(); z.avail_in = (uInt)in.size();
    std::vector<char> buf(1 << 20);
    for (;;) {
        z.next_out = (Bytef*)buf.data(); z.avail_out = (uInt)buf.size();
        const int rc = inflate(&z, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&z); err = "corrupt gzip stream"; return false; }
        out.append(buf.data(), buf.size() - z.avail_out);
        if (rc == Z_STREAM_END) {
            if (z.avail_in == 0) break;
            if (inflateReset(&z) != Z_OK) { inflateEnd(&z); err = "corrupt gzip stream"; return false; }
        } else if (z.avail_in == 0 && z.avail_out != 0) { inflateEnd(&z); err = "truncated gzip stream"; return false; }
    }
    inflateEnd(&z);
    return true;
}

struct Region { std::string name, chrom; uint64_t start = 0, end = 0; bool fwd = true, absent = false; std::vector<uint64_t> es, ee; };

bool read_coordinates(const Value* v, std::string& chrom, uint64_t& start, uint64_t& end) {
    if (!v || v->kind != Value::Object || !v->get("chrom") || !v->get("start") || !v->get("end")) return false;
    chrom = v->get("chrom")->as_str(); start = (uint64_t)v->get("start")->as_int(); end = (uint64_t)v->get("end")->as_int();
    return true;
}

// GeneCollection::gene_dict -> regions in key order
bool read_gene_dict(const Value* dict, std::vector<Region>& out, std::string& err) {
    if (!dict || dict->kind != Value::Object) return true;
    for (const auto& kv : dict->obj) {
        Region r; r.name = kv.first;
        const Value& g = kv.second;
        if (!read_coordinates(g.get("coordinates"), r.chrom, r.start, r.end)) { err = "gene " + r.name + " has no coordinates"; return false; }
        r.fwd = g.get("is_forward_strand") ? g.get("is_forward_strand")->as_bool(true) : true;
        r.absent = g.get("is_absent_capable") ? g.get("is_absent_capable")->as_bool(false) : false;
        if (const Value* ex = g.get("exons")) for (const Value& e : ex->arr) {
            std::string c; uint64_t s, t;
            if (!read_coordinates(&e, c, s, t)) { err = "gene " + r.name + " has a malformed exon"; return false; }
            r.es.push_back(s); r.ee.push_back(t);
        }
        out.push_back(std::move(r));
    }
    std::sort(out.begin(), out.end(), [](const Region& a, const Region& b) { return a.name < b.name; });
    return true;
}

template <class T> const T* data_or_dummy(const std::vector<T>& v) { static const T dummy[1] = {}; return v.empty() ? dummy : v.data(); }

struct CStrings {                       // an array of C strings that may hold NULLs
    std::vector<std::string> store; std::vector<int> null; std::vector<const char*> ptr;
    void clear() { store.clear(); null.clear(); ptr.clear(); }
    void add(const std::string& s) { store.push_back(s); null.push_back(0); }
    void add_null() { store.emplace_back(); null.push_back(1); }
    const char* const* finish() { ptr.resize(std::max<size_t>(1, store.size()), nullptr); for (size_t i = 0; i < store.size(); ++i) ptr[i] = null[i] ? nullptr : store[i].c_str(); return ptr.data(); }
};

} // namespace

struct sp_database {
    Value root; std::string err;
    std::string md[5];
    std::vector<Region> hla_genes, collection;
    Value cyp_cfg_default, hla_cfg_default;
    const Value* cyp_cfg = nullptr; bool has_cyp_cfg = false, has_hla_cfg = false;
    std::vector<std::string> hla_keys;                       // hla_sequences in key order
    std::vector<std::string> cyp_keys, entry_keys;
    // last HLA flatten
    struct { std::vector<uint32_t> gene_of, exon_off; std::string dna, cdna, ref; std::vector<uint64_t> dna_off, cdna_off, ref_off; std::vector<uint8_t> fwd;
             std::vector<int32_t> es, ee; std::vector<std::string> id, gene, star; } hf;
    // last CYP flatten
    struct { std::string chrom; CStrings star, ref, alt, id, vi, tk, tv, ca, cb, sg; std::vector<uint32_t> off; std::vector<uint64_t> pos; } cf;
};

namespace {

int32_t db_fail(sp_database* db, const std::string& m) { db->err = m; return SP_ERR_INVALID_ARG; }

const Value* member(const Value& v, const char* k) { const Value* m = v.get(k); return (m && !m->is_null()) ? m : nullptr; }

std::vector<std::string> sorted_keys(const Value* obj) {
    std::vector<std::string> k;
    if (obj && obj->kind == Value::Object) for (const auto& kv : obj->obj) k.push_back(kv.first);
    std::sort(k.begin(), k.end());
    return k;
}

int32_t build(sp_database* db, std::string& err) {
    const Value& root = db->root;
    if (root.kind != Value::Object) { err = "the database is not a JSON object"; return SP_ERR_INVALID_ARG; }
    static const char* md_keys[5] = { "pbstarphase_version", "cpic_version", "hla_version", "pharmvar_version", "build_time" };
    const Value* md = member(root, "database_metadata");
    if (!md) { err = "missing field `database_metadata`"; return SP_ERR_INVALID_ARG; }
    for (int i = 0; i < 5; ++i) { const Value* v = md->get(md_keys[i]); if (!v) { err = std::string("missing field `") + md_keys[i] + "`"; return SP_ERR_INVALID_ARG; } db->md[i] = v->as_str(); }
    if (const Value* gc = member(root, "gene_collection")) if (!read_gene_dict(gc->get("gene_dict"), db->collection, err)) return SP_ERR_INVALID_ARG;
    // hla_config: the gene_collection form, the older three-map form, or the defaults
    { spj::Parser p(SP_DEFAULT_HLA_CONFIG, sizeof SP_DEFAULT_HLA_CONFIG - 1); p.parse(db->hla_cfg_default); }
    { spj::Parser p(SP_DEFAULT_CYP2D6_CONFIG, sizeof SP_DEFAULT_CYP2D6_CONFIG - 1); p.parse(db->cyp_cfg_default); }
    const Value* hc = member(root, "hla_config");
    db->has_hla_cfg = hc != nullptr;
    if (hc && member(*hc, "gene_collection")) {
        if (!read_gene_dict(member(*hc, "gene_collection")->get("gene_dict"), db->hla_genes, err)) return SP_ERR_INVALID_ARG;
    } else {
        const Value& legacy = (hc && hc->get("hla_coordinates")) ? *hc : db->hla_cfg_default;
        const Value* co = legacy.get("hla_coordinates"); const Value* ex = legacy.get("hla_exons"); const Value* st = legacy.get("hla_is_forward_strand");
        for (const std::string& g : sorted_keys(co)) {
            Region r; r.name = g;
            if (!read_coordinates(co->get(g.c_str()), r.chrom, r.start, r.end)) { err = "hla_config: bad coordinates for " + g; return SP_ERR_INVALID_ARG; }
            r.fwd = st && st->get(g.c_str()) ? st->get(g.c_str())->as_bool(true) : true;
            if (ex && ex->get(g.c_str())) for (const Value& e : ex->get(g.c_str())->arr) {
                std::string c; uint64_t s, t;
                if (!read_coordinates(&e, c, s, t)) { err = "hla_config: bad exon for " + g; return SP_ERR_INVALID_ARG; }
                r.es.push_back(s); r.ee.push_back(t);
            }
            db->hla_genes.push_back(std::move(r));
        }
    }
    for (const Region& r : db->hla_genes)          // HlaConfig::validate_config (src/hla/alleles.rs:84-102)
        if (r.es.empty()) { err = "Found 0 exons for \"" + r.name + "\", expected >0."; return SP_ERR_INVALID_ARG; }
    db->hla_keys = sorted_keys(member(root, "hla_sequences"));
    db->cyp_keys = sorted_keys(member(root, "cyp2d6_gene_def"));
    db->entry_keys = sorted_keys(member(root, "gene_entries"));
    db->cyp_cfg = member(root, "cyp2d6_config");
    db->has_cyp_cfg = db->cyp_cfg != nullptr;
    if (!db->cyp_cfg) db->cyp_cfg = &db->cyp_cfg_default;
    return SP_OK;
}

void put_err(char* err, uint32_t cap, const std::string& m) { if (err && cap) { const size_t k = std::min<size_t>(cap - 1, m.size()); std::memcpy(err, m.data(), k); err[k] = '\0'; } }

} // namespace

extern "C" {

int32_t sp_database_parse(const char* text, uint64_t len, sp_database** out, char* err, uint32_t err_cap) {
    if (!text || !out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    std::string plain, msg;
    if (len >= 2 && (unsigned char)text[0] == 0x1f && (unsigned char)text[1] == 0x8b) {
        if (!gunzip(std::string(text, (size_t)len), plain, msg)) { put_err(err, err_cap, msg); return SP_ERR_INVALID_ARG; }
        text = plain.data(); len = plain.size();
    }
    auto* db = new sp_database();
    spj::Parser p(text, (size_t)len);
    bool ok = p.parse(db->root);
    if (ok) { p.ws(); if (p.p != p.end) ok = p.fail("trailing characters"); }
    if (!ok) { put_err(err, err_cap, "JSON: " + p.err + " at byte " + std::to_string((size_t)(p.p - text))); delete db; return SP_ERR_INVALID_ARG; }
    const int32_t rc = build(db, msg);
    if (rc != SP_OK) { put_err(err, err_cap, msg); delete db; return rc; }
    *out = db;
    return SP_OK;
}

int32_t sp_database_load(const char* path, sp_database** out, char* err, uint32_t err_cap) {
    if (!path || !out) return SP_ERR_INVALID_ARG;
    std::string bytes, msg;
    if (!read_file(path, bytes, msg)) { put_err(err, err_cap, msg); return SP_ERR_INVALID_ARG; }
    return sp_database_parse(bytes.data(), bytes.size(), out, err, err_cap);
}

void sp_database_free(sp_database* db) { delete db; }
const char* sp_database_last_error(const sp_database* db) { return db ? db->err.c_str() : ""; }

int32_t sp_database_get_metadata(const sp_database* db, sp_database_metadata* out) {
    if (!db || !out) return SP_ERR_INVALID_ARG;
    out->pbstarphase_version = db->md[0].c_str(); out->cpic_version = db->md[1].c_str(); out->hla_version = db->md[2].c_str();
    out->pharmvar_version = db->md[3].c_str(); out->build_time = db->md[4].c_str();
    return SP_OK;
}

int32_t sp_database_info(const sp_database* db, sp_database_stats* out) {
    if (!db || !out) return SP_ERR_INVALID_ARG;
    *out = sp_database_stats{ (uint32_t)db->entry_keys.size(), (uint32_t)db->hla_keys.size(), (uint32_t)db->hla_genes.size(), (uint32_t)db->cyp_keys.size(),
                              (uint32_t)db->collection.size(), db->has_hla_cfg ? 1 : 0, db->has_cyp_cfg ? 1 : 0, 0 };
    return SP_OK;
}

int32_t sp_database_hla_gene(const sp_database* db, uint32_t g, sp_gene_region* out) {
    if (!db || !out || g >= db->hla_genes.size()) return SP_ERR_INVALID_ARG;
    const Region& r = db->hla_genes[g];
    *out = sp_gene_region{ r.name.c_str(), r.chrom.c_str(), r.start, r.end, r.fwd ? 1 : 0, r.absent ? 1 : 0, (uint32_t)r.es.size(), 0, data_or_dummy(r.es), data_or_dummy(r.ee) };
    return SP_OK;
}

int32_t sp_database_gene_entry(const sp_database* db, uint32_t i, const char** gene_name, const char** chromosome) {
    if (!db || i >= db->entry_keys.size()) return SP_ERR_INVALID_ARG;
    const Value* g = db->root.get("gene_entries")->get(db->entry_keys[i].c_str());
    if (gene_name) *gene_name = db->entry_keys[i].c_str();
    if (chromosome) *chromosome = g && g->get("chromosome") ? g->get("chromosome")->as_str().c_str() : "";
    return SP_OK;
}

int32_t sp_database_hla_flatten(sp_database* db, uint32_t n_genes, const char* const* gene_names, const char* const* gene_ref, int32_t ref_buffer,
                                sp_hla_db_desc* desc) {
    if (!db || !desc || !gene_ref || ref_buffer < 0) return SP_ERR_INVALID_ARG;
    std::vector<const Region*> genes;
    if (gene_names) {
        for (uint32_t g = 0; g < n_genes; ++g) {
            const Region* hit = nullptr;
            for (const Region& r : db->hla_genes) if (gene_names[g] && r.name == gene_names[g]) hit = &r;
            if (!hit) return db_fail(db, std::string("hla_config has no gene ") + (gene_names[g] ? gene_names[g] : "(null)"));
            genes.push_back(hit);
        }
    } else {
        if (n_genes != db->hla_genes.size()) return db_fail(db, "gene_names is NULL: n_genes must be the number of genes in hla_config");
        for (const Region& r : db->hla_genes) genes.push_back(&r);
    }
    auto& f = db->hf;
    f = {};
    f.ref_off.push_back(0); f.exon_off.push_back(0);
    for (size_t g = 0; g < genes.size(); ++g) {
        const Region& r = *genes[g];
        if (!gene_ref[g]) return db_fail(db, "gene_ref holds a NULL");
        const uint64_t want = r.end - r.start + 2ull * (uint64_t)ref_buffer;
        if (r.start < (uint64_t)ref_buffer) return db_fail(db, "the buffer leaves the chromosome at " + r.name);
        if (std::strlen(gene_ref[g]) != want) return db_fail(db, "gene_ref of " + r.name + " must hold " + std::to_string(want) + " bases");
        f.ref.append(gene_ref[g], want); f.ref_off.push_back(f.ref.size());
        f.fwd.push_back(r.fwd ? 1 : 0);
        const int64_t lo = (int64_t)r.start - ref_buffer;
        for (size_t e = 0; e < r.es.size(); ++e) { f.es.push_back((int32_t)((int64_t)r.es[e] - lo)); f.ee.push_back((int32_t)((int64_t)r.ee[e] - lo)); }
        f.exon_off.push_back((uint32_t)f.es.size());
    }
    const Value* seqs = db->root.get("hla_sequences");
    f.dna_off.push_back(0); f.cdna_off.push_back(0);
    for (const std::string& key : db->hla_keys) {
        const Value* a = seqs->get(key.c_str());
        const std::string& gname = a->get("gene_name") ? a->get("gene_name")->as_str() : std::string();
        int gi = -1;
        for (size_t g = 0; g < genes.size(); ++g) if (genes[g]->name == gname) gi = (int)g;
        if (gi < 0) continue;
        const Value* cd = a->get("cdna_sequence");
        if (!cd || cd->kind != Value::String) return db_fail(db, key + ": missing field `cdna_sequence`");
        f.gene_of.push_back((uint32_t)gi);
        const Value* dn = a->get("dna_sequence");
        if (dn && dn->kind == Value::String) f.dna += dn->s;
        f.dna_off.push_back(f.dna.size());
        f.cdna += cd->s; f.cdna_off.push_back(f.cdna.size());
        f.id.push_back(a->get("hla_id") ? a->get("hla_id")->as_str() : key);
        f.gene.push_back(gname);
        std::string star;
        if (const Value* sa = a->get("star_allele")) for (size_t k = 0; k < sa->arr.size(); ++k) { if (k) star += ':'; star += sa->arr[k].as_str(); }
        f.star.push_back(star);
    }
    *desc = sp_hla_db_desc{};
    desc->n_alleles = (uint32_t)f.gene_of.size(); desc->n_genes = (uint32_t)genes.size();
    desc->gene_of = data_or_dummy(f.gene_of);
    desc->dna = f.dna.c_str(); desc->dna_off = f.dna_off.data(); desc->cdna = f.cdna.c_str(); desc->cdna_off = f.cdna_off.data();
    desc->gene_ref = f.ref.c_str(); desc->gene_ref_off = f.ref_off.data(); desc->gene_fwd = data_or_dummy(f.fwd);
    desc->exon_off = f.exon_off.data(); desc->exon_start = data_or_dummy(f.es); desc->exon_end = data_or_dummy(f.ee);
    desc->ref_buffer = ref_buffer;
    return SP_OK;
}

int32_t sp_database_hla_allele(const sp_database* db, uint32_t i, const char** hla_id, const char** gene_name, const char** star_allele) {
    if (!db || i >= db->hf.id.size()) return SP_ERR_INVALID_ARG;
    if (hla_id) *hla_id = db->hf.id[i].c_str();
    if (gene_name) *gene_name = db->hf.gene[i].c_str();
    if (star_allele) *star_allele = db->hf.star[i].c_str();
    return SP_OK;
}

int32_t sp_database_cyp_window(const sp_database* db, const char** chrom, uint64_t* start, uint64_t* end) {
    if (!db || !start || !end) return SP_ERR_INVALID_ARG;
    uint64_t lo = UINT64_MAX, hi = 0; const std::string* c = nullptr;
    auto take = [&](const Value* v) { std::string ch; uint64_t s, e; if (read_coordinates(v, ch, s, e)) { lo = std::min(lo, s); hi = std::max(hi, e); if (!c && v->get("chrom")) c = &v->get("chrom")->s; } };
    const Value& cfg = *db->cyp_cfg;
    if (const Value* cc = cfg.get("cyp_coordinates")) for (const auto& kv : cc->obj) take(&kv.second);
    if (const Value* cr = cfg.get("cyp_regions")) for (const auto& g : cr->obj) for (const auto& kv : g.second.obj) take(&kv.second);
    take(cfg.get("cyp2d6_star5_del"));
    if (hi == 0) return SP_ERR_INVALID_ARG;
    if (chrom) *chrom = c ? c->c_str() : "chr22";
    *start = lo; *end = hi;
    return SP_OK;
}

int32_t sp_database_cyp_flatten(sp_database* db, const char* chrom_seq, uint64_t window_start, uint64_t window_len,
                                sp_cyp_locus* L, sp_cyp_gene_def* G, sp_cyp_config* K) {
    if (!db || !chrom_seq || !L || !G || !K) return SP_ERR_INVALID_ARG;
    const Value& cfg = *db->cyp_cfg;
    const Value* cc = cfg.get("cyp_coordinates"); const Value* reg = cfg.get("cyp_regions");
    if (!cc || !reg || !cfg.get("cyp2d6_star5_del")) return db_fail(db, "cyp2d6_config lacks cyp_coordinates / cyp_regions / cyp2d6_star5_del");
    auto& f = db->cf;
    f = {};
    *L = sp_cyp_locus{};
    std::string ch; bool ok = true;
    auto coord = [&](const char* key, uint64_t& s, uint64_t& e) { if (!read_coordinates(cc->get(key), ch, s, e)) { ok = false; db->err = std::string("cyp_coordinates lacks ") + key; } };
    coord("CYP2D6", L->d6_start, L->d6_end); f.chrom = ch;
    coord("CYP2D7", L->d7_start, L->d7_end); coord("REP6", L->rep6_start, L->rep6_end); coord("REP7", L->rep7_start, L->rep7_end);
    coord("spacer", L->spacer_start, L->spacer_end); coord("link_region", L->link_start, L->link_end); coord("CYP2D6_wfa_backbone", L->backbone_start, L->backbone_end);
    if (!ok) return SP_ERR_INVALID_ARG;
    read_coordinates(cfg.get("cyp2d6_star5_del"), ch, L->star5_start, L->star5_end);
    for (int x = 0; x < 9; ++x) {
        const std::string name = "exon" + std::to_string(x + 1);
        const Value* g6 = reg->get("CYP2D6"); const Value* g7 = reg->get("CYP2D7");
        if (!g6 || !g7 || !read_coordinates(g6->get(name.c_str()), ch, L->d6_exon_start[x], L->d6_exon_end[x]) ||
            !read_coordinates(g7->get(name.c_str()), ch, L->d7_exon_start[x], L->d7_exon_end[x])) return db_fail(db, "cyp_regions lacks " + name);
    }
    L->chrom_name = f.chrom.c_str(); L->chrom_seq = chrom_seq; L->window_start = window_start; L->window_len = window_len;
    const Value* defs = db->root.get("cyp2d6_gene_def");
    f.off.push_back(0);
    for (const std::string& key : db->cyp_keys) {
        const Value* a = defs->get(key.c_str());
        f.star.add(a->get("star_allele") ? a->get("star_allele")->as_str() : std::string());
        if (const Value* vs = a->get("variants")) for (const Value& v : vs->arr) {
            f.pos.push_back((uint64_t)(v.get("position") ? v.get("position")->as_int() : 0));
            f.ref.add(v.get("reference") ? v.get("reference")->as_str() : std::string());
            f.alt.add(v.get("alternate") ? v.get("alternate")->as_str() : std::string());
            const Value* id = v.get("id");
            if (id && id->kind == Value::String) f.id.add(id->s); else f.id.add_null();
            const Value* ex = v.get("extras"); const Value* vi = ex ? ex->get("VI") : nullptr;
            if (vi && vi->kind == Value::String) f.vi.add(vi->s); else f.vi.add_null();
        }
        f.off.push_back((uint32_t)f.pos.size());
    }
    *G = sp_cyp_gene_def{};
    G->n_alleles = (uint32_t)db->cyp_keys.size(); G->star_allele = f.star.finish(); G->var_off = f.off.data(); G->var_pos = data_or_dummy(f.pos);
    G->var_ref = f.ref.finish(); G->var_alt = f.alt.finish(); G->var_id = f.id.finish(); G->var_vi = f.vi.finish();
    // BTreeMap<String, String>, BTreeSet<(String, String)>, BTreeSet<String>
    std::vector<std::pair<std::string, std::string>> tr, con; std::vector<std::string> sg;
    if (const Value* t = cfg.get("cyp_translate")) for (const auto& kv : t->obj) tr.emplace_back(kv.first, kv.second.as_str());
    if (const Value* c = cfg.get("inferred_connections")) for (const Value& p : c->arr) if (p.arr.size() == 2) con.emplace_back(p.arr[0].as_str(), p.arr[1].as_str());
    if (const Value* s = cfg.get("unexpected_singletons")) for (const Value& x : s->arr) sg.push_back(x.as_str());
    std::sort(tr.begin(), tr.end()); std::sort(con.begin(), con.end()); con.erase(std::unique(con.begin(), con.end()), con.end());
    std::sort(sg.begin(), sg.end()); sg.erase(std::unique(sg.begin(), sg.end()), sg.end());
    for (const auto& p : tr) { f.tk.add(p.first); f.tv.add(p.second); }
    for (const auto& p : con) { f.ca.add(p.first); f.cb.add(p.second); }
    for (const auto& s : sg) f.sg.add(s);
    *K = sp_cyp_config{};
    K->n_translate = (uint32_t)tr.size(); K->translate_key = f.tk.finish(); K->translate_val = f.tv.finish();
    K->n_connections = (uint32_t)con.size(); K->connection_a = f.ca.finish(); K->connection_b = f.cb.finish();
    K->n_singletons = (uint32_t)sg.size(); K->singletons = f.sg.finish();
    return SP_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------ one variant-typed gene
namespace {

struct NormVar {                        // NormalizedVariant with its derived Ord: (chrom, position, reference, alternate, sv_stats)
    std::string chrom; uint64_t pos = 0; std::string ref, alt; bool sv = false; uint64_t sv_start = 0, sv_end = 0; std::string sv_label;
    auto key() const { return std::tie(chrom, pos, ref, alt, sv, sv_start, sv_end, sv_label); }     // None < Some; SvType is Deletion throughout
    bool operator<(const NormVar& o) const { return key() < o.key(); }
    bool operator==(const NormVar& o) const { return key() == o.key(); }
};
struct VarMeta { int64_t id = -1; std::string name, dbsnp; bool has_dbsnp = false, core = true; };
struct Hap { std::string name, core_allele; bool has_core_allele = false; std::vector<std::vector<int>> slots; };   // slot: indices into vars, -1 = None

} // namespace

struct sp_variant_gene {
    std::string err, chrom, gene_name;
    const char* seq = nullptr; uint64_t seq_len = 0;
    std::vector<NormVar> vars; std::vector<VarMeta> meta; std::vector<Hap> haps; uint32_t skipped = 0;
    // structural variants
    std::vector<Region> genes;          // the database's gene collection, name order
    std::vector<std::string> full_labels, partial_labels;
    std::vector<int64_t> g_start, g_end, e_start, e_end; std::vector<uint8_t> g_fwd, full_generic, partial_generic;
    std::vector<int32_t> e_off, full_off, full_gene, partial_off, partial_gene, partial_first, partial_end;
    bool sv_missing_gene = false; std::string sv_missing_name; bool sv_split_chrom = false;
    std::string sv_chrom; uint64_t sv_lo = 0, sv_hi = 0; bool has_sv = false;
    // last problem
    struct { std::vector<NormVar> vars; std::vector<int32_t> db_index; std::vector<std::string> labels; std::vector<uint8_t> hap_is_sv, hap_is_core, var_is_core;
             std::vector<int32_t> slot_off, alt_off, alt_var, obs_var, obs_gt, obs_sv; std::vector<int64_t> obs_ps; } pr;
};

namespace {

int32_t vg_fail(sp_variant_gene* g, const std::string& m) { g->err = m; return SP_ERR_INVALID_ARG; }

bool normalize(const sp_variant_gene* g, uint64_t pos, const std::string& ref, const std::string& alt, NormVar& out) {
    std::vector<char> r(ref.size() + alt.size() + 64 + (g->seq ? 4096 : 0)), a(r.size());
    uint64_t p = 0;
    // the left shift only ever rolls the alleles, it never lengthens them: the longer allele + 2 bounds both
    if (sp_variant_normalize(g->seq, g->seq_len, pos, ref.c_str(), alt.c_str(), &p, r.data(), a.data(), (uint32_t)r.size()) != SP_OK) return false;
    out = NormVar{}; out.chrom = g->chrom; out.pos = p; out.ref = r.data(); out.alt = a.data();
    return true;
}

} // namespace

extern "C" {

const char* sp_variant_gene_last_error(const sp_variant_gene* gene) { return gene ? gene->err.c_str() : ""; }
void sp_variant_gene_free(sp_variant_gene* gene) { delete gene; }

int32_t sp_variant_gene_create(sp_database* db, const char* gene_name, const char* chrom_seq, uint64_t chrom_len, sp_variant_gene** out) {
    if (!db || !gene_name || !out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    const Value* entries = db->root.get("gene_entries");
    const Value* ge = entries ? entries->get(gene_name) : nullptr;
    if (!ge) return db_fail(db, std::string("gene_entries has no gene ") + gene_name);
    auto g = std::make_unique<sp_variant_gene>();
    g->gene_name = gene_name; g->chrom = ge->get("chromosome") ? ge->get("chromosome")->as_str() : std::string();
    g->seq = chrom_seq; g->seq_len = chrom_len;
    const Value* variants = ge->get("variants"); const Value* haps = ge->get("defined_haplotypes");
    // load_database_haplotypes (src/diplotyper.rs:437-538)
    std::vector<std::pair<NormVar, VarMeta>> table;         // kept sorted by NormVar: BTreeMap<NormalizedVariant, VariantMeta>
    struct RawHap { std::string name, core; bool has_core; std::vector<std::vector<std::pair<bool, NormVar>>> slots; };
    std::vector<RawHap> raw;
    for (const std::string& hname : sorted_keys(haps)) {
        const Value* h = haps->get(hname.c_str());
        RawHap rh; rh.name = hname;
        const Value* ca = h->get("core_allele");
        rh.has_core = ca && ca->kind == Value::String; if (rh.has_core) rh.core = ca->s;
        std::vector<std::pair<uint64_t, const Value*>> alleles;        // BTreeMap<u64, String>
        if (const Value* hv = h->get("haplotype")) for (const auto& kv : hv->obj) alleles.emplace_back(std::strtoull(kv.first.c_str(), nullptr, 10), &kv.second);
        std::sort(alleles.begin(), alleles.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        std::vector<VarMeta> metas; bool normalized = true;
        for (const auto& [vid, allele] : alleles) {
            const Value* var = variants ? variants->get(std::to_string(vid).c_str()) : nullptr;
            if (!var) return db_fail(db, "variant " + std::to_string(vid) + " is referenced but not defined");
            const Value* al = var->get("alleles");
            if (!al || al->arr.size() < 2) return db_fail(db, "Encountered variant " + std::to_string(vid) + " with fewer than two alleles.");
            for (const Value& x : al->arr) if (x.kind != Value::String) return db_fail(db, "Encountered variant " + std::to_string(vid) + " with undefined alleles.");
            const std::string& ref = al->arr[0].s; const std::string& alt = allele->as_str();
            if (ref == alt) continue;
            // NormalizedVariant::multi_new: an IUPAC code or "A; B" lists alternatives, the reference allele among them is None
            const uint64_t pos0 = (uint64_t)(var->get("position") ? var->get("position")->as_int() : 0) - 1;
            uint32_t n_alt = 0; uint8_t none[16]; uint64_t apos[16];
            const uint32_t cap = (uint32_t)(ref.size() + alt.size() + 64 + (chrom_seq ? 4096 : 0));
            std::vector<char> rbuf((size_t)cap * 16), abuf((size_t)cap * 16);
            const int32_t rc = sp_variant_multi_normalize(chrom_seq, chrom_len, pos0, ref.c_str(), alt.c_str(), 16, &n_alt, none, apos, rbuf.data(), abuf.data(), cap);
            if (rc != SP_OK) { normalized = false; break; }
            std::vector<std::pair<bool, NormVar>> slot;
            for (uint32_t k = 0; k < n_alt; ++k) {
                NormVar nv; nv.chrom = g->chrom;
                if (!none[k]) { nv.pos = apos[k]; nv.ref = rbuf.data() + (size_t)k * cap; nv.alt = abuf.data() + (size_t)k * cap; }
                slot.emplace_back(!none[k], nv);
            }
            rh.slots.push_back(std::move(slot));
            VarMeta m; m.id = (int64_t)vid; m.name = var->get("name") ? var->get("name")->as_str() : std::string();
            const Value* rs = var->get("dbsnp_id"); m.has_dbsnp = rs && rs->kind == Value::String; if (m.has_dbsnp) m.dbsnp = rs->s;
            m.core = var->get("is_core_variant") ? var->get("is_core_variant")->as_bool(true) : true;
            metas.push_back(std::move(m));
        }
        if (!normalized) { ++g->skipped; continue; }
        for (size_t s = 0; s < rh.slots.size(); ++s) for (const auto& [some, nv] : rh.slots[s]) {
            if (!some) continue;
            auto at = std::lower_bound(table.begin(), table.end(), nv, [](const auto& e, const NormVar& v) { return e.first < v; });
            if (at == table.end() || !(at->first == nv)) table.insert(at, { nv, metas[s] });
        }
        raw.push_back(std::move(rh));
    }
    for (auto& e : table) { g->vars.push_back(e.first); g->meta.push_back(e.second); }
    for (const RawHap& rh : raw) {
        Hap h; h.name = rh.name; h.core_allele = rh.core; h.has_core_allele = rh.has_core;
        for (const auto& slot : rh.slots) {
            std::vector<int> ids;
            for (const auto& [some, nv] : slot) ids.push_back(some ? (int)(std::lower_bound(g->vars.begin(), g->vars.end(), nv) - g->vars.begin()) : -1);
            h.slots.push_back(std::move(ids));
        }
        g->haps.push_back(std::move(h));
    }
    // PgxStructuralVariants + the gene collection, flattened for is_deletion (src/diplotyper.rs:1020-1174)
    g->genes = db->collection;
    g->e_off.push_back(0);
    for (const Region& r : g->genes) {
        g->g_start.push_back((int64_t)r.start); g->g_end.push_back((int64_t)r.end); g->g_fwd.push_back(r.fwd ? 1 : 0);
        for (size_t e = 0; e < r.es.size(); ++e) { g->e_start.push_back((int64_t)r.es[e]); g->e_end.push_back((int64_t)r.ee[e]); }
        g->e_off.push_back((int32_t)g->e_start.size());
    }
    auto gene_id = [&](const std::string& name) { for (size_t i = 0; i < g->genes.size(); ++i) if (g->genes[i].name == name) return (int32_t)i; return (int32_t)-1; };
    g->full_off.push_back(0); g->partial_off.push_back(0);
    std::set<std::string> sv_genes;
    if (const Value* sv = member(*ge, "structural_variants")) {
        const Value* full = sv->get("full_gene_deletions"); const Value* part = sv->get("partial_gene_deletions");
        g->full_labels = sorted_keys(full); g->partial_labels = sorted_keys(part);
        for (const std::string& k : g->full_labels) {
            const Value* d = full->get(k.c_str());
            std::vector<std::string> names;
            if (const Value* l = d->get("full_genes_deleted")) for (const Value& x : l->arr) names.push_back(x.as_str());
            std::sort(names.begin(), names.end());
            for (const std::string& n : names) { g->full_gene.push_back(gene_id(n)); sv_genes.insert(n); }
            g->full_off.push_back((int32_t)g->full_gene.size());
            g->full_generic.push_back(d->get("is_generic") && d->get("is_generic")->as_bool() ? 1 : 0);
        }
        for (const std::string& k : g->partial_labels) {
            const Value* d = part->get(k.c_str());
            const Value* ex = d->get("exons_deleted");
            for (const std::string& n : sorted_keys(ex)) {
                const Value* range = ex->get(n.c_str());
                g->partial_gene.push_back(gene_id(n)); sv_genes.insert(n);
                g->partial_first.push_back((int32_t)(range->get("start") ? range->get("start")->as_int() : 0));
                g->partial_end.push_back((int32_t)(range->get("end") ? range->get("end")->as_int() : 0));
            }
            g->partial_off.push_back((int32_t)g->partial_gene.size());
            g->partial_generic.push_back(d->get("is_generic") && d->get("is_generic")->as_bool() ? 1 : 0);
        }
        g->has_sv = true;
        // the span load_sv_vcf_variants fetches (:760-790): all genes any definition names, on one chromosome
        bool first = true;
        for (const std::string& n : sv_genes) {
            const int32_t id = gene_id(n);
            if (id < 0) { if (!g->sv_missing_gene) { g->sv_missing_gene = true; g->sv_missing_name = n; } continue; }
            const Region& r = g->genes[(size_t)id];
            if (first) { g->sv_chrom = r.chrom; g->sv_lo = r.start; g->sv_hi = r.end; first = false; }
            else { if (r.chrom != g->sv_chrom) g->sv_split_chrom = true; g->sv_lo = std::min(g->sv_lo, r.start); g->sv_hi = std::max(g->sv_hi, r.end); }
        }
        if (first) g->has_sv = false;
    }
    *out = g.release();
    return SP_OK;
}

int32_t sp_variant_gene_info(const sp_variant_gene* g, sp_variant_gene_stats* out) {
    if (!g || !out) return SP_ERR_INVALID_ARG;
    *out = sp_variant_gene_stats{ (uint32_t)g->haps.size(), (uint32_t)g->vars.size(), g->skipped, (uint32_t)g->full_labels.size(), (uint32_t)g->partial_labels.size(), 0 };
    return SP_OK;
}

int32_t sp_variant_gene_haplotype(const sp_variant_gene* g, uint32_t h, const char** name, const char** core_allele) {
    if (!g || h >= g->haps.size()) return SP_ERR_INVALID_ARG;
    if (name) *name = g->haps[h].name.c_str();
    if (core_allele) *core_allele = g->haps[h].has_core_allele ? g->haps[h].core_allele.c_str() : nullptr;
    return SP_OK;
}

int32_t sp_variant_gene_variant(const sp_variant_gene* g, uint32_t v, uint64_t* position, const char** ref, const char** alt,
                                const char** name, const char** dbsnp_id, int64_t* variant_id, int32_t* is_core) {
    if (!g || v >= g->vars.size()) return SP_ERR_INVALID_ARG;
    if (position) *position = g->vars[v].pos;
    if (ref) *ref = g->vars[v].ref.c_str();
    if (alt) *alt = g->vars[v].alt.c_str();
    if (name) *name = g->meta[v].name.c_str();
    if (dbsnp_id) *dbsnp_id = g->meta[v].has_dbsnp ? g->meta[v].dbsnp.c_str() : nullptr;
    if (variant_id) *variant_id = g->meta[v].id;
    if (is_core) *is_core = g->meta[v].core ? 1 : 0;
    return SP_OK;
}

int32_t sp_variant_gene_sv_definitions(const sp_variant_gene* g, sp_sv_definitions* d) {
    if (!g || !d) return SP_ERR_INVALID_ARG;
    *d = sp_sv_definitions{};
    d->n_genes = (int32_t)g->genes.size(); d->gene_start = data_or_dummy(g->g_start); d->gene_end = data_or_dummy(g->g_end); d->gene_forward = data_or_dummy(g->g_fwd);
    d->exon_off = g->e_off.data(); d->exon_start = data_or_dummy(g->e_start); d->exon_end = data_or_dummy(g->e_end);
    d->n_full = (int32_t)g->full_labels.size(); d->full_generic = data_or_dummy(g->full_generic); d->full_off = g->full_off.data(); d->full_gene = data_or_dummy(g->full_gene);
    d->n_partial = (int32_t)g->partial_labels.size(); d->partial_generic = data_or_dummy(g->partial_generic); d->partial_off = g->partial_off.data();
    d->partial_gene = data_or_dummy(g->partial_gene); d->partial_first = data_or_dummy(g->partial_first); d->partial_end = data_or_dummy(g->partial_end);
    return SP_OK;
}

int32_t sp_variant_gene_sv_label(const sp_variant_gene* g, int32_t kind, int32_t index, const char** label) {
    if (!g || !label) return SP_ERR_INVALID_ARG;
    const auto& l = kind == 1 ? g->full_labels : g->partial_labels;
    if ((kind != 1 && kind != 2) || index < 0 || (size_t)index >= l.size()) return SP_ERR_INVALID_ARG;
    *label = l[(size_t)index].c_str();
    return SP_OK;
}

int32_t sp_variant_gene_problem(sp_variant_gene* g, uint32_t n_alleles, const sp_vcf_allele* alleles, uint32_t n_deletions,
                                const sp_vcf_deletion* deletions, uint64_t max_sv_length, sp_variant_problem* problem) {
    if (!g || !problem || (n_alleles && !alleles) || (n_deletions && !deletions)) return SP_ERR_INVALID_ARG;
    if (max_sv_length == 0) max_sv_length = 1000000;
    struct Obs { NormVar v; int32_t gt; int64_t ps; };
    std::vector<Obs> obs;
    // load_vcf_variants (src/diplotyper.rs:551-737): every database variant is looked up among the records within +-50 bp
    std::vector<std::pair<bool, NormVar>> norm(n_alleles);
    for (uint32_t i = 0; i < n_alleles; ++i) norm[i].first = alleles[i].ref && alleles[i].alt && normalize(g, alleles[i].position, alleles[i].ref, alleles[i].alt, norm[i].second);
    for (const NormVar& v : g->vars) {
        const uint64_t lo = v.pos > 50 ? v.pos - 50 : 0, hi = v.pos + 50;
        bool found = false; int32_t gt = 0; int64_t ps = -1;
        for (uint32_t i = 0; i < n_alleles; ++i) {
            if (!norm[i].first || !(norm[i].second == v)) continue;
            const uint64_t p0 = alleles[i].position, p1 = p0 + std::strlen(alleles[i].ref);
            if (!(p0 < hi && p1 > lo)) continue;
            switch (alleles[i].gt) {
                case SP_GT_HOM_ALT:
                    if (alleles[i].ps >= 0) return vg_fail(g, "Homozygous record detected with a phase set ID (PS)");
                    found = true; gt = SP_GT_HOM_ALT; ps = -1; break;
                case SP_GT_HET_PHASED: case SP_GT_HET_FLIP:
                    found = true;
                    if (alleles[i].ps >= 0) { gt = alleles[i].gt; ps = alleles[i].ps; } else { gt = SP_GT_HET_UNPHASED; ps = -1; }   // phased without PS: unphased
                    break;
                case SP_GT_HET_UNPHASED: found = true; gt = SP_GT_HET_UNPHASED; ps = -1; break;
                default: break;
            }
        }
        if (found) obs.push_back(Obs{ v, gt, ps });
    }
    // load_sv_vcf_variants (:739-857)
    if (g->has_sv && n_deletions) {
        if (g->sv_missing_gene) return vg_fail(g, "Missing gene definition (" + g->sv_missing_name + ") for structural variant");
        if (g->sv_split_chrom) return vg_fail(g, "Structural variant gene set is not all on the same chromosome");
        sp_sv_definitions defs; sp_variant_gene_sv_definitions(g, &defs);
        std::vector<NormVar> seen;
        for (uint32_t i = 0; i < n_deletions; ++i) {
            const sp_vcf_deletion& d = deletions[i];
            if (!(d.start < g->sv_hi && d.end > g->sv_lo)) continue;
            if (d.end < d.start || d.end - d.start > max_sv_length) continue;
            int32_t kind = 0, index = -1;
            if (sp_variant_is_deletion(&defs, d.start, d.end, &kind, &index) != SP_OK) return vg_fail(g, "Gene collection does not contain a definition for a deletable gene");
            if (kind == 0) continue;
            int32_t gt = d.gt; int64_t ps = d.ps;
            if (gt == SP_GT_HOM_REF) continue;
            if ((gt == SP_GT_HET_PHASED || gt == SP_GT_HET_FLIP) && ps < 0) gt = SP_GT_HET_UNPHASED;
            if (gt == SP_GT_HET_UNPHASED || gt == SP_GT_HOM_ALT) ps = -1;
            NormVar v; v.chrom = g->sv_chrom; v.pos = d.start; v.sv = true; v.sv_start = d.start; v.sv_end = d.end;
            v.sv_label = (kind == 1 ? g->full_labels : g->partial_labels)[(size_t)index];
            for (const NormVar& s : seen) if (s == v) return vg_fail(g, "Detected duplicate entry for normalized SV");
            seen.push_back(v);
            obs.push_back(Obs{ v, gt, ps });
        }
    }
    // the integer problem: variants = the gene's and the observed deletions, in NormalizedVariant order
    auto& P = g->pr;
    P = {};
    P.vars = g->vars;
    for (const Obs& o : obs) if (o.v.sv) P.vars.push_back(o.v);
    std::sort(P.vars.begin(), P.vars.end());
    P.vars.erase(std::unique(P.vars.begin(), P.vars.end()), P.vars.end());
    auto id_of = [&](const NormVar& v) { return (int32_t)(std::lower_bound(P.vars.begin(), P.vars.end(), v) - P.vars.begin()); };
    std::set<std::string> labels;
    for (const Obs& o : obs) if (o.v.sv) labels.insert(o.v.sv_label);
    P.labels.assign(labels.begin(), labels.end());
    for (const NormVar& v : P.vars) {
        if (v.sv) { P.db_index.push_back(-1); P.var_is_core.push_back(1); }
        else { const int32_t k = (int32_t)(std::lower_bound(g->vars.begin(), g->vars.end(), v) - g->vars.begin()); P.db_index.push_back(k); P.var_is_core.push_back(g->meta[(size_t)k].core ? 1 : 0); }
    }
    P.slot_off.push_back(0); P.alt_off.push_back(0);
    for (const Hap& h : g->haps) {
        for (const auto& slot : h.slots) {
            for (int k : slot) P.alt_var.push_back(k < 0 ? -1 : id_of(g->vars[(size_t)k]));
            P.alt_off.push_back((int32_t)P.alt_var.size());
        }
        P.slot_off.push_back((int32_t)P.alt_off.size() - 1);
        P.hap_is_sv.push_back(0); P.hap_is_core.push_back(h.has_core_allele ? 0 : 1);
    }
    std::sort(obs.begin(), obs.end(), [](const Obs& a, const Obs& b) { return a.v < b.v; });
    for (const Obs& o : obs) {
        P.obs_var.push_back(id_of(o.v)); P.obs_gt.push_back(o.gt); P.obs_ps.push_back(o.ps);
        P.obs_sv.push_back(o.v.sv ? (int32_t)(std::lower_bound(P.labels.begin(), P.labels.end(), o.v.sv_label) - P.labels.begin()) : -1);
    }
    *problem = sp_variant_problem{};
    problem->n_haps = (int32_t)g->haps.size(); problem->hap_is_sv = data_or_dummy(P.hap_is_sv); problem->hap_is_core = data_or_dummy(P.hap_is_core);
    problem->slot_off = P.slot_off.data(); problem->alt_off = P.alt_off.data(); problem->alt_var = data_or_dummy(P.alt_var);
    problem->n_vars = (int32_t)P.vars.size(); problem->var_is_core = data_or_dummy(P.var_is_core);
    problem->n_obs = (int32_t)obs.size(); problem->obs_var = data_or_dummy(P.obs_var); problem->obs_gt = data_or_dummy(P.obs_gt);
    problem->obs_ps = data_or_dummy(P.obs_ps); problem->obs_sv_label = data_or_dummy(P.obs_sv);
    return SP_OK;
}

int32_t sp_variant_gene_problem_variant(const sp_variant_gene* g, int32_t id, int32_t* db_variant, const char** sv_label, uint64_t* sv_start, uint64_t* sv_end) {
    if (!g || id < 0 || (size_t)id >= g->pr.vars.size()) return SP_ERR_INVALID_ARG;
    const NormVar& v = g->pr.vars[(size_t)id];
    if (db_variant) *db_variant = g->pr.db_index[(size_t)id];
    if (sv_label) *sv_label = v.sv ? v.sv_label.c_str() : nullptr;
    if (sv_start) *sv_start = v.sv_start;
    if (sv_end) *sv_end = v.sv_end;
    return SP_OK;
}

int32_t sp_variant_gene_problem_sv_label(const sp_variant_gene* g, int32_t label_id, const char** label) {
    if (!g || !label || label_id < 0 || (size_t)label_id >= g->pr.labels.size()) return SP_ERR_INVALID_ARG;
    *label = g->pr.labels[(size_t)label_id].c_str();
    return SP_OK;
}

} // extern "C"
