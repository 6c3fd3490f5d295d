// sp_result.hip -- the result file (host only): StarphaseJson / PgxGeneDetails (src/data_types/starphase_json.rs:11-326) and the
// types they hold -- Diplotype, InexactDiplotype, InexactHaplotype (src/data_types/pgx_diplotype.rs), RegionVariant
// (src/data_types/region_variants.rs), NormalizedVariant / NormalizedGenotype / StructuralVariantStats
// (src/data_types/normalized_variant.rs:16-28,290-300,334-361), HlaMappingStats / MappingStats (src/hla/mapping.rs:8-14,
// src/data_types/mapping.rs:6-18) -- written as serde_json::to_writer_pretty writes them (save_json, src/util/file_io.rs:37-52):
// struct fields in declaration order, maps in key order, Option::None as null, unit enum variants as their names (Genotype with its
// serde renames), std::ops::Range as {"start", "end"}.
#include "sp_internal.h"
#include "sp_json.h"
#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <map>

using spj::Value;

struct sp_gene_details {
    Value diplotypes = spj::array(), simple = spj::array(), inexact = spj::array(), variants = spj::array(), mappings = spj::array(), multi = spj::array();
    bool has_simple = false;
};

struct sp_result {
    std::string version, err, text, tsv;
    std::string md[5];
    std::map<std::string, Value> genes;           // BTreeMap<String, PgxGeneDetails>
};

namespace {

Value diplotype(const char* h1, const char* h2) {
    const std::string a(h1 ? h1 : ""), b(h2 ? h2 : "");
    Value d = spj::object();
    d.obj.emplace_back("hap1", spj::str(a)); d.obj.emplace_back("hap2", spj::str(b)); d.obj.emplace_back("diplotype", spj::str(a + "/" + b));
    return d;
}

const char* relationship_name(int32_t s) {
    static const char* names[] = { "Unknown", "Match", "Unexpected", "Missing", "AmbiguousUnexpected", "AmbiguousMissing", "UnknownUnexpected", "UnknownMissing" };
    return (s >= 0 && s < 8) ? names[s] : "Unknown";
}

// InexactHaplotype::new: the BTreeSet<RegionVariant> (label, is_vi, variant_state) and the match type; returns full_haplotype()
std::string inexact_haplotype(const char* base, uint32_t n, const char* const* labels, const uint8_t* is_vi, const int32_t* states, Value& out) {
    struct Rv { std::string label; int vi, state; };
    std::vector<Rv> set;
    for (uint32_t i = 0; i < n; ++i) set.push_back(Rv{ labels[i] ? labels[i] : "", is_vi[i] ? 1 : 0, states[i] });
    std::sort(set.begin(), set.end(), [](const Rv& a, const Rv& b) { if (a.label != b.label) return a.label < b.label; if (a.vi != b.vi) return a.vi < b.vi; return a.state < b.state; });
    set.erase(std::unique(set.begin(), set.end(), [](const Rv& a, const Rv& b) { return a.label == b.label && a.vi == b.vi && a.state == b.state; }), set.end());
    int32_t match_type = 0;
    std::vector<char> full(64 + std::strlen(base ? base : ""));
    for (const Rv& v : set) full.resize(full.size() + v.label.size() + 4);
    sp_inexact_haplotype(base, n, labels, is_vi, states, &match_type, full.data(), (uint32_t)full.size());
    static const char* match_names[] = { "Unknown", "NoMatch", "CoreMatch", "SubAlleleMatch" };
    out = spj::object();
    out.obj.emplace_back("base_haplotype", spj::str(base ? base : ""));
    out.obj.emplace_back("match_type", spj::str(match_names[match_type]));
    Value rel = spj::array();
    for (const Rv& v : set) {
        Value r = spj::object();
        r.obj.emplace_back("label", spj::str(v.label)); r.obj.emplace_back("is_vi", spj::boolean(v.vi != 0)); r.obj.emplace_back("variant_state", spj::str(relationship_name(v.state)));
        rel.arr.push_back(std::move(r));
    }
    out.obj.emplace_back("variant_relationships", std::move(rel));
    return std::string(full.data());
}

Value option_u64(bool some, uint64_t v) { return some ? spj::num((int64_t)v) : Value(); }

Value mapping_stats(const sp_mapping_stats* s) {
    if (!s || !s->present) return Value();
    Value m = spj::object();
    m.obj.emplace_back("seq_len", spj::num((int64_t)s->seq_len)); m.obj.emplace_back("nm", spj::num((int64_t)s->nm)); m.obj.emplace_back("unmapped", spj::num((int64_t)s->unmapped));
    m.obj.emplace_back("clipped_start", option_u64(s->has_clips != 0, s->clipped_start)); m.obj.emplace_back("clipped_end", option_u64(s->has_clips != 0, s->clipped_end));
    return m;
}

int32_t fail(sp_result* r, const std::string& m) { r->err = m; return SP_ERR_INVALID_ARG; }

} // namespace

extern "C" {

int32_t sp_result_create(const sp_database* db, const char* pbstarphase_version, sp_result** out) {
    if (!out) return SP_ERR_INVALID_ARG;
    auto* r = new sp_result();
    r->version = pbstarphase_version ? pbstarphase_version : "";
    if (db) {
        sp_database_metadata md;
        sp_database_get_metadata(db, &md);
        r->md[0] = md.pbstarphase_version; r->md[1] = md.cpic_version; r->md[2] = md.hla_version; r->md[3] = md.pharmvar_version; r->md[4] = md.build_time;
    } else r->md[4] = "1970-01-01T00:00:00Z";        // chrono::DateTime<Utc>::default()
    *out = r;
    return SP_OK;
}
void sp_result_free(sp_result* result) { delete result; }
const char* sp_result_last_error(const sp_result* result) { return result ? result->err.c_str() : ""; }

int32_t sp_gene_details_create(sp_gene_details** out) { if (!out) return SP_ERR_INVALID_ARG; *out = new sp_gene_details(); return SP_OK; }
void sp_gene_details_free(sp_gene_details* details) { delete details; }

int32_t sp_gene_details_add_diplotype(sp_gene_details* d, const char* hap1, const char* hap2) {
    if (!d) return SP_ERR_INVALID_ARG;
    d->diplotypes.arr.push_back(diplotype(hap1, hap2));
    return SP_OK;
}
int32_t sp_gene_details_add_simple_diplotype(sp_gene_details* d, const char* hap1, const char* hap2) {
    if (!d) return SP_ERR_INVALID_ARG;
    d->simple.arr.push_back(diplotype(hap1, hap2)); d->has_simple = true;
    return SP_OK;
}
int32_t sp_gene_details_set_simple_diplotypes(sp_gene_details* d, int32_t some) {
    if (!d) return SP_ERR_INVALID_ARG;
    d->has_simple = some != 0; if (!some) d->simple.arr.clear();
    return SP_OK;
}
int32_t sp_gene_details_add_inexact_diplotype(sp_gene_details* d,
                                              const char* base1, uint32_t n1, const char* const* labels1, const uint8_t* is_vi1, const int32_t* states1,
                                              const char* base2, uint32_t n2, const char* const* labels2, const uint8_t* is_vi2, const int32_t* states2) {
    if (!d || (n1 && (!labels1 || !is_vi1 || !states1)) || (n2 && (!labels2 || !is_vi2 || !states2))) return SP_ERR_INVALID_ARG;
    Value h1, h2;
    const std::string f1 = inexact_haplotype(base1, n1, labels1, is_vi1, states1, h1), f2 = inexact_haplotype(base2, n2, labels2, is_vi2, states2, h2);
    Value x = spj::object();
    x.obj.emplace_back("basic_diplotype", diplotype(f1.c_str(), f2.c_str()));
    x.obj.emplace_back("haplotype_1", std::move(h1)); x.obj.emplace_back("haplotype_2", std::move(h2));
    d->inexact.arr.push_back(std::move(x));
    return SP_OK;
}
int32_t sp_gene_details_add_diplotype_only(sp_gene_details* d, const char* hap1, const char* hap2) {
    if (!d) return SP_ERR_INVALID_ARG;
    Value x = spj::object();
    x.obj.emplace_back("basic_diplotype", diplotype(hap1, hap2)); x.obj.emplace_back("haplotype_1", Value()); x.obj.emplace_back("haplotype_2", Value());
    d->inexact.arr.push_back(std::move(x));
    return SP_OK;
}
int32_t sp_gene_details_add_variant(sp_gene_details* d, const sp_variant_detail* v) {
    if (!d || !v || v->genotype < SP_GT_HOM_REF || v->genotype > SP_GT_HOM_ALT) return SP_ERR_INVALID_ARG;
    static const char* gt_names[] = { "0/0", "0/1", "0|1", "1|0", "1/1" };
    Value nv = spj::object();
    nv.obj.emplace_back("chrom", spj::str(v->chrom ? v->chrom : "")); nv.obj.emplace_back("position", spj::num((int64_t)v->position));
    nv.obj.emplace_back("reference", spj::str(v->reference ? v->reference : "")); nv.obj.emplace_back("alternate", spj::str(v->alternate ? v->alternate : ""));
    if (v->sv_label) {
        Value sv = spj::object();
        sv.obj.emplace_back("sv_type", spj::str("Deletion")); sv.obj.emplace_back("start", spj::num((int64_t)v->sv_start)); sv.obj.emplace_back("end", spj::num((int64_t)v->sv_end));
        sv.obj.emplace_back("haplotype_label", spj::str(v->sv_label));
        nv.obj.emplace_back("sv_stats", std::move(sv));
    } else nv.obj.emplace_back("sv_stats", Value());
    Value g = spj::object();
    g.obj.emplace_back("genotype", spj::str(gt_names[v->genotype])); g.obj.emplace_back("phase_set", option_u64(v->phase_set >= 0, (uint64_t)v->phase_set));
    Value x = spj::object();
    x.obj.emplace_back("variant_id", spj::num((int64_t)v->variant_id)); x.obj.emplace_back("variant_name", spj::str(v->variant_name ? v->variant_name : ""));
    x.obj.emplace_back("dbsnp", v->dbsnp ? spj::str(v->dbsnp) : Value());
    x.obj.emplace_back("normalized_variant", std::move(nv)); x.obj.emplace_back("normalized_genotype", std::move(g));
    x.obj.emplace_back("is_core_variant", spj::boolean(v->is_core_variant != 0));
    d->variants.arr.push_back(std::move(x));
    return SP_OK;
}
int32_t sp_gene_details_add_mapping(sp_gene_details* d, const char* read_qname, const char* best_hla_id, const char* best_star_allele,
                                    const sp_mapping_stats* cdna, const sp_mapping_stats* dna, int32_t is_ignored) {
    if (!d) return SP_ERR_INVALID_ARG;
    Value st = spj::object();
    st.obj.emplace_back("cdna_stats", mapping_stats(cdna)); st.obj.emplace_back("dna_stats", mapping_stats(dna));
    Value x = spj::object();
    x.obj.emplace_back("read_qname", spj::str(read_qname ? read_qname : "")); x.obj.emplace_back("best_hla_id", spj::str(best_hla_id ? best_hla_id : ""));
    x.obj.emplace_back("best_star_allele", spj::str(best_star_allele ? best_star_allele : "")); x.obj.emplace_back("best_mapping_stats", std::move(st));
    x.obj.emplace_back("is_ignored", spj::boolean(is_ignored != 0));
    d->mappings.arr.push_back(std::move(x));
    return SP_OK;
}
int32_t sp_gene_details_add_multi_mapping(sp_gene_details* d, const char* read_qname, uint64_t read_start, uint64_t read_end,
                                          uint64_t consensus_id, const char* consensus_star_allele) {
    if (!d) return SP_ERR_INVALID_ARG;
    Value range = spj::object();
    range.obj.emplace_back("start", spj::num((int64_t)read_start)); range.obj.emplace_back("end", spj::num((int64_t)read_end));
    Value x = spj::object();
    x.obj.emplace_back("read_qname", spj::str(read_qname ? read_qname : "")); x.obj.emplace_back("read_position", std::move(range));
    x.obj.emplace_back("consensus_id", spj::num((int64_t)consensus_id)); x.obj.emplace_back("consensus_star_allele", spj::str(consensus_star_allele ? consensus_star_allele : ""));
    d->multi.arr.push_back(std::move(x));
    return SP_OK;
}

int32_t sp_result_insert(sp_result* r, const char* gene, const sp_gene_details* d, int32_t constructor) {
    if (!r || !gene || (!d && constructor != SP_DETAILS_NO_MATCH)) return SP_ERR_INVALID_ARG;
    static const sp_gene_details empty;
    if (!d) d = &empty;
    Value none;
    Value no_match = spj::array(); no_match.arr.push_back(diplotype("NO_MATCH", "NO_MATCH"));
    Value dip = d->diplotypes, simple = d->has_simple ? d->simple : none, inexact = none, variants = none, mappings = none, multi = none;
    auto simple_length = [&]() { return !d->has_simple || d->simple.arr.size() == d->diplotypes.arr.size(); };
    switch (constructor) {
        case SP_DETAILS_SUBALLELE_MATCH:           // new_suballele_match (:78-92)
            if (!simple_length()) return fail(r, "diplotypes and simple_diplotypes must be the same length");
            variants = d->variants; break;
        case SP_DETAILS_CORE_MATCH:                // new_core_match (:101-119)
            if (!simple_length()) return fail(r, "diplotypes and simple_diplotypes must be the same length");
            if (d->inexact.arr.size() != d->diplotypes.arr.size()) return fail(r, "diplotypes and inexact_diplotypes must be the same length");
            inexact = d->inexact; variants = d->variants; break;
        case SP_DETAILS_INEXACT_DIPLOTYPES:        // new_inexact_diplotypes (:125-140)
            dip = no_match; simple = none; inexact = d->inexact; variants = d->variants; break;
        case SP_DETAILS_FROM_MAPPINGS:             // new_from_mappings (:147-161)
            if (!simple_length()) return fail(r, "diplotypes and simple_diplotypes must be the same length");
            mappings = d->mappings; break;
        case SP_DETAILS_FROM_MULTI_MAPPINGS:       // new_from_multi_mappings (:169-188): inexact_diplotypes is an Option the caller fills or not
            if (!simple_length()) return fail(r, "diplotypes and simple_diplotypes must be the same length");
            if (!d->inexact.arr.empty()) inexact = d->inexact;
            multi = d->multi; break;
        case SP_DETAILS_NO_MATCH:                  // no_match (:192-204)
            dip = no_match; simple = none; break;
        default: return fail(r, "unknown constructor");
    }
    if (r->genes.count(gene)) return fail(r, std::string("Entry for ") + gene + " is already occupied.");
    Value g = spj::object();
    g.obj.emplace_back("diplotypes", std::move(dip)); g.obj.emplace_back("simple_diplotypes", std::move(simple)); g.obj.emplace_back("inexact_diplotypes", std::move(inexact));
    g.obj.emplace_back("variant_details", std::move(variants)); g.obj.emplace_back("mapping_details", std::move(mappings)); g.obj.emplace_back("multi_mapping_details", std::move(multi));
    r->genes.emplace(gene, std::move(g));
    return SP_OK;
}

int32_t sp_result_json(sp_result* r, const char** text, uint64_t* len) {
    if (!r || !text) return SP_ERR_INVALID_ARG;
    Value root = spj::object();
    root.obj.emplace_back("pbstarphase_version", spj::str(r->version));
    Value md = spj::object();
    static const char* md_keys[5] = { "pbstarphase_version", "cpic_version", "hla_version", "pharmvar_version", "build_time" };
    for (int i = 0; i < 5; ++i) md.obj.emplace_back(md_keys[i], spj::str(r->md[i]));
    root.obj.emplace_back("database_metadata", std::move(md));
    Value genes = spj::object();
    for (const auto& kv : r->genes) genes.obj.emplace_back(kv.first, kv.second);
    root.obj.emplace_back("gene_details", std::move(genes));
    r->text.clear();
    spj::write_pretty(r->text, root);
    *text = r->text.c_str();
    if (len) *len = r->text.size();
    return SP_OK;
}

int32_t sp_result_pharmcat_tsv(sp_result* r, const char** text, uint64_t* len) {
    if (!r || !text) return SP_ERR_INVALID_ARG;
    // the csv crate quotes a field that holds the delimiter, a quote or a line break
    auto field = [](const std::string& f) {
        if (f.find_first_of("\t\"\r\n") == std::string::npos) return f;
        std::string q = "\"";
        for (char c : f) { if (c == '"') q += '"'; q += c; }
        return q + "\"";
    };
    std::string out = "#gene\tdiplotype\n";
    for (const auto& kv : r->genes) {
        const Value* simple = kv.second.get("simple_diplotypes");
        const Value* list = (simple && simple->kind == Value::Array) ? simple : kv.second.get("diplotypes");
        if (!list || list->arr.empty()) return fail(r, "gene " + kv.first + " has no diplotype to report");
        // dedup_simple_diplotypes: a BTreeSet over Diplotype, whose order compares the two haplotypes as a sorted pair; the first of equal ones stays
        struct D { std::string h1, h2; std::pair<std::string, std::string> key; };
        std::vector<D> set;
        for (const Value& d : list->arr) {
            D x; x.h1 = d.get("hap1") ? d.get("hap1")->as_str() : std::string(); x.h2 = d.get("hap2") ? d.get("hap2")->as_str() : std::string();
            x.key = x.h1 < x.h2 ? std::make_pair(x.h1, x.h2) : std::make_pair(x.h2, x.h1);
            bool seen = false;
            for (const D& y : set) if (y.key == x.key) seen = true;
            if (!seen) set.push_back(std::move(x));
        }
        std::sort(set.begin(), set.end(), [](const D& a, const D& b) { return a.key < b.key; });
        std::string h1 = set[0].h1, h2 = set[0].h2;
        if (set.size() > 1) h1 = h2 = "Multiple";
        std::string dip;
        if (kv.first == "MT-RNR1") dip = h1 == h2 ? h1 : "Unknown";
        else {
            char buf[4]; const uint32_t need = sp_diplotype_string(h1.c_str(), h2.c_str(), 1, buf, 0);
            std::vector<char> full(need + 1);
            sp_diplotype_string(h1.c_str(), h2.c_str(), 1, full.data(), need + 1);
            dip = full.data();
        }
        out += field(kv.first) + "\t" + field(dip) + "\n";
    }
    r->tsv.swap(out);
    *text = r->tsv.c_str();
    if (len) *len = r->tsv.size();
    return SP_OK;
}

int32_t sp_result_save_pharmcat_tsv(sp_result* r, const char* path) {
    if (!r || !path) return SP_ERR_INVALID_ARG;
    const char* text; uint64_t len;
    const int32_t rc = sp_result_pharmcat_tsv(r, &text, &len);
    if (rc != SP_OK) return rc;
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(r, std::string("cannot create ") + path);
    const bool ok = std::fwrite(text, 1, (size_t)len, f) == (size_t)len;
    if (std::fclose(f) != 0 || !ok) return fail(r, std::string("cannot write ") + path);
    return SP_OK;
}

int32_t sp_result_save(sp_result* r, const char* path) {
    if (!r || !path) return SP_ERR_INVALID_ARG;
    const char* text; uint64_t len;
    sp_result_json(r, &text, &len);
    const std::string p(path);
    if (p.size() >= 3 && p.compare(p.size() - 3, 3, ".gz") == 0) {
        gzFile f = gzopen(path, "wb9");
        if (!f) return fail(r, "cannot create " + p);
        const bool ok = gzwrite(f, text, (unsigned)len) == (int)len;
        if (gzclose(f) != Z_OK || !ok) return fail(r, "cannot write " + p);
        return SP_OK;
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(r, "cannot create " + p);
    const bool ok = std::fwrite(text, 1, (size_t)len, f) == (size_t)len;
    if (std::fclose(f) != 0 || !ok) return fail(r, "cannot write " + p);
    return SP_OK;
}

} // extern "C"
