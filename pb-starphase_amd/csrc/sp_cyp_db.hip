// sp_cyp_db.hip -- the CYP2D6 search templates and typing tables, host side of the library.
//
// Replaces, behind the C ABI:
//   generate_cyp_hybrids                     src/cyp2d6/definitions.rs:346-464   (the 39 template sequences)
//   LoadedVariants::load_variant_database    src/cyp2d6/haplotyper.rs:650-773    (ordered variant table, labels, VI flags)
//   Cyp2d6Extractor::new                     src/cyp2d6/haplotyper.rs:45-132     (haplotype_lookup 0/1 vectors, mapped_hybrids)
// No device arithmetic: the tables are built once per database; the templates are then packed into HBM as an sp_seqset
// (k-mer indexed on first use by K3) and everything sp_cyp_diplotype needs is handed out as an sp_cyp_problem.
#include "sp_internal.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

struct sp_cyp_db {
    sp_ctx* ctx = nullptr;
    // templates in find_base_type_in_sequence's visiting order: sorted by full_allele() (haplotyper.rs:175-183)
    std::vector<int32_t> t_type; std::vector<std::string> t_sub, t_full, t_seq; std::vector<uint8_t> t_has_sub, t_deep;
    sp_seqset* templates = nullptr;
    // backbone + ordered variants (positions: chromosome and backbone-relative)
    std::string backbone; uint64_t backbone_start = 0;
    std::vector<int64_t> v_pos; std::vector<int32_t> v_rel; std::vector<std::string> v_ref, v_alt, v_label; std::vector<uint8_t> v_vi;
    std::map<std::tuple<uint64_t, std::string, std::string>, uint32_t> v_lookup;
    std::map<std::string, uint32_t> label_lookup;
    // haplotype_lookup in BTreeMap<Cyp2d6RegionLabel, _> order: (Cyp2d6, Some(star_allele)) sorts by the star-allele string
    std::vector<std::string> a_sub; std::vector<uint8_t> hap_matrix;
    // Cyp2d6Config tables (definitions.rs:242-301)
    std::vector<std::string> tr_key, tr_val, con_a, con_b, singles;
    // pointer arrays handed out through sp_cyp_problem
    std::vector<const char*> p_tsub, p_vref, p_valt, p_vlab, p_asub, p_trk, p_trv, p_ca, p_cb, p_sing;
};

namespace {

const char* cyp_type_name(int t) {                                           // strum strings (region_label.rs:5-34)
    static const char* names[] = {"UNKNOWN", "REP6", "CYP2D6", "link_region", "REP7", "spacer", "CYP2D7", "CYP2D6*5", "Hybrid", "FalseAllele"};
    return (t >= 0 && t <= 9) ? names[t] : "UNKNOWN";
}
std::string cyp_full_allele(int t, bool has_sub, const std::string& sub) {   // region_label.rs:131-170
    if (t == SP_CYP_CYP2D6) return has_sub ? "CYP2D6*" + sub : std::string("CYP2D6");
    if (t == SP_CYP_HYBRID) return has_sub ? sub : std::string("Hybrid");
    if (t == SP_CYP_FALSE_ALLELE) return has_sub ? "FalseAllele_" + sub : std::string("FalseAllele");
    return cyp_type_name(t);
}

struct Tmpl { int type; bool has_sub; std::string sub, seq; };

} // namespace

extern "C" {

int32_t sp_cyp_db_create(sp_ctx* ctx, const sp_cyp_locus* L, const sp_cyp_gene_def* G, const sp_cyp_config* C, sp_cyp_db** out) {
    auto fail = [&](int code, const std::string& msg) { return ctx ? sp_fail(ctx, code, msg) : code; };
    if (!out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!L || !L->chrom_seq || !G || (G->n_alleles && (!G->star_allele || !G->var_off))) return fail(SP_ERR_INVALID_ARG, "sp_cyp_db_create: null argument");
    const uint64_t total_vars = G->n_alleles ? G->var_off[G->n_alleles] : 0;
    if (total_vars && (!G->var_pos || !G->var_ref || !G->var_alt)) return fail(SP_ERR_INVALID_ARG, "sp_cyp_db_create: null variant arrays");
    const uint64_t w0 = L->window_start, w1 = L->window_start + L->window_len;
    bool bad = false;
    auto slice = [&](uint64_t s, uint64_t e) -> std::string {                 // reference_genome.get_slice(chrom, s, e)
        if (s > e || s < w0 || e > w1) { bad = true; return std::string(); }
        return std::string(L->chrom_seq + (s - w0), L->chrom_seq + (e - w0));
    };
    sp_cyp_db* db = new (std::nothrow) sp_cyp_db();
    if (!db) return fail(SP_ERR_OUT_OF_MEMORY, "sp_cyp_db_create");
    db->ctx = ctx;

    // ---- generate_cyp_hybrids (definitions.rs:346-464)
    std::vector<Tmpl> T;
    T.push_back({SP_CYP_CYP2D6, false, "", slice(L->d6_start, L->d6_end)});
    T.push_back({SP_CYP_CYP2D7, false, "", slice(L->d7_start, L->d7_end)});
    const uint64_t pre = 500, post = 3000;                                    // STAR5_PRE_BUFFER / STAR5_POST_BUFFER (definitions.rs:13-14)
    if (L->star5_start < pre) bad = true;
    T.push_back({SP_CYP_DELETION, false, "", bad ? std::string() : slice(L->star5_start - pre, L->star5_start) + slice(L->star5_end, L->star5_end + post)});
    for (int x = 1; x <= 9; ++x) {
        const std::string n = std::to_string(x);
        if (x != 1) {                                                         // breakpoint at the start of the exon = its end() on the forward strand
            const uint64_t b1 = L->d6_exon_end[x - 1], b2 = L->d7_exon_end[x - 1];
            T.push_back({SP_CYP_HYBRID, true, "CYP2D6::CYP2D7::exon" + n, slice(L->d7_start, b2) + slice(b1, L->d6_end)});
            T.push_back({SP_CYP_HYBRID, true, "CYP2D7::CYP2D6::exon" + n, slice(L->d6_start, b1) + slice(b2, L->d7_end)});
        }
        if (x != 9) {                                                         // breakpoint at the end of the exon (start of the intron) = its start()
            const uint64_t b1 = L->d6_exon_start[x - 1], b2 = L->d7_exon_start[x - 1];
            T.push_back({SP_CYP_HYBRID, true, "CYP2D6::CYP2D7::intron" + n, slice(L->d7_start, b2) + slice(b1, L->d6_end)});
            T.push_back({SP_CYP_HYBRID, true, "CYP2D7::CYP2D6::intron" + n, slice(L->d6_start, b1) + slice(b2, L->d7_end)});
        }
    }
    T.push_back({SP_CYP_REP6, false, "", slice(L->rep6_start, L->rep6_end)});
    T.push_back({SP_CYP_REP7, false, "", slice(L->rep7_start, L->rep7_end)});
    T.push_back({SP_CYP_SPACER, false, "", slice(L->spacer_start, L->spacer_end)});
    T.push_back({SP_CYP_LINK_REGION, false, "", slice(L->link_start, L->link_end)});
    db->backbone = slice(L->backbone_start, L->backbone_end); db->backbone_start = L->backbone_start;
    if (bad) { delete db; return fail(SP_ERR_INVALID_ARG, "sp_cyp_db_create: a locus coordinate lies outside the chromosome window"); }
    std::vector<size_t> order(T.size());
    for (size_t i = 0; i < T.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return cyp_full_allele(T[a].type, T[a].has_sub, T[a].sub) < cyp_full_allele(T[b].type, T[b].has_sub, T[b].sub); });
    for (size_t i : order) {
        db->t_type.push_back(T[i].type); db->t_sub.push_back(T[i].sub); db->t_has_sub.push_back(T[i].has_sub ? 1 : 0);
        db->t_full.push_back(cyp_full_allele(T[i].type, T[i].has_sub, T[i].sub)); db->t_seq.push_back(std::move(T[i].seq));
        // mapped_hybrids (haplotyper.rs:117-123): the labels that go through deep typing
        db->t_deep.push_back((T[i].type == SP_CYP_CYP2D6 && !T[i].has_sub) || (T[i].type == SP_CYP_HYBRID && T[i].sub == "CYP2D6::CYP2D7::exon9") ? 1 : 0);
    }

    // ---- LoadedVariants::load_variant_database (haplotyper.rs:650-773): first occurrence of every (pos, ref, alt) in gene_def
    // order, stable sort by position, VI when any definition flags it, label = id or "chrom:pos+1ref>alt"
    struct Raw { uint64_t pos; std::string ref, alt, label; };
    std::vector<Raw> raw;
    std::map<std::tuple<uint64_t, std::string, std::string>, bool> vi_of;      // key -> flagged VI by some definition (also the inserted set)
    const char* chrom = L->chrom_name ? L->chrom_name : "chr22";
    for (uint32_t a = 0; a < G->n_alleles; ++a) for (uint32_t x = G->var_off[a]; x < G->var_off[a + 1]; ++x) {
        if (!G->var_ref[x] || !G->var_alt[x]) { delete db; return fail(SP_ERR_INVALID_ARG, "sp_cyp_db_create: null allele string"); }
        const auto key = std::make_tuple(G->var_pos[x], std::string(G->var_ref[x]), std::string(G->var_alt[x]));
        const bool vi = G->var_vi && G->var_vi[x];
        auto it = vi_of.find(key);
        if (it == vi_of.end()) {
            vi_of.emplace(key, vi);
            Raw r; r.pos = G->var_pos[x]; r.ref = G->var_ref[x]; r.alt = G->var_alt[x];
            if (G->var_id && G->var_id[x]) r.label = G->var_id[x];
            else r.label = std::string(chrom) + ":" + std::to_string(r.pos + 1) + r.ref + ">" + r.alt;      // variant_string (data_types/alleles.rs:99-101)
            raw.push_back(std::move(r));
        } else if (vi) it->second = true;
    }
    std::stable_sort(raw.begin(), raw.end(), [](const Raw& p, const Raw& q) { return p.pos < q.pos; });
    for (uint32_t i = 0; i < raw.size(); ++i) {
        const auto key = std::make_tuple(raw[i].pos, raw[i].ref, raw[i].alt);
        db->v_pos.push_back((int64_t)raw[i].pos); db->v_ref.push_back(raw[i].ref); db->v_alt.push_back(raw[i].alt); db->v_label.push_back(raw[i].label);
        db->v_vi.push_back(vi_of[key] ? 1 : 0);
        db->v_lookup[key] = i;
        db->label_lookup[raw[i].label] = i;                                   // collect() into a map: the last index of a repeated label wins
    }
    const uint32_t NV = (uint32_t)raw.size();
    // the designated region must contain the variant list (haplotyper.rs:110-111)
    if (NV && (L->d6_start > (uint64_t)db->v_pos.front() || L->d6_end < (uint64_t)db->v_pos.back())) { delete db; return fail(SP_ERR_INVALID_ARG, "sp_cyp_db_create: CYP2D6 coordinates do not contain every variant"); }
    for (uint32_t i = 0; i < NV; ++i) {
        const int64_t rel = db->v_pos[i] - (int64_t)L->backbone_start;
        if (rel < 0 || (uint64_t)rel + db->v_ref[i].size() > db->backbone.size()) { delete db; return fail(SP_ERR_INVALID_ARG, "sp_cyp_db_create: a variant lies outside CYP2D6_wfa_backbone"); }
        db->v_rel.push_back((int32_t)rel);
    }

    // ---- haplotype_lookup (haplotyper.rs:57-80): one 0/1 row per star allele, rows in label order, a repeated star allele replaces its row
    std::map<std::string, std::vector<uint8_t>> rows;
    for (uint32_t a = 0; a < G->n_alleles; ++a) {
        std::vector<uint8_t> row(NV, 0);
        for (uint32_t x = G->var_off[a]; x < G->var_off[a + 1]; ++x) row[db->v_lookup[std::make_tuple(G->var_pos[x], std::string(G->var_ref[x]), std::string(G->var_alt[x]))]] = 1;
        rows[G->star_allele[a] ? G->star_allele[a] : ""] = std::move(row);
    }
    for (auto& kv : rows) { db->a_sub.push_back(kv.first); db->hap_matrix.insert(db->hap_matrix.end(), kv.second.begin(), kv.second.end()); }

    if (C) {
        for (uint32_t i = 0; i < C->n_translate; ++i) { db->tr_key.push_back(C->translate_key[i]); db->tr_val.push_back(C->translate_val[i]); }
        for (uint32_t i = 0; i < C->n_connections; ++i) { db->con_a.push_back(C->connection_a[i]); db->con_b.push_back(C->connection_b[i]); }
        for (uint32_t i = 0; i < C->n_singletons; ++i) db->singles.push_back(C->singletons[i]);
    }
    for (size_t i = 0; i < db->t_sub.size(); ++i) db->p_tsub.push_back(db->t_has_sub[i] ? db->t_sub[i].c_str() : nullptr);
    for (uint32_t i = 0; i < NV; ++i) { db->p_vref.push_back(db->v_ref[i].c_str()); db->p_valt.push_back(db->v_alt[i].c_str()); db->p_vlab.push_back(db->v_label[i].c_str()); }
    for (auto& s : db->a_sub) db->p_asub.push_back(s.c_str());
    for (size_t i = 0; i < db->tr_key.size(); ++i) { db->p_trk.push_back(db->tr_key[i].c_str()); db->p_trv.push_back(db->tr_val[i].c_str()); }
    for (size_t i = 0; i < db->con_a.size(); ++i) { db->p_ca.push_back(db->con_a[i].c_str()); db->p_cb.push_back(db->con_b[i].c_str()); }
    for (auto& s : db->singles) db->p_sing.push_back(s.c_str());

    if (ctx) {                                                                // the templates go to HBM once
        std::string blob; std::vector<uint64_t> off(1, 0);
        for (auto& s : db->t_seq) { blob += s; off.push_back(blob.size()); }
        const int32_t rc = sp_seqset_upload(ctx, blob.data(), off.data(), (uint32_t)db->t_seq.size(), &db->templates);
        if (rc != SP_OK) { delete db; return rc; }
    }
    *out = db;
    return SP_OK;
}

void sp_cyp_db_free(sp_cyp_db* db) {
    if (!db) return;
    if (db->templates) sp_seqset_free(db->templates);
    delete db;
}

int32_t sp_cyp_db_info(const sp_cyp_db* db, sp_cyp_db_stats* s) {
    if (!db || !s) return SP_ERR_INVALID_ARG;
    s->n_templates = (uint32_t)db->t_seq.size(); s->n_variants = (uint32_t)db->v_pos.size(); s->n_alleles = (uint32_t)db->a_sub.size();
    s->n_vi = 0; for (uint8_t v : db->v_vi) s->n_vi += v;
    s->first_variant_pos = db->v_pos.empty() ? -1 : db->v_pos.front(); s->last_variant_pos = db->v_pos.empty() ? -1 : db->v_pos.back();
    s->backbone_len = (uint32_t)db->backbone.size();
    return SP_OK;
}

int32_t sp_cyp_db_template(const sp_cyp_db* db, uint32_t i, int32_t* type, const char** subtype, const char** full_allele, const char** seq, uint32_t* len, int32_t* deep) {
    if (!db || i >= db->t_seq.size()) return SP_ERR_INVALID_ARG;
    if (type) *type = db->t_type[i];
    if (subtype) *subtype = db->t_has_sub[i] ? db->t_sub[i].c_str() : nullptr;
    if (full_allele) *full_allele = db->t_full[i].c_str();
    if (seq) *seq = db->t_seq[i].c_str();
    if (len) *len = (uint32_t)db->t_seq[i].size();
    if (deep) *deep = db->t_deep[i];
    return SP_OK;
}

int32_t sp_cyp_db_variant(const sp_cyp_db* db, uint32_t i, int64_t* chrom_pos, const char** ref, const char** alt, const char** label, int32_t* is_vi) {
    if (!db || i >= db->v_pos.size()) return SP_ERR_INVALID_ARG;
    if (chrom_pos) *chrom_pos = db->v_pos[i];
    if (ref) *ref = db->v_ref[i].c_str();
    if (alt) *alt = db->v_alt[i].c_str();
    if (label) *label = db->v_label[i].c_str();
    if (is_vi) *is_vi = db->v_vi[i];
    return SP_OK;
}

int32_t sp_cyp_db_index_label(const sp_cyp_db* db, const char* label, uint32_t* idx) {            // LoadedVariants::index_label (haplotyper.rs:793-798)
    if (!db || !label || !idx) return SP_ERR_INVALID_ARG;
    auto it = db->label_lookup.find(label);
    if (it == db->label_lookup.end()) return SP_ERR_INVALID_ARG;
    *idx = it->second;
    return SP_OK;
}

int32_t sp_cyp_db_index_variant(const sp_cyp_db* db, uint64_t position, const char* ref, const char* alt, uint32_t* idx) {   // index_variant (:781-786)
    if (!db || !ref || !alt || !idx) return SP_ERR_INVALID_ARG;
    auto it = db->v_lookup.find(std::make_tuple(position, std::string(ref), std::string(alt)));
    if (it == db->v_lookup.end()) return SP_ERR_INVALID_ARG;
    *idx = it->second;
    return SP_OK;
}

int32_t sp_cyp_db_allele(const sp_cyp_db* db, uint32_t a, const char** subtype, const uint8_t** row) {
    if (!db || a >= db->a_sub.size()) return SP_ERR_INVALID_ARG;
    if (subtype) *subtype = db->a_sub[a].c_str();
    if (row) *row = db->hap_matrix.data() + (size_t)a * db->v_pos.size();
    return SP_OK;
}

int32_t sp_cyp_db_problem(const sp_cyp_db* db, sp_cyp_problem* pr) {
    if (!db || !pr) return SP_ERR_INVALID_ARG;
    if (!db->templates) return db->ctx ? sp_fail(db->ctx, SP_ERR_INVALID_ARG, "sp_cyp_db_problem: the database was created without a context") : SP_ERR_INVALID_ARG;
    std::memset(pr, 0, sizeof *pr);
    pr->templates = db->templates; pr->template_type = db->t_type.data(); pr->template_subtype = db->p_tsub.data(); pr->template_deep = db->t_deep.data();
    pr->backbone = db->backbone.c_str(); pr->backbone_len = (uint32_t)db->backbone.size();
    pr->n_variants = (uint32_t)db->v_pos.size(); pr->var_pos = db->v_rel.data(); pr->var_ref = db->p_vref.data(); pr->var_alt = db->p_valt.data(); pr->var_is_vi = db->v_vi.data();
    pr->n_alleles = (uint32_t)db->a_sub.size(); pr->allele_subtype = db->p_asub.data(); pr->hap_matrix = db->hap_matrix.data();
    pr->n_translate = (uint32_t)db->tr_key.size(); pr->translate_key = db->p_trk.data(); pr->translate_val = db->p_trv.data();
    pr->n_connections = (uint32_t)db->con_a.size(); pr->connection_a = db->p_ca.data(); pr->connection_b = db->p_cb.data();
    pr->n_singletons = (uint32_t)db->singles.size(); pr->singletons = db->p_sing.data();
    // defaults of the CLI (src/cli/diplotype.rs:155-183)
    pr->min_consensus_count = 3; pr->dual_max_ed_delta = 100; pr->min_consensus_fraction = 0.10; pr->infer_connections = 0; pr->normalize_d6_only = 0;
    pr->var_label = db->p_vlab.data();
    return SP_OK;
}

} // extern "C"
