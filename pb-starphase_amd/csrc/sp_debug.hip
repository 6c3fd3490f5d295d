// sp_debug.hip -- the HLA debug file (host only): HlaDebug / ReadMappingStats / PairedMappingStats / DetailedMappingStats /
// DualPassingStats (src/hla/debug.rs:7-221) written as save_json writes them (src/util/file_io.rs:37-52), and the CIGAR / MD strings
// minimap2 attaches to a mapping (Alignment::cigar_str / md; what DetailedMappingStats::from_mapping copies, :148-172) derived from
// the edit events of the library's own alignment.
#include "sp_internal.h"
#include "sp_json.h"
#include <zlib.h>
#include <map>

using spj::Value;

struct sp_hla_debug {
    struct Read { bool has_best = false; std::string best_id, best_star; std::map<std::string, Value> mappings; };
    std::map<std::string, std::map<std::string, Read>> reads;          // gene -> qname -> ReadMappingStats
    bool has_dual = false;
    std::map<std::string, Value> dual;                                  // Option<BTreeMap<String, DualPassingStats>>
    std::string err, text;
};

namespace {

int32_t fail(sp_hla_debug* d, const std::string& m) { d->err = m; return SP_ERR_INVALID_ARG; }

Value detailed(const sp_detailed_mapping* m) {
    if (!m || !m->present) return Value();
    Value v = spj::object();
    v.obj.emplace_back("query_len", spj::num((int64_t)m->query_len)); v.obj.emplace_back("target_len", spj::num((int64_t)m->target_len));
    v.obj.emplace_back("match_len", spj::num((int64_t)m->match_len)); v.obj.emplace_back("nm", spj::num((int64_t)m->nm));
    v.obj.emplace_back("query_unmapped", spj::num((int64_t)m->query_unmapped)); v.obj.emplace_back("target_unmapped", spj::num((int64_t)m->target_unmapped));
    v.obj.emplace_back("cigar", spj::str(m->cigar ? m->cigar : "")); v.obj.emplace_back("md", spj::str(m->md ? m->md : ""));
    return v;
}

} // namespace

extern "C" {

// cigar_str as minimap2 builds it without --eqx (M for aligned columns, I = bases of the query only, D = bases of the target only) and the
// MD tag of the SAM specification over the same columns; A is the query, B the target (include/starphase_hip.h: events)
int32_t sp_aln_strings(const sp_aln* aln, const uint32_t* events, const char* target, uint64_t target_len,
                       char* cigar, uint32_t cigar_cap, char* md, uint32_t md_cap, uint64_t* match_len) {
    if (!aln || (!events && aln->nm > 0) || !target) return SP_ERR_INVALID_ARG;
    if (!aln->ok || aln->b_start < 0 || aln->b_end < aln->b_start || (uint64_t)aln->b_end > target_len) return SP_ERR_INVALID_ARG;
    std::string cg, tag;
    char last_op = 0; int64_t run = 0, same = 0, mismatches = 0, deleted = 0; bool in_del = false;
    auto push = [&](char op, int64_t n) {
        if (n <= 0) return;
        if (op == last_op) { run += n; return; }
        if (run > 0) { cg += std::to_string(run); cg += last_op; }
        last_op = op; run = n;
    };
    int64_t j = aln->b_start;
    for (int32_t e = 0; e < aln->nm; ++e) {
        const uint32_t type = events[e] >> 30; const int64_t bpos = (int64_t)(events[e] & 0x3FFFFFFFu);
        if (bpos < j || bpos > aln->b_end || (type != SP_EV_I && bpos >= aln->b_end)) return SP_ERR_INVALID_ARG;
        if (bpos > j) { push('M', bpos - j); same += bpos - j; in_del = false; j = bpos; }
        if (type == SP_EV_X) {
            push('M', 1); tag += std::to_string(same); tag += target[j]; same = 0; in_del = false; ++mismatches; ++j;
        } else if (type == SP_EV_D) {
            push('D', 1);
            if (!in_del) { tag += std::to_string(same); tag += '^'; same = 0; in_del = true; }
            tag += target[j]; ++deleted; ++j;
        } else if (type == SP_EV_I) {
            push('I', 1);                                             // the query-only bases do not appear in MD; a deletion run on either side of them stays one run
        } else return SP_ERR_INVALID_ARG;
    }
    if (aln->b_end > j) { push('M', aln->b_end - j); same += aln->b_end - j; }
    if (run > 0) { cg += std::to_string(run); cg += last_op; }
    tag += std::to_string(same);
    if (match_len) *match_len = (uint64_t)((aln->b_end - aln->b_start) - mismatches - deleted);
    int32_t rc = SP_OK;
    if (cigar && cigar_cap) { const size_t n = std::min<size_t>(cg.size(), cigar_cap - 1); std::memcpy(cigar, cg.data(), n); cigar[n] = 0; if (n < cg.size()) rc = SP_ERR_CAPACITY; }
    if (md && md_cap) { const size_t n = std::min<size_t>(tag.size(), md_cap - 1); std::memcpy(md, tag.data(), n); md[n] = 0; if (n < tag.size()) rc = SP_ERR_CAPACITY; }
    return rc;
}

int32_t sp_hla_debug_create(sp_hla_debug** out) {
    if (!out) return SP_ERR_INVALID_ARG;
    *out = new sp_hla_debug();
    return SP_OK;
}
void sp_hla_debug_free(sp_hla_debug* d) { delete d; }
const char* sp_hla_debug_last_error(const sp_hla_debug* d) { return d ? d->err.c_str() : ""; }

// HlaDebug::add_read (:29-40) of a ReadMappingStats whose best match is set (set_best_match) or not (best_id NULL)
int32_t sp_hla_debug_add_read(sp_hla_debug* d, const char* gene, const char* qname, const char* best_id, const char* best_star) {
    if (!d || !gene || !qname) return SP_ERR_INVALID_ARG;
    auto& g = d->reads[gene];
    if (g.count(qname)) return fail(d, std::string("Entry ") + qname + " is already occupied");
    sp_hla_debug::Read r;
    if (best_id) { r.has_best = true; r.best_id = best_id; r.best_star = best_star ? best_star : ""; }
    g.emplace(qname, std::move(r));
    return SP_OK;
}

// ReadMappingStats::add_mapping (:105-122) on a read added before
int32_t sp_hla_debug_add_mapping(sp_hla_debug* d, const char* gene, const char* qname, const char* hla_id,
                                 const sp_detailed_mapping* cdna, const sp_detailed_mapping* dna) {
    if (!d || !gene || !qname || !hla_id) return SP_ERR_INVALID_ARG;
    auto g = d->reads.find(gene);
    if (g == d->reads.end() || !g->second.count(qname)) return fail(d, std::string("no read ") + qname + " in " + gene);
    auto& r = g->second[qname];
    if (r.mappings.count(hla_id)) return fail(d, std::string("Entry ") + hla_id + " is already occupied!");
    Value p = spj::object();
    p.obj.emplace_back("cdna_mapping", detailed(cdna)); p.obj.emplace_back("dna_mapping", detailed(dna));
    r.mappings.emplace(hla_id, std::move(p));
    return SP_OK;
}

// HlaDebug::add_dual_passing_stats (:46-61) with what is_passing_dual returns for the gene (src/hla/caller.rs:1225-1247): new_dual when
// a dual consensus was found, new_non_dual otherwise
int32_t sp_hla_debug_add_dual_stats(sp_hla_debug* d, const char* gene, const sp_hla_call* call) {
    if (!d || !gene || !call) return SP_ERR_INVALID_ARG;
    if (d->dual.count(gene)) return fail(d, std::string("Entry ") + gene + " is already occupied");
    Value v = spj::object();
    const bool dual = call->is_dual != 0;
    v.obj.emplace_back("is_passing", spj::boolean(dual && call->dual_passed)); v.obj.emplace_back("is_dual", spj::boolean(dual));
    v.obj.emplace_back("counts1", dual ? spj::num(call->counts1) : Value()); v.obj.emplace_back("counts2", dual ? spj::num(call->counts2) : Value());
    v.obj.emplace_back("maf", dual ? spj::real(call->maf) : Value()); v.obj.emplace_back("cdf", dual ? spj::real(call->cdf) : Value());
    d->has_dual = true;
    d->dual.emplace(gene, std::move(v));
    return SP_OK;
}

int32_t sp_hla_debug_json(sp_hla_debug* d, const char** text, uint64_t* len) {
    if (!d || !text) return SP_ERR_INVALID_ARG;
    Value root = spj::object(), genes = spj::object();
    for (const auto& g : d->reads) {
        Value reads = spj::object();
        for (const auto& r : g.second) {
            Value rs = spj::object();
            rs.obj.emplace_back("best_match_id", r.second.has_best ? spj::str(r.second.best_id) : Value());
            rs.obj.emplace_back("best_match_star", r.second.has_best ? spj::str(r.second.best_star) : Value());
            Value maps = spj::object();
            for (const auto& m : r.second.mappings) maps.obj.emplace_back(m.first, m.second);
            rs.obj.emplace_back("mapping_stats", std::move(maps));
            reads.obj.emplace_back(r.first, std::move(rs));
        }
        genes.obj.emplace_back(g.first, std::move(reads));
    }
    root.obj.emplace_back("read_mapping_stats", std::move(genes));
    Value dual;
    if (d->has_dual) { dual = spj::object(); for (const auto& kv : d->dual) dual.obj.emplace_back(kv.first, kv.second); }
    root.obj.emplace_back("dual_passing_stats", std::move(dual));
    d->text.clear();
    spj::write_pretty(d->text, root);
    *text = d->text.c_str();
    if (len) *len = d->text.size();
    return SP_OK;
}

int32_t sp_hla_debug_save(sp_hla_debug* d, const char* path) {
    if (!d || !path) return SP_ERR_INVALID_ARG;
    const char* text; uint64_t len;
    sp_hla_debug_json(d, &text, &len);
    const std::string p(path);
    if (p.size() >= 3 && p.compare(p.size() - 3, 3, ".gz") == 0) {
        gzFile f = gzopen(path, "wb9");
        if (!f) return fail(d, "cannot create " + p);
        const bool ok = gzwrite(f, text, (unsigned)len) == (int)len;
        if (gzclose(f) != Z_OK || !ok) return fail(d, "cannot write " + p);
        return SP_OK;
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(d, "cannot create " + p);
    const bool ok = std::fwrite(text, 1, (size_t)len, f) == (size_t)len;
    if (std::fclose(f) != 0 || !ok) return fail(d, "cannot write " + p);
    return SP_OK;
}

} // extern "C"
