// sp_host.hip -- the scalar host-side decisions of the hot path (no device work), behind the same C ABI.
#include "sp_internal.h"
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <map>
#include <set>

namespace {
// statrs::function::factorial::ln_factorial: products up to 170!, ln_gamma beyond
double ln_fact(uint64_t n) {
    if (n <= 170) { double f = 1.0; for (uint64_t i = 2; i <= n; ++i) f *= (double)i; return std::log(f); }
    return std::lgamma((double)n + 1.0);
}
// statrs Binomial::ln_pmf
double binom_ln_pmf(double p, uint64_t n, uint64_t x) {
    if (x > n) return -INFINITY;
    if (p == 0.0) return x == 0 ? 0.0 : -INFINITY;
    if (p == 1.0) return x == n ? 0.0 : -INFINITY;
    return ln_fact(n) - ln_fact(x) - ln_fact(n - x) + (double)x * std::log(p) + (double)(n - x) * std::log(1.0 - p);
}
// statrs Binomial::cdf = I_{1-p}(n - x, x + 1), evaluated as the finite sum
double binom_cdf(double p, uint64_t n, uint64_t x) {
    if (x >= n) return 1.0;
    double acc = 0.0;
    for (uint64_t k = 0; k <= x; ++k) acc += std::exp(binom_ln_pmf(p, n, k));
    return acc > 1.0 ? 1.0 : acc;
}
double normal_ln_pdf(double mean, double sd, double x) {
    const double ln_sqrt_2pi = 0.91893853320467274178032973640561763986139747363778341281715;
    const double d = (x - mean) / sd;
    return (-0.5 * d * d) - ln_sqrt_2pi - std::log(sd);
}
bool cyp2d(int t) { return t == SP_CYP_CYP2D6 || t == SP_CYP_CYP2D7 || t == SP_CYP_DELETION || t == SP_CYP_HYBRID; }
// parse_sequence (normalized_variant.rs:262-279)
std::string cpic_bases(const std::string& t) {
    const size_t open = t.find('(');
    if (open != std::string::npos && open > 0 && t.size() >= open + 3 && t.back() == ')' &&
        std::all_of(t.begin(), t.begin() + open, [](char c) { return c >= 'A' && c <= 'Z'; }) &&
        std::all_of(t.begin() + open + 1, t.end() - 1, [](char c) { return c >= '0' && c <= '9'; })) {
        const unsigned long long count = std::strtoull(t.c_str() + open + 1, nullptr, 10);
        std::string unit = t.substr(0, open), out;
        if (count * unit.size() > (1u << 20)) return std::string(1, '?');          // refused later by the ACGT check
        for (unsigned long long i = 0; i < count; ++i) out += unit;
        return out;
    }
    if (t.rfind("delins", 0) == 0) return t.substr(6);
    if (t.rfind("ins", 0) == 0) return t.substr(3);
    if (t.rfind("del", 0) == 0) return std::string();
    return t;
}

} // namespace

extern "C" {

int32_t sp_hla_is_passing_dual(uint64_t counts1, uint64_t counts2, double min_consensus_fraction, double expected_maf, double min_cdf,
                               double* maf_out, double* cdf_out) {
    const uint64_t total = counts1 + counts2, minor = counts1 < counts2 ? counts1 : counts2;
    const double maf = (double)minor / (double)total;
    const double cdf = binom_cdf(expected_maf, total, minor);
    if (maf_out) *maf_out = maf;
    if (cdf_out) *cdf_out = cdf;
    return (maf >= min_consensus_fraction && cdf >= min_cdf) ? 1 : 0;
}

int32_t sp_hla_is_hemizygous_better(const int64_t* s1, const int64_t* s2, const uint8_t* is_c1, uint32_t n, int32_t is_dual,
                                    uint64_t dual_max_ed_delta, double normalized_coverage, double* haploid_cost_out, double* diploid_cost_out) {
    uint64_t min_ed = 0;
    if (is_dual) {
        uint64_t c1 = 0, c2 = 0;
        for (uint32_t i = 0; i < n; ++i) {
            // a missing score means the read hit dual_max_ed_delta relative to the other consensus (caller.rs:1597-1599)
            const uint64_t a = s1[i] >= 0 ? (uint64_t)s1[i] : (s2[i] >= 0 ? (uint64_t)s2[i] : 0) + dual_max_ed_delta;
            const uint64_t b = s2[i] >= 0 ? (uint64_t)s2[i] : (s1[i] >= 0 ? (uint64_t)s1[i] : 0) + dual_max_ed_delta;
            const uint64_t m = a < b ? a : b;
            c1 += a - m; c2 += b - m;
        }
        min_ed = c1 < c2 ? c1 : c2;
    }
    const double read_count = (double)n;
    const double haploid_ed_cost = 2.0 * (double)min_ed;                       // ln_ed_penalty
    const double nc_hap = normalized_coverage >= 0.0 ? normalized_coverage : read_count;
    const double nc_dev = nc_hap * 0.1;
    const double haploid_cost = haploid_ed_cost + std::fabs(normal_ln_pdf(nc_hap, nc_dev, read_count));
    uint64_t obs1 = 0; for (uint32_t i = 0; i < n; ++i) obs1 += is_c1[i] ? 1 : 0;
    const double balance = is_dual ? 2.0 * std::fabs(binom_ln_pmf(0.5, n, obs1)) : 0.0;     // diploid_balance_penalty
    const double diploid_cost = balance + std::fabs(normal_ln_pdf(2.0 * nc_hap, nc_dev, read_count));
    if (haploid_cost_out) *haploid_cost_out = haploid_cost;
    if (diploid_cost_out) *diploid_cost_out = diploid_cost;
    return haploid_cost < diploid_cost ? 1 : 0;
}

uint64_t sp_hpc_pos(const char* seq, uint64_t len, uint64_t position) {
    uint64_t total = 0, offset = 0, i = 0;
    while (i < len) {
        uint64_t run = 1; while (i + run < len && seq[i + run] == seq[i]) ++run;
        total += run;
        if (position < total) break;
        ++offset; i += run;
    }
    return offset;
}

uint64_t sp_hpc(const char* seq, uint64_t len, char* out) {
    uint64_t o = 0;
    for (uint64_t i = 0; i < len; ++i) if (i == 0 || seq[i] != seq[i - 1]) out[o++] = seq[i];
    return o;
}

uint32_t sp_cyp_chain_to_hap(const int32_t* chain, uint32_t n, const int32_t* hap_type, const char* const* hap_subtype,
                             uint32_t n_translate, const char* const* translate_key, const char* const* translate_val,
                             int32_t detail, char* out, uint32_t cap) {
    auto label = [&](int h) -> std::string {                                    // simplify_allele (region_label.rs:77-128)
        const int t = hap_type[h]; const char* sub = hap_subtype ? hap_subtype[h] : nullptr;
        if (t == SP_CYP_DELETION) return "*5";
        if (!sub) return t == SP_CYP_CYP2D6 ? "CYP2D6" : "Hybrid";
        for (uint32_t i = 0; i < n_translate; ++i) if (std::strcmp(translate_key[i], sub) == 0) return std::string("*") + translate_val[i];
        if (detail == 1) return std::string("*") + sub;
        char* endp = nullptr; const double v = std::strtod(sub, &endp);
        if (sub[0] && sub[0] != ' ' && endp && *endp == '\0' && !(sub[0] == '0' && (sub[1] == 'x' || sub[1] == 'X'))) return "*" + std::to_string((long long)std::floor(v));
        return std::string("*") + sub;
    };
    int non_deletion = 0;
    for (uint32_t x = 0; x < n; ++x) { const int t = hap_type[chain[x]]; if (cyp2d(t) && t != SP_CYP_CYP2D7 && t != SP_CYP_DELETION) ++non_deletion; }
    std::string res, prev; int run = 0;
    auto flush = [&]() { if (run > 0) { if (!res.empty()) res += " + "; res += prev; if (run > 1) res += "x" + std::to_string(run); } };
    for (int x = (int)n - 1; x >= 0; --x) {                                     // chains are reported in reverse (caller.rs:913-915)
        const int h = chain[x], t = hap_type[h];
        if (!(cyp2d(t) && t != SP_CYP_CYP2D7)) continue;
        if (t == SP_CYP_DELETION && non_deletion > 0) continue;
        const std::string cur = label(h);
        if (run > 0 && cur == prev) ++run; else { flush(); prev = cur; run = 1; }
    }
    flush();
    if (out && cap) { const size_t k = std::min<size_t>(cap - 1, res.size()); std::memcpy(out, res.data(), k); out[k] = '\0'; }
    return (uint32_t)res.size();
}

int32_t sp_variant_normalize(const char* chrom_seq, uint64_t chrom_len, uint64_t position, const char* ref_allele, const char* alt_allele,
                             uint64_t* out_position, char* out_ref, char* out_alt, uint32_t cap) {
    if (!ref_allele || !alt_allele || !out_position || !out_ref || !out_alt || cap == 0) return SP_ERR_INVALID_ARG;
    const std::string ref_text(ref_allele), alt_text(alt_allele);
    if (ref_text.empty()) return SP_ERR_BAD_VARIANT;
    if (ref_text == "del" && alt_text.rfind("ins", 0) != 0) return SP_ERR_BAD_VARIANT;
    std::string r = cpic_bases(ref_text), a = cpic_bases(alt_text);
    uint64_t pos = position;
    if (chrom_seq) {
        if (pos + r.size() > chrom_len) return SP_ERR_BAD_VARIANT;                  // the reference would panic on the slice
        if (r.compare(0, r.size(), chrom_seq + pos, r.size()) != 0) return SP_ERR_BAD_VARIANT;
    }
    if (r.empty() && a.empty()) return SP_ERR_BAD_VARIANT;
    if (r.empty() || a.empty()) {                                                   // one side empty: anchor on a reference base
        if (a.empty()) { if (pos == 0) return SP_ERR_BAD_VARIANT; if (chrom_seq) --pos; }
        if (chrom_seq) { if (pos >= chrom_len) return SP_ERR_BAD_VARIANT; r.insert(r.begin(), chrom_seq[pos]); a.insert(a.begin(), chrom_seq[pos]); }
    }
    if (r.empty() || a.empty()) return SP_ERR_BAD_VARIANT;                          // (Rust indexes [len-1] and panics)
    // shared suffix, then shared prefix, never below one base on either side
    size_t room = std::min(r.size(), a.size()) - 1, cut = 0;
    while (cut < room && r[r.size() - 1 - cut] == a[a.size() - 1 - cut]) ++cut;
    r.resize(r.size() - cut); a.resize(a.size() - cut);
    room = std::min(r.size(), a.size()) - 1; cut = 0;
    while (cut < room && r[cut] == a[cut]) ++cut;
    r.erase(0, cut); a.erase(0, cut); pos += cut;
    // left shift: while the last bases agree, roll both alleles one reference base to the left
    while (r.back() == a.back() && pos > 0 && chrom_seq) {
        --pos;
        r.pop_back(); a.pop_back();
        r.insert(r.begin(), chrom_seq[pos]); a.insert(a.begin(), chrom_seq[pos]);
    }
    auto acgt = [](const std::string& t) { return std::all_of(t.begin(), t.end(), [](char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }); };
    if (!acgt(r) || !acgt(a)) return SP_ERR_BAD_VARIANT;
    if (r.size() + 1 > cap || a.size() + 1 > cap) return SP_ERR_CAPACITY;
    *out_position = pos;
    std::memcpy(out_ref, r.c_str(), r.size() + 1); std::memcpy(out_alt, a.c_str(), a.size() + 1);
    return SP_OK;
}

int32_t sp_variant_multi_normalize(const char* chrom_seq, uint64_t chrom_len, uint64_t position, const char* ref_allele, const char* alt_allele,
                                   uint32_t max_out, uint32_t* n_out, uint8_t* is_none, uint64_t* out_position, char* out_ref, char* out_alt, uint32_t cap) {
    if (!ref_allele || !alt_allele || !n_out || !is_none || !out_position || !out_ref || !out_alt) return SP_ERR_INVALID_ARG;
    static const struct { char code; const char* bases; } iupac[] = { {'K', "GT"}, {'M', "AC"}, {'R', "AG"}, {'S', "CG"}, {'W', "AT"}, {'Y', "CT"},
                                                                      {'B', "CGT"}, {'D', "AGT"}, {'H', "ACT"}, {'V', "ACG"} };
    std::vector<std::string> alts;
    const std::string alt_text(alt_allele);
    if (alt_text.size() == 1) for (const auto& e : iupac) if (e.code == alt_text[0]) for (const char* b = e.bases; *b; ++b) alts.emplace_back(1, *b);
    if (alts.empty()) {                                                             // str::split("; ") keeps empty pieces
        size_t from = 0;
        for (;;) { const size_t at = alt_text.find("; ", from); alts.push_back(alt_text.substr(from, at == std::string::npos ? at : at - from)); if (at == std::string::npos) break; from = at + 2; }
    }
    *n_out = (uint32_t)alts.size();
    if (alts.size() > max_out) return SP_ERR_CAPACITY;
    for (size_t i = 0; i < alts.size(); ++i) {
        is_none[i] = alts[i] == ref_allele;
        if (is_none[i]) continue;
        const int32_t rc = sp_variant_normalize(chrom_seq, chrom_len, position, ref_allele, alts[i].c_str(), out_position + i, out_ref + i * (size_t)cap, out_alt + i * (size_t)cap, cap);
        if (rc != SP_OK) return rc;
    }
    return SP_OK;
}

uint32_t sp_diplotype_string(const char* hap1, const char* hap2, int32_t pharmcat, char* out, uint32_t cap) {
    auto side = [&](const char* h) { std::string t(h ? h : ""); return (pharmcat && t.find('+') != std::string::npos) ? "[" + t + "]" : t; };
    const std::string res = side(hap1) + "/" + side(hap2);
    if (out && cap) { const size_t k = std::min<size_t>(cap - 1, res.size()); std::memcpy(out, res.data(), k); out[k] = '\0'; }
    return (uint32_t)res.size();
}

uint32_t sp_inexact_haplotype(const char* base_haplotype, uint32_t n_variants, const char* const* labels, const uint8_t* is_vi,
                              const int32_t* states, int32_t* match_type, char* out, uint32_t cap) {
    struct Rv { std::string label; int vi, state; };
    std::vector<Rv> set;
    for (uint32_t i = 0; i < n_variants; ++i) set.push_back(Rv{ labels[i] ? labels[i] : "", is_vi[i] ? 1 : 0, states[i] });
    auto less = [](const Rv& a, const Rv& b) { if (a.label != b.label) return a.label < b.label; if (a.vi != b.vi) return a.vi < b.vi; return a.state < b.state; };
    std::sort(set.begin(), set.end(), less);
    set.erase(std::unique(set.begin(), set.end(), [](const Rv& a, const Rv& b) { return a.label == b.label && a.vi == b.vi && a.state == b.state; }), set.end());
    bool core = true, sub = true;
    std::string hap(base_haplotype ? base_haplotype : "");
    bool modified = false;
    for (const Rv& v : set) {
        if (v.state == SP_REL_MATCH) continue;
        sub = false; if (v.vi) core = false;
        hap += ' ';
        hap += v.state == SP_REL_UNEXPECTED ? '+' : (v.state == SP_REL_MISSING ? '-' : '?');
        hap += v.label;
        modified = true;
    }
    if (modified) hap = "(" + hap + ")";
    if (match_type) *match_type = sub ? SP_INEXACT_SUBALLELE_MATCH : (core ? SP_INEXACT_CORE_MATCH : SP_INEXACT_NO_MATCH);
    if (out && cap) { const size_t k = std::min<size_t>(cap - 1, hap.size()); std::memcpy(out, hap.data(), k); out[k] = '\0'; }
    return (uint32_t)hap.size();
}

int32_t sp_cyp_build_chains(uint32_t n_haps, const int32_t* hap_type, uint32_t n_reads, const uint32_t* read_seg_off,
                            const uint64_t* ed, const uint8_t* kept,
                            uint32_t* read_index, uint32_t* read_chain_off, uint32_t* chain_off, uint32_t chain_cap,
                            uint32_t* chain_items, uint32_t item_cap, uint32_t* read_w_off, uint32_t* w_seg,
                            uint64_t* unique_counts, uint8_t* false_allele, sp_chain_build_info* info) {
    if (!hap_type || !read_seg_off || !ed || !kept || !read_index || !read_chain_off || !chain_off || !chain_items || !read_w_off ||
        !w_seg || !unique_counts || !false_allele || !info || n_haps == 0) return SP_ERR_INVALID_ARG;
    // pass 1: the minimum-edit choice list of every kept segment, and the unique-support counts.  A unique minimum is counted
    // once per chain being extended (caller.rs:470-483), i.e. by the product of the choice counts of the read's earlier segments.
    std::vector<uint32_t> choice_off(1, 0), choices;          // per recorded row
    std::vector<uint32_t> rec_read, rec_row0, rec_rows;
    std::fill(unique_counts, unique_counts + n_haps, 0ull);
    uint32_t n_rows = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint32_t row0 = n_rows;
        const size_t cmark = choices.size();
        std::vector<uint64_t> pending(n_haps, 0);             // counts are committed even if the read ends up unrecorded (as in the reference)
        uint64_t width = 1;
        for (uint32_t sg = read_seg_off[r]; sg < read_seg_off[r + 1]; ++sg) {
            if (!kept[sg]) continue;
            const uint64_t* w = ed + (size_t)sg * n_haps;
            const uint64_t mn = *std::min_element(w, w + n_haps);
            const size_t before = choices.size();
            for (uint32_t c = 0; c < n_haps; ++c) if (w[c] == mn) choices.push_back(c);
            const uint32_t k = (uint32_t)(choices.size() - before);
            if (k == 1) pending[choices.back()] += width;
            width *= k;
            if (width > (1ull << 31)) return SP_ERR_CAPACITY;
            choice_off.push_back((uint32_t)choices.size());
            w_seg[n_rows++] = sg;
        }
        for (uint32_t c = 0; c < n_haps; ++c) unique_counts[c] += pending[c];
        if (n_rows == row0) { choices.resize(cmark); continue; }   // no segment or none kept: the read is not recorded
        rec_read.push_back(r); rec_row0.push_back(row0); rec_rows.push_back(n_rows - row0);
    }
    // pass 2: enumerate each read's chains as a mixed-radix counter (first segment most significant = the reference's order)
    uint64_t nc = 0, ni = 0; bool fits = true, collapse = false;
    read_chain_off[0] = 0; read_w_off[0] = 0;
    if (chain_cap) chain_off[0] = 0;
    for (size_t k = 0; k < rec_read.size(); ++k) {
        const uint32_t row0 = rec_row0[k], m = rec_rows[k];
        read_index[k] = rec_read[k];
        read_w_off[k + 1] = row0 + m;
        bool supported = true;                                  // segments with several minima keep only supported consensuses
        std::vector<std::vector<uint32_t>> opts(m);
        for (uint32_t x = 0; x < m; ++x) {
            for (uint32_t o = choice_off[row0 + x]; o < choice_off[row0 + x + 1]; ++o) if (unique_counts[choices[o]] > 0) opts[x].push_back(choices[o]);
            if (opts[x].empty()) supported = false;
        }
        if (!supported) { collapse = true; read_chain_off[k + 1] = (uint32_t)nc; continue; }
        std::vector<uint32_t> digit(m, 0);
        for (;;) {
            if (fits && nc + 1 <= chain_cap && ni + m <= item_cap) {
                for (uint32_t x = 0; x < m; ++x) chain_items[ni + x] = opts[x][digit[x]];
                chain_off[nc + 1] = (uint32_t)(ni + m);
            } else fits = false;
            ++nc; ni += m;
            int x = (int)m - 1;
            while (x >= 0 && ++digit[x] == opts[x].size()) digit[x--] = 0;
            if (x < 0) break;
        }
        read_chain_off[k + 1] = (uint32_t)nc;
    }
    for (uint32_t h = 0; h < n_haps; ++h)
        false_allele[h] = unique_counts[h] == 0 && hap_type[h] != SP_CYP_UNKNOWN && hap_type[h] != SP_CYP_FALSE_ALLELE;
    info->n_reads = (uint32_t)rec_read.size(); info->n_chains = (uint32_t)nc; info->n_items = (uint32_t)ni; info->n_rows = n_rows;
    if (collapse) return SP_ERR_CHAIN_COLLAPSE;
    return fits ? SP_OK : SP_ERR_CAPACITY;
}

// is_deletion (src/diplotyper.rs:1020-1174).  Kept on the host: a handful of interval tests per SV record.
int32_t sp_variant_is_deletion(const sp_sv_definitions* defs, uint64_t start, uint64_t end, int32_t* kind, int32_t* index) {
    if (!defs || !kind || !index || defs->n_genes < 0 || defs->n_full < 0 || defs->n_partial < 0) return SP_ERR_INVALID_ARG;
    *kind = 0; *index = -1;
    const sp_sv_definitions& d = *defs;
    auto inside = [&](int64_t s, int64_t e) { return (uint64_t)s >= start && (uint64_t)e <= end; };
    auto known = [&](int32_t g) { return g >= 0 && g < d.n_genes; };

    // full-gene deletions: the set of named genes the region swallows whole
    if (d.n_full > 0) {
        std::set<int32_t> gone;
        for (int32_t i = 0; i < d.full_off[d.n_full]; ++i) {
            const int32_t g = d.full_gene[i];
            if (!known(g)) return SP_ERR_INVALID_ARG;
            if (inside(d.gene_start[g], d.gene_end[g])) gone.insert(g);
        }
        for (int32_t k = 0; k < d.n_full; ++k) {
            const std::set<int32_t> want(d.full_gene + d.full_off[k], d.full_gene + d.full_off[k + 1]);
            if (d.full_generic[k]) {
                if (std::includes(gone.begin(), gone.end(), want.begin(), want.end())) *index = k;     // a later specific one may still win
            } else if (want == gone) { *index = k; break; }
        }
        if (*index >= 0) { *kind = 1; return SP_OK; }
    }

    // partial deletions: per named gene, the transcript-order range of exons the region swallows whole
    if (d.n_partial > 0) {
        std::map<int32_t, std::pair<int32_t, int32_t>> gone;
        for (int32_t i = 0; i < d.partial_off[d.n_partial]; ++i) if (!known(d.partial_gene[i])) return SP_ERR_INVALID_ARG;
        for (int32_t i = 0; i < d.partial_off[d.n_partial]; ++i) {
            const int32_t g = d.partial_gene[i];
            if (gone.count(g)) continue;
            const int32_t e0 = d.exon_off[g], ne = d.exon_off[g + 1] - e0;
            int32_t lo = -1, hi = -1;
            for (int32_t x = 0; x < ne; ++x) if (inside(d.exon_start[e0 + x], d.exon_end[e0 + x])) { if (lo < 0) lo = x; hi = x; }
            if (lo < 0) continue;
            gone[g] = d.gene_forward[g] ? std::make_pair(lo, hi + 1) : std::make_pair(ne - 1 - hi, ne - lo);
        }
        for (int32_t k = 0; k < d.n_partial; ++k) {
            std::map<int32_t, std::pair<int32_t, int32_t>> want;
            bool keys_gone = true;
            for (int32_t i = d.partial_off[k]; i < d.partial_off[k + 1]; ++i) {
                want[d.partial_gene[i]] = std::make_pair(d.partial_first[i], d.partial_end[i]);
                keys_gone = keys_gone && gone.count(d.partial_gene[i]) > 0;
            }
            if (d.partial_generic[k]) { if (keys_gone) *index = k; }
            else if (want == gone) { *index = k; break; }
        }
        if (*index >= 0) *kind = 2;
    }
    return SP_OK;
}


int32_t sp_hla_normalized_coverage(const sp_hla_realign* realign, uint32_t n_reads, const uint32_t* normalizing_genes, uint32_t n_normalizing,
                                   double* normalized_coverage) {
    if ((n_reads && !realign) || (n_normalizing && !normalizing_genes) || !normalized_coverage) return SP_ERR_INVALID_ARG;
    uint64_t read_total = 0, hap_total = 0;
    for (uint32_t k = 0; k < n_normalizing; ++k) {
        uint64_t in_bucket = 0;                                        // gene_buckets holds the reads realign_record placed in the gene (caller.rs:584)
        for (uint32_t r = 0; r < n_reads; ++r) if (realign[r].status == 0 && realign[r].gene == (int32_t)normalizing_genes[k]) ++in_bucket;
        if (in_bucket) { read_total += in_bucket; hap_total += 2; }
    }
    *normalized_coverage = hap_total ? (double)read_total / (double)hap_total : -1.0;
    return SP_OK;
}
} // extern "C"
