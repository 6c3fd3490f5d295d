// sp_io.hip -- decoding the input files (host only): BGZF / BAM / BAI and VCF.
//
// Replaces what the reference gets from rust-htslib: IndexedReader::fetch + records() over a gene region (src/hla/caller.rs:523-596,
// src/cyp2d6/caller.rs:96-139) and the region fetches of load_vcf_variants / load_sv_vcf_variants (src/diplotyper.rs:551-857).  The
// formats are the published ones (SAMv1 section 4: BGZF blocks, BAM records, the BAI binning index; VCFv4.2); nothing here is taken from
// htslib.  Outputs are the library's own inputs: ASCII bases + offsets for sp_seqset_upload, sp_vcf_allele / sp_vcf_deletion rows for
// sp_variant_gene_problem.
#include "sp_internal.h"
#include <atomic>
#include <thread>
#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

namespace {

void put_err(char* err, uint32_t cap, const std::string& m) { if (err && cap) { const size_t k = std::min<size_t>(cap - 1, m.size()); std::memcpy(err, m.data(), k); err[k] = '\0'; } }

// ------------------------------------------------------------------------------------------------ BGZF
// a series of gzip members of at most 64 KiB, each with a 'BC' extra field holding its size; a virtual offset is
// (file offset of the block << 16) | offset inside the inflated block
struct Bgzf {
    FILE* f = nullptr;
    // A batch of consecutive blocks, inflated together by a few threads (a block is an independent gzip member).  The first batch after a
    // seek is small -- a region fetch of a handful of reads reads little more than it needs -- and batches grow while the reading goes on.
    std::vector<uint8_t> raw, data;               // the blocks' deflate payloads back to back / the inflated bytes of the batch
    struct Blk { uint64_t at; size_t raw_off, raw_len, out_off; uint32_t isize, crc; };
    std::vector<Blk> blks;
    uint64_t next_at = 0;                         // file offset behind the batch
    size_t pos = 0, cur = 0;                      // read position inside `data`, block it lies in
    int grow = 4;
    bool eof = false; std::string err;

    bool open(const char* path) { f = std::fopen(path, "rb"); return f != nullptr; }
    ~Bgzf() { if (f) std::fclose(f); }
    // reads the header and payload of the block at `at`; false + err on a damaged block; eof when the file ends there
    bool read_block(uint64_t at, Blk& k, bool& at_end) {
        at_end = false;
        uint8_t h[18];
        const size_t got = std::fread(h, 1, 18, f);
        if (got == 0) { at_end = true; return true; }
        if (got < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block"; return false; }
        const unsigned xlen = h[10] | (h[11] << 8);
        // the BC subfield is the first one in every file htslib / bgzip writes; walk the extra field all the same
        std::vector<uint8_t> extra(xlen);
        std::memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
        if (xlen > 6 && std::fread(extra.data() + 6, 1, xlen - 6, f) != xlen - 6) { err = "truncated BGZF header"; return false; }
        int bsize = -1;
        for (size_t x = 0; x + 4 <= xlen;) {
            const unsigned slen = extra[x + 2] | (extra[x + 3] << 8);
            if (extra[x] == 'B' && extra[x + 1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = extra[x + 4] | (extra[x + 5] << 8);
            x += 4 + slen;
        }
        if (bsize < 0) { err = "BGZF block without a BC field"; return false; }
        const size_t total = (size_t)bsize + 1, header = 12 + xlen;
        if (total < header + 8) { err = "corrupt BGZF block size"; return false; }
        const size_t body = total - header;
        k.at = at; k.raw_off = raw.size(); k.raw_len = body - 8;
        raw.resize(raw.size() + body);
        uint8_t* dst = raw.data() + k.raw_off;
        const size_t had = xlen < 6 ? 6 - xlen : 0;                   // bytes of the payload already read with the fixed 18
        if (had) std::memcpy(dst, h + 12 + xlen, had);
        if (std::fread(dst + had, 1, body - had, f) != body - had) { err = "truncated BGZF block"; return false; }
        const uint8_t* tail = dst + body - 8;
        k.crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
        k.isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (k.isize > 65536u) { err = "BGZF block claims more than 64 KiB"; return false; }        // (the format's limit: nothing larger is allocated on a file's say-so)
        next_at = at + total;
        return true;
    }
    static const char* inflate_one(const uint8_t* src, size_t n, uint8_t* dst, uint32_t isize, uint32_t crc) {
        if (!isize) return nullptr;
        z_stream z{};
        if (inflateInit2(&z, -15) != Z_OK) return "inflateInit2 failed";
        z.next_in = const_cast<uint8_t*>(src); z.avail_in = (uInt)n; z.next_out = dst; z.avail_out = isize;
        const int rc = inflate(&z, Z_FINISH);
        inflateEnd(&z);
        if (rc != Z_STREAM_END || z.avail_out != 0) return "corrupt BGZF block";
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, isize) != crc) return "BGZF block fails its CRC";
        return nullptr;
    }
    // the batch of up to max_blocks blocks that starts at file offset `at`
    bool load(uint64_t at, int max_blocks) {
        raw.clear(); data.clear(); blks.clear(); pos = 0; cur = 0;
        if (fseeko(f, (off_t)at, SEEK_SET) != 0) { err = "seek failed"; return false; }
        next_at = at;
        size_t out = 0;
        for (int k = 0; k < max_blocks; ++k) {
            Blk b{}; bool at_end = false;
            const uint64_t here = next_at;
            if (!read_block(next_at, b, at_end)) {
                // a damaged or truncated block behind good ones: the batch ends in front of it, and the error is raised only if the reader ever gets there
                // (a region fetch whose records all lie before it must succeed, as it did when blocks were read one at a time)
                if (blks.empty()) return false;
                err.clear(); next_at = here; break;
            }
            if (at_end) { if (blks.empty()) eof = true; break; }
            b.out_off = out; out += b.isize;
            blks.push_back(b);
        }
        data.resize(out);
        const size_t nb = blks.size();
        unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1;
        const size_t nt = std::min<size_t>({ (size_t)hw, (size_t)16, nb / 4 });
        std::atomic<size_t> next_blk(0);
        std::vector<const char*> blk_err(nb, nullptr);
        auto work = [&]() {
            for (;;) {
                const size_t i = next_blk.fetch_add(1);
                if (i >= nb) break;
                blk_err[i] = inflate_one(raw.data() + blks[i].raw_off, blks[i].raw_len, data.data() + blks[i].out_off, blks[i].isize, blks[i].crc);
            }
        };
        if (nt <= 1) work();
        else {
            std::vector<std::thread> th;
            try { for (size_t t = 0; t + 1 < nt; ++t) th.emplace_back(work); } catch (...) { }
            work();
            for (auto& t : th) t.join();
        }
        for (size_t i = 0; i < nb; ++i) if (blk_err[i]) {
            if (i == 0) { err = blk_err[0]; return false; }
            next_at = blks[i].at; data.resize(blks[i].out_off); blks.resize(i);       // (as above: the blocks in front of the bad one are served)
            break;
        }
        return true;
    }
    bool seek(uint64_t voffset) {
        eof = false; grow = 4;
        if (!load(voffset >> 16, grow)) return false;
        const size_t in = (size_t)(voffset & 0xFFFF);
        if (blks.empty()) return in == 0;
        if (in > blks[0].isize) return false;
        pos = in;
        return true;
    }
    uint64_t tell() {
        while (cur + 1 < blks.size() && pos >= blks[cur + 1].out_off) ++cur;
        if (!blks.empty() && pos < data.size()) return (blks[cur].at << 16) | (uint64_t)(pos - blks[cur].out_off);
        return next_at << 16;
    }
    // n bytes, crossing blocks; false at the end of the file (or on an error: err is set)
    bool read(void* out, size_t n) {
        uint8_t* o = (uint8_t*)out;
        while (n) {
            if (pos >= data.size()) {
                if (eof) return false;
                grow = std::min(grow * 4, 256);
                if (!load(next_at, grow)) return false;
                if (eof) return false;
                continue;
            }
            const size_t k = std::min(n, data.size() - pos);
            std::memcpy(o, data.data() + pos, k); o += k; pos += k; n -= k;
        }
        return true;
    }
};

// one text line (without its newline) from the read position on; false at the end of the file (or on an error: err is set)
bool bgzf_getline(Bgzf& z, std::string& line) {
    line.clear();
    for (;;) {
        if (z.pos >= z.data.size()) {
            if (z.eof) return !line.empty();
            z.grow = std::min(z.grow * 4, 256);
            if (!z.load(z.next_at, z.grow)) return false;
            if (z.eof) return !line.empty();
            continue;
        }
        const uint8_t* from = z.data.data() + z.pos;
        const uint8_t* nl = (const uint8_t*)std::memchr(from, '\n', z.data.size() - z.pos);
        if (nl) { line.append((const char*)from, (size_t)(nl - from)); z.pos += (size_t)(nl - from) + 1; return true; }
        line.append((const char*)from, z.data.size() - z.pos); z.pos = z.data.size();
    }
}

uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

// the bins of the BAI scheme that overlap [beg, end) (SAMv1 5.3)
void reg2bins(uint64_t beg, uint64_t end, std::vector<uint32_t>& bins) {
    --end;
    bins.push_back(0);
    for (uint32_t k = 1 + (uint32_t)(beg >> 26); k <= 1 + (uint32_t)(end >> 26); ++k) bins.push_back(k);
    for (uint32_t k = 9 + (uint32_t)(beg >> 23); k <= 9 + (uint32_t)(end >> 23); ++k) bins.push_back(k);
    for (uint32_t k = 73 + (uint32_t)(beg >> 20); k <= 73 + (uint32_t)(end >> 20); ++k) bins.push_back(k);
    for (uint32_t k = 585 + (uint32_t)(beg >> 17); k <= 585 + (uint32_t)(end >> 17); ++k) bins.push_back(k);
    for (uint32_t k = 4681 + (uint32_t)(beg >> 14); k <= 4681 + (uint32_t)(end >> 14); ++k) bins.push_back(k);
}

bool slurp(const char* path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    uint8_t buf[1 << 16]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    std::fclose(f);
    return true;
}

} // namespace

struct sp_bam {
    Bgzf z; std::string err;
    std::vector<std::string> ref_names; std::vector<uint64_t> ref_len; std::vector<const char*> ref_ptr;
    uint64_t first_record = 0;                    // virtual offset of the first alignment
    // index
    bool indexed = false;
    struct RefIndex { std::vector<std::pair<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins; std::vector<uint64_t> linear; };
    std::vector<RefIndex> index;
    std::set<std::string> seen;
    // the last fetch
    std::vector<sp_bam_read> reads; std::vector<std::string> names; std::vector<std::vector<uint32_t>> cigars; std::string bases; std::vector<uint64_t> offsets;
    std::vector<uint8_t> seq4; std::vector<uint64_t> seq4_off; std::vector<uint32_t> seq_len;      // the SEQ fields as stored (4 bits per base)
};

namespace {

int32_t bam_fail(sp_bam* b, const std::string& m) { b->err = m; return SP_ERR_INVALID_ARG; }

bool load_bai(sp_bam* b, const std::string& path) {
    std::vector<uint8_t> d;
    if (!slurp(path.c_str(), d) || d.size() < 8 || std::memcmp(d.data(), "BAI\1", 4) != 0) return false;
    size_t at = 4;
    auto need = [&](size_t n) { return at + n <= d.size(); };
    if (!need(4)) return false;
    const uint32_t n_ref = le32(d.data() + at); at += 4;
    if (n_ref > (1u << 24)) return false;
    b->index.assign(n_ref, {});
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!need(4)) return false;
        const uint32_t n_bin = le32(d.data() + at); at += 4;
        for (uint32_t k = 0; k < n_bin; ++k) {
            if (!need(8)) return false;
            const uint32_t bin = le32(d.data() + at), n_chunk = le32(d.data() + at + 4); at += 8;
            if (!need((size_t)n_chunk * 16)) return false;
            std::vector<std::pair<uint64_t, uint64_t>> chunks(n_chunk);
            for (uint32_t c = 0; c < n_chunk; ++c) { chunks[c] = { le64(d.data() + at), le64(d.data() + at + 8) }; at += 16; }
            if (bin != 37450) b->index[r].bins.emplace_back(bin, std::move(chunks));          // (37450: the metadata pseudo-bin)
        }
        if (!need(4)) return false;
        const uint32_t n_intv = le32(d.data() + at); at += 4;
        if (!need((size_t)n_intv * 8)) return false;
        b->index[r].linear.resize(n_intv);
        for (uint32_t k = 0; k < n_intv; ++k) { b->index[r].linear[k] = le64(d.data() + at); at += 8; }
    }
    return true;
}

} // namespace

extern "C" {

int32_t sp_bam_open(const char* path, sp_bam** out, char* err, uint32_t err_cap) {
    if (!path || !out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    auto b = std::make_unique<sp_bam>();
    if (!b->z.open(path)) { put_err(err, err_cap, std::string("cannot open ") + path); return SP_ERR_INVALID_ARG; }
    auto bad = [&](const std::string& m) { put_err(err, err_cap, m.empty() ? "truncated BAM header" : m); return SP_ERR_INVALID_ARG; };
    if (!b->z.load(0, 4)) return bad(b->z.err);
    uint8_t h[12];
    if (!b->z.read(h, 8) || std::memcmp(h, "BAM\1", 4) != 0) return bad(b->z.err.empty() ? "not a BAM file" : b->z.err);
    const uint32_t l_text = le32(h + 4);
    if (l_text > (1u << 30)) return bad("corrupt BAM header");
    std::vector<uint8_t> skip(l_text);
    if (l_text && !b->z.read(skip.data(), l_text)) return bad(b->z.err);
    if (!b->z.read(h, 4)) return bad(b->z.err);
    const uint32_t n_ref = le32(h);
    if (n_ref > (1u << 24)) return bad("corrupt BAM header");
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!b->z.read(h, 4)) return bad(b->z.err);
        const uint32_t l_name = le32(h);
        if (l_name > (1u << 16)) return bad("corrupt BAM header");
        std::string name(l_name, '\0');
        if (l_name && !b->z.read(&name[0], l_name)) return bad(b->z.err);
        while (!name.empty() && name.back() == '\0') name.pop_back();
        if (!b->z.read(h, 4)) return bad(b->z.err);
        b->ref_names.push_back(name); b->ref_len.push_back(le32(h));
    }
    for (const std::string& s : b->ref_names) b->ref_ptr.push_back(s.c_str());
    b->first_record = b->z.tell();
    const std::string p(path);
    b->indexed = load_bai(b.get(), p + ".bai");
    if (!b->indexed && p.size() > 4 && p.compare(p.size() - 4, 4, ".bam") == 0) b->indexed = load_bai(b.get(), p.substr(0, p.size() - 4) + ".bai");
    if (b->indexed && b->index.size() != b->ref_names.size()) { b->indexed = false; b->index.clear(); }
    *out = b.release();
    return SP_OK;
}

void sp_bam_free(sp_bam* bam) { delete bam; }
const char* sp_bam_last_error(const sp_bam* bam) { return bam ? bam->err.c_str() : ""; }
int32_t sp_bam_last_seq4(const sp_bam* bam, const uint8_t** seq4, const uint64_t** byte_offsets, const uint32_t** lengths, uint32_t* n) {
    if (!bam || !seq4 || !byte_offsets || !lengths) return SP_ERR_INVALID_ARG;
    *seq4 = bam->seq4.data(); *byte_offsets = bam->seq4_off.data(); *lengths = bam->seq_len.data();
    if (n) *n = (uint32_t)bam->seq_len.size();
    return SP_OK;
}
int32_t sp_bam_forget(sp_bam* bam) { if (!bam) return SP_ERR_INVALID_ARG; bam->seen.clear(); return SP_OK; }

int32_t sp_bam_references(const sp_bam* bam, uint32_t* n, const char* const** names, const uint64_t** lengths) {
    if (!bam || !n) return SP_ERR_INVALID_ARG;
    *n = (uint32_t)bam->ref_names.size();
    if (names) *names = bam->ref_ptr.data();
    if (lengths) *lengths = bam->ref_len.data();
    return SP_OK;
}

int32_t sp_bam_fetch(sp_bam* b, const char* chrom, uint64_t start, uint64_t end, uint32_t exclude_flags, int32_t dedupe,
                     const sp_bam_read** reads, uint32_t* n, const char** bases, const uint64_t** offsets) {
    if (!b || !chrom || !reads || !n || end <= start) return SP_ERR_INVALID_ARG;
    b->reads.clear(); b->names.clear(); b->cigars.clear(); b->bases.clear(); b->offsets.assign(1, 0);
    b->seq4.clear(); b->seq4_off.assign(1, 0); b->seq_len.clear();
    *reads = nullptr; *n = 0;
    int ref = -1;
    for (size_t r = 0; r < b->ref_names.size(); ++r) if (b->ref_names[r] == chrom) ref = (int)r;
    if (ref < 0) return bam_fail(b, std::string("the BAM header has no reference ") + chrom);
    // where to read: the chunks of the index that can hold overlapping records, or everything
    std::vector<std::pair<uint64_t, uint64_t>> chunks;
    if (b->indexed) {
        const sp_bam::RefIndex& ix = b->index[(size_t)ref];
        std::vector<uint32_t> bins; reg2bins(start, end, bins);
        const size_t win = (size_t)(start >> 14);
        const uint64_t min_off = ix.linear.empty() ? 0 : ix.linear[std::min(win, ix.linear.size() - 1)];
        for (const auto& bin : ix.bins) if (std::find(bins.begin(), bins.end(), bin.first) != bins.end())
            for (const auto& c : bin.second) if (c.second > min_off) chunks.emplace_back(std::max(c.first, min_off), c.second);
        std::sort(chunks.begin(), chunks.end());
        std::vector<std::pair<uint64_t, uint64_t>> merged;
        for (const auto& c : chunks) { if (!merged.empty() && c.first <= merged.back().second) merged.back().second = std::max(merged.back().second, c.second); else merged.push_back(c); }
        chunks.swap(merged);
    } else chunks.emplace_back(b->first_record, UINT64_MAX);
    static const char decode[] = "=ACMGRSVTWYHKDBN";
    std::vector<uint8_t> rec;
    bool past = false;                        // (chunks are visited in file order and the file is sorted: a record behind the region ends the fetch, not only its chunk -- htslib's iterator does the same)
    for (const auto& chunk : chunks) {
        if (past) break;
        if (!b->z.seek(chunk.first)) return bam_fail(b, b->z.err.empty() ? "bad virtual offset in the index" : b->z.err);
        while (!past && b->z.tell() < chunk.second) {
            uint8_t h4[4];
            if (!b->z.read(h4, 4)) { if (!b->z.err.empty()) return bam_fail(b, b->z.err); break; }
            const uint32_t block_size = le32(h4);
            if (block_size < 32 || block_size > (1u << 28)) return bam_fail(b, "corrupt BAM record");
            rec.resize(block_size);
            if (!b->z.read(rec.data(), block_size)) return bam_fail(b, b->z.err.empty() ? "truncated BAM record" : b->z.err);
            const int32_t ref_id = (int32_t)le32(rec.data()); const int64_t pos = (int32_t)le32(rec.data() + 4);
            const uint32_t l_name = rec[8], mapq = rec[9], n_cigar = rec[12] | (rec[13] << 8), flag = rec[14] | (rec[15] << 8), l_seq = le32(rec.data() + 16);
            if (32ull + l_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq > block_size) return bam_fail(b, "corrupt BAM record");
            // an indexed file is coordinate-sorted: past the region once a record of this reference starts behind it or a later reference
            // begins (a file without index is scanned to its end: it need not be sorted)
            if (b->indexed && ((ref_id == ref && pos >= (int64_t)end) || ref_id > ref || ref_id < 0)) { past = true; break; }
            if (ref_id != ref) continue;
            const uint8_t* cg = rec.data() + 32 + l_name;
            int64_t span = 0;
            std::vector<uint32_t> cigar(n_cigar);
            for (uint32_t c = 0; c < n_cigar; ++c) { cigar[c] = le32(cg + 4 * c); const uint32_t op = cigar[c] & 0xF; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += cigar[c] >> 4; }
            const int64_t rend = pos + (span > 0 ? span : 1);
            if (!(pos < (int64_t)end && rend > (int64_t)start)) continue;
            if (flag & exclude_flags) continue;
            std::string qname((const char*)rec.data() + 32, l_name ? l_name - 1 : 0);
            if (dedupe && !b->seen.insert(qname).second) continue;
            const uint8_t* sq = cg + 4 * n_cigar;
            for (uint32_t x = 0; x < l_seq; ++x) b->bases.push_back(decode[(sq[x >> 1] >> ((x & 1) ? 0 : 4)) & 0xF]);
            b->offsets.push_back(b->bases.size());
            b->seq4.insert(b->seq4.end(), sq, sq + (l_seq + 1) / 2); b->seq4_off.push_back(b->seq4.size()); b->seq_len.push_back(l_seq);
            b->names.push_back(std::move(qname)); b->cigars.push_back(std::move(cigar));
            sp_bam_read r{}; r.flag = flag; r.mapq = mapq; r.ref_id = ref_id; r.pos = pos; r.end = rend; r.l_seq = l_seq; r.n_cigar = n_cigar;
            b->reads.push_back(r);
        }
    }
    for (size_t i = 0; i < b->reads.size(); ++i) { b->reads[i].qname = b->names[i].c_str(); b->reads[i].cigar = b->cigars[i].empty() ? nullptr : b->cigars[i].data(); }
    *reads = b->reads.data(); *n = (uint32_t)b->reads.size();
    if (bases) *bases = b->bases.c_str();
    if (offsets) *offsets = b->offsets.data();
    return SP_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------ VCF
struct sp_fasta {
    struct Seq { std::string name; uint64_t length = 0, offset = 0, line_bases = 0, line_bytes = 0; std::string bases; };
    std::string path, err, slice;
    std::vector<Seq> seqs; std::vector<const char*> name_ptr; std::vector<uint64_t> lengths;
    bool indexed = false; FILE* file = nullptr;
};

struct sp_vcf {
    std::string err;
    std::vector<std::string> samples; std::vector<const char*> sample_ptr;
    struct Record { std::string chrom; uint64_t pos0; std::string ref; std::vector<std::string> alts; std::string info, format; std::vector<std::string> calls; };
    std::vector<Record> records;                  // linear mode: every record of the file; indexed mode: the records of the last region fetch
    std::vector<sp_vcf_allele> alleles; std::vector<sp_vcf_deletion> deletions;
    // indexed mode (a BGZF file with a .tbi / .csi beside it): only the header is read at open; a query reads the chunks the index names
    // (bcf::IndexedReader::fetch, src/diplotyper.rs:569-575,800)
    bool indexed = false; Bgzf z;
    int min_shift = 14, depth = 5;
    struct RefIndex { std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::map<uint32_t, uint64_t> loffset; std::vector<uint64_t> linear; };
    std::map<std::string, RefIndex> index;
    uint64_t n_fetched_lines = 0;                 // lines parsed by region fetches so far (the tests' evidence that a fetch does not scan the file)
};

namespace {

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out; size_t from = 0;
    for (;;) { const size_t at = s.find(sep, from); out.push_back(s.substr(from, at == std::string::npos ? at : at - from)); if (at == std::string::npos) break; from = at + 1; }
    return out;
}

int32_t vcf_fail(sp_vcf* v, const std::string& m) { v->err = m; return SP_ERR_INVALID_ARG; }

int sample_index(sp_vcf* v, const char* sample) {
    if (!sample) return v->samples.empty() ? -1 : 0;
    for (size_t i = 0; i < v->samples.size(); ++i) if (v->samples[i] == sample) return (int)i;
    return -1;
}

// GT and PS of one sample column: false when the genotype is missing or not diploid
bool genotype(const sp_vcf::Record& r, int si, int& g1, int& g2, bool& phased, int64_t& ps) {
    if ((size_t)si >= r.calls.size()) return false;
    const std::vector<std::string> keys = split(r.format, ':'), vals = split(r.calls[(size_t)si], ':');
    std::string gt, psv = ".";
    for (size_t k = 0; k < keys.size() && k < vals.size(); ++k) { if (keys[k] == "GT") gt = vals[k]; else if (keys[k] == "PS") psv = vals[k]; }
    phased = gt.find('|') != std::string::npos;
    std::string norm = gt; std::replace(norm.begin(), norm.end(), '|', '/');
    const std::vector<std::string> a = split(norm, '/');
    if (a.size() != 2 || a[0].empty() || a[1].empty() || a[0] == "." || a[1] == ".") return false;
    g1 = std::atoi(a[0].c_str()); g2 = std::atoi(a[1].c_str());
    ps = (phased && psv != "." && !psv.empty()) ? std::atoll(psv.c_str()) : -1;
    return true;
}

// one record line -> Record; false + message on a malformed line
bool vcf_parse_record(const std::string& line, sp_vcf::Record& r, std::string& why) {
    const std::vector<std::string> col = split(line, '\t');
    if (col.size() < 8) { why = "VCF record with fewer than 8 columns"; return false; }
    r = sp_vcf::Record();
    r.chrom = col[0]; r.pos0 = (uint64_t)std::strtoull(col[1].c_str(), nullptr, 10) - 1; r.ref = col[3]; r.alts = split(col[4], ','); r.info = col[7];
    if (col.size() > 8) r.format = col[8];
    for (size_t c = 9; c < col.size(); ++c) r.calls.push_back(col[c]);
    return true;
}

// a whole gzip / BGZF file inflated (the index files are BGZF themselves)
bool inflate_file(const std::string& path, std::vector<uint8_t>& out) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t buf[1 << 16]; int k;
    while ((k = gzread(f, buf, sizeof buf)) > 0) out.insert(out.end(), buf, buf + k);
    gzclose(f);
    return k == 0;
}

// tabix (.tbi, SAM/VCF specification "The tabix index format") and CSI (.csi, CSIv1) indices of a BGZF text file.  Both give, per reference name, bins of
// chunks (pairs of virtual offsets); the bin scheme is UCSC's with min_shift / depth (14 / 5 in a .tbi).  Bin 37450 of a .tbi (depth-5 scheme + 1) is metadata.
bool load_vcf_index(sp_vcf* v, const std::string& path) {
    std::vector<uint8_t> d;
    bool csi = false;
    if (inflate_file(path + ".tbi", d) && d.size() >= 36 && std::memcmp(d.data(), "TBI\1", 4) == 0) csi = false;
    else { d.clear(); if (inflate_file(path + ".csi", d) && d.size() >= 16 && std::memcmp(d.data(), "CSI\1", 4) == 0) csi = true; else return false; }
    size_t at = 4;
    auto need = [&](size_t n) { return at + n <= d.size(); };
    auto i32 = [&]() { const int32_t x = (int32_t)le32(d.data() + at); at += 4; return x; };
    int32_t n_ref = 0; std::vector<std::string> names;
    auto read_names = [&](size_t l_nm) {
        if (!need(l_nm)) return false;
        size_t from = at; const size_t stop = at + l_nm;
        while (from < stop) { const void* z = std::memchr(d.data() + from, 0, stop - from); const size_t to = z ? (size_t)((const uint8_t*)z - d.data()) : stop; names.emplace_back((const char*)d.data() + from, to - from); from = to + 1; }
        at = stop;
        return true;
    };
    if (!csi) {
        if (!need(32)) return false;
        n_ref = i32(); at += 24;                                   // format, col_seq, col_beg, col_end, meta, skip
        const int32_t l_nm = i32();
        if (l_nm < 0 || !read_names((size_t)l_nm)) return false;
        v->min_shift = 14; v->depth = 5;
    } else {
        if (!need(12)) return false;
        v->min_shift = i32(); v->depth = i32(); const int32_t l_aux = i32();
        if (v->min_shift < 1 || v->min_shift > 30 || v->depth < 1 || v->depth > 10 || l_aux < 0 || !need((size_t)l_aux)) return false;
        const size_t aux_end = at + (size_t)l_aux;
        if (l_aux >= 28) { at += 24; const int32_t l_nm = i32(); if (l_nm < 0 || at + (size_t)l_nm > aux_end || !read_names((size_t)l_nm)) return false; }
        at = aux_end;
        if (!need(4)) return false;
        n_ref = i32();
    }
    if (n_ref < 0 || n_ref > (1 << 24) || (size_t)n_ref > names.size()) return false;
    const uint32_t meta_bin = csi ? 0xFFFFFFFFu : 37450u;
    for (int32_t r = 0; r < n_ref; ++r) {
        sp_vcf::RefIndex ix;
        if (!need(4)) return false;
        const int32_t n_bin = i32();
        if (n_bin < 0) return false;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (!need(csi ? 16 : 8)) return false;
            const uint32_t bin = le32(d.data() + at); at += 4;
            if (csi) { ix.loffset[bin] = le64(d.data() + at); at += 8; }
            const int32_t n_chunk = i32();
            if (n_chunk < 0 || !need((size_t)n_chunk * 16)) return false;
            std::vector<std::pair<uint64_t, uint64_t>> chunks((size_t)n_chunk);
            for (int32_t c = 0; c < n_chunk; ++c) { chunks[(size_t)c] = { le64(d.data() + at), le64(d.data() + at + 8) }; at += 16; }
            if (bin != meta_bin) ix.bins[bin] = std::move(chunks);
        }
        if (!csi) {
            if (!need(4)) return false;
            const int32_t n_intv = i32();
            if (n_intv < 0 || !need((size_t)n_intv * 8)) return false;
            ix.linear.resize((size_t)n_intv);
            for (int32_t k = 0; k < n_intv; ++k) { ix.linear[(size_t)k] = le64(d.data() + at); at += 8; }
        }
        v->index[names[(size_t)r]] = std::move(ix);
    }
    return true;
}

// the records of chrom that may overlap [start, end), read through the index into v->records
int32_t vcf_fetch(sp_vcf* v, const char* chrom, uint64_t start, uint64_t end) {
    v->records.clear();
    const auto it = v->index.find(chrom);
    if (it == v->index.end() || end <= start) return SP_OK;         // (a chromosome without records is not in the index)
    const sp_vcf::RefIndex& ix = it->second;
    std::vector<std::pair<uint64_t, uint64_t>> chunks;
    {
        const uint64_t cap = 1ull << (v->min_shift + 3 * v->depth);
        const uint64_t b0 = std::min(start, cap - 1), e0 = std::min(end, cap) - 1;
        uint64_t t = 0; int s = v->min_shift + 3 * v->depth;
        for (int l = 0; l <= v->depth; ++l, s -= 3) {
            for (uint64_t k = t + (b0 >> s); k <= t + (e0 >> s); ++k) { const auto b = ix.bins.find((uint32_t)k); if (b != ix.bins.end()) chunks.insert(chunks.end(), b->second.begin(), b->second.end()); }
            t += 1ull << (3 * l);
        }
    }
    const uint64_t min_off = ix.linear.empty() ? 0 : ix.linear[std::min<size_t>((size_t)(start >> 14), ix.linear.size() - 1)];
    std::vector<std::pair<uint64_t, uint64_t>> kept;
    for (const auto& c : chunks) if (c.second > min_off) kept.emplace_back(std::max(c.first, min_off), c.second);
    std::sort(kept.begin(), kept.end());
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    for (const auto& c : kept) { if (!merged.empty() && c.first <= merged.back().second) merged.back().second = std::max(merged.back().second, c.second); else merged.push_back(c); }
    std::string line, why;
    bool past = false;
    for (const auto& c : merged) {
        if (past) break;
        if (!v->z.seek(c.first)) return vcf_fail(v, v->z.err.empty() ? "bad virtual offset in the VCF index" : v->z.err);
        while (!past && v->z.tell() < c.second) {
            if (!bgzf_getline(v->z, line)) { if (!v->z.err.empty()) return vcf_fail(v, v->z.err); break; }
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line[0] == '#') continue;
            v->n_fetched_lines += 1;
            sp_vcf::Record r;
            if (!vcf_parse_record(line, r, why)) return vcf_fail(v, why);
            if (r.chrom != chrom) continue;
            if (r.pos0 >= end) { past = true; break; }              // sorted: nothing behind it starts inside the region
            v->records.push_back(std::move(r));
        }
    }
    return SP_OK;
}

} // namespace

extern "C" {

int32_t sp_vcf_open(const char* path, sp_vcf** out, char* err, uint32_t err_cap) {
    if (!path || !out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    // a BGZF file with a tabix / CSI index beside it: header only, records come through the index query by query
    try {
        auto v = std::make_unique<sp_vcf>();
        if (load_vcf_index(v.get(), path) && v->z.open(path) && v->z.seek(0)) {
            std::string line; bool have_header = false;
            while (bgzf_getline(v->z, line)) {
                if (!line.empty() && line.back() == '\r') line.pop_back();
                if (line.empty() || line.compare(0, 2, "##") == 0) continue;
                if (line[0] != '#') break;
                const std::vector<std::string> col = split(line, '\t');
                for (size_t c = 9; c < col.size(); ++c) v->samples.push_back(col[c]);
                have_header = true; break;
            }
            if (v->z.err.empty() && have_header) {
                for (const std::string& s : v->samples) v->sample_ptr.push_back(s.c_str());
                v->indexed = true;
                *out = v.release();
                return SP_OK;
            }
        }
    } catch (const std::bad_alloc&) { put_err(err, err_cap, "out of memory reading the VCF index"); return SP_ERR_OUT_OF_MEMORY; }
    catch (const std::exception&) { }                // (an index that cannot be used: the file is read as a whole)
    gzFile f = gzopen(path, "rb");                 // plain text, gzip and BGZF (gzip members one after the other) alike
    if (!f) { put_err(err, err_cap, std::string("cannot open ") + path); return SP_ERR_INVALID_ARG; }
    std::string text; char buf[1 << 16]; int k;
    while ((k = gzread(f, buf, sizeof buf)) > 0) text.append(buf, (size_t)k);
    const bool bad = k < 0;
    gzclose(f);
    if (bad) { put_err(err, err_cap, std::string("cannot read ") + path); return SP_ERR_INVALID_ARG; }
    try {
    auto v = std::make_unique<sp_vcf>();
    size_t from = 0; bool have_header = false;
    while (from < text.size()) {
        size_t to = text.find('\n', from); if (to == std::string::npos) to = text.size();
        std::string line = text.substr(from, to - from); from = to + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line.compare(0, 2, "##") == 0) continue;
        if (line[0] == '#') { const std::vector<std::string> col = split(line, '\t'); for (size_t c = 9; c < col.size(); ++c) v->samples.push_back(col[c]); have_header = true; continue; }
        sp_vcf::Record r; std::string why;
        if (!vcf_parse_record(line, r, why)) { put_err(err, err_cap, why); return SP_ERR_INVALID_ARG; }
        v->records.push_back(std::move(r));
    }
    if (!have_header) { put_err(err, err_cap, "no #CHROM header line"); return SP_ERR_INVALID_ARG; }
    for (const std::string& s : v->samples) v->sample_ptr.push_back(s.c_str());
    *out = v.release();
    return SP_OK;
    } catch (const std::bad_alloc&) { put_err(err, err_cap, "out of memory reading the VCF"); return SP_ERR_OUT_OF_MEMORY; }
    catch (const std::exception& e) { put_err(err, err_cap, std::string("VCF: ") + e.what()); return SP_ERR_INVALID_ARG; }
}

void sp_vcf_free(sp_vcf* vcf) { delete vcf; }
int32_t sp_vcf_index_info(const sp_vcf* vcf, int32_t* indexed, uint64_t* lines_parsed_by_fetches) {
    if (!vcf) return SP_ERR_INVALID_ARG;
    if (indexed) *indexed = vcf->indexed ? 1 : 0;
    if (lines_parsed_by_fetches) *lines_parsed_by_fetches = vcf->n_fetched_lines;
    return SP_OK;
}
const char* sp_vcf_last_error(const sp_vcf* vcf) { return vcf ? vcf->err.c_str() : ""; }
int32_t sp_vcf_samples(const sp_vcf* vcf, uint32_t* n, const char* const** names) {
    if (!vcf || !n) return SP_ERR_INVALID_ARG;
    *n = (uint32_t)vcf->samples.size();
    if (names) *names = vcf->sample_ptr.data();
    return SP_OK;
}

int32_t sp_vcf_alleles(sp_vcf* v, const char* sample, const char* chrom, uint64_t start, uint64_t end, const sp_vcf_allele** out, uint32_t* n) {
    if (!v || !chrom || !out || !n) return SP_ERR_INVALID_ARG;
    v->alleles.clear(); *out = nullptr; *n = 0;
    const int si = sample_index(v, sample);
    if (si < 0) return vcf_fail(v, std::string("the VCF has no sample ") + (sample ? sample : "(first)"));
    if (v->indexed) { const int32_t rc = vcf_fetch(v, chrom, start, end); if (rc != SP_OK) return rc; }
    for (const sp_vcf::Record& r : v->records) {
        if (r.chrom != chrom || !(r.pos0 < end && r.pos0 + r.ref.size() > start)) continue;
        int g1, g2; bool phased; int64_t ps;
        if (!genotype(r, si, g1, g2, phased, ps)) continue;
        for (size_t a = 0; a < r.alts.size(); ++a) {
            const int ai = (int)a + 1;
            int32_t gt = SP_GT_HOM_REF;
            if (ai == g1 && ai == g2) gt = SP_GT_HOM_ALT;
            else if (ai == g1 && phased) gt = SP_GT_HET_FLIP;
            else if (ai == g2 && phased) gt = SP_GT_HET_PHASED;
            else if (ai == g1 || ai == g2) gt = SP_GT_HET_UNPHASED;
            v->alleles.push_back(sp_vcf_allele{ r.pos0, r.ref.c_str(), r.alts[a].c_str(), gt, 0, ps });
        }
    }
    *out = v->alleles.data(); *n = (uint32_t)v->alleles.size();
    return SP_OK;
}

int32_t sp_vcf_deletions(sp_vcf* v, const char* sample, const char* chrom, uint64_t start, uint64_t end, const sp_vcf_deletion** out, uint32_t* n) {
    if (!v || !chrom || !out || !n) return SP_ERR_INVALID_ARG;
    v->deletions.clear(); *out = nullptr; *n = 0;
    const int si = sample_index(v, sample);
    if (si < 0) return vcf_fail(v, std::string("the VCF has no sample ") + (sample ? sample : "(first)"));
    if (v->indexed) { const int32_t rc = vcf_fetch(v, chrom, start, end); if (rc != SP_OK) return rc; }
    for (const sp_vcf::Record& r : v->records) {
        if (r.chrom != chrom || r.alts.size() != 1) continue;
        if (r.pos0 >= end) continue;                                 // the reference only sees what its region fetch returns (src/diplotyper.rs:796-815)
        std::string svtype, endv; bool has_type = false, has_end = false;
        for (const std::string& kv : split(r.info, ';')) {
            if (kv.compare(0, 7, "SVTYPE=") == 0) { svtype = kv.substr(7); has_type = true; }
            else if (kv.compare(0, 4, "END=") == 0) { endv = kv.substr(4); has_end = true; }
        }
        // a record's extent for the fetch is INFO/END when it has one, its REF allele otherwise (htslib's rlen)
        const uint64_t e = has_end ? (uint64_t)std::strtoull(endv.c_str(), nullptr, 10) : r.pos0 + r.ref.size();
        if (!(e > start)) continue;                                  // outside the region: never fetched, its INFO fields are never looked at
        if (!has_type) return vcf_fail(v, "No INFO:SVTYPE in record");
        if (svtype != "DEL") continue;
        if (!has_end) return vcf_fail(v, "No INFO:END in record");
        int g1, g2; bool phased; int64_t ps;
        if (!genotype(r, si, g1, g2, phased, ps)) continue;
        int32_t gt;
        if (g1 == g2) gt = g1 == 0 ? SP_GT_HOM_REF : SP_GT_HOM_ALT;
        else if (phased) gt = g1 == 0 ? SP_GT_HET_PHASED : SP_GT_HET_FLIP;
        else gt = SP_GT_HET_UNPHASED;
        v->deletions.push_back(sp_vcf_deletion{ r.pos0, e, gt, 0, ps });
    }
    *out = v->deletions.data(); *n = (uint32_t)v->deletions.size();
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------------------------ reference FASTA
// What the reference gets from ReferenceGenome::from_fasta / get_slice (rust-lib-reference-genome; src/cli/diplotype.rs loads the file
// once, the callers take 0-based half-open slices: HlaRealigner::new src/hla/realigner.rs:74-81, Cyp2d6Extractor::new
// src/cyp2d6/haplotyper.rs:45-132, load_database_haplotypes src/diplotyper.rs:437-548).  A plain file with a "<path>.fai" next to it is
// read slice by slice through the index (name, length, offset, bases per line, bytes per line); anything else (no index, gzip / BGZF) is
// read into memory once.  Bases are handed out upper-cased.
int32_t sp_fasta_open(const char* path, sp_fasta** out, char* err, uint32_t err_cap) {
    if (!path || !out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    auto fa = std::make_unique<sp_fasta>();
    fa->path = path;
    // gzip magic?
    bool gz = false;
    { FILE* f = std::fopen(path, "rb"); if (!f) { put_err(err, err_cap, std::string("cannot open ") + path); return SP_ERR_INVALID_ARG; }
      unsigned char m[2] = { 0, 0 }; gz = std::fread(m, 1, 2, f) == 2 && m[0] == 0x1f && m[1] == 0x8b; std::fclose(f); }
    FILE* idx = gz ? nullptr : std::fopen((fa->path + ".fai").c_str(), "rb");
    if (idx) {
        char line[4096];
        while (std::fgets(line, sizeof line, idx)) {
            std::string l(line); while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
            if (l.empty()) continue;
            const std::vector<std::string> col = split(l, '\t');
            if (col.size() < 5) { std::fclose(idx); put_err(err, err_cap, "FASTA index line with fewer than 5 columns"); return SP_ERR_INVALID_ARG; }
            sp_fasta::Seq q; q.name = col[0]; q.length = std::strtoull(col[1].c_str(), nullptr, 10); q.offset = std::strtoull(col[2].c_str(), nullptr, 10);
            q.line_bases = std::strtoull(col[3].c_str(), nullptr, 10); q.line_bytes = std::strtoull(col[4].c_str(), nullptr, 10);
            if (q.line_bases == 0 || q.line_bytes < q.line_bases) { std::fclose(idx); put_err(err, err_cap, "FASTA index line with an impossible line width"); return SP_ERR_INVALID_ARG; }
            fa->seqs.push_back(std::move(q));
        }
        std::fclose(idx);
        fa->file = std::fopen(path, "rb");
        if (!fa->file) { put_err(err, err_cap, std::string("cannot open ") + path); return SP_ERR_INVALID_ARG; }
        fa->indexed = true;
    } else {
        gzFile f = gzopen(path, "rb");
        if (!f) { put_err(err, err_cap, std::string("cannot open ") + path); return SP_ERR_INVALID_ARG; }
        std::string text; char buf[1 << 16]; int k;
        while ((k = gzread(f, buf, sizeof buf)) > 0) text.append(buf, (size_t)k);
        const bool bad = k < 0;
        gzclose(f);
        if (bad) { put_err(err, err_cap, std::string("cannot read ") + path); return SP_ERR_INVALID_ARG; }
        size_t from = 0;
        while (from < text.size()) {
            size_t to = text.find('\n', from); if (to == std::string::npos) to = text.size();
            size_t stop = to; if (stop > from && text[stop - 1] == '\r') --stop;
            if (stop > from && text[from] == '>') {
                sp_fasta::Seq q; size_t e = from + 1; while (e < stop && text[e] != ' ' && text[e] != '\t') ++e;      // the name ends at the first blank
                q.name = text.substr(from + 1, e - from - 1);
                fa->seqs.push_back(std::move(q));
            } else if (stop > from) {
                if (fa->seqs.empty()) { put_err(err, err_cap, "FASTA bases before the first header line"); return SP_ERR_INVALID_ARG; }
                std::string& b = fa->seqs.back().bases;
                for (size_t i = from; i < stop; ++i) { const char c = text[i]; b += (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }
            }
            from = to + 1;
        }
        for (auto& q : fa->seqs) q.length = q.bases.size();
    }
    for (auto& q : fa->seqs) { fa->name_ptr.push_back(q.name.c_str()); fa->lengths.push_back(q.length); }
    *out = fa.release();
    return SP_OK;
}

void sp_fasta_free(sp_fasta* fa) { if (fa) { if (fa->file) std::fclose(fa->file); delete fa; } }
const char* sp_fasta_last_error(const sp_fasta* fa) { return fa ? fa->err.c_str() : ""; }

int32_t sp_fasta_sequences(sp_fasta* fa, uint32_t* n, const char* const** names, const uint64_t** lengths) {
    if (!fa || !n) return SP_ERR_INVALID_ARG;
    *n = (uint32_t)fa->seqs.size();
    if (names) *names = fa->name_ptr.data();
    if (lengths) *lengths = fa->lengths.data();
    return SP_OK;
}

// get_slice(chrom, start, end): 0-based half-open; the slice has to lie inside the sequence
int32_t sp_fasta_fetch(sp_fasta* fa, const char* chrom, uint64_t start, uint64_t end, const char** bases, uint64_t* len) {
    if (!fa || !chrom || !bases) return SP_ERR_INVALID_ARG;
    *bases = nullptr; if (len) *len = 0;
    const sp_fasta::Seq* q = nullptr;
    for (const auto& s : fa->seqs) if (s.name == chrom) { q = &s; break; }
    if (!q) { fa->err = std::string("the FASTA has no sequence ") + chrom; return SP_ERR_INVALID_ARG; }
    if (start > end || end > q->length) { fa->err = std::string("slice outside of ") + chrom; return SP_ERR_INVALID_ARG; }
    try {
    if (!fa->indexed) fa->slice.assign(q->bases, (size_t)start, (size_t)(end - start));
    else {
        fa->slice.clear(); fa->slice.reserve((size_t)(end - start));
        const uint64_t first = q->offset + (start / q->line_bases) * q->line_bytes + start % q->line_bases;
        const uint64_t last = end == start ? first : q->offset + ((end - 1) / q->line_bases) * q->line_bytes + (end - 1) % q->line_bases + 1;
        // what the index says is held against the file before anything of that size is allocated (a damaged .fai can name any length)
        if (fseeko(fa->file, 0, SEEK_END) != 0) { fa->err = "cannot read " + fa->path; return SP_ERR_INVALID_ARG; }
        const uint64_t file_size = (uint64_t)ftello(fa->file);
        if (last < first || last > file_size) { fa->err = "the FASTA index does not describe " + fa->path; return SP_ERR_INVALID_ARG; }
        std::string raw((size_t)(last - first), '\0');
        if (fseeko(fa->file, (off_t)first, SEEK_SET) != 0 || std::fread(&raw[0], 1, raw.size(), fa->file) != raw.size()) { fa->err = "cannot read " + fa->path; return SP_ERR_INVALID_ARG; }
        for (char c : raw) if (c != '\n' && c != '\r') fa->slice += (c >= 'a' && c <= 'z') ? (char)(c - 32) : c;
        if (fa->slice.size() != end - start) { fa->err = "the FASTA index does not describe " + fa->path; return SP_ERR_INVALID_ARG; }
    }
    } catch (const std::bad_alloc&) { fa->err = "out of memory reading " + fa->path; return SP_ERR_OUT_OF_MEMORY; }
    catch (const std::exception& e) { fa->err = std::string("FASTA: ") + e.what(); return SP_ERR_INVALID_ARG; }
    *bases = fa->slice.c_str();
    if (len) *len = fa->slice.size();
    return SP_OK;
}

} // extern "C"
