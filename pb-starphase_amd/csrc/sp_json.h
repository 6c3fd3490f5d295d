// sp_json.h -- a small JSON document reader / writer for the database loader and the result writer (host only).
// Objects keep their members in file order (serde_json's BTreeMap-backed types are re-sorted by the callers that need key order).
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <algorithm>
#include <string>
#include <vector>

namespace spj {

struct Value {
    enum Kind { Null, Bool, Int, Double, String, Array, Object } kind = Null;
    bool b = false; int64_t i = 0; double d = 0.0; std::string s;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;
    std::vector<uint32_t> by_key;                 // large objects: member numbers sorted by (key, position), built when the object has been read
    // the member of that name; of members with one name the LAST counts, as in the maps serde fills (a later insert replaces an earlier one)
    const Value* get(const char* key) const {
        if (kind != Object) return nullptr;
        if (by_key.size() == obj.size() && !obj.empty()) {
            size_t lo = 0, hi = by_key.size();                                   // first member whose key is greater than `key`
            while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (obj[by_key[mid]].first.compare(key) <= 0) lo = mid + 1; else hi = mid; }
            return (lo > 0 && obj[by_key[lo - 1]].first == key) ? &obj[by_key[lo - 1]].second : nullptr;
        }
        const Value* found = nullptr;
        for (const auto& kv : obj) if (kv.first == key) found = &kv.second;
        return found;
    }
    void index_members() {                        // (a linear scan is as good for the few members of an ordinary object)
        if (kind != Object || obj.size() < 32) return;
        by_key.resize(obj.size());
        for (uint32_t k = 0; k < (uint32_t)obj.size(); ++k) by_key[k] = k;
        std::sort(by_key.begin(), by_key.end(), [&](uint32_t a, uint32_t b) { const int c = obj[a].first.compare(obj[b].first); return c < 0 || (c == 0 && a < b); });
    }
    bool is_null() const { return kind == Null; }
    int64_t as_int(int64_t dflt = 0) const { return kind == Int ? i : kind == Double ? (int64_t)d : dflt; }
    const std::string& as_str() const { static const std::string empty; return kind == String ? s : empty; }
    bool as_bool(bool dflt = false) const { return kind == Bool ? b : dflt; }
};

struct Parser {
    const char* p; const char* end; std::string err; int depth = 0;
    Parser(const char* b, size_t n) : p(b), end(b + n) {}
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool parse_string(std::string& out) {
        if (p >= end || *p != '"') return fail("expected a string");
        ++p; out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break; case '/': out += '/'; break; case '\\': out += '\\'; break; case '"': out += '"'; break;
                    case 'u': {
                        if (end - p < 5) return fail("bad \\u escape");
                        unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16); p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: return fail("bad escape");
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool parse(Value& v) {
        struct Depth { int& d; explicit Depth(int& x) : d(x) { ++d; } ~Depth() { --d; } } guard(depth);
        if (depth > 256) return fail("nested deeper than 256 levels");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            v.kind = Value::Object; ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                v.obj.emplace_back(std::move(key), Value());
                if (!parse(v.obj.back().second)) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; v.index_members(); return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.kind = Value::Array; ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!parse(v.arr.back())) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { v.kind = Value::String; return parse_string(v.s); }
        if (end - p >= 4 && std::strncmp(p, "true", 4) == 0) { v.kind = Value::Bool; v.b = true; p += 4; return true; }
        if (end - p >= 5 && std::strncmp(p, "false", 5) == 0) { v.kind = Value::Bool; v.b = false; p += 5; return true; }
        if (end - p >= 4 && std::strncmp(p, "null", 4) == 0) { v.kind = Value::Null; p += 4; return true; }
        const char* q = p; bool is_double = false;
        if (q < end && (*q == '-' || *q == '+')) ++q;
        while (q < end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '-' || *q == '+')) { if (*q == '.' || *q == 'e' || *q == 'E') is_double = true; ++q; }
        if (q == p) return fail("unexpected character");
        const std::string num(p, q);
        if (is_double) { v.kind = Value::Double; v.d = std::strtod(num.c_str(), nullptr); }
        else { v.kind = Value::Int; v.i = std::strtoll(num.c_str(), nullptr, 10); }
        p = q;
        return true;
    }
};

// serde_json::to_writer_pretty layout: two-space indent, "key": value, empty containers as [] / {}
inline void write_string(std::string& out, const std::string& s) {
    out += '"';
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break; case '\\': out += "\\\\"; break; case '\n': out += "\\n"; break; case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break; case '\b': out += "\\b"; break; case '\f': out += "\\f"; break;
            default:
                if (c < 0x20) { char buf[8]; std::snprintf(buf, sizeof buf, "\\u%04x", c); out += buf; } else out += (char)c;
        }
    }
    out += '"';
}
// f64 as serde_json writes it (ryu's shortest round-trip digits in ryu's layout: 12.34, 0.001234, 12340000000.0, 1.234e33, 1e-7;
// non-finite values become null)
inline void write_double(std::string& out, double d) {
    if (!std::isfinite(d)) { out += "null"; return; }
    if (d == 0.0) { out += std::signbit(d) ? "-0.0" : "0.0"; return; }
    char buf[48];
    auto res = std::to_chars(buf, buf + sizeof buf, d, std::chars_format::scientific);      // shortest digits: d[.ddd]e[+-]xx
    std::string t(buf, res.ptr), digits; bool neg = false; size_t k = 0;
    if (t[0] == '-') { neg = true; k = 1; }
    for (; k < t.size() && t[k] != 'e'; ++k) if (t[k] != '.') digits += t[k];
    const int exp10 = std::atoi(t.c_str() + k + 1);
    const int length = (int)digits.size(), kexp = exp10 - (length - 1), kk = length + kexp;   // value = digits * 10^kexp; 10^(kk-1) <= v < 10^kk
    if (neg) out += '-';
    if (kexp >= 0 && kk <= 16) { out += digits; out.append((size_t)kexp, '0'); out += ".0"; }
    else if (kk > 0 && kk <= 16) { out.append(digits, 0, (size_t)kk); out += '.'; out.append(digits, (size_t)kk, std::string::npos); }
    else if (kk > -5 && kk <= 0) { out += "0."; out.append((size_t)-kk, '0'); out += digits; }
    else {
        out += digits[0];
        if (length > 1) { out += '.'; out.append(digits, 1, std::string::npos); }
        out += 'e'; out += std::to_string(kk - 1);
    }
}
inline void write_pretty(std::string& out, const Value& v, int depth = 0) {
    auto indent = [&](int d) { out.append((size_t)d * 2, ' '); };
    switch (v.kind) {
        case Value::Null: out += "null"; break;
        case Value::Bool: out += v.b ? "true" : "false"; break;
        case Value::Int: out += std::to_string(v.i); break;
        case Value::Double: write_double(out, v.d); break;
        case Value::String: write_string(out, v.s); break;
        case Value::Array:
            if (v.arr.empty()) { out += "[]"; break; }
            out += "[\n";
            for (size_t k = 0; k < v.arr.size(); ++k) { indent(depth + 1); write_pretty(out, v.arr[k], depth + 1); out += k + 1 < v.arr.size() ? ",\n" : "\n"; }
            indent(depth); out += ']';
            break;
        case Value::Object:
            if (v.obj.empty()) { out += "{}"; break; }
            out += "{\n";
            for (size_t k = 0; k < v.obj.size(); ++k) { indent(depth + 1); write_string(out, v.obj[k].first); out += ": "; write_pretty(out, v.obj[k].second, depth + 1); out += k + 1 < v.obj.size() ? ",\n" : "\n"; }
            indent(depth); out += '}';
            break;
    }
}
inline Value str(const std::string& s) { Value v; v.kind = Value::String; v.s = s; return v; }
inline Value num(int64_t i) { Value v; v.kind = Value::Int; v.i = i; return v; }
inline Value real(double d) { Value v; v.kind = Value::Double; v.d = d; return v; }
inline Value boolean(bool b) { Value v; v.kind = Value::Bool; v.b = b; return v; }
inline Value object() { Value v; v.kind = Value::Object; return v; }
inline Value array() { Value v; v.kind = Value::Array; return v; }

} // namespace spj
