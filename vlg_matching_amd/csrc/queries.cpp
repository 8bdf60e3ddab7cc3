// Query batches: the two dialects of the reference's pattern syntax parsed on the host, the batch uploaded to HBM.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include "common.hpp"
#include "search_types.hpp"

using namespace vlg;

// =============================================================================================
// Query batches
// =============================================================================================

namespace {

// std::stoull on [s,e): optional blanks, optional sign, digits; trailing characters ignored
bool parse_u64(const char* s, const char* e, uint64_t& out)
{
    while (s < e && (*s == ' ' || (*s >= 9 && *s <= 13))) ++s;
    bool neg = false;
    if (s < e && (*s == '+' || *s == '-')) { neg = *s == '-'; ++s; }
    if (s >= e || *s < '0' || *s > '9') return false;
    uint64_t v = 0;
    while (s < e && *s >= '0' && *s <= '9') {
        uint64_t d = (uint64_t)(*s - '0');
        if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10) return false;
        v = v * 10 + d;
        ++s;
    }
    out = neg ? (uint64_t)(0 - v) : v;
    return true;
}

struct Parsed {
    std::vector<std::pair<uint64_t, uint64_t>> sub;   // (offset, length) into the query text
    std::vector<uint64_t> lo, hi;                     // per sub-pattern (entry 0 unused)
    uint64_t end_len = 0;
};

// gapped_pattern_query (include/sdsl/vlg_index.hpp:54-105) / gapped_pattern (benchmark utils.hpp:25-70)
vlg_status parse_one(const char* re, uint64_t len, int dialect, Parsed& out, std::string& why)
{
    std::vector<uint64_t> raw_lo(1, 0), raw_hi(1, 0);
    uint64_t start = 0;
    for (;;) {
        uint64_t gp = std::string::npos;
        for (uint64_t i = start; i + 1 < len; ++i) if (re[i] == '.' && re[i + 1] == '{') { gp = i; break; }
        if (gp == std::string::npos) break;
        if (out.sub.size() + 1 >= VLG_MAX_SUBPATTERNS) { why = "too many sub-patterns"; return VLG_E_INVALID; }
        uint64_t ge = std::string::npos, comma = std::string::npos;
        for (uint64_t i = gp; i < len; ++i) if (re[i] == '}') { ge = i; break; }
        if (ge == std::string::npos) { why = "invalid gap description"; return VLG_E_PARSE; }
        for (uint64_t i = gp; i <= ge; ++i) if (re[i] == ',') { comma = i; break; }
        uint64_t a = 0, b = 0;
        if (comma == std::string::npos || !parse_u64(re + gp + 2, re + comma, a) || !parse_u64(re + comma + 1, re + ge, b)) {
            why = "invalid gap description";
            return VLG_E_PARSE;
        }
        if (a > b) { why = "invalid gap description: min-gap > max-gap"; return VLG_E_PARSE; }           // vlg_index.hpp:92-94
        // the reference adds |s| modulo 2^64 (vlg_index.hpp:95); bounds that large are rejected instead of wrapped
        if (b >= (1ull << 62)) { why = "gap bound too large (>= 2^62)"; return VLG_E_INVALID; }
        out.sub.emplace_back(start, gp - start);
        raw_lo.push_back(a);
        raw_hi.push_back(b);
        if (dialect == VLG_DIALECT_LIBRARY) {
            if (ge + 1 == len || re[ge + 1] != '?') {                                                   // vlg_index.hpp:97-99
                why = "invalid gap description: expected '?' (lazy semantics)";
                return VLG_E_PARSE;
            }
            start = ge + 2;
        } else {
            start = ge + 1;
        }
    }
    out.sub.emplace_back(start, len - start);
    for (auto& s : out.sub) if (s.second == 0) { why = "empty sub-pattern"; return VLG_E_INVALID; }
    size_t k = out.sub.size();
    out.lo.assign(k, 0);
    out.hi.assign(k, 0);
    if (dialect == VLG_DIALECT_LIBRARY) {
        for (size_t i = 1; i < k; ++i) {                                                                // vlg_index.hpp:95
            out.lo[i] = raw_lo[i] + out.sub[i - 1].second;
            out.hi[i] = raw_hi[i] + out.sub[i - 1].second;
        }
        out.end_len = out.sub[k - 1].second;                                                            // vlg_index.hpp:262,306
    } else {
        for (size_t i = 1; i < k; ++i) {                                                                // index_sasearch.hpp:68-69
            out.lo[i] = raw_lo[1] + out.sub[0].second;
            out.hi[i] = raw_hi[1] + out.sub[0].second;
        }
        out.end_len = out.sub[0].second;                                                                // index_sasearch.hpp:113
    }
    return VLG_OK;
}

vlg_status upload_queries(vlg_queries* q)
{
    q->kmax = 0; q->kmin = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < q->nq; ++i) {
        uint32_t k = (uint32_t)(q->qsub[i + 1] - q->qsub[i]);
        q->kmax = std::max<uint32_t>(q->kmax, k);
        if (k) q->kmin = std::min<uint32_t>(q->kmin, k);
    }
    if (q->kmin == 0xFFFFFFFFu) q->kmin = 0;
    VLG_HIP_TRY(hipMalloc((void**)&q->d_blob, q->blob.size() + 16));
    VLG_HIP_TRY(hipMalloc((void**)&q->d_suboff, (q->nsub + 1) * 8));
    if (!q->blob.empty()) VLG_HIP_TRY(hipMemcpy(q->d_blob, q->blob.data(), q->blob.size(), hipMemcpyHostToDevice));
    VLG_HIP_TRY(hipMemcpy(q->d_suboff, q->suboff.data(), (q->nsub + 1) * 8, hipMemcpyHostToDevice));
    VLG_HIP_TRY(hipMalloc((void**)&q->d_qsub, (q->nq + 1) * 8));
    VLG_HIP_TRY(hipMemcpy(q->d_qsub, q->qsub.data(), (q->nq + 1) * 8, hipMemcpyHostToDevice));
    return VLG_OK;
}

}  // namespace

extern "C" vlg_status vlg_parse_query(const char* re, uint64_t len, int dialect, vlg_parsed_query* out)
{
    if (!out || (len && !re)) return fail(VLG_E_INVALID, "null argument");
    if (dialect != VLG_DIALECT_LIBRARY && dialect != VLG_DIALECT_BENCHMARK) return fail(VLG_E_INVALID, "unknown dialect");
    memset(out, 0, sizeof *out);
    Parsed p;
    std::string why;
    vlg_status st = parse_one(re, len, dialect, p, why);
    if (st) return fail(st, why);
    out->k = (uint32_t)p.sub.size();
    for (uint32_t i = 0; i < out->k; ++i) {
        out->sub_off[i] = p.sub[i].first; out->sub_len[i] = p.sub[i].second;
        out->lo[i] = p.lo[i]; out->hi[i] = p.hi[i];
    }
    out->end_len = p.end_len;
    return VLG_OK;
}

extern "C" vlg_status vlg_queries_parse(const char* h_text, const uint64_t* h_off, uint64_t n_queries, int dialect, int* h_status,
                                        vlg_queries** out)
{
    if (!out || (n_queries && (!h_text || !h_off))) return fail(VLG_E_INVALID, "null argument");
    if (dialect != VLG_DIALECT_LIBRARY && dialect != VLG_DIALECT_BENCHMARK) return fail(VLG_E_INVALID, "unknown dialect");
    *out = nullptr;
    vlg_queries* q = new vlg_queries();
    q->nq = n_queries;
    q->qsub.assign(1, 0);
    q->suboff.assign(1, 0);
    // The batch is parsed by slices on host threads (a query is a small state machine over its own characters; 10^5 of them take
    // 25 ms on one core) and the slices' pieces are stitched together in batch order.
    struct Slice {
        std::vector<uint8_t> blob;
        std::vector<uint64_t> sublen, lo, hi, end_len;
        std::vector<uint32_t> k;
        vlg_status first_err = VLG_OK;
        std::string first_why;
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t n_slices = n_queries >= 8192 ? std::min<uint64_t>(std::min<unsigned>(hw, 16u), n_queries / 2048) : 1;
    std::vector<Slice> slices(std::max<uint64_t>(n_slices, 1));
    // (no exception may leave a worker thread -- std::terminate -- or cross the C boundary: a slice that runs out of memory records
    // VLG_E_OOM like any other error of its queries)
    auto work_body = [&](uint64_t si) {
        Slice& sl = slices[si];
        const uint64_t b = n_queries * si / slices.size(), e = n_queries * (si + 1) / slices.size();
        sl.k.reserve(e - b); sl.end_len.reserve(e - b);
        for (uint64_t i = b; i < e; ++i) {
            Parsed p;
            std::string why;
            const char* re = h_text + h_off[i];
            vlg_status st = parse_one(re, h_off[i + 1] - h_off[i], dialect, p, why);
            if (h_status) h_status[i] = st;
            if (st) {
                if (!sl.first_err) { sl.first_err = st; sl.first_why = "query " + std::to_string(i) + ": " + why; }
            } else {
                for (size_t s = 0; s < p.sub.size(); ++s) {
                    sl.blob.insert(sl.blob.end(), re + p.sub[s].first, re + p.sub[s].first + p.sub[s].second);
                    sl.sublen.push_back(p.sub[s].second);
                    sl.lo.push_back(p.lo[s]);
                    sl.hi.push_back(p.hi[s]);
                }
            }
            sl.k.push_back(st ? 0u : (uint32_t)p.sub.size());      // a failed query keeps zero sub-patterns
            sl.end_len.push_back(st ? 0 : p.end_len);
        }
    };
    bool oom = false;
    auto work = [&](uint64_t si) {
        try { work_body(si); }
        catch (...) { slices[si].first_err = VLG_E_OOM; slices[si].first_why = "out of host memory while parsing the batch"; }
    };
    if (slices.size() == 1) work(0);
    else {
        std::vector<std::thread> th;
        uint64_t started = 0;
        try {
            th.reserve(slices.size());
            for (; started < slices.size(); ++started) th.emplace_back(work, started);
        } catch (...) {}                                           // no more threads to be had: the rest of the slices on this one
        for (auto& t : th) t.join();
        for (uint64_t si = started; si < slices.size(); ++si) work(si);
    }
    for (const Slice& sl : slices) oom |= sl.first_err == VLG_E_OOM;
    if (oom) { delete q; return fail(VLG_E_OOM, "out of host memory while parsing the batch"); }
    try {
    vlg_status first_err = VLG_OK;
    std::string first_why;
    uint64_t tot_blob = 0, tot_sub = 0;
    for (const Slice& sl : slices) { tot_blob += sl.blob.size(); tot_sub += sl.sublen.size(); }
    q->blob.reserve(tot_blob + 16); q->suboff.reserve(tot_sub + 1); q->lo.reserve(tot_sub); q->hi.reserve(tot_sub);
    q->qsub.reserve(n_queries + 1); q->end_len.reserve(n_queries);
    for (const Slice& sl : slices) {
        if (sl.first_err && !first_err) { first_err = sl.first_err; first_why = sl.first_why; }
        q->blob.insert(q->blob.end(), sl.blob.begin(), sl.blob.end());
        for (uint64_t len : sl.sublen) q->suboff.push_back(q->suboff.back() + len);
        q->lo.insert(q->lo.end(), sl.lo.begin(), sl.lo.end());
        q->hi.insert(q->hi.end(), sl.hi.begin(), sl.hi.end());
        for (uint32_t k : sl.k) q->qsub.push_back(q->qsub.back() + k);
        q->end_len.insert(q->end_len.end(), sl.end_len.begin(), sl.end_len.end());
    }
    q->nsub = q->suboff.size() - 1;
    if (first_err && !h_status) { delete q; return fail(first_err, first_why); }
    } catch (const std::bad_alloc&) { delete q; return fail(VLG_E_OOM, "out of host memory while assembling the batch"); }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete q; return fail(VLG_E_NO_DEVICE, "no HIP device available"); }
    if (vlg_status st = upload_queries(q)) { vlg_queries_destroy(q); return st; }
    *out = q;
    return VLG_OK;
}

// Integer alphabet (gapped_pattern_query<int_alphabet_tag>, include/sdsl/vlg_index.hpp:57-69): a sub-pattern is read with
// `istringstream >> uint64_t` until the first token that is not a number; gaps count symbols (:95).
// The device indexes hold uint32_t symbols.  Tokens are read as the reference reads them -- 64 bits -- and then either must fit 32
// bits (no map) or go through a vlg_symbol_map: the sorted distinct 64-bit symbols of the text, symbol -> rank + 1.
struct vlg_symbol_map {
    std::vector<uint64_t> symbols;                             // ascending, distinct
    // rank + 1 of a symbol of the text; a symbol that does not occur maps to sigma + 1, which occurs nowhere in a mapped text
    uint32_t map(uint64_t x) const
    {
        const auto it = std::lower_bound(symbols.begin(), symbols.end(), x);
        if (it == symbols.end() || *it != x) return (uint32_t)symbols.size() + 1u;
        return (uint32_t)(it - symbols.begin()) + 1u;
    }
};

extern "C" vlg_status vlg_symbol_map_create(const uint64_t* h_text, uint64_t n_symbols, vlg_symbol_map** out)
{
    if (!out || (n_symbols && !h_text)) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    try {
        std::vector<uint64_t> v(h_text, h_text + n_symbols);
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        if (v.size() >= 0xFFFFFFF0ull) return fail(VLG_E_UNSUPPORTED, "more than 2^32 - 16 distinct symbols");
        vlg_symbol_map* m = new vlg_symbol_map();
        m->symbols.swap(v);
        *out = m;
    } catch (const std::bad_alloc&) { return fail(VLG_E_OOM, "out of host memory while collecting the alphabet"); }
    return VLG_OK;
}

extern "C" uint64_t vlg_symbol_map_sigma(const vlg_symbol_map* m) { return m ? m->symbols.size() : 0; }

extern "C" vlg_status vlg_symbol_map_symbols(const vlg_symbol_map* m, uint64_t* h_out)
{
    if (!m || (!h_out && !m->symbols.empty())) return fail(VLG_E_INVALID, "null argument");
    if (!m->symbols.empty()) memcpy(h_out, m->symbols.data(), m->symbols.size() * 8);
    return VLG_OK;
}

extern "C" vlg_status vlg_symbol_map_apply(const vlg_symbol_map* m, const uint64_t* h_in, uint64_t n, uint32_t* h_out)
{
    if (!m || (n && (!h_in || !h_out))) return fail(VLG_E_INVALID, "null argument");
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const uint64_t nt = n >= (1u << 20) ? hw : 1;
    auto work = [&](uint64_t a, uint64_t b) { for (uint64_t i = a; i < b; ++i) h_out[i] = m->map(h_in[i]); };
    if (nt == 1) { work(0, n); return VLG_OK; }
    std::vector<std::thread> th;
    uint64_t started = 0;
    try { for (; started < nt; ++started) th.emplace_back(work, n * started / nt, n * (started + 1) / nt); } catch (...) {}
    for (auto& t : th) t.join();
    if (started < nt) work(n * started / nt, n);               // no more threads to be had: the rest on this one
    return VLG_OK;
}

extern "C" void vlg_symbol_map_destroy(vlg_symbol_map* m) { delete m; }

namespace {
vlg_status parse_int_batch(const vlg_symbol_map* map, const char* h_text, const uint64_t* h_off, uint64_t n_queries, int* h_status, vlg_queries** out)
{
    if (!out || (n_queries && (!h_text || !h_off))) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    vlg_queries* q = new vlg_queries();
    q->nq = n_queries;
    q->sym_bytes = 4;
    q->qsub.assign(1, 0);
    q->suboff.assign(1, 0);
    vlg_status first_err = VLG_OK;
    std::string first_why;
    for (uint64_t i = 0; i < n_queries; ++i) {
        Parsed p;
        std::string why;
        const char* re = h_text + h_off[i];
        vlg_status st = parse_one(re, h_off[i + 1] - h_off[i], VLG_DIALECT_LIBRARY, p, why);
        std::vector<std::vector<uint32_t>> syms;
        if (!st) {
            for (auto& sp : p.sub) {
                std::vector<uint32_t> v;
                const char* c = re + sp.first;
                const char* e = c + sp.second;
                for (;;) {
                    while (c < e && (*c == ' ' || (*c >= 9 && *c <= 13))) ++c;
                    if (c < e && *c == '+') ++c;
                    if (c >= e || *c < '0' || *c > '9') break;
                    uint64_t x = 0;
                    bool over = false;
                    while (c < e && *c >= '0' && *c <= '9') {
                        const uint64_t dgt = (uint64_t)(*c - '0');
                        if (x > 0xFFFFFFFFFFFFFFFFull / 10 || (x == 0xFFFFFFFFFFFFFFFFull / 10 && dgt > 0xFFFFFFFFFFFFFFFFull % 10)) over = true;      // beyond 2^64 - 1
                        x = x * 10 + dgt;
                        ++c;
                    }
                    if (over) break;                                                    // the stream extraction fails: the rest is ignored
                    if (map) v.push_back(map->map(x));
                    else {
                        if (x > 0xFFFFFFFFull) { st = VLG_E_INVALID; why = "symbol does not fit 32 bits (larger symbols: vlg_symbol_map_create + vlg_queries_parse_int_mapped)"; break; }
                        v.push_back((uint32_t)x);
                    }
                }
                if (!st && v.empty()) { st = VLG_E_INVALID; why = "empty sub-pattern"; }
                if (st) break;
                syms.push_back(v);
            }
        }
        if (h_status) h_status[i] = st;
        if (st) {
            if (!first_err) { first_err = st; first_why = "query " + std::to_string(i) + ": " + why; }
        } else {
            for (size_t s = 0; s < syms.size(); ++s) {
                const uint8_t* b = reinterpret_cast<const uint8_t*>(syms[s].data());
                q->blob.insert(q->blob.end(), b, b + syms[s].size() * 4);
                q->suboff.push_back(q->blob.size());
                // the parser added the sub-pattern's length in characters; gaps count symbols
                const uint64_t raw_lo = s ? p.lo[s] - p.sub[s - 1].second : 0, raw_hi = s ? p.hi[s] - p.sub[s - 1].second : 0;
                q->lo.push_back(s ? raw_lo + syms[s - 1].size() : 0);
                q->hi.push_back(s ? raw_hi + syms[s - 1].size() : 0);
            }
        }
        q->qsub.push_back(q->suboff.size() - 1);
        q->end_len.push_back(st ? 0 : syms.back().size());
    }
    q->nsub = q->suboff.size() - 1;
    if (first_err && !h_status) { delete q; return fail(first_err, first_why); }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete q; return fail(VLG_E_NO_DEVICE, "no HIP device available"); }
    if (vlg_status st = upload_queries(q)) { vlg_queries_destroy(q); return st; }
    *out = q;
    return VLG_OK;
}
}  // namespace

extern "C" vlg_status vlg_queries_parse_int(const char* h_text, const uint64_t* h_off, uint64_t n_queries, int* h_status, vlg_queries** out)
{
    return parse_int_batch(nullptr, h_text, h_off, n_queries, h_status, out);
}

extern "C" vlg_status vlg_queries_parse_int_mapped(const vlg_symbol_map* map, const char* h_text, const uint64_t* h_off, uint64_t n_queries,
                                                   int* h_status, vlg_queries** out)
{
    if (!map) return fail(VLG_E_INVALID, "null argument");
    return parse_int_batch(map, h_text, h_off, n_queries, h_status, out);
}

extern "C" vlg_status vlg_queries_create(const uint8_t* h_blob, const uint64_t* h_suboff, const uint64_t* h_qsub, const uint64_t* h_lo,
                                         const uint64_t* h_hi, const uint64_t* h_end_len, uint64_t n_queries, vlg_queries** out)
{
    if (!out || (n_queries && (!h_suboff || !h_qsub || !h_lo || !h_hi || !h_end_len))) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    vlg_queries* q = new vlg_queries();
    q->nq = n_queries;
    q->nsub = n_queries ? h_qsub[n_queries] : 0;
    q->qsub.assign(h_qsub, h_qsub + n_queries + 1);
    if (!n_queries) q->qsub.assign(1, 0);
    q->suboff.assign(1, 0);
    if (q->nsub) q->suboff.assign(h_suboff, h_suboff + q->nsub + 1);
    for (uint64_t i = 0; i < n_queries; ++i)
        if (q->qsub[i + 1] < q->qsub[i] || q->qsub[i + 1] - q->qsub[i] > VLG_MAX_SUBPATTERNS) { delete q; return fail(VLG_E_INVALID, "bad query offsets"); }
    for (uint64_t s = 0; s < q->nsub; ++s)
        if (q->suboff[s + 1] <= q->suboff[s]) { delete q; return fail(VLG_E_INVALID, "empty sub-pattern"); }
    if (q->nsub && !h_blob) { delete q; return fail(VLG_E_INVALID, "null argument"); }
    if (q->nsub) q->blob.assign(h_blob, h_blob + q->suboff[q->nsub]);
    q->lo.assign(h_lo, h_lo + q->nsub);
    q->hi.assign(h_hi, h_hi + q->nsub);
    for (uint64_t i = 0; i < n_queries; ++i)
        for (uint64_t sidx = q->qsub[i] + 1; sidx < q->qsub[i + 1]; ++sidx)
            if (q->lo[sidx] > q->hi[sidx] || q->hi[sidx] >= (1ull << 63)) { delete q; return fail(VLG_E_INVALID, "bad gap bounds"); }
    q->end_len.assign(h_end_len, h_end_len + n_queries);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete q; return fail(VLG_E_NO_DEVICE, "no HIP device available"); }
    if (vlg_status st = upload_queries(q)) { vlg_queries_destroy(q); return st; }
    *out = q;
    return VLG_OK;
}

extern "C" uint64_t vlg_queries_count(const vlg_queries* q) { return q ? q->nq : 0; }
extern "C" uint64_t vlg_queries_subpatterns(const vlg_queries* q) { return q ? q->nsub : 0; }
extern "C" vlg_status vlg_queries_k(const vlg_queries* q, uint32_t* h_k)
{
    if (!q || (q->nq && !h_k)) return fail(VLG_E_INVALID, "null argument");
    for (uint64_t i = 0; i < q->nq; ++i) h_k[i] = (uint32_t)(q->qsub[i + 1] - q->qsub[i]);
    return VLG_OK;
}
extern "C" void vlg_queries_destroy(vlg_queries* q)
{
    if (!q) return;
    if (q->d_blob) (void)hipFree(q->d_blob);
    if (q->d_suboff) (void)hipFree(q->d_suboff);
    if (q->d_qsub) (void)hipFree(q->d_qsub);
    delete q;
}

