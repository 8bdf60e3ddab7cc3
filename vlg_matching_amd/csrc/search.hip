// Query batches, workspace, the fused search (backward search -> locate -> sort -> join) and results.
//
// Join (K5).  The reference's merge join (benchmark/gapped-matching/include/index_sasearch.hpp:85-116;
// semantics of vlg_iterator, include/sdsl/vlg_index.hpp:227-291) advances k monotone pointers one step
// at a time.  Every step only ever raises one pointer to the least value a gap constraint forces, so a
// match is the component-wise LEAST tuple (p_0 >= a, p_1, ..., p_{k-1}) that satisfies all constraints
// lo_i <= L_i[p_i] - L_{i-1}[p_{i-1}] <= hi_i, and the next search restarts at the first p_0 with
// L_0[p_0] >= L_{k-1}[p_{k-1}] + end_len.  That fixed-point view is data parallel:
//   back to front, every element of list i learns whether a feasible chain to the last list starts at
//   it, which element of list i+1 it links to (the first feasible one inside its window) and where the
//   chain ends ("link pass", one binary search per element);  a reverse min-scan gives "nearest feasible
//   element at or after j";  a per-query wavefront then hops along list 0 (window of 64 jump targets per
//   load) emitting the non-overlapping matches in order;  a gather pass writes the tuples.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string.h>
#include <string>
#include <vector>
#include "common.hpp"
#include "kernels.hpp"
#include <rocprim/rocprim.hpp>

using namespace vlg;

// =============================================================================================
// Query batches
// =============================================================================================
struct vlg_queries {
    uint64_t nq = 0, nsub = 0;
    std::vector<uint64_t> qsub;      // [nq+1]
    std::vector<uint64_t> suboff;    // [nsub+1]
    std::vector<uint8_t> blob;
    std::vector<uint64_t> lo, hi;    // [nsub]
    std::vector<uint64_t> end_len;   // [nq]
    uint32_t kmax = 0, kmin = 0;     // over queries with at least one sub-pattern
    uint8_t* d_blob = nullptr;
    uint64_t* d_suboff = nullptr;
};

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
    vlg_status first_err = VLG_OK;
    std::string first_why;
    for (uint64_t i = 0; i < n_queries; ++i) {
        Parsed p;
        std::string why;
        const char* re = h_text + h_off[i];
        vlg_status st = parse_one(re, h_off[i + 1] - h_off[i], dialect, p, why);
        if (h_status) h_status[i] = st;
        if (st) {
            if (!first_err) { first_err = st; first_why = "query " + std::to_string(i) + ": " + why; }
        } else {
            for (size_t s = 0; s < p.sub.size(); ++s) {
                q->blob.insert(q->blob.end(), re + p.sub[s].first, re + p.sub[s].first + p.sub[s].second);
                q->suboff.push_back(q->blob.size());
                q->lo.push_back(p.lo[s]);
                q->hi.push_back(p.hi[s]);
            }
        }
        q->qsub.push_back(q->suboff.size() - 1);     // a failed query keeps zero sub-patterns
        q->end_len.push_back(st ? 0 : p.end_len);
    }
    q->nsub = q->suboff.size() - 1;
    if (first_err && !h_status) { delete q; return fail(first_err, first_why); }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete q; return fail(VLG_E_NO_DEVICE, "no HIP device available"); }
    if (vlg_status st = upload_queries(q)) { vlg_queries_destroy(q); return st; }
    *out = q;
    return VLG_OK;
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
    delete q;
}

// =============================================================================================
// Workspace
// =============================================================================================
enum { KS_BSEARCH = 0, KS_EXPAND, KS_LOCATE, KS_LOCATE_PART, KS_LOCATE_RESOLVE, KS_SORT, KS_FILTER_PIVOT, KS_FILTER_PASS, KS_FILTER_COMPACT,
       KS_JOIN_INIT, KS_JOIN_LINK, KS_JOIN_SCAN, KS_JOIN_CHAIN, KS_GATHER, KS_COUNT };
static const char* kKernelNames[KS_COUNT] = {"backward_search", "expand", "locate", "locate_partition", "locate_resolve", "sort", "filter_pivot",
                                             "filter_pass", "filter_compact", "join_init", "join_link", "join_scan", "join_chain", "gather"};

struct vlg_workspace {
    hipStream_t stream = nullptr;
    uint64_t cap_bytes = 0;
    uint8_t* arena = nullptr;
    uint64_t arena_bytes = 0;
    bool profile = false;
    bool dedup = true;
    bool sweep = true;          // sorted-sweep locate (n <= 2^32) instead of the random-access persistent kernel
    bool trail = true;          // sorted-sweep locate: elements that step onto an SA index another element has visited share its LF trail
                                // (needs dedup; with dedup off every occurrence walks its own LF steps like the reference)
    bool filter = true;         // window filter: drop the list elements that can be in no match before the join
    uint64_t filter_min = 1ull << 16;   // queries with fewer join slots are joined as they are
    bool filter_pivot = true;   // filter from the shortest list of a query outwards when it is much shorter than the rest
    uint64_t filter_pivot_ratio = 12;   // ... i.e. when all lists together are at least this many times longer (measured on C3: 12)
    uint64_t global_sort_min = 1ull << 20;  // at least this many occurrences: all lists are sorted by one radix sort of (list, position) keys
    uint64_t sweep_min = 1ull << 22;    // below this many occurrences the persistent random-access kernel is used
    uint64_t sweep_tail = 1ull << 20;   // stragglers of a sweep are finished one lane each
    vlg_kernel_stat stats[KS_COUNT];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[KS_COUNT];
    std::vector<hipEvent_t> free_events;
};

namespace {

// VLG_TRACE=1: wall time of the host phases of a batch on stderr (the stream is drained at every mark, so the figures
// include the kernels launched in the phase)
struct PhaseTrace {
    bool on;
    hipStream_t st;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTrace(hipStream_t s) : on(getenv("VLG_TRACE") != nullptr), st(s), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what)
    {
        if (!on) return;
        (void)hipStreamSynchronize(st);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[vlg trace] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

hipEvent_t ws_event(vlg_workspace* ws)
{
    if (!ws->free_events.empty()) { hipEvent_t e = ws->free_events.back(); ws->free_events.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct Timed {      // RAII: HIP events around one launch (or one library call) on the workspace stream
    vlg_workspace* ws; int k; hipEvent_t a = nullptr, b = nullptr;
    Timed(vlg_workspace* w, int kernel, uint64_t alg_bytes) : ws(w), k(kernel)
    {
        ws->stats[k].launches++;
        ws->stats[k].algorithmic_bytes += alg_bytes;
        if (ws->profile) { a = ws_event(ws); b = ws_event(ws); if (a) (void)hipEventRecord(a, ws->stream); }
    }
    ~Timed() { if (a && b) { (void)hipEventRecord(b, ws->stream); ws->pending[k].emplace_back(a, b); } }
};

struct SweepTimer : LaunchTimer {      // one event pair per launch of the sweep, accounted per kernel class
    vlg_workspace* ws; hipEvent_t a = nullptr, b = nullptr;
    explicit SweepTimer(vlg_workspace* w) : ws(w) {}
    void begin(int which) override
    {
        int k = which == 0 ? KS_LOCATE : (which == 1 ? KS_LOCATE_PART : KS_LOCATE_RESOLVE);
        ws->stats[k].launches++;
        a = b = nullptr;
        if (ws->profile) { a = ws_event(ws); b = ws_event(ws); if (a) (void)hipEventRecord(a, ws->stream); }
    }
    void end(int which) override
    {
        int k = which == 0 ? KS_LOCATE : (which == 1 ? KS_LOCATE_PART : KS_LOCATE_RESOLVE);
        if (a && b) { (void)hipEventRecord(b, ws->stream); ws->pending[k].emplace_back(a, b); }
    }
};

void ws_collect(vlg_workspace* ws)
{
    for (int k = 0; k < KS_COUNT; ++k) {
        for (auto& pr : ws->pending[k]) {
            float ms = 0;
            (void)hipEventSynchronize(pr.second);
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) ws->stats[k].total_ms += ms;
            ws->free_events.push_back(pr.first);
            ws->free_events.push_back(pr.second);
        }
        ws->pending[k].clear();
    }
}

void ws_reset_stats(vlg_workspace* ws)
{
    ws_collect(ws);
    for (int k = 0; k < KS_COUNT; ++k) {
        memset(&ws->stats[k], 0, sizeof(vlg_kernel_stat));
        strncpy(ws->stats[k].name, kKernelNames[k], sizeof(ws->stats[k].name) - 1);
    }
}

void drain_result_cache();

vlg_status ws_reserve(vlg_workspace* ws, uint64_t bytes)
{
    if (bytes <= ws->arena_bytes) return VLG_OK;
    if (ws->arena) { (void)hipFree(ws->arena); ws->arena = nullptr; ws->arena_bytes = 0; }
    if (hipMalloc((void**)&ws->arena, bytes) != hipSuccess) {        // parked result buffers may be in the way
        (void)hipGetLastError();
        ws->arena = nullptr;
        drain_result_cache();
        VLG_HIP_TRY(hipMalloc((void**)&ws->arena, bytes));
    }
    ws->arena_bytes = bytes;
    return VLG_OK;
}

}  // namespace

extern "C" vlg_status vlg_workspace_create(uint64_t max_hbm_bytes, void* stream, vlg_workspace** out)
{
    if (!out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available");
    vlg_workspace* ws = new vlg_workspace();
    ws->stream = (hipStream_t)stream;
    ws->cap_bytes = max_hbm_bytes ? max_hbm_bytes : (8ull << 30);
    ws_reset_stats(ws);
    *out = ws;
    return VLG_OK;
}

extern "C" void vlg_workspace_destroy(vlg_workspace* ws)
{
    if (!ws) return;
    ws_collect(ws);
    for (hipEvent_t e : ws->free_events) (void)hipEventDestroy(e);
    if (ws->arena) (void)hipFree(ws->arena);
    delete ws;
}

extern "C" vlg_status vlg_workspace_profile(vlg_workspace* ws, int enable)
{
    if (!ws) return fail(VLG_E_INVALID, "null argument");
    ws_reset_stats(ws);
    ws->profile = enable != 0;
    return VLG_OK;
}

extern "C" vlg_status vlg_workspace_set_option(vlg_workspace* ws, const char* name, int64_t value)
{
    if (!ws || !name) return fail(VLG_E_INVALID, "null argument");
    if (!strcmp(name, "dedup")) { ws->dedup = value != 0; return VLG_OK; }
    if (!strcmp(name, "sweep")) { ws->sweep = value != 0; return VLG_OK; }
    if (!strcmp(name, "sweep_min")) { ws->sweep_min = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "global_sort_min")) { ws->global_sort_min = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "trail")) { ws->trail = value != 0; return VLG_OK; }
    if (!strcmp(name, "filter")) { ws->filter = value != 0; return VLG_OK; }
    if (!strcmp(name, "filter_min")) { ws->filter_min = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "filter_pivot")) { ws->filter_pivot = value != 0; return VLG_OK; }
    if (!strcmp(name, "filter_pivot_ratio")) { ws->filter_pivot_ratio = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "sweep_tail")) { ws->sweep_tail = (uint64_t)value; return VLG_OK; }
    return fail(VLG_E_INVALID, std::string("unknown workspace option ") + name);
}

extern "C" vlg_status vlg_workspace_kernel_stats(vlg_workspace* ws, vlg_kernel_stat* out, uint32_t cap, uint32_t* n)
{
    if (!ws || !n) return fail(VLG_E_INVALID, "null argument");
    ws_collect(ws);
    *n = KS_COUNT;
    for (uint32_t k = 0; k < KS_COUNT && k < cap && out; ++k) out[k] = ws->stats[k];
    return VLG_OK;
}

// =============================================================================================
// Results
// =============================================================================================
struct ResultPiece {
    uint64_t q0 = 0, q1 = 0;       // query range of the chunk
    uint64_t matches = 0, tuple_vals = 0;
    uint64_t* d_first = nullptr;   // [matches]
    uint64_t* d_tuples = nullptr;  // [tuple_vals]
    uint64_t first_bytes = 0, tuple_bytes = 0;   // sizes of the allocations (a parked buffer may be larger than needed)
};

struct vlg_result {
    vlg_result_summary sum;
    std::vector<uint64_t> counts;          // host: per query
    std::vector<uint32_t> k;               // host: sub-patterns per query
    std::vector<ResultPiece> pieces;
};

// Result buffers are large and batches come one after the other: freed buffers are parked (up to kResultCacheBytes) and
// handed to the next result of a similar size instead of going through hipFree / hipMalloc every batch.
namespace {
constexpr uint64_t kResultCacheBytes = 48ull << 30;
struct ResultCache {
    struct Entry { void* p; uint64_t size; int device; };
    std::mutex mu;
    std::vector<Entry> free_list;
    uint64_t bytes = 0;
    static int device() { int d = 0; (void)hipGetDevice(&d); return d; }
    void* take(uint64_t need, uint64_t* size)
    {
        const int dev = device();
        std::lock_guard<std::mutex> g(mu);
        size_t best = free_list.size();
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].device == dev && free_list[i].size >= need && free_list[i].size <= need + need / 4 + (1u << 20) &&
                (best == free_list.size() || free_list[i].size < free_list[best].size)) best = i;
        if (best == free_list.size()) return nullptr;
        void* p = free_list[best].p;
        *size = free_list[best].size;
        bytes -= free_list[best].size;
        free_list.erase(free_list.begin() + best);
        return p;
    }
    void give(void* p, uint64_t size)
    {
        {
            const int dev = device();             // results are created and destroyed with their device current
            std::lock_guard<std::mutex> g(mu);
            if (bytes + size <= kResultCacheBytes && free_list.size() < 64) { free_list.push_back(Entry{p, size, dev}); bytes += size; return; }
        }
        (void)hipFree(p);
    }
    void drain()
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto& e : free_list) (void)hipFree(e.p);
        free_list.clear();
        bytes = 0;
    }
};
ResultCache& result_cache() { static ResultCache* c = new ResultCache(); return *c; }     // never destroyed: no HIP calls at exit
void drain_result_cache() { result_cache().drain(); }

hipError_t result_alloc(uint64_t** out, uint64_t bytes, uint64_t* got)
{
    if (void* p = result_cache().take(bytes, got)) { *out = (uint64_t*)p; return hipSuccess; }
    hipError_t e = hipMalloc((void**)out, bytes);
    if (e != hipSuccess) {                                   // memory may be parked in the cache: release it and retry once
        (void)hipGetLastError();
        result_cache().drain();
        e = hipMalloc((void**)out, bytes);
    }
    *got = bytes;
    return e;
}
}  // namespace

namespace vlg {
void release_cached_device_memory() { result_cache().drain(); }
}

extern "C" void vlg_result_destroy(vlg_result* r)
{
    if (!r) return;
    for (auto& p : r->pieces) {
        if (p.d_first) result_cache().give(p.d_first, p.first_bytes);
        if (p.d_tuples) result_cache().give(p.d_tuples, p.tuple_bytes);
    }
    delete r;
}

extern "C" vlg_status vlg_result_summary_get(const vlg_result* r, vlg_result_summary* s)
{
    if (!r || !s) return fail(VLG_E_INVALID, "null argument");
    *s = r->sum;
    return VLG_OK;
}

extern "C" vlg_status vlg_result_fetch(const vlg_result* r, uint64_t* h_counts, uint64_t* h_offsets, uint64_t* h_first, uint64_t* h_tuples)
{
    if (!r) return fail(VLG_E_INVALID, "null argument");
    uint64_t nq = r->counts.size();
    if (h_counts && nq) memcpy(h_counts, r->counts.data(), nq * 8);
    if (h_offsets) {
        uint64_t acc = 0;
        for (uint64_t q = 0; q < nq; ++q) { h_offsets[q] = acc; acc += r->counts[q]; }
        h_offsets[nq] = acc;
    }
    uint64_t fo = 0, to = 0;
    for (const auto& p : r->pieces) {
        if (h_first && p.matches) VLG_HIP_TRY(hipMemcpy(h_first + fo, p.d_first, p.matches * 8, hipMemcpyDeviceToHost));
        if (h_tuples && p.tuple_vals) VLG_HIP_TRY(hipMemcpy(h_tuples + to, p.d_tuples, p.tuple_vals * 8, hipMemcpyDeviceToHost));
        fo += p.matches;
        to += p.tuple_vals;
    }
    return VLG_OK;
}

// =============================================================================================
// Join kernels
// =============================================================================================
namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

// Occurrence lists are PHYSICAL: one sorted list per distinct SA interval of the batch, shared by every
// query that uses the sub-pattern.  Join state (link / end / feasibility ...) is LOGICAL: one slot per
// element of every (query, level) whose level is not the query's last one (the last list is only searched).
// Logical slots are laid out class-major: all segments with the same `dist` (sub-patterns after them in their
// query) are contiguous, so every pass of the join streams exactly the slots it works on.
struct SegMeta {            // one per sub-pattern of the chunk (device array, class-major order)
    uint32_t begin, end;    // logical slots (begin == end for the last level of a k>=2 query)
    uint32_t pbegin, pend;  // physical list inside P
    uint32_t dist;          // sub-patterns after it in its query (0 = last)
    uint32_t level;         // index inside the query (0 = first)
    uint32_t next;          // segment of the query's next sub-pattern (valid when dist > 0)
    uint32_t query;         // query of the chunk
    uint64_t lo, hi;        // gap bounds between the previous sub-pattern and this one
};

struct QueryMeta {          // one per query of the chunk
    uint32_t seg0;          // first segment (level 0); kNone if the query is dead
    uint32_t k;
    uint64_t end_len;
    uint64_t out_first;     // offsets into the chunk's result arrays (filled before gather)
    uint64_t out_tuple;
};

__device__ __forceinline__ uint64_t sat_add(uint64_t a, uint64_t b) { uint64_t c = a + b; return c < a ? ~0ull : c; }
__device__ __forceinline__ uint32_t phys_of(const SegMeta& m, uint32_t e) { return m.pbegin + (e - m.begin); }

template <typename pos_t>
__device__ __forceinline__ uint32_t lower_bound_dev(const pos_t* __restrict__ P, uint32_t a, uint32_t b, uint64_t key)
{
    while (a < b) {
        uint32_t mid = a + ((b - a) >> 1);
        if ((uint64_t)P[mid] < key) a = mid + 1; else b = mid;
    }
    return a;
}

// Lower bound by galloping from a known lower fence: all indices below `lo` hold values < key.
// Consecutive slots of a list have ascending keys, so the previous answer is a tight fence and the search costs
// O(log distance) probes into lines the neighbouring lanes touch too, instead of log2 |list| cold probes.
template <typename pos_t>
__device__ __forceinline__ uint32_t gallop_lower_bound(const pos_t* __restrict__ P, uint32_t lo, uint32_t b, uint64_t key)
{
    uint32_t step = 1, hi = b;
    bool found = false;
    while (lo < b) {
        uint32_t p = lo + step - 1;
        if (p >= b) p = b - 1;
        if ((uint64_t)P[p] < key) { lo = p + 1; step <<= 1; }
        else { hi = p; found = true; break; }
    }
    if (!found) return b;
    return lower_bound_dev(P, lo, hi, key);
}

// Lower bounds of 64 ascending keys in one sorted list, as a wave: the answers of a step lie just behind the last
// answer of the previous step, so the wave loads consecutive 64-element windows of the list with ONE coalesced load
// each and every lane ranks its key inside the window through cross-lane reads (6 steps) -- a merge of two sorted
// runs, without the ~15 scattered probes per lane of an independent search.  `wb` (wave-uniform) must be a fence:
// every element before it is smaller than every key.  Lanes still unresolved after kCoopWindows windows fall back
// to galloping from the last window's end.
constexpr uint32_t kCoopWindows = 4;
template <typename pos_t>
__device__ __forceinline__ uint32_t wave_lower_bound(const pos_t* __restrict__ P, uint32_t wb, uint32_t b, uint64_t key, bool need)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t res = b;
    for (uint32_t it = 0; it < kCoopWindows; ++it) {
        if (!__any(need)) break;
        const uint32_t idx = wb + lane;
        const uint64_t w = idx < b ? (uint64_t)P[idx] : ~0ull;       // +inf behind the list
        const uint64_t wlast = __shfl(w, 63);
        const bool can = need && key <= wlast;
        uint32_t lo = 0, hi = 63;                                    // for `can` lanes w[63] >= key, so the answer is in [0,63]
#pragma unroll
        for (uint32_t st = 0; st < 6; ++st) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint64_t v = __shfl(w, (int)mid);
            if (v < key) lo = mid + 1; else hi = mid;
        }
        if (can) { res = wb + lo; need = false; }
        wb += 64;
    }
    if (need) res = gallop_lower_bound(P, wb < b ? wb : b, b, key);
    return res < b ? res : b;
}

// ---- wave-private list tiles -----------------------------------------------------------------------------------
// A wave that needs the lower bounds of many keys in one sorted list stages the list in LDS, kTB elements at a time
// (coalesced loads that do not depend on any answer), and every lane searches its kKeys keys there: log2(kTB) LDS probes
// per key, several independent searches per lane in flight, no dependent global round trip per key.
constexpr uint32_t kKeys = 8;                 // keys per lane and block
constexpr uint32_t kBlk = 64 * kKeys;         // slots per block
constexpr uint32_t kTB = 1024;                // list elements per tile
constexpr uint32_t kKeyGroup = 4;             // searches interleaved per lane
constexpr uint32_t kMinPiece = 128;           // shorter pieces of a segment take the per-lane path

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T>
__device__ __forceinline__ T wave_min(T v)
{
    for (int o = 32; o > 0; o >>= 1) { const T u = __shfl_xor(v, o); v = u < v ? u : v; }
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(v, o); v = u > v ? u : v; }
    return v;
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform(uint64_t v)
{
    return (uint64_t)uniform((uint32_t)v) | ((uint64_t)uniform((uint32_t)(v >> 32)) << 32);
}

// Lower bound of one wave-uniform key in P[a,b): 64 probes per round narrow the range 64-fold.
template <typename pos_t>
__device__ __forceinline__ uint32_t wave_kary_lower_bound(const pos_t* __restrict__ P, uint32_t a, uint32_t b, uint64_t key)
{
    const uint32_t lane = threadIdx.x & 63;
    while (b - a > 64) {
        const uint32_t step = (b - a + 63) / 64;
        const uint64_t idx = (uint64_t)a + (uint64_t)(lane + 1) * step - 1;
        const bool in = idx < b;
        const uint64_t v = in ? (uint64_t)P[idx] : 0;
        const uint32_t c = (uint32_t)__popcll(__ballot(in && v < key));          // probes 0..c-1 are smaller than the key
        const uint64_t na = (uint64_t)a + (uint64_t)c * step;                    // <= b
        const uint64_t nb = (uint64_t)a + (uint64_t)(c + 1) * step - 1;          // probe c (if it exists) is not smaller
        a = (uint32_t)na;
        b = nb < b ? (uint32_t)nb : b;
    }
    const uint32_t idx = a + lane;
    const bool in = idx < b;
    const uint64_t v = in ? (uint64_t)P[idx] : 0;
    return a + (uint32_t)__popcll(__ballot(in && v < key));
}

// Lower bounds in P[.,pend) of the keys flagged in `need` (bit i = key[i]).  `wb` is a wave-uniform fence: every element
// before it is smaller than every flagged key of the wave.  j[i] = the lower bound (pend if there is none), v[i] = P[j[i]].
// The keys of the wave need not be ordered; tiles that cannot hold an answer are skipped with one probe.
template <typename pos_t>
__device__ __forceinline__ void tile_lower_bounds(const pos_t* __restrict__ P, uint32_t wb, const uint32_t pend, pos_t* __restrict__ tile,
                                                  const pos_t (&key)[kKeys], uint32_t need, uint32_t (&j)[kKeys], pos_t (&v)[kKeys])
{
    const uint32_t lane = threadIdx.x & 63;
    constexpr pos_t kInf = (pos_t)~(pos_t)0;
#pragma unroll
    for (uint32_t i = 0; i < kKeys; ++i) { j[i] = pend; v[i] = 0; }
    while (__any(need != 0) && wb < pend) {
#pragma unroll
        for (uint32_t r = 0; r < kTB / 64; ++r) {
            const uint64_t idx = (uint64_t)wb + lane + 64 * r;
            tile[lane + 64 * r] = idx < pend ? P[idx] : kInf;                    // +inf behind the list
        }
        wave_sync();
        const pos_t tile_last = tile[kTB - 1];
#pragma unroll
        for (uint32_t g = 0; g < kKeys; g += kKeyGroup) {
            bool take[kKeyGroup];
            bool any = false;
#pragma unroll
            for (uint32_t i = 0; i < kKeyGroup; ++i) { take[i] = ((need >> (g + i)) & 1) && key[g + i] <= tile_last; any |= take[i]; }
            if (!__any(any)) continue;
            uint32_t pos[kKeyGroup];
#pragma unroll
            for (uint32_t i = 0; i < kKeyGroup; ++i) pos[i] = 0;
#pragma unroll
            for (uint32_t step = kTB / 2; step; step >>= 1) {
#pragma unroll
                for (uint32_t i = 0; i < kKeyGroup; ++i)
                    if (tile[pos[i] + step - 1] < key[g + i]) pos[i] += step;
            }
#pragma unroll
            for (uint32_t i = 0; i < kKeyGroup; ++i) {
                const pos_t val = tile[pos[i]];
                if (take[i]) {                                                   // tile[kTB-1] >= key, so pos is the lower bound
                    const uint64_t at = (uint64_t)wb + pos[i];
                    j[g + i] = at < pend ? (uint32_t)at : pend;
                    v[g + i] = val;
                    need &= ~(1u << (g + i));
                }
            }
        }
        wave_sync();                                                            // the tile is overwritten next
        if (!__any(need != 0)) break;
        // next tile; when even its last element is below the smallest open key, jump to that key's lower bound
        uint64_t kmin = ~0ull;
#pragma unroll
        for (uint32_t i = 0; i < kKeys; ++i)
            if ((need >> i) & 1) kmin = (uint64_t)key[i] < kmin ? (uint64_t)key[i] : kmin;
        kmin = wave_min(kmin);
        const uint64_t nwb = (uint64_t)wb + kTB;
        if (nwb >= pend) { wb = pend; break; }
        wb = (uint32_t)nwb;
        const uint64_t probe_at = nwb + kTB - 1 < pend ? nwb + kTB - 1 : (uint64_t)pend - 1;
        if ((uint64_t)P[probe_at] < kmin) wb = wave_kary_lower_bound(P, (uint32_t)probe_at + 1, pend, kmin);
    }
}

// start-to-start window of one element (position x) in the key domain of the lists; false if no position can be in it
template <typename pos_t>
__device__ __forceinline__ bool gap_window(uint64_t x, uint64_t lo, uint64_t hi, pos_t& tlo, pos_t& thi);
template <>
__device__ __forceinline__ bool gap_window<uint64_t>(uint64_t x, uint64_t lo, uint64_t hi, uint64_t& tlo, uint64_t& thi)
{
    tlo = sat_add(x, lo); thi = sat_add(x, hi);
    return true;
}
template <>
__device__ __forceinline__ bool gap_window<uint32_t>(uint64_t x, uint64_t lo, uint64_t hi, uint32_t& tlo, uint32_t& thi)
{
    const uint64_t a = sat_add(x, lo), b = sat_add(x, hi);
    tlo = (uint32_t)a;
    thi = b > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)b;
    return a <= 0xFFFFFFFFull;                                                   // positions fit 32 bits
}

// Slots are dealt to waves in contiguous runs so a wave can carry the segment it is in and the last answer
// of its searches from one 64-slot step to the next.
constexpr uint32_t kRun = 2048;

__device__ __forceinline__ uint32_t seg_find(const uint32_t* __restrict__ seg_begin, uint32_t nseg, uint64_t slot)
{
    uint32_t lo = 0, hi = nseg;                        // last p with seg_begin[p] <= slot
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (seg_begin[mid] <= slot) lo = mid; else hi = mid; }
    return lo;
}

// Feasibility of every slot is ONE BIT; "nearest feasible slot at or after j" is a successor query on a hierarchical
// bitset: level 0 = the feasibility bits, bit i of level l+1 = (word i of level l != 0).  A query reads one word per
// level it has to climb (almost always just level 0), so the per-level reverse scans of a 4-byte-per-slot array are gone.
constexpr uint32_t kBitLevels = 6;          // 64^6 slots > 2^32
struct FeasBits { const uint64_t* lvl[kBitLevels]; uint64_t words[kBitLevels]; };

// Kernels get level 0 as a plain pointer (the fast path) and the level table through device memory (the rare climb).
struct FeasRef { const uint64_t* lvl0; const FeasBits* table; };

__device__ __noinline__ uint32_t next_feasible_slow(const FeasBits* __restrict__ fb, uint64_t w0)
{
    // no set bit in word w0 behind the position: climb until a set bit is found, then descend to the lowest such bit
    uint64_t pos = w0 + 1;
    uint32_t l = 1;
    for (;; ++l) {
        if (l == kBitLevels) return kNone;
        const uint64_t w = pos >> 6;
        if (w >= fb->words[l]) return kNone;
        const uint64_t bits = fb->lvl[l][w] >> (pos & 63);
        if (bits) { pos += (uint64_t)__ffsll((long long)bits) - 1; break; }
        pos = w + 1;
    }
    while (l) {                                         // pos = index of a non-zero word of level l-1
        --l;
        const uint64_t bits = fb->lvl[l][pos];
        pos = pos * 64 + (uint64_t)__ffsll((long long)bits) - 1;
    }
    return pos < 0xFFFFFFFFull ? (uint32_t)pos : kNone;
}
__device__ __forceinline__ uint32_t next_feasible(const FeasRef& fb, uint64_t j)
{
    const uint64_t w0 = j >> 6;
    const uint64_t bits = fb.lvl0[w0] >> (j & 63);                   // almost always answers the query
    if (bits) return (uint32_t)(j + (uint64_t)__ffsll((long long)bits) - 1);
    return next_feasible_slow(fb.table, w0);
}
__device__ __forceinline__ bool is_feasible(const FeasRef& fb, uint64_t e) { return (fb.lvl0[e >> 6] >> (e & 63)) & 1; }

// one level of the summary: out word i = bitmap of (in[64 i + b] != 0)
__global__ void bits_summary_kernel(const uint64_t* __restrict__ in, uint64_t in_words, uint64_t w0, uint64_t w1 /* output word range */,
                                    uint64_t* __restrict__ out)
{
    for (uint64_t i = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < w1; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = 0;
        for (uint32_t b = 0; b < 64; ++b) {
            const uint64_t idx = i * 64 + b;
            if (idx < in_words && in[idx]) v |= 1ull << b;
        }
        out[i] = v;
    }
}

// single-sub-pattern queries (class dist 0): every element is a feasible chain that ends at itself
template <typename pos_t>
__global__ void __launch_bounds__(256) join_init_kernel(const pos_t* __restrict__ P, const uint32_t* __restrict__ seg_begin, uint32_t nseg,
                                                        const SegMeta* __restrict__ sm, uint64_t r0, uint64_t r1, uint64_t* __restrict__ fbits,
                                                        pos_t* __restrict__ endp)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t run_begin = r0 + wave * kRun;
    if (run_begin >= r1) return;
    const uint64_t run_end = run_begin + kRun < r1 ? run_begin + kRun : r1;
    uint32_t s_w = seg_find(seg_begin, nseg, run_begin);
    for (uint64_t base = run_begin; base < run_end; base += 64) {
        const uint64_t e = base + lane;
        if (e < run_end) {
            uint32_t s = s_w;
            while (seg_begin[s + 1] <= e) ++s;
            const SegMeta m = sm[s];
            endp[e] = P[phys_of(m, (uint32_t)e)];
            s_w = s;
        }
        const unsigned long long act = __ballot(e < run_end);
        if (lane == 0) fbits[base >> 6] = act;         // class ranges and runs are 64-aligned: one word per step
        s_w = __shfl(s_w, 0);                          // lane 0 is always in range and holds the smallest segment
    }
}

// link pass over the class [r0,r1) of slots that have `dist` sub-patterns after them.
// Steps that lie inside one segment (almost all of them: lists are long) keep the segment's metadata in registers,
// search as a wave behind the previous step's answer and have the next step's positions already in flight.
template <typename pos_t>
__global__ void __launch_bounds__(256) join_link_kernel(const pos_t* __restrict__ P, const uint32_t* __restrict__ seg_begin, uint32_t nseg,
                                                        const SegMeta* __restrict__ sm, uint64_t r0, uint64_t r1, uint32_t dist,
                                                        FeasRef fb, uint64_t* __restrict__ fbits_out,
                                                        pos_t* __restrict__ endp, uint32_t* __restrict__ link)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t run_begin = r0 + wave * kRun;
    if (run_begin >= r1) return;
    const uint64_t run_end = run_begin + kRun < r1 ? run_begin + kRun : r1;
    uint32_t s_w = seg_find(seg_begin, nseg, run_begin);          // wave-uniform: segment of `base`
    uint64_t seg_end = seg_begin[s_w + 1];
    SegMeta m = sm[s_w], nx = sm[m.next];
    uint32_t hint_seg = kNone, hint = 0;                           // answer of the last lane of the previous step and its segment
    uint64_t x_pre = 0;
    bool have_pre = false;
    for (uint64_t base = run_begin; base < run_end; base += 64) {
        if (base >= seg_end) {                                     // entered a new segment (skips empty ones)
            while (seg_begin[s_w + 1] <= base) ++s_w;
            seg_end = seg_begin[s_w + 1];
            m = sm[s_w]; nx = sm[m.next];
            have_pre = false;
        }
        const uint64_t e = base + lane;
        const bool active = e < run_end;
        const uint64_t step_last = base + 63 < run_end ? base + 63 : run_end - 1;
        uint32_t j = 0, s_last = s_w;
        if (step_last < seg_end) {
            // ---- fast path: one segment ---------------------------------------------------------------
            uint64_t x = have_pre ? x_pre : (active ? (uint64_t)P[phys_of(m, (uint32_t)e)] : 0);
            const uint64_t en = e + 64;                            // next step's position, in flight during the search
            have_pre = base + 64 < run_end && (base + 127 < run_end ? base + 127 : run_end - 1) < seg_end;
            if (have_pre) x_pre = en < run_end ? (uint64_t)P[phys_of(m, (uint32_t)en)] : 0;
            const uint64_t tlo = sat_add(x, nx.lo), thi = sat_add(x, nx.hi);
            if (hint_seg == s_w) j = wave_lower_bound(P, hint, nx.pend, tlo, active);
            else if (active) j = gallop_lower_bound(P, nx.pbegin, nx.pend, tlo);
            bool ok = false;
            if (active) {
                if (dist == 1) {                                   // next list is the last one: every element is feasible
                    if (j < nx.pend) { const uint64_t v = P[j]; ok = v <= thi; if (ok) { link[e] = j; endp[e] = (pos_t)v; } }
                } else if (j < nx.pend) {
                    uint32_t ej = next_feasible(fb, (uint64_t)nx.begin + (j - nx.pbegin));   // nearest feasible logical element at or after it
                    if (ej < nx.end && (uint64_t)P[phys_of(nx, ej)] <= thi) { ok = true; link[e] = ej; endp[e] = endp[ej]; }
                }
            }
            const unsigned long long okm = __ballot(ok);
            if (lane == 0) fbits_out[base >> 6] = okm;
        } else {
            // ---- a segment border inside the step: every lane looks its own segment up ------------------
            have_pre = false;
            uint32_t s = s_w;
            bool ok = false;
            if (active) {
                while (seg_begin[s + 1] <= e) ++s;
                const SegMeta ml = sm[s];
                const SegMeta nl = sm[ml.next];
                const uint64_t x = P[phys_of(ml, (uint32_t)e)];
                const uint64_t tlo = sat_add(x, nl.lo), thi = sat_add(x, nl.hi);
                j = gallop_lower_bound(P, (s == hint_seg) ? hint : nl.pbegin, nl.pend, tlo);
                if (e < ml.end) {                                  // padding slots between classes belong to no segment
                    if (dist == 1) {
                        ok = j < nl.pend && (uint64_t)P[j] <= thi;
                        if (ok) { link[e] = j; endp[e] = P[j]; }
                    } else if (j < nl.pend) {
                        uint32_t ej = next_feasible(fb, (uint64_t)nl.begin + (j - nl.pbegin));
                        if (ej < nl.end && (uint64_t)P[phys_of(nl, ej)] <= thi) { ok = true; link[e] = ej; endp[e] = endp[ej]; }
                    }
                }
            }
            const unsigned long long okm = __ballot(ok);
            if (lane == 0) fbits_out[base >> 6] = okm;
            s_last = __shfl(s, (int)(step_last - base));
        }
        hint_seg = s_last;
        hint = __shfl(j, (int)(step_last - base));
    }
}

// jump[e] for level-0 elements: first feasible element of list 0 at or after end(e)+end_len (kNone = none);
// slots of other levels get kNone so the tile pass can treat every slot alike.  Also the start of each chain.
// Same walk as the link pass; the list searched is the element's own (the answers lie behind the element itself).
template <typename pos_t>
__global__ void __launch_bounds__(256) join_jump_kernel(const pos_t* __restrict__ P, const uint32_t* __restrict__ seg_begin, uint32_t nseg,
                                                        const SegMeta* __restrict__ sm, const QueryMeta* __restrict__ qm, uint64_t r0,
                                                        uint64_t r1, FeasRef fb, const pos_t* __restrict__ endp,
                                                        uint32_t* __restrict__ jump, uint32_t* __restrict__ qstart)
{
    __shared__ pos_t s_tile[4][kTB];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t wave = uniform(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint64_t run_begin = r0 + wave * kRun;
    if (run_begin >= r1) return;
    const uint64_t run_end = run_begin + kRun < r1 ? run_begin + kRun : r1;
    pos_t* tile = s_tile[wv];
    uint32_t s = uniform(seg_find(seg_begin, nseg, run_begin));
    uint64_t cur = run_begin;
    while (cur < run_end) {
        while (seg_begin[s + 1] <= cur) ++s;
        const SegMeta m = sm[s];
        const uint64_t seg_end = seg_begin[s + 1];                     // beyond m.end only behind the last segment of a class
        if (m.level != 0) {
            // ---- not a first sub-pattern: no chain passes through these slots --------------------------------
            const uint64_t piece_end = run_end < seg_end ? run_end : seg_end;
            for (uint64_t a = cur + lane; a < piece_end; a += 64) jump[a] = kNone;
            cur = piece_end;
        } else if (m.end - m.begin >= kMinPiece && cur < m.end) {
            // ---- long list: blocks of kBlk slots against tiles of the same list ------------------------------
            const uint64_t piece_end = run_end < m.end ? run_end : (uint64_t)m.end;
            const uint64_t end_len = qm[m.query].end_len;
            if (cur == m.begin && lane == 0) { const uint32_t me = next_feasible(fb, cur); qstart[m.query] = me < m.end ? me : kNone; }
            uint32_t fence = 0;
            for (uint64_t blk0 = cur; blk0 < piece_end; blk0 += kBlk) {
                const uint64_t blk1 = blk0 + kBlk < piece_end ? blk0 + kBlk : piece_end;
                pos_t key[kKeys], v[kKeys];
                uint32_t j[kKeys];
                uint32_t need = 0;
#pragma unroll
                for (uint32_t i = 0; i < kKeys; ++i) {
                    const uint64_t a = blk0 + lane + 64 * i;
                    key[i] = 0;
                    if (a < blk1 && is_feasible(fb, a)) {                                  // feasible start
                        pos_t unused;
                        if (gap_window<pos_t>((uint64_t)endp[a], end_len, end_len, key[i], unused)) need |= 1u << i;
                    }
                }
                const uint32_t own = phys_of(m, (uint32_t)blk0) + 1;                       // every answer lies behind its own element
                fence = fence > own ? fence : own;
                const uint32_t asked = need;
                tile_lower_bounds<pos_t>(P, fence, m.pend, tile, key, need, j, v);
                uint32_t jm = fence;
#pragma unroll
                for (uint32_t i = 0; i < kKeys; ++i) {
                    const uint64_t a = blk0 + lane + 64 * i;
                    if (a < blk1) {
                        uint32_t out = kNone;
                        if ((asked >> i) & 1) {
                            jm = j[i] > jm ? j[i] : jm;
                            if (j[i] < m.pend) {
                                const uint32_t ej = next_feasible(fb, (uint64_t)m.begin + (j[i] - m.pbegin));
                                if (ej < m.end) out = ej;
                            }
                        }
                        jump[a] = out;
                    }
                }
                // ends ascend along a list: the largest answer of this block is a fence for the next one
                jm = wave_max_u32(jm);
                fence = jm;
            }
            cur = piece_end;
        } else {
            // ---- short lists (and the slots between two classes): 64 slots, every lane on its own --------------
            const uint64_t e = cur + lane;
            const uint64_t grp_end = cur + 64 < run_end ? cur + 64 : run_end;
            if (e < grp_end) {
                uint32_t sl = s;
                while (seg_begin[sl + 1] <= e) ++sl;
                const SegMeta ml = sm[sl];
                uint32_t out = kNone;
                if (ml.level == 0 && e < ml.end) {
                    if (is_feasible(fb, e)) {
                        const uint64_t lim = sat_add((uint64_t)endp[e], qm[ml.query].end_len);
                        const uint32_t jp = gallop_lower_bound(P, phys_of(ml, (uint32_t)e) + 1, ml.pend, lim);
                        if (jp < ml.pend) {
                            const uint32_t ej = next_feasible(fb, (uint64_t)ml.begin + (jp - ml.pbegin));
                            if (ej < ml.end) out = ej;
                        }
                    }
                    if ((uint32_t)e == ml.begin) { const uint32_t me = next_feasible(fb, e); qstart[ml.query] = me < ml.end ? me : kNone; }
                }
                jump[e] = out;
            }
            cur = grp_end;
        }
    }
}

// The chain a -> jump[a] -> ... of a query is resolved in three data-parallel passes instead of one serial walk:
//   tiles : inside every tile of kTile slots, pointer doubling in LDS gives each slot its exit (first chain
//           element beyond the tile) and the number of chain elements it covers inside the tile;
//   walk  : one lane per query hops tile to tile (a heavy query costs |list|/kTile dependent loads, not |matches|),
//           leaving one record per tile visited and the query's match count;
//   emit  : one lane per record lists the matches inside its tile.
constexpr uint32_t kTile = 1024;

// per-slot state of the doubling, one word: [0,10) next element inside the tile, bit 10 = chain left the tile,
// [11,21) chain elements covered so far minus one, [21,31) last chain element inside the tile
__global__ void __launch_bounds__(256) chain_tiles_kernel(const uint32_t* __restrict__ jump, uint64_t t0 /* multiple of kTile */,
                                                          uint64_t r1, uint2* __restrict__ xh)
{
    static_assert(kTile == 1024, "the packed word holds 10-bit tile offsets");
    __shared__ uint32_t s_ext[kTile];
    __shared__ uint32_t s_st[2][kTile];
    constexpr uint32_t kDone = 1u << 10;
    const uint64_t base = t0 + (uint64_t)blockIdx.x * kTile;
    const uint64_t tile_end = base + kTile;
    uint32_t st[4];
    bool open = false;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t li = threadIdx.x + 256 * r;
        const uint64_t e = base + li;
        const uint32_t j = e < r1 ? jump[e] : kNone;
        const bool inside = j != kNone && (uint64_t)j < tile_end;
        s_ext[li] = j;                                   // where the chain goes when this is its last element inside the tile
        st[r] = (inside ? (uint32_t)(j - base) : kDone) | (li << 21);
        s_st[0][li] = st[r];
        open |= inside;
    }
    uint32_t cur = 0;
    // jump[e] > e, so a chain inside a tile has fewer than 2^10 elements: at most 10 doublings
    for (uint32_t round = 0; round < 10 && __syncthreads_or(open); ++round) {
        open = false;
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t li = threadIdx.x + 256 * r;
            if (!(st[r] & kDone)) {
                const uint32_t nx = s_st[cur][st[r] & 1023u];
                const uint32_t hops = ((st[r] >> 11) & 1023u) + ((nx >> 11) & 1023u) + 1;
                st[r] = (nx & 0x7FFu) | (hops << 11) | (nx & (1023u << 21));
                open |= !(st[r] & kDone);
            }
            s_st[cur ^ 1][li] = st[r];
        }
        cur ^= 1;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t li = threadIdx.x + 256 * r;
        const uint64_t e = base + li;
        if (e < r1) xh[e] = make_uint2(s_ext[st[r] >> 21], ((st[r] >> 11) & 1023u) + 1);
    }
}

__global__ void chain_walk_kernel(const SegMeta* __restrict__ sm, const QueryMeta* __restrict__ qm, uint32_t nq,
                                  const uint32_t* __restrict__ qstart, const uint2* __restrict__ xh, const uint32_t* __restrict__ rec_begin,
                                  uint2* __restrict__ records, uint32_t* __restrict__ rec_count, unsigned long long* __restrict__ counts)
{
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const QueryMeta Q = qm[q];
    unsigned long long n_match = 0;
    uint32_t nrec = 0;
    if (Q.seg0 != kNone) {
        const uint32_t mbegin = sm[Q.seg0].begin;
        uint2* rec = records + rec_begin[q];
        uint32_t cur = qstart[q];
        while (cur != kNone) {
            uint2 v = xh[cur];
            rec[nrec++] = make_uint2(cur, mbegin + (uint32_t)n_match);
            n_match += v.y;
            cur = v.x;
        }
    }
    rec_count[q] = nrec;
    counts[q] = n_match;
}

__global__ void chain_emit_kernel(const uint32_t* __restrict__ rec_begin, const uint32_t* __restrict__ rec_count,
                                  const uint2* __restrict__ records, uint32_t total_rec_slots, const uint32_t* __restrict__ rec_query,
                                  const uint32_t* __restrict__ jump, uint32_t* __restrict__ mlist)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < total_rec_slots; r += gridDim.x * blockDim.x) {
        uint32_t q = rec_query[r];                       // slot r belongs to query q; used only if r - rec_begin[q] < rec_count[q]
        if (r - rec_begin[q] >= rec_count[q]) continue;
        uint2 rc = records[r];
        uint32_t cur = rc.x, out = rc.y;
        const uint32_t tile_end = (cur / kTile + 1) * kTile;
        while (cur != kNone && cur < tile_end) {
            mlist[out++] = cur;
            cur = jump[cur];
        }
    }
}

// tuples of every match: walk the links from the level-0 element.  One thread per (query, match) slot of list 0.
template <typename pos_t>
__global__ void __launch_bounds__(256) join_gather_kernel(const pos_t* __restrict__ P, const uint32_t* __restrict__ seg_begin, uint32_t nseg,
                                                          const SegMeta* __restrict__ sm, const QueryMeta* __restrict__ qm, uint64_t r0,
                                                          uint64_t r1, const uint32_t* __restrict__ link, const uint32_t* __restrict__ mlist,
                                                          const unsigned long long* __restrict__ counts, uint64_t* __restrict__ out_first,
                                                          uint64_t* __restrict__ out_tuples, unsigned long long* __restrict__ checksum)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t run_begin = r0 + wave * kRun;
    if (run_begin >= r1) return;
    const uint64_t run_end = run_begin + kRun < r1 ? run_begin + kRun : r1;
    uint32_t s_w = seg_find(seg_begin, nseg, run_begin);
    unsigned long long local = 0;
    for (uint64_t base = run_begin; base < run_end; base += 64) {
        const uint64_t e = base + lane;
        uint32_t s = s_w;
        if (e < run_end) {
            while (seg_begin[s + 1] <= e) ++s;
            const SegMeta m = sm[s];
            const uint64_t t = e - m.begin;
            if (m.level == 0 && t < counts[m.query]) {
                const QueryMeta Q = qm[m.query];
                uint32_t el = mlist[e];                                // logical element of level 0
                uint64_t first = P[phys_of(m, el)];
                out_first[Q.out_first + t] = first;
                local += first;
                uint64_t* tp = out_tuples + Q.out_tuple + t * Q.k;
                tp[0] = first;
                if (Q.k > 1) {
                    uint32_t cur = link[el];
                    uint32_t sg = m.next;
                    for (uint32_t i = 1; i < Q.k; ++i) {
                        const SegMeta mi = sm[sg];
                        if (mi.dist == 0) { tp[i] = P[cur]; }          // link of a dist-1 element is a physical index
                        else { tp[i] = P[phys_of(mi, cur)]; cur = link[cur]; sg = mi.next; }
                    }
                }
            }
        }
        s_w = __shfl(s, 0);
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if (lane == 0 && local) atomicAdd(checksum, local);
}

inline uint32_t runs_grid(uint64_t slots) { return (uint32_t)((((slots + kRun - 1) / kRun) + 3) / 4); }   // 4 waves per workgroup

inline uint32_t grid_for(uint64_t n, uint32_t cap = 16384) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, cap)); }

struct Arena {
    uint8_t* base; uint64_t size; uint64_t used = 0;
    template <class T> T* take(uint64_t count)
    {
        uint64_t bytes = align_up(count * sizeof(T), 256);
        if (used + bytes > size) return nullptr;
        T* p = reinterpret_cast<T*>(base + used);
        used += bytes;
        return p;
    }
};

struct Plan {                       // host view of the batch after backward search
    std::vector<uint64_t> occ;      // per sub-pattern, 0 for dead queries
    std::vector<uint32_t> did;      // per sub-pattern: distinct-interval id (valid when occ > 0)
    std::vector<uint64_t> dl, docc; // per distinct interval: left border, size
};

template <typename pos_t> constexpr uint64_t kPhysScratchPerElem() { return 20; }   // sweep scratch; the sorted lists reuse it
constexpr uint64_t kJoinBytesPerSlot = 4 + 8 + 1;      // link, endp(<=8), feasibility bits + summaries (any slot)
constexpr uint64_t kJoinBytesPerSlot0 = 4 + 4 + 8 + 1; // jump, mlist, (exit,hops), chain records (slots of list 0)

// ---- sort of all lists at once: one radix sort of (list, position) keys instead of one sort per list -------------
template <typename pos_t>
__global__ void sort_compose_kernel(const pos_t* __restrict__ P, const uint64_t* __restrict__ off /* [nd+1] */, uint64_t nd, uint64_t total,
                                    uint32_t pos_bits, uint64_t* __restrict__ keys)
{
    __shared__ uint64_t s_first;
    for (uint64_t base = (uint64_t)blockIdx.x * 256; base < total; base += (uint64_t)gridDim.x * 256) {
        if (threadIdx.x == 0) {
            uint64_t lo = 0, hi = nd;                      // last list with off[l] <= base
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (off[mid] <= base) lo = mid; else hi = mid; }
            s_first = lo;
        }
        __syncthreads();
        const uint64_t t = base + threadIdx.x;
        if (t < total) {
            uint64_t l = s_first;
            while (off[l + 1] <= t) ++l;                   // empty lists are skipped too
            keys[t] = (l << pos_bits) | (uint64_t)P[t];
        }
        __syncthreads();
    }
}

template <typename pos_t>
__global__ void sort_narrow_kernel(const uint64_t* __restrict__ keys, uint64_t total, uint32_t pos_bits, pos_t* __restrict__ P)
{
    const uint64_t mask = (1ull << pos_bits) - 1;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) P[t] = (pos_t)(keys[t] & mask);
}

// ---- physical pass: locate + sort every distinct interval used by queries [Q0,Q1) -------------------
template <typename pos_t>
vlg_status build_physical(const vlg_index* idx, vlg_workspace* ws, vlg_result* res, const std::vector<uint32_t>& dlist /* distinct ids */,
                          const Plan& pl, Arena& A, pos_t*& P_out, std::vector<uint32_t>& poff /* per distinct id -> offset (size dl) */,
                          uint64_t& Tphys, size_t sort_tmp, unsigned long long* d_stats, pos_t*& Pc_out, uint64_t& pc_cap, bool share_trails)
{
    hipStream_t st = ws->stream;
    const uint32_t nd = (uint32_t)dlist.size();
    std::vector<uint64_t> off64(nd + 1), lh(nd);
    std::vector<uint32_t> off32(nd + 1);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < nd; ++i) {
        off64[i] = acc; off32[i] = (uint32_t)acc; lh[i] = pl.dl[dlist[i]];
        poff[dlist[i]] = (uint32_t)acc;
        acc += pl.docc[dlist[i]];
    }
    off64[nd] = acc; off32[nd] = (uint32_t)acc;
    Tphys = acc;
    P_out = nullptr;
    Pc_out = nullptr;
    pc_cap = 0;
    if (!acc) return VLG_OK;
    const unsigned bits = std::max(1u, bit_width64(idx->hdr.n >= 2 ? idx->hdr.n - 2 : 0));   // the largest position is n - 2 (n - 1 is the sentinel)
    const bool use_sweep = ws->sweep && acc >= ws->sweep_min && idx->hdr.n <= (1ull << (sizeof(pos_t) == 4 ? 32 : 33));
    pos_t* Pa = A.take<pos_t>(acc);
    // scratch of the sweep (20 B per element); the sorted lists Pb reuse it once locate is done
    uint8_t* scratch = A.take<uint8_t>(acc * kPhysScratchPerElem<pos_t>());
    uint64_t* d_off64 = A.take<uint64_t>(nd + 1);
    uint32_t* d_off32 = A.take<uint32_t>(nd + 1);
    uint64_t* d_lh = A.take<uint64_t>(nd);
    unsigned long long* d_counter = A.take<unsigned long long>(1);
    void* d_tmp = A.take<uint8_t>(sort_tmp + 256);
    if (!d_tmp) return fail(VLG_E_INTERNAL, "arena carve failed (physical)");
    pos_t* Pb = reinterpret_cast<pos_t*>(scratch);
    VLG_HIP_TRY(hipMemcpyAsync(d_off64, off64.data(), (nd + 1) * 8, hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(d_off32, off32.data(), (nd + 1) * 4, hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(d_lh, lh.data(), nd * 8, hipMemcpyHostToDevice, st));
    if (use_sweep) {
        const uint64_t cap = std::min<uint64_t>(acc, sweep_batch_max<pos_t>());
        uint64_t* trail = nullptr;
        uint64_t* rec = nullptr;
        if (share_trails) {                                                       // (trails are shared inside one sweep)
            trail = A.take<uint64_t>(idx->hdr.n);
            rec = A.take<uint64_t>(acc);
            if (!trail || !rec) return fail(VLG_E_INTERNAL, "arena carve failed (trail table)");
        }
        uint64_t* val_a = reinterpret_cast<uint64_t*>(scratch);
        uint64_t* val_b = val_a + cap;
        uint16_t* key_a = reinterpret_cast<uint16_t*>(val_b + cap);
        uint16_t* key_b = key_a + cap;
        SweepTimer timer(ws);
        if (vlg_status s = launch_locate_sweep<pos_t>(idx->view, d_lh, d_off64, nd, acc, Pa, val_a, val_b, key_a, key_b,
                                               d_tmp, sort_tmp, d_counter, d_stats, ws->sweep_tail, st, &timer, trail, rec)) return s;
    } else {
        {
            Timed t(ws, KS_EXPAND, 0);
            if (vlg_status s = launch_expand<pos_t>(d_lh, d_off64, nd, acc, Pa, nullptr, st)) return s;
        }
        {
            Timed t(ws, KS_LOCATE, 0);
            if (vlg_status s = launch_locate<pos_t>(idx->view, Pa, acc, d_stats, st)) return s;
        }
    }
    // sort every occurrence list ascending (std::sort, index_sasearch.hpp:80)
    const unsigned list_bits = bit_width64(nd);
    const bool global_sort = acc >= ws->global_sort_min && bits + list_bits <= 64;
    uint64_t dead_bytes;                                  // free bytes behind the sorted lists (the survivors of the window filter go there)
    if (global_sort) {
        // one radix sort of (list, position) keys: the passes stream the whole batch whatever the list sizes are
        uint64_t* ka = reinterpret_cast<uint64_t*>(scratch);
        uint64_t* kb = ka + acc;
        Timed t(ws, KS_SORT, 2ull * acc * sizeof(pos_t));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sort_compose_kernel<pos_t>), dim3(grid_for(acc, 32768)), dim3(256), 0, st, Pa, d_off64, (uint64_t)nd, acc, bits, ka);
        rocprim::double_buffer<uint64_t> keys(ka, kb);
        size_t tb = sort_tmp;
        VLG_HIP_TRY(rocprim::radix_sort_keys(d_tmp, tb, keys, acc, 0, bits + list_bits, st));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sort_narrow_kernel<pos_t>), dim3(grid_for(acc, 32768)), dim3(256), 0, st, keys.current(), acc, bits, Pa);
        VLG_HIP_TRY(hipGetLastError());
        P_out = Pa;
        dead_bytes = (uint64_t)(scratch - reinterpret_cast<uint8_t*>(Pa)) + acc * kPhysScratchPerElem<pos_t>() - acc * sizeof(pos_t);
    } else {
        Timed t(ws, KS_SORT, 2ull * acc * sizeof(pos_t));
        size_t tb = sort_tmp;
        VLG_HIP_TRY(rocprim::segmented_radix_sort_keys(d_tmp, tb, Pa, Pb, (unsigned)acc, nd, d_off32, d_off32 + 1, 0, bits, st));
        P_out = Pb;
        dead_bytes = acc * kPhysScratchPerElem<pos_t>() - acc * sizeof(pos_t);
    }
    VLG_HIP_TRY(hipStreamSynchronize(st));        // host staging vectors go out of scope
    {
        const uint64_t pc_first = align_up(acc, 64);
        if (dead_bytes > (pc_first - acc + 64) * sizeof(pos_t) && pc_first < 0xFFFFFF00ull) {
            Pc_out = P_out + pc_first;
            pc_cap = std::min<uint64_t>(dead_bytes / sizeof(pos_t) - (pc_first - acc) - 64, 0xFFFFFF00ull - pc_first);
        }
    }
    res->sum.located_occurrences += acc;
    return VLG_OK;
}

// =============================================================================================
// Window filter (semi-join reduction of the lists of one query)
// =============================================================================================
// Most elements of a long occurrence list can be in no match at all: an element of sub-pattern i matters only if some
// element of sub-pattern i+1 lies inside its gap window, and so on to the last sub-pattern -- and likewise towards the
// first one.  Dropping the others changes no match (a match is a chain of elements that all have such neighbours) but
// shrinks the lists the join evaluates element by element.  The test is made on blocks of 2^g text positions: a
// backward sweep (last sub-pattern to first) marks, in a block bitmap per query, the blocks in which an element of the
// previous sub-pattern could start a chain; the elements of that list inside marked blocks stay active and mark blocks
// for the list before them.  A forward sweep does the same from the surviving elements of the first list.  Every pass
// streams sorted lists (coalesced) and touches a bitmap that stays in L2; the survivors are compacted into private
// lists of the query, which the join then uses in place of the shared ones.
struct RSeg {                 // one per sub-pattern of a filtered query
    uint32_t pbegin, pend;    // physical list
    uint64_t lo, hi;          // gap bounds to the previous sub-pattern (level > 0)
    uint64_t nlo, nhi;        // gap bounds to the next sub-pattern (dist > 0)
    uint64_t abit;            // first activity bit of the segment (64-aligned); the last sub-pattern has none (~0)
    uint32_t fq;              // filtered-query ordinal: selects the query's pair of block bitmaps
    uint32_t level, dist;
    uint32_t pad;
};

struct RPass {
    int32_t test_buf;         // bitmap an element's block is looked up in (-1: none)
    int32_t scatter_buf;      // bitmap the windows of the active elements are marked in
    int32_t dir;              // -1: windows towards the previous sub-pattern, +1: towards the next, 0: no marking
    int32_t use_bits;         // start from the activity bits of an earlier pass
    int32_t write_bits;
};

// Block ranges to mark, merged on the way: the ranges a wave produces ascend (sorted list, one pair of bounds), so
// overlapping ones fuse into runs and a run is written once, a word per lane, when the next range starts beyond it.
// With a window (kMarkWin words of LDS per wave) the words are combined on chip first and reach the bitmap once when the
// ranges have moved past them: marks of sparse survivors cost an LDS atomic instead of a 64-byte request each.
constexpr uint32_t kMarkWin = 32;
struct MarkRun {
    uint64_t* bm;
    uint64_t* win = nullptr;              // LDS, zeroed, private to the wave (null: every word goes to memory directly)
    uint32_t wbase = 0;
    uint32_t S = 0, E = 0;
    bool open = false;
    // write the window out and move it to start at word w
    __device__ __forceinline__ void slide(uint32_t w)
    {
        const uint32_t lane = threadIdx.x & 63;
        wave_sync();
        if (lane < kMarkWin) {
            const uint64_t m = win[lane];
            if (m) { atomicOr((unsigned long long*)(bm + wbase + lane), (unsigned long long)m); win[lane] = 0; }
        }
        wave_sync();
        wbase = w;
    }
    // make room for bits up to word w1 of ranges that start at word w0 or later; false if they do not fit the window
    __device__ __forceinline__ bool fits(uint32_t w0, uint32_t w1)
    {
        if (!win) return false;
        if (w1 >= wbase + kMarkWin || w0 < wbase) slide(w0);
        return w1 < wbase + kMarkWin;
    }
    __device__ __forceinline__ void flush()
    {
        if (!open) return;
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t w0 = S >> 6, w1 = E >> 6;
        const bool local = fits(w0, w1);
        for (uint32_t w = w0 + lane; w <= w1; w += 64) {
            const uint32_t b0 = w == w0 ? (S & 63) : 0, b1 = w == w1 ? (E & 63) : 63;
            const uint64_t m = (~0ull << b0) & (~0ull >> (63 - b1));
            if (local) win[w - wbase] |= m;                                         // one lane per word
            else atomicOr((unsigned long long*)(bm + w), (unsigned long long)m);
        }
        open = false;
    }
    __device__ __forceinline__ void finish()
    {
        flush();
        if (win) slide(0);
    }
    // ranges [sb,eb] of the lanes with `on`, ascending with the lane
    __device__ __forceinline__ void add(uint32_t sb, uint32_t eb, bool on)
    {
        const uint32_t lane = threadIdx.x & 63;
        const unsigned long long amask = __ballot(on);
        if (!amask) return;
        const unsigned long long below = amask & ((1ull << lane) - 1ull);
        const int prev = below ? 63 - __clzll((long long)below) : 0;
        const uint32_t e_prev = __shfl(eb, prev);
        const bool head = on && (!below || sb > e_prev + 1);                       // first lane of a run inside the wave
        unsigned long long H = __ballot(head);
        if (__popcll(H) > 2) {
            // many short runs (sparse survivors): every head lane writes its own run, all of them at once
            flush();
            const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1)) << (lane + 1);
            const unsigned long long in_run = above ? amask & ((1ull << (__ffsll((long long)above) - 1)) - 1ull) : amask;
            const uint32_t e = __shfl(eb, in_run ? 63 - __clzll((long long)in_run) : 0);
            const uint32_t first_w = uniform(__shfl(sb, __ffsll((long long)amask) - 1)) >> 6;
            const uint32_t last_w = uniform(__shfl(eb, 63 - __clzll((long long)amask))) >> 6;
            const bool local = fits(first_w, last_w);
            if (head) {
                const uint32_t w0 = sb >> 6, w1 = e >> 6;
                for (uint32_t w = w0; w <= w1; ++w) {                             // no look first: nothing here waits for memory
                    const uint32_t b0 = w == w0 ? (sb & 63) : 0, b1 = w == w1 ? (e & 63) : 63;
                    const unsigned long long m = (~0ull << b0) & (~0ull >> (63 - b1));
                    if (local) atomicOr((unsigned long long*)(win + (w - wbase)), m);
                    else atomicOr((unsigned long long*)(bm + w), m);
                }
            }
            return;
        }
        while (H) {
            const int h = __ffsll((long long)H) - 1;
            H &= H - 1;
            const unsigned long long in_run = H ? amask & ((1ull << (__ffsll((long long)H) - 1)) - 1ull) : amask;
            const uint32_t s = uniform(__shfl(sb, h)), e = uniform(__shfl(eb, 63 - __clzll((long long)in_run)));
            if (open && s <= E + 1) { E = e > E ? e : E; }
            else { flush(); S = s; E = e; open = true; }
        }
    }
};

// task of a run: last t with run0[t] <= run
__device__ __forceinline__ uint32_t task_find(const uint64_t* __restrict__ run0, uint32_t ntasks, uint64_t run)
{
    uint32_t lo = 0, hi = ntasks;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (run0[mid] <= run) lo = mid; else hi = mid; }
    return lo;
}
// the same for a whole wave asking about one run: three rounds of 64 probes instead of a chain of dependent loads
__device__ __forceinline__ uint32_t wave_task_find(const uint64_t* __restrict__ run0, uint32_t ntasks, uint64_t run)
{
    return uniform(wave_kary_lower_bound<uint64_t>(run0, 0, ntasks + 1, run + 1)) - 1;
}

constexpr uint32_t kFilterGroups = 8;         // 64-element groups of a run in flight per wave

template <typename pos_t>
__global__ void __launch_bounds__(256) filter_pass_kernel(const pos_t* __restrict__ P, const RSeg* __restrict__ segs,
                                                          const uint32_t* __restrict__ task_seg, const uint64_t* __restrict__ task_run0,
                                                          uint32_t ntasks, uint64_t* __restrict__ bitmaps, uint64_t nbw, uint32_t g,
                                                          uint64_t nblocks, uint64_t* __restrict__ abits, RPass ps)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t run = uniform(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (run >= task_run0[ntasks]) return;
    const uint32_t t = wave_task_find(task_run0, ntasks, run);
    const RSeg sg = segs[task_seg[t]];
    const uint64_t len = sg.pend - sg.pbegin;
    const uint64_t off0 = (run - task_run0[t]) * kRun;
    const uint64_t off1 = off0 + kRun < len ? off0 + kRun : len;
    const uint64_t* bm_test = ps.test_buf >= 0 ? bitmaps + ((uint64_t)sg.fq * 2 + (uint32_t)ps.test_buf) * nbw : nullptr;
    const bool has_bits = sg.abit != ~0ull;
    const bool mark = ps.dir < 0 ? sg.level > 0 : (ps.dir > 0 ? sg.dist >= 2 : false);
    __shared__ uint64_t s_win[4][kMarkWin];
    MarkRun mr;
    mr.bm = bitmaps + ((uint64_t)sg.fq * 2 + (uint32_t)ps.scatter_buf) * nbw;
    mr.win = s_win[threadIdx.x >> 6];
    if (lane < kMarkWin) mr.win[lane] = 0;
    wave_sync();
    for (uint64_t base = off0; base < off1; base += 64 * kFilterGroups) {
        uint64_t x[kFilterGroups];
        bool act[kFilterGroups];
#pragma unroll
        for (uint32_t i = 0; i < kFilterGroups; ++i) {
            const uint64_t gb = base + 64 * i, idx = gb + lane;
            uint64_t cur = gb < off1 ? ~0ull : 0;
            if (ps.use_bits && cur) cur = abits[(sg.abit + gb) >> 6];          // groups without a survivor read nothing of the list
            act[i] = idx < off1 && ((cur >> lane) & 1);
            x[i] = 0;
            if (act[i]) x[i] = P[sg.pbegin + idx];
        }
        if (bm_test) {
#pragma unroll
            for (uint32_t i = 0; i < kFilterGroups; ++i)
                if (act[i]) { const uint64_t blk = x[i] >> g; act[i] = (bm_test[blk >> 6] >> (blk & 63)) & 1; }
        }
#pragma unroll
        for (uint32_t i = 0; i < kFilterGroups; ++i) {
            const uint64_t gb = base + 64 * i;
            const unsigned long long mask = __ballot(act[i]);
            if (ps.write_bits && has_bits && lane == 0 && gb < off1) abits[(sg.abit + gb) >> 6] = mask;
            if (mark && mask) {
                uint32_t sb = 0, eb = 0;
                bool on = act[i];
                if (ps.dir < 0) {                                               // positions p with lo <= x - p <= hi
                    if (x[i] < sg.lo) on = false;
                    else { eb = (uint32_t)((x[i] - sg.lo) >> g); sb = (uint32_t)((x[i] > sg.hi ? x[i] - sg.hi : 0) >> g); }
                } else {                                                        // positions p with nlo <= p - x <= nhi
                    const uint64_t a = sat_add(x[i], sg.nlo) >> g, b = sat_add(x[i], sg.nhi) >> g;
                    if (a >= nblocks) on = false;
                    else { sb = (uint32_t)a; eb = (uint32_t)(b >= nblocks ? nblocks - 1 : b); }
                }
                mr.add(sb, eb, on);
            }
        }
    }
    mr.finish();
}

// Index ranges [i0,i1) (relative to pbegin) of the list elements inside the position windows [a,b] of the lanes with `on`,
// for kPivotGroups groups of 64 windows at once; `on` is cleared for empty ranges.  Everything runs in lockstep over the
// groups so that every round has one load per group in flight instead of one in all: the windows of a group ascend
// with the lane, so first 2 x kPivotGroups wave-wide 64-ary searches bracket each group's answers between the lower bounds
// of its smallest a and its largest b + 1, then every lane bisects its own a and b + 1 inside its group's bracket.
constexpr uint32_t kPivotGroups = 4;
template <typename pos_t>
__device__ __forceinline__ void pivot_ranges(const pos_t* __restrict__ P, uint32_t pbegin, uint32_t pend, const uint64_t (&a)[kPivotGroups],
                                             const uint64_t (&b)[kPivotGroups], bool (&on)[kPivotGroups], uint32_t (&i0)[kPivotGroups],
                                             uint32_t (&i1)[kPivotGroups])
{
    constexpr uint32_t G = kPivotGroups;
    const uint32_t lane = threadIdx.x & 63;
    // ---- brackets: searches 0..G-1 for min a, G..2G-1 for max b + 1, over the whole list ------------------------
    uint64_t key[2 * G];
    uint32_t A[2 * G], B[2 * G];
    bool any[G];
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        const unsigned long long m = __ballot(on[g]);
        any[g] = m != 0;
        const int first = m ? __ffsll((long long)m) - 1 : 0, last = m ? 63 - __clzll((long long)m) : 0;
        key[g] = uniform(__shfl(a[g], first));
        const uint64_t bmax = uniform(__shfl(b[g], last));
        key[G + g] = bmax == ~0ull ? ~0ull : bmax + 1;
        A[g] = A[G + g] = pbegin;
        B[g] = B[G + g] = any[g] ? pend : pbegin;           // nothing to search for an empty group
    }
    for (;;) {
        bool more = false;
#pragma unroll
        for (uint32_t s = 0; s < 2 * G; ++s) more |= B[s] - A[s] > 64;
        if (!more) break;
        uint64_t v[2 * G];
        bool in[2 * G];
#pragma unroll
        for (uint32_t s = 0; s < 2 * G; ++s) {
            const uint32_t step = (B[s] - A[s] + 63) / 64;
            const uint64_t idx = (uint64_t)A[s] + (uint64_t)(lane + 1) * step - 1;
            in[s] = B[s] - A[s] > 64 && idx < B[s];
            v[s] = in[s] ? (uint64_t)P[idx] : 0;
        }
#pragma unroll
        for (uint32_t s = 0; s < 2 * G; ++s) {
            if (B[s] - A[s] > 64) {
                const uint32_t step = (B[s] - A[s] + 63) / 64;
                const uint32_t c = (uint32_t)__popcll(__ballot(in[s] && v[s] < key[s]));
                const uint64_t na = (uint64_t)A[s] + (uint64_t)c * step, nb = (uint64_t)A[s] + (uint64_t)(c + 1) * step - 1;
                A[s] = (uint32_t)na;
                B[s] = nb < B[s] ? (uint32_t)nb : B[s];
            }
        }
    }
    uint32_t lo[G], hi[G];
    {
        uint64_t v[2 * G];
        bool in[2 * G];
#pragma unroll
        for (uint32_t s = 0; s < 2 * G; ++s) { in[s] = A[s] + lane < B[s]; v[s] = in[s] ? (uint64_t)P[A[s] + lane] : 0; }
#pragma unroll
        for (uint32_t s = 0; s < 2 * G; ++s) {
            const uint32_t r = A[s] + (uint32_t)__popcll(__ballot(in[s] && v[s] < key[s]));
            if (s < G) lo[s] = r; else hi[s - G] = r;
        }
    }
    // ---- every lane inside its group's bracket: lower bounds of a and of b + 1 -------------------------------------
    uint32_t l0[G], r0[G], l1[G], r1[G];
    uint32_t widest = 0;
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        if (hi[g] < lo[g]) hi[g] = lo[g];
        l0[g] = l1[g] = lo[g];
        r0[g] = r1[g] = on[g] ? hi[g] : lo[g];
        widest = hi[g] - lo[g] > widest ? hi[g] - lo[g] : widest;
    }
    for (uint32_t w = uniform(widest); w; w >>= 1) {        // bit_width(widest) rounds bisect any range of that size
        uint64_t v0[G], v1[G];
        uint32_t m0[G], m1[G];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            m0[g] = l0[g] + ((r0[g] - l0[g]) >> 1);
            m1[g] = l1[g] + ((r1[g] - l1[g]) >> 1);
            v0[g] = l0[g] < r0[g] ? (uint64_t)P[m0[g]] : 0;
            v1[g] = l1[g] < r1[g] ? (uint64_t)P[m1[g]] : 0;
        }
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            if (l0[g] < r0[g]) { if (v0[g] < a[g]) l0[g] = m0[g] + 1; else r0[g] = m0[g]; }
            if (l1[g] < r1[g]) { if (v1[g] <= b[g]) l1[g] = m1[g] + 1; else r1[g] = m1[g]; }
        }
    }
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        i0[g] = l0[g] - pbegin;
        i1[g] = l1[g] - pbegin;
        on[g] = on[g] && l0[g] < l1[g];
    }
}

// Pivot mode: when one list of the query is much shorter than the others, the survivors are found from its elements
// outwards instead of streaming the long lists.  A lane takes one element of the pivot list (kPivotGroups of them, one per
// 64-element group of the wave's run) and follows it level by level: the elements of the neighbouring list inside its gap
// window form an index range, which is marked in that list's activity bits; the hull of the range's positions is the
// "element" followed to the next level (a superset of what the exact windows would mark, which is all the filter needs).
struct PTask { uint32_t seg0, k, p, pad; };          // first segment of the query, sub-patterns, pivot level
constexpr uint32_t kPivotRun = 64 * kPivotGroups;     // pivot elements per wave

template <typename pos_t>
__global__ void __launch_bounds__(256) filter_pivot_kernel(const pos_t* __restrict__ P, const RSeg* __restrict__ segs,
                                                           const PTask* __restrict__ tasks, const uint64_t* __restrict__ task_run0,
                                                           uint32_t ntasks, uint64_t* __restrict__ abits)
{
    constexpr uint32_t G = kPivotGroups;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t run = uniform(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (run >= task_run0[ntasks]) return;
    const uint32_t t = wave_task_find(task_run0, ntasks, run);
    const PTask tk = tasks[t];
    const RSeg pv = segs[tk.seg0 + tk.p];
    const uint64_t len = pv.pend - pv.pbegin;
    const uint64_t off0 = (run - task_run0[t]) * kPivotRun;
    const uint64_t off1 = off0 + kPivotRun < len ? off0 + kPivotRun : len;
    uint64_t x[G];
    bool on0[G];
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        const uint64_t i = off0 + 64 * g + lane;
        on0[g] = i < off1;
        x[g] = on0[g] ? (uint64_t)P[pv.pbegin + i] : 0;
        if (pv.abit != ~0ull && off0 + 64 * g < off1) {                       // every element of the pivot list stays
            const unsigned long long m = __ballot(on0[g]);
            if (lane == 0) abits[(pv.abit + off0 + 64 * g) >> 6] = m;
        }
    }
    // the marks of one level ascend over the groups: one MarkRun per level collects them
    auto follow = [&](const RSeg& sg, const uint64_t (&a)[G], const uint64_t (&b)[G], bool (&on)[G], uint64_t (&lo_pos)[G], uint64_t (&hi_pos)[G],
                      bool more_levels) {
        uint32_t i0[G], i1[G];
        pivot_ranges(P, sg.pbegin, sg.pend, a, b, on, i0, i1);
        MarkRun mr;
        mr.bm = abits + (sg.abit >> 6);
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) mr.add(i0[g], i1[g] - 1, on[g]);
        mr.flush();
        if (more_levels) {
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) if (on[g]) { lo_pos[g] = P[sg.pbegin + i0[g]]; hi_pos[g] = P[sg.pbegin + i1[g] - 1]; }
        }
    };
    // towards the first sub-pattern
    {
        uint64_t lo_pos[G], hi_pos[G];
        bool on[G];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) { lo_pos[g] = hi_pos[g] = x[g]; on[g] = on0[g]; }
        for (int l = (int)tk.p - 1; l >= 0; --l) {
            const RSeg sg = segs[tk.seg0 + l], up = segs[tk.seg0 + l + 1];      // gap bounds between l and l+1 belong to l+1
            uint64_t a[G], b[G];
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                a[g] = b[g] = 0;
                if (on[g]) {
                    if (hi_pos[g] < up.lo) on[g] = false;
                    else { a[g] = lo_pos[g] > up.hi ? lo_pos[g] - up.hi : 0; b[g] = hi_pos[g] - up.lo; }
                }
            }
            follow(sg, a, b, on, lo_pos, hi_pos, l > 0);
        }
    }
    // towards the last sub-pattern (which keeps no join state itself)
    {
        uint64_t lo_pos[G], hi_pos[G];
        bool on[G];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) { lo_pos[g] = hi_pos[g] = x[g]; on[g] = on0[g]; }
        for (uint32_t l = tk.p + 1; l + 1 < tk.k; ++l) {
            const RSeg sg = segs[tk.seg0 + l];
            uint64_t a[G], b[G];
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                a[g] = sat_add(lo_pos[g], sg.lo);
                b[g] = sat_add(hi_pos[g], sg.hi);
                if (b[g] == ~0ull) b[g] = ~0ull - 1;                             // (b + 1 is searched)
            }
            follow(sg, a, b, on, lo_pos, hi_pos, l + 2 < tk.k);
        }
    }
}

// survivors per run (for the compaction offsets): the activity bits of a filtered list start on a run boundary, so run r of
// the group owns the words [32 r, 32 r + 32).  Half a wave per run.
__global__ void __launch_bounds__(256) filter_count_runs_kernel(const uint64_t* __restrict__ abits, uint64_t total_runs,
                                                                uint32_t* __restrict__ runcnt)
{
    static_assert(kRun == 2048, "one run = 32 activity words");
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    uint32_t c = 0;
    if (r < total_runs) c = (uint32_t)__popcll(abits[r * 32 + (threadIdx.x & 31)]);
    for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o);                  // both halves of the wave reduce on their own
    if (r < total_runs && (threadIdx.x & 31) == 0) runcnt[r] = c;
}

// survivors per list (for the host's plan): a wave per list
__global__ void __launch_bounds__(256) filter_count_lists_kernel(const uint64_t* __restrict__ crun0, uint32_t ncseg,
                                                                 const uint32_t* __restrict__ runcnt, unsigned long long* __restrict__ segcnt)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t c = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= ncseg) return;
    unsigned long long sum = 0;
    for (uint64_t r = crun0[c] + lane; r < crun0[c + 1]; r += 64) sum += runcnt[r];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) segcnt[c] = sum;
}

// survivor counts of the runs of a chunk's segments, gathered in task order for the scan
__global__ void filter_gather_counts_kernel(const uint32_t* __restrict__ task_cidx, const uint64_t* __restrict__ task_run0, uint32_t ntasks,
                                            const uint64_t* __restrict__ crun0, const uint32_t* __restrict__ runcnt, uint32_t* __restrict__ out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= task_run0[ntasks]) return;
    const uint32_t t = task_find(task_run0, ntasks, r);
    out[r] = runcnt[crun0[task_cidx[t]] + (r - task_run0[t])];
}

// survivors of the chunk's segments -> Pc, in task order; run_cnt = the gathered counts, run_off = their exclusive scan.
// A wave looks at kCompactRuns runs and works on the ones that have survivors.
constexpr uint32_t kCompactRuns = 16;
template <typename pos_t>
__global__ void __launch_bounds__(256) filter_compact_kernel(const pos_t* __restrict__ P, const RSeg* __restrict__ segs,
                                                             const uint32_t* __restrict__ task_seg, const uint64_t* __restrict__ task_run0,
                                                             uint32_t ntasks, const uint64_t* __restrict__ abits,
                                                             const uint32_t* __restrict__ run_cnt, const uint32_t* __restrict__ run_off,
                                                             pos_t* __restrict__ Pc)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r0 = uniform(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kCompactRuns;
    const uint64_t total = task_run0[ntasks];
    if (r0 >= total) return;
    unsigned long long todo = __ballot(lane < kCompactRuns && r0 + lane < total && run_cnt[r0 + lane] != 0);
    while (todo) {
        const uint64_t run = r0 + (uint32_t)(__ffsll((long long)todo) - 1);
        todo &= todo - 1;
        const uint32_t t = wave_task_find(task_run0, ntasks, run);
        const RSeg sg = segs[task_seg[t]];
        const uint64_t len = sg.pend - sg.pbegin;
        const uint64_t off0 = (run - task_run0[t]) * kRun;
        const uint64_t off1 = off0 + kRun < len ? off0 + kRun : len;
        // the 32 activity words of the run in one load; every lane then knows where each word's survivors go
        const uint64_t w0 = (sg.abit + off0) >> 6;
        const uint32_t nw = (uint32_t)((off1 - off0 + 63) >> 6);
        const uint64_t mine = lane < nw ? abits[w0 + lane] : 0;
        uint32_t before = (uint32_t)__popcll(mine);                              // inclusive scan over the words
        for (int o = 1; o < 32; o <<= 1) { const uint32_t v = __shfl_up(before, o); if ((int)lane >= o) before += v; }
        before -= (uint32_t)__popcll(mine);
        const uint32_t out0 = run_off[run];
        unsigned long long todo_w = __ballot(mine != 0);
        while (todo_w) {
            const int wi = __ffsll((long long)todo_w) - 1;
            todo_w &= todo_w - 1;
            const uint64_t bits = __shfl(mine, wi);
            const uint32_t out = out0 + __shfl(before, wi);
            if ((bits >> lane) & 1) Pc[out + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull))] = P[sg.pbegin + off0 + 64ull * wi + lane];
        }
    }
}

struct FilterGroup {                 // outcome of the window filter for the queries [g0,g1)
    uint64_t g0 = 0, g1 = 0;
    uint64_t sub0 = 0;               // first sub-pattern of the group
    std::vector<uint8_t> want;       // per query of the group: run the filter on it (set by the planner)
    std::vector<uint64_t> eff;       // per sub-pattern of the group: list length the join sees (0 for a dead query)
    std::vector<uint32_t> cidx;      // per sub-pattern: place in the compaction order, kNone = list used as it is
    std::vector<uint64_t> crun0;     // [ncseg+1] first run of every compacted segment
    std::vector<uint32_t> cseg;      // [ncseg] segment (index into d_segs) of every compacted segment
    uint32_t ncseg = 0;
    RSeg* d_segs = nullptr;
    uint32_t* d_cseg = nullptr;
    uint64_t* d_crun0 = nullptr;
    uint64_t* d_abits = nullptr;
    uint32_t* d_runcnt = nullptr;
    uint64_t pc_cap = 0;             // compacted elements one join chunk may hold
    bool any = false;
};

inline uint32_t filter_block_shift(uint64_t n) { const unsigned b = bit_width64(n); return b > 31 ? b - 23 : 8; }   // <= 2^23 blocks

// How a query is filtered: 0 = not at all, 1 = streaming sweeps over block bitmaps, 2 = from its shortest list outwards.
// pivot receives the level of the shortest list.
inline int filter_mode(const vlg_queries* q, const Plan& pl, const vlg_workspace* ws, uint64_t qi, uint32_t* pivot = nullptr)
{
    const uint64_t s0 = q->qsub[qi], k = q->qsub[qi + 1] - s0;
    if (!ws->filter || k < 2 || !pl.occ[s0]) return 0;
    uint64_t slots = 0, all = 0, best = ~0ull;
    uint32_t p = 0;
    for (uint64_t i = 0; i < k; ++i) {
        if (i + 1 < k) slots += pl.occ[s0 + i];
        all += pl.occ[s0 + i];
        if (pl.occ[s0 + i] < best) { best = pl.occ[s0 + i]; p = (uint32_t)i; }
    }
    if (slots < ws->filter_min || !slots) return 0;
    if (pivot) *pivot = p;
    // two binary searches per pivot element and level against a pass (or two) over every element of every list
    return ws->filter_pivot && best * ws->filter_pivot_ratio <= all ? 2 : 1;
}

// Bytes of filter state a query needs (0 = the query is not filtered).
inline uint64_t filter_bytes(const vlg_queries* q, const Plan& pl, const vlg_workspace* ws, uint64_t qi, uint64_t nbw)
{
    const int mode = filter_mode(q, pl, ws, qi);
    if (!mode) return 0;
    const uint64_t s0 = q->qsub[qi], k = q->qsub[qi + 1] - s0;
    uint64_t bytes = mode == 1 ? 2 * nbw * 8 : 0;
    for (uint64_t i = 0; i + 1 < k; ++i) bytes += ((pl.occ[s0 + i] + kRun - 1) / kRun) * (kRun / 8 + 4);
    return bytes + k * (sizeof(RSeg) + 32) + 64;
}

template <typename pos_t>
vlg_status filter_group(const vlg_index* idx, const vlg_queries* q, vlg_workspace* ws, const Plan& pl, const std::vector<uint32_t>& poff,
                        const pos_t* P, Arena& A /* advanced past the state the join chunks still need */, FilterGroup& fg)
{
    hipStream_t st = ws->stream;
    const uint32_t g = filter_block_shift(idx->hdr.n);
    const uint64_t nblocks = (idx->hdr.n >> g) + 1, nbw = (nblocks + 63) / 64;
    const uint64_t nsub = q->qsub[fg.g1] - q->qsub[fg.g0];
    fg.sub0 = q->qsub[fg.g0];
    fg.eff.resize(nsub);
    fg.cidx.assign(nsub, kNone);
    for (uint64_t s = 0; s < nsub; ++s) fg.eff[s] = pl.occ[fg.sub0 + s];
    // ---- segments of the filtered queries ------------------------------------------------------------
    std::vector<RSeg> segs;
    std::vector<uint32_t> cseg;                       // segments that keep activity bits, in (query, level) order
    std::vector<uint64_t> crun0(1, 0);
    std::vector<uint32_t> seg_sub;                    // sub-pattern (group relative) of every segment
    uint32_t nfq = 0, kmaxf = 0;
    uint64_t abit = 0;
    std::vector<PTask> ptasks;                        // queries filtered from a pivot list
    std::vector<uint64_t> prun0(1, 0);
    for (uint64_t qi = fg.g0; qi < fg.g1; ++qi) {
        if (!fg.want[qi - fg.g0]) continue;
        const uint64_t s0 = q->qsub[qi];
        const uint32_t k = (uint32_t)(q->qsub[qi + 1] - s0);
        uint32_t pivot = 0;
        const bool by_pivot = filter_mode(q, pl, ws, qi, &pivot) == 2;
        if (by_pivot) {
            ptasks.push_back(PTask{(uint32_t)segs.size(), k, pivot, 0});
            prun0.push_back(prun0.back() + (pl.occ[s0 + pivot] + kPivotRun - 1) / kPivotRun);
        } else {
            kmaxf = std::max(kmaxf, k);
        }
        for (uint32_t i = 0; i < k; ++i) {
            RSeg r;
            memset(&r, 0, sizeof r);
            r.pbegin = poff[pl.did[s0 + i]];
            r.pend = r.pbegin + (uint32_t)pl.occ[s0 + i];
            r.lo = q->lo[s0 + i]; r.hi = q->hi[s0 + i];
            if (i + 1 < k) { r.nlo = q->lo[s0 + i + 1]; r.nhi = q->hi[s0 + i + 1]; }
            r.fq = by_pivot ? kNone : nfq; r.level = i; r.dist = k - 1 - i;
            r.abit = ~0ull;
            if (i + 1 < k) {
                r.abit = abit;                                   // on a run boundary
                abit += (pl.occ[s0 + i] + kRun - 1) / kRun * kRun;
                fg.cidx[s0 + i - fg.sub0] = (uint32_t)cseg.size();
                cseg.push_back((uint32_t)segs.size());
                crun0.push_back(crun0.back() + (pl.occ[s0 + i] + kRun - 1) / kRun);
            }
            seg_sub.push_back((uint32_t)(s0 + i - fg.sub0));
            segs.push_back(r);
        }
        if (!by_pivot) ++nfq;
    }
    fg.any = !segs.empty();
    if (!fg.any) return VLG_OK;
    fg.ncseg = (uint32_t)cseg.size();
    fg.crun0 = crun0;
    fg.cseg = cseg;
    const uint64_t total_runs = crun0.back();
    // ---- device state: what the chunks need first, the bitmaps and task lists (dead after the passes) last ----------
    fg.d_segs = A.take<RSeg>(segs.size());
    fg.d_cseg = A.take<uint32_t>(cseg.size());
    fg.d_crun0 = A.take<uint64_t>(crun0.size());
    fg.d_abits = A.take<uint64_t>(abit / 64 + 1);
    fg.d_runcnt = A.take<uint32_t>(total_runs + 1);
    const uint64_t keep = A.used;
    unsigned long long* d_segcnt = A.take<unsigned long long>(cseg.size());
    uint64_t* d_bm = A.take<uint64_t>((uint64_t)nfq * 2 * nbw + 1);
    uint32_t* d_task_seg = A.take<uint32_t>(segs.size());
    uint64_t* d_task_run0 = A.take<uint64_t>(segs.size() + 1);
    PTask* d_ptasks = A.take<PTask>(ptasks.size() + 1);
    uint64_t* d_prun0 = A.take<uint64_t>(prun0.size());
    if (!d_prun0 || !d_bm) return fail(VLG_E_INTERNAL, "arena carve failed (filter)");
    VLG_HIP_TRY(hipMemsetAsync(fg.d_abits, 0, (abit / 64 + 1) * 8, st));
    VLG_HIP_TRY(hipMemcpyAsync(fg.d_segs, segs.data(), segs.size() * sizeof(RSeg), hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(fg.d_cseg, cseg.data(), cseg.size() * 4, hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(fg.d_crun0, crun0.data(), crun0.size() * 8, hipMemcpyHostToDevice, st));
    if (nfq) VLG_HIP_TRY(hipMemsetAsync(d_bm, 0, (uint64_t)nfq * 2 * nbw * 8, st));
    if (!ptasks.empty()) {
        VLG_HIP_TRY(hipMemcpyAsync(d_ptasks, ptasks.data(), ptasks.size() * sizeof(PTask), hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(d_prun0, prun0.data(), prun0.size() * 8, hipMemcpyHostToDevice, st));
        Timed t(ws, KS_FILTER_PIVOT, 0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_pivot_kernel<pos_t>), dim3((uint32_t)((prun0.back() + 3) / 4)), dim3(256), 0, st, P, fg.d_segs,
                           d_ptasks, d_prun0, (uint32_t)ptasks.size(), fg.d_abits);
        VLG_HIP_TRY(hipGetLastError());
    }
    auto clear_buf = [&](uint32_t buf) -> vlg_status {
        if (nfq) VLG_HIP_TRY(hipMemset2DAsync(d_bm + (uint64_t)buf * nbw, 2 * nbw * 8, 0, nbw * 8, nfq, st));
        return VLG_OK;
    };
    std::vector<uint32_t> task_seg;
    std::vector<uint64_t> task_run0;
    auto run_pass = [&](const RPass& ps, auto&& pick) -> vlg_status {
        task_seg.clear(); task_run0.assign(1, 0);
        uint64_t elems = 0;
        for (uint32_t i = 0; i < segs.size(); ++i)
            if (pick(segs[i])) {
                task_seg.push_back(i);
                const uint64_t len = segs[i].pend - segs[i].pbegin;
                task_run0.push_back(task_run0.back() + (len + kRun - 1) / kRun);
                elems += len;
            }
        if (task_seg.empty()) return VLG_OK;
        VLG_HIP_TRY(hipMemcpyAsync(d_task_seg, task_seg.data(), task_seg.size() * 4, hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(d_task_run0, task_run0.data(), task_run0.size() * 8, hipMemcpyHostToDevice, st));
        {
            Timed t(ws, KS_FILTER_PASS, elems * sizeof(pos_t));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_pass_kernel<pos_t>), dim3((uint32_t)((task_run0.back() + 3) / 4)), dim3(256), 0, st, P,
                               fg.d_segs, d_task_seg, d_task_run0, (uint32_t)task_seg.size(), d_bm, nbw, g, nblocks, fg.d_abits, ps);
        }
        VLG_HIP_TRY(hipGetLastError());
        VLG_HIP_TRY(hipStreamSynchronize(st));      // the task vectors are rebuilt for the next pass
        return VLG_OK;
    };
    // ---- backward sweep: pass j handles the sub-patterns that have j sub-patterns after them -------------------------
    for (uint32_t j = 0; j < kmaxf; ++j) {
        if (j >= 2) if (vlg_status s = clear_buf((j + 1) & 1)) return s;          // it held the marks pass j-1 looked up
        const RPass ps{j ? (int32_t)(j & 1) : -1, (int32_t)((j + 1) & 1), -1, 0, 1};
        if (vlg_status s = run_pass(ps, [&](const RSeg& r) { return r.fq != kNone && r.dist == j; })) return s;
    }
    // ---- forward sweep: pass l handles the sub-patterns at level l (the last one of a query has no join state) --------
    if (kmaxf >= 3) {
        if (nfq) VLG_HIP_TRY(hipMemsetAsync(d_bm, 0, (uint64_t)nfq * 2 * nbw * 8, st));
        for (uint32_t l = 0; l + 1 < kmaxf; ++l) {
            if (l >= 2 && l + 3 <= kmaxf) if (vlg_status s = clear_buf((l + 1) & 1)) return s;   // it held the marks pass l-1 looked up
            const RPass ps{l ? (int32_t)(l & 1) : -1, (int32_t)((l + 1) & 1), +1, 1, 1};
            if (vlg_status s = run_pass(ps, [&](const RSeg& r) { return r.fq != kNone && r.level == l && r.dist >= 1 && (l >= 1 || r.dist >= 2); })) return s;
        }
    }
    // ---- survivors ------------------------------------------------------------------------------------
    {
        Timed t(ws, KS_FILTER_COMPACT, abit / 8);
        hipLaunchKernelGGL(filter_count_runs_kernel, dim3((uint32_t)((total_runs + 7) / 8)), dim3(256), 0, st, fg.d_abits, total_runs, fg.d_runcnt);
        hipLaunchKernelGGL(filter_count_lists_kernel, dim3((uint32_t)((cseg.size() + 3) / 4)), dim3(256), 0, st, fg.d_crun0, fg.ncseg, fg.d_runcnt,
                           d_segcnt);
    }
    VLG_HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> segcnt(cseg.size());
    VLG_HIP_TRY(hipMemcpyAsync(segcnt.data(), d_segcnt, cseg.size() * 8, hipMemcpyDeviceToHost, st));
    VLG_HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t c = 0; c < cseg.size(); ++c) fg.eff[seg_sub[cseg[c]]] = segcnt[c];
    // a query that lost a whole list has no match; one whose survivors do not fit a chunk is joined on its full lists
    for (uint64_t qi = fg.g0; qi < fg.g1; ++qi) {
        const uint64_t s0 = q->qsub[qi] - fg.sub0, k = q->qsub[qi + 1] - q->qsub[qi];
        if (!k || fg.cidx[s0] == kNone) continue;
        bool dead = false;
        uint64_t sum = 0;
        for (uint64_t i = 0; i + 1 < k; ++i) { dead |= fg.eff[s0 + i] == 0; sum += fg.eff[s0 + i]; }
        if (dead) for (uint64_t i = 0; i < k; ++i) fg.eff[s0 + i] = 0;
        else if (sum > fg.pc_cap) for (uint64_t i = 0; i < k; ++i) { fg.eff[s0 + i] = pl.occ[fg.sub0 + s0 + i]; fg.cidx[s0 + i] = kNone; }
    }
    A.used = keep;                                   // bitmaps, counters and task lists are dead
    return VLG_OK;
}

// ---- join of the queries [q0,q1) against the physical lists -------------------------------------------
template <typename pos_t>
vlg_status run_join_chunk(const vlg_index* idx, const vlg_queries* q, vlg_workspace* ws, vlg_result* res, uint64_t q0, uint64_t q1,
                          const Plan& pl, const std::vector<uint32_t>& poff, const pos_t* P, Arena A /* by value: scratch past P */,
                          unsigned long long* d_stats, const FilterGroup* fg, pos_t* Pc /* survivors of filtered lists go here */)
{
    (void)idx;
    hipStream_t st = ws->stream;
    PhaseTrace jt(st);
    // list lengths as the join sees them: the survivors of the window filter where it ran
    auto eo = [&](uint64_t s) -> uint64_t { return fg ? fg->eff[s - fg->sub0] : pl.occ[s]; };
    auto filtered = [&](uint64_t s) -> bool { return fg && fg->cidx[s - fg->sub0] != kNone; };
    const uint64_t s0 = q->qsub[q0], s1 = q->qsub[q1];
    const uint32_t nseg = (uint32_t)(s1 - s0), nq = (uint32_t)(q1 - q0);
    ResultPiece piece;
    piece.q0 = q0; piece.q1 = q1;
    // ---- host-side metadata of the chunk: segments in class-major order (dist descending) -------------
    std::vector<QueryMeta> qm(nq);
    uint32_t kmax = 0;
    for (uint64_t qi = q0; qi < q1; ++qi) {
        uint32_t k = (uint32_t)(q->qsub[qi + 1] - q->qsub[qi]);
        QueryMeta& Q = qm[qi - q0];
        Q.k = k; Q.end_len = q->end_len[qi]; Q.out_first = Q.out_tuple = 0; Q.seg0 = kNone;
        bool live = k > 0 && eo(q->qsub[qi]) > 0;
        if (live) kmax = std::max(kmax, k);
    }
    std::vector<uint32_t> cls_count(kmax + 1, 0), cls_first(kmax + 2, 0);   // segments per dist class
    for (uint64_t qi = q0; qi < q1; ++qi) {
        uint32_t k = qm[qi - q0].k;
        if (!(k > 0 && eo(q->qsub[qi]) > 0)) continue;
        for (uint32_t i = 0; i < k; ++i) cls_count[k - 1 - i]++;
    }
    // classes are stored from the highest dist down to 0
    uint32_t nlive = 0;
    for (int d = (int)kmax - 1; d >= 0; --d) { cls_first[d] = nlive; nlive += cls_count[d]; }
    std::vector<SegMeta> sm(nlive + 1);
    std::vector<uint32_t> seg_begin(nlive + 2, 0), fill(kmax + 1, 0);
    std::vector<uint32_t> seg_of_sub(nseg, kNone);
    for (uint64_t qi = q0; qi < q1; ++qi) {
        uint32_t k = qm[qi - q0].k;
        if (!(k > 0 && eo(q->qsub[qi]) > 0)) continue;
        for (uint32_t i = 0; i < k; ++i) {
            uint32_t d = k - 1 - i;
            seg_of_sub[q->qsub[qi] + i - s0] = cls_first[d] + fill[d]++;
        }
        qm[qi - q0].seg0 = seg_of_sub[q->qsub[qi] - s0];
    }
    uint64_t pc_used = 0;
    std::vector<uint32_t> pc_tasks;                              // filtered sub-patterns of the chunk (group relative), in Pc order
    for (uint64_t qi = q0; qi < q1; ++qi) {
        uint32_t k = qm[qi - q0].k;
        if (qm[qi - q0].seg0 == kNone) continue;
        for (uint32_t i = 0; i < k; ++i) {
            uint64_t s = q->qsub[qi] + i;
            SegMeta& m = sm[seg_of_sub[s - s0]];
            m.level = i; m.dist = k - 1 - i;
            m.lo = q->lo[s]; m.hi = q->hi[s];
            if (filtered(s)) {                                   // private list of the query: the survivors, compacted behind P
                m.pbegin = (uint32_t)((Pc - P) + pc_used);
                pc_tasks.push_back((uint32_t)(s - fg->sub0));
                pc_used += eo(s);
            } else {
                m.pbegin = poff[pl.did[s]];
            }
            m.pend = m.pbegin + (uint32_t)eo(s);
            m.next = (i + 1 < k) ? seg_of_sub[s + 1 - s0] : kNone;
            m.query = (uint32_t)(qi - q0);
        }
    }
    // slots: class-major, segments of a class in query order
    uint64_t acc = 0;
    std::vector<uint64_t> cls_slot_begin(kmax + 1, 0), cls_slot_end(kmax + 1, 0);
    for (int d = (int)kmax - 1; d >= 0; --d) {
        acc = align_up(acc, 64);                                             // a wave step of a class owns whole words of the feasibility bitmap
        cls_slot_begin[d] = acc;
        for (uint32_t j = 0; j < cls_count[d]; ++j) {
            SegMeta& m = sm[cls_first[d] + j];
            seg_begin[cls_first[d] + j] = (uint32_t)acc;
            m.begin = (uint32_t)acc;
            if (m.dist > 0 || m.level == 0) acc += (m.pend - m.pbegin);      // the last list of a k>=2 query needs no join state
            m.end = (uint32_t)acc;
        }
        cls_slot_end[d] = acc;
    }
    const uint64_t T = acc;
    seg_begin[nlive] = (uint32_t)acc;
    seg_begin[nlive + 1] = 0xFFFFFFFFu;
    sm[nlive] = SegMeta{(uint32_t)acc, (uint32_t)acc, 0, 0, 0, 0, kNone, 0, 0, 0};
    if (T == 0 || nlive == 0) { res->pieces.push_back(piece); return VLG_OK; }
    // level-0 slots span the classes k-1 of every live query: [lvl0_begin, lvl0_end) bounds them
    uint64_t lvl0_begin = T, lvl0_end = 0;
    for (uint32_t i = 0; i < nq; ++i)
        if (qm[i].seg0 != kNone) {
            const SegMeta& m0 = sm[qm[i].seg0];
            if (m0.end > m0.begin) { lvl0_begin = std::min<uint64_t>(lvl0_begin, m0.begin); lvl0_end = std::max<uint64_t>(lvl0_end, m0.end); }
        }
    if (lvl0_end <= lvl0_begin) { res->pieces.push_back(piece); return VLG_OK; }
    jt.mark("  chunk: host metadata");
    // ---- private lists: compact the survivors of the chunk's filtered lists behind P ------------------------
    if (!pc_tasks.empty()) {
        std::vector<uint32_t> t_seg(pc_tasks.size()), t_cidx(pc_tasks.size());
        std::vector<uint64_t> t_run0(pc_tasks.size() + 1, 0);
        for (size_t i = 0; i < pc_tasks.size(); ++i) {
            const uint32_t c = fg->cidx[pc_tasks[i]];
            t_cidx[i] = c;
            t_seg[i] = fg->cseg[c];
            t_run0[i + 1] = t_run0[i] + (fg->crun0[c + 1] - fg->crun0[c]);
        }
        const uint64_t runs = t_run0.back();
        uint32_t* d_tseg = A.take<uint32_t>(t_seg.size());
        uint32_t* d_tcidx = A.take<uint32_t>(t_cidx.size());
        uint64_t* d_trun0 = A.take<uint64_t>(t_run0.size());
        uint32_t* d_cnt = A.take<uint32_t>(runs + 1);
        uint32_t* d_off = A.take<uint32_t>(runs + 1);
        size_t scan_tmp = 0;
        VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp, d_cnt, d_off, 0u, runs, rocprim::plus<uint32_t>(), st));
        void* d_scan = A.take<uint8_t>(scan_tmp + 256);
        if (!d_scan || !d_off) return fail(VLG_E_INTERNAL, "arena carve failed (compaction)");
        VLG_HIP_TRY(hipMemcpyAsync(d_tseg, t_seg.data(), t_seg.size() * 4, hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(d_tcidx, t_cidx.data(), t_cidx.size() * 4, hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(d_trun0, t_run0.data(), t_run0.size() * 8, hipMemcpyHostToDevice, st));
        {
            Timed t(ws, KS_FILTER_COMPACT, 2 * pc_used * sizeof(pos_t));
            hipLaunchKernelGGL(filter_gather_counts_kernel, dim3((uint32_t)((runs + 255) / 256)), dim3(256), 0, st, d_tcidx, d_trun0,
                               (uint32_t)t_seg.size(), fg->d_crun0, fg->d_runcnt, d_cnt);
            VLG_HIP_TRY(rocprim::exclusive_scan(d_scan, scan_tmp, d_cnt, d_off, 0u, runs, rocprim::plus<uint32_t>(), st));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_compact_kernel<pos_t>), dim3((uint32_t)(((runs + kCompactRuns - 1) / kCompactRuns + 3) / 4)),
                               dim3(256), 0, st, P, fg->d_segs, d_tseg, d_trun0, (uint32_t)t_seg.size(), fg->d_abits, d_cnt, d_off, Pc);
        }
        VLG_HIP_TRY(hipGetLastError());
        VLG_HIP_TRY(hipStreamSynchronize(st));        // host task vectors go out of scope
    }
    jt.mark("  chunk: compaction");
    // ---- carve the arena ---------------------------------------------------------------------------
    uint32_t* link = A.take<uint32_t>(T);
    pos_t* endp = A.take<pos_t>(T);
    // feasibility bitset + summaries
    FeasBits fb;
    uint64_t* lvl_ptr[kBitLevels];
    {
        uint64_t words = (T + 63) / 64 + 1;
        for (uint32_t l = 0; l < kBitLevels; ++l) {
            lvl_ptr[l] = A.take<uint64_t>(words);
            fb.lvl[l] = lvl_ptr[l];
            fb.words[l] = words;
            if (lvl_ptr[l]) VLG_HIP_TRY(hipMemsetAsync(lvl_ptr[l], 0, words * 8, st));
            words = (words + 63) / 64 + 1;
        }
    }
    FeasBits* d_fb = A.take<FeasBits>(1);
    if (d_fb) VLG_HIP_TRY(hipMemcpyAsync(d_fb, &fb, sizeof fb, hipMemcpyHostToDevice, st));
    const FeasRef fref{lvl_ptr[0], d_fb};
    // arrays that exist for the slots of list 0 only (indexed by absolute slot through an offset pointer)
    const uint64_t t0 = lvl0_begin / kTile * kTile;
    const uint64_t n0 = lvl0_end - t0;
    uint32_t* jump_a = A.take<uint32_t>(n0);
    uint32_t* mlist_a = A.take<uint32_t>(n0);
    uint2* xh_a = A.take<uint2>(n0);
    uint32_t* jump = jump_a ? jump_a - t0 : nullptr;
    uint32_t* mlist = mlist_a ? mlist_a - t0 : nullptr;
    uint2* xh = xh_a ? xh_a - t0 : nullptr;
    SegMeta* d_sm = A.take<SegMeta>(nlive + 1);
    QueryMeta* d_qm = A.take<QueryMeta>(nq);
    uint32_t* d_segb = A.take<uint32_t>(nlive + 2);
    unsigned long long* d_counts = A.take<unsigned long long>(nq);
    // chain records: one per tile a level-0 list overlaps (upper bound of the tiles its chain can visit)
    std::vector<uint32_t> rec_begin(nq + 1, 0), rec_query;
    for (uint32_t i = 0; i < nq; ++i) {
        uint32_t cnt = 0;
        if (qm[i].seg0 != kNone) {
            const SegMeta& m0 = sm[qm[i].seg0];
            if (m0.end > m0.begin) cnt = (m0.end - 1) / kTile - m0.begin / kTile + 1;
        }
        rec_begin[i + 1] = rec_begin[i] + cnt;
        rec_query.insert(rec_query.end(), cnt, i);
    }
    const uint32_t n_rec = rec_begin[nq];
    uint32_t* d_qstart = A.take<uint32_t>(nq);
    uint32_t* d_recb = A.take<uint32_t>(nq + 1);
    uint32_t* d_recc = A.take<uint32_t>(nq);
    uint32_t* d_recq = A.take<uint32_t>(n_rec + 1);
    uint2* d_rec = A.take<uint2>(n_rec + 1);
    if (!d_rec || !xh_a || !lvl_ptr[kBitLevels - 1]) return fail(VLG_E_INTERNAL, "arena carve failed (join)");
    VLG_HIP_TRY(hipMemcpyAsync(d_recb, rec_begin.data(), (nq + 1) * 4, hipMemcpyHostToDevice, st));
    if (n_rec) VLG_HIP_TRY(hipMemcpyAsync(d_recq, rec_query.data(), n_rec * 4, hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemsetAsync(d_qstart, 0xFF, nq * 4, st));
    VLG_HIP_TRY(hipMemcpyAsync(d_sm, sm.data(), (nlive + 1) * sizeof(SegMeta), hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(d_segb, seg_begin.data(), (nlive + 2) * 4, hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(d_qm, qm.data(), nq * sizeof(QueryMeta), hipMemcpyHostToDevice, st));
    // after a class's pass wrote its words of level 0, refresh the summary words above them
    auto summarize_class = [&](uint32_t d) -> vlg_status {
        uint64_t b0 = cls_slot_begin[d], b1 = cls_slot_end[d];
        if (b1 <= b0) return VLG_OK;
        Timed t(ws, KS_JOIN_SCAN, (b1 - b0) / 8);
        uint64_t w0 = b0 >> 6, w1 = (b1 + 63) >> 6;                          // word range written at the level below
        for (uint32_t l = 1; l < kBitLevels; ++l) {
            w0 >>= 6; w1 = (w1 + 63) >> 6;
            hipLaunchKernelGGL(bits_summary_kernel, dim3(grid_for(w1 - w0, 4096)), dim3(256), 0, st, fb.lvl[l - 1], fb.words[l - 1], w0, w1, lvl_ptr[l]);
        }
        VLG_HIP_TRY(hipGetLastError());
        return VLG_OK;
    };
    if (cls_slot_end[0] > cls_slot_begin[0]) {
        Timed t(ws, KS_JOIN_INIT, 0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(join_init_kernel<pos_t>), dim3(runs_grid(cls_slot_end[0] - cls_slot_begin[0])), dim3(256), 0, st, P,
                           d_segb, nlive, d_sm, cls_slot_begin[0], cls_slot_end[0], lvl_ptr[0], endp);
    }
    if (vlg_status s = summarize_class(0)) return s;
    for (uint32_t dist = 1; dist < kmax; ++dist) {
        uint64_t b0 = cls_slot_begin[dist], b1 = cls_slot_end[dist];
        if (b1 > b0) {
            Timed t(ws, KS_JOIN_LINK, 8ull * (b1 - b0));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(join_link_kernel<pos_t>), dim3(runs_grid(b1 - b0)), dim3(256), 0, st, P, d_segb, nlive, d_sm, b0, b1,
                               dist, fref, lvl_ptr[0], endp, link);
        }
        if (vlg_status s = summarize_class(dist)) return s;
    }
    {
        Timed t(ws, KS_JOIN_CHAIN, 8ull * (lvl0_end - lvl0_begin));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(join_jump_kernel<pos_t>), dim3(runs_grid(lvl0_end - lvl0_begin)), dim3(256), 0, st, P, d_segb, nlive, d_sm,
                           d_qm, lvl0_begin, lvl0_end, fref, endp, jump, d_qstart);
        if (t0 < lvl0_begin) VLG_HIP_TRY(hipMemsetAsync(jump + t0, 0xFF, (lvl0_begin - t0) * 4, st));   // slots of the first tile before the range
        hipLaunchKernelGGL(chain_tiles_kernel, dim3((uint32_t)((lvl0_end - t0 + kTile - 1) / kTile)), dim3(256), 0, st, jump, t0, lvl0_end, xh);
        hipLaunchKernelGGL(chain_walk_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, d_sm, d_qm, nq, d_qstart, xh, d_recb, d_rec, d_recc,
                           d_counts);
        if (n_rec)
            hipLaunchKernelGGL(chain_emit_kernel, dim3(grid_for(n_rec)), dim3(256), 0, st, d_recb, d_recc, d_rec, n_rec, d_recq, jump, mlist);
    }
    VLG_HIP_TRY(hipGetLastError());
    // ---- sizes of the result, then gather -------------------------------------------------------------
    std::vector<unsigned long long> counts(nq);
    VLG_HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, nq * 8, hipMemcpyDeviceToHost, st));
    VLG_HIP_TRY(hipStreamSynchronize(st));
    jt.mark("  chunk: link + chain");
    uint64_t M = 0, TV = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        qm[i].out_first = M; qm[i].out_tuple = TV;
        M += counts[i]; TV += counts[i] * qm[i].k;
        res->counts[q0 + i] = counts[i];
    }
    piece.matches = M; piece.tuple_vals = TV;
    if (M) {
        VLG_HIP_TRY(result_alloc(&piece.d_first, M * 8, &piece.first_bytes));
        VLG_HIP_TRY(result_alloc(&piece.d_tuples, TV * 8, &piece.tuple_bytes));
        res->pieces.push_back(piece);
        jt.mark("  chunk: result malloc");
        VLG_HIP_TRY(hipMemcpyAsync(d_qm, qm.data(), nq * sizeof(QueryMeta), hipMemcpyHostToDevice, st));
        Timed t(ws, KS_GATHER, 8ull * (M + TV));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(join_gather_kernel<pos_t>), dim3(runs_grid(lvl0_end - lvl0_begin)), dim3(256), 0, st, P, d_segb, nlive,
                           d_sm, d_qm, lvl0_begin, lvl0_end, link, mlist, d_counts, piece.d_first, piece.d_tuples, d_stats + 2);
        VLG_HIP_TRY(hipGetLastError());
    } else {
        res->pieces.push_back(piece);
    }
    VLG_HIP_TRY(hipStreamSynchronize(st));     // qm / counts host buffers are read by the async copies above
    jt.mark("  chunk: gather");
    res->sum.n_matches += M;
    res->sum.n_tuple_values += TV;
    res->sum.n_chunks++;
    return VLG_OK;
}

// A super-chunk = a run of queries whose distinct occurrence lists fit the physical budget; inside it
// the queries are joined in chunks bounded by the logical budget.
template <typename pos_t>
vlg_status run_batch(const vlg_index* idx, const vlg_queries* q, vlg_workspace* ws, vlg_result* res, const Plan& pl,
                     unsigned long long* d_stats)
{
    const uint64_t fixed = 8ull << 20;      // alignment slack + per-chunk metadata
    if (ws->cap_bytes <= 2 * fixed) return fail(VLG_E_WORKSPACE, "workspace cap too small");
    const uint64_t budget = ws->cap_bytes - fixed;
    // physical lists take at most half of the budget (two buffers during the sort)
    const uint64_t phys_per = sizeof(pos_t) + kPhysScratchPerElem<pos_t>();
    const uint64_t phys_cap = std::min<uint64_t>(budget / 2 / (phys_per + 1), 0xFFFFFF00ull);
    std::vector<uint32_t> stamp(pl.dl.size(), 0xFFFFFFFFu);
    std::vector<uint32_t> poff(pl.dl.size(), 0);
    uint64_t Q0 = 0;
    uint32_t epoch = 0;
    PhaseTrace tr(ws->stream);
    while (Q0 < q->nq) {
        // ---- choose the super-chunk ----------------------------------------------------------------
        std::vector<uint32_t> dlist;
        uint64_t phys = 0, Q1 = Q0;
        while (Q1 < q->nq) {
            uint64_t add = 0;
            size_t mark = dlist.size();
            for (uint64_t s = q->qsub[Q1]; s < q->qsub[Q1 + 1]; ++s)
                if (pl.occ[s] && stamp[pl.did[s]] != epoch) { stamp[pl.did[s]] = epoch; dlist.push_back(pl.did[s]); add += pl.occ[s]; }
            if (phys + add > phys_cap) {
                for (size_t i = mark; i < dlist.size(); ++i) stamp[dlist[i]] = 0xFFFFFFFFu;
                dlist.resize(mark);
                if (Q1 == Q0)
                    return fail(VLG_E_WORKSPACE, "query " + std::to_string(Q1) + " needs " + std::to_string(add) +
                                                     " occurrence slots; workspace cap allows " + std::to_string(phys_cap));
                break;
            }
            phys += add;
            ++Q1;
        }
        ++epoch;
        // ---- arena: physical lists first, filter state and join scratch behind them ---------------------
        const bool uniform_k = q->kmin == q->kmax;
        // cost of a query in bytes of join scratch / in join slots, for list lengths given by occ_of(sub-pattern)
        auto bytes_of = [&](uint64_t qi, auto&& occ_of) -> uint64_t {
            uint32_t k = (uint32_t)(q->qsub[qi + 1] - q->qsub[qi]);
            uint64_t t = 0;
            for (uint32_t i = 0; i < k; ++i) if (i + 1 < k || k == 1) t += occ_of(q->qsub[qi] + i);
            // list-0 arrays cover the slot range of all lists 0, which is exactly those slots when every query has the same k
            uint64_t t0s = k ? (uniform_k ? occ_of(q->qsub[qi]) : t) : 0;
            return t * kJoinBytesPerSlot + t0s * kJoinBytesPerSlot0;
        };
        auto slots_of = [&](uint64_t qi, auto&& occ_of) -> uint64_t {        // slot indices are 32-bit inside a chunk
            uint32_t k = (uint32_t)(q->qsub[qi + 1] - q->qsub[qi]);
            uint64_t t = 64ull * k;                                           // class alignment slack
            for (uint32_t i = 0; i < k; ++i) if (i + 1 < k || k == 1) t += occ_of(q->qsub[qi] + i);
            return t;
        };
        auto full = [&](uint64_t s) -> uint64_t { return pl.occ[s]; };
        uint64_t logical_total = 0, logical_max_query = 0;
        for (uint64_t qi = Q0; qi < Q1; ++qi) {
            uint64_t t = bytes_of(qi, full);
            logical_total += t;
            logical_max_query = std::max(logical_max_query, t);
        }
        size_t sort_tmp = 0;
        if (phys) {
            pos_t* np = nullptr; uint32_t* nu = nullptr;
            VLG_HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, sort_tmp, np, np, (unsigned)phys, (unsigned)dlist.size(), nu, nu, 0,
                                                           bit_width64(idx->hdr.n), ws->stream));
            if (ws->sweep && phys >= ws->sweep_min)
                sort_tmp = std::max(sort_tmp, sweep_temp_bytes(phys, idx->hdr.sigma, ws->stream));
            if (phys >= ws->global_sort_min) {
                size_t tb = 0;
                rocprim::double_buffer<uint64_t> nk(nullptr, nullptr);
                VLG_HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, nk, phys, 0, 64, ws->stream));
                sort_tmp = std::max(sort_tmp, tb);
            }
        }
        const bool will_sweep = ws->sweep && phys >= ws->sweep_min;
        // the trail table (8 B per text position) and the records (8 B per occurrence) must leave most of the workspace to the rest
        uint64_t trail_bytes = will_sweep && ws->trail && ws->dedup ? (idx->hdr.n + phys) * 8 + 512 : 0;
        if (trail_bytes > budget / 4) trail_bytes = 0;
        const bool share_trails = trail_bytes != 0;
        const uint64_t phys_bytes = phys * phys_per + sort_tmp + (dlist.size() + 2) * 24 + (8ull << 20) + trail_bytes;
        const uint64_t join_budget = budget > phys_bytes ? budget - phys_bytes : 0;
        // window filter: state of the filtered queries of a group (at most a third of the budget), dropped query by query if it
        // would not leave room for the largest unfiltered join
        const uint64_t nbw = (((idx->hdr.n >> filter_block_shift(idx->hdr.n)) + 1) + 63) / 64;
        const uint64_t group_cap = join_budget / 3;
        std::vector<uint64_t> fbytes(Q1 - Q0, 0);
        uint64_t filter_total = 0;
        if (ws->filter && logical_max_query + group_cap <= join_budget)
            for (uint64_t qi = Q0; qi < Q1; ++qi) {
                uint64_t b = filter_bytes(q, pl, ws, qi, nbw);
                if (b > group_cap) b = 0;
                fbytes[qi - Q0] = b;
                filter_total += b;
            }
        uint64_t filter_runs = 0;                                             // runs the compaction of one chunk may have to index
        for (uint64_t qi = Q0; qi < Q1; ++qi)
            if (fbytes[qi - Q0]) for (uint64_t s = q->qsub[qi]; s + 1 < q->qsub[qi + 1]; ++s) filter_runs += pl.occ[s] / kRun + 1;
        const uint64_t filter_need = std::min(filter_total, group_cap) + filter_runs * 8;
        const uint64_t cap_bytes = join_budget - filter_need;
        const uint64_t max_chunk_slots = 0xF0000000ull;
        if (logical_max_query > cap_bytes)
            return fail(VLG_E_WORKSPACE, "a query needs " + std::to_string(logical_max_query) + " bytes of join scratch; workspace cap allows " +
                                             std::to_string(cap_bytes));
        const uint64_t want_bytes = std::min<uint64_t>(logical_total, cap_bytes);
        uint64_t meta = (q->qsub[Q1] - q->qsub[Q0] + 4) * (sizeof(SegMeta) + 48) + (Q1 - Q0 + 4) * (sizeof(QueryMeta) + 96) +
                        (want_bytes / 8192 + (Q1 - Q0) + 8) * 48 + (1ull << 20);
        if (vlg_status s = ws_reserve(ws, phys_bytes + filter_need + want_bytes + meta + fixed)) return s;
        Arena A{ws->arena, ws->arena_bytes};
        pos_t* P = nullptr;
        uint64_t Tphys = 0;
        tr.mark("plan super-chunk");
        pos_t* Pc = nullptr;
        uint64_t pc_cap = 0;
        if (vlg_status s = build_physical<pos_t>(idx, ws, res, dlist, pl, A, P, poff, Tphys, sort_tmp, d_stats, Pc, pc_cap, share_trails)) return s;
        tr.mark("locate + sort");
        // ---- groups of queries that share one run of the filter; join chunks inside a group -------------------
        uint64_t g0 = Q0;
        while (g0 < Q1) {
            Arena GA = A;
            FilterGroup fg;
            fg.g0 = g0; fg.pc_cap = pc_cap;
            uint64_t fb = 0, g1 = g0;
            while (g1 < Q1) {
                const uint64_t b = pc_cap ? fbytes[g1 - Q0] : 0;
                if (fb + b > group_cap && g1 > g0) break;
                fb += b;
                fg.want.push_back(b > 0);
                ++g1;
            }
            fg.g1 = g1;
            const FilterGroup* fgp = nullptr;
            if (fb) {
                if (vlg_status s = filter_group<pos_t>(idx, q, ws, pl, poff, P, GA, fg)) return s;
                if (fg.any) fgp = &fg;
                tr.mark("filter group");
            }
            auto eff = [&](uint64_t s) -> uint64_t { return fgp ? fgp->eff[s - fgp->sub0] : pl.occ[s]; };
            auto pc_of = [&](uint64_t qi) -> uint64_t {                      // survivors the query puts into Pc
                uint64_t t = 0;
                if (fgp) for (uint64_t s = q->qsub[qi]; s < q->qsub[qi + 1]; ++s) if (fgp->cidx[s - fgp->sub0] != kNone) t += fgp->eff[s - fgp->sub0];
                return t;
            };
            uint64_t q0 = g0;
            while (q0 < g1) {
                uint64_t T = 0, S = 0, C = 0, q1 = q0;
                while (q1 < g1) {
                    uint64_t t = bytes_of(q1, eff), sl = slots_of(q1, eff), pc = pc_of(q1);
                    if (sl > max_chunk_slots) return fail(VLG_E_WORKSPACE, "a query has more than 2^32 join slots");
                    if (((T + t > want_bytes || S + sl > max_chunk_slots || C + pc > pc_cap) && q1 > q0) || (q1 - q0) >= (1u << 22)) break;
                    T += t; S += sl; C += pc;
                    ++q1;
                }
                vlg_status s = run_join_chunk<pos_t>(idx, q, ws, res, q0, q1, pl, poff, P, GA, d_stats, fgp, Pc);
                if (s) return s;
                tr.mark("join chunk");
                q0 = q1;
            }
            g0 = g1;
        }
        Q0 = Q1;
    }
    return VLG_OK;
}

}  // namespace

extern "C" vlg_status vlg_search_batch(const vlg_index* idx, const vlg_queries* q, vlg_workspace* ws, vlg_result** out)
{
    if (!idx || !q || !ws || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    hipStream_t st = ws->stream;
    vlg_result* res = new vlg_result();
    memset(&res->sum, 0, sizeof res->sum);
    res->sum.n_queries = q->nq;
    res->counts.assign(q->nq, 0);
    res->k.resize(q->nq);
    for (uint64_t i = 0; i < q->nq; ++i) res->k[i] = (uint32_t)(q->qsub[i + 1] - q->qsub[i]);
    uint64_t* d_l = nullptr;
    uint64_t* d_r = nullptr;
    unsigned long long* d_stats = nullptr;
    PhaseTrace tr(st);
    static std::chrono::steady_clock::time_point last_end;
    if (tr.on && last_end.time_since_epoch().count())
        fprintf(stderr, "[vlg trace] %-28s %9.3f ms\n", "(between two batches)",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - last_end).count());
    auto run = [&]() -> vlg_status {
        const uint64_t nsub = q->nsub;
        VLG_HIP_TRY(hipMalloc((void**)&d_l, (nsub + 1) * 8));
        VLG_HIP_TRY(hipMalloc((void**)&d_r, (nsub + 1) * 8));
        VLG_HIP_TRY(hipMalloc((void**)&d_stats, 4 * 8));
        VLG_HIP_TRY(hipMemsetAsync(d_stats, 0, 4 * 8, st));
        // ---- K2: every sub-pattern's SA interval ------------------------------------------------------
        {
            Timed t(ws, KS_BSEARCH, 0);
            if (vlg_status s = launch_backward_search(idx->view, q->d_blob, q->d_suboff, nsub, d_l, d_r, d_stats + 3, st)) return s;
        }
        tr.mark("backward search");
        std::vector<uint64_t> l(nsub), r(nsub);
        if (nsub) {
            VLG_HIP_TRY(hipMemcpyAsync(l.data(), d_l, nsub * 8, hipMemcpyDeviceToHost, st));
            VLG_HIP_TRY(hipMemcpyAsync(r.data(), d_r, nsub * 8, hipMemcpyDeviceToHost, st));
        }
        VLG_HIP_TRY(hipStreamSynchronize(st));
        // a query with an empty occurrence list has no match: none of its lists is materialised
        // (vlg_index.hpp:315-316 returns at the first empty range)
        Plan pl;
        pl.occ.assign(nsub, 0);
        pl.did.assign(nsub, 0);
        for (uint64_t qi = 0; qi < q->nq; ++qi) {
            bool live = q->qsub[qi + 1] > q->qsub[qi];
            for (uint64_t s = q->qsub[qi]; s < q->qsub[qi + 1] && live; ++s) live = (r[s] + 1 - l[s]) > 0;
            if (live) for (uint64_t s = q->qsub[qi]; s < q->qsub[qi + 1]; ++s) pl.occ[s] = r[s] + 1 - l[s];
        }
        // identical SA intervals are the same occurrence list: locate + sort each distinct one once per
        // super-chunk and let every query that uses it share the sorted list.
        {
            std::vector<uint64_t> order;
            order.reserve(nsub);
            for (uint64_t s = 0; s < nsub; ++s) if (pl.occ[s]) { order.push_back(s); res->sum.logical_occurrences += pl.occ[s]; }
            if (ws->dedup) {
                // open-addressing table keyed by the interval; ids in order of first appearance
                uint64_t cap = 16;
                while (cap < 2 * order.size()) cap <<= 1;
                std::vector<uint32_t> table(cap, 0xFFFFFFFFu);
                for (uint64_t s : order) {
                    uint64_t h = (l[s] * 0x9E3779B97F4A7C15ull) ^ (r[s] * 0xC2B2AE3D27D4EB4Full);
                    h ^= h >> 29;
                    uint64_t at = h & (cap - 1);
                    for (;; at = (at + 1) & (cap - 1)) {
                        const uint32_t d = table[at];
                        if (d == 0xFFFFFFFFu) {
                            table[at] = (uint32_t)pl.dl.size();
                            pl.did[s] = (uint32_t)pl.dl.size();
                            pl.dl.push_back(l[s]); pl.docc.push_back(pl.occ[s]);
                            break;
                        }
                        if (pl.dl[d] == l[s] && pl.docc[d] == pl.occ[s]) { pl.did[s] = d; break; }      // same l and same length = same interval
                    }
                }
            } else {
                for (uint64_t s : order) { pl.did[s] = (uint32_t)pl.dl.size(); pl.dl.push_back(l[s]); pl.docc.push_back(pl.occ[s]); }
            }
        }
        tr.mark("intervals to host + plan");
        const uint64_t pos_bytes = idx->hdr.sample_bytes;
        vlg_status s = (pos_bytes == 4) ? run_batch<uint32_t>(idx, q, ws, res, pl, d_stats) : run_batch<uint64_t>(idx, q, ws, res, pl, d_stats);
        if (s) return s;
        unsigned long long hs[4];
        VLG_HIP_TRY(hipMemcpyAsync(hs, d_stats, sizeof hs, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        res->sum.lf_steps = hs[0];
        res->sum.wt_levels_locate = hs[1];
        res->sum.checksum = hs[2];
        res->sum.wt_levels_bsearch = hs[3];
        // algorithmic bytes: 32 B per super-block read (+ one sample per occurrence)
        ws->stats[KS_LOCATE].algorithmic_bytes += 32ull * hs[1] + pos_bytes * res->sum.located_occurrences;
        ws->stats[KS_BSEARCH].algorithmic_bytes += 32ull * hs[3];
        return VLG_OK;
    };
    vlg_status stt = run();
    tr.mark("statistics");
    if (d_l) (void)hipFree(d_l);
    if (d_r) (void)hipFree(d_r);
    if (d_stats) (void)hipFree(d_stats);
    tr.mark("free");
    last_end = std::chrono::steady_clock::now();
    if (stt) { vlg_result_destroy(res); return stt; }
    *out = res;
    return VLG_OK;
}
