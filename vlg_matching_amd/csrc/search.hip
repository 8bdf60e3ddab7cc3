// Workspace, results and the fused search (backward search -> locate -> sort -> window filter -> join).
//   queries.cpp      query batches (both dialects parsed on the host, uploaded)
//   join_device.hpp  the device side of the join: search helpers, feasibility bitset, link / jump / chain / gather kernels
//   filter.hpp       the window filter before the join (kernels + driver of one filter group)
//   kernels.hip      backward search, locate (random access and sorted sweep with shared LF trails)
//
// Join (K5).  The reference's merge join (benchmark/gapped-matching/include/index_sasearch.hpp:85-116;
// semantics of vlg_iterator, include/sdsl/vlg_index.hpp:227-291) advances k monotone pointers one step
// at a time.  Every step only ever raises one pointer to the least value a gap constraint forces, so a
// match is the component-wise LEAST tuple (p_0 >= a, p_1, ..., p_{k-1}) that satisfies all constraints
// lo_i <= L_i[p_i] - L_{i-1}[p_{i-1}] <= hi_i, and the next search restarts at the first p_0 with
// L_0[p_0] >= L_{k-1}[p_{k-1}] + end_len.  That fixed-point view is data parallel:
//   back to front, every element of list i learns whether a feasible chain to the last list starts at
//   it, which element of list i+1 it links to (the first feasible one inside its window) and where the
//   chain ends ("link pass", one merge search per element);  a hierarchical bitset answers "nearest feasible
//   element at or after j";  the chain a, jump[a], jump[jump[a]], ... of a query is resolved by pointer doubling
//   inside tiles, a per-query walk over tiles and a per-tile emit;  a gather pass writes the tuples.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <string.h>
#include <string>
#include <vector>
#include "common.hpp"
#include "kernels.hpp"
#include <rocprim/rocprim.hpp>

using namespace vlg;

#include "search_types.hpp"

// =============================================================================================
// Workspace
// =============================================================================================
enum { KS_BSEARCH = 0, KS_EXPAND, KS_LOCATE, KS_LOCATE_PART, KS_LOCATE_RESOLVE, KS_SORT, KS_FILTER_LADDER, KS_FILTER_PIVOT, KS_FILTER_PASS, KS_FILTER_COMPACT,
       KS_JOIN_INIT, KS_JOIN_LINK, KS_JOIN_SCAN, KS_JOIN_CHAIN, KS_GATHER, KS_EXCHANGE, KS_COUNT };
static const char* kKernelNames[KS_COUNT] = {"backward_search", "expand", "locate", "locate_partition", "locate_resolve", "sort", "filter_ladder", "filter_pivot",
                                             "filter_pass", "filter_compact", "join_init", "join_link", "join_scan", "join_chain", "gather", "exchange"};

struct vlg_workspace {
    hipStream_t stream = nullptr;
    uint64_t cap_bytes = 0;
    uint8_t* arena = nullptr;
    uint64_t arena_bytes = 0;
    HostPool host;              // pinned staging memory of the batch in flight
    bool profile = false;
    bool dedup = true;
    bool sweep = true;          // sorted-sweep locate (n <= 2^32) instead of the random-access persistent kernel
    bool tuples = true;         // materialise every sub-pattern position of every match (sdsl::locate); off: first positions only, which
                                // is all the benchmark's gapped_search_result holds (index_sasearch.hpp:58-118)
    bool trail = true;          // sorted-sweep locate: a walk that steps onto an SA index which is itself an element of the batch stops there
                                // and shares that element's LF steps (kernels.hip: sweep_element; needs dedup -- with dedup off every
                                // occurrence walks its own LF steps like the reference)
    bool filter = true;         // window filter: drop the list elements that can be in no match before the join
    uint64_t filter_min = 1ull << 12;   // queries with fewer join slots are joined as they are (C3, ms per batch: 2^18 271, 2^16 247, 2^14 240,
                                        // 2^12 237.7, 2^10 237.6, 2^7 239)
    uint64_t filter_stream_min = 1ull << 16;   // ... and so are those below this that would be filtered by streaming sweeps
    bool filter_pivot = true;   // filter from the shortest list of a query outwards when it is much shorter than the rest
    uint64_t filter_group_bytes = 0;    // cap of the filter state of one group of queries (0: a third of the join budget)
    uint64_t filter_pivot_ratio = 3;    // ... i.e. when all lists together are at least this many times longer (C3, ms per batch, with the
                                        // ladder: 8 -> 173.2, 6 -> 171.1, 4 -> 169.0, 3 -> 168.9, 2 -> 168.8; with bracket + bisection it was
                                        // 24 -> 262, 12 -> 251, 6 -> 246.6, 4 -> 246.4, <= 3 -> 248): the descents win wherever a list is the shortest
    void* fences = nullptr;     // F[g] = P[64 g + 63] over the lists of the super-chunk in work (join_device.hpp), or null
    void* rungs = nullptr;      // the 4-ary ladder over the same lists (join_device.hpp), or null
    uint64_t* rung_off = nullptr;   // device: first entry of every level
    uint32_t compact_dense_min = 256;        // compaction: runs with fewer survivors move them half a word per lane, fuller ones word by word
                                             // (C3, ms for the class: 0 18.2, 64 18.3, 256 16.6, 512 16.8, 1024 19.2, always half words 31.5)
    uint32_t pivot_rungs = 1;   // the pivot filter searches through the ladder: 0 never (fences + bisection), 1 when it pays, 2 always
    bool want_rungs = false;    // ... and the super-chunk in work has enough pivot searches to pay for building it
    int list_sort = 1;          // 32-bit positions: every list sorted inside itself (list_sort.hpp; 2: without the window passes); 0: the two rocPRIM paths below
    uint64_t global_sort_min = 1ull << 20;  // at least this many occurrences: all lists are sorted by one radix sort of (list, position) keys
    uint64_t sweep_min = 1ull << 22;    // below this many occurrences the persistent random-access kernel is used
    uint64_t sweep_tail = 1ull << 22;   // stragglers of a sweep are finished one lane each (C3, ms per batch: 2^20 214.0, 2^22 213.4, 2^24 213.7, 2^26 213.8)
    // K3u (kernels.hip): a batch that locates at least unsample_pct per cent of all text positions rebuilds the whole suffix array from
    // the samples (n LF steps, no trails, no records) and copies its SA intervals out of it.  0 = never (the default: measured on
    // BASELINE config 3, 60 % of all positions located -- 76.6 ms against 61.0 ms for sweep + partition + resolve: ~110 sparse rounds
    // stream the whole wavelet tree each and scatter 4 bytes per step); needs dedup + trail (the options that let the walks of a
    // batch share LF steps at all) and SA-order samples.
    uint32_t unsample_pct = 0;
    uint64_t unsample_min = 1ull << 28;     // ... and at least this many occurrences: ~110 rounds of two launches each are a fixed cost
    uint64_t unsample_tail = 1ull << 20;    // walkers left when the sorted rounds end and the lanes finish them one by one
    uint64_t sample_reads = 0;              // SA samples read by the locate stage of the batch in work (algorithmic bytes)
    // one process per GPU: a collective search shards the DISTINCT LISTS of a batch over the ranks for locate + sort,
    // exchanges the sorted lists (all-gather) and shards the QUERIES for filter + join
    // device memory of a batch's first steps (SA intervals, the interval plan's arrays, counters): kept from batch to batch -- a
    // hipMalloc / hipFree pair per batch cost more than the kernels between them
    uint8_t* head = nullptr;
    uint64_t head_bytes = 0;
    uint64_t tmp_key[6] = {0, 0, 0, 0, 0, 0};   // what the cached temporary-storage size of the sorts was asked for
    size_t tmp_bytes = 0;
    int x_ranks = 1, x_rank = 0;
    vlg_exchange_fn x_fn = nullptr;     // in-place all-gather of device pieces; null with x_comm set = vlg_comm_allgatherv
    vlg_alltoall_fn x_a2a = nullptr;    // pairwise exchange of packed pieces; null with x_comm set = vlg_comm_alltoallv
    void* x_ctx = nullptr;
    void* x_comm = nullptr;
    bool exchange_all = false;          // every sorted list to every rank (in-place all-gather) instead of needed lists, pairwise
    uint64_t* x_status = nullptr;       // device: one status word per rank (the agreement before every exchange)
    uint32_t x_agreed = 0;              // agreements made by the batch in work
    vlg_kernel_stat stats[KS_COUNT];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[KS_COUNT];
    std::vector<hipEvent_t> free_events;
};

namespace {

// VLG_TRACE=1: wall time of the host phases of a batch on stderr (the stream is drained at every mark, so the figures
// include the kernels launched in the phase)
struct PhaseTrace {
    bool on, sync;          // VLG_TRACE=host: host time between the marks only (no wait for the stream: the batch runs as it does untraced)
    hipStream_t st;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTrace(hipStream_t s) : on(getenv("VLG_TRACE") != nullptr), sync(!on || strcmp(getenv("VLG_TRACE"), "host") != 0), st(s), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what)
    {
        if (!on) return;
        if (sync) (void)hipStreamSynchronize(st);
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
    void begin(int which, uint64_t algorithmic_bytes) override
    {
        int k = which == 0 ? KS_LOCATE : (which == 1 ? KS_LOCATE_PART : KS_LOCATE_RESOLVE);
        ws->stats[k].launches++;
        ws->stats[k].algorithmic_bytes += algorithmic_bytes;
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

vlg_status ws_head(vlg_workspace* ws, uint64_t bytes, uint8_t** out)
{
    if (bytes > ws->head_bytes) {
        if (ws->head) { (void)hipFree(ws->head); ws->head = nullptr; ws->head_bytes = 0; }
        const uint64_t want = align_up(bytes + bytes / 4, 1 << 20);
        VLG_HIP_TRY(hipMalloc((void**)&ws->head, want));
        ws->head_bytes = want;
    }
    *out = ws->head;
    return VLG_OK;
}

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
    if (ws->head) (void)hipFree(ws->head);
    if (ws->x_status) (void)hipFree(ws->x_status);
    ws->host.release();
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
    if (!strcmp(name, "list_sort")) { ws->list_sort = value > 2 ? 1 : (int)value; return VLG_OK; }
    if (!strcmp(name, "trail")) { ws->trail = value != 0; return VLG_OK; }
    if (!strcmp(name, "tuples")) { ws->tuples = value != 0; return VLG_OK; }
    if (!strcmp(name, "filter")) { ws->filter = value != 0; return VLG_OK; }
    if (!strcmp(name, "filter_min")) { ws->filter_min = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "filter_pivot")) { ws->filter_pivot = value != 0; return VLG_OK; }
    if (!strcmp(name, "filter_pivot_ratio")) { ws->filter_pivot_ratio = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "compact_dense_min")) { ws->compact_dense_min = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20)); return VLG_OK; }
    if (!strcmp(name, "pivot_rungs")) { ws->pivot_rungs = value <= 0 ? 0u : (value == 1 ? 1u : 2u); return VLG_OK; }
    if (!strcmp(name, "filter_stream_min")) { ws->filter_stream_min = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "filter_group_bytes")) { ws->filter_group_bytes = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "reserve")) {                           // allocate the scratch now (+ the per-chunk metadata a batch adds on top of its budget)
        const uint64_t b = std::min<uint64_t>((uint64_t)value, ws->cap_bytes);
        return ws_reserve(ws, b + b / 96 + (256ull << 20));
    }
    if (!strcmp(name, "sweep_tail")) { ws->sweep_tail = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "exchange_all")) { ws->exchange_all = value != 0; return VLG_OK; }
    if (!strcmp(name, "unsample_pct")) { ws->unsample_pct = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20)); return VLG_OK; }
    if (!strcmp(name, "unsample_tail")) { ws->unsample_tail = (uint64_t)value; return VLG_OK; }
    if (!strcmp(name, "unsample_min")) { ws->unsample_min = (uint64_t)value; return VLG_OK; }
    return fail(VLG_E_INVALID, std::string("unknown workspace option ") + name);
}

namespace {
vlg_status ws_exchange_reset(vlg_workspace* ws, int n_ranks)
{
    ws->x_comm = nullptr; ws->x_fn = nullptr; ws->x_a2a = nullptr; ws->x_ctx = nullptr; ws->x_ranks = 1; ws->x_rank = 0;
    if (n_ranks > 1 && !ws->x_status) VLG_HIP_TRY(hipMalloc((void**)&ws->x_status, 8 * 1024));      // (room for 1024 ranks)
    return VLG_OK;
}
}  // namespace

extern "C" vlg_status vlg_workspace_set_comm(vlg_workspace* ws, void* nccl_comm)
{
    if (!ws) return fail(VLG_E_INVALID, "null argument");
    if (vlg_status s = ws_exchange_reset(ws, 1)) return s;
    if (!nccl_comm) return VLG_OK;
    int n = 0, r = 0;
    if (vlg_status s = vlg_comm_info(nccl_comm, &n, &r)) return s;
    if (n > 1024) return fail(VLG_E_UNSUPPORTED, "more than 1024 ranks");
    if (vlg_status s = ws_exchange_reset(ws, n)) return s;
    ws->x_comm = nccl_comm; ws->x_ranks = n; ws->x_rank = r;
    return VLG_OK;
}

extern "C" vlg_status vlg_workspace_set_exchange(vlg_workspace* ws, int n_ranks, int rank, vlg_exchange_fn fn, void* ctx)
{
    if (!ws || n_ranks < 1 || n_ranks > 1024 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !fn)) return fail(VLG_E_INVALID, "bad exchange arguments");
    if (vlg_status s = ws_exchange_reset(ws, n_ranks)) return s;
    ws->x_fn = n_ranks > 1 ? fn : nullptr; ws->x_ctx = ctx; ws->x_ranks = n_ranks; ws->x_rank = rank;
    return VLG_OK;
}

extern "C" vlg_status vlg_workspace_set_exchange_alltoall(vlg_workspace* ws, int n_ranks, int rank, vlg_alltoall_fn fn, void* ctx)
{
    if (!ws || n_ranks < 1 || n_ranks > 1024 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !fn)) return fail(VLG_E_INVALID, "bad exchange arguments");
    if (vlg_status s = ws_exchange_reset(ws, n_ranks)) return s;
    ws->x_a2a = n_ranks > 1 ? fn : nullptr; ws->x_ctx = ctx; ws->x_ranks = n_ranks; ws->x_rank = rank;
    return VLG_OK;
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
    void* d_first = nullptr;       // [matches] positions of `width` bytes
    void* d_tuples = nullptr;      // [tuple_vals]
    uint32_t width = 8;            // 4 when the text's positions fit 32 bits (vlg_result_fetch widens on the way to the host)
    uint64_t first_bytes = 0, tuple_bytes = 0;   // sizes of the allocations (a parked buffer may be larger than needed)
};

struct vlg_result {
    vlg_result_summary sum;
    std::vector<uint64_t> counts;          // host: per query
    std::vector<uint32_t> k;               // host: sub-patterns per query
    std::vector<ResultPiece> pieces;
    std::vector<uint64_t> owned;           // collective search: [begin, end) pairs of the queries this rank joined (empty: all of them)
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

hipError_t result_alloc(void** out, uint64_t bytes, uint64_t* got)
{
    if (void* p = result_cache().take(bytes, got)) { *out = p; return hipSuccess; }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {                                   // memory may be parked in the cache: release it and retry once
        (void)hipGetLastError();
        result_cache().drain();
        e = hipMalloc(out, bytes);
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

extern "C" vlg_status vlg_result_owned_queries(const vlg_result* r, uint64_t* h_ranges, uint32_t cap_ranges, uint32_t* n_ranges)
{
    if (!r || !n_ranges) return fail(VLG_E_INVALID, "null argument");
    *n_ranges = (uint32_t)(r->owned.size() / 2);
    for (uint32_t i = 0; i < *n_ranges && i < cap_ranges && h_ranges; ++i) { h_ranges[2 * i] = r->owned[2 * i]; h_ranges[2 * i + 1] = r->owned[2 * i + 1]; }
    return VLG_OK;
}

// ---- narrow results on their way to the host ----------------------------------------------------------------------------------
// Pieces travel as they are stored: 4 bytes per position when the text's positions fit 32 bits (half the PCIe time of the
// 8-byte values the caller receives).  Up to 16 host threads each take a slice of the piece, copy it block by block into their own
// two pinned staging blocks on their own stream and widen block i into the caller's array while block i + 1 is in flight.
namespace {
#ifndef VLG_FETCH_THREADS
#define VLG_FETCH_THREADS 16
#endif
constexpr uint32_t kFetchThreads = VLG_FETCH_THREADS;
constexpr uint64_t kFetchBlock = 8ull << 20;                 // bytes per staging block
struct FetchLane { void* blk[2] = {nullptr, nullptr}; hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; int device = -1; };
struct FetchStage {
    std::mutex mu;                                           // one narrow fetch at a time uses the staging blocks
    FetchLane lane[kFetchThreads];
    hipError_t ready(FetchLane& l, int dev)
    {
        hipError_t e = hipSuccess;
        for (int b = 0; b < 2 && e == hipSuccess; ++b)
            if (!l.blk[b]) e = hipHostMalloc(&l.blk[b], kFetchBlock, hipHostMallocPortable);
        if (e == hipSuccess && l.device != dev) {            // streams and events belong to a device
            if (l.st) { (void)hipStreamDestroy(l.st); (void)hipEventDestroy(l.ev[0]); (void)hipEventDestroy(l.ev[1]); l.st = nullptr; }
            e = hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking);
            for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventCreateWithFlags(&l.ev[b], hipEventDisableTiming);
            if (e == hipSuccess) l.device = dev;
        }
        return e;
    }
};
FetchStage& fetch_stage() { static FetchStage* s = new FetchStage; return *s; }

inline void widen_block(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) __builtin_nontemporal_store((uint64_t)in[i], out + i);     // the caller's array is written once: keep it out of the caches
}

// elements [a, b) of a narrow device array into h[a, b)
hipError_t fetch_slice_widened(FetchLane& l, const uint32_t* d, uint64_t a, uint64_t b, uint64_t* h)
{
    const uint64_t per = kFetchBlock / 4;
    uint64_t at = a;
    uint64_t pend_at[2] = {0, 0}, pend_n[2] = {0, 0};
    int slot = 0;
    hipError_t e = hipSuccess;
    auto drain = [&](int sl) -> hipError_t {
        if (!pend_n[sl]) return hipSuccess;
        hipError_t w = hipEventSynchronize(l.ev[sl]);
        if (w == hipSuccess) widen_block(static_cast<const uint32_t*>(l.blk[sl]), h + pend_at[sl], pend_n[sl]);
        pend_n[sl] = 0;
        return w;
    };
    while (at < b && e == hipSuccess) {
        const uint64_t n = std::min(per, b - at);
        e = drain(slot);                                         // the block this copy lands in must have been widened
        if (e != hipSuccess) break;
        e = hipMemcpyAsync(l.blk[slot], d + at, n * 4, hipMemcpyDeviceToHost, l.st);
        if (e == hipSuccess) e = hipEventRecord(l.ev[slot], l.st);
        pend_at[slot] = at; pend_n[slot] = e == hipSuccess ? n : 0;
        at += n;
        slot ^= 1;
        if (e == hipSuccess) e = drain(slot);                    // widen the older block while this one is in flight
    }
    for (int sl = 0; sl < 2; ++sl) { hipError_t w = drain(sl); if (e == hipSuccess) e = w; }
    return e;
}

hipError_t fetch_widened(const void* d_narrow, uint64_t count, uint64_t* h)
{
    if (!count) return hipSuccess;
    int cur = 0, dev = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    dev = cur;
    hipPointerAttribute_t attr;                                  // the copies run on streams of the device that holds the result
    if (hipPointerGetAttributes(&attr, d_narrow) == hipSuccess) dev = attr.device; else (void)hipGetLastError();
    struct Restore { int from, to; ~Restore() { if (from != to) (void)hipSetDevice(from); } } restore{cur, dev};
    if (dev != cur && (e = hipSetDevice(dev)) != hipSuccess) return e;
    FetchStage& fs = fetch_stage();
    std::lock_guard<std::mutex> g(fs.mu);
    // (C3, 1.2 GB: 8 threads 31-35 ms, 16 threads 26 ms; PCIe alone would be 21)
    static const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t nt = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(kFetchThreads, hw), count / (1u << 20)));
    for (uint32_t t = 0; t < nt; ++t) if ((e = fs.ready(fs.lane[t], dev)) != hipSuccess) return e;
    const uint32_t* d = static_cast<const uint32_t*>(d_narrow);
    if (nt == 1) return fetch_slice_widened(fs.lane[0], d, 0, count, h);
    std::vector<hipError_t> err(nt, hipSuccess);
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            err[t] = hipSetDevice(dev);                          // the current device is a per-thread setting
            if (err[t] == hipSuccess) err[t] = fetch_slice_widened(fs.lane[t], d, count * t / nt, count * (t + 1) / nt, h);
        });
    for (auto& x : th) x.join();
    for (uint32_t t = 0; t < nt; ++t) if (err[t] != hipSuccess) return err[t];
    return hipSuccess;
}
}  // namespace

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
    if (h_tuples && r->sum.n_matches && !r->sum.n_tuple_values)
        return fail(VLG_E_INVALID, "tuples were not materialised (workspace option \"tuples\" is 0)");
    uint64_t fo = 0, to = 0;
    for (const auto& p : r->pieces) {
        if (p.width == 4) {
            if (h_first && p.matches) VLG_HIP_TRY(fetch_widened(p.d_first, p.matches, h_first + fo));
            if (h_tuples && p.tuple_vals) VLG_HIP_TRY(fetch_widened(p.d_tuples, p.tuple_vals, h_tuples + to));
        } else {
            if (h_first && p.matches) VLG_HIP_TRY(hipMemcpy(h_first + fo, p.d_first, p.matches * 8, hipMemcpyDeviceToHost));
            if (h_tuples && p.tuple_vals) VLG_HIP_TRY(hipMemcpy(h_tuples + to, p.d_tuples, p.tuple_vals * 8, hipMemcpyDeviceToHost));
        }
        fo += p.matches;
        to += p.tuple_vals;
    }
    return VLG_OK;
}

extern "C" vlg_status vlg_result_fetch32(const vlg_result* r, uint32_t* h_first, uint32_t* h_tuples)
{
    if (!r) return fail(VLG_E_INVALID, "null argument");
    if (h_tuples && r->sum.n_matches && !r->sum.n_tuple_values)
        return fail(VLG_E_INVALID, "tuples were not materialised (workspace option \"tuples\" is 0)");
    for (const auto& p : r->pieces)
        if (p.matches && p.width != 4) return fail(VLG_E_INVALID, "the positions of this result are 64 bits wide (vlg_result_fetch)");
    uint64_t fo = 0, to = 0;
    for (const auto& p : r->pieces) {
        if (h_first && p.matches) VLG_HIP_TRY(hipMemcpy(h_first + fo, p.d_first, p.matches * 4, hipMemcpyDeviceToHost));
        if (h_tuples && p.tuple_vals) VLG_HIP_TRY(hipMemcpy(h_tuples + to, p.d_tuples, p.tuple_vals * 4, hipMemcpyDeviceToHost));
        fo += p.matches;
        to += p.tuple_vals;
    }
    return VLG_OK;
}

#include "join_device.hpp"

// device counters of a batch: [0] LF steps, [1] tree levels of locate, [3] tree levels of the backward searches, [4..5] flags of
// vlg_join_batch's input check, [kStatsChecksum ..) partial checksums
constexpr uint32_t kStatsChecksum = 8, kStatsWords = kStatsChecksum + kChecksumSlots;

namespace {

inline uint32_t runs_grid(uint64_t slots, uint32_t run = kRun) { return (uint32_t)((((slots + run - 1) / run) + 3) / 4); }   // 4 waves per workgroup

inline uint32_t grid_for(uint64_t n, uint32_t cap = 16384) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, cap)); }

// Host loops over the queries (or sub-patterns) of a big batch, cut into slices for a few threads: fn(begin, end, thread).
// Batches below 2 x min_per_thread items run on the caller's thread.
constexpr uint32_t kHostThreads = 8;
template <class F>
void parallel_slices(uint64_t a, uint64_t b, uint64_t min_per_thread, F&& fn)
{
    const uint64_t n = b > a ? b - a : 0;
    static const uint32_t hw = std::max(1u, std::min(kHostThreads, std::thread::hardware_concurrency()));
    const uint32_t nt = (uint32_t)std::min<uint64_t>(hw, n / std::max<uint64_t>(1, min_per_thread));
    if (nt <= 1) { fn(a, b, 0u); return; }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (uint32_t t = 1; t < nt; ++t) th.emplace_back([&, t] { fn(a + n * t / nt, a + n * (t + 1) / nt, t); });
    fn(a, a + n / nt, 0u);
    for (auto& x : th) x.join();
}

struct Arena {
    uint8_t* base; uint64_t size; uint64_t used = 0;
    bool failed = false;         // sticky: one carve that did not fit poisons the arena, so a check after a block of carves sees it
    template <class T> T* take(uint64_t count)
    {
        uint64_t bytes = align_up(count * sizeof(T), 256);
        if (failed || bytes > size || used > size - bytes) { failed = true; return nullptr; }
        T* p = reinterpret_cast<T*>(base + used);
        used += bytes;
        return p;
    }
};

struct Plan {                       // host view of the batch after backward search (staged memory: the device plan copies straight into it)
    rvec<uint64_t> occ;             // per sub-pattern, 0 for dead queries
    rvec<uint32_t> did;             // per sub-pattern: distinct-interval id (valid when occ > 0)
    rvec<uint64_t> dl, docc;        // per distinct interval: left border, size
};

template <typename pos_t> constexpr uint64_t kPhysScratchPerElem() { return 20; }   // sweep scratch; the sorted lists reuse it
constexpr uint64_t kJoinBytesPerSlot = 4 + 8 + 1;      // link, endp(<=8), feasibility bits + summaries (any slot)
constexpr uint64_t kJoinBytesPerSlot0 = 4 + 4 + 8 + 1; // jump, mlist, (exit,hops), chain records (slots of list 0)

// ---- sort of all lists at once: one radix sort of (list, position) keys instead of one sort per list -------------
template <typename pos_t>
__global__ void sort_compose_kernel(const pos_t* __restrict__ P, const uint64_t* __restrict__ off /* [nd+1] */, uint64_t nd, uint64_t total,
                                    uint32_t pos_bits, uint64_t* __restrict__ keys)
{
    constexpr uint32_t kPer = 8;                           // elements per thread: one list lookup per 2048 elements
    __shared__ uint64_t s_first;
    for (uint64_t base = (uint64_t)blockIdx.x * 256 * kPer; base < total; base += (uint64_t)gridDim.x * 256 * kPer) {
        if (threadIdx.x == 0) {
            uint64_t lo = 0, hi = nd;                      // last list with off[l] <= base
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (off[mid] <= base) lo = mid; else hi = mid; }
            s_first = lo;
        }
        __syncthreads();
        uint64_t l = s_first;
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint64_t t = base + i * 256 + threadIdx.x;
            if (t < total) {
                while (off[l + 1] <= t) ++l;               // empty lists are skipped too
                keys[t] = (l << pos_bits) | (uint64_t)P[t];
            }
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

// flags[0] |= 1 when some list is not ascending; flags[1] = largest element of all lists
template <typename T>
__global__ void lists_check_kernel(const T* __restrict__ lists, const uint64_t* __restrict__ off, uint64_t n_lists, uint64_t total,
                                   unsigned long long* __restrict__ flags)
{
    constexpr uint32_t kPer = 8;
    __shared__ uint64_t s_first;
    bool bad = false;
    unsigned long long mx = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * 256 * kPer; base < total; base += (uint64_t)gridDim.x * 256 * kPer) {
        if (threadIdx.x == 0) {
            uint64_t lo = 0, hi = n_lists;                 // last list with off[l] <= base
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (off[mid] <= base) lo = mid; else hi = mid; }
            s_first = lo;
        }
        __syncthreads();
        uint64_t l = s_first;
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint64_t t = base + i * 256 + threadIdx.x;
            if (t < total) {
                while (off[l + 1] <= t) ++l;
                const uint64_t v = (uint64_t)lists[t];
                if (t > off[l] && (uint64_t)lists[t - 1] > v) bad = true;
                mx = v > mx ? v : mx;
            }
        }
        __syncthreads();
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1ull);
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long u = __shfl_xor(mx, o); mx = u > mx ? u : mx; }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&flags[1], mx);
}


#include "list_sort.hpp"

inline bool check_sort_enabled() { static const bool on = [] { const char* e = getenv("VLG_CHECK_SORT"); return e && e[0] == '1'; }(); return on; }

// ---- collective search: agreement before every exchange ------------------------------------------------------------------------
// Every rank all-gathers ONE status word before the payload of an exchange step moves: a rank whose share of the batch failed up to
// there says so and all ranks return an error, instead of the healthy ones waiting inside a collective for a peer that has left.
// An Agreement lives for one super-chunk of the batch; leaving its scope without having settled (an early error return anywhere
// between the plan and the exchange) settles with the error on the way out.
struct Agreement {
    vlg_workspace* ws;
    bool done = false;
    vlg_status mine = VLG_E_INTERNAL;          // what a silent exit reports
    explicit Agreement(vlg_workspace* w) : ws(w) {}
    Agreement(const Agreement&) = delete;
    // -> VLG_OK when every rank said OK; this rank's own status when it failed; VLG_E_INTERNAL naming the first failed peer otherwise
    vlg_status settle(vlg_status status)
    {
        if (done || ws->x_ranks <= 1) { done = true; return status; }
        done = true;
        ++ws->x_agreed;
        const int n = ws->x_ranks, me = ws->x_rank;
        hipStream_t st = ws->stream;
        std::vector<uint64_t> ones((size_t)n, 1), words((size_t)n, 0);
        uint64_t word = (uint64_t)status;
        uint64_t* d = ws->x_status;                                     // [0, n): all ranks' words, mine at [me]; [n, 2n): my word n times (all-to-all)
        if (!d) return fail(VLG_E_INTERNAL, "collective search: no status buffer");
        VLG_HIP_TRY(hipMemcpyAsync(d + me, &word, 8, hipMemcpyHostToDevice, st));
        int rc = 0;
        if (ws->x_fn) rc = ws->x_fn(ws->x_ctx, d, ones.data(), 8u, n, me, st);
        else if (ws->x_a2a) {
            std::vector<uint64_t> mine_n((size_t)n, word);
            VLG_HIP_TRY(hipMemcpyAsync(d + n, mine_n.data(), 8 * (size_t)n, hipMemcpyHostToDevice, st));
            VLG_HIP_TRY(hipStreamSynchronize(st));                    // (mine_n is read by the copy)
            rc = ws->x_a2a(ws->x_ctx, d + n, ones.data(), d, ones.data(), 8u, n, me, st);
        } else if (ws->x_comm) rc = (int)vlg_comm_allgatherv(ws->x_comm, d + me, ones.data(), 8u, d, st);
        else return fail(VLG_E_INTERNAL, "collective search without a communicator");
        if (rc) return fail(VLG_E_INTERNAL, "collective search: the status exchange failed with code " + std::to_string(rc));
        VLG_HIP_TRY(hipMemcpyAsync(words.data(), d, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        if (status) return status;                                     // (the message of the failure is already set)
        for (int r = 0; r < n; ++r)
            if (words[r]) return fail(VLG_E_INTERNAL, "collective search: rank " + std::to_string(r) + " failed with status " + std::to_string(words[r]) +
                                                      " before the exchange; nothing was exchanged");
        return VLG_OK;
    }
    ~Agreement() { if (!done) { const std::string keep = vlg_last_error(); (void)settle(mine); set_error(keep); } }
};

// packed[dst_off[k] + j] <-> base[src_off[k] + j] for the segments k < n_seg (dst_off has n_seg + 1 entries); kGather: base -> packed
template <typename pos_t, bool kGather>
__global__ void __launch_bounds__(256) segments_copy_kernel(pos_t* __restrict__ base, pos_t* __restrict__ packed, const uint64_t* __restrict__ src_off,
                                                            const uint64_t* __restrict__ dst_off, uint64_t n_seg, uint64_t total)
{
    constexpr uint32_t kPer = 8;
    __shared__ uint64_t s_first;
    for (uint64_t b0 = (uint64_t)blockIdx.x * 256 * kPer; b0 < total; b0 += (uint64_t)gridDim.x * 256 * kPer) {
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t lo = 0, hi = n_seg;                       // last segment with dst_off[k] <= b0
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (dst_off[mid] <= b0) lo = mid; else hi = mid; }
            s_first = lo;
        }
        __syncthreads();
        uint64_t k = s_first;
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint64_t t = b0 + i * 256 + threadIdx.x;
            if (t < total) {
                while (dst_off[k + 1] <= t) ++k;               // (empty segments are skipped too)
                const uint64_t at = src_off[k] + (t - dst_off[k]);
                if (kGather) packed[t] = base[at]; else base[at] = packed[t];
            }
        }
    }
}

// K3u (kernels.hip): does a share of `acc` occurrences of this index's text pay for rebuilding the whole suffix array?
inline bool unsample_applies(const vlg_index* idx, const vlg_workspace* ws, uint64_t acc)
{
    if (!ws->unsample_pct || !ws->sweep || !ws->trail || !ws->dedup || idx->is_int) return false;
    if (idx->hdr.sampling != kSamplingSaOrder || idx->hdr.dens < 2 || idx->hdr.n > (1ull << 32) + 1 || idx->hdr.sigma >= 0x7FFFu) return false;
    return acc >= ws->sweep_min && acc >= ws->unsample_min && (__uint128_t)acc * 100 >= (__uint128_t)idx->hdr.n * ws->unsample_pct;
}

// ---- physical pass: locate + sort every distinct interval used by queries [Q0,Q1) -------------------
template <typename pos_t>
vlg_status build_physical(const vlg_index* idx, vlg_workspace* ws, vlg_result* res, const std::vector<uint32_t>& dlist_in /* distinct ids */,
                          const Plan& pl, Arena& A, pos_t*& P_out, std::vector<uint32_t>& poff /* per distinct id -> offset (size dl) */,
                          uint64_t& Tphys, size_t sort_tmp, unsigned long long* d_stats, pos_t*& Pc_out, uint64_t& pc_cap,
                          bool share_steps /* the sweep's walks stop at elements of the batch (room for the member bit-vector and the records) */,
                          bool wide /* SA indices need 33 bits / 64-bit samples (always so for 64-bit positions) */,
                          bool allow_unsample /* the caller planned the workspace for K3u (no trail table) */,
                          const vlg_queries* q = nullptr, const std::vector<uint64_t>* xq = nullptr /* collective search: the ranks' query cuts */,
                          Agreement* ag = nullptr /* collective search: settled here, right before the exchange */)
{
    hipStream_t st = ws->stream;
    PhaseTrace bt(st);
    // all lists of the super-chunk ("g": global), laid out one after the other in SA order of their intervals -- first the ones that
    // are pairwise disjoint ("outer"), then the ones nested inside another list of the chunk (the interval of a pattern that continues
    // another pattern; SA intervals of patterns nest or are disjoint, they never overlap in part).  With the outer lists in SA order
    // at the front, the slot of an SA index inside one of them is the number of such indices before it: what the sweep's member
    // bit-vector returns (kernels.hip: sweep_element).
    const uint32_t gnd = (uint32_t)dlist_in.size();
    std::vector<uint32_t> dlist(dlist_in);
    bool ascending = true;
    for (uint32_t i = 1; i < gnd && ascending; ++i) ascending = pl.dl[dlist[i - 1]] <= pl.dl[dlist[i]];
    if (!ascending)                                                        // (a plan made on the host numbers the intervals as they come)
        std::sort(dlist.begin(), dlist.end(), [&](uint32_t a, uint32_t b) { return pl.dl[a] != pl.dl[b] ? pl.dl[a] < pl.dl[b] : pl.docc[a] < pl.docc[b]; });
    uint32_t n_outer = gnd;
    {
        std::vector<uint32_t> inner;
        uint64_t cover_end = 0;
        uint32_t w = 0;
        for (uint32_t i = 0; i < gnd;) {
            uint32_t j = i, longest = i;                                   // lists that start at the same index: the longest holds the others
            for (; j < gnd && pl.dl[dlist[j]] == pl.dl[dlist[i]]; ++j) if (pl.docc[dlist[j]] > pl.docc[dlist[longest]]) longest = j;
            const bool outer = pl.dl[dlist[i]] >= cover_end;
            for (uint32_t t = i; t < j; ++t) {
                if (t == longest && outer) continue;
                inner.push_back(dlist[t]);
            }
            if (outer) { const uint32_t d = dlist[longest]; cover_end = pl.dl[d] + pl.docc[d]; dlist[w++] = d; }
            i = j;
        }
        n_outer = w;
        for (uint32_t d : inner) dlist[w++] = d;
    }
    svec<uint64_t> goff64(gnd + 1);
    uint64_t gacc = 0;
    for (uint32_t i = 0; i < gnd; ++i) {
        goff64[i] = gacc;
        poff[dlist[i]] = (uint32_t)gacc;
        gacc += pl.docc[dlist[i]];
    }
    goff64[gnd] = gacc;
    Tphys = gacc;
    P_out = nullptr;
    Pc_out = nullptr;
    pc_cap = 0;
    ws->fences = nullptr;
    ws->rungs = nullptr;
    if (!gacc) return ag ? ag->settle(VLG_OK) : VLG_OK;              // (every rank sees the same empty plan and settles too)
    // Collective search: this rank locates and sorts a contiguous share [sl, sh) of the lists -- cut so that every rank gets the
    // same number of occurrences, identically on every rank -- and the ranks exchange their sorted pieces afterwards.
    uint32_t sl = 0, sh = gnd;
    std::vector<uint64_t> xcounts;
    std::vector<uint32_t> cut(ws->x_ranks + 1, gnd);
    if (ws->x_ranks > 1) {
        cut[0] = 0;
        for (int r = 1; r < ws->x_ranks; ++r) {
            const uint64_t target = gacc / (uint64_t)ws->x_ranks * (uint64_t)r;
            const uint32_t c = (uint32_t)(std::lower_bound(goff64.begin(), goff64.end(), target) - goff64.begin());
            cut[r] = std::max(cut[r - 1], std::min(c, gnd));
        }
        xcounts.resize(ws->x_ranks);
        for (int r = 0; r < ws->x_ranks; ++r) xcounts[r] = goff64[cut[r + 1]] - goff64[cut[r]];
        sl = cut[ws->x_rank]; sh = cut[ws->x_rank + 1];
    }
    const uint32_t nd = sh - sl;                                       // from here to the sort: this rank's share
    svec<uint64_t> off64(nd + 1), lh(std::max<uint32_t>(nd, 1));
    svec<uint32_t> off32(nd + 1);
    for (uint32_t i = 0; i <= nd; ++i) { off64[i] = goff64[sl + i] - goff64[sl]; off32[i] = (uint32_t)off64[i]; }
    for (uint32_t i = 0; i < nd; ++i) lh[i] = pl.dl[dlist[sl + i]];
    const uint64_t acc = off64[nd];
    const bool use_sweep = ws->sweep && acc >= ws->sweep_min && idx->hdr.n <= (1ull << (wide ? 33 : 32)) && (!idx->is_int || (int_sweep_possible(idx->iview) && sizeof(pos_t) == 4 && !wide));
    // K3u: the whole suffix array from the samples, inside the sweep's scratch (n x 4 B + 20 B per sample <= 20 B per occurrence)
    const uint64_t n_walkers = idx->is_int ? 0 : idx->view.n_samples;
    const bool use_unsample = allow_unsample && use_sweep && sizeof(pos_t) == 4 && unsample_applies(idx, ws, acc) &&
                              align_up(idx->hdr.n * 4, 256) + n_walkers * 20 + 1024 <= gacc * kPhysScratchPerElem<pos_t>();
    pos_t* Pg = A.take<pos_t>(gacc);
    // scratch of the sweep (20 B per element); the sorted lists Pb reuse it once locate is done
    uint8_t* scratch = A.take<uint8_t>(gacc * kPhysScratchPerElem<pos_t>());
    pos_t* Pa = Pg ? Pg + goff64[sl] : nullptr;
    uint64_t* d_goff64 = (ws->x_ranks > 1) ? A.take<uint64_t>(gnd + 1) : nullptr;
    uint64_t* d_off64 = A.take<uint64_t>(nd + 1);
    uint32_t* d_off32 = A.take<uint32_t>(nd + 1);
    uint64_t* d_lh = A.take<uint64_t>(nd);
    unsigned long long* d_counter = A.take<unsigned long long>(4);     // [0] the sweep's counter, [1..3] diagnostics (VLG_RESOLVE_STATS)
    void* d_tmp = A.take<uint8_t>(sort_tmp + 256);
    if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (physical)");
    pos_t* Pb = reinterpret_cast<pos_t*>(scratch);
    VLG_HIP_TRY(hipMemcpyAsync(d_off64, off64.data(), (nd + 1) * 8, hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(d_off32, off32.data(), (nd + 1) * 4, hipMemcpyHostToDevice, st));
    if (nd) VLG_HIP_TRY(hipMemcpyAsync(d_lh, lh.data(), nd * 8, hipMemcpyHostToDevice, st));
    if (d_goff64) VLG_HIP_TRY(hipMemcpyAsync(d_goff64, goff64.data(), (gnd + 1) * 8, hipMemcpyHostToDevice, st));
    uint64_t* rec = nullptr;
    Block* member = nullptr;
    uint8_t* front = nullptr;
    const uint32_t n_member_lists = n_outer > sl ? std::min(n_outer - sl, nd) : 0u;      // this rank's outer lists: a prefix of its share
    if (use_sweep && !use_unsample && share_steps && acc <= 0xFFFFFF00ull) {
        rec = A.take<uint64_t>(acc);
        member = A.take<Block>(member_blocks(idx->hdr.n));
        if (!idx->is_int) front = A.take<uint8_t>(acc);             // the symbol in front of every element that stops on its first step (kernels.hpp: SweepKernels)
        if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (member bit-vector, records)");
    }
    // the sort's tables only depend on the list lengths: they are built and uploaded while the first step of locate runs
    const uint64_t sort_mark = A.used;
    ListSortPlan lsp;
    const unsigned bits = std::max(1u, bit_width64(idx->hdr.n >= 2 ? idx->hdr.n - 2 : 0));   // the largest position is n - 2 (n - 1 is the sentinel)
    const std::function<vlg_status()> plan_sort = [&]() -> vlg_status {
        if (sizeof(pos_t) != 4 || !ws->list_sort) return VLG_OK;
        const vlg_status ls = list_sort_prepare(off64, nd, A, st, lsp);
        if (ls != VLG_OK && ls != VLG_E_WORKSPACE) return ls;      // no room for its tables: the device-wide sort below
        if (ls != VLG_OK) A.used = sort_mark;
        return VLG_OK;
    };
    if (!acc) {
        // (a rank without a share: nothing to locate or sort, the exchange below still takes place)
    } else if (use_unsample) {
        uint32_t* sa_full = reinterpret_cast<uint32_t*>(scratch);
        uint64_t* val_a = reinterpret_cast<uint64_t*>(scratch + align_up(idx->hdr.n * 4, 256));
        uint64_t* val_b = val_a + n_walkers;
        uint16_t* key_a = reinterpret_cast<uint16_t*>(val_b + n_walkers);
        uint16_t* key_b = key_a + n_walkers;
        svec<unsigned long long> h_done(1, 0);
        SweepTimer timer(ws);
        bt.mark("  physical: tables + sort plan");
        vlg_status s = VLG_OK;
        if constexpr (sizeof(pos_t) == 4) {
            if (wide) s = launch_unsample<true>(idx->view, d_lh, d_off64, nd, acc, Pa, sa_full, val_a, val_b, key_a, key_b, d_tmp, sort_tmp, d_counter, h_done.data(),
                                                d_stats, ws->unsample_tail, st, &timer, &plan_sort);
            else s = launch_unsample<false>(idx->view, d_lh, d_off64, nd, acc, Pa, sa_full, val_a, val_b, key_a, key_b, d_tmp, sort_tmp, d_counter, h_done.data(),
                                            d_stats, ws->unsample_tail, st, &timer, &plan_sort);
        }
        if (s) return s;
        ws->sample_reads += n_walkers;
        res->sum.locate_mode = VLG_LOCATE_UNSAMPLE;
        bt.mark("  physical: unsampling");
    } else if (use_sweep) {
        const uint64_t cap = std::min<uint64_t>(acc, wide ? sweep_batch_max<true>() : sweep_batch_max<false>());
        uint64_t* val_a = reinterpret_cast<uint64_t*>(scratch);
        uint64_t* val_b = val_a + cap;
        uint16_t* key_a = reinterpret_cast<uint16_t*>(val_b + cap);
        uint16_t* key_b = key_a + cap;
        SweepTimer timer(ws);
        bt.mark("  physical: tables + sort plan");
        vlg_status s = VLG_OK;
        if (idx->is_int) {
            if constexpr (sizeof(pos_t) == 4)
                s = launch_int_locate_sweep(idx->iview, d_lh, d_off64, nd, acc, Pa, val_a, val_b, key_a, key_b, d_tmp, sort_tmp, d_counter, d_stats, ws->sweep_tail, st,
                                            &timer, member, n_member_lists, rec, &plan_sort);
            else s = fail(VLG_E_INTERNAL, "integer-alphabet index with 64-bit positions");
        } else if (wide)
            s = launch_locate_sweep<pos_t, true>(idx->view, d_lh, d_off64, nd, acc, Pa, val_a, val_b, key_a, key_b, d_tmp, sort_tmp, d_counter, d_stats,
                                                 ws->sweep_tail, st, &timer, member, n_member_lists, rec, &plan_sort, front);
        else if constexpr (sizeof(pos_t) == 4)                      // (a 64-bit position type always comes with wide indices)
            s = launch_locate_sweep<uint32_t, false>(idx->view, d_lh, d_off64, nd, acc, Pa, val_a, val_b, key_a, key_b, d_tmp, sort_tmp, d_counter, d_stats,
                                                     ws->sweep_tail, st, &timer, member, n_member_lists, rec, &plan_sort, front);
        else s = fail(VLG_E_INTERNAL, "64-bit positions with 32-bit SA indices");
        if (s) return s;
        ws->sample_reads += acc;
        res->sum.locate_mode = idx->hdr.dens == 1 && idx->hdr.sampling == kSamplingSaOrder ? VLG_LOCATE_COPY : VLG_LOCATE_SWEEP;
        bt.mark("  physical: sweep");
    } else if (idx->is_int) {
        // integer-alphabet index (int_index.hpp): one lane per occurrence on the wavelet matrix of the BWT
        ws->sample_reads += acc;
        res->sum.locate_mode = VLG_LOCATE_WALKS;
        if (vlg_status s = plan_sort()) return s;
        if constexpr (sizeof(pos_t) == 4) {
            {
                Timed t(ws, KS_EXPAND, 0);
                if (vlg_status s = launch_expand<uint32_t>(d_lh, d_off64, nd, acc, Pa, nullptr, st)) return s;
            }
            Timed t(ws, KS_LOCATE, 0);
            if (vlg_status s = launch_int_locate(idx->iview, Pa, acc, d_stats, st)) return s;
        } else return fail(VLG_E_INTERNAL, "integer-alphabet index with 64-bit positions");
    } else if (wide && sizeof(pos_t) == 4) {
        // few occurrences, 33-bit SA indices, 32-bit positions: the in-place kernel walks in 64-bit words of the scratch, then narrows
        ws->sample_reads += acc;
        res->sum.locate_mode = VLG_LOCATE_WALKS;
        if (vlg_status s = plan_sort()) return s;
        uint64_t* io64 = reinterpret_cast<uint64_t*>(scratch);
        {
            Timed t(ws, KS_EXPAND, 0);
            if (vlg_status s = launch_expand<uint64_t>(d_lh, d_off64, nd, acc, io64, nullptr, st)) return s;
        }
        {
            Timed t(ws, KS_LOCATE, 0);
            if (vlg_status s = launch_locate<uint64_t>(idx->view, io64, acc, d_stats, st)) return s;
            if (vlg_status s = launch_narrow<uint32_t>(io64, reinterpret_cast<uint32_t*>(Pa), acc, st)) return s;
        }
    } else {
        ws->sample_reads += acc;
        res->sum.locate_mode = VLG_LOCATE_WALKS;
        if (vlg_status s = plan_sort()) return s;
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
    uint64_t dead_bytes = 0;                              // free bytes behind the sorted lists (the survivors of the window filter go there)
    bool sorted = false;
    if (!acc) {
        sorted = true;
        P_out = Pa;
    } else if (lsp.ready) {
        // 32-bit positions: sorted inside every list (list_sort.hpp) -- 4 passes of 8 B per element instead of 6 of 16 B
        Timed t(ws, KS_SORT, 2ull * acc * sizeof(pos_t));
        if (vlg_status ls = list_sort_enqueue(lsp, reinterpret_cast<uint32_t*>(Pa), reinterpret_cast<uint32_t*>(scratch), d_off64, bits, st, ws->list_sort == 1)) return ls;
        A.used = sort_mark;                               // its tables are dead once its kernels have run (stream order)
        sorted = true;
        P_out = Pa;
    }
    if (sorted) {
    } else if (global_sort) {
        // one radix sort of (list, position) keys: the passes stream the whole batch whatever the list sizes are
        uint64_t* ka = reinterpret_cast<uint64_t*>(scratch);
        uint64_t* kb = ka + acc;
        Timed t(ws, KS_SORT, 2ull * acc * sizeof(pos_t));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sort_compose_kernel<pos_t>), dim3(grid_for((acc + 7) / 8, 32768)), dim3(256), 0, st, Pa, d_off64, (uint64_t)nd, acc, bits, ka);
        rocprim::double_buffer<uint64_t> keys(ka, kb);
        size_t tb = sort_tmp;
        VLG_HIP_TRY(rocprim::radix_sort_keys(d_tmp, tb, keys, acc, 0, bits + list_bits, st));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sort_narrow_kernel<pos_t>), dim3(grid_for(acc, 32768)), dim3(256), 0, st, keys.current(), acc, bits, Pa);
        VLG_HIP_TRY(hipGetLastError());
        P_out = Pa;
    } else {
        Timed t(ws, KS_SORT, 2ull * acc * sizeof(pos_t));
        size_t tb = sort_tmp;
        VLG_HIP_TRY(rocprim::segmented_radix_sort_keys(d_tmp, tb, Pa, Pb, (unsigned)acc, nd, d_off32, d_off32 + 1, 0, bits, st));
        P_out = Pb;
        if (ws->x_ranks > 1) {                                // the pieces of all ranks meet in ONE array: back to the share's place
            VLG_HIP_TRY(hipMemcpyAsync(Pa, Pb, acc * sizeof(pos_t), hipMemcpyDeviceToDevice, st));
            P_out = Pa;
        }
    }
    // ---- the exchange step: the ranks' sorted shares meet ----------------------------------------------------------------------------------
    const uint64_t* d_check_off = d_off64;
    uint64_t check_nd = nd, check_acc = acc;
    const pos_t* check_base = nullptr;                                  // (null: P_out)
    if (ws->x_ranks > 1) {
        // every rank has come this far, or says so now (Agreement): nothing moves unless all of them are ready
        if (ag) if (vlg_status as = ag->settle(VLG_OK)) return as;
        const int n = ws->x_ranks, me = ws->x_rank;
        const bool pairwise = !ws->exchange_all && !ws->x_fn && (ws->x_a2a || ws->x_comm) && q && xq && (int)xq->size() == n + 1;
        if (!pairwise) {
            // every list to every rank: one in-place all-gather of the shares
            Timed t(ws, KS_EXCHANGE, (gacc - acc) * sizeof(pos_t));
            int rc = 0;
            if (ws->x_fn) rc = ws->x_fn(ws->x_ctx, Pg, xcounts.data(), (uint32_t)sizeof(pos_t), n, me, st);
            else if (ws->x_comm) rc = (int)vlg_comm_allgatherv(ws->x_comm, Pa, xcounts.data(), (uint32_t)sizeof(pos_t), Pg, st);
            else return fail(VLG_E_INTERNAL, "collective search without a communicator (or an all-to-all callback with \"exchange_all\")");
            if (rc) return ws->x_fn ? fail(VLG_E_INTERNAL, "the exchange callback failed with code " + std::to_string(rc)) : (vlg_status)rc;
            d_check_off = d_goff64; check_nd = gnd; check_acc = gacc;
        } else {
            // needed lists only, pairwise.  Every rank computes the same table: need[r] = the lists the queries of rank r use; rank s
            // owes rank r the lists of ITS share among them.  Lists that follow each other in the layout travel as one segment.
            std::vector<uint32_t> place(pl.dl.size(), 0);               // distinct id -> place in the layout
            for (uint32_t i = 0; i < gnd; ++i) place[dlist[i]] = i;
            std::vector<uint8_t> need((size_t)n * gnd, 0);
            for (int r = 0; r < n; ++r) {
                uint8_t* nr = need.data() + (size_t)r * gnd;
                for (uint64_t qi = (*xq)[r]; qi < (*xq)[r + 1]; ++qi)
                    for (uint64_t sidx = q->qsub[qi]; sidx < q->qsub[qi + 1]; ++sidx)
                        if (pl.occ[sidx]) nr[place[pl.did[sidx]]] = 1;
            }
            // segments (first element in Pg, length) of what `from` owes `to`, in layout order
            auto segments = [&](int from, int to, svec<uint64_t>& src, svec<uint64_t>& dst, uint64_t& at) {
                const uint8_t* nt = need.data() + (size_t)to * gnd;
                for (uint32_t i = cut[from]; i < cut[from + 1];) {
                    if (!nt[i] || goff64[i + 1] == goff64[i]) { ++i; continue; }
                    uint32_t j = i;
                    while (j < cut[from + 1] && nt[j]) ++j;
                    src.push_back(goff64[i]);
                    dst.push_back(at);
                    at += goff64[j] - goff64[i];
                    i = j;
                }
            };
            svec<uint64_t> s_src, s_dst, r_src, r_dst;
            std::vector<uint64_t> scount((size_t)n, 0), rcount((size_t)n, 0);
            uint64_t s_at = 0, r_at = 0;
            for (int r = 0; r < n; ++r) {
                if (r == me) continue;                                   // (a rank's own lists are where they belong already)
                uint64_t b = s_at;
                segments(me, r, s_src, s_dst, s_at);
                scount[r] = s_at - b;
                b = r_at;
                segments(r, me, r_src, r_dst, r_at);
                rcount[r] = r_at - b;
            }
            s_dst.push_back(s_at);
            r_dst.push_back(r_at);
            const uint64_t ns = s_src.size(), nr_ = r_src.size();
            pos_t* d_send = A.take<pos_t>(s_at + 1);
            pos_t* d_recv = A.take<pos_t>(r_at + 1);
            uint64_t* d_seg = A.take<uint64_t>(2 * (ns + nr_) + 4);
            if (A.failed) return fail(VLG_E_WORKSPACE, "collective search: no room for the exchange buffers");
            uint64_t* d_s_src = d_seg; uint64_t* d_s_dst = d_s_src + ns; uint64_t* d_r_src = d_s_dst + ns + 1; uint64_t* d_r_dst = d_r_src + nr_;
            if (ns) VLG_HIP_TRY(hipMemcpyAsync(d_s_src, s_src.data(), ns * 8, hipMemcpyHostToDevice, st));
            VLG_HIP_TRY(hipMemcpyAsync(d_s_dst, s_dst.data(), (ns + 1) * 8, hipMemcpyHostToDevice, st));
            if (nr_) VLG_HIP_TRY(hipMemcpyAsync(d_r_src, r_src.data(), nr_ * 8, hipMemcpyHostToDevice, st));
            VLG_HIP_TRY(hipMemcpyAsync(d_r_dst, r_dst.data(), (nr_ + 1) * 8, hipMemcpyHostToDevice, st));
            Timed t(ws, KS_EXCHANGE, r_at * sizeof(pos_t));
            if (s_at) hipLaunchKernelGGL(HIP_KERNEL_NAME(segments_copy_kernel<pos_t, true>), dim3(grid_for((s_at + 7) / 8, 16384)), dim3(256), 0, st, Pg, d_send, d_s_src, d_s_dst, ns, s_at);
            VLG_HIP_TRY(hipGetLastError());
            int rc = 0;
            if (ws->x_a2a) rc = ws->x_a2a(ws->x_ctx, d_send, scount.data(), d_recv, rcount.data(), (uint32_t)sizeof(pos_t), n, me, st);
            else rc = (int)vlg_comm_alltoallv(ws->x_comm, d_send, scount.data(), d_recv, rcount.data(), (uint32_t)sizeof(pos_t), st);
            if (rc) return ws->x_a2a ? fail(VLG_E_INTERNAL, "the exchange callback failed with code " + std::to_string(rc)) : (vlg_status)rc;
            if (r_at) hipLaunchKernelGGL(HIP_KERNEL_NAME(segments_copy_kernel<pos_t, false>), dim3(grid_for((r_at + 7) / 8, 16384)), dim3(256), 0, st, Pg, d_recv, d_r_src, d_r_dst, nr_, r_at);
            VLG_HIP_TRY(hipGetLastError());
            check_base = Pa;                                           // (lists no query of this rank uses hold nothing: only the own share is checked)
        }
        P_out = Pg;
    }
    if (P_out == Pb) dead_bytes = gacc * kPhysScratchPerElem<pos_t>() - gacc * sizeof(pos_t);
    else dead_bytes = (uint64_t)(scratch - reinterpret_cast<uint8_t*>(Pg)) + gacc * kPhysScratchPerElem<pos_t>() - gacc * sizeof(pos_t);
    if (ws->x_ranks == 1 && P_out == Pa) P_out = Pg;          // (the same place: the one share starts at the beginning)
    // VLG_CHECK_SORT=1 (set by the tests): every list is verified ascending on the device after the sort -- list_sort.hpp leans on
    // the key order of rocPRIM's block_radix_rank, which a library upgrade could change silently
    if (check_sort_enabled() && P_out) {
        unsigned long long* d_flags = A.take<unsigned long long>(2);
        if (!d_flags) return fail(VLG_E_INTERNAL, "arena carve failed (sort check)");
        unsigned long long flags[2] = {0, 0};
        VLG_HIP_TRY(hipMemsetAsync(d_flags, 0, 16, st));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(lists_check_kernel<pos_t>), dim3(grid_for((check_acc + 7) / 8, 8192)), dim3(256), 0, st, check_base ? check_base : P_out, d_check_off, check_nd, check_acc, d_flags);
        VLG_HIP_TRY(hipGetLastError());
        VLG_HIP_TRY(hipMemcpyAsync(flags, d_flags, 16, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        if (flags[0]) return fail(VLG_E_INTERNAL, "an occurrence list is not ascending after the sort (VLG_CHECK_SORT)");
    }
    // no wait here: the staging vectors live in the workspace's pinned pool until the batch ends, so the caller plans the
    // window filter while the sort runs
    {
        const uint64_t pc_first = align_up(gacc, 64);
        if (dead_bytes > (pc_first - gacc + 64) * sizeof(pos_t) && pc_first < 0xFFFFFF00ull) {
            Pc_out = P_out + pc_first;
            pc_cap = std::min<uint64_t>(dead_bytes / sizeof(pos_t) - (pc_first - gacc) - 64, 0xFFFFFF00ull - pc_first);
        }
    }
    // fences of the sorted lists (and room for those of the survivors' lists behind them)
    ws->fences = nullptr;
    ws->rungs = nullptr;
    if (P_out) {
        const uint64_t cover = Pc_out ? (uint64_t)(Pc_out - P_out) + pc_cap : gacc;
        const uint64_t entries = cover / 64 + 2;
        if (!A.failed && A.size - A.used > entries * sizeof(pos_t) + 4096) {
            pos_t* F = A.take<pos_t>(entries);
            ws->fences = F;
        }
        // the ladder for the pivot filter (a third of the lists' size): only when the window filter will run on these lists
        bool fences_written = false;
        if (ws->fences && ws->want_rungs && gacc >= 64) {
            const RungLayout rl = rung_layout(gacc);
            if (A.size - A.used > rl.entries * sizeof(pos_t) + 8192) {
                pos_t* R = A.take<pos_t>(rl.entries);
                uint64_t* d_off = A.take<uint64_t>(kMaxRungs + 1);
                svec<uint64_t> h_off(rl.off, rl.off + kMaxRungs + 1);
                VLG_HIP_TRY(hipMemcpyAsync(d_off, h_off.data(), (kMaxRungs + 1) * 8, hipMemcpyHostToDevice, st));
                fences_written = kRungsHoldFences && rl.levels >= 6 / kRungShift;      // the fences are one of its levels: written on the way
                {
                    Timed t(ws, KS_FILTER_LADDER, gacc * sizeof(pos_t));
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(rung_build_kernel<pos_t>), dim3(grid_for(gacc >> kRungShift, 16384)), dim3(256), 0, st, P_out, gacc, R, d_off,
                                       rl.levels, fences_written ? static_cast<pos_t*>(ws->fences) : (pos_t*)nullptr);
                }
                VLG_HIP_TRY(hipGetLastError());
                ws->rungs = R;
                ws->rung_off = d_off;
            }
        }
        if (ws->fences && !fences_written && gacc >= 64) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(fence_build_kernel<pos_t>), dim3(grid_for(gacc / 64, 8192)), dim3(256), 0, st, P_out, (uint64_t)0, gacc / 64,
                               static_cast<pos_t*>(ws->fences));
            VLG_HIP_TRY(hipGetLastError());
        }
    }
    res->sum.located_occurrences += acc;
    return VLG_OK;
}

}  // namespace

#include "filter.hpp"

namespace {

// ---- join of the queries [q0,q1) against the physical lists -------------------------------------------
template <typename pos_t>
vlg_status run_join_chunk(const vlg_queries* q, vlg_workspace* ws, vlg_result* res, uint64_t q0, uint64_t q1,
                          const Plan& pl, const std::vector<uint32_t>& poff /* per sub-pattern: its list inside P */, const pos_t* P,
                          Arena A /* by value: scratch past P */,
                          unsigned long long* d_stats, const FilterGroup* fg, pos_t* Pc /* survivors of filtered lists go here */)
{
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
    // The filtered sub-patterns of the chunk are collected on the same walk (in Pc order): their compaction is launched first so
    // that it runs while the rest of the metadata is built.
    svec<QueryMeta> qm(nq);
    uint32_t kmax = 0;
    svec<uint32_t> t_seg, t_cidx;
    svec<uint64_t> t_run0;
    if (fg) { t_seg.reserve(nseg); t_cidx.reserve(nseg); t_run0.reserve(nseg + 1); }
    t_run0.push_back(0);
    uint64_t pc_total = 0;
    for (uint64_t qi = q0; qi < q1; ++qi) {
        const uint64_t a = q->qsub[qi];
        uint32_t k = (uint32_t)(q->qsub[qi + 1] - a);
        QueryMeta& Q = qm[qi - q0];
        Q.k = k; Q.end_len = q->end_len[qi]; Q.out_first = Q.out_tuple = 0; Q.seg0 = kNone;
        if (!(k > 0 && eo(a) > 0)) continue;
        kmax = std::max(kmax, k);
        if (fg && !fg->speculated)
            for (uint32_t i = 0; i < k; ++i) {
                const uint32_t c = fg->cidx[a + i - fg->sub0];
                if (c == kNone) continue;
                t_cidx.push_back(c);
                t_seg.push_back(fg->cseg[c]);
                t_run0.push_back(t_run0.back() + (fg->crun0[c + 1] - fg->crun0[c]));
                pc_total += fg->eff[a + i - fg->sub0];
            }
    }
    // ---- private lists: compact the survivors of the chunk's filtered lists behind P ------------------------
    if (!t_seg.empty()) {
        const uint64_t runs = t_run0.back();
        uint32_t* d_tseg = A.take<uint32_t>(t_seg.size());
        uint32_t* d_tcidx = A.take<uint32_t>(t_cidx.size());
        uint64_t* d_trun0 = A.take<uint64_t>(t_run0.size());
        uint32_t* d_cnt = A.take<uint32_t>(runs + 1);
        uint32_t* d_off = A.take<uint32_t>(runs + 1);
        size_t scan_tmp = 0;
        VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp, d_cnt, d_off, 0u, runs, rocprim::plus<uint32_t>(), st));
        void* d_scan = A.take<uint8_t>(scan_tmp + 256);
        if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (compaction)");
        VLG_HIP_TRY(hipMemcpyAsync(d_tseg, t_seg.data(), t_seg.size() * 4, hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(d_tcidx, t_cidx.data(), t_cidx.size() * 4, hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(d_trun0, t_run0.data(), t_run0.size() * 8, hipMemcpyHostToDevice, st));
        {
            Timed t(ws, KS_FILTER_COMPACT, 2 * pc_total * sizeof(pos_t));
            hipLaunchKernelGGL(filter_gather_counts_kernel, dim3((uint32_t)((runs + 255) / 256)), dim3(256), 0, st, d_tcidx, d_trun0,
                               (uint32_t)t_seg.size(), fg->d_crun0, fg->d_runcnt, d_cnt);
            VLG_HIP_TRY(rocprim::exclusive_scan(d_scan, scan_tmp, d_cnt, d_off, 0u, runs, rocprim::plus<uint32_t>(), st));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_compact_kernel<pos_t>), dim3((uint32_t)(((runs + kCompactRuns - 1) / kCompactRuns + 3) / 4)),
                               dim3(256), 0, st, P, fg->d_segs, d_tseg, d_trun0, (uint32_t)t_seg.size(), fg->d_abits, d_cnt, d_off, Pc, ~0ull, ws->compact_dense_min);
        }
        // fences of the survivors' lists: whole blocks of [Pc, Pc + pc_total) (Pc starts on a block)
        if (ws->fences && pc_total >= 64) {
            const uint64_t g0 = (uint64_t)(Pc - P) / 64;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(fence_build_kernel<pos_t>), dim3(grid_for(pc_total / 64, 8192)), dim3(256), 0, st, P, g0, g0 + pc_total / 64,
                               static_cast<pos_t*>(ws->fences));
        }
        VLG_HIP_TRY(hipGetLastError());
    }
    const pos_t* F = static_cast<const pos_t*>(ws->fences);
    jt.mark("  chunk: compaction launched");
    std::vector<uint32_t> cls_count(kmax + 1, 0), cls_first(kmax + 2, 0);   // segments per dist class
    for (uint64_t qi = q0; qi < q1; ++qi) {
        uint32_t k = qm[qi - q0].k;
        if (!(k > 0 && eo(q->qsub[qi]) > 0)) continue;
        for (uint32_t i = 0; i < k; ++i) cls_count[k - 1 - i]++;
    }
    // classes are stored from the highest dist down to 0
    uint32_t nlive = 0;
    for (int d = (int)kmax - 1; d >= 0; --d) { cls_first[d] = nlive; nlive += cls_count[d]; }
    svec<SegMeta> sm(nlive + 1);
    svec<uint32_t> seg_begin(nlive + 2, 0);
    std::vector<uint32_t> fill(kmax + 1, 0);
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
    for (uint64_t qi = q0; qi < q1; ++qi) {
        uint32_t k = qm[qi - q0].k;
        if (qm[qi - q0].seg0 == kNone) continue;
        for (uint32_t i = 0; i < k; ++i) {
            uint64_t s = q->qsub[qi] + i;
            SegMeta& m = sm[seg_of_sub[s - s0]];
            m.level = i; m.dist = k - 1 - i;
            m.lo = q->lo[s]; m.hi = q->hi[s];
            if (filtered(s) && fg->speculated) {                 // compacted with its whole group: segment c starts at cpre[c]
                m.pbegin = (uint32_t)((Pc - P) + fg->cpre[fg->cidx[s - fg->sub0]]);
            } else if (filtered(s)) {                            // private list of the query: the survivors, compacted behind P
                m.pbegin = (uint32_t)((Pc - P) + pc_used);                 // the order of the compaction tasks
                pc_used += eo(s);
            } else {
                m.pbegin = poff[s];
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
    jt.mark("  chunk: compaction");
    // ---- carve the arena ---------------------------------------------------------------------------
    uint32_t* link = ws->tuples ? A.take<uint32_t>(T) : nullptr;      // only the tuples walk the links
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
    svec<uint32_t> rec_begin(nq + 1, 0), rec_query;
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
    if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (join)");
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
            if (dist == 1)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(join_link_kernel<pos_t, true>), dim3(runs_grid(b1 - b0, kLinkRun)), dim3(256), 0, st, P, F, d_segb, nlive, d_sm,
                                   (uint32_t)b0, (uint32_t)b1, fref, lvl_ptr[0], endp, link);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(join_link_kernel<pos_t, false>), dim3(runs_grid(b1 - b0, kLinkRun)), dim3(256), 0, st, P, F, d_segb, nlive, d_sm,
                                   (uint32_t)b0, (uint32_t)b1, fref, lvl_ptr[0], endp, link);
        }
        if (vlg_status s = summarize_class(dist)) return s;
    }
    {
        Timed t(ws, KS_JOIN_CHAIN, 8ull * (lvl0_end - lvl0_begin));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(join_jump_kernel<pos_t>), dim3(runs_grid(lvl0_end - lvl0_begin)), dim3(256), 0, st, P, F, d_segb, nlive, d_sm,
                           d_qm, lvl0_begin, lvl0_end, fref, endp, jump, d_qstart);
        if (t0 < lvl0_begin) VLG_HIP_TRY(hipMemsetAsync(jump + t0, 0xFF, (lvl0_begin - t0) * 4, st));   // slots of the first tile before the range
        hipLaunchKernelGGL(chain_tiles_kernel, dim3((uint32_t)((lvl0_end - t0 + kTile - 1) / kTile)), dim3(256), 0, st, jump, t0, lvl0_end, xh);
        hipLaunchKernelGGL(chain_walk_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, d_sm, d_qm, nq, d_qstart, xh, d_recb, d_rec, d_recc,
                           d_counts);
        if (n_rec)
            hipLaunchKernelGGL(chain_emit_kernel, dim3((n_rec + 3) / 4), dim3(256), 0, st, d_recb, d_recc, d_rec, n_rec, d_recq, jump, xh, lvl0_end, mlist);
    }
    VLG_HIP_TRY(hipGetLastError());
    // ---- sizes of the result, then gather -------------------------------------------------------------
    svec<unsigned long long> counts(nq);
    VLG_HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, nq * 8, hipMemcpyDeviceToHost, st));
    VLG_HIP_TRY(hipStreamSynchronize(st));
    jt.mark("  chunk: link + chain");
    uint64_t M = 0, TV = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        qm[i].out_first = M; qm[i].out_tuple = TV;
        M += counts[i]; TV += ws->tuples ? counts[i] * qm[i].k : 0;
        res->counts[q0 + i] = counts[i];
    }
    piece.matches = M; piece.tuple_vals = TV;
    if (M) {
        piece.width = sizeof(pos_t);
        VLG_HIP_TRY(result_alloc(&piece.d_first, M * sizeof(pos_t), &piece.first_bytes));
        if (TV) VLG_HIP_TRY(result_alloc(&piece.d_tuples, TV * sizeof(pos_t), &piece.tuple_bytes));
        res->pieces.push_back(piece);
        jt.mark("  chunk: result malloc");
        VLG_HIP_TRY(hipMemcpyAsync(d_qm, qm.data(), nq * sizeof(QueryMeta), hipMemcpyHostToDevice, st));
        svec<unsigned long long> first_of(nq);                      // where every query's matches start, dense (the counts have been read: their place is free)
        for (uint32_t i = 0; i < nq; ++i) first_of[i] = qm[i].out_first;
        VLG_HIP_TRY(hipMemcpyAsync(d_counts, first_of.data(), nq * 8, hipMemcpyHostToDevice, st));
        Timed t(ws, KS_GATHER, sizeof(pos_t) * (M + TV));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(join_gather_kernel<pos_t>), dim3(runs_grid(M)), dim3(256), 0, st, P, d_sm, d_qm, d_counts, nq, M, link, mlist,
                           static_cast<pos_t*>(piece.d_first), static_cast<pos_t*>(piece.d_tuples), d_stats + kStatsChecksum);
        VLG_HIP_TRY(hipGetLastError());
    } else {
        res->pieces.push_back(piece);
    }
    VLG_HIP_TRY(hipStreamSynchronize(st));     // qm / counts host buffers are read by the async copies above
    jt.mark("  chunk: gather");
    res->sum.n_matches += M;
    res->sum.n_tuple_values += TV;
    res->sum.n_chunks++;
    res->sum.join_slots += T;
    return VLG_OK;
}

// ---- planning and running the joins of the queries [Q0,Q1) over lists that are already in P --------------------------------
// cost of a query in bytes of join scratch / in join slots, for list lengths given by occ_of(sub-pattern)
template <class F>
uint64_t join_bytes_of(const vlg_queries* q, uint64_t qi, F&& occ_of)
{
    const bool uniform_k = q->kmin == q->kmax;
    uint32_t k = (uint32_t)(q->qsub[qi + 1] - q->qsub[qi]);
    uint64_t t = 0;
    for (uint32_t i = 0; i < k; ++i) if (i + 1 < k || k == 1) t += occ_of(q->qsub[qi] + i);
    // list-0 arrays cover the slot range of all lists 0, which is exactly those slots when every query has the same k
    uint64_t t0s = k ? (uniform_k ? occ_of(q->qsub[qi]) : t) : 0;
    return t * kJoinBytesPerSlot + t0s * kJoinBytesPerSlot0;
}
template <class F>
uint64_t join_slots_of(const vlg_queries* q, uint64_t qi, F&& occ_of)        // slot indices are 32-bit inside a chunk
{
    uint32_t k = (uint32_t)(q->qsub[qi + 1] - q->qsub[qi]);
    uint64_t t = 64ull * k;                                           // class alignment slack
    for (uint32_t i = 0; i < k; ++i) if (i + 1 < k || k == 1) t += occ_of(q->qsub[qi] + i);
    return t;
}

struct JoinPlan {
    std::vector<uint64_t> fbytes;        // per query of [Q0,Q1): bytes of filter state (0 = joined on its full lists)
    uint64_t group_cap = 0;              // filter state of the queries filtered together
    uint64_t filter_need = 0;            // arena bytes the filter of one group may take
    uint64_t want_bytes = 0;             // join scratch of one chunk
    uint64_t logical_max_query = 0;
    uint64_t meta = 0;                   // per-chunk metadata on top
};

// join_budget = arena bytes left behind the lists; n_positions bounds every list element (sizes the block bitmaps of the filter)
vlg_status plan_joins(const vlg_queries* q, const Plan& pl, const vlg_workspace* ws, uint64_t Q0, uint64_t Q1, uint64_t join_budget,
                      uint64_t n_positions, JoinPlan& jp)
{
    auto full = [&](uint64_t s) -> uint64_t { return pl.occ[s]; };
    uint64_t logical_total = 0;
    jp.logical_max_query = 0;
    {
        uint64_t part_sum[kHostThreads] = {0}, part_max[kHostThreads] = {0};
        parallel_slices(Q0, Q1, 1u << 16, [&](uint64_t a, uint64_t b, uint32_t th) {
            uint64_t sum = 0, mx = 0;
            for (uint64_t qi = a; qi < b; ++qi) {
                const uint64_t t = join_bytes_of(q, qi, full);
                sum += t;
                mx = std::max(mx, t);
            }
            part_sum[th] = sum; part_max[th] = mx;
        });
        for (uint32_t th = 0; th < kHostThreads; ++th) { logical_total += part_sum[th]; jp.logical_max_query = std::max(jp.logical_max_query, part_max[th]); }
    }
    // window filter: state of the filtered queries of a group (at most a third of the budget), dropped query by query if it
    // would not leave room for the largest unfiltered join
    const uint64_t nbw = (((n_positions >> filter_block_shift(n_positions)) + 1) + 63) / 64;
    jp.group_cap = ws->filter_group_bytes ? std::min<uint64_t>(ws->filter_group_bytes, join_budget / 3) : join_budget / 3;
    jp.fbytes.assign(Q1 - Q0, 0);
    uint64_t filter_total = 0;
    uint64_t filter_runs = 0;                                             // runs the compaction of one chunk may have to index
    if (ws->filter && jp.logical_max_query + jp.group_cap <= join_budget) {
        uint64_t part_bytes[kHostThreads] = {0}, part_runs[kHostThreads] = {0}, part_slots[kHostThreads] = {0};
        parallel_slices(Q0, Q1, 1u << 16, [&](uint64_t a, uint64_t b_, uint32_t th) {
            uint64_t bytes = 0, runs = 0, slots = 0;
            for (uint64_t qi = a; qi < b_; ++qi) {
                uint64_t b = filter_bytes(q, pl, ws, qi, nbw);
                if (b > jp.group_cap) b = 0;
                jp.fbytes[qi - Q0] = b;
                bytes += b;
                if (b) for (uint64_t s = q->qsub[qi]; s + 1 < q->qsub[qi + 1]; ++s) { runs += pl.occ[s] / kRun + 1; slots += pl.occ[s]; }
            }
            part_bytes[th] = bytes; part_runs[th] = runs; part_slots[th] = slots;
        });
        uint64_t cand_slots = 0;
        for (uint32_t th = 0; th < kHostThreads; ++th) { filter_total += part_bytes[th]; filter_runs += part_runs[th]; cand_slots += part_slots[th]; }
        // The filter has a host cost per query of the batch (its tables are laid out query by query, ~25 ns each) and saves at most
        // the join of the candidates' slots (~12 ps each): a batch of very many small queries (BASELINE config 4: 10^6 queries, a few
        // thousand slots in those that qualify) is joined as it is.  filter_min = 0 (tests) keeps every candidate.
        if (cand_slots < (Q1 - Q0) * (ws->filter_min / 2)) {
            std::fill(jp.fbytes.begin(), jp.fbytes.end(), 0);
            filter_total = 0; filter_runs = 0;
        }
    }
    jp.filter_need = std::min(filter_total, jp.group_cap) + filter_runs * 8;
    if (jp.filter_need >= join_budget || jp.logical_max_query > join_budget - jp.filter_need) {
        // the filter state would not leave room for the largest join: these queries are joined on their full lists
        std::fill(jp.fbytes.begin(), jp.fbytes.end(), 0);
        jp.filter_need = 0;
    }
    const uint64_t cap_bytes = join_budget - jp.filter_need;
    if (jp.logical_max_query > cap_bytes)
        return fail(VLG_E_WORKSPACE, "a query needs " + std::to_string(jp.logical_max_query) + " bytes of join scratch; workspace cap allows " +
                                         std::to_string(cap_bytes));
    jp.want_bytes = std::min<uint64_t>(logical_total, cap_bytes);
    jp.meta = (q->qsub[Q1] - q->qsub[Q0] + 4) * (sizeof(SegMeta) + 48) + (Q1 - Q0 + 4) * (sizeof(QueryMeta) + 96) +
              (jp.want_bytes / 8192 + (Q1 - Q0) + 8) * 48 + (1ull << 20);
    return VLG_OK;
}

// groups of queries that share one run of the filter; join chunks inside a group.  A = arena behind the lists; Pc / pc_cap = where
// (inside the allocation of P) the survivors of filtered lists may go.
template <typename pos_t>
vlg_status run_joins(uint64_t n_positions, const vlg_queries* q, vlg_workspace* ws, vlg_result* res, const Plan& pl,
                     const std::vector<uint32_t>& poff, const pos_t* P, const Arena& A, pos_t* Pc, uint64_t pc_cap, uint64_t Q0, uint64_t Q1,
                     const JoinPlan& jp, unsigned long long* d_stats, PhaseTrace& tr)
{
    const uint64_t max_chunk_slots = 0xF0000000ull;
    uint64_t g0 = Q0;
    while (g0 < Q1) {
        Arena GA = A;
        FilterGroup fg;
        fg.g0 = g0; fg.pc_cap = pc_cap;
        uint64_t fb = 0, g1 = g0;
        while (g1 < Q1) {
            const uint64_t b = pc_cap ? jp.fbytes[g1 - Q0] : 0;
            if (fb + b > jp.group_cap && g1 > g0) break;
            fb += b;
            fg.want.push_back(b > 0);
            ++g1;
        }
        fg.g1 = g1;
        const FilterGroup* fgp = nullptr;
        if (fb) {
            if (vlg_status s = filter_group<pos_t>(n_positions, q, ws, pl, poff, P, GA, fg, Pc)) return s;
            if (fg.any) fgp = &fg;
            tr.mark("filter group");
        }
        // one walk over the lists of a query: its join scratch (join_bytes_of), its slots (join_slots_of), the survivors it puts into Pc
        const bool uniform_k = q->kmin == q->kmax;
        auto cost_of = [&](uint64_t qi, uint64_t& bytes, uint64_t& slots, uint64_t& pc) {
            const uint64_t a = q->qsub[qi];
            const uint32_t k = (uint32_t)(q->qsub[qi + 1] - a);
            uint64_t t = 0, first = 0;
            pc = 0;
            for (uint32_t i = 0; i < k; ++i) {
                const uint64_t e = fgp ? fgp->eff[a + i - fgp->sub0] : pl.occ[a + i];
                if (i == 0) first = e;
                if (i + 1 < k || k == 1) t += e;
                if (fgp && !fgp->speculated && fgp->cidx[a + i - fgp->sub0] != kNone) pc += e;      // (speculated: the survivors are in Pc already)
            }
            bytes = t * kJoinBytesPerSlot + (k ? (uniform_k ? first : t) : 0) * kJoinBytesPerSlot0;
            slots = t + 64ull * k;
        };
        uint64_t q0 = g0;
        while (q0 < g1) {
            uint64_t T = 0, S = 0, C = 0, q1 = q0;
            while (q1 < g1) {
                uint64_t t, sl, pc;
                cost_of(q1, t, sl, pc);
                if (sl > max_chunk_slots) return fail(VLG_E_WORKSPACE, "a query has more than 2^32 join slots");
                if (((T + t > jp.want_bytes || S + sl > max_chunk_slots || C + pc > pc_cap) && q1 > q0) || (q1 - q0) >= (1u << 22)) break;
                T += t; S += sl; C += pc;
                ++q1;
            }
            vlg_status s = run_join_chunk<pos_t>(q, ws, res, q0, q1, pl, poff, P, GA, d_stats, fgp, Pc);
            if (s) return s;
            tr.mark("join chunk");
            q0 = q1;
        }
        g0 = g1;
    }
    return VLG_OK;
}

// A super-chunk = a run of queries whose distinct occurrence lists fit the physical budget; inside it
// the queries are joined in chunks bounded by the logical budget.
template <typename pos_t>
vlg_status run_batch(const vlg_index* idx, const vlg_queries* q, vlg_workspace* ws, vlg_result* res, const Plan& pl,
                     unsigned long long* d_stats, bool wide)
{
    const uint64_t fixed = 8ull << 20;      // alignment slack + per-chunk metadata
    if (ws->cap_bytes <= 2 * fixed) return fail(VLG_E_WORKSPACE, "workspace cap too small");
    const uint64_t budget = ws->cap_bytes - fixed;
    // physical lists take at most half of the budget (two buffers during the sort)
    const uint64_t phys_per = sizeof(pos_t) + kPhysScratchPerElem<pos_t>();
    const uint64_t phys_cap = std::min<uint64_t>(budget / 2 / (phys_per + 1), 0xFFFFFF00ull);
    std::vector<uint32_t> stamp(pl.dl.size(), 0xFFFFFFFFu);
    std::vector<uint32_t> poff(pl.dl.size(), 0);
    std::vector<uint32_t> poff_sub(q->nsub, 0);
    uint64_t Q0 = 0;
    uint32_t epoch = 0;
    PhaseTrace tr(ws->stream);
    while (Q0 < q->nq) {
        // ---- choose the super-chunk ----------------------------------------------------------------
        std::vector<uint32_t> dlist;
        dlist.reserve(pl.dl.size());
        uint64_t phys = 0, Q1 = Q0;
        if (Q0 == 0 && ws->dedup) {
            // the usual case, without a walk over the sub-patterns: every distinct list of the batch fits one super-chunk
            // (with dedup every id belongs to a live sub-pattern; the ids are already in ascending order)
            uint64_t all = 0;
            for (uint64_t d = 0; d < pl.docc.size() && all <= phys_cap; ++d) all += pl.docc[d];
            if (all <= phys_cap) {
                phys = all;
                Q1 = q->nq;
                dlist.resize(pl.dl.size());
                for (uint32_t d = 0; d < (uint32_t)pl.dl.size(); ++d) dlist[d] = d;
                std::fill(stamp.begin(), stamp.end(), epoch);
            }
        }
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
        // distinct lists in ascending id order: SA order when the plan came from the device (ids are ranks of the intervals), which is
        // the order the sorted sweep wants to start in
        if (dlist.size() > 1) {
            if (dlist.size() * 8 >= pl.dl.size()) {               // most ids are used: walk the stamps instead of sorting
                size_t w = 0;
                for (uint32_t d = 0; d < (uint32_t)pl.dl.size(); ++d) if (stamp[d] == epoch - 1) dlist[w++] = d;
            } else std::sort(dlist.begin(), dlist.end());
        }
        // ---- arena: physical lists first, filter state and join scratch behind them ---------------------
        size_t sort_tmp = 0;
        // (the library's size queries are not free -- a few hundred microseconds each -- and batches repeat: the last answer is kept)
        const uint64_t tmp_key[6] = {phys, dlist.size(), idx->hdr.n, ((uint64_t)idx->hdr.sigma << 8) | ((uint64_t)ws->sweep << 1) | (uint64_t)idx->is_int | (sizeof(pos_t) << 4),
                                     ws->global_sort_min, ws->sweep_min};
        if (phys && !memcmp(tmp_key, ws->tmp_key, sizeof tmp_key)) sort_tmp = ws->tmp_bytes;
        else if (phys) {
            // the sort build_physical will choose (same condition there): one radix sort of (list, position) keys, or a segmented one
            const unsigned pos_bits = std::max(1u, bit_width64(idx->hdr.n >= 2 ? idx->hdr.n - 2 : 0));
            if (phys >= ws->global_sort_min && pos_bits + bit_width64(dlist.size()) <= 64) {
                rocprim::double_buffer<uint64_t> nk(nullptr, nullptr);
                VLG_HIP_TRY(rocprim::radix_sort_keys(nullptr, sort_tmp, nk, phys, 0, 64, ws->stream));
            } else {
                pos_t* np = nullptr; uint32_t* nu = nullptr;
                VLG_HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, sort_tmp, np, np, (unsigned)phys, (unsigned)dlist.size(), nu, nu, 0,
                                                               pos_bits, ws->stream));
            }
            if (ws->sweep && phys >= ws->sweep_min && (!idx->is_int || int_sweep_possible(idx->iview)))
                sort_tmp = std::max(sort_tmp, sweep_temp_bytes(phys, idx->hdr.sigma, ws->stream));
            memcpy(ws->tmp_key, tmp_key, sizeof tmp_key);
            ws->tmp_bytes = sort_tmp;
        }
        const bool will_sweep = ws->sweep && phys >= ws->sweep_min && (!idx->is_int || int_sweep_possible(idx->iview));
        // per query: the largest join, and how many pivot elements the window filter will search from (for the ladder, below)
        uint64_t logical_max_query = 0, pivot_elems = 0;
        {
            uint64_t part_max[kHostThreads] = {0}, part_piv[kHostThreads] = {0};
            const bool count_pivots = ws->filter && ws->filter_pivot && ws->pivot_rungs;
            parallel_slices(Q0, Q1, 1u << 16, [&](uint64_t a, uint64_t b, uint32_t t) {
                uint64_t mx = 0, pv_sum = 0;
                for (uint64_t qi = a; qi < b; ++qi) {
                    mx = std::max(mx, join_bytes_of(q, qi, [&](uint64_t s) -> uint64_t { return pl.occ[s]; }));
                    uint32_t pv = 0;
                    if (count_pivots && filter_mode(q, pl, ws, qi, &pv) == 2) pv_sum += pl.occ[q->qsub[qi] + pv];
                }
                part_max[t] = mx; part_piv[t] = pv_sum;
            });
            for (uint32_t t = 0; t < kHostThreads; ++t) { logical_max_query = std::max(logical_max_query, part_max[t]); pivot_elems += part_piv[t]; }
        }
        // the member bit-vector (32 B per 224 SA indices) and the records (8 B per occurrence) must leave room for the joins
        // (an index that keeps the whole suffix array -- SA-order samples of density 1 -- walks nothing: no shared steps, no records)
        const bool dense_sa = idx->hdr.dens == 1 && idx->hdr.sampling == kSamplingSaOrder;
        // (nor does a batch dense enough to rebuild the whole suffix array, K3u: its arrays fit the sweep's scratch)
        const bool will_unsample = will_sweep && !dense_sa && sizeof(pos_t) == 4 && unsample_applies(idx, ws, phys / (uint64_t)std::max(1, ws->x_ranks));
        uint64_t trail_bytes = will_sweep && ws->trail && ws->dedup && !dense_sa && !will_unsample ? member_blocks(idx->hdr.n) * sizeof(Block) + phys * 9 + 2048 : 0;
        // (+ fences: < 1 B per element; + the pivot filter's ladder, a third of the lists, when its searches outweigh building it:
        // one pass over the lists against two descents per pivot element)
        ws->want_rungs = pivot_elems && (ws->pivot_rungs == 2 || (pivot_elems >= phys / 16 && pivot_elems >= 4096));
        const uint64_t rung_bytes = ws->want_rungs ? rung_layout(phys).entries * sizeof(pos_t) + 8192 : 0;
        const uint64_t phys_plain = phys * phys_per + sort_tmp + (dlist.size() + 2) * 24 + (8ull << 20) + phys + rung_bytes;
        if (trail_bytes) {
            const uint64_t left = budget > phys_plain + trail_bytes ? budget - phys_plain - trail_bytes : 0;
            if (left < std::max<uint64_t>(2 * logical_max_query, budget / 8)) trail_bytes = 0;
        }
        const bool share_trails = trail_bytes != 0;
        if (tr.on) fprintf(stderr, "[vlg trace] super-chunk: %llu occurrences, %.1f GB physical, LF steps %s (%.1f GB), budget %.1f GB, largest join %.1f GB\n",
                           (unsigned long long)phys, phys_plain / 1e9, share_trails ? "shared" : "not shared",
                           (member_blocks(idx->hdr.n) * sizeof(Block) + phys * 8) / 1e9, budget / 1e9, logical_max_query / 1e9);
        if (tr.on) {                                                          // where the occurrences are: distinct lists by size class
            uint64_t cnt[40] = {0}, sum[40] = {0};
            for (uint32_t d : dlist) { const unsigned b = bit_width64(pl.docc[d]); cnt[b]++; sum[b] += pl.docc[d]; }
            fprintf(stderr, "[vlg trace] distinct lists by size (2^b: lists/occurrences):");
            for (unsigned b = 0; b < 40; ++b) if (cnt[b]) fprintf(stderr, " %u:%llu/%llu", b, (unsigned long long)cnt[b], (unsigned long long)sum[b]);
            fprintf(stderr, "\n");
        }
        const uint64_t phys_bytes = phys_plain + trail_bytes;
        const uint64_t join_budget = budget > phys_bytes ? budget - phys_bytes : 0;
        // When the arena already holds whatever a plan can ask for (the caller reserved the whole cap), locate + sort are launched
        // first and the joins are planned while the GPU works; otherwise the plan decides how much to allocate.
        const uint64_t meta_upper = (q->qsub[Q1] - q->qsub[Q0] + 4) * (sizeof(SegMeta) + 48) + (Q1 - Q0 + 4) * (sizeof(QueryMeta) + 96) +
                                    (budget / 8192 + (Q1 - Q0) + 8) * 48 + (1ull << 20);
        const bool launch_first = ws->arena_bytes >= budget + meta_upper + 2 * fixed;
        // collective search: the queries of the super-chunk are cut into one contiguous piece per rank of equal join work (slots of the
        // non-final lists, from the list lengths every rank knows); this rank filters and joins its piece [qa, qb) only
        uint64_t qa = Q0, qb = Q1;
        std::vector<uint64_t> xq;
        Agreement ag(ws);
        if (ws->x_ranks > 1) {
            std::vector<uint64_t> cum(Q1 - Q0 + 1, 0);
            for (uint64_t qi = Q0; qi < Q1; ++qi)
                cum[qi - Q0 + 1] = cum[qi - Q0] + 1 + (pl.occ[q->qsub[qi]] ? join_slots_of(q, qi, [&](uint64_t sidx) -> uint64_t { return pl.occ[sidx]; }) : 0);
            auto cut_at = [&](int r) -> uint64_t {
                if (r <= 0) return Q0;
                if (r >= ws->x_ranks) return Q1;
                const uint64_t target = cum.back() / (uint64_t)ws->x_ranks * (uint64_t)r;
                return Q0 + (uint64_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
            };
            xq.resize(ws->x_ranks + 1);
            xq[0] = Q0;
            for (int r = 1; r <= ws->x_ranks; ++r) xq[r] = std::max(xq[r - 1], std::min(cut_at(r), Q1));
            qa = xq[ws->x_rank];
            qb = xq[ws->x_rank + 1];
            res->owned.push_back(qa);
            res->owned.push_back(qb);
        }
        JoinPlan jp;
        if (!launch_first) {
            if (vlg_status s = plan_joins(q, pl, ws, qa, qb, join_budget, idx->hdr.n, jp)) { ag.mine = s; return s; }
            if (vlg_status s = ws_reserve(ws, phys_bytes + jp.filter_need + jp.want_bytes + jp.meta + fixed)) { ag.mine = s; return s; }
        }
        Arena A{ws->arena, ws->arena_bytes};
        pos_t* P = nullptr;
        uint64_t Tphys = 0;
        tr.mark("plan super-chunk");
        pos_t* Pc = nullptr;
        uint64_t pc_cap = 0;
        if (vlg_status s = build_physical<pos_t>(idx, ws, res, dlist, pl, A, P, poff, Tphys, sort_tmp, d_stats, Pc, pc_cap, share_trails, wide, will_unsample, q,
                                                   ws->x_ranks > 1 ? &xq : nullptr, ws->x_ranks > 1 ? &ag : nullptr)) return s;
        if (launch_first)
            if (vlg_status s = plan_joins(q, pl, ws, qa, qb, join_budget, idx->hdr.n, jp)) return s;
        tr.mark("locate + sort");
        for (uint64_t s = q->qsub[Q0]; s < q->qsub[Q1]; ++s) poff_sub[s] = pl.occ[s] ? poff[pl.did[s]] : 0;
        if (vlg_status s = run_joins<pos_t>(idx->hdr.n, q, ws, res, pl, poff_sub, P, A, Pc, pc_cap, qa, qb, jp, d_stats, tr)) return s;
        Q0 = Q1;
    }
    return VLG_OK;
}

}  // namespace

namespace {

// ---- distinct SA intervals of a batch, found on the device ----------------------------------------------------------------------
// key of a sub-pattern: l << kbits | (occurrences - 1), ~0 for the sub-patterns of a query that has an empty list.  kbits = 64 - (bits
// of the largest SA index): 32 for n <= 2^32, 31 for BASELINE config 4 (n = 2^32 + 1); an interval too long for its field raises
// *overflow and the caller plans on the host instead.
__global__ void interval_keys_kernel(const uint64_t* __restrict__ l, const uint64_t* __restrict__ r, const uint64_t* __restrict__ qsub,
                                     uint64_t nq, uint32_t kbits, uint64_t* __restrict__ keys, uint32_t* __restrict__ sub, uint32_t* __restrict__ overflow)
{
    for (uint64_t qi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; qi < nq; qi += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = qsub[qi], b = qsub[qi + 1];
        bool live = b > a;
        for (uint64_t s = a; s < b && live; ++s) live = r[s] + 1 - l[s] > 0;
        for (uint64_t s = a; s < b; ++s) {
            if (live && ((r[s] - l[s]) >> kbits)) *overflow = 1;
            keys[s] = live ? (l[s] << kbits) | (r[s] - l[s]) : ~0ull;
            sub[s] = (uint32_t)s;
        }
    }
}
__global__ void interval_heads_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ head)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        head[j] = (keys[j] != ~0ull && (j == 0 || keys[j] != keys[j - 1])) ? 1u : 0u;
}
// gid = inclusive scan of the heads: the (gid-1)-th distinct interval, in ascending SA order.  Also every sub-pattern's list length
// (0 in a query that has an empty list) and their sum.
__global__ void interval_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ sub, const uint32_t* __restrict__ gid,
                                        uint64_t n, uint32_t kbits, uint32_t* __restrict__ did, uint64_t* __restrict__ occ, uint64_t* __restrict__ dl,
                                        uint64_t* __restrict__ docc, unsigned long long* __restrict__ logical)
{
    unsigned long long local = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[j];
        if (k == ~0ull) { did[sub[j]] = 0; occ[sub[j]] = 0; continue; }
        const uint32_t g = gid[j] - 1;
        const uint64_t len = (k & ((1ull << kbits) - 1)) + 1;
        did[sub[j]] = g;
        occ[sub[j]] = len;
        local += len;
        if (j == 0 || keys[j - 1] != k) { dl[g] = k >> kbits; docc[g] = len; }
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(logical, local);
}

// Plan of a batch (which sub-patterns are live, which distinct interval each one is) from the intervals in d_l / d_r.
// Identical SA intervals are the same occurrence list: each distinct one is located + sorted once per super-chunk and shared.
// The distinct intervals are numbered in ascending SA order, so the sweep that locates them starts globally sorted.
// *fallback is set (and nothing planned) when an interval does not fit its key field: the caller then plans on the host.
// device scratch plan_on_device needs for nsub sub-patterns of nq queries
uint64_t plan_device_bytes(uint64_t nsub, uint64_t nq, hipStream_t st, size_t* sort_tb_out = nullptr, size_t* scan_tb_out = nullptr)
{
    size_t sort_tb = 0, scan_tb = 0;
    rocprim::double_buffer<uint64_t> nk(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> nv(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, sort_tb, nk, nv, nsub, 0, 64, st);
    uint32_t* nu = nullptr;
    (void)rocprim::inclusive_scan(nullptr, scan_tb, nu, nu, nsub, rocprim::plus<uint32_t>(), st);
    if (sort_tb_out) *sort_tb_out = sort_tb;
    if (scan_tb_out) *scan_tb_out = scan_tb;
    const uint64_t n8 = align_up(nsub * 8, 256), n4 = align_up(nsub * 4, 256);
    return 4 * n8 + 5 * n4 + align_up((nq + 1) * 8, 256) + align_up(std::max(sort_tb, scan_tb), 256) + 1024 + 256;
}

vlg_status plan_on_device(const vlg_queries* q, vlg_workspace* ws, const uint64_t* d_l, const uint64_t* d_r, uint64_t n, Plan& pl, uint64_t& logical,
                          bool* fallback, uint8_t* mem /* plan_device_bytes() of device scratch */, uint64_t mem_bytes)
{
    *fallback = false;
    const uint32_t kbits = 64 - std::max(32u, bit_width64(n - 1));
    hipStream_t st = ws->stream;
    const uint64_t nsub = q->nsub, nq = q->nq;
    pl.occ.resize(nsub);                 // (filled by the copies below: every sub-pattern gets a value on the device)
    pl.did.resize(nsub);
    if (!nsub) return VLG_OK;
    size_t sort_tb = 0, scan_tb = 0;
    const uint64_t n8 = align_up(nsub * 8, 256), n4 = align_up(nsub * 4, 256);
    const uint64_t bytes = plan_device_bytes(nsub, nq, st, &sort_tb, &scan_tb);
    if (!mem || mem_bytes < bytes) return fail(VLG_E_INTERNAL, "interval plan: scratch too small");
    uint64_t* keys_a = (uint64_t*)mem;
    uint64_t* keys_b = (uint64_t*)(mem + n8);
    uint64_t* d_dl = (uint64_t*)(mem + 2 * n8);
    uint64_t* d_docc = (uint64_t*)(mem + 3 * n8);
    uint32_t* sub_a = (uint32_t*)(mem + 4 * n8);
    uint32_t* sub_b = (uint32_t*)(mem + 4 * n8 + n4);
    uint32_t* d_head = (uint32_t*)(mem + 4 * n8 + 2 * n4);
    uint32_t* d_gid = (uint32_t*)(mem + 4 * n8 + 3 * n4);
    uint32_t* d_did = (uint32_t*)(mem + 4 * n8 + 4 * n4);
    void* d_tmp = mem + 4 * n8 + 5 * n4 + align_up((nq + 1) * 8, 256);
    uint32_t* d_flags = (uint32_t*)(mem + bytes - 256);          // [0] overflow, [2..3] sum of the list lengths
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipMemsetAsync(d_flags, 0, 16, st));
        hipLaunchKernelGGL(interval_keys_kernel, dim3(grid_for(nq, 2048)), dim3(256), 0, st, d_l, d_r, q->d_qsub, nq, kbits, keys_a, sub_a, d_flags);
        rocprim::double_buffer<uint64_t> dk(keys_a, keys_b);
        rocprim::double_buffer<uint32_t> dv(sub_a, sub_b);
        size_t tb = sort_tb;
        VLG_HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tb, dk, dv, nsub, 0, 64, st));
        hipLaunchKernelGGL(interval_heads_kernel, dim3(grid_for(nsub, 2048)), dim3(256), 0, st, dk.current(), nsub, d_head);
        tb = scan_tb;
        VLG_HIP_TRY(rocprim::inclusive_scan(d_tmp, tb, d_head, d_gid, nsub, rocprim::plus<uint32_t>(), st));
        uint64_t* d_occ = dk.alternate();                        // the sort's other key buffer is free now
        hipLaunchKernelGGL(interval_scatter_kernel, dim3(grid_for(nsub, 2048)), dim3(256), 0, st, dk.current(), dv.current(), d_gid, nsub, kbits, d_did,
                           d_occ, d_dl, d_docc, reinterpret_cast<unsigned long long*>(d_flags + 2));
        VLG_HIP_TRY(hipGetLastError());
        svec<uint32_t> h_flags(5, 0);                            // overflow, -, sum lo, sum hi, distinct intervals
        VLG_HIP_TRY(hipMemcpyAsync(h_flags.data(), d_flags, 16, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipMemcpyAsync(h_flags.data() + 4, d_gid + (nsub - 1), 4, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipMemcpyAsync(pl.did.data(), d_did, nsub * 4, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipMemcpyAsync(pl.occ.data(), d_occ, nsub * 8, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        if (h_flags[0]) { *fallback = true; return VLG_OK; }
        const uint32_t nd = h_flags[4];
        pl.dl.resize(nd);
        pl.docc.resize(nd);
        if (nd) {
            VLG_HIP_TRY(hipMemcpyAsync(pl.dl.data(), d_dl, (uint64_t)nd * 8, hipMemcpyDeviceToHost, st));
            VLG_HIP_TRY(hipMemcpyAsync(pl.docc.data(), d_docc, (uint64_t)nd * 8, hipMemcpyDeviceToHost, st));
            VLG_HIP_TRY(hipStreamSynchronize(st));
        }
        logical += (uint64_t)h_flags[2] | ((uint64_t)h_flags[3] << 32);
        return VLG_OK;
    };
    return run();
}

}  // namespace

extern "C" vlg_status vlg_search_batch(const vlg_index* idx, const vlg_queries* q, vlg_workspace* ws, vlg_result** out)
{
    if (!idx || !q || !ws || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (!idx->is_int && q->sym_bytes != 1) return fail(VLG_E_INVALID, "integer-alphabet query batch: it takes an integer-alphabet index (vlg_index_build_int) or vlg_wtsa_*");
    if (idx->is_int && q->sym_bytes != 4 && q->nsub) return fail(VLG_E_INVALID, "an integer-alphabet index takes query batches parsed by vlg_queries_parse_int");
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
    static thread_local std::chrono::steady_clock::time_point last_end;
    if (tr.on && last_end.time_since_epoch().count())
        fprintf(stderr, "[vlg trace] %-28s %9.3f ms\n", "(between two batches)",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - last_end).count());
    auto run = [&]() -> vlg_status {
        const uint64_t nsub = q->nsub;
        // SA intervals, counters and the interval plan's arrays live in the workspace's head buffer (no allocation per batch)
        const bool may_plan_on_device = ws->dedup && idx->hdr.n <= (1ull << 33) && nsub > 0 && nsub < 0xFFFFFFF0ull;
        const uint64_t lr_bytes = align_up((nsub + 1) * 8, 256), st_bytes = align_up(kStatsWords * 8, 256);
        const uint64_t plan_bytes = may_plan_on_device ? plan_device_bytes(nsub, q->nq, st) : 0;
        uint8_t* head = nullptr;
        if (vlg_status hs = ws_head(ws, 2 * lr_bytes + st_bytes + plan_bytes, &head)) return hs;
        d_l = (uint64_t*)head;
        d_r = (uint64_t*)(head + lr_bytes);
        d_stats = (unsigned long long*)(head + 2 * lr_bytes);
        uint8_t* plan_mem = head + 2 * lr_bytes + st_bytes;
        VLG_HIP_TRY(hipMemsetAsync(d_stats, 0, kStatsWords * 8, st));
        ws->sample_reads = 0;
        ws->x_agreed = 0;
        // ---- K2: every sub-pattern's SA interval ------------------------------------------------------
        {
            Timed t(ws, KS_BSEARCH, 0);
            if (vlg_status s = idx->is_int ? launch_int_backward_search(idx->iview, q->d_blob, q->d_suboff, nsub, d_l, d_r, d_stats + 3, st)
                                           : launch_backward_search(idx->view, q->d_blob, q->d_suboff, nsub, d_l, d_r, d_stats + 3, st)) return s;
        }
        tr.mark("backward search");
        Plan pl;
        bool device_plan = may_plan_on_device;
        if (device_plan) {
            bool fallback = false;
            if (vlg_status s = plan_on_device(q, ws, d_l, d_r, idx->hdr.n, pl, res->sum.logical_occurrences, &fallback, plan_mem, plan_bytes)) return s;
            if (fallback) { device_plan = false; pl = Plan(); res->sum.logical_occurrences = 0; }
        }
        if (!device_plan) {
        svec<uint64_t> l(nsub), r(nsub);
        if (nsub) {
            VLG_HIP_TRY(hipMemcpyAsync(l.data(), d_l, nsub * 8, hipMemcpyDeviceToHost, st));
            VLG_HIP_TRY(hipMemcpyAsync(r.data(), d_r, nsub * 8, hipMemcpyDeviceToHost, st));
        }
        VLG_HIP_TRY(hipStreamSynchronize(st));
        // a query with an empty occurrence list has no match: none of its lists is materialised
        // (vlg_index.hpp:315-316 returns at the first empty range)
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
        }
        tr.mark("intervals to host + plan");
        // SA indices are wide (33 bits, 64-bit samples) for n > 2^32; text positions still fit 32 bits up to n = 2^32 + 1 (the largest
        // one is n - 2), and then everything behind locate -- sort, fences, filter, join -- runs on 32-bit positions.
        // VLG_FORCE_POS64=1 keeps 64-bit positions on any text (the instantiations for longer texts), =2 only the wide indices.
        const bool wide = idx->hdr.sample_bytes == 8;
        const char* f64 = getenv("VLG_FORCE_POS64");
        const bool pos64 = wide && (idx->hdr.n > (1ull << 32) + 1 || (f64 && f64[0] == '1'));
        vlg_status s = pos64 ? run_batch<uint64_t>(idx, q, ws, res, pl, d_stats, true) : run_batch<uint32_t>(idx, q, ws, res, pl, d_stats, wide);
        if (s) return s;
        unsigned long long hs[kStatsWords];
        VLG_HIP_TRY(hipMemcpyAsync(hs, d_stats, sizeof hs, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        res->sum.lf_steps = hs[0];
        res->sum.wt_levels_locate = hs[1];
        res->sum.checksum = 0;
        for (uint32_t i = 0; i < kChecksumSlots; ++i) res->sum.checksum += hs[kStatsChecksum + i];      // modulo 2^64, like gm_search.cpp:110-114
        res->sum.wt_levels_bsearch = hs[3];
        // algorithmic bytes: 32 B per super-block read (+ one sample per occurrence)
        ws->stats[KS_LOCATE].algorithmic_bytes += 32ull * hs[1] + (uint64_t)idx->hdr.sample_bytes * ws->sample_reads;
        ws->stats[KS_BSEARCH].algorithmic_bytes += 32ull * hs[3];
        return VLG_OK;
    };
    vlg_status stt;
    {
        HostPoolScope staging(&ws->host);
        try { stt = run(); }
        catch (const std::bad_alloc&) { stt = fail(VLG_E_OOM, "out of host memory while planning the batch (pinned staging)"); }
    }
    tr.mark("statistics");
    tr.mark("free");
    last_end = std::chrono::steady_clock::now();
    if (stt && ws->x_ranks > 1 && ws->x_agreed == 0 && q->nq) {
        // a collective search that failed before its first super-chunk: the other ranks wait in that chunk's agreement
        const std::string keep = vlg_last_error();
        Agreement early(ws);
        (void)early.settle(stt);
        set_error(keep);
    }
    if (stt) { vlg_result_destroy(res); return stt; }
    *out = res;
    return VLG_OK;
}

// =============================================================================================
// sdsl::count for every sub-pattern of a batch (what a multi-GPU host shards the batch by)
// =============================================================================================
extern "C" vlg_status vlg_queries_intervals(const vlg_index* idx, const vlg_queries* q, uint64_t* h_l, uint64_t* h_r, void* stream)
{
    if (!idx || !q || (q->nsub && (!h_l || !h_r))) return fail(VLG_E_INVALID, "null argument");
    if (idx->is_int != (q->sym_bytes == 4) && q->nsub) return fail(VLG_E_INVALID, "query batch and index are of different alphabets");
    const uint64_t nsub = q->nsub;
    if (!nsub) return VLG_OK;
    hipStream_t st = (hipStream_t)stream;
    uint64_t* d_lr = nullptr;
    VLG_HIP_TRY(hipMalloc((void**)&d_lr, 2 * nsub * 8));
    vlg_status s = idx->is_int ? launch_int_backward_search(idx->iview, q->d_blob, q->d_suboff, nsub, d_lr, d_lr + nsub, nullptr, st)
                               : launch_backward_search(idx->view, q->d_blob, q->d_suboff, nsub, d_lr, d_lr + nsub, nullptr, st);
    hipError_t e = hipSuccess;
    if (!s) e = hipMemcpyAsync(h_l, d_lr, nsub * 8, hipMemcpyDeviceToHost, st);
    if (!s && e == hipSuccess) e = hipMemcpyAsync(h_r, d_lr + nsub, nsub * 8, hipMemcpyDeviceToHost, st);
    if (!s && e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_lr);
    if (s) return s;
    VLG_HIP_TRY(e);
    return VLG_OK;
}

extern "C" vlg_status vlg_queries_occurrences(const vlg_index* idx, const vlg_queries* q, uint64_t* h_occ, void* stream)
{
    if (!idx || !q || (q->nsub && !h_occ)) return fail(VLG_E_INVALID, "null argument");
    std::vector<uint64_t> r(q->nsub + 1);
    if (vlg_status s = vlg_queries_intervals(idx, q, h_occ, r.data(), stream)) return s;
    for (uint64_t i = 0; i < q->nsub; ++i) h_occ[i] = r[i] + 1 - h_occ[i];        // r + 1 - l (suffix_array_algorithm.hpp:325)
    return VLG_OK;
}

// =============================================================================================
// K5 on caller-provided lists
// =============================================================================================
// K4 stand-alone: every list d_pos[off[l], off[l + 1]) sorted ascending in place (std::sort, index_sasearch.hpp:80) by the per-list sort
// of the batch path (list_sort.hpp), for tests and callers that bring their own lists of 32-bit positions.
extern "C" vlg_status vlg_sort_lists_u32(uint32_t* d_pos, const uint64_t* h_off, uint64_t n_lists, uint32_t position_bits, int mode,
                                         uint64_t* n_clustered, void* stream)
{
    if (!h_off || (n_lists && h_off[n_lists] && !d_pos)) return fail(VLG_E_INVALID, "null argument");
    if (position_bits < 1 || position_bits > 32 || (mode != 1 && mode != 2)) return fail(VLG_E_INVALID, "position_bits in [1,32], mode 1 or 2");
    if (n_lists >= 0xFFFFFFF0ull || h_off[0] != 0) return fail(VLG_E_INVALID, "bad list offsets");
    for (uint64_t l = 0; l < n_lists; ++l) if (h_off[l + 1] < h_off[l]) return fail(VLG_E_INVALID, "list offsets must ascend");
    if (n_clustered) *n_clustered = 0;
    const uint64_t total = h_off[n_lists];
    if (!total) return VLG_OK;
    if (total > 0xFFFFFF00ull) return fail(VLG_E_UNSUPPORTED, "more than 2^32 keys in one vlg_sort_lists_u32 call: split the batch");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available (the VLG library has no CPU fallback)");
    hipStream_t st = (hipStream_t)stream;
    HostPool pool;
    void* d_mem = nullptr;
    auto run = [&]() -> vlg_status {
        HostPoolScope scope(&pool);
        svec<uint64_t> off64(h_off, h_off + n_lists + 1);
        uint64_t n_long = 0, n_tiles = 0, n_chunks = 0;
        for (uint64_t l = 0; l < n_lists; ++l) {
            const uint64_t len = off64[l + 1] - off64[l];
            if (len > kSortTile) { const uint64_t t = (len + kSortTile - 1) / kSortTile; ++n_long; n_tiles += t; n_chunks += (t + kSortChunk - 1) / kSortChunk; }
        }
        const uint64_t bytes = align_up(total * 4, 256) + align_up((n_lists + 1) * 8, 256) + align_up(n_lists * 4, 256) + kSortClasses * 256 +
                               list_sort_scratch_bytes(n_long, n_tiles, n_chunks) + 16 * 256 + 8192;
        VLG_HIP_TRY(hipMalloc(&d_mem, bytes));
        Arena A{reinterpret_cast<uint8_t*>(d_mem), bytes};
        uint32_t* other = A.take<uint32_t>(total);
        uint64_t* d_off = A.take<uint64_t>(n_lists + 1);
        if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (vlg_sort_lists_u32)");
        VLG_HIP_TRY(hipMemcpyAsync(d_off, off64.data(), (n_lists + 1) * 8, hipMemcpyHostToDevice, st));
        ListSortPlan lp;
        if (vlg_status s = list_sort_prepare(off64, (uint32_t)n_lists, A, st, lp)) return s;
        if (vlg_status s = list_sort_enqueue(lp, d_pos, other, d_off, position_bits, st, mode == 1)) return s;
        if (n_clustered && lp.n_long) {
            std::vector<SortList> back(lp.n_long);
            VLG_HIP_TRY(hipMemcpyAsync(back.data(), lp.d_longs, lp.n_long * sizeof(SortList), hipMemcpyDeviceToHost, st));
            VLG_HIP_TRY(hipStreamSynchronize(st));
            for (const SortList& L : back) *n_clustered += L.pad ? 1 : 0;
        }
        VLG_HIP_TRY(hipStreamSynchronize(st));
        return VLG_OK;
    };
    vlg_status s;
    try { s = run(); } catch (const std::bad_alloc&) { s = fail(VLG_E_OOM, "host memory (vlg_sort_lists_u32)"); }
    if (d_mem) { (void)hipStreamSynchronize(st); (void)hipFree(d_mem); }
    pool.release();
    return s;
}

extern "C" vlg_status vlg_join_batch(const uint64_t* d_lists, const uint64_t* h_list_off, uint64_t n_lists, const uint64_t* h_join_list,
                                     const uint64_t* h_lo, const uint64_t* h_hi, const uint64_t* h_end_len, uint64_t n_joins,
                                     vlg_workspace* ws, vlg_result** out)
{
    if (!ws || !out || !h_list_off || !h_join_list || (n_lists && (!h_lo || !h_hi)) || (n_joins && !h_end_len))
        return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available");
    if (h_list_off[0] != 0 || h_join_list[0] != 0 || h_join_list[n_joins] > n_lists) return fail(VLG_E_INVALID, "bad list / join offsets");
    for (uint64_t t = 0; t < n_lists; ++t) if (h_list_off[t + 1] < h_list_off[t]) return fail(VLG_E_INVALID, "list offsets must ascend");
    const uint64_t total = h_list_off[n_lists];
    if (total && !d_lists) return fail(VLG_E_INVALID, "null argument");
    if (total > 0xFFFFFF00ull) return fail(VLG_E_UNSUPPORTED, "more than 2^32 list elements in one vlg_join_batch call: split the batch");
    // the joins as a query batch without patterns: the join passes only look at list lengths, gap bounds and end lengths
    vlg_queries qq;
    qq.nq = n_joins;
    qq.nsub = h_join_list[n_joins];
    qq.qsub.assign(h_join_list, h_join_list + n_joins + 1);
    qq.lo.assign(h_lo, h_lo + qq.nsub);
    qq.hi.assign(h_hi, h_hi + qq.nsub);
    qq.end_len.assign(h_end_len, h_end_len + n_joins);
    qq.kmax = 0; qq.kmin = 0xFFFFFFFFu;
    hipStream_t st = ws->stream;
    vlg_result* res = new vlg_result();
    memset(&res->sum, 0, sizeof res->sum);
    res->sum.n_queries = n_joins;
    res->counts.assign(n_joins, 0);
    res->k.resize(n_joins);
    for (uint64_t j = 0; j < n_joins; ++j) res->k[j] = (uint32_t)(qq.qsub[j + 1] - qq.qsub[j]);
    unsigned long long* d_stats = nullptr;
    auto run = [&]() -> vlg_status {
        // (the plan's arrays are staged memory: they live inside the scope this runs in)
        Plan pl;
        pl.occ.assign(qq.nsub, 0);
        pl.did.assign(qq.nsub, 0);
        std::vector<uint32_t> poff(qq.nsub, 0);
        for (uint64_t j = 0; j < n_joins; ++j) {
            if (qq.qsub[j + 1] < qq.qsub[j] || qq.qsub[j + 1] - qq.qsub[j] > VLG_MAX_SUBPATTERNS) return fail(VLG_E_INVALID, "bad join offsets");
            const uint32_t k = (uint32_t)(qq.qsub[j + 1] - qq.qsub[j]);
            qq.kmax = std::max(qq.kmax, k);
            if (k) qq.kmin = std::min(qq.kmin, k);
            // the reference's loop would not advance with a zero length (index_sasearch.hpp:113)
            if (k && h_end_len[j] == 0) return fail(VLG_E_INVALID, "end_len must be at least 1");
            if (h_end_len[j] >= (1ull << 63)) return fail(VLG_E_INVALID, "end_len out of range");
            bool live = k > 0;
            for (uint64_t s = qq.qsub[j]; s < qq.qsub[j + 1]; ++s) {
                if (s > qq.qsub[j] && (h_lo[s] > h_hi[s] || h_hi[s] >= (1ull << 63))) return fail(VLG_E_INVALID, "bad gap bounds");
                live = live && h_list_off[s + 1] > h_list_off[s];
            }
            // a join with an empty list has no match: none of its lists is looked at (vlg_index.hpp:315-316)
            if (live) for (uint64_t s = qq.qsub[j]; s < qq.qsub[j + 1]; ++s) { pl.occ[s] = h_list_off[s + 1] - h_list_off[s]; poff[s] = (uint32_t)h_list_off[s]; }
        }
        if (qq.kmin == 0xFFFFFFFFu) qq.kmin = 0;
        for (uint64_t s = 0; s < qq.nsub; ++s) res->sum.logical_occurrences += pl.occ[s];
        VLG_HIP_TRY(hipMalloc((void**)&d_stats, kStatsWords * 8));
        VLG_HIP_TRY(hipMemsetAsync(d_stats, 0, kStatsWords * 8, st));
        PhaseTrace tr(st);
        const uint64_t fixed = 8ull << 20;
        if (ws->cap_bytes <= 2 * fixed) return fail(VLG_E_WORKSPACE, "workspace cap too small");
        const uint64_t budget = ws->cap_bytes - fixed;
        // arena: [copy of the lists][room for the survivors of the window filter][filter state][join scratch]
        const uint64_t pc_first = align_up(total, 64);
        uint64_t pc_cap = ws->filter && total ? std::min<uint64_t>(total, 0xFFFFFF00ull - pc_first) : 0;
        const uint64_t fence_entries = (pc_first + pc_cap) / 64 + 2;
        const uint64_t list_bytes = align_up((pc_first + pc_cap + 64) * 8, 256) + align_up((n_lists + 1) * 8, 256) + align_up(fence_entries * 8, 256) + 4096;
        if (list_bytes >= budget) return fail(VLG_E_WORKSPACE, "the lists do not fit the workspace cap");
        // the input is checked where it lies (ascending lists, largest position) BEFORE anything is planned: the bound on the
        // positions sizes the block bitmaps of the window filter, so the one plan made below is the one that is executed and the
        // arena is reserved for exactly that plan
        unsigned long long flags[2] = {0, 0};
        uint64_t* d_off_in = nullptr;
        svec<uint64_t> off_stage(h_list_off, h_list_off + n_lists + 1);
        if (total) {
            VLG_HIP_TRY(hipMalloc((void**)&d_off_in, (n_lists + 1) * 8));
            hipError_t e = hipMemcpyAsync(d_off_in, off_stage.data(), (n_lists + 1) * 8, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(lists_check_kernel<uint64_t>), dim3(grid_for((total + 7) / 8, 8192)), dim3(256), 0, st, d_lists, d_off_in, n_lists, total, d_stats + 4);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(flags, d_stats + 4, sizeof flags, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            (void)hipFree(d_off_in);
            VLG_HIP_TRY(e);
        }
        if (flags[0]) return fail(VLG_E_INVALID, "every list must be ascending");
        if (flags[1] > (1ull << 63)) return fail(VLG_E_INVALID, "positions above 2^63 are not supported");
        const uint64_t n_positions = flags[1] + 1;
        JoinPlan jp;
        if (vlg_status s = plan_joins(&qq, pl, ws, 0, n_joins, budget - list_bytes, n_positions, jp)) return s;
        if (vlg_status s = ws_reserve(ws, list_bytes + jp.filter_need + jp.want_bytes + jp.meta + fixed)) return s;
        ws->fences = nullptr;
        ws->rungs = nullptr;
        Arena A{ws->arena, ws->arena_bytes};
        uint64_t* P = A.take<uint64_t>(pc_first + pc_cap + 64);
        uint64_t* F = A.take<uint64_t>(fence_entries);
        if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (join lists)");
        if (total) {
            VLG_HIP_TRY(hipMemcpyAsync(P, d_lists, total * 8, hipMemcpyDeviceToDevice, st));
            if (total >= 64)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(fence_build_kernel<uint64_t>), dim3(grid_for(total / 64, 8192)), dim3(256), 0, st, P, (uint64_t)0, total / 64, F);
            VLG_HIP_TRY(hipGetLastError());
            ws->fences = F;
        }
        tr.mark("join lists copied + checked");
        if (vlg_status s = run_joins<uint64_t>(n_positions, &qq, ws, res, pl, poff, P, A, pc_cap ? P + pc_first : nullptr, pc_cap, 0, n_joins, jp,
                                               d_stats, tr)) return s;
        unsigned long long hs[kStatsWords];
        VLG_HIP_TRY(hipMemcpyAsync(hs, d_stats, sizeof hs, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        res->sum.checksum = 0;
        for (uint32_t i = 0; i < kChecksumSlots; ++i) res->sum.checksum += hs[kStatsChecksum + i];
        return VLG_OK;
    };
    vlg_status stt;
    {
        HostPoolScope staging(&ws->host);
        try { stt = run(); }
        catch (const std::bad_alloc&) { stt = fail(VLG_E_OOM, "out of host memory while planning the joins (pinned staging)"); }
    }
    if (d_stats) (void)hipFree(d_stats);
    if (stt) { vlg_result_destroy(res); return stt; }
    *out = res;
    return VLG_OK;
}

#include "wtsa.hpp"
#include "int_index.hpp"
