// X1: RCCL on the product side (SURVEY.md 8b "broadcast(handle, ncclComm)", 8e): one process per GPU, the read-only index image
// broadcast over xGMI at load time, counters reduced, sorted occurrence lists exchanged (vlg_comm_allgatherv).
//
// RCCL is bound at run time, not at link time: a process that already holds an RCCL (PyTorch ships its own librccl.so, a C++ host
// may link /opt/rocm's) must have ITS library called with ITS communicator, and two RCCLs with the same symbol names must not be
// mixed.  So the entry points are looked up first among the libraries the process has loaded (dlsym on the global scope) and only
// then in librccl.so.1 / librccl.so.  ncclComm_t crosses the C-ABI as void*.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "common.hpp"
#include "rccl_decl.hpp"

using namespace vlg;

namespace {

// (the few declarations of rccl.h this file needs live in rccl_decl.hpp; rccl_abi_check.cpp asserts them against the real header)
using ncclComm_t = vlg_rccl::comm_t;
using ncclUniqueId = vlg_rccl::UniqueId;
constexpr int ncclSuccess = vlg_rccl::kSuccess, ncclUint8 = vlg_rccl::kUint8, ncclUint64 = vlg_rccl::kUint64, ncclSum = vlg_rccl::kSum;

struct Rccl {
    bool ok = false;
    std::string err, path;                                   // path: the library the entry points came from
    vlg_rccl::GetUniqueId_fn GetUniqueId = nullptr;
    vlg_rccl::CommInitRank_fn CommInitRank = nullptr;
    vlg_rccl::CommDestroy_fn CommDestroy = nullptr;
    vlg_rccl::CommCount_fn CommCount = nullptr;
    vlg_rccl::CommUserRank_fn CommUserRank = nullptr;
    vlg_rccl::Broadcast_fn Broadcast = nullptr;
    vlg_rccl::AllReduce_fn AllReduce = nullptr;
    vlg_rccl::AllGather_fn AllGather = nullptr;
    vlg_rccl::Send_fn Send = nullptr;
    vlg_rccl::Recv_fn Recv = nullptr;
    vlg_rccl::Group_fn GroupStart = nullptr;
    vlg_rccl::Group_fn GroupEnd = nullptr;
    vlg_rccl::GetErrorString_fn GetErrorString = nullptr;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;                                   // nullptr: RTLD_DEFAULT, what the process already exports globally
        const char* forced = getenv("VLG_RCCL_LIBRARY");     // an explicit choice wins over every probe, and nothing else is tried
        if (forced && *forced) {
            h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!h) { const char* e = dlerror(); r.err = std::string("RCCL not found (VLG_RCCL_LIBRARY=") + forced + "): " + (e ? e : ""); return; }
        } else if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
            // an RCCL mapped into the process without global symbols (PyTorch's own librccl.so is): take a handle to THAT copy
            std::string mapped;
            if (FILE* f = fopen("/proc/self/maps", "r")) {
                char line[4096];
                while (mapped.empty() && fgets(line, sizeof line, f)) {
                    const char* at = strstr(line, "librccl.so");
                    const char* path = at ? strchr(line, '/') : nullptr;
                    if (path && path < at) { mapped = path; while (!mapped.empty() && (mapped.back() == '\n' || mapped.back() == ' ')) mapped.pop_back(); }
                }
                fclose(f);
            }
            if (!mapped.empty()) h = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            std::string last_error;
            for (const char* n : names) {
                if (h) break;
                h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (!h) { const char* e = dlerror(); if (e) last_error = e; }      // (dlerror() clears the message: read it once)
            }
            if (!h) { r.err = "RCCL not found (librccl.so.1): " + last_error; return; }
        }
        if (h) {
            Dl_info di;
            void* any = dlsym(h, "ncclGetUniqueId");
            if (any && dladdr(any, &di) && di.dli_fname) r.path = di.dli_fname;
        } else {
            Dl_info di;
            if (dladdr(dlsym(RTLD_DEFAULT, "ncclCommInitRank"), &di) && di.dli_fname) r.path = di.dli_fname;
        }
        auto sym = [&](const char* n) -> void* { return h ? dlsym(h, n) : dlsym(RTLD_DEFAULT, n); };
#define VLG_BIND(field, name) r.field = reinterpret_cast<decltype(r.field)>(sym(name)); if (!r.field) { r.err = std::string("RCCL symbol missing: ") + name; return; }
        VLG_BIND(GetUniqueId, "ncclGetUniqueId")
        VLG_BIND(CommInitRank, "ncclCommInitRank")
        VLG_BIND(CommDestroy, "ncclCommDestroy")
        VLG_BIND(CommCount, "ncclCommCount")
        VLG_BIND(CommUserRank, "ncclCommUserRank")
        VLG_BIND(Broadcast, "ncclBroadcast")
        VLG_BIND(AllReduce, "ncclAllReduce")
        VLG_BIND(AllGather, "ncclAllGather")
        VLG_BIND(Send, "ncclSend")
        VLG_BIND(Recv, "ncclRecv")
        VLG_BIND(GroupStart, "ncclGroupStart")
        VLG_BIND(GroupEnd, "ncclGroupEnd")
        VLG_BIND(GetErrorString, "ncclGetErrorString")
#undef VLG_BIND
        r.ok = true;
    });
    return r;
}

vlg_status need_rccl()
{
    if (!rccl().ok) return fail(VLG_E_UNSUPPORTED, rccl().err);
    return VLG_OK;
}

#define VLG_NCCL_TRY(expr)                                                                                       \
    do {                                                                                                         \
        const int _r = (expr);                                                                                   \
        if (_r != ncclSuccess) return fail(VLG_E_INTERNAL, std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
    } while (0)

vlg_status comm_shape(void* comm, int& n, int& rank)
{
    if (!comm) return fail(VLG_E_INVALID, "null communicator");
    if (vlg_status s = need_rccl()) return s;
    VLG_NCCL_TRY(rccl().CommCount((ncclComm_t)comm, &n));
    VLG_NCCL_TRY(rccl().CommUserRank((ncclComm_t)comm, &rank));
    return VLG_OK;
}

}  // namespace

extern "C" const char* vlg_comm_library(void)
{
    return rccl().ok ? rccl().path.c_str() : "";
}

extern "C" vlg_status vlg_comm_unique_id(vlg_comm_id* out)
{
    static_assert(sizeof(vlg_comm_id) == sizeof(ncclUniqueId), "vlg_comm_id carries an ncclUniqueId");
    if (!out) return fail(VLG_E_INVALID, "null argument");
    if (vlg_status s = need_rccl()) return s;
    ncclUniqueId id;
    VLG_NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(out->bytes, id.internal, sizeof id.internal);
    return VLG_OK;
}

extern "C" vlg_status vlg_comm_create(const vlg_comm_id* id, int n_ranks, int rank, void** comm)
{
    if (!id || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(VLG_E_INVALID, "bad communicator arguments");
    *comm = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available");
    if (vlg_status s = need_rccl()) return s;
    ncclUniqueId u;
    memcpy(u.internal, id->bytes, sizeof u.internal);
    ncclComm_t c = nullptr;
    VLG_NCCL_TRY(rccl().CommInitRank(&c, n_ranks, u, rank));       // binds to the current HIP device of the calling thread
    *comm = c;
    return VLG_OK;
}

extern "C" void vlg_comm_destroy(void* comm)
{
    if (comm && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)comm);
}

extern "C" vlg_status vlg_comm_info(void* comm, int* n_ranks, int* rank)
{
    int n = 0, r = 0;
    if (vlg_status s = comm_shape(comm, n, r)) return s;
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    return VLG_OK;
}

// The index image is ONE contiguous allocation (common.hpp), so the broadcast needs no staging copy on the root: its size goes
// first (8 bytes), then the blob itself straight out of / into the HBM the index lives in.
extern "C" vlg_status vlg_index_broadcast(const vlg_index* idx_or_null, void* nccl_comm, int root, void* stream, vlg_index** out)
{
    if (out) *out = nullptr;
    int n = 0, rank = 0;
    if (vlg_status s = comm_shape(nccl_comm, n, rank)) return s;
    if (root < 0 || root >= n) return fail(VLG_E_INVALID, "no such root rank");
    if (rank == root && !idx_or_null) return fail(VLG_E_INVALID, "the root rank must pass its index");
    if (rank != root && !out) return fail(VLG_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    unsigned long long* d_bytes = nullptr;
    void* d_blob = nullptr;
    vlg_index* idx = nullptr;
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipMalloc((void**)&d_bytes, 8));
        unsigned long long bytes = rank == root ? idx_or_null->hdr.total_bytes : 0;
        VLG_HIP_TRY(hipMemcpyAsync(d_bytes, &bytes, 8, hipMemcpyHostToDevice, st));
        VLG_NCCL_TRY(rccl().Broadcast(d_bytes, d_bytes, 8, ncclUint8, root, comm, st));
        VLG_HIP_TRY(hipMemcpyAsync(&bytes, d_bytes, 8, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        if (bytes < sizeof(BlobHeader)) return fail(VLG_E_INTERNAL, "broadcast of the index size failed");
        if (rank == root) {
            VLG_NCCL_TRY(rccl().Broadcast(idx_or_null->d_blob, idx_or_null->d_blob, bytes, ncclUint8, root, comm, st));
            VLG_HIP_TRY(hipStreamSynchronize(st));
            return VLG_OK;
        }
        VLG_HIP_TRY(hipMalloc(&d_blob, bytes));
        VLG_NCCL_TRY(rccl().Broadcast(d_blob, d_blob, bytes, ncclUint8, root, comm, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        if (vlg_status s = vlg_index_attach_blob(d_blob, bytes, &idx)) return s;
        idx->owns_blob = true;
        d_blob = nullptr;
        return VLG_OK;
    };
    const vlg_status s = run();
    if (d_bytes) (void)hipFree(d_bytes);
    if (d_blob) (void)hipFree(d_blob);
    if (s) { if (idx) vlg_index_destroy(idx); return s; }
    if (out) *out = rank == root ? const_cast<vlg_index*>(idx_or_null) : idx;
    return VLG_OK;
}

// counters of a sharded batch (num_results, checksum modulo 2^64 like gm_search.cpp:110-114, located occurrences): summed over
// the ranks in place; unsigned 64-bit sums wrap, which is the arithmetic the reference's checksum has
extern "C" vlg_status vlg_comm_allreduce_sum_u64(void* nccl_comm, uint64_t* h_vals, uint32_t count, void* stream)
{
    int n = 0, rank = 0;
    if (vlg_status s = comm_shape(nccl_comm, n, rank)) return s;
    if (!count) return VLG_OK;
    if (!h_vals) return fail(VLG_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    uint64_t* d = nullptr;
    VLG_HIP_TRY(hipMalloc((void**)&d, count * 8ull));
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipMemcpyAsync(d, h_vals, count * 8ull, hipMemcpyHostToDevice, st));
        VLG_NCCL_TRY(rccl().AllReduce(d, d, count, ncclUint64, ncclSum, (ncclComm_t)nccl_comm, st));
        VLG_HIP_TRY(hipMemcpyAsync(h_vals, d, count * 8ull, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        return VLG_OK;
    };
    const vlg_status s = run();
    (void)hipFree(d);
    return s;
}

// Variable-size all-gather of device buffers: rank r contributes h_counts[r] elements of elem_bytes at d_send; afterwards every
// rank holds all contributions in rank order at d_recv (sum of the counts).  Built from one broadcast per rank inside a group, so
// that every contribution travels once per peer whatever the sizes are (ncclAllGather wants equal sizes).
extern "C" vlg_status vlg_comm_allgatherv(void* nccl_comm, const void* d_send, const uint64_t* h_counts, uint32_t elem_bytes, void* d_recv,
                                          void* stream)
{
    int n = 0, rank = 0;
    if (vlg_status s = comm_shape(nccl_comm, n, rank)) return s;
    if (!h_counts || !elem_bytes) return fail(VLG_E_INVALID, "null argument");
    uint64_t total = 0;
    for (int r = 0; r < n; ++r) total += h_counts[r];
    if (total && !d_recv) return fail(VLG_E_INVALID, "null argument");
    if (h_counts[rank] && !d_send) return fail(VLG_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    VLG_NCCL_TRY(rccl().GroupStart());
    uint64_t off = 0;
    int rc = ncclSuccess;
    for (int r = 0; r < n && rc == ncclSuccess; ++r) {
        uint8_t* dst = (uint8_t*)d_recv + off * elem_bytes;
        if (h_counts[r]) rc = rccl().Broadcast(r == rank ? d_send : (const void*)dst, dst, h_counts[r] * (uint64_t)elem_bytes, ncclUint8, r, (ncclComm_t)nccl_comm, st);
        off += h_counts[r];
    }
    const int rc2 = rccl().GroupEnd();
    VLG_NCCL_TRY(rc);
    VLG_NCCL_TRY(rc2);
    return VLG_OK;
}

// All-to-all-v of device buffers over the xGMI mesh: this rank sends h_send_counts[r] elements to rank r -- packed one after the other
// in rank order at d_send -- and receives h_recv_counts[r] elements from rank r, packed likewise at d_recv; its own piece is a
// device copy.  One grouped ncclSend / ncclRecv pair per peer: every pair of GPUs has its own xGMI link, so the N - 1 transfers of a
// rank run side by side at link rate instead of queueing behind each other on a ring (what a group of broadcasts does).
extern "C" vlg_status vlg_comm_alltoallv(void* nccl_comm, const void* d_send, const uint64_t* h_send_counts, void* d_recv,
                                         const uint64_t* h_recv_counts, uint32_t elem_bytes, void* stream)
{
    int n = 0, rank = 0;
    if (vlg_status s = comm_shape(nccl_comm, n, rank)) return s;
    if (!h_send_counts || !h_recv_counts || !elem_bytes) return fail(VLG_E_INVALID, "null argument");
    if (h_send_counts[rank] != h_recv_counts[rank]) return fail(VLG_E_INVALID, "a rank sends itself as much as it receives from itself");
    uint64_t ts = 0, tr = 0;
    for (int r = 0; r < n; ++r) { ts += h_send_counts[r]; tr += h_recv_counts[r]; }
    if ((ts && !d_send) || (tr && !d_recv)) return fail(VLG_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    VLG_NCCL_TRY(rccl().GroupStart());
    uint64_t so = 0, ro = 0;
    int rc = ncclSuccess;
    hipError_t he = hipSuccess;
    for (int r = 0; r < n && rc == ncclSuccess && he == hipSuccess; ++r) {
        const uint8_t* sp = (const uint8_t*)d_send + so * elem_bytes;
        uint8_t* rp = (uint8_t*)d_recv + ro * elem_bytes;
        if (r == rank) {
            if (h_send_counts[r]) he = hipMemcpyAsync(rp, sp, h_send_counts[r] * (uint64_t)elem_bytes, hipMemcpyDeviceToDevice, st);
        } else {
            if (h_send_counts[r]) rc = rccl().Send(sp, h_send_counts[r] * (uint64_t)elem_bytes, ncclUint8, r, (ncclComm_t)nccl_comm, st);
            if (rc == ncclSuccess && h_recv_counts[r]) rc = rccl().Recv(rp, h_recv_counts[r] * (uint64_t)elem_bytes, ncclUint8, r, (ncclComm_t)nccl_comm, st);
        }
        so += h_send_counts[r];
        ro += h_recv_counts[r];
    }
    const int rc2 = rccl().GroupEnd();
    VLG_NCCL_TRY(rc);
    VLG_NCCL_TRY(rc2);
    VLG_HIP_TRY(he);
    return VLG_OK;
}
