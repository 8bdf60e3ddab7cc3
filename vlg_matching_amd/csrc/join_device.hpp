// Device side of the join (K5): search helpers, feasibility bitset, link / jump / chain / gather kernels.
// Internal to search.hip's translation unit (everything has internal linkage); included exactly once from there.
#pragma once
// =============================================================================================
// Join kernels
// =============================================================================================
namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kChecksumSlots = 64;     // partial checksums of the gather kernel (added up by the host)

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

// ---- fences ------------------------------------------------------------------------------------------------------------------
// F[g] = P[64 g + 63] for every whole block of 64 elements of the arrays the joins search (fence_build_kernel: the sorted
// physical lists, and the survivors' private lists once they are compacted).  A search that cannot start from a previous answer
// first ranks its key among the fences of the list -- dense entries, ONE 256-byte load for 64 blocks, where probing the list
// itself touches 64 lines -- and then looks at one block of the list.  The fences with index in [a >> 6, b >> 6) are elements of
// the list P[a,b) whatever its borders, so no list needs fences of its own.  F == nullptr: the lists are probed directly.
template <typename pos_t>
__global__ void fence_build_kernel(const pos_t* __restrict__ P, uint64_t g0, uint64_t g1, pos_t* __restrict__ F)
{
    for (uint64_t g = g0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < g1; g += (uint64_t)gridDim.x * blockDim.x) F[g] = P[64 * g + 63];
}

// ---- rungs: a search ladder over the sorted lists -----------------------------------------------------------------------------
// R_j[i] = P[f^j (i + 1) - 1], j = 1 .. levels, f = kRungFan: the last element of every whole block of f^j elements.  A lower bound
// descends from the level whose one group of f entries covers the list: ONE aligned load of f entries per level picks one of f
// blocks, so a search touches log_f(length) lines instead of the log2(length) of a bisection -- the pivot filter, whose loads go to
// a different line in almost every lane, is bound by exactly those line requests (L1 misses on their way to L2), not by bytes.
// Entry i of level j lies inside the list P[a,b) iff a >> (s j) <= i < b >> (s j), s = log2 f, whatever the list's borders, so the
// lists share one ladder; entries before the list count as smaller than any key, entries behind it as larger.
#ifndef VLG_RUNG_SHIFT
#define VLG_RUNG_SHIFT 2
#endif
constexpr uint32_t kRungShift = VLG_RUNG_SHIFT, kRungFan = 1u << kRungShift;
constexpr uint32_t kMaxRungs = 32 / kRungShift;         // f^kMaxRungs elements: more than any index here (32-bit)
struct RungLayout { uint64_t off[kMaxRungs + 1]; uint32_t levels; uint64_t entries; };
inline RungLayout rung_layout(uint64_t total)
{
    RungLayout L{};
    L.levels = 0;
    while (L.levels < kMaxRungs && (total >> (kRungShift * (L.levels + 1))) != 0) ++L.levels;   // the block of level `levels + 1` covers every index
    uint64_t at = 0;
    for (uint32_t j = 1; j <= L.levels; ++j) {                                                  // whole groups, and one to spare
        L.off[j] = at;
        at += (((total >> (kRungShift * j)) + kRungFan - 1) & ~(uint64_t)(kRungFan - 1)) + kRungFan;
    }
    L.entries = at;
    return L;
}

// (F: the fences of the same array, written on the way when the ladder has a level of blocks of 64 -- it is that level -- or null)
template <typename pos_t>
__global__ void rung_build_kernel(const pos_t* __restrict__ P, uint64_t total, pos_t* __restrict__ R, const uint64_t* __restrict__ off, uint32_t levels,
                                  pos_t* __restrict__ F)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total >> kRungShift); i += (uint64_t)gridDim.x * blockDim.x) {
        const pos_t v = P[kRungFan * i + kRungFan - 1];
        R[off[1] + i] = v;
        uint64_t k = i;
        for (uint32_t j = 2; j <= levels && (k & (kRungFan - 1)) == kRungFan - 1; ++j) {
            k >>= kRungShift;
            R[off[j] + k] = v;
            if (F && kRungShift * j == 6) F[k] = v;
        }
    }
}
constexpr bool kRungsHoldFences = 6 % kRungShift == 0 && kRungShift < 6;       // some level has blocks of 64

template <typename pos_t> struct alignas(4 * sizeof(pos_t)) RungQuad { pos_t v[4]; };

// Keys the windows did not reach (the list is much denser than the keys): two-level search.  Lane t reads the LAST element of
// block t behind the fence `wb` -- one round trip covers 64 blocks -- every lane ranks its key among these 64 fences through
// cross-lane reads and then bisects the one block that holds its answer (6 probes inside 256 bytes).  A galloping search
// would pay ~2 log2(distance) dependent, scattered round trips instead.
constexpr uint32_t kFarRounds = 4;
template <typename pos_t>
__device__ __forceinline__ void wave_far_lower_bound(const pos_t* __restrict__ P, const pos_t* __restrict__ F, uint32_t wb, uint32_t b, pos_t key,
                                                     bool need, uint32_t& j, pos_t& val)
{
    const uint32_t lane = threadIdx.x & 63;
    constexpr pos_t kInf = (pos_t)~(pos_t)0;
    for (uint32_t round = 0; round < kFarRounds && wb < b; ++round) {
        if (!__any(need)) return;
        // without fences the blocks start at wb; with them they are the aligned blocks of the array, the first one cut at wb
        const uint32_t g0 = F ? wb >> 6 : 0;
        pos_t f;
        if (F) f = g0 + lane < (b >> 6) ? F[g0 + lane] : kInf;       // blocks that reach behind the list end with +inf
        else { const uint64_t fi = (uint64_t)wb + 64 * lane + 63; f = fi < b ? P[fi] : kInf; }
        const pos_t flast = __shfl(f, 63);
        const bool can = need && key <= flast;
        uint32_t lo = 0;                                             // first block whose last element is >= key
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) {
            const pos_t v = __shfl(f, (int)(lo + step - 1));
            lo += v < key ? step : 0u;
        }
        if (can) {
            uint64_t w0 = F ? ((uint64_t)(g0 + lo) << 6) : (uint64_t)wb + 64 * lo;   // the answer is in [w0, w0+63] (or b, in a clipped block)
            const uint64_t w1 = w0 + 64 < b ? w0 + 64 : b;
            if (w0 < wb) w0 = wb;
            uint32_t pos = 0;
#pragma unroll
            for (uint32_t step = 32; step; step >>= 1) {
                const uint64_t at = w0 + pos + step - 1;
                if (at < w1 && P[at] < key) pos += step;
            }
            const uint64_t at = w0 + pos;
            if (at < b) { j = (uint32_t)at; val = P[at]; } else j = b;
            need = false;
        }
        const uint64_t nwb = F ? ((uint64_t)(g0 + 64) << 6) : (uint64_t)wb + 4096;
        wb = nwb < b ? (uint32_t)nwb : b;
    }
    if (need) {
        j = gallop_lower_bound(P, wb < b ? wb : b, b, (uint64_t)key);
        if (j < b) val = P[j]; else j = b;
    }
}

// Lower bounds of 64 ascending keys in one sorted list, as a wave: the answers of a step lie just behind the last
// answer of the previous step, so the wave loads consecutive 64-element windows of the list with ONE coalesced load
// each and every lane ranks its key inside the window through cross-lane reads (6 steps) -- a merge of two sorted
// runs, without the ~15 scattered probes per lane of an independent search.  `wb` (wave-uniform) must be a fence:
// every element before it is smaller than every key.  Lanes still unresolved after kCoopWindows windows fall back
// to galloping from the last window's end.
constexpr uint32_t kCoopWindows = 4;
// Keys and list elements are compared as pos_t (one cross-lane read per step for 32-bit positions); `val` receives the list
// element at the answer, so the caller needs no load of its own for it.
template <typename pos_t>
__device__ __forceinline__ uint32_t wave_lower_bound(const pos_t* __restrict__ P, const pos_t* __restrict__ F, uint32_t wb, uint32_t b, pos_t key, bool need, pos_t& val)
{
    const uint32_t lane = threadIdx.x & 63;
    constexpr pos_t kInf = (pos_t)~(pos_t)0;
    uint32_t res = b;
    for (uint32_t it = 0; it < kCoopWindows; ++it) {
        if (!__any(need)) break;
        const uint32_t idx = wb + lane;
        const pos_t w = idx < b ? P[idx] : kInf;                     // +inf behind the list
        const pos_t wlast = __shfl(w, 63);
        const bool can = need && key <= wlast;
        uint32_t lo = 0;                                             // for `can` lanes w[63] >= key, so the answer is in [0,63]
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) {                 // lo = elements of w[0,63) below the key
            const pos_t v = __shfl(w, (int)(lo + step - 1));
            lo += v < key ? step : 0u;
        }
        const pos_t at = __shfl(w, (int)lo);
        if (can) { res = wb + lo; val = at; need = false; }
        wb += 64;
    }
    if (__any(need)) wave_far_lower_bound(P, F, wb < b ? wb : b, b, key, need, res, val);
    return res < b ? res : b;
}

// Two steps at once: 128 ascending keys (lane i holds keys i and 64+i) share every window load, its fence test and the
// loop around them, which is most of what a step costs.
#ifndef VLG_COOP_WINDOWS2
#define VLG_COOP_WINDOWS2 4
#endif
constexpr uint32_t kCoopWindows2 = VLG_COOP_WINDOWS2;      // round 3, link class on C3: 3 -> 22.2, 4 -> 21.9, 6 -> 22.5, 9 -> 22.8 ms (flat: the
                                                             // far search is about as cheap per key as a window); round 2:            // measured on C3: 2 -> 35.2, 3 -> 33.8, 4 -> 33.2, 6 -> 31.9, 8..16 -> 33..34 ms
template <typename pos_t>
__device__ __forceinline__ void wave_lower_bound2(const pos_t* __restrict__ P, const pos_t* __restrict__ F, uint32_t wb, uint32_t b, pos_t key0, bool need0, pos_t key1,
                                                  bool need1, uint32_t& j0, pos_t& v0, uint32_t& j1, pos_t& v1)
{
    const uint32_t lane = threadIdx.x & 63;
    constexpr pos_t kInf = (pos_t)~(pos_t)0;
    j0 = b; j1 = b;
    for (uint32_t it = 0; it < kCoopWindows2; ++it) {
        if (!__any(need0 || need1)) break;
        const uint32_t idx = wb + lane;
        const pos_t w = idx < b ? P[idx] : kInf;
        const pos_t wlast = __shfl(w, 63);
        const bool can0 = need0 && key0 <= wlast, can1 = need1 && key1 <= wlast;
        uint32_t lo0 = 0, lo1 = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) {                 // lo = elements of w[0,63) below the key (w[63] >= key for `can` lanes)
            const pos_t a0 = __shfl(w, (int)(lo0 + step - 1)), a1 = __shfl(w, (int)(lo1 + step - 1));
            lo0 += a0 < key0 ? step : 0u;
            lo1 += a1 < key1 ? step : 0u;
        }
        const pos_t at0 = __shfl(w, (int)lo0), at1 = __shfl(w, (int)lo1);
        if (can0) { j0 = wb + lo0; v0 = at0; need0 = false; }
        if (can1) { j1 = wb + lo1; v1 = at1; need1 = false; }
        wb += 64;
    }
    if (__any(need0 || need1)) {
        wb = wb < b ? wb : b;
        wave_far_lower_bound(P, F, wb, b, key0, need0, j0, v0);
        wave_far_lower_bound(P, F, wb, b, key1, need1, j1, v1);
    }
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
// The same through the fences: the key is ranked among the list's fences (the same 64-ary search, over F), then inside one block.
template <typename pos_t>
__device__ __forceinline__ uint32_t wave_kary_lower_bound(const pos_t* __restrict__ P, const pos_t* __restrict__ F, uint32_t a, uint32_t b, uint64_t key)
{
    if (!F) return wave_kary_lower_bound(P, a, b, key);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t g = wave_kary_lower_bound(F, a >> 6, b >> 6, key);            // first block of the list whose last element is >= key
    uint64_t w0 = (uint64_t)g << 6;
    const uint64_t w1 = w0 + 64 < b ? w0 + 64 : b;                               // (behind the last whole block: the rest of the list)
    if (w0 < a) w0 = a;
    const uint64_t idx = w0 + lane;
    const bool in = idx < w1;
    const uint64_t v = in ? (uint64_t)P[idx] : 0;
    return (uint32_t)w0 + (uint32_t)__popcll(__ballot(in && v < key));
}

// Lower bounds in P[.,pend) of the keys flagged in `need` (bit i = key[i]).  `wb` is a wave-uniform fence: every element
// before it is smaller than every flagged key of the wave.  j[i] = the lower bound (pend if there is none), v[i] = P[j[i]].
// The keys of the wave need not be ordered; tiles that cannot hold an answer are skipped with one probe.
template <typename pos_t>
__device__ __forceinline__ void tile_lower_bounds(const pos_t* __restrict__ P, const pos_t* __restrict__ F, uint32_t wb, const uint32_t pend, pos_t* __restrict__ tile,
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
        if ((uint64_t)P[probe_at] < kmin) wb = wave_kary_lower_bound(P, F, (uint32_t)probe_at + 1, pend, kmin);
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
#ifndef VLG_LINK_RUN
#define VLG_LINK_RUN 2048
#endif
constexpr uint32_t kLinkRun = VLG_LINK_RUN;      // slots a wave of the link pass takes (round 4, C3: 2048 -> 21.3 ms, 4096 -> 21.9 ms: what stands in front of
                                                 // a run -- segment, metadata, the first fence -- is not what the pass waits for; it is issue-bound)

__device__ __forceinline__ uint32_t seg_find(const uint32_t* __restrict__ seg_begin, uint32_t nseg, uint64_t slot)
{
    uint32_t lo = 0, hi = nseg;                        // last p with seg_begin[p] <= slot
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (seg_begin[mid] <= slot) lo = mid; else hi = mid; }
    return lo;
}
// the same for a whole wave asking about one slot (all 64 lanes call it): three rounds of 64 probes instead of a chain of seventeen
// dependent loads in front of the wave's run
__device__ __forceinline__ uint32_t wave_seg_find(const uint32_t* __restrict__ seg_begin, uint32_t nseg, uint64_t slot)
{
    const uint32_t lb = wave_kary_lower_bound<uint32_t>(seg_begin, 0, nseg, slot + 1);       // first segment that begins behind the slot
    return uniform(lb ? lb - 1 : 0u);
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

// what a slot does with its lower bound j (value v) in the next list: the first FEASIBLE element at or after it, if still
// inside the window, becomes its link
template <typename pos_t, bool kLast>
__device__ __forceinline__ bool link_finish(const pos_t* __restrict__ P, const SegMeta& nx, const FeasRef& fb, uint32_t e, bool want, uint32_t j,
                                            pos_t v, pos_t thi, pos_t* __restrict__ endp, uint32_t* __restrict__ link)
{
    bool ok = false;
    if (want && j < nx.pend) {
        if (kLast) {
            ok = v <= thi;
            if (ok) { if (link) link[e] = j; endp[e] = v; }
        } else {
            const uint32_t at = nx.begin + (j - nx.pbegin);
            const uint32_t ej = next_feasible(fb, at);     // nearest feasible logical element at or after it
            if (ej < nx.end) {
                const pos_t pv = ej == at ? v : P[phys_of(nx, ej)];
                if (pv <= thi) { ok = true; if (link) link[e] = ej; endp[e] = endp[ej]; }
            }
        }
    }
    return ok;
}

// (link == nullptr: only the chain ends are wanted -- first positions without tuples)
// link pass over the class [r0,r1) of slots that have `dist` sub-patterns after them (kLast: dist == 1, the next list is the
// query's last one: all its elements are feasible and it has no join state).
// Steps that lie inside one segment (almost all of them: lists are long) keep the segment's metadata in registers,
// search as a wave behind the previous step's answer and have the next step's positions already in flight.
// Slot numbers are 32-bit inside a chunk (r1 <= 0xF0000000 + alignment), so all loop arithmetic is.
template <typename pos_t, bool kLast>
__global__ void __launch_bounds__(256) join_link_kernel(const pos_t* __restrict__ P, const pos_t* __restrict__ F /* fences of P, or null */,
                                                        const uint32_t* __restrict__ seg_begin, uint32_t nseg,
                                                        const SegMeta* __restrict__ sm, uint32_t r0, uint32_t r1,
                                                        FeasRef fb, uint64_t* __restrict__ fbits_out,
                                                        pos_t* __restrict__ endp, uint32_t* __restrict__ link)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave * kLinkRun >= (uint64_t)(r1 - r0)) return;
    const uint32_t run_begin = r0 + (uint32_t)wave * kLinkRun;
    const uint32_t run_end = r1 - run_begin > kLinkRun ? run_begin + kLinkRun : r1;
    uint32_t s_w = wave_seg_find(seg_begin, nseg, run_begin);     // wave-uniform: segment of `base`
    uint32_t seg_end = seg_begin[s_w + 1];
    SegMeta m = sm[s_w], nx = sm[m.next];
    uint32_t hint_seg = kNone, hint = 0;                           // answer of the last lane of the previous step and its segment
    pos_t x_pre = 0, x_pre1 = 0;
    bool have_pre = false, have_pre1 = false;
    for (uint32_t base = run_begin; base < run_end; base += 64) {
        if (base >= seg_end) {                                     // entered a new segment (skips empty ones)
            while (seg_begin[s_w + 1] <= base) ++s_w;
            seg_end = seg_begin[s_w + 1];
            m = sm[s_w]; nx = sm[m.next];
            have_pre = false; have_pre1 = false;
        }
        if (run_end - base >= 128 && base + 127 < seg_end) {
            // ---- two full steps inside one segment: 128 keys per window ---------------------------------
            const uint32_t e0 = base + lane, e1 = e0 + 64;
            const pos_t x0 = have_pre ? x_pre : P[phys_of(m, e0)];
            const pos_t x1 = have_pre1 ? x_pre1 : P[phys_of(m, e1)];
            const uint32_t left = run_end - base - 128;            // slots of the run behind this pair of steps
            const uint32_t room = seg_end - base - 128;            // and of the segment
            have_pre = left > 0 && (left < 64 ? left : 64u) <= room;
            have_pre1 = left >= 128 && room >= 128;
            if (have_pre) x_pre = e0 + 128 < run_end ? P[phys_of(m, e0 + 128)] : (pos_t)0;
            if (have_pre1) x_pre1 = P[phys_of(m, e1 + 128)];
            pos_t tlo0, thi0, tlo1, thi1, v0 = 0, v1 = 0;
            const bool want0 = gap_window<pos_t>((uint64_t)x0, nx.lo, nx.hi, tlo0, thi0);
            const bool want1 = gap_window<pos_t>((uint64_t)x1, nx.lo, nx.hi, tlo1, thi1);
            uint32_t j0 = nx.pend, j1 = nx.pend;
            if (hint_seg != s_w)                                   // first step of the run in this segment: the smallest key's bound is the fence
                hint = wave_kary_lower_bound(P, F, nx.pbegin, nx.pend, (uint64_t)__shfl(tlo0, 0));
            wave_lower_bound2(P, F, hint, nx.pend, tlo0, want0, tlo1, want1, j0, v0, j1, v1);
            if (!want0) j0 = nx.pend;
            if (!want1) j1 = nx.pend;
            const bool ok0 = link_finish<pos_t, kLast>(P, nx, fb, e0, want0, j0, v0, thi0, endp, link);
            const bool ok1 = link_finish<pos_t, kLast>(P, nx, fb, e1, want1, j1, v1, thi1, endp, link);
            const unsigned long long okm0 = __ballot(ok0), okm1 = __ballot(ok1);
            if (lane == 0) { fbits_out[base >> 6] = okm0; fbits_out[(base >> 6) + 1] = okm1; }
            hint_seg = s_w;
            hint = __shfl(j1, 63);
            base += 64;                                            // the loop adds the other 64
            continue;
        }
        have_pre1 = false;
        const uint32_t e = base + lane;
        const bool active = e < run_end;
        const uint32_t step_last = run_end - base > 64 ? base + 63 : run_end - 1;
        uint32_t j = 0, s_last = s_w;
        if (step_last < seg_end) {
            // ---- fast path: one segment ---------------------------------------------------------------
            const pos_t x = have_pre ? x_pre : (active ? P[phys_of(m, e)] : (pos_t)0);
            const uint32_t en = e + 64;                            // next step's position, in flight during the search
            have_pre = run_end - base > 64 && (run_end - base > 128 ? base + 127 : run_end - 1) < seg_end;
            if (have_pre) x_pre = en < run_end ? P[phys_of(m, en)] : (pos_t)0;
            pos_t tlo, thi, v = 0;
            const bool want = gap_window<pos_t>((uint64_t)x, nx.lo, nx.hi, tlo, thi) && active;    // false: no position can be in the window
            j = nx.pend;
            if (hint_seg != s_w) hint = wave_kary_lower_bound(P, F, nx.pbegin, nx.pend, (uint64_t)__shfl(tlo, 0));
            j = wave_lower_bound(P, F, hint, nx.pend, tlo, want, v);
            if (!want) j = nx.pend;
            const bool ok = link_finish<pos_t, kLast>(P, nx, fb, e, want, j, v, thi, endp, link);
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
                const uint64_t x = P[phys_of(ml, e)];
                const uint64_t tlo = sat_add(x, nl.lo), thi = sat_add(x, nl.hi);
                j = gallop_lower_bound(P, (s == hint_seg) ? hint : nl.pbegin, nl.pend, tlo);
                if (e < ml.end) {                                  // padding slots between classes belong to no segment
                    if (kLast) {
                        ok = j < nl.pend && (uint64_t)P[j] <= thi;
                        if (ok) { if (link) link[e] = j; endp[e] = P[j]; }
                    } else if (j < nl.pend) {
                        const uint32_t ej = next_feasible(fb, (uint64_t)nl.begin + (j - nl.pbegin));
                        if (ej < nl.end && (uint64_t)P[phys_of(nl, ej)] <= thi) { ok = true; if (link) link[e] = ej; endp[e] = endp[ej]; }
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
__global__ void __launch_bounds__(256) join_jump_kernel(const pos_t* __restrict__ P, const pos_t* __restrict__ F, const uint32_t* __restrict__ seg_begin, uint32_t nseg,
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
    uint32_t s = wave_seg_find(seg_begin, nseg, run_begin);
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
                tile_lower_bounds<pos_t>(P, F, fence, m.pend, tile, key, need, j, v);
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

// A wave per record: the matches inside its tile are entry, jump(entry), jump^2(entry), ..., `cnt` of them (the hop count the tile pass
// left in xh).  The tile's jump targets are staged in LDS as tile-relative 16-bit offsets and the wave builds jump^(2^k) by doubling
// there; lane r composes the tables selected by the bits of its rank r, so every match is found by <= 10 independent LDS lookups
// instead of one lane hopping through the whole chain (380 dependent hops per tile on BASELINE config 3).
__global__ void __launch_bounds__(256) chain_emit_kernel(const uint32_t* __restrict__ rec_begin, const uint32_t* __restrict__ rec_count,
                                                         const uint2* __restrict__ records, uint32_t total_rec_slots,
                                                         const uint32_t* __restrict__ rec_query, const uint32_t* __restrict__ jump,
                                                         const uint2* __restrict__ xh, uint64_t r1 /* end of the slots that have a jump */,
                                                         uint32_t* __restrict__ mlist)
{
    constexpr uint16_t kOut = 0xFFFFu;                               // the chain leaves the tile
    constexpr uint32_t kPer = kTile / 64;                            // slots (and ranks) per lane
    __shared__ uint16_t s_t[4][2][kTile];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t r = uniform((uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (r >= total_rec_slots) return;
    const uint32_t q = rec_query[r];                         // slot r belongs to query q; used only if r - rec_begin[q] < rec_count[q]
    if (r - rec_begin[q] >= rec_count[q]) return;
    const uint2 rc = records[r];
    const uint64_t tile_base = (uint64_t)(rc.x / kTile) * kTile;
    const uint32_t cnt = uniform(xh[rc.x].y);                        // chain elements inside the tile, entry included (1 .. kTile)
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint64_t e = tile_base + lane + 64 * i;
        const uint32_t j = e < r1 ? jump[e] : kNone;
        s_t[wv][0][lane + 64 * i] = (j != kNone && (uint64_t)j < tile_base + kTile) ? (uint16_t)(j - (uint32_t)tile_base) : kOut;
    }
    wave_sync();
    uint32_t cur[kPer];
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) cur[i] = rc.x - (uint32_t)tile_base;
    uint32_t buf = 0;
    for (uint32_t k = 0; (1u << k) < cnt; ++k) {
        const uint16_t* __restrict__ T = s_t[wv][buf];
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint32_t rank = lane + 64 * i;
            if (rank < cnt && ((rank >> k) & 1)) cur[i] = T[cur[i]];         // (rank < cnt: the walk stays inside the tile)
        }
        if ((2u << k) < cnt) {                                               // jump^(2^(k+1)) for the next round
            uint16_t* __restrict__ N = s_t[wv][buf ^ 1];
#pragma unroll
            for (uint32_t i = 0; i < kPer; ++i) {
                const uint32_t sl = lane + 64 * i;
                const uint16_t a = T[sl];
                N[sl] = a == kOut ? kOut : T[a];
            }
            wave_sync();
            buf ^= 1;
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t rank = lane + 64 * i;
        if (rank < cnt) mlist[rc.y + rank] = (uint32_t)tile_base + cur[i];
    }
}

// tuples of every match: walk the links from the level-0 element.  One thread per MATCH of the chunk (match g belongs to the query q with
// out_first[q] <= g < out_first[q + 1]; it is that query's match number g - out_first[q]) -- the kernel used to run over every slot of
// every first list and let three quarters of its lanes find out that their slot held no match.
// Results are written as wide as the positions are (ResultPiece::width); vlg_result_fetch widens them on the way to the host.
template <typename pos_t>
__global__ void __launch_bounds__(256) join_gather_kernel(const pos_t* __restrict__ P, const SegMeta* __restrict__ sm, const QueryMeta* __restrict__ qm,
                                                          const unsigned long long* __restrict__ first_of /* [nq] = qm[.].out_first, dense */, uint32_t nq,
                                                          uint64_t n_matches, const uint32_t* __restrict__ link, const uint32_t* __restrict__ mlist,
                                                          pos_t* __restrict__ out_first, pos_t* __restrict__ out_tuples, unsigned long long* __restrict__ checksum)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t run_begin = wave * kRun;
    const bool live = run_begin < n_matches;               // (no early return: every wave of the workgroup reaches the barriers below)
    const uint64_t run_end = !live ? run_begin : (run_begin + kRun < n_matches ? run_begin + kRun : n_matches);
    // the query of the run's first match: the last one that starts at or before it (queries without matches start where their successor starts)
    uint32_t q_w = 0, q_end = 0;                           // ... and the query of its last match: a lane looks its own query up between the two
    if (live) {                                            // (`live` is wave-uniform)
        const uint32_t ub = wave_kary_lower_bound<unsigned long long>(first_of, 0, nq, run_begin + 1);
        q_w = uniform(ub ? ub - 1 : 0u);
        const uint32_t ue = wave_kary_lower_bound<unsigned long long>(first_of, 0, nq, run_end);
        q_end = uniform(ue ? ue - 1 : 0u);
    }
    unsigned long long local = 0;
    for (uint64_t base = run_begin; base < run_end; base += 64) {
        const uint64_t g = base + lane;
        uint32_t q = q_w;
        if (g < run_end) {
            uint32_t hi = q_end;                                       // last query that starts at or before match g (a batch with few matches has
            while (q < hi) {                                           // thousands of queries between two of them: no walk from query to query)
                const uint32_t mid = q + ((hi - q + 1) >> 1);
                if (first_of[mid] <= g) q = mid; else hi = mid - 1;
            }
            const QueryMeta Q = qm[q];
            const SegMeta m = sm[Q.seg0];
            const uint64_t t = g - Q.out_first;
            const uint32_t el = mlist[m.begin + t];                    // logical element of level 0
            const pos_t first = P[phys_of(m, el)];
            out_first[g] = first;
            local += first;
            if (out_tuples) {                                          // null: first positions only (workspace option "tuples" = 0)
                pos_t* tp = out_tuples + Q.out_tuple + t * Q.k;
                tp[0] = first;
                uint32_t cur = Q.k > 1 ? link[el] : 0;
                uint32_t sg = m.next;
                for (uint32_t i = 1; i < Q.k; ++i) {
                    const SegMeta mi = sm[sg];
                    if (mi.dist == 0) { tp[i] = P[cur]; }              // link of a dist-1 element is a physical index
                    else { tp[i] = P[phys_of(mi, cur)]; cur = link[cur]; sg = mi.next; }
                }
            }
        }
        const uint32_t last = (uint32_t)((run_end - base < 64 ? run_end - base : 64) - 1);
        q_w = __shfl(q, (int)last);                                    // the next step's matches come behind this step's last one
    }
    // one atomic per workgroup, spread over kChecksumSlots words (a single word takes ~90 atomics per microsecond; there are
    // 10^5..10^6 waves here); the host adds the slots up
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if (lane == 0 && local) atomicAdd(&s_sum, local);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum) atomicAdd(checksum + (blockIdx.x % kChecksumSlots), s_sum);
}


}  // namespace
