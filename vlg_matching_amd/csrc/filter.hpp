// Window filter: structures, kernels and the host driver of one filter group.
// Internal to search.hip's translation unit (everything has internal linkage); included exactly once from there,
// behind Arena / Plan / Timed / vlg_workspace.
#pragma once
namespace {

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
#ifndef VLG_PIVOT_GROUPS
#define VLG_PIVOT_GROUPS 2
#endif
constexpr uint32_t kPivotGroups = VLG_PIVOT_GROUPS;
template <typename pos_t>
__device__ __forceinline__ void pivot_ranges(const pos_t* __restrict__ P, const pos_t* __restrict__ F, uint32_t pbegin, uint32_t pend, const uint64_t (&a)[kPivotGroups],
                                             const uint64_t (&b)[kPivotGroups], bool (&on)[kPivotGroups], uint32_t (&i0)[kPivotGroups],
                                             uint32_t (&i1)[kPivotGroups])
{
    constexpr uint32_t G = kPivotGroups;
    const uint32_t lane = threadIdx.x & 63;
    // ---- brackets: searches 0..G-1 for min a, G..2G-1 for max b + 1, over the whole list ------------------------
    // With fences (join_device.hpp) the 64-ary rounds run over F -- fences [pbegin >> 6, pend >> 6) are elements of this list --
    // and end in one block of the list; without them the rounds probe the list itself.
    uint64_t key[2 * G];
    uint32_t A[2 * G], B[2 * G];
    bool any[G];
    const pos_t* __restrict__ S = F ? F : P;                 // what the rounds probe
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        const unsigned long long m = __ballot(on[g]);
        any[g] = m != 0;
        const int first = m ? __ffsll((long long)m) - 1 : 0, last = m ? 63 - __clzll((long long)m) : 0;
        key[g] = uniform(__shfl(a[g], first));
        const uint64_t bmax = uniform(__shfl(b[g], last));
        key[G + g] = bmax == ~0ull ? ~0ull : bmax + 1;
        A[g] = A[G + g] = F ? pbegin >> 6 : pbegin;
        B[g] = B[G + g] = any[g] ? (F ? pend >> 6 : pend) : A[g];           // nothing to search for an empty group
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
            v[s] = in[s] ? (uint64_t)S[idx] : 0;
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
    if (F) {
        // rank among the last <= 64 fences, then the block behind the fences that are smaller: [A, B) becomes that block of the list
        uint64_t v[2 * G];
        bool in[2 * G];
#pragma unroll
        for (uint32_t s = 0; s < 2 * G; ++s) { in[s] = A[s] + lane < B[s]; v[s] = in[s] ? (uint64_t)F[A[s] + lane] : 0; }
#pragma unroll
        for (uint32_t s = 0; s < 2 * G; ++s) {
            const uint32_t blk = A[s] + (uint32_t)__popcll(__ballot(in[s] && v[s] < key[s]));
            const uint64_t w0 = (uint64_t)blk << 6, w1 = w0 + 64;
            A[s] = w0 > pbegin ? (uint32_t)w0 : pbegin;
            B[s] = w1 < pend ? (uint32_t)w1 : pend;
            if (!any[s < G ? s : s - G]) B[s] = A[s] = pbegin;
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
    // wide brackets first shrink 64-fold: lane t reads the last element of the bracket's t-th slice (one load per group), every
    // lane ranks its two keys among these 64 fences through cross-lane reads and goes on inside one slice
    if (F && uniform(widest) > 4096) {
        // fences: every lane finds the block of the list that holds each of its two answers by bisecting the bracket's FENCES
        // (dense: a bracket of 65 536 elements has 4 KiB of them, shared by the whole group), then bisects inside the two blocks
        uint32_t fl0[G], fr0[G], fl1[G], fr1[G];
        uint32_t widest_f = 0;
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t gl = lo[g] >> 6, gh = on[g] ? hi[g] >> 6 : gl;          // fences [gl, gh) lie inside [lo, hi)
            fl0[g] = fl1[g] = gl;
            fr0[g] = fr1[g] = gh > gl ? gh : gl;
            widest_f = fr0[g] - gl > widest_f ? fr0[g] - gl : widest_f;
        }
        for (uint32_t w = uniform(wave_max_u32(widest_f)); w; w >>= 1) {
            uint64_t v0[G], v1[G];
            uint32_t m0[G], m1[G];
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                m0[g] = fl0[g] + ((fr0[g] - fl0[g]) >> 1);
                m1[g] = fl1[g] + ((fr1[g] - fl1[g]) >> 1);
                v0[g] = fl0[g] < fr0[g] ? (uint64_t)F[m0[g]] : 0;
                v1[g] = fl1[g] < fr1[g] ? (uint64_t)F[m1[g]] : 0;
            }
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                if (fl0[g] < fr0[g]) { if (v0[g] < a[g]) fl0[g] = m0[g] + 1; else fr0[g] = m0[g]; }
                if (fl1[g] < fr1[g]) { if (v1[g] <= b[g]) fl1[g] = m1[g] + 1; else fr1[g] = m1[g]; }
            }
        }
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            if (on[g]) {
                const uint64_t s0 = (uint64_t)fl0[g] << 6, s1 = (uint64_t)fl1[g] << 6;
                l0[g] = s0 > lo[g] ? (uint32_t)s0 : lo[g];
                l1[g] = s1 > lo[g] ? (uint32_t)s1 : lo[g];
                r0[g] = s0 + 64 < hi[g] ? (uint32_t)(s0 + 64) : hi[g];
                r1[g] = s1 + 64 < hi[g] ? (uint32_t)(s1 + 64) : hi[g];
            }
        }
        widest = 64;                                                             // what is left is at most one block per key
    } else if (uniform(widest) > 256) {
        pos_t fence[G];
        uint32_t step[G];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t W = hi[g] - lo[g];
            step[g] = (W + 63) / 64;
            const uint64_t at = (uint64_t)lo[g] + (uint64_t)(lane + 1) * step[g] - 1;
            fence[g] = W > 256 ? P[at < hi[g] ? at : hi[g] - 1] : (pos_t)0;      // slices behind the bracket repeat its last element
        }
        widest = 0;
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t W = hi[g] - lo[g];
            if (W > 256) {                                                   // wave-uniform
                const uint64_t k1 = b[g] + 1;                                    // b < ~0 (the callers see to it)
                uint32_t s0 = 0, e0 = 64, s1 = 0, e1 = 64;                       // first fence >= a / >= b + 1 (64 = none)
#pragma unroll
                for (uint32_t st = 0; st < 7; ++st) {
                    const uint32_t c0 = (s0 + e0) >> 1, c1 = (s1 + e1) >> 1;
                    const uint64_t f0 = (uint64_t)__shfl(fence[g], (int)(c0 & 63)), f1 = (uint64_t)__shfl(fence[g], (int)(c1 & 63));
                    if (s0 < e0) { if (f0 < a[g]) s0 = c0 + 1; else e0 = c0; }
                    if (s1 < e1) { if (f1 < k1) s1 = c1 + 1; else e1 = c1; }
                }
                if (on[g]) {
                    // the answer is behind fence s-1 and not behind fence s: indices [lo + s step, lo + (s+1) step - 1], clipped to hi
                    const uint64_t b0 = (uint64_t)lo[g] + (uint64_t)s0 * step[g], b1 = (uint64_t)lo[g] + (uint64_t)s1 * step[g];
                    l0[g] = b0 < hi[g] ? (uint32_t)b0 : hi[g];
                    l1[g] = b1 < hi[g] ? (uint32_t)b1 : hi[g];
                    const uint64_t t0 = b0 + step[g] - 1, t1 = b1 + step[g] - 1;
                    r0[g] = t0 < hi[g] ? (uint32_t)t0 : hi[g];
                    r1[g] = t1 < hi[g] ? (uint32_t)t1 : hi[g];
                }
                widest = step[g] > widest ? step[g] : widest;
            } else {
                widest = W > widest ? W : widest;
            }
        }
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

// The same ranges through the ladder of join_device.hpp (rungs): every lane descends for its a and for its b + 1 from the level
// whose group of kRungFan entries covers the whole list, all 2 x kPivotGroups descents in lockstep (one aligned load of a group each
// per level).  The upper levels are the same address in every lane; no bracket searches, no bisection.
template <typename pos_t>
__device__ __forceinline__ void pivot_ranges_rungs(const pos_t* __restrict__ P, const pos_t* __restrict__ R, const uint64_t* __restrict__ roff,
                                                   uint32_t pbegin, uint32_t pend, const uint64_t (&a)[kPivotGroups], const uint64_t (&b)[kPivotGroups],
                                                   bool (&on)[kPivotGroups], uint32_t (&i0)[kPivotGroups], uint32_t (&i1)[kPivotGroups])
{
    constexpr uint32_t G = kPivotGroups, F = kRungFan, S = kRungShift;
    constexpr pos_t kMax = (pos_t)~(pos_t)0;
    using Quad = RungQuad<pos_t>;
    // keys as positions: nothing in a list reaches kMax, so a >= kMax finds the end and b >= kMax - 1 takes everything
    pos_t ka[G], kb[G];
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        ka[g] = a[g] >= (uint64_t)kMax ? kMax : (pos_t)a[g];
        kb[g] = b[g] >= (uint64_t)kMax - 1 ? (pos_t)(kMax - 1) : (pos_t)b[g];
    }
    const uint32_t diff = pbegin ^ (pend - 1);                                  // (the list is not empty)
    const uint32_t jt = diff ? (31u - (uint32_t)__builtin_clz(diff)) / S : 0;   // its indices agree above bit S (jt + 1) - 1
    uint32_t ga[G], gb[G];
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) ga[g] = gb[g] = S * (jt + 1) < 32 ? pbegin >> (S * (jt + 1)) : 0;
    for (int j = (int)jt; j >= 0; --j) {
        const pos_t* __restrict__ L = j ? R + roff[j] : P;
        const uint32_t A = pbegin >> (S * j), B = pend >> (S * j);              // entries [A, B) of this level are elements of the list
        Quad ea[G][F / 4], eb[G][F / 4];
        bool edge = false;
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            if (on[g]) {
#pragma unroll
                for (uint32_t c = 0; c < F / 4; ++c) {
                    ea[g][c] = *reinterpret_cast<const Quad*>(L + F * (uint64_t)ga[g] + 4 * c);
                    eb[g][c] = *reinterpret_cast<const Quad*>(L + F * (uint64_t)gb[g] + 4 * c);
                }
                edge |= F * ga[g] < A || F * ga[g] + F > B || F * gb[g] < A || F * gb[g] + F > B;
            }
        }
        if (__any(edge)) {                                                       // a group reaches over a border of the list
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                if (on[g]) {
#pragma unroll
                    for (uint32_t t = 0; t < F; ++t) {
                        const uint32_t ia = F * ga[g] + t, ib = F * gb[g] + t;
                        pos_t& xa = ea[g][t / 4].v[t % 4];
                        pos_t& xb = eb[g][t / 4].v[t % 4];
                        xa = ia < A ? (pos_t)0 : (ia >= B ? kMax : xa);
                        xb = ib < A ? (pos_t)0 : (ib >= B ? kMax : xb);
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            if (on[g]) {
                uint32_t ca = 0, cb = 0;
#pragma unroll
                for (uint32_t t = 0; t + 1 < F; ++t) {
                    ca += (uint32_t)(ea[g][t / 4].v[t % 4] < ka[g]);
                    cb += (uint32_t)(eb[g][t / 4].v[t % 4] <= kb[g]);
                }
                if (j == 0) {                                                    // (above: the last block is where larger keys go)
                    ca += (uint32_t)(ea[g][F / 4 - 1].v[3] < ka[g]);
                    cb += (uint32_t)(eb[g][F / 4 - 1].v[3] <= kb[g]);
                }
                ga[g] = F * ga[g] + ca;
                gb[g] = F * gb[g] + cb;
            }
        }
    }
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        uint32_t l0 = ga[g] < pbegin ? pbegin : (ga[g] > pend ? pend : ga[g]);
        uint32_t l1 = gb[g] < pbegin ? pbegin : (gb[g] > pend ? pend : gb[g]);
        if (!on[g]) l0 = l1 = pbegin;
        i0[g] = l0 - pbegin;
        i1[g] = l1 - pbegin;
        on[g] = on[g] && l0 < l1;
    }
}

// Pivot mode: when one list of the query is much shorter than the others, the survivors are found from its elements
// outwards instead of streaming the long lists.  A lane takes one element of the pivot list (kPivotGroups of them, one per
// 64-element group of the wave's run) and follows it level by level: the elements of the neighbouring list inside its gap
// window form an index range, which is marked in that list's activity bits; the hull of the range's positions is the
// "element" followed to the next level (a superset of what the exact windows would mark, which is all the filter needs).
struct PTask { uint32_t seg0, k, p, pad; };          // first segment of the query, sub-patterns, pivot level
constexpr uint32_t kPivotRun = 64 * kPivotGroups;     // pivot elements per wave and turn
#ifndef VLG_PIVOT_TURNS
#define VLG_PIVOT_TURNS 1
#endif
constexpr uint32_t kPivotTurns = VLG_PIVOT_TURNS;     // runs a wave takes one after the other (round 4, C3: 1 -> 21.9 ms, 4 -> 24.1 ms: the task
                                                      // look-up in front of a run is not what the kernel waits for; more, shorter waves win)

template <typename pos_t, bool kRungs>
__global__ void __launch_bounds__(256) filter_pivot_kernel(const pos_t* __restrict__ P, const pos_t* __restrict__ F /* fences of P, or null */,
                                                           const pos_t* __restrict__ R, const uint64_t* __restrict__ roff /* kRungs: the ladder over P */,
                                                           const RSeg* __restrict__ segs,
                                                           const PTask* __restrict__ tasks, const uint64_t* __restrict__ task_run0,
                                                           uint32_t ntasks, uint64_t* __restrict__ abits)
{
    constexpr uint32_t G = kPivotGroups;
    const uint32_t lane = threadIdx.x & 63;
    // a wave takes kPivotTurns consecutive runs (1: measured, see kPivotTurns)
    const uint64_t run_first = uniform(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kPivotTurns;
    const uint64_t total_runs = task_run0[ntasks];
    if (run_first >= total_runs) return;
    const uint64_t run_last = run_first + kPivotTurns < total_runs ? run_first + kPivotTurns : total_runs;
    uint32_t t = wave_task_find(task_run0, ntasks, run_first);
    uint64_t t_begin = task_run0[t], t_end = task_run0[t + 1];
    PTask tk = tasks[t];
    RSeg pv = segs[tk.seg0 + tk.p];
#pragma unroll 1
    for (uint64_t run = run_first; run < run_last; ++run) {
    if (run >= t_end) {
        do { ++t; t_begin = t_end; t_end = task_run0[t + 1]; } while (run >= t_end);
        tk = tasks[t];
        pv = segs[tk.seg0 + tk.p];
    }
    const uint64_t len = pv.pend - pv.pbegin;
    const uint64_t off0 = (run - t_begin) * kPivotRun;
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
        if (kRungs) pivot_ranges_rungs(P, R, roff, sg.pbegin, sg.pend, a, b, on, i0, i1);
        else pivot_ranges(P, F, sg.pbegin, sg.pend, a, b, on, i0, i1);
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
}

// survivors per run (for the compaction offsets): the activity bits of a filtered list start on a run boundary, so run r of
// the group owns the words [32 r, 32 r + 32).  Sixteen lanes per run (16 bytes each), sixteen runs per wave with their four
// loads per lane in flight together.
constexpr uint32_t kCountRunsPerWave = 16;
__global__ void __launch_bounds__(256) filter_count_runs_kernel(const uint64_t* __restrict__ abits, uint64_t total_runs,
                                                                uint32_t* __restrict__ runcnt)
{
    static_assert(kRun == 2048, "one run = 32 activity words");
    const uint32_t lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const uint64_t r0 = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kCountRunsPerWave;
    const ulonglong2* words = reinterpret_cast<const ulonglong2*>(abits);
    ulonglong2 w[4];
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
        const uint64_t r = r0 + it * 4 + grp;
        w[it] = r < total_runs ? words[r * 16 + sub] : make_ulonglong2(0, 0);
    }
#pragma unroll
    for (uint32_t it = 0; it < 4; ++it) {
        const uint64_t r = r0 + it * 4 + grp;
        uint32_t c = (uint32_t)(__popcll(w[it].x) + __popcll(w[it].y));
        for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o);                // the four groups of the wave reduce on their own
        if (r < total_runs && sub == 0) runcnt[r] = c;
    }
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
#ifndef VLG_COMPACT_RUNS
#define VLG_COMPACT_RUNS 16
#endif
constexpr uint32_t kCompactRuns = VLG_COMPACT_RUNS;
#ifndef VLG_SPARSE_TURN
#define VLG_SPARSE_TURN 8
#endif
constexpr uint32_t kSparseTurn = VLG_SPARSE_TURN;      // survivors of a lane's half word moved per turn of the sparse path: their loads are in flight
                                                       // together (round 4, C3, the class: 2 -> 15.7 ms, 4 -> 13.6, 8 -> 13.3)
template <typename pos_t>
__global__ void __launch_bounds__(256) filter_compact_kernel(const pos_t* __restrict__ P, const RSeg* __restrict__ segs,
                                                             const uint32_t* __restrict__ task_seg, const uint64_t* __restrict__ task_run0,
                                                             uint32_t ntasks, const uint64_t* __restrict__ abits,
                                                             const uint32_t* __restrict__ run_cnt, const uint32_t* __restrict__ run_off,
                                                             pos_t* __restrict__ Pc, uint64_t pc_cap /* elements Pc can take */,
                                                             uint32_t dense_min /* runs with fewer survivors move them half a word per lane */)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r0 = uniform(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kCompactRuns;
    const uint64_t total = task_run0[ntasks];
    if (r0 >= total) return;
    unsigned long long todo = __ballot(lane < kCompactRuns && r0 + lane < total && run_cnt[r0 + lane] != 0);
    uint32_t t = 0;
    uint64_t t_begin = 0, t_end = 0;                                             // runs before t_end belong to task t (or to none yet)
    RSeg sg;
    uint64_t ahead_run = ~0ull, ahead_words = 0;                                 // the activity words of the next run, asked for a run early
    while (todo) {
        const uint64_t run = r0 + (uint32_t)(__ffsll((long long)todo) - 1);
        todo &= todo - 1;
        if (run >= t_end) {                                                      // neighbouring runs mostly share their task
            t = wave_task_find(task_run0, ntasks, run);
            t_begin = task_run0[t];
            t_end = task_run0[t + 1];
            sg = segs[task_seg[t]];
        }
        const uint64_t len = sg.pend - sg.pbegin;
        const uint64_t off0 = (run - t_begin) * kRun;
        const uint64_t off1 = off0 + kRun < len ? off0 + kRun : len;
        // the 32 activity words of the run in one load; every lane then knows where each word's survivors go
        const uint64_t w0 = (sg.abit + off0) >> 6;
        const uint32_t nw = (uint32_t)((off1 - off0 + 63) >> 6);
        const uint64_t mine = ahead_run == run ? ahead_words : (lane < nw ? abits[w0 + lane] : 0);
        ahead_run = ~0ull;
        if (todo) {                                                              // the next run with survivors, if it is of the same list:
            const uint64_t nr = r0 + (uint32_t)(__ffsll((long long)todo) - 1);   // its words are in flight while this run's survivors move
            if (nr < t_end) {
                const uint64_t n0 = (nr - t_begin) * kRun, n1 = n0 + kRun < len ? n0 + kRun : len;
                ahead_words = lane < (uint32_t)((n1 - n0 + 63) >> 6) ? abits[((sg.abit + n0) >> 6) + lane] : 0;
                ahead_run = nr;
            }
        }
        uint32_t before = (uint32_t)__popcll(mine);                              // inclusive scan over the words
        for (int o = 1; o < 32; o <<= 1) { const uint32_t v = __shfl_up(before, o); if ((int)lane >= o) before += v; }
        before -= (uint32_t)__popcll(mine);
        const uint32_t out0 = run_off[run];
        if ((uint64_t)out0 + run_cnt[run] > pc_cap) continue;                    // (a compaction launched before the counts were known: the host sees
                                                                                 // the same total and falls back to compacting chunk by chunk)
        if (uniform(run_cnt[run]) < dense_min) {
            // sparse run (the usual case: few elements survive the filter): every lane moves the survivors of its own half word, so
            // the loop runs as often as the fullest half word has survivors, not once per word
            const uint64_t wbits = __shfl(mine, (int)(lane >> 1));
            uint32_t hb = (uint32_t)(wbits >> (32 * (lane & 1)));
            const uint32_t cnt = (uint32_t)__popc(hb);
            uint32_t inc = cnt;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
            uint32_t out = out0 + inc - cnt;
            const pos_t* __restrict__ src = P + sg.pbegin + off0 + 32ull * lane;
            while (hb) {                                                         // kSparseTurn survivors per turn: their loads in flight together
                pos_t v[kSparseTurn];
                uint32_t got = 0;
#pragma unroll
                for (uint32_t u = 0; u < kSparseTurn; ++u) {
                    if (hb) {
                        const uint32_t b = (uint32_t)__ffs((int)hb) - 1;
                        hb &= hb - 1;
                        v[u] = src[b];
                        got = u + 1;
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kSparseTurn; ++u) if (u < got) Pc[out + u] = v[u];
                out += got;
            }
            continue;
        }
        // fuller runs word by word: the 64 lanes take the 64 slots of a word, so every line of the run is requested once.
        // (Measured and dropped, round 3: four neighbouring slots per lane with one 16-byte load and all eight loads of the run in
        // flight before the first store -- 29 vs 16.6 ms for the class: 130 registers, and the 16-byte loads are not aligned.)
        unsigned long long todo_w = __ballot(mine != 0);
        const pos_t* __restrict__ src = P + sg.pbegin + off0 + lane;
        while (todo_w) {                                                         // four words per turn: their loads are in flight together
            constexpr uint32_t U = 4;
            int wi[U];
            bool sel[U];
            uint32_t at[U];
            pos_t v[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                wi[u] = todo_w ? __ffsll((long long)todo_w) - 1 : -1;
                todo_w &= todo_w - 1;
                sel[u] = false;
                if (wi[u] >= 0) {                                                // (wave-uniform)
                    const uint64_t bits = __shfl(mine, wi[u]);
                    sel[u] = (bits >> lane) & 1;
                    at[u] = out0 + __shfl(before, wi[u]) + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));
                    if (sel[u]) v[u] = src[64 * wi[u]];
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) if (sel[u]) Pc[at[u]] = v[u];
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
    // The survivors of the WHOLE group are compacted right behind the count kernels, before the host has read the counts back (the GPU
    // would idle through that round trip and the chunk planning otherwise): cpre[c] = where compacted segment c starts inside Pc.
    // Valid when all survivors of the group fit pc_cap; otherwise the chunks compact their own lists as before.
    bool speculated = false;
    std::vector<uint64_t> cpre;      // [ncseg + 1]
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
    if (ws->filter_pivot && best * ws->filter_pivot_ratio <= all) return 2;
    return slots >= ws->filter_stream_min ? 1 : 0;         // the block bitmaps of a streamed query are a fixed cost (2 x n / 2^g bits)
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
vlg_status filter_group(uint64_t n_positions /* every list element is smaller */, const vlg_queries* q, vlg_workspace* ws, const Plan& pl,
                        const std::vector<uint32_t>& poff /* per sub-pattern: its list inside P */, const pos_t* P,
                        Arena& A /* advanced past the state the join chunks still need */, FilterGroup& fg, pos_t* Pc /* survivors go here */)
{
    hipStream_t st = ws->stream;
    PhaseTrace ft(st);
    const uint32_t g = filter_block_shift(n_positions);
    const uint64_t nblocks = (n_positions >> g) + 1, nbw = (nblocks + 63) / 64;
    const uint64_t nsub = q->qsub[fg.g1] - q->qsub[fg.g0];
    fg.sub0 = q->qsub[fg.g0];
    fg.eff.resize(nsub);
    fg.cidx.assign(nsub, kNone);
    for (uint64_t s = 0; s < nsub; ++s) fg.eff[s] = pl.occ[fg.sub0 + s];
    // ---- segments of the filtered queries ------------------------------------------------------------
    svec<RSeg> segs;
    svec<uint32_t> cseg;                       // segments that keep activity bits, in (query, level) order
    svec<uint64_t> crun0(1, 0);
    std::vector<uint32_t> seg_sub;                    // sub-pattern (group relative) of every segment
    uint32_t nfq = 0, kmaxf = 0;
    uint64_t abit = 0;
    svec<PTask> ptasks;                        // queries filtered from a pivot list
    svec<uint64_t> prun0(1, 0);
    for (uint64_t qi = fg.g0; qi < fg.g1; ++qi) {
        if (!fg.want[qi - fg.g0]) continue;
        const uint64_t s0 = q->qsub[qi];
        const uint32_t k = (uint32_t)(q->qsub[qi + 1] - s0);
        uint32_t pivot = 0;
        const bool by_pivot = filter_mode(q, pl, ws, qi, &pivot) == 2;
        // a pivot list keeps every element and the last list is never filtered: with two lists and the first one as the pivot
        // there is nothing to mark
        if (by_pivot && k <= 2 && pivot == 0) continue;
        if (by_pivot) {
            ptasks.push_back(PTask{(uint32_t)segs.size(), k, pivot, 0});
            prun0.push_back(prun0.back() + (pl.occ[s0 + pivot] + kPivotRun - 1) / kPivotRun);
        } else {
            kmaxf = std::max(kmaxf, k);
        }
        for (uint32_t i = 0; i < k; ++i) {
            RSeg r;
            memset(&r, 0, sizeof r);
            r.pbegin = poff[s0 + i];
            r.pend = r.pbegin + (uint32_t)pl.occ[s0 + i];
            r.lo = q->lo[s0 + i]; r.hi = q->hi[s0 + i];
            if (i + 1 < k) { r.nlo = q->lo[s0 + i + 1]; r.nhi = q->hi[s0 + i + 1]; }
            r.fq = by_pivot ? kNone : nfq; r.level = i; r.dist = k - 1 - i;
            r.abit = ~0ull;
            // the pivot list keeps every element: the join reads it where it lies (no activity bits, no private copy)
            if (i + 1 < k && !(by_pivot && i == pivot)) {
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
    if (getenv("VLG_TRACE")) {                         // pivot work by density: neighbour list length / pivot list length
        uint64_t h[24] = {0}, total = 0;
        for (const PTask& t : ptasks) {
            const uint64_t pl_ = segs[t.seg0 + t.p].pend - segs[t.seg0 + t.p].pbegin;
            for (uint32_t i = 0; i + 1 < t.k; ++i) {
                if (i == t.p) continue;
                const uint64_t nl = segs[t.seg0 + i].pend - segs[t.seg0 + i].pbegin;
                const unsigned b = bit_width64(pl_ ? nl / pl_ : 0);
                h[b < 23 ? b : 23] += pl_; total += pl_;
            }
        }
        fprintf(stderr, "[vlg trace] pivot probes by log2(neighbour/pivot) (pivot elements x levels, total %llu):", (unsigned long long)total);
        for (unsigned b = 0; b < 24; ++b) if (h[b]) fprintf(stderr, " %u:%llu", b, (unsigned long long)h[b]);
        fprintf(stderr, "\n");
    }
    fg.any = !cseg.empty();
    if (!fg.any) return VLG_OK;
    fg.ncseg = (uint32_t)cseg.size();
    fg.crun0.assign(crun0.begin(), crun0.end());
    fg.cseg.assign(cseg.begin(), cseg.end());
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
    if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (filter)");
    VLG_HIP_TRY(hipMemsetAsync(fg.d_abits, 0, (abit / 64 + 1) * 8, st));    // (11 GB on C3; not clearing them at all would save 1 ms of the step)
    VLG_HIP_TRY(hipMemcpyAsync(fg.d_segs, segs.data(), segs.size() * sizeof(RSeg), hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(fg.d_cseg, cseg.data(), cseg.size() * 4, hipMemcpyHostToDevice, st));
    VLG_HIP_TRY(hipMemcpyAsync(fg.d_crun0, crun0.data(), crun0.size() * 8, hipMemcpyHostToDevice, st));
    if (nfq) VLG_HIP_TRY(hipMemsetAsync(d_bm, 0, (uint64_t)nfq * 2 * nbw * 8, st));
    if (!ptasks.empty()) {
        VLG_HIP_TRY(hipMemcpyAsync(d_ptasks, ptasks.data(), ptasks.size() * sizeof(PTask), hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(d_prun0, prun0.data(), prun0.size() * 8, hipMemcpyHostToDevice, st));
        uint64_t probes = 0;                                      // (pivot element, level) pairs: two lower bounds of 8 bytes each
        for (const PTask& pt : ptasks) probes += (uint64_t)(segs[pt.seg0 + pt.p].pend - segs[pt.seg0 + pt.p].pbegin) * (pt.k >= 2 ? pt.k - 2 + (pt.p + 1 == pt.k ? 1 : 0) : 0);
        Timed t(ws, KS_FILTER_PIVOT, 16 * probes);
        if (ws->rungs && ws->pivot_rungs)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_pivot_kernel<pos_t, true>), dim3((uint32_t)((prun0.back() + 4 * kPivotTurns - 1) / (4 * kPivotTurns))), dim3(256), 0, st, P,
                               static_cast<const pos_t*>(ws->fences), static_cast<const pos_t*>(ws->rungs), ws->rung_off, fg.d_segs, d_ptasks, d_prun0,
                               (uint32_t)ptasks.size(), fg.d_abits);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_pivot_kernel<pos_t, false>), dim3((uint32_t)((prun0.back() + 4 * kPivotTurns - 1) / (4 * kPivotTurns))), dim3(256), 0, st, P,
                               static_cast<const pos_t*>(ws->fences), (const pos_t*)nullptr, (const uint64_t*)nullptr, fg.d_segs, d_ptasks, d_prun0,
                               (uint32_t)ptasks.size(), fg.d_abits);
        VLG_HIP_TRY(hipGetLastError());
    }
    auto clear_buf = [&](uint32_t buf) -> vlg_status {
        if (nfq) VLG_HIP_TRY(hipMemset2DAsync(d_bm + (uint64_t)buf * nbw, 2 * nbw * 8, 0, nbw * 8, nfq, st));
        return VLG_OK;
    };
    auto run_pass = [&](const RPass& ps, auto&& pick) -> vlg_status {
        svec<uint32_t> task_seg;                      // fresh pinned storage per pass: the copies below run later, in stream order
        svec<uint64_t> task_run0(1, 0);
        task_seg.reserve(segs.size());
        task_run0.reserve(segs.size() + 1);
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
        hipLaunchKernelGGL(filter_count_runs_kernel, dim3((uint32_t)((total_runs + 4 * kCountRunsPerWave - 1) / (4 * kCountRunsPerWave))), dim3(256), 0, st, fg.d_abits, total_runs, fg.d_runcnt);
        hipLaunchKernelGGL(filter_count_lists_kernel, dim3((uint32_t)((cseg.size() + 3) / 4)), dim3(256), 0, st, fg.d_crun0, fg.ncseg, fg.d_runcnt,
                           d_segcnt);
    }
    VLG_HIP_TRY(hipGetLastError());
    svec<unsigned long long> segcnt(cseg.size());
    VLG_HIP_TRY(hipMemcpyAsync(segcnt.data(), d_segcnt, cseg.size() * 8, hipMemcpyDeviceToHost, st));
    hipEvent_t counts_ready = ws_event(ws);                          // the host waits for the counts, not for what is enqueued behind them
    if (counts_ready) VLG_HIP_TRY(hipEventRecord(counts_ready, st));
    // compaction of the whole group, launched before the counts are known (the run counts and their scan stay on the device)
    bool spec_launched = false;
    const bool spec_on = [] { const char* e = getenv("VLG_NO_SPECULATIVE_COMPACT"); return !(e && e[0] == '1'); }();
    if (spec_on && Pc && fg.pc_cap && total_runs && total_runs < 0x7FFFFFFFull) {
        uint32_t* d_off = A.take<uint32_t>(total_runs + 1);
        size_t scan_tmp = 0;
        VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp, fg.d_runcnt, d_off, 0u, total_runs, rocprim::plus<uint32_t>(), st));
        void* d_scan = A.take<uint8_t>(scan_tmp + 256);
        if (!A.failed) {
            Timed t(ws, KS_FILTER_COMPACT, 0);
            VLG_HIP_TRY(rocprim::exclusive_scan(d_scan, scan_tmp, fg.d_runcnt, d_off, 0u, total_runs, rocprim::plus<uint32_t>(), st));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_compact_kernel<pos_t>), dim3((uint32_t)(((total_runs + kCompactRuns - 1) / kCompactRuns + 3) / 4)),
                               dim3(256), 0, st, P, fg.d_segs, fg.d_cseg, fg.d_crun0, fg.ncseg, fg.d_abits, fg.d_runcnt, d_off, Pc, fg.pc_cap, ws->compact_dense_min);
            VLG_HIP_TRY(hipGetLastError());
            spec_launched = true;
        } else A.failed = false;                                     // no room for the scan: the chunks compact their own lists
    }
    ft.mark("  filter: launched");
    if (counts_ready) {
        const hipError_t e = hipEventSynchronize(counts_ready);
        ws->free_events.push_back(counts_ready);
        VLG_HIP_TRY(e);
    } else VLG_HIP_TRY(hipStreamSynchronize(st));
    ft.mark("  filter: counts back");
    for (uint32_t c = 0; c < cseg.size(); ++c) fg.eff[seg_sub[cseg[c]]] = segcnt[c];
    if (spec_launched) {
        fg.cpre.assign(cseg.size() + 1, 0);
        for (uint32_t c = 0; c < cseg.size(); ++c) fg.cpre[c + 1] = fg.cpre[c] + segcnt[c];
        fg.speculated = fg.cpre.back() <= fg.pc_cap;
        if (fg.speculated) {
            ws->stats[KS_FILTER_COMPACT].algorithmic_bytes += 2 * fg.cpre.back() * sizeof(pos_t);
            // fences of the survivors' lists: whole blocks of [Pc, Pc + total) (Pc starts on a block)
            if (ws->fences && fg.cpre.back() >= 64) {
                const uint64_t g0 = (uint64_t)(Pc - P) / 64;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(fence_build_kernel<pos_t>), dim3(grid_for(fg.cpre.back() / 64, 8192)), dim3(256), 0, st, P, g0, g0 + fg.cpre.back() / 64,
                                   static_cast<pos_t*>(ws->fences));
                VLG_HIP_TRY(hipGetLastError());
            }
        }
    }
    // a query that lost a whole list has no match; one whose survivors do not fit a chunk is joined on its full lists
    for (uint64_t qi = fg.g0; qi < fg.g1; ++qi) {
        const uint64_t s0 = q->qsub[qi] - fg.sub0, k = q->qsub[qi + 1] - q->qsub[qi];
        bool was_filtered = false;
        for (uint64_t i = 0; i < k; ++i) was_filtered |= fg.cidx[s0 + i] != kNone;
        if (!was_filtered) continue;
        bool dead = false;
        uint64_t sum = 0;
        for (uint64_t i = 0; i + 1 < k; ++i) { dead |= fg.eff[s0 + i] == 0; if (fg.cidx[s0 + i] != kNone) sum += fg.eff[s0 + i]; }
        if (dead) for (uint64_t i = 0; i < k; ++i) fg.eff[s0 + i] = 0;
        else if (sum > fg.pc_cap && !fg.speculated) for (uint64_t i = 0; i < k; ++i) { fg.eff[s0 + i] = pl.occ[fg.sub0 + s0 + i]; fg.cidx[s0 + i] = kNone; }
    }
    A.used = keep;                                   // bitmaps, counters and task lists are dead
    return VLG_OK;
}

}  // namespace
