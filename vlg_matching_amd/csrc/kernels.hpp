// Internal launch interface between kernels.hip and search.hip.
#pragma once
#include <functional>
#include "common.hpp"

namespace vlg {

vlg_status launch_backward_search(const IndexView& iv, const uint8_t* d_blob, const uint64_t* d_off, uint64_t n_pat, uint64_t* d_l,
                                  uint64_t* d_r, unsigned long long* d_stat_levels, hipStream_t stream);
template <typename pos_t>
vlg_status launch_expand(const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total, pos_t* d_io, uint32_t* d_seg,
                         hipStream_t stream);
template <typename pos_t>
vlg_status launch_locate(const IndexView& iv, pos_t* d_io, uint64_t total, unsigned long long* d_stats, hipStream_t stream);

// Optional per-launch timing hooks (HIP events on the stream), implemented by the workspace.
struct LaunchTimer {
    virtual ~LaunchTimer() {}
    virtual void begin(int which, uint64_t algorithmic_bytes = 0) = 0;     // which: 0 = LF step kernel, 1 = stable partition by symbol, 2 = trail record resolution
    virtual void end(int which) = 0;
};

size_t sweep_temp_bytes(uint64_t total, uint32_t sigma, hipStream_t stream);
// occurrences one sweep can cover: slots share a 64-bit word with the SA index (32 + 32 bits, or 31 + 33 when SA indices are wide)
template <bool kWide> constexpr uint64_t sweep_batch_max() { return kWide ? (1ull << 31) : 0xFFFFFF00ull; }
// kWide: the index keeps 64-bit samples and SA indices may need 33 bits (n > 2^32, or VLG_FORCE_POS64); pos_t is the width of the
// text positions written -- uint32_t whenever the text has at most 2^32 characters, whatever the width of the SA indices.
template <typename pos_t, bool kWide>
vlg_status launch_locate_sweep(const IndexView& iv, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total,
                               pos_t* d_out, uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp,
                               size_t temp_bytes, unsigned long long* d_counter, unsigned long long* d_stats, uint64_t tail_threshold,
                               hipStream_t stream, LaunchTimer* timer, Block* member, uint32_t n_member_lists, uint64_t* rec,
                               const std::function<vlg_status()>* while_first_step = nullptr,
                               uint8_t* front = nullptr /* with rec: one byte per element, the symbol in front of it (see SweepKernels) */);
// The three launches of a sorted sweep over some index; run_locate_sweep owns the rounds, the partitions, the member bit-vector, the
// records and their resolution.  `out` is the sweep's slice of the position array (pos_t*), `rec` null when no LF step is shared.
constexpr uint32_t kSweepChunk = 2048;          // elements a workgroup of the first round takes per turn
// the list that holds the first element of every chunk of a sweep [t0, t1): looked up for all chunks at once, in parallel, instead of by
// one lane of every workgroup in front of its work (a binary search is seventeen dependent loads)
void launch_sweep_chunk_lists(const uint64_t* d_out_off, uint64_t n_pat, uint64_t t0, uint64_t t1, uint32_t* chunk_list, hipStream_t stream);
struct SweepKernels {
    // front[slot] (optional, with records): the compact symbol read in front of an element that stopped on its first step, 0xFF otherwise
    // -- what the resolve regroups a workgroup's records by, so that the first hop of neighbouring lanes reads neighbouring records
    uint8_t* front = nullptr;
    uint64_t n = 0;
    uint32_t sigma = 0;                  // partition keys are the symbols 0 .. sigma - 1 (16 bits at most); sigma = finished
    // round 0 of the elements [t0, t1): their words follow from their places (lists l / out_off are the launcher's business)
    std::function<void(uint64_t t0, uint64_t t1, uint64_t* val, uint16_t* key, void* out, unsigned long long* n_done, const Block* member, uint64_t* rec, bool ahead,
                       uint32_t* chunk_list /* scratch, one word per kSweepChunk elements: sweep_chunk_lists_kernel */)> first;
    // one round of the elements val / key [0, alive)
    std::function<void(uint64_t* val, uint16_t* key, uint64_t alive, uint32_t step, void* out, unsigned long long* n_done, const Block* member, uint64_t* rec, uint64_t t0,
                       bool probed)> step;
    // the stragglers, unsorted: a wave owns per_wave elements of val
    std::function<void(void* out, uint64_t alive, uint32_t per_wave, const uint64_t* val, uint32_t step, uint64_t* rec_or_null, uint64_t t0, const Block* member,
                       uint32_t blocks)> tail;
};
template <typename pos_t, bool kWide>
vlg_status run_locate_sweep(const SweepKernels& K, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total,
                            pos_t* d_out, uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp,
                            size_t temp_bytes, unsigned long long* d_counter, uint64_t tail_threshold, hipStream_t stream, LaunchTimer* timer,
                            Block* member, uint32_t n_member_lists, uint64_t* rec, const std::function<vlg_status()>* while_first_step = nullptr);
// super-blocks of the member bit-vector over the SA indices of a text of n - 1 characters (kernels.hip: sweep_element)
inline uint64_t member_blocks(uint64_t n) { return n / kBlockBits + 1; }
// K3u: the whole suffix array reconstructed from the samples (one LF walker per sample, n LF steps in all) into sa_full (n x 4 B),
// then the SA intervals of the lists copied into d_out.  For batches that locate a large part of all text positions.
// Scratch: val_a / val_b (8 B) and key_a / key_b (2 B) for n_samples walkers, temp as for the sweep, counter 8 B.
template <bool kWide>
vlg_status launch_unsample(const IndexView& iv, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total, uint32_t* d_out,
                           uint32_t* sa_full, uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp, size_t temp_bytes,
                           unsigned long long* d_counter, unsigned long long* h_done /* 8 pinned bytes */, unsigned long long* d_stats, uint64_t tail_threshold,
                           hipStream_t stream, LaunchTimer* timer, const std::function<vlg_status()>* while_first_step = nullptr);
template <typename T> vlg_status launch_narrow(const uint64_t* d_in, T* d_out, uint64_t count, hipStream_t stream);
template <typename T> vlg_status launch_widen(const T* d_in, uint64_t* d_out, uint64_t count, hipStream_t stream);

// integer-alphabet FM-index (int_index.hpp): backward_search over uint32_t symbols, csa[i] in place (n <= 2^32)
vlg_status launch_int_backward_search(const IntView& v, const uint8_t* d_blob, const uint64_t* d_off, uint64_t n_pat, uint64_t* d_l, uint64_t* d_r,
                                      unsigned long long* d_stat_levels, hipStream_t st);
vlg_status launch_int_locate(const IntView& v, uint32_t* d_io, uint64_t total, unsigned long long* d_stats, hipStream_t st);
vlg_status launch_int_locate_sweep(const IntView& v, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total, uint32_t* d_out,
                                   uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp, size_t temp_bytes, unsigned long long* d_counter,
                                   unsigned long long* d_stats, uint64_t tail_threshold, hipStream_t stream, LaunchTimer* timer, Block* member,
                                   uint32_t n_member_lists, uint64_t* rec, const std::function<vlg_status()>* while_first_step = nullptr);
// can this index's occurrences be located by the sorted sweep at all (the byte index always; the integer index with a 16-bit key)
inline bool int_sweep_possible(const IntView& v) { return v.sigma < 0xFFFFu && v.n_levels >= 1 && v.n <= (1ull << 32); }

}  // namespace vlg
