// Internal launch interface between kernels.hip and search.hip.
#pragma once
#include "common.hpp"

namespace vlg {

vlg_status launch_backward_search(const IndexView& iv, const uint8_t* d_blob, const uint64_t* d_off, uint64_t n_pat, uint64_t* d_l,
                                  uint64_t* d_r, unsigned long long* d_stat_levels, hipStream_t stream);
template <typename pos_t>
vlg_status launch_expand(const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total, pos_t* d_io, uint32_t* d_seg,
                         hipStream_t stream);
template <typename pos_t>
vlg_status launch_locate(const IndexView& iv, pos_t* d_io, uint64_t total, unsigned long long* d_stats, hipStream_t stream);

}  // namespace vlg
