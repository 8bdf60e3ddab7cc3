// Types shared by the translation units of the search path.
#pragma once
#include <cstdint>
#include <vector>

struct vlg_queries {
    uint64_t nq = 0, nsub = 0;
    std::vector<uint64_t> qsub;      // [nq+1]
    std::vector<uint64_t> suboff;    // [nsub+1]
    std::vector<uint8_t> blob;
    std::vector<uint64_t> lo, hi;    // [nsub]
    std::vector<uint64_t> end_len;   // [nq]
    uint32_t kmax = 0, kmin = 0;     // over queries with at least one sub-pattern
    uint32_t sym_bytes = 1;          // 1: byte sub-patterns; 4: integer alphabet (vlg_queries_parse_int), symbols little-endian in blob
    uint8_t* d_blob = nullptr;
    uint64_t* d_suboff = nullptr;
    uint64_t* d_qsub = nullptr;      // [nq+1] on the device (the interval plan groups sub-patterns by query)
};
