// Host-side mirror of the reference's query/collection types for the VLG path.
//   gapped_pattern        ~ benchmark/gapped-matching/include/utils.hpp:19-71   (benchmark dialect, no '?')
//   gapped_search_result  ~ utils.hpp:73-80
//   collection            ~ benchmark/gapped-matching/include/collection.hpp:19-44 (only the TEXT entry is used)
// Parsing is done by the library (vlg_parse_query), so the C++ and Python hosts and the kernels agree by construction.
#pragma once
#include <cstdint>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "../../include/vlg_hip.h"

namespace vlg_host {

typedef std::vector<uint64_t> string_type;

inline void check(vlg_status st)
{
    if (st != VLG_OK) throw std::runtime_error(std::string("vlg: ") + vlg_last_error());
}

struct gapped_pattern {
    std::string raw_regexp;
    std::vector<string_type> subpatterns;
    std::vector<std::pair<uint64_t, uint64_t>> gaps;      // raw (min,max) as written, like utils.hpp:60

    // throws std::runtime_error on a malformed gap, like the reference (utils.hpp:57-59); the driver skips such lines
    gapped_pattern(const std::string& p, bool string_patterns = true, int dialect = VLG_DIALECT_BENCHMARK) : raw_regexp(p)
    {
        if (!string_patterns) throw std::runtime_error("integer-alphabet patterns are not supported by the GPU path");
        vlg_parsed_query q;
        check(vlg_parse_query(p.data(), p.size(), dialect, &q));
        for (uint32_t i = 0; i < q.k; ++i) {
            string_type s;
            for (uint64_t j = 0; j < q.sub_len[i]; ++j) s.push_back((uint8_t)p[q.sub_off[i] + j]);
            subpatterns.push_back(s);
            if (i) {
                uint64_t base = dialect == VLG_DIALECT_BENCHMARK ? q.sub_len[0] : q.sub_len[i - 1];
                gaps.emplace_back(q.lo[i] - base, q.hi[i] - base);
            }
        }
    }
};

struct gapped_search_result {
    std::vector<uint64_t> positions;
    gapped_search_result() = default;
    explicit gapped_search_result(size_t n) { positions.resize(n); }
};

struct collection {
    std::string path;
    std::map<std::string, std::string> file_map;
    collection() = default;
    explicit collection(const std::string& p) : path(p)
    {
        file_map["TEXT"] = path + "/text.TEXT";
        std::ifstream probe(file_map["TEXT"]);
        if (!probe) throw std::runtime_error("collection has no text.TEXT: " + path);
    }
};

// sdsl int_vector<0> on disk: u64 size-in-bits, u8 width, ceil(bits/64) words (include/sdsl/int_vector.hpp:584-600,1507-1557)
inline std::vector<uint8_t> read_text_file(const std::string& file)
{
    std::ifstream in(file, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + file);
    uint64_t bits = 0;
    uint8_t width = 0;
    in.read((char*)&bits, 8);
    in.read((char*)&width, 1);
    if (!in || width == 0 || width > 8) throw std::runtime_error("not a byte-alphabet int_vector<0>: " + file);
    std::vector<uint64_t> words((bits + 63) / 64 + 1, 0);
    in.read((char*)words.data(), (std::streamsize)(((bits + 63) / 64) * 8));
    uint64_t n = bits / width;
    std::vector<uint8_t> text(n);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t b = i * width, w = b >> 6, o = b & 63;
        uint64_t v = words[w] >> o;
        if (o + width > 64) v |= words[w + 1] << (64 - o);
        text[i] = (uint8_t)(v & ((1u << width) - 1));
    }
    return text;
}

// the writer used by create_collection (benchmark/gapped-matching/src/create_collection.cpp:79-93)
inline void write_text_file(const std::string& file, const std::vector<uint8_t>& text)
{
    uint8_t mx = 1;
    for (uint8_t c : text) if (c > mx) mx = c;
    uint8_t width = 1;
    while ((1u << width) <= mx) ++width;
    uint64_t bits = (uint64_t)text.size() * width;
    std::vector<uint64_t> words((bits + 63) / 64 + 1, 0);
    for (uint64_t i = 0; i < text.size(); ++i) {
        uint64_t b = i * width, w = b >> 6, o = b & 63;
        words[w] |= (uint64_t)text[i] << o;
        if (o + width > 64) words[w + 1] |= (uint64_t)text[i] >> (64 - o);
    }
    std::ofstream out(file, std::ios::binary);
    out.write((const char*)&bits, 8);
    out.write((const char*)&width, 1);
    out.write((const char*)words.data(), (std::streamsize)(((bits + 63) / 64) * 8));
}

}  // namespace vlg_host
