// vlg_index_gpu<alphabet_tag> -- the reference's LIBRARY concept on top of the C-ABI (include/sdsl/vlg_index.hpp):
//   vlg_index<alphabet_tag, t_wt>        :109-198   text + wavelet tree over the suffix array (here: in HBM, vlg_wtsa_*)
//   construct / construct_im             :375-392, construct.hpp
//   locate(idx, query) -> container<vlg_iterator>   :394-401; iterator surface :293-373 (operator*, operator[], size(), is_end(), ++)
//   count(idx, query)                    :403-411
// The iterator is lazy like the reference's: it asks the device for the first matches only and, when the caller walks past them,
// for four times as many (vlg_wtsa_search_batch's max_matches_per_query) -- a caller that stops early never pays for the rest.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "gapped_pattern.hpp"

namespace vlg_host {

struct byte_alphabet_tag { static const uint8_t WIDTH = 8; };
struct int_alphabet_tag { static const uint8_t WIDTH = 0; };

template <typename alphabet_tag = byte_alphabet_tag>
class vlg_index_gpu
{
    struct handles {
        vlg_wtsa* idx = nullptr;
        vlg_workspace* ws = nullptr;
        ~handles() { if (ws) vlg_workspace_destroy(ws); if (idx) vlg_wtsa_destroy(idx); }
    };
    std::shared_ptr<handles> m_h;          // value semantics like the reference's index (copies share the immutable device image)

  public:
    typedef alphabet_tag alphabet_category;
    typedef uint64_t size_type;
    typedef std::string query_type;        // the regex-style query string; parsed by the library (vlg_index.hpp:54-105)
    static const bool byte_symbols = alphabet_tag::WIDTH == 8;

    vlg_index_gpu() = default;
    // construct_im(idx, text, num_bytes): byte alphabet = the characters; integer alphabet = the values
    void build(const void* symbols, uint64_t n)
    {
        std::shared_ptr<handles> h(new handles());
        check(vlg_wtsa_build(symbols, n, byte_symbols ? 1 : 4, &h->idx));
        check(vlg_workspace_create(0, nullptr, &h->ws));
        m_h = h;
    }
    bool empty() const { return !m_h; }
    size_type size() const
    {
        vlg_wtsa_info i;
        check(vlg_wtsa_get_info(m_h->idx, &i));
        return i.n - 1;
    }

    // the first `cap` matches (0 = all) of one query: tuples, k values per match
    void fetch(const std::string& query, uint64_t cap, std::vector<uint64_t>& tuples, uint32_t& k, uint64_t& matches) const
    {
        if (!m_h) throw std::runtime_error("vlg_index_gpu: not constructed");
        uint64_t off[2] = {0, query.size()};
        vlg_queries* q = nullptr;
        check(byte_symbols ? vlg_queries_parse(query.data(), off, 1, VLG_DIALECT_LIBRARY, nullptr, &q)
                           : vlg_queries_parse_int(query.data(), off, 1, nullptr, &q));          // runtime_error texts of vlg_index.hpp:92-99
        vlg_status st = vlg_queries_k(q, &k);
        vlg_result* r = nullptr;
        if (!st) st = vlg_workspace_set_option(m_h->ws, "tuples", 1);
        if (!st) st = vlg_wtsa_search_batch(m_h->idx, q, cap, m_h->ws, &r);
        vlg_queries_destroy(q);
        check(st);
        vlg_result_summary s;
        st = vlg_result_summary_get(r, &s);
        tuples.assign(st ? 1 : s.n_tuple_values + 1, 0);
        if (!st) st = vlg_result_fetch(r, nullptr, nullptr, nullptr, tuples.data());
        vlg_result_destroy(r);
        check(st);
        matches = s.n_matches;
        tuples.resize(s.n_tuple_values);
    }
};

// vlg_iterator (vlg_index.hpp:209-373): forward iterator over the matches of one query
template <typename type_index>
class vlg_iterator_gpu
{
    const type_index* m_idx = nullptr;
    std::string m_query;
    std::vector<uint64_t> m_tuples;
    uint32_t m_k = 0;
    uint64_t m_have = 0, m_cap = 0, m_at = 0;
    bool m_all = true, m_end = true;

    void refill(uint64_t cap)
    {
        m_cap = cap;
        m_idx->fetch(m_query, cap, m_tuples, m_k, m_have);
        m_all = m_have < cap;                                  // fewer than asked for: that was everything
    }

  public:
    typedef uint64_t position_type;
    vlg_iterator_gpu() = default;
    vlg_iterator_gpu(const type_index& idx, const std::string& query) : m_idx(&idx), m_query(query), m_end(false)
    {
        refill(16);
        m_end = m_have == 0;
    }
    bool is_end() const { return m_end; }
    size_t size() const { return m_k; }
    position_type operator[](int i) const { return m_tuples[m_at * m_k + (uint64_t)i]; }
    position_type operator*() const { return (*this)[0]; }
    vlg_iterator_gpu& operator++()
    {
        if (m_end) return *this;
        ++m_at;
        if (m_at >= m_have) {
            if (m_all) m_end = true;
            else { refill(m_cap * 4); m_end = m_at >= m_have; }   // the matches come in the same order: the first m_at are the ones already seen
        }
        return *this;
    }
    friend bool operator==(const vlg_iterator_gpu& a, const vlg_iterator_gpu& b) { return a.is_end() && b.is_end(); }
    friend bool operator!=(const vlg_iterator_gpu& a, const vlg_iterator_gpu& b) { return !(a == b); }
};

template <typename t_iter>
struct container {                                             // include/sdsl/iterators.hpp:169-189
    t_iter m_begin, m_end;
    container(t_iter b, t_iter e) : m_begin(b), m_end(e) {}
    t_iter begin() const { return m_begin; }
    t_iter end() const { return m_end; }
};

inline void construct_im(vlg_index_gpu<byte_alphabet_tag>& idx, const std::string& text, uint8_t /*num_bytes*/ = 1) { idx.build(text.data(), text.size()); }
inline void construct_im(vlg_index_gpu<int_alphabet_tag>& idx, const std::vector<uint64_t>& text)
{
    std::vector<uint32_t> t(text.size());
    for (size_t i = 0; i < text.size(); ++i) {
        if (text[i] > 0xFFFFFFFFull) throw std::runtime_error("integer alphabet: symbols beyond 32 bits are not supported");
        t[i] = (uint32_t)text[i];
    }
    idx.build(t.data(), t.size());
}
// construct(idx, file, num_bytes): the raw file as bytes (num_bytes = 1; the reference's other encodings are not read here)
inline void construct(vlg_index_gpu<byte_alphabet_tag>& idx, const std::string& file, uint8_t /*num_bytes*/ = 1)
{
    std::ifstream in(file, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + file);
    std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    idx.build(text.data(), text.size());
}

template <typename type_index>
container<vlg_iterator_gpu<type_index>> locate(const type_index& idx, const typename type_index::query_type& query)
{
    return container<vlg_iterator_gpu<type_index>>(vlg_iterator_gpu<type_index>(idx, query), vlg_iterator_gpu<type_index>());
}
template <typename type_index>
typename type_index::size_type count(const type_index& idx, const typename type_index::query_type& query)
{
    typename type_index::size_type result = 0;
    auto cont = locate(idx, query);
    for (auto it = cont.begin(); it != cont.end(); ++it) ++result;
    return result;
}

}  // namespace vlg_host
