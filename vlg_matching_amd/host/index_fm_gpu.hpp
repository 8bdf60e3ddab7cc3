// index_fm_gpu -- satisfies the benchmark's index concept (what bench_index<t_idx> / create_and_store<t_idx> use:
// benchmark/gapped-matching/src/gm_search.cpp:62-105, gm_index.cpp:38-60; exemplar index_sasearch.hpp:7-119) on top of
// the C-ABI, plus the batched entry a GPU needs (search_batch).  Value semantics like the reference: the object owns
// its HBM-resident index.
#pragma once
#include <algorithm>
#include <cstring>
#include <iostream>
#include <memory>
#include <thread>
#include "gapped_pattern.hpp"

namespace vlg_host {

class index_fm_gpu
{
  private:
    vlg_index* m_idx = nullptr;
    // search() is const in the reference's concept (index_sasearch.hpp:59, gm_search.cpp:96-105): scratch memory and the
    // replicas on other devices are caches, not state of the index.  Like the reference's, a const index may be searched by one
    // host thread at a time.
    mutable vlg_workspace* m_ws = nullptr;
    struct replica { int device; vlg_index* idx; vlg_workspace* ws; };
    mutable std::vector<replica> m_replicas;       // device d = m_replicas[d-1] (device 0 holds m_idx itself)

    void ensure_ws() const
    {
        if (!m_ws) check(vlg_workspace_create(0, nullptr, &m_ws));
    }
    void drop_replicas() const
    {
        for (auto& r : m_replicas) { if (r.ws) vlg_workspace_destroy(r.ws); if (r.idx) vlg_index_destroy(r.idx); }
        m_replicas.clear();
    }
    struct parsed_batch {
        vlg_queries* q = nullptr;
        ~parsed_batch() { if (q) vlg_queries_destroy(q); }
    };
    static void parse(const std::vector<gapped_pattern>& pats, size_t b, size_t e, int dialect, parsed_batch& out)
    {
        std::string text;
        std::vector<uint64_t> off(1, 0);
        for (size_t i = b; i < e; ++i) { text += pats[i].raw_regexp; off.push_back(text.size()); }
        std::vector<int> status(e - b + 1, 0);
        check(vlg_queries_parse(text.data(), off.data(), e - b, dialect, status.data(), &out.q));
    }
    // one pass of the hot path over pats[b,e) on the current device; positions of query i go to out[i]
    static void search_range(const vlg_index* idx, vlg_workspace* ws, const std::vector<gapped_pattern>& pats, size_t b, size_t e, int dialect,
                             std::vector<gapped_search_result>& out, vlg_result_summary* summary)
    {
        check(vlg_workspace_set_option(ws, "tuples", 0));        // gapped_search_result holds first positions only
        parsed_batch pb;
        parse(pats, b, e, dialect, pb);
        vlg_result* r = nullptr;
        check(vlg_search_batch(idx, pb.q, ws, &r));
        vlg_result_summary s;
        vlg_status st = vlg_result_summary_get(r, &s);
        std::vector<uint64_t> offsets(e - b + 1), first(st ? 1 : s.n_matches + 1);
        if (!st) st = vlg_result_fetch(r, nullptr, offsets.data(), first.data(), nullptr);
        vlg_result_destroy(r);
        check(st);
        if (summary) *summary = s;
        for (size_t i = b; i < e; ++i) out[i].positions.assign(first.begin() + offsets[i - b], first.begin() + offsets[i - b + 1]);
    }

  public:
    typedef uint64_t size_type;
    std::string name() const { return "FMGPU-csa_wt_wt_huff_d32"; }

    index_fm_gpu() {}
    // index_*(collection&): build from <col>/text.TEXT (index_sasearch.hpp:23-31) -- suffix sort, BWT, wavelet tree on the device
    explicit index_fm_gpu(collection& col)
    {
        std::vector<uint8_t> text = read_text_file(col.file_map["TEXT"]);
        check(vlg_index_build(text.data(), text.size(), 32, &m_idx));
    }
    explicit index_fm_gpu(const std::vector<uint8_t>& text) { check(vlg_index_build(text.data(), text.size(), 32, &m_idx)); }
    index_fm_gpu(const index_fm_gpu&) = delete;
    index_fm_gpu& operator=(const index_fm_gpu&) = delete;
    ~index_fm_gpu()
    {
        drop_replicas();
        if (m_ws) vlg_workspace_destroy(m_ws);
        if (m_idx) vlg_index_destroy(m_idx);
    }

    // serialize / load: the index parts in the reference's own layout (bit-vector, tree nodes, C, samples)
    size_type serialize(std::ostream& out, void* = nullptr, std::string = "") const
    {
        vlg_index_parts sz;
        check(vlg_index_export_parts(m_idx, &sz, nullptr));
        std::vector<uint8_t> c2c(256);
        std::vector<uint64_t> C(257), bv((sz.bv_bits + 63) / 64 + 1), smp(sz.n_samples + 1);
        std::vector<vlg_wt_node> nodes(sz.n_nodes + 1);
        vlg_index_parts_out o{c2c.data(), C.data(), bv.data(), nodes.data(), smp.data()};
        check(vlg_index_export_parts(m_idx, &sz, &o));
        const char magic[8] = {'V', 'L', 'G', 'P', 'A', 'R', 'T', '1'};
        uint64_t hdr[6] = {sz.n, sz.sigma, sz.sa_sample_dens, sz.bv_bits, sz.n_nodes, sz.n_samples};
        out.write(magic, 8);
        out.write((const char*)hdr, sizeof hdr);
        out.write((const char*)c2c.data(), 256);
        out.write((const char*)C.data(), 257 * 8);
        out.write((const char*)bv.data(), (std::streamsize)(((sz.bv_bits + 63) / 64) * 8));
        out.write((const char*)nodes.data(), (std::streamsize)(sz.n_nodes * sizeof(vlg_wt_node)));
        out.write((const char*)smp.data(), (std::streamsize)(sz.n_samples * 8));
        return 8 + sizeof hdr + 256 + 257 * 8 + ((sz.bv_bits + 63) / 64) * 8 + sz.n_nodes * sizeof(vlg_wt_node) + sz.n_samples * 8;
    }

    void load(std::istream& in)
    {
        char magic[8];
        uint64_t hdr[6];
        in.read(magic, 8);
        in.read((char*)hdr, sizeof hdr);
        if (!in || std::string(magic, 8) != "VLGPART1") throw std::runtime_error("not a VLG index file");
        std::vector<uint8_t> c2c(256);
        std::vector<uint64_t> C(257), bv((hdr[3] + 63) / 64 + 1), smp(hdr[5] + 1);
        std::vector<vlg_wt_node> nodes(hdr[4] + 1);
        in.read((char*)c2c.data(), 256);
        in.read((char*)C.data(), 257 * 8);
        in.read((char*)bv.data(), (std::streamsize)(((hdr[3] + 63) / 64) * 8));
        in.read((char*)nodes.data(), (std::streamsize)(hdr[4] * sizeof(vlg_wt_node)));
        in.read((char*)smp.data(), (std::streamsize)(hdr[5] * 8));
        if (!in) throw std::runtime_error("truncated VLG index file");
        vlg_index_parts p{hdr[0], (uint32_t)hdr[1], (uint32_t)hdr[2], c2c.data(), C.data(), bv.data(), hdr[3], nodes.data(), (uint32_t)hdr[4],
                          smp.data(), hdr[5]};
        drop_replicas();
        if (m_idx) { vlg_index_destroy(m_idx); m_idx = nullptr; }
        check(vlg_index_from_parts(&p, &m_idx));
    }

    // The whole suffix array resident in HBM: the loaded index is replaced by csa_wt<wt_huff<>, 1, .> over the same BWT
    // (t_dens of csa_wt.hpp:60-72 at its densest; 4 B per character more), made on the device; locate then copies SA intervals instead
    // of walking LF.  Same results.  What serialize() / save_sdsl() wrote before stays the benchmark's t_dens = 32 index.
    void keep_suffix_array()
    {
        vlg_index* dense = nullptr;
        check(vlg_index_resample(m_idx, VLG_SAMPLING_SA_ORDER, 1, &dense));
        drop_replicas();
        vlg_index_destroy(m_idx);
        m_idx = dense;
    }

    // the reference's own on-disk format of csa_wt<wt_huff<>,32,64> (what store_to_file / load_from_file of stock sdsl use)
    void save_sdsl(const std::string& path) const { check(vlg_index_save_sdsl(m_idx, path.c_str())); }
    void load_sdsl(const std::string& path)
    {
        drop_replicas();
        if (m_idx) { vlg_index_destroy(m_idx); m_idx = nullptr; }
        check(vlg_index_load_sdsl(path.c_str(), 32, &m_idx));
    }

    void swap(index_fm_gpu& o)
    {
        std::swap(m_idx, o.m_idx);
        std::swap(m_ws, o.m_ws);
        std::swap(m_replicas, o.m_replicas);
    }

    std::string info(const gapped_pattern&) const { return ""; }
    void prepare(const gapped_pattern&) {}

    // idx.search(pat): one query (index_sasearch.hpp:58-118).  Per-query calls cannot feed a GPU; use search_batch.
    gapped_search_result search(const gapped_pattern& pat) const
    {
        std::vector<gapped_search_result> r = search_batch({pat});
        return r[0];
    }

    // all patterns in one pass of the hot path; results[i].positions = first sub-pattern starts of query i
    std::vector<gapped_search_result> search_batch(const std::vector<gapped_pattern>& pats, int dialect = VLG_DIALECT_BENCHMARK,
                                                   vlg_result_summary* summary = nullptr) const
    {
        ensure_ws();
        std::vector<gapped_search_result> out(pats.size());
        search_range(m_idx, m_ws, pats, 0, pats.size(), dialect, out, summary);
        return out;
    }

    // Contiguous slices of equal estimated work for n shards: the sum of the SA-interval sizes of a query's sub-patterns (0 when one
    // of them does not occur: such a query locates nothing), from one backward-search pass on this index -- what SURVEY.md 8(e)
    // shards by.  Every rank of a multi-process run computes the same cuts from its own replica.  -> cut[n + 1]
    // work of every query as the matcher sees it before it runs: 1 + the sum of its sub-patterns' occurrence counts (0 + 1 for a
    // query with an empty list: nothing of it is located) -- sdsl::count per sub-pattern, one backward-search pass on the device
    std::vector<double> query_weights(const std::vector<gapped_pattern>& pats, int dialect = VLG_DIALECT_BENCHMARK) const
    {
        std::vector<double> w(pats.size(), 1.0);
        if (pats.empty()) return w;
        parsed_batch pb;
        parse(pats, 0, pats.size(), dialect, pb);
        std::vector<uint32_t> k(pats.size() + 1);
        std::vector<uint64_t> occ(vlg_queries_subpatterns(pb.q) + 1);
        check(vlg_queries_k(pb.q, k.data()));
        check(vlg_queries_occurrences(m_idx, pb.q, occ.data(), nullptr));
        size_t s = 0;
        for (size_t i = 0; i < pats.size(); ++i) {
            double sum = 0;
            bool dead = false;
            for (uint32_t j = 0; j < k[i]; ++j) { sum += (double)occ[s + j]; dead |= occ[s + j] == 0; }
            s += k[i];
            w[i] = (dead ? 0.0 : sum) + 1.0;
        }
        return w;
    }

    std::vector<uint64_t> work_cuts(const std::vector<gapped_pattern>& pats, int n, int dialect = VLG_DIALECT_BENCHMARK) const
    {
        std::vector<uint64_t> cut(n + 1, pats.size());
        cut[0] = 0;
        if (n <= 1 || pats.empty()) return cut;
        const std::vector<double> w = query_weights(pats, dialect);
        std::vector<double> cum(pats.size() + 1, 0.0);
        for (size_t i = 0; i < pats.size(); ++i) cum[i + 1] = cum[i] + w[i];
        for (int d = 1; d < n; ++d) {                             // the nearer of the two cuts around the target
            const double target = cum.back() * d / n;
            size_t i = (size_t)(std::lower_bound(cum.begin() + 1, cum.end(), target) - (cum.begin() + 1));
            if (i < pats.size() && cum[i + 1] - target <= target - cum[i]) ++i;
            cut[d] = std::max<uint64_t>(cut[d - 1], std::min<uint64_t>(i, pats.size()));
        }
        return cut;
    }

    // pats[b, e) on the current device (one rank's slice of a sharded batch); the other entries of the result stay empty
    std::vector<gapped_search_result> search_slice(const std::vector<gapped_pattern>& pats, size_t b, size_t e, int dialect = VLG_DIALECT_BENCHMARK,
                                                   vlg_result_summary* summary = nullptr) const
    {
        ensure_ws();
        std::vector<gapped_search_result> out(pats.size());
        if (summary) memset(summary, 0, sizeof *summary);
        if (b < e) search_range(m_idx, m_ws, pats, b, e, dialect, out, summary);
        return out;
    }

    // One process per GPU (RCCL over xGMI): rank `root` holds the index, every other rank of the communicator
    // (vlg_comm_create) receives its image by one broadcast into its own HBM -- `idx.load` on every rank replaced by one load + one
    // collective.  Collective call.
    void broadcast(void* nccl_comm, int root)
    {
        int n = 0, rank = 0;
        check(vlg_comm_info(nccl_comm, &n, &rank));
        if (rank == root) { check(vlg_index_broadcast(m_idx, nccl_comm, root, nullptr, nullptr)); return; }
        drop_replicas();
        if (m_idx) { vlg_index_destroy(m_idx); m_idx = nullptr; }
        check(vlg_index_broadcast(nullptr, nccl_comm, root, nullptr, &m_idx));
    }

    // The query loop of gm_search.cpp:91-121 sharded over the GPUs of the node: the index is replicated once
    // (peer copies over xGMI), the batch is cut into contiguous slices of equal estimated work -- the sum of the SA-interval sizes
    // of a query's sub-patterns, from one backward-search pass -- and one host thread per device searches its slice.  Queries are
    // independent, so nothing is exchanged; results come back in batch order.  A device count above the node's wraps around
    // (several slices on one GPU: only useful to rehearse the path on a smaller machine).
    std::vector<gapped_search_result> search_batch_devices(const std::vector<gapped_pattern>& pats, int n_devices,
                                                           int dialect = VLG_DIALECT_BENCHMARK, std::vector<vlg_result_summary>* summaries = nullptr) const
    {
        int ndev = 0;
        check(vlg_device_count(&ndev));
        if (n_devices < 1 || ndev < 1) throw std::runtime_error("search_batch_devices: no device");
        ensure_ws();
        check(vlg_set_device(0));
        while ((int)m_replicas.size() + 1 < n_devices) {
            replica r{(int)(m_replicas.size() + 1) % ndev, nullptr, nullptr};
            check(vlg_index_replicate(m_idx, r.device, &r.idx));
            m_replicas.push_back(r);
        }
        const std::vector<uint64_t> cut = work_cuts(pats, n_devices, dialect);
        std::vector<gapped_search_result> out(pats.size());
        std::vector<std::string> errors(n_devices);
        if (summaries) summaries->assign(n_devices, vlg_result_summary());
        std::vector<std::thread> threads;
        for (int d = 0; d < n_devices; ++d)
            threads.emplace_back([&, d]() {
                try {
                    const int device = d == 0 ? 0 : m_replicas[d - 1].device;
                    check(vlg_set_device(device));                 // per host thread
                    vlg_workspace*& ws = d == 0 ? m_ws : m_replicas[d - 1].ws;
                    if (!ws) check(vlg_workspace_create(0, nullptr, &ws));
                    search_range(d == 0 ? m_idx : m_replicas[d - 1].idx, ws, pats, cut[d], cut[d + 1], dialect, out,
                                 summaries ? &(*summaries)[d] : nullptr);
                } catch (const std::exception& e) { errors[d] = e.what(); }
            });
        for (auto& t : threads) t.join();
        for (int d = 0; d < n_devices; ++d) if (!errors[d].empty()) throw std::runtime_error("device slice " + std::to_string(d) + ": " + errors[d]);
        return out;
    }

    // sdsl::locate(idx, query) (include/sdsl/vlg_index.hpp:395-401): every sub-pattern position of every match
    std::vector<std::vector<uint64_t>> locate(const std::string& query) const
    {
        ensure_ws();
        check(vlg_workspace_set_option(m_ws, "tuples", 1));
        uint64_t off[2] = {0, query.size()};
        vlg_queries* q = nullptr;
        check(vlg_queries_parse(query.data(), off, 1, VLG_DIALECT_LIBRARY, nullptr, &q));
        uint32_t k = 0;
        check(vlg_queries_k(q, &k));
        vlg_result* r = nullptr;
        vlg_status st = vlg_search_batch(m_idx, q, m_ws, &r);
        vlg_queries_destroy(q);
        check(st);
        vlg_result_summary s;
        check(vlg_result_summary_get(r, &s));
        std::vector<uint64_t> tuples(s.n_tuple_values + 1);
        st = vlg_result_fetch(r, nullptr, nullptr, nullptr, tuples.data());
        vlg_result_destroy(r);
        check(st);
        std::vector<std::vector<uint64_t>> out(s.n_matches);
        for (uint64_t m = 0; m < s.n_matches; ++m) out[m].assign(tuples.begin() + m * k, tuples.begin() + (m + 1) * k);
        return out;
    }
    uint64_t count(const std::string& query) const { return locate(query).size(); }

    vlg_index* handle() const { return m_idx; }
};

}  // namespace vlg_host
