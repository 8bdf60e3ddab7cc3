// gm_search equivalent for the GPU index (benchmark/gapped-matching/src/gm_search.cpp): same command line (-c, -p),
// same machine-readable "# key = value" lines on stdout.  Queries are searched as ONE batch; the per-query "TIMING"
// and quartile lines therefore report the batch time divided by the number of patterns (-1 runs query by query, with real
// per-query times).  -g N shards the pattern file over N GPUs of the node (index replicated, no exchange between the slices).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <unistd.h>
#include "index_fm_gpu.hpp"

using namespace vlg_host;
using namespace std::chrono;

static std::vector<gapped_pattern> parse_pattern_file(const std::string& file)
{
    std::vector<gapped_pattern> pats;
    std::ifstream in(file);
    if (!in) { std::cerr << "Cannot open pattern file '" << file << "'\n"; return pats; }
    std::string line;
    while (std::getline(in, line)) {
        try { pats.emplace_back(line, true); }
        catch (...) { std::cerr << "Could not parse pattern '" << line << "'. Skipped\n"; }   // utils.hpp:94-99
    }
    return pats;
}

int main(int argc, char* const argv[])
{
    std::string col_dir, pat_file;
    bool one_by_one = false;
    int op, n_gpus = 1;
    while ((op = getopt(argc, argv, "c:p:1g:")) != -1) {
        if (op == 'c') col_dir = optarg;
        else if (op == 'p') pat_file = optarg;
        else if (op == '1') one_by_one = true;
        else if (op == 'g') n_gpus = atoi(optarg);
    }
    if (col_dir.empty() || pat_file.empty() || n_gpus < 1) {
        fprintf(stdout, "%s -c <collection directory> -p <pattern file> [-1] [-g <GPUs>]\n", argv[0]);
        return EXIT_FAILURE;
    }
    try {
        collection col(col_dir);
        index_fm_gpu idx;
        std::string index_file = col.path + "/index/index-" + idx.name() + ".vlg";
        auto t_load = high_resolution_clock::now();
        std::ifstream ifs(index_file, std::ios::binary);
        if (ifs.is_open()) idx.load(ifs);
        else { index_fm_gpu built(col); idx.swap(built); }
        auto load_us = duration_cast<microseconds>(high_resolution_clock::now() - t_load).count();
        std::vector<gapped_pattern> pats = parse_pattern_file(pat_file);
        size_t num_results = 0, checksum = 0;
        std::vector<long long> timings;
        auto t0 = high_resolution_clock::now();
        if (one_by_one) {
            for (auto& p : pats) {
                auto a = high_resolution_clock::now();
                auto r = idx.search(p);
                timings.push_back(duration_cast<microseconds>(high_resolution_clock::now() - a).count());
                for (auto pos : r.positions) { checksum += pos; num_results++; }
            }
        } else {
            const index_fm_gpu& cidx = idx;                       // search is const, as in the reference's bench_index
            auto res = n_gpus > 1 ? cidx.search_batch_devices(pats, n_gpus) : cidx.search_batch(pats);
            for (auto& r : res) for (auto pos : r.positions) { checksum += pos; num_results++; }
        }
        long long total_us = duration_cast<microseconds>(high_resolution_clock::now() - t0).count();
        if (!one_by_one) timings.assign(pats.size(), pats.empty() ? 0 : total_us / (long long)pats.size());
        for (auto t : timings) std::cout << "TIMING = " << t << std::endl;
        std::sort(timings.begin(), timings.end());
        auto q = [&](double f) { return timings.empty() ? 0LL : timings[std::min(timings.size() - 1, (size_t)(f * timings.size()))]; };
        std::cout << "# info =" << std::endl;
        std::cout << "# num_results = " << num_results << std::endl;
        std::cout << "# checksum = " << checksum << std::endl;
        std::cout << "# total_time_mus = " << total_us << std::endl;
        std::cout << "# min_time_mus = " << q(0.0) << std::endl;
        std::cout << "# qrt_1st_time_mus = " << q(0.25) << std::endl;
        std::cout << "# mean_time_mus = " << (timings.empty() ? 0 : total_us / (long long)timings.size()) << std::endl;
        std::cout << "# median_time_mus = " << q(0.5) << std::endl;
        std::cout << "# qrt_3rd_time_mus = " << q(0.75) << std::endl;
        std::cout << "# max_time_mus = " << q(1.0) << std::endl;
        for (const char* k : {"total", "min", "qrt_1st", "mean", "median", "qrt_3rd", "max"}) std::cout << "# prep_" << k << "_time_mus = 0" << std::endl;
        std::cout << "# load_time_mus = " << load_us << std::endl;
        std::cout << "# num_patterns = " << pats.size() << std::endl;
        std::cout << "# num_gpus = " << n_gpus << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return 0;
}
