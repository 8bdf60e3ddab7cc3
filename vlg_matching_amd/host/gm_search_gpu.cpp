// gm_search equivalent for the GPU index (benchmark/gapped-matching/src/gm_search.cpp): same command line (-c, -p),
// same machine-readable "# key = value" lines on stdout.  Queries are searched as ONE batch, so no clock can be put around a
// single query as gm_search.cpp:96-107 does: the per-query "TIMING" and quartile lines (gm_search.cpp:142-160) carry the batch
// time APPORTIONED BY WORK -- query i gets total x w_i / sum(w), w = 1 + the sum of its sub-patterns' occurrence counts, what
// the matcher locates and joins for it -- and `# timing_mode` says so; -1 runs query by query with real per-query clocks.  -g N shards the pattern file over N GPUs of the node (index replicated, no exchange between the slices):
// by default one process drives all of them (a host thread per device, replicas by peer copies); -g N -P starts ONE PROCESS PER
// GPU instead -- rank 0 loads the index, its image goes to the others by one RCCL broadcast (vlg_index_broadcast), every rank
// searches its slice and the counters are all-reduced.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sys/wait.h>
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

static void print_summary(std::vector<long long> timings, size_t num_results, size_t checksum, long long total_us, long long load_us, size_t n_pat, int n_gpus,
                          const char* mode, const char* timing_mode);

// the batch's time shared out over its queries by their work (index_fm_gpu::query_weights)
static std::vector<long long> apportion(const index_fm_gpu& idx, const std::vector<gapped_pattern>& pats, long long total_us)
{
    std::vector<long long> t(pats.size(), 0);
    if (pats.empty()) return t;
    const std::vector<double> w = idx.query_weights(pats);
    double sum = 0;
    for (double x : w) sum += x;
    for (size_t i = 0; i < pats.size(); ++i) t[i] = (long long)((double)total_us * w[i] / sum);
    return t;
}

// One rank of `-g N -P`: nothing here has touched a GPU before the fork in main().  The communicator's id travels from rank 0 to
// every other rank through a pipe the parent made before forking (id_fd: rank 0 holds the write ends of all of them, rank r the
// read end of its own) -- nothing on the file system, nothing another user could pre-create or a recycled pid could leave behind.
static bool write_all(int fd, const char* p, size_t n) { while (n) { const ssize_t w = write(fd, p, n); if (w <= 0) return false; p += w; n -= (size_t)w; } return true; }
static bool read_all(int fd, char* p, size_t n) { while (n) { const ssize_t r = read(fd, p, n); if (r <= 0) return false; p += r; n -= (size_t)r; } return true; }

static int run_rank(const std::string& col_dir, const std::string& pat_file, int rank, int n_ranks, const std::vector<int>& id_fd, bool keep_sa)
{
    try {
        int ndev = 0;
        check(vlg_device_count(&ndev));
        check(vlg_set_device(rank % ndev));
        vlg_comm_id id;
        if (rank == 0) {
            check(vlg_comm_unique_id(&id));
            for (int r = 1; r < n_ranks; ++r) {
                if (!write_all(id_fd[r], id.bytes, sizeof id.bytes)) throw std::runtime_error("cannot hand the communicator id to rank " + std::to_string(r));
                close(id_fd[r]);
            }
        } else {
            if (!read_all(id_fd[rank], id.bytes, sizeof id.bytes)) throw std::runtime_error("rank 0 never sent the communicator id");   // (EOF: rank 0 died)
            close(id_fd[rank]);
        }
        void* comm = nullptr;
        check(vlg_comm_create(&id, n_ranks, rank, &comm));
        collection col(col_dir);
        index_fm_gpu idx;
        auto t_load = high_resolution_clock::now();
        if (rank == 0) {                                             // one load (or build), then one broadcast of the HBM image
            std::ifstream ifs(col.path + "/index/index-" + idx.name() + ".vlg", std::ios::binary);
            if (ifs.is_open()) idx.load(ifs);
            else { index_fm_gpu built(col); idx.swap(built); }
        }
        idx.broadcast(comm, 0);
        if (keep_sa) idx.keep_suffix_array();                        // (every rank expands its own copy: cheaper than 5x the broadcast)
        auto load_us = duration_cast<microseconds>(high_resolution_clock::now() - t_load).count();
        std::vector<gapped_pattern> pats = parse_pattern_file(pat_file);
        const index_fm_gpu& cidx = idx;
        uint64_t sums[3] = {0, 0, 0};                                // num_results, checksum (mod 2^64), slowest... time is max, below
        auto t0 = high_resolution_clock::now();
        const std::vector<uint64_t> cut = cidx.work_cuts(pats, n_ranks);              // every rank computes the same cuts from its replica
        auto res = cidx.search_slice(pats, cut[rank], cut[rank + 1]);
        for (auto& r : res) for (auto pos : r.positions) { sums[1] += pos; sums[0]++; }
        long long my_us = duration_cast<microseconds>(high_resolution_clock::now() - t0).count();
        check(vlg_comm_allreduce_sum_u64(comm, sums, 2, nullptr));
        // the batch time is the slowest rank's: ranks publish theirs in a vector that is summed (one non-zero entry each)
        std::vector<uint64_t> times(n_ranks, 0);
        times[rank] = (uint64_t)my_us;
        check(vlg_comm_allreduce_sum_u64(comm, times.data(), (uint32_t)n_ranks, nullptr));
        const long long total_us = (long long)*std::max_element(times.begin(), times.end());
        if (rank == 0) {
            print_summary(apportion(cidx, pats, total_us), sums[0], sums[1], total_us, load_us, pats.size(), n_ranks, "processes", "batch_apportioned_by_occurrences");
        }
        vlg_comm_destroy(comm);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "rank " << rank << ": error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}

int main(int argc, char* const argv[])
{
    std::string col_dir, pat_file;
    bool one_by_one = false, per_process = false, keep_sa = false;
    int op, n_gpus = 1;
    while ((op = getopt(argc, argv, "c:p:1g:PS")) != -1) {
        if (op == 'c') col_dir = optarg;
        else if (op == 'p') pat_file = optarg;
        else if (op == '1') one_by_one = true;
        else if (op == 'g') n_gpus = atoi(optarg);
        else if (op == 'P') per_process = true;
        else if (op == 'S') keep_sa = true;                          // the whole suffix array resident in HBM (index_fm_gpu::keep_suffix_array)
    }
    if (col_dir.empty() || pat_file.empty() || n_gpus < 1) {
        fprintf(stdout, "%s -c <collection directory> -p <pattern file> [-1] [-g <GPUs> [-P]] [-S]\n", argv[0]);
        return EXIT_FAILURE;
    }
    if (per_process) {
        // fork BEFORE anything initialises the GPU (no HIP call has been made: the C-ABI is only touched inside the children).
        // Not under a profiler whose preloaded library initialises the GPU before main() (rocprofv3 --pmc does): profile -g N
        // without -P, or a single rank.
        std::vector<int> rd(n_gpus, -1), wr(n_gpus, -1);             // pipe r carries the communicator id from rank 0 to rank r
        for (int r = 1; r < n_gpus; ++r) {
            int fds[2];
            if (pipe(fds) != 0) { perror("pipe"); return EXIT_FAILURE; }
            rd[r] = fds[0]; wr[r] = fds[1];
        }
        std::vector<pid_t> kids;
        for (int r = 0; r < n_gpus; ++r) {
            const pid_t pid = fork();
            if (pid < 0) { perror("fork"); return EXIT_FAILURE; }
            if (pid == 0) {
                // rank 0 keeps the write ends, rank r its own read end; every other descriptor is closed so that a dead rank 0 reads as EOF
                for (int j = 1; j < n_gpus; ++j) {
                    if (r != 0) close(wr[j]);
                    if (j != r) close(rd[j]);
                }
                _exit(run_rank(col_dir, pat_file, r, n_gpus, r == 0 ? wr : rd, keep_sa));
            }
            kids.push_back(pid);
        }
        for (int r = 1; r < n_gpus; ++r) { close(rd[r]); close(wr[r]); }
        int rc = 0;
        size_t left = kids.size();
        while (left) {                                               // any rank that fails ends the others (they would wait in a collective)
            int status = 0;
            const pid_t done = wait(&status);
            if (done < 0) break;
            --left;
            const int code = WIFEXITED(status) ? WEXITSTATUS(status) : EXIT_FAILURE;
            if (code && !rc) { rc = code; for (pid_t k : kids) if (k != done) kill(k, SIGTERM); }
        }
        return rc;
    }
    try {
        collection col(col_dir);
        index_fm_gpu idx;
        std::string index_file = col.path + "/index/index-" + idx.name() + ".vlg";
        auto t_load = high_resolution_clock::now();
        std::ifstream ifs(index_file, std::ios::binary);
        if (ifs.is_open()) idx.load(ifs);
        else { index_fm_gpu built(col); idx.swap(built); }
        if (keep_sa) idx.keep_suffix_array();
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
        if (!one_by_one) timings = apportion(idx, pats, total_us);       // (after the clock has stopped: one more backward-search pass)
        print_summary(timings, num_results, checksum, total_us, load_us, pats.size(), n_gpus, n_gpus > 1 ? "threads" : "single",
                      one_by_one ? "per_query_clock" : "batch_apportioned_by_occurrences");
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return 0;
}

static void print_summary(std::vector<long long> timings, size_t num_results, size_t checksum, long long total_us, long long load_us, size_t n_pat, int n_gpus,
                          const char* mode, const char* timing_mode)
{
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
    std::cout << "# num_patterns = " << n_pat << std::endl;
    std::cout << "# num_gpus = " << n_gpus << std::endl;
    std::cout << "# gpu_mode = " << mode << std::endl;
    std::cout << "# timing_mode = " << timing_mode << std::endl;
}
