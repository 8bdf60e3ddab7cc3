// The build's own examples/vlg_matching.cpp (reference: examples/vlg_matching.cpp:6-47): same output format, index on
// the GPU.  Without arguments it indexes "abracadabrasimsalabim" and runs the reference's three byte-alphabet queries;
// with arguments: vlg_matching_example <text file> <query> [<query> ...].
#include <iostream>
#include <iterator>
#include "index_fm_gpu.hpp"

using namespace vlg_host;

static void dump_query_results(index_fm_gpu& idx, const std::string& qry)
{
    std::cout << std::endl;
    auto res = idx.locate(qry);
    std::cout << "count(" << qry << ")=" << res.size() << std::endl;
    std::cout << "locate(" << qry << ")=" << std::endl;
    size_t occ = 1;
    for (auto& t : res) {
        std::cout << "  " << occ++ << ". occ starting at position " << t[0] << std::endl;
        std::cout << "     Subpattern positions:";
        for (auto p : t) std::cout << " " << p;
        std::cout << std::endl;
    }
}

int main(int argc, char* argv[])
{
    try {
        if (argc <= 1) {
            std::string t = "abracadabrasimsalabim";
            index_fm_gpu idx(std::vector<uint8_t>(t.begin(), t.end()));
            dump_query_results(idx, "ac.{2,5}?a.{4,8}?b");
            dump_query_results(idx, "a.{0,10}?a.{0,10}?a");
            dump_query_results(idx, "foo.{0,10}?bar");
        } else {
            std::ifstream in(argv[1], std::ios::binary);
            if (!in) throw std::runtime_error(std::string("cannot open ") + argv[1]);
            std::vector<uint8_t> text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            index_fm_gpu idx(text);
            for (int i = 2; i < argc; ++i) dump_query_results(idx, argv[i]);
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
