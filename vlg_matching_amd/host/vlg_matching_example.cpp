// Example driver of the GPU index with the reference's library surface (construct_im / count / locate and the lazy iterator of
// vlg_index_gpu.hpp).  It prints what the reference's example prints for the same inputs (examples/vlg_matching.cpp:10-22 defines
// the format), so its output can be diffed against a run of the reference:
//     vlg_matching_example                         the example text, byte alphabet and integer alphabet, three queries each
//     vlg_matching_example <file> <query>...       a byte text from a file, queries from the command line
#include <cstdio>
#include <string>
#include <vector>
#include "vlg_index_gpu.hpp"

namespace vh = vlg_host;

// one query: the count line, then every match with the positions of all its sub-patterns
template <class Index>
static void report(const Index& index, const std::string& query)
{
    std::printf("\ncount(%s)=%llu\n", query.c_str(), (unsigned long long)vh::count(index, query));
    std::printf("locate(%s)=\n", query.c_str());
    auto matches = vh::locate(index, query);
    unsigned long long number = 0;
    for (auto m = matches.begin(); m != matches.end(); ++m) {
        std::printf("  %llu. occ starting at position %llu\n", ++number, (unsigned long long)*m);
        std::string line = "     Subpattern positions:";
        for (size_t s = 0; s < m.size(); ++s) line += " " + std::to_string(m[(int)s]);
        std::puts(line.c_str());
    }
}

int main(int argc, char* argv[])
{
    try {
        if (argc >= 2) {                                    // text from a file
            vh::vlg_index_gpu<vh::byte_alphabet_tag> from_file;
            vh::construct(from_file, argv[1], 1);
            for (int a = 2; a < argc; ++a) report(from_file, argv[a]);
            return 0;
        }
        const std::string sample = "abracadabrasimsalabim";
        vh::vlg_index_gpu<vh::byte_alphabet_tag> bytes;
        vh::construct_im(bytes, sample, 1);
        for (const char* q : {"ac.{2,5}?a.{4,8}?b", "a.{0,10}?a.{0,10}?a", "foo.{0,10}?bar"}) report(bytes, q);
        // the same text as integers: sub-patterns are whitespace-separated decimals (vlg_index.hpp:63-69)
        vh::vlg_index_gpu<vh::int_alphabet_tag> ints;
        vh::construct_im(ints, std::vector<uint64_t>(sample.begin(), sample.end()));
        for (const char* q : {"97 99 .{2,5}? 97 .{4,8}? 98", "97 .{0,10}? 97 .{0,10}? 97", "1337 .{0,10}? 42"}) report(ints, q);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
