// gm_index equivalent (benchmark/gapped-matching/src/gm_index.cpp:38-60): build the index of a collection on the GPU and
// store it under <col>/index/.  With -i <raw file> it first creates the collection like create_collection.cpp:70-93
// (bytes 0, \n, \r, \f become spaces; text.TEXT is a bit-compressed int_vector<0>).  With -s the index is also stored as
// <col>/index/index-<name>.sdsl in the reference's own format of csa_wt<wt_huff<>,32,64>: sdsl::load_from_file picks it up.
#include <cstdio>
#include <iostream>
#include <sys/stat.h>
#include <unistd.h>
#include "index_fm_gpu.hpp"

using namespace vlg_host;

int main(int argc, char* const argv[])
{
    std::string col_dir, raw;
    bool sdsl_too = false;
    int op;
    while ((op = getopt(argc, argv, "c:i:s")) != -1) {
        if (op == 'c') col_dir = optarg;
        else if (op == 'i') raw = optarg;
        else if (op == 's') sdsl_too = true;
    }
    if (col_dir.empty()) { fprintf(stdout, "%s -c <collection directory> [-i <raw text file>] [-s]\n", argv[0]); return EXIT_FAILURE; }
    try {
        if (!raw.empty()) {
            mkdir(col_dir.c_str(), 0755);
            std::ifstream in(raw, std::ios::binary);
            if (!in) throw std::runtime_error("cannot open " + raw);
            std::vector<uint8_t> text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            for (auto& c : text) if (c == 0 || c == '\n' || c == '\r' || c == '\f') c = ' ';      // create_collection.cpp:87-90
            write_text_file(col_dir + "/text.TEXT", text);
        }
        collection col(col_dir);
        mkdir((col.path + "/index").c_str(), 0755);
        index_fm_gpu idx(col);
        std::string out_file = col.path + "/index/index-" + idx.name() + ".vlg";
        std::ofstream ofs(out_file, std::ios::binary);
        auto bytes = idx.serialize(ofs);
        std::cout << "# index_file = " << out_file << std::endl;
        std::cout << "# index_bytes = " << bytes << std::endl;
        if (sdsl_too) {
            const std::string sdsl_file = col.path + "/index/index-" + idx.name() + ".sdsl";
            idx.save_sdsl(sdsl_file);
            std::cout << "# sdsl_file = " << sdsl_file << std::endl;
        }
        vlg_index_info info;
        check(vlg_index_get_info(idx.handle(), &info));
        std::cout << "# n = " << info.n << std::endl << "# sigma = " << info.sigma << std::endl << "# hbm_bytes = " << info.hbm_bytes << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return 0;
}
