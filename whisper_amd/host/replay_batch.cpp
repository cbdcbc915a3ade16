// replay_batch.cpp - drives the C++ mirror classes (gpu_verifier.hpp) the way the reference's
// loops would: per read preprocess(), per candidate dynamicProgramming()/LevDiag(), one flush.
// Input / output are flat binary files written and checked by tests/test_host_mirror.py.
//   replay_batch <in.bin> <out.bin>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "gpu_verifier.hpp"

using namespace whisper_amd;

template <class T> static std::vector<T> read_vec(std::ifstream& f)
{
    uint64_t n = 0;
    f.read((char*)&n, 8);
    std::vector<T> v(n);
    f.read((char*)v.data(), (std::streamsize)(n * sizeof(T)));
    return v;
}

int main(int argc, char** argv)
{
    if (argc != 3 && !(argc == 4 && std::string(argv[3]) == "--accepted-only")) {
        std::cerr << "usage: replay_batch <in.bin> <out.bin> [--accepted-only]\n";
        return 2;
    }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        uint32_t hdr[4];   // ref_size, cur_offset, read_len, n_reads
        f.read((char*)hdr, sizeof(hdr));
        auto genome = read_vec<uchar_t>(f);
        auto reads = read_vec<uchar_t>(f);          // n_reads x ceil(read_len / 2), nibble-packed
        auto dist = read_vec<wamd_dist_task>(f);    // sorted by read
        auto lev = read_vec<wamd_levdiag_task>(f);  // sorted by read
        const uint32_t stride = (hdr[2] + 1) / 2;

        GpuReference ref(0, genome.data(), genome.size(), hdr[0], hdr[1]);
        if (argc == 4)   // what a mapping stage asks for: rejected candidates as {k + 1, 0}
            ref.setAcceptedOnly(true);
        GpuLevMyers lm(ref);
        std::vector<size_t> dist_tickets(dist.size()), lev_tickets(lev.size());
        size_t di = 0, li = 0;
        for (uint32_t r = 0; r < hdr[3]; ++r) {
            lm.preprocess(reads.data() + (size_t)r * stride, hdr[2], genome_t::direct);
            for (; di < dist.size() && dist[di].read_idx == r; ++di)
                dist_tickets[di] = lm.dynamicProgramming(dist[di].ref_pos, dist[di].max_distance_in_ref, dist[di].max_ed);
            for (; li < lev.size() && lev[li].read_idx == r; ++li)
                lev_tickets[li] = lm.LevDiag(lev[li].position_in_text, lev[li].fixed_left, lev[li].fixed_right,
                                             lev[li].max_mismatches, lev[li].dir_gen_flag != 0);
        }
        lm.flush();
        std::ofstream o(argv[2], std::ios::binary);
        for (size_t i = 0; i < dist.size(); ++i) {
            ref_pos_t pos;
            uint32_t ed;
            lm.result(dist_tickets[i], pos, ed);
            o.write((char*)&ed, 4);
            o.write((char*)&pos, 4);
        }
        for (size_t i = 0; i < lev.size(); ++i) {
            uint32_t e = lm.levDiagResult(lev_tickets[i]);
            o.write((char*)&e, 4);
        }
        std::cout << "replayed " << dist.size() << " dynamicProgramming and " << lev.size() << " LevDiag calls\n";
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
