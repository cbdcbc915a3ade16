// replay_batch.cpp - drives the C++ mirror classes (gpu_verifier.hpp) the way the reference's
// loops would: per read preprocess(), per candidate dynamicProgramming()/LevDiag(), one flush.
// Input / output are flat binary files written and checked by tests/test_host_mirror.py.
//   replay_batch <in.bin> <out.bin> [--accepted-only]
//   replay_batch --rescue <in.bin> <out.bin>
//   replay_batch --mates <in.bin> <out.bin>           find_pairs<LevMyers>'s chain per hit through GpuMateRescue
//   replay_batch --bin <in.bin> <out.bin> [repeats]   one bin of one stage through GpuSaBin: per read addRead(), per
//                                           suffix-array entry candidate(), ONE flush; prints host-to-host candidates/s
#include <chrono>
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

// replay_batch --rescue <in.bin> <out.bin>: the sensitive-mode matcher and the hashing rescue through their
// mirrors, one Preprocess / preprocessRawSeq per read and one Match / match per candidate, one flush each.
static int replay_rescue(const char* in, const char* out)
{
    std::ifstream f(in, std::ios::binary);
    uint32_t hdr[4];   // ref_size, cur_offset, cigar stride of the rescue, cigar stride of GetExtCigarNew
    f.read((char*)hdr, sizeof(hdr));
    auto genome = read_vec<uchar_t>(f);
    auto ip = read_vec<wamd_indel_params>(f);
    auto iblob = read_vec<uchar_t>(f);           // nibble-packed reads of the matcher
    auto ioff = read_vec<uint64_t>(f);           // n + 1 offsets
    auto ilen = read_vec<uint32_t>(f);
    auto itasks = read_vec<wamd_indel_task>(f);  // sorted by read
    auto sp = read_vec<wamd_softclip_params>(f);
    auto sblob = read_vec<uchar_t>(f);           // ASCII reads of the rescue
    auto soff = read_vec<uint64_t>(f);
    auto slen = read_vec<uint32_t>(f);
    auto stasks = read_vec<wamd_softclip_task>(f);

    GpuReference ref(0, genome.data(), genome.size(), hdr[0], hdr[1]);
    ref.setMaxTextLen(16384);   // sam.cpp:190
    GpuIndelMatching im(ref, ip.at(0));
    std::vector<size_t> it(itasks.size());
    size_t k = 0;
    for (uint32_t r = 0; r < ilen.size(); ++r) {
        im.Preprocess(iblob.data() + ioff[r], ilen[r], genome_t::direct);
        for (; k < itasks.size() && itasks[k].read_idx == r; ++k)
            it[k] = im.Match(itasks[k].ref_pos, itasks[k].max_indel_size, itasks[k].fixed_segment_pos,
                             itasks[k].fixed_segment_size, itasks[k].max_no_mismatches);
    }
    im.flush();
    // GetExtCigarNew of what Match found, as fill_cigar_with_indel_matching asks for it (sam_matching.cpp:824-852): the
    // read as ASCII through PreprocessRawSeq, the record, the window start
    GpuIndelExtCigar ic(ref, hdr[3] - 1);
    std::vector<size_t> ict;
    {
        std::vector<uchar_t> ascii;
        for (size_t i = 0; i < itasks.size(); ++i) {
            const wamd_mapping& mp = im.result(it[i]);
            if (!mp.found)
                continue;
            const uint32_t r = itasks[i].read_idx;
            ascii.resize(ilen[r]);
            for (uint32_t p = 0; p < ilen[r]; ++p) {
                const uchar_t b = iblob[ioff[r] + (p >> 1)];
                const uint32_t c = (p & 1u) ? (b & 15u) : (b >> 4);
                ascii[p] = (uchar_t)"ACGTN"[c < 4 ? c : 4];
            }
            ic.PreprocessRawSeq(ascii.data(), ilen[r], genome_t::direct);
            ict.push_back(ic.GetExtCigarNew(mp, mp.pos));
        }
    }
    ic.flush();
    GpuSoftClipping sc(ref, sp.at(0), hdr[2] - 1);
    std::vector<size_t> st(stasks.size());
    k = 0;
    for (uint32_t r = 0; r < slen.size(); ++r) {
        // the task's flag is the orientation the reference passes to preprocessRawSeq (sam_matching.cpp:556)
        for (; k < stasks.size() && stasks[k].read_idx == r; ++k) {
            sc.preprocessRawSeq(sblob.data() + soff[r], slen[r],
                                (stasks[k].flags & WAMD_TASK_FLAG_REV_COMP) ? genome_t::rev_comp : genome_t::direct);
            st[k] = sc.match(stasks[k].ref_pos, stasks[k].max_distance_in_ref, stasks[k].min_exact_len);
        }
    }
    sc.flush();
    std::ofstream o(out, std::ios::binary);
    for (size_t i = 0; i < itasks.size(); ++i)
        o.write((const char*)&im.result(it[i]), sizeof(wamd_mapping));
    for (size_t i = 0; i < stasks.size(); ++i)
        o.write((const char*)&sc.result(st[i]), sizeof(wamd_softclip_result));
    for (size_t i = 0; i < stasks.size(); ++i)
        o.write((const char*)sc.extCigar(st[i]), hdr[2]);
    for (size_t t : ict)
        o.write((const char*)&ic.result(t), sizeof(wamd_indel_cigar_result));
    for (size_t t : ict)
        o.write((const char*)ic.extCigar(t), hdr[3]);
    std::cout << "replayed " << itasks.size() << " Match and " << stasks.size() << " match calls\n";
    return 0;
}

// replay_batch --mates <in.bin> <out.bin>: the hits of find_pairs<LevMyers> (sam_matching.cpp:379-457) through
// GpuMateRescue - addMate per read, hit per task with the chromosome table as CRefSeqDesc::GetDescription() hands it
// out (the five numeric fields of seq_desc_t), one flush.
struct SeqDesc {   // the fields of seq_desc_t (common/types.h:145-153) the mirror reads
    uint64_t size, pos_in_ref_seq, no_initial_Ns, no_final_Ns, no_starting_Ns;
};
struct MateHit {
    uint32_t mate, hit_pos, distance_threshold, max_mate_edit_distance, hit_num_events;
    int32_t ref_seq_id;
    uint32_t paired_dir_direct, reserved;
};
static int replay_mates(const char* in, const char* out)
{
    std::ifstream f(in, std::ios::binary);
    uint32_t hdr[4];   // ref_size, cur_offset, ext-CIGAR stride, SEPARATE_N_LEN
    f.read((char*)hdr, sizeof(hdr));
    auto genome = read_vec<uchar_t>(f);
    auto prm = read_vec<wamd_rescue_params>(f);
    auto sc = read_vec<wamd_scoring>(f);
    auto desc = read_vec<SeqDesc>(f);
    auto blob = read_vec<uchar_t>(f);            // ASCII mates
    auto off = read_vec<uint64_t>(f);
    auto len = read_vec<uint32_t>(f);
    auto hits = read_vec<MateHit>(f);
    GpuReference ref(0, genome.data(), genome.size(), hdr[0], hdr[1]);
    ref.setMaxTextLen(16384);   // sam.cpp:190
    check(wamd_ctx_set_scoring(ref.ctx(), &sc.at(0)), "wamd_ctx_set_scoring");
    GpuMateRescue rescue(ref, hdr[2] - 1, prm.at(0).enable_short_indel_refinement != 0, prm.at(0).enable_boundary_clipping != 0);
    for (uint32_t r = 0; r < len.size(); ++r)
        rescue.addMate(blob.data() + off[r], len[r]);
    std::vector<size_t> t(hits.size());
    for (size_t i = 0; i < hits.size(); ++i)
        t[i] = rescue.hit(hits[i].mate, hits[i].hit_pos, hits[i].distance_threshold, hits[i].max_mate_edit_distance,
                          hits[i].paired_dir_direct ? genome_t::direct : genome_t::rev_comp, hits[i].hit_num_events, desc,
                          hits[i].ref_seq_id, hdr[3]);
    rescue.flush();
    std::ofstream o(out, std::ios::binary);
    for (size_t i = 0; i < hits.size(); ++i)
        o.write((const char*)&rescue.result(t[i]), sizeof(wamd_rescue_result));
    for (size_t i = 0; i < hits.size(); ++i)
        o.write((const char*)rescue.extCigar(t[i]), hdr[2]);
    std::cout << "replayed " << hits.size() << " hits\n";
    return 0;
}

// reverse_read (mapping_core.cpp:1063-1108): the reverse complement of a nibble-packed read, first symbol in the high nibble
static void reverse_read(const uchar_t* data, uint32_t size, uchar_t* rc)
{
    static uchar_t lut[256];
    static bool have = false;
    if (!have) {
        auto comp = [](uint32_t c) -> uint32_t { return c < 4 ? 3 - c : c; };
        for (uint32_t b = 0; b < 256; ++b)
            lut[b] = (uchar_t)(comp(b & 15u) << 4 | comp(b >> 4));
        have = true;
    }
    const uint32_t nb = (size + 1) / 2;
    if ((size & 1u) == 0) {      // whole bytes: the last byte, nibbles swapped and complemented, comes first
        for (uint32_t i = 0; i < nb; ++i)
            rc[i] = lut[data[nb - 1 - i]];
        return;
    }
    auto sym = [&](uint32_t i) -> uint32_t { return (i & 1u) ? (data[i >> 1] & 15u) : (data[i >> 1] >> 4); };
    memset(rc, 0, nb);
    for (uint32_t i = 0; i < size; ++i) {
        const uint32_t c = sym(size - 1 - i);
        const uint32_t r = c < 4 ? 3 - c : c;
        rc[i >> 1] |= (uchar_t)((i & 1u) ? r : r << 4);
    }
}

// replay_batch --bin: the per-bin form of INTEGRATION.md section 3 on a synthetic bin.  The input holds what the
// bin loop and the suffix-array loops would have at hand: the reads, per read its max_mismatches and how many
// suffix-array entries it has on either strand, the entries.  Enqueue as the loops would, flush once, write the
// answers in enqueue order.
static int replay_bin(const char* in, const char* out, int repeats)
{
    std::ifstream f(in, std::ios::binary);
    uint32_t hdr[8];   // ref_size, cur_offset, n_reads, start_pos, end_pos, min_read_len, lev_alg_thr, bytes per read
    f.read((char*)hdr, sizeof(hdr));
    auto genome = read_vec<uchar_t>(f);
    auto reads = read_vec<uchar_t>(f);       // n_reads x hdr[7], nibble-packed
    auto lens = read_vec<uint32_t>(f);
    auto read_k = read_vec<uint8_t>(f);
    auto counts = read_vec<uint16_t>(f);     // direct, other strand per read
    auto sa = read_vec<uint32_t>(f);
    const uint32_t n = hdr[2], stride = hdr[7];

    GpuReference ref(0, genome.data(), genome.size(), hdr[0], hdr[1]);
    GpuSaBin bin(ref, hdr[6]);
    std::vector<uchar_t> rc(stride + 1);
    std::vector<uint8_t> answers(sa.size());
    double best = 0;
    for (int rep = 0; rep < repeats; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        bin.beginBin(hdr[3], hdr[4], hdr[5]);
        size_t c = 0;
        for (uint32_t r = 0; r < n; ++r) {
            const uchar_t* data = reads.data() + (size_t)r * stride;
            reverse_read(data, lens[r], rc.data());
            bin.addRead(data, rc.data(), lens[r], read_k[r]);
            for (uint32_t j = 0; j < counts[2 * r]; ++j)
                bin.candidate(sa[c++], true);
            for (uint32_t j = 0; j < counts[2 * r + 1]; ++j)
                bin.candidate(sa[c++], false);
        }
        const auto t1 = std::chrono::steady_clock::now();
        bin.flush();
        const auto t2 = std::chrono::steady_clock::now();
        c = 0;
        for (uint32_t r = 0; r < n; ++r)
            for (int strand = 1; strand >= 0; --strand)
                for (uint32_t j = 0; j < bin.count(r, strand != 0); ++j)
                    answers[c++] = (uint8_t)bin.curMatch(r, strand != 0, j);
        const auto t3 = std::chrono::steady_clock::now();
        const double enq = std::chrono::duration<double>(t1 - t0).count(), fl = std::chrono::duration<double>(t2 - t1).count(),
                     cons = std::chrono::duration<double>(t3 - t2).count();
        std::cout << "bin of " << n << " reads, " << sa.size() << " candidates: enqueue " << enq * 1e3 << " ms, flush "
                  << fl * 1e3 << " ms, consume " << cons * 1e3 << " ms; flush alone " << sa.size() / fl << " candidates/s, "
                  << "host to host " << sa.size() / (enq + fl + cons) << " candidates/s\n";
        best = std::max(best, sa.size() / fl);
    }
    std::ofstream o(out, std::ios::binary);
    o.write((const char*)answers.data(), (std::streamsize)answers.size());
    std::cout << "flush_candidates_per_s " << best << "\n";
    return 0;
}

int main(int argc, char** argv)
{
    if ((argc == 4 || argc == 5) && std::string(argv[1]) == "--bin") {
        try {
            return replay_bin(argv[2], argv[3], argc == 5 ? atoi(argv[4]) : 1);
        } catch (const std::exception& e) {
            std::cerr << "error: " << e.what() << "\n";
            return 1;
        }
    }
    if (argc == 4 && std::string(argv[1]) == "--mates") {
        try {
            return replay_mates(argv[2], argv[3]);
        } catch (const std::exception& e) {
            std::cerr << "replay_batch: " << e.what() << "\n";
            return 1;
        }
    }
    if (argc == 4 && std::string(argv[1]) == "--rescue") {
        try {
            return replay_rescue(argv[2], argv[3]);
        } catch (const std::exception& e) {
            std::cerr << "error: " << e.what() << "\n";
            return 1;
        }
    }
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
