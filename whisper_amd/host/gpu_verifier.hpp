// gpu_verifier.hpp - host-side mirror of the reference's verifier objects over the C-ABI.
//
// The reference calls one verifier per candidate from inside its per-thread loops
// (mapper/mapping_core_sa.cpp:215-348, mapper/sam_matching.cpp:689-752).  These classes
// keep the reference's method names and argument meaning, but a call only ENQUEUES the
// candidate and returns a ticket; flush() verifies the whole batch on the GPU through
// include/whisper_amd.h, after which the results are read back by ticket in the order the
// loop enqueued them.  One object per worker thread, as in the reference
// (mapper/mapping_core.cpp:154-191); objects are not thread-safe; a GpuReference (wamd_ctx) may be
// shared by the worker threads - its host-pointer calls are serialised by a lock inside the context.
//
// Header-only C++17, no dependency besides the C-ABI; errors become std::runtime_error
// like the reference's own (mapper/LevMyers64.cpp:244-252).
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/whisper_amd.h"

namespace whisper_amd {

using uchar_t = unsigned char;
using ref_pos_t = uint32_t;
enum class genome_t { direct, rev_comp };   // common/defs.h:79
// Orientation arguments are templates: inside the reference the call sites pass the reference's own
// ::genome_t (same enumerators), stand-alone callers pass whisper_amd::genome_t.
template <class G>
inline bool is_rev_comp(G orientation) { return orientation == G::rev_comp; }

inline void check(int rc, const char* what)
{
    if (rc != WAMD_OK)
        throw std::runtime_error(std::string(what) + ": " + wamd_last_error());
}

// CReference::GetData() uploaded once per device; replaces setReference() everywhere.
class GpuReference {
public:
    GpuReference(int device, const uchar_t* ref_ptr, uint64_t packed_bytes, uint32_t ref_size, uint32_t cur_offset)
    {
        check(wamd_ctx_create(device, ref_ptr, packed_bytes, ref_size, cur_offset, &ctx_), "wamd_ctx_create");
    }
    ~GpuReference() { wamd_ctx_destroy(ctx_); }
    GpuReference(const GpuReference&) = delete;
    GpuReference& operator=(const GpuReference&) = delete;
    // what a new stage changes (mapper/mapping_core.cpp:156: SetReference(ref, size, start_pos))
    void setReference(uint32_t ref_size, uint32_t cur_offset)
    {
        check(wamd_ctx_set_reference_params(ctx_, ref_size, cur_offset), "wamd_ctx_set_reference_params");
    }
    void setMaxTextLen(uint32_t max_text_len) { check(wamd_ctx_set_max_text_len(ctx_, max_text_len), "max_text_len"); }
    // Mapping stages only compare the distance with max_mismatches (mapper/mapping_core_sa.cpp:237-241):
    // with accepted_only the library reports rejected candidates as {k + 1, 0} and computes just the
    // band of diagonals an accepted alignment can use (about twice as fast at 150 bp).
    void setAcceptedOnly(bool accepted_only)
    {
        check(wamd_ctx_set_distance_mode(ctx_, accepted_only ? WAMD_DISTANCE_ACCEPTED_ONLY : WAMD_DISTANCE_EXACT),
              "wamd_ctx_set_distance_mode");
    }
    wamd_ctx* ctx() const { return ctx_; }

private:
    wamd_ctx* ctx_ = nullptr;
};

// Mirrors LevMyers (mapper/LevMyers.h:27-68) and CEditDist::LevDiag (mapper/edit_dist.h:50-51)
// for the mapping stages: reads are nibble-packed, only the edit distance is consumed.
class GpuLevMyers {
public:
    explicit GpuLevMyers(GpuReference& ref) : ref_(ref) {}

    // LevMyers::preprocess(seq, seq_len, orientation): the read every following
    // dynamicProgramming()/LevDiag() call verifies (mapper/mapping_core_sa.cpp:212-213).
    // Pass what the reference passes: the call site's own pattern_ptr with genome_t::direct, on
    // BOTH strands - in the rc branch pattern_ptr already is rc_data, the reverse-complemented read
    // (mapper/mapping_core.cpp:563-569, 1063), and LevDiag() verifies these very symbols as they
    // are.  genome_t::rev_comp makes dynamicProgramming() verify the reverse complement of `seq`
    // (what LevMyers::preprocess does with it, LevMyers64.cpp:184-194); no mapping-stage call site
    // asks for that.
    template <class G>
    bool preprocess(const uchar_t* seq, uint32_t seq_len, G orientation)
    {
        offsets_.push_back(blob_.size());
        lens_.push_back(seq_len);
        blob_.insert(blob_.end(), seq, seq + (seq_len + 1) / 2);
        cur_read_ = (uint32_t)lens_.size() - 1;
        cur_flags_ = is_rev_comp(orientation) ? WAMD_TASK_FLAG_REV_COMP : 0;
        need_rc_ = need_rc_ || cur_flags_ != 0;
        return true;
    }

    // LevMyers::dynamicProgramming(ref_pos, max_distance_in_ref, max_mate_edit_distance, pos, ed):
    // enqueue; returns the ticket to read the result with after flush().
    size_t dynamicProgramming(ref_pos_t ref_pos, uint32_t max_distance_in_ref, uint32_t max_mate_edit_distance)
    {
        wamd_dist_task t{};
        t.read_idx = cur_read_;
        t.ref_pos = ref_pos;
        t.max_distance_in_ref = max_distance_in_ref;
        t.max_ed = (uint16_t)max_mate_edit_distance;
        t.variant = WAMD_VARIANT_AUTO;   // by read length, mapping_core_sa.cpp:207-210
        t.flags = (uint8_t)cur_flags_;
        dist_tasks_.push_back(t);
        return dist_tasks_.size() - 1;
    }

    // CEditDist::LevDiag(pattern, len, position_in_text, fixed_left, fixed_right, max_mismatches, dir_gen_flag)
    size_t LevDiag(uint32_t position_in_text, uint32_t fixed_left, uint32_t fixed_right, uint32_t max_mismatches,
                   bool dir_gen_flag)
    {
        wamd_levdiag_task t{};
        t.read_idx = cur_read_;
        t.position_in_text = position_in_text;
        t.fixed_left = (uint16_t)fixed_left;
        t.fixed_right = (uint16_t)fixed_right;
        t.max_mismatches = (uint16_t)max_mismatches;
        t.dir_gen_flag = dir_gen_flag ? 1 : 0;
        lev_tasks_.push_back(t);
        return lev_tasks_.size() - 1;
    }
    // the reference's own argument list (edit_dist.h:50-51): pattern / len name the read of the last preprocess()
    size_t LevDiag(const uchar_t* /*pattern_ptr*/, uint32_t /*pattern_len*/, uint32_t position_in_text, uint32_t fixed_left,
                   uint32_t fixed_right, uint32_t max_mismatches, bool dir_gen_flag)
    {
        return LevDiag(position_in_text, fixed_left, fixed_right, max_mismatches, dir_gen_flag);
    }
    size_t pendingReads() const { return lens_.size(); }

    // Verifies everything enqueued since the last flush (one bin of one stage is the natural
    // unit: results never influence which candidates are enumerated).
    void flush()
    {
        dist_results_.assign(dist_tasks_.size(), wamd_dist_result{});
        lev_results_.assign(lev_tasks_.size(), 0u);
        if (lens_.empty())
            return;
        offsets_.push_back(blob_.size());
        wamd_reads* reads = nullptr;
        // pattern vectors of the stored orientation only, unless a caller asked for rev_comp
        check(wamd_reads_upload(ref_.ctx(), blob_.data(), offsets_.data(), lens_.data(), (uint32_t)lens_.size(),
                                WAMD_READS_NIBBLE, need_rc_ ? 3 : 1, &reads), "wamd_reads_upload");
        need_rc_ = false;
        int rc = WAMD_OK;
        if (!dist_tasks_.empty())
            rc = wamd_verify_distance(ref_.ctx(), reads, dist_tasks_.data(), (uint32_t)dist_tasks_.size(),
                                      WAMD_LAST_SYMBOL_FROM_READ, dist_results_.data());
        if (rc == WAMD_OK && !lev_tasks_.empty())
            rc = wamd_levdiag(ref_.ctx(), reads, lev_tasks_.data(), (uint32_t)lev_tasks_.size(), lev_results_.data());
        wamd_reads_free(reads);
        blob_.clear();
        offsets_.clear();
        lens_.clear();
        dist_tasks_.clear();
        lev_tasks_.clear();
        check(rc, "verification batch");
    }

    // the bool / pos / edit_distance dynamicProgramming() would have handed back
    bool result(size_t ticket, ref_pos_t& pos, uint32_t& edit_distance) const
    {
        const wamd_dist_result& r = dist_results_.at(ticket);
        pos = r.pos;
        edit_distance = r.edit_distance;
        return r.pos != 0 || r.edit_distance == 0;
    }
    uint32_t levDiagResult(size_t ticket) const { return lev_results_.at(ticket); }

private:
    GpuReference& ref_;
    std::vector<uchar_t> blob_;
    std::vector<uint64_t> offsets_;
    std::vector<uint32_t> lens_;
    std::vector<wamd_dist_task> dist_tasks_;
    std::vector<wamd_levdiag_task> lev_tasks_;
    std::vector<wamd_dist_result> dist_results_;
    std::vector<uint32_t> lev_results_;
    uint32_t cur_read_ = 0;
    uint32_t cur_flags_ = 0;
    bool need_rc_ = false;
};

// Mirrors the post-processing use: preprocessRawSeq + dynamicProgramming + getExtCigar
// (+ refine_ext_cigar), mapper/sam_matching.cpp:689-752 and 401-426.
class GpuLevMyersAligner {
public:
    GpuLevMyersAligner(GpuReference& ref, uint32_t max_cigar_len) : ref_(ref), stride_((max_cigar_len + 3) / 4 * 4) {}

    template <class G>
    bool preprocessRawSeq(const uchar_t* seq, uint32_t seq_len, G orientation)
    {
        offsets_.push_back(blob_.size());
        lens_.push_back(seq_len);
        blob_.insert(blob_.end(), seq, seq + seq_len);
        cur_read_ = (uint32_t)lens_.size() - 1;
        cur_flags_ = is_rev_comp(orientation) ? WAMD_TASK_FLAG_REV_COMP : 0;
        return true;
    }
    size_t dynamicProgramming(ref_pos_t ref_pos, uint32_t max_distance_in_ref, uint32_t max_mate_edit_distance)
    {
        wamd_dist_task t{};
        t.read_idx = cur_read_;
        t.ref_pos = ref_pos;
        t.max_distance_in_ref = max_distance_in_ref;
        t.max_ed = (uint16_t)max_mate_edit_distance;
        t.flags = (uint8_t)cur_flags_;
        tasks_.push_back(t);
        return tasks_.size() - 1;
    }
    void flush(bool enable_short_indel_refinement)
    {
        results_.assign(tasks_.size(), wamd_align_result{});
        cigars_.assign(tasks_.size() * (size_t)stride_, 0);
        if (tasks_.empty())
            return;
        offsets_.push_back(blob_.size());
        wamd_reads* reads = nullptr;
        check(wamd_reads_upload(ref_.ctx(), blob_.data(), offsets_.data(), lens_.data(), (uint32_t)lens_.size(),
                                WAMD_READS_ASCII, 3, &reads), "wamd_reads_upload");
        // CSamGenerator's objects only ever see preprocessRawSeq: last_symbol stays 0
        int rc = wamd_align(ref_.ctx(), reads, tasks_.data(), (uint32_t)tasks_.size(), 0,
                            enable_short_indel_refinement ? 1 : 0, results_.data(), cigars_.data(), stride_);
        wamd_reads_free(reads);
        blob_.clear();
        offsets_.clear();
        lens_.clear();
        tasks_.clear();
        check(rc, "wamd_align");
    }
    const wamd_align_result& result(size_t ticket) const { return results_.at(ticket); }
    const uchar_t* extCigar(size_t ticket) const { return cigars_.data() + ticket * (size_t)stride_; }

private:
    GpuReference& ref_;
    uint32_t stride_;
    std::vector<uchar_t> blob_;
    std::vector<uint64_t> offsets_;
    std::vector<uint32_t> lens_;
    std::vector<wamd_dist_task> tasks_;
    std::vector<wamd_align_result> results_;
    std::vector<uchar_t> cigars_;
    uint32_t cur_read_ = 0;
    uint32_t cur_flags_ = 0;
};

// Mirrors find_pairs<LevMyers>'s chain per hit (mapper/sam_matching.cpp:379-457) through wamd_rescue_mates: the window
// beside the hit (:392-400), preprocessRawSeq + dynamicProgramming, getExtCigar (+ refine_ext_cigar), Translate's
// verdict, the num_events tests, the mate's mapping_desc_t (lev), clipIllegalPositions and clipBoundaryPositions - one
// device call for every hit enqueued since the last flush.  The hit's own two clipping passes (:451, :455) stay where
// they are (or go through wamd_clip_illegal / wamd_clip_boundary on the caller's records).
class GpuMateRescue {
public:
    GpuMateRescue(GpuReference& ref, uint32_t max_cigar_len, bool short_indel_refinement, bool boundary_clipping)
        : ref_(ref), stride_((max_cigar_len + 3) / 4 * 4)
    {
        prm_.enable_short_indel_refinement = short_indel_refinement;
        prm_.enable_boundary_clipping = boundary_clipping;
    }
    // the mate's sequence as find_pairs holds it (sequence[!r], ASCII); returns the handle hits refer to
    uint32_t addMate(const uchar_t* seq, uint32_t seq_len)
    {
        offsets_.push_back(blob_.size());
        lens_.push_back(seq_len);
        blob_.insert(blob_.end(), seq, seq + seq_len);
        return (uint32_t)lens_.size() - 1;
    }
    // one hit: this_hit.mapping.pos, distanceThreshold (:348), max_mate_edit_distance (:390), paired_dir (:389),
    // this_hit.num_events, and the hit's chromosome: seq_desc = CRefSeqDesc::GetDescription(), this_hit.ref_seq_id
    // indexes it (ref_desc.cpp:81-118 is what the four derived fields restate; separate_n_len = SEPARATE_N_LEN)
    template <class G, class SeqDescVector>
    size_t hit(uint32_t mate, ref_pos_t hit_pos, ref_pos_t distance_threshold, uint32_t max_mate_edit_distance, G paired_dir,
               uint32_t hit_num_events, const SeqDescVector& seq_desc, int32_t ref_seq_id, uint32_t separate_n_len)
    {
        const auto& d = seq_desc.at((size_t)ref_seq_id);
        wamd_rescue_task t{};
        t.mate_read_idx = mate;
        t.hit_pos = hit_pos;
        t.distance_threshold = distance_threshold;
        t.seq_lo = (uint32_t)(d.pos_in_ref_seq + separate_n_len / 2);
        t.seq_hi = (size_t)ref_seq_id + 1 < seq_desc.size()
                       ? (uint32_t)(seq_desc[(size_t)ref_seq_id + 1].pos_in_ref_seq + separate_n_len / 2)
                       : (uint32_t)(d.pos_in_ref_seq + d.size + 2 * separate_n_len);
        t.seq_delta = (int32_t)((int64_t)d.no_initial_Ns - (int64_t)d.no_starting_Ns + 1 - (int64_t)d.pos_in_ref_seq);
        t.ref_size = (int32_t)(d.size + d.no_final_Ns + d.no_initial_Ns);
        t.max_mate_edit_distance = (uint16_t)max_mate_edit_distance;
        t.hit_num_events = (uint16_t)hit_num_events;
        t.paired_dir_direct = is_rev_comp(paired_dir) ? 0 : 1;
        tasks_.push_back(t);
        return tasks_.size() - 1;
    }
    void flush()
    {
        results_.assign(tasks_.size(), wamd_rescue_result{});
        cigars_.assign(tasks_.size() * (size_t)stride_, 0);
        if (tasks_.empty())
            return;
        offsets_.push_back(blob_.size());
        wamd_reads* reads = nullptr;
        check(wamd_reads_upload(ref_.ctx(), blob_.data(), offsets_.data(), lens_.data(), (uint32_t)lens_.size(),
                                WAMD_READS_ASCII, 3, &reads), "wamd_reads_upload");
        int rc = wamd_rescue_mates(ref_.ctx(), reads, tasks_.data(), (uint32_t)tasks_.size(), &prm_, results_.data(),
                                   cigars_.data(), stride_);
        wamd_reads_free(reads);
        blob_.clear();
        offsets_.clear();
        lens_.clear();
        tasks_.clear();
        check(rc, "wamd_rescue_mates");
    }
    // ok: the pair candidate exists (:430-433 all passed); then pos / ref_seq_pos / ref_length / err_edit_distance /
    // clippings / score are the mate's mapping_desc_t behind :456 and extCigar its ext-CIGAR
    const wamd_rescue_result& result(size_t ticket) const { return results_.at(ticket); }
    const uchar_t* extCigar(size_t ticket) const { return cigars_.data() + ticket * (size_t)stride_; }

private:
    GpuReference& ref_;
    uint32_t stride_;
    wamd_rescue_params prm_{};
    std::vector<uchar_t> blob_;
    std::vector<uint64_t> offsets_;
    std::vector<uint32_t> lens_;
    std::vector<wamd_rescue_task> tasks_;
    std::vector<wamd_rescue_result> results_;
    std::vector<uchar_t> cigars_;
};

// Mirrors CIndelMatching (mapper/indels.h:277-286) for the sensitive-mode matcher of the mapping stages
// (mapper/mapping_core_sa.cpp:258-272, 310-326): Preprocess + Match enqueue, flush() runs wamd_indel_match,
// result(ticket, candidate_mapping) fills the reference's candidate_mapping_t (common/types.h:224-318) the way
// Match leaves it - direction and the rc position fix-up stay with the caller, as in the reference.
class GpuIndelMatching {
public:
    GpuIndelMatching(GpuReference& ref, const wamd_indel_params& params) : ref_(ref), params_(params) {}

    template <class G>
    bool Preprocess(const uchar_t* seq, uint32_t seq_len, G orientation)   // nibble-packed read (indels.cpp:125-160)
    {
        return add_read(seq, (seq_len + 1) / 2, seq_len, WAMD_READS_NIBBLE, is_rev_comp(orientation));
    }
    template <class G>
    bool PreprocessRawSeq(const uchar_t* seq, uint32_t seq_len, G orientation)   // one ASCII symbol per byte
    {
        return add_read(seq, seq_len, seq_len, WAMD_READS_ASCII, is_rev_comp(orientation));
    }
    // Match(ref_pos, max_indel_size, fixed_segment_pos, fixed_segment_size, max_no_mismatches, candidate_mapping)
    size_t Match(ref_pos_t ref_pos, uint32_t max_indel_size, uint32_t fixed_segment_pos, uint32_t fixed_segment_size,
                 uint32_t max_no_mismatches)
    {
        wamd_indel_task t{};
        t.read_idx = cur_read_;
        t.ref_pos = ref_pos;
        t.max_indel_size = (uint16_t)max_indel_size;
        t.fixed_segment_pos = (uint16_t)fixed_segment_pos;
        t.fixed_segment_size = (uint16_t)fixed_segment_size;
        t.max_no_mismatches = (uint8_t)max_no_mismatches;
        t.flags = (uint8_t)cur_flags_;
        tasks_.push_back(t);
        return tasks_.size() - 1;
    }
    void flush()
    {
        results_.assign(tasks_.size(), wamd_mapping{});
        if (tasks_.empty()) {
            clear();
            return;
        }
        offsets_.push_back(blob_.size());
        wamd_reads* reads = nullptr;
        check(wamd_reads_upload(ref_.ctx(), blob_.data(), offsets_.data(), lens_.data(), (uint32_t)lens_.size(), encoding_,
                                need_rc_ ? 3 : 1, &reads), "wamd_reads_upload");
        const int rc = wamd_indel_match(ref_.ctx(), reads, tasks_.data(), (uint32_t)tasks_.size(), &params_, results_.data());
        wamd_reads_free(reads);
        clear();
        check(rc, "wamd_indel_match");
    }
    const wamd_mapping& result(size_t ticket) const { return results_.at(ticket); }
    // Match's bool and out-parameter; M is the reference's candidate_mapping_t, T its mapping_type_t
    template <class M>
    bool result(size_t ticket, M& candidate_mapping) const
    {
        const wamd_mapping& r = results_.at(ticket);
        if (!r.found)
            return false;
        candidate_mapping.type = static_cast<decltype(candidate_mapping.type)>(r.type);
        candidate_mapping.pos = r.pos;
        candidate_mapping.no_mismatches = r.no_mismatches;
        candidate_mapping.matching_len1 = r.matching_len1;
        candidate_mapping.matching_len2 = r.matching_len2;
        candidate_mapping.indel_len1 = r.indel_len1;
        candidate_mapping.indel_len2 = r.indel_len2;
        candidate_mapping.clipping_len = r.clipping_len;
        candidate_mapping.penalty = r.penalty;
        return true;
    }

private:
    bool add_read(const uchar_t* seq, size_t bytes, uint32_t seq_len, int encoding, bool rc)
    {
        if (!lens_.empty() && encoding != encoding_)
            throw std::runtime_error("GpuIndelMatching: one encoding per batch");
        encoding_ = encoding;
        offsets_.push_back(blob_.size());
        lens_.push_back(seq_len);
        blob_.insert(blob_.end(), seq, seq + bytes);
        cur_read_ = (uint32_t)lens_.size() - 1;
        cur_flags_ = rc ? WAMD_TASK_FLAG_REV_COMP : 0;
        need_rc_ = need_rc_ || rc;
        return true;
    }
    void clear()
    {
        blob_.clear();
        offsets_.clear();
        lens_.clear();
        tasks_.clear();
        need_rc_ = false;
    }
    GpuReference& ref_;
    wamd_indel_params params_;
    std::vector<uchar_t> blob_;
    std::vector<uint64_t> offsets_;
    std::vector<uint32_t> lens_;
    std::vector<wamd_indel_task> tasks_;
    std::vector<wamd_mapping> results_;
    int encoding_ = WAMD_READS_NIBBLE;
    uint32_t cur_read_ = 0;
    uint32_t cur_flags_ = 0;
    bool need_rc_ = false;
};

// Mirrors CIndelMatching::{PreprocessRawSeq, prefetch_genome, GetExtCigarNew} as fill_cigar_with_indel_matching uses them
// (mapper/sam_matching.cpp:824-852): the ext-CIGAR, score and error counts of a mapping the matcher found in the mapping
// stage, rebuilt from its compact record.  M is the reference's candidate_mapping_t (its mapping_type_t values are the
// WAMD_MAP_* values).
class GpuIndelExtCigar {
public:
    GpuIndelExtCigar(GpuReference& ref, uint32_t max_cigar_len) : ref_(ref), stride_((max_cigar_len + 3) / 4 * 4) {}

    template <class G>
    bool PreprocessRawSeq(const uchar_t* seq, uint32_t seq_len, G orientation)
    {
        offsets_.push_back(blob_.size());
        lens_.push_back(seq_len);
        blob_.insert(blob_.end(), seq, seq + seq_len);
        cur_read_ = (uint32_t)lens_.size() - 1;
        cur_flags_ = is_rev_comp(orientation) ? WAMD_TASK_FLAG_REV_COMP : 0;
        return true;
    }
    template <class M>
    size_t GetExtCigarNew(const M& mapping, ref_pos_t examined_start_pos)
    {
        wamd_indel_cigar_task t{};
        t.read_idx = cur_read_;
        t.examined_start_pos = examined_start_pos;
        t.type = (uint32_t)mapping.type;
        t.matching_len1 = mapping.matching_len1;
        t.matching_len2 = mapping.matching_len2;
        t.indel_len1 = mapping.indel_len1;
        t.indel_len2 = mapping.indel_len2;
        t.clipping_len = mapping.clipping_len;
        t.flags = cur_flags_;
        tasks_.push_back(t);
        return tasks_.size() - 1;
    }
    void flush()
    {
        results_.assign(tasks_.size(), wamd_indel_cigar_result{});
        cigars_.assign(tasks_.size() * (size_t)stride_, 0);
        if (tasks_.empty())
            return;
        offsets_.push_back(blob_.size());
        wamd_reads* reads = nullptr;
        check(wamd_reads_upload(ref_.ctx(), blob_.data(), offsets_.data(), lens_.data(), (uint32_t)lens_.size(),
                                WAMD_READS_ASCII, 3, &reads), "wamd_reads_upload");
        const int rc = wamd_indel_ext_cigar(ref_.ctx(), reads, tasks_.data(), (uint32_t)tasks_.size(), results_.data(),
                                            cigars_.data(), stride_);
        wamd_reads_free(reads);
        blob_.clear();
        offsets_.clear();
        lens_.clear();
        tasks_.clear();
        check(rc, "wamd_indel_ext_cigar");
    }
    // ok = GetExtCigarNew's bool; the other fields are what it leaves in mapping_desc_t
    const wamd_indel_cigar_result& result(size_t ticket) const { return results_.at(ticket); }
    const uchar_t* extCigar(size_t ticket) const { return cigars_.data() + ticket * (size_t)stride_; }

private:
    GpuReference& ref_;
    uint32_t stride_;
    std::vector<uchar_t> blob_;
    std::vector<uint64_t> offsets_;
    std::vector<uint32_t> lens_;
    std::vector<wamd_indel_cigar_task> tasks_;
    std::vector<wamd_indel_cigar_result> results_;
    std::vector<uchar_t> cigars_;
    uint32_t cur_read_ = 0;
    uint32_t cur_flags_ = 0;
};

// Mirrors CSoftClipping's rescue use (mapper/soft_clipping.h:122-137; sam_matching.cpp:556-575, 771-797):
// preprocessRawSeq + match enqueue, flush() runs wamd_softclip_match (match and getExtCigar in one go),
// result(ticket) holds match's out-parameters, extCigar(ticket) / result(ticket).affine_score getExtCigar's.
class GpuSoftClipping {
public:
    GpuSoftClipping(GpuReference& ref, const wamd_softclip_params& params, uint32_t max_cigar_len)
        : ref_(ref), params_(params), stride_((max_cigar_len + 4) / 4 * 4) {}

    template <class G>
    bool preprocessRawSeq(const uchar_t* seq, uint32_t seq_len, G orientation)
    {
        offsets_.push_back(blob_.size());
        lens_.push_back(seq_len);
        blob_.insert(blob_.end(), seq, seq + seq_len);
        cur_read_ = (uint32_t)lens_.size() - 1;
        cur_flags_ = is_rev_comp(orientation) ? WAMD_TASK_FLAG_REV_COMP : 0;
        return true;
    }
    // match(ref_pos, max_distance_in_ref, min_exact_len, pos, left_clipping, ..., no_mismatches)
    size_t match(ref_pos_t ref_pos, uint32_t max_distance_in_ref, uint32_t min_exact_len)
    {
        wamd_softclip_task t{};
        t.read_idx = cur_read_;
        t.ref_pos = ref_pos;
        t.max_distance_in_ref = max_distance_in_ref;
        t.min_exact_len = (uint16_t)min_exact_len;
        t.flags = (uint8_t)cur_flags_;
        tasks_.push_back(t);
        return tasks_.size() - 1;
    }
    void flush()
    {
        results_.assign(tasks_.size(), wamd_softclip_result{});
        cigars_.assign(tasks_.size() * (size_t)stride_, 0);
        if (!tasks_.empty()) {
            offsets_.push_back(blob_.size());
            wamd_reads* reads = nullptr;
            check(wamd_reads_upload(ref_.ctx(), blob_.data(), offsets_.data(), lens_.data(), (uint32_t)lens_.size(),
                                    WAMD_READS_ASCII, 3, &reads), "wamd_reads_upload");
            const int rc = wamd_softclip_match(ref_.ctx(), reads, tasks_.data(), (uint32_t)tasks_.size(), &params_,
                                               results_.data(), cigars_.data(), stride_);
            wamd_reads_free(reads);
            params_.fresh_object = 0;   // the reference's object keeps its touched cells from call to call
            blob_.clear();
            offsets_.clear();
            lens_.clear();
            tasks_.clear();
            check(rc, "wamd_softclip_match");
        }
    }
    // match's bool and out-parameters
    bool result(size_t ticket, ref_pos_t& pos, uint32_t& left_clipping, uint32_t& right_clipping, uint32_t& left_match,
                uint32_t& right_match, int32_t& del_size, uint32_t& no_mismatches) const
    {
        const wamd_softclip_result& r = results_.at(ticket);
        pos = r.pos;
        left_clipping = r.left_clipping;
        right_clipping = r.right_clipping;
        left_match = r.left_match;
        right_match = r.right_match;
        del_size = r.del_size;
        no_mismatches = r.no_mismatches;
        return r.found != 0;
    }
    const wamd_softclip_result& result(size_t ticket) const { return results_.at(ticket); }
    const uchar_t* extCigar(size_t ticket) const { return cigars_.data() + ticket * (size_t)stride_; }

private:
    GpuReference& ref_;
    wamd_softclip_params params_;
    uint32_t stride_;
    std::vector<uchar_t> blob_;
    std::vector<uint64_t> offsets_;
    std::vector<uint32_t> lens_;
    std::vector<wamd_softclip_task> tasks_;
    std::vector<wamd_softclip_result> results_;
    std::vector<uchar_t> cigars_;
    uint32_t cur_read_ = 0;
    uint32_t cur_flags_ = 0;
};

// ---- one bin of one stage in the compact form ------------------------------------------------------------------
// The suffix-array loops (mapper/mapping_core_sa.cpp:193-348, 355-513) verify candidate after candidate; what the
// bin loop around them (mapper/mapping_core.cpp:860-1010) does with a read's answers - Push, best_error, pass_read -
// depends on that read's answers alone, and nothing the loops enumerate depends on an answer.  So a bin is enqueued
// as the loops walk it, verified with ONE flush behind the bin's read loop, and consumed read by read in the same
// order.  Per candidate the loops hand over the 4-byte suffix-array entry; the CF pre-filter of :228-230 / :283-285
// runs on the device in front of the verifier (wamd_verify_sa_candidates), so the host computes no histograms.
// Reads whose max_mismatches is below lev_alg_thr go to CEditDist::LevDiag, the others to dynamicProgramming
// (:236-239): two batches inside, each with its reads in page-locked memory.
template <class T>
class PinnedVec {
public:
    PinnedVec() = default;
    PinnedVec(const PinnedVec&) = delete;
    PinnedVec& operator=(const PinnedVec&) = delete;
    ~PinnedVec() { wamd_host_free(p_); }
    void push_back(const T& v)
    {
        if (n_ == cap_)
            grow(cap_ ? 2 * cap_ : 4096);
        p_[n_++] = v;
    }
    T* append(size_t k)      // k more elements, uninitialised
    {
        if (n_ + k > cap_)
            grow(n_ + k > 2 * cap_ ? n_ + k : 2 * cap_);
        T* at = p_ + n_;
        n_ += k;
        return at;
    }
    void resize(size_t k)
    {
        if (k > cap_)
            grow(k);
        n_ = k;
    }
    void clear() { n_ = 0; }
    size_t size() const { return n_; }
    T* data() { return p_; }
    const T* data() const { return p_; }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }

private:
    void grow(size_t want)
    {
        void* q = nullptr;
        check(wamd_host_alloc(&q, want * sizeof(T)), "wamd_host_alloc");
        if (n_)
            memcpy(q, p_, n_ * sizeof(T));
        wamd_host_free(p_);
        p_ = (T*)q;
        cap_ = want;
    }
    T* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

class GpuSaBin {
public:
    GpuSaBin(GpuReference& ref, uint32_t lev_alg_thr) : ref_(ref), lev_alg_thr_(lev_alg_thr) {}
    ~GpuSaBin()
    {
        for (Sub& s : sub_)
            wamd_reads_free(s.reads);
    }
    // what the stage holds fixed: the anchor [start_pos, end_pos) and min_read_len (mapping_core.cpp:150-160)
    void beginBin(uint32_t start_pos, uint32_t end_pos, uint32_t min_read_len)
    {
        start_pos_ = start_pos;
        end_pos_ = end_pos;
        min_read_len_ = min_read_len;
        for (Sub& s : sub_)
            s.clear();
        where_.clear();
    }
    // A read the loops are about to search: its symbols as the bin holds them, the reversed read the loop built
    // from them (reverse_read, mapping_core.cpp:563) and the max_mismatches it is searched with.  Returns its handle.
    uint32_t addRead(const uchar_t* data, const uchar_t* rc_data, uint32_t size, uint32_t max_mismatches)
    {
        const uint32_t which = max_mismatches >= lev_alg_thr_ ? 0u : 1u;
        Sub& s = sub_[which];
        const size_t bytes = (size + 1) / 2;
        s.off.push_back(s.blob.size());
        memcpy(s.blob.append(bytes), data, bytes);
        if (which == 1)      // LevDiag on the other strand verifies the reversed read itself (:291)
            memcpy(s.rc_blob.append(bytes), rc_data, bytes);
        s.lens.push_back(size);
        s.max_len = size > s.max_len ? size : s.max_len;
        s.k.push_back((uint8_t)max_mismatches);
        s.counts.push_back(0);
        s.counts.push_back(0);
        s.first.push_back((uint32_t)s.sa.size());
        where_.push_back(which << 31 | (uint32_t)(s.lens.size() - 1));
        return (uint32_t)where_.size() - 1;
    }
    // one suffix-array entry of the last added read, in the loops' order: its direct-strand entries first
    void candidate(uint32_t sa_entry, bool dir_gen_flag)
    {
        Sub& s = sub_[where_.back() >> 31];
        uint16_t& c = s.counts[s.counts.size() - (dir_gen_flag ? 2 : 1)];
        if (c == 0xffff || (dir_gen_flag && s.counts[s.counts.size() - 1]))
            throw std::runtime_error("GpuSaBin: more than 65535 entries of a read on one strand, or a direct-strand entry "
                                     "after the other strand's");
        ++c;
        s.sa.push_back(sa_entry);
    }
    size_t pendingCandidates() const { return sub_[0].sa.size() + sub_[1].sa.size(); }

    void flush()
    {
        for (uint32_t which = 0; which < 2; ++which) {
            Sub& s = sub_[which];
            const uint32_t n = (uint32_t)s.lens.size();
            s.errors.resize(s.sa.size());
            if (!n || !s.sa.size())
                continue;
            if (which == 1) {      // the reversed reads behind the reads: read i's copy is read n + i
                const size_t fwd = s.blob.size();
                memcpy(s.blob.append(s.rc_blob.size()), s.rc_blob.data(), s.rc_blob.size());
                for (uint32_t i = 0; i < n; ++i) {
                    s.off.push_back(fwd + s.off[i]);
                    s.lens.push_back(s.lens[i]);
                }
            }
            const uint32_t n_up = (uint32_t)s.lens.size();
            if (!s.reads)
                check(wamd_reads_upload(ref_.ctx(), s.blob.data(), s.off.data(), s.lens.data(), n_up, WAMD_READS_NIBBLE, 3,
                                        &s.reads), "wamd_reads_upload");
            else {
                // the bin's bytes and its longest read are known here: no second pass over the reads on the host
                check(wamd_reads_refill_async(s.reads, s.blob.data(), s.blob.size(), s.off.data(), s.lens.data(), n_up,
                                              s.max_len, nullptr), "wamd_reads_refill_async");
                check(wamd_ctx_synchronize(ref_.ctx()), "wamd_ctx_synchronize");   // the verification runs on other streams
            }
            wamd_sa_batch b{};
            b.read_first = 0;
            b.n_reads = n;
            b.reverse_read_offset = n;
            b.max_mismatches = 0;      // every read brings its own
            b.fixed_left = (uint16_t)start_pos_;
            b.fixed_right = (uint16_t)end_pos_;
            b.min_read_len = (uint16_t)min_read_len_;
            b.verifier = which == 1 ? WAMD_SA_VERIFIER_LEVDIAG : WAMD_SA_VERIFIER_DISTANCE;
            check(wamd_verify_sa_candidates(ref_.ctx(), s.reads, &b, s.counts.data(), s.k.data(), s.sa.data(),
                                            (uint32_t)s.sa.size(), s.errors.data(), nullptr), "wamd_verify_sa_candidates");
        }
    }
    // after flush(): the answers of read `h`, strand by strand, in the order they were enqueued
    uint32_t count(uint32_t h, bool dir_gen_flag) const
    {
        const Sub& s = sub_[where_[h] >> 31];
        return s.counts[2 * (size_t)(where_[h] & 0x7fffffffu) + (dir_gen_flag ? 0 : 1)];
    }
    // cur_match as the loops go on with it: the errors if <= max_mismatches, else max_mismatches + 1
    uint32_t curMatch(uint32_t h, bool dir_gen_flag, uint32_t j) const { return sub_[where_[h] >> 31].errors[index(h, dir_gen_flag, j)]; }
    uint32_t saEntry(uint32_t h, bool dir_gen_flag, uint32_t j) const { return sub_[where_[h] >> 31].sa[index(h, dir_gen_flag, j)]; }

private:
    struct Sub {
        PinnedVec<uchar_t> blob, rc_blob;
        PinnedVec<uint64_t> off;
        PinnedVec<uint32_t> lens;
        PinnedVec<uint16_t> counts;
        PinnedVec<uint8_t> k;
        PinnedVec<uint32_t> sa;
        PinnedVec<uint8_t> errors;
        std::vector<uint32_t> first;
        wamd_reads* reads = nullptr;
        uint32_t max_len = 0;
        void clear()
        {
            max_len = 0;
            blob.clear(); rc_blob.clear(); off.clear(); lens.clear(); counts.clear(); k.clear(); sa.clear(); errors.clear();
            first.clear();
        }
    };
    size_t index(uint32_t h, bool dir_gen_flag, uint32_t j) const
    {
        const Sub& s = sub_[where_[h] >> 31];
        const size_t r = where_[h] & 0x7fffffffu;
        return (size_t)s.first[r] + (dir_gen_flag ? 0 : s.counts[2 * r]) + j;
    }
    GpuReference& ref_;
    uint32_t lev_alg_thr_;
    uint32_t start_pos_ = 0, end_pos_ = 0, min_read_len_ = 0;
    Sub sub_[2];
    std::vector<uint32_t> where_;     // per handle: batch (bit 31) and the read's index in it
};

} // namespace whisper_amd
