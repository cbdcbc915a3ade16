#!/usr/bin/env python3
"""Compile the INTEGRATION.md patch against the reference's own headers.

What this does: builds a scratch tree OUTSIDE the repository (a temporary directory) whose
files are symlinks into /root/reference/src, except the five files the patch touches, which
(six in the per-bin form) are copied there and edited at the lines INTEGRATION.md sections 3 and 4 name; then compiles the
patched translation units with g++ (-fsyntax-only, the flags of the reference's makefile)
against whisper_amd/host/gpu_verifier.hpp and include/whisper_amd.h.

This file holds only the replacement lines and, per edit, the line range it replaces and one
identifier that must occur in the first replaced line (so a reference of another version is
detected instead of being patched at the wrong place).  Nothing of the reference is stored here
and nothing is written under /root/reference.

    python scripts/check_integration_patch.py [--keep]     exit 0 = both objects compile

Two forms of the patch are compiled, each in its own scratch tree:
  per-range  every candidate of one suffix-array range is enqueued, the range is flushed, the results are
             consumed in the order the loop enqueued them - the function still returns best_match to its
             caller, so nothing outside it changes (the smallest edit; one device round trip per range).
  per-bin    the form to run: search_in_* only enqueue the 4-byte suffix-array entries (GpuSaBin, the compact
             form of wamd_verify_sa_candidates: the CF pre-filter runs on the device), process_reads keeps one
             record per read of the bin, the bin is flushed ONCE behind its read loop, and what the loop did
             with a read's answers (Push, best_error, pass_read - mapping_core.cpp:973-1036) runs read by read
             afterwards, in the same order.  A malicious group's q-gram filter (search_in_malicious_group_with_DP)
             stays on the host; the entries it lets through are enqueued like the others.
"""
import os
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WAMD_REFERENCE", "/root/reference") + "/src"

# (first line, last line, identifier expected in the first line, replacement) - 1-based, inclusive;
# last < first means "insert before `first`".
HEADER_CORE = [
    (28, 28, "LevMyers.h", '''#include "LevMyers.h"
#include "gpu_verifier.hpp"
'''),
    (109, 109, "indel_matching_rc", '''	CIndelMatching* indel_matching_rc;
	// MI355X verification path (INTEGRATION.md section 3)
	whisper_amd::GpuReference* gpu_ref;         // one per process and device in production; per core here
	whisper_amd::GpuLevMyers* gpu_lev;
	whisper_amd::GpuIndelMatching* gpu_indel_dir;
	whisper_amd::GpuIndelMatching* gpu_indel_rc;
	struct wamd_pending_t {
		size_t ticket, indel_ticket;
		uint32_t sa;
		bool verified, with_indel;
	};
	std::vector<wamd_pending_t> wamd_pending;
	uint32_t wamd_consume(uint32_t pattern_len, bool dir_gen_flag, bool known_range, uint32_t max_mismatches);
'''),
]

CORE_CPP = [
    # construction next to levMyers64/128/256 and the two CIndelMatching objects
    (191, 191, "indel_matching_rc->SetReference", '''	indel_matching_rc->SetReference(reference->GetData(), 0, 0);
	{
		wamd_indel_params ip{};
		ip.scoring.match = params->match_score;
		ip.scoring.mismatch = params->mismatch_score;
		ip.scoring.gap_ins_open = params->gap_ins_open;
		ip.scoring.gap_ins_extend = params->gap_ins_extend;
		ip.scoring.gap_del_open = params->gap_del_open;
		ip.scoring.gap_del_extend = params->gap_del_extend;
		ip.scoring.clipping = params->clipping_score;
		ip.min_clipped_factor = params->min_clipped_factor;
		ip.max_indel_bound = params->max_approx_indel_len;
		gpu_ref = new whisper_amd::GpuReference(0, reference->GetData(), ((uint64_t) sa_dir->GetRefSize() + 1) / 2,
			sa_dir->GetRefSize(), start_pos);
		gpu_ref->setMaxTextLen(max_ref_len);
		gpu_ref->setAcceptedOnly(true);
		gpu_lev = new whisper_amd::GpuLevMyers(*gpu_ref);
		gpu_indel_dir = new whisper_amd::GpuIndelMatching(*gpu_ref, ip);
		gpu_indel_rc = new whisper_amd::GpuIndelMatching(*gpu_ref, ip);
	}
'''),
    # the read both matcher objects see (the rc object reverse-complements it itself)
    (871, 872, "indel_matching_dir->Preprocess", '''				gpu_indel_dir->Preprocess(data, size, genome_t::direct);
				gpu_indel_rc->Preprocess(data, size, genome_t::rev_comp);
'''),
]

# The consume block: what lines 241-247 / 293-299 (lev), 266-267 / 319-321 (indel) and 334-344 do with a result.
CONSUME = '''
// *** MI355X path: results of one flushed range, in enumeration order (INTEGRATION.md section 3)
uint32_t CMappingCore::wamd_consume(uint32_t pattern_len, bool dir_gen_flag, bool known_range, uint32_t max_mismatches)
{
	uint32_t best_match = max_mismatches + 1;
	whisper_amd::GpuIndelMatching* gpu_indel = dir_gen_flag ? gpu_indel_dir : gpu_indel_rc;

	for (auto& p : wamd_pending)
	{
		candidate_mapping_t candidate_mapping;
		uint32_t cur_match = max_mismatches + 1;

		if (p.verified)
		{
			ref_pos_t ref_pos;
			if (max_mismatches >= lev_alg_thr)
				gpu_lev->result(p.ticket, ref_pos, cur_match);
			else
				cur_match = gpu_lev->levDiagResult(p.ticket);
		}
		if (cur_match <= max_mismatches)
		{
			if (dir_gen_flag)
				candidate_mapping = candidate_mapping_t::construct<mapping_type_t::lev>(
					known_range ? p.sa - start_pos : p.sa - CUR_OFFSET, genome_t::direct, cur_match);
			else
				candidate_mapping = candidate_mapping_t::construct<mapping_type_t::lev>(
					ref_size - (p.sa - CUR_OFFSET) - 1, genome_t::rev_comp, cur_match);
		}
		if (p.with_indel)
		{
			candidate_mapping_t candidate_mapping_indel;
			if (gpu_indel->result(p.indel_ticket, candidate_mapping_indel))
			{
				candidate_mapping_indel.direction = dir_gen_flag ? genome_t::direct : genome_t::rev_comp;
				if (!dir_gen_flag)
					candidate_mapping_indel.pos += (pattern_len - 1);
				candidate_mapping = candidate_mapping_t::better(candidate_mapping, candidate_mapping_indel);
			}
		}
		if (candidate_mapping.type != mapping_type_t::none && candidate_mapping.penalty < best_match)
			best_match = candidate_mapping.penalty;
		if (candidate_mapping.type != mapping_type_t::none && candidate_mapping.penalty <= max_mismatches)
		{
			if (dir_gen_flag)
				dir_match_positions.push_back(candidate_mapping);
			else
				rc_match_positions.push_back(candidate_mapping);
		}
	}
	wamd_pending.clear();
	return best_match;
}
'''


def enqueue(indent):
    """The body that replaces one candidate's verification (both strands share it)."""
    t = indent
    return (
        t + "wamd_pending_t wamd_p{0, 0, sa_part[i], false, false};\n" +
        t + "if (dir_gen_flag)\n" +
        t + "{\n" +
        t + "	if (is_short_read || test_CF_128b_SSE_difference(CF_value_SSE_for_a_dir_pattern, CF_vector_128b_SSE_dir_gen[i], 4 * max_mismatches))\n" +
        t + "	{\n" +
        t + "		wamd_p.verified = true;\n" +
        t + "		wamd_p.ticket = (max_mismatches >= lev_alg_thr)\n" +
        t + "			? gpu_lev->dynamicProgramming(sa_part[i] - start_pos - max_mismatches, pattern_len + 2 * max_mismatches, max_mismatches)\n" +
        t + "			: gpu_lev->LevDiag(pattern_ptr, pattern_len, sa_part[i], start_pos, end_pos, max_mismatches, 1);\n" +
        t + "	}\n" +
        t + "	if (sensitive_mode && enable_mapping_indels)\n" +
        t + "	{\n" +
        t + "		wamd_p.with_indel = true;\n" +
        t + "		wamd_p.indel_ticket = gpu_indel_dir->Match(sa_part[i] - start_pos, max_approx_indel_len, start_pos, end_pos - start_pos, max_mismatches);\n" +
        t + "	}\n" +
        t + "}\n" +
        t + "else\n" +
        t + "{\n" +
        t + "	if (is_short_read || test_CF_128b_SSE_difference(CF_value_SSE_for_a_rc_pattern, CF_vector_128b_SSE_rc_gen[i], 4 * max_mismatches))\n" +
        t + "	{\n" +
        t + "		wamd_p.verified = true;\n" +
        t + "		wamd_p.ticket = (max_mismatches >= lev_alg_thr)\n" +
        t + "			? gpu_lev->dynamicProgramming(ref_size - sa_part[i] - pattern_len + start_pos - max_mismatches, pattern_len + 2 * max_mismatches, max_mismatches)\n" +
        t + "			: gpu_lev->LevDiag(pattern_ptr, pattern_len, sa_part[i], pattern_len - end_pos, pattern_len - start_pos, max_mismatches, 0);\n" +
        t + "	}\n" +
        t + "	if (sensitive_mode && enable_mapping_indels)\n" +
        t + "	{\n" +
        t + "		wamd_p.with_indel = true;\n" +
        t + "		wamd_p.indel_ticket = gpu_indel_rc->Match(ref_size - sa_part[i] - pattern_len + start_pos, max_approx_indel_len, pattern_len - end_pos, end_pos - start_pos, max_mismatches);\n" +
        t + "	}\n" +
        t + "}\n" +
        t + "wamd_pending.push_back(wamd_p);\n")


CF_OF_CANDIDATE = '''		if (!is_short_read)
		{
			if (dir_gen_flag)
				calculate_CF_128b_SEE_for_dir_and_rc_gen(start_pos, end_pos, min_read_len, sa_part[i], CF_vector_128b_SSE_dir_gen[i], dir_gen_flag);
			else
				calculate_CF_128b_SEE_for_dir_and_rc_gen(min_read_len - end_pos, min_read_len - start_pos, min_read_len, sa_part[i], CF_vector_128b_SSE_rc_gen[i], 0);
		}
'''

SA_CPP = [
    # ---- search_in_unknown_range_dir_and_rc (193-353) ----
    (212, 213, "lev_alg_thr", '''	gpu_lev->preprocess(pattern_ptr, pattern_len, genome_t::direct);   // LevDiag verifies the same upload
	wamd_pending.clear();
'''),
    # the loop body up to the anchor comparison of the next entry: the candidate's CF histogram as at 228 / 283,
    # then enqueue instead of verify (230-272, 285-326); 334-344 move into wamd_consume
    (217, 344, "candidate_mapping_t", CF_OF_CANDIDATE + enqueue("\t\t")),
    (350, 352, "last_index", '''	*last_index = i;

	gpu_lev->flush();
	if (sensitive_mode && enable_mapping_indels)
		(dir_gen_flag ? gpu_indel_dir : gpu_indel_rc)->flush();
	return wamd_consume(pattern_len, dir_gen_flag, false, max_mismatches);
'''),
    # ---- search_in_known_range_dir_and_rc (359-513): the CF histograms are those of the unknown-range pass ----
    (376, 377, "lev_alg_thr", '''	gpu_lev->preprocess(pattern_ptr, pattern_len, genome_t::direct);
	wamd_pending.clear();
'''),
    (383, 504, "candidate_mapping_t", enqueue("\t\t\t")),
    (509, 509, "best_match", '''		gpu_lev->flush();
		if (sensitive_mode && enable_mapping_indels)
			(dir_gen_flag ? gpu_indel_dir : gpu_indel_rc)->flush();
		return wamd_consume(pattern_len, dir_gen_flag, true, max_mismatches);
'''),
    (514, 513, "", CONSUME),
]

HEADER_SAM = [
    (28, 28, "LevMyers.h", '''#include "LevMyers.h"
#include "gpu_verifier.hpp"
'''),
    (89, 89, "softClipping", '''	CSoftClipping* softClipping;
	whisper_amd::GpuReference* gpu_ref;
	whisper_amd::GpuLevMyersAligner* gpu_aligner;
	whisper_amd::GpuMateRescue* gpu_rescue;
	whisper_amd::GpuSoftClipping* gpu_softclip;
	whisper_amd::GpuIndelExtCigar* gpu_indel_cigar;
'''),
]

SAM_MATCHING_CPP = [
    # find_pairs<LevMyers> (318-504): the chain per hit (379-457) through GpuMateRescue / wamd_rescue_mates - every hit
    # of the pair enqueued (the fill_cigar of an unmatched hit stays in front, :386-387), one flush (per block of pairs
    # in production), then the loop's second half in the loop's order
    (331, 338, "LevMyers", ""),
    (352, 482, "for (uint32_t r = 0", '''	struct wamd_hit_t {
		mapping_desc_t* h;
		uint32_t r;
		size_t ticket;
		genome_t paired_dir;
	};
	std::vector<wamd_hit_t> wamd_hits;
	const uint32_t wamd_mate[2] = { gpu_rescue->addMate(sequence[0], sequence_len[0]), gpu_rescue->addMate(sequence[1], sequence_len[1]) };
	for (uint32_t r = 0; r < 2; ++r) {
		for (auto h : hits[r]) {
			if (h->err_edit_distance < editDistanceRange.first || h->err_edit_distance > editDistanceRange.second)
				continue; // ignore hits if edit distance is outside specified range
			mapping_desc_t& this_hit = *h;
			if (this_hit.method == MatchingMethod::Enumeration::Unmatched)
				fill_cigar(this_hit, id[r], sequence[r], sequence_len[r], quality[r], false, true);
			genome_t paired_dir = get_direction_using_orientation(this_hit.mapping.direction, mapping_orientation);
			uint32_t max_mate_edit_distance = (uint32_t)(sequence_len[!r] * max_mate_edit_distance_frac) * 2;
			wamd_hits.push_back({ h, r, gpu_rescue->hit(wamd_mate[!r], this_hit.mapping.pos, distanceThreshold, max_mate_edit_distance,
				paired_dir, this_hit.num_events, ref_seq_desc->GetDescription(), this_hit.ref_seq_id, SEPARATE_N_LEN), paired_dir });
		}
	}
	gpu_rescue->flush();
	for (auto& wh : wamd_hits) {
		const uint32_t r = wh.r;
		mapping_desc_t& this_hit = *wh.h;
		const wamd_rescue_result& wr = gpu_rescue->result(wh.ticket);
		if (!wr.ok)		// dynamicProgramming, Translate or one of the num_events tests said no (:407, :430-433)
			continue;
		uchar_t* ext_cigar;
		mp_ext_cigar->Reserve(ext_cigar);
		memcpy(ext_cigar, gpu_rescue->extCigar(wh.ticket), (size_t) wr.ext_cigar_len + 1);
		paired_raw_pos = wr.pos;
		paired_ref_seq_id = this_hit.ref_seq_id;

		candidate_mapping_t paired_mapping = candidate_mapping_t::construct<mapping_type_t::lev>(paired_raw_pos, wh.paired_dir, wr.edit_distance);

		mapping_desc[!r].emplace_back(paired_mapping, paired_ref_seq_id, wr.ref_seq_pos, sequence_len[!r], ext_cigar,
			params->gap_del_open, params->gap_del_extend, params->gap_ins_open, params->gap_ins_extend, params->mismatch_score);
		auto& mate_hit = mapping_desc[!r].back();

		// the mate as clipIllegalPositions / clipBoundaryPositions left it on the device (:444-456)
		mate_hit.method = MatchingMethod::Enumeration::LevMyers;
		mate_hit.ref_length = wr.ref_length;
		mate_hit.err_edit_distance = wr.err_edit_distance;
		mate_hit.left_clipping = wr.left_clipping;
		mate_hit.right_clipping = wr.right_clipping;
		mate_hit.score = wr.score;
		mate_hit.score_type = ScoringType::Affine;

		softClipping->clipIllegalPositions(this_hit, insertSizeModel->getScoring(), ref_seq_desc->GetDescription(), *mp_ext_cigar);
		if (params->enable_boundary_clipping)
			softClipping->clipBoundaryPositions(this_hit, insertSizeModel->getScoring());

		mapping_pair_t candidate(
			r == 0 ? &this_hit : &mate_hit,
			r == 0 ? &mate_hit : &this_hit);

		candidate.score = pairEvaluator(candidate);
		pushMappingHeap(mapping_pairs, candidate, mapping_heap_size);
	}
'''),
    # find_pairs<Clipping> (507-639): preprocessRawSeq + match + getExtCigar through GpuSoftClipping / wamd_softclip_match.
    # A first pass in front of the loop enqueues every hit's window (the fill_cigar of an unmatched hit with it: the
    # loop then finds the hit matched), one flush, and the loop reads its answers in its own order.
    (528, 527, "", '''	std::vector<size_t> wamd_clip_tickets;
	for (uint32_t r = 0; r < 2; ++r) {
		for (auto h : hits[r]) {
			if (h->err_edit_distance < editDistanceRange.first || h->err_edit_distance > editDistanceRange.second ||
				h->method == MatchingMethod::Enumeration::LevMyers) {
				wamd_clip_tickets.push_back((size_t) -1);	// (fill_cigar below may move err_edit_distance: the loop must not ask again)
				continue;
			}
			if (h->method == MatchingMethod::Enumeration::Unmatched)
				fill_cigar(*h, id[r], sequence[r], sequence_len[r], quality[r], false, true);
			genome_t wamd_dir = get_direction_using_orientation(h->mapping.direction, mapping_orientation);
			gpu_softclip->preprocessRawSeq(sequence[!r], sequence_len[!r], wamd_dir);
			wamd_clip_tickets.push_back(gpu_softclip->match(
				wamd_dir == genome_t::direct ? h->mapping.pos - distanceThreshold + sequence_len[!r] : h->mapping.pos + 1,
				distanceThreshold, HashTable::hash_len));
		}
	}
	gpu_softclip->flush();		// per block of pairs in production
	size_t wamd_clip_next = 0;
'''),
    (530, 533, "editDistanceRange", '''			const size_t wamd_t = wamd_clip_tickets[wamd_clip_next++];
			if (wamd_t == (size_t) -1)
				continue; // ignored in the first pass: edit distance outside the range, or a LevMyers hit
'''),
    (556, 558, "preprocessRawSeq", '''			if (gpu_softclip->result(wamd_t, relative_begin_pos, left_clipping, right_clipping,
				left_match, right_match, del_size, no_mismatches)) {
'''),
    (562, 575, "direct", '''						uchar_t* ext_cigar;
						mp_ext_cigar->Reserve(ext_cigar);
						memcpy(ext_cigar, gpu_softclip->extCigar(wamd_t), (size_t) gpu_softclip->result(wamd_t).cigar_len + 1);
						double affine_score = gpu_softclip->result(wamd_t).affine_score;
'''),
    # fill_cigar_with_lev (689-752): preprocessRawSeq + dynamicProgramming, then getExtCigar + refine_ext_cigar
    (691, 694, "LevMyers", ""),
    (702, 703, "preprocessRawSeq", '''	gpu_aligner->preprocessRawSeq(sequence, sequence_len, mapping_desc.mapping.direction);
	size_t wamd_t = gpu_aligner->dynamicProgramming(start_pos, length, mapping_desc.err_edit_distance);
	gpu_aligner->flush(params->enable_short_indel_refinement);   // per block of reads in production
	const wamd_align_result& wamd_r = gpu_aligner->result(wamd_t);
	bool ok = wamd_r.ok != 0;
	edit_distance = wamd_r.edit_distance;
	relative_end_pos = wamd_r.end_pos;
'''),
    (706, 722, "direct", '''		if (!mapping_desc.ext_cigar)
			mp_ext_cigar->Reserve(mapping_desc.ext_cigar);
		memcpy(mapping_desc.ext_cigar, gpu_aligner->extCigar(wamd_t), (size_t) wamd_r.cigar_len + 1);

		double affine_score = wamd_r.affine_score;
		uint32_t num_events = wamd_r.num_events;
		ref_pos_t relative_begin_pos = wamd_r.begin_pos;
'''),
    (727, 735, "enable_short_indel_refinement", '''		if (wamd_r.refined)
		{
			++stat_refined_short_indels_mapping;
			mapping_desc.err_edit_distance = (uint16_t) (mapping_desc.err_edit_distance + 1);
		}
'''),
    # fill_cigar_with_clipping (755-821): a mapping the matcher found, located again by the hashing rescue
    (771, 774, "preprocessRawSeq", '''	gpu_softclip->preprocessRawSeq(sequence, sequence_len, mapping_desc.mapping.direction);
	const size_t wamd_ct = gpu_softclip->match(examined_start_pos, examined_size, HashTable::hash_len);
	gpu_softclip->flush();		// per block of reads in production

	if (gpu_softclip->result(wamd_ct, relative_begin_pos, left_clipping, right_clipping,
		left_match, right_match, del_size, no_mismatches))
'''),
    (783, 797, "direct", '''			if (!mapping_desc.ext_cigar)
				mp_ext_cigar->Reserve(mapping_desc.ext_cigar);
			memcpy(mapping_desc.ext_cigar, gpu_softclip->extCigar(wamd_ct), (size_t) gpu_softclip->result(wamd_ct).cigar_len + 1);
			double affine_score = gpu_softclip->result(wamd_ct).affine_score;
'''),
    # fill_cigar_with_indel_matching (824-852): GetExtCigarNew of the matcher's record
    (831, 837, "PreprocessRawSeq", '''	gpu_indel_cigar->PreprocessRawSeq(sequence, sequence_len, mapping_desc.mapping.direction);
	const size_t wamd_it = gpu_indel_cigar->GetExtCigarNew(mapping_desc.mapping, examined_start_pos);
	gpu_indel_cigar->flush();		// per block of reads in production
	const wamd_indel_cigar_result& wamd_ir = gpu_indel_cigar->result(wamd_it);

	if (!mapping_desc.ext_cigar)
		mp_ext_cigar->Reserve(mapping_desc.ext_cigar);
	if (wamd_ir.ok)
	{
		memcpy(mapping_desc.ext_cigar, gpu_indel_cigar->extCigar(wamd_it), (size_t) wamd_ir.cigar_len + 1);
		mapping_desc.score = wamd_ir.score;
		mapping_desc.score_type = ScoringType::Affine;
		mapping_desc.err_edit_distance = wamd_ir.err_edit_distance;
		mapping_desc.num_events = wamd_ir.num_events;
		mapping_desc.ref_length = wamd_ir.ref_length;
		mapping_desc.left_clipping = wamd_ir.left_clipping;
		mapping_desc.right_clipping = wamd_ir.right_clipping;
		mapping_desc.method = (MatchingMethod::Enumeration) wamd_ir.method;
	}
	if (wamd_ir.ok)
'''),
]

# ---------------------------------------------------------------------------------------------------------
# The per-bin form
HEADER_CORE_BIN = [
    (28, 28, "LevMyers.h", """#include "LevMyers.h"
#include "gpu_verifier.hpp"
"""),
    (109, 109, "indel_matching_rc", """	CIndelMatching* indel_matching_rc;
	// MI355X verification path, per-bin form (INTEGRATION.md section 3)
	whisper_amd::GpuReference* gpu_ref;
	whisper_amd::GpuSaBin* gpu_bin;
	whisper_amd::GpuIndelMatching* gpu_indel_dir;
	whisper_amd::GpuIndelMatching* gpu_indel_rc;
	struct wamd_strand_t {
		bool enqueued = false, known_range = false;
		uint32_t mismatch = 0;                           // the host path's answer (no range; a malicious group in sensitive mode) ...
		std::vector<candidate_mapping_t> positions;      // ... and what it pushed
		std::vector<size_t> indel_tickets;               // sensitive mode: one per enqueued entry
	};
	struct wamd_read_t {
		read_id_t id;
		uchar_t* data;
		uint32_t size, best_error, bin_id, max_mismatches, handle;
		bool searched = false, duplicate = false;
		wamd_strand_t strand[2];                         // [0] direct, [1] other strand
	};
	std::vector<wamd_read_t> wamd_bin_reads;
	wamd_strand_t* wamd_cur_strand;
	uint32_t wamd_consume(wamd_read_t& wr, bool dir_gen_flag);
	void wamd_finish_read(wamd_read_t& wr, uint32_t& dir_mismatch, uint32_t& rc_mismatch, int64_t& no_of_reads);
"""),
]

FIND_ARGS_DIR = "data, data_sft, size, matching_part_is_new, sa_dir_part, sa_dir_size, &sa_dir_last_index, &sa_dir_start_index, 1, &dir_inside_malicious_group, &dir_aux_struct_valid, max_mismatches"
FIND_ARGS_RC = "rc_data, rc_data_sft, size, matching_part_is_new, sa_rc_part, sa_rc_size, &sa_rc_last_index, &sa_rc_start_index, 0, &rc_inside_malicious_group, &rc_aux_struct_valid, max_mismatches"

CORE_CPP_BIN = [
    (191, 191, "indel_matching_rc->SetReference", """	indel_matching_rc->SetReference(reference->GetData(), 0, 0);
	{
		wamd_indel_params ip{};
		ip.scoring.match = params->match_score;
		ip.scoring.mismatch = params->mismatch_score;
		ip.scoring.gap_ins_open = params->gap_ins_open;
		ip.scoring.gap_ins_extend = params->gap_ins_extend;
		ip.scoring.gap_del_open = params->gap_del_open;
		ip.scoring.gap_del_extend = params->gap_del_extend;
		ip.scoring.clipping = params->clipping_score;
		ip.min_clipped_factor = params->min_clipped_factor;
		ip.max_indel_bound = params->max_approx_indel_len;
		gpu_ref = new whisper_amd::GpuReference(0, reference->GetData(), ((uint64_t) sa_dir->GetRefSize() + 1) / 2,
			sa_dir->GetRefSize(), start_pos);
		gpu_ref->setMaxTextLen(max_ref_len);
		gpu_bin = new whisper_amd::GpuSaBin(*gpu_ref, lev_alg_thr);
		gpu_indel_dir = new whisper_amd::GpuIndelMatching(*gpu_ref, ip);
		gpu_indel_rc = new whisper_amd::GpuIndelMatching(*gpu_ref, ip);
	}
"""),
    # process_reads_init_stage (the first stage's bins, 388-777): read by read - the same calls, one flush per read
    (635, 636, "findMismatches_dir_and_rc", """					{
						wamd_read_t wr;
						wr.searched = true;
						wr.size = size;
						wr.max_mismatches = stage_major;
						gpu_bin->beginBin(start_pos, end_pos, min_read_len);
						wr.handle = gpu_bin->addRead(data, rc_data, size, stage_major);
						if (sensitive_mode && enable_mapping_indels)	// (the reference's matcher objects still hold the last read of process_reads here)
						{
							gpu_indel_dir->Preprocess(data, size, genome_t::direct);
							gpu_indel_rc->Preprocess(data, size, genome_t::rev_comp);
						}
						wamd_cur_strand = &wr.strand[0];
						wr.strand[0].mismatch = findMismatches_dir_and_rc(%s);
						if (!wr.strand[0].enqueued)
							wr.strand[0].positions = dir_match_positions;
						wamd_cur_strand = &wr.strand[1];
						wr.strand[1].mismatch = findMismatches_dir_and_rc(%s);
						if (!wr.strand[1].enqueued)
							wr.strand[1].positions = rc_match_positions;
						gpu_bin->flush();
						if (sensitive_mode && enable_mapping_indels)
						{
							gpu_indel_dir->flush();
							gpu_indel_rc->flush();
						}
						dir_match_positions.clear();
						rc_match_positions.clear();
						dir_mismatch = wamd_consume(wr, true);
						rc_mismatch = wamd_consume(wr, false);
					}
""" % (FIND_ARGS_DIR.replace("max_mismatches", "stage_major"), FIND_ARGS_RC.replace("max_mismatches", "stage_major"))),
    # the bin begins
    (853, 853, "reads_deliverer->Pop", """	gpu_bin->beginBin(start_pos, end_pos, min_read_len);
	wamd_bin_reads.clear();
	while (reads_deliverer->Pop(id, data, data_sft, size, best_error))//id number was assigned internally
"""),
    (871, 872, "indel_matching_dir->Preprocess", """				gpu_indel_dir->Preprocess(data, size, genome_t::direct);
				gpu_indel_rc->Preprocess(data, size, genome_t::rev_comp);
"""),
    # a new read: its record, and both strands searched - the loops enqueue into it
    (967, 968, "findMismatches_dir_and_rc", """				wamd_bin_reads.emplace_back();
				{
					wamd_read_t& wr = wamd_bin_reads.back();
					wr.searched = true;
					wr.handle = gpu_bin->addRead(data, rc_data, size, max_mismatches);
					wamd_cur_strand = &wr.strand[0];
					wr.strand[0].mismatch = findMismatches_dir_and_rc(%s);
					if (!wr.strand[0].enqueued)
						wr.strand[0].positions = dir_match_positions;
					wamd_cur_strand = &wr.strand[1];
					wr.strand[1].mismatch = findMismatches_dir_and_rc(%s);
					if (!wr.strand[1].enqueued)
						wr.strand[1].positions = rc_match_positions;
				}
""" % (FIND_ARGS_DIR, FIND_ARGS_RC)),
    # what the loop did with the answers (973-1014) waits for the flush; a repeated read shares the answers of the read before it
    (973, 1014, "dir_mismatch <= max_mismatches", """			if (!newly_read_data)
			{
				wamd_bin_reads.emplace_back();
				wamd_bin_reads.back().searched = true;
				wamd_bin_reads.back().duplicate = true;
			}
"""),
    (1017, 1036, "pass_read", """		if (!(size >= end_pos))
			wamd_bin_reads.emplace_back();
		{
			wamd_read_t& wr = wamd_bin_reads.back();
			wr.id = id;
			wr.data = data;
			wr.size = size;
			wr.best_error = best_error;
			wr.bin_id = bin_id;
			wr.max_mismatches = max_mismatches;
		}
"""),
    # behind the bin's read loop: one flush, then every read in its turn
    (1038, 1037, "", """
	gpu_bin->flush();
	if (sensitive_mode && enable_mapping_indels)
	{
		gpu_indel_dir->flush();
		gpu_indel_rc->flush();
	}
	dir_mismatch = error_unknown;
	rc_mismatch = error_unknown;
	for (auto& wr : wamd_bin_reads)
		wamd_finish_read(wr, dir_mismatch, rc_mismatch, no_of_reads);
	wamd_bin_reads.clear();
"""),
]


def enqueue_bin(indent):
    t = indent
    return (
        t + "gpu_bin->candidate(sa_part[i], dir_gen_flag);\n" +
        t + "if (sensitive_mode && enable_mapping_indels)\n" +
        t + "	wamd_cur_strand->indel_tickets.push_back(dir_gen_flag\n" +
        t + "		? gpu_indel_dir->Match(sa_part[i] - start_pos, max_approx_indel_len, start_pos, end_pos - start_pos, max_mismatches)\n" +
        t + "		: gpu_indel_rc->Match(ref_size - sa_part[i] - pattern_len + start_pos, max_approx_indel_len, pattern_len - end_pos, end_pos - start_pos, max_mismatches));\n")


CONSUME_BIN = """
// *** MI355X path, per-bin form: the answers of one read's strand, in enumeration order (INTEGRATION.md section 3)
uint32_t CMappingCore::wamd_consume(wamd_read_t& wr, bool dir_gen_flag)
{
	wamd_strand_t& s = wr.strand[dir_gen_flag ? 0 : 1];
	std::vector<candidate_mapping_t>& out = dir_gen_flag ? dir_match_positions : rc_match_positions;
	if (!s.enqueued)		// no range with this anchor, or a malicious group that is not searched (:29-35): the host path has answered already
	{
		out = s.positions;
		return s.mismatch;
	}
	const uint32_t max_mismatches = wr.max_mismatches;
	const uint32_t pattern_len = wr.size;
	uint32_t best_match = max_mismatches + 1;
	whisper_amd::GpuIndelMatching* gpu_indel = dir_gen_flag ? gpu_indel_dir : gpu_indel_rc;
	const uint32_t n = gpu_bin->count(wr.handle, dir_gen_flag);

	for (uint32_t j = 0; j < n; ++j)
	{
		candidate_mapping_t candidate_mapping;
		const uint32_t sa = gpu_bin->saEntry(wr.handle, dir_gen_flag, j);
		const uint32_t cur_match = gpu_bin->curMatch(wr.handle, dir_gen_flag, j);

		if (cur_match <= max_mismatches)
		{
			if (dir_gen_flag)
				candidate_mapping = candidate_mapping_t::construct<mapping_type_t::lev>(
					s.known_range ? sa - start_pos : sa - CUR_OFFSET, genome_t::direct, cur_match);
			else
				candidate_mapping = candidate_mapping_t::construct<mapping_type_t::lev>(
					ref_size - (sa - CUR_OFFSET) - 1, genome_t::rev_comp, cur_match);
		}
		if (!s.indel_tickets.empty())
		{
			candidate_mapping_t candidate_mapping_indel;
			if (gpu_indel->result(s.indel_tickets[j], candidate_mapping_indel))
			{
				candidate_mapping_indel.direction = dir_gen_flag ? genome_t::direct : genome_t::rev_comp;
				if (!dir_gen_flag)
					candidate_mapping_indel.pos += (pattern_len - 1);
				candidate_mapping = candidate_mapping_t::better(candidate_mapping, candidate_mapping_indel);
			}
		}
		if (candidate_mapping.type != mapping_type_t::none && candidate_mapping.penalty < best_match)
			best_match = candidate_mapping.penalty;
		if (candidate_mapping.type != mapping_type_t::none && candidate_mapping.penalty <= max_mismatches)
			out.push_back(candidate_mapping);
	}
	return best_match;
}

// what process_reads does with a read once both strands have answered (mapping_core.cpp:973-1036), behind the flush
void CMappingCore::wamd_finish_read(wamd_read_t& wr, uint32_t& dir_mismatch, uint32_t& rc_mismatch, int64_t& no_of_reads)
{
	uint32_t best_error = wr.best_error;
	const uint32_t max_mismatches = wr.max_mismatches;

	if (wr.searched)
	{
		if (!wr.duplicate)
		{
			dir_match_positions.clear();
			rc_match_positions.clear();
			collected_mappings.Clear(max_no_mappings);
			dir_mismatch = wamd_consume(wr, true);
			rc_mismatch = wamd_consume(wr, false);
		}
		if ((dir_mismatch <= max_mismatches) || (rc_mismatch <= max_mismatches))
		{
			if (dir_mismatch <= max_mismatches)
				for (uint32_t i = 0; i < dir_match_positions.size(); i++)
					collected_mappings.Push(dir_match_positions[i]);
			if (rc_mismatch <= max_mismatches)
				for (uint32_t i = 0; i < rc_match_positions.size(); i++)
					collected_mappings.Push(rc_match_positions[i]);
			while (!collected_mappings.Empty())
			{
				auto x = collected_mappings.PopUnsorted();
				results_collector->Push(wr.id, x);
			}
			uint32_t tmp_best_error = MIN(dir_mismatch, rc_mismatch);
			best_error = MIN(best_error, tmp_best_error);
		}
	}

	bool pass_read = false;
	if (stage_minor < stage_major)
		pass_read = true;
	else if (mapping_mode == mapping_mode_t::first)
	{
		if (best_error > stage_major)
			pass_read = true;
	}
	else if (mapping_mode == mapping_mode_t::second)
	{
		if (best_error >= stage_major)
			pass_read = true;
	}
	if (pass_read)
	{
		reads_collector->Push(wr.bin_id, wr.id, wr.data, wr.size, best_error);
		no_of_reads++;
	}
}
"""

SA_CPP_BIN = [
    (212, 213, "lev_alg_thr", """	wamd_cur_strand->enqueued = true;
	wamd_cur_strand->known_range = false;
"""),
    (217, 344, "candidate_mapping_t", enqueue_bin("\t\t")),
    (376, 377, "lev_alg_thr", """	wamd_cur_strand->enqueued = true;
	wamd_cur_strand->known_range = true;
"""),
    (383, 504, "candidate_mapping_t", enqueue_bin("\t\t\t")),
    (514, 513, "", CONSUME_BIN),
]



# search_in_malicious_group_with_DP (mapping_core_dp.cpp:23-201): the q-gram filter of the group stays on the host (its
# lists are the host's); what it lets through goes the way of every other suffix-array entry of the bin.  The CF
# vectors the loop compares (:132, :161) are those of calculate_CF_128b_SEE_for_dir_and_rc_gen, the ones the device
# filter computes; the positions pushed (:192, :195) are the unknown-range loop's expressions.
DP_CPP_BIN = [
    (122, 123, "lev_alg_thr", """	wamd_cur_strand->enqueued = true;
	wamd_cur_strand->known_range = false;
"""),
    (125, 198, "sum_of_localizations", """	for (auto p = (*sum_of_localizations).begin(); p != q; ++p)
		gpu_bin->candidate(sub_SA[*p], dir_gen_flag);		// (identical_sequences_* is all false: its only writer is compiled out, :305-326)
"""),
]


def apply(src_lines, edits):
    out = list(src_lines)
    for first, last, ident, repl in sorted(edits, key=lambda e: -e[0]):
        if last >= first and ident and ident not in out[first - 1]:
            raise SystemExit("anchor %r not in line %d: %r - another version of the reference?" % (ident, first, out[first - 1].strip()[:60]))
        new = repl.splitlines(keepends=True)
        if last >= first:
            out[first - 1:last] = new
        else:
            out[first - 1:first - 1] = new
    return out


def check(name, patched, keep):
    scratch = tempfile.mkdtemp(prefix="wamd_patch_%s_" % name)
    assert not os.path.abspath(scratch).startswith(REPO + os.sep)
    try:
        for d in os.listdir(REF):
            s = os.path.join(REF, d)
            if d != "mapper":
                os.symlink(s, os.path.join(scratch, d))
        os.mkdir(os.path.join(scratch, "mapper"))
        for f in os.listdir(os.path.join(REF, "mapper")):
            s = os.path.join(REF, "mapper", f)
            t = os.path.join(scratch, "mapper", f)
            if f in patched:
                with open(s, encoding="latin-1") as fh:
                    lines = fh.readlines()
                with open(t, "w", encoding="latin-1") as fh:
                    fh.writelines(apply(lines, patched[f]))
            else:
                os.symlink(s, t)
        flags = ["-O0", "-m64", "-std=c++17", "-msse4.2", "-mpopcnt", "-fabi-version=6", "-pthread", "-w", "-fsyntax-only",
                 "-I", os.path.join(scratch, "libs"), "-I", os.path.join(REPO, "whisper_amd", "host"),
                 "-I", os.path.join(REPO, "include")]
        rc = 0
        for f in ("mapping_core_sa.cpp", "mapping_core.cpp", "sam_matching.cpp") + (("mapping_core_dp.cpp",) if "mapping_core_dp.cpp" in patched else ()):
            cmd = [os.environ.get("CXX", "g++")] + flags + [os.path.join(scratch, "mapper", f)]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            print("%-10s %-22s %s" % (name, f, "compiles" if r.returncode == 0 else "FAILS"))
            if r.returncode:
                print(r.stdout[-4000:])
                rc = 1
        if keep:
            print("scratch tree kept at", scratch)
        return rc
    finally:
        if not keep:
            shutil.rmtree(scratch, ignore_errors=True)


def main():
    keep = "--keep" in sys.argv
    if not os.path.isdir(REF):
        print("reference sources absent (%s): nothing to check" % REF)
        return 0
    per_range = {
        "mapping_core.h": HEADER_CORE,
        "mapping_core.cpp": CORE_CPP,
        "mapping_core_sa.cpp": SA_CPP,
        "sam.h": HEADER_SAM,
        "sam_matching.cpp": SAM_MATCHING_CPP,
    }
    per_bin = {
        "mapping_core.h": HEADER_CORE_BIN,
        "mapping_core.cpp": CORE_CPP_BIN,
        "mapping_core_sa.cpp": SA_CPP_BIN,
        "mapping_core_dp.cpp": DP_CPP_BIN,
        "sam.h": HEADER_SAM,
        "sam_matching.cpp": SAM_MATCHING_CPP,
    }
    return check("per-range", per_range, keep) | check("per-bin", per_bin, keep)


if __name__ == "__main__":
    sys.exit(main())
