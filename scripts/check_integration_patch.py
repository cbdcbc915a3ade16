#!/usr/bin/env python3
"""Compile the INTEGRATION.md patch against the reference's own headers.

What this does: builds a scratch tree OUTSIDE the repository (a temporary directory) whose
files are symlinks into /root/reference/src, except the five files the patch touches, which
are copied there and edited at the lines INTEGRATION.md sections 3 and 4 name; then compiles the
patched translation units with g++ (-fsyntax-only, the flags of the reference's makefile)
against whisper_amd/host/gpu_verifier.hpp and include/whisper_amd.h.

This file holds only the replacement lines and, per edit, the line range it replaces and one
identifier that must occur in the first replaced line (so a reference of another version is
detected instead of being patched at the wrong place).  Nothing of the reference is stored here
and nothing is written under /root/reference.

    python scripts/check_integration_patch.py [--keep]     exit 0 = both objects compile

The patch is the per-range form: every candidate of one suffix-array range is enqueued, the
range is flushed, the results are consumed in the order the loop enqueued them - the function
still returns best_match to its caller, so nothing outside it changes.  Flushing once per bin
instead moves the consume block behind the bin loop of mapping_core.cpp:560-1050.
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
'''),
]

SAM_MATCHING_CPP = [
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


def main():
    keep = "--keep" in sys.argv
    if not os.path.isdir(REF):
        print("reference sources absent (%s): nothing to check" % REF)
        return 0
    scratch = tempfile.mkdtemp(prefix="wamd_patch_")
    assert not os.path.abspath(scratch).startswith(REPO + os.sep)
    try:
        for d in os.listdir(REF):
            s = os.path.join(REF, d)
            if d != "mapper":
                os.symlink(s, os.path.join(scratch, d))
        os.mkdir(os.path.join(scratch, "mapper"))
        patched = {
            "mapping_core.h": HEADER_CORE,
            "mapping_core.cpp": CORE_CPP,
            "mapping_core_sa.cpp": SA_CPP,
            "sam.h": HEADER_SAM,
            "sam_matching.cpp": SAM_MATCHING_CPP,
        }
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
        for f in ("mapping_core_sa.cpp", "mapping_core.cpp", "sam_matching.cpp"):
            cmd = [os.environ.get("CXX", "g++")] + flags + [os.path.join(scratch, "mapper", f)]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            print("%-22s %s" % (f, "compiles" if r.returncode == 0 else "FAILS"))
            if r.returncode:
                print(r.stdout[-4000:])
                rc = 1
        if keep:
            print("scratch tree kept at", scratch)
        return rc
    finally:
        if not keep:
            shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
