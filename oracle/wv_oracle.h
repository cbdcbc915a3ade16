/*
 * wv_oracle.h - CPU restatement of Whisper 2.0.2's read-verification path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under whisper_amd/ may include, link or
 * load this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker.
 *
 * Parity pin: every function here is checked, on seeded random tasks and on the
 * committed fixtures under tests/golden/, against the reference's own objects
 * compiled from /root/reference/src into oracle/_ref/libwhisper_ref.so
 * (oracle/Makefile, oracle/ref_shim.cpp; tests/test_oracle_vs_reference.py,
 * tests/golden/make_golden.py).
 */
#ifndef WV_ORACLE_H
#define WV_ORACLE_H

#include <stdint.h>
#include "../include/whisper_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define WO_MAX_WORDS 16 /* 1024 symbols */

/* Pattern-match table of one preprocessed read: LevMyers64's bp_PM
 * (LevMyers.h:80-82) plus the two scalars the DP reads. */
typedef struct wo_pattern {
    uint64_t pm[16][WO_MAX_WORDS];
    uint32_t seq_len;
    uint8_t last_symbol;
} wo_pattern;

/* Bit-vector matrix of one dynamicProgramming call, column-major and UNSKEWED:
 * cell [col * words + w], col = 0..n_cols.  The skewed variants store lane w of
 * iteration j = col + w (LevMyers256.hpp:276); this layout holds the same values. */
typedef struct wo_cell {
    uint64_t D0, VP, VN, HN, HP;
} wo_cell;

void wo_pattern_init(wo_pattern* p);
/* LevMyers64::preprocess, LevMyers64.cpp:167-196 (nibble-packed read). */
void wo_preprocess(wo_pattern* p, const uint8_t* seq, uint32_t seq_len, int rev_comp);
/* LevMyers64::preprocessRawSeq, LevMyers64.cpp:199-220 (ASCII read; last_symbol untouched). */
void wo_preprocess_raw(wo_pattern* p, const uint8_t* seq, uint32_t seq_len, int rev_comp);

/* variant by read length, mapping_core_sa.cpp:207-210. */
int wo_variant_for_len(uint32_t seq_len);

/* dynamicProgramming of the three variants.  matrix may be NULL; otherwise it
 * must hold (W + 2) * words cells, words = (seq_len - 1) / 64 + 1.
 * Returns 1/0 like the reference's bool, -1 when the reference would throw. */
int wo_myers_dp(const wo_pattern* p, int variant, const uint8_t* genome, uint32_t ref_pos,
                uint32_t W, uint32_t k, uint32_t max_text_len, uint32_t* pos, uint32_t* ed,
                wo_cell* matrix);

/* getExtCigar, LevMyers64.cpp:392-520 (identical in the 128/256 variants up to
 * the matrix addressing).  read_codes is 1-based (guard at [0]), ref_codes 0-based. */
uint32_t wo_ext_cigar(const wo_cell* matrix, uint32_t words, uint8_t* ext_cigar,
                      const uint8_t* ref_codes, const uint8_t* read_codes, uint32_t pos,
                      uint32_t edit_dist, uint32_t seq_len, const wamd_scoring* sc,
                      double* affine_score, uint32_t* num_events);

/* CSamGenerator::refine_ext_cigar, sam_matching.cpp:858-924. */
int wo_refine_ext_cigar(uint8_t* ext_cigar, const uint8_t* ref_codes, uint32_t* edit_distance,
                        uint32_t* num_events, double* score, const wamd_scoring* sc);

/* copy_direct / convert_to_rev_comp / ref_copy_direct, sam.cpp:2296-2375. */
void wo_read_codes(uint8_t* dest, const uint8_t* ascii, uint32_t len, int rev_comp);
void wo_ref_codes(uint8_t* dest, const uint8_t* genome, uint32_t pos, uint32_t len);

/* CEditDist::LevDiag, edit_dist.cpp:106-344. */
uint32_t wo_levdiag(const uint8_t* genome, uint32_t ref_size, uint32_t cur_offset,
                    const uint8_t* pattern, uint32_t pattern_len, uint32_t position_in_text,
                    uint32_t fixed_left, uint32_t fixed_right, uint32_t max_mismatches,
                    int dir_gen_flag);

/* ---- batch drivers over the C-ABI task structs (one verifier "object" per call) ---- */
/* reads: data/offsets/lens as wamd_reads_upload takes them. */
int wo_distance_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                      const uint32_t* lens, int encoding, const wamd_dist_task* tasks,
                      uint32_t n, int last_symbol, uint32_t max_text_len,
                      wamd_dist_result* results);
int wo_levdiag_batch(const uint8_t* genome, uint32_t ref_size, uint32_t cur_offset,
                     const uint8_t* reads, const uint64_t* offsets, const uint32_t* lens,
                     const wamd_levdiag_task* tasks, uint32_t n, uint32_t* results);
int wo_align_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                   const uint32_t* lens, int encoding, const wamd_dist_task* tasks, uint32_t n,
                   int last_symbol, int refine, uint32_t max_text_len, const wamd_scoring* sc,
                   wamd_align_result* results, uint8_t* cigars, uint32_t cigar_stride);

/* ---- long-indel / clipping matcher, its ext-CIGAR, boundary clipping, wire records ---- */

/* CIndelMatching::Preprocess / PreprocessRawSeq (indels.cpp:960-991): one code per symbol. */
void wo_indel_query(uint8_t* query, const uint8_t* seq, uint32_t len, int encoding, int rev_comp);

/* CIndelMatching::Match, indels.cpp:125-461.  query: unpacked read codes. */
int wo_indel_match(const uint8_t* genome, const uint8_t* query, uint32_t query_len, uint32_t ref_pos,
                   uint32_t max_indel_size, uint32_t fixed_segment_pos, uint32_t fixed_segment_size,
                   uint32_t max_no_mismatches, const wamd_indel_params* prm, wamd_mapping* out);

/* CIndelMatching::GetExtCigarNew, indels.cpp:938-957 (+ 680-935). */
int wo_indel_ext_cigar(const uint8_t* genome, const uint8_t* query, uint32_t query_len,
                       const wamd_indel_cigar_task* t, const wamd_scoring* sc,
                       wamd_indel_cigar_result* r, uint8_t* ext_cigar);

/* CSoftClipping::clipBoundaryPositions, soft_clipping.cpp:827-994. */
void wo_clip_boundary(wamd_clip_record* rec, uint8_t* ext_cigar, const wamd_scoring* sc);

/* candidate_mapping_t::serialize, results.cpp:161-217; returns the record size. */
uint32_t wo_serialize_mapping(const wamd_mapping* m, int direct, uint8_t* out);

int wo_indel_match_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                         const uint32_t* lens, int encoding, const wamd_indel_task* tasks, uint32_t n,
                         const wamd_indel_params* prm, wamd_mapping* results);
int wo_indel_ext_cigar_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                             const uint32_t* lens, int encoding, const wamd_indel_cigar_task* tasks,
                             uint32_t n, const wamd_scoring* sc, wamd_indel_cigar_result* results,
                             uint8_t* cigars, uint32_t cigar_stride);
int wo_clip_boundary_batch(uint32_t n, wamd_clip_record* records, uint8_t* cigars,
                           uint32_t cigar_stride, const wamd_scoring* sc);
int wo_serialize_batch(uint32_t n, const wamd_mapping* m, const uint8_t* direction, uint8_t* out,
                       uint8_t* sizes);

/* independent consistency check of ext-CIGARs against read and genome (tests only) */
int wo_check_ext_cigar_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                             const uint32_t* lens, const wamd_dist_task* tasks, uint32_t n,
                             const wamd_align_result* results, const uint8_t* cigars, uint32_t cigar_stride,
                             uint8_t* fail);

/* CF pre-filter: mapping_core_dp.cpp:349-483, mapping_core.h:260-263 */
int wo_cf_genome(const uint8_t* ref_ptr, uint32_t ref_size, uint32_t cur_offset, uint32_t fixed_left,
                 uint32_t fixed_right, uint32_t pattern_len, uint32_t position, int dir_gen_flag, int8_t* cf);
int wo_cf_pattern(const uint8_t* pattern, const uint8_t* pattern_sft, uint32_t pattern_len, uint32_t fixed_left,
                  uint32_t fixed_right, int8_t* cf);
uint32_t wo_cf_difference(const int8_t* a, const int8_t* b);
int wo_cf_batch(const uint8_t* genome, uint32_t ref_size, uint32_t cur_offset, const uint8_t* reads,
                const uint64_t* offsets, const uint32_t* lens, const wamd_cf_task* tasks, uint32_t n,
                uint32_t* results);

/* ext-CIGAR -> SAM CIGAR / BAM operations / MD:Z: sam.cpp:805-928, 932-1051 */
void wo_sam_cigar(const uint8_t* ext_cigar, uint8_t* cigar, uint8_t* mdz, uint32_t* cigar_bin,
                  wamd_sam_cigar_len* len);
int wo_sam_cigar_batch(uint32_t n, const uint8_t* ext_cigars, uint32_t ext_stride, uint8_t* cigars,
                       uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride, uint32_t* cigar_bin,
                       uint32_t bin_stride, wamd_sam_cigar_len* lens);

/* clipIllegalPositions (soft_clipping.cpp:592-729) and clipOverlappingPairs with clipLeft /
 * clipRight (997-1151).  Both return 0, or 1 when the record does not describe its ext-CIGAR
 * (where the reference would run off the string); stride bounds every access. */
int wo_clip_illegal(wamd_clip_record* rec, const wamd_clip_illegal_task* t, uint8_t* ext_cigar, uint32_t stride,
                    const wamd_scoring* sc);
int wo_clip_overlapping_pair(wamd_clip_record* rec2, const wamd_clip_pair_task* t, uint8_t* ext_cigar_first,
                             uint8_t* ext_cigar_second, uint32_t stride);
int wo_clip_illegal_batch(uint32_t n, const wamd_clip_illegal_task* tasks, wamd_clip_record* records,
                          uint8_t* cigars, uint32_t cigar_stride, const wamd_scoring* sc);
int wo_clip_pairs_batch(uint32_t n_pairs, const wamd_clip_pair_task* tasks, wamd_clip_record* records,
                        uint8_t* cigars, uint32_t cigar_stride);

/* CSoftClipping::{preprocess, preprocessRawSeq, match, getExtCigar}, soft_clipping.cpp:139-589 */
uint32_t wo_softclip_table_size(uint32_t max_query_len);
int wo_softclip(const uint8_t* genome, const uint8_t* query, uint32_t query_len, uint32_t ref_pos,
                uint32_t max_distance_in_ref, uint32_t min_exact_len, const wamd_softclip_params* prm,
                const wamd_scoring* sc, wamd_softclip_result* res, uint8_t* ext_cigar);
int wo_softclip_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets, const uint32_t* lens,
                      int encoding, const wamd_softclip_task* tasks, uint32_t n, const wamd_softclip_params* prm,
                      const wamd_scoring* sc, wamd_softclip_result* results, uint8_t* cigars, uint32_t cigar_stride);

/* verified candidates -> candidate_mapping per task (mapping_core_sa.cpp:241-344) -> the bytes
 * CMappingResultsCollector leaves in a group buffer (mapping_core.cpp:976-997, results.cpp:409-452, 546-598) */
int wo_commit_batch(uint32_t n, const wamd_commit_task* tasks, const wamd_mapping* indel, wamd_mapping* out);
int wo_result_groups_batch(uint32_t n_tasks, const wamd_commit_task* tasks, const wamd_mapping* committed,
                           uint32_t n_reads, const uint32_t* read_first, const uint64_t* read_ids,
                           const wamd_group_params* prm, uint8_t* out, uint32_t out_capacity, uint32_t* out_bytes);

/* store_mapped_pair_reads' finish of the chosen pair: fill_cigar, the three clipping passes,
 * extract_cigar_and_mdz (sam.cpp:1871-1900, 1992-2007; sam_matching.cpp:667-852). */
int wo_finish_pairs_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets, const uint32_t* lens,
                          const wamd_finish_task* tasks, uint32_t n_pairs, const wamd_finish_params* prm,
                          const wamd_scoring* sc, wamd_finish_result* results, uint8_t* ext_cigars,
                          uint32_t ext_stride, uint8_t* cigars, uint32_t cigar_stride, uint8_t* mdz,
                          uint32_t mdz_stride);

#ifdef __cplusplus
}
#endif
#endif
