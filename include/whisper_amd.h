/*
 * whisper_amd.h - C-ABI of the MI355X read-verification path.
 *
 * This is the drop-in boundary for Whisper 2.0.2's verification stage
 * (SURVEY.md section 8b).  The reference has no FFI: its boundary is a set of
 * per-thread, one-candidate-per-call C++ objects.  Every entry point below
 * names the reference interface (file:line under /root/reference/src) whose
 * work it performs for a whole batch of candidates.
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types, no exceptions;
 *   - every function returns a wamd_status (0 = WAMD_OK);
 *   - "*_dev" entry points take DEVICE pointers and a hipStream_t passed as
 *     void*; they are asynchronous on that stream and allocate nothing;
 *   - the other entry points take HOST pointers, stage through the context's
 *     buffers and return when the results are in the caller's memory; they may be
 *     called from several threads on one context (a lock inside the context
 *     serialises them - threads that want to overlap use one context each over
 *     wamd_ctx_create_dev, or the "*_dev" forms on streams of their own);
 *   - symbol codes are the reference's: A 0, C 1, G 2, T 3, N_read 4,
 *     N_ref 5 (common/defs.h:55-60); the genome is the reference's own
 *     packed layout: two 4-bit symbols per byte, even position in the HIGH
 *     nibble (index/ref_seq.cpp:248-281, common/reference.cpp:33-55).
 */
#ifndef WHISPER_AMD_H
#define WHISPER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAMD_API __attribute__((visibility("default")))

typedef enum wamd_status {
    WAMD_OK = 0,
    WAMD_ERR_INVALID_ARG = 1,   /* null pointer, bad enum, inconsistent sizes            */
    WAMD_ERR_HIP = 2,           /* a HIP runtime call failed; see wamd_last_error()       */
    WAMD_ERR_NO_DEVICE = 3,     /* no gfx950 device / HIP runtime unavailable            */
    WAMD_ERR_TOO_LONG = 4,      /* read longer than WAMD_MAX_READ_LEN                     */
    WAMD_ERR_TEXT_LEN = 5,      /* max_distance_in_ref == 0 or > max_text_len: the        */
                                /* reference throws std::runtime_error here               */
                                /* (LevMyers64.cpp:244-252, LevMyers256.hpp:102-110)      */
    WAMD_ERR_RANGE = 6,         /* a task's window leaves the packed genome              */
    WAMD_ERR_NO_MEMORY = 7,
    WAMD_ERR_CAPACITY = 8       /* no-alloc entry point: batch beyond its static scratch */
} wamd_status;

/* Verifier variants.  AUTO follows the reference's choice by read length:
 * len < 128 -> M128, len < 256 -> M256, else M64
 * (mapper/mapping_core_sa.cpp:207-210, mapper/sam_matching.cpp:331-338, 691-694).
 * The three differ in column range, tie-break and early exit, so end positions
 * are only bit-exact when the variant matches the reference's. */
enum {
    WAMD_VARIANT_AUTO = 0,
    WAMD_VARIANT_M64 = 1,   /* LevMyers64::dynamicProgramming   LevMyers64.cpp:223-389   */
    WAMD_VARIANT_M128 = 2,  /* LevMyers128<>::dynamicProgramming LevMyers128.hpp:80-234  */
    WAMD_VARIANT_M256 = 3   /* LevMyers256<>::dynamicProgramming LevMyers256.hpp:83-249  */
};

/* Read encodings accepted by wamd_reads_upload. */
enum {
    WAMD_READS_NIBBLE = 0,  /* 4-bit codes, first symbol in the high nibble:            */
                            /* what LevMyers::preprocess / CEditDist::LevDiag /          */
                            /* CIndelMatching::Preprocess take in the mapping stage      */
    WAMD_READS_ASCII = 1    /* 'A','C','G','T','N' (anything else -> code 6):           */
                            /* what preprocessRawSeq takes in post-processing            */
};

#define WAMD_MAX_READ_LEN 1024u        /* symbols; 16 x 64-bit words per bit-vector      */
#define WAMD_TASK_FLAG_REV_COMP 1u     /* genome_t::rev_comp orientation of the read     */
/* read_idx of a slot that holds no task: wamd_align[_dev] and wamd_indel_ext_cigar[_dev] skip it
 * (ok = 0, edit_distance = 0xffffffff / ok = 0; its ext-CIGAR slot is left alone by
 * wamd_indel_ext_cigar and emptied by wamd_align) and raise no error flag.  Lets a caller keep
 * task i, result i and slot i aligned across kernels that each serve a subset. */
#define WAMD_READ_NONE 0xffffffffu
#define WAMD_LAST_SYMBOL_FROM_READ (-1)

/* scoring_t (common/types.h:574-612); defaults: match 1, mismatch -5,
 * gap open -5 / extend -0.4 (ins and del), clipping -6 (mapper/params.h:111-192). */
typedef struct wamd_scoring {
    double match;
    double mismatch;
    double gap_ins_open;
    double gap_ins_extend;
    double gap_del_open;
    double gap_del_extend;
    double clipping;
} wamd_scoring;

/* One LevMyers::dynamicProgramming call (LevMyers.h:51):
 * dynamicProgramming(ref_pos, max_distance_in_ref, max_mate_edit_distance, pos, ed)
 * on the read read_idx of a wamd_reads batch, preprocessed in the orientation
 * given by flags (preprocess / preprocessRawSeq, LevMyers64.cpp:167-220). */
typedef struct wamd_dist_task {
    uint32_t read_idx;
    uint32_t ref_pos;              /* raw genome position of the window start          */
    uint32_t max_distance_in_ref;  /* W: window length                                  */
    uint16_t max_ed;               /* k: max_mate_edit_distance / max_mismatches        */
    uint8_t variant;               /* WAMD_VARIANT_*                                    */
    uint8_t flags;                 /* WAMD_TASK_FLAG_*                                  */
} wamd_dist_task;

/* What dynamicProgramming hands back: edit_distance is always the minimum seen
 * before the variant's early exit (so values > k are reproduced as well);
 * pos is the 1-based end column, 0 when edit_distance > k (return false). */
typedef struct wamd_dist_result {
    uint32_t edit_distance;
    uint32_t pos;
} wamd_dist_result;

/* One CEditDist::LevDiag call (mapper/edit_dist.h:50-51, edit_dist.cpp:106-344):
 * LevDiag(pattern, pattern_len, position_in_text, fixed_left, fixed_right,
 *         max_mismatches, dir_gen_flag) -> errors, or max_mismatches + 1.
 * The reference only calls it below lev_alg_thr (common/defs.h:108,
 * mapping_core_sa.cpp:236-239): max_mismatches <= 6 here, reads nibble-packed. */
typedef struct wamd_levdiag_task {
    uint32_t read_idx;
    uint32_t position_in_text;
    uint16_t fixed_left;
    uint16_t fixed_right;
    uint16_t max_mismatches;
    uint16_t dir_gen_flag;
} wamd_levdiag_task;

/* Result of dynamicProgramming + getExtCigar (+ refine_ext_cigar):
 * LevMyers.h:62-63, LevMyers64.cpp:392-520, LevMyers128.hpp:238-367,
 * LevMyers256.hpp:253-384, sam_matching.cpp:858-924. */
typedef struct wamd_align_result {
    double affine_score;      /* after refinement when refined != 0                    */
    uint32_t ok;              /* dynamicProgramming's bool                              */
    uint32_t edit_distance;   /* as returned by dynamicProgramming (not refined)        */
    uint32_t end_pos;         /* pos                                                    */
    uint32_t begin_pos;       /* getExtCigar's return value                             */
    uint32_t num_events;      /* gap opens (+1 when refined)                            */
    uint32_t cigar_len;       /* bytes of ext-CIGAR, NUL not counted                    */
    uint32_t refined;         /* refine_ext_cigar returned true                         */
    uint32_t reserved;
} wamd_align_result;

/* mapping_type_t (common/defs.h:83-84). */
enum {
    WAMD_MAP_LEV = 0, WAMD_MAP_INDEL1 = 1, WAMD_MAP_INDEL2 = 2, WAMD_MAP_INDEL_CLIPPING = 3,
    WAMD_MAP_CLIPPING_INDEL = 4, WAMD_MAP_MISMATCHES_CLIPPING = 5, WAMD_MAP_CLIPPING_MISMATCHES = 6,
    WAMD_MAP_CLIPPING_CLIPPING = 7, WAMD_MAP_NONE = 8
};

/* One CIndelMatching::Match call (mapper/indels.h:283-284, indels.cpp:125-461):
 * Match(ref_pos, max_indel_size, fixed_segment_pos, fixed_segment_size,
 *       max_no_mismatches, candidate_mapping) on a read preprocessed with
 * Preprocess / PreprocessRawSeq in the orientation given by flags. */
typedef struct wamd_indel_task {
    uint32_t read_idx;
    uint32_t ref_pos;
    uint16_t max_indel_size;
    uint16_t fixed_segment_pos;
    uint16_t fixed_segment_size;
    uint8_t max_no_mismatches;
    uint8_t flags;                 /* WAMD_TASK_FLAG_*                                  */
} wamd_indel_task;

/* candidate_mapping_t (common/types.h:224-318) as plain data.  `found` is Match's
 * bool; the other fields hold whatever Match left in a default-constructed
 * candidate_mapping_t, direction excluded (the caller sets it,
 * mapping_core_sa.cpp:266, 319-320). */
typedef struct wamd_mapping {
    uint32_t found;
    uint32_t type;                 /* WAMD_MAP_*                                        */
    uint32_t pos;
    uint32_t no_mismatches;
    uint32_t matching_len1;
    uint32_t matching_len2;
    int32_t indel_len1;
    int32_t indel_len2;
    uint32_t clipping_len;
    uint32_t penalty;              /* calc_penalty(), results.cpp:327-343               */
} wamd_mapping;

/* Parameters CIndelMatching takes from CParams (indels.cpp:21-32). */
typedef struct wamd_indel_params {
    wamd_scoring scoring;
    double min_clipped_factor;     /* mapper/params.h:158: 1.0                          */
    uint32_t max_indel_bound;      /* wamd_indel_match_dev only: an upper bound of max_indel_size   */
                                   /* over the batch (params->max_approx_indel_len at both call     */
                                   /* sites, mapper/params.h:135: 50); it picks the kernel and sizes */
                                   /* its LDS.  0 = 80.  The host-pointer form scans the tasks.      */
    uint32_t reserved;
} wamd_indel_params;

/* Input of CIndelMatching::GetExtCigarNew (indels.cpp:938-957) as called from
 * fill_cigar_with_indel_matching (sam_matching.cpp:824-852): the record of one
 * mapping, the read and the window start (examined_start_pos). */
typedef struct wamd_indel_cigar_task {
    uint32_t read_idx;
    uint32_t examined_start_pos;
    uint32_t type;                 /* WAMD_MAP_*, not LEV                               */
    uint32_t matching_len1;
    uint32_t matching_len2;
    int32_t indel_len1;
    int32_t indel_len2;
    uint32_t clipping_len;
    uint32_t flags;                /* WAMD_TASK_FLAG_*                                  */
    uint32_t reserved;
} wamd_indel_cigar_task;

/* What GetExtCigarNew leaves in mapping_desc_t. */
typedef struct wamd_indel_cigar_result {
    double score;
    uint32_t ok;
    uint32_t cigar_len;
    uint16_t err_edit_distance;
    uint16_t num_events;
    uint16_t ref_length;
    uint16_t left_clipping;
    uint16_t right_clipping;
    uint16_t method;               /* MatchingMethod::Enumeration (common/types.h)      */
    uint32_t reserved;
} wamd_indel_cigar_result;

/* The fields of mapping_desc_t that CSoftClipping::clipBoundaryPositions
 * (soft_clipping.cpp:827-994) reads and rewrites, next to the ext-CIGAR itself. */
typedef struct wamd_clip_record {
    double score;
    int32_t ref_seq_pos;
    uint16_t ref_length;
    uint16_t err_edit_distance;
    uint16_t left_clipping;
    uint16_t right_clipping;
    uint32_t cigar_len;            /* out: length after clipping                        */
    uint32_t no_clippings;         /* out: sides clipped by this call (0..2)            */
    uint32_t reserved;
} wamd_clip_record;

/* One CF pre-filter decision of the mapping loops (mapping_core_sa.cpp:228-230, 283-285,
 * 392, 444; mapping_core_dp.cpp:132, 161): dinucleotide histograms (16 saturating int8 bins,
 * CF_SSE_lut, mapping_core.cpp:196-217) of the read and of the genome under it, both taken
 * over pattern_len (= min_read_len) symbols outside the anchor [fixed_left, fixed_right);
 * the candidate goes on to the verifier iff their L1 distance <= 4 * max_mismatches.
 *   genome side  calculate_CF_128b_SEE_for_dir_and_rc_gen(fixed_left, fixed_right, pattern_len,
 *                position, CF, dir_gen_flag)                       mapping_core_dp.cpp:349-428
 *   read side    calculate_CF_128b_SSE_for_a_pattern(pattern + read_offset ..., pattern_len,
 *                fixed_left, fixed_right, CF)                      mapping_core_dp.cpp:431-483
 * read_offset: where the pattern_len window starts inside the read (0 on the direct strand,
 * pattern length - min_read_len on the other, mapping_core_diff_len_reads.cpp:118-126).
 * A region of exactly one symbol is rejected (the reference's byte walk underflows there). */
typedef struct wamd_cf_task {
    uint32_t read_idx;
    uint32_t position;            /* position_read_from_SA_array                        */
    uint16_t read_offset;
    uint16_t pattern_len;
    uint16_t fixed_left;
    uint16_t fixed_right;
    uint8_t dir_gen_flag;
    uint8_t flags;                /* WAMD_TASK_FLAG_*: orientation of the read           */
    uint16_t reserved;
} wamd_cf_task;

typedef struct wamd_ctx wamd_ctx;       /* one per (process, device)                  */
typedef struct wamd_reads wamd_reads;   /* a device-resident batch of reads           */

/* ---- library ----------------------------------------------------------- */
WAMD_API const char* wamd_version(void);
WAMD_API const char* wamd_last_error(void);   /* thread-local text of the last failure */
WAMD_API int wamd_device_count(void);

/* ---- context: replaces setReference() on every verifier ---------------- *
 * LevMyers::setReference (LevMyers64.cpp:83-89), CEditDist::SetReference
 * (edit_dist.cpp:98-103), CIndelMatching::SetReference (indels.cpp:116-121).
 * genome: packed reference bytes as CReference::GetData() returns them,
 * genome_bytes of them; ref_size / cur_offset as passed to SetReference.
 * The genome is uploaded once and stays resident in HBM. */
WAMD_API int wamd_ctx_create(int device, const uint8_t* genome, uint64_t genome_bytes,
                             uint32_t ref_size, uint32_t cur_offset, wamd_ctx** out);
/* Same, for a genome already resident on the device (not copied, not owned).  The kernels
 * read it as whole dwords: d_genome must be 4-byte aligned and genome_bytes a multiple of 4
 * (pad with 0xff = two symbols that match nothing), else WAMD_ERR_INVALID_ARG. */
WAMD_API int wamd_ctx_create_dev(int device, const uint8_t* d_genome, uint64_t genome_bytes,
                                 uint32_t ref_size, uint32_t cur_offset, wamd_ctx** out);
WAMD_API int wamd_ctx_destroy(wamd_ctx* ctx);
WAMD_API int wamd_ctx_set_scoring(wamd_ctx* ctx, const wamd_scoring* scoring);
/* LevMyers constructor argument max_text_len (LevMyers.h:45): tasks with
 * max_distance_in_ref above it fail with WAMD_ERR_TEXT_LEN.  Default 16384
 * (mapper/sam.cpp:190). */
WAMD_API int wamd_ctx_set_max_text_len(wamd_ctx* ctx, uint32_t max_text_len);
/* What wamd_verify_distance[_dev] reports for candidates the reference rejects.
 * WAMD_DISTANCE_EXACT (default): edit_distance is what dynamicProgramming leaves in its
 * reference argument, also above k (LevMyers256.hpp:239-244).
 * WAMD_DISTANCE_ACCEPTED_ONLY: accepted candidates (edit_distance <= k) come back exactly as
 * above - distance and end column with the variant's tie-break - and every other one as
 * {k + 1, 0}.  That is all the mapping stages consume (`cur_match <= max_mismatches`,
 * mapper/mapping_core_sa.cpp:237-241, 289-293), and it lets the library compute only the band
 * of diagonals an accepted alignment can use (Hyyro's banded form of the recurrence).
 * wamd_align[_dev] takes its first pass the same way: accepted tasks are aligned exactly as in the default
 * mode, a rejected one reports ok = 0 with {k + 1, 0} (its callers stop at dynamicProgramming's false,
 * sam_matching.cpp:704-705 / 741-744).  wamd_finish_pairs[_dev] shows nothing of a rejected record's distance
 * and always verifies this way, whatever the mode. */
enum { WAMD_DISTANCE_EXACT = 0, WAMD_DISTANCE_ACCEPTED_ONLY = 1 };
WAMD_API int wamd_ctx_set_distance_mode(wamd_ctx* ctx, int mode);
/* What SetReference's ref_size / cur_offset arguments change between stages
 * (every CMappingCore is built with its stage's anchor offset, mapping_core.cpp:156);
 * the genome itself stays where it is. */
WAMD_API int wamd_ctx_set_reference_params(wamd_ctx* ctx, uint32_t ref_size, uint32_t cur_offset);
WAMD_API void* wamd_ctx_stream(wamd_ctx* ctx);   /* the context's own hipStream_t */
WAMD_API int wamd_ctx_synchronize(wamd_ctx* ctx);
/* "*_dev" calls cannot return per-task errors: kernels OR them into a device word
 * (bit 0 text length, 1 window out of the genome, 2 read index, 3 orientation not
 * uploaded, 4 variant/length, 5 malformed task, 6 cigar_stride too small) and mark
 * the task's result with edit_distance = 0xffffffff.  This call waits for `stream`
 * (NULL = the context's), returns the word in *flags and clears it.  The
 * host-pointer entry points do this themselves and turn it into a status. */
WAMD_API int wamd_ctx_take_flags(wamd_ctx* ctx, void* stream, uint32_t* flags);

/* ---- reads: replaces preprocess()/preprocessRawSeq() ------------------- *
 * Uploads n reads and builds, per read and orientation, the pattern-match
 * bit-vectors (LevMyers64.cpp:167-220).  data/offsets: read i starts at byte
 * offsets[i] of data and has lens[i] symbols.  orientations: bit 0 direct,
 * bit 1 rev_comp. */
WAMD_API int wamd_reads_upload(wamd_ctx* ctx, const uint8_t* data, const uint64_t* offsets,
                               const uint32_t* lens, uint32_t n, int encoding,
                               int orientations, wamd_reads** out);
/* The same on a stream of the caller's (NULL = the context's): the upload of the next bin's reads from a second host
 * thread, beside the kernels of the current one.  Returns when the batch is ready (waits for that stream only). */
WAMD_API int wamd_reads_upload_stream(wamd_ctx* ctx, const uint8_t* data, const uint64_t* offsets,
                                      const uint32_t* lens, uint32_t n, int encoding,
                                      int orientations, void* stream, wamd_reads** out);
/* Re-fill an existing batch with the next bin's reads (its encoding and orientations): buffers that are large
 * enough are kept, so a mapper that alternates between two batches allocates nothing after its first bins.  No
 * kernel may still be reading the batch.  On failure the batch is left empty. */
WAMD_API int wamd_reads_refill(wamd_reads* reads, const uint8_t* data, const uint64_t* offsets,
                               const uint32_t* lens, uint32_t n, void* stream);
/* The same without a host pass over the reads, and without waiting: the caller states what its bin already knows -
 * the bytes of `data` and an upper bound on the read lengths (the reference sizes every buffer of a stage by
 * params->max_read_len, mapping_core.cpp:95-108, and a bin arrives as one block) - and the call returns with its
 * copies and kernels enqueued on `stream` (NULL = the context's): data / offsets / lens must stay untouched until that
 * stream has passed them, and verification calls on other streams must wait for it.  The device holds every read
 * against the two statements first; a read of length 0, longer than max_len or reaching past data_bytes raises error
 * bit 5 (wamd_ctx_take_flags; WAMD_ERR_INVALID_ARG from the host-pointer calls) and is replaced by a one-symbol read -
 * such a batch must not be used.  The batch's dword count follows max_len (results do not depend on it).
 * wamd_reads_refill spends 0.5 of its 1.5 ms per 2^19 reads in that host pass and 0.1 in the wait. */
WAMD_API int wamd_reads_refill_async(wamd_reads* reads, const uint8_t* data, uint64_t data_bytes,
                                     const uint64_t* offsets, const uint32_t* lens, uint32_t n,
                                     uint32_t max_len, void* stream);
WAMD_API int wamd_reads_free(wamd_reads* reads);
WAMD_API uint32_t wamd_reads_count(const wamd_reads* reads);

/* ---- verification: edit distance only (mapping stages) ----------------- *
 * Replaces the per-candidate calls at mapping_core_sa.cpp:237/289/399/401/448/450
 * and mapping_core_dp.cpp:137/139/165/167.
 * last_symbol: WAMD_LAST_SYMBOL_FROM_READ after preprocess() (mapping stage), or the
 * value the object's last preprocess() left behind when the read came through
 * preprocessRawSeq() (0 in CSamGenerator) - only the M256 tie-break reads it
 * (LevMyers256.hpp:191-196).  WAMD_LAST_SYMBOL_FROM_READ with reads uploaded as WAMD_READS_ASCII
 * means 0, the value of an object that has only seen preprocessRawSeq() (LevMyers64.cpp:35). */
WAMD_API int wamd_verify_distance(wamd_ctx* ctx, const wamd_reads* reads,
                                  const wamd_dist_task* tasks, uint32_t n, int last_symbol,
                                  wamd_dist_result* results);
WAMD_API int wamd_verify_distance_dev(wamd_ctx* ctx, const wamd_reads* reads,
                                      const wamd_dist_task* d_tasks, uint32_t n,
                                      int last_symbol, wamd_dist_result* d_results,
                                      void* stream);

WAMD_API int wamd_levdiag(wamd_ctx* ctx, const wamd_reads* reads,
                          const wamd_levdiag_task* tasks, uint32_t n, uint32_t* results);
WAMD_API int wamd_levdiag_dev(wamd_ctx* ctx, const wamd_reads* reads,
                              const wamd_levdiag_task* d_tasks, uint32_t n,
                              uint32_t* d_results, void* stream);

/* ---- CF pre-filter (decides which candidates reach the verifier) -------- *
 * results[i] = horizontal_add(abs(CF_read - CF_genome)) of test_CF_128b_SSE_difference
 * (mapping_core.h:260-263) exactly as the reference's unsigned comparison sees it - VCL
 * truncates that sum to int8 and sign-extends it, so 128..255 arrive as huge values and
 * 256+ wrap around; the candidate passes iff results[i] <= 4 * max_mismatches.
 * Reads: nibble-packed, as in the mapping stage. */
WAMD_API int wamd_cf_filter(wamd_ctx* ctx, const wamd_reads* reads, const wamd_cf_task* tasks,
                            uint32_t n, uint32_t* results);
WAMD_API int wamd_cf_filter_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_cf_task* d_tasks,
                                uint32_t n, uint32_t* d_results, void* stream);

/* ---- the filter in front of its verifier: one iteration of the suffix-array loops' body ---- *
 * mapping_core_sa.cpp:223-255 / 279-307 (and :388-460 of the variable-length version):
 *     if (is_short_read || test_CF_128b_SSE_difference(..., 4 * max_mismatches))
 *         cur_match = dynamicProgramming(...) or LevDiag(...);
 *     else
 *         cur_match = max_mismatches + 1;
 * cf_tasks[i] and tasks[i] describe candidate i exactly as wamd_cf_filter and wamd_verify_distance /
 * wamd_levdiag take them; a CF task with pattern_len == 0 is a short read (no filter, always verified).
 * results[i] is the verifier's answer for a candidate the filter passed and {max_ed + 1, 0}
 * (max_mismatches + 1 for LevDiag) for one it turned away.  The verifier only runs on the survivors,
 * packed into full waves on the device; wamd_filter_verify_distance_dev follows the context's distance
 * mode like wamd_verify_distance.  d_workspace: wamd_filter_verify_workspace_bytes(n) bytes (28 n + 1 KB),
 * 256-byte aligned, the caller's; nothing is allocated and nothing synchronises. */
WAMD_API uint64_t wamd_filter_verify_workspace_bytes(uint32_t n);
WAMD_API int wamd_filter_verify_distance_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_cf_task* d_cf_tasks,
                                             const wamd_dist_task* d_tasks, uint32_t n, int last_symbol,
                                             wamd_dist_result* d_results, void* d_workspace, uint64_t workspace_bytes,
                                             void* stream);
WAMD_API int wamd_filter_levdiag_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_cf_task* d_cf_tasks,
                                     const wamd_levdiag_task* d_tasks, uint32_t n, uint32_t* d_results,
                                     void* d_workspace, uint64_t workspace_bytes, void* stream);

/* ---- the suffix-array loops' candidates in their compact form ---------------------------------------- *
 * search_in_unknown_range_dir_and_rc / search_in_known_range_dir_and_rc (mapping_core_sa.cpp:193-348, 355-513) hold
 * everything fixed inside a stage but the read and the suffix-array entry.  A batch is therefore: the stage's
 * constants (wamd_sa_batch), per read of reads[read_first .. read_first + n_reads) how many suffix-array entries
 * it has on the direct and on the other strand (counts[2 i], counts[2 i + 1]), and the 4-byte entries themselves,
 * read after read, a read's direct-strand entries before its other-strand ones - the order the loops visit them.
 * The device writes the tasks wamd_filter_verify_distance_dev / wamd_filter_levdiag_dev take (as :228-239 and
 * :283-291 form them; ref_size is the context's) and runs those kernels.  errors[j] = cur_match as the loop goes on
 * with it: the verifier's errors if <= max_mismatches, else max_mismatches + 1 (also for a candidate the CF
 * pre-filter turns away); end_positions (may be NULL; distance verifier only): dynamicProgramming's end column,
 * 0 for a rejected candidate.  4 + 1 bytes per candidate and 4 per read cross the link.
 * min_read_len == 0: no CF pre-filter (the loops' is_short_read for every read).  The other strand's LevDiag takes
 * the REVERSED read (reverse_read, :291): upload those behind the reads and say where in reverse_read_offset.
 * The distance verifier runs in its accepted-only form whatever the context's mode: nothing of a rejected candidate
 * is reported.  read_max_mismatches (may be NULL): one byte per read replacing batch->max_mismatches for that read's
 * candidates - mapping_mode second verifies with min(stage, best_error + 1) (mapping_core.cpp:955-956); the verifier
 * stays the batch's, so a caller sorts its reads by "k >= lev_alg_thr" into two batches. */
#define WAMD_SA_VERIFIER_DISTANCE 0   /* max_mismatches >= lev_alg_thr: LevMyers::dynamicProgramming */
#define WAMD_SA_VERIFIER_LEVDIAG 1    /* below it: CEditDist::LevDiag                                */
typedef struct wamd_sa_batch {
    uint32_t read_first;
    uint32_t n_reads;
    uint32_t reverse_read_offset;   /* LevDiag: index of read i's reversed copy = i + this          */
    uint16_t max_mismatches;
    uint16_t fixed_left;            /* start_pos / end_pos of the stage's anchor in the direct read */
    uint16_t fixed_right;
    uint16_t min_read_len;
    uint8_t verifier;               /* WAMD_SA_VERIFIER_*                                           */
    uint8_t reserved[3];
} wamd_sa_batch;

WAMD_API uint64_t wamd_sa_candidates_workspace_bytes(uint32_t n_candidates, uint32_t n_reads);
/* counts, sa_positions, errors 4-byte aligned; d_workspace 256-byte aligned, the caller's; nothing is allocated
 * and nothing synchronises.  A batch whose counts do not add up to n_candidates raises flag bit 5. */
WAMD_API int wamd_verify_sa_candidates_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_sa_batch* batch,
                                           const uint16_t* d_counts, const uint8_t* d_read_max_mismatches,
                                           const uint32_t* d_sa_positions,
                                           uint32_t n_candidates, uint8_t* d_errors, uint32_t* d_end_positions,
                                           void* d_workspace, uint64_t workspace_bytes, void* stream);
/* The same from host buffers, as the mapper holds them: cut into chunks of whole reads that alternate between two
 * streams, so that the upload of one chunk, the kernels of the previous one and the download of the one before
 * overlap.  Buffers from wamd_host_alloc (page-locked) make the copies asynchronous; pageable ones work, slower. */
WAMD_API int wamd_verify_sa_candidates(wamd_ctx* ctx, const wamd_reads* reads, const wamd_sa_batch* batch,
                                       const uint16_t* counts, const uint8_t* read_max_mismatches,
                                       const uint32_t* sa_positions, uint32_t n_candidates,
                                       uint8_t* errors, uint32_t* end_positions);
WAMD_API int wamd_host_alloc(void** out, uint64_t bytes);   /* page-locked host memory for the calls above */
WAMD_API int wamd_host_free(void* p);

/* ---- verification with alignment (post-processing) --------------------- *
 * dynamicProgramming + getExtCigar (+ refine_ext_cigar when refine != 0) as in
 * fill_cigar_with_lev (sam_matching.cpp:689-752) and find_pairs<LevMyers>
 * (sam_matching.cpp:401-426).  ext-CIGARs are written NUL-terminated at
 * cigars + i * cigar_stride; cigar_stride is a multiple of 4 and at least
 * 2 * (read length + max_ed) + 4 (the reference's pool slots are 5 * max_read_len,
 * mapper.cpp:697).  Only reads uploaded as WAMD_READS_ASCII take the reference's
 * copy_direct/convert_to_rev_comp symbol mapping (sam.cpp:2296-2347). */
WAMD_API int wamd_align(wamd_ctx* ctx, const wamd_reads* reads, const wamd_dist_task* tasks,
                        uint32_t n, int last_symbol, int refine,
                        wamd_align_result* results, uint8_t* cigars, uint32_t cigar_stride);
/* d_workspace: at least wamd_align_workspace_bytes(...) bytes of device scratch, 256-byte aligned; the
 * call uses all of what it is given.  The minimum (about 270 B per task at 150 bp) serves batches whose
 * accepted distances stay below 15 at full speed and the rare wider ones slowly; a caller whose tasks
 * allow larger distances (the mate rescue with k' = 2 * floor(0.09 m), sam.cpp:190-203; 250 bp reads
 * at high error rates) asks wamd_align_workspace_bytes_for with its largest max_ed and gets room for
 * eight resident waves per CU of the wide traceback classes (0.8 GB at 150 bp, whatever n is). */
WAMD_API int wamd_align_dev(wamd_ctx* ctx, const wamd_reads* reads,
                            const wamd_dist_task* d_tasks, uint32_t n, int last_symbol,
                            int refine, wamd_align_result* d_results, uint8_t* d_cigars,
                            uint32_t cigar_stride, void* d_workspace,
                            uint64_t workspace_bytes, void* stream);
WAMD_API uint64_t wamd_align_workspace_bytes(const wamd_reads* reads, uint32_t n);
WAMD_API uint64_t wamd_align_workspace_bytes_for(const wamd_reads* reads, uint32_t n, uint32_t max_ed);

/* ---- long-indel / clipping matcher (sensitive mapping stages) ----------- *
 * Replaces indel_matching_{dir,rc}->Match(...) at mapping_core_sa.cpp:264/317 and
 * mapping_core_dp.cpp.  Reads: nibble-packed (Preprocess) or ASCII (PreprocessRawSeq). */
WAMD_API int wamd_indel_match(wamd_ctx* ctx, const wamd_reads* reads, const wamd_indel_task* tasks,
                              uint32_t n, const wamd_indel_params* params, wamd_mapping* results);
WAMD_API int wamd_indel_match_dev(wamd_ctx* ctx, const wamd_reads* reads,
                                  const wamd_indel_task* d_tasks, uint32_t n,
                                  const wamd_indel_params* params, wamd_mapping* d_results,
                                  void* stream);

/* Replaces indelMatching->{PreprocessRawSeq, prefetch_genome, GetExtCigarNew} in
 * fill_cigar_with_indel_matching (sam_matching.cpp:824-852). */
WAMD_API int wamd_indel_ext_cigar(wamd_ctx* ctx, const wamd_reads* reads,
                                  const wamd_indel_cigar_task* tasks, uint32_t n,
                                  wamd_indel_cigar_result* results, uint8_t* cigars,
                                  uint32_t cigar_stride);
WAMD_API int wamd_indel_ext_cigar_dev(wamd_ctx* ctx, const wamd_reads* reads,
                                      const wamd_indel_cigar_task* d_tasks, uint32_t n,
                                      wamd_indel_cigar_result* d_results, uint8_t* d_cigars,
                                      uint32_t cigar_stride, void* stream);

/* Replaces softClipping->clipBoundaryPositions(hit, scoring) (sam_matching.cpp:454-457,
 * sam.cpp:1889-1900): ext-CIGARs at cigars + i * cigar_stride and records are
 * rewritten in place. */
WAMD_API int wamd_clip_boundary(wamd_ctx* ctx, uint32_t n, wamd_clip_record* records,
                                uint8_t* cigars, uint32_t cigar_stride);
WAMD_API int wamd_clip_boundary_dev(wamd_ctx* ctx, uint32_t n, wamd_clip_record* d_records,
                                    uint8_t* d_cigars, uint32_t cigar_stride, void* stream);

/* What CSoftClipping::clipIllegalPositions (soft_clipping.cpp:592-729) reads besides the
 * record: the length of the record's chromosome with its N borders
 * (seq_desc: size + no_initial_Ns + no_final_Ns, soft_clipping.cpp:594-595), the read length,
 * and whether the mapping came from a perfect match (method Perfect / PerfectDistant), in
 * which case it has no ext-CIGAR yet and one of read_length '.' is made first (614-618). */
typedef struct wamd_clip_illegal_task {
    int32_t ref_size;
    uint16_t read_length;
    uint16_t perfect;
} wamd_clip_illegal_task;

/* Replaces softClipping->clipIllegalPositions(hit, scoring, seq_desc, pool) (sam.cpp:1339, 1889;
 * sam_matching.cpp:451-452, 599-600, 655): the part of a mapping that hangs over either end of
 * its chromosome becomes '$'; score and err_edit_distance are recomputed from what is left.
 * Records and ext-CIGARs are rewritten in place; no_clippings = sides that hung over (0 = the
 * record was left alone).  The routine's own habits are kept: the tail loop does not advance
 * its input on an ordinary symbol (it pads with '$' up to read_length), and a '$' already in
 * the kept part is scored as a mismatch. */
WAMD_API int wamd_clip_illegal(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* tasks,
                               wamd_clip_record* records, uint8_t* cigars, uint32_t cigar_stride);
WAMD_API int wamd_clip_illegal_dev(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_tasks,
                                   wamd_clip_record* d_records, uint8_t* d_cigars, uint32_t cigar_stride,
                                   void* stream);

/* What CSoftClipping::clipOverlappingPairs (soft_clipping.cpp:997-1044) reads of the two
 * mappings of a pair besides their records; index 0 = p.first, 1 = p.second. */
typedef struct wamd_clip_pair_task {
    int32_t ref_seq_id[2];
    uint16_t read_length[2];
    uint8_t direct[2];             /* mapping.direction == genome_t::direct                 */
    uint8_t perfect[2];            /* method Perfect / PerfectDistant                       */
} wamd_clip_pair_task;

/* Replaces softClipping->clipOverlappingPairs(mp, scoring, pool) (sam.cpp:1898-1900) with its
 * clipLeft / clipRight (soft_clipping.cpp:1047-1151): when the reverse read lies left of the
 * forward read and overlaps it, the left read loses what precedes the right read's start and
 * the right read what follows the left read's end.  Pair i owns records 2i (p.first) and
 * 2i + 1 (p.second) and the ext-CIGAR slots of the same indices; no_clippings of both records
 * is 1 when the pair was clipped.  A record whose ref_length disagrees with its ext-CIGAR
 * (the reference would run off the string) raises the malformed-task flag. */
WAMD_API int wamd_clip_overlapping_pairs(wamd_ctx* ctx, uint32_t n_pairs, const wamd_clip_pair_task* tasks,
                                         wamd_clip_record* records, uint8_t* cigars, uint32_t cigar_stride);
WAMD_API int wamd_clip_overlapping_pairs_dev(wamd_ctx* ctx, uint32_t n_pairs, const wamd_clip_pair_task* d_tasks,
                                             wamd_clip_record* d_records, uint8_t* d_cigars,
                                             uint32_t cigar_stride, void* stream);

/* ---- hashing rescue: CSoftClipping::match + getExtCigar ------------------- *
 * One softClipping->preprocessRawSeq(seq, len, dir) + match(ref_pos, max_distance_in_ref,
 * min_exact_len, ...) (+ getExtCigar) as in find_pairs<Clipping> (sam_matching.cpp:556-575) and
 * fill_cigar_with_clipping (sam_matching.cpp:771-797). */
typedef struct wamd_softclip_task {
    uint32_t read_idx;
    uint32_t ref_pos;              /* raw genome position of the window start (start_pos)   */
    uint32_t max_distance_in_ref;  /* window length searched for 20-mer hits                */
    uint16_t min_exact_len;        /* HashTable::hash_len = 20 at both call sites           */
    uint8_t flags;                 /* WAMD_TASK_FLAG_*                                      */
    uint8_t reserved;
} wamd_softclip_task;

/* Constructor arguments of CSoftClipping (soft_clipping.cpp:28-48, built at sam.cpp:220 from
 * params->max_read_len, min/max_approx_indel_len, max_approx_indel_mismatches).  max_query_len
 * fixes the size of the open-addressing table (smallest power of two >= 256 with
 * size / max_query_len >= 50), hence which 20-mers share a probe chain: results depend on it.
 * fresh_object: the reference's checked_pos_left starts zero-filled and only cells it has touched
 * are reset to "empty" (1 << 30) afterwards, so a hit at query position 1 on a text position no
 * earlier call of the same object has marked is skipped.  1 reproduces a newly built object,
 * 0 one whose cells have all been touched (the long-run state). */
typedef struct wamd_softclip_params {
    uint32_t max_query_len;
    uint32_t min_approx_indel_len;         /* mapper/params.h:134: 3                        */
    uint32_t max_approx_indel_len;         /* mapper/params.h:135: 50                       */
    uint32_t max_approx_indel_mismatches;  /* mapper/params.h:136                           */
    uint32_t fresh_object;
    uint32_t reserved;
} wamd_softclip_params;

/* What match() hands back, and getExtCigar's score.  found = match's bool; when it is 0 every
 * other field is 0 and the ext-CIGAR is empty.  pos is relative to ref_pos and can be
 * 0xffffffff (= -1: the reference subtracts the dummy symbol in front of its window). */
typedef struct wamd_softclip_result {
    double affine_score;
    uint32_t found;
    uint32_t pos;
    uint32_t left_clipping;
    uint32_t right_clipping;
    uint32_t left_match;
    uint32_t right_match;
    int32_t del_size;
    uint32_t no_mismatches;
    uint32_t cigar_len;
    uint32_t reserved;
} wamd_softclip_result;

/* Replaces softClipping->{preprocessRawSeq | preprocess, match, getExtCigar}
 * (soft_clipping.cpp:139-231, 234-495, 498-589).  Window lengths above max_text_len
 * (wamd_ctx_set_max_text_len, 16384 as in sam.cpp:190) and reads shorter than 20 symbols are
 * errors.  ext-CIGARs are written NUL-terminated at cigars + i * cigar_stride
 * (cigar_stride >= read length + 2 * max_approx_indel_len * 2 + 1). */
WAMD_API int wamd_softclip_match(wamd_ctx* ctx, const wamd_reads* reads, const wamd_softclip_task* tasks,
                                 uint32_t n, const wamd_softclip_params* params, wamd_softclip_result* results,
                                 uint8_t* cigars, uint32_t cigar_stride);
/* max_window: an upper bound of max_distance_in_ref over the batch (sizes the kernel's LDS). */
WAMD_API int wamd_softclip_match_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_softclip_task* d_tasks,
                                     uint32_t n, const wamd_softclip_params* params, uint32_t max_window,
                                     wamd_softclip_result* d_results, uint8_t* d_cigars, uint32_t cigar_stride,
                                     void* stream);

/* What CSamGenerator::extract_cigar_and_mdz / extract_cigarbin_and_mdz leave (sam.cpp:805-928,
 * 932-1051): lengths of the three outputs of one ext-CIGAR. */
typedef struct wamd_sam_cigar_len {
    uint32_t cigar_len;            /* bytes of the text CIGAR, NUL not counted (the return value) */
    uint32_t mdz_len;              /* bytes of MD:Z, NUL not counted                              */
    uint32_t cigar_bin_len;        /* BAM operations (count << 4 | op), the other return value    */
    uint32_t reserved;
} wamd_sam_cigar_len;

/* Replaces extract_cigar_and_mdz and extract_cigarbin_and_mdz (called at sam.cpp:1352-1357,
 * 2002-2007, 2174-2179): ext-CIGAR i at ext_cigars + i * ext_stride (NUL-terminated) becomes
 * the SAM CIGAR text at cigars + i * cigar_stride, MD:Z at mdz + i * mdz_stride (both
 * NUL-terminated) and, when cigar_bin is not NULL, BAM operations at cigar_bin + i * bin_stride.
 * A text output can be twice as long as its ext-CIGAR; outputs that do not fit are an error. */
WAMD_API int wamd_sam_cigar(wamd_ctx* ctx, uint32_t n, const uint8_t* ext_cigars, uint32_t ext_stride,
                            uint8_t* cigars, uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride,
                            uint32_t* cigar_bin, uint32_t bin_stride, wamd_sam_cigar_len* lens);
WAMD_API int wamd_sam_cigar_dev(wamd_ctx* ctx, uint32_t n, const uint8_t* d_ext_cigars, uint32_t ext_stride,
                                uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz, uint32_t mdz_stride,
                                uint32_t* d_cigar_bin, uint32_t bin_stride, wamd_sam_cigar_len* d_lens,
                                void* stream);

/* candidate_mapping_t::serialize (results.cpp:161-217): the 5..11-byte wire records
 * that are gathered across GPUs.  sizes[i] bytes are written at out + i * 12;
 * direction[i] != 0 means genome_t::direct. */
WAMD_API int wamd_serialize_mappings(wamd_ctx* ctx, uint32_t n, const wamd_mapping* mappings,
                                     const uint8_t* direction, uint8_t* out, uint8_t* sizes);
WAMD_API int wamd_serialize_mappings_dev(wamd_ctx* ctx, uint32_t n, const wamd_mapping* d_mappings,
                                         const uint8_t* d_direction, uint8_t* d_out,
                                         uint8_t* d_sizes, void* stream);

/* Mapping records of the accepted candidates of a distance batch, in task order: what
 * the mapping loop pushes to the results collector (mapping_core_sa.cpp:241-243, 293-298,
 * mapping_core.cpp:980-997) and what is gathered across GPUs.  A candidate is accepted when
 * edit_distance <= its task's max_ed.  Each record is 12 bytes:
 *   [0..3] read_idx (little-endian)   [4..8] candidate_mapping_t::serialize of a lev mapping
 *   (4 B big-endian position, then 0x80 if direct + 0x40 + errors & 0x3f)   [9..11] zero.
 * record_pos[i] / direction[i]: the position and strand the host loop would record for
 * task i.  *d_count receives the number of records; d_scratch needs
 * wamd_records_scratch_bytes(n) bytes. */
WAMD_API uint64_t wamd_records_scratch_bytes(uint32_t n);
WAMD_API int wamd_emit_lev_records_dev(wamd_ctx* ctx, uint32_t n, const wamd_dist_task* d_tasks,
                                       const wamd_dist_result* d_results, const uint32_t* d_record_pos,
                                       const uint8_t* d_direction, uint8_t* d_records, uint32_t* d_count,
                                       void* d_scratch, void* stream);

/* ---- finishing the chosen pair: ext-CIGAR, clipping, SAM CIGAR / MD:Z in one call -------- *
 * What store_mapped_pair_reads does to the two mappings of the pair it has picked
 * (mapper/sam.cpp:1871-1900, then 1992-2007), for a batch of pairs, without leaving the device:
 *   fill_cigar                  sam_matching.cpp:667-686   perfect match: no alignment at all
 *     fill_cigar_with_lev       sam_matching.cpp:689-752   preprocessRawSeq + dynamicProgramming +
 *                                                          getExtCigar + refine_ext_cigar
 *     fill_cigar_with_indel_matching  sam_matching.cpp:824-852   GetExtCigarNew
 *   clipIllegalPositions        sam.cpp:1889               soft_clipping.cpp:592-729
 *   clipBoundaryPositions       sam.cpp:1890-1892          soft_clipping.cpp:827-994
 *   clipOverlappingPairs        sam.cpp:1898-1900          soft_clipping.cpp:997-1151
 *   extract_cigar_and_mdz       sam.cpp:2002-2007          sam.cpp:805-928 (perfect matches keep
 *                                                          "<len>M" / "<len>", sam.cpp:1504-1515)
 * Read slot 2i is p.first of pair i, slot 2i + 1 p.second.  A task is the mapping_desc_t the
 * generator built from the mapping record (sam.cpp:1555-1580): the deserialised
 * candidate_mapping_t, Translate()'s chromosome and position, the constructor's
 * err_edit_distance (common/types.h:390-425); `distant` is fill_cigar's argument (the insert
 * size against the model, sam.cpp:1879-1883: host logic).  For the non-lev types the host passes
 * the chromosome (ref_seq_id, ref_seq_pos, ref_size) of Translate(examined_start_pos),
 * examined_start_pos = direct ? pos : pos - (length - 1): what fill_cigar_with_indel_matching
 * stores (sam_matching.cpp:826-828, 843-849) - the constructor's values unless the window start
 * lies across a chromosome border. */
typedef struct wamd_finish_task {
    uint32_t read_idx;             /* ASCII read of the batch                               */
    uint32_t pos;                  /* candidate_mapping_t::pos                              */
    uint32_t type;                 /* WAMD_MAP_*                                            */
    uint32_t penalty;              /* candidate_mapping_t::penalty as the generator holds it */
    uint32_t matching_len1;
    uint32_t matching_len2;
    int32_t indel_len1;
    int32_t indel_len2;
    uint32_t clipping_len;
    int32_t ref_seq_id;
    int32_t ref_seq_pos;
    int32_t ref_size;              /* the chromosome's size + no_initial_Ns + no_final_Ns   */
    uint16_t err_edit_distance;
    uint8_t direct;                /* mapping.direction == genome_t::direct                 */
    uint8_t distant;               /* 0 / 1: fill_cigar's argument as the host computed it; WAMD_DISTANT_FROM_PAIR:  */
                                   /* computed on the device as sam.cpp:1881-1884 does, see below                    */
    int32_t isize_min;             /* WAMD_DISTANT_FROM_PAIR only: ceil(mean - getMyersDev) and floor(mean +          */
    int32_t isize_max;             /* getMyersDev) of the insert-size model, the same in both slots of a pair        */
    uint32_t flags;                /* WAMD_FINISH_*                                                                  */
} wamd_finish_task;

/* distant = WAMD_DISTANT_FROM_PAIR: isDistant of sam.cpp:1881-1884, `isize > mean + MyersDev || isize < mean -
 * MyersDev` with isize = mapping_pair_t::calculateInsertSize (common/types.h:482-515) of the pair AS IT STANDS WHEN
 * fill_cigar IS CALLED: for slot 2i both mappings as constructed (ref_length = read length, sam.cpp:1581), for
 * slot 2i + 1 the first mapping after its own fill_cigar, clipIllegalPositions and clipBoundaryPositions have moved
 * its position and length - which no host can know before the call.  isize is an int, so the two comparisons are
 * exact against the integer bounds isize_min / isize_max.  (For the non-lev types the pair's chromosome and position
 * are the task's, i.e. Translate(examined_start_pos): the constructor's unless the window start lies across a
 * chromosome border.) */
#define WAMD_DISTANT_FROM_PAIR 2u
/* The host's Translate(examined_start_pos) failed for this non-lev record (a window start inside the N border in front
 * of the first chromosome): fill_cigar_with_indel_matching then leaves mapping.pos and the chromosome alone
 * (sam_matching.cpp:843-849); ref_seq_id / ref_seq_pos / ref_size are the constructor's. */
#define WAMD_FINISH_NO_TRANSLATE 1u

/* CParams fields the chain reads. */
typedef struct wamd_finish_params {
    uint32_t flank;                          /* sam_matching.cpp:696: uint32(sensitive_mode ? max_no_errors *  */
                                             /* sensitivity_factor : max_no_errors) + 1                        */
    uint32_t enable_short_indel_refinement;  /* mapper/params.h: 1                                             */
    uint32_t enable_boundary_clipping;       /* 1                                                              */
    uint32_t enable_paired_clipping;         /* 1                                                              */
} wamd_finish_params;

/* MatchingMethod::Enumeration (common/types.h:168-175) values this path sets. */
enum {
    WAMD_METHOD_PERFECT = 0, WAMD_METHOD_PERFECT_DISTANT = 1, WAMD_METHOD_FROM_MAPPING = 2,
    WAMD_METHOD_FROM_MAPPING_DISTANT = 3, WAMD_METHOD_UNMATCHED = 26
};

/* The fields of mapping_desc_t the SAM record is written from (store_mapping_result,
 * sam.cpp:2013-2017).  ok = 0: fill_cigar could not align the record.  The reference counts it in
 * stat_mapped_pe_errors and goes on with a null ext-CIGAR, which clipBoundaryPositions (err_edit_distance != 0)
 * and extract_cigar_and_mdz then dereference: it does not survive such a record (the mapping stage's error count
 * rules it out).  Here the slot reports ok = 0 with empty outputs and stays out of the pair clipping; its mate is
 * finished as usual. */
typedef struct wamd_finish_result {
    double score;                  /* AS:i: is Double2PChar of it                            */
    uint32_t pos;                  /* mapping.pos after fill_cigar                           */
    int32_t ref_seq_pos;
    uint16_t ref_length;
    uint16_t err_edit_distance;    /* NM:i:                                                  */
    uint16_t num_events;
    uint16_t left_clipping;
    uint16_t right_clipping;
    uint16_t method;               /* XM:Z:                                                  */
    uint32_t ext_cigar_len;
    uint32_t cigar_len;
    uint32_t mdz_len;
    uint32_t ok;
    uint32_t reserved;
} wamd_finish_result;

/* ext-CIGARs at ext_cigars + slot * ext_stride (a multiple of 4, >= 2 * (read length + flank + 1) + 8),
 * SAM CIGAR text at cigars + slot * cigar_stride, MD:Z at mdz + slot * mdz_stride, all NUL-terminated. */
WAMD_API uint64_t wamd_finish_workspace_bytes(const wamd_reads* reads, uint32_t n_pairs);
WAMD_API int wamd_finish_pairs(wamd_ctx* ctx, const wamd_reads* reads, const wamd_finish_task* tasks,
                               uint32_t n_pairs, const wamd_finish_params* params, wamd_finish_result* results,
                               uint8_t* ext_cigars, uint32_t ext_stride, uint8_t* cigars, uint32_t cigar_stride,
                               uint8_t* mdz, uint32_t mdz_stride);
WAMD_API int wamd_finish_pairs_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_finish_task* d_tasks,
                                   uint32_t n_pairs, const wamd_finish_params* params,
                                   wamd_finish_result* d_results, uint8_t* d_ext_cigars, uint32_t ext_stride,
                                   uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz, uint32_t mdz_stride,
                                   void* d_workspace, uint64_t workspace_bytes, void* stream);

/* ---- mate rescue by LevMyers: find_pairs<LevMyers>'s chain per hit (sam_matching.cpp:379-457) -------------------- *
 * For a hit of one read the mate is aligned in the window the insert-size model allows beside it:
 *     start_pos = paired_dir == direct ? hit_pos + len(mate) - distance_threshold : hit_pos + 1 - max_mate_edit_distance
 *     dynamicProgramming(start_pos, distance_threshold + 2 * max_mate_edit_distance, max_mate_edit_distance)   :392-400
 *     getExtCigar (+ refine_ext_cigar), paired_raw_pos = start_pos + relative_begin_pos                         :414-422
 *     Translate(paired_raw_pos) must give the hit's chromosome; num_events <= max_mate_edit_distance / 2 for the
 *     mate's alignment and for the hit                                                                          :424-426
 *     the mate's mapping_desc_t (lev: err_edit_distance = the edit distance, ref_length = end - begin,
 *     score = the affine score), clipIllegalPositions, clipBoundaryPositions                                    :433-450
 * all of it on the device, one call per batch of hits: results[i].ok says whether the pair candidate exists, and then
 * holds the mate's mapping as the pair evaluator sees it; its ext-CIGAR is in ext_cigars + i * ext_stride.  The hit's
 * own two clipping passes (:444, :448) are wamd_clip_illegal / wamd_clip_boundary on the caller's records.
 * What Translate needs of the hit's chromosome travels in the task (ref_desc.cpp:81-118): raw positions p with
 * seq_lo < p + 1 <= seq_hi belong to it, ref_seq_pos = p + seq_delta.  Reads: ASCII, both orientations. */
typedef struct wamd_rescue_task {
    uint32_t mate_read_idx;
    uint32_t hit_pos;                 /* this_hit.mapping.pos                                             */
    uint32_t distance_threshold;      /* (ref_pos_t) (mean + MyersDev), :347                              */
    uint32_t seq_lo;                  /* pos_in_ref_seq + SEPARATE_N_LEN / 2 of the hit's chromosome      */
    uint32_t seq_hi;                  /* the same of the next one; the last: pos_in_ref_seq + size + 2 * SEPARATE_N_LEN */
    int32_t seq_delta;                /* no_initial_Ns - no_starting_Ns + 1 - pos_in_ref_seq              */
    int32_t ref_size;                 /* size + no_final_Ns + no_initial_Ns (soft_clipping.cpp:594-595)   */
    uint16_t max_mate_edit_distance;  /* (uint32_t) (len(mate) * max_mate_edit_distance_frac) * 2, :388   */
    uint16_t hit_num_events;
    uint8_t paired_dir_direct;        /* get_direction_using_orientation(...) == genome_t::direct         */
    uint8_t reserved[3];
} wamd_rescue_task;

typedef struct wamd_rescue_result {
    double score;                     /* affine, after the clipping passes                                */
    uint32_t ok;                      /* the pair candidate exists                                        */
    uint16_t aligned;                 /* dynamicProgramming returned true (:402)                          */
    uint16_t edit_distance;           /* its edit distance, + 1 when refine_ext_cigar rewrote (:425): the mate's    *
                                       * mapping.no_mismatches and mapping_desc_t::num_events (the clipping passes  *
                                       * change err_edit_distance only)                                            */
    uint32_t pos;                     /* paired_raw_pos                                                   */
    int32_t ref_seq_pos;
    uint16_t ref_length;
    uint16_t err_edit_distance;       /* after the clipping passes                                        */
    uint16_t left_clipping;
    uint16_t right_clipping;
    uint32_t ext_cigar_len;
    uint32_t cigar_events;            /* num_events of getExtCigar / refine_ext_cigar (:424)              */
} wamd_rescue_result;

typedef struct wamd_rescue_params {
    uint32_t enable_short_indel_refinement;
    uint32_t enable_boundary_clipping;
} wamd_rescue_params;

WAMD_API uint64_t wamd_rescue_workspace_bytes(const wamd_reads* reads, uint32_t n, uint32_t max_mate_edit_distance);
WAMD_API int wamd_rescue_mates(wamd_ctx* ctx, const wamd_reads* reads, const wamd_rescue_task* tasks, uint32_t n,
                               const wamd_rescue_params* params, wamd_rescue_result* results, uint8_t* ext_cigars,
                               uint32_t ext_stride);
/* ext_stride: a multiple of 4, >= 2 * (read length + max_mate_edit_distance) + 8; d_workspace 256-byte aligned */
WAMD_API int wamd_rescue_mates_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_rescue_task* d_tasks, uint32_t n,
                                   const wamd_rescue_params* params, wamd_rescue_result* d_results, uint8_t* d_ext_cigars,
                                   uint32_t ext_stride, void* d_workspace, uint64_t workspace_bytes, void* stream);

/* ---- from verified candidates to the bytes of a result group ---------------------------- *
 * One suffix-array candidate after its verifiers have answered, as the loop of
 * search_in_unknown_range_dir_and_rc holds it at mapping_core_sa.cpp:241-344 (the known-range and
 * malicious-group loops are the same): cur_match of dynamicProgramming / LevDiag, the position and
 * strand a lev hit is recorded with (:243 sa - CUR_OFFSET, :298 ref_size - (sa - CUR_OFFSET) - 1),
 * and whether CIndelMatching::Match ran for it (sensitive mode, :258-272 / :310-326). */
typedef struct wamd_commit_task {
    uint32_t lev_pos;
    uint32_t lev_errors;           /* > max_mismatches: no lev mapping                       */
    uint16_t max_mismatches;
    uint16_t pattern_len;          /* rc strand: Match's pos += pattern_len - 1 (:320)        */
    uint8_t direct;
    uint8_t has_indel;
    uint16_t reserved;
} wamd_commit_task;

/* candidate_mapping of every task: the lev mapping if cur_match <= max_mismatches, Match's
 * mapping (d_indel[i], as wamd_indel_match left it; may be NULL when no task has one) with its
 * strand fix-up, candidate_mapping_t::better of the two (results.cpp:376-387), kept only when its
 * penalty <= max_mismatches (:338); otherwise found = 0 / type = WAMD_MAP_NONE. */
WAMD_API int wamd_commit_candidates_dev(wamd_ctx* ctx, uint32_t n, const wamd_commit_task* d_tasks,
                                        const wamd_mapping* d_indel, wamd_mapping* d_committed, void* stream);

/* CParams fields of CMappingResultsCollector / CMappingsHeapGatherer (results.cpp:484-503, 393-405). */
typedef struct wamd_group_params {
    uint32_t id_bytes;             /* BITS2BYTES(id_bits_total)                               */
    uint32_t mapping_counter_size; /* mapper/params.h:128: 4                                  */
    uint32_t max_no_mappings;      /* heap size: the best of at most this many per read       */
    uint32_t reserved;
} wamd_group_params;

/* The bytes CMappingResultsCollector::Push (results.cpp:546-598) leaves in a group buffer for the
 * reads of one bin, in bin order: per read with at least one kept mapping its id (id_bytes,
 * big-endian), the mapping count (mapping_counter_size bytes) and the serialized records
 * (results.cpp:161-217) in the order the heap gatherer hands them over (mapping_core.cpp:976-997:
 * pushed dir candidates first, then rc; popped from the back of the heap's array - the replay of
 * std::push_heap / pop_heap keeps that order byte for byte).  Tasks of read r are
 * d_read_first[r] .. d_read_first[r + 1] - 1 (n_reads + 1 entries), in the loop's order.
 * *d_out_bytes receives the group's size; d_read_off (may be NULL) the offset of every read's
 * block; a group beyond out_capacity raises flag bit 6.  d_workspace:
 * wamd_result_groups_workspace_bytes(n_tasks, n_reads) bytes. */
WAMD_API uint64_t wamd_result_groups_workspace_bytes(uint32_t n_tasks, uint32_t n_reads);
WAMD_API int wamd_emit_result_groups_dev(wamd_ctx* ctx, uint32_t n_tasks, const wamd_commit_task* d_tasks,
                                         const wamd_mapping* d_committed, uint32_t n_reads,
                                         const uint32_t* d_read_first, const uint64_t* d_read_ids,
                                         const wamd_group_params* params, uint8_t* d_out, uint32_t out_capacity,
                                         uint32_t* d_out_bytes, uint32_t* d_read_off, void* d_workspace,
                                         void* stream);

/* ---- gathering the mapping records of all GPUs (the path's only exchange step) -------- *
 * No counterpart in the reference, whose worker threads push records into one process-wide
 * CMappingResultsCollector (mapper/results.h:55-159; mapping_core.cpp:980-997).  With the read
 * bins of a stage sharded over the GPUs of a node (the per-stage workers of mapper.cpp:1847-1862,
 * one group per GPU), every GPU ends a bin with a block of records in its own HBM; this call
 * brings the blocks to the GPU of `root` - the rank that writes result groups / SAM - over
 * RCCL point-to-point sends on xGMI, packed in rank order.
 *   one process per GPU:      rank 0 calls wamd_comm_get_unique_id and hands the 128 bytes to the
 *                             other ranks by whatever channel the host has; every rank calls
 *                             wamd_comm_create, then wamd_gather_records_dev per bin;
 *   one process, all GPUs:    wamd_comm_create_all over the node's contexts, then
 *                             wamd_gather_records_all (every rank's calls inside one RCCL group,
 *                             so one host thread cannot deadlock itself).
 * d_records: this rank's records (record_bytes each: 12 for wamd_emit_lev_records_dev), *d_count of
 * them (a device word, as the emit call leaves it); capacity: the most any rank can hold (sizes
 * d_out on the root: n_ranks * capacity records at most are written, contiguously);
 * counts_out: n_ranks HOST words, every rank's count, on every rank.  The call waits for the
 * counts (one small read-back per bin), then enqueues the transfers on `stream` and returns.
 * librccl is opened on first use; without it these calls fail with WAMD_ERR_NO_DEVICE. */
typedef struct wamd_comm wamd_comm;
#define WAMD_COMM_ID_BYTES 128
WAMD_API int wamd_comm_get_unique_id(uint8_t* id /* WAMD_COMM_ID_BYTES */);
WAMD_API int wamd_comm_create(wamd_ctx* ctx, int rank, int n_ranks, const uint8_t* id, wamd_comm** out);
WAMD_API int wamd_comm_create_all(wamd_ctx* const* ctxs, int n, wamd_comm** out /* n handles */);
WAMD_API int wamd_comm_destroy(wamd_comm* comm);
WAMD_API int wamd_gather_records_dev(wamd_comm* comm, const uint8_t* d_records, const uint32_t* d_count,
                                     uint32_t record_bytes, uint32_t capacity, int root, uint8_t* d_out,
                                     uint32_t* counts_out, void* stream);
WAMD_API int wamd_gather_records_all(wamd_comm* const* comms, int n, const uint8_t* const* d_records,
                                     const uint32_t* const* d_count, uint32_t record_bytes, uint32_t capacity,
                                     int root, uint8_t* d_out, uint32_t* counts_out, void* const* streams);

/* The same emission for callers that cannot allocate device memory or bring a scratch
 * buffer: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * No counterpart in the reference.  The block counts of the record scan live in a static
 * device array of the library (256.25 KiB per GPU, held until the process ends) instead of
 * d_scratch; the call allocates nothing and is asynchronous on `stream`.  Output is byte for
 * byte that of wamd_emit_lev_records_dev.  n above WAMD_STATIC_RECORDS_MAX_TASKS fails with
 * WAMD_ERR_CAPACITY.  All calls on one GPU share the array: calls that overlap in time on one
 * GPU (two streams, two contexts of one device) race - serialise them. */
#define WAMD_STATIC_RECORDS_MAX_TASKS (1u << 24)
WAMD_API int wamd_emit_lev_records_static_dev(wamd_ctx* ctx, uint32_t n, const wamd_dist_task* d_tasks,
                                              const wamd_dist_result* d_results, const uint32_t* d_record_pos,
                                              const uint8_t* d_direction, uint8_t* d_records, uint32_t* d_count,
                                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_AMD_H */
