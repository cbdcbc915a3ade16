// wamd_internal.h - shared definitions of the MI355X read-verification library.
// Host objects behind the opaque C-ABI handles, device-side views and the small
// device helpers every kernel uses.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/whisper_amd.h"

namespace wamd {

// ---- device error flags (OR-ed by kernels into ctx->d_flags) ---------------------
enum : uint32_t {
    FLAG_TEXT_LEN = 1u << 0,   // W == 0 or W > max_text_len (reference throws)
    FLAG_RANGE = 1u << 1,      // window leaves the genome
    FLAG_READ_IDX = 1u << 2,   // read_idx >= reads in batch
    FLAG_ORIENT = 1u << 3,     // orientation not prepared by wamd_reads_upload
    FLAG_VARIANT = 1u << 4,    // variant cannot hold the read (M128 needs len <= 128 ...)
    FLAG_ARG = 1u << 5,        // malformed task (anchor outside read, k too large ...)
    FLAG_CIGAR = 1u << 6,      // cigar_stride too small for a task
};
// test hooks (wamd_ctx::test_hooks): code paths ordinary inputs seldom reach
enum : uint32_t {
    HOOK_POST_NO_HINTS = 1u << 0,   // post passes: ignore the callers' length hints, every wave in two halves
    HOOK_LEVDIAG_RUNS = 1u << 1,    // LevDiag: the run-extension kernel for every read length
};
constexpr uint32_t RESULT_ERROR = 0xFFFFFFFFu;
constexpr int BAND_SLOTS = 16;

// ---- device view of a reads batch ------------------------------------------------
// codes : canonical 4-bit codes of every read in DIRECT orientation, two per byte,
//         even position in the high nibble (the reference's own packing), read i at
//         byte code_off[i] (4-byte aligned).  ASCII input is converted with raw_code
//         (LevMyers64.cpp:50-60): ACGT 0..3, N 4, anything else 6.
// pm    : pattern-match bit-vectors, rows for text symbols 0..3 only (a text symbol
//         >= 4 can only match a read holding that same code, which `presence` tells):
//         pm[((read * n_orient + slot) * 4 + c) * words + w]
// pm32  : the same vectors as 32-bit dwords for the distance kernel, TOP-aligned: symbol
//         len-1 of the read sits in bit 31 of dword words32-1, the unused low bits
//         ("phantom rows" before the read's first symbol) are 1 in every row:
//         pm32[((read * n_orient + slot) * words32 + j) * 4 + c]  (dword-major: the four rows of
//         one dword are 16 contiguous bytes)
// meta  : per (read, slot): bits 0..15 presence of each code in the read,
//         bits 16..19 last symbol in that orientation (LevMyers64.cpp:182/192)
struct ReadsView {
    const uint8_t* codes;
    const uint64_t* code_off;
    const uint32_t* lens;
    const uint64_t* pm;
    const uint32_t* pm32;
    const uint32_t* meta;
    uint32_t n;
    uint32_t words;        // 64-bit words per bit-vector in this batch (max over reads)
    uint32_t words32;      // dwords per top-aligned bit-vector (ceil(max_len / 32))
    uint32_t pm32_stride;  // 16-byte units from one (read, slot)'s pm32 rows to the next: words32 rounded up so
                           // that a row set never straddles a 128-byte line (4, 8, then multiples of 8)
    uint32_t n_orient;     // 1 or 2 slots per read
    int32_t slot_direct;   // slot index of the direct orientation, -1 if absent
    int32_t slot_rc;
};

struct GenomeView {
    const uint32_t* dwords;   // packed genome viewed as little-endian dwords
    uint64_t n_symbols;       // 2 * genome_bytes
    uint32_t last_dword;      // index of the last dword that may be loaded
};

} // namespace wamd

struct wamd_ctx {
    int device;
    hipStream_t stream;
    uint8_t* d_genome;          // padded copy (owned) or caller's pointer
    bool owns_genome;
    uint64_t genome_bytes;
    uint32_t ref_size;
    uint32_t cur_offset;
    uint32_t max_text_len;
    int distance_mode;          // WAMD_DISTANCE_*
    uint32_t test_hooks;        // wamd::HOOK_* (wamd_internal_set_test_hooks; 0 outside the tests)
    wamd_scoring scoring;
    uint32_t* d_flags;          // device error flags; behind them BAND_SLOTS pairs of "something was
                                // deferred" words of the banded distance chain, one pair per stream
    void* band_stream[16];      // which stream owns which pair (assigned at its first launch)
    int band_streams;
    int band_lock;
    uint32_t* h_flags;          // pinned mirror
    // staging for the host-pointer entry points (grown on demand)
    void* d_stage;
    uint64_t d_stage_bytes;
    void* h_stage;
    uint64_t h_stage_bytes;
    hipStream_t pipe[4];        // wamd_verify_sa_candidates: chunks take turns on these (created at first use)
    // Host-pointer entry points share d_stage, h_flags and `stream`: each of them holds this lock
    // from its first use of the staging buffer to its last read of the results, so worker threads
    // that flush on one shared context are serialised, never interleaved.
    std::mutex host_mutex;
};

struct wamd_reads {
    wamd_ctx* ctx;
    uint32_t n;
    int encoding;
    int orientations;
    uint32_t max_len;
    uint32_t words;
    uint8_t* d_codes;
    uint64_t* d_code_off;
    uint32_t* d_lens;
    uint64_t* d_pm;
    uint32_t* d_pm32;
    uint32_t words32;
    uint32_t pm32_stride;
    uint32_t* d_meta;
    uint8_t* d_raw;             // the reads as uploaded (read by the prepare kernels only; kept until the batch is freed)
    uint64_t* d_raw_off;
    uint64_t cap[8];            // bytes behind d_codes, d_code_off, d_lens, d_pm, d_pm32, d_meta, d_raw, d_raw_off
    uint64_t codes_bytes;
    // the 64-bit pattern vectors are only read by the traceback's wide classes: built by the first
    // wamd_align call on the batch (ensure_pm), on its stream; other streams wait for pm_event
    int pm_state, pm_lock;
    hipStream_t pm_stream;
    hipEvent_t pm_event;
    wamd::ReadsView view() const;
};

extern "C" WAMD_API int wamd_internal_set_test_hooks(wamd_ctx* ctx, uint32_t hooks);

namespace wamd {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
GenomeView genome_view(const wamd_ctx* ctx);
int ensure_stage(wamd_ctx* ctx, uint64_t dev_bytes, uint64_t host_bytes);
int flags_to_status(uint32_t flags);

#define WAMD_HIP(call)                                                           \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess)                                                    \
            return ::wamd::hip_fail(e_, #call, __FILE__, __LINE__);              \
    } while (0)

// kernel launchers (each defined next to its kernel)
int launch_reads_prepare(wamd_ctx* ctx, wamd_reads* r, const uint8_t* d_raw,
                         const uint64_t* d_raw_off, hipStream_t s, uint64_t stated_bytes = 0);
int ensure_pm(const wamd_reads* r, hipStream_t s);
int launch_distance(wamd_ctx* ctx, const wamd_reads* r, const wamd_dist_task* d_tasks, uint32_t n,
                    int last_symbol, wamd_dist_result* d_results, int mode, hipStream_t s,
                    const uint32_t* n_dev = nullptr);   // n_dev: device word holding the batch's real length (<= n)
int launch_levdiag(wamd_ctx* ctx, const wamd_reads* r, const wamd_levdiag_task* d_tasks,
                   uint32_t n, uint32_t* d_results, hipStream_t s, const uint32_t* n_dev = nullptr);
int launch_align(wamd_ctx* ctx, const wamd_reads* r, const wamd_dist_task* d_tasks, uint32_t n,
                 int last_symbol, int refine, wamd_align_result* d_results, uint8_t* d_cigars,
                 uint32_t cigar_stride, void* d_ws, uint64_t ws_bytes, hipStream_t s,
                 int distance_mode = -1);   // -1: the context's; finish_pairs never reads a rejected task's distance
uint64_t align_workspace_bytes(const wamd_reads* r, uint32_t n, uint32_t max_ed = 0);
int launch_indel_match(wamd_ctx* ctx, const wamd_reads* r, const wamd_indel_task* d_tasks, uint32_t n,
                       const wamd_indel_params* prm, wamd_mapping* d_results, hipStream_t s);
int launch_indel_ext_cigar(wamd_ctx* ctx, const wamd_reads* r, const wamd_indel_cigar_task* d_tasks,
                           uint32_t n, wamd_indel_cigar_result* d_results, uint8_t* d_cigars,
                           uint32_t cigar_stride, hipStream_t s);
int launch_clip_boundary(wamd_ctx* ctx, uint32_t n, wamd_clip_record* d_records, uint8_t* d_cigars,
                         uint32_t cigar_stride, hipStream_t s);
int launch_clip_illegal(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_tasks, wamd_clip_record* d_records,
                        uint8_t* d_cigars, uint32_t cigar_stride, hipStream_t s);
int launch_clip_pairs(wamd_ctx* ctx, uint32_t n_pairs, const wamd_clip_pair_task* d_tasks, wamd_clip_record* d_records,
                      uint8_t* d_cigars, uint32_t cigar_stride, hipStream_t s);
int launch_serialize(wamd_ctx* ctx, uint32_t n, const wamd_mapping* d_mappings, const uint8_t* d_direction,
                     uint8_t* d_out, uint8_t* d_sizes, hipStream_t s);

int launch_post_passes(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_it, const wamd_clip_pair_task* d_pt,
                       wamd_clip_record* d_rec, uint8_t* d_ext, uint32_t ext_stride, const uint32_t* d_cap_hint,
                       int do_boundary, int do_pairs, uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz,
                       uint32_t mdz_stride, wamd_sam_cigar_len* d_lens, int2* d_mid, hipStream_t s);

int launch_clip_records(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_it, wamd_clip_record* d_rec, uint8_t* d_ext,
                        uint32_t ext_stride, const uint32_t* d_cap_hint, int do_boundary, hipStream_t s);
uint64_t rescue_workspace_bytes(const wamd_reads* r, uint32_t n, uint32_t max_ed);
int launch_rescue_mates(wamd_ctx* ctx, const wamd_reads* r, const wamd_rescue_task* d_tasks, uint32_t n,
                        const wamd_rescue_params* prm, wamd_rescue_result* d_results, uint8_t* d_ext, uint32_t ext_stride,
                        void* d_ws, uint64_t ws_bytes, hipStream_t s);

int launch_cf_filter(wamd_ctx* ctx, const wamd_reads* r, const wamd_cf_task* d_tasks, uint32_t n,
                     uint32_t* d_results, hipStream_t s);

uint64_t filter_verify_workspace_bytes(uint32_t n);
int launch_filter_distance(wamd_ctx* ctx, const wamd_reads* r, const wamd_cf_task* d_cf, const wamd_dist_task* d_tasks,
                           uint32_t n, int last_symbol, wamd_dist_result* d_results, void* d_ws, hipStream_t s,
                           int distance_mode = -1);   // -1: the context's
uint64_t sa_candidates_workspace_bytes(uint32_t n, uint32_t n_reads);
int launch_sa_candidates(wamd_ctx* ctx, const wamd_reads* r, const wamd_sa_batch* b, const uint16_t* d_counts,
                         const uint8_t* d_read_k, const uint32_t* d_sa_positions, uint32_t n, uint8_t* d_errors, uint32_t* d_end_positions,
                         void* d_ws, hipStream_t s);
int launch_filter_levdiag(wamd_ctx* ctx, const wamd_reads* r, const wamd_cf_task* d_cf, const wamd_levdiag_task* d_tasks,
                          uint32_t n, uint32_t* d_results, void* d_ws, hipStream_t s);

int launch_sam_cigar(wamd_ctx* ctx, uint32_t n, const uint8_t* d_ext, uint32_t ext_stride, uint8_t* d_cigars,
                     uint32_t cigar_stride, uint8_t* d_mdz, uint32_t mdz_stride, uint32_t* d_bin, uint32_t bin_stride,
                     wamd_sam_cigar_len* d_lens, hipStream_t s);

// a stream's pair of "something was left to the next kernel of the chain" words (myers_distance.hip); nullptr when
// more streams than there are pairs have asked
uint32_t* band_flags(wamd_ctx* ctx, hipStream_t s);
int launch_softclip_match(wamd_ctx* ctx, const wamd_reads* r, const wamd_softclip_task* d_tasks, uint32_t n,
                          const wamd_softclip_params* prm, uint32_t max_window, wamd_softclip_result* d_results,
                          uint8_t* d_cigars, uint32_t cigar_stride, hipStream_t s);

uint64_t records_scratch_bytes(uint32_t n);
int launch_emit_lev_records(wamd_ctx* ctx, uint32_t n, const wamd_dist_task* d_tasks,
                            const wamd_dist_result* d_results, const uint32_t* d_record_pos,
                            const uint8_t* d_direction, uint8_t* d_records, uint32_t* d_count, void* d_scratch,
                            hipStream_t s);

uint64_t finish_workspace_bytes(const wamd_reads* r, uint32_t n_pairs);
int launch_finish_pairs(wamd_ctx* ctx, const wamd_reads* r, const wamd_finish_task* d_tasks, uint32_t n_pairs,
                        const wamd_finish_params* prm, wamd_finish_result* d_results, uint8_t* d_ext, uint32_t ext_stride,
                        uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz, uint32_t mdz_stride, void* d_ws,
                        uint64_t ws_bytes, hipStream_t s);

uint64_t result_groups_workspace_bytes(uint32_t n_tasks, uint32_t n_reads);
int launch_commit(wamd_ctx* ctx, uint32_t n, const wamd_commit_task* d_tasks, const wamd_mapping* d_indel,
                  wamd_mapping* d_out, hipStream_t s);
int launch_result_groups(wamd_ctx* ctx, uint32_t n_tasks, const wamd_commit_task* d_tasks, const wamd_mapping* d_committed,
                         uint32_t n_reads, const uint32_t* d_read_first, const uint64_t* d_read_ids,
                         const wamd_group_params* prm, uint8_t* d_out, uint32_t out_capacity, uint32_t* d_out_bytes,
                         uint32_t* d_read_off, void* d_ws, hipStream_t s);

int records_static_init(int device);   // with `device` current; once per device
int launch_emit_lev_records_static(wamd_ctx* ctx, uint32_t n, const wamd_dist_task* d_tasks,
                                   const wamd_dist_result* d_results, const uint32_t* d_record_pos,
                                   const uint8_t* d_direction, uint8_t* d_records, uint32_t* d_count, hipStream_t s);

#ifdef __HIPCC__
// ---- device helpers -------------------------------------------------------------

// swap the two nibbles of every byte: afterwards symbol s of the dword (s = 0..7 in
// genome order) sits at bits [4s+3 : 4s].
__device__ __forceinline__ uint32_t nibble_swap(uint32_t d)
{
    return ((d & 0x0f0f0f0fu) << 4) | ((d >> 4) & 0x0f0f0f0fu);
}

__device__ __forceinline__ uint32_t load_genome_dword(const GenomeView& g, uint32_t idx)
{
    idx = idx < g.last_dword ? idx : g.last_dword;
    return nibble_swap(g.dwords[idx]);
}

// Four consecutive raw genome dwords from idx on with ONE 16-byte load (dword-aligned).  A
// lane's window lives in its own cache lines, so what a wave's load instruction costs is the 64
// lines it touches, not the bytes: four times fewer instructions for the same text.  The
// genome's very last dwords are fetched one by one (clamped like load_genome_dword).
typedef uint32_t wamd_u32x4 __attribute__((ext_vector_type(4)));
typedef wamd_u32x4 wamd_u32x4_a4 __attribute__((aligned(4)));
__device__ __forceinline__ void load_genome_raw4(const GenomeView& g, uint32_t idx, uint32_t (&out)[4])
{
    if (idx + 3u <= g.last_dword) {
        const wamd_u32x4_a4 v = *reinterpret_cast<const wamd_u32x4_a4*>(g.dwords + idx);
        out[0] = v.x;
        out[1] = v.y;
        out[2] = v.z;
        out[3] = v.w;
    } else {
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c)
            out[c] = g.dwords[idx + c < g.last_dword ? idx + c : g.last_dword];
    }
}

// one symbol of a packed sequence (GET_CHAR_FROM_GENOME / _PATTERN, LevMyers64.cpp:22-26)
__device__ __forceinline__ uint32_t packed_symbol(const uint8_t* p, uint64_t i)
{
    uint32_t b = p[i >> 1];
    return (i & 1) ? (b & 15u) : (b >> 4);
}

// rev_comp_code, LevMyers64.cpp:40-48
__device__ __forceinline__ uint32_t rc_code(uint32_t c)
{
    return c < 4 ? 3 - c : (c < 8 ? c : 7);
}

__device__ __forceinline__ void raise_flag(uint32_t* flags, uint32_t f)
{
    atomicOr(flags, f);
}
#endif

} // namespace wamd
