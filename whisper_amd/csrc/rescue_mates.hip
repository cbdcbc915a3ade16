// rescue_mates.hip - find_pairs<LevMyers>'s chain per hit (mapper/sam_matching.cpp:379-457) on the device.
//
// The reference aligns the mate of every hit in the window the insert-size model allows beside it, turns the alignment
// into a mapping_desc_t, clips it and hands the pair to its evaluator.  The verifier kernels of this library do the
// work (align = dynamicProgramming + getExtCigar + refine_ext_cigar; the event-driven clipping passes); this file holds
// the glue between them, as two small kernels, so that hit i's window, alignment, record and ext-CIGAR slot i stay on
// the device from the hit to the clipped mate:
//   rescue_prepare   hit -> the dynamicProgramming task of :392-400
//   rescue_merge     what :422-441 make of the alignment: paired_raw_pos, Translate's verdict, the two num_events tests,
//                    the mate's mapping_desc_t as the record the clipping passes rewrite
//   (clipIllegalPositions, clipBoundaryPositions: post_passes.hip)
//   rescue_final     record -> wamd_rescue_result
#include "wamd_internal.h"

namespace wamd {

struct RescueWs {
    wamd_dist_task* dt;
    wamd_align_result* ar;
    wamd_clip_record* rec;
    wamd_clip_illegal_task* it;
    uint32_t* cap;
    void* align_ws;
    uint64_t align_ws_bytes, total;
};

static inline uint64_t rup(uint64_t x) { return (x + 255) / 256 * 256; }

static RescueWs rescue_layout(const wamd_reads* r, uint32_t n, uint32_t max_ed, char* base)
{
    RescueWs w{};
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) {
        char* p = base + off;
        off += rup(bytes);
        return p;
    };
    w.dt = (wamd_dist_task*)take((uint64_t)n * sizeof(wamd_dist_task));
    w.ar = (wamd_align_result*)take((uint64_t)n * sizeof(wamd_align_result));
    w.rec = (wamd_clip_record*)take((uint64_t)n * sizeof(wamd_clip_record));
    w.it = (wamd_clip_illegal_task*)take((uint64_t)n * sizeof(wamd_clip_illegal_task));
    w.cap = (uint32_t*)take((uint64_t)n * sizeof(uint32_t));
    w.align_ws = base + off;
    w.align_ws_bytes = align_workspace_bytes(r, n, max_ed);
    w.total = off + w.align_ws_bytes;
    return w;
}

uint64_t rescue_workspace_bytes(const wamd_reads* r, uint32_t n, uint32_t max_ed)
{
    return rescue_layout(r, n, max_ed, nullptr).total;
}

__device__ __forceinline__ uint32_t rescue_start_pos(const wamd_rescue_task& t, uint32_t mate_len)
{
    // ref_pos_t arithmetic, :392-394
    return t.paired_dir_direct ? t.hit_pos + mate_len - t.distance_threshold : t.hit_pos + 1u - t.max_mate_edit_distance;
}

__global__ __launch_bounds__(256) void rescue_prepare_kernel(ReadsView rv, const wamd_rescue_task* __restrict__ tasks, uint32_t n,
                                                             wamd_dist_task* __restrict__ dt, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const wamd_rescue_task t = tasks[i];
    wamd_dist_task d{};
    d.read_idx = WAMD_READ_NONE;
    if (t.mate_read_idx >= rv.n) {
        raise_flag(flags, FLAG_READ_IDX);
    } else {
        d.read_idx = t.mate_read_idx;
        d.ref_pos = rescue_start_pos(t, rv.lens[t.mate_read_idx]);
        d.max_distance_in_ref = t.distance_threshold + 2u * t.max_mate_edit_distance;
        d.max_ed = t.max_mate_edit_distance;
        d.variant = WAMD_VARIANT_AUTO;       // by the mate's length, :331-337
        d.flags = t.paired_dir_direct ? 0 : WAMD_TASK_FLAG_REV_COMP;
    }
    dt[i] = d;
}

__global__ __launch_bounds__(256) void rescue_merge_kernel(ReadsView rv, const wamd_rescue_task* __restrict__ tasks, uint32_t n,
                                                           const wamd_align_result* __restrict__ ar,
                                                           wamd_clip_record* __restrict__ rec,
                                                           wamd_clip_illegal_task* __restrict__ it,
                                                           wamd_rescue_result* __restrict__ results, uint8_t* __restrict__ ext,
                                                           uint32_t ext_stride, uint32_t* __restrict__ cap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const wamd_rescue_task t = tasks[i];
    const wamd_align_result a = ar[i];
    const uint32_t m = t.mate_read_idx < rv.n ? rv.lens[t.mate_read_idx] : 0u;
    wamd_rescue_result o{};
    wamd_clip_record r{};
    wamd_clip_illegal_task ti{};
    const bool aligned = m && a.ok && a.edit_distance != RESULT_ERROR;
    bool ok = false;
    if (aligned) {
        const uint32_t paired_raw_pos = rescue_start_pos(t, m) + a.begin_pos;   // :422
        const uint32_t p1 = paired_raw_pos + 1u;
        const bool same_seq = p1 > t.seq_lo && p1 <= t.seq_hi;                  // Translate gives the hit's chromosome
        const uint32_t half = t.max_mate_edit_distance / 2u;
        ok = same_seq && a.num_events <= half && t.hit_num_events <= half;      // :426-427
        o.aligned = 1;
        o.edit_distance = (uint16_t)(a.edit_distance + (a.refined ? 1u : 0u));
        o.pos = paired_raw_pos;
        o.cigar_events = a.num_events;
        o.ref_seq_pos = (int32_t)((int64_t)paired_raw_pos + t.seq_delta);
        r.ref_seq_pos = o.ref_seq_pos;
        r.ref_length = (uint16_t)(a.end_pos - a.begin_pos);                     // :439
        r.err_edit_distance = (uint16_t)(a.edit_distance + (a.refined ? 1u : 0u));   // construct<lev>(.., edit_distance), :433
        r.score = a.affine_score;                                               // :442
        ti.ref_size = t.ref_size;
        ti.read_length = (uint16_t)m;
    }
    o.ok = ok;
    if (!ok) {
        // inert for the passes that follow: inside its chromosome, no errors, an empty ext-CIGAR (the reference frees it)
        r = wamd_clip_record{};
        r.ref_seq_pos = 1;
        ti.ref_size = 0x7fffffff;
        ti.read_length = (uint16_t)m;
        ext[(uint64_t)i * ext_stride] = 0;
    }
    rec[i] = r;
    it[i] = ti;
    results[i] = o;
    const uint32_t len = ok ? a.cigar_len : 0u;
    cap[i] = (len > m ? len : m) + 1 + 8;
}

__global__ __launch_bounds__(256) void rescue_final_kernel(uint32_t n, const wamd_clip_record* __restrict__ rec,
                                                           wamd_rescue_result* __restrict__ results)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    wamd_rescue_result o = results[i];
    if (o.ok) {
        const wamd_clip_record r = rec[i];
        o.score = r.score;
        o.ref_seq_pos = r.ref_seq_pos;
        o.ref_length = r.ref_length;
        o.err_edit_distance = r.err_edit_distance;
        o.left_clipping = r.left_clipping;
        o.right_clipping = r.right_clipping;
        o.ext_cigar_len = r.cigar_len;
    }
    results[i] = o;
}

int launch_rescue_mates(wamd_ctx* ctx, const wamd_reads* r, const wamd_rescue_task* d_tasks, uint32_t n,
                        const wamd_rescue_params* prm, wamd_rescue_result* d_results, uint8_t* d_ext, uint32_t ext_stride,
                        void* d_ws, uint64_t ws_bytes, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (r->encoding != WAMD_READS_ASCII) {
        set_error("wamd_rescue_mates: post-processing takes raw (ASCII) reads, as preprocessRawSeq does");
        return WAMD_ERR_INVALID_ARG;
    }
    const RescueWs w = rescue_layout(r, n, 31, (char*)d_ws);
    (void)ws_bytes;
    const ReadsView rv = r->view();
    const uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(rescue_prepare_kernel, dim3(blocks), dim3(256), 0, s, rv, d_tasks, n, w.dt, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    // CSamGenerator's verifier objects only ever see preprocessRawSeq: last_symbol stays 0.  Nothing of a rejected
    // window's distance is used (:402), so the banded first pass would do - but the band never fits here (it needs
    // W + 2k - 61 <= m, and the window is the insert-size model's, several reads long): its two kernels only found
    // that out for every task (0.13 ms per 2^18), so the first pass is the whole-column kernel at once
    int rc = launch_align(ctx, r, w.dt, n, 0, prm->enable_short_indel_refinement ? 1 : 0, w.ar, d_ext, ext_stride, w.align_ws,
                          (ws_bytes - (uint64_t)((char*)w.align_ws - (char*)d_ws)) & ~255ull, s, WAMD_DISTANCE_EXACT);
    if (rc)
        return rc;
    hipLaunchKernelGGL(rescue_merge_kernel, dim3(blocks), dim3(256), 0, s, rv, d_tasks, n, w.ar, w.rec, w.it, d_results, d_ext,
                       ext_stride, w.cap);
    WAMD_HIP(hipGetLastError());
    rc = launch_clip_records(ctx, n, w.it, w.rec, d_ext, ext_stride, w.cap, prm->enable_boundary_clipping != 0, s);
    if (rc)
        return rc;
    hipLaunchKernelGGL(rescue_final_kernel, dim3(blocks), dim3(256), 0, s, n, w.rec, d_results);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

} // namespace wamd
