// finish_pairs.hip - the chosen pair from its two mapping records to the SAM fields, on the device.
//
// Replaces what store_mapped_pair_reads does between picking a pair and writing its records
// (mapper/sam.cpp:1871-1900, 1992-2007):
//   fill_cigar (sam_matching.cpp:667-686) -> fill_cigar_with_lev (689-752) | fill_cigar_with_indel_matching
//   (824-852) | nothing for a perfect match; clipIllegalPositions, clipBoundaryPositions per read;
//   clipOverlappingPairs per pair; extract_cigar_and_mdz unless the method is Perfect / PerfectDistant.
// The verifier kernels of this library do the work (align, indel_ext_cigar, clip_*, sam_cigar); this file
// holds the glue the reference has between those calls, as three small kernels, so that task i, record i
// and ext-CIGAR slot i stay on the device from the mapping record to the CIGAR text:
//   finish_prepare  mapping_desc_t -> one dynamicProgramming task (lev) or one GetExtCigarNew task (the
//                   other types) per slot; the slot's other task is WAMD_READ_NONE and is skipped
//   finish_merge    what fill_cigar_with_lev / _indel_matching write back into the mapping_desc_t
//                   (delta of the begin column, ref_length, method, refined distance), as the records the
//                   clipping kernels rewrite, plus their task structs
//   finish_final    records + lengths -> wamd_finish_result; perfect matches get "<len>M" / "<len>"
#include "wamd_internal.h"

namespace wamd {

enum : uint32_t { FK_PERFECT = 0, FK_LEV = 1, FK_INDEL = 2, FK_BAD = 3 };

struct FinishWs {
    wamd_dist_task* dt;
    wamd_indel_cigar_task* ct;
    wamd_align_result* ar;
    wamd_indel_cigar_result* cr;
    wamd_clip_record* rec;
    wamd_clip_illegal_task* it;
    wamd_clip_pair_task* pt;
    wamd_sam_cigar_len* lens;
    uint32_t* cap;   // per slot: bytes its ext-CIGAR can need in the post-passes (their LDS packing)
    int2* mid;       // per slot: (ref_seq_pos, ref_length) between clipBoundaryPositions and clipOverlappingPairs
    void* align_ws;
    uint64_t align_ws_bytes;
    uint64_t total;
};

static inline uint64_t fup(uint64_t x) { return (x + 255) / 256 * 256; }

static FinishWs finish_layout(const wamd_reads* r, uint32_t n_pairs, char* base)
{
    const uint64_t n = 2ull * n_pairs;
    FinishWs w{};
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) {
        char* p = base + off;
        off += fup(bytes);
        return p;
    };
    w.dt = (wamd_dist_task*)take(n * sizeof(wamd_dist_task));
    w.ct = (wamd_indel_cigar_task*)take(n * sizeof(wamd_indel_cigar_task));
    w.ar = (wamd_align_result*)take(n * sizeof(wamd_align_result));
    w.cr = (wamd_indel_cigar_result*)take(n * sizeof(wamd_indel_cigar_result));
    w.rec = (wamd_clip_record*)take(n * sizeof(wamd_clip_record));
    w.it = (wamd_clip_illegal_task*)take(n * sizeof(wamd_clip_illegal_task));
    w.pt = (wamd_clip_pair_task*)take((uint64_t)n_pairs * sizeof(wamd_clip_pair_task));
    w.lens = (wamd_sam_cigar_len*)take(n * sizeof(wamd_sam_cigar_len));
    w.cap = (uint32_t*)take(n * sizeof(uint32_t));
    w.mid = (int2*)take(n * sizeof(int2));
    w.align_ws = base + off;
    w.align_ws_bytes = align_workspace_bytes(r, (uint32_t)n);
    w.total = off + w.align_ws_bytes;
    return w;
}

uint64_t finish_workspace_bytes(const wamd_reads* r, uint32_t n_pairs)
{
    return finish_layout(r, n_pairs, nullptr).total;
}

__device__ __forceinline__ uint32_t finish_kind(const wamd_finish_task& t, const ReadsView& rv)
{
    if (t.read_idx >= rv.n || t.type > WAMD_MAP_CLIPPING_CLIPPING)
        return FK_BAD;
    if (t.err_edit_distance == 0 && t.penalty == 0)   // sam_matching.cpp:670
        return FK_PERFECT;
    return t.type == WAMD_MAP_LEV ? FK_LEV : FK_INDEL;
}

__global__ __launch_bounds__(256) void finish_prepare_kernel(ReadsView rv, const wamd_finish_task* __restrict__ tasks,
                                                             uint32_t n, uint32_t flank,
                                                             wamd_dist_task* __restrict__ dt,
                                                             wamd_indel_cigar_task* __restrict__ ct,
                                                             uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const wamd_finish_task t = tasks[i];
    const uint32_t kind = finish_kind(t, rv);
    wamd_dist_task d{};
    wamd_indel_cigar_task c{};
    d.read_idx = WAMD_READ_NONE;
    c.read_idx = WAMD_READ_NONE;
    if (kind == FK_BAD) {
        raise_flag(flags, t.read_idx >= rv.n ? FLAG_READ_IDX : FLAG_ARG);
    } else if (kind == FK_LEV) {   // sam_matching.cpp:696-703
        const uint32_t m = rv.lens[t.read_idx];
        const uint32_t begin = t.direct ? t.pos : t.pos - m + 1;
        d.read_idx = t.read_idx;
        d.ref_pos = begin - flank;
        d.max_distance_in_ref = m + 2 * flank + 1;
        d.max_ed = t.err_edit_distance;
        d.variant = WAMD_VARIANT_AUTO;
        d.flags = t.direct ? 0 : WAMD_TASK_FLAG_REV_COMP;
    } else if (kind == FK_INDEL) {   // sam_matching.cpp:826-838
        const uint32_t m = rv.lens[t.read_idx];
        c.read_idx = t.read_idx;
        c.examined_start_pos = t.direct ? t.pos : t.pos - (m - 1);
        c.type = t.type;
        c.matching_len1 = t.matching_len1;
        c.matching_len2 = t.matching_len2;
        c.indel_len1 = t.indel_len1;
        c.indel_len2 = t.indel_len2;
        c.clipping_len = t.clipping_len;
        c.flags = t.direct ? 0 : WAMD_TASK_FLAG_REV_COMP;
    }
    dt[i] = d;
    ct[i] = c;
}

// did fill_cigar leave this slot with a usable mapping (perfect, aligned, or rebuilt)?
__device__ __forceinline__ bool finish_slot_ok(uint32_t kind, const wamd_align_result& a, const wamd_indel_cigar_result& c)
{
    return kind == FK_PERFECT || (kind == FK_LEV && a.ok && a.edit_distance != RESULT_ERROR) || (kind == FK_INDEL && c.ok);
}

// mapping_pair_t::calculateInsertSize (common/types.h:482-515) > mean + MyersDev || < mean - MyersDev, sam.cpp:1881-1884
__device__ __forceinline__ bool finish_is_distant(int32_t id0, int32_t pos0, int32_t len0, bool direct0, int32_t id1,
                                                  int32_t pos1, int32_t len1, bool direct1, int32_t isize_min,
                                                  int32_t isize_max)
{
    int32_t isize = 0x7fffffff;
    if (id0 == id1) {
        const bool left_is_first = pos0 < pos1 || (pos0 == pos1 && direct0);
        const bool left_direct = left_is_first ? direct0 : direct1;
        if (direct0 != direct1 && left_direct)
            isize = left_is_first ? pos1 - pos0 + len1 : pos0 - pos1 + len0;
    }
    return isize > isize_max || isize < isize_min;
}

__device__ __forceinline__ uint16_t finish_method(uint32_t kind, bool distant)
{
    if (kind == FK_PERFECT)
        return distant ? WAMD_METHOD_PERFECT_DISTANT : WAMD_METHOD_PERFECT;
    return distant ? WAMD_METHOD_FROM_MAPPING_DISTANT : WAMD_METHOD_FROM_MAPPING;
}

__global__ __launch_bounds__(256) void finish_merge_kernel(ReadsView rv, const wamd_finish_task* __restrict__ tasks,
                                                           uint32_t n, uint32_t flank, double match_score,
                                                           const wamd_align_result* __restrict__ ar,
                                                           const wamd_indel_cigar_result* __restrict__ cr,
                                                           wamd_clip_record* __restrict__ rec,
                                                           wamd_clip_illegal_task* __restrict__ it,
                                                           wamd_clip_pair_task* __restrict__ pt,
                                                           wamd_finish_result* __restrict__ results,
                                                           uint8_t* __restrict__ ext, uint32_t ext_stride,
                                                           uint32_t* __restrict__ cap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const wamd_finish_task t = tasks[i], tm = tasks[i ^ 1u];
    const uint32_t kind = finish_kind(t, rv), km = finish_kind(tm, rv);
    const wamd_align_result a = ar[i];
    const wamd_indel_cigar_result c = cr[i];
    // A read fill_cigar cannot align (stat_mapped_pe_errors, sam_matching.cpp:748-751) keeps a null ext-CIGAR in the
    // reference, which clipBoundaryPositions and extract_cigar_and_mdz then dereference: it does not survive such a
    // record.  Here that slot alone is reported as not finished; its mate goes through the chain as usual and the
    // pair stays out of clipOverlappingPairs.
    const bool ok = finish_slot_ok(kind, a, c);
    const bool ok_mate = finish_slot_ok(km, ar[i ^ 1u], cr[i ^ 1u]);
    const uint32_t m = kind == FK_BAD ? 0u : rv.lens[t.read_idx];
    const uint32_t mm = km == FK_BAD ? 0u : rv.lens[tm.read_idx];

    // fill_cigar's `distant`: the host's, or (WAMD_DISTANT_FROM_PAIR) what sam.cpp:1881-1884 computes from the pair
    // as it stands - for the first read both mappings as constructed (ref_length = read length, sam.cpp:1581); the
    // second read's needs the first one's record after its clipping passes: finish_final_kernel settles that one
    bool distant = t.distant == 1;
    if (t.distant == WAMD_DISTANT_FROM_PAIR && !(i & 1u))
        distant = finish_is_distant(t.ref_seq_id, t.ref_seq_pos, (int32_t)m, t.direct != 0, tm.ref_seq_id, tm.ref_seq_pos,
                                    (int32_t)mm, tm.direct != 0, t.isize_min, t.isize_max);

    wamd_clip_record r{};
    wamd_finish_result o{};
    wamd_clip_illegal_task ti{};
    r.ref_seq_pos = t.ref_seq_pos;
    r.ref_length = (uint16_t)m;   // sam.cpp:1581
    r.err_edit_distance = t.err_edit_distance;
    r.score = -1;
    o.pos = t.pos;
    o.method = WAMD_METHOD_UNMATCHED;
    o.ok = ok;
    ti.ref_size = t.ref_size;
    ti.read_length = (uint16_t)m;
    if (!ok) {
        // inert for the passes that follow: inside its chromosome, no errors, an empty ext-CIGAR
        r.ref_seq_pos = 1;
        r.ref_length = 0;
        r.err_edit_distance = 0;
        ti.ref_size = 0x7fffffff;
        ext[(uint64_t)i * ext_stride] = 0;
    } else if (kind == FK_PERFECT) {   // sam_matching.cpp:670-680
        o.method = finish_method(kind, distant);
        r.score = m * match_score;
        ti.perfect = 1;
        ext[(uint64_t)i * ext_stride] = 0;
    } else if (kind == FK_LEV) {       // sam_matching.cpp:723-746
        const uint32_t begin = t.direct ? t.pos : t.pos - m + 1;
        const int64_t delta = ((int64_t)(begin - flank) + a.begin_pos) - (int64_t)begin;
        if (a.refined)
            r.err_edit_distance = (uint16_t)(t.err_edit_distance + 1);
        o.pos = (uint32_t)((int64_t)t.pos + delta);
        r.ref_seq_pos = (int32_t)((int64_t)t.ref_seq_pos + delta);
        r.ref_length = (uint16_t)(a.end_pos - a.begin_pos);
        o.method = finish_method(kind, distant);
        o.num_events = (uint16_t)a.num_events;
        r.score = a.affine_score;
    } else {                           // sam_matching.cpp:840-848
        if (!(t.flags & WAMD_FINISH_NO_TRANSLATE))
            o.pos = t.direct ? t.pos : t.pos - (m - 1);
        r.score = c.score;
        r.err_edit_distance = c.err_edit_distance;
        r.ref_length = c.ref_length;
        r.left_clipping = c.left_clipping;
        r.right_clipping = c.right_clipping;
        o.method = c.method;
        o.num_events = c.num_events;
    }
    rec[i] = r;
    it[i] = ti;
    results[i] = o;
    {   // what the slot's string can grow to: itself (clipping never lengthens it) or read_length dots, NUL, slack
        const uint32_t len = !ok ? 0u : (kind == FK_LEV ? a.cigar_len : (kind == FK_PERFECT ? 0u : c.cigar_len));
        cap[i] = (len > m ? len : m) + 1 + 8;
    }
    if ((i & 1u) == 0) {   // clipOverlappingPairs' view of the pair, soft_clipping.cpp:997-1044
        wamd_clip_pair_task p{};
        p.ref_seq_id[0] = t.ref_seq_id;
        p.ref_seq_id[1] = (ok && ok_mate) ? tm.ref_seq_id : t.ref_seq_id + 1;   // a pair with an unfinished read is left alone
        p.read_length[0] = (uint16_t)m;
        p.read_length[1] = (uint16_t)mm;
        p.direct[0] = t.direct;
        p.direct[1] = tm.direct;
        p.perfect[0] = ok && kind == FK_PERFECT;
        p.perfect[1] = ok_mate && km == FK_PERFECT;
        pt[i >> 1] = p;
    }
}

__device__ __forceinline__ uint32_t finish_put_decimal(uint8_t* dst, uint32_t v)
{
    uint32_t n = 1;
    for (uint32_t t = v; t >= 10; t /= 10)
        ++n;
    for (uint32_t i = n; i-- > 0; v /= 10)
        dst[i] = (uint8_t)('0' + v % 10);
    return n;
}

__global__ __launch_bounds__(256) void finish_final_kernel(ReadsView rv, const wamd_finish_task* __restrict__ tasks,
                                                           uint32_t n, const wamd_clip_record* __restrict__ rec,
                                                           const wamd_align_result* __restrict__ ar,
                                                           const wamd_indel_cigar_result* __restrict__ cr,
                                                           const int2* __restrict__ mid,
                                                           const wamd_sam_cigar_len* __restrict__ lens,
                                                           wamd_finish_result* __restrict__ results,
                                                           const uint8_t* __restrict__ ext, uint32_t ext_stride,
                                                           uint8_t* __restrict__ cigars, uint32_t cigar_stride,
                                                           uint8_t* __restrict__ mdz, uint32_t mdz_stride)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    wamd_finish_result o = results[i];
    const wamd_clip_record r = rec[i];
    uint8_t* cg = cigars + (uint64_t)i * cigar_stride;
    uint8_t* mz = mdz + (uint64_t)i * mdz_stride;
    if (!o.ok) {
        wamd_finish_result z{};
        results[i] = z;
        cg[0] = mz[0] = 0;
        return;
    }
    if ((i & 1u) && tasks[i].distant == WAMD_DISTANT_FROM_PAIR &&
        (o.method == WAMD_METHOD_PERFECT || o.method == WAMD_METHOD_FROM_MAPPING)) {
        // sam.cpp:1881-1884 for the second read: the first mapping as its own fill_cigar, clipIllegalPositions and
        // clipBoundaryPositions left it (as constructed when it could not be aligned), this one as constructed
        const wamd_finish_task t = tasks[i], t0 = tasks[i ^ 1u];
        const uint32_t k0 = finish_kind(t0, rv);
        const int32_t m1 = (int32_t)rv.lens[t.read_idx];
        int32_t pos0 = t0.ref_seq_pos, len0 = k0 == FK_BAD ? 0 : (int32_t)rv.lens[t0.read_idx];
        if (finish_slot_ok(k0, ar[i ^ 1u], cr[i ^ 1u])) {
            pos0 = mid[i ^ 1u].x;
            len0 = mid[i ^ 1u].y;
        }
        if (finish_is_distant(t0.ref_seq_id, pos0, len0, t0.direct != 0, t.ref_seq_id, t.ref_seq_pos, m1, t.direct != 0,
                              t.isize_min, t.isize_max))
            o.method = o.method == WAMD_METHOD_PERFECT ? WAMD_METHOD_PERFECT_DISTANT : WAMD_METHOD_FROM_MAPPING_DISTANT;
    }
    o.score = r.score;
    o.ref_seq_pos = r.ref_seq_pos;
    o.ref_length = r.ref_length;
    o.err_edit_distance = r.err_edit_distance;
    o.left_clipping = r.left_clipping;
    o.right_clipping = r.right_clipping;
    uint32_t el = lens[i].reserved;   // the fused post-passes measured it on their LDS copy
    if (el >= 2) {
        el -= 2;
    } else {
        el = 0;
        const uint8_t* e = ext + (uint64_t)i * ext_stride;
        while (el < ext_stride && e[el])
            ++el;
    }
    o.ext_cigar_len = el;
    if (o.method == WAMD_METHOD_PERFECT || o.method == WAMD_METHOD_PERFECT_DISTANT) {
        // perfect_cigar / perfect_mdz (sam.cpp:1504-1515): a perfect match keeps them even when a clipping
        // pass has given it an ext-CIGAR
        const uint32_t m = rv.lens[tasks[i].read_idx];
        uint32_t k = finish_put_decimal(cg, m);
        cg[k] = 'M';
        cg[k + 1] = 0;
        o.cigar_len = k + 1;
        k = finish_put_decimal(mz, m);
        mz[k] = 0;
        o.mdz_len = k;
    } else {
        o.cigar_len = lens[i].cigar_len;
        o.mdz_len = lens[i].mdz_len;
    }
    results[i] = o;
}

int launch_finish_pairs(wamd_ctx* ctx, const wamd_reads* r, const wamd_finish_task* d_tasks, uint32_t n_pairs,
                        const wamd_finish_params* prm, wamd_finish_result* d_results, uint8_t* d_ext, uint32_t ext_stride,
                        uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz, uint32_t mdz_stride, void* d_ws,
                        uint64_t ws_bytes, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (r->encoding != WAMD_READS_ASCII) {
        set_error("wamd_finish_pairs: post-processing takes raw (ASCII) reads, as preprocessRawSeq does");
        return WAMD_ERR_INVALID_ARG;
    }
    if (n_pairs > 0x3fffffffu || ((uintptr_t)d_ws & 255u) || ws_bytes < finish_workspace_bytes(r, n_pairs)) {
        set_error("wamd_finish_pairs: workspace too small or not 256-byte aligned");
        return WAMD_ERR_INVALID_ARG;
    }
    const uint32_t n = 2 * n_pairs;
    const FinishWs w = finish_layout(r, n_pairs, (char*)d_ws);
    const ReadsView rv = r->view();
    const uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(finish_prepare_kernel, dim3(blocks), dim3(256), 0, s, rv, d_tasks, n, prm->flank, w.dt, w.ct,
                       ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    // CSamGenerator's verifier objects only ever see preprocessRawSeq: last_symbol stays 0 (LevMyers64.cpp:35).
    // fill_cigar_with_lev gives up where dynamicProgramming returns false (sam_matching.cpp:704-705) and nothing of
    // a rejected record's distance reaches the output: the first pass always takes the banded kernels.
    int rc = launch_align(ctx, r, w.dt, n, 0, prm->enable_short_indel_refinement ? 1 : 0, w.ar, d_ext, ext_stride,
                          w.align_ws, w.align_ws_bytes, s, WAMD_DISTANCE_ACCEPTED_ONLY);
    if (rc)
        return rc;
    rc = launch_indel_ext_cigar(ctx, r, w.ct, n, w.cr, d_ext, ext_stride, s);
    if (rc)
        return rc;
    hipLaunchKernelGGL(finish_merge_kernel, dim3(blocks), dim3(256), 0, s, rv, d_tasks, n, prm->flank,
                       ctx->scoring.match, w.ar, w.cr, w.rec, w.it, w.pt, d_results, d_ext, ext_stride, w.cap);
    WAMD_HIP(hipGetLastError());
    rc = launch_post_passes(ctx, n, w.it, w.pt, w.rec, d_ext, ext_stride, w.cap, prm->enable_boundary_clipping != 0,
                            prm->enable_paired_clipping != 0, d_cigars, cigar_stride, d_mdz, mdz_stride, w.lens, w.mid, s);
    if (rc)
        return rc;
    hipLaunchKernelGGL(finish_final_kernel, dim3(blocks), dim3(256), 0, s, rv, d_tasks, n, w.rec, w.ar, w.cr, w.mid, w.lens, d_results,
                       d_ext, ext_stride, d_cigars, cigar_stride, d_mdz, mdz_stride);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

} // namespace wamd
