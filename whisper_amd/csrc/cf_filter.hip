// cf_filter.hip - the CF pre-filter of the mapping loops, one candidate per lane.
//
// Replaces calculate_CF_128b_SEE_for_dir_and_rc_gen + calculate_CF_128b_SSE_for_a_pattern +
// test_CF_128b_SSE_difference (mapper/mapping_core_dp.cpp:349-483, mapping_core.h:260-263):
// the candidate reaches the verifier iff the dinucleotide histograms of the read and of the
// genome under it, both taken outside the anchor, differ by at most 4k in L1.
//
// What the reference's byte walks add up - whatever the parity of the start and of the
// length - is every adjacent symbol pair that lies inside a region, one-hot per ACGT x ACGT
// pair (CF_SSE_lut, mapping_core.cpp:196-217; a pair with any other symbol adds nothing),
// with int8 saturation.  Here a dword of eight 4-bit symbols becomes four class masks (one
// flag bit per nibble); a pair (X, Y) at position p is "class X at p" & "class Y at p + 1",
// the latter a funnel shift by one nibble, and sixteen v_bcnt count them.  The other strand
// is counted in the stored read's own coordinates and the bins are permuted at the end
// ((a, b) seen from the other strand is (3 - b, 3 - a)).
#include "wamd_internal.h"

namespace wamd {

struct Classes {
    uint32_t c[4];   // bit 4p set when symbol p of the dword is A / C / G / T
};

__device__ __forceinline__ Classes classify(uint32_t s)
{
    const uint32_t m = 0x11111111u;
    const uint32_t valid = ~((s >> 2) | (s >> 3)) & m;
    const uint32_t b0 = s & m, b1 = (s >> 1) & m;
    Classes k;
    k.c[0] = valid & ~b0 & ~b1;
    k.c[1] = valid & b0 & ~b1;
    k.c[2] = valid & ~b0 & b1;
    k.c[3] = valid & b0 & b1;
    return k;
}

__device__ __forceinline__ uint32_t low_nibbles(int32_t n)   // flag bits of nibbles 0 .. n-1
{
    n = n < 0 ? 0 : (n > 8 ? 8 : n);
    return n == 8 ? 0x11111111u : (0x11111111u & ((1u << (4 * n)) - 1u));
}

// flag bits of the positions of dword d that fall into [lo, hi)
__device__ __forceinline__ uint32_t range_flags(int32_t lo, int32_t hi, int32_t d)
{
    return low_nibbles(hi - 8 * d) & ~low_nibbles(lo - 8 * d);
}

// Adds to cnt the pairs starting in [lo1, hi1) or [lo2, hi2) (symbol indices into `seq`,
// a packed 4-bit sequence read as dwords 0 .. last_dw).
// load(d): dword d; load4(d, out): dwords d .. d + 3 with one 16-byte load.
template <class Load, class Load4>
__device__ __forceinline__ void count_pairs(Load load, Load4 load4, int32_t lo1, int32_t hi1, int32_t lo2, int32_t hi2,
                                            uint32_t (&cnt)[16])
{
    const bool r1 = lo1 < hi1, r2 = lo2 < hi2;   // the caller guarantees at least one
    const int32_t lo = r1 && r2 ? min(lo1, lo2) : (r1 ? lo1 : lo2);
    const int32_t hi = r1 && r2 ? max(hi1, hi2) : (r1 ? hi1 : hi2);
    const int32_t first = lo >> 3, last = (hi - 1) >> 3;
    Classes cur = classify(load(first));
    // four dwords per round trip, fetched by one 16-byte load
    for (int32_t d0 = first; d0 <= last; d0 += 4) {
        uint32_t raw[4];
        load4(d0 + 1, raw);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int32_t d = d0 + q;
            const Classes nxt = classify(raw[q]);
            if (d <= last) {
                const uint32_t r = range_flags(lo1, hi1, d) | range_flags(lo2, hi2, d);
                uint32_t second[4];
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    second[b] = __builtin_amdgcn_alignbit(nxt.c[b], cur.c[b], 4);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const uint32_t fa = cur.c[a] & r;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        cnt[4 * a + b] += (uint32_t)__builtin_popcount(fa & second[b]);
                }
            }
            cur = nxt;
        }
    }
}

// The filter's answer for one candidate: the L1 distance of the two histograms as the reference's
// horizontal_add leaves it; `err` (FLAG_*) when the task is malformed.
__device__ __forceinline__ uint32_t cf_value(const GenomeView& g, const ReadsView& rv, const wamd_cf_task& t,
                                             uint32_t ref_size, uint32_t cur_offset, uint32_t& err)
{
    err = 0;
    uint32_t m = 0, left_end = 0;
    const uint32_t plen = t.pattern_len, fl = t.fixed_left, fr = t.fixed_right;
    if (t.read_idx >= rv.n) {
        err = FLAG_READ_IDX;
    } else {
        m = rv.lens[t.read_idx];
        // mapping_core_dp.cpp:353-356
        left_end = t.dir_gen_flag ? t.position - cur_offset : ref_size - t.position - plen + cur_offset;
        if ((uint32_t)t.read_offset + plen > m || fl > fr || fr > plen || fl == 1 || plen - fr == 1)
            err = FLAG_ARG;   // a one-symbol region makes the reference's byte walk underflow
        else if ((uint64_t)left_end + plen + 1 > g.n_symbols)
            err = FLAG_RANGE;
    }
    if (err)
        return RESULT_ERROR;

    // pair starts of the two regions, relative to the window's first symbol
    const int32_t p_lo1 = 0, p_hi1 = fl ? (int32_t)fl - 1 : 0;
    const int32_t p_lo2 = (int32_t)fr, p_hi2 = fr != plen ? (int32_t)plen - 1 : (int32_t)fr;

    uint32_t cg[16], cr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        cg[i] = cr[i] = 0;

    if (p_hi1 > p_lo1 || p_hi2 > p_lo2) {
        {
            const int32_t w0 = (int32_t)(left_end & 7u);   // count from the dword the window starts in
            const uint32_t d0 = left_end >> 3;
            auto load = [&](int32_t d) -> uint32_t { return load_genome_dword(g, d0 + (uint32_t)d); };
            auto load4 = [&](int32_t d, uint32_t (&out)[4]) {
                load_genome_raw4(g, d0 + (uint32_t)d, out);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    out[q] = nibble_swap(out[q]);
            };
            count_pairs(load, load4, w0 + p_lo1, w0 + p_hi1, w0 + p_lo2, w0 + p_hi2, cg);
        }
        const uint32_t* codes = reinterpret_cast<const uint32_t*>(rv.codes + rv.code_off[t.read_idx]);
        const int32_t last_dw = (int32_t)((m + 7) / 8) - 1;
        auto load = [&](int32_t d) -> uint32_t { return d <= last_dw ? nibble_swap(codes[d]) : 0xffffffffu; };
        // the codes buffer has 64 bytes of slack behind the last read
        auto load4 = [&](int32_t d, uint32_t (&out)[4]) {
            const wamd_u32x4 v = d <= last_dw ? *reinterpret_cast<const wamd_u32x4_a4*>(codes + d) : wamd_u32x4{0, 0, 0, 0};
            const uint32_t b[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[q] = d + q <= last_dw ? nibble_swap(b[q]) : 0xffffffffu;
        };
        if (!(t.flags & WAMD_TASK_FLAG_REV_COMP)) {
            const int32_t w0 = (int32_t)t.read_offset;
            count_pairs(load, load4, w0 + p_lo1, w0 + p_hi1, w0 + p_lo2, w0 + p_hi2, cr);
        } else {
            // oriented pair start i sits at stored pair start m - 2 - i
            const int32_t e = (int32_t)m - 1 - (int32_t)t.read_offset;   // one past the stored start of oriented pair 0
            uint32_t tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                tmp[i] = 0;
            count_pairs(load, load4, e - p_hi1, e - p_lo1, e - p_hi2, e - p_lo2, tmp);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    cr[4 * (3 - b) + (3 - a)] = tmp[4 * a + b];
        }
    }

    // add_saturated on int8 lanes, then horizontal_add(abs(.)) truncated to int8 and
    // sign-extended (libs/vectori128.h:1496-1502) - compared as unsigned by the caller
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int32_t a = (int32_t)min(cr[i], 127u), b = (int32_t)min(cg[i], 127u);
        sum += (uint32_t)(a > b ? a - b : b - a);
    }
    return (uint32_t)(int32_t)(int8_t)(sum & 0xffu);
}

__global__ __launch_bounds__(256) void cf_filter_kernel(GenomeView g, ReadsView rv,
                                                        const wamd_cf_task* __restrict__ tasks, uint32_t n,
                                                        uint32_t ref_size, uint32_t cur_offset,
                                                        uint32_t* __restrict__ results, uint32_t* __restrict__ flags)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n)
        return;
    uint32_t err;
    const uint32_t v = cf_value(g, rv, tasks[gid], ref_size, cur_offset, err);
    if (err)
        raise_flag(flags, err);
    results[gid] = v;
}

// =============================================================================================
// The filter in front of its verifier, as the suffix-array loops have it (mapping_core_sa.cpp:223-255,
// 279-307, 388-460): a candidate reaches dynamicProgramming / LevDiag only when the filter passes it,
// cur_match = max_mismatches + 1 otherwise.  One candidate per lane cannot skip work for the lanes the
// filter turns away - the wave runs as long as its slowest lane - so the survivors are packed first:
//   filter_pack_kernel    CF decision per candidate; rejected ones get their final answer, passed ones
//                         are copied (verifier task + index) to the front of a list, in task order inside
//                         a block, blocks in the order they arrive (results go back by index, so the
//                         answers do not depend on that order);
//   the verifier's own kernels over the list, its length read from the device (n_dev);
//   filter_unpack_kernel  answers back to their candidates' slots.
// A CF task with pattern_len == 0 is the loops' is_short_read: no filter, always verified.
// =============================================================================================
template <class VTask>
__device__ __forceinline__ uint32_t verifier_budget(const VTask& t);
template <>
__device__ __forceinline__ uint32_t verifier_budget<wamd_dist_task>(const wamd_dist_task& t) { return t.max_ed; }
template <>
__device__ __forceinline__ uint32_t verifier_budget<wamd_levdiag_task>(const wamd_levdiag_task& t) { return t.max_mismatches; }

__device__ __forceinline__ void store_rejected(wamd_dist_result* results, uint32_t i, uint32_t value, uint32_t pos)
{
    results[i].edit_distance = value;
    results[i].pos = pos;
}
__device__ __forceinline__ void store_rejected(uint32_t* results, uint32_t i, uint32_t value, uint32_t) { results[i] = value; }

template <class VTask, class VResult>
__global__ __launch_bounds__(256) void filter_pack_kernel(GenomeView g, ReadsView rv, const wamd_cf_task* __restrict__ cf_tasks,
                                                          const VTask* __restrict__ v_tasks, uint32_t n, uint32_t ref_size,
                                                          uint32_t cur_offset, VResult* __restrict__ results,
                                                          VTask* __restrict__ packed_tasks, uint32_t* __restrict__ packed_idx,
                                                          uint32_t* __restrict__ count, uint32_t* __restrict__ flags)
{
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    bool pass = false;
    VTask vt{};
    if (gid < n) {
        const wamd_cf_task ct = cf_tasks[gid];
        vt = v_tasks[gid];
        const uint32_t k = verifier_budget<VTask>(vt);
        if (ct.pattern_len == 0) {
            pass = true;
        } else {
            uint32_t err;
            const uint32_t v = cf_value(g, rv, ct, ref_size, cur_offset, err);
            if (err) {
                raise_flag(flags, err);
                store_rejected(results, gid, RESULT_ERROR, err);
            } else if (v <= 4u * k) {   // test_CF_128b_SSE_difference(..., 4 * max_mismatches), mapping_core.h:260-263
                pass = true;
            } else {
                store_rejected(results, gid, k + 1u, 0u);
            }
        }
    }
    __shared__ uint32_t wave_cnt[4];
    __shared__ uint32_t block_base;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t mask = __ballot(pass);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
    if (lane == 0)
        wave_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t c = wave_cnt[w];
            wave_cnt[w] = total;
            total += c;
        }
        block_base = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    if (pass) {
        const uint32_t slot = block_base + wave_cnt[wave] + rank;
        packed_tasks[slot] = vt;
        packed_idx[slot] = gid;
    }
}

template <class VResult>
__global__ __launch_bounds__(256) void filter_unpack_kernel(const VResult* __restrict__ packed_results,
                                                            const uint32_t* __restrict__ packed_idx,
                                                            const uint32_t* __restrict__ count, VResult* __restrict__ results)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot < *count)
        results[packed_idx[slot]] = packed_results[slot];
}

// workspace: [count, 256 B][packed tasks][packed results][packed indices], every part 256-byte aligned
struct FilterLayout {
    uint64_t tasks, results, idx, total;
};
static FilterLayout filter_layout(uint32_t n, uint64_t task_bytes, uint64_t result_bytes)
{
    auto up = [](uint64_t x) { return (x + 255u) & ~255ull; };
    FilterLayout l;
    l.tasks = 256;
    l.results = l.tasks + up((uint64_t)n * task_bytes);
    l.idx = l.results + up((uint64_t)n * result_bytes);
    l.total = l.idx + up((uint64_t)n * 4u);
    return l;
}

uint64_t filter_verify_workspace_bytes(uint32_t n)
{
    static_assert(sizeof(wamd_dist_task) >= sizeof(wamd_levdiag_task) && sizeof(wamd_dist_result) >= sizeof(uint32_t),
                  "one size serves both verifiers");
    return filter_layout(n, sizeof(wamd_dist_task), sizeof(wamd_dist_result)).total;
}

int launch_filter_distance(wamd_ctx* ctx, const wamd_reads* r, const wamd_cf_task* d_cf, const wamd_dist_task* d_tasks,
                           uint32_t n, int last_symbol, wamd_dist_result* d_results, void* d_ws, hipStream_t s,
                           int distance_mode)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0)
        return WAMD_OK;
    const FilterLayout l = filter_layout(n, sizeof(wamd_dist_task), sizeof(wamd_dist_result));
    char* ws = (char*)d_ws;
    uint32_t* count = (uint32_t*)ws;
    auto* p_tasks = (wamd_dist_task*)(ws + l.tasks);
    auto* p_res = (wamd_dist_result*)(ws + l.results);
    auto* p_idx = (uint32_t*)(ws + l.idx);
    WAMD_HIP(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
    const uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL((filter_pack_kernel<wamd_dist_task, wamd_dist_result>), dim3(blocks), dim3(256), 0, s, genome_view(ctx),
                       r->view(), d_cf, d_tasks, n, ctx->ref_size, ctx->cur_offset, d_results, p_tasks, p_idx, count,
                       ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    const int rc = launch_distance(ctx, r, p_tasks, n, last_symbol, p_res, distance_mode < 0 ? ctx->distance_mode : distance_mode, s, count);
    if (rc)
        return rc;
    hipLaunchKernelGGL((filter_unpack_kernel<wamd_dist_result>), dim3(blocks), dim3(256), 0, s, p_res, p_idx, count, d_results);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_filter_levdiag(wamd_ctx* ctx, const wamd_reads* r, const wamd_cf_task* d_cf, const wamd_levdiag_task* d_tasks,
                          uint32_t n, uint32_t* d_results, void* d_ws, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0)
        return WAMD_OK;
    const FilterLayout l = filter_layout(n, sizeof(wamd_dist_task), sizeof(wamd_dist_result));
    char* ws = (char*)d_ws;
    uint32_t* count = (uint32_t*)ws;
    auto* p_tasks = (wamd_levdiag_task*)(ws + l.tasks);
    auto* p_res = (uint32_t*)(ws + l.results);
    auto* p_idx = (uint32_t*)(ws + l.idx);
    WAMD_HIP(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
    const uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL((filter_pack_kernel<wamd_levdiag_task, uint32_t>), dim3(blocks), dim3(256), 0, s, genome_view(ctx),
                       r->view(), d_cf, d_tasks, n, ctx->ref_size, ctx->cur_offset, d_results, p_tasks, p_idx, count,
                       ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    const int rc = launch_levdiag(ctx, r, p_tasks, n, p_res, s, count);
    if (rc)
        return rc;
    hipLaunchKernelGGL((filter_unpack_kernel<uint32_t>), dim3(blocks), dim3(256), 0, s, p_res, p_idx, count, d_results);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_cf_filter(wamd_ctx* ctx, const wamd_reads* r, const wamd_cf_task* d_tasks, uint32_t n,
                     uint32_t* d_results, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0)
        return WAMD_OK;
    hipLaunchKernelGGL(cf_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, genome_view(ctx), r->view(),
                       d_tasks, n, ctx->ref_size, ctx->cur_offset, d_results, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

} // namespace wamd
