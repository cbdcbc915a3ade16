// levdiag.hip - anchored banded verification, one candidate per lane.
//
// Replaces CEditDist::LevDiag (mapper/edit_dist.cpp:106-344), the verifier of every stage
// with max_mismatches < lev_alg_thr = 7 (common/defs.h:108, mapper/mapping_core_sa.cpp:236-239):
// the anchor must match exactly and is grown greedily, then the read prefix (walking left)
// and the suffix (walking right, with the budget the prefix left over) each run through a
// DP over the 2k+1 diagonals around the anchor's, thresholded at the budget.
//
// What the reference's band computes (restated cell by cell in oracle/wv_oracle.c) is the
// plain edit-distance recurrence with the boundary row D[-1][delta] = |delta| on diagonals
// delta = -kk .. kk, free text end, minimum over the last row, "kk + 1" when that minimum
// exceeds kk: narrowing only ever drops edge cells that are already above kk, and the
// single-diagonal shortcut (edit_dist.cpp:175-189) is taken exactly when that diagonal
// holds kk.  Values <= kk of that recurrence do not depend on how it is evaluated, so the
// kernel evaluates it the cheap way for a thresholded distance: furthest-reaching rows per
// (errors, diagonal) with word-wide match runs (Landau-Vishkin / Ukkonen) - (kk+1)^2 run
// extensions per side instead of (2kk+1) x rows cells.
//
// MI355X mapping: the read and its genome window are staged once per lane in LDS as
// 4-bit symbols, [dword][lane], so a lane fetches any 8-symbol window of its own data with
// two conflict-free ds_read_b32 and a funnel shift, and a run extension compares 8 symbols
// per step (xor + count of trailing / leading zero nibbles).  The 13 diagonals live in VGPRs
// with compile-time indices.
#include "wamd_internal.h"

namespace wamd {

constexpr int LD_K = 6;   // lev_alg_thr - 1
constexpr int LD_NONE = -100000;

// 8 symbols starting at symbol s (s >= -8) of a staged sequence: symbol i sits at nibble i + 8
template <int BLOCK>
__device__ __forceinline__ uint32_t window8(const uint32_t* lane, int s)
{
    const uint32_t p = (uint32_t)(s + 8);
    const uint32_t q = p >> 3;
    return __builtin_amdgcn_alignbit(lane[(q + 1) * BLOCK], lane[q * BLOCK], (p & 7u) * 4u);
}

// length of the common run of pat[pi ...] and gen[gi ...] walking by dir (+1 / -1), at most
// limit symbols; lanes with limit <= 0 get 0.  Wave-uniform loop.
template <int BLOCK, int DIR>
__device__ __forceinline__ int common_run(const uint32_t* pat, const uint32_t* gen, int pi, int gi, int limit)
{
    int n = 0;
    bool go = limit > 0;
    while (__any(go)) {
        if (go) {
            uint32_t x;
            int c;
            if (DIR > 0) {
                x = window8<BLOCK>(pat, pi + n) ^ window8<BLOCK>(gen, gi + n);
                c = x ? (__builtin_ctz(x) >> 2) : 8;
            } else {
                x = window8<BLOCK>(pat, pi - n - 7) ^ window8<BLOCK>(gen, gi - n - 7);
                c = x ? (__builtin_clz(x) >> 2) : 8;
            }
            n += c;
            if (c < 8 || n >= limit)
                go = false;
        }
    }
    return n < limit ? n : (limit > 0 ? limit : 0);
}

// One side of LevDiag for every lane of the wave (rows = 0: the lane sits this side out).
// Row r of the side is pattern symbol p0 + DIR * r; on diagonal delta it faces window symbol
// (p0 + DIR * r) + k + DIR * delta.  Returns the side's error count, kk + 1 when above kk.
template <int BLOCK, int DIR>
__device__ int levdiag_side(const uint32_t* pat, const uint32_t* gen, int p0, int k, int rows, int kk)
{
    // furthest row reached (last consumed pattern row of the side) per diagonal, index delta + LD_K
    int prev[2 * LD_K + 3], cur[2 * LD_K + 3];   // one guard cell at either end
#pragma unroll
    for (int i = 0; i < 2 * LD_K + 3; ++i)
        prev[i] = cur[i] = LD_NONE;
    int result = kk + 1;
    bool open = rows > 0;
    if (rows <= 0)
        result = 0;

    for (int e = 0; __any(open && e <= kk); ++e) {
        const bool lane_on = open && e <= kk;
#pragma unroll
        for (int delta = -LD_K; delta <= LD_K; ++delta) {
            const int ad = delta < 0 ? -delta : delta;
            if (ad > e)   // wave-uniform: e is
                continue;
            const int i = delta + LD_K + 1;
            // boundary row D[-1][delta] = |delta| (edit_dist.cpp:165-166), then the three moves
            int base = ad == e ? -1 : LD_NONE;
            const int sub = prev[i] + 1;          // mismatch: next row, same diagonal
            const int hor = prev[i - 1];          // text symbol skipped: same row, from delta - 1
            const int ver = prev[i + 1] + 1;      // pattern symbol unmatched: next row, from delta + 1
            base = sub > base ? sub : base;
            base = hor > base ? hor : base;
            base = ver > base ? ver : base;
            const bool reach = lane_on && ad <= kk && base >= -1;
            const int pi = p0 + DIR * (base + 1);
            const int run = common_run<BLOCK, DIR>(pat, gen, pi, pi + k + DIR * delta, reach ? rows - 1 - base : 0);
            const int far = reach ? base + run : LD_NONE;
            cur[i] = far;
            if (reach && far >= rows - 1 && open) {
                result = e;
                open = false;
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * LD_K + 3; ++i)
            prev[i] = cur[i];
    }
    return result;
}

template <int PW, int BLOCK>
__global__ __launch_bounds__(BLOCK) void levdiag_kernel(GenomeView g, ReadsView rv,
                                                        const wamd_levdiag_task* __restrict__ tasks, uint32_t n,
                                                        uint32_t ref_size, uint32_t cur_offset,
                                                        uint32_t* __restrict__ results, uint32_t* __restrict__ flags)
{
    constexpr int GW = PW + 2;
    __shared__ uint32_t pat_lds[PW * BLOCK];   // [dword][lane]: pad, read symbols, pad
    __shared__ uint32_t gen_lds[GW * BLOCK];   // [dword][lane]: pad, window of m + 2k + 1 symbols, pad

    const uint32_t tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * BLOCK + tid;
    bool valid = gid < n;
    uint32_t k = 0, fl = 0, fr = 0, m = 0, g0 = 0, read_idx = 0;
    if (valid) {
        const wamd_levdiag_task t = tasks[gid];
        uint32_t err = 0;
        k = t.max_mismatches;
        fl = t.fixed_left;
        fr = t.fixed_right;
        read_idx = t.read_idx;
        if (read_idx >= rv.n) {
            err = FLAG_READ_IDX;
        } else {
            m = rv.lens[read_idx];
            // window start, edit_dist.cpp:112-115 (unsigned, as the reference computes it)
            g0 = t.dir_gen_flag ? t.position_in_text - cur_offset - k
                                : ref_size - t.position_in_text - m + cur_offset - k;
            if (k > (uint32_t)LD_K || fl > fr || fr > m || m > 8u * (PW - 2))
                err = FLAG_ARG;
            else if ((uint64_t)g0 + m + 2ull * k + 1 > g.n_symbols)
                err = FLAG_RANGE;
        }
        if (err) {
            raise_flag(flags, err);
            results[gid] = RESULT_ERROR;
            valid = false;
        }
    }
    if (!valid) {
        k = fl = fr = m = g0 = read_idx = 0;
    }

    // ---- stage the read and its window: symbol i at nibble i + 8 --------------------------------
    uint32_t* pat = pat_lds + tid;
    uint32_t* gen = gen_lds + tid;
    {
        // 16-byte loads: what a wave's load costs is the 64 cache lines it touches, not its width
        // (the codes buffer has 64 bytes of slack behind the last read)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(rv.codes + rv.code_off[read_idx]);
        const uint32_t pd = valid ? (m + 7) / 8 : 0;
        pat[0] = 0;
#pragma unroll
        for (int j4 = 0; j4 < PW - 1; j4 += 4) {
            wamd_u32x4 v = {0, 0, 0, 0};
            if ((uint32_t)j4 < pd)
                v = *reinterpret_cast<const wamd_u32x4_a4*>(src + j4);
            const uint32_t b[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (j4 + c + 1 < PW)
                    pat[(j4 + c + 1) * BLOCK] = (uint32_t)(j4 + c) < pd ? nibble_swap(b[c]) : 0xffffffffu;
        }
        const uint32_t shift = (g0 & 7u) * 4u;
        const uint32_t d0 = g0 >> 3;
        const uint32_t gd = valid ? (m + 2 * k + 1 + 7) / 8 : 0;
        gen[0] = 0;
        uint32_t lo = load_genome_dword(g, d0);
#pragma unroll
        for (int j4 = 0; j4 < GW - 1; j4 += 4) {
            uint32_t b[4] = {0, 0, 0, 0};
            if ((uint32_t)j4 < gd)
                load_genome_raw4(g, d0 + 1u + (uint32_t)j4, b);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (j4 + c + 1 < GW) {
                    const uint32_t hi = nibble_swap(b[c]);
                    gen[(j4 + c + 1) * BLOCK] = (uint32_t)(j4 + c) < gd ? __builtin_amdgcn_alignbit(hi, lo, shift) : 0u;
                    lo = hi;
                }
            }
        }
    }

    bool alive = valid;
    // anchor must match exactly (edit_dist.cpp:121-124)
    {
        const int need = alive ? (int)(fr - fl) : 0;
        const int got = common_run<BLOCK, 1>(pat, gen, (int)fl, (int)(fl + k), need);
        if (alive && got < need)
            alive = false;
    }
    // grow the anchor to the left (edit_dist.cpp:150-154)
    {
        const int got = common_run<BLOCK, -1>(pat, gen, (int)fl - 1, (int)(fl + k) - 1, alive ? (int)fl : 0);
        fl -= (uint32_t)got;
    }
    int ep = 0;
    {
        const bool run = alive && fl > 0;
        if (__any(run)) {
            const int e = levdiag_side<BLOCK, -1>(pat, gen, (int)fl - 1, (int)k, run ? (int)fl : 0, (int)k);
            if (run) {
                ep = e;
                if (e > (int)k)
                    alive = false;
            }
        }
    }
    // grow the anchor to the right (edit_dist.cpp:248-252)
    {
        const int got = common_run<BLOCK, 1>(pat, gen, (int)fr, (int)(fr + k), alive ? (int)(m - fr) : 0);
        fr += (uint32_t)got;
    }
    int es = 0;
    {
        const bool run = alive && fr < m;
        const int kk = (int)k - ep;   // edit_dist.cpp:256
        if (__any(run)) {
            const int e = levdiag_side<BLOCK, 1>(pat, gen, (int)fr, (int)k, run ? (int)(m - fr) : 0, run ? kk : 0);
            if (run) {
                es = e;
                if (e > kk)
                    alive = false;
            }
        }
    }
    if (valid)
        results[gid] = alive ? (uint32_t)(ep + es) : k + 1;
}

template <int PW, int BLOCK>
static int launch_ld(wamd_ctx* ctx, const wamd_reads* r, const wamd_levdiag_task* d_tasks, uint32_t n,
                     uint32_t* d_results, hipStream_t s)
{
    const uint32_t blocks = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL((levdiag_kernel<PW, BLOCK>), dim3(blocks), dim3(BLOCK), 0, s, genome_view(ctx), r->view(),
                       d_tasks, n, ctx->ref_size, ctx->cur_offset, d_results, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_levdiag(wamd_ctx* ctx, const wamd_reads* r, const wamd_levdiag_task* d_tasks, uint32_t n,
                   uint32_t* d_results, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0)
        return WAMD_OK;
    // staged dwords per lane: read symbols / 8 plus a pad dword at either end
    const uint32_t need = (r->max_len + 7) / 8 + 2;
    if (need <= 22)
        return launch_ld<22, 256>(ctx, r, d_tasks, n, d_results, s);   // reads up to 160 symbols
    if (need <= 34)
        return launch_ld<34, 128>(ctx, r, d_tasks, n, d_results, s);   // up to 256
    if (need <= 66)
        return launch_ld<66, 64>(ctx, r, d_tasks, n, d_results, s);    // up to 512
    return launch_ld<130, 64>(ctx, r, d_tasks, n, d_results, s);       // up to 1024
}

} // namespace wamd
