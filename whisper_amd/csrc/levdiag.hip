// levdiag.hip - anchored banded verification, one candidate per lane.
//
// Replaces CEditDist::LevDiag (mapper/edit_dist.cpp:106-344), the verifier of
// every stage with max_mismatches < lev_alg_thr = 7 (common/defs.h:108,
// mapper/mapping_core_sa.cpp:236-239): the anchor must match exactly, is grown
// greedily, then the read prefix (walking left) and suffix (walking right, with
// the budget the prefix left over) run through a banded DP whose band narrows
// from either edge when an edge cell exceeds the budget.
//
// The reference indexes the band by absolute genome offset j and slides it one
// cell per row.  Here the band is held in diagonal coordinates d = j - row, which
// never move: 2k+1 <= 13 cells sit in VGPRs with compile-time indices, the
// genome symbols under the band are one 4-bit-per-cell shift register, and
// narrowing is two lane-private bounds.  Every comparison, tie and shortcut of
// the reference (right edge without an upper neighbour, the single-diagonal
// shortcut that reports exactly the budget) is kept, so results are bit-exact.
#include "wamd_internal.h"

namespace wamd {

constexpr int LD_K = 6;              // lev_alg_thr - 1
constexpr int LD_CELLS = 2 * LD_K + 1;

// forward / backward stream of 4-bit symbols out of packed memory
struct SymStream {
    const uint8_t* base;
    int64_t pos;
    int64_t limit;   // number of symbols addressable
    int step;
    __device__ __forceinline__ uint32_t get()
    {
        int64_t p = pos < 0 ? 0 : (pos >= limit ? limit - 1 : pos);
        uint32_t s = packed_symbol(base, (uint64_t)p);
        pos += step;
        return s;
    }
};

// One side of LevDiag for every lane of the wave.  rows = 0 for lanes that skip it.
// Returns the side's error count; kk + 1 means "failed".
//
// Coordinates: row r compares pattern symbol r of the side with genome symbols
// G(j), j = r + d, d = 1 .. 2kk+1.  P[d] holds the previous row's cell with the
// same d (its own coordinates), so for the current cell d the reference's
// prev[j-1] (diagonal) is P[d] and prev[j] (above) is P[d+1].
__device__ int levdiag_side(SymStream gen, SymStream pat, int rows, int kk)
{
    uint32_t P[LD_CELLS + 2], C[LD_CELLS + 2];
#pragma unroll
    for (int d = 0; d < LD_CELLS + 2; ++d) {
        const int v = d - 1 - kk;          // prev[kk +- t] = t   (edit_dist.cpp:165-166)
        P[d] = (uint32_t)(v < 0 ? -v : v);
        C[d] = 0;
    }
    int dl = 1, dr = 2 * kk + 1;           // left_side - r, right_side - r
    bool failed = false, single = false;
    int result = 0;

    // genome symbols under the band: nibble (d - 1) = G(r + d); nibbles above the
    // band stay zero.  The reference prefetches j = 1 .. 2kk and then one symbol per
    // row (edit_dist.cpp:169-170, 215): the same symbols, one row earlier.
    uint64_t gw = 0;
    for (int d = 1; d <= LD_CELLS; ++d)
        if (rows > 0 && d <= 2 * kk + 1)
            gw |= (uint64_t)gen.get() << (4 * (d - 1));

    for (int r = 0; __any(r < rows && !failed); ++r) {
        const bool act = r < rows && !failed;
        const uint32_t pc = act ? pat.get() : 0u;
        // bit 4(d-1) set where the band's genome symbol equals the pattern symbol
        uint64_t x = gw ^ (0x1111111111111111ull * pc);
        x |= x >> 1;
        x |= x >> 2;
        const uint64_t hit = ~x & 0x1111111111111111ull;

        if (act && !single && dl == dr) {
            // one diagonal left (edit_dist.cpp:175-189): prev[left-1] must be within the
            // budget, the rest of the side must match exactly, and the side then
            // reports exactly kk
            uint32_t v = 0;
#pragma unroll
            for (int d = 1; d <= LD_CELLS; ++d)
                v = (d == dl) ? P[d] : v;
            if ((int32_t)v > kk)
                failed = true;
            single = true;
            result = kk;
        }
        if (act && !failed && single) {
            if (!((hit >> (4 * (dl - 1))) & 1ull))
                failed = true;
        } else if (act && !failed) {
            uint32_t low = 0;   // prev[left - 1]
#pragma unroll
            for (int d = 1; d <= LD_CELLS; ++d)
                low = (d == dl) ? P[d] : low;
#pragma unroll
            for (int d = 1; d <= LD_CELLS; ++d) {
                const bool m = (hit >> (4 * (d - 1))) & 1ull;
                const uint32_t diag = P[d], up = P[d + 1];
                if (d >= dl && d < dr) {   // edit_dist.cpp:196-213
                    uint32_t v;
                    if (m) {
                        v = diag;
                        low = v < up ? v : up;
                    } else if (up > low) {
                        v = ++low;
                    } else {
                        low = up;
                        v = low + 1;
                    }
                    C[d] = v;
                } else if (d == dr) {      // right edge, no upper neighbour: 215-219
                    C[d] = m ? diag : low + 1;
                }
            }
            // narrow the band from either edge (edit_dist.cpp:222-227)
            uint32_t cl = 0, cr = 0;
#pragma unroll
            for (int d = 1; d <= LD_CELLS; ++d) {
                cl = (d == dl) ? C[d] : cl;
                cr = (d == dr) ? C[d] : cr;
            }
            if ((int32_t)cl > kk && ++dl > dr)
                failed = true;
            if (!failed && (int32_t)cr > kk && dl > --dr)
                failed = true;
            if (!failed && r == rows - 1) {
                // minimum over what is left of the last row (edit_dist.cpp:238-240)
                int best = kk + 1;
#pragma unroll
                for (int d = 1; d <= LD_CELLS; ++d)
                    if (d >= dl && d <= dr && (int32_t)C[d] < best)
                        best = (int32_t)C[d];
                result = best;
            }
#pragma unroll
            for (int d = 1; d <= LD_CELLS; ++d)
                P[d] = C[d];
        }
        // slide the genome window: the next row needs G(r + 1 + d)
        gw >>= 4;
        if (r + 1 < rows && !failed)
            gw |= (uint64_t)gen.get() << (8 * kk);
    }
    return failed ? kk + 1 : result;
}

__global__ __launch_bounds__(256) void levdiag_kernel(const uint8_t* __restrict__ genome,
                                                      uint64_t genome_symbols, ReadsView rv,
                                                      const wamd_levdiag_task* __restrict__ tasks,
                                                      uint32_t n, uint32_t ref_size,
                                                      uint32_t cur_offset,
                                                      uint32_t* __restrict__ results,
                                                      uint32_t* __restrict__ flags)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = gid < n;
    uint32_t k = 0, fl = 0, fr = 0, m = 0, g0 = 0;
    const uint8_t* pat = nullptr;
    if (valid) {
        const wamd_levdiag_task t = tasks[gid];
        uint32_t err = 0;
        k = t.max_mismatches;
        fl = t.fixed_left;
        fr = t.fixed_right;
        if (t.read_idx >= rv.n) {
            err = FLAG_READ_IDX;
        } else {
            m = rv.lens[t.read_idx];
            pat = rv.codes + rv.code_off[t.read_idx];
            // window start, edit_dist.cpp:112-115 (unsigned, as the reference computes it)
            g0 = t.dir_gen_flag ? t.position_in_text - cur_offset - k
                                : ref_size - t.position_in_text - m + cur_offset - k;
            if (k > (uint32_t)LD_K || fl > fr || fr > m)
                err = FLAG_ARG;
            else if ((uint64_t)g0 + m + 2ull * k + 1 > genome_symbols)
                err = FLAG_RANGE;
        }
        if (err) {
            raise_flag(flags, err);
            results[gid] = RESULT_ERROR;
            valid = false;
        }
    }

    bool alive = valid;
    // anchor must match exactly (edit_dist.cpp:121-124)
    if (alive) {
        for (uint32_t i = fl; i < fr; ++i)
            if (packed_symbol(genome, (uint64_t)g0 + i + k) != packed_symbol(pat, i)) {
                alive = false;
                break;
            }
    }
    // grow the anchor to the left (edit_dist.cpp:150-154)
    if (alive) {
        while (fl && packed_symbol(genome, (uint64_t)g0 + fl + k - 1) == packed_symbol(pat, fl - 1))
            --fl;
    }

    int ep = 0;
    {
        const bool run = alive && fl > 0;
        SymStream gs{genome, run ? (int64_t)g0 + fl + 2 * k - 1 : 0, (int64_t)genome_symbols, -1};
        SymStream ps{run ? pat : genome, run ? (int64_t)fl - 1 : 0, run ? (int64_t)m : 1, -1};
        if (__any(run)) {
            const int e = levdiag_side(gs, ps, run ? (int)fl : 0, (int)k);
            if (run) {
                ep = e;
                if (e > (int)k)
                    alive = false;
            }
        }
    }
    // grow the anchor to the right (edit_dist.cpp:248-252)
    if (alive) {
        while (fr < m && packed_symbol(genome, (uint64_t)g0 + fr + k) == packed_symbol(pat, fr))
            ++fr;
    }
    int es = 0;
    {
        const bool run = alive && fr < m;
        const int kk = (int)k - ep;   // edit_dist.cpp:256
        // genome_offset = k + fixed_right - kk - 1; the stream starts at j = 1
        SymStream gs{genome, run ? (int64_t)g0 + k + fr - kk : 0, (int64_t)genome_symbols, 1};
        SymStream ps{run ? pat : genome, run ? (int64_t)fr : 0, run ? (int64_t)m : 1, 1};
        if (__any(run)) {
            const int e = levdiag_side(gs, ps, run ? (int)(m - fr) : 0, run ? kk : 0);
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

int launch_levdiag(wamd_ctx* ctx, const wamd_reads* r, const wamd_levdiag_task* d_tasks, uint32_t n,
                   uint32_t* d_results, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    const uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(levdiag_kernel, dim3(blocks), dim3(256), 0, s, ctx->d_genome,
                       2 * ctx->genome_bytes, r->view(), d_tasks, n, ctx->ref_size, ctx->cur_offset,
                       d_results, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

} // namespace wamd
