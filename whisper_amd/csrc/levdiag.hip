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
#include <stdlib.h>

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
                                                        uint32_t* __restrict__ results, uint32_t* __restrict__ flags,
                                                        int chained, const uint32_t* __restrict__ n_dev,
                                                        const uint32_t* __restrict__ only_deferred)
{
    constexpr int GW = PW + 2;
    if (only_deferred && *only_deferred == 0)   // the banded kernel left nothing
        return;
    __shared__ uint32_t pat_lds[PW * BLOCK];   // [dword][lane]: pad, read symbols, pad
    __shared__ uint32_t gen_lds[GW * BLOCK];   // [dword][lane]: pad, window of m + 2k + 1 symbols, pad

    const uint32_t tid = threadIdx.x;
    if (n_dev)   // the batch is a compacted list whose length only the device knows (cf_filter.hip)
        n = min(n, *n_dev);
    // (a chained launch walks the batch with a small grid: almost every wave only looks at its results and moves on.
    //  No barriers below, and a lane only ever touches its own LDS column, so waves loop independently.)
    for (uint32_t base = blockIdx.x * BLOCK; base < n; base += gridDim.x * BLOCK) {
    const uint32_t gid = base + tid;
    bool valid = gid < n;
    if (chained) {   // only what the banded kernel left (RESULT_DEFERRED)
        if (valid && results[gid] != 0xFFFFFFFEu)
            valid = false;
        if (!__any(valid))
            continue;
    }
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
}

// =============================================================================================
// Banded bit-vector form (reads up to 256 symbols): the kernel that normally runs.
//
// Each side of LevDiag is the edit-distance matrix of the remaining pattern symbols (rows) against the
// text walking away from the anchor (columns), started in the corner (D[i][0] = i, D[0][j] = j), free
// in the text's end, thresholded at the budget - see the header of this file.  Cells within k of the
// main diagonal decide the answer, so a frame of 32 rows slides down the diagonal one row per column
// (Hyyro's band form of Myers' recurrence: D0 shifts right where the column form shifts HP / HN left):
// bit b of column j is row j - k + b, cells outside count as infinity, values <= k are exact.  Rows
// above the first pattern symbol are phantom rows carrying D = j + |i| (no matches, vertical delta -1),
// which reproduces the corner start without a special case; rows below the last symbol are never read
// by a real cell.  The score that is carried along is the band's lower edge D[j + k][j] (bit 2k of D0):
// where that edge meets the last row the walk along the last row starts, by the horizontal deltas of
// its bit - the side's answer is the minimum over the last row within k columns of the corner.
//
// What one column costs.  The read's pattern vectors are kept in LDS as three bit planes (code bit 0,
// code bit 1, "is A/C/G/T"), [dword][plane][lane]; a group of eight columns reads the three dwords per
// plane its frames overlap and funnels them to one 64-bit window, a column takes its frame from that
// window with one v_alignbit per plane (constant shift) and turns the text symbol's two code bits into
// the match vector with two sign-extending bit-field extracts and two bit operations.  The text comes
// one funnelled dword per eight columns straight from the genome, prefetched a group ahead.  Then nine
// bit operations of recurrence and two of score.  No load address depends on an earlier result - the
// run-extension form above waits for LDS at every step.  Groups come in four builds: with or without
// the last-row walk (only the groups around the corner need it), and with or without the checks for
// symbols >= 4 (phantom rows of the first group, a text symbol >= 4 in the group, a read of the wave
// holding one) - the common group is the bare one.
// The planes only describe A, C, G, T: a read symbol >= 4 facing the same code in the genome would
// match in the reference.  Lanes whose text shows a code >= 4 that their read holds too are handed to
// the run-extension kernel (RESULT_DEFERRED) - never with Whisper's own codes, where a read's N is 4
// and the reference's is 5 - as are batches with reads above 256 symbols.
// =============================================================================================
constexpr uint32_t LD_DEFERRED = 0xFFFFFFFEu;
constexpr int LDB_PAD = 2;   // zero dwords under and over the planes: a group's window may hang over either end

__device__ __forceinline__ uint32_t ld_fetch8(const GenomeView& g, int64_t s)
{
    const int64_t d = s >> 3;
    const uint32_t sh = ((uint32_t)s & 7u) * 4u;
    const uint32_t i0 = d < 0 ? 0u : (uint32_t)d, i1 = d + 1 < 0 ? 0u : (uint32_t)(d + 1);
    return __builtin_amdgcn_alignbit(load_genome_dword(g, i1), load_genome_dword(g, i0), sh);
}

__device__ __forceinline__ uint32_t ldb_dword(const GenomeView& g, int32_t idx)
{
    return load_genome_dword(g, idx < 0 ? 0u : (uint32_t)idx);
}

// Four consecutive raw genome dwords from idx on (any idx: clamped into the genome one by one at either end)
__device__ __forceinline__ void ldb_load4(const GenomeView& g, int32_t idx, uint32_t (&v)[4])
{
    if (idx >= 0 && (uint32_t)idx + 3u <= g.last_dword) {
        const wamd_u32x4_a4 x = *reinterpret_cast<const wamd_u32x4_a4*>(g.dwords + idx);
        v[0] = x.x;
        v[1] = x.y;
        v[2] = x.z;
        v[3] = x.w;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int32_t i = idx + c;
            const uint32_t u = i < 0 ? 0u : (uint32_t)i;
            v[c] = g.dwords[u < g.last_dword ? u : g.last_dword];
        }
    }
}

// The text of a side as dwords in walking order: D[i] = dword dw0 + i for DIR > 0, dword dw0 + 1 - i for DIR < 0
// (group g takes its eight symbols from D[g] and D[g + 1]).  out[j] = D[i + j], raw.
template <int DIR>
__device__ __forceinline__ void ldb_walk4(const GenomeView& g, int32_t dw0, int i, uint32_t (&out)[4])
{
    if (DIR > 0) {
        ldb_load4(g, dw0 + i, out);
    } else {
        uint32_t v[4];
        ldb_load4(g, dw0 + 1 - i - 3, v);
        out[0] = v[3];
        out[1] = v[2];
        out[2] = v[1];
        out[3] = v[0];
    }
}

struct LdbState {
    uint32_t VP, VN;
    int S;      // columns whose lower-edge diagonal step was free: D[j + k][j] = k + j - S
    int rowR;   // D[R][j] once the walk along the last row has started
    int best;
};

// Eight columns j0 + 1 .. j0 + 8 of one side.  t: the lane's planes at the dword holding table bit Lg, the
// lowest bit any frame of the group touches; w: the eight text symbols (column j0 + 1 in nibble 0 for
// DIR > 0, in nibble 7 for DIR < 0).
struct LdbWin {   // the three dwords per plane a group's frames overlap
    uint32_t a0, a1, a2, b0, b1, b2;
};
template <int BLOCK>
__device__ __forceinline__ LdbWin ldb_window(const uint32_t* t)
{
    LdbWin x;
    x.a0 = t[0];
    x.a1 = t[3 * BLOCK];
    x.a2 = t[6 * BLOCK];
    x.b0 = t[BLOCK];
    x.b1 = t[4 * BLOCK];
    x.b2 = t[7 * BLOCK];
    return x;
}

template <int BLOCK, int DIR, bool SLOW, bool END>
__device__ __forceinline__ void ldb_group(const LdbWin& x, const uint32_t* t, uint32_t sh, uint32_t w, int j0, int k, int R,
                                          LdbState& st)
{
    const uint32_t a0 = x.a0, a1 = x.a1, a2 = x.a2, b0 = x.b0, b1 = x.b1, b2 = x.b2;
    uint32_t A0 = __builtin_amdgcn_alignbit(a1, a0, sh), A1 = __builtin_amdgcn_alignbit(a2, a1, sh);
    uint32_t B0 = __builtin_amdgcn_alignbit(b1, b0, sh), B1 = __builtin_amdgcn_alignbit(b2, b1, sh);
    // (the four builds of a group share their first instructions per column; left alone, the compiler computes all of
    //  them ahead of the branch that picks the build and keeps 30 registers alive for it)
    asm volatile("" : "+v"(w), "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1));
    uint32_t V0 = 0, V1 = 0;
    if (SLOW) {
        const uint32_t v0 = t[2 * BLOCK], v1 = t[5 * BLOCK], v2 = t[8 * BLOCK];
        V0 = __builtin_amdgcn_alignbit(v1, v0, sh);
        V1 = __builtin_amdgcn_alignbit(v2, v1, sh);
    }
    uint32_t VP = st.VP, VN = st.VN;
    int S = st.S;
    const uint32_t k2 = 2u * (uint32_t)k;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int nib = DIR > 0 ? u : 7 - u;   // also the frame's offset in the window
        const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)w, 4 * nib, 1);
        const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)w, 4 * nib + 1, 1);
        const uint32_t a = nib ? __builtin_amdgcn_alignbit(A1, A0, (uint32_t)nib) : A0;
        const uint32_t b = nib ? __builtin_amdgcn_alignbit(B1, B0, (uint32_t)nib) : B0;
        uint32_t eq = ~(a ^ m0) & ~(b ^ m1);
        if (SLOW) {
            eq &= nib ? __builtin_amdgcn_alignbit(V1, V0, (uint32_t)nib) : V0;
            if ((w >> (4 * nib)) & 0xCu)
                eq = 0u;
        }
        if (DIR < 0)
            eq = __brev(eq);
        if (SLOW) {   // phantom rows (row <= 0) still inside the frame: no matches there
            const int ph = k - j0 - u;
            eq &= ph > 0 ? ~0u << ph : ~0u;
        }
        // Hyyro's band recurrence, one dword (slide_core of myers_distance.hip)
        const uint32_t x = (((eq & VP) + VP) ^ VP) | eq;
        const uint32_t d0 = x | VN;
        const uint32_t hp = VN | ~(d0 | VP);
        const uint32_t hn = d0 & VP;
        const uint32_t d0s = d0 >> 1;
        VP = hn | ~(d0s | hp);
        VN = d0s & hp;
        S += (int)((d0 >> k2) & 1u);
        if (END) {
            const int j = j0 + u + 1;
            const uint32_t bit = (uint32_t)(R - j + k);   // the last row's bit in this column's frame
            if (bit < k2)
                st.rowR += (int)((hp >> bit) & 1u) - (int)((hn >> bit) & 1u);
            if (bit == k2)
                st.rowR = k + j - S;
            if (bit <= k2)
                st.best = st.rowR < st.best ? st.rowR : st.best;
        }
    }
    st.VP = VP;
    st.VN = VN;
    st.S = S;
}

// One side for every lane of the wave.  pm: this lane's column of the plane table; edge_bit: table bit of
// the pattern symbol next to the anchor's edge (row 1 is edge_bit for DIR > 0, edge_bit - 1 for DIR < 0);
// gsym1: genome symbol of column 1 (DIR > 0 walks right, DIR < 0 left); R rows; band half-width k; budget
// kk <= k.  Returns the side's distance, kk + 1 when above kk; `odd` collects the codes >= 4 the text showed.
template <int BLOCK, int DIR, bool PF>
__device__ int ld_band_side(const GenomeView& g, const uint32_t* pm, int edge_bit, int64_t gsym1, int R, int k, int kk,
                            bool run, bool careful, uint32_t& odd)
{
    LdbState st;
    st.VP = ~0u << k;
    st.VN = ~st.VP;
    st.S = 0;
    st.rowR = R;   // D[R][0]: where the walk starts when the last row is inside the band from the corner on
    st.best = R;
    bool dead = false;
    const int cols = run ? R + k : 0;
    if (!run)
        edge_bit = 32 * LDB_PAD;   // lanes sitting this side out read (and ignore) the table's first dwords
    const int jlast = cols > 0 ? (cols - 1) & ~7 : 0;
    // text: 16-byte loads (what a wave's load costs is the 64 lines it touches), one of them ahead
    const int64_t s0 = DIR > 0 ? gsym1 : gsym1 - 7;
    const int32_t dw0 = (int32_t)(s0 >> 3);
    const uint32_t tsh = ((uint32_t)s0 & 7u) * 4u;
    uint32_t q0, nx[4];
    q0 = ldb_dword(g, DIR > 0 ? dw0 : dw0 + 1);
    ldb_walk4<DIR>(g, dw0, 1, nx);
    // the lowest table bit a frame of group j0 touches (lanes past their last group stay on it)
    auto window_bit = [&](int j0) {
        const int jc = j0 < jlast ? j0 : jlast;
        return DIR > 0 ? edge_bit + jc - k : edge_bit - jc + k - 39;
    };
    int Lg = window_bit(0);
    LdbWin win = ldb_window<BLOCK>(pm + ((uint32_t)Lg >> 5) * (3u * BLOCK));
    bool more = true;
    for (int jc0 = 0; more; jc0 += 32) {   // 32 columns = four text dwords per pass
        // q0 = D[jc0 / 8]; the four behind it have landed: ask for the four after them
        uint32_t q1 = nibble_swap(nx[0]), q2 = nibble_swap(nx[1]), q3 = nibble_swap(nx[2]), q4 = nibble_swap(nx[3]);
        ldb_walk4<DIR>(g, dw0, (jc0 >> 3) + 5, nx);
#pragma unroll 1
        for (int j0 = jc0; j0 < jc0 + 32; j0 += 8) {
            const bool active = !dead && j0 < cols;
            if (!__any(active)) {
                more = false;
                break;
            }
            const uint32_t w = DIR > 0 ? __builtin_amdgcn_alignbit(q1, q0, tsh) : __builtin_amdgcn_alignbit(q0, q1, tsh);
            q0 = q1;
            q1 = q2;
            q2 = q3;
            q3 = q4;
            const bool nz = active && (w & 0xCCCCCCCCu) != 0u;
            const bool any_nz = __any(nz);
            if (any_nz && nz) {   // which codes >= 4 the text shows this lane (bit c for code c, as in the read's meta word)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t c = (w >> (4 * u)) & 15u;
                    odd |= c >= 4u ? 1u << c : 0u;
                }
            }
            const bool slow = careful || any_nz;
            const bool end = __any(active && j0 + 8 + k >= R);
            const uint32_t* t = pm + ((uint32_t)Lg >> 5) * (3u * BLOCK);
            const uint32_t sh = (uint32_t)Lg & 31u;
            const LdbWin cur = PF ? win : ldb_window<BLOCK>(t);
            Lg = window_bit(j0 + 8);
            if (PF)
                win = ldb_window<BLOCK>(pm + ((uint32_t)Lg >> 5) * (3u * BLOCK));   // the next group's, in flight while this one computes
            if (!slow) {
                if (!end)
                    ldb_group<BLOCK, DIR, false, false>(cur, t, sh, w, j0, k, R, st);
                else
                    ldb_group<BLOCK, DIR, false, true>(cur, t, sh, w, j0, k, R, st);
            } else {
                if (!end)
                    ldb_group<BLOCK, DIR, true, false>(cur, t, sh, w, j0, k, R, st);
                else
                    ldb_group<BLOCK, DIR, true, true>(cur, t, sh, w, j0, k, R, st);
            }
            // the lower edge of column j0 + 8 minus 2k bounds every cell of it, hence of all later ones, from below
            // (only while no cell of the last row has been seen: those of earlier columns would still count)
            if (j0 + 8 + k < R && (j0 + 8 - st.S) - k > kk)
                dead = true;
        }
    }
    if (!run)
        return 0;
    return (!dead && st.best <= kk) ? st.best : kk + 1;
}

// PACK: between the two sides the block closes ranks.  A candidate's prefix side either ends it (more than k errors:
// most unrelated loci, whose answer is then final) or leaves the suffix side to do; the lanes with a suffix to run move
// into the block's first lanes - three words each (task, the prefix's errors, the odd codes seen) through the planes'
// LDS, the task itself is read again, the new lanes stage their reads' planes - and waves left without a lane leave.
// Three barriers, outside wave-divergent code (myers_band_kernel of myers_distance.hip does the same mid-band).
template <int NW, int BLOCK, bool PF, bool PACK>
__device__ __forceinline__ void levdiag_band_body(const GenomeView& g, const ReadsView& rv,
                                                             const wamd_levdiag_task* __restrict__ tasks, uint32_t n,
                                                             uint32_t ref_size, uint32_t cur_offset,
                                                             uint32_t* __restrict__ results, uint32_t* __restrict__ flags, uint32_t* pm_lds,
                                                             uint32_t* __restrict__ deferred)
{
    constexpr int ROWS = NW + 2 * LDB_PAD;

    const uint32_t tid = threadIdx.x;
    uint32_t gid = blockIdx.x * BLOCK + tid;
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
        } else if (rv.slot_direct < 0) {
            err = FLAG_ORIENT;
        } else {
            m = rv.lens[read_idx];
            g0 = t.dir_gen_flag ? t.position_in_text - cur_offset - k
                                : ref_size - t.position_in_text - m + cur_offset - k;   // edit_dist.cpp:112-115
            if (k > (uint32_t)LD_K || fl > fr || fr > m || m > 32u * NW)
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
    if (!valid)
        k = fl = fr = m = g0 = read_idx = 0;

    // ---- the read's pattern vectors -> three bit planes in LDS ---------------------------------------
    uint32_t* pm = pm_lds + tid;
    uint32_t read_odd = 0;   // bit c: the read holds code c >= 4
    auto stage_planes = [&]() {
        read_odd = 0;
        const uint32_t slot = (uint32_t)(rv.slot_direct < 0 ? 0 : rv.slot_direct);
        const uint4* src = reinterpret_cast<const uint4*>(rv.pm32) + ((uint64_t)read_idx * rv.n_orient + slot) * rv.pm32_stride;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);   // rows A, C, G, T of dword j - LDB_PAD
            if (valid && j >= LDB_PAD && (uint32_t)(j - LDB_PAD) < rv.words32)
                v = src[j - LDB_PAD];
            pm[(j * 3 + 0) * BLOCK] = v.y | v.w;
            pm[(j * 3 + 1) * BLOCK] = v.z | v.w;
            pm[(j * 3 + 2) * BLOCK] = (v.x | v.y) | (v.z | v.w);
        }
        if (valid)
            read_odd = rv.meta[(uint64_t)read_idx * rv.n_orient + slot] & 0xfff0u;
    };
    stage_planes();
    // The bare groups leave out two things.  One is the "is A/C/G/T" plane: fine while no read of the wave holds
    // another code.  The other is the masking of the phantom rows in a side's first columns: with k or more anchor
    // symbols next to the side the frame's rows above row 1 are those symbols, facing the text they equal, and the
    // matrix of the strings so extended has the side's values in the side's cells (a common prefix does not change
    // an edit distance) - and the same start vectors, k matching columns down the diagonal from its own corner.
    bool careful = __any(read_odd != 0u || (valid && fr - fl < k));
    // table bit of pattern symbol i: the vector is top-aligned in dwords LDB_PAD .. LDB_PAD + words32 - 1
    int bitbase = 32 * LDB_PAD + 32 * (int)rv.words32 - (int)m;

    bool alive = valid;
    uint32_t odd = 0;
    // anchor must match exactly (edit_dist.cpp:121-124): packed symbols, 32 per step
    {
        const uint32_t* cd = reinterpret_cast<const uint32_t*>(rv.codes + rv.code_off[read_idx]);
        uint32_t i = fl;
        while (__any(alive && i < fr)) {
            if (alive && i < fr) {
                // (the codes buffer has 64 bytes of slack behind the last read)
                const wamd_u32x4_a4 pv = *reinterpret_cast<const wamd_u32x4_a4*>(cd + (i >> 3));
                const uint32_t pr[5] = {pv.x, pv.y, pv.z, pv.w, cd[(i >> 3) + 4]};
                const uint64_t gs = (uint64_t)g0 + k + i;
                uint32_t gr[5], g4[4];
                ldb_load4(g, (int32_t)(gs >> 3), g4);
                gr[0] = g4[0];
                gr[1] = g4[1];
                gr[2] = g4[2];
                gr[3] = g4[3];
                gr[4] = g.dwords[(uint32_t)(gs >> 3) + 4u < g.last_dword ? (uint32_t)(gs >> 3) + 4u : g.last_dword];
                const uint32_t psh = (i & 7u) * 4u, gsh = ((uint32_t)gs & 7u) * 4u;
                uint32_t diff = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t p8 = __builtin_amdgcn_alignbit(nibble_swap(pr[c + 1]), nibble_swap(pr[c]), psh);
                    const uint32_t g8 = __builtin_amdgcn_alignbit(nibble_swap(gr[c + 1]), nibble_swap(gr[c]), gsh);
                    const int left = (int)(fr - i) - 8 * c;   // symbols of the anchor from this dword on
                    uint32_t x = p8 ^ g8;
                    if (left < 8)
                        x = left > 0 ? x & ((1u << (4 * left)) - 1u) : 0u;
                    diff |= x;
                }
                if (diff)
                    alive = false;
                i += 32;
            }
        }
    }
    // prefix: rows fl - 1, fl - 2, ... against the text walking left from g0 + fl + k - 1
    int ep = 0;
    {
        const bool run = alive && fl > 0;
        if (__any(run)) {
            const int e = ld_band_side<BLOCK, -1, PF>(g, pm, bitbase + (int)fl, (int64_t)g0 + fl + k - 1, run ? (int)fl : 0, (int)k,
                                                  (int)k, run, careful, odd);
            if (run) {
                ep = e;
                if (e > (int)k)
                    alive = false;
            }
        }
    }
    if (PACK) {
        // ---- who is through, who has a suffix side to run -------------------------------------------------
        constexpr int WAVES = BLOCK / 64;
        __shared__ uint32_t sh_cnt[WAVES];
        const bool more = alive && fr < m;
        if (valid && !more) {   // the answer is final: the prefix ended the candidate, or there is no suffix
            const bool dfr = (odd & read_odd) != 0u;
            results[gid] = dfr ? LD_DEFERRED : (alive ? (uint32_t)ep : k + 1);
            if (deferred && dfr && *reinterpret_cast<volatile uint32_t*>(deferred) == 0)
                *reinterpret_cast<volatile uint32_t*>(deferred) = 1u;
            valid = false;
        }
        alive = valid;
        const uint64_t live = __ballot(valid);
        if ((tid & 63u) == 0)
            sh_cnt[tid >> 6] = (uint32_t)__popcll(live);
        __syncthreads();   // (every wave is also through with its planes)
        uint32_t total = 0, before = 0, waves_now = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = sh_cnt[w];
            before += (uint32_t)w < (tid >> 6) ? c : 0u;
            total += c;
            waves_now += c ? 1u : 0u;
        }
        if ((total + 63u) / 64u < waves_now) {   // block-uniform: the move frees at least one wave
            if (valid) {
                const uint32_t p = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0));
                pm_lds[p] = gid;
                pm_lds[BLOCK + p] = (uint32_t)ep;
                pm_lds[2 * BLOCK + p] = odd;
            }
            __syncthreads();
            valid = tid < total;
            if (valid) {
                gid = pm_lds[tid];
                ep = (int)pm_lds[BLOCK + tid];
                odd = pm_lds[2 * BLOCK + tid];
                const wamd_levdiag_task t = tasks[gid];   // (held against the batch when it was first decoded)
                k = t.max_mismatches;
                fl = t.fixed_left;
                fr = t.fixed_right;
                read_idx = t.read_idx;
                m = rv.lens[read_idx];
                g0 = t.dir_gen_flag ? t.position_in_text - cur_offset - k : ref_size - t.position_in_text - m + cur_offset - k;
            } else {
                k = fl = fr = m = g0 = read_idx = 0;
                ep = 0;
                odd = 0;
            }
            alive = valid;
            __syncthreads();   // everybody has its new task: the planes may go over the three words
            if (__any(valid)) {
                stage_planes();
                careful = __any(read_odd != 0u || (valid && fr - fl < k));
                bitbase = 32 * LDB_PAD + 32 * (int)rv.words32 - (int)m;
            }
        }
    }
    // suffix: rows fr, fr + 1, ... against the text walking right from g0 + fr + k, budget k - ep (edit_dist.cpp:256)
    int es = 0;
    {
        const bool run = alive && fr < m;
        const int kk = (int)k - ep;
        if (__any(run)) {
            const int e = ld_band_side<BLOCK, 1, PF>(g, pm, bitbase + (int)fr, (int64_t)g0 + fr + k, run ? (int)(m - fr) : 0, (int)k,
                                                 run ? kk : 0, run, careful, odd);
            if (run) {
                es = e;
                if (e > kk)
                    alive = false;
            }
        }
    }
    if (valid)
        results[gid] = (odd & read_odd) ? LD_DEFERRED : (alive ? (uint32_t)(ep + es) : k + 1);
    // the run-extension kernel behind this one only asks "any?": one word, written by the waves that still read 0
    if (deferred && __any(valid && (odd & read_odd)) && (tid & 63u) == 0 && *reinterpret_cast<volatile uint32_t*>(deferred) == 0)
        *reinterpret_cast<volatile uint32_t*>(deferred) = 1u;
}

// 90 VGPRs: five waves per SIMD, which is also what the planes' LDS allows at 160 symbols (four at 256)
template <int NW, int BLOCK>
__global__ __launch_bounds__(BLOCK) void levdiag_band_kernel(GenomeView g, ReadsView rv,
                                                             const wamd_levdiag_task* __restrict__ tasks, uint32_t n,
                                                             uint32_t ref_size, uint32_t cur_offset,
                                                             uint32_t* __restrict__ results, uint32_t* __restrict__ flags,
                                                             const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ deferred)
{
    __shared__ uint32_t pm_lds[(NW + 2 * LDB_PAD) * 3 * BLOCK];
    if (n_dev) {
        n = min(n, *n_dev);
        if (blockIdx.x * BLOCK >= n)
            return;
    }
    levdiag_band_body<NW, BLOCK, true, (BLOCK > 128)>(g, rv, tasks, n, ref_size, cur_offset, results, flags, pm_lds, deferred);
}
template <int NW, int BLOCK>
static int launch_ld_band(wamd_ctx* ctx, const wamd_reads* r, const wamd_levdiag_task* d_tasks, uint32_t n,
                          uint32_t* d_results, hipStream_t s, const uint32_t* n_dev, uint32_t* deferred)
{
    const uint32_t blocks = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL((levdiag_band_kernel<NW, BLOCK>), dim3(blocks), dim3(BLOCK), 0, s, genome_view(ctx), r->view(),
                       d_tasks, n, ctx->ref_size, ctx->cur_offset, d_results, ctx->d_flags, n_dev, deferred);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

template <int PW, int BLOCK>
static int launch_ld(wamd_ctx* ctx, const wamd_reads* r, const wamd_levdiag_task* d_tasks, uint32_t n,
                     uint32_t* d_results, hipStream_t s, const uint32_t* n_dev, int chained = 0,
                     const uint32_t* only_deferred = nullptr)
{
    uint32_t blocks = (n + BLOCK - 1) / BLOCK;
    if (chained && blocks > 1024)
        blocks = 1024;
    hipLaunchKernelGGL((levdiag_kernel<PW, BLOCK>), dim3(blocks), dim3(BLOCK), 0, s, genome_view(ctx), r->view(),
                       d_tasks, n, ctx->ref_size, ctx->cur_offset, d_results, ctx->d_flags, chained, n_dev, only_deferred);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_levdiag(wamd_ctx* ctx, const wamd_reads* r, const wamd_levdiag_task* d_tasks, uint32_t n,
                   uint32_t* d_results, hipStream_t s, const uint32_t* n_dev)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0)
        return WAMD_OK;
    // staged dwords per lane: read symbols / 8 plus a pad dword at either end
    const uint32_t need = (r->max_len + 7) / 8 + 2;
    // reads up to 256 symbols, uploaded in the direct orientation: the banded bit-vector kernel, then the
    // run-extension kernel for the lanes it deferred (none in practice: it leaves at once on the band kernel's word)
    const bool band = (r->orientations & 1) && r->max_len <= 256 && !(ctx->test_hooks & HOOK_LEVDIAG_RUNS);
    uint32_t* deferred = nullptr;   // (the word is the stream's first of myers_distance.hip's pair: launches on a stream are ordered)
    if (band) {
        deferred = band_flags(ctx, s);
        if (deferred)
            WAMD_HIP(hipMemsetAsync(deferred, 0, sizeof(uint32_t), s));
#ifndef WAMD_LD_BLOCK
#define WAMD_LD_BLOCK 256   // blocks of four waves pack their survivors between the sides; 128: the kernel without it
#endif
        int rc = r->words32 <= 5 ? launch_ld_band<5, WAMD_LD_BLOCK>(ctx, r, d_tasks, n, d_results, s, n_dev, deferred)
                                 : launch_ld_band<8, WAMD_LD_BLOCK>(ctx, r, d_tasks, n, d_results, s, n_dev, deferred);
        if (rc)
            return rc;
    }
    if (need <= 22)
        return launch_ld<22, 256>(ctx, r, d_tasks, n, d_results, s, n_dev, band, deferred);   // reads up to 160 symbols
    if (need <= 34)
        return launch_ld<34, 128>(ctx, r, d_tasks, n, d_results, s, n_dev, band, deferred);   // up to 256
    if (need <= 66)
        return launch_ld<66, 64>(ctx, r, d_tasks, n, d_results, s, n_dev);    // up to 512
    return launch_ld<130, 64>(ctx, r, d_tasks, n, d_results, s, n_dev);       // up to 1024
}

} // namespace wamd
