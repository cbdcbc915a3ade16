// myers_align.hip - verification WITH alignment (post-processing half of the path).
//
// Replaces dynamicProgramming + getExtCigar (+ refine_ext_cigar) as called from
// fill_cigar_with_lev and find_pairs<LevMyers> (mapper/sam_matching.cpp:689-752,
// 401-426): end position with the variant's tie-break, begin position, ext-CIGAR,
// affine score, number of gap opens, optional short-indel refinement.
//
// The reference keeps the whole bit-vector matrix of the forward pass (5 vectors x
// words x columns: 28 KB for 150 bp x 175 columns, over 100 KB for a mate-rescue
// window) and walks back through it.  Spilling that per lane would make the kernel
// HBM-bound, so the work is split:
//
//   pass 1  the distance kernel (myers_distance.hip): edit distance and end column,
//           nothing stored.  Most post-processing calls stop here (not ok).
//   pass 2  accepted tasks only, compacted into lists by how wide a band the
//           traceback can touch.  The forward recurrence is re-run over just the
//           last m + ed columns before the end column - cells on an optimal path
//           to (m, pos), and their neighbours whenever those can be predecessors,
//           have optimal paths that start inside that window, so their values are
//           the true ones - and per column only a window of 2ed+3 rows around the
//           end diagonal is kept, of VP, VN and D0 (24 B per column instead of
//           40 B x words).  HP/HN at a cell follow from VP/VN of the column to the
//           left.  For bands of up to 64 rows (every practical case) the forward pass
//           stores what the walk decides on instead: "diagonal with match", "diagonal
//           with mismatch", "horizontal", "vertical" as four windows (16 B per column
//           for bands of up to 32 rows), so the walk neither fetches symbols nor
//           tracks the cell value.  The traceback records 2 bits per step; a forward replay writes
//           the ext-CIGAR in its final order and spots the refinement pattern on
//           the fly.  The affine score is summed during the backward walk in the
//           reference's order (double adds only), so it is bit-identical.
#include "wamd_internal.h"

namespace wamd {

enum { OP_NONE = 0, OP_M = 1, OP_X = 2, OP_D = 3, OP_I = 4 };
enum { DIR_D = 0, DIR_H = 1, DIR_V = 2, DIR_X = 3 };   // DIR_X: a diagonal step over a mismatch, recorded by the split
                                                        // class only (its replay then needs no symbols for matches)

// reader of 4-bit symbols, either direction, over dwords staged in LDS as [dword][lane]
// (already nibble-swapped: symbol s of a dword at bits 4s); dword 0 holds symbols
// 8 * origin_dw .. 8 * origin_dw + 7 of the sequence the indices refer to
struct NibStream {
    const uint32_t* base;
    int64_t idx;
    int64_t origin_dw;
    int32_t n_dw;
    uint32_t stride;
    uint32_t cur, nxt;
    int64_t cur_dw, nxt_dw;
    bool packed;   // dwords as they lie in memory (two symbols per byte, even one high): swap on load
    __device__ __forceinline__ uint32_t fetch(int64_t dw) const
    {
        int64_t d = dw - origin_dw;
        d = d < 0 ? 0 : (d >= n_dw ? n_dw - 1 : d);
        return base[d * stride];
    }
    __device__ __forceinline__ void load()
    {
        cur_dw = idx >> 3;
        const uint32_t v = fetch(cur_dw);
        cur = packed ? nibble_swap(v) : v;
    }
    __device__ __forceinline__ void init(const uint32_t* b, uint32_t stride_, int64_t origin, int32_t n, int64_t start,
                                         bool packed_ = false)
    {
        base = b;
        stride = stride_;
        packed = packed_;
        origin_dw = origin;
        n_dw = n;
        idx = start;
        nxt_dw = INT64_MIN;
        nxt = 0;
        load();
    }
    __device__ __forceinline__ uint32_t get(int step)
    {
        if ((idx >> 3) != cur_dw) {
            if (packed && (idx >> 3) == nxt_dw) {   // the dword asked for eight symbols ago
                cur = nibble_swap(nxt);
                cur_dw = nxt_dw;
            } else {
                load();
            }
            if (packed) {   // straight from memory: the following dword is requested now, used eight symbols on
                nxt_dw = cur_dw + (step > 0 ? 1 : -1);
                nxt = fetch(nxt_dw);
            }
        }
        const uint32_t s = (cur >> (4 * (uint32_t)(idx & 7))) & 15u;
        idx += step;
        return s;
    }
};

// ---- pass 2a: sort accepted tasks into classes by band width ------------------------
// class 0: 2ed+3 <= 32 rows, class 1: <= 64 rows, class 2: <= 128 rows, class 3: whole columns
__global__ __launch_bounds__(1024) void align_classify_kernel(
    ReadsView rv, const wamd_dist_task* __restrict__ tasks, const wamd_dist_result* __restrict__ dist,
    uint32_t n, wamd_align_result* __restrict__ results, uint8_t* __restrict__ cigars,
    uint32_t cigar_stride, uint32_t* __restrict__ lists, uint32_t* __restrict__ counters)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cls = 4;   // 4: not accepted, no traceback
    if (gid < n) {
        const wamd_dist_result d = dist[gid];
        wamd_align_result r;
        r.affine_score = 0;
        r.ok = 0;
        r.edit_distance = d.edit_distance;
        r.end_pos = d.pos;
        r.begin_pos = 0;
        r.num_events = 0;
        r.cigar_len = 0;
        r.refined = 0;
        r.reserved = 0;
        const bool error = d.edit_distance == RESULT_ERROR;
        const bool ok = !error && d.edit_distance <= tasks[gid].max_ed;
        if (!ok) {
            cigars[(uint64_t)gid * cigar_stride] = 0;   // accepted tasks get theirs from the traceback
            if (error)
                r.end_pos = 0;
            results[gid] = r;
        } else {
            const uint32_t rows = 2 * d.edit_distance + 3;
            cls = rows <= 32 ? 0u : (rows <= 64 ? 1u : (rows <= 128 ? 2u : 3u));
        }
    }
    // one atomic per block and class (same-address atomics serialise: per wave they were most of
    // this kernel's time); inside the block slots go by wave, inside the wave by lane rank
    __shared__ uint32_t wave_cnt[16][4];
    __shared__ uint32_t block_base[4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        const uint64_t mask = __ballot(cls == c);
        if (cls == c)
            rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
        if (lane == 0)
            wave_cnt[wave][c] = (uint32_t)__popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
            const uint32_t v = wave_cnt[w][threadIdx.x];
            wave_cnt[w][threadIdx.x] = total;   // exclusive prefix over the block's waves
            total += v;
        }
        block_base[threadIdx.x] = total ? atomicAdd(counters + threadIdx.x, total) : 0u;
    }
    __syncthreads();
    if (cls < 4)
        lists[(uint64_t)cls * n + block_base[cls] + wave_cnt[wave][cls] + rank] = gid;
}

template <int NW, int WW>
__device__ __forceinline__ void extract_window(const uint64_t (&V)[NW], uint32_t off, uint64_t (&out)[WW])
{
    if constexpr (WW >= NW) {
#pragma unroll
        for (int w = 0; w < WW; ++w)
            out[w] = w < NW ? V[w] : 0;
    } else {
        const uint32_t wq = off >> 6, sh = off & 63u;
        uint64_t S[WW + 1];
#pragma unroll
        for (int t = 0; t <= WW; ++t) {
            S[t] = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                S[t] = (wq + t == (uint32_t)w) ? V[w] : S[t];
        }
#pragma unroll
        for (int t = 0; t < WW; ++t)
            out[t] = sh ? ((S[t] >> sh) | (S[t + 1] << (64u - sh))) : S[t];
    }
}

// 32 rows of V starting at row bit off
template <int NW>
__device__ __forceinline__ uint32_t extract32(const uint64_t (&V)[NW], uint32_t off)
{
    const uint32_t wq = off >> 6, sh = off & 63u;
    uint64_t s0 = 0, s1 = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        s0 = (wq == (uint32_t)w) ? V[w] : s0;
        s1 = (wq + 1 == (uint32_t)w) ? V[w] : s1;
    }
    return (uint32_t)(sh ? ((s0 >> sh) | (s1 << (64u - sh))) : s0);
}

// Which way getExtCigar leaves a cell (LevMyers64.cpp:431-442): diagonal before horizontal before vertical, but a
// move that continues the previous direction wins.  Entry (prev_dir, from_d, from_h, from_v) -> DIR_*, two bits each.
constexpr uint64_t walk_table()
{
    uint64_t t = 0;
    for (uint32_t prev = 0; prev < 3; ++prev)
        for (uint32_t i = 0; i < 8; ++i) {
            const bool fd = i & 4u, fh = i & 2u, fv = i & 1u;
            const bool take_d = fd && (prev == DIR_D || (prev == DIR_H && !fh) || (prev == DIR_V && !fv));
            const bool take_h = !take_d && fh && (prev == DIR_H || (prev == DIR_D && !fd) || (prev == DIR_V && !fv));
            const uint64_t dir = take_d ? DIR_D : (take_h ? DIR_H : DIR_V);
            t |= dir << (2u * (prev * 8u + i));
        }
    return t;
}
constexpr uint64_t WALK_TABLE = walk_table();

struct AlignParams {
    const wamd_dist_task* tasks;
    const wamd_dist_result* dist;
    const uint32_t* list;
    const uint32_t* count;
    wamd_align_result* results;
    uint8_t* cigars;
    uint64_t* slots;          // per block: [col][vec][ww][tid] then path dwords [q][tid]
    uint64_t slot_words;      // 64-bit words per block
    uint32_t* paths;          // split classes: [group of 64 tasks][16 steps][lane]
    uint32_t* flags;
    wamd_scoring sc;
    uint32_t cigar_stride;
    uint32_t max_cols;
    uint32_t path_dwords;
    int refine;
    int ascii;
    uint32_t rd_cap, gd_cap;   // replay kernel: staged read / genome dwords per lane (from the batch's longest read)
    uint32_t rp_sw;            // staged replay: dwords of a lane's string row in LDS
};

// ---- pass 2b: windowed recompute + traceback + forward emission --------------------
// PHASE 0: everything in one kernel.  The banded class is split in two so that each half only
// pays for its own LDS: PHASE 1 = forward pass + walk (path and totals go to memory),
// PHASE 2 = forward replay of the recorded path (ext-CIGAR, refinement).
template <int NW, int WW, int BLOCK, int PHASE>
#ifndef WAMD_WALK_WAVES
#define WAMD_WALK_WAVES 4
#endif
// (second launch bound = waves per SIMD the register budget must allow: the walk kernel sat at 133
// VGPRs, one wave per SIMD short of what its LDS permits)
__global__ __launch_bounds__(BLOCK, (PHASE == 1 && NW <= 4 ? (WW == -1 ? 2 : (WW == -2 ? 1 : WAMD_WALK_WAVES)) : 1)) void align_traceback_kernel(GenomeView g, ReadsView rv, AlignParams ap)
{
    // WW = 0 / -1 / -2: banded (nothing of the matrix is stored; a band of 64 / 128 / 256 rows, the walk's windows the
    // 32 / 64 / 128 rows in its middle); otherwise WW 64-bit words per stored window (at most NW)
    constexpr bool HALF = WW <= 0;
    constexpr int BW = WW == -1 ? 2 : (WW == -2 ? 4 : 1);   // 64-bit words of the band
    constexpr int MARGIN = 16 * BW;                         // band rows below the walk's window
    constexpr int WWE = HALF ? 1 : (WW >= NW ? NW : WW);   // words actually kept per vector and column
    constexpr bool DECIDED = WWE == 1;   // the forward pass stores the walk's four decision bits per cell
    constexpr int NVEC = DECIDED ? 4 : 3;
    using WT = typename std::conditional<HALF && BW == 1, uint32_t,
                                         typename std::conditional<HALF && BW == 4, unsigned __int128, uint64_t>::type>::type;
    constexpr uint32_t WMASK = (HALF && BW == 1) ? 31u : ((HALF && BW == 4) ? 127u : 63u);
    constexpr bool WINDOWED = HALF || WW < NW;   // windows slide with the column; else whole columns from row 0
    // One wave per block: its LDS holds the pattern-match vectors during the forward pass and,
    // once that is over, the read and the genome window the two walks fetch symbols from
    // (a fetch from HBM on every eighth step of every lane would stall the whole wave).
    static_assert(BLOCK == 64, "phases reuse the LDS without a barrier: one wave per block");
    constexpr int RD = 8 * NW + 1;                                               // read dwords
    constexpr int GD = (WW <= 1 && NW > 1 ? 64 * NW + 32 : (WW == 2 && NW > 2 ? 64 * NW + 64 : 128 * NW + 2)) / 8 + 4;
    constexpr int CH = 8;                                      // columns per refill of the walk's LDS chunk
    constexpr int CHB = WWE == 2 ? (CH + 1) * 3 * WWE * 2 : 0;  // chunk dwords per lane
    // banded pass: dwords per shifted pattern row = bits 0 .. (columns + 63), one more for the funnel
    constexpr int SD = (64 * NW + 16 * BW + 64 * BW + 31) / 32 + 1;
    constexpr int LDS_A = HALF ? 4 * SD : 12 * NW;              // forward pass
    constexpr int LDS_B = RD + GD + CHB;                        // staged symbols (+ the old walk's chunk)
    // the replay kernel (PHASE 2) sizes its staging by the batch's longest read, in dynamic LDS:
    // it is the LDS that limits its waves per CU, and with them its speed
    constexpr int LDS_DW = PHASE == 1 ? LDS_A : (PHASE == 2 ? 2 : (LDS_A > LDS_B ? LDS_A : LDS_B));
    extern __shared__ uint64_t lds_dyn[];
    static_assert(PHASE == 0 || HALF, "only the banded class is split");
    __shared__ uint64_t lds_all[LDS_DW * BLOCK / 2];
    uint64_t* pm_lds = lds_all;

    const uint32_t tid = threadIdx.x;
    const uint32_t count = *ap.count;
    uint64_t* mat = ap.slots + (uint64_t)blockIdx.x * ap.slot_words;
    WT* matw = reinterpret_cast<WT*>(mat);   // decision windows: [col][4][lane]
    uint32_t* const block_path = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(mat) +
                                                             (uint64_t)ap.max_cols * NVEC * WWE * sizeof(WT) * BLOCK);
    uint64_t* my_pm = pm_lds + tid;
    // decode = "ACGTNNXX" (LevMyers64.cpp:398), one byte per code
    auto decode = [](uint32_t c) -> uint32_t { return (uint32_t)(0x58584e4e54474341ull >> (8 * (c & 7u))) & 0xffu; };

    for (uint32_t base = blockIdx.x * BLOCK; base < count; base += gridDim.x * BLOCK) {
        const uint32_t li = base + tid;
        bool valid = li < count;
        const uint32_t gid = valid ? ap.list[li] : 0;
        // recorded steps, [16 steps][lane]: in the block's slot, or per group of 64 tasks when split
        uint32_t* const path = PHASE == 0 ? block_path : ap.paths + (uint64_t)(base / BLOCK) * ap.path_dwords * BLOCK;

        uint32_t m = 1, wn = 0, ed = 0, end_pos = 1, ref_pos = 0, read_idx = 0, presence = 0;
        bool rev = false;
        if (valid) {
            const wamd_dist_task task = ap.tasks[gid];
            const wamd_dist_result d = ap.dist[gid];
            read_idx = task.read_idx;
            m = rv.lens[read_idx];
            wn = (m - 1) / 64;
            ed = d.edit_distance;
            end_pos = d.pos;
            ref_pos = task.ref_pos;
            rev = task.flags & WAMD_TASK_FLAG_REV_COMP;
        }
        // first recomputed column (1-based) and number of columns
        const uint32_t span = m + ed;
        const uint32_t s_col = end_pos > span ? end_pos - span : 1u;
        uint32_t n_cols = valid ? end_pos - s_col + 1 : 0;
        if (valid && (n_cols > ap.max_cols || end_pos == 0)) {
            // cannot happen for ed <= m; keep the kernel safe anyway
            raise_flag(ap.flags, FLAG_ARG);
            valid = false;
            n_cols = 0;
        }

        // ---- banded class: common pieces of the forward pass and of the walk's recomputation ----
        const uint32_t F = (ap.max_cols + 7u) & ~7u;   // iterations of every lane
        uint64_t VPb[BW], VNb[BW];
#pragma unroll
        for (int w = 0; w < BW; ++w)
            VPb[w] = VNb[w] = 0;
        auto low_ones = [](int32_t nbits) -> uint64_t {   // ones in bits 0 .. nbits-1
            return nbits <= 0 ? 0ull : (nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull));
        };
        // Genome symbols under iterations fb .. fb + 7 (fb a multiple of 8) are the funnel of text
        // dwords tx_i0 + fb / 8 and the next one.  Iterations before the lane's first column may lie
        // before the genome's first symbol: any symbol will do there (dwords < 0 read as 0).
        const int64_t tx0 = (int64_t)ref_pos + (int64_t)s_col - 1 - (int64_t)(F - n_cols);
        const int32_t tx_i0 = (int32_t)(tx0 >> 3);
        const uint32_t tx_sh = 4u * ((uint32_t)tx0 & 7u);
        auto text_dword = [&](int32_t i) -> uint32_t { return i < 0 ? 0u : load_genome_dword(g, (uint32_t)i); };
        auto text_raw4 = [&](int32_t i, uint32_t (&out)[4]) {   // raw dwords i .. i + 3
            if (i >= 0) {
                load_genome_raw4(g, (uint32_t)i, out);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    out[c] = i + c < 0 ? 0u : g.dwords[(uint32_t)(i + c) < g.last_dword ? (uint32_t)(i + c) : g.last_dword];
            }
        };
        // one banded column: updates VPb / VNb, returns the four decision windows of the column
        // CHECKED: some lane of the wave may see a symbol >= 4 in this group of eight columns (asked once per group,
        // so that the common group is eight columns of straight-line code)
        auto band_col = [&](auto checked, uint32_t f, uint32_t c, WT (&win)[4]) {
            const uint32_t* S = reinterpret_cast<const uint32_t*>(lds_all) + tid;
            const uint32_t delay = F - n_cols;
            const bool act = valid && f >= delay;
            const uint32_t kd = f >> 5, sh = f & 31u;
            const uint32_t* sp = S + ((c & 3u) * SD + kd) * BLOCK;
            uint32_t a[2 * BW + 1];
#pragma unroll
            for (int q = 0; q <= 2 * BW; ++q)
                a[q] = sp[q * BLOCK];
            uint64_t eq[BW], we[BW];
#pragma unroll
            for (int w = 0; w < BW; ++w) {
                eq[w] = ((uint64_t)__builtin_amdgcn_alignbit(a[2 * w + 2], a[2 * w + 1], sh) << 32) |
                        __builtin_amdgcn_alignbit(a[2 * w + 1], a[2 * w], sh);
                we[w] = eq[w];
            }
            if constexpr (decltype(checked)::value) {   // N runs and other non-ACGT symbols
                if (c >= 4u) {
                    // read row of bit 0
                    const int32_t r0 = (int32_t)f + (int32_t)s_col + (int32_t)m - 2 - (int32_t)ed - (int32_t)end_pos - MARGIN -
                                       (int32_t)delay;
#pragma unroll
                    for (int w = 0; w < BW; ++w) {
                        eq[w] = low_ones(-r0 - 64 * w);
                        we[w] = 0;
                    }
                    // the walk compares min(c, N_ref) with the read's walk code (> 3 -> N_read for ASCII reads)
                    const uint32_t cw = c > 5u ? 5u : c;
                    if (act && (((presence >> c) | (presence >> cw)) & 1u)) {
                        const uint8_t* codes = rv.codes + rv.code_off[read_idx];
                        for (int32_t t = 0; t < 64 * BW; ++t) {
                            const int32_t r = r0 + t;
                            if (r >= 0 && r < (int32_t)m) {
                                const uint32_t rcode = rev ? rc_code(packed_symbol(codes, m - 1 - (uint32_t)r))
                                                           : packed_symbol(codes, (uint32_t)r);
                                const uint64_t e1 = (uint64_t)(rcode == c) << (t & 63);
                                const uint64_t e2 = (uint64_t)(((ap.ascii && rcode > 3u) ? 4u : rcode) == cw) << (t & 63);
#pragma unroll
                                for (int w = 0; w < BW; ++w)
                                    if ((t >> 6) == w) {
                                        eq[w] |= e1;
                                        we[w] |= e2;
                                    }
                            }
                        }
                    }
                }
            }
            // D0 = (((eq & VP) + VP) ^ VP) | eq | VN over the whole band (the carry crosses its words)
            uint64_t D0[BW], HP[BW], HN[BW];
            uint64_t carry = 0;
#pragma unroll
            for (int w = 0; w < BW; ++w) {
                const uint64_t x = eq[w] & VPb[w];
                const uint64_t s1 = x + VPb[w];
                const uint64_t s2 = s1 + carry;
                carry = (uint64_t)(s1 < x) | (uint64_t)(s2 < s1);
                D0[w] = (s2 ^ VPb[w]) | eq[w] | VNb[w];
                HP[w] = VNb[w] | ~(D0[w] | VPb[w]);
                HN[w] = D0[w] & VPb[w];
            }
            uint64_t vp_new[BW], vn_new[BW];
#pragma unroll
            for (int w = 0; w < BW; ++w) {
                const uint64_t D0s = (D0[w] >> 1) | (w + 1 < BW ? D0[w + 1 < BW ? w + 1 : w] << 63 : 0ull);
                vp_new[w] = HN[w] | ~(D0s | HP[w]);   // vertical deltas of this column, in the NEXT column's frame
                vn_new[w] = D0s & HP[w];
            }
#pragma unroll
            for (int w = 0; w < BW; ++w) {
                VPb[w] = act ? vp_new[w] : VPb[w];
                VNb[w] = act ? vn_new[w] : VNb[w];
            }
            // what getExtCigar tests at a cell (LevMyers64.cpp:419-442), as bits:
            // d_val + mismatch == c  <=>  D0 == match;  h_val + 1 == c  <=>  HP;  v_val + 1 == c  <=>  VP
            // the window: the 32 BW rows above the margin (vp_new is one row further down: the next column's frame)
            auto window = [](const uint64_t (&x)[BW], int below) -> WT {
                if constexpr (BW == 1) {
                    return (WT)(x[0] >> (below ? 15 : 16));
                } else if constexpr (BW == 2) {
                    return below ? (WT)((x[0] >> 31) | (x[1] << 33)) : (WT)((x[0] >> 32) | (x[1] << 32));
                } else {
                    const uint64_t lo = below ? (x[0] >> 63) | (x[1] << 1) : x[1];
                    const uint64_t hi = below ? (x[1] >> 63) | (x[2] << 1) : x[2];
                    return ((WT)hi << 64) | (WT)lo;
                }
            };
            uint64_t dm[BW], dx[BW];
#pragma unroll
            for (int w = 0; w < BW; ++w) {
                dm[w] = D0[w] & we[w];
                dx[w] = ~(D0[w] | we[w]);
            }
            win[0] = window(dm, 0);
            win[1] = window(dx, 0);
            win[2] = window(HP, 0);
            win[3] = window(vp_new, 1);
        };
        const uint8_t* my_codes = rv.codes + rv.code_off[read_idx];
        if constexpr (PHASE == 2) {
            // the path and the totals are already in memory
        } else if constexpr (HALF) {
            // ---- banded forward pass (bands of up to 32 rows, ed <= 14) --------------------------
            // Only a diagonal band of 64 rows is computed (Hyyro's band form of the recurrence:
            // the frame moves down one row per column, so D0 shifts right instead of HP/HN
            // shifting left).  Bit t of a column is read row  column + const + t; the 32 rows
            // the walk may look at are bits 16..47.  A cell the walk tests as a predecessor has
            // value <= ed and sits within ed + 1 diagonals of the end diagonal, so its optimal
            // paths stay within 2ed + 1 diagonals: inside the band (16 rows of margin below the
            // window, 61 - 2ed above), hence its banded value is the true one; every other value
            // can only be too large, which fails the same tests.  Rows above the read's first
            // symbol are phantom rows (match everything, VP = 0: the free start of the text).
            //
            // Nothing of the matrix is stored.  Every lane runs the same F iterations and its
            // columns are the LAST n_cols of them, so "iteration f" is "column end_pos - (F-1-f)"
            // in every lane; the state before every eighth iteration is kept (16 B), and the
            // walk recomputes eight columns at a time from it into registers.  The pattern-match
            // rows go to LDS pre-shifted per lane so that iteration f reads bits f .. f + 63 of
            // them: dword index and funnel shift are wave-uniform.
            static_assert(4 * SD <= LDS_DW, "LDS plan");
            uint32_t* S = reinterpret_cast<uint32_t*>(lds_all) + tid;                 // [c][d][lane]
            const uint32_t delay = F - n_cols;                                        // iterations before column s_col
            // read row under bit 0 of iteration f is f + Zf
            const int32_t Zf = (int32_t)s_col + (int32_t)m - 2 - (int32_t)ed - (int32_t)end_pos - MARGIN - (int32_t)delay;
            {
                const uint32_t slot = valid ? (uint32_t)(rev ? rv.slot_rc : rv.slot_direct) : 0;
                const uint32_t* src = rv.pm32 + ((uint64_t)read_idx * rv.n_orient + slot) * 4 * rv.pm32_stride;
                if (valid)
                    presence = rv.meta[(uint64_t)read_idx * rv.n_orient + slot] & 0xffffu;
                // pm32 is top-aligned: read row r is bit r + pad, the bits below are ones (phantom rows)
                const int32_t zq = Zf + (int32_t)(32u * rv.words32 - m);   // bit of the pm32 row under x = 0
                const int32_t k0 = zq >> 5;                                // floor
                const uint32_t o = (uint32_t)zq & 31u;
                // dword k of all four rows: one 16-byte load (pm32 is dword-major)
                auto pext4 = [&](int32_t k, uint32_t (&out)[4]) {
                    const int32_t kc = k < 0 ? 0 : (k >= (int32_t)rv.words32 ? (int32_t)rv.words32 - 1 : k);
                    const uint4 v = reinterpret_cast<const uint4*>(src)[kc];
                    const uint32_t fill = k < 0 ? ~0u : 0u;
                    const bool inside = k >= 0 && k < (int32_t)rv.words32;
                    out[0] = inside ? v.x : fill;
                    out[1] = inside ? v.y : fill;
                    out[2] = inside ? v.z : fill;
                    out[3] = inside ? v.w : fill;
                };
                uint32_t lo[4];
                pext4(k0, lo);
#pragma unroll
                for (int d = 0; d < SD; ++d) {
                    uint32_t hi[4];
                    pext4(k0 + d + 1, hi);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        S[(c * SD + d) * BLOCK] = valid ? __builtin_amdgcn_alignbit(hi[c], lo[c], o) : 0u;
                        lo[c] = hi[c];
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < BW; ++w) {
                VPb[w] = ~low_ones(-(Zf + (int32_t)delay) - 64 * w);   // column s_col - 1: D = row, phantom rows flat
                VNb[w] = 0;
            }
            uint32_t t_lo = text_dword(tx_i0);
            for (uint32_t fb4 = 0; fb4 < F; fb4 += 32) {   // four text dwords per 16-byte load
                uint32_t raw[4];
                text_raw4(tx_i0 + 1 + (int32_t)(fb4 >> 3), raw);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t fb = fb4 + 8u * q;
                    if (fb < F) {
                        uint64_t* ck = mat + ((uint64_t)(fb / 8) * 2 * BW) * BLOCK + tid;
#pragma unroll
                        for (int w = 0; w < BW; ++w) {
                            ck[(2 * w) * BLOCK] = VPb[w];
                            ck[(2 * w + 1) * BLOCK] = VNb[w];
                        }
                        const uint32_t t_hi = nibble_swap(raw[q]);
                        const uint32_t syms = __builtin_amdgcn_alignbit(t_hi, t_lo, tx_sh);
                        t_lo = t_hi;
                        if (__any(valid && (syms & 0xccccccccu))) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                WT win[4];
                                band_col(std::true_type{}, fb + u, (syms >> (4 * u)) & 15u, win);
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                WT win[4];
                                band_col(std::false_type{}, fb + u, (syms >> (4 * u)) & 15u, win);
                            }
                        }
                    }
                }
            }
        } else {
        // ---- pattern-match vectors -> LDS (as in the distance kernel) ---------------
        {
            const uint32_t slot = valid ? (uint32_t)(rev ? rv.slot_rc : rv.slot_direct) : 0;
            const uint64_t* src = rv.pm + ((uint64_t)read_idx * rv.n_orient + slot) * 4 * rv.words;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    my_pm[(c * NW + w) * BLOCK] = (valid && (uint32_t)w <= wn) ? src[c * rv.words + w] : 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                my_pm[(4 * NW + w) * BLOCK] = 0;
                my_pm[(5 * NW + w) * BLOCK] = 0;
            }
            if (valid)
                presence = rv.meta[(uint64_t)read_idx * rv.n_orient + slot] & 0xffffu;
        }

        // ---- forward recompute over columns s_col .. end_pos --------------------------
        // window of column j starts at row bit off(j) = max(0, m - 2 - ed - (end_pos - j))
        uint64_t VP[NW], VN[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            VP[w] = ~0ull;
            VN[w] = 0;
        }
        {
            const uint32_t first_sym = ref_pos + s_col - 1;
            const uint32_t shift = (first_sym & 7u) * 4u;
            uint32_t didx = first_sym >> 3;
            uint32_t d_lo = load_genome_dword(g, didx);
            uint32_t d_hi = load_genome_dword(g, didx + 1);
            uint32_t d_nx = load_genome_dword(g, didx + 2);
            didx += 3;
            for (uint32_t cb = 0; __any(cb < n_cols); cb += 8) {
                const uint32_t syms = __builtin_amdgcn_alignbit(d_hi, d_lo, shift);
                d_lo = d_hi;
                d_hi = d_nx;
                d_nx = load_genome_dword(g, didx++);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t ci = cb + u;           // 0-based recomputed column
                    const uint32_t c = (syms >> (4 * u)) & 15u;
                    const uint32_t row = c < 4u ? c : 4u;
                    const bool act = ci < n_cols;
                    // the walk compares min(c, N_ref) with the read's walk code (>3 -> N_read for ASCII reads)
                    const uint32_t cw = c > 5u ? 5u : c;
                    const bool exotic = act && c >= 4u && (((presence >> c) | (presence >> cw)) & 1u);
                    if (exotic) {
                        for (uint32_t w = 0; w <= wn; ++w) {
                            uint64_t eq = 0;
                            const uint32_t lo = w * 64, hi = min(m, lo + 64);
                            uint64_t weq = 0;
                            for (uint32_t i = lo; i < hi; ++i) {
                                const uint32_t rcode = rev ? rc_code(packed_symbol(my_codes, m - 1 - i))
                                                           : packed_symbol(my_codes, i);
                                eq |= (uint64_t)(rcode == c) << (i - lo);
                                weq |= (uint64_t)(((ap.ascii && rcode > 3u) ? 4u : rcode) == cw) << (i - lo);
                            }
                            my_pm[(4 * NW + w) * BLOCK] = eq;
                            my_pm[(5 * NW + w) * BLOCK] = weq;
                        }
                    }
                    const uint64_t* eqrow = my_pm + row * (NW * BLOCK);
                    const uint64_t* weqrow = my_pm + (row < 4u ? row : 5u) * (NW * BLOCK);
                    uint64_t hn_in = 0, hp_in = 0;
                    uint64_t D0v[NW];
                    uint64_t FDM[DECIDED ? NW : 1], FDX[DECIDED ? NW : 1], FH[DECIDED ? NW : 1];
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        uint64_t X = eqrow[w * BLOCK] | hn_in;
                        const uint64_t vp = VP[w], vn = VN[w];
                        const uint64_t D0 = (((X & vp) + vp) ^ vp) | X | vn;
                        const uint64_t HP = vn | ~(D0 | vp);
                        const uint64_t HN = D0 & vp;
                        X = (HP << 1) | hp_in;
                        VP[w] = (HN << 1) | ~(D0 | X) | hn_in;
                        VN[w] = D0 & X;
                        hn_in = HN >> 63;
                        hp_in = HP >> 63;
                        D0v[w] = D0;
                        if constexpr (DECIDED) {
                            // what getExtCigar tests at a cell (LevMyers64.cpp:419-442), as bits:
                            // d_val + mismatch == c  <=>  D0 == match;  h_val + 1 == c  <=>  HP;  v_val + 1 == c  <=>  VP
                            const uint64_t we = weqrow[w * BLOCK];
                            FDM[w] = D0 & we;
                            FDX[w] = ~(D0 | we);
                            FH[w] = HP;
                        }
                    }
                    if (exotic) {
                        for (uint32_t w = 0; w <= wn; ++w) {
                            my_pm[(4 * NW + w) * BLOCK] = 0;
                            my_pm[(5 * NW + w) * BLOCK] = 0;
                        }
                    }
                    if (act) {
                        const uint32_t j = s_col + ci;
                        const int32_t o = (int32_t)m - 2 - (int32_t)ed - (int32_t)(end_pos - j);
                        const uint32_t off = (WINDOWED && o > 0) ? (uint32_t)o : 0u;
                        if constexpr (DECIDED) {
                            WT* dw = matw + ((uint64_t)ci * 4) * BLOCK + tid;
                            if constexpr (HALF) {
                                dw[0 * BLOCK] = extract32<NW>(FDM, off);
                                dw[1 * BLOCK] = extract32<NW>(FDX, off);
                                dw[2 * BLOCK] = extract32<NW>(FH, off);
                                dw[3 * BLOCK] = extract32<NW>(VP, off);
                            } else {
                                uint64_t w1[1];
                                extract_window<NW, 1>(FDM, off, w1);
                                dw[0 * BLOCK] = w1[0];
                                extract_window<NW, 1>(FDX, off, w1);
                                dw[1 * BLOCK] = w1[0];
                                extract_window<NW, 1>(FH, off, w1);
                                dw[2 * BLOCK] = w1[0];
                                extract_window<NW, 1>(VP, off, w1);
                                dw[3 * BLOCK] = w1[0];
                            }
                        } else {
                        uint64_t wv[WWE];
                        uint64_t* dst = mat + ((uint64_t)ci * 3 * WWE) * BLOCK + tid;
                        extract_window<NW, WWE>(VP, off, wv);
#pragma unroll
                        for (int t = 0; t < WWE; ++t)
                            dst[(0 * WWE + t) * BLOCK] = wv[t];
                        extract_window<NW, WWE>(VN, off, wv);
#pragma unroll
                        for (int t = 0; t < WWE; ++t)
                            dst[(1 * WWE + t) * BLOCK] = wv[t];
                        extract_window<NW, WWE>(D0v, off, wv);
#pragma unroll
                        for (int t = 0; t < WWE; ++t)
                            dst[(2 * WWE + t) * BLOCK] = wv[t];
                        }
                    }
                }
            }
        }

        }
        // ---- backward walk (getExtCigar, LevMyers64.cpp:413-513) ----------------------
        // The walk reads columns pos and pos - 1 and moves left one column at a time.  Fetching
        // their words from HBM as it goes leaves every step waiting for memory, so the narrow
        // classes keep CH + 1 columns per lane in LDS: when any lane of the wave runs out of
        // its chunk, every lane reloads the CH + 1 columns ending at its own position, all
        // requests in flight at once.  The widest class reads the words where they lie.
        constexpr bool CACHED = WWE == 2;
        constexpr int CW = CACHED ? 3 * WWE : 1;
        auto window_off = [&](uint32_t j) -> uint32_t {
            if constexpr (WW < NW) {
                const int32_t o = (int32_t)m - 2 - (int32_t)ed - (int32_t)(end_pos - j);
                return o > 0 ? (uint32_t)o : 0u;
            } else {
                return 0u;
            }
        };
        uint64_t* chunk = reinterpret_cast<uint64_t*>(reinterpret_cast<uint32_t*>(lds_all) + (RD + GD) * BLOCK) + tid;
        uint32_t chunk_hi = 0;   // column held in slot 0; slot c holds column chunk_hi - c
        auto refill = [&](uint32_t top, bool on) {
            if constexpr (CACHED) {
                chunk_hi = top;
                uint64_t v[(CH + 1) * CW];
#pragma unroll
                for (int c = 0; c <= CH; ++c) {
                    const uint32_t j = top - (uint32_t)c;
                    const bool inside = on && top >= (uint32_t)c && j >= s_col && j <= end_pos;
                    const uint64_t* src = mat + ((uint64_t)(inside ? j - s_col : 0) * 3 * WWE) * BLOCK + tid;
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        // the column before the window: VP all ones, VN and D0 zero
                        v[c * CW + q] = q < WWE ? ~0ull : 0ull;
                        if (inside)
                            v[c * CW + q] = src[q * BLOCK];
                    }
                }
#pragma unroll
                for (int i = 0; i < (CH + 1) * CW; ++i)
                    chunk[i * BLOCK] = v[i];
            }
        };
        auto col_bit = [&](uint32_t j, int vec, uint32_t beta) -> uint32_t {
            const uint32_t b = beta - (j >= s_col ? window_off(j) : 0u);
            const uint32_t slot = chunk_hi - j;
            uint32_t w = vec * WWE;
            if constexpr (WWE > 1)
                w += (b >> 6) ? (WWE - 1) : 0;
            const uint64_t word = chunk[(slot * CW + w) * BLOCK];
            return (uint32_t)(word >> (b & 63u)) & 1u;
        };
        // bit of vector `vec` in the window of column j (1-based) at row bit beta
        auto mat_bit = [&](uint32_t j, int vec, uint32_t beta) -> uint32_t {
            if (j < s_col)   // the column before the window: VP all ones, VN zero
                return vec == 0 ? 1u : 0u;
            const uint32_t ci = j - s_col;
            const uint32_t b = beta - window_off(j);
            const uint64_t word = mat[(((uint64_t)ci * 3 + vec) * WWE + (b >> 6)) * BLOCK + tid];
            return (uint32_t)(word >> (b & 63u)) & 1u;
        };

        uint32_t read_pos = valid ? m : 0, pos = valid ? end_pos : 0;
        uint32_t c_val = ed, events = 0, nsteps = 0, pacc = 0;
        uint32_t prev_dir = DIR_D;
        uint32_t in_indel = 0;   // 0 / 1
        double score = 0;
        // ---- stage the read and the genome symbols under the recomputed columns ------------
        const int RDc = PHASE == 2 ? (int)ap.rd_cap : RD, GDc = PHASE == 2 ? (int)ap.gd_cap : GD;
        // (the replay kernel reads the read straight from its code dwords, one dword per eight
        // symbols and lane: staging it as well would cost a third of the waves the LDS allows)
        uint32_t* st_read = reinterpret_cast<uint32_t*>(PHASE == 2 ? lds_dyn : lds_all) + tid;
        uint32_t* st_gen = st_read + RDc * BLOCK;
        const int32_t rd_n = valid ? (int32_t)((m + 7) / 8) : 1;
        // the walk reads genome symbols ref_pos + s_col - 2 .. ref_pos + end_pos - 1
        const int64_t g_first = (int64_t)ref_pos + (int64_t)s_col - 2;
        const int64_t g_origin = (g_first < 0 ? 0 : g_first) >> 3;
        int32_t gd_n = valid ? (int32_t)((((int64_t)ref_pos + end_pos - 1) >> 3) - g_origin + 1) : 1;
        gd_n = gd_n > GDc ? GDc : (gd_n < 1 ? 1 : gd_n);
        if constexpr (PHASE != 1) {
            // 16-byte loads (a wave's load costs the 64 cache lines it touches, whatever its width);
            // the codes buffer has 64 bytes of slack behind the last read
            const uint32_t* src = reinterpret_cast<const uint32_t*>(my_codes);
            for (int d4 = 0; d4 < RDc; d4 += 4) {   // RDc = 0 in the replay kernel
                if (d4 < rd_n) {
                    const wamd_u32x4 v = *reinterpret_cast<const wamd_u32x4_a4*>(src + d4);
                    const uint32_t b[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (d4 + c < RDc && d4 + c < rd_n)
                            st_read[(d4 + c) * BLOCK] = nibble_swap(b[c]);
                }
            }
            for (int d4 = 0; d4 < GDc; d4 += 4) {
                if (d4 < gd_n) {
                    uint32_t b[4];
                    load_genome_raw4(g, (uint32_t)(g_origin + d4), b);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (d4 + c < GDc && d4 + c < gd_n)
                            st_gen[(d4 + c) * BLOCK] = nibble_swap(b[c]);
                }
            }
        }
        // symbol s of a staged sequence (branch-free: the walk fetches both symbols every step)
        auto staged_sym = [&](const uint32_t* base, uint32_t origin_dw, int32_t n_dw, uint32_t idx) -> uint32_t {
            int32_t d = (int32_t)((idx >> 3) - origin_dw);
            d = d < 0 ? 0 : (d >= n_dw ? n_dw - 1 : d);
            return (base[d * BLOCK] >> (4u * (idx & 7u))) & 15u;
        };
        auto read_code = [&](uint32_t raw) -> uint32_t {
            uint32_t c = rev ? rc_code(raw) : raw;
            // copy_direct / convert_to_rev_comp: anything but ACGT is N_read (sam.cpp:2296-2347)
            return (ap.ascii && c > 3u) ? 4u : c;
        };
        // oriented read symbol q (0-based) = code[q], or rc(code[m-1-q]) when reversed
        auto read_sym = [&](uint32_t q) -> uint32_t {
            return read_code(staged_sym(st_read, 0u, rd_n, rev ? m - 1 - q : q));
        };
        auto genome_sym = [&](uint32_t col) -> uint32_t {   // 1-based column of the task's window
            const uint32_t x = staged_sym(st_gen, (uint32_t)g_origin, gd_n, ref_pos + col - 1);
            return x > 5u ? 5u : x;   // ref_copy_direct, sam.cpp:2372-2374
        };
        // step codes enter `pacc` from the top: after sixteen of them the first sits in bits 0..1
        auto push_step = [&](uint32_t dir) {
            pacc = (pacc >> 2) | (dir << 30);
            ++nsteps;
            if ((nsteps & 15u) == 0)
                path[(uint64_t)(nsteps / 16 - 1) * BLOCK + tid] = pacc;
        };
        const double c_match = ap.sc.match, c_mism = ap.sc.mismatch, c_do = ap.sc.gap_del_open,
                     c_de = ap.sc.gap_del_extend, c_io = ap.sc.gap_ins_open, c_ie = ap.sc.gap_ins_extend;
        // the six of them, one per lane (see `step`)
        const double cost_lane = tid == 0 ? c_match : (tid == 1 ? c_mism : (tid == 2 ? c_do : (tid == 3 ? c_de : (tid == 4 ? c_io : (tid == 5 ? c_ie : 0.0)))));
        const int cost_lo = __double2loint(cost_lane), cost_hi = __double2hiint(cost_lane);

        if constexpr (PHASE == 2) {
            // totals of the walk, left in the result record by the first half
            if (valid) {
                const wamd_align_result h = ap.results[gid];
                score = h.affine_score;
                events = h.num_events;
                nsteps = h.cigar_len;
                pos = h.begin_pos;
            }
            read_pos = 0;
        } else if constexpr (DECIDED) {
            // ---- bands of up to 64 rows: column-synchronous walk over decision bits ----------
            // Super-step s serves column end_pos - s in every lane: any number of vertical
            // moves, then the one diagonal or horizontal move that leaves the column.  Which
            // column a lane is in is then a function of s alone, so the four window words of
            // a column sit in registers with static indices (four columns per batch of loads,
            // the next batch already in flight); no symbol is fetched and the cell value is
            // not tracked - the forward pass has already turned them into the four tests.
            const int32_t A = WINDOWED ? (int32_t)m - 2 - (int32_t)ed : 0;   // window_off(end_pos - s) = max(0, A - s), A - s when banded
            const uint32_t last_s = end_pos - s_col;          // columns further left do not exist for the walk
            auto load_col = [&](uint32_t sc_, WT (&dst)[4]) {
                const bool inside = valid && sc_ <= last_s;
                const WT* src = matw + ((uint64_t)(inside ? last_s - sc_ : 0u) * 4) * BLOCK + tid;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    dst[v] = src[v * BLOCK];
            };
            // One move costs what the table says (six wave-uniform doubles, lane i of `cost_lane` holds entry i and
            // lane 6 zero for lanes that sit a round out): fetched across lanes, not picked by a chain of selects.
            //   0 match, 1 mismatch, 2 deletion opens, 3 extends, 4 insertion opens, 5 extends
            auto step = [&](const WT (&cur)[4], uint32_t s_) {
                const int32_t oc = A - (int32_t)s_;
                const uint32_t off_c = (HALF || oc > 0) ? (uint32_t)oc : 0u;   // banded windows keep sliding above row 0
                bool pending = read_pos && pos;
                if (!__any(pending))
                    return;
                do {
                    const uint32_t b = ((pending ? read_pos : 1u) - 1u - off_c) & WMASK;
                    const uint32_t fdm = (uint32_t)(cur[0] >> b) & 1u, fdx = (uint32_t)(cur[1] >> b) & 1u;
                    const uint32_t fh = (uint32_t)(cur[2] >> b) & 1u, fv = (uint32_t)(cur[3] >> b) & 1u;
                    const uint32_t idx = (prev_dir << 3) | ((fdm | fdx) << 2) | (fh << 1) | fv;
                    const uint32_t dir = (uint32_t)(WALK_TABLE >> (2u * idx)) & 3u;
                    const uint32_t cidx = pending ? (dir ? 2u * dir + in_indel : fdx) : 6u;
                    const double cost = __hiloint2double(__builtin_amdgcn_ds_bpermute((int)(cidx * 4u), cost_hi),
                                                         __builtin_amdgcn_ds_bpermute((int)(cidx * 4u), cost_lo));
                    score += cost;   // + 0.0 for a lane that is not pending
                    if (pending) {
                        const uint32_t gap = dir ? 1u : 0u;
                        events += gap & (in_indel ^ 1u);
                        in_indel = gap;
                        read_pos -= (5u >> dir) & 1u;   // diagonal, vertical
                        pos -= (3u >> dir) & 1u;        // diagonal, horizontal
                        prev_dir = dir;
                        push_step(PHASE == 1 && dir == DIR_D ? 3u * fdx : dir);   // DIR_X = 3: a diagonal step over a mismatch
                        pending = dir == DIR_V && read_pos != 0;
                    }
                } while (__any(pending));
            };
            if constexpr (HALF) {
                // eight columns at a time: reload the state kept before iteration F - 8(q+1), run the
                // band forward into registers, then walk those columns right to left
                const uint32_t n_chunks = F / 8;
                auto ck_load = [&](uint32_t q, uint64_t (&vp)[BW], uint64_t (&vn)[BW]) {
                    const uint32_t idx = q < n_chunks ? n_chunks - 1 - q : 0u;
                    const uint64_t* ck = mat + ((uint64_t)idx * 2 * BW) * BLOCK + tid;
#pragma unroll
                    for (int w = 0; w < BW; ++w) {
                        vp[w] = ck[(2 * w) * BLOCK];
                        vn[w] = ck[(2 * w + 1) * BLOCK];
                    }
                };
                uint64_t nvp[BW], nvn[BW];
                ck_load(0, nvp, nvn);
                // text dwords right to left: the low dword of a group is the high one of the next
                uint32_t t_lo = text_dword(tx_i0 + (int32_t)n_chunks - 1);
                uint32_t nsyms = __builtin_amdgcn_alignbit(text_dword(tx_i0 + (int32_t)n_chunks), t_lo, tx_sh);
                for (uint32_t q = 0; q < n_chunks && __any(read_pos && pos); ++q) {
                    const uint32_t fb = F - 8 * (q + 1);
#pragma unroll
                    for (int w = 0; w < BW; ++w) {
                        VPb[w] = nvp[w];
                        VNb[w] = nvn[w];
                    }
                    const uint32_t syms = nsyms;
                    ck_load(q + 1, nvp, nvn);
                    {
                        const uint32_t t_hi = t_lo;
                        t_lo = text_dword(tx_i0 + (int32_t)(fb >> 3) - 1);
                        nsyms = __builtin_amdgcn_alignbit(t_hi, t_lo, tx_sh);
                    }
                    WT D[8][4];
                    if (__any(valid && (syms & 0xccccccccu))) {
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            band_col(std::true_type{}, fb + u, (syms >> (4 * u)) & 15u, D[u]);
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            band_col(std::false_type{}, fb + u, (syms >> (4 * u)) & 15u, D[u]);
                    }
#pragma unroll
                    for (int u = 7; u >= 0; --u)
                        step(D[u], F - 1 - (fb + u));
                }
            } else {
            WT C[4][4], N[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                load_col(u, N[u]);
            for (uint32_t k = 0; __any(read_pos && pos); ++k) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        C[u][v] = N[u][v];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    load_col(4 * k + 4 + u, N[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    step(C[u], 4 * k + u);
            }
            }
        } else {
        refill(end_pos, valid);
        while (__any(read_pos && pos)) {
            const bool on = read_pos && pos;
            const uint32_t rp = on ? read_pos : 1u, cp = on ? pos : chunk_hi;   // finished lanes idle on safe indices
            if constexpr (CACHED) {
                if (__any(on && cp < chunk_hi - (uint32_t)CH + 1u && chunk_hi >= (uint32_t)CH))
                    refill(cp, on);
            }
            const uint32_t rsym = read_sym(rp - 1);
            const uint32_t gsym = genome_sym(cp);
            const uint32_t beta = rp - 1;
            uint32_t vp, vn, d0, vp_l, vn_l;
            if constexpr (CACHED) {
                vp = col_bit(cp, 0, beta);
                vn = col_bit(cp, 1, beta);
                d0 = col_bit(cp, 2, beta);
                vp_l = col_bit(cp - 1, 0, beta);
                vn_l = col_bit(cp - 1, 1, beta);
            } else {
                vp = mat_bit(cp, 0, beta);
                vn = mat_bit(cp, 1, beta);
                d0 = mat_bit(cp, 2, beta);
                vp_l = mat_bit(cp - 1, 0, beta);
                vn_l = mat_bit(cp - 1, 1, beta);
            }
            const uint32_t d_val = c_val - (d0 ^ 1u);
            const uint32_t v_val = c_val - vp + vn;
            const uint32_t h_val = d_val + vp_l - vn_l;   // C(i, j-1) = C(i-1, j-1) + V(i, j-1)
            const uint32_t mism = rsym != gsym;
            const bool from_h = h_val + 1 == c_val;
            const bool from_v = v_val + 1 == c_val;
            const bool from_d = d_val + mism == c_val;
            // D > H > V, but a move that continues the previous direction wins (LevMyers64.cpp:431-442)
            const bool take_d = from_d && (prev_dir == DIR_D || (prev_dir == DIR_H && !from_h) ||
                                           (prev_dir == DIR_V && !from_v));
            const bool take_h = !take_d && from_h && (prev_dir == DIR_H || (prev_dir == DIR_D && !from_d) ||
                                                      (prev_dir == DIR_V && !from_v));
            const int dir = take_d ? DIR_D : (take_h ? DIR_H : DIR_V);
            const double cost_d = mism ? c_mism : c_match;
            const double cost_h = in_indel ? c_de : c_do;
            const double cost_v = in_indel ? c_ie : c_io;
            const double cost = take_d ? cost_d : (take_h ? cost_h : cost_v);
            if (on) {
                score += cost;
                events += (!take_d && !in_indel) ? 1u : 0u;
                in_indel = !take_d;
                read_pos -= take_h ? 0u : 1u;
                pos -= (take_d || take_h) ? 1u : 0u;
                c_val = take_d ? d_val : (take_h ? h_val : v_val);
                prev_dir = dir;
                push_step(dir);
            }
        }
        }
        while (__any(read_pos)) {   // LevMyers64.cpp:498-513
            if (read_pos) {
                if (in_indel) {
                    score += ap.sc.gap_ins_extend;
                } else {
                    score += ap.sc.gap_ins_open;
                    ++events;
                    in_indel = true;
                }
                --read_pos;
                push_step(DIR_V);
            }
        }
        if (PHASE != 2 && (nsteps & 15u))
            path[(uint64_t)(nsteps / 16) * BLOCK + tid] = pacc >> (32u - 2u * (nsteps & 15u));
        const uint32_t begin_pos = pos;
        if constexpr (PHASE == 1) {   // hand over to the replay kernel
            if (li < count) {
                wamd_align_result h{};
                h.affine_score = score;
                h.num_events = events;
                h.cigar_len = nsteps;
                h.begin_pos = begin_pos;
                h.edit_distance = valid ? ed : RESULT_ERROR;
                ap.results[gid] = h;
            }
            continue;
        }

        // ---- forward replay: ext-CIGAR in final order + refinement -----------------------
        uint32_t cur = 0, acc = 0, refined = 0;
        uint8_t* slot = ap.cigars + (uint64_t)gid * ap.cigar_stride;
        if constexpr (PHASE == 2) {
            if (valid && ap.results[gid].edit_distance == RESULT_ERROR)
                valid = false;
        }
        if (valid && (2 * nsteps + 4 > ap.cigar_stride)) {
            raise_flag(ap.flags, FLAG_CIGAR);
            valid = false;
        }
        // Bytes gather in `acc`, dwords in a 16-byte buffer that goes out with one store: every lane
        // writes its own slot, so a wave's store costs the 64 lines it touches whatever its width.
        uint32_t qb0 = 0, qb1 = 0, qb2 = 0, qb3 = 0;
        auto emit = [&](uint32_t byte) {
            acc |= byte << (8 * (cur & 3u));
            ++cur;
            if ((cur & 3u) == 0) {
                const uint32_t qi = ((cur >> 2) - 1u) & 3u;
                qb0 = qi == 0 ? acc : qb0;
                qb1 = qi == 1 ? acc : qb1;
                qb2 = qi == 2 ? acc : qb2;
                qb3 = qi == 3 ? acc : qb3;
                acc = 0;
                if (qi == 3) {
                    wamd_u32x4 v = {qb0, qb1, qb2, qb3};
                    *reinterpret_cast<wamd_u32x4_a4*>(slot + cur - 16) = v;
                }
            }
        };
        // memory holds every byte emitted so far (the refinement patches bytes in place)
        auto flush_partial = [&]() {
            const uint32_t blk = cur & ~15u, full = (cur & 15u) >> 2;
            if (full > 0)
                *reinterpret_cast<uint32_t*>(slot + blk) = qb0;
            if (full > 1)
                *reinterpret_cast<uint32_t*>(slot + blk + 4) = qb1;
            if (full > 2)
                *reinterpret_cast<uint32_t*>(slot + blk + 8) = qb2;
            for (uint32_t q = cur & ~3u; q < cur; ++q)
                slot[q] = (uint8_t)(acc >> (8 * (q & 3u)));
        };
        // the buffer and `acc` as memory holds them at position `cur`
        auto reload_partial = [&]() {
            const uint32_t blk = cur & ~15u, full = (cur & 15u) >> 2;
            qb0 = full > 0 ? *reinterpret_cast<const uint32_t*>(slot + blk) : 0u;
            qb1 = full > 1 ? *reinterpret_cast<const uint32_t*>(slot + blk + 4) : 0u;
            qb2 = full > 2 ? *reinterpret_cast<const uint32_t*>(slot + blk + 8) : 0u;
            qb3 = 0;
            acc = 0;
            for (uint32_t q = cur & ~3u; q < cur; ++q)
                acc |= (uint32_t)slot[q] << (8 * (q & 3u));
        };
        NibStream rf, gf;
        if constexpr (PHASE == 2)
            rf.init(reinterpret_cast<const uint32_t*>(my_codes), 1, 0, rd_n, rev ? (int64_t)m - 1 : 0, true);
        else
            rf.init(st_read, BLOCK, 0, rd_n, rev ? (int64_t)m - 1 : 0);
        gf.init(st_gen, BLOCK, g_origin, gd_n, (int64_t)ref_pos + begin_pos);
        uint32_t hist = 0, ref_hist = 0;
        uint32_t pw = 0, pw_next = 0;
        const uint32_t total = valid ? nsteps : 0;
        for (uint32_t k = 0; __any(k < total); ++k) {
            if (k < total) {
                const uint32_t st = total - 1 - k;   // steps were recorded end to start
                if (k == 0) {
                    pw = path[(uint64_t)(st / 16) * BLOCK + tid];
                    pw_next = st >= 16 ? path[(uint64_t)(st / 16 - 1) * BLOCK + tid] : 0u;
                } else if ((st & 15u) == 15u) {   // the dword asked for sixteen steps ago; ask for the one after it
                    pw = pw_next;
                    pw_next = st >= 16 ? path[(uint64_t)(st / 16 - 1) * BLOCK + tid] : 0u;
                }
                const uint32_t dir = (pw >> (2 * (st & 15u))) & 3u;
                uint32_t op, b0, b1 = 0;
                if (dir == DIR_D) {
                    const uint32_t r = read_code(rf.get(rev ? -1 : 1));
                    const uint32_t x = gf.get(1);
                    const uint32_t gsy = x > 5u ? 5u : x;
                    ref_hist = (ref_hist << 4) | gsy;
                    op = r == gsy ? OP_M : OP_X;
                    b0 = r == gsy ? (uint32_t)'.' : decode(gsy);
                } else if (dir == DIR_H) {
                    const uint32_t x = gf.get(1);
                    const uint32_t gsy = x > 5u ? 5u : x;
                    ref_hist = (ref_hist << 4) | gsy;
                    op = OP_D;
                    b0 = '^';
                    b1 = decode(gsy);
                } else {
                    const uint32_t r = read_code(rf.get(rev ? -1 : 1));
                    op = OP_I;
                    b0 = '#';
                    b1 = decode(r);
                }
                // refine_ext_cigar (sam_matching.cpp:858-924): first ". ^x . . #y ." or
                // ". #y . . ^x ." becomes three mismatches.  hist = the five previous ops.
                if (ap.refine && !refined && op == OP_M) {
                    const uint32_t pat_a = (OP_M << 12) | (OP_D << 9) | (OP_M << 6) | (OP_M << 3) | OP_I;
                    const uint32_t pat_b = (OP_M << 12) | (OP_I << 9) | (OP_M << 6) | (OP_M << 3) | OP_D;
                    // the current match has just pushed its own reference symbol
                    if (hist == pat_a || hist == pat_b) {
                        refined = 1;
                        flush_partial();
                        const uint32_t q0 = cur - 6;   // "^x..#y" / "#y..^x"
                        slot[q0 + 0] = (uint8_t)decode((ref_hist >> 12) & 15u);
                        slot[q0 + 1] = (uint8_t)decode((ref_hist >> 8) & 15u);
                        slot[q0 + 2] = (uint8_t)decode((ref_hist >> 4) & 15u);
                        __threadfence_block();   // the patched bytes are read back just below
                        cur = q0 + 3;
                        reload_partial();
                        ++events;
                        score += 3 * ap.sc.mismatch;
                        score -= ap.sc.gap_ins_open + ap.sc.gap_del_open;
                    }
                }
                hist = ((hist << 3) | op) & 0x7fffu;
                emit(b0);
                if (op == OP_D || op == OP_I)
                    emit(b1);
            }
        }
        if (valid) {
            const uint32_t len = cur;
            emit(0);
            {   // what the buffer still holds: its full dwords and the one being filled
                const uint32_t blk = cur & ~15u, full = (cur & 15u) >> 2;
                if (full > 0)
                    *reinterpret_cast<uint32_t*>(slot + blk) = qb0;
                if (full > 1)
                    *reinterpret_cast<uint32_t*>(slot + blk + 4) = qb1;
                if (full > 2)
                    *reinterpret_cast<uint32_t*>(slot + blk + 8) = qb2;
                if (cur & 3u)
                    *reinterpret_cast<uint32_t*>(slot + (cur & ~3u)) = acc;
            }
            wamd_align_result r;
            r.affine_score = score;
            r.ok = 1;
            r.edit_distance = ed;
            r.end_pos = end_pos;
            r.begin_pos = begin_pos;
            r.num_events = events;
            r.cigar_len = len;
            r.refined = refined;
            r.reserved = 0;
            ap.results[gid] = r;
        } else if (li < count) {
            wamd_align_result r{};
            r.edit_distance = RESULT_ERROR;
            ap.results[gid] = r;
            slot[0] = 0;
        }
    }
}

// ---- replay of the banded class: the recorded path becomes the ext-CIGAR ----------------------
// The walk (PHASE 1) leaves two bits per step, last step first: match, mismatch, deletion ('^' + genome
// letter), insertion ('#' + read letter).  A 150 bp alignment is some 160 steps of which a handful are
// not matches, so the string is written as what it mostly is - dots, sixteen per store - and then only
// the edits are visited: a lane skips the matches between two edits with one count-leading-zeros, and a
// wave's trip count per path dword is the largest number of edits any of its lanes has in it, not 16.
// Symbols are fetched for edits only (a deletion's or mismatch's genome letter, an insertion's read
// letter), straight from memory: no staging, no LDS.
// refine_ext_cigar (sam_matching.cpp:858-924) rewrites the string's first ". ^x . . #y ." or
// ". #y . . ^x ." into three mismatches.  In operations that is: an indel whose predecessor is a match,
// the opposite indel exactly three operations later with no edit in between, and a match behind it -
// all of which is known at the NEXT edit (or at the end), where the six bytes are rewritten and
// everything behind them moves down by three (nothing behind them has been written yet but dots).
// STAGED (reads short enough for a wave's strings to fit in LDS; one wave per block): the string is built in
// the lane's LDS row (an odd number of dwords: the lanes' rows start in different banks) and the finished
// rows leave as the dots do in the other form - the wave writing its 64 slots together, 16 bytes a lane.
// Of the direct form's 0.33 ms per 2^20 alignments 0.11 - 0.15 ms were the edits' byte stores (64 lines per
// instruction): 0.21 ms with the rows in LDS, 0.18 ms with an edit's symbol asked for one edit ahead.
// (Also staging the alignment's genome and code dwords - no memory access left in the loop - was slower,
// 0.29 ms: 22 KB of LDS per wave leave seven waves per CU, and they wait for their first loads instead.)
template <bool STAGED>
__global__ __launch_bounds__(STAGED ? 64 : 256) void align_replay_kernel(GenomeView g, ReadsView rv, AlignParams ap)
{
    extern __shared__ uint32_t replay_lds[];   // STAGED: [lane][rp_sw], the wave's strings
    constexpr uint32_t BLOCK = STAGED ? 64u : 256u;
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t count = *ap.count;
    const bool inside = li < count;   // lanes behind the list's end still help with the dots
    const uint32_t gid = inside ? ap.list[li] : 0u;
    const uint32_t* const path = ap.paths + (uint64_t)(li >> 6) * ap.path_dwords * 64u + (li & 63u);   // [16 steps][lane]
    wamd_align_result h{};   // totals of the walk
    if (inside)
        h = ap.results[gid];
    uint8_t* const gslot = ap.cigars + (uint64_t)gid * ap.cigar_stride;
    uint32_t* const row = replay_lds + lane * ap.rp_sw;
    uint8_t* const slot = STAGED ? reinterpret_cast<uint8_t*>(row) : gslot;
    const uint32_t nsteps = h.cigar_len;
    bool valid = inside && h.edit_distance != RESULT_ERROR;
    // the string has nsteps + (indels) <= nsteps + ed bytes and its terminator
    uint32_t need = valid ? (nsteps + h.edit_distance + 1u + 3u) & ~3u : 0u;   // <= 2 nsteps + 4
    if (valid && (2 * nsteps + 4 > ap.cigar_stride || (STAGED && need > 4u * ap.rp_sw))) {
        raise_flag(ap.flags, FLAG_CIGAR);
        valid = false;
        need = 0;
    }
    if (inside && !valid) {
        wamd_align_result r{};
        r.edit_distance = RESULT_ERROR;
        ap.results[gid] = r;
        gslot[0] = 0;
    }
    uint32_t need_max = need;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)need_max, o);
        need_max = other > need_max ? other : need_max;
    }
    // ---- dots ------------------------------------------------------------------------------------
    // A lane filling its own slot in memory would touch a line of its own with every store (64 lines per
    // instruction).  The wave serves its 64 slots together instead: 16-byte block j of the wave belongs to
    // slot j / nb, so neighbouring lanes write neighbouring blocks of one slot.  (STAGED: the same loop
    // carries the finished rows out at the end; here the lane only fills its LDS row.)
    auto wave_out = [&](bool from_rows) {
        const uint32_t nb = (need_max + 15u) >> 4;            // blocks per slot (wave-uniform)
        const uint32_t inv = nb ? (1u << 20) / nb + 1u : 0u;   // j / nb = (j * inv) >> 20 for j < 64 nb <= 2^13
        for (uint32_t i = 0; i < nb; ++i) {
            const uint32_t j = i * 64u + lane;
            const uint32_t t = (j * inv) >> 20;
            const uint32_t off = (j - t * nb) * 16u;
            const uint32_t t_need = (uint32_t)__shfl((int)need, (int)t);
            const uint32_t t_gid = (uint32_t)__shfl((int)gid, (int)t);
            uint8_t* const dst = ap.cigars + (uint64_t)t_gid * ap.cigar_stride + off;
            wamd_u32x4 v = {0x2e2e2e2eu, 0x2e2e2e2eu, 0x2e2e2e2eu, 0x2e2e2e2eu};
            if (from_rows && off < t_need) {   // (rows are rp_sw dwords: the last block may reach past need, never past the row's use)
                const uint32_t* const src = replay_lds + t * ap.rp_sw + (off >> 2);
                const uint32_t left = (t_need - off) >> 2;
                v.x = src[0];
                v.y = left > 1u ? src[1] : 0u;
                v.z = left > 2u ? src[2] : 0u;
                v.w = left > 3u ? src[3] : 0u;
            }
            if (off + 16u <= t_need) {
                *reinterpret_cast<wamd_u32x4_a4*>(dst) = v;
            } else {
                const uint32_t piece[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c)
                    if (off + 4u * c < t_need)
                        *reinterpret_cast<uint32_t*>(dst + 4u * c) = piece[c];
            }
        }
    };
    if (STAGED) {
        for (uint32_t d = 0; 4u * d < need_max; ++d)
            if (4u * d < need)
                row[d] = 0x2e2e2e2eu;
    } else {
        wave_out(false);
        // the edits below overwrite bytes that other lanes have just written: those stores first
        __threadfence_block();
    }
    if (valid) {
    const wamd_dist_task task = ap.tasks[gid];
    const wamd_dist_result d = ap.dist[gid];
    const uint32_t m = rv.lens[task.read_idx];
    const bool rev = task.flags & WAMD_TASK_FLAG_REV_COMP;
    const uint8_t* const codes = rv.codes + rv.code_off[task.read_idx];
    const uint32_t* const code_dwords = reinterpret_cast<const uint32_t*>(codes);   // reads start on dwords (reads_prepare.hip)
    uint64_t gi = (uint64_t)task.ref_pos + h.begin_pos;         // next genome symbol
    // raw dword holding genome symbol `at` / read code `at` (clamped into the genome like load_genome_dword)
    auto fetch_raw = [&](bool from_read, uint64_t at) -> uint32_t {
        const uint32_t dw = (uint32_t)(at >> 3);
        const uint32_t* const src = from_read ? code_dwords + dw : g.dwords + (dw < g.last_dword ? dw : g.last_dword);
        return *src;
    };
    // decode = "ACGTNNXX" (LevMyers64.cpp:398)
    auto decode = [](uint32_t c) -> uint32_t { return (uint32_t)(0x58584e4e54474341ull >> (8 * (c & 7u))) & 0xffu; };
    auto genome_letter = [&](uint64_t at) -> uint32_t {   // ref_copy_direct clamps to N_ref (sam.cpp:2372-2374)
        const uint32_t x = (nibble_swap(fetch_raw(false, at)) >> (4u * ((uint32_t)at & 7u))) & 15u;
        return decode(x > 5u ? 5u : x);
    };

    uint32_t cur = 0, ri = 0, opi = 0;                          // next byte, next read symbol, next operation
    // refinement: the last edit, the indel that may open the pattern, the pattern waiting for the match behind it
    int32_t prev_t = -1;
    bool open = false, pend = false;
    uint32_t t1 = 0, k1 = 0, c1 = 0, t2 = 0, pc = 0, refined = 0;
    uint64_t g1 = 0, pg = 0;
    double score = h.affine_score;
    uint32_t events = h.num_events;
    auto refine_now = [&]() {
        slot[pc + 0] = (uint8_t)genome_letter(pg);   // the three genome symbols under the four operations
        slot[pc + 1] = (uint8_t)genome_letter(pg + 1);
        slot[pc + 2] = (uint8_t)genome_letter(pg + 2);
        slot[pc + 3] = '.';
        slot[pc + 4] = '.';
        slot[pc + 5] = '.';
        cur -= 3;
        refined = 1;
        ++events;
        score += 3 * ap.sc.mismatch;
        score -= ap.sc.gap_ins_open + ap.sc.gap_del_open;
    };

    // the edit whose symbol is on its way
    bool e_have = false;
    uint32_t e_raw = 0, e_sh = 0, e_cur = 0, e_k = 0;
    auto emit = [&]() {
        if (!e_have)
            return;
        const bool ins = e_k == DIR_V;
        const uint32_t sym = (nibble_swap(e_raw) >> e_sh) & 15u;
        const uint32_t oriented = rev ? rc_code(sym) : sym;
        // copy_direct / convert_to_rev_comp: anything but ACGT is N_read (sam.cpp:2296-2347);
        // ref_copy_direct clamps to N_ref (sam.cpp:2372-2374)
        const uint32_t letter = decode(ins ? ((ap.ascii && oriented > 3u) ? 4u : oriented) : (sym > 5u ? 5u : sym));
        slot[e_cur] = (uint8_t)(e_k == DIR_X ? letter : (ins ? (uint32_t)'#' : (uint32_t)'^'));
        if (e_k != DIR_X)
            slot[e_cur + 1] = (uint8_t)letter;
        e_have = false;
    };
    const uint32_t q_last = nsteps ? (nsteps - 1u) >> 4 : 0u;
    uint32_t w_next = nsteps ? path[(uint64_t)q_last * 64u] : 0u;
    for (uint32_t qq = 0; nsteps && qq <= q_last; ++qq) {
        const uint32_t q = q_last - qq;
        const uint32_t w = w_next;
        if (q)
            w_next = path[(uint64_t)(q - 1u) * 64u];   // in flight while this one is served
        uint32_t left = q == q_last ? nsteps - 16u * q : 16u;   // steps of this dword, the highest field first
        uint32_t wl = w << (2u * (16u - left));
        while (wl) {
            const uint32_t z = (uint32_t)__builtin_clz(wl) >> 1;   // matches before the edit
            cur += z;
            ri += z;
            gi += z;
            opi += z;
            wl <<= 2u * z;
            const uint32_t k = wl >> 30;
            wl <<= 2;
            left -= z + 1u;
            // one fetch whatever the edit: the genome dword of a mismatch or deletion, the code dword of an insertion.
            // It is asked for here and looked at one edit later (emit): the wait is the next edit's bookkeeping.
            const bool ins = k == DIR_V;
            const uint64_t at = ins ? (uint64_t)(rev ? m - 1u - ri : ri) : gi;
            const uint32_t raw = fetch_raw(ins, at);
            emit();
            if (ap.refine && !refined) {
                if (pend) {
                    pend = false;
                    if (opi > t2 + 1u)   // a match behind the second indel
                        refine_now();
                }
                const bool indel = k == DIR_H || k == DIR_V;
                if (!refined && open && indel && opi == t1 + 3u && k != k1) {   // (no edit since t1: two matches between)
                    pend = true;
                    t2 = opi;
                    pc = c1;
                    pg = g1;
                }
                open = indel && (int32_t)opi - 1 > prev_t;   // its predecessor exists and is a match
                t1 = opi;
                k1 = k;
                c1 = cur;
                g1 = gi;
                prev_t = (int32_t)opi;
            }
            e_have = true;
            e_raw = raw;
            e_sh = 4u * ((uint32_t)at & 7u);
            e_cur = cur;
            e_k = k;
            cur += k == DIR_X ? 1u : 2u;
            ri += k != DIR_H ? 1u : 0u;
            gi += ins ? 0u : 1u;
            ++opi;
        }
        cur += left;
        ri += left;
        gi += left;
        opi += left;
    }
    emit();
    if (pend && t2 + 1u < nsteps)
        refine_now();
    slot[cur] = 0;
    wamd_align_result r;
    r.affine_score = score;
    r.ok = 1;
    r.edit_distance = d.edit_distance;
    r.end_pos = d.pos;
    r.begin_pos = h.begin_pos;
    r.num_events = events;
    r.cigar_len = cur;
    r.refined = refined;
    r.reserved = 0;
    ap.results[gid] = r;
    }
    if (STAGED) {
        __syncthreads();   // one wave per block: the rows are complete
        wave_out(true);
    }
}

// ---- host side -----------------------------------------------------------------------

static inline uint64_t up256(uint64_t x) { return (x + 255) / 256 * 256; }

struct AlignLayout {
    uint64_t dist_off, lists_off, counters_off, paths_off, slots_off, total;
    uint64_t slot_bytes_max;   // one block of the widest class
    uint32_t blocks;           // blocks of class 0 the slot region is sized for
    uint32_t block;            // threads per block
    uint32_t cols[4], path_dwords;   // per class: columns recomputed at most
    uint64_t per_block[4];           // per class: bytes of one block's slot
};

static AlignLayout align_layout(const wamd_reads* r, uint32_t n, uint32_t max_ed)
{
    AlignLayout L{};
    const uint32_t nw = r->words ? r->words : 1;
    const uint32_t m = r->max_len ? r->max_len : 1;
    L.block = 64;   // one wave per block, see align_traceback_kernel
    L.path_dwords = (2 * m + 15) / 16 + 1;
    // class 0: ed <= 14 (banded: only the state before every eighth column is kept, 16 B), 1: ed <= 30
    // (four 64-row decision windows per column), 2: ed <= 62 and 3: ed <= m (VP, VN and D0 windows)
    L.cols[0] = m + 14 + 2;
    L.cols[1] = m + 30 + 2;
    L.cols[2] = m + 62 + 2;
    L.cols[3] = 2 * m + 2;
    const uint64_t col_bytes[4] = {0, 0, nw < 2 ? 4 * 8u : 3 * 2 * 8u, nw < 2 ? 4 * 8u : 3 * nw * 8u};
    uint64_t* per_block = L.per_block;
    L.slot_bytes_max = 0;
    for (int c = 0; c < 4; ++c) {
        per_block[c] = ((uint64_t)L.cols[c] * col_bytes[c] + (uint64_t)L.path_dwords * 4) * L.block;
        if (c < 3)   // banded: the state before every eighth column (16 / 32 / 64 bytes); the paths live in their own region
            per_block[c] = (uint64_t)((L.cols[c] + 7) / 8) * 16 * (1u << c) * L.block;
        per_block[c] = up256(per_block[c]);
        L.slot_bytes_max = per_block[c] > L.slot_bytes_max ? per_block[c] : L.slot_bytes_max;
    }
    const uint32_t need_blocks = (n + L.block - 1) / L.block;
    const uint32_t cap_blocks = 256 * 32;   // twice the sixteen waves per CU that are resident: finished waves are replaced at once
    L.blocks = need_blocks < cap_blocks ? need_blocks : cap_blocks;
    if (L.blocks == 0)
        L.blocks = 1;
    // the region serves the classes one after the other: all blocks of the first banded class (ed <= 14; the others,
    // ed <= 30 and ed <= 62, take two and four times as much a block and run that many fewer at a time; whole columns -
    // ed beyond 62 - one block at least).  The launcher uses whatever the caller gives beyond the minimum.
    uint64_t region = (uint64_t)L.blocks * per_block[0];
    (void)max_ed;
    if (region < L.slot_bytes_max)
        region = L.slot_bytes_max;
    L.dist_off = 0;
    L.lists_off = up256((uint64_t)n * sizeof(wamd_dist_result));
    L.counters_off = L.lists_off + up256((uint64_t)4 * n * sizeof(uint32_t));
    // recorded paths of the split (banded) class, per group of 64 tasks
    L.paths_off = L.counters_off + 256;
    L.slots_off = L.paths_off + up256((uint64_t)need_blocks * L.path_dwords * 4 * L.block);
    L.total = L.slots_off + region;
    return L;
}

uint64_t align_workspace_bytes(const wamd_reads* r, uint32_t n, uint32_t max_ed)
{
    return align_layout(r, n, max_ed).total;
}

template <int NW, int WW, int BLOCK>
static int launch_tb(const GenomeView& g, const ReadsView& rv, AlignParams ap, uint32_t blocks, uint32_t groups,
                     hipStream_t s)
{
    if constexpr (WW <= 0) {
        hipLaunchKernelGGL((align_traceback_kernel<NW, WW, BLOCK, 1>), dim3(blocks), dim3(BLOCK), 0, s, g, rv, ap);
        const uint32_t lds = 256u * ap.rp_sw;   // a wave's rows
        if (lds <= 40u * 1024u)   // (four waves per CU at least; longer reads take the form without LDS)
            hipLaunchKernelGGL(align_replay_kernel<true>, dim3(groups ? groups : 1u), dim3(64), lds, s, g, rv, ap);
        else
            hipLaunchKernelGGL(align_replay_kernel<false>, dim3((groups + 3u) / 4u ? (groups + 3u) / 4u : 1u), dim3(256), 0, s, g, rv, ap);
    } else {
        hipLaunchKernelGGL((align_traceback_kernel<NW, WW, BLOCK, 0>), dim3(blocks), dim3(BLOCK), 0, s, g, rv, ap);
    }
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

template <int NW, int BLOCK>
static int launch_tb_class(int cls, const GenomeView& g, const ReadsView& rv, const AlignParams& ap,
                           uint32_t blocks, uint32_t groups, hipStream_t s)
{
    if (cls == 0)
        return launch_tb<NW, 0, BLOCK>(g, rv, ap, blocks, groups, s);
    if (cls == 1)
        return launch_tb<NW, -1, BLOCK>(g, rv, ap, blocks, groups, s);
    if (cls == 2)
        return launch_tb<NW, -2, BLOCK>(g, rv, ap, blocks, groups, s);
    return launch_tb<NW, NW, BLOCK>(g, rv, ap, blocks, groups, s);
}

int launch_align(wamd_ctx* ctx, const wamd_reads* r, const wamd_dist_task* d_tasks, uint32_t n,
                 int last_symbol, int refine, wamd_align_result* d_results, uint8_t* d_cigars,
                 uint32_t cigar_stride, void* d_ws, uint64_t ws_bytes, hipStream_t s, int distance_mode)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (distance_mode < 0)
        distance_mode = ctx->distance_mode;
    if ((cigar_stride & 3u) || ((uintptr_t)d_cigars & 3u) || ((uintptr_t)d_ws & 255u)) {
        set_error("wamd_align: cigar_stride must be a multiple of 4, buffers aligned (cigars 4, workspace 256)");
        return WAMD_ERR_INVALID_ARG;
    }
    const AlignLayout L = align_layout(r, n, 0);
    if (ws_bytes < L.total)
        return WAMD_ERR_INVALID_ARG;
    char* ws = (char*)d_ws;
    auto* d_dist = (wamd_dist_result*)(ws + L.dist_off);
    auto* d_lists = (uint32_t*)(ws + L.lists_off);
    auto* d_counters = (uint32_t*)(ws + L.counters_off);
    auto* d_slots = (uint64_t*)(ws + L.slots_off);
    const uint64_t region = (ws_bytes & ~255ull) - L.slots_off;   // everything the caller gave

    WAMD_HIP(hipMemsetAsync(d_counters, 0, 256, s));
    int rc = ensure_pm(r, s);   // the wide classes read the 64-bit vectors
    if (rc != WAMD_OK)
        return rc;
    // pass 1 follows the context's distance mode: fill_cigar_with_lev and find_pairs only go on when
    // dynamicProgramming returned true (sam_matching.cpp:704-705), so in accepted-only mode the rejected
    // tasks come back as {k + 1, 0}, not ok, from the banded kernels
    rc = launch_distance(ctx, r, d_tasks, n, last_symbol, d_dist, distance_mode, s);
    if (rc)
        return rc;
    const ReadsView rv = r->view();
    const GenomeView g = genome_view(ctx);
    hipLaunchKernelGGL(align_classify_kernel, dim3((n + 1023) / 1024), dim3(1024), 0, s, rv, d_tasks, d_dist,
                       n, d_results, d_cigars, cigar_stride, d_lists, d_counters);
    WAMD_HIP(hipGetLastError());

    for (int cls = 0; cls < 4; ++cls) {
        AlignParams ap;
        ap.tasks = d_tasks;
        ap.dist = d_dist;
        ap.list = d_lists + (uint64_t)cls * n;
        ap.count = d_counters + cls;
        ap.results = d_results;
        ap.cigars = d_cigars;
        ap.slots = d_slots;
        ap.paths = (uint32_t*)(ws + L.paths_off);
        ap.flags = ctx->d_flags;
        ap.sc = ctx->scoring;
        ap.cigar_stride = cigar_stride;
        ap.max_cols = L.cols[cls];
        ap.path_dwords = L.path_dwords;
        ap.refine = refine;
        ap.ascii = r->encoding == WAMD_READS_ASCII;
        // replay of the banded class (ed <= 14): the read, and the m + ed + 2 genome symbols the
        // alignment can span, from whatever nibble of a dword they start at
        ap.rd_cap = 0;   // the read comes straight from memory
        const uint32_t ed_max = cls == 0 ? 14u : (cls == 1 ? 30u : 62u);   // of the banded classes
        ap.gd_cap = (r->max_len + ed_max + 16) / 8 + 2;
        // staged replay (ed <= 14): a string of at most m + 29 bytes and its terminator, rounded up to dwords - an odd
        // number of them
        ap.rp_sw = ((r->max_len + 2u * ed_max + 4u) >> 2) | 1u;
        const uint64_t per_block = L.per_block[cls];
        ap.slot_words = per_block / 8;
        uint64_t blocks64 = region / per_block;
        const uint32_t need = (n + L.block - 1) / L.block;
        uint32_t blocks = (uint32_t)(blocks64 < need ? blocks64 : need);
        if (blocks == 0)
            blocks = 1;
        switch (rv.words) {
        case 1: rc = launch_tb_class<1, 64>(cls, g, rv, ap, blocks, need, s); break;
        case 2: rc = launch_tb_class<2, 64>(cls, g, rv, ap, blocks, need, s); break;
        case 3: rc = launch_tb_class<3, 64>(cls, g, rv, ap, blocks, need, s); break;
        case 4: rc = launch_tb_class<4, 64>(cls, g, rv, ap, blocks, need, s); break;
        default:
            if (rv.words <= 8)
                rc = launch_tb_class<8, 64>(cls, g, rv, ap, blocks, need, s);
            else if (rv.words <= 16)
                rc = launch_tb_class<16, 64>(cls, g, rv, ap, blocks, need, s);
            else
                rc = WAMD_ERR_TOO_LONG;
        }
        if (rc)
            return rc;
    }
    return WAMD_OK;
}

} // namespace wamd
