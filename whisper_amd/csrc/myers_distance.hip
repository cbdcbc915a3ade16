// myers_distance.hip - edit-distance verification, one candidate per lane.
//
// Replaces LevMyers{64,128,256}::dynamicProgramming in "distance only" mode
// (mapping stages: mapper/mapping_core_sa.cpp:237/289; also the first pass of the
// alignment path).  The DP is the one the reference computes with the Myers/Hyyro block
// recurrence (mapper/LevMyers64.cpp:312-325); since that recurrence is exact, any exact
// bit-parallel evaluation of the same columns yields the same HP/HN/D0 bits in the score
// row, and those bits are all the reference's bookkeeping reads.
//
// MI355X mapping
//   * one task per lane; the whole bit-vector (NW dwords of VP and VN) lives in that
//     lane's VGPRs, so the add carries straight through v_addc_co chains and the two
//     shifts are v_alignbit funnels: no per-word carry bookkeeping and none of the
//     reference's anti-diagonal skew (LevMyers256.hpp:138-159).  What the skew changes
//     observably - iteration count, sentinel iterations, early exit on the iteration
//     index - is carried over arithmetically;
//   * vectors are TOP-aligned (reads_prepare.hip): the read's last symbol is bit 31 of
//     the top dword whatever its length, the rows below its first symbol are "phantom"
//     rows that match every text symbol and start with VP = 0, which keeps D = 0 along
//     them - the semi-global boundary the real row 0 needs.  The score bits are then
//     always the sign bits of the top dword's HP/HN/D0;
//   * pattern-match vectors for A,C,G,T live in LDS as [dword][symbol][lane]: a wave's 64
//     lanes read 64 consecutive dwords whatever symbol each lane asks for
//     (conflict-free ds_read_b32), and the symbol only shifts into the address;
//   * the genome stays in the reference's 4-bit packing in HBM; every lane fetches its
//     window TEXT_CHUNK dwords at a time (back-to-back loads of one or two cache lines,
//     one batch ahead) and keeps them in registers;
//   * groups of 8 columns run on a branch-free fast path when the wave is uniform enough
//     (same variant, no lane at its last columns, only A/C/G/T in the text): score,
//     minimum, tie-break, early exit and freezing of finished lanes are all arithmetic
//     on a packed (distance, tie flag, column) key.  Everything else - text symbols >= 4,
//     ragged ends, mixed variants, oversized windows - takes a per-lane slow group with
//     the reference's bookkeeping spelled out (LevMyers256.hpp:174-234,
//     LevMyers128.hpp:175-221, LevMyers64.cpp:328-366).
#include "wamd_internal.h"

namespace wamd {

#ifndef WAMD_TEXT_CHUNK
#define WAMD_TEXT_CHUNK 8
#endif
constexpr int TEXT_CHUNK = WAMD_TEXT_CHUNK;   // genome dwords (8 columns each) fetched per batch and lane

enum { RULE_M64 = 0, RULE_M128 = 1, RULE_M256 = 2 };
constexpr uint32_t RESULT_DEFERRED = 0xFFFFFFFEu;   // left to the next kernel of the banded chain

// a value the compiler must keep in a VGPR: VALU ops with an SGPR operand issue at the
// slow (VOP3) rate on gfx950, see scripts/ubench/valu_rates.hip
__device__ __forceinline__ uint32_t in_vgpr(uint32_t x)
{
    uint32_t r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
}

// ---- hand-picked VALU forms --------------------------------------------------------------
// Measured on gfx950 (scripts/ubench/valu_rates.hip, cycles per wave-instruction at 4 waves
// per SIMD): and/or/xor/not/add/sub/lshr/ashr with VGPR or literal operands 2.6; v_bitop3
// 3.2; every other VOP3 form, min/max, left shifts, carry adds and any VOP2 with an SGPR
// operand 4.3.  The compiler prices them all alike and fuses into v_bitop3 / v_or3 / v_add3
// wherever it can, so the hot expressions are spelled as the instructions they should be.
#define WAMD_VOP2(name, insn)                                                     \
    __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b)              \
    {                                                                             \
        uint32_t r;                                                               \
        asm(insn " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));                       \
        return r;                                                                 \
    }
WAMD_VOP2(v_and, "v_and_b32")
WAMD_VOP2(v_or, "v_or_b32")
WAMD_VOP2(v_add, "v_add_u32")
WAMD_VOP2(v_sub, "v_sub_u32")
WAMD_VOP2(v_min, "v_min_u32")
#undef WAMD_VOP2
__device__ __forceinline__ uint32_t v_not(uint32_t a)
{
    uint32_t r;
    asm("v_not_b32 %0, %1" : "=v"(r) : "v"(a));
    return r;
}
template <int SH>
__device__ __forceinline__ uint32_t v_lshr(uint32_t a)
{
    static_assert(SH >= 0 && SH < 32, "shift out of range");
    if (SH == 0)
        return a;
    uint32_t r;
    asm("v_lshrrev_b32 %0, %1, %2" : "=v"(r) : "n"(SH), "v"(a));
    return r;
}
template <uint32_t LIT>
__device__ __forceinline__ uint32_t v_and_lit(uint32_t a)
{
    uint32_t r;
    asm("v_and_b32 %0, %1, %2" : "=v"(r) : "n"(LIT), "v"(a));
    return r;
}
template <uint32_t LIT>
__device__ __forceinline__ uint32_t v_add_lit(uint32_t a)
{
    uint32_t r;
    asm("v_add_u32 %0, %1, %2" : "=v"(r) : "n"(LIT), "v"(a));
    return r;
}
// v_bitop3_b32: result bit = TT[(a << 2) | (b << 1) | c]; with a = 0xf0, b = 0xcc, c = 0xaa the
// table of an expression is the expression evaluated on those three constants.
template <int TT>
__device__ __forceinline__ uint32_t v_bitop3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(r) : "v"(a), "v"(b), "v"(c), "n"(TT));
    return r;
}
constexpr int TT_XOR_OR = (0xf0 ^ 0xcc) | 0xaa;            // (a ^ b) | c
constexpr int TT_OR_NOR = 0xf0 | (~(0xcc | 0xaa) & 0xff);  // a | ~(b | c)
constexpr int TT_OR_AND = 0xf0 | (0xcc & 0xaa);            // a | (b & c)
constexpr int TT_AND3 = 0xf0 & 0xcc & 0xaa;                // a & b & c

// One text column over the whole NW-dword vector (LevMyers64.cpp:312-325 as a single
// wide word).  Returns HP, HN and D0 of the top dword (score row = bit 31).
// HIN: the row above the vector is a real matrix row that is not computed (banded kernel);
// its horizontal delta is taken as +1 where `hin` is all ones, 0 where it is zero.
template <int NW, bool HIN = false>
__device__ __forceinline__ void column_core(const uint32_t (&eq)[NW], uint32_t (&VP)[NW], uint32_t (&VN)[NW],
                                            uint32_t& hp_top, uint32_t& hn_top, uint32_t& d0_top, uint32_t hin = 0)
{
    // stage-major order: the dwords of one stage are independent, so no instruction has to
    // wait for its neighbour's result (the compiler pads a dependent pair of asm ops)
    uint32_t x[NW], d0[NW], hp[NW], hn[NW], hps[NW], hns[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w)
        x[w] = v_and(eq[w], VP[w]);
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t cout;
        x[w] = __builtin_addc(x[w], VP[w], carry, &cout);
        carry = cout;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
        x[w] = v_bitop3<TT_XOR_OR>(x[w], VP[w], eq[w]);
#pragma unroll
    for (int w = 0; w < NW; ++w)
        d0[w] = v_or(x[w], VN[w]);
#pragma unroll
    for (int w = 0; w < NW; ++w)
        hp[w] = v_bitop3<TT_OR_NOR>(VN[w], d0[w], VP[w]);
#pragma unroll
    for (int w = 0; w < NW; ++w)
        hn[w] = v_and(d0[w], VP[w]);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        hps[w] = w ? __builtin_amdgcn_alignbit(hp[w], hp[w ? w - 1 : 0], 31)
                   : (HIN ? __builtin_amdgcn_alignbit(hp[w], hin, 31) : v_add(hp[w], hp[w]));
        hns[w] = w ? __builtin_amdgcn_alignbit(hn[w], hn[w ? w - 1 : 0], 31) : v_add(hn[w], hn[w]);
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
        VP[w] = v_bitop3<TT_OR_NOR>(hns[w], d0[w], hps[w]);
#pragma unroll
    for (int w = 0; w < NW; ++w)
        VN[w] = v_and(d0[w], hps[w]);
    hp_top = hp[NW - 1];
    hn_top = hn[NW - 1];
    d0_top = d0[NW - 1];
}

// phantom-row mask of dword j for a read of m symbols in an NW-dword vector
template <int NW>
__device__ __forceinline__ uint32_t phantom_mask(uint32_t m, int j)
{
    const uint32_t s = m >= 32u * NW ? 0u : 32u * NW - m;   // phantom rows
    const uint32_t lo = 32u * (uint32_t)j;
    return s >= lo + 32u ? ~0u : (s <= lo ? 0u : ((1u << (s - lo)) - 1u));
}

// Per-lane state between groups.  `plain` bookkeeping is the reference's own variables;
// the fast path packs it (see group_fast).
struct Lane {
    uint32_t m, wn, n_cols, exit_t, k, last_sym, presence;
    uint32_t rule;           // RULE_*
    bool rev;
    // plain bookkeeping
    uint32_t curr_ed, min_ed, min_pos;
    bool stopped;
};

// ---- fast group: 8 columns, wave-uniform rule, text symbols all < 4, no lane ends ------------
//   E     = curr_ed + column            (so the early exit "curr_ed + col > exit_t" is E > exit_t)
//   key   = min_ed << 17 | 0 << 16 | (0xffff - min_pos)
//   cand  = ced    << 17 | t << 16 | (0xffff - col)     t = 1 when a tie may NOT move min_pos
//   key'  = min(key, cand) & ~(1 << 16): a smaller distance always wins, an equal one wins
//           exactly when t = 0 (later column = smaller low half)
//   deadf : bit 31 set once the lane has left the loop (early exit, or past its last column:
//           n_left = n_cols - base); it lifts cand above every key
struct FastGroup {
    uint32_t lane_addr, syms, symsA, nz, nzA, gconst, nbase;   // per group
    uint32_t exit_t, n_left, rowmask, sgn, bit16;              // per lane / constants held in VGPRs
    uint32_t hin;                                              // banded kernel: see column_core
};

// LDS reads of the pattern-match vectors for symbol U of the group
template <int NW, int BLOCK, int U>
__device__ __forceinline__ void load_eq(const FastGroup& f, uint32_t (&eq)[NW])
{
    typedef const __attribute__((address_space(3))) uint32_t* lds_ptr;
    constexpr int RSH = BLOCK == 256 ? 10 : (BLOCK == 128 ? 9 : 8);   // log2(row stride in bytes)
    // symbol U moved to the row-stride position of the LDS byte address
    const uint32_t srow = U < 5 ? v_lshr<(U < 5 ? 12 + 4 * U - RSH : 0)>(f.symsA) : v_lshr<(U < 5 ? 0 : 4 * U - RSH)>(f.syms);
    const uint32_t addr = v_bitop3<TT_OR_AND>(f.lane_addr, srow, f.rowmask);
#pragma unroll
    for (int w = 0; w < NW; ++w)
        eq[w] = *(lds_ptr)(uintptr_t)(addr + (uint32_t)w * (4u * BLOCK * 4u));
}

// LIGHT: every column of the group lies before column m - k of every lane, where the score row
// is still above k (D[m][j] >= m - j): which column holds the minimum cannot reach the result
// (pos is 0 unless the distance ends at or under k, and that takes a strictly smaller value
// later on), so the tie-break terms are left out.
template <int NW, int BLOCK, int RULE, bool ENDS, bool HIN, bool LIGHT, int U>
__device__ __forceinline__ void fast_column(const FastGroup& f, uint32_t (&eq)[NW], uint32_t (&VP)[NW],
                                            uint32_t (&VN)[NW], uint32_t& E, uint32_t& key, uint32_t& deadf)
{
    // the next column's vectors are requested before this column's arithmetic
    uint32_t eq_next[NW];
    if (U < 7)
        load_eq<NW, BLOCK, (U < 7 ? U + 1 : 0)>(f, eq_next);
    uint32_t hp, hn, d0;
    column_core<NW, HIN>(eq, VP, VN, hp, hn, d0, f.hin);
    if (U < 7) {
#pragma unroll
        for (int w = 0; w < NW; ++w)
            eq[w] = eq_next[w];
    }

    const uint32_t nhn = v_not(hn);
    E = v_add(E, v_lshr<31>(hp));
    E = v_add(E, v_lshr<31>(nhn));   // + 1 for the column, - 1 when HN is set
    uint32_t cand = v_add_lit<0u - ((uint32_t)(U + 1) << 17) - (uint32_t)(U + 1)>((E << 17) + f.gconst);
    if (LIGHT) {
    } else if (RULE == RULE_M256) {
        // a tie moves min_pos unless D0 is set and the text symbol differs from last_symbol
        const uint32_t nzs = U < 4 ? v_lshr<(U < 4 ? 4 * U : 0)>(f.nzA) : v_lshr<(U < 4 ? 0 : 4 * U - 16)>(f.nz);
        cand = v_or(cand, v_bitop3<TT_AND3>(v_lshr<15>(d0), nzs, f.bit16));
    } else {
        // a tie moves min_pos only from the previous column (LevMyers128.hpp:205,
        // LevMyers64.cpp:349): key's low half is 0xffff - min_pos, that column's 0x10000 - col
        const uint32_t x = v_and_lit<0xffffu>(key ^ v_add_lit<0x10000u - (uint32_t)(U + 1)>(f.nbase));
        cand = v_or(cand, v_and_lit<0x10000u>(v_add_lit<0xffffu>(x)));
    }
    if (ENDS)   // negative past the lane's last column
        deadf = v_or(deadf, v_add_lit<0u - (uint32_t)(U + 1)>(f.n_left));
    cand = v_bitop3<TT_OR_AND>(cand, deadf, f.sgn);
    key = LIGHT ? v_min(key, cand) : v_and_lit<~0x10000u>(v_min(key, cand));
    if (RULE != RULE_M64)   // ced + col > exit_t and HN clear (LevMyers256.hpp:232-233)
        deadf = v_bitop3<TT_OR_AND>(deadf, v_sub(f.exit_t, E), nhn);
}

template <int NW, int BLOCK, int RULE, bool ENDS, bool HIN, bool LIGHT>
__device__ __forceinline__ void group_fast(uint32_t lane_addr, uint32_t syms, uint32_t base, uint32_t lsrep,
                                           uint32_t exit_t, uint32_t n_left, uint32_t rowmask, uint32_t sgn,
                                           uint32_t bit16, uint32_t hin, uint32_t (&VP)[NW], uint32_t (&VN)[NW],
                                           uint32_t& E, uint32_t& key, uint32_t& deadf)
{
    FastGroup f;
    f.hin = hin;
    f.lane_addr = lane_addr;
    f.syms = syms;
    f.symsA = syms << 12;
    f.nz = f.nzA = 0;
    if (RULE == RULE_M256 && !LIGHT) {
        // bit 4u = (text symbol u != last_symbol), LevMyers256.hpp:191-196
        uint32_t t = syms ^ lsrep;
        t |= t >> 1;
        t |= t >> 2;
        f.nz = t & 0x11111111u;
        f.nzA = f.nz << 16;
    }
    // cand = (ced << 17) + 0xffff - col with ced = E - col and col = base + u + 1
    //      = (E << 17) + gconst - ((u + 1) << 17) - (u + 1)
    f.gconst = in_vgpr((0u - (base << 17)) + (0xffffu - base));
    f.nbase = in_vgpr(0u - base);
    f.exit_t = exit_t;
    f.n_left = n_left;
    f.rowmask = rowmask;
    f.sgn = sgn;
    f.bit16 = bit16;
    uint32_t eq[NW];
    load_eq<NW, BLOCK, 0>(f, eq);
    fast_column<NW, BLOCK, RULE, ENDS, HIN, LIGHT, 0>(f, eq, VP, VN, E, key, deadf);
    fast_column<NW, BLOCK, RULE, ENDS, HIN, LIGHT, 1>(f, eq, VP, VN, E, key, deadf);
    fast_column<NW, BLOCK, RULE, ENDS, HIN, LIGHT, 2>(f, eq, VP, VN, E, key, deadf);
    fast_column<NW, BLOCK, RULE, ENDS, HIN, LIGHT, 3>(f, eq, VP, VN, E, key, deadf);
    fast_column<NW, BLOCK, RULE, ENDS, HIN, LIGHT, 4>(f, eq, VP, VN, E, key, deadf);
    fast_column<NW, BLOCK, RULE, ENDS, HIN, LIGHT, 5>(f, eq, VP, VN, E, key, deadf);
    fast_column<NW, BLOCK, RULE, ENDS, HIN, LIGHT, 6>(f, eq, VP, VN, E, key, deadf);
    fast_column<NW, BLOCK, RULE, ENDS, HIN, LIGHT, 7>(f, eq, VP, VN, E, key, deadf);
}

// What one task asks for, derived exactly as the reference's caller and loop do.
__device__ __forceinline__ uint32_t decode_task(const wamd_dist_task& task, const ReadsView& rv, const GenomeView& g,
                                                uint32_t max_text_len, uint32_t max_m, Lane& t)
{
    const uint32_t W = task.max_distance_in_ref;
    t.k = task.max_ed;
    t.rev = task.flags & WAMD_TASK_FLAG_REV_COMP;
    if (task.read_idx >= rv.n)
        return FLAG_READ_IDX;
    if ((t.rev ? rv.slot_rc : rv.slot_direct) < 0)
        return FLAG_ORIENT;
    if (W == 0 || W > max_text_len)
        return FLAG_TEXT_LEN;   // LevMyers64.cpp:244-252
    t.m = rv.lens[task.read_idx];
    uint32_t variant = task.variant;
    if (variant == WAMD_VARIANT_AUTO)   // mapping_core_sa.cpp:207-210
        variant = t.m < 128 ? WAMD_VARIANT_M128 : (t.m < 256 ? WAMD_VARIANT_M256 : WAMD_VARIANT_M64);
    t.wn = (t.m - 1) / 64;   // the reference's 64-bit word of the score bit
    if (variant > WAMD_VARIANT_M256 || (variant == WAMD_VARIANT_M128 && t.m > 128) ||
        (variant == WAMD_VARIANT_M256 && t.m > 256) || t.m > max_m)
        return FLAG_VARIANT;
    t.rule = variant == WAMD_VARIANT_M64 ? RULE_M64 : (variant == WAMD_VARIANT_M256 ? RULE_M256 : RULE_M128);
    // skewed variants iterate j = 1..W+1 and the score lane runs wn columns behind
    // (LevMyers256.hpp:138,180); M64 iterates j = 1..W (LevMyers64.cpp:257)
    t.n_cols = t.rule != RULE_M64 ? (W + 1 - t.wn) : W;
    if (t.rule != RULE_M64 && W + 1 < t.wn)
        t.n_cols = 0;
    t.exit_t = W + 1 - t.wn + t.k;
    if ((uint64_t)task.ref_pos + t.n_cols > g.n_symbols)
        return FLAG_RANGE;
    return 0;
}

static_assert(TEXT_CHUNK % 4 == 0, "the text stream fetches 16-byte pieces");
// first TEXT_CHUNK + 1 dwords of a lane's text, nibble-swapped
__device__ __forceinline__ void load_text_first(const GenomeView& g, uint32_t didx, uint32_t (&R)[TEXT_CHUNK + 1])
{
    uint32_t a[TEXT_CHUNK / 4][4];
#pragma unroll
    for (int q = 0; q < TEXT_CHUNK / 4; ++q)
        load_genome_raw4(g, didx + 4 * q, a[q]);
    const uint32_t last = g.dwords[didx + TEXT_CHUNK < g.last_dword ? didx + TEXT_CHUNK : g.last_dword];
#pragma unroll
    for (int q = 0; q < TEXT_CHUNK / 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            R[4 * q + c] = nibble_swap(a[q][c]);
    R[TEXT_CHUNK] = nibble_swap(last);
}
// the next TEXT_CHUNK raw dwords
__device__ __forceinline__ void load_text_chunk(const GenomeView& g, uint32_t didx, uint32_t (&NX)[TEXT_CHUNK])
{
#pragma unroll
    for (int q = 0; q < TEXT_CHUNK / 4; ++q) {
        uint32_t a[4];
        load_genome_raw4(g, didx + 4 * q, a);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            NX[4 * q + c] = a[c];
    }
}

// ---- the column loop: columns base0 + 1 .. n_cols of every lane, bookkeeping in `t` -----------
// pm_lds: this block's pattern-match rows [NW dwords][A,C,G,T][BLOCK lanes] (the banded kernel
// passes the top NW dwords of longer vectors); VP / VN: the state before column base0 + 1;
// t.curr_ed / min_ed / min_pos / stopped as they stand there.  `valid` lanes take part in the
// wave-uniform decisions, the others only ride along.
template <int NW, int BLOCK, bool HIN>
__device__ __forceinline__ void run_columns(const GenomeView& g, Lane& t, const bool valid, uint32_t (&VP)[NW],
                                            uint32_t (&VN)[NW], const uint32_t* pm_lds, const uint32_t tid,
                                            const uint8_t* my_codes, const uint32_t ref_pos, const uint32_t base0,
                                            const uint32_t hin)
{
    // ---- wave-uniform facts ------------------------------------------------------------------
    const uint64_t live0 = __ballot(valid);
    const uint32_t rule0 =
        __builtin_amdgcn_readfirstlane(__shfl((int)t.rule, live0 ? __ffsll((long long)live0) - 1 : 0));
    const bool uniform_rule = __all(!valid || t.rule == rule0);
    // the packed key holds 15 bits of distance + column and 16 bits of column
    const bool fits = __all(!valid || t.m + t.n_cols < 32000u);
    const bool can_fast = uniform_rule && fits;
    // columns before m - k of every lane: the score row is above k there (see fast_column, LIGHT)
    uint32_t light_until = valid ? (t.m > t.k ? t.m - t.k : 0u) : 0xffffffffu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        light_until = min(light_until, (uint32_t)__shfl_xor((int)light_until, o));
    light_until = __builtin_amdgcn_readfirstlane(light_until);

    const uint32_t lsrep = t.last_sym * 0x11111111u;
    const uint32_t rowmask = in_vgpr(3u << (BLOCK == 256 ? 10 : (BLOCK == 128 ? 9 : 8)));
    const uint32_t sgn = in_vgpr(0x80000000u);
    const uint32_t bit16 = in_vgpr(0x10000u);
    // LDS byte address of this lane's slot in row A of dword 0
    const uint32_t lane_addr =
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)pm_lds + tid * 4u;

    // ---- text stream: R = nibble-swapped dwords of the current chunk (+1 for the funnel),
    // NX = raw dwords of the next one, already in flight
    bool packed = false;   // wave-uniform: which bookkeeping form is current
    uint32_t E = 0, key = 0, deadf = 0;
    uint32_t base = base0;   // columns done (wave-uniform, a multiple of 8)

    const uint32_t shift = (ref_pos & 7u) * 4u;
    uint32_t didx = (ref_pos >> 3) + (base0 >> 3);
    uint32_t R[TEXT_CHUNK + 1], NX[TEXT_CHUNK];
    load_text_first(g, didx, R);
    didx += TEXT_CHUNK + 1;

    for (;;) {
        // next chunk's loads go out before this chunk's columns - when a lane has columns there
        if (__any(valid && t.n_cols > base + 8u * TEXT_CHUNK))
            load_text_chunk(g, didx, NX);
        didx += TEXT_CHUNK;

        bool done = false;
        for (int gi = 0; gi < TEXT_CHUNK; ++gi, base += 8) {
            const bool alive = packed ? (int32_t)deadf >= 0 : !t.stopped;
            if (!__any(alive)) {
                done = true;
                break;
            }
            const uint32_t syms = __builtin_amdgcn_alignbit(R[1], R[0], shift);
#pragma unroll
            for (int c = 0; c < TEXT_CHUNK; ++c)
                R[c] = R[c + 1];

            const bool fast = can_fast && !__any(alive && (syms & 0xccccccccu));
            if (fast) {
                if (!packed) {
                    E = t.curr_ed + base;
                    key = (t.min_ed << 17) | (0xffffu - t.min_pos);
                    deadf = t.stopped ? 0x80000000u : 0u;
                    packed = true;
                }
                const uint32_t n_left = t.n_cols - base;
                const bool ends = __any(alive && n_left < 8u);
                const bool light = !ends && base + 8u < light_until;
#define WAMD_GROUP(RULE, ENDS, LIGHT)                                                                        \
    group_fast<NW, BLOCK, RULE, ENDS, HIN, LIGHT>(lane_addr, syms, base, lsrep, t.exit_t, n_left, rowmask, sgn, bit16, \
                                                  hin, VP, VN, E, key, deadf)
                if (rule0 == RULE_M256) {
                    if (ends)
                        WAMD_GROUP(RULE_M256, true, false);
                    else if (light)
                        WAMD_GROUP(RULE_M256, false, true);
                    else
                        WAMD_GROUP(RULE_M256, false, false);
                } else if (rule0 == RULE_M128) {
                    if (ends)
                        WAMD_GROUP(RULE_M128, true, false);
                    else if (light)
                        WAMD_GROUP(RULE_M128, false, true);
                    else
                        WAMD_GROUP(RULE_M128, false, false);
                } else {
                    if (ends)
                        WAMD_GROUP(RULE_M64, true, false);
                    else if (light)
                        WAMD_GROUP(RULE_M64, false, true);
                    else
                        WAMD_GROUP(RULE_M64, false, false);
                }
#undef WAMD_GROUP
                continue;
            }

            // ---- slow group: per-lane bookkeeping, any text symbol ----------------------------
            if (packed) {
                if (!t.stopped) {   // lanes that were finished when packing keep their values
                    t.curr_ed = E - base;
                    t.min_ed = key >> 17;
                    t.min_pos = 0xffffu - (key & 0xffffu);
                    t.stopped = (int32_t)deadf < 0;
                }
                packed = false;
            }
#pragma unroll 1
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t col = base + u + 1;
                const uint32_t c = (syms >> (4 * u)) & 15u;
                const bool act = !t.stopped && col <= t.n_cols;
                // a text symbol >= 4 matches only a read holding that very code, which real
                // data never does (genome: 5/15, reads: 4/6); stay exact anyway
                const bool exotic = act && c >= 4u && ((t.presence >> c) & 1u);
                uint32_t eq[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    uint32_t v = phantom_mask<NW>(t.m, w);
                    if (c < 4u)
                        v = pm_lds[(w * 4 + c) * BLOCK + tid];
                    if (exotic) {
                        const int32_t first = 32 * w - (int32_t)(32u * NW - t.m);
                        for (int32_t b = 0; b < 32; ++b) {
                            const int32_t i = first + b;
                            if (i < 0 || i >= (int32_t)t.m)
                                continue;
                            const uint32_t rcode = t.rev ? rc_code(packed_symbol(my_codes, t.m - 1 - (uint32_t)i))
                                                         : packed_symbol(my_codes, (uint32_t)i);
                            v |= (uint32_t)(rcode == c) << b;
                        }
                    }
                    eq[w] = v;
                }
                uint32_t hp, hn, d0;
                column_core<NW, HIN>(eq, VP, VN, hp, hn, d0, hin);

                const uint32_t hnb = hn >> 31, hpb = hp >> 31, d0b = d0 >> 31;
                const uint32_t ced = t.curr_ed + hpb - hnb;
                const bool lt = ced < t.min_ed;
                const bool tie = ced == t.min_ed;
                const bool move = tie && (t.rule == RULE_M256 ? (!d0b || c == t.last_sym) : (t.min_pos + 1 == col));
                if (act) {
                    t.curr_ed = ced;
                    if (lt)
                        t.min_ed = ced;
                    if (lt || move)
                        t.min_pos = col;
                    // early exit on the iteration index j = col + wn
                    // (LevMyers256.hpp:232-233, LevMyers128.hpp:125, 219-220)
                    if (t.rule != RULE_M64 && !hnb && ced + col > t.exit_t)
                        t.stopped = true;
                }
                if (col >= t.n_cols)
                    t.stopped = true;   // last column of this lane
            }
        }
        if (done)
            break;
        // rotate the prefetched chunk in: R[0] already holds the funnel's low dword
#pragma unroll
        for (int c = 0; c < TEXT_CHUNK; ++c)
            R[c + 1] = nibble_swap(NX[c]);
    }

    if (packed && !t.stopped) {   // t.stopped: as it was when the state was packed
        t.min_ed = key >> 17;
        t.min_pos = 0xffffu - (key & 0xffffu);
    }
}

#ifndef WAMD_KATTR
#define WAMD_KATTR
#endif
// `chained`: the launch finishes what the banded kernels left (results marked RESULT_DEFERRED);
// `only_deferred`, when given, is the flag the previous kernel of the chain raised if it left
// anything: zero ends every block at once.  `clamp`: accepted-only mode, distances above k are
// reported as k + 1.
template <int NW, int BLOCK>
__global__ __launch_bounds__(BLOCK) WAMD_KATTR void myers_distance_kernel(
    GenomeView g, ReadsView rv, const wamd_dist_task* __restrict__ tasks, uint32_t n,
    int last_symbol, uint32_t max_text_len, wamd_dist_result* __restrict__ results,
    uint32_t* __restrict__ flags, int chained, const uint32_t* __restrict__ only_deferred, int clamp,
    const uint32_t* __restrict__ n_dev)
{
    // [dword][symbol A,C,G,T][lane]
    __shared__ uint32_t pm_lds[NW * 4 * BLOCK];

    if (only_deferred && *only_deferred == 0)
        return;
    if (n_dev) {   // the batch is a compacted list whose length only the device knows (cf_filter.hip)
        n = min(n, *n_dev);
        if (blockIdx.x * BLOCK >= n)
            return;
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * BLOCK + tid;
    bool valid = gid < n;
    if (valid && chained && results[gid].edit_distance != RESULT_DEFERRED)
        valid = false;
    if (chained && !__any(valid))   // wave-uniform; the block has no barriers
        return;

    Lane t{};
    uint32_t read_idx = 0, ref_pos = 0;
    if (valid) {
        const wamd_dist_task task = tasks[gid];
        const uint32_t err = decode_task(task, rv, g, max_text_len, 32u * NW, t);
        read_idx = task.read_idx;
        ref_pos = task.ref_pos;
        if (task.read_idx == WAMD_READ_NONE) {   // a slot without a task: skipped, no flag
            results[gid].edit_distance = RESULT_ERROR;
            results[gid].pos = 0;
            valid = false;
        } else if (err) {
            raise_flag(flags, err);
            results[gid].edit_distance = RESULT_ERROR;
            results[gid].pos = err;
            valid = false;
        }
    }
    if (!valid) {
        t = Lane{};
        t.m = 1;
        ref_pos = 0;
        read_idx = 0;
    }

    // ---- this lane's pattern-match vectors -> LDS; VP = 1 on real rows, 0 on phantom rows ----
    uint32_t VP[NW], VN[NW];
    {
        const uint32_t slot = valid ? (uint32_t)(t.rev ? rv.slot_rc : rv.slot_direct) : 0;
        const uint32_t pad = NW - rv.words32;   // launcher: NW >= words32
        const uint4* src = reinterpret_cast<const uint4*>(rv.pm32) + ((uint64_t)read_idx * rv.n_orient + slot) * rv.pm32_stride;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);   // dwords below the batch's own vectors: phantom rows
            if (valid && (uint32_t)j >= pad)
                v = src[(uint32_t)j - pad];              // A, C, G, T rows of this dword: one 16-byte load
            pm_lds[(j * 4 + 0) * BLOCK + tid] = v.x;
            pm_lds[(j * 4 + 1) * BLOCK + tid] = v.y;
            pm_lds[(j * 4 + 2) * BLOCK + tid] = v.z;
            pm_lds[(j * 4 + 3) * BLOCK + tid] = v.w;
            VP[j] = ~phantom_mask<NW>(t.m, j);
            VN[j] = 0;
        }
        if (valid) {
            const uint32_t meta = rv.meta[(uint64_t)read_idx * rv.n_orient + slot];
            t.presence = meta & 0xffffu;
            t.last_sym = last_symbol >= 0 ? (uint32_t)last_symbol : ((meta >> 16) & 15u);
        }
    }

    t.min_ed = t.m;
    t.min_pos = 0;
    t.curr_ed = t.m;
    t.stopped = !valid;

    // ---- sentinel iterations of the skewed variants (j = 1..wn): the score lane still sees
    // "no symbol", only the bookkeeping runs (LevMyers256.hpp:120-122, 206-234;
    // LevMyers128.hpp:115, 199-221)
    if (valid && t.rule != RULE_M64) {
        for (uint32_t j = 1; j <= t.wn && j <= t.n_cols + t.wn; ++j) {
            if (t.rule == RULE_M256)
                t.min_pos = j - t.wn;
            else if (t.min_pos + 1 == j - t.wn)
                t.min_pos = j - t.wn;
            if (t.curr_ed + j > t.exit_t + t.wn) {   // curr_ed > W + 1 - j + k
                t.stopped = true;
                break;
            }
        }
    }
    if (valid && (t.stopped || t.n_cols == 0)) {
        // finished before the first text column
        results[gid].edit_distance = (clamp && t.min_ed > t.k) ? t.k + 1u : t.min_ed;
        results[gid].pos = t.min_ed > t.k ? 0u : t.min_pos;
        valid = false;
        t.stopped = true;
    }

    run_columns<NW, BLOCK, false>(g, t, valid, VP, VN, pm_lds, tid, rv.codes + rv.code_off[read_idx], ref_pos, 0u, 0u);

    if (valid) {
        results[gid].edit_distance = (clamp && t.min_ed > t.k) ? t.k + 1u : t.min_ed;
        results[gid].pos = t.min_ed > t.k ? 0u : t.min_pos;
    }
}

// ===========================================================================================
// Banded kernel (accepted-only mode): the same answers for every candidate the reference
// accepts (edit_distance <= k: distance and end column with the variant's tie-break), k + 1
// for the rest - which is all the mapping stages read (mapping_core_sa.cpp:237-241, 289).
//
// A candidate can only be accepted through cells on the diagonals -k .. n_cols - m + k of its
// matrix (a path of cost <= k starts on a diagonal >= 0, ends on one in
// [-k, n_cols - m], and leaving that range on either side costs more than k to come back).
// When those fit in 32 NB rows, only a band of 32 NB rows is computed:
//   * columns 1 .. FS ("sliding"): the frame moves down one row per column (Hyyro's band form:
//     D0 shifts right instead of HP / HN shifting left), its bottom reaching the read's last
//     row after column FS.  Cells outside count as infinity, so every value is an upper bound
//     and exact wherever the true value is <= k.  The value under the frame's bottom-left
//     corner is known (D[r][0] = r) and is followed down the diagonal by counting D0's top bit;
//   * columns FS + 1 .. n_cols ("fixed"): the frame stays on the read's last 32 NB rows and the
//     ordinary recurrence runs there (run_columns with NW = NB), the unknown row above coming
//     in with a horizontal delta of +1 (an upper bound again).  The score row is the top bit,
//     so the reference's bookkeeping is the exact kernel's, fed with upper bounds: the first
//     column that reaches the final minimum and the ties after it are cells of value <= k,
//     hence the true ones; the early exit may fire sooner but only where the true value could
//     not come back under k either.
// pm32 is top-aligned, so the frame of column f + 1 starts at bit 32 words32 - FS + f of a
// row whatever the read's length: dword index and funnel shift are wave-uniform.  FS is the
// wave's maximum of n_cols + k - 32 NB, rounded up to a multiple of 8 (a text dword); a lane
// fits when FS + k + 2 <= m (its accepted end columns then all lie in the fixed phase, one
// diagonal to spare).  Lanes that do not fit, waves that mix tie-break rules, k >= m and
// oversized windows are left to the next kernel of the chain (RESULT_DEFERRED).
// ===========================================================================================

template <int NB>
__device__ __forceinline__ uint32_t slide_core(const uint32_t (&eq)[NB], uint32_t (&VP)[NB], uint32_t (&VN)[NB])
{
    uint32_t x[NB], d0[NB], hp[NB], hn[NB], d0s[NB];
#pragma unroll
    for (int w = 0; w < NB; ++w)
        x[w] = v_and(eq[w], VP[w]);
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < NB; ++w) {
        uint32_t cout;
        x[w] = __builtin_addc(x[w], VP[w], carry, &cout);
        carry = cout;
    }
#pragma unroll
    for (int w = 0; w < NB; ++w)
        x[w] = v_bitop3<TT_XOR_OR>(x[w], VP[w], eq[w]);
#pragma unroll
    for (int w = 0; w < NB; ++w)
        d0[w] = v_or(x[w], VN[w]);
#pragma unroll
    for (int w = 0; w < NB; ++w)
        hp[w] = v_bitop3<TT_OR_NOR>(VN[w], d0[w], VP[w]);
#pragma unroll
    for (int w = 0; w < NB; ++w)
        hn[w] = v_and(d0[w], VP[w]);
#pragma unroll
    for (int w = 0; w < NB; ++w)
        d0s[w] = w < NB - 1 ? __builtin_amdgcn_alignbit(d0[w < NB - 1 ? w + 1 : w], d0[w], 1) : v_lshr<1>(d0[w]);
#pragma unroll
    for (int w = 0; w < NB; ++w)
        VP[w] = v_bitop3<TT_OR_NOR>(hn[w], d0s[w], hp[w]);   // vertical deltas in the NEXT column's frame
#pragma unroll
    for (int w = 0; w < NB; ++w)
        VN[w] = v_and(d0s[w], hp[w]);
    return d0[NB - 1];
}

// pattern-match bits of the frame for symbol U of the group: NB + 1 dwords, funnelled by the
// wave-uniform shift.  The first Z of them lie below the vector's dword 0 - phantom rows, all
// ones - and are not fetched; the others come from rows 0.. of grp_addr.
template <int NB, int BLOCK, int U, int Z>
__device__ __forceinline__ void slide_load(uint32_t grp_addr, uint32_t syms, uint32_t symsA, uint32_t rowmask,
                                           uint32_t (&raw)[NB + 1])
{
    typedef const __attribute__((address_space(3))) uint32_t* lds_ptr;
    constexpr int RSH = BLOCK == 256 ? 10 : (BLOCK == 128 ? 9 : 8);
    const uint32_t srow = U < 5 ? v_lshr<(U < 5 ? 12 + 4 * U - RSH : 0)>(symsA) : v_lshr<(U < 5 ? 0 : 4 * U - RSH)>(syms);
    const uint32_t addr = v_bitop3<TT_OR_AND>(grp_addr, srow, rowmask);
#pragma unroll
    for (int w = 0; w <= NB; ++w)
        raw[w] = w < Z ? ~0u : *(lds_ptr)(uintptr_t)(addr + (uint32_t)(w < Z ? 0 : w - Z) * (4u * BLOCK * 4u));
}

template <int NB, int BLOCK, int U, int Z>
__device__ __forceinline__ void slide_column(uint32_t grp_addr, uint32_t syms, uint32_t symsA, uint32_t rowmask,
                                             uint32_t s0, uint32_t (&raw)[NB + 1], uint32_t (&VP)[NB],
                                             uint32_t (&VN)[NB], uint32_t& cnt)
{
    uint32_t nxt[NB + 1];
    if (U < 7)
        slide_load<NB, BLOCK, (U < 7 ? U + 1 : 0), Z>(grp_addr, syms, symsA, rowmask, nxt);
    uint32_t eq[NB];
#pragma unroll
    for (int w = 0; w < NB; ++w)
        eq[w] = w + 1 < Z ? ~0u : __builtin_amdgcn_alignbit(raw[w + 1], raw[w], s0 + (uint32_t)U);
    const uint32_t d0 = slide_core<NB>(eq, VP, VN);
    cnt = v_add(cnt, v_lshr<31>(d0));
    if (U < 7) {
#pragma unroll
        for (int w = 0; w <= NB; ++w)
            raw[w] = nxt[w];
    }
}

template <int NB, int BLOCK, int Z>
__device__ __forceinline__ void slide_group(uint32_t grp_addr, uint32_t syms, uint32_t rowmask, uint32_t s0,
                                            uint32_t (&VP)[NB], uint32_t (&VN)[NB], uint32_t& cnt)
{
    const uint32_t symsA = syms << 12;
    uint32_t raw[NB + 1];
    slide_load<NB, BLOCK, 0, Z>(grp_addr, syms, symsA, rowmask, raw);
    slide_column<NB, BLOCK, 0, Z>(grp_addr, syms, symsA, rowmask, s0, raw, VP, VN, cnt);
    slide_column<NB, BLOCK, 1, Z>(grp_addr, syms, symsA, rowmask, s0, raw, VP, VN, cnt);
    slide_column<NB, BLOCK, 2, Z>(grp_addr, syms, symsA, rowmask, s0, raw, VP, VN, cnt);
    slide_column<NB, BLOCK, 3, Z>(grp_addr, syms, symsA, rowmask, s0, raw, VP, VN, cnt);
    slide_column<NB, BLOCK, 4, Z>(grp_addr, syms, symsA, rowmask, s0, raw, VP, VN, cnt);
    slide_column<NB, BLOCK, 5, Z>(grp_addr, syms, symsA, rowmask, s0, raw, VP, VN, cnt);
    slide_column<NB, BLOCK, 6, Z>(grp_addr, syms, symsA, rowmask, s0, raw, VP, VN, cnt);
    slide_column<NB, BLOCK, 7, Z>(grp_addr, syms, symsA, rowmask, s0, raw, VP, VN, cnt);
}

// PACK (reads up to 256 symbols): the lanes a block still needs are moved together once the
// band has shown which candidates are hopeless.  An accepted candidate has a path of cost <= k inside the band, and
// from column n_cols - m + k + 1 on every such path has left row 0 and crosses each column in one of the frame's
// cells: a column whose 64 band values all exceed k ends the candidate ({k + 1, 0}; while paths may still start the
// frame holds a phantom row of value 0 and the test cannot fire).  The test runs once the frame's top row has 4 k read
// symbols above it (unrelated text collects 0.3 errors a row: every cell is above k by then; column 96 of S1's 173), or
// at the end of the sliding phase if that comes sooner.  It is a bound, not the minimum: the column's values at every
// eighth row are exact (population counts of the vertical deltas), between two of them a walk of eight steps of +-1
// cannot dip more than (8 - |difference|) / 2 below the smaller.  The survivors' state - ten words - goes through the
// LDS the rows occupied into the block's first lanes, which re-stage their new reads' rows, and waves left without a
// lane leave.  Barriers all outside wave-divergent code; FS and the tie-break rule are the block's, not the wave's, so
// that lanes can change waves.
constexpr uint32_t PACK_MIN = 64;   // no test before this column: too few rows for it to say anything

template <int NB, int BLOCK, bool PACK>
__global__ __launch_bounds__(BLOCK) void myers_band_kernel(
    GenomeView g, ReadsView rv, const wamd_dist_task* __restrict__ tasks, uint32_t n, int last_symbol,
    uint32_t max_text_len, wamd_dist_result* __restrict__ results, uint32_t* __restrict__ flags,
    int chained, const uint32_t* __restrict__ only_deferred, uint32_t* __restrict__ deferred,
    const uint32_t* __restrict__ n_dev)
{
    // [the read's words32 dwords][A,C,G,T][lane], as in the exact kernel
    extern __shared__ uint32_t pm_band[];
    constexpr uint32_t ROW = 4u * BLOCK;   // dwords per vector dword
    constexpr int WAVES = BLOCK / 64;
    // the block's few shared words sit BEHIND the rows (PACK launches bring 64 bytes more): the rows must start at LDS
    // address 0, the symbol's row offset is OR-ed into a lane's address (load_eq, slide_load)
    uint32_t* const sh_fs = pm_band + rv.words32 * ROW;
    uint32_t* const sh_rule = sh_fs + WAVES;
    uint32_t* const sh_cnt = sh_rule + WAVES;
    uint32_t* const sh_at = sh_cnt + WAVES;

    if (only_deferred && *only_deferred == 0)
        return;
    if (n_dev) {
        n = min(n, *n_dev);
        if (blockIdx.x * BLOCK >= n)
            return;
    }
    const uint32_t tid = threadIdx.x;
    uint32_t gid = blockIdx.x * BLOCK + tid;
    bool valid = gid < n;
    if (valid && chained && results[gid].edit_distance != RESULT_DEFERRED)
        valid = false;
    if (!PACK && chained && !__any(valid))
        return;

    Lane t{};
    uint32_t read_idx = 0, ref_pos = 0;
    if (valid) {
        const wamd_dist_task task = tasks[gid];
        const uint32_t err = decode_task(task, rv, g, max_text_len, 32u * rv.words32, t);
        read_idx = task.read_idx;
        ref_pos = task.ref_pos;
        if (task.read_idx == WAMD_READ_NONE) {   // a slot without a task: skipped, no flag
            results[gid].edit_distance = RESULT_ERROR;
            results[gid].pos = 0;
            valid = false;
        } else if (err) {
            raise_flag(flags, err);
            results[gid].edit_distance = RESULT_ERROR;
            results[gid].pos = err;
            valid = false;
        }
    }
    if (!valid) {
        t = Lane{};
        t.m = 1;
        ref_pos = 0;
        read_idx = 0;
    }
    // no end column can come under k
    if (valid && t.n_cols + t.k < t.m) {
        results[gid].edit_distance = t.k + 1u;
        results[gid].pos = 0;
        valid = false;
    }

    // ---- does the lane fit the band, and where does the wave's (PACK: the block's) fixed phase start ----
    const uint32_t fm_own = t.n_cols + t.k > 32u * NB ? t.n_cols + t.k - 32u * NB : 0u;
    const bool selffit = valid && t.k < t.m && fm_own + t.k + 2u <= t.m && t.m + t.n_cols < 32000u;
    uint32_t FS = selffit ? fm_own : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        FS = max(FS, (uint32_t)__shfl_xor((int)FS, o));
    FS = __builtin_amdgcn_readfirstlane((FS + 7u) & ~7u);
    const uint64_t cand = __ballot(selffit);
    uint32_t rule0 = __builtin_amdgcn_readfirstlane(__shfl((int)t.rule, cand ? __ffsll((long long)cand) - 1 : 0));
    int32_t test_at = INT32_MIN;   // PACK: the column of the test, relative to FS (the wave's)
    if (PACK) {
        // the frame's top row at column f is row m - FS - 32 NB + f: it reaches 4 k at f = FS + (32 NB - m + 4 k)
        int32_t a = selffit ? 32 * NB - (int32_t)t.m + 4 * (int32_t)t.k : INT32_MIN;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            a = max(a, __shfl_xor(a, o));
        test_at = __builtin_amdgcn_readfirstlane(a);
    }
    const bool fit = selffit && t.rule == rule0 && FS + t.k + 2u <= t.m;
    if (valid && !fit) {
        results[gid].edit_distance = RESULT_DEFERRED;
        results[gid].pos = 0;
    }
    {
        // the next kernel only asks "any?": one flag, written by the waves that still read 0
        // (an atomic per wave on one address would serialise the whole launch)
        const uint64_t left = __ballot(valid && !fit);
        if (deferred && left && (tid & 63u) == 0 && *reinterpret_cast<volatile uint32_t*>(deferred) == 0)
            *reinterpret_cast<volatile uint32_t*>(deferred) = 1u;
    }
    valid = fit;
    if (!PACK && !__any(valid))   // wave-uniform; without PACK the block has no barriers
        return;

    // ---- rows -> LDS ----------------------------------------------------------------------------
    const uint32_t W32 = rv.words32;
    const uint8_t* my_codes = rv.codes;
    auto stage_rows = [&]() {   // this lane's rows into its column of pm_band, its read's facts into t
        const uint32_t slot = valid ? (uint32_t)(t.rev ? rv.slot_rc : rv.slot_direct) : 0;
        const uint4* src = reinterpret_cast<const uint4*>(rv.pm32) + ((uint64_t)read_idx * rv.n_orient + slot) * rv.pm32_stride;
        for (uint32_t d = 0; d < W32; ++d) {
            uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
            if (valid)
                v = src[d];
            pm_band[(d * 4 + 0) * BLOCK + tid] = v.x;
            pm_band[(d * 4 + 1) * BLOCK + tid] = v.y;
            pm_band[(d * 4 + 2) * BLOCK + tid] = v.z;
            pm_band[(d * 4 + 3) * BLOCK + tid] = v.w;
        }
        if (valid) {
            const uint32_t meta = rv.meta[(uint64_t)read_idx * rv.n_orient + slot];
            t.presence = meta & 0xffffu;
            t.last_sym = last_symbol >= 0 ? (uint32_t)last_symbol : ((meta >> 16) & 15u);
        }
        my_codes = rv.codes + rv.code_off[read_idx];
    };
    if (!PACK || __any(valid))
        stage_rows();

    // frame of column f (0-based): bit b is read symbol f + Z0 + b; symbols < 0 are phantom rows
    int32_t Z0 = (int32_t)t.m - 32 * NB - (int32_t)FS;
    uint32_t VP[NB], VN[NB];
#pragma unroll
    for (int w = 0; w < NB; ++w) {
        const int32_t ph = -Z0 - 32 * w;   // phantom bits of this dword
        VP[w] = ph <= 0 ? ~0u : (ph >= 32 ? 0u : ~((1u << ph) - 1u));
        VN[w] = 0;
    }

    // ---- sliding phase: columns 1 .. FS ---------------------------------------------------------
    uint32_t cnt = 0;   // D0 bits seen under the frame's bottom row
    const uint32_t rowmask = in_vgpr(3u << (BLOCK == 256 ? 10 : (BLOCK == 128 ? 9 : 8)));
    const uint32_t lane_addr =
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)pm_band + tid * 4u;
    // columns from + 1 .. to of the sliding phase (both multiples of 8), with a text stream of their own
    auto slide = [&](const uint32_t from, const uint32_t to) {
        const uint32_t shift = (ref_pos & 7u) * 4u;
        uint32_t didx = (ref_pos >> 3) + (from >> 3);
        uint32_t R[TEXT_CHUNK + 1], NX[TEXT_CHUNK];
        load_text_first(g, didx, R);
        didx += TEXT_CHUNK + 1;
        for (uint32_t base = from; base < to;) {
            if (base + 8u * TEXT_CHUNK < to)   // another chunk of sliding columns follows
                load_text_chunk(g, didx, NX);
            didx += TEXT_CHUNK;
            for (int gi = 0; gi < TEXT_CHUNK && base < to; ++gi, base += 8) {
                const uint32_t syms = __builtin_amdgcn_alignbit(R[1], R[0], shift);
#pragma unroll
                for (int c = 0; c < TEXT_CHUNK; ++c)
                    R[c] = R[c + 1];
                // frame start as a bit offset into the vector (negative: still among the phantom rows
                // below its dword 0): wave-uniform, a multiple of 8
                const int32_t q = (int32_t)(32u * (W32 - NB)) - (int32_t)FS + (int32_t)base;
                const int32_t qi = q >> 5;
                const uint32_t s0 = (uint32_t)q & 31u;
                if (!__any(valid && (syms & 0xccccccccu))) {
                    const uint32_t grp_addr = lane_addr + (uint32_t)(qi < 0 ? 0 : qi) * (ROW * 4u);
                    const int z = qi < 0 ? -qi : 0;   // dwords of the frame that lie below dword 0
                    if (z == 0)
                        slide_group<NB, BLOCK, 0>(grp_addr, syms, rowmask, s0, VP, VN, cnt);
                    else if (z == 1)
                        slide_group<NB, BLOCK, 1>(grp_addr, syms, rowmask, s0, VP, VN, cnt);
                    else if (z == 2 || NB < 3)
                        slide_group<NB, BLOCK, 2>(grp_addr, syms, rowmask, s0, VP, VN, cnt);
                    else
                        slide_group<NB, BLOCK, (NB < 3 ? 2 : 3)>(grp_addr, syms, rowmask, s0, VP, VN, cnt);
                    continue;
                }
                // ---- a text symbol >= 4 somewhere in the wave: spelled out per lane -----------------
#pragma unroll 1
                for (uint32_t u = 0; u < 8; ++u) {
                    const uint32_t f = base + u;
                    const uint32_t c = (syms >> (4 * u)) & 15u;
                    const int32_t first = (int32_t)f + Z0;   // read symbol under bit 0
                    const bool exotic = valid && c >= 4u && ((t.presence >> c) & 1u);
                    uint32_t eq[NB];
#pragma unroll
                    for (int w = 0; w < NB; ++w) {
                        const int32_t r0 = qi + w, r1 = qi + w + 1;
                        const uint32_t a0 = r0 < 0 ? ~0u : pm_band[((uint32_t)r0 * 4 + (c & 3u)) * BLOCK + tid];
                        const uint32_t a1 = r1 < 0 ? ~0u : pm_band[((uint32_t)r1 * 4 + (c & 3u)) * BLOCK + tid];
                        uint32_t v = __builtin_amdgcn_alignbit(a1, a0, s0 + u);
                        if (c >= 4u) {
                            const int32_t ph = -first - 32 * w;   // phantom rows match everything
                            v = ph <= 0 ? 0u : (ph >= 32 ? ~0u : ((1u << ph) - 1u));
                            if (exotic) {
                                for (int32_t b = 0; b < 32; ++b) {
                                    const int32_t i = first + 32 * w + b;
                                    if (i < 0 || i >= (int32_t)t.m)
                                        continue;
                                    const uint32_t rcode = t.rev ? rc_code(packed_symbol(my_codes, t.m - 1 - (uint32_t)i))
                                                                 : packed_symbol(my_codes, (uint32_t)i);
                                    v |= (uint32_t)(rcode == c) << b;
                                }
                            }
                        }
                        eq[w] = v;
                    }
                    cnt += slide_core<NB>(eq, VP, VN) >> 31;
                }
            }
#pragma unroll
            for (int c = 0; c < TEXT_CHUNK; ++c)
                R[c + 1] = nibble_swap(NX[c]);
        }
    };

    // where the hopeless are told from the rest (the wave's own column)
    uint32_t T1 = FS;
    if (PACK && FS >= PACK_MIN) {
        const int32_t want = test_at == INT32_MIN ? (int32_t)FS : (((int32_t)FS + test_at + 7) & ~7);
        T1 = want >= (int32_t)FS ? FS : (want < (int32_t)PACK_MIN ? PACK_MIN : (uint32_t)want);
    }
    if (T1 && (!PACK || __any(valid)))
        slide(0u, T1);

    if (PACK) {   // (every wave of the block comes through here, with lanes or without: the barriers are the block's)
        // ---- the smallest value in the frame of column T1, from its bottom cell (row m - FS - 1 + T1: that value is
        // the corner's plus one per column, minus the free diagonal steps counted in cnt) up the vertical deltas; VP / VN
        // are held in the next column's frame, one row down: bit b is the step into row bottom + b - 62
        if (valid && FS >= PACK_MIN) {
            int32_t cur = (int32_t)(t.m - FS - 1u + T1) - (int32_t)cnt, mn = cur;
            // eight rows at a time, bottom up: bits 8 s - 1 .. 8 s + 6 are the steps between rows bottom - 8 (7 - s) - 8
            // and bottom - 8 (7 - s); the topmost stretch has seven
#pragma unroll
            for (int sgm = 4 * NB - 1; sgm >= 0; --sgm) {
                const int lo = sgm ? 8 * sgm - 1 : 0, hi = 8 * sgm + 6;   // bit range of the stretch, inside one dword or across two
                uint32_t vp, vn;
                if ((lo >> 5) == (hi >> 5)) {
                    const uint32_t msk = ((hi - lo + 1) >= 32 ? ~0u : ((1u << (hi - lo + 1)) - 1u)) << (lo & 31);
                    vp = VP[lo >> 5] & msk;
                    vn = VN[lo >> 5] & msk;
                } else {   // the stretch starts on a dword's top bit
                    vp = (VP[lo >> 5] >> 31) | ((VP[hi >> 5] & 0x7fu) << 1);
                    vn = (VN[lo >> 5] >> 31) | ((VN[hi >> 5] & 0x7fu) << 1);
                }
                const int32_t nxt = cur + (int32_t)__popc(vn) - (int32_t)__popc(vp);
                const int32_t len = hi - lo + 1, diff = nxt > cur ? nxt - cur : cur - nxt;
                const int32_t low = (nxt < cur ? nxt : cur) - (len - diff) / 2;
                mn = low < mn ? low : mn;
                cur = nxt;
            }
            if (mn > (int32_t)t.k) {
                results[gid].edit_distance = t.k + 1u;
                results[gid].pos = 0;
                valid = false;
            }
        }
        // ---- the block's survivors into its first lanes ----------------------------------------------
        // FS, the test column and the tie-break rule are each wave's own (no barrier before the columns start); lanes
        // can only change waves where the block's waves agree on them - they do in a batch of one read length and k
        const uint64_t live = __ballot(valid);
        if ((tid & 63u) == 0) {
            sh_cnt[tid >> 6] = (uint32_t)__popcll(live);
            sh_fs[tid >> 6] = FS;
            sh_at[tid >> 6] = T1;
            sh_rule[tid >> 6] = rule0;
        }
        __syncthreads();   // (every wave is also through with the rows it read so far)
        uint32_t total = 0, before = 0, waves_now = 0;
        bool agree = true;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = sh_cnt[w];
            before += (uint32_t)w < (tid >> 6) ? c : 0u;
            total += c;
            waves_now += c ? 1u : 0u;
            // (a wave without a lane left has nothing to agree on)
            agree = agree && (c == 0 || (sh_fs[w] == FS && sh_at[w] == T1 && sh_rule[w] == rule0) || live == 0);
        }
        // every wave reaches the same verdict: a wave with lanes compares the others with itself, and two waves with
        // lanes that differ see each other; a wave without lanes only needs to know whether to stay for the barriers
        // (the 96-row band's state is half as large again and its test comes later: one wave freed does not pay there)
        constexpr uint32_t MUST_FREE = NB >= 3 ? 2u : 1u;
        bool go = agree && (total + 63u) / 64u + MUST_FREE <= waves_now;
        if (live == 0) {   // an empty wave takes the verdict of the waves that have lanes: all pairs among them agree?
            bool all_same = true;
            uint32_t f0 = 0, a0 = 0, r0 = 0;
            bool have = false;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                if (sh_cnt[w]) {
                    if (!have) {
                        f0 = sh_fs[w];
                        a0 = sh_at[w];
                        r0 = sh_rule[w];
                        have = true;
                    } else {
                        all_same = all_same && sh_fs[w] == f0 && sh_at[w] == a0 && sh_rule[w] == r0;
                    }
                }
            }
            go = all_same && (total + 63u) / 64u + MUST_FREE <= waves_now;
            if (go) {   // it may be given lanes: it needs the block's FS / rule / test column
                FS = __builtin_amdgcn_readfirstlane(f0);
                T1 = __builtin_amdgcn_readfirstlane(a0);
                rule0 = __builtin_amdgcn_readfirstlane(r0);
            }
        }
        if (go) {   // block-uniform: the waves agree and the move frees at least one of them
            constexpr uint32_t F_GID = 0, F_READ = 1, F_REF = 2, F_LEN = 3, F_K = 4, F_CNT = 5, F_VP = 6, F_VN = 6 + NB;
            static_assert(6 + 2 * NB <= 12, "the state must fit the rows of a three-dword read");
            if (valid) {
                const uint32_t p = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0));
                pm_band[F_GID * BLOCK + p] = gid;
                pm_band[F_READ * BLOCK + p] = read_idx;
                pm_band[F_REF * BLOCK + p] = ref_pos;
                pm_band[F_LEN * BLOCK + p] = t.m | (t.n_cols << 16);
                pm_band[F_K * BLOCK + p] = t.k | ((uint32_t)t.rev << 16);
                pm_band[F_CNT * BLOCK + p] = cnt;
#pragma unroll
                for (int w = 0; w < NB; ++w) {
                    pm_band[(F_VP + w) * BLOCK + p] = VP[w];
                    pm_band[(F_VN + w) * BLOCK + p] = VN[w];
                }
            }
            __syncthreads();
            valid = tid < total;
            if (valid) {
                gid = pm_band[F_GID * BLOCK + tid];
                read_idx = pm_band[F_READ * BLOCK + tid];
                ref_pos = pm_band[F_REF * BLOCK + tid];
                const uint32_t ln = pm_band[F_LEN * BLOCK + tid], kk = pm_band[F_K * BLOCK + tid];
                t.m = ln & 0xffffu;
                t.n_cols = ln >> 16;
                t.k = kk & 0xffffu;
                t.rev = (kk >> 16) & 1u;
                t.rule = rule0;
                t.wn = (t.m - 1) / 64;
                t.exit_t = t.n_cols + t.k;   // W + 1 - wn + k of the skewed variants; M64 does not read it
                cnt = pm_band[F_CNT * BLOCK + tid];
#pragma unroll
                for (int w = 0; w < NB; ++w) {
                    VP[w] = pm_band[(F_VP + w) * BLOCK + tid];
                    VN[w] = pm_band[(F_VN + w) * BLOCK + tid];
                }
                Z0 = (int32_t)t.m - 32 * NB - (int32_t)FS;
            } else {
                t = Lane{};
                t.m = 1;
                ref_pos = 0;
                read_idx = 0;
            }
            __syncthreads();   // everybody has its new state: the rows may go over it
            if (__any(valid))
                stage_rows();
        }
        if (!__any(valid))   // wave-uniform, and the last barrier is behind
            return;
        if (T1 < FS)
            slide(T1, FS);
    }

    // ---- fixed phase ----------------------------------------------------------------------------
    // bottom cell of column FS's frame (row m - 1) = (m - FS - 1) + FS - cnt; the row below it,
    // row m, came in through VP's top bit
    t.min_ed = t.m;
    t.min_pos = 0;
    t.curr_ed = FS ? t.m - 1u - cnt + (VP[NB - 1] >> 31) : t.m;
    t.stopped = !valid;
    const uint32_t hin = t.m > 32u * NB ? ~0u : 0u;
    run_columns<NB, BLOCK, true>(g, t, valid, VP, VN, pm_band + (W32 - NB) * ROW, tid, my_codes, ref_pos, FS, hin);

    if (valid) {
        results[gid].edit_distance = t.min_ed > t.k ? t.k + 1u : t.min_ed;
        results[gid].pos = t.min_ed > t.k ? 0u : t.min_pos;
    }
}

template <int NW, int BLOCK>
static int launch_one(const GenomeView& g, const ReadsView& rv, const wamd_dist_task* d_tasks,
                      uint32_t n, int last_symbol, uint32_t max_text_len, wamd_dist_result* d_results,
                      uint32_t* d_flags, int chained, const uint32_t* only_deferred, int clamp, const uint32_t* n_dev,
                      hipStream_t s)
{
    const uint32_t blocks = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL((myers_distance_kernel<NW, BLOCK>), dim3(blocks), dim3(BLOCK), 0, s, g, rv,
                       d_tasks, n, last_symbol, max_text_len, d_results, d_flags, chained, only_deferred, clamp, n_dev);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

template <int NB>
static int launch_band(const GenomeView& g, const ReadsView& rv, const wamd_dist_task* d_tasks, uint32_t n,
                       int last_symbol, uint32_t max_text_len, wamd_dist_result* d_results, uint32_t* d_flags,
                       int chained, const uint32_t* only_deferred, uint32_t* deferred, const uint32_t* n_dev, hipStream_t s)
{
#ifndef WAMD_BAND_BLOCK
#define WAMD_BAND_BLOCK 128
#endif
    constexpr int BLOCK = WAMD_BAND_BLOCK;   // small blocks: the LDS rows are what limits the waves per CU
    {
        // reads up to 256 symbols: blocks of four waves that pack their survivors
#ifndef WAMD_NO_PACK   // (scripts/dist_variants.sh builds the comparison without it)
        if (rv.words32 <= 8) {
            constexpr int PB = 256;
            const uint32_t blocks = (n + PB - 1) / PB;
            const size_t lds = (size_t)rv.words32 * 4 * PB * sizeof(uint32_t) + 64;
            hipLaunchKernelGGL((myers_band_kernel<NB, PB, true>), dim3(blocks), dim3(PB), lds, s, g, rv, d_tasks, n,
                               last_symbol, max_text_len, d_results, d_flags, chained, only_deferred, deferred, n_dev);
            WAMD_HIP(hipGetLastError());
            return WAMD_OK;
        }
#endif
    }
    const uint32_t blocks = (n + BLOCK - 1) / BLOCK;
    const size_t lds = (size_t)rv.words32 * 4 * BLOCK * sizeof(uint32_t);
    hipLaunchKernelGGL((myers_band_kernel<NB, BLOCK, false>), dim3(blocks), dim3(BLOCK), lds, s, g, rv, d_tasks, n,
                       last_symbol, max_text_len, d_results, d_flags, chained, only_deferred, deferred, n_dev);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

static int launch_exact(const GenomeView& g, const ReadsView& rv, const wamd_dist_task* d_tasks, uint32_t n,
                        int last_symbol, uint32_t mt, wamd_dist_result* d_results, uint32_t* f, int chained,
                        const uint32_t* only_deferred, int clamp, const uint32_t* n_dev, hipStream_t s)
{
#define WAMD_DIST_CASE(NW, BLOCK) \
    if (rv.words32 <= NW)         \
        return launch_one<NW, BLOCK>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, chained, only_deferred, clamp, n_dev, s);
    WAMD_DIST_CASE(1, 256)
    WAMD_DIST_CASE(2, 256)
    WAMD_DIST_CASE(3, 256)
    WAMD_DIST_CASE(4, 256)
    WAMD_DIST_CASE(5, 256)
    WAMD_DIST_CASE(6, 256)
    WAMD_DIST_CASE(7, 256)
    WAMD_DIST_CASE(8, 256)
    WAMD_DIST_CASE(10, 64)
    WAMD_DIST_CASE(12, 64)
    WAMD_DIST_CASE(16, 64)
    WAMD_DIST_CASE(24, 64)
    WAMD_DIST_CASE(32, 64)
#undef WAMD_DIST_CASE
    set_error("reads of %u dwords are not supported", rv.words32);
    return WAMD_ERR_TOO_LONG;
}

// The pair of "something was deferred" flags of a stream: a stream keeps the pair it was given at
// its first accepted-only launch (launches on one stream are ordered, so they may share it).
// nullptr once BAND_SLOTS streams have been seen: the chain then runs without the early-out.
uint32_t* band_flags(wamd_ctx* ctx, hipStream_t s)
{
    while (__sync_lock_test_and_set(&ctx->band_lock, 1)) {
    }
    int slot = -1;
    for (int i = 0; i < ctx->band_streams; ++i)
        if (ctx->band_stream[i] == (void*)s)
            slot = i;
    if (slot < 0 && ctx->band_streams < BAND_SLOTS) {
        slot = ctx->band_streams++;
        ctx->band_stream[slot] = (void*)s;
    }
    __sync_lock_release(&ctx->band_lock);
    return slot < 0 ? nullptr : ctx->d_flags + 4 + 2 * slot;
}

int launch_distance(wamd_ctx* ctx, const wamd_reads* r, const wamd_dist_task* d_tasks, uint32_t n,
                    int last_symbol, wamd_dist_result* d_results, int mode, hipStream_t s, const uint32_t* n_dev)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0)
        return WAMD_OK;
    // WAMD_LAST_SYMBOL_FROM_READ is what preprocess() leaves behind (LevMyers64.cpp:182, 192).  preprocessRawSeq() never
    // writes last_symbol, so a batch that came in as ASCII gets what a fresh object holds: 0 (LevMyers64.cpp:35)
    if (last_symbol < 0 && r->encoding == WAMD_READS_ASCII)
        last_symbol = 0;
    const GenomeView g = genome_view(ctx);
    const ReadsView rv = r->view();
    const uint32_t mt = ctx->max_text_len;
    uint32_t* f = ctx->d_flags;
    // the LDS rows of the banded kernels must fit a block: words32 x 2 KB <= 64 KB
    if (mode != WAMD_DISTANCE_ACCEPTED_ONLY || rv.words32 <= 2 || rv.words32 > 32)
        return launch_exact(g, rv, d_tasks, n, last_symbol, mt, d_results, f, 0, nullptr,
                            mode == WAMD_DISTANCE_ACCEPTED_ONLY, n_dev, s);
    // 64-row band, then 96 rows for what did not fit, then whole columns
    uint32_t* cnt2 = band_flags(ctx, s);
    uint32_t* cnt3 = cnt2 ? cnt2 + 1 : nullptr;
    if (cnt2)
        WAMD_HIP(hipMemsetAsync(cnt2, 0, 2 * sizeof(uint32_t), s));
    int rc = launch_band<2>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, 0, nullptr, cnt2, n_dev, s);
    if (rc != WAMD_OK)
        return rc;
    const uint32_t* last = cnt2;
    if (rv.words32 > 3) {
        rc = launch_band<3>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, 1, cnt2, cnt3, n_dev, s);
        if (rc != WAMD_OK)
            return rc;
        last = cnt3;
    }
    return launch_exact(g, rv, d_tasks, n, last_symbol, mt, d_results, f, 1, last, 1, n_dev, s);
}

} // namespace wamd
