// indel_match.hip - one long indel / clipping around an exact anchor, one candidate per lane.
//
// Replaces CIndelMatching::Match (mapper/indels.cpp:125-461), called for every SA
// candidate of the sensitive stages (mapper/mapping_core_sa.cpp:264/317).  It is not a
// DP: raw mismatch counts and the best clipping point on either side of the anchor, then
// 7-mer seeded candidates for a single long indel per side (one-hot popcount signature,
// at most one error), each scored by the best split of prefix / suffix mismatch counts,
// and finally eight combinations tried in a fixed order with strict `>` on double costs.
//
// MI355X mapping: one wave per 64 candidates, both sequences of a lane staged once in LDS
// as dwords of eight 4-bit symbols ([dword][lane], conflict-free), everything else in
// lane-private registers.  Comparisons run eight symbols at a time: two funnel-shifted
// dwords, an xor, and one flag bit per differing nibble; the loops that have to visit every
// position (running costs in the reference's order, best split) then only peel bits off
// that mask.  The split search keeps running prefix / suffix counts instead of the
// reference's two arrays.  Seed hits are listed and scored in lock step.  Costs are
// evaluated with the reference's expressions in the reference's order (-ffp-contract=off),
// so accepted mappings and their tie-breaks are identical.
#include "wamd_internal.h"

namespace wamd {

constexpr int IM_SIDE = 7;          // min_side_size, indels.h:210
constexpr int IM_SIDE_ERRORS = 1;   // max_no_side_errors
constexpr uint32_t IM_MAX_READ = 256;   // longest read the LDS-staged variant takes
constexpr uint32_t IM_MAX_INDEL = 80;   // largest max_indel_size it takes
constexpr uint32_t IM_HITS = 12;        // seed hits per side and lane that wait for the lock-step pass

struct IndelKernelParams {
    wamd_scoring sc;
    double min_clipped_factor;
};

// rev_comp_code of CIndelMatching (indels.cpp:40-49): 8..15 -> 15
__device__ __forceinline__ uint32_t rc_code_indel(uint32_t c) { return c < 4 ? 3 - c : (c < 8 ? c : 15); }

template <bool STAGED>
struct Seqs {
    // STAGED: nibble dwords in LDS ([dword][lane], symbol s of a dword at bits 4s);
    // otherwise straight from packed global memory
    const uint32_t* lds_g;
    const uint32_t* lds_q;
    const uint8_t* genome;
    const uint8_t* codes;
    uint64_t g0;
    uint32_t m;
    bool rev;
    __device__ __forceinline__ uint32_t G(uint32_t i) const
    {
        if constexpr (STAGED)
            return (lds_g[(i >> 3) * 64] >> (4 * (i & 7u))) & 15u;
        else
            return packed_symbol(genome, g0 + i);
    }
    __device__ __forceinline__ uint32_t Q(uint32_t i) const
    {
        if constexpr (STAGED)
            return (lds_q[(i >> 3) * 64] >> (4 * (i & 7u))) & 15u;
        else
            return i < m ? (rev ? rc_code_indel(packed_symbol(codes, m - 1 - i)) : packed_symbol(codes, i)) : 15u;
    }
    // symbols i .. i+7, symbol i in the low nibble (whatever lies past the sequence comes along:
    // callers mask the positions they mean)
    __device__ __forceinline__ uint32_t G8(uint32_t i) const
    {
        if constexpr (STAGED) {
            return __builtin_amdgcn_alignbit(lds_g[((i >> 3) + 1) * 64], lds_g[(i >> 3) * 64], 4 * (i & 7u));
        } else {
            uint32_t w = 0;
            for (uint32_t t = 0; t < 8; ++t)
                w |= G(i + t) << (4 * t);
            return w;
        }
    }
    __device__ __forceinline__ uint32_t Q8(uint32_t i) const
    {
        if constexpr (STAGED) {
            return __builtin_amdgcn_alignbit(lds_q[((i >> 3) + 1) * 64], lds_q[(i >> 3) * 64], 4 * (i & 7u));
        } else {
            uint32_t w = 0;
            for (uint32_t t = 0; t < 8; ++t)
                w |= Q(i + t) << (4 * t);
            return w;
        }
    }
    // symbols 8d .. 8d+7 of the genome window
    __device__ __forceinline__ uint32_t Gw(uint32_t d) const
    {
        if constexpr (STAGED)
            return lds_g[d * 64];
        else
            return G8(8 * d);
    }
    // bit 4t set when genome symbol gi + t differs from read symbol qi + t
    __device__ __forceinline__ uint32_t mm8(uint32_t gi, uint32_t qi) const
    {
        const uint32_t x = G8(gi) ^ Q8(qi);
        return (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u;
    }
};

__device__ __forceinline__ uint32_t low_flags(uint32_t n)   // flag bits of positions 0 .. n-1, n <= 8
{
    return n >= 8 ? 0x11111111u : (0x11111111u & ((1u << (4 * n)) - 1u));
}

// calc_mismatches_with_indel, indels.cpp:495-563, without the two arrays
template <bool STAGED>
__device__ void split_mismatches(const Seqs<STAGED>& s, uint32_t query_len, uint32_t g_left, uint32_t g_right,
                                 uint32_t q_left, uint32_t q_right, bool left_margin, uint32_t& no_mis,
                                 uint32_t& indel_pos)
{
    const uint32_t text_size = g_right - g_left, query_size = q_right - q_left;
    const uint32_t stats = text_size < query_size ? text_size : query_size;
    if (stats <= (uint32_t)IM_SIDE) {
        no_mis = query_len + 1;
        indel_pos = 0;
        return;
    }
    const uint32_t start = left_margin ? IM_SIDE : 0;
    const uint32_t end = stats - (left_margin ? 0 : IM_SIDE);
    const uint32_t g_r = g_left + text_size - stats, q_r = q_left + query_size - stats;
    uint32_t R = 0;
    for (uint32_t t = 0; t < stats; t += 8)
        R += (uint32_t)__popc(s.mm8(g_r + t, q_r + t) & low_flags(stats - t));
    uint32_t L = 0;
    int best = (int)query_len + 1, best_pos = 0;
    for (uint32_t ib = 0; ib < end; ib += 8) {
        const uint32_t nl = s.mm8(g_left + ib, q_left + ib), nr = s.mm8(g_r + ib, q_r + ib);
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
            const uint32_t i = ib + t;
            if (i < end) {
                if (i >= start && (int)(L + R) < best) {
                    best = (int)(L + R);
                    best_pos = (int)i;
                }
                L += (nl >> (4 * t)) & 1u;
                R -= (nr >> (4 * t)) & 1u;
            }
        }
    }
    no_mis = (uint32_t)best;
    indel_pos = q_left + (uint32_t)best_pos;
}

template <bool STAGED>
__global__ __launch_bounds__(64) void indel_match_kernel(const uint8_t* __restrict__ genome,
                                                         uint64_t genome_symbols, ReadsView rv, int ascii,
                                                         const wamd_indel_task* __restrict__ tasks, uint32_t n,
                                                         IndelKernelParams prm, uint32_t text_dw,
                                                         uint32_t hits_offset, wamd_mapping* __restrict__ results,
                                                         uint32_t* __restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // (text_dw + read dwords) x 64 dwords, then the hit lists
    const uint32_t lane = threadIdx.x;
    const uint32_t gid = blockIdx.x * 64 + lane;
    if (gid >= n)
        return;
    const wamd_indel_task t = tasks[gid];

    wamd_mapping cm;   // a default-constructed candidate_mapping_t, types.h:250-261
    cm.found = 0;
    cm.type = WAMD_MAP_NONE;
    cm.pos = 0;
    cm.no_mismatches = (uint32_t)t.max_no_mismatches + 1;
    cm.matching_len1 = cm.matching_len2 = 0;
    cm.indel_len1 = cm.indel_len2 = 0;
    cm.clipping_len = 0;
    cm.penalty = 0;

    uint32_t err = 0;
    uint32_t query_len = 0;
    if (t.read_idx >= rv.n)
        err = FLAG_READ_IDX;
    else
        query_len = rv.lens[t.read_idx];
    const uint32_t max_indel = t.max_indel_size, fpos = t.fixed_segment_pos, fsize = t.fixed_segment_size;
    const uint32_t max_mis = t.max_no_mismatches, ref_pos = t.ref_pos;
    const uint32_t text_size = query_len + 2 * max_indel;
    if (!err && (fpos + fsize > query_len ||
                 (STAGED && (query_len > IM_MAX_READ || max_indel > IM_MAX_INDEL || (text_size + 2 + 7) / 8 + 1 > text_dw))))
        err = FLAG_ARG;
    if (!err && (ref_pos < max_indel || (uint64_t)ref_pos - max_indel + text_size + 11 > genome_symbols))
        err = FLAG_RANGE;
    if (err) {
        raise_flag(flags, err);
        cm.type = RESULT_ERROR;
        results[gid] = cm;
        return;
    }

    Seqs<STAGED> s;
    s.genome = genome;
    s.codes = rv.codes + rv.code_off[t.read_idx];
    s.g0 = (uint64_t)ref_pos - max_indel;   // prefetch_genome(ref_pos - max_indel_size, text_size)
    s.m = query_len;
    s.rev = t.flags & WAMD_TASK_FLAG_REV_COMP;
    s.lds_g = reinterpret_cast<const uint32_t*>(lds) + lane;
    s.lds_q = reinterpret_cast<const uint32_t*>(lds) + (STAGED ? text_dw * 64 : 0) + lane;
    if constexpr (STAGED) {
        uint32_t* lg = reinterpret_cast<uint32_t*>(lds) + lane;
        uint32_t* lq = lg + text_dw * 64;
        // genome window: dword d = symbols g0 + 8d .. g0 + 8d + 7, funnel-shifted out of the packed genome
        const uint32_t* gd = reinterpret_cast<const uint32_t*>(genome);
        const uint32_t n_gd = (text_size + 2 + 7) / 8 + 1;
        const uint32_t gk = (uint32_t)(s.g0 >> 3), gsh = 4u * ((uint32_t)s.g0 & 7u);
        const uint32_t last_gd = (uint32_t)((genome_symbols - 1) >> 3);
        // 16-byte loads: a wave's load costs the 64 cache lines it touches, whatever its width
        uint32_t lo = nibble_swap(gd[gk]);
        for (uint32_t d = 0; d < n_gd; d += 4) {
            const uint32_t k = gk + d + 1;
            uint32_t b[4];
            if (k + 3u <= last_gd) {
                const wamd_u32x4 v = *reinterpret_cast<const wamd_u32x4_a4*>(gd + k);
                b[0] = v.x, b[1] = v.y, b[2] = v.z, b[3] = v.w;
            } else {
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c)
                    b[c] = gd[k + c < last_gd ? k + c : last_gd];
            }
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
                if (d + c < n_gd) {
                    const uint32_t hi = nibble_swap(b[c]);
                    lg[(d + c) * 64] = __builtin_amdgcn_alignbit(hi, lo, gsh);
                    lo = hi;
                }
            }
        }
        // oriented read; nibbles past its end are 15
        const uint32_t* cd = reinterpret_cast<const uint32_t*>(s.codes);
        const uint32_t n_qd = (query_len + 7) / 8;
        uint32_t fwd[4] = {0, 0, 0, 0};
        for (uint32_t d = 0; d <= n_qd; ++d) {
            uint32_t w = 0xffffffffu;
            if (d < n_qd) {
                if (!s.rev) {
                    if ((d & 3u) == 0) {   // the codes buffer has 64 bytes of slack behind the last read
                        const wamd_u32x4 v = *reinterpret_cast<const wamd_u32x4_a4*>(cd + d);
                        fwd[0] = v.x, fwd[1] = v.y, fwd[2] = v.z, fwd[3] = v.w;
                    }
                    w = nibble_swap((d & 3u) == 0 ? fwd[0] : ((d & 3u) == 1 ? fwd[1] : ((d & 3u) == 2 ? fwd[2] : fwd[3])));
                } else {   // oriented symbol p = rc(stored symbol m - 1 - p)
                    w = 0;
                    for (uint32_t z = 0; z < 8; ++z) {
                        const uint32_t pp = 8 * d + z;
                        if (pp < query_len) {
                            const uint32_t st = query_len - 1 - pp;
                            w |= rc_code_indel((nibble_swap(cd[st >> 3]) >> (4 * (st & 7u))) & 15u) << (4 * z);
                        }
                    }
                }
                const uint32_t left = query_len - 8 * d;   // symbols of the read in this dword
                if (left < 8)
                    w |= 0xffffffffu << (4 * left);
            }
            lq[d * 64] = w;
        }
    }
    (void)ascii;   // ASCII reads were canonicalised with raw_code at upload (indels.cpp:977-991)

    const double match_score = prm.sc.match, mismatch_score = prm.sc.mismatch;
    const double gap_del_open = prm.sc.gap_del_open, gap_del_extend = prm.sc.gap_del_extend;
    const double gap_ins_open = prm.sc.gap_ins_open, gap_ins_extend = prm.sc.gap_ins_extend;
    const double clipping_score = prm.sc.clipping;

    // anchor, indels.cpp:167-169
    bool anchor_ok = true;
    for (uint32_t i = 0; i < fsize; i += 8)
        if (s.mm8(max_indel + fpos + i, fpos + i) & low_flags(fsize - i)) {
            anchor_ok = false;
            break;
        }
    if (!anchor_ok) {
        results[gid] = cm;
        return;
    }

    double total_cost = 0;
    const int min_size_after_clipping = (int)(query_len * prm.min_clipped_factor);
    int left_indel_size = 0, right_indel_size = 0;
    uint32_t no_left_mis_indel = query_len, no_right_mis_indel = query_len;
    double best_left_cost = 0, best_right_cost = 0;
    uint32_t best_left_indel_pos = 0, best_right_indel_pos = 0;
    bool left_indel_found = false, right_indel_found = false;
    double best_left_clip_cost = 0, best_right_clip_cost = 0;
    uint32_t best_no_left_clipped = 0, best_no_right_clipped = 0;
    uint32_t no_left_mis_clipped = 0, no_right_mis_clipped = 0;
    uint32_t best_no_left_mis_clipped = 0, best_no_right_mis_clipped = 0;
    const uint32_t sig_mask = ~0u >> (4 * (8 - IM_SIDE));

    // Seed hits are scored in lock step: every lane first lists where its 7-mer signature hits
    // (cheap, same trip count for everybody), then round h scores the h-th hit of every lane
    // together.  Scoring a hit where it is found would make the wave run the long split search
    // once per hit of every lane.  Order per lane is kept (strict > on costs: the first best wins).
    uint16_t* hit_list = reinterpret_cast<uint16_t*>(lds + hits_offset) + lane;   // [side * IM_HITS + h][lane]
    uint32_t n_left_hits = 0, n_right_hits = 0;
    auto score_left = [&](uint32_t i) {   // indels.cpp:205-230
        const uint32_t g_left = i - IM_SIDE + 1;
        uint32_t c_mis, indel_pos;
        split_mismatches(s, query_len, g_left, max_indel + fpos + 1, 0, fpos + 1, true, c_mis, indel_pos);
        if (c_mis > max_mis)
            return;
        const int c_indel = (int)max_indel - (int)g_left;
        if ((c_indel < 0 ? -c_indel : c_indel) > (int)max_indel)
            return;
        double cost = (fpos - c_mis) * match_score + c_mis * mismatch_score;
        if (c_indel > 0)
            cost += gap_del_open + (c_indel - 1) * gap_del_extend;
        else if (c_indel < 0)
            cost += gap_ins_open + (-c_indel - 1) * gap_ins_extend;
        if (cost > best_left_cost) {
            best_left_cost = cost;
            no_left_mis_indel = c_mis;
            best_left_indel_pos = indel_pos;
            left_indel_size = c_indel;
            left_indel_found = c_indel != 0;
        }
    };
    auto score_right = [&](uint32_t i) {   // indels.cpp:262-292
        const uint32_t right_pos = i - IM_SIDE + 1;
        uint32_t c_mis, indel_pos;
        split_mismatches(s, query_len, fpos + fsize - 1 + max_indel, right_pos + IM_SIDE, fpos + fsize - 1, query_len,
                         false, c_mis, indel_pos);
        if (c_mis > max_mis)
            return;
        const int c_indel = ((int)right_pos + IM_SIDE) - ((int)text_size - (int)max_indel);
        if ((c_indel < 0 ? -c_indel : c_indel) > (int)max_indel)
            return;
        double cost = (query_len - fpos - fsize - c_mis) * match_score + c_mis * mismatch_score;
        if (c_indel > 0)
            cost += gap_del_open + (c_indel - 1) * gap_del_extend;
        else if (c_indel < 0)
            cost += gap_ins_open + (-c_indel - 1) * gap_ins_extend;
        if (cost > best_right_cost) {
            best_right_cost = cost;
            no_right_mis_indel = c_mis;
            best_right_indel_pos = indel_pos;
            right_indel_size = c_indel;
            right_indel_found = c_indel != 0;
        }
    };

    // ---- left part: raw mismatches and best clipping point, 173-194
    uint32_t no_left_mis_raw = 0;
    double clipping_cost = clipping_score;
    for (int base = ((int)fpos - 1) & ~7; base >= 0 && fpos > 0; base -= 8) {
        const uint32_t nz = s.mm8(max_indel + (uint32_t)base, (uint32_t)base);
#pragma unroll
        for (int tt = 7; tt >= 0; --tt) {
            const int i = base + tt;
            if (i >= (int)fpos)
                continue;
            if ((nz >> (4 * tt)) & 1u) {
                ++no_left_mis_raw;
                ++no_left_mis_clipped;
                clipping_cost += mismatch_score;
            } else {
                clipping_cost += match_score;
                if (clipping_cost > best_left_clip_cost) {
                    best_left_clip_cost = clipping_cost;
                    best_no_left_clipped = (uint32_t)i;
                    best_no_left_mis_clipped = no_left_mis_clipped;
                }
            }
        }
    }
    if (fpos > (uint32_t)IM_SIDE) {   // 198-231
        // CBitCharsSingleWord (indels.h:162-206): newest symbol in the lowest 4-bit cell;
        // a symbol >= 4 is not inserted at all (the window does not advance)
        uint32_t qsig = 0;   // seven one-hot cells of 4 bits: 28 bits
        {
            const uint32_t w = s.Q8(0);
#pragma unroll
            for (int i = 0; i < IM_SIDE; ++i) {
                const uint32_t c = (w >> (4 * i)) & 15u;
                if (c < 4)
                    qsig += 1u << (c + 4 * (IM_SIDE - i - 1));
            }
        }
        uint32_t d = 0;
        {
            const uint32_t w = s.Gw(0);
#pragma unroll
            for (int i = 0; i < IM_SIDE - 1; ++i) {
                const uint32_t c = (w >> (4 * i)) & 15u;
                if (c < 4)
                    d += 1u << (c + 4 * (IM_SIDE - i - 2));
            }
        }
        const uint32_t end = max_indel + fpos;
        for (uint32_t base = 0; base < end; base += 8) {
            const uint32_t w = s.Gw(base >> 3);
#pragma unroll
            for (uint32_t tt = 0; tt < 8; ++tt) {
                const uint32_t i = base + tt;
                if (i < (uint32_t)IM_SIDE - 1 || i >= end)
                    continue;
                const uint32_t c = (w >> (4 * tt)) & 15u;
                if (c < 4)
                    d = ((d << 4) + (1u << c)) & sig_mask;
                if (__popc(d & qsig) < IM_SIDE - IM_SIDE_ERRORS)
                    continue;
                // a seed hit: listed, and scored below together with everybody else's
                if (n_left_hits == IM_HITS) {   // list full (repeats): score what is listed, in order, and go on
                    for (uint32_t h = 0; h < IM_HITS; ++h)
                        score_left(hit_list[h * 64]);
                    n_left_hits = 0;
                }
                hit_list[(n_left_hits++) * 64] = (uint16_t)i;
            }
        }
    }

    // ---- right part, 235-293
    uint32_t no_right_mis_raw = 0;
    clipping_cost = clipping_score;
    for (uint32_t base = (fpos + fsize) & ~7u; base < query_len; base += 8) {
        const uint32_t nz = s.mm8(max_indel + base, base);
#pragma unroll
        for (uint32_t tt = 0; tt < 8; ++tt) {
            const uint32_t i = base + tt;
            if (i < fpos + fsize || i >= query_len)
                continue;
            if ((nz >> (4 * tt)) & 1u) {
                ++no_right_mis_raw;
                ++no_right_mis_clipped;
                clipping_cost += mismatch_score;
            } else {
                clipping_cost += match_score;
                if (clipping_cost > best_right_clip_cost) {
                    best_right_clip_cost = clipping_cost;
                    best_no_right_mis_clipped = no_right_mis_clipped;
                    best_no_right_clipped = query_len - i - 1;
                }
            }
        }
    }
    if (fpos + fsize + IM_SIDE < query_len) {
        uint32_t qsig = 0;
        {
            const uint32_t w = s.Q8(query_len - IM_SIDE);
#pragma unroll
            for (int i = 0; i < IM_SIDE; ++i) {
                const uint32_t c = (w >> (4 * i)) & 15u;
                if (c < 4)
                    qsig += 1u << (c + 4 * (IM_SIDE - i - 1));
            }
        }
        const uint32_t begin = max_indel + fpos + fsize;
        uint32_t d = 0;
        {
            const uint32_t w = s.G8(begin);
#pragma unroll
            for (int i = 0; i < IM_SIDE - 1; ++i) {
                const uint32_t c = (w >> (4 * i)) & 15u;
                if (c < 4)
                    d += 1u << (c + 4 * (IM_SIDE - i - 2));
            }
        }
        for (uint32_t base = (begin + IM_SIDE - 1) & ~7u; base < text_size; base += 8) {
            const uint32_t w = s.Gw(base >> 3);
#pragma unroll
            for (uint32_t tt = 0; tt < 8; ++tt) {
                const uint32_t i = base + tt;
                if (i < begin + IM_SIDE - 1 || i >= text_size)
                    continue;
                const uint32_t c = (w >> (4 * tt)) & 15u;
                if (c < 4)
                    d = ((d << 4) + (1u << c)) & sig_mask;
                if (__popc(d & qsig) < IM_SIDE - IM_SIDE_ERRORS)
                    continue;
                if (n_right_hits == IM_HITS) {
                    for (uint32_t h = 0; h < IM_HITS; ++h)
                        score_right(hit_list[(IM_HITS + h) * 64]);
                    n_right_hits = 0;
                }
                hit_list[(IM_HITS + n_right_hits++) * 64] = (uint16_t)i;
            }
        }
    }

    for (uint32_t h = 0; __any(h < n_left_hits); ++h)
        if (h < n_left_hits)
            score_left(hit_list[h * 64]);
    for (uint32_t h = 0; __any(h < n_right_hits); ++h)
        if (h < n_right_hits)
            score_right(hit_list[(IM_HITS + h) * 64]);

    auto rescale = [&](uint32_t size) { return size * max_mis / query_len; };   // indels.h:252-255
    const uint32_t left_pos =
        left_indel_size > 0 ? ref_pos - (uint32_t)left_indel_size : ref_pos + (uint32_t)(-left_indel_size);

    // ---- the eight combinations in the reference's order, 297-457
    if (left_indel_found && no_left_mis_indel + no_right_mis_raw < max_mis) {
        const double c = best_left_cost + (query_len - fpos - no_right_mis_raw) * match_score +
                         no_right_mis_raw * mismatch_score;
        if (c > total_cost) {
            total_cost = c;
            cm.pos = left_pos;
            cm.type = WAMD_MAP_INDEL1;
            cm.matching_len1 = best_left_indel_pos;
            cm.indel_len1 = left_indel_size;
            cm.no_mismatches = no_left_mis_indel + no_right_mis_raw;
        }
    }
    if (right_indel_found && no_left_mis_raw + no_right_mis_indel < max_mis) {
        const double c = best_right_cost + (fpos + fsize - no_left_mis_raw) * match_score +
                         no_left_mis_raw * mismatch_score;
        if (c > total_cost) {
            total_cost = c;
            cm.pos = ref_pos;
            cm.type = WAMD_MAP_INDEL1;
            cm.matching_len1 = best_right_indel_pos;
            cm.indel_len1 = right_indel_size;
            cm.no_mismatches = no_left_mis_raw + no_right_mis_indel;
        }
    }
    if (left_indel_found && right_indel_found && no_left_mis_indel + no_right_mis_indel < max_mis) {
        const double c = best_left_cost + best_right_cost + fsize * match_score;
        if (c > total_cost) {
            total_cost = c;
            cm.type = WAMD_MAP_INDEL2;
            cm.pos = left_pos;
            cm.matching_len1 = best_left_indel_pos;
            cm.indel_len1 = left_indel_size;
            cm.matching_len2 = left_indel_size > 0
                                   ? best_right_indel_pos - best_left_indel_pos
                                   : best_right_indel_pos - (best_left_indel_pos + (uint32_t)(-left_indel_size));
            cm.indel_len2 = right_indel_size;
            cm.no_mismatches = no_left_mis_indel + no_right_mis_indel;
        }
    }
    if (best_no_left_clipped > 0 &&
        best_no_left_mis_clipped + no_right_mis_raw < rescale(query_len - best_no_left_clipped)) {
        const double c = best_left_clip_cost + (query_len - fpos - no_right_mis_raw) * match_score +
                         no_right_mis_raw * mismatch_score;
        if (c > total_cost && query_len - best_no_left_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm.pos = ref_pos + best_no_left_clipped;
            cm.type = WAMD_MAP_CLIPPING_MISMATCHES;
            cm.clipping_len = best_no_left_clipped;
            cm.matching_len1 = query_len - best_no_left_clipped;
            cm.no_mismatches = best_no_left_mis_clipped + no_right_mis_raw;
        }
    }
    if (best_no_right_clipped > 0 &&
        no_left_mis_raw + best_no_right_mis_clipped < rescale(query_len - best_no_right_clipped)) {
        const double c = best_right_clip_cost + (fpos + fsize - no_left_mis_raw) * match_score +
                         no_left_mis_raw * mismatch_score;
        if (c > total_cost && query_len - best_no_right_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm.pos = ref_pos;
            cm.type = WAMD_MAP_MISMATCHES_CLIPPING;
            cm.clipping_len = best_no_right_clipped;
            cm.matching_len1 = query_len - best_no_right_clipped;
            cm.no_mismatches = no_left_mis_raw + best_no_right_mis_clipped;
        }
    }
    if (best_no_left_clipped > 0 && right_indel_found &&
        best_no_left_mis_clipped + no_right_mis_indel < rescale(query_len - best_no_left_clipped)) {
        const double c = best_left_clip_cost + best_right_cost + fsize * match_score;
        if (c > total_cost && query_len - best_no_left_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm.pos = ref_pos + best_no_left_clipped;
            cm.type = WAMD_MAP_CLIPPING_INDEL;
            cm.indel_len1 = right_indel_size;
            cm.clipping_len = best_no_left_clipped;
            cm.matching_len1 = best_right_indel_pos - best_no_left_clipped;
            cm.no_mismatches = best_no_left_mis_clipped + no_right_mis_indel;
        }
    }
    if (best_no_right_clipped > 0 && left_indel_found &&
        best_no_right_mis_clipped + no_left_mis_indel < rescale(query_len - best_no_right_clipped)) {
        const double c = best_right_clip_cost + best_left_cost + fsize * match_score;
        if (c > total_cost && query_len - best_no_right_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm.pos = left_pos;
            cm.type = WAMD_MAP_INDEL_CLIPPING;
            cm.indel_len1 = left_indel_size;
            cm.clipping_len = best_no_right_clipped;
            cm.matching_len1 = best_left_indel_pos;
            cm.no_mismatches = best_no_right_mis_clipped + no_left_mis_indel;
        }
    }
    if (best_no_left_clipped > 0 && best_no_right_clipped > 0 &&
        best_no_left_mis_clipped + best_no_right_mis_clipped <
            rescale(query_len - best_no_left_clipped - best_no_right_clipped)) {
        const double c = best_left_clip_cost + best_right_clip_cost + fsize * match_score;
        if (c > total_cost &&
            query_len - best_no_left_clipped - best_no_right_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm.pos = ref_pos + best_no_left_clipped;
            cm.type = WAMD_MAP_CLIPPING_CLIPPING;
            cm.clipping_len = best_no_left_clipped;
            cm.matching_len1 = query_len - best_no_left_clipped - best_no_right_clipped;
            cm.no_mismatches = best_no_left_mis_clipped + best_no_right_mis_clipped;
        }
    }
    if (cm.type != WAMD_MAP_NONE) {
        cm.found = 1;
        // calc_penalty, results.cpp:327-343 with types.h:242-247
        const uint32_t add = (cm.type == WAMD_MAP_INDEL1 || cm.type == WAMD_MAP_CLIPPING_MISMATCHES ||
                              cm.type == WAMD_MAP_MISMATCHES_CLIPPING) ? 1u : 2u;
        cm.penalty = cm.no_mismatches + add;
    }
    results[gid] = cm;
}

int launch_indel_match(wamd_ctx* ctx, const wamd_reads* r, const wamd_indel_task* d_tasks, uint32_t n,
                       const wamd_indel_params* prm, wamd_mapping* d_results, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    IndelKernelParams kp;
    kp.sc = prm->scoring;
    kp.min_clipped_factor = prm->min_clipped_factor;
    const uint32_t blocks = (n + 63) / 64;
    const ReadsView rv = r->view();
    const int ascii = r->encoding == WAMD_READS_ASCII;
    const uint32_t hit_bytes = 2 * IM_HITS * 2 * 64;   // seed-hit lists: two sides, uint16 per hit and lane
    // staged variant: reads up to 256 symbols with max_indel up to 80, LDS sized by the batch's bound
    // (a task beyond the bound is flagged by the kernel); anything larger runs unstaged
    const uint32_t bound = prm->max_indel_bound ? prm->max_indel_bound : IM_MAX_INDEL;
    if (r->max_len <= IM_MAX_READ && bound <= IM_MAX_INDEL) {
        const uint32_t text_dw = (r->max_len + 2 * bound + 2 + 7) / 8 + 1;
        const uint32_t seq_bytes = (text_dw + (r->max_len + 7) / 8 + 1) * 4 * 64;
        hipLaunchKernelGGL(indel_match_kernel<true>, dim3(blocks), dim3(64), seq_bytes + hit_bytes, s, ctx->d_genome,
                           2 * ctx->genome_bytes, rv, ascii, d_tasks, n, kp, text_dw, seq_bytes, d_results,
                           ctx->d_flags);
    } else {
        hipLaunchKernelGGL(indel_match_kernel<false>, dim3(blocks), dim3(64), hit_bytes, s, ctx->d_genome,
                           2 * ctx->genome_bytes, rv, ascii, d_tasks, n, kp, 0u, 0u, d_results, ctx->d_flags);
    }
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

} // namespace wamd
