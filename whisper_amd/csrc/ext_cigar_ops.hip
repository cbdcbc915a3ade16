// ext_cigar_ops.hip - the linear passes of the post-processing half of the path:
//   * CIndelMatching::GetExtCigarNew   (mapper/indels.cpp:938-957, 680-935)
//   * CSoftClipping::clipBoundaryPositions (mapper/soft_clipping.cpp:827-994)
//   * candidate_mapping_t::serialize   (mapper/results.cpp:161-217)
// One record per lane.  All three are byte-stream scans with a handful of double adds
// in a fixed order, so they are HBM/latency bound; they exist on the device so that
// ext-CIGARs produced by the align kernels need not travel to the host and back.
#include <mutex>

#include "wamd_internal.h"

namespace wamd {

// rev_comp_code of CIndelMatching (indels.cpp:40-49)
__device__ __forceinline__ uint32_t rc_code_indel2(uint32_t c) { return c < 4 ? 3 - c : (c < 8 ? c : 15); }

// code2symbol of CIndelMatching (indels.cpp:69-76): ACGT, everything else 'N'
__device__ __forceinline__ uint32_t code2symbol(uint32_t c)
{
    return c < 4 ? ((0x54474341u >> (8 * c)) & 0xffu) : (uint32_t)'N';
}

// Bytes of an ext-CIGAR gather in a dword, dwords in a 16-byte buffer that leaves with one store where the slot is
// 16-byte aligned (every lane writes its own slot: a wave's store costs the 64 lines it touches whatever its width,
// and a line that is written in four pieces at four times is four partial writes once it has left L2 in between).
struct CigarWriter {
    uint8_t* slot;
    bool wide;
    uint32_t cur, acc;
    uint32_t q0, q1, q2;
    __device__ __forceinline__ void word(uint32_t at_dw, uint32_t v)
    {
        if (!wide) {
            *reinterpret_cast<uint32_t*>(slot + 4 * at_dw) = v;
            return;
        }
        const uint32_t qi = at_dw & 3u;
        if (qi == 3) {
            const wamd_u32x4 o = {q0, q1, q2, v};
            *reinterpret_cast<wamd_u32x4*>(slot + 4 * (at_dw - 3)) = o;
        }
        q0 = qi == 0 ? v : q0;
        q1 = qi == 1 ? v : q1;
        q2 = qi == 2 ? v : q2;
    }
    __device__ __forceinline__ void put(uint32_t b)
    {
        acc |= b << (8 * (cur & 3u));
        ++cur;
        if ((cur & 3u) == 0) {
            word(cur / 4 - 1, acc);
            acc = 0;
        }
    }
    __device__ __forceinline__ uint32_t finish()
    {
        const uint32_t len = cur;
        put(0);
        uint32_t n_dw = cur / 4;   // whole dwords written so far
        if (cur & 3u) {
            word(n_dw, acc);
            ++n_dw;
        }
        if (wide) {   // what the buffer still holds
            const uint32_t blk = n_dw & ~3u, left = n_dw & 3u;
            uint32_t* dst = reinterpret_cast<uint32_t*>(slot) + blk;
            if (left > 0)
                dst[0] = q0;
            if (left > 1)
                dst[1] = q1;
            if (left > 2)
                dst[2] = q2;
        }
        return len;
    }
};

// One record per lane, one wave per block.  The symbol loops read nothing from memory: the lane first fetches its
// read and its genome window with 16-byte loads, all in flight together, into its columns of an LDS buffer
// ([dword][lane]: no two lanes share a bank, no barrier is needed - a lane only ever touches its own columns); the
// ext-CIGAR leaves through CigarWriter's 16-byte buffer.  (Round 2 before this: a dword load on demand per eight
// symbols, each waited for together with the dword stores of the string before it, and every one of those stores a
// partial write of a line that had left L2 since the last one: 6.8 KB of traffic per record, SQ_WAIT_ANY 0.86,
// 0.65 ms per 425 K records; now 0.44.  Keeping the string in a third column until the end cost more in waves per
// CU than it saved; one loop over all of a lane's segments instead of the reference's call sequence per mapping
// type was slower as well - the bench's records are nearly all of one type.)  Symbols beyond the staged dwords
// (a window longer than qd / gd say) still come from memory.
__global__ __launch_bounds__(64) void indel_ext_cigar_kernel(GenomeView gv, ReadsView rv,
                                                             const wamd_indel_cigar_task* __restrict__ tasks,
                                                             uint32_t n, wamd_scoring sc,
                                                             wamd_indel_cigar_result* __restrict__ results,
                                                             uint8_t* __restrict__ cigars, uint32_t cigar_stride,
                                                             uint32_t qd, uint32_t gd, uint32_t* __restrict__ flags)
{
    extern __shared__ uint32_t gx_lds[];
    const uint8_t* genome = reinterpret_cast<const uint8_t*>(gv.dwords);
    const uint64_t genome_symbols = gv.n_symbols;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n)
        return;
    uint32_t* const q_col = gx_lds + threadIdx.x;
    uint32_t* const g_col = q_col + qd * 64;
    const wamd_indel_cigar_task t = tasks[gid];
    wamd_indel_cigar_result r;
    r.score = 0;
    r.ok = 0;
    r.cigar_len = 0;
    r.err_edit_distance = r.num_events = r.ref_length = r.left_clipping = r.right_clipping = r.method = 0;
    r.reserved = 0;
    uint8_t* slot = cigars + (uint64_t)gid * cigar_stride;
    if (t.read_idx == WAMD_READ_NONE) {   // a slot without a task: skipped, its ext-CIGAR left alone, no flag
        results[gid] = r;
        return;
    }
    slot[0] = 0;
    uint32_t err = 0;
    uint32_t m = 0;
    if (t.read_idx >= rv.n)
        err = FLAG_READ_IDX;
    else
        m = rv.lens[t.read_idx];
    const uint32_t a1 = t.indel_len1 < 0 ? -t.indel_len1 : t.indel_len1;
    const uint32_t a2 = t.indel_len2 < 0 ? -t.indel_len2 : t.indel_len2;
    if (!err && (uint64_t)t.examined_start_pos + m + a1 + a2 + 1 > genome_symbols)
        err = FLAG_RANGE;
    if (!err && 2 * (m + a1 + a2) + 8 > cigar_stride)
        err = FLAG_CIGAR;
    if (!err && (t.matching_len1 > m || t.matching_len2 > m || t.clipping_len > m))
        err = FLAG_ARG;
    if (err) {
        raise_flag(flags, err);
        results[gid] = r;
        return;
    }
    const uint8_t* codes = rv.codes + rv.code_off[t.read_idx];
    const bool rev = t.flags & WAMD_TASK_FLAG_REV_COMP;
    // Both sequences are walked symbol by symbol; a byte load per symbol is a line fetch per lane and symbol
    // (24 KB of line traffic per record), so each keeps the dword it is in (eight symbols, either direction).
    // Genome and code buffers are 4-byte aligned and padded to whole dwords.
    const uint32_t* codes_dw = reinterpret_cast<const uint32_t*>(codes);
    const uint32_t* genome_dw = reinterpret_cast<const uint32_t*>(genome);
    // staged: the read's dwords 0 .. q_n - 1 and the genome's g_first .. g_first + g_n - 1, nibble-swapped
    const uint32_t q_n = (m + 7) / 8 < qd ? (m + 7) / 8 : qd;
    const uint32_t g_first = (uint32_t)(t.examined_start_pos >> 3);
    const uint32_t g_need = (uint32_t)(((uint64_t)t.examined_start_pos + m + a1 + a2 + 1 + 7) >> 3) - g_first;
    const uint32_t g_n = g_need < gd ? g_need : gd;
    for (uint32_t d4 = 0; d4 < q_n; d4 += 4) {   // (the codes buffer has 64 bytes of slack behind the last read)
        const wamd_u32x4_a4 v = *reinterpret_cast<const wamd_u32x4_a4*>(codes_dw + d4);
        const uint32_t b[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c)
            if (d4 + c < q_n)
                q_col[(d4 + c) * 64] = nibble_swap(b[c]);
    }
    for (uint32_t d4 = 0; d4 < g_n; d4 += 4) {
        uint32_t b[4];
        load_genome_raw4(gv, g_first + d4, b);
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c)
            if (d4 + c < g_n)
                g_col[(d4 + c) * 64] = nibble_swap(b[c]);
    }
    uint32_t q_at = 0xffffffffu, q_dw = 0;
    uint32_t g_at = 0xffffffffu, g_dw = 0;
    auto Qraw = [&](uint32_t i) -> uint32_t {
        if ((i >> 3) != q_at) {
            q_at = i >> 3;
            q_dw = q_at < q_n ? q_col[q_at * 64] : nibble_swap(codes_dw[q_at]);
        }
        return (q_dw >> (4 * (i & 7u))) & 15u;
    };
    auto Q = [&](uint32_t i) -> uint32_t { return rev ? rc_code_indel2(Qraw(m - 1 - i)) : Qraw(i); };
    auto G = [&](uint32_t i) -> uint32_t {
        const uint64_t p = (uint64_t)t.examined_start_pos + i;
        const uint32_t at = (uint32_t)(p >> 3);
        if (at != g_at) {
            g_at = at;
            g_dw = at - g_first < g_n ? g_col[(at - g_first) * 64] : nibble_swap(genome_dw[at]);
        }
        return (g_dw >> (4 * ((uint32_t)p & 7u))) & 15u;
    };

    CigarWriter w{slot, (((uintptr_t)cigars | cigar_stride) & 15u) == 0, 0, 0, 0, 0, 0};
    uint32_t i_genome = 0, i_query = 0;
    uint16_t edist = 0, events = 0;
    double score = 0;
    bool overflow = false;

    auto matching = [&](uint32_t len) {   // append_matching_part, indels.cpp:680-702
        if (i_query + len > m) {
            overflow = true;
            return;
        }
        int no_mis = 0;
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t g = G(i_genome), q = Q(i_query);
            if (g == q) {
                w.put('.');
            } else {
                w.put(code2symbol(g));
                ++no_mis;
            }
            ++i_genome;
            ++i_query;
        }
        score += (len - no_mis) * sc.match + no_mis * sc.mismatch;
        events = (uint16_t)(events + no_mis);
        edist = (uint16_t)(edist + no_mis);
    };
    auto indel = [&](int len) {   // append_indel, indels.cpp:705-732
        if (len > 0) {
            for (uint32_t i = 0; i < (uint32_t)len; ++i) {
                w.put('^');
                w.put(code2symbol(G(i_genome++)));
            }
            score += sc.gap_del_open + (len - 1) * sc.gap_del_extend;
            events = (uint16_t)(events + 1);
            edist = (uint16_t)(edist + len);
        } else {
            if (i_query + (uint32_t)-len > m) {
                overflow = true;
                return;
            }
            for (uint32_t i = 0; i < (uint32_t)-len; ++i) {
                w.put('#');
                w.put(code2symbol(Q(i_query++)));
            }
            // the reference multiplies by (indel_len - 1) with indel_len negative (728)
            score += sc.gap_ins_open + (len - 1) * sc.gap_ins_extend;
            events = (uint16_t)(events + 1);
            edist = (uint16_t)(edist + -len);
        }
    };
    auto clipping = [&](int len) {   // append_clipping, indels.cpp:735-745
        if (len < 0 || i_query + (uint32_t)len > m) {
            overflow = true;
            return;
        }
        for (uint32_t i = 0; i < (uint32_t)len; ++i)
            w.put('$');
        score += sc.clipping;
        events = (uint16_t)(events + 1);
        i_query += (uint32_t)len;
    };

    r.ok = 1;
    switch (t.type) {
    case WAMD_MAP_INDEL1:   // indels.cpp:748-773
        matching(t.matching_len1);
        indel(t.indel_len1);
        matching(m - i_query);
        r.method = t.indel_len1 > 0 ? 14 : 13;
        r.ref_length = (uint16_t)((int)m + t.indel_len1);
        break;
    case WAMD_MAP_INDEL2:   // 776-811
        matching(t.matching_len1);
        indel(t.indel_len1);
        matching(t.matching_len2);
        indel(t.indel_len2);
        matching(m - i_query);
        r.method = t.indel_len1 > 0 ? (t.indel_len2 > 0 ? 18 : 17) : (t.indel_len2 > 0 ? 15 : 16);
        r.ref_length = (uint16_t)((int)m + t.indel_len1 + t.indel_len2);
        break;
    case WAMD_MAP_CLIPPING_MISMATCHES:
    case WAMD_MAP_MISMATCHES_CLIPPING:   // 814-848
        if (t.type == WAMD_MAP_CLIPPING_MISMATCHES) {
            r.method = 19;
            clipping((int)(m - t.matching_len1));
            r.left_clipping = (uint16_t)(m - t.matching_len1);
        }
        matching(t.matching_len1);
        if (t.type == WAMD_MAP_MISMATCHES_CLIPPING) {
            r.method = 20;
            clipping((int)(m - t.matching_len1));
            r.right_clipping = (uint16_t)(m - t.matching_len1);
        }
        r.ref_length = (uint16_t)t.matching_len1;
        break;
    case WAMD_MAP_CLIPPING_INDEL:
    case WAMD_MAP_INDEL_CLIPPING:   // 851-906
        if (t.type == WAMD_MAP_CLIPPING_INDEL) {
            r.method = t.indel_len1 > 0 ? 23 : 21;
            clipping((int)t.clipping_len);
            matching(t.matching_len1);
            indel(t.indel_len1);
            matching(m - i_query);
            r.left_clipping = (uint16_t)t.clipping_len;
        } else {
            r.method = t.indel_len1 > 0 ? 24 : 22;
            matching(t.matching_len1);
            indel(t.indel_len1);
            if (t.clipping_len > m - i_query)   // (the reference would run off the read here)
                overflow = true;
            else
                matching(m - i_query - t.clipping_len);
            clipping((int)t.clipping_len);
            r.right_clipping = (uint16_t)t.clipping_len;
        }
        r.ref_length = (uint16_t)((int)m - (int)t.clipping_len + t.indel_len1);
        break;
    case WAMD_MAP_CLIPPING_CLIPPING:   // 909-935
        r.method = 25;
        r.left_clipping = (uint16_t)t.clipping_len;
        clipping((int)r.left_clipping);
        matching(t.matching_len1);
        clipping((int)(m - t.matching_len1 - r.left_clipping));
        r.right_clipping = (uint16_t)(m - t.matching_len1 - r.left_clipping);
        r.ref_length = (uint16_t)t.matching_len1;
        break;
    default:
        r.ok = 0;
        break;
    }
    if (overflow) {   // a record that does not describe this read
        raise_flag(flags, FLAG_ARG);
        r.ok = 0;
    }
    r.cigar_len = w.finish();
    r.score = score;
    r.err_edit_distance = edist;
    r.num_events = events;
    results[gid] = r;
}

__device__ __forceinline__ bool is_alpha(uint32_t c) { return ((c | 32u) - 'a') < 26u; }

// Strings by the dword.  Most of an ext-CIGAR is runs of '.', and a lane walking its string byte by byte pays a whole
// state-machine step per symbol; where the next four bytes are dots the routines below take them in one step (what
// four steps of the reference's loop would do to their state is written out at each site).  `str4` reads four
// bytes at any offset of a string whose base is dword-aligned; callers keep e + 8 <= the string's capacity.
constexpr uint32_t FOUR_DOTS = 0x2E2E2E2Eu;
__device__ __forceinline__ bool str_aligned(const uint8_t* s) { return ((uintptr_t)s & 3u) == 0; }
__device__ __forceinline__ uint32_t str4(const uint8_t* s, uint32_t e)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(s) + (e >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], e & 3u);
}

// strlen bounded by the slot: whole dwords while none of their bytes is zero
__device__ __forceinline__ uint32_t bounded_len(const uint8_t* ec, uint32_t stride)
{
    uint32_t l = 0;
    if (str_aligned(ec)) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(ec);
        while (l + 4 <= stride) {
            const uint32_t v = w[l >> 2];
            if ((v - 0x01010101u) & ~v & 0x80808080u)
                break;
            l += 4;
        }
    }
    while (l < stride && ec[l])
        ++l;
    return l;
}

// One record: the body of clipBoundaryPositions.  `ec` may point to global memory or to an LDS copy of the slot.
__device__ void clip_boundary_one(wamd_clip_record& d, uint8_t* ec, uint32_t cigar_stride, const wamd_scoring& sc)
{
    d.no_clippings = 0;

    auto length = [&]() { return bounded_len(ec, cigar_stride); };
    if (d.err_edit_distance == 0) {
        d.cigar_len = length();
        return;
    }
    int mini_pos = -1, maxi_pos = -1;
    double mini = 1.7976931348623157e308, maxi = -1.7976931348623157e308, cur = 0;
    const double left_penalty = d.left_clipping > 0 ? 0 : sc.clipping;
    const double right_penalty = d.right_clipping > 0 ? 0 : sc.clipping;
    uint32_t in = d.left_clipping;
    bool in_indel = false;
    int pos;
    // four dots at once: with a positive match score the running score rises through them, so only the first can
    // be a new minimum and only the last a new maximum (the comparisons are strict); the sums are taken one by one
    const bool by4 = str_aligned(ec) && sc.match > 0;
    for (pos = 0; in < cigar_stride && ec[in] && ec[in] != '$'; ++pos) {   // 849-890
        if (by4 && in + 8 <= cigar_stride && str4(ec, in) == FOUR_DOTS) {
            cur += sc.match;
            if (cur < mini) {
                mini_pos = pos;
                mini = cur;
            }
            cur += sc.match;
            cur += sc.match;
            cur += sc.match;
            pos += 3;
            in += 4;
            in_indel = false;
            if (cur > maxi) {
                maxi_pos = pos;
                maxi = cur;
            }
            continue;
        }
        const uint32_t c = ec[in];
        if (c == '.') {
            cur += sc.match;
            in_indel = false;
            ++in;
        } else if (c == '#') {
            cur += in_indel ? sc.gap_ins_extend : sc.gap_ins_open;
            in_indel = true;
            in += 2;
            ++pos;
        } else if (c == '^') {
            cur += in_indel ? sc.gap_del_extend : sc.gap_del_open;
            in_indel = true;
            in += 2;
            ++pos;
        } else if (is_alpha(c)) {
            cur += sc.mismatch;
            in_indel = false;
            ++in;
        } else {
            ++in;
        }
        if (cur < mini) {
            mini_pos = pos;
            mini = cur;
        }
        if (cur > maxi) {
            maxi_pos = pos;
            maxi = cur;
        }
    }
    if (mini_pos + 20 > maxi_pos) {
        d.cigar_len = length();
        return;
    }
    int no_clippings = 0;
    if (maxi + right_penalty > cur) {   // right clipping, 898-941
        ++no_clippings;
        in = d.left_clipping + (uint32_t)maxi_pos + 1;
        uint32_t out = in;
        while (ec[in] && ec[in] != '$') {
            const uint32_t c = ec[in];
            if (c == '.') {
                ec[out++] = '$';
                ++in;
                ++d.right_clipping;
                --d.ref_length;
            } else if (c == '#') {
                ec[out++] = '$';
                in += 2;
                ++d.right_clipping;
                --d.err_edit_distance;
            } else if (c == '^') {
                in += 2;
                --d.ref_length;
                --d.err_edit_distance;
            } else if (is_alpha(c)) {
                ec[out++] = '$';
                ++in;
                ++d.right_clipping;
                --d.ref_length;
                --d.err_edit_distance;
            } else {
                ++in;
            }
        }
        while (ec[in])
            ec[out++] = ec[in++];
        ec[out] = 0;
    }
    if (mini < left_penalty) {   // left clipping, 944-991
        ++no_clippings;
        in = d.left_clipping;
        uint32_t out = in;
        const uint32_t guard = in + (uint32_t)mini_pos + 1;
        while (in != guard) {
            const uint32_t c = ec[in];
            if (c == '.') {
                ec[out++] = '$';
                ++in;
                ++d.left_clipping;
                ++d.ref_seq_pos;
                --d.ref_length;
            } else if (c == '#') {
                ec[out++] = '$';
                in += 2;
                ++d.left_clipping;
                --d.err_edit_distance;
            } else if (c == '^') {
                in += 2;
                ++d.ref_seq_pos;
                --d.ref_length;
                --d.err_edit_distance;
            } else if (is_alpha(c)) {
                ec[out++] = '$';
                ++in;
                ++d.left_clipping;
                ++d.ref_seq_pos;
                --d.ref_length;
                --d.err_edit_distance;
            } else {
                ++in;
            }
        }
        while (ec[in])
            ec[out++] = ec[in++];
        ec[out] = 0;
    }
    // 993
    d.score = (0.0 > maxi ? 0.0 : maxi) - (0.0 < mini ? 0.0 : mini) + no_clippings * sc.clipping;
    d.no_clippings = (uint32_t)no_clippings;
    d.cigar_len = length();
}

__global__ __launch_bounds__(256) void clip_boundary_kernel(uint32_t n, wamd_clip_record* __restrict__ records,
                                                            uint8_t* __restrict__ cigars, uint32_t cigar_stride,
                                                            wamd_scoring sc)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n)
        return;
    wamd_clip_record d = records[gid];
    clip_boundary_one(d, cigars + (uint64_t)gid * cigar_stride, cigar_stride, sc);
    records[gid] = d;
}

// ---- row L: clipIllegalPositions (soft_clipping.cpp:592-729) ---------------------------------
// One record per lane.  The three loops of the reference, byte for byte - including the tail
// loop that does not advance its input on an ordinary symbol and so pads with '$' up to
// read_length, and the '$' of an earlier clipping being scored as a mismatch.
__device__ void clip_illegal_one(wamd_clip_record& d, const wamd_clip_illegal_task& t, uint8_t* ec, uint32_t stride,
                                 const wamd_scoring& sc, uint32_t* flags)
{
    const int32_t ref_size = t.ref_size;
    const int32_t mapping_last = d.ref_seq_pos + (int32_t)d.ref_length - 1;   // 596
    const int read_length = t.read_length;
    int to_left = 0, to_right = 0;
    int i_read = 0, i_ref = d.ref_seq_pos;
    d.no_clippings = 0;
    if (d.ref_seq_pos <= 0)   // 604-610
        to_left = -d.ref_seq_pos + 1;
    if (mapping_last > ref_size)
        to_right = mapping_last - ref_size;
    if (!to_left && !to_right) {
        d.cigar_len = bounded_len(ec, stride);
        return;
    }
    if ((uint32_t)read_length + 1 > stride) {
        raise_flag(flags, FLAG_CIGAR);
        return;
    }
    if (t.perfect) {   // 614-618
        for (int i = 0; i < read_length; ++i)
            ec[i] = '.';
        ec[read_length] = 0;
    }
    d.left_clipping = (uint16_t)(d.left_clipping + to_left);   // 620-625
    d.right_clipping = (uint16_t)(d.right_clipping + to_right);
    d.ref_length = (uint16_t)(d.ref_length - (to_left + to_right));
    d.ref_seq_pos += to_left;
    d.err_edit_distance = 0;
    double score = 0;
    d.no_clippings = (uint32_t)((to_left != 0) + (to_right != 0));

    uint32_t in = 0, out = 0;
    const uint32_t lim = stride - 1;
    bool bad = false;
    auto at = [&](uint32_t i) -> uint32_t { return i < stride ? ec[i] : 0u; };
    uint32_t c;
    while (i_ref < 1 && (c = at(in)) != 0) {   // 632-652
        if (c == '^') {
            in += 2;
            ++i_ref;
        } else if (c == '#') {
            in += 2;
            ec[out++] = '$';
            ++i_read;
        } else {
            ++in;
            ec[out++] = '$';
            ++i_read;
            ++i_ref;
        }
    }
    bool gap_opened = false;
    while (i_read < read_length && i_ref <= ref_size && (c = at(in)) != 0) {   // 655-703
        if (c == '^' || c == '#') {
            if (in + 1 >= lim) {
                bad = true;
                break;
            }
            ec[out++] = (uint8_t)c;
            ec[out++] = ec[in + 1];
            in += 2;
            d.err_edit_distance++;
            if (c == '^') {
                ++i_ref;
                score += gap_opened ? sc.gap_del_extend : sc.gap_del_open;
            } else {
                ++i_read;
                score += gap_opened ? sc.gap_ins_extend : sc.gap_ins_open;
            }
            gap_opened = true;
        } else {
            if (c == '.')
                score += sc.match;
            else {
                score += sc.mismatch;
                d.err_edit_distance++;
            }
            ec[out++] = (uint8_t)c;
            ++in;
            ++i_read;
            ++i_ref;
            gap_opened = false;
        }
    }
    while (!bad && i_read < read_length && (c = at(in)) != 0) {   // 706-725
        if (c == '^') {
            in += 2;
            ++i_ref;
        } else if (c == '#') {
            in += 2;
            ec[out++] = '$';
            ++i_read;
        } else {   // 719-724: the input does not move
            if (out >= lim) {
                bad = true;
                break;
            }
            ec[out++] = '$';
            ++i_read;
            ++i_ref;
        }
    }
    if (out > lim)
        out = lim;
    ec[out] = 0;   // 727
    if (bad)
        raise_flag(flags, FLAG_ARG);
    d.score = score;
    d.cigar_len = out;
}

__global__ __launch_bounds__(256) void clip_illegal_kernel(uint32_t n, const wamd_clip_illegal_task* __restrict__ tasks,
                                                           wamd_clip_record* __restrict__ records,
                                                           uint8_t* __restrict__ cigars, uint32_t stride,
                                                           wamd_scoring sc, uint32_t* __restrict__ flags)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n)
        return;
    wamd_clip_record d = records[gid];
    clip_illegal_one(d, tasks[gid], cigars + (uint64_t)gid * stride, stride, sc, flags);
    records[gid] = d;
}

// ---- row L: clipOverlappingPairs with clipLeft / clipRight (soft_clipping.cpp:997-1151) -------
// One pair per lane.  Returns false where the reference would run off the string.
__device__ bool clip_left(wamd_clip_record& d, uint8_t* ec, uint32_t stride, int count)
{
    uint32_t in = d.left_clipping, out = in;
    int clipped = 0;
    auto at = [&](uint32_t i) -> uint32_t { return i < stride ? ec[i] : 0u; };
    uint32_t c;
    while (clipped < count || (clipped == count && at(in) == '#')) {
        c = at(in);
        if (c == '.') {
            ec[out++] = '$';
            ++in;
            ++d.left_clipping;
            ++d.ref_seq_pos;
            --d.ref_length;
            ++clipped;
        } else if (c == '#') {
            ec[out++] = '$';
            in += 2;
            ++d.left_clipping;
            --d.err_edit_distance;
        } else if (c == '^') {
            in += 2;
            ++d.ref_seq_pos;
            --d.ref_length;
            --d.err_edit_distance;
            ++clipped;
        } else if (is_alpha(c)) {
            ec[out++] = '$';
            ++in;
            ++d.left_clipping;
            ++d.ref_seq_pos;
            --d.ref_length;
            --d.err_edit_distance;
            ++clipped;
        } else {
            return false;
        }
    }
    if (out != in) {   // 1092-1097
        while ((c = at(in)) != 0) {
            ec[out++] = (uint8_t)c;
            ++in;
        }
        ec[out] = 0;
    }
    return true;
}

__device__ bool clip_right(wamd_clip_record& d, uint8_t* ec, uint32_t stride, int count)
{
    const int len = (int)bounded_len(ec, stride);
    int in = len - (int)d.right_clipping - 1, out = in + 1;
    int clipped = 0;
    auto prev = [&]() -> uint32_t { return in >= 1 ? ec[in - 1] : 0u; };
    while (clipped < count || (clipped == count && prev() == '#')) {
        if (in < 0)
            return false;
        const uint32_t c = ec[in];
        if (c == '.') {
            ec[--out] = '$';
            --in;
            ++d.right_clipping;
            --d.ref_length;
            ++clipped;
        } else if (is_alpha(c)) {
            const uint32_t p = prev();
            if (p == '#') {
                ec[--out] = '$';
                in -= 2;
                ++d.right_clipping;
                --d.err_edit_distance;
            } else if (p == '^') {
                in -= 2;
                --d.ref_length;
                --d.err_edit_distance;
                ++clipped;
            } else {
                ec[--out] = '$';
                --in;
                ++d.right_clipping;
                --d.ref_length;
                --d.err_edit_distance;
                ++clipped;
            }
        } else {
            return false;
        }
    }
    ++in;   // 1144-1150: the string becomes [0, in) + [out, len)
    if (in != out) {
        while (out < len)
            ec[in++] = ec[out++];
        ec[in] = 0;
    }
    return true;
}

__device__ void clip_pair_one(const wamd_clip_pair_task& t, wamd_clip_record (&rec)[2], uint8_t* ec0, uint8_t* ec1,
                              uint32_t stride, uint32_t* flags)
{
    uint8_t* ec[2] = {ec0, ec1};
    rec[0].no_clippings = rec[1].no_clippings = 0;
    bool touched = false;
    if (t.ref_seq_id[0] == t.ref_seq_id[1] && !rec[0].left_clipping && !rec[0].right_clipping &&
        !rec[1].left_clipping && !rec[1].right_clipping) {   // 999-1003
        int left;   // 1005-1025
        if (rec[0].ref_seq_pos < rec[1].ref_seq_pos)
            left = 0;
        else if (rec[0].ref_seq_pos == rec[1].ref_seq_pos)
            left = t.direct[0] ? 0 : 1;
        else
            left = 1;
        const int right = 1 - left;
        // dynamic indexing of the two records would put them in scratch memory: select instead
        wamd_clip_record l = left ? rec[1] : rec[0], r = left ? rec[0] : rec[1];
        uint8_t *ecl = left ? ec[1] : ec[0], *ecr = left ? ec[0] : ec[1];
        const bool left_direct = left ? t.direct[1] : t.direct[0];
        if (!left_direct && l.ref_seq_pos + (int32_t)l.ref_length > r.ref_seq_pos) {   // 1028
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (t.perfect[q]) {   // 1029-1036
                    const uint32_t rl = t.read_length[q];
                    if (rl + 1 > stride) {
                        ok = false;
                    } else {
                        for (uint32_t i = 0; i < rl; ++i)
                            ec[q][i] = '.';
                        ec[q][rl] = 0;
                    }
                }
            if (ok) {
                const int to_left = r.ref_seq_pos - l.ref_seq_pos;   // 1038-1039
                const int to_right = (r.ref_seq_pos + (int32_t)r.ref_length) - (l.ref_seq_pos + (int32_t)l.ref_length);
                ok = clip_left(l, ecl, stride, to_left);
                ok = clip_right(r, ecr, stride, to_right) && ok;
            }
            if (!ok)
                raise_flag(flags, FLAG_ARG);
            l.no_clippings = r.no_clippings = 1;
            rec[0] = left ? r : l;
            rec[1] = left ? l : r;
            touched = true;
        }
        (void)right;
    }
    (void)touched;
    rec[0].cigar_len = bounded_len(ec[0], stride);
    rec[1].cigar_len = bounded_len(ec[1], stride);
}

__global__ __launch_bounds__(256) void clip_pairs_kernel(uint32_t n_pairs, const wamd_clip_pair_task* __restrict__ tasks,
                                                         wamd_clip_record* __restrict__ records,
                                                         uint8_t* __restrict__ cigars, uint32_t stride,
                                                         uint32_t* __restrict__ flags)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_pairs)
        return;
    wamd_clip_record rec[2] = {records[2 * (uint64_t)gid], records[2 * (uint64_t)gid + 1]};
    clip_pair_one(tasks[gid], rec, cigars + 2 * (uint64_t)gid * stride, cigars + (2 * (uint64_t)gid + 1) * stride, stride,
                  flags);
    records[2 * (uint64_t)gid] = rec[0];
    records[2 * (uint64_t)gid + 1] = rec[1];
}

__device__ __forceinline__ uint8_t indel_size_encode(int32_t x)   // types.h:296-302
{
    return x > 0 ? (uint8_t)x : (uint8_t)(-x + 128);
}

__global__ __launch_bounds__(256) void serialize_kernel(uint32_t n, const wamd_mapping* __restrict__ mappings,
                                                        const uint8_t* __restrict__ direction,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ sizes)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n)
        return;
    const wamd_mapping m = mappings[gid];
    uint8_t b[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t k = 0;
    b[k++] = (uint8_t)(m.pos >> 24);   // StoreUInt big-endian, common/utils.cpp:431-445
    b[k++] = (uint8_t)(m.pos >> 16);
    b[k++] = (uint8_t)(m.pos >> 8);
    b[k++] = (uint8_t)m.pos;
    b[k++] = (uint8_t)((direction[gid] ? 0x80u : 0) + (m.type == WAMD_MAP_LEV ? 0x40u : 0) + (m.no_mismatches & 0x3f));
    if (m.type != WAMD_MAP_LEV) {
        b[k++] = (uint8_t)m.type;
        const uint8_t hi1 = (uint8_t)((m.matching_len1 >> 8) & 0xf);
        switch (m.type) {
        case WAMD_MAP_INDEL1:
            b[k++] = hi1;
            b[k++] = (uint8_t)(m.matching_len1 & 0xff);
            b[k++] = indel_size_encode(m.indel_len1);
            break;
        case WAMD_MAP_INDEL2:
            b[k++] = (uint8_t)(hi1 + ((m.matching_len2 >> 4) & 0xf0u));
            b[k++] = (uint8_t)(m.matching_len1 & 0xffu);
            b[k++] = (uint8_t)(m.matching_len2 & 0xffu);
            b[k++] = indel_size_encode(m.indel_len1);
            b[k++] = indel_size_encode(m.indel_len2);
            break;
        case WAMD_MAP_INDEL_CLIPPING:
        case WAMD_MAP_CLIPPING_INDEL:
            b[k++] = (uint8_t)(hi1 + ((m.clipping_len >> 4) & 0xf0u));
            b[k++] = (uint8_t)(m.matching_len1 & 0xffu);
            b[k++] = (uint8_t)(m.clipping_len & 0xffu);
            b[k++] = indel_size_encode(m.indel_len1);
            break;
        case WAMD_MAP_MISMATCHES_CLIPPING:
        case WAMD_MAP_CLIPPING_MISMATCHES:
            b[k++] = hi1;
            b[k++] = (uint8_t)(m.matching_len1 & 0xffu);
            break;
        case WAMD_MAP_CLIPPING_CLIPPING:
            b[k++] = (uint8_t)(hi1 + ((m.clipping_len >> 4) & 0xf0u));
            b[k++] = (uint8_t)(m.matching_len1 & 0xffu);
            b[k++] = (uint8_t)(m.clipping_len & 0xffu);
            break;
        default:
            break;
        }
    }
    uint32_t* o = reinterpret_cast<uint32_t*>(out + (uint64_t)gid * 12);
    o[0] = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
    o[1] = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
    o[2] = b[8] | (b[9] << 8) | (b[10] << 16) | ((uint32_t)b[11] << 24);
    sizes[gid] = (uint8_t)k;
}

// ---- accepted candidates -> mapping records, deterministic order ---------------------
// pass 1: accepted candidates per 256-task block; pass 2: exclusive scan of the block counts
// (one workgroup); pass 3: every block writes its records at its offset, in task order.
__global__ __launch_bounds__(256) void records_count_kernel(uint32_t n, const wamd_dist_task* __restrict__ tasks,
                                                            const wamd_dist_result* __restrict__ res,
                                                            uint32_t* __restrict__ block_counts)
{
    __shared__ uint32_t wave_cnt[4];
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const bool ok = gid < n && res[gid].edit_distance <= tasks[gid].max_ed;
    const uint64_t b = __ballot(ok);
    if ((threadIdx.x & 63) == 0)
        wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0)
        block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(1024) void records_scan_kernel(uint32_t nb, uint32_t* __restrict__ block_counts,
                                                            uint32_t* __restrict__ total)
{
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(nb, lo + per);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i)
        s += block_counts[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {   // Hillis-Steele inclusive scan
        const uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = block_counts[i];
        block_counts[i] = run;
        run += c;
    }
    if (t == 1023)
        *total = part[1023];
}

__global__ __launch_bounds__(256) void records_write_kernel(uint32_t n, const wamd_dist_task* __restrict__ tasks,
                                                            const wamd_dist_result* __restrict__ res,
                                                            const uint32_t* __restrict__ record_pos,
                                                            const uint8_t* __restrict__ direction,
                                                            const uint32_t* __restrict__ block_offsets,
                                                            uint8_t* __restrict__ records)
{
    __shared__ uint32_t wave_cnt[4];
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ed = 0;
    bool ok = false;
    if (gid < n) {
        ed = res[gid].edit_distance;
        ok = ed <= tasks[gid].max_ed;
    }
    const uint64_t b = __ballot(ok);
    if (lane == 0)
        wave_cnt[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    if (!ok)
        return;
    uint32_t rank = (uint32_t)__popcll(b & ((1ull << lane) - 1));
    for (uint32_t w = 0; w < wave; ++w)
        rank += wave_cnt[w];
    const uint32_t pos = record_pos[gid];
    // candidate_mapping_t::serialize for type lev, results.cpp:161-173
    const uint32_t flag = (direction[gid] ? 0x80u : 0u) + 0x40u + (ed & 0x3fu);
    uint32_t* o = reinterpret_cast<uint32_t*>(records + (uint64_t)(block_offsets[blockIdx.x] + rank) * 12);
    o[0] = tasks[gid].read_idx;
    o[1] = __builtin_bswap32(pos);
    o[2] = flag;
}

uint64_t records_scratch_bytes(uint32_t n) { return ((uint64_t)(n + 255) / 256 + 1) * sizeof(uint32_t) + 256; }

int launch_emit_lev_records(wamd_ctx* ctx, uint32_t n, const wamd_dist_task* d_tasks,
                            const wamd_dist_result* d_results, const uint32_t* d_record_pos,
                            const uint8_t* d_direction, uint8_t* d_records, uint32_t* d_count, void* d_scratch,
                            hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    const uint32_t nb = (n + 255) / 256;
    uint32_t* counts = (uint32_t*)d_scratch;
    hipLaunchKernelGGL(records_count_kernel, dim3(nb), dim3(256), 0, s, n, d_tasks, d_results, counts);
    hipLaunchKernelGGL(records_scan_kernel, dim3(1), dim3(1024), 0, s, nb, counts, d_count);
    hipLaunchKernelGGL(records_write_kernel, dim3(nb), dim3(256), 0, s, n, d_tasks, d_results, d_record_pos,
                       d_direction, counts, d_records);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

// ---- the same emission for callers that cannot allocate (wamd_emit_lev_records_static_dev) ----
// The block counts of the record scan live in a file-scope device array instead of a caller
// buffer.  Every GPU holds its own copy at its own address: the address is looked up once per
// device (records_static_init, called by wamd_ctx_create with that device current) and only that
// pointer is ever handed to a kernel - the array's name is a host-side stand-in here.
// Nothing in it must start at zero: the count kernel rewrites words [0, nb) before the scan reads
// them.  Calls that overlap on one GPU share the array and race.
constexpr uint32_t REC_STATIC_WORDS = WAMD_STATIC_RECORDS_MAX_TASKS / 256 + 64;   // 256.25 KiB
static __device__ __align__(256) uint32_t scratch[REC_STATIC_WORDS];
constexpr int REC_MAX_DEVICES = 64;
static uint32_t* g_scratch_ptr[REC_MAX_DEVICES];
static std::mutex g_scratch_lock;

int records_static_init(int device)
{
    if (device < 0 || device >= REC_MAX_DEVICES)
        return WAMD_OK;   // such a device simply has no static form (the entry point says so)
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (g_scratch_ptr[device])
        return WAMD_OK;
    uint32_t* ptr = nullptr;
    WAMD_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = ptr;
    return WAMD_OK;
}

int launch_emit_lev_records_static(wamd_ctx* ctx, uint32_t n, const wamd_dist_task* d_tasks,
                                   const wamd_dist_result* d_results, const uint32_t* d_record_pos,
                                   const uint8_t* d_direction, uint8_t* d_records, uint32_t* d_count, hipStream_t s)
{
    if (n > WAMD_STATIC_RECORDS_MAX_TASKS) {
        set_error("wamd_emit_lev_records_static_dev: %u tasks, the static scratch holds %u at most", n,
                  WAMD_STATIC_RECORDS_MAX_TASKS);
        return WAMD_ERR_CAPACITY;
    }
    uint32_t* ptr = nullptr;
    if (ctx->device >= 0 && ctx->device < REC_MAX_DEVICES) {
        std::lock_guard<std::mutex> hold(g_scratch_lock);
        ptr = g_scratch_ptr[ctx->device];
    }
    if (!ptr) {
        set_error("wamd_emit_lev_records_static_dev: no static scratch on device %d", ctx->device);
        return WAMD_ERR_CAPACITY;
    }
    if (n == 0)   // the entry point's check that this device has its scratch; nothing to launch
        return WAMD_OK;
    return launch_emit_lev_records(ctx, n, d_tasks, d_results, d_record_pos, d_direction, d_records, d_count, ptr, s);
}

int launch_indel_ext_cigar(wamd_ctx* ctx, const wamd_reads* r, const wamd_indel_cigar_task* d_tasks,
                           uint32_t n, wamd_indel_cigar_result* d_results, uint8_t* d_cigars,
                           uint32_t cigar_stride, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0)
        return WAMD_OK;
    // LDS columns per lane: the longest read, a genome window of that plus 64 symbols of deletions, and a string of
    // up to 1.25 bytes per read symbol (what is longer than that finishes in memory) - 31 KB per wave at 150 bp
    // (64 KB at most: reads above 560 symbols keep their tails in memory)
    uint32_t qd = (r->max_len + 7) / 8 + 1, gd = (r->max_len + 64 + 7) / 8 + 2;
    qd = qd < 72u ? qd : 72u;
    gd = gd < 88u ? gd : 88u;
    hipLaunchKernelGGL(indel_ext_cigar_kernel, dim3((n + 63) / 64), dim3(64), (size_t)(qd + gd) * 256, s,
                       genome_view(ctx), r->view(), d_tasks, n, ctx->scoring, d_results, d_cigars, cigar_stride, qd, gd,
                       ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

// ---- ext-CIGAR -> SAM CIGAR text, BAM operations, MD:Z (row X) ------------------------------
// CSamGenerator::extract_cigar_and_mdz / extract_cigarbin_and_mdz (sam.cpp:805-928, 932-1051):
// one state machine over the ext-CIGAR bytes; an operation is flushed when the state changes,
// matches and mismatches share one 'M' run, MD:Z counts matches across insertions and writes
// the count (also a 0) in front of every mismatch and of EVERY deleted base.
__device__ __forceinline__ uint32_t put_decimal(uint8_t* dst, uint32_t v)
{
    uint32_t n = 1;
    for (uint32_t t = v; t >= 10; t /= 10)
        ++n;
    for (uint32_t i = n; i-- > 0; v /= 10)
        dst[i] = (uint8_t)('0' + v % 10);
    return n;
}

__device__ wamd_sam_cigar_len sam_cigar_one(const uint8_t* ext, uint32_t ext_stride, uint8_t* cg, uint32_t cigar_stride,
                                            uint8_t* md, uint32_t mdz_stride, uint32_t* cb, uint32_t bin_stride,
                                            uint32_t* flags)
{
    enum { Match, Mismatch, Insertion, Deletion, Begin, End, Clipping };
    uint32_t e = 0, c = 0, d = 0, b = 0;       // cursors: ext-CIGAR, CIGAR text, MD:Z, BAM operations
    // The reference keeps one counter per operation kind, bumps the current kind's at every step and clears the
    // previous kind's at every change of kind: what it reads at a change - the previous kind's counter - is the
    // length of the run that has just ended, so one running length serves (an array indexed by the kind would be
    // a chain of selects per step here).  Matches and mismatches are counted by run_m instead, as there.
    uint32_t run = 0, ended_run = 0;
    uint32_t run_m = 0, run_md = 0;            // cnt_match_mismatch, cnt_match_ignoring_insert
    int cur = Begin, prev = Begin;
    bool overflow = false;
    const bool by4 = str_aligned(ext);
    while (cur != End) {
        const uint32_t ch = e < ext_stride ? ext[e] : 0u;   // a slot without a NUL ends at its stride
        // 12 bytes of head room cover the longest single step (a count and two symbols)
        if (c + 12 > cigar_stride || d + 14 > mdz_stride || (cb && b + 1 > bin_stride)) {
            overflow = true;
            break;
        }
        // inside a run of matches four more dots change three counts and nothing else (no flush: cur == prev)
        if (by4 && prev == Match && e + 8 <= ext_stride && str4(ext, e) == FOUR_DOTS) {
            run_m += 4;
            run_md += 4;
            e += 4;
            continue;
        }
        if (ch == '$') {
            cur = Clipping;
            ++e;
        } else if (ch == 0) {
            cur = End;
            if (run_md > 0)
                d += put_decimal(md + d, run_md);
        } else if (ch == '.') {
            ++e;
            cur = Match;
        } else if (ch == '#') {
            e += 2;
            cur = Insertion;
        } else {
            d += put_decimal(md + d, run_md);
            run_md = 0;
            if (ch == '^') {
                md[d++] = (uint8_t)ch;
                md[d++] = e + 1 < ext_stride ? ext[e + 1] : 0;
                e += 2;
                cur = Deletion;
            } else {
                md[d++] = (uint8_t)ch;
                ++e;
                cur = Mismatch;
            }
        }
        ended_run = run;
        run = cur == prev ? run + 1 : 1;
        if (cur == Match) {
            ++run_m;
            ++run_md;
        } else if (cur == Mismatch) {
            ++run_m;
        }
        if (cur != prev && prev != Begin) {
            uint32_t count = 0, op = 0;
            bool flush = false;
            if (prev == Clipping) {
                count = ended_run, op = 4, flush = true;    // S_op
            } else if (prev == Insertion) {
                count = ended_run, op = 1, flush = true;   // I_op
            } else if (prev == Deletion) {
                count = ended_run, op = 2, flush = true;    // D_op
            } else if ((prev == Match && cur != Mismatch) || (prev == Mismatch && cur != Match)) {
                count = run_m, op = 0, flush = true;                 // M_op
                run_m = 0;
            }
            if (flush) {
                c += put_decimal(cg + c, count);
                cg[c++] = (uint8_t)(0x534e44494dull >> (8 * op));   // "MIDNS"[op]
                if (cb)
                    cb[b++] = count << 4 | op;
            }
        }
        prev = cur;
    }
    if (overflow) {
        raise_flag(flags, FLAG_CIGAR);
        c = d = b = 0;
    }
    cg[c] = 0;
    md[d] = 0;
    wamd_sam_cigar_len l;
    l.cigar_len = c;
    l.mdz_len = d;
    l.cigar_bin_len = b;
    l.reserved = overflow ? 1u : 0u;
    return l;
}

__global__ __launch_bounds__(256) void sam_cigar_kernel(uint32_t n, const uint8_t* __restrict__ ext_cigars,
                                                        uint32_t ext_stride, uint8_t* __restrict__ cigars,
                                                        uint32_t cigar_stride, uint8_t* __restrict__ mdz,
                                                        uint32_t mdz_stride, uint32_t* __restrict__ cigar_bin,
                                                        uint32_t bin_stride, wamd_sam_cigar_len* __restrict__ lens,
                                                        uint32_t* __restrict__ flags)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n)
        return;
    lens[gid] = sam_cigar_one(ext_cigars + (uint64_t)gid * ext_stride, ext_stride, cigars + (uint64_t)gid * cigar_stride,
                              cigar_stride, mdz + (uint64_t)gid * mdz_stride, mdz_stride,
                              cigar_bin ? cigar_bin + (uint64_t)gid * bin_stride : nullptr, bin_stride, flags);
}

// ---- rows L, B, L, X behind one another on an LDS copy of the ext-CIGARs --------------------------
// clipIllegalPositions, clipBoundaryPositions, clipOverlappingPairs and extract_cigar_and_mdz
// (sam.cpp:1889-1900, 2002-2007) for 64 consecutive slots (32 pairs) per wave - all four behind one another
// (finish_pairs) or any one of them (the stand-alone entry points of these rows: on the slots in HBM every byte
// a lane touches is its own cache line, and lines do not survive in L1 between a lane's accesses - 8 to 37 KB of
// line traffic per 150-byte string).  The wave copies its slots from global memory into LDS with coalesced loads
// (the slots are contiguous), every lane then runs the routines on its own copy - the per-record state machines
// above, unchanged, so the bytes they produce are the same - and the wave writes the slots back once.  A lane's
// copy starts at an odd dword stride, so lanes reading the same offset of their strings hit different banks.
// Records travel between the two lanes of a pair by shuffles.
enum { POST_ILLEGAL = 1, POST_BOUNDARY = 2, POST_PAIRS = 4, POST_SAM = 8,
       POST_LEN_IN_RESERVED = 16,   // lens[].reserved = length of the ext-CIGAR + 2 (finish_pairs' bookkeeping)
       POST_TWO_HALVES = 32 };      // test hook: never pack

__global__ __launch_bounds__(64, 4) void post_passes_kernel(uint32_t n, const wamd_clip_illegal_task* __restrict__ it,
                                                         const wamd_clip_pair_task* __restrict__ pt,
                                                         wamd_clip_record* __restrict__ records, uint8_t* __restrict__ ext,
                                                         uint32_t ext_stride, const uint32_t* __restrict__ cap_hint,
                                                         uint32_t budget_dw, wamd_scoring sc, uint32_t passes,
                                                         uint8_t* __restrict__ cigars, uint32_t cigar_stride,
                                                         uint8_t* __restrict__ mdz, uint32_t mdz_stride,
                                                         uint32_t* __restrict__ cigar_bin, uint32_t bin_stride,
                                                         wamd_sam_cigar_len* __restrict__ lens, int2* __restrict__ mid,
                                                         uint32_t* __restrict__ flags)
{
    extern __shared__ uint32_t post_lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t base = blockIdx.x * 64;
    const uint32_t here = n - base < 64u ? n - base : 64u;   // slots of this wave (pairs stay together: base is even)
    const uint32_t slot = base + lane;
    const bool live = lane < here;
    const uint32_t full_dw = (ext_stride / 4) | 1u;   // odd: lanes at the same offset of their strings hit different banks
    // What a lane's string needs.  The wave packs its strings by their lengths: the LDS of a wave, not its
    // registers, is what limits the waves per CU here.  A caller that knows how long the strings are (and can get: a
    // clipping pass never lengthens one, a perfect match may receive read_length dots) says so per slot; otherwise
    // the wave measures them, one slot after the other with the lanes side by side on its dwords.  Both strings of
    // a pair get the room of the longer one (the pair routine takes one capacity).  When the packed strings do not
    // fit the budget, the wave serves its 32 even / odd pairs in two halves at the full stride.
    //
    // Staging: a lane fetches ITS OWN slot, sixteen bytes at a time into registers (strings up to 16 * POST_CH
    // bytes), all requests in flight at once - a loop in which the wave copies one slot after the other, coalesced
    // as it is, waits for memory sixty-four times in a row, and that wait was most of these kernels' time.  The
    // loop form remains for longer strings, unaligned strides and the two-halves case.
    constexpr int POST_CH = 12;
    const bool regs_ok = (ext_stride & 15u) == 0 && ((uintptr_t)ext & 15u) == 0;
    wamd_u32x4 rg[POST_CH];
    uint32_t my_dw = 0;
    bool in_regs = regs_ok;   // (wave-uniform from here on)
    if (cap_hint) {
        uint32_t c = 0;
        if (live) {
            c = cap_hint[slot] < ext_stride ? cap_hint[slot] : ext_stride;
            my_dw = ((c + 3) / 4) | 1u;
            my_dw = my_dw > full_dw ? full_dw : my_dw;
        }
        {   // (a pair shares the longer capacity, see below: what is fetched must cover it)
            const uint32_t other = (uint32_t)__shfl_xor((int)my_dw, 1);
            my_dw = my_dw > other ? my_dw : (live ? other : 0u);
        }
        in_regs = regs_ok && !__any(live && my_dw * 4 > 16u * POST_CH);
        if (in_regs) {
            const wamd_u32x4* src = reinterpret_cast<const wamd_u32x4*>(ext + (uint64_t)(live ? slot : base) * ext_stride);
#pragma unroll
            for (int q = 0; q < POST_CH; ++q)
                if (live && 16u * q < my_dw * 4 && 16u * q < ext_stride)
                    rg[q] = src[q];
        }
    } else {
        uint32_t my_len = ext_stride;   // bytes up to and including the dword that holds the NUL
        bool found = false;
        if (regs_ok) {
            const wamd_u32x4* src = reinterpret_cast<const wamd_u32x4*>(ext + (uint64_t)(live ? slot : base) * ext_stride);
#pragma unroll
            for (int q = 0; q < POST_CH; ++q)
                if (live && 16u * q < ext_stride)
                    rg[q] = src[q];
#pragma unroll
            for (int q = 0; q < POST_CH; ++q)
                if (live && !found && 16u * q < ext_stride) {
                    const uint32_t w[4] = {rg[q].x, rg[q].y, rg[q].z, rg[q].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!found && ((w[j] - 0x01010101u) & ~w[j] & 0x80808080u)) {
                            found = true;
                            my_len = 16u * q + 4u * j + 4u;
                        }
                }
            if (ext_stride <= 16u * POST_CH)
                found = true;   // the whole slot is in the registers
        }
        const bool measured = regs_ok && !__any(live && !found);
        if (!measured) {   // the wave measures its slots one after the other, the lanes side by side on a slot's dwords
            for (uint32_t sl = 0; sl < here; ++sl) {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(ext + (uint64_t)(base + sl) * ext_stride);
                uint32_t len = ext_stride;
                for (uint32_t i0 = 0; i0 < ext_stride / 4; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    bool z = false;
                    if (i < ext_stride / 4) {
                        const uint32_t v = src[i];
                        z = ((v - 0x01010101u) & ~v & 0x80808080u) != 0;
                    }
                    const uint64_t m = __ballot(z);
                    if (m) {
                        len = 4 * (i0 + (uint32_t)__builtin_ctzll(m)) + 4;
                        break;
                    }
                }
                if (lane == sl)
                    my_len = len;
            }
        }
        if (live) {
            // a perfect match may be filled with read_length dots (clipIllegalPositions, 614-618), and the tail loop
            // of that routine pads up to read_length
            // (so may clipOverlappingPairs, 1029-1036)
            uint32_t grow = (passes & POST_ILLEGAL) ? (uint32_t)it[slot].read_length + 1u : 0u;
            if (passes & POST_PAIRS) {
                const uint32_t g2 = (uint32_t)pt[slot >> 1].read_length[lane & 1u] + 1u;
                grow = grow > g2 ? grow : g2;
            }
            uint32_t c = (my_len > grow ? my_len : grow) + 8u;
            c = c < ext_stride ? c : ext_stride;
            my_dw = ((c + 3) / 4) | 1u;
            my_dw = my_dw > full_dw ? full_dw : my_dw;
        }
        in_regs = measured && !__any(live && my_dw * 4 > 16u * POST_CH && ext_stride > 16u * POST_CH);
    }
    {
        const uint32_t other = (uint32_t)__shfl_xor((int)my_dw, 1);
        my_dw = my_dw > other ? my_dw : (live ? other : 0u);
    }
    uint32_t incl = my_dw;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= (uint32_t)d)
            incl += o;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    // lanes served together when the strings are not packed: as many (a power of two, pairs stay together) as fit
    // the budget at the full stride - 32 unless the caller's stride is long (the launcher caps the budget so that
    // the stride does not decide how many waves a CU holds)
    uint32_t per = 32, per_sh = 5;
    while (per > 2 && per * full_dw > budget_dw) {
        per >>= 1;
        --per_sh;
    }
    const int halves = (total <= budget_dw && !(passes & POST_TWO_HALVES)) ? 1 : (int)(64u / per);
    for (int h = 0; h < halves; ++h) {
        // lanes of this pass, where their strings start and how many dwords they hold
        const bool mine = halves == 1 || (lane >> per_sh) == (uint32_t)h;
        uint32_t ndw = mine ? my_dw : 0u, off = incl - my_dw;
        if (halves != 1) {
            ndw = (mine && live) ? full_dw : 0u;
            off = (lane & (per - 1u)) * full_dw;
        }
        const uint32_t cap = ndw * 4 < ext_stride ? ndw * 4 : ext_stride;   // bytes of the string the routines may use
        const uint32_t lim_dw = ndw < ext_stride / 4 ? ndw : ext_stride / 4;   // dwords that travel
        const bool by_regs = in_regs && halves == 1;
        if (by_regs) {
#pragma unroll
            for (int q = 0; q < POST_CH; ++q) {
                const uint32_t w[4] = {rg[q].x, rg[q].y, rg[q].z, rg[q].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4u * q + j < lim_dw)
                        post_lds[off + 4 * q + j] = w[j];
            }
        } else {
            for (uint32_t sl = 0; sl < here; ++sl) {   // coalesced: the wave copies one string after the other
                const uint32_t n_s = (uint32_t)__builtin_amdgcn_readlane((int)lim_dw, (int)sl);
                if (!n_s)
                    continue;
                const uint32_t o_s = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)sl);
                const uint32_t* src = reinterpret_cast<const uint32_t*>(ext + (uint64_t)(base + sl) * ext_stride);
                for (uint32_t i = lane; i < n_s; i += 64)
                    post_lds[o_s + i] = src[i];
            }
        }
        __syncthreads();
        const bool act = live && mine;
        uint8_t* ec = reinterpret_cast<uint8_t*>(post_lds + off);
        wamd_clip_record d{};
        if (act && (passes & (POST_ILLEGAL | POST_BOUNDARY | POST_PAIRS))) {
            d = records[slot];
            if (passes & POST_ILLEGAL)
                clip_illegal_one(d, it[slot], ec, cap, sc, flags);
            if (passes & POST_BOUNDARY)
                clip_boundary_one(d, ec, cap, sc);
            if (mid)   // where the mapping lies before the pair is clipped (finish_pairs: isDistant of the mate)
                mid[slot] = make_int2(d.ref_seq_pos, (int)d.ref_length);
        }
        __syncthreads();   // every lane's string is as its own passes left it before a partner touches it
        if (passes & POST_PAIRS) {
            // the even lane of a pair takes its partner's record, clips both strings, hands the record back
            uint32_t w[8], v[8];
            memcpy(w, &d, 32);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = (uint32_t)__shfl_down((int)w[q], 1);
            const uint32_t off1 = (uint32_t)__shfl_down((int)off, 1);
            wamd_clip_record rec[2];
            rec[0] = d;
            memcpy(&rec[1], v, 32);
            if (act && !(lane & 1u) && lane + 1 < here)
                clip_pair_one(pt[slot >> 1], rec, ec, reinterpret_cast<uint8_t*>(post_lds + off1), cap, flags);
            memcpy(w, &rec[1], 32);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = (uint32_t)__shfl_up((int)w[q], 1);
            if (lane & 1u)
                memcpy(&d, v, 32);
            else
                d = rec[0];
            __syncthreads();
        }
        if (act) {
            if (passes & POST_SAM) {
                wamd_sam_cigar_len l = sam_cigar_one(ec, cap, cigars + (uint64_t)slot * cigar_stride, cigar_stride,
                                                     mdz + (uint64_t)slot * mdz_stride, mdz_stride,
                                                     cigar_bin ? cigar_bin + (uint64_t)slot * bin_stride : nullptr, bin_stride,
                                                     flags);
                // the length of the ext-CIGAR as the passes left it, + 2 (0 / 1 = not measured), for the caller's
                // bookkeeping: here it costs LDS reads, from the slots in memory a line fetch per lane and byte
                if (passes & POST_LEN_IN_RESERVED)
                    l.reserved = bounded_len(ec, cap) + 2;
                lens[slot] = l;
            }
            if (passes & (POST_ILLEGAL | POST_BOUNDARY | POST_PAIRS))
                records[slot] = d;
        }
        __syncthreads();
        if (passes & (POST_ILLEGAL | POST_BOUNDARY | POST_PAIRS)) {   // (extract_cigar_and_mdz only reads)
            if (by_regs) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(ext + (uint64_t)(live ? slot : base) * ext_stride);
#pragma unroll
                for (int q = 0; q < POST_CH; ++q) {
                    if (4u * q + 4u <= lim_dw) {
                        wamd_u32x4 v = {post_lds[off + 4 * q], post_lds[off + 4 * q + 1], post_lds[off + 4 * q + 2],
                                        post_lds[off + 4 * q + 3]};
                        *reinterpret_cast<wamd_u32x4*>(dst + 4 * q) = v;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (4u * q + j < lim_dw)
                                dst[4 * q + j] = post_lds[off + 4 * q + j];
                    }
                }
            } else {
                for (uint32_t sl = 0; sl < here; ++sl) {
                    const uint32_t n_s = (uint32_t)__builtin_amdgcn_readlane((int)lim_dw, (int)sl);
                    if (!n_s)
                        continue;
                    const uint32_t o_s = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)sl);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(ext + (uint64_t)(base + sl) * ext_stride);
                    for (uint32_t i = lane; i < n_s; i += 64)
                        dst[i] = post_lds[o_s + i];
                }
            }
            __syncthreads();
        }
    }
}

static bool rows_can_stage(const void* slots, uint32_t stride)
{
    return (stride & 3u) == 0 && stride >= 4 && stride <= 2000 && ((uintptr_t)slots & 3u) == 0;
}

static int launch_post_kernel(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* it, const wamd_clip_pair_task* pt,
                              wamd_clip_record* rec, uint8_t* ext, uint32_t ext_stride, const uint32_t* cap_hint,
                              uint32_t passes, uint8_t* cigars, uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride,
                              uint32_t* bin, uint32_t bin_stride, wamd_sam_cigar_len* lens, hipStream_t s,
                              int2* mid = nullptr)
{
    if (n == 0)
        return WAMD_OK;
    if (ctx->test_hooks & HOOK_POST_NO_HINTS) {   // test hook: the two-halves path of every wave
        cap_hint = nullptr;
        passes |= POST_TWO_HALVES;
    }
    // LDS of a wave: 32 strings at the full (odd) stride - what a half needs when the strings are not packed
    // ... capped at 11.4 KB (fourteen waves per CU; what 32 strings of 352 bytes take): longer strides are served
    // in smaller groups when their strings do not pack
    const uint32_t full_dw = (ext_stride / 4) | 1u;
    uint32_t budget_dw = 32u * full_dw;
    if (budget_dw > 2912u)
        budget_dw = 2912u > 2u * full_dw ? 2912u : 2u * full_dw;
    hipLaunchKernelGGL(post_passes_kernel, dim3((n + 63) / 64), dim3(64), (size_t)budget_dw * 4, s, n, it, pt, rec, ext,
                       ext_stride, cap_hint, budget_dw, ctx->scoring, passes, cigars, cigar_stride, mdz, mdz_stride, bin,
                       bin_stride, lens, mid, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

__global__ __launch_bounds__(256) void post_mid_kernel(uint32_t n, const wamd_clip_record* __restrict__ rec,
                                                       int2* __restrict__ mid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        mid[i] = make_int2(rec[i].ref_seq_pos, (int)rec[i].ref_length);
}

int launch_clip_boundary(wamd_ctx* ctx, uint32_t n, wamd_clip_record* d_records, uint8_t* d_cigars,
                         uint32_t cigar_stride, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (rows_can_stage(d_cigars, cigar_stride))
        return launch_post_kernel(ctx, n, nullptr, nullptr, d_records, d_cigars, cigar_stride, nullptr, POST_BOUNDARY,
                                  nullptr, 0, nullptr, 0, nullptr, 0, nullptr, s);
    hipLaunchKernelGGL(clip_boundary_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_records, d_cigars,
                       cigar_stride, ctx->scoring);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_clip_illegal(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_tasks, wamd_clip_record* d_records,
                        uint8_t* d_cigars, uint32_t cigar_stride, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (rows_can_stage(d_cigars, cigar_stride))
        return launch_post_kernel(ctx, n, d_tasks, nullptr, d_records, d_cigars, cigar_stride, nullptr, POST_ILLEGAL,
                                  nullptr, 0, nullptr, 0, nullptr, 0, nullptr, s);
    hipLaunchKernelGGL(clip_illegal_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_tasks, d_records, d_cigars,
                       cigar_stride, ctx->scoring, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_clip_pairs(wamd_ctx* ctx, uint32_t n_pairs, const wamd_clip_pair_task* d_tasks, wamd_clip_record* d_records,
                      uint8_t* d_cigars, uint32_t cigar_stride, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (rows_can_stage(d_cigars, cigar_stride) && n_pairs <= 0x7fffffffu)
        return launch_post_kernel(ctx, 2 * n_pairs, nullptr, d_tasks, d_records, d_cigars, cigar_stride, nullptr, POST_PAIRS,
                                  nullptr, 0, nullptr, 0, nullptr, 0, nullptr, s);
    hipLaunchKernelGGL(clip_pairs_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, s, n_pairs, d_tasks, d_records,
                       d_cigars, cigar_stride, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

// the fused form when half a wave's slots fit LDS (64 KB: ext_stride up to 2000), else the four kernels in turn
int launch_post_passes(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_it, const wamd_clip_pair_task* d_pt,
                       wamd_clip_record* d_rec, uint8_t* d_ext, uint32_t ext_stride, const uint32_t* d_cap_hint,
                       int do_boundary, int do_pairs, uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz,
                       uint32_t mdz_stride, wamd_sam_cigar_len* d_lens, int2* d_mid, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (rows_can_stage(d_ext, ext_stride) && (n & 1u) == 0)
        return launch_post_kernel(ctx, n, d_it, d_pt, d_rec, d_ext, ext_stride, d_cap_hint,
                                  POST_ILLEGAL | (do_boundary ? POST_BOUNDARY : 0) | (do_pairs ? POST_PAIRS : 0) | POST_SAM |
                                      POST_LEN_IN_RESERVED,
                                  d_cigars, cigar_stride, d_mdz, mdz_stride, nullptr, 0, d_lens, s, d_mid);
    int rc = launch_clip_illegal(ctx, n, d_it, d_rec, d_ext, ext_stride, s);
    if (!rc && do_boundary)
        rc = launch_clip_boundary(ctx, n, d_rec, d_ext, ext_stride, s);
    if (!rc && d_mid && n) {
        hipLaunchKernelGGL(post_mid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_rec, d_mid);
        WAMD_HIP(hipGetLastError());
    }
    if (!rc && do_pairs)
        rc = launch_clip_pairs(ctx, n / 2, d_pt, d_rec, d_ext, ext_stride, s);
    if (!rc)
        rc = launch_sam_cigar(ctx, n, d_ext, ext_stride, d_cigars, cigar_stride, d_mdz, mdz_stride, nullptr, 0, d_lens, s);
    return rc;
}

int launch_sam_cigar(wamd_ctx* ctx, uint32_t n, const uint8_t* d_ext, uint32_t ext_stride, uint8_t* d_cigars,
                     uint32_t cigar_stride, uint8_t* d_mdz, uint32_t mdz_stride, uint32_t* d_bin, uint32_t bin_stride,
                     wamd_sam_cigar_len* d_lens, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (rows_can_stage(d_ext, ext_stride))
        return launch_post_kernel(ctx, n, nullptr, nullptr, nullptr, const_cast<uint8_t*>(d_ext), ext_stride, nullptr, POST_SAM,
                                  d_cigars, cigar_stride, d_mdz, mdz_stride, d_bin, bin_stride, d_lens, s);
    hipLaunchKernelGGL(sam_cigar_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_ext, ext_stride, d_cigars,
                       cigar_stride, d_mdz, mdz_stride, d_bin, bin_stride, d_lens, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_serialize(wamd_ctx* ctx, uint32_t n, const wamd_mapping* d_mappings, const uint8_t* d_direction,
                     uint8_t* d_out, uint8_t* d_sizes, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(serialize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_mappings, d_direction,
                       d_out, d_sizes);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

} // namespace wamd
