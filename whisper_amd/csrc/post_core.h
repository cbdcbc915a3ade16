// post_core.h - rows B, L, X of the path on one ext-CIGAR: clipBoundaryPositions, clipIllegalPositions,
// clipOverlappingPairs (clipLeft / clipRight) and extract_cigar_and_mdz (mapper/soft_clipping.cpp:592-1151,
// mapper/sam.cpp:805-1051), as EVENT-DRIVEN routines over a string held as dwords plus a bitmap of its events.
//
// An ext-CIGAR is mostly '.', and everything the four reference routines compute is a function of where the other
// symbols ("events": a mismatch letter, '#'x, '^'x, '$') lie and of the lengths of the dot runs between them.  So a
// string is digested once, a dword at a time (SWAR byte tests), into a bitmap with one bit per byte - set where a symbol
// other than '.' starts - and every routine then hops from event to event:
//   * the running affine score is folded in the reference's order of additions; a run of n dots is n additions of
//     the match score, taken in one multiplication when that is exact (the run's partial sums all lie on a 2^-20 grid
//     inside 2^22), one by one otherwise - the sum is the same double either way;
//   * "the position where the k-th reference-consuming symbol ends" (clipLeft, clipRight, the chromosome borders of
//     clipIllegalPositions) is an event walk with the dot runs taken arithmetically;
//   * a clipping never re-reads the string: the clipped stretch becomes a run of '$' whose length follows from the
//     counted symbol classes, the kept stretch moves down by a funnel-shift copy - of the bytes and, with the same
//     routine at one eighth of the offsets, of the event bitmap, so the next pass needs no new digest;
//   * CIGAR and MD:Z come from the event walk as runs, never from a per-byte state machine.
// One string per lane; the loops below run in lock step across a wave, as long as the lane with the most events.
// The same header compiles for the host (tests/native): plain C++, no wave intrinsics in here.
//
// Grammar.  A string is ('.' | letter | '#' letter | '^' letter | '$')*.  Two adjacent marks, a mark in front of the
// NUL and a string without a NUL inside its capacity are refused (the reference reads past the string there, or never
// leaves its loop); any other byte counts as a letter.
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/whisper_amd.h"

#if defined(__HIPCC__)
#define PC_FN __device__ __forceinline__
#define PC_MEMBER __device__ __forceinline__
#else
#define PC_FN static inline
#define PC_MEMBER inline
#endif

namespace wamd {
namespace post {

enum : uint32_t { PC_BAD_STRING = 1u << 5 /* FLAG_ARG */, PC_NO_ROOM = 1u << 6 /* FLAG_CIGAR */ };

struct Str {
    uint32_t* dw;    // the string, four bytes per dword, cap / 4 dwords
    uint32_t* ev;    // bit i set: byte i starts a symbol other than '.'; (cap / 4 + 7) / 8 + 1 words
    uint32_t cap;    // bytes the string may occupy, NUL included (a multiple of 4)
    uint32_t len;    // bytes before the NUL
};

PC_FN uint32_t zero_bytes(uint32_t x) { return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); }
PC_FN uint32_t eq_bytes(uint32_t x, uint32_t c4) { return zero_bytes(x ^ c4); }
// bits 7, 15, 23, 31 -> bits 0 .. 3
PC_FN uint32_t pack4(uint32_t m80) { return (((m80 >> 7) * 0x00204081u) >> 21) & 0xfu; }
PC_FN uint32_t byte_at(const Str& s, uint32_t i) { return (s.dw[i >> 2] >> (8u * (i & 3u))) & 0xffu; }
PC_FN uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// ---- bit ranges of a dword array (bytes: the same with eight times the offsets) --------------------------------
// copy n bits from src to dst, dst <= src, front to back, in place
PC_FN void bits_copy(uint32_t* a, uint32_t last_word, uint32_t dst, uint32_t src, uint32_t n)
{
    if (n == 0 || dst == src)
        return;
    const uint32_t d0 = dst >> 5, d1 = (dst + n - 1) >> 5, delta = src - dst;
    for (uint32_t d = d0; d <= d1; ++d) {
        const uint32_t sbit = 32u * d + delta, si = sbit >> 5, sh = sbit & 31u;
        const uint32_t lo = a[si < last_word ? si : last_word], hi = a[si + 1 < last_word ? si + 1 : last_word];
        const uint32_t v = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
        uint32_t m = 0xffffffffu;
        if (d == d0)
            m &= 0xffffffffu << (dst & 31u);
        if (d == d1)
            m &= 0xffffffffu >> (31u - ((dst + n - 1) & 31u));
        a[d] = (a[d] & ~m) | (v & m);
    }
}

PC_FN void bits_fill(uint32_t* a, uint32_t dst, uint32_t n, uint32_t pattern)
{
    if (n == 0)
        return;
    const uint32_t d0 = dst >> 5, d1 = (dst + n - 1) >> 5;
    for (uint32_t d = d0; d <= d1; ++d) {
        uint32_t m = 0xffffffffu;
        if (d == d0)
            m &= 0xffffffffu << (dst & 31u);
        if (d == d1)
            m &= 0xffffffffu >> (31u - ((dst + n - 1) & 31u));
        a[d] = (a[d] & ~m) | (pattern & m);
    }
}

PC_FN uint32_t str_words(const Str& s) { return s.cap / 4; }
PC_FN uint32_t ev_words(const Str& s) { return (s.cap / 4 + 7) / 8 + 1; }

// n bytes of the string from src down to dst, with their events
PC_FN void move_bytes(Str& s, uint32_t dst, uint32_t src, uint32_t n)
{
    bits_copy(s.dw, str_words(s) - 1, 8 * dst, 8 * src, 8 * n);
    bits_copy(s.ev, ev_words(s) - 1, dst, src, n);
}

// n bytes c at dst; every one of them an event, or none
PC_FN void fill_bytes(Str& s, uint32_t dst, uint32_t n, uint32_t c, bool events)
{
    bits_fill(s.dw, 8 * dst, 8 * n, c * 0x01010101u);
    bits_fill(s.ev, dst, n, events ? 0xffffffffu : 0u);
}

PC_FN void set_len(Str& s, uint32_t len)
{
    s.len = len;
    bits_fill(s.dw, 8 * len, 8, 0u);
}

// ---- the digest: events and length of a string -----------------------------------------------------------------
// Returns 0, or PC_BAD_STRING when the string breaks the grammar (see the head of this file).
PC_FN uint32_t digest(Str& s)
{
    // The byte tests assume 7-bit bytes (x + 0x7f sets bit 7 of a byte iff it is not 0, without a carry into the next
    // one); a byte above 0x7f inside the string is refused, one behind the NUL can only disturb what lies behind it.
    constexpr uint32_t H = 0x80808080u, L = 0x7f7f7f7fu;
    const uint32_t nd = s.cap / 4;
    uint32_t acc = 0, carry = 0, bad = 0;
    bool open = true;
    s.len = s.cap;
    for (uint32_t k = 0; k < nd; ++k) {
        const uint32_t w = s.dw[k];
        const uint32_t z = ~(w + L) & H;                                        // bytes that are 0
        const uint32_t not_dot = (w ^ 0x2e2e2e2eu) + L;                         // bit 7 of a byte: it is not '.'
        const uint32_t mark = ~(((w ^ 0x23232323u) + L) & ((w ^ 0x5e5e5e5eu) + L)) & H;     // '#' or '^'
        uint32_t valid = open ? H : 0u;
        if (open && z) {
            const uint32_t j = (uint32_t)__builtin_ctz(z) >> 3;     // first zero byte of this dword
            valid = j ? H & ((1u << (8u * j)) - 1u) : 0u;
            s.len = 4 * k + j;
            open = false;
            // a mark in front of the NUL has no symbol behind it
            bad |= j ? (mark >> (8u * (j - 1))) & 0x80u : carry;
        }
        const uint32_t pay = (mark << 8) | carry;
        bad |= ((mark & pay) | w) & valid;                          // two marks in a row; a byte above 0x7f
        const uint32_t ev80 = not_dot & ~pay & valid;
        acc |= (((ev80 >> 7) | (ev80 >> 14) | (ev80 >> 21) | (ev80 >> 28)) & 0xfu) << (4u * (k & 7u));
        if ((k & 7u) == 7u || k + 1 == nd) {
            s.ev[k >> 3] = acc;
            acc = 0;
        }
        carry = (mark & valid) >> 24;
    }
    s.ev[ev_words(s) - 1] = 0;
    return (bad || open) ? PC_BAD_STRING : 0u;
}

// smallest event >= pos, or len
PC_FN uint32_t next_event(const Str& s, uint32_t pos)
{
    if (pos >= s.len)
        return s.len;
    uint32_t w = pos >> 5;
    const uint32_t nw = (s.len + 31) >> 5;
    uint32_t bits = s.ev[w] & (0xffffffffu << (pos & 31u));
    while (!bits) {
        if (++w >= nw)
            return s.len;
        bits = s.ev[w];
    }
    const uint32_t e = 32u * w + (uint32_t)__builtin_ctz(bits);
    return e < s.len ? e : s.len;
}

// largest event <= pos, or -1
PC_FN int prev_event(const Str& s, int pos)
{
    if (pos < 0)
        return -1;
    uint32_t w = (uint32_t)pos >> 5;
    uint32_t bits = s.ev[w] & (0xffffffffu >> (31u - ((uint32_t)pos & 31u)));
    while (!bits) {
        if (w == 0)
            return -1;
        bits = s.ev[--w];
    }
    return (int)(32u * w + 31u - (uint32_t)__builtin_clz(bits));
}

// ---- the running score ---------------------------------------------------------------------------------------------
struct Fold {
    wamd_scoring sc;
    bool grid;     // the match score is a multiple of 2^-20 of magnitude <= 2^10
};

PC_FN Fold make_fold(const wamd_scoring& sc)
{
    Fold f;
    f.sc = sc;
    const double t = sc.match * 1048576.0;
    f.grid = t == rint(t) && fabs(sc.match) <= 1024.0;
    return f;
}

// cur + a, n times (n >= 1), as n additions in a row would leave it
PC_FN double add_run(double cur, double a, uint32_t n, bool grid)
{
    const double t = cur * 1048576.0;
    if (grid && t == rint(t) && fabs(cur) < 1048576.0 && n <= 2048u)
        return cur + (double)n * a;     // every partial sum is a multiple of 2^-20 below 2^22: all n additions are exact
    for (uint32_t i = 0; i < n; ++i)
        cur += a;
    return cur;
}

PC_FN uint16_t u16(uint32_t v) { return (uint16_t)v; }

// ---- row B: CSoftClipping::clipBoundaryPositions (soft_clipping.cpp:827-994) ------------------------------------------
// The prefix score of the alignment between the clippings, its first minimum and first maximum; the tail behind the
// maximum goes when it loses more than a clipping costs, the head up to the minimum when it is below the clipping
// penalty; positions are byte offsets of a symbol's last byte from the scan's start, as the reference counts them.
PC_FN uint32_t clip_boundary_scan(Str& s, wamd_clip_record& d, const Fold& f)
{
    const wamd_scoring& sc = f.sc;
    const uint32_t start = d.left_clipping;
    double cur = 0, mini = 1.7976931348623157e308, maxi = -1.7976931348623157e308;
    int mini_pos = -1, maxi_pos = -1;
    auto check = [&](double v, uint32_t at) {
        if (v < mini) {
            mini = v;
            mini_pos = (int)(at - start);
        }
        if (v > maxi) {
            maxi = v;
            maxi_pos = (int)(at - start);
        }
    };
    uint32_t pos = start, end = s.len;
    bool in_indel = false;
    for (;;) {
        const uint32_t e = next_event(s, pos);
        const uint32_t c = e < s.len ? byte_at(s, e) : 0u;
        const uint32_t n = e - pos;
        if (n) {     // a run of matches: monotone, so only its ends can be new extremes (either sign of the score)
            cur += sc.match;
            check(cur, pos);
            if (n > 1) {
                cur = add_run(cur, sc.match, n - 1, f.grid);
                check(cur, e - 1);
            }
            in_indel = false;
        }
        if (e >= s.len || c == '$') {
            end = e;
            break;
        }
        if (c == '#') {
            cur += in_indel ? sc.gap_ins_extend : sc.gap_ins_open;
            in_indel = true;
            pos = e + 2;
        } else if (c == '^') {
            cur += in_indel ? sc.gap_del_extend : sc.gap_del_open;
            in_indel = true;
            pos = e + 2;
        } else {
            cur += sc.mismatch;
            in_indel = false;
            pos = e + 1;
        }
        check(cur, pos - 1);
    }
    if (mini_pos + 20 > maxi_pos)
        return 0;
    const bool right = maxi + (d.right_clipping > 0 ? 0.0 : sc.clipping) > cur;
    const bool left = mini < (d.left_clipping > 0 ? 0.0 : sc.clipping);
    const uint32_t g = start + (uint32_t)mini_pos + 1, h = start + (uint32_t)maxi_pos + 1;
    // symbol classes inside the two stretches (dots = bytes that belong to no event)
    uint32_t lx = 0, li = 0, ld = 0, rx = 0, ri = 0, rd = 0;
    for (uint32_t p = start;;) {
        const uint32_t e = next_event(s, p);
        if (e >= end)
            break;
        const uint32_t c = byte_at(s, e);
        const bool is_i = c == '#', is_d = c == '^';
        if (left && e < g) {
            lx += !(is_i || is_d);
            li += is_i;
            ld += is_d;
        } else if (right && e >= h) {
            rx += !(is_i || is_d);
            ri += is_i;
            rd += is_d;
        }
        p = e + ((is_i || is_d) ? 2u : 1u);
    }
    uint32_t out = start, mid0 = start, mid1 = end, no_clippings = 0;
    uint32_t r_dollars = 0;
    if (right) {
        const uint32_t dots = (end - h) - rx - 2 * ri - 2 * rd;
        r_dollars = dots + rx + ri;
        d.right_clipping = u16(d.right_clipping + r_dollars);
        d.ref_length = u16(d.ref_length - (dots + rx + rd));
        d.err_edit_distance = u16(d.err_edit_distance - (rx + ri + rd));
        mid1 = h;
        ++no_clippings;
    }
    if (left) {
        const uint32_t dots = (g - start) - lx - 2 * li - 2 * ld;
        const uint32_t dollars = dots + lx + li;
        d.left_clipping = u16(d.left_clipping + dollars);
        d.ref_seq_pos += (int32_t)(dots + lx + ld);
        d.ref_length = u16(d.ref_length - (dots + lx + ld));
        d.err_edit_distance = u16(d.err_edit_distance - (lx + li + ld));
        fill_bytes(s, start, dollars, '$', true);
        out = start + dollars;
        mid0 = g;
        ++no_clippings;
    }
    if (left || right) {
        move_bytes(s, out, mid0, mid1 - mid0);
        out += mid1 - mid0;
        fill_bytes(s, out, r_dollars, '$', true);
        out += r_dollars;
        const uint32_t tail = s.len - end;
        move_bytes(s, out, end, tail);
        set_len(s, out + tail);
    }
    d.score = (0.0 > maxi ? 0.0 : maxi) - (0.0 < mini ? 0.0 : mini) + no_clippings * sc.clipping;
    d.no_clippings = no_clippings;
    d.cigar_len = s.len;
    return 0;
}

PC_FN uint32_t clip_boundary(Str& s, wamd_clip_record& d, const Fold& f)
{
    d.no_clippings = 0;
    d.cigar_len = s.len;
    if (d.err_edit_distance == 0)
        return 0;
    const uint32_t start = d.left_clipping;
    if (start > s.len)
        return PC_BAD_STRING;
    // left_clipping counts read symbols, the scan starts that many BYTES into the string: after clipIllegalPositions
    // has cut through a deletion in front of the chromosome the two differ, and the scan may start on the second
    // byte of an indel symbol, which it then reads as a mismatch.  Same here: that byte becomes an event for this
    // pass, and the events are taken afresh from what the pass leaves.
    const bool mid_symbol = start > 0 && start < s.len && !((s.ev[start >> 5] >> (start & 31u)) & 1u) && byte_at(s, start) != '.';
    if (mid_symbol)
        s.ev[start >> 5] |= 1u << (start & 31u);
    uint32_t rc = clip_boundary_scan(s, d, f);
    if (mid_symbol)
        rc |= digest(s);
    return rc;
}

// ---- row L: CSoftClipping::clipIllegalPositions (soft_clipping.cpp:592-729) ---------------------------------------------
// What hangs over either end of the chromosome becomes '$'; score and edit count are taken again over what is kept.
// The reference's three loops as three event walks: up to the first base of the chromosome (reference-consuming
// symbols are counted, insertions on the way go too), the kept part until the read or the chromosome is used up, and
// the tail, whose first ordinary symbol makes the routine pad with '$' up to read_length without moving on.
PC_FN uint32_t clip_illegal(Str& s, wamd_clip_record& d, const wamd_clip_illegal_task& t, const Fold& f)
{
    const wamd_scoring& sc = f.sc;
    const int32_t ref_size = t.ref_size;
    const int32_t mapping_last = d.ref_seq_pos + (int32_t)d.ref_length - 1;
    const int read_length = t.read_length;
    d.no_clippings = 0;
    d.cigar_len = s.len;
    const int to_left = d.ref_seq_pos <= 0 ? -d.ref_seq_pos + 1 : 0;
    const int to_right = mapping_last > ref_size ? mapping_last - ref_size : 0;
    if (!to_left && !to_right)
        return 0;
    if ((uint32_t)read_length + 1 > s.cap)
        return PC_NO_ROOM;
    if (t.perfect) {     // a perfect match has no ext-CIGAR yet: read_length dots
        fill_bytes(s, 0, (uint32_t)read_length, '.', false);
        set_len(s, (uint32_t)read_length);
    }
    int i_read = 0, i_ref = d.ref_seq_pos;
    uint32_t pos = 0;
    while (i_ref < 1) {     // before the chromosome's first base
        const uint32_t e = next_event(s, pos);
        if (e > pos) {
            const uint32_t take = umin(e - pos, (uint32_t)(1 - i_ref));
            i_ref += (int)take;
            i_read += (int)take;
            pos += take;
            continue;
        }
        if (e >= s.len)
            break;
        const uint32_t c = byte_at(s, e);
        if (c == '^') {
            ++i_ref;
            pos = e + 2;
        } else if (c == '#') {
            ++i_read;
            pos = e + 2;
        } else {
            ++i_read;
            ++i_ref;
            pos = e + 1;
        }
    }
    const uint32_t b1 = pos, c1 = (uint32_t)i_read;
    double score = 0;
    uint32_t ed = 0;
    bool gap = false;
    while (i_read < read_length && i_ref <= ref_size) {     // the part that stays
        const uint32_t e = next_event(s, pos);
        if (e > pos) {
            uint32_t take = umin(e - pos, (uint32_t)(read_length - i_read));
            take = umin(take, (uint32_t)(ref_size - i_ref + 1));
            score = add_run(score, sc.match, take, f.grid);
            i_read += (int)take;
            i_ref += (int)take;
            pos += take;
            gap = false;
            continue;
        }
        if (e >= s.len)
            break;
        const uint32_t c = byte_at(s, e);
        ++ed;
        if (c == '^') {
            score += gap ? sc.gap_del_extend : sc.gap_del_open;
            gap = true;
            ++i_ref;
            pos = e + 2;
        } else if (c == '#') {
            score += gap ? sc.gap_ins_extend : sc.gap_ins_open;
            gap = true;
            ++i_read;
            pos = e + 2;
        } else {     // a letter, or a '$' an earlier clipping left: scored as a mismatch
            score += sc.mismatch;
            gap = false;
            ++i_read;
            ++i_ref;
            pos = e + 1;
        }
    }
    const uint32_t b2 = pos;
    uint32_t c3 = 0;
    while (i_read < read_length) {     // behind the chromosome's last base
        const uint32_t e = next_event(s, pos);
        if (e >= s.len && e == pos)
            break;
        const uint32_t c = e > pos ? (uint32_t)'.' : byte_at(s, e);
        if (c == '^') {
            pos = e + 2;
        } else if (c == '#') {
            ++c3;
            ++i_read;
            pos = e + 2;
        } else {     // the reference does not advance here: '$' up to read_length
            c3 += (uint32_t)(read_length - i_read);
            i_read = read_length;
        }
    }
    const uint32_t kept = b2 - b1, new_len = c1 + kept + c3;
    if (new_len + 1 > s.cap)
        return PC_BAD_STRING;
    fill_bytes(s, 0, c1, '$', true);
    move_bytes(s, c1, b1, kept);
    fill_bytes(s, c1 + kept, c3, '$', true);
    set_len(s, new_len);
    d.left_clipping = u16(d.left_clipping + (uint32_t)to_left);
    d.right_clipping = u16(d.right_clipping + (uint32_t)to_right);
    d.ref_length = u16(d.ref_length - (uint32_t)(to_left + to_right));
    d.ref_seq_pos += to_left;
    d.err_edit_distance = u16(ed);
    d.score = score;
    d.no_clippings = (uint32_t)((to_left != 0) + (to_right != 0));
    d.cigar_len = s.len;
    return 0;
}

// ---- row L: clipLeft / clipRight of CSoftClipping::clipOverlappingPairs (soft_clipping.cpp:1047-1151) -------------------
// `count` reference-consuming symbols go from the left (right) end, and the insertions that follow (precede) them.
PC_FN uint32_t clip_left(Str& s, wamd_clip_record& d, int count)
{
    const uint32_t start = d.left_clipping;
    if (start > s.len)
        return PC_BAD_STRING;
    uint32_t pos = start, dots = 0, nx = 0, ni = 0, nd = 0;
    int clipped = 0;
    for (;;) {
        if (clipped < count) {
            const uint32_t e = next_event(s, pos);
            if (e > pos) {
                const uint32_t take = umin(e - pos, (uint32_t)(count - clipped));
                dots += take;
                clipped += (int)take;
                pos += take;
                continue;
            }
            if (e >= s.len)
                return PC_BAD_STRING;     // the record claims more reference than the string spans
            const uint32_t c = byte_at(s, e);
            if (c == '#') {
                ++ni;
                pos = e + 2;
            } else if (c == '^') {
                ++nd;
                ++clipped;
                pos = e + 2;
            } else if (c == '$') {
                return PC_BAD_STRING;
            } else {
                ++nx;
                ++clipped;
                pos = e + 1;
            }
        } else if (clipped == count && pos < s.len && byte_at(s, pos) == '#') {
            ++ni;
            pos += 2;
        } else {
            break;
        }
    }
    const uint32_t dollars = dots + nx + ni;
    d.left_clipping = u16(d.left_clipping + dollars);
    d.ref_seq_pos += (int32_t)(dots + nx + nd);
    d.ref_length = u16(d.ref_length - (dots + nx + nd));
    d.err_edit_distance = u16(d.err_edit_distance - (nx + ni + nd));
    fill_bytes(s, start, dollars, '$', true);
    if (start + dollars != pos) {
        const uint32_t tail = s.len - pos;
        move_bytes(s, start + dollars, pos, tail);
        set_len(s, start + dollars + tail);
    }
    return 0;
}

PC_FN uint32_t clip_right(Str& s, wamd_clip_record& d, int count)
{
    const int in0 = (int)s.len - (int)d.right_clipping - 1;
    int in = in0, clipped = 0;
    uint32_t dots = 0, nx = 0, ni = 0, nd = 0;
    for (;;) {
        if (clipped < count) {
            if (in < 0)
                return PC_BAD_STRING;
            const int p = prev_event(s, in);
            const uint32_t cp = p >= 0 ? byte_at(s, (uint32_t)p) : 0u;
            const int p_end = p < 0 ? -1 : ((cp == '#' || cp == '^') ? p + 1 : p);     // last byte of that symbol
            if (in > p_end) {     // dots between it and here
                const uint32_t take = umin((uint32_t)(in - p_end), (uint32_t)(count - clipped));
                dots += take;
                clipped += (int)take;
                in -= (int)take;
                continue;
            }
            if (cp == '#') {
                ++ni;
            } else if (cp == '^') {
                ++nd;
                ++clipped;
            } else if (cp == '$') {
                return PC_BAD_STRING;
            } else {
                ++nx;
                ++clipped;
            }
            in = p - 1;
        } else if (clipped == count && in >= 1 && byte_at(s, (uint32_t)(in - 1)) == '#') {
            ++ni;
            in -= 2;
        } else {
            break;
        }
    }
    const uint32_t dollars = dots + nx + ni, at = (uint32_t)(in + 1), old_tail = (uint32_t)(in0 + 1);
    d.right_clipping = u16(d.right_clipping + dollars);
    d.ref_length = u16(d.ref_length - (dots + nx + nd));
    d.err_edit_distance = u16(d.err_edit_distance - (nx + ni + nd));
    fill_bytes(s, at, dollars, '$', true);
    if (at + dollars != old_tail) {
        const uint32_t tail = s.len - old_tail;
        move_bytes(s, at + dollars, old_tail, tail);
        set_len(s, at + dollars + tail);
    }
    return 0;
}

// One mapping of a pair: `me` is this record (index 0 = p.first), `mate` the other as it stood when the routine was
// entered.  Both lanes of a pair take the same decision from the same two records; the left mapping's lane runs
// clipLeft on its string, the right one's clipRight.
PC_FN uint32_t clip_pair_member(Str& s, wamd_clip_record& me, const wamd_clip_record& mate, const wamd_clip_pair_task& t,
                                uint32_t my_index)
{
    const wamd_clip_record& r0 = my_index ? mate : me;
    const wamd_clip_record& r1 = my_index ? me : mate;
    me.no_clippings = 0;
    me.cigar_len = s.len;
    if (t.ref_seq_id[0] != t.ref_seq_id[1] || r0.left_clipping || r0.right_clipping || r1.left_clipping ||
        r1.right_clipping)
        return 0;
    const bool left_is_first = r0.ref_seq_pos < r1.ref_seq_pos || (r0.ref_seq_pos == r1.ref_seq_pos && t.direct[0]);
    const wamd_clip_record& l = left_is_first ? r0 : r1;
    const wamd_clip_record& r = left_is_first ? r1 : r0;
    const bool left_direct = left_is_first ? t.direct[0] : t.direct[1];
    if (left_direct || !(l.ref_seq_pos + (int32_t)l.ref_length > r.ref_seq_pos))
        return 0;     // only a reverse read that lies left of and overlaps its forward mate
    const int to_left = r.ref_seq_pos - l.ref_seq_pos;
    const int to_right = (r.ref_seq_pos + (int32_t)r.ref_length) - (l.ref_seq_pos + (int32_t)l.ref_length);
    uint32_t rc = 0;
    if (t.perfect[my_index]) {
        const uint32_t rl = t.read_length[my_index];
        if (rl + 1 > s.cap)
            return PC_BAD_STRING;
        fill_bytes(s, 0, rl, '.', false);
        set_len(s, rl);
    }
    const bool i_am_left = (my_index == 0) == left_is_first;
    rc = i_am_left ? clip_left(s, me, to_left) : clip_right(s, me, to_right);
    me.no_clippings = 1;
    me.cigar_len = s.len;
    return rc;
}

// ---- row X: extract_cigar_and_mdz / extract_cigarbin_and_mdz (sam.cpp:805-1051) -------------------------------------------
// CIGAR operations are the runs of symbol classes ('$' S, '.' and letters M, '#' I, '^' D); MD:Z counts matches
// across insertions and clippings and writes the count - also a 0 - in front of every mismatch and of EVERY deleted
// base.  Text leaves through a four-byte accumulator.
struct Text {
    uint8_t* p;
    uint32_t n, cap, acc;
    bool wide;
    PC_MEMBER void put(uint32_t b)
    {
        if (wide) {
            acc |= b << (8u * (n & 3u));
            if ((n & 3u) == 3u) {
                *reinterpret_cast<uint32_t*>(p + (n - 3u)) = acc;
                acc = 0;
            }
        } else {
            p[n] = (uint8_t)b;
        }
        ++n;
    }
    PC_MEMBER void digits(uint32_t v, uint32_t at_least)     // up to eight of them, most significant first
    {
        uint64_t buf = 0;
        uint32_t nd = 0;
        do {
            buf = (buf << 8) | ('0' + v % 10u);
            v /= 10u;
            ++nd;
        } while (v || nd < at_least);
        for (; nd; --nd, buf >>= 8)
            put((uint32_t)buf & 0xffu);
    }
    PC_MEMBER void number(uint32_t v)
    {
        if (v >= 100000000u) {
            digits(v / 100000000u, 1);
            digits(v % 100000000u, 8);
        } else {
            digits(v, 1);
        }
    }
    PC_MEMBER void finish()     // NUL, and what the accumulator still holds
    {
        if (!cap)
            return;
        const uint32_t len = n;
        put(0);
        if (wide && (n & 3u))
            *reinterpret_cast<uint32_t*>(p + (n & ~3u)) = acc;
        n = len;
    }
};

PC_FN Text make_text(uint8_t* p, uint32_t cap)
{
    Text t;
    t.p = p;
    t.n = 0;
    t.cap = cap;
    t.acc = 0;
    t.wide = (((uintptr_t)p | cap) & 3u) == 0;
    return t;
}

PC_FN uint32_t sam_cigar(const Str& s, uint8_t* cigar, uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride, uint32_t* bin,
                         uint32_t bin_stride, wamd_sam_cigar_len& out)
{
    enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_S = 4, OP_NONE = 9 };
    Text cg = make_text(cigar, cigar_stride), md = make_text(mdz, mdz_stride);
    uint32_t nb = 0, op = OP_NONE, count = 0, run_md = 0, pos = 0;
    bool overflow = false;
    auto flush = [&]() {
        if (op == OP_NONE)
            return;
        cg.number(count);
        cg.put((0x534e44494dull >> (8 * op)) & 0xffu);     // "MIDNS"[op]
        if (bin)
            bin[nb++] = count << 4 | op;
    };
    // room for the longest single step - a count and an operation, a count and two symbols - asked for where the
    // reference's byte walk would ask with these cursor values: at the first and at the second symbol of a run
    auto room = [&]() { return !(cg.n + 12 > cigar_stride || md.n + 14 > mdz_stride || (bin && nb + 1 > bin_stride)); };
    for (;;) {
        if (!room()) {
            overflow = true;
            break;
        }
        const uint32_t e = next_event(s, pos);
        if (e > pos) {     // a run of matches
            if (op != OP_M) {
                flush();
                op = OP_M;
                count = 0;
            }
            count += e - pos;
            run_md += e - pos;
            if (!room()) {
                overflow = true;
                break;
            }
        }
        if (e >= s.len)
            break;
        const uint32_t c = byte_at(s, e);
        uint32_t now, step = 1;
        if (c == '$') {
            now = OP_S;
        } else if (c == '#') {
            now = OP_I;
            step = 2;
        } else {
            md.number(run_md);
            run_md = 0;
            md.put(c);
            if (c == '^') {
                md.put(byte_at(s, e + 1));
                now = OP_D;
                step = 2;
            } else {
                now = OP_M;
            }
        }
        if (now != op) {
            flush();
            op = now;
            count = 0;
        }
        ++count;
        pos = e + step;
    }
    if (!overflow) {
        flush();
        if (run_md > 0)
            md.number(run_md);
    } else {
        cg = make_text(cigar, cigar_stride);
        md = make_text(mdz, mdz_stride);
        nb = 0;
    }
    cg.finish();
    md.finish();
    out.cigar_len = cg.n;
    out.mdz_len = md.n;
    out.cigar_bin_len = nb;
    out.reserved = overflow ? 1u : 0u;
    return overflow ? PC_NO_ROOM : 0u;
}

} // namespace post
} // namespace wamd
