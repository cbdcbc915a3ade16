/*
 * wv_oracle.c - CPU restatement of Whisper 2.0.2's read-verification path.
 *
 * TEST INFRASTRUCTURE ONLY (see wv_oracle.h).  Plain C99, no SIMD: the three
 * LevMyers variants are restated UNSKEWED (one text column at a time, words
 * low to high with in-column carries), which computes the same bit-vectors as
 * the reference's anti-diagonal SIMD pipelines; only the iteration index used
 * by the early exit and the sentinel iterations are carried over explicitly.
 * Citations are file:line under /root/reference/src.
 */
#include "wv_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ symbols */

/* GET_CHAR_FROM_PATTERN / GET_CHAR_FROM_GENOME, mapper/LevMyers64.cpp:22-26:
 * even positions sit in the high nibble. */
static inline uint32_t nib(const uint8_t* p, uint64_t i)
{
    uint8_t b = p[i >> 1];
    return (i & 1) ? (uint32_t)(b & 0x0f) : (uint32_t)(b >> 4);
}

/* rev_comp_code, LevMyers64.cpp:40-48: ACGT complemented, 4..7 fixed, 8..15 -> 7. */
static inline uint32_t rc_code(uint32_t c)
{
    if (c < 4)
        return 3 - c;
    return c < 8 ? c : 7;
}

/* raw_code / raw_rev_comp_code, LevMyers64.cpp:50-66. */
static inline uint32_t ascii_code(uint8_t ch, int rev_comp)
{
    switch (ch) {
    case 'A': return rev_comp ? 3u : 0u;
    case 'C': return rev_comp ? 2u : 1u;
    case 'G': return rev_comp ? 1u : 2u;
    case 'T': return rev_comp ? 0u : 3u;
    case 'N': return 4u;
    default:  return 6u;
    }
}

void wo_pattern_init(wo_pattern* p)
{
    memset(p, 0, sizeof(*p)); /* last_symbol(0): LevMyers64.cpp:35 */
}

void wo_preprocess(wo_pattern* p, const uint8_t* seq, uint32_t seq_len, int rev_comp)
{
    memset(p->pm, 0, sizeof(p->pm));
    p->seq_len = seq_len;
    for (uint32_t i = 0; i < seq_len; ++i) {
        uint32_t c = rev_comp ? rc_code(nib(seq, seq_len - i - 1)) : nib(seq, i);
        p->pm[c][i / 64] |= 1ull << (i % 64);
    }
    /* LevMyers64.cpp:182 / 192 */
    p->last_symbol = (uint8_t)(rev_comp ? rc_code(nib(seq, 0)) : nib(seq, seq_len - 1));
}

void wo_preprocess_raw(wo_pattern* p, const uint8_t* seq, uint32_t seq_len, int rev_comp)
{
    memset(p->pm, 0, sizeof(p->pm));
    p->seq_len = seq_len;
    for (uint32_t i = 0; i < seq_len; ++i) {
        uint8_t ch = rev_comp ? seq[seq_len - i - 1] : seq[i];
        uint32_t c = ascii_code(ch & 0x7f, rev_comp); /* tables have 128 entries */
        p->pm[c][i / 64] |= 1ull << (i % 64);
    }
    /* last_symbol keeps its previous value: LevMyers64.cpp:199-220 never writes it */
}

int wo_variant_for_len(uint32_t seq_len)
{
    if (seq_len < 128)
        return WAMD_VARIANT_M128;
    return seq_len < 256 ? WAMD_VARIANT_M256 : WAMD_VARIANT_M64;
}

/* ------------------------------------------------------- dynamicProgramming */

int wo_myers_dp(const wo_pattern* p, int variant, const uint8_t* genome, uint32_t ref_pos,
                uint32_t W, uint32_t k, uint32_t max_text_len, uint32_t* pos, uint32_t* ed,
                wo_cell* matrix)
{
    const uint32_t m = p->seq_len;
    if (variant == WAMD_VARIANT_AUTO)
        variant = wo_variant_for_len(m);

    /* LevMyers64.cpp:244-252, LevMyers128.hpp:97-105, LevMyers256.hpp:102-110 */
    if (W == 0 || W > max_text_len)
        return -1;

    const uint32_t word_num = (m - 1) / 64;   /* word that holds the score bit */
    const uint32_t words = word_num + 1;      /* local_bp_n_words, LevMyers64.cpp:235 */
    const uint64_t score_mask = 1ull << ((m - 1) % 64);
    const int skewed = (variant != WAMD_VARIANT_M64);

    uint64_t VP[WO_MAX_WORDS], VN[WO_MAX_WORDS];
    for (uint32_t w = 0; w < words; ++w) {
        VP[w] = ~0ull;
        VN[w] = 0;
    }
    if (matrix) {
        /* column 0: LevMyers256.hpp:65-73 (D0 ones, HN/HP zero) */
        for (uint32_t w = 0; w < words; ++w) {
            matrix[w].D0 = ~0ull;
            matrix[w].VP = ~0ull;
            matrix[w].VN = matrix[w].HN = matrix[w].HP = 0;
        }
    }

    uint32_t min_ed = m, min_ed_pos = 0, curr_ed = m;
    int stopped = 0;

    /* Sentinel iterations of the skewed variants: for j = 1..word_num the score
     * lane still sees "no symbol" (7), its vectors stay initial and only the
     * bookkeeping below runs (LevMyers256.hpp:120-122, 206-234;
     * LevMyers128.hpp:115, 199-221). */
    if (skewed) {
        for (uint32_t j = 1; j <= word_num && j <= W + 1; ++j) {
            if (variant == WAMD_VARIANT_M256) {
                /* curr_ed == min_ed, D0 bit is 0 -> position moves (wraps as unsigned) */
                min_ed_pos = j - word_num;
            } else if (min_ed_pos + 1 == j - word_num) {
                min_ed_pos = j - word_num;
            }
            if (curr_ed > W + 1 - j + k) {
                stopped = 1;
                break;
            }
        }
    }

    /* columns the score lane really sees: j = col + word_num <= W + 1 (skewed),
     * j = col <= W (LevMyers64.cpp:257) */
    const uint32_t n_cols = skewed ? (W + 1 >= word_num ? W + 1 - word_num : 0) : W;

    for (uint32_t col = 1; col <= n_cols && !stopped; ++col) {
        const uint32_t c = nib(genome, (uint64_t)ref_pos + col - 1);
        uint64_t hn_in = 0, hp_in = 0; /* carries into word 0 are zero: free text start */
        uint64_t D0 = 0, HP = 0, HN = 0;
        for (uint32_t w = 0; w < words; ++w) {
            /* LevMyers64.cpp:312-325 */
            uint64_t X = p->pm[c][w] | hn_in;
            D0 = (((X & VP[w]) + VP[w]) ^ VP[w]) | X | VN[w];
            HP = VN[w] | ~(D0 | VP[w]);
            HN = D0 & VP[w];
            X = (HP << 1) | hp_in;
            VP[w] = (HN << 1) | ~(D0 | X) | hn_in;
            VN[w] = D0 & X;
            hn_in = HN >> 63;
            hp_in = HP >> 63;
            if (matrix) {
                wo_cell* cell = &matrix[(uint64_t)col * words + w];
                cell->D0 = D0;
                cell->HP = HP;
                cell->HN = HN;
                cell->VP = VP[w];
                cell->VN = VN[w];
            }
        }
        /* D0/HP/HN now belong to the score word */
        const int hn_bit = (HN & score_mask) != 0;
        const int hp_bit = (HP & score_mask) != 0;
        int tie = 0;
        if (hn_bit) {
            if (--curr_ed < min_ed) {
                min_ed = curr_ed;
                min_ed_pos = col;
            } else if (curr_ed == min_ed) {
                tie = 1;
            }
        } else if (hp_bit) {
            ++curr_ed;
        } else if (curr_ed == min_ed) {
            tie = 1;
        }
        if (tie) {
            if (variant == WAMD_VARIANT_M256) {
                /* LevMyers256.hpp:191-196, 213-218 */
                const int d0_bit = (D0 & score_mask) != 0;
                if (!d0_bit || c == p->last_symbol)
                    min_ed_pos = col;
            } else if (min_ed_pos + 1 == col) {
                /* LevMyers128.hpp:195-196, 206-207; LevMyers64.cpp:340-341, 355-356 */
                min_ed_pos = col;
            }
        }
        if (skewed && !hn_bit) {
            /* LevMyers256.hpp:232-233, LevMyers128.hpp:125, 219-220 */
            const uint32_t j = col + word_num;
            if (curr_ed > W + 1 - j + k)
                stopped = 1;
        }
    }

    *ed = min_ed;
    if (min_ed > k) {
        *pos = 0;
        return 0;
    }
    *pos = min_ed_pos;
    return 1;
}

/* --------------------------------------------------------------- getExtCigar */

uint32_t wo_ext_cigar(const wo_cell* matrix, uint32_t words, uint8_t* ext_cigar,
                      const uint8_t* ref_codes, const uint8_t* read_codes, uint32_t pos,
                      uint32_t edit_dist, uint32_t seq_len, const wamd_scoring* sc,
                      double* affine_score, uint32_t* num_events)
{
    static const char decode[] = "ACGTNNXX";
    enum { DIR_H, DIR_V, DIR_D };
    uint32_t read_pos = seq_len, n = 0, c_val = edit_dist, events = 0;
    int prev_dir = DIR_D, in_indel = 0;
    double score = 0;

    ext_cigar[0] = 0;
    while (read_pos && pos) {
        const uint32_t w = (read_pos - 1) / 64;
        const uint64_t bit = 1ull << ((read_pos - 1) % 64);
        const wo_cell* cell = &matrix[(uint64_t)pos * words + w];
        uint32_t d_val = c_val, h_val = c_val, v_val = c_val;
        if (!(cell->D0 & bit)) --d_val;
        if (cell->HP & bit) --h_val;
        if (cell->HN & bit) ++h_val;
        if (cell->VP & bit) --v_val;
        if (cell->VN & bit) ++v_val;

        const uint32_t mismatch = read_codes[read_pos] != ref_codes[pos - 1];
        const int from_h = (h_val + 1 == c_val);
        const int from_v = (v_val + 1 == c_val);
        const int from_d = (d_val + mismatch == c_val);
        int dir;
        /* direction choice with stickiness, LevMyers64.cpp:431-442 */
        if (from_d && (prev_dir == DIR_D || (prev_dir == DIR_H && !from_h) ||
                       (prev_dir == DIR_V && !from_v)))
            dir = DIR_D;
        else if (from_h && (prev_dir == DIR_H || (prev_dir == DIR_D && !from_d) ||
                            (prev_dir == DIR_V && !from_v)))
            dir = DIR_H;
        else
            dir = DIR_V;

        if (dir == DIR_D) {
            if (!mismatch) {
                score += sc->match;
                ext_cigar[n++] = '.';
            } else {
                score += sc->mismatch;
                ext_cigar[n++] = (uint8_t)decode[ref_codes[pos - 1]];
            }
            in_indel = 0;
            --read_pos;
            --pos;
            c_val = d_val;
        } else if (dir == DIR_H) { /* deletion */
            if (in_indel) {
                score += sc->gap_del_extend;
            } else {
                score += sc->gap_del_open;
                ++events;
                in_indel = 1;
            }
            ext_cigar[n++] = (uint8_t)decode[ref_codes[pos - 1]];
            ext_cigar[n++] = '^';
            --pos;
            c_val = h_val;
        } else { /* insertion */
            if (in_indel) {
                score += sc->gap_ins_extend;
            } else {
                score += sc->gap_ins_open;
                ++events;
                in_indel = 1;
            }
            ext_cigar[n++] = (uint8_t)decode[read_codes[read_pos]];
            ext_cigar[n++] = '#';
            --read_pos;
            c_val = v_val;
        }
        prev_dir = dir;
    }
    while (read_pos) { /* LevMyers64.cpp:498-513: rest of the read is inserted */
        if (in_indel) {
            score += sc->gap_ins_extend;
        } else {
            score += sc->gap_ins_open;
            ++events;
            in_indel = 1;
        }
        ext_cigar[n++] = (uint8_t)decode[read_codes[read_pos]];
        ext_cigar[n++] = '#';
        --read_pos;
    }
    ext_cigar[n] = 0;
    for (uint32_t a = 0, b = n; a + 1 < b; ++a) { /* reverse, LevMyers64.cpp:517 */
        --b;
        uint8_t t = ext_cigar[a];
        ext_cigar[a] = ext_cigar[b];
        ext_cigar[b] = t;
    }
    *affine_score = score;
    *num_events = events;
    return pos;
}

/* ---------------------------------------------------------- refine_ext_cigar */

int wo_refine_ext_cigar(uint8_t* ec, const uint8_t* ref_codes, uint32_t* edit_distance,
                        uint32_t* num_events, double* score, const wamd_scoring* sc)
{
    static const char decode[] = "ACGTNNXX";
    int ref_pos = 0, i = 0;

    /* first ". ^x . . #y ." or ". #y . . ^x .", sam_matching.cpp:865-902 */
    while (ec[i]) {
        const uint8_t ch = ec[i];
        if (ch == '^' || ch == '#') {
            const uint8_t other = (ch == '^') ? '#' : '^';
            if (i > 0 && ec[i - 1] == '.' && ec[i + 2] == '.' && ec[i + 3] == '.' &&
                ec[i + 4] == other && ec[i + 6] && ec[i + 6] == '.')
                break;
            i += 2;
            if (ch == '^')
                ++ref_pos;
        } else {
            ++i;
            ++ref_pos;
        }
    }
    if (!ec[i])
        return 0;

    /* the reference tests ec[i+2], [i+3], [i+4] for NUL before comparing; a NUL
     * there fails the comparison as well, so the chain above is equivalent.
     * ec[i+5] can only be NUL if ec[i+4] was the last byte, in which case
     * ec[i+6] is read past the terminator by the reference too; the ext-CIGAR
     * buffers are pool slots of 5 * max_read_len bytes (mapper.cpp:697). */
    ec[i++] = (uint8_t)decode[ref_codes[ref_pos++]];
    ec[i++] = (uint8_t)decode[ref_codes[ref_pos++]];
    ec[i++] = (uint8_t)decode[ref_codes[ref_pos++]];
    for (; ec[i + 3]; ++i)
        ec[i] = ec[i + 3];
    ec[i] = 0;

    ++*num_events;
    ++*edit_distance;
    *score += 3 * sc->mismatch;
    *score -= sc->gap_ins_open + sc->gap_del_open;
    return 1;
}

/* ------------------------------------------------ sequence unpacking helpers */

void wo_read_codes(uint8_t* dest, const uint8_t* ascii, uint32_t len, int rev_comp)
{
    /* copy_direct / convert_to_rev_comp, sam.cpp:2296-2347: anything but ACGT -> N_read */
    for (uint32_t i = 0; i < len; ++i) {
        uint8_t ch = rev_comp ? ascii[len - 1 - i] : ascii[i];
        uint32_t c = ascii_code(ch < 128 ? ch : 0, rev_comp);
        dest[i] = (uint8_t)(c < 4 ? c : 4);
    }
}

void wo_ref_codes(uint8_t* dest, const uint8_t* genome, uint32_t pos, uint32_t len)
{
    /* ref_copy_direct, sam.cpp:2350-2375: codes above N_ref collapse to N_ref */
    for (uint32_t i = 0; i < len; ++i) {
        uint32_t c = nib(genome, (uint64_t)pos + i);
        dest[i] = (uint8_t)(c > 5 ? 5 : c);
    }
}

/* ------------------------------------------------------------------- LevDiag */

typedef struct band_side {
    const uint8_t* genome;
    const uint8_t* pattern;
    uint32_t g_base;   /* genome index of band coordinate 0 */
    int g_step;        /* +1 (suffix) or -1 (prefix)         */
    uint32_t p_base;   /* pattern index of row 0             */
    int p_step;
} band_side;

static inline uint32_t band_g(const band_side* s, int j)
{
    return nib(s->genome, (uint64_t)((int64_t)s->g_base + (int64_t)s->g_step * j));
}

static inline uint32_t band_p(const band_side* s, int i)
{
    return nib(s->pattern, (uint64_t)((int64_t)s->p_base + (int64_t)s->p_step * i));
}

/* One side of LevDiag: rows = pattern symbols walking away from the anchor,
 * band coordinates j = genome symbols walking the same way, budget kk.
 * Returns the side's error count, or -1 where the reference returns
 * max_mismatches + 1.  edit_dist.cpp:156-244 (prefix) and 254-341 (suffix)
 * are this routine with mirrored accessors. */
static int levdiag_side(const band_side* s, int rows, int kk, uint32_t* prev, uint32_t* curr)
{
    int left = 1, right = 2 * kk + 1;

    for (int t = 0; t <= kk; ++t)
        prev[kk + t] = prev[kk - t] = (uint32_t)t;

    for (int i = 0; i < rows; ++i) {
        if (left == right) {
            /* one diagonal left: the remainder must match exactly (175-189, 275-289) */
            if ((int32_t)prev[left - 1] > kk)
                return -1;
            for (; i < rows; ++i) {
                if (band_p(s, i) != band_g(s, left))
                    return -1;
                ++left;
            }
            --left;
            curr[left] = (uint32_t)kk;
            right = left;
        } else {
            const uint32_t pc = band_p(s, i);
            uint32_t low = prev[left - 1];
            for (int j = left; j < right; ++j) {
                if (pc == band_g(s, j)) {
                    curr[j] = prev[j - 1];
                    low = curr[j] < prev[j] ? curr[j] : prev[j];
                } else if (prev[j] > low) {
                    curr[j] = ++low;
                } else {
                    low = prev[j];
                    curr[j] = low + 1;
                }
            }
            /* right edge has no upper neighbour inside the band (215-219) */
            if (pc == band_g(s, right))
                curr[right] = prev[right - 1];
            else
                curr[right] = low + 1;

            if ((int32_t)curr[left] > kk)
                if (++left > right)
                    return -1;
            if ((int32_t)curr[right] > kk)
                if (left > --right)
                    return -1;
        }
        uint32_t* t = curr;
        curr = prev;
        prev = t;
        ++left;
        ++right;
    }
    if (left > right)
        return -1;

    int best = kk + 1;
    for (int j = left - 1; j < right; ++j)
        if ((int32_t)prev[j] < best)
            best = (int32_t)prev[j];
    return best;
}

uint32_t wo_levdiag(const uint8_t* genome, uint32_t ref_size, uint32_t cur_offset,
                    const uint8_t* pattern, uint32_t pattern_len, uint32_t position_in_text,
                    uint32_t fixed_left, uint32_t fixed_right, uint32_t k, int dir_gen_flag)
{
    /* window start, edit_dist.cpp:112-115 (unsigned arithmetic as in the reference) */
    uint32_t g0;
    if (dir_gen_flag)
        g0 = position_in_text - cur_offset - k;
    else
        g0 = ref_size - position_in_text - pattern_len + cur_offset - k;

    for (uint32_t i = fixed_left; i < fixed_right; ++i) /* anchor, 121-124 */
        if (nib(genome, (uint64_t)g0 + i + k) != nib(pattern, i))
            return k + 1;

    const size_t cap = (size_t)pattern_len + 2u * k + 3;
    uint32_t* prev = (uint32_t*)malloc(cap * sizeof(uint32_t));
    uint32_t* curr = (uint32_t*)malloc(cap * sizeof(uint32_t));
    int errors_prefix = 0, errors_suffix = 0;
    uint32_t result = k + 1;

    while (fixed_left && nib(genome, (uint64_t)g0 + fixed_left + k - 1) == nib(pattern, fixed_left - 1))
        --fixed_left; /* 150-154 */

    if (fixed_left > 0) {
        band_side s;
        s.genome = genome;
        s.pattern = pattern;
        s.g_base = g0 + fixed_left + 2 * k; /* genome_offset, 158-160 */
        s.g_step = -1;
        s.p_base = fixed_left - 1;
        s.p_step = -1;
        errors_prefix = levdiag_side(&s, (int)fixed_left, (int)k, prev, curr);
        if (errors_prefix < 0 || errors_prefix > (int)k)
            goto done;
    }

    while (fixed_right < pattern_len &&
           nib(genome, (uint64_t)g0 + fixed_right + k) == nib(pattern, fixed_right))
        ++fixed_right; /* 248-252 */

    if (fixed_right < pattern_len) {
        const int kk = (int)k - errors_prefix; /* 256 */
        band_side s;
        s.genome = genome;
        s.pattern = pattern;
        s.g_base = g0 + k + fixed_right - (uint32_t)kk - 1; /* 266 */
        s.g_step = 1;
        s.p_base = fixed_right;
        s.p_step = 1;
        errors_suffix = levdiag_side(&s, (int)(pattern_len - fixed_right), kk, prev, curr);
        if (errors_suffix < 0)
            goto done;
    }
    result = (uint32_t)(errors_prefix + errors_suffix);
done:
    free(prev);
    free(curr);
    return result;
}

/* -------------------------------------------------------------- batch drivers */

static void load_pattern(wo_pattern* p, const uint8_t* reads, const uint64_t* offsets,
                         const uint32_t* lens, int encoding, uint32_t idx, int rev_comp)
{
    if (encoding == WAMD_READS_NIBBLE)
        wo_preprocess(p, reads + offsets[idx], lens[idx], rev_comp);
    else
        wo_preprocess_raw(p, reads + offsets[idx], lens[idx], rev_comp);
}

int wo_distance_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                      const uint32_t* lens, int encoding, const wamd_dist_task* tasks,
                      uint32_t n, int last_symbol, uint32_t max_text_len,
                      wamd_dist_result* results)
{
    wo_pattern* p = (wo_pattern*)malloc(sizeof(wo_pattern));
    int rc = 0;
    wo_pattern_init(p);
    for (uint32_t i = 0; i < n; ++i) {
        const wamd_dist_task* t = &tasks[i];
        if (lens[t->read_idx] == 0 || lens[t->read_idx] > 64 * WO_MAX_WORDS) {
            rc = WAMD_ERR_TOO_LONG;
            break;
        }
        load_pattern(p, reads, offsets, lens, encoding, t->read_idx,
                     t->flags & WAMD_TASK_FLAG_REV_COMP);
        if (last_symbol != WAMD_LAST_SYMBOL_FROM_READ)
            p->last_symbol = (uint8_t)last_symbol;
        int ok = wo_myers_dp(p, t->variant, genome, t->ref_pos, t->max_distance_in_ref,
                             t->max_ed, max_text_len, &results[i].pos,
                             &results[i].edit_distance, NULL);
        if (ok < 0) {
            rc = WAMD_ERR_TEXT_LEN;
            break;
        }
    }
    free(p);
    return rc;
}

int wo_levdiag_batch(const uint8_t* genome, uint32_t ref_size, uint32_t cur_offset,
                     const uint8_t* reads, const uint64_t* offsets, const uint32_t* lens,
                     const wamd_levdiag_task* tasks, uint32_t n, uint32_t* results)
{
    for (uint32_t i = 0; i < n; ++i) {
        const wamd_levdiag_task* t = &tasks[i];
        results[i] = wo_levdiag(genome, ref_size, cur_offset, reads + offsets[t->read_idx],
                                lens[t->read_idx], t->position_in_text, t->fixed_left,
                                t->fixed_right, t->max_mismatches, t->dir_gen_flag != 0);
    }
    return 0;
}

int wo_align_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                   const uint32_t* lens, int encoding, const wamd_dist_task* tasks, uint32_t n,
                   int last_symbol, int refine, uint32_t max_text_len, const wamd_scoring* sc,
                   wamd_align_result* results, uint8_t* cigars, uint32_t cigar_stride)
{
    wo_pattern* p = (wo_pattern*)malloc(sizeof(wo_pattern));
    wo_cell* matrix = NULL;
    size_t matrix_cap = 0;
    uint8_t *ref_codes = NULL, *read_codes = NULL;
    size_t ref_cap = 0;
    int rc = 0;

    wo_pattern_init(p);
    read_codes = (uint8_t*)malloc(64 * WO_MAX_WORDS + 2);
    for (uint32_t i = 0; i < n && !rc; ++i) {
        const wamd_dist_task* t = &tasks[i];
        wamd_align_result* r = &results[i];
        uint8_t* ec = cigars + (size_t)i * cigar_stride;
        const uint32_t m = lens[t->read_idx];
        const int rev = t->flags & WAMD_TASK_FLAG_REV_COMP;
        const uint32_t W = t->max_distance_in_ref;
        memset(r, 0, sizeof(*r));
        ec[0] = 0;
        if (m == 0 || m > 64 * WO_MAX_WORDS) {
            rc = WAMD_ERR_TOO_LONG;
            break;
        }
        if (cigar_stride < 2 * m + 2) {
            rc = WAMD_ERR_INVALID_ARG;
            break;
        }
        const uint32_t words = (m - 1) / 64 + 1;
        const size_t need = ((size_t)W + 2) * words;
        if (need > matrix_cap) {
            free(matrix);
            matrix = (wo_cell*)malloc(need * sizeof(wo_cell));
            matrix_cap = need;
        }
        if ((size_t)W + 8 > ref_cap) {
            free(ref_codes);
            ref_cap = (size_t)W + 8;
            ref_codes = (uint8_t*)malloc(ref_cap);
        }
        load_pattern(p, reads, offsets, lens, encoding, t->read_idx, rev);
        if (last_symbol != WAMD_LAST_SYMBOL_FROM_READ)
            p->last_symbol = (uint8_t)last_symbol;
        int ok = wo_myers_dp(p, t->variant, genome, t->ref_pos, W, t->max_ed, max_text_len,
                             &r->end_pos, &r->edit_distance, matrix);
        if (ok < 0) {
            rc = WAMD_ERR_TEXT_LEN;
            break;
        }
        r->ok = (uint32_t)ok;
        if (!ok)
            continue;

        /* fill_cigar_with_lev, sam_matching.cpp:705-735 */
        if (encoding == WAMD_READS_ASCII) {
            wo_read_codes(read_codes + 1, reads + offsets[t->read_idx], m, rev);
        } else {
            const uint8_t* s = reads + offsets[t->read_idx];
            for (uint32_t q = 0; q < m; ++q)
                read_codes[1 + q] = (uint8_t)(rev ? rc_code(nib(s, m - 1 - q)) : nib(s, q));
        }
        read_codes[0] = 0;
        /* The reference unpacks W symbols (ref_copy_direct, sam_matching.cpp:712).  The
         * skewed variants can end in column W + 1 when the read fits one word
         * (word_num = 0), and getExtCigar then reads tmp_ref_sequence[W]: a stale byte
         * of an earlier call.  That value is not a function of the inputs, so it
         * cannot be reproduced; oracle and HIP path use the genome symbol the DP
         * itself compared against in that column.  Reads above 64 symbols never get
         * there (end column <= W + 1 - word_num). */
        wo_ref_codes(ref_codes, genome, t->ref_pos, W + 1);
        uint32_t events = 0, edist = r->edit_distance;
        double score = 0;
        r->begin_pos = wo_ext_cigar(matrix, words, ec, ref_codes, read_codes, r->end_pos,
                                    r->edit_distance, m, sc, &score, &events);
        if (refine) {
            r->refined =
                (uint32_t)wo_refine_ext_cigar(ec, ref_codes + r->begin_pos, &edist, &events,
                                              &score, sc);
        }
        r->affine_score = score;
        r->num_events = events;
        r->cigar_len = (uint32_t)strlen((const char*)ec);
    }
    free(p);
    free(matrix);
    free(ref_codes);
    free(read_codes);
    return rc;
}

/* ======================================================================= */
/* CIndelMatching (mapper/indels.cpp), CSoftClipping::clipBoundaryPositions */
/* (mapper/soft_clipping.cpp), candidate_mapping_t::serialize (results.cpp)  */
/* ======================================================================= */

#include <ctype.h>
#include <float.h>

#define MIN_SIDE 7   /* min_side_size, indels.h:210 */
#define MAX_SIDE_ERRORS 1

/* rev_comp_code of CIndelMatching, indels.cpp:40-49: 8..15 -> 15 */
static inline uint32_t rc_code_indel(uint32_t c)
{
    if (c < 4)
        return 3 - c;
    return c < 8 ? c : 15;
}

void wo_indel_query(uint8_t* query, const uint8_t* seq, uint32_t len, int encoding, int rev_comp)
{
    for (uint32_t i = 0; i < len; ++i) {
        if (encoding == WAMD_READS_NIBBLE)
            query[i] = (uint8_t)(rev_comp ? rc_code_indel(nib(seq, len - i - 1)) : nib(seq, i));
        else
            query[i] = (uint8_t)ascii_code((rev_comp ? seq[len - i - 1] : seq[i]) & 0x7f, rev_comp);
    }
}

static inline uint32_t popcount64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }

/* calc_penalty, results.cpp:327-343 with types.h:242-247 */
static uint32_t mapping_penalty(uint32_t type, uint32_t no_mismatches)
{
    switch (type) {
    case WAMD_MAP_LEV: return no_mismatches;
    case WAMD_MAP_CLIPPING_INDEL:
    case WAMD_MAP_INDEL_CLIPPING: return no_mismatches + 2;
    case WAMD_MAP_CLIPPING_MISMATCHES:
    case WAMD_MAP_MISMATCHES_CLIPPING: return no_mismatches + 1;
    case WAMD_MAP_INDEL1: return no_mismatches + 1;
    case WAMD_MAP_INDEL2: return no_mismatches + 2;
    case WAMD_MAP_CLIPPING_CLIPPING: return no_mismatches + 2;
    default: return 0;
    }
}

/* CBitCharsSingleWord (indels.h:162-206): 4-bit one-hot cells, newest symbol in the
 * lowest cell; a symbol >= 4 is NOT inserted at all, the window does not advance. */
static uint64_t side_signature(const uint8_t* q)
{
    uint64_t d = 0;
    for (int i = 0; i < MIN_SIDE; ++i)
        if (q[i] < 4)
            d += 1ull << (q[i] + 4 * (MIN_SIDE - i - 1));
    return d;
}

/* find_candidates, indels.cpp:472-492.  cand must hold end - begin entries. */
static uint32_t find_candidates(const uint8_t* g, uint64_t qsig, uint32_t begin, uint32_t end,
                                uint32_t* cand)
{
    const uint64_t mask = ~0ull >> (4 * (16 - MIN_SIDE));
    uint64_t d = 0;
    uint32_t n = 0;
    for (int i = 0; i < MIN_SIDE - 1; ++i)
        if (g[begin + i] < 4)
            d += 1ull << (g[begin + i] + 4 * (MIN_SIDE - i - 2));
    for (int i = (int)begin + MIN_SIDE - 1; i < (int)end; ++i) {
        if (g[i] < 4)
            d = ((d << 4) + (1ull << g[i])) & mask;
        if (popcount64(d & qsig) >= MIN_SIDE - MAX_SIDE_ERRORS)
            cand[n++] = (uint32_t)(i - MIN_SIDE + 1);
    }
    return n;
}

/* calc_mismatches_with_indel, indels.cpp:495-563 */
static void mismatches_with_indel(const uint8_t* g, const uint8_t* q, uint32_t query_len, uint32_t g_left,
                                  uint32_t g_right, uint32_t q_left, uint32_t q_right, int left_margin,
                                  uint32_t* left_mis, uint32_t* right_mis, uint32_t* no_mis,
                                  uint32_t* indel_pos)
{
    const uint32_t text_size = g_right - g_left, query_size = q_right - q_left;
    const uint32_t stats = text_size < query_size ? text_size : query_size;
    if (stats <= MIN_SIDE) {
        *no_mis = query_len + 1;
        *indel_pos = 0;
        return;
    }
    const int start = left_margin ? MIN_SIDE : 0;
    const int end = (int)stats - (left_margin ? 0 : MIN_SIDE);
    left_mis[0] = 0;
    for (uint32_t i = 1; i <= stats; ++i)
        left_mis[i] = left_mis[i - 1] + (g[g_left + i - 1] != q[q_left + i - 1]);
    right_mis[stats] = 0;
    for (uint32_t i = 1; i <= stats; ++i)
        right_mis[stats - i] = right_mis[stats - i + 1] +
                               (g[g_left + text_size - i] != q[q_left + query_size - i]);
    int best = (int)query_len + 1, best_pos = 0;
    for (int i = start; i < end; ++i) {
        const int tot = (int)(left_mis[i] + right_mis[i]);
        if (tot < best) {
            best_pos = i;
            best = tot;
        }
    }
    *no_mis = (uint32_t)best;
    *indel_pos = q_left + (uint32_t)best_pos;
}

int wo_indel_match(const uint8_t* genome_packed, const uint8_t* query, uint32_t query_len,
                   uint32_t ref_pos, uint32_t max_indel, uint32_t fpos, uint32_t fsize,
                   uint32_t max_mis, const wamd_indel_params* prm, wamd_mapping* cm)
{
    const double match_score = prm->scoring.match, mismatch_score = prm->scoring.mismatch;
    const double gap_del_open = prm->scoring.gap_del_open, gap_del_extend = prm->scoring.gap_del_extend;
    const double gap_ins_open = prm->scoring.gap_ins_open, gap_ins_extend = prm->scoring.gap_ins_extend;
    const double clipping_score = prm->scoring.clipping;

    /* a default-constructed candidate_mapping_t, types.h:250-261 */
    memset(cm, 0, sizeof(*cm));
    cm->type = WAMD_MAP_NONE;
    cm->no_mismatches = max_mis + 1;

    const uint32_t text_size = query_len + 2 * max_indel;
    int rc = 0;
    uint8_t* g = (uint8_t*)malloc((size_t)text_size + 8);
    uint32_t* cand = (uint32_t*)malloc(((size_t)text_size + 8) * sizeof(uint32_t));
    uint32_t* lm = (uint32_t*)malloc(((size_t)query_len + 8) * sizeof(uint32_t));
    uint32_t* rm = (uint32_t*)malloc(((size_t)query_len + 8) * sizeof(uint32_t));
    /* prefetch_genome(ref_pos - max_indel_size, text_size): raw 4-bit codes */
    for (uint32_t i = 0; i < text_size + 2; ++i)
        g[i] = (uint8_t)nib(genome_packed, (uint64_t)(ref_pos - max_indel) + i);

    double total_cost = 0; /* min_acceptable_score */
    const int min_size_after_clipping = (int)(query_len * prm->min_clipped_factor);

    int left_indel_size = 0, right_indel_size = 0;
    uint32_t no_left_mis_indel = query_len, no_right_mis_indel = query_len;
    double best_left_cost = 0, best_right_cost = 0;
    uint32_t best_left_indel_pos = 0, best_right_indel_pos = 0;
    int left_indel_found = 0, right_indel_found = 0;
    double best_left_clip_cost = 0, best_right_clip_cost = 0;
    uint32_t best_no_left_clipped = 0, best_no_right_clipped = 0;
    uint32_t no_left_mis_clipped = 0, no_right_mis_clipped = 0;
    uint32_t best_no_left_mis_clipped = 0, best_no_right_mis_clipped = 0;

    /* anchor, indels.cpp:167-169 */
    for (uint32_t i = 0; i < fsize; ++i)
        if (query[fpos + i] != g[max_indel + fpos + i])
            goto done;

    /* left part: raw mismatches and best clipping, 173-194 */
    uint32_t no_left_mis_raw = 0;
    double clipping_cost = clipping_score;
    for (int i = (int)fpos - 1; i >= 0; --i) {
        if (query[i] != g[max_indel + i]) {
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
    if (fpos > (uint32_t)MIN_SIDE) { /* 198-231 */
        const uint64_t qsig = side_signature(query);
        const uint32_t nc = find_candidates(g, qsig, 0, max_indel + fpos, cand);
        for (uint32_t c = 0; c < nc; ++c) {
            uint32_t c_mis, indel_pos;
            mismatches_with_indel(g, query, query_len, cand[c], max_indel + fpos + 1, 0, fpos + 1, 1, lm, rm,
                                  &c_mis, &indel_pos);
            if (c_mis > max_mis)
                continue;
            const int c_indel = (int)max_indel - (int)cand[c];
            if (abs(c_indel) > (int)max_indel)
                continue;
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
        }
    }

    /* right part, 235-293 */
    uint32_t no_right_mis_raw = 0;
    clipping_cost = clipping_score;
    for (uint32_t i = fpos + fsize; i < query_len; ++i) {
        if (query[i] != g[max_indel + i]) {
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
    if (fpos + fsize + MIN_SIDE < query_len) {
        const uint64_t qsig = side_signature(query + query_len - MIN_SIDE);
        const uint32_t nc = find_candidates(g, qsig, max_indel + fpos + fsize, text_size, cand);
        for (uint32_t c = 0; c < nc; ++c) {
            uint32_t c_mis, indel_pos;
            mismatches_with_indel(g, query, query_len, fpos + fsize - 1 + max_indel, cand[c] + MIN_SIDE,
                                  fpos + fsize - 1, query_len, 0, lm, rm, &c_mis, &indel_pos);
            if (c_mis > max_mis)
                continue;
            const int c_indel = ((int)cand[c] + MIN_SIDE) - ((int)text_size - (int)max_indel);
            if (abs(c_indel) > (int)max_indel)
                continue;
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
        }
    }

#define RESCALE(size) ((uint32_t)(size) * max_mis / query_len) /* indels.h:252-255 */
#define LEFT_POS() (left_indel_size > 0 ? ref_pos - (uint32_t)left_indel_size : ref_pos + (uint32_t)(-left_indel_size))

    /* the eight combined cases in the reference's order, 297-457 */
    if (left_indel_found && no_left_mis_indel + no_right_mis_raw < max_mis) {
        double c = best_left_cost + (query_len - fpos - no_right_mis_raw) * match_score +
                   no_right_mis_raw * mismatch_score;
        if (c > total_cost) {
            total_cost = c;
            cm->pos = LEFT_POS();
            cm->type = WAMD_MAP_INDEL1;
            cm->matching_len1 = best_left_indel_pos;
            cm->indel_len1 = left_indel_size;
            cm->no_mismatches = no_left_mis_indel + no_right_mis_raw;
        }
    }
    if (right_indel_found && no_left_mis_raw + no_right_mis_indel < max_mis) {
        double c = best_right_cost + (fpos + fsize - no_left_mis_raw) * match_score +
                   no_left_mis_raw * mismatch_score;
        if (c > total_cost) {
            total_cost = c;
            cm->pos = ref_pos;
            cm->type = WAMD_MAP_INDEL1;
            cm->matching_len1 = best_right_indel_pos;
            cm->indel_len1 = right_indel_size;
            cm->no_mismatches = no_left_mis_raw + no_right_mis_indel;
        }
    }
    if (left_indel_found && right_indel_found && no_left_mis_indel + no_right_mis_indel < max_mis) {
        double c = best_left_cost + best_right_cost + fsize * match_score;
        if (c > total_cost) {
            total_cost = c;
            cm->type = WAMD_MAP_INDEL2;
            cm->pos = LEFT_POS();
            cm->matching_len1 = best_left_indel_pos;
            cm->indel_len1 = left_indel_size;
            if (left_indel_size > 0)
                cm->matching_len2 = best_right_indel_pos - best_left_indel_pos;
            else
                cm->matching_len2 = best_right_indel_pos - (best_left_indel_pos + (uint32_t)(-left_indel_size));
            cm->indel_len2 = right_indel_size;
            cm->no_mismatches = no_left_mis_indel + no_right_mis_indel;
        }
    }
    if (best_no_left_clipped > 0 &&
        best_no_left_mis_clipped + no_right_mis_raw < RESCALE(query_len - best_no_left_clipped)) {
        double c = best_left_clip_cost + (query_len - fpos - no_right_mis_raw) * match_score +
                   no_right_mis_raw * mismatch_score;
        if (c > total_cost && query_len - best_no_left_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm->pos = ref_pos + best_no_left_clipped;
            cm->type = WAMD_MAP_CLIPPING_MISMATCHES;
            cm->clipping_len = best_no_left_clipped;
            cm->matching_len1 = query_len - best_no_left_clipped;
            cm->no_mismatches = best_no_left_mis_clipped + no_right_mis_raw;
        }
    }
    if (best_no_right_clipped > 0 &&
        no_left_mis_raw + best_no_right_mis_clipped < RESCALE(query_len - best_no_right_clipped)) {
        double c = best_right_clip_cost + (fpos + fsize - no_left_mis_raw) * match_score +
                   no_left_mis_raw * mismatch_score;
        if (c > total_cost && query_len - best_no_right_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm->pos = ref_pos;
            cm->type = WAMD_MAP_MISMATCHES_CLIPPING;
            cm->clipping_len = best_no_right_clipped;
            cm->matching_len1 = query_len - best_no_right_clipped;
            cm->no_mismatches = no_left_mis_raw + best_no_right_mis_clipped;
        }
    }
    if (best_no_left_clipped > 0 && right_indel_found &&
        best_no_left_mis_clipped + no_right_mis_indel < RESCALE(query_len - best_no_left_clipped)) {
        double c = best_left_clip_cost + best_right_cost + fsize * match_score;
        if (c > total_cost && query_len - best_no_left_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm->pos = ref_pos + best_no_left_clipped;
            cm->type = WAMD_MAP_CLIPPING_INDEL;
            cm->indel_len1 = right_indel_size;
            cm->clipping_len = best_no_left_clipped;
            cm->matching_len1 = best_right_indel_pos - best_no_left_clipped;
            cm->no_mismatches = best_no_left_mis_clipped + no_right_mis_indel;
        }
    }
    if (best_no_right_clipped > 0 && left_indel_found &&
        best_no_right_mis_clipped + no_left_mis_indel < RESCALE(query_len - best_no_right_clipped)) {
        double c = best_right_clip_cost + best_left_cost + fsize * match_score;
        if (c > total_cost && query_len - best_no_right_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm->pos = LEFT_POS();
            cm->type = WAMD_MAP_INDEL_CLIPPING;
            cm->indel_len1 = left_indel_size;
            cm->clipping_len = best_no_right_clipped;
            cm->matching_len1 = best_left_indel_pos;
            cm->no_mismatches = best_no_right_mis_clipped + no_left_mis_indel;
        }
    }
    if (best_no_left_clipped > 0 && best_no_right_clipped > 0 &&
        best_no_left_mis_clipped + best_no_right_mis_clipped <
            RESCALE(query_len - best_no_left_clipped - best_no_right_clipped)) {
        double c = best_left_clip_cost + best_right_clip_cost + fsize * match_score;
        if (c > total_cost &&
            query_len - best_no_left_clipped - best_no_right_clipped >= (uint32_t)min_size_after_clipping) {
            total_cost = c;
            cm->pos = ref_pos + best_no_left_clipped;
            cm->type = WAMD_MAP_CLIPPING_CLIPPING;
            cm->clipping_len = best_no_left_clipped;
            cm->matching_len1 = query_len - best_no_left_clipped - best_no_right_clipped;
            cm->no_mismatches = best_no_left_mis_clipped + best_no_right_mis_clipped;
        }
    }
    rc = cm->type != WAMD_MAP_NONE;
    if (rc)
        cm->penalty = mapping_penalty(cm->type, cm->no_mismatches);
done:
    cm->found = (uint32_t)rc;
    free(g);
    free(cand);
    free(lm);
    free(rm);
    return rc;
}

/* ---- GetExtCigarNew -------------------------------------------------------- */

typedef struct ec_state {
    const uint8_t* g;
    const uint8_t* q;
    uint8_t* ec;
    uint32_t i_genome, i_query;
    uint16_t edist, events;
    double score;
} ec_state;

static const char code2symbol[17] = "ACGTNNNNNNNNNNNN"; /* indels.cpp:69-76 */

static void ec_matching(ec_state* s, uint32_t len, const wamd_scoring* sc)
{
    int no_mis = 0; /* append_matching_part, indels.cpp:680-702 */
    for (uint32_t i = 0; i < len; ++i) {
        if (s->g[s->i_genome] == s->q[s->i_query]) {
            *s->ec++ = '.';
        } else {
            *s->ec++ = (uint8_t)code2symbol[s->g[s->i_genome] & 15];
            ++no_mis;
        }
        ++s->i_genome;
        ++s->i_query;
    }
    s->score += (len - no_mis) * sc->match + no_mis * sc->mismatch;
    s->events = (uint16_t)(s->events + no_mis);
    s->edist = (uint16_t)(s->edist + no_mis);
}

static void ec_indel(ec_state* s, int len, const wamd_scoring* sc)
{
    if (len > 0) { /* append_indel, indels.cpp:705-732 */
        for (uint32_t i = 0; i < (uint32_t)len; ++i) {
            *s->ec++ = '^';
            *s->ec++ = (uint8_t)code2symbol[s->g[s->i_genome++] & 15];
        }
        s->score += sc->gap_del_open + (len - 1) * sc->gap_del_extend;
        s->events = (uint16_t)(s->events + 1);
        s->edist = (uint16_t)(s->edist + len);
    } else {
        for (uint32_t i = 0; i < (uint32_t)-len; ++i) {
            *s->ec++ = '#';
            *s->ec++ = (uint8_t)code2symbol[s->q[s->i_query++] & 15];
        }
        /* the reference multiplies by (indel_len - 1) with indel_len NEGATIVE (728) */
        s->score += sc->gap_ins_open + (len - 1) * sc->gap_ins_extend;
        s->events = (uint16_t)(s->events + 1);
        s->edist = (uint16_t)(s->edist + -len);
    }
}

static void ec_clipping(ec_state* s, int len, const wamd_scoring* sc)
{
    for (uint32_t i = 0; i < (uint32_t)len; ++i) /* append_clipping, indels.cpp:735-745 */
        *s->ec++ = '$';
    s->score += sc->clipping;
    s->events = (uint16_t)(s->events + 1);
    s->i_query += (uint32_t)len;
}

int wo_indel_ext_cigar(const uint8_t* genome_packed, const uint8_t* query, uint32_t query_len,
                       const wamd_indel_cigar_task* t, const wamd_scoring* sc,
                       wamd_indel_cigar_result* r, uint8_t* ext_cigar)
{
    /* fill_cigar_with_indel_matching prefetches read_len + 2 * max_indel + 1 symbols from
     * examined_start_pos; the walk below never needs more than read_len + |indels| */
    const uint32_t span = query_len + (uint32_t)abs(t->indel_len1) + (uint32_t)abs(t->indel_len2) + 8;
    uint8_t* g = (uint8_t*)malloc(span);
    for (uint32_t i = 0; i < span; ++i)
        g[i] = (uint8_t)nib(genome_packed, (uint64_t)t->examined_start_pos + i);
    ec_state s = {g, query, ext_cigar, 0, 0, 0, 0, 0.0};
    const uint32_t read_length = query_len;
    memset(r, 0, sizeof(*r));
    r->ok = 1;
    switch (t->type) {
    case WAMD_MAP_INDEL1: /* 748-773 */
        ec_matching(&s, t->matching_len1, sc);
        ec_indel(&s, t->indel_len1, sc);
        ec_matching(&s, read_length - s.i_query, sc);
        r->method = t->indel_len1 > 0 ? 14 : 13;
        r->ref_length = (uint16_t)((int)read_length + t->indel_len1);
        break;
    case WAMD_MAP_INDEL2: /* 776-811 */
        ec_matching(&s, t->matching_len1, sc);
        ec_indel(&s, t->indel_len1, sc);
        ec_matching(&s, t->matching_len2, sc);
        ec_indel(&s, t->indel_len2, sc);
        ec_matching(&s, read_length - s.i_query, sc);
        if (t->indel_len1 > 0)
            r->method = t->indel_len2 > 0 ? 18 : 17;
        else
            r->method = t->indel_len2 > 0 ? 15 : 16;
        r->ref_length = (uint16_t)((int)read_length + t->indel_len1 + t->indel_len2);
        break;
    case WAMD_MAP_CLIPPING_MISMATCHES:
    case WAMD_MAP_MISMATCHES_CLIPPING: /* 814-848 */
        if (t->type == WAMD_MAP_CLIPPING_MISMATCHES) {
            r->method = 19;
            ec_clipping(&s, (int)(read_length - t->matching_len1), sc);
            r->left_clipping = (uint16_t)(read_length - t->matching_len1);
        }
        ec_matching(&s, t->matching_len1, sc);
        if (t->type == WAMD_MAP_MISMATCHES_CLIPPING) {
            r->method = 20;
            ec_clipping(&s, (int)(read_length - t->matching_len1), sc);
            r->right_clipping = (uint16_t)(read_length - t->matching_len1);
        }
        r->ref_length = (uint16_t)t->matching_len1;
        break;
    case WAMD_MAP_CLIPPING_INDEL:
    case WAMD_MAP_INDEL_CLIPPING: /* 851-906 */
        if (t->type == WAMD_MAP_CLIPPING_INDEL) {
            r->method = t->indel_len1 > 0 ? 23 : 21;
            ec_clipping(&s, (int)t->clipping_len, sc);
            ec_matching(&s, t->matching_len1, sc);
            ec_indel(&s, t->indel_len1, sc);
            ec_matching(&s, query_len - s.i_query, sc);
            r->left_clipping = (uint16_t)t->clipping_len;
        } else {
            r->method = t->indel_len1 > 0 ? 24 : 22;
            ec_matching(&s, t->matching_len1, sc);
            ec_indel(&s, t->indel_len1, sc);
            ec_matching(&s, query_len - s.i_query - t->clipping_len, sc);
            ec_clipping(&s, (int)t->clipping_len, sc);
            r->right_clipping = (uint16_t)t->clipping_len;
        }
        r->ref_length = (uint16_t)((int)read_length - (int)t->clipping_len + t->indel_len1);
        break;
    case WAMD_MAP_CLIPPING_CLIPPING: /* 909-935 */
        r->method = 25;
        r->left_clipping = (uint16_t)t->clipping_len;
        ec_clipping(&s, r->left_clipping, sc);
        ec_matching(&s, t->matching_len1, sc);
        ec_clipping(&s, (int)(read_length - t->matching_len1 - r->left_clipping), sc);
        r->right_clipping = (uint16_t)(read_length - t->matching_len1 - r->left_clipping);
        r->ref_length = (uint16_t)t->matching_len1;
        break;
    default:
        r->ok = 0;
        break;
    }
    *s.ec = 0;
    r->score = s.score;
    r->err_edit_distance = s.edist;
    r->num_events = s.events;
    r->cigar_len = (uint32_t)(s.ec - ext_cigar);
    free(g);
    return (int)r->ok;
}

/* ---- clipBoundaryPositions -------------------------------------------------- */

void wo_clip_boundary(wamd_clip_record* d, uint8_t* ext_cigar, const wamd_scoring* sc)
{
    d->cigar_len = (uint32_t)strlen((const char*)ext_cigar);
    d->no_clippings = 0;
    if (d->err_edit_distance == 0)
        return;
    int mini_pos = -1, maxi_pos = -1;
    double mini = DBL_MAX, maxi = -DBL_MAX, cur = 0;
    const double left_penalty = d->left_clipping > 0 ? 0 : sc->clipping;
    const double right_penalty = d->right_clipping > 0 ? 0 : sc->clipping;
    uint8_t* in = ext_cigar + d->left_clipping;
    uint8_t* out;
    int in_indel = 0, pos;

    for (pos = 0; *in && *in != '$'; ++pos) { /* 849-890 */
        if (*in == '.') {
            cur += sc->match;
            in_indel = 0;
            ++in;
        } else if (*in == '#') {
            cur += in_indel ? sc->gap_ins_extend : sc->gap_ins_open;
            in_indel = 1;
            in += 2;
            ++pos;
        } else if (*in == '^') {
            cur += in_indel ? sc->gap_del_extend : sc->gap_del_open;
            in_indel = 1;
            in += 2;
            ++pos;
        } else if (isalpha(*in)) {
            cur += sc->mismatch;
            in_indel = 0;
            ++in;
        } else {
            ++in; /* the reference would spin here; no such byte occurs in an ext-CIGAR */
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
    if (mini_pos + 20 > maxi_pos)
        return;

    int no_clippings = 0;
    if (maxi + right_penalty > cur) { /* right clipping, 898-941 */
        ++no_clippings;
        in = ext_cigar + d->left_clipping + maxi_pos + 1;
        out = in;
        while (*in && *in != '$') {
            if (*in == '.') {
                *out++ = '$';
                ++in;
                ++d->right_clipping;
                --d->ref_length;
            } else if (*in == '#') {
                *out++ = '$';
                in += 2;
                ++d->right_clipping;
                --d->err_edit_distance;
            } else if (*in == '^') {
                in += 2;
                --d->ref_length;
                --d->err_edit_distance;
            } else if (isalpha(*in)) {
                *out++ = '$';
                ++in;
                ++d->right_clipping;
                --d->ref_length;
                --d->err_edit_distance;
            } else {
                ++in;
            }
        }
        while (*in)
            *out++ = *in++;
        *out = 0;
    }
    if (mini < left_penalty) { /* left clipping, 944-991 */
        ++no_clippings;
        in = ext_cigar + d->left_clipping;
        out = in;
        uint8_t* guard = in + mini_pos + 1;
        while (in != guard) {
            if (*in == '.') {
                *out++ = '$';
                ++in;
                ++d->left_clipping;
                ++d->ref_seq_pos;
                --d->ref_length;
            } else if (*in == '#') {
                *out++ = '$';
                in += 2;
                ++d->left_clipping;
                --d->err_edit_distance;
            } else if (*in == '^') {
                in += 2;
                ++d->ref_seq_pos;
                --d->ref_length;
                --d->err_edit_distance;
            } else if (isalpha(*in)) {
                *out++ = '$';
                ++in;
                ++d->left_clipping;
                ++d->ref_seq_pos;
                --d->ref_length;
                --d->err_edit_distance;
            } else {
                ++in;
            }
        }
        while (*in)
            *out++ = *in++;
        *out = 0;
    }
    /* 993 */
    d->score = (0.0 > maxi ? 0.0 : maxi) - (0.0 < mini ? 0.0 : mini) + no_clippings * sc->clipping;
    d->no_clippings = (uint32_t)no_clippings;
    d->cigar_len = (uint32_t)strlen((const char*)ext_cigar);
}

/* ---- clipIllegalPositions ----------------------------------------------------- */

static uint32_t bounded_len(const uint8_t* ec, uint32_t stride)
{
    uint32_t l = 0;
    while (l < stride && ec[l])
        ++l;
    return l;
}

int wo_clip_illegal(wamd_clip_record* d, const wamd_clip_illegal_task* t, uint8_t* ec, uint32_t stride,
                    const wamd_scoring* sc)
{
    const int32_t ref_size = t->ref_size;                                   /* 594-595 */
    const int32_t mapping_last = d->ref_seq_pos + (int32_t)d->ref_length - 1; /* 596 */
    int to_left = 0, to_right = 0;
    int i_read = 0, i_ref = d->ref_seq_pos;
    const int read_length = t->read_length;
    int bad = 0;
    d->no_clippings = 0;
    if (d->ref_seq_pos <= 0) /* 604-606 */
        to_left = -d->ref_seq_pos + 1;
    if (mapping_last > ref_size) /* 608-610 */
        to_right = mapping_last - ref_size;
    if (!to_left && !to_right) {
        d->cigar_len = bounded_len(ec, stride);
        return 0;
    }
    if ((uint32_t)read_length + 1 > stride)
        return 1;
    if (t->perfect) { /* 614-618 */
        memset(ec, '.', (size_t)read_length);
        ec[read_length] = 0;
    }
    d->left_clipping = (uint16_t)(d->left_clipping + to_left); /* 620-625 */
    d->right_clipping = (uint16_t)(d->right_clipping + to_right);
    d->ref_length = (uint16_t)(d->ref_length - (to_left + to_right));
    d->ref_seq_pos += to_left;
    d->err_edit_distance = 0;
    d->score = 0;
    d->no_clippings = (uint32_t)((to_left != 0) + (to_right != 0));

    uint32_t in = 0, out = 0;
    const uint32_t lim = stride - 1; /* last byte that may hold the NUL */
#define AT(i) ((i) < stride ? ec[i] : 0)
    while (i_ref < 1 && AT(in)) { /* 632-652 */
        if (ec[in] == '^') {
            in += 2;
            ++i_ref;
        } else if (ec[in] == '#') {
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
    int gap_opened = 0;
    while (i_read < read_length && i_ref <= ref_size && AT(in)) { /* 655-703 */
        const uint8_t c = ec[in];
        if (c == '^' || c == '#') {
            if (in + 1 >= lim) {
                bad = 1;
                break;
            }
            ec[out++] = ec[in++];
            ec[out++] = ec[in++];
            d->err_edit_distance++;
            if (c == '^') {
                ++i_ref;
                d->score += gap_opened ? sc->gap_del_extend : sc->gap_del_open;
            } else {
                ++i_read;
                d->score += gap_opened ? sc->gap_ins_extend : sc->gap_ins_open;
            }
            gap_opened = 1;
        } else {
            if (c == '.')
                d->score += sc->match;
            else {
                d->score += sc->mismatch;
                d->err_edit_distance++;
            }
            ec[out++] = ec[in++];
            ++i_read;
            ++i_ref;
            gap_opened = 0;
        }
    }
    while (!bad && i_read < read_length && AT(in)) { /* 706-725 */
        if (ec[in] == '^') {
            in += 2;
            ++i_ref;
        } else if (ec[in] == '#') {
            in += 2;
            ec[out++] = '$';
            ++i_read;
        } else { /* 719-724: `in` stays where it is - the rest of the read is padded with '$' */
            if (out >= lim) {
                bad = 1;
                break;
            }
            ec[out++] = '$';
            ++i_read;
            ++i_ref;
        }
    }
#undef AT
    if (out > lim)
        out = lim;
    ec[out] = 0; /* 727 */
    d->cigar_len = out;
    return bad;
}

/* ---- clipOverlappingPairs, clipLeft, clipRight --------------------------------- */

static int wo_isalpha(uint8_t c) { return (uint8_t)((c | 32) - 'a') < 26; }

static int clip_left(wamd_clip_record* d, uint8_t* ec, uint32_t stride, int count) /* 1047-1098 */
{
    uint32_t in = d->left_clipping, out = in;
    int clipped = 0;
#define AT(i) ((i) < stride ? ec[i] : 0)
    while (clipped < count || (clipped == count && AT(in) == '#')) {
        const uint8_t c = AT(in);
        if (c == '.') {
            ec[out++] = '$';
            ++in;
            ++d->left_clipping;
            ++d->ref_seq_pos;
            --d->ref_length;
            ++clipped;
        } else if (c == '#') {
            ec[out++] = '$';
            in += 2;
            ++d->left_clipping;
            --d->err_edit_distance;
        } else if (c == '^') {
            in += 2;
            ++d->ref_seq_pos;
            --d->ref_length;
            --d->err_edit_distance;
            ++clipped;
        } else if (wo_isalpha(c)) {
            ec[out++] = '$';
            ++in;
            ++d->left_clipping;
            ++d->ref_seq_pos;
            --d->ref_length;
            --d->err_edit_distance;
            ++clipped;
        } else {
            return 1; /* NUL or a foreign byte: the reference spins here */
        }
    }
    if (out != in) { /* 1092-1097 */
        while (AT(in))
            ec[out++] = ec[in++];
        ec[out] = 0;
    }
#undef AT
    return 0;
}

static int clip_right(wamd_clip_record* d, uint8_t* ec, uint32_t stride, int count) /* 1101-1151 */
{
    const int len = (int)bounded_len(ec, stride);
    int in = len - (int)d->right_clipping - 1, out = in + 1;
    int clipped = 0;
#define PREV() (in >= 1 ? ec[in - 1] : 0)
    while (clipped < count || (clipped == count && PREV() == '#')) {
        if (in < 0)
            return 1;
        const uint8_t c = ec[in];
        if (c == '.') {
            ec[--out] = '$';
            --in;
            ++d->right_clipping;
            --d->ref_length;
            ++clipped;
        } else if (wo_isalpha(c)) {
            if (PREV() == '#') {
                ec[--out] = '$';
                in -= 2;
                ++d->right_clipping;
                --d->err_edit_distance;
            } else if (PREV() == '^') {
                in -= 2;
                --d->ref_length;
                --d->err_edit_distance;
                ++clipped;
            } else {
                ec[--out] = '$';
                --in;
                ++d->right_clipping;
                --d->ref_length;
                --d->err_edit_distance;
                ++clipped;
            }
        } else {
            return 1;
        }
    }
#undef PREV
    ++in; /* 1144-1150: the string becomes [0, in) + [out, len); what the reference leaves behind
           * the new NUL is whatever followed the old one and is not reproduced */
    if (in != out) {
        while (out < len)
            ec[in++] = ec[out++];
        ec[in] = 0;
    }
    return 0;
}

int wo_clip_overlapping_pair(wamd_clip_record* rec, const wamd_clip_pair_task* t, uint8_t* ec_first,
                             uint8_t* ec_second, uint32_t stride)
{
    wamd_clip_record *first = &rec[0], *second = &rec[1];
    uint8_t* ec[2] = {ec_first, ec_second};
    int rc = 0;
    first->no_clippings = second->no_clippings = 0;
    first->cigar_len = bounded_len(ec_first, stride);
    second->cigar_len = bounded_len(ec_second, stride);
    if (t->ref_seq_id[0] != t->ref_seq_id[1] || first->left_clipping || first->right_clipping ||
        second->left_clipping || second->right_clipping) /* 999-1003 */
        return 0;
    int left; /* index of the left mapping, 1005-1025 */
    if (first->ref_seq_pos < second->ref_seq_pos)
        left = 0;
    else if (first->ref_seq_pos == second->ref_seq_pos)
        left = t->direct[0] ? 0 : 1;
    else
        left = 1;
    const int right = 1 - left;
    wamd_clip_record *l = &rec[left], *r = &rec[right];
    if (!t->direct[left] && l->ref_seq_pos + (int32_t)l->ref_length > r->ref_seq_pos) { /* 1028 */
        for (int q = 0; q < 2; ++q)
            if (t->perfect[q]) { /* 1029-1036 */
                if ((uint32_t)t->read_length[q] + 1 > stride)
                    return 1;
                memset(ec[q], '.', t->read_length[q]);
                ec[q][t->read_length[q]] = 0;
            }
        const int to_left = r->ref_seq_pos - l->ref_seq_pos; /* 1038-1039 */
        const int to_right = (r->ref_seq_pos + (int32_t)r->ref_length) - (l->ref_seq_pos + (int32_t)l->ref_length);
        rc |= clip_left(l, ec[left], stride, to_left);
        rc |= clip_right(r, ec[right], stride, to_right);
        first->no_clippings = second->no_clippings = 1;
        first->cigar_len = bounded_len(ec_first, stride);
        second->cigar_len = bounded_len(ec_second, stride);
    }
    return rc;
}

int wo_clip_illegal_batch(uint32_t n, const wamd_clip_illegal_task* tasks, wamd_clip_record* records,
                          uint8_t* cigars, uint32_t cigar_stride, const wamd_scoring* sc)
{
    int rc = 0;
    for (uint32_t i = 0; i < n; ++i)
        rc |= wo_clip_illegal(&records[i], &tasks[i], cigars + (size_t)i * cigar_stride, cigar_stride, sc);
    return rc;
}

int wo_clip_pairs_batch(uint32_t n_pairs, const wamd_clip_pair_task* tasks, wamd_clip_record* records,
                        uint8_t* cigars, uint32_t cigar_stride)
{
    int rc = 0;
    for (uint32_t i = 0; i < n_pairs; ++i)
        rc |= wo_clip_overlapping_pair(&records[2 * (size_t)i], &tasks[i], cigars + (size_t)(2 * i) * cigar_stride,
                                       cigars + (size_t)(2 * i + 1) * cigar_stride, cigar_stride);
    return rc;
}

/* ---- CSoftClipping::match + getExtCigar (soft_clipping.cpp:28-48, 139-231, 234-589) ------- */

#define SC_HASH_LEN 20u                      /* HashTable::hash_len, soft_clipping.cpp:25 */
#define SC_EMPTY 0xffffffffu                 /* HashTable::EMPTY_CELL, soft_clipping.cpp:24 */
#define SC_EMPTY_POS (1u << 30)              /* empty_pos_in_query, soft_clipping.h:102 */

static uint64_t sc_increment(uint64_t key, uint8_t symbol) /* soft_clipping.h:69-71 */
{
    return ((key << 2) + symbol) & ((1ull << 2 * SC_HASH_LEN) - 1);
}

uint32_t wo_softclip_table_size(uint32_t max_query_len) /* soft_clipping.cpp:45-48 */
{
    uint32_t ht_size;
    for (ht_size = 256; ht_size / max_query_len < 50; ht_size *= 2)
        ;
    return ht_size;
}

/* query: oriented read codes as preprocess / preprocessRawSeq leave them; tref / tread: what
 * ref_copy_direct and copy_direct / convert_to_rev_comp hand to getExtCigar. */
int wo_softclip(const uint8_t* genome, const uint8_t* query, uint32_t query_len, uint32_t ref_pos,
                uint32_t max_distance_in_ref, uint32_t min_exact_len, const wamd_softclip_params* prm,
                const wamd_scoring* sc, wamd_softclip_result* res, uint8_t* ext_cigar)
{
    const uint32_t W = max_distance_in_ref;
    const uint32_t ht_size = wo_softclip_table_size(prm->max_query_len), ht_mask = ht_size - 1;
    const uint64_t ht_mult = 0x5555555555555555ull;
    memset(res, 0, sizeof(*res));
    ext_cigar[0] = 0;
    if (query_len < SC_HASH_LEN || query_len > prm->max_query_len)
        return -1;
    uint32_t* ht = (uint32_t*)malloc(sizeof(uint32_t) * ht_size);
    uint8_t* gp = (uint8_t*)malloc((size_t)W + query_len + 16);
    uint32_t* checked_left = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)W + 16));
    uint32_t* checked_right = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)W + 16));
    uint32_t* left_mm = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)query_len + 1));
    uint32_t* right_mm = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)query_len + 1));
    memset(ht, 0xff, sizeof(uint32_t) * ht_size);
    for (uint32_t i = 0; i < W + 16; ++i) /* zero in a new object, empty_pos_in_query once touched and cleared */
        checked_left[i] = checked_right[i] = prm->fresh_object ? 0u : SC_EMPTY_POS;

    /* preprocess*, 139-231: every 20-mer of the read, in order, linear probing */
    uint64_t ht_val = 0;
    for (uint32_t i = 0; i < query_len; ++i) {
        ht_val = sc_increment(ht_val, query[i]);
        if (i >= SC_HASH_LEN - 1) {
            uint32_t p = (uint32_t)(ht_val * ht_mult) & ht_mask;
            while (ht[p] != SC_EMPTY)
                p = (p + 1) & ht_mask;
            ht[p] = i - (SC_HASH_LEN - 1);
        }
    }

    /* 242-243: dummy symbol, then the window and what an extension can run into */
    gp[0] = 7;
    for (uint32_t i = 0; i < W + query_len + 8; ++i)
        gp[1 + i] = (uint8_t)nib(genome, (uint64_t)ref_pos + i);

    uint32_t no_mismatches = 0;
    uint64_t key = 0;
    uint32_t best_text_pos = 0, best_query_pos = 0, best_exact_len = 0;
    int32_t best_score = 0, left_exact_len = 0, left_text_pos = 0;
    uint32_t best_left_len = 0, best_right_len = 0;
    uint32_t j;
    for (j = 1; j < SC_HASH_LEN; ++j) /* 259-260 */
        key = sc_increment(key, gp[j]);
    for (; j <= W; ++j) { /* 265-461 */
        key = sc_increment(key, gp[j]);
        uint32_t pos_in_query;
        uint32_t ht_pos = (uint32_t)(key * ht_mult) & ht_mask;
        while ((pos_in_query = ht[ht_pos]) != SC_EMPTY) {
            const uint32_t pos_in_text = j - (SC_HASH_LEN - 1);
            const uint32_t front = pos_in_text - pos_in_query;
            if (front <= W) {
                int32_t len = 0;
                if (checked_left[front] + 1 != pos_in_query) {
                    const uint8_t* qq = query + pos_in_query;
                    const uint8_t* gg = gp + pos_in_text;
                    const int len_range = (int)(query_len - pos_in_query);
                    for (; len < len_range && *qq == *gg; ++len, ++qq, ++gg)
                        ;
                    if (len >= (int)min_exact_len) {
                        if (len > best_score) { /* 295-303 */
                            best_score = len;
                            best_query_pos = pos_in_query;
                            best_text_pos = front;
                            best_left_len = 0;
                            best_right_len = 0;
                            best_exact_len = (uint32_t)len;
                        }
                        if (pos_in_query == 0) { /* 305-309 */
                            left_exact_len = len;
                            left_text_pos = (int32_t)front;
                        } else if (pos_in_query + (uint32_t)len == query_len && left_exact_len > 0) { /* 310-447 */
                            const int32_t text_dist = (int32_t)(pos_in_text + (uint32_t)len) - left_text_pos;
                            int exact_found = 0;
                            if (left_exact_len + len >= (int32_t)query_len) { /* deletion, 317-338 */
                                left_exact_len -= left_exact_len + len - (int32_t)query_len;
                                if (text_dist > (int32_t)query_len) {
                                    const int32_t score = (int32_t)query_len - (text_dist - (int32_t)query_len);
                                    exact_found = 1;
                                    if (score > best_score) {
                                        best_score = score;
                                        best_left_len = (uint32_t)left_exact_len;
                                        best_right_len = (uint32_t)len;
                                        best_query_pos = 0;
                                        best_text_pos = (uint32_t)left_text_pos;
                                        left_exact_len = 0;
                                        best_exact_len = 0;
                                        no_mismatches = 0;
                                    }
                                }
                            } else if (left_exact_len + len == text_dist) { /* insertion, 339-355 */
                                const int32_t score = text_dist;
                                exact_found = 1;
                                if (score > best_score) {
                                    best_score = score;
                                    best_left_len = (uint32_t)left_exact_len;
                                    best_right_len = (uint32_t)len;
                                    best_query_pos = 0;
                                    best_text_pos = (uint32_t)left_text_pos;
                                    left_exact_len = 0;
                                    best_exact_len = 0;
                                    no_mismatches = 0;
                                }
                            }
                            const int potential_indel_len = abs(text_dist - (int)query_len); /* 358-360 */
                            if (!exact_found && potential_indel_len >= (int)prm->min_approx_indel_len &&
                                potential_indel_len <= (int)prm->max_approx_indel_len &&
                                checked_right[front] != query_len) {
                                const uint32_t max_cmp_dist = (uint32_t)text_dist < query_len ? (uint32_t)text_dist : query_len;
                                const uint8_t* q0 = query;
                                const uint8_t* g0 = gp + left_text_pos;
                                left_mm[0] = 0;
                                for (uint32_t i = 0; i < max_cmp_dist; ++i)
                                    left_mm[i + 1] = left_mm[i] + (q0[i] != g0[i]);
                                q0 = query + query_len - 1;
                                g0 = gp + left_text_pos + text_dist - 1;
                                right_mm[0] = 0;
                                for (uint32_t i = 0; i < max_cmp_dist; ++i)
                                    right_mm[i + 1] = right_mm[i] + (*(q0 - i) != *(g0 - i));
                                if (text_dist > (int)query_len) { /* deletion, 383-412 */
                                    int mid_point = 0, best_val_mid = (int)query_len + 1;
                                    for (int i = 0; i <= (int)query_len; ++i)
                                        if (left_mm[i] + right_mm[query_len - i] < (uint32_t)best_val_mid) {
                                            best_val_mid = (int)(left_mm[i] + right_mm[query_len - i]);
                                            mid_point = i;
                                        }
                                    if (best_val_mid <= (int)prm->max_approx_indel_mismatches) {
                                        const int32_t score = (int32_t)query_len - (text_dist - (int32_t)query_len) - best_val_mid;
                                        if (score > best_score) {
                                            best_score = score;
                                            best_left_len = (uint32_t)mid_point;
                                            best_right_len = query_len - (uint32_t)mid_point;
                                            best_query_pos = 0;
                                            best_text_pos = (uint32_t)left_text_pos;
                                            left_exact_len = 0;
                                            best_exact_len = 0;
                                            no_mismatches = (uint32_t)best_val_mid;
                                        }
                                    }
                                } else { /* insertion, 413-442 */
                                    int mid_point = 0, best_val_mid = (int)query_len + 1;
                                    const int del_len = (int)query_len - text_dist;
                                    for (int i = 0; i <= (int)query_len - del_len; ++i)
                                        if (left_mm[i] + right_mm[text_dist - i] < (uint32_t)best_val_mid) {
                                            best_val_mid = (int)(left_mm[i] + right_mm[text_dist - i]);
                                            mid_point = i;
                                        }
                                    if (best_val_mid <= (int)prm->max_approx_indel_mismatches) {
                                        const int32_t score = text_dist - best_val_mid;
                                        if (score > best_score) {
                                            best_score = score;
                                            best_left_len = (uint32_t)mid_point;
                                            best_right_len = (uint32_t)(text_dist - mid_point);
                                            best_query_pos = 0;
                                            best_text_pos = (uint32_t)left_text_pos;
                                            left_exact_len = 0;
                                            best_exact_len = 0;
                                            no_mismatches = (uint32_t)best_val_mid;
                                        }
                                    }
                                }
                            }
                            checked_right[front] = query_len; /* 445 */
                        }
                    }
                }
                if (len >= (int32_t)min_exact_len) /* 451-455 */
                    checked_left[front] = pos_in_query;
            }
            if (++ht_pos >= ht_size) /* 459 */
                ht_pos = 0;
        }
    }

    int found = best_score >= (int32_t)min_exact_len; /* 463-464 */
    uint32_t pos = 0, left_clipping = 0, right_clipping = 0, left_match = 0, right_match = 0;
    int32_t del_size = 0;
    if (found) {
        pos = best_text_pos - 1;
        if (best_exact_len) { /* 466-474 */
            left_clipping = best_query_pos;
            right_clipping = query_len - best_query_pos - best_exact_len;
        } else { /* 475-492 */
            left_match = best_left_len;
            right_match = best_right_len;
            if (best_left_len + best_right_len == query_len)
                del_size = (int32_t)(query_len - ((uint32_t)best_score + no_mismatches));
            else
                del_size = -(int32_t)(query_len - left_match - right_match);
            if ((uint32_t)abs(del_size) > 2 * prm->max_approx_indel_len)
                found = 0;
        }
    }
    if (found) {
        res->found = 1;
        res->pos = pos;
        res->left_clipping = left_clipping;
        res->right_clipping = right_clipping;
        res->left_match = left_match;
        res->right_match = right_match;
        res->del_size = del_size;
        res->no_mismatches = no_mismatches;

        /* getExtCigar, 498-589.  tmp_ref_sequence[i] = min(genome[ref_pos + i], N_ref)
         * (ref_copy_direct, sam.cpp:2350-2375) = min(gp[i + 1], 5); tmp_read_sequence[i + 1] =
         * min(query[i], N_read) (copy_direct / convert_to_rev_comp, sam.cpp:2296-2347). */
        static const char decode[] = "ACGTNNXX";
#define TREF(i) (gp[(i) + 1] > 5 ? 5 : gp[(i) + 1])
#define TREAD(i) (query[i] > 4 ? 4 : query[i])
        const uint32_t seq_len = query_len;
        if (left_match == 0 && right_match == 0) {
            memset(ext_cigar, '$', left_clipping);
            memset(ext_cigar + left_clipping, '.', seq_len - right_clipping - left_clipping);
            memset(ext_cigar + seq_len - right_clipping, '$', right_clipping);
            ext_cigar[seq_len] = 0;
            res->cigar_len = seq_len;
            res->affine_score = (seq_len - left_clipping - right_clipping) * sc->match +
                                ((left_clipping > 0) ? sc->clipping : 0) + ((right_clipping > 0) ? sc->clipping : 0);
        } else {
            int cigar_pos = 0, read_pos = 0;
            uint32_t p = pos;
            if (no_mismatches) {
                for (int i = 0; i < (int)left_match; ++i, ++p, ++read_pos)
                    ext_cigar[cigar_pos++] = TREF(p) == TREAD(read_pos) ? '.' : (uint8_t)decode[TREF(p)];
            } else {
                memset(ext_cigar, '.', left_match);
                cigar_pos += (int)left_match;
                read_pos += (int)left_match;
                p += left_match;
            }
            if (del_size > 0) {
                for (int i = 0; i < del_size; ++i, ++p) {
                    ext_cigar[cigar_pos++] = '^';
                    ext_cigar[cigar_pos++] = (uint8_t)decode[TREF(p)];
                }
            } else {
                for (int i = 0; i < -del_size; ++i, ++read_pos) {
                    ext_cigar[cigar_pos++] = '#';
                    ext_cigar[cigar_pos++] = (uint8_t)decode[TREAD(read_pos)];
                }
            }
            if (no_mismatches) {
                for (int i = 0; i < (int)right_match; ++i, ++p, ++read_pos)
                    ext_cigar[cigar_pos++] = TREF(p) == TREAD(read_pos) ? '.' : (uint8_t)decode[TREF(p)];
            } else {
                memset(ext_cigar + cigar_pos, '.', right_match);
                cigar_pos += (int)right_match;
            }
            ext_cigar[cigar_pos] = 0;
            res->cigar_len = (uint32_t)cigar_pos;
            res->affine_score = (double)((int64_t)left_match + right_match - no_mismatches) * sc->match +
                                no_mismatches * sc->mismatch;
            if (del_size > 0)
                res->affine_score += sc->gap_del_open + (double)(abs(del_size) - 1ll) * sc->gap_del_extend;
            else if (del_size < 0)
                res->affine_score += sc->gap_ins_open + (double)(abs(del_size) - 1ll) * sc->gap_ins_extend;
        }
#undef TREF
#undef TREAD
    }
    free(ht);
    free(gp);
    free(checked_left);
    free(checked_right);
    free(left_mm);
    free(right_mm);
    return found;
}

int wo_softclip_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets, const uint32_t* lens,
                      int encoding, const wamd_softclip_task* tasks, uint32_t n, const wamd_softclip_params* prm,
                      const wamd_scoring* sc, wamd_softclip_result* results, uint8_t* cigars, uint32_t cigar_stride)
{
    uint8_t* q = (uint8_t*)malloc(64 * WO_MAX_WORDS + 8);
    int rc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const wamd_softclip_task* t = &tasks[i];
        const uint32_t m = lens[t->read_idx];
        uint8_t* ec = cigars + (size_t)i * cigar_stride;
        if (m + 4 * prm->max_approx_indel_len + 1 > cigar_stride) {
            rc = 1;
            memset(&results[i], 0, sizeof(results[i]));
            ec[0] = 0;
            continue;
        }
        { /* preprocess (rev_comp_code: 8..15 -> 7) / preprocessRawSeq, soft_clipping.cpp:50-76, 139-231 */
            const uint8_t* seq = reads + offsets[t->read_idx];
            const int rev = t->flags & WAMD_TASK_FLAG_REV_COMP;
            for (uint32_t z = 0; z < m; ++z)
                q[z] = encoding == WAMD_READS_NIBBLE
                           ? (uint8_t)(rev ? rc_code(nib(seq, m - z - 1)) : nib(seq, z))
                           : (uint8_t)ascii_code((rev ? seq[m - z - 1] : seq[z]) & 0x7f, rev);
        }
        if (wo_softclip(genome, q, m, t->ref_pos, t->max_distance_in_ref, t->min_exact_len, prm, sc, &results[i], ec) < 0)
            rc = 1;
    }
    free(q);
    return rc;
}

/* ---- wire records ------------------------------------------------------------- */

static uint8_t indel_size_encode(int32_t x) /* types.h:296-302 */
{
    return x > 0 ? (uint8_t)x : (uint8_t)(-x + 128);
}

uint32_t wo_serialize_mapping(const wamd_mapping* m, int direct, uint8_t* p)
{
    uint8_t* p0 = p;
    /* StoreUInt(ptr, pos, 4): big-endian, common/utils.cpp:431-445 */
    *p++ = (uint8_t)(m->pos >> 24);
    *p++ = (uint8_t)(m->pos >> 16);
    *p++ = (uint8_t)(m->pos >> 8);
    *p++ = (uint8_t)m->pos;
    *p++ = (uint8_t)((direct ? 0x80u : 0) + (m->type == WAMD_MAP_LEV ? 0x40u : 0) + (m->no_mismatches & 0x3f));
    if (m->type == WAMD_MAP_LEV)
        return (uint32_t)(p - p0);
    *p++ = (uint8_t)m->type;
    switch (m->type) {
    case WAMD_MAP_INDEL1:
        *p++ = (uint8_t)((m->matching_len1 >> 8) & 0xf);
        *p++ = (uint8_t)(m->matching_len1 & 0xff);
        *p++ = indel_size_encode(m->indel_len1);
        break;
    case WAMD_MAP_INDEL2:
        *p++ = (uint8_t)(((m->matching_len1 >> 8) & 0xf) + ((m->matching_len2 >> 4) & 0xf0u));
        *p++ = (uint8_t)(m->matching_len1 & 0xffu);
        *p++ = (uint8_t)(m->matching_len2 & 0xffu);
        *p++ = indel_size_encode(m->indel_len1);
        *p++ = indel_size_encode(m->indel_len2);
        break;
    case WAMD_MAP_INDEL_CLIPPING:
    case WAMD_MAP_CLIPPING_INDEL:
        *p++ = (uint8_t)(((m->matching_len1 >> 8) & 0xf) + ((m->clipping_len >> 4) & 0xf0u));
        *p++ = (uint8_t)(m->matching_len1 & 0xffu);
        *p++ = (uint8_t)(m->clipping_len & 0xffu);
        *p++ = indel_size_encode(m->indel_len1);
        break;
    case WAMD_MAP_MISMATCHES_CLIPPING:
    case WAMD_MAP_CLIPPING_MISMATCHES:
        *p++ = (uint8_t)((m->matching_len1 >> 8) & 0xf);
        *p++ = (uint8_t)(m->matching_len1 & 0xffu);
        break;
    case WAMD_MAP_CLIPPING_CLIPPING:
        *p++ = (uint8_t)(((m->matching_len1 >> 8) & 0xf) + ((m->clipping_len >> 4) & 0xf0u));
        *p++ = (uint8_t)(m->matching_len1 & 0xffu);
        *p++ = (uint8_t)(m->clipping_len & 0xffu);
        break;
    default:
        break;
    }
    return (uint32_t)(p - p0);
}

/* ---- batch drivers --------------------------------------------------------- */

int wo_indel_match_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                         const uint32_t* lens, int encoding, const wamd_indel_task* tasks, uint32_t n,
                         const wamd_indel_params* prm, wamd_mapping* results)
{
    uint8_t* q = (uint8_t*)malloc(64 * WO_MAX_WORDS + 8);
    for (uint32_t i = 0; i < n; ++i) {
        const wamd_indel_task* t = &tasks[i];
        const uint32_t m = lens[t->read_idx];
        wo_indel_query(q, reads + offsets[t->read_idx], m, encoding, t->flags & WAMD_TASK_FLAG_REV_COMP);
        wo_indel_match(genome, q, m, t->ref_pos, t->max_indel_size, t->fixed_segment_pos,
                       t->fixed_segment_size, t->max_no_mismatches, prm, &results[i]);
    }
    free(q);
    return 0;
}

int wo_indel_ext_cigar_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                             const uint32_t* lens, int encoding, const wamd_indel_cigar_task* tasks,
                             uint32_t n, const wamd_scoring* sc, wamd_indel_cigar_result* results,
                             uint8_t* cigars, uint32_t cigar_stride)
{
    uint8_t* q = (uint8_t*)malloc(64 * WO_MAX_WORDS + 8);
    for (uint32_t i = 0; i < n; ++i) {
        const wamd_indel_cigar_task* t = &tasks[i];
        const uint32_t m = lens[t->read_idx];
        wo_indel_query(q, reads + offsets[t->read_idx], m, encoding, t->flags & WAMD_TASK_FLAG_REV_COMP);
        wo_indel_ext_cigar(genome, q, m, t, sc, &results[i], cigars + (size_t)i * cigar_stride);
    }
    free(q);
    return 0;
}

int wo_clip_boundary_batch(uint32_t n, wamd_clip_record* records, uint8_t* cigars,
                           uint32_t cigar_stride, const wamd_scoring* sc)
{
    for (uint32_t i = 0; i < n; ++i)
        wo_clip_boundary(&records[i], cigars + (size_t)i * cigar_stride, sc);
    return 0;
}

int wo_serialize_batch(uint32_t n, const wamd_mapping* m, const uint8_t* direction, uint8_t* out,
                       uint8_t* sizes)
{
    for (uint32_t i = 0; i < n; ++i)
        sizes[i] = (uint8_t)wo_serialize_mapping(&m[i], direction[i], out + (size_t)i * 12);
    return 0;
}


/* ------------------------------------------------ size-independent property check */

/* Replays an ext-CIGAR (format of getExtCigar, LevMyers64.cpp:392-520) over the read and the
 * genome window it claims to align: '.' read == ref, a letter = mismatch naming the ref base,
 * "^x" deletion of ref base x, "#y" insertion of read base y.  Not a restatement of anything:
 * an independent consistency check the tests run at sizes the oracle itself is too slow for.
 * fail[i] = 0 when task i is consistent (or not ok), else a small code saying what broke. */
int wo_check_ext_cigar_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets,
                             const uint32_t* lens, const wamd_dist_task* tasks, uint32_t n,
                             const wamd_align_result* results, const uint8_t* cigars, uint32_t cigar_stride,
                             uint8_t* fail)
{
    static const char decode[] = "ACGTNNXX";
    uint8_t read_codes[64 * WO_MAX_WORDS + 2];
    for (uint32_t i = 0; i < n; ++i) {
        const wamd_dist_task* t = &tasks[i];
        const wamd_align_result* r = &results[i];
        fail[i] = 0;
        if (!r->ok)
            continue;
        const uint32_t m = lens[t->read_idx];
        wo_read_codes(read_codes, reads + offsets[t->read_idx], m, t->flags & WAMD_TASK_FLAG_REV_COMP);
        const uint8_t* c = cigars + (uint64_t)i * cigar_stride;
        uint32_t rp = 0, gp = r->begin_pos, edits = 0, opens = 0, k = 0;
        int in_gap = 0;
        if (r->cigar_len + 1 > cigar_stride || c[r->cigar_len] != 0) { fail[i] = 1; continue; }
        while (k < r->cigar_len && !fail[i]) {
            const uint8_t ch = c[k];
            uint32_t g = nib(genome, (uint64_t)t->ref_pos + gp);
            g = g > 5 ? 5 : g;
            if (ch == '.') {
                if (rp >= m || read_codes[rp] != g) fail[i] = 2;
                ++rp; ++gp; ++k; in_gap = 0;
            } else if (ch == '^') {
                if (c[k + 1] != (uint8_t)decode[g & 7]) fail[i] = 3;
                ++gp; k += 2; ++edits; opens += !in_gap; in_gap = 1;
            } else if (ch == '#') {
                if (rp >= m || c[k + 1] != (uint8_t)decode[read_codes[rp] & 7]) fail[i] = 4;
                ++rp; k += 2; ++edits; opens += !in_gap; in_gap = 1;
            } else {
                /* mismatch: names the reference base; the read base differs (a refined triple may
                 * hold an accidental match, so only the naming is checked when refined) */
                if (rp >= m || ch != (uint8_t)decode[g & 7]) fail[i] = 5;
                else if (!r->refined && read_codes[rp] == g) fail[i] = 6;
                ++rp; ++gp; ++k; ++edits; in_gap = 0;
            }
        }
        if (fail[i])
            continue;
        if (rp != m) fail[i] = 7;
        else if (gp != r->end_pos) fail[i] = 8;
        else if (edits != r->edit_distance + (r->refined ? 1u : 0u)) fail[i] = 9;
        else if (!r->refined && opens != r->num_events) fail[i] = 10;
    }
    return 0;
}


/* ------------------------------------------------------------- CF pre-filter (row C) */

/* CF_SSE_lut (mapping_core.cpp:196-217): a byte holding two ACGT symbols selects the one-hot
 * vector of its dinucleotide, every other byte the zero vector; add_saturated on int8 lanes. */
static void cf_add(int8_t* cf, uint32_t byte)
{
    const uint32_t a = (byte >> 4) & 15u, b = byte & 15u;
    if (a < 4 && b < 4 && cf[4 * a + b] < 127)
        ++cf[4 * a + b];
}

/* one region of calculate_CF_128b_SEE_for_dir_and_rc_gen, mapping_core_dp.cpp:361-391 / 395-427 */
static int cf_genome_region(int8_t* cf, const uint8_t* ref_ptr, uint32_t left_end, uint32_t len_in_chars)
{
    const uint8_t* T_ptr = ref_ptr + (left_end >> 1);
    if (left_end & 1) {
        uint32_t byte = (uint32_t)(*T_ptr & 15u) << 4;
        T_ptr++;
        byte += *T_ptr >> 4;
        cf_add(cf, byte);
        --len_in_chars;
    }
    if (len_in_chars == 0)
        return 1;   /* (len_in_chars - 1) >> 1 underflows in the reference */
    const uint32_t no_full_bytes = (len_in_chars - 1) >> 1;
    for (uint32_t i = 0; i < no_full_bytes; ++i) {
        cf_add(cf, *T_ptr);
        uint32_t byte = (uint32_t)(*T_ptr & 15u) << 4;
        T_ptr++;
        byte += *T_ptr >> 4;
        cf_add(cf, byte);
    }
    if ((len_in_chars & 1) == 0)
        cf_add(cf, *T_ptr);
    return 0;
}

int wo_cf_genome(const uint8_t* ref_ptr, uint32_t ref_size, uint32_t cur_offset, uint32_t fixed_left,
                 uint32_t fixed_right, uint32_t pattern_len, uint32_t position, int dir_gen_flag, int8_t* cf)
{
    uint32_t left_end = dir_gen_flag ? position - cur_offset : ref_size - position - pattern_len + cur_offset;
    int bad = 0;
    memset(cf, 0, 16);
    if (fixed_left)
        bad |= cf_genome_region(cf, ref_ptr, left_end, fixed_left);
    if (fixed_right != pattern_len)
        bad |= cf_genome_region(cf, ref_ptr, left_end + fixed_right, pattern_len - fixed_right);
    return bad;
}

/* calculate_CF_128b_SSE_for_a_pattern, mapping_core_dp.cpp:431-483 */
int wo_cf_pattern(const uint8_t* pattern, const uint8_t* pattern_sft, uint32_t pattern_len, uint32_t fixed_left,
                  uint32_t fixed_right, int8_t* cf)
{
    const uint8_t *p, *q;
    memset(cf, 0, 16);
    if (fixed_left) {
        const uint32_t len_in_chars = fixed_left;
        const uint32_t no_full_bytes = (len_in_chars - 1) / 2;
        p = &pattern[0];
        q = &pattern_sft[1];
        for (uint32_t i = 0; i < no_full_bytes; ++i) {
            cf_add(cf, *p++);
            cf_add(cf, *q++);
        }
        if ((len_in_chars & 1) == 0)
            cf_add(cf, *p++);
    }
    if (fixed_right != pattern_len) {
        const uint32_t len_in_chars = pattern_len - fixed_right;
        const uint32_t no_full_bytes = (len_in_chars - 1) / 2;
        const uint32_t fixed_right_div2 = fixed_right / 2;
        if (fixed_right & 1) {
            p = &pattern_sft[fixed_right_div2 + 1];
            q = &pattern[fixed_right_div2 + 1];
        } else {
            p = &pattern[fixed_right_div2];
            q = &pattern_sft[fixed_right_div2 + 1];
        }
        for (uint32_t i = 0; i < no_full_bytes; ++i) {
            cf_add(cf, *p++);
            cf_add(cf, *q++);
        }
        if ((len_in_chars & 1) == 0)
            cf_add(cf, *p++);
    }
    return 0;
}

/* test_CF_128b_SSE_difference's left-hand side, mapping_core.h:260-263.  VCL's
 * horizontal_add(Vec16c) truncates the sum to 8 bits and sign-extends it (libs/vectori128.h),
 * and the comparison with the unsigned max_diff is unsigned: sums of 128..255 never pass,
 * sums of 256 and more wrap around to small values.  Returned as the caller compares it. */
uint32_t wo_cf_difference(const int8_t* a, const int8_t* b)
{
    uint32_t s = 0;
    for (int i = 0; i < 16; ++i) {
        const int8_t d = (int8_t)(a[i] - b[i]);
        s += (uint32_t)(d < 0 ? -d : d);
    }
    return (uint32_t)(int32_t)(int8_t)(s & 0xffu);
}

int wo_cf_batch(const uint8_t* genome, uint32_t ref_size, uint32_t cur_offset, const uint8_t* reads,
                const uint64_t* offsets, const uint32_t* lens, const wamd_cf_task* tasks, uint32_t n,
                uint32_t* results)
{
    uint8_t* data = (uint8_t*)malloc(64 * WO_MAX_WORDS / 2 + 4);
    uint8_t* data_sft = (uint8_t*)malloc(64 * WO_MAX_WORDS / 2 + 4);
    int rc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const wamd_cf_task* t = &tasks[i];
        const uint32_t m = lens[t->read_idx];
        const uint8_t* src = reads + offsets[t->read_idx];
        int8_t cf_r[16], cf_g[16];
        results[i] = 0xffffffffu;
        if ((uint32_t)t->read_offset + t->pattern_len > m || t->fixed_left > t->fixed_right ||
            t->fixed_right > t->pattern_len || t->fixed_left == 1 || t->pattern_len - t->fixed_right == 1) {
            rc = WAMD_ERR_INVALID_ARG;
            continue;
        }
        /* the read in the orientation asked for, packed as the mapper holds it, and its copy
         * shifted by one symbol (reads.cpp:381-391) */
        memset(data, 0, (m + 1) / 2 + 2);
        for (uint32_t j = 0; j < m; ++j) {
            uint32_t c = (t->flags & WAMD_TASK_FLAG_REV_COMP) ? rc_code(nib(src, m - 1 - j)) : nib(src, j);
            data[j >> 1] |= (uint8_t)((j & 1) ? c : c << 4);
        }
        data_sft[0] = data[0] >> 4;
        for (uint32_t j = 1; j <= m / 2 + 1; ++j)
            data_sft[j] = (uint8_t)(((data[j - 1] & 15u) << 4) + (data[j] >> 4));
        /* window start as at the call sites (mapping_core_diff_len_reads.cpp:118-126) */
        const uint32_t d = t->read_offset;
        if (d & 1)
            wo_cf_pattern(data_sft + (d >> 1) + 1, data + (d >> 1), t->pattern_len, t->fixed_left, t->fixed_right, cf_r);
        else
            wo_cf_pattern(data + (d >> 1), data_sft + (d >> 1), t->pattern_len, t->fixed_left, t->fixed_right, cf_r);
        if (wo_cf_genome(genome, ref_size, cur_offset, t->fixed_left, t->fixed_right, t->pattern_len, t->position,
                         t->dir_gen_flag, cf_g)) {
            rc = WAMD_ERR_INVALID_ARG;
            continue;
        }
        results[i] = wo_cf_difference(cf_r, cf_g);
    }
    free(data);
    free(data_sft);
    return rc;
}


/* ---------------------------------------------- ext-CIGAR -> SAM CIGAR + MD:Z (row X) */

/* CNumericConversions::Int2PChar (common/utils.h:219-): plain decimal, no terminator */
static int int2pchar(uint32_t v, uint8_t* str)
{
    char tmp[12];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    for (int i = 0; i < n; ++i)
        str[i] = (uint8_t)tmp[n - 1 - i];
    return n;
}

/* CSamGenerator::extract_cigar_and_mdz (sam.cpp:805-928) and extract_cigarbin_and_mdz
 * (sam.cpp:932-1051) in one pass: the two differ only in how an operation is written. */
void wo_sam_cigar(const uint8_t* ext_cigar, uint8_t* cigar, uint8_t* mdz, uint32_t* cigar_bin,
                  wamd_sam_cigar_len* len)
{
    enum { Match, Mismatch, Insertion, Deletion, Begin, End, Clipping };
    enum { M_op, I_op, D_op, N_op, S_op };
    const uint8_t* ext_c = ext_cigar;
    uint8_t *cigar_c = cigar, *mdz_c = mdz;
    uint32_t* bin_c = cigar_bin;
    int counters[7] = {0};
    int cnt_match_mismatch = 0, cnt_match_ignoring_insert = 0;
    int cur = Begin, prev = Begin;
    while (cur != End) {
        if (*ext_c == '$') {
            cur = Clipping;
            ++ext_c;
        } else if (*ext_c == 0) {
            cur = End;
            if (cnt_match_ignoring_insert > 0)
                mdz_c += int2pchar((uint32_t)cnt_match_ignoring_insert, mdz_c);
        } else if (*ext_c == '.') {
            ++ext_c;
            cur = Match;
        } else if (*ext_c == '#') {
            ext_c += 2;
            cur = Insertion;
        } else {
            mdz_c += int2pchar((uint32_t)cnt_match_ignoring_insert, mdz_c);
            cnt_match_ignoring_insert = 0;
            if (*ext_c == '^') {
                *mdz_c++ = *ext_c++;
                *mdz_c++ = *ext_c++;
                cur = Deletion;
            } else {
                *mdz_c++ = *ext_c++;
                cur = Mismatch;
            }
        }
        ++counters[cur];
        if (cur == Match) {
            ++cnt_match_mismatch;
            ++cnt_match_ignoring_insert;
        } else if (cur == Mismatch) {
            ++cnt_match_mismatch;
        }
        if (cur != prev && prev != Begin) {
            int count = -1, op = 0;
            if (prev == Clipping) { count = counters[Clipping]; op = S_op; }
            else if (prev == Insertion) { count = counters[Insertion]; op = I_op; }
            else if (prev == Deletion) { count = counters[Deletion]; op = D_op; }
            else if (prev == Match) {
                if (cur != Mismatch) { count = cnt_match_mismatch; op = M_op; cnt_match_mismatch = 0; }
            } else if (prev == Mismatch) {
                if (cur != Match) { count = cnt_match_mismatch; op = M_op; cnt_match_mismatch = 0; }
            }
            if (count >= 0) {
                cigar_c += int2pchar((uint32_t)count, cigar_c);
                *cigar_c++ = (uint8_t)"MIDNS"[op];
                if (bin_c)
                    *bin_c++ = (uint32_t)count << 4 | (uint32_t)op;
            }
            counters[prev] = 0;
        }
        prev = cur;
    }
    *cigar_c = 0;
    *mdz_c = 0;
    len->cigar_len = (uint32_t)(cigar_c - cigar);
    len->mdz_len = (uint32_t)(mdz_c - mdz);
    len->cigar_bin_len = cigar_bin ? (uint32_t)(bin_c - cigar_bin) : 0;
    len->reserved = 0;
}

int wo_sam_cigar_batch(uint32_t n, const uint8_t* ext_cigars, uint32_t ext_stride, uint8_t* cigars,
                       uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride, uint32_t* cigar_bin,
                       uint32_t bin_stride, wamd_sam_cigar_len* lens)
{
    for (uint32_t i = 0; i < n; ++i)
        wo_sam_cigar(ext_cigars + (uint64_t)i * ext_stride, cigars + (uint64_t)i * cigar_stride,
                     mdz + (uint64_t)i * mdz_stride, cigar_bin ? cigar_bin + (uint64_t)i * bin_stride : NULL, &lens[i]);
    return 0;
}


/* ---- the chosen pair, from its two mapping_desc_t to the SAM fields -----------------------------
 * store_mapped_pair_reads, sam.cpp:1871-1900 and 1992-2007, restated as a chain of the routines
 * above with the glue of fill_cigar (sam_matching.cpp:667-686), fill_cigar_with_lev (689-752) and
 * fill_cigar_with_indel_matching (824-852).  Input: the descriptors as the generator constructs
 * them (sam.cpp:1555-1580); slot 2i = p.first, 2i + 1 = p.second. */
/* mapping_pair_t::calculateInsertSize, common/types.h:482-515 */
static int wo_insert_size(int32_t id0, int32_t pos0, int32_t len0, int direct0, int32_t id1, int32_t pos1, int32_t len1,
                          int direct1)
{
    if (id0 != id1)
        return 0x7fffffff;
    int left_is_first;
    if (pos0 < pos1)
        left_is_first = 1;
    else if (pos0 == pos1)
        left_is_first = direct0 ? 1 : 0;
    else
        left_is_first = 0;
    const int left_direct = left_is_first ? direct0 : direct1;
    if (direct0 == direct1 || !left_direct)
        return 0x7fffffff;
    return left_is_first ? pos1 - pos0 + len1 : pos0 - pos1 + len0;
}

int wo_finish_pairs_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets, const uint32_t* lens,
                          const wamd_finish_task* tasks, uint32_t n_pairs, const wamd_finish_params* prm,
                          const wamd_scoring* sc, wamd_finish_result* results, uint8_t* ext_cigars,
                          uint32_t ext_stride, uint8_t* cigars, uint32_t cigar_stride, uint8_t* mdz,
                          uint32_t mdz_stride)
{
    uint8_t* big_c = (uint8_t*)malloc(4 * (size_t)ext_stride + 64);
    uint8_t* big_m = (uint8_t*)malloc(4 * (size_t)ext_stride + 64);
    uint8_t* query = (uint8_t*)malloc(64 * WO_MAX_WORDS + 16);
    int rc = 0;
    for (uint32_t i = 0; i < n_pairs && !rc; ++i) {
        wamd_clip_record rec[2];
        int perfect[2] = {0, 0}, method[2] = {WAMD_METHOD_UNMATCHED, WAMD_METHOD_UNMATCHED};
        uint32_t pos[2], events[2] = {0, 0};
        int slot_ok[2] = {1, 1};
        for (int r = 0; r < 2; ++r) {   /* the two mappings as constructed, sam.cpp:1579-1581 */
            const wamd_finish_task* t = &tasks[2 * i + r];
            memset(&rec[r], 0, sizeof(rec[r]));
            rec[r].ref_seq_pos = t->ref_seq_pos;
            rec[r].ref_length = (uint16_t)lens[t->read_idx];
            rec[r].err_edit_distance = t->err_edit_distance;
            rec[r].score = -1;
            pos[r] = t->pos;
        }
        for (int r = 0; r < 2; ++r) {
            const uint32_t slot = 2 * i + r;
            const wamd_finish_task* t = &tasks[slot];
            uint8_t* ec = ext_cigars + (size_t)slot * ext_stride;
            const uint32_t m = lens[t->read_idx];
            ec[0] = 0;
            int distant = t->distant;
            if (distant == (int)WAMD_DISTANT_FROM_PAIR) {   /* sam.cpp:1881-1884: the pair as it stands now */
                const wamd_finish_task* t0 = &tasks[2 * i];
                const wamd_finish_task* t1 = &tasks[2 * i + 1];
                const int isize = wo_insert_size(t0->ref_seq_id, rec[0].ref_seq_pos, rec[0].ref_length, t0->direct,
                                                 t1->ref_seq_id, rec[1].ref_seq_pos, rec[1].ref_length, t1->direct);
                distant = isize > t->isize_max || isize < t->isize_min;
            }
            if (t->err_edit_distance == 0 && t->penalty == 0) {   /* sam_matching.cpp:670-680 */
                perfect[r] = 1;
                method[r] = distant ? WAMD_METHOD_PERFECT_DISTANT : WAMD_METHOD_PERFECT;
                rec[r].score = m * sc->match;
            } else if (t->type == WAMD_MAP_LEV) {                  /* sam_matching.cpp:689-752 */
                wamd_dist_task dt;
                wamd_align_result ar;
                const uint32_t begin = t->direct ? t->pos : t->pos - m + 1;
                dt.read_idx = t->read_idx;
                dt.ref_pos = begin - prm->flank;
                dt.max_distance_in_ref = m + 2 * prm->flank + 1;
                dt.max_ed = t->err_edit_distance;
                dt.variant = WAMD_VARIANT_AUTO;
                dt.flags = t->direct ? 0 : WAMD_TASK_FLAG_REV_COMP;
                rc = wo_align_batch(genome, reads, offsets, lens, WAMD_READS_ASCII, &dt, 1, 0,
                                    (int)prm->enable_short_indel_refinement, 16384, sc, &ar, ec, ext_stride);
                if (rc)
                    break;
                if (!ar.ok) {
                    /* stat_mapped_pe_errors: the reference goes on with a null ext-CIGAR and dereferences it in
                     * clipBoundaryPositions / extract_cigar_and_mdz - it does not survive this record.  Here the
                     * slot is reported as not finished, its mate is finished as usual, the pair is not pair-clipped. */
                    slot_ok[r] = 0;
                    ec[0] = 0;
                    continue;
                }
                const int64_t delta = ((int64_t)dt.ref_pos + ar.begin_pos) - (int64_t)begin;
                if (ar.refined)
                    rec[r].err_edit_distance = (uint16_t)(t->err_edit_distance + 1);
                pos[r] = (uint32_t)((int64_t)t->pos + delta);
                rec[r].ref_seq_pos = (int32_t)((int64_t)t->ref_seq_pos + delta);
                rec[r].ref_length = (uint16_t)(ar.end_pos - ar.begin_pos);
                method[r] = distant ? WAMD_METHOD_FROM_MAPPING_DISTANT : WAMD_METHOD_FROM_MAPPING;
                events[r] = ar.num_events;
                rec[r].score = ar.affine_score;
            } else {                                               /* sam_matching.cpp:824-852 */
                wamd_indel_cigar_task ct;
                wamd_indel_cigar_result cr;
                memset(&ct, 0, sizeof(ct));
                ct.read_idx = t->read_idx;
                ct.examined_start_pos = t->direct ? t->pos : t->pos - (m - 1);
                ct.type = t->type;
                ct.matching_len1 = t->matching_len1;
                ct.matching_len2 = t->matching_len2;
                ct.indel_len1 = t->indel_len1;
                ct.indel_len2 = t->indel_len2;
                ct.clipping_len = t->clipping_len;
                ct.flags = t->direct ? 0 : WAMD_TASK_FLAG_REV_COMP;
                wo_indel_query(query, reads + offsets[t->read_idx], m, WAMD_READS_ASCII, !t->direct);
                if (!wo_indel_ext_cigar(genome, query, m, &ct, sc, &cr, ec)) {
                    slot_ok[r] = 0;
                    ec[0] = 0;
                    continue;
                }
                if (!(t->flags & WAMD_FINISH_NO_TRANSLATE))       /* sam_matching.cpp:843-848 */
                    pos[r] = ct.examined_start_pos;
                rec[r].score = cr.score;
                rec[r].err_edit_distance = cr.err_edit_distance;
                rec[r].ref_length = cr.ref_length;
                rec[r].left_clipping = cr.left_clipping;
                rec[r].right_clipping = cr.right_clipping;
                method[r] = cr.method;
                events[r] = cr.num_events;
            }
            /* sam.cpp:1889-1892 */
            wamd_clip_illegal_task it;
            it.ref_size = t->ref_size;
            it.read_length = (uint16_t)m;
            it.perfect = (uint16_t)perfect[r];
            if (wo_clip_illegal(&rec[r], &it, ec, ext_stride, sc))
                rc = WAMD_ERR_INVALID_ARG;
            if (prm->enable_boundary_clipping)
                wo_clip_boundary(&rec[r], ec, sc);
        }
        if (rc)
            break;
        if (slot_ok[0] && slot_ok[1] && prm->enable_paired_clipping) {   /* sam.cpp:1898-1900 */
            wamd_clip_pair_task pt;
            for (int r = 0; r < 2; ++r) {
                pt.ref_seq_id[r] = tasks[2 * i + r].ref_seq_id;
                pt.read_length[r] = (uint16_t)lens[tasks[2 * i + r].read_idx];
                pt.direct[r] = tasks[2 * i + r].direct;
                pt.perfect[r] = (uint8_t)perfect[r];
            }
            if (wo_clip_overlapping_pair(rec, &pt, ext_cigars + (size_t)(2 * i) * ext_stride,
                                         ext_cigars + (size_t)(2 * i + 1) * ext_stride, ext_stride))
                rc = WAMD_ERR_INVALID_ARG;
        }
        for (int r = 0; r < 2; ++r) {
            const uint32_t slot = 2 * i + r;
            const uint32_t m = lens[tasks[slot].read_idx];
            wamd_finish_result* o = &results[slot];
            uint8_t* ec = ext_cigars + (size_t)slot * ext_stride;
            uint8_t* cg = cigars + (size_t)slot * cigar_stride;
            uint8_t* mz = mdz + (size_t)slot * mdz_stride;
            memset(o, 0, sizeof(*o));
            cg[0] = mz[0] = 0;
            if (!slot_ok[r]) {
                ec[0] = 0;
                continue;
            }
            o->ok = 1;
            o->score = rec[r].score;
            o->pos = pos[r];
            o->ref_seq_pos = rec[r].ref_seq_pos;
            o->ref_length = rec[r].ref_length;
            o->err_edit_distance = rec[r].err_edit_distance;
            o->num_events = (uint16_t)events[r];
            o->left_clipping = rec[r].left_clipping;
            o->right_clipping = rec[r].right_clipping;
            o->method = (uint16_t)method[r];
            if (!perfect[r]) {   /* sam.cpp:1992-2007 */
                wamd_sam_cigar_len l;
                wo_sam_cigar(ec, big_c, big_m, NULL, &l);
                if (l.cigar_len + 1 > cigar_stride || l.mdz_len + 1 > mdz_stride) {
                    rc = WAMD_ERR_INVALID_ARG;
                    break;
                }
                memcpy(cg, big_c, l.cigar_len + 1);
                memcpy(mz, big_m, l.mdz_len + 1);
            } else {             /* perfect_cigar / perfect_mdz, sam.cpp:1504-1515 */
                snprintf((char*)cg, cigar_stride, "%uM", m);
                snprintf((char*)mz, mdz_stride, "%u", m);
            }
            o->ext_cigar_len = (uint32_t)strlen((const char*)ec);
            o->cigar_len = (uint32_t)strlen((const char*)cg);
            o->mdz_len = (uint32_t)strlen((const char*)mz);
        }
    }
    free(big_c);
    free(big_m);
    free(query);
    return rc;
}


/* ---- verified candidates -> bytes of a result group -----------------------------------------------
 * commit: mapping_core_sa.cpp:241-247 / 293-299 (lev mapping), 258-272 / 310-326 (Match's mapping and its
 * strand fix-up), candidate_mapping_t::better (results.cpp:376-387), the penalty test of :338. */
int wo_commit_batch(uint32_t n, const wamd_commit_task* tasks, const wamd_mapping* indel, wamd_mapping* out)
{
    for (uint32_t i = 0; i < n; ++i) {
        const wamd_commit_task* t = &tasks[i];
        wamd_mapping cm;
        memset(&cm, 0, sizeof(cm));
        cm.type = WAMD_MAP_NONE;
        if (t->lev_errors <= t->max_mismatches) {
            cm.type = WAMD_MAP_LEV;
            cm.pos = t->lev_pos;
            cm.no_mismatches = t->lev_errors;
            cm.penalty = t->lev_errors;
        }
        if (t->has_indel && indel && indel[i].found) {
            wamd_mapping y = indel[i];
            if (!t->direct)
                y.pos += t->pattern_len - 1u;
            if (cm.type == WAMD_MAP_NONE) {
                cm = y;
            } else {   /* x < y ? x : y on (penalty, pos) */
                const int x_lt_y = cm.penalty != y.penalty ? cm.penalty < y.penalty : cm.pos < y.pos;
                if (!x_lt_y)
                    cm = y;
            }
        }
        if (cm.type != WAMD_MAP_NONE && cm.penalty > t->max_mismatches) {
            memset(&cm, 0, sizeof(cm));
            cm.type = WAMD_MAP_NONE;
        }
        cm.found = cm.type != WAMD_MAP_NONE;
        if (!cm.found) {
            memset(&cm, 0, sizeof(cm));
            cm.type = WAMD_MAP_NONE;
        }
        out[i] = cm;
    }
    return 0;
}

static int rg_less(const wamd_mapping* m, uint32_t a, uint32_t b)   /* operator<, results.cpp:367-373 */
{
    return m[a].penalty != m[b].penalty ? m[a].penalty < m[b].penalty : m[a].pos < m[b].pos;
}

static void rg_push_heap(uint32_t* h, const wamd_mapping* m, uint32_t hole, uint32_t top, uint32_t value)
{
    while (hole > top) {   /* libstdc++ std::__push_heap */
        const uint32_t parent = (hole - 1) / 2;
        if (!rg_less(m, h[parent], value))
            break;
        h[hole] = h[parent];
        hole = parent;
    }
    h[hole] = value;
}

/* mapping_core.cpp:976-997 with CMappingsHeapGatherer (results.cpp:409-452) and
 * CMappingResultsCollector::Push (results.cpp:546-598), one group buffer, reads in bin order. */
int wo_result_groups_batch(uint32_t n_tasks, const wamd_commit_task* tasks, const wamd_mapping* m, uint32_t n_reads,
                           const uint32_t* read_first, const uint64_t* read_ids, const wamd_group_params* prm,
                           uint8_t* out, uint32_t out_capacity, uint32_t* out_bytes)
{
    uint32_t* h = (uint32_t*)malloc(((size_t)n_tasks + 1) * sizeof(uint32_t));
    uint32_t used = 0;
    int rc = 0;
    for (uint32_t r = 0; r < n_reads && !rc; ++r) {
        uint32_t size = 0;
        for (uint32_t i = read_first[r]; i < read_first[r + 1]; ++i) {
            if (!m[i].found)
                continue;
            if (size < prm->max_no_mappings) {
                ++size;
                rg_push_heap(h, m, size - 1, 0, i);
            } else if (rg_less(m, i, h[0])) {
                /* std::pop_heap(first, last): value = last[-1]; last[-1] = first[0]; __adjust_heap(first, 0, len, value) */
                const uint32_t len = size - 1, value = h[size - 1];
                uint32_t hole = 0, child = 0;
                h[size - 1] = h[0];
                if (len > 0) {
                    while (child < (len - 1) / 2) {
                        child = 2 * (child + 1);
                        if (rg_less(m, h[child], h[child - 1]))
                            --child;
                        h[hole] = h[child];
                        hole = child;
                    }
                    if ((len & 1u) == 0 && child == (len - 2) / 2) {
                        child = 2 * (child + 1);
                        h[hole] = h[child - 1];
                        hole = child - 1;
                    }
                    rg_push_heap(h, m, hole, 0, value);
                }
                h[size - 1] = i;
                rg_push_heap(h, m, size - 1, 0, i);
            }
        }
        if (!size)
            continue;
        if ((uint64_t)used + prm->id_bytes + prm->mapping_counter_size + 11ull * size > out_capacity) {
            rc = WAMD_ERR_CAPACITY;
            break;
        }
        for (uint32_t b = 0; b < prm->id_bytes; ++b)   /* StoreUInt, common/utils.cpp:431-453 */
            out[used++] = (uint8_t)(read_ids[r] >> (8 * (prm->id_bytes - 1 - b)));
        for (uint32_t b = 0; b < prm->mapping_counter_size; ++b) {
            const uint32_t sh = 8 * (prm->mapping_counter_size - 1 - b);
            out[used++] = (uint8_t)(sh < 32 ? size >> sh : 0);
        }
        for (uint32_t j = size; j-- > 0;)              /* PopUnsorted: v_mappings[--size] */
            used += wo_serialize_mapping(&m[h[j]], tasks[h[j]].direct, out + used);
    }
    *out_bytes = used;
    free(h);
    return rc;
}

/* ---- find_pairs<LevMyers>'s chain per hit (sam_matching.cpp:379-457) -----------------------------------------------
 * The mate of a hit aligned in the window the insert-size model allows, made a mapping_desc_t, clipped. */
int wo_rescue_mates_batch(const uint8_t* genome, const uint8_t* reads, const uint64_t* offsets, const uint32_t* lens,
                          const wamd_rescue_task* tasks, uint32_t n, const wamd_rescue_params* prm, const wamd_scoring* sc,
                          wamd_rescue_result* results, uint8_t* ext_cigars, uint32_t ext_stride)
{
    int rc = 0;
    for (uint32_t i = 0; i < n && !rc; ++i) {
        const wamd_rescue_task* t = &tasks[i];
        wamd_rescue_result* o = &results[i];
        uint8_t* ec = ext_cigars + (size_t)i * ext_stride;
        const uint32_t m = lens[t->mate_read_idx];
        const uint32_t k = t->max_mate_edit_distance;
        memset(o, 0, sizeof(*o));
        ec[0] = 0;
        wamd_dist_task dt;
        wamd_align_result ar;
        /* :392-400, ref_pos_t arithmetic */
        dt.read_idx = t->mate_read_idx;
        dt.ref_pos = t->paired_dir_direct ? t->hit_pos + m - t->distance_threshold : t->hit_pos + 1u - k;
        dt.max_distance_in_ref = t->distance_threshold + 2u * k;
        dt.max_ed = (uint16_t)k;
        dt.variant = WAMD_VARIANT_AUTO;
        dt.flags = t->paired_dir_direct ? 0 : WAMD_TASK_FLAG_REV_COMP;
        rc = wo_align_batch(genome, reads, offsets, lens, WAMD_READS_ASCII, &dt, 1, 0,
                            (int)prm->enable_short_indel_refinement, 16384, sc, &ar, ec, ext_stride);
        if (rc)
            break;
        if (!ar.ok) {   /* :402 */
            ec[0] = 0;
            continue;
        }
        o->aligned = 1;
        o->edit_distance = (uint16_t)(ar.edit_distance + (ar.refined ? 1u : 0u));   /* :402, :425 */
        o->pos = dt.ref_pos + ar.begin_pos;                                     /* :422 */
        o->cigar_events = ar.num_events;
        o->ref_seq_pos = (int32_t)((int64_t)o->pos + t->seq_delta);            /* ref_desc.cpp:109-115 */
        const uint32_t p1 = o->pos + 1u;
        const int same_seq = p1 > t->seq_lo && p1 <= t->seq_hi;                /* Translate gives the hit's chromosome */
        if (!(same_seq && ar.num_events <= k / 2 && t->hit_num_events <= k / 2)) {   /* :424-427; the ext-CIGAR is freed */
            ec[0] = 0;
            continue;
        }
        o->ok = 1;
        wamd_clip_record rec;
        memset(&rec, 0, sizeof(rec));
        rec.ref_seq_pos = o->ref_seq_pos;
        rec.ref_length = (uint16_t)(ar.end_pos - ar.begin_pos);                /* :439 */
        rec.err_edit_distance = (uint16_t)(ar.edit_distance + (ar.refined ? 1u : 0u));   /* construct<lev>(.., edit_distance) */
        rec.score = ar.affine_score;                                           /* :442 */
        wamd_clip_illegal_task it;
        it.ref_size = t->ref_size;
        it.read_length = (uint16_t)m;
        it.perfect = 0;
        if (wo_clip_illegal(&rec, &it, ec, ext_stride, sc))                    /* :445 */
            rc = WAMD_ERR_INVALID_ARG;
        if (prm->enable_boundary_clipping)                                     /* :447-450 */
            wo_clip_boundary(&rec, ec, sc);
        o->score = rec.score;
        o->ref_seq_pos = rec.ref_seq_pos;
        o->ref_length = rec.ref_length;
        o->err_edit_distance = rec.err_edit_distance;
        o->left_clipping = rec.left_clipping;
        o->right_clipping = rec.right_clipping;
        o->ext_cigar_len = (uint32_t)strlen((const char*)ec);
    }
    return rc;
}
