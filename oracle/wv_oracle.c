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
