// myers_distance.hip - edit-distance verification, one candidate per lane.
//
// Replaces LevMyers{64,128,256}::dynamicProgramming in "distance only" mode
// (mapping stages: mapper/mapping_core_sa.cpp:237/289; also the first pass of the
// alignment path).  The recurrence is the Myers/Hyyro block algorithm exactly as
// the reference states it (mapper/LevMyers64.cpp:312-325): no carry through the
// add between 64-bit words, HN>>63 enters Eq and VP of the next word, HP>>63 the
// shifted HP.
//
// MI355X mapping
//   * one task per lane, all words of the bit-vectors in that lane's VGPRs:
//     carries between words never leave the lane, so the reference's
//     anti-diagonal skew (LevMyers256.hpp:138-159) is not needed; what the skew
//     changes observably - iteration count, sentinel iterations, early exit on
//     the iteration index - is carried over arithmetically;
//   * pattern-match vectors live in LDS, laid out [row][word][thread] so that a
//     wave's 64 lanes read 64 consecutive 8-byte slots whatever row (text
//     symbol) each lane asks for: ds_read_b64 without bank conflicts;
//   * the genome stays in the reference's 4-bit packing in HBM; every lane
//     streams its own window dword by dword (8 symbols), two dwords ahead, and
//     all lanes reload on the same iteration so the loop stays wave-uniform;
//   * no matrix is stored: the kernel is VALU-integer bound
//     (about 45 lane-ops per 64-bit word-column).
#include "wamd_internal.h"

namespace wamd {

struct LaneTask {
    uint32_t m;          // read length
    uint32_t wn;         // word holding the score bit
    uint32_t sb;         // score bit inside that word
    uint32_t n_cols;     // text columns the score word sees
    uint32_t exit_t;     // W + 1 - wn + k   (early exit: curr_ed + col > exit_t)
    uint32_t k;
    uint32_t ref_pos;
    uint32_t last_sym;
    uint32_t presence;
    bool skewed, is256, rev;
};

template <int NW, int BLOCK>
__global__ __launch_bounds__(BLOCK) void myers_distance_kernel(
    GenomeView g, ReadsView rv, const wamd_dist_task* __restrict__ tasks, uint32_t n,
    int last_symbol, uint32_t max_text_len, wamd_dist_result* __restrict__ results,
    uint32_t* __restrict__ flags)
{
    // rows 0..3 = text symbols A,C,G,T; row 4 = zeros (every other text symbol)
    __shared__ uint64_t pm_lds[5 * NW * BLOCK];

    const uint32_t tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * BLOCK + tid;
    bool valid = gid < n;

    LaneTask t{};
    uint32_t err = 0;
    uint32_t read_idx = 0;
    if (valid) {
        const wamd_dist_task task = tasks[gid];
        read_idx = task.read_idx;
        const uint32_t W = task.max_distance_in_ref;
        t.k = task.max_ed;
        t.ref_pos = task.ref_pos;
        t.rev = task.flags & WAMD_TASK_FLAG_REV_COMP;
        if (read_idx >= rv.n) {
            err = FLAG_READ_IDX;
        } else if ((t.rev ? rv.slot_rc : rv.slot_direct) < 0) {
            err = FLAG_ORIENT;
        } else if (W == 0 || W > max_text_len) {
            err = FLAG_TEXT_LEN;   // LevMyers64.cpp:244-252
        } else {
            t.m = rv.lens[read_idx];
            uint32_t variant = task.variant;
            if (variant == WAMD_VARIANT_AUTO)   // mapping_core_sa.cpp:207-210
                variant = t.m < 128 ? WAMD_VARIANT_M128 : (t.m < 256 ? WAMD_VARIANT_M256 : WAMD_VARIANT_M64);
            t.wn = (t.m - 1) / 64;
            t.sb = (t.m - 1) % 64;
            t.skewed = variant != WAMD_VARIANT_M64;
            t.is256 = variant == WAMD_VARIANT_M256;
            if (variant > WAMD_VARIANT_M256 || (variant == WAMD_VARIANT_M128 && t.m > 128) ||
                (variant == WAMD_VARIANT_M256 && t.m > 256) || t.wn >= (uint32_t)NW) {
                err = FLAG_VARIANT;
            } else {
                // skewed variants iterate j = 1..W+1 and the score lane runs wn columns
                // behind (LevMyers256.hpp:138,180); M64 iterates j = 1..W (LevMyers64.cpp:257)
                t.n_cols = t.skewed ? (W + 1 - t.wn) : W;
                if (t.skewed && W + 1 < t.wn)
                    t.n_cols = 0;
                t.exit_t = W + 1 - t.wn + t.k;
                if ((uint64_t)t.ref_pos + t.n_cols > g.n_symbols)
                    err = FLAG_RANGE;
            }
        }
        if (err) {
            raise_flag(flags, err);
            results[gid].edit_distance = RESULT_ERROR;
            results[gid].pos = err;
            valid = false;
        }
    }

    // ---- this lane's pattern-match vectors -> LDS ---------------------------------
    uint64_t* my_pm = pm_lds + tid;
    {
        const uint32_t slot = valid ? (uint32_t)(t.rev ? rv.slot_rc : rv.slot_direct) : 0;
        const uint64_t* src = rv.pm + ((uint64_t)read_idx * rv.n_orient + slot) * 4 * rv.words;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int w = 0; w < NW; ++w)
                my_pm[(c * NW + w) * BLOCK] = (valid && (uint32_t)w <= t.wn) ? src[c * rv.words + w] : 0;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            my_pm[(4 * NW + w) * BLOCK] = 0;
        if (valid) {
            const uint32_t meta = rv.meta[(uint64_t)read_idx * rv.n_orient + slot];
            t.presence = meta & 0xffffu;
            t.last_sym = last_symbol >= 0 ? (uint32_t)last_symbol : ((meta >> 16) & 15u);
        }
    }

    uint32_t min_ed = t.m, min_pos = 0, curr_ed = t.m;
    bool stopped = !valid;

    // ---- sentinel iterations of the skewed variants (j = 1..wn): the score lane
    // still sees "no symbol", only the bookkeeping runs
    // (LevMyers256.hpp:120-122, 206-234; LevMyers128.hpp:115, 199-221)
    if (valid && t.skewed) {
        for (uint32_t j = 1; j <= t.wn && j <= t.n_cols + t.wn; ++j) {
            if (t.is256)
                min_pos = j - t.wn;
            else if (min_pos + 1 == j - t.wn)
                min_pos = j - t.wn;
            if (curr_ed + j > t.exit_t + t.wn) {   // curr_ed > W + 1 - j + k
                stopped = true;
                break;
            }
        }
    }

    uint64_t VP[NW], VN[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        VP[w] = ~0ull;
        VN[w] = 0;
    }

    // is the score word the same (the top one) for every lane of this wave?
    const bool top_word = __all(!valid || t.wn == (uint32_t)(NW - 1));
    const uint32_t sb_lo = t.sb & 31u;
    const bool sb_hi = t.sb >= 32u;

    // ---- text stream: three swapped dwords in flight --------------------------------
    const uint32_t shift = (t.ref_pos & 7u) * 4u;
    uint32_t didx = t.ref_pos >> 3;
    uint32_t d_lo = load_genome_dword(g, didx);
    uint32_t d_hi = load_genome_dword(g, didx + 1);
    uint32_t d_nx = load_genome_dword(g, didx + 2);
    didx += 3;

    const uint8_t* my_codes = valid ? rv.codes + rv.code_off[read_idx] : nullptr;

    for (uint32_t base = 0; __any(!stopped && base < t.n_cols); base += 8) {
        uint32_t syms = __builtin_amdgcn_alignbit(d_hi, d_lo, shift);
        d_lo = d_hi;
        d_hi = d_nx;
        d_nx = load_genome_dword(g, didx++);

#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t col = base + u + 1;
            const uint32_t c = (syms >> (4 * u)) & 15u;
            const uint32_t row = c < 4u ? c : 4u;

            // a text symbol >= 4 matches only a read holding that very code, which
            // real data never does (genome: 5/15, reads: 4/6); stay exact anyway
            const bool exotic = c >= 4u && ((t.presence >> c) & 1u) && !stopped && col <= t.n_cols;
            if (exotic) {
                for (uint32_t w = 0; w <= t.wn; ++w) {
                    uint64_t eq = 0;
                    const uint32_t lo = w * 64, hi = min(t.m, lo + 64);
                    for (uint32_t i = lo; i < hi; ++i) {
                        const uint32_t rcode = t.rev ? rc_code(packed_symbol(my_codes, t.m - 1 - i))
                                                     : packed_symbol(my_codes, i);
                        eq |= (uint64_t)(rcode == c) << (i - lo);
                    }
                    my_pm[(4 * NW + w) * BLOCK] = eq;
                }
            }

            const uint64_t* eqrow = my_pm + row * (NW * BLOCK);
            uint64_t hn_in = 0, hp_in = 0;
            uint64_t sD0 = 0, sHP = 0, sHN = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                // LevMyers64.cpp:312-325
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
                if (top_word) {
                    if (w == NW - 1) {
                        sD0 = D0;
                        sHP = HP;
                        sHN = HN;
                    }
                } else if ((uint32_t)w == t.wn) {
                    sD0 = D0;
                    sHP = HP;
                    sHN = HN;
                }
            }
            if (exotic) {
                for (uint32_t w = 0; w <= t.wn; ++w)
                    my_pm[(4 * NW + w) * BLOCK] = 0;
            }

            // ---- score bookkeeping (LevMyers256.hpp:174-234, LevMyers128.hpp:175-221,
            // LevMyers64.cpp:328-366)
            const uint32_t hnb = ((sb_hi ? (uint32_t)(sHN >> 32) : (uint32_t)sHN) >> sb_lo) & 1u;
            const uint32_t hpb = ((sb_hi ? (uint32_t)(sHP >> 32) : (uint32_t)sHP) >> sb_lo) & 1u;
            const uint32_t d0b = ((sb_hi ? (uint32_t)(sD0 >> 32) : (uint32_t)sD0) >> sb_lo) & 1u;
            const bool act = !stopped && col <= t.n_cols;
            const uint32_t ced = curr_ed + hpb - hnb;
            const bool lt = ced < min_ed;
            const bool eq = ced == min_ed;
            const bool move = eq && (t.is256 ? (!d0b || c == t.last_sym) : (min_pos + 1 == col));
            if (act) {
                curr_ed = ced;
                if (lt)
                    min_ed = ced;
                if (lt || move)
                    min_pos = col;
                // early exit on the iteration index j = col + wn
                // (LevMyers256.hpp:232-233, LevMyers128.hpp:125, 219-220)
                if (t.skewed && !hnb && ced + col > t.exit_t)
                    stopped = true;
            }
        }
    }

    if (valid) {
        results[gid].edit_distance = min_ed;
        results[gid].pos = min_ed > t.k ? 0u : min_pos;
    }
}

template <int NW, int BLOCK>
static int launch_one(const GenomeView& g, const ReadsView& rv, const wamd_dist_task* d_tasks,
                      uint32_t n, int last_symbol, uint32_t max_text_len, wamd_dist_result* d_results,
                      uint32_t* d_flags, hipStream_t s)
{
    const uint32_t blocks = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL((myers_distance_kernel<NW, BLOCK>), dim3(blocks), dim3(BLOCK), 0, s, g, rv,
                       d_tasks, n, last_symbol, max_text_len, d_results, d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_distance(wamd_ctx* ctx, const wamd_reads* r, const wamd_dist_task* d_tasks, uint32_t n,
                    int last_symbol, wamd_dist_result* d_results, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    const GenomeView g = genome_view(ctx);
    const ReadsView rv = r->view();
    const uint32_t mt = ctx->max_text_len;
    uint32_t* f = ctx->d_flags;
    switch (rv.words) {
    case 1: return launch_one<1, 256>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, s);
    case 2: return launch_one<2, 256>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, s);
    case 3: return launch_one<3, 256>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, s);
    case 4: return launch_one<4, 256>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, s);
    default: break;
    }
    if (rv.words <= 8)
        return launch_one<8, 64>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, s);
    if (rv.words <= 16)
        return launch_one<16, 64>(g, rv, d_tasks, n, last_symbol, mt, d_results, f, s);
    set_error("reads of %u words are not supported", rv.words);
    return WAMD_ERR_TOO_LONG;
}

} // namespace wamd
