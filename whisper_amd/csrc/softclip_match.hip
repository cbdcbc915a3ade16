// softclip_match.hip - the hashing mate rescue, one candidate window per wave.
//
// Replaces CSoftClipping::{preprocess | preprocessRawSeq, match, getExtCigar}
// (mapper/soft_clipping.cpp:139-231, 234-495, 498-589): all 20-mers of the read go into an
// open-addressing table, the window is scanned for 20-mer hits, every hit is extended to a
// maximal exact run, and the answer is the longest run (-> clippings on both sides) or a
// left-boundary run + a right-boundary run = one long indel, optionally with a few
// mismatches around it.
//
// What the result depends on, and is therefore reproduced literally:
//   * the table itself - its size (from max_query_len), the multiplicative hash of which only
//     the low bits survive the mask (so 20-mers sharing their last ~7 symbols share a probe
//     chain), insertion order and linear probing: a chain is walked in cell order and every
//     entry on it is "a hit", real or not;
//   * the order of hits (text position, then chain order), because best-so-far, the remembered
//     left-boundary run and the checked_pos marks are all updated on the way;
//   * checked_pos_left's habit of suppressing only every other hit of a diagonal.
//
// Mapping to the wave: the 64 lanes look up 64 text positions at once (the table, the window
// and the read sit in LDS; only the low 15 bits of a key reach the table index, i.e. the last
// 8 symbols); positions with a non-empty first cell are then served one by one, in order, with
// wave-uniform state, and the expensive parts of serving a hit are cooperative again: the
// extension compares 512 symbols per step, the split search of the approximate indel is a
// wave scan over mismatch differences.  Blocks are persistent: the table is cleared through
// the insertion log, as the reference does.
//
// One window per wave means wave-uniform code: what it costs is instructions, 63 of 64 lanes idle.  Three
// shortcuts keep the serial parts short without touching their outcome (round 2: 6.75 -> 2.9 ms per 2^18
// windows of 660 symbols):
//   * insertion: a batch of 64 entries sits down at once when their cells are empty and distinct, else it is
//     redone in order, probing LDS only for entries whose cell was taken;
//   * hits whose chain holds one entry that agrees with the text on fewer than min_exact_len symbols are
//     dropped by their own lanes before the serial loop (serving them changes nothing);
//   * the hits of a read's own diagonal that only move the diagonal's mark are retired a stretch at a time.
#include "wamd_internal.h"

namespace wamd {

namespace {

constexpr uint32_t HASH_LEN = 20;          // HashTable::hash_len, soft_clipping.cpp:25
constexpr uint32_t EMPTY16 = 0xffffu;      // EMPTY_CELL / empty_pos_in_query in 16 bits

// CELL: a table cell holds a read position; bytes do when every read of the batch is short enough
// (positions <= 254), which halves the kernel's LDS and nearly doubles the waves per CU
template <typename CELL>
struct SoftclipLds {
    uint32_t* text;        // window, nibble-swapped dwords: symbol s of a dword at bits 4s
    uint32_t* query;       // oriented read codes, same packing
    CELL* table;           // HashTable::data
    uint16_t* cell;        // per read position: hashed cell, later the cell it was stored in (data_log)
    uint16_t* checked_left;
    uint32_t* checked_right;   // bit per text position: checked_pos_right[p] == query_len
};

__device__ __forceinline__ uint32_t sym(const uint32_t* a, uint32_t i) { return (a[i >> 3] >> (4 * (i & 7))) & 15u; }

// eight symbols i .. i+7 as one dword (symbol i in the low nibble)
__device__ __forceinline__ uint32_t sym8(const uint32_t* a, uint32_t i)
{
    return __builtin_amdgcn_alignbit(a[(i >> 3) + 1], a[i >> 3], 4 * (i & 7));
}

// low 15 bits of the rolling key after symbol i: HashTable::increment adds, it does not or, so
// codes above 3 carry into the next symbol's bits (soft_clipping.h:69-71)
__device__ __forceinline__ uint32_t key_low(const uint32_t* a, uint32_t i)
{
    const uint32_t w = sym8(a, i - 7);   // symbol i-7 in nibble 0 ... symbol i in nibble 7
    uint32_t k = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        k += ((w >> (4 * (7 - t))) & 15u) << (2 * t);
    return k & 0x7fffu;
}

__device__ __forceinline__ uint32_t wave_incl_scan(int32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(v, d);
        if (lane >= (uint32_t)d)
            v += o;
    }
    return (uint32_t)v;
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        v = min(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}

}   // namespace

template <typename CELL>
__global__ __launch_bounds__(64) void softclip_match_kernel(GenomeView g, ReadsView rv,
                                                            const wamd_softclip_task* __restrict__ tasks, uint32_t n,
                                                            wamd_softclip_params prm, wamd_scoring sc,
                                                            uint32_t ht_size, uint32_t max_window, uint32_t max_len,
                                                            uint32_t max_text_len,
                                                            wamd_softclip_result* __restrict__ results,
                                                            uint8_t* __restrict__ cigars, uint32_t cigar_stride,
                                                            uint32_t* __restrict__ flags)
{
    extern __shared__ uint32_t lds_raw[];
    const uint32_t lane = threadIdx.x;
    const uint32_t text_dw = (max_window + max_len + 16 + 7) / 8 + 2;
    const uint32_t query_dw = (max_len + 7) / 8 + 2;
    constexpr uint32_t EMPTY = (uint32_t)(CELL)~(CELL)0;   // EMPTY_CELL in a cell
    SoftclipLds<CELL> L;
    L.text = lds_raw;
    L.query = L.text + text_dw;
    L.checked_right = L.query + query_dw;
    L.table = reinterpret_cast<CELL*>(L.checked_right + (max_window + 32) / 32);
    L.cell = reinterpret_cast<uint16_t*>(L.table + ht_size);
    L.checked_left = L.cell + ((max_len + 1) & ~1u);
    const uint32_t ht_mask = ht_size - 1;

    for (uint32_t i = lane; i < ht_size * sizeof(CELL) / 4; i += 64)
        reinterpret_cast<uint32_t*>(L.table)[i] = 0xffffffffu;
    __syncthreads();

    for (uint32_t task = blockIdx.x; task < n; task += gridDim.x) {
        const wamd_softclip_task t = tasks[task];
        wamd_softclip_result res;
        res.affine_score = 0;
        res.found = res.pos = res.left_clipping = res.right_clipping = res.left_match = res.right_match = 0;
        res.del_size = 0;
        res.no_mismatches = res.cigar_len = res.reserved = 0;
        uint8_t* ec = cigars + (uint64_t)task * cigar_stride;

        uint32_t err = 0, m = 0;
        const uint32_t W = t.max_distance_in_ref;
        const bool rc = t.flags & WAMD_TASK_FLAG_REV_COMP;
        if (t.read_idx >= rv.n) {
            err = FLAG_READ_IDX;
        } else {
            m = rv.lens[t.read_idx];
            if (m < HASH_LEN || m > prm.max_query_len || m > max_len || m - HASH_LEN >= EMPTY)
                err = FLAG_ARG;   // the reference reads past the read / past its query buffer here
            else if (W > max_text_len || W > max_window)
                err = FLAG_TEXT_LEN;
            else if ((uint64_t)t.ref_pos + W + m + 16 > g.n_symbols)
                err = FLAG_RANGE;
            else if (m + 4 * prm.max_approx_indel_len + 1 > cigar_stride)
                err = FLAG_CIGAR;
        }
        if (err) {
            if (lane == 0) {
                raise_flag(flags, err);
                res.found = RESULT_ERROR;
                results[task] = res;
                ec[0] = 0;
            }
            continue;
        }

        // ---- stage the window, the oriented read, and clear the marks -------------------
        const uint32_t skew = t.ref_pos & 7u;          // gp[j] = text symbol skew + j - 1
        const uint32_t n_text_dw = (skew + W + m + 16 + 7) / 8;
        for (uint32_t i = lane; i < n_text_dw; i += 64)
            L.text[i] = load_genome_dword(g, (t.ref_pos >> 3) + i);
        {
            const uint32_t* codes = reinterpret_cast<const uint32_t*>(rv.codes + rv.code_off[t.read_idx]);
            const uint32_t n_dw = (m + 7) / 8;
            for (uint32_t i = lane; i <= n_dw; i += 64) {
                uint32_t w;
                if (!rc) {
                    w = i < n_dw ? nibble_swap(codes[i]) : 0u;
                } else {   // oriented symbol p = rc_code(stored symbol m - 1 - p)
                    w = 0;
                    for (uint32_t s = 0; s < 8; ++s) {
                        const uint32_t p = 8 * i + s;
                        if (p < m) {
                            const uint32_t z = m - 1 - p;
                            const uint32_t c = (nibble_swap(codes[z >> 3]) >> (4 * (z & 7))) & 15u;
                            w |= rc_code(c) << (4 * s);
                        }
                    }
                }
                L.query[i] = w;   // nibbles past the read are never compared: every run is bounded by its range
            }
        }
        const uint32_t never = prm.fresh_object ? 0u : 0xffffffffu;
        for (uint32_t i = lane; i <= (W + 1) / 2; i += 64)
            reinterpret_cast<uint32_t*>(L.checked_left)[i] = never;
        for (uint32_t i = lane; i <= W / 32; i += 64)
            L.checked_right[i] = 0;
        __syncthreads();

        // ---- preprocess: insert every 20-mer in order, linear probing (139-231) ----------
        // (the wave's DS operations execute in program order, so a probe sees every earlier insert
        // without a barrier; the hashed cells of 64 consecutive 20-mers sit one per lane)
        // A batch of 64 consecutive 20-mers first tries to sit down all at once: when every hashed cell is empty
        // and no two of them are the same cell, linear probing in order puts each entry in its own cell (64
        // entries for one LDS round trip; at 131 entries in 8192 cells about every other batch).  Otherwise the
        // batch is undone and inserted one by one - still without a round trip per entry while an entry's cell
        // was empty before the batch and no earlier entry of the batch ended up in it.
        for (uint32_t ib = HASH_LEN - 1; ib < m; ib += 64) {
            const uint32_t cnt = min(64u, m - ib);
            const bool valid = lane < cnt;
            const uint32_t mine = valid ? (key_low(L.query, ib + lane) * 0x5555u) & ht_mask : 0u;
            const uint32_t value = ib + lane - (HASH_LEN - 1);
            uint32_t pre = EMPTY;
            if (valid) {
                pre = L.table[mine];
                L.table[mine] = (CELL)value;
            }
            // (volatile: the compiler must not assume a lane reads back what it has just stored)
            const bool lost = valid && (pre != EMPTY || const_cast<volatile CELL*>(L.table)[mine] != (CELL)value);
            if (!__any(lost)) {
                if (valid)
                    L.cell[ib + lane] = (uint16_t)mine;
                continue;
            }
            if (valid)
                L.table[mine] = (CELL)pre;   // every writer of a cell saw the same earlier content
            uint32_t fin = 0xffffffffu;   // lane u: where entry u of this batch went
            for (uint32_t u = 0; u < cnt; ++u) {
                uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)u);
                const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)pre, (int)u);
                if (before != EMPTY || __any(fin == p)) {
                    while (L.table[p] != EMPTY)
                        p = (p + 1) & ht_mask;
                }
                fin = lane == u ? p : fin;
                if (lane == 0) {
                    L.table[p] = (CELL)(ib + u - (HASH_LEN - 1));
                    L.cell[ib + u] = (uint16_t)p;
                }
            }
        }
        __syncthreads();

        // ---- match (234-495) ------------------------------------------------------------
        const uint32_t min_exact = t.min_exact_len;
        uint32_t no_mismatches = 0;
        uint32_t best_text_pos = 0, best_query_pos = 0, best_exact_len = 0;
        int32_t best_score = 0, left_exact_len = 0, left_text_pos = 0;
        uint32_t best_left_len = 0, best_right_len = 0;
        // the last extension: query[run_q ..) and the text on diagonal run_front agree up to query index run_end
        uint32_t run_front = 0xffffffffu, run_q = 0, run_end = 0;
        // checked_pos_left of the text position touched last (hits of one run share it)
        uint32_t cl_front = 0xffffffffu, cl_val = 0;

        // symbol of gp[] / query[] helpers; gp[0] (the dummy 7) is never read below
        auto gp8 = [&](uint32_t j) { return sym8(L.text, skew + j - 1); };
        auto gp1 = [&](uint32_t j) { return sym(L.text, skew + j - 1); };

        for (uint32_t jb = HASH_LEN; jb <= W; jb += 64) {
            const uint32_t j = jb + lane;
            // every lane reads the first two cells of its chain now; serving a hit below then costs no
            // LDS round trip unless the chain is longer (readlane hands the values to the whole wave)
            uint32_t first_cell = 0, cell_q0 = EMPTY, cell_q1 = EMPTY;
            if (j <= W) {
                first_cell = (key_low(L.text, skew + j - 1) * 0x5555u) & ht_mask;
                cell_q0 = L.table[first_cell];
                cell_q1 = L.table[first_cell + 1 >= ht_size ? 0u : first_cell + 1];
            }
            // A chain of one entry whose read position and this text position agree on fewer than min_exact_len
            // symbols changes nothing when it is served (nothing below happens under that length): such hits -
            // the 20-mers that merely share their last eight symbols with a 20-mer of the read - drop out here,
            // all lanes comparing their own at once.
            bool dud = false;
            if (cell_q0 != EMPTY && cell_q1 == EMPTY && min_exact <= 64u) {
                const uint32_t q = cell_q0, pt = j - (HASH_LEN - 1);
                if (m - q < min_exact) {
                    dud = true;
                } else {
                    for (uint32_t o = 0; o < min_exact && !dud; o += 8) {
                        uint32_t diff = sym8(L.query, q + o) ^ gp8(pt + o);
                        if (min_exact - o < 8)
                            diff &= (1u << (4 * (min_exact - o))) - 1u;
                        dud = diff != 0;
                    }
                }
            }
            uint64_t pending = __ballot(cell_q0 != EMPTY && !dud);
            while (pending) {
                const int src = __builtin_ctzll(pending);
                // ---- hits that change nothing but the diagonal's mark, a whole stretch at once -------------
                // A read that matches its window produces one hit per text position on one diagonal, all
                // inside the run the first of them found.  Served one by one, such a hit (not the read's
                // first 20-mer, no longer than the best so far, and - when the run ends with the read and a
                // left-boundary run is remembered - neither completing an indel nor due for the approximate
                // search) only moves checked_pos_left of its diagonal, every other hit.  The lanes decide that
                // for their own hit in parallel; the leading stretch of such lanes is retired arithmetically.
                if (run_front <= W) {
                    const uint32_t q = cell_q0;
                    bool plain = ((pending >> lane) & 1ull) && cell_q1 == EMPTY && q != 0 && q >= run_q && q < run_end &&
                                 j - (HASH_LEN - 1) - q == run_front;
                    const int32_t len = (int32_t)(run_end - q);
                    const bool to_end = run_end == m && left_exact_len > 0;
                    int32_t text_dist = 0;
                    if (plain && len >= (int32_t)min_exact) {
                        plain = len <= best_score;
                        if (to_end) {
                            text_dist = (int32_t)(run_front + run_end) - left_text_pos;
                            plain = plain && left_exact_len + len < (int32_t)m && left_exact_len + len != text_dist;
                        }
                    }
                    const uint64_t pm = __ballot(plain) >> src;
                    uint32_t cnt = pm == ~0ull ? 64u - (uint32_t)src : (uint32_t)__builtin_ctzll(~pm);
                    if (cnt) {
                        const uint32_t qa = (uint32_t)__builtin_amdgcn_readlane((int)cell_q0, src);
                        // hits long enough to leave a mark: qa .. lim
                        const uint32_t qb = qa + cnt - 1;
                        const bool marks = run_end >= min_exact && qa <= run_end - min_exact;
                        bool ok = true;
                        if (marks && to_end) {   // the approximate search must not be due (358-360)
                            const int32_t td = (int32_t)(run_front + run_end) - left_text_pos;
                            const int32_t indel_len = abs(td - (int32_t)m);
                            if (indel_len >= (int32_t)prm.min_approx_indel_len && indel_len <= (int32_t)prm.max_approx_indel_len &&
                                !((L.checked_right[run_front >> 5] >> (run_front & 31)) & 1u))
                                ok = false;
                        }
                        if (ok) {
                            if (marks) {
                                const uint32_t lim = min(qb, run_end - min_exact);
                                if (run_front != cl_front) {
                                    cl_front = run_front;
                                    cl_val = L.checked_left[run_front];
                                }
                                const uint32_t cl = cl_val == EMPTY16 ? (1u << 30) : cl_val;
                                const uint32_t s0 = cl + 1 != qa ? qa : qa + 1;   // the first hit that is served
                                if (s0 <= lim) {
                                    const uint32_t last = s0 + ((lim - s0) & ~1u);
                                    cl_val = last;
                                    if (lane == 0) {
                                        L.checked_left[run_front] = (uint16_t)last;
                                        if (to_end)   // 445
                                            L.checked_right[run_front >> 5] |= 1u << (run_front & 31);
                                    }
                                }
                            }
                            pending = cnt >= 64u - (uint32_t)src ? 0ull : pending & ~(((1ull << cnt) - 1ull) << src);
                            continue;
                        }
                    }
                }
                pending &= pending - 1;
                const uint32_t jh = jb + (uint32_t)src;
                uint32_t ht_pos = (uint32_t)__builtin_amdgcn_readlane((int)first_cell, src);
                const uint32_t second = (uint32_t)__builtin_amdgcn_readlane((int)cell_q1, src);
                uint32_t pos_in_query = (uint32_t)__builtin_amdgcn_readlane((int)cell_q0, src);
                for (uint32_t link = 0; pos_in_query != EMPTY; ++link) {   // 273-460
                    const uint32_t pos_in_text = jh - (HASH_LEN - 1);
                    const uint32_t front = pos_in_text - pos_in_query;
                    if (front <= W) {
                        int32_t len = 0;
                        if (front != cl_front) {
                            cl_front = front;
                            cl_val = L.checked_left[front];
                        }
                        const uint32_t cl = cl_val;
                        if ((cl == EMPTY16 ? (1u << 30) : cl) + 1 != pos_in_query) {
                            // 286-291: common run of query[pos_in_query..] and gp[pos_in_text..]
                            const uint32_t range = m - pos_in_query;
                            len = (int32_t)range;
                            // a hit on the diagonal of the last extension, inside its run, ends where that run ended
                            const bool known = front == run_front && pos_in_query >= run_q && pos_in_query < run_end;
                            if (known)
                                len = (int32_t)(run_end - pos_in_query);
                            for (uint32_t off = 0; !known && off < range; off += 512) {
                                const uint32_t o = off + 8 * lane;
                                uint32_t diff = 0xffffffffu;
                                if (o < range) {
                                    diff = sym8(L.query, pos_in_query + o) ^ gp8(pos_in_text + o);
                                    if (range - o < 8)
                                        diff |= 0xffffffffu << (4 * (range - o));
                                }
                                const uint64_t bad = __ballot(diff != 0);
                                if (bad) {
                                    const uint32_t l0 = (uint32_t)__builtin_ctzll(bad);
                                    const uint32_t d0 = (uint32_t)__shfl((int)diff, (int)l0);
                                    const uint32_t stop = off + 8 * l0 + ((uint32_t)__builtin_ctz(d0) >> 2);
                                    len = (int32_t)min(stop, range);
                                    break;
                                }
                            }
                            if (!known) {
                                run_front = front;
                                run_q = pos_in_query;
                                run_end = pos_in_query + (uint32_t)len;
                            }
                            if (len >= (int32_t)min_exact) {
                                if (len > best_score) {   // 295-303
                                    best_score = len;
                                    best_query_pos = pos_in_query;
                                    best_text_pos = front;
                                    best_left_len = 0;
                                    best_right_len = 0;
                                    best_exact_len = (uint32_t)len;
                                }
                                if (pos_in_query == 0) {   // 305-309
                                    left_exact_len = len;
                                    left_text_pos = (int32_t)front;
                                } else if (pos_in_query + (uint32_t)len == m && left_exact_len > 0) {   // 310-447
                                    const int32_t text_dist = (int32_t)(pos_in_text + (uint32_t)len) - left_text_pos;
                                    bool exact_found = false;
                                    bool take = false;
                                    int32_t score = 0;
                                    uint32_t new_left = 0, new_right = 0;
                                    if (left_exact_len + len >= (int32_t)m) {   // deletion, 317-338
                                        left_exact_len -= left_exact_len + len - (int32_t)m;
                                        if (text_dist > (int32_t)m) {
                                            score = (int32_t)m - (text_dist - (int32_t)m);
                                            exact_found = true;
                                            take = score > best_score;
                                            new_left = (uint32_t)left_exact_len;
                                            new_right = (uint32_t)len;
                                        }
                                    } else if (left_exact_len + len == text_dist) {   // insertion, 339-355
                                        score = text_dist;
                                        exact_found = true;
                                        take = score > best_score;
                                        new_left = (uint32_t)left_exact_len;
                                        new_right = (uint32_t)len;
                                    }
                                    if (take) {
                                        best_score = score;
                                        best_left_len = new_left;
                                        best_right_len = new_right;
                                        best_query_pos = 0;
                                        best_text_pos = (uint32_t)left_text_pos;
                                        left_exact_len = 0;
                                        best_exact_len = 0;
                                        no_mismatches = 0;
                                    }
                                    const int32_t indel_len = abs(text_dist - (int32_t)m);   // 358-360
                                    if (!exact_found && indel_len >= (int32_t)prm.min_approx_indel_len &&
                                        indel_len <= (int32_t)prm.max_approx_indel_len &&
                                        !((L.checked_right[front >> 5] >> (front & 31)) & 1u)) {
                                        // 362-442.  With lm[i] = the left alignment mismatches at query i and
                                        // rm[i] = the right alignment mismatches at query i + qs,
                                        // left_mismatches[i] + right_mismatches[N - i] = sum(rm) + sum_{x<i} (lm - rm)
                                        const bool deletion = text_dist > (int32_t)m;
                                        const uint32_t N = deletion ? m : (uint32_t)text_dist;
                                        const uint32_t qs = deletion ? 0u : m - (uint32_t)text_dist;
                                        const uint32_t tl = (uint32_t)left_text_pos;                       // left: query i <-> gp[tl + i]
                                        const uint32_t tr = (uint32_t)(left_text_pos + text_dist) - m + qs;   // right: query i + qs <-> gp[tr + i]
                                        uint32_t total_rm = 0;
                                        for (uint32_t off = 0; off < N; off += 64) {
                                            const uint32_t i = off + lane;
                                            const bool rm = i < N && sym(L.query, i + qs) != gp1(tr + i);
                                            total_rm += (uint32_t)__popcll(__ballot(rm));
                                        }
                                        uint32_t best = total_rm << 16;   // i = 0
                                        uint32_t run = total_rm;
                                        for (uint32_t off = 0; off < N; off += 64) {
                                            const uint32_t i = off + lane;
                                            int32_t d = 0;
                                            if (i < N)
                                                d = (int32_t)(sym(L.query, i) != gp1(tl + i)) -
                                                    (int32_t)(sym(L.query, i + qs) != gp1(tr + i));
                                            const uint32_t incl = wave_incl_scan(d, lane);
                                            const uint32_t f = run + incl;
                                            best = min(best, wave_min(i < N ? (f << 16) | (i + 1) : 0xffffffffu));
                                            run += (uint32_t)__shfl((int)incl, 63);
                                        }
                                        const int32_t best_val_mid = (int32_t)(best >> 16);
                                        const uint32_t mid_point = best & 0xffffu;
                                        if (best_val_mid <= (int32_t)prm.max_approx_indel_mismatches) {
                                            const int32_t sc2 = deletion ? (int32_t)m - (text_dist - (int32_t)m) - best_val_mid
                                                                         : text_dist - best_val_mid;
                                            if (sc2 > best_score) {
                                                best_score = sc2;
                                                best_left_len = mid_point;
                                                best_right_len = N - mid_point;
                                                best_query_pos = 0;
                                                best_text_pos = (uint32_t)left_text_pos;
                                                left_exact_len = 0;
                                                best_exact_len = 0;
                                                no_mismatches = (uint32_t)best_val_mid;
                                            }
                                        }
                                    }
                                    if (lane == 0)   // 445
                                        L.checked_right[front >> 5] |= 1u << (front & 31);
                                }
                            }
                        }
                        if (len >= (int32_t)min_exact) {   // 451-455
                            if (lane == 0)
                                L.checked_left[front] = (uint16_t)pos_in_query;
                            cl_val = pos_in_query;
                        }
                    }
                    if (++ht_pos >= ht_size)   // 459
                        ht_pos = 0;
                    pos_in_query = link == 0 ? second : L.table[ht_pos];
                }
            }
        }

        // ---- outcome (463-494) -----------------------------------------------------------
        bool found = best_score >= (int32_t)min_exact;
        uint32_t pos = 0, left_clipping = 0, right_clipping = 0, left_match = 0, right_match = 0;
        int32_t del_size = 0;
        if (found) {
            pos = best_text_pos - 1;
            if (best_exact_len) {
                left_clipping = best_query_pos;
                right_clipping = m - best_query_pos - best_exact_len;
            } else {
                left_match = best_left_len;
                right_match = best_right_len;
                if (best_left_len + best_right_len == m)
                    del_size = (int32_t)(m - ((uint32_t)best_score + no_mismatches));
                else
                    del_size = -(int32_t)(m - left_match - right_match);
                if ((uint32_t)abs(del_size) > 2 * prm.max_approx_indel_len)
                    found = false;
            }
        }

        // ---- getExtCigar (498-589), one output byte per lane and step -------------------
        if (found) {
            // tmp_ref_sequence[i] = min(gp[i + 1], N_ref), tmp_read_sequence[i + 1] = min(query[i], N_read)
            auto tref = [&](uint32_t i) { return min(sym(L.text, skew + i), 5u); };
            auto tread = [&](uint32_t i) { return min(sym(L.query, i), 4u); };
            const uint64_t decode = 0x58584e4e54474341ull;   // "ACGTNNXX"
            auto letter = [&](uint32_t c) { return (uint8_t)(decode >> (8 * c)); };
            uint32_t cigar_len;
            double score;
            if (left_match == 0 && right_match == 0) {
                cigar_len = m;
                for (uint32_t c = lane; c < m; c += 64)
                    ec[c] = (c < left_clipping || c >= m - right_clipping) ? '$' : '.';
                score = (double)(m - left_clipping - right_clipping) * sc.match + ((left_clipping > 0) ? sc.clipping : 0) +
                        ((right_clipping > 0) ? sc.clipping : 0);
            } else {
                const uint32_t dels = del_size > 0 ? (uint32_t)del_size : 0u, ins = del_size < 0 ? (uint32_t)-del_size : 0u;
                const uint32_t mid = 2 * (dels + ins);
                cigar_len = left_match + mid + right_match;
                for (uint32_t c = lane; c < cigar_len; c += 64) {
                    uint8_t ch;
                    if (c < left_match) {
                        ch = (!no_mismatches || tref(pos + c) == tread(c)) ? '.' : letter(tref(pos + c));
                    } else if (c < left_match + mid) {
                        const uint32_t k = (c - left_match) >> 1;
                        if ((c - left_match) & 1)
                            ch = dels ? letter(tref(pos + left_match + k)) : letter(tread(left_match + k));
                        else
                            ch = dels ? '^' : '#';
                    } else {
                        const uint32_t i = c - left_match - mid;
                        const uint32_t rp = pos + left_match + dels + i, qp = left_match + ins + i;
                        ch = (!no_mismatches || tref(rp) == tread(qp)) ? '.' : letter(tref(rp));
                    }
                    ec[c] = ch;
                }
                score = (double)((int64_t)left_match + right_match - no_mismatches) * sc.match + no_mismatches * sc.mismatch;
                if (del_size > 0)
                    score += sc.gap_del_open + (double)((int64_t)dels - 1) * sc.gap_del_extend;
                else if (del_size < 0)
                    score += sc.gap_ins_open + (double)((int64_t)ins - 1) * sc.gap_ins_extend;
            }
            res.found = 1;
            res.pos = pos;
            res.left_clipping = left_clipping;
            res.right_clipping = right_clipping;
            res.left_match = left_match;
            res.right_match = right_match;
            res.del_size = del_size;
            res.no_mismatches = no_mismatches;
            res.cigar_len = cigar_len;
            res.affine_score = score;
            if (lane == 0)
                ec[cigar_len] = 0;
        } else if (lane == 0) {
            ec[0] = 0;
        }
        if (lane == 0)
            results[task] = res;

        // ---- HashTable::reset through the log (soft_clipping.h:45-49) --------------------
        __syncthreads();
        for (uint32_t i = HASH_LEN - 1 + lane; i < m; i += 64)
            L.table[L.cell[i]] = (CELL)EMPTY;
        __syncthreads();
    }
}

uint32_t softclip_table_size(uint32_t max_query_len)   // soft_clipping.cpp:45-48
{
    uint64_t s = 256;
    while (s / max_query_len < 50)
        s *= 2;
    return s > 0xffffffffull ? 0u : (uint32_t)s;
}

static uint64_t softclip_lds_bytes(uint32_t ht_size, uint32_t cell_bytes, uint32_t max_window, uint32_t max_len)
{
    const uint64_t text_dw = (max_window + max_len + 16 + 7) / 8 + 2, query_dw = (max_len + 7) / 8 + 2;
    const uint64_t right_dw = (max_window + 32) / 32;
    return 4 * (text_dw + query_dw + right_dw) + (uint64_t)ht_size * cell_bytes +
           2 * ((uint64_t)((max_len + 1) & ~1u) + max_window + 4);
}

int launch_softclip_match(wamd_ctx* ctx, const wamd_reads* r, const wamd_softclip_task* d_tasks, uint32_t n,
                          const wamd_softclip_params* prm, uint32_t max_window, wamd_softclip_result* d_results,
                          uint8_t* d_cigars, uint32_t cigar_stride, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0)
        return WAMD_OK;
    const uint32_t ht_size = softclip_table_size(prm->max_query_len);
    if (max_window > ctx->max_text_len)
        max_window = ctx->max_text_len;
    const bool bytes = r->max_len >= 20 && r->max_len - 20 < 255;   // every read position fits a byte cell
    const uint64_t lds = softclip_lds_bytes(ht_size, bytes ? 1 : 2, max_window, r->max_len);
    if (ht_size == 0 || ht_size > 32768 || lds > 160 * 1024) {
        set_error("wamd_softclip_match: max_query_len %u needs a %u-cell table (limit 32768 cells = max_query_len 655)",
                  prm->max_query_len, ht_size);
        return WAMD_ERR_TOO_LONG;
    }
    if (lds > 64 * 1024)
        WAMD_HIP(hipFuncSetAttribute(bytes ? (const void*)softclip_match_kernel<uint8_t>
                                           : (const void*)softclip_match_kernel<uint16_t>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint32_t per_cu = (uint32_t)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu);
    const uint32_t blocks = n < 256u * per_cu ? n : 256u * per_cu;
    if (bytes)
        hipLaunchKernelGGL(softclip_match_kernel<uint8_t>, dim3(blocks), dim3(64), (uint32_t)lds, s, genome_view(ctx),
                           r->view(), d_tasks, n, *prm, ctx->scoring, ht_size, max_window, r->max_len,
                           ctx->max_text_len, d_results, d_cigars, cigar_stride, ctx->d_flags);
    else
        hipLaunchKernelGGL(softclip_match_kernel<uint16_t>, dim3(blocks), dim3(64), (uint32_t)lds, s, genome_view(ctx),
                           r->view(), d_tasks, n, *prm, ctx->scoring, ht_size, max_window, r->max_len,
                           ctx->max_text_len, d_results, d_cigars, cigar_stride, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

}   // namespace wamd
