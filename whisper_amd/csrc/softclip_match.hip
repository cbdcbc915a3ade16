// softclip_match.hip - the hashing mate rescue, four candidate windows per wave.
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
//     the low bits survive the mask (so 20-mers sharing their last ~8 symbols share a probe
//     chain), insertion order and linear probing: a chain is walked in cell order and every
//     entry on it is "a hit", real or not;
//   * the order of hits (text position, then chain order), because best-so-far, the remembered
//     left-boundary run and the checked_pos marks are all updated on the way;
//   * checked_pos_left's habit of suppressing only every other hit of a diagonal.
//
// Mapping to the wave (round 3).  A window's bookkeeping is serial and was wave-uniform code with 63 of 64
// lanes idle; what bounded the kernel was the 8 KB table per window in LDS (15 waves a CU) and the serial
// instruction stream.  Now:
//   * a wave holds FOUR windows, one per 16-lane group.  Group g's lanes look up 16 text positions at once,
//     serve their hits in order with group-uniform state (ballots cut to the group's 16 bits, broadcasts by
//     ds_bpermute), and extend a hit 128 symbols per step.  The four groups run the same instruction stream
//     under their own exec bits, so a serial step costs one issue slot for four windows;
//   * the table is kept as what it is, a sparse set: an occupancy bitmap of its cells (ht_size / 8 bytes), the
//     number of entries in front of every bitmap dword, and the entries' read positions in cell order.  A chain
//     is a run of set bits, its entries sit next to each other in the value array, and linear probing is a
//     count-trailing-ones on the bitmap.  1.5 KB instead of 8 KB at 150 bp - 3 KB of LDS per window, 13 waves =
//     52 windows a CU - and a 65 536-cell table (reads to 1 024 symbols and beyond) fits where 32 768 cells of
//     two bytes did not;
//   * the 16 low bits of the rolling key come from eight nibbles by a byte swap and three shift-or steps (codes
//     above 3, which carry into the next symbol's bits, take the sum).
// Kept from round 2: a batch of entries sits down at once when nobody collides; hits whose chain holds one entry
// that agrees with the text on fewer than min_exact_len symbols are dropped by their own lanes before the serial
// loop; hits of a read's own diagonal that only move the diagonal's mark are retired a stretch at a time.
#include "wamd_internal.h"

namespace wamd {

namespace {

constexpr uint32_t HASH_LEN = 20;          // HashTable::hash_len, soft_clipping.cpp:25
constexpr uint32_t GROUP = 16;             // lanes per window
constexpr uint32_t GROUPS = 64 / GROUP;    // windows per wave
constexpr uint32_t FILL = 4;               // look-up steps in flight

// Byte offsets of one window's arrays inside its slice of LDS (the same on host and device)
struct SliceLayout {
    uint32_t text, query, right, rank, vals, left, cell, queue, bytes;
    uint32_t occ_dw, text_dw, queue_cap;
};

// CELL: a read position; a byte when every read of the batch is short enough (positions <= 254)
template <typename CELL>
struct Slice {
    uint32_t* occ;             // bit per table cell: HashTable::data[cell] != EMPTY_CELL
    uint32_t* text;            // window, nibble-swapped dwords: symbol s of a dword at bits 4s
    uint32_t* query;           // oriented read codes, same packing
    uint32_t* checked_right;   // bit per text position: checked_pos_right[p] == query_len
    CELL* rank;                // per bitmap dword: entries in the cells in front of it
    CELL* vals;                // HashTable::data of the occupied cells, in cell order
    CELL* checked_left;
    uint16_t* cell;            // per read position: the cell its 20-mer was stored in (data_log)
    uint16_t* queue;           // text positions whose hashed cell is taken, waiting to be served
};

__device__ __forceinline__ uint32_t sym(const uint32_t* a, uint32_t i) { return (a[i >> 3] >> (4 * (i & 7))) & 15u; }

// eight symbols i .. i+7 as one dword (symbol i in the low nibble)
__device__ __forceinline__ uint32_t sym8(const uint32_t* a, uint32_t i)
{
    return __builtin_amdgcn_alignbit(a[(i >> 3) + 1], a[i >> 3], 4 * (i & 7));
}

// low 16 bits of the rolling key after the last of eight symbols (w: the oldest in nibble 0).
// HashTable::increment adds, it does not or, so a code above 3 carries into the next symbol's bits
// (soft_clipping.h:69-71); bits 16 and up belong to older symbols and never reach a table of <= 65 536 cells
__device__ __forceinline__ uint32_t key16(uint32_t w)
{
    if (w & 0xccccccccu) {
        uint32_t k = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            k += ((w >> (4 * (7 - t))) & 15u) << (2 * t);
        return k & 0xffffu;
    }
    uint32_t x = __builtin_bswap32(w);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);   // nibble k = symbol 7 - k
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    return (x | (x >> 8)) & 0xffffu;
}

__device__ __forceinline__ uint32_t gballot(bool p, uint32_t gbase) { return (uint32_t)(__ballot(p) >> gbase) & 0xffffu; }

__device__ __forceinline__ uint32_t gbcast(uint32_t v, uint32_t gbase, uint32_t src)
{
    return (uint32_t)__shfl((int)v, (int)(gbase + src));
}

__device__ __forceinline__ uint32_t group_incl_scan(int32_t v, uint32_t gl)
{
#pragma unroll
    for (int d = 1; d < (int)GROUP; d <<= 1) {
        const int32_t o = __shfl_up(v, d, GROUP);
        if (gl >= (uint32_t)d)
            v += o;
    }
    return (uint32_t)v;
}

__device__ __forceinline__ uint32_t group_min(uint32_t v)
{
#pragma unroll
    for (int d = GROUP / 2; d > 0; d >>= 1)
        v = min(v, (uint32_t)__shfl_xor((int)v, d, GROUP));
    return v;
}

__host__ __device__ inline uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

}   // namespace

template <typename CELL>
__global__ __launch_bounds__(64) void softclip_match_kernel(GenomeView g, ReadsView rv,
                                                            const wamd_softclip_task* __restrict__ tasks, uint32_t n,
                                                            wamd_softclip_params prm, wamd_scoring sc,
                                                            uint32_t ht_size, uint32_t max_window, uint32_t max_len,
                                                            uint32_t max_text_len, SliceLayout lay, uint32_t groups,
                                                            wamd_softclip_result* __restrict__ results,
                                                            uint8_t* __restrict__ cigars, uint32_t cigar_stride,
                                                            uint32_t* __restrict__ flags, uint32_t* __restrict__ work)
{
    extern __shared__ uint32_t lds_raw[];
    const uint32_t lane = threadIdx.x, gl = lane & (GROUP - 1), gbase = lane & ~(GROUP - 1);
    constexpr uint32_t EMPTY = (uint32_t)(CELL)~(CELL)0;   // EMPTY_CELL / empty_pos_in_query in a cell
    Slice<CELL> L;
    {
        uint8_t* base = reinterpret_cast<uint8_t*>(lds_raw) + (size_t)min(lane / GROUP, groups - 1) * lay.bytes;
        L.occ = reinterpret_cast<uint32_t*>(base);
        L.text = reinterpret_cast<uint32_t*>(base + lay.text);
        L.query = reinterpret_cast<uint32_t*>(base + lay.query);
        L.checked_right = reinterpret_cast<uint32_t*>(base + lay.right);
        L.rank = reinterpret_cast<CELL*>(base + lay.rank);
        L.vals = reinterpret_cast<CELL*>(base + lay.vals);
        L.checked_left = reinterpret_cast<CELL*>(base + lay.left);
        L.cell = reinterpret_cast<uint16_t*>(base + lay.cell);
        L.queue = reinterpret_cast<uint16_t*>(base + lay.queue);
    }
    const uint32_t ht_mask = ht_size - 1, occ_dw = lay.occ_dw;

    if (lane / GROUP < groups)
        for (uint32_t i = gl; i < occ_dw; i += GROUP)
            L.occ[i] = 0;
    __syncthreads();

    // (groups: how many of the wave's four groups hold a window - fewer when long reads in long windows leave no room
    //  for four sets of arrays in LDS; the others sit idle)
    // Windows differ a great deal in what they cost (nothing found / a read's whole diagonal served), so the waves take
    // their next four from a counter (`work`, zeroed by the launcher) instead of a fixed share; without a counter
    // (more streams than the context keeps words for) they stride.
    // (eight steps' worth per visit to the counter: unrelated windows take so little that one address would not keep up)
    constexpr uint32_t GRAB = 8;
    uint32_t grabbed = 0;
    for (uint32_t first = blockIdx.x * groups;; first += work ? groups : gridDim.x * groups) {
        if (work && grabbed == 0) {
            uint32_t f = 0;
            if (lane == 0)
                f = atomicAdd(work, groups * GRAB);
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
            grabbed = GRAB;
        }
        if (work)
            --grabbed;
        if (first >= n)
            break;
        const uint32_t task = first + lane / GROUP;
        bool live = task < n && lane / GROUP < groups;
        const wamd_softclip_task t = tasks[live ? task : first];
        wamd_softclip_result res;
        res.affine_score = 0;
        res.found = res.pos = res.left_clipping = res.right_clipping = res.left_match = res.right_match = 0;
        res.del_size = 0;
        res.no_mismatches = res.cigar_len = res.reserved = 0;
        uint8_t* ec = cigars + (uint64_t)(live ? task : first) * cigar_stride;

        uint32_t m = 0;
        uint32_t W = t.max_distance_in_ref;
        const bool rc = t.flags & WAMD_TASK_FLAG_REV_COMP;
        if (live) {
            uint32_t err = 0;
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
                if (gl == 0) {
                    raise_flag(flags, err);
                    res.found = RESULT_ERROR;
                    results[task] = res;
                    ec[0] = 0;
                }
                live = false;
            }
        }
        if (!live) {   // every loop below is empty for this group
            m = 0;
            W = 0;
        }

        // ---- stage the window, the oriented read, and clear the marks -------------------
        const uint32_t skew = t.ref_pos & 7u;          // gp[j] = text symbol skew + j - 1
        if (live) {
            const uint32_t n_text_dw = (skew + W + m + 16 + 7) / 8;
            for (uint32_t i = gl; i < n_text_dw; i += GROUP)
                L.text[i] = load_genome_dword(g, (t.ref_pos >> 3) + i);
            const uint32_t* codes = reinterpret_cast<const uint32_t*>(rv.codes + rv.code_off[t.read_idx]);
            const uint32_t n_dw = (m + 7) / 8;
            for (uint32_t i = gl; i <= n_dw; i += GROUP) {
                uint32_t w = 0;
                if (!rc) {
                    w = i < n_dw ? nibble_swap(codes[i]) : 0u;
                } else if (i < n_dw) {   // oriented symbol p = rc_code(stored symbol m - 1 - p)
                    const int32_t z0 = (int32_t)m - 8 - 8 * (int32_t)i;   // stored symbols z0 .. z0 + 7, backwards
                    bool slow = z0 < 0;
                    if (!slow) {
                        const uint32_t lo = nibble_swap(codes[z0 >> 3]);
                        const uint32_t hi = (z0 & 7) ? nibble_swap(codes[(z0 >> 3) + 1]) : 0u;
                        uint32_t u = __builtin_amdgcn_alignbit(hi, lo, 4 * (z0 & 7));
                        slow = (u & 0x88888888u) != 0;   // codes 8..15 turn into 7
                        u = __builtin_bswap32(u);
                        u = ((u >> 4) & 0x0f0f0f0fu) | ((u & 0x0f0f0f0fu) << 4);
                        const uint32_t high = (u >> 2) & 0x11111111u;   // codes 4..7 stay
                        w = u ^ ((0x11111111u & ~high) * 3u);           // 0..3 -> 3 - c
                    }
                    if (slow) {
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
                }
                L.query[i] = w;   // nibbles past the read are never compared: every run is bounded by its range
            }
            const uint32_t never = prm.fresh_object ? 0u : 0xffffffffu;
            for (uint32_t i = gl; i <= (W + 1) * sizeof(CELL) / 4; i += GROUP)
                reinterpret_cast<uint32_t*>(L.checked_left)[i] = never;
            for (uint32_t i = gl; i <= W / 32; i += GROUP)
                L.checked_right[i] = 0;
        }
        __syncthreads();

        // ---- preprocess: insert every 20-mer in order, linear probing (139-231) ----------
        // (the wave's DS operations execute in program order, so a probe sees every earlier insert without a
        // barrier.)  A batch of 16 consecutive 20-mers first tries to sit down all at once: every lane sets the
        // bit of its hashed cell atomically and learns whether it was set before.  When nobody finds it set,
        // every cell was empty and no two lanes share one, so linear probing in order puts each entry in its own
        // cell.  Otherwise the lanes that found their bit set probe on, in order, with a count of the trailing ones
        // from their cell upward - the others keep their cells, which is what inserting in order gives unless one
        // that is LATER in the read sits between a prober's cell and the free cell it finds (in order, the prober
        // would have been first there).  Then, and when two lanes share a cell, the batch gives its bits back from
        // the first such lane on and sits down one entry after the other.
        for (uint32_t ib = HASH_LEN - 1; ib < m; ib += GROUP) {
            const uint32_t cnt = min(GROUP, m - ib);
            const bool valid = gl < cnt;
            uint32_t mine = 0, home = 0;
            bool lost = false;
            if (valid) {
                home = mine = (key16(sym8(L.query, ib + gl - 7)) * 0x5555u) & ht_mask;
                const uint32_t bit = 1u << (mine & 31);
                lost = (atomicOr(&L.occ[mine >> 5], bit) & bit) != 0;
            }
            const uint32_t lostm = gballot(lost, gbase);
            if (lostm) {
                auto probe = [&](uint32_t p) {   // the first cell from p on whose bit is clear
                    for (;;) {
                        const uint32_t sh = p & 31;
                        const uint32_t free = ~(L.occ[p >> 5] >> sh);   // bit k: cell p + k is empty (k < 32 - sh)
                        const uint32_t z = free ? (uint32_t)__builtin_ctz(free) : 32u;
                        if (z < 32u - sh)
                            return p + z;
                        p = (p + 32u - sh) & ht_mask;
                    }
                };
                const uint32_t u0 = (uint32_t)__builtin_ctz(lostm);
                bool seated = valid && !lost;
                bool redo = false;
                for (uint32_t todo = lostm; todo;) {
                    const uint32_t v = (uint32_t)__builtin_ctz(todo);
                    todo &= todo - 1;
                    const uint32_t hv = gbcast(home, gbase, v);
                    const uint32_t p = probe(hv);
                    if (gballot(seated && gl > v && ((mine - hv) & ht_mask) < ((p - hv) & ht_mask), gbase)) {
                        redo = true;
                        break;
                    }
                    if (gl == v) {
                        L.occ[p >> 5] |= 1u << (p & 31);
                        mine = p;
                        seated = true;
                    }
                }
                if (redo) {
                    if (seated && gl >= u0)
                        atomicAnd(&L.occ[mine >> 5], ~(1u << (mine & 31)));
                    for (uint32_t u = u0; u < cnt; ++u) {
                        if (gl == u) {
                            mine = probe(home);
                            L.occ[mine >> 5] |= 1u << (mine & 31);
                        }
                    }
                }
            }
            if (valid)
                L.cell[ib + gl] = (uint16_t)mine;
        }
        __syncthreads();

        // entries in front of every bitmap dword, then every entry to its place in cell order
        if (live) {
            const uint32_t chunk = occ_dw / GROUP;
            uint32_t local = 0;
            for (uint32_t k = 0; k < chunk; ++k)
                local += (uint32_t)__popc(L.occ[gl * chunk + k]);
            uint32_t run = group_incl_scan((int32_t)local, gl) - local;
            for (uint32_t k = 0; k < chunk; ++k) {
                L.rank[gl * chunk + k] = (CELL)run;
                run += (uint32_t)__popc(L.occ[gl * chunk + k]);
            }
        }
        __syncthreads();
        for (uint32_t i = HASH_LEN - 1 + gl; i < m; i += GROUP) {
            const uint32_t c = L.cell[i];
            const uint32_t r = (uint32_t)L.rank[c >> 5] + (uint32_t)__popc(L.occ[c >> 5] & ((1u << (c & 31)) - 1u));
            L.vals[r] = (CELL)(i - (HASH_LEN - 1));
        }
        __syncthreads();

        // ---- match (234-495) ------------------------------------------------------------
        // Everything below that is not indexed by gl is the same in the 16 lanes of a group.
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
        auto occupied = [&](uint32_t c) { return (L.occ[c >> 5] >> (c & 31)) & 1u; };

        // Look-up and serving are two passes.  The look-up - 16 text positions of each of the four windows per
        // step - only tests the bitmap and appends the positions whose cell is taken to the window's queue; the
        // queue is then served 16 entries at a time.  Positions that hit nothing (98 in 100 on unrelated text)
        // never reach the serving code, and the four groups enter it together, so its serial steps are issued once
        // for four windows instead of once per window.  A queue that fills up (low-complexity text) is served and
        // the look-up goes on where it stopped: the order of hits is the order of text positions either way.
        uint32_t qn = 0, jb = HASH_LEN;
        while (jb <= W || qn) {
        for (; jb <= W && qn + FILL * GROUP <= lay.queue_cap; jb += FILL * GROUP) {
            uint32_t w8[FILL], ow[FILL], cell[FILL];   // (loads of all four steps in flight together)
#pragma unroll
            for (uint32_t u = 0; u < FILL; ++u)
                w8[u] = sym8(L.text, skew + min(jb + u * GROUP + gl, W) - 8);
#pragma unroll
            for (uint32_t u = 0; u < FILL; ++u) {
                cell[u] = (key16(w8[u]) * 0x5555u) & ht_mask;
                ow[u] = L.occ[cell[u] >> 5];
            }
#pragma unroll
            for (uint32_t u = 0; u < FILL; ++u) {
                const uint32_t j = jb + u * GROUP + gl;
                const bool hit = j <= W && ((ow[u] >> (cell[u] & 31)) & 1u);
                const uint32_t hm = gballot(hit, gbase);
                if (hit)
                    L.queue[qn + (uint32_t)__popc(hm & ((1u << gl) - 1u))] = (uint16_t)j;
                qn += (uint32_t)__popc(hm);
            }
        }
        for (uint32_t e0 = 0; e0 < qn; e0 += GROUP) {
            const bool valid = e0 + gl < qn;
            const uint32_t j = valid ? (uint32_t)L.queue[e0 + gl] : 0u;
            // every lane reads the first two cells of its chain now; serving a hit below then costs no
            // LDS round trip unless the chain is longer
            uint32_t first_cell = 0, first_rank = 0, cell_q0 = EMPTY, cell_q1 = EMPTY;
            if (valid) {
                first_cell = (key16(sym8(L.text, skew + j - 8)) * 0x5555u) & ht_mask;
                const uint32_t ow = L.occ[first_cell >> 5], sh = first_cell & 31;
                first_rank = (uint32_t)L.rank[first_cell >> 5] + (uint32_t)__popc(ow & ((1u << sh) - 1u));
                cell_q0 = L.vals[first_rank];
                const uint32_t c1 = (first_cell + 1) & ht_mask;
                if (sh == 31 ? occupied(c1) : (ow >> (sh + 1)) & 1u)
                    cell_q1 = L.vals[c1 ? first_rank + 1 : 0u];
            }
            // An entry whose read position and this text position agree on fewer than min_exact_len symbols
            // changes nothing when it is served (nothing below happens under that length).  Such entries - the
            // 20-mers that merely share their last eight symbols with a 20-mer of the read, and whatever sits in the
            // cells behind a real hit - are weeded out here, every lane walking its own chain: a chain left with no
            // entry drops out, a chain left with one is served (and retired by the stretch) as that entry alone.
            // Chains of more than four cells or with two entries that do agree are served cell by cell.
            bool dud = false;
            if (cell_q0 != EMPTY && min_exact <= 64u && first_cell + 5 <= ht_mask) {   // (no wrap-around)
                const uint32_t pt = j - (HASH_LEN - 1);
                uint32_t q = cell_q0, alive = 0, alive_q = 0;
                for (uint32_t k = 1;; ++k) {
                    bool agree = m - q >= min_exact;
                    {   // the first 24 symbols with their loads in flight together (the read's buffer and the window
                        // reach that far behind any position a 20-mer starts at), the rest eight at a time
                        const uint32_t d0 = sym8(L.query, q) ^ gp8(pt), d1 = sym8(L.query, q + 8) ^ gp8(pt + 8);
                        uint32_t d2 = sym8(L.query, q + 16) ^ gp8(pt + 16);
                        if (min_exact < 24)
                            d2 &= min_exact <= 16 ? 0u : (1u << (4 * (min_exact - 16))) - 1u;
                        uint32_t d01 = d0 | d1;
                        if (min_exact < 16)
                            d01 = min_exact <= 8 ? d0 & ((min_exact == 8 ? 0u : 1u << (4 * min_exact)) - 1u)
                                                 : d0 | (d1 & ((1u << (4 * (min_exact - 8))) - 1u));
                        agree = agree && (d01 | d2) == 0;
                    }
                    for (uint32_t o = 24; o < min_exact && agree; o += 8) {
                        uint32_t diff = sym8(L.query, q + o) ^ gp8(pt + o);
                        if (min_exact - o < 8)
                            diff &= (1u << (4 * (min_exact - o))) - 1u;
                        agree = diff == 0;
                    }
                    if (agree) {
                        if (!alive)
                            alive_q = q;
                        ++alive;
                    }
                    if (k == 1)
                        q = cell_q1;
                    else
                        q = occupied(first_cell + k) ? (uint32_t)L.vals[first_rank + k] : EMPTY;
                    if (q == EMPTY || k == 4)
                        break;
                }
                if (q == EMPTY) {
                    dud = alive == 0;
                    if (alive == 1) {
                        cell_q0 = alive_q;
                        cell_q1 = EMPTY;
                    }
                }
            }
            uint32_t pending = gballot(cell_q0 != EMPTY && !dud, gbase);
            while (pending) {
                const uint32_t src = (uint32_t)__builtin_ctz(pending);
                // ---- hits that change nothing but the diagonal's mark, a whole stretch at once -------------
                // A read that matches its window produces one hit per text position on one diagonal, all
                // inside the run the first of them found.  Served one by one, such a hit (not the read's
                // first 20-mer, no longer than the best so far, and - when the run ends with the read and a
                // left-boundary run is remembered - neither completing an indel nor due for the approximate
                // search) only moves checked_pos_left of its diagonal, every other hit.  The lanes decide that
                // for their own hit in parallel; the leading stretch of such lanes is retired arithmetically.
                if (run_front <= W) {
                    const uint32_t q = cell_q0;
                    bool plain = ((pending >> gl) & 1u) && cell_q1 == EMPTY && q != 0 && q >= run_q && q < run_end &&
                                 j - (HASH_LEN - 1) - q == run_front;
                    const int32_t len = (int32_t)(run_end - q);
                    const bool to_end = run_end == m && left_exact_len > 0;
                    if (plain && len >= (int32_t)min_exact) {
                        plain = len <= best_score;
                        if (to_end) {
                            const int32_t text_dist = (int32_t)(run_front + run_end) - left_text_pos;
                            plain = plain && left_exact_len + len < (int32_t)m && left_exact_len + len != text_dist;
                        }
                    }
                    const uint32_t cnt = (uint32_t)__builtin_ctz(~(gballot(plain, gbase) >> src));   // <= 16 - src
                    if (cnt) {
                        const uint32_t qa = gbcast(cell_q0, gbase, src);
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
                                const uint32_t cl = cl_val == EMPTY ? (1u << 30) : cl_val;
                                const uint32_t s0 = cl + 1 != qa ? qa : qa + 1;   // the first hit that is served
                                if (s0 <= lim) {
                                    const uint32_t last = s0 + ((lim - s0) & ~1u);
                                    cl_val = last;
                                    if (gl == 0) {
                                        L.checked_left[run_front] = (CELL)last;
                                        if (to_end)   // 445
                                            L.checked_right[run_front >> 5] |= 1u << (run_front & 31);
                                    }
                                }
                            }
                            pending &= ~(((1u << cnt) - 1u) << src);
                            continue;
                        }
                    }
                }
                pending &= pending - 1;
                const uint32_t jh = gbcast(j, gbase, src);
                uint32_t ht_pos, ht_rank, second, pos_in_query;
                {
                    const uint32_t a = gbcast(first_cell | (first_rank << 16), gbase, src);
                    const uint32_t b = gbcast(cell_q0 | (cell_q1 << 16), gbase, src);
                    ht_pos = a & 0xffffu;
                    ht_rank = a >> 16;
                    pos_in_query = b & 0xffffu;
                    second = b >> 16;
                    if (sizeof(CELL) == 1) {   // EMPTY is 0xff in a byte cell
                        pos_in_query &= 0xffu;
                        second &= 0xffu;
                    }
                }
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
                        if ((cl == EMPTY ? (1u << 30) : cl) + 1 != pos_in_query) {
                            // 286-291: common run of query[pos_in_query..] and gp[pos_in_text..]
                            const uint32_t range = m - pos_in_query;
                            len = (int32_t)range;
                            // a hit on the diagonal of the last extension, inside its run, ends where that run ended
                            const bool known = front == run_front && pos_in_query >= run_q && pos_in_query < run_end;
                            if (known)
                                len = (int32_t)(run_end - pos_in_query);
                            for (uint32_t off = 0; !known && off < range; off += 8 * GROUP) {
                                const uint32_t o = off + 8 * gl;
                                uint32_t diff = 0xffffffffu;
                                if (o < range) {
                                    diff = sym8(L.query, pos_in_query + o) ^ gp8(pos_in_text + o);
                                    if (range - o < 8)
                                        diff |= 0xffffffffu << (4 * (range - o));
                                }
                                const uint32_t bad = gballot(diff != 0, gbase);
                                if (bad) {
                                    const uint32_t l0 = (uint32_t)__builtin_ctz(bad);
                                    const uint32_t d0 = gbcast(diff, gbase, l0);
                                    const uint32_t stop = off + 8 * l0 + ((uint32_t)__builtin_ctz(d0) >> 2);
                                    len = (int32_t)min(stop, range);
                                    break;
                                }
                            }
                            if (!known && (len >= (int32_t)min_exact || run_front > W)) {
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
                                        for (uint32_t off = 0; off < N; off += GROUP) {
                                            const uint32_t i = off + gl;
                                            const bool rm = i < N && sym(L.query, i + qs) != gp1(tr + i);
                                            total_rm += (uint32_t)__popc(gballot(rm, gbase));
                                        }
                                        uint32_t best = total_rm << 16;   // i = 0
                                        uint32_t run = total_rm;
                                        for (uint32_t off = 0; off < N; off += GROUP) {
                                            const uint32_t i = off + gl;
                                            int32_t d = 0;
                                            if (i < N)
                                                d = (int32_t)(sym(L.query, i) != gp1(tl + i)) -
                                                    (int32_t)(sym(L.query, i + qs) != gp1(tr + i));
                                            const uint32_t incl = group_incl_scan(d, gl);
                                            const uint32_t f = run + incl;
                                            best = min(best, group_min(i < N ? (f << 16) | (i + 1) : 0xffffffffu));
                                            run += gbcast(incl, gbase, GROUP - 1);
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
                                    if (gl == 0)   // 445
                                        L.checked_right[front >> 5] |= 1u << (front & 31);
                                }
                            }
                        }
                        if (len >= (int32_t)min_exact) {   // 451-455
                            if (gl == 0)
                                L.checked_left[front] = (CELL)pos_in_query;
                            cl_val = pos_in_query;
                        }
                    }
                    if (++ht_pos >= ht_size) {   // 459
                        ht_pos = 0;
                        ht_rank = 0;
                    } else {
                        ++ht_rank;
                    }
                    if (link == 0)
                        pos_in_query = second;
                    else
                        pos_in_query = occupied(ht_pos) ? (uint32_t)L.vals[ht_rank] : EMPTY;
                }
            }
        }
        qn = 0;
        }

        // ---- outcome (463-494) -----------------------------------------------------------
        bool found = live && best_score >= (int32_t)min_exact;
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
                for (uint32_t c = gl; c < m; c += GROUP)
                    ec[c] = (c < left_clipping || c >= m - right_clipping) ? '$' : '.';
                score = (double)(m - left_clipping - right_clipping) * sc.match + ((left_clipping > 0) ? sc.clipping : 0) +
                        ((right_clipping > 0) ? sc.clipping : 0);
            } else {
                const uint32_t dels = del_size > 0 ? (uint32_t)del_size : 0u, ins = del_size < 0 ? (uint32_t)-del_size : 0u;
                const uint32_t mid = 2 * (dels + ins);
                cigar_len = left_match + mid + right_match;
                for (uint32_t c = gl; c < cigar_len; c += GROUP) {
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
            if (gl == 0)
                ec[cigar_len] = 0;
        } else if (live && gl == 0) {
            ec[0] = 0;
        }
        if (live && gl == 0)
            results[task] = res;

        // ---- HashTable::reset (soft_clipping.h:45-49): the bitmap is all there is to clear ------
        __syncthreads();
        if (live)
            for (uint32_t i = gl; i < occ_dw; i += GROUP)
                L.occ[i] = 0;
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

static SliceLayout softclip_layout(uint32_t ht_size, uint32_t cell_bytes, uint32_t max_window, uint32_t max_len)
{
    SliceLayout l;
    l.occ_dw = ht_size / 32;
    l.text_dw = (max_window + max_len + 16 + 7) / 8 + 2 + 1;   // + 1: the look-up's sym8 reads one dword ahead
    const uint32_t query_dw = (max_len + 7) / 8 + 2, right_dw = (max_window + 32) / 32;
    uint32_t o = 4 * l.occ_dw;
    l.text = o;
    o += 4 * l.text_dw;
    l.query = o;
    o += 4 * query_dw;
    l.right = o;
    o += 4 * right_dw;
    l.rank = o;
    o += round_up(l.occ_dw * cell_bytes, 4);
    l.vals = o;
    o += round_up(max_len * cell_bytes, 4);
    l.left = o;
    o += round_up((max_window + 2) * cell_bytes, 4) + 4;   // cleared a dword at a time
    // (the log of cells is through with when the values are in place, before the first position is queued: one space)
    l.cell = o;
    l.queue = o;
    l.queue_cap = round_up(max_len + 16, GROUP) + FILL * GROUP;   // a read's own hits and a few strangers: one pass as a rule
    o += round_up(2 * (l.queue_cap > max_len + 1 ? l.queue_cap : max_len + 1), 4);
    // an odd number of 64-byte lines: the four slices start on different halves of the banks
    o = round_up(o, 64);
    if (!((o / 64) & 1))
        o += 64;
    l.bytes = o;
    return l;
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
    // the table holds max_query_len / 50 of its cells at most; cells and read positions travel as 16 bits
    if (ht_size == 0 || ht_size > 65536 || ht_size < 32 * GROUP) {
        set_error("wamd_softclip_match: max_query_len %u needs a %u-cell table (limit 65536 cells = max_query_len 1310)",
                  prm->max_query_len, ht_size);
        return WAMD_ERR_TOO_LONG;
    }
    const SliceLayout lay = softclip_layout(ht_size, bytes ? 1 : 2, max_window, r->max_len);
    uint32_t groups = GROUPS;
    while (groups > 1 && (uint64_t)groups * lay.bytes > 160 * 1024)
        groups /= 2;
    const uint64_t lds = (uint64_t)groups * lay.bytes;
    if (lds > 160 * 1024) {
        set_error("wamd_softclip_match: a window of %u symbols and a read of %u need %llu bytes of LDS (limit 163840)",
                  max_window, r->max_len, (unsigned long long)lds);
        return WAMD_ERR_TOO_LONG;
    }
    if (lds > 64 * 1024)
        WAMD_HIP(hipFuncSetAttribute(bytes ? (const void*)softclip_match_kernel<uint8_t>
                                           : (const void*)softclip_match_kernel<uint16_t>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent blocks, as many as are resident at once by the runtime's count (one fewer when they have to stride: a
    // block too many per CU would run its share in a second round, alone - and the count has been seen one too high)
    int resident = 0;
    WAMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, bytes ? (const void*)softclip_match_kernel<uint8_t>
                                                                           : (const void*)softclip_match_kernel<uint16_t>,
                                                          64, (size_t)lds));
    uint32_t* work = band_flags(ctx, s);   // the stream's first word (myers_distance.hip): launches on a stream are ordered
    if (work)
        WAMD_HIP(hipMemsetAsync(work, 0, sizeof(uint32_t), s));
    const uint32_t per_cu = resident < 1 ? 1u : (uint32_t)(work || resident < 2 ? resident : resident - 1);
    const uint32_t waves = (n + groups - 1) / groups;
    const uint32_t blocks = waves < 256u * per_cu ? waves : 256u * per_cu;   // 256 CUs
    if (bytes)
        hipLaunchKernelGGL(softclip_match_kernel<uint8_t>, dim3(blocks), dim3(64), (uint32_t)lds, s, genome_view(ctx),
                           r->view(), d_tasks, n, *prm, ctx->scoring, ht_size, max_window, r->max_len,
                           ctx->max_text_len, lay, groups, d_results, d_cigars, cigar_stride, ctx->d_flags, work);
    else
        hipLaunchKernelGGL(softclip_match_kernel<uint16_t>, dim3(blocks), dim3(64), (uint32_t)lds, s, genome_view(ctx),
                           r->view(), d_tasks, n, *prm, ctx->scoring, ht_size, max_window, r->max_len,
                           ctx->max_text_len, lay, groups, d_results, d_cigars, cigar_stride, ctx->d_flags, work);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

}   // namespace wamd
