// ext_cigar_ops.hip - record-shaped passes of the post-processing half of the path:
//   * CIndelMatching::GetExtCigarNew   (mapper/indels.cpp:938-957, 680-935)
//   * candidate_mapping_t::serialize   (mapper/results.cpp:161-217)
//   * accepted candidates -> mapping records (row K)
// One record per lane.  (The clipping passes and extract_cigar_and_mdz live in post_passes.hip.)
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
