// result_groups.hip - from verified candidates to the bytes a result group holds, on the device.
//
//   commit_kernel        what the SA loop does with a candidate once its verifiers have answered
//                        (mapper/mapping_core_sa.cpp:241-247 / 293-299, 258-272 / 310-326, 334-344): the lev
//                        mapping when cur_match <= max_mismatches, Match's mapping with its strand fix-up,
//                        candidate_mapping_t::better of the two (results.cpp:376-387), kept when its penalty
//                        is within max_mismatches.  One candidate per lane.
//   group_order_kernel   per read, what happens between the match_positions vectors and the collector
//                        (mapping_core.cpp:976-997): every kept mapping is pushed into CMappingsHeapGatherer
//                        (a max-heap on (penalty, pos) of at most max_no_mappings entries, results.cpp:409-425)
//                        and popped "unsorted", i.e. from the back of the heap's array (results.cpp:449-452).
//                        That order is part of the byte stream, so std::push_heap / std::pop_heap are replayed
//                        step for step (libstdc++'s sift-up / __adjust_heap).  One read per lane; the heap is an
//                        array of candidate indices in the caller's workspace.
//   group_write_kernel   CMappingResultsCollector::Push (results.cpp:546-598): per read its id (id_bytes,
//                        big-endian), the number of mappings (mapping_counter_size bytes, big-endian) and the
//                        serialized records (results.cpp:161-217), reads in bin order.  Offsets come from an
//                        exclusive scan of the per-read sizes.
// The result is what rank 0 would find in a group buffer had one thread pushed the bin's reads in order:
// final bytes, ready for the gather.
#include "wamd_internal.h"

namespace wamd {

__device__ __forceinline__ uint8_t rg_indel_encode(int32_t x) { return x > 0 ? (uint8_t)x : (uint8_t)(-x + 128); }

// candidate_mapping_t::serialize, results.cpp:161-217; returns the record size (5 .. 11)
__device__ uint32_t rg_serialize(const wamd_mapping& m, bool direct, uint8_t* b)
{
    uint32_t k = 0;
    b[k++] = (uint8_t)(m.pos >> 24);
    b[k++] = (uint8_t)(m.pos >> 16);
    b[k++] = (uint8_t)(m.pos >> 8);
    b[k++] = (uint8_t)m.pos;
    b[k++] = (uint8_t)((direct ? 0x80u : 0) + (m.type == WAMD_MAP_LEV ? 0x40u : 0) + (m.no_mismatches & 0x3f));
    if (m.type == WAMD_MAP_LEV)
        return k;
    b[k++] = (uint8_t)m.type;
    const uint8_t hi1 = (uint8_t)((m.matching_len1 >> 8) & 0xf);
    switch (m.type) {
    case WAMD_MAP_INDEL1:
        b[k++] = hi1;
        b[k++] = (uint8_t)(m.matching_len1 & 0xff);
        b[k++] = rg_indel_encode(m.indel_len1);
        break;
    case WAMD_MAP_INDEL2:
        b[k++] = (uint8_t)(hi1 + ((m.matching_len2 >> 4) & 0xf0u));
        b[k++] = (uint8_t)(m.matching_len1 & 0xffu);
        b[k++] = (uint8_t)(m.matching_len2 & 0xffu);
        b[k++] = rg_indel_encode(m.indel_len1);
        b[k++] = rg_indel_encode(m.indel_len2);
        break;
    case WAMD_MAP_INDEL_CLIPPING:
    case WAMD_MAP_CLIPPING_INDEL:
        b[k++] = (uint8_t)(hi1 + ((m.clipping_len >> 4) & 0xf0u));
        b[k++] = (uint8_t)(m.matching_len1 & 0xffu);
        b[k++] = (uint8_t)(m.clipping_len & 0xffu);
        b[k++] = rg_indel_encode(m.indel_len1);
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
    return k;
}

__device__ __forceinline__ uint32_t rg_record_size(uint32_t type)   // check_rec_size, results.cpp:278-297
{
    switch (type) {
    case WAMD_MAP_LEV: return 5;
    case WAMD_MAP_INDEL1: return 9;
    case WAMD_MAP_INDEL2: return 11;
    case WAMD_MAP_INDEL_CLIPPING:
    case WAMD_MAP_CLIPPING_INDEL: return 10;
    case WAMD_MAP_MISMATCHES_CLIPPING:
    case WAMD_MAP_CLIPPING_MISMATCHES: return 8;
    case WAMD_MAP_CLIPPING_CLIPPING: return 9;
    default: return 0;
    }
}

__global__ __launch_bounds__(256) void commit_kernel(uint32_t n, const wamd_commit_task* __restrict__ tasks,
                                                     const wamd_mapping* __restrict__ indel,
                                                     wamd_mapping* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const wamd_commit_task t = tasks[i];
    wamd_mapping cm{};   // candidate_mapping_t(), types.h:250-261
    cm.type = WAMD_MAP_NONE;
    if (t.lev_errors <= t.max_mismatches) {   // mapping_core_sa.cpp:241-243 / 293-298
        cm.type = WAMD_MAP_LEV;
        cm.pos = t.lev_pos;
        cm.no_mismatches = t.lev_errors;
        cm.penalty = t.lev_errors;
    }
    if (t.has_indel && indel && indel[i].found && indel[i].type != RESULT_ERROR) {   // :264-267 / :317-321
        wamd_mapping y = indel[i];
        if (!t.direct)
            y.pos += t.pattern_len - 1u;
        // better(): x unless it is none, or y is strictly smaller on (penalty, pos) - results.cpp:376-387
        const bool x_lt_y = cm.penalty != y.penalty ? cm.penalty < y.penalty : cm.pos < y.pos;
        if (cm.type == WAMD_MAP_NONE || !x_lt_y)
            cm = y;
    }
    if (cm.type != WAMD_MAP_NONE && cm.penalty > t.max_mismatches) {   // :338: nothing of it is pushed
        cm = wamd_mapping{};
        cm.type = WAMD_MAP_NONE;
    }
    cm.found = cm.type != WAMD_MAP_NONE;
    out[i] = cm;
}

// a < b on (penalty, pos): candidate_mapping_t::operator<, results.cpp:367-373
__device__ __forceinline__ bool rg_less(const wamd_mapping* m, uint32_t a, uint32_t b)
{
    const uint32_t pa = m[a].penalty, pb = m[b].penalty;
    return pa != pb ? pa < pb : m[a].pos < m[b].pos;
}

__device__ void rg_push_heap(uint32_t* h, const wamd_mapping* m, uint32_t hole, uint32_t top, uint32_t value)
{
    while (hole > top) {   // std::__push_heap
        const uint32_t parent = (hole - 1) / 2;
        if (!rg_less(m, h[parent], value))
            break;
        h[hole] = h[parent];
        hole = parent;
    }
    h[hole] = value;
}

__global__ __launch_bounds__(64) void group_order_kernel(uint32_t n_reads, const uint32_t* __restrict__ read_first,
                                                         const wamd_mapping* __restrict__ m, uint32_t max_no_mappings,
                                                         uint32_t header_bytes, uint32_t* __restrict__ heap,
                                                         uint32_t* __restrict__ counts, uint32_t* __restrict__ sizes)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads)
        return;
    const uint32_t lo = read_first[r], hi = read_first[r + 1];
    uint32_t* h = heap + lo;   // at most hi - lo entries
    uint32_t size = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        if (!m[i].found)
            continue;
        if (size < max_no_mappings) {   // results.cpp:411-415
            ++size;
            rg_push_heap(h, m, size - 1, 0, i);
        } else if (rg_less(m, i, h[0])) {   // :418-423: pop_heap over the whole array, replace the back, push_heap
            const uint32_t len = size - 1;
            const uint32_t value = h[size - 1];
            h[size - 1] = h[0];
            uint32_t hole = 0, child = 0;   // std::__adjust_heap(first, 0, len, value)
            while (child < (len - 1) / 2 && len > 0) {
                child = 2 * (child + 1);
                if (rg_less(m, h[child], h[child - 1]))
                    --child;
                h[hole] = h[child];
                hole = child;
            }
            if (len > 0 && (len & 1u) == 0 && child == (len - 2) / 2) {
                child = 2 * (child + 1);
                h[hole] = h[child - 1];
                hole = child - 1;
            }
            if (len > 0)
                rg_push_heap(h, m, hole, 0, value);
            h[size - 1] = i;
            rg_push_heap(h, m, size - 1, 0, i);
        }
    }
    uint32_t bytes = 0;
    for (uint32_t j = 0; j < size; ++j)
        bytes += rg_record_size(m[h[j]].type);
    counts[r] = size;
    sizes[r] = size ? header_bytes + bytes : 0;   // a read without mappings leaves nothing in the group
}

// exclusive scan of n words in three launches (block sums, one-block scan of those, spread)
__global__ __launch_bounds__(256) void rg_block_sums(uint32_t n, const uint32_t* __restrict__ v, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t w[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t x = i < n ? v[i] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        x += __shfl_xor((int)x, o);
    if ((threadIdx.x & 63) == 0)
        w[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0)
        sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(1024) void rg_scan_sums(uint32_t nb, uint32_t* __restrict__ sums, uint32_t* __restrict__ total)
{
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(nb, lo + per);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i)
        s += sums[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = sums[i];
        sums[i] = run;
        run += c;
    }
    if (t == 1023)
        *total = part[1023];
}

__global__ __launch_bounds__(256) void group_write_kernel(uint32_t n_reads, const uint32_t* __restrict__ read_first,
                                                          const uint64_t* __restrict__ read_ids,
                                                          const wamd_commit_task* __restrict__ tasks,
                                                          const wamd_mapping* __restrict__ m,
                                                          const uint32_t* __restrict__ heap,
                                                          const uint32_t* __restrict__ counts,
                                                          const uint32_t* __restrict__ sizes,
                                                          const uint32_t* __restrict__ block_off, uint32_t id_bytes,
                                                          uint32_t counter_bytes, uint8_t* __restrict__ out,
                                                          uint32_t out_capacity, uint32_t* __restrict__ read_off,
                                                          uint32_t* __restrict__ flags)
{
    __shared__ uint32_t wsum[4];
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t mine = r < n_reads ? sizes[r] : 0;
    uint32_t incl = mine;   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
        if ((int)lane >= o)
            incl += v;
    }
    if (lane == 63)
        wsum[wave] = incl;
    __syncthreads();
    uint32_t off = block_off[blockIdx.x] + incl - mine;
    for (uint32_t w = 0; w < wave; ++w)
        off += wsum[w];
    if (r >= n_reads)
        return;
    if (read_off)
        read_off[r] = off;
    if (!mine)
        return;
    if ((uint64_t)off + mine > out_capacity) {
        raise_flag(flags, FLAG_CIGAR);   // the group does not fit the caller's buffer
        return;
    }
    uint8_t* p = out + off;
    const uint64_t id = read_ids[r];
    for (uint32_t b = 0; b < id_bytes; ++b)   // StoreUInt: big-endian, common/utils.cpp:431-453
        *p++ = (uint8_t)(id >> (8 * (id_bytes - 1 - b)));
    const uint32_t cnt = counts[r];
    for (uint32_t b = 0; b < counter_bytes; ++b)
        *p++ = (uint8_t)(b + 4 >= counter_bytes ? cnt >> (8 * (counter_bytes - 1 - b)) : 0);
    const uint32_t* h = heap + read_first[r];
    for (uint32_t j = cnt; j-- > 0;) {   // PopUnsorted: from the back of the heap's array
        const uint32_t i = h[j];
        p += rg_serialize(m[i], tasks[i].direct != 0, p);
    }
}

uint64_t result_groups_workspace_bytes(uint32_t n_tasks, uint32_t n_reads)
{
    return ((uint64_t)n_tasks + 2ull * n_reads + (n_reads + 255) / 256 + 64) * sizeof(uint32_t) + 1024;
}

int launch_commit(wamd_ctx* ctx, uint32_t n, const wamd_commit_task* d_tasks, const wamd_mapping* d_indel,
                  wamd_mapping* d_out, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(commit_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_tasks, d_indel, d_out);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_result_groups(wamd_ctx* ctx, uint32_t n_tasks, const wamd_commit_task* d_tasks, const wamd_mapping* d_committed,
                         uint32_t n_reads, const uint32_t* d_read_first, const uint64_t* d_read_ids,
                         const wamd_group_params* prm, uint8_t* d_out, uint32_t out_capacity, uint32_t* d_out_bytes,
                         uint32_t* d_read_off, void* d_ws, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (prm->id_bytes == 0 || prm->id_bytes > 8 || prm->mapping_counter_size == 0 || prm->mapping_counter_size > 8 ||
        prm->max_no_mappings == 0) {
        set_error("wamd_emit_result_groups_dev: id_bytes / mapping_counter_size must be 1..8, max_no_mappings > 0");
        return WAMD_ERR_INVALID_ARG;
    }
    uint32_t* heap = (uint32_t*)d_ws;
    uint32_t* counts = heap + n_tasks;
    uint32_t* sizes = counts + n_reads;
    uint32_t* sums = sizes + n_reads;
    const uint32_t nb = (n_reads + 255) / 256;
    hipLaunchKernelGGL(group_order_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, n_reads, d_read_first, d_committed,
                       prm->max_no_mappings, prm->id_bytes + prm->mapping_counter_size, heap, counts, sizes);
    hipLaunchKernelGGL(rg_block_sums, dim3(nb), dim3(256), 0, s, n_reads, sizes, sums);
    hipLaunchKernelGGL(rg_scan_sums, dim3(1), dim3(1024), 0, s, nb, sums, d_out_bytes);
    hipLaunchKernelGGL(group_write_kernel, dim3(nb), dim3(256), 0, s, n_reads, d_read_first, d_read_ids, d_tasks,
                       d_committed, heap, counts, sizes, sums, prm->id_bytes, prm->mapping_counter_size, d_out,
                       out_capacity, d_read_off, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

} // namespace wamd
