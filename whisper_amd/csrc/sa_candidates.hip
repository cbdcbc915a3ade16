// sa_candidates.hip - the compact submission form of the suffix-array loops (mapping_core_sa.cpp:193-348, 355-513).
//
// Inside one stage the loops hold everything fixed but the read and the suffix-array entry: k, the anchor segment,
// min_read_len, the verifier.  What a caller must send per candidate is therefore its 4-byte suffix-array position;
// per read, how many entries it has on either strand.  The tasks the verifier kernels take (a CF pre-filter task and
// a dynamicProgramming or LevDiag task, 36 bytes a candidate) are written on the device, where they are read:
//   counts -> first candidate of every read (block sums, one-block scan, in-block scan)
//   expand    a wave serves 64 reads; lane l of a round takes the wave's candidate 64 * round + l and finds its read
//             among the wave's 64 prefix sums by six shuffles
//   verify    the kernels of wamd_filter_verify_* / wamd_verify_distance / wamd_levdiag, unchanged; the distance
//             verifier in its accepted-only form: the loop asks nothing of a rejected candidate (:241, :293)
//   finish    one byte a candidate: min(errors, k + 1); end columns only where asked for
// 4 + 1 bytes cross the link per candidate, plus 4 per read.
#include "wamd_internal.h"

namespace wamd {

struct SaLayout {
    uint64_t first, sums, cf, tasks, results, filter, total;
};

static inline uint64_t sup(uint64_t x) { return (x + 255) / 256 * 256; }

static SaLayout sa_layout(uint32_t n, uint32_t n_reads)
{
    SaLayout l{};
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) {
        const uint64_t at = off;
        off += sup(bytes);
        return at;
    };
    l.first = take(((uint64_t)n_reads + 1) * 4);
    l.sums = take(((uint64_t)(n_reads + 255) / 256 + 2) * 4);
    l.cf = take((uint64_t)n * sizeof(wamd_cf_task));
    l.tasks = take((uint64_t)n * sizeof(wamd_dist_task));
    l.results = take((uint64_t)n * sizeof(wamd_dist_result));
    l.filter = take(filter_verify_workspace_bytes(n));
    l.total = off;
    return l;
}

uint64_t sa_candidates_workspace_bytes(uint32_t n, uint32_t n_reads) { return sa_layout(n, n_reads).total; }

__global__ __launch_bounds__(256) void sa_block_sums(uint32_t n_reads, const uint16_t* __restrict__ counts,
                                                     uint32_t* __restrict__ sums)
{
    __shared__ uint32_t w[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t x = 0;
    if (i < n_reads) {
        const uint32_t c = reinterpret_cast<const uint32_t*>(counts)[i];
        x = (c & 0xffffu) + (c >> 16);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        x += __shfl_xor((int)x, o);
    if ((threadIdx.x & 63) == 0)
        w[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0)
        sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// exclusive scan of the block sums in one block; a total that is not the caller's n is a malformed batch
__global__ __launch_bounds__(1024) void sa_scan_sums(uint32_t nb, uint32_t* __restrict__ sums, uint32_t n_expected,
                                                     uint32_t* __restrict__ flags)
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
    if (t == 1023) {
        sums[nb] = part[1023];     // ok[0]: the expansion writes nothing when the total is wrong
        if (part[1023] != n_expected)
            raise_flag(flags, FLAG_ARG);
    }
}

struct SaParams {
    wamd_sa_batch b;
    uint32_t ref_size;
    uint32_t n;
};

template <bool LEVDIAG>
__global__ __launch_bounds__(256) void sa_expand_kernel(ReadsView rv, SaParams p, const uint16_t* __restrict__ counts,
                                                        const uint8_t* __restrict__ read_k,
                                                        const uint32_t* __restrict__ sa_pos,
                                                        const uint32_t* __restrict__ sums, uint32_t* __restrict__ first,
                                                        wamd_cf_task* __restrict__ cf, void* __restrict__ tasks,
                                                        uint32_t* __restrict__ flags)
{
    __shared__ uint32_t wave_total[4];
    const uint32_t nb = gridDim.x;
    if (sums[nb] != p.n)
        return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;   // read of this thread, relative to read_first
    uint32_t n_dir = 0, n_rc = 0;
    if (i < p.b.n_reads) {
        const uint32_t c = reinterpret_cast<const uint32_t*>(counts)[i];
        n_dir = c & 0xffffu;
        n_rc = c >> 16;
    }
    uint32_t incl = n_dir + n_rc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= (uint32_t)d)
            incl += o;
    }
    if (lane == 63)
        wave_total[wave] = incl;
    __syncthreads();
    uint32_t base = sums[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w)
        base += wave_total[w];
    const uint32_t total = wave_total[wave];
    const uint32_t my_first = base + incl - (n_dir + n_rc);
    if (i < p.b.n_reads)
        first[i] = my_first;
    if (i + 1 == p.b.n_reads)
        first[i + 1] = base + incl;
    const uint32_t read = p.b.read_first + i;
    uint32_t m = 0;
    if (i < p.b.n_reads && n_dir + n_rc) {
        if (read >= rv.n)
            raise_flag(flags, FLAG_READ_IDX);
        else if ((m = rv.lens[read]) == 0)
            raise_flag(flags, FLAG_ARG);
    }
    const uint32_t fl = p.b.fixed_left, fr = p.b.fixed_right, mrl = p.b.min_read_len;
    // max_mismatches of this thread's read: the stage's, or the read's own (mapping_mode second: min(stage, best_error + 1))
    const uint32_t my_k = (read_k && i < p.b.n_reads) ? read_k[i] : p.b.max_mismatches;
    for (uint32_t c0 = 0; c0 < total; c0 += 64) {
        const uint32_t j = c0 + lane;   // candidate of this lane inside the wave
        // its read: the first lane whose inclusive prefix exceeds j
        uint32_t lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const uint32_t probe = (uint32_t)__shfl((int)incl, (int)(lo + step - 1));
            lo += probe <= j ? (uint32_t)step : 0u;
        }
        const uint32_t src = lo < 64u ? lo : 63u;
        const uint32_t r_first = (uint32_t)__shfl((int)(incl - (n_dir + n_rc)), (int)src);
        const uint32_t r_dir = (uint32_t)__shfl((int)n_dir, (int)src);
        const uint32_t r_m = (uint32_t)__shfl((int)m, (int)src);
        const uint32_t r_read = (uint32_t)__shfl((int)read, (int)src);
        const uint32_t k = (uint32_t)__shfl((int)my_k, (int)src);
        if (j >= total || r_m == 0)
            continue;
        const uint32_t g = base + j;
        const bool direct = j - r_first < r_dir;
        const uint32_t sp = sa_pos[g];
        wamd_cf_task c{};
        c.read_idx = r_read;
        c.position = sp;
        c.pattern_len = (uint16_t)((mrl == 0 || r_m < mrl) ? 0u : mrl);   // 0: a short read, no filter
        c.read_offset = (uint16_t)(direct || r_m < mrl ? 0u : r_m - mrl);
        c.fixed_left = (uint16_t)(direct ? fl : mrl - fr);
        c.fixed_right = (uint16_t)(direct ? fr : mrl - fl);
        c.dir_gen_flag = direct;
        c.flags = direct ? 0 : WAMD_TASK_FLAG_REV_COMP;
        cf[g] = c;
        if constexpr (LEVDIAG) {
            wamd_levdiag_task t{};   // mapping_core_sa.cpp:239 / :291 (the other strand verifies the reversed read)
            t.read_idx = direct ? r_read : r_read + p.b.reverse_read_offset;
            t.position_in_text = sp;
            t.fixed_left = (uint16_t)(direct ? fl : r_m - fr);
            t.fixed_right = (uint16_t)(direct ? fr : r_m - fl);
            t.max_mismatches = (uint16_t)k;
            t.dir_gen_flag = direct;
            reinterpret_cast<wamd_levdiag_task*>(tasks)[g] = t;
        } else {
            wamd_dist_task t{};      // mapping_core_sa.cpp:237 / :289
            t.read_idx = r_read;
            t.ref_pos = direct ? sp - fl - k : p.ref_size - sp - r_m + fl - k;
            t.max_distance_in_ref = r_m + 2 * k;
            t.max_ed = (uint16_t)k;
            t.variant = WAMD_VARIANT_AUTO;
            t.flags = direct ? 0 : WAMD_TASK_FLAG_REV_COMP;
            reinterpret_cast<wamd_dist_task*>(tasks)[g] = t;
        }
    }
}

template <bool LEVDIAG>
__global__ __launch_bounds__(256) void sa_finish_kernel(uint32_t n, const void* __restrict__ tasks,
                                                        const void* __restrict__ results, uint8_t* __restrict__ errors,
                                                        uint32_t* __restrict__ end_pos)
{
    // four candidates a thread: their bytes leave as one dword
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g0 = 4 * q;
    if (g0 >= n)
        return;
    uint32_t packed = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t g = g0 + u;
        if (g >= n)
            break;
        uint32_t e, pos = 0, k;     // k: the candidate's own (its read's) max_mismatches, as its task holds it
        if constexpr (LEVDIAG) {
            e = reinterpret_cast<const uint32_t*>(results)[g];
            k = reinterpret_cast<const wamd_levdiag_task*>(tasks)[g].max_mismatches;
        } else {
            const wamd_dist_result r = reinterpret_cast<const wamd_dist_result*>(results)[g];
            e = r.edit_distance;
            pos = r.pos;
            k = reinterpret_cast<const wamd_dist_task*>(tasks)[g].max_ed;
        }
        const bool ok = e <= k;     // (RESULT_ERROR is above every k)
        packed |= (ok ? e : k + 1) << (8 * u);
        if (end_pos)
            end_pos[g] = ok ? pos : 0u;
    }
    if (g0 + 4 <= n) {
        reinterpret_cast<uint32_t*>(errors)[q] = packed;
    } else {
        for (uint32_t u = 0; g0 + u < n; ++u)
            errors[g0 + u] = (uint8_t)(packed >> (8 * u));
    }
}

int launch_sa_candidates(wamd_ctx* ctx, const wamd_reads* r, const wamd_sa_batch* b, const uint16_t* d_counts,
                         const uint8_t* d_read_k, const uint32_t* d_sa_positions, uint32_t n, uint8_t* d_errors, uint32_t* d_end_positions,
                         void* d_ws, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n == 0 || b->n_reads == 0)
        return WAMD_OK;
    const SaLayout l = sa_layout(n, b->n_reads);
    char* ws = (char*)d_ws;
    auto* first = (uint32_t*)(ws + l.first);
    auto* sums = (uint32_t*)(ws + l.sums);
    auto* cf = (wamd_cf_task*)(ws + l.cf);
    void* tasks = ws + l.tasks;
    void* results = ws + l.results;
    const uint32_t nb = (b->n_reads + 255) / 256;
    SaParams p;
    p.b = *b;
    p.ref_size = ctx->ref_size;
    p.n = n;
    const bool lev = b->verifier == WAMD_SA_VERIFIER_LEVDIAG;
    hipLaunchKernelGGL(sa_block_sums, dim3(nb), dim3(256), 0, s, b->n_reads, d_counts, sums);
    hipLaunchKernelGGL(sa_scan_sums, dim3(1), dim3(1024), 0, s, nb, sums, n, ctx->d_flags);
    if (lev)
        hipLaunchKernelGGL(sa_expand_kernel<true>, dim3(nb), dim3(256), 0, s, r->view(), p, d_counts, d_read_k, d_sa_positions, sums,
                           first, cf, tasks, ctx->d_flags);
    else
        hipLaunchKernelGGL(sa_expand_kernel<false>, dim3(nb), dim3(256), 0, s, r->view(), p, d_counts, d_read_k, d_sa_positions, sums,
                           first, cf, tasks, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    int rc;
    if (lev) {
        // (a candidate whose task was not written - malformed batch, read out of range - holds whatever the workspace
        //  held: the flag raised above fails the call)
        rc = b->min_read_len ? launch_filter_levdiag(ctx, r, cf, (const wamd_levdiag_task*)tasks, n, (uint32_t*)results, ws + l.filter, s)
                             : launch_levdiag(ctx, r, (const wamd_levdiag_task*)tasks, n, (uint32_t*)results, s);
    } else {
        rc = b->min_read_len ? launch_filter_distance(ctx, r, cf, (const wamd_dist_task*)tasks, n, WAMD_LAST_SYMBOL_FROM_READ,
                                                      (wamd_dist_result*)results, ws + l.filter, s, WAMD_DISTANCE_ACCEPTED_ONLY)
                             : launch_distance(ctx, r, (const wamd_dist_task*)tasks, n, WAMD_LAST_SYMBOL_FROM_READ,
                                               (wamd_dist_result*)results, WAMD_DISTANCE_ACCEPTED_ONLY, s);
    }
    if (rc)
        return rc;
    const uint32_t fb = ((n + 3) / 4 + 255) / 256;
    if (lev)
        hipLaunchKernelGGL(sa_finish_kernel<true>, dim3(fb), dim3(256), 0, s, n, tasks, results, d_errors,
                           (uint32_t*)nullptr);
    else
        hipLaunchKernelGGL(sa_finish_kernel<false>, dim3(fb), dim3(256), 0, s, n, tasks, results, d_errors,
                           d_end_positions);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

} // namespace wamd
