// reads_prepare.hip - what LevMyers64::preprocess / preprocessRawSeq do per read
// (mapper/LevMyers64.cpp:167-220), once per uploaded batch:
//   1. canonical 4-bit codes of every read (nibble input is kept, ASCII goes
//      through raw_code: ACGT 0..3, N 4, anything else 6 - LevMyers64.cpp:50-60);
//   2. per read and orientation, the pattern-match bit-vectors for the text
//      symbols A, C, G, T plus a 16-bit presence mask and the last symbol.
#include "wamd_internal.h"

namespace wamd {

// t = q * d + r for a thread index: a 32-bit division whenever the index fits (a 64-bit division
// by a run-time value is a call of some 200 instructions, more than these kernels' own work)
__device__ __forceinline__ void thread_divmod(uint64_t t, uint32_t d, uint64_t& q, uint32_t& r)
{
    if ((t >> 32) == 0) {
        const uint32_t t32 = (uint32_t)t;
        q = t32 / d;
        r = t32 % d;
    } else {
        q = t / d;
        r = (uint32_t)(t % d);
    }
}


// one thread per four output dwords (32 symbols) of the canonical code array
__global__ __launch_bounds__(256) void canonical_codes_kernel(
    const uint8_t* __restrict__ raw, const uint64_t* __restrict__ raw_off,
    const uint32_t* __restrict__ lens, const uint64_t* __restrict__ code_off, uint32_t n,
    uint32_t dwords_per_read, int ascii, uint8_t* __restrict__ codes)
{
    // one thread per four output dwords (32 symbols) of a read
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t quads_per_read = (dwords_per_read + 3) / 4;
    uint64_t read64;
    uint32_t dq;
    thread_divmod(t, quads_per_read, read64, dq);
    if (read64 >= n)
        return;
    const uint32_t read = (uint32_t)read64;
    const uint32_t len = lens[read];
    const uint32_t padded_dw = ((len + 1) / 2 + 3) / 4;
    const uint32_t d0 = 4 * dq;
    if (d0 >= padded_dw)
        return;
    // the thread's input bytes (16 packed or 32 ASCII) from aligned dwords and a byte funnel: reads
    // start at any byte, and byte loads cost an instruction each (the raw buffer has 64 bytes of
    // slack behind the last read)
    const uint64_t at = raw_off[read] + (ascii ? 8ull : 4ull) * d0;
    const uint32_t* adw = reinterpret_cast<const uint32_t*>(raw) + (at >> 2);
    const uint32_t sh = (uint32_t)at & 3u;
    uint32_t in[8];
    {
        uint32_t a[9];
#pragma unroll
        for (int q = 0; q < 9; ++q)
            a[q] = (q < 5 || ascii) ? adw[q] : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            in[q] = __builtin_amdgcn_alignbyte(a[q + 1], a[q], sh);
    }
    uint32_t outv[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t d = d0 + j;
        uint32_t out = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t b = 4 * d + k;   // output byte: symbols 2b (high nibble) and 2b + 1
            uint32_t c0 = 15, c1 = 15;      // padding: code 15
            if (!ascii) {
                if (2 * b < len) {
                    const uint32_t x = (in[j] >> (8 * k)) & 0xffu;
                    c0 = x >> 4;
                    if (2 * b + 1 < len)
                        c1 = x & 15u;
                }
            } else {
                // raw_code, LevMyers64.cpp:50-60: ACGT 0..3, N 4, anything else 6
                auto code = [](uint32_t ch) -> uint32_t {
                    return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : ch == 'N' ? 4u : 6u;
                };
                const uint32_t pair = (in[2 * j + (k >> 1)] >> (16 * (k & 1u))) & 0xffffu;
                if (2 * b < len)
                    c0 = code(pair & 0xffu);
                if (2 * b + 1 < len)
                    c1 = code(pair >> 8);
            }
            out |= ((c0 << 4) | c1) << (8 * k);
        }
        outv[j] = out;
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(codes + code_off[read]) + d0;
    if (d0 + 4 <= padded_dw) {
        const wamd_u32x4 v = {outv[0], outv[1], outv[2], outv[3]};
        *reinterpret_cast<wamd_u32x4_a4*>(dst) = v;
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
            if (d0 + j < padded_dw)
                dst[j] = outv[j];
    }
}

// Eight symbols i0 .. i0+7 of a read in the orientation asked for, symbol k at bits 4k; symbols
// outside the read come back as code 15.  codes: the read's canonical dwords (padded with 15).
__device__ __forceinline__ uint32_t oriented_group8(const uint32_t* codes, uint32_t len, bool rc, int32_t i0)
{
    const int32_t last_dw = (int32_t)((len + 7) / 8) - 1;
    auto dw = [&](int32_t q) -> uint32_t { return (q < 0 || q > last_dw) ? 0xffffffffu : nibble_swap(codes[q]); };
    auto window = [&](int32_t a) -> uint32_t {   // stored symbols a .. a+7
        const int32_t q = a >> 3;                // floor: a may be negative
        const uint32_t sh = (uint32_t)(a & 7) * 4u;
        return __builtin_amdgcn_alignbit(dw(q + 1), dw(q), sh);
    };
    if (!rc)
        return window(i0);
    // oriented symbol i = rc_code(stored[len-1-i]): take stored len-8-i0 .. len-1-i0 and reverse
    uint32_t w = window((int32_t)len - 8 - i0);
    w = __builtin_bswap32(w);
    w = ((w & 0x0f0f0f0fu) << 4) | ((w >> 4) & 0x0f0f0f0fu);
    // rc_code per nibble (LevMyers64.cpp:40-48): c < 4 -> 3 - c, c < 8 -> c, else 7
    const uint32_t m = 0x11111111u;
    const uint32_t ge8 = (w >> 3) & m, ge4 = ((w >> 2) | (w >> 3)) & m;
    const uint32_t lt4 = ~ge4 & m;
    w ^= lt4 * 3u;                       // 3 - c for A, C, G, T
    w = (w & ~(ge8 * 15u)) | (ge8 * 7u);
    return w;
}

// one thread per (read, slot, word)
__global__ __launch_bounds__(256) void pattern_vectors_kernel(ReadsView rv, uint64_t* __restrict__ pm)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t w;
    uint64_t rs;
    thread_divmod(t, rv.words, rs, w);
    const uint32_t slot = rv.n_orient == 2 ? (uint32_t)(rs & 1u) : 0u;
    const uint64_t read64 = rv.n_orient == 2 ? rs >> 1 : rs;
    if (read64 >= rv.n)
        return;
    const uint32_t read = (uint32_t)read64;
    const bool rc = ((int32_t)slot == rv.slot_rc);
    const uint32_t len = rv.lens[read];
    const uint8_t* codes = rv.codes + rv.code_off[read];
    const uint32_t* code_dw = reinterpret_cast<const uint32_t*>(codes);
    uint64_t row[4] = {0, 0, 0, 0};
    const uint32_t lo = w * 64, hi = min(len, lo + 64);
    for (uint32_t g0 = lo; g0 < hi; g0 += 8) {
        const uint32_t syms = oriented_group8(code_dw, len, rc, (int32_t)g0);
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            // LevMyers64.cpp:176-190 / 206-217
            const uint32_t i = g0 + k;
            const uint32_t c = (syms >> (4 * k)) & 15u;
            const uint64_t bit = i < hi ? 1ull << (i - lo) : 0ull;
            row[0] |= c == 0 ? bit : 0;
            row[1] |= c == 1 ? bit : 0;
            row[2] |= c == 2 ? bit : 0;
            row[3] |= c == 3 ? bit : 0;
        }
    }
    uint64_t* dst = pm + ((uint64_t)(read * (uint64_t)rv.n_orient + slot) * 4) * rv.words + w;
    for (int c = 0; c < 4; ++c)
        dst[(uint64_t)c * rv.words] = row[c];
}

// one thread per read: which codes it holds (bits 0..15) and its last symbol (bits 16..19,
// LevMyers64.cpp:182 / 192), for either orientation - plain stores, no atomics
__global__ __launch_bounds__(256) void reads_meta_kernel(ReadsView rv, uint32_t* __restrict__ meta)
{
    const uint32_t read = blockIdx.x * blockDim.x + threadIdx.x;
    if (read >= rv.n)
        return;
    const uint32_t len = rv.lens[read];
    const uint8_t* codes = rv.codes + rv.code_off[read];
    const uint32_t* code_dw = reinterpret_cast<const uint32_t*>(codes);
    const uint32_t nd = (len + 7) / 8;
    uint32_t presence = 0;
    for (uint32_t d4 = 0; d4 < nd; d4 += 4) {   // the codes buffer has 64 bytes of slack behind the last read
        const wamd_u32x4 v = *reinterpret_cast<const wamd_u32x4_a4*>(code_dw + d4);
        const uint32_t b[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            const uint32_t d = d4 + c;
            if (d < nd) {
                const uint32_t w = nibble_swap(b[c]);
                const uint32_t ns = len - 8 * d;   // symbols of the read in this dword (>= 1)
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k)
                    presence |= k < ns ? 1u << ((w >> (4 * k)) & 15u) : 0u;
            }
        }
    }
    if (len == 0)
        return;
    if (rv.slot_direct >= 0)
        meta[read * (uint64_t)rv.n_orient + (uint32_t)rv.slot_direct] = presence | (packed_symbol(codes, len - 1) << 16);
    if (rv.slot_rc >= 0) {
        // rc_code (LevMyers64.cpp:40-48): A <-> T, C <-> G, 4..7 stay, 8..15 -> 7
        const uint32_t p = ((presence & 1u) << 3) | ((presence & 2u) << 1) | ((presence & 4u) >> 1) | ((presence & 8u) >> 3) |
                           (presence & 0xf0u) | ((presence & 0xff00u) ? 0x80u : 0u);
        meta[read * (uint64_t)rv.n_orient + (uint32_t)rv.slot_rc] = p | (rc_code(packed_symbol(codes, 0)) << 16);
    }
}

// one thread per (read, slot, dword) of the top-aligned 32-bit vectors
__global__ __launch_bounds__(256) void pattern_vectors32_kernel(ReadsView rv, uint32_t* __restrict__ pm32)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t j;
    uint64_t rs;
    thread_divmod(t, rv.words32, rs, j);
    const uint32_t slot = rv.n_orient == 2 ? (uint32_t)(rs & 1u) : 0u;
    const uint64_t read64 = rv.n_orient == 2 ? rs >> 1 : rs;
    if (read64 >= rv.n)
        return;
    const uint32_t read = (uint32_t)read64;
    const bool rc = ((int32_t)slot == rv.slot_rc);
    const uint32_t len = rv.lens[read];
    const uint8_t* codes = rv.codes + rv.code_off[read];
    const int32_t phantom = (int32_t)(32 * rv.words32 - len);   // rows below the read's first symbol
    const uint32_t* code_dw = reinterpret_cast<const uint32_t*>(codes);
    uint32_t row[4] = {0, 0, 0, 0};
    const int32_t i0 = (int32_t)(32 * j) - phantom;
    // eight symbols at a time: one flag bit per nibble and class, the eight flags squeezed into a byte
    auto squeeze = [](uint32_t f) -> uint32_t {   // bits 0, 4, .., 28 -> bits 0 .. 7
        f = (f | (f >> 3)) & 0x03030303u;
        f = (f | (f >> 6)) & 0x000f000fu;
        return (f | (f >> 12)) & 0xffu;
    };
    // the 32 oriented symbols i0 .. i0 + 31 are 32 consecutive stored symbols either way (read
    // backwards for the other strand): five dwords, four funnels
    uint32_t grp[4];
    {
        const int32_t last_dw = (int32_t)((len + 7) / 8) - 1;
        const int32_t start = rc ? (int32_t)len - 32 - i0 : i0;   // first stored symbol of the span
        const int32_t q0 = start >> 3;                             // floor: start may be negative
        const uint32_t sh = (uint32_t)(start & 7) * 4u;
        uint32_t D[5];
#pragma unroll
        for (int32_t k = 0; k < 5; ++k) {
            const int32_t q = q0 + k;
            D[k] = (q < 0 || q > last_dw) ? 0xffffffffu : nibble_swap(code_dw[q]);
        }
#pragma unroll
        for (int32_t k = 0; k < 4; ++k) {
            uint32_t w = __builtin_amdgcn_alignbit(D[k + 1], D[k], sh);   // stored symbols start + 8k ..
            if (rc) {
                // oriented symbol i = rc_code(stored[len - 1 - i]): reverse the eight symbols ...
                w = __builtin_bswap32(w);
                w = ((w & 0x0f0f0f0fu) << 4) | ((w >> 4) & 0x0f0f0f0fu);
                // ... and rc_code per nibble (LevMyers64.cpp:40-48): c < 4 -> 3 - c, c < 8 -> c, else 7
                const uint32_t mm = 0x11111111u;
                const uint32_t ge8 = (w >> 3) & mm, ge4 = ((w >> 2) | (w >> 3)) & mm;
                const uint32_t lt4 = ~ge4 & mm;
                w ^= lt4 * 3u;
                w = (w & ~(ge8 * 15u)) | (ge8 * 7u);
            }
            grp[rc ? 3 - k : k] = w;
        }
    }
#pragma unroll
    for (int32_t g0 = 0; g0 < 32; g0 += 8) {
        const uint32_t syms = grp[g0 >> 3];
        const uint32_t m1 = 0x11111111u;
        const uint32_t acgt = ~((syms >> 2) | (syms >> 3)) & m1;   // nibble < 4
        const uint32_t b0 = syms & m1, b1 = (syms >> 1) & m1;
        // a phantom row (i < 0) matches every text symbol: D stays 0 down to the read's row 0
        const int32_t np = -(i0 + g0);   // nibbles of this group that are phantom rows
        const uint32_t ph = np <= 0 ? 0u : (np >= 8 ? m1 : (m1 & ((1u << (4 * np)) - 1u)));
        row[0] |= squeeze((acgt & ~b0 & ~b1) | ph) << g0;
        row[1] |= squeeze((acgt & b0 & ~b1) | ph) << g0;
        row[2] |= squeeze((acgt & ~b0 & b1) | ph) << g0;
        row[3] |= squeeze((acgt & b0 & b1) | ph) << g0;
    }
    // [dword][A,C,G,T]: the verification kernels fetch a dword of all four rows with one 16-byte load
    uint4* dst = reinterpret_cast<uint4*>(pm32) + (uint64_t)(read * (uint64_t)rv.n_orient + slot) * rv.pm32_stride + j;
    *dst = make_uint4(row[0], row[1], row[2], row[3]);
}

// ---- where every read's codes start: an exclusive scan of the reads' code bytes, on the device --------------------
// (a host loop that fills 8 bytes per read and sends them was a third of an upload's time)
__device__ __forceinline__ uint32_t code_bytes_of(uint32_t len) { return ((len + 1) / 2 + 3) / 4 * 4; }

__global__ __launch_bounds__(256) void code_off_sums_kernel(uint32_t n, const uint32_t* __restrict__ lens,
                                                            uint64_t* __restrict__ sums)
{
    __shared__ uint32_t w[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t x = i < n ? code_bytes_of(lens[i]) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        x += __shfl_xor((int)x, o);
    if ((threadIdx.x & 63) == 0)
        w[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0)
        sums[blockIdx.x] = (uint64_t)w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(1024) void code_off_scan_kernel(uint32_t nb, uint64_t* __restrict__ sums)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(nb, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i)
        s += sums[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = t ? part[t - 1] : 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t c = sums[i];
        sums[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void code_off_spread_kernel(uint32_t n, const uint32_t* __restrict__ lens,
                                                              const uint64_t* __restrict__ sums,
                                                              uint64_t* __restrict__ code_off)
{
    __shared__ uint32_t w[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t mine = i < n ? code_bytes_of(lens[i]) : 0;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= (uint32_t)d)
            incl += o;
    }
    if (lane == 63)
        w[wave] = incl;
    __syncthreads();
    uint64_t base = sums[blockIdx.x];
    for (uint32_t q = 0; q < wave; ++q)
        base += w[q];
    if (i < n)
        code_off[i] = base + incl - mine;
    if (i + 1 == n)
        code_off[n] = base + incl;
}

// wamd_reads_refill_async: the caller stated the block's bytes and its longest read and no host pass has looked at
// the reads.  Every read is held against both before any kernel uses it; one that breaks them raises the malformed-task
// flag and becomes a one-symbol read at the start of the block, so that nothing downstream leaves its buffers.
__global__ __launch_bounds__(256) void reads_check_kernel(uint32_t n, uint32_t* __restrict__ lens, uint64_t* __restrict__ raw_off,
                                                          uint32_t max_len, uint64_t data_bytes, int nibble,
                                                          uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t len = lens[i];
    const uint64_t off = raw_off[i];
    const uint64_t bytes = nibble ? ((uint64_t)len + 1) / 2 : len;
    if (len == 0 || len > max_len || off > data_bytes || bytes > data_bytes - off) {
        raise_flag(flags, FLAG_ARG);
        lens[i] = 1;
        raw_off[i] = 0;
    }
}

int launch_reads_prepare(wamd_ctx* ctx, wamd_reads* r, const uint8_t* d_raw,
                         const uint64_t* d_raw_off, hipStream_t s, uint64_t stated_bytes)
{
    if (stated_bytes) {
        hipLaunchKernelGGL(reads_check_kernel, dim3((r->n + 255) / 256), dim3(256), 0, s, r->n, r->d_lens, r->d_raw_off,
                           r->max_len, stated_bytes, r->encoding == WAMD_READS_NIBBLE ? 1 : 0, ctx->d_flags);
        WAMD_HIP(hipGetLastError());
    }
    const uint32_t dwords_per_read = ((r->max_len + 1) / 2 + 3) / 4;
    {   // d_meta (4 bytes per read and orientation) lends its front to the block sums; it is cleared afterwards
        const uint32_t nb = (r->n + 255) / 256;
        uint64_t* sums = reinterpret_cast<uint64_t*>(r->d_meta);
        hipLaunchKernelGGL(code_off_sums_kernel, dim3(nb), dim3(256), 0, s, r->n, r->d_lens, sums);
        hipLaunchKernelGGL(code_off_scan_kernel, dim3(1), dim3(1024), 0, s, nb, sums);
        hipLaunchKernelGGL(code_off_spread_kernel, dim3(nb), dim3(256), 0, s, r->n, r->d_lens, sums, r->d_code_off);
        WAMD_HIP(hipGetLastError());
        const uint32_t n_orient = r->orientations == 3 ? 2 : 1;
        WAMD_HIP(hipMemsetAsync(r->d_meta, 0, (size_t)r->n * n_orient * sizeof(uint32_t), s));
    }
    {
        const uint64_t threads = (uint64_t)r->n * ((dwords_per_read + 3) / 4);
        const uint32_t blocks = (uint32_t)((threads + 255) / 256);
        hipLaunchKernelGGL(canonical_codes_kernel, dim3(blocks), dim3(256), 0, s, d_raw, d_raw_off,
                           r->d_lens, r->d_code_off, r->n, dwords_per_read,
                           r->encoding == WAMD_READS_ASCII ? 1 : 0, r->d_codes);
        WAMD_HIP(hipGetLastError());
    }
    {
        ReadsView rv = r->view();
        hipLaunchKernelGGL(reads_meta_kernel, dim3((r->n + 255) / 256), dim3(256), 0, s, rv, r->d_meta);
        WAMD_HIP(hipGetLastError());
    }
    {
        ReadsView rv = r->view();
        const uint64_t threads = (uint64_t)r->n * rv.n_orient * rv.words32;
        const uint32_t blocks = (uint32_t)((threads + 255) / 256);
        hipLaunchKernelGGL(pattern_vectors32_kernel, dim3(blocks), dim3(256), 0, s, rv, r->d_pm32);
        WAMD_HIP(hipGetLastError());
    }
    return WAMD_OK;
}

// The 64-bit vectors of a batch, built on the stream of the first caller that needs them.
int ensure_pm(const wamd_reads* cr, hipStream_t s)
{
    wamd_reads* r = const_cast<wamd_reads*>(cr);
    if (r->n == 0)
        return WAMD_OK;
    while (__sync_lock_test_and_set(&r->pm_lock, 1)) {
    }
    hipError_t e = hipSuccess;
    if (r->pm_state == 0) {
        ReadsView rv = r->view();
        const uint64_t threads = (uint64_t)r->n * rv.n_orient * rv.words;
        const uint32_t blocks = (uint32_t)((threads + 255) / 256);
        hipLaunchKernelGGL(pattern_vectors_kernel, dim3(blocks), dim3(256), 0, s, rv, r->d_pm);
        e = hipGetLastError();
        if (e == hipSuccess)
            e = hipEventRecord(r->pm_event, s);
        if (e == hipSuccess) {
            r->pm_stream = s;
            r->pm_state = 1;
        }
    } else if (r->pm_stream != s) {
        e = hipStreamWaitEvent(s, r->pm_event, 0);
    }
    __sync_lock_release(&r->pm_lock);
    WAMD_HIP(e);
    return WAMD_OK;
}

} // namespace wamd
