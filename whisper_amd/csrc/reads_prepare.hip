// reads_prepare.hip - what LevMyers64::preprocess / preprocessRawSeq do per read
// (mapper/LevMyers64.cpp:167-220), once per uploaded batch:
//   1. canonical 4-bit codes of every read (nibble input is kept, ASCII goes
//      through raw_code: ACGT 0..3, N 4, anything else 6 - LevMyers64.cpp:50-60);
//   2. per read and orientation, the pattern-match bit-vectors for the text
//      symbols A, C, G, T plus a 16-bit presence mask and the last symbol.
#include "wamd_internal.h"

namespace wamd {

// one thread per output byte (two symbols) of the canonical code array
__global__ __launch_bounds__(256) void canonical_codes_kernel(
    const uint8_t* __restrict__ raw, const uint64_t* __restrict__ raw_off,
    const uint32_t* __restrict__ lens, const uint64_t* __restrict__ code_off, uint32_t n,
    uint32_t bytes_per_read, int ascii, uint8_t* __restrict__ codes)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t read = (uint32_t)(t / bytes_per_read);
    const uint32_t b = (uint32_t)(t % bytes_per_read);
    if (read >= n)
        return;
    const uint32_t len = lens[read];
    const uint32_t padded = ((len + 1) / 2 + 3) / 4 * 4;
    if (b >= padded)
        return;
    uint32_t out = 0xff;   // padding: code 15
    if (b < (len + 1) / 2) {
        const uint8_t* src = raw + raw_off[read];
        if (!ascii) {
            out = src[b];
            if ((len & 1) && b == len / 2)
                out |= 0x0f;
        } else {
            uint32_t c[2] = {15, 15};
            for (int h = 0; h < 2; ++h) {
                const uint32_t i = 2 * b + h;
                if (i < len) {
                    const uint32_t ch = src[i];
                    c[h] = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u
                         : ch == 'N' ? 4u : 6u;
                }
            }
            out = (c[0] << 4) | c[1];
        }
    }
    codes[code_off[read] + b] = (uint8_t)out;
}

// one thread per (read, slot, word)
__global__ __launch_bounds__(256) void pattern_vectors_kernel(ReadsView rv, uint64_t* __restrict__ pm,
                                                              uint32_t* __restrict__ meta)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t w = (uint32_t)(t % rv.words);
    const uint64_t rs = t / rv.words;
    const uint32_t slot = (uint32_t)(rs % rv.n_orient);
    const uint32_t read = (uint32_t)(rs / rv.n_orient);
    if (read >= rv.n)
        return;
    const bool rc = ((int32_t)slot == rv.slot_rc);
    const uint32_t len = rv.lens[read];
    const uint8_t* codes = rv.codes + rv.code_off[read];
    uint64_t row[4] = {0, 0, 0, 0};
    uint32_t presence = 0;
    const uint32_t lo = w * 64, hi = min(len, lo + 64);
    for (uint32_t i = lo; i < hi; ++i) {
        // LevMyers64.cpp:176-190 / 206-217
        const uint32_t c = rc ? rc_code(packed_symbol(codes, len - 1 - i)) : packed_symbol(codes, i);
        const uint64_t bit = 1ull << (i - lo);
        row[0] |= c == 0 ? bit : 0;
        row[1] |= c == 1 ? bit : 0;
        row[2] |= c == 2 ? bit : 0;
        row[3] |= c == 3 ? bit : 0;
        presence |= 1u << c;
    }
    uint64_t* dst = pm + ((uint64_t)(read * (uint64_t)rv.n_orient + slot) * 4) * rv.words + w;
    for (int c = 0; c < 4; ++c)
        dst[(uint64_t)c * rv.words] = row[c];
    if (w == 0) {
        // last_symbol, LevMyers64.cpp:182 / 192
        const uint32_t last = rc ? rc_code(packed_symbol(codes, 0)) : packed_symbol(codes, len - 1);
        presence |= last << 16;
    }
    if (presence)
        atomicOr(meta + (read * (uint64_t)rv.n_orient + slot), presence);
}

// one thread per (read, slot, dword) of the top-aligned 32-bit vectors
__global__ __launch_bounds__(256) void pattern_vectors32_kernel(ReadsView rv, uint32_t* __restrict__ pm32)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = (uint32_t)(t % rv.words32);
    const uint64_t rs = t / rv.words32;
    const uint32_t slot = (uint32_t)(rs % rv.n_orient);
    const uint32_t read = (uint32_t)(rs / rv.n_orient);
    if (read >= rv.n)
        return;
    const bool rc = ((int32_t)slot == rv.slot_rc);
    const uint32_t len = rv.lens[read];
    const uint8_t* codes = rv.codes + rv.code_off[read];
    const int32_t phantom = (int32_t)(32 * rv.words32 - len);   // rows below the read's first symbol
    uint32_t row[4] = {0, 0, 0, 0};
    for (int32_t b = 0; b < 32; ++b) {
        const int32_t i = (int32_t)(32 * j) + b - phantom;
        const uint32_t bit = 1u << b;
        if (i < 0) {
            // a phantom row matches every text symbol: D stays 0 down to the read's row 0
            row[0] |= bit;
            row[1] |= bit;
            row[2] |= bit;
            row[3] |= bit;
        } else {
            const uint32_t c = rc ? rc_code(packed_symbol(codes, len - 1 - (uint32_t)i))
                                  : packed_symbol(codes, (uint32_t)i);
            if (c < 4)
                row[c] |= bit;
        }
    }
    uint32_t* dst = pm32 + ((uint64_t)(read * (uint64_t)rv.n_orient + slot) * 4) * rv.words32 + j;
    for (int c = 0; c < 4; ++c)
        dst[(uint64_t)c * rv.words32] = row[c];
}

int launch_reads_prepare(wamd_ctx* ctx, wamd_reads* r, const uint8_t* d_raw,
                         const uint64_t* d_raw_off, hipStream_t s)
{
    (void)ctx;
    const uint32_t bytes_per_read = ((r->max_len + 1) / 2 + 3) / 4 * 4;
    {
        const uint64_t threads = (uint64_t)r->n * bytes_per_read;
        const uint32_t blocks = (uint32_t)((threads + 255) / 256);
        hipLaunchKernelGGL(canonical_codes_kernel, dim3(blocks), dim3(256), 0, s, d_raw, d_raw_off,
                           r->d_lens, r->d_code_off, r->n, bytes_per_read,
                           r->encoding == WAMD_READS_ASCII ? 1 : 0, r->d_codes);
        WAMD_HIP(hipGetLastError());
    }
    {
        ReadsView rv = r->view();
        const uint64_t threads = (uint64_t)r->n * rv.n_orient * rv.words;
        const uint32_t blocks = (uint32_t)((threads + 255) / 256);
        hipLaunchKernelGGL(pattern_vectors_kernel, dim3(blocks), dim3(256), 0, s, rv, r->d_pm,
                           r->d_meta);
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

} // namespace wamd
