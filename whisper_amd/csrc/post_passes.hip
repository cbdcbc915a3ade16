// post_passes.hip - rows B, L, X of the path: clipBoundaryPositions, clipIllegalPositions,
// clipOverlappingPairs (mapper/soft_clipping.cpp:592-1151) and extract_cigar_and_mdz / extract_cigarbin_and_mdz
// (mapper/sam.cpp:805-1051) on the ext-CIGARs of a batch, any one of them (the stand-alone entry points) or all four
// behind one another in the order of sam.cpp:1889-1900 / 2002-2007 (finish_pairs).
//
// One string per lane, 64 slots (32 pairs) per wave.  A lane fetches its own slot into registers sixteen bytes at a
// time - all requests in flight at once - and parks it in LDS, where the wave's strings are packed by their lengths
// (LDS, not registers, limits the waves per CU here).  It digests the string once into a bitmap of its events (every
// symbol that is not '.') and then runs the passes of post_core.h over events and dot runs, never over bytes: a
// 150-symbol read with a handful of differences is a handful of iterations per pass, a clipping is a funnel-shift
// move of the kept dwords and of the bitmap, so one digest serves all four passes.  The slot goes back to memory once.
// Nothing crosses lanes but the two records of a pair (shuffles): each mate's lane clips its own string.
#include <hip/hip_runtime.h>

#include "post_core.h"
#include "wamd_internal.h"

namespace wamd {

using namespace post;

enum { POST_ILLEGAL = 1, POST_BOUNDARY = 2, POST_PAIRS = 4, POST_SAM = 8,
       POST_LEN_IN_RESERVED = 16,   // lens[].reserved = length of the ext-CIGAR + 2 (finish_pairs' bookkeeping)
       POST_TWO_HALVES = 32 };      // test hook: never pack

// dwords of LDS a string of `str_dw` dwords takes with its event bitmap; odd, so that lanes at the same offset of
// their strings hit different banks
__host__ __device__ static inline uint32_t region_dw(uint32_t str_dw) { return (str_dw + (str_dw + 7) / 8 + 1) | 1u; }

__device__ __forceinline__ wamd_clip_record mate_record(const wamd_clip_record& d)
{
    uint32_t w[8], v[8];
    static_assert(sizeof(wamd_clip_record) == 32, "eight dwords travel between the lanes of a pair");
    memcpy(w, &d, 32);
#pragma unroll
    for (int q = 0; q < 8; ++q)
        v[q] = (uint32_t)__shfl_xor((int)w[q], 1);
    wamd_clip_record m;
    memcpy(&m, v, 32);
    return m;
}

__global__ __launch_bounds__(64, 3) void post_passes_kernel(uint32_t n, const wamd_clip_illegal_task* __restrict__ it,
                                                         const wamd_clip_pair_task* __restrict__ pt,
                                                         wamd_clip_record* __restrict__ records, uint8_t* __restrict__ ext,
                                                         uint32_t ext_stride, const uint32_t* __restrict__ cap_hint,
                                                         uint32_t budget_dw, wamd_scoring sc, uint32_t passes,
                                                         uint8_t* __restrict__ cigars, uint32_t cigar_stride,
                                                         uint8_t* __restrict__ mdz, uint32_t mdz_stride,
                                                         uint32_t* __restrict__ cigar_bin, uint32_t bin_stride,
                                                         wamd_sam_cigar_len* __restrict__ lens, int2* __restrict__ mid,
                                                         uint32_t* __restrict__ flags)
{
    extern __shared__ uint32_t post_lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t base = blockIdx.x * 64;
    const uint32_t here = n - base < 64u ? n - base : 64u;   // slots of this wave (pairs stay together: base is even)
    const uint32_t slot = base + lane;
    const bool live = lane < here;
    const uint32_t full_dw = (ext_stride / 4) | 1u;
    // What a lane's string needs.  A caller that knows how long the strings are (and can get: a clipping pass never
    // lengthens one, a perfect match may receive read_length dots) says so per slot; otherwise the lane finds the NUL
    // in the registers it has fetched, or - strings beyond them - the wave measures its slots one after the other
    // with the lanes side by side on a slot's dwords.  When the packed strings do not fit the budget, the wave serves
    // its lanes in groups at the full stride.
    constexpr int POST_CH = 12;
    const bool regs_ok = (ext_stride & 15u) == 0 && ((uintptr_t)ext & 15u) == 0;
    wamd_u32x4 rg[POST_CH];
    uint32_t my_dw = 0;
    bool in_regs = regs_ok;   // (wave-uniform from here on)
    if (cap_hint) {
        if (live) {
            const uint32_t c = cap_hint[slot] < ext_stride ? cap_hint[slot] : ext_stride;
            my_dw = ((c + 3) / 4) | 1u;
            my_dw = my_dw > full_dw ? full_dw : my_dw;
        }
        in_regs = regs_ok && !__any(live && my_dw * 4 > 16u * POST_CH);
        if (in_regs) {
            const wamd_u32x4* src = reinterpret_cast<const wamd_u32x4*>(ext + (uint64_t)(live ? slot : base) * ext_stride);
#pragma unroll
            for (int q = 0; q < POST_CH; ++q)
                if (live && 16u * q < my_dw * 4 && 16u * q < ext_stride)
                    rg[q] = src[q];
        }
    } else {
        uint32_t my_len = ext_stride;   // bytes up to and including the dword that holds the NUL
        bool found = false;
        if (regs_ok) {
            const wamd_u32x4* src = reinterpret_cast<const wamd_u32x4*>(ext + (uint64_t)(live ? slot : base) * ext_stride);
#pragma unroll
            for (int q = 0; q < POST_CH; ++q)
                if (live && 16u * q < ext_stride)
                    rg[q] = src[q];
#pragma unroll
            for (int q = 0; q < POST_CH; ++q)
                if (live && !found && 16u * q < ext_stride) {
                    const uint32_t w[4] = {rg[q].x, rg[q].y, rg[q].z, rg[q].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!found && zero_bytes(w[j])) {
                            found = true;
                            my_len = 16u * q + 4u * j + 4u;
                        }
                }
            if (ext_stride <= 16u * POST_CH)
                found = true;   // the whole slot is in the registers
        }
        const bool measured = regs_ok && !__any(live && !found);
        if (!measured) {
            for (uint32_t sl = 0; sl < here; ++sl) {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(ext + (uint64_t)(base + sl) * ext_stride);
                uint32_t len = ext_stride;
                for (uint32_t i0 = 0; i0 < ext_stride / 4; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    const bool z = i < ext_stride / 4 && zero_bytes(src[i]);
                    const uint64_t m = __ballot(z);
                    if (m) {
                        len = 4 * (i0 + (uint32_t)__builtin_ctzll(m)) + 4;
                        break;
                    }
                }
                if (lane == sl)
                    my_len = len;
            }
        }
        if (live) {
            // a perfect match may be filled with read_length dots (clipIllegalPositions 614-618, clipOverlappingPairs
            // 1029-1036), and the tail loop of clipIllegalPositions pads up to read_length
            uint32_t grow = (passes & POST_ILLEGAL) ? (uint32_t)it[slot].read_length + 1u : 0u;
            if (passes & POST_PAIRS) {
                const uint32_t g2 = (uint32_t)pt[slot >> 1].read_length[lane & 1u] + 1u;
                grow = grow > g2 ? grow : g2;
            }
            uint32_t c = (my_len > grow ? my_len : grow) + 8u;
            c = c < ext_stride ? c : ext_stride;
            my_dw = ((c + 3) / 4) | 1u;
            my_dw = my_dw > full_dw ? full_dw : my_dw;
        }
        in_regs = measured && !__any(live && my_dw * 4 > 16u * POST_CH && ext_stride > 16u * POST_CH);
    }
    const uint32_t my_reg = live ? region_dw(my_dw) : 0u;
    uint32_t incl = my_reg;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= (uint32_t)d)
            incl += o;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    // lanes served together when the strings are not packed: as many (a power of two, pairs stay together) as fit
    // the budget at the full stride
    const uint32_t full_reg = region_dw(full_dw);
    uint32_t per = 64, per_sh = 6;
    while (per > 2 && per * full_reg > budget_dw) {
        per >>= 1;
        --per_sh;
    }
    const bool packed = total <= budget_dw && !(passes & POST_TWO_HALVES);
    if (!packed && per == 64) {   // (the test hook on short strides: still two groups)
        per = 32;
        per_sh = 5;
    }
    const int groups = packed ? 1 : (int)(64u / per);
    const Fold fold = make_fold(sc);
    const bool writes = passes & (POST_ILLEGAL | POST_BOUNDARY | POST_PAIRS);   // (extract_cigar_and_mdz only reads)
    for (int h = 0; h < groups; ++h) {
        const bool mine = packed || (lane >> per_sh) == (uint32_t)h;
        uint32_t ndw = mine ? my_dw : 0u, off = incl - my_reg;
        if (!packed) {
            ndw = (mine && live) ? full_dw : 0u;
            off = (lane & (per - 1u)) * full_reg;
        }
        const uint32_t cap = ndw * 4 < ext_stride ? ndw * 4 : ext_stride;   // bytes of the string the routines may use
        const uint32_t lim_dw = cap / 4;                                    // dwords that travel
        const bool by_regs = in_regs && packed;
        if (by_regs) {
#pragma unroll
            for (int q = 0; q < POST_CH; ++q) {
                const uint32_t w[4] = {rg[q].x, rg[q].y, rg[q].z, rg[q].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4u * q + j < lim_dw)
                        post_lds[off + 4 * q + j] = w[j];
            }
        } else {
            for (uint32_t sl = 0; sl < here; ++sl) {   // coalesced: the wave copies one string after the other
                const uint32_t n_s = (uint32_t)__builtin_amdgcn_readlane((int)lim_dw, (int)sl);
                if (!n_s)
                    continue;
                const uint32_t o_s = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)sl);
                const uint32_t* src = reinterpret_cast<const uint32_t*>(ext + (uint64_t)(base + sl) * ext_stride);
                for (uint32_t i = lane; i < n_s; i += 64)
                    post_lds[o_s + i] = src[i];
            }
            __syncthreads();
        }
        bool act = live && mine;
        Str s;
        s.dw = post_lds + off;
        s.ev = post_lds + off + ndw;
        s.cap = cap;
        s.len = 0;
        uint32_t err = 0;
        if (act) {
            err = digest(s);
            act = err == 0;     // a string that breaks the grammar stays as it is; the call fails
        }
        wamd_clip_record d{};
        if (writes) {
            if (act) {
                d = records[slot];
                if (passes & POST_ILLEGAL)
                    err |= clip_illegal(s, d, it[slot], fold);
                if (passes & POST_BOUNDARY)
                    err |= clip_boundary(s, d, fold);
                if (mid)   // where the mapping lies before the pair is clipped (finish_pairs: isDistant of the mate)
                    mid[slot] = make_int2(d.ref_seq_pos, (int)d.ref_length);
            }
            if (passes & POST_PAIRS) {
                const wamd_clip_record mate = mate_record(d);
                const bool both = __shfl_xor((int)act, 1) != 0 && (lane ^ 1u) < here;
                if (act && both)
                    err |= clip_pair_member(s, d, mate, pt[slot >> 1], lane & 1u);
            }
            if (act)
                records[slot] = d;
        }
        if (act && (passes & POST_SAM)) {
            wamd_sam_cigar_len l;
            err |= sam_cigar(s, cigars + (uint64_t)slot * cigar_stride, cigar_stride, mdz + (uint64_t)slot * mdz_stride,
                             mdz_stride, cigar_bin ? cigar_bin + (uint64_t)slot * bin_stride : nullptr, bin_stride, l);
            if (passes & POST_LEN_IN_RESERVED)
                l.reserved = s.len + 2;   // (0 / 1 = not measured)
            lens[slot] = l;
        }
        if (err)
            raise_flag(flags, err);
        if (writes) {
            if (by_regs) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(ext + (uint64_t)(live ? slot : base) * ext_stride);
#pragma unroll
                for (int q = 0; q < POST_CH; ++q) {
                    if (4u * q + 4u <= lim_dw) {
                        wamd_u32x4 v = {post_lds[off + 4 * q], post_lds[off + 4 * q + 1], post_lds[off + 4 * q + 2],
                                        post_lds[off + 4 * q + 3]};
                        *reinterpret_cast<wamd_u32x4*>(dst + 4 * q) = v;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (4u * q + j < lim_dw)
                                dst[4 * q + j] = post_lds[off + 4 * q + j];
                    }
                }
            } else {
                __syncthreads();
                for (uint32_t sl = 0; sl < here; ++sl) {
                    const uint32_t n_s = (uint32_t)__builtin_amdgcn_readlane((int)lim_dw, (int)sl);
                    if (!n_s)
                        continue;
                    const uint32_t o_s = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)sl);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(ext + (uint64_t)(base + sl) * ext_stride);
                    for (uint32_t i = lane; i < n_s; i += 64)
                        dst[i] = post_lds[o_s + i];
                }
            }
        }
        if (!by_regs)
            __syncthreads();   // the next group parks its strings where these lie
    }
}

// Slots are staged a dword at a time: the stride is a multiple of 4, the first slot 4-byte aligned.  Two strings at
// the full stride, with their bitmaps, must fit a block's LDS.
constexpr uint32_t POST_MAX_STRIDE = 16384;
constexpr uint32_t POST_BUDGET_DW = 3392;   // 13.25 KB of LDS per wave: twelve waves per CU

static int launch_post_kernel(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* it, const wamd_clip_pair_task* pt,
                              wamd_clip_record* rec, uint8_t* ext, uint32_t ext_stride, const uint32_t* cap_hint,
                              uint32_t passes, uint8_t* cigars, uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride,
                              uint32_t* bin, uint32_t bin_stride, wamd_sam_cigar_len* lens, hipStream_t s,
                              int2* mid = nullptr)
{
    if ((ext_stride & 3u) != 0 || ext_stride < 4 || ext_stride > POST_MAX_STRIDE || ((uintptr_t)ext & 3u) != 0) {
        set_error("ext-CIGAR slots: the stride must be a multiple of 4 in 4 .. %u and the slots 4-byte aligned (stride %u)",
                  POST_MAX_STRIDE, ext_stride);
        return WAMD_ERR_INVALID_ARG;
    }
    if (n == 0)
        return WAMD_OK;
    if (ctx->test_hooks & HOOK_POST_NO_HINTS) {   // test hook: no packing, every wave serves its slots in groups
        cap_hint = nullptr;
        passes |= POST_TWO_HALVES;
    }
    const uint32_t full_reg = region_dw((ext_stride / 4) | 1u);
    uint32_t budget_dw = 64u * full_reg;
    if (budget_dw > POST_BUDGET_DW)
        budget_dw = POST_BUDGET_DW > 2u * full_reg ? POST_BUDGET_DW : 2u * full_reg;
    hipLaunchKernelGGL(post_passes_kernel, dim3((n + 63) / 64), dim3(64), (size_t)budget_dw * 4, s, n, it, pt, rec, ext,
                       ext_stride, cap_hint, budget_dw, ctx->scoring, passes, cigars, cigar_stride, mdz, mdz_stride, bin,
                       bin_stride, lens, mid, ctx->d_flags);
    WAMD_HIP(hipGetLastError());
    return WAMD_OK;
}

int launch_clip_boundary(wamd_ctx* ctx, uint32_t n, wamd_clip_record* d_records, uint8_t* d_cigars,
                         uint32_t cigar_stride, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    return launch_post_kernel(ctx, n, nullptr, nullptr, d_records, d_cigars, cigar_stride, nullptr, POST_BOUNDARY, nullptr, 0,
                              nullptr, 0, nullptr, 0, nullptr, s);
}

int launch_clip_illegal(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_tasks, wamd_clip_record* d_records,
                        uint8_t* d_cigars, uint32_t cigar_stride, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    return launch_post_kernel(ctx, n, d_tasks, nullptr, d_records, d_cigars, cigar_stride, nullptr, POST_ILLEGAL, nullptr, 0,
                              nullptr, 0, nullptr, 0, nullptr, s);
}

int launch_clip_pairs(wamd_ctx* ctx, uint32_t n_pairs, const wamd_clip_pair_task* d_tasks, wamd_clip_record* d_records,
                      uint8_t* d_cigars, uint32_t cigar_stride, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n_pairs > 0x7fffffffu) {
        set_error("clip_overlapping_pairs: %u pairs are more than one call takes", n_pairs);
        return WAMD_ERR_INVALID_ARG;
    }
    return launch_post_kernel(ctx, 2 * n_pairs, nullptr, d_tasks, d_records, d_cigars, cigar_stride, nullptr, POST_PAIRS,
                              nullptr, 0, nullptr, 0, nullptr, 0, nullptr, s);
}

int launch_post_passes(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_it, const wamd_clip_pair_task* d_pt,
                       wamd_clip_record* d_rec, uint8_t* d_ext, uint32_t ext_stride, const uint32_t* d_cap_hint,
                       int do_boundary, int do_pairs, uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz,
                       uint32_t mdz_stride, wamd_sam_cigar_len* d_lens, int2* d_mid, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    if (n & 1u) {
        set_error("post passes: %u slots are not whole pairs", n);
        return WAMD_ERR_INVALID_ARG;
    }
    return launch_post_kernel(ctx, n, d_it, d_pt, d_rec, d_ext, ext_stride, d_cap_hint,
                              POST_ILLEGAL | (do_boundary ? POST_BOUNDARY : 0) | (do_pairs ? POST_PAIRS : 0) | POST_SAM |
                                  POST_LEN_IN_RESERVED,
                              d_cigars, cigar_stride, d_mdz, mdz_stride, nullptr, 0, d_lens, s, d_mid);
}

// clipIllegalPositions and (optionally) clipBoundaryPositions of n independent records (the mate rescue, sam_matching.cpp:444-450)
int launch_clip_records(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_it, wamd_clip_record* d_rec, uint8_t* d_ext,
                        uint32_t ext_stride, const uint32_t* d_cap_hint, int do_boundary, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    return launch_post_kernel(ctx, n, d_it, nullptr, d_rec, d_ext, ext_stride, d_cap_hint,
                              POST_ILLEGAL | (do_boundary ? POST_BOUNDARY : 0), nullptr, 0, nullptr, 0, nullptr, 0, nullptr, s);
}

int launch_sam_cigar(wamd_ctx* ctx, uint32_t n, const uint8_t* d_ext, uint32_t ext_stride, uint8_t* d_cigars,
                     uint32_t cigar_stride, uint8_t* d_mdz, uint32_t mdz_stride, uint32_t* d_bin, uint32_t bin_stride,
                     wamd_sam_cigar_len* d_lens, hipStream_t s)
{
    WAMD_HIP(hipSetDevice(ctx->device));
    return launch_post_kernel(ctx, n, nullptr, nullptr, nullptr, const_cast<uint8_t*>(d_ext), ext_stride, nullptr, POST_SAM,
                              d_cigars, cigar_stride, d_mdz, mdz_stride, d_bin, bin_stride, d_lens, s);
}

} // namespace wamd
