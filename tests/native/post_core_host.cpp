// Test harness, not product: whisper_amd/csrc/post_core.h - the per-string routines of rows B / L / X that the HIP
// kernel runs one lane each - compiled for the host, so that the CPU suite can hold them against the oracle record by
// record (the same header, the same arithmetic; only the staging around it is the kernel's own).  After every pass the
// event bitmap the routines maintained by splicing is compared with a fresh digest of the string they left.
#include <cstring>
#include <vector>

#include "../../whisper_amd/csrc/post_core.h"

using namespace wamd::post;

namespace {
struct Held {
    std::vector<uint32_t> dw, ev;
    Str s;
    uint32_t bad = 0;
    Held(const uint8_t* slot, uint32_t stride)
    {
        const uint32_t cap = stride & ~3u;
        dw.assign(cap / 4 + 1, 0);
        ev.assign((cap / 4 + 7) / 8 + 2, 0);
        memcpy(dw.data(), slot, cap);
        s.dw = dw.data();
        s.ev = ev.data();
        s.cap = cap;
        s.len = 0;
        bad = digest(s);
    }
    void store(uint8_t* slot) const { memcpy(slot, dw.data(), s.cap); }
    // 0 when the maintained bitmap and length equal those of a fresh digest
    int stale() const
    {
        std::vector<uint32_t> d2(dw), e2(ev.size(), 0);
        Str t = s;
        t.dw = d2.data();
        t.ev = e2.data();
        if (digest(t))
            return 1;
        if (t.len != s.len)
            return 2;
        for (uint32_t i = 0; i < s.len; ++i)
            if (((ev[i >> 5] >> (i & 31)) & 1u) != ((e2[i >> 5] >> (i & 31)) & 1u))
                return 3;
        return 0;
    }
};
}

extern "C" {

int pc_clip_boundary_batch(uint32_t n, wamd_clip_record* rec, uint8_t* cig, uint32_t stride, const wamd_scoring* sc)
{
    const Fold f = make_fold(*sc);
    int rc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        Held h(cig + (size_t)i * stride, stride);
        if (h.bad) {
            rc |= (int)h.bad;
            continue;
        }
        rc |= (int)clip_boundary(h.s, rec[i], f);
        if (h.stale())
            rc |= 1 << 20;
        h.store(cig + (size_t)i * stride);
    }
    return rc;
}

int pc_clip_illegal_batch(uint32_t n, const wamd_clip_illegal_task* t, wamd_clip_record* rec, uint8_t* cig, uint32_t stride,
                          const wamd_scoring* sc)
{
    const Fold f = make_fold(*sc);
    int rc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        Held h(cig + (size_t)i * stride, stride);
        if (h.bad) {
            rc |= (int)h.bad;
            continue;
        }
        rc |= (int)clip_illegal(h.s, rec[i], t[i], f);
        if (h.stale())
            rc |= 1 << 20;
        h.store(cig + (size_t)i * stride);
    }
    return rc;
}

int pc_clip_pairs_batch(uint32_t n_pairs, const wamd_clip_pair_task* t, wamd_clip_record* rec, uint8_t* cig, uint32_t stride)
{
    int rc = 0;
    for (uint32_t p = 0; p < n_pairs; ++p) {
        const wamd_clip_record before[2] = {rec[2 * p], rec[2 * p + 1]};
        for (uint32_t q = 0; q < 2; ++q) {
            uint8_t* slot = cig + (size_t)(2 * p + q) * stride;
            Held h(slot, stride);
            if (h.bad) {
                rc |= (int)h.bad;
                continue;
            }
            rc |= (int)clip_pair_member(h.s, rec[2 * p + q], before[1 - q], t[p], q);
            if (h.stale())
                rc |= 1 << 20;
            h.store(slot);
        }
    }
    return rc;
}

int pc_sam_cigar_batch(uint32_t n, const uint8_t* ext, uint32_t ext_stride, uint8_t* cig, uint32_t cigar_stride, uint8_t* mdz,
                       uint32_t mdz_stride, uint32_t* bin, uint32_t bin_stride, wamd_sam_cigar_len* lens)
{
    int rc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        Held h(ext + (size_t)i * ext_stride, ext_stride);
        if (h.bad) {
            rc |= (int)h.bad;
            continue;
        }
        rc |= (int)sam_cigar(h.s, cig + (size_t)i * cigar_stride, cigar_stride, mdz + (size_t)i * mdz_stride, mdz_stride,
                             bin ? bin + (size_t)i * bin_stride : nullptr, bin_stride, lens[i]);
    }
    return rc;
}

// rows L, B, L, X behind one another on one digest, as the fused kernel runs them
int pc_post_passes(uint32_t n, const wamd_clip_illegal_task* it, const wamd_clip_pair_task* pt, wamd_clip_record* rec,
                   uint8_t* ext, uint32_t ext_stride, const wamd_scoring* sc, int do_boundary, int do_pairs, uint8_t* cig,
                   uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride, wamd_sam_cigar_len* lens)
{
    const Fold f = make_fold(*sc);
    int rc = 0;
    for (uint32_t p = 0; p + 1 < n; p += 2) {
        Held h0(ext + (size_t)p * ext_stride, ext_stride), h1(ext + (size_t)(p + 1) * ext_stride, ext_stride);
        Held* h[2] = {&h0, &h1};
        if (h0.bad || h1.bad) {
            rc |= (int)(h0.bad | h1.bad);
            continue;
        }
        for (uint32_t q = 0; q < 2; ++q) {
            rc |= (int)clip_illegal(h[q]->s, rec[p + q], it[p + q], f);
            if (h[q]->stale())
                rc |= 1 << 21;
            if (do_boundary)
                rc |= (int)clip_boundary(h[q]->s, rec[p + q], f);
            if (h[q]->stale())
                rc |= 1 << 22;
        }
        if (do_pairs) {
            const wamd_clip_record before[2] = {rec[p], rec[p + 1]};
            for (uint32_t q = 0; q < 2; ++q)
                rc |= (int)clip_pair_member(h[q]->s, rec[p + q], before[1 - q], pt[p / 2], q);
        }
        for (uint32_t q = 0; q < 2; ++q) {
            if (h[q]->stale())
                rc |= 1 << 20;
            rc |= (int)sam_cigar(h[q]->s, cig + (size_t)(p + q) * cigar_stride, cigar_stride, mdz + (size_t)(p + q) * mdz_stride,
                                 mdz_stride, nullptr, 0, lens[p + q]);
            h[q]->store(ext + (size_t)(p + q) * ext_stride);
        }
    }
    return rc;
}
}
