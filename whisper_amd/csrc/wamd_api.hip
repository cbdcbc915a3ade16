// wamd_api.hip - host side of the C-ABI (include/whisper_amd.h): contexts, read
// batches, staging for the host-pointer entry points.  No verification happens
// here and there is no CPU fallback: every entry point ends in a kernel launch.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "wamd_internal.h"

namespace wamd {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    set_error("%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return WAMD_ERR_HIP;
}

GenomeView genome_view(const wamd_ctx* ctx)
{
    GenomeView g;
    g.dwords = reinterpret_cast<const uint32_t*>(ctx->d_genome);
    g.n_symbols = 2 * ctx->genome_bytes;
    // an owned copy is padded to a multiple of 4 bytes plus 64; a borrowed one may
    // only be read up to its last whole dword
    uint64_t dwords = ctx->owns_genome ? (ctx->genome_bytes + 3) / 4 + 8 : ctx->genome_bytes / 4;
    g.last_dword = dwords ? (uint32_t)(dwords - 1) : 0;
    return g;
}

int ensure_stage(wamd_ctx* ctx, uint64_t dev_bytes, uint64_t host_bytes)
{
    if (dev_bytes > ctx->d_stage_bytes) {
        if (ctx->d_stage)
            WAMD_HIP(hipFree(ctx->d_stage));
        ctx->d_stage = nullptr;
        ctx->d_stage_bytes = 0;
        uint64_t want = dev_bytes + dev_bytes / 4 + 4096;
        WAMD_HIP(hipMalloc(&ctx->d_stage, want));
        ctx->d_stage_bytes = want;
    }
    if (host_bytes > ctx->h_stage_bytes) {
        if (ctx->h_stage)
            WAMD_HIP(hipHostFree(ctx->h_stage));
        ctx->h_stage = nullptr;
        ctx->h_stage_bytes = 0;
        uint64_t want = host_bytes + host_bytes / 4 + 4096;
        WAMD_HIP(hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault));
        ctx->h_stage_bytes = want;
    }
    return WAMD_OK;
}

int flags_to_status(uint32_t f)
{
    if (!f)
        return WAMD_OK;
    if (f & FLAG_TEXT_LEN) {
        set_error("max_distance_in_ref == 0 or above max_text_len (the reference throws here)");
        return WAMD_ERR_TEXT_LEN;
    }
    if (f & FLAG_RANGE) {
        set_error("a task's window leaves the packed genome");
        return WAMD_ERR_RANGE;
    }
    if (f & FLAG_VARIANT) {
        set_error("a task's variant cannot hold its read, or the read is too long");
        return WAMD_ERR_TOO_LONG;
    }
    set_error("malformed task (device flags 0x%x)", f);
    return WAMD_ERR_INVALID_ARG;
}

static int fetch_flags(wamd_ctx* ctx, hipStream_t s, uint32_t* out)
{
    WAMD_HIP(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemsetAsync(ctx->d_flags, 0, sizeof(uint32_t), s));
    WAMD_HIP(hipStreamSynchronize(s));
    *out = *ctx->h_flags;
    return WAMD_OK;
}

// host-pointer form of the two clipping post-passes: tasks up, records and ext-CIGARs up and back
template <class Task, class Call>
static int clip_pass_host(wamd_ctx* ctx, uint32_t n_tasks, uint32_t n_records, const Task* tasks,
                          wamd_clip_record* records, uint8_t* cigars, uint32_t cigar_stride, const char* who, Call call)
{
    if (n_tasks == 0)
        return WAMD_OK;
    if (!ctx || !tasks || !records || !cigars || cigar_stride < 2)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    const uint64_t tb = (uint64_t)n_tasks * sizeof(Task), rb = (uint64_t)n_records * sizeof(wamd_clip_record),
                   cb = (uint64_t)n_records * cigar_stride;
    int rc = ensure_stage(ctx, up(tb) + up(rb) + up(cb), 0);
    if (rc)
        return rc;
    char* base = (char*)ctx->d_stage;
    auto* d_tasks = (Task*)base;
    auto* d_rec = (wamd_clip_record*)(base + up(tb));
    auto* d_cig = (uint8_t*)(base + up(tb) + up(rb));
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    WAMD_HIP(hipMemcpyAsync(d_rec, records, rb, hipMemcpyHostToDevice, s));
    WAMD_HIP(hipMemcpyAsync(d_cig, cigars, cb, hipMemcpyHostToDevice, s));
    rc = call(d_tasks, d_rec, d_cig, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(records, d_rec, rb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(cigars, d_cig, cb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    if (!rc && flags) {
        set_error("%s: a record does not describe its ext-CIGAR, or the ext-CIGAR does not fit its stride", who);
        return WAMD_ERR_INVALID_ARG;
    }
    return rc;
}

} // namespace wamd

using namespace wamd;

wamd::ReadsView wamd_reads::view() const
{
    ReadsView v;
    v.codes = d_codes;
    v.code_off = d_code_off;
    v.lens = d_lens;
    v.pm = d_pm;
    v.pm32 = d_pm32;
    v.meta = d_meta;
    v.n = n;
    v.words = words;
    v.words32 = words32;
    v.pm32_stride = pm32_stride;
    v.n_orient = (orientations == 3) ? 2 : 1;
    v.slot_direct = (orientations & 1) ? 0 : -1;
    v.slot_rc = (orientations & 2) ? ((orientations & 1) ? 1 : 0) : -1;
    return v;
}

extern "C" {

const char* wamd_version(void) { return "whisper_amd 0.1 (gfx950; verification path of Whisper 2.0.2)"; }

const char* wamd_last_error(void) { return g_error; }

int wamd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

static int ctx_create_common(int device, const uint8_t* genome, bool on_device, uint64_t genome_bytes,
                             uint32_t ref_size, uint32_t cur_offset, wamd_ctx** out)
{
    if (!out || !genome || genome_bytes == 0) {
        set_error("wamd_ctx_create: null or empty genome");
        return WAMD_ERR_INVALID_ARG;
    }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: this library has no CPU path");
        return WAMD_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (%d visible)", device, n);
        return WAMD_ERR_INVALID_ARG;
    }
    WAMD_HIP(hipSetDevice(device));
    {   // this device's address of the static scratch of wamd_emit_lev_records_static_dev
        int rc0 = records_static_init(device);
        if (rc0)
            return rc0;
    }
    wamd_ctx* c = new wamd_ctx();   // value-initialised: every plain member starts at zero
    c->device = device;
    c->genome_bytes = genome_bytes;
    c->ref_size = ref_size;
    c->cur_offset = cur_offset;
    c->max_text_len = 16384;   // mapper/sam.cpp:190
    c->scoring = wamd_scoring{1.0, -5.0, -5.0, -0.4, -5.0, -0.4, -6.0};   // mapper/params.h:145-154
    int rc = WAMD_OK;
    do {
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
            break;
        }
        if (on_device) {
            if ((genome_bytes & 3u) || ((uintptr_t)genome & 3u)) {
                set_error("wamd_ctx_create_dev: a borrowed genome is read as dwords - 4-byte aligned, a multiple of "
                          "4 bytes (pad with 0xff)");
                rc = WAMD_ERR_INVALID_ARG;
                break;
            }
            c->d_genome = const_cast<uint8_t*>(genome);
            c->owns_genome = false;
        } else {
            uint64_t padded = (genome_bytes + 3) / 4 * 4 + 64;
            if ((e = hipMalloc(&c->d_genome, padded)) != hipSuccess) {
                rc = hip_fail(e, "hipMalloc(genome)", __FILE__, __LINE__);
                break;
            }
            c->owns_genome = true;
            // pad with 0xff: code 15 never matches anything (index/ref_seq.cpp:148-154)
            if ((e = hipMemsetAsync(c->d_genome + genome_bytes / 4 * 4, 0xff,
                                    padded - genome_bytes / 4 * 4, c->stream)) != hipSuccess ||
                (e = hipMemcpyAsync(c->d_genome, genome, genome_bytes, hipMemcpyHostToDevice,
                                    c->stream)) != hipSuccess) {
                rc = hip_fail(e, "genome upload", __FILE__, __LINE__);
                break;
            }
        }
        if ((e = hipMalloc(&c->d_flags, (4 + 2 * BAND_SLOTS) * sizeof(uint32_t))) != hipSuccess ||
            (e = hipHostMalloc(&c->h_flags, sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess ||
            (e = hipMemsetAsync(c->d_flags, 0, (4 + 2 * BAND_SLOTS) * sizeof(uint32_t), c->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
            rc = hip_fail(e, "context buffers", __FILE__, __LINE__);
            break;
        }
    } while (0);
    if (rc != WAMD_OK) {
        wamd_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return WAMD_OK;
}

int wamd_ctx_create(int device, const uint8_t* genome, uint64_t genome_bytes, uint32_t ref_size,
                    uint32_t cur_offset, wamd_ctx** out)
{
    return ctx_create_common(device, genome, false, genome_bytes, ref_size, cur_offset, out);
}

int wamd_ctx_create_dev(int device, const uint8_t* d_genome, uint64_t genome_bytes,
                        uint32_t ref_size, uint32_t cur_offset, wamd_ctx** out)
{
    return ctx_create_common(device, d_genome, true, genome_bytes, ref_size, cur_offset, out);
}

int wamd_ctx_destroy(wamd_ctx* c)
{
    if (!c)
        return WAMD_OK;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    if (c->owns_genome && c->d_genome)
        (void)hipFree(c->d_genome);
    if (c->d_flags)
        (void)hipFree(c->d_flags);
    if (c->h_flags)
        (void)hipHostFree(c->h_flags);
    if (c->d_stage)
        (void)hipFree(c->d_stage);
    if (c->h_stage)
        (void)hipHostFree(c->h_stage);
    for (hipStream_t ps : c->pipe)
        if (ps)
            (void)hipStreamDestroy(ps);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return WAMD_OK;
}

int wamd_ctx_set_scoring(wamd_ctx* ctx, const wamd_scoring* s)
{
    if (!ctx || !s)
        return WAMD_ERR_INVALID_ARG;
    ctx->scoring = *s;
    return WAMD_OK;
}

int wamd_ctx_set_max_text_len(wamd_ctx* ctx, uint32_t max_text_len)
{
    if (!ctx || max_text_len < 10)   // LevMyers64.cpp:124-125 exits below 10
        return WAMD_ERR_INVALID_ARG;
    ctx->max_text_len = max_text_len;
    return WAMD_OK;
}

// not part of include/whisper_amd.h: switches the tests use to reach code paths that ordinary inputs seldom take
int wamd_internal_set_test_hooks(wamd_ctx* ctx, uint32_t hooks)
{
    if (!ctx)
        return WAMD_ERR_INVALID_ARG;
    ctx->test_hooks = hooks;
    return WAMD_OK;
}

int wamd_ctx_set_distance_mode(wamd_ctx* ctx, int mode)
{
    if (!ctx || (mode != WAMD_DISTANCE_EXACT && mode != WAMD_DISTANCE_ACCEPTED_ONLY))
        return WAMD_ERR_INVALID_ARG;
    ctx->distance_mode = mode;
    return WAMD_OK;
}

int wamd_ctx_set_reference_params(wamd_ctx* ctx, uint32_t ref_size, uint32_t cur_offset)
{
    if (!ctx)
        return WAMD_ERR_INVALID_ARG;
    ctx->ref_size = ref_size;
    ctx->cur_offset = cur_offset;
    return WAMD_OK;
}

void* wamd_ctx_stream(wamd_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int wamd_ctx_synchronize(wamd_ctx* ctx)
{
    if (!ctx)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    WAMD_HIP(hipStreamSynchronize(ctx->stream));
    return WAMD_OK;
}

int wamd_ctx_take_flags(wamd_ctx* ctx, void* stream, uint32_t* flags)
{
    if (!ctx || !flags)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);   // h_flags is one pinned word per context
    return fetch_flags(ctx, stream ? (hipStream_t)stream : ctx->stream, flags);
}

// ---- reads -----------------------------------------------------------------------

// what a batch's buffers are sized by: found by a host pass over offsets / lens, or stated by the caller
struct ReadSizes {
    uint32_t max_len;
    uint64_t raw_end;   // bytes of `data` the reads reach
    uint64_t acc;       // bytes of all reads' canonical codes (an upper bound when stated)
    bool stated;        // the device holds every read against max_len / raw_end before anything uses it
};
static int reads_upload_on(wamd_ctx* ctx, const uint8_t* data, const uint64_t* offsets, const uint32_t* lens, uint32_t n,
                           int encoding, int orientations, hipStream_t s, wamd_reads* reuse, wamd_reads** out,
                           const ReadSizes* stated = nullptr);

int wamd_reads_upload(wamd_ctx* ctx, const uint8_t* data, const uint64_t* offsets,
                      const uint32_t* lens, uint32_t n, int encoding, int orientations,
                      wamd_reads** out)
{
    if (out)
        *out = nullptr;
    return reads_upload_on(ctx, data, offsets, lens, n, encoding, orientations, ctx ? ctx->stream : nullptr, nullptr, out);
}

int wamd_reads_upload_stream(wamd_ctx* ctx, const uint8_t* data, const uint64_t* offsets, const uint32_t* lens, uint32_t n,
                             int encoding, int orientations, void* stream, wamd_reads** out)
{
    if (out)
        *out = nullptr;
    return reads_upload_on(ctx, data, offsets, lens, n, encoding, orientations,
                           ctx && !stream ? ctx->stream : (hipStream_t)stream, nullptr, out);
}

int wamd_reads_refill(wamd_reads* reads, const uint8_t* data, const uint64_t* offsets, const uint32_t* lens, uint32_t n,
                      void* stream)
{
    if (!reads) {
        set_error("wamd_reads_refill: null batch");
        return WAMD_ERR_INVALID_ARG;
    }
    wamd_reads* same = nullptr;
    return reads_upload_on(reads->ctx, data, offsets, lens, n, reads->encoding, reads->orientations,
                           stream ? (hipStream_t)stream : reads->ctx->stream, reads, &same);
}

int wamd_reads_refill_async(wamd_reads* reads, const uint8_t* data, uint64_t data_bytes, const uint64_t* offsets,
                            const uint32_t* lens, uint32_t n, uint32_t max_len, void* stream)
{
    if (!reads || max_len == 0 || max_len > WAMD_MAX_READ_LEN || (n && data_bytes == 0)) {
        set_error("wamd_reads_refill_async: null batch, no bytes, or max_len outside 1..%u", WAMD_MAX_READ_LEN);
        return WAMD_ERR_INVALID_ARG;
    }
    ReadSizes sz;
    sz.max_len = max_len;
    sz.raw_end = data_bytes;
    sz.acc = (uint64_t)n * (((uint64_t)(max_len + 1) / 2 + 3) / 4 * 4);
    sz.stated = true;
    wamd_reads* same = nullptr;
    return reads_upload_on(reads->ctx, data, offsets, lens, n, reads->encoding, reads->orientations,
                           stream ? (hipStream_t)stream : reads->ctx->stream, reads, &same, &sz);
}

// device buffer of at least `want` bytes; an existing one that is large enough stays (a refill allocates nothing)
static hipError_t reads_buffer(void** p, uint64_t* cap, uint64_t want, bool generous)
{
    if (*p && *cap >= want)
        return hipSuccess;
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const uint64_t take = generous ? want + want / 8 : want;
    const hipError_t e = hipMalloc(p, take);
    if (e == hipSuccess)
        *cap = take;
    return e;
}

static int reads_upload_on(wamd_ctx* ctx, const uint8_t* data, const uint64_t* offsets, const uint32_t* lens, uint32_t n,
                           int encoding, int orientations, hipStream_t s, wamd_reads* reuse, wamd_reads** out,
                           const ReadSizes* stated)
{
    if (!ctx || !out || (n && (!data || !offsets || !lens)) ||
        (encoding != WAMD_READS_NIBBLE && encoding != WAMD_READS_ASCII) || orientations < 1 ||
        orientations > 3) {
        set_error("wamd_reads_upload: bad argument");
        return WAMD_ERR_INVALID_ARG;
    }
    WAMD_HIP(hipSetDevice(ctx->device));

    uint32_t max_len = 0;
    uint64_t raw_end = 0;
    uint64_t acc = 0;     // bytes of all reads' codes (where each read's start: a scan on the device)
    uint32_t min_len = n ? 0xffffffffu : 1u;
    const uint32_t nib = encoding == WAMD_READS_NIBBLE ? 1u : 0u;
    for (uint32_t i = 0; i < n && !stated; ++i) {     // (no early exit inside: the loop vectorises)
        const uint32_t len = lens[i];
        max_len = len > max_len ? len : max_len;
        min_len = len < min_len ? len : min_len;
        const uint64_t end = offsets[i] + (nib ? (len + 1) / 2 : len);
        raw_end = end > raw_end ? end : raw_end;
        acc += ((uint64_t)(len + 1) / 2 + 3) / 4 * 4;
    }
    if (stated) {     // wamd_reads_refill_async: no host pass (half a millisecond per 2^19 reads); the device checks
        max_len = stated->max_len;
        raw_end = stated->raw_end;
        acc = stated->acc;
        min_len = 1;
    }
    if (min_len == 0 || max_len > WAMD_MAX_READ_LEN) {
        uint32_t i = 0;
        while (i < n && lens[i] != 0 && lens[i] <= WAMD_MAX_READ_LEN)
            ++i;
        set_error("read %u has length %u (allowed 1..%u)", i, i < n ? lens[i] : 0u, WAMD_MAX_READ_LEN);
        return WAMD_ERR_TOO_LONG;
    }

    wamd_reads* r = reuse;
    if (!r) {
        r = new wamd_reads();
        memset(r, 0, sizeof(*r));
    }
    r->ctx = ctx;
    r->n = n;
    r->encoding = encoding;
    r->orientations = orientations;
    r->max_len = max_len;
    r->words = max_len ? (max_len - 1) / 64 + 1 : 1;
    r->words32 = max_len ? (max_len - 1) / 32 + 1 : 1;
    r->pm32_stride = r->words32 <= 4 ? (r->words32 <= 2 ? r->words32 : 4u) : (r->words32 + 7u) / 8u * 8u;
    r->codes_bytes = acc;
    r->pm_state = 0;     // the 64-bit pattern vectors belong to the reads that were here before
    const uint32_t n_orient = orientations == 3 ? 2 : 1;
    int rc = WAMD_OK;
    do {
        hipError_t e;
        const bool gen = reuse != nullptr;
        const uint64_t pm_bytes = (uint64_t)n * n_orient * 4 * r->words * sizeof(uint64_t);
        const uint64_t pm32_bytes = (uint64_t)n * n_orient * 4 * r->pm32_stride * sizeof(uint32_t);
        // (the raw reads stay with the batch until it is freed: releasing device memory waits for the whole device,
        //  and an upload that runs beside another bin's kernels must not)
        if ((e = reads_buffer((void**)&r->d_codes, &r->cap[0], acc + 64, gen)) != hipSuccess ||
            (e = reads_buffer((void**)&r->d_code_off, &r->cap[1], ((uint64_t)n + 1) * sizeof(uint64_t), gen)) != hipSuccess ||
            (e = reads_buffer((void**)&r->d_lens, &r->cap[2], ((uint64_t)n + 1) * sizeof(uint32_t), gen)) != hipSuccess ||
            (e = reads_buffer((void**)&r->d_pm, &r->cap[3], pm_bytes + 64, gen)) != hipSuccess ||
            (e = reads_buffer((void**)&r->d_pm32, &r->cap[4], pm32_bytes + 64, gen)) != hipSuccess ||
            (e = reads_buffer((void**)&r->d_meta, &r->cap[5], ((uint64_t)n * n_orient + 1) * sizeof(uint32_t), gen)) != hipSuccess ||
            (e = reads_buffer((void**)&r->d_raw, &r->cap[6], raw_end + 64, gen)) != hipSuccess ||
            (e = reads_buffer((void**)&r->d_raw_off, &r->cap[7], ((uint64_t)n + 1) * sizeof(uint64_t), gen)) != hipSuccess) {
            rc = hip_fail(e, "hipMalloc(reads)", __FILE__, __LINE__);
            break;
        }
        if (!r->pm_event && (e = hipEventCreateWithFlags(&r->pm_event, hipEventDisableTiming)) != hipSuccess) {
            rc = hip_fail(e, "hipEventCreate(reads)", __FILE__, __LINE__);
            break;
        }
        if (n == 0)
            break;
        if ((e = hipMemcpyAsync(r->d_raw, data, raw_end, hipMemcpyHostToDevice, s)) != hipSuccess ||
            (e = hipMemcpyAsync(r->d_raw_off, offsets, (size_t)n * sizeof(uint64_t),
                                hipMemcpyHostToDevice, s)) != hipSuccess ||
            (e = hipMemcpyAsync(r->d_lens, lens, (size_t)n * sizeof(uint32_t),
                                hipMemcpyHostToDevice, s)) != hipSuccess) {
            rc = hip_fail(e, "reads upload", __FILE__, __LINE__);
            break;
        }
        rc = launch_reads_prepare(ctx, r, r->d_raw, r->d_raw_off, s, stated ? raw_end : 0);
        if (rc != WAMD_OK)
            break;
        if (stated)     // returns with its work enqueued: the caller's buffers stay untouched until `s` has passed it
            break;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) {
            rc = hip_fail(e, "reads prepare", __FILE__, __LINE__);
            break;
        }
    } while (0);
    if (rc != WAMD_OK) {
        if (!reuse)
            wamd_reads_free(r);
        else
            r->n = 0;     // the batch stays the caller's, empty
        return rc;
    }
    *out = r;
    return WAMD_OK;
}

int wamd_reads_free(wamd_reads* r)
{
    if (!r)
        return WAMD_OK;
    (void)hipSetDevice(r->ctx->device);
    if (r->d_codes) (void)hipFree(r->d_codes);
    if (r->d_code_off) (void)hipFree(r->d_code_off);
    if (r->d_lens) (void)hipFree(r->d_lens);
    if (r->d_pm) (void)hipFree(r->d_pm);
    if (r->d_pm32) (void)hipFree(r->d_pm32);
    if (r->d_meta) (void)hipFree(r->d_meta);
    if (r->d_raw) (void)hipFree(r->d_raw);
    if (r->d_raw_off) (void)hipFree(r->d_raw_off);
    if (r->pm_event) (void)hipEventDestroy(r->pm_event);
    delete r;
    return WAMD_OK;
}

uint32_t wamd_reads_count(const wamd_reads* r) { return r ? r->n : 0; }

// ---- host-pointer entry points: stage in, launch, stage out ----------------------

static int check_batch(wamd_ctx* ctx, const wamd_reads* reads, const void* a, const void* b,
                       const char* who)
{
    if (!ctx || !reads || reads->ctx != ctx || !a || !b) {
        set_error("%s: null argument or reads of another context", who);
        return WAMD_ERR_INVALID_ARG;
    }
    return WAMD_OK;
}

int wamd_verify_distance_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_dist_task* d_tasks,
                             uint32_t n, int last_symbol, wamd_dist_result* d_results, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, d_tasks, d_results, "wamd_verify_distance_dev");
    if (rc)
        return rc;
    if (last_symbol < -1 || last_symbol > 15)
        return WAMD_ERR_INVALID_ARG;
    return launch_distance(ctx, reads, d_tasks, n, last_symbol, d_results, ctx->distance_mode,
                           stream ? (hipStream_t)stream : ctx->stream);
}

int wamd_verify_distance(wamd_ctx* ctx, const wamd_reads* reads, const wamd_dist_task* tasks,
                         uint32_t n, int last_symbol, wamd_dist_result* results)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_verify_distance");
    if (rc)
        return rc;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    const uint64_t tb = (uint64_t)n * sizeof(wamd_dist_task), rb = (uint64_t)n * sizeof(wamd_dist_result);
    rc = ensure_stage(ctx, tb + rb, 0);
    if (rc)
        return rc;
    auto* d_tasks = (wamd_dist_task*)ctx->d_stage;
    auto* d_res = (wamd_dist_result*)((char*)ctx->d_stage + tb);
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    rc = wamd_verify_distance_dev(ctx, reads, d_tasks, n, last_symbol, d_res, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    return rc ? rc : flags_to_status(flags);
}

int wamd_levdiag_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_levdiag_task* d_tasks,
                     uint32_t n, uint32_t* d_results, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, d_tasks, d_results, "wamd_levdiag_dev");
    if (rc)
        return rc;
    return launch_levdiag(ctx, reads, d_tasks, n, d_results, stream ? (hipStream_t)stream : ctx->stream);
}

int wamd_levdiag(wamd_ctx* ctx, const wamd_reads* reads, const wamd_levdiag_task* tasks, uint32_t n,
                 uint32_t* results)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_levdiag");
    if (rc)
        return rc;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    const uint64_t tb = (uint64_t)n * sizeof(wamd_levdiag_task), rb = (uint64_t)n * sizeof(uint32_t);
    rc = ensure_stage(ctx, tb + rb, 0);
    if (rc)
        return rc;
    auto* d_tasks = (wamd_levdiag_task*)ctx->d_stage;
    auto* d_res = (uint32_t*)((char*)ctx->d_stage + tb);
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    rc = wamd_levdiag_dev(ctx, reads, d_tasks, n, d_res, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    return rc ? rc : flags_to_status(flags);
}

int wamd_cf_filter_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_cf_task* d_tasks, uint32_t n,
                       uint32_t* d_results, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, d_tasks, d_results, "wamd_cf_filter_dev");
    if (rc)
        return rc;
    return launch_cf_filter(ctx, reads, d_tasks, n, d_results, stream ? (hipStream_t)stream : ctx->stream);
}

int wamd_cf_filter(wamd_ctx* ctx, const wamd_reads* reads, const wamd_cf_task* tasks, uint32_t n,
                   uint32_t* results)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_cf_filter");
    if (rc)
        return rc;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    const uint64_t tb = (uint64_t)n * sizeof(wamd_cf_task), rb = (uint64_t)n * sizeof(uint32_t);
    rc = ensure_stage(ctx, tb + rb, 0);
    if (rc)
        return rc;
    auto* d_tasks = (wamd_cf_task*)ctx->d_stage;
    auto* d_res = (uint32_t*)((char*)ctx->d_stage + tb);
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    rc = wamd_cf_filter_dev(ctx, reads, d_tasks, n, d_res, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    return rc ? rc : flags_to_status(flags);
}

uint64_t wamd_filter_verify_workspace_bytes(uint32_t n) { return filter_verify_workspace_bytes(n); }

static int check_filter_args(wamd_ctx* ctx, const wamd_reads* reads, const void* d_cf, const void* d_tasks, const void* d_results,
                             const void* d_ws, uint64_t ws_bytes, uint32_t n, const char* who)
{
    int rc = check_batch(ctx, reads, d_tasks, d_results, who);
    if (rc)
        return rc;
    if (!d_cf || !d_ws || ((uintptr_t)d_ws & 255u)) {
        set_error("%s: null CF tasks or workspace, or a workspace that is not 256-byte aligned", who);
        return WAMD_ERR_INVALID_ARG;
    }
    if (ws_bytes < filter_verify_workspace_bytes(n)) {
        set_error("%s: workspace of %llu bytes, wamd_filter_verify_workspace_bytes(%u) = %llu", who,
                  (unsigned long long)ws_bytes, n, (unsigned long long)filter_verify_workspace_bytes(n));
        return WAMD_ERR_INVALID_ARG;
    }
    return WAMD_OK;
}

int wamd_filter_verify_distance_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_cf_task* d_cf_tasks,
                                    const wamd_dist_task* d_tasks, uint32_t n, int last_symbol, wamd_dist_result* d_results,
                                    void* d_workspace, uint64_t workspace_bytes, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_filter_args(ctx, reads, d_cf_tasks, d_tasks, d_results, d_workspace, workspace_bytes, n,
                               "wamd_filter_verify_distance_dev");
    if (rc)
        return rc;
    if (last_symbol < -1 || last_symbol > 15)
        return WAMD_ERR_INVALID_ARG;
    return launch_filter_distance(ctx, reads, d_cf_tasks, d_tasks, n, last_symbol, d_results, d_workspace,
                                  stream ? (hipStream_t)stream : ctx->stream);
}

int wamd_filter_levdiag_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_cf_task* d_cf_tasks,
                            const wamd_levdiag_task* d_tasks, uint32_t n, uint32_t* d_results, void* d_workspace,
                            uint64_t workspace_bytes, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_filter_args(ctx, reads, d_cf_tasks, d_tasks, d_results, d_workspace, workspace_bytes, n,
                               "wamd_filter_levdiag_dev");
    if (rc)
        return rc;
    return launch_filter_levdiag(ctx, reads, d_cf_tasks, d_tasks, n, d_results, d_workspace,
                                 stream ? (hipStream_t)stream : ctx->stream);
}

uint64_t wamd_sa_candidates_workspace_bytes(uint32_t n_candidates, uint32_t n_reads)
{
    return sa_candidates_workspace_bytes(n_candidates, n_reads);
}

static int check_sa_batch(wamd_ctx* ctx, const wamd_reads* reads, const wamd_sa_batch* b, const void* counts,
                          const void* positions, const void* errors, const char* who)
{
    int rc = check_batch(ctx, reads, counts, positions, who);
    if (rc)
        return rc;
    if (!b || !errors || (((uintptr_t)counts | (uintptr_t)positions | (uintptr_t)errors) & 3u)) {
        set_error("%s: null batch or output, or counts / positions / errors not 4-byte aligned", who);
        return WAMD_ERR_INVALID_ARG;
    }
    if (b->verifier > WAMD_SA_VERIFIER_LEVDIAG || b->fixed_left >= b->fixed_right ||
        (b->min_read_len && b->fixed_right > b->min_read_len) || b->max_mismatches > 254 ||
        (uint64_t)b->read_first + b->n_reads > reads->n) {
        set_error("%s: bad verifier, anchor [%u, %u) outside min_read_len %u, max_mismatches %u above 254, or reads "
                  "[%u, +%u) beyond the %u uploaded", who, b->fixed_left, b->fixed_right, b->min_read_len,
                  b->max_mismatches, b->read_first, b->n_reads, reads->n);
        return WAMD_ERR_INVALID_ARG;
    }
    return WAMD_OK;
}

int wamd_verify_sa_candidates_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_sa_batch* batch,
                                  const uint16_t* d_counts, const uint8_t* d_read_max_mismatches,
                                  const uint32_t* d_sa_positions, uint32_t n_candidates,
                                  uint8_t* d_errors, uint32_t* d_end_positions, void* d_workspace,
                                  uint64_t workspace_bytes, void* stream)
{
    if (n_candidates == 0)
        return WAMD_OK;
    int rc = check_sa_batch(ctx, reads, batch, d_counts, d_sa_positions, d_errors, "wamd_verify_sa_candidates_dev");
    if (rc)
        return rc;
    if (!d_workspace || ((uintptr_t)d_workspace & 255u) ||
        workspace_bytes < sa_candidates_workspace_bytes(n_candidates, batch->n_reads)) {
        set_error("wamd_verify_sa_candidates_dev: workspace missing, not 256-byte aligned or below "
                  "wamd_sa_candidates_workspace_bytes(%u, %u)", n_candidates, batch->n_reads);
        return WAMD_ERR_INVALID_ARG;
    }
    return launch_sa_candidates(ctx, reads, batch, d_counts, d_read_max_mismatches, d_sa_positions, n_candidates, d_errors,
                                d_end_positions,
                                d_workspace, stream ? (hipStream_t)stream : ctx->stream);
}

int wamd_verify_sa_candidates(wamd_ctx* ctx, const wamd_reads* reads, const wamd_sa_batch* batch, const uint16_t* counts,
                              const uint8_t* read_max_mismatches, const uint32_t* sa_positions, uint32_t n_candidates,
                              uint8_t* errors, uint32_t* end_positions)
{
    if (n_candidates == 0)
        return WAMD_OK;
    int rc = check_sa_batch(ctx, reads, batch, counts, sa_positions, errors, "wamd_verify_sa_candidates");
    if (rc)
        return rc;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    for (hipStream_t& ps : ctx->pipe)
        if (!ps)
            WAMD_HIP(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    // chunks of whole reads: about 1 M candidates, at most 1 M reads
    constexpr uint32_t CHUNK_C = 1u << 20, CHUNK_R = 1u << 20, CMAX = CHUNK_C + 2u * 65535u;
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    const uint32_t cmax = n_candidates < CMAX ? n_candidates : CMAX;
    const uint32_t rmax = batch->n_reads < CHUNK_R ? batch->n_reads : CHUNK_R;
    const uint64_t o_counts = 0, o_k = o_counts + up((uint64_t)rmax * 4), o_pos = o_k + up(rmax),
                   o_err = o_pos + up((uint64_t)cmax * 4),
                   o_end = o_err + up(cmax), o_ws = o_end + (end_positions ? up((uint64_t)cmax * 4) : 0),
                   slot_bytes = o_ws + up(sa_candidates_workspace_bytes(cmax, rmax));
    // chunks in flight: with two, the copies of chunk i + 2 wait for everything of chunk i (same stream, same slot) and
    // the kernels run dry between them whenever the link is shared with a reads upload; four keep the link busy
    constexpr uint32_t DEPTH = sizeof(ctx->pipe) / sizeof(ctx->pipe[0]);
    const uint32_t n_chunks_est = (n_candidates + CHUNK_C - 1) / CHUNK_C;
    const uint32_t depth = n_chunks_est < 2 ? 2u : (n_chunks_est < DEPTH ? n_chunks_est : DEPTH);
    rc = ensure_stage(ctx, depth * slot_bytes, 0);
    if (rc)
        return rc;
    uint32_t r0 = 0, c0 = 0, chunk = 0;
    uint64_t seen = 0;
    // (an error inside the loop must not return past the synchronisation below: copies into the caller's buffers
    //  may still be on their way)
#define PIPE_HIP(call)                                                                  \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess && !rc)                                                    \
            rc = ::wamd::hip_fail(e_, #call, __FILE__, __LINE__);                       \
    } while (0)
    while (r0 < batch->n_reads) {
        uint32_t r1 = r0, c = 0;
        while (r1 < batch->n_reads && r1 - r0 < CHUNK_R && c < CHUNK_C) {
            c += (uint32_t)counts[2 * (uint64_t)r1] + counts[2 * (uint64_t)r1 + 1];
            ++r1;
        }
        seen += c;
        if (seen > n_candidates)
            break;
        if (c) {
            char* slot = (char*)ctx->d_stage + (uint64_t)(chunk % depth) * slot_bytes;
            hipStream_t st = ctx->pipe[chunk % depth];
            wamd_sa_batch b = *batch;
            b.read_first = batch->read_first + r0;
            b.n_reads = r1 - r0;
            PIPE_HIP(hipMemcpyAsync(slot + o_counts, counts + 2 * (uint64_t)r0, (uint64_t)(r1 - r0) * 4, hipMemcpyHostToDevice, st));
            PIPE_HIP(hipMemcpyAsync(slot + o_pos, sa_positions + c0, (uint64_t)c * 4, hipMemcpyHostToDevice, st));
            if (read_max_mismatches)
                PIPE_HIP(hipMemcpyAsync(slot + o_k, read_max_mismatches + r0, r1 - r0, hipMemcpyHostToDevice, st));
            if (rc)
                break;
            rc = launch_sa_candidates(ctx, reads, &b, (const uint16_t*)(slot + o_counts),
                                      read_max_mismatches ? (const uint8_t*)(slot + o_k) : nullptr, (const uint32_t*)(slot + o_pos), c,
                                      (uint8_t*)(slot + o_err), end_positions ? (uint32_t*)(slot + o_end) : nullptr,
                                      slot + o_ws, st);
            if (rc)
                break;
            PIPE_HIP(hipMemcpyAsync(errors + c0, slot + o_err, c, hipMemcpyDeviceToHost, st));
            if (end_positions)
                PIPE_HIP(hipMemcpyAsync(end_positions + c0, slot + o_end, (uint64_t)c * 4, hipMemcpyDeviceToHost, st));
            if (rc)
                break;
            ++chunk;
        }
        r0 = r1;
        c0 += c;
    }
    for (hipStream_t ps : ctx->pipe)
        PIPE_HIP(hipStreamSynchronize(ps));
#undef PIPE_HIP
    if (rc)
        return rc;
    if (seen != n_candidates) {
        set_error("wamd_verify_sa_candidates: the counts add up to %llu candidates, the call says %u",
                  (unsigned long long)seen, n_candidates);
        return WAMD_ERR_INVALID_ARG;
    }
    uint32_t flags = 0;
    rc = fetch_flags(ctx, ctx->stream, &flags);
    return rc ? rc : flags_to_status(flags);
}

int wamd_host_alloc(void** out, uint64_t bytes)
{
    if (!out)
        return WAMD_ERR_INVALID_ARG;
    *out = nullptr;
    WAMD_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return WAMD_OK;
}

int wamd_host_free(void* p)
{
    if (p)
        WAMD_HIP(hipHostFree(p));
    return WAMD_OK;
}

uint64_t wamd_align_workspace_bytes(const wamd_reads* reads, uint32_t n)
{
    return reads ? align_workspace_bytes(reads, n) : 0;
}

uint64_t wamd_align_workspace_bytes_for(const wamd_reads* reads, uint32_t n, uint32_t max_ed)
{
    return reads ? align_workspace_bytes(reads, n, max_ed) : 0;
}

int wamd_align_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_dist_task* d_tasks, uint32_t n,
                   int last_symbol, int refine, wamd_align_result* d_results, uint8_t* d_cigars,
                   uint32_t cigar_stride, void* d_workspace, uint64_t workspace_bytes, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, d_tasks, d_results, "wamd_align_dev");
    if (rc)
        return rc;
    if (!d_cigars || !d_workspace || workspace_bytes < align_workspace_bytes(reads, n) ||
        cigar_stride < 16 || last_symbol < -1 || last_symbol > 15) {
        set_error("wamd_align_dev: missing buffer, workspace too small or bad argument");
        return WAMD_ERR_INVALID_ARG;
    }
    return launch_align(ctx, reads, d_tasks, n, last_symbol, refine, d_results, d_cigars,
                        cigar_stride, d_workspace, workspace_bytes,
                        stream ? (hipStream_t)stream : ctx->stream);
}

int wamd_align(wamd_ctx* ctx, const wamd_reads* reads, const wamd_dist_task* tasks, uint32_t n,
               int last_symbol, int refine, wamd_align_result* results, uint8_t* cigars,
               uint32_t cigar_stride)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_align");
    if (rc)
        return rc;
    if (!cigars)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    const uint64_t tb = (uint64_t)n * sizeof(wamd_dist_task);
    const uint64_t rb = (uint64_t)n * sizeof(wamd_align_result);
    const uint64_t cb = (uint64_t)n * cigar_stride;
    uint32_t max_ed = 0;   // the host form sees the tasks: room for the wide classes when they can occur
    for (uint32_t i = 0; i < n; ++i)
        max_ed = tasks[i].max_ed > max_ed ? tasks[i].max_ed : max_ed;
    const uint64_t wb = align_workspace_bytes(reads, n, max_ed);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    rc = ensure_stage(ctx, up(tb) + up(rb) + up(cb) + up(wb), 0);
    if (rc)
        return rc;
    char* base = (char*)ctx->d_stage;
    auto* d_tasks = (wamd_dist_task*)base;
    auto* d_res = (wamd_align_result*)(base + up(tb));
    auto* d_cig = (uint8_t*)(base + up(tb) + up(rb));
    void* d_ws = base + up(tb) + up(rb) + up(cb);
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    rc = wamd_align_dev(ctx, reads, d_tasks, n, last_symbol, refine, d_res, d_cig, cigar_stride,
                        d_ws, wb, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(cigars, d_cig, cb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    return rc ? rc : flags_to_status(flags);
}

// ---- long-indel matcher, its ext-CIGAR, boundary clipping, wire records ---------------

static hipStream_t pick(wamd_ctx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

int wamd_indel_match_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_indel_task* d_tasks, uint32_t n,
                         const wamd_indel_params* params, wamd_mapping* d_results, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, d_tasks, d_results, "wamd_indel_match_dev");
    if (rc)
        return rc;
    if (!params)
        return WAMD_ERR_INVALID_ARG;
    return launch_indel_match(ctx, reads, d_tasks, n, params, d_results, pick(ctx, stream));
}

int wamd_indel_match(wamd_ctx* ctx, const wamd_reads* reads, const wamd_indel_task* tasks, uint32_t n,
                     const wamd_indel_params* params, wamd_mapping* results)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_indel_match");
    if (rc)
        return rc;
    if (!params)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    const uint64_t tb = (uint64_t)n * sizeof(wamd_indel_task), rb = (uint64_t)n * sizeof(wamd_mapping);
    rc = ensure_stage(ctx, tb + rb + 256, 0);
    if (rc)
        return rc;
    auto* d_tasks = (wamd_indel_task*)ctx->d_stage;
    auto* d_res = (wamd_mapping*)((char*)ctx->d_stage + (tb + 255) / 256 * 256);
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    wamd_indel_params prm = *params;
    prm.max_indel_bound = 1;
    for (uint32_t i = 0; i < n; ++i)
        prm.max_indel_bound = tasks[i].max_indel_size > prm.max_indel_bound ? tasks[i].max_indel_size : prm.max_indel_bound;
    rc = wamd_indel_match_dev(ctx, reads, d_tasks, n, &prm, d_res, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    return rc ? rc : flags_to_status(flags);
}

int wamd_indel_ext_cigar_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_indel_cigar_task* d_tasks,
                             uint32_t n, wamd_indel_cigar_result* d_results, uint8_t* d_cigars,
                             uint32_t cigar_stride, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, d_tasks, d_results, "wamd_indel_ext_cigar_dev");
    if (rc)
        return rc;
    if (!d_cigars || (cigar_stride & 3u) || ((uintptr_t)d_cigars & 3u) || cigar_stride < 16)
        return WAMD_ERR_INVALID_ARG;
    return launch_indel_ext_cigar(ctx, reads, d_tasks, n, d_results, d_cigars, cigar_stride, pick(ctx, stream));
}

int wamd_indel_ext_cigar(wamd_ctx* ctx, const wamd_reads* reads, const wamd_indel_cigar_task* tasks, uint32_t n,
                         wamd_indel_cigar_result* results, uint8_t* cigars, uint32_t cigar_stride)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_indel_ext_cigar");
    if (rc)
        return rc;
    if (!cigars)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    const uint64_t tb = (uint64_t)n * sizeof(wamd_indel_cigar_task), rb = (uint64_t)n * sizeof(wamd_indel_cigar_result);
    const uint64_t cb = (uint64_t)n * cigar_stride;
    rc = ensure_stage(ctx, up(tb) + up(rb) + up(cb), 0);
    if (rc)
        return rc;
    char* base = (char*)ctx->d_stage;
    auto* d_tasks = (wamd_indel_cigar_task*)base;
    auto* d_res = (wamd_indel_cigar_result*)(base + up(tb));
    auto* d_cig = (uint8_t*)(base + up(tb) + up(rb));
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    rc = wamd_indel_ext_cigar_dev(ctx, reads, d_tasks, n, d_res, d_cig, cigar_stride, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(cigars, d_cig, cb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    return rc ? rc : flags_to_status(flags);
}

int wamd_clip_boundary_dev(wamd_ctx* ctx, uint32_t n, wamd_clip_record* d_records, uint8_t* d_cigars,
                           uint32_t cigar_stride, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !d_records || !d_cigars || cigar_stride == 0)
        return WAMD_ERR_INVALID_ARG;
    return launch_clip_boundary(ctx, n, d_records, d_cigars, cigar_stride, pick(ctx, stream));
}

int wamd_clip_boundary(wamd_ctx* ctx, uint32_t n, wamd_clip_record* records, uint8_t* cigars, uint32_t cigar_stride)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !records || !cigars || cigar_stride == 0)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    const uint64_t rb = (uint64_t)n * sizeof(wamd_clip_record), cb = (uint64_t)n * cigar_stride;
    int rc = ensure_stage(ctx, up(rb) + up(cb), 0);
    if (rc)
        return rc;
    auto* d_rec = (wamd_clip_record*)ctx->d_stage;
    auto* d_cig = (uint8_t*)ctx->d_stage + up(rb);
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_rec, records, rb, hipMemcpyHostToDevice, s));
    WAMD_HIP(hipMemcpyAsync(d_cig, cigars, cb, hipMemcpyHostToDevice, s));
    rc = wamd_clip_boundary_dev(ctx, n, d_rec, d_cig, cigar_stride, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(records, d_rec, rb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(cigars, d_cig, cb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipStreamSynchronize(s));
    return WAMD_OK;
}

int wamd_clip_illegal_dev(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* d_tasks, wamd_clip_record* d_records,
                          uint8_t* d_cigars, uint32_t cigar_stride, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !d_tasks || !d_records || !d_cigars || cigar_stride < 2)
        return WAMD_ERR_INVALID_ARG;
    return launch_clip_illegal(ctx, n, d_tasks, d_records, d_cigars, cigar_stride, pick(ctx, stream));
}

int wamd_clip_overlapping_pairs_dev(wamd_ctx* ctx, uint32_t n_pairs, const wamd_clip_pair_task* d_tasks,
                                    wamd_clip_record* d_records, uint8_t* d_cigars, uint32_t cigar_stride,
                                    void* stream)
{
    if (n_pairs == 0)
        return WAMD_OK;
    if (!ctx || !d_tasks || !d_records || !d_cigars || cigar_stride < 2 || n_pairs > 0x7fffffffu)
        return WAMD_ERR_INVALID_ARG;
    return launch_clip_pairs(ctx, n_pairs, d_tasks, d_records, d_cigars, cigar_stride, pick(ctx, stream));
}

int wamd_clip_illegal(wamd_ctx* ctx, uint32_t n, const wamd_clip_illegal_task* tasks, wamd_clip_record* records,
                      uint8_t* cigars, uint32_t cigar_stride)
{
    return clip_pass_host(ctx, n, n, tasks, records, cigars, cigar_stride, "wamd_clip_illegal",
                          [&](const wamd_clip_illegal_task* dt, wamd_clip_record* dr, uint8_t* dc, hipStream_t s) {
                              return wamd_clip_illegal_dev(ctx, n, dt, dr, dc, cigar_stride, s);
                          });
}

int wamd_clip_overlapping_pairs(wamd_ctx* ctx, uint32_t n_pairs, const wamd_clip_pair_task* tasks,
                                wamd_clip_record* records, uint8_t* cigars, uint32_t cigar_stride)
{
    if (n_pairs > 0x7fffffffu)
        return WAMD_ERR_INVALID_ARG;
    return clip_pass_host(ctx, n_pairs, 2 * n_pairs, tasks, records, cigars, cigar_stride, "wamd_clip_overlapping_pairs",
                          [&](const wamd_clip_pair_task* dt, wamd_clip_record* dr, uint8_t* dc, hipStream_t s) {
                              return wamd_clip_overlapping_pairs_dev(ctx, n_pairs, dt, dr, dc, cigar_stride, s);
                          });
}

int wamd_softclip_match_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_softclip_task* d_tasks, uint32_t n,
                            const wamd_softclip_params* params, uint32_t max_window, wamd_softclip_result* d_results,
                            uint8_t* d_cigars, uint32_t cigar_stride, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !reads || !d_tasks || !params || !d_results || !d_cigars || params->max_query_len == 0 ||
        reads->ctx != ctx) {
        set_error("wamd_softclip_match_dev: missing argument, or reads of another context");
        return WAMD_ERR_INVALID_ARG;
    }
    return launch_softclip_match(ctx, reads, d_tasks, n, params, max_window, d_results, d_cigars, cigar_stride,
                                 pick(ctx, stream));
}

int wamd_softclip_match(wamd_ctx* ctx, const wamd_reads* reads, const wamd_softclip_task* tasks, uint32_t n,
                        const wamd_softclip_params* params, wamd_softclip_result* results, uint8_t* cigars,
                        uint32_t cigar_stride)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_softclip_match");
    if (rc)
        return rc;
    if (!params || !cigars)
        return WAMD_ERR_INVALID_ARG;
    uint32_t max_window = 0;
    for (uint32_t i = 0; i < n; ++i)
        max_window = tasks[i].max_distance_in_ref > max_window ? tasks[i].max_distance_in_ref : max_window;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    const uint64_t tb = (uint64_t)n * sizeof(wamd_softclip_task), rb = (uint64_t)n * sizeof(wamd_softclip_result),
                   cb = (uint64_t)n * cigar_stride;
    rc = ensure_stage(ctx, up(tb) + up(rb) + up(cb), 0);
    if (rc)
        return rc;
    char* base = (char*)ctx->d_stage;
    auto* d_tasks = (wamd_softclip_task*)base;
    auto* d_res = (wamd_softclip_result*)(base + up(tb));
    auto* d_cig = (uint8_t*)(base + up(tb) + up(rb));
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    rc = wamd_softclip_match_dev(ctx, reads, d_tasks, n, params, max_window, d_res, d_cig, cigar_stride, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(cigars, d_cig, cb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    return rc ? rc : flags_to_status(flags);
}

int wamd_sam_cigar_dev(wamd_ctx* ctx, uint32_t n, const uint8_t* d_ext_cigars, uint32_t ext_stride,
                       uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz, uint32_t mdz_stride,
                       uint32_t* d_cigar_bin, uint32_t bin_stride, wamd_sam_cigar_len* d_lens, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !d_ext_cigars || !d_cigars || !d_mdz || !d_lens || ext_stride == 0 || cigar_stride < 16 ||
        mdz_stride < 16 || (d_cigar_bin && bin_stride == 0)) {
        set_error("wamd_sam_cigar_dev: missing buffer or stride too small (outputs need 16 bytes at least)");
        return WAMD_ERR_INVALID_ARG;
    }
    return launch_sam_cigar(ctx, n, d_ext_cigars, ext_stride, d_cigars, cigar_stride, d_mdz, mdz_stride,
                            d_cigar_bin, bin_stride, d_lens, pick(ctx, stream));
}

int wamd_sam_cigar(wamd_ctx* ctx, uint32_t n, const uint8_t* ext_cigars, uint32_t ext_stride, uint8_t* cigars,
                   uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride, uint32_t* cigar_bin,
                   uint32_t bin_stride, wamd_sam_cigar_len* lens)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !ext_cigars || !cigars || !mdz || !lens)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    const uint64_t eb = (uint64_t)n * ext_stride, cb = (uint64_t)n * cigar_stride, mb = (uint64_t)n * mdz_stride,
                   bb = cigar_bin ? (uint64_t)n * bin_stride * sizeof(uint32_t) : 0,
                   lb = (uint64_t)n * sizeof(wamd_sam_cigar_len);
    int rc = ensure_stage(ctx, up(eb) + up(cb) + up(mb) + up(bb) + up(lb), 0);
    if (rc)
        return rc;
    char* base = (char*)ctx->d_stage;
    auto* d_ext = (uint8_t*)base;
    auto* d_cig = (uint8_t*)(base + up(eb));
    auto* d_mdz = (uint8_t*)(base + up(eb) + up(cb));
    auto* d_bin = cigar_bin ? (uint32_t*)(base + up(eb) + up(cb) + up(mb)) : nullptr;
    auto* d_len = (wamd_sam_cigar_len*)(base + up(eb) + up(cb) + up(mb) + up(bb));
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_ext, ext_cigars, eb, hipMemcpyHostToDevice, s));
    rc = wamd_sam_cigar_dev(ctx, n, d_ext, ext_stride, d_cig, cigar_stride, d_mdz, mdz_stride, d_bin, bin_stride,
                            d_len, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(cigars, d_cig, cb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(mdz, d_mdz, mb, hipMemcpyDeviceToHost, s));
    if (cigar_bin)
        WAMD_HIP(hipMemcpyAsync(cigar_bin, d_bin, bb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(lens, d_len, lb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    if (!rc && (flags & FLAG_CIGAR)) {
        set_error("wamd_sam_cigar: an output does not fit its stride");
        return WAMD_ERR_INVALID_ARG;
    }
    return rc ? rc : flags_to_status(flags);
}

int wamd_serialize_mappings_dev(wamd_ctx* ctx, uint32_t n, const wamd_mapping* d_mappings,
                                const uint8_t* d_direction, uint8_t* d_out, uint8_t* d_sizes, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !d_mappings || !d_direction || !d_out || !d_sizes || ((uintptr_t)d_out & 3u))
        return WAMD_ERR_INVALID_ARG;
    return launch_serialize(ctx, n, d_mappings, d_direction, d_out, d_sizes, pick(ctx, stream));
}

int wamd_serialize_mappings(wamd_ctx* ctx, uint32_t n, const wamd_mapping* mappings, const uint8_t* direction,
                            uint8_t* out, uint8_t* sizes)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !mappings || !direction || !out || !sizes)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    const uint64_t mb = (uint64_t)n * sizeof(wamd_mapping), ob = (uint64_t)n * 12;
    int rc = ensure_stage(ctx, up(mb) + up(n) + up(ob) + up(n), 0);
    if (rc)
        return rc;
    char* base = (char*)ctx->d_stage;
    auto* d_m = (wamd_mapping*)base;
    auto* d_dir = (uint8_t*)(base + up(mb));
    auto* d_out = (uint8_t*)(base + up(mb) + up(n));
    auto* d_sz = (uint8_t*)(base + up(mb) + up(n) + up(ob));
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_m, mappings, mb, hipMemcpyHostToDevice, s));
    WAMD_HIP(hipMemcpyAsync(d_dir, direction, n, hipMemcpyHostToDevice, s));
    rc = wamd_serialize_mappings_dev(ctx, n, d_m, d_dir, d_out, d_sz, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(out, d_out, ob, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(sizes, d_sz, n, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipStreamSynchronize(s));
    return WAMD_OK;
}

uint64_t wamd_records_scratch_bytes(uint32_t n) { return records_scratch_bytes(n); }

int wamd_emit_lev_records_dev(wamd_ctx* ctx, uint32_t n, const wamd_dist_task* d_tasks,
                              const wamd_dist_result* d_results, const uint32_t* d_record_pos,
                              const uint8_t* d_direction, uint8_t* d_records, uint32_t* d_count, void* d_scratch,
                              void* stream)
{
    if (!ctx || !d_count)
        return WAMD_ERR_INVALID_ARG;
    if (n == 0) {
        WAMD_HIP(hipSetDevice(ctx->device));
        WAMD_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), pick(ctx, stream)));
        return WAMD_OK;
    }
    if (!d_tasks || !d_results || !d_record_pos || !d_direction || !d_records || !d_scratch ||
        ((uintptr_t)d_records & 3u) || ((uintptr_t)d_scratch & 3u))
        return WAMD_ERR_INVALID_ARG;
    return launch_emit_lev_records(ctx, n, d_tasks, d_results, d_record_pos, d_direction, d_records, d_count,
                                   d_scratch, pick(ctx, stream));
}

int wamd_commit_candidates_dev(wamd_ctx* ctx, uint32_t n, const wamd_commit_task* d_tasks, const wamd_mapping* d_indel,
                               wamd_mapping* d_committed, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    if (!ctx || !d_tasks || !d_committed)
        return WAMD_ERR_INVALID_ARG;
    return launch_commit(ctx, n, d_tasks, d_indel, d_committed, pick(ctx, stream));
}

uint64_t wamd_result_groups_workspace_bytes(uint32_t n_tasks, uint32_t n_reads)
{
    return result_groups_workspace_bytes(n_tasks, n_reads);
}

int wamd_emit_result_groups_dev(wamd_ctx* ctx, uint32_t n_tasks, const wamd_commit_task* d_tasks,
                                const wamd_mapping* d_committed, uint32_t n_reads, const uint32_t* d_read_first,
                                const uint64_t* d_read_ids, const wamd_group_params* params, uint8_t* d_out,
                                uint32_t out_capacity, uint32_t* d_out_bytes, uint32_t* d_read_off, void* d_workspace,
                                void* stream)
{
    if (!ctx || !d_out_bytes || !params)
        return WAMD_ERR_INVALID_ARG;
    if (n_reads == 0) {
        WAMD_HIP(hipSetDevice(ctx->device));
        WAMD_HIP(hipMemsetAsync(d_out_bytes, 0, sizeof(uint32_t), pick(ctx, stream)));
        return WAMD_OK;
    }
    if (!d_tasks || !d_committed || !d_read_first || !d_read_ids || !d_out || !d_workspace ||
        ((uintptr_t)d_workspace & 3u))
        return WAMD_ERR_INVALID_ARG;
    return launch_result_groups(ctx, n_tasks, d_tasks, d_committed, n_reads, d_read_first, d_read_ids, params, d_out,
                                out_capacity, d_out_bytes, d_read_off, d_workspace, pick(ctx, stream));
}

uint64_t wamd_finish_workspace_bytes(const wamd_reads* reads, uint32_t n_pairs)
{
    return reads ? finish_workspace_bytes(reads, n_pairs) : 0;
}

int wamd_finish_pairs_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_finish_task* d_tasks, uint32_t n_pairs,
                          const wamd_finish_params* params, wamd_finish_result* d_results, uint8_t* d_ext_cigars,
                          uint32_t ext_stride, uint8_t* d_cigars, uint32_t cigar_stride, uint8_t* d_mdz,
                          uint32_t mdz_stride, void* d_workspace, uint64_t workspace_bytes, void* stream)
{
    if (n_pairs == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, d_tasks, d_results, "wamd_finish_pairs_dev");
    if (rc)
        return rc;
    if (!params || !d_ext_cigars || !d_cigars || !d_mdz || !d_workspace || (ext_stride & 3u) || ext_stride < 16 ||
        ((uintptr_t)d_ext_cigars & 3u) || cigar_stride < 16 || mdz_stride < 16) {
        set_error("wamd_finish_pairs_dev: missing buffer, or a stride too small / ext_stride not a multiple of 4");
        return WAMD_ERR_INVALID_ARG;
    }
    return launch_finish_pairs(ctx, reads, d_tasks, n_pairs, params, d_results, d_ext_cigars, ext_stride, d_cigars,
                               cigar_stride, d_mdz, mdz_stride, d_workspace, workspace_bytes, pick(ctx, stream));
}

int wamd_finish_pairs(wamd_ctx* ctx, const wamd_reads* reads, const wamd_finish_task* tasks, uint32_t n_pairs,
                      const wamd_finish_params* params, wamd_finish_result* results, uint8_t* ext_cigars,
                      uint32_t ext_stride, uint8_t* cigars, uint32_t cigar_stride, uint8_t* mdz, uint32_t mdz_stride)
{
    if (n_pairs == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_finish_pairs");
    if (rc)
        return rc;
    if (!params || !ext_cigars || !cigars || !mdz || n_pairs > 0x3fffffffu)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    const uint64_t n = 2ull * n_pairs;
    const uint64_t tb = n * sizeof(wamd_finish_task), rb = n * sizeof(wamd_finish_result), eb = n * ext_stride,
                   cb = n * cigar_stride, mb = n * mdz_stride, wb = finish_workspace_bytes(reads, n_pairs);
    rc = ensure_stage(ctx, up(tb) + up(rb) + up(eb) + up(cb) + up(mb) + up(wb) + 256, 0);
    if (rc)
        return rc;
    char* base = (char*)(((uintptr_t)ctx->d_stage + 255) & ~(uintptr_t)255);
    auto* d_tasks = (wamd_finish_task*)base;
    auto* d_res = (wamd_finish_result*)(base + up(tb));
    auto* d_ext = (uint8_t*)(base + up(tb) + up(rb));
    auto* d_cig = d_ext + up(eb);
    auto* d_mdz = d_cig + up(cb);
    void* d_ws = d_mdz + up(mb);
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    rc = wamd_finish_pairs_dev(ctx, reads, d_tasks, n_pairs, params, d_res, d_ext, ext_stride, d_cig, cigar_stride,
                               d_mdz, mdz_stride, d_ws, wb, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(ext_cigars, d_ext, eb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(cigars, d_cig, cb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(mdz, d_mdz, mb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    if (!rc && (flags & FLAG_CIGAR)) {
        set_error("wamd_finish_pairs: an ext-CIGAR, CIGAR or MD:Z does not fit its stride");
        return WAMD_ERR_INVALID_ARG;
    }
    return rc ? rc : flags_to_status(flags);
}

uint64_t wamd_rescue_workspace_bytes(const wamd_reads* reads, uint32_t n, uint32_t max_mate_edit_distance)
{
    return reads ? rescue_workspace_bytes(reads, n, max_mate_edit_distance) : 0;
}

int wamd_rescue_mates_dev(wamd_ctx* ctx, const wamd_reads* reads, const wamd_rescue_task* d_tasks, uint32_t n,
                          const wamd_rescue_params* params, wamd_rescue_result* d_results, uint8_t* d_ext_cigars,
                          uint32_t ext_stride, void* d_workspace, uint64_t workspace_bytes, void* stream)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, d_tasks, d_results, "wamd_rescue_mates_dev");
    if (rc)
        return rc;
    if (!params || !d_ext_cigars || !d_workspace || ((uintptr_t)d_workspace & 255u) ||
        workspace_bytes < rescue_workspace_bytes(reads, n, 0)) {
        set_error("wamd_rescue_mates_dev: missing argument, or a workspace that is not 256-byte aligned or below "
                  "wamd_rescue_workspace_bytes");
        return WAMD_ERR_INVALID_ARG;
    }
    return launch_rescue_mates(ctx, reads, d_tasks, n, params, d_results, d_ext_cigars, ext_stride, d_workspace,
                               workspace_bytes, pick(ctx, stream));
}

int wamd_rescue_mates(wamd_ctx* ctx, const wamd_reads* reads, const wamd_rescue_task* tasks, uint32_t n,
                      const wamd_rescue_params* params, wamd_rescue_result* results, uint8_t* ext_cigars, uint32_t ext_stride)
{
    if (n == 0)
        return WAMD_OK;
    int rc = check_batch(ctx, reads, tasks, results, "wamd_rescue_mates");
    if (rc)
        return rc;
    if (!params || !ext_cigars)
        return WAMD_ERR_INVALID_ARG;
    WAMD_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> hold(ctx->host_mutex);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    uint32_t max_ed = 0;
    for (uint32_t i = 0; i < n; ++i)
        max_ed = tasks[i].max_mate_edit_distance > max_ed ? tasks[i].max_mate_edit_distance : max_ed;
    const uint64_t tb = (uint64_t)n * sizeof(wamd_rescue_task), rb = (uint64_t)n * sizeof(wamd_rescue_result),
                   eb = (uint64_t)n * ext_stride, wb = rescue_workspace_bytes(reads, n, max_ed);
    rc = ensure_stage(ctx, up(tb) + up(rb) + up(eb) + up(wb) + 256, 0);
    if (rc)
        return rc;
    char* base = (char*)(((uintptr_t)ctx->d_stage + 255) & ~(uintptr_t)255);
    auto* d_tasks = (wamd_rescue_task*)base;
    auto* d_res = (wamd_rescue_result*)(base + up(tb));
    auto* d_ext = (uint8_t*)(base + up(tb) + up(rb));
    void* d_ws = d_ext + up(eb);
    hipStream_t s = ctx->stream;
    WAMD_HIP(hipMemcpyAsync(d_tasks, tasks, tb, hipMemcpyHostToDevice, s));
    rc = wamd_rescue_mates_dev(ctx, reads, d_tasks, n, params, d_res, d_ext, ext_stride, d_ws, wb, s);
    if (rc)
        return rc;
    WAMD_HIP(hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, s));
    WAMD_HIP(hipMemcpyAsync(ext_cigars, d_ext, eb, hipMemcpyDeviceToHost, s));
    uint32_t flags = 0;
    rc = fetch_flags(ctx, s, &flags);
    if (!rc && (flags & FLAG_CIGAR)) {
        set_error("wamd_rescue_mates: an ext-CIGAR does not fit ext_stride");
        return WAMD_ERR_INVALID_ARG;
    }
    return rc ? rc : flags_to_status(flags);
}

int wamd_emit_lev_records_static_dev(wamd_ctx* ctx, uint32_t n, const wamd_dist_task* d_tasks,
                                     const wamd_dist_result* d_results, const uint32_t* d_record_pos,
                                     const uint8_t* d_direction, uint8_t* d_records, uint32_t* d_count, void* stream)
{
    if (!ctx || !d_count)
        return WAMD_ERR_INVALID_ARG;
    if (n == 0) {   // (an empty bin is still refused on a device that has no static scratch)
        const int rc = launch_emit_lev_records_static(ctx, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc)
            return rc;
        WAMD_HIP(hipSetDevice(ctx->device));
        WAMD_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), pick(ctx, stream)));
        return WAMD_OK;
    }
    if (!d_tasks || !d_results || !d_record_pos || !d_direction || !d_records || ((uintptr_t)d_records & 3u))
        return WAMD_ERR_INVALID_ARG;
    return launch_emit_lev_records_static(ctx, n, d_tasks, d_results, d_record_pos, d_direction, d_records, d_count,
                                          pick(ctx, stream));
}

} // extern "C"
