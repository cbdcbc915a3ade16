// gather_rccl.hip - the path's only exchange step, callable from the C++ host: variable-length blocks of
// mapping records, one per rank (= per GPU), gathered to the rank that emits SAM, over RCCL / xGMI.
//
// The reference has no counterpart: its worker threads push records into one process-wide collector
// (mapper/results.h:55-159, mapping_core.cpp:980-997).  With read bins sharded over the GPUs of a node
// (mapper.cpp:1847-1862 is where the per-stage workers are created), each GPU holds the records of its
// bins; they travel to the writer's GPU as
//     ncclAllGather of the per-rank record counts  ->  counts on the host (one small read-back)
//     ncclGroupStart; ncclSend on the senders / one ncclRecv per sender on the root; ncclGroupEnd
// and land packed in rank order - a deterministic layout.  Per link that is a few MB per bin, far below
// xGMI's 153 GB/s: no ring or tree tuning is wanted, point-to-point sends are the natural form.
// One process per GPU (wamd_comm_create, the unique id travels by whatever channel the host has) or one
// process driving all GPUs of the node (wamd_comm_create_all + wamd_gather_records_all, every call of all
// ranks inside one group so that a single thread cannot deadlock itself).
// librccl is opened at the first use (dlopen), so a host that never gathers does not need it.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "wamd_internal.h"

struct wamd_comm {
    wamd_ctx* ctx;
    ncclComm_t comm;
    int rank, n_ranks;
    uint32_t* d_counts;   // n_ranks words
    uint32_t* h_counts;   // pinned mirror
};

namespace wamd {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;
static std::mutex g_rccl_lock;

static int rccl_load()
{
    std::lock_guard<std::mutex> hold(g_rccl_lock);
    if (g_rccl.lib)
        return WAMD_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
            break;
    if (!h) {
        set_error("librccl not found (%s): the record gather needs RCCL", dlerror());
        return WAMD_ERR_NO_DEVICE;
    }
    Rccl r;
    r.lib = h;
#define WAMD_SYM(field, name)                                                      \
    if (!(*(void**)(&r.field) = dlsym(h, name))) {                                 \
        set_error("librccl lacks %s", name);                                       \
        dlclose(h);                                                                \
        return WAMD_ERR_NO_DEVICE;                                                 \
    }
    WAMD_SYM(GetUniqueId, "ncclGetUniqueId")
    WAMD_SYM(CommInitRank, "ncclCommInitRank")
    WAMD_SYM(CommInitAll, "ncclCommInitAll")
    WAMD_SYM(CommDestroy, "ncclCommDestroy")
    WAMD_SYM(GroupStart, "ncclGroupStart")
    WAMD_SYM(GroupEnd, "ncclGroupEnd")
    WAMD_SYM(Send, "ncclSend")
    WAMD_SYM(Recv, "ncclRecv")
    WAMD_SYM(AllGather, "ncclAllGather")
    WAMD_SYM(GetErrorString, "ncclGetErrorString")
#undef WAMD_SYM
    g_rccl = r;
    return WAMD_OK;
}

static int rccl_fail(ncclResult_t e, const char* what)
{
    set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return WAMD_ERR_HIP;
}

#define WAMD_RCCL(call, what)                                \
    do {                                                     \
        ncclResult_t e_ = (call);                            \
        if (e_ != ncclSuccess)                               \
            return rccl_fail(e_, what);                      \
    } while (0)

static int comm_finish(wamd_comm* c)
{
    WAMD_HIP(hipSetDevice(c->ctx->device));
    WAMD_HIP(hipMalloc(&c->d_counts, (size_t)c->n_ranks * sizeof(uint32_t)));
    const hipError_t e = hipHostMalloc(&c->h_counts, (size_t)c->n_ranks * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipFree(c->d_counts);
        c->d_counts = nullptr;
        c->h_counts = nullptr;
        return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
    }
    return WAMD_OK;
}

// Inside ncclGroupStart / ncclGroupEnd nothing may return: a thread that leaves a group open has every later
// collective queued behind it and never issued, while the other ranks have already posted theirs and wait.  Calls
// between the two record the first failure and go on to ncclGroupEnd; the failure is reported after the group closed.
struct GroupError {
    int rc = WAMD_OK;
    void hip(hipError_t e, const char* what)
    {
        if (e != hipSuccess && rc == WAMD_OK)
            rc = hip_fail(e, what, __FILE__, __LINE__);
    }
    void rccl(ncclResult_t e, const char* what)
    {
        if (e != ncclSuccess && rc == WAMD_OK)
            rc = rccl_fail(e, what);
    }
};

// the gather of every local rank in `cs` (one entry per process, or all ranks of the node in one process)
static int gather_impl(wamd_comm* const* cs, int n_local, const uint8_t* const* d_records, const uint32_t* const* d_count,
                       uint32_t record_bytes, uint32_t capacity, int root, uint8_t* d_out, uint32_t* counts_out,
                       void* const* streams)
{
    // everything that can be wrong on this rank alone is refused before the first collective is posted, so that a
    // bad argument never leaves the other ranks waiting for this one
    if (n_local <= 0 || !cs || !d_records || !d_count || record_bytes == 0)
        return WAMD_ERR_INVALID_ARG;
    for (int i = 0; i < n_local; ++i)
        if (!cs[i] || !cs[i]->ctx || !cs[i]->comm || !d_count[i] || (!d_records[i] && capacity))
            return WAMD_ERR_INVALID_ARG;
    const int n_ranks = cs[0]->n_ranks;
    if (root < 0 || root >= n_ranks)
        return WAMD_ERR_INVALID_ARG;
    for (int i = 0; i < n_local; ++i) {
        if (cs[i]->n_ranks != n_ranks)
            return WAMD_ERR_INVALID_ARG;
        if (cs[i]->rank == root && !d_out && capacity) {
            set_error("wamd_gather_records: the root rank passes no output buffer");
            return WAMD_ERR_INVALID_ARG;
        }
    }
    auto st = [&](int i) { return streams && streams[i] ? (hipStream_t)streams[i] : cs[i]->ctx->stream; };
    // 1. everybody learns everybody's count
    GroupError ge;
    WAMD_RCCL(g_rccl.GroupStart(), "ncclGroupStart");
    for (int i = 0; i < n_local; ++i) {
        ge.hip(hipSetDevice(cs[i]->ctx->device), "hipSetDevice");
        ge.rccl(g_rccl.AllGather(d_count[i], cs[i]->d_counts, 1, ncclUint32, cs[i]->comm, st(i)), "ncclAllGather");
    }
    ge.rccl(g_rccl.GroupEnd(), "ncclGroupEnd");
    if (ge.rc)
        return ge.rc;
    for (int i = 0; i < n_local; ++i) {
        WAMD_HIP(hipSetDevice(cs[i]->ctx->device));
        WAMD_HIP(hipMemcpyAsync(cs[i]->h_counts, cs[i]->d_counts, (size_t)n_ranks * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, st(i)));
    }
    for (int i = 0; i < n_local; ++i) {
        WAMD_HIP(hipSetDevice(cs[i]->ctx->device));
        WAMD_HIP(hipStreamSynchronize(st(i)));
    }
    // every rank holds the same counts here, so a block beyond the capacity is refused by all of them alike and
    // nobody posts a transfer
    const uint32_t* counts = cs[0]->h_counts;
    for (int r = 0; r < n_ranks; ++r)
        if (counts[r] > capacity) {
            set_error("wamd_gather_records: rank %d holds %u records, the blocks are sized for %u", r, counts[r], capacity);
            return WAMD_ERR_CAPACITY;
        }
    if (counts_out)
        for (int r = 0; r < n_ranks; ++r)
            counts_out[r] = counts[r];
    // 2. the blocks: packed at the root in rank order
    WAMD_RCCL(g_rccl.GroupStart(), "ncclGroupStart");
    for (int i = 0; i < n_local; ++i) {
        wamd_comm* c = cs[i];
        ge.hip(hipSetDevice(c->ctx->device), "hipSetDevice");
        const size_t mine = (size_t)counts[c->rank] * record_bytes;
        if (c->rank != root) {
            if (mine)
                ge.rccl(g_rccl.Send(d_records[i], mine, ncclUint8, root, c->comm, st(i)), "ncclSend");
            continue;
        }
        uint64_t off = 0;
        for (int r = 0; r < n_ranks; ++r) {
            const size_t bytes = (size_t)counts[r] * record_bytes;
            if (bytes) {
                if (r == root)
                    ge.hip(hipMemcpyAsync(d_out + off, d_records[i], bytes, hipMemcpyDeviceToDevice, st(i)), "hipMemcpyAsync");
                else
                    ge.rccl(g_rccl.Recv(d_out + off, bytes, ncclUint8, r, c->comm, st(i)), "ncclRecv");
            }
            off += bytes;
        }
    }
    ge.rccl(g_rccl.GroupEnd(), "ncclGroupEnd");
    return ge.rc;
}

} // namespace wamd

using namespace wamd;

extern "C" {

int wamd_comm_get_unique_id(uint8_t* id)
{
    if (!id)
        return WAMD_ERR_INVALID_ARG;
    int rc = rccl_load();
    if (rc)
        return rc;
    static_assert(WAMD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    WAMD_RCCL(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return WAMD_OK;
}

int wamd_comm_create(wamd_ctx* ctx, int rank, int n_ranks, const uint8_t* id, wamd_comm** out)
{
    if (!ctx || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return WAMD_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = rccl_load();
    if (rc)
        return rc;
    WAMD_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    wamd_comm* c = new wamd_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->n_ranks = n_ranks;
    ncclResult_t e = g_rccl.CommInitRank(&c->comm, n_ranks, u, rank);
    if (e != ncclSuccess) {
        delete c;
        return rccl_fail(e, "ncclCommInitRank");
    }
    rc = comm_finish(c);
    if (rc) {
        wamd_comm_destroy(c);
        return rc;
    }
    *out = c;
    return WAMD_OK;
}

int wamd_comm_create_all(wamd_ctx* const* ctxs, int n, wamd_comm** out)
{
    if (!ctxs || !out || n < 1)
        return WAMD_ERR_INVALID_ARG;
    int rc = rccl_load();
    if (rc)
        return rc;
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i])
            return WAMD_ERR_INVALID_ARG;
        devs[i] = ctxs[i]->device;
        out[i] = nullptr;
    }
    WAMD_RCCL(g_rccl.CommInitAll(comms.data(), n, devs.data()), "ncclCommInitAll");
    for (int i = 0; i < n; ++i) {
        wamd_comm* c = new wamd_comm();
        c->ctx = ctxs[i];
        c->comm = comms[i];
        c->rank = i;
        c->n_ranks = n;
        out[i] = c;
        rc = comm_finish(c);
        if (rc) {   // nothing half-made is left behind: the wrapped handles, and the communicators not yet wrapped
            for (int j = 0; j <= i; ++j) {
                wamd_comm_destroy(out[j]);
                out[j] = nullptr;
            }
            for (int j = i + 1; j < n; ++j)
                (void)g_rccl.CommDestroy(comms[j]);
            return rc;
        }
    }
    return WAMD_OK;
}

int wamd_comm_destroy(wamd_comm* c)
{
    if (!c)
        return WAMD_OK;
    (void)hipSetDevice(c->ctx->device);
    if (c->comm && g_rccl.CommDestroy)
        (void)g_rccl.CommDestroy(c->comm);
    if (c->d_counts)
        (void)hipFree(c->d_counts);
    if (c->h_counts)
        (void)hipHostFree(c->h_counts);
    delete c;
    return WAMD_OK;
}

int wamd_gather_records_dev(wamd_comm* comm, const uint8_t* d_records, const uint32_t* d_count, uint32_t record_bytes,
                            uint32_t capacity, int root, uint8_t* d_out, uint32_t* counts_out, void* stream)
{
    if (!comm)
        return WAMD_ERR_INVALID_ARG;
    return gather_impl(&comm, 1, &d_records, &d_count, record_bytes, capacity, root, d_out, counts_out, &stream);
}

int wamd_gather_records_all(wamd_comm* const* comms, int n, const uint8_t* const* d_records,
                            const uint32_t* const* d_count, uint32_t record_bytes, uint32_t capacity, int root,
                            uint8_t* d_out, uint32_t* counts_out, void* const* streams)
{
    if (!comms || n < 1 || !comms[0] || n != comms[0]->n_ranks)
        return WAMD_ERR_INVALID_ARG;
    return gather_impl(comms, n, d_records, d_count, record_bytes, capacity, root, d_out, counts_out, streams);
}

} // extern "C"
