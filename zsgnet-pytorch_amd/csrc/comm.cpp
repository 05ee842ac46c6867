// comm.cpp — gradient exchange of the data-parallel step over RCCL (xGMI): zsg_comm_{unique_id,init,allreduce_bucket,
// broadcast,wait,destroy}.  Replaces the NCCL collectives torch DistributedDataParallel issues for the reference
// (main_dist.py:36-40: C1 bucketed gradient all-reduce, C2 buffer broadcast, C3 initial parameter broadcast).
// One communicator per process (one process per GPU), a dedicated non-blocking HIP stream for the collectives and a
// ring of events that fence it against the caller's compute stream: a bucket is enqueued as soon as the launches that
// fill it are on the compute stream, and the optimizer waits through zsg_comm_wait.  librccl is bound at run time
// (dlopen, preferring the copy PyTorch already loaded: one RCCL per process), so libzsg.so itself has no link-time
// dependency on it and loads on machines without RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "missing symbols";       // what went wrong while loading (dlerror() is captured once, right after the failed dlopen)
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);       // the copy already in the process (PyTorch's)
        if (g_rccl.h) break;
    }
    for (const char* n : names) {
        if (g_rccl.h) break;
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!g_rccl.h) {
        const char* e = dlerror();               // (one call: it clears the error)
        snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", e ? e : "dlopen failed");
        return;
    }
#define ZSG_SYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.h, name)
    ZSG_SYM(GetUniqueId, "ncclGetUniqueId");
    ZSG_SYM(CommInitRank, "ncclCommInitRank");
    ZSG_SYM(CommDestroy, "ncclCommDestroy");
    ZSG_SYM(AllReduce, "ncclAllReduce");
    ZSG_SYM(Broadcast, "ncclBroadcast");
    ZSG_SYM(GetErrorString, "ncclGetErrorString");
#undef ZSG_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast && g_rccl.GetErrorString;
}
int need_rccl() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) ZSG_FAIL(-4, "zsg_comm: librccl.so.1 could not be loaded (%s)", g_rccl.why);
    return 0;
}
#define ZSG_RCCL(call, what)                                                                         \
    do {                                                                                             \
        ncclResult_t r__ = (call);                                                                   \
        if (r__ != ncclSuccess) ZSG_FAIL(-4, "%s: RCCL error %d: %s", what, (int)r__, g_rccl.GetErrorString(r__)); \
    } while (0)
#define ZSG_HIP(call, what)                                                                \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) ZSG_FAIL(-3, "%s: %s", what, hipGetErrorString(e__));       \
    } while (0)

constexpr int kEvents = 64;
}  // namespace

struct zsg_comm {
    ncclComm_t comm;
    hipStream_t stream;            // the collectives' own stream
    hipEvent_t ready[kEvents];     // compute stream -> comm stream ("the bucket is filled")
    hipEvent_t done;               // comm stream -> compute stream
    int nranks, rank, next_ev, pending;
};

extern "C" int zsg_comm_unique_id(void* id128) {
    ZSG_REQUIRE(id128, "comm_unique_id: null argument");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    ZSG_RCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int zsg_comm_init(zsg_comm** out, const void* id128, int32_t nranks, int32_t rank) {
    ZSG_REQUIRE(out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad argument (nranks=%d rank=%d)", nranks, rank);
    if (int rc = need_rccl()) return rc;
    zsg_comm* c = new zsg_comm();
    memset(c, 0, sizeof(*c));
    c->nranks = nranks;
    c->rank = rank;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);     // collective: every rank calls it, on its own device
    if (r != ncclSuccess) {
        delete c;
        ZSG_FAIL(-4, "ncclCommInitRank(nranks=%d, rank=%d): RCCL error %d: %s", nranks, rank, (int)r, g_rccl.GetErrorString(r));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < kEvents; ++i) e = hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {                       // undo what was created (the struct was zeroed: null handles mark the rest)
        for (int i = 0; i < kEvents; ++i)
            if (c->ready[i]) hipEventDestroy(c->ready[i]);
        if (c->done) hipEventDestroy(c->done);
        if (c->stream) hipStreamDestroy(c->stream);
        g_rccl.CommDestroy(c->comm);
        delete c;
        ZSG_FAIL(-3, "comm_init: %s", hipGetErrorString(e));
    }
    *out = c;
    return 0;
}

// In-place SUM all-reduce of buf[0:count] (fp32), ordered after everything enqueued on `compute_stream` so far; returns at
// once.  The caller must not touch buf on the compute stream before zsg_comm_wait.
extern "C" int zsg_comm_allreduce_bucket(zsg_comm* c, float* buf, int64_t count, void* compute_stream) {
    ZSG_REQUIRE(c && buf && count > 0, "comm_allreduce_bucket: bad argument");
    hipEvent_t ev = c->ready[c->next_ev];
    c->next_ev = (c->next_ev + 1) % kEvents;
    ZSG_HIP(hipEventRecord(ev, (hipStream_t)compute_stream), "comm_allreduce_bucket: event record");
    ZSG_HIP(hipStreamWaitEvent(c->stream, ev, 0), "comm_allreduce_bucket: stream wait");
    ZSG_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, c->comm, c->stream), "ncclAllReduce");
    c->pending += 1;
    return 0;
}

// buf[0:count] of rank `root` to every rank, on the communicator's stream, ordered after `compute_stream`; the compute
// stream then waits for it (BatchNorm running statistics before a training forward; parameters at wrap time).
extern "C" int zsg_comm_broadcast(zsg_comm* c, float* buf, int64_t count, int32_t root, void* compute_stream) {
    ZSG_REQUIRE(c && buf && count > 0, "comm_broadcast: bad argument");      // (the root is validated by RCCL: error -4)
    hipEvent_t ev = c->ready[c->next_ev];
    c->next_ev = (c->next_ev + 1) % kEvents;
    ZSG_HIP(hipEventRecord(ev, (hipStream_t)compute_stream), "comm_broadcast: event record");
    ZSG_HIP(hipStreamWaitEvent(c->stream, ev, 0), "comm_broadcast: stream wait");
    ZSG_RCCL(g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat32, root, c->comm, c->stream), "ncclBroadcast");
    ZSG_HIP(hipEventRecord(c->done, c->stream), "comm_broadcast: event record");
    ZSG_HIP(hipStreamWaitEvent((hipStream_t)compute_stream, c->done, 0), "comm_broadcast: stream wait");
    return 0;
}

// `compute_stream` waits (on the device; the host does not block) for every collective enqueued so far.
extern "C" int zsg_comm_wait(zsg_comm* c, void* compute_stream) {
    ZSG_REQUIRE(c, "comm_wait: null communicator");
    if (!c->pending) return 0;
    ZSG_HIP(hipEventRecord(c->done, c->stream), "comm_wait: event record");
    ZSG_HIP(hipStreamWaitEvent((hipStream_t)compute_stream, c->done, 0), "comm_wait: stream wait");
    c->pending = 0;
    return 0;
}

extern "C" int zsg_comm_destroy(zsg_comm* c) {
    if (!c) return 0;
    hipStreamSynchronize(c->stream);
    ncclResult_t r = g_rccl.CommDestroy(c->comm);
    for (int i = 0; i < kEvents; ++i) hipEventDestroy(c->ready[i]);
    hipEventDestroy(c->done);
    hipStreamDestroy(c->stream);
    delete c;
    if (r != ncclSuccess) ZSG_FAIL(-4, "ncclCommDestroy: RCCL error %d: %s", (int)r, g_rccl.GetErrorString(r));
    return 0;
}
