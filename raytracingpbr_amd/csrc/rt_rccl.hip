// rt_rccl.hip — the ONE collective of the multi-GPU path, on RCCL directly.
//
// The frame is tile-partitioned (rtpbr_set_tiles); each rank renders its tiles into its own image_buffer (T7).  At
// the end of a render (or at a progressive checkpoint) every rank packs its tiles tile-major into one contiguous
// buffer and ONE ncclGather (rccl.h:745, an RCCL extension: per-rank point-to-point over xGMI, no ring) moves
// P/G x 16 B per rank to rank 0, which scatters the buffers back into the full frame.  Pack, gather and unpack are
// enqueued on the context's own HIP stream: no host synchronisation inside.  The reference has no multi-device
// path; these entry points let ANY host (C, Go/cgo, Java/JNI ...) do N > 1 without PyTorch:
//   one process per GPU   : rank 0 calls rtpbr_rccl_unique_id, ships the 128 bytes to the others by its own means
//                           (MPI, a file, torch.distributed), every rank calls rtpbr_rccl_init, later
//                           rtpbr_gather_tiles;
//   one process, G devices: rtpbr_rccl_init_all(ctxs, G) then rtpbr_gather_tiles_all(ctxs, G) (ncclCommInitAll,
//                           rccl.h:236, and a ncclGroupStart/End around the G gathers).
// librccl is loaded with dlopen at first use, so the library has no link-time dependency on it and single-GPU
// hosts never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt_ctx.hpp"

using namespace rt;

namespace {
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;           // optional
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return RTPBR_OK;
    // ROCm's own librccl first (a PyTorch wheel bundles its own copy under the same soname, and a process that has
    // imported torch would otherwise get that one: measured 7 minutes of communicator set-up against 2.6 s);
    // RTPBR_RCCL_LIB overrides.  RTLD_LOCAL: two copies of RCCL in one process must not see each other's symbols.
    const char* names[] = {getenv("RTPBR_RCCL_LIB"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return rt_fail(RTPBR_EHIP, "cannot load librccl: %s", dlerror());
#define RT_SYM(field, name)                                                      \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));     \
    if (!g_rccl.field) return rt_fail(RTPBR_EHIP, "librccl lacks %s", name)
    RT_SYM(GetUniqueId, "ncclGetUniqueId");
    RT_SYM(CommInitRank, "ncclCommInitRank");
    RT_SYM(CommInitAll, "ncclCommInitAll");
    RT_SYM(CommDestroy, "ncclCommDestroy");
    RT_SYM(Gather, "ncclGather");
    RT_SYM(GroupStart, "ncclGroupStart");
    RT_SYM(GroupEnd, "ncclGroupEnd");
    RT_SYM(GetErrorString, "ncclGetErrorString");
    RT_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
    RT_SYM(CommCount, "ncclCommCount");
    RT_SYM(CommUserRank, "ncclCommUserRank");
    RT_SYM(GetVersion, "ncclGetVersion");
#undef RT_SYM
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
    g_rccl.h = h;
    return RTPBR_OK;
}
}  // namespace

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) return rt_fail(RTPBR_EHIP, #expr " failed: %s", g_rccl.GetErrorString(r_)); \
    } while (0)

extern "C" int rtpbr_rccl_unique_id(void* out, size_t nbytes) {
    if (!out || nbytes < NCCL_UNIQUE_ID_BYTES) return rt_fail(RTPBR_EINVAL, "need a %s-byte buffer", "128");
    if (int r = load_rccl()) return r;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(out, &id, NCCL_UNIQUE_ID_BYTES);
    return RTPBR_OK;
}

// Buffers of one rank: its packed tiles, and on rank 0 room for every rank's.  The two capacities are tracked separately:
// the receive side scales with the world size, which can change (rtpbr_set_tiles) while the local share stays the same.
static int ensure_gather_buffers(rtpbr_ctx* c) {
    const size_t bytes = (size_t)c->P.np * sizeof(float4);
    const size_t recv_bytes = c->rank == 0 ? bytes * (size_t)c->world : 0;
    if (bytes > c->gather_cap) {
        RT_HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->gather_send);
        c->gather_send = nullptr;
        c->gather_cap = 0;
        RT_HIP_TRY(hipMalloc(&c->gather_send, bytes));
        c->gather_cap = bytes;
    }
    if (recv_bytes > c->gather_recv_cap) {
        RT_HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->gather_recv);
        c->gather_recv = nullptr;
        c->gather_recv_cap = 0;
        RT_HIP_TRY(hipMalloc(&c->gather_recv, recv_bytes));
        c->gather_recv_cap = recv_bytes;
    }
    return RTPBR_OK;
}

static int check_comm(rtpbr_ctx* c) {
    if (!c || !c->have_cfg) return rt_fail(RTPBR_ESTATE, "set_config first");
    if (!c->comm) return rt_fail(RTPBR_ESTATE, "rtpbr_rccl_init first");
    if (c->comm_world != c->world || c->comm_rank != c->rank)
        return rt_fail(RTPBR_ESTATE, "rtpbr_set_tiles rank/world differ from the communicator's");
    return RTPBR_OK;
}

extern "C" int rtpbr_rccl_init(rtpbr_ctx* c, const void* unique_id, size_t nbytes, int rank, int world) {
    if (!c || !unique_id || nbytes < NCCL_UNIQUE_ID_BYTES) return rt_fail(RTPBR_EINVAL, "bad rccl_init arguments");
    if (world < 1 || rank < 0 || rank >= world) return rt_fail(RTPBR_EINVAL, "bad rank/world");
    if (int r = load_rccl()) return r;
    RT_HIP_TRY(hipSetDevice(c->device));
    if (c->comm) {
        (void)g_rccl.CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm;
    NCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    return RTPBR_OK;
}

extern "C" int rtpbr_rccl_init_all(rtpbr_ctx** ctxs, int n) {
    if (!ctxs || n < 1 || n > 64) return rt_fail(RTPBR_EINVAL, "bad rccl_init_all arguments");
    if (int r = load_rccl()) return r;
    int devs[64];
    ncclComm_t comms[64];
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return rt_fail(RTPBR_EINVAL, "null ctx");
        devs[i] = ctxs[i]->device;
    }
    NCCL_TRY(g_rccl.CommInitAll(comms, n, devs));
    for (int i = 0; i < n; i++) {
        if (ctxs[i]->comm) (void)g_rccl.CommDestroy((ncclComm_t)ctxs[i]->comm);
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_world = n;
    }
    return RTPBR_OK;
}

// What the communicator says about work it has already been given (rccl.h: ncclCommGetAsyncError).  Asked on EVERY rank
// right after the enqueue (a sender's failure must not be silent: only the root used to ask), and again by rtpbr_sync()
// once the stream has drained, when the gather has actually run.
int rt_rccl_check_async(rtpbr_ctx* c) {
    if (!c || !c->comm || !g_rccl.CommGetAsyncError) return RTPBR_OK;
    ncclResult_t async = ncclSuccess;
    NCCL_TRY(g_rccl.CommGetAsyncError((ncclComm_t)c->comm, &async));
    if (async != ncclSuccess && async != ncclInProgress)
        return rt_fail(RTPBR_EHIP, "RCCL asynchronous error on this rank: %s", g_rccl.GetErrorString(async));
    return RTPBR_OK;
}
// A communicator whose COLLECTIVE could not be enqueued is unusable (the other ranks may be waiting inside it): it is
// aborted (ncclCommAbort) so that nothing hangs in a later call, and the context forgets it — the caller handles the error
// and sets up a new one (rtpbr_rccl_init / rtpbr_rccl_init_all) before the next gather.  A librccl without ncclCommAbort gets
// no ncclCommDestroy in its place: destroying a communicator whose peers sit in a half-enqueued collective can itself hang —
// the handle is leaked instead.  Purely LOCAL failures before the collective (hipSetDevice, the pack launch) leave the
// communicator alone: nobody is waiting yet.
static void abort_comm(rtpbr_ctx* c) {
    if (!c->comm) return;
    if (g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)c->comm);
    c->comm = nullptr;
    c->comm_rank = 0;
    c->comm_world = 1;
}

// pack -> ncclGather -> (root) unpack, all on the context's stream.  *collective_failed: the failure happened at (or after)
// the ncclGather call.
static int enqueue_gather(rtpbr_ctx* c, bool* collective_failed) {
    *collective_failed = false;
    RT_HIP_TRY(hipSetDevice(c->device));
    c->P.cfg = c->cfg;
    c->P.image_buffer = c->image_buffer;
    launch_pack(c->P, (float4*)c->gather_send, c->stream);
    RT_HIP_TRY(hipGetLastError());
    *collective_failed = true;
    NCCL_TRY(g_rccl.Gather(c->gather_send, c->gather_recv, (size_t)c->P.np * 4, ncclFloat, 0, (ncclComm_t)c->comm, c->stream));
    *collective_failed = false;
    return RTPBR_OK;
}
static int enqueue_unpack(rtpbr_ctx* c) {
    if (c->rank != 0) return RTPBR_OK;
    RT_HIP_TRY(hipSetDevice(c->device));
    if (int r = rt_order_after_reads(c, 1u << RTPBR_BUF_IMAGE_BUFFER)) return r;      // (an asynchronous read-back of T7 may still be copying)
    for (int src = 1; src < c->world; src++) {       // rank 0's own tiles are already in place
        Params P = c->P;
        P.cfg = c->cfg;
        P.image_buffer = c->image_buffer;
        P.rank = src;
        launch_unpack(P, (const float4*)c->gather_recv + (size_t)src * (size_t)c->P.np, c->stream);
    }
    RT_HIP_TRY(hipGetLastError());
    return RTPBR_OK;
}

extern "C" int rtpbr_gather_tiles(rtpbr_ctx* c) {
    if (int r = check_comm(c)) return r;
    RT_HIP_TRY(hipSetDevice(c->device));
    if (int r = ensure_gather_buffers(c)) return r;
    bool collective_failed = false;
    if (int r = enqueue_gather(c, &collective_failed)) {
        if (collective_failed) abort_comm(c);
        return r;
    }
    if (int r = enqueue_unpack(c)) return r;
    return rt_rccl_check_async(c);      // every rank, sender or root
}

extern "C" int rtpbr_gather_tiles_all(rtpbr_ctx** ctxs, int n) {
    if (!ctxs || n < 1) return rt_fail(RTPBR_EINVAL, "bad gather_tiles_all arguments");
    for (int i = 0; i < n; i++) {
        if (int r = check_comm(ctxs[i])) return r;
        RT_HIP_TRY(hipSetDevice(ctxs[i]->device));
        if (int r = ensure_gather_buffers(ctxs[i])) return r;     // allocations and their syncs stay outside the group
    }
    NCCL_TRY(g_rccl.GroupStart());
    int rc = RTPBR_OK;
    bool collective_failed = false;     // (inside a group every member that WAS enqueued waits for the others: any failure aborts all)
    for (int i = 0; i < n && rc == RTPBR_OK; i++) rc = enqueue_gather(ctxs[i], &collective_failed);
    const ncclResult_t ge = g_rccl.GroupEnd();        // the group is always closed, whatever happened inside it
    if (rc != RTPBR_OK || ge != ncclSuccess) {
        // a member could not be enqueued (or the group could not be launched): the ranks that were are waiting for it —
        // abort every communicator of the group so that nothing hangs, and report the first error
        if (rc == RTPBR_OK) rc = rt_fail(RTPBR_EHIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(ge));
        for (int i = 0; i < n; i++) abort_comm(ctxs[i]);
        return rc;
    }
    for (int i = 0; i < n; i++)
        if (int r = enqueue_unpack(ctxs[i])) return r;
    for (int i = 0; i < n; i++)
        if (int r = rt_rccl_check_async(ctxs[i])) return r;
    return RTPBR_OK;
}

extern "C" int rtpbr_rccl_info(rtpbr_ctx* c, int* nranks, int* rank, int* version) {
    if (!c) return rt_fail(RTPBR_EINVAL, "null ctx");
    if (!c->comm) return rt_fail(RTPBR_ESTATE, "rtpbr_rccl_init first");
    int n = 0, r = 0, v = 0;
    NCCL_TRY(g_rccl.CommCount((ncclComm_t)c->comm, &n));
    NCCL_TRY(g_rccl.CommUserRank((ncclComm_t)c->comm, &r));
    NCCL_TRY(g_rccl.GetVersion(&v));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (version) *version = v;
    return RTPBR_OK;
}

// called by rtpbr_destroy
void rt_rccl_release(rtpbr_ctx* c) {
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    (void)hipFree(c->gather_send);
    (void)hipFree(c->gather_recv);
    c->gather_send = c->gather_recv = nullptr;
    c->gather_cap = c->gather_recv_cap = 0;
}
