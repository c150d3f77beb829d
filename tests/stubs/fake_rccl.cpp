// fake_rccl.cpp — TEST INFRASTRUCTURE: an in-process stand-in for librccl with exactly the entry points rt_rccl.hip
// resolves (RTPBR_RCCL_LIB points the product library at it).  The box the tests run on has ONE GPU and RCCL refuses two
// ranks on one device, so everything of the multi-GPU path that is NOT RCCL — per-rank packing, the receive offsets
// (rank r's tiles land at r * count), the root's unpack loop over src = 1 .. world-1, the grouped launch of G contexts
// from one process, buffer growth when the world changes — would otherwise only ever run with world = 1.  This library
// implements ncclCommInitAll / ncclGather / ncclGroupStart / ncclGroupEnd for G communicators that live in ONE process
// (all may sit on the same device): a gather is G device-to-device copies on the root's stream, each ordered after the
// sender's stream with an event — the same stream semantics RCCL gives.  It moves bytes and nothing else: no arithmetic
// of the product or of the oracle.
// ncclCommInitRank with world > 1 (round 4): ONE PROCESS PER RANK, all on the same GPU — the shape of the real job
// (torchrun, bench.py --gpus N --transport rccl), which RCCL itself refuses on one device.  The ranks meet in a POSIX
// shared-memory segment named after the 128-byte id; a gather moves the senders' packed tiles ACROSS PROCESSES: each
// sender publishes a hipIpcMemHandle of its send buffer (after draining its own stream: the pack has finished), the root
// opens the handles, copies device-to-device on its stream into recv + rank * bytes, drains, and releases the senders.
// Host-synchronous where RCCL is stream-asynchronous (a stand-in, not a transport), data path and offsets identical.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclFloat = 7 } ncclDataType_t;
// the rendezvous of a multi-process communicator (one cache line per field would be nicer; this is a test stub)
struct Shm {
    std::atomic<int> arrived;                 // ranks that have mapped the segment
    std::atomic<unsigned> posted[64];         // per rank: sequence number of the gather its slot describes
    std::atomic<unsigned> done;               // sequence number of the last gather the root has completed
    std::atomic<int> failed;
    hipIpcMemHandle_t handle[64];
    unsigned long long bytes[64];
};
struct Group { int world; std::vector<struct Comm*> members; Shm* shm = nullptr; unsigned seq = 0; char name[64] = {0}; };
struct Comm { Group* group; int rank; int device; };
typedef Comm* ncclComm_t;
}

namespace {
std::mutex g_mu;
int g_depth = 0;
struct Op { const void* send; void* recv; size_t bytes; int root; Comm* comm; hipStream_t stream; };
std::vector<Op> g_pending;

ncclResult_t flush() {
    // every member of a group must have posted its part; the root's recv buffer is the one that counts
    std::vector<Op> ops;
    ops.swap(g_pending);
    for (const Op& r : ops) {
        if (r.comm->rank != r.root) continue;
        for (const Op& s : ops) {
            if (s.comm->group != r.comm->group) continue;
            hipEvent_t ev;
            if (hipSetDevice(s.comm->device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
            if (hipEventRecord(ev, s.stream) != hipSuccess) return ncclUnhandledCudaError;                  // after the sender's pack
            if (hipSetDevice(r.comm->device) != hipSuccess || hipStreamWaitEvent(r.stream, ev, 0) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemcpyAsync((char*)r.recv + (size_t)s.comm->rank * s.bytes, s.send, s.bytes, hipMemcpyDeviceToDevice, r.stream) != hipSuccess)
                return ncclUnhandledCudaError;
            (void)hipEventDestroy(ev);                                                                     // released when it has fired
        }
        size_t posted = 0;
        for (const Op& s : ops) posted += s.comm->group == r.comm->group;
        if ((int)posted != r.comm->group->world) return ncclInvalidUsage;                                  // a rank did not take part
    }
    return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetVersion(int* v) { *v = 99999; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0x5a, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%lld_%ld", (int)getpid(), (long long)ts.tv_sec, ts.tv_nsec);   // names the rendezvous
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
    if (!comms || n < 1) return ncclInvalidArgument;
    Group* g = new Group{n, {}};
    for (int i = 0; i < n; i++) {
        comms[i] = new Comm{g, i, devs ? devs[i] : i};
        g->members.push_back(comms[i]);
    }
    return ncclSuccess;
}
static bool wait_for(const std::atomic<int>& v, int want, double seconds) {
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (v.load() < want) {
        usleep(200);
        clock_gettime(CLOCK_MONOTONIC, &t);
        if ((t.tv_sec - t0.tv_sec) + 1e-9 * (t.tv_nsec - t0.tv_nsec) > seconds) return false;
    }
    return true;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return ncclInvalidArgument;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Group* g = new Group{world, {}};
    *comm = new Comm{g, rank, dev};
    g->members.push_back(*comm);
    if (world == 1) return ncclSuccess;
    // one process per rank: meet in the segment the id names (whoever comes first creates it; ftruncate zero-fills)
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(g->name, sizeof g->name, "%s", id.internal);
    const int fd = shm_open(g->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) return ncclSystemError;
    void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    g->shm = static_cast<Shm*>(p);
    g->shm->arrived.fetch_add(1);
    if (!wait_for(g->shm->arrived, world, 120.0)) return ncclSystemError;       // collective, like the real call
    if (rank == 0) shm_unlink(g->name);                                          // everybody has it mapped
    return ncclSuccess;
}
// gather of a multi-process communicator (see the header): sequence number s, senders publish, root copies and releases
static ncclResult_t gather_ipc(const void* send, void* recv, size_t bytes, int root, Comm* c, hipStream_t st) {
    Group* g = c->group;
    Shm* m = g->shm;
    const unsigned s = ++g->seq;
    auto spin = [&](auto cond) {
        struct timespec t0, t;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (!cond()) {
            if (m->failed.load()) return false;
            usleep(50);
            clock_gettime(CLOCK_MONOTONIC, &t);
            if (t.tv_sec - t0.tv_sec > 120) return false;
        }
        return true;
    };
    if (c->rank != root) {
        if (hipStreamSynchronize(st) != hipSuccess) { m->failed = 1; return ncclUnhandledCudaError; }      // the pack has finished
        if (hipIpcGetMemHandle(&m->handle[c->rank], const_cast<void*>(send)) != hipSuccess) { m->failed = 1; return ncclUnhandledCudaError; }
        m->bytes[c->rank] = bytes;
        m->posted[c->rank].store(s);
        // the send buffer must stay untouched until the root has copied it
        return spin([&] { return m->done.load() >= s; }) ? ncclSuccess : ncclSystemError;
    }
    for (int r = 0; r < g->world; r++) {
        if (r == root) {
            if (hipMemcpyAsync((char*)recv + (size_t)r * bytes, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { m->failed = 1; return ncclUnhandledCudaError; }
            continue;
        }
        if (!spin([&] { return m->posted[r].load() >= s; })) return ncclSystemError;
        if (m->bytes[r] != bytes) { m->failed = 1; return ncclInvalidUsage; }
        void* peer = nullptr;
        if (hipIpcOpenMemHandle(&peer, m->handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { m->failed = 1; return ncclUnhandledCudaError; }
        const hipError_t e = hipMemcpyAsync((char*)recv + (size_t)r * bytes, peer, bytes, hipMemcpyDeviceToDevice, st);
        const hipError_t e2 = hipStreamSynchronize(st);
        (void)hipIpcCloseMemHandle(peer);
        if (e != hipSuccess || e2 != hipSuccess) { m->failed = 1; return ncclUnhandledCudaError; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { m->failed = 1; return ncclUnhandledCudaError; }
    m->done.store(s);
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) { if (c->group->shm) c->group->shm->failed = 1; delete c; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }      // (groups are leaked: test processes are short)
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->group->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t* e) { *e = ncclSuccess; return ncclSuccess; }
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> l(g_mu); g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    std::lock_guard<std::mutex> l(g_mu);
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth == 0 ? flush() : ncclSuccess;
}
ncclResult_t ncclGather(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st) {
    if (t != ncclFloat || !c || root < 0 || root >= c->group->world) return ncclInvalidArgument;
    if (c->group->shm) return gather_ipc(send, recv, count * 4, root, c, st);
    std::lock_guard<std::mutex> l(g_mu);
    g_pending.push_back(Op{send, recv, count * 4, root, c, st});
    return g_depth == 0 ? flush() : ncclSuccess;               // outside a group only a 1-rank communicator can complete
}
}
