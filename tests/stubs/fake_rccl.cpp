// fake_rccl.cpp — TEST INFRASTRUCTURE: an in-process stand-in for librccl with exactly the entry points rt_rccl.hip
// resolves (RTPBR_RCCL_LIB points the product library at it).  The box the tests run on has ONE GPU and RCCL refuses two
// ranks on one device, so everything of the multi-GPU path that is NOT RCCL — per-rank packing, the receive offsets
// (rank r's tiles land at r * count), the root's unpack loop over src = 1 .. world-1, the grouped launch of G contexts
// from one process, buffer growth when the world changes — would otherwise only ever run with world = 1.  This library
// implements ncclCommInitAll / ncclGather / ncclGroupStart / ncclGroupEnd for G communicators that live in ONE process
// (all may sit on the same device): a gather is G device-to-device copies on the root's stream, each ordered after the
// sender's stream with an event — the same stream semantics RCCL gives.  It moves bytes and nothing else: no arithmetic
// of the product or of the oracle.  ncclCommInitRank is supported for world = 1 only (a real multi-process job needs RCCL).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclFloat = 7 } ncclDataType_t;
struct Group { int world; std::vector<struct Comm*> members; };
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
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0x5a, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
    if (!comms || n < 1) return ncclInvalidArgument;
    Group* g = new Group{n, {}};
    for (int i = 0; i < n; i++) {
        comms[i] = new Comm{g, i, devs ? devs[i] : i};
        g->members.push_back(comms[i]);
    }
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId, int rank) {
    if (world != 1 || rank != 0) return ncclInvalidUsage;      // one process = one rank needs the real library
    int dev = 0;
    (void)hipGetDevice(&dev);
    Group* g = new Group{1, {}};
    *comm = new Comm{g, 0, dev};
    g->members.push_back(*comm);
    return ncclSuccess;
}
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
    std::lock_guard<std::mutex> l(g_mu);
    g_pending.push_back(Op{send, recv, count * 4, root, c, st});
    return g_depth == 0 ? flush() : ncclSuccess;               // outside a group only a 1-rank communicator can complete
}
}
