// TEST DOUBLE of the six RCCL entry points csrc/dev_rccl.hip binds with dlsym -- test infrastructure for the CPU box, used together with hipstub.cpp (it is found as
// "librccl.so.1" through LD_LIBRARY_PATH by the tests only).  One process drives every rank (ncclCommInitAll), as the product does: a group of ncclBroadcast calls is
// carried out at ncclGroupEnd as device -> device copies from the root's buffer, each on the stream its rank passed, behind an event recorded on the root's stream
// (the ordering a real broadcast gives).  The copies go through the HIP test double, so they are bounds-checked and, under HIPSTUB_DEFER=1, run late.
//   RCCLSTUB_FAIL=init | bcast : ncclCommInitAll / ncclBroadcast fail, for the product's fallback to host -> device copies
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct ncclComm { int rank, nranks, device; unsigned magic; };

namespace {
struct Pending { const void *send; void *recv; size_t bytes; int root; ncclComm *comm; hipStream_t stream; int device; };
std::vector<Pending> g_group;
int g_depth = 0;
unsigned long g_broadcasts = 0, g_inits = 0;
bool fail(const char *what) { const char *e = std::getenv("RCCLSTUB_FAIL"); return e && !std::strcmp(e, what); }

ncclResult_t flush()
{
    std::vector<Pending> ops;
    ops.swap(g_group);
    int keep = 0;
    (void)hipGetDevice(&keep);
    for (const Pending &root : ops) {
        if (root.comm->rank != root.root) continue;
        hipEvent_t ev;
        if (hipSetDevice(root.device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, root.stream) != hipSuccess) return ncclUnhandledCudaError;
        for (const Pending &p : ops) {
            if (p.comm->rank == p.root || p.root != root.root || p.bytes != root.bytes) continue;
            if (hipSetDevice(p.device) != hipSuccess || hipStreamWaitEvent(p.stream, ev, 0) != hipSuccess ||
                hipMemcpyAsync(p.recv, root.send, p.bytes, hipMemcpyDeviceToDevice, p.stream) != hipSuccess) return ncclUnhandledCudaError;
        }
        (void)hipEventDestroy(ev);
        g_broadcasts++;
    }
    (void)hipSetDevice(keep);
    return ncclSuccess;
}
} // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (fail("init")) return ncclSystemError;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) return ncclUnhandledCudaError;
    for (int i = 0; i < ndev; i++) {
        const int d = devlist ? devlist[i] : i;
        if (d < 0 || d >= have) return ncclInvalidArgument;
        comms[i] = new ncclComm{ i, ndev, d, 0xCC11u };
    }
    g_inits++;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { if (!c || c->magic != 0xCC11u) return ncclInvalidArgument; c->magic = 0; delete c; return ncclSuccess; }
ncclResult_t ncclGroupStart(void) { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth) return ncclSuccess;
    return flush();
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream)
{
    if (fail("bcast")) return ncclSystemError;
    if (!comm || comm->magic != 0xCC11u || type != ncclUint8 || root < 0 || root >= comm->nranks) return ncclInvalidArgument;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != comm->device) { std::fprintf(stderr, "rcclstub: ncclBroadcast for rank %d (GPU %d) called with GPU %d current\n", comm->rank, comm->device, cur); std::abort(); }
    g_group.push_back(Pending{ send, recv, count, root, comm, stream, comm->device });
    if (!g_depth) return flush();
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rcclstub error"; }

unsigned long rcclstub_broadcasts(void) { return g_broadcasts; }
unsigned long rcclstub_inits(void) { return g_inits; }

} // extern "C"
