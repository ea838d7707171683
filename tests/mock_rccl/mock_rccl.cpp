// TEST INFRASTRUCTURE: a stand-in for the six RCCL entry points trpl_multi_* binds (TRPL_RCCL_LIBRARY), so that the
// multi-rank logic of trpl_loglik_multi_dev -- shard loop, pointer tables, padded exchange, unpadding -- can run with
// several "ranks" on the ONE device of a test box (TRPL_MULTI_ALLOW_DUP=1).  Not RCCL and not a product path: an
// all-gather here is a set of stream-ordered device-to-device copies.  The real library is exercised by the one-rank
// tests (tests/test_gpu_round2.py).
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

extern "C" {

typedef struct mockComm { int rank, n, dev; } *ncclComm_t;
typedef int ncclResult_t;          // 0 = ncclSuccess
typedef int ncclDataType_t;        // 7 = ncclFloat32, 8 = ncclFloat64

namespace {
struct Op { const void *send; void *recv; size_t count; int dtype; ncclComm_t comm; hipStream_t st; };
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;
size_t elem(int dt) { return dt == 8 ? 8 : 4; }

ncclResult_t flush()
{
    std::vector<hipEvent_t> ev(g_ops.size());
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < g_ops.size(); i++) {                 // every rank's send buffer is ready after its event
        if (hipSetDevice(g_ops[i].comm->dev) != hipSuccess) return 1;
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return 1;
        if (hipEventRecord(ev[i], g_ops[i].st) != hipSuccess) return 1;
    }
    for (size_t r = 0; r < g_ops.size(); r++) {
        const Op &o = g_ops[r];
        if (hipSetDevice(o.comm->dev) != hipSuccess) return 1;
        for (size_t s = 0; s < g_ops.size(); s++) {
            const Op &q = g_ops[s];
            if (hipStreamWaitEvent(o.st, ev[s], 0) != hipSuccess) return 1;
            const size_t bytes = q.count * elem(q.dtype);
            if (hipMemcpyAsync((char *)o.recv + (size_t)q.comm->rank * bytes, q.send, bytes, hipMemcpyDefault, o.st) != hipSuccess)
                return 1;
        }
    }
    for (size_t i = 0; i < ev.size(); i++) (void)hipEventDestroy(ev[i]);      // destruction is deferred until complete
    (void)hipSetDevice(prev);
    g_ops.clear();
    return 0;
}
}  // namespace

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs)
{
    for (int r = 0; r < n; r++) { comms[r] = (ncclComm_t)malloc(sizeof(struct mockComm)); comms[r]->rank = r; comms[r]->n = n; comms[r]->dev = devs[r]; }
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { free(c); return 0; }
ncclResult_t ncclGroupStart(void) { g_depth++; return 0; }
ncclResult_t ncclGroupEnd(void) { if (--g_depth == 0) return flush(); return 0; }
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st)
{
    g_ops.push_back({send, recv, count, dt, comm, st});
    if (g_depth == 0) return flush();
    return 0;
}
const char *ncclGetErrorString(ncclResult_t r) { return r ? "mock RCCL: a HIP call failed" : "no error"; }

}  // extern "C"
