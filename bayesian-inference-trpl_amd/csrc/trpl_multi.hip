// Multi-device entry points of include/trpl.h: the sample batch cut into contiguous shards, one per device.
//   trpl_loglik_multi      host arrays in, host arrays out; no device-to-device exchange
//   trpl_loglik_multi_dev  device-resident: per-device shards in, ONE RCCL all-gather over xGMI, the complete
//                          likelihood vector in every device's memory out
// The reference has no communication at all (one SLURM array task per GPU, bayeslib.py:131,:231; the threaded
// driver :235-246 is commented out).
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only: the library is bound at run time (rccl_api())
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "api_util.hpp"

using namespace trpl;

namespace {

// one shard's fused solve on the current device: trpl_loglik_dev / trpl_loglik_obs_dev
int shard_loglik(const double *X, int64_t n, int32_t C, const double *lengths_nm, double time_ns, int32_t L, int64_t T,
                 int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, const double *obs,
                 const int32_t *obs_hi, const double *obs_dx, const double *obs_h, int64_t obs_ld, const int64_t *n_obs,
                 double *P, double *sse, int32_t *status, int64_t *iters_total, int32_t *floor_col, uint32_t flags,
                 hipStream_t st)
{
    if (obs_hi)
        return trpl_loglik_obs_dev(X, n, C, lengths_nm, time_ns, L, T, tol_exp, max_iter, dN, obs, obs_hi, obs_dx, obs_h,
                                   obs_ld, n_obs, P, sse, status, iters_total, floor_col, flags, st);
    return trpl_loglik_dev(X, n, C, lengths_nm, time_ns, L, T, plT, tol_exp, max_iter, dN, obs, obs_ld, n_obs, P, sse,
                           status, iters_total, floor_col, flags, st);
}

}  // namespace

extern "C" {

int trpl_shard_bounds(int64_t S, int32_t n_shards, int32_t shard, int64_t *lo, int64_t *hi)
{
    if (S < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || !lo || !hi)
        return api_fail(TRPL_ERR_ARG, "shard %d of %d over S=%lld is not a valid request", shard, n_shards, (long long)S);
    const int64_t base = S / n_shards, rem = S % n_shards;
    *lo = shard * base + (shard < rem ? shard : rem);
    *hi = *lo + base + (shard < rem ? 1 : 0);
    return TRPL_OK;
}

int64_t trpl_shard_of(int64_t S, int32_t n_shards, int64_t s)
{
    if (S < 1 || n_shards < 1 || s < 0 || s >= S) return -1;
    return shard_of(S, n_shards, s);
}

}  // extern "C" (reopened below)

namespace {
struct Shard {                       // one device's share of the samples; released on its own device
    int dev = 0;
    int64_t lo = 0, hi = 0;
    hipStream_t st = nullptr;
    DevBuf X, dN, obs, ohi, odx, oh, P, sse, status, iters, floorc;
    ~Shard()
    {
        (void)hipSetDevice(dev);
        for (DevBuf *b : {&X, &dN, &obs, &ohi, &odx, &oh, &P, &sse, &status, &iters, &floorc}) b->release();   // stream-ordered frees
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    }
};
}  // namespace


extern "C" {

int trpl_loglik_multi(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns, int32_t L,
                      int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, const double *obs,
                      const int32_t *obs_hi, const double *obs_dx, const double *obs_h, int64_t obs_ld,
                      const int64_t *n_obs, double *P, double *sse, int32_t *status, int64_t *iters_total,
                      int32_t *floor_col, uint32_t flags, const int32_t *devices, int32_t n_devices, double *seconds)
{
    ProfRange range("trpl_loglik_multi (sample shards over the visible devices)");
    if (int rc = check_grid(L, T, plT, max_iter, time_ns)) return rc;
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (C < 1 || C > TRPL_MAX_CURVES) return api_fail(TRPL_ERR_ARG, "C=%d must be in [1, %d]", C, TRPL_MAX_CURVES);
    if (seconds) *seconds = 0.0;
    if (!X || !lengths_nm || !dN || !obs || !n_obs || !P) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (obs_ld < 1) return api_fail(TRPL_ERR_ARG, "obs_ld must be >= 1");
    const bool interp = obs_hi || obs_dx || obs_h;
    if (interp && !(obs_hi && obs_dx && obs_h)) return api_fail(TRPL_ERR_ARG, "obs_hi, obs_dx and obs_h go together");
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return api_fail(TRPL_ERR_NODEVICE, "no HIP device visible");
    if (n_devices <= 0) {
        if (devices) return api_fail(TRPL_ERR_ARG, "a device list needs n_devices >= 1");
        n_devices = visible;
    }
    if (n_devices > 64) return api_fail(TRPL_ERR_ARG, "n_devices=%d exceeds 64", n_devices);
    for (int r = 0; devices && r < n_devices; r++)
        if (devices[r] < 0 || devices[r] >= visible)
            return api_fail(TRPL_ERR_ARG, "devices[%d]=%d out of range (%d visible)", r, devices[r], visible);
    if (int rc = check_variant_flags(flags, L)) return rc;
    if (flags & 0xF00u) return api_fail(TRPL_ERR_UNSUPPORTED, "TRPL_FLAG_BUNDLE couples neighbouring samples: a sharded batch would depend on where it is cut");
    for (int c = 0; c < C; c++)
        if (n_obs[c] < 1 || n_obs[c] > obs_ld)
            return api_fail(TRPL_ERR_ARG, "n_obs[%d]=%lld out of range (obs_ld %lld)", c, (long long)n_obs[c], (long long)obs_ld);
    if (interp) {                                    // host data: the same checks as trpl_loglik_obs
        if (plT != 1) return api_fail(TRPL_ERR_ARG, "off-grid observations need plT = 1");
        if (int rc = check_brackets(obs_hi, obs_dx, obs_h, C, obs_ld, n_obs, T)) return rc;
    }
    if (S == 0) return TRPL_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    // ONE kernel variant for the whole logical batch, whatever the shard sizes: a sample's bits then do not
    // depend on how the batch is cut (the two FAST kernels agree to rounding only)
    flags = pin_variant(flags, S * (int64_t)C, L, loglik_steps(interp, C, n_obs, plT, T));

    std::vector<Shard> sh(n_devices);
    const size_t nobs = (size_t)C * obs_ld;
    const double t0 = now_s();
    int rc = TRPL_OK;
    // stage and launch on every device before waiting for any of them
    for (int r = 0; r < n_devices && rc == TRPL_OK; r++) {
        Shard &q = sh[r];
        q.dev = devices ? devices[r] : r;
        (void)trpl_shard_bounds(S, n_devices, r, &q.lo, &q.hi);
        const int64_t n = q.hi - q.lo;
        if (n == 0) continue;
        const size_t nsys = (size_t)n * C;
        rc = [&]() -> int {
            HIP_TRY(hipSetDevice(q.dev));
            HIP_TRY(hipStreamCreateWithFlags(&q.st, hipStreamNonBlocking));
            HIP_TRY(q.X.alloc((size_t)n * 13 * 8, q.st)); HIP_TRY(q.dN.alloc((size_t)C * L * 8, q.st)); HIP_TRY(q.obs.alloc(nobs * 8, q.st));
            HIP_TRY(q.P.alloc((size_t)n * 8, q.st)); HIP_TRY(q.sse.alloc(nsys * 8, q.st)); HIP_TRY(q.status.alloc(nsys * 4, q.st));
            HIP_TRY(q.iters.alloc(nsys * 8, q.st));
            if (floor_col) HIP_TRY(q.floorc.alloc(nsys * 4, q.st));
            HIP_TRY(hipMemcpyAsync(q.X.p, X + q.lo * 13, (size_t)n * 13 * 8, hipMemcpyHostToDevice, q.st));
            HIP_TRY(hipMemcpyAsync(q.dN.p, dN, (size_t)C * L * 8, hipMemcpyHostToDevice, q.st));
            HIP_TRY(hipMemcpyAsync(q.obs.p, obs, nobs * 8, hipMemcpyHostToDevice, q.st));
            HIP_TRY(hipMemcpyAsync(q.P.p, P + q.lo, (size_t)n * 8, hipMemcpyHostToDevice, q.st));
            if (interp) {
                HIP_TRY(q.ohi.alloc(nobs * 4, q.st)); HIP_TRY(q.odx.alloc(nobs * 8, q.st)); HIP_TRY(q.oh.alloc(nobs * 8, q.st));
                HIP_TRY(hipMemcpyAsync(q.ohi.p, obs_hi, nobs * 4, hipMemcpyHostToDevice, q.st));
                HIP_TRY(hipMemcpyAsync(q.odx.p, obs_dx, nobs * 8, hipMemcpyHostToDevice, q.st));
                HIP_TRY(hipMemcpyAsync(q.oh.p, obs_h, nobs * 8, hipMemcpyHostToDevice, q.st));
            }
            return shard_loglik(q.X.as<double>(), n, C, lengths_nm, time_ns, L, T, plT, tol_exp, max_iter, q.dN.as<double>(),
                                q.obs.as<double>(), interp ? q.ohi.as<int32_t>() : nullptr,
                                interp ? q.odx.as<double>() : nullptr, interp ? q.oh.as<double>() : nullptr, obs_ld,
                                n_obs, q.P.as<double>(), q.sse.as<double>(), q.status.as<int32_t>(),
                                q.iters.as<int64_t>(), q.floorc.as<int32_t>(), flags, q.st);
        }();
    }
    // only now the copies back: a device-to-host copy into pageable memory blocks the calling thread until
    // the shard's kernel has finished, so issuing it inside the loop above would run the devices one
    // after the other
    for (int r = 0; r < n_devices && rc == TRPL_OK; r++) {
        Shard &q = sh[r];
        const int64_t n = q.hi - q.lo;
        if (n == 0 || !q.st) continue;
        rc = [&]() -> int {
            HIP_TRY(hipSetDevice(q.dev));
            HIP_TRY(hipMemcpyAsync(P + q.lo, q.P.p, (size_t)n * 8, hipMemcpyDeviceToHost, q.st));
            // per-curve outputs are [C][S] on the host and [C][n] on the device: one strided copy each
            if (sse)
                HIP_TRY(hipMemcpy2DAsync(sse + q.lo, (size_t)S * 8, q.sse.p, (size_t)n * 8, (size_t)n * 8, C,
                                         hipMemcpyDeviceToHost, q.st));
            if (status)
                HIP_TRY(hipMemcpy2DAsync(status + q.lo, (size_t)S * 4, q.status.p, (size_t)n * 4, (size_t)n * 4, C,
                                         hipMemcpyDeviceToHost, q.st));
            if (iters_total)
                HIP_TRY(hipMemcpy2DAsync(iters_total + q.lo, (size_t)S * 8, q.iters.p, (size_t)n * 8, (size_t)n * 8, C,
                                         hipMemcpyDeviceToHost, q.st));
            if (floor_col)
                HIP_TRY(hipMemcpy2DAsync(floor_col + q.lo, (size_t)S * 4, q.floorc.p, (size_t)n * 4, (size_t)n * 4, C,
                                         hipMemcpyDeviceToHost, q.st));
            return TRPL_OK;
        }();
    }
    // drain every stream that was started, also after a failure (the host buffers are borrowed)
    for (int r = 0; r < n_devices; r++) {
        if (!sh[r].st) continue;
        hipError_t e = hipSetDevice(sh[r].dev);
        if (e == hipSuccess) e = hipStreamSynchronize(sh[r].st);
        if (e != hipSuccess && rc == TRPL_OK)
            rc = api_fail(TRPL_ERR_HIP, "device %d (shard %d): %s", sh[r].dev, r, hipGetErrorString(e));
    }
    if (seconds) *seconds = now_s() - t0;
    sh.clear();
    (void)hipSetDevice(prev);
    return rc;
}

/* ------------------------------------------------------------------ device-resident, RCCL ---- */
namespace {

// RCCL bound at first use: librccl.so.1 is ~0.5 GB of code objects, and a process that has torch loaded
// already holds a copy under the same soname -- dlopen() by soname then returns that one, so there is
// never more than one RCCL in a process.  TRPL_RCCL_LIBRARY names another file.
struct RcclApi {
    void *dl = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

const RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("TRPL_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            api.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.dl) break;
            snprintf(api.why, sizeof api.why, "%s", dlerror());
        }
        if (!api.dl) return;
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(api.dl, name);
            if (!p) { ok = false; snprintf(api.why, sizeof api.why, "librccl lacks %s", name); }
            return p;
        };
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(api.dl); api.dl = nullptr; }
    });
    return &api;
}

#define RCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return api_fail(TRPL_ERR_HIP, "%s: %s", #expr, rccl_api()->GetErrorString(r_));       \
    } while (0)

// out[s] = gathered[r * widest + (s - lo_r)]: the all-gather moves equal counts, shards differ by one sample
__global__ void unpad_kernel(const double *__restrict__ gathered, double *__restrict__ out, int64_t S, int n_shards,
                             int64_t widest)
{
    const int64_t base = S / n_shards, rem = S % n_shards;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = shard_of(S, n_shards, s);
        const int64_t lo = r * base + (r < rem ? r : rem);
        out[s] = gathered[r * widest + (s - lo)];
    }
}

}  // namespace

struct trpl_multi {
    int n = 0;
    std::vector<int> dev;
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> ev_in, ev_out; // ordering against the caller's streams (trpl_multi_wait_stream / _release_stream)
    std::vector<ncclComm_t> comm;
    std::vector<double *> send, recv;      // per device: padded shard [widest], gathered [n * widest]
    int64_t cap = 0;                       // `widest` the scratch buffers were sized for
};

int trpl_multi_create(const int32_t *devices, int32_t n_devices, trpl_multi_t **handle)
{
    return trpl_multi_create_ex(devices, n_devices, 0u, handle);
}

int trpl_multi_create_ex(const int32_t *devices, int32_t n_devices, uint32_t create_flags, trpl_multi_t **handle)
{
    if (!handle) return api_fail(TRPL_ERR_ARG, "handle must not be NULL");
    *handle = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return api_fail(TRPL_ERR_NODEVICE, "no HIP device visible");
    if (n_devices <= 0) {
        if (devices) return api_fail(TRPL_ERR_ARG, "a device list needs n_devices >= 1");
        n_devices = visible;
    }
    if (n_devices > 64) return api_fail(TRPL_ERR_ARG, "n_devices=%d exceeds 64", n_devices);
    trpl_multi *h = new trpl_multi;
    h->n = n_devices;
    for (int r = 0; r < n_devices; r++) {
        const int d = devices ? devices[r] : r;
        // RCCL wants one rank per device; TRPL_MULTI_ALLOW_DUPLICATE_DEVICES lets the tests run several "ranks" on one
        // device against a stand-in collective library (tests/mock_rccl)
        const bool allow_dup = (create_flags & TRPL_MULTI_ALLOW_DUPLICATE_DEVICES) != 0;
        bool dup = false;
        for (int q = 0; q < r && !allow_dup; q++) dup = dup || h->dev[q] == d;
        if (d < 0 || d >= visible || dup) {
            delete h;
            return api_fail(TRPL_ERR_ARG, "devices[%d]=%d: out of range (%d visible) or listed twice (one RCCL rank per device)", r, d, visible);
        }
        h->dev.push_back(d);
    }
    const RcclApi *api = rccl_api();
    if (!api->dl) { delete h; return api_fail(TRPL_ERR_UNSUPPORTED, "RCCL could not be loaded: %s", api->why); }
    int prev = 0;
    (void)hipGetDevice(&prev);
    h->ev_in.assign(n_devices, nullptr); h->ev_out.assign(n_devices, nullptr);
    h->st.assign(n_devices, nullptr); h->send.assign(n_devices, nullptr); h->recv.assign(n_devices, nullptr);
    h->comm.assign(n_devices, nullptr);
    int rc = [&]() -> int {
        for (int r = 0; r < n_devices; r++) {
            HIP_TRY(hipSetDevice(h->dev[r]));
            HIP_TRY(hipStreamCreateWithFlags(&h->st[r], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_in[r], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_out[r], hipEventDisableTiming));
        }
        RCCL_TRY(api->CommInitAll(h->comm.data(), n_devices, h->dev.data()));
        return TRPL_OK;
    }();
    (void)hipSetDevice(prev);
    if (rc != TRPL_OK) {
        for (int r = 0; r < n_devices; r++) {
            (void)hipSetDevice(h->dev[r]);
            if (h->st[r]) (void)hipStreamDestroy(h->st[r]);
            if (h->ev_in[r]) (void)hipEventDestroy(h->ev_in[r]);
            if (h->ev_out[r]) (void)hipEventDestroy(h->ev_out[r]);
        }
        (void)hipSetDevice(prev);
        delete h;
        return rc;
    }
    *handle = h;
    return TRPL_OK;
}

int trpl_multi_device_count(const trpl_multi_t *h) { return h ? h->n : 0; }

int trpl_multi_synchronize(trpl_multi_t *h)
{
    if (!h) return api_fail(TRPL_ERR_ARG, "handle must not be NULL");
    int prev = 0, rc = TRPL_OK;
    (void)hipGetDevice(&prev);
    for (int r = 0; r < h->n; r++) {
        hipError_t e = hipSetDevice(h->dev[r]);
        if (e == hipSuccess) e = hipStreamSynchronize(h->st[r]);
        if (e != hipSuccess && rc == TRPL_OK) rc = api_fail(TRPL_ERR_HIP, "device %d (rank %d): %s", h->dev[r], r, hipGetErrorString(e));
    }
    (void)hipSetDevice(prev);
    return rc;
}

int trpl_multi_destroy(trpl_multi_t *h)
{
    if (!h) return TRPL_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)trpl_multi_synchronize(h);
    const RcclApi *api = rccl_api();
    for (int r = 0; r < h->n; r++) {
        (void)hipSetDevice(h->dev[r]);
        if (h->comm[r] && api->dl) (void)api->CommDestroy(h->comm[r]);
        if (h->send[r]) (void)hipFree(h->send[r]);
        if (h->recv[r]) (void)hipFree(h->recv[r]);
        if (h->st[r]) (void)hipStreamDestroy(h->st[r]);
        if (h->ev_in[r]) (void)hipEventDestroy(h->ev_in[r]);
        if (h->ev_out[r]) (void)hipEventDestroy(h->ev_out[r]);
    }
    (void)hipSetDevice(prev);
    delete h;
    return TRPL_OK;
}

// The handle works on its own non-blocking streams, which nothing orders against the streams the caller filled X /
// obs / dN on or reads P_full from.  These two calls add that order without a host wait:
//   wait_stream:    everything the handle enqueues on rank r from now on runs after what `stream` holds now
//   release_stream: everything `stream` is given from now on runs after what the handle has enqueued on rank r
static int multi_order(trpl_multi_t *h, int32_t rank, void *stream, bool handle_waits)
{
    if (!h) return api_fail(TRPL_ERR_ARG, "handle must not be NULL");
    if (rank < 0 || rank >= h->n) return api_fail(TRPL_ERR_ARG, "rank %d out of range (%d devices)", rank, h->n);
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = [&]() -> int {
        HIP_TRY(hipSetDevice(h->dev[rank]));
        hipStream_t user = (hipStream_t)stream;
        if (handle_waits) {
            HIP_TRY(hipEventRecord(h->ev_in[rank], user));
            HIP_TRY(hipStreamWaitEvent(h->st[rank], h->ev_in[rank], 0));
        } else {
            HIP_TRY(hipEventRecord(h->ev_out[rank], h->st[rank]));
            HIP_TRY(hipStreamWaitEvent(user, h->ev_out[rank], 0));
        }
        return TRPL_OK;
    }();
    (void)hipSetDevice(prev);
    return rc;
}
int trpl_multi_wait_stream(trpl_multi_t *h, int32_t rank, void *stream) { return multi_order(h, rank, stream, true); }
int trpl_multi_release_stream(trpl_multi_t *h, int32_t rank, void *stream) { return multi_order(h, rank, stream, false); }

int trpl_loglik_multi_dev(trpl_multi_t *h, const double *const *X, int64_t S, int32_t C, const double *lengths_nm,
                          double time_ns, int32_t L, int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter,
                          const double *const *dN, const double *const *obs, const int32_t *const *obs_hi,
                          const double *const *obs_dx, const double *const *obs_h, int64_t obs_ld, const int64_t *n_obs,
                          double *const *P_full, double *const *sse, int32_t *const *status, int64_t *const *iters_total,
                          int32_t *const *floor_col, uint32_t flags)
{
    if (!h) return api_fail(TRPL_ERR_ARG, "handle must not be NULL");
    if (int rc = check_grid(L, T, plT, max_iter, time_ns)) return rc;
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (C < 1 || C > TRPL_MAX_CURVES) return api_fail(TRPL_ERR_ARG, "C=%d must be in [1, %d]", C, TRPL_MAX_CURVES);
    if (!X || !lengths_nm || !dN || !obs || !n_obs || !P_full) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    const bool interp = obs_hi || obs_dx || obs_h;
    if (interp && !(obs_hi && obs_dx && obs_h)) return api_fail(TRPL_ERR_ARG, "obs_hi, obs_dx and obs_h go together");
    if (interp && plT != 1) return api_fail(TRPL_ERR_ARG, "off-grid observations need plT = 1");
    if (int rc = check_variant_flags(flags, L)) return rc;
    if (flags & 0xF00u) return api_fail(TRPL_ERR_UNSUPPORTED, "TRPL_FLAG_BUNDLE couples neighbouring samples: a sharded batch would depend on where it is cut");
    for (int c = 0; c < C; c++)
        if (n_obs[c] < 1 || n_obs[c] > obs_ld)
            return api_fail(TRPL_ERR_ARG, "n_obs[%d]=%lld out of range (obs_ld %lld)", c, (long long)n_obs[c], (long long)obs_ld);
    if (S == 0) return TRPL_OK;
    const int n = h->n;
    for (int r = 0; r < n; r++)                          // every rank receives the gathered vector, also one whose shard is empty
        if (!P_full[r]) return api_fail(TRPL_ERR_ARG, "P_full[%d] must not be NULL", r);
    const int64_t widest = (S + n - 1) / n;
    const RcclApi *api = rccl_api();
    int prev = 0;
    (void)hipGetDevice(&prev);
    flags = pin_variant(flags, S * (int64_t)C, L, loglik_steps(interp, C, n_obs, plT, T));     // see trpl_loglik_multi

    int rc = [&]() -> int {
        if (widest > h->cap) {                       // scratch for the padded exchange, grown on demand
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(h->dev[r]));
                HIP_TRY(hipStreamSynchronize(h->st[r]));
                if (h->send[r]) { HIP_TRY(hipFree(h->send[r])); h->send[r] = nullptr; }
                if (h->recv[r]) { HIP_TRY(hipFree(h->recv[r])); h->recv[r] = nullptr; }
                HIP_TRY(hipMalloc((void **)&h->send[r], (size_t)widest * 8));
                HIP_TRY(hipMalloc((void **)&h->recv[r], (size_t)widest * n * 8));
            }
            h->cap = widest;
        }
        // every device solves its shard; nothing waits
        for (int r = 0; r < n; r++) {
            int64_t lo, hi;
            (void)trpl_shard_bounds(S, n, r, &lo, &hi);
            const int64_t nr = hi - lo;
            HIP_TRY(hipSetDevice(h->dev[r]));
            HIP_TRY(hipMemsetAsync(h->send[r], 0, (size_t)widest * 8, h->st[r]));       // P starts at 0; the pad stays 0
            if (nr == 0) continue;
            if (!X[r] || !dN[r] || !obs[r] || (interp && (!obs_hi[r] || !obs_dx[r] || !obs_h[r])))
                return api_fail(TRPL_ERR_ARG, "NULL device pointer in the tables of rank %d", r);
            double *sse_r = sse ? sse[r] : nullptr;
            DevBuf tmp;                              // the fused call needs somewhere to put the per-curve sums
            if (!sse_r) { HIP_TRY(tmp.alloc((size_t)nr * C * 8, h->st[r])); sse_r = tmp.as<double>(); }
            if (int e = shard_loglik(X[r], nr, C, lengths_nm, time_ns, L, T, plT, tol_exp, max_iter, dN[r], obs[r],
                                     interp ? obs_hi[r] : nullptr, interp ? obs_dx[r] : nullptr,
                                     interp ? obs_h[r] : nullptr, obs_ld, n_obs, h->send[r], sse_r,
                                     status ? status[r] : nullptr, iters_total ? iters_total[r] : nullptr,
                                     floor_col ? floor_col[r] : nullptr, flags, h->st[r]))
                return e;
        }
        // ONE collective: all-gather of `widest` fp64 per rank (RCCL over xGMI), straight into P_full when the
        // shards are equal, through the padded scratch + an unpadding pass otherwise
        const bool force_pad = (flags & TRPL_FLAG_MULTI_FORCE_PAD) != 0;     // tests
        const bool even = S % n == 0 && !force_pad;
        RCCL_TRY(api->GroupStart());
        for (int r = 0; r < n; r++) {
            ncclResult_t e = api->AllGather(h->send[r], even ? (void *)P_full[r] : (void *)h->recv[r], (size_t)widest,
                                            ncclDouble, h->comm[r], h->st[r]);
            if (e != ncclSuccess) { (void)api->GroupEnd(); return api_fail(TRPL_ERR_HIP, "ncclAllGather (rank %d): %s", r, api->GetErrorString(e)); }
        }
        RCCL_TRY(api->GroupEnd());
        if (!even)
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(h->dev[r]));
                const unsigned blocks = (unsigned)std::min<int64_t>((S + 255) / 256, 4096);
                hipLaunchKernelGGL(unpad_kernel, dim3(blocks), dim3(256), 0, h->st[r], h->recv[r], P_full[r], S, n, widest);
                HIP_TRY(hipGetLastError());
            }
        return TRPL_OK;
    }();
    (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"
