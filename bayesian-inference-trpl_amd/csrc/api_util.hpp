// Host-side helpers shared by the translation units that implement include/trpl.h (trpl_api.hip,
// trpl_multi.hip): the thread-local error message, RAII for the private streams and stream-ordered
// allocations of the host-buffer calls, argument checks.  Nothing here throws.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <chrono>

#include "../../include/trpl.h"
#include "trpl_common.hpp"

namespace trpl {

// records the message trpl_last_error() returns on this thread and returns `code` (trpl_api.hip)
int api_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return ::trpl::api_fail(TRPL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// shard that holds sample s when S samples are cut by trpl_shard_bounds into n_shards ranges: the first
// S % n_shards shards hold one sample more
__host__ __device__ inline int64_t shard_of(int64_t S, int64_t n_shards, int64_t s)
{
    const int64_t base = S / n_shards, rem = S % n_shards, cut = rem * (base + 1);
    return s < cut ? s / (base + 1) : rem + (s - cut) / (base ? base : 1);
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Every host-buffer call works on a private non-blocking stream with stream-ordered allocations, so
// that calls issued from different host threads (or for different devices) overlap on the GPU: nothing
// synchronises the whole device.  Declare the CallScope before the DevBufs of a call: the buffers
// are released (hipFreeAsync) first, then the scope drains and destroys the stream.
struct CallScope {
    hipStream_t st = nullptr;
    hipError_t open() { return hipStreamCreateWithFlags(&st, hipStreamNonBlocking); }
    ~CallScope()
    {
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    }
};
struct DevBuf {                      // RAII device allocation for the host-buffer calls
    void *p = nullptr;
    hipStream_t st = nullptr;
    void release() { if (p) (void)hipFreeAsync(p, st); p = nullptr; }
    ~DevBuf() { release(); }
    hipError_t alloc(size_t n, hipStream_t s) { st = s; return hipMallocAsync(&p, n ? n : 1, s); }
    template <typename T> T *as() { return (T *)p; }
};

// Thresholds (bytes) above which a host-buffer call pins (HostPin) / maps (HostMap) the caller's memory for its duration.
// Measured with environment overrides in round 2 (profiles/r2_dropin_simulate.txt); constants since round 5: the
// library reads no process-wide switch but TRPL_RCCL_LIBRARY.
constexpr long long kHostPinMinBytes = (long long)8 << 20;
constexpr long long kHostDirectMinBytes = (long long)8 << 20;

// The caller's (pageable) host buffer pinned for the duration of a call, so that copies to and from it are
// real asynchronous DMA at PCIe rate instead of being staged through the runtime's bounce buffers.  Pinning
// costs time per page, so only large buffers are worth it; a refused registration (memory that cannot be
// page-locked) is not an error -- the copy then takes the pageable path.  Memory that is already pinned
// (hipHostMalloc, or registered by the caller) is left alone.
// Declare a HostPin / HostMap BEFORE the call's CallScope: it must outlive the stream's last operation.
struct HostPin {
    void *p = nullptr;
    bool pinned = false;
    static bool already_pinned(const void *ptr)
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ptr) == hipSuccess) return at.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        return false;
    }
    void pin(const void *ptr, size_t bytes, unsigned flags = hipHostRegisterDefault)
    {
        constexpr long long min_bytes = kHostPinMinBytes;
        if (!ptr || min_bytes < 0 || bytes < (size_t)min_bytes || already_pinned(ptr)) return;
        if (hipHostRegister((void *)ptr, bytes, flags) == hipSuccess) { p = (void *)ptr; pinned = true; }
        else (void)hipGetLastError();            // clear the sticky error of a refused registration
    }
    ~HostPin() { if (pinned) (void)hipHostUnregister(p); }
};

// The caller's host buffer mapped into the device's address space for the duration of a call: a kernel
// writes its output straight into it across PCIe (no device copy of the matrix, no copy after the kernel).
struct HostMap {
    HostPin pin_;
    // device-visible alias of [ptr, ptr + bytes), or nullptr (too small, refused, switched off)
    void *map(void *ptr, size_t bytes)
    {
        constexpr long long min_bytes = kHostDirectMinBytes;
        if (!ptr || min_bytes < 0 || bytes < (size_t)min_bytes) return nullptr;
        if (!HostPin::already_pinned(ptr)) {
            if (hipHostRegister(ptr, bytes, hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            pin_.p = ptr; pin_.pinned = true;
        }
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, ptr, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return d;
    }
};

// Optional profiler ranges (ROCTx) around the host-buffer entry points, so that a marker trace of an UNMODIFIED caller
// (`rocprofv3 --marker-trace --kernel-trace -- python parallel_bayes_gpu.py`) shows the reference's three timed phases
// by name -- pvSim (pvSimPCR.py:378-381), fastlog (probs.py:79-84), prob (probs.py:51-61) -- instead of anonymous kernels
// and copies.  No link dependency (like RCCL): the two symbols are taken from whatever the process already exports
// (rocprofv3 --marker-trace preloads librocprofiler-sdk-roctx.so), else from libroctx64.so if dlopen finds it; absent
// => every range is a no-op (one predictable branch per call).  trpl_api.hip holds the binding.
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
const RoctxApi &roctx();
struct ProfRange {
    bool on;
    explicit ProfRange(const char *name) : on(roctx().push != nullptr) { if (on) (void)roctx().push(name); }
    ~ProfRange() { if (on) (void)roctx().pop(); }
    ProfRange(const ProfRange &) = delete;
    ProfRange &operator=(const ProfRange &) = delete;
};

int select_device(int32_t device);          // hipSetDevice with range check (trpl_api.hip)
int check_grid(int32_t L, int64_t T, int32_t plT, int32_t max_iter, double time_ns);
// the observation brackets of trpl_loglik_obs are host data in the host-buffer calls: sorted, in [1, T]
int check_brackets(const int32_t *obs_hi, const double *obs_dx, const double *obs_h, int32_t C, int64_t obs_ld,
                   const int64_t *n_obs, int64_t T);
// which FAST kernel a logical batch of nsys systems runs (flags may force it); see trpl_kernel_variant
bool pick_pair_kernel(int64_t nsys, int32_t L, int64_t steps, uint32_t flags);
int check_variant_flags(uint32_t flags, int32_t L);
// every flag / shape combination a stepper launch refuses (trpl_api.hip); launch() and trpl_kernel_name share it
int check_launch(uint32_t flags, int32_t L, int64_t steps, bool snap, bool resume);
// flags with the kernel variant of the logical batch pinned (TRPL_FLAG_KERNEL_PAIR / _SINGLE set)
uint32_t pin_variant(uint32_t flags, int64_t nsys, int32_t L, int64_t steps);
// time steps a likelihood launch takes: up to the last observation (on-grid), T off-grid
int64_t loglik_steps(bool interp, int32_t C, const int64_t *n_obs, int32_t plT, int64_t T);

}  // namespace trpl
