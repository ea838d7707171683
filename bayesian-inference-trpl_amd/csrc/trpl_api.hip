// extern "C" entry points of libtrpl_hip.so (declared in include/trpl.h).  Host-side glue only:
// argument checks, the per-curve constants of pvSim (pvSimPCR.py:314-331), staging for the
// host-buffer calls, launches.  Nothing here throws across the ABI.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "api_util.hpp"
#include "crosslane.hpp"      // TRPL_PAIR_OPTIMISTIC (trpl_kernel_name)

namespace trpl {

namespace { thread_local char g_err[512] = ""; }

int api_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const RoctxApi &roctx()
{
    static const RoctxApi api = [] {
        RoctxApi a;
        void *push = dlsym(RTLD_DEFAULT, "roctxRangePushA"), *pop = dlsym(RTLD_DEFAULT, "roctxRangePop");
        if (!push || !pop) {
            void *dl = nullptr;
            for (const char *n : {"libroctx64.so.4", "libroctx64.so"})
                if ((dl = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
            push = dl ? dlsym(dl, "roctxRangePushA") : nullptr;
            pop = dl ? dlsym(dl, "roctxRangePop") : nullptr;
        }
        if (push && pop) {
            a.push = (int (*)(const char *))push;
            a.pop = (int (*)())pop;
        }
        return a;
    }();
    return api;
}

int check_grid(int32_t L, int64_t T, int32_t plT, int32_t max_iter, double time_ns)
{
    if (!pow2(L) || L < 4 || L > 512) return api_fail(TRPL_ERR_ARG, "L=%d must be a power of two in [4, 512]", L);
    if (T < 1) return api_fail(TRPL_ERR_ARG, "T=%lld must be >= 1", (long long)T);
    // the steppers keep steps and PL columns in 32-bit scalars: with T <= 2^30 - 16 and plT clamped to T + 1 (kernel_plT)
    // none of t + plT, pl_col * plT or (t0 + plT - 1) / plT can leave the int32 range
    if (T > 0x3ffffff0LL) return api_fail(TRPL_ERR_ARG, "T=%lld is too large (at most 2^30 - 16 steps)", (long long)T);
    if (plT < 1) return api_fail(TRPL_ERR_ARG, "plT=%d must be >= 1", plT);
    if (max_iter < 1) return api_fail(TRPL_ERR_ARG, "max_iter=%d must be >= 1", max_iter);
    if (!(time_ns > 0)) return api_fail(TRPL_ERR_ARG, "time_ns must be > 0");
    return TRPL_OK;
}

// plT as the kernels see it: a stride beyond the window stores column 0 only, whatever its size
static inline int32_t kernel_plT(int32_t plT, int64_t T) { return (int64_t)plT > T + 1 ? (int32_t)(T + 1) : plT; }

int check_brackets(const int32_t *obs_hi, const double *obs_dx, const double *obs_h, int32_t C, int64_t obs_ld,
                   const int64_t *n_obs, int64_t T)
{
    for (int c = 0; c < C; c++)
        for (int64_t i = 0; i < n_obs[c] && i < obs_ld; i++) {
            const int64_t at = (int64_t)c * obs_ld + i;
            const int32_t h = obs_hi[at];
            if (h < 1 || h > T || (i && h < obs_hi[at - 1]))
                return api_fail(TRPL_ERR_ARG, "obs_hi[%d][%lld]=%d must be sorted and in [1, T]", c, (long long)i, h);
            if (!(obs_h[at] > 0) || !(obs_dx[at] >= 0) || !(obs_dx[at] <= obs_h[at]))
                return api_fail(TRPL_ERR_ARG, "observation %lld of curve %d: need 0 <= obs_dx <= obs_h and obs_h > 0",
                                (long long)i, c);
        }
    return TRPL_OK;
}

int64_t loglik_steps(bool interp, int32_t C, const int64_t *n_obs, int32_t plT, int64_t T)
{
    if (interp) return T;                            // the kernel stops at the last observation (PlSink::t_last)
    int64_t steps = 0;
    for (int c = 0; c < C; c++) steps = std::max<int64_t>(steps, (n_obs[c] - 1) * (int64_t)plT);
    return steps;
}

namespace {

// pvSim's non-dimensionalisation (pvSimPCR.py:314-331, :393).  Python's float `**` is C pow().
void curve_const(double length, double time_ns, int L, int64_t T, trpl::CurveConst &cc)
{
    const double dx = length / L, dt = time_ns / (double)T;
    const double dx3 = pow(dx, 3.0), dtdx = dt / dx, dtdx2 = dtdx / dx;
    const double dtdx6 = dt / pow(dx, 6.0);
    const double s[12] = {dx3, dx3, dtdx2, dtdx2, dtdx2 / dx, dtdx, dtdx, dtdx6, dtdx6, 1 / dt, 1 / dt, 1 / dx};
    memcpy(cc.scales, s, sizeof s);
    cc.dx3 = dx3;
    cc.plnorm = pow(dx, 2.0) * dt;
    cc.dx = dx;
    cc.n_obs = 0;
}

// FAST, L = 128 has two kernels: one system per wavefront (3 waves per SIMD: 12 systems per CU in
// flight) and two systems per wavefront (2 waves per SIMD: 16 systems per CU, +25..40 % throughput once
// the chip is kept full, but ~15 % slower per wave when it is not).  A launch's duration is set by its slowest
// chain of systems, so the paired kernel only pays when the chip stays full for most of the launch: more
// systems than the one-system kernel holds at once, and -- for short windows, where the first time steps'
// 20-900 inner iterations make the work per system very uneven -- at least two and a half such fills.
// Measured crossover (MI355X, Power_scan, round-2 kernels, tools/small_launch_probe.py; pair / single time):
//   steps = 8000:  1024 systems 1.15, 3072: 1.01, 4096: 0.91, 8192: 0.86, 12 288: 0.82
//   steps = 1000:  4096: 0.96, 6144: 0.99, 8192: 0.92, 12 288: 0.88
// TRPL_FLAG_KERNEL_PAIR / _SINGLE force the choice per call.
bool use_pair_kernel(int64_t nsys, int64_t steps)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        static thread_local int cached_dev = -1, cached_cus = 256;
        if (cached_dev != dev) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached_cus = n;
            cached_dev = dev;
        }
        cus = cached_cus;
    }
    const int64_t fill = (int64_t)cus * 12;
    return nsys > fill && (steps >= 4000 || 2 * nsys >= 5 * fill);
}

constexpr uint32_t kVariantBits = TRPL_FLAG_KERNEL_PAIR | TRPL_FLAG_KERNEL_SINGLE;

// Who shares a wavefront in the paired kernel.  Two curves of ONE sample have the same material parameters and need
// similar iteration counts step by step, which is what a pair pays for (max of the two per time step); adjacent
// samples of one curve do not.  Curves are grouped by identical grid (thickness) and observation count; inside a
// group consecutive curves -- neighbouring excitation powers in the reference's files -- are paired for each of the
// two samples of a period; the first curve of a group of odd size pairs with itself across the two samples.  Every
// (curve, sample) of the period appears exactly once.  A system's bits do not depend on its partner (tested), so the
// table is purely a scheduling matter; TRPL_FLAG_PAIR_ADJACENT switches it off for A/B measurements.
void build_pair_table(trpl::StepArgs &a)
{
    a.pair_n = 0;
    if ((a.flags & TRPL_FLAG_PAIR_ADJACENT) || a.C < 2 || a.obs_hi != nullptr || a.pl != nullptr) return;
    bool used[trpl::kMaxCurves] = {};
    int k = 0;
    for (int c0 = 0; c0 < a.C; c0++) {
        if (used[c0]) continue;
        int group[trpl::kMaxCurves], n = 0;
        for (int c = c0; c < a.C; c++)
            if (!used[c] && memcmp(a.curve[c].scales, a.curve[c0].scales, sizeof a.curve[c].scales) == 0 &&
                a.curve[c].plnorm == a.curve[c0].plnorm && a.curve[c].n_obs == a.curve[c0].n_obs) {
                group[n++] = c;
                used[c] = true;
            }
        for (int i = n & 1; i + 1 < n; i += 2)
            for (int off = 0; off < 2; off++) {
                a.pair_cA[k] = (uint8_t)group[i]; a.pair_oA[k] = (uint8_t)off;
                a.pair_cB[k] = (uint8_t)group[i + 1]; a.pair_oB[k] = (uint8_t)off;
                k++;
            }
        if (n & 1) {
            a.pair_cA[k] = a.pair_cB[k] = (uint8_t)group[0]; a.pair_oA[k] = 0; a.pair_oB[k] = 1;
            k++;
        }
    }
    a.pair_n = k;                                   // == a.C
}

}  // namespace

int check_variant_flags(uint32_t flags, int32_t L)
{
    if ((flags & kVariantBits) == kVariantBits)
        return api_fail(TRPL_ERR_ARG, "TRPL_FLAG_KERNEL_PAIR and TRPL_FLAG_KERNEL_SINGLE exclude each other");
    if ((flags & TRPL_FLAG_KERNEL_PAIR) && (L != 128 || (flags & (TRPL_FLAG_STRICT | TRPL_FLAG_FP32 | TRPL_FLAG_MIXED))))
        return api_fail(TRPL_ERR_ARG, "TRPL_FLAG_KERNEL_PAIR needs L = 128 (got %d) without TRPL_FLAG_STRICT / _FP32 / _MIXED", L);
    return TRPL_OK;
}

bool pick_pair_kernel(int64_t nsys, int32_t L, int64_t steps, uint32_t flags)
{
    if (L != 128 || (flags & (TRPL_FLAG_STRICT | TRPL_FLAG_FP32 | TRPL_FLAG_MIXED))) return false;
    if (flags & 0xF00u) return false;                          // TRPL_FLAG_BUNDLE(m > 1): one system per wavefront, one bundle per workgroup
    if (flags & TRPL_FLAG_KERNEL_PAIR) return true;
    if (flags & TRPL_FLAG_KERNEL_SINGLE) return false;
    return use_pair_kernel(nsys, steps);
}

uint32_t pin_variant(uint32_t flags, int64_t nsys, int32_t L, int64_t steps)
{
    if (flags & kVariantBits) return flags;
    return flags | (pick_pair_kernel(nsys, L, steps, flags) ? TRPL_FLAG_KERNEL_PAIR : TRPL_FLAG_KERNEL_SINGLE);
}

// Everything a stepper launch refuses because of its flags and shape, in ONE place: launch() and trpl_kernel_name run the
// same checks, so a kernel name is only ever returned for an instantiation that exists and that the launch would run.
// snap: state snapshots requested; resume: the launch continues from a checkpoint.
int check_launch(uint32_t flags, int32_t L, int64_t steps, bool snap, bool resume)
{
    if (!pow2(L) || L < 4 || L > 512) return api_fail(TRPL_ERR_ARG, "L=%d must be a power of two in [4, 512]", L);
    if (int rc = check_variant_flags(flags, L)) return rc;
    if (((flags >> 14) & 7u) > 5u) return api_fail(TRPL_ERR_ARG, "TRPL_FLAG_BDF_ORDER(%u): the order cap must be 1 .. 5 (0: the reference's ramp)", (flags >> 14) & 7u);
    const int32_t bundle = (int32_t)((flags >> 8) & 0xF) + 1;             // TRPL_FLAG_BUNDLE(m)
#ifndef TRPL_EXPERIMENTAL
    if (flags & (TRPL_FLAG_MIXED | TRPL_FLAG_HIST32))
        return api_fail(TRPL_ERR_UNSUPPORTED, "%s: this library was built without the experimental steppers (measured and rejected, "
                        "DESIGN.md section 7); rebuild with `make EXPERIMENTAL=1` to run them",
                        (flags & TRPL_FLAG_MIXED) ? "TRPL_FLAG_MIXED" : "TRPL_FLAG_HIST32");
#endif
    if (flags & TRPL_FLAG_HIST32) {
        if (flags & (TRPL_FLAG_STRICT | TRPL_FLAG_FP32 | TRPL_FLAG_MIXED | TRPL_FLAG_KERNEL_PAIR))
            return api_fail(TRPL_ERR_ARG, "TRPL_FLAG_HIST32 excludes TRPL_FLAG_STRICT, TRPL_FLAG_FP32, TRPL_FLAG_MIXED and TRPL_FLAG_KERNEL_PAIR");
        if (L != 256 && L != 512) return api_fail(TRPL_ERR_UNSUPPORTED, "TRPL_FLAG_HIST32: the fp32-difference history is built for L = 256 and 512 (got %d)", L);
        if (snap || resume || bundle > 1)
            return api_fail(TRPL_ERR_UNSUPPORTED, "snapshots, resume and bundles are not available with TRPL_FLAG_HIST32");
    }
    if (bundle > 1 && (flags & (TRPL_FLAG_FP32 | TRPL_FLAG_MIXED | TRPL_FLAG_KERNEL_PAIR)))
        return api_fail(TRPL_ERR_ARG, "TRPL_FLAG_BUNDLE goes with TRPL_FLAG_STRICT or the plain fp64 one-system stepper only");
    if (bundle > 1 && !(flags & TRPL_FLAG_STRICT) && L > 128)
        return api_fail(TRPL_ERR_UNSUPPORTED, "TRPL_FLAG_BUNDLE without TRPL_FLAG_STRICT is built for L <= 128 (got %d)", L);
    if (bundle > trpl::bundle_cap(L))
        return api_fail(TRPL_ERR_ARG, "TRPL_FLAG_BUNDLE(%d): at most %d systems per bundle at L = %d", bundle, trpl::bundle_cap(L), L);
    if (flags & TRPL_FLAG_FP32) {
        if (flags & (TRPL_FLAG_STRICT | TRPL_FLAG_MIXED)) return api_fail(TRPL_ERR_ARG, "TRPL_FLAG_FP32 excludes TRPL_FLAG_STRICT and TRPL_FLAG_MIXED");
        if (L < 128) return api_fail(TRPL_ERR_UNSUPPORTED, "the fp32 stepper is built for L >= 128 (got %d)", L);
        if (snap) return api_fail(TRPL_ERR_UNSUPPORTED, "state snapshots are not available with TRPL_FLAG_FP32");
        if (steps > TRPL_FP32_MAX_STEPS && !(flags & TRPL_FLAG_FP32_LONG))
            return api_fail(TRPL_ERR_UNSUPPORTED, "TRPL_FLAG_FP32 over %lld time steps: an fp32 state loses the decay beyond ~%d steps "
                            "(PL errors of percents, then tens of percents: include/trpl.h); use the fp64 path, or add "
                            "TRPL_FLAG_FP32_LONG (Python wrappers: fp32=\"long\") for a screening pass", (long long)steps, TRPL_FP32_MAX_STEPS);
    }
    if (flags & TRPL_FLAG_MIXED) {
        if (flags & TRPL_FLAG_STRICT) return api_fail(TRPL_ERR_ARG, "TRPL_FLAG_MIXED and TRPL_FLAG_STRICT exclude each other");
        if (L < 128) return api_fail(TRPL_ERR_UNSUPPORTED, "the mixed-precision stepper is built for L >= 128 (got %d)", L);
    }
    return TRPL_OK;
}

int select_device(int32_t device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return api_fail(TRPL_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return api_fail(TRPL_ERR_ARG, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    return TRPL_OK;
}

}  // namespace trpl

using namespace trpl;

namespace {

// steps: how many time steps the launch will take (T, or up to the last observation in likelihood mode)
int launch(const trpl::StepArgs &a_in, uint32_t flags, hipStream_t st, int64_t steps)
{
    if (int rc = check_launch(flags, a_in.L, steps, a_in.n_snap > 0, a_in.resN != nullptr)) return rc;
    trpl::StepArgs a = a_in;
    a.bundle = (int32_t)((flags >> 8) & 0xF) + 1;             // TRPL_FLAG_BUNDLE(m)
    if (flags & TRPL_FLAG_FP32) {
        hipError_t e32 = trpl::launch_stepper_f32(a, st);
        if (e32 != hipSuccess) return api_fail(TRPL_ERR_HIP, "fp32 stepper launch: %s", hipGetErrorString(e32));
        return TRPL_OK;
    }
#ifdef TRPL_EXPERIMENTAL
    if (flags & TRPL_FLAG_MIXED) {
        hipError_t em = trpl::launch_stepper_mixed(a, st);
        if (em != hipSuccess) return api_fail(TRPL_ERR_HIP, "mixed stepper launch: %s", hipGetErrorString(em));
        return TRPL_OK;
    }
    if (flags & TRPL_FLAG_HIST32) {
        hipError_t eh = trpl::launch_stepper_hist32(a, st);
        if (eh != hipSuccess) return api_fail(TRPL_ERR_HIP, "hist32 stepper launch: %s", hipGetErrorString(eh));
        return TRPL_OK;
    }
#endif
    if (pick_pair_kernel(a.S * a.C, a.L, steps, flags)) {
        build_pair_table(a);
        hipError_t ep = trpl::launch_stepper_pair(a, st);
        if (ep != hipSuccess) return api_fail(TRPL_ERR_HIP, "pair stepper launch: %s", hipGetErrorString(ep));
        return TRPL_OK;
    }
    hipError_t e = (flags & TRPL_FLAG_STRICT) ? trpl::launch_stepper_strict(a, st) : trpl::launch_stepper_fast(a, st);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "stepper launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

}  // namespace

extern "C" {

int trpl_abi_version(void) { return TRPL_ABI_VERSION; }
int trpl_has_experimental(void)
{
#ifdef TRPL_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}
const char *trpl_last_error(void) { return g_err; }

int trpl_kernel_variant(int64_t nsys, int32_t L, int64_t steps, uint32_t flags)
{
    if (flags & TRPL_FLAG_FP32) return TRPL_KERNEL_FP32;
    if (flags & TRPL_FLAG_STRICT) return TRPL_KERNEL_STRICT;
    if (flags & TRPL_FLAG_MIXED) return TRPL_KERNEL_MIXED;
    if (flags & TRPL_FLAG_HIST32) return TRPL_KERNEL_HIST32;
    return pick_pair_kernel(nsys, L, steps, flags) ? TRPL_KERNEL_FAST_PAIR : TRPL_KERNEL_FAST;
}

int trpl_kernel_name(int64_t nsys, int32_t L, int64_t steps, uint32_t flags, int32_t snapshots, char *buf, int64_t buflen)
{
    if (!buf || buflen < 1) return api_fail(TRPL_ERR_ARG, "buf must hold at least one byte");
    buf[0] = 0;
    // the checks of a launch: no name for a combination launch() refuses or for an instantiation that does not exist
    // (`snapshots` covers snapshots AND resume; the fp32 stepper has one instantiation and accepts a resume)
    if (int rc = check_launch(flags, L, steps, snapshots != 0 && !(flags & TRPL_FLAG_FP32), false)) return rc;
    const char *tf[2] = {"false", "true"};
    const int snap = snapshots != 0, bundle = ((flags >> 8) & 0xF) != 0;
    int n;
    if (flags & TRPL_FLAG_FP32)
        n = snprintf(buf, (size_t)buflen, "trpl::f32::stepper_kernel<%d>", L);
    else if (!(flags & (TRPL_FLAG_STRICT | TRPL_FLAG_MIXED | TRPL_FLAG_HIST32)) && pick_pair_kernel(nsys, L, steps, flags))
        n = snprintf(buf, (size_t)buflen, "trpl::pair::stepper_pair_kernel<true, %s, %s>", tf[snap],
                     tf[TRPL_PAIR_OPTIMISTIC != 0 && !(flags & TRPL_FLAG_PAIR_ALWAYS_SEAM)]);
    else
        n = snprintf(buf, (size_t)buflen, "trpl::stepper_kernel<%d, %s, %s, %s, %s, %s>", L, tf[(flags & TRPL_FLAG_STRICT) != 0],
                     tf[snap], tf[(flags & TRPL_FLAG_MIXED) != 0], tf[bundle], tf[(flags & TRPL_FLAG_HIST32) != 0]);
    if (n < 0 || n >= buflen) return api_fail(TRPL_ERR_ARG, "buflen=%lld is too small for the kernel name", (long long)buflen);
    return TRPL_OK;
}

int trpl_pair_table(const double *lengths_nm, const int64_t *n_obs, int32_t C, int32_t L, int64_t T, double time_ns,
                    int32_t *cA, int32_t *oA, int32_t *cB, int32_t *oB)
{
    if (C < 1 || C > trpl::kMaxCurves) return -api_fail(TRPL_ERR_ARG, "C=%d must be in [1, %d]", C, trpl::kMaxCurves);
    if (!lengths_nm || !n_obs || !cA || !oA || !cB || !oB) return -api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (int rc = check_grid(L, T, 1, 1, time_ns)) return -rc;
    trpl::StepArgs a;
    memset(&a, 0, sizeof a);
    a.C = C; a.L = L; a.T = T;
    for (int c = 0; c < C; c++) {
        if (!(lengths_nm[c] > 0)) return -api_fail(TRPL_ERR_ARG, "lengths_nm[%d] must be > 0", c);
        curve_const(lengths_nm[c], time_ns, L, T, a.curve[c]);
        a.curve[c].n_obs = n_obs[c];
    }
    build_pair_table(a);
    for (int k = 0; k < a.pair_n; k++) { cA[k] = a.pair_cA[k]; oA[k] = a.pair_oA[k]; cB[k] = a.pair_cB[k]; oB[k] = a.pair_oB[k]; }
    return a.pair_n;
}

int trpl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* ------------------------------------------------------------------ solve_pl ------------ */
static int solve_pl_dev_impl(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L, int64_t T,
                             int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, int64_t t0,
                             const double *resN, const double *resP, const double *resE, void *plI,
                             int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status, int64_t *iters_total,
                             const int64_t *snap_steps, int32_t n_snap, double *plN, double *plP, double *plE,
                             uint32_t flags, void *stream)
{
    const bool resume = resN || resP || resE;
    if (int rc = check_grid(L, T, plT, max_iter, time_ns)) return rc;
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (n_snap < 0 || n_snap > trpl::kMaxSnaps) return api_fail(TRPL_ERR_ARG, "n_snap=%d must be in [0, %d]", n_snap, trpl::kMaxSnaps);
    if (n_snap > 0 && !snap_steps) return api_fail(TRPL_ERR_ARG, "snap_steps must not be NULL when n_snap > 0");
    if (resume && !(resN && resP && resE)) return api_fail(TRPL_ERR_ARG, "resN, resP and resE go together");
    if (resume && (t0 < 4 || t0 > T)) return api_fail(TRPL_ERR_ARG, "t0=%lld must be in [4, T]: a resume needs five BDF levels", (long long)t0);
    if (resume && (flags & TRPL_FLAG_FP32)) return api_fail(TRPL_ERR_UNSUPPORTED, "resume is not available with TRPL_FLAG_FP32");
    if (S == 0) return TRPL_OK;
    if (!matpar || (!dN && !resume) || !plI) return api_fail(TRPL_ERR_ARG, "matpar, dN and plI must not be NULL");
    if (pl_elem_bytes != 4 && pl_elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "pl_elem_bytes must be 4 or 8");
    if (pl_ld < T / plT + 1) return api_fail(TRPL_ERR_ARG, "pl_ld=%lld < T/plT+1", (long long)pl_ld);
    if (!(length_nm > 0)) return api_fail(TRPL_ERR_ARG, "length_nm must be > 0");
    if (S > 0x7fffffffLL) return api_fail(TRPL_ERR_ARG, "S too large for one launch");
    trpl::StepArgs a;
    memset(&a, 0, sizeof a);
    a.X = matpar; a.xld = 12; a.dN = dN;      // NULL on a resume: the kernels take the state from res* and never read it
    a.pl = plI; a.pl_bytes = pl_elem_bytes; a.pl_ld = pl_ld;
    a.status = status; a.iters_total = iters_total;
    a.S = S; a.C = 1; a.L = L; a.T = T; a.plT = kernel_plT(plT, T); a.MAX = max_iter; a.flags = flags;
    a.TOL = pow(10.0, -(double)tol_exp);                        /* pvSimPCR.py:112 */
    curve_const(length_nm, time_ns, L, T, a.curve[0]);
    if (resume) { a.resN = resN; a.resP = resP; a.resE = resE; a.t0 = t0; }
    if (n_snap > 0 && (plN || plP || plE)) {
        // (step, slot) pairs, steps strictly ascending: the slot of a step is its FIRST position in the
        // caller's list (Legacy/pvSim.py:122 `pT.index(t)`); steps outside [t0, T] are never reached
        a.snapN = plN; a.snapP = plP; a.snapE = plE; a.snap_ld = n_snap;
        for (int i = 0; i < n_snap; i++) {
            const int64_t st = snap_steps[i];
            if (st < (resume ? t0 : 0) || st > T) continue;
            bool seen = false;
            for (int k = 0; k < a.n_snap; k++) seen = seen || a.snap_t[k] == (int32_t)st;
            if (seen) continue;
            int at = a.n_snap++;
            while (at > 0 && a.snap_t[at - 1] > (int32_t)st) {
                a.snap_t[at] = a.snap_t[at - 1]; a.snap_slot[at] = a.snap_slot[at - 1]; at--;
            }
            a.snap_t[at] = (int32_t)st; a.snap_slot[at] = i;
        }
    }
    return launch(a, flags, (hipStream_t)stream, T - (resume ? t0 : 0));
}

int trpl_solve_pl_snap_dev(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L, int64_t T,
                           int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, void *plI,
                           int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status, int64_t *iters_total,
                           const int64_t *snap_steps, int32_t n_snap, double *plN, double *plP, double *plE,
                           uint32_t flags, void *stream)
{
    return solve_pl_dev_impl(matpar, S, length_nm, time_ns, L, T, plT, tol_exp, max_iter, dN, 0, nullptr, nullptr, nullptr,
                             plI, pl_elem_bytes, pl_ld, status, iters_total, snap_steps, n_snap, plN, plP, plE, flags, stream);
}

int trpl_solve_pl_resume_dev(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L, int64_t T,
                             int32_t plT, int32_t tol_exp, int32_t max_iter, int64_t t0, const double *resN,
                             const double *resP, const double *resE, void *plI, int32_t pl_elem_bytes, int64_t pl_ld,
                             int32_t *status, int64_t *iters_total, const int64_t *snap_steps, int32_t n_snap,
                             double *plN, double *plP, double *plE, uint32_t flags, void *stream)
{
    if (!resN || !resP || !resE) return api_fail(TRPL_ERR_ARG, "resN, resP and resE must not be NULL");
    return solve_pl_dev_impl(matpar, S, length_nm, time_ns, L, T, plT, tol_exp, max_iter, nullptr, t0, resN, resP, resE, plI,
                             pl_elem_bytes, pl_ld, status, iters_total, snap_steps, n_snap, plN, plP, plE, flags, stream);
}

int trpl_solve_pl_dev(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L, int64_t T,
                      int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, void *plI,
                      int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status, int64_t *iters_total, uint32_t flags,
                      void *stream)
{
    return trpl_solve_pl_snap_dev(matpar, S, length_nm, time_ns, L, T, plT, tol_exp, max_iter, dN, plI, pl_elem_bytes,
                                  pl_ld, status, iters_total, nullptr, 0, nullptr, nullptr, nullptr, flags, stream);
}

static int solve_pl_host_impl(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L, int64_t T,
                              int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, int64_t t0,
                              const double *resN, const double *resP, const double *resE, void *plI,
                              int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status, int64_t *iters_total,
                              const int64_t *snap_steps, int32_t n_snap, double *plN, double *plP, double *plE,
                              uint32_t flags, int32_t device, double *seconds)
{
    const bool resume = resN || resP || resE;
    if (resume && !(resN && resP && resE)) return api_fail(TRPL_ERR_ARG, "resN, resP and resE go together");
    if (int rc = check_grid(L, T, plT, max_iter, time_ns)) return rc;
    if (pl_elem_bytes != 4 && pl_elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "pl_elem_bytes must be 4 or 8");
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (n_snap < 0 || n_snap > trpl::kMaxSnaps) return api_fail(TRPL_ERR_ARG, "n_snap=%d must be in [0, %d]", n_snap, trpl::kMaxSnaps);
    if (seconds) *seconds = 0.0;
    if (S == 0) return TRPL_OK;
    if (!matpar || (!dN && !resume) || !plI) return api_fail(TRPL_ERR_ARG, "matpar, dN and plI must not be NULL");
    const int64_t ncol = T / plT + 1;
    if (pl_ld < ncol) return api_fail(TRPL_ERR_ARG, "pl_ld=%lld < T/plT+1", (long long)pl_ld);
    if (int rc = select_device(device)) return rc;
    // A large PL matrix is written by the kernel STRAIGHT INTO the caller's buffer (pinned and mapped for
    // the duration of the call): the stores cross PCIe while the time-stepping goes on (a 1024 x 80 001 fp32
    // block is 328 MB over a ~0.3 s kernel, ~1 GB/s), so there is no device copy of the matrix and no
    // device-to-host copy after the kernel.  Small or unmappable buffers are staged through device memory.
    HostMap map;
    CallScope cs;
    HIP_TRY(cs.open());
    const size_t span = ((size_t)(S - 1) * pl_ld + ncol) * pl_elem_bytes;
    void *pl_direct = map.map(plI, span);
    DevBuf dm, dn, dp, ds, di, sN, sP, sE, rN, rP, rE;
    HIP_TRY(dm.alloc((size_t)S * 12 * 8, cs.st));
    HIP_TRY(dn.alloc((size_t)L * 8, cs.st));
    if (resume) {                                    // the five BDF levels of every system, solver units
        const size_t bNP = (size_t)S * 5 * L * 8, bE = (size_t)S * 5 * (L + 1) * 8;
        HIP_TRY(rN.alloc(bNP, cs.st)); HIP_TRY(rP.alloc(bNP, cs.st)); HIP_TRY(rE.alloc(bE, cs.st));
        HIP_TRY(hipMemcpyAsync(rN.p, resN, bNP, hipMemcpyHostToDevice, cs.st));
        HIP_TRY(hipMemcpyAsync(rP.p, resP, bNP, hipMemcpyHostToDevice, cs.st));
        HIP_TRY(hipMemcpyAsync(rE.p, resE, bE, hipMemcpyHostToDevice, cs.st));
    }
    if (!pl_direct) HIP_TRY(dp.alloc((size_t)S * ncol * pl_elem_bytes, cs.st));
    HIP_TRY(ds.alloc((size_t)S * 4, cs.st));
    HIP_TRY(di.alloc((size_t)S * 8, cs.st));
    HIP_TRY(hipMemcpyAsync(dm.p, matpar, (size_t)S * 12 * 8, hipMemcpyHostToDevice, cs.st));
    if (dN) HIP_TRY(hipMemcpyAsync(dn.p, dN, (size_t)L * 8, hipMemcpyHostToDevice, cs.st));
    if (resume && !pl_direct)                        // the staged PL matrix starts as the caller's: columns before t0 are kept
        HIP_TRY(hipMemcpy2DAsync(dp.p, (size_t)ncol * pl_elem_bytes, plI, (size_t)pl_ld * pl_elem_bytes,
                                 (size_t)ncol * pl_elem_bytes, (size_t)S, hipMemcpyHostToDevice, cs.st));
    // snapshot buffers start as copies of the caller's (pvSimPCR.py:366-368): unfilled slots keep their contents
    const size_t nNP = (size_t)S * n_snap * L * 8, nE = (size_t)S * n_snap * (L + 1) * 8;
    if (n_snap > 0 && plN) { HIP_TRY(sN.alloc(nNP, cs.st)); HIP_TRY(hipMemcpyAsync(sN.p, plN, nNP, hipMemcpyHostToDevice, cs.st)); }
    if (n_snap > 0 && plP) { HIP_TRY(sP.alloc(nNP, cs.st)); HIP_TRY(hipMemcpyAsync(sP.p, plP, nNP, hipMemcpyHostToDevice, cs.st)); }
    if (n_snap > 0 && plE) { HIP_TRY(sE.alloc(nE, cs.st)); HIP_TRY(hipMemcpyAsync(sE.p, plE, nE, hipMemcpyHostToDevice, cs.st)); }
    const double tic = now_s();
    if (int rc = solve_pl_dev_impl(dm.as<double>(), S, length_nm, time_ns, L, T, plT, tol_exp, max_iter,
                                   dn.as<double>(), t0, rN.as<double>(), rP.as<double>(), rE.as<double>(),
                                   pl_direct ? pl_direct : dp.p, pl_elem_bytes, pl_direct ? pl_ld : ncol,
                                   ds.as<int32_t>(), di.as<int64_t>(), snap_steps, n_snap, sN.as<double>(), sP.as<double>(),
                                   sE.as<double>(), flags, cs.st))
        return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    if (seconds) *seconds = now_s() - tic;                      /* pvSimPCR.py:378-381 */
    if (!pl_direct)
        HIP_TRY(hipMemcpy2DAsync(plI, (size_t)pl_ld * pl_elem_bytes, dp.p, (size_t)ncol * pl_elem_bytes,
                                 (size_t)ncol * pl_elem_bytes, (size_t)S, hipMemcpyDeviceToHost, cs.st));
    if (status) HIP_TRY(hipMemcpyAsync(status, ds.p, (size_t)S * 4, hipMemcpyDeviceToHost, cs.st));
    if (iters_total) HIP_TRY(hipMemcpyAsync(iters_total, di.p, (size_t)S * 8, hipMemcpyDeviceToHost, cs.st));
    if (sN.p) HIP_TRY(hipMemcpyAsync(plN, sN.p, nNP, hipMemcpyDeviceToHost, cs.st));
    if (sP.p) HIP_TRY(hipMemcpyAsync(plP, sP.p, nNP, hipMemcpyDeviceToHost, cs.st));
    if (sE.p) HIP_TRY(hipMemcpyAsync(plE, sE.p, nE, hipMemcpyDeviceToHost, cs.st));
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

int trpl_solve_pl_snap(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L, int64_t T,
                       int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, void *plI,
                       int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status, int64_t *iters_total,
                       const int64_t *snap_steps, int32_t n_snap, double *plN, double *plP, double *plE,
                       uint32_t flags, int32_t device, double *seconds)
{
    ProfRange range("trpl_solve_pl (pvSim)");
    return solve_pl_host_impl(matpar, S, length_nm, time_ns, L, T, plT, tol_exp, max_iter, dN, 0, nullptr, nullptr, nullptr,
                              plI, pl_elem_bytes, pl_ld, status, iters_total, snap_steps, n_snap, plN, plP, plE, flags,
                              device, seconds);
}

int trpl_solve_pl_resume(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L, int64_t T,
                         int32_t plT, int32_t tol_exp, int32_t max_iter, int64_t t0, const double *resN,
                         const double *resP, const double *resE, void *plI, int32_t pl_elem_bytes, int64_t pl_ld,
                         int32_t *status, int64_t *iters_total, const int64_t *snap_steps, int32_t n_snap, double *plN,
                         double *plP, double *plE, uint32_t flags, int32_t device, double *seconds)
{
    ProfRange range("trpl_solve_pl_resume (pvSim, init_mode continue)");
    if (!resN || !resP || !resE) return api_fail(TRPL_ERR_ARG, "resN, resP and resE must not be NULL");
    if (t0 < 4 || t0 > T) return api_fail(TRPL_ERR_ARG, "t0=%lld must be in [4, T]: a resume needs five BDF levels", (long long)t0);
    return solve_pl_host_impl(matpar, S, length_nm, time_ns, L, T, plT, tol_exp, max_iter, nullptr, t0, resN, resP, resE, plI,
                              pl_elem_bytes, pl_ld, status, iters_total, snap_steps, n_snap, plN, plP, plE, flags, device,
                              seconds);
}

int trpl_solve_pl(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L, int64_t T,
                  int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, void *plI,
                  int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status, int64_t *iters_total, uint32_t flags,
                  int32_t device, double *seconds)
{
    return trpl_solve_pl_snap(matpar, S, length_nm, time_ns, L, T, plT, tol_exp, max_iter, dN, plI, pl_elem_bytes, pl_ld,
                              status, iters_total, nullptr, 0, nullptr, nullptr, nullptr, flags, device, seconds);
}

/* ------------------------------------------------------------------ log10 clamp --------- */
int trpl_log10_clamp_dev(void *x, int32_t elem_bytes, int64_t rows, int64_t cols, int64_t ld, double min,
                         void *stream)
{
    if (elem_bytes != 4 && elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "elem_bytes must be 4 or 8");
    if (rows < 0 || cols < 0 || ld < cols) return api_fail(TRPL_ERR_ARG, "bad shape rows=%lld cols=%lld ld=%lld",
                                                       (long long)rows, (long long)cols, (long long)ld);
    if (rows == 0 || cols == 0) return TRPL_OK;
    if (!x) return api_fail(TRPL_ERR_ARG, "x must not be NULL");
    hipError_t e = trpl::launch_log10_clamp(x, elem_bytes, rows, cols, ld, min, (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "log10_clamp launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

int trpl_log10_clamp(void *x, int32_t elem_bytes, int64_t rows, int64_t cols, int64_t ld, double min,
                     int32_t device, double *seconds)
{
    ProfRange range("trpl_log10_clamp (fastlog)");
    if (elem_bytes != 4 && elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "elem_bytes must be 4 or 8");
    if (rows < 0 || cols < 0 || ld < cols) return api_fail(TRPL_ERR_ARG, "bad shape");
    if (seconds) *seconds = 0.0;
    if (rows == 0 || cols == 0) return TRPL_OK;
    if (!x) return api_fail(TRPL_ERR_ARG, "x must not be NULL");
    if (int rc = select_device(device)) return rc;
    HostPin pin;
    CallScope cs;
    HIP_TRY(cs.open());
    const double t0 = now_s();                                   /* probs.py:79: includes the copies */
    DevBuf dx;
    const size_t rowb = (size_t)cols * elem_bytes;
    pin.pin(x, ((size_t)(rows - 1) * ld + cols) * elem_bytes);
    HIP_TRY(dx.alloc(rowb * rows, cs.st));
    HIP_TRY(hipMemcpy2DAsync(dx.p, rowb, x, (size_t)ld * elem_bytes, rowb, (size_t)rows, hipMemcpyHostToDevice, cs.st));
    if (int rc = trpl_log10_clamp_dev(dx.p, elem_bytes, rows, cols, cols, min, cs.st)) return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    HIP_TRY(hipMemcpy2DAsync(x, (size_t)ld * elem_bytes, dx.p, rowb, rowb, (size_t)rows, hipMemcpyDeviceToHost, cs.st));
    if (seconds) *seconds = now_s() - t0;
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

/* ------------------------------------------------------------------ sse accumulate ------ */
int trpl_sse_accumulate_dev(double *P, const void *plI, int32_t elem_bytes, int64_t rows, int64_t n_obs,
                            int64_t ld, const double *values, const double *mag, void *stream)
{
    if (elem_bytes != 4 && elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "elem_bytes must be 4 or 8");
    if (rows < 0 || n_obs < 0 || ld < n_obs) return api_fail(TRPL_ERR_ARG, "bad shape");
    if (rows == 0) return TRPL_OK;
    if (!P || !mag || (n_obs && (!plI || !values))) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    hipError_t e = trpl::launch_sse_accumulate(P, plI, elem_bytes, rows, n_obs, ld, values, mag, (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "sse_accumulate launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

int trpl_sse_accumulate(double *P, const void *plI, int32_t elem_bytes, int64_t rows, int64_t n_obs, int64_t ld,
                        const double *values, const double *mag, int32_t device, double *seconds)
{
    ProfRange range("trpl_sse_accumulate (prob)");
    if (elem_bytes != 4 && elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "elem_bytes must be 4 or 8");
    if (rows < 0 || n_obs < 0 || ld < n_obs) return api_fail(TRPL_ERR_ARG, "bad shape");
    if (seconds) *seconds = 0.0;
    if (rows == 0) return TRPL_OK;
    if (!P || !mag || (n_obs && (!plI || !values))) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (int rc = select_device(device)) return rc;
    HostPin pin;
    CallScope cs;
    HIP_TRY(cs.open());
    const double t0 = now_s();                                   /* probs.py:51 */
    DevBuf dP, dpl, dv, dm;
    const size_t rowb = (size_t)n_obs * elem_bytes;
    if (n_obs) pin.pin(plI, ((size_t)(rows - 1) * ld + n_obs) * elem_bytes);
    HIP_TRY(dP.alloc((size_t)rows * 8, cs.st));
    HIP_TRY(dpl.alloc(rowb * rows, cs.st));
    HIP_TRY(dv.alloc((size_t)n_obs * 8, cs.st));
    HIP_TRY(dm.alloc((size_t)rows * 8, cs.st));
    HIP_TRY(hipMemcpyAsync(dP.p, P, (size_t)rows * 8, hipMemcpyHostToDevice, cs.st));
    if (n_obs) {
        HIP_TRY(hipMemcpy2DAsync(dpl.p, rowb, plI, (size_t)ld * elem_bytes, rowb, (size_t)rows, hipMemcpyHostToDevice, cs.st));
        HIP_TRY(hipMemcpyAsync(dv.p, values, (size_t)n_obs * 8, hipMemcpyHostToDevice, cs.st));
    }
    HIP_TRY(hipMemcpyAsync(dm.p, mag, (size_t)rows * 8, hipMemcpyHostToDevice, cs.st));
    if (int rc = trpl_sse_accumulate_dev(dP.as<double>(), dpl.p, elem_bytes, rows, n_obs, n_obs, dv.as<double>(),
                                         dm.as<double>(), cs.st))
        return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    HIP_TRY(hipMemcpyAsync(P, dP.p, (size_t)rows * 8, hipMemcpyDeviceToHost, cs.st));
    if (seconds) *seconds = now_s() - t0;
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

/* ------------------------------------------------------------------ loglik from stored PL */
int trpl_loglik_from_pl_dev(const void *plI, int32_t elem_bytes, int64_t rows, int64_t ncol, int64_t ld,
                            const double *obs, const int32_t *obs_hi, const double *obs_dx, const double *obs_h,
                            int64_t n_obs, const double *mag, const int32_t *status, double *P, double *sse,
                            uint32_t flags, void *stream)
{
    if (elem_bytes != 4 && elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "elem_bytes must be 4 or 8");
    if (rows < 0 || ncol < 1 || ld < ncol || n_obs < 0) return api_fail(TRPL_ERR_ARG, "bad shape");
    const bool interp = obs_hi || obs_dx || obs_h;
    if (interp && !(obs_hi && obs_dx && obs_h)) return api_fail(TRPL_ERR_ARG, "obs_hi, obs_dx and obs_h go together");
    if (!interp && n_obs > ncol) return api_fail(TRPL_ERR_ARG, "n_obs=%lld exceeds the %lld PL columns", (long long)n_obs, (long long)ncol);
    if (rows > 0x7fffffffLL) return api_fail(TRPL_ERR_ARG, "too many rows for one launch");
    if (rows == 0) return TRPL_OK;
    if (!plI || !mag || (n_obs && !obs) || (!P && !sse)) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    hipError_t e = trpl::launch_pl_loglik(plI, elem_bytes, rows, ld, obs, obs_hi, obs_dx, obs_h, n_obs, mag, status, P, sse, flags,
                                          (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "pl_loglik launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

/* ------------------------------------------------------------------ fused loglik -------- */
static int loglik_dev_impl(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns, int32_t L,
                           int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN,
                           const double *obs, const int32_t *obs_hi, const double *obs_dx, const double *obs_h,
                           int64_t obs_ld, const int64_t *n_obs, double *P, double *sse, int32_t *status,
                           int64_t *iters_total, int32_t *floor_col, uint32_t flags, void *stream)
{
    if (int rc = check_grid(L, T, plT, max_iter, time_ns)) return rc;
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (C < 1 || C > TRPL_MAX_CURVES) return api_fail(TRPL_ERR_ARG, "C=%d must be in [1, %d]", C, TRPL_MAX_CURVES);
    if (S == 0) return TRPL_OK;
    if (!X || !lengths_nm || !dN || !obs || !n_obs || !P || !sse) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    const bool interp = obs_hi || obs_dx || obs_h;
    if (interp && !(obs_hi && obs_dx && obs_h)) return api_fail(TRPL_ERR_ARG, "obs_hi, obs_dx and obs_h go together");
    if (interp && plT != 1) return api_fail(TRPL_ERR_ARG, "off-grid observations need plT = 1");
    const int64_t ncol = T / plT + 1;
    for (int c = 0; c < C; c++) {
        if (!(lengths_nm[c] > 0)) return api_fail(TRPL_ERR_ARG, "lengths_nm[%d] must be > 0", c);
        if (n_obs[c] < 1 || n_obs[c] > obs_ld || (!interp && n_obs[c] > ncol))
            return api_fail(TRPL_ERR_ARG, "n_obs[%d]=%lld out of range (obs_ld %lld, grid columns %lld)", c,
                        (long long)n_obs[c], (long long)obs_ld, (long long)ncol);
    }
    // A launch carries the constants of at most kMaxCurves curves in its argument block; bayeslib.simulate loops over any
    // number of curves (bayeslib.py:117), so more are run as consecutive launches of up to kMaxCurves on the same stream,
    // each writing its own rows of the [C][S] outputs, before ONE reduction over all C in curve order (probs.py:44).  The
    // stepper variant is chosen once from the whole batch, so a system's bits do not depend on the grouping.
    const int64_t steps = loglik_steps(interp, C, n_obs, plT, T);
    if (C > trpl::kMaxCurves) flags = pin_variant(flags, S * (int64_t)C, L, steps);
    for (int c0 = 0; c0 < C; c0 += trpl::kMaxCurves) {
        const int Cg = C - c0 < trpl::kMaxCurves ? C - c0 : trpl::kMaxCurves;
        if (S * (int64_t)Cg > 0x7fffffffLL) return api_fail(TRPL_ERR_ARG, "S*C too large for one launch");
        trpl::StepArgs a;
        memset(&a, 0, sizeof a);
        a.X = X; a.xld = 13; a.dN = dN + (int64_t)c0 * L; a.obs = obs + (int64_t)c0 * obs_ld;
        a.obs_hi = obs_hi ? obs_hi + (int64_t)c0 * obs_ld : nullptr;
        a.obs_dx = obs_dx ? obs_dx + (int64_t)c0 * obs_ld : nullptr;
        a.obs_h = obs_h ? obs_h + (int64_t)c0 * obs_ld : nullptr;
        a.obs_ld = obs_ld; a.sse = sse + (int64_t)c0 * S;
        a.status = status ? status + (int64_t)c0 * S : nullptr;
        a.iters_total = iters_total ? iters_total + (int64_t)c0 * S : nullptr;
        a.floor_col = floor_col ? floor_col + (int64_t)c0 * S : nullptr;
        a.S = S; a.C = Cg; a.L = L; a.T = T; a.plT = kernel_plT(plT, T); a.MAX = max_iter; a.flags = flags;
        a.TOL = pow(10.0, -(double)tol_exp);
        for (int c = 0; c < Cg; c++) {
            curve_const(lengths_nm[c0 + c], time_ns, L, T, a.curve[c]);
            a.curve[c].n_obs = n_obs[c0 + c];
        }
        if (int rc = launch(a, flags, (hipStream_t)stream, steps)) return rc;
    }
    hipError_t e = trpl::launch_reduce_curves(P, sse, S, C, (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "reduce_curves launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

int trpl_loglik_dev(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns, int32_t L,
                    int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN,
                    const double *obs, int64_t obs_ld, const int64_t *n_obs, double *P, double *sse,
                    int32_t *status, int64_t *iters_total, int32_t *floor_col, uint32_t flags, void *stream)
{
    return loglik_dev_impl(X, S, C, lengths_nm, time_ns, L, T, plT, tol_exp, max_iter, dN, obs, nullptr, nullptr,
                           nullptr, obs_ld, n_obs, P, sse, status, iters_total, floor_col, flags, stream);
}

int trpl_loglik_obs_dev(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns, int32_t L,
                        int64_t T, int32_t tol_exp, int32_t max_iter, const double *dN, const double *obs,
                        const int32_t *obs_hi, const double *obs_dx, const double *obs_h, int64_t obs_ld,
                        const int64_t *n_obs, double *P, double *sse, int32_t *status, int64_t *iters_total,
                        int32_t *floor_col, uint32_t flags, void *stream)
{
    if (!obs_hi || !obs_dx || !obs_h) return api_fail(TRPL_ERR_ARG, "obs_hi, obs_dx and obs_h must not be NULL");
    return loglik_dev_impl(X, S, C, lengths_nm, time_ns, L, T, 1, tol_exp, max_iter, dN, obs, obs_hi, obs_dx, obs_h,
                           obs_ld, n_obs, P, sse, status, iters_total, floor_col, flags, stream);
}

static int loglik_host_impl(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns, int32_t L,
                            int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN,
                            const double *obs, const int32_t *obs_hi, const double *obs_dx, const double *obs_h,
                            int64_t obs_ld, const int64_t *n_obs, double *P, double *sse, int32_t *status,
                            int64_t *iters_total, int32_t *floor_col, uint32_t flags, int32_t device, double *seconds)
{
    if (int rc = check_grid(L, T, plT, max_iter, time_ns)) return rc;
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (C < 1 || C > TRPL_MAX_CURVES) return api_fail(TRPL_ERR_ARG, "C=%d must be in [1, %d]", C, TRPL_MAX_CURVES);
    if (seconds) *seconds = 0.0;
    if (S == 0) return TRPL_OK;
    if (!X || !lengths_nm || !dN || !obs || !n_obs || !P) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (obs_ld < 1) return api_fail(TRPL_ERR_ARG, "obs_ld must be >= 1");
    if (int rc = select_device(device)) return rc;
    CallScope cs;
    HIP_TRY(cs.open());
    const bool interp = obs_hi != nullptr;
    if (interp) {                                    // the brackets are host data here: validate them
        if (int rc = check_brackets(obs_hi, obs_dx, obs_h, C, obs_ld, n_obs, T)) return rc;
    }
    DevBuf dX, ddN, dobs, dhi, ddx, dh, dP, dsse, dst, dit, dfl;
    const size_t nsys = (size_t)S * C, nobs = (size_t)C * obs_ld;
    HIP_TRY(dX.alloc((size_t)S * 13 * 8, cs.st));
    HIP_TRY(ddN.alloc((size_t)C * L * 8, cs.st));
    HIP_TRY(dobs.alloc(nobs * 8, cs.st));
    HIP_TRY(dP.alloc((size_t)S * 8, cs.st));
    HIP_TRY(dsse.alloc(nsys * 8, cs.st));
    HIP_TRY(dst.alloc(nsys * 4, cs.st));
    HIP_TRY(dit.alloc(nsys * 8, cs.st));
    if (floor_col) HIP_TRY(dfl.alloc(nsys * 4, cs.st));
    HIP_TRY(hipMemcpyAsync(dX.p, X, (size_t)S * 13 * 8, hipMemcpyHostToDevice, cs.st));
    HIP_TRY(hipMemcpyAsync(ddN.p, dN, (size_t)C * L * 8, hipMemcpyHostToDevice, cs.st));
    HIP_TRY(hipMemcpyAsync(dobs.p, obs, nobs * 8, hipMemcpyHostToDevice, cs.st));
    HIP_TRY(hipMemcpyAsync(dP.p, P, (size_t)S * 8, hipMemcpyHostToDevice, cs.st));
    if (interp) {
        HIP_TRY(dhi.alloc(nobs * 4, cs.st)); HIP_TRY(ddx.alloc(nobs * 8, cs.st)); HIP_TRY(dh.alloc(nobs * 8, cs.st));
        HIP_TRY(hipMemcpyAsync(dhi.p, obs_hi, nobs * 4, hipMemcpyHostToDevice, cs.st));
        HIP_TRY(hipMemcpyAsync(ddx.p, obs_dx, nobs * 8, hipMemcpyHostToDevice, cs.st));
        HIP_TRY(hipMemcpyAsync(dh.p, obs_h, nobs * 8, hipMemcpyHostToDevice, cs.st));
    }
    const double t0 = now_s();
    if (int rc = loglik_dev_impl(dX.as<double>(), S, C, lengths_nm, time_ns, L, T, plT, tol_exp, max_iter,
                                 ddN.as<double>(), dobs.as<double>(), interp ? dhi.as<int32_t>() : nullptr,
                                 interp ? ddx.as<double>() : nullptr, interp ? dh.as<double>() : nullptr, obs_ld, n_obs,
                                 dP.as<double>(), dsse.as<double>(), dst.as<int32_t>(), dit.as<int64_t>(),
                                 dfl.as<int32_t>(), flags, cs.st))
        return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    if (seconds) *seconds = now_s() - t0;
    HIP_TRY(hipMemcpyAsync(P, dP.p, (size_t)S * 8, hipMemcpyDeviceToHost, cs.st));
    if (sse) HIP_TRY(hipMemcpyAsync(sse, dsse.p, nsys * 8, hipMemcpyDeviceToHost, cs.st));
    if (status) HIP_TRY(hipMemcpyAsync(status, dst.p, nsys * 4, hipMemcpyDeviceToHost, cs.st));
    if (iters_total) HIP_TRY(hipMemcpyAsync(iters_total, dit.p, nsys * 8, hipMemcpyDeviceToHost, cs.st));
    if (floor_col) HIP_TRY(hipMemcpyAsync(floor_col, dfl.p, nsys * 4, hipMemcpyDeviceToHost, cs.st));
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

int trpl_loglik(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns, int32_t L,
                int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN, const double *obs,
                int64_t obs_ld, const int64_t *n_obs, double *P, double *sse, int32_t *status,
                int64_t *iters_total, int32_t *floor_col, uint32_t flags, int32_t device, double *seconds)
{
    ProfRange range("trpl_loglik (pvSim + fastlog + prob, fused)");
    return loglik_host_impl(X, S, C, lengths_nm, time_ns, L, T, plT, tol_exp, max_iter, dN, obs, nullptr, nullptr,
                            nullptr, obs_ld, n_obs, P, sse, status, iters_total, floor_col, flags, device, seconds);
}

int trpl_loglik_obs(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns, int32_t L,
                    int64_t T, int32_t tol_exp, int32_t max_iter, const double *dN, const double *obs,
                    const int32_t *obs_hi, const double *obs_dx, const double *obs_h, int64_t obs_ld,
                    const int64_t *n_obs, double *P, double *sse, int32_t *status, int64_t *iters_total,
                    int32_t *floor_col, uint32_t flags, int32_t device, double *seconds)
{
    ProfRange range("trpl_loglik_obs (pvSim + fastlog + griddata + prob, fused)");
    if (!obs_hi || !obs_dx || !obs_h) return api_fail(TRPL_ERR_ARG, "obs_hi, obs_dx and obs_h must not be NULL");
    return loglik_host_impl(X, S, C, lengths_nm, time_ns, L, T, 1, tol_exp, max_iter, dN, obs, obs_hi, obs_dx, obs_h,
                            obs_ld, n_obs, P, sse, status, iters_total, floor_col, flags, device, seconds);
}

/* ------------------------------------------------------------------ host interpolation --- */
int trpl_interp_rows(const void *pl, int32_t elem_bytes, int64_t rows, int64_t ncol, int64_t ld, const int32_t *hi,
                     const double *dx, const double *h, int64_t n_obs, double *out, int64_t out_ld)
{
    if (elem_bytes != 4 && elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "elem_bytes must be 4 or 8");
    if (rows < 0 || n_obs < 0 || ncol < 2 || ld < ncol || out_ld < n_obs) return api_fail(TRPL_ERR_ARG, "bad shape");
    if (rows == 0 || n_obs == 0) return TRPL_OK;
    if (!pl || !hi || !dx || !h || !out) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    for (int64_t i = 0; i < n_obs; i++)
        if (hi[i] < 1 || hi[i] >= ncol) return api_fail(TRPL_ERR_ARG, "hi[%lld]=%d must be in [1, ncol - 1]", (long long)i, hi[i]);
    trpl::interp_rows_any(pl, elem_bytes, rows, ld, hi, dx, h, n_obs, out, out_ld);
    return TRPL_OK;
}

/* ------------------------------------------------------------------ posterior core ------ */
int64_t trpl_posterior_workspace_bytes(int32_t D) { return (int64_t)trpl::posterior_workspace_bytes(D); }

int trpl_posterior_weights_dev(const double *LL, int64_t S, double tf, double *W, double *stats, void *workspace,
                               int64_t workspace_bytes, void *stream)
{
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (S == 0) return TRPL_OK;
    if (!LL || !W || !workspace) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (!(tf > 0)) return api_fail(TRPL_ERR_ARG, "tf must be > 0");
    if (workspace_bytes < (int64_t)trpl::posterior_workspace_bytes(1)) return api_fail(TRPL_ERR_ARG, "workspace too small");
    hipError_t e = trpl::launch_posterior_weights(LL, S, tf, W, stats, (double *)workspace, (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "posterior weights launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

int trpl_posterior_weights(const double *LL, int64_t S, double tf, double *W, double *stats, int32_t device,
                           double *seconds)
{
    if (seconds) *seconds = 0.0;
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (S == 0) return TRPL_OK;
    if (!LL || !W) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (!(tf > 0)) return api_fail(TRPL_ERR_ARG, "tf must be > 0");
    if (int rc = select_device(device)) return rc;
    CallScope cs;
    HIP_TRY(cs.open());
    DevBuf dL, dW, dSt, ws;
    const size_t wsb = trpl::posterior_workspace_bytes(1);
    HIP_TRY(dL.alloc((size_t)S * 8, cs.st)); HIP_TRY(dW.alloc((size_t)S * 8, cs.st)); HIP_TRY(dSt.alloc(16, cs.st)); HIP_TRY(ws.alloc(wsb, cs.st));
    HIP_TRY(hipMemcpyAsync(dL.p, LL, (size_t)S * 8, hipMemcpyHostToDevice, cs.st));
    const double t0 = now_s();
    if (int rc = trpl_posterior_weights_dev(dL.as<double>(), S, tf, dW.as<double>(), dSt.as<double>(), ws.p, (int64_t)wsb, cs.st))
        return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    if (seconds) *seconds = now_s() - t0;
    HIP_TRY(hipMemcpyAsync(W, dW.p, (size_t)S * 8, hipMemcpyDeviceToHost, cs.st));
    if (stats) HIP_TRY(hipMemcpyAsync(stats, dSt.p, 16, hipMemcpyDeviceToHost, cs.st));
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

int trpl_posterior_moments_dev(const double *V, int64_t S, int32_t D, const double *W, const double *mean_in, double *sums,
                               double *central, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (D < 1 || D > 16) return api_fail(TRPL_ERR_ARG, "D=%d must be in [1, 16]", D);
    if (S == 0) return TRPL_OK;
    if (!V || !W || !sums || !central || !workspace) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (workspace_bytes < (int64_t)trpl::posterior_workspace_bytes(D)) return api_fail(TRPL_ERR_ARG, "workspace too small");
    hipError_t e = trpl::launch_posterior_moments(V, W, S, D, mean_in, sums, central, (double *)workspace, (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "posterior moments launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

int trpl_posterior_moments(const double *V, int64_t S, int32_t D, const double *W, const double *mean_in, double *sums,
                           double *central, int32_t device, double *seconds)
{
    if (seconds) *seconds = 0.0;
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (D < 1 || D > 16) return api_fail(TRPL_ERR_ARG, "D=%d must be in [1, 16]", D);
    if (!sums || !central) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    memset(sums, 0, sizeof(double) * (2 + D));
    memset(central, 0, sizeof(double) * D * (D + 2));
    if (S == 0) return TRPL_OK;
    if (!V || !W) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (int rc = select_device(device)) return rc;
    CallScope cs;
    HIP_TRY(cs.open());
    DevBuf dV, dW, dS, dC, dM, ws;
    const size_t wsb = trpl::posterior_workspace_bytes(D);
    if (mean_in) { HIP_TRY(dM.alloc((size_t)D * 8, cs.st)); HIP_TRY(hipMemcpyAsync(dM.p, mean_in, (size_t)D * 8, hipMemcpyHostToDevice, cs.st)); }
    HIP_TRY(dV.alloc((size_t)S * D * 8, cs.st)); HIP_TRY(dW.alloc((size_t)S * 8, cs.st)); HIP_TRY(dS.alloc((2 + D) * 8, cs.st));
    HIP_TRY(dC.alloc((size_t)D * (D + 2) * 8, cs.st)); HIP_TRY(ws.alloc(wsb, cs.st));
    HIP_TRY(hipMemcpyAsync(dV.p, V, (size_t)S * D * 8, hipMemcpyHostToDevice, cs.st));
    HIP_TRY(hipMemcpyAsync(dW.p, W, (size_t)S * 8, hipMemcpyHostToDevice, cs.st));
    const double t0 = now_s();
    if (int rc = trpl_posterior_moments_dev(dV.as<double>(), S, D, dW.as<double>(), mean_in ? dM.as<double>() : nullptr,
                                            dS.as<double>(), dC.as<double>(), ws.p,
                                            (int64_t)wsb, cs.st))
        return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    if (seconds) *seconds = now_s() - t0;
    HIP_TRY(hipMemcpyAsync(sums, dS.p, (2 + D) * 8, hipMemcpyDeviceToHost, cs.st));
    HIP_TRY(hipMemcpyAsync(central, dC.p, (size_t)D * (D + 2) * 8, hipMemcpyDeviceToHost, cs.st));
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

static int check_hist(int64_t S, double xlo, double xhi, int32_t xb, const double *y, double ylo, double yhi, int32_t yb)
{
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (xb < 1 || !(xhi > xlo)) return api_fail(TRPL_ERR_ARG, "x axis needs bins >= 1 and hi > lo");
    if (y && (yb < 1 || !(yhi > ylo))) return api_fail(TRPL_ERR_ARG, "y axis needs bins >= 1 and hi > lo");
    if ((int64_t)xb * (y ? yb : 1) > (1 << 24)) return api_fail(TRPL_ERR_ARG, "too many bins");
    return TRPL_OK;
}

int trpl_posterior_hist_dev(const double *x, const double *y, const double *W, int64_t S, double xlo, double xhi,
                            int32_t xbins, double ylo, double yhi, int32_t ybins, double *out, void *stream)
{
    if (int rc = check_hist(S, xlo, xhi, xbins, y, ylo, yhi, ybins)) return rc;
    if (S == 0) return TRPL_OK;
    if (!x || !out) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    hipError_t e = trpl::launch_posterior_hist(x, y, W, S, xlo, xhi, xbins, ylo, yhi, ybins, out, (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "posterior histogram launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

int trpl_posterior_hist(const double *x, const double *y, const double *W, int64_t S, double xlo, double xhi,
                        int32_t xbins, double ylo, double yhi, int32_t ybins, double *out, int32_t device,
                        double *seconds)
{
    if (seconds) *seconds = 0.0;
    if (int rc = check_hist(S, xlo, xhi, xbins, y, ylo, yhi, ybins)) return rc;
    if (!out) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    const size_t nb = (size_t)xbins * (y ? ybins : 1);
    memset(out, 0, nb * 8);
    if (S == 0) return TRPL_OK;
    if (!x) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (int rc = select_device(device)) return rc;
    CallScope cs;
    HIP_TRY(cs.open());
    DevBuf dx, dy, dW, dO;
    HIP_TRY(dx.alloc((size_t)S * 8, cs.st)); HIP_TRY(dO.alloc(nb * 8, cs.st));
    HIP_TRY(hipMemcpyAsync(dx.p, x, (size_t)S * 8, hipMemcpyHostToDevice, cs.st));
    if (y) { HIP_TRY(dy.alloc((size_t)S * 8, cs.st)); HIP_TRY(hipMemcpyAsync(dy.p, y, (size_t)S * 8, hipMemcpyHostToDevice, cs.st)); }
    if (W) { HIP_TRY(dW.alloc((size_t)S * 8, cs.st)); HIP_TRY(hipMemcpyAsync(dW.p, W, (size_t)S * 8, hipMemcpyHostToDevice, cs.st)); }
    HIP_TRY(hipMemsetAsync(dO.p, 0, nb * 8, cs.st));
    const double t0 = now_s();
    if (int rc = trpl_posterior_hist_dev(dx.as<double>(), y ? dy.as<double>() : nullptr, W ? dW.as<double>() : nullptr, S, xlo,
                                         xhi, xbins, ylo, yhi, ybins, dO.as<double>(), cs.st))
        return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    if (seconds) *seconds = now_s() - t0;
    HIP_TRY(hipMemcpyAsync(out, dO.p, nb * 8, hipMemcpyDeviceToHost, cs.st));
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

/* ------------------------------------------------------------------ sampler -------------- */
static int check_box(int64_t S, int32_t ncol, const double *lo, const double *hi, const int32_t *do_log)
{
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (ncol < 1 || ncol > 16) return api_fail(TRPL_ERR_ARG, "ncol=%d must be in [1, 16]", ncol);
    if (!lo || !hi || !do_log) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    for (int c = 0; c < ncol; c++) {
        if (!(lo[c] <= hi[c])) return api_fail(TRPL_ERR_ARG, "column %d: lo must be <= hi", c);
        if (do_log[c] && lo[c] != hi[c] && !(lo[c] > 0)) return api_fail(TRPL_ERR_ARG, "column %d: log-uniform needs lo > 0", c);
    }
    return TRPL_OK;
}

int trpl_sample_box_dev(uint32_t seed, int64_t S, int32_t ncol, const double *lo, const double *hi, const int32_t *do_log,
                        uint32_t flags, double *X, void *stream)
{
    if (int rc = check_box(S, ncol, lo, hi, do_log)) return rc;
    if (S == 0) return TRPL_OK;
    if (!X) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    hipError_t e = trpl::launch_sample_box(seed, S, ncol, lo, hi, do_log, flags, X, (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "sampler launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

int trpl_sample_box(uint32_t seed, int64_t S, int32_t ncol, const double *lo, const double *hi, const int32_t *do_log,
                    uint32_t flags, double *X, int32_t device, double *seconds)
{
    if (seconds) *seconds = 0.0;
    if (int rc = check_box(S, ncol, lo, hi, do_log)) return rc;
    if (S == 0) return TRPL_OK;
    if (!X) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (int rc = select_device(device)) return rc;
    CallScope cs;
    HIP_TRY(cs.open());
    DevBuf dX;
    HIP_TRY(dX.alloc((size_t)S * ncol * 8, cs.st));
    const double t0 = now_s();
    if (int rc = trpl_sample_box_dev(seed, S, ncol, lo, hi, do_log, flags, dX.as<double>(), cs.st)) return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    if (seconds) *seconds = now_s() - t0;
    HIP_TRY(hipMemcpyAsync(X, dX.p, (size_t)S * ncol * 8, hipMemcpyDeviceToHost, cs.st));
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

/* ------------------------------------------------------------------ batched PCR --------- */
int trpl_pcr_solve_batched_dev(const void *ld, const void *d, const void *ud, const void *b, void *x, int64_t S,
                               int32_t L, int32_t elem_bytes, uint32_t flags, void *stream)
{
    if (!pow2(L) || L < 4 || L > 512) return api_fail(TRPL_ERR_ARG, "L=%d must be a power of two in [4, 512]", L);
    if (elem_bytes != 4 && elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "elem_bytes must be 4 or 8");
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (S == 0) return TRPL_OK;
    if (!ld || !d || !ud || !b || !x) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    hipError_t e = (flags & TRPL_FLAG_STRICT)
                       ? trpl::launch_pcr_batched_strict(ld, d, ud, b, x, S, L, elem_bytes, (hipStream_t)stream)
                       : trpl::launch_pcr_batched_fast(ld, d, ud, b, x, S, L, elem_bytes, (hipStream_t)stream);
    if (e != hipSuccess) return api_fail(TRPL_ERR_HIP, "pcr_batched launch: %s", hipGetErrorString(e));
    return TRPL_OK;
}

int trpl_pcr_solve_batched(const void *ld, const void *d, const void *ud, const void *b, void *x, int64_t S,
                           int32_t L, int32_t elem_bytes, uint32_t flags, int32_t device, double *seconds)
{
    if (!pow2(L) || L < 4 || L > 512) return api_fail(TRPL_ERR_ARG, "L=%d must be a power of two in [4, 512]", L);
    if (elem_bytes != 4 && elem_bytes != 8) return api_fail(TRPL_ERR_ARG, "elem_bytes must be 4 or 8");
    if (S < 0) return api_fail(TRPL_ERR_ARG, "S must be >= 0");
    if (seconds) *seconds = 0.0;
    if (S == 0) return TRPL_OK;
    if (!ld || !d || !ud || !b || !x) return api_fail(TRPL_ERR_ARG, "NULL pointer argument");
    if (int rc = select_device(device)) return rc;
    CallScope cs;
    HIP_TRY(cs.open());
    const size_t n = (size_t)S * L * elem_bytes;
    DevBuf bl, bd, bu, bb, bx;
    HIP_TRY(bl.alloc(n, cs.st)); HIP_TRY(bd.alloc(n, cs.st)); HIP_TRY(bu.alloc(n, cs.st)); HIP_TRY(bb.alloc(n, cs.st)); HIP_TRY(bx.alloc(n, cs.st));
    HIP_TRY(hipMemcpyAsync(bl.p, ld, n, hipMemcpyHostToDevice, cs.st));
    HIP_TRY(hipMemcpyAsync(bd.p, d, n, hipMemcpyHostToDevice, cs.st));
    HIP_TRY(hipMemcpyAsync(bu.p, ud, n, hipMemcpyHostToDevice, cs.st));
    HIP_TRY(hipMemcpyAsync(bb.p, b, n, hipMemcpyHostToDevice, cs.st));
    const double t0 = now_s();
    if (int rc = trpl_pcr_solve_batched_dev(bl.p, bd.p, bu.p, bb.p, bx.p, S, L, elem_bytes, flags, cs.st)) return rc;
    HIP_TRY(hipStreamSynchronize(cs.st));
    if (seconds) *seconds = now_s() - t0;
    HIP_TRY(hipMemcpyAsync(x, bx.p, n, hipMemcpyDeviceToHost, cs.st));
    HIP_TRY(hipStreamSynchronize(cs.st));        // the copies back have landed (and their errors surface here)
    return TRPL_OK;
}

}  // extern "C"
