// Shared declarations for the gfx950 TRPL kernels and the C-ABI glue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trpl {

constexpr int kMaxCurves = 16;
constexpr int kMaxSnaps = 16;      // state snapshots per solve (trpl_solve_pl_snap)
// Systems sharing one convergence test (TRPL_FLAG_BUNDLE): one wavefront each in ONE workgroup, so at most 16 (1024
// threads).  The reference takes what its 48 KB of shared memory hold at 14 L-vectors per system (pvSimPCR.py:113-125):
// 3 at L = 128, 6 at L = 64, 13 at L = 32.  Grids of up to 64 nodes keep a lane's whole state in a few registers and
// run the full 16; from L = 128 on the register budget of a wavefront (and, FAST, its 9.7 KB history ring) allows 4.
constexpr int kMaxBundle = 16;
__host__ __device__ constexpr int bundle_cap(int L) { return L <= 64 ? 16 : 4; }

// Per-curve constants, computed on the host exactly as pvSim does (pvSimPCR.py:314-331,
// :393) so that the in-kernel products X[s][i] * scales[i] round like numpy's.
struct CurveConst {
    double scales[12];   // dx3,dx3,dt/dx2,dt/dx2,dt/dx3,dt/dx,dt/dx,dt/dx6,dt/dx6,1/dt,1/dt,1/dx
    double dx3;          // excitation scale (pvSimPCR.py:356)
    double plnorm;       // dx^2 * dt        (pvSimPCR.py:393)
    double dx;           // Length / L: re-dimensionalises the field snapshots (Legacy/pvSim.py:171)
    int64_t n_obs;       // likelihood mode: number of PL columns compared (<= T/plT + 1)
};

// Kernel argument block of the time-stepper (passed by value; ~2.4 KB).
struct StepArgs {
    const double *X;        // [S][xld]   physical units; columns 0..11 = matPar, 12 = mag offset
    const double *dN;       // [C][L]     excitation, nm^-3
    const double *obs;      // [C][obs_ld] log10 observations (likelihood mode) or nullptr
    const int32_t *obs_hi;  // [C][obs_ld] upper bracketing grid index of each observation, or nullptr (on-grid)
    const double *obs_dx;   // [C][obs_ld] t_i - t_lo
    const double *obs_h;    // [C][obs_ld] t_hi - t_lo
    void *pl;               // [C*S][pl_ld] PL out (solve mode) or nullptr
    double *sse;            // [C][S] out (likelihood mode) or nullptr
    int32_t *status;        // [C][S] out or nullptr
    int64_t *iters_total;   // [C][S] out or nullptr
    int32_t *floor_col;     // [C][S] out or nullptr: first compared PL column with r = PL / (B L n0p0) < TRPL_PL_FLOOR_EXCESS (or a
                            // non-positive / NaN PL); -1: none (floor-free); -2: a flagged system (status != 0, sse = +inf)
    // state snapshots (solve mode only; pvSimPCR.py:283-288, Legacy/pvSim.py:121-126,:169-171): the state
    // at time step snap_t[i] goes to slot snap_slot[i] of snapN/snapP [C*S][snap_ld][L], snapE [..][L+1]
    double *snapN, *snapP, *snapE;
    // resume (solve mode, the SNAP instantiation): the five newest levels U^{t0-4} .. U^{t0} of every system in
    // SOLVER units as a TRPL_FLAG_SNAP_RAW snapshot of steps t0-4 .. t0 stores them: [C*S][5][L] / [..][5][L+1];
    // the time loop then starts at t0 with the full BDF history
    const double *resN, *resP, *resE;
    int64_t t0;
    int64_t S;
    int64_t T;
    int64_t pl_ld;
    int64_t obs_ld;
    double TOL;             // 10^-tol (pvSimPCR.py:112)
    int32_t xld;            // 12 or 13
    int32_t C;
    int32_t L;
    int32_t plT;
    int32_t MAX;
    int32_t pl_bytes;       // 4 or 8
    uint32_t flags;         // TRPL_FLAG_*
    int32_t bundle;         // the reference's max_sims_per_block, 1 .. bundle_cap(L) (1: every system converges alone)
    int32_t n_snap;         // number of (step, slot) pairs below, steps strictly ascending
    int32_t snap_ld;        // slots per system in the snapshot arrays
    int32_t snap_t[kMaxSnaps];
    int32_t snap_slot[kMaxSnaps];
    // stepper_pair only: which two systems share a wavefront.  pair_n = 0: adjacent samples of one curve.  pair_n = C:
    // the C blocks of the period that covers samples 2p, 2p + 1 run the systems (curve pair_cA[k], sample 2p + pair_oA[k])
    // and (pair_cB[k], 2p + pair_oB[k]) -- every system of the period exactly once (build_pair_table, trpl_api.hip)
    int32_t pair_n;
    uint8_t pair_cA[kMaxCurves], pair_oA[kMaxCurves], pair_cB[kMaxCurves], pair_oB[kMaxCurves];
    CurveConst curve[kMaxCurves];
};

// BDF coefficient table of tEvol (pvSimPCR.py:241-250): time step t takes the row min(t, 4) -- order 1 (Euler) at t = 0,
// ramping to order 5 from t = 4 on.  TRPL_FLAG_BDF_ORDER(k) (bits 14-16 of the flags, k = 1 .. 5; 0 = the reference's ramp)
// caps the order at k: the row is min(t, k - 1).  k = 2 is the scheme of the reference's older solver Legacy/pvSim.py:94-97
// (Euler, then BDF2), which makes that file a whole-curve parity reference.  The cap is wave-uniform
// scalar code outside the iterations: no cost when it is off.
__host__ __device__ constexpr int32_t bdf_row_cap(uint32_t flags) { return ((flags >> 14) & 7u) ? (int32_t)((flags >> 14) & 7u) - 1 : 4; }
template <typename T>
__device__ __forceinline__ void bdf_row(int32_t row, T &a0, T &a1, T &a2, T &a3, T &a4, T &a5)
{
    if (row == 0)      { a0 = (T)1.0; a1 = (T)-1.0; a2 = (T)0.0; a3 = (T)0.0; a4 = (T)0.0; a5 = (T)0.0; }
    else if (row == 1) { a0 = (T)1.5; a1 = (T)-2.0; a2 = (T)0.5; a3 = (T)0.0; a4 = (T)0.0; a5 = (T)0.0; }
    else if (row == 2) { a0 = (T)(11.0 / 6); a1 = (T)-3.0; a2 = (T)1.5; a3 = (T)(-1.0 / 3); a4 = (T)0.0; a5 = (T)0.0; }
    else if (row == 3) { a0 = (T)(25.0 / 12); a1 = (T)-4.0; a2 = (T)3.0; a3 = (T)(-4.0 / 3); a4 = (T)0.25; a5 = (T)0.0; }
    else               { a0 = (T)(137.0 / 60); a1 = (T)-5.0; a2 = (T)5.0; a3 = (T)(-10.0 / 3); a4 = (T)1.25; a5 = (T)-0.2; }
}

// Launchers (one translation unit per arithmetic mode, see stepper_strict.hip / stepper_fast.hip).
// host-side row interpolation of the unfused call sequence (likelihood.hip; -ffp-contract=off)
void interp_rows_any(const void *pl, int elem_bytes, int64_t rows, int64_t ld, const int32_t *hi, const double *dx,
                     const double *h, int64_t n_obs, double *out, int64_t out_ld);
hipError_t launch_stepper_strict(const StepArgs &a, hipStream_t stream);
hipError_t launch_stepper_fast(const StepArgs &a, hipStream_t stream);
hipError_t launch_stepper_f32(const StepArgs &a, hipStream_t stream);   // stepper_f32.hip, L >= 128
hipError_t launch_stepper_mixed(const StepArgs &a, hipStream_t stream);  // stepper_mixed.hip, L >= 128
hipError_t launch_stepper_hist32(const StepArgs &a, hipStream_t stream); // stepper_hist32.hip, L = 256 / 512
// stepper_pair.hip: FAST, L = 128, two systems per wavefront
hipError_t launch_stepper_pair(const StepArgs &a, hipStream_t stream);

// likelihood.hip
hipError_t launch_log10_clamp(void *x, int elem_bytes, int64_t rows, int64_t cols, int64_t ld, double mn,
                              hipStream_t stream);
hipError_t launch_sse_accumulate(double *P, const void *pl, int elem_bytes, int64_t rows, int64_t n_obs,
                                 int64_t ld, const double *values, const double *mag, hipStream_t stream);
hipError_t launch_reduce_curves(double *P, const double *sse, int64_t S, int C, hipStream_t stream);
hipError_t launch_pl_loglik(const void *pl, int elem_bytes, int64_t rows, int64_t ld, const double *obs,
                            const int32_t *obs_hi, const double *obs_dx, const double *obs_h, int64_t n_obs,
                            const double *mag, const int32_t *status, double *P, double *sse_out, uint32_t flags,
                            hipStream_t stream);

// posterior.hip: the consumer of P[S] (weights, weighted moments, weighted histograms)
size_t posterior_workspace_bytes(int D);
hipError_t launch_posterior_weights(const double *LL, int64_t S, double tf, double *W, double *stats, double *ws,
                                    hipStream_t st);
hipError_t launch_posterior_moments(const double *V, const double *W, int64_t S, int D, const double *mean_in, double *sums,
                                    double *central, double *ws, hipStream_t st);
hipError_t launch_posterior_hist(const double *x, const double *y, const double *W, int64_t S, double xlo, double xhi,
                                 int xb, double ylo, double yhi, int yb, double *out, hipStream_t st);

// sampler.hip: bayeslib.random_grid on the device (lo / hi / do_log are host arrays)
hipError_t launch_sample_box(uint32_t seed, int64_t S, int ncol, const double *lo, const double *hi, const int32_t *do_log,
                             uint32_t flags, double *X, hipStream_t st);

// batched tridiagonal solve (pcr_batched_impl.hpp, instantiated in both arithmetic modes)
hipError_t launch_pcr_batched_strict(const void *ld, const void *d, const void *ud, const void *b, void *x,
                                     int64_t S, int L, int elem_bytes, hipStream_t stream);
hipError_t launch_pcr_batched_fast(const void *ld, const void *d, const void *ud, const void *b, void *x,
                                   int64_t S, int L, int elem_bytes, hipStream_t stream);

}  // namespace trpl
