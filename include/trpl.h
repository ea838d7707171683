/*
 * trpl.h -- C ABI of libtrpl_hip.so: the MI355X (gfx950) drop-in for the batched TRPL
 * drift-diffusion solve + log-likelihood hot path of HagesLab/Bayesian-Inference-TRPL.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns an int status
 * (TRPL_OK = 0) and never throws; the message for the last failure on the calling thread is
 * returned by trpl_last_error().  Each declaration cites the reference interface it
 * replaces (file:line in the reference checkout).  The Python-side binding a maintainer of
 * the reference would add is shown in INTEGRATION.md; this repo's own binding is
 * bayesian-inference-trpl_amd/_abi.py.
 *
 * Two families:
 *   host-buffer calls   (trpl_solve_pl, trpl_log10_clamp, trpl_sse_accumulate, trpl_loglik):
 *       borrow caller-owned host memory for the duration of the call, exactly like the
 *       reference's numpy-in / numpy-in-place callables; device memory is allocated, used
 *       and released inside the call (pvSimPCR.py:365-384, probs.py:53-60, :80-83).
 *   device-resident calls (`*_dev`): every pointer is a HIP device pointer on the current
 *       device and `stream` is a hipStream_t (NULL = default stream); nothing is allocated,
 *       copied or synchronised -- the caller owns residency and ordering.
 *
 * Data layout (all row-major, C-contiguous):
 *   matpar  [S][12] fp64, physical units nm / ns / V, column order
 *           N0, P0, DN, DP, rate, sr0, srL, CN, CP, tauN, tauP, Lambda   (pvSimPCR.py:97-108)
 *   X       [S][13] fp64 = matpar columns + mag_offset                  (bayeslib.py:144,:195)
 *   excitation dN [C][L] fp64, nm^-3, node n at x = (n + 1/2) dx        (pvSimPCR.py:355-356)
 *   PL      [rows][ld] fp32 or fp64 (elem_bytes 4 / 8), column t/plT    (pvSimPCR.py:281)
 */
#ifndef TRPL_H
#define TRPL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRPL_ABI_VERSION 5   /* 5 (round 6): trpl_has_experimental(); TRPL_FLAG_MIXED / TRPL_FLAG_HIST32 exist only in a library built with
                                `make EXPERIMENTAL=1` (the default library answers TRPL_ERR_UNSUPPORTED); trpl_kernel_name validates like a
                                launch; roctx ranges around the host-buffer calls when libroctx64.so is loadable; trpl_interp_rows (host).
                                4 (round 5): TRPL_FLAG_BDF_ORDER, TRPL_FLAG_PAIR_ALWAYS_SEAM / _PAIR_ADJACENT / _MULTI_FORCE_PAD (were
                                process-wide environment switches), trpl_multi_create_ex, TRPL_PL_ENVELOPE_K_L512; floor_col = -2 for
                                flagged systems, T <= 2^30 - 16, up to TRPL_MAX_CURVES curves and TRPL_FLAG_HIST32 (round 4, then
                                still under version 3) */

/* status codes */
#define TRPL_OK 0
#define TRPL_ERR_ARG 1          /* invalid argument (message says which) */
#define TRPL_ERR_HIP 2          /* a HIP runtime call failed */
#define TRPL_ERR_NODEVICE 3     /* no usable gfx950 device */
#define TRPL_ERR_UNSUPPORTED 4  /* valid request this build has no kernel for */

/* flags for the solver entry points */
#define TRPL_FLAG_STRICT 0x1      /* bit-reproducible arithmetic: no FMA contraction, IEEE divides, the
                                     reference's operation order (state is bit-identical to the
                                     sequentially executed reference); 6.4x slower.
                                     WITHOUT it (the default, "FAST": FMA contraction, v_rcp_f64 + refinement, cyclic reduction +
                                     PCR, reordered sums) a system follows the reference's ITERATION PATH -- the same number of
                                     inner iterations at every time step -- up to knife-edge decisions of the convergence test:
                                     measured against STRICT, iteration totals are identical on every system over the bench window
                                     (T = 8000: Power_scan x 64 x 3, Twothick x 32 x 6, L = 512 x 16 x 3 -- in the -m gpu suite
                                     against the oracle, which allows one system one iteration) and differ by ONE iteration on 8 of
                                     196 608 systems (in totals of ~160 000 each) over the reference's full window T = 80 000
                                     (profiles/r4_validate_twothick_32768_T80000.txt); PL then agrees to the envelope stated at
                                     trpl_loglik below */
#define TRPL_FLAG_PL_F32 0x2      /* trpl_loglik*: round PL and log10 PL through fp32 exactly where the
                                     reference's float32 plI buffer does (bayeslib.py:137) */
#define TRPL_FLAG_NORMALIZE 0x4   /* trpl_loglik*: self-normalise each PL curve to its t = 0 value
                                     (bayeslib.py:150-154) */
#define TRPL_FLAG_FP32 0x8        /* solver state, BDF history and PCR in fp32 (node sums, PL, log10 and the
                                     squared error stay fp64); L >= 128, not combinable with STRICT; use
                                     tol_exp 3-4 (fp32's residual floor is ~1e-7).  A SCREENING mode: an fp32
                                     state cannot hold the BDF history differences, and over thousands of time
                                     steps the PL error grows to percents and, on the decayed tail, tens of
                                     percent (measured at L = 512, T = 8000: DESIGN.md section 7); use the
                                     default fp64 path at tol_exp 6 for results.  No
                                     reference exists for this mode (the reference is fp64 only).  A launch that
                                     takes more than TRPL_FP32_MAX_STEPS time steps is refused (TRPL_ERR_UNSUPPORTED)
                                     unless TRPL_FLAG_FP32_LONG is set too */
#define TRPL_FP32_MAX_STEPS 256   /* measured at L = 512 (tests/test_gpu_l512.py): PL error
                                     ~1e-3 after 60 steps, percents after 1000, 0.4 after 8000 */
#define TRPL_FLAG_FP32_LONG 0x1000 /* with TRPL_FLAG_FP32: run a window longer than TRPL_FP32_MAX_STEPS anyway -- the
                                     caller has read the paragraph above and wants the screening pass */

/* The next two flags name EXPERIMENTAL steppers -- measured, not faster, never selected by default.  They
 * are compiled only into a library built with `make EXPERIMENTAL=1` (libtrpl_hip_exp.so; trpl_has_experimental() == 1); the
 * default library refuses them with TRPL_ERR_UNSUPPORTED and a message that says so.  The bits and TRPL_KERNEL_MIXED / _HIST32
 * keep their values in both builds. */
#define TRPL_FLAG_MIXED 0x40      /* fp64 state, history, assembly, residuals, PL and likelihood; each inner iteration
                                     solves its tridiagonal CORRECTION equation A delta = b - A c in fp32 (L >= 128, not
                                     combinable with STRICT / FP32).  Same convergence test as fp64 (the fp64 residual of
                                     the reference's norm2), so the accuracy is that of the fp64 solver at the same tol;
                                     an fp32 solve resolves ~1e-5 of a correction, so tol_exp 5-6 converges in the fp64
                                     iteration count (measured: tol_exp 7 too).  Measured on MI355X it is NOT faster than
                                     the fp64 stepper (a plain fp32 VALU instruction only issues faster than an fp64 one
                                     with two or more wavefronts per SIMD, tools/ubench_f32_f64.hip; the L = 512 stepper
                                     holds one); it exists as the measured point of DESIGN.md section 7.  No reference exists for it */
#define TRPL_FLAG_HIST32 0x2000    /* fp64 state, assembly, solves, residuals, PL and likelihood; the BDF history kept in
                                     difference form with the older differences stored in fp32 (every row of the BDF table,
                                     pvSimPCR.py:241-250, sums to zero: only the newest level is needed in full).  One-system
                                     stepper at L = 256 / 512; no snapshots, resume or bundles; not combinable with STRICT /
                                     FP32 / MIXED.  An experiment of round 4: the oracle's iteration
                                     totals and PL within 1.4e-9 over 8000 steps at L = 512, but no occupancy gain (-5 %;
                                     +1.6 % in the best build variant) -- never selected by default.  No reference exists
                                     for it */
#define TRPL_FLAG_SNAP_RAW 0x80   /* trpl_solve_pl_snap / _resume: snapshots in SOLVER units (no division by dx^3 / dx), the
                                     form trpl_solve_pl_resume reads back bit for bit */
#define TRPL_FLAG_BUNDLE(m) ((uint32_t)(((m) - 1) & 0xF) << 8)
                                  /* the reference's max_sims_per_block = m (pvSimPCR.py:211-216,:258-266;
                                     bayes_validate.connect_to_gpu defaults to 3).  The reference takes what its 48 KB of
                                     shared memory hold (pvSimPCR.py:113-125: 3 systems at L = 128, 6 at L = 64, 13 at
                                     L = 32); here m <= 16 for L <= 64 and m <= 4 from L = 128 on (one wavefront per system,
                                     one workgroup per bundle; a larger m is TRPL_ERR_ARG).
                                     The samples p .. p+m-1 (p a multiple of m, counted inside the call's batch)
                                     iterate in lockstep until the LARGEST residual of the bundle is below tolerance;
                                     iteration counts and status are the bundle's.  With TRPL_FLAG_STRICT bit-identical to
                                     the reference run that way (tests/golden/pvsim_bundle.npz), any L; without it the
                                     one-system fp64 stepper, L <= 128, to rounding (1e-9).  Not with _FP32 / _MIXED /
                                     _KERNEL_PAIR, and not in the trpl_loglik_multi* calls (a sharded batch would depend on
                                     where it is cut).  m = 1 (no bits set): every sample converges on its own */
#define TRPL_FLAG_BDF_ORDER(k) ((uint32_t)((k) & 0x7) << 14)
                                  /* cap the order of the BDF ramp (pvSimPCR.py:241-250: order min(t + 1, 5) at step t) at
                                     k = 1 .. 5: step t takes the coefficient row of step min(t, k - 1).  No bits set (k = 0): the
                                     reference's ramp.  k = 2 is the time discretisation of the reference's older solver
                                     Legacy/pvSim.py:94-97 (Euler at t = 0, BDF2 ever after), which with CN = CP = 0 makes that
                                     file a WHOLE-CURVE parity reference for every stepper here (tests/golden/legacy_full.npz);
                                     k = 1 is implicit Euler.  Every arithmetic mode and kernel; k > 5 is TRPL_ERR_ARG.  A
                                     wave-uniform select of the coefficient row outside the iterations: no cost when off */
#define TRPL_FLAG_PAIR_ALWAYS_SEAM 0x20000  /* test / measurement: the two-systems-per-wavefront stepper clears every value that
                                     crosses the seam between its two systems in EVERY iteration (the form rounds 1-3 shipped)
                                     instead of only when a time step is repeated ("optimistic seam", round 4).  Results are
                                     bit-identical either way -- this is the reference side of the differential tests on hostile
                                     inputs; ~1.6 % slower.  Ignored by the other kernels */
#define TRPL_FLAG_PAIR_ADJACENT 0x40000     /* measurement: the two-systems-per-wavefront stepper pairs adjacent samples of one curve
                                     (the round-2 rule) instead of two curves of one sample (trpl_pair_table); a scheduling
                                     matter only, results are bit-identical either way */
#define TRPL_FLAG_MULTI_FORCE_PAD 0x80000   /* trpl_loglik_multi_dev, tests: take the padded all-gather + unpadding pass even when
                                     the shards are equal */
#define TRPL_FLAG_KERNEL_PAIR 0x10    /* run the two-systems-per-wavefront stepper whatever the launch size (L = 128,
                                        fp64, not STRICT -- anything else is TRPL_ERR_ARG) */
#define TRPL_FLAG_KERNEL_SINGLE 0x20  /* run the one-system-per-wavefront stepper whatever the launch size */
/* Without either bit the library picks by launch size (below).  The two FAST kernels agree to rounding
 * (~1e-12 relative on a likelihood: their tridiagonal eliminations and node sums are ordered differently),
 * not bit for bit, so a caller that cuts ONE logical batch into several launches -- sample shards over
 * devices or ranks, blocks of a larger run -- and wants every sample's bits to be independent of the cut
 * pins the variant of the whole batch: flags |= the bit trpl_kernel_variant(total systems, ...) names.
 * trpl_loglik_multi and trpl_loglik_multi_dev do this themselves. */

/* which time-stepper kernel a launch of nsys = S * C systems on L nodes taking `steps` time steps (T, or up
 * to the last observation in likelihood mode) with these flags runs on the current device */
#define TRPL_KERNEL_FAST 0        /* one system per wavefront */
#define TRPL_KERNEL_FAST_PAIR 1   /* two systems per wavefront: L = 128, launches that keep the chip full */
#define TRPL_KERNEL_STRICT 2
#define TRPL_KERNEL_FP32 3
#define TRPL_KERNEL_MIXED 4
#define TRPL_KERNEL_HIST32 5
int trpl_kernel_variant(int64_t nsys, int32_t L, int64_t steps, uint32_t flags);
/* The C++ name (namespace, template arguments; no return type, no parameter list) of the time-stepper kernel such a launch runs
 * -- the name rocprofv3's kernel trace lists it under, after "void " -- written to buf as a NUL-terminated string.  snapshots
 * != 0: a launch with state snapshots or a resume (their own instantiation).  bench.py names its `roofline.rocprof_name` with
 * it instead of guessing the instantiation.  Runs the flag / shape checks of a launch first: a combination a launch would
 * refuse (TRPL_FLAG_KERNEL_PAIR with _STRICT, _HIST32 at L = 128 or with snapshots, _FP32 over more than TRPL_FP32_MAX_STEPS
 * steps without _FP32_LONG, a bundle the grid cannot hold, a flag of the experimental build in the default library ...) returns
 * the launch's error code and message and an empty string -- never the name of an instantiation that does not exist. */
int trpl_kernel_name(int64_t nsys, int32_t L, int64_t steps, uint32_t flags, int32_t snapshots, char *buf, int64_t buflen);

/* Who shares a wavefront in the two-systems-per-wavefront stepper of a fused on-grid likelihood launch (a scheduling
 * matter only: a system's bits do not depend on its partner).  Curves with the same thickness and observation count
 * form a group; inside a group consecutive curves -- neighbouring excitation powers in the reference's files -- are
 * paired for each of the two samples of a period (samples 2p, 2p + 1), and the first curve of a group of odd size pairs
 * with itself across the two samples: two curves of ONE sample need similar iteration counts step by step, adjacent
 * samples of one curve do not.  Fills cA/oA/cB/oB [C]: block k of a period runs the systems (curve
 * cA[k], sample 2p + oA[k]) and (cB[k], 2p + oB[k]); returns the number of entries (C; 0 when the table is not used: one
 * curve), or -TRPL_ERR_* on a bad argument.  Host only, no device needed. */
int trpl_pair_table(const double *lengths_nm, const int64_t *n_obs, int32_t C, int32_t L, int64_t T, double time_ns,
                    int32_t *cA, int32_t *oA, int32_t *cB, int32_t *oB);

int trpl_abi_version(void);
/* 1 when this library contains the experimental steppers (TRPL_FLAG_MIXED, TRPL_FLAG_HIST32: `make EXPERIMENTAL=1`), else 0 */
int trpl_has_experimental(void);
const char *trpl_last_error(void);
/* number of visible HIP devices (0 with none); never fails */
int trpl_device_count(void);

/* ---------------------------------------------------------------------------------------
 * trpl_solve_pl -- replaces pvSimPCR.pvSim(plI, plN, plP, plE, matPar, simPar, iniPar, TPB,
 * BPG, max_sims_per_block, init_mode="points")  (pvSimPCR.py:309-401; kernel tEvol :227-306,
 * iterate :93-225, pcreduce :42-81, norm2 :14-40).
 *
 * Time-steps S independent systems (one curve) for t = 0..T with the reference's
 * variable-order BDF / Newton-Picard / PCR scheme and writes PL(t) = B dx sum_i (N_i P_i -
 * n0 p0) for t % plT == 0 into plI[s][t/plT] in the buffer's dtype, re-dimensionalised
 * (pvSimPCR.py:393).
 *   status[s]      0, or 1+t when iterate() reached max_iter at step t (pvSimPCR.py:269);
 *                  that system stops there and its remaining PL entries are NaN (the
 *                  reference leaves them uninitialised and stops the whole launch).
 *   iters_total[s] (nullable) inner iterations summed over the steps taken.
 *   seconds        (nullable) kernel time, like pvSim's return value (pvSimPCR.py:378-381).
 * L must be a power of two, 4 <= L <= 512.
 * ------------------------------------------------------------------------------------- */
int trpl_solve_pl(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L,
                  int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN,
                  void *plI, int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status,
                  int64_t *iters_total, uint32_t flags, int32_t device, double *seconds);

int trpl_solve_pl_dev(const double *matpar, int64_t S, double length_nm, double time_ns,
                      int32_t L, int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter,
                      const double *dN, void *plI, int32_t pl_elem_bytes, int64_t pl_ld,
                      int32_t *status, int64_t *iters_total, uint32_t flags, void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_solve_pl_snap -- trpl_solve_pl that also fills pvSim's debug outputs plN, plP, plE
 * (pvSimPCR.py:309 arguments 2-4; recording hook :283-288, disabled in that file; the working form is
 * Legacy/pvSim.py:121-126 with the re-dimensionalisation of :169-171; consumer Testing/compare.py:22-31):
 * the carrier densities on the L nodes and the field on the L + 1 edges of the state at the time steps
 * snap_steps[i] -- the state PL(t) is computed from -- in nm^-3 and nm^-1.
 *   snap_steps [n_snap] HOST int64 time-step indices (the reference's pT after bayeslib.py:123),
 *              n_snap <= 16, any order.  Like Legacy's `pT.index(t)`, a step listed twice fills its
 *              first slot only; a step outside [0, T] is never reached.  Slots that are not filled keep
 *              the caller's contents.
 *   plN, plP   [S][n_snap][L] fp64 (each nullable);  plE [S][n_snap][L+1] fp64 (nullable), E_0 = E_L = 0.
 * A system flagged at step t (status = 1 + t) gets NaN in the slots of steps >= t, like its PL.
 * Not available with TRPL_FLAG_FP32 (TRPL_ERR_UNSUPPORTED).
 * ------------------------------------------------------------------------------------- */
int trpl_solve_pl_snap(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L,
                       int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN,
                       void *plI, int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status,
                       int64_t *iters_total, const int64_t *snap_steps, int32_t n_snap, double *plN,
                       double *plP, double *plE, uint32_t flags, int32_t device, double *seconds);
int trpl_solve_pl_snap_dev(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L,
                           int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, const double *dN,
                           void *plI, int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status,
                           int64_t *iters_total, const int64_t *snap_steps /*host*/, int32_t n_snap,
                           double *plN, double *plP, double *plE, uint32_t flags, void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_solve_pl_resume -- pvSim's init_mode = "continue" (pvSimPCR.py:357-358), which is only a stub in the
 * reference (`pass`: dN is undefined and the call raises; the commented block :294-306 shows the intent: keep the
 * last time levels in plN / plP / plE and start the next call from them).  Here: the time loop starts at step
 * t0 >= 4 (dN is not an argument: the state comes from the checkpoint) from the five newest BDF levels U^{t0-4} .. U^{t0} of every system,
 *   resN, resP [S][5][L], resE [S][5][L+1] fp64 in SOLVER units, level m <-> step t0 - 4 + m,
 * exactly what trpl_solve_pl_snap[_dev] stores for snap_steps = {t0-4, .., t0} under TRPL_FLAG_SNAP_RAW.
 * A run of T steps and a run to t0 followed by a resume to T give the same PL columns, snapshots and status
 * BIT FOR BIT, in every arithmetic mode (tested): a long window can be cut into segments, checkpointed and
 * continued.  That includes a system that was flagged BEFORE t0: the snapshot slots of a flagged system hold a
 * quiet NaN whose low 31 payload bits are its status word (1 + failing step), the resume finds it in the newest
 * level, reports that status, takes no step (iters_total 0 for this call) and fills PL columns >= the failing
 * step and later snapshots with NaN, as the uninterrupted run does.  Iteration totals add up once the step at t0 is counted once: the time loop runs t = 0 .. T
 * inclusive (pvSimPCR.py:237, the step taken at t = T is computed and dropped), so the run to t0 has taken the
 * step that the resume takes again (a resume with T = t0 counts exactly that step).  PL columns before t0 / plT are not written (the caller's buffer keeps
 * them); snapshot steps before t0 are ignored; iters_total counts the steps taken by this call.
 * Not available with TRPL_FLAG_FP32.
 * ------------------------------------------------------------------------------------- */
int trpl_solve_pl_resume(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L,
                         int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, int64_t t0,
                         const double *resN, const double *resP, const double *resE, void *plI,
                         int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status, int64_t *iters_total,
                         const int64_t *snap_steps, int32_t n_snap, double *plN, double *plP, double *plE,
                         uint32_t flags, int32_t device, double *seconds);
int trpl_solve_pl_resume_dev(const double *matpar, int64_t S, double length_nm, double time_ns, int32_t L,
                             int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter, int64_t t0,
                             const double *resN, const double *resP, const double *resE, void *plI,
                             int32_t pl_elem_bytes, int64_t pl_ld, int32_t *status, int64_t *iters_total,
                             const int64_t *snap_steps /*host*/, int32_t n_snap, double *plN, double *plP,
                             double *plE, uint32_t flags, void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_log10_clamp -- replaces probs.fastlog(plI, MIN, TPB, BPG)  (probs.py:64-85):
 * x <- log10(max(x, min)) in place, in the buffer's dtype.
 * ------------------------------------------------------------------------------------- */
int trpl_log10_clamp(void *x, int32_t elem_bytes, int64_t rows, int64_t cols, int64_t ld,
                     double min, int32_t device, double *seconds);
int trpl_log10_clamp_dev(void *x, int32_t elem_bytes, int64_t rows, int64_t cols, int64_t ld,
                         double min, void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_sse_accumulate -- replaces probs.prob(P, plI, values, uncertainty, mag_grid, TPB, BPG)
 * (probs.py:20-62):  P[j] -= sum_i (plI[j][i] + mag[j] - values[i])^2, fp64 accumulation in
 * index order.  `uncertainty` is not part of the ABI because the reference never reads it
 * (probs.py:40).
 * ------------------------------------------------------------------------------------- */
int trpl_sse_accumulate(double *P, const void *plI, int32_t elem_bytes, int64_t rows,
                        int64_t n_obs, int64_t ld, const double *values, const double *mag,
                        int32_t device, double *seconds);
int trpl_sse_accumulate_dev(double *P, const void *plI, int32_t elem_bytes, int64_t rows,
                            int64_t n_obs, int64_t ld, const double *values, const double *mag,
                            void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_interp_rows -- the time interpolation of the UNFUSED call sequence, bayeslib.py:184-191 (a Python loop of
 * scipy.interpolate.griddata over the rows of plI): every row of the host matrix pl [rows][ld] (fp32 / fp64, elem_bytes), of which
 * ncol columns are valid, is interpolated linearly onto n_obs observation times given as brackets -- hi[i] in [1, ncol - 1] the
 * column right of time i, dx[i] = t_i - t[hi[i] - 1], h[i] = t[hi[i]] - t[hi[i] - 1] (trpl_loglik_obs's convention) -- into
 * out [rows][out_ld] fp64:  out = ((pl[hi] - pl[hi - 1]) / h) * dx + pl[hi - 1], the difference in pl's own type, the rest in fp64,
 * no fused multiply-add: scipy's interp1d form, bit for bit.  PLAIN HOST CODE (no device, no stream; callable from several threads
 * at once): it exists so that the worker threads of an unfused driver do not serialise on an interpreter lock; the fused entry
 * points interpolate in the kernel.
 * ------------------------------------------------------------------------------------- */
int trpl_interp_rows(const void *pl, int32_t elem_bytes, int64_t rows, int64_t ncol, int64_t ld, const int32_t *hi,
                     const double *dx, const double *h, int64_t n_obs, double *out, int64_t out_ld);

/* ---------------------------------------------------------------------------------------
 * trpl_loglik_from_pl_dev -- log-likelihood of PL rows that are ALREADY in device memory (the output of
 * trpl_solve_pl_dev) against one observation set, in one pass: replaces, for one experiment,
 * bayeslib.simulate's normalise / fastlog / griddata / prob sequence (bayeslib.py:150-157, :173-201) on a
 * resident PL block, so that one solve serves every experiment (the reference's loop order
 * curves -> blocks -> experiments, bayeslib.py:117-171) without the PL matrix crossing PCIe.
 *   plI [rows][ld] fp32/fp64 PL as written by trpl_solve_pl_dev (ncol = T/plT + 1 valid columns)
 *   obs [n_obs] log10 observations; obs_hi/obs_dx/obs_h all NULL: observation i sits on grid column i;
 *   all non-NULL: off-grid times bracketed like trpl_loglik_obs (obs_hi in [1, ncol-1], plT = 1)
 *   mag [rows] log offsets (X[:, 12]);  P [rows] (nullable): P[j] -= sse_j;  sse [rows] (nullable) out
 *   status [rows] (nullable) as written by trpl_solve_pl_dev: a flagged system scores +inf, like trpl_loglik
 *   flags: TRPL_FLAG_PL_F32 (implied by a 4-byte buffer), TRPL_FLAG_NORMALIZE
 * The squared errors are summed by a wave reduction (trpl_log10_clamp + trpl_sse_accumulate stay the pair
 * that is bit-identical to probs.prob's serial sum).  Device pointers only; nothing is allocated.
 * ------------------------------------------------------------------------------------- */
int trpl_loglik_from_pl_dev(const void *plI, int32_t elem_bytes, int64_t rows, int64_t ncol, int64_t ld,
                            const double *obs, const int32_t *obs_hi, const double *obs_dx,
                            const double *obs_h, int64_t n_obs, const double *mag, const int32_t *status,
                            double *P, double *sse, uint32_t flags, void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_loglik -- the fused path: replaces the body of bayeslib.simulate (bayeslib.py:117-201)
 * for one experiment whose observation times are the first n_obs[c] points of the simulation
 * grid: for every sample s and curve c it time-steps the system, and accumulates
 *     sse[c][s] = sum_{i < n_obs[c]} ( log10(max(PL_{s,c}(t_i), DBL_MIN)) + X[s][12] - obs[c][i] )^2
 * without ever materialising PL in memory; then P[s] -= sse[0][s] + ... + sse[C-1][s] in
 * curve order (probs.py:44,:60).  A system that does not converge gets sse = +inf.
 *   lengths [C] host doubles (per-curve thickness, bayeslib.py:109-119)
 *   dN      [C][L], obs [C][obs_ld] log10 observations, n_obs [C] host int64 (<= T/plT + 1)
 *   sse     [C][S] out;  status [C][S] out (nullable);  iters_total [C][S] out (nullable)
 *   floor_col [C][S] out (nullable): the CANCELLATION FLOOR indicator.  PL = B (sum_i N_i P_i - L n0 p0) is a
 *           difference of nearly equal numbers once the excess carriers have decayed.  With
 *               r(t) = PL(t) / (B L n0 p0)        (mean excess product per node over the equilibrium product)
 *           this library's default arithmetic and the reference's order of operations (TRPL_FLAG_STRICT, the
 *           sequentially executed reference bit for bit) agree to
 *               |dPL / PL|  <=  1e-9 + K / r(t),    K = TRPL_PL_ENVELOPE_K_THICK = 5e-13 on the 2000 nm films at L = 128,
 *                                                   K = TRPL_PL_ENVELOPE_K_THIN  = 1e-11 on the 311 nm films at L = 128,
 *                                                   K = TRPL_PL_ENVELOPE_K_L512  = 2e-12 on the 2000 nm film at L = 512
 *           (K grows with the stencil's stiffness D dt / dx^2 -- the three grids have dx = 15.6, 2.43 and 3.9 nm; measured
 *           2e-13 / 3.7e-12 / < 2e-12 with the r-independent part at 5.5e-12 for L = 512; the -m gpu tests assert exactly
 *           these constants: tests/gpu_common.py, tests/test_gpu_l512.py, and tests/test_abi.py that header, binding and
 *           tests agree).
 *           floor_col[c][s] = first compared PL column (observation index; the grid step with off-grid observations)
 *           with r < TRPL_PL_FLOOR_EXCESS = 1e-4 or a non-positive / NaN PL; -1 if there is none; -2 for a system
 *           flagged as non-converged (status != 0: sse = +inf, nothing to compare).
 *           CONTRACT: a system with floor_col = -1 has sse within 1e-8 (relative) of the reference evaluation's; every
 *           system whose sse differs by more than 1e-6 has floor_col >= 0, and from that column on its PL depends on
 *           the evaluation order at the level above -- compare such systems across implementations on the columns
 *           before floor_col, or not at all (their posterior weight is 0).  Measurements: DESIGN.md section 2.
 * Any number of curves up to TRPL_MAX_CURVES (bayeslib.py:117 loops over them all): more than 16 run as consecutive launches of
 * up to 16 curves on the same stream; a system's bits do not depend on that grouping.
 * ------------------------------------------------------------------------------------- */
#define TRPL_PL_FLOOR_EXCESS 1e-4
#define TRPL_PL_ENVELOPE_K_THICK 5e-13
#define TRPL_PL_ENVELOPE_K_THIN 1e-11
#define TRPL_PL_ENVELOPE_K_L512 2e-12
#define TRPL_MAX_CURVES 1024
int trpl_loglik(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns,
                int32_t L, int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter,
                const double *dN, const double *obs, int64_t obs_ld, const int64_t *n_obs,
                double *P, double *sse, int32_t *status, int64_t *iters_total, int32_t *floor_col,
                uint32_t flags, int32_t device, double *seconds);

int trpl_loglik_dev(const double *X, int64_t S, int32_t C, const double *lengths_nm /*host*/,
                    double time_ns, int32_t L, int64_t T, int32_t plT, int32_t tol_exp,
                    int32_t max_iter, const double *dN, const double *obs, int64_t obs_ld,
                    const int64_t *n_obs /*host*/, double *P, double *sse, int32_t *status,
                    int64_t *iters_total, int32_t *floor_col, uint32_t flags, void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_loglik_obs -- trpl_loglik for observation times that do NOT lie on the simulation grid:
 * replaces the per-row time interpolation of bayeslib.simulate (scipy griddata, 1-D linear,
 * bayeslib.py:184-191) followed by probs.prob, fused into the time-stepper.  For observation i of
 * curve c (sorted by time, 0 <= t_i <= time_ns) the host passes the bracketing it would hand to
 * interp1d:  obs_hi[c][i] in [1, T] = index of the upper grid point, obs_dx = t_i - t_lo,
 * obs_h = t_hi - t_lo; the kernel forms ((y_hi - y_lo) / h) * dx + y_lo from log10 PL at the two
 * grid points (difference in fp32 under TRPL_FLAG_PL_F32, like interp1d on the reference's
 * float32 buffer) as soon as step hi has been taken.  plT is 1 on this path.
 * ------------------------------------------------------------------------------------- */
int trpl_loglik_obs(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns,
                    int32_t L, int64_t T, int32_t tol_exp, int32_t max_iter, const double *dN,
                    const double *obs, const int32_t *obs_hi, const double *obs_dx, const double *obs_h,
                    int64_t obs_ld, const int64_t *n_obs, double *P, double *sse, int32_t *status,
                    int64_t *iters_total, int32_t *floor_col, uint32_t flags, int32_t device, double *seconds);

int trpl_loglik_obs_dev(const double *X, int64_t S, int32_t C, const double *lengths_nm /*host*/,
                        double time_ns, int32_t L, int64_t T, int32_t tol_exp, int32_t max_iter,
                        const double *dN, const double *obs, const int32_t *obs_hi, const double *obs_dx,
                        const double *obs_h, int64_t obs_ld, const int64_t *n_obs /*host*/, double *P,
                        double *sse, int32_t *status, int64_t *iters_total, int32_t *floor_col, uint32_t flags,
                        void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_loglik_multi -- trpl_loglik / trpl_loglik_obs over several devices from ONE host thread:
 * replaces the reference's distribution of 1024-sample blocks over GPUs (bayeslib.py:131 and the
 * commented-out threaded driver :235-246; one SLURM array task per GPU, :231).  The samples are cut
 * into n_devices contiguous shards (trpl_shard_bounds); every shard is staged, solved and copied back
 * on its own device and stream, all devices run concurrently, and the call returns when the host
 * arrays P[S], sse[C][S], status[C][S], iters_total[C][S] (the last three nullable) are complete.  There is
 * no device-to-device exchange: the systems are independent and the result lives on the host.
 *   devices   [n_devices] HIP device ordinals (an ordinal may repeat: each entry gets its own stream),
 *             or NULL for 0..n_devices-1;  n_devices <= 0 with devices == NULL means every visible device.
 *   obs_hi / obs_dx / obs_h  all NULL: observations on the simulation grid (as trpl_loglik);
 *             all non-NULL: off-grid observations (as trpl_loglik_obs, plT must be 1).
 * The stepper variant is chosen once, from the size of the WHOLE batch (S * C systems), and pinned for every
 * shard, and a system's result does not depend on which other systems share its launch or its wavefront:
 * the results are bit-identical to a single-device call on the same batch, however it is cut.
 * ------------------------------------------------------------------------------------- */
int trpl_loglik_multi(const double *X, int64_t S, int32_t C, const double *lengths_nm, double time_ns,
                      int32_t L, int64_t T, int32_t plT, int32_t tol_exp, int32_t max_iter,
                      const double *dN, const double *obs, const int32_t *obs_hi, const double *obs_dx,
                      const double *obs_h, int64_t obs_ld, const int64_t *n_obs, double *P, double *sse,
                      int32_t *status, int64_t *iters_total, int32_t *floor_col /*[C][S], nullable*/,
                      uint32_t flags, const int32_t *devices, int32_t n_devices, double *seconds);

/* ---------------------------------------------------------------------------------------
 * trpl_multi_* / trpl_loglik_multi_dev -- the device-resident multi-GPU form: ONE process drives
 * n_devices HIP devices; the samples are cut into contiguous shards (trpl_shard_bounds), every device solves
 * its shard, and ONE collective -- an RCCL ncclAllGather over xGMI of ceil(S / n_devices) fp64 per rank --
 * leaves the complete likelihood vector P[S] in the memory of EVERY device, where the posterior core
 * (trpl_posterior_*_dev) consumes it.  Replaces what the reference's commented-out threaded driver
 * (bayeslib.py:235-246) and its one-SLURM-task-per-GPU distribution (:131,:231) leave to separate .npy files.
 *
 * trpl_multi_create: ncclCommInitAll over `devices` (NULL: 0..n_devices-1; n_devices <= 0: every visible
 *   device; ordinals must be distinct), one non-blocking stream per device.  RCCL (librccl.so.1) is bound at
 *   this call, not at library load.  The handle is reusable and not thread-safe.
 * trpl_loglik_multi_dev: per-device pointer tables (host arrays of n_devices device pointers, entry r valid
 *   on devices[r]):
 *     X[r]      [n_r][13]   that shard's samples, n_r = hi_r - lo_r of trpl_shard_bounds(S, n_devices, r)
 *     dN[r]     [C][L]      replicated;  obs[r] [C][obs_ld] replicated (obs_hi / obs_dx / obs_h: tables of
 *               replicated bracket arrays, or all three NULL for on-grid observations)
 *     P_full[r] [S]         OUT on every device: P[s] = - sum_c sse[c][s], the all-gathered likelihoods
 *     sse[r], status[r], iters_total[r], floor_col[r]  [C][n_r] per-shard outputs (tables nullable, as are entries)
 *   Everything is enqueued on the handle's streams (solve, all-gather, unpadding) and the call returns
 *   without waiting; trpl_multi_synchronize waits for all devices.  The kernel variant is pinned from the
 *   whole batch, as in trpl_loglik_multi.
 * ------------------------------------------------------------------------------------- */
typedef struct trpl_multi trpl_multi_t;
int trpl_multi_create(const int32_t *devices, int32_t n_devices, trpl_multi_t **handle);
/* trpl_multi_create with options.  TRPL_MULTI_ALLOW_DUPLICATE_DEVICES: a device ordinal may be listed more than once (one
 * "rank" and stream each) -- for tests that run several ranks on one GPU against a stand-in collective library named by
 * TRPL_RCCL_LIBRARY (tests/mock_rccl); real RCCL refuses duplicate devices. */
#define TRPL_MULTI_ALLOW_DUPLICATE_DEVICES 0x1
int trpl_multi_create_ex(const int32_t *devices, int32_t n_devices, uint32_t create_flags, trpl_multi_t **handle);
int trpl_multi_destroy(trpl_multi_t *handle);
int trpl_multi_device_count(const trpl_multi_t *handle);
int trpl_multi_synchronize(trpl_multi_t *handle);
int trpl_loglik_multi_dev(trpl_multi_t *handle, const double *const *X, int64_t S, int32_t C,
                          const double *lengths_nm /*host*/, double time_ns, int32_t L, int64_t T, int32_t plT,
                          int32_t tol_exp, int32_t max_iter, const double *const *dN, const double *const *obs,
                          const int32_t *const *obs_hi, const double *const *obs_dx, const double *const *obs_h,
                          int64_t obs_ld, const int64_t *n_obs /*host*/, double *const *P_full,
                          double *const *sse, int32_t *const *status, int64_t *const *iters_total,
                          int32_t *const *floor_col, uint32_t flags);
/* The handle's streams are its own (non-blocking): nothing orders them against the streams on which the caller
 * produced X / dN / obs or will consume P_full.  Either wait on the host (trpl_multi_synchronize on both sides), or
 * add the order on the device:
 *   trpl_multi_wait_stream(h, r, s):    what the handle enqueues on rank r from now on runs after what stream s (a
 *                                       hipStream_t of devices[r]; NULL = its default stream) holds now -- call it for
 *                                       every rank BEFORE trpl_loglik_multi_dev when the inputs were just written on s;
 *   trpl_multi_release_stream(h, r, s): what s is given from now on runs after what the handle has enqueued on rank r
 *                                       -- call it AFTER trpl_loglik_multi_dev before reading P_full[r] on s.
 * (trpl_amd.device.MultiDevice.loglik does both with torch's current stream of each device.) */
int trpl_multi_wait_stream(trpl_multi_t *handle, int32_t rank, void *stream);
int trpl_multi_release_stream(trpl_multi_t *handle, int32_t rank, void *stream);

/* [lo, hi) of shard `shard` of S samples cut into n_shards contiguous ranges; the first S % n_shards
 * shards hold one more.  The same rule shards the samples over ranks in the one-process-per-GPU
 * driver (bench.py, trpl_amd.dist.shard_bounds). */
int trpl_shard_bounds(int64_t S, int32_t n_shards, int32_t shard, int64_t *lo, int64_t *hi);
/* the shard whose range contains sample s (the inverse of trpl_shard_bounds; the unpadding pass after the
 * all-gather of trpl_loglik_multi_dev indexes with it); -1 for arguments out of range */
int64_t trpl_shard_of(int64_t S, int32_t n_shards, int64_t s);

/* ---------------------------------------------------------------------------------------
 * trpl_sample_box -- replaces bayeslib.random_grid(minX, maxX, do_log, num_points) after
 * numpy.random.seed(seed) (bayeslib.py:18-32, parallel_bayes_gpu.py:35) and the make_grid overrides
 * (bayeslib.py:67-75), generating X[S][ncol] in device memory: the same MT19937 stream, the same draw
 * order (column after column, fixed columns draw nothing), the same 53-bit doubles.  Linear columns are
 * bit-identical to the reference's; log-uniform columns go through the device's pow() (<= 1 ulp from
 * the host's).  lo / hi / do_log are HOST arrays [ncol] (bounds already unit-converted, as the reference
 * passes them).  flags: 1 = equal mobilities (X[:,2] = X[:,3]), 2 = equal surface velocities
 * (X[:,6] = X[:,5]), 4 = equal Auger coefficients (X[:,8] = X[:,7]).
 * ------------------------------------------------------------------------------------- */
#define TRPL_BOX_EQUAL_MU 0x1
#define TRPL_BOX_EQUAL_S 0x2
#define TRPL_BOX_EQUAL_AUGER 0x4
int trpl_sample_box(uint32_t seed, int64_t S, int32_t ncol, const double *lo, const double *hi,
                    const int32_t *do_log, uint32_t flags, double *X, int32_t device, double *seconds);
int trpl_sample_box_dev(uint32_t seed, int64_t S, int32_t ncol, const double *lo /*host*/,
                        const double *hi /*host*/, const int32_t *do_log /*host*/, uint32_t flags,
                        double *X /*device*/, void *stream);

/* ---------------------------------------------------------------------------------------
 * Posterior core -- the consumer of the likelihood vector: replaces the numpy reductions
 * of Visualization/utils.py on *_BAYRAN_{P,X}.npy.  Streaming, HBM-bound (8 B of likelihood + 8 B per
 * parameter column per sample); results are small arrays a multi-GPU caller can all-reduce.
 *
 * trpl_posterior_weights: normalize(LL / tf)  (utils.py:157-166, marginalization_visual.py:589-591):
 *     W[i] = exp(LL[i]/tf - nanmax(LL/tf) + 1000 ln 2 - ln S) / nansum(same);  NaN stays NaN, -inf gives 0.
 * trpl_posterior_moments: V is [D][S] (one contiguous column per parameter, D <= 16), W the weights;
 *     sums[2+D]      = { sum w, sum w^2, sum w v_d }
 *     central[D][D+2] = { sum w (v_d - m_d)(v_e - m_e) for e < D,  sum w (v_d - m_d)^3,  sum w (v_d - m_d)^4 }
 *     with m = sum w v / sum w: w_mean :197-199, w_variance :202-204, covariance :222-227, w_skew :207-210,
 *     w_kurtosis :212-215 and the weighted sample deviation :168-170 are quotients of these.
 * trpl_posterior_hist: weighted counts (W == NULL: plain counts) of x (and y, for 2-D) in `bins` equal
 *     bins whose edges are lo + (hi - lo) * k / bins exactly as marginalize_1D :243-244 / marginalize_2D
 *     :270-277 build them, with numpy.histogram's rules (left-closed, last bin closed, outside and NaN
 *     dropped); out[xbins] or out[xbins][ybins]; density normalisation is the caller's one-liner.
 * For callers that hold a shard of the samples (one process per GPU): `stats` = { nanmax(LL/tf), nansum of
 * the unnormalised weights } lets the shards' weights be renormalised to the global sum; `mean_in` [D]
 * centres the second call about the all-reduced means; histograms and sums add across shards.
 * The _dev forms take device pointers (out must be zeroed by the caller for hist) and a workspace of
 * trpl_posterior_workspace_bytes(D) bytes (D = 1 for the weights); nothing is allocated.
 * ------------------------------------------------------------------------------------- */
int64_t trpl_posterior_workspace_bytes(int32_t D);
int trpl_posterior_weights(const double *LL, int64_t S, double tf, double *W, double *stats /*nullable [2]*/,
                           int32_t device, double *seconds);
int trpl_posterior_weights_dev(const double *LL, int64_t S, double tf, double *W, double *stats, void *workspace,
                               int64_t workspace_bytes, void *stream);
int trpl_posterior_moments(const double *V, int64_t S, int32_t D, const double *W, const double *mean_in /*nullable*/,
                           double *sums, double *central, int32_t device, double *seconds);
int trpl_posterior_moments_dev(const double *V, int64_t S, int32_t D, const double *W, const double *mean_in,
                               double *sums, double *central, void *workspace, int64_t workspace_bytes,
                               void *stream);
int trpl_posterior_hist(const double *x, const double *y /*nullable: 1-D*/, const double *W /*nullable*/,
                        int64_t S, double xlo, double xhi, int32_t xbins, double ylo, double yhi,
                        int32_t ybins, double *out, int32_t device, double *seconds);
int trpl_posterior_hist_dev(const double *x, const double *y, const double *W, int64_t S, double xlo,
                            double xhi, int32_t xbins, double ylo, double yhi, int32_t ybins, double *out,
                            void *stream);

/* ---------------------------------------------------------------------------------------
 * trpl_pcr_solve_batched_dev -- the stand-alone batched tridiagonal solve (unit U1 of the
 * measurement plan): S independent systems  ld[i] x[i-1] + d[i] x[i] + ud[i] x[i+1] = b[i],
 * i < L, the problem pcreduce solves (pvSimPCR.py:42-81), operands and result in HBM, arrays
 * [S][L], elem_bytes 8 (fp64) or 4 (fp32).  Inputs are not modified.  With TRPL_FLAG_STRICT:
 * parallel cyclic reduction in pcreduce's elimination order with IEEE divides, bit-identical to
 * it.  Default (FAST): in-lane cyclic-reduction levels, then PCR on one row per lane with
 * Newton-refined reciprocals, then back-substitution -- the same solution to rounding
 * (both are exact eliminations), not the same operation order.
 * Algorithmic traffic 5 * L * elem_bytes per system.  Placement: five arrays that sit at the same offset
 * modulo a large power of two (separate 64 MiB allocations) send a wavefront's four loads to the same HBM
 * channel; offsetting each array by a further 4 KiB is worth ~6 % (5.26 -> 5.57 TB/s on MI355X).
 * ------------------------------------------------------------------------------------- */
int trpl_pcr_solve_batched_dev(const void *ld, const void *d, const void *ud, const void *b,
                               void *x, int64_t S, int32_t L, int32_t elem_bytes, uint32_t flags,
                               void *stream);
int trpl_pcr_solve_batched(const void *ld, const void *d, const void *ud, const void *b, void *x,
                           int64_t S, int32_t L, int32_t elem_bytes, uint32_t flags,
                           int32_t device, double *seconds);

#ifdef __cplusplus
}
#endif
#endif /* TRPL_H */
