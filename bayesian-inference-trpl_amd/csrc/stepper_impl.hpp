// TRPL time-stepper for gfx950: ONE WAVEFRONT OWNS ONE SYSTEM (sample x curve) FOR ALL T STEPS.
//
// What it computes (reference: pvSimPCR.py tEvol :227-306, iterate :93-225, pcreduce :42-81,
// norm2 :14-40; likelihood probs.py:20-47, :64-75, time interpolation bayeslib.py:184-191):
// variable-order BDF in time; per step a Newton/Picard iteration whose two tridiagonal systems are
// solved by parallel cyclic reduction; PL(t) by midpoint quadrature; optionally log10 + squared error
// against observations (on the grid or interpolated), fused, so PL never reaches memory.
//
// Two arithmetic modes share this source (stepper_strict.hip: -ffp-contract=off, STRICT = true;
// stepper_fast.hip: contraction on, STRICT = false) and differ in node layout (LAY):
//   LAY 0  STRICT, blocked layout i = lane + W*j: the reference's operation order and IEEE divides;
//          its power-of-two reduction tree (norm2 :32-38) becomes in-lane adds + an xor butterfly
//          with the same association, so residual norms, every convergence decision and the N/P/E
//          state are bit-identical to the sequentially executed reference.  History in registers.
//   LAY 2  FAST, L >= 128, interleaved layout i = NR*lane + j: select-free lane shifts (crosslane.hpp),
//          LDS-staged PCR exchange (pcr.hpp), DPP reductions, reciprocals by v_rcp_f64 + one refinement step
//          (third order and unbiased in the solver's quotients, Newton in the pointwise ones: crosslane.hpp), N/P
//          history in a 4-slot LDS ring, 3 waves per SIMD.
//   LAY 1  FAST for L < 128 (blocked layout, DPP unit shifts): small grids, not performance critical.
// The 12 material parameters are wave-uniform (SGPRs).  HBM traffic per system is 13 doubles in and
// one double out in likelihood mode: the kernel is bound by fp64 VALU issue and the CU's LDS, not by
// HBM (tools/iter_bench.hip for the cost breakdown).
#pragma once
#include "pcr.hpp"

namespace trpl {

constexpr uint32_t kFlagPlF32 = 0x2;       // TRPL_FLAG_PL_F32
constexpr uint32_t kFlagNormalize = 0x4;   // TRPL_FLAG_NORMALIZE
constexpr uint32_t kFlagSnapRaw = 0x80;    // TRPL_FLAG_SNAP_RAW
constexpr uint32_t kFlagPairAlwaysSeam = 0x20000;   // TRPL_FLAG_PAIR_ALWAYS_SEAM
constexpr double kPlFloorExcess = 1e-4;    // TRPL_PL_FLOOR_EXCESS

// A flagged system's snapshot / checkpoint slots hold a quiet NaN whose payload is its status word (1 + failing step):
// a resume that finds it in the newest level knows the system was flagged, and when, without iterating on NaNs.
__device__ __forceinline__ double nan_status(int status)
{
    return __longlong_as_double(0x7FF8000000000000LL | (long long)(unsigned)status);
}
// status word carried by a checkpoint value (0: a finite value, the system is alive)
__device__ __forceinline__ int status_of_checkpoint(double v, int64_t t0)
{
    if (v == v) return 0;
    const int payload = (int)(__double_as_longlong(v) & 0x7FFFFFFFLL);
    return payload ? payload : 1 + (int)t0;          // a NaN of other provenance: flagged at the resume step
}

// ---- layout dispatch: LAY 0 = blocked/strict, 1 = blocked/fast (L < 128), 2 = interleaved/fast ----
template <int LAY, int NR, int W>
__device__ __forceinline__ constexpr int node_of(int ln, int j) { return LAY == 2 ? NR * ln + j : ln + W * j; }

template <int LAY, int NR, int W>
__device__ __forceinline__ void shift_up1(const double (&x)[NR], double (&y)[NR], int ln)
{
    if constexpr (LAY == 0) fetch_up<double, NR, W, 1>(x, y, ln);
    else if constexpr (LAY == 1) fetch_up1<double, NR, W>(x, y, ln);
    else nbrB_up<double, NR, 1>(x, y, ln);
}
template <int LAY, int NR, int W>
__device__ __forceinline__ void shift_dn1(const double (&x)[NR], double (&y)[NR], int ln)
{
    if constexpr (LAY == 0) fetch_dn<double, NR, W, 1>(x, y, ln);
    else if constexpr (LAY == 1) fetch_dn1<double, NR, W>(x, y, ln);
    else nbrB_dn<double, NR, 1>(x, y, ln);
}
// sum over all nodes, wave-uniform
template <int LAY, int NR, int W>
__device__ __forceinline__ double sum_nodes(double (&v)[NR])
{
    if constexpr (LAY == 2) {
        double s = v[0];
#pragma unroll
        for (int j = 1; j < NR; j++) s += v[j];
        return wave_sum(s);
    } else {
        return uniform_d(tree_sum<double, NR, W>(v));
    }
}
template <int LAY, int NR, int W, int L>
__device__ __forceinline__ void solve_lay(double (&ld)[NR], double (&d)[NR], double (&ud)[NR], double (&B)[NR],
                                          double (&x)[NR], int ln, double *xch)
{
    if constexpr (LAY == 0) pcr_solve<double, NR, W, L>(ld, d, ud, B, x, ln);
    else if constexpr (LAY == 1) pcr_solve_fast<double, NR, W, L>(ld, d, ud, B, x, ln);
    else cr_pcr_solve<double, NR>(ld, d, ud, B, x, ln, xch);
}

// Relative L1 residual of iterate c in the system (lower l, diagonal dg, upper u | b):
// norm2, pvSimPCR.py:14-40.  Returns true when sum|A c - b| / sum|b| < TOL.  STRICT (LAY 0) forms the
// quotient like the reference; the FAST layouts compare sum|r| < TOL * sum|b| (no divide).
template <int LAY, int NR, int W>
__device__ __forceinline__ bool residual_below(const double (&l)[NR], const double (&dg)[NR],
                                               const double (&u)[NR], const double (&b)[NR],
                                               const double (&c)[NR], double TOL, int ln)
{
    double cm[NR], cp[NR], r[NR], ab[NR];
    shift_dn1<LAY, NR, W>(c, cm, ln);
    shift_up1<LAY, NR, W>(c, cp, ln);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        // l = 0 on row 0 and u = 0 on row L-1, so the wrapped neighbour contributes +-0
        if constexpr (LAY == 0) r[j] = fabs(l[j] * cm[j] + dg[j] * c[j] + u[j] * cp[j] - b[j]);      // the reference's order
        else r[j] = fabs(__builtin_fma(l[j], cm[j], __builtin_fma(dg[j], c[j], __builtin_fma(u[j], cp[j], -b[j]))));
        ab[j] = fabs(b[j]);
    }
    if constexpr (LAY == 2) {
        // one reduction instead of two: sum|r| < TOL*sum|b|  <=>  sum(|r| - TOL*|b|) < 0
        double q[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) q[j] = __builtin_fma(-TOL, ab[j], r[j]);
        double s = q[0];
#pragma unroll
        for (int j = 1; j < NR; j++) s += q[j];
        return wave_sum_negative(s);
    } else {
        const double sr = sum_nodes<LAY, NR, W>(r);
        const double sb = sum_nodes<LAY, NR, W>(ab);
        if constexpr (LAY == 0) return sr / sb < TOL;
        else                    return sr < TOL * sb;
    }
}

// STRICT, bundled systems (max_sims_per_block > 1): the quotient itself, because shared_array_max (pvSimPCR.py:83-90)
// treats a NaN differently in the first system of a bundle (it sticks) and in the others (`>` skips it)
template <int NR, int W>
__device__ __forceinline__ double residual_err_strict(const double (&l)[NR], const double (&dg)[NR], const double (&u)[NR],
                                                      const double (&b)[NR], const double (&c)[NR], int ln)
{
    double cm[NR], cp[NR], r[NR], ab[NR];
    shift_dn1<0, NR, W>(c, cm, ln);
    shift_up1<0, NR, W>(c, cp, ln);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        r[j] = fabs(l[j] * cm[j] + dg[j] * c[j] + u[j] * cp[j] - b[j]);      // the reference's order
        ab[j] = fabs(b[j]);
    }
    const double sr = sum_nodes<0, NR, W>(r);
    const double sb = sum_nodes<0, NR, W>(ab);
    return sr / sb;
}


// MIXED arithmetic (TRPL_FLAG_MIXED): the same test as residual_below<2> and, beside it, the residual itself,
// r = b - A c in fp64 -- the right-hand side of the correction equation A delta = r that is then solved in fp32.
template <int NR>
__device__ __forceinline__ bool residual_vec(const double (&l)[NR], const double (&dg)[NR], const double (&u)[NR],
                                             const double (&b)[NR], const double (&c)[NR], double TOL, int ln,
                                             double (&r)[NR])
{
    double cm[NR], cp[NR], q[NR];
    nbrB_dn<double, NR, 1>(c, cm, ln);
    nbrB_up<double, NR, 1>(c, cp, ln);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        r[j] = b[j] - (l[j] * cm[j] + dg[j] * c[j] + u[j] * cp[j]);
        q[j] = __builtin_fma(-TOL, fabs(b[j]), fabs(r[j]));
    }
    double s = q[0];
#pragma unroll
    for (int j = 1; j < NR; j++) s += q[j];
    return wave_sum_negative(s);
}

// One correction of iterate c of the system (lo, dg, up | bb): delta from the fp32 solve of A delta = r,
// c += delta in fp64.  The fp32 solve has a relative error of ~cond(A) * 6e-8 ON THE CORRECTION, which is
// itself O(change per time step) in the first and O(tolerance) in the last inner iteration: the state keeps
// fp64 accuracy, the residual that decides convergence is the fp64 one.
template <int NR>
__device__ __forceinline__ bool correct_mixed(const double (&lo)[NR], const double (&dg)[NR], const double (&up)[NR],
                                              const double (&bb)[NR], double (&c)[NR], double TOL, int ln, float *xch)
{
    double r[NR];
    const bool ok = residual_vec<NR>(lo, dg, up, bb, c, TOL, ln, r);
    float lf[NR], df[NR], uf[NR], rf[NR], xf[NR];
#pragma unroll
    for (int j = 0; j < NR; j++) { lf[j] = (float)lo[j]; df[j] = (float)dg[j]; uf[j] = (float)up[j]; rf[j] = (float)r[j]; }
    cr_pcr_solve<float, NR>(lf, df, uf, rf, xf, ln, xch);
#pragma unroll
    for (int j = 0; j < NR; j++) c[j] += (double)xf[j];
    return ok;
}


// What a system emits: PL(t) to memory (pvSim mode, pvSimPCR.py:281,:393) and/or the running squared
// log-error against the observations (fused likelihood: bayeslib.py:150-157,:184-191, probs.py:29-44).
// Shared by the fp64 and fp32 steppers; every member is wave-uniform.
struct PlSink {
    const StepArgs &a;
    const CurveConst &cc;
    int64_t orow;            // output row (curve-major: c*S + s)
    const double *obs;
    const int32_t *obs_hi;   // off-grid observation times (trpl_loglik_obs) or nullptr
    const double *obs_dx, *obs_h;
    // columns and steps are 32-bit (T <= 2^30 - 16 and plT <= T + 1, enforced at the ABI: no sum below can overflow): the per-step bookkeeping is scalar code, and
    // 64-bit scalar adds / compares are two instructions each; only addresses are formed in 64 bits
    int32_t ncol_ll;         // number of observations of this curve (0 outside likelihood mode)
    int32_t t_last;          // last step that can influence an output
    int32_t next_obs = 0;
    double mag, lg_prev = 0.0, sse = 0.0, pl0_d = 1.0;
    float pl0_f = 1.0f;
    bool want_pl, want_ll, interp;
    // batched emission (FAST): lane k parks column base+k; a batch of up to 64 columns is processed at once
    double pend = 0.0;
    int32_t base = 0;
    // the cancellation floor (include/trpl.h, floor_col): first compared column whose PL = B (sum N P - L n0p0) is
    // below kPlFloorExcess of B L n0p0, i.e. whose mean excess product is a 1e-4 of the equilibrium product: the
    // ~1e-12 by which two correct fp64 evaluations of the state differ is then amplified to >= 1e-8 of PL
    double pl_floor = 0.0;                   // set_floor(); 0: only a non-positive PL counts
    int32_t first_floor = -1;
    __device__ __forceinline__ void set_floor(double rate, double n0p0, int L) { pl_floor = kPlFloorExcess * (rate * ((double)L * n0p0)); }

    int lane_;               // lane within the wavefront (== threadIdx.x except in the multi-wave bundled kernel)

    __device__ PlSink(const StepArgs &a_, const CurveConst &cc_, int c, int64_t s, double mag_, int lane = (int)threadIdx.x)
        : a(a_), cc(cc_), orow((int64_t)c * a_.S + s), mag(mag_), lane_(lane)
    {
        want_pl = a.pl != nullptr;
        want_ll = a.sse != nullptr;
        ncol_ll = want_ll ? (int32_t)cc.n_obs : 0;
        obs = want_ll ? a.obs + (int64_t)c * a.obs_ld : nullptr;
        interp = want_ll && a.obs_hi != nullptr;
        obs_hi = interp ? a.obs_hi + (int64_t)c * a.obs_ld : nullptr;
        obs_dx = interp ? a.obs_dx + (int64_t)c * a.obs_ld : nullptr;
        obs_h = interp ? a.obs_h + (int64_t)c * a.obs_ld : nullptr;
        // all T+1 steps when PL is stored (the reference runs them all), otherwise up to the last
        // observation
        t_last = want_pl ? (int32_t)a.T : (interp ? obs_hi[ncol_ll - 1] : (ncol_ll - 1) * a.plT);
    }

    // a spare wavefront of a short bundle: computes, emits nothing (t_last stays the workgroup's)
    __device__ void mute() { want_pl = false; want_ll = false; interp = false; ncol_ll = 0; }

    // plv = rate * (sum N P - L n0 p0) of the state at time t = col * plT, non-dimensional.  The steppers count
    // PL columns instead of dividing t by plT every step (a 64-bit scalar division is ~130 instructions).
    __device__ __forceinline__ void emit(int32_t col, double plv)
    {
        if (a.floor_col && (interp || col < ncol_ll) && first_floor < 0 && !(plv >= pl_floor)) first_floor = (int32_t)col;
        if (want_pl && lane_ == 0) {                                                   // :281,:393
            if (a.pl_bytes == 4) ((float *)a.pl)[orow * a.pl_ld + col] = (float)plv / (float)cc.plnorm;
            else                 ((double *)a.pl)[orow * a.pl_ld + col] = plv / cc.plnorm;
        }
        if (!(interp || col < ncol_ll)) return;
        double lg;
        if (a.flags & kFlagPlF32) {                // the reference's float32 plI buffer (bayeslib.py:137)
            float f = (float)plv / (float)cc.plnorm;
            if (a.flags & kFlagNormalize) { if (col == 0) pl0_f = f; f = f / pl0_f; }
            if ((double)f < DBL_MIN) f = (float)DBL_MIN;
            lg = (double)(float)log10((double)f);
        } else {
            double v = plv / cc.plnorm;
            if (a.flags & kFlagNormalize) { if (col == 0) pl0_d = v; v = v / pl0_d; }
            if (v < DBL_MIN) v = DBL_MIN;
            lg = log10(v);
        }
        if (!interp) {
            double err = lg + mag;
            err -= obs[col];
            sse += err * err;
        } else {
            // every observation bracketed by grid points (col-1, col): scipy interp1d's
            // slope * (x - x_lo) + y_lo (bayeslib.py:189)
            while (next_obs < ncol_ll && obs_hi[next_obs] == col) {
                const double dy = (a.flags & kFlagPlF32) ? (double)((float)lg - (float)lg_prev) : lg - lg_prev;
                const double y = (dy / obs_h[next_obs]) * obs_dx[next_obs] + lg_prev;
                double err = y + mag;
                err -= obs[next_obs];
                sse += err * err;
                next_obs++;
            }
            lg_prev = lg;
        }
    }

    // FAST-mode emission, batched over time: push() parks PL(t) in lane (col - base); every 64 columns
    // (and at the end) flush_batch() does ONE vectorised pass for up to 64 time points -- the
    // re-dimensionalisation, the float32 staging if requested, log10, the squared error against a
    // coalesced load of 64 observations, a coalesced PL store -- instead of a wave-uniform fp64 log10,
    // an IEEE divide, a scalar load and a single-lane store on every step.  The squared errors of a
    // batch are added by a wave reduction, so the likelihood sum is associated differently from the
    // reference's serial loop (~1e-16 relative; STRICT keeps emit()).  Off-grid observation times take the same
    // route since round 5 (the brackets of a batch's observations are fetched across lanes, see flush_batch).
    __device__ __forceinline__ void push(int32_t col, double plv)
    {
        if (lane_ == col - base) pend = plv;
        if (col - base == 63) flush_batch(64);
    }

    __device__ __forceinline__ void flush_batch(int n)      // columns base .. base+n-1, wave-uniform n in [0, 64]
    {
        if (n > 0) {
            const int lane = lane_;
            const int32_t col = base + lane;
            const bool live = lane < n;
            const bool f32 = (a.flags & kFlagPlF32) != 0;
            double v;                                       // re-dimensionalised PL (pvSimPCR.py:393)
            float vf = 0.0f;
            if (f32) { vf = (float)pend / (float)cc.plnorm; v = (double)vf; }
            else     { v = pend / cc.plnorm; }
            if (want_pl && live) {                                                         // :281
                if (a.pl_bytes == 4) ((float *)a.pl)[orow * a.pl_ld + col] = (float)pend / (float)cc.plnorm;
                else                 ((double *)a.pl)[orow * a.pl_ld + col] = pend / cc.plnorm;
            }
            if (a.floor_col && (interp || base < ncol_ll) && first_floor < 0) {
                const uint64_t below = __builtin_amdgcn_ballot_w64(live && (interp || col < ncol_ll) && !(pend >= pl_floor));
                if (below) first_floor = (int32_t)(base + __builtin_ctzll(below));
            }
            if (interp || base < ncol_ll) {                 // bayeslib.py:150-157, probs.py:29-44
                if (a.flags & kFlagNormalize) {
                    if (base == 0) { pl0_d = uniform_d(v); pl0_f = (float)pl0_d; }
                    if (f32) { vf = vf / pl0_f; v = (double)vf; } else { v = v / pl0_d; }
                }
                double lg;
                if (f32) {
                    if ((double)vf < DBL_MIN) vf = (float)DBL_MIN;
                    lg = (double)(float)log10((double)vf);
                } else {
                    if (v < DBL_MIN) v = DBL_MIN;
                    lg = log10(v);
                }
                if (!interp) {
                    const bool use = live && col < ncol_ll;
                    double err = lg + mag;
                    err -= obs[use ? col : 0];
                    sse += wave_sum(use ? err * err : 0.0);
                } else {
                    // Off-grid observation times (trpl_loglik_obs; bayeslib.py:184-191): observation i is bracketed by the grid
                    // columns (hi_i - 1, hi_i).  The observations whose upper column lies in this batch are taken 64 at a
                    // time, one per lane -- they are sorted by time, so they are the next ones -- each lane fetching the two
                    // log10 PL values of its bracket from the lanes that hold them (ds_bpermute; column base - 1 is carried from
                    // the previous batch) and forming scipy interp1d's slope * (x - x_lo) + y_lo.  Round 5: until then this
                    // path emitted column by column (emit(): a wave-uniform fp64 log10 and an IEEE divide per time step), 21 %
                    // slower on the production shape than the on-grid path.  Sums are wave reductions, as for on-grid batches.
                    while (next_obs < ncol_ll) {
                        const int32_t idx = next_obs + lane;
                        const bool in = idx < ncol_ll;
                        const int32_t hi = in ? obs_hi[idx] : 0x7fffffff;
                        const bool mine = in && hi < base + n;
                        const uint64_t m = __builtin_amdgcn_ballot_w64(mine);
                        if (!m) break;
                        const int k = mine ? hi - base : 0;                     // 0 .. n-1 (earlier brackets were consumed earlier)
                        const double y_hi = __shfl(lg, k, 64);
                        const double y_nb = __shfl(lg, k > 0 ? k - 1 : 0, 64);
                        const double y_lo = k > 0 ? y_nb : lg_prev;
                        const double dy = f32 ? (double)((float)y_hi - (float)y_lo) : y_hi - y_lo;
                        const double h = mine ? obs_h[idx] : 1.0, dx = mine ? obs_dx[idx] : 0.0;
                        double err = ((dy / h) * dx + y_lo) + mag;
                        err -= obs[mine ? idx : 0];
                        sse += wave_sum(mine ? err * err : 0.0);
                        const int cnt = __builtin_popcountll(m);
                        next_obs += cnt;
                        if (cnt < 64) break;                                    // the next observation's bracket ends beyond this batch
                    }
                    lg_prev = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lg), n - 1),
                                               __builtin_amdgcn_readlane(__double2loint(lg), n - 1));
                }
            }
            base += n;
        }
    }

    // status = 0, or 1+t of the step whose iteration hit MAX (pvSimPCR.py:269)
    __device__ __forceinline__ void finish(int status, int64_t itot)
    {
        if (lane_ != 0) return;
        if (status && want_pl) {                   // undefined in the reference; NaN here
            for (int64_t tt = status - 1; tt <= a.T; tt++)
                if (tt % a.plT == 0) {
                    const int64_t col = tt / a.plT;
                    if (a.pl_bytes == 4) ((float *)a.pl)[orow * a.pl_ld + col] = __builtin_nanf("");
                    else                 ((double *)a.pl)[orow * a.pl_ld + col] = __builtin_nan("");
                }
        }
        if (want_ll) a.sse[orow] = status ? __builtin_inf() : sse;
        if (a.status) a.status[orow] = status;
        if (a.iters_total) a.iters_total[orow] = itot;
        if (a.floor_col) a.floor_col[orow] = status ? -2 : first_floor;      // a flagged system: sse = +inf, no column to report
    }
};

// State snapshots (the reference's debug outputs plN / plP / plE: pvSimPCR.py:283-288 (disabled
// there), working form Legacy/pvSim.py:121-126): the state of time step snap_t[i] -- level k, the state
// PL(t) is computed from -- goes to slot snap_slot[i], re-dimensionalised like Legacy/pvSim.py:169-171
// (N, P / dx^3, E / dx; E on the L + 1 edges, E_0 = E_L = 0).  One system's lanes call take(); `row` and
// `live` may differ between the halves of a wavefront that holds two systems.
struct SnapSink {
    const StepArgs &a;
    const CurveConst &cc;
    int next = 0;
    __device__ SnapSink(const StepArgs &a_, const CurveConst &cc_) : a(a_), cc(cc_) {}
    __device__ __forceinline__ bool due(int64_t t) const { return next < a.n_snap && t == (int64_t)a.snap_t[next]; }

    // node(j) = node index of this lane's row j
    template <int NR, int L, typename NodeOf>
    __device__ __forceinline__ void take(const double (&N)[NR], const double (&P)[NR], const double (&E)[NR],
                                         int64_t row, bool live, NodeOf node)
    {
        const int64_t at = row * a.snap_ld + a.snap_slot[next];
        const bool raw = (a.flags & kFlagSnapRaw) != 0;          // solver units: what a resume reads back bit for bit
        if (live) {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int i = node(j);
                if (a.snapN) a.snapN[at * L + i] = raw ? N[j] : N[j] / cc.dx3;
                if (a.snapP) a.snapP[at * L + i] = raw ? P[j] : P[j] / cc.dx3;
                if (a.snapE) {
                    a.snapE[at * (L + 1) + i] = raw ? E[j] : E[j] / cc.dx;
                    if (i == L - 1) a.snapE[at * (L + 1) + L] = 0.0;            // E_L is never written (:205)
                }
            }
        }
        next++;
    }

    // a system flagged at step status-1 (pvSimPCR.py:269): its snapshots from that step on are NaN, like its PL
    template <int L>
    __device__ __forceinline__ void fail_fill(int64_t row, int status, int lane_in_sys, int lanes)
    {
        const double mark = nan_status(status);
        for (int i = 0; i < a.n_snap; i++) {
            if ((int64_t)a.snap_t[i] < (int64_t)status - 1) continue;
            const int64_t at = row * a.snap_ld + a.snap_slot[i];
            for (int n = lane_in_sys; n <= L; n += lanes) {
                if (n < L && a.snapN) a.snapN[at * L + n] = mark;
                if (n < L && a.snapP) a.snapP[at * L + n] = mark;
                if (a.snapE) a.snapE[at * (L + 1) + n] = mark;
            }
        }
    }
};

// the 12 non-dimensional material parameters of one system (wave-uniform) + N0*P0
struct MatPar {
    double N0, P0, DN, DP, rate, sr0, srL, CN, CP, tauN, tauP, Lambda, n0p0;
    double mfirst = 0.0, mlast = 0.0;   // SURF_FMA only: 1.0 on the lane that owns node 0 / node L-1, else 0.0
    // FAST assembly only (set by fast_constants): 2 C, C n0p0 and tau_other n0p0 of each equation
    double Co2N = 0.0, Con0N = 0.0, tVn0N = 0.0, Co2P = 0.0, Con0P = 0.0, tVn0P = 0.0;
    double LDP = 0.0, LDN = 0.0, hLDP = 0.0, hLDN = 0.0;   // Lambda DP, Lambda DN and their halves (field update)
    // SURF_FMA only: D and D/2 of each equation, zeroed on the lane that owns the system's first / last node, so
    // that the boundary rows' stencil entries vanish without a select or an extra fma
    double DNf = 0.0, hDNf = 0.0, DPf = 0.0, hDPf = 0.0, DNl = 0.0, hDNl = 0.0, DPl = 0.0, hDPl = 0.0;
    __device__ __forceinline__ void boundary_constants()
    {
        const double kf = 1.0 - mfirst, kl = 1.0 - mlast;
        DNf = kf * DN; hDNf = 0.5 * DNf; DPf = kf * DP; hDPf = -0.5 * DPf;
        DNl = kl * DN; hDNl = 0.5 * DNl; DPl = kl * DP; hDPl = -0.5 * DPl;
    }
    __device__ __forceinline__ void fast_constants()
    {
        Co2N = 2.0 * CN; Con0N = CN * n0p0; tVn0N = tauP * n0p0;
        Co2P = 2.0 * CP; Con0P = CP * n0p0; tVn0P = tauN * n0p0;
        LDP = Lambda * DP; LDN = Lambda * DN; hLDP = 0.5 * LDP; hLDN = 0.5 * LDN;
    }
};

// Assemble the electron (IS_N) or hole tridiagonal system of one Newton/Picard iteration:
// pvSimPCR.py:148-170 (electrons) / :178-198 (holes).  lo = A2 (sub-diagonal), dg = A1,
// up = A0 (super-diagonal), bb = right-hand side.  Ep[j] = E at node i+1.
template <int LAY, bool IS_N, int NR, int W, int L, bool SURF_FMA = false>
__device__ __forceinline__ void assemble(const MatPar &m, double a0, const double (&Nk)[NR], const double (&Pk)[NR],
                                         const double (&Ek)[NR], const double (&Ep)[NR], const double (&bU)[NR],
                                         double (&lo)[NR], double (&dg)[NR], double (&up)[NR], double (&bb)[NR],
                                         int ln)
{
    const double D = IS_N ? m.DN : m.DP;
    const double Co = IS_N ? m.CN : m.CP;        // Auger coefficient of the equation's own carrier
    const double Cx = IS_N ? m.CP : m.CN;
    const double tauV = IS_N ? m.tauP : m.tauN;
    constexpr bool STRICT = LAY == 0;
    double inv_tp[NR];
    if constexpr (LAY == 2) {
        double tpv[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) tpv[j] = Nk[j] * m.tauP + Pk[j] * m.tauN;
        rcp_rows<NR>(tpv, inv_tp);
    }
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int i = node_of<LAY, NR, W>(ln, j);
        const bool first = i == 0, last = i == L - 1;
        const double U = IS_N ? Nk[j] : Pk[j];   // the unknown of this equation
        const double V = IS_N ? Pk[j] : Nk[j];   // the other carrier
        if constexpr (STRICT) {
            const double sE = IS_N ? Ek[j] : -Ek[j], sEp = IS_N ? Ep[j] : -Ep[j];
            const double u_i = last ? 0.0 : D * (-sEp / 2 - 1);    // A0[i]
            const double l_i = first ? 0.0 : D * (sE / 2 - 1);     // A2[i]
            const double u_m = first ? 0.0 : D * (-sE / 2 - 1);    // A0[i-1]
            const double l_p = last ? 0.0 : D * (sEp / 2 - 1);     // A2[i+1]
            const double tp = Nk[j] * m.tauP + Pk[j] * m.tauN;
            const double np_ = Nk[j] * Pk[j] - m.n0p0;
            const double ds = -m.rate * V - (V * tp - tauV * np_) / (tp * tp)
                            - (Co * Nk[j] * Pk[j] + Cx * (V * V) + Co * np_);
            up[j] = u_i; lo[j] = l_i;
            dg[j] = a0 - u_m - l_p - ds;
            bb[j] = -(m.CN * Nk[j] + m.CP * Pk[j] + m.rate + 1 / tp) * np_ - ds * U - bU[j];
        } else {
            // The same linearisation with the common subexpressions of ds and bb taken once and two
            // cancellations done on paper (21 instead of 29 operations per interior row):
            //   V tp - tauV np  =  tau_other V^2 + tauV n0p0      (the N P tauV terms cancel exactly)
            //   Co N P + Co np  =  Co (2 np + n0p0)
            // s = -ds.  Same values to rounding; STRICT keeps the reference's expression tree.
            // The TRANSPORT part of the diagonal is NOT rewritten: it is minus the two off-diagonal entries of its
            // column AS ROUNDED (A0[i-1] and A2[i+1], pvSimPCR.py:159), like the reference's.  Rounds 2-3 used the closed
            // form hD (E_i - E_{i+1}) + 2 D on a lane's inner rows (one operation less): algebraically equal, but each
            // column sum of the transport operator then misses zero by ~eps D instead of exactly 0, a spurious source
            // of eps D dt/dx^2 per step that does not average out -- measured on the CPU (tools/forensics): it alone
            // moved the state of the 311 nm films (D dt/dx^2 ~ 500) by 1e-10 .. 6e-10 from the reference evaluation over
            // 8000 steps, where every other choice of this arithmetic (solver order, fused multiply-adds, the rewrites
            // above) stays within 1e-13 .. 1e-12.  An inner row finds both entries in its own lane: no extra operation.
            const double hD = IS_N ? 0.5 * D : -0.5 * D;
            const double tauO = IS_N ? m.tauN : m.tauP;
            const double Co2 = IS_N ? m.Co2N : m.Co2P, Con0 = IS_N ? m.Con0N : m.Con0P, tVn0 = IS_N ? m.tVn0N : m.tVn0P;
            // boundary rows: the paired kernel carries per-lane diffusivities that are exact zeros on the lane owning
            // the system's first / last node (hD E -+ D is then 0 - 0): no select, no extra operation
            const double Dfj = SURF_FMA && j == 0 ? (IS_N ? m.DNf : m.DPf) : D;
            const double hDfj = SURF_FMA && j == 0 ? (IS_N ? m.hDNf : m.hDPf) : hD;
            const double Dlj = SURF_FMA && j == NR - 1 ? (IS_N ? m.DNl : m.DPl) : D;
            const double hDlj = SURF_FMA && j == NR - 1 ? (IS_N ? m.hDNl : m.hDPl) : hD;
            auto zero_if_first = [&](double x) { return SURF_FMA ? x : (first ? 0.0 : x); };
            auto zero_if_last = [&](double x) { return SURF_FMA ? x : (last ? 0.0 : x); };
            const double u_i = zero_if_last(__builtin_fma(-hDlj, Ep[j], -Dlj));
            const double l_i = zero_if_first(__builtin_fma(hDfj, Ek[j], -Dfj));
            const double tp = __builtin_fma(Pk[j], m.tauN, Nk[j] * m.tauP);
            const double np_ = __builtin_fma(Nk[j], Pk[j], -m.n0p0);
            const double inv = LAY == 2 ? inv_tp[j] : rcp_nr(tp);
            const double V2 = V * V;
            const double X = __builtin_fma(tauO, V2, tVn0) * (inv * inv);
            const double Y = __builtin_fma(Cx, V2, __builtin_fma(Co2, np_, Con0));
            const double s = __builtin_fma(m.rate, V, X) + Y;
            const double t = __builtin_fma(m.CP, Pk[j], __builtin_fma(m.CN, Nk[j], m.rate)) + inv;
            up[j] = u_i; lo[j] = l_i;
            // rows 0 and NR-1: the column's other entries belong to the neighbouring lanes' rows -- the same expression
            // on the same operands, hence the same bits; inner rows take them from this lane below
            const double u_m = zero_if_first(__builtin_fma(-hDfj, Ek[j], -Dfj));       // A0[i-1]
            const double l_p = zero_if_last(__builtin_fma(hDlj, Ep[j], -Dlj));         // A2[i+1]
            dg[j] = (j >= 1 && j <= NR - 2) ? s : a0 - u_m - l_p + s;
            bb[j] = __builtin_fma(-t, np_, __builtin_fma(s, U, -bU[j]));
        }
    }
    if constexpr (!STRICT && NR >= 3) {
#pragma unroll
        for (int j = 1; j <= NR - 2; j++) dg[j] = a0 - up[j - 1] - lo[j + 1] + dg[j];
    }
    // surfaces (:164-170 / :192-198): node 0 is (lane 0, row 0), node L-1 is (lane W-1, row NR-1)
    if constexpr (STRICT) {
        const double N0_ = Nk[0], P0_ = Pk[0], NL = Nk[NR - 1], PL = Pk[NR - 1];
        const double V0 = IS_N ? P0_ : N0_, VL = IS_N ? PL : NL, U0 = IS_N ? N0_ : P0_, UL = IS_N ? NL : PL;
        const double s0 = N0_ + P0_, sL = NL + PL;
        const double ds0 = -m.sr0 * (V0 * V0 + m.n0p0) / (s0 * s0);
        const double dsL = -m.srL * (VL * VL + m.n0p0) / (sL * sL);
        const double f0 = m.sr0 * (N0_ * P0_ - m.n0p0) / s0 + ds0 * U0;
        const double fL = m.srL * (NL * PL - m.n0p0) / sL + dsL * UL;
        if (ln == 0) { dg[0] -= ds0; bb[0] -= f0; }
        if (ln == W - 1) { dg[NR - 1] -= dsL; bb[NR - 1] -= fL; }
    } else {
        // one evaluation serves both surfaces: the upper half of the wave works on node L-1
        const bool hiHalf = ln >= W / 2;
        const double Ns = hiHalf ? +Nk[NR - 1] : +Nk[0], Ps = hiHalf ? +Pk[NR - 1] : +Pk[0];
        const double sr = hiHalf ? +m.srL : +m.sr0;
        const double Vs = IS_N ? Ps : Ns, Us = IS_N ? Ns : Ps;
        const double inv = LAY == 2 ? rcp_row(Ns + Ps) : rcp_nr(Ns + Ps);
        // ds_s = -sr (V^2 + n0p0) / (N+P)^2,  f_s = sr (N P - n0p0) / (N+P) + ds_s U   (:165-170), with g = sr / (N+P)
        const double g = sr * inv;
        const double dss = -(g * inv) * __builtin_fma(Vs, Vs, m.n0p0);
        const double fs = __builtin_fma(g, __builtin_fma(Ns, Ps, -m.n0p0), dss * Us);
        if constexpr (SURF_FMA) {
            // branch-free: the two lane-conditional updates as four fmas with 0/1 lane masks, so that the
            // iteration body stays one basic block (the paired kernel: every wave holds surface lanes)
            dg[0] = __builtin_fma(-m.mfirst, dss, dg[0]);
            bb[0] = __builtin_fma(-m.mfirst, fs, bb[0]);
            dg[NR - 1] = __builtin_fma(-m.mlast, dss, dg[NR - 1]);
            bb[NR - 1] = __builtin_fma(-m.mlast, fs, bb[NR - 1]);
        } else {
            if (ln == 0) { dg[0] -= dss; bb[0] -= fs; }
            if (ln == W - 1) { dg[NR - 1] -= dss; bb[NR - 1] -= fs; }
        }
    }
}

// Pointwise field update on edges 1..L-1 (pvSimPCR.py:205-209); edge 0 keeps its value (0).
template <int LAY, int NR, int W>
__device__ __forceinline__ void update_field(const MatPar &m, double a0, const double (&Nk)[NR],
                                             const double (&Pk)[NR], const double (&bE)[NR], double (&Ek)[NR],
                                             int ln)
{
    constexpr bool STRICT = LAY == 0;
    double Nm[NR], Pm[NR];
    shift_dn1<LAY, NR, W>(Nk, Nm, ln);
    shift_dn1<LAY, NR, W>(Pk, Pm, ln);
    if constexpr (LAY == 2) {
        double A[NR], b[NR], rA[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) {     // (:206-208) with Lambda folded into the diffusivities
            A[j] = __builtin_fma(m.hLDP, Pk[j] + Pm[j], __builtin_fma(m.hLDN, Nk[j] + Nm[j], a0));
            b[j] = __builtin_fma(m.LDP, Pk[j] - Pm[j], __builtin_fma(-m.LDN, Nk[j] - Nm[j], -bE[j]));
        }
        rcp_rows<NR>(A, rA);
#pragma unroll
        for (int j = 0; j < NR; j++) Ek[j] = node_of<LAY, NR, W>(ln, j) >= 1 ? b[j] * rA[j] : Ek[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int i = node_of<LAY, NR, W>(ln, j);
        double e;
        if constexpr (STRICT) {
            const double A = m.Lambda * (m.DP * (Pk[j] + Pm[j]) + m.DN * (Nk[j] + Nm[j])) / 2 + a0;
            const double b = m.Lambda * (m.DP * (Pk[j] - Pm[j]) - m.DN * (Nk[j] - Nm[j])) - bE[j];
            e = b / A;
        } else {
            const double A = (0.5 * m.Lambda) * (m.DP * (Pk[j] + Pm[j]) + m.DN * (Nk[j] + Nm[j])) + a0;
            const double b = m.Lambda * (m.DP * (Pk[j] - Pm[j]) - m.DN * (Nk[j] - Nm[j])) - bE[j];
            e = b * rcp_nr(A);
        }
        Ek[j] = i >= 1 ? e : Ek[j];
    }
}

// BUNDLE: the reference's max_sims_per_block > 1 -- a.bundle consecutive samples of a curve share ONE
// convergence test per inner iteration (pvSimPCR.py:211-216,:258-266).  One workgroup = one bundle, one wavefront per
// system, the per-system verdicts exchanged through LDS with one barrier per iteration.
// HIST32 (TRPL_FLAG_HIST32, L >= 256): the BDF history in DIFFERENCE FORM with fp32 storage.  Every row of the BDF table
// sums to zero (pvSimPCR.py:241-250), so with d_m = U^m - U^{m+1}
//     bU = a1 U^t + a2 U^{t-1} + .. + a5 U^{t-4} = -a0 U^t + w1 d_{t-1} + w2 d_{t-2} + w3 d_{t-3} + w4 d_{t-4},
//     w4 = a5, w3 = a4 + w4, w2 = a3 + w3, w1 = a2 + w2.
// Only U^t (registers) and U^{t-1} (LDS, fp64) are kept in full; d_{t-1} is formed from them exactly, the three older
// differences are read from a 3-slot fp32 ring (d_m in slot m mod 3, each rounded ONCE, when it is stored).  All three
// fields live in LDS: 20 B per node and field instead of 32 B (N, P) / 16 registers (E).  State, assembly, solves,
// residuals and PL stay fp64.  What it buys and what it costs: DESIGN.md section 8 (round 4).
template <int L, bool STRICT, bool SNAP = false, bool MIXED = false, bool BUNDLE = false, bool HIST32 = false>
__global__ void __launch_bounds__(BUNDLE ? 64 * bundle_cap(L) : 64, BUNDLE ? 1 : ((STRICT || L > 128) ? (L > 256 ? 1 : 2) : 3))
stepper_kernel(const StepArgs a)
{
    constexpr int W = L < 64 ? L : 64;
    constexpr int NR = L / W;
    constexpr int LAY = STRICT ? 0 : (L >= 128 ? 2 : 1);   // node layout / arithmetic flavour
    static_assert(!MIXED || LAY == 2, "the mixed-precision correction exists for the interleaved layout (L >= 128)");
    static_assert(!HIST32 || (LAY == 2 && !BUNDLE && !SNAP && !MIXED), "the fp32-difference history exists for the plain FAST one-system stepper (L >= 128), without snapshots / resume");
    static_assert(!BUNDLE || (!MIXED && (STRICT || L <= 128)), "bundles: STRICT at any L, FAST up to L = 128 (LDS: one history ring per system)");
    const int wv = BUNDLE ? (int)(threadIdx.x >> 6) : 0;                 // which system of the bundle
    const int lane64 = BUNDLE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int ln = lane64 & (W - 1);               // lanes >= W replicate lane (lane mod W)
    const int64_t sys = blockIdx.x;
    const int c = (int)(sys % a.C);
    // a short last bundle: its spare wavefronts recompute the bundle's first system, store nothing and always agree
    const int64_t s_first = BUNDLE ? (sys / a.C) * a.bundle : sys / a.C;
    const bool valid = !BUNDLE || s_first + wv < a.S;
    const int64_t s = valid ? s_first + wv : s_first;
    const CurveConst &cc = a.curve[c];

    // ---- non-dimensional material parameters (pvSimPCR.py:327-331) ----
    const double *xs = a.X + s * a.xld;
    const double N0 = xs[0] * cc.scales[0], P0 = xs[1] * cc.scales[1], DN = xs[2] * cc.scales[2],
                 DP = xs[3] * cc.scales[3], rate = xs[4] * cc.scales[4], sr0 = xs[5] * cc.scales[5],
                 srL = xs[6] * cc.scales[6], CN = xs[7] * cc.scales[7], CP = xs[8] * cc.scales[8],
                 tauN = xs[9] * cc.scales[9], tauP = xs[10] * cc.scales[10],
                 Lambda = xs[11] * cc.scales[11];
    const double n0p0 = N0 * P0;
    MatPar mp_ = {N0, P0, DN, DP, rate, sr0, srL, CN, CP, tauN, tauP, Lambda, n0p0};
    if constexpr (!STRICT) mp_.fast_constants();
    const MatPar mp = mp_;
    const double mag = a.xld > 12 ? xs[12] : 0.0;
    const double TOL = a.TOL;
    const int MAX = a.MAX;

    // ---- state U^t (registers) and the four older BDF levels U^{t-1..t-4} ----
    // STRICT and the small grids (L < 128: one row per lane) keep the older levels in registers.  FAST, L >= 128
    // keeps those of N and P in LDS as a 4-slot ring, slot (t' mod 4) holding U^{t'}: nothing is ever moved, a
    // step reads the four slots and then overwrites the oldest with U^t (8 KB per wave for L = 128); E's stay in
    // registers so that ring + PCR exchange buffer leave room for 3 waves per SIMD.
    constexpr bool HREG = LAY != 2;                 // N / P history in registers
    constexpr int HSLOT = 2 * NR * 64;              // N and P; the E history stays in registers
    constexpr int XCH = LAY != 2 ? 2 : 3 * 64;      // PCR exchange buffer, doubles
    // HIST32: {N, P}^{t-1} fp64 [row][lane] | E^{t-1} fp64 | d{N, P} fp32 [3][row][lane] | dE fp32 [3][row][lane], in doubles:
    // (round 4 also measured the field's history in registers and a rounding-residual feedback into the stored differences:
    // DESIGN_HISTORY.md section 7; neither changed the verdict and both left the tree in round 5)
    constexpr int H32 = NR * 64 * 2 + NR * 64 + 3 * NR * 64 + (3 * NR * 64) / 2;
    constexpr int LDSW = HREG ? 2 : (HIST32 ? H32 + XCH : 4 * HSLOT + XCH);                // per wavefront
    __shared__ __attribute__((aligned(16))) double lds[LDSW * (BUNDLE ? bundle_cap(L) : 1)];
    double *hist = lds + (BUNDLE ? wv * LDSW : 0);
    // ring layout [slot][row][lane]{N, P}: a lane's N and P of one row and level are ONE 16-byte LDS access
    double2 *hist2 = reinterpret_cast<double2 *>(hist);
    double *xch = hist + (HREG ? 0 : (HIST32 ? H32 : 4 * HSLOT));    // PCR exchange buffer (LAY 2)
    double *prevE = hist + NR * 64 * 2;                               // HIST32 only (hist2[row * 64 + lane] = {N, P}^{t-1})
    float2 *d32 = reinterpret_cast<float2 *>(prevE + NR * 64);       // d32[(slot * NR + row) * 64 + lane] = {dN, dP}
    float *dE32 = reinterpret_cast<float *>(d32 + 3 * NR * 64);
    const int hl = lane64;                          // this lane's column of the ring
    double Nk[NR], Pk[NR], Ek[NR];
    double hN[4][NR], hP[4][NR];                    // HREG only: levels t-1 .. t-4
    double hE[4][NR];                               // field history, registers (not HIST32)
#pragma unroll
    for (int j = 0; j < NR; j++) {                 // pvSimPCR.py:356-362
        // (a resume takes its state from the checkpoint; dN is not read -- it may be NULL there)
        // (the select sits on the loaded value, so that product and sum are the same expression -- and the same
        // contraction -- in the SNAP and the plain instantiation: their results must agree bit for bit)
        const double raw = (SNAP && a.resN != nullptr) ? 0.0 : a.dN[(int64_t)c * L + node_of<LAY, NR, W>(ln, j)];
        const double dn = raw * cc.dx3;
        Nk[j] = N0 + dn;
        Pk[j] = P0 + dn;
        Ek[j] = 0.0;
        if constexpr (HIST32) {                    // U^{-1} := U^0 (its weight is zero at t = 0), no older differences
            hist2[j * 64 + hl] = make_double2(Nk[j], Pk[j]);
            prevE[j * 64 + hl] = 0.0;
#pragma unroll
            for (int m = 0; m < 3; m++) {
                d32[(m * NR + j) * 64 + hl] = make_float2(0.0f, 0.0f);
                dE32[(m * NR + j) * 64 + hl] = 0.0f;
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                hE[m][j] = 0.0;
                if constexpr (HREG) { hN[m][j] = 0.0; hP[m][j] = 0.0; }
                else hist2[(m * NR + j) * 64 + hl] = make_double2(0.0, 0.0);
            }
        }
    }

    PlSink sink(a, cc, c, s, mag, lane64);
    sink.set_floor(rate, n0p0, L);
    if constexpr (BUNDLE) { if (!valid) sink.mute(); }
    SnapSink snap(a, cc);
    int status = 0;
    int64_t itot = 0;
    // BUNDLE: the systems' verdicts, double-buffered by the parity of a counter that runs ACROSS time steps: a wave
    // can only write buffer b again after the barrier of the iteration in between, which every wave reaches only
    // after its reads of b (a parity that restarted with each time step had no such edge between the last
    // iteration of one step and the first of the next)
    __shared__ int agree[2][BUNDLE ? bundle_cap(L) : 1];
    unsigned phase = 0;

    int32_t t_begin = 0;
    if constexpr (SNAP) {
        if (a.resN != nullptr) {                   // resume at t0 >= 4 from the five levels U^{t0-4} .. U^{t0} (level m <-> t0-4+m)
            t_begin = (int32_t)a.t0;
            const int64_t r5 = sink.orow * 5;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int i = node_of<LAY, NR, W>(ln, j);
                Nk[j] = a.resN[(r5 + 4) * L + i]; Pk[j] = a.resP[(r5 + 4) * L + i]; Ek[j] = a.resE[(r5 + 4) * (L + 1) + i];
#pragma unroll
                for (int m = 0; m < 4; m++) {      // level t0-1-m
                    const double n_ = a.resN[(r5 + 3 - m) * L + i], p_ = a.resP[(r5 + 3 - m) * L + i];
                    hE[m][j] = a.resE[(r5 + 3 - m) * (L + 1) + i];
                    if constexpr (HREG) { hN[m][j] = n_; hP[m][j] = p_; }
                    else {
                        const int slot = (int)((a.t0 - 1 - m) & 3);      // the ring: slot (t' mod 4) holds U^{t'}
                        hist2[(slot * NR + j) * 64 + hl] = make_double2(n_, p_);
                    }
                }
            }
        }
    }
    int32_t pl_next = 0, pl_col = 0;               // next step with t % plT == 0 and its PL column t / plT (:276)
    if constexpr (SNAP) {
        if (t_begin > 0) { pl_col = (t_begin + a.plT - 1) / a.plT; pl_next = pl_col * a.plT; sink.base = pl_col; }
        if (a.resN != nullptr) {
            // a system that was flagged before the checkpoint (its newest level carries the status word): it keeps that
            // status and takes no step.  In a bundle the verdict must be the workgroup's (one barrier sequence): the
            // reference stops the whole block at the first flagged system (pvSimPCR.py:269-274), so do its partners.
            int st0 = status_of_checkpoint(a.resN[(sink.orow * 5 + 4) * L], a.t0);
            if constexpr (BUNDLE) {
                if (lane64 == 0) agree[0][wv] = valid ? st0 : 0;
                __syncthreads();
                for (int q = 0; q < a.bundle; q++) if (st0 == 0) st0 = agree[0][q];
                __syncthreads();
            }
            if (st0) { status = st0; t_begin = sink.t_last + 1; }
        }
    }
    int h32_slot = 2;                              // HIST32: (t - 1) mod 3 at t = 0
    const int32_t row_cap = bdf_row_cap(a.flags);   // TRPL_FLAG_BDF_ORDER: highest row of the BDF table this run uses
    for (int32_t t = t_begin; t <= sink.t_last; t++) {   // tEvol, pvSimPCR.py:237
        if constexpr (SNAP) {                      // the state at time t, before it is stepped (:283-288)
            if (snap.due(t))
                snap.template take<NR, L>(Nk, Pk, Ek, sink.orow, valid && lane64 < W,
                                          [&](int j) { return node_of<LAY, NR, W>(ln, j); });
        }
        double a0, a1, a2, a3, a4, a5;             // :241-250
        bdf_row<double>(t < row_cap ? t : row_cap, a0, a1, a2, a3, a4, a5);

        // PL of the state at time t (level k), pvSimPCR.py:276-281: rate * (sum_i N_i P_i - L N0 P0).
        double plv = 0.0;
        const bool pl_step = t == pl_next;
        if (pl_step) {
            if constexpr (STRICT) {
                // the reference's serial loop, node by node (:278-280), so that PL is bit-identical
                // too (blocked layout: node i = lane + W*j)
                double Sum = -(double)L * n0p0;
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const double q = Nk[j] * Pk[j];
                    for (int l = 0; l < W; l++)
                        Sum += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), l),
                                                __builtin_amdgcn_readlane(__double2loint(q), l));
                }
                plv = rate * Sum;
            } else {
                // per-node excess N_i P_i - N0 P0 by one fma, then the sum: algebraically the same, but
                // the cancellation happens before the accumulation, so a fully decayed system's PL is
                // limited by the solver's state error instead of the rounding of a 128-term sum
                double q[NR];
#pragma unroll
                for (int j = 0; j < NR; j++) q[j] = __builtin_fma(Nk[j], Pk[j], -n0p0);
                plv = rate * sum_nodes<LAY, NR, W>(q);
            }
        }

        // ---------------- iterate, pvSimPCR.py:93-225 ----------------
        double bN[NR], bP[NR], bE[NR];
        // BDF right-hand sides from U^t and the four older levels (:128-135); then U^t enters the history at once (no
        // copy of it is carried through the iterations)
        if constexpr (HREG) {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                bN[j] = a1 * Nk[j] + a2 * hN[0][j] + a3 * hN[1][j] + a4 * hN[2][j] + a5 * hN[3][j];
                bP[j] = a1 * Pk[j] + a2 * hP[0][j] + a3 * hP[1][j] + a4 * hP[2][j] + a5 * hP[3][j];
                bE[j] = a1 * Ek[j] + a2 * hE[0][j] + a3 * hE[1][j] + a4 * hE[2][j] + a5 * hE[3][j];
#pragma unroll
                for (int m = 3; m >= 1; m--) { hN[m][j] = hN[m - 1][j]; hP[m][j] = hP[m - 1][j]; hE[m][j] = hE[m - 1][j]; }
                hN[0][j] = Nk[j]; hP[0][j] = Pk[j]; hE[0][j] = Ek[j];
            }
        } else if constexpr (HIST32) {
            const double w4 = a5, w3 = a4 + w4, w2 = a3 + w3, w1 = a2 + w2;
            // d_m sits in slot m mod 3: d_{t-2}, d_{t-3}, d_{t-4}; the new d_{t-1} takes d_{t-4}'s slot
            const int q2 = h32_slot == 0 ? 2 : h32_slot - 1;                  // (t-2) mod 3, with h32_slot = (t-1) mod 3 (= (t-4) mod 3)
            const int q3 = q2 == 0 ? 2 : q2 - 1;
            const int o1 = h32_slot * NR, o2 = q2 * NR, o3 = q3 * NR;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const double2 u1 = hist2[j * 64 + hl];
                const double e1 = prevE[j * 64 + hl];
                const float2 f2 = d32[(o2 + j) * 64 + hl], f3 = d32[(o3 + j) * 64 + hl], f4 = d32[(o1 + j) * 64 + hl];
                const float g2 = dE32[(o2 + j) * 64 + hl], g3 = dE32[(o3 + j) * 64 + hl], g4 = dE32[(o1 + j) * 64 + hl];
                const double dn = u1.x - Nk[j], dp = u1.y - Pk[j], de = e1 - Ek[j];       // d_{t-1}, exact to fp64 rounding
                bN[j] = -a0 * Nk[j] + w1 * dn + w2 * (double)f2.x + w3 * (double)f3.x + w4 * (double)f4.x;
                bP[j] = -a0 * Pk[j] + w1 * dp + w2 * (double)f2.y + w3 * (double)f3.y + w4 * (double)f4.y;
                bE[j] = -a0 * Ek[j] + w1 * de + w2 * (double)g2 + w3 * (double)g3 + w4 * (double)g4;
                d32[(o1 + j) * 64 + hl] = make_float2((float)dn, (float)dp);
                hist2[j * 64 + hl] = make_double2(Nk[j], Pk[j]);
                dE32[(o1 + j) * 64 + hl] = (float)de;
                prevE[j * 64 + hl] = Ek[j];
            }
            h32_slot = h32_slot == 2 ? 0 : h32_slot + 1;
        } else {
            const int s1 = (int)((t + 3) & 3) * NR, s2 = (int)((t + 2) & 3) * NR,
                      s3 = (int)((t + 1) & 3) * NR, s4 = (int)(t & 3) * NR;        // slots of t-1 .. t-4
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const double2 h1 = hist2[(s1 + j) * 64 + hl], h2 = hist2[(s2 + j) * 64 + hl],
                              h3 = hist2[(s3 + j) * 64 + hl], h4 = hist2[(s4 + j) * 64 + hl];
                bN[j] = a1 * Nk[j] + a2 * h1.x + a3 * h2.x + a4 * h3.x + a5 * h4.x;
                bP[j] = a1 * Pk[j] + a2 * h1.y + a3 * h2.y + a4 * h3.y + a5 * h4.y;
                bE[j] = a1 * Ek[j] + a2 * hE[0][j] + a3 * hE[1][j] + a4 * hE[2][j] + a5 * hE[3][j];
                hist2[(s4 + j) * 64 + hl] = make_double2(Nk[j], Pk[j]);            // U^t replaces U^{t-4} (same slot, t mod 4)
#pragma unroll
                for (int m = 3; m >= 1; m--) hE[m][j] = hE[m - 1][j];
                hE[0][j] = Ek[j];
            }
        }
        int it = MAX;                              // value if the loop runs to exhaustion (:225)
        for (int iters = 0; iters < MAX; iters++) {
            double lo_[NR], dg[NR], up[NR], bb[NR], Ep[NR];
            shift_up1<LAY, NR, W>(Ek, Ep, ln);
            bool okN, okP;
            // ---- electrons (:148-175) ----
            assemble<LAY, true, NR, W, L>(mp, a0, Nk, Pk, Ek, Ep, bN, lo_, dg, up, bb, ln);
            if constexpr (MIXED) {
                okN = correct_mixed<NR>(lo_, dg, up, bb, Nk, TOL, ln, (float *)xch);
            } else if constexpr (BUNDLE && STRICT) {   // shared_array_max: a NaN sticks in the first system, is skipped in the others
                const double e = residual_err_strict<NR, W>(lo_, dg, up, bb, Nk, ln);
                okN = wv == 0 ? e < TOL : !(e >= TOL);
                solve_lay<LAY, NR, W, L>(lo_, dg, up, bb, Nk, ln, xch);
            } else {
                okN = residual_below<LAY, NR, W>(lo_, dg, up, bb, Nk, TOL, ln);                    // :172
                solve_lay<LAY, NR, W, L>(lo_, dg, up, bb, Nk, ln, xch);                                 // :175
            }
            // ---- holes, with the updated electrons (:178-202) ----
            assemble<LAY, false, NR, W, L>(mp, a0, Nk, Pk, Ek, Ep, bP, lo_, dg, up, bb, ln);
            if constexpr (MIXED) {
                okP = correct_mixed<NR>(lo_, dg, up, bb, Pk, TOL, ln, (float *)xch);
            } else {
                // the holes' norm decides nothing unless the electrons' has passed (:213): skipped otherwise (a
                // wave-uniform branch; on the first iteration of a time step it practically always is)
                if constexpr (BUNDLE && STRICT) {
                    okP = false;
                    if (okN) {
                        const double e = residual_err_strict<NR, W>(lo_, dg, up, bb, Pk, ln);
                        okP = wv == 0 ? e < TOL : !(e >= TOL);
                    }
                } else {
                    okP = okN ? residual_below<LAY, NR, W>(lo_, dg, up, bb, Pk, TOL, ln) : false;   // :200
                }
                solve_lay<LAY, NR, W, L>(lo_, dg, up, bb, Pk, ln, xch);                                 // :202
            }
            // ---- field on edges 1..L-1 (:205-209) ----
            update_field<LAY, NR, W>(mp, a0, Nk, Pk, bE, Ek, ln);
            if constexpr (BUNDLE) {                // max over the bundle of errN and errP below TOL (:211-216)
                if (lane64 == 0) agree[phase & 1][wv] = !valid || (okN && okP);
                __syncthreads();
                bool all = true;
                for (int q = 0; q < a.bundle; q++) all = all && agree[phase & 1][q] != 0;
                phase++;
                if (all) { it = iters + 1; break; }
            } else {
                if (okN && okP) { it = iters + 1; break; }                                         // :213-216
            }
        }
        itot += it;
        if (it >= MAX) { status = 1 + (int)t; break; }                                     // :269-274

        if (pl_step) {
            if (STRICT) sink.emit(pl_col, plv);
            else sink.push(pl_col, plv);
            pl_next += a.plT;
            pl_col++;
        }

    }

    if (!STRICT && valid) {                        // columns parked since the last full batch
        const int64_t done = status ? (int64_t)(status - 1) : sink.t_last + 1;      // steps whose PL was emitted
        sink.flush_batch((int)((done + a.plT - 1) / a.plT - sink.base));
    }
    if constexpr (SNAP) {
        if (status && valid) snap.template fail_fill<L>(sink.orow, status, lane64, 64);
    }
    if (valid) sink.finish(status, itot);
}

template <bool STRICT>
hipError_t launch_stepper(const StepArgs &a, hipStream_t stream)
{
    const int64_t nsys = a.S * a.C;
    if (nsys <= 0) return hipSuccess;
    dim3 grid((unsigned)nsys), block(64);
    const bool snap = a.n_snap > 0 || a.resN != nullptr;   // snapshot / resume code only exists in its own instantiation
    if (a.bundle > 1) {                            // one workgroup per bundle of a.bundle consecutive samples of a curve
        if (a.bundle > bundle_cap(a.L) || (!STRICT && a.L > 128)) return hipErrorInvalidValue;
        grid = dim3((unsigned)(((a.S + a.bundle - 1) / a.bundle) * a.C));
        block = dim3(64 * a.bundle);
        switch (a.L) {
#define TRPL_CASE(LL) \
    case LL: \
        if (snap) hipLaunchKernelGGL((stepper_kernel<LL, STRICT, true, false, true>), grid, block, 0, stream, a); \
        else      hipLaunchKernelGGL((stepper_kernel<LL, STRICT, false, false, true>), grid, block, 0, stream, a); \
        break;
            TRPL_CASE(4) TRPL_CASE(8) TRPL_CASE(16) TRPL_CASE(32) TRPL_CASE(64) TRPL_CASE(128)
#undef TRPL_CASE
#define TRPL_CASE(LL) \
    case LL: \
        if constexpr (STRICT) { \
            if (snap) hipLaunchKernelGGL((stepper_kernel<LL, true, true, false, true>), grid, block, 0, stream, a); \
            else      hipLaunchKernelGGL((stepper_kernel<LL, true, false, false, true>), grid, block, 0, stream, a); \
        } \
        break;
            TRPL_CASE(256) TRPL_CASE(512)
#undef TRPL_CASE
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (a.L) {
#define TRPL_CASE(LL) \
    case LL: \
        if (snap) hipLaunchKernelGGL((stepper_kernel<LL, STRICT, true>), grid, block, 0, stream, a); \
        else      hipLaunchKernelGGL((stepper_kernel<LL, STRICT, false>), grid, block, 0, stream, a); \
        break;
        TRPL_CASE(4) TRPL_CASE(8) TRPL_CASE(16) TRPL_CASE(32) TRPL_CASE(64) TRPL_CASE(128)
        TRPL_CASE(256) TRPL_CASE(512)
#undef TRPL_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace trpl
