/*
 * trpl_oracle.c -- CPU restatement of the reference's TRPL hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the checker /
 * reported CPU baseline.  The product path (bayesian-inference-trpl_amd/) never links,
 * imports or falls back to it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * golden vectors produced by the reference's own code (pvSimPCR.py, probs.py, bayeslib.py
 * executed sequentially through oracle/refshim by oracle/gen_golden.py in the development
 * container; fixtures committed under tests/golden/).
 *
 * Every routine cites the reference lines it follows (paths relative to the reference
 * checkout).  Arithmetic is written operation-for-operation in the reference's evaluation
 * order (Python left-to-right, unary minus before '/'), and this file must be compiled
 * with -ffp-contract=off so no multiply-add is fused; under those two conditions the
 * results are bit-identical to the sequentially executed reference (see the test).
 *
 * All quantities are fp64 (`floatY = float64`, pvSimPCR.py:11).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_VERSION 1

int oracle_version(void) { return ORACLE_VERSION; }

/* ------------------------------------------------------------------------------------
 * pcreduce -- pvSimPCR.py:42-81.  In-place parallel cyclic reduction of
 *     ld[i]*x[i-1] + d[i]*x[i] + ud[i]*x[i+1] = B[i],   N a power of two.
 * `buffer` holds the per-level snapshot (4N doubles), exactly as the reference's shared
 * buffer does (:49-54), so reads of neighbours see pre-level values.
 * ---------------------------------------------------------------------------------- */
void oracle_pcreduce(double *ld, double *d, double *ud, double *B, double *c,
                     double *buffer, int N)
{
    int rf = 1;
    while (N > 2 * rf) {                                   /* :48 */
        for (int i = 0; i < N; i++) {                      /* :49-54 snapshot */
            buffer[i] = ld[i];
            buffer[i + N] = d[i];
            buffer[i + 2 * N] = ud[i];
            buffer[i + 3 * N] = B[i];
        }
        for (int i = 0; i < N; i++) {                      /* :57-69 */
            if (i >= rf) {
                double k1 = buffer[i] / buffer[i + N - rf];
                d[i] -= buffer[i + 2 * N - rf] * k1;
                ld[i] = -buffer[i - rf] * k1;
                B[i] -= buffer[i + 3 * N - rf] * k1;
            }
            if (i < N - rf) {
                double k2 = buffer[i + 2 * N] / buffer[i + N + rf];
                d[i] -= buffer[i + rf] * k2;
                ud[i] = -buffer[i + 2 * N + rf] * k2;
                B[i] -= buffer[i + 3 * N + rf] * k2;
            }
        }
        rf *= 2;
    }
    for (int i = 0; i < rf; i++) {                         /* :75-79 2x2 solves */
        double k = ud[i] / d[i + rf];
        c[i] = (B[i] - B[i + rf] * k) / (d[i] - ld[i + rf] * k);
        c[i + rf] = (B[i + rf] - ld[i + rf] * c[i]) / d[i + rf];
    }
}

/* ------------------------------------------------------------------------------------
 * norm2 -- pvSimPCR.py:14-40.  Relative L1 residual  sum|A c - b| / sum|b|  of the
 * iterate c in the assembled system (A0 = upper, A1 = diagonal, A2 = lower), summed with
 * the reference's power-of-two tree (:32-38).  buffer: 2N doubles.
 * ---------------------------------------------------------------------------------- */
double oracle_norm2(const double *A0, const double *A1, const double *A2, const double *b,
                    const double *c, double *buffer, int N)
{
    buffer[0] = fabs(A1[0] * c[0] + A0[0] * c[1] - b[0]);                       /* :20 */
    buffer[N] = fabs(b[0]);
    buffer[N - 1] = fabs(A2[N - 1] * c[N - 2] + A1[N - 1] * c[N - 1] - b[N - 1]); /* :22 */
    buffer[2 * N - 1] = fabs(b[N - 1]);
    for (int i = 1; i < N - 1; i++) {                                           /* :26-29 */
        buffer[i] = fabs(A2[i] * c[i - 1] + A1[i] * c[i] + A0[i] * c[i + 1] - b[i]);
        buffer[i + N] = fabs(b[i]);
    }
    for (int rf = N >> 1; rf >= 1; rf >>= 1)                                    /* :32-38 */
        for (int i = 0; i < rf; i++) {
            buffer[i] = buffer[i] + buffer[i + rf];
            buffer[i + N] = buffer[i + N] + buffer[i + N + rf];
        }
    return buffer[0] / buffer[N];                                               /* :40 */
}

/* Per-system workspace: the reference's shared arrays (pvSimPCR.py:113-125) plus the
 * 6-slot history ring of tEvol (:339-341, :251-256). */
typedef struct {
    int L;
    double *N, *P, *E;          /* [6][L], [6][L], [6][L+1] */
    double *Nk, *Pk, *Ek, *bN, *bP, *bE, *bb, *A0, *A1, *A2, *buffer;
} sysws;

static int ws_alloc(sysws *w, int L)
{
    w->L = L;
    size_t tot = (size_t)6 * L * 2 + (size_t)6 * (L + 1) + (size_t)10 * L + (size_t)4 * L;
    double *m = (double *)calloc(tot, sizeof(double));
    if (!m) return -1;
    w->N = m; m += 6 * L;
    w->P = m; m += 6 * L;
    w->E = m; m += 6 * (L + 1);
    w->Nk = m; m += L; w->Pk = m; m += L; w->Ek = m; m += L;
    w->bN = m; m += L; w->bP = m; m += L; w->bE = m; m += L;
    w->bb = m; m += L; w->A0 = m; m += L; w->A1 = m; m += L; w->A2 = m; m += L;
    w->buffer = m;
    return 0;
}
static void ws_free(sysws *w) { free(w->N); }

/* ------------------------------------------------------------------------------------
 * iterate -- pvSimPCR.py:93-225, MSPB = 1.  One implicit BDF step of one system.
 * mp = the 12 non-dimensional material parameters; a[0..5] BDF coefficients; ring
 * indices k (current), kp (new), ko[0..3] (older).  Returns the iteration count
 * (`iters+1`, :225).
 * ---------------------------------------------------------------------------------- */
/* iterate() in three pieces, so that max_sims_per_block > 1 (several systems sharing ONE convergence test,
 * pvSimPCR.py:213-216) can sweep its systems in lockstep: it_begin = :128-139, it_sweep = one pass of the
 * loop body :148-209 returning errN / errP, it_end = :218-222. */
static void it_begin(sysws *w, const double *a, int k, const int *ko)
{
    const int L = w->L;
    const double a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
    double *Nk = w->Nk, *Pk = w->Pk, *Ek = w->Ek, *bN = w->bN, *bP = w->bP, *bE = w->bE, *A0 = w->A0, *A2 = w->A2;
    const double *Nh = w->N, *Ph = w->P, *Eh = w->E;
    const int LE = L + 1;
    for (int n = 0; n < L; n++) {                                               /* :128-135 */
        Nk[n] = Nh[k * L + n];
        Pk[n] = Ph[k * L + n];
        Ek[n] = Eh[k * LE + n];
        bN[n] = a1 * Nk[n] + a2 * Nh[ko[0] * L + n] + a3 * Nh[ko[1] * L + n]
              + a4 * Nh[ko[2] * L + n] + a5 * Nh[ko[3] * L + n];
        bP[n] = a1 * Pk[n] + a2 * Ph[ko[0] * L + n] + a3 * Ph[ko[1] * L + n]
              + a4 * Ph[ko[2] * L + n] + a5 * Ph[ko[3] * L + n];
        bE[n] = a1 * Ek[n] + a2 * Eh[ko[0] * LE + n] + a3 * Eh[ko[1] * LE + n]
              + a4 * Eh[ko[2] * LE + n] + a5 * Eh[ko[3] * LE + n];
    }
    A0[L - 1] = 0;                                                              /* :138-139 */
    A2[0] = 0;
}

static void it_sweep(sysws *w, const double *mp, double a0, double *errN_out, double *errP_out)
{
    const int L = w->L;
    const double N0 = mp[0], P0 = mp[1], DN = mp[2], DP = mp[3], rate = mp[4],
                 sr0 = mp[5], srL = mp[6], CN = mp[7], CP = mp[8], tauN = mp[9],
                 tauP = mp[10], Lambda = mp[11];
    double *Nk = w->Nk, *Pk = w->Pk, *Ek = w->Ek, *bN = w->bN, *bP = w->bP, *bE = w->bE,
           *bb = w->bb, *A0 = w->A0, *A1 = w->A1, *A2 = w->A2, *buffer = w->buffer;
    double errN = 0, errP = 0;
    const double n0p0 = N0 * P0;
    {
        /* ---- electrons ---- */
        for (int n = 1; n < L; n++) {                                           /* :148-151 */
            A0[n - 1] = DN * (-Ek[n] / 2 - 1);
            A2[n] = DN * (+Ek[n] / 2 - 1);
        }
        for (int n = 0; n < L; n++) {                                           /* :154-161 */
            double tp = Nk[n] * tauP + Pk[n] * tauN;
            double np_ = Nk[n] * Pk[n] - n0p0;
            double ds = -rate * Pk[n] - (Pk[n] * tp - tauP * np_) / (tp * tp)
                      - (CN * Nk[n] * Pk[n] + CP * (Pk[n] * Pk[n]) + CN * np_);
            A1[n] = a0 - A0[(n + L - 1) % L] - A2[(n + 1) % L] - ds;
            bb[n] = -(CN * Nk[n] + CP * Pk[n] + rate + 1 / tp) * np_ - ds * Nk[n] - bN[n];
        }
        {                                                                       /* :164-170 */
            double s0 = Nk[0] + Pk[0], sL = Nk[L - 1] + Pk[L - 1];
            double ds0 = -sr0 * (Pk[0] * Pk[0] + n0p0) / (s0 * s0);
            double dsL = -srL * (Pk[L - 1] * Pk[L - 1] + n0p0) / (sL * sL);
            A1[0] -= ds0;
            A1[L - 1] -= dsL;
            bb[0] -= sr0 * (Nk[0] * Pk[0] - n0p0) / s0 + ds0 * Nk[0];
            bb[L - 1] -= srL * (Nk[L - 1] * Pk[L - 1] - n0p0) / sL + dsL * Nk[L - 1];
        }
        errN = oracle_norm2(A0, A1, A2, bb, Nk, buffer, L);                     /* :172 */
        oracle_pcreduce(A2, A1, A0, bb, Nk, buffer, L);                         /* :175 */

        /* ---- holes, with the UPDATED electrons ---- */
        for (int n = 1; n < L; n++) {                                           /* :178-181 */
            A0[n - 1] = DP * (+Ek[n] / 2 - 1);
            A2[n] = DP * (-Ek[n] / 2 - 1);
        }
        for (int n = 0; n < L; n++) {                                           /* :183-190 */
            double np_ = Nk[n] * Pk[n] - n0p0;
            double tp = Nk[n] * tauP + Pk[n] * tauN;
            double ds = -rate * Nk[n] - (Nk[n] * tp - tauN * np_) / (tp * tp)
                      - (CP * Nk[n] * Pk[n] + CN * (Nk[n] * Nk[n]) + CP * np_);
            A1[n] = a0 - A0[(n + L - 1) % L] - A2[(n + 1) % L] - ds;
            bb[n] = -(CN * Nk[n] + CP * Pk[n] + rate + 1 / tp) * np_ - ds * Pk[n] - bP[n];
        }
        {                                                                       /* :192-198 */
            double s0 = Nk[0] + Pk[0], sL = Nk[L - 1] + Pk[L - 1];
            double ds0 = -sr0 * (Nk[0] * Nk[0] + n0p0) / (s0 * s0);
            double dsL = -srL * (Nk[L - 1] * Nk[L - 1] + n0p0) / (sL * sL);
            A1[0] -= ds0;
            A1[L - 1] -= dsL;
            bb[0] -= sr0 * (Nk[0] * Pk[0] - n0p0) / s0 + ds0 * Pk[0];
            bb[L - 1] -= srL * (Nk[L - 1] * Pk[L - 1] - n0p0) / sL + dsL * Pk[L - 1];
        }
        errP = oracle_norm2(A0, A1, A2, bb, Pk, buffer, L);                     /* :200 */
        oracle_pcreduce(A2, A1, A0, bb, Pk, buffer, L);                         /* :202 */

        /* ---- field, edges 1..L-1 ---- */
        for (int n = 1; n < L; n++) {                                           /* :205-209 */
            A1[n] = Lambda * (DP * (Pk[n] + Pk[n - 1]) + DN * (Nk[n] + Nk[n - 1])) / 2 + a0;
            bb[n] = Lambda * (DP * (Pk[n] - Pk[n - 1]) - DN * (Nk[n] - Nk[n - 1])) - bE[n];
            Ek[n] = bb[n] / A1[n];
        }
    }
    *errN_out = errN;
    *errP_out = errP;
}

static void it_end(sysws *w, int kp)
{
    const int L = w->L, LE = L + 1;
    double *Nw = w->N + kp * L, *Pw = w->P + kp * L, *Ew = w->E + kp * LE;
    for (int n = 0; n < L; n++) {                                               /* :218-222 */
        Nw[n] = w->Nk[n];
        Pw[n] = w->Pk[n];
        Ew[n] = w->Ek[n];
    }
}

/* shared_array_max, pvSimPCR.py:83-90 (a NaN in arr[i > 0] is skipped by `>`, one in arr[0] sticks) */
static double shared_array_max(const double *arr, int n)
{
    double m = arr[0];
    for (int i = 1; i < n; i++)
        if (arr[i] > m) m = arr[i];
    return m;
}

/* iterate() for num_sims systems that share the convergence test (num_sims = 1: the plain case) */
static int iterate_bundle(sysws *w, int num_sims, const double *mp /*[num_sims][12]*/, const double *a, int k, int kp,
                          const int *ko, double TOL, int MAX)
{
    double errN[16] = {0}, errP[16] = {0};                                      /* :143-145 */
    for (int y = 0; y < num_sims; y++) it_begin(&w[y], a, k, ko);
    int iters;
    for (iters = 0; iters < MAX; iters++) {                                     /* :147 */
        for (int y = 0; y < num_sims; y++) it_sweep(&w[y], mp + 12 * y, a[0], &errN[y], &errP[y]);
        if (shared_array_max(errN, num_sims) < TOL && shared_array_max(errP, num_sims) < TOL) break;   /* :211-216 */
    }
    for (int y = 0; y < num_sims; y++) it_end(&w[y], kp);
    /* Python: a loop that ran to exhaustion leaves iters = MAX-1; `return iters+1` (:225) */
    return (iters < MAX ? iters : MAX - 1) + 1;
}

static int iterate(sysws *w, const double *mp, const double *a, int k, int kp,
                   const int *ko, double TOL, int MAX)
{
    return iterate_bundle(w, 1, mp, a, k, kp, ko, TOL, MAX);
}

/* BDF coefficient table, tEvol pvSimPCR.py:241-250: step t uses order min(t + 1, 5).
 * g_max_order (oracle_set_max_order, default 5 = the reference) caps the order: step t then takes the row of step
 * min(t, max_order - 1).  max_order = 2 is the scheme of the reference's older solver Legacy/pvSim.py:94-97 (Euler at
 * t = 0, BDF2 ever after); with CN = CP = 0 the two files then discretise the same equations the same way and can be
 * compared over a whole curve, not only on the steps where BDF2 and the ramp coincide. */
static int g_max_order = 5;
int oracle_set_max_order(int k) { int old = g_max_order; if (k >= 1 && k <= 5) g_max_order = k; return old; }

static void bdf_coeffs(long t, double *a)
{
    if (t > g_max_order - 1) t = g_max_order - 1;
    if (t == 0)      { a[0] = 1.0;  a[1] = -1.0; a[2] = 0.0; a[3] = 0.0; a[4] = 0.0; a[5] = 0.0; }
    else if (t == 1) { a[0] = 1.5;  a[1] = -2.0; a[2] = 0.5; a[3] = 0.0; a[4] = 0.0; a[5] = 0.0; }
    else if (t == 2) { a[0] = 11.0 / 6; a[1] = -3.0; a[2] = 1.5; a[3] = -1.0 / 3; a[4] = 0.0; a[5] = 0.0; }
    else if (t == 3) { a[0] = 25.0 / 12; a[1] = -4.0; a[2] = 3.0; a[3] = -4.0 / 3; a[4] = 0.25; a[5] = 0.0; }
    else             { a[0] = 137.0 / 60; a[1] = -5.0; a[2] = 5.0; a[3] = -10.0 / 3; a[4] = 1.25; a[5] = -0.2; }
}

static int pymod6(long x) { long m = x % 6; return (int)(m < 0 ? m + 6 : m); }

/* Non-dimensionalisation scales, pvSim pvSimPCR.py:327-331 (Python float `**` is C pow). */
void oracle_scales(double length, double time_, int L, long T, double *scales /*12*/,
                   double *dx3_out, double *plnorm_out)
{
    double dx = length / L, dt = time_ / T;
    double dx3 = pow(dx, 3.0), dtdx = dt / dx, dtdx2 = dtdx / dx;
    double dtdx6 = dt / pow(dx, 6.0);
    double s[12] = { dx3, dx3, dtdx2, dtdx2, dtdx2 / dx, dtdx, dtdx, dtdx6, dtdx6,
                     1 / dt, 1 / dt, 1 / dx };
    memcpy(scales, s, sizeof s);
    if (dx3_out) *dx3_out = dx3;
    if (plnorm_out) *plnorm_out = pow(dx, 2.0) * dt;                            /* :393 */
}

/* ------------------------------------------------------------------------------------
 * oracle_pvsim -- pvSim (pvSimPCR.py:309-401, init_mode="points") + tEvol (:227-306).
 *
 *   matpar   [S][12] physical units (nm, ns), row-major: N0,P0,DN,DP,rate,sr0,srL,CN,CP,
 *            tauN,tauP,Lambda (:97-108)
 *   inipar   [L] excess carrier density, nm^-3 (:355-356)
 *   plI      [S][ldp] output, element size plI_bytes (4: float, 8: double).  As in the
 *            reference the kernel stores rate*Sum in the buffer's dtype (:281) and the host
 *            then divides by dx^2*dt in that dtype (:393).
 *   status   [S] 0 = converged everywhere; 1+t = `iters >= MAX` at step t (:269).  Unlike
 *            the reference (whose launch-wide race flag stops every block, :290-292) only
 *            the offending system stops; its remaining PL entries are set to NaN.
 *   iters_total/iters_max [S] (optional) sum / max of iterate()'s return over the steps run.
 *   step_iters [S][T+1] (optional) iterate()'s return per step.
 * Returns 0, or -1 on bad arguments / allocation failure.
 * ---------------------------------------------------------------------------------- */
/*
 * oracle_pvsim_snap additionally fills the reference's debug outputs plN, plP [S][n_snap][L] and
 * plE [S][n_snap][L+1] (each nullable): the state of level k at the time steps snap_steps[i].  In
 * pvSimPCR.py the recording hook is commented out (:283-288, right after the PL sum, reading level k);
 * the working form is Legacy/pvSim.py:121-126 -- `if t in pT: ind = pT.index(t)` (first position of a
 * repeated step), all L + 1 field edges copied -- with the re-dimensionalisation of :169-171
 * (plN, plP /= dx**3; plE /= dx).  Slots that are not reached keep their contents; a flagged system's
 * slots from its failing step on are NaN (oracle convention, like its PL).
 */
int oracle_pvsim_snap(const double *matpar, long S, double length, double time_, int L, long T,
                      int plT, int tol, int MAX, const double *inipar, void *plI, int plI_bytes,
                      long ldp, int32_t *status, int64_t *iters_total, int32_t *iters_max,
                      int32_t *step_iters, const long *snap_steps, int n_snap, double *plN, double *plP,
                      double *plE, int nthreads)
{
    if (L < 4 || (L & (L - 1)) || T < 1 || plT < 1 || (plI_bytes != 4 && plI_bytes != 8) || n_snap < 0)
        return -1;
    const double dx = length / L;                                               /* Legacy/pvSim.py:133 */
    double scales[12], dx3, plnorm;
    oracle_scales(length, time_, L, T, scales, &dx3, &plnorm);
    const double TOL = pow(10.0, -(double)tol);                                 /* :112 */
    int fail = 0;
    (void)nthreads;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (long y = 0; y < S; y++) {
        sysws w;
        if (ws_alloc(&w, L)) { fail = 1; continue; }
        double mp[12];
        for (int i = 0; i < 12; i++) mp[i] = matpar[y * 12 + i] * scales[i];    /* :331 */
        for (int n = 0; n < L; n++) {                                           /* :356-362 */
            double dN = inipar[n] * dx3;
            w.N[n] = mp[0] + dN;
            w.P[n] = mp[1] + dN;
        }
        int st = 0, imax = 0;
        int64_t itot = 0;
        float *pf = (float *)plI + y * ldp;
        double *pd = (double *)plI + y * ldp;
        long t;
        for (t = 0; t <= T; t++) {                                              /* :237 */
            double a[6];
            bdf_coeffs(t, a);
            int kp = pymod6(t + 1), k = pymod6(t);                              /* :251-256 */
            int ko[4] = { pymod6(t - 1), pymod6(t - 2), pymod6(t - 3), pymod6(t - 4) };
            int it = iterate(&w, mp, a, k, kp, ko, TOL, MAX);                   /* :266 */
            itot += it;
            if (it > imax) imax = it;
            if (step_iters) step_iters[y * (T + 1) + t] = it;
            if (it >= MAX) { st = 1 + (int)t; break; }                          /* :269-274 */
            if (t % plT == 0) {                                                 /* :276-281 */
                double Sum = -(double)L * (mp[0] * mp[1]);
                const double *Nc = w.N + k * L, *Pc = w.P + k * L;
                for (int n = 0; n < L; n++) Sum += Nc[n] * Pc[n];
                double v = mp[4] * Sum;
                if (plI_bytes == 4) pf[t / plT] = (float)v / (float)plnorm;     /* :281,:393 */
                else                pd[t / plT] = v / plnorm;
            }
            for (int ind = 0; ind < n_snap; ind++) {                            /* :283-288, Legacy :121-126 */
                if (snap_steps[ind] != t) continue;
                const size_t at = (size_t)y * n_snap + ind;
                for (int n = 0; n < L; n++) {
                    if (plN) plN[at * L + n] = w.N[k * L + n] / dx3;            /* Legacy :169 */
                    if (plP) plP[at * L + n] = w.P[k * L + n] / dx3;            /* :170 */
                }
                if (plE) for (int n = 0; n <= L; n++) plE[at * (L + 1) + n] = w.E[k * (L + 1) + n] / dx;   /* :171 */
                break;                                                          /* pT.index(t): first match */
            }
        }
        if (st) {
            for (long tt = t; tt <= T; tt++)
                if (tt % plT == 0) {
                    if (plI_bytes == 4) pf[tt / plT] = NAN; else pd[tt / plT] = NAN;
                }
            for (int ind = 0; ind < n_snap; ind++) {
                int first = 1;
                for (int q = 0; q < ind; q++) if (snap_steps[q] == snap_steps[ind]) first = 0;
                if (!first || snap_steps[ind] < t || snap_steps[ind] > T) continue;
                const size_t at = (size_t)y * n_snap + ind;
                for (int n = 0; n < L; n++) { if (plN) plN[at * L + n] = NAN; if (plP) plP[at * L + n] = NAN; }
                if (plE) for (int n = 0; n <= L; n++) plE[at * (L + 1) + n] = NAN;
            }
        }
        if (status) status[y] = st;
        if (iters_total) iters_total[y] = itot;
        if (iters_max) iters_max[y] = imax;
        ws_free(&w);
    }
    return fail ? -1 : 0;
}

/* ------------------------------------------------------------------------------------
 * oracle_pvsim_bundle -- pvSim with max_sims_per_block = mspb > 1 (pvSimPCR.py:258-266): the samples
 * p .. min(p + mspb, S) - 1 form a bundle whose systems iterate in lockstep until the LARGEST errN and errP of
 * the bundle are below TOL (:211-216, shared_array_max :83-90); iterate()'s return is the bundle's.  A bundle
 * that reaches MAX is flagged as a whole (every sample gets status 1 + t and NaN from there on; the reference
 * stops the launch, :269-274,:290-292).  Arguments as oracle_pvsim; mspb in [1, 16].
 * ---------------------------------------------------------------------------------- */
int oracle_pvsim_bundle(const double *matpar, long S, double length, double time_, int L, long T,
                        int plT, int tol, int MAX, const double *inipar, void *plI, int plI_bytes,
                        long ldp, int32_t *status, int64_t *iters_total, int32_t *iters_max,
                        int32_t *step_iters, int mspb, int nthreads)
{
    if (L < 4 || (L & (L - 1)) || T < 1 || plT < 1 || (plI_bytes != 4 && plI_bytes != 8) || mspb < 1 || mspb > 16)
        return -1;
    double scales[12], dx3, plnorm;
    oracle_scales(length, time_, L, T, scales, &dx3, &plnorm);
    const double TOL = pow(10.0, -(double)tol);                                 /* :112 */
    const long nb = (S + mspb - 1) / mspb;
    int fail = 0;
    (void)nthreads;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (long b = 0; b < nb; b++) {
        const long p = b * mspb;
        const int ns = (int)((p + mspb <= S ? p + mspb : S) - p);               /* p_lim - p, :265 */
        sysws w[16];
        double mp[16 * 12];
        int ok = 1;
        for (int y = 0; y < ns; y++) if (ws_alloc(&w[y], L)) { ok = 0; for (int q = 0; q < y; q++) ws_free(&w[q]); break; }
        if (!ok) { fail = 1; continue; }
        for (int y = 0; y < ns; y++) {
            for (int i = 0; i < 12; i++) mp[12 * y + i] = matpar[(p + y) * 12 + i] * scales[i];   /* :331 */
            for (int n = 0; n < L; n++) {                                       /* :356-362 */
                double dN = inipar[n] * dx3;
                w[y].N[n] = mp[12 * y] + dN;
                w[y].P[n] = mp[12 * y + 1] + dN;
            }
        }
        int st = 0, imax = 0;
        int64_t itot = 0;
        long t;
        for (t = 0; t <= T; t++) {                                              /* :237 */
            double a[6];
            bdf_coeffs(t, a);
            int kp = pymod6(t + 1), k = pymod6(t);                              /* :251-256 */
            int ko[4] = { pymod6(t - 1), pymod6(t - 2), pymod6(t - 3), pymod6(t - 4) };
            int it = iterate_bundle(w, ns, mp, a, k, kp, ko, TOL, MAX);         /* :266 */
            itot += it;
            if (it > imax) imax = it;
            if (step_iters) for (int y = 0; y < ns; y++) step_iters[(p + y) * (T + 1) + t] = it;
            if (it >= MAX) { st = 1 + (int)t; break; }                          /* :269-274 */
            if (t % plT == 0) {                                                 /* :276-281 */
                for (int y = 0; y < ns; y++) {
                    double Sum = -(double)L * (mp[12 * y] * mp[12 * y + 1]);
                    const double *Nc = w[y].N + k * L, *Pc = w[y].P + k * L;
                    for (int n = 0; n < L; n++) Sum += Nc[n] * Pc[n];
                    double v = mp[12 * y + 4] * Sum;
                    if (plI_bytes == 4) ((float *)plI)[(p + y) * ldp + t / plT] = (float)v / (float)plnorm;   /* :281,:393 */
                    else                ((double *)plI)[(p + y) * ldp + t / plT] = v / plnorm;
                }
            }
        }
        for (int y = 0; y < ns; y++) {
            if (st)
                for (long tt = t; tt <= T; tt++)
                    if (tt % plT == 0) {
                        if (plI_bytes == 4) ((float *)plI)[(p + y) * ldp + tt / plT] = NAN;
                        else                ((double *)plI)[(p + y) * ldp + tt / plT] = NAN;
                    }
            if (status) status[p + y] = st;
            if (iters_total) iters_total[p + y] = itot;
            if (iters_max) iters_max[p + y] = imax;
            ws_free(&w[y]);
        }
    }
    return fail ? -1 : 0;
}

int oracle_pvsim(const double *matpar, long S, double length, double time_, int L, long T,
                 int plT, int tol, int MAX, const double *inipar, void *plI, int plI_bytes,
                 long ldp, int32_t *status, int64_t *iters_total, int32_t *iters_max,
                 int32_t *step_iters, int nthreads)
{
    return oracle_pvsim_snap(matpar, S, length, time_, L, T, plT, tol, MAX, inipar, plI, plI_bytes, ldp, status,
                             iters_total, iters_max, step_iters, NULL, 0, NULL, NULL, NULL, nthreads);
}

/* ------------------------------------------------------------------------------------
 * fastlog / log_kernel -- probs.py:64-85:  x <- log10(max(x, MIN)) in place, in the
 * buffer's dtype.  For a float32 buffer a clamped value is stored as (float)MIN, which is
 * 0.0f for MIN = DBL_MIN (bayeslib.py:157), so the result is -inf.
 * ---------------------------------------------------------------------------------- */
void oracle_fastlog(void *x, int bytes, long rows, long cols, long ld, double MIN)
{
    for (long i = 0; i < rows; i++)
        for (long j = 0; j < cols; j++) {
            if (bytes == 4) {
                float *p = (float *)x + i * ld + j;
                if ((double)*p < MIN) *p = (float)MIN;                          /* :72-73 */
                *p = (float)log10((double)*p);                                  /* :75 */
            } else {
                double *p = (double *)x + i * ld + j;
                if (*p < MIN) *p = MIN;
                *p = log10(*p);
            }
        }
}

/* ------------------------------------------------------------------------------------
 * prob / kernel_lnP -- probs.py:20-62:  P[j] -= sum_i (plI[j,i] + mag[j] - values[i])^2,
 * accumulated sequentially in fp64 (:32-41); `uncertainty` is unused by the reference (:40).
 * ---------------------------------------------------------------------------------- */
void oracle_prob(double *P, const void *plI, int bytes, long rows, long nobs, long ld,
                 const double *values, const double *mag)
{
    for (long j = 0; j < rows; j++) {
        double acc = 0;
        for (long i = 0; i < nobs; i++) {
            double v = bytes == 4 ? (double)((const float *)plI)[j * ld + i]
                                  : ((const double *)plI)[j * ld + i];
            double err = v + mag[j];
            err -= values[i];
            err = err * err;
            acc += err;
        }
        P[j] += (0.0 - acc);                                                    /* :44,:57,:60 */
    }
}
