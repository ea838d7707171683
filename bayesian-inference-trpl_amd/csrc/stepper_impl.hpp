// TRPL time-stepper for gfx950: ONE WAVEFRONT OWNS ONE SYSTEM (sample x curve) FOR ALL T STEPS.
//
// What it computes (reference: pvSimPCR.py tEvol :227-306, iterate :93-225, pcreduce :42-81,
// norm2 :14-40; likelihood probs.py:20-47, :64-75): variable-order BDF in time; per step a
// Newton/Picard iteration whose two tridiagonal systems are solved by parallel cyclic
// reduction; PL(t) by midpoint quadrature; optionally log10 + squared error against
// observations, fused, so PL never reaches memory.
//
// Layout: node i = ln + W*j, ln = lane (W = min(L,64) lanes), j < NR = L/W rows per lane.
//   * PCR strides 1..W/2 are cross-lane rotations, strides >= W and the final 2x2 solves
//     (pairs i, i+L/2) are intra-lane when NR >= 2;
//   * the reference's power-of-two reduction tree (norm2 :32-38) becomes intra-lane adds
//     followed by an xor butterfly: same association, so the residual norms -- and with
//     them every convergence decision -- are bit-identical in STRICT mode.
// State (N,P,E), the 5 BDF history levels and the BDF right-hand sides live in VGPRs for the
// whole run; the 12 material parameters are wave-uniform.  HBM traffic per system is 13
// doubles in, one double out (likelihood mode) -- the kernel is fp64-VALU / cross-lane bound
// by construction, not HBM bound.
//
// Included by stepper_strict.hip (compiled -ffp-contract=off, STRICT=true: IEEE divides,
// reference operation order -> bit-identical state) and stepper_fast.hip (contraction on,
// STRICT=false).
#pragma once
#include <math.h>
#include <float.h>

#include "trpl_common.hpp"

namespace trpl {

constexpr uint32_t kFlagPlF32 = 0x2;       // TRPL_FLAG_PL_F32
constexpr uint32_t kFlagNormalize = 0x4;   // TRPL_FLAG_NORMALIZE

__device__ __forceinline__ double uniform_d(double v)
{
    // broadcast lane 0's value through SGPRs so the compiler knows it is wave-uniform
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
}

// y[j] = x at node i+RF (any finite in-array value when i+RF >= L)
template <typename T, int NR, int W, int RF>
__device__ __forceinline__ void fetch_up(const T (&x)[NR], T (&y)[NR], int ln)
{
    if constexpr (RF >= W) {
        constexpr int m = RF / W;
#pragma unroll
        for (int j = 0; j < NR; j++) y[j] = x[(j + m) % NR];
    } else {
        const int src = (ln + RF) & (W - 1);
        const bool wrap = ln + RF >= W;
        T s[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) s[j] = __shfl(x[j], src, 64);
#pragma unroll
        for (int j = 0; j < NR; j++) y[j] = wrap ? s[(j + 1) % NR] : s[j];
    }
}

// y[j] = x at node i-RF (any finite in-array value when i < RF)
template <typename T, int NR, int W, int RF>
__device__ __forceinline__ void fetch_dn(const T (&x)[NR], T (&y)[NR], int ln)
{
    if constexpr (RF >= W) {
        constexpr int m = RF / W;
#pragma unroll
        for (int j = 0; j < NR; j++) y[j] = x[(j + NR - m) % NR];
    } else {
        const int src = (ln - RF) & (W - 1);
        const bool wrap = ln < RF;
        T s[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) s[j] = __shfl(x[j], src, 64);
#pragma unroll
        for (int j = 0; j < NR; j++) y[j] = wrap ? s[(j + NR - 1) % NR] : s[j];
    }
}

// Sum over all L nodes with the reference's tree association (norm2, pvSimPCR.py:32-38):
// level rf pairs (i, i+rf), rf = L/2 ... 1.  Every lane ends with the same value.
template <typename T, int NR, int W>
__device__ __forceinline__ T tree_sum(T (&v)[NR])
{
#pragma unroll
    for (int m = NR / 2; m >= 1; m /= 2)
#pragma unroll
        for (int j = 0; j < m; j++) v[j] = v[j] + v[j + m];
    T r = v[0];
#pragma unroll
    for (int off = W / 2; off >= 1; off /= 2) r = r + __shfl_xor(r, off, 64);
    return r;
}

// One PCR level (pvSimPCR.py:57-69) with stride RF on the snapshot semantics of :49-54.
template <typename T, int NR, int W, int L, int RF>
__device__ __forceinline__ void pcr_level(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], int ln)
{
    T ld_m[NR], d_m[NR], ud_m[NR], B_m[NR], ld_p[NR], d_p[NR], ud_p[NR], B_p[NR];
    fetch_dn<T, NR, W, RF>(ld, ld_m, ln);
    fetch_dn<T, NR, W, RF>(d, d_m, ln);
    fetch_dn<T, NR, W, RF>(ud, ud_m, ln);
    fetch_dn<T, NR, W, RF>(B, B_m, ln);
    fetch_up<T, NR, W, RF>(ld, ld_p, ln);
    fetch_up<T, NR, W, RF>(d, d_p, ln);
    fetch_up<T, NR, W, RF>(ud, ud_p, ln);
    fetch_up<T, NR, W, RF>(B, B_p, ln);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int i = ln + W * j;
        const bool lo = i >= RF, hi = i < L - RF;
        const T k1 = lo ? ld[j] / d_m[j] : T(0);
        const T k2 = hi ? ud[j] / d_p[j] : T(0);
        T dn = d[j] - ud_m[j] * k1;
        T Bn = B[j] - B_m[j] * k1;
        const T ldn = lo ? -ld_m[j] * k1 : ld[j];
        dn = dn - ld_p[j] * k2;
        Bn = Bn - B_p[j] * k2;
        const T udn = hi ? -ud_p[j] * k2 : ud[j];
        d[j] = dn; B[j] = Bn; ld[j] = ldn; ud[j] = udn;
    }
}

template <typename T, int NR, int W, int L, int RF>
__device__ __forceinline__ void pcr_levels(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], int ln)
{
    if constexpr (L > 2 * RF) {
        pcr_level<T, NR, W, L, RF>(ld, d, ud, B, ln);
        pcr_levels<T, NR, W, L, RF * 2>(ld, d, ud, B, ln);
    }
}

// Tridiagonal solve (pcreduce, pvSimPCR.py:42-81): destroys ld,d,ud,B; result in x.
template <typename T, int NR, int W, int L>
__device__ __forceinline__ void pcr_solve(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], T (&x)[NR],
                                          int ln)
{
    pcr_levels<T, NR, W, L, 1>(ld, d, ud, B, ln);
    if constexpr (NR >= 2) {                       // pairs (i, i+L/2) are (j, j+NR/2) in-lane
        constexpr int H = NR / 2;
#pragma unroll
        for (int j = 0; j < H; j++) {              // pvSimPCR.py:75-79
            const T k = ud[j] / d[j + H];
            x[j] = (B[j] - B[j + H] * k) / (d[j] - ld[j + H] * k);
            x[j + H] = (B[j + H] - ld[j + H] * x[j]) / d[j + H];
        }
    } else {                                        // L <= 64: partner lane ln ^ L/2
        constexpr int H = W / 2;
        const bool low = (ln & H) == 0;
        const T ud_o = __shfl_xor(ud[0], H, 64), d_o = __shfl_xor(d[0], H, 64),
                B_o = __shfl_xor(B[0], H, 64), ld_o = __shfl_xor(ld[0], H, 64);
        const T l_ud = low ? ud[0] : ud_o, l_d = low ? d[0] : d_o, l_B = low ? B[0] : B_o;
        const T h_d = low ? d_o : d[0], h_B = low ? B_o : B[0], h_ld = low ? ld_o : ld[0];
        const T k = l_ud / h_d;
        const T xl = (l_B - h_B * k) / (l_d - h_ld * k);
        const T xh = (h_B - h_ld * xl) / h_d;
        x[0] = low ? xl : xh;
    }
}

// Relative L1 residual of iterate c in the system (lower l, diagonal dg, upper u | b):
// norm2, pvSimPCR.py:14-40.
template <int NR, int W>
__device__ __forceinline__ double residual_norm(const double (&l)[NR], const double (&dg)[NR],
                                                const double (&u)[NR], const double (&b)[NR],
                                                const double (&c)[NR], int ln)
{
    double cm[NR], cp[NR], r[NR], ab[NR];
    fetch_dn<double, NR, W, 1>(c, cm, ln);
    fetch_up<double, NR, W, 1>(c, cp, ln);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        // l = 0 on row 0 and u = 0 on row L-1, so the wrapped neighbour contributes +-0
        r[j] = fabs(l[j] * cm[j] + dg[j] * c[j] + u[j] * cp[j] - b[j]);
        ab[j] = fabs(b[j]);
    }
    const double sr = tree_sum<double, NR, W>(r);
    const double sb = tree_sum<double, NR, W>(ab);
    return sr / sb;
}

template <int L, bool STRICT>
__global__ void __launch_bounds__(64) stepper_kernel(const StepArgs a)
{
    constexpr int W = L < 64 ? L : 64;
    constexpr int NR = L / W;
    const int ln = threadIdx.x & (W - 1);          // lanes >= W replicate lane (lane mod W)
    const int64_t sys = blockIdx.x;
    const int c = (int)(sys % a.C);
    const int64_t s = sys / a.C;
    const CurveConst &cc = a.curve[c];
    const int64_t orow = (int64_t)c * a.S + s;

    // ---- non-dimensional material parameters (pvSimPCR.py:327-331) ----
    const double *xs = a.X + s * a.xld;
    const double N0 = xs[0] * cc.scales[0], P0 = xs[1] * cc.scales[1], DN = xs[2] * cc.scales[2],
                 DP = xs[3] * cc.scales[3], rate = xs[4] * cc.scales[4], sr0 = xs[5] * cc.scales[5],
                 srL = xs[6] * cc.scales[6], CN = xs[7] * cc.scales[7], CP = xs[8] * cc.scales[8],
                 tauN = xs[9] * cc.scales[9], tauP = xs[10] * cc.scales[10],
                 Lambda = xs[11] * cc.scales[11];
    const double n0p0 = N0 * P0;
    const double mag = a.xld > 12 ? xs[12] : 0.0;
    const double TOL = a.TOL;
    const int MAX = a.MAX;

    // ---- state + history: h?[0] = level k (time t), h?[m] = level k-m ----
    double hN[5][NR], hP[5][NR], hE[5][NR];
#pragma unroll
    for (int m = 0; m < 5; m++)
#pragma unroll
        for (int j = 0; j < NR; j++) { hN[m][j] = 0.0; hP[m][j] = 0.0; hE[m][j] = 0.0; }
#pragma unroll
    for (int j = 0; j < NR; j++) {                 // pvSimPCR.py:356-362
        const double dn = a.dN[(int64_t)c * L + ln + W * j] * cc.dx3;
        hN[0][j] = N0 + dn;
        hP[0][j] = P0 + dn;
    }

    const bool want_pl = a.pl != nullptr;
    const bool want_ll = a.sse != nullptr;
    const int64_t ncol_ll = want_ll ? cc.n_obs : 0;
    // last step that can influence an output: all T+1 of them when PL is stored (the reference
    // runs them all), otherwise up to the last compared column
    const int64_t t_last = want_pl ? a.T : (ncol_ll - 1) * a.plT;
    const double *obs = want_ll ? a.obs + (int64_t)c * a.obs_ld : nullptr;
    double sse = 0.0;
    double pl0_d = 1.0;
    float pl0_f = 1.0f;
    int status = 0;
    int64_t itot = 0;

    for (int64_t t = 0; t <= t_last; t++) {        // tEvol, pvSimPCR.py:237
        double a0, a1, a2, a3, a4, a5;             // :241-250
        if (t == 0)      { a0 = 1.0; a1 = -1.0; a2 = 0.0; a3 = 0.0; a4 = 0.0; a5 = 0.0; }
        else if (t == 1) { a0 = 1.5; a1 = -2.0; a2 = 0.5; a3 = 0.0; a4 = 0.0; a5 = 0.0; }
        else if (t == 2) { a0 = 11.0 / 6; a1 = -3.0; a2 = 1.5; a3 = -1.0 / 3; a4 = 0.0; a5 = 0.0; }
        else if (t == 3) { a0 = 25.0 / 12; a1 = -4.0; a2 = 3.0; a3 = -4.0 / 3; a4 = 0.25; a5 = 0.0; }
        else             { a0 = 137.0 / 60; a1 = -5.0; a2 = 5.0; a3 = -10.0 / 3; a4 = 1.25; a5 = -0.2; }

        // PL of the state at time t (level k), pvSimPCR.py:276-281.  Summation order: tree
        // instead of the reference's serial loop (documented deviation, ~1e-16 relative).
        double plv = 0.0;
        const bool pl_step = (t % a.plT) == 0;
        if (pl_step) {
            double q[NR];
#pragma unroll
            for (int j = 0; j < NR; j++) q[j] = hN[0][j] * hP[0][j];
            const double Sum = tree_sum<double, NR, W>(q) + (-(double)L * n0p0);
            plv = rate * Sum;
        }

        // ---------------- iterate, pvSimPCR.py:93-225 ----------------
        double Nk[NR], Pk[NR], Ek[NR], bN[NR], bP[NR], bE[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) {             // :128-135
            Nk[j] = hN[0][j]; Pk[j] = hP[0][j]; Ek[j] = hE[0][j];
            bN[j] = a1 * Nk[j] + a2 * hN[1][j] + a3 * hN[2][j] + a4 * hN[3][j] + a5 * hN[4][j];
            bP[j] = a1 * Pk[j] + a2 * hP[1][j] + a3 * hP[2][j] + a4 * hP[3][j] + a5 * hP[4][j];
            bE[j] = a1 * Ek[j] + a2 * hE[1][j] + a3 * hE[2][j] + a4 * hE[3][j] + a5 * hE[4][j];
        }
        int it = MAX;                              // value if the loop runs to exhaustion (:225)
        for (int iters = 0; iters < MAX; iters++) {
            double lo_[NR], dg[NR], up[NR], bb[NR], Ep[NR];
            // ---- electrons (:148-175) ----
            fetch_up<double, NR, W, 1>(Ek, Ep, ln);
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int i = ln + W * j;
                const bool first = i == 0, last = i == L - 1;
                const double u_i = last ? 0.0 : DN * (-Ep[j] / 2 - 1);    // A0[i]
                const double l_i = first ? 0.0 : DN * (+Ek[j] / 2 - 1);   // A2[i]
                const double u_m = first ? 0.0 : DN * (-Ek[j] / 2 - 1);   // A0[i-1]
                const double l_p = last ? 0.0 : DN * (+Ep[j] / 2 - 1);    // A2[i+1]
                const double tp = Nk[j] * tauP + Pk[j] * tauN;
                const double np_ = Nk[j] * Pk[j] - n0p0;
                const double ds = -rate * Pk[j] - (Pk[j] * tp - tauP * np_) / (tp * tp)
                                - (CN * Nk[j] * Pk[j] + CP * (Pk[j] * Pk[j]) + CN * np_);
                up[j] = u_i; lo_[j] = l_i;
                dg[j] = a0 - u_m - l_p - ds;
                bb[j] = -(CN * Nk[j] + CP * Pk[j] + rate + 1 / tp) * np_ - ds * Nk[j] - bN[j];
            }
            {   // surfaces (:164-170): node 0 is (lane 0, row 0), node L-1 is (lane W-1, row NR-1)
                const double s0 = Nk[0] + Pk[0], sL = Nk[NR - 1] + Pk[NR - 1];
                const double ds0 = -sr0 * (Pk[0] * Pk[0] + n0p0) / (s0 * s0);
                const double dsL = -srL * (Pk[NR - 1] * Pk[NR - 1] + n0p0) / (sL * sL);
                const double f0 = sr0 * (Nk[0] * Pk[0] - n0p0) / s0 + ds0 * Nk[0];
                const double fL = srL * (Nk[NR - 1] * Pk[NR - 1] - n0p0) / sL + dsL * Nk[NR - 1];
                if (ln == 0) { dg[0] -= ds0; bb[0] -= f0; }
                if (ln == W - 1) { dg[NR - 1] -= dsL; bb[NR - 1] -= fL; }
            }
            const double errN = uniform_d(residual_norm<NR, W>(lo_, dg, up, bb, Nk, ln));  // :172
            pcr_solve<double, NR, W, L>(lo_, dg, up, bb, Nk, ln);                                  // :175

            // ---- holes, with the updated electrons (:178-202) ----
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int i = ln + W * j;
                const bool first = i == 0, last = i == L - 1;
                const double u_i = last ? 0.0 : DP * (+Ep[j] / 2 - 1);
                const double l_i = first ? 0.0 : DP * (-Ek[j] / 2 - 1);
                const double u_m = first ? 0.0 : DP * (+Ek[j] / 2 - 1);
                const double l_p = last ? 0.0 : DP * (-Ep[j] / 2 - 1);
                const double np_ = Nk[j] * Pk[j] - n0p0;
                const double tp = Nk[j] * tauP + Pk[j] * tauN;
                const double ds = -rate * Nk[j] - (Nk[j] * tp - tauN * np_) / (tp * tp)
                                - (CP * Nk[j] * Pk[j] + CN * (Nk[j] * Nk[j]) + CP * np_);
                up[j] = u_i; lo_[j] = l_i;
                dg[j] = a0 - u_m - l_p - ds;
                bb[j] = -(CN * Nk[j] + CP * Pk[j] + rate + 1 / tp) * np_ - ds * Pk[j] - bP[j];
            }
            {   // :192-198
                const double s0 = Nk[0] + Pk[0], sL = Nk[NR - 1] + Pk[NR - 1];
                const double ds0 = -sr0 * (Nk[0] * Nk[0] + n0p0) / (s0 * s0);
                const double dsL = -srL * (Nk[NR - 1] * Nk[NR - 1] + n0p0) / (sL * sL);
                const double f0 = sr0 * (Nk[0] * Pk[0] - n0p0) / s0 + ds0 * Pk[0];
                const double fL = srL * (Nk[NR - 1] * Pk[NR - 1] - n0p0) / sL + dsL * Pk[NR - 1];
                if (ln == 0) { dg[0] -= ds0; bb[0] -= f0; }
                if (ln == W - 1) { dg[NR - 1] -= dsL; bb[NR - 1] -= fL; }
            }
            const double errP = uniform_d(residual_norm<NR, W>(lo_, dg, up, bb, Pk, ln));  // :200
            pcr_solve<double, NR, W, L>(lo_, dg, up, bb, Pk, ln);                                  // :202

            // ---- field on edges 1..L-1 (:205-209) ----
            double Nm[NR], Pm[NR];
            fetch_dn<double, NR, W, 1>(Nk, Nm, ln);
            fetch_dn<double, NR, W, 1>(Pk, Pm, ln);
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int i = ln + W * j;
                const double A = Lambda * (DP * (Pk[j] + Pm[j]) + DN * (Nk[j] + Nm[j])) / 2 + a0;
                const double b = Lambda * (DP * (Pk[j] - Pm[j]) - DN * (Nk[j] - Nm[j])) - bE[j];
                const double e = b / A;
                Ek[j] = i >= 1 ? e : Ek[j];
            }
            if (errN < TOL && errP < TOL) { it = iters + 1; break; }                       // :213-216
        }
        itot += it;
        if (it >= MAX) { status = 1 + (int)t; break; }                                     // :269-274

        // ---- emit PL(t) ----
        if (pl_step) {
            const int64_t col = t / a.plT;
            if (want_pl && threadIdx.x == 0) {                                             // :281,:393
                if (a.pl_bytes == 4) ((float *)a.pl)[orow * a.pl_ld + col] = (float)plv / (float)cc.plnorm;
                else                 ((double *)a.pl)[orow * a.pl_ld + col] = plv / cc.plnorm;
            }
            if (col < ncol_ll) {                   // bayeslib.py:150-157, probs.py:29-44
                double lg;
                if (a.flags & kFlagPlF32) {
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
                double err = lg + mag;
                err -= obs[col];
                sse += err * err;
            }
        }

        // ---- rotate history: level kp becomes level k ----
#pragma unroll
        for (int j = 0; j < NR; j++) {
#pragma unroll
            for (int m = 4; m >= 1; m--) { hN[m][j] = hN[m - 1][j]; hP[m][j] = hP[m - 1][j]; hE[m][j] = hE[m - 1][j]; }
            hN[0][j] = Nk[j]; hP[0][j] = Pk[j]; hE[0][j] = Ek[j];
        }
    }

    if (threadIdx.x == 0) {
        if (status && want_pl) {                   // undefined in the reference; NaN here
            const int64_t t0 = status - 1;
            for (int64_t tt = t0; tt <= a.T; tt++)
                if (tt % a.plT == 0) {
                    const int64_t col = tt / a.plT;
                    if (a.pl_bytes == 4) ((float *)a.pl)[orow * a.pl_ld + col] = __builtin_nanf("");
                    else                 ((double *)a.pl)[orow * a.pl_ld + col] = __builtin_nan("");
                }
        }
        if (want_ll) a.sse[orow] = status ? __builtin_inf() : sse;
        if (a.status) a.status[orow] = status;
        if (a.iters_total) a.iters_total[orow] = itot;
    }
}

template <bool STRICT>
hipError_t launch_stepper(const StepArgs &a, hipStream_t stream)
{
    const int64_t nsys = a.S * a.C;
    if (nsys <= 0) return hipSuccess;
    dim3 grid((unsigned)nsys), block(64);
    switch (a.L) {
#define TRPL_CASE(LL) \
    case LL: hipLaunchKernelGGL((stepper_kernel<LL, STRICT>), grid, block, 0, stream, a); break;
        TRPL_CASE(4) TRPL_CASE(8) TRPL_CASE(16) TRPL_CASE(32) TRPL_CASE(64) TRPL_CASE(128)
        TRPL_CASE(256) TRPL_CASE(512)
#undef TRPL_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace trpl
