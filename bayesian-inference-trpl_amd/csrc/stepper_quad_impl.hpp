// FAST time-stepper for L = 128 with FOUR SYSTEMS PER WAVEFRONT (same curve, adjacent samples): a system owns
// one DPP row of 16 lanes, a lane 8 adjacent rows of the matrix.
//
// Why: the in-lane cyclic-reduction levels of the tridiagonal solve are work-efficient, the cross-lane PCR
// levels are not (stepper_pair_impl.hpp).  8 rows per lane: 3 CR levels + 3 PCR levels + the pair step, and
// every cross-lane move stays inside a 16-lane row, i.e. is a plain row DPP move (no LDS crossbar round trip,
// no seam select: a lane whose neighbour lies outside the row reads an exact 0, so nothing of another system
// is ever read -- isolation is structural).  Price: ~300 live registers, ONE wave per SIMD.
//
// Convergence, failure handling and PL emission are per system, exactly as in the paired kernel.
#pragma once
#include "stepper_impl.hpp"

namespace trpl {
namespace quad {

constexpr int L = 128;      // nodes per system
constexpr int NR = 8;       // adjacent rows per lane
constexpr int WS = 16;      // lanes per system
constexpr int NS = 4;       // systems per wave
constexpr int XM = 8;       // row-DPP exchanges (pcr.hpp)

__device__ __forceinline__ double lane_value(double v, int l)      // lane l's value, wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

// sums of v over each row of 16 lanes, wave-uniform; the same association in every row
__device__ __forceinline__ void row_sums(double v, double (&s)[NS])
{
    v = dpp_add<0x111, 0xF>(v);          // row_shr:1
    v = dpp_add<0x112, 0xF>(v);          // row_shr:2
    v = dpp_add<0x114, 0xF>(v);          // row_shr:4
    v = dpp_add<0x118, 0xF>(v);          // row_shr:8   -> lane 15 of each row holds the row sum
#pragma unroll
    for (int g = 0; g < NS; g++) s[g] = lane_value(v, 16 * g + 15);
}

// y[j] = x at node i-1 / i+1; beyond the system's ends: 0
__device__ __forceinline__ void nbr_dn(const double (&x)[NR], double (&y)[NR])
{
    y[0] = dpp_row<0x111>(x[NR - 1]);
#pragma unroll
    for (int j = 1; j < NR; j++) y[j] = x[j - 1];
}
__device__ __forceinline__ void nbr_up(const double (&x)[NR], double (&y)[NR])
{
#pragma unroll
    for (int j = 0; j < NR - 1; j++) y[j] = x[j + 1];
    y[NR - 1] = dpp_row<0x101>(x[0]);
}

// norm2 (pvSimPCR.py:14-40) of the four systems: ok = sum|A c - b| < TOL * sum|b|, one reduction each
__device__ __forceinline__ void residual_below4(const double (&l)[NR], const double (&dg)[NR], const double (&u)[NR],
                                                const double (&b)[NR], const double (&c)[NR], double TOL, bool (&ok)[NS])
{
    double cm[NR], cp[NR];
    nbr_dn(c, cm);
    nbr_up(c, cp);
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const double r = fabs(__builtin_fma(l[j], cm[j], __builtin_fma(dg[j], c[j], __builtin_fma(u[j], cp[j], -b[j]))));
        const double qj = __builtin_fma(-TOL, fabs(b[j]), r);
        q = j == 0 ? qj : q + qj;
    }
    double s[NS];
    row_sums(q, s);
#pragma unroll
    for (int g = 0; g < NS; g++) ok[g] = s[g] < 0.0;
}

// field update on edges 1..L-1 (pvSimPCR.py:205-209) of the systems whose lanes have act set
__device__ __forceinline__ void update_field4(const MatPar &m, double a0, const double (&Nk)[NR], const double (&Pk)[NR],
                                              const double (&bE)[NR], double (&Ek)[NR], int ln, bool act)
{
    double Nm[NR], Pm[NR], A[NR], b[NR], rA[NR];
    nbr_dn(Nk, Nm);                     // a system's first lane: 0 (edge 0 is never written, its A stays finite)
    nbr_dn(Pk, Pm);
#pragma unroll
    for (int j = 0; j < NR; j++) {      // (:206-208) with Lambda folded into the diffusivities
        A[j] = __builtin_fma(m.hLDP, Pk[j] + Pm[j], __builtin_fma(m.hLDN, Nk[j] + Nm[j], a0));
        b[j] = __builtin_fma(m.LDP, Pk[j] - Pm[j], __builtin_fma(-m.LDN, Nk[j] - Nm[j], -bE[j]));
    }
    rcp_rows<NR>(A, rA);
    const bool act0 = act && ln != 0;
    Ek[0] = act0 ? b[0] * rA[0] : Ek[0];
#pragma unroll
    for (int j = 1; j < NR; j++) Ek[j] = act ? b[j] * rA[j] : Ek[j];
}

__global__ void __launch_bounds__(64, 1) stepper_quad_kernel(const StepArgs a)
{
    constexpr int LAY = 2;
    const int lane = threadIdx.x;
    const int ln = lane & (WS - 1);                 // lane within the system
    const int grp = lane >> 4;                      // which of the wave's systems
    const int64_t blk = blockIdx.x;
    const int c = (int)(blk % a.C);
    const int64_t s0 = NS * (blk / a.C);
    int64_t sg[NS];
    bool valid[NS];
#pragma unroll
    for (int g = 0; g < NS; g++) { valid[g] = s0 + g < a.S; sg[g] = valid[g] ? s0 + g : s0; }   // a short tail is computed again, stored once
    const int64_t s = grp == 0 ? sg[0] : grp == 1 ? sg[1] : grp == 2 ? sg[2] : sg[3];
    const CurveConst &cc = a.curve[c];

    // ---- non-dimensional material parameters (pvSimPCR.py:327-331), per lane: four samples per wave ----
    const double *xs = a.X + s * a.xld;
    const double N0 = xs[0] * cc.scales[0], P0 = xs[1] * cc.scales[1], DN = xs[2] * cc.scales[2],
                 DP = xs[3] * cc.scales[3], rate = xs[4] * cc.scales[4], sr0 = xs[5] * cc.scales[5],
                 srL = xs[6] * cc.scales[6], CN = xs[7] * cc.scales[7], CP = xs[8] * cc.scales[8],
                 tauN = xs[9] * cc.scales[9], tauP = xs[10] * cc.scales[10],
                 Lambda = xs[11] * cc.scales[11];
    const double n0p0 = N0 * P0;
    MatPar mp_ = {N0, P0, DN, DP, rate, sr0, srL, CN, CP, tauN, tauP, Lambda, n0p0,
                  ln == 0 ? 1.0 : 0.0, ln == WS - 1 ? 1.0 : 0.0};
    mp_.fast_constants();
    mp_.boundary_constants();
    const MatPar mp = mp_;
    const double mag = a.xld > 12 ? xs[12] : 0.0;
    const double TOL = a.TOL;
    const int MAX = a.MAX;

    // ---- state U^t in registers; U^{t-1..t-4} of N and P in a 4-slot LDS ring (slot = t mod 4), E's in registers ----
    constexpr int HSLOT = 2 * NR * 64;
    __shared__ __attribute__((aligned(16))) double hist[4 * HSLOT];
    double Nk[NR], Pk[NR], Ek[NR], hE[4][NR];
#pragma unroll
    for (int j = 0; j < NR; j++) {                  // pvSimPCR.py:356-362
        const double dn = a.dN[(int64_t)c * L + NR * ln + j] * cc.dx3;
        Nk[j] = N0 + dn;
        Pk[j] = P0 + dn;
        Ek[j] = 0.0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            hE[m][j] = 0.0;
            hist[m * HSLOT + (0 * NR + j) * 64 + lane] = 0.0;
            hist[m * HSLOT + (1 * NR + j) * 64 + lane] = 0.0;
        }
    }

    PlSink sink0(a, cc, c, sg[0], lane_value(mag, 0)), sink1(a, cc, c, sg[1], lane_value(mag, 16)),
           sink2(a, cc, c, sg[2], lane_value(mag, 32)), sink3(a, cc, c, sg[3], lane_value(mag, 48));
    PlSink *const sinks[NS] = {&sink0, &sink1, &sink2, &sink3};
    double rateg[NS];
    int status[NS];
    bool dead[NS];                                  // flagged non-converged (or the tail's duplicates)
    int64_t itot[NS];
#pragma unroll
    for (int g = 0; g < NS; g++) { rateg[g] = lane_value(rate, 16 * g); status[g] = 0; dead[g] = !valid[g]; itot[g] = 0; }

    int64_t pl_next = 0, pl_col = 0;                // next step with t % plT == 0 and its PL column t / plT (:276)
    for (int64_t t = 0; t <= sink0.t_last; t++) {   // tEvol, pvSimPCR.py:237
        if (dead[0] && dead[1] && dead[2] && dead[3]) break;
        double a0, a1, a2, a3, a4, a5;              // :241-250
        if (t == 0)      { a0 = 1.0; a1 = -1.0; a2 = 0.0; a3 = 0.0; a4 = 0.0; a5 = 0.0; }
        else if (t == 1) { a0 = 1.5; a1 = -2.0; a2 = 0.5; a3 = 0.0; a4 = 0.0; a5 = 0.0; }
        else if (t == 2) { a0 = 11.0 / 6; a1 = -3.0; a2 = 1.5; a3 = -1.0 / 3; a4 = 0.0; a5 = 0.0; }
        else if (t == 3) { a0 = 25.0 / 12; a1 = -4.0; a2 = 3.0; a3 = -4.0 / 3; a4 = 0.25; a5 = 0.0; }
        else             { a0 = 137.0 / 60; a1 = -5.0; a2 = 5.0; a3 = -10.0 / 3; a4 = 1.25; a5 = -0.2; }

        // PL of the state at time t, pvSimPCR.py:276-281, per-node excess first (see stepper_impl.hpp)
        double plv[NS] = {0.0, 0.0, 0.0, 0.0};
        const bool pl_step = t == pl_next;
        if (pl_step) {
            double q = __builtin_fma(Nk[0], Pk[0], -n0p0);
#pragma unroll
            for (int j = 1; j < NR; j++) q += __builtin_fma(Nk[j], Pk[j], -n0p0);
            double h[NS];
            row_sums(q, h);
#pragma unroll
            for (int g = 0; g < NS; g++) plv[g] = rateg[g] * h[g];
        }

        // BDF right-hand sides (:128-135); U^t replaces U^{t-4} in the ring
        double bN[NR], bP[NR], bE[NR], cE[NR];
        {
            const int s1 = (int)((t + 3) & 3) * HSLOT, s2 = (int)((t + 2) & 3) * HSLOT,
                      s3 = (int)((t + 1) & 3) * HSLOT, s4 = (int)(t & 3) * HSLOT;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int oN = (0 * NR + j) * 64 + lane, oP = (1 * NR + j) * 64 + lane;
                cE[j] = Ek[j];
                bN[j] = a1 * Nk[j] + a2 * hist[s1 + oN] + a3 * hist[s2 + oN] + a4 * hist[s3 + oN] + a5 * hist[s4 + oN];
                bP[j] = a1 * Pk[j] + a2 * hist[s1 + oP] + a3 * hist[s2 + oP] + a4 * hist[s3 + oP] + a5 * hist[s4 + oP];
                bE[j] = a1 * Ek[j] + a2 * hE[0][j] + a3 * hE[1][j] + a4 * hE[2][j] + a5 * hE[3][j];
                hist[s4 + oN] = Nk[j];
                hist[s4 + oP] = Pk[j];
            }
        }

        // ---------------- iterate, pvSimPCR.py:93-225, four systems ----------------
        bool done[NS];
        int its[NS];
#pragma unroll
        for (int g = 0; g < NS; g++) { done[g] = dead[g]; its[g] = MAX; }   // MAX: the value if the loop runs to exhaustion (:225)
        // One inner iteration of the four systems; FROZEN = false while all of them iterate (no selects), true
        // once one has converged in this time step (or is dead) and keeps its state.  Same source, compiled with
        // -ffp-contract=on: a system's arithmetic is bit-identical in the two (see stepper_pair_impl.hpp).
        auto iterate_once = [&](auto frozen_c, int iters) {
            constexpr bool FROZEN = decltype(frozen_c)::value;
            const bool act = FROZEN ? !(grp == 0 ? done[0] : grp == 1 ? done[1] : grp == 2 ? done[2] : done[3]) : true;
            double lo_[NR], dg[NR], up[NR], bb[NR], Ep[NR], x[NR];
            nbr_up(Ek, Ep);                         // a system's last lane reads 0: E_L = 0
            bool okN[NS], okP[NS];
            // ---- electrons (:148-175) ----
            assemble<LAY, true, NR, WS, L, true>(mp, a0, Nk, Pk, Ek, Ep, bN, lo_, dg, up, bb, ln);
            residual_below4(lo_, dg, up, bb, Nk, TOL, okN);                                        // :172
            cr_pcr_solve<double, NR, WS, false, XM>(lo_, dg, up, bb, x, lane, (double *)nullptr);  // :175
#pragma unroll
            for (int j = 0; j < NR; j++) Nk[j] = act ? x[j] : Nk[j];
            // ---- holes, with the updated electrons (:178-202) ----
            assemble<LAY, false, NR, WS, L, true>(mp, a0, Nk, Pk, Ek, Ep, bP, lo_, dg, up, bb, ln);
            // the holes' norm only matters if the electrons' passed for a system that is still iterating (:213)
            bool any = false;
#pragma unroll
            for (int g = 0; g < NS; g++) any = any || ((FROZEN ? !done[g] : true) && okN[g]);
            if (any) residual_below4(lo_, dg, up, bb, Pk, TOL, okP);                               // :200
            else okP[0] = okP[1] = okP[2] = okP[3] = false;
            cr_pcr_solve<double, NR, WS, false, XM>(lo_, dg, up, bb, x, lane, (double *)nullptr);  // :202
#pragma unroll
            for (int j = 0; j < NR; j++) Pk[j] = act ? x[j] : Pk[j];
            // ---- field on edges 1..L-1 (:205-209) ----
            update_field4(mp, a0, Nk, Pk, bE, Ek, ln, act);
#pragma unroll
            for (int g = 0; g < NS; g++)
                if (!done[g] && okN[g] && okP[g]) { done[g] = true; its[g] = iters + 1; }          // :213-216
        };
        int iters = 0;
        for (; iters < MAX && !(done[0] || done[1] || done[2] || done[3]); iters++) iterate_once(std::false_type{}, iters);
        for (; iters < MAX && !(done[0] && done[1] && done[2] && done[3]); iters++) iterate_once(std::true_type{}, iters);
        bool kill[NS], anykill = false;
        // :269-274 -- like the reference, converging only in iteration MAX itself counts as a failure
#pragma unroll
        for (int g = 0; g < NS; g++) {
            kill[g] = false;
            if (!dead[g]) { itot[g] += its[g]; if (its[g] >= MAX) { status[g] = 1 + (int)t; kill[g] = true; anykill = true; } }
        }
        if (anykill) {
            // park the flagged system at equilibrium (finite, converges trivially) for the rest of the run
            const bool mine = grp == 0 ? kill[0] : grp == 1 ? kill[1] : grp == 2 ? kill[2] : kill[3];
#pragma unroll
            for (int j = 0; j < NR; j++) {
                if (mine) {
                    Nk[j] = N0; Pk[j] = P0; Ek[j] = 0.0; cE[j] = 0.0;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        hE[m][j] = 0.0;
                        hist[m * HSLOT + (0 * NR + j) * 64 + lane] = N0;
                        hist[m * HSLOT + (1 * NR + j) * 64 + lane] = P0;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NS; g++) dead[g] = dead[g] || kill[g];
        }

        if (pl_step) {
#pragma unroll
            for (int g = 0; g < NS; g++)
                if (!dead[g]) { if (sinks[g]->interp) sinks[g]->emit(pl_col, plv[g]); else sinks[g]->push(pl_col, plv[g]); }
            pl_next += a.plT;
            pl_col++;
        }

#pragma unroll
        for (int j = 0; j < NR; j++) {              // shift the field history by one level
#pragma unroll
            for (int m = 3; m >= 1; m--) hE[m][j] = hE[m - 1][j];
            hE[0][j] = cE[j];
        }
    }

#pragma unroll
    for (int g = 0; g < NS; g++) {
        if (!sinks[g]->interp && valid[g]) {        // columns parked since the last full batch
            const int64_t done_ = status[g] ? (int64_t)(status[g] - 1) : sinks[g]->t_last + 1;
            sinks[g]->flush_batch((int)((done_ + a.plT - 1) / a.plT - sinks[g]->base));
        }
        if (valid[g]) sinks[g]->finish(status[g], itot[g]);
    }
}

}  // namespace quad

inline hipError_t launch_stepper_quad_t(const StepArgs &a, hipStream_t stream)
{
    if (a.L != quad::L || a.n_snap > 0 || a.resN != nullptr) return hipErrorInvalidValue;
    const int64_t nblk = ((a.S + quad::NS - 1) / quad::NS) * a.C;
    if (nblk <= 0) return hipSuccess;
    hipLaunchKernelGGL(quad::stepper_quad_kernel, dim3((unsigned)nblk), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace trpl
