// FAST time-stepper for L = 128 with TWO SYSTEMS PER WAVEFRONT (two curves of one sample, or adjacent samples of one curve).
//
// Why: the tridiagonal solve is the largest part of an inner iteration, and its in-lane
// cyclic-reduction levels are work-efficient (O(rows)) while the cross-lane PCR levels are not
// (O(rows log rows)).  With one L = 128 system per wave a lane holds 2 adjacent rows: 1 CR level +
// 5 PCR levels + the pair step.  Here a lane holds 4 adjacent rows of ONE of two systems (lanes 0-31:
// system A, lanes 32-63: system B): 2 CR levels + 4 PCR levels + the pair step for twice the nodes --
// measured +29 % node throughput for this shape (the L = 256 kernel), at 2 waves per SIMD.
//
// The two systems are numerically independent: every value that crosses the seam between lanes 31 | 32
// (or wraps 63 | 0) inside an iteration is multiplied by a coefficient that is exactly zero (first-row
// sub-diagonal, last-row super-diagonal and their PCR/CR descendants), and the one that is not (edge 0 of the
// field update) is always replaced by 0.  A NaN / Inf would still cross (0 * NaN): the SEAM flavour of the
// iteration clears every crossing value first; the kernel runs the flavour without those selects and repeats a
// time step in the SEAM flavour when a system is flagged in it or leaves it with a non-finite state ("optimistic
// seam", see the time loop) -- the results are those of a kernel that always clears.  Each system is computed by the
// same instruction sequence whichever half it sits in: results do not depend on the pairing.
//
// WHICH two systems share a wavefront does not change their bits (tested), only how many iterations the wave runs:
// each time step costs max(itA, itB).  Default pairing: the two curves of ONE sample that the host table names
// (StepArgs::pair_*: curves of equal thickness and observation count, neighbouring excitation powers) -- same material
// parameters, similar stiffness: 2.1 % of the wave-iterations lost to the partner at T = 8000 against 4.2 % for adjacent
// samples of one curve (oracle traces, 1024 samples x 3 curves).  Without a table (one curve,
// off-grid observations): adjacent samples of one curve.
//
// Convergence is per system (pvSimPCR.py:213-216): a system that has converged in this time step is
// frozen (its lanes keep their state) while its partner iterates on; a system that hits MAX is
// flagged (:269) and parked -- replaced by a benign system at its equilibrium (see park()) -- for the rest of the run.
// Reference for the arithmetic: see stepper_impl.hpp (assemble / PlSink are shared with it).
#pragma once
#include "stepper_impl.hpp"

namespace trpl {
namespace pair {

constexpr int L = 128;      // nodes per system
constexpr int NR = 4;       // adjacent rows per lane
constexpr int WS = 32;      // lanes per system

__device__ __forceinline__ double lane_value(double v, int l)      // lane l's value, wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

// sums of v over lanes 0-31 (lo) and 32-63 (hi), wave-uniform; same association in both halves
__device__ __forceinline__ void half_sums(double v, double &lo, double &hi)
{
    v = dpp_add<0x111, 0xF>(v);          // row_shr:1
    v = dpp_add<0x112, 0xF>(v);          // row_shr:2
    v = dpp_add<0x114, 0xF>(v);          // row_shr:4
    v = dpp_add<0x118, 0xF>(v);          // row_shr:8   -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xA>(v);          // row_bcast:15 into rows 1,3 -> lanes 31 and 63 hold the half sums
    lo = lane_value(v, 31);
    hi = lane_value(v, 63);
}

// norm2 (pvSimPCR.py:14-40) of both systems: ok = sum|A c - b| < TOL * sum|b|, i.e. sum(|r| - TOL |b|) < 0, in two parts.
//   residual_terms2: every lane's term q and the two lane votes (q < 0, q >= 0);
//   residual_verdict2: where all 32 lanes of a system agree on the sign of their term, that is the sign of its sum
//     (wave_sum_negative, crosslane.hpp); only otherwise the half-wave reduction.
// The verdict follows the terms directly; taking it only where it is first needed (the electrons' after their solve and the
// holes' assembly, the holes' at the end of the iteration, so that the iteration stays one basic block up to there) was
// measured 0.3 % slower (round 4).
// needA / needB: the systems whose verdict is used (a frozen or parked system's is not, and must not force a reduction).
struct Terms2 {
    double q;                                   // this lane's sum over its rows of |r| - TOL |b|
    unsigned long long neg, nonneg;             // lanes with q < 0 / q >= 0 (a NaN is in neither)
};
__device__ __forceinline__ Terms2 no_terms2() { return {0.0, 0ull, ~0ull}; }      // "not below", decided without a reduction

template <bool ISO>
__device__ __forceinline__ Terms2 residual_terms2(const double (&l)[NR], const double (&dg)[NR], const double (&u)[NR],
                                                  const double (&b)[NR], const double (&c)[NR], double TOL, int lane)
{
    double cm[NR], cp[NR];
    nbrB_dn<double, NR, 1>(c, cm, lane);
    nbrB_up<double, NR, 1>(c, cp, lane);
    cm[0] = seam_first<ISO, WS>(cm[0], lane);          // times l = 0 on a system's first row
    cp[NR - 1] = seam_last<ISO, WS>(cp[NR - 1], lane); // times u = 0 on its last row
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        // (A c - b)_j as three nested fmas (one operation less than sum-then-subtract; rounding-level difference)
        const double r = fabs(__builtin_fma(l[j], cm[j], __builtin_fma(dg[j], c[j], __builtin_fma(u[j], cp[j], -b[j]))));
        const double qj = __builtin_fma(-TOL, fabs(b[j]), r);
        q = j == 0 ? qj : q + qj;
    }
    Terms2 t = {q, 0ull, 0ull};
    if constexpr (TRPL_NORM_VOTE != 0) {
        t.neg = __builtin_amdgcn_ballot_w64(q < 0.0);
        t.nonneg = __builtin_amdgcn_ballot_w64(q >= 0.0);
    }
    return t;
}

__device__ __forceinline__ void residual_verdict2(const Terms2 &t, bool needA, bool needB, bool &okA, bool &okB,
                                                  int *reductions = nullptr)
{
    if constexpr (TRPL_NORM_VOTE != 0) {
        // the two common outcomes with one 64-bit scalar compare each: every lane of the wave >= 0 (the first iteration
        // of a time step: both systems far from converged) / every lane < 0 (its last one)
        if (t.nonneg == ~0ull) { okA = okB = false; return; }
        if (t.neg == ~0ull) { okA = okB = true; return; }
        // the upper halves behind an empty asm: the optimiser would otherwise fold the `>> 32` test into a 64-bit unsigned
        // compare, which only the VALU has
        const unsigned negA = (unsigned)t.neg, nnA = (unsigned)t.nonneg;
        unsigned negB = (unsigned)(t.neg >> 32), nnB = (unsigned)(t.nonneg >> 32);
        asm volatile("" : "+s"(negB), "+s"(nnB));
        const bool allnegA = negA == ~0u, allnegB = negB == ~0u;
        const bool decidedA = !needA || allnegA || nnA == ~0u, decidedB = !needB || allnegB || nnB == ~0u;
        if (decidedA && decidedB) {
            okA = allnegA;
            okB = allnegB;
            return;
        }
    }
    if (reductions) (*reductions)++;
    double sA, sB;
    half_sums(t.q, sA, sB);
    okA = sA < 0.0;
    okB = sB < 0.0;
}

// field update on edges 1..L-1 (pvSimPCR.py:205-209) of the systems whose lanes have act set
template <bool ISO>
__device__ __forceinline__ void update_field2(const MatPar &m, double a0, const double (&Nk)[NR], const double (&Pk)[NR],
                                              const double (&bE)[NR], double (&Ek)[NR], int lane, bool act)
{
    double Nm[NR], Pm[NR], A[NR], b[NR], rA[NR];
    nbrB_dn<double, NR, 1>(Nk, Nm, lane);
    nbrB_dn<double, NR, 1>(Pk, Pm, lane);
    if constexpr (ISO) {     // edge 0 is never written, but its A enters a paired reciprocal: a full zero here
        const bool first = (lane & (WS - 1)) == 0;
        Nm[0] = first ? 0.0 : Nm[0];
        Pm[0] = first ? 0.0 : Pm[0];
    }
#pragma unroll
    for (int j = 0; j < NR; j++) {     // (:206-208) with Lambda folded into the diffusivities: 2 fma + 2 adds each
        A[j] = __builtin_fma(m.hLDP, Pk[j] + Pm[j], __builtin_fma(m.hLDN, Nk[j] + Nm[j], a0));
        b[j] = __builtin_fma(m.LDP, Pk[j] - Pm[j], __builtin_fma(-m.LDN, Nk[j] - Nm[j], -bE[j]));
    }
    rcp_rows<NR>(A, rA);
    const bool act0 = act && (lane & (WS - 1)) != 0;
    Ek[0] = act0 ? b[0] * rA[0] : Ek[0];
#pragma unroll
    for (int j = 1; j < NR; j++) Ek[j] = act ? b[j] * rA[j] : Ek[j];
}

// OPT: the optimistic seam (see the time loop); false = the selects in every iteration, the form rounds 1-3 shipped.  Both
// are in the library: the second as the reference of the differential tests (TRPL_FLAG_PAIR_ALWAYS_SEAM selects it per call).
template <bool ISO, bool SNAP = false, bool OPT = (TRPL_PAIR_OPTIMISTIC != 0)>
__global__ void __launch_bounds__(64, 2) stepper_pair_kernel(const StepArgs a)
{
    constexpr int LAY = 2;
    const int lane = threadIdx.x;
    const int ln = lane & (WS - 1);                 // lane within the system
    const bool hi = lane >= WS;
    const int64_t blk = blockIdx.x;
    int cA = (int)(blk % a.C), cB = cA;
    int64_t sA = 2 * (blk / a.C), sB = sA + 1;
    if (a.pair_n > 0) {                             // the host's table: block k of the period that covers samples 2p, 2p + 1
        const int k = cA;
        cA = a.pair_cA[k]; cB = a.pair_cB[k];
        sB = sA + a.pair_oB[k];
        sA = sA + a.pair_oA[k];
        if (sA >= a.S) return;                      // an odd batch: the second sample of the last period does not exist
    }
    const bool validB = sB < a.S;
    if (!validB) { sB = sA; cB = cA; }              // an odd tail is computed twice and stored once
    const int64_t s = hi ? sB : sA;
    const int c = hi ? cB : cA;
    const CurveConst &cc = a.curve[cA];             // paired curves share thickness, grid and window: one set of scales

    // ---- non-dimensional material parameters (pvSimPCR.py:327-331), per lane: two samples per wave ----
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
    MatPar mp = mp_;                               // constant but for park(): a flagged system's lanes get benign parameters
    const double mag = a.xld > 12 ? xs[12] : 0.0;
    const double TOL = a.TOL;
    const int MAX = a.MAX;

    // ---- state U^t in registers; U^{t-1..t-4} of N and P in a 4-slot LDS ring (slot = t mod 4), E's in registers ----
    // ring layout [slot][row][lane]{N, P}: a lane's N and P of one row and level are one ds_read_b128 / ds_write_b128
    // (20 DS instructions per time step instead of 40); the field history hE[m] = E^{t-1-m} stays in registers
    constexpr int HSLOT = 2 * NR * 64;
    __shared__ __attribute__((aligned(16))) double lds[4 * HSLOT];
    double2 *hist2 = reinterpret_cast<double2 *>(lds);            // hist2[(slot * NR + row) * 64 + lane] = {N, P}
    double *xch = nullptr;                          // the solver's exchanges are DPP moves and ds_swizzle rotates: no buffer
    double Nk[NR], Pk[NR], Ek[NR], hE[4][NR];
#pragma unroll
    for (int j = 0; j < NR; j++) {                  // pvSimPCR.py:356-362
        // (a resume takes its state from the checkpoint; dN is not read -- it may be NULL there)
        const double raw = (SNAP && a.resN != nullptr) ? 0.0 : a.dN[(int64_t)c * L + NR * ln + j];
        const double dn = raw * cc.dx3;
        Nk[j] = N0 + dn;
        Pk[j] = P0 + dn;
        Ek[j] = 0.0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            hE[m][j] = 0.0;
            hist2[(m * NR + j) * 64 + lane] = make_double2(0.0, 0.0);
        }
    }

    PlSink sinkA(a, cc, cA, sA, lane_value(mag, 0));
    PlSink sinkB(a, cc, cB, sB, lane_value(mag, WS));      // same n_obs and plnorm as cA's by construction of the table
    const double rateA = lane_value(rate, 0), rateB = lane_value(rate, WS);
    sinkA.set_floor(rateA, lane_value(n0p0, 0), L);
    sinkB.set_floor(rateB, lane_value(n0p0, WS), L);
    int statusA = 0, statusB = 0;
    bool deadA = false, deadB = !validB;            // dead: flagged non-converged (or the odd tail's duplicate)
    int64_t itotA = 0, itotB = 0;
#if TRPL_VOTE_STATS
    int nredN = 0, nredP = 0;                      // measurement build: reductions the votes did not save
#define TRPL_STAT(x) &x
#else
#define TRPL_STAT(x) nullptr
#endif
    SnapSink snap(a, cc);
    // Park this lane's system for the rest of the run: a flagged system (or one found flagged in a checkpoint) is REPLACED
    // by a benign one at its equilibrium -- unit densities, diffusivities and lifetimes, no recombination coefficients, no
    // field coupling.  Its lanes go on executing every iteration (results discarded), and with the sample's own parameters
    // (a NaN lifetime, an infinite rate) they would form non-finite coefficients at every later step, which the optimistic
    // seam below must not meet; with these they form finite ones, converge trivially and cross the seam as finite values
    // times exact zeros.
    auto park = [&](bool mine) {
        if (mine) {
            MatPar b_ = {1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 1.0,
                         ln == 0 ? 1.0 : 0.0, ln == WS - 1 ? 1.0 : 0.0};
            b_.fast_constants();
            b_.boundary_constants();
            mp = b_;
        }
#pragma unroll
        for (int j = 0; j < NR; j++) {
            if (mine) {
                Nk[j] = 1.0; Pk[j] = 1.0; Ek[j] = 0.0;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    hE[m][j] = 0.0;
                    hist2[(m * NR + j) * 64 + lane] = make_double2(1.0, 1.0);
                }
            }
        }
    };

    int32_t t_begin = 0;
    if constexpr (SNAP) {
        if (a.resN != nullptr) {                    // resume at t0 >= 4 (see stepper_impl.hpp)
            t_begin = (int32_t)a.t0;
            const int64_t r5 = (hi ? sinkB.orow : sinkA.orow) * 5;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int i = NR * ln + j;
                Nk[j] = a.resN[(r5 + 4) * L + i]; Pk[j] = a.resP[(r5 + 4) * L + i]; Ek[j] = a.resE[(r5 + 4) * (L + 1) + i];
#pragma unroll
                for (int m = 0; m < 4; m++) {       // level t0-1-m: ring slot (t0-1-m) mod 4 for N and P, register level m for E
                    const int slot = (int)((a.t0 - 1 - m) & 3);
                    const double e_ = a.resE[(r5 + 3 - m) * (L + 1) + i];
                    hE[m][j] = e_;
                    hist2[(slot * NR + j) * 64 + lane] = make_double2(a.resN[(r5 + 3 - m) * L + i], a.resP[(r5 + 3 - m) * L + i]);
                }
            }
            // a system that was flagged before the checkpoint (its newest level carries the status word, see
            // nan_status): it keeps that status, takes no step and is parked like a system flagged in this run
            statusA = status_of_checkpoint(a.resN[(sinkA.orow * 5 + 4) * L], a.t0);
            statusB = validB ? status_of_checkpoint(a.resN[(sinkB.orow * 5 + 4) * L], a.t0) : 0;
            deadA = statusA != 0;
            deadB = deadB || statusB != 0;
            if (statusA || statusB) park(hi ? statusB != 0 : statusA != 0);
        }
    }
    int32_t pl_next = 0, pl_col = 0;                // next step with t % plT == 0 and its PL column t / plT (:276)
    if constexpr (SNAP) {
        if (t_begin > 0) { pl_col = (t_begin + a.plT - 1) / a.plT; pl_next = pl_col * a.plT; sinkA.base = sinkB.base = pl_col; }
    }
    const int32_t row_cap = bdf_row_cap(a.flags);    // TRPL_FLAG_BDF_ORDER: highest row of the BDF table this run uses
    for (int32_t t = t_begin; t <= sinkA.t_last; t++) {   // tEvol, pvSimPCR.py:237
        if (deadA && deadB) break;
        if constexpr (SNAP) {                       // the state at time t, before it is stepped (:283-288)
            if (snap.due(t))
                snap.template take<NR, L>(Nk, Pk, Ek, hi ? sinkB.orow : sinkA.orow, hi ? !deadB : !deadA,
                                          [&](int j) { return NR * ln + j; });
        }
        double a0, a1, a2, a3, a4, a5;              // :241-250
        bdf_row<double>(t < row_cap ? t : row_cap, a0, a1, a2, a3, a4, a5);

        // PL of the state at time t, pvSimPCR.py:276-281, per-node excess first (see stepper_impl.hpp)
        double plA = 0.0, plB = 0.0;
        const bool pl_step = t == pl_next;
        if (pl_step) {
            double q = __builtin_fma(Nk[0], Pk[0], -n0p0);
#pragma unroll
            for (int j = 1; j < NR; j++) q += __builtin_fma(Nk[j], Pk[j], -n0p0);
            double hA, hB;
            half_sums(q, hA, hB);
            plA = rateA * hA;
            plB = rateB * hB;
        }

        // BDF right-hand sides (:128-135); U^t replaces U^{t-4} in the ring
        double bN[NR], bP[NR], bE[NR];
        {
            const int s1 = (int)((t + 3) & 3) * NR, s2 = (int)((t + 2) & 3) * NR,
                      s3 = (int)((t + 1) & 3) * NR, s4 = (int)(t & 3) * NR;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const double2 h1 = hist2[(s1 + j) * 64 + lane], h2 = hist2[(s2 + j) * 64 + lane],
                              h3 = hist2[(s3 + j) * 64 + lane], h4 = hist2[(s4 + j) * 64 + lane];
                bN[j] = a1 * Nk[j] + a2 * h1.x + a3 * h2.x + a4 * h3.x + a5 * h4.x;
                bP[j] = a1 * Pk[j] + a2 * h1.y + a3 * h2.y + a4 * h3.y + a5 * h4.y;
                hist2[(s4 + j) * 64 + lane] = make_double2(Nk[j], Pk[j]);
            }
            // the field's history stays in registers and is shifted here, before the iterations (12 v_mov_b64; no copy of
            // E^t is carried through them).  A register ring without the shift was measured and spills: DESIGN.md section 8
#pragma unroll
            for (int j = 0; j < NR; j++) {
                bE[j] = a1 * Ek[j] + a2 * hE[0][j] + a3 * hE[1][j] + a4 * hE[2][j] + a5 * hE[3][j];
#pragma unroll
                for (int m = 3; m >= 1; m--) hE[m][j] = hE[m - 1][j];
                hE[0][j] = Ek[j];
            }
        }

        // ---------------- iterate, pvSimPCR.py:93-225, both systems ----------------
        bool doneA = deadA, doneB = deadB;
        int itA = MAX, itB = MAX;                   // value if the loop runs to exhaustion (:225)
        // One inner iteration of both systems.  FROZEN = false is the common case -- both systems still
        // iterating: every lane takes its solve's result, no selects (25 fewer instructions); FROZEN = true keeps
        // the state of a system that has already converged in this time step (or is dead) while its partner
        // iterates on.  Both flavours are the same source: this translation unit is compiled with
        // -ffp-contract=on (fusion decided by the syntax of each expression, not by the optimiser's view of the
        // surrounding code), so a system's arithmetic is bit-identical in the two -- its result must not depend
        // on when its partner converges.
        auto iterate_once = [&](auto frozen_c, auto seam_c, int iters) {
            constexpr bool FROZEN = decltype(frozen_c)::value;
            constexpr bool SEAM = decltype(seam_c)::value;     // void what crosses the seam inside the iteration
            const bool act = FROZEN ? (hi ? !doneB : !doneA) : true;   // lanes of a system that is still iterating
            double lo_[NR], dg[NR], up[NR], bb[NR], Ep[NR], x[NR];
            nbrB_up<double, NR, 1>(Ek, Ep, lane);   // a system's last lane reads the partner's E_0 = 0: E_L = 0
            bool okNA, okNB, okPA, okPB;
            const bool needNA = FROZEN ? !doneA : true, needNB = FROZEN ? !doneB : true;
            // ---- electrons (:148-175) ----
            assemble<LAY, true, NR, WS, L, true>(mp, a0, Nk, Pk, Ek, Ep, bN, lo_, dg, up, bb, ln);
            const Terms2 tN = residual_terms2<SEAM>(lo_, dg, up, bb, Nk, TOL, lane);                // :172
            residual_verdict2(tN, needNA, needNB, okNA, okNB, TRPL_STAT(nredN));
            cr_pcr_solve<double, NR, WS, SEAM, true>(lo_, dg, up, bb, x, lane, xch);                    // :175
#pragma unroll
            for (int j = 0; j < NR; j++) Nk[j] = act ? x[j] : Nk[j];
            // ---- holes, with the updated electrons (:178-202) ----
            assemble<LAY, false, NR, WS, L, true>(mp, a0, Nk, Pk, Ek, Ep, bP, lo_, dg, up, bb, ln);
            // the holes' norm only matters if the electrons' passed for a system that is still iterating (:213):
            // on the first iteration of a time step it practically never has (a wave-uniform branch)
            const bool needPA = needNA && okNA, needPB = needNB && okNB;
            Terms2 tP = no_terms2();
            if (needPA || needPB) {
                tP = residual_terms2<SEAM>(lo_, dg, up, bb, Pk, TOL, lane);                         // :200
                residual_verdict2(tP, needPA, needPB, okPA, okPB, TRPL_STAT(nredP));
            } else {
                okPA = okPB = false;
            }
            cr_pcr_solve<double, NR, WS, SEAM, true>(lo_, dg, up, bb, x, lane, xch);                    // :202
#pragma unroll
            for (int j = 0; j < NR; j++) Pk[j] = act ? x[j] : Pk[j];
            // ---- field on edges 1..L-1 (:205-209) ----
            update_field2<ISO>(mp, a0, Nk, Pk, bE, Ek, lane, act);
            if (!doneA && okNA && okPA) { doneA = true; itA = iters + 1; }                         // :213-216
            if (!doneB && okNB && okPB) { doneB = true; itB = iters + 1; }
        };
        // within a time step a system only ever goes from iterating to done: first the iterations with both
        // systems active, then those with one of them frozen (two loops, not a branch inside one: the register
        // allocator handles them separately; a branch inside the loop spilled 89 VGPRs)
        auto iterate_step = [&](auto seam_c) {
            int iters = 0;
            for (; iters < MAX && !(doneA || doneB); iters++) iterate_once(std::false_type{}, seam_c, iters);
            for (; iters < MAX && !(doneA && doneB); iters++) iterate_once(std::true_type{}, seam_c, iters);
        };
        if constexpr (ISO && OPT) {
            // OPTIMISTIC SEAM.  Everything that crosses the seam inside an iteration meets an exact-zero coefficient
            // (update_field2's edge 0 apart, which is always voided), so while both systems are finite the voiding
            // selects change no bit -- they only keep a NaN / Inf of one system out of the other (0 * NaN).  So: iterate
            // WITHOUT the selects (33 of the iteration's 46 v_cndmask), look at the outcome, and if a non-finite value may
            // have been on the seam put both systems back to U^t -- N and P from the ring slot written above, E from
            // hE[0] -- and repeat the step WITH them.  The repeat is the arithmetic of the always-voiding kernel, hence so
            // is every result; the common step pays nothing.  What can put a non-finite value on the seam: a live system
            // whose state turns non-finite in this step -- it either never converges (flagged at the step's end) or passes
            // an iteration's test and turns non-finite in that iteration's solve (its partner, polluted in the same
            // solve, is then marked converged as well): both show below.  A parked system (flagged earlier) is a benign
            // finite one (park()), the odd tail's duplicate a copy of its partner: neither can.
            iterate_step(std::false_type{});
            // repeat the step if a live system is flagged in it -- or if a system's new state is not finite: the test of an
            // iteration precedes its solve, so a system can pass it and turn non-finite in that same, last iteration
            // (it is then flagged in the NEXT step), and its partner, polluted in that solve, is marked converged too.
            // The witness costs one compare: the field update forms the reciprocals of a lane's four A_j (each holding that
            // row's N and P, times Lambda D -- a zero factor keeps a NaN) from ONE v_rcp_f64 of their product (rcp_rows), so a
            // non-finite N or P anywhere in the lane makes every new E of the lane non-finite, the last row's included.
            // (measured alternatives, round 4: the reduced excess sums, every lane's own term of them, the sum of the lane's
            // four new P -- all bit-identical on the hostile batches, 0.1 - 0.3 % slower)
            static_assert(NR % 4 == 0, "the witness relies on rcp_rows taking ONE reciprocal of the product of a lane's rows (pcr.hpp)");
            const double wit = Ek[NR - 1];
            const bool finite2 = TRPL_PAIR_WITNESS == 0 || __builtin_amdgcn_ballot_w64(__builtin_isfinite(wit)) == ~0ull;
            if ((!deadA && itA >= MAX) || (!deadB && itB >= MAX) || !finite2) {
                const int s4 = (int)(t & 3) * NR;
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const double2 h = hist2[(s4 + j) * 64 + lane];
                    Nk[j] = h.x; Pk[j] = h.y; Ek[j] = hE[0][j];
                }
                doneA = deadA; doneB = deadB; itA = itB = MAX;
                iterate_step(std::true_type{});
            }
        } else {
            iterate_step(std::integral_constant<bool, ISO>{});
        }
        bool killA = false, killB = false;
        // :269-274 -- like the reference, converging only in iteration MAX itself counts as a failure
        if (!deadA) { itotA += itA; if (itA >= MAX) { statusA = 1 + (int)t; killA = true; } }
        if (!deadB) { itotB += itB; if (itB >= MAX) { statusB = 1 + (int)t; killB = true; } }
        if (killA || killB) {
            // park the flagged system at equilibrium (finite, converges trivially) for the rest of the run
            park(hi ? killB : killA);
            deadA = deadA || killA;
            deadB = deadB || killB;
        }

        if (pl_step) {
            if (!deadA) sinkA.push(pl_col, plA);
            if (!deadB) sinkB.push(pl_col, plB);
            pl_next += a.plT;
            pl_col++;
        }

    }

    {                                               // columns parked since the last full batch
        const int64_t doneA_ = statusA ? (int64_t)(statusA - 1) : sinkA.t_last + 1;
        const int64_t doneB_ = statusB ? (int64_t)(statusB - 1) : sinkB.t_last + 1;
        sinkA.flush_batch((int)((doneA_ + a.plT - 1) / a.plT - sinkA.base));
        if (validB) sinkB.flush_batch((int)((doneB_ + a.plT - 1) / a.plT - sinkB.base));
    }
    if constexpr (SNAP) {
        if (statusA && !hi) snap.template fail_fill<L>(sinkA.orow, statusA, ln, WS);
        if (statusB && hi && validB) snap.template fail_fill<L>(sinkB.orow, statusB, ln, WS);
    }
#if TRPL_VOTE_STATS
    // iteration totals stay below 2^24 in the measured windows: reductions of the N tests << 24, of the P tests << 44
    itotA += ((int64_t)nredN << 24) + ((int64_t)nredP << 44);
#endif
    sinkA.finish(statusA, itotA);
    if (validB) sinkB.finish(statusB, itotB);
#undef TRPL_STAT
}

}  // namespace pair

template <bool ISO>
hipError_t launch_stepper_pair_t(const StepArgs &a, hipStream_t stream)
{
    if (a.L != pair::L) return hipErrorInvalidValue;
    const int64_t nblk = ((a.S + 1) / 2) * a.C;      // with and without a pairing table
    if (nblk <= 0) return hipSuccess;
    // TRPL_FLAG_PAIR_ALWAYS_SEAM (tests, measurements): the always-isolating kernel instead of the optimistic one
    const bool always_seam = (a.flags & kFlagPairAlwaysSeam) != 0;
    const bool snap = a.n_snap > 0 || a.resN != nullptr;
    const dim3 grid((unsigned)nblk), block(64);
    if (always_seam) {
        if (snap) hipLaunchKernelGGL((pair::stepper_pair_kernel<ISO, true, false>), grid, block, 0, stream, a);
        else      hipLaunchKernelGGL((pair::stepper_pair_kernel<ISO, false, false>), grid, block, 0, stream, a);
    } else {
        if (snap) hipLaunchKernelGGL((pair::stepper_pair_kernel<ISO, true>), grid, block, 0, stream, a);
        else      hipLaunchKernelGGL((pair::stepper_pair_kernel<ISO, false>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace trpl
