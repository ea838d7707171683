// Tridiagonal solvers by parallel cyclic reduction, one wavefront per system (reference:
// pcreduce, pvSimPCR.py:42-81).  Three flavours (a fourth, pure PCR in the interleaved layout, lost to cr_pcr_solve in
// round 1 and left the tree in round 5):
//   pcr_solve       STRICT, blocked layout: the reference's operation order, IEEE divides, guards
//   pcr_solve_fast  FAST, blocked layout (L < 128): normalised rows, Newton-refined reciprocals
//   cr_pcr_solve    FAST, L >= 128, fp64 or fp32: log2(NR) in-lane cyclic-reduction levels, then PCR on
//                   64 unknowns (one row per lane), back-substitution -- the production path
#pragma once
#include "crosslane.hpp"

namespace trpl {

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


// ------------------------------------------------------------------------------------------
// FAST-mode building blocks (STRICT=false).  Same mathematics, cheaper arithmetic:
//   * 1/x by v_rcp_f64 + two Newton steps (~1 ulp) instead of the IEEE divide expansion;
//   * PCR on normalised rows: each row publishes (ld, ud, B)/d, so a neighbour fetch moves 3
//     values instead of 4 and one reciprocal per row per level replaces two divides; boundary
//     rows need no guards because their ld / ud are exact zeros (pvSimPCR.py:59,:65 guard the
//     same rows);
//   * unit shifts (i +- 1) by DPP wave rotates on the VALU instead of ds_bpermute through LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ float rcp_nr(float d) { return 1.0f / d; }

template <typename T, int NR, int W, int L, int RF>
__device__ __forceinline__ void pcr_level_fast(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], int ln)
{
    T nl[NR], nu[NR], nB[NR];
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const T r = rcp_nr(d[j]);
        nl[j] = ld[j] * r; nu[j] = ud[j] * r; nB[j] = B[j] * r;
    }
    T l_m[NR], u_m[NR], B_m[NR], l_p[NR], u_p[NR], B_p[NR];
    nb_dn<T, NR, W, RF>(nl, l_m, ln);
    nb_dn<T, NR, W, RF>(nu, u_m, ln);
    nb_dn<T, NR, W, RF>(nB, B_m, ln);
    nb_up<T, NR, W, RF>(nl, l_p, ln);
    nb_up<T, NR, W, RF>(nu, u_p, ln);
    nb_up<T, NR, W, RF>(nB, B_p, ln);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        // rows i < RF have ld == 0 and rows i >= L-RF have ud == 0 (exactly), so the wrapped
        // neighbour values they fetched drop out
        d[j] = d[j] - ld[j] * u_m[j] - ud[j] * l_p[j];
        B[j] = B[j] - ld[j] * B_m[j] - ud[j] * B_p[j];
        ld[j] = -ld[j] * l_m[j];
        ud[j] = -ud[j] * u_p[j];
    }
}

template <typename T, int NR, int W, int L, int RF>
__device__ __forceinline__ void pcr_levels_fast(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], int ln)
{
    if constexpr (L > 2 * RF) {
        pcr_level_fast<T, NR, W, L, RF>(ld, d, ud, B, ln);
        pcr_levels_fast<T, NR, W, L, RF * 2>(ld, d, ud, B, ln);
    }
}

template <typename T, int NR, int W, int L>
__device__ __forceinline__ void pcr_solve_fast(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], T (&x)[NR],
                                               int ln)
{
    pcr_levels_fast<T, NR, W, L, 1>(ld, d, ud, B, ln);
    if constexpr (NR >= 2) {
        constexpr int H = NR / 2;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const T r1 = rcp_nr(d[j + H]);
            const T k = ud[j] * r1;
            const T den = d[j] - ld[j + H] * k;
            const T num = B[j] - B[j + H] * k;
            x[j] = num * rcp_nr(den);
            x[j + H] = (B[j + H] - ld[j + H] * x[j]) * r1;
        }
    } else {
        constexpr int H = W / 2;
        const bool low = (ln & H) == 0;
        const T ud_o = __shfl_xor(ud[0], H, 64), d_o = __shfl_xor(d[0], H, 64), B_o = __shfl_xor(B[0], H, 64),
                ld_o = __shfl_xor(ld[0], H, 64);
        const T l_ud = low ? ud[0] : ud_o, l_d = low ? d[0] : d_o, l_B = low ? B[0] : B_o;
        const T h_d = low ? d_o : d[0], h_B = low ? B_o : B[0], h_ld = low ? ld_o : ld[0];
        const T r1 = rcp_nr(h_d);
        const T k = l_ud * r1;
        const T xl = (l_B - h_B * k) * rcp_nr(l_d - h_ld * k);
        const T xh = (h_B - h_ld * xl) * r1;
        x[0] = low ? xl : xh;
    }
}

// mode dispatch
template <bool STRICT, typename T, int NR, int W, int L>
__device__ __forceinline__ void tridiag_solve(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], T (&x)[NR], int ln)
{
    if constexpr (STRICT) pcr_solve<T, NR, W, L>(ld, d, ud, B, x, ln);
    else                  pcr_solve_fast<T, NR, W, L>(ld, d, ud, B, x, ln);
}

__device__ __forceinline__ double rcp_nr1(double d)      // one Newton step: 2e-15 relative (measured)
{
    const double r = __builtin_amdgcn_rcp(d);     // (a cvt + v_rcp_f32 + cvt seed measured 4 % slower)
    return __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
}
__device__ __forceinline__ float rcp_nr1(float d) { return 1.0f / d; }
// the reciprocal at the accuracy a build-time knob selects (crosslane.hpp)
// third-order step: r0 (1 + e + e^2), e = 1 - d r0.  One Newton step leaves r0 (1 - e^2) -- never above 1 / d: a BIAS of up
// to 2e-15, and in the solver's normalised rows a bias is not a rounding error: every eliminated unknown is substituted
// (1 - e^2) times too small, a spurious sink of e^2 D dt/dx^2 per elimination that adds up over the steps instead of
// averaging out (measured: it alone moved the 311 nm films' state by 1e-10 .. 1e-8 over 8000 steps).
// The cubic step costs one fma more and leaves e^3 ~ 1e-22: the result is the rounding of its last fma.
__device__ __forceinline__ double rcp_cubic(double d)
{
    const double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
template <int STEPS>
__device__ __forceinline__ double rcp_steps(double d)
{
    if constexpr (STEPS == 0) return 1.0 / d;
    else if constexpr (STEPS == 2) return rcp_nr(d);
    else if constexpr (STEPS == 3) return rcp_cubic(d);
    else return rcp_nr1(d);
}
__device__ __forceinline__ double rcp_row(double d) { return rcp_steps<TRPL_RCP_ROWS_STEPS>(d); }

// Reciprocals of all NR values of a lane.  v_rcp_f64 costs ~3.5 fp64 multiplies, so values are
// paired: r = 1/(a*b), 1/a = b*r, 1/b = a*r (one reciprocal + 3 multiplies instead of two
// reciprocals).  The operands here are O(1e-4 .. 1e4), far from over/underflow of the product.
template <int NR>
__device__ __forceinline__ void rcp_rows(const double (&d)[NR], double (&r)[NR])
{
    if constexpr (NR % 4 == 0) {
        // four values share ONE reciprocal: r = 1/(ab cd), 1/(ab) = cd r, 1/(cd) = ab r, then as for pairs
        // (9 multiplies + 1 reciprocal instead of 6 + 2; the operands are O(1e-4 .. 1e4))
#pragma unroll
        for (int j = 0; j < NR; j += 4) {
            const double ab = d[j] * d[j + 1], cd = d[j + 2] * d[j + 3];
            const double rq = rcp_row(ab * cd);
            const double rab = cd * rq, rcd = ab * rq;
            r[j] = d[j + 1] * rab;
            r[j + 1] = d[j] * rab;
            r[j + 2] = d[j + 3] * rcd;
            r[j + 3] = d[j + 2] * rcd;
        }
    } else if constexpr (NR % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NR; j += 2) {
            const double rp = rcp_row(d[j] * d[j + 1]);
            r[j] = d[j + 1] * rp;
            r[j + 1] = d[j] * rp;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NR; j++) r[j] = rcp_row(d[j]);
    }
}

// FAST solve for L >= 128, interleaved layout (lane l holds the NR = 2^k adjacent rows NR*l .. NR*l+NR-1):
// k IN-LANE CYCLIC-REDUCTION LEVELS, THEN PCR ON 64 UNKNOWNS, THEN BACK-SUBSTITUTION.
//   forward, level h = 1, 2, .. NR/2: rows j = h (mod 2h) are normalised by 1/d and eliminated from
//     the rows j = 0 (mod 2h):  with L = row j-h and R = row j+h (normalised: a^, c^, b^)
//         A_j = -a_j a^_L,  C_j = -c_j c^_R,  D_j = d_j - a_j c^_L - c_j a^_R,  B_j = b_j - a_j b^_L - c_j b^_R
//     everything is in-lane except L of row 0, which is row NR-h of lane l-1 (one DPP rotate);
//   PCR on the remaining row 0 of every lane: 5 levels (lane shifts 1..16: DPP for 1, 8 bytes per
//     lane through LDS for the rest) + the lane^32 pairs by Cramer's rule;
//   backward: x_j = b^_j - a^_j x_{j-h} - c^_j x_{j+h}, where x_{NR} is row 0 of lane l+1 (one DPP).
// Versus pure PCR (log2(L)-1 levels on NR rows per lane) this is (NR-1) eliminations + 6 one-row
// levels: ~40 % fewer solve flops at L = 128, ~4x fewer at L = 512, and the LDS only ever carries
// one value per lane.  Same solution as the reference's PCR up to rounding (both are exact
// eliminations of a diagonally dominant system).  The corner coefficients stay exact zeros (a of
// row 0 and c of row L-1 propagate), so wrapped neighbour values drop out.
template <typename T> __device__ __forceinline__ T rcp_fast(T d);
template <> __device__ __forceinline__ double rcp_fast<double>(double d) { return rcp_steps<TRPL_RCP_SOLVE_STEPS>(d); }
template <> __device__ __forceinline__ float rcp_fast<float>(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);      // v_rcp_f32 (1 ulp) + one Newton step
    return __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.0f), r);
}

// value held by lane ^ 32, through the LDS crossbar (ds_bpermute): the VALU is this kernel's saturated unit, the LDS is not
// (v_permlane32_swap measured 1.9 % slower, and 0.9 % slower for the lone wave of the L = 512 stepper)
__device__ __forceinline__ double partner32(double v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float partner32(float v) { return __shfl_xor(v, 32, 64); }

// Two systems per wavefront (stepper_pair_impl.hpp) run this solver on WS = 32 lanes each: the wave's
// 2 x 32 reduced unknowns form one block-diagonal system whose coupling across the seams is exactly
// zero, so the same lane shifts work; with ISO the values that cross a seam are replaced by 0 (and
// the LDS indices stay inside the half) so that a non-finite value of one system can never reach
// the other through a 0 * NaN.
// ds_swizzle_b32 in rotate mode (offset 0xC000 | dir << 10 | n << 5; probed on gfx950 with
// tools/swizzle_probe.hip): every group of 32 lanes is rotated by n, dir 0: lane i <- lane (i+n) & 31,
// dir 1: lane i <- lane (i-n) & 31.  No LDS memory is touched (crossbar only) and the rotation never
// leaves a 32-lane group: exactly the neighbour fetch of a system that owns 32 lanes.
template <int PAT, typename T>
__device__ __forceinline__ T swizzle(T v)
{
    if constexpr (sizeof(T) == 8) {
        return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), PAT),
                                __builtin_amdgcn_ds_swizzle(__double2loint(v), PAT));
    } else {
        return __builtin_bit_cast(T, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), PAT));
    }
}
template <int N, typename T> __device__ __forceinline__ T rot32_up(T v) { return swizzle<0xC000 | (N << 5)>(v); }
template <int N, typename T> __device__ __forceinline__ T rot32_dn(T v) { return swizzle<0xC000 | (1 << 10) | (N << 5)>(v); }
template <typename T> __device__ __forceinline__ T swap16(T v) { return swizzle<(0x10 << 10) | 0x1f>(v); }   // lane ^ 16

// A voided fp64 value only has its HIGH dword cleared (one v_cndmask instead of two): what is left is
// a denormal (|v| < 2^-1022), finite whatever the original was, and the exact-zero coefficient it
// meets turns it into +-0.  void_value() clears both dwords where the value itself matters.
template <typename T>
__device__ __forceinline__ T void_hi(bool c, T v)
{
    if constexpr (sizeof(T) == 8) return __hiloint2double(c ? 0 : __double2hiint(v), __double2loint(v));
    else return c ? T(0) : v;
}
template <bool ISO, int WS, typename T>
__device__ __forceinline__ T seam_first(T v, int lane)      // v arrived from lane-1: void on a system's first lane
{
    if constexpr (ISO) return void_hi((lane & (WS - 1)) == 0, v);
    else return v;
}
template <bool ISO, int WS, typename T>
__device__ __forceinline__ T seam_last(T v, int lane)       // v arrived from lane+1: void on a system's last lane
{
    if constexpr (ISO) return void_hi((lane & (WS - 1)) == WS - 1, v);
    else return v;
}

// SWZ (32-lane systems only): the strides >= 2 and the pair step on ds_swizzle rotates instead of the LDS exchange buffer
// (+2.8 %; the stride-1 level and the stride-1 fetches of the CR levels stay on DPP: on ds_swizzle 0 / -1 %, measured twice).
template <typename T, int S, int WS = 64, bool ISO = false, bool SWZ = false>
__device__ __forceinline__ void pcr64_levels(T &A, T &D, T &C, T &Bv, int lane, T *xch)
{
    static_assert(!SWZ || WS == 32, "swizzle rotates work on 32-lane groups");
    if constexpr (S < WS / 2) {
        const T r = rcp_fast<T>(D);
        const T nA = A * r, nC = C * r, nB = Bv * r;
        T Am, Cm, Bm, Ap, Cp, Bp;
        if constexpr (S > 1 && SWZ) {             // rotate inside the system's 32 lanes
            Am = rot32_dn<S>(nA); Cm = rot32_dn<S>(nC); Bm = rot32_dn<S>(nB);
            Ap = rot32_up<S>(nA); Cp = rot32_up<S>(nC); Bp = rot32_up<S>(nB);
        } else if constexpr (S == 1) {            // DPP wave rotates
            Am = seam_first<ISO, WS>(lane_dn<1>(nA, lane), lane); Cm = seam_first<ISO, WS>(lane_dn<1>(nC, lane), lane);
            Bm = seam_first<ISO, WS>(lane_dn<1>(nB, lane), lane);
            Ap = seam_last<ISO, WS>(lane_up<1>(nA, lane), lane); Cp = seam_last<ISO, WS>(lane_up<1>(nC, lane), lane);
            Bp = seam_last<ISO, WS>(lane_up<1>(nB, lane), lane);
        } else {                                   // staged through LDS, one value per lane and array (3 writes + 6 reads of 8 bytes
                                                   // per level; ds_bpermute -- one LDS trip, 12 instructions -- measured 8-10 % slower)
            xch[0 * 64 + lane] = nA;
            xch[1 * 64 + lane] = nC;
            xch[2 * 64 + lane] = nB;
            // ISO: wrap inside the system's own lanes (its own rows, multiplied by exact zeros)
            const int dn = ISO ? (((lane - S) & (WS - 1)) | (lane & (64 - WS))) : ((lane - S) & 63);
            const int up = ISO ? (((lane + S) & (WS - 1)) | (lane & (64 - WS))) : ((lane + S) & 63);
            Am = xch[0 * 64 + dn]; Cm = xch[1 * 64 + dn]; Bm = xch[2 * 64 + dn];
            Ap = xch[0 * 64 + up]; Cp = xch[1 * 64 + up]; Bp = xch[2 * 64 + up];
        }
        D = D - A * Cm - C * Ap;
        Bv = Bv - A * Bm - C * Bp;
        A = -A * Am;
        C = -C * Cp;
        pcr64_levels<T, S * 2, WS, ISO, SWZ>(A, D, C, Bv, lane, xch);
    }
}

// one forward cyclic-reduction level with in-lane stride H (rows H, 3H, .. eliminated)
template <typename T, int NR, int H, int WS = 64, bool ISO = false>
__device__ __forceinline__ void cr_forward(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], int lane)
{
    if constexpr (H < NR) {
        // normalise the eliminated rows in place: (ld, ud, B)[q] become (a^, c^, b^); in fp64 two rows share
        // one v_rcp_f64 (r = 1/(d_q d_q'), 1/d_q = d_q' r: the diagonals are O(1..1e3))
        constexpr int NQ = (NR - H + 2 * H - 1) / (2 * H);          // rows H, 3H, 5H, ...
        if constexpr (sizeof(T) == 8 && NQ % 2 == 0) {
#pragma unroll
            for (int q = H; q < NR; q += 4 * H) {
                const int q2 = q + 2 * H;
                const T rp = rcp_fast<T>(d[q] * d[q2]);
                const T r1 = d[q2] * rp, r2 = d[q] * rp;
                ld[q] *= r1; ud[q] *= r1; B[q] *= r1;
                ld[q2] *= r2; ud[q2] *= r2; B[q2] *= r2;
            }
        } else {
#pragma unroll
            for (int q = H; q < NR; q += 2 * H) {
                const T r = rcp_fast<T>(d[q]);
                ld[q] *= r; ud[q] *= r; B[q] *= r;
            }
        }
        // the left neighbour of row 0 is row NR-H of lane l-1
        const T aL0 = seam_first<ISO, WS>(lane_dn<1>(ld[NR - H], lane), lane);
        const T cL0 = seam_first<ISO, WS>(lane_dn<1>(ud[NR - H], lane), lane);
        const T bL0 = seam_first<ISO, WS>(lane_dn<1>(B[NR - H], lane), lane);
#pragma unroll
        for (int p = 0; p < NR; p += 2 * H) {
            const T aL = p == 0 ? aL0 : ld[p == 0 ? 0 : p - H], cL = p == 0 ? cL0 : ud[p == 0 ? 0 : p - H],
                    bL = p == 0 ? bL0 : B[p == 0 ? 0 : p - H];
            const T aR = ld[p + H], cR = ud[p + H], bR = B[p + H];
            const T a = ld[p], c = ud[p];
            d[p] = d[p] - a * cL - c * aR;
            B[p] = B[p] - a * bL - c * bR;
            ld[p] = -a * aL;
            ud[p] = -c * cR;
        }
        cr_forward<T, NR, 2 * H, WS, ISO>(ld, d, ud, B, lane);
    }
}

// back-substitution of the rows eliminated at stride H (after all coarser levels); xnext = row 0 of lane l+1
template <typename T, int NR, int H>
__device__ __forceinline__ void cr_backward(const T (&ld)[NR], const T (&ud)[NR], const T (&B)[NR], T (&x)[NR], T xnext)
{
    if constexpr (H >= 1) {
#pragma unroll
        for (int q = H; q < NR; q += 2 * H) {
            const T xr = q + H < NR ? x[q + H < NR ? q + H : 0] : xnext;
            x[q] = B[q] - ld[q] * x[q - H] - ud[q] * xr;
        }
        cr_backward<T, NR, H / 2>(ld, ud, B, x, xnext);
    }
}

// WS lanes per system (64: one system per wave; 32: two); the final pairs sit in lanes l, l ^ (WS/2)
template <typename T, int NR, int WS = 64, bool ISO = false, bool SWZ = false>
__device__ __forceinline__ void cr_pcr_solve(T (&ld)[NR], T (&d)[NR], T (&ud)[NR], T (&B)[NR], T (&x)[NR], int lane,
                                             T *xch)
{
    cr_forward<T, NR, 1, WS, ISO>(ld, d, ud, B, lane);
    T A = ld[0], D = d[0], C = ud[0], Bv = B[0];
    // the cross-lane levels are one long dependent chain (reciprocal -> normalise -> exchange -> eliminate):
    // a wave inside it is issued ahead of its SIMD neighbour, which then fills the gaps with its assembly
    // (+0.6..0.8 % on the paired kernel, +1.2 % on the one-system kernel, same-box A/B; priority over the
    // whole solve: +0.1 %; levels 0 / 1 / 3 re-measured in round 4: -0.4 / +0.1 / +0.15 %, noise)
    __builtin_amdgcn_s_setprio(2);
    pcr64_levels<T, 1, WS, ISO, SWZ>(A, D, C, Bv, lane, xch);
    // the final pairs (lanes l, l ^ WS/2) by Cramer's rule, own unknown only
    // the coupling to the partner row: C in the lower half, A in the upper.  After the levels S = 1 .. WS/4 the other one
    // is an exact zero (rows below 2 S have lost their sub-diagonal, rows above WS - 2 S their super-diagonal: products
    // with the exact zeros of rows 0 and WS - 1), so A + C is that coupling, bit for bit -- one add instead of two selects
    const T c_own = A + C;
    T D_oth, B_oth, c_oth;
    if constexpr (WS == 64) {
        D_oth = partner32(D); B_oth = partner32(Bv); c_oth = partner32(c_own);
    } else if constexpr (WS == 32 && SWZ) {
        D_oth = swap16(D); B_oth = swap16(Bv); c_oth = swap16(c_own);
    } else {
        D_oth = __shfl_xor(D, WS / 2, 64); B_oth = __shfl_xor(Bv, WS / 2, 64); c_oth = __shfl_xor(c_own, WS / 2, 64);
    }
    const T X = (Bv * D_oth - c_own * B_oth) * rcp_fast<T>(D * D_oth - c_own * c_oth);
    x[0] = X;
    __builtin_amdgcn_s_setprio(0);
    const T xnext = seam_last<ISO, WS>(lane_up<1>(X, lane), lane);      // a system's last lane: times c^ = 0
    cr_backward<T, NR, NR / 2>(ld, ud, B, x, xnext);
}

}  // namespace trpl
