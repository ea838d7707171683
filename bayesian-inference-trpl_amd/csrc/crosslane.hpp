// Cross-lane primitives of the gfx950 TRPL kernels: neighbour fetches in the blocked layout
// (ds_bpermute / DPP with row wrap), the interleaved layout (compile-time lane shifts: in-lane,
// DPP wave rotate, ds_bpermute), wave reductions (reference-order butterfly, DPP row reduction),
// lane-pair exchange (v_permlane32_swap).  64-lane wavefronts throughout.
#pragma once
#include <math.h>
#include <float.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "trpl_common.hpp"

// Build-time switches that remain (round 5 removed those of rejected experiments -- TRPL_FAST_WAVES, TRPL_L512_WAVES,
// TRPL_H32_*, TRPL_CR_HYBRID, TRPL_PARTNER_BPERMUTE, TRPL_PCR_S1_LDS, TRPL_PCR_SETPRIO, TRPL_NORM_VOTE_DEFER[1], TRPL_PCR_BPERMUTE,
// TRPL_PAIRSTEP_ADD, TRPL_VOTE_FASTPATH, TRPL_WITNESS_P, TRPL_RCP_QUAD / _PAIR, TRPL_PAIR_XM, TRPL_PCRB_*: the measured winner of
// each is now simply the code; the alternatives and their numbers are in git history and DESIGN_HISTORY.md section 8).
#ifndef TRPL_PAIR_OPTIMISTIC
#define TRPL_PAIR_OPTIMISTIC 1    // paired kernel: iterate without the seam selects, repeat a time step with them when a system is
                                  // flagged in it or leaves it non-finite (stepper_pair_impl.hpp); 0 = the selects in every iteration
                                  // (that form is in the library anyway: TRPL_FLAG_PAIR_ALWAYS_SEAM)
#endif
#ifndef TRPL_NORM_VOTE
#define TRPL_NORM_VOTE 1          // FAST residual tests: the sign of sum(|r| - TOL |b|) from a lane vote where all lanes agree
                                  // (no reduction); 0 = always reduce.  Same decisions either way.
#endif
#ifndef TRPL_PAIR_WITNESS
#define TRPL_PAIR_WITNESS 1       // optimistic seam: 0 drops the finiteness witness (the first, flawed form; to show that the tests see it)
#endif
#if TRPL_PAIR_WITNESS == 0 && !defined(TRPL_DEBUG)
#error "TRPL_PAIR_WITNESS=0 is the known-flawed form of the optimistic seam: it only builds with -DTRPL_DEBUG (to show that the differential tests catch it)"
#endif
#ifndef TRPL_VOTE_STATS
#define TRPL_VOTE_STATS 0         // measurement build only: the paired kernel packs its count of residual reductions into iters_total
#endif
// Refinement of v_rcp_f64 (1: one Newton step, 2e-15 relative and always BELOW 1 / x; 2: two steps; 3: one third-order step,
// ~1 ulp, unbiased; 0: the IEEE divide expansion), separately for the quotients of the tridiagonal solver (rcp_fast<double>:
// CR and PCR levels, the final pairs) and for the pointwise reciprocals of the assembly and the field update (rcp_rows, the
// surface term).  Round 4: in the solver the one-step form's bias acts as a spurious sink proportional to
// the grid's stiffness D dt/dx^2 -- with 3 the paired kernel's distance from the reference evaluation over 8000 steps drops from
// 1e-10 (median) / 7e-9 (max) to 3e-13 / 2e-11 on the 311 nm films for 1.5 % of throughput (2 steps: the same accuracy for
// 3.5 %); in the pointwise reciprocals it changes nothing measurable.
#ifndef TRPL_RCP_SOLVE_STEPS
#define TRPL_RCP_SOLVE_STEPS 3
#endif
#ifndef TRPL_RCP_ROWS_STEPS
#define TRPL_RCP_ROWS_STEPS 1
#endif

namespace trpl {

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
        for (int j = 0; j < NR; j++) y[j] = wrap ? +s[(j + 1) % NR] : +s[j];
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
        for (int j = 0; j < NR; j++) y[j] = wrap ? +s[(j + NR - 1) % NR] : +s[j];
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

// Hide a value's provenance from the optimiser.  Without it LLVM packs the shuffled rows into a
// vector and turns `wrap ? s[j+1] : s[j]` into a dynamically indexed extract, which lands in
// scratch memory (seen at the stride-32 level, where the up and down sources coincide).
template <typename T>
__device__ __forceinline__ T pick(bool c, T a, T b)
{
    asm volatile("" : "+v"(a));
    asm volatile("" : "+v"(b));
    return c ? a : b;
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], CTRL, 0xF, 0xF, false);   // every lane is written:
    r.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], CTRL, 0xF, 0xF, false);   // no destination init needed
    return r.d;
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int kDppWaveRol1 = 0x134;   // lane l <- lane (l+1) & 63
constexpr int kDppWaveRor1 = 0x13C;   // lane l <- lane (l-1) & 63

// y[j] = x at node i+1 / i-1; the out-of-range entry (last row's last lane / first row's first
// lane) holds an arbitrary in-array value.  W == 64 uses DPP, narrower systems the generic path.
template <typename T, int NR, int W>
__device__ __forceinline__ void fetch_up1(const T (&x)[NR], T (&y)[NR], int ln)
{
    if constexpr (W == 64) {
        T r[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) r[j] = dpp_mov<kDppWaveRol1>(x[j]);
#pragma unroll
        for (int j = 0; j < NR - 1; j++) y[j] = ln == 63 ? +r[j + 1] : +r[j];
        y[NR - 1] = r[NR - 1];
    } else {
        fetch_up<T, NR, W, 1>(x, y, ln);
    }
}
template <typename T, int NR, int W>
__device__ __forceinline__ void fetch_dn1(const T (&x)[NR], T (&y)[NR], int ln)
{
    if constexpr (W == 64) {
        T r[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) r[j] = dpp_mov<kDppWaveRor1>(x[j]);
        y[0] = r[0];
#pragma unroll
        for (int j = 1; j < NR; j++) y[j] = ln == 0 ? +r[j - 1] : +r[j];
    } else {
        fetch_dn<T, NR, W, 1>(x, y, ln);
    }
}

// neighbour fetch for the fast PCR: like fetch_up/fetch_dn but the entry that is always out of
// range is not fixed up (saves the select), and stride 1 goes through DPP.
template <typename T, int NR, int W, int RF>
__device__ __forceinline__ void nb_up(const T (&x)[NR], T (&y)[NR], int ln)
{
    if constexpr (RF == 1) {
        fetch_up1<T, NR, W>(x, y, ln);
    } else if constexpr (RF >= W) {
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
        for (int j = 0; j < NR - 1; j++) y[j] = pick(wrap, s[j + 1], s[j]);
        y[NR - 1] = s[NR - 1];
    }
}
template <typename T, int NR, int W, int RF>
__device__ __forceinline__ void nb_dn(const T (&x)[NR], T (&y)[NR], int ln)
{
    if constexpr (RF == 1) {
        fetch_dn1<T, NR, W>(x, y, ln);
    } else if constexpr (RF >= W) {
        constexpr int m = RF / W;
#pragma unroll
        for (int j = 0; j < NR; j++) y[j] = x[(j + NR - m) % NR];
    } else {
        const int src = (ln - RF) & (W - 1);
        const bool wrap = ln < RF;
        T s[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) s[j] = __shfl(x[j], src, 64);
        y[0] = s[0];
#pragma unroll
        for (int j = 1; j < NR; j++) y[j] = pick(wrap, s[j - 1], s[j]);
    }
}


// ------------------------------------------------------------------------------------------
// FAST mode, L >= 128: INTERLEAVED layout  node i = NR*lane + j  (NR = L/64 consecutive nodes per
// lane).  Every neighbour i +- RF is then (lane +- K, row j') with K and j' known at compile
// time, so a fetch is a pure lane shift -- no per-lane selects at all:
//      K = 0 : in-lane register move          K = 1 : DPP wave rotate (VALU, no LDS)
//      K >= 2: ds_bpermute                    final pairing (lane ^ 32): v_permlane32_swap
// Wave-wide sums (residual norms, PL) are DPP row reductions ending in lane 63 + v_readlane.
// The reduction order differs from the reference's tree, which is why STRICT mode keeps the
// blocked layout above.
// ------------------------------------------------------------------------------------------
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int K, typename T>
__device__ __forceinline__ T lane_up(T v, int lane)       // value held by lane + K (mod 64)
{
    if constexpr (K == 0) return v;
    else if constexpr (K == 1) return dpp_mov<kDppWaveRol1>(v);
    else return __shfl(v, (lane + K) & 63, 64);
}
template <int K, typename T>
__device__ __forceinline__ T lane_dn(T v, int lane)       // value held by lane - K (mod 64)
{
    if constexpr (K == 0) return v;
    else if constexpr (K == 1) return dpp_mov<kDppWaveRor1>(v);
    else return __shfl(v, (lane - K) & 63, 64);
}

// y[j] = x at node i+RF / i-RF in the interleaved layout (wrapped lanes give in-array values)
template <typename T, int NR, int RF>
__device__ __forceinline__ void nbrB_up(const T (&x)[NR], T (&y)[NR], int lane)
{
    static_for<NR>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        y[j] = lane_up<(j + RF) / NR>(x[(j + RF) % NR], lane);
    });
}
template <typename T, int NR, int RF>
__device__ __forceinline__ void nbrB_dn(const T (&x)[NR], T (&y)[NR], int lane)
{
    static_for<NR>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int K = RF > j ? (RF - j + NR - 1) / NR : 0;
        y[j] = lane_dn<K>(x[((j - RF) % NR + NR) % NR], lane);
    });
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_add(double v)
{
    int lo, hi;
    if constexpr (ROWMASK == 0xF) {      // all rows written (out-of-row sources read 0): no init
        lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    } else {                             // masked rows keep the 0 they are initialised with
        lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, true);
        hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, true);
    }
    return v + __hiloint2double(hi, lo);
}
// Sum of v over the 64 lanes, returned wave-uniform (SGPRs).
__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add<0x111, 0xF>(v);          // row_shr:1
    v = dpp_add<0x112, 0xF>(v);          // row_shr:2
    v = dpp_add<0x114, 0xF>(v);          // row_shr:4
    v = dpp_add<0x118, 0xF>(v);          // row_shr:8   -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xA>(v);          // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xC>(v);          // row_bcast:31 into rows 2,3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Is the wave-wide sum of v negative?  Where every lane's term is negative the sum is, where every lane's term is >= 0 it
// is not -- whatever the order of summation -- and a vote (two v_cmp, scalar compares) replaces the 6-step DPP reduction.
// Only with mixed signs (or a NaN, which fails both votes) does the sum itself decide.  In the steppers the terms are
// |r| - TOL |b| per lane: in a time step's first iteration every lane is far above zero, in its last one nearly always
// every lane is below.
__device__ __forceinline__ bool wave_sum_negative(double v)
{
    if constexpr (TRPL_NORM_VOTE != 0) {
        const unsigned long long neg = __builtin_amdgcn_ballot_w64(v < 0.0), nonneg = __builtin_amdgcn_ballot_w64(v >= 0.0);
        if (neg == ~0ull) return true;
        if (nonneg == ~0ull) return false;
    }
    return wave_sum(v) < 0.0;
}

// (value of the lower-half lane, value of the upper-half lane) of each lane pair (l, l^32), in
// every lane: v_permlane32_swap on two copies of v.
__device__ __forceinline__ void pair32(double v, double &lo_half, double &hi_half)
{
    const unsigned a = (unsigned)__double2loint(v), b = (unsigned)__double2hiint(v);
    const auto r0 = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    lo_half = __hiloint2double((int)r1[0], (int)r0[0]);
    hi_half = __hiloint2double((int)r1[1], (int)r0[1]);
}

}  // namespace trpl
