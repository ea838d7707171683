// Stand-alone batched tridiagonal solve (measurement unit U1): S systems of L unknowns,
// operands and result in HBM ([S][L] arrays), one wavefront per system; STRICT: the elimination order
// of pcreduce (pvSimPCR.py:42-81), FAST: cyclic reduction in-lane + PCR on one row per lane (pcr.hpp).
// Bound: HBM bandwidth, 5*L*sizeof(T) algorithmic bytes per system (read ld, d, ud, b; write x).
// Also measured (round 2): a persistent launch whose waves load system i+1 before solving system i is SLOWER
// than letting the hardware switch between many one-system waves (5.17 vs 5.41 TB/s); the placement of the five
// arrays matters (include/trpl.h).
#pragma once
#include "stepper_f32_impl.hpp"

namespace trpl {

// Launch shape and load policy, measured with the operands rotated through 1.34 GB (tools/bench_pcr_ab.py, GB/s at
// L = 128 fp64 | L = 128 fp32 | L = 256 fp64 | L = 512 fp32 | L = 512 fp64):
//   4 waves per workgroup, 2 systems per wave (round 1)      5519 | 5382 | 5835 | 5829 | 5530
//   1 wave per workgroup, one system per wave                5690 | 5583 | 5936 | 5929 | 5706
//   ... + non-temporal loads                                 6467 | 6040 | 5429 | 5424 | 3854
// Non-temporal (streaming) loads pay when every cache line is consumed by ONE load instruction -- a lane's
// chunk of a row is at most 16 bytes -- and cost up to 2x HBM reads when a lane needs several 16-byte loads
// per row (interleaved layout, NR * sizeof(T) > 16: the line is gone before the second instruction asks for
// its other half).  So: one wave per workgroup everywhere, non-temporal loads only where an instruction
// consumes whole lines: directly for 16-byte chunks, through an LDS transpose for wider rows (L = 512 fp64 5.72 -> 5.97 TB/s,
// L = 256 fp64 / L = 512 fp32 +1 %); since round 6 the wide rows' RESULT takes the same route back (pcrb_store_row).  (These were build switches TRPL_PCRB_WAVES / _NT / _STAGE / _CAP until round 5.)
constexpr int kPcrbGridCap = 256 * 256;     // workgroups; beyond that a wave loops over systems

template <bool NT, typename T> __device__ __forceinline__ T pcrb_load(const T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}

// a lane's NR adjacent elements, 16 bytes per load instruction where the row allows it (global memory needs
// dword alignment only, so the vector type is declared with the element's alignment)
template <bool NT, typename T, int NR>
__device__ __forceinline__ void pcrb_load_row(const T *p, T (&v)[NR])
{
    constexpr int PER = 16 / sizeof(T);
    if constexpr (NR % PER == 0) {
        typedef T vec16 __attribute__((ext_vector_type(PER), aligned(sizeof(T))));
#pragma unroll
        for (int k = 0; k < NR / PER; k++) {
            vec16 t;
            if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const vec16 *>(p) + k);
            else t = reinterpret_cast<const vec16 *>(p)[k];
#pragma unroll
            for (int e = 0; e < PER; e++) v[k * PER + e] = t[e];
        }
    } else {
#pragma unroll
        for (int j = 0; j < NR; j++) v[j] = pcrb_load<NT>(p + j);
    }
}

// A row of L elements streamed with fully coalesced 16-byte-per-lane non-temporal loads (every cache line is
// consumed by one instruction), parked in LDS in node order and read back as the lane's NR adjacent elements.
// A wavefront's DS instructions execute in order: no barrier.
template <typename T, int L, int NR>
__device__ __forceinline__ void pcrb_issue_row(const T *row, int lane, T (&stage)[NR])
{
    constexpr int PER = 16 / sizeof(T), CH = L / (64 * PER);        // elements per 16 bytes, chunks of 1 KiB per row
    typedef T vec16 __attribute__((ext_vector_type(PER), aligned(sizeof(T))));
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const vec16 t = __builtin_nontemporal_load(reinterpret_cast<const vec16 *>(row) + k * 64 + lane);
#pragma unroll
        for (int e = 0; e < PER; e++) stage[k * PER + e] = t[e];
    }
}
template <typename T, int L, int NR>
__device__ __forceinline__ void pcrb_transpose_row(const T (&stage)[NR], T *buf, int lane, T (&v)[NR])
{
    constexpr int PER = 16 / sizeof(T), CH = L / (64 * PER);
    typedef T vec16 __attribute__((ext_vector_type(PER), aligned(16)));
#pragma unroll
    for (int k = 0; k < CH; k++) {
        vec16 t;
#pragma unroll
        for (int e = 0; e < PER; e++) t[e] = stage[k * PER + e];
        reinterpret_cast<vec16 *>(buf)[k * 64 + lane] = t;
    }
#pragma unroll
    for (int k = 0; k < NR / PER; k++) {
        const vec16 t = reinterpret_cast<const vec16 *>(buf)[lane * (NR / PER) + k];
#pragma unroll
        for (int e = 0; e < PER; e++) v[k * PER + e] = t[e];
    }
}

// The result row the other way round (round 6): the lane's NR adjacent elements parked in LDS, read back in node order and
// stored with fully coalesced 16-byte-per-lane NON-TEMPORAL stores -- a wave-instruction writes 1 KiB of whole lines, where the
// direct form writes 16 bytes out of every NR * sizeof(T) per instruction.  Same-box A/B, three alternating repetitions
// (profiles/r6_pcr_store_ab.txt): L = 512 fp64 5.92 -> 6.13 TB/s (+3.6 %), L = 512 fp32 6.01 -> 6.13 (+2.0 %), L = 256 fp64
// 5.99 -> 6.12 (+2.2 %); through LDS with ordinary stores -0.5 %; non-temporal stores on the rows that are already one
// 16-byte chunk per lane (L = 128 fp64, L = 256 fp32): no difference.
template <typename T, int L, int NR>
__device__ __forceinline__ void pcrb_store_row(const T (&v)[NR], T *buf, int lane, T *row)
{
    constexpr int PER = 16 / sizeof(T), CH = L / (64 * PER);
    typedef T vec16 __attribute__((ext_vector_type(PER), aligned(16)));
    typedef T gvec16 __attribute__((ext_vector_type(PER), aligned(sizeof(T))));
#pragma unroll
    for (int k = 0; k < NR / PER; k++) {
        vec16 t;
#pragma unroll
        for (int e = 0; e < PER; e++) t[e] = v[k * PER + e];
        reinterpret_cast<vec16 *>(buf)[lane * (NR / PER) + k] = t;
    }
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const vec16 t = reinterpret_cast<const vec16 *>(buf)[k * 64 + lane];
        gvec16 g;
#pragma unroll
        for (int e = 0; e < PER; e++) g[e] = t[e];
        __builtin_nontemporal_store(g, reinterpret_cast<gvec16 *>(row) + k * 64 + lane);
    }
}

template <typename T, int L, bool STRICT>
__global__ void __launch_bounds__(64) pcr_batched_kernel(const T *__restrict__ ld, const T *__restrict__ d,
                                                          const T *__restrict__ ud, const T *__restrict__ b,
                                                          T *__restrict__ x, int64_t S)
{
    constexpr int W = L < 64 ? L : 64;
    constexpr int NR = L / W;
    constexpr bool NT = NR * sizeof(T) <= 16;
    const int lane = threadIdx.x;
    const int ln = lane & (W - 1);
    // exchange buffer of the solve: the cyclic-reduction + PCR solver stages one value per lane and array (3 x 64)
    __shared__ __attribute__((aligned(16))) T xch[(!STRICT && L >= 128) ? 3 * 64 : 4];
    (void)xch;
    constexpr bool STAGE = !STRICT && L >= 128 && NR * sizeof(T) > 16;
    __shared__ __attribute__((aligned(16))) T rowbuf[STAGE ? L : 4];
    (void)rowbuf;
    const int64_t wave = blockIdx.x;
    const int64_t nwaves = gridDim.x;
    for (int64_t s = wave; s < S; s += nwaves) {
        const int64_t base = s * L;
        T vl[NR], vd[NR], vu[NR], vb[NR], vx[NR];
        if constexpr (!STRICT && L >= 128) {
            // interleaved layout: lane owns nodes NR*lane .. NR*lane+NR-1 -> 16-byte loads, fully
            // coalesced 1 KiB per wave-instruction
            if constexpr (STAGE) {
                T sl[NR], sd[NR], su[NR], sb[NR];              // all four rows in flight, then through LDS one by one
                pcrb_issue_row<T, L, NR>(ld + base, lane, sl); pcrb_issue_row<T, L, NR>(d + base, lane, sd);
                pcrb_issue_row<T, L, NR>(ud + base, lane, su); pcrb_issue_row<T, L, NR>(b + base, lane, sb);
                pcrb_transpose_row<T, L, NR>(sl, rowbuf, lane, vl); pcrb_transpose_row<T, L, NR>(sd, rowbuf, lane, vd);
                pcrb_transpose_row<T, L, NR>(su, rowbuf, lane, vu); pcrb_transpose_row<T, L, NR>(sb, rowbuf, lane, vb);
            } else {
                const int64_t o = base + NR * lane;
                pcrb_load_row<NT>(ld + o, vl); pcrb_load_row<NT>(d + o, vd);
                pcrb_load_row<NT>(ud + o, vu); pcrb_load_row<NT>(b + o, vb);
            }
            cr_pcr_solve<T, NR>(vl, vd, vu, vb, vx, lane, xch);
            if constexpr (STAGE) {
                pcrb_store_row<T, L, NR>(vx, rowbuf, lane, x + base);
            } else {
#pragma unroll
                for (int j = 0; j < NR; j++) x[base + NR * lane + j] = vx[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int64_t o = base + ln + W * j;
                vl[j] = pcrb_load<NT>(ld + o); vd[j] = pcrb_load<NT>(d + o); vu[j] = pcrb_load<NT>(ud + o); vb[j] = pcrb_load<NT>(b + o);
            }
            tridiag_solve<STRICT, T, NR, W, L>(vl, vd, vu, vb, vx, ln);
            if (lane < W) {
#pragma unroll
                for (int j = 0; j < NR; j++) x[base + ln + W * j] = vx[j];
            }
        }
    }
}

template <typename T, bool STRICT>
hipError_t launch_pcr_batched_t(const void *ld, const void *d, const void *ud, const void *b, void *x, int64_t S,
                                int L, hipStream_t stream)
{
    if (S <= 0) return hipSuccess;
    const int64_t blocks = S < kPcrbGridCap ? S : kPcrbGridCap;      // one wavefront per workgroup and system
    dim3 grid((unsigned)blocks), block(64);
    switch (L) {
#define TRPL_CASE(LL)                                                                                      \
    case LL:                                                                                               \
        hipLaunchKernelGGL((pcr_batched_kernel<T, LL, STRICT>), grid, block, 0, stream, (const T *)ld, (const T *)d, \
                           (const T *)ud, (const T *)b, (T *)x, S);                                        \
        break;
        TRPL_CASE(4) TRPL_CASE(8) TRPL_CASE(16) TRPL_CASE(32) TRPL_CASE(64) TRPL_CASE(128) TRPL_CASE(256)
        TRPL_CASE(512)
#undef TRPL_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace trpl
