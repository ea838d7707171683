// Stand-alone batched tridiagonal solve (measurement unit U1): S systems of L unknowns,
// operands and result in HBM ([S][L] arrays), one wavefront per system; STRICT: the elimination order
// of pcreduce (pvSimPCR.py:42-81), FAST: cyclic reduction in-lane + PCR on one row per lane (pcr.hpp).
// Bound: HBM bandwidth, 5*L*sizeof(T) algorithmic bytes per system (read ld, d, ud, b; write x).
// Measured and NOT adopted (round 2, operands rotated through 1.34 GB so that nothing is served by the
// Infinity Cache; tools/bench_pcr_ab.py): a persistent launch whose waves load system i+1 before solving
// system i: 5.17 TB/s against 5.41 TB/s for this one-pass form (the hardware's own wave switching already
// overlaps the loads of 16 resident waves per CU with the solves); the same with non-temporal loads: 5.48.
// Four read streams + one write stream saturate at ~5.5 TB/s here (the guide's 6.3 TB/s is a 1:1 copy).
#pragma once
#include "stepper_f32_impl.hpp"

namespace trpl {

template <typename T, int L, bool STRICT>
__global__ void __launch_bounds__(256) pcr_batched_kernel(const T *__restrict__ ld, const T *__restrict__ d,
                                                          const T *__restrict__ ud, const T *__restrict__ b,
                                                          T *__restrict__ x, int64_t S)
{
    constexpr int W = L < 64 ? L : 64;
    constexpr int NR = L / W;
    const int lane = threadIdx.x & 63;
    const int ln = lane & (W - 1);
    // exchange buffer of the solve, private to each of the 4 waves: the cyclic-reduction + PCR solver stages one
    // value per lane and array (3 x 64), the pure-PCR variants all L rows
    constexpr int XW = TRPL_CR_HYBRID != 0 ? 64 : L;
    __shared__ __attribute__((aligned(16))) T xch_all[(!STRICT && L >= 128) ? 4 * 3 * XW : 4];
    T *xch = xch_all + (threadIdx.x >> 6) * 3 * XW;
    (void)xch;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t s = wave; s < S; s += nwaves) {
        const int64_t base = s * L;
        T vl[NR], vd[NR], vu[NR], vb[NR], vx[NR];
        if constexpr (!STRICT && L >= 128) {
            // interleaved layout: lane owns nodes NR*lane .. NR*lane+NR-1 -> 16-byte loads, fully
            // coalesced 1 KiB per wave-instruction
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int64_t o = base + NR * lane + j;
                vl[j] = ld[o]; vd[j] = d[o]; vu[j] = ud[o]; vb[j] = b[o];
            }
            if constexpr (TRPL_CR_HYBRID != 0) cr_pcr_solve<T, NR>(vl, vd, vu, vb, vx, lane, xch);
            else if constexpr (sizeof(T) == 8) pcr_solve_L<NR, L>(vl, vd, vu, vb, vx, lane, xch);
            else                          f32::pcr_solve<NR, L>(vl, vd, vu, vb, vx, lane, xch);
#pragma unroll
            for (int j = 0; j < NR; j++) x[base + NR * lane + j] = vx[j];
        } else {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int64_t o = base + ln + W * j;
                vl[j] = ld[o]; vd[j] = d[o]; vu[j] = ud[o]; vb[j] = b[o];
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
    int64_t blocks = (S + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks), block(256);
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
