// Stand-alone likelihood kernels for the unfused (drop-in) path: log10 clamp (probs.py:64-75),
// squared-error accumulation (probs.py:20-47) and the curve-order reduction of the fused path.
// All three are HBM-bandwidth bound; arithmetic is fp64.
#include <float.h>
#include <math.h>

#include "trpl_common.hpp"

namespace trpl {

// x <- log10(max(x, mn)) in the buffer's dtype.  For float the clamp value is stored as
// (float)mn first -- 0.0f for mn = DBL_MIN, hence -inf -- exactly like an assignment into the
// reference's float32 array (probs.py:72-75).
template <typename T>
__global__ void __launch_bounds__(256) log10_clamp_kernel(T *x, int64_t rows, int64_t cols, int64_t ld, double mn)
{
    const int64_t n = rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / cols, c = idx - r * cols;
        T *p = x + r * ld + c;
        T v = *p;
        if ((double)v < mn) v = (T)mn;
        *p = (T)log10((double)v);
    }
}

hipError_t launch_log10_clamp(void *x, int elem_bytes, int64_t rows, int64_t cols, int64_t ld, double mn,
                              hipStream_t stream)
{
    const int64_t n = rows * cols;
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (elem_bytes == 4)
        hipLaunchKernelGGL(log10_clamp_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (float *)x, rows,
                           cols, ld, mn);
    else
        hipLaunchKernelGGL(log10_clamp_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, stream, (double *)x,
                           rows, cols, ld, mn);
    return hipGetLastError();
}

// P[j] -= sum_i (pl[j][i] + mag[j] - values[i])^2, accumulated in index order in fp64 so the
// result is bit-identical to the reference's serial loop (probs.py:32-44).  A wavefront owns 64
// rows: it loads [64 rows x 64 columns] tiles with the lanes along the columns (256/512-byte
// contiguous segments), transposes through LDS, and each lane then adds its own row's 64 values
// in order.
template <typename T>
__global__ void __launch_bounds__(64) sse_accumulate_kernel(double *P, const T *pl, int64_t rows, int64_t n_obs,
                                                            int64_t ld, const double *values, const double *mag)
{
    __shared__ double tile[64][65];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int64_t myrow = row0 + lane;
    const bool live = myrow < rows;
    double acc = 0.0;
    for (int64_t c0 = 0; c0 < n_obs; c0 += 64) {
        const int64_t col = c0 + lane;
        const int ncol = (int)((n_obs - c0) < 64 ? (n_obs - c0) : 64);
        const double val = col < n_obs ? values[col] : 0.0;
#pragma unroll 8
        for (int r = 0; r < 64; r++) {
            const int64_t rr = row0 + r;
            double v = 0.0;
            if (rr < rows && col < n_obs) {
                double e = (double)pl[rr * ld + col] + mag[rr];     // probs.py:33
                e -= val;                                           // :37
                v = e * e;                                          // :39
            }
            tile[r][lane] = v;
        }
        __syncthreads();
        for (int k = 0; k < ncol; k++) acc += tile[lane][k];        // :41, in index order
        __syncthreads();
    }
    if (live) P[myrow] += (0.0 - acc);                              // :44, :57-60
}

hipError_t launch_sse_accumulate(double *P, const void *pl, int elem_bytes, int64_t rows, int64_t n_obs,
                                 int64_t ld, const double *values, const double *mag, hipStream_t stream)
{
    if (rows <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((rows + 63) / 64);
    if (elem_bytes == 4)
        hipLaunchKernelGGL(sse_accumulate_kernel<float>, dim3(blocks), dim3(64), 0, stream, P, (const float *)pl,
                           rows, n_obs, ld, values, mag);
    else
        hipLaunchKernelGGL(sse_accumulate_kernel<double>, dim3(blocks), dim3(64), 0, stream, P, (const double *)pl,
                           rows, n_obs, ld, values, mag);
    return hipGetLastError();
}

// P[s] -= sse[c][s] for c = 0..C-1 in curve order (the order bayeslib.simulate calls prob in,
// bayeslib.py:117,:195).
__global__ void __launch_bounds__(256) reduce_curves_kernel(double *P, const double *sse, int64_t S, int C)
{
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (int64_t)gridDim.x * blockDim.x) {
        double p = P[s];
        for (int c = 0; c < C; c++) p += (0.0 - sse[(int64_t)c * S + s]);
        P[s] = p;
    }
}

hipError_t launch_reduce_curves(double *P, const double *sse, int64_t S, int C, hipStream_t stream)
{
    if (S <= 0) return hipSuccess;
    int64_t blocks = (S + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_curves_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, P, sse, S, C);
    return hipGetLastError();
}

}  // namespace trpl
