// Stand-alone likelihood kernels for the unfused (drop-in) path: log10 clamp (probs.py:64-75),
// squared-error accumulation (probs.py:20-47) and the curve-order reduction of the fused path.
// All three are HBM-bandwidth bound; arithmetic is fp64.
#include <float.h>
#include <math.h>

#include "trpl_common.hpp"

namespace trpl {

// x <- log10(max(x, mn)) in the buffer's dtype.  For float the clamp value is stored as
// (float)mn first -- 0.0f for mn = DBL_MIN, hence -inf -- exactly like an assignment into the
// reference's float32 array (probs.py:72-75).  The logarithm is evaluated in fp64 and rounded to
// the buffer's dtype (the CPU oracle's definition).
template <typename T>
__device__ __forceinline__ T log10_clamp_one(T v, double mn)
{
    if ((double)v < mn) v = (T)mn;
    return (T)log10((double)v);
}

// contiguous buffer (ld == cols): 16 bytes per lane per access, no index arithmetic
template <typename T>
__global__ void __launch_bounds__(256) log10_clamp_flat_kernel(T *x, int64_t n, double mn)
{
    constexpr int V = 16 / sizeof(T);
    struct alignas(16) Vec { T v[V]; };
    const int64_t nvec = n / V;
    Vec *xv = reinterpret_cast<Vec *>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        Vec t = xv[i];
#pragma unroll
        for (int k = 0; k < V; k++) t.v[k] = log10_clamp_one<T>(t.v[k], mn);
        xv[i] = t;
    }
    const int64_t tail = nvec * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tail < n) x[tail] = log10_clamp_one<T>(x[tail], mn);      // n % V < V <= 4 elements, first lanes of block 0
}

// padded rows (ld > cols): rows over blockIdx.y, columns strided by the block
template <typename T>
__global__ void __launch_bounds__(256) log10_clamp_rows_kernel(T *x, int64_t rows, int64_t cols, int64_t ld, double mn)
{
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        T *row = x + r * ld;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (int64_t)gridDim.x * blockDim.x)
            row[c] = log10_clamp_one<T>(row[c], mn);
    }
}

template <typename T>
static hipError_t launch_log10_clamp_t(T *x, int64_t rows, int64_t cols, int64_t ld, double mn, hipStream_t stream)
{
    if (ld == cols && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int64_t n = rows * cols;
        int64_t blocks = (n / (16 / sizeof(T)) + 255) / 256;
        blocks = blocks < 1 ? 1 : (blocks > 256 * 32 ? 256 * 32 : blocks);
        hipLaunchKernelGGL(log10_clamp_flat_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, mn);
    } else {
        const unsigned bx = (unsigned)((cols + 255) / 256 > 64 ? 64 : (cols + 255) / 256);
        const unsigned by = (unsigned)(rows > 8192 ? 8192 : rows);
        hipLaunchKernelGGL(log10_clamp_rows_kernel<T>, dim3(bx, by), dim3(256), 0, stream, x, rows, cols, ld, mn);
    }
    return hipGetLastError();
}

hipError_t launch_log10_clamp(void *x, int elem_bytes, int64_t rows, int64_t cols, int64_t ld, double mn,
                              hipStream_t stream)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    return elem_bytes == 4 ? launch_log10_clamp_t<float>((float *)x, rows, cols, ld, mn, stream)
                           : launch_log10_clamp_t<double>((double *)x, rows, cols, ld, mn, stream);
}

// P[j] -= sum_i (pl[j][i] + mag[j] - values[i])^2, accumulated in index order in fp64 so the
// result is bit-identical to the reference's serial loop (probs.py:32-44).  The serial chain of
// n_obs dependent fp64 adds per row is the floor of this formulation (~0.2 ms at 80 001 columns);
// everything else is arranged around it: a wavefront owns R rows (R small when there are few
// rows, so that the chip is covered with workgroups), loads [R rows x 64 columns] tiles with the
// lanes along the columns (256/512-byte contiguous segments), squares the residuals in parallel,
// transposes through LDS, and lanes 0..R-1 add their row's 64 values in order while the loads of
// the next tile are already in flight.
template <typename T, int R>
__global__ void __launch_bounds__(64) sse_accumulate_kernel(double *P, const T *pl, int64_t rows, int64_t n_obs,
                                                            int64_t ld, const double *values, const double *mag)
{
    __shared__ double tile[R][65];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    double mg[R];
    const T *rowp[R];
#pragma unroll
    for (int r = 0; r < R; r++) {                 // rows past the end alias the last row: loaded, never stored
        const int64_t rr = row0 + r < rows ? row0 + r : rows - 1;
        mg[r] = mag[rr];
        rowp[r] = pl + rr * ld;
    }
    // D tiles of loads stay in flight per wave: a ring of RAW register tiles, statically indexed.
    // Loads are unconditional (column index clamped) and nothing depends on them until the slot is
    // consumed D tiles later, so the compiler can keep them outstanding (a bounds branch around
    // load+arithmetic made it wait for every single load: 0.1 % of HBM peak).
    constexpr int D = 32 / R;
    T raw[D][R];
    double vraw[D];
    auto issue = [&](int64_t c0, T (&v)[R], double &val) {
        int64_t col = c0 + lane;
        col = col < n_obs ? col : n_obs - 1;
        val = values[col];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = rowp[r][col];
    };
    double acc = 0.0;
    if (n_obs > 0) {
#pragma unroll
        for (int d = 0; d < D; d++) issue((int64_t)d * 64, raw[d], vraw[d]);
    }
    for (int64_t c0 = 0; c0 < n_obs; c0 += 64 * D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int64_t cc = c0 + (int64_t)d * 64;
            if (cc >= n_obs) break;                                // wave-uniform
#pragma unroll
            for (int r = 0; r < R; r++) {
                double e = (double)raw[d][r] + mg[r];              // probs.py:33
                e -= vraw[d];                                      // :37
                tile[r][lane] = e * e;                             // :39
            }
            issue(cc + 64 * D, raw[d], vraw[d]);                   // refill this slot D tiles ahead
            const int ncol = (int)((n_obs - cc) < 64 ? (n_obs - cc) : 64);
            if (lane < R) {
                const double *t = tile[lane];
                if (ncol == 64) {
#pragma unroll 16
                    for (int k = 0; k < 64; k++) acc += t[k];      // :41, in index order
                } else {
                    for (int k = 0; k < ncol; k++) acc += t[k];    // columns >= n_obs are never read
                }
            }
        }
    }
    if (lane < R && row0 + lane < rows) P[row0 + lane] += (0.0 - acc);   // :44, :57-60
}

template <typename T>
static hipError_t launch_sse_t(double *P, const T *pl, int64_t rows, int64_t n_obs, int64_t ld, const double *values,
                               const double *mag, hipStream_t stream)
{
#define TRPL_SSE(RR)                                                                                              \
    hipLaunchKernelGGL((sse_accumulate_kernel<T, RR>), dim3((unsigned)((rows + RR - 1) / RR)), dim3(64), 0, stream, \
                       P, pl, rows, n_obs, ld, values, mag)
    if (rows <= 4 * 1024) TRPL_SSE(4);               // >= rows/4 workgroups: cover the 256 CUs
    else if (rows <= 16 * 1024) TRPL_SSE(8);
    else TRPL_SSE(16);
#undef TRPL_SSE
    return hipGetLastError();
}

hipError_t launch_sse_accumulate(double *P, const void *pl, int elem_bytes, int64_t rows, int64_t n_obs,
                                 int64_t ld, const double *values, const double *mag, hipStream_t stream)
{
    if (rows <= 0) return hipSuccess;
    return elem_bytes == 4 ? launch_sse_t<float>(P, (const float *)pl, rows, n_obs, ld, values, mag, stream)
                           : launch_sse_t<double>(P, (const double *)pl, rows, n_obs, ld, values, mag, stream);
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

// ---------------------------------------------------------------------------------------------
// Likelihood of PL rows that are already in HBM (the output of trpl_solve_pl_dev) against one set of
// observations, in ONE pass: re-dimensionalised PL -> optional self-normalisation -> clamp -> log10
// (bayeslib.py:150-157) -> observations on the grid or linearly interpolated between the bracketing
// grid points (bayeslib.py:184-191) -> squared error (probs.py:29-44).  This is what lets one solve
// serve several experiments (the reference's loop order curves -> blocks -> experiments) without the
// PL matrix ever crossing PCIe.  One 256-thread block per row (the reference's 1024-row blocks would
// otherwise leave most of the chip idle behind the fp64 log10), threads stride over the observations,
// fixed-order block reduction at the end: the sum is associated differently from probs.prob's serial loop (~1e-16);
// trpl_log10_clamp + trpl_sse_accumulate remain the bit-exact pair.  HBM-bound: n_obs (or 2 n_obs,
// off-grid) PL elements per row.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ double log_pl(T v, T v0, bool normalize, bool f32_staging)
{
    if (f32_staging) {                       // the reference's float32 plI buffer (bayeslib.py:137)
        float f = (float)v;
        if (normalize) f = f / (float)v0;
        if ((double)f < DBL_MIN) f = (float)DBL_MIN;
        return (double)(float)log10((double)f);
    }
    double d = (double)v;
    if (normalize) d = d / (double)v0;
    if (d < DBL_MIN) d = DBL_MIN;
    return log10(d);
}

template <typename T>
__global__ void __launch_bounds__(256) pl_loglik_kernel(const T *pl, int64_t rows, int64_t ld, const double *obs,
                                                        const int32_t *obs_hi, const double *obs_dx, const double *obs_h,
                                                        int64_t n_obs, const double *mag, const int32_t *status,
                                                        double *P, double *sse_out, uint32_t flags)
{
    __shared__ double part[4];
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x;
    const T *r = pl + row * ld;
    const bool normalize = (flags & 0x4u) != 0, f32 = (flags & 0x2u) != 0 || sizeof(T) == 4;
    const T v0 = r[0];
    const double m = mag[row];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n_obs; i += 256) {
        double y;
        if (obs_hi) {
            const int64_t hi = obs_hi[i];
            const double lg_hi = log_pl<T>(r[hi], v0, normalize, f32), lg_lo = log_pl<T>(r[hi - 1], v0, normalize, f32);
            const double dy = f32 ? (double)((float)lg_hi - (float)lg_lo) : lg_hi - lg_lo;
            y = (dy / obs_h[i]) * obs_dx[i] + lg_lo;            // scipy interp1d: slope * (x - x_lo) + y_lo
        } else {
            y = log_pl<T>(r[i], v0, normalize, f32);
        }
        double err = y + m;
        err -= obs[i];
        acc += err * err;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        acc = ((part[0] + part[1]) + part[2]) + part[3];
        // a system flagged non-converged (its PL is NaN from that step on) scores +inf, like the fused path
        if ((status && status[row] != 0) || !(acc == acc)) acc = INFINITY;
        if (sse_out) sse_out[row] = acc;
        if (P) P[row] -= acc;                                   // probs.py:44,:60
    }
}

hipError_t launch_pl_loglik(const void *pl, int elem_bytes, int64_t rows, int64_t ld, const double *obs,
                            const int32_t *obs_hi, const double *obs_dx, const double *obs_h, int64_t n_obs,
                            const double *mag, const int32_t *status, double *P, double *sse_out, uint32_t flags,
                            hipStream_t stream)
{
    if (rows <= 0) return hipSuccess;
    const dim3 grid((unsigned)rows), block(256);
    if (elem_bytes == 4)
        hipLaunchKernelGGL(pl_loglik_kernel<float>, grid, block, 0, stream, (const float *)pl, rows, ld, obs, obs_hi, obs_dx,
                           obs_h, n_obs, mag, status, P, sse_out, flags);
    else
        hipLaunchKernelGGL(pl_loglik_kernel<double>, grid, block, 0, stream, (const double *)pl, rows, ld, obs, obs_hi, obs_dx,
                           obs_h, n_obs, mag, status, P, sse_out, flags);
    return hipGetLastError();
}

// ---- host-side time interpolation of the unfused call sequence (bayeslib.py:184-191) ----
// out[r][i] = ((pl[r][hi_i] - pl[r][hi_i - 1]) / h_i) * dx_i + pl[r][hi_i - 1]: the arithmetic of scipy's interp1d / griddata as
// the reference applies it row by row -- the difference in the matrix's own type (float32 for the reference's buffer), then
// float64, two roundings, no fused multiply-add (this translation unit is compiled with -ffp-contract=off) -- bit for bit what
// driver.interp_rows computed with NumPy, without the interpreter lock: the worker threads of driver.simulate run it side by side
// (NumPy's fancy indexing serialised them: 0.21 s of a 0.60 s task, profiles/r6_levelb_task_phases.txt).  Plain host code.
template <typename T>
static void interp_rows_host(const T *pl, int64_t rows, int64_t ld, const int32_t *hi, const double *dx, const double *h,
                             int64_t n_obs, double *out, int64_t out_ld)
{
    for (int64_t r = 0; r < rows; r++) {
        const T *p = pl + r * ld;
        double *o = out + r * out_ld;
        for (int64_t i = 0; i < n_obs; i++) {
            const T lo = p[hi[i] - 1];
            const T d = p[hi[i]] - lo;
            const double slope = (double)d / h[i];
            const double prod = slope * dx[i];
            o[i] = prod + (double)lo;
        }
    }
}

void interp_rows_any(const void *pl, int elem_bytes, int64_t rows, int64_t ld, const int32_t *hi, const double *dx,
                     const double *h, int64_t n_obs, double *out, int64_t out_ld)
{
    if (elem_bytes == 4) interp_rows_host<float>((const float *)pl, rows, ld, hi, dx, h, n_obs, out, out_ld);
    else interp_rows_host<double>((const double *)pl, rows, ld, hi, dx, h, n_obs, out, out_ld);
}

}  // namespace trpl
