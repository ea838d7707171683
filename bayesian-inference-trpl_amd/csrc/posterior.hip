// Posterior core on the device: the consumer of the likelihood vector P[S].
// Reference (host numpy): Visualization/utils.py normalize :157-166, w_mean/w_variance/w_skew/w_kurtosis
// :197-220, covariance :222-227, marginalize_1D :239-262, marginalize_2D :264-285; tempering LL / tf at
// marginalization_visual.py:589-591.
//
// All kernels are streaming reductions over S samples (HBM-bound: 8 B of likelihood + 8 B per parameter
// column per sample), written as fixed-grid block partials + a one-block final pass so that sums are
// deterministic and a multi-GPU caller can all-reduce the same small result arrays.  Histograms bin
// against the reference's own edge values lo + (hi - lo) * k / bins with numpy's rules (left-closed
// bins, last bin closed on the right, out-of-range and NaN samples dropped); bin sums use fp64 atomics
// (LDS per block, then global), so their association is not fixed (~1e-16 relative).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "trpl_common.hpp"

namespace trpl {
namespace post {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;
constexpr int kMaxDim = 16;
constexpr int kLdsBins = 4096;

__device__ __forceinline__ double wave_add(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
// block-wide sum / max of one value per thread (kThreads = 4 waves); result valid in thread 0
template <bool MAX>
__device__ __forceinline__ double block_reduce(double v, double *sm)
{
    v = MAX ? wave_max(v) : wave_add(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    double r = sm[0];
#pragma unroll
    for (int i = 1; i < kThreads / 64; i++) r = MAX ? fmax(r, sm[i]) : r + sm[i];
    return r;
}

// ---- weights: q = LL / tf; W = exp(q - nanmax(q) + c_up - c_size); W /= nansum(W) ----
__global__ void __launch_bounds__(kThreads) nanmax_partial(const double *LL, int64_t S, double tf, double *part)
{
    __shared__ double sm[kThreads / 64];
    double m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < S; i += (int64_t)gridDim.x * kThreads) {
        const double q = LL[i] / tf;
        m = fmax(m, q);                      // fmax ignores NaN operands, like np.nanmax
    }
    m = block_reduce<true>(m, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = m;
}
// part is [gridDim.y][nb][ncol]; block (c, y) reduces column c of slab y over the nb block partials:
// thread t takes b = t, t + 256, ... in order, then the fixed block tree -- deterministic
__global__ void __launch_bounds__(kThreads) final_reduce(const double *part, int nb, int ncol, bool is_max, double *out)
{
    __shared__ double sm[kThreads / 64];
    const int c = blockIdx.x;
    const double *p = part + (int64_t)blockIdx.y * nb * ncol;
    double r = is_max ? -INFINITY : 0.0;
    for (int b = threadIdx.x; b < nb; b += kThreads) r = is_max ? fmax(r, p[(int64_t)b * ncol + c]) : r + p[(int64_t)b * ncol + c];
    r = is_max ? block_reduce<true>(r, sm) : block_reduce<false>(r, sm);
    if (threadIdx.x == 0) out[(int64_t)blockIdx.y * ncol + c] = r;
}
__global__ void __launch_bounds__(kThreads) weights_partial(const double *LL, int64_t S, double tf, const double *mx,
                                                            double c_up, double c_size, double *W, double *part)
{
    __shared__ double sm[kThreads / 64];
    const double m = mx[0];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < S; i += (int64_t)gridDim.x * kThreads) {
        const double q = LL[i] / tf;
        const double w = exp(((q - m) + c_up) - c_size);      // utils.py:164, in its order of operations
        W[i] = w;
        if (w == w) acc += w;                                 // np.nansum
    }
    acc = block_reduce<false>(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(kThreads) scale_kernel(double *W, int64_t S, const double *sum)
{
    const double s = sum[0];
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < S; i += (int64_t)gridDim.x * kThreads)
        W[i] = W[i] / s;                                      // utils.py:165
}

// ---- moments pass 1: sum w, sum w^2, sum w v_d  (V is [D][S]: one contiguous column per parameter) ----
__global__ void __launch_bounds__(kThreads) moments1_partial(const double *V, const double *W, int64_t S, int D, double *part)
{
    __shared__ double sm[kThreads / 64];
    double sw = 0.0, sw2 = 0.0, sv[kMaxDim];
#pragma unroll
    for (int d = 0; d < kMaxDim; d++) sv[d] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < S; i += (int64_t)gridDim.x * kThreads) {
        const double w = W[i];
        sw += w;
        sw2 += w * w;
        // unconditional loads (a column index beyond D re-reads the last column, whose sum is never stored): no
        // branch between the loads, all 1 + kMaxDim of a sample are in flight together
#pragma unroll
        for (int d = 0; d < kMaxDim; d++) sv[d] += V[(int64_t)(d < D ? d : D - 1) * S + i] * w;
    }
    double *row = part + (int64_t)blockIdx.x * (2 + D);
    double r = block_reduce<false>(sw, sm);
    if (threadIdx.x == 0) row[0] = r;
    r = block_reduce<false>(sw2, sm);
    if (threadIdx.x == 0) row[1] = r;
#pragma unroll
    for (int d = 0; d < kMaxDim; d++) {
        if (d < D) {
            r = block_reduce<false>(sv[d], sm);
            if (threadIdx.x == 0) row[2 + d] = r;
        }
    }
}
// ---- moments pass 2, ONE pass over the columns: sum w (v_d - m_d)(v_e - m_e) for e >= d (the matrix is symmetric:
//      the partial of (d, e) is also stored as (e, d), the same bits) and the third and fourth central sums of
//      every v_d;  sums[0] = sum w, sums[2 + d] = sum w v_d from pass 1.  DM >= D is the compiled size: columns
//      D .. DM-1 are never loaded (their accumulators add zeros).  DM (DM + 1) / 2 + 2 DM fp64 accumulators per
//      thread -- 117 at DM = 13 -- is one wave per SIMD, which a thread's 14 independent loads per sample make up
//      for: the first version read every column once per tile of four rows (five reads of V in all). ----
template <int DM>
__global__ void __launch_bounds__(kThreads, 1) moments2_partial(const double *V, const double *W, int64_t S, int D,
                                                                const double *sums, const double *mean_in, double *part)
{
    __shared__ double sm[kThreads / 64];
    const double sw = sums[0];
    double mean[DM];
#pragma unroll
    for (int e = 0; e < DM; e++) mean[e] = e < D ? (mean_in ? mean_in[e] : sums[2 + e] / sw) : 0.0;      // np.average
    double c[DM * (DM + 1) / 2], m3[DM], m4[DM];
#pragma unroll
    for (int k = 0; k < DM * (DM + 1) / 2; k++) c[k] = 0.0;
#pragma unroll
    for (int d = 0; d < DM; d++) { m3[d] = 0.0; m4[d] = 0.0; }
    // the loads of the next kAhead samples (1 + D each) are in flight while the current sample's ~3 DM^2 / 2
    // operations run: with one wave per SIMD nothing else hides the HBM latency (no prefetch 1.8 ms, one sample
    // ahead 0.8 ms at S = 16.7 M, D = 13)
    constexpr int kAhead = DM <= 8 ? 1 : 2;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double w[kAhead + 1], x[kAhead + 1][DM];
    auto fetch = [&](int64_t idx, int slot) {
        // clamped, unconditional loads (no branch between them): a sample index beyond S re-reads the last sample
        // with weight 0, a column beyond D the last column, which xc[] then ignores
        const int64_t at = idx < S ? idx : S - 1;
        w[slot] = idx < S ? W[at] : 0.0;
#pragma unroll
        for (int e = 0; e < DM; e++) x[slot][e] = V[(int64_t)(e < D ? e : D - 1) * S + at];
    };
#pragma unroll
    for (int q = 0; q < kAhead; q++) fetch(i + q * stride, q);
    for (; i < S; i += stride) {
        fetch(i + kAhead * stride, kAhead);
        double xc[DM];
#pragma unroll
        for (int e = 0; e < DM; e++) xc[e] = e < D ? x[0][e] - mean[e] : 0.0;
        const double w0 = w[0];
        int k = 0;
#pragma unroll
        for (int d = 0; d < DM; d++) {
            const double xd = xc[d];
#pragma unroll
            for (int e = d; e < DM; e++) c[k++] += (xd * xc[e]) * w0;
            const double x2 = xd * xd;
            m3[d] += (x2 * xd) * w0;
            m4[d] += (x2 * x2) * w0;
        }
#pragma unroll
        for (int q = 0; q < kAhead; q++) {
            w[q] = w[q + 1];
#pragma unroll
            for (int e = 0; e < DM; e++) x[q][e] = x[q + 1][e];
        }
    }
    // part is [D][gridDim.x][D + 2]
    int k = 0;
#pragma unroll
    for (int d = 0; d < DM; d++) {
#pragma unroll
        for (int e = d; e < DM; e++, k++) {
            if (e < D) {                                             // wave-uniform (d <= e < D)
                const double r = block_reduce<false>(c[k], sm);
                if (threadIdx.x == 0) {
                    part[((int64_t)d * gridDim.x + blockIdx.x) * (D + 2) + e] = r;
                    part[((int64_t)e * gridDim.x + blockIdx.x) * (D + 2) + d] = r;
                }
            }
        }
        if (d < D) {
            double r = block_reduce<false>(m3[d], sm);
            if (threadIdx.x == 0) part[((int64_t)d * gridDim.x + blockIdx.x) * (D + 2) + D] = r;
            r = block_reduce<false>(m4[d], sm);
            if (threadIdx.x == 0) part[((int64_t)d * gridDim.x + blockIdx.x) * (D + 2) + D + 1] = r;
        }
    }
}

// ---- histograms ----
__device__ __forceinline__ double edge(double lo, double hi, int k, int bins) { return lo + ((hi - lo) * k) / bins; }
// numpy's bin of x against edges e_0..e_bins (e_k as the reference builds them, utils.py:243-244): -1 = dropped.
// tab (LDS, bins + 1 entries) holds those edges when the axis is small enough, `scale` = bins / (hi - lo):
// a multiply finds the candidate bin, comparisons against the exact edges settle it.
constexpr int kMaxAxisTab = 1024;
__device__ __forceinline__ int bin_of(double x, double lo, double hi, int bins, double scale, const double *tab)
{
    if (!(x >= lo && x <= hi)) return -1;                       // also drops NaN
    int k = (int)((x - lo) * scale);
    k = k < 0 ? 0 : (k > bins - 1 ? bins - 1 : k);
    if (tab) {
        while (k > 0 && x < tab[k]) k--;
        while (k < bins - 1 && x >= tab[k + 1]) k++;
    } else {
        while (k > 0 && x < edge(lo, hi, k, bins)) k--;
        while (k < bins - 1 && x >= edge(lo, hi, k + 1, bins)) k++;
    }
    return k;
}
__global__ void __launch_bounds__(kThreads) hist_kernel(const double *x, const double *y, const double *W, int64_t S,
                                                        double xlo, double xhi, int xb, double ylo, double yhi, int yb,
                                                        double *out)
{
    __shared__ double bins[kLdsBins];
    __shared__ double xtab[kMaxAxisTab + 1], ytab[kMaxAxisTab + 1];
    const double *xt = xb <= kMaxAxisTab ? xtab : nullptr, *yt = (y && yb <= kMaxAxisTab) ? ytab : nullptr;
    if (xt) for (int k = threadIdx.x; k <= xb; k += kThreads) xtab[k] = edge(xlo, xhi, k, xb);
    if (yt) for (int k = threadIdx.x; k <= yb; k += kThreads) ytab[k] = edge(ylo, yhi, k, yb);
    const double xs = xb / (xhi - xlo), ys = y ? yb / (yhi - ylo) : 0.0;
    __syncthreads();
    const int nb = y ? xb * yb : xb;
    const bool use_lds = nb <= kLdsBins;
    // few bins mean many lanes adding to the same LDS word: keep up to 16 replicas, one per lane group
    int rep = 1;
    while (rep < 16 && 2 * rep * nb <= kLdsBins) rep *= 2;
    const int my = (threadIdx.x & (rep - 1)) * nb;
    if (use_lds) {
        for (int k = threadIdx.x; k < rep * nb; k += kThreads) bins[k] = 0.0;
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < S; i += (int64_t)gridDim.x * kThreads) {
        // all of a sample's loads first, unconditionally: a load behind a data-dependent branch waits for the
        // previous one (the coordinates and the weight are three independent streams)
        const double xi = x[i], yi = y ? y[i] : 0.0, w = W ? W[i] : 1.0;
        int k = bin_of(xi, xlo, xhi, xb, xs, xt);
        if (k >= 0 && y) {
            const int ky = bin_of(yi, ylo, yhi, yb, ys, yt);
            k = ky < 0 ? -1 : k * yb + ky;                      // [x bin][y bin], like np.histogram2d
        }
        if (k < 0) continue;
        if (use_lds) atomicAdd(&bins[my + k], w);
        else atomicAdd(&out[k], w);
    }
    if (use_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < nb; k += kThreads) {
            double t = 0.0;
            for (int r = 0; r < rep; r++) t += bins[r * nb + k];
            if (t != 0.0) atomicAdd(&out[k], t);
        }
    }
}

inline int grid_for(int64_t S)
{
    int64_t nb = (S + kThreads - 1) / kThreads;
    return (int)(nb < 1 ? 1 : (nb > kMaxBlocks ? kMaxBlocks : nb));
}

}  // namespace post

size_t posterior_workspace_bytes(int D) { return sizeof(double) * ((size_t)post::kMaxBlocks * (size_t)(D > 0 ? D : 1) * (D + 2) + 64); }

__global__ void copy2_kernel(const double *a, const double *b, double *out) { out[0] = a[0]; out[1] = b[0]; }

// W[S] <- normalised posterior weights of LL[S] tempered by tf.  ws: >= posterior_workspace_bytes(1).
// stats (nullable, device): {nanmax(LL/tf), nansum of the unnormalised weights} -- what a caller that
// holds only a shard of the samples needs to renormalise across shards.
hipError_t launch_posterior_weights(const double *LL, int64_t S, double tf, double *W, double *stats, double *ws,
                                    hipStream_t st)
{
    using namespace post;
    if (S <= 0) return hipSuccess;
    const int nb = grid_for(S);
    double *part = ws, *mx = ws + kMaxBlocks, *sum = mx + 1;
    const double c_up = 1000.0 * log(2.0), c_size = log((double)S);                   // utils.py:164
    hipLaunchKernelGGL(nanmax_partial, dim3(nb), dim3(kThreads), 0, st, LL, S, tf, part);
    hipLaunchKernelGGL(final_reduce, dim3(1, 1), dim3(kThreads), 0, st, part, nb, 1, true, mx);
    hipLaunchKernelGGL(weights_partial, dim3(nb), dim3(kThreads), 0, st, LL, S, tf, mx, c_up, c_size, W, part);
    hipLaunchKernelGGL(final_reduce, dim3(1, 1), dim3(kThreads), 0, st, part, nb, 1, false, sum);
    hipLaunchKernelGGL(scale_kernel, dim3(nb), dim3(kThreads), 0, st, W, S, sum);
    if (stats) hipLaunchKernelGGL(copy2_kernel, dim3(1), dim3(1), 0, st, mx, sum, stats);
    return hipGetLastError();
}

// sums[2 + D]   = {sum w, sum w^2, sum w v_d}
// central[D][D+2] = {sum w (v_d - m_d)(v_e - m_e) for e < D, sum w (v_d - m_d)^3, sum w (v_d - m_d)^4}
// ws: >= posterior_workspace_bytes(D).
// mean_in (nullable, device, [D]): centre about these means instead of this call's own (sharded callers).
hipError_t launch_posterior_moments(const double *V, const double *W, int64_t S, int D, const double *mean_in, double *sums,
                                    double *central, double *ws, hipStream_t st)
{
    using namespace post;
    if (D < 1 || D > kMaxDim) return hipErrorInvalidValue;
    if (S <= 0) return hipSuccess;
    const int nb = grid_for(S);
    hipLaunchKernelGGL(moments1_partial, dim3(nb), dim3(kThreads), 0, st, V, W, S, D, ws);
    hipLaunchKernelGGL(final_reduce, dim3(2 + D, 1), dim3(kThreads), 0, st, ws, nb, 2 + D, false, sums);
    if (D <= 4)       hipLaunchKernelGGL(moments2_partial<4>, dim3(nb), dim3(kThreads), 0, st, V, W, S, D, sums, mean_in, ws);
    else if (D <= 8)  hipLaunchKernelGGL(moments2_partial<8>, dim3(nb), dim3(kThreads), 0, st, V, W, S, D, sums, mean_in, ws);
    else if (D <= 13) hipLaunchKernelGGL(moments2_partial<13>, dim3(nb), dim3(kThreads), 0, st, V, W, S, D, sums, mean_in, ws);
    else              hipLaunchKernelGGL(moments2_partial<16>, dim3(nb), dim3(kThreads), 0, st, V, W, S, D, sums, mean_in, ws);
    hipLaunchKernelGGL(final_reduce, dim3(D + 2, D), dim3(kThreads), 0, st, ws, nb, D + 2, false, central);
    return hipGetLastError();
}

// out[xb] (y == nullptr) or out[xb][yb] += weighted counts; out must be zeroed by the caller
hipError_t launch_posterior_hist(const double *x, const double *y, const double *W, int64_t S, double xlo, double xhi,
                                 int xb, double ylo, double yhi, int yb, double *out, hipStream_t st)
{
    using namespace post;
    if (S <= 0) return hipSuccess;
    int nb = grid_for(S);
    if (nb > 768) nb = 768;                 // 3 blocks per CU (48 KB of LDS each): enough loads in flight, few bin flushes
    hipLaunchKernelGGL(hist_kernel, dim3(nb), dim3(kThreads), 0, st, x, y, W, S, xlo, xhi, xb, ylo, yhi, yb, out);
    return hipGetLastError();
}

}  // namespace trpl
