// Parameter-box sampler on the device: bayeslib.random_grid (bayeslib.py:18-32)
// and the make_grid overrides (:67-75), drawing the SAME stream as the reference's
// numpy.random.seed(seed) + np.random.uniform: MT19937 (Matsumoto & Nishimura 1998, the generator of
// numpy's legacy RandomState), seeded by init_genrand, 53-bit doubles (a >> 5, b >> 6), value =
// low + (high - low) * u with separate multiply and add, columns drawn one after the other and fixed
// columns (min == max) drawing nothing.  Linear columns are bit-identical to the reference's; log
// columns are 10 ** uniform(log10 lo, log10 hi) with the device's pow (<= 1 ulp from the host libm's).
//
// The generator is a serial recurrence, so ONE workgroup walks the stream: the 624-word state lives in
// LDS and is regenerated in three dependency phases (words 0-226 need only old words, 227-453 the first
// phase, 454-623 the second), each block of 624 words then yields 312 doubles in parallel.  S = 65 536
// x 10 random columns is 2 100 regenerations (~10 ms); it replaces the S x 13 host-to-device copy.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "trpl_common.hpp"

namespace trpl {
namespace smp {

constexpr int kN = 624, kM = 397, kThreads = 256;
constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrixA = 0x9908b0dfu;

__device__ __forceinline__ uint32_t twist(uint32_t cur, uint32_t nxt, uint32_t far_)
{
    const uint32_t y = (cur & kUpper) | (nxt & kLower);
    return far_ ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
}
__device__ __forceinline__ uint32_t temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

struct BoxArgs {
    double lo[16], hi[16];     // bounds as random_grid receives them (already unit-converted)
    int32_t do_log[16];
    int32_t ncol;
    uint32_t seed;
    uint32_t flags;            // bit 0: X[:,2] = X[:,3]; bit 1: X[:,6] = X[:,5]; bit 2: X[:,8] = X[:,7]  (bayeslib.py:67-75)
};

__global__ void __launch_bounds__(kThreads) sample_box_kernel(const BoxArgs a, int64_t S, double *X)
{
    __shared__ uint32_t mt[kN];
    __shared__ int colmap[16];
    __shared__ int nactive;
    const int tid = threadIdx.x;
    if (tid == 0) {                                            // init_genrand(seed)
        uint32_t v = a.seed;
        mt[0] = v;
        for (int i = 1; i < kN; i++) {
            v = 1812433253u * (v ^ (v >> 30)) + (uint32_t)i;
            mt[i] = v;
        }
        int n = 0;
        for (int c = 0; c < a.ncol; c++)
            if (a.lo[c] != a.hi[c]) colmap[n++] = c;
        nactive = n;
    }
    __syncthreads();
    // fixed columns (bayeslib.py:24-25)
    for (int c = 0; c < a.ncol; c++)
        if (a.lo[c] == a.hi[c])
            for (int64_t r = tid; r < S; r += kThreads) X[r * a.ncol + c] = a.lo[c];

    const int64_t total = (int64_t)nactive * S;               // doubles to draw, column after column
    for (int64_t base = 0; base < total; base += kN / 2) {
        // ---- regenerate the 624 words: three read / barrier / write phases ----
#pragma unroll
        for (int ph = 0; ph < 3; ph++) {
            const int k0 = ph * 227, k1 = ph == 2 ? kN : k0 + 227;
            const int k = k0 + tid;
            uint32_t nv = 0;
            if (k < k1) {
                const int far_ = k + kM < kN ? k + kM : k + kM - kN;
                nv = twist(mt[k], mt[k + 1 < kN ? k + 1 : 0], mt[far_]);
            }
            __syncthreads();
            if (k < k1) mt[k] = nv;
            __syncthreads();
        }
        // ---- 312 doubles from this block of words ----
        for (int t = tid; t < kN / 2; t += kThreads) {
            const int64_t k = base + t;
            if (k >= total) break;
            const uint32_t w0 = temper(mt[2 * t]) >> 5, w1 = temper(mt[2 * t + 1]) >> 6;
            const double u = ((double)w0 * 67108864.0 + (double)w1) / 9007199254740992.0;     // genrand_res53
            const int c = colmap[k / S];
            const int64_t r = k % S;
            double v;
            if (a.do_log[c]) {
                const double l = log10(a.lo[c]);
                v = pow(10.0, l + (log10(a.hi[c]) - l) * u);   // bayeslib.py:28
            } else {
                v = a.lo[c] + (a.hi[c] - a.lo[c]) * u;         // bayeslib.py:30
            }
            X[r * a.ncol + c] = v;
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // ---- make_grid's overrides (bayeslib.py:67-75) ----
    for (int64_t r = tid; r < S; r += kThreads) {
        double *row = X + r * a.ncol;
        if ((a.flags & 1u) && a.ncol > 3) row[2] = row[3];
        if ((a.flags & 2u) && a.ncol > 6) row[6] = row[5];
        if ((a.flags & 4u) && a.ncol > 8) row[8] = row[7];
    }
}

}  // namespace smp

hipError_t launch_sample_box(uint32_t seed, int64_t S, int ncol, const double *lo, const double *hi, const int32_t *do_log,
                             uint32_t flags, double *X, hipStream_t st)
{
    if (ncol < 1 || ncol > 16) return hipErrorInvalidValue;
    if (S <= 0) return hipSuccess;
    smp::BoxArgs a;
    for (int c = 0; c < 16; c++) {
        a.lo[c] = c < ncol ? lo[c] : 0.0;
        a.hi[c] = c < ncol ? hi[c] : 0.0;
        a.do_log[c] = c < ncol ? do_log[c] : 0;
    }
    a.ncol = ncol; a.seed = seed; a.flags = flags;
    hipLaunchKernelGGL(smp::sample_box_kernel, dim3(1), dim3(smp::kThreads), 0, st, a, S, X);
    return hipGetLastError();
}

}  // namespace trpl
