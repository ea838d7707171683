// Cost breakdown of ONE inner iteration of the fast stepper (development aid).  Runs the library's
// own loop body (assemble / residual / PCR / field, interleaved layout, L = 128) for a fixed number
// of iterations per wave, with the ablation switches of stepper_impl.hpp (-DTRPL_ABLATE=bits).
//   for a in 0 1 2 4 8 3 15; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast \
//       -DTRPL_ABLATE=$a -Ibayesian-inference-trpl_amd/csrc tools/iter_bench.hip -o /tmp/ib$a && /tmp/ib$a; done
#include "stepper_impl.hpp"
#include <cstdio>
#include <vector>
using namespace trpl;

template <int WPS>
__global__ void __launch_bounds__(64, WPS) kern(double* out, int iters)
{
    constexpr int L = 128, W = 64, NR = 2, LAY = 2;
    __shared__ __attribute__((aligned(16))) double lds[(160 * 1024 / (4 * WPS)) / 8 - 64];   // forces WPS waves/SIMD at most
    double* xch = lds;
    const int ln = threadIdx.x;
    MatPar mp = {3.8e-7, 1.1e-2, 1.9e-3, 2.2e-3, 4.4e-6, 3.2e-5, 3.2e-5, 1.3e-9, 1.3e-9, 2.0e4, 3.5e4, 4.5, 4.2e-9};
    double Nk[NR], Pk[NR], Ek[NR], bN[NR], bP[NR], bE[NR];
    for (int j = 0; j < NR; j++) {
        const int i = NR * ln + j;
        Nk[j] = 0.05 * exp(-0.09 * (i + 0.5)) + mp.N0; Pk[j] = Nk[j] + mp.P0; Ek[j] = 1e-4 * sin(0.1 * i);
        bN[j] = -Nk[j]; bP[j] = -Pk[j]; bE[j] = -Ek[j];
    }
    const double a0 = 137.0 / 60, TOL = 1e-7;
    int conv = 0;
    for (int it = 0; it < iters; it++) {
        double lo_[NR], dg[NR], up[NR], bb[NR], Ep[NR];
        shift_up1<LAY, NR, W>(Ek, Ep, ln);
        assemble<LAY, true, NR, W, L>(mp, a0, Nk, Pk, Ek, Ep, bN, lo_, dg, up, bb, ln);
        const bool okN = residual_below<LAY, NR, W>(lo_, dg, up, bb, Nk, TOL, ln);
        solve_lay<LAY, NR, W, L>(lo_, dg, up, bb, Nk, ln, xch);
        assemble<LAY, false, NR, W, L>(mp, a0, Nk, Pk, Ek, Ep, bP, lo_, dg, up, bb, ln);
        const bool okP = residual_below<LAY, NR, W>(lo_, dg, up, bb, Pk, TOL, ln);
        solve_lay<LAY, NR, W, L>(lo_, dg, up, bb, Pk, ln, xch);
        update_field<LAY, NR, W>(mp, a0, Nk, Pk, bE, Ek, ln);
        conv += (okN && okP) ? 1 : 0;
    }
    out[blockIdx.x * 64 + ln] = Nk[0] + Pk[1] + Ek[0] + conv;
}

template <int WPS>
void run(int iters)
{
    const int nblk = 256 * 4 * WPS;
    double* out; hipMalloc(&out, (size_t)nblk * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<WPS>, dim3(nblk), dim3(64), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<WPS>, dim3(nblk), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double it_per_s = (double)nblk * iters / (ms * 1e-3);
    printf("ABLATE=%d waves/SIMD=%d: %.3e iterations/s  -> %.0f cycles per iteration per SIMD @2.4GHz\n", TRPL_ABLATE, WPS,
           it_per_s, 2.4e9 * 1024 / it_per_s);
    hipFree(out);
}
int main() { run<1>(2000); run<2>(2000); run<3>(2000); return 0; }
