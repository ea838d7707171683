// Instruction-throughput microbenchmark (development aid): cycles per wave-instruction on one
// SIMD with 1, 2, 3 resident waves, measured with s_memtime around long unrolled independent
// sequences.  Build+run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int OP>
__global__ void k(double* out, long long* cyc, int iters)
{
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
    double c = 1.000001, e = 0.5;
    int lane = threadIdx.x & 63;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            double& x = a[r & 7];
            if constexpr (OP == 0) x = __builtin_fma(x, c, e);
            if constexpr (OP == 1) x = x * c;
            if constexpr (OP == 2) x = x + c;
            if constexpr (OP == 3) x = __builtin_amdgcn_rcp(x);
            if constexpr (OP == 4) { int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x134, 0xF, 0xF, false); x = __hiloint2double(__double2hiint(x), lo); }
            if constexpr (OP == 5) { int lo = __builtin_amdgcn_ds_bpermute(((lane + 2) & 63) << 2, __double2loint(x)); x = __hiloint2double(__double2hiint(x), lo); }
            if constexpr (OP == 6) { auto q = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2hiint(x), false, false); x = __hiloint2double((int)q[1], (int)q[0]); }
            if constexpr (OP == 7) { float f = (float)x; f = __builtin_amdgcn_rcpf(f); x = (double)f; }
            if constexpr (OP == 8) { int lo = __double2loint(x); lo = (lane & 1) ? lo : __double2hiint(x); x = __hiloint2double(__double2hiint(x), lo); }
            if constexpr (OP == 13) a[0] = __builtin_fma(a[0], c, e);                         // ONE dependent chain
            if constexpr (OP == 14) a[r & 1] = __builtin_fma(a[r & 1], c, e);                 // two chains
            if constexpr (OP == 15) { a[0] = __builtin_amdgcn_rcp(a[0]); a[1] = __builtin_fma(a[1], c, e); a[2] = __builtin_fma(a[2], c, e); a[3] = __builtin_fma(a[3], c, e); }
            if constexpr (OP == 10) x = __builtin_fma(a[(r + 3) & 7], a[(r + 5) & 7], x);      // 3 VGPR-pair operands
            if constexpr (OP == 11) x = a[(r + 3) & 7] * a[(r + 5) & 7];                     // mul, 2 VGPR-pair operands, no RAW on x
            if constexpr (OP == 12) x = __builtin_fma(a[(r + 3) & 7], c, x);                 // 2 VGPR pairs + uniform
            if constexpr (OP == 9) { float f = __builtin_bit_cast(float, __double2loint(x)); f = __builtin_fmaf(f, 1.0001f, 0.5f); x = __hiloint2double(__double2hiint(x), __builtin_bit_cast(int, f)); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name)
{
    const int iters = 2000;
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 16 * 8);
    for (int wps : {1, 2, 3}) {
        int threads = 64 * 4 * wps;              // wps waves on each of the CU's 4 SIMDs
        hipLaunchKernelGGL(k<OP>, dim3(256 * (wps > 4 ? 2 : 1)), dim3(threads > 1024 ? 1024 : threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 4 * wps);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= h.size();
        double per_wave = avg / ((double)iters * REP);
        printf("%-26s waves/SIMD=%d  cycles/instr per wave %.2f   per SIMD (throughput) %.2f\n", name, wps, per_wave, per_wave / wps);
    }
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0>("v_fma_f64 x*c+e"); run<13>("fma_f64 1 dependent chain"); run<14>("fma_f64 2 chains"); run<15>("rcp + 3 fma (per 4 instr)");
    run<10>("v_fma_f64 3xVGPR"); run<11>("v_mul_f64 2xVGPR"); run<12>("v_fma_f64 2xVGPR+uniform");
    return 0;
}
