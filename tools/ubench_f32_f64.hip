// Issue rate of plain fp32, packed fp32 and fp64 FMAs on one SIMD with ONE and with TWO resident wavefronts (round-3
// review, "Next" 4: DESIGN.md contradicted itself on whether v_fma_f32 issues faster than v_fma_f64 on CDNA4).
// Each wavefront runs 8 independent dependent-chains of FMAs (enough to cover the pipeline latency by itself); a
// workgroup puts `wps` wavefronts on each of its CU's four SIMDs; 256 workgroups = one per CU.  Timed with HIP events;
// cycles = time x the shader clock the kernel itself measures (s_memrealtime, 100 MHz, against clock64()).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_f32_f64.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float *out, unsigned long long *clk, int iters)
{
    const int lane = threadIdx.x;
    double d[8]; float f[8]; float2_ p[8];
    for (int i = 0; i < 8; i++) { d[i] = 1.0 + 1e-3 * (lane + i); f[i] = (float)d[i]; p[i] = float2_{f[i], f[i] + 1.0f}; }
    const double cd = 0.999999, ed = 1e-6;
    const float cf = 0.999999f, ef = 1e-6f;
    const float2_ cp = {cf, cf}, ep = {ef, ef};
    const unsigned long long c0 = clock64(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if constexpr (OP == 0) d[r & 7] = __builtin_fma(d[r & 7], cd, ed);
            if constexpr (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[r & 7]) : "s"(cf), "v"(ef));   // (plain C gets SLP-vectorised into v_pk_fma_f32)
            if constexpr (OP == 2) p[r & 7] = __builtin_elementwise_fma(p[r & 7], cp, ep);      // v_pk_fma_f32
            if constexpr (OP == 3) d[r & 7] = __builtin_fma(d[r & 7], d[(r + 3) & 7], d[(r + 5) & 7]);   // 3 VGPR-pair operands
            if constexpr (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[r & 7]) : "v"(f[(r + 3) & 7]), "v"(f[(r + 5) & 7]));
        }
    }
    const unsigned long long c1 = clock64(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += (float)d[i] + f[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int OP>
void run(const char *name, double flop_per_lane_instr)
{
    const int iters = 4000;
    float *out; unsigned long long *clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4}) {
        const int threads = 64 * 4 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, clk, 10);       // warm-up
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);                       // clock64 ticks per ns (s_memrealtime: 100 MHz)
        const double instr_per_simd = (double)iters * REP * wps;
        const double cyc = ms * 1e6 * ghz / instr_per_simd;
        printf("%-28s waves/SIMD=%d  %.3f ms  clock %.2f GHz  SIMD cycles per wave64 instruction %.2f  (%.1f TFLOP/s chip-wide)\n",
               name, wps, ms, ghz, cyc, 1024.0 * instr_per_simd * 64 * flop_per_lane_instr / (ms * 1e-3) * 1e-12);
    }
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0>("v_fma_f64 (x*c+e)", 2);
    run<3>("v_fma_f64 (3 VGPR operands)", 2);
    run<1>("v_fma_f32 (x*c+e)", 2);
    run<4>("v_fma_f32 (3 VGPR operands)", 2);
    run<2>("v_pk_fma_f32", 4);
    return 0;
}
