// Calibrate s_memtime ticks against wall time and measure dependent-chain latencies (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long* out, int iters) {
    double x = 1.0 + threadIdx.x * 1e-3, c = 1.0000001;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 64; r++) x = __builtin_fma(x, c, 1e-9);   // dependent chain
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)x; }
}
template <int OP>
__global__ void chain(long long* out, int iters) {
    double x = 1.0 + threadIdx.x * 1e-3, c = 1.0000001;
    int lane = threadIdx.x & 63;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
            if constexpr (OP == 0) x = __builtin_fma(x, c, 1e-9);
            if constexpr (OP == 1) x = __builtin_amdgcn_rcp(x);
            if constexpr (OP == 2) { int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x134, 0xF, 0xF, false); x = __hiloint2double(__double2hiint(x), lo); }
            if constexpr (OP == 3) { int lo = __builtin_amdgcn_ds_bpermute(((lane + 2) & 63) << 2, __double2loint(x)); x = __hiloint2double(__double2hiint(x), lo); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0 && threadIdx.x < 64) { out[0] = t1 - t0; out[1] = (long long)x; }
}
int main() {
    long long* d; hipMalloc(&d, 16); long long h[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 200000;
    spin<<<1, 64>>>(d, 1000); hipDeviceSynchronize();
    hipEventRecord(e0); spin<<<1, 64>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("s_memtime: %lld ticks in %.3f ms -> %.1f MHz ; dependent fma_f64 = %.2f ticks each\n", h[0], ms, h[0] / (ms * 1e3), h[0] / (64.0 * iters));
    const char* n[] = {"fma_f64", "rcp_f64", "dpp mov (b32 half)", "ds_bpermute (b32 half)"};
    chain<0><<<1, 64>>>(d, 2000); hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("dependent %-24s %.2f ticks\n", n[0], h[0] / (64.0 * 2000));
    chain<1><<<1, 64>>>(d, 2000); hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("dependent %-24s %.2f ticks\n", n[1], h[0] / (64.0 * 2000));
    chain<2><<<1, 64>>>(d, 2000); hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("dependent %-24s %.2f ticks\n", n[2], h[0] / (64.0 * 2000));
    chain<3><<<1, 64>>>(d, 2000); hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("dependent %-24s %.2f ticks\n", n[3], h[0] / (64.0 * 2000));
    return 0;
}
