// Hardware probe (development aid, not part of the library): accuracy of v_rcp_f64 with 0/1/2
// Newton steps, and the lane semantics of the DPP / permlane ops the fast path relies on.
// Build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/hw_probe.hip -o /tmp/hw_probe && /tmp/hw_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void rcp_k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r); r1[i] = r;
    e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r); r2[i] = r;
}
__global__ void lanes_k(int* out) {
    int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x134, 0xF, 0xF, false);        // wave_rol:1
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x13C, 0xF, 0xF, false);   // wave_ror:1
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
    out[128 + l] = r[0]; out[192 + l] = r[1];
    out[256 + l] = __builtin_amdgcn_update_dpp(0, l, 0x142, 0xA, 0xF, true);    // row_bcast:15 rows 1,3
    out[320 + l] = __builtin_amdgcn_update_dpp(0, l, 0x143, 0xC, 0xF, true);    // row_bcast:31 rows 2,3
    out[384 + l] = __builtin_amdgcn_update_dpp(0, l, 0x111, 0xF, 0xF, true);    // row_shr:1
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-12, 12);
    for (auto& v : h) v = std::pow(10.0, u(g)) * ((g() & 1) ? 1 : -1);
    double *x, *r0, *r1, *r2;
    hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    rcp_k<<<n / 256, 256>>>(x, r0, r1, r2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)h[i];
        e0 = std::fmax(e0, (double)fabsl((a[i] - t) / t));
        e1 = std::fmax(e1, (double)fabsl((b[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl((c[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err: raw %.3e (2^%.1f)  1 NR %.3e  2 NR %.3e\n", e0, std::log2(e0), e1, e2);
    int* o; hipMalloc(&o, 448 * 4);
    lanes_k<<<1, 64>>>(o);
    std::vector<int> ho(448);
    hipMemcpy(ho.data(), o, 448 * 4, hipMemcpyDeviceToHost);
    const char* names[] = {"wave_rol:1", "wave_ror:1", "swap32 r0(a=l,b=100+l)", "swap32 r1", "row_bcast:15 m=0xA", "row_bcast:31 m=0xC", "row_shr:1"};
    for (int k = 0; k < 7; k++) {
        printf("%-24s lanes 0,1,15,16,17,31,32,33,47,48,62,63: ", names[k]);
        for (int l : {0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 62, 63}) printf("%d ", ho[k * 64 + l]);
        printf("\n");
    }
    return 0;
}
