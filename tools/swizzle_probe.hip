// Probe: what ds_swizzle_b32's rotate mode (offset >= 0xC000) does on gfx950.
// hipcc --offload-arch=gfx950 -O2 tools/swizzle_probe.hip -o /tmp/swizzle_probe && /tmp/swizzle_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int PAT>
__global__ void probe(int *out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_ds_swizzle(lane, PAT);
}

template <int PAT>
void run(const char *name)
{
    int *d, h[64];
    hipMalloc(&d, 64 * sizeof(int));
    hipLaunchKernelGGL(probe<PAT>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-28s pattern 0x%04x:", name, PAT);
    for (int i = 0; i < 64; i++) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}

int main()
{
    run<0xC000 | (0 << 10) | (1 << 5)>("rotate dir0 n1");
    run<0xC000 | (1 << 10) | (1 << 5)>("rotate dir1 n1");
    run<0xC000 | (0 << 10) | (2 << 5)>("rotate dir0 n2");
    run<0xC000 | (1 << 10) | (8 << 5)>("rotate dir1 n8");
    run<(0x10 << 10) | 0x1f>("bitmode xor16");
    return 0;
}
