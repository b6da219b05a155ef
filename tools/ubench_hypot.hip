// Cost of rs_hypot (the f64 formulation of glibc hypotf used by the kernels) for a wavefront alone on
// its SIMD, in shader-clock ticks per call, dependent chain (as on stage 2's critical path).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I sameold_amd/csrc tools/ubench_hypot.hip -o tools/ubench_hypot
#include <hip/hip_runtime.h>
#include <cstdio>
#include "same_dev_common.h"
constexpr int N = 256;
__global__ void k(float *out, unsigned long long *cyc, float a, float b)
{
    float x = a + threadIdx.x * 1e-3f, y = b;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
    for (int i = 0; i < N; ++i) { x = same::rs_hypot(x, y); asm volatile("" : "+v"(x)); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.5f, 0.25f);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (rep) printf("rs_hypot: %.1f shader-clock ticks per dependent call (%d calls)\n", (double)h / N, N);
    }
    return 0;
}
