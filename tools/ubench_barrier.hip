// Cost of the step barrier (s_waitcnt lgkmcnt(0); s_barrier) by wavefronts per workgroup: 4 x 2 workgroups per CU against 12 x 1
// (round 5: the twelve-wavefront form of same_kernels_sym.hip).  Each wavefront does `work` dependent FMAs of its own between
// barriers (0: the bare barrier), wave w doing work * (1 + w % 3) / 2 so that arrivals are staggered as in a pipeline.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_barrier.hip -o tools/ubench_barrier && tools/ubench_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NW>
__global__ __launch_bounds__(NW * 64) void bar(unsigned long long *out, int iters, int work, float *sink)
{
    __shared__ float box[NW * 64];
    const int w = threadIdx.x >> 6;
    float a = threadIdx.x * 0.001f;
    const int mine = work * (1 + w % 3) / 2;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        for (int k = 0; k < mine; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
        box[threadIdx.x] = a;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        a += box[(threadIdx.x + 64) % (NW * 64)];
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a == 1234.5f) *sink = a;
}
template <int NW>
static void run(int grid, int work)
{
    unsigned long long *out; float *sink;
    hipMalloc(&out, grid * 8); hipMalloc(&sink, 4);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(bar<NW>, dim3(grid), dim3(NW * 64), 0, 0, out, iters, work, sink); hipDeviceSynchronize(); }
    unsigned long long h[4096]; hipMemcpy(h, out, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
    printf("%2d waves per workgroup, %4d workgroups, work %4d: %.0f clk per step\n", NW, grid, work, s / grid / iters);
    hipFree(out); hipFree(sink);
}
int main()
{
    for (int work : {0, 100, 400}) {
        run<4>(256, work); run<4>(512, work); run<6>(256, work); run<8>(256, work); run<12>(256, work); run<16>(256, work);
    }
    return 0;
}
