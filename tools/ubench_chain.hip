// Dependent-chain cost on gfx950 for a wavefront alone on its SIMD: cycles per step of
//   (a) one v_pk_add_f32 chain          (the matched filter's accumulation of one filter: re, im packed)
//   (b) two interleaved v_add_f32 chains (the same arithmetic unpacked)
//   (c) one v_add_f32 chain
//   (d) pk_mul + pk_add per step, products independent (what a filter tap costs besides its loads)
// hipcc --offload-arch=gfx950 -O2 tools/ubench_chain.hip -o tools/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int N = 512;
template <int MODE>
__global__ void chain(float *out, unsigned long long *cyc, float seed)
{
    float2v a = {seed, seed + 1.0f}, p = {seed * 0.5f, seed * 0.25f}, x = {seed, seed};
    float s0 = seed, s1 = seed + 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(p)); }
        if (MODE == 1) { asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(s0), "+v"(s1) : "v"(p.x), "v"(p.y)); }
        if (MODE == 2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(p.x)); }
        if (MODE == 3) { float2v q; asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_add_f32 %1, %1, %0" : "=&v"(q), "+v"(a) : "v"(x), "v"(p)); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a.x + a.y + s0 + s1;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    const char *names[] = {"one v_pk_add_f32 chain", "two interleaved v_add_f32 chains", "one v_add_f32 chain", "v_pk_mul_f32 + dependent v_pk_add_f32"};
    for (int m = 0; m < 4; ++m)
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f);
            if (m == 1) hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f);
            if (m == 2) hipLaunchKernelGGL(chain<2>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f);
            if (m == 3) hipLaunchKernelGGL(chain<3>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f);
            hipDeviceSynchronize();
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-42s %7.2f shader-clock ticks per step (%d steps)\n", names[m], (double)h / N, N);
        }
    return 0;
}
