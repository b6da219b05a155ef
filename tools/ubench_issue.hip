// Issue cost of VALU sequences on gfx950, per instruction, for 1 / 2 / 4 / 8 wavefronts per SIMD:
//   0 independent v_fma_f32 (8 accumulators round-robin)        1 one dependent v_fma_f32 chain
//   2 dependent pair fma -> med3 (the relaxed AGC's gain chain)  3 the relaxed AGC step: fma(a) mul(out) fma(g) med3(g)
//   4 same + ds_write_b32 per sample                            5 independent v_pk_fma_f32 (4 accumulators)
//   6 v_pk_fma_f32 with op_sel broadcast, 4 accumulators        7 dependent v_pk_fma_f32 chain
// hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int N = 256, REP = 64;
template <int MODE>
__global__ __launch_bounds__(64) void bench(float *out, unsigned long long *cyc, float seed)
{
    __shared__ float lds[64 * 64];
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    float2v p[4] = {{seed, seed}, {seed, 1.f}, {2.f, seed}, {seed, 3.f}}, w = {seed, seed * 2}, h = {0.5f, 0.25f};
    float g = seed, y = seed * 0.001f, bw = 1e-5f, lo = 0.0f, hi = 1e6f, o = 0.f, aa = 0.f;
    float *wl = lds + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 7]) : "v"(y), "v"(bw));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g) : "v"(y), "v"(bw));
            if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_med3_f32 %0, %0, %3, %4" : "+v"(g) : "v"(y), "v"(bw), "v"(lo), "v"(hi));
            if (MODE == 3) asm volatile("v_fma_f32 %1, -%4, |%3|, 1.0\n\tv_mul_f32 %2, %3, %0\n\tv_fma_f32 %0, %0, %1, %4\n\tv_med3_f32 %0, %0, %5, %6"
                                        : "+v"(g), "=&v"(aa), "=&v"(o) : "v"(y), "v"(bw), "v"(lo), "v"(hi));
            if (MODE == 4) { asm volatile("v_fma_f32 %1, -%4, |%3|, 1.0\n\tv_mul_f32 %2, %3, %0\n\tv_fma_f32 %0, %0, %1, %4\n\tv_med3_f32 %0, %0, %5, %6"
                                        : "+v"(g), "=&v"(aa), "=&v"(o) : "v"(y), "v"(bw), "v"(lo), "v"(hi));
                             wl[(i & 63) * 64] = o; }
            if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i & 3]) : "v"(w), "v"(h));
            if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i & 3]) : "v"(w), "v"(h));
            if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[0]) : "v"(w), "v"(h));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = g + o + aa;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s + lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, int per_step, float *out, unsigned long long *cyc)
{
    printf("%-58s", name);
    for (int wps : {1, 2, 4, 8}) {
        unsigned long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(bench<MODE>, dim3(256 * 4 * wps), dim3(64), 0, 0, out, cyc, 1.0f);
            hipDeviceSynchronize();
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("  %d/SIMD: %6.2f", wps, (double)h / ((double)N * REP * per_step));
    }
    printf("   clk per instruction (wavefront 0)\n");
}
int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4); hipMalloc(&cyc, 64);
    run<0>("independent v_fma_f32 (8 accumulators)", 1, out, cyc);
    run<1>("dependent v_fma_f32 chain", 1, out, cyc);
    run<2>("dependent fma -> med3 chain", 2, out, cyc);
    run<3>("relaxed AGC step (4 VALU, 2 on the chain)", 4, out, cyc);
    run<4>("relaxed AGC step + ds_write_b32 (5 instructions)", 5, out, cyc);
    run<5>("independent v_pk_fma_f32 (4 accumulators)", 1, out, cyc);
    run<6>("v_pk_fma_f32 op_sel broadcast (4 accumulators)", 1, out, cyc);
    run<7>("dependent v_pk_fma_f32 chain", 1, out, cyc);
    return 0;
}
