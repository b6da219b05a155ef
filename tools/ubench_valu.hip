// Micro-benchmark: VALU issue cost of the matched-filter inner step on gfx950 for scalar
// (v_mul_f32 + v_add_f32) vs packed (v_pk_mul_f32 + v_pk_add_f32) forms, at 1, 2 and 4
// waves per SIMD.  Prints cycles per tap (4 multiplies + 4 dependent adds).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void k_scalar(float *out, const float *taps, int iters, unsigned long long *cyc)
{
    float x = out[threadIdx.x];
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 42; ++i) {
            const int j = i & 7; float h0 = taps[4 * j], h1 = taps[4 * j + 1], h2 = taps[4 * j + 2], h3 = taps[4 * j + 3];
            float p0 = x * h0, p1 = x * h1, p2 = x * h2, p3 = x * h3;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));   // keep hipcc from SLP-packing
            a0 += p0; a1 += p1; a2 += p2; a3 += p3;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            x = __builtin_amdgcn_mov_dpp(x, 0, 0xf, 0xf, false) ;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_packed(float *out, const float *taps, int iters, unsigned long long *cyc)
{
    float x = out[threadIdx.x];
    float2v a01 = {0, 0}, a23 = {0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 42; ++i) {
            const int j = i & 7; float2v h01 = {taps[4 * j], taps[4 * j + 1]}, h23 = {taps[4 * j + 2], taps[4 * j + 3]};
            float2v xx = {x, x};
            float2v p01 = xx * h01, p23 = xx * h23;
            a01 += p01; a23 += p23;
            x = __builtin_amdgcn_mov_dpp(x, 0, 0xf, 0xf, false);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a01.x + a01.y + a23.x + a23.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out, *taps; unsigned long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&taps, 4096); hipMalloc(&cyc, 8 * 4096);
    hipMemset(out, 0, 1 << 20); hipMemset(taps, 0, 4096);
    const int iters = 2000;
    for (int wpb : {1, 2, 4, 8}) {           // waves per block; one block per CU -> wpb/4 waves per SIMD
        for (int variant = 0; variant < 2; ++variant) {
            for (int rep = 0; rep < 2; ++rep) {
                if (variant == 0) hipLaunchKernelGGL(k_scalar, dim3(256), dim3(64 * wpb), 0, 0, out, taps, iters, cyc);
                else hipLaunchKernelGGL(k_packed, dim3(256), dim3(64 * wpb), 0, 0, out, taps, iters, cyc);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(256);
            hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
            printf("%s waves/block=%d (per SIMD %.2f): %.2f cycles per tap-step per wave\n",
                   variant ? "packed" : "scalar", wpb, wpb / 4.0, avg / (iters * 42.0));
        }
    }
    return 0;
}
