// What ONE SIMD of gfx950 issues per clock when W wavefronts share it (round 6): W = 1, 2, 3, 4 (one workgroup of 4 W wavefronts
// per CU: the dispatcher deals a workgroup's wavefronts round the four SIMDs) and 8 (two such workgroups of 16), each wavefront
// running a straight-line stream of one instruction mix.  Per mix and W: shader-clock ticks per instruction and WAVEFRONT
// (s_memtime around the loop, median wavefront) and instructions per tick and SIMD (W / that).  The symbol-paced kernel runs
// three wavefronts per SIMD at ~0.3 instructions per clock and SIMD: is that the SIMD's limit for its mix, or each wavefront's?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_simd.hip -o tools/ubench_simd && tools/ubench_simd
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int UNROLL = 64, ITERS = 256;
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, float seed, int iters)
{
    __shared__ float lds[4096];
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float2v p0 = {seed, seed}, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f, p4 = p0 + 4.0f, p5 = p0 + 5.0f, p6 = p0 + 6.0f, p7 = p0 + 7.0f;
    const float b = seed * 0.999f, c = 0.001f;
    const float2v b2 = {b, b}, c2 = {c, c};
    unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), s1 = 1;
    float l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    const unsigned la = (threadIdx.x & 63u) * 4u;
    lds[threadIdx.x & 4095] = seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; ++u) {
            if (MODE == 0)       // eight independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if (MODE == 1)       // eight independent v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                             "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(b2), "v"(c2));
            if (MODE == 2)       // one dependent v_fma_f32 chain
                asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                             "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                             : "+v"(a0) : "v"(b), "v"(c));
            if (MODE == 3)       // VALU and SALU alternating, all independent
                asm volatile("v_fma_f32 %0, %0, %6, %7\n\ts_add_u32 %4, %4, %5\n\tv_fma_f32 %1, %1, %6, %7\n\ts_add_u32 %4, %4, %5\n\t"
                             "v_fma_f32 %2, %2, %6, %7\n\ts_add_u32 %4, %4, %5\n\tv_fma_f32 %3, %3, %6, %7\n\ts_add_u32 %4, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0) : "s"(s1), "v"(b), "v"(c) : "scc");
            if (MODE == 4)       // SALU only
                asm volatile("s_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\t"
                             "s_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1"
                             : "+s"(s0) : "s"(s1) : "scc");
            if (MODE == 5)       // 4 ds_read_b32 + 4 v_fma_f32 on earlier loads (one lgkmcnt wait per eight)
                asm volatile("ds_read_b32 %4, %8\n\tds_read_b32 %5, %8 offset:256\n\tds_read_b32 %6, %8 offset:512\n\tds_read_b32 %7, %8 offset:768\n\t"
                             "v_fma_f32 %0, %0, %9, %10\n\tv_fma_f32 %1, %1, %9, %10\n\tv_fma_f32 %2, %2, %9, %10\n\tv_fma_f32 %3, %3, %9, %10\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(la), "v"(b), "v"(c) : "memory");
            if (MODE == 6)       // v_pk_fma_f32 dependent chain
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\t"
                             "v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2"
                             : "+v"(p0) : "v"(b2), "v"(c2));
            if (MODE == 7)       // the kernel's own flavour: v_cndmask + v_cmp + s_and (select-heavy symbol-rate code)
                asm volatile("v_cmp_lt_f32 vcc, %0, %4\n\tv_cndmask_b32 %1, %1, %0, vcc\n\ts_and_b64 vcc, vcc, exec\n\tv_cmp_gt_f32 vcc, %2, %4\n\t"
                             "v_cndmask_b32 %3, %3, %2, vcc\n\ts_and_b64 vcc, vcc, exec\n\tv_add_f32 %0, %0, %5\n\tv_add_f32 %2, %2, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "scc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + (float)s0 + l0 + l1 + l2 + l3;
    if ((threadIdx.x & 63u) == 0u) cyc[blockIdx.x * 16u + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> static void run(const char *name, float *out, unsigned long long *cyc)
{
    printf("%-58s", name);
    for (int W : {1, 2, 3, 4, 8}) {
        const int waves = W == 8 ? 16 : 4 * W, grid = W == 8 ? 512 : 256;
        hipMemset(cyc, 0, 512 * 16 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * waves), 0, 0, out, cyc, 1.0f, ITERS);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * waves), 0, 0, out, cyc, 1.0f, ITERS);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(512 * 16);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) v.push_back((double)h[b * 16 + w]);
        std::sort(v.begin(), v.end());
        const double per = v[v.size() / 2] / ((double)ITERS * UNROLL);
        // wall: instructions per SIMD / (ms * clock) -- the clock is not known here; print the wall time per instruction and SIMD in ns
        const double ns = (double)ms * 1e6 / ((double)ITERS * UNROLL * W);
        printf(" | W=%d %6.2f tick/inst/wave %5.3f inst/tick/SIMD %5.3f ns/inst/SIMD", W, per, W / per, ns);
    }
    printf("\n");
}
int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 512 * 16 * 8);
    run<0>("8 independent v_fma_f32", out, cyc);
    run<1>("8 independent v_pk_fma_f32", out, cyc);
    run<2>("dependent v_fma_f32 chain", out, cyc);
    run<6>("dependent v_pk_fma_f32 chain", out, cyc);
    run<3>("v_fma_f32 / s_add_u32 alternating", out, cyc);
    run<4>("s_add_u32 chain", out, cyc);
    run<5>("4 ds_read_b32 + 4 v_fma_f32 + s_waitcnt", out, cyc);
    run<7>("v_cmp / v_cndmask / s_and / v_add (select code)", out, cyc);
    return 0;
}
