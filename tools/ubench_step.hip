// What does one step of the wavefront pipeline cost before any work is done?  A workgroup of W wavefronts runs N steps
// of:  [optional LDS reads a stage does at the start of its step] -> s_waitcnt lgkmcnt(0); s_barrier -> [read the
// feedback word from LDS, readfirstlane, branch], in the variants
//   0  barrier only
//   1  barrier + feedback word (what every stage of the pipeline does)
//   2  one mailbox read before the barrier, then 1   (an idle stage 3 / stage 2)
//   3  two dependent mailbox reads before the barrier, then 1   (an idle helper before its reads were merged)
// Grid: one workgroup per CU (256) or two (512, 70 KB of LDS each), 5 or 4 wavefronts -- the shapes of the strict
// configs[1] launch and of the full-chip regime.  Prints shader clocks per step of workgroup 0's first wavefront.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_step.hip -o tools/ubench_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef volatile __attribute__((address_space(3))) uint32_t lds_u32;

template <int MODE>
__global__ void steps(uint32_t n, unsigned long long *out, uint32_t *sink)
{
    extern __shared__ float lds[];
    lds_u32 *box = (lds_u32 *)lds;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    box[threadIdx.x] = 0u; box[1024 + lane] = lane;
    __syncthreads();
    uint32_t acc = 0;
    const unsigned long long t0 = clock64();
    for (uint32_t s = 0; s < n; ++s) {
        if (MODE >= 2) { const uint32_t a = box[1024 + lane]; if (a & 64u) acc += 1u; if (MODE >= 3) { const uint32_t b = box[1024 + ((a + s) & 63u)]; if (b & 64u) acc += 2u; } }
        if (wave == 2u && lane == 0u) box[512 + (s & 1u)] = 0u;                    // (stage 3 posts the word)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE >= 1) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)box[512 + (s & 1u)]);
            if (w & 1u) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); acc += 3u; }
        }
    }
    const unsigned long long t1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0xffffffffu) sink[0] = acc;
}

template <int MODE> static void run(const char *what, int grid, int waves, size_t lds, unsigned long long *d_out, uint32_t *d_sink)
{
    const uint32_t n = 20000;
    hipFuncSetAttribute((const void *)steps<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(steps<MODE>, dim3(grid), dim3(waves * 64), lds, 0, n, d_out, d_sink);
        hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    printf("  mode %d (%s): %6.1f clk per step\n", MODE, what, (double)h / n);
}

int main()
{
    unsigned long long *d_out; uint32_t *d_sink;
    hipMalloc(&d_out, 64); hipMalloc(&d_sink, 64);
    const char *names[4] = {"barrier only", "barrier + feedback word", "mailbox read, barrier, feedback word", "two dependent mailbox reads, barrier, feedback word"};
    struct { const char *what; int grid, waves; size_t lds; } shapes[] = {
        {"one workgroup of 5 wavefronts per CU", 256, 5, 100 * 1024}, {"one workgroup of 4 wavefronts per CU", 256, 4, 100 * 1024},
        {"two workgroups of 4 wavefronts per CU", 512, 4, 70 * 1024}};
    for (auto &sh : shapes) {
        printf("%s:\n", sh.what);
        run<0>(names[0], sh.grid, sh.waves, sh.lds, d_out, d_sink);
        run<1>(names[1], sh.grid, sh.waves, sh.lds, d_out, d_sink);
        run<2>(names[2], sh.grid, sh.waves, sh.lds, d_out, d_sink);
        run<3>(names[3], sh.grid, sh.waves, sh.lds, d_out, d_sink);
    }
    return 0;
}
