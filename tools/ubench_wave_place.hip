// Where the waves of six-wave workgroups land (round 5): 384 threads, 77 KB of LDS, ~168 registers -- two workgroups fit a CU only if
// their twelve waves spread three per SIMD.  Per wave: HW_ID (SIMD, CU), start and end time; the host prints, per CU, which
// workgroups were resident together and the SIMD of each of their waves.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_wave_place.hip -o tools/ubench_wave_place && tools/ubench_wave_place [waves_per_wg] [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 12 ? 3 : 3) void where(unsigned *out, unsigned long long *t, int spin)
{
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r[140];
#pragma unroll
    for (int i = 0; i < 140; ++i) r[i] = (float)(threadIdx.x * (i + 1));
    float a = threadIdx.x;
    for (int k = 0; k < spin; ++k) {
#pragma unroll
        for (int i = 0; i < 140; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(a));
        lds[threadIdx.x] = r[k % 140 == 0 ? 0 : 1];
    }
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 140; ++i) sum += r[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * NW + (threadIdx.x >> 6);
        out[w] = (xcc & 15u) << 16 | (hw & 0xffffu); t[2 * w] = t0; t[2 * w + 1] = t1;
    }
    if (sum == 12345.f) out[0] = 0;
}
template <int NW>
static void run(int n, size_t lds)
{
    unsigned *out; unsigned long long *t;
    hipMalloc(&out, n * NW * 4); hipMalloc(&t, n * NW * 16);
    hipFuncSetAttribute((const void *)where<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(where<NW>, dim3(n), dim3(NW * 64), lds, 0, out, t, 300); hipDeviceSynchronize(); }
    std::vector<unsigned> h(n * NW); std::vector<unsigned long long> ht(2 * n * NW);
    hipMemcpy(h.data(), out, n * NW * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, n * NW * 16, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0; for (int i = 0; i < n * NW; ++i) { tmin = std::min(tmin, ht[2 * i]); tmax = std::max(tmax, ht[2 * i + 1]); }
    std::map<unsigned, std::vector<int>> by_cu;
    for (int g = 0; g < n; ++g) { const unsigned v = h[g * NW]; const unsigned key = (v >> 16) << 12 | ((v >> 13) & 7u) << 8 | ((v >> 12) & 1u) << 4 | ((v >> 8) & 15u); by_cu[key].push_back(g); }
    printf("%d waves per workgroup, %zu B LDS: %d workgroups on %zu CUs; launch %llu ticks (100 MHz)\n", NW, lds, n, by_cu.size(), tmax - tmin);
    int shown = 0; size_t together = 0, cus = 0;
    for (auto &kv : by_cu) {
        // workgroups whose lifetimes overlap the first one's
        const int g0 = kv.second[0]; int n_over = 0;
        for (int g : kv.second) if (ht[2 * g * NW] < ht[2 * g0 * NW + 1] && ht[2 * g0 * NW] < ht[2 * g * NW + 1]) ++n_over;
        together += n_over; ++cus;
        if (shown++ < 10) {
            printf("cu %05x:", kv.first);
            for (int g : kv.second) { printf("  wg %d (+%llu..%llu) simd", g, (ht[2 * g * NW] - tmin) / 10, (ht[2 * g * NW + 1] - tmin) / 10); for (int w = 0; w < NW; ++w) printf(" %u", (h[g * NW + w] >> 4) & 3u); }
            printf("\n");
        }
    }
    printf("workgroups resident together with a CU's first one (itself included): %.2f on average\n", (double)together / cus);
    hipFree(out); hipFree(t);
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    run<6>(n, 78848); run<4>(n, 78848); run<12>(n / 2, 157696); run<8>(n, 78848);
    return 0;
}
