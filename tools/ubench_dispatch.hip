// Where the dispatcher puts the workgroups of a launch that fills the chip twice over (256 threads, 46 KB of LDS: two per CU):
// per workgroup the XCD, shader engine and CU it ran on and when it started.  Prints which grid positions shared a CU.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_dispatch.hip -o tools/ubench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void where(unsigned *out, unsigned long long *t, int spin)
{
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float a = threadIdx.x;
    for (int i = 0; i < spin * (1 + (int)(blockIdx.x % 3)); ++i) { a = a * 1.0001f + 0.5f; lds[threadIdx.x] = a; }
    if (threadIdx.x == 0) { out[blockIdx.x] = (xcc & 15u) << 16 | (hw & 0xffffu); t[blockIdx.x] = t0; }
    if (a == 12345.f) out[0] = 0;
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 576;
    unsigned *out; unsigned long long *t;
    hipMalloc(&out, n * 4); hipMalloc(&t, n * 8);
    hipFuncSetAttribute((const void *)where, hipFuncAttributeMaxDynamicSharedMemorySize, 47104);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(where, dim3(n), dim3(256), 47104, 0, out, t, 20000); hipDeviceSynchronize(); }
    std::vector<unsigned> h(n); std::vector<unsigned long long> ht(n);
    hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, n * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ht[0]; for (auto v : ht) t0 = v < t0 ? v : t0;
    std::map<unsigned, std::vector<int>> by_cu;
    for (int i = 0; i < n; ++i) { const unsigned key = (h[i] >> 16) << 12 | ((h[i] >> 13) & 7u) << 8 | ((h[i] >> 12) & 1u) << 4 | ((h[i] >> 8) & 15u); by_cu[key].push_back(i); }
    printf("%d workgroups on %zu distinct (xcd, se, sh, cu)\n", n, by_cu.size());
    for (int i = 0; i < 40 && i < n; ++i) printf("wg %3d: xcd %u se %u sh %u cu %2u  start +%llu\n", i, h[i] >> 16, (h[i] >> 13) & 7u, (h[i] >> 12) & 1u, (h[i] >> 8) & 15u, ht[i] - t0);
    int shown = 0;
    for (auto &kv : by_cu) { if (shown++ >= 24) break; printf("cu key %05x:", kv.first); for (int i : kv.second) printf(" %d(+%llu)", i, (ht[i] - t0) / 100); printf("\n"); }
    return 0;
}
