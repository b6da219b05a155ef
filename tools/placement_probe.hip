// Where does the dispatcher put the workgroups of a launch that fits the machine in one round?
// 512 workgroups of 256 threads, 70 KB of LDS each, two per CU (the time-parallel launch's shape): every
// workgroup records XCC_ID and HW_ID, spins long enough for all of them to be resident together, and the host
// prints which grid positions share a CU.  Second part: workgroup w spins len[w] "samples"; the launch time for
// several orders of the same lengths shows whether a CU's second workgroup speeds up when the first has left
// (it does not here: a spin has no issue contention; the real kernel is measured with tools/tp_cm_once.py).
// hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o tools/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>

__global__ __launch_bounds__(256, 2) void probe(uint32_t *out, unsigned long long spin)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (lds[threadIdx.x] < 0.0f) out[0] = 0;
}

int main()
{
    const int G = 512;
    uint32_t *d; hipMalloc(&d, G * 8);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 70 * 1024, 0, d, 200000ull);   // 100 MHz clock: 2 ms
        hipDeviceSynchronize();
        std::vector<uint32_t> h(2 * G);
        hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        std::map<uint32_t, std::vector<int>> by_cu;
        for (int w = 0; w < G; ++w) {
            const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 15u;
            const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(w);
        }
        printf("launch %d: %zu distinct CUs hold the %d workgroups\n", rep, by_cu.size(), G);
        if (rep == 2) {
            for (int w = 0; w < 40; ++w) {
                const uint32_t hw = h[2 * w];
                printf("  wg %3d: xcc %u se %u sh %u cu %2u simd %u\n", w, h[2 * w + 1] & 15u, (hw >> 13) & 7u, (hw >> 12) & 1u, (hw >> 8) & 15u, (hw >> 4) & 3u);
            }
            std::map<int, int> gap;
            int singles = 0;
            for (auto &kv : by_cu) {
                auto &v = kv.second;
                if (v.size() == 1) ++singles;
                for (size_t i = 1; i < v.size(); ++i) gap[v[i] - v[0]]++;
            }
            printf("  CUs with one workgroup: %d; grid distance between the workgroups sharing a CU (distance: count):", singles);
            for (auto &kv : gap) printf(" %d:%d", kv.first, kv.second);
            printf("\n  partners of wg 0..15:");
            for (auto &kv : by_cu) for (int w : kv.second) if (w < 16) { printf(" ["); for (int u : kv.second) printf("%d ", u); printf("]"); }
            printf("\n");
        }
    }
    return 0;
}
