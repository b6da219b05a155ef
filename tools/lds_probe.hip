// How much dynamic LDS can one workgroup have on this GPU?  hipcc --offload-arch=gfx950 tools/lds_probe.hip -o tools/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, int n) { extern __shared__ float lds[]; lds[threadIdx.x] = 1.0f; lds[n - 1 - threadIdx.x] = 2.0f; __syncthreads(); out[threadIdx.x] = lds[threadIdx.x] + lds[n - 1 - threadIdx.x]; }
int main() {
    float *d; hipMalloc(&d, 1024);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  sharedMemPerBlockOptin %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
    for (size_t kb : {48, 64, 65, 96, 112, 128, 160}) {
        size_t bytes = kb * 1024;
        hipError_t a = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), bytes, 0, d, (int)(bytes / 4));
        hipError_t e = hipGetLastError(); hipError_t s = hipDeviceSynchronize();
        float h[2] = {0, 0}; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%zu KB: setattr %s, launch %s, sync %s, out %.1f\n", kb, hipGetErrorName(a), hipGetErrorName(e), hipGetErrorName(s), h[0]);
    }
    return 0;
}
