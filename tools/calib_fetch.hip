// Calibration of rocprofv3 FETCH_SIZE for this project's access pattern: every lane reads one
// dword per time step from a time-major array x[t][channel] (256 contiguous bytes per
// wavefront-instruction), exactly like the demodulation kernel.  Known byte count: T*C*4.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_time_major(const float *x, float *out, unsigned C, unsigned T)
{
    unsigned c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
    for (unsigned t = 0; t < T; t += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = x[(size_t)(t + k) * C + c];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    out[c] = acc;
}
int main()
{
    const unsigned C = 4096, T = 220496;
    float *x, *out;
    hipMalloc(&x, (size_t)C * T * 4); hipMalloc(&out, C * 4);
    hipMemset(x, 0, (size_t)C * T * 4);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(read_time_major, dim3(C / 64), dim3(64), 0, 0, x, out, C, T);
    hipDeviceSynchronize();
    printf("known bytes read per launch: %zu\n", (size_t)C * T * 4);
    return 0;
}
