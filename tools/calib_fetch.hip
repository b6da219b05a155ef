// Calibration of rocprofv3 FETCH_SIZE for this project's access pattern: every live lane reads one
// dword per time step from a time-major array x[t][channel], exactly like the demodulation
// kernels.  Two launches shapes, three launches each, known byte count T*C*4 for both:
//   read_time_major<64>: 64 channels per wavefront (256 contiguous bytes per wavefront-instruction),
//                        the one-wavefront kernels and the 64-channel pipeline workgroups;
//   read_time_major<16>: 16 channels per wavefront (64 contiguous bytes), the narrow pipeline
//                        workgroups small batches use.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LANES>
__global__ void read_time_major(const float *x, float *out, unsigned C, unsigned T)
{
    if (threadIdx.x >= LANES) return;
    unsigned c = blockIdx.x * LANES + threadIdx.x;
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
// the time-parallel launch on a channel-major input: every lane streams its own contiguous channel with 16-byte loads,
// 20 samples (five loads) per step, 64 lanes = 64 streams C_STRIDE samples apart
__global__ void read_channel_major(const float *x, float *out, unsigned C, unsigned T)
{
    unsigned c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const float4 *p = reinterpret_cast<const float4 *>(x + (size_t)c * T);
    float acc = 0.f;
    for (unsigned t = 0; t + 20 <= T; t += 20) {
        float4 v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = p[t / 4 + k];
#pragma unroll
        for (int k = 0; k < 5; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    out[c] = acc;
}
int main()
{
    const unsigned C = 4096, T = 220496;
    float *x, *out;
    hipMalloc(&x, (size_t)C * T * 4); hipMalloc(&out, C * 4);
    hipMemset(x, 0, (size_t)C * T * 4);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(read_time_major<64>, dim3(C / 64), dim3(64), 0, 0, x, out, C, T);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(read_time_major<16>, dim3(C / 16), dim3(64), 0, 0, x, out, C, T);
    // (T = 220480 here: whole 20-sample steps, 16-byte aligned streams)
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(read_channel_major, dim3(C / 64), dim3(64), 0, 0, x, out, C, 220480u);
    hipDeviceSynchronize();
    printf("known bytes read per launch: %zu (time-major shapes), %zu (channel-major streams)\n", (size_t)C * T * 4, (size_t)C * 220480u * 4);
    return 0;
}
