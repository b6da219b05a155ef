// ubench_pair.hip -- go / no-go for "two channels per lane" in the sample phase of the symbol-paced pipeline (round 5).
//
// The per-sample work of a channel -- DC blocker (rx/dcblock.rs:45-49, 104-108), relaxed AGC (rx/agc.rs:72-77) and the push
// into the window ring -- is serial per lane and the floor under every other cut.  v_pk_add / v_pk_mul / v_pk_fma_f32 are
// full rate on gfx950, so with lane = a PAIR of channels every packed operation serves two channels.  What does not pack:
// |y| (no abs modifier on packed f32), the gain's clamp (no packed med3 / min / max for f32).
//
// Two roles per workgroup, as in same_kernels_sym.hip: T = input loads + DC blocker into an LDS ring, S = AGC in place; one
// LDS barrier per 36-sample block.  MODE 1: one channel per lane, exactly the instruction sequences of the shipped kernel
// (sample PAIRS packed: 3.5 + 3.5 vector instructions per channel-sample).  MODE 2: two channels per lane (channel pairs
// packed: 2.5 + 3).  Reports shader clocks per block and per channel-sample, for one and two workgroups' worth of waves per
// SIMD, and the register counts (hipcc --save-temps / -Rpass-analysis=kernel-resource-usage).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench_pair tools/ubench_pair.hip && tools/ubench_pair
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int kB = 36, kDCL = 16, kWave = 64, kRingBlocks = 3;

template <typename F, int... I>
__device__ __forceinline__ void static_for_(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------- MODE 1: one channel per lane, sample pairs packed
struct Dc1 {
    typedef float2v Pairs[kB / 2];
    float sum0 = 0.0f, sum1 = 0.0f;
    float2v xp[kDCL / 2], sp[kDCL / 2];
    __device__ __forceinline__ void block(float *y, const Pairs &X)
    {
        Pairs snew;
        constexpr float kScale = -1.0f / 256.0f;
        const float2v nscale = {kScale, kScale};
        float s1_last = 0.0f;
        static_for<kB / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            auto xw = [&](auto i_) __attribute__((always_inline)) -> float2v { constexpr int i = decltype(i_)::value; if constexpr (i < kDCL / 2) return xp[i]; else return X[i - kDCL / 2]; };
            auto sw = [&](auto i_) __attribute__((always_inline)) -> float2v { constexpr int i = decltype(i_)::value; if constexpr (i < kDCL / 2) return sp[i]; else return snew[i - kDCL / 2]; };
            const float2v xo = xw(std::integral_constant<int, h>{});
            const float2v d0 = X[h] - xo;
            const float s0a = sum0 + d0.x, s0b = s0a + d0.y;
            sum0 = s0b;
            const float2v s0 = {s0a, s0b};
            snew[h] = s0;
            const float2v d1 = s0 - sw(std::integral_constant<int, h>{});
            const float s1a = sum1 + d1.x, s1b = s1a + d1.y;
            sum1 = s1b;
            if constexpr (h == 0) y[0] = __builtin_fmaf(s1a, kScale, xo.y);
            else { const float2v yy = __builtin_elementwise_fma(float2v{s1_last, s1a}, nscale, xo); y[(2 * h - 1) * kWave] = yy.x; y[(2 * h) * kWave] = yy.y; }
            s1_last = s1b;
            if constexpr (h == kB / 2 - 1) y[(kB - 1) * kWave] = __builtin_fmaf(s1b, kScale, xw(std::integral_constant<int, h + 1>{}).x);
        });
        static_for<kDCL / 2>([&](auto h_) __attribute__((always_inline)) { constexpr int h = decltype(h_)::value; xp[h] = X[(kB - kDCL) / 2 + h]; sp[h] = snew[(kB - kDCL) / 2 + h]; });
    }
};
__device__ __forceinline__ void agc1(float *w, float &gain, float bw, float gmin, float gmax)
{
    float2v yv[kB / 2];
#pragma unroll
    for (int h = 0; h < kB / 2; ++h) yv[h] = float2v{w[(2 * h) * kWave], w[(2 * h + 1) * kWave]};
#pragma unroll
    for (int h = 0; h < kB / 2; ++h) {
        const float a0 = __builtin_fmaf(-bw, fabsf(yv[h].x), 1.0f), a1 = __builtin_fmaf(-bw, fabsf(yv[h].y), 1.0f);
        const float g1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(gain, a0, bw), gmin, gmax);
        const float g2 = __builtin_amdgcn_fmed3f(__builtin_fmaf(g1, a1, bw), gmin, gmax);
        const float2v o = yv[h] * float2v{gain, g1};
        gain = g2;
        w[(2 * h) * kWave] = o.x; w[(2 * h + 1) * kWave] = o.y;
    }
}

// ---------------------------------------------------------------- MODE 2: two channels per lane, channel pairs packed
struct Dc2 {
    typedef float2v Block[kB];
    float2v sum0 = {0.0f, 0.0f}, sum1 = {0.0f, 0.0f};
    float2v xp[kDCL], sp[kDCL];                  // the last 16 inputs / first-stage sums of both channels
    __device__ __forceinline__ void block(float2v *y, const Block &X)
    {
        Block snew;
        constexpr float kScale = -1.0f / 256.0f;
        const float2v nscale = {kScale, kScale};
        static_for<kB>([&](auto k_) __attribute__((always_inline)) {
            constexpr int k = decltype(k_)::value;
            auto xw = [&](auto i_) __attribute__((always_inline)) -> float2v { constexpr int i = decltype(i_)::value; if constexpr (i < kDCL) return xp[i]; else return X[i - kDCL]; };
            auto sw = [&](auto i_) __attribute__((always_inline)) -> float2v { constexpr int i = decltype(i_)::value; if constexpr (i < kDCL) return sp[i]; else return snew[i - kDCL]; };
            sum0 = sum0 + (X[k] - xw(std::integral_constant<int, k>{}));
            snew[k] = sum0;
            sum1 = sum1 + (sum0 - sw(std::integral_constant<int, k>{}));
            y[k * kWave] = __builtin_elementwise_fma(sum1, nscale, xw(std::integral_constant<int, k + 1>{}));
        });
        static_for<kDCL>([&](auto k_) __attribute__((always_inline)) { constexpr int k = decltype(k_)::value; xp[k] = X[kB - kDCL + k]; sp[k] = snew[kB - kDCL + k]; });
    }
};
__device__ __forceinline__ void agc2(float2v *w, float2v &gain, float bw, float gmin, float gmax)
{
    float2v yv[kB];
#pragma unroll
    for (int k = 0; k < kB; ++k) yv[k] = w[k * kWave];
    const float2v bw2 = {bw, bw};
#pragma unroll
    for (int k = 0; k < kB; ++k) {
        const float2v a = {__builtin_fmaf(-bw, fabsf(yv[k].x), 1.0f), __builtin_fmaf(-bw, fabsf(yv[k].y), 1.0f)};
        const float2v o = yv[k] * gain;
        const float2v g = __builtin_elementwise_fma(gain, a, bw2);
        gain = float2v{__builtin_amdgcn_fmed3f(g.x, gmin, gmax), __builtin_amdgcn_fmed3f(g.y, gmin, gmax)};
        w[k * kWave] = o;
    }
}

// one workgroup = T + S over 64 lanes; n_blocks blocks of 36 samples; x: time-major rows of `row` floats.
// ROLES: 1 = T alone (S only keeps the barrier), 2 = S alone, 3 = both.  T requests block b + 1 before it computes block b (the
// shipped kernel's prefetch), so its loads have a whole step to arrive.
template <int MODE, int ROLES>
__global__ __launch_bounds__(2 * kWave, 2) void pair_kernel(const float *__restrict__ x, uint32_t row, uint32_t n_blocks, float bw, float gmin, float gmax,
                                                            float *out, unsigned long long *clk)
{
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x & 63u, role = threadIdx.x >> 6;
    constexpr int CH = MODE;                      // channels per lane
    const uint32_t col = (blockIdx.x * kWave + lane) * CH;
    float *ring = lds + lane * CH;                // [kRingBlocks][kB][64 * CH]
    const unsigned long long t0 = clock64();
    if (role == 0) {
        if constexpr (MODE == 1) {
            Dc1 D;
            static_for<kDCL / 2>([&](auto h_) __attribute__((always_inline)) { constexpr int h = decltype(h_)::value; D.xp[h] = float2v{0.0f, 0.0f}; D.sp[h] = float2v{0.0f, 0.0f}; });
            Dc1::Pairs XA, XB;
            auto request = [&](Dc1::Pairs &X, uint32_t b) __attribute__((always_inline)) {
                const float *xr = x + (size_t)(b < n_blocks ? b : n_blocks - 1u) * kB * row + col;
                static_for<kB / 2>([&](auto h_) __attribute__((always_inline)) { constexpr int h = decltype(h_)::value; X[h] = float2v{xr[(size_t)(2 * h) * row], xr[(size_t)(2 * h + 1) * row]}; });
            };
            request(XA, 0u);
            for (uint32_t b = 0; b < n_blocks; b += 2u) {
                request(XB, b + 1u);
                if (ROLES & 1) D.block(ring + ((b % kRingBlocks) * kB) * kWave, XA);
                lds_barrier();
                request(XA, b + 2u);
                if (ROLES & 1) D.block(ring + (((b + 1u) % kRingBlocks) * kB) * kWave, XB);
                lds_barrier();
            }
            out[col] = D.sum1 + XA[0].x;
        } else {
            Dc2 D;
            static_for<kDCL>([&](auto k_) __attribute__((always_inline)) { constexpr int k = decltype(k_)::value; D.xp[k] = float2v{0.0f, 0.0f}; D.sp[k] = float2v{0.0f, 0.0f}; });
            Dc2::Block XA, XB;
            auto request = [&](Dc2::Block &X, uint32_t b) __attribute__((always_inline)) {
                const float *xr = x + (size_t)(b < n_blocks ? b : n_blocks - 1u) * kB * row + col;
                static_for<kB>([&](auto k_) __attribute__((always_inline)) { constexpr int k = decltype(k_)::value; X[k] = *reinterpret_cast<const float2v *>(xr + (size_t)k * row); });
            };
            request(XA, 0u);
            for (uint32_t b = 0; b < n_blocks; b += 2u) {
                request(XB, b + 1u);
                if (ROLES & 1) D.block(reinterpret_cast<float2v *>(ring + ((b % kRingBlocks) * kB) * kWave * 2), XA);
                lds_barrier();
                request(XA, b + 2u);
                if (ROLES & 1) D.block(reinterpret_cast<float2v *>(ring + (((b + 1u) % kRingBlocks) * kB) * kWave * 2), XB);
                lds_barrier();
            }
            out[col] = D.sum1.x + D.sum1.y + XA[0].x;
        }
    } else {
        if constexpr (MODE == 1) {
            float gain = 1.0e-4f;
            for (uint32_t b = 0; b < n_blocks; ++b) {
                lds_barrier();
                if (ROLES & 2) agc1(ring + ((b % kRingBlocks) * kB) * kWave, gain, bw, gmin, gmax);
            }
            out[col] += gain;
        } else {
            float2v gain = {1.0e-4f, 1.0e-4f};
            for (uint32_t b = 0; b < n_blocks; ++b) {
                lds_barrier();
                if (ROLES & 2) agc2(reinterpret_cast<float2v *>(ring + ((b % kRingBlocks) * kB) * kWave * 2), gain, bw, gmin, gmax);
            }
            out[col] += gain.x + gain.y;
        }
    }
    if (lane == 0) clk[blockIdx.x * 2 + role] = clock64() - t0;
}

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int ROLES>
static void run(uint32_t wgs, uint32_t n_blocks, const float *x, uint32_t row, float *out, unsigned long long *clk)
{
    const size_t lds = (size_t)kRingBlocks * kB * kWave * MODE * sizeof(float);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_kernel<MODE, ROLES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((pair_kernel<MODE, ROLES>), dim3(wgs), dim3(2 * kWave), lds, 0, x, row, n_blocks, 1.9e-5f, 3.05e-5f, 5.0e-3f, out, clk);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    }
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * wgs);
    CHECK(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double t = 0;
    for (uint32_t i = 0; i < wgs; ++i) t += (double)h[2 * i];
    t /= wgs;
    const double ch = 64.0 * MODE, samples = (double)n_blocks * kB * ch * wgs;
    printf("%d channel%s per lane, %s  %5u workgroups  %.3f ms  %7.1f Gsample/s | %.0f clk per 36-sample block = %.2f clk per sample of 64 channels\n",
           MODE, MODE == 1 ? " " : "s", ROLES == 1 ? "T alone (loads + DC blocker)" : (ROLES == 2 ? "S alone (AGC in place)      " : "T and S                     "),
           wgs, ms, samples / ms * 1e-6, t / n_blocks, t / n_blocks / kB / MODE);
}

int main()
{
    const uint32_t n_blocks = 300, row = 2048 * 64 * 2;       // 2 048 workgroups x 64 lanes x 2 channels
    float *x, *out; unsigned long long *clk;
    const size_t rows = (size_t)n_blocks * kB;
    CHECK(hipMalloc(&x, rows * row * sizeof(float)));
    CHECK(hipMemset(x, 0x3c, rows * row * sizeof(float)));
    CHECK(hipMalloc(&out, (size_t)row * 2 * sizeof(float)));
    CHECK(hipMalloc(&clk, 2 * 8192 * sizeof(unsigned long long)));
    // 1 024 SIMDs and two wavefronts per workgroup: 512 workgroups = one wavefront per SIMD, 1 024 = two, 1 536 = three (the shipped kernel)
    for (uint32_t wgs : {512u, 1024u, 1536u}) {
        run<1, 1>(wgs, n_blocks, x, row, out, clk); run<1, 2>(wgs, n_blocks, x, row, out, clk); run<1, 3>(wgs, n_blocks, x, row, out, clk);
        run<2, 1>(wgs / 2, n_blocks, x, row, out, clk); run<2, 2>(wgs / 2, n_blocks, x, row, out, clk); run<2, 3>(wgs / 2, n_blocks, x, row, out, clk);   // the same channels on half the wavefronts
    }
    printf("(clk per sample of 64 channels: wavefront-clocks per workgroup-sample, the unit of DESIGN.md's instruction budgets; two channels per lane: per 64 of its 128 channels)\n");
    return 0;
}
