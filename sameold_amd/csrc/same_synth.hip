// same_synth.hip -- seeded synthetic multi-channel SAME/AFSK workload, generated on the
// device so that benchmark inputs are resident in HBM without crossing PCIe.
//
// Per channel: a random lead-in of silence, then a
// repeating schedule  H gap H gap H gap E gap E gap E gap2  where H = 16 x 0xAB preamble +
// a random valid header, E = 16 x 0xAB + "NNNN", gap = 1.0 s and gap2 = 2.0 s of zeros.
// Continuous-phase AFSK (mark 2083.3 Hz, space 1562.5 Hz) at the true 520.83 baud with a
// per-channel clock skew of up to +-0.25 %, amplitude uniform in [2000, 30000] (i16-range
// values stored as f32, which is what the receiver's AGC is tuned for, lib.rs:78-81).
// With flag bit 0 the symbol length is the even integer the reference's own test
// modulator uses (rx/waveform.rs:73-104: 42 samples at 22.05 kHz).
#include <hip/hip_runtime.h>

#include "same_launch.h"

namespace same {

__host__ __device__ inline uint64_t splitmix64(uint64_t &s)
{
    s += 0x9e3779b97f4a7c15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// header text of one channel: "ZCZC-ORG-EEE-PSSCCC(-PSSCCC)*+TTTT-JJJHHMM-LLLLLLLL-"
__host__ __device__ inline uint32_t build_payload(uint64_t seed, uint32_t channel, uint8_t *out, uint32_t cap)
{
    uint64_t s = seed ^ (0xd1b54a32d192ed03ull * (uint64_t)(channel + 1));
    const char *orgs = "EASCIVWXRPEP";
    uint32_t n = 0;
    auto put = [&](uint8_t b) { if (n < cap) out[n] = b; ++n; };
    put('Z'); put('C'); put('Z'); put('C'); put('-');
    uint32_t o = (uint32_t)(splitmix64(s) % 4);
    put((uint8_t)orgs[3 * o]); put((uint8_t)orgs[3 * o + 1]); put((uint8_t)orgs[3 * o + 2]); put('-');
    uint64_t r = splitmix64(s);
    for (int i = 0; i < 3; ++i) { put((uint8_t)('A' + r % 26)); r /= 26; }
    uint32_t nloc = 1 + (uint32_t)(splitmix64(s) % 6);
    for (uint32_t l = 0; l < nloc; ++l) {
        put('-');
        r = splitmix64(s);
        for (int i = 0; i < 6; ++i) { put((uint8_t)('0' + r % 10)); r /= 10; }
    }
    put('+');
    r = splitmix64(s);
    for (int i = 0; i < 4; ++i) { put((uint8_t)('0' + r % 10)); r /= 10; }
    put('-');
    for (int i = 0; i < 7; ++i) { put((uint8_t)('0' + r % 10)); r /= 10; }
    put('-');
    r = splitmix64(s);
    const char *cs = "ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789/";
    for (int i = 0; i < 8; ++i) { put((uint8_t)cs[r % 37]); r /= 37; }
    put('-');
    return n;
}

uint32_t synth_payload(uint64_t seed, uint32_t channel, uint8_t *out, uint32_t cap)
{ return build_payload(seed, channel, out, cap); }

__global__ __launch_bounds__(kWave) void synth_kernel(SynthParams sp, float *__restrict__ x, size_t n_samples)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sp.n_channels) return;
    const uint32_t C = sp.n_channels;
    uint8_t hdr[128];
    const uint32_t hlen = build_payload(sp.seed, c, hdr, sizeof(hdr));
    uint64_t s = sp.seed ^ (0x2545f4914f6cdd1dull * (uint64_t)(c + 1));
    const double fs = (double)sp.input_rate;
    const double lead = (double)(splitmix64(s) % 1000000) * 1e-6 * fs;          // U(0, 1 s)
    const float amp = 2000.0f + 28000.0f * (float)(splitmix64(s) % 65536) * (1.0f / 65536.0f);
    const double skew = ((double)(splitmix64(s) % 2001) - 1000.0) * 2.5e-6;     // +-0.25 %
    double sps = fs / (520.83 * (1.0 + skew));
    if (sp.flags & 1u) {
        int il = (int)floor(fs / 520.83);
        sps = (double)((il % 2) ? il + 1 : il);
    }
    const uint32_t dphi_mark = (uint32_t)llround(4294967296.0 * 2083.3 / fs);
    const uint32_t dphi_space = (uint32_t)llround(4294967296.0 * 1562.5 / fs);
    const uint32_t hbytes = 16 + hlen, ebytes = 16 + 4;
    const double gap = fs, gap2 = 2.0 * fs;
    const double hdur = (double)hbytes * 8.0 * sps, edur = (double)ebytes * 8.0 * sps;
    const double cycle = 3.0 * (hdur + gap) + 2.0 * (edur + gap) + edur + gap2;

    uint32_t phase = 0;
    uint64_t ns = sp.seed * 0x9e3779b97f4a7c15ull + (uint64_t)c * 0x632be59bd9b4e019ull;
    for (size_t t = 0; t < n_samples; ++t) {
        float v = 0.0f;
        double tt = (double)t - lead;
        if (tt >= 0.0) {
            double u = tt - cycle * floor(tt / cycle);
            // locate the burst (if any) that contains u
            int kind = -1;          // 0 header, 1 eom
            double start = 0.0, pos = 0.0;
            for (int b = 0; b < 6; ++b) {
                const double dur = (b < 3) ? hdur : edur;
                if (u >= pos && u < pos + dur) { kind = (b < 3) ? 0 : 1; start = pos; break; }
                pos += dur + ((b == 5) ? gap2 : gap);
            }
            if (kind >= 0) {
                uint32_t sym = (uint32_t)floor((u - start) / sps);
                uint32_t bi = sym >> 3;
                uint8_t byte;
                if (bi < 16) byte = 0xab;
                else if (kind == 0) byte = hdr[min(bi - 16, hlen - 1)];
                else byte = 'N';
                const uint32_t bit = (byte >> (sym & 7u)) & 1u;
                phase += bit ? dphi_mark : dphi_space;
                v = amp * cospif((float)phase * (1.0f / 2147483648.0f));
            } else {
                phase = 0;
            }
        }
        if (sp.noise_sigma > 0.0f) {
            uint64_t r = splitmix64(ns);
            float u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);
            float u2 = (float)(uint32_t)((r >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
            float g = sqrtf(-2.0f * __logf(u1)) * cospif(2.0f * u2);
            v += sp.noise_sigma * amp * g;
        }
        x[t * C + c] = v;
    }
}

// ---------------------------------------------------------------------------------
// AWGN Monte-Carlo trials (BASELINE.json configs[4], SURVEY.md section 8d "Config 5")
// ---------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11):
// counter-based, so sample block b of trial t is a pure function of (seed, t, b).
__device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// One burst per trial: 0.1 s + U(0, one symbol) of noise-only lead-in, 16 x 0xAB + the header
// of "channel" t (same_synth_payload(seed, t)), then noise to the end of the buffer.  Trial t
// runs at Eb/N0 = lo + (t mod n_grid) * step dB.  Eb = A^2/2 * Tb and N0 = 2 sigma^2 / fs for
// real white noise sampled at fs, so sigma = A * sqrt(sps / (4 * EbN0)).
__global__ __launch_bounds__(kWave) void trials_kernel(TrialParams tp, float *__restrict__ x, size_t n_samples)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= tp.n_trials) return;
    const uint32_t C = tp.n_trials;
    const uint32_t trial = tp.first_trial + c;
    uint8_t hdr[128];
    const uint32_t hlen = build_payload(tp.seed, trial, hdr, sizeof(hdr));
    uint64_t s = tp.seed ^ (0x2545f4914f6cdd1dull * (uint64_t)(trial + 1));
    const double fs = (double)tp.input_rate;
    const float amp = 2000.0f + 28000.0f * (float)(splitmix64(s) % 65536) * (1.0f / 65536.0f);
    const double skew = ((double)(splitmix64(s) % 2001) - 1000.0) * 2.5e-6;     // +-0.25 %
    const double sps = fs / (520.83 * (1.0 + skew));
    const double lead = 0.1 * fs + (double)(splitmix64(s) % 65536) * (1.0 / 65536.0) * sps;
    const float ebn0_db = tp.ebn0_db_lo + (float)(trial % tp.n_grid) * tp.ebn0_db_step;
    const float ebn0 = exp10f(0.1f * ebn0_db);
    const float sigma = amp * sqrtf((float)(fs / 520.83) / (4.0f * ebn0));
    const uint32_t dphi_mark = (uint32_t)llround(4294967296.0 * 2083.3 / fs);
    const uint32_t dphi_space = (uint32_t)llround(4294967296.0 * 1562.5 / fs);
    const double dur = (double)(16 + hlen) * 8.0 * sps;
    uint32_t phase = 0;
    float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t t = 0; t < n_samples; ++t) {
        float v = 0.0f;
        const double u = (double)t - lead;
        if (u >= 0.0 && u < dur) {
            const uint32_t sym = (uint32_t)floor(u / sps);
            const uint32_t bi = sym >> 3;
            const uint8_t byte = bi < 16 ? 0xab : hdr[min(bi - 16, hlen - 1)];
            phase += ((byte >> (sym & 7u)) & 1u) ? dphi_mark : dphi_space;
            v = amp * cospif((float)phase * (1.0f / 2147483648.0f));
        }
        if ((t & 3) == 0) {
            uint32_t ctr[4] = {(uint32_t)(t >> 2), (uint32_t)((uint64_t)t >> 34), trial, 0x5a4d4531u};
            philox4x32_10(ctr, (uint32_t)tp.seed, (uint32_t)(tp.seed >> 32));
            // Box-Muller in f32: two uniforms -> two normals
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float u1 = ((float)(ctr[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
                const float u2 = (float)(ctr[2 * p + 1] >> 8) * (1.0f / 16777216.0f);        // [0, 1)
                const float r = sqrtf(-2.0f * logf(u1));
                g[2 * p] = r * cospif(2.0f * u2);
                g[2 * p + 1] = r * sinpif(2.0f * u2);
            }
        }
        x[t * C + c] = v + sigma * g[t & 3];
    }
}

hipError_t launch_trials(const TrialParams &tp, float *x, size_t n_samples, hipStream_t stream)
{
    const uint32_t grid = (tp.n_trials + kWave - 1) / kWave;
    hipLaunchKernelGGL(trials_kernel, dim3(grid), dim3(kWave), 0, stream, tp, x, n_samples);
    return hipGetLastError();
}

hipError_t launch_synth(const SynthParams &sp, float *x, size_t n_samples, hipStream_t stream)
{
    const uint32_t grid = (sp.n_channels + kWave - 1) / kWave;
    hipLaunchKernelGGL(synth_kernel, dim3(grid), dim3(kWave), 0, stream, sp, x, n_samples);
    return hipGetLastError();
}

}  // namespace same
