// same_fast_common.h -- device code shared by the latency-optimised kernels
// (same_kernels_fast.hip: one wavefront per 64 channels; same_kernels_pipe.hip: a two-stage
// wavefront pipeline per 64 channels): block geometry, the register-resident equalizer
// context, the matched filters over the LDS window, the AGC step.
#pragma once

#include <hip/hip_runtime.h>

#include "same_dev_common.h"
#include "same_device.h"
#include "same_profile.h"

namespace same {

typedef float float2v __attribute__((ext_vector_type(2)));

// Block length of the fast kernel: 16 samples, or 18 for the mirrored-window variant.  Both
// are below the shortest interval between two TED instants at the standard rates (see
// choose_block_len: 18 is the bound at 22.05 kHz), so a block holds at most one instant.
constexpr int kBlock = 16;
constexpr int kBlockMirror = 18;
// The 22.05 kHz wavefront pipeline runs 20-sample blocks: instants are at least 19.45 samples apart
// there (the bound above), so a block can hold a SECOND instant when the timing loop runs at its
// fastest -- never a third -- and stage 2 handles that rare one on the spot (same_kernels_pipe.hip).
// Of the two, exactly one completes a symbol (the TED alternates), so everything downstream still
// sees at most one symbol per block.
constexpr int kBlockPipe22 = 20;
// 48 kHz (92 taps) and 44.1 kHz (84 taps): instants are 46 / 42 samples apart and the bound is
// 43 / 39, so the block is 32 -- the block-rate passes (the long filters, timing loop, symbol
// path) serve twice the samples.
constexpr int kBlock48k = 32;
template <int NT, bool MIRROR> struct FastBlock {
    static constexpr int len = MIRROR ? kBlockMirror : (NT >= 84 ? kBlock48k : kBlock);
};
// LDS window ring: a whole number of blocks (a block never wraps); a power of two when the
// per-tap address wraps with a mask, four blocks when the window is mirrored
template <int NT, bool MIRROR> struct FastRing {
    static constexpr int B = FastBlock<NT, MIRROR>::len;
    static constexpr int slots = MIRROR ? 4 * B : ((NT + B - 1 <= 64) ? 64 : 128);
};


template <int NFF, int NFB>
struct FastCtx : TickRingGlobal, ProfMarks {
    float *hist;                       // this lane's column of the squelch history: slot i at hist[i * hstride]
    uint32_t hstride = kWave;          // (LDS: [64][64]; the dense one-wavefront build reads the state array itself, stride = channels)
    float ffc[NFF], ffw[NFF], fbc[NFB], fbw[NFB];
    float sffc[NFF], sffw[NFF], sfbc[NFB], sfbw[NFB];   // equalizer at the last completed byte
    __device__ __forceinline__ void emit(const Params &P, const State &S, const Output &O, uint32_t c, uint32_t kind,
                                         uint64_t sample_counter, uint64_t symbols, uint32_t burst_len)
    { emit_event(P, S, O, c, kind, sample_counter, symbols, burst_len); }
    __device__ __forceinline__ void eq_snapshot(const Params &)
    {
#pragma unroll
        for (int i = 0; i < NFF; ++i) { sffc[i] = ffc[i]; sffw[i] = ffw[i]; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { sfbc[i] = fbc[i]; sfbw[i] = fbw[i]; }
    }
    __device__ __forceinline__ void eq_restore(const Params &)
    {
#pragma unroll
        for (int i = 0; i < NFF; ++i) { ffc[i] = sffc[i]; ffw[i] = sffw[i]; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { fbc[i] = sfbc[i]; fbw[i] = sfbw[i]; }
    }
    __device__ __forceinline__ void hist_put(uint32_t slot, float v) { hist[(size_t)slot * hstride] = v; }
    __device__ __forceinline__ float hist_get(uint32_t slot) const { return hist[(size_t)slot * hstride]; }
    __device__ __forceinline__ uint32_t eq_symbols(const Params &P, Lane &L, const float *samples, int nsym)
    {
        uint32_t bits = 0;
#pragma unroll 1
        for (int b = 0; b < nsym; ++b)
            bits |= eq_symbol_core<NFF, NFB>(P, L, ffc, ffw, fbc, fbw, samples[2 * b], samples[2 * b + 1]) << b;
        return bits;
    }
    __device__ __forceinline__ uint32_t eq_symbol1(const Params &P, Lane &L, float in0, float in1)
    { return eq_symbol_core<NFF, NFB>(P, L, ffc, ffw, fbc, fbw, in0, in1); }
    __device__ __forceinline__ void eq_reset(const Params &)
    {
        // Equalizer::reset rx/equalize.rs:191-196
#pragma unroll
        for (int i = 0; i < NFF; ++i) { ffc[i] = (i == 0) ? 1.0f : 0.0f; ffw[i] = 0.0f; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { fbc[i] = (i == 0) ? 1.0f : 0.0f; fbw[i] = 0.0f; }
    }
};

// Sample count (since the last TED instant) at which the sample clock fires next:
// receiver.rs:352-353 tests rem = samples_until_next_ted - clock after each increment and
// fires when rem <= 0 || |rem| < 0.5, i.e. at the first count c > clock with
// fl(s - c) < 0.5.  fl(s - c) is non-increasing in c, so the first hit can be searched
// from just below s.
__device__ __forceinline__ int next_fire_count(float s, uint32_t clock)
{
    float f = floorf(s) - 1.0f;
    int c = (int)clock + 1;
    if (f > (float)c) c = (int)f;
    // from floor(s) - 1 the test passes within three counts; two select steps cover that
    // without a divergent loop, and the loop stays as the (never iterating) general case
    c += ((s - (float)c) < 0.5f) ? 0 : 1;
    c += ((s - (float)c) < 0.5f) ? 0 : 1;
    while (!((s - (float)c) < 0.5f)) ++c;
    return c;
}

// Window storage in LDS.  The ring position is relative to the first sample of the launch, so
// a block of 16 pushes starts at a multiple of 16 and never wraps inside the block (immediate
// offsets from one base address).  MIRROR: every sample is stored at slot s and s + RING, and
// the filters read tap i at slot (newest + RING) - i, which never wraps either -- the per-tap
// address arithmetic (2-3 VALU per tap, 42 taps) disappears at the price of 16 KB of LDS.
template <int NT, int RING, bool MIRROR>
__device__ __forceinline__ float demod_fast(const float4 *tlds, const float *wring, uint32_t lane,
                                            uint32_t newest)
{
    // FskDemod::demod_now rx/demod.rs:156-164 over multiply_accumulate rx/filter.rs:363-377:
    // acc += window[newest - i] * h[i], i = 0 first; (mark.re, mark.im) and (space.re, space.im)
    // ride in the two halves of packed f32 operations (per-element IEEE: same roundings).
    //
    // Taps live in LDS and are read with a wave-uniform address (a broadcast, no bank
    // conflict): the scalar-cache round trip of s_load'ing 4*CH tap words per chunk was the
    // largest s_waitcnt item of the kernel, and holding all 4*NT words in SGPRs spills them.
    // The chunk loop stays rolled so the tap registers of one chunk are reused.
    constexpr int CH = 14;
    constexpr uint32_t PITCH = kWave * 4u;
    constexpr uint32_t WRAP = (uint32_t)RING * PITCH - 1u;
    float2v am = {0.0f, 0.0f}, as = {0.0f, 0.0f};
    // !MIRROR: byte address of tap i is slot (newest - i) mod RING, 256 bytes per slot, so
    // stepping back one tap is "subtract 256, wrap at RING*256" (lane*4 < 256 stays intact)
    uint32_t addr = newest * PITCH + lane * 4u;
    const char *wbase = reinterpret_cast<const char *>(wring);
    // MIRROR: tap i of the current chunk sits at wm[(CH - 1 - i) * 64]
    const float *wm = wring + ((int)newest + RING - (CH - 1)) * (int)kWave + (int)lane;
#pragma unroll 1
    for (int base = 0; base + CH <= NT; base += CH) {
        float w[CH];
        float4 h[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (MIRROR) {
                w[j] = wm[(CH - 1 - j) * (int)kWave];
            } else {
                w[j] = *reinterpret_cast<const float *>(wbase + addr);
                addr = (addr - PITCH) & WRAP;
            }
        }
        if (MIRROR) wm -= CH * (int)kWave;
#pragma unroll
        for (int j = 0; j < CH; ++j) h[j] = tlds[base + j];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float2v x2 = {w[j], w[j]};
            const float2v hm = {h[j].x, h[j].y}, hs = {h[j].z, h[j].w};
            const float2v pm = x2 * hm, ps = x2 * hs;
            am += pm; as += ps;
        }
    }
    constexpr int REM = NT % CH;
    if (REM) {
        float w[REM ? REM : 1];
#pragma unroll
        for (int j = 0; j < REM; ++j) {
            if (MIRROR) {
                w[j] = wm[(CH - 1 - j) * (int)kWave];
            } else {
                w[j] = *reinterpret_cast<const float *>(wbase + addr);
                addr = (addr - PITCH) & WRAP;
            }
        }
#pragma unroll
        for (int j = 0; j < REM; ++j) {
            const float4 t = tlds[NT - REM + j];
            const float2v x2 = {w[j], w[j]};
            const float2v hm = {t.x, t.y}, hs = {t.z, t.w};
            const float2v pm = x2 * hm, ps = x2 * hs;
            am += pm; as += ps;
        }
    }
    float d = rs_hypot(am.x, am.y) - rs_hypot(as.x, as.y);
    return rs_clamp(d, -1.0f, 1.0f);
}

// One AGC step, rx/agc.rs:72-77:  gain += (!locked as f32) * (1 - |out|) * bandwidth, clamped.
// `bw_eff` is the bandwidth for an unlocked AGC and 0.0 for a locked one: (1*e)*bw == e*bw
// exactly, and (0*e)*bw and e*0 are both a zero whose sign cannot matter because the gain is
// never -0.0 (it is a clamp output or the sum of a non-negative-zero gain and an update).
// MED3: v_med3_f32 is bit-identical to f32::clamp for every non-NaN gain unless a bound is
// -0.0 (the host checks the bounds); otherwise the compare/select form is used.
template <bool MED3>
__device__ __forceinline__ float agc_step(const Params &P, float y, float &gain, float bw_eff)
{
    float out = y * gain;
    float e = 1.0f - fabsf(out);
    float upd = e * bw_eff;
    gain += upd;
    if (MED3) gain = __builtin_amdgcn_fmed3f(gain, P.agc_min, P.agc_max);
    else gain = rs_clamp(gain, P.agc_min, P.agc_max);
    return out;
}

}  // namespace same
