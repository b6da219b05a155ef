// same_kernels_relaxed.hip -- the "fast mode" demodulation kernel: relaxed arithmetic, built for throughput.
//
// The strict kernels (same_kernels.hip, same_kernels_fast.hip, same_kernels_pipe.hip) reproduce the reference's
// scalar f32 evaluation order operation for operation, and pay for it: a 42-deep dependent accumulate per matched
// filter, an f64 square root per magnitude, IEEE divisions, five dependent operations per AGC sample.  This kernel
// keeps the reference's ALGORITHM -- every stage, every decision, every feedback path at the sample it happens --
// and gives up only the rounding of the floating-point expressions (SURVEY.md section 7 hard part 1(b), the mode
// BASELINE.json's north_star describes):
//
//   * matched filters (rx/filter.rs:363-377, rx/demod.rs:156-164): fused multiply-adds into four independent
//     partial sums per tone pair instead of one newest-first chain of separately rounded products;
//   * |mark|, |space| (Complex::norm = hypot, rx/demod.rs:163): sqrt(fma(re, re, im * im)) in f32 (v_sqrt_f32);
//   * AGC (rx/agc.rs:72-77): gain * (1 - bw * |x|) + bw, the same update algebraically (gain >= 0), as one
//     fused multiply-add on the gain's dependency chain -- two dependent operations per sample instead of five;
//   * timing loop (rx/symsync.rs:219-244) and equalizer (rx/equalize.rs:249-386): fused multiply-adds, the
//     divisions as v_rcp_f32 times a product.
// The DC blocker (rx/dcblock.rs:45-49, 104-108), the squelch (rx/codesquelch.rs:228-304) and the framer
// (rx/framing.rs:109-164) are the strict code: they gain nothing from re-association.
//
// Parity contract (include/same_rx.h, SAME_BATCH_RELAXED; tests/test_relaxed.py, tests/test_time_parallel.py):
// the rounding-chaotic timing trajectory is no longer the reference's to the bit, so burst bytes that were
// transmitted and transport messages are EQUAL, link events lie within SAME_TP_EVENT_TOLERANCE_SYMBOLS symbols,
// soft symbols of an open squelch within 0.05 with equal sign.  Strict mode stays the bit-exact form.
//
// Shape: one wavefront owns 64 state columns for the whole launch (no pipeline, no mailboxes, no barriers), blocks
// of 42 samples at 22.05 kHz = two sub-blocks of 21: per sub-block the sample phase (DC blocker, AGC, window push)
// runs for all lanes, then every lane with a TED instant in it (one, as a rule: instants are 21.17 samples apart)
// runs matched filters, timing loop and -- every other instant -- the symbol path.  The window lives in an LDS ring
// of three sub-blocks whose first 13 slots are stored twice, so a 14-tap chunk of the filter never wraps; the
// squelch's sample history stays in the HBM state arrays (two loads per sub-block, issued ahead).  19.5 KB of LDS and
// at most 256 registers per wavefront: eight wavefronts per CU, two per SIMD.
//
// Time-parallel chunks (PipeChunks) are taken exactly as the pipeline kernel takes them.
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>

#include "same_dev_common.h"
#include "same_device.h"
#include "same_fast_common.h"
#include "same_launch.h"
#include "same_profile.h"

namespace same {

// Geometry per sample rate (filter length NT): DC-blocker window, samples per sub-block (just below the nominal
// distance of two TED instants, so that nearly every lane has exactly one per sub-block), sub-blocks per block.
template <int NT> struct RelaxGeom;
template <> struct RelaxGeom<42> { static constexpr int DCL = 16, SB = 21; };     // 22.05 kHz: instants 21.17 apart
constexpr int kRelaxChunk = 14;                 // taps per filter chunk; the ring's first kRelaxChunk - 1 slots are mirrored
template <int NT> struct RelaxLayout {
    static constexpr int SB = RelaxGeom<NT>::SB, B = 2 * SB, DCL = RelaxGeom<NT>::DCL;
    static constexpr int RING = 3 * SB, MIRROR = kRelaxChunk - 1;
    static constexpr uint32_t tap_floats = (uint32_t)((NT * 4 + 63) / 64 * 64);
    static constexpr size_t lds_bytes = ((size_t)tap_floats + (size_t)(RING + MIRROR) * kWave) * sizeof(float);
    static_assert(NT - 1 + SB <= RING, "a filter at the start of a sub-block reaches into the sub-block being written");
    static_assert(NT % kRelaxChunk == 0, "whole chunks");
    static_assert(SB >= DCL && SB - DCL >= 1, "the DC windows are the tail of the previous sub-block");
};

// acc += {w.lo, w.lo} * h   /   acc += {w.hi, w.hi} * h: packed f32 FMA with the window sample broadcast by op_sel
// (the compiler would materialise the splat with a v_mov per tap).  volatile: they stay in program order between the
// load statements of demod_relaxed.
__device__ __forceinline__ void pk_fma_lo(float2v &acc, float2v w, float2v h)
{ asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(h)); }
__device__ __forceinline__ void pk_fma_hi(float2v &acc, float2v w, float2v h)
{ asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h)); }

typedef float float4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ uint32_t lds_addr(const float *p) { return (uint32_t)(uintptr_t)(const lds_float *)p; }

// FskDemod::demod_now rx/demod.rs:156-164 at the instant whose sample sits in ring slot `newest`.  Tap j multiplies
// slot (newest - j) mod RING; a chunk of 14 taps is read upwards from its lowest slot, moved into the mirror when it
// would wrap.  Even / odd taps accumulate separately: four independent chains of 21 fused multiply-adds.
//
// The loads of a chunk -- 7 x ds_read2st64_b32 (two window slots each) and 14 x ds_read_b128 (one tap: mark re/im,
// space re/im; a wave-uniform address, i.e. a broadcast) -- are issued back to back from asm statements.  Left to
// itself the compiler loads two taps, waits, multiplies, loads the next two ...: fourteen exposed LDS round trips per
// chunk.  WIDE (the build with a SIMD to itself, 512 registers): all 21 loads at once, the first half's products
// under the second half's latency (70 registers of loads); otherwise two rounds of 7 taps (36 registers).
template <int NT, int RING, bool WIDE>
__device__ __forceinline__ float demod_relaxed(uint32_t taps_lds, uint32_t wcol_lds, int newest)
{
    constexpr int CH = kRelaxChunk;
    static_assert(CH == 14, "the load sequence below is written out for 14 taps");
    float2v am0 = {0.0f, 0.0f}, am1 = {0.0f, 0.0f}, as0 = {0.0f, 0.0f}, as1 = {0.0f, 0.0f};
    int top = newest;
#define RELAX_TAP(acc_m, acc_s, fma, win_, tap_) do { const float2v hm_ = {tap_.x, tap_.y}, hs_ = {tap_.z, tap_.w}; fma(acc_m, win_, hm_); fma(acc_s, win_, hs_); } while (0)
#pragma unroll 1
    for (int base = 0; base < NT; base += CH) {
        int s = top;
        s += s < 0 ? RING : 0;
        s += s < CH - 1 ? RING : 0;                          // slots RING .. RING + CH - 2 repeat slots 0 .. CH - 2
        const uint32_t wa = wcol_lds + (uint32_t)(s - (CH - 1)) * (kWave * 4u);   // lowest slot of the chunk = tap base + 13
        const uint32_t ta = taps_lds + (uint32_t)base * 16u;
        float2v w0, w1, w2, w3, w4, w5, w6;                  // wK = {tap base + 13 - 2K, tap base + 12 - 2K}
        float4v t0, t1, t2, t3, t4, t5, t6;
        if constexpr (WIDE) {
            float4v t7, t8, t9, t10, t11, t12, t13;
            asm volatile(
                "ds_read2st64_b32 %[w0], %[wa] offset1:1\n\t"
                "ds_read2st64_b32 %[w1], %[wa] offset0:2 offset1:3\n\t"
                "ds_read2st64_b32 %[w2], %[wa] offset0:4 offset1:5\n\t"
                "ds_read2st64_b32 %[w3], %[wa] offset0:6 offset1:7\n\t"
                "ds_read_b128 %[t13], %[ta] offset:208\n\t"
                "ds_read_b128 %[t12], %[ta] offset:192\n\t"
                "ds_read_b128 %[t11], %[ta] offset:176\n\t"
                "ds_read_b128 %[t10], %[ta] offset:160\n\t"
                "ds_read_b128 %[t9], %[ta] offset:144\n\t"
                "ds_read_b128 %[t8], %[ta] offset:128\n\t"
                "ds_read_b128 %[t7], %[ta] offset:112\n\t"
                "ds_read2st64_b32 %[w4], %[wa] offset0:8 offset1:9\n\t"
                "ds_read2st64_b32 %[w5], %[wa] offset0:10 offset1:11\n\t"
                "ds_read2st64_b32 %[w6], %[wa] offset0:12 offset1:13\n\t"
                "ds_read_b128 %[t6], %[ta] offset:96\n\t"
                "ds_read_b128 %[t5], %[ta] offset:80\n\t"
                "ds_read_b128 %[t4], %[ta] offset:64\n\t"
                "ds_read_b128 %[t3], %[ta] offset:48\n\t"
                "ds_read_b128 %[t2], %[ta] offset:32\n\t"
                "ds_read_b128 %[t1], %[ta] offset:16\n\t"
                "ds_read_b128 %[t0], %[ta]\n\t"
                "s_waitcnt lgkmcnt(10)"
                : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [w5] "=&v"(w5), [w6] "=&v"(w6),
                  [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6),
                  [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9), [t10] "=&v"(t10), [t11] "=&v"(t11), [t12] "=&v"(t12), [t13] "=&v"(t13)
                : [wa] "v"(wa), [ta] "v"(ta)
                : "memory");
            RELAX_TAP(am0, as0, pk_fma_lo, w0, t13); RELAX_TAP(am1, as1, pk_fma_hi, w0, t12);
            RELAX_TAP(am0, as0, pk_fma_lo, w1, t11); RELAX_TAP(am1, as1, pk_fma_hi, w1, t10);
            RELAX_TAP(am0, as0, pk_fma_lo, w2, t9); RELAX_TAP(am1, as1, pk_fma_hi, w2, t8);
            RELAX_TAP(am0, as0, pk_fma_lo, w3, t7);
            // the second half has landed by now, as a rule; the values travel through the statement so that nothing that
            // reads them can be moved above it
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(w4), "+v"(w5), "+v"(w6), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6));
        } else {
            asm volatile(
                "ds_read2st64_b32 %[w0], %[wa] offset1:1\n\t"
                "ds_read2st64_b32 %[w1], %[wa] offset0:2 offset1:3\n\t"
                "ds_read2st64_b32 %[w2], %[wa] offset0:4 offset1:5\n\t"
                "ds_read2st64_b32 %[w3], %[wa] offset0:6 offset1:7\n\t"
                "ds_read_b128 %[t6], %[ta] offset:208\n\t"
                "ds_read_b128 %[t5], %[ta] offset:192\n\t"
                "ds_read_b128 %[t4], %[ta] offset:176\n\t"
                "ds_read_b128 %[t3], %[ta] offset:160\n\t"
                "ds_read_b128 %[t2], %[ta] offset:144\n\t"
                "ds_read_b128 %[t1], %[ta] offset:128\n\t"
                "ds_read_b128 %[t0], %[ta] offset:112\n\t"
                "s_waitcnt lgkmcnt(0)"
                : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3),
                  [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6)
                : [wa] "v"(wa), [ta] "v"(ta)
                : "memory");
            RELAX_TAP(am0, as0, pk_fma_lo, w0, t6); RELAX_TAP(am1, as1, pk_fma_hi, w0, t5);
            RELAX_TAP(am0, as0, pk_fma_lo, w1, t4); RELAX_TAP(am1, as1, pk_fma_hi, w1, t3);
            RELAX_TAP(am0, as0, pk_fma_lo, w2, t2); RELAX_TAP(am1, as1, pk_fma_hi, w2, t1);
            RELAX_TAP(am0, as0, pk_fma_lo, w3, t0);
            asm volatile(
                "ds_read2st64_b32 %[w4], %[wa] offset0:8 offset1:9\n\t"
                "ds_read2st64_b32 %[w5], %[wa] offset0:10 offset1:11\n\t"
                "ds_read2st64_b32 %[w6], %[wa] offset0:12 offset1:13\n\t"
                "ds_read_b128 %[t6], %[ta] offset:96\n\t"
                "ds_read_b128 %[t5], %[ta] offset:80\n\t"
                "ds_read_b128 %[t4], %[ta] offset:64\n\t"
                "ds_read_b128 %[t3], %[ta] offset:48\n\t"
                "ds_read_b128 %[t2], %[ta] offset:32\n\t"
                "ds_read_b128 %[t1], %[ta] offset:16\n\t"
                "ds_read_b128 %[t0], %[ta]\n\t"
                "s_waitcnt lgkmcnt(0)"
                : [w4] "=&v"(w4), [w5] "=&v"(w5), [w6] "=&v"(w6),
                  [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6)
                : [wa] "v"(wa), [ta] "v"(ta)
                : "memory");
        }
        RELAX_TAP(am1, as1, pk_fma_hi, w3, t6);
        RELAX_TAP(am0, as0, pk_fma_lo, w4, t5); RELAX_TAP(am1, as1, pk_fma_hi, w4, t4);
        RELAX_TAP(am0, as0, pk_fma_lo, w5, t3); RELAX_TAP(am1, as1, pk_fma_hi, w5, t2);
        RELAX_TAP(am0, as0, pk_fma_lo, w6, t1); RELAX_TAP(am1, as1, pk_fma_hi, w6, t0);
        top -= CH;
    }
#undef RELAX_TAP
    const float2v am = am0 + am1, as = as0 + as1;
    const float mm = __builtin_amdgcn_sqrtf(__builtin_fmaf(am.x, am.x, am.y * am.y));
    const float ms = __builtin_amdgcn_sqrtf(__builtin_fmaf(as.x, as.x, as.y * as.y));
    return __builtin_amdgcn_fmed3f(mm - ms, -1.0f, 1.0f);
}

// ZeroCrossingTed::input + TimingLoop::advance_loop (rx/symsync.rs:198-287), the relaxed form of ted_timing
__device__ __forceinline__ bool ted_timing_relaxed(const Params &P, Lane &L, float inv_spt, float sa_low, float rem,
                                                   float *zero_out, float *sym_out, float *terr_out)
{
    L.h0 = L.h1; L.h1 = L.h2; L.h2 = sa_low;
    L.flags ^= F_TED_PHASE;
    const bool have = (L.flags & F_TED_PHASE) != 0;
    const float dsg = rs_signum(L.h0) - rs_signum(L.h2);
    const float terr = L.h1 * dsg;
    const float offset = __builtin_amdgcn_fmed3f(rem, -0.5f, 0.5f);
    const float e = __builtin_amdgcn_fmed3f(__builtin_fmaf(-offset, inv_spt, terr), -1.0f, 1.0f);
    const bool bw_locked = (L.flags & F_BW_LOCKED) != 0;
    const float alpha = bw_locked ? P.alpha_locked : P.alpha_unlocked;
    const float beta = bw_locked ? P.beta_locked : P.beta_unlocked;
    const float avg1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(beta, e, L.period_avg), P.period_min, P.period_max);
    float inst1 = __builtin_fmaf(alpha, e, avg1) + offset;
    inst1 = (inst1 < 0.0f) ? avg1 : inst1;
    const float inst0 = L.period_inst + offset;
    L.period_avg = have ? avg1 : L.period_avg;
    L.period_inst = have ? inst1 : inst0;
    L.until_next_ted = L.period_inst;
    *zero_out = L.h1; *sym_out = L.h2; *terr_out = terr;
    return have;
}

// Equalizer::estimate_symbol + evolve (rx/equalize.rs:249-332, 354-386), the relaxed form of eq_symbol_core:
// fused multiply-adds, two partial sums per filter, v_rcp_f32 for the NLMS gains.
template <int NFF, int NFB>
__device__ __forceinline__ uint32_t eq_symbol_relaxed(const Params &P, Lane &L, float (&ffc)[NFF], float (&ffw)[NFF],
                                                      float (&fbc)[NFB], float (&fbw)[NFB], float in0, float in1)
{
    uint32_t mode = (L.flags & F_EQ_MODE_MASK) >> F_EQ_MODE_SHIFT;
    if (NFF >= 2) {
#pragma unroll
        for (int i = 0; i + 2 < NFF; ++i) ffw[i] = ffw[i + 2];
        ffw[NFF >= 2 ? NFF - 2 : 0] = in0;
        ffw[NFF - 1] = in1;
    } else {
        ffw[0] = in1;
    }
    float f0 = 0.0f, f1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
#pragma unroll
    for (int i = 0; i < NFF; ++i) {
        if (i & 1) { f1 = __builtin_fmaf(ffw[NFF - 1 - i], ffc[i], f1); q1 = __builtin_fmaf(ffw[i], ffw[i], q1); }
        else { f0 = __builtin_fmaf(ffw[NFF - 1 - i], ffc[i], f0); q0 = __builtin_fmaf(ffw[i], ffw[i], q0); }
    }
    // the feedback window holds an exact 0.0 in every other slot (push(&[decision, 0.0]), rx/equalize.rs:304): those
    // taps contribute nothing and are never updated (see eq_symbol_core)
    constexpr auto fb_zero = [](int widx) { return ((NFB - 1 - widx) & 1) == 0; };
    float fb = 0.0f, qb = 0.0f;
#pragma unroll
    for (int i = 0; i < NFB; ++i) {
        if (!fb_zero(NFB - 1 - i)) fb = __builtin_fmaf(fbw[NFB - 1 - i], fbc[i], fb);
        if (!fb_zero(i)) qb = __builtin_fmaf(fbw[i], fbw[i], qb);
    }
    const float sym_val = (f0 + f1) - fb;
    float sym_est, err;
    bool evolve = true;
    if (mode == 2u) {                                  // EnabledTraining :278-301
        sym_est = (L.eq_word & 1u) ? 1.0f : -1.0f;
        L.eq_word >>= 1;
        err = sym_est - sym_val;
        L.eq_count += 1;
        if (L.eq_count >= 32u) mode = 1u;
    } else if (mode == 1u) {                           // EnabledFeedback :266-277
        sym_est = rs_signum(sym_val);
        err = sym_est - sym_val;
    } else {                                           // Disabled :262-265
        sym_est = rs_signum(sym_val); err = 0.0f; evolve = false;
    }
    if (evolve) {
        const float gf = P.eq_relaxation * __builtin_amdgcn_rcpf(P.eq_regularization + (q0 + q1));
        const float gb = P.eq_relaxation * __builtin_amdgcn_rcpf(P.eq_regularization + qb);
        const float ge = gf * err, gn = -(gb * err);
#pragma unroll
        for (int i = 0; i < NFF; ++i) ffc[i] = __builtin_fmaf(ge, ffw[NFF - 1 - i], ffc[i]);
#pragma unroll
        for (int i = 0; i < NFB; ++i) { if (!fb_zero(NFB - 1 - i)) fbc[i] = __builtin_fmaf(gn, fbw[NFB - 1 - i], fbc[i]); }
    }
    if (NFB >= 2) {
#pragma unroll
        for (int i = 0; i + 2 < NFB; ++i) fbw[i] = fbw[i + 2];
        fbw[NFB >= 2 ? NFB - 2 : 0] = sym_est;
        fbw[NFB - 1] = 0.0f;
    } else {
        fbw[0] = 0.0f;
    }
    L.flags = (L.flags & ~F_EQ_MODE_MASK) | (mode << F_EQ_MODE_SHIFT);
    return sym_est >= 0.0f ? 1u : 0u;
}

// The symbol path's context: the equalizer (20 floats) in registers with the relaxed step; the squelch's sample
// history and the equalizer as of the last completed byte (written once per byte while a preamble is being acquired,
// read back on a byte-clock re-alignment: rare, and 20 registers) in the HBM state arrays.
template <int NFF, int NFB>
struct RelaxCtx : TickRingGlobal {
    const State *S;
    uint32_t c, C;
    float *hist;                       // this lane's column of S.sq_hist: slot i at hist[i * C]
    float ffc[NFF], ffw[NFF], fbc[NFB], fbw[NFB];
    __device__ __forceinline__ void mark(int) const {}
    __device__ __forceinline__ void emit(const Params &P, const State &St, const Output &O, uint32_t ch, uint32_t kind,
                                         uint64_t sample_counter, uint64_t symbols, uint32_t burst_len)
    { emit_event(P, St, O, ch, kind, sample_counter, symbols, burst_len); }
    __device__ __forceinline__ void hist_put(uint32_t slot, float v) { hist[(size_t)slot * C] = v; }
    __device__ __forceinline__ float hist_get(uint32_t slot) const { return hist[(size_t)slot * C]; }
    __device__ __forceinline__ void eq_snapshot(const Params &)
    {
#pragma unroll
        for (int i = 0; i < NFF; ++i) { S->eq_snap_ffc[i * C + c] = ffc[i]; S->eq_snap_ffw[i * C + c] = ffw[i]; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { S->eq_snap_fbc[i * C + c] = fbc[i]; S->eq_snap_fbw[i * C + c] = fbw[i]; }
    }
    __device__ __forceinline__ void eq_restore(const Params &)
    {
#pragma unroll
        for (int i = 0; i < NFF; ++i) { ffc[i] = S->eq_snap_ffc[i * C + c]; ffw[i] = S->eq_snap_ffw[i * C + c]; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { fbc[i] = S->eq_snap_fbc[i * C + c]; fbw[i] = S->eq_snap_fbw[i * C + c]; }
    }
    __device__ __forceinline__ void eq_reset(const Params &)
    {
#pragma unroll
        for (int i = 0; i < NFF; ++i) { ffc[i] = (i == 0) ? 1.0f : 0.0f; ffw[i] = 0.0f; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { fbc[i] = (i == 0) ? 1.0f : 0.0f; fbw[i] = 0.0f; }
    }
    __device__ __forceinline__ uint32_t eq_symbols(const Params &P, Lane &L, const float *samples, int nsym)
    {
        uint32_t bits = 0;
#pragma unroll 1
        for (int b = 0; b < nsym; ++b)
            bits |= eq_symbol_relaxed<NFF, NFB>(P, L, ffc, ffw, fbc, fbw, samples[2 * b], samples[2 * b + 1]) << b;
        return bits;
    }
    __device__ __forceinline__ uint32_t eq_symbol1(const Params &P, Lane &L, float in0, float in1)
    { return eq_symbol_relaxed<NFF, NFB>(P, L, ffc, ffw, fbc, fbw, in0, in1); }
};

// lane_store without the transport wake-up words (a launch without wake-ups never changes them, and never loading
// them keeps eight registers free)
__device__ __forceinline__ void lane_store_link(const Lane &L, const State &S, uint32_t c)
{
    S.dc_sum0[c] = L.sum0; S.dc_sum1[c] = L.sum1; S.agc_gain[c] = L.gain;
    S.until_next_ted[c] = L.until_next_ted; S.ted_clock[c] = L.ted_clock;
    S.ted_h0[c] = L.h0; S.ted_h1[c] = L.h1; S.ted_h2[c] = L.h2;
    S.period_avg[c] = L.period_avg; S.period_inst[c] = L.period_inst;
    S.sq_data[c] = L.sq_data; S.sq_power[c] = L.sq_power; S.sq_phist[c] = L.sq_phist;
    S.sq_fill[c] = L.sq_fill; S.sq_clock[c] = L.sq_clock; S.sq_symbols[c] = L.sq_symbols;
    S.eq_word[c] = L.eq_word; S.eq_count[c] = L.eq_count;
    S.fr_word[c] = L.fr_word; S.fr_count[c] = L.fr_count; S.fr_invalid[c] = L.fr_invalid;
    S.fr_len[c] = L.fr_len; S.flags[c] = L.flags;
}
__device__ __forceinline__ void lane_load_link(Lane &L, const State &S, uint32_t c)
{
    L.sum0 = S.dc_sum0[c]; L.sum1 = S.dc_sum1[c]; L.gain = S.agc_gain[c];
    L.until_next_ted = S.until_next_ted[c]; L.ted_clock = S.ted_clock[c];
    L.h0 = S.ted_h0[c]; L.h1 = S.ted_h1[c]; L.h2 = S.ted_h2[c];
    L.period_avg = S.period_avg[c]; L.period_inst = S.period_inst[c];
    L.sq_data = S.sq_data[c]; L.sq_power = S.sq_power[c]; L.sq_phist = S.sq_phist[c];
    L.sq_fill = S.sq_fill[c]; L.sq_clock = S.sq_clock[c]; L.sq_symbols = S.sq_symbols[c];
    L.eq_word = S.eq_word[c]; L.eq_count = S.eq_count[c];
    L.fr_word = S.fr_word[c]; L.fr_count = S.fr_count[c]; L.fr_invalid = S.fr_invalid[c];
    L.fr_len = S.fr_len[c]; L.flags = S.flags[c];
    L.tk_next = 0; L.tk_last = 0; L.wake_sample = 0; L.wake_fired = 0; L.ended = 0;
}

// CM: channel-major f32 input, every lane its own contiguous stream from its own first row (time-parallel launches
// with per-channel boundaries); otherwise time-major rows, read with buffer loads (wave-uniform row offset in an
// SGPR, the lane's column as the vector offset: no address arithmetic on the vector unit).
// TICKS: transport wake-ups on the device (P.ticks).  OCC: wavefronts per SIMD the build is sized for -- 2 for
// launches of more than one wavefront per SIMD (256 registers), 1 otherwise (512: the filter's loads all at once).
template <int NT, int NFF, int NFB, typename SampleT, bool CM, bool TICKS, int OCC>
__global__ __launch_bounds__(kWave, OCC) void demod_relaxed_kernel(Params P, State S, Output O,
                                                                   const float4 *__restrict__ taps,
                                                                   const SampleT *__restrict__ x,
                                                                   uint32_t n_blocks, uint64_t counter0, PipeChunks K)
{
    using LY = RelaxLayout<NT>;
    constexpr int SB = LY::SB, B = LY::B, DCL = LY::DCL, RING = LY::RING, MIR = LY::MIRROR;
    constexpr int HOFF = SB - DCL;                           // where the last DCL entries of a sub-block's array start
    constexpr uint32_t LP = kWave;
    static_assert(!CM || std::is_same<SampleT, float>::value, "channel-major streams are f32");
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x;
    float4 *tlds = reinterpret_cast<float4 *>(lds);
    for (uint32_t i = lane; i < (uint32_t)NT; i += kWave) tlds[i] = taps[i];
    __syncthreads();                                         // (one wavefront: orders the staging before the reads)
    const uint32_t C = P.n_channels;
    // state column of this lane (time-parallel launches may permute them: pieces of similar length share a wavefront)
    uint32_t c = blockIdx.x * kWave + lane;
    if (c >= C) return;                                      // no barriers below
    if (K.n_chunks > 1u && K.col_perm) c = K.col_perm[c];
    // Time-parallel chunks, as the pipeline kernel takes them: state column c = chunk * Cin + cin
    // reads input column cin from its own first row on.
    uint32_t cin = c, Cin = C, n_nominal = n_blocks;
    bool may_leave = false;
    int32_t row_l = 0;
    const float *xl = nullptr;                               // CM: this lane's own stream
    uint32_t avail_l = 0xffffffffu;
    if constexpr (CM) {
        Cin = K.in_channels;
        const uint32_t chunk_l = c / Cin;
        cin = c - chunk_l * Cin;
        // wavefronts are homogeneous: all chunk 0, all last chunk, or all in between (the planner sees to it)
        may_leave = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk_l) + 1u < K.n_chunks;
        const uint32_t row_abs = K.col_row0[c];
        xl = reinterpret_cast<const float *>(x) + (size_t)cin * K.in_samples + row_abs;
        avail_l = (K.whole_samples - row_abs) / (uint32_t)B;
        n_blocks = K.wg_blocks[blockIdx.x];
        n_nominal = may_leave ? K.col_nominal[c] : n_blocks;
        const uint32_t row_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_abs);
        counter0 += (uint64_t)row_first;
        row_l = (int32_t)(row_abs - row_first);
    } else if (K.n_chunks > 1u) {
        Cin = K.in_channels;
        const uint32_t wgs = K.in_channels / kWave;
        const uint32_t chunk = blockIdx.x / wgs;
        cin = (blockIdx.x - chunk * wgs) * kWave + lane;
        may_leave = chunk + 1u < K.n_chunks;
        const uint32_t first_block = chunk * K.stride_blocks;
        x += (size_t)first_block * B * Cin;
        counter0 += (uint64_t)first_block * B;
        n_blocks -= first_block;
        n_nominal = may_leave ? K.nominal_blocks : n_blocks;
    }
    float *wcol = lds + LY::tap_floats + lane;               // ring slot 0 of this lane
    const uint32_t taps_lds = lds_addr(lds), wcol_lds = lds_addr(wcol);

    Lane L;
    if constexpr (TICKS) lane_load(L, S, c); else lane_load_link(L, S, c);
    RelaxCtx<NFF, NFB> X;
    X.S = &S; X.c = c; X.C = C;
    X.hist = S.sq_hist + c;
#pragma unroll
    for (int i = 0; i < NFF; ++i) { X.ffc[i] = S.eq_ffc[i * C + c]; X.ffw[i] = S.eq_ffw[i * C + c]; }
#pragma unroll
    for (int i = 0; i < NFB; ++i) { X.fbc[i] = S.eq_fbc[i * C + c]; X.fbw[i] = S.eq_fbw[i * C + c]; }
    // window: sample counter0 - m (m = 1 ..) sits in the state's slot (counter0 - m) mod win_ring; here the launch's
    // first sample goes to ring slot 0, so it belongs in slot RING - m (and in the mirror when that is one of the
    // first MIR slots -- never: RING - m >= RING - (NT - 1) > MIR)
    {
        const uint32_t G = P.win_ring;
#pragma unroll 2
        for (uint32_t m = 1; m < (uint32_t)NT; ++m) {
            const uint32_t g = (uint32_t)(counter0 - (uint64_t)m) & (G - 1u);
            const float *row = S.win_ring + (size_t)g * C;
            wcol[((uint32_t)RING - m) * LP] = row[c];
        }
    }
    static_assert(RING - (NT - 1) > MIR, "the carried-over window would need mirroring");

    // Input registers: xa = samples 0 .. SB of a block (SB + 1 of them), xb = samples SB + 1 .. B - 1.  Sub-block 0
    // consumes xa[0 .. SB-1], sub-block 1 xa[SB] and xb; the DC blocker's input window (the last DCL inputs) is the
    // tail of the other sub-block's registers, so nothing is copied: xb is refilled after sub-block 0's DC phase (it
    // was that phase's history), xa after sub-block 1's.  ma / mb: the first moving average's outputs likewise.
    float xa[SB + 1], xb[SB - 1], ma[SB], mb[SB];
    {
        const uint32_t dpos = (uint32_t)(counter0 % (uint64_t)DCL);
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            uint32_t slot = dpos + (uint32_t)k;
            if (slot >= (uint32_t)DCL) slot -= (uint32_t)DCL;
            const float *r0 = S.dc_ff_ring + (size_t)slot * C, *r1 = S.dc_fb_ring + (size_t)slot * C;
            xb[k + HOFF - 1] = r0[c];
            mb[k + HOFF] = r1[c];
        }
    }
    // samples k0 .. k0 + N - 1 of block b into dst[0 .. N-1]
    const uint32_t voff = cin * (uint32_t)sizeof(SampleT), row_bytes = Cin * (uint32_t)sizeof(SampleT);
    auto load_rows = [&](float *dst, uint32_t b, auto k0_, auto n_) {
        constexpr int k0 = decltype(k0_)::value, N = decltype(n_)::value;
        if constexpr (CM) {
            // 8-byte loads from this lane's stream (blocks of 42 floats are 8-byte aligned); past its end: silence
            static_assert((k0 & 1) == 0 && (N & 1) == 0, "whole 8-byte words");
            if (b < avail_l) {
                const float2 *p = reinterpret_cast<const float2 *>(xl + (size_t)b * B) + k0 / 2;
#pragma unroll
                for (int j = 0; j < N / 2; ++j) { const float2 v = p[j]; dst[2 * j] = v.x; dst[2 * j + 1] = v.y; }
            } else {
#pragma unroll
                for (int k = 0; k < N; ++k) dst[k] = 0.0f;
            }
        } else {
            // buffer loads: the block's first row is the resource's base (re-based per block: a launch may exceed
            // the 4 GB a resource spans), row k at the scalar offset k * row_bytes
            const SampleT *xr = x + ((size_t)b * B + (uint32_t)k0) * Cin;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<SampleT *>(xr), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if constexpr (sizeof(SampleT) == 4) dst[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (uint32_t)k * row_bytes, 0));
                else dst[k] = (float)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, (uint32_t)k * row_bytes, 0);
            }
        }
    };
    auto load_a = [&](uint32_t b) { load_rows(xa, b, std::integral_constant<int, 0>{}, std::integral_constant<int, SB + 1>{}); };
    auto load_b = [&](uint32_t b) { load_rows(xb, b, std::integral_constant<int, SB + 1>{}, std::integral_constant<int, SB - 1>{}); };
    static_assert(((SB + 1) & 1) == 0, "8-byte loads: SB + 1 even");

    const float inv_spt = 1.0f / P.samples_per_ted;
    int cstar = next_fire_count(L.until_next_ted, L.ted_clock);
    int until = cstar - (int)L.ted_clock - 1;                // index of the next instant, relative to the sub-block about to run
    uint32_t wbase = 0;                                      // ring slot of the sub-block's first sample: 0, SB or 2 * SB
    bool lane_done = false;
    RX_T0();

    // One sub-block: DC blocker, AGC and window push of its SB samples, then every TED instant that falls into it.
    auto sub = [&](auto par, uint32_t blk) {
        constexpr int PAR = decltype(par)::value;
        auto in = [&](int k) -> float { if constexpr (PAR == 0) return xa[k]; else return k == 0 ? xa[SB] : xb[k - 1]; };
        // the input window before this sub-block, oldest first: entries 0 .. DCL-1, then this sub-block's inputs
        auto xw = [&](int i) -> float {
            if (i >= DCL) return in(i - DCL);
            if constexpr (PAR == 0) return xb[i + HOFF - 1]; else return xa[i + HOFF];
        };
        float (&mnew)[SB] = *(PAR == 0 ? &ma : &mb);
        float (&mold)[SB] = *(PAR == 0 ? &mb : &ma);
        auto mw = [&](int k) -> float { return k < DCL ? mold[k + HOFF] : mnew[k - DCL]; };
        // the two history samples the symbol's equalizer step takes (rx_symbol: slots +16 / +17 from the squelch's
        // write position), fetched a sub-block's worth of work ahead of their use; a sub-block completes at most
        // one symbol and only a symbol writes the history, so they cannot go stale
        const uint32_t pslot = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;
        const float hpre0 = X.hist_get((pslot + 16u) & 63u), hpre1 = X.hist_get((pslot + 17u) & 63u);

        // ---- DC blocker rx/dcblock.rs:45-49, 104-108 (strict: the reference's operations in its order) ----
        float ys[SB];
        float sum0 = L.sum0, sum1 = L.sum1;
        const float2v inv = {P.dc_inv_len, P.dc_inv_len};
#pragma unroll
        for (int k = 0; k + 1 < SB; k += 2) {
            const float2v x2 = {in(k), in(k + 1)}, xo = {xw(k), xw(k + 1)};
            const float2v d0 = x2 - xo;
            const float s0a = sum0 + d0.x, s0b = s0a + d0.y;
            sum0 = s0b;
            const float2v s0 = {s0a, s0b};
            const float2v m0 = s0 * inv;
            mnew[k] = m0.x; mnew[k + 1] = m0.y;
            const float2v sig = {xw(k + 1), xw(k + 2)};
            const float2v mo = {mw(k), mw(k + 1)};
            const float2v d1 = m0 - mo;
            const float s1a = sum1 + d1.x, s1b = s1a + d1.y;
            sum1 = s1b;
            const float2v s1 = {s1a, s1b};
            const float2v m1 = s1 * inv;
            const float2v y2 = sig - m1;
            ys[k] = y2.x; ys[k + 1] = y2.y;
        }
        if constexpr (SB & 1) {
            constexpr int k = SB - 1;
            const float d0 = in(k) - xw(k);
            sum0 += d0;
            const float m0 = sum0 * P.dc_inv_len;
            mnew[k] = m0;
            const float d1 = m0 - mw(k);
            sum1 += d1;
            const float m1 = sum1 * P.dc_inv_len;
            ys[k] = xw(k + 1) - m1;
        }
        L.sum0 = sum0; L.sum1 = sum1;
        // the registers this phase has just released take the next inputs
        if constexpr (PAR == 0) load_b(blk);
        else if (blk + 1u < n_blocks) load_a(blk + 1u);
        RX_LAP(0); RX_COUNT(0);

        // ---- AGC rx/agc.rs:72-77 (relaxed: gain * (1 - bw |y|) + bw) and window push receiver.rs:345-346 ----
        const float g0 = L.gain;
        float *wblk = wcol + wbase * LP;
        const bool mirror = wbase == 0u;                     // wave-uniform: this sub-block's first MIR slots are stored twice
        const float bw0 = (L.flags & F_AGC_LOCKED) ? 0.0f : P.agc_bw;
        auto agc_pass = [&](float g, int fk, float bwa, float bwb) -> float {
            float out[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const float bw = (k <= fk) ? bwa : bwb;
                const float a = __builtin_fmaf(-bw, fabsf(ys[k]), 1.0f);
                out[k] = ys[k] * g;
                g = __builtin_amdgcn_fmed3f(__builtin_fmaf(g, a, bw), P.agc_min, P.agc_max);
            }
            if (mirror) {
#pragma unroll
                for (int k = 0; k < MIR; ++k) { wblk[k * LP] = out[k]; wblk[(k + RING) * LP] = out[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < MIR; ++k) wblk[k * LP] = out[k];
            }
#pragma unroll
            for (int k = MIR; k < SB; ++k) wblk[k * LP] = out[k];
            return g;
        };
        L.gain = agc_pass(g0, SB, bw0, bw0);
        RX_LAP(1);

        // ---- TED instants of this sub-block: matched filters, timing loop, symbol path ----
        while (until < SB) {
            const int fk = until;
            const float sa_low = demod_relaxed<NT, RING, OCC == 1>(taps_lds, wcol_lds, (int)wbase + fk);
            RX_LAP(2); RX_COUNT(1);
            const float rem = L.until_next_ted - (float)cstar;              // receiver.rs:352
            const uint32_t locked_before = L.flags & F_AGC_LOCKED;
            float zero, sym, terr;
            if (ted_timing_relaxed(P, L, inv_spt, sa_low, rem, &zero, &sym, &terr)) {
                const uint64_t counter = counter0 + (int64_t)row_l + (uint64_t)blk * B + (uint32_t)(PAR * SB + fk) + 1u;
                uint32_t burst_len = 0;
                bool emit = false;
                const uint32_t link = symbol_link(P, L, S, X, c, zero, sym, terr, L.until_next_ted, counter, &burst_len, &emit, true, hpre0, hpre1);
                if constexpr (TICKS) symbol_io(P, L, S, O, X, c, link, emit, counter, burst_len);
                else if (emit) X.emit(P, S, O, c, link, counter, L.sq_symbols, burst_len);
            }
            cstar = next_fire_count(L.until_next_ted, 0u);
            until = fk + cstar;
            RX_LAP(3);
            if ((L.flags & F_AGC_LOCKED) != locked_before) {
                // the lock changed at sample fk: redo the sub-block's AGC with the old lock up to fk and the new one
                // after it (twice per burst)
                const float bw1 = (L.flags & F_AGC_LOCKED) ? 0.0f : P.agc_bw;
                L.gain = agc_pass(g0, fk, bw0, bw1);
            }
        }
        until -= SB;
        wbase = wbase == (uint32_t)(RING - SB) ? 0u : wbase + (uint32_t)SB;
        RX_LAP(4);
    };

    load_a(0u);
    uint32_t blk = 0;
    for (; blk < n_blocks; ++blk) {
        sub(std::integral_constant<int, 0>{}, blk);
        sub(std::integral_constant<int, 1>{}, blk);
        if (may_leave) {
            // Time-parallel chunk that hands over: from its nominal end on, a lane's hand-over instant is the end of
            // the first block after which its link state is NoCarrier; once every lane has one the wavefront leaves
            // (its state is not needed: the chunk that owns the instant carries on)
            if (!lane_done && blk + 1u >= n_nominal && (L.flags & F_LINK_MASK) == 0u && blk < avail_l) {
                lane_done = true;
                K.handover[c] = counter0 + (int64_t)row_l + (uint64_t)(blk + 1u) * B;
            }
            if (__builtin_amdgcn_ballot_w64(!lane_done) == 0ull) { RX_REPORT(); return; }
        }
        RX_LAP(5);
    }
    RX_REPORT();

    // ---- write the state back ----
    L.ted_clock = (uint32_t)(cstar - until - 1);
    if constexpr (TICKS) lane_store(L, S, c); else lane_store_link(L, S, c);
#pragma unroll
    for (int i = 0; i < NFF; ++i) { S.eq_ffc[i * C + c] = X.ffc[i]; S.eq_ffw[i * C + c] = X.ffw[i]; }
#pragma unroll
    for (int i = 0; i < NFB; ++i) { S.eq_fbc[i * C + c] = X.fbc[i]; S.eq_fbw[i * C + c] = X.fbw[i]; }
    const uint64_t counter1 = counter0 + (uint64_t)n_blocks * B;
    {
        // the last RING samples are in the ring (wbase = slot of the next sample); the state keeps win_ring of them,
        // of which the filters only ever read the newest NT - 1
        const uint32_t G = P.win_ring;
#pragma unroll 2
        for (uint32_t m = 1; m <= G; ++m) {
            const uint32_t g = (uint32_t)(counter1 - (uint64_t)m) & (G - 1u);
            float *row = S.win_ring + (size_t)g * C;
            const uint32_t j = wbase >= m ? wbase - m : wbase + (uint32_t)RING - m;
            row[c] = m <= (uint32_t)RING ? wcol[j * LP] : 0.0f;
        }
    }
    {
        const uint32_t dpos = (uint32_t)(counter1 % (uint64_t)DCL);
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            uint32_t slot = dpos + (uint32_t)k;
            if (slot >= (uint32_t)DCL) slot -= (uint32_t)DCL;
            float *r0 = S.dc_ff_ring + (size_t)slot * C, *r1 = S.dc_fb_ring + (size_t)slot * C;
            r0[c] = xb[k + HOFF - 1];
            r1[c] = mb[k + HOFF];
        }
    }
}

// ---------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------
// The relaxed kernel exists for 22.05 kHz with the reference's default DC-blocker length, the default or the disabled
// equalizer, a non-negative AGC floor (|x * gain| = |x| * gain) and a timing loop that cannot put three instants
// into one sub-block.
bool relaxed_kernel_supported(const Params &P)
{
    if (!(P.ntaps == 42u && P.dc_len == 16u && P.win_ring >= 64u)) return false;
    if (!((P.eq_nff == 6u && P.eq_nfb == 4u) || (P.eq_nff == 1u && P.eq_nfb == 1u))) return false;
    if (!(P.agc_min >= 0.0f)) return false;
    return max_block_len(P) >= (uint32_t)kBlockMirror;
}
uint32_t relaxed_block_len(const Params &P) { (void)P; return (uint32_t)RelaxLayout<42>::B; }

template <int NFF, int NFB, typename SampleT, bool CM, bool TICKS, int OCC>
static void launch_relaxed_one(const Params &P, const State &S, const Output &O, const float4 *taps, const SampleT *x,
                               uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{
    const uint32_t grid = (P.n_channels + kWave - 1) / kWave;
    hipLaunchKernelGGL((demod_relaxed_kernel<42, NFF, NFB, SampleT, CM, TICKS, OCC>), dim3(grid), dim3(kWave), RelaxLayout<42>::lds_bytes,
                       stream, P, S, O, taps, x, n_blocks, counter0, K);
}

template <typename SampleT>
static hipError_t launch_relaxed_t(const Params &P, const State &S, const Output &O, const float4 *taps,
                                   const SampleT *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                                   const PipeChunks &K)
{
    if (K.n_chunks > 1u && (K.in_channels % kWave) != 0u) return hipErrorInvalidValue;   // a wavefront would straddle chunks
    const bool cm = K.n_chunks > 1u && K.col_row0 != nullptr;
    if (cm && !std::is_same<SampleT, float>::value) return hipErrorInvalidValue;
    if (P.n_channels > 0x7fffffffu / 64u / sizeof(SampleT)) return hipErrorInvalidValue;   // a block of rows within a buffer resource
    const uint32_t grid = (P.n_channels + kWave - 1) / kWave;
    const bool eq64 = P.eq_nff == 6u && P.eq_nfb == 4u, ticks = P.ticks != 0u;
    // more wavefronts than SIMDs: the build for two per SIMD; otherwise a wavefront has its SIMD's registers to itself
    const bool wide = grid <= 1024u && !ticks && std::is_same<SampleT, float>::value;
#define SAME_RELAX_GO(NFF, NFB, CM_, TK, OC) launch_relaxed_one<NFF, NFB, SampleT, CM_, TK, OC>(P, S, O, taps, x, n_blocks, counter0, stream, K)
    if constexpr (std::is_same<SampleT, float>::value) {
        if (cm) {
            if (eq64) { if (wide) SAME_RELAX_GO(6, 4, true, false, 1); else SAME_RELAX_GO(6, 4, true, false, 2); }
            else { if (wide) SAME_RELAX_GO(1, 1, true, false, 1); else SAME_RELAX_GO(1, 1, true, false, 2); }
            return hipGetLastError();
        }
        if (wide) {
            if (eq64) SAME_RELAX_GO(6, 4, false, false, 1); else SAME_RELAX_GO(1, 1, false, false, 1);
            return hipGetLastError();
        }
    }
    if (eq64) { if (ticks) SAME_RELAX_GO(6, 4, false, true, 2); else SAME_RELAX_GO(6, 4, false, false, 2); }
    else { if (ticks) SAME_RELAX_GO(1, 1, false, true, 2); else SAME_RELAX_GO(1, 1, false, false, 2); }
#undef SAME_RELAX_GO
    return hipGetLastError();
}

hipError_t launch_demod_relaxed(const Params &P, const State &S, const Output &O, const float4 *taps,
                                const float *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_relaxed_t<float>(P, S, O, taps, x, n_blocks, counter0, stream, K); }
hipError_t launch_demod_relaxed_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                                    const int16_t *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_relaxed_t<int16_t>(P, S, O, taps, x, n_blocks, counter0, stream, K); }

}  // namespace same

RELAXED_PROFILE_EXPORTS()
