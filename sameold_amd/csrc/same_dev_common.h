// same_dev_common.h -- device functions shared by the demodulation kernels: the Rust f32
// semantics, the per-lane register state, event emission and the whole symbol-rate path
// (timing loop, squelch, equalizer, framer, transport wake-ups).
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#pragma once

#include <hip/hip_runtime.h>

#include "same_device.h"

namespace same {

// ---------------------------------------------------------------------------------
// Rust f32 semantics
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float rs_clamp(float x, float mn, float mx)
{
    // f32::clamp: NaN and the sign of zero pass through
    x = (x < mn) ? mn : x;
    x = (x > mx) ? mx : x;
    return x;
}
__device__ __forceinline__ float rs_signum(float x)
{
    // f32::signum: +1 for +0.0, -1 for -0.0 (NaN inputs are outside the contract)
    return __uint_as_float((__float_as_uint(x) & 0x80000000u) | 0x3f800000u);
}
__device__ __forceinline__ float rs_hypot(float re, float im)
{
    // Complex::norm() = re.hypot(im) -> glibc hypotf == (float)sqrt((double)x*x + (double)y*y)
    const double a = (double)re, b = (double)im;
    const double aa = a * a, bb = b * b;
    const double x = aa + bb;
    // sqrt(x) in f64: the compiler's own expansion (v_rsq_f64 seed, Goldschmidt step, two
    // residual corrections) written out without its 2^256 pre-scaling of inputs below 2^-767:
    // x is 0 or at least (2^-149)^2, so that branch can never be taken and the result is
    // bit-identical to sqrt(x) as the compiler lowers it (tests/test_gpu_parity.py::test_hypot)
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y, h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    const double g3 = __builtin_fma(d1, h1, g2);
    const double r = (x == 0.0 || x == __builtin_inf()) ? x : g3;
    return (float)r;
}

// ---------------------------------------------------------------------------------
// per-lane state held in registers for the duration of a launch
// ---------------------------------------------------------------------------------
struct Lane {
    float sum0, sum1, gain;
    float until_next_ted;
    uint32_t ted_clock;
    float h0, h1, h2, period_avg, period_inst;
    uint32_t sq_data;
    float sq_power;
    uint32_t sq_phist, sq_fill;
    int32_t sq_clock;
    uint64_t sq_symbols;
    uint32_t eq_word, eq_count;
    uint32_t fr_word, fr_count, fr_invalid, fr_len;
    uint32_t flags;
    uint64_t tk_next, tk_last, wake_sample, wake_fired;
    uint32_t ended;        // set by rx_end(); scratch for the pipelined kernel, never stored
};

__device__ __forceinline__ void lane_load(Lane &L, const State &S, uint32_t c)
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
    L.tk_next = S.tk_next[c]; L.tk_last = S.tk_last[c]; L.wake_sample = S.wake_sample[c]; L.wake_fired = S.wake_fired[c];
}
__device__ __forceinline__ void lane_store(const Lane &L, const State &S, uint32_t c)
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
    S.tk_next[c] = L.tk_next; S.tk_last[c] = L.tk_last; S.wake_fired[c] = L.wake_fired;   // wake_sample is the host's
}

// ---------------------------------------------------------------------------------
// events
// ---------------------------------------------------------------------------------
// Framer row -> burst pool, 288 bytes.  Written as load-all-then-store-all per group of three
// 16-byte words (more in flight spills registers in the kernels that are at their limit): a plain copy loop compiles to load, wait, store, load, wait, ... (the two
// pointers may alias as far as the compiler knows) -- 18 dependent HBM round trips with one
// lane live, ~10 000 cycles per burst on the stage that emits it.
__device__ __forceinline__ void copy_burst_row(uint8_t *dst_row, const uint8_t *src_row)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(src_row);
    uint4 *dst = reinterpret_cast<uint4 *>(dst_row);
    static_assert(kBurstCap % 48 == 0, "burst rows are copied in groups of three 16-byte words");
#pragma unroll
    for (int g = 0; g < kBurstCap / 48; ++g) {
        uint4 t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = src[g * 3 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) dst[g * 3 + i] = t[i];
    }
}

// Out of line, and fed scalars only: a by-reference struct argument would force the
// kernel's Params/State/Lane copies out of registers into scratch around every call.
static __device__ __noinline__ void emit_event_raw(DevEvent *events, uint32_t *counters, uint32_t event_cap,
                                                   uint8_t *bursts, uint32_t burst_cap, const uint8_t *fr_row,
                                                   uint32_t c, uint32_t kind, uint64_t sample_counter,
                                                   uint64_t symbols, uint32_t burst_len)
{
    // counters: [0] events cursor, [1] bursts cursor, [2] overflow flags
    uint32_t slot = 0xffffffffu;
    if (kind == 3u) {  // SAME_LINK_BURST: copy the framer buffer row into the pool
        uint32_t b = atomicAdd(counters + 1, 1u);
        if (b < burst_cap) {
            slot = b;
            copy_burst_row(bursts + (size_t)b * kBurstCap, fr_row);
        } else {
            atomicOr(counters + 2, 2u);
        }
    }
    uint32_t e = atomicAdd(counters, 1u);
    if (e < event_cap) {
        DevEvent ev;
        ev.channel = c; ev.kind = kind; ev.sample_counter = sample_counter;
        ev.symbol_count = symbols; ev.burst_len = burst_len; ev.burst_slot = slot;
        events[e] = ev;
    } else {
        atomicOr(counters + 2, 1u);
    }
}
__device__ __forceinline__ void emit_event(const Params &P, const State &S, const Output &O,
                                           uint32_t c, uint32_t kind, uint64_t sample_counter,
                                           uint64_t symbols, uint32_t burst_len)
{
    emit_event_raw(O.events, O.n_events, O.event_cap, O.bursts, O.burst_cap,
                   S.fr_msg + (size_t)c * kBurstCap, c, kind, sample_counter, symbols, burst_len);
}

// ---------------------------------------------------------------------------------
// framer (rx/framing.rs) -- integer work, one byte at a time
// ---------------------------------------------------------------------------------
__device__ __forceinline__ bool is_allowed_byte(uint32_t c)
{
    // rx/combiner.rs:105-137: '-', '0'-'9', 'A'-'Z', 'a'-'z', and "/?()[]._,+ " -- as a 128-bit
    // membership bitmap (one select chain and a shift instead of fifteen range tests)
    const uint32_t w = c < 32u ? 0x00000000u : (c < 64u ? 0x83fffb01u : (c < 96u ? 0xaffffffeu : (c < 128u ? 0x07fffffeu : 0u)));
    return ((w >> (c & 31u)) & 1u) != 0u;
}
__device__ __forceinline__ uint32_t fr_state(const Lane &L)
{ return (L.flags & F_FR_STATE_MASK) >> F_FR_STATE_SHIFT; }
__device__ __forceinline__ void fr_set_state(Lane &L, uint32_t s)
{ L.flags = (L.flags & ~F_FR_STATE_MASK) | (s << F_FR_STATE_SHIFT); }
// Framer::state() rx/framing.rs:191-197 -- framer state numbers equal LinkState kinds 0..2
__device__ __forceinline__ uint32_t framer_state(const Lane &L) { return fr_state(L); }

// Framer::end() rx/framing.rs:174-186.  Returns the LinkState kind; for a Burst, *burst_len
// is its length and the bytes are still in S.fr_msg (emit_event copies them).
__device__ __forceinline__ uint32_t framer_end(Lane &L, uint32_t *burst_len)
{
    uint32_t st = fr_state(L);
    fr_set_state(L, 0);
    if (st == 2u) { *burst_len = L.fr_len; return 3u; }
    return 0u;
}
__device__ __forceinline__ void framer_push(Lane &L, const State &S, uint32_t c, uint32_t byte)
{
    if (L.fr_len < (uint32_t)kBurstCap) S.fr_msg[(size_t)c * kBurstCap + L.fr_len] = (uint8_t)byte;
    L.fr_len += 1;
}
// the non-restart arm of Framer::input rx/framing.rs:124-164, with the two live states
// evaluated side by side and committed by selects (only the byte stores are predicated)
__device__ __forceinline__ uint32_t framer_feed(const Params &P, Lane &L, const State &S,
                                                uint32_t c, uint32_t data, uint32_t *burst_len)
{
    const uint32_t st = fr_state(L);
    const bool searching = st == 1u, reading = st == 2u;
    // PrefixSearch :128-150
    const uint32_t word = (L.fr_word << 8) | data;
    const uint32_t count = L.fr_count + 1u;
    const uint32_t e0 = __popc(word ^ 0x5a435a43u);          // "ZCZC" rx/framing.rs:235-243
    const uint32_t e1 = __popc(word ^ 0x4e4e4e4eu);          // "NNNN"
    const bool found = searching & (min(e0, e1) <= P.fr_max_prefix_errors);
    const bool give_up = searching & !found & (count > 21u); // PREFIX_SEARCH_LEN :201
    // DataRead :153-163
    const uint32_t invalid = L.fr_invalid + (is_allowed_byte(data) ? 0u : 1u);
    const bool over = reading & (invalid > P.fr_max_invalid);
    const bool keep = reading & !over;

    L.fr_word = searching ? word : L.fr_word;
    L.fr_count = searching ? count : L.fr_count;
    L.fr_invalid = found ? 0u : (reading ? invalid : L.fr_invalid);
    if (found) {
        uint8_t *row = S.fr_msg + (size_t)c * kBurstCap;     // the prefix as received seeds the burst :136-138
        row[0] = (uint8_t)(word >> 24); row[1] = (uint8_t)(word >> 16);
        row[2] = (uint8_t)(word >> 8); row[3] = (uint8_t)word;
    }
    if (keep && L.fr_len < (uint32_t)kBurstCap) S.fr_msg[(size_t)c * kBurstCap + L.fr_len] = (uint8_t)data;
    if (over) *burst_len = L.fr_len;                         // Framer::end(): the burst without this byte
    L.fr_len = found ? 4u : (keep ? L.fr_len + 1u : L.fr_len);
    const uint32_t nst = found ? 2u : ((give_up | over) ? 0u : st);
    fr_set_state(L, nst);
    return over ? 3u : nst;                                  // LinkState kind: framer state numbers coincide
}

// ---------------------------------------------------------------------------------
// equalizer (rx/equalize.rs): NLMS decision-feedback equalizer, 16 samples -> 1 byte
// ---------------------------------------------------------------------------------
// Window arrays are stored oldest-first: w[0] oldest ... w[N-1] newest.
// One Equalizer::estimate_symbol (rx/equalize.rs:249-308) on coefficient/window arrays held
// in registers; returns the decided bit.  Window arrays are oldest-first.
template <int NFF, int NFB>
__device__ __forceinline__ uint32_t eq_symbol_core(const Params &P, Lane &L, float (&ffc)[NFF],
                                                   float (&ffw)[NFF], float (&fbc)[NFB],
                                                   float (&fbw)[NFB], float in0, float in1)
{
    uint32_t mode = (L.flags & F_EQ_MODE_MASK) >> F_EQ_MODE_SHIFT;
    // feedforward_wind.push(input) rx/equalize.rs:253 (Window::push rx/filter.rs:258-275)
    if (NFF >= 2) {
#pragma unroll
        for (int i = 0; i + 2 < NFF; ++i) ffw[i] = ffw[i + 2];
        ffw[NFF >= 2 ? NFF - 2 : 0] = in0;
        ffw[NFF - 1] = in1;
    } else {
        ffw[0] = in1;
    }
    // multiply_accumulate: newest sample times coeff[0] first (rx/filter.rs:363-377)
    float ff = 0.0f;
#pragma unroll
    for (int i = 0; i < NFF; ++i) { float p = ffw[NFF - 1 - i] * ffc[i]; ff += p; }
    // Feedback window: push(&[decision, 0.0]) (rx/equalize.rs:304) puts an exact +0.0 at every other slot --
    // indices NFB-1, NFB-3, ... -- from the reset on (the window starts as zeros).  A tap that meets one of them
    // contributes fbc * 0 = +-0 to a sum that is +0.0 or non-zero (x + (+-0) == x, +0 + (-0) == +0), is updated
    // by ge * 0 = +-0 (same argument: coefficients start at 1.0 / +0.0), and adds 0 * 0 to the sum of squares:
    // all three are skipped, bit for bit the same result.
    constexpr auto fb_zero = [](int widx) { return ((NFB - 1 - widx) & 1) == 0; };
    float fb = 0.0f;
#pragma unroll
    for (int i = 0; i < NFB; ++i) { if (fb_zero(NFB - 1 - i)) continue; float p = fbw[NFB - 1 - i] * fbc[i]; fb += p; }
    float sym_val = ff - fb;
    float sym_est, err;
    bool evolve = true;
    if (mode == 2u) {                                  // EnabledTraining :278-301
        float bit = (float)(L.eq_word & 1u);
        float tb = 2.0f * bit;
        sym_est = tb - 1.0f;
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
        // nlms_update rx/equalize.rs:354-386: gain = relax / (reg + sum w^2, oldest first)
        float sumsq = 0.0f;
#pragma unroll
        for (int i = 0; i < NFF; ++i) { float q = ffw[i] * ffw[i]; sumsq += q; }
        float gain = P.eq_relaxation / (P.eq_regularization + sumsq);
        float ge = gain * err;
#pragma unroll
        for (int i = 0; i < NFF; ++i) { float p = ge * ffw[NFF - 1 - i]; ffc[i] += p; }
        float nerr = -err;
        sumsq = 0.0f;
#pragma unroll
        for (int i = 0; i < NFB; ++i) { if (fb_zero(i)) continue; float q = fbw[i] * fbw[i]; sumsq += q; }
        gain = P.eq_relaxation / (P.eq_regularization + sumsq);
        ge = gain * nerr;
#pragma unroll
        for (int i = 0; i < NFB; ++i) { if (fb_zero(NFB - 1 - i)) continue; float p = ge * fbw[NFB - 1 - i]; fbc[i] += p; }
    }
    // feedback_wind.push(&[out.0, 0.0]) :304
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

// the same with the state fetched from / returned to the HBM state arrays
template <int NFF, int NFB>
__device__ __forceinline__ uint32_t eq_symbols_regs(const Params &P, Lane &L, const State &S, uint32_t c,
                                                    const float *samples, int nsym)
{
    const uint32_t C = P.n_channels;
    float ffc[NFF], ffw[NFF], fbc[NFB], fbw[NFB];
#pragma unroll
    for (int i = 0; i < NFF; ++i) { ffc[i] = S.eq_ffc[i * C + c]; ffw[i] = S.eq_ffw[i * C + c]; }
#pragma unroll
    for (int i = 0; i < NFB; ++i) { fbc[i] = S.eq_fbc[i * C + c]; fbw[i] = S.eq_fbw[i * C + c]; }
    uint32_t bits = 0;
#pragma unroll 1
    for (int b = 0; b < nsym; ++b)
        bits |= eq_symbol_core<NFF, NFB>(P, L, ffc, ffw, fbc, fbw, samples[2 * b], samples[2 * b + 1]) << b;
#pragma unroll
    for (int i = 0; i < NFF; ++i) { S.eq_ffc[i * C + c] = ffc[i]; S.eq_ffw[i * C + c] = ffw[i]; }
#pragma unroll
    for (int i = 0; i < NFB; ++i) { S.eq_fbc[i * C + c] = fbc[i]; S.eq_fbw[i * C + c] = fbw[i]; }
    return bits;
}

// any filter order up to kMaxEqTaps, working directly on the state arrays
__device__ __forceinline__ uint32_t eq_symbols_generic(const Params &P, Lane &L, const State &S,
                                                       uint32_t c, const float *samples, int nsym)
{
    const uint32_t C = P.n_channels;
    const int NFF = (int)P.eq_nff, NFB = (int)P.eq_nfb;
    float *ffc = S.eq_ffc + c, *ffw = S.eq_ffw + c, *fbc = S.eq_fbc + c, *fbw = S.eq_fbw + c;
    uint32_t mode = (L.flags & F_EQ_MODE_MASK) >> F_EQ_MODE_SHIFT;
    uint32_t byte = 0;
    for (int b = 0; b < nsym; ++b) {
        float in0 = samples[2 * b], in1 = samples[2 * b + 1];
        if (NFF >= 2) {
            for (int i = 0; i + 2 < NFF; ++i) ffw[i * C] = ffw[(i + 2) * C];
            ffw[(NFF - 2) * C] = in0; ffw[(NFF - 1) * C] = in1;
        } else {
            ffw[0] = in1;
        }
        float ff = 0.0f;
        for (int i = 0; i < NFF; ++i) { float p = ffw[(NFF - 1 - i) * C] * ffc[i * C]; ff += p; }
        float fb = 0.0f;
        for (int i = 0; i < NFB; ++i) { float p = fbw[(NFB - 1 - i) * C] * fbc[i * C]; fb += p; }
        float sym_val = ff - fb;
        float sym_est, err;
        bool evolve = true;
        if (mode == 2u) {
            float bit = (float)(L.eq_word & 1u);
            float tb = 2.0f * bit;
            sym_est = tb - 1.0f;
            L.eq_word >>= 1;
            err = sym_est - sym_val;
            L.eq_count += 1;
            if (L.eq_count >= 32u) mode = 1u;
        } else if (mode == 1u) {
            sym_est = rs_signum(sym_val);
            err = sym_est - sym_val;
        } else {
            sym_est = rs_signum(sym_val); err = 0.0f; evolve = false;
        }
        if (evolve) {
            float sumsq = 0.0f;
            for (int i = 0; i < NFF; ++i) { float v = ffw[i * C]; float q = v * v; sumsq += q; }
            float gain = P.eq_relaxation / (P.eq_regularization + sumsq);
            float ge = gain * err;
            for (int i = 0; i < NFF; ++i) { float p = ge * ffw[(NFF - 1 - i) * C]; ffc[i * C] += p; }
            float nerr = -err;
            sumsq = 0.0f;
            for (int i = 0; i < NFB; ++i) { float v = fbw[i * C]; float q = v * v; sumsq += q; }
            gain = P.eq_relaxation / (P.eq_regularization + sumsq);
            ge = gain * nerr;
            for (int i = 0; i < NFB; ++i) { float p = ge * fbw[(NFB - 1 - i) * C]; fbc[i * C] += p; }
        }
        if (NFB >= 2) {
            for (int i = 0; i + 2 < NFB; ++i) fbw[i * C] = fbw[(i + 2) * C];
            fbw[(NFB - 2) * C] = sym_est; fbw[(NFB - 1) * C] = 0.0f;
        } else {
            fbw[0] = 0.0f;
        }
        byte |= (sym_est >= 0.0f ? 1u : 0u) << b;
    }
    L.flags = (L.flags & ~F_EQ_MODE_MASK) | (mode << F_EQ_MODE_SHIFT);
    return byte;
}

// Equalizer::reset rx/equalize.rs:191-196 (mode is preserved)
__device__ __forceinline__ void eq_reset(const Params &P, const State &S, uint32_t c)
{
    const uint32_t C = P.n_channels;
    for (uint32_t i = 0; i < P.eq_nff; ++i) { S.eq_ffc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_ffw[i * C + c] = 0.0f; }
    for (uint32_t i = 0; i < P.eq_nfb; ++i) { S.eq_fbc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_fbw[i * C + c] = 0.0f; }
}

// The deadline ring (oldest first) is reached through the kernel's context: tk_count / tk_at /
// tk_set.  TickRingGlobal keeps it in the HBM state arrays; the pipelined kernel keeps it in
// registers for the launch.
struct TickRingGlobal {
    __device__ __forceinline__ uint32_t tk_count(const State &S, uint32_t c) const { return S.tk_n[c]; }
    __device__ __forceinline__ void tk_set_count(const State &S, uint32_t c, uint32_t n) const { S.tk_n[c] = n; }
    __device__ __forceinline__ uint64_t tk_at(const State &S, uint32_t C, uint32_t c, uint32_t i) const { return S.tk_ring[i * C + c]; }
    __device__ __forceinline__ void tk_set(const State &S, uint32_t C, uint32_t c, uint32_t i, uint64_t v) const { S.tk_ring[i * C + c] = v; }
};

// Where the symbol-rate state that is too big for registers lives.  GlobalCtx keeps the
// squelch sample history and the equalizer in the HBM state arrays (any configuration);
// the fast kernel supplies a context with the history in LDS and the equalizer in VGPRs.
struct GlobalCtx : TickRingGlobal {
    const State &S;
    uint32_t c, C;
    __device__ __forceinline__ GlobalCtx(const State &S_, uint32_t c_, uint32_t C_) : S(S_), c(c_), C(C_) {}
    __device__ __forceinline__ void mark(int) const {}      // profiling hook of the fast kernel
    __device__ __forceinline__ void emit(const Params &P, const State &S, const Output &O, uint32_t c, uint32_t kind,
                                         uint64_t sample_counter, uint64_t symbols, uint32_t burst_len) const
    { emit_event(P, S, O, c, kind, sample_counter, symbols, burst_len); }
    __device__ __forceinline__ void hist_put(uint32_t slot, float v) const { S.sq_hist[slot * C + c] = v; }
    __device__ __forceinline__ float hist_get(uint32_t slot) const { return S.sq_hist[slot * C + c]; }
    // run the equalizer over nsym symbols (2 samples each); bit b of the result = symbol b
    __device__ __forceinline__ uint32_t eq_symbols(const Params &P, Lane &L, const float *samples, int nsym) const
    {
        if (P.eq_nff == 6u && P.eq_nfb == 4u) return eq_symbols_regs<6, 4>(P, L, S, c, samples, nsym);
        if (P.eq_nff == 1u && P.eq_nfb == 1u) return eq_symbols_regs<1, 1>(P, L, S, c, samples, nsym);
        return eq_symbols_generic(P, L, S, c, samples, nsym);
    }
    __device__ __forceinline__ uint32_t eq_symbol1(const Params &P, Lane &L, float in0, float in1) const
    { float two[2] = {in0, in1}; return eq_symbols(P, L, two, 1); }
    __device__ __forceinline__ void eq_reset(const Params &P) const { same::eq_reset(P, S, c); }
    __device__ __forceinline__ void eq_snapshot(const Params &P) const
    {
        for (uint32_t i = 0; i < P.eq_nff; ++i) { S.eq_snap_ffc[i * C + c] = S.eq_ffc[i * C + c]; S.eq_snap_ffw[i * C + c] = S.eq_ffw[i * C + c]; }
        for (uint32_t i = 0; i < P.eq_nfb; ++i) { S.eq_snap_fbc[i * C + c] = S.eq_fbc[i * C + c]; S.eq_snap_fbw[i * C + c] = S.eq_fbw[i * C + c]; }
    }
    __device__ __forceinline__ void eq_restore(const Params &P) const
    {
        for (uint32_t i = 0; i < P.eq_nff; ++i) { S.eq_ffc[i * C + c] = S.eq_snap_ffc[i * C + c]; S.eq_ffw[i * C + c] = S.eq_snap_ffw[i * C + c]; }
        for (uint32_t i = 0; i < P.eq_nfb; ++i) { S.eq_fbc[i * C + c] = S.eq_snap_fbc[i * C + c]; S.eq_fbw[i * C + c] = S.eq_snap_fbw[i * C + c]; }
    }
};

// SameReceiver::end receiver.rs:479-490
template <typename Ctx>
__device__ __forceinline__ void rx_end(const Params &P, Lane &L, Ctx &X)
{
    L.flags &= ~(F_AGC_LOCKED | F_SQ_LOCK | F_BW_LOCKED | F_TED_PHASE);
    L.ended = 1u;
    L.sq_clock = -1;                                       // squelch.end() rx/codesquelch.rs:336-339
    X.eq_reset(P);
    // symsync.set_loop_bandwidth(unlocked); symsync.reset() rx/symsync.rs:166-170, 265-271
    L.h0 = 0.0f; L.h1 = 0.0f; L.h2 = 0.0f;
    L.period_avg = P.samples_per_ted; L.period_inst = P.samples_per_ted;
}

// ---------------------------------------------------------------------------------
// symbol-rate processing: SameReceiver::process_linklayer_symbol receiver.rs:407-474
// returns the LinkState kind
// ---------------------------------------------------------------------------------
template <typename Ctx>
__device__ __forceinline__ uint32_t rx_symbol(const Params &P, Lane &L, const State &S, Ctx &X,
                                              uint32_t c, float zero, float sym, uint32_t *burst_len,
                                              bool have_pre = false, float pre0 = 0.0f, float pre1 = 0.0f)
{
    // Written as straight-line selects wherever the work is a few integer operations: a
    // 64-lane wavefront has lanes in every state at once, so every branch region executes
    // anyway and each exec-mask region costs scalar work, hazards and register copies.
    // Branches are kept only around the expensive or rare pieces (equalizer step, byte
    // handling, end-of-burst reset, event emission).

    // --- CodeAndPowerSquelch::input rx/codesquelch.rs:228-304
    const uint32_t slot = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;
    // the symbol the equalizer may step over below (history offset 14/15 from the oldest sample
    // once this symbol is in): slots slot+16/+17, distinct from the two written here, so the
    // reads are issued first and their latency hides under the squelch arithmetic
    // (have_pre: the caller has read them already, together with the symbol itself -- one LDS round trip instead of two)
    const float eq_in0 = have_pre ? pre0 : X.hist_get((slot + 16u) & 63u), eq_in1 = have_pre ? pre1 : X.hist_get((slot + 17u) & 63u);
    X.hist_put(slot, zero);
    X.hist_put(slot + 1u, sym);
    const uint32_t fill = min(64u, L.sq_fill + 2u);
    L.sq_fill = fill;
    const uint32_t bit = (sym >= 0.0f) ? 1u : 0u;           // CodeCorrelator::search :421-428
    L.sq_data = (L.sq_data >> 1) | (bit << 31);
    const uint32_t nerr = __popc(P.sync_word ^ L.sq_data);
    const float pw = sym * sym;                             // PowerTracker::track :483-488
    const float dp = pw - L.sq_power;
    const float up = dp * P.sq_bw;
    L.sq_power += up;
    L.sq_power = fmaxf(L.sq_power, 0.0f);
    const float pwr = L.sq_power;
    L.sq_phist = (L.sq_phist << 1) | ((pwr >= P.sq_power_close) ? 1u : 0u);
    L.sq_symbols += 1;

    const int32_t clock_before = L.sq_clock;                // byte clock before this symbol (-1: no sync)
    const bool full = fill >= 64u;                          // sample_history.is_full() :237
    const bool locked = (L.flags & F_SQ_LOCK) != 0;
    // :244-265 sync acquired / re-affirmed / adjusted
    const bool sync_now = full & !locked & (nerr <= P.sq_max_errors) & (pwr >= P.sq_power_open);
    const bool adjusted = sync_now & (clock_before != 0);
    // :266-273 lost sync: power_history.front() is the flag pushed 31 symbols ago
    const bool drop = full & !sync_now & (clock_before >= 0) & ((L.sq_phist & 0x80000000u) == 0u);
    int32_t clk = sync_now ? 0 : clock_before;
    clk = drop ? -1 : clk;
    // :277-303 byte clock
    const bool ready = full & (clk == 0);
    const bool reading = full & (clk > 0);
    clk = ready ? 1 : (reading ? ((clk + 1) & 7) : clk);
    L.sq_clock = clk;
    if (drop) L.flags &= ~F_SQ_LOCK;                        // squelch.end() inside the squelch :336-339

    // Equalizer schedule.  The reference runs the equalizer over the 8 symbols of a byte when
    // the squelch's byte clock wraps (rx/codesquelch.rs:283-299 -> rx/equalize.rs:173-186): the
    // symbols are the OLDEST 16 history samples, i.e. they were all known 24 symbols earlier.
    // While a channel stays in byte sync the byte boundaries are known in advance, so symbol j
    // of the next byte is equalized during the j-th symbol before it is due; at that moment it
    // sits at history offset 14/15 from the oldest sample.  The equalizer sees the same symbols
    // in the same order, so its state and output are bit-identical; the work arrives as one
    // short step per symbol instead of an 8-symbol burst per byte, which is what keeps a
    // 64-lane wavefront from serialising on whichever lane has a byte due.
    //  * carrier drop / end of burst mid-byte: end() resets the equalizer (receiver.rs:482),
    //    exactly as the reference discards an equalizer that never saw those symbols;
    //  * the byte clock is re-aligned while the squelch is still unlocked ("adjust byte sync",
    //    rx/codesquelch.rs:255-264): the symbols equalized ahead belonged to a byte the
    //    reference never forms, so the state saved at the last completed byte is restored
    //    before training restarts.
    const uint32_t head = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;   // oldest sample
    X.mark(3);
    if ((clock_before >= 0) & !adjusted & (ready | reading)) {
        const uint32_t j = (uint32_t)(clock_before + 7) & 7u;   // clock 1..7 -> symbol 0..6, clock 0 -> 7
        const uint32_t ebit = X.eq_symbol1(P, L, eq_in0, eq_in1);   // history slots head+14, head+15
        uint32_t bits = (j == 0u) ? 0u : ((L.flags & F_EQ_BITS_MASK) >> F_EQ_BITS_SHIFT);
        bits |= ebit << j;
        L.flags = (L.flags & ~F_EQ_BITS_MASK) | (bits << F_EQ_BITS_SHIFT);
    }

    X.mark(4);
    uint32_t link;
    if (ready) {
        // --- Ready: a byte is due.  receiver.rs:423-446
        uint32_t byte;
        if (adjusted) {
            if (clock_before >= 0) X.eq_restore(P);             // drop the symbols equalized ahead
            L.flags |= F_AGC_LOCKED | F_BW_LOCKED;              // agc.lock(true); locked loop bandwidth
            L.flags = (L.flags & ~F_EQ_MODE_MASK) | (2u << F_EQ_MODE_SHIFT);   // equalizer.train()
            L.eq_word = P.sync_word; L.eq_count = 0;
            // the oldest 16 samples of the history (rx/codesquelch.rs:288-294)
            float samples[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) samples[i] = X.hist_get((head + i) & 63u);
            byte = X.eq_symbols(P, L, samples, 8);
        } else {
            byte = (L.flags & F_EQ_BITS_MASK) >> F_EQ_BITS_SHIFT;   // symbol 7 was equalized just above
        }
        // --- framer receiver.rs:457-471, Framer::input rx/framing.rs:109-123
        if (adjusted) {
            uint32_t blen = 0;
            const uint32_t out = framer_end(L, &blen);
            // fr_msg is only rewritten when a new prefix is found, which cannot happen on the
            // first byte of a search, so the row is still intact when the caller emits it
            if (out == 3u) *burst_len = blen;
            fr_set_state(L, 1); L.fr_word = 0; L.fr_count = 0;
            uint32_t dummy = 0;
            (void)framer_feed(P, L, S, c, byte, &dummy);
            link = (out == 3u) ? 3u : 1u;
        } else {
            link = framer_feed(P, L, S, c, byte, burst_len);
        }
        if (link == 2u) L.flags |= F_SQ_LOCK;                   // squelch.lock(true)
        else if (link == 0u || link == 3u) rx_end(P, L, X);
        // a re-alignment is still possible: remember the equalizer as of this completed byte
        if (L.sq_clock >= 0 && !(L.flags & F_SQ_LOCK)) X.eq_snapshot(P);
    } else {
        // Reading: framer.state(); NoCarrier / DroppedCarrier: framer.end()  receiver.rs:410-422
        const uint32_t fst = fr_state(L);
        const bool was_reading_burst = !reading & (fst == 2u);
        link = reading ? fst : (was_reading_burst ? 3u : 0u);
        if (was_reading_burst) *burst_len = L.fr_len;
        if (!reading) fr_set_state(L, 0);
        if (drop) rx_end(P, L, X);
    }
    return link;
}

// ---------------------------------------------------------------------------------
// Transport wake-ups.  The reference polls its Assembler on every symbol whose link
// state is NoCarrier or Burst (receiver.rs:292-315), but the answer can only change
//   (1) at a Burst,
//   (2) at the first poll on/after a pending message's deadline, burst + 682 symbols
//       (rx/assembler.rs:294-299), or after the history empties, last burst + 5652,
//   (3) at the first poll after the forced-EOM sample instant (receiver.rs:300-309),
//   (4) at the poll following any of the above (a Message state decays to Assembling/Idle).
// The device reports those poll instants as SAME_DEV_TICK events (kind 8) so the host can
// replay the Assembler with identical sample counters.
// ---------------------------------------------------------------------------------
static constexpr uint64_t kNoDeadline = ~0ull;

template <typename Ctx>
__device__ __forceinline__ uint64_t tick_min(const Lane &L, const State &S, const Ctx &X, uint32_t C, uint32_t c, uint32_t n)
{
    uint64_t m = L.tk_last;
    if (n) { uint64_t f = X.tk_at(S, C, c, 0u); m = f < m ? f : m; }   // ring is oldest (smallest) first
    return m;
}
template <typename Ctx>
__device__ __forceinline__ void tick_on_burst(const Params &P, Lane &L, const State &S, Ctx &X, uint32_t c)
{
    const uint32_t C = P.n_channels;
    uint32_t n = X.tk_count(S, c);
    if (n == (uint32_t)kTickRing) {          // full: drop the oldest deadline
        for (uint32_t i = 1; i < n; ++i) X.tk_set(S, C, c, i - 1u, X.tk_at(S, C, c, i));
        n -= 1;
    }
    X.tk_set(S, C, c, n, L.sq_symbols + P.tick_interburst);
    n += 1;
    X.tk_set_count(S, c, n);
    L.tk_last = L.sq_symbols + P.tick_history;
    L.tk_next = tick_min(L, S, X, C, c, n);
    L.flags |= F_TICK_AGAIN;
}
template <typename Ctx>
__device__ __forceinline__ void tick_poll(const Params &P, Lane &L, const State &S, const Output &O, Ctx &X,
                                       uint32_t c, uint64_t counter)
{
    const uint32_t C = P.n_channels;
    const uint64_t sym = L.sq_symbols;
    const bool expired = sym >= L.tk_next;
    const bool woke = L.wake_sample != 0 && L.wake_sample != L.wake_fired && counter > L.wake_sample;
    X.emit(P, S, O, c, 8u, counter, sym, 0);
    L.flags &= ~F_TICK_AGAIN;
    if (woke) L.wake_fired = L.wake_sample;
    if (expired) {
        uint32_t n = X.tk_count(S, c), drop = 0;
        while (drop < n && X.tk_at(S, C, c, drop) <= sym) ++drop;
        if (drop) {
            for (uint32_t i = drop; i < n; ++i) X.tk_set(S, C, c, i - drop, X.tk_at(S, C, c, i));
            n -= drop;
            X.tk_set_count(S, c, n);
        }
        if (L.tk_last <= sym) L.tk_last = kNoDeadline;
        L.tk_next = tick_min(L, S, X, C, c, n);
    }
    if (expired || woke) L.flags |= F_TICK_AGAIN;
}

// ---------------------------------------------------------------------------------
// One TED instant: matched filters -> timing loop -> (every other instant) a symbol.
// process_linklayer_low_rate receiver.rs:376-395.  `newest` is the window-ring slot of
// the sample on which the TED fired; `rem` is clock_remaining_sa; `counter` is
// input_sample_counter at that sample.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float demod_now(const Params &P, const float4 *__restrict__ taps,
                                           const float *win, uint32_t newest, uint32_t lane)
{
    // FskDemod::demod_now rx/demod.rs:156-164; multiply_accumulate rx/filter.rs:363-377:
    // acc += window[newest - i] * h[i], i = 0 first, each product and each sum rounded.
    const uint32_t mask = P.win_ring - 1u;
    float mre = 0.0f, mim = 0.0f, sre = 0.0f, sim = 0.0f;
    for (uint32_t i = 0; i < P.ntaps; ++i) {
        float x = win[((newest - i) & mask) * kWave + lane];
        float4 h = taps[i];                                 // wave-uniform: scalar load
        float p0 = x * h.x, p1 = x * h.y, p2 = x * h.z, p3 = x * h.w;
        mre += p0; mim += p1; sre += p2; sim += p3;
    }
    float d = rs_hypot(mre, mim) - rs_hypot(sre, sim);
    return rs_clamp(d, -1.0f, 1.0f);
}

// The timing half of a TED instant: ZeroCrossingTed + TimingLoop.  Returns true when this
// instant completes a symbol (receiver.rs:383), with its two samples and the timing error.
__device__ __forceinline__ bool ted_timing(const Params &P, Lane &L, float sa_low, float rem,
                                           float *zero_out, float *sym_out, float *terr_out)
{
    // ZeroCrossingTed::input rx/symsync.rs:278-287
    L.h0 = L.h1; L.h1 = L.h2; L.h2 = sa_low;
    L.flags ^= F_TED_PHASE;
    const bool have = (L.flags & F_TED_PHASE) != 0;
    const float dsg = rs_signum(L.h0) - rs_signum(L.h2);    // zero_crossing_metric :311-322
    const float terr = L.h1 * dsg;
    // TimingLoop::advance_loop rx/symsync.rs:219-244 -- both arms computed, one committed
    const float offset = rs_clamp(rem, -0.5f, 0.5f);
    const float q = offset / P.samples_per_ted;
    const float e0 = terr - q;
    const float e = rs_clamp(e0, -1.0f, 1.0f);
    const bool bw_locked = (L.flags & F_BW_LOCKED) != 0;
    const float alpha = bw_locked ? P.alpha_locked : P.alpha_unlocked;
    const float beta = bw_locked ? P.beta_locked : P.beta_unlocked;
    const float bi = beta * e;
    const float avg1 = rs_clamp(L.period_avg + bi, P.period_min, P.period_max);
    const float ai = alpha * e;
    const float t = avg1 + ai;
    float inst1 = t + offset;
    inst1 = (inst1 < 0.0f) ? avg1 : inst1;
    const float inst0 = L.period_inst + offset;              // no symbol this time: period_inst += offset
    L.period_avg = have ? avg1 : L.period_avg;
    L.period_inst = have ? inst1 : inst0;
    L.until_next_ted = L.period_inst;                       // receiver.rs:382
    *zero_out = L.h1; *sym_out = L.h2; *terr_out = terr;
    return have;
}

// ted_timing in two halves, for a stage that waits for the instant's soft sample (the matched filters run on
// other wavefronts).  Of the new sample the timing loop uses one bit: terr = h1 * (signum(h0) - signum(h2))
// (rx/symsync.rs:311-322) with h2 the new sample, and on the instants that complete no symbol not even that
// (period_inst += offset, rx/symsync.rs:236-241).  ted_ahead evaluates everything ted_timing would for both
// signs of the sample that is still on its way -- the same operations on the same operands in the same
// order, so whichever half is committed is bit-identical to ted_timing's result -- including the sample
// count of the next instant (next_fire_count); ted_commit, once the sample is there, selects by its sign bit.
struct TedAhead {
    float zero;              // h1 after the shift: the symbol estimate's first sample (and the TED's middle tap)
    // ..0: the new sample is >= +0.0, ..1: it carries a sign bit.  Scalars, selected -- never indexed: the struct lives
    // across loop iterations in the pipeline's stage 2, and a runtime index would send it to scratch memory
    float terr0, terr1;
    float avg0, avg1, inst0, inst1;   // period_avg / period_inst after the instant
    int cstar0, cstar1;               // next_fire_count(period_inst, 0)
    uint32_t flags;          // with the TED phase toggled
    bool have;
};
__device__ __forceinline__ int next_fire_count(float s, uint32_t clock);   // same_fast_common.h
__device__ __forceinline__ TedAhead ted_ahead(const Params &P, const Lane &L, float rem)
{
    TedAhead A;
    const float h0 = L.h1, h1 = L.h2;                       // ZeroCrossingTed::input rx/symsync.rs:278-287
    A.zero = h1;
    A.flags = L.flags ^ F_TED_PHASE;
    A.have = (A.flags & F_TED_PHASE) != 0;
    const float offset = rs_clamp(rem, -0.5f, 0.5f);        // TimingLoop::advance_loop rx/symsync.rs:219-244
    const float q = offset / P.samples_per_ted;
    const bool bw_locked = (L.flags & F_BW_LOCKED) != 0;
    const float alpha = bw_locked ? P.alpha_locked : P.alpha_unlocked;
    const float beta = bw_locked ? P.beta_locked : P.beta_unlocked;
    const float inst_plain = L.period_inst + offset;
    auto half = [&](float sg2, float *terr_out, float *avg_out, float *inst_out, int *cstar_out) {   // sg2 = rs_signum(h2)
        const float dsg = rs_signum(h0) - sg2;
        const float terr = h1 * dsg;
        const float e0 = terr - q;
        const float e = rs_clamp(e0, -1.0f, 1.0f);
        const float bi = beta * e;
        const float avg1 = rs_clamp(L.period_avg + bi, P.period_min, P.period_max);
        const float ai = alpha * e;
        const float t = avg1 + ai;
        float inst1 = t + offset;
        inst1 = (inst1 < 0.0f) ? avg1 : inst1;
        *terr_out = terr;
        *avg_out = A.have ? avg1 : L.period_avg;
        *inst_out = A.have ? inst1 : inst_plain;
        *cstar_out = next_fire_count(*inst_out, 0u);
    };
    half(1.0f, &A.terr0, &A.avg0, &A.inst0, &A.cstar0);
    half(-1.0f, &A.terr1, &A.avg1, &A.inst1, &A.cstar1);
    return A;
}
__device__ __forceinline__ bool ted_commit(Lane &L, const TedAhead &A, float sa_low, float *zero_out, float *sym_out,
                                           float *terr_out, int *cstar_out)
{
    const bool neg = (__float_as_uint(sa_low) >> 31) != 0u;     // the sign bit, as rs_signum reads it
    L.h0 = L.h1; L.h1 = L.h2; L.h2 = sa_low;
    L.flags = A.flags;
    L.period_avg = neg ? A.avg1 : A.avg0;
    L.period_inst = neg ? A.inst1 : A.inst0;
    L.until_next_ted = L.period_inst;                       // receiver.rs:382
    *zero_out = A.zero; *sym_out = sa_low; *terr_out = neg ? A.terr1 : A.terr0;
    *cstar_out = neg ? A.cstar1 : A.cstar0;
    return A.have;
}

// The symbol half, in two parts so that a pipelined kernel can run them on different wavefronts.
// symbol_link: trace and link layer (squelch, equalizer, framer); returns the LinkState kind and
// whether an event is due (receiver.rs:246-253: report on change; a Burst always differs from its
// predecessor).  `until_next_ted` is only recorded in the trace.
template <typename Ctx>
__device__ __forceinline__ uint32_t symbol_link(const Params &P, Lane &L, const State &S, Ctx &X, uint32_t c,
                                                float zero, float sym, float terr, float until_next_ted,
                                                uint64_t counter, uint32_t *burst_len, bool *emit,
                                                bool have_pre = false, float pre0 = 0.0f, float pre1 = 0.0f)
{
    if (P.trace_cap) {
        uint32_t n = S.trace_n[c];
        if (n < P.trace_cap) {
            float *t = S.trace + ((size_t)c * P.trace_cap + n) * 4;
            t[0] = zero; t[1] = sym; t[2] = terr; t[3] = until_next_ted;
            S.trace_idx[(size_t)c * P.trace_cap + n] = counter;
        }
        S.trace_n[c] = n + 1;
    }
    X.mark(2);
    uint32_t link = rx_symbol(P, L, S, X, c, zero, sym, burst_len, have_pre, pre0, pre1);
    X.mark(5);
    const uint32_t last = (L.flags & F_LINK_MASK) >> F_LINK_SHIFT;
    *emit = link != last || link == 3u;
    if (*emit) L.flags = (L.flags & ~F_LINK_MASK) | (link << F_LINK_SHIFT);
    return link;
}

// symbol_io: the link event and the transport wake-ups of one symbol.  Uses of L: sq_symbols (the
// count after this symbol), tk_next, tk_last, wake_sample, wake_fired and the F_TICK_AGAIN flag.
template <typename Ctx>
__device__ __forceinline__ void symbol_io(const Params &P, Lane &L, const State &S, const Output &O, Ctx &X,
                                          uint32_t c, uint32_t link, bool emit, uint64_t counter, uint32_t burst_len)
{
    if (emit) X.emit(P, S, O, c, link, counter, L.sq_symbols, burst_len);
    if (P.ticks) {
        if (link == 3u) {
            tick_on_burst(P, L, S, X, c);
        } else if (link == 0u) {
            if ((L.flags & F_TICK_AGAIN) || L.sq_symbols >= L.tk_next ||
                (L.wake_sample != 0 && L.wake_sample != L.wake_fired && counter > L.wake_sample))
                tick_poll(P, L, S, O, X, c, counter);
        }
    }
    X.mark(6);
}

template <typename Ctx>
__device__ __forceinline__ void ted_symbol(const Params &P, Lane &L, const State &S, const Output &O,
                                           Ctx &X, uint32_t c, float zero, float sym, float terr,
                                           float until_next_ted, uint64_t counter,
                                           bool have_pre = false, float pre0 = 0.0f, float pre1 = 0.0f)
{
    uint32_t burst_len = 0;
    bool emit = false;
    const uint32_t link = symbol_link(P, L, S, X, c, zero, sym, terr, until_next_ted, counter, &burst_len, &emit, have_pre, pre0, pre1);
    symbol_io(P, L, S, O, X, c, link, emit, counter, burst_len);
}

template <typename Ctx>
__device__ __forceinline__ void ted_instant(const Params &P, Lane &L, const State &S,
                                            const Output &O, Ctx &X, uint32_t c, float sa_low,
                                            float rem, uint64_t counter,
                                            bool have_pre = false, float pre0 = 0.0f, float pre1 = 0.0f)
{
    float zero, sym, terr;
    if (!ted_timing(P, L, sa_low, rem, &zero, &sym, &terr)) return;
    ted_symbol(P, L, S, O, X, c, zero, sym, terr, L.until_next_ted, counter, have_pre, pre0, pre1);
}


}  // namespace same
