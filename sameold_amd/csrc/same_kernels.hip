// same_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched SAME demodulator.
//
// One wavefront lane owns one channel (one `SameReceiver`); a workgroup is one 64-lane
// wavefront.  All channels advance in lockstep over a time-major input x[t][channel],
// so each global load of a wavefront is 256 contiguous bytes.
//
// Arithmetic contract: every f32 operation is evaluated in the order, and with the
// single IEEE rounding, of the reference's scalar Rust code (no FMA contraction: this
// file is built with -ffp-contract=off; IEEE division; hypot as
// (float)sqrt((double)re*re + (double)im*im) == glibc hypotf).  The kernel is therefore
// bit-identical to the scalar CPU restatement used as the test oracle: identical TED
// instants, soft symbols, bytes and
// event sample counters.  Only the *schedule* differs (see "deferred TED" below).
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#include <hip/hip_runtime.h>

#include "same_device.h"
#include "same_launch.h"

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
    double a = (double)re, b = (double)im;
    double aa = a * a, bb = b * b;
    return (float)sqrt(aa + bb);
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
    uint64_t tk_next, tk_last, wake_sample;
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
    L.tk_next = S.tk_next[c]; L.tk_last = S.tk_last[c]; L.wake_sample = S.wake_sample[c];
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
    S.tk_next[c] = L.tk_next; S.tk_last[c] = L.tk_last; S.wake_sample[c] = L.wake_sample;
}

// ---------------------------------------------------------------------------------
// events
// ---------------------------------------------------------------------------------
__device__ __noinline__ void emit_event(const Params &P, const State &S, const Output &O,
                                        uint32_t c, uint32_t kind, uint64_t sample_counter,
                                        uint64_t symbols, uint32_t burst_len)
{
    uint32_t slot = 0xffffffffu;
    if (kind == 3u) {  // SAME_LINK_BURST: copy the framer buffer row into the pool
        uint32_t b = atomicAdd(O.n_bursts, 1u);
        if (b < O.burst_cap) {
            slot = b;
            const uint4 *src = reinterpret_cast<const uint4 *>(S.fr_msg + (size_t)c * kBurstCap);
            uint4 *dst = reinterpret_cast<uint4 *>(O.bursts + (size_t)b * kBurstCap);
#pragma unroll
            for (int i = 0; i < kBurstCap / 16; ++i) dst[i] = src[i];
        } else {
            atomicOr(O.overflow, 2u);
        }
    }
    uint32_t e = atomicAdd(O.n_events, 1u);
    if (e < O.event_cap) {
        DevEvent ev;
        ev.channel = c; ev.kind = kind; ev.sample_counter = sample_counter;
        ev.symbol_count = symbols; ev.burst_len = burst_len; ev.burst_slot = slot;
        O.events[e] = ev;
    } else {
        atomicOr(O.overflow, 1u);
    }
}

// ---------------------------------------------------------------------------------
// framer (rx/framing.rs) -- integer work, one byte at a time
// ---------------------------------------------------------------------------------
__device__ __forceinline__ bool is_allowed_byte(uint32_t c)
{
    // rx/combiner.rs:105-137
    return c == '-' || (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') ||
           (c >= 'a' && c <= 'z') || c == '/' || c == '?' || c == '(' || c == ')' ||
           c == '[' || c == ']' || c == '.' || c == '_' || c == ',' || c == '+' || c == ' ';
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
// the non-restart arm of Framer::input rx/framing.rs:124-164
__device__ __forceinline__ uint32_t framer_feed(const Params &P, Lane &L, const State &S,
                                                uint32_t c, uint32_t data, uint32_t *burst_len)
{
    uint32_t st = fr_state(L);
    if (st == 0u) return 0u;
    if (st == 1u) {
        L.fr_word = (L.fr_word << 8) | data;
        L.fr_count += 1;
        uint32_t e0 = __popc(L.fr_word ^ 0x5a435a43u);   // "ZCZC" rx/framing.rs:235-243
        uint32_t e1 = __popc(L.fr_word ^ 0x4e4e4e4eu);   // "NNNN"
        if (min(e0, e1) <= P.fr_max_prefix_errors) {
            L.fr_len = 0;
            framer_push(L, S, c, (L.fr_word >> 24) & 0xff);
            framer_push(L, S, c, (L.fr_word >> 16) & 0xff);
            framer_push(L, S, c, (L.fr_word >> 8) & 0xff);
            framer_push(L, S, c, L.fr_word & 0xff);
            L.fr_invalid = 0;
            fr_set_state(L, 2);
        } else if (L.fr_count > 21u) {                   // PREFIX_SEARCH_LEN :201
            fr_set_state(L, 0);
        }
        return framer_state(L);
    }
    L.fr_invalid += is_allowed_byte(data) ? 0u : 1u;
    if (L.fr_invalid > P.fr_max_invalid) return framer_end(L, burst_len);
    framer_push(L, S, c, data);
    return framer_state(L);
}

// ---------------------------------------------------------------------------------
// equalizer (rx/equalize.rs): NLMS decision-feedback equalizer, 16 samples -> 1 byte
// ---------------------------------------------------------------------------------
// Window arrays are stored oldest-first: w[0] oldest ... w[N-1] newest.
template <int NFF, int NFB>
__device__ __forceinline__ uint32_t eq_byte_regs(const Params &P, Lane &L, const State &S,
                                                 uint32_t c, const float *samples)
{
    const uint32_t C = P.n_channels;
    float ffc[NFF], ffw[NFF], fbc[NFB], fbw[NFB];
#pragma unroll
    for (int i = 0; i < NFF; ++i) { ffc[i] = S.eq_ffc[i * C + c]; ffw[i] = S.eq_ffw[i * C + c]; }
#pragma unroll
    for (int i = 0; i < NFB; ++i) { fbc[i] = S.eq_fbc[i * C + c]; fbw[i] = S.eq_fbw[i * C + c]; }
    uint32_t mode = (L.flags & F_EQ_MODE_MASK) >> F_EQ_MODE_SHIFT;
    uint32_t byte = 0;
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {
        float in0 = samples[2 * b], in1 = samples[2 * b + 1];
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
        float fb = 0.0f;
#pragma unroll
        for (int i = 0; i < NFB; ++i) { float p = fbw[NFB - 1 - i] * fbc[i]; fb += p; }
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
            for (int i = 0; i < NFB; ++i) { float q = fbw[i] * fbw[i]; sumsq += q; }
            gain = P.eq_relaxation / (P.eq_regularization + sumsq);
            ge = gain * nerr;
#pragma unroll
            for (int i = 0; i < NFB; ++i) { float p = ge * fbw[NFB - 1 - i]; fbc[i] += p; }
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
        byte |= (sym_est >= 0.0f ? 1u : 0u) << b;
    }
    L.flags = (L.flags & ~F_EQ_MODE_MASK) | (mode << F_EQ_MODE_SHIFT);
#pragma unroll
    for (int i = 0; i < NFF; ++i) { S.eq_ffc[i * C + c] = ffc[i]; S.eq_ffw[i * C + c] = ffw[i]; }
#pragma unroll
    for (int i = 0; i < NFB; ++i) { S.eq_fbc[i * C + c] = fbc[i]; S.eq_fbw[i * C + c] = fbw[i]; }
    return byte;
}

// any filter order up to kMaxEqTaps, working directly on the state arrays
__device__ __noinline__ uint32_t eq_byte_generic(const Params &P, Lane &L, const State &S,
                                                 uint32_t c, const float *samples)
{
    const uint32_t C = P.n_channels;
    const int NFF = (int)P.eq_nff, NFB = (int)P.eq_nfb;
    float *ffc = S.eq_ffc + c, *ffw = S.eq_ffw + c, *fbc = S.eq_fbc + c, *fbw = S.eq_fbw + c;
    uint32_t mode = (L.flags & F_EQ_MODE_MASK) >> F_EQ_MODE_SHIFT;
    uint32_t byte = 0;
    for (int b = 0; b < 8; ++b) {
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
__device__ __noinline__ void eq_reset(const Params &P, const State &S, uint32_t c)
{
    const uint32_t C = P.n_channels;
    for (uint32_t i = 0; i < P.eq_nff; ++i) { S.eq_ffc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_ffw[i * C + c] = 0.0f; }
    for (uint32_t i = 0; i < P.eq_nfb; ++i) { S.eq_fbc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_fbw[i * C + c] = 0.0f; }
}

// SameReceiver::end receiver.rs:479-490
__device__ __forceinline__ void rx_end(const Params &P, Lane &L, const State &S, uint32_t c)
{
    L.flags &= ~(F_AGC_LOCKED | F_SQ_LOCK | F_BW_LOCKED | F_TED_PHASE);
    L.sq_clock = -1;                                       // squelch.end() rx/codesquelch.rs:336-339
    eq_reset(P, S, c);
    // symsync.set_loop_bandwidth(unlocked); symsync.reset() rx/symsync.rs:166-170, 265-271
    L.h0 = 0.0f; L.h1 = 0.0f; L.h2 = 0.0f;
    L.period_avg = P.samples_per_ted; L.period_inst = P.samples_per_ted;
}

// ---------------------------------------------------------------------------------
// symbol-rate processing: SameReceiver::process_linklayer_symbol receiver.rs:407-474
// returns the LinkState kind
// ---------------------------------------------------------------------------------
__device__ __noinline__ uint32_t rx_symbol(const Params &P, Lane &L, const State &S, uint32_t c,
                                           float zero, float sym, uint32_t *burst_len)
{
    const uint32_t C = P.n_channels;
    // --- CodeAndPowerSquelch::input rx/codesquelch.rs:228-304
    uint32_t slot = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;
    S.sq_hist[slot * C + c] = zero;
    S.sq_hist[(slot + 1u) * C + c] = sym;
    L.sq_fill = min(64u, L.sq_fill + 2u);
    uint32_t bit = (sym >= 0.0f) ? 1u : 0u;                 // CodeCorrelator::search :421-428
    L.sq_data = (L.sq_data >> 1) | (bit << 31);
    uint32_t nerr = __popc(P.sync_word ^ L.sq_data);
    float pw = sym * sym;                                   // PowerTracker::track :483-488
    float dp = pw - L.sq_power;
    float up = dp * P.sq_bw;
    L.sq_power += up;
    L.sq_power = fmaxf(L.sq_power, 0.0f);
    float pwr = L.sq_power;
    L.sq_phist = (L.sq_phist << 1) | ((pwr >= P.sq_power_close) ? 1u : 0u);
    L.sq_symbols += 1;

    enum { NO_CARRIER, DROPPED, READING, READY };
    int st;
    bool adjusted = false;
    if (L.sq_fill < 64u) {
        st = NO_CARRIER;
    } else {
        st = -1;
        if (!(L.flags & F_SQ_LOCK) && nerr <= P.sq_max_errors && pwr >= P.sq_power_open) {
            adjusted = (L.sq_clock != 0);
            L.sq_clock = 0;
        } else if (L.sq_clock >= 0 && !(L.sq_phist & 0x80000000u)) {
            // power_history.front(): the flag pushed 31 symbols ago
            L.flags &= ~F_SQ_LOCK; L.sq_clock = -1;
            st = DROPPED;
        }
        if (st < 0) {
            if (L.sq_clock < 0) st = NO_CARRIER;
            else if (L.sq_clock == 0) { L.sq_clock = 1; st = READY; }
            else { L.sq_clock = (L.sq_clock + 1) % 8; st = READING; }
        }
    }

    // --- receiver.rs:409-443
    if (st == NO_CARRIER) return framer_end(L, burst_len);
    if (st == DROPPED) { rx_end(P, L, S, c); return framer_end(L, burst_len); }
    if (st == READING) return framer_state(L);

    if (adjusted) {
        L.flags |= F_AGC_LOCKED | F_BW_LOCKED;              // agc.lock(true); locked loop bandwidth
        L.flags = (L.flags & ~F_EQ_MODE_MASK) | (2u << F_EQ_MODE_SHIFT);   // equalizer.train()
        L.eq_word = P.sync_word; L.eq_count = 0;
    }
    // the oldest 16 samples of the history (rx/codesquelch.rs:288-294)
    float samples[16];
    uint32_t head = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;
#pragma unroll
    for (int i = 0; i < 16; ++i) samples[i] = S.sq_hist[((head + i) & 63u) * C + c];

    // --- equalizer receiver.rs:446
    uint32_t byte;
    if (P.eq_nff == 6u && P.eq_nfb == 4u) byte = eq_byte_regs<6, 4>(P, L, S, c, samples);
    else if (P.eq_nff == 1u && P.eq_nfb == 1u) byte = eq_byte_regs<1, 1>(P, L, S, c, samples);
    else byte = eq_byte_generic(P, L, S, c, samples);

    // --- framer receiver.rs:457-471, Framer::input rx/framing.rs:109-123
    uint32_t link;
    if (adjusted) {
        uint32_t blen = 0;
        uint32_t out = framer_end(L, &blen);
        if (out == 3u) {
            // the burst must be copied out before the restarted search overwrites nothing:
            // fr_msg is only rewritten when a new prefix is found, which cannot happen on
            // the first byte of a search, so the row is still intact when the caller emits.
            *burst_len = blen;
        }
        fr_set_state(L, 1); L.fr_word = 0; L.fr_count = 0;
        uint32_t dummy = 0;
        (void)framer_feed(P, L, S, c, byte, &dummy);
        link = (out == 3u) ? 3u : 1u;
    } else {
        link = framer_feed(P, L, S, c, byte, burst_len);
    }
    if (link == 2u) L.flags |= F_SQ_LOCK;                   // squelch.lock(true)
    else if (link == 0u || link == 3u) rx_end(P, L, S, c);
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
constexpr uint64_t kNoDeadline = ~0ull;

__device__ __forceinline__ uint64_t tick_min(const Lane &L, const State &S, uint32_t C, uint32_t c, uint32_t n)
{
    uint64_t m = L.tk_last;
    if (n) { uint64_t f = S.tk_ring[c]; m = f < m ? f : m; }   // ring is oldest (smallest) first
    return m;
}
__device__ __noinline__ void tick_on_burst(const Params &P, Lane &L, const State &S, uint32_t c)
{
    const uint32_t C = P.n_channels;
    uint32_t n = S.tk_n[c];
    if (n == (uint32_t)kTickRing) {          // full: drop the oldest deadline
        for (uint32_t i = 1; i < n; ++i) S.tk_ring[(i - 1) * C + c] = S.tk_ring[i * C + c];
        n -= 1;
    }
    S.tk_ring[n * C + c] = L.sq_symbols + P.tick_interburst;
    n += 1;
    S.tk_n[c] = n;
    L.tk_last = L.sq_symbols + P.tick_history;
    L.tk_next = tick_min(L, S, C, c, n);
    L.flags |= F_TICK_AGAIN;
}
__device__ __noinline__ void tick_poll(const Params &P, Lane &L, const State &S, const Output &O,
                                       uint32_t c, uint64_t counter)
{
    const uint32_t C = P.n_channels;
    const uint64_t sym = L.sq_symbols;
    const bool expired = sym >= L.tk_next;
    const bool woke = L.wake_sample != 0 && counter > L.wake_sample;
    emit_event(P, S, O, c, 8u, counter, sym, 0);
    L.flags &= ~F_TICK_AGAIN;
    if (woke) L.wake_sample = 0;
    if (expired) {
        uint32_t n = S.tk_n[c], drop = 0;
        while (drop < n && S.tk_ring[drop * C + c] <= sym) ++drop;
        if (drop) {
            for (uint32_t i = drop; i < n; ++i) S.tk_ring[(i - drop) * C + c] = S.tk_ring[i * C + c];
            n -= drop;
            S.tk_n[c] = n;
        }
        if (L.tk_last <= sym) L.tk_last = kNoDeadline;
        L.tk_next = tick_min(L, S, C, c, n);
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

__device__ __forceinline__ void ted_instant(const Params &P, Lane &L, const State &S,
                                            const Output &O, uint32_t c, float sa_low, float rem,
                                            uint64_t counter)
{
    // ZeroCrossingTed::input rx/symsync.rs:278-287
    L.h0 = L.h1; L.h1 = L.h2; L.h2 = sa_low;
    L.flags ^= F_TED_PHASE;
    const bool have = (L.flags & F_TED_PHASE) != 0;
    float zero = L.h1, sym = L.h2, terr = 0.0f;
    if (have) {
        float d = rs_signum(L.h0) - rs_signum(L.h2);        // zero_crossing_metric :311-322
        terr = L.h1 * d;
    }
    // TimingLoop::advance_loop rx/symsync.rs:219-244
    float offset = rs_clamp(rem, -0.5f, 0.5f);
    if (have) {
        float q = offset / P.samples_per_ted;
        float e0 = terr - q;
        float e = rs_clamp(e0, -1.0f, 1.0f);
        const bool locked = (L.flags & F_BW_LOCKED) != 0;
        float alpha = locked ? P.alpha_locked : P.alpha_unlocked;
        float beta = locked ? P.beta_locked : P.beta_unlocked;
        float bi = beta * e;
        L.period_avg += bi;
        L.period_avg = rs_clamp(L.period_avg, P.period_min, P.period_max);
        float ai = alpha * e;
        float t = L.period_avg + ai;
        L.period_inst = t + offset;
        if (L.period_inst < 0.0f) L.period_inst = L.period_avg;
    } else {
        L.period_inst += offset;
    }
    L.until_next_ted = L.period_inst;                       // receiver.rs:382
    if (!have) return;                                      // receiver.rs:383

    if (P.trace_cap) {
        uint32_t n = S.trace_n[c];
        if (n < P.trace_cap) {
            float *t = S.trace + ((size_t)c * P.trace_cap + n) * 4;
            t[0] = zero; t[1] = sym; t[2] = terr; t[3] = L.until_next_ted;
            S.trace_idx[(size_t)c * P.trace_cap + n] = counter;
        }
        S.trace_n[c] = n + 1;
    }

    uint32_t burst_len = 0;
    uint32_t link = rx_symbol(P, L, S, c, zero, sym, &burst_len);
    // receiver.rs:246-253: report on change (a Burst always differs from its predecessor)
    uint32_t last = (L.flags & F_LINK_MASK) >> F_LINK_SHIFT;
    if (link != last || link == 3u) {
        L.flags = (L.flags & ~F_LINK_MASK) | (link << F_LINK_SHIFT);
        emit_event(P, S, O, c, link, counter, L.sq_symbols, burst_len);
    }
    if (P.ticks) {
        if (link == 3u) {
            tick_on_burst(P, L, S, c);
        } else if (link == 0u) {
            if ((L.flags & F_TICK_AGAIN) || L.sq_symbols >= L.tk_next ||
                (L.wake_sample != 0 && counter > L.wake_sample))
                tick_poll(P, L, S, O, c, counter);
        }
    }
}

// ---------------------------------------------------------------------------------
// The demodulation kernel.
//
// Deferred TED: time advances in blocks of B samples for all lanes together.  B is
// chosen on the host so that a lane's TED can fire at most once per block (the timing
// loop cannot command a period shorter than B + 0.5 samples).  Within a block every
// lane runs the uniform per-sample work (DC blocker, AGC, window push, sample clock)
// and notes the offset at which its TED fired; after the block, all lanes that fired
// evaluate their matched filters TOGETHER over the window as it stood at their own fire
// offset.  Without this, each of the ~21 different phases in a wavefront would pay for
// its own divergent 42-tap filter pass.
//
// Nothing after a TED instant can influence the per-sample stages of the same block
// except the AGC lock (agc.lock(true) on sync, lock(false) in end()); when the symbol
// processing of a lane changes its lock, the AGC outputs after the fire offset are
// recomputed from the saved gain ("replay"), which reproduces the sequential result
// exactly.
// ---------------------------------------------------------------------------------
template <int B, typename SampleT>
__global__ __launch_bounds__(kWave) void demod_kernel(Params P, State S, Output O,
                                                      const float4 *__restrict__ taps,
                                                      const SampleT *__restrict__ x,
                                                      uint32_t n_samples, uint64_t counter0)
{
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t C = P.n_channels;
    const uint32_t c_raw = blockIdx.x * kWave + lane;
    if (c_raw >= C) return;                                 // no barriers below: tail lanes just leave
    const uint32_t c = c_raw;
    const uint32_t DL = P.dc_len;
    float *ff = lds;                                        // [DL][64]
    float *fb = lds + DL * kWave;                           // [DL][64]
    float *win = lds + 2 * DL * kWave;                      // [win_ring][64]
    const uint32_t wmask = P.win_ring - 1u;

    // ring positions are common to all channels: every channel pushes once per sample
    uint32_t dpos = (uint32_t)(counter0 % DL);
    uint32_t wpos = (uint32_t)(counter0 & wmask);           // slot the next sample is written to

    for (uint32_t i = 0; i < DL; ++i) {
        ff[i * kWave + lane] = S.dc_ff_ring[i * C + c];
        fb[i * kWave + lane] = S.dc_fb_ring[i * C + c];
    }
    for (uint32_t i = 0; i < P.win_ring; ++i) win[i * kWave + lane] = S.win_ring[i * C + c];

    Lane L;
    lane_load(L, S, c);

    for (uint32_t t0 = 0; t0 < n_samples; t0 += B) {
        const uint32_t nb = min((uint32_t)B, n_samples - t0);
        float xs[B];
#pragma unroll
        for (int k = 0; k < B; ++k)
            xs[k] = (k < (int)nb) ? (float)x[(size_t)(t0 + k) * C + c] : 0.0f;

        float ys[B];
        int fire_k = -1;
        float fire_rem = 0.0f, fire_gain = 0.0f;
        const float unlocked0 = (L.flags & F_AGC_LOCKED) ? 0.0f : 1.0f;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (k < (int)nb) {
                // DCBlocker::filter rx/dcblock.rs:45-49, MovingAverage::filter :104-108
                const uint32_t dnext = (dpos + 1u == DL) ? 0u : dpos + 1u;
                float aged0 = ff[dpos * kWave + lane];
                ff[dpos * kWave + lane] = xs[k];
                float d0 = xs[k] - aged0;
                L.sum0 += d0;
                float ma0 = L.sum0 * P.dc_inv_len;
                float sig = ff[dnext * kWave + lane];       // window.front() after the push
                float aged1 = fb[dpos * kWave + lane];
                fb[dpos * kWave + lane] = ma0;
                float d1 = ma0 - aged1;
                L.sum1 += d1;
                float ma1 = L.sum1 * P.dc_inv_len;
                float km = P.dc_k * ma1;
                float y = sig - km;
                ys[k] = y;
                dpos = dnext;
                // Agc::input rx/agc.rs:72-77
                float out = y * L.gain;
                float e = 1.0f - fabsf(out);
                float ke = unlocked0 * e;
                float upd = ke * P.agc_bw;
                L.gain += upd;
                L.gain = rs_clamp(L.gain, P.agc_min, P.agc_max);
                // demod.push_scalar receiver.rs:346
                win[((wpos + k) & wmask) * kWave + lane] = out;
                // sample clock receiver.rs:347-355
                L.ted_clock += 1;
                float rem = L.until_next_ted - (float)L.ted_clock;
                bool fire = (fire_k < 0) && (rem <= 0.0f || fabsf(rem) < 0.5f);
                if (fire) { fire_k = k; fire_rem = rem; fire_gain = L.gain; L.ted_clock = 0; }
            }
        }

        if (fire_k >= 0) {
            const uint32_t newest = (wpos + (uint32_t)fire_k) & wmask;
            float sa_low = demod_now(P, taps, win, newest, lane);
            const uint32_t locked_before = L.flags & F_AGC_LOCKED;
            ted_instant(P, L, S, O, c, sa_low, fire_rem, counter0 + t0 + (uint32_t)fire_k + 1u);
            if ((L.flags & F_AGC_LOCKED) != locked_before) {
                // replay the AGC over the samples after the TED instant with the new lock
                const float unl = (L.flags & F_AGC_LOCKED) ? 0.0f : 1.0f;
                float g = fire_gain;
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    if (k > fire_k && k < (int)nb) {
                        float out = ys[k] * g;
                        float e = 1.0f - fabsf(out);
                        float ke = unl * e;
                        float upd = ke * P.agc_bw;
                        g += upd;
                        g = rs_clamp(g, P.agc_min, P.agc_max);
                        win[((wpos + k) & wmask) * kWave + lane] = out;
                    }
                }
                L.gain = g;
            }
        }
        wpos = (wpos + nb) & wmask;
    }

    lane_store(L, S, c);
    for (uint32_t i = 0; i < DL; ++i) {
        S.dc_ff_ring[i * C + c] = ff[i * kWave + lane];
        S.dc_fb_ring[i * C + c] = fb[i * kWave + lane];
    }
    for (uint32_t i = 0; i < P.win_ring; ++i) S.win_ring[i * C + c] = win[i * kWave + lane];
}

// ---------------------------------------------------------------------------------
// layout adaptor: channel-major x[c][t] -> time-major y[t][c], 64x64 tiles through LDS
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_to_time_major(const T *__restrict__ in,
                                                               T *__restrict__ out,
                                                               uint32_t n_channels, uint32_t n_samples)
{
    __shared__ T tile[64][65];
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;   // 64 x 4
    const uint32_t t0 = blockIdx.x * 64u, c0 = blockIdx.y * 64u;
    for (uint32_t r = ty; r < 64u; r += 4u) {
        uint32_t cc = c0 + r, tt = t0 + tx;
        if (cc < n_channels && tt < n_samples) tile[r][tx] = in[(size_t)cc * n_samples + tt];
    }
    __syncthreads();
    for (uint32_t r = ty; r < 64u; r += 4u) {
        uint32_t tt = t0 + r, cc = c0 + tx;
        if (cc < n_channels && tt < n_samples) out[(size_t)tt * n_channels + cc] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------------
// state initialisation: SameReceiver::from(&builder) receiver.rs:539-558 and reset() :182-198
// ---------------------------------------------------------------------------------
__global__ void init_state_kernel(Params P, State S, int is_reset)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t C = P.n_channels;
    if (c >= C) return;
    for (uint32_t i = 0; i < P.dc_len; ++i) { S.dc_ff_ring[i * C + c] = 0.0f; S.dc_fb_ring[i * C + c] = 0.0f; }
    S.dc_sum0[c] = 0.0f; S.dc_sum1[c] = 0.0f;
    // Agc::new starts at min(1, min_gain) (rx/agc.rs:55) but reset() sets 1.0 (:61)
    S.agc_gain[c] = is_reset ? 1.0f : P.agc_gain0;
    for (uint32_t i = 0; i < P.win_ring; ++i) S.win_ring[i * C + c] = 0.0f;
    S.ted_clock[c] = 0; S.until_next_ted[c] = P.samples_per_ted;
    S.ted_h0[c] = 0.0f; S.ted_h1[c] = 0.0f; S.ted_h2[c] = 0.0f;
    S.period_avg[c] = P.samples_per_ted; S.period_inst[c] = P.samples_per_ted;
    S.sq_data[c] = 0; S.sq_power[c] = 0.0f; S.sq_phist[c] = 0; S.sq_fill[c] = 0;
    S.sq_clock[c] = -1; S.sq_symbols[c] = 0;
    for (uint32_t i = 0; i < (uint32_t)kSquelchHist; ++i) S.sq_hist[i * C + c] = 0.0f;
    for (uint32_t i = 0; i < P.eq_nff; ++i) { S.eq_ffc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_ffw[i * C + c] = 0.0f; }
    for (uint32_t i = 0; i < P.eq_nfb; ++i) { S.eq_fbc[i * C + c] = (i == 0) ? 1.0f : 0.0f; S.eq_fbw[i * C + c] = 0.0f; }
    // Equalizer::reset() preserves its mode, training word and count (rx/equalize.rs:191-196)
    if (!is_reset) { S.eq_word[c] = 0; S.eq_count[c] = 0; }
    S.fr_word[c] = 0; S.fr_count[c] = 0; S.fr_invalid[c] = 0; S.fr_len[c] = 0;
    // the timing loop keeps the bandwidth it had (receiver.rs:186 only calls symsync.reset());
    // the framer goes idle and the reported link state returns to NoCarrier.
    uint32_t fl = is_reset ? (S.flags[c] & (F_EQ_MODE_MASK | F_BW_LOCKED)) : (1u << F_EQ_MODE_SHIFT);
    S.flags[c] = fl;
    S.tk_next[c] = kNoDeadline; S.tk_last[c] = kNoDeadline; S.tk_n[c] = 0; S.wake_sample[c] = 0;
    for (uint32_t i = 0; i < (uint32_t)kTickRing; ++i) S.tk_ring[i * C + c] = kNoDeadline;
    if (P.trace_cap) S.trace_n[c] = 0;
}

// ---------------------------------------------------------------------------------
// launchers (called from same_batch.cpp)
// ---------------------------------------------------------------------------------
template <typename SampleT>
static hipError_t launch_demod_t(const Params &P, const State &S, const Output &O, const float4 *taps,
                                 const SampleT *x, uint32_t n_samples, uint64_t counter0,
                                 hipStream_t stream)
{
    const uint32_t grid = (P.n_channels + kWave - 1) / kWave;
    const size_t lds = (size_t)(2 * P.dc_len + P.win_ring) * kWave * sizeof(float);
#define SAME_LAUNCH(BV)                                                                          \
    hipLaunchKernelGGL((demod_kernel<BV, SampleT>), dim3(grid), dim3(kWave), lds, stream, P, S, \
                       O, taps, x, n_samples, counter0)
    switch (P.block_len) {
    case 16: SAME_LAUNCH(16); break;
    case 8: SAME_LAUNCH(8); break;
    case 4: SAME_LAUNCH(4); break;
    case 2: SAME_LAUNCH(2); break;
    default: SAME_LAUNCH(1); break;
    }
#undef SAME_LAUNCH
    return hipGetLastError();
}

hipError_t launch_demod(const Params &P, const State &S, const Output &O, const float4 *taps,
                        const float *x, uint32_t n_samples, uint64_t counter0, hipStream_t stream)
{ return launch_demod_t<float>(P, S, O, taps, x, n_samples, counter0, stream); }

hipError_t launch_demod_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                            const int16_t *x, uint32_t n_samples, uint64_t counter0, hipStream_t stream)
{ return launch_demod_t<int16_t>(P, S, O, taps, x, n_samples, counter0, stream); }

size_t demod_lds_bytes(const Params &P)
{ return (size_t)(2 * P.dc_len + P.win_ring) * kWave * sizeof(float); }

hipError_t launch_init_state(const Params &P, const State &S, int is_reset, hipStream_t stream)
{
    const uint32_t grid = (P.n_channels + 255) / 256;
    hipLaunchKernelGGL(init_state_kernel, dim3(grid), dim3(256), 0, stream, P, S, is_reset);
    return hipGetLastError();
}

hipError_t launch_transpose_f32(const float *in, float *out, uint32_t n_channels, uint32_t n_samples,
                                hipStream_t stream)
{
    dim3 grid((n_samples + 63) / 64, (n_channels + 63) / 64);
    hipLaunchKernelGGL(transpose_to_time_major<float>, grid, dim3(256), 0, stream, in, out, n_channels, n_samples);
    return hipGetLastError();
}
hipError_t launch_transpose_i16(const int16_t *in, int16_t *out, uint32_t n_channels, uint32_t n_samples,
                                hipStream_t stream)
{
    dim3 grid((n_samples + 63) / 64, (n_channels + 63) / 64);
    hipLaunchKernelGGL(transpose_to_time_major<int16_t>, grid, dim3(256), 0, stream, in, out, n_channels, n_samples);
    return hipGetLastError();
}

}  // namespace same
