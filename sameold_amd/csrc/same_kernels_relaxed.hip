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
#include "same_relaxed_common.h"

namespace same {

// Geometry per sample rate (filter length NT): DC-blocker window, samples per sub-block (just below the nominal
// distance of two TED instants, so that nearly every lane has exactly one per sub-block), sub-blocks per block.
template <int NT> struct RelaxGeom;
template <> struct RelaxGeom<42> { static constexpr int DCL = 16, SB = 21; };     // 22.05 kHz: instants 21.17 apart
// (kRelaxChunk taps per filter chunk: the ring's first kRelaxChunk - 1 slots are mirrored)
template <int NT> struct RelaxLayout {
    static constexpr int SB = RelaxGeom<NT>::SB, B = 2 * SB, DCL = RelaxGeom<NT>::DCL;
    static constexpr int RING = 3 * SB, MIRROR = kRelaxChunk - 1;
    static constexpr uint32_t tap_floats = (uint32_t)((NT * 4 + 63) / 64 * 64);
    static constexpr size_t lds_bytes = ((size_t)tap_floats + (size_t)(RING + MIRROR) * kWave) * sizeof(float);
    static_assert(NT - 1 + SB <= RING, "a filter at the start of a sub-block reaches into the sub-block being written");
    static_assert(NT % kRelaxChunk == 0, "whole chunks");
    static_assert(SB >= DCL && SB - DCL >= 1, "the DC windows are the tail of the previous sub-block");
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
    auto load_rows = [&](float *dst, uint32_t b, auto k0_, auto n_) __attribute__((always_inline)) {
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
    auto sub = [&](auto par, uint32_t blk) __attribute__((always_inline)) {
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
        auto agc_pass = [&](float g, int fk, float bwa, float bwb) __attribute__((always_inline)) -> float {
            auto one = [&](int k) __attribute__((always_inline)) -> float {
                const float bw = (k <= fk) ? bwa : bwb;
                const float a = __builtin_fmaf(-bw, fabsf(ys[k]), 1.0f);
                const float out = ys[k] * g;
                g = __builtin_amdgcn_fmed3f(__builtin_fmaf(g, a, bw), P.agc_min, P.agc_max);
                return out;
            };
            if (mirror) {
#pragma unroll
                for (int k = 0; k < MIR; ++k) { const float o = one(k); wblk[k * LP] = o; wblk[(k + RING) * LP] = o; }
            } else {
#pragma unroll
                for (int k = 0; k < MIR; ++k) wblk[k * LP] = one(k);
            }
#pragma unroll
            for (int k = MIR; k < SB; ++k) wblk[k * LP] = one(k);
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

// =====================================================================================
// Two wavefronts per 64 state columns ("duo"): the same sub-blocks cut at the one place the chain has no per-sample
// feedback across -- wavefront A runs the sample phase (DC blocker, AGC, window push) of sub-block s while wavefront B
// runs the TED instants (matched filters, timing loop, symbol path) of sub-block s - 1, on another SIMD of the same CU,
// through the LDS window ring (four sub-blocks: B's filters reach back over three while A writes the fourth).  The only
// thing that ever travels back is the AGC lock (agc.lock() on sync, unlock in end(), receiver.rs:431, 480): A runs ahead
// on its belief, and when a symbol of sub-block s - 1 flips the lock at sample fk, B posts {fk, new lock}, both meet at
// a barrier, A redoes the AGC of that sub-block from its saved DC outputs and starting gain (bandwidth old up to fk, new
// after) and of the sub-block it has just produced, and B goes on -- to a second instant of the same sub-block, if there
// is one, over the corrected window.  One barrier per sub-block otherwise.  Either wavefront holds half the state: 25 KB
// of LDS per pair and at most 168 registers, twelve wavefronts per CU (three per SIMD).
// =====================================================================================
template <int NT> struct DuoLayout {
    static constexpr int SB = RelaxGeom<NT>::SB, B = 2 * SB, DCL = RelaxGeom<NT>::DCL;
    static constexpr int RING = 4 * SB, MIRROR = kRelaxChunk - 1;
    static constexpr uint32_t tap_floats = (uint32_t)((NT * 4 + 63) / 64 * 64);
    static constexpr uint32_t mail_words = 2u * kWave + 64u;        // feedback word per lane, two rounds; the two flag words
    static constexpr size_t lds_bytes = ((size_t)tap_floats + mail_words + (size_t)(RING + MIRROR) * kWave) * sizeof(float);
    static_assert(NT - 1 + SB <= 3 * SB, "B's filters reach into the sub-block A writes");
};
typedef volatile __attribute__((address_space(3))) uint32_t duo_u32;
__device__ __forceinline__ void duo_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT, int NFF, int NFB, typename SampleT, bool CM, bool TICKS>
__global__ __launch_bounds__(2 * kWave, 2) void demod_duo_kernel(Params P, State S, Output O,
                                                                 const float4 *__restrict__ taps,
                                                                 const SampleT *__restrict__ x,
                                                                 uint32_t n_blocks, uint64_t counter0, PipeChunks K)
{
    using LY = DuoLayout<NT>;
    constexpr int SB = LY::SB, B = LY::B, DCL = LY::DCL, RING = LY::RING, MIR = LY::MIRROR;
    constexpr int HOFF = SB - DCL;
    constexpr uint32_t LP = kWave;
    static_assert(!CM || std::is_same<SampleT, float>::value, "channel-major streams are f32");
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x & (kWave - 1u);
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0: A (sample phase), 1: B (instants)
    float4 *tlds = reinterpret_cast<float4 *>(lds);
    if (role == 0u) for (uint32_t i = lane; i < (uint32_t)NT; i += kWave) tlds[i] = taps[i];
    duo_u32 *mail = (duo_u32 *)(lds + LY::tap_floats);
    duo_u32 *fbbox = mail;                                   // [2][64]: bit 0 lock flipped, bit 1 new lock, bits 8.. sample index
    duo_u32 *flagbox = mail + 2u * kWave;                    // [2]: bit 0 some lane's lock flipped, bit 1 another pass follows, bit 2 leave
    const uint32_t C = P.n_channels;                         // whole groups of 64 (launcher)
    uint32_t c = blockIdx.x * kWave + lane;
    if (K.n_chunks > 1u && K.col_perm) c = K.col_perm[c];
    uint32_t cin = c, Cin = C, n_nominal = n_blocks;
    bool may_leave = false;
    int32_t row_l = 0;
    const float *xl = nullptr;
    uint32_t avail_l = 0xffffffffu;
    if constexpr (CM) {
        Cin = K.in_channels;
        const uint32_t chunk_l = c / Cin;
        cin = c - chunk_l * Cin;
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
    float *wcol = lds + LY::tap_floats + LY::mail_words + lane;      // ring slot 0 of this lane
    const uint32_t n_sub = 2u * n_blocks;
    const uint64_t counter1 = counter0 + (uint64_t)n_blocks * B;

    if (role == 0u) {
        // ------------------------------------------------ A: sample phase of sub-block s ------------------------------
        float sum0 = S.dc_sum0[c], sum1 = S.dc_sum1[c], gain = S.agc_gain[c];
        bool locked = (S.flags[c] & F_AGC_LOCKED) != 0u;             // this wavefront's belief of the AGC lock
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
        float xa[SB + 1], xb[SB - 1], ma[SB], mb[SB];
        float ys0[SB], ys1[SB];                  // DC-blocker outputs of the last even / odd sub-block (for a replay)
        float g00 = gain, g01 = gain;            // the gain each of them started with
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
#pragma unroll
            for (int k = 0; k < SB; ++k) { ys0[k] = 0.0f; ys1[k] = 0.0f; }
        }
        const uint32_t voff = cin * (uint32_t)sizeof(SampleT), row_bytes = Cin * (uint32_t)sizeof(SampleT);
        RX_T0();
        auto load_rows = [&](float *dst, uint32_t b, auto k0_, auto n_) __attribute__((always_inline)) {
            constexpr int k0 = decltype(k0_)::value, N = decltype(n_)::value;
            if constexpr (CM) {
                if (b < avail_l) {
                    const float2 *p = reinterpret_cast<const float2 *>(xl + (size_t)b * B) + k0 / 2;
#pragma unroll
                    for (int j = 0; j < N / 2; ++j) { const float2 v = p[j]; dst[2 * j] = v.x; dst[2 * j + 1] = v.y; }
                } else {
#pragma unroll
                    for (int k = 0; k < N; ++k) dst[k] = 0.0f;
                }
            } else {
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
        // AGC (relaxed) and window push of one sub-block whose first sample has ring slot `base`: bandwidth bwa up to
        // sample fk, bwb after it
        auto agc_pass = [&](const float (&ys)[SB], uint32_t base, float g, int fk, float bwa, float bwb) -> float {
            float *wblk = wcol + base * LP;
            auto one = [&](int k) __attribute__((always_inline)) -> float {
                const float bw = (k <= fk) ? bwa : bwb;
                const float a = __builtin_fmaf(-bw, fabsf(ys[k]), 1.0f);
                const float out = ys[k] * g;
                g = __builtin_amdgcn_fmed3f(__builtin_fmaf(g, a, bw), P.agc_min, P.agc_max);
                return out;
            };
            if (base == 0u) {                                // (wave-uniform) the ring's first MIR slots are stored twice
#pragma unroll
                for (int k = 0; k < MIR; ++k) { const float o = one(k); wblk[k * LP] = o; wblk[(k + RING) * LP] = o; }
            } else {
#pragma unroll
                for (int k = 0; k < MIR; ++k) wblk[k * LP] = one(k);
            }
#pragma unroll
            for (int k = MIR; k < SB; ++k) wblk[k * LP] = one(k);
            return g;
        };
        auto sample_phase = [&](auto par, uint32_t sb) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;
            auto in = [&](int k) -> float { if constexpr (PAR == 0) return xa[k]; else return k == 0 ? xa[SB] : xb[k - 1]; };
            auto xw = [&](int i) -> float {
                if (i >= DCL) return in(i - DCL);
                if constexpr (PAR == 0) return xb[i + HOFF - 1]; else return xa[i + HOFF];
            };
            float (&mnew)[SB] = *(PAR == 0 ? &ma : &mb);
            float (&mold)[SB] = *(PAR == 0 ? &mb : &ma);
            float (&ys)[SB] = *(PAR == 0 ? &ys0 : &ys1);
            auto mw = [&](int k) -> float { return k < DCL ? mold[k + HOFF] : mnew[k - DCL]; };
            // ---- DC blocker rx/dcblock.rs:45-49, 104-108 (strict) ----
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
            const uint32_t blk = sb >> 1;
            if constexpr (PAR == 0) load_b(blk);
            else if (blk + 1u < n_blocks) load_a(blk + 1u);
            RX_LAP(0);
            // ---- AGC + window push, on this wavefront's belief of the lock ----
            if constexpr (PAR == 0) g00 = gain; else g01 = gain;
            const float bw = locked ? 0.0f : P.agc_bw;
            gain = agc_pass(ys, (sb & 3u) * (uint32_t)SB, gain, SB, bw, bw);
        };
        // the lock flipped at sample fk of sub-block sb - 1 (the one before the sub-block just produced, sb; sb == n_sub:
        // nothing was produced after it): redo the AGC from the sub-block's saved start
        auto replay = [&](auto par_prev, uint32_t sb, int fk, bool new_locked, bool mine) __attribute__((always_inline)) {
            constexpr int PP = decltype(par_prev)::value;
            if (!mine) return;
            const float bw0 = locked ? 0.0f : P.agc_bw;
            locked = new_locked;
            const float bw1 = locked ? 0.0f : P.agc_bw;
            float g = agc_pass(*(PP == 0 ? &ys0 : &ys1), ((sb - 1u) & 3u) * (uint32_t)SB, PP == 0 ? g00 : g01, fk, bw0, bw1);
            if (sb < n_sub) {
                if constexpr (PP == 0) g01 = g; else g00 = g;
                g = agc_pass(*(PP == 0 ? &ys1 : &ys0), (sb & 3u) * (uint32_t)SB, g, -1, bw1, bw1);
            }
            gain = g;
        };
        load_a(0u);
        duo_barrier();                                       // taps and the carried-over window are in LDS
        uint32_t round = 0;
        bool left = false;
        auto step = [&](auto par, uint32_t sb) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;
            if (sb < n_sub) sample_phase(par, sb);
            RX_LAP(1); RX_COUNT(0);
            uint32_t fw;
            do {
                duo_barrier();                               // B's pass over sub-block sb - 1 is done
                RX_LAP(5);
                fw = (uint32_t)__builtin_amdgcn_readfirstlane((int)flagbox[round & 1u]);
                if (fw & 1u) {
                    const uint32_t v = fbbox[(round & 1u) * kWave + lane];
                    replay(std::integral_constant<int, PAR ^ 1>{}, sb, (int)(v >> 8), (v & 2u) != 0u, (v & 1u) != 0u);
                    duo_barrier();                           // the window is corrected
                }
                ++round;
            } while (fw & 2u);
            left = (fw & 4u) != 0u;
        };
        for (uint32_t sb = 0; sb <= n_sub && !left; sb += 2u) {
            step(std::integral_constant<int, 0>{}, sb);
            if (!left && sb + 1u <= n_sub) step(std::integral_constant<int, 1>{}, sb + 1u);
        }
        RX_REPORT();
        if (left) return;
        // ---- A's share of the state ----
        S.dc_sum0[c] = sum0; S.dc_sum1[c] = sum1; S.agc_gain[c] = gain;
        {
            const uint32_t G = P.win_ring;
            const uint32_t wnext = (n_sub & 3u) * (uint32_t)SB;       // ring slot of the next sample
#pragma unroll 2
            for (uint32_t m = 1; m <= G; ++m) {
                const uint32_t g = (uint32_t)(counter1 - (uint64_t)m) & (G - 1u);
                float *row = S.win_ring + (size_t)g * C;
                const uint32_t j = wnext >= m ? wnext - m : wnext + (uint32_t)RING - m;
                row[c] = m <= (uint32_t)RING ? wcol[j * LP] : 0.0f;       // (a state ring longer than this kernel's: nothing older is kept)
            }
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
    } else {
        // ------------------------------------------------ B: the TED instants of sub-block s - 1 -----------------------
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
        const float inv_spt = 1.0f / P.samples_per_ted;
        int cstar = next_fire_count(L.until_next_ted, L.ted_clock);
        int until = cstar - (int)L.ted_clock - 1;            // index of the next instant, relative to the sub-block about to be processed
        bool lane_done = false;
        uint32_t round = 0;
        flagbox[0] = 0u; flagbox[1] = 0u;
        duo_barrier();                                       // (A's prologue)
        duo_barrier();                                       // step 0: A produces sub-block 0, nothing to do here
        ++round;
        bool left = false;
        RX_T0();
        for (uint32_t sb = 0; sb < n_sub && !left; ++sb) {
            const uint32_t base = (sb & 3u) * (uint32_t)SB, blk = sb >> 1;
            const uint32_t pslot = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;
            const float hpre0 = X.hist_get((pslot + 16u) & 63u), hpre1 = X.hist_get((pslot + 17u) & 63u);
            bool more;
            do {
                uint32_t fbv = 0u;
                if (until < SB) {
                    const int fk = until;
                    const float sa_low = demod_relaxed<NT, RING, false>(taps_lds, wcol_lds, (int)base + fk);
                    RX_LAP(2); RX_COUNT(1);
                    const float rem = L.until_next_ted - (float)cstar;          // receiver.rs:352
                    const uint32_t locked_before = L.flags & F_AGC_LOCKED;
                    float zero, sym, terr;
                    if (ted_timing_relaxed(P, L, inv_spt, sa_low, rem, &zero, &sym, &terr)) {
                        const uint64_t counter = counter0 + (int64_t)row_l + (uint64_t)sb * SB + (uint32_t)fk + 1u;
                        uint32_t burst_len = 0;
                        bool emit = false;
                        const uint32_t link = symbol_link(P, L, S, X, c, zero, sym, terr, L.until_next_ted, counter, &burst_len, &emit, true, hpre0, hpre1);
                        if constexpr (TICKS) symbol_io(P, L, S, O, X, c, link, emit, counter, burst_len);
                        else if (emit) X.emit(P, S, O, c, link, counter, L.sq_symbols, burst_len);
                    }
                    cstar = next_fire_count(L.until_next_ted, 0u);
                    until = fk + cstar;
                    const uint32_t locked_after = L.flags & F_AGC_LOCKED;
                    if (locked_after != locked_before) fbv = 1u | (locked_after ? 2u : 0u) | ((uint32_t)fk << 8);
                    RX_LAP(3);
                }
                more = __builtin_amdgcn_ballot_w64(until < SB) != 0ull;
                const bool any_fb = __builtin_amdgcn_ballot_w64(fbv != 0u) != 0ull;
                uint32_t leave = 0u;
                if (!more && (sb & 1u) && may_leave) {
                    // end of a block of a chunk that hands over (see demod_relaxed_kernel)
                    if (!lane_done && blk + 1u >= n_nominal && (L.flags & F_LINK_MASK) == 0u && blk < avail_l) {
                        lane_done = true;
                        K.handover[c] = counter0 + (int64_t)row_l + (uint64_t)(blk + 1u) * B;
                    }
                    if (__builtin_amdgcn_ballot_w64(!lane_done) == 0ull) leave = 4u;
                }
                if (any_fb) fbbox[(round & 1u) * kWave + lane] = fbv;
                if (lane == 0u) flagbox[round & 1u] = (any_fb ? 1u : 0u) | (more ? 2u : 0u) | leave;
                duo_barrier();
                if (any_fb) duo_barrier();                   // A has corrected the window
                RX_LAP(4);
                ++round;
                left = leave != 0u;
            } while (more);
            until -= SB;
        }
        RX_REPORT();
        if (left) return;
        // the last round: A's step n_sub has nothing to produce and waits for a flag word
        // (already posted: the loop above ran n_sub sub-blocks = A's steps 1 .. n_sub)
        L.ted_clock = (uint32_t)(cstar - until - 1);
        // ---- B's share of the state (sums and gain are A's) ----
        S.until_next_ted[c] = L.until_next_ted; S.ted_clock[c] = L.ted_clock;
        S.ted_h0[c] = L.h0; S.ted_h1[c] = L.h1; S.ted_h2[c] = L.h2;
        S.period_avg[c] = L.period_avg; S.period_inst[c] = L.period_inst;
        S.sq_data[c] = L.sq_data; S.sq_power[c] = L.sq_power; S.sq_phist[c] = L.sq_phist;
        S.sq_fill[c] = L.sq_fill; S.sq_clock[c] = L.sq_clock; S.sq_symbols[c] = L.sq_symbols;
        S.eq_word[c] = L.eq_word; S.eq_count[c] = L.eq_count;
        S.fr_word[c] = L.fr_word; S.fr_count[c] = L.fr_count; S.fr_invalid[c] = L.fr_invalid;
        S.fr_len[c] = L.fr_len; S.flags[c] = L.flags;
        if constexpr (TICKS) { S.tk_next[c] = L.tk_next; S.tk_last[c] = L.tk_last; S.wake_fired[c] = L.wake_fired; }
#pragma unroll
        for (int i = 0; i < NFF; ++i) { S.eq_ffc[i * C + c] = X.ffc[i]; S.eq_ffw[i * C + c] = X.ffw[i]; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { S.eq_fbc[i * C + c] = X.fbc[i]; S.eq_fbw[i * C + c] = X.fbw[i]; }
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
// (Since round 4 these kernels take relaxed batches that are not whole groups of 64 channels, time-parallel calls beyond
// 16 384 channels and what SAME_RELAXED_KERNEL sends them; everything else runs the symbol-paced pipeline, which is faster
// at every channel count under sustained launches: same_batch.cpp.)
// Which form runs a launch over P.n_channels state columns: 1 duo (two wavefronts per 64 columns) while that leaves the
// launch at no more than two wavefronts per SIMD (65 536 columns), 0 solo beyond -- or what SAME_RELAXED_KERNEL asks for.  Whole groups
// of 64 columns for duo.  (A third form -- sample phase | filters + timing loop | symbol path on three wavefronts, 18-sample
// sub-blocks, the symbol stage on every step or on every other -- was built and measured in round 3: 3.9-4.1 ms where the
// pipeline's FASTMATH build takes 3.8; not kept.)
uint32_t relaxed_kernel_kind(const Params &P)
{
    const bool whole = (P.n_channels % kWave) == 0u;
    if (P.knob_relaxed_kernel == 1 || !whole) return 0u;
    if (P.knob_relaxed_kernel == 2) return 1u;
    return P.n_channels <= 65536u ? 1u : 0u;      // (measured, 2 s launches: 65 536 columns duo 6.5 against solo 7.0 ms; 81 920: 11.8 against 9.0; 98 304: 12.1 against 9.8)
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
template <int NFF, int NFB, typename SampleT, bool CM, bool TICKS>
static void launch_duo_one(const Params &P, const State &S, const Output &O, const float4 *taps, const SampleT *x,
                           uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{
    hipLaunchKernelGGL((demod_duo_kernel<42, NFF, NFB, SampleT, CM, TICKS>), dim3(P.n_channels / kWave), dim3(2 * kWave), DuoLayout<42>::lds_bytes,
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
    const uint32_t kind = relaxed_kernel_kind(P);
    if (kind == 1u) {
#define SAME_DUO_GO(NFF, NFB, CM_, TK) launch_duo_one<NFF, NFB, SampleT, CM_, TK>(P, S, O, taps, x, n_blocks, counter0, stream, K)
        if constexpr (std::is_same<SampleT, float>::value) {
            if (cm) { if (eq64) SAME_DUO_GO(6, 4, true, false); else SAME_DUO_GO(1, 1, true, false); return hipGetLastError(); }
        }
        if (eq64) { if (ticks) SAME_DUO_GO(6, 4, false, true); else SAME_DUO_GO(6, 4, false, false); }
        else { if (ticks) SAME_DUO_GO(1, 1, false, true); else SAME_DUO_GO(1, 1, false, false); }
#undef SAME_DUO_GO
        return hipGetLastError();
    }
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
