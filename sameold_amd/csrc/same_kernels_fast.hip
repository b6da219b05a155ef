// same_kernels_fast.hip -- latency-optimised demodulation kernel for the standard sample
// rates (22.05 / 44.1 / 48 kHz with the reference's default DC-blocker length).
//
// Same arithmetic contract and the same "deferred TED" schedule as demod_kernel in
// same_kernels.hip (one lane = one channel, blocks of 16 / 18 / 32 samples -- see
// same_fast_common.h -- with at most one TED instant per lane per block), so results are
// bit-identical; what changes is where state lives and how memory latency is taken off the
// critical path.  One wavefront owns 64 channels for the whole launch; it is the variant for
// batches that fill the machine (from 49 152 channels at 22.05 kHz) and for 44.1 / 48 kHz.
// Smaller 22.05 kHz batches go to the wavefront pipeline of same_kernels_pipe.hip.
//
//   * DC blocker (22.05 kHz: window length 16 <= block length): both moving-average
//     windows are the last 16 inputs / 16 averages, held in VGPRs with static indices -- no
//     LDS traffic at all.  Other rates read all of a block's aged-off ring entries from LDS
//     up front (they cannot alias this block's writes because the window is longer than a
//     block).
//   * the next block's input samples are fetched while the current block is computed;
//   * the TED instant of a lane is known in advance (the sample clock is a counter
//     against a fixed period), so the per-sample clock test collapses to one compare per
//     block;
//   * matched filters: window samples and taps are pulled from LDS in batches (the window
//     mirrored for 42 taps, so no per-tap address arithmetic) and the four accumulation
//     chains run as two v_pk_mul_f32 / v_pk_add_f32 pairs per tap (per-element IEEE,
//     identical rounding to the scalar form);
//   * squelch sample history lives in LDS, the equalizer's 20 floats in VGPRs.
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "same_dev_common.h"
#include "same_device.h"
#include "same_fast_common.h"
#include "same_launch.h"

namespace same {


// DENSE: built for two wavefronts per SIMD (batches of more than 1 024 wavefronts, i.e. beyond 65 536 channels): at
// most 256 registers, and the squelch sample history -- 16 of the 33 KB of LDS a wavefront takes, touched four times
// per symbol -- stays in the HBM state array it comes from, so that eight wavefronts fit a CU's LDS instead of four.
template <int NT, int DCL, int NFF, int NFB, bool MED3, bool MIRROR, typename SampleT, bool DENSE = false>
__global__ __launch_bounds__(kWave, DENSE ? 2 : 1) void demod_fast_kernel(Params P, State S, Output O,
                                                           const float4 *__restrict__ taps,
                                                           const SampleT *__restrict__ x,
                                                           uint32_t n_blocks, uint64_t counter0)
{
    constexpr int kB = FastBlock<NT, MIRROR>::len;
    constexpr int RING = FastRing<NT, MIRROR>::slots;
    constexpr bool DC_REGS = (DCL <= kB) && DCL <= 16;      // the windows in registers: short ones only
    static_assert(DC_REGS || DCL >= kB, "LDS DC path needs a window at least as long as a block");
    static_assert(!DC_REGS || (DCL & (DCL - 1)) == 0, "register DC path indexes the state ring with a mask");
    static_assert(NT + kB - 1 <= RING && RING % kB == 0, "window ring too small or not a whole number of blocks");
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t C = P.n_channels;
    constexpr uint32_t LP = kWave;                               // one lane per channel
    // matched-filter taps: [NT] float4 at the start of LDS, staged by the first NT lanes (before
    // the tail-wave exit below, so every wavefront has them)
    float4 *tlds = reinterpret_cast<float4 *>(lds);
    for (uint32_t i = lane; i < (uint32_t)NT; i += kWave) tlds[i] = taps[i];
    constexpr uint32_t TAPF = (uint32_t)((NT * 4 + PIPE_PROF_TAP_PAD + 63) / 64 * 64);   // floats reserved for the taps
    const uint32_t c = blockIdx.x * kWave + lane;
    if (c >= C) return;                                          // no barriers below
    // LDS: taps | squelch history [64][64] | DC rings [2*DCL][64] (LDS DC path) | window.
    // Mirrored window: logical slots 0 .. 2*RING-1, of which the first block (0 .. kB-1) is
    // never read (the filters reach down to slot RING - NT + 1 at most) and is not stored.
    constexpr int WSKIP = MIRROR ? kB : 0;
    static_assert(!MIRROR || kB <= RING - NT + 1, "the first block's low copy would be read");
    constexpr int HROWS = DENSE ? 0 : kSquelchHist;              // rows of LDS the squelch history takes
    float *hcol = lds + TAPF + lane;                             // [64][64] (not DENSE)
    float *ffcol = hcol + HROWS * LP;                            // [DCL][64] (LDS DC path)
    float *fbcol = ffcol + DCL * LP;
    float *wring = lds + TAPF + (HROWS + (DC_REGS ? 0 : 2 * DCL) - WSKIP) * LP;   // logical slot 0
    float *wcol = wring + lane;

    Lane L;
    lane_load(L, S, c);
    FastCtx<NFF, NFB> X;
    if constexpr (DENSE) { X.hist = S.sq_hist + c; X.hstride = C; } else { X.hist = hcol; }
    FAST_MARKS_BEGIN(X, lds, NT);
#pragma unroll
    for (int i = 0; i < NFF; ++i) {
        X.ffc[i] = S.eq_ffc[i * C + c]; X.ffw[i] = S.eq_ffw[i * C + c];
        X.sffc[i] = S.eq_snap_ffc[i * C + c]; X.sffw[i] = S.eq_snap_ffw[i * C + c];
    }
#pragma unroll
    for (int i = 0; i < NFB; ++i) {
        X.fbc[i] = S.eq_fbc[i * C + c]; X.fbw[i] = S.eq_fbw[i * C + c];
        X.sfbc[i] = S.eq_snap_fbc[i * C + c]; X.sfbw[i] = S.eq_snap_fbw[i * C + c];
    }
    // state arrays are [slot][channel]: a wave-uniform row pointer plus the lane's channel
    // index keeps the address arithmetic on the scalar unit
    // the window ring is re-based: LDS slot j holds the state's slot (j + counter0) mod RING,
    // so the first sample of this launch lands in slot 0 (see demod_fast)
    // (sample counter0 - m, m = 1.., sits in the state's slot (counter0 - m) mod win_ring and in
    // LDS slot RING - m; the LDS ring is at least as long as the state's)
    const uint32_t G = P.win_ring;
#pragma unroll 2
    for (uint32_t m = 1; m <= G; ++m) {
        const uint32_t g = (uint32_t)(counter0 - (uint64_t)m) & (G - 1u);
        const float *row = S.win_ring + (size_t)g * C;
        const float v = row[c];
        if (!MIRROR || (uint32_t)RING - m >= (uint32_t)WSKIP) wcol[((uint32_t)RING - m) * LP] = v;
        if (MIRROR) wcol[(2u * (uint32_t)RING - m) * LP] = v;
    }
#pragma unroll 2
    for (int i = 0; i < HROWS; ++i) { const float *row = S.sq_hist + (size_t)i * C; hcol[i * LP] = row[c]; }

    // ring positions common to all channels
    uint32_t dpos = (uint32_t)(counter0 % (uint64_t)DCL);
    uint32_t wpos = 0;                 // LDS ring slot of the block's first sample: a multiple of 16
    constexpr int DCR = DC_REGS ? DCL : 1;
    float xp[DCR], mp[DCR];            // DC_REGS: the last DCL inputs / first-stage averages, oldest first
    if constexpr (DC_REGS) {
#pragma unroll
        for (int k = 0; k < DCR; ++k) {
            const uint32_t slot = (dpos + (uint32_t)k) & (uint32_t)(DCR - 1);
            const float *r0 = S.dc_ff_ring + (size_t)slot * C, *r1 = S.dc_fb_ring + (size_t)slot * C;
            xp[k] = r0[c];
            mp[k] = r1[c];
        }
    } else {
#pragma unroll 2
        for (int i = 0; i < DCL; ++i) {
            const float *r0 = S.dc_ff_ring + (size_t)i * C, *r1 = S.dc_fb_ring + (size_t)i * C;
            ffcol[i * LP] = r0[c];
            fbcol[i * LP] = r1[c];
        }
    }

    int cstar = next_fire_count(L.until_next_ted, L.ted_clock);
    int until = cstar - (int)L.ted_clock - 1;      // block-relative index of the firing sample

    float xn[kB];
#pragma unroll
    for (int k = 0; k < kB; ++k) { const SampleT *row = x + (size_t)k * C; xn[k] = (float)row[c]; }

    FAST_MARKS_START(X);
    for (uint32_t blk = 0; blk < n_blocks; ++blk) {
        // DENSE: the two history samples a symbol's equalizer step takes (rx_symbol: slots +16 / +17 from the squelch's
        // write position) come from HBM; fetched here, a block's worth of work ahead of their use.  A block completes
        // at most one symbol, and only a symbol writes the history, so they cannot go stale in between.
        [[maybe_unused]] float hpre0 = 0.0f, hpre1 = 0.0f;
        if constexpr (DENSE) {
            const uint32_t pslot = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;
            hpre0 = X.hist_get((pslot + 16u) & 63u); hpre1 = X.hist_get((pslot + 17u) & 63u);
        }
        float xs[kB];
#pragma unroll
        for (int k = 0; k < kB; ++k) xs[k] = xn[k];
        if (blk + 1 < n_blocks) {
            const SampleT *xb = x + ((size_t)(blk + 1) * kB) * C;      // wave-uniform
#pragma unroll
            for (int k = 0; k < kB; ++k) { const SampleT *row = xb + (size_t)k * C; xn[k] = (float)row[c]; }
        }

        // ---- DC blocker rx/dcblock.rs:45-49, 104-108 --------------------------------
        float ys[kB];
        if constexpr (DC_REGS) {
            float mnew[kB];
            // two samples per step: only the running sums are serial, the differences and
            // scalings of samples k and k+1 are independent and ride in packed f32 operations
            // (per-element IEEE, same roundings as the scalar form).  Entry i of the window as
            // it stood before this block is xp[i] for i < DCL and this block's xs[i - DCL] after.
            auto xw = [&](int i) { return i < DCL ? xp[i < DCL ? i : 0] : xs[i >= DCL ? i - DCL : 0]; };
#pragma unroll
            for (int k = 0; k < kB; k += 2) {
                const float2v x2 = {xs[k], xs[k + 1]}, xo = {xw(k), xw(k + 1)};
                const float2v d0 = x2 - xo;                  // input - aged
                const float s0a = L.sum0 + d0.x, s0b = s0a + d0.y;
                L.sum0 = s0b;
                const float2v s0 = {s0a, s0b}, inv = {P.dc_inv_len, P.dc_inv_len};
                const float2v ma0 = s0 * inv;
                // window.front() after the push
                const float2v sig = {xw(k + 1), xw(k + 2)};
                const float2v mo = {k < DCL ? mp[k < DCL ? k : 0] : mnew[k >= DCL ? k - DCL : 0],
                                    k + 1 < DCL ? mp[k + 1 < DCL ? k + 1 : 0] : mnew[k + 1 >= DCL ? k + 1 - DCL : 0]};
                const float2v d1 = ma0 - mo;
                const float s1a = L.sum1 + d1.x, s1b = s1a + d1.y;
                L.sum1 = s1b;
                const float2v s1 = {s1a, s1b};
                const float2v ma1 = s1 * inv;
                const float2v y2 = sig - ma1;                // (len > 1) as f32 == 1.0: exact
                ys[k] = y2.x; ys[k + 1] = y2.y;
                mnew[k] = ma0.x; mnew[k + 1] = ma0.y;
            }
#pragma unroll
            for (int k = 0; k < DCR; ++k) { xp[k] = xs[kB - DCR + k]; mp[k] = mnew[kB - DCR + k]; }
        } else {
            float a0[kB + 1], a1[kB];
            uint32_t slots[kB + 1];
#pragma unroll
            for (int k = 0; k <= kB; ++k) {
                uint32_t s = dpos + (uint32_t)k;
                slots[k] = (s >= (uint32_t)DCL) ? s - (uint32_t)DCL : s;
            }
#pragma unroll
            for (int k = 0; k <= kB; ++k) a0[k] = ffcol[slots[k] * LP];
            // window exactly one block long: the entry aged at the next block's first step is this
            // block's first input, which is not in the ring yet
            if (DCL == kB) a0[kB] = xs[0];
#pragma unroll
            for (int k = 0; k < kB; ++k) a1[k] = fbcol[slots[k] * LP];
#pragma unroll
            for (int k = 0; k < kB; ++k) {
                float d0 = xs[k] - a0[k];
                L.sum0 += d0;
                float ma0 = L.sum0 * P.dc_inv_len;
                float sig = a0[k + 1];                       // read before this block's writes
                float d1 = ma0 - a1[k];
                L.sum1 += d1;
                float ma1 = L.sum1 * P.dc_inv_len;
                ys[k] = sig - ma1;
                ffcol[slots[k] * LP] = xs[k];
                fbcol[slots[k] * LP] = ma0;
            }
            dpos = slots[kB];
        }

        // ---- AGC rx/agc.rs:72-77 and window push receiver.rs:345-346 -----------------
        const float g0 = L.gain;
        float *wblk = wcol + wpos * LP;
        // mirrored window: the low copy of the ring's first block is not stored (never read);
        // its writes are pointed at the high copy instead, which takes the same value twice
        float *wlow = wcol + ((MIRROR && wpos == 0u) ? (uint32_t)RING : wpos) * LP;
        const float bw0 = (L.flags & F_AGC_LOCKED) ? 0.0f : P.agc_bw;
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            float out = agc_step<MED3>(P, ys[k], L.gain, bw0);
            wlow[k * LP] = out;
            if (MIRROR) wblk[(k + RING) * LP] = out;
        }

        X.mark(0);
        // ---- deferred TED instant ----------------------------------------------------
        if (until < kB) {
            const int fk = until;
            const uint32_t newest = wpos + (uint32_t)fk;
            const float sa_low = demod_fast<NT, RING, MIRROR>(tlds, wring, lane, newest);
            X.mark(1);
            const float rem = L.until_next_ted - (float)cstar;          // receiver.rs:352
            const uint32_t locked_before = L.flags & F_AGC_LOCKED;
            ted_instant(P, L, S, O, X, c, sa_low, rem,
                        counter0 + (uint64_t)blk * kB + (uint32_t)fk + 1u, DENSE, hpre0, hpre1);
            cstar = next_fire_count(L.until_next_ted, 0u);
            until = fk + cstar;
            if ((L.flags & F_AGC_LOCKED) != locked_before) {
                // the lock changed at sample fk: redo the block's AGC with the old lock up
                // to fk and the new one after it (rare: twice per burst)
                const float bw1 = (L.flags & F_AGC_LOCKED) ? 0.0f : P.agc_bw;
                float g = g0;
#pragma unroll
                for (int k = 0; k < kB; ++k) {
                    float out = agc_step<MED3>(P, ys[k], g, (k <= fk) ? bw0 : bw1);
                    wlow[k * LP] = out;
                    if (MIRROR) wblk[(k + RING) * LP] = out;
                }
                L.gain = g;
            }
        }
        until -= kB;
        wpos += kB;
        if (wpos == (uint32_t)RING) wpos = 0;
        X.mark(7);
        X.mark(8);                     // (two marks back to back: what a mark itself costs)
    }
    FAST_MARKS_REPORT(X);

    // ---- write the state back --------------------------------------------------------
    L.ted_clock = (uint32_t)(cstar - until - 1);
    lane_store(L, S, c);
#pragma unroll
    for (int i = 0; i < NFF; ++i) {
        S.eq_ffc[i * C + c] = X.ffc[i]; S.eq_ffw[i * C + c] = X.ffw[i];
        S.eq_snap_ffc[i * C + c] = X.sffc[i]; S.eq_snap_ffw[i * C + c] = X.sffw[i];
    }
#pragma unroll
    for (int i = 0; i < NFB; ++i) {
        S.eq_fbc[i * C + c] = X.fbc[i]; S.eq_fbw[i * C + c] = X.fbw[i];
        S.eq_snap_fbc[i * C + c] = X.sfbc[i]; S.eq_snap_fbw[i * C + c] = X.sfbw[i];
    }
    const uint64_t counter1 = counter0 + (uint64_t)n_blocks * kB;
#pragma unroll 2
    for (uint32_t m = 1; m <= G; ++m) {
        const uint32_t g = (uint32_t)(counter1 - (uint64_t)m) & (G - 1u);
        const uint32_t j = wpos >= m ? wpos - m : wpos + (uint32_t)RING - m;
        float *row = S.win_ring + (size_t)g * C;
        row[c] = wcol[(MIRROR ? j + (uint32_t)RING : j) * LP];       // the high copy is always there
    }
#pragma unroll 2
    for (int i = 0; i < HROWS; ++i) { float *row = S.sq_hist + (size_t)i * C; row[c] = hcol[i * LP]; }
    if constexpr (DC_REGS) {
        dpos = (uint32_t)(counter1 % (uint64_t)DCL);
#pragma unroll
        for (int k = 0; k < DCR; ++k) {
            const uint32_t slot = (dpos + (uint32_t)k) & (uint32_t)(DCR - 1);
            float *r0 = S.dc_ff_ring + (size_t)slot * C, *r1 = S.dc_fb_ring + (size_t)slot * C;
            r0[c] = xp[k];
            r1[c] = mp[k];
        }
    } else {
#pragma unroll 2
        for (int i = 0; i < DCL; ++i) {
            float *r0 = S.dc_ff_ring + (size_t)i * C, *r1 = S.dc_fb_ring + (size_t)i * C;
            r0[c] = ffcol[i * LP];
            r1[c] = fbcol[i * LP];
        }
    }
}

// ---------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------
template <int NT, int DCL, bool MIRROR, bool DENSE = false>
static constexpr size_t fast_lds_bytes()
{
    constexpr int kB = FastBlock<NT, MIRROR>::len;
    constexpr int RING = FastRing<NT, MIRROR>::slots;
    constexpr size_t TAPF = (size_t)((NT * 4 + PIPE_PROF_TAP_PAD + 63) / 64 * 64);
    return (TAPF + (size_t)((MIRROR ? 2 * RING - kB : RING) + (DENSE ? 0 : kSquelchHist) + ((DCL <= kB && DCL <= 16) ? 0 : 2 * DCL)) * kWave) * sizeof(float);
}

// The mirrored window costs 16 KB of LDS per wavefront: 3 wavefronts fit a CU's 160 KB instead
// of 4, so it is used while the batch needs at most 3 wavefronts per CU (and only for the
// 42-tap filters, whose ring is 64 slots).
static bool fast_use_mirror(const Params &P, uint32_t max_block)
{
    if (max_block < (uint32_t)kBlockMirror) return false;     // 18-sample blocks need the timing bound
    if (P.knob_mirror != 0) return P.knob_mirror > 0 && P.ntaps == 42u;
    return P.ntaps == 42u && (P.n_channels + kWave - 1) / kWave <= 3u * 256u;
}

template <int NT, int DCL, typename SampleT>
static hipError_t launch_fast_cfg(const Params &P, const State &S, const Output &O, const float4 *taps,
                                  const SampleT *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream)
{
    const uint32_t grid = (P.n_channels + kWave - 1) / kWave;
    constexpr bool CAN_MIRROR = (NT == 42);
    const bool mirror = CAN_MIRROR && fast_use_mirror(P, max_block_len(P));
    // more wavefronts than one per SIMD (1 024): the dense build, two per SIMD (22.05 kHz; SAME_FAST_DENSE=0/1 overrides)
    constexpr bool CAN_DENSE = (NT == 42);
    const bool dense = CAN_DENSE && !mirror && (P.knob_fast_dense != 0 ? P.knob_fast_dense > 0 : grid > 1024u);
    const size_t lds = mirror ? fast_lds_bytes<NT, DCL, CAN_MIRROR>() : (dense ? fast_lds_bytes<NT, DCL, false, CAN_DENSE>() : fast_lds_bytes<NT, DCL, false>());
    // v_med3_f32 == f32::clamp unless a bound is -0.0 (or NaN, which the builder rejects)
    const bool med3 = !(P.agc_min == 0.0f && std::signbit(P.agc_min)) && !(P.agc_max == 0.0f && std::signbit(P.agc_max));
#define SAME_FAST_M(NFF, NFB, M3, MI)                                                                       \
    hipLaunchKernelGGL((demod_fast_kernel<NT, DCL, NFF, NFB, M3, MI, SampleT>), dim3(grid), dim3(kWave), lds, \
                       stream, P, S, O, taps, x, n_blocks, counter0)
#define SAME_FAST(NFF, NFB, M3)                                                                         \
    do {                                                                                                \
        if (mirror) SAME_FAST_M(NFF, NFB, M3, CAN_MIRROR);                                              \
        else if (dense) hipLaunchKernelGGL((demod_fast_kernel<NT, DCL, NFF, NFB, M3, false, SampleT, CAN_DENSE>), dim3(grid), dim3(kWave), lds, \
                                           stream, P, S, O, taps, x, n_blocks, counter0);               \
        else SAME_FAST_M(NFF, NFB, M3, false);                                                          \
    } while (0)
    if (P.eq_nff == 6u && P.eq_nfb == 4u) { if (med3) SAME_FAST(6, 4, true); else SAME_FAST(6, 4, false); }
    else { if (med3) SAME_FAST(1, 1, true); else SAME_FAST(1, 1, false); }
#undef SAME_FAST
#undef SAME_FAST_M
    return hipGetLastError();
}

bool fast_kernel_supported(const Params &P)
{
    if (P.block_len != (uint32_t)kBlock || P.win_ring > 128u) return false;
    const bool eq_ok = (P.eq_nff == 6u && P.eq_nfb == 4u) || (P.eq_nff == 1u && P.eq_nfb == 1u);
    if (!eq_ok) return false;
    if (P.ntaps >= 84u && max_block_len(P) < (uint32_t)kBlock48k) return false;   // their block is 32 samples
    return (P.ntaps == 42u && P.dc_len == 16u) || (P.ntaps == 92u && P.dc_len == 35u) ||
           (P.ntaps == 84u && P.dc_len == 32u);
}

uint32_t fast_win_ring(const Params &P) { return (P.ntaps + kBlock - 1 <= 64u) ? 64u : 128u; }

// samples per block of the variant launch_demod_fast will pick for this batch
uint32_t fast_block_len(const Params &P)
{
    if (pipe_kernel_selected(P)) return pipe_block_len(P);
    if (P.ntaps == 42u && fast_use_mirror(P, max_block_len(P))) return (uint32_t)kBlockMirror;
    return P.ntaps >= 84u ? (uint32_t)kBlock48k : (uint32_t)kBlock;
}

template <typename SampleT>
static hipError_t launch_fast_t(const Params &P, const State &S, const Output &O, const float4 *taps,
                                const SampleT *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream)
{
    if (P.ntaps == 42u) return launch_fast_cfg<42, 16, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream);
    if (P.ntaps == 92u) return launch_fast_cfg<92, 35, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream);
    return launch_fast_cfg<84, 32, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream);
}

hipError_t launch_demod_fast(const Params &P, const State &S, const Output &O, const float4 *taps,
                             const float *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream)
{ return launch_fast_t<float>(P, S, O, taps, x, n_blocks, counter0, stream); }
hipError_t launch_demod_fast_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                                 const int16_t *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream)
{ return launch_fast_t<int16_t>(P, S, O, taps, x, n_blocks, counter0, stream); }

}  // namespace same

FAST_PROFILE_EXPORTS()
