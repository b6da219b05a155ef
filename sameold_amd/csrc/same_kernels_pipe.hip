// same_kernels_pipe.hip -- four-stage wavefront pipeline for small and medium batches at the
// standard rates (22.05 / 44.1 / 48 kHz).
//
// With one wavefront per 64 channels (same_kernels_fast.hip) a launch of 4 096 channels is 64
// serial instruction streams on a machine with 1 024 SIMDs, and a stream's length per block is
// what it is whichever lanes are live.  Here a workgroup of FOUR wavefronts owns up to 64
// channels: the stream is cut in stages that run concurrently on different SIMDs of one CU, one
// block apart, handing data over through LDS (see "Four stages" below).  The arithmetic per
// channel and its order are exactly those of the one-wavefront kernels, so results are
// bit-identical (tests/test_gpu_parity.py runs every standard-rate case of up to 32 768 channels
// through this kernel; test_fast_kernel_equals_generic_kernel,
// test_pipeline_kernel_equals_single_wavefront_kernel and test_pipeline_workgroup_widths pit the
// variants against each other).  At 4 096 channels x 10 s, 22.05 kHz: 35.3 ms (one wavefront per
// 64 channels) -> 27.8 ms (two stages: sample phase | everything else) -> 19.2 ms (three) ->
// 17.3 ms (four) -> 15.1 ms (16 channels per workgroup, LANES) -> 13.9 ms (stage 2's space
// filter on stage 4's wavefront, SPLIT) -> 13.2 ms (20-sample blocks with the rare second instant
// of a block handled in stage 2, same_fast_common.h).
//
// Window ring: 5 blocks (of 20 slots at 22.05 kHz, 32 at 44.1 / 48 kHz), mirrored (see
// same_fast_common.h): while stage 2 reads the NT slots ending at an instant of block i, stage 1
// writes block i+1, and with five blocks the two never touch the same slot (four would: an
// instant early in block i still needs the tail of block i-3, which block i+1 overwrites).
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "same_dev_common.h"
#include "same_device.h"
#include "same_fast_common.h"
#include "same_launch.h"
#include "same_pipe_common.h"
#include "same_profile.h"
#include "same_relaxed_common.h"

namespace same {

// Geometry per sample rate (filter length NT): DC-blocker window, samples per block, window ring.
// The ring is five blocks at every rate: NT - 1 samples back from an instant early in block s-1
// reach into block s-4 (41 = 2*20 + 1, 91 = 2*32 + 27, 83 = 2*32 + 19) while stage 1 writes block s.
// Samples per block at 48 / 44.1 kHz (bounds 43 / 39, one instant per block).  Measured at 16 384
// channels x 2 s: 48 kHz 6.03 ms with 32, 6.75 with 36, 6.48 with 40 (stage 1's registers);
// 44.1 kHz 5.35 ms with 32, 5.20 with 36.
constexpr int kBlockPipe48 = 32, kBlockPipe44 = 36;
template <int NT> struct PipeGeom;
template <> struct PipeGeom<42> { static constexpr int DCL = 16, B = kBlockPipe22; };   // 22.05 kHz
template <> struct PipeGeom<92> { static constexpr int DCL = 35, B = kBlockPipe48; };       // 48 kHz
template <> struct PipeGeom<84> { static constexpr int DCL = 32, B = kBlockPipe44; };       // 44.1 kHz
template <int NT> struct PipeLayout {
    static constexpr int B = PipeGeom<NT>::B, RING = 5 * B;
    // stage 1 keeps the DC-blocker outputs of its last three blocks for a replay: in registers
    // (3 x 20) at 22.05 kHz, in an LDS ring of three blocks at 44.1 / 48 kHz, where 3 x 32 more
    // registers per lane would spill and the CU's LDS has room
    static constexpr bool YLDS = B >= kBlock48k;
    static constexpr uint32_t yring_floats = YLDS ? 3u * (uint32_t)B * kWave : 0u;
    // DCW builds (a fifth wavefront runs the DC blocker a block ahead, see DcStage): its outputs of four
    // blocks -- the one being written, the one stage 1 consumes, two more for a replay
    static constexpr uint32_t dcw_blocks = 4u;
    static constexpr uint32_t dcw_ring_floats = dcw_blocks * (uint32_t)B * kWave;
    // (the FASTMATH filters take the taps in chunks of kRelaxChunk: 92 taps are followed by 6 zero taps, which meet
    // window slots further back -- finite values, the ring holds 5 blocks)
    static constexpr int NTP = (NT + kRelaxChunk - 1) / kRelaxChunk * kRelaxChunk;
    // 44.1 / 48 kHz: the centred tap table of the FASTMATH build's filters (demod_pair_centred: NT / 2 entries of Re mark, Re space,
    // Im mark, Im space) behind the taps
    static constexpr uint32_t ctap_off = (uint32_t)(NTP * 4 + PIPE_PROF_TAP_PAD);
    static constexpr uint32_t ctap_floats = NT == 42 ? 0u : (uint32_t)(NT / 2 * 4);
    static constexpr uint32_t tap_floats = (ctap_off + ctap_floats + 63u) / 64u * 64u;
    static_assert(ctap_off % 4u == 0u, "16-byte entries");
    static_assert(B <= RING - NTP + 1, "the first block's low copy would be read");
};

// =====================================================================================
// Four stages, one wavefront each: sample phase (block s) | matched filters + timing loop (block
// s-1) | symbol path (block s-2) | link events and transport wake-ups (block s-3).  The last one
// has no feedback into the others: it takes the global-memory round trips (event-log slots, the
// deadline ring in HBM) off the stage that is critical.
//
// Stage 2 hands every completed symbol (its two soft samples, the timing error and the sample
// index) to stage 3 through an LDS mailbox.  Stage 3's feedback is rare (about three symbols
// per burst) and goes both ways back: the AGC lock to stage 1, and to stage 2 the loop
// bandwidth (locked on sync, receiver.rs:432) and the timing-loop reset of end()
// (receiver.rs:479-490 -> rx/symsync.rs:166-170, 265-271).  Both earlier stages run ahead on
// their belief.  When a symbol of block b changes any of it, stage 1 replays the lane's AGC
// from the sample after that instant through blocks b+1 and b+2, and stage 2 -- which keeps a
// copy of its lane state from before block b+1 -- goes back to it, applies the change and
// processes block b+1 again over the corrected window, replacing what it had handed on.
// Every channel therefore sees exactly the sequential order of operations.
// =====================================================================================
constexpr uint32_t kP3SymWords = 5u * kWave;              // per parity: header, zero, sym, terr, until
constexpr uint32_t kP3FbWords = kWave + 32u;              // per parity: one word per lane + the any-flag
constexpr uint32_t kP3IoWords = 3u * kWave;               // per parity: symbol word, burst-pool slot, burst length
constexpr uint32_t kP3MailWords = 2u * kP3SymWords + 2u * kP3FbWords + 2u * kP3IoWords +
                                  2u * kWave +             // + the final TED-phase and wake-up flag bits
                                  4u * kWave +             // + SPLIT: instant positions [2][64], space and mark magnitudes [64] each
                                  kIoRingWords;            // + stage 4's deadline ring and its count (IoCtxLds)
// (the log-chunk words and SPLIT's sequence word live in the padding of the first feedback box)

// One of the two matched filters (WHICH 0: mark, 1: space) over the mirrored window: the half of
// demod_fast that one wavefront of the split stage 2 computes.  Same products, same order of
// accumulation, same hypot -- the two magnitudes are subtracted and clamped by the caller.
template <int NT, int RING, int WHICH>
__device__ __forceinline__ float demod_half(const float4 *tlds, const float *wring, uint32_t lane, uint32_t newest)
{
    constexpr int CH = 14;
    float2v acc = {0.0f, 0.0f};
    const float *wm = wring + ((int)newest + RING - (CH - 1)) * (int)kWave + (int)lane;
    const float2v *t2 = reinterpret_cast<const float2v *>(tlds) + WHICH;      // tap i: t2[2 * i]
#pragma unroll 1
    for (int base = 0; base + CH <= NT; base += CH) {
        float w[CH];
        float2v h[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) w[j] = wm[(CH - 1 - j) * (int)kWave];
        wm -= CH * (int)kWave;
#pragma unroll
        for (int j = 0; j < CH; ++j) h[j] = t2[2 * (base + j)];
#pragma unroll
        for (int j = 0; j < CH; ++j) { const float2v x2 = {w[j], w[j]}; const float2v pr = x2 * h[j]; acc += pr; }
    }
    constexpr int REM = NT % CH;
    if (REM) {
        float w[REM ? REM : 1];
#pragma unroll
        for (int j = 0; j < REM; ++j) w[j] = wm[(CH - 1 - j) * (int)kWave];
#pragma unroll
        for (int j = 0; j < REM; ++j) {
            const float2v t = t2[2 * (NT - REM + j)];
            const float2v x2 = {w[j], w[j]};
            const float2v pr = x2 * t;
            acc += pr;
        }
    }
    return rs_hypot(acc.x, acc.y);
}

// Both matched filters in one wavefront, their accumulation chains interleaved (the dependent packed adds of one
// hide under the other's): the two magnitudes separately.
template <int NT, int RING>
__device__ __forceinline__ void demod_pair(const float4 *tlds, const float *wring, uint32_t lane, uint32_t newest, float *hm_out, float *hs_out)
{
    constexpr int CH = 14;
    float2v am = {0.0f, 0.0f}, as = {0.0f, 0.0f};
    const float *wm = wring + ((int)newest + RING - (CH - 1)) * (int)kWave + (int)lane;
#pragma unroll 1
    for (int base = 0; base + CH <= NT; base += CH) {
        float w[CH];
        float4 h[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) w[j] = wm[(CH - 1 - j) * (int)kWave];
        wm -= CH * (int)kWave;
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
        for (int j = 0; j < REM; ++j) w[j] = wm[(CH - 1 - j) * (int)kWave];
#pragma unroll
        for (int j = 0; j < REM; ++j) {
            const float4 t = tlds[NT - REM + j];
            const float2v x2 = {w[j], w[j]};
            const float2v hm = {t.x, t.y}, hs = {t.z, t.w};
            const float2v pm = x2 * hm, ps = x2 * hs;
            am += pm; as += ps;
        }
    }
    *hm_out = rs_hypot(am.x, am.y);
    *hs_out = rs_hypot(as.x, as.y);
}

// The same with the filter chosen per lane (which 0: mark, 1: space): narrow workgroups (16 or 32 channels)
// leave half of the helper wavefront's lanes idle, so lanes 0 .. LANES-1 take the mark filter of channel
// `ch` and lanes LANES .. 2*LANES-1 its space filter -- both filters in the time of one.  The taps are then
// read with two addresses per wavefront instead of one (still conflict-free: adjacent 8-byte words).
template <int NT, int RING>
__device__ __forceinline__ float demod_half_dyn(const float4 *tlds, const float *wring, uint32_t ch, uint32_t newest, uint32_t which)
{
    constexpr int CH = 14;
    float2v acc = {0.0f, 0.0f};
    const float *wm = wring + ((int)newest + RING - (CH - 1)) * (int)kWave + (int)ch;
    const float2v *t2 = reinterpret_cast<const float2v *>(tlds) + which;      // tap i: t2[2 * i]
#pragma unroll 1
    for (int base = 0; base + CH <= NT; base += CH) {
        float w[CH];
        float2v h[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) w[j] = wm[(CH - 1 - j) * (int)kWave];
        wm -= CH * (int)kWave;
#pragma unroll
        for (int j = 0; j < CH; ++j) h[j] = t2[2 * (base + j)];
#pragma unroll
        for (int j = 0; j < CH; ++j) { const float2v x2 = {w[j], w[j]}; const float2v pr = x2 * h[j]; acc += pr; }
    }
    constexpr int REM = NT % CH;
    if (REM) {
        float w[REM ? REM : 1];
#pragma unroll
        for (int j = 0; j < REM; ++j) w[j] = wm[(CH - 1 - j) * (int)kWave];
#pragma unroll
        for (int j = 0; j < REM; ++j) {
            const float2v t = t2[2 * (NT - REM + j)];
            const float2v x2 = {w[j], w[j]};
            const float2v pr = x2 * t;
            acc += pr;
        }
    }
    return rs_hypot(acc.x, acc.y);
}

// Stage 1 state: DC blocker, AGC, input prefetch, and what a replay needs of its last three blocks
// FM: relaxed AGC step (same_relaxed_common.h) -- the FASTMATH build of time-parallel launches
// CMODE: 1 = the input is channel-major with per-lane streams (xl), 0 = it is not, 2 = decided at run time.  The two
// compile-time forms exist because a conditional load anywhere between two reads of the prefetch registers makes the
// compiler wait for all outstanding loads at the reads (see fetch).
template <int NT_, bool MED3, typename SampleT, bool FM = false, int CMODE = 2>
struct SampleStage {
    static constexpr int NT = NT_, DCL = PipeGeom<NT_>::DCL, kB = PipeLayout<NT_>::B, RING = PipeLayout<NT_>::RING;
    static constexpr uint32_t LP = kWave;
    static constexpr bool YLDS = PipeLayout<NT_>::YLDS;
    float *ycol;                         // YLDS: this lane's column of the ring [3][kB][64]
    uint32_t ycur;                       // YLDS: ring block of the newest block (slot j back: (ycur + 3 - j) % 3)
    float sum0, sum1, gain;
    bool locked;                         // this stage's belief of the AGC lock
    float xp[DCL], mp[DCL];              // the last DCL inputs / first-stage averages, oldest first
    float xn[2][kB];                     // prefetched inputs of the next two blocks: block b waits in xn[b & 1]
                                         // (a step is shorter than an HBM round trip with a TLB miss, so
                                         // the prefetch distance is two steps)
    float ys[YLDS ? 1 : 3][kB];          // DC-blocker outputs: [0] newest block ... [2] two blocks back
    float g0[3];                         // AGC gain each of them started with
    uint32_t wp[3];                      // their ring positions
    uint32_t wnext;                      // ring position of the block computed next

    // c / C index the state arrays, cin / Cin the input (they differ for time-parallel chunks, where
    // several state columns read the same input column at different rows)
    // xl != nullptr: channel-major input, this lane's own contiguous stream (16-byte loads: four lanes' worth of a
    // 64-byte sector per step and lane, the rest of the sector in the next three); avail = blocks it may read
    const SampleT *xl = nullptr;
    uint32_t avail = 0;
    __device__ __forceinline__ void load_block_cm(float *dst, uint32_t blk) const
    {
        if constexpr (std::is_same<SampleT, float>::value) {
            const float4 *p4 = reinterpret_cast<const float4 *>(xl + (size_t)blk * kB);
#pragma unroll
            for (int j = 0; j < kB / 4; ++j) { const float4 v = p4[j]; dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w; }
        }
    }
    __device__ __forceinline__ void load(const Params &P, const State &S, const SampleT *__restrict__ x,
                                         uint32_t c, uint32_t C, uint32_t cin, uint32_t Cin, uint64_t counter0,
                                         float *wcol, uint32_t n_blocks)
    {
        const uint32_t G = P.win_ring;
#pragma unroll 2
        for (uint32_t m = 1; m <= G; ++m) {
            const uint32_t g = (uint32_t)(counter0 - (uint64_t)m) & (G - 1u);
            const float *row = S.win_ring + (size_t)g * C;
            const float v = row[c];
            if ((uint32_t)RING - m >= (uint32_t)kB) wcol[((uint32_t)RING - m) * LP] = v;
            wcol[(2u * (uint32_t)RING - m) * LP] = v;
        }
        sum0 = S.dc_sum0[c]; sum1 = S.dc_sum1[c]; gain = S.agc_gain[c];
        locked = (S.flags[c] & F_AGC_LOCKED) != 0u;
        const uint32_t dpos = (uint32_t)(counter0 % (uint64_t)DCL);
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            uint32_t slot = dpos + (uint32_t)k;
            if (slot >= (uint32_t)DCL) slot -= (uint32_t)DCL;
            const float *r0 = S.dc_ff_ring + (size_t)slot * C, *r1 = S.dc_fb_ring + (size_t)slot * C;
            xp[k] = r0[c];
            mp[k] = r1[c];
        }
        if (CMODE == 1 || (CMODE == 2 && xl)) {
#pragma unroll
            for (int k = 0; k < kB; ++k) { xn[0][k] = 0.0f; xn[1][k] = 0.0f; }
            if (avail > 0u) load_block_cm(xn[0], 0u);
            if (avail > 1u) load_block_cm(xn[1], 1u);
        } else {
#pragma unroll
        for (int k = 0; k < kB; ++k) { const SampleT *row = x + (size_t)k * Cin; xn[0][k] = (float)row[cin]; }
        if (n_blocks > 1u) {
            const SampleT *xb = x + (size_t)kB * Cin;
#pragma unroll
            for (int k = 0; k < kB; ++k) { const SampleT *row = xb + (size_t)k * Cin; xn[1][k] = (float)row[cin]; }
        }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            g0[j] = gain; wp[j] = 0;
            if (!YLDS || j == 0) {
#pragma unroll
                for (int k = 0; k < kB; ++k) ys[YLDS ? 0 : j][k] = 0.0f;
            }
        }
        ycur = 0;
        wnext = 0;
    }

    __device__ __forceinline__ void rotate()
    {
        if constexpr (YLDS) {
            ycur = ycur == 2u ? 0u : ycur + 1u;          // every step, like the register rotation
        } else {
#pragma unroll
            for (int k = 0; k < kB; ++k) { ys[2][k] = ys[1][k]; ys[1][k] = ys[0][k]; }
        }
        g0[2] = g0[1]; g0[1] = g0[0];
        wp[2] = wp[1]; wp[1] = wp[0];
    }

    __device__ __forceinline__ void push_block(const Params &P, float *wcol, int j, float &g, int fk, float bw0, float bw1)
    {
        if constexpr (YLDS) {
            if (j != 0) {                  // (only a replay gets here: the block's outputs come back from the ring)
                const uint32_t yb = ycur + 3u - (uint32_t)j;
                const float *y = ycol + ((yb >= 3u ? yb - 3u : yb) * (uint32_t)kB) * LP;
                float yv[kB];
#pragma unroll
                for (int k = 0; k < kB; ++k) yv[k] = y[k * LP];
                float *wblk = wcol + wp[j] * LP;
                float *wlow = wcol + (wp[j] == 0u ? (uint32_t)RING : wp[j]) * LP;
#pragma unroll
                for (int k = 0; k < kB; ++k) {
                    const float out = FM ? agc_step_relaxed(P, yv[k], g, (k <= fk) ? bw0 : bw1) : agc_step<MED3>(P, yv[k], g, (k <= fk) ? bw0 : bw1);
                    wlow[k * LP] = out;
                    wblk[(k + RING) * LP] = out;
                }
                return;
            }
        }
        const int jj = YLDS ? 0 : j;
        // AGC (rx/agc.rs:72-77) and window push (receiver.rs:345-346) of the block in slot j:
        // bandwidth bw0 up to sample fk, bw1 after it
        float *wblk = wcol + wp[j] * LP;
        float *wlow = wcol + (wp[j] == 0u ? (uint32_t)RING : wp[j]) * LP;
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            const float out = FM ? agc_step_relaxed(P, ys[jj][k], g, (k <= fk) ? bw0 : bw1) : agc_step<MED3>(P, ys[jj][k], g, (k <= fk) ? bw0 : bw1);
            wlow[k * LP] = out;
            wblk[(k + RING) * LP] = out;
        }
    }

    float xs[kB];                        // inputs of the block being computed
    // take the prefetched inputs of block `blk` (BUF = blk & 1) and start fetching block blk + 2 into
    // the registers they leave
    template <int BUF>
    __device__ __forceinline__ void fetch(const SampleT *__restrict__ x, uint32_t blk, uint32_t n_blocks,
                                          uint32_t cin, uint32_t Cin)
    {
        if constexpr (CMODE == 1) {
            // per lane: its stream ends where the input does, and silence follows it (a lane that runs on with its workgroup
            // past the end of its stream must not chew on stale samples); the load itself is unconditional, see below
            const bool live = blk < avail;
#pragma unroll
            for (int k = 0; k < kB; ++k) xs[k] = live ? xn[BUF][k] : 0.0f;
            load_block_cm(xn[BUF], min(blk + 2u, avail - 1u));      // (avail >= 1: the planner leaves two scout blocks behind every cut)
            return;
        }
#pragma unroll
        for (int k = 0; k < kB; ++k) xs[k] = xn[BUF][k];
        if (CMODE == 1 || (CMODE == 2 && xl)) {
            // (this form stays conditional: the unconditional one measured the same here, 3.6 against 3.55 ms, and one build
            // of it ran the sorted time-parallel launch three times slower)
            if (blk + 2 < avail) {
                load_block_cm(xn[BUF], blk + 2u);       // per lane: its stream ends where the input does
            } else {
                // ... and silence follows it: a lane that runs on with its workgroup past the end of its stream must not
                // chew on stale registers (its events from there on are never kept, but its state may be)
#pragma unroll
                for (int k = 0; k < kB; ++k) xn[BUF][k] = 0.0f;
            }
        } else {
            // Issued unconditionally (past the end: the last block once more): a load behind a condition makes the compiler
            // wait for ALL outstanding loads (`s_waitcnt vmcnt(0)`) where the registers are read, i.e. also for the block
            // fetched one step ago -- which halves the prefetch distance and left this stage waiting ~1 400 clk per step for
            // memory (SAME_P1_SPLIT profile, round 3: 32 768 ch x 2 s 3.78 -> 3.58 ms relaxed; configs[1] strict 12.1 -> 11.4 ms).
            // It only pays where the other form is compiled out (CMODE 0): with both in one kernel the wait stays as it was.
            const SampleT *xb = x + ((size_t)min(blk + 2u, n_blocks - 1u) * kB) * Cin;      // wave-uniform
#pragma unroll
            for (int k = 0; k < kB; ++k) { const SampleT *row = xb + (size_t)k * Cin; xn[BUF][k] = (float)row[cin]; }
        }
    }
    // DC blocker (rx/dcblock.rs:45-49, 104-108), AGC and window push of the fetched block into slot 0
    __device__ __forceinline__ void block(const Params &P, float *wcol)
    {
        float mnew[kB];
        auto xw = [&](int i) { return i < DCL ? xp[i < DCL ? i : 0] : xs[i >= DCL ? i - DCL : 0]; };
#pragma unroll
        for (int k = 0; k < kB; k += 2) {
            const float2v x2 = {xs[k], xs[k + 1]}, xo = {xw(k), xw(k + 1)};
            const float2v d0 = x2 - xo;
            const float s0a = sum0 + d0.x, s0b = s0a + d0.y;
            sum0 = s0b;
            const float2v s0 = {s0a, s0b}, inv = {P.dc_inv_len, P.dc_inv_len};
            const float2v ma0 = s0 * inv;
            const float2v sig = {xw(k + 1), xw(k + 2)};
            const float2v mo = {k < DCL ? mp[k < DCL ? k : 0] : mnew[k >= DCL ? k - DCL : 0],
                                k + 1 < DCL ? mp[k + 1 < DCL ? k + 1 : 0] : mnew[k + 1 >= DCL ? k + 1 - DCL : 0]};
            const float2v d1 = ma0 - mo;
            const float s1a = sum1 + d1.x, s1b = s1a + d1.y;
            sum1 = s1b;
            const float2v s1 = {s1a, s1b};
            const float2v ma1 = s1 * inv;
            const float2v y2 = sig - ma1;
            ys[0][k] = y2.x; ys[0][k + 1] = y2.y;
            mnew[k] = ma0.x; mnew[k + 1] = ma0.y;
        }
        // the windows move on by one block (ascending: entry k only takes from entries above it)
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            xp[k] = kB + k < DCL ? xp[kB + k < DCL ? kB + k : 0] : xs[kB + k >= DCL ? kB + k - DCL : 0];
            mp[k] = kB + k < DCL ? mp[kB + k < DCL ? kB + k : 0] : mnew[kB + k >= DCL ? kB + k - DCL : 0];
        }
        if constexpr (YLDS) {
            float *y = ycol + (ycur * (uint32_t)kB) * LP;
#pragma unroll
            for (int k = 0; k < kB; ++k) y[k * LP] = ys[0][k];
        }
        g0[0] = gain;
        wp[0] = wnext;
        const float bw = locked ? 0.0f : P.agc_bw;
        push_block(P, wcol, 0, gain, kB, bw, bw);
        wnext += kB;
        if (wnext == (uint32_t)RING) wnext = 0;
    }

    // the lock flipped at sample fk of the block in slot 2: redo the AGC from there on
    __device__ __forceinline__ void replay(const Params &P, float *wcol, int fk, bool new_locked, bool valid1, bool valid0)
    {
        const float bw0 = locked ? 0.0f : P.agc_bw;
        locked = new_locked;
        const float bw1 = locked ? 0.0f : P.agc_bw;
        float g = g0[2];
        push_block(P, wcol, 2, g, fk, bw0, bw1);
        if (valid1) { g0[1] = g; push_block(P, wcol, 1, g, -1, bw1, bw1); }
        if (valid0) { g0[0] = g; push_block(P, wcol, 0, g, -1, bw1, bw1); }
        gain = g;
    }

    __device__ __forceinline__ void store(const Params &P, const State &S, uint32_t c, uint32_t C, uint64_t counter1,
                                          const float *wcol)
    {
        S.dc_sum0[c] = sum0; S.dc_sum1[c] = sum1; S.agc_gain[c] = gain;
        const uint32_t dpos = (uint32_t)(counter1 % (uint64_t)DCL);
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            uint32_t slot = dpos + (uint32_t)k;
            if (slot >= (uint32_t)DCL) slot -= (uint32_t)DCL;
            float *r0 = S.dc_ff_ring + (size_t)slot * C, *r1 = S.dc_fb_ring + (size_t)slot * C;
            r0[c] = xp[k];
            r1[c] = mp[k];
        }
        const uint32_t G = P.win_ring;
#pragma unroll 2
        for (uint32_t m = 1; m <= G; ++m) {
            const uint32_t g = (uint32_t)(counter1 - (uint64_t)m) & (G - 1u);
            const uint32_t j = wnext >= m ? wnext - m : wnext + (uint32_t)RING - m;
            float *row = S.win_ring + (size_t)g * C;
            row[c] = wcol[(j + (uint32_t)RING) * LP];           // the high copy is always there
        }
    }
};

// ---- stage 1 cut in two (DCW builds: 16- and 32-channel workgroups at 22.05 kHz, 64-channel ones at 44.1 / 48 kHz) ----
// The DC blocker takes no feedback from anything downstream (rx/dcblock.rs:45-49), so a fifth wavefront runs
// it one block AHEAD of the AGC and hands its outputs over through a four-block LDS ring; what is left of
// stage 1 -- AGC, window push, the replay of an AGC lock flip -- reads them from there.  Same operations in
// the same order per channel; only the wavefront that executes them changes.
template <int NT_, typename SampleT>
struct DcStage {
    static constexpr int DCL = PipeGeom<NT_>::DCL, kB = PipeLayout<NT_>::B;
    static constexpr uint32_t LP = kWave;
    float sum0, sum1;
    float xp[DCL], mp[DCL];              // the last DCL inputs / first-stage averages, oldest first
    float xn[2][kB];                     // prefetched inputs of the next two blocks: block b waits in xn[b & 1]
    float xs[kB];
    __device__ __forceinline__ void load(const Params &P, const State &S, const SampleT *__restrict__ x, uint32_t c, uint32_t C,
                                         uint32_t cin, uint32_t Cin, uint64_t counter0, uint32_t n_blocks)
    {
        sum0 = S.dc_sum0[c]; sum1 = S.dc_sum1[c];
        const uint32_t dpos = (uint32_t)(counter0 % (uint64_t)DCL);
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            uint32_t slot = dpos + (uint32_t)k;
            if (slot >= (uint32_t)DCL) slot -= (uint32_t)DCL;
            const float *r0 = S.dc_ff_ring + (size_t)slot * C, *r1 = S.dc_fb_ring + (size_t)slot * C;
            xp[k] = r0[c];
            mp[k] = r1[c];
        }
#pragma unroll
        for (int k = 0; k < kB; ++k) { const SampleT *row = x + (size_t)k * Cin; xn[0][k] = (float)row[cin]; }
        if (n_blocks > 1u) {
            const SampleT *xb = x + (size_t)kB * Cin;
#pragma unroll
            for (int k = 0; k < kB; ++k) { const SampleT *row = xb + (size_t)k * Cin; xn[1][k] = (float)row[cin]; }
        }
    }
    template <int BUF>
    __device__ __forceinline__ void fetch(const SampleT *__restrict__ x, uint32_t blk, uint32_t n_blocks, uint32_t cin, uint32_t Cin)
    {
#pragma unroll
        for (int k = 0; k < kB; ++k) xs[k] = xn[BUF][k];
        // (unconditional, past the end the last block once more: see SampleStage::fetch)
        const SampleT *xb = x + ((size_t)min(blk + 2u, n_blocks - 1u) * kB) * Cin;      // wave-uniform
#pragma unroll
        for (int k = 0; k < kB; ++k) { const SampleT *row = xb + (size_t)k * Cin; xn[BUF][k] = (float)row[cin]; }
    }
    // DC blocker (rx/dcblock.rs:45-49, 104-108) of the fetched block; outputs to y[k * LP]
    __device__ __forceinline__ void block(const Params &P, float *y)
    {
        float mnew[kB];
        auto xw = [&](int i) { return i < DCL ? xp[i < DCL ? i : 0] : xs[i >= DCL ? i - DCL : 0]; };
#pragma unroll
        for (int k = 0; k < kB; k += 2) {
            const float2v x2 = {xs[k], xs[k + 1]}, xo = {xw(k), xw(k + 1)};
            const float2v d0 = x2 - xo;
            const float s0a = sum0 + d0.x, s0b = s0a + d0.y;
            sum0 = s0b;
            const float2v s0 = {s0a, s0b}, inv = {P.dc_inv_len, P.dc_inv_len};
            const float2v ma0 = s0 * inv;
            const float2v sig = {xw(k + 1), xw(k + 2)};
            const float2v mo = {k < DCL ? mp[k < DCL ? k : 0] : mnew[k >= DCL ? k - DCL : 0],
                                k + 1 < DCL ? mp[k + 1 < DCL ? k + 1 : 0] : mnew[k + 1 >= DCL ? k + 1 - DCL : 0]};
            const float2v d1 = ma0 - mo;
            const float s1a = sum1 + d1.x, s1b = s1a + d1.y;
            sum1 = s1b;
            const float2v s1 = {s1a, s1b};
            const float2v ma1 = s1 * inv;
            const float2v y2 = sig - ma1;
            y[k * LP] = y2.x; y[(k + 1) * LP] = y2.y;
            mnew[k] = ma0.x; mnew[k + 1] = ma0.y;
        }
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            xp[k] = kB + k < DCL ? xp[kB + k < DCL ? kB + k : 0] : xs[kB + k >= DCL ? kB + k - DCL : 0];
            mp[k] = kB + k < DCL ? mp[kB + k < DCL ? kB + k : 0] : mnew[kB + k >= DCL ? kB + k - DCL : 0];
        }
    }
    __device__ __forceinline__ void store(const State &S, uint32_t c, uint32_t C, uint64_t counter1)
    {
        S.dc_sum0[c] = sum0; S.dc_sum1[c] = sum1;
        const uint32_t dpos = (uint32_t)(counter1 % (uint64_t)DCL);
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            uint32_t slot = dpos + (uint32_t)k;
            if (slot >= (uint32_t)DCL) slot -= (uint32_t)DCL;
            float *r0 = S.dc_ff_ring + (size_t)slot * C, *r1 = S.dc_fb_ring + (size_t)slot * C;
            r0[c] = xp[k];
            r1[c] = mp[k];
        }
    }
};

template <int NT_, bool MED3, bool FM = false>
struct AgcStage {
    static constexpr int kB = PipeLayout<NT_>::B, RING = PipeLayout<NT_>::RING;
    static constexpr int DEPTH = 3;                     // blocks a replay reaches back over, the current one included
    static constexpr uint32_t LP = kWave, YMASK = PipeLayout<NT_>::dcw_blocks - 1u;
    const float *ycol;                   // this lane's column of the DC wave's ring [4 or 8][kB][64]; block b in slot b & YMASK
    float gain;
    bool locked;                         // this stage's belief of the AGC lock
    float g0[DEPTH];                     // AGC gain the last blocks started with ([0] newest)
    uint32_t wp[DEPTH];                  // their window-ring positions
    uint32_t wnext;
    __device__ __forceinline__ void load(const Params &P, const State &S, uint32_t c, uint32_t C, uint64_t counter0, float *wcol)
    {
        const uint32_t G = P.win_ring;
#pragma unroll 2
        for (uint32_t m = 1; m <= G; ++m) {
            const uint32_t g = (uint32_t)(counter0 - (uint64_t)m) & (G - 1u);
            const float *row = S.win_ring + (size_t)g * C;
            const float v = row[c];
            if ((uint32_t)RING - m >= (uint32_t)kB) wcol[((uint32_t)RING - m) * LP] = v;
            wcol[(2u * (uint32_t)RING - m) * LP] = v;
        }
        gain = S.agc_gain[c];
        locked = (S.flags[c] & F_AGC_LOCKED) != 0u;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) { g0[j] = gain; wp[j] = 0; }
        wnext = 0;
    }
    __device__ __forceinline__ void rotate()
    {
#pragma unroll
        for (int j = DEPTH - 1; j > 0; --j) { g0[j] = g0[j - 1]; wp[j] = wp[j - 1]; }
    }
    // AGC (rx/agc.rs:72-77) and window push (receiver.rs:345-346) of block `blk` (history slot j): bandwidth bw0
    // up to sample fk, bw1 after it
    __device__ __forceinline__ void push(const Params &P, float *wcol, int j, uint32_t blk, float &g, int fk, float bw0, float bw1)
    {
        const float *y = ycol + ((blk & YMASK) * (uint32_t)kB) * LP;
        float yv[kB];
#pragma unroll
        for (int k = 0; k < kB; ++k) yv[k] = y[k * LP];
        float *wblk = wcol + wp[j] * LP;
        float *wlow = wcol + (wp[j] == 0u ? (uint32_t)RING : wp[j]) * LP;
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            const float out = FM ? agc_step_relaxed(P, yv[k], g, (k <= fk) ? bw0 : bw1) : agc_step<MED3>(P, yv[k], g, (k <= fk) ? bw0 : bw1);
            wlow[k * LP] = out;
            wblk[(k + RING) * LP] = out;
        }
    }
    __device__ __forceinline__ void block(const Params &P, float *wcol, uint32_t blk)
    {
        g0[0] = gain;
        wp[0] = wnext;
        const float bw = locked ? 0.0f : P.agc_bw;
        push(P, wcol, 0, blk, gain, kB, bw, bw);
        wnext += kB;
        if (wnext == (uint32_t)RING) wnext = 0;
    }
    // the lock flipped at sample fk of block s - 2 (history slot 2): redo
    // the AGC from there on, through every later block that exists (blocks < n_blocks)
    __device__ __forceinline__ void replay(const Params &P, float *wcol, uint32_t s, int fk, bool new_locked, uint32_t n_blocks)
    {
        const float bw0 = locked ? 0.0f : P.agc_bw;
        locked = new_locked;
        const float bw1 = locked ? 0.0f : P.agc_bw;
        float g = g0[DEPTH - 1];
        push(P, wcol, DEPTH - 1, s - 2u, g, fk, bw0, bw1);
#pragma unroll
        for (int j = DEPTH - 2; j >= 0; --j) {
            const uint32_t blk = s - 2u + (uint32_t)(DEPTH - 1 - j);
            if (blk < n_blocks) { g0[j] = g; push(P, wcol, j, blk, g, -1, bw1, bw1); }
        }
        gain = g;
    }
    __device__ __forceinline__ void store(const Params &P, const State &S, uint32_t c, uint32_t C, uint64_t counter1, const float *wcol)
    {
        S.agc_gain[c] = gain;
        const uint32_t G = P.win_ring;
#pragma unroll 2
        for (uint32_t m = 1; m <= G; ++m) {
            const uint32_t g = (uint32_t)(counter1 - (uint64_t)m) & (G - 1u);
            const uint32_t j = wnext >= m ? wnext - m : wnext + (uint32_t)RING - m;
            float *row = S.win_ring + (size_t)g * C;
            row[c] = wcol[(j + (uint32_t)RING) * LP];           // the high copy is always there
        }
    }
};

// wavefronts per workgroup: five in DCW builds.  44.1 / 48 kHz (one workgroup per CU, 32-sample blocks): stage 1 was one of
// the two long stages there; with the DC blocker on a wavefront of its own 16 384 channels x 2 s run 6.41 -> 5.80 ms at
// 48 kHz and 5.49 -> 4.86 ms at 44.1 kHz (round 1: 6.10 / 5.14; the 6 % in between came in with the time-parallel
// bookkeeping of every stage's step loop, commit 145fa23, and a build without it did not win it back).
template <int NT, int LANES, bool SPLIT> constexpr bool pipe_dcw() { return SPLIT && ((NT == 42 && LANES <= 32) || (NT != 42 && LANES == 64)); }
// SHARE: built for two wavefronts per SIMD (half of the 512-entry register file each, a few
// spills) -- what lets two workgroups, eight wavefronts, share a CU's four SIMDs beyond 16 384
// channels.  Smaller batches use the unconstrained build (2 % faster).
//
// LANES: channels per workgroup.  A wavefront's time per step does not depend on how many of its
// lanes are live, but it does depend on which paths ANY live lane takes: with 64 channels nearly
// every step has a lane with a byte due, a lane whose symbol changes the AGC lock, ...; with 16
// most steps skip those sections.  Small batches (which leave CUs idle anyway) therefore spread
// over more workgroups of fewer channels; lanes >= LANES retire at once.
//
// SPLIT: stage 4's wavefront, the lightest, takes the space filter off stage 2 (which keeps the
// mark filter and the timing loop).  Both filters need only the position of the block's instant,
// known a block ahead; stage 2 posts it, stage 4 computes the space magnitude first thing in its
// step, posts it and bumps a sequence word that stage 2 polls before it combines the two -- a
// hand-over inside the step, no extra block of latency.
// FM (FASTMATH): the relaxed arithmetic of same_relaxed_common.h in every stage -- matched filters as fused multiply-adds
// into partial sums with an f32 square root, the AGC's two-operation gain chain, the equalizer's fused steps.  Built for
// the 64-channel two-per-CU form the time-parallel launches use; its parity contract is that mode's (include/same_rx.h).
template <int NT, int NFF, int NFB, bool MED3, bool SHARE, int LANES, bool SPLIT, typename SampleT, bool FM = false, int CMODE = 2>
__global__ __launch_bounds__((pipe_dcw<NT, LANES, SPLIT>() ? 5 : 4) * kWave, SHARE ? 2 : 1) void demod_pipe_kernel(Params P, State S, Output O,
                                                                const float4 *__restrict__ taps,
                                                                const SampleT *__restrict__ x,
                                                                uint32_t n_blocks, uint64_t counter0, PipeChunks K)
{
    static_assert(!FM || (SPLIT && LANES == 64 && SHARE == (NT == 42)),
                  "FASTMATH is built for 64-channel workgroups: two per CU at 22.05 kHz, one (with the DC wavefront) at 44.1 / 48 kHz");
    constexpr int NTP = PipeLayout<NT>::NTP;
    if constexpr (CMODE == 0) { K.col_row0 = nullptr; K.col_perm = nullptr; }       // (the host launches this build for nothing else)
    constexpr int kB = PipeLayout<NT>::B, RING = PipeLayout<NT>::RING;
    constexpr uint32_t LP = kWave, kPipeTapFloats = PipeLayout<NT>::tap_floats;
    // PACKED: workgroups of 16 or 32 channels -- the helper wavefront computes BOTH matched filters (mark on its
    // lanes 0 .. LANES-1, space on LANES .. 2*LANES-1) and stage 2 keeps only the timing loop
    constexpr bool PACKED = SPLIT && LANES <= 32;
    // HELPER_BOTH: stage 2 computes no filter of its own.  Besides the PACKED workgroups: the two-per-CU build at
    // 22.05 kHz, where the helper runs both filters in every lane with their accumulation chains interleaved
    // (demod_pair) -- 32 768 channels x 2 s 4.34 -> 4.21 ms, the time-parallel launch of configs[1] 4.49 -> 4.39 ms
    // on one box; at 48 kHz (92 taps, one workgroup per CU) the same is 4 % slower, so not there.
    // (FASTMATH: both filters share their window loads, demod_pair_relaxed -- always the helper's)
    constexpr bool HELPER_BOTH = PACKED || (SPLIT && SHARE && NT == 42) || FM;
    // DCW: a fifth wavefront runs the DC blocker one block ahead (DcStage / AgcStage above)
    constexpr bool DCW = pipe_dcw<NT, LANES, SPLIT>();
    static_assert(kB <= 64, "the sample index travels in six bits of the stage 3 -> 4 word");
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x & (kWave - 1u);
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0..3 = stage 1..4, 4 = DC wave
    const uint32_t C = P.n_channels;
    // state column of this lane: its grid position, or -- per-column geometry with pieces sorted by length -- what the
    // planner's permutation puts there
    const uint32_t c = (K.n_chunks > 1u && K.col_perm) ? K.col_perm[blockIdx.x * (uint32_t)LANES + lane]
                                                       : blockIdx.x * (uint32_t)LANES + lane;  // C % LANES == 0 (host)
    // Time-parallel chunks: state column c = chunk * Cin + cin reads input column cin
    // from the chunk's first row on; a workgroup never straddles chunks (Cin % LANES == 0, host).
    uint32_t cin = c, Cin = C, n_nominal = n_blocks;
    bool may_leave = false;                                  // chunk that hands over: leave once every lane is idle
    int32_t row_l = 0;                                       // per-column geometry: this lane's first sample, relative to the workgroup's first lane's
    const SampleT *xl = nullptr;                             // ... and its own stream in a channel-major input
    uint32_t avail_l = 0;
    if (K.n_chunks > 1u) {
        Cin = K.in_channels;
        if (K.col_row0) {
            const uint32_t chunk_l = c / Cin;                                // (per lane when the columns are permuted)
            cin = c - chunk_l * Cin;
            // workgroups are homogeneous: all chunk 0, all last chunk, or all in between (the planner sees to it)
            may_leave = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk_l) + 1u < K.n_chunks;
            const uint32_t row_abs = K.col_row0[c];
            xl = x + (size_t)cin * K.in_samples + row_abs;
            avail_l = (K.whole_samples - row_abs) / (uint32_t)kB;
            n_blocks = K.wg_blocks[blockIdx.x];
            n_nominal = may_leave ? K.col_nominal[c] : n_blocks;      // (per lane)
            // counter0 stays wave-uniform: the first lane's row (state is loaded / stored by chunk 0 and by the last
            // chunk only, whose lanes share their row); what a lane's row differs by goes into its event stamps
            const uint32_t row_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_abs);
            counter0 += (uint64_t)row_first;
            row_l = (int32_t)(row_abs - row_first);
        } else {
            const uint32_t wgs = K.in_channels / (uint32_t)LANES;
            const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x / wgs));
            cin = (blockIdx.x - chunk * wgs) * (uint32_t)LANES + lane;
            may_leave = chunk + 1u < K.n_chunks;
            const uint32_t first_block = chunk * K.stride_blocks;
            x += (size_t)first_block * kB * Cin;
            counter0 += (uint64_t)first_block * kB;
            n_blocks -= first_block;                             // everything up to the end of the input is available
            n_nominal = may_leave ? K.nominal_blocks : n_blocks;
        }
    }
    // LDS: taps | mailboxes | squelch history [64][64] | window (logical slots kB .. 2*RING-1) | stage 1's ring (YLDS)
    float4 *tlds = reinterpret_cast<float4 *>(lds);
    lds_u32 *mail = (lds_u32 *)(lds + kPipeTapFloats);
    lds_u32 *symbox = mail;                                    // [2][5][64]
    lds_u32 *fbbox = mail + 2u * kP3SymWords;                  // [2][64 + flag]
    lds_u32 *iobox = fbbox + 2u * kP3FbWords;                  // [2][3][64]
    lds_u32 *phasebox = iobox + 2u * kP3IoWords;               // [64] stage 2's final TED phase bit
    lds_u32 *againbox = phasebox + kWave;                      // [64] stage 4's final F_TICK_AGAIN bit
    lds_u32 *chunkbox = fbbox + kWave + 2u;                    // [2], in the first box's padding
    lds_u32 *seqbox = fbbox + kWave + 4u;                      // SPLIT: stage 4's progress with the space filter, 2 * step + pass
    lds_u32 *posbox = againbox + kWave;                        // SPLIT: [2][64] sample index of block b's instant
    lds_u32 *spacebox = posbox + 2u * kWave;                   // SPLIT: [64] space-filter magnitude
    lds_u32 *markbox = spacebox + kWave;                       // PACKED: [64] mark-filter magnitude
    lds_u32 *tkbox = markbox + kWave;                          // [kTickRing][64] u64 deadlines, then [64] their count (stage 4's own)
    float *hcol = lds + kPipeTapFloats + kP3MailWords + lane;
    float *wring = lds + kPipeTapFloats + kP3MailWords + (kSquelchHist - kB) * LP;   // logical slot 0
    float *wcol = wring + lane;
    const uint64_t counter1 = counter0 + (uint64_t)n_blocks * kB;
    const uint32_t n_steps = n_blocks + 3u;
    const uint32_t last_fb_step = n_blocks + 1u;                // stage 3 runs in steps 2 .. n_blocks + 1

    float *dcw_ring = wring + (2u * (uint32_t)RING) * LP;             // DCW: [4][kB][64] behind the window
    if (DCW && role == 4u) {
        // ------------------------------ DC wave: DC blocker of block s + 1 ------------------------
        if (LANES < (int)kWave && lane >= (uint32_t)LANES) return;
        P3_HWID(4);
        DcStage<NT, SampleT> D;
        D.load(P, S, x, c, C, cin, Cin, counter0, n_blocks);
        float *ycol = dcw_ring + lane;
        constexpr uint32_t YMASK = PipeLayout<NT>::dcw_blocks - 1u;
        D.template fetch<0>(x, 0u, n_blocks, cin, Cin);               // block 0 before anyone starts
        D.block(P, ycol);
        lds_barrier();                                                 // prologue (every role of a DCW build has one)
        uint32_t stop_at = 0xffffffffu;
        P3_T0();
        auto step = [&](uint32_t s, auto buf) -> bool {
            const uint32_t blk = s + 1u;                               // BUF = blk & 1
            if (blk < n_blocks && !PROF_SKIP(P, 128)) {
                D.template fetch<decltype(buf)::value>(x, blk, n_blocks, cin, Cin);
                D.block(P, ycol + ((blk & YMASK) * (uint32_t)kB) * LP);
            }
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (s >= 2u && s <= last_fb_step) {
                const lds_u32 *fb = fbbox + (s & 1u) * kP3FbWords;
                const uint32_t fbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb[kWave]);
                if (fbw & 2u) stop_at = s + 1u;
                if (fbw & 1u) { lds_barrier(); lds_barrier(); P3_LAP(p3_fb); }        // B, C: the others replay (the DC blocker never does)
            }
            return s == stop_at;
        };
        bool left = false;
        for (uint32_t s = 0; s < n_steps && !left; s += 2u) {
            left = step(s, std::integral_constant<int, 1>{});          // block s + 1 is odd when s is even
            if (!left && s + 1u < n_steps) left = step(s + 1u, std::integral_constant<int, 0>{});
        }
        P3_REPORT(4);
        if (left) return;
        lds_barrier();                                                 // (the final hand-shake of the other roles)
        D.store(S, c, C, counter1);
    } else if (DCW && role == 0u) {
        // ------------------------------ stage 1 (DCW): AGC + window push, block s ------------------
        for (uint32_t i = lane; i < (uint32_t)NTP; i += kWave) tlds[i] = i < (uint32_t)NT ? taps[i] : float4{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (FM && NT != 42) { float4 *ctl = reinterpret_cast<float4 *>(lds + PipeLayout<NT>::ctap_off); for (uint32_t i = lane; i < (uint32_t)(NT / 2); i += kWave) ctl[i] = taps[NT + i]; }
        if (LANES < (int)kWave && lane >= (uint32_t)LANES) return;
        P3_HWID(0);
        AgcStage<NT, MED3, FM> M;
        M.ycol = dcw_ring + lane;
        M.load(P, S, c, C, counter0, wcol);
        lds_barrier();                                                 // prologue: block 0's DC outputs are in the ring
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        bool left = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            M.rotate();
            if (s < n_blocks && !PROF_SKIP(P, 64)) M.block(P, wcol, s);
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (s >= 2u && s <= last_fb_step) {
                const lds_u32 *fb = fbbox + (s & 1u) * kP3FbWords;
                const uint32_t fbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb[kWave]);
                if (fbw & 2u) stop_at = s + 1u;
                if (fbw & 1u) {
                    const uint32_t v = fb[lane];
                    const bool new_locked = (v & 2u) != 0u;
                    if ((v & 1u) && new_locked != M.locked)
                        M.replay(P, wcol, s, (int)(v >> 8), new_locked, n_blocks);
                    lds_barrier();                                     // B: the window is corrected
                    lds_barrier();                                     // C: stage 2 has redone its block
                    P3_LAP(p3_fb);
                }
            }
            if (s == stop_at) { left = true; break; }
        }
        P3_REPORT(0);
        if (left) return;
        lds_barrier();                                                 // (stage 2 -> 3: final TED phase)
        M.store(P, S, c, C, counter1, wcol);
    } else if (role == 0u) {
        // ------------------------------ stage 1: sample phase, block s -------------------------
        for (uint32_t i = lane; i < (uint32_t)NTP; i += kWave) tlds[i] = i < (uint32_t)NT ? taps[i] : float4{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (FM && NT != 42) { float4 *ctl = reinterpret_cast<float4 *>(lds + PipeLayout<NT>::ctap_off); for (uint32_t i = lane; i < (uint32_t)(NT / 2); i += kWave) ctl[i] = taps[NT + i]; }
        if (LANES < (int)kWave && lane >= (uint32_t)LANES) return;      // (the wavefront goes on without them)
        P3_HWID(0);
        SampleStage<NT, MED3, SampleT, FM, CMODE> M;
        M.ycol = wring + (2u * (uint32_t)RING) * LP + lane;           // behind the window
        M.xl = xl; M.avail = avail_l;
        M.load(P, S, x, c, C, cin, Cin, counter0, wcol, n_blocks);
        P3_T0();
        // one step; BUF = s & 1 names the prefetch registers statically, so the loop runs two steps a turn
        uint32_t stop_at = 0xffffffffu;                                // time-parallel chunk: leave after this step
        auto step = [&](uint32_t s, auto buf) -> bool {
            M.rotate();
            if (s < n_blocks && !PROF_SKIP(P, 64)) {
                M.template fetch<decltype(buf)::value>(x, s, n_blocks, cin, Cin);
                P1_INPUTS_ARRIVED(kB);
                M.block(P, wcol);
            }
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (s >= 2u && s <= last_fb_step) {
                const lds_u32 *fb = fbbox + (s & 1u) * kP3FbWords;
                const uint32_t fbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb[kWave]);
                if (fbw & 2u) stop_at = s + 1u;
                if (fbw & 1u) {
                    const uint32_t v = fb[lane];
                    const bool new_locked = (v & 2u) != 0u;
                    if ((v & 1u) && new_locked != M.locked)
                        M.replay(P, wcol, (int)(v >> 8), new_locked, s - 1u < n_blocks, s < n_blocks);
                    lds_barrier();                                     // B: the window is corrected
                    lds_barrier();                                     // C: stage 2 has redone its block
                    P3_LAP(p3_fb);
                }
            }
            return s == stop_at;
        };
        bool left = false;
        for (uint32_t s = 0; s < n_steps && !left; s += 2u) {
            left = step(s, std::integral_constant<int, 0>{});
            if (!left && s + 1u < n_steps) left = step(s + 1u, std::integral_constant<int, 1>{});
        }
        P3_REPORT(0);
        if (left) return;                                              // handed over: this chunk's state is not needed
        lds_barrier();                                                 // (stage 2 -> 3: final TED phase)
        M.store(P, S, c, C, counter1, wcol);
    } else if (role == 1u) {
        // ------------------------------ stage 2: filters + timing loop, block s-1 --------------
        if (LANES < (int)kWave && lane >= (uint32_t)LANES) return;
        P3_HWID(1);
        if constexpr (DCW) lds_barrier();                              // prologue
        Lane L;
        lane_load(L, S, c);
        int cstar = next_fire_count(L.until_next_ted, L.ted_clock);
        int until = cstar - (int)L.ted_clock - 1;      // block-relative index of the firing sample
        uint32_t wpos = 0;
        [[maybe_unused]] const float inv_spt = 1.0f / P.samples_per_ted;
        // one block: the instant, if this lane has one in it; hands a completed symbol to stage 3
        S2_BEGIN();
        // A block's SECOND instant when the first one completed the symbol (-1: none).  A symbol may flip the AGC lock at its
        // sample (sync: receiver.rs:431); the correction that follows -- stage 1 replays the AGC from that sample on, this stage
        // goes back to before its current block -- reaches the next block's instants, but an instant that shares the SYMBOL's block
        // and follows its sample was filtered over the window as it stood before the replay.  It completes nothing, so its
        // soft sample is all that is wrong: the TED's newest tap, which the next symbol hands to the squelch as its first sample
        // (and the equalizer reads 24 symbols later).  Round 5 found it as one trial in 65 536 of configs[4] whose burst tail,
        // decoded from noise, differed from the oracle's in two bits; `a2_last` is what the correction below needs to redo it.
        int a2_now = -1, a2_last = -1;
        auto do_block = [&](uint32_t blk, uint32_t seq) {
            S2_LAP(4);
            a2_now = -1;
            uint32_t hdr = 0;
            float zero = 0.0f, sym = 0.0f, terr = 0.0f, next = 0.0f;
            if (until < kB) {
                const int fk = until;
                float sa_low;
                const float rem = L.until_next_ted - (float)cstar;          // receiver.rs:352
                if constexpr (FM) {
                    // FASTMATH: while the helper wavefront is still filtering, the relaxed timing loop for both signs of the
                    // soft sample it will deliver (ted_ahead_relaxed: bit-identical to running ted_timing_relaxed afterwards);
                    // what follows the filters on the chain to the next instant is then ted_commit's selection
                    const TedAhead A = ted_ahead_relaxed(P, L, inv_spt, rem);
                    SPIN_BEGIN();
                    while ((int32_t)(seqbox[0] - seq) < 0) {}
                    SPIN_END();
                    const float hm = __uint_as_float(markbox[lane]), hs = __uint_as_float(spacebox[lane]);
                    S2_LAP(1);
                    sa_low = __builtin_amdgcn_fmed3f(hm - hs, -1.0f, 1.0f);
                    int cs;
                    if (ted_commit(L, A, sa_low, &zero, &sym, &terr, &cs)) { hdr = 1u | ((uint32_t)fk << 8); next = L.until_next_ted; }
                    cstar = cs;
                } else if constexpr (SPLIT) {
                    float hm = 0.0f;
                    if constexpr (!HELPER_BOTH) hm = demod_half<NT, RING, 0>(tlds, wring, lane, wpos + (uint32_t)fk);
                    S2_LAP(0);
                    // while the helper wavefront is still filtering: the timing loop for both signs of the soft
                    // sample it will deliver (same_dev_common.h: ted_ahead / ted_commit)
                    const TedAhead A = ted_ahead(P, L, rem);
                    SPIN_BEGIN();
                    while ((int32_t)(seqbox[0] - seq) < 0) {}        // stage 4 has posted this pass
                    SPIN_END();
                    if constexpr (HELPER_BOTH) hm = __uint_as_float(markbox[lane]);
                    const float hs = __uint_as_float(spacebox[lane]);
                    S2_LAP(1);
                    sa_low = rs_clamp(hm - hs, -1.0f, 1.0f);
                    int cs;
                    if (ted_commit(L, A, sa_low, &zero, &sym, &terr, &cs)) { hdr = 1u | ((uint32_t)fk << 8); next = L.until_next_ted; }
                    cstar = cs;
                } else {
                    sa_low = demod_fast<NT, RING, true>(tlds, wring, lane, wpos + (uint32_t)fk);
                    if (ted_timing(P, L, sa_low, rem, &zero, &sym, &terr)) { hdr = 1u | ((uint32_t)fk << 8); next = L.until_next_ted; }
                    cstar = next_fire_count(L.until_next_ted, 0u);
                }
                until = fk + cstar;
                if (until < kB) {
                    // A second instant in the same block (20-sample blocks at 22.05 kHz, the loop at its
                    // fastest): rare, so this wavefront computes both filters itself.  Exactly one of the
                    // two instants completes a symbol.
                    const int fk2 = until;
                    COUNT_SECOND_INSTANT();
                    if (hdr & 1u) a2_now = fk2;                       // (the first instant completed the symbol: this one follows its sample)
                    float sa2;
                    if constexpr (FM) {
                        float hm2, hs2;
                        if constexpr (NT != 42) demod_pair_centred<NT, RING>(lds_addr(lds + PipeLayout<NT>::ctap_off), lds_addr(wcol), wpos + (uint32_t)fk2, &hm2, &hs2);
                        else demod_pair_relaxed<NTP, RING, false>(lds_addr(lds), lds_addr(wcol), wpos + (uint32_t)fk2, &hm2, &hs2);
                        sa2 = __builtin_amdgcn_fmed3f(hm2 - hs2, -1.0f, 1.0f);
                    } else {
                        sa2 = demod_fast<NT, RING, true>(tlds, wring, lane, wpos + (uint32_t)fk2);
                    }
                    const float rem2 = L.until_next_ted - (float)cstar;
                    float z2 = 0.0f, s2 = 0.0f, e2 = 0.0f;
                    if (FM ? ted_timing_relaxed(P, L, inv_spt, sa2, rem2, &z2, &s2, &e2) : ted_timing(P, L, sa2, rem2, &z2, &s2, &e2)) {
                        hdr = 1u | ((uint32_t)fk2 << 8); zero = z2; sym = s2; terr = e2; next = L.until_next_ted;
                    }
                    cstar = next_fire_count(L.until_next_ted, 0u);
                    until = fk2 + cstar;
                }
            }
            until -= kB;
            S2_LAP(2);
            if constexpr (SPLIT) posbox[((blk + 1u) & 1u) * kWave + lane] = (uint32_t)until;   // block blk + 1's instant
            lds_u32 *sb = symbox + (blk & 1u) * kP3SymWords + lane;
            sb[0] = hdr;
            sb[kWave] = __float_as_uint(zero); sb[2 * kWave] = __float_as_uint(sym);
            if (P.trace_cap) {                         // only the symbol trace records these two
                sb[3 * kWave] = __float_as_uint(terr); sb[4 * kWave] = __float_as_uint(next);
            }
            S2_LAP(3);
        };
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        bool left = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            // this lane's state before block s-1, in case stage 3 sends it back there
            const float k_h0 = L.h0, k_h1 = L.h1, k_h2 = L.h2, k_avg = L.period_avg, k_inst = L.period_inst,
                        k_unt = L.until_next_ted;
            const uint32_t k_flags = L.flags;
            const int k_cstar = cstar, k_until = until;
            const bool active = s >= 1u && s <= n_blocks;
            if (active && !PROF_SKIP(P, 32)) do_block(s - 1u, 2u * s + 1u);
            else if (SPLIT && s == 0u) posbox[lane] = (uint32_t)until;       // block 0's instant
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (s >= 2u && s <= last_fb_step) {
                const lds_u32 *fb = fbbox + (s & 1u) * kP3FbWords;
                const uint32_t fbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb[kWave]);
                if (fbw & 2u) stop_at = s + 1u;
                if (fbw & 1u) {
                    const uint32_t v = fb[lane];
                    if (v & 1u) {
                        L.h0 = k_h0; L.h1 = k_h1; L.h2 = k_h2; L.period_avg = k_avg; L.period_inst = k_inst;
                        L.until_next_ted = k_unt; L.flags = k_flags; cstar = k_cstar; until = k_until;
                        L.flags = (L.flags & ~F_BW_LOCKED) | ((v & 4u) ? F_BW_LOCKED : 0u);
                        if (v & 8u) {                                    // end(): symsync.reset()
                            L.flags &= ~F_TED_PHASE;
                            L.h0 = 0.0f; L.h1 = 0.0f; L.h2 = 0.0f;
                            L.period_avg = P.samples_per_ted; L.period_inst = P.samples_per_ted;
                        }
                    }
                    lds_barrier();                                     // B: stage 1 has corrected the window
                    if ((v & 1u) && a2_last >= 0 && (int)(v >> 8) < a2_last && !(v & 8u)) {
                        // the symbol's block (s - 2) had an instant behind the symbol's sample: its soft sample once more, over
                        // the corrected window; it is the newest tap of the TED this lane has just gone back to.  (After an
                        // end() the TED's taps have just been zeroed by the reset: nothing to correct.)
                        const uint32_t wprev = (wpos >= (uint32_t)kB ? wpos : wpos + (uint32_t)RING) - (uint32_t)kB;
                        if constexpr (FM) {
                            float hm2, hs2;
                            if constexpr (NT != 42) demod_pair_centred<NT, RING>(lds_addr(lds + PipeLayout<NT>::ctap_off), lds_addr(wcol), wprev + (uint32_t)a2_last, &hm2, &hs2);
                            else demod_pair_relaxed<NTP, RING, false>(lds_addr(lds), lds_addr(wcol), wprev + (uint32_t)a2_last, &hm2, &hs2);
                            L.h2 = __builtin_amdgcn_fmed3f(hm2 - hs2, -1.0f, 1.0f);
                        } else {
                            L.h2 = demod_fast<NT, RING, true>(tlds, wring, lane, wprev + (uint32_t)a2_last);
                        }
                    }
                    if ((v & 1u) && active) do_block(s - 1u, 2u * s + 2u);
                    lds_barrier();                                     // C
                    P3_LAP(p3_fb);
                }
            }
            a2_last = active ? a2_now : -1;
            if (active) { wpos += kB; if (wpos == (uint32_t)RING) wpos = 0; }
            if (s == stop_at) { left = true; break; }
        }
        P3_REPORT(1);
        S2_REPORT();
        if (left) return;
        phasebox[lane] = L.flags & F_TED_PHASE;
        lds_barrier();                                                 // stage 3 merges the phase bit
        L.ted_clock = (uint32_t)(cstar - until - 1);
        S.until_next_ted[c] = L.until_next_ted; S.ted_clock[c] = L.ted_clock;
        S.ted_h0[c] = L.h0; S.ted_h1[c] = L.h1; S.ted_h2[c] = L.h2;
        S.period_avg[c] = L.period_avg; S.period_inst[c] = L.period_inst;
    } else if (role == 2u) {
        // ------------------------------ stage 3: symbol path, block s-2 ------------------------
        if (LANES < (int)kWave && lane >= (uint32_t)LANES) return;
        P3_HWID(2);
        if constexpr (DCW) lds_barrier();                              // prologue
        Lane L;
        lane_load(L, S, c);
        L.ended = 0u;
        std::conditional_t<FM, RelaxFastCtx<NFF, NFB>, FastCtx<NFF, NFB>> X;
        X.hist = hcol;
        P3_MARKS_BEGIN(X, lds, NT);       // (profile builds: per-section marks of the symbol path)
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
#pragma unroll 2
        for (int i = 0; i < kSquelchHist; ++i) { const float *row = S.sq_hist + (size_t)i * C; hcol[i * LP] = row[c]; }
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        bool left = false, lane_done = false, leave_posted = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            bool any = false;
            if (s >= 2u && s <= last_fb_step) {
                const uint32_t blk = s - 2u;
                const lds_u32 *sb = symbox + (blk & 1u) * kP3SymWords + lane;
                // everything this step reads from LDS in one round trip: the symbol stage 2 handed on and the two
                // history samples its equalizer step takes (rx_symbol: slots +16/+17 from the squelch's write position)
                const uint32_t pslot = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;
                const float pre0 = X.hist_get((pslot + 16u) & 63u), pre1 = X.hist_get((pslot + 17u) & 63u);
                const uint32_t hdr = sb[0], zero_w = sb[kWave], sym_w = sb[2 * kWave];
                uint32_t fbv = 0, io0 = 0, io1 = 0xffffffffu, io2 = 0;
                bool want_slot = false;                                // this lane has just finished a burst
                if ((hdr & 1u) && !PROF_SKIP(P, 16)) {
                    const uint32_t fk = hdr >> 8;
                    const float zero = __uint_as_float(zero_w), sym = __uint_as_float(sym_w);
                    float terr = 0.0f, unt = 0.0f;
                    if (P.trace_cap) { terr = __uint_as_float(sb[3 * kWave]); unt = __uint_as_float(sb[4 * kWave]); }
                    const uint32_t before = L.flags & (F_AGC_LOCKED | F_BW_LOCKED);
                    L.ended = 0u;
                    uint32_t burst_len = 0;
                    bool emit = false;
                    const uint32_t link = symbol_link(P, L, S, X, c, zero, sym, terr, unt,
                                                      counter0 + (int64_t)row_l + (uint64_t)blk * kB + fk + 1u, &burst_len, &emit,
                                                      true, pre0, pre1);
                    if constexpr (NT == 42) want_slot = emit && link == 3u;
                    else if (emit && link == 3u) io1 = burst_to_pool(S, O, c);    // (44.1 / 48 kHz: the kernels are at the register limit and the shared copy below cost them 10 %)
                    io0 = 1u | (link << 1) | (emit ? 8u : 0u) | (fk << 4);
                    io2 = burst_len;
                    const uint32_t after = L.flags & (F_AGC_LOCKED | F_BW_LOCKED);
                    if (after != before || L.ended)
                        fbv = 1u | ((after & F_AGC_LOCKED) ? 2u : 0u) | ((after & F_BW_LOCKED) ? 4u : 0u) |
                              (L.ended ? 8u : 0u) | (fk << 8);
                }
                // Finished bursts go into the pool here, outside the lanes' divergent paths and with the whole wavefront:
                // one slot reservation for all of them and one round trip through memory per burst (72 words: a word per
                // lane and eight more at 64 lanes), where a lane by itself made seven round trips of ~1 us each -- with the whole
                // workgroup waiting at the step barrier for it, in ~2 % of a 64-channel workgroup's steps.
                if constexpr (NT == 42) {
                    uint64_t pend = __builtin_amdgcn_ballot_w64(want_slot);
                    if (pend != 0ull) {
                        const uint32_t n_new = (uint32_t)__popcll(pend);
                        uint32_t base = 0;
                        if (lane == 0u) base = atomicAdd(O.n_events + 1, n_new);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        uint32_t k = 0;
                        while (pend != 0ull) {
                            const int j = __builtin_ctzll(pend);
                            pend &= pend - 1ull;
                            const uint32_t b = base + k++;
                            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
                            if (b < O.burst_cap) {
                                const uint32_t *src = reinterpret_cast<const uint32_t *>(S.fr_msg + (size_t)cj * kBurstCap);
                                uint32_t *dst = reinterpret_cast<uint32_t *>(O.bursts + (size_t)b * kBurstCap);
                                // (narrow workgroups have only LANES lanes left in this wavefront: more words per lane)
                                constexpr uint32_t kWords = (uint32_t)kBurstCap / 4u, kPer = (kWords + (uint32_t)LANES - 1u) / (uint32_t)LANES;
                                uint32_t t[kPer];
#pragma unroll
                                for (uint32_t i = 0; i < kPer; ++i) { const uint32_t w = lane + i * (uint32_t)LANES; t[i] = w < kWords ? src[w] : 0u; }
#pragma unroll
                                for (uint32_t i = 0; i < kPer; ++i) { const uint32_t w = lane + i * (uint32_t)LANES; if (w < kWords) dst[w] = t[i]; }
                                if ((int)lane == j) io1 = b;
                            } else {
                                if (lane == 0u) atomicOr(O.n_events + 2, 2u);
                                if ((int)lane == j) io1 = 0xffffffffu;
                            }
                        }
                    }
                }
                lds_u32 *io = iobox + (s & 1u) * kP3IoWords + lane;
                io[0] = io0;
                if (__builtin_amdgcn_ballot_w64((io0 & 8u) != 0u && ((io0 >> 1) & 3u) == 3u) != 0ull) { io[kWave] = io1; io[2 * kWave] = io2; }
                lds_u32 *fb = fbbox + (s & 1u) * kP3FbWords;
                fb[lane] = fbv;
                any = __builtin_amdgcn_ballot_w64(fbv != 0u) != 0ull;
                // Time-parallel chunk that hands over: from its nominal end on, a lane's
                // hand-over instant is the end of the first block after which its link state is NoCarrier;
                // once every lane has one the workgroup leaves (one more step: stage 4 still has to log
                // this block's events).
                uint32_t leave = 0u;
                if (may_leave && !leave_posted) {
                    // (n_nominal is per lane when the chunk boundaries are per channel)
                    if (!lane_done && blk + 1u >= n_nominal && (L.flags & F_LINK_MASK) == 0u && (xl == nullptr || blk < avail_l)) {
                        lane_done = true;
                        K.handover[c] = counter0 + (int64_t)row_l + (uint64_t)(blk + 1u) * kB;
                    }
                    if (__builtin_amdgcn_ballot_w64(!lane_done) == 0ull) { leave = 2u; leave_posted = true; stop_at = s + 1u; }
                }
                if (lane == 0u) fb[kWave] = (any ? 1u : 0u) | leave;
            }
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (any) { lds_barrier(); lds_barrier(); P3_LAP(p3_fb); }   // B, C: the earlier stages catch up
            if (s == stop_at) { left = true; break; }
        }
        P3_REPORT(2);
        P3_MARKS_REPORT(X);
        if (left) return;
        lds_barrier();                                                 // stage 2's TED phase, stage 4's wake-up flag
        L.flags = (L.flags & ~(F_TED_PHASE | F_TICK_AGAIN)) | (phasebox[lane] & F_TED_PHASE) | (againbox[lane] & F_TICK_AGAIN);
        S.sq_data[c] = L.sq_data; S.sq_power[c] = L.sq_power; S.sq_phist[c] = L.sq_phist;
        S.sq_fill[c] = L.sq_fill; S.sq_clock[c] = L.sq_clock; S.sq_symbols[c] = L.sq_symbols;
        S.eq_word[c] = L.eq_word; S.eq_count[c] = L.eq_count;
        S.fr_word[c] = L.fr_word; S.fr_count[c] = L.fr_count; S.fr_invalid[c] = L.fr_invalid;
        S.fr_len[c] = L.fr_len; S.flags[c] = L.flags;
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
#pragma unroll 2
        for (int i = 0; i < kSquelchHist; ++i) { float *row = S.sq_hist + (size_t)i * C; row[c] = hcol[i * LP]; }
    } else {
        // ------------------------------ stage 4: link events + wake-ups, block s-3 -------------
        // PACKED: lanes LANES .. 2*LANES-1 stay for the filter half of every step (the space filter of channel
        // lane - LANES, while lanes 0 .. LANES-1 compute the mark filter) and sit the event half out.
        const bool evt_lane = lane < (uint32_t)LANES;
        if (LANES < (int)kWave && lane >= (uint32_t)(PACKED ? 2 * LANES : LANES)) return;
        if constexpr (DCW) lds_barrier();                              // prologue
        const uint32_t fch = PACKED ? (lane & (uint32_t)(LANES - 1)) : lane;      // the channel this lane filters for
        const uint32_t which = PACKED ? ((lane & (uint32_t)LANES) ? 1u : 0u) : 1u;   // 0 mark, 1 space
        Lane L;
        if (evt_lane) lane_load(L, S, c);      // uses sq_symbols, tk_next, tk_last, wake_*, F_TICK_AGAIN
        P3_HWID(3);
        IoCtxLds X;
        X.chunk = chunkbox;
        chunkbox[0] = 0u; chunkbox[1] = kEvChunk;       // nothing reserved yet
        X.pending_slot = 0xffffffffu;
        X.tk = tkbox + lane;
        if (evt_lane) X.ring_load(P, S, c);
        uint32_t wpos = 0;                     // SPLIT: ring slot of block s-1's first sample
        if (SPLIT && lane == 0u) seqbox[0] = 0u;
        uint32_t stop_at = 0xffffffffu;
        bool left = false;
        P3_T0();
        auto filter = [&](uint32_t pos) {
            if constexpr (PACKED) {
                const float mag = demod_half_dyn<NT, RING>(tlds, wring, fch, wpos + pos, which);
                (which ? spacebox : markbox)[fch] = __float_as_uint(mag);
            } else if constexpr (HELPER_BOTH) {
                float hm, hs;
                if constexpr (FM && NT == 42) demod_pair_relaxed_42<RING>(lds_addr(lds), lds_addr(wcol), wpos + pos, &hm, &hs);
                else if constexpr (FM) demod_pair_centred<NT, RING>(lds_addr(lds + PipeLayout<NT>::ctap_off), lds_addr(wcol), wpos + pos, &hm, &hs);      // 44.1 / 48 kHz: centred taps, 5 instructions per tap pair
                else demod_pair<NT, RING>(tlds, wring, lane, wpos + pos, &hm, &hs);
                markbox[lane] = __float_as_uint(hm); spacebox[lane] = __float_as_uint(hs);
            } else {
                spacebox[lane] = __float_as_uint(demod_half<NT, RING, 1>(tlds, wring, lane, wpos + pos));
            }
        };
        for (uint32_t s = 0; s < n_steps; ++s) {
            // SPLIT: first the matched filter(s) of block s-1 for stage 2, which waits for them
            const bool active = SPLIT && s >= 1u && s <= n_blocks;
            uint32_t pos = 0xffffffffu;
            // what stage 3 posted last step, read together with the instant's position (one LDS round trip, not two)
            const bool evt_step = s >= 3u && evt_lane && !PROF_SKIP(P, 8);
            const lds_u32 *io = iobox + ((s - 1u) & 1u) * kP3IoWords + lane;
            const uint32_t io0 = evt_step ? io[0] : 0u;
            HELP_BEGIN();
            if (active) {
                pos = posbox[((s - 1u) & 1u) * kWave + fch];               // posted by stage 2 during the last step
                if (pos < (uint32_t)kB && !PROF_SKIP(P, 256)) filter(pos);
                if (lane == 0u) seqbox[0] = 2u * s + 1u;                    // (LDS operations of a wavefront stay in order)
            }
            HELP_END(SPLIT);
            if (evt_step) {
                const uint32_t blk = s - 3u;
                if (io0 & 1u) {
                    L.sq_symbols += 1;         // as rx_symbol counted it (rx/codesquelch.rs:232)
                    const uint32_t link = (io0 >> 1) & 3u, fk = (io0 >> 4) & 63u;
                    const bool burst = (io0 & 8u) != 0u && link == 3u;
                    uint32_t burst_len = 0;
                    if (burst) { X.pending_slot = io[kWave]; burst_len = io[2 * kWave]; }
                    symbol_io(P, L, S, O, X, c, link, (io0 & 8u) != 0u, counter0 + (int64_t)row_l + (uint64_t)blk * kB + fk + 1u, burst_len);
                }
            }
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (s >= 2u && s <= last_fb_step) {
                const lds_u32 *fb = fbbox + (s & 1u) * kP3FbWords;
                const uint32_t fbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb[kWave]);
                if (fbw & 2u) stop_at = s + 1u;
                if (fbw & 1u) {
                    const uint32_t v = SPLIT ? fb[fch] : 0u;
                    lds_barrier();                                     // B: stage 1 has corrected the window
                    if (SPLIT) {
                        if ((v & 1u) && active && pos < (uint32_t)kB) filter(pos);
                        if (lane == 0u) seqbox[0] = 2u * s + 2u;
                    }
                    lds_barrier();                                     // C
                }
            }
            if (active) { wpos += kB; if (wpos == (uint32_t)RING) wpos = 0; }
            if (s == stop_at) { left = true; break; }
        }
        P3_REPORT(3);
        if (evt_lane) X.retire(O, lane, (uint32_t)LANES);
        if (left) return;
        if (evt_lane) againbox[lane] = L.flags & F_TICK_AGAIN;
        lds_barrier();                                                 // stage 3 merges the flag bits
        if (evt_lane) { S.tk_next[c] = L.tk_next; S.tk_last[c] = L.tk_last; S.wake_fired[c] = L.wake_fired; X.ring_store(P, S, c); }
    }
}

// ---------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------
template <int NT, bool DCW = false>
static constexpr size_t pipe_lds_bytes()
{
    using LY = PipeLayout<NT>;
    return ((size_t)LY::tap_floats + kP3MailWords + (DCW ? LY::dcw_ring_floats : LY::yring_floats) +
            (size_t)(kSquelchHist + 2 * LY::RING - LY::B) * kWave) * sizeof(float);
}

// The pipeline pays while SIMDs are idle.  Whole groups of 64 channels only.  Measured at
// 22.05 kHz: it wins up to 32 768 channels (two workgroups of four wavefronts per CU), the
// one-wavefront kernel from 49 152 on.  At 44.1 / 48 kHz a workgroup's window ring is 72 KB of
// the CU's 160 KB of LDS, so one workgroup per CU and 16 384 channels at a time; two rounds of
// them (32 768 channels: 13.1 ms for 2 s at 48 kHz) still beat one wavefront per 64 channels
// (16.0 ms), three do not.  Returns 0 (not selected) or non-zero.
// v_med3_f32 == f32::clamp unless a bound is -0.0 (or NaN, which the builder rejects)
static bool agc_clamp_is_med3(const Params &P)
{ return !(P.agc_min == 0.0f && std::signbit(P.agc_min)) && !(P.agc_max == 0.0f && std::signbit(P.agc_max)); }

// channels per workgroup: 16 while that still leaves the batch within one workgroup per CU
// (default equalizer only: fewer kernels to build), else 64
static uint32_t pipe_lanes(const Params &P)
{
    if (!(P.eq_nff == 6u && P.eq_nfb == 4u) || !agc_clamp_is_med3(P)) return kWave;
    if (P.knob_pipe_lanes == 16 || P.knob_pipe_lanes == 32 || P.knob_pipe_lanes == 64) return (uint32_t)P.knob_pipe_lanes;
    if (P.n_channels <= 16u * 256u && P.n_channels % 16u == 0u) return 16u;
    if (P.n_channels <= 32u * 256u && P.n_channels % 32u == 0u) return 32u;
    return kWave;
}

uint32_t pipe_kernel_stages(const Params &P)
{
    const bool r22 = P.ntaps == 42u && P.dc_len == 16u, r48 = P.ntaps == 92u && P.dc_len == 35u,
               r44 = P.ntaps == 84u && P.dc_len == 32u;
    if (!(r22 || r48 || r44) || (P.n_channels % pipe_lanes(P)) != 0u) return 0;
    if (!((P.eq_nff == 6u && P.eq_nfb == 4u) || (P.eq_nff == 1u && P.eq_nfb == 1u))) return 0;
    if (P.block_len != 16u || max_block_len(P) < (r22 ? (uint32_t)kBlockMirror : pipe_block_len(P))) return 0;
    if (P.knob_pipe != 0) return P.knob_pipe > 0 ? 4u : 0u;
    // beyond two workgroups per CU the pipeline runs in rounds; 22.05 kHz, sustained 2 s launches with the transport layer on
    // (tools/big_sustained_strict.py): 65 536 channels 7.97 ms against the one-wavefront kernel's 9.06, 131 072: 15.6 against
    // 15.0, 262 144: 30.5 against 29.2
    return P.n_channels <= (r22 ? 65536u : 32768u) ? 4u : 0u;
}
bool pipe_kernel_selected(const Params &P) { return pipe_kernel_stages(P) != 0u; }
uint32_t pipe_block_len(const Params &P)
{ return P.ntaps == 42u ? (uint32_t)kBlockPipe22 : (P.ntaps == 92u ? (uint32_t)PipeGeom<92>::B : (uint32_t)PipeGeom<84>::B); }

template <int NT, int NFF, int NFB, bool M3, bool SHARE, int LANES, bool SPLIT, typename SampleT, bool FM = false>
static hipError_t launch_pipe_one(const Params &P, const State &S, const Output &O, const float4 *taps,
                                  const SampleT *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                                  const PipeChunks &K)
{
    constexpr bool DCW = pipe_dcw<NT, LANES, SPLIT>();
    constexpr size_t lds = pipe_lds_bytes<NT, DCW>();
    // The two-per-CU builds (what time-parallel launches and batches beyond 16 384 channels run): one build per input
    // form (see SampleStage); everything else decides at run time
    constexpr bool FORMS = (FM && NT == 42) || (SHARE && SPLIT && LANES == 64);
    const bool cm = K.n_chunks > 1u && K.col_row0 != nullptr;
    auto *kernel = !FORMS ? demod_pipe_kernel<NT, NFF, NFB, M3, SHARE, LANES, SPLIT, SampleT, FM, 2>
                          : (cm ? demod_pipe_kernel<NT, NFF, NFB, M3, SHARE, LANES, SPLIT, SampleT, FM, FORMS ? 1 : 2>
                                : demod_pipe_kernel<NT, NFF, NFB, M3, SHARE, LANES, SPLIT, SampleT, FM, FORMS ? 0 : 2>);
    if (lds > 64u * 1024u) {
        // more than the default 64 KB of dynamic LDS per workgroup: opt in, once per kernel and device
        static bool opted_in[2][64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!opted_in[cm][dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            opted_in[cm][dev] = true;
        }
    }
    if (K.n_chunks > 1u && (K.in_channels % (uint32_t)LANES) != 0u) return hipErrorInvalidValue;   // a workgroup would straddle chunks
    hipLaunchKernelGGL(kernel, dim3(P.n_channels / (uint32_t)LANES), dim3((DCW ? 5 : 4) * kWave), lds, stream, P, S, O, taps, x, n_blocks, counter0, K);
    return hipGetLastError();
}

template <int NT, typename SampleT>
static hipError_t launch_pipe_cfg(const Params &P, const State &S, const Output &O, const float4 *taps,
                                  const SampleT *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                                  const PipeChunks &K, bool relaxed)
{
    if constexpr (NT == 42) {
        // FASTMATH: the 64-channel two-per-CU form with the relaxed arithmetic (pipe_relaxed_supported)
        if (relaxed) {
            if (!pipe_relaxed_supported(P)) return hipErrorInvalidValue;
            if (P.eq_nff == 6u && P.eq_nfb == 4u)
                return launch_pipe_one<NT, 6, 4, true, true, 64, true, SampleT, true>(P, S, O, taps, x, n_blocks, counter0, stream, K);
            return launch_pipe_one<NT, 1, 1, true, true, 64, true, SampleT, true>(P, S, O, taps, x, n_blocks, counter0, stream, K);
        }
    } else {
        // FASTMATH at 44.1 / 48 kHz: the one-per-CU form with the DC wavefront
        if (relaxed) {
            if (!pipe_relaxed_supported(P)) return hipErrorInvalidValue;
            if (P.eq_nff == 6u && P.eq_nfb == 4u)
                return launch_pipe_one<NT, 6, 4, true, false, 64, true, SampleT, true>(P, S, O, taps, x, n_blocks, counter0, stream, K);
            return launch_pipe_one<NT, 1, 1, true, false, 64, true, SampleT, true>(P, S, O, taps, x, n_blocks, counter0, stream, K);
        }
    }
    // two workgroups per CU (22.05 kHz only, where their LDS allows it): the register-capped build, with
    // stage 2 split (same box, 32 768 channels x 2 s: 4.27-4.29 ms unsplit, 4.18-4.24 ms split).  Which stages
    // of the two workgroups meet on a SIMD makes no measurable difference there (dealt by SIMD id: like + like
    // 4.14 ms, stage 1 + 2 and 3 + 4 4.19 ms, 1 + 4 and 2 + 3 4.20 ms, by wavefront number 4.20 ms, one box),
    // so they are left where they fall.
    constexpr bool CAN_SHARE = (NT == 42);
    const bool share = CAN_SHARE && (P.knob_pipe_share != 0 ? P.knob_pipe_share > 0 : P.n_channels > 16384u);
    const bool med3 = agc_clamp_is_med3(P);
    const bool share_split = P.knob_pipe_split != 0 ? P.knob_pipe_split > 0 : true;
#define SAME_PIPE_LAUNCH(NFF, NFB, M3)                                                                                  \
    (share ? (share_split ? launch_pipe_one<NT, NFF, NFB, M3, CAN_SHARE, 64, CAN_SHARE, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K)    \
                          : launch_pipe_one<NT, NFF, NFB, M3, CAN_SHARE, 64, false, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K))     \
           : launch_pipe_one<NT, NFF, NFB, M3, false, 64, false, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K))
    const uint32_t lanes = pipe_lanes(P);
    if (P.eq_nff == 6u && P.eq_nfb == 4u && med3 && !share) {
        // the default configuration: narrow workgroups for small batches, and stage 2 split with stage 4's
        // wavefront wherever stage 2 is (one of) the longest -- everywhere except 64-channel workgroups at
        // 22.05 kHz, whose symbol stage is longer still
        const bool split = P.knob_pipe_split != 0 ? P.knob_pipe_split > 0 : (NT != 42 || lanes != kWave);
#define SAME_PIPE_LANES_LAUNCH(LN)                                                                                          \
        (split ? launch_pipe_one<NT, 6, 4, true, false, LN, true, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K)    \
               : launch_pipe_one<NT, 6, 4, true, false, LN, false, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K))
        if (lanes == 16u) return SAME_PIPE_LANES_LAUNCH(16);
        if (lanes == 32u) return SAME_PIPE_LANES_LAUNCH(32);
        return SAME_PIPE_LANES_LAUNCH(64);
#undef SAME_PIPE_LANES_LAUNCH
    }
    if (P.eq_nff == 6u && P.eq_nfb == 4u) return med3 ? SAME_PIPE_LAUNCH(6, 4, true) : SAME_PIPE_LAUNCH(6, 4, false);
    return med3 ? SAME_PIPE_LAUNCH(1, 1, true) : SAME_PIPE_LAUNCH(1, 1, false);
#undef SAME_PIPE_LAUNCH
}

template <typename SampleT>
static hipError_t launch_pipe_t(const Params &P, const State &S, const Output &O, const float4 *taps,
                                const SampleT *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                                const PipeChunks &K, bool relaxed)
{
    if (P.ntaps == 42u) return launch_pipe_cfg<42, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K, relaxed);
    if (P.ntaps == 92u) return launch_pipe_cfg<92, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K, relaxed);
    return launch_pipe_cfg<84, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K, relaxed);
}

hipError_t launch_demod_pipe(const Params &P, const State &S, const Output &O, const float4 *taps,
                             const float *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K, bool relaxed)
{ return launch_pipe_t<float>(P, S, O, taps, x, n_blocks, counter0, stream, K, relaxed); }
hipError_t launch_demod_pipe_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                                 const int16_t *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K, bool relaxed)
{ return launch_pipe_t<int16_t>(P, S, O, taps, x, n_blocks, counter0, stream, K, relaxed); }
// The FASTMATH build exists for the three rates the pipeline is built for, in 64-channel workgroups (whole groups of 64
// state columns), default or disabled equalizer, a non-negative AGC floor
bool pipe_relaxed_supported(const Params &P)
{
    const bool geom = (P.ntaps == 42u && P.dc_len == 16u) || (P.ntaps == 92u && P.dc_len == 35u) || (P.ntaps == 84u && P.dc_len == 32u);
    return geom && (P.n_channels % kWave) == 0u && P.agc_min >= 0.0f && pipe_kernel_stages(P) != 0u &&
           pipe_lanes(P) == kWave && ((P.eq_nff == 6u && P.eq_nfb == 4u) || (P.eq_nff == 1u && P.eq_nfb == 1u));
}
uint32_t pipe_workgroup_channels(const Params &P) { return pipe_lanes(P); }

}  // namespace same

// ---- the software-pipelined matched filters against the chunk-at-a-time loop ----------------------------------------
// demod_pair_relaxed_42 / demod_pair_relaxed_chunks keep up to 21 LDS loads in flight and name the part that has landed
// with s_waitcnt lgkmcnt(9 / 12): that relies on LDS returning in order and on the counter (4 bits) holding issue back
// rather than wrapping.  They add the same products in the same order as demod_pair_relaxed, whose waits are all
// lgkmcnt(0): the three must agree bit for bit on any window (tests/test_gpu_parity.py).
namespace same {
constexpr int kFormsRing = 160;
template <int NCH>
__global__ __launch_bounds__(kWave) void filter_forms_kernel(const float4 *__restrict__ taps, const float *__restrict__ win, float *__restrict__ out)
{
    constexpr int NTP = NCH * kRelaxChunk, RING = kFormsRing;
    constexpr uint32_t tap_floats = (uint32_t)((NTP * 4 + 63) / 64 * 64);
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x;
    float4 *tlds = reinterpret_cast<float4 *>(lds);
    for (uint32_t i = lane; i < (uint32_t)NTP; i += kWave) tlds[i] = taps[i];
    float *wcol = lds + tap_floats + lane;
    for (int s = 0; s < RING; ++s) { const float v = win[s * (int)kWave + (int)lane]; wcol[s * (int)kWave] = v; wcol[(s + RING) * (int)kWave] = v; }
    __syncthreads();
    for (uint32_t newest = 0; newest < (uint32_t)RING; ++newest) {
        float am, as, bm, bs, cm, cs;
        demod_pair_relaxed<NTP, RING, false>(lds_addr(lds), lds_addr(wcol), newest, &am, &as);
        demod_pair_relaxed_chunks<NCH, RING>(lds_addr(lds), lds_addr(wcol), newest, &bm, &bs);
        if constexpr (NCH == 3) demod_pair_relaxed_42<RING>(lds_addr(lds), lds_addr(wcol), newest, &cm, &cs);
        else { cm = bm; cs = bs; }
        float *o = out + ((size_t)newest * kWave + lane) * 6u;
        o[0] = am; o[1] = as; o[2] = bm; o[3] = bs; o[4] = cm; o[5] = cs;
    }
}
}  // namespace same

// taps: n_chunks * 14 float4 (mark re/im, space re/im), win: [160][64] f32, out: [160][64][6] f32 -- all device pointers
extern "C" int same_debug_filter_forms(int n_chunks, const void *d_taps, const float *d_win, float *d_out)
{
    using namespace same;
    const size_t lds = ((size_t)((n_chunks * kRelaxChunk * 4 + 63) / 64 * 64) + (size_t)2 * kFormsRing * kWave) * sizeof(float);
    const float4 *t = static_cast<const float4 *>(d_taps);
    auto go = [&](auto kernel) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        hipLaunchKernelGGL(kernel, dim3(1), dim3(kWave), lds, nullptr, t, d_win, d_out);
        return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : 1;
    };
    switch (n_chunks) {
    case 3: return go(filter_forms_kernel<3>);
    case 6: return go(filter_forms_kernel<6>);
    case 7: return go(filter_forms_kernel<7>);
    default: return 2;
    }
}

PIPE_PROFILE_EXPORTS()
