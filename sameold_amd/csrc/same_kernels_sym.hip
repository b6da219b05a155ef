// same_kernels_sym.hip -- the symbol-paced wavefront pipeline: relaxed arithmetic, 22.05 kHz, 36-sample steps.
//
// The four-stage pipeline of same_kernels_pipe.hip steps in blocks of 20 samples, because a block may hold at most one
// TED instant for the matched filters to be evaluated "after the block".  A symbol, however, is two instants (42.3
// samples): with 64 lanes at 64 symbol phases every 20-sample step runs the filters, the timing loop and the whole
// symbol path for the ~47 % of its lanes that happen to have an instant in it -- twice per symbol at half occupancy --
// and the chain filters -> timing update -> next instant sits inside every step (DESIGN.md 4.7, round 3's attribution).
//
// Here the unit of work downstream of the sample phase is the SYMBOL, not the block:
//   * of the two instants of a symbol only the second one (B, the one that completes the symbol) feeds a decision back
//     into the timing loop; the first (A) just shifts the TED's history and adds the clock offset to the period
//     (rx/symsync.rs:236-241), so where B falls is known as soon as A's position is -- before either filter has run.
//     The two matched-filter pairs of a symbol are therefore evaluated side by side, on two wavefronts (S at A, E at
//     B), and E runs the two timing updates after them: once per symbol and lane;
//   * a step is 36 samples, less than the shortest symbol the timing loop can command (two instants at least 19 samples
//     apart each: max_block_len), so every lane completes AT MOST one symbol per step and ~85 % of the lanes complete
//     exactly one: filters, timing loop and symbol path run once per step at ~85 % occupancy;
//   * the sample phase (DC blocker, AGC, window push) runs 36 samples per step and barrier: the fixed costs of a step
//     (barrier, feedback word, mailboxes) are paid per 36 samples instead of per 20.
//
// Four wavefronts per 64 state columns, on the four SIMDs of a CU, their work per step balanced by instruction count
// (the first cut -- sample phase | filters + timing | symbol path | events -- ran 4 900 / 5 400 / 3 200 / 700 clk per step):
//   T  input prefetch (a whole step ahead) and DC blocker of block s + 1 -- it takes no feedback from anything
//      (rx/dcblock.rs:45-49) -- handed to S through a two-block LDS ring; then the link events and transport wake-ups of
//      what Y handed over one step earlier
//   S  the matched-filter pair at the FIRST instant of the step's symbol (E posts where); then AGC (relaxed) and window push
//      of block s from the DC blocker's outputs (it keeps those of its last three blocks, packed f16: the gain an
//      AGC lock freezes is that of a sample up to two blocks back)
//   E  the symbol whose instants lie in finished samples (blocks < s): the matched-filter pair at its SECOND instant, then
//      the two timing updates and where the next symbol's instants fall
//   Y  symbol path of the symbol E handed over one step earlier: squelch, equalizer, framer (same_dev_common.h); the
//      squelch's sample history stays in the HBM state arrays (two loads a step, issued ahead)
// Feedback (agc.lock / loop bandwidth / symsync.reset, receiver.rs:431-432, 479-490) never sends a wavefront back: Y posts
// a change with the symbol it happened at, one word per lane.  S freezes the AGC from its next block on AT THE GAIN IT HAD
// after that symbol's sample (recomputed from the block's start out of the packed history; the gain every soft symbol
// of the burst is scaled by is therefore strict mode's to rounding), or releases it from its next block on; E switches the
// loop bandwidth, or resets the TED, behind its next symbol -- two symbols late.  The first form of this kernel replayed a
// lock at its sample (two more barriers in ~5 % of the steps, S and E filtering the symbol again): 8 % slower, and
// nothing the contract below asks for came of it.
//
// Window ring: five blocks of 36 slots, the first 13 slots stored twice (a 14-tap filter chunk never wraps).  T reads at
// most 119 samples back from the end of block s-1 (a lane may lag up to 52 samples behind after a symsync.reset, the
// symbol's first filter reaches 25 + 41 further) while S writes block s: four readable blocks and the one being written.
//
// Parity contract: that of SAME_BATCH_RELAXED / the time-parallel mode (include/same_rx.h): transmitted bytes and
// transport messages equal strict mode's, link events within SAME_TP_EVENT_TOLERANCE_SYMBOLS symbols, soft symbols of
// an open squelch within 0.05.  The arithmetic is same_relaxed_common.h's.
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "same_dev_common.h"
#include "same_device.h"
#include "same_fast_common.h"
#include "same_launch.h"
#include "same_pipe_common.h"
#include "same_profile.h"
#include "same_relaxed_common.h"

namespace same {

constexpr int kSymBlock = 36;
constexpr uint32_t kSymDrain = 5u;        // steps after the last block in which T processes the instants left before the end of the input, one per step
template <int NT> struct SymLayout {
    // The window ring holds SIX blocks: the four the filters may reach into, the one S is turning from DC-blocker outputs into
    // AGC outputs IN PLACE, and the one T is writing DC-blocker outputs to (there is no hand-over ring between T and S).  Its first
    // NT - 1 slots are stored twice: a filter's 42 samples never wrap.
    static constexpr int B = kSymBlock, DCL = 16, NBLK = 6, RING = NBLK * B, MIR = NT - 1;
    static constexpr uint32_t tap_floats = (uint32_t)((NT * 4 + PIPE_PROF_TAP_PAD + 63) / 64 * 64);
    static constexpr uint32_t sym_words = 5u * kWave;             // per parity: header, zero, sym, terr, until
    static constexpr uint32_t fb_words = kWave + 32u;             // per parity: one word per lane + the any-flag
    static constexpr uint32_t io_words = 3u * kWave;              // per parity: symbol word, burst-pool slot, burst length
    static constexpr uint32_t pos_words = 2u * kWave;             // per parity: ring slots of the symbol's two instants (-1: none)
    static constexpr uint32_t mail_words = 2u * sym_words + 2u * fb_words + 2u * io_words + 2u * kWave +   // + final TED phase, wake-up flag
                                           2u * pos_words + kWave +                                       // + the first instant's soft sample, from S
                                           kIoRingWords +                                                 // + T's deadline ring and its count
                                           kWave;                                                         // + Y -> T: this lane has handed over
    static constexpr size_t lds_bytes = ((size_t)tap_floats + mail_words + (size_t)(RING + MIR) * kWave) * sizeof(float);
    static_assert(NT == 42, "the filter's load sequence is written out for 42 taps");
    static_assert(B % 4 == 0 && B >= DCL && B % 2 == 0, "16-byte loads per lane; the DC windows are the tail of a block");
    static_assert(MIR <= 2 * B, "the mirrored slots are the first block and the head of the second");
    // reach: lag <= 52, first instant of a symbol <= 25 before the second, NT - 1 taps back
    static_assert(52 + 1 + 25 + (NT - 1) <= (NBLK - 2) * B, "the filters would read a block that is being written");
    static_assert(lds_bytes <= 80u * 1024u, "two workgroups per CU");
};

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop that is straight-line code from the start.  A
// `#pragma unroll` loop over a register array is still a loop with a runtime index when the first scalar-replacement pass
// runs; where the vectoriser then gets at the array before the next one (overlapping pair loads of the DC blocker's input
// window), the array stays in scratch memory -- and on gfx950 a scratch load counts in vmcnt, so every read of it also
// waited for the input prefetch just issued (T's DC blocker: 5 800 clk per step instead of 2 000).
template <typename F, int... I>
__device__ __forceinline__ void sym_static_for_(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sym_static_for(F &&f) { sym_static_for_(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }

// FskDemod::demod_now rx/demod.rs:156-164 at one instant, with the taps in registers.  The matched filter is a cisoid, h[i] = (2/N) e^{-j a (N-1-i)}
// (rx/waveform.rs:39-64), and only its output's magnitude is used, so it may be turned by a unit phasor: u[k] = h[k] e^{j a (N-1)/2}
// has u[N-1-k] = conj(u[k]) (same_config.cpp derives u from the reference's f32 taps), and for a real window w (w_i = the
// sample tap i meets)
//     sum_i w_i u_i = sum_{k<N/2} (w_k + w_{N-1-k}) Re u_k  +  j sum_{k<N/2} (w_k - w_{N-1-k}) Im u_k.
// Per k: ONE LDS load brings w_k and w_{N-1-k} (slots base + N-1-k and base + k of a ring whose first N - 1 slots are stored
// twice, so nothing wraps), one packed add makes (sum, difference), and two packed multiply-adds serve the real parts of both
// tones and the imaginary parts of both tones: 63 vector instructions and 21 loads for both filters (the A + C conj(B)
// form of round 4: 84 multiply-adds and a complex fix-up per tone).  42 registers pairs hold the taps for the whole launch.
// Relaxed arithmetic: another association of the same sum (rx/filter.rs:363-377 adds 42 products newest first).
struct SymWin { float2v w0, w1, w2, w3, w4, w5, w6; };
// pair K of group G: k = 7 G + K -> slots base + k (the sample tap N-1-k meets) and base + N-1-k (tap k)
#define SYM_WLOAD_(W_, wa_, o0_, o1_, o2_, o3_, o4_, o5_, o6_, p0_, p1_, p2_, p3_, p4_, p5_, p6_)                     \
    asm volatile("ds_read2st64_b32 %[w0], %[wa] offset0:" #o0_ " offset1:" #p0_ "\n\t"                           \
                 "ds_read2st64_b32 %[w1], %[wa] offset0:" #o1_ " offset1:" #p1_ "\n\t"                           \
                 "ds_read2st64_b32 %[w2], %[wa] offset0:" #o2_ " offset1:" #p2_ "\n\t"                           \
                 "ds_read2st64_b32 %[w3], %[wa] offset0:" #o3_ " offset1:" #p3_ "\n\t"                           \
                 "ds_read2st64_b32 %[w4], %[wa] offset0:" #o4_ " offset1:" #p4_ "\n\t"                           \
                 "ds_read2st64_b32 %[w5], %[wa] offset0:" #o5_ " offset1:" #p5_ "\n\t"                           \
                 "ds_read2st64_b32 %[w6], %[wa] offset0:" #o6_ " offset1:" #p6_                                     \
                 : [w0] "=&v"(W_.w0), [w1] "=&v"(W_.w1), [w2] "=&v"(W_.w2), [w3] "=&v"(W_.w3), [w4] "=&v"(W_.w4),   \
                   [w5] "=&v"(W_.w5), [w6] "=&v"(W_.w6)                                                             \
                 : [wa] "v"(wa_) : "memory")
#define SYM_WLOAD0(W_, wa_) SYM_WLOAD_(W_, wa_, 0, 1, 2, 3, 4, 5, 6, 41, 40, 39, 38, 37, 36, 35)
#define SYM_WLOAD1(W_, wa_) SYM_WLOAD_(W_, wa_, 7, 8, 9, 10, 11, 12, 13, 34, 33, 32, 31, 30, 29, 28)
#define SYM_WLOAD2(W_, wa_) SYM_WLOAD_(W_, wa_, 14, 15, 16, 17, 18, 19, 20, 27, 26, 25, 24, 23, 22, 21)
#define SYM_WWAIT(W_, n_) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(W_.w0), "+v"(W_.w1), "+v"(W_.w2), "+v"(W_.w3), "+v"(W_.w4), "+v"(W_.w5), "+v"(W_.w6))
// (lo, hi) = (w.hi + w.lo, w.hi - w.lo): the sample tap k meets is the pair's second word
__device__ __forceinline__ float2v sym_sum_diff(float2v w)
{
    float2v r;
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(w));
    return r;
}
template <int NT>
struct SymTaps {
    static constexpr int H = NT / 2;
    static_assert(NT == 42, "three groups of 7 tap pairs");
    float2v tc[H], ts[H];                // Re u_k (mark, space), Im u_k (mark, space)
    __device__ __forceinline__ void load(const float4 *__restrict__ taps)
    {
        sym_static_for<H>([&](auto k_) __attribute__((always_inline)) {
            constexpr int k = decltype(k_)::value;
            const float4 t = taps[NT + k];
            tc[k] = float2v{t.x, t.y}; ts[k] = float2v{t.z, t.w};
            // (wave-uniform values: left to itself the compiler keeps them in scalar registers, runs out of those, and
            // fetches every operand back with v_readlane -- they are vector operands of every product, so vector registers)
            asm volatile("" : "+v"(tc[k]), "+v"(ts[k]));
        });
    }
    // |mark| - |space| clamped to +-1 (rx/demod.rs:156-164) at the instant whose sample sits in ring slot n
    template <int RING>
    __device__ __forceinline__ float demod(uint32_t wcol_lds, int n) const
    {
        int base = n - (NT - 1);
        base += base < 0 ? RING : 0;                          // slots RING .. RING + NT - 2 repeat slots 0 .. NT - 2
        const uint32_t wa = wcol_lds + (uint32_t)base * (kWave * 4u);
        // [parity of k]: two chains per sum; .x mark, .y space
        float2v re[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, im[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
        SymWin X, Y, Z;
        SYM_WLOAD0(X, wa);
        SYM_WLOAD1(Y, wa);
        SYM_WLOAD2(Z, wa);
        auto group = [&](auto g_, const SymWin &W) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            auto pair = [&](auto K_, float2v w) __attribute__((always_inline)) {
                constexpr int k = 7 * g + decltype(K_)::value;
                const float2v sd = sym_sum_diff(w);
                pk_fma_lo(re[k & 1], sd, tc[k]);
                pk_fma_hi(im[k & 1], sd, ts[k]);
            };
            pair(std::integral_constant<int, 0>{}, W.w0); pair(std::integral_constant<int, 1>{}, W.w1);
            pair(std::integral_constant<int, 2>{}, W.w2); pair(std::integral_constant<int, 3>{}, W.w3);
            pair(std::integral_constant<int, 4>{}, W.w4); pair(std::integral_constant<int, 5>{}, W.w5);
            pair(std::integral_constant<int, 6>{}, W.w6);
        };
        SYM_WWAIT(X, 14); group(std::integral_constant<int, 0>{}, X);
        SYM_WWAIT(Y, 7);  group(std::integral_constant<int, 1>{}, Y);
        SYM_WWAIT(Z, 0);  group(std::integral_constant<int, 2>{}, Z);
        const float2v r = re[0] + re[1], i = im[0] + im[1];
        const float2v q = __builtin_elementwise_fma(i, i, r * r);       // (|mark|^2, |space|^2)
        return __builtin_amdgcn_fmed3f(__builtin_amdgcn_sqrtf(q.x) - __builtin_amdgcn_sqrtf(q.y), -1.0f, 1.0f);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// T's sample half: input prefetch and DC blocker, one block ahead of S.  Inputs alternate between two register buffers
// (block b in buffer b & 1); the loads of block b + 1 are issued just before block b is computed, unconditionally (a load
// behind a condition makes the compiler wait for every outstanding load at the reads), so they have a whole
// step to arrive.  CMODE: 1 = channel-major input with per-lane streams, 0 = time-major rows.
// ---------------------------------------------------------------------------------------------------------------------
template <typename SampleT, int CMODE>
struct SymDc {
    static constexpr int B = SymLayout<42>::B, DCL = SymLayout<42>::DCL, RING = SymLayout<42>::RING;
    static constexpr uint32_t LP = kWave;
    // Everything is kept as aligned PAIRS (samples 2 i, 2 i + 1) and every access is a whole pair with a compile-time index:
    // the packed operations want aligned register pairs anyway, and an array that is read as pairs at both even and odd
    // offsets does not survive as registers (see sym_static_for).
    typedef float2v Pairs[B / 2];
    // The DC blocker (rx/dcblock.rs:45-49, 104-108) is two moving averages of DCL = 16 samples:
    //     sum0 += x - x[-16];  ma0 = sum0 / 16;  sum1 += ma0 - ma0[-16];  ma1 = sum1 / 16;  y = x[-15] - ma1.
    // A division by 16 is exact, and adding, subtracting and rounding commute with a scaling by a power of two, so the second
    // average may run on the UNSCALED first sums -- S1 = 16 sum1 exactly, sample for sample -- and y = x[-15] - S1 / 256 is one
    // fused multiply-add whose only rounding is the reference's own (S1 / 256 is exact): the same bits with 3.5 vector
    // instructions per sample instead of 5.  The state arrays keep the reference's scaling (load / store).
    float sum0, sum1;                    // sum1: 16 x the reference's
    float2v xp[DCL / 2], sp[DCL / 2];    // the last DCL inputs / first-stage SUMS, oldest first
    Pairs xa, xb;                        // inputs: block b waits in (b & 1 ? xb : xa)
    uint32_t wpos = 0;                   // ring slot of the block written next
    const SampleT *xl = nullptr;         // CMODE 1: this lane's own stream
    uint32_t avail = 0;                  // ... and the blocks it holds
    bool done = false;                   // ... and whether the lane's piece has handed over: nothing it computes from here on is kept

    __device__ __forceinline__ void request(Pairs &dst, const SampleT *__restrict__ x, uint32_t blk, uint32_t n_blocks, uint32_t cin, uint32_t Cin) const
    {
        if constexpr (CMODE == 1) {
            // (a lane that has handed over stops reading: in grid order the pieces of a workgroup differ in length, and the
            // short ones otherwise read on to the end of the longest -- 0.3 x the algorithmic bytes of a configs[1] launch)
            if (done) return;
            const uint32_t b = min(blk, avail - 1u);                 // (avail >= 1: the planner leaves two scout blocks behind every cut)
            const float4 *p4 = reinterpret_cast<const float4 *>(xl + (size_t)b * B);
            sym_static_for<B / 4>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value;
                const float4 v = p4[j];
                dst[2 * j] = float2v{v.x, v.y}; dst[2 * j + 1] = float2v{v.z, v.w};
            });
        } else {
            // buffer loads: the block's first row is the resource's base (re-based per block: a launch may exceed the 4 GB a
            // resource spans), row k at the scalar offset k * row_bytes, the lane's column as the vector offset -- one
            // instruction per sample and no address arithmetic on the vector unit (36 x 64-bit adds otherwise)
            const SampleT *xr = x + ((size_t)min(blk, n_blocks - 1u) * B) * Cin;      // wave-uniform
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<SampleT *>(xr), 0, 0x7fffffff, 0x00020000);
            const uint32_t voff = cin * (uint32_t)sizeof(SampleT);
            uint32_t row_bytes = Cin * (uint32_t)sizeof(SampleT);
            asm volatile("" : "+s"(row_bytes));      // (opaque: the 36 row offsets are made here, one multiply each, not kept in 36 scalar registers for the whole launch)
            auto one = [&](uint32_t k) __attribute__((always_inline)) -> float {
                if constexpr (sizeof(SampleT) == 4) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, k * row_bytes, 0));
                else return (float)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, k * row_bytes, 0);
            };
            sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value;
                dst[h] = float2v{one(2u * h), one(2u * h + 1u)};
            });
        }
    }
    __device__ __forceinline__ void load(const State &S, const SampleT *__restrict__ x, uint32_t c, uint32_t C, uint32_t cin, uint32_t Cin,
                                         uint64_t counter0, uint32_t n_blocks)
    {
        sum0 = S.dc_sum0[c]; sum1 = S.dc_sum1[c] * (float)DCL;
        const uint32_t dpos = (uint32_t)(counter0 % (uint64_t)DCL);
        sym_static_for<DCL / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            uint32_t s0 = dpos + (uint32_t)(2 * h), s1 = s0 + 1u;
            if (s0 >= (uint32_t)DCL) s0 -= (uint32_t)DCL;
            if (s1 >= (uint32_t)DCL) s1 -= (uint32_t)DCL;
            xp[h] = float2v{(S.dc_ff_ring + (size_t)s0 * C)[c], (S.dc_ff_ring + (size_t)s1 * C)[c]};
            sp[h] = float2v{(S.dc_fb_ring + (size_t)s0 * C)[c] * (float)DCL, (S.dc_fb_ring + (size_t)s1 * C)[c] * (float)DCL};
        });
        sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) { xb[decltype(h_)::value] = float2v{0.0f, 0.0f}; });
        request(xa, x, 0u, n_blocks, cin, Cin);
    }
    // DC blocker of block `blk`, whose inputs are X; outputs into the window ring's next block (S turns them into AGC
    // outputs in place, one step later)
    __device__ __forceinline__ void block(const Params &P, float *wcol, Pairs &X, uint32_t blk)
    {
        if constexpr (CMODE == 1) {
            // per lane: its stream ends where the input does, and silence follows it (the last steps of a launch only)
            const bool live = blk < avail;
            if (__builtin_amdgcn_ballot_w64(!live) != 0ull) {
                sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
                    constexpr int h = decltype(h_)::value;
                    X[h] = float2v{live ? X[h].x : 0.0f, live ? X[h].y : 0.0f};
                });
            }
        }
        float *y = wcol + wpos * LP;
        wpos += (uint32_t)B;
        if (wpos == (uint32_t)RING) wpos = 0u;
        Pairs snew;
        constexpr float kScale = -1.0f / (float)(DCL * DCL);
        static_assert(DCL == 16, "a power of two: the scalings above are exact");
        const float2v nscale = {kScale, kScale};
        float s1_last = 0.0f;                                          // S1 after the odd sample of the pair before
        sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;                     // samples k = 2 h, 2 h + 1
            // the windows before this block, oldest first, as pairs: 0 .. DCL/2 - 1 the history, then this block's
            auto xw = [&](auto i_) __attribute__((always_inline)) -> float2v {
                constexpr int i = decltype(i_)::value;
                if constexpr (i < DCL / 2) return xp[i]; else return X[i - DCL / 2];
            };
            auto sw = [&](auto i_) __attribute__((always_inline)) -> float2v {
                constexpr int i = decltype(i_)::value;
                if constexpr (i < DCL / 2) return sp[i]; else return snew[i - DCL / 2];
            };
            const float2v xo = xw(std::integral_constant<int, h>{});   // inputs 2 h - 16, 2 h - 15
            const float2v d0 = X[h] - xo;
            const float s0a = sum0 + d0.x, s0b = s0a + d0.y;
            sum0 = s0b;
            const float2v s0 = {s0a, s0b};
            snew[h] = s0;
            const float2v d1 = s0 - sw(std::integral_constant<int, h>{});
            const float s1a = sum1 + d1.x, s1b = s1a + d1.y;
            sum1 = s1b;
            // y[k] = x[k - 15] - S1[k] / 256: the pair (y[2 h - 1], y[2 h]) meets the ALIGNED input pair xo
            if constexpr (h == 0) {
                y[0] = __builtin_fmaf(s1a, kScale, xo.y);
            } else {
                const float2v yy = __builtin_elementwise_fma(float2v{s1_last, s1a}, nscale, xo);
                y[(2 * h - 1) * LP] = yy.x; y[(2 * h) * LP] = yy.y;
            }
            s1_last = s1b;
            if constexpr (h == B / 2 - 1) y[(B - 1) * LP] = __builtin_fmaf(s1b, kScale, xw(std::integral_constant<int, h + 1>{}).x);
        });
        sym_static_for<DCL / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            xp[h] = X[(B - DCL) / 2 + h]; sp[h] = snew[(B - DCL) / 2 + h];
        });
    }
    __device__ __forceinline__ void store(const State &S, uint32_t c, uint32_t C, uint64_t counter1)
    {
        constexpr float inv = 1.0f / (float)DCL;
        S.dc_sum0[c] = sum0; S.dc_sum1[c] = sum1 * inv;
        const uint32_t dpos = (uint32_t)(counter1 % (uint64_t)DCL);
        sym_static_for<DCL / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            uint32_t s0 = dpos + (uint32_t)(2 * h), s1 = s0 + 1u;
            if (s0 >= (uint32_t)DCL) s0 -= (uint32_t)DCL;
            if (s1 >= (uint32_t)DCL) s1 -= (uint32_t)DCL;
            (S.dc_ff_ring + (size_t)s0 * C)[c] = xp[h].x; (S.dc_ff_ring + (size_t)s1 * C)[c] = xp[h].y;
            (S.dc_fb_ring + (size_t)s0 * C)[c] = sp[h].x * inv; (S.dc_fb_ring + (size_t)s1 * C)[c] = sp[h].y * inv;
        });
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// S: AGC over the window ring's newest block, in place (T left the DC blocker's outputs there), and the gain a lock freezes
// ---------------------------------------------------------------------------------------------------------------------
struct SymAgc {
    static constexpr int B = SymLayout<42>::B, RING = SymLayout<42>::RING, MIR = SymLayout<42>::MIR;
    static constexpr uint32_t LP = kWave;
    float gain;
    bool locked;                         // this wavefront's belief of the AGC lock
    float g0a, g0b, g0c;                 // the AGC gain the last three blocks started with: a newest (scalars: an array indexed by a lane's block ends up in scratch memory)
    uint32_t wnext;                      // ring position of the block processed next
    uint32_t last_blk;                   // the block g0a belongs to

    __device__ __forceinline__ void load(const Params &P, const State &S, uint32_t c, uint32_t C, uint64_t counter0, float *wcol)
    {
        // the window the last launch left: sample counter0 - m sits in the state's slot (counter0 - m) mod win_ring; the
        // launch's first sample goes to ring slot 0, so it belongs in slot RING - m (never one of the mirrored slots, nor one
        // of the two blocks T writes before the filters first run)
        const uint32_t G = P.win_ring;
#pragma unroll 2
        for (uint32_t m = 1; m <= G; ++m) {
            const uint32_t g = (uint32_t)(counter0 - (uint64_t)m) & (G - 1u);
            const float *row = S.win_ring + (size_t)g * C;
            if (m <= (uint32_t)(RING - 2 * B)) wcol[((uint32_t)RING - m) * LP] = row[c];
        }
        gain = S.agc_gain[c];
        locked = (S.flags[c] & F_AGC_LOCKED) != 0u;
        g0a = gain; g0b = gain; g0c = gain;
        wnext = 0; last_blk = 0;
    }

    // AGC (rx/agc.rs:72-77, relaxed: same_relaxed_common.h agc_step_relaxed) of block `blk`, whose DC-blocker outputs wait in
    // the ring's block at wnext; the AGC outputs replace them
    __device__ __forceinline__ void block(const Params &P, float *wcol, uint32_t blk)
    {
        float *wblk = wcol + wnext * LP;
        float2v yv[B / 2];
#pragma unroll
        for (int h = 0; h < B / 2; ++h) yv[h] = float2v{wblk[(2 * h) * LP], wblk[(2 * h + 1) * LP]};
        g0c = g0b; g0b = g0a;
        g0a = gain; last_blk = blk;
        const float bw = locked ? 0.0f : P.agc_bw;
        float2v ov[B / 2];
#pragma unroll
        for (int h = 0; h < B / 2; ++h) {
            // out = y * gain;  gain <- clamp(gain (1 - bw |y|) + bw): two operations on the gain's chain per sample, the
            // two products of a pair as one packed multiply
            const float a0 = __builtin_fmaf(-bw, fabsf(yv[h].x), 1.0f), a1 = __builtin_fmaf(-bw, fabsf(yv[h].y), 1.0f);
            const float g1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(gain, a0, bw), P.agc_min, P.agc_max);
            const float g2 = __builtin_amdgcn_fmed3f(__builtin_fmaf(g1, a1, bw), P.agc_min, P.agc_max);
            ov[h] = yv[h] * float2v{gain, g1};
            gain = g2;
            wblk[(2 * h) * LP] = ov[h].x; wblk[(2 * h + 1) * LP] = ov[h].y;
        }
        // ring slots 0 .. MIR - 1 (the first block and the head of the second) once more behind the ring
        if (wnext == 0u) {                                             // wave-uniform
#pragma unroll
            for (int h = 0; h < B / 2; ++h) { wblk[(2 * h + RING) * LP] = ov[h].x; wblk[(2 * h + 1 + RING) * LP] = ov[h].y; }
        } else if (wnext == (uint32_t)B) {
#pragma unroll
            for (int k = 0; k < MIR - B; ++k) wblk[(k + RING) * LP] = (k & 1) ? ov[k / 2].y : ov[k / 2].x;
        }
        wnext += B;
        if (wnext == (uint32_t)RING) wnext = 0;
    }

    // The AGC gain after sample `fk` of block `b` (b <= last_blk, this lane's AGC unlocked since then): the reference's own
    // recurrence, gain += bw (1 - |out|) (rx/agc.rs:72-77), once more from the block's start over the AGC OUTPUTS, which are
    // what the window ring holds -- f32, whatever the input's scale, and no history of its own (round 4 kept three blocks of
    // DC-blocker outputs as packed f16 in 54 registers, rotated through 64 moves a step, for this).  Further back than three
    // blocks: the oldest one's start.  Rare (a lock: once per burst and lane), so a loop.
    __device__ __forceinline__ float gain_at(const Params &P, const float *wcol, uint32_t b, int fk) const
    { return gain_at_(P, wcol, wnext, last_blk - b, g0a, g0b, g0c, fk); }
    // (the three gains BY VALUE: selected through the struct's members the compiler selects an address, and the whole struct
    // stays in scratch memory)
    static __device__ __forceinline__ float gain_at_(const Params &P, const float *wcol, uint32_t wnext_, uint32_t j, float ga, float gb, float gc, int fk)
    {
        float g = j == 0u ? ga : (j == 1u ? gb : gc);
        if (j > 2u) return g;
        // block last_blk sits B slots before wnext
        int pos = (int)wnext_ - (int)(j + 1u) * B;
        pos += pos < 0 ? RING : 0;
        const float *w = wcol + (uint32_t)pos * LP;
        const float bw = P.agc_bw;
#pragma unroll 4
        for (int k = 0; k < B; ++k) {
            const float o = w[k * LP];
            const float g1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(bw, 1.0f - fabsf(o), g), P.agc_min, P.agc_max);
            g = (k <= fk) ? g1 : g;
        }
        return g;
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
            row[c] = wcol[j * LP];
        }
    }
};

// Y's context: RelaxFastCtx (equalizer and its snapshot in registers, relaxed equalizer step) with the squelch's sample
// history in global memory.  The two samples the NEXT symbol's equalizer step takes (rx_symbol: slots +16 / +17 from the
// squelch's write position, i.e. +18 / +19 from this symbol's) are requested where this symbol's samples are stored: a
// global round trip is a step long under load, and on gfx950 loads and stores retire in one queue (vmcnt) -- requested
// at the end of the step they would also wait for the framer's scattered byte stores issued before them.  Only this lane's
// own symbols write its history, and never those two slots before they are read: they cannot go stale.
template <int NFF, int NFB>
struct SymCtx : RelaxFastCtx<NFF, NFB> {
    float nxt0 = 0.0f, nxt1 = 0.0f;
    __device__ __forceinline__ void hist_put(uint32_t slot, float v)
    {
        if ((slot & 1u) == 0u) {
            nxt0 = this->hist[(size_t)((slot + 18u) & 63u) * this->hstride];
            nxt1 = this->hist[(size_t)((slot + 19u) & 63u) * this->hstride];
        }
        this->hist[(size_t)slot * this->hstride] = v;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
template <int NFF, int NFB, typename SampleT, int CMODE>
__global__ __launch_bounds__(4 * kWave, 2) void demod_sym_kernel(Params P, State S, Output O, const float4 *__restrict__ taps,
                                                                 const SampleT *__restrict__ x, uint32_t n_blocks, uint64_t counter0,
                                                                 PipeChunks K)
{
    constexpr int NT = 42;
    using LY = SymLayout<NT>;
    constexpr int kB = LY::B, RING = LY::RING;
    constexpr uint32_t LP = kWave;
    static_assert(CMODE == 0 || std::is_same<SampleT, float>::value, "channel-major streams are f32");
    if constexpr (CMODE == 0) { K.col_row0 = nullptr; K.col_perm = nullptr; }       // (the host launches this build for nothing else)
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x & (kWave - 1u);
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0 S, 1 T, 2 Y, 3 E
    const uint32_t C = P.n_channels;
    // state column of this lane (time-parallel launches may permute them: pieces of similar length share a workgroup)
    const uint32_t c = (K.n_chunks > 1u && K.col_perm) ? K.col_perm[blockIdx.x * kWave + lane] : blockIdx.x * kWave + lane;
    // Time-parallel chunks, exactly as demod_pipe_kernel takes them
    uint32_t cin = c, Cin = C, n_nominal = n_blocks;
    bool may_leave = false;
    int32_t row_l = 0;
    const SampleT *xl = nullptr;
    uint32_t avail_l = 0;
    if (K.n_chunks > 1u) {
        Cin = K.in_channels;
        if (K.col_row0) {
            const uint32_t chunk_l = c / Cin;
            cin = c - chunk_l * Cin;
            may_leave = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk_l) + 1u < K.n_chunks;
            const uint32_t row_abs = K.col_row0[c];
            xl = x + (size_t)cin * K.in_samples + row_abs;
            avail_l = (K.whole_samples - row_abs) / (uint32_t)kB;
            n_blocks = K.wg_blocks[blockIdx.x];
            n_nominal = may_leave ? K.col_nominal[c] : n_blocks;
            const uint32_t row_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_abs);
            counter0 += (uint64_t)row_first;
            row_l = (int32_t)(row_abs - row_first);
        } else {
            const uint32_t wgs = K.in_channels / kWave;
            const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x / wgs));
            cin = (blockIdx.x - chunk * wgs) * kWave + lane;
            may_leave = chunk + 1u < K.n_chunks;
            const uint32_t first_block = chunk * K.stride_blocks;
            x += (size_t)first_block * kB * Cin;
            counter0 += (uint64_t)first_block * kB;
            n_blocks -= first_block;
            n_nominal = may_leave ? K.nominal_blocks : n_blocks;
        }
    }
    // LDS: taps | mailboxes | window ring [RING + MIR][64]
    float4 *tlds = reinterpret_cast<float4 *>(lds);
    lds_u32 *mail = (lds_u32 *)(lds + LY::tap_floats);
    lds_u32 *symbox = mail;                                    // [2][5][64]
    lds_u32 *fbbox = mail + 2u * LY::sym_words;                // [2][64 + flag]
    lds_u32 *iobox = fbbox + 2u * LY::fb_words;                // [2][3][64]
    lds_u32 *phasebox = iobox + 2u * LY::io_words;             // [64] T's final TED phase bit
    lds_u32 *againbox = phasebox + kWave;                      // [64] E's final F_TICK_AGAIN bit
    lds_u32 *posbox = againbox + kWave;                        // [2][2][64] ring slots of the two instants of step s's symbol (parity s & 1)
    lds_u32 *sabox = posbox + 2u * LY::pos_words;              // [64] the first instant's soft sample
    lds_u32 *tkbox = sabox + kWave;                            // [kTickRing][64] u64 deadlines, then [64] their count (T's own)
    lds_u32 *donebox = tkbox + kIoRingWords;                   // [64] Y -> T: the lane's piece has handed over (its input is no longer needed)
    lds_u32 *chunkbox = fbbox + kWave + 2u;                    // [2], in the first feedback box's padding
    lds_u32 *seqbox = fbbox + kWave + 4u;                      // S's progress with the first instants' filters: 2 * step + pass
    float *wring = lds + LY::tap_floats + LY::mail_words;     // ring slot 0
    float *wcol = wring + lane;
    const uint64_t counter1 = counter0 + (uint64_t)n_blocks * kB;
    // Steps: T computes the DC blocker of block s + 1 in step s (block 0 before the first), S the AGC of block s in step
    // s < n_blocks; T finishes symbols that end before sample 36 s in steps 1 .. n_blocks and the instants left before the
    // end of the input, one per step, in the kSymDrain steps after; Y runs one step behind T, E's events one behind Y.
    const uint32_t last_t_step = n_blocks + kSymDrain;
    const uint32_t last_fb_step = last_t_step + 1u;            // Y runs in steps 2 .. last_t_step + 1
    const uint32_t n_steps = last_t_step + 3u;
    // A symbol travels as `off` = its sample index - 36 * base, base = s_T - 2 for the step s_T <= n_blocks T finished it
    // in, n_blocks - 3 in the steps after: 0 <= off < 128.  Y (one step later) and E (two) rebuild the index.
    auto sym_index = [&](uint32_t s_t, uint32_t off) __attribute__((always_inline)) -> int64_t {
        const int64_t base = s_t <= n_blocks ? (int64_t)s_t - 2 : (int64_t)n_blocks - 3;
        return (int64_t)kB * base + (int64_t)off;
    };

    if (role == 0u) {
        // ------------------------------------------ S: AGC + window push, block s -------------------------------------
        P3_HWID(0);
        SymAgc M;
        M.load(P, S, c, C, counter0, wcol);
        const uint32_t wcol_lds = lds_addr(wcol);
        SymTaps<NT> TP;
        TP.load(taps);
        if (lane == 0u) seqbox[0] = 0u;
        // the matched filters of the FIRST instant of the step's symbol, at the positions E posted (E takes the second)
        auto filter_a = [&](uint32_t s, uint32_t seq) __attribute__((always_inline)) {
            const uint32_t n1 = posbox[(s & 1u) * LY::pos_words + lane];
            if (__builtin_amdgcn_ballot_w64(n1 != 0xffffffffu) == 0ull) return;
            // (a profile build's knock-out skips the filter, never the hand-over E waits for)
            const float sa1 = PROF_SKIP(P, 256) ? 0.0f : TP.template demod<RING>(wcol_lds, n1 == 0xffffffffu ? 0 : (int)n1);
            sabox[lane] = __float_as_uint(sa1);
            if (lane == 0u) seqbox[0] = seq;                           // (LDS operations of a wavefront stay in order)
        };
        lds_barrier();                                                 // prologue: block 0's DC outputs are in the ring
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        bool left = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            if (s >= 1u && s <= last_t_step) filter_a(s, 2u * s + 1u);
            if (s < n_blocks && !PROF_SKIP(P, 64)) M.block(P, wcol, s);
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (s >= 2u && s <= last_fb_step) {
                const lds_u32 *fb = fbbox + (s & 1u) * LY::fb_words;
                const uint32_t fbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb[kWave]);
                if (fbw & 2u) stop_at = s + 1u;
                if (fbw & 4u) {
                    // a lock or an unlock: from the next block on
                    const uint32_t v = fb[lane];
                    if (v & 1u) {
                        const bool new_locked = (v & 2u) != 0u;
                        if (new_locked && !M.locked) {
                            // the gain freezes at the value it had after the symbol's sample
                            const int64_t idx = sym_index(s - 1u, v >> 8);
                            const uint32_t b = (uint32_t)(idx / kB);
                            M.gain = M.gain_at(P, wcol, b, (int)(idx - (int64_t)b * kB));
                        }
                        M.locked = new_locked;
                    }
                    P3_LAP(p3_fb);
                }
            }
            if (s == stop_at) { left = true; break; }
        }
        SYM_REPORT(0);
        if (left) return;                                              // handed over: this chunk's state is not needed
        lds_barrier();                                                 // (E -> Y: final TED phase)
        M.store(P, S, c, C, counter1, wcol);
    } else if (role == 1u) {
        // ------------------------------------------ T: input prefetch and DC blocker of block s + 1; link events + wake-ups ----
        P3_HWID(1);
        SymDc<SampleT, CMODE> D;
        D.xl = xl; D.avail = avail_l;
        D.load(S, x, c, C, cin, Cin, counter0, n_blocks);
        Lane L;
        lane_load(L, S, c);      // the event half uses sq_symbols, tk_next, tk_last, wake_*, F_TICK_AGAIN
        IoCtxLds X;
        X.chunk = chunkbox;
        chunkbox[0] = 0u; chunkbox[1] = kEvChunk;       // nothing reserved yet
        X.pending_slot = 0xffffffffu;
        X.tk = tkbox + lane;
        X.ring_load(P, S, c);
        donebox[lane] = 0u;
        // prologue: block 0's DC outputs
        if (n_blocks > 1u) D.request(D.xb, x, 1u, n_blocks, cin, Cin);
        D.block(P, wcol, D.xa, 0u);
        lds_barrier();
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        auto step = [&](uint32_t s, auto buf) __attribute__((always_inline)) -> bool {
            constexpr int BUF = decltype(buf)::value;                  // block s + 1 waits in buffer BUF = (s + 1) & 1
            if constexpr (CMODE == 1) D.done = donebox[lane] != 0u;
            if (s + 1u < n_blocks && !PROF_SKIP(P, 128)) {
                if constexpr (BUF == 0) { D.request(D.xb, x, s + 2u, n_blocks, cin, Cin); D.block(P, wcol, D.xa, s + 1u); }
                else { D.request(D.xa, x, s + 2u, n_blocks, cin, Cin); D.block(P, wcol, D.xb, s + 1u); }
            }
            // the link event and the wake-ups of what Y handed over in the last step
            if (s >= 3u && !PROF_SKIP(P, 8)) {
                const lds_u32 *io = iobox + ((s - 1u) & 1u) * LY::io_words + lane;
                const uint32_t io0 = io[0];
                if (io0 & 1u) {
                    L.sq_symbols += 1;         // as rx_symbol counted it (rx/codesquelch.rs:232)
                    const uint32_t link = (io0 >> 1) & 3u, off = (io0 >> 4) & 127u;
                    const bool burst = (io0 & 8u) != 0u && link == 3u;
                    uint32_t burst_len = 0;
                    if (burst) { X.pending_slot = io[kWave]; burst_len = io[2 * kWave]; }
                    const uint64_t counter = counter0 + (int64_t)row_l + (uint64_t)sym_index(s - 2u, off) + 1u;
                    symbol_io(P, L, S, O, X, c, link, (io0 & 8u) != 0u, counter, burst_len);
                }
            }
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (s >= 2u && s <= last_fb_step) {
                const lds_u32 *fb = fbbox + (s & 1u) * LY::fb_words;
                const uint32_t fbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb[kWave]);
                if (fbw & 2u) stop_at = s + 1u;
            }
            return s == stop_at;
        };
        bool left = false;
        for (uint32_t s = 0; s < n_steps && !left; s += 2u) {
            left = step(s, std::integral_constant<int, 1>{});          // block s + 1 is odd when s is even
            if (!left && s + 1u < n_steps) left = step(s + 1u, std::integral_constant<int, 0>{});
        }
        SYM_REPORT(1);
        X.retire(O, lane, kWave);
        if (left) return;
        againbox[lane] = L.flags & F_TICK_AGAIN;
        lds_barrier();                                                 // Y merges the flag bits
        D.store(S, c, C, counter1);
        S.tk_next[c] = L.tk_next; S.tk_last[c] = L.tk_last; S.wake_fired[c] = L.wake_fired;
        X.ring_store(P, S, c);
    } else if (role == 2u) {
        // ------------------------------------------ Y: symbol path --------------------------------------------------
        P3_HWID(2);
        Lane L;
        lane_load(L, S, c);
        L.ended = 0u;
        SymCtx<NFF, NFB> X;
        // the squelch's sample history stays in global memory: the state array itself, or -- where the columns of a wavefront
        // are permuted -- a copy by grid position (coalesced; the state array is read and written once per launch)
        const bool hist_copy = K.n_chunks > 1u && K.col_perm != nullptr && K.hist_scratch != nullptr;
        X.hist = hist_copy ? K.hist_scratch + (blockIdx.x * kWave + lane) : S.sq_hist + c;
        X.hstride = C;
        if (hist_copy) {
#pragma unroll 4
            for (int i = 0; i < kSquelchHist; ++i) X.hist[(size_t)i * C] = S.sq_hist[(size_t)i * C + c];
        }
        P3_MARKS_BEGIN(X, lds, NT);
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
        {
            const uint32_t pslot = (uint32_t)(2u * (uint32_t)L.sq_symbols) & 63u;
            X.nxt0 = X.hist_get((pslot + 16u) & 63u); X.nxt1 = X.hist_get((pslot + 17u) & 63u);
        }
        lds_barrier();                                                 // prologue
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        bool left = false, lane_done = false, leave_posted = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            if (s >= 2u && s <= last_fb_step) {
                const uint32_t blk = min(s - 2u, n_blocks - 1u);       // every symbol up to the end of this block has been seen after this step
                const lds_u32 *sb = symbox + ((s - 1u) & 1u) * LY::sym_words + lane;
                // (the two history samples the symbol's equalizer step takes were requested with the lane's last symbol: SymCtx)
                const float pre0 = X.nxt0, pre1 = X.nxt1;
                const uint32_t hdr = sb[0], zero_w = sb[kWave], sym_w = sb[2 * kWave];
                uint32_t fbv = 0, io0 = 0, io1 = 0xffffffffu, io2 = 0;
                bool want_slot = false;                                // this lane has just finished a burst
                if ((hdr & 1u) && !PROF_SKIP(P, 16)) {
                    const uint32_t off = hdr >> 8;
                    const float zero = __uint_as_float(zero_w), sym = __uint_as_float(sym_w);
                    float terr = 0.0f, unt = 0.0f;
                    if (P.trace_cap) { terr = __uint_as_float(sb[3 * kWave]); unt = __uint_as_float(sb[4 * kWave]); }
                    const uint32_t before = L.flags & (F_AGC_LOCKED | F_BW_LOCKED);
                    L.ended = 0u;
                    uint32_t burst_len = 0;
                    bool emit = false;
                    // the counter is that of the sample after the symbol's
                    const uint64_t counter = counter0 + (int64_t)row_l + (uint64_t)sym_index(s - 1u, off) + 1u;
                    const uint32_t link = symbol_link(P, L, S, X, c, zero, sym, terr, unt, counter, &burst_len, &emit, true, pre0, pre1);
                    want_slot = emit && link == 3u;
                    io0 = 1u | (link << 1) | (emit ? 8u : 0u) | (off << 4);
                    io2 = burst_len;
                    const uint32_t after = L.flags & (F_AGC_LOCKED | F_BW_LOCKED);
                    if (after != before || L.ended) {
                        // The lock at sync (agc.lock(true), locked loop bandwidth: receiver.rs:431-432) and what end() undoes
                        // (receiver.rs:479-490) reach S and E late: the AGC freezes -- at the gain of this symbol's sample -- or
                        // is released from S's next block on, the timing loop follows two symbols later.  Link events after
                        // an end() move by less than a symbol (the carrier is gone by then).
                        fbv = 1u | ((after & F_AGC_LOCKED) ? 2u : 0u) | ((after & F_BW_LOCKED) ? 4u : 0u) | (L.ended ? 8u : 0u) | (off << 8);
                    }
                }
                // finished bursts go into the pool with the whole wavefront: one slot reservation for all of them and one
                // coalesced round trip per burst (same_kernels_pipe.hip)
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
                            constexpr uint32_t kWords = (uint32_t)kBurstCap / 4u, kPer = (kWords + kWave - 1u) / kWave;
                            uint32_t t[kPer];
#pragma unroll
                            for (uint32_t i = 0; i < kPer; ++i) { const uint32_t w = lane + i * kWave; t[i] = w < kWords ? src[w] : 0u; }
#pragma unroll
                            for (uint32_t i = 0; i < kPer; ++i) { const uint32_t w = lane + i * kWave; if (w < kWords) dst[w] = t[i]; }
                            if ((int)lane == j) io1 = b;
                        } else {
                            if (lane == 0u) atomicOr(O.n_events + 2, 2u);
                            if ((int)lane == j) io1 = 0xffffffffu;
                        }
                    }
                }
                lds_u32 *io = iobox + (s & 1u) * LY::io_words + lane;
                io[0] = io0;
                if (__builtin_amdgcn_ballot_w64((io0 & 8u) != 0u && ((io0 >> 1) & 3u) == 3u) != 0ull) { io[kWave] = io1; io[2 * kWave] = io2; }
                lds_u32 *fb = fbbox + (s & 1u) * LY::fb_words;
                fb[lane] = fbv;
                const bool any_late = __builtin_amdgcn_ballot_w64(fbv != 0u) != 0ull;
                // Time-parallel chunk that hands over: from its nominal end on, a lane's hand-over instant is
                // the end of the first block after which its link state is NoCarrier; once every lane has one the workgroup
                // leaves (one more step: E still has to log this step's events)
                uint32_t leave = 0u;
                if (may_leave && !leave_posted) {
                    if (!lane_done && blk + 1u >= n_nominal && (L.flags & F_LINK_MASK) == 0u && (xl == nullptr || blk < avail_l)) {
                        lane_done = true;
                        K.handover[c] = counter0 + (int64_t)row_l + (uint64_t)(blk + 1u) * kB;
                        donebox[lane] = 1u;
                    }
                    if (__builtin_amdgcn_ballot_w64(!lane_done) == 0ull) { leave = 2u; leave_posted = true; stop_at = s + 1u; }
                }
                if (lane == 0u) fb[kWave] = leave | (any_late ? 4u : 0u);
            }
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            if (s == stop_at) { left = true; break; }
        }
        SYM_REPORT(2);
        SYM_COUNT(18, 1);                                              // launches of workgroup 0 ...
        SYM_COUNT(19, left ? stop_at + 1u : n_steps);                  // ... and the steps they ran
        P3_MARKS_REPORT(X);
        if (left) return;
        lds_barrier();                                                 // T's TED phase, E's wake-up flag
        L.flags = (L.flags & ~(F_TED_PHASE | F_TICK_AGAIN)) | (phasebox[lane] & F_TED_PHASE) | (againbox[lane] & F_TICK_AGAIN);
        S.sq_data[c] = L.sq_data; S.sq_power[c] = L.sq_power; S.sq_phist[c] = L.sq_phist;
        S.sq_fill[c] = L.sq_fill; S.sq_clock[c] = L.sq_clock; S.sq_symbols[c] = L.sq_symbols;
        S.eq_word[c] = L.eq_word; S.eq_count[c] = L.eq_count;
        S.fr_word[c] = L.fr_word; S.fr_count[c] = L.fr_count; S.fr_invalid[c] = L.fr_invalid;
        S.fr_len[c] = L.fr_len; S.flags[c] = L.flags;
        if (hist_copy) {
#pragma unroll 4
            for (int i = 0; i < kSquelchHist; ++i) S.sq_hist[(size_t)i * C + c] = X.hist[(size_t)i * C];
        }
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
    } else {
        // ------------------------------------------ E: one symbol per lane and step: matched filters, timing loop --------
        P3_HWID(3);
        const uint32_t wcol_lds = lds_addr(wcol);
        SymTaps<NT> TP;
        TP.load(taps);
        Lane L;
        lane_load(L, S, c);
        const float inv_spt = 1.0f / P.samples_per_ted;
        int cstar = next_fire_count(L.until_next_ted, L.ted_clock);
        int rel = cstar - (int)L.ted_clock - 1;        // index of the next instant, relative to the end of the finished samples
        // The plan of step s, made at the end of step s - 1: per lane the next symbol -- instant A
        // (completes nothing: where B falls does not depend on A's sample) and instant B, or B alone right after a
        // symsync.reset -- if both lie in finished samples; in the steps after the last block: one instant whatever it is.
        // S filters at A (where there is one), this wavefront at B.
        SYM_T_DECL();
        bool pl_typeA = false, pl_ready = false, pl_single = false;
        int pl_p2 = 0, pl_c2 = 0;
        uint32_t pl_n2 = 0;
        float pl_rem1 = 0.0f, pl_instA = 0.0f;
        auto plan = [&](uint32_t s) __attribute__((always_inline)) {
            pl_single = s > n_blocks;
            const uint32_t wb = (min(s, n_blocks) % (uint32_t)LY::NBLK) * (uint32_t)kB;      // ring slot of the sample at rel 0
            pl_typeA = !pl_single && (L.flags & F_TED_PHASE) != 0u;
            pl_rem1 = L.until_next_ted - (float)cstar;                                 // receiver.rs:352
            pl_instA = L.period_inst + __builtin_amdgcn_fmed3f(pl_rem1, -0.5f, 0.5f);   // rx/symsync.rs:236-241
            pl_c2 = next_fire_count(pl_instA, 0u);
            pl_p2 = pl_typeA ? rel + pl_c2 : rel;
            pl_ready = pl_p2 < 0 && s >= 1u && s <= last_t_step;
            auto slot = [&](int p) __attribute__((always_inline)) { int n = (int)wb + p; n += n < 0 ? RING : 0; return (uint32_t)n; };
            pl_n2 = pl_ready ? slot(pl_p2) : 0u;
            posbox[(s & 1u) * LY::pos_words + lane] = (pl_ready && pl_typeA) ? slot(rel) : 0xffffffffu;
        };
        auto work = [&](uint32_t s, uint32_t seq) __attribute__((always_inline)) {
            uint32_t hdr = 0;
            float zero = 0.0f, sym = 0.0f, terr = 0.0f, next = 0.0f;
            if (__builtin_amdgcn_ballot_w64(pl_ready) != 0ull) {
                SYM_T_BEGIN();
                SYM_TCOUNT(13, 1);
                const float sa2 = PROF_SKIP(P, 512) ? 0.0f : TP.template demod<RING>(wcol_lds, (int)pl_n2);
                SYM_T_LAP(15);
                float sa1 = 0.0f;
                if (__builtin_amdgcn_ballot_w64(pl_ready && pl_typeA) != 0ull) {
                    while ((int32_t)(seqbox[0] - seq) < 0) {}           // S has posted this pass
                    sa1 = __uint_as_float(sabox[lane]);
                }
                SYM_T_LAP(17);
                if (pl_ready) {
                    float z, sy, te;
                    bool have;
                    if (pl_typeA) {
                        // ZeroCrossingTed::input + TimingLoop::input without a symbol, rx/symsync.rs:236-241, 278-287
                        L.h0 = L.h1; L.h1 = L.h2; L.h2 = sa1;
                        L.flags ^= F_TED_PHASE;
                        L.period_inst = pl_instA; L.until_next_ted = pl_instA;
                        have = ted_timing_relaxed(P, L, inv_spt, sa2, pl_instA - (float)pl_c2, &z, &sy, &te);
                    } else {
                        have = ted_timing_relaxed(P, L, inv_spt, sa2, pl_rem1, &z, &sy, &te);
                    }
                    cstar = next_fire_count(L.until_next_ted, 0u);
                    rel = pl_p2 + cstar;
                    if (have) { hdr = 1u | ((uint32_t)(pl_p2 + (pl_single ? 3 * kB : 2 * kB)) << 8); zero = z; sym = sy; terr = te; next = L.until_next_ted; }
                }
                SYM_T_LAP(16);
            }
            lds_u32 *sb = symbox + (s & 1u) * LY::sym_words + lane;
            sb[0] = hdr;
            sb[kWave] = __float_as_uint(zero); sb[2 * kWave] = __float_as_uint(sym);
            if (P.trace_cap) { sb[3 * kWave] = __float_as_uint(terr); sb[4 * kWave] = __float_as_uint(next); }
        };
        // the loop bandwidth of a lock at sync (receiver.rs:431-432) and what end() undoes (receiver.rs:479-490: unlocked
        // loop bandwidth, symsync.reset()), late: see Y
        uint32_t late_now = 0u;
        auto late = [&](uint32_t v) __attribute__((always_inline)) {
            if (v & 1u) {
                L.flags = (L.flags & ~F_BW_LOCKED) | ((v & 4u) ? F_BW_LOCKED : 0u);
                if (v & 8u) {                                            // rx/symsync.rs:166-170, 265-271
                    L.flags &= ~F_TED_PHASE;
                    L.h0 = 0.0f; L.h1 = 0.0f; L.h2 = 0.0f;
                    L.period_avg = P.samples_per_ted; L.period_inst = P.samples_per_ted;
                }
            }
        };
        plan(0u);                                                      // (nothing: step 0 finishes no symbol)
        lds_barrier();                                                 // prologue
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        bool left = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            if (!PROF_SKIP(P, 32)) {
                work(s, 2u * s + 1u);
                late(late_now);
                if (s + 1u <= n_blocks) rel -= kB;                     // block s is finished when step s + 1 begins
                plan(s + 1u);
            }
            P3_LAP(p3_work);
            lds_barrier();                                             // A
            P3_LAP(p3_wait);
            uint32_t late_next = 0u;
            if (s >= 2u && s <= last_fb_step) {
                const lds_u32 *fb = fbbox + (s & 1u) * LY::fb_words;
                const uint32_t fbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb[kWave]);
                if (fbw & 2u) stop_at = s + 1u;
                if (fbw & 4u) {
                    // a change of the loop bandwidth or a symsync.reset(): applied behind the next step's symbol (the
                    // positions of that step are already with S)
                    const uint32_t v = fb[lane];
                    if (v & 1u) late_next = v;
                }
            }
            late_now = late_next;
            if (s == stop_at) { left = true; break; }
        }
        late(late_now);                                                // (one that arrived with the last step)
        SYM_REPORT(3);
        SYM_T_REPORT();
        SYM_COUNT(12, n_steps);
        if (left) return;
        phasebox[lane] = L.flags & F_TED_PHASE;
        lds_barrier();                                                 // Y merges the phase bit
        L.ted_clock = (uint32_t)(cstar - rel - 1);
        S.until_next_ted[c] = L.until_next_ted; S.ted_clock[c] = L.ted_clock;
        S.ted_h0[c] = L.h0; S.ted_h1[c] = L.h1; S.ted_h2[c] = L.h2;
        S.period_avg[c] = L.period_avg; S.period_inst[c] = L.period_inst;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------------------------------
// 22.05 kHz with the reference's default DC-blocker length, the default or the disabled equalizer, a non-negative AGC
// floor, whole groups of 64 state columns, and a timing loop whose shortest symbol is longer than a step (two instants at
// least max_block_len + 1 = 19 samples apart each)
bool sym_kernel_supported(const Params &P)
{
    if (P.knob_sym < 0) return false;
    if (!(P.ntaps == 42u && P.dc_len == 16u && P.win_ring >= 64u && (P.n_channels % kWave) == 0u)) return false;
    if (!((P.eq_nff == 6u && P.eq_nfb == 4u) || (P.eq_nff == 1u && P.eq_nfb == 1u))) return false;
    if (!(P.agc_min >= 0.0f)) return false;
    return 2u * (max_block_len(P) + 1u) > (uint32_t)kSymBlock;
}
uint32_t sym_block_len(const Params &P) { (void)P; return (uint32_t)kSymBlock; }

template <int NFF, int NFB, typename SampleT>
static hipError_t launch_sym_one(const Params &P, const State &S, const Output &O, const float4 *taps, const SampleT *x,
                                 uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{
    constexpr size_t lds = SymLayout<42>::lds_bytes;
    const bool cm = K.n_chunks > 1u && K.col_row0 != nullptr;
    if (cm && !std::is_same<SampleT, float>::value) return hipErrorInvalidValue;
    if (K.n_chunks > 1u && (K.in_channels % kWave) != 0u) return hipErrorInvalidValue;   // a workgroup would straddle chunks
    if (n_blocks == 0u) return hipSuccess;
    auto go = [&](auto kernel) -> hipError_t {
        if (lds > 64u * 1024u) {
            // more than the default 64 KB of dynamic LDS per workgroup: opt in, once per kernel and device
            static bool opted_in[64] = {};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
            if (!opted_in[dev]) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                opted_in[dev] = true;
            }
        }
        hipLaunchKernelGGL(kernel, dim3(P.n_channels / kWave), dim3(4 * kWave), lds, stream, P, S, O, taps, x, n_blocks, counter0, K);
        return hipGetLastError();
    };
    if constexpr (std::is_same<SampleT, float>::value) {
        if (cm) return go(demod_sym_kernel<NFF, NFB, float, 1>);
    }
    return go(demod_sym_kernel<NFF, NFB, SampleT, 0>);
}

template <typename SampleT>
static hipError_t launch_sym_t(const Params &P, const State &S, const Output &O, const float4 *taps, const SampleT *x,
                               uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{
    if (!sym_kernel_supported(P)) return hipErrorInvalidValue;
    if (P.eq_nff == 6u && P.eq_nfb == 4u) return launch_sym_one<6, 4, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K);
    return launch_sym_one<1, 1, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K);
}
hipError_t launch_demod_sym(const Params &P, const State &S, const Output &O, const float4 *taps, const float *x,
                            uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_sym_t<float>(P, S, O, taps, x, n_blocks, counter0, stream, K); }
hipError_t launch_demod_sym_i16(const Params &P, const State &S, const Output &O, const float4 *taps, const int16_t *x,
                                uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_sym_t<int16_t>(P, S, O, taps, x, n_blocks, counter0, stream, K); }

}  // namespace same

SYM_PROFILE_EXPORTS()
