// same_kernels_sym.hip -- the symbol-paced wavefront pipeline: relaxed arithmetic; 22.05 kHz in 36-sample steps (two groups of 64
// columns per CU), 44.1 / 48 kHz in 72-sample steps (one group per CU: SymGeom, SymDcRot, SymHalfTaps -- round 6).
// What follows describes the 22.05 kHz form; the other rates differ in geometry only.
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
//     The two matched-filter pairs of a symbol are therefore evaluated side by side, on two wavefronts (A at A, E at
//     B), and E runs the two timing updates after them: once per symbol and lane;
//   * a step is 36 samples, less than the shortest symbol the timing loop can command (two instants at least 19 samples
//     apart each: max_block_len), so every lane completes AT MOST one symbol per step and ~85 % of the lanes complete
//     exactly one: filters, timing loop and symbol path run once per step at ~85 % occupancy;
//   * the sample phase (DC blocker, AGC, window push) runs 36 samples per step: the fixed costs of a step (progress words,
//     feedback words, mailboxes) are paid per 36 samples instead of per 20.
//
// Six role-wavefronts per 64 state columns, one step apart (round 5; round 4 had four -- sample phase | filters + timing | symbol
// path | events -- at 4 900 / 5 400 / 3 200 / 700 clk per step), two such groups per twelve-wavefront workgroup:
//   T  input prefetch and DC blocker of block s + 1 -- it takes no feedback from anything (rx/dcblock.rs:45-49) -- written straight
//      into the window ring's next block
//   S  AGC (relaxed) of block s IN PLACE in the ring; the gain an AGC lock freezes is recomputed from the ring's AGC outputs
//   A  the matched-filter pair at the FIRST instant of the step's symbol (E posts where); then the link events and transport
//      wake-ups of what Y2 handed over one step earlier
//   E  the symbol whose instants lie in finished samples (blocks < s): the matched-filter pair at its SECOND instant, then the
//      two timing updates and where the next symbol's instants fall
//   Y1 squelch and equalizer of the symbol E handed over one step earlier (the squelch's sample history stays in the HBM state
//      arrays: two loads a step, issued a symbol ahead)
//   Y2 framer, link state, burst rows into the pool, hand-over of time-parallel pieces: one step behind Y1
// There is no step barrier: every role publishes one progress word per step and waits for the roles it exchanges data with.
// Feedback (agc.lock / loop bandwidth / symsync.reset, receiver.rs:431-432, 479-490) never sends a wavefront back: Y1 / Y2 post
// a change with the symbol it happened at, one word per lane.  S freezes the AGC from its next block on AT THE GAIN IT HAD
// after that symbol's sample (the gain every soft symbol of the burst is scaled by is therefore strict mode's to rounding), or
// releases it from its next block on; E switches the loop bandwidth, or resets the TED, behind its next symbol -- two symbols
// late.  The first form of this kernel replayed a lock at its sample (two more barriers in ~5 % of the steps, S and E filtering
// the symbol again): 8 % slower, and nothing the contract below asks for came of it.
//
// Window ring: six blocks of 36 slots -- the four the filters may reach into (a lane may lag up to 52 samples behind after a
// symsync.reset, the symbol's first filter reaches 25 + 41 further), the one S is turning into AGC outputs, the one T is writing
// -- with the first 41 slots stored twice (a filter's 42 samples never wrap).  DESIGN.md 4.6 has the full account.
//
// Parity contract: that of SAME_BATCH_RELAXED / the time-parallel mode (include/same_rx.h): transmitted bytes and
// transport messages equal strict mode's, link events within SAME_TP_EVENT_TOLERANCE_SYMBOLS symbols, soft symbols of
// an open squelch within 0.05.  The arithmetic is same_relaxed_common.h's.
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <set>
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

constexpr int kSymSub = 36;               // samples per SUB-BLOCK: what T loads and DC-blocks at a time (nine 16-byte loads per lane)
constexpr uint32_t kSymDrain = 5u;        // steps after the last block in which T processes the instants left before the end of the input, one per step
constexpr uint32_t kSymOffMask = 255u;    // a symbol's offset on its way through the mailboxes: < 3 B = 216 at 44.1 / 48 kHz
// Geometry per sample rate (filter length NT).  A STEP is B = NSUB sub-blocks: 36 samples at 22.05 kHz, 72 at 44.1 / 48 kHz -- less than
// the shortest symbol the timing loop can command at that rate (sym_kernel_supported checks it against the configuration), so a
// lane completes at most one symbol per step and ~78-85 % of the lanes complete exactly one.  HALVES: groups of 64 state columns per
// workgroup -- the ring of 6 x 72 + 91 slots is 134 KB at 48 kHz: ONE group per CU there (six wavefronts on four SIMDs).
template <int NT> struct SymGeom;
template <> struct SymGeom<42> { static constexpr int DCL = 16, NSUB = 1, HALVES = 2; };     // 22.05 kHz
template <> struct SymGeom<92> { static constexpr int DCL = 35, NSUB = 2, HALVES = 1; };     // 48 kHz
template <> struct SymGeom<84> { static constexpr int DCL = 32, NSUB = 2, HALVES = 1; };     // 44.1 kHz
template <int NT> struct SymLayout {
    // The window ring holds SIX blocks: the four the filters may reach into, the one S is turning from DC-blocker outputs into
    // AGC outputs IN PLACE, and the one T is writing DC-blocker outputs to (there is no hand-over ring between T and S).  Its first
    // NT - 1 slots are stored twice: a filter's NT samples never wrap.
    static constexpr int SB = kSymSub, NSUB = SymGeom<NT>::NSUB, B = SB * NSUB, DCL = SymGeom<NT>::DCL, NBLK = 6, RING = NBLK * B, MIR = NT - 1;
    static constexpr int HALVES = SymGeom<NT>::HALVES;
    // 22.05 kHz: the taps live in registers (the LDS words are the profile builds' section marks); 44.1 / 48 kHz: the centred tap
    // table (NT / 2 entries of Re mark, Re space, Im mark, Im space) at the start of the group's LDS
    static constexpr uint32_t tap_floats = (uint32_t)((NT * 4 + PIPE_PROF_TAP_PAD + 63) / 64 * 64);
    static constexpr uint32_t split_words = NT == 42 ? 0u : (2u + 8u) * kWave;
    static constexpr uint32_t sym_words = 5u * kWave;             // per parity: header, zero, sym, terr, until
    static constexpr uint32_t fb_words = kWave + 32u;             // per parity: one word per lane + the any-flag
    static constexpr uint32_t io_words = 3u * kWave;              // per parity: symbol word, burst-pool slot, burst length
    static constexpr uint32_t mail_words = 2u * sym_words + 2u * kWave +           // E -> Y1, Y1 -> Y2
                                           4u * fb_words + 2u * io_words +          // feedback from Y1 and from Y2, Y2 -> A
                                           3u * kWave +                             // final TED phase, wake-up flag, Y1's flag bits
                                           2u * kWave + kWave +                     // ring slot of the symbol's first instant (per parity), its soft sample
                                           kIoRingWords +                           // A's deadline ring and its count
                                           kWave +                                  // Y2 -> T: this lane has handed over
                                           kWave +                                  // the roles' progress words (six of them)
                                           (SYM_TL_WORDS + 63u) / 64u * 64u +       // (timeline builds: their marks)
                                           split_words;                             // 44.1 / 48 kHz: second instants' slots (per parity), A's partial sums
    static constexpr size_t lds_bytes = ((size_t)tap_floats + mail_words + (size_t)(RING + MIR) * kWave) * sizeof(float);
    static_assert(NT % 2 == 0 && NT - 1 <= 255, "tap pairs; ds_read2st64's 8-bit slot offsets");
    static_assert(SB % 4 == 0 && SB >= DCL + (DCL & 1), "16-byte loads per lane; the DC windows are the tail of a sub-block");
    static_assert(MIR <= 2 * B && MIR >= B, "the mirrored slots are the first block and the head of the second");
    static_assert(3 * B <= (int)kSymOffMask, "a symbol's offset is eight bits");
    static_assert((size_t)HALVES * lds_bytes <= 160u * 1024u, "the groups of a workgroup share a CU's LDS");
    static_assert(NT != 42 || lds_bytes <= 80u * 1024u, "22.05 kHz: two groups of 64 columns per workgroup and CU");
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
        // (all seven sums / differences of a group first, then the products: a packed operation that reads the result of the
        // instruction before it costs a wait state -- an s_nop and an issue slot each, 21 per filter)
        auto group = [&](auto g_, SymWin &W) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            W.w0 = sym_sum_diff(W.w0); W.w1 = sym_sum_diff(W.w1); W.w2 = sym_sum_diff(W.w2); W.w3 = sym_sum_diff(W.w3);
            W.w4 = sym_sum_diff(W.w4); W.w5 = sym_sum_diff(W.w5); W.w6 = sym_sum_diff(W.w6);
            auto pair = [&](auto K_, float2v sd) __attribute__((always_inline)) {
                constexpr int k = 7 * g + decltype(K_)::value;
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

// 44.1 / 48 kHz: 42 / 46 tap pairs are 168 / 184 registers -- next to the window words in flight they do not fit a wavefront's 256,
// and read from LDS (round 5's demod_pair_centred: one ds_read_b128 per pair) every group of seven pairs waits a whole LDS round trip
// for 14 loads: the filter was 2 100 clk of E's 3 800-clk step.  So the two filter wavefronts split the TAPS, not the instants: A
// holds pairs [0, H/2), E pairs [H/2, H) in registers, each adds its partial sums (Re / Im x mark / space) at BOTH instants of the
// step's symbol, A hands its eight words per lane to E, and E finishes both magnitudes.  No tap loads, half the registers, and the
// window loads of a whole instant (21 / 23) go out before the first product.
template <int K, int NT>
__device__ __forceinline__ void sym_win_load(float2v &w, uint32_t wa)
{
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(w) : "v"(wa), "n"(K), "n"(NT - 1 - K) : "memory");
}
// (the loaded words' first consumer, sym_sum_diff, is a volatile asm like the loads and this wait: the three keep their order; and
// every word the wait covers is redefined behind it -- sym_win_landed -- so that nothing made of it before the wait can be used after)
template <int N>
__device__ __forceinline__ void sym_win_wait()
{
    static_assert(N <= 15, "lgkmcnt is four bits");
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void sym_win_landed(float2v &w) { asm volatile("" : "+v"(w)); }
template <int NT, int K0, int K1>
struct SymHalfTaps {
    static constexpr int NP = K1 - K0, G = 8, NG = (NP + G - 1) / G;
    static_assert(NP >= 1 && NP <= 24, "three groups of at most eight pairs");
    float2v tc[NP], ts[NP];              // Re u_k (mark, space), Im u_k (mark, space), k = K0 + i
    __device__ __forceinline__ void load(const float4 *__restrict__ taps)
    {
        sym_static_for<NP>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value;
            const float4 t = taps[NT + K0 + i];
            tc[i] = float2v{t.x, t.y}; ts[i] = float2v{t.z, t.w};
            asm volatile("" : "+v"(tc[i]), "+v"(ts[i]));               // (vector registers: see SymTaps::load)
        });
    }
    // this half's share of sum_k (w_k + w_{N-1-k}) Re u_k and of sum_k (w_k - w_{N-1-k}) Im u_k at the instant whose sample sits in ring slot n
    template <int RING>
    __device__ __forceinline__ void partial(uint32_t wcol_lds, int n, float2v &re_out, float2v &im_out) const
    {
        int base = n - (NT - 1);
        base += base < 0 ? RING : 0;                          // slots RING .. RING + NT - 2 repeat slots 0 .. NT - 2
        const uint32_t wa = wcol_lds + (uint32_t)base * (kWave * 4u);
        float2v w[NP];
        sym_static_for<NP>([&](auto i_) __attribute__((always_inline)) { constexpr int i = decltype(i_)::value; sym_win_load<K0 + i, NT>(w[i], wa); });
        float2v re[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, im[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};      // [parity of the pair]: two chains per sum
        sym_static_for<NG>([&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value, i0 = g * G, i1 = (g + 1) * G < NP ? (g + 1) * G : NP;
            constexpr int left = NP - i1;                     // loads still to land behind this group (LDS returns in order)
            sym_win_wait<(left < 15 ? left : 15)>();
            sym_static_for<i1 - i0>([&](auto j_) __attribute__((always_inline)) { sym_win_landed(w[i0 + decltype(j_)::value]); });
            // (all sums / differences of a group first, then the products: a packed operation that reads the result of the
            // instruction before it costs a wait state)
            sym_static_for<i1 - i0>([&](auto j_) __attribute__((always_inline)) { constexpr int i = i0 + decltype(j_)::value; w[i] = sym_sum_diff(w[i]); });
            sym_static_for<i1 - i0>([&](auto j_) __attribute__((always_inline)) {
                constexpr int i = i0 + decltype(j_)::value;
                pk_fma_lo(re[i & 1], w[i], tc[i]);
                pk_fma_hi(im[i & 1], w[i], ts[i]);
            });
        });
        re_out = re[0] + re[1]; im_out = im[0] + im[1];
    }
};
// |mark| - |space| clamped to +-1 (rx/demod.rs:156-164) from the whole sums
__device__ __forceinline__ float sym_soft_sample(float2v r, float2v i)
{
    const float2v q = __builtin_elementwise_fma(i, i, r * r);          // (|mark|^2, |space|^2)
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_sqrtf(q.x) - __builtin_amdgcn_sqrtf(q.y), -1.0f, 1.0f);
}

// ---------------------------------------------------------------------------------------------------------------------
// T's sample half: input prefetch and DC blocker, one block ahead of S.  Inputs alternate between two register buffers
// (block b in buffer b & 1); the loads of block b + 1 are issued just before block b is computed, unconditionally (a load
// behind a condition makes the compiler wait for every outstanding load at the reads), so they have a whole
// step to arrive.  CMODE: 1 = channel-major input with per-lane streams, 0 = time-major rows.
// ---------------------------------------------------------------------------------------------------------------------
template <typename SampleT, int CMODE, int NT>
struct SymDc {
    // B: a SUB-BLOCK (36 samples at every rate: nine 16-byte loads per lane); a step is SymLayout::NSUB of them
    static constexpr int B = SymLayout<NT>::SB, DCL = SymLayout<NT>::DCL, RING = SymLayout<NT>::RING;
    static constexpr int HL = DCL + (DCL & 1);        // samples of history kept (whole pairs): 16, 32, 36 (48 kHz: the 36th is never read)
    static constexpr uint32_t LP = kWave;
    static_assert(CMODE == 0 || NT == 42, "per-lane streams: 22.05 kHz only (same_batch.cpp transposes the other rates' channel-major input)");
    static_assert(B >= HL && (B - HL) % 2 == 0 && ((DCL & 1) == 0 || HL == B), "the history is the tail of a sub-block, in whole pairs");
    // Everything is kept as aligned PAIRS (samples 2 i, 2 i + 1) and every access is a whole pair with a compile-time index:
    // the packed operations want aligned register pairs anyway, and an array that is read as pairs at both even and odd
    // offsets does not survive as registers (see sym_static_for).
    typedef float2v Pairs[B / 2];
    // The DC blocker (rx/dcblock.rs:45-49, 104-108) is two moving averages of DCL samples:
    //     sum0 += x - x[-DCL];  ma0 = sum0 / DCL;  sum1 += ma0 - ma0[-DCL];  ma1 = sum1 / DCL;  y = x[-(DCL-1)] - ma1.
    // DCL = 16 / 32 (22.05 / 44.1 kHz): a division by a power of two is exact, and adding, subtracting and rounding commute with
    // such a scaling, so the second average may run on the UNSCALED first sums -- S1 = DCL sum1 exactly, sample for sample -- and
    // y = x[-(DCL-1)] - S1 / DCL^2 is one fused multiply-add whose only rounding is the reference's own (S1 / DCL^2 is exact): the same
    // bits with 3.5 vector instructions per sample instead of 5.  The state arrays keep the reference's scaling (load / store).
    // DCL = 35 (48 kHz): ma = sum * (1 / 35) rounds, so every operation of the reference is made, in its order (5.5 per sample:
    // x[-35] and ma0[-35] of an aligned pair straddle two register pairs -- two scalar subtractions where the even lengths take one
    // packed one).  The same bits at every rate.
    float sum0, sum1;                    // sum1: DCL x the reference's (even DCL), the reference's (DCL = 35)
    float2v xp[HL / 2], sp[HL / 2];      // the last HL inputs / first-stage SUMS (DCL = 35: averages), oldest first
    // Input buffers: two, block b in (b & 1 ? xb : xa).  Time-major rows (CMODE 0): 36 buffer loads a block, issued just before the
    // block before theirs is computed, whose waits the compiler counts exactly.  Channel-major streams (CMODE 1): nine 16-byte loads
    // a block, issued by hand (inline assembly, explicit s_waitcnt vmcnt) INTO THE BUFFER A BLOCK HAS JUST BEEN COMPUTED FROM, i.e.
    // for the block after next: ~1.6 steps ahead with no third buffer (three do not fit the register file).  By hand, because the
    // compiler's wait insertion gives up on loads in flight across the loop's back edge and waited for ALL of them -- the ones just
    // issued included -- before a block's first read: a whole round trip to HBM in every step of the role that paces the headline
    // launch.  (Nothing in that role's loop may spill: a scratch access counts in vmcnt.)
    struct InBuf { Pairs p; float4v q[B / 4]; };
    InBuf xa, xb;
    uint32_t wpos = 0;                   // ring slot of the block written next
    const SampleT *xl = nullptr;         // CMODE 1: this lane's own stream
    uint32_t avail = 0;                  // ... and the blocks it holds
    bool done = false;                   // ... and whether the lane's piece has handed over: nothing it computes from here on is kept

    __device__ __forceinline__ void request(InBuf &buf, const SampleT *__restrict__ x, uint32_t blk, uint32_t n_blocks, uint32_t cin, uint32_t Cin) const
    {
        if constexpr (CMODE == 1) {
            // (a lane that has handed over stops streaming: in grid order the pieces of a workgroup differ in length, and the
            // short ones otherwise read on to the end of the longest -- 0.3 x the algorithmic bytes of a configs[1] launch.  It
            // re-reads its first block instead, from the cache: no load is ever behind a condition, and the count of loads in
            // flight -- what take() waits by -- is the same for every lane and step)
            const uint32_t b = done ? 0u : min(blk, avail - 1u);     // (avail >= 1: the planner leaves two scout blocks behind every cut)
            const float4 *p4 = reinterpret_cast<const float4 *>(xl + (size_t)b * B);
            static_assert(B / 4 == 9, "nine loads per block");
#define SYM_LD_(j_) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(buf.q[j_]) : "v"(p4), "n"(16 * (j_)) : "memory")
            SYM_LD_(0); SYM_LD_(1); SYM_LD_(2); SYM_LD_(3); SYM_LD_(4); SYM_LD_(5); SYM_LD_(6); SYM_LD_(7); SYM_LD_(8);
#undef SYM_LD_
        } else {
            Pairs &dst = buf.p;
            // buffer loads: the block's first row is the resource's base (re-based per block: a launch may exceed the 4 GB a
            // resource spans), row k at the scalar offset k * row_bytes, the lane's column as the vector offset -- one
            // instruction per sample and no address arithmetic on the vector unit (36 x 64-bit adds otherwise)
            const SampleT *xr = x + ((size_t)min(blk, n_blocks - 1u) * B) * Cin;      // wave-uniform
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<SampleT *>(xr), 0, 0x7fffffff, 0x00020000);
            const uint32_t voff = cin * (uint32_t)sizeof(SampleT);
            const uint32_t row_bytes = Cin * (uint32_t)sizeof(SampleT);     // (the 36 row offsets k * row_bytes are loop invariants: scalar registers for the whole launch)
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
    // The inputs of a requested block, once they have arrived.  CMODE 1: YOUNGER = the loads issued after this block's (nine: the
    // next block's); nothing may touch buf.q between request() and here -- the registers are not written yet.
    template <int YOUNGER>
    __device__ __forceinline__ void take(InBuf &buf, Pairs &X) const
    {
        if constexpr (CMODE == 1) {
            static_assert(B / 4 == 9, "nine loads per block");
            asm volatile("s_waitcnt vmcnt(%9)"
                         : "+v"(buf.q[0]), "+v"(buf.q[1]), "+v"(buf.q[2]), "+v"(buf.q[3]), "+v"(buf.q[4]), "+v"(buf.q[5]), "+v"(buf.q[6]),
                           "+v"(buf.q[7]), "+v"(buf.q[8])
                         : "n"(YOUNGER) : "memory");
            sym_static_for<B / 4>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value;
                X[2 * j] = float2v{buf.q[j].x, buf.q[j].y}; X[2 * j + 1] = float2v{buf.q[j].z, buf.q[j].w};
            });
        } else {
            asm volatile("" :: "v"(buf.p[B / 2 - 1]));        // (one s_waitcnt for the block's loads instead of one per register: SymDcRot::sub)
            sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) { X[decltype(h_)::value] = buf.p[decltype(h_)::value]; });
        }
    }
    // history sample i (0 = oldest of HL) sits in the state ring's slot (dpos + i - (HL - DCL)) mod DCL
    static __device__ __forceinline__ uint32_t ring_slot(uint32_t dpos, int i)
    {
        uint32_t s = dpos + (uint32_t)(i - (HL - DCL));
        if (s >= (uint32_t)DCL) s -= (uint32_t)DCL;
        return s;
    }
    __device__ __forceinline__ void load(const State &S, const SampleT *__restrict__ x, uint32_t c, uint32_t C, uint32_t cin, uint32_t Cin,
                                         uint64_t counter0, uint32_t n_blocks)
    {
        constexpr float up = (DCL & 1) ? 1.0f : (float)DCL;
        sum0 = S.dc_sum0[c]; sum1 = S.dc_sum1[c] * up;
        const uint32_t dpos = (uint32_t)(counter0 % (uint64_t)DCL);
        sym_static_for<HL / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            if constexpr (2 * h < HL - DCL) {                      // (DCL = 35: history sample 0 is x[-36], which nothing reads)
                const uint32_t s1 = ring_slot(dpos, 2 * h + 1);
                xp[h] = float2v{0.0f, (S.dc_ff_ring + (size_t)s1 * C)[c]};
                sp[h] = float2v{0.0f, (S.dc_fb_ring + (size_t)s1 * C)[c] * up};
            } else {
                const uint32_t s0 = ring_slot(dpos, 2 * h), s1 = ring_slot(dpos, 2 * h + 1);
                xp[h] = float2v{(S.dc_ff_ring + (size_t)s0 * C)[c], (S.dc_ff_ring + (size_t)s1 * C)[c]};
                sp[h] = float2v{(S.dc_fb_ring + (size_t)s0 * C)[c] * up, (S.dc_fb_ring + (size_t)s1 * C)[c] * up};
            }
        });
        if constexpr (CMODE == 0) sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) { xb.p[decltype(h_)::value] = float2v{0.0f, 0.0f}; });
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
        // the windows before this block, oldest first, as pairs: 0 .. HL/2 - 1 the history, then this block's
        auto xw = [&](auto i_) __attribute__((always_inline)) -> float2v {
            constexpr int i = decltype(i_)::value;
            if constexpr (i < HL / 2) return xp[i]; else return X[i - HL / 2];
        };
        auto sw = [&](auto i_) __attribute__((always_inline)) -> float2v {
            constexpr int i = decltype(i_)::value;
            if constexpr (i < HL / 2) return sp[i]; else return snew[i - HL / 2];
        };
        if constexpr ((DCL & 1) == 0) {
            constexpr float kScale = -1.0f / (float)(DCL * DCL);
            static_assert((DCL & (DCL - 1)) == 0, "a power of two: the scalings above are exact");
            const float2v nscale = {kScale, kScale};
            float s1_last = 0.0f;                                          // S1 after the odd sample of the pair before
            sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value;                     // samples k = 2 h, 2 h + 1
                const float2v xo = xw(std::integral_constant<int, h>{});   // inputs 2 h - DCL, 2 h - DCL + 1
                const float2v d0 = X[h] - xo;
                const float s0a = sum0 + d0.x, s0b = s0a + d0.y;
                sum0 = s0b;
                const float2v s0 = {s0a, s0b};
                snew[h] = s0;
                const float2v d1 = s0 - sw(std::integral_constant<int, h>{});
                const float s1a = sum1 + d1.x, s1b = s1a + d1.y;
                sum1 = s1b;
                // y[k] = x[k - (DCL - 1)] - S1[k] / DCL^2: the pair (y[2 h - 1], y[2 h]) meets the ALIGNED input pair xo
                if constexpr (h == 0) {
                    y[0] = __builtin_fmaf(s1a, kScale, xo.y);
                } else {
                    const float2v yy = __builtin_elementwise_fma(float2v{s1_last, s1a}, nscale, xo);
                    y[(2 * h - 1) * LP] = yy.x; y[(2 * h) * LP] = yy.y;
                }
                s1_last = s1b;
                if constexpr (h == B / 2 - 1) y[(B - 1) * LP] = __builtin_fmaf(s1b, kScale, xw(std::integral_constant<int, h + 1>{}).x);
            });
        } else {
            // DCL = 35 with 36 samples of history: x[k - 35] is history sample k + 1, x[k - 34] (the delayed signal) sample k + 2
            const float inv = P.dc_inv_len;
            const float2v inv2 = {inv, inv};
            sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value;                     // samples k = 2 h, 2 h + 1
                const float2v xlo = xw(std::integral_constant<int, h>{}), xhi = xw(std::integral_constant<int, h + 1>{});
                const float d0a = X[h].x - xlo.y, d0b = X[h].y - xhi.x;    // MovingAverage::filter: moving_sum += input - aged
                const float s0a = sum0 + d0a, s0b = s0a + d0b;
                sum0 = s0b;
                const float2v m0 = float2v{s0a, s0b} * inv2;               // ma0 = moving_sum * inv_len
                snew[h] = m0;
                const float2v mlo = sw(std::integral_constant<int, h>{}), mhi = sw(std::integral_constant<int, h + 1>{});
                const float d1a = m0.x - mlo.y, d1b = m0.y - mhi.x;
                const float s1a = sum1 + d1a, s1b = s1a + d1b;
                sum1 = s1b;
                const float2v yy = xhi - float2v{s1a, s1b} * inv2;         // sig - ma1 (rx/dcblock.rs:45-49): both of the reference's roundings
                y[(2 * h) * LP] = yy.x; y[(2 * h + 1) * LP] = yy.y;
            });
        }
        sym_static_for<HL / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            xp[h] = X[(B - HL) / 2 + h]; sp[h] = snew[(B - HL) / 2 + h];
        });
    }
    __device__ __forceinline__ void store(const State &S, uint32_t c, uint32_t C, uint64_t counter1)
    {
        constexpr float inv = (DCL & 1) ? 1.0f : 1.0f / (float)DCL;
        S.dc_sum0[c] = sum0; S.dc_sum1[c] = sum1 * inv;
        const uint32_t dpos = (uint32_t)(counter1 % (uint64_t)DCL);
        sym_static_for<HL / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            if constexpr (2 * h >= HL - DCL) {
                const uint32_t s0 = ring_slot(dpos, 2 * h);
                (S.dc_ff_ring + (size_t)s0 * C)[c] = xp[h].x; (S.dc_fb_ring + (size_t)s0 * C)[c] = sp[h].x * inv;
            }
            const uint32_t s1 = ring_slot(dpos, 2 * h + 1);
            (S.dc_ff_ring + (size_t)s1 * C)[c] = xp[h].y; (S.dc_fb_ring + (size_t)s1 * C)[c] = sp[h].y * inv;
        });
    }
};

// T at 44.1 / 48 kHz (time-major rows, one group per CU: up to 256 registers).  The same DC blocker, but NOTHING IS COPIED: the
// inputs of a sub-block stay in the buffer they were loaded into and ARE the history of the next one (three buffers in rotation: q - 1
// the history, q being computed, q + 1 in flight), and the first-stage sums / averages alternate between two sets.  SymDc copies the
// tail of every sub-block into history registers -- 36 64-bit moves per 36 samples at 48 kHz, 12 % of the instructions of the role
// that paces configs[2].  The pattern repeats every six sub-blocks = three steps: PH = q mod 6 is a compile-time constant.
template <typename SampleT, int NT>
struct SymDcRot {
    static constexpr int B = SymLayout<NT>::SB, DCL = SymLayout<NT>::DCL, RING = SymLayout<NT>::RING;
    static constexpr int HL = DCL + (DCL & 1), OFF = (B - HL) / 2;          // the history: pairs OFF .. B/2 - 1 of the sub-block before
    static constexpr uint32_t LP = kWave;
    static_assert(B >= HL && (B - HL) % 2 == 0, "the history is the tail of a sub-block, in whole pairs");
    typedef float2v Pairs[B / 2];
    float sum0, sum1;                    // sum1: DCL x the reference's (even DCL), the reference's (DCL = 35)
    Pairs xb[3];                         // inputs of sub-blocks q - 1, q, q + 1 (q mod 3)
    Pairs mb[2];                         // first-stage sums (DCL = 35: averages) of sub-blocks q - 1, q (q mod 2)
    uint32_t wpos = 0;                   // ring slot of the sub-block written next

    __device__ __forceinline__ void request(Pairs &dst, const SampleT *__restrict__ x, uint32_t blk, uint32_t n_sub, uint32_t cin, uint32_t Cin) const
    {
        // (buffer loads with scalar row offsets: see SymDc::request)
        const SampleT *xr = x + ((size_t)min(blk, n_sub - 1u) * B) * Cin;      // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<SampleT *>(xr), 0, 0x7fffffff, 0x00020000);
        const uint32_t voff = cin * (uint32_t)sizeof(SampleT);
        const uint32_t row_bytes = Cin * (uint32_t)sizeof(SampleT);
        auto one = [&](uint32_t k) __attribute__((always_inline)) -> float {
            if constexpr (sizeof(SampleT) == 4) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, k * row_bytes, 0));
            else return (float)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, k * row_bytes, 0);
        };
        sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            dst[h] = float2v{one(2u * h), one(2u * h + 1u)};
        });
    }
    // history sample i (0 = oldest of HL) sits in the state ring's slot (dpos + i - (HL - DCL)) mod DCL
    static __device__ __forceinline__ uint32_t ring_slot(uint32_t dpos, int i)
    {
        uint32_t s = dpos + (uint32_t)(i - (HL - DCL));
        if (s >= (uint32_t)DCL) s -= (uint32_t)DCL;
        return s;
    }
    // the state's history as "sub-block -1": buffer 2, set 1
    __device__ __forceinline__ void load(const State &S, const SampleT *__restrict__ x, uint32_t c, uint32_t C, uint32_t cin, uint32_t Cin,
                                         uint64_t counter0, uint32_t n_sub)
    {
        constexpr float up = (DCL & 1) ? 1.0f : (float)DCL;
        sum0 = S.dc_sum0[c]; sum1 = S.dc_sum1[c] * up;
        const uint32_t dpos = (uint32_t)(counter0 % (uint64_t)DCL);
        sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            xb[0][h] = float2v{0.0f, 0.0f}; xb[1][h] = float2v{0.0f, 0.0f}; xb[2][h] = float2v{0.0f, 0.0f};
            mb[0][h] = float2v{0.0f, 0.0f}; mb[1][h] = float2v{0.0f, 0.0f};
        });
        sym_static_for<HL / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            const uint32_t s1 = ring_slot(dpos, 2 * h + 1);
            float x0 = 0.0f, m0 = 0.0f;
            if constexpr (2 * h >= HL - DCL) {                     // (DCL = 35: history sample 0 is x[-36], which nothing reads)
                const uint32_t s0 = ring_slot(dpos, 2 * h);
                x0 = (S.dc_ff_ring + (size_t)s0 * C)[c]; m0 = (S.dc_fb_ring + (size_t)s0 * C)[c] * up;
            }
            xb[2][OFF + h] = float2v{x0, (S.dc_ff_ring + (size_t)s1 * C)[c]};
            mb[1][OFF + h] = float2v{m0, (S.dc_fb_ring + (size_t)s1 * C)[c] * up};
        });
        request(xb[0], x, 0u, n_sub, cin, Cin);
    }
    // Sub-block q (PH = q mod 6): its successor's loads go out, then its DC-blocker outputs go into the window ring's next 36 slots
    template <int PH>
    __device__ __forceinline__ void sub(const Params &P, float *wcol, const SampleT *__restrict__ x, uint32_t q, uint32_t n_sub, uint32_t cin, uint32_t Cin)
    {
        constexpr int I = PH % 3, IP = (PH + 2) % 3, IN = (PH + 1) % 3, J = PH % 2, JP = 1 - J;
        request(xb[IN], x, q + 1u, n_sub, cin, Cin);
        Pairs &X = xb[I], &XP = xb[IP], &M = mb[J], &MP = mb[JP];
        // (the compiler waits before the FIRST use of every loaded register, each time for one load more: 25 s_waitcnt per sub-block for
        // loads that landed a sub-block ago.  Loads return in order: naming the youngest of this sub-block's first makes it ONE wait)
        asm volatile("" :: "v"(X[B / 2 - 1]));
        float *y = wcol + wpos * LP;
        wpos += (uint32_t)B;
        if (wpos == (uint32_t)RING) wpos = 0u;
        // the windows before this sub-block, oldest first, as pairs: 0 .. HL/2 - 1 the history (the tail of the one before), then this one's
        auto xw = [&](auto i_) __attribute__((always_inline)) -> float2v {
            constexpr int i = decltype(i_)::value;
            if constexpr (i < HL / 2) return XP[OFF + i]; else return X[i - HL / 2];
        };
        auto sw = [&](auto i_) __attribute__((always_inline)) -> float2v {
            constexpr int i = decltype(i_)::value;
            if constexpr (i < HL / 2) return MP[OFF + i]; else return M[i - HL / 2];
        };
        if constexpr ((DCL & 1) == 0) {
            // the exact-scaling form (SymDc::block): S1 = DCL sum1, y = x[-(DCL-1)] - S1 / DCL^2 in one fused multiply-add
            constexpr float kScale = -1.0f / (float)(DCL * DCL);
            static_assert((DCL & (DCL - 1)) == 0, "a power of two: the scalings are exact");
            const float2v nscale = {kScale, kScale};
            float s1_last = 0.0f;
            sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value;                     // samples k = 2 h, 2 h + 1
                const float2v xo = xw(std::integral_constant<int, h>{});   // inputs 2 h - DCL, 2 h - DCL + 1
                const float2v d0 = X[h] - xo;
                const float s0a = sum0 + d0.x, s0b = s0a + d0.y;
                sum0 = s0b;
                const float2v s0 = {s0a, s0b};
                M[h] = s0;
                const float2v d1 = s0 - sw(std::integral_constant<int, h>{});
                const float s1a = sum1 + d1.x, s1b = s1a + d1.y;
                sum1 = s1b;
                if constexpr (h == 0) {
                    y[0] = __builtin_fmaf(s1a, kScale, xo.y);
                } else {
                    const float2v yy = __builtin_elementwise_fma(float2v{s1_last, s1a}, nscale, xo);
                    y[(2 * h - 1) * LP] = yy.x; y[(2 * h) * LP] = yy.y;
                }
                s1_last = s1b;
                if constexpr (h == B / 2 - 1) y[(B - 1) * LP] = __builtin_fmaf(s1b, kScale, xw(std::integral_constant<int, h + 1>{}).x);
            });
        } else {
            // DCL = 35 (SymDc::block): every operation of the reference, in its order
            const float inv = P.dc_inv_len;
            const float2v inv2 = {inv, inv};
            sym_static_for<B / 2>([&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value;
                const float2v xlo = xw(std::integral_constant<int, h>{}), xhi = xw(std::integral_constant<int, h + 1>{});
                const float d0a = X[h].x - xlo.y, d0b = X[h].y - xhi.x;
                const float s0a = sum0 + d0a, s0b = s0a + d0b;
                sum0 = s0b;
                const float2v m0 = float2v{s0a, s0b} * inv2;
                M[h] = m0;
                const float2v mlo = sw(std::integral_constant<int, h>{}), mhi = sw(std::integral_constant<int, h + 1>{});
                const float d1a = m0.x - mlo.y, d1b = m0.y - mhi.x;
                const float s1a = sum1 + d1a, s1b = s1a + d1b;
                sum1 = s1b;
                const float2v yy = xhi - float2v{s1a, s1b} * inv2;
                y[(2 * h) * LP] = yy.x; y[(2 * h + 1) * LP] = yy.y;
            });
        }
    }
    // `q_last`: the last sub-block computed (its inputs and sums are the history the state keeps)
    __device__ __forceinline__ void store(const State &S, uint32_t c, uint32_t C, uint64_t counter1, uint32_t q_last)
    {
        constexpr float inv = (DCL & 1) ? 1.0f : 1.0f / (float)DCL;
        S.dc_sum0[c] = sum0; S.dc_sum1[c] = sum1 * inv;
        const uint32_t dpos = (uint32_t)(counter1 % (uint64_t)DCL);
        const uint32_t i3 = q_last % 3u, j2 = q_last & 1u;             // wave-uniform
        sym_static_for<HL / 2>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value;
            // (values first, then the choice: a `?:` of two array elements selects an ADDRESS, and the arrays stay in scratch memory)
            float2v x0 = xb[0][OFF + h], x1 = xb[1][OFF + h], x2 = xb[2][OFF + h], m0 = mb[0][OFF + h], m1 = mb[1][OFF + h];
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(m0), "+v"(m1));
            const float2v xh = i3 == 0u ? x0 : (i3 == 1u ? x1 : x2);
            const float2v mh = j2 == 0u ? m0 : m1;
            if constexpr (2 * h >= HL - DCL) {
                const uint32_t s0 = ring_slot(dpos, 2 * h);
                (S.dc_ff_ring + (size_t)s0 * C)[c] = xh.x; (S.dc_fb_ring + (size_t)s0 * C)[c] = mh.x * inv;
            }
            const uint32_t s1 = ring_slot(dpos, 2 * h + 1);
            (S.dc_ff_ring + (size_t)s1 * C)[c] = xh.y; (S.dc_fb_ring + (size_t)s1 * C)[c] = mh.y * inv;
        });
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// S: AGC over the window ring's newest block, in place (T left the DC blocker's outputs there), and the gain a lock freezes
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
struct SymAgc {
    static constexpr int B = SymLayout<NT>::B, RING = SymLayout<NT>::RING, MIR = SymLayout<NT>::MIR;
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
        // (naming the last of them here -- ONE lgkmcnt wait for the block's reads, as T does for its loads -- measured equal within
        // noise, same box, three alternations: 1.845 / 2.421 ms against 1.854 / 2.415 at 22.05 / 48 kHz: LDS returns early enough
        // for the waits the compiler inserts per use to be no-ops)
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

// ---------------------------------------------------------------------------------------------------------------------
// Y1: squelch and equalizer of one symbol (rx/codesquelch.rs:228-304, rx/equalize.rs:249-308), the relaxed-only cut of
// same_dev_common.h's rx_symbol.  The strict kernels keep that function bit for bit; here the symbol path is TWO wavefronts --
// Y1 = squelch + equalizer, Y2 = framer + link state + bursts -- one step apart, so what the framer decides (squelch.lock(true)
// on Reading, end() on NoCarrier / Burst: receiver.rs:457-471) reaches the squelch one symbol late.  Neither can matter to a
// symbol in between: a locked squelch only stops looking for a NEW sync, which needs 32 matching preamble bits, and after an
// end() the byte clock Y1 kept running for one more symbol is overwritten; the equalizer step it took is undone by the reset.
// ---------------------------------------------------------------------------------------------------------------------
// Y1 -> Y2, one word per lane and step
enum : uint32_t { YM_VALID = 1u, YM_READY = 2u, YM_ADJUSTED = 4u, YM_READING = 8u, YM_DROP = 16u, YM_BYTE_SHIFT = 8, YM_OFF_SHIFT = 16 };
// feedback words (Y1 / Y2 -> S, E; Y2 -> Y1): bit 0 valid, 1 AGC locked, 2 loop bandwidth locked, 3 end(), 8.. the symbol's offset;
// bit 4 (Y2 -> Y1 only, without bit 0): squelch.lock(true)
enum : uint32_t { FB_VALID = 1u, FB_AGC = 2u, FB_BW = 4u, FB_END = 8u, FB_SQLOCK = 16u };

template <int NFF, int NFB>
struct SymSquelch {
    Lane L;                              // sq_*, eq_word, eq_count, flags (AGC / BW / squelch locks, equalizer mode and bits)
    uint32_t nsym;                       // low word of the symbol counter
    float ffc[NFF], ffw[NFF], fbc[NFB], fbw[NFB];
    float sffc[NFF], sffw[NFF], sfbc[NFB], sfbw[NFB];   // equalizer at the last completed byte
    // The squelch's sample history stays in global memory (LDS is full): one wave-uniform base and 32-bit byte offsets, so an
    // access is two vector instructions (round 4: a 64-bit multiply-add and a 64-bit add each).  The two samples the NEXT
    // symbol's equalizer step takes (slots +16 / +17 from the squelch's write position, i.e. +18 / +19 from this symbol's) are
    // requested where this symbol's are stored: a global round trip is a step long under load.  Only this lane's own symbols
    // write its history, and never those two slots before they are read.
    char *hbase;
    uint32_t hcol4, hrow4;               // byte offset of the lane's column; bytes between two slots
    float nxt0 = 0.0f, nxt1 = 0.0f;
    __device__ __forceinline__ float *hptr(uint32_t slot) const { return reinterpret_cast<float *>(hbase + (size_t)(__umul24(slot, hrow4) + hcol4)); }

    __device__ __forceinline__ void eq_reset()
    {
#pragma unroll
        for (int i = 0; i < NFF; ++i) { ffc[i] = (i == 0) ? 1.0f : 0.0f; ffw[i] = 0.0f; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { fbc[i] = (i == 0) ? 1.0f : 0.0f; fbw[i] = 0.0f; }
    }
    // what SameReceiver::end (receiver.rs:479-490) does to this wavefront's state; S and E get theirs as feedback
    __device__ __forceinline__ void end()
    {
        L.flags &= ~(F_AGC_LOCKED | F_SQ_LOCK | F_BW_LOCKED);
        L.sq_clock = -1;                                   // squelch.end() rx/codesquelch.rs:336-339
        eq_reset();                                        // Equalizer::reset rx/equalize.rs:191-196 (mode preserved)
    }
    // ---- the select-committed form (round 6; -DSYM_Y1_SELECT: NOT the default) ----------------------------------------------------
    // MEASURED SLOWER than symbol() below although it is fewer instructions in the listing (514 against 669 between the role's step
    // marks, 105 moves against 177: tools/sym_role_mix.py) and delivers the same events bit for bit (tools/sym_hash.py: five digests
    // equal): same box, three alternations -- 32 768 channels 1.887 against 1.858 ms, the headline launch 1.839 against 1.775, 48 kHz
    // 2.49 against 2.45.  Every lane executing every masked step costs more issue slots than the moves it removes, and the moves at
    // the loop head stay (the rare regions' results live in other registers than the loop's, and the compiler copies on the hot
    // edge).  Kept as the differential form of symbol(): what the round-5 review's item 1a asked to be tried.
    // Y1's state is ~60 registers (equalizer 20, its snapshot 20, squelch and byte clock), and symbol() below changes them inside
    // exec-mask regions nested three deep: for every value that leaves a divergent region changed on some lanes the compiler keeps
    // two copies -- 177 of the 669 instructions between this role's step marks were register moves, 38 of them at the head of
    // every step (tools/sym_role_mix.py).  Here nothing is changed under an exec mask: every lane computes the step, the lanes it
    // applies to COMMIT it with selects, and what is rare (training at sync, restoring / taking the snapshot, end()) sits behind a
    // wave-uniform branch on a ballot -- an ordinary join the register allocator resolves in place.  Same operations on the same
    // values: a launch delivers what symbol() delivers, event for event (tools/sym_hash.py).
    //
    // Equalizer::estimate_symbol + evolve for the lanes in `act` (eq_symbol_relaxed, same_relaxed_common.h, with the commit masked);
    // returns the decided bit
    __device__ __forceinline__ uint32_t eq_step_masked(const Params &P, bool act, float in0, float in1)
    {
        const uint32_t mode = (L.flags & F_EQ_MODE_MASK) >> F_EQ_MODE_SHIFT;
        float w[NFF];                                      // the feed-forward window with this symbol's two samples pushed
        if constexpr (NFF >= 2) {
#pragma unroll
            for (int i = 0; i + 2 < NFF; ++i) w[i] = ffw[i + 2];
            w[NFF - 2] = in0; w[NFF - 1] = in1;
        } else {
            w[0] = in1;
        }
        float f0 = 0.0f, f1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
#pragma unroll
        for (int i = 0; i < NFF; ++i) {
            if (i & 1) { f1 = __builtin_fmaf(w[NFF - 1 - i], ffc[i], f1); q1 = __builtin_fmaf(w[i], w[i], q1); }
            else { f0 = __builtin_fmaf(w[NFF - 1 - i], ffc[i], f0); q0 = __builtin_fmaf(w[i], w[i], q0); }
        }
        constexpr auto fb_zero = [](int widx) { return ((NFB - 1 - widx) & 1) == 0; };      // (exact zeros in every other slot: eq_symbol_relaxed)
        float fbs = 0.0f, qb = 0.0f;
#pragma unroll
        for (int i = 0; i < NFB; ++i) {
            if (!fb_zero(NFB - 1 - i)) fbs = __builtin_fmaf(fbw[NFB - 1 - i], fbc[i], fbs);
            if (!fb_zero(i)) qb = __builtin_fmaf(fbw[i], fbw[i], qb);
        }
        const float sym_val = (f0 + f1) - fbs;
        const bool training = act & (mode == 2u), evolve = act & (mode != 0u);
        const float sym_est = (mode == 2u) ? ((L.eq_word & 1u) ? 1.0f : -1.0f) : rs_signum(sym_val);
        const float err = evolve ? sym_est - sym_val : 0.0f;                                // (0: the updates below add exact zeros)
        L.eq_word = training ? L.eq_word >> 1 : L.eq_word;
        L.eq_count = training ? L.eq_count + 1u : L.eq_count;
        const uint32_t mode1 = (training && L.eq_count >= 32u) ? 1u : mode;
        const float gf = P.eq_relaxation * __builtin_amdgcn_rcpf(P.eq_regularization + (q0 + q1));
        const float gb = P.eq_relaxation * __builtin_amdgcn_rcpf(P.eq_regularization + qb);
        const float ge = gf * err, gn = -(gb * err);
#pragma unroll
        for (int i = 0; i < NFF; ++i) ffc[i] = __builtin_fmaf(ge, w[NFF - 1 - i], ffc[i]);
#pragma unroll
        for (int i = 0; i < NFB; ++i) { if (!fb_zero(NFB - 1 - i)) fbc[i] = __builtin_fmaf(gn, fbw[NFB - 1 - i], fbc[i]); }
#pragma unroll
        for (int i = 0; i < NFF; ++i) ffw[i] = act ? w[i] : ffw[i];
        if constexpr (NFB >= 2) {
#pragma unroll
            for (int i = 0; i + 2 < NFB; ++i) fbw[i] = act ? fbw[i + 2] : fbw[i];
            fbw[NFB - 2] = act ? sym_est : fbw[NFB - 2];
            fbw[NFB - 1] = act ? 0.0f : fbw[NFB - 1];
        } else {
            fbw[0] = act ? 0.0f : fbw[0];
        }
        L.flags = (L.flags & ~F_EQ_MODE_MASK) | (mode1 << F_EQ_MODE_SHIFT);
        return sym_est >= 0.0f ? 1u : 0u;
    }
    // Equalizer::reset rx/equalize.rs:191-196 (mode preserved) + squelch.end() + the locks, for the lanes in `m`
    __device__ __forceinline__ void end_masked(bool m)
    {
        L.flags = m ? (L.flags & ~(F_AGC_LOCKED | F_SQ_LOCK | F_BW_LOCKED)) : L.flags;
        L.sq_clock = m ? -1 : L.sq_clock;
#pragma unroll
        for (int i = 0; i < NFF; ++i) { ffc[i] = m ? ((i == 0) ? 1.0f : 0.0f) : ffc[i]; ffw[i] = m ? 0.0f : ffw[i]; }
#pragma unroll
        for (int i = 0; i < NFB; ++i) { fbc[i] = m ? ((i == 0) ? 1.0f : 0.0f) : fbc[i]; fbw[i] = m ? 0.0f : fbw[i]; }
    }
    // One symbol for the lanes in `valid` (the others keep their state).  Returns the word for Y2 (0 without a symbol); *fb = the
    // feedback word for S and E (0: none).
    __device__ __forceinline__ uint32_t symbol_masked(const Params &P, bool valid, float zero, float sym, uint32_t off, uint32_t *fb)
    {
        const uint32_t before = L.flags & (F_AGC_LOCKED | F_BW_LOCKED);
        // --- CodeAndPowerSquelch::input rx/codesquelch.rs:228-304
        const uint32_t slot = (2u * nsym) & 63u;
        const float eq_in0 = nxt0, eq_in1 = nxt1;          // history slots slot + 16 / + 17, requested a symbol ago
        {
            float *pn = hptr((slot + 18u) & 63u), *pw = hptr(slot);
            const float l0 = pn[0], l1 = *reinterpret_cast<float *>(reinterpret_cast<char *>(pn) + hrow4);      // (every lane: the address is its own column's)
            nxt0 = valid ? l0 : nxt0; nxt1 = valid ? l1 : nxt1;
            if (valid) { pw[0] = zero; *reinterpret_cast<float *>(reinterpret_cast<char *>(pw) + hrow4) = sym; }      // (stores only: nothing leaves this region)
        }
        const uint32_t fill = min(64u, L.sq_fill + 2u);
        L.sq_fill = valid ? fill : L.sq_fill;
        const uint32_t data = (L.sq_data >> 1) | ((sym >= 0.0f) ? 0x80000000u : 0u);        // CodeCorrelator::search :421-428
        L.sq_data = valid ? data : L.sq_data;
        const uint32_t nerr = __popc(P.sync_word ^ data);
        const float pwr = fmaxf(__builtin_fmaf(__builtin_fmaf(sym, sym, -L.sq_power), P.sq_bw, L.sq_power), 0.0f);   // PowerTracker::track :483-488
        L.sq_power = valid ? pwr : L.sq_power;
        const uint32_t phist = (L.sq_phist << 1) | ((pwr >= P.sq_power_close) ? 1u : 0u);
        L.sq_phist = valid ? phist : L.sq_phist;
        nsym += valid ? 1u : 0u;
        const int32_t clock_before = L.sq_clock;           // byte clock before this symbol (-1: no sync)
        const bool full = valid & (fill >= 64u);           // sample_history.is_full() :237
        const bool locked = (L.flags & F_SQ_LOCK) != 0u;
        const bool sync_now = full & !locked & (nerr <= P.sq_max_errors) & (pwr >= P.sq_power_open);     // :244-265
        const bool adjusted = sync_now & (clock_before != 0);
        const bool drop = full & !sync_now & (clock_before >= 0) & ((phist & 0x80000000u) == 0u);      // :266-273
        int32_t clk = sync_now ? 0 : clock_before;
        clk = drop ? -1 : clk;
        const bool ready = full & (clk == 0);              // :277-303 byte clock
        const bool reading = full & (clk > 0);
        clk = ready ? 1 : (reading ? ((clk + 1) & 7) : clk);
        L.sq_clock = clk;                                  // (unchanged where the lane has no symbol: full is false)
        // Equalizer schedule: see symbol()
        const bool act = (clock_before >= 0) & !adjusted & (ready | reading);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(act) != 0ull, 1)) {
            const uint32_t j = (uint32_t)(clock_before + 7) & 7u;   // clock 1..7 -> symbol 0..6, clock 0 -> 7
            const uint32_t ebit = eq_step_masked(P, act, eq_in0, eq_in1);
            uint32_t bits = (j == 0u) ? 0u : ((L.flags & F_EQ_BITS_MASK) >> F_EQ_BITS_SHIFT);
            bits |= ebit << j;
            L.flags = act ? ((L.flags & ~F_EQ_BITS_MASK) | (bits << F_EQ_BITS_SHIFT)) : L.flags;
        }
        uint32_t byte = (L.flags & F_EQ_BITS_MASK) >> F_EQ_BITS_SHIFT;
        const bool train = ready & adjusted;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(train) != 0ull, 0)) {
            // sync acquired or the byte clock re-aligned (receiver.rs:423-446): lock AGC and loop bandwidth, train on the sync
            // word over the oldest 16 samples of the history (rx/codesquelch.rs:288-294)
            const bool back = train & (clock_before >= 0);         // drop the symbols equalized ahead for a byte the reference never forms
#pragma unroll
            for (int i = 0; i < NFF; ++i) { ffc[i] = back ? sffc[i] : ffc[i]; ffw[i] = back ? sffw[i] : ffw[i]; }
#pragma unroll
            for (int i = 0; i < NFB; ++i) { fbc[i] = back ? sfbc[i] : fbc[i]; fbw[i] = back ? sfbw[i] : fbw[i]; }
            L.flags = train ? ((L.flags & ~F_EQ_MODE_MASK) | (2u << F_EQ_MODE_SHIFT) | F_AGC_LOCKED | F_BW_LOCKED) : L.flags;   // equalizer.train()
            L.eq_word = train ? P.sync_word : L.eq_word; L.eq_count = train ? 0u : L.eq_count;
            const uint32_t head = (2u * nsym) & 63u;       // oldest sample
            float samples[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) samples[i] = *hptr((head + (uint32_t)i) & 63u);
            uint32_t tb = 0u;
#pragma unroll 1
            for (int b = 0; b < 8; ++b) tb |= eq_step_masked(P, train, samples[2 * b], samples[2 * b + 1]) << b;
            byte = train ? tb : byte;
        }
        // a re-alignment is still possible: remember the equalizer as of this completed byte
        const bool snap = ready & ((L.flags & F_SQ_LOCK) == 0u);
        if (__builtin_amdgcn_ballot_w64(snap) != 0ull) {
#pragma unroll
            for (int i = 0; i < NFF; ++i) { sffc[i] = snap ? ffc[i] : sffc[i]; sffw[i] = snap ? ffw[i] : sffw[i]; }
#pragma unroll
            for (int i = 0; i < NFB; ++i) { sfbc[i] = snap ? fbc[i] : sfbc[i]; sfbw[i] = snap ? fbw[i] : sfbw[i]; }
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(drop) != 0ull, 0)) end_masked(drop);      // lost sync: receiver.rs:410-422 -> end()
        const uint32_t after = L.flags & (F_AGC_LOCKED | F_BW_LOCKED);
        *fb = (valid & ((after != before) | drop)) ? (FB_VALID | ((after & F_AGC_LOCKED) ? FB_AGC : 0u) | ((after & F_BW_LOCKED) ? FB_BW : 0u) | (drop ? FB_END : 0u) | (off << 8)) : 0u;
        return valid ? (YM_VALID | (ready ? YM_READY : 0u) | (adjusted ? YM_ADJUSTED : 0u) | (reading ? YM_READING : 0u) | (drop ? YM_DROP : 0u) |
                        (byte << YM_BYTE_SHIFT) | (off << YM_OFF_SHIFT)) : 0u;
    }
    // One symbol.  Returns the word for Y2; *fb = the feedback word for S and E (0: none).
    __device__ __forceinline__ uint32_t symbol(const Params &P, float zero, float sym, uint32_t off, uint32_t *fb)
    {
        const uint32_t before = L.flags & (F_AGC_LOCKED | F_BW_LOCKED);
        // --- CodeAndPowerSquelch::input rx/codesquelch.rs:228-304
        const uint32_t slot = (2u * nsym) & 63u;
        const float eq_in0 = nxt0, eq_in1 = nxt1;          // history slots slot + 16 / + 17, requested a symbol ago
        {
            float *pn = hptr((slot + 18u) & 63u), *pw = hptr(slot);
            nxt0 = pn[0]; nxt1 = *reinterpret_cast<float *>(reinterpret_cast<char *>(pn) + hrow4);
            pw[0] = zero; *reinterpret_cast<float *>(reinterpret_cast<char *>(pw) + hrow4) = sym;
        }
        const uint32_t fill = min(64u, L.sq_fill + 2u);
        L.sq_fill = fill;
        L.sq_data = (L.sq_data >> 1) | ((sym >= 0.0f) ? 0x80000000u : 0u);      // CodeCorrelator::search :421-428
        const uint32_t nerr = __popc(P.sync_word ^ L.sq_data);
        const float pwr = fmaxf(__builtin_fmaf(__builtin_fmaf(sym, sym, -L.sq_power), P.sq_bw, L.sq_power), 0.0f);   // PowerTracker::track :483-488
        L.sq_power = pwr;
        L.sq_phist = (L.sq_phist << 1) | ((pwr >= P.sq_power_close) ? 1u : 0u);
        nsym += 1u;
        const int32_t clock_before = L.sq_clock;           // byte clock before this symbol (-1: no sync)
        const bool full = fill >= 64u;                     // sample_history.is_full() :237
        const bool locked = (L.flags & F_SQ_LOCK) != 0u;
        const bool sync_now = full & !locked & (nerr <= P.sq_max_errors) & (pwr >= P.sq_power_open);     // :244-265
        const bool adjusted = sync_now & (clock_before != 0);
        const bool drop = full & !sync_now & (clock_before >= 0) & ((L.sq_phist & 0x80000000u) == 0u);  // :266-273
        int32_t clk = sync_now ? 0 : clock_before;
        clk = drop ? -1 : clk;
        const bool ready = full & (clk == 0);              // :277-303 byte clock
        const bool reading = full & (clk > 0);
        clk = ready ? 1 : (reading ? ((clk + 1) & 7) : clk);
        L.sq_clock = clk;
        // Equalizer schedule (same_dev_common.h rx_symbol): while a channel is in byte sync, symbol j of the next byte is
        // equalized during the j-th symbol before the byte is due -- the reference equalizes all eight when it is due,
        // rx/codesquelch.rs:283-299, over the OLDEST 16 history samples, which were known 24 symbols earlier.
        if ((clock_before >= 0) & !adjusted & (ready | reading)) {
            const uint32_t j = (uint32_t)(clock_before + 7) & 7u;   // clock 1..7 -> symbol 0..6, clock 0 -> 7
            const uint32_t ebit = eq_symbol_relaxed<NFF, NFB>(P, L, ffc, ffw, fbc, fbw, eq_in0, eq_in1);
            uint32_t bits = (j == 0u) ? 0u : ((L.flags & F_EQ_BITS_MASK) >> F_EQ_BITS_SHIFT);
            bits |= ebit << j;
            L.flags = (L.flags & ~F_EQ_BITS_MASK) | (bits << F_EQ_BITS_SHIFT);
        }
        uint32_t byte = (L.flags & F_EQ_BITS_MASK) >> F_EQ_BITS_SHIFT;
        if (ready & adjusted) {
            // sync acquired or the byte clock re-aligned (receiver.rs:423-446): lock AGC and loop bandwidth, train on the sync
            // word over the oldest 16 samples of the history (rx/codesquelch.rs:288-294)
            if (clock_before >= 0) {                       // drop the symbols equalized ahead for a byte the reference never forms
#pragma unroll
                for (int i = 0; i < NFF; ++i) { ffc[i] = sffc[i]; ffw[i] = sffw[i]; }
#pragma unroll
                for (int i = 0; i < NFB; ++i) { fbc[i] = sfbc[i]; fbw[i] = sfbw[i]; }
            }
            L.flags |= F_AGC_LOCKED | F_BW_LOCKED;
            L.flags = (L.flags & ~F_EQ_MODE_MASK) | (2u << F_EQ_MODE_SHIFT);   // equalizer.train()
            L.eq_word = P.sync_word; L.eq_count = 0;
            const uint32_t head = (2u * nsym) & 63u;       // oldest sample
            float samples[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) samples[i] = *hptr((head + (uint32_t)i) & 63u);
            byte = 0u;
#pragma unroll 1
            for (int b = 0; b < 8; ++b) byte |= eq_symbol_relaxed<NFF, NFB>(P, L, ffc, ffw, fbc, fbw, samples[2 * b], samples[2 * b + 1]) << b;
        }
        // a re-alignment is still possible: remember the equalizer as of this completed byte (a byte the framer ends the burst
        // on takes a snapshot nobody restores: the end() that follows leaves no byte clock to re-align)
        if (ready & !(L.flags & F_SQ_LOCK)) {
#pragma unroll
            for (int i = 0; i < NFF; ++i) { sffc[i] = ffc[i]; sffw[i] = ffw[i]; }
#pragma unroll
            for (int i = 0; i < NFB; ++i) { sfbc[i] = fbc[i]; sfbw[i] = fbw[i]; }
        }
        if (drop) end();                                   // lost sync: receiver.rs:410-422 -> end()
        const uint32_t after = L.flags & (F_AGC_LOCKED | F_BW_LOCKED);
        *fb = (after != before || drop) ? (FB_VALID | ((after & F_AGC_LOCKED) ? FB_AGC : 0u) | ((after & F_BW_LOCKED) ? FB_BW : 0u) | (drop ? FB_END : 0u) | (off << 8)) : 0u;
        return YM_VALID | (ready ? YM_READY : 0u) | (adjusted ? YM_ADJUSTED : 0u) | (reading ? YM_READING : 0u) | (drop ? YM_DROP : 0u) |
               (byte << YM_BYTE_SHIFT) | (off << YM_OFF_SHIFT);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Y2: framer and link state of the symbol Y1 handed over (receiver.rs:410-471, rx/framing.rs:109-186), the relaxed-only cut of
// same_dev_common.h's framer_end / framer_feed / rx_symbol tail: the same decisions as ONE straight line committed by selects
// (the strict kernels' form -- restart arm, ordinary arm and "no byte" arm as three regions, the feed expanded twice -- cost
// this wavefront ~35 branches and their register shuffling a step; it is the role with the longest step).  Returns the LinkState
// kind; *burst_len = the length of a burst that has just ended (its bytes are still in `row`), *fb2 = what goes back to Y1
// (squelch.lock(true)) and, for an end(), to S and E as well; *emit = report the link state (receiver.rs:246-253).
// ---------------------------------------------------------------------------------------------------------------------
struct SymFramer { uint32_t st, last, word, count, invalid, len; };      // Framer's state (0 idle, 1 prefix search, 2 data read), the last reported link kind
__device__ __forceinline__ uint32_t sym_framer_step(const Params &P, SymFramer &F, uint8_t *row, uint32_t m, uint32_t *burst_len, uint32_t *fb2, bool *emit)
{
    const bool valid = (m & YM_VALID) != 0u, ready = (m & YM_READY) != 0u, reading = (m & YM_READING) != 0u;
    const uint32_t byte = (m >> YM_BYTE_SHIFT) & 0xffu;
    // Framer::input's restart arm rx/framing.rs:109-123: end() -- a burst being read ends here -- then a prefix search from scratch
    const bool restart = (m & (YM_READY | YM_ADJUSTED)) == (YM_READY | YM_ADJUSTED);
    const uint32_t st = F.st;
    const bool burst_at_restart = restart & (st == 2u);
    const uint32_t st1 = restart ? 1u : st;
    const uint32_t word0 = restart ? 0u : F.word, count0 = restart ? 0u : F.count;
    // PrefixSearch rx/framing.rs:128-150
    const bool searching = ready & (st1 == 1u), rd = ready & (st1 == 2u);
    const uint32_t word = (word0 << 8) | byte, count = count0 + 1u;
    const uint32_t e0 = __popc(word ^ 0x5a435a43u);          // "ZCZC" rx/framing.rs:235-243
    const uint32_t e1 = __popc(word ^ 0x4e4e4e4eu);          // "NNNN"
    const bool found = searching & (min(e0, e1) <= P.fr_max_prefix_errors);
    const bool give_up = searching & !found & (count > 21u); // PREFIX_SEARCH_LEN :201
    // DataRead :153-163
    const uint32_t invalid = F.invalid + (is_allowed_byte(byte) ? 0u : 1u);
    const bool over = rd & (invalid > P.fr_max_invalid);
    const bool keep = rd & !over;
    F.word = searching ? word : word0;
    F.count = searching ? count : count0;
    F.invalid = found ? 0u : (rd ? invalid : F.invalid);
    if (found) *reinterpret_cast<uint32_t *>(row) = __builtin_bswap32(word);      // the prefix as received seeds the burst :136-138 (rows are 288 bytes apart)
    if (keep && F.len < (uint32_t)kBurstCap) row[F.len] = (uint8_t)byte;
    // Reading without a byte: framer.state(); NoCarrier / DroppedCarrier: framer.end()  receiver.rs:410-422.  (The symbol after
    // an end() arrives as "reading" -- Y1 had not heard of it -- and finds the framer idle: NoCarrier, as in the reference.)
    const bool was_reading_burst = valid & !ready & !reading & (st == 2u);
    *burst_len = (over | burst_at_restart | was_reading_burst) ? F.len : 0u;       // Framer::end() rx/framing.rs:174-186: the burst without this byte
    F.len = found ? 4u : (keep ? F.len + 1u : F.len);
    const uint32_t nst = found ? 2u : ((give_up | over) ? 0u : st1);
    const uint32_t link_ready = restart ? (burst_at_restart ? 3u : 1u) : (over ? 3u : nst);
    const uint32_t link_quiet = reading ? st : (st == 2u ? 3u : 0u);
    const uint32_t link = ready ? link_ready : link_quiet;
    F.st = ready ? nst : ((valid & !reading) ? 0u : st);
    // squelch.lock(true) receiver.rs:462; end() receiver.rs:466-470
    const uint32_t fb_end = FB_VALID | FB_END | (((m >> YM_OFF_SHIFT) & kSymOffMask) << 8);
    *fb2 = ready ? (link == 2u ? (uint32_t)FB_SQLOCK : ((link == 0u || link == 3u) ? fb_end : 0u)) : 0u;
    // receiver.rs:246-253: report on change; a Burst always differs from its predecessor
    *emit = valid & ((link != F.last) | (link == 3u));
    F.last = *emit ? link : F.last;
    return link;
}

// ---------------------------------------------------------------------------------------------------------------------
// Issue priority of a role (s_setprio).  Three wavefronts share a SIMD and the arbiter prefers the OLDEST: left alone, the second
// half's filter wavefronts (A, E -- the longest chain of a step: filters -> timing update -> where the next symbol falls) sit behind
// the first half's AGC / DC / squelch wavefronts, which have slack to spare, and the whole workgroup's step is theirs (measured:
// E 3 690 clk of a 4 150-clk step in the second half against 2 650 in the first).  Profile builds pick other tables with
// SAME_PIPE_PRIO bits 12-13 (1: no priorities).
#ifndef SYM_PRIOS
#define SYM_PRIOS 0x0031322      /* S T A A(events) E Y1 Y2, one hex digit each */
#endif
constexpr int sym_prio_of(int k) { return (int)((SYM_PRIOS >> (4 * (6 - k))) & 3); }
// 44.1 / 48 kHz: one group per CU, S and Y1 share a SIMD and so do T and Y2 -- and there the SAMPLE roles carry the step (72 samples
// of AGC / DC blocker against one symbol): they go first
#ifndef SYM_PRIOS_HI
#define SYM_PRIOS_HI 0x3321200   /* S T A A(events) E Y1 Y2 */
#endif
template <int NT> constexpr int sym_prio(int k) { return NT == 42 ? sym_prio_of(k) : (int)((SYM_PRIOS_HI >> (4 * (6 - k))) & 3); }
template <int PRIO> __device__ __forceinline__ void sym_setprio(const Params &P)
{
#ifdef SAME_PROFILE
    const uint32_t variant = ((uint32_t)P.knob_prio >> 12) & 3u;
    if (variant == 1u) return;
    if (variant == 2u) { __builtin_amdgcn_s_setprio(PRIO >= 2 ? 3 : 0); return; }
#else
    (void)P;
#endif
    __builtin_amdgcn_s_setprio(PRIO);
}

// next_fire_count (same_fast_common.h) for a clock at zero, in closed form: the sample clock fires at the first count c >= 1 with
// fl(s - c) < 0.5 (receiver.rs:352-353).  For counts next to s (s / 2 <= c <= 2 s) the difference is exact in f32, and so is
// s - 0.5 (a multiple of s's ulp), so the condition is c > s - 0.5, i.e. c = floor(s - 0.5) + 1 -- three operations on the chain
// timing update -> where the next instant falls -> next filter, where the search from floor(s) - 1 took a dozen.
__device__ __forceinline__ float sym_next_fire(float s)
{
    return fmaxf(floorf(s - 0.5f) + 1.0f, 1.0f);
}

// The kernel's argument list as a struct: the kernarg segment has this layout (natural alignment, in order).  The roles read
// State / Output / PipeChunks from it WHEN THEY NEED THEM -- before and after their loops -- through a pointer the compiler cannot
// see through.  Taken as ordinary by-value parameters, all of the ~70 pointers are loaded at the kernel's entry and stay live
// across every role's loop for the stores at its end: a hundred scalar registers spilled into vector-register lanes, and every
// use of one of them inside a loop a v_readlane -- vector issue slots, the resource this kernel is short of (345 of them in the
// six-role build before this).
struct SymKernArgs { Params P; State S; Output O; const float4 *taps; const void *x; uint32_t n_blocks; uint64_t counter0; PipeChunks K; };
typedef const __attribute__((address_space(4))) char *sym_kernarg_ptr;
template <typename T> __device__ __forceinline__ T sym_fresh_arg(sym_kernarg_ptr ka, size_t off)
{
    uint64_t p = (uint64_t)(uintptr_t)ka + off;
    asm volatile("" : "+s"(p));
    T r;
    __builtin_memcpy(&r, reinterpret_cast<const __attribute__((address_space(4))) void *>(p), sizeof(T));      // scalar loads of the fields that are used
    return r;
}

// Six role-wavefronts per 64 state columns, and TWO such groups per workgroup: twelve wavefronts land three per SIMD whatever
// else runs (the dispatcher deals a workgroup's wavefronts round the four SIMDs in turn), while two six-wavefront workgroups
// never share a CU -- the second one's wavefronts would pile four-deep on the SIMDs that already hold two, past the register
// file (tools/ubench_wave_place.hip: one resident workgroup per CU, half the machine).  The two halves share nothing but the
// step barrier; a half that is done simply ends (s_barrier counts the surviving wavefronts only).
// 44.1 / 48 kHz: ONE group per workgroup and CU (SymGeom::HALVES; the window ring alone is 131-134 KB): six wavefronts on four SIMDs,
// up to 256 registers each.
constexpr int kSymRoles = 6;
template <int NT, int NFF, int NFB, typename SampleT, int CMODE>
__global__ __launch_bounds__(SymGeom<NT>::HALVES * kSymRoles * kWave, NT == 42 ? 3 : 2) void demod_sym_kernel(Params P, State S_arg_, Output O_arg_, const float4 *__restrict__ taps,
                                                                         const SampleT *__restrict__ x, uint32_t n_blocks, uint64_t counter0,
                                                                         PipeChunks K)
{
    (void)S_arg_; (void)O_arg_;                                     // read through the kernarg segment where they are needed (SymKernArgs)
    const sym_kernarg_ptr ka = (sym_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    auto fresh_state = [&]() __attribute__((always_inline)) -> State { return sym_fresh_arg<State>(ka, offsetof(SymKernArgs, S)); };
    auto fresh_output = [&]() __attribute__((always_inline)) -> Output { return sym_fresh_arg<Output>(ka, offsetof(SymKernArgs, O)); };
    using LY = SymLayout<NT>;
    constexpr int kB = LY::B, RING = LY::RING, kSB = LY::SB, kNSUB = LY::NSUB;
    constexpr uint32_t kSymHalves = (uint32_t)LY::HALVES;
    constexpr uint32_t LP = kWave;
    static_assert(CMODE == 0 || std::is_same<SampleT, float>::value, "channel-major streams are f32");
    if constexpr (CMODE == 0) { K.col_row0 = nullptr; K.col_perm = nullptr; }       // (the host launches this build for nothing else)
    extern __shared__ float lds_all[];
    const uint32_t lane = threadIdx.x & (kWave - 1u);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // wavefront -> (half, role), 4 bits each.  Wavefronts w, w + 4, w + 8 share a SIMD; which roles do is worth +-2-5 % (profile
    // builds pick a table with SAME_PIPE_PRIO bits 14-15): the order of the launch by role is within 1 % of the best found.
    constexpr uint64_t kDeal0 = 0xdcba98543210ull;     // S0 T0 A0 E0 | Y1_0 Y2_0 S1 T1 | A1 E1 Y1_1 Y2_1
#ifdef SAME_PROFILE
    const uint32_t deal = ((uint32_t)P.knob_prio >> 14) & 3u;
    constexpr uint64_t kDeal1 = 0x805d192ab3c4ull;     // Y1_0 Y1_1 E0 E1 | A1 A0 T1 T0 | Y2_1 Y2_0 S0 S1
    constexpr uint64_t kDeal2 = 0x08d591a2b3c4ull;     // Y1_0 Y1_1 E0 E1 | A0 A1 T0 T1 | Y2_0 Y2_1 S1 S0
    constexpr uint64_t kDeal3 = 0x2ad51908b3c4ull;     // Y1_0 Y1_1 E0 E1 | S1 S0 T1 T0 | Y2_0 Y2_1 A1 A0
    const uint64_t tbl = deal == 0u ? kDeal0 : (deal == 1u ? kDeal1 : (deal == 2u ? kDeal2 : kDeal3));
#else
    constexpr uint64_t tbl = kDeal0;
#endif
    // (one group per workgroup: wavefronts 0 .. 5 are S T A E Y1 Y2; SIMDs 0 and 1 hold two of them -- S + Y1 and T + Y2, the sample
    // roles each beside a symbol role with slack)
    const uint32_t hr = LY::HALVES == 2 ? (uint32_t)(tbl >> (4u * wave)) & 15u : wave;
    const uint32_t half = LY::HALVES == 2 ? hr >> 3 : 0u;
    const uint32_t role = hr & 7u;                                                             // 0 S, 1 T, 2 A, 3 E, 4 Y1, 5 Y2
    const uint32_t C = P.n_channels;
    const uint32_t vwg = blockIdx.x * kSymHalves + half;                              // this half's group of 64 state columns
    if (vwg * kWave >= C) return;                                                              // (an odd number of groups: the last workgroup's second half)
    float *lds = lds_all + half * (uint32_t)(LY::lds_bytes / sizeof(float));
    // state column of this lane (time-parallel launches may permute them: pieces of similar length share a group)
    const uint32_t c = (K.n_chunks > 1u && K.col_perm) ? K.col_perm[vwg * kWave + lane] : vwg * kWave + lane;
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
            avail_l = (K.whole_samples - row_abs) / (uint32_t)kSB;              // (per-lane streams: 22.05 kHz, where a step is one sub-block)
            n_blocks = K.wg_blocks[vwg];
            n_nominal = may_leave ? K.col_nominal[c] : n_blocks;
            const uint32_t row_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_abs);
            counter0 += (uint64_t)row_first;
            row_l = (int32_t)(row_abs - row_first);
        } else {
            const uint32_t wgs = K.in_channels / kWave;
            const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(vwg / wgs));
            cin = (vwg - chunk * wgs) * kWave + lane;
            may_leave = chunk + 1u < K.n_chunks;
            const uint32_t first_block = chunk * K.stride_blocks;
            x += (size_t)first_block * kB * Cin;
            counter0 += (uint64_t)first_block * kB;
            n_blocks -= first_block;
            n_nominal = may_leave ? K.nominal_blocks : n_blocks;
        }
    }
    // LDS: taps (profile builds: section marks) | mailboxes | window ring [RING + MIR][64]
    lds_u32 *mail = (lds_u32 *)(lds + LY::tap_floats);
    lds_u32 *symbox = mail;                                    // [2][5][64]  E -> Y1: header, the symbol's two soft samples (trace: timing error, period)
    lds_u32 *ybox = symbox + 2u * LY::sym_words;               // [2][64]     Y1 -> Y2
    lds_u32 *fb1box = ybox + 2u * kWave;                       // [2][64 + flag]  Y1 -> S, E: the lock at sync; end() on a lost sync
    lds_u32 *fb2box = fb1box + 2u * LY::fb_words;              // [2][64 + flag]  Y2 -> S, E, Y1: end() from the framer; -> Y1: squelch.lock; flag word bit 1: leave
    lds_u32 *iobox = fb2box + 2u * LY::fb_words;               // [2][3][64]  Y2 -> A: link word, burst-pool slot, burst length
    lds_u32 *phasebox = iobox + 2u * LY::io_words;             // [64] E's final TED phase bit
    lds_u32 *againbox = phasebox + kWave;                      // [64] A's final F_TICK_AGAIN bit
    lds_u32 *flagbox = againbox + kWave;                       // [64] Y1's final flag bits
    lds_u32 *posbox = flagbox + kWave;                         // [2][64] ring slot of the FIRST instant of step s's symbol (parity s & 1; -1: none)
    lds_u32 *sabox = posbox + 2u * kWave;                      // [64] that instant's soft sample, from A
    lds_u32 *tkbox = sabox + kWave;                            // [kTickRing][64] u64 deadlines, then [64] their count (A's own)
    lds_u32 *donebox = tkbox + kIoRingWords;                   // [64] Y2 -> T: the lane's piece has handed over (its input is no longer needed)
    lds_u32 *chunkbox = fb1box + kWave + 2u;                   // [2], in the first feedback box's padding: A's event-log run
    lds_u32 *seqbox = fb1box + kWave + 4u;                     // A's progress with the first instants' filters: 2 * step + 1
    static_assert((size_t)(2u * LY::sym_words + 2u * kWave + 4u * LY::fb_words + 2u * LY::io_words + 3u * kWave + 2u * kWave + kWave + kIoRingWords + kWave + kWave) <= LY::mail_words, "mailboxes");
    float *wring = lds + LY::tap_floats + LY::mail_words;     // ring slot 0
    float *wcol = wring + lane;
    const uint64_t counter1 = counter0 + (uint64_t)n_blocks * kB;
    // Steps: T computes the DC blocker of block s + 1 in step s (block 0 before the first), S the AGC of block s in step
    // s < n_blocks; E finishes symbols that end before sample 36 s in steps 1 .. n_blocks and the instants left before the
    // end of the input, one per step, in the kSymDrain steps after; Y1 runs one step behind E, Y2 one behind Y1, A's events
    // one behind Y2.
    const uint32_t last_e_step = n_blocks + kSymDrain;
    const uint32_t last_y1_step = last_e_step + 1u, last_y2_step = last_e_step + 2u, last_a_step = last_e_step + 3u;
    const uint32_t n_steps = last_e_step + 4u;
    // A symbol travels as `off` = its sample index - 36 * base, base = s_E - 2 for the step s_E <= n_blocks E finished it
    // in, n_blocks - 3 in the steps after: 0 <= off < 128.  The wavefronts behind E rebuild the index.
    auto sym_index = [&](uint32_t s_e, uint32_t off) __attribute__((always_inline)) -> int64_t {
        const int64_t base = s_e <= n_blocks ? (int64_t)s_e - 2 : (int64_t)n_blocks - 3;
        return (int64_t)kB * base + (int64_t)off;
    };
    // Synchronisation: no step barrier.  Every role publishes, once per step, ONE word -- steps completed << 16 | the step's
    // flags in the byte of its parity -- and a role starts step s when the roles it exchanges data with have completed step
    // s - 1 (kDeps: producers of what it reads, consumers of the double-buffered boxes it is about to overwrite).  One LDS
    // read per step brings all six words, i.e. the go-ahead AND the feedback flags (a barrier followed by two or three flag
    // reads cost three round trips); a wavefront waits for the ones it needs, not for the slowest of twelve, so a long step of one
    // role is absorbed by the slack of the others instead of stalling the workgroup (measured with the barrier: every role
    // waited >= 730 clk of a 3 780-clk step although the longest worked 2 900), and the two halves never meet.
    lds_u32 *prog = donebox + kWave;                               // [6]
    // 44.1 / 48 kHz (the filter wavefronts split the taps): ring slot of the symbol's SECOND instant per parity (-1: no symbol this
    // step), and A's partial sums [instant B: re m, re s, im m, im s | instant A: the same][64]
    lds_u32 *pos2box = prog + kWave + (SYM_TL_WORDS + 63u) / 64u * 64u;
    lds_u32 *partbox = pos2box + 2u * kWave;
#ifdef SAME_SYM_TL
    lds_u32 *tlbox = prog + kWave;
#endif
    // (A waiting wavefront costs vector issue slots -- the resource this kernel is short of -- with every poll: roles with slack
    // sleep longer between polls than the ones on the step's critical chain.  The poll as ONE assembly statement -- no inner loop
    // for the compiler to see -- was 1 % faster with the link layer alone and 7 % slower with the transport layer on: not kept.)
    const uint32_t prog_idx = lane < (uint32_t)kSymRoles ? lane : 0u;
    uint32_t *const err_flags = fresh_output().n_events + 2;
    // A wait that gives up (a protocol error: the launch reports SAME_EKERNEL instead of hanging the GPU) makes every later wait
    // of its wavefront give up after a few polls -- the call is void, and ~27 000 steps of three waits at half a second each
    // would be hours -- and the wavefront skips its state stores (`gave_up()`); the roles that wait for it follow the same way.
    uint32_t spin_limit = 1u << 22;
    auto gave_up = [&]() __attribute__((always_inline)) -> bool { return spin_limit != (1u << 22); };
    // The progress words only grow, and a word that shows "step s - 1 completed" carries that step's flags whenever it was read (its
    // writer cannot be two steps ahead of a reader, see flags_of_last): a wait first looks at the words this wavefront read LAST --
    // `seen`, refreshed by peek() where a role has work to hide the LDS round trip behind -- and polls only if they do not say enough
    // yet.  A self-bound role (most roles in most steps: the ones it waits for published long ago) used to spend 350-460 clk per
    // wait on a read whose answer was old news.
    // Measured (round 6, one box each): 44.1 / 48 kHz 2.18 -> 2.15 / 2.45 -> 2.42 ms; 22.05 kHz 1.885 -> 1.895 ms with the link layer
    // alone and 1.92-2.00 -> 2.05-2.18 ms launch by launch with the transport layer on (twelve wavefronts per CU: roles that start
    // sooner take issue slots from the ones on the step's critical chain) -- so the 22.05 kHz build keeps polling afresh.
    constexpr bool kPeek = NT != 42;
    uint32_t seen = 0u;
    auto peek = [&]() __attribute__((always_inline)) { if constexpr (kPeek) seen = prog[prog_idx]; };
    auto wait_for = [&](uint32_t s, uint32_t deps, uint32_t *w_y1, uint32_t *w_y2, auto nap_) __attribute__((always_inline)) {
        constexpr int NAP = decltype(nap_)::value;
        uint32_t v = seen, spins = 0;
        if constexpr (!kPeek) v = prog[prog_idx];
        for (;;) {
            const uint32_t ok = (uint32_t)__builtin_amdgcn_ballot_w64(v >= (s << 16));     // (the flags below bit 16 cannot carry)
            if ((ok & deps) == deps) break;
            if (!kPeek || spins != 0u) __builtin_amdgcn_s_sleep(NAP);
            if (++spins > spin_limit) { if (lane == 0u) atomicOr(err_flags, 4u); spin_limit = 8u; break; }
            v = prog[prog_idx];
        }
        if constexpr (kPeek) seen = v;
        // (the ordering of the roles' LDS traffic rests on these words: nothing the wait guards may be read before it -- the
        // compiler may move plain accesses across volatile ones, and the step barrier whose clobber used to stop it is gone)
        asm volatile("" ::: "memory");
        *w_y1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 4);
        *w_y2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 5);
    };
    // (Round 6: naps four times as long -- 4 / 16 -- measured equal, 1.892 / 1.806 ms against 1.87-1.88 / 1.80; and `s_wakeup` behind
    // every publish, to end the sleepers' naps at once, ends in a memory access fault on this stack (gfx950, ROCm 7.2) although the
    // assembler takes it: not used)
    constexpr std::integral_constant<int, 1> kNapShort{};
    constexpr std::integral_constant<int, 4> kNapLong{};
    // flags of step s - 1 in a progress word read at the start of step s.  Two parities: its writer may be ONE step further, never
    // two -- every reader of a role's flags is among the roles that role waits for before it publishes (S, T, A, E wait for Y2 and
    // Y2 for all of them; S, E wait for Y1 and Y1 for both)
    auto flags_of_last = [&](uint32_t w, uint32_t s) __attribute__((always_inline)) -> uint32_t { return s == 0u ? 0u : (w >> (8u * ((s - 1u) & 1u))) & 0xffu; };
    auto publish = [&](uint32_t s, uint32_t flags, uint32_t flags_before) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");                                 // (... and nothing the step wrote may sink below its progress word)
        if (lane == 0u) prog[role] = ((s + 1u) << 16) | (flags << (8u * (s & 1u))) | (flags_before << (8u * ((s + 1u) & 1u)));
    };
    enum : uint32_t { R_S = 1u, R_T = 2u, R_A = 4u, R_E = 8u, R_Y1 = 16u, R_Y2 = 32u };
    enum : uint32_t { Y1F_ANY = 1u, Y2F_END = 1u, Y2F_LEAVE = 2u, Y2F_LOCK = 4u };
    // What a role waits for, and where in its step (a wait names the roles that must have completed step s - 1):
    //   S  before anything: T (the DC outputs of its block), A and E (done with the mirrored slots it rewrites), Y1 and Y2 (feedback)
    //   T  before the DC blocker: A and E (done with the ring block it overwrites); before publishing: Y2 (hand-over flag)
    //   A  before its filter: S (blocks), E (positions; E is done with the last soft sample); before the events: Y2 (link words)
    //   E  before its filter: S; before the timing updates: A (done with the positions), Y1 (done with the symbol box; feedback), Y2
    //   Y1 before the symbol: E (symbols), Y2 (feedback; done with the word box); before posting feedback: S (has read the box of two steps ago)
    //   Y2 before the framer: Y1 (words); before posting: everybody (has read the flags and feedback of two steps ago; A the link box)
    constexpr uint32_t kDepsS = R_T | R_A | R_E | R_Y1 | R_Y2;
    if (role == 0u && lane < (uint32_t)kSymRoles) prog[lane] = 0u;

    if (role == 0u) {
        // ------------------------------------------ S: AGC of block s, in place ------------------------------------------
        sym_setprio<sym_prio<NT>(0)>(P);
        SymAgc<NT> M;
        { const State S = fresh_state(); M.load(P, S, c, C, counter0, wcol); }
        lds_barrier();                                                 // prologue: block 0's DC outputs are in the ring, every box is initialised
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        bool left = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            uint32_t w1, w2;
            SYM_TRACE(0, s, 0);
            wait_for(s, kDepsS, &w1, &w2, kNapLong);
            SYM_TRACE(0, s, 1);
            P3_LAP(p3_wait);
            const uint32_t f1 = flags_of_last(w1, s), f2 = flags_of_last(w2, s);
            if (f2 & Y2F_LEAVE) stop_at = s;
            // Feedback of step s - 1 (agc.lock receiver.rs:431, 480), from this block on.  Y2's (an end() at an older symbol) first.
            if ((f2 & Y2F_END) && s >= 1u) {
                if (fb2box[((s - 1u) & 1u) * LY::fb_words + lane] & FB_VALID) M.locked = false;
            }
            if ((f1 & Y1F_ANY) && s >= 1u) {
                const uint32_t v = fb1box[((s - 1u) & 1u) * LY::fb_words + lane];
                if (v & FB_VALID) {
                    const bool new_locked = (v & FB_AGC) != 0u;
                    if (new_locked && !M.locked) {
                        // the gain freezes at the value it had after the symbol's sample (Y1 saw it in step s - 1, E finished it in s - 2)
                        const int64_t idx = sym_index(s - 2u, v >> 8);
                        const uint32_t b = (uint32_t)(idx / kB);
                        M.gain = M.gain_at(P, wcol, b, (int)(idx - (int64_t)b * kB));
                    }
                    M.locked = new_locked;
                }
                P3_LAP(p3_fb);
            }
            if (s < n_blocks && !PROF_SKIP(P, 64)) M.block(P, wcol, s);
            SYM_TRACE(0, s, 2);
            peek();                                                        // (the next step's first wait looks at these words)
            publish(s, 0u, 0u);
            P3_LAP(p3_work);
            if (s == stop_at) { left = true; break; }
        }
        SYM_REPORT(0);
        if (left || gave_up()) return;                                              // handed over: this chunk's state is not needed
        { const State S = fresh_state(); M.store(P, S, c, C, counter1, wcol); }
    } else if (role == 1u) {
        // ------------------------------------------ T: input prefetch and DC blocker of block s + 1 --------------------
        sym_setprio<sym_prio<NT>(1)>(P);
        if constexpr (NT != 42) {
            // 44.1 / 48 kHz: SymDcRot -- two sub-blocks a step, three input buffers in rotation (sub-block q in buffer q mod 3)
            static_assert(CMODE == 0 && kNSUB == 2, "time-major rows, two sub-blocks a step");
            SymDcRot<SampleT, NT> D;
            const uint32_t n_sub = n_blocks * (uint32_t)kNSUB;
            { const State S = fresh_state(); D.load(S, x, c, C, cin, Cin, counter0, n_sub); }
            donebox[lane] = 0u;
            D.template sub<0>(P, wcol, x, 0u, n_sub, cin, Cin);          // prologue: block 0's DC outputs
            D.template sub<1>(P, wcol, x, 1u, n_sub, cin, Cin);
            lds_barrier();
            P3_T0();
            uint32_t stop_at = 0xffffffffu;
            auto step = [&](uint32_t s, auto ph_) __attribute__((always_inline)) -> bool {
                constexpr int PH = 2 * decltype(ph_)::value;              // block s + 1 = sub-blocks 2 (s + 1), 2 (s + 1) + 1: PH = 2 ((s + 1) mod 3)
                uint32_t w1, w2;
                SYM_TRACE(1, s, 0);
                wait_for(s, R_A | R_E, &w1, &w2, kNapLong);                          // the filters are done with the ring block this step overwrites
                SYM_TRACE(1, s, 1);
                P3_LAP(p3_wait);
                if (s + 1u < n_blocks && !PROF_SKIP(P, 128)) {
                    D.template sub<PH>(P, wcol, x, 2u * (s + 1u), n_sub, cin, Cin);
                    D.template sub<PH + 1>(P, wcol, x, 2u * (s + 1u) + 1u, n_sub, cin, Cin);
                }
                if (may_leave) {                                                     // (wave-uniform; only a time-parallel piece hands over)
                    wait_for(s, R_Y2, &w1, &w2, kNapLong);                           // has Y2 called the hand-over?
                    if (flags_of_last(w2, s) & Y2F_LEAVE) stop_at = s;
                }
                SYM_TRACE(1, s, 2);
                peek();                                                        // (the next step's first wait looks at these words)
                publish(s, 0u, 0u);
                P3_LAP(p3_work);
                return s == stop_at;
            };
            bool left = false;
            for (uint32_t s = 0; s < n_steps && !left; s += 3u) {
                left = step(s, std::integral_constant<int, 1>{});
                if (!left && s + 1u < n_steps) left = step(s + 1u, std::integral_constant<int, 2>{});
                if (!left && s + 2u < n_steps) left = step(s + 2u, std::integral_constant<int, 0>{});
            }
            SYM_REPORT(1);
            if (left || gave_up()) return;
            { const State S = fresh_state(); D.store(S, c, C, counter1, n_sub - 1u); }
            return;
        } else {
        using Dc = SymDc<SampleT, CMODE, NT>;
        Dc D;
        D.xl = xl; D.avail = avail_l;
        // (T counts SUB-BLOCKS of 36 samples: a step's block is kNSUB of them, sub-block q in buffer q & 1)
        const uint32_t n_sub = n_blocks * (uint32_t)kNSUB;
        { const State S = fresh_state(); D.load(S, x, c, C, cin, Cin, counter0, n_sub); }
        donebox[lane] = 0u;
        // time-major rows: the loads of sub-block q + 1 go out just before sub-block q (waiting in buffer PB) is computed
        auto sub_tm = [&](uint32_t q, auto pb_) __attribute__((always_inline)) {
            typename Dc::Pairs X;
            if constexpr (decltype(pb_)::value == 0) { D.request(D.xb, x, q + 1u, n_sub, cin, Cin); D.template take<0>(D.xa, X); }
            else { D.request(D.xa, x, q + 1u, n_sub, cin, Cin); D.template take<0>(D.xb, X); }
            D.block(P, wcol, X, q);
        };
        // prologue: block 0's DC outputs
        {
            if constexpr (CMODE == 1) {
                typename Dc::Pairs X0;
                D.request(D.xb, x, 1u, n_sub, cin, Cin);                   // (clamped to the lane's stream)
                D.template take<9>(D.xa, X0);
                D.block(P, wcol, X0, 0u);
                D.request(D.xa, x, 2u, n_sub, cin, Cin);
            } else {
                sub_tm(0u, std::integral_constant<int, 0>{});
                if constexpr (kNSUB == 2) sub_tm(1u, std::integral_constant<int, 1>{});
            }
        }
        lds_barrier();
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        auto step = [&](uint32_t s, auto buf) __attribute__((always_inline)) -> bool {
            constexpr int BUF = decltype(buf)::value;                  // one sub-block per step: block s + 1 waits in buffer BUF = (s + 1) & 1
            uint32_t w1, w2;
            SYM_TRACE(1, s, 0);
            wait_for(s, R_A | R_E, &w1, &w2, kNapLong);                          // the filters are done with the ring block this step overwrites
            SYM_TRACE(1, s, 1);
            P3_LAP(p3_wait);
            if constexpr (CMODE == 1) D.done = donebox[lane] != 0u;
            if (s + 1u < n_blocks && !PROF_SKIP(P, 128)) {
                if constexpr (CMODE == 1) {
                    typename Dc::Pairs X;
                    // in flight: this block's loads and the next one's; block s + 3's go out when this block's registers are free
                    if constexpr (BUF == 0) { D.template take<9>(D.xa, X); D.block(P, wcol, X, s + 1u); D.request(D.xa, x, s + 3u, n_sub, cin, Cin); }
                    else { D.template take<9>(D.xb, X); D.block(P, wcol, X, s + 1u); D.request(D.xb, x, s + 3u, n_sub, cin, Cin); }
                } else if constexpr (kNSUB == 1) {
                    sub_tm(s + 1u, buf);
                } else {
                    sub_tm(2u * (s + 1u), std::integral_constant<int, 0>{});
                    sub_tm(2u * (s + 1u) + 1u, std::integral_constant<int, 1>{});
                }
            }
            if (may_leave) {                                                     // (wave-uniform; only a time-parallel piece hands over)
                wait_for(s, R_Y2, &w1, &w2, kNapLong);                           // has Y2 called the hand-over?
                if (flags_of_last(w2, s) & Y2F_LEAVE) stop_at = s;
            }
            SYM_TRACE(1, s, 2);
            peek();                                                        // (the next step's first wait looks at these words)
            publish(s, 0u, 0u);
            P3_LAP(p3_work);
            return s == stop_at;
        };
        bool left = false;
        for (uint32_t s = 0; s < n_steps && !left; s += 2u) {
            left = step(s, std::integral_constant<int, 1>{});          // block s + 1 is odd when s is even
            if (!left && s + 1u < n_steps) left = step(s + 1u, std::integral_constant<int, 0>{});
        }
        SYM_REPORT(1);
        if (left || gave_up()) return;
        { const State S = fresh_state(); D.store(S, c, C, counter1); }
        }
    } else if (role == 2u) {
        // ------------------------------------------ A: the matched-filter pair at the FIRST instant of the step's symbol; link events + wake-ups ----
        sym_setprio<sym_prio<NT>(2)>(P);
        const uint32_t wcol_lds = lds_addr(wcol);
        std::conditional_t<NT == 42, SymTaps<42>, SymHalfTaps<NT, 0, NT / 4>> TP;       // (44.1 / 48 kHz: the first half of the tap pairs)
        TP.load(taps);
        Lane L;
        IoCtxLds X;
        X.chunk = chunkbox;
        if (lane == 0u) { chunkbox[0] = 0u; chunkbox[1] = kEvChunk; seqbox[0] = 0u; }      // nothing reserved yet; no pass posted yet
        X.pending_slot = 0xffffffffu;
        X.tk = tkbox + lane;
        {
            const State S = fresh_state();
            lane_load(L, S, c);      // the event half uses sq_symbols, tk_next, tk_last, wake_*, F_TICK_AGAIN
            X.ring_load(P, S, c);
        }
        const State Snone{};         // (the event log's context keeps what it needs in LDS; nothing of the state arrays)
        Output O{};
        { const Output Of = fresh_output(); O.events = Of.events; O.n_events = Of.n_events; O.event_cap = Of.event_cap; }
        lds_barrier();                                                 // prologue
        P3_T0();
        uint32_t stop_at = 0xffffffffu;
        bool left = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            uint32_t w1, w2;
            SYM_TRACE(2, s, 0);
            wait_for(s, R_S | R_E, &w1, &w2, kNapShort);                          // S's blocks, E's positions (and E is done with the last soft sample)
            SYM_TRACE(2, s, 1);
            P3_LAP(p3_wait);
            // the filters at the positions E posted (E takes the second instant and waits for this one's soft sample)
            if (s >= 1u && s <= last_e_step) {
                const uint32_t n1 = posbox[(s & 1u) * kWave + lane];
                if constexpr (NT == 42) {
                    if (__builtin_amdgcn_ballot_w64(n1 != 0xffffffffu) != 0ull) {
                        // (a profile build's knock-out skips the filter, never the hand-over E waits for)
                        const float sa1 = PROF_SKIP(P, 256) ? 0.0f : TP.template demod<RING>(wcol_lds, n1 == 0xffffffffu ? 0 : (int)n1);
                        sabox[lane] = __float_as_uint(sa1);
                        if (lane == 0u) seqbox[0] = 2u * s + 1u;       // (LDS operations of a wavefront stay in order)
                    }
                } else {
                    // this wavefront's half of the taps at both instants of every lane's symbol (lanes without one: slot 0, unused)
                    const uint32_t n2 = pos2box[(s & 1u) * kWave + lane];
                    if (__builtin_amdgcn_ballot_w64(n2 != 0xffffffffu) != 0ull) {
                        float2v r2 = {0.0f, 0.0f}, i2 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f}, i1 = {0.0f, 0.0f};
                        if (!PROF_SKIP(P, 256)) {
                            TP.template partial<RING>(wcol_lds, n2 == 0xffffffffu ? 0 : (int)n2, r2, i2);
                            if (__builtin_amdgcn_ballot_w64(n1 != 0xffffffffu) != 0ull) TP.template partial<RING>(wcol_lds, n1 == 0xffffffffu ? 0 : (int)n1, r1, i1);
                        }
                        lds_u32 *pb = partbox + lane;
                        pb[0] = __float_as_uint(r2.x); pb[kWave] = __float_as_uint(r2.y); pb[2 * kWave] = __float_as_uint(i2.x); pb[3 * kWave] = __float_as_uint(i2.y);
                        pb[4 * kWave] = __float_as_uint(r1.x); pb[5 * kWave] = __float_as_uint(r1.y); pb[6 * kWave] = __float_as_uint(i1.x); pb[7 * kWave] = __float_as_uint(i1.y);
                        if (lane == 0u) seqbox[0] = 2u * s + 1u;
                    }
                }
            }
            SYM_TRACE(2, s, 3);
            sym_setprio<sym_prio<NT>(3)>(P);
            wait_for(s, R_Y2, &w1, &w2, kNapLong);                               // Y2's link words; has it called the hand-over?
            if (flags_of_last(w2, s) & Y2F_LEAVE) stop_at = s;
            // the link event and the wake-ups of what Y2 handed over in the last step
            if (s >= 4u && s <= last_a_step && !PROF_SKIP(P, 8)) {
                const lds_u32 *io = iobox + ((s - 1u) & 1u) * LY::io_words + lane;
                const uint32_t io0 = io[0];
                if (io0 & 1u) {
                    L.sq_symbols += 1;         // as the squelch counted it (rx/codesquelch.rs:232)
                    const uint32_t link = (io0 >> 1) & 3u, off = (io0 >> 4) & kSymOffMask;
                    const bool burst = (io0 & 8u) != 0u && link == 3u;
                    uint32_t burst_len = 0;
                    if (burst) { X.pending_slot = io[kWave]; burst_len = io[2 * kWave]; }
                    // the counter is that of the sample after the symbol's
                    const uint64_t counter = counter0 + (int64_t)row_l + (uint64_t)sym_index(s - 3u, off) + 1u;
                    symbol_io(P, L, Snone, O, X, c, link, (io0 & 8u) != 0u, counter, burst_len);
                }
            }
            if (s + 1u == n_steps) againbox[lane] = L.flags & F_TICK_AGAIN;      // (Y2 merges the flag bits)
            SYM_TRACE(2, s, 2);
            peek();                                                        // (the next step's first wait looks at these words)
            publish(s, 0u, 0u);
            sym_setprio<sym_prio<NT>(2)>(P);
            P3_LAP(p3_work);
            if (s == stop_at) { left = true; break; }
        }
        SYM_REPORT(2);
        X.retire(O, lane, kWave);
        if (left || gave_up()) return;
        {
            const State S = fresh_state();
            S.tk_next[c] = L.tk_next; S.tk_last[c] = L.tk_last; S.wake_fired[c] = L.wake_fired;
            X.ring_store(P, S, c);
        }
    } else if (role == 3u) {
        // ------------------------------------------ E: one symbol per lane and step: the filter pair at its SECOND instant, timing loop ----
        sym_setprio<sym_prio<NT>(4)>(P);
        const uint32_t wcol_lds = lds_addr(wcol);
        std::conditional_t<NT == 42, SymTaps<42>, SymHalfTaps<NT, NT / 4, NT / 2>> TP;  // (44.1 / 48 kHz: the second half of the tap pairs)
        TP.load(taps);
        Lane L;
        { const State S = fresh_state(); lane_load(L, S, c); }
        const float inv_spt = 1.0f / P.samples_per_ted;
        int cstar = next_fire_count(L.until_next_ted, L.ted_clock);
        int rel = cstar - (int)L.ted_clock - 1;        // index of the next instant, relative to the end of the finished samples
        float cstar_f = (float)cstar;
        // The plan of step s, made at the end of step s - 1: per lane the next symbol -- instant A
        // (completes nothing: where B falls does not depend on A's sample) and instant B, or B alone right after a
        // symsync.reset -- if both lie in finished samples; in the steps after the last block: one instant whatever it is.
        // Wavefront A filters at A (where there is one), this one at B.
        SYM_T_DECL();
        bool pl_typeA = false, pl_ready = false, pl_single = false;
        int pl_p2 = 0;
        uint32_t pl_n2 = 0, pl_n1 = 0xffffffffu;
        float pl_rem1 = 0.0f, pl_instA = 0.0f, pl_c2 = 0.0f;
        auto plan = [&](uint32_t s) __attribute__((always_inline)) {
            pl_single = s > n_blocks;
            const uint32_t wb = (min(s, n_blocks) % (uint32_t)LY::NBLK) * (uint32_t)kB;      // ring slot of the sample at rel 0
            pl_typeA = !pl_single && (L.flags & F_TED_PHASE) != 0u;
            pl_rem1 = L.until_next_ted - cstar_f;                                      // receiver.rs:352
            pl_instA = L.period_inst + __builtin_amdgcn_fmed3f(pl_rem1, -0.5f, 0.5f);   // rx/symsync.rs:236-241
            pl_c2 = sym_next_fire(pl_instA);
            pl_p2 = pl_typeA ? rel + (int)pl_c2 : rel;
            pl_ready = pl_p2 < 0 && s >= 1u && s <= last_e_step;
            auto slot = [&](int p) __attribute__((always_inline)) { int n = (int)wb + p; n += n < 0 ? RING : 0; return (uint32_t)n; };
            pl_n2 = pl_ready ? slot(pl_p2) : 0u;
            pl_n1 = (pl_ready && pl_typeA) ? slot(rel) : 0xffffffffu;
            posbox[(s & 1u) * kWave + lane] = pl_n1;
            if constexpr (NT != 42) pos2box[(s & 1u) * kWave + lane] = pl_ready ? pl_n2 : 0xffffffffu;
        };
        float sa1 = 0.0f, sa2 = 0.0f;
        // the filters of the step's symbol: its second instant here, its first on wavefront A
        auto filters = [&](uint32_t seq) __attribute__((always_inline)) {
            sa1 = 0.0f; sa2 = 0.0f;
            if (__builtin_amdgcn_ballot_w64(pl_ready) != 0ull) {
                SYM_T_BEGIN();
                SYM_TCOUNT(20, 1);
                // A has posted this pass (both wavefronts decide from the same position words whether there is one).  Bounded:
                // should the two ever disagree, the launch reports an error instead of hanging the GPU.
                auto wait_for_a = [&]() __attribute__((always_inline)) {
                    uint32_t spins = 0;
                    while ((int32_t)(seqbox[0] - seq) < 0) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > spin_limit) { if (lane == 0u) atomicOr(err_flags, 4u); spin_limit = 8u; break; }
                    }
                    asm volatile("" ::: "memory");
                };
                if constexpr (NT == 42) {
                    sa2 = PROF_SKIP(P, 512) ? 0.0f : TP.template demod<RING>(wcol_lds, (int)pl_n2);
                    SYM_T_LAP(22);
                    if (__builtin_amdgcn_ballot_w64(pl_ready && pl_typeA) != 0ull) {
                        wait_for_a();
                        sa1 = __uint_as_float(sabox[lane]);
                    }
                } else {
                    // the second half of the taps at both instants; A's half arrives through partbox
                    float2v r2 = {0.0f, 0.0f}, i2 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f}, i1 = {0.0f, 0.0f};
                    const bool any_a = __builtin_amdgcn_ballot_w64(pl_n1 != 0xffffffffu) != 0ull;
                    if (!PROF_SKIP(P, 512)) {
                        TP.template partial<RING>(wcol_lds, (int)pl_n2, r2, i2);
                        if (any_a) TP.template partial<RING>(wcol_lds, pl_n1 == 0xffffffffu ? 0 : (int)pl_n1, r1, i1);
                    }
                    SYM_T_LAP(22);
                    wait_for_a();
                    const lds_u32 *pb = partbox + lane;
                    r2 += float2v{__uint_as_float(pb[0]), __uint_as_float(pb[kWave])}; i2 += float2v{__uint_as_float(pb[2 * kWave]), __uint_as_float(pb[3 * kWave])};
                    sa2 = sym_soft_sample(r2, i2);
                    if (any_a) {
                        r1 += float2v{__uint_as_float(pb[4 * kWave]), __uint_as_float(pb[5 * kWave])}; i1 += float2v{__uint_as_float(pb[6 * kWave]), __uint_as_float(pb[7 * kWave])};
                        sa1 = sym_soft_sample(r1, i1);
                    }
                }
                SYM_T_LAP(24);
            }
        };
        // The two timing updates of the symbol, and the symbol itself to Y1: straight-line, committed by selects (a wavefront has
        // lanes with both instants in this step, lanes with one, lanes with none: as branches the update existed twice, with
        // the register shuffling of two exec-mask regions around it -- ~210 vector instructions a step).
        auto work = [&](uint32_t s) __attribute__((always_inline)) {
            SYM_T_BEGIN();
            const bool rdy = pl_ready, tA = pl_typeA;
            // the TED as the second instant finds it: behind the first one (ZeroCrossingTed::input + TimingLoop::input without
            // a symbol, rx/symsync.rs:236-241, 278-287) where the step has one
            const float p1 = tA ? L.h2 : L.h1, p2 = tA ? sa1 : L.h2;
            const float inst_pre = tA ? pl_instA : L.period_inst;
            const float rem = tA ? pl_instA - pl_c2 : pl_rem1;
            const uint32_t flags_new = tA ? L.flags : (L.flags ^ F_TED_PHASE);       // (two toggles, or one)
            const bool have = (flags_new & F_TED_PHASE) != 0u;
            // ZeroCrossingTed::input + TimingLoop::advance_loop (rx/symsync.rs:198-287), relaxed: same_relaxed_common.h ted_timing_relaxed
            const float dsg = rs_signum(p1) - rs_signum(sa2);
            const float te = p2 * dsg;
            const float offset = __builtin_amdgcn_fmed3f(rem, -0.5f, 0.5f);
            const float e = __builtin_amdgcn_fmed3f(__builtin_fmaf(-offset, inv_spt, te), -1.0f, 1.0f);
            const bool bw_locked = (L.flags & F_BW_LOCKED) != 0u;
            const float alpha = bw_locked ? P.alpha_locked : P.alpha_unlocked;
            const float beta = bw_locked ? P.beta_locked : P.beta_unlocked;
            const float avg1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(beta, e, L.period_avg), P.period_min, P.period_max);
            float inst1 = __builtin_fmaf(alpha, e, avg1) + offset;
            inst1 = (inst1 < 0.0f) ? avg1 : inst1;
            const float inst_new = have ? inst1 : inst_pre + offset;
            const float cs = sym_next_fire(inst_new);
            // committed lane by lane with selects: as one region under `if (ready)` every value that leaves it is copied twice
            // (~60 register moves a step around ~45 instructions of arithmetic)
            const bool take = rdy & have;
            L.h0 = rdy ? p1 : L.h0; L.h1 = rdy ? p2 : L.h1; L.h2 = rdy ? sa2 : L.h2;
            L.flags = rdy ? flags_new : L.flags;
            L.period_avg = take ? avg1 : L.period_avg;
            L.period_inst = rdy ? inst_new : L.period_inst; L.until_next_ted = rdy ? inst_new : L.until_next_ted;
            cstar_f = rdy ? cs : cstar_f;
            rel = rdy ? pl_p2 + (int)cs : rel;
            const uint32_t hdr = take ? (1u | ((uint32_t)(pl_p2 + (pl_single ? 3 * kB : 2 * kB)) << 8)) : 0u;
            SYM_T_LAP(23);
            lds_u32 *sb = symbox + (s & 1u) * LY::sym_words + lane;
            sb[0] = hdr;
            sb[kWave] = __float_as_uint(p2); sb[2 * kWave] = __float_as_uint(sa2);             // (read where the header says so)
            if (P.trace_cap) { sb[3 * kWave] = __float_as_uint(te); sb[4 * kWave] = __float_as_uint(inst_new); }
        };
        // the loop bandwidth of a lock at sync (receiver.rs:431-432) and what end() undoes (receiver.rs:479-490: unlocked
        // loop bandwidth, symsync.reset()), late: see Y1 / Y2
        auto late = [&](uint32_t v) __attribute__((always_inline)) {
            if (v & FB_VALID) {
                L.flags = (L.flags & ~F_BW_LOCKED) | ((v & FB_BW) ? F_BW_LOCKED : 0u);
                if (v & FB_END) {                                        // rx/symsync.rs:166-170, 265-271
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
            // The step's chain -- filters -> timing updates -> where the next symbol's instants fall -> (next step) filters -- is the
            // longest of the pipeline: it starts as soon as S has the samples, and only then waits for what is needed later.
            uint32_t w1, w2;
            SYM_TRACE(3, s, 0);
            wait_for(s, R_S, &w1, &w2, kNapShort);
            SYM_TRACE(3, s, 1);
            P3_LAP(p3_wait);
            if (!PROF_SKIP(P, 32)) filters(2u * s + 1u);
            wait_for(s, R_A | R_Y1 | R_Y2, &w1, &w2, kNapShort);                  // A is done with the positions, Y1 with the symbol box; feedback of Y1 and Y2
            SYM_TRACE(3, s, 3);
            const uint32_t f1 = flags_of_last(w1, s), f2 = flags_of_last(w2, s);
            if (f2 & Y2F_LEAVE) stop_at = s;
            // a change of the loop bandwidth or a symsync.reset() posted in step s - 1: applied behind this step's symbol (its
            // positions are already with A)
            uint32_t late1 = 0u, late2 = 0u;
            if ((f1 & Y1F_ANY) && s >= 1u) late1 = fb1box[((s - 1u) & 1u) * LY::fb_words + lane];
            if ((f2 & Y2F_END) && s >= 1u) late2 = fb2box[((s - 1u) & 1u) * LY::fb_words + lane];
            if (!PROF_SKIP(P, 32)) {
                work(s);
                if (__builtin_amdgcn_ballot_w64(((late1 | late2) & FB_VALID) != 0u) != 0ull) { late(late2); late(late1); }       // (Y2's is of an older symbol)
                if (s + 1u <= n_blocks) rel -= kB;                     // block s is finished when step s + 1 begins
                plan(s + 1u);
            }
            if (s + 1u == n_steps) phasebox[lane] = L.flags & F_TED_PHASE;       // (Y2 merges the phase bit)
            SYM_TRACE(3, s, 2);
            peek();                                                        // (the next step's first wait looks at these words)
            publish(s, 0u, 0u);
            P3_LAP(p3_work);
            if (s == stop_at) { left = true; break; }
        }
        SYM_REPORT(3);
        SYM_T_REPORT();
        if (left || gave_up()) return;
        L.ted_clock = (uint32_t)((int)cstar_f - rel - 1);
        {
            const State S = fresh_state();
            S.until_next_ted[c] = L.until_next_ted; S.ted_clock[c] = L.ted_clock;
            S.ted_h0[c] = L.h0; S.ted_h1[c] = L.h1; S.ted_h2[c] = L.h2;
            S.period_avg[c] = L.period_avg; S.period_inst[c] = L.period_inst;
        }
    } else if (role == 4u) {
        // ------------------------------------------ Y1: squelch + equalizer ---------------------------------------------
        sym_setprio<sym_prio<NT>(5)>(P);
        SymSquelch<NFF, NFB> Q;
        uint64_t symbols0;
        // the squelch's sample history stays in global memory: the state array itself, or -- where the columns of a wavefront
        // are permuted -- a copy by grid position (coalesced; the state array is read and written once per launch)
        const bool hist_copy = K.n_chunks > 1u && K.col_perm != nullptr && K.hist_scratch != nullptr;
        {
            const State S = fresh_state();
            lane_load(Q.L, S, c);
            symbols0 = Q.L.sq_symbols;
            Q.nsym = (uint32_t)symbols0;
            Q.hbase = reinterpret_cast<char *>(hist_copy ? K.hist_scratch : S.sq_hist);
            Q.hcol4 = (hist_copy ? vwg * kWave + lane : c) * 4u;
            Q.hrow4 = C * 4u;
            if (hist_copy) {
#pragma unroll 4
                for (int i = 0; i < kSquelchHist; ++i) *Q.hptr((uint32_t)i) = S.sq_hist[(size_t)i * C + c];
            }
#pragma unroll
            for (int i = 0; i < NFF; ++i) {
                Q.ffc[i] = S.eq_ffc[i * C + c]; Q.ffw[i] = S.eq_ffw[i * C + c];
                Q.sffc[i] = S.eq_snap_ffc[i * C + c]; Q.sffw[i] = S.eq_snap_ffw[i * C + c];
            }
#pragma unroll
            for (int i = 0; i < NFB; ++i) {
                Q.fbc[i] = S.eq_fbc[i * C + c]; Q.fbw[i] = S.eq_fbw[i * C + c];
                Q.sfbc[i] = S.eq_snap_fbc[i * C + c]; Q.sfbw[i] = S.eq_snap_fbw[i * C + c];
            }
            const uint32_t pslot = (2u * Q.nsym) & 63u;
            Q.nxt0 = *Q.hptr((pslot + 16u) & 63u); Q.nxt1 = *Q.hptr((pslot + 17u) & 63u);
        }
        lds_barrier();                                                 // prologue
        P3_T0();
        uint32_t stop_at = 0xffffffffu, flags_before = 0u, last_msg = 0u;
        bool left = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            uint32_t w1, w2;
            SYM_TRACE(4, s, 0);
            // E's symbols, Y2's feedback (and Y2 is done with the word box); and S, which reads this role's flags of step s - 1 from the
            // parity byte that publishing step s + 1 overwrites: without it nothing but timing kept Y1 within one step of S (Y1(s + 1)
            // needs E(s) and Y2(s), Y2(s) only S(s - 1)), and a lock or unlock posted in step s - 1 could be lost.  S runs ahead: free.
            wait_for(s, R_E | R_Y2 | R_S, &w1, &w2, kNapShort);
            SYM_TRACE(4, s, 1);
            P3_LAP(p3_wait);
            const uint32_t f2 = flags_of_last(w2, s);
            if (f2 & Y2F_LEAVE) stop_at = s;
            // What the framer decided in step s - 1 (squelch.lock(true) on Reading, end() on NoCarrier / Burst) -- about the symbol
            // BEFORE the one this wavefront saw in that step.  If that newer symbol already ended the acquisition (lost sync:
            // end() here) or began a new one (sync), the framer's word is about a state that no longer exists: a lock applied
            // behind the end() it preceded would close the squelch for good, an end() behind a fresh sync would wipe it.
            if ((f2 & (Y2F_END | Y2F_LOCK)) && s >= 1u) {
                const uint32_t from_y2 = fb2box[((s - 1u) & 1u) * LY::fb_words + lane];
                const bool superseded = (last_msg & (YM_DROP | YM_ADJUSTED)) != 0u;
#ifdef SYM_Y1_SELECT
                Q.L.flags |= ((from_y2 & FB_SQLOCK) && !superseded) ? (uint32_t)F_SQ_LOCK : 0u;
                const bool y2_end = (from_y2 & FB_END) && !superseded;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(y2_end) != 0ull, 0)) Q.end_masked(y2_end);
#else
                if ((from_y2 & FB_SQLOCK) && !superseded) Q.L.flags |= F_SQ_LOCK;
                if ((from_y2 & FB_END) && !superseded) Q.end();
#endif
            }
            last_msg = 0u;
            uint32_t flags = 0u;
            if (s >= 2u && s <= last_y1_step) {
                const lds_u32 *sb = symbox + ((s - 1u) & 1u) * LY::sym_words + lane;
                const uint32_t hdr = sb[0], zero_w = sb[kWave], sym_w = sb[2 * kWave];
                uint32_t msg = 0u, fbv = 0u;
#ifdef SYM_Y1_SELECT
                {
                    const bool valid = (hdr & 1u) && !PROF_SKIP(P, 16);
                    const uint32_t off = hdr >> 8;
                    const float zero = __uint_as_float(zero_w), sym = __uint_as_float(sym_w);
                    if (P.trace_cap && valid) {
                        // the soft-symbol trace (tests): the counter is that of the sample after the symbol's
                        const State S = fresh_state();
                        const uint64_t counter = counter0 + (int64_t)row_l + (uint64_t)sym_index(s - 1u, off) + 1u;
                        const uint32_t n = S.trace_n[c];
                        if (n < P.trace_cap) {
                            float *t = S.trace + ((size_t)c * P.trace_cap + n) * 4;
                            t[0] = zero; t[1] = sym; t[2] = __uint_as_float(sb[3 * kWave]); t[3] = __uint_as_float(sb[4 * kWave]);
                            S.trace_idx[(size_t)c * P.trace_cap + n] = counter;
                        }
                        S.trace_n[c] = n + 1;
                    }
                    msg = Q.symbol_masked(P, valid, zero, sym, off, &fbv);
                }
#else
                if ((hdr & 1u) && !PROF_SKIP(P, 16)) {
                    const uint32_t off = hdr >> 8;
                    const float zero = __uint_as_float(zero_w), sym = __uint_as_float(sym_w);
                    if (P.trace_cap) {
                        // the soft-symbol trace (tests): the counter is that of the sample after the symbol's
                        const State S = fresh_state();
                        const uint64_t counter = counter0 + (int64_t)row_l + (uint64_t)sym_index(s - 1u, off) + 1u;
                        const uint32_t n = S.trace_n[c];
                        if (n < P.trace_cap) {
                            float *t = S.trace + ((size_t)c * P.trace_cap + n) * 4;
                            t[0] = zero; t[1] = sym; t[2] = __uint_as_float(sb[3 * kWave]); t[3] = __uint_as_float(sb[4 * kWave]);
                            S.trace_idx[(size_t)c * P.trace_cap + n] = counter;
                        }
                        S.trace_n[c] = n + 1;
                    }
                    msg = Q.symbol(P, zero, sym, off, &fbv);
                }
#endif
                ybox[(s & 1u) * kWave + lane] = msg;
                last_msg = msg;
                if (__builtin_amdgcn_ballot_w64(fbv != 0u) != 0ull) {
                    uint32_t u1, u2;
                    wait_for(s, R_S, &u1, &u2, kNapLong);                        // S (like E) has read the feedback box of two steps ago
                    fb1box[(s & 1u) * LY::fb_words + lane] = fbv; flags = Y1F_ANY;
                }
            }
            if (s + 1u == n_steps) flagbox[lane] = Q.L.flags & (F_AGC_LOCKED | F_BW_LOCKED | F_SQ_LOCK | F_EQ_MODE_MASK | F_EQ_BITS_MASK);   // (Y2 merges the flag bits)
            SYM_TRACE(4, s, 2);
            peek();                                                        // (the next step's first wait looks at these words)
            publish(s, flags, flags_before);
            flags_before = flags;
            P3_LAP(p3_work);
            if (s == stop_at) { left = true; break; }
        }
        SYM_REPORT(4);
        if (left || gave_up()) return;
        const State S = fresh_state();
        S.sq_data[c] = Q.L.sq_data; S.sq_power[c] = Q.L.sq_power; S.sq_phist[c] = Q.L.sq_phist;
        S.sq_fill[c] = Q.L.sq_fill; S.sq_clock[c] = Q.L.sq_clock; S.sq_symbols[c] = symbols0 + (uint64_t)(Q.nsym - (uint32_t)symbols0);
        S.eq_word[c] = Q.L.eq_word; S.eq_count[c] = Q.L.eq_count;
        if (hist_copy) {
#pragma unroll 4
            for (int i = 0; i < kSquelchHist; ++i) S.sq_hist[(size_t)i * C + c] = *Q.hptr((uint32_t)i);
        }
#pragma unroll
        for (int i = 0; i < NFF; ++i) {
            S.eq_ffc[i * C + c] = Q.ffc[i]; S.eq_ffw[i * C + c] = Q.ffw[i];
            S.eq_snap_ffc[i * C + c] = Q.sffc[i]; S.eq_snap_ffw[i * C + c] = Q.sffw[i];
        }
#pragma unroll
        for (int i = 0; i < NFB; ++i) {
            S.eq_fbc[i * C + c] = Q.fbc[i]; S.eq_fbw[i * C + c] = Q.fbw[i];
            S.eq_snap_fbc[i * C + c] = Q.sfbc[i]; S.eq_snap_fbw[i * C + c] = Q.sfbw[i];
        }
    } else {
        // ------------------------------------------ Y2: framer, link state, bursts, hand-over ----------------------------
        sym_setprio<sym_prio<NT>(6)>(P);
        Lane L;
        SymFramer F;
        uint8_t *fr_rows;            // in the loop: the framer's rows only
        { const State S0 = fresh_state(); lane_load(L, S0, c); fr_rows = S0.fr_msg; }
        F.st = fr_state(L); F.last = (L.flags & F_LINK_MASK) >> F_LINK_SHIFT;
        F.word = L.fr_word; F.count = L.fr_count; F.invalid = L.fr_invalid; F.len = L.fr_len;
        uint8_t *const row = fr_rows + (size_t)c * kBurstCap;
        Output O{};
        { const Output Of = fresh_output(); O.n_events = Of.n_events; O.bursts = Of.bursts; O.burst_cap = Of.burst_cap; }
        lds_barrier();                                                 // prologue
        P3_T0();
        uint32_t stop_at = 0xffffffffu, flags_before = 0u;
        bool left = false, lane_done = false, leave_posted = false;
        for (uint32_t s = 0; s < n_steps; ++s) {
            uint32_t w1, w2;
            SYM_TRACE(5, s, 0);
            // Y1's words; and everybody has read the flags and feedback of two steps ago, A is done with the link box (ONE wait: this
            // role's step is the longest of the six, the others have long published when it begins)
            wait_for(s, R_Y1 | R_S | R_T | R_A | R_E, &w1, &w2, kNapShort);
            SYM_TRACE(5, s, 1);
            P3_LAP(p3_wait);
            uint32_t flags = 0u;
            if (s >= 3u && s <= last_y2_step) {
                const uint32_t m = PROF_SKIP(P, 1024) ? 0u : ybox[((s - 1u) & 1u) * kWave + lane];
                uint32_t fbv = 0, burst_len = 0, io1 = 0xffffffffu;
                bool emit = false;
                const uint32_t link = sym_framer_step(P, F, row, m, &burst_len, &fbv, &emit);
                const bool want_slot = emit & (link == 3u);            // this lane has just finished a burst
                const uint32_t io0 = (m & YM_VALID) ? (1u | (link << 1) | (emit ? 8u : 0u) | (((m >> YM_OFF_SHIFT) & kSymOffMask) << 4)) : 0u;
                // finished bursts go into the pool with the whole wavefront: one slot reservation for all of them and one
                // coalesced round trip per burst (same_kernels_pipe.hip)
                uint64_t pend = __builtin_amdgcn_ballot_w64(want_slot);
                lds_u32 *io = iobox + (s & 1u) * LY::io_words + lane;
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
                            const uint32_t *src = reinterpret_cast<const uint32_t *>(fr_rows + (size_t)cj * kBurstCap);
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
                    io[kWave] = io1; io[2 * kWave] = burst_len;
                }
                io[0] = io0;
                const bool any_end = __builtin_amdgcn_ballot_w64((fbv & FB_VALID) != 0u) != 0ull;
                const bool any_lock = __builtin_amdgcn_ballot_w64((fbv & FB_SQLOCK) != 0u) != 0ull;
                if (any_end || any_lock) fb2box[(s & 1u) * LY::fb_words + lane] = fbv;
                flags = (any_end ? Y2F_END : 0u) | (any_lock ? Y2F_LOCK : 0u);
                // Time-parallel chunk that hands over: from its nominal end on, a lane's hand-over instant is
                // the end of the first block after which its link state is NoCarrier; once every lane has one the group
                // leaves (one more step: A still has to log this step's events)
                if (may_leave && !leave_posted) {
                    const uint32_t blk = min(s - 3u, n_blocks - 1u);   // every symbol up to the end of this block has been seen after this step
                    if (!lane_done && blk + 1u >= n_nominal && F.last == 0u && (xl == nullptr || blk < avail_l)) {
                        lane_done = true;
                        K.handover[c] = counter0 + (int64_t)row_l + (uint64_t)(blk + 1u) * kB;
                        donebox[lane] = 1u;
                    }
                    if (__builtin_amdgcn_ballot_w64(!lane_done) == 0ull) { flags |= Y2F_LEAVE; leave_posted = true; stop_at = s + 1u; }
                }
            }
            SYM_TRACE(5, s, 2);
            peek();                                                        // (the next step's first wait looks at these words)
            publish(s, flags, flags_before);
            flags_before = flags;
            P3_LAP(p3_work);
            if (s == stop_at) { left = true; break; }
        }
        SYM_REPORT(5);
        SYM_COUNT(18, 1);                                              // launches of the reporting group ...
        SYM_COUNT(19, left ? stop_at + 1u : n_steps);                  // ... and the steps they ran
        if (left || gave_up()) return;
        // E's TED phase, A's wake-up flag, Y1's locks and equalizer bits: written before their last step was published
        uint32_t w1, w2;
        wait_for(n_steps, R_A | R_E | R_Y1, &w1, &w2, kNapShort);
        if (gave_up()) return;
        const uint32_t flags_out = (F.st << F_FR_STATE_SHIFT) | (F.last << F_LINK_SHIFT) | (phasebox[lane] & F_TED_PHASE) | (againbox[lane] & F_TICK_AGAIN) | flagbox[lane];
        const State S1 = fresh_state();
        S1.fr_word[c] = F.word; S1.fr_count[c] = F.count; S1.fr_invalid[c] = F.invalid;
        S1.fr_len[c] = F.len; S1.flags[c] = flags_out;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------------------------------
// 22.05 / 44.1 / 48 kHz with the reference's default DC-blocker length, the default or the disabled equalizer, a non-negative AGC
// floor, whole groups of 64 state columns, and a timing loop whose shortest symbol is longer than a step (two instants at
// least max_block_len + 1 samples apart each: 19 / 40 / 44) and whose filters stay inside the four finished blocks of the ring
static uint32_t sym_rate_nt(const Params &P)
{
    if (P.ntaps == 42u && P.dc_len == 16u) return 42u;
    if (P.ntaps == 92u && P.dc_len == 35u) return 92u;
    if (P.ntaps == 84u && P.dc_len == 32u) return 84u;
    return 0u;
}
// Two translation units (round 6).  The 44.1 / 48 kHz instantiations are compiled a second time from this file by
// same_kernels_sym_hi.hip (SYM_TU_HI), which the build schedules for instruction-level parallelism (`-mllvm
// -amdgpu-sched-strategy=max-ilp`, sameold_amd/build.py): there six wavefronts share four SIMDs and a role's own dependent chains
// are what it waits for -- 2.43-2.45 -> 2.36 ms at 48 kHz, 2.13-2.16 -> 2.07 at 44.1 kHz, the same events (tools/sym_hash.py).  At
// 22.05 kHz (twelve wavefronts per CU) the same option is equal with the link layer alone and 10 % SLOWER with the transport layer
// on (1.91 -> 2.10 ms at 32 768 channels), `max-memory-clause` likewise: this unit keeps the compiler's default.  Profile and
// timeline builds keep one unit (their counters are per-unit device globals).
#if !defined(SAME_PROFILE) && !defined(SAME_SYM_TL)
#define SYM_SPLIT_TU 1
#endif
#if !defined(SYM_TU_HI)
uint32_t sym_block_len(const Params &P) { return sym_rate_nt(P) == 42u ? (uint32_t)SymLayout<42>::B : (uint32_t)SymLayout<92>::B; }
bool sym_kernel_supported(const Params &P)
{
    static_assert(SymLayout<92>::B == SymLayout<84>::B && SymLayout<92>::NBLK == SymLayout<42>::NBLK, "one step length beyond 22.05 kHz");
    if (P.knob_sym < 0) return false;
    const uint32_t nt = sym_rate_nt(P);
    if (nt == 0u || P.win_ring < 64u || P.win_ring < nt || (P.n_channels % kWave) != 0u) return false;
    if (P.n_channels >= (1u << 22)) return false;                    // (the squelch history's 24-bit row pitch, SymSquelch::hptr)
    if (!((P.eq_nff == 6u && P.eq_nfb == 4u) || (P.eq_nff == 1u && P.eq_nfb == 1u))) return false;
    if (!(P.agc_min >= 0.0f)) return false;
    const uint32_t B = sym_block_len(P), apart = max_block_len(P) + 1u;
    if (!(2u * apart > B)) return false;                             // at most one symbol per lane and step
    // How far back a filter reaches from the end of the finished samples: a lane completes its symbol up to B behind it, one more
    // B - apart after a symsync.reset() (the next instant completes a symbol by itself), the symbol's first instant lies up to
    // period_max + alpha + 0.5 (+ rounding) before its second, and the filter takes ntaps - 1 samples before that.
    const float a = P.alpha_unlocked > P.alpha_locked ? P.alpha_unlocked : P.alpha_locked;
    const uint32_t reach = (2u * B - apart) + 1u + (uint32_t)std::ceil(P.period_max + a + 1.5f) + (nt - 1u);
    return reach <= (uint32_t)(SymLayout<42>::NBLK - 2) * B;
}
#endif      // !SYM_TU_HI

template <int NT, int NFF, int NFB, typename SampleT>
static hipError_t launch_sym_one(const Params &P, const State &S, const Output &O, const float4 *taps, const SampleT *x,
                                 uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{
    constexpr uint32_t kSymHalves = (uint32_t)SymLayout<NT>::HALVES;
    constexpr size_t lds = (size_t)kSymHalves * SymLayout<NT>::lds_bytes;
    static_assert(lds <= 160u * 1024u, "one workgroup per CU");
    const bool cm = K.n_chunks > 1u && K.col_row0 != nullptr;
    if (cm && (!std::is_same<SampleT, float>::value || NT != 42)) return hipErrorInvalidValue;
    if (K.n_chunks > 1u && (K.in_channels % kWave) != 0u) return hipErrorInvalidValue;   // a workgroup would straddle chunks
    if (n_blocks == 0u) return hipSuccess;
    auto go = [&](auto kernel) -> hipError_t {
        if (lds > 64u * 1024u) {
            // more than the default 64 KB of dynamic LDS per workgroup: opt in, once per kernel and device.  (Keyed on the
            // kernel's address: the time-major and the channel-major build share this lambda's instantiation.)
            static std::mutex mu;
            static std::set<std::pair<const void *, int>> opted_in;
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
            const std::pair<const void *, int> key(reinterpret_cast<const void *>(kernel), dev);
            std::lock_guard<std::mutex> lock(mu);
            if (opted_in.find(key) == opted_in.end()) {
                const hipError_t e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                opted_in.insert(key);
            }
        }
        hipLaunchKernelGGL(kernel, dim3((P.n_channels / kWave + kSymHalves - 1u) / kSymHalves), dim3(kSymHalves * kSymRoles * kWave), lds, stream, P, S, O, taps, x, n_blocks, counter0, K);
        return hipGetLastError();
    };
    if constexpr (std::is_same<SampleT, float>::value && NT == 42) {
        if (cm) return go(demod_sym_kernel<NT, NFF, NFB, float, 1>);
    }
    return go(demod_sym_kernel<NT, NFF, NFB, SampleT, 0>);
}

// the 44.1 / 48 kHz instantiations (in same_kernels_sym_hi.hip's unit where the build is split)
template <typename SampleT>
static hipError_t launch_sym_hi_t(const Params &P, const State &S, const Output &O, const float4 *taps, const SampleT *x,
                                  uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{
    const bool eq = P.eq_nff == 6u && P.eq_nfb == 4u;
    if (sym_rate_nt(P) == 92u)
        return eq ? launch_sym_one<92, 6, 4, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K) : launch_sym_one<92, 1, 1, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K);
    return eq ? launch_sym_one<84, 6, 4, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K) : launch_sym_one<84, 1, 1, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K);
}
#if defined(SYM_TU_HI)
#if defined(SYM_SPLIT_TU)
hipError_t launch_demod_sym_hi(const Params &P, const State &S, const Output &O, const float4 *taps, const float *x,
                               uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_sym_hi_t<float>(P, S, O, taps, x, n_blocks, counter0, stream, K); }
hipError_t launch_demod_sym_hi_i16(const Params &P, const State &S, const Output &O, const float4 *taps, const int16_t *x,
                                   uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_sym_hi_t<int16_t>(P, S, O, taps, x, n_blocks, counter0, stream, K); }
#endif
#else
#if defined(SYM_SPLIT_TU)
hipError_t launch_demod_sym_hi(const Params &P, const State &S, const Output &O, const float4 *taps, const float *x,
                               uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K);
hipError_t launch_demod_sym_hi_i16(const Params &P, const State &S, const Output &O, const float4 *taps, const int16_t *x,
                                   uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K);
static hipError_t launch_sym_hi(const Params &P, const State &S, const Output &O, const float4 *taps, const float *x,
                                uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_demod_sym_hi(P, S, O, taps, x, n_blocks, counter0, stream, K); }
static hipError_t launch_sym_hi(const Params &P, const State &S, const Output &O, const float4 *taps, const int16_t *x,
                                uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_demod_sym_hi_i16(P, S, O, taps, x, n_blocks, counter0, stream, K); }
#else
template <typename SampleT>
static hipError_t launch_sym_hi(const Params &P, const State &S, const Output &O, const float4 *taps, const SampleT *x,
                                uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_sym_hi_t<SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K); }
#endif
template <typename SampleT>
static hipError_t launch_sym_t(const Params &P, const State &S, const Output &O, const float4 *taps, const SampleT *x,
                               uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{
    if (!sym_kernel_supported(P)) return hipErrorInvalidValue;
    const bool eq = P.eq_nff == 6u && P.eq_nfb == 4u;
    if (sym_rate_nt(P) == 42u)
        return eq ? launch_sym_one<42, 6, 4, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K) : launch_sym_one<42, 1, 1, SampleT>(P, S, O, taps, x, n_blocks, counter0, stream, K);
    return launch_sym_hi(P, S, O, taps, x, n_blocks, counter0, stream, K);
}
hipError_t launch_demod_sym(const Params &P, const State &S, const Output &O, const float4 *taps, const float *x,
                            uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_sym_t<float>(P, S, O, taps, x, n_blocks, counter0, stream, K); }
hipError_t launch_demod_sym_i16(const Params &P, const State &S, const Output &O, const float4 *taps, const int16_t *x,
                                uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const PipeChunks &K)
{ return launch_sym_t<int16_t>(P, S, O, taps, x, n_blocks, counter0, stream, K); }
#endif      // !SYM_TU_HI

}  // namespace same

#if !defined(SYM_TU_HI)
SYM_PROFILE_EXPORTS()
#endif
