// same_relaxed_common.h -- the relaxed-arithmetic building blocks shared by same_kernels_relaxed.hip (one wavefront per
// 64 state columns) and the FASTMATH build of the wavefront pipeline (same_kernels_pipe.hip): matched-filter chunks
// with asm-issued LDS loads, the relaxed AGC step, timing loop and equalizer step.  What "relaxed" means and what it
// guarantees: include/same_rx.h, SAME_BATCH_RELAXED.
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "same_dev_common.h"
#include "same_device.h"
#include "same_fast_common.h"

namespace same {

constexpr int kRelaxChunk = 14;                 // taps per filter chunk

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

// One chunk of 14 taps of both matched filters (rx/filter.rs:363-377 re-associated): `wa` = LDS byte address of the
// chunk's LOWEST window slot (the sample tap base + 13 multiplies; slots are 256 bytes apart), `ta` = LDS byte address
// of tap `base` (16 bytes per tap: mark re/im, space re/im).  Even / odd taps accumulate separately.
//
// The loads -- 7 x ds_read2st64_b32 (two window slots each) and 14 x ds_read_b128 (one tap; a wave-uniform address,
// i.e. a broadcast) -- are issued back to back from asm statements.  Left to itself the compiler loads two taps,
// waits, multiplies, loads the next two ...: fourteen exposed LDS round trips per chunk.  WIDE (builds with registers
// to spare): all 21 loads at once, the first half's products under the second half's latency (70 registers of
// loads); otherwise two rounds of 7 taps (36 registers).
#define RELAX_TAP(acc_m, acc_s, fma, win_, tap_) do { const float2v hm_ = {tap_.x, tap_.y}, hs_ = {tap_.z, tap_.w}; fma(acc_m, win_, hm_); fma(acc_s, win_, hs_); } while (0)
template <bool WIDE>
__device__ __forceinline__ void relax_filter_chunk(uint32_t wa, uint32_t ta, float2v &am0, float2v &am1, float2v &as0, float2v &as1)
{
    static_assert(kRelaxChunk == 14, "the load sequence below is written out for 14 taps");
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
}
#undef RELAX_TAP

// |mark| - |space| clamped to +-1 (rx/demod.rs:163-164) with the magnitudes as f32 square roots
__device__ __forceinline__ float relax_magnitude(float2v a) { return __builtin_amdgcn_sqrtf(__builtin_fmaf(a.x, a.x, a.y * a.y)); }

// FskDemod::demod_now rx/demod.rs:156-164 at the instant whose sample sits in ring slot `newest` of a ring whose first
// 13 slots are stored twice (same_kernels_relaxed.hip): tap j multiplies slot (newest - j) mod RING; a chunk is read
// upwards from its lowest slot, moved into the mirror when it would wrap.
template <int NT, int RING, bool WIDE>
__device__ __forceinline__ float demod_relaxed(uint32_t taps_lds, uint32_t wcol_lds, int newest)
{
    constexpr int CH = kRelaxChunk;
    float2v am0 = {0.0f, 0.0f}, am1 = {0.0f, 0.0f}, as0 = {0.0f, 0.0f}, as1 = {0.0f, 0.0f};
    int top = newest;
#pragma unroll 1
    for (int base = 0; base < NT; base += CH) {
        int s = top;
        s += s < 0 ? RING : 0;
        s += s < CH - 1 ? RING : 0;                          // slots RING .. RING + CH - 2 repeat slots 0 .. CH - 2
        relax_filter_chunk<WIDE>(wcol_lds + (uint32_t)(s - (CH - 1)) * (kWave * 4u), taps_lds + (uint32_t)base * 16u, am0, am1, as0, as1);
        top -= CH;
    }
    return __builtin_amdgcn_fmed3f(relax_magnitude(am0 + am1) - relax_magnitude(as0 + as1), -1.0f, 1.0f);
}

// The same over the pipeline's fully mirrored window (same_fast_common.h): tap j at slot newest + RING - j, no wrap.
// `wlane_lds` = LDS byte address of logical slot 0 of this lane's column; the two magnitudes separately.
template <int NT, int RING, bool WIDE>
__device__ __forceinline__ void demod_pair_relaxed(uint32_t taps_lds, uint32_t wlane_lds, uint32_t newest, float *hm_out, float *hs_out)
{
    constexpr int CH = kRelaxChunk;
    static_assert(NT % CH == 0, "whole chunks");
    float2v am0 = {0.0f, 0.0f}, am1 = {0.0f, 0.0f}, as0 = {0.0f, 0.0f}, as1 = {0.0f, 0.0f};
    uint32_t wa = wlane_lds + (newest + (uint32_t)RING - (uint32_t)(CH - 1)) * (kWave * 4u);
    uint32_t ta = taps_lds;
#pragma unroll 1
    for (int base = 0; base < NT; base += CH) {
        relax_filter_chunk<WIDE>(wa, ta, am0, am1, as0, as1);
        wa -= (uint32_t)CH * (kWave * 4u);
        ta += (uint32_t)CH * 16u;
    }
    *hm_out = relax_magnitude(am0 + am1);
    *hs_out = relax_magnitude(as0 + as1);
}

// The same for 42 taps with the loads of the three chunks software-pipelined across each other: a chunk is read in two
// parts (A: its 8 lowest window slots = taps base+13 .. base+6, 12 loads; B: the 6 above = taps base+5 .. base, 9 loads),
// three register sets take the parts in turn, and while one part's products issue the next two parts' loads are in
// flight.  LDS returns in order, so "at most N outstanding" names the part that has landed.  For the pipeline's helper
// wavefront, whose filters are the first link of the chain a step waits for.
struct RelaxPart { float2v w0, w1, w2, w3; float4v t0, t1, t2, t3, t4, t5, t6, t7; };
#define RELAX_LOAD_A(P_, wa_, ta_)                                                                         \
    asm volatile("ds_read2st64_b32 %[w0], %[wa] offset0:0 offset1:1\n\t"                            \
                 "ds_read2st64_b32 %[w1], %[wa] offset0:2 offset1:3\n\t"                            \
                 "ds_read2st64_b32 %[w2], %[wa] offset0:4 offset1:5\n\t"                            \
                 "ds_read2st64_b32 %[w3], %[wa] offset0:6 offset1:7\n\t"                            \
                 "ds_read_b128 %[t7], %[ta] offset:208\n\t"                                                   \
                 "ds_read_b128 %[t6], %[ta] offset:192\n\t"                                                   \
                 "ds_read_b128 %[t5], %[ta] offset:176\n\t"                                                   \
                 "ds_read_b128 %[t4], %[ta] offset:160\n\t"                                                   \
                 "ds_read_b128 %[t3], %[ta] offset:144\n\t"                                                   \
                 "ds_read_b128 %[t2], %[ta] offset:128\n\t"                                                   \
                 "ds_read_b128 %[t1], %[ta] offset:112\n\t"                                                   \
                 "ds_read_b128 %[t0], %[ta] offset:96"                                                        \
                 : [w0] "=&v"(P_.w0), [w1] "=&v"(P_.w1), [w2] "=&v"(P_.w2), [w3] "=&v"(P_.w3), [t0] "=&v"(P_.t0), [t1] "=&v"(P_.t1), \
                   [t2] "=&v"(P_.t2), [t3] "=&v"(P_.t3), [t4] "=&v"(P_.t4), [t5] "=&v"(P_.t5), [t6] "=&v"(P_.t6), [t7] "=&v"(P_.t7)   \
                 : [wa] "v"(wa_), [ta] "v"(ta_) : "memory")
#define RELAX_LOAD_B(P_, wa_, ta_)                                                                         \
    asm volatile("ds_read2st64_b32 %[w0], %[wa] offset0:8 offset1:9\n\t"                            \
                 "ds_read2st64_b32 %[w1], %[wa] offset0:10 offset1:11\n\t"                          \
                 "ds_read2st64_b32 %[w2], %[wa] offset0:12 offset1:13\n\t"                          \
                 "ds_read_b128 %[t5], %[ta] offset:80\n\t"                                                    \
                 "ds_read_b128 %[t4], %[ta] offset:64\n\t"                                                    \
                 "ds_read_b128 %[t3], %[ta] offset:48\n\t"                                                    \
                 "ds_read_b128 %[t2], %[ta] offset:32\n\t"                                                    \
                 "ds_read_b128 %[t1], %[ta] offset:16\n\t"                                                    \
                 "ds_read_b128 %[t0], %[ta] offset:0"                                                         \
                 : [w0] "=&v"(P_.w0), [w1] "=&v"(P_.w1), [w2] "=&v"(P_.w2), [t0] "=&v"(P_.t0), [t1] "=&v"(P_.t1),        \
                   [t2] "=&v"(P_.t2), [t3] "=&v"(P_.t3), [t4] "=&v"(P_.t4), [t5] "=&v"(P_.t5)                              \
                 : [wa] "v"(wa_), [ta] "v"(ta_) : "memory")
// (the values travel through the wait statement so that nothing that reads them can be moved above it)
#define RELAX_WAIT_A(P_, n_) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(P_.w0), "+v"(P_.w1), "+v"(P_.w2), "+v"(P_.w3), "+v"(P_.t0), "+v"(P_.t1), \
                                          "+v"(P_.t2), "+v"(P_.t3), "+v"(P_.t4), "+v"(P_.t5), "+v"(P_.t6), "+v"(P_.t7))
#define RELAX_WAIT_B(P_, n_) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(P_.w0), "+v"(P_.w1), "+v"(P_.w2), "+v"(P_.t0), "+v"(P_.t1), \
                                          "+v"(P_.t2), "+v"(P_.t3), "+v"(P_.t4), "+v"(P_.t5))
#define RELAX_TAP2(fma, win_, tap_) do { const float2v hm_ = {tap_.x, tap_.y}, hs_ = {tap_.z, tap_.w}; fma(am, win_, hm_); fma(as, win_, hs_); } while (0)
// part A: window pairs w0..w3 = {base+13, base+12} .. {base+7, base+6} against taps t7 (base+13) .. t0 (base+6)
#define RELAX_FMA_A(P_) do { { float2v &am = am0, &as = as0; RELAX_TAP2(pk_fma_lo, P_.w0, P_.t7); } { float2v &am = am1, &as = as1; RELAX_TAP2(pk_fma_hi, P_.w0, P_.t6); } \
                             { float2v &am = am0, &as = as0; RELAX_TAP2(pk_fma_lo, P_.w1, P_.t5); } { float2v &am = am1, &as = as1; RELAX_TAP2(pk_fma_hi, P_.w1, P_.t4); } \
                             { float2v &am = am0, &as = as0; RELAX_TAP2(pk_fma_lo, P_.w2, P_.t3); } { float2v &am = am1, &as = as1; RELAX_TAP2(pk_fma_hi, P_.w2, P_.t2); } \
                             { float2v &am = am0, &as = as0; RELAX_TAP2(pk_fma_lo, P_.w3, P_.t1); } { float2v &am = am1, &as = as1; RELAX_TAP2(pk_fma_hi, P_.w3, P_.t0); } } while (0)
// part B: window pairs w0..w2 = {base+5, base+4} .. {base+1, base} against taps t5 (base+5) .. t0 (base)
#define RELAX_FMA_B(P_) do { { float2v &am = am0, &as = as0; RELAX_TAP2(pk_fma_lo, P_.w0, P_.t5); } { float2v &am = am1, &as = as1; RELAX_TAP2(pk_fma_hi, P_.w0, P_.t4); } \
                             { float2v &am = am0, &as = as0; RELAX_TAP2(pk_fma_lo, P_.w1, P_.t3); } { float2v &am = am1, &as = as1; RELAX_TAP2(pk_fma_hi, P_.w1, P_.t2); } \
                             { float2v &am = am0, &as = as0; RELAX_TAP2(pk_fma_lo, P_.w2, P_.t1); } { float2v &am = am1, &as = as1; RELAX_TAP2(pk_fma_hi, P_.w2, P_.t0); } } while (0)
template <int RING>
__device__ __forceinline__ void demod_pair_relaxed_42(uint32_t taps_lds, uint32_t wlane_lds, uint32_t newest, float *hm_out, float *hs_out)
{
    static_assert(kRelaxChunk == 14, "three chunks of 14 taps, written out");
    float2v am0 = {0.0f, 0.0f}, am1 = {0.0f, 0.0f}, as0 = {0.0f, 0.0f}, as1 = {0.0f, 0.0f};
    // chunk c (taps 14 c .. 14 c + 13): its lowest window slot is newest + RING - 13 - 14 c, its first tap at 224 c bytes
    const uint32_t wa0 = wlane_lds + (newest + (uint32_t)RING - 13u) * (kWave * 4u);
    const uint32_t wa1 = wa0 - 14u * (kWave * 4u), wa2 = wa1 - 14u * (kWave * 4u);
    const uint32_t ta0 = taps_lds, ta1 = taps_lds + 224u, ta2 = taps_lds + 448u;
    RelaxPart X, Y, Z;
    RELAX_LOAD_A(X, wa0, ta0);             // 12 loads
    RELAX_LOAD_B(Y, wa0, ta0);             //  9
    RELAX_WAIT_A(X, 9);  RELAX_FMA_A(X);
    RELAX_LOAD_A(Z, wa1, ta1);             // in flight: B0 9, A1 12
    RELAX_WAIT_B(Y, 12); RELAX_FMA_B(Y);
    RELAX_LOAD_B(X, wa1, ta1);             // A1 12, B1 9
    RELAX_WAIT_A(Z, 9);  RELAX_FMA_A(Z);
    RELAX_LOAD_A(Y, wa2, ta2);             // B1 9, A2 12
    RELAX_WAIT_B(X, 12); RELAX_FMA_B(X);
    RELAX_LOAD_B(Z, wa2, ta2);             // A2 12, B2 9
    RELAX_WAIT_A(Y, 9);  RELAX_FMA_A(Y);
    RELAX_WAIT_B(Z, 0);  RELAX_FMA_B(Z);
    *hm_out = relax_magnitude(am0 + am1);
    *hs_out = relax_magnitude(as0 + as1);
}

// ... and for any number of whole chunks (84 taps: 6; 92 taps padded with zero taps: 7), the same software pipeline written
// once: part k (chunk k / 2, A then B) lives in register set k % 3, is waited for with only part k + 1 still in flight, and
// part k + 2 is requested as soon as its products have been issued.  (The chunk-at-a-time loop of demod_pair_relaxed
// exposes two LDS round trips per chunk: at 48 kHz the helper wavefront, which runs both filters of the FASTMATH build, was
// 14 round trips per instant behind.)
template <typename F, int... I>
__device__ __forceinline__ void relax_static_for_(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int NCH, int RING>
__device__ __forceinline__ void demod_pair_relaxed_chunks(uint32_t taps_lds, uint32_t wlane_lds, uint32_t newest, float *hm_out, float *hs_out)
{
    static_assert(kRelaxChunk == 14, "chunks of 14 taps, written out");
    float2v am0 = {0.0f, 0.0f}, am1 = {0.0f, 0.0f}, as0 = {0.0f, 0.0f}, as1 = {0.0f, 0.0f};
    const uint32_t wa0 = wlane_lds + (newest + (uint32_t)RING - 13u) * (kWave * 4u);
    RelaxPart X, Y, Z;
    constexpr int NP = 2 * NCH;
    auto part = [&](auto kc, auto &&fn) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k % 3 == 0) fn(X); else if constexpr (k % 3 == 1) fn(Y); else fn(Z);
    };
    auto load = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, c = k / 2;
        const uint32_t wa = wa0 - (uint32_t)(14 * c) * (kWave * 4u), ta = taps_lds + (uint32_t)(224 * c);
        part(kc, [&](RelaxPart &R) __attribute__((always_inline)) { if constexpr (k % 2 == 0) RELAX_LOAD_A(R, wa, ta); else RELAX_LOAD_B(R, wa, ta); });
    };
    load(std::integral_constant<int, 0>{});
    load(std::integral_constant<int, 1>{});
    relax_static_for_([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        part(kc, [&](RelaxPart &R) __attribute__((always_inline)) {
            if constexpr (k % 2 == 0) {
                if constexpr (k + 1 < NP) RELAX_WAIT_A(R, 9); else RELAX_WAIT_A(R, 0);
                RELAX_FMA_A(R);
            } else {
                if constexpr (k + 1 < NP) RELAX_WAIT_B(R, 12); else RELAX_WAIT_B(R, 0);
                RELAX_FMA_B(R);
            }
        });
        if constexpr (k + 2 < NP) load(std::integral_constant<int, k + 2>{});
    }, std::make_integer_sequence<int, NP>{});
    *hm_out = relax_magnitude(am0 + am1);
    *hs_out = relax_magnitude(as0 + as1);
}

// Both matched filters at one instant on CENTRED taps, taps in LDS: the 44.1 / 48 kHz form of same_kernels_sym.hip's SymTaps::demod
// (which keeps its 42 taps in registers).  The filter is a cisoid (rx/waveform.rs:39-64) and only its output's magnitude is used
// (rx/demod.rs:156-164), so it may be turned by a unit phasor: u[k] = h[k] e^{j a (N-1)/2} has u[N-1-k] = conj(u[k]), and for a real
// window  sum_i w_i u_i = sum_{k<N/2} (w_k + w_{N-1-k}) Re u_k + j (w_k - w_{N-1-k}) Im u_k.  Per tap PAIR: one window load (both
// samples), one tap load (Re mark, Re space, Im mark, Im space: same_config.cpp appends that table behind the taps), one packed add
// and two packed multiply-adds -- 5 instructions where demod_pair_relaxed_chunks spends 7 on the two taps (and 98 padded taps for
// 92).  `ctaps_lds`: LDS byte address of the centred table; `wlane_lds` / `newest` as demod_pair_relaxed (fully mirrored window).
// Loads in groups of seven pairs, two groups in flight; LDS returns in order, so "at most 14 outstanding" names the older group.
__device__ __forceinline__ float2v relax_sum_diff(float2v w)       // (lo, hi) = (w.hi + w.lo, w.hi - w.lo): the sample tap k meets is the pair's second word
{
    float2v r;
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(w));
    return r;
}
// (the loads as functions of their own: inline assembly inside a generic lambda under `if constexpr` does not compile)
template <int K, int NT>
__device__ __forceinline__ void relax_centred_load(float2v &w, float4v &t, uint32_t wa, uint32_t ta)
{
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(w) : "v"(wa), "n"(K), "n"(NT - 1 - K) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(t) : "v"(ta), "n"(16 * K) : "memory");
}
struct RelaxCentredGroup { float2v w[7]; float4v t[7]; };      // (lgkmcnt counts to 15: seven pairs = 14 loads per group)
template <int N>
__device__ __forceinline__ void relax_centred_wait(RelaxCentredGroup &R)
{
    static_assert(N <= 15, "lgkmcnt is four bits");
    asm volatile("s_waitcnt lgkmcnt(%14)"
                 : "+v"(R.w[0]), "+v"(R.w[1]), "+v"(R.w[2]), "+v"(R.w[3]), "+v"(R.w[4]), "+v"(R.w[5]), "+v"(R.w[6]),
                   "+v"(R.t[0]), "+v"(R.t[1]), "+v"(R.t[2]), "+v"(R.t[3]), "+v"(R.t[4]), "+v"(R.t[5]), "+v"(R.t[6])
                 : "n"(N));
}
// `wa`: LDS byte address of the lane's window slot that tap N-1 meets (the oldest of the NT samples, which must not wrap behind it)
template <int NT>
__device__ __forceinline__ void demod_pair_centred_at(uint32_t ctaps_lds, uint32_t wa, float *hm_out, float *hs_out)
{
    constexpr int H = NT / 2, G = 7, NG = (H + G - 1) / G;
    static_assert(NT % 2 == 0 && NT - 1 <= 255, "tap pairs; ds_read2st64's 8-bit slot offsets");
    using Group = RelaxCentredGroup;
    Group X, Y;
    float2v re[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, im[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};      // [parity of k]: .x mark, .y space
    auto load = [&](auto g_, Group &R) __attribute__((always_inline)) {
        constexpr int g = decltype(g_)::value;
        relax_static_for_([&](auto j_) __attribute__((always_inline)) {
            constexpr int j = decltype(j_)::value, k = g * G + j;
            if constexpr (k < H) relax_centred_load<k, NT>(R.w[j], R.t[j], wa, ctaps_lds);
        }, std::make_integer_sequence<int, G>{});
    };
    auto wait = [&](auto n_, Group &R) __attribute__((always_inline)) { relax_centred_wait<decltype(n_)::value>(R); };
    auto products = [&](auto g_, Group &R) __attribute__((always_inline)) {
        constexpr int g = decltype(g_)::value;
        // (all sums / differences of a group first, then the products: a packed operation that reads the result of the
        // instruction before it costs a wait state)
        relax_static_for_([&](auto j_) __attribute__((always_inline)) {
            constexpr int j = decltype(j_)::value;
            if constexpr (g * G + j < H) R.w[j] = relax_sum_diff(R.w[j]);
        }, std::make_integer_sequence<int, G>{});
        relax_static_for_([&](auto j_) __attribute__((always_inline)) {
            constexpr int j = decltype(j_)::value, k = g * G + j;
            if constexpr (k < H) {
                pk_fma_lo(re[k & 1], R.w[j], float2v{R.t[j].x, R.t[j].y});
                pk_fma_hi(im[k & 1], R.w[j], float2v{R.t[j].z, R.t[j].w});
            }
        }, std::make_integer_sequence<int, G>{});
    };
    // groups still to land behind group g: the next one's loads, if there is one (a short last group issues fewer)
    auto pending_after = [](int g) constexpr { const int k0 = (g + 1) * G; const int n = k0 >= H ? 0 : (H - k0 < G ? H - k0 : G); return 2 * n; };
    load(std::integral_constant<int, 0>{}, X);
    relax_static_for_([&](auto g_) __attribute__((always_inline)) {
        constexpr int g = decltype(g_)::value;
        if constexpr (g % 2 == 0) {
            if constexpr (g + 1 < NG) load(std::integral_constant<int, g + 1>{}, Y);
            wait(std::integral_constant<int, pending_after(g)>{}, X);
            products(g_, X);
        } else {
            if constexpr (g + 1 < NG) load(std::integral_constant<int, g + 1>{}, X);
            wait(std::integral_constant<int, pending_after(g)>{}, Y);
            products(g_, Y);
        }
    }, std::make_integer_sequence<int, NG>{});
    const float2v r = re[0] + re[1], i = im[0] + im[1];
    const float2v q = __builtin_elementwise_fma(i, i, r * r);         // (|mark|^2, |space|^2)
    *hm_out = __builtin_amdgcn_sqrtf(q.x);
    *hs_out = __builtin_amdgcn_sqrtf(q.y);
}
template <int NT, int RING>
__device__ __forceinline__ void demod_pair_centred(uint32_t ctaps_lds, uint32_t wlane_lds, uint32_t newest, float *hm_out, float *hs_out)
{
    demod_pair_centred_at<NT>(ctaps_lds, wlane_lds + (newest + (uint32_t)RING - (uint32_t)(NT - 1)) * (kWave * 4u), hm_out, hs_out);
}

// One AGC step, rx/agc.rs:72-77, as gain * (1 - bw |y|) + bw: the same update algebraically while gain >= 0 (the
// launchers check the floor), one fused multiply-add and the clamp on the gain's dependency chain.  bw_eff = 0 for a
// locked AGC: gain * 1 + 0.
__device__ __forceinline__ float agc_step_relaxed(const Params &P, float y, float &gain, float bw_eff)
{
    const float a = __builtin_fmaf(-bw_eff, fabsf(y), 1.0f);
    const float out = y * gain;
    gain = __builtin_amdgcn_fmed3f(__builtin_fmaf(gain, a, bw_eff), P.agc_min, P.agc_max);
    return out;
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

// ted_timing_relaxed evaluated AHEAD of the sample, for both of its signs (the loop reads the new sample only through
// rs_signum): the same operations on the same operands, so ted_commit (same_dev_common.h), once the sample is there,
// selects a result that is bit-identical to ted_timing_relaxed's.  The pipeline's stage 2 computes this while it
// waits for the helper's filters; what is left on the chain filters -> next instant is the selection.
__device__ __forceinline__ TedAhead ted_ahead_relaxed(const Params &P, const Lane &L, float inv_spt, float rem)
{
    TedAhead A;
    const float h0 = L.h1, h1 = L.h2;
    A.zero = h1;
    A.flags = L.flags ^ F_TED_PHASE;
    A.have = (A.flags & F_TED_PHASE) != 0;
    const float offset = __builtin_amdgcn_fmed3f(rem, -0.5f, 0.5f);
    const bool bw_locked = (L.flags & F_BW_LOCKED) != 0;
    const float alpha = bw_locked ? P.alpha_locked : P.alpha_unlocked;
    const float beta = bw_locked ? P.beta_locked : P.beta_unlocked;
    const float inst_plain = L.period_inst + offset;
    auto half = [&](float sg2, float *terr_out, float *avg_out, float *inst_out, int *cstar_out) __attribute__((always_inline)) {
        const float dsg = rs_signum(h0) - sg2;
        const float terr = h1 * dsg;
        const float e = __builtin_amdgcn_fmed3f(__builtin_fmaf(-offset, inv_spt, terr), -1.0f, 1.0f);
        const float avg1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(beta, e, L.period_avg), P.period_min, P.period_max);
        float inst1 = __builtin_fmaf(alpha, e, avg1) + offset;
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
    // The three modes -- EnabledTraining :278-301, EnabledFeedback :266-277, Disabled :262-265 -- as selects: a wavefront has
    // lanes in all of them at once (a burst trains for 32 symbols, then decides by itself), and three exec-mask regions of
    // three instructions each cost more than the instructions.  Disabled: err = 0, and the update below adds exact zeros.
    const bool training = mode == 2u, evolve = mode != 0u;
    const float sym_est = training ? ((L.eq_word & 1u) ? 1.0f : -1.0f) : rs_signum(sym_val);
    const float err = evolve ? sym_est - sym_val : 0.0f;
    L.eq_word = training ? L.eq_word >> 1 : L.eq_word;
    L.eq_count = training ? L.eq_count + 1u : L.eq_count;
    mode = (training && L.eq_count >= 32u) ? 1u : mode;
    {
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

// FastCtx (equalizer and its snapshot in registers, squelch history in LDS) with the relaxed equalizer step: the
// symbol stage of the pipeline's FASTMATH build
template <int NFF, int NFB>
struct RelaxFastCtx : FastCtx<NFF, NFB> {
    __device__ __forceinline__ uint32_t eq_symbols(const Params &P, Lane &L, const float *samples, int nsym)
    {
        uint32_t bits = 0;
#pragma unroll 1
        for (int b = 0; b < nsym; ++b)
            bits |= eq_symbol_relaxed<NFF, NFB>(P, L, this->ffc, this->ffw, this->fbc, this->fbw, samples[2 * b], samples[2 * b + 1]) << b;
        return bits;
    }
    __device__ __forceinline__ uint32_t eq_symbol1(const Params &P, Lane &L, float in0, float in1)
    { return eq_symbol_relaxed<NFF, NFB>(P, L, this->ffc, this->ffw, this->fbc, this->fbw, in0, in1); }
};

}  // namespace same
