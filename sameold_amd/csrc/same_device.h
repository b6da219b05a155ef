// same_device.h -- structures shared by the host side (same_batch.cpp) and the gfx950
// kernels (same_kernels.hip).  Plain PODs, passed to kernels by value.
//
// Vocabulary follows the reference (sameold 0.6.0): channel = one SameReceiver,
// TED = timing error detector instant, burst = one framed transmission.
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace same {

constexpr int kWave = 64;            // gfx950 wavefront: one lane per channel
constexpr int kMaxEqTaps = 16;       // SAME_MAX_EQ_TAPS
constexpr int kBurstCap = 288;       // SAME_EVENT_MAX_BYTES: per-channel framer buffer
constexpr int kSquelchHist = 64;     // rx/codesquelch.rs:143 sample_history
constexpr int kMaxTaps = 512;        // matched-filter taps held in constant memory

// Per-batch constants derived once from the builder (receiver.rs:502-560).  Uniform
// across channels, so they live in SGPRs / scalar loads.
struct Params {
    uint32_t n_channels;
    uint32_t input_rate;
    // DC blocker (rx/dcblock.rs)
    uint32_t dc_len;          // (dc_blocker_len * sps) as usize, receiver.rs:509
    float dc_inv_len;         // 1.0 / len, rx/dcblock.rs:78
    float dc_k;               // (len > 1) as f32, rx/dcblock.rs:48
    // AGC (rx/agc.rs)
    float agc_bw, agc_min, agc_max, agc_gain0;
    // matched filters (rx/waveform.rs:39-64)
    uint32_t ntaps;
    uint32_t win_ring;        // LDS ring slots for the AGC-output window (power of two)
    // timing loop (rx/symsync.rs:142-163, 329-337)
    float samples_per_ted, period_min, period_max;
    float alpha_unlocked, beta_unlocked, alpha_locked, beta_locked;
    // squelch (rx/codesquelch.rs:190-212)
    uint32_t sync_word, sq_max_errors;
    float sq_power_open, sq_power_close, sq_bw;
    // equalizer (rx/equalize.rs:127-153; disabled -> 1/1 taps, relaxation 0, receiver.rs:585-590)
    uint32_t eq_nff, eq_nfb;
    float eq_relaxation, eq_regularization;
    // framer (rx/framing.rs:71-78)
    uint32_t fr_max_prefix_errors, fr_max_invalid;
    // deferred-TED block length: at most one TED instant per lane per block
    uint32_t block_len;
    uint32_t trace_cap;       // soft-symbol trace entries per channel (0 = off)
    // transport wake-ups (SAME_DEV_TICK events) for the host-side assembler; 0 = link only
    uint32_t ticks;
    uint32_t tick_interburst; // MAX_INTERBURST_SYMBOLS  (rx/assembler.rs:85)
    uint32_t tick_history;    // MAX_HISTORY_DURATION    (rx/assembler.rs:92-93)
    // Test / profiling overrides of the kernel dispatch.  Read from the environment ONCE, when the
    // batch is created (same_batch.cpp: read_knobs); nothing on the launch path calls getenv.
    // Tri-state knobs: 0 = unset (a zero-initialised Params behaves like the defaults), +1 = on, -1 = off.
    int32_t knob_pipe;        // SAME_PIPE=0 -> -1 never the wavefront pipeline, SAME_PIPE=1 -> +1 whenever it applies
    int32_t knob_pipe_lanes;  // SAME_PIPE_LANES: 0 unset, else 16 / 32 / 64 channels per workgroup
    int32_t knob_pipe_split;  // SAME_PIPE_SPLIT
    int32_t knob_mirror;      // SAME_MIRROR (one-wavefront kernel's mirrored window)
    int32_t knob_fast_dense;  // SAME_FAST_DENSE (one-wavefront kernel: the two-per-SIMD build whatever the channel count)
    int32_t knob_pipe_share;  // SAME_PIPE_SHARE (the two-workgroups-per-CU register budget whatever the channel count)
    int32_t knob_relaxed_kernel;  // SAME_RELAXED_KERNEL: 0 choose (incl. the pipeline's FASTMATH build), 1 "solo" (one wavefront per 64 columns), 2 "duo" (two)
    int32_t knob_sym;         // SAME_SYM=0 -> -1: relaxed launches keep the 20-sample pipeline's FASTMATH build instead of the symbol-paced pipeline (same_kernels_sym.hip)
    int32_t knob_prio;        // SAME_PIPE_PRIO: knock-out mask of SAME_PROFILE builds (same_profile.h PROF_SKIP); unused otherwise
};

// flag bits of State::flags
enum : uint32_t {
    F_AGC_LOCKED = 1u << 0,     // Agc::locked
    F_TED_PHASE = 1u << 1,      // ZeroCrossingTed::sample_counter (mod 2)
    F_BW_LOCKED = 1u << 2,      // timing loop uses the locked (alpha,beta)
    F_SQ_LOCK = 1u << 3,        // CodeAndPowerSquelch::sync_lock
    F_EQ_MODE_SHIFT = 4,        // 2 bits: 0 disabled (unused), 1 feedback, 2 training
    F_EQ_MODE_MASK = 3u << 4,
    F_FR_STATE_SHIFT = 6,       // 2 bits: 0 idle, 1 prefix search, 2 data read
    F_FR_STATE_MASK = 3u << 6,
    F_LINK_SHIFT = 8,           // 2 bits: last reported LinkState kind
    F_LINK_MASK = 3u << 8,
    F_TICK_AGAIN = 1u << 10,    // report the next transport poll instant too
    F_EQ_BITS_SHIFT = 16,       // 8 bits: symbols of the current byte already equalized
    F_EQ_BITS_MASK = 0xffu << 16,
};
constexpr int kTickRing = 8;     // pending burst+interburst deadlines per channel

// Per-channel receiver state, structure-of-arrays in HBM: every pointer is an array
// indexed [channel] (or [slot * n_channels + channel] for rings) so a wavefront's 64
// lanes touch 256 contiguous bytes.  It is read at the start of a launch and written
// back at the end, so a stream can be fed in arbitrary chunks and paused at any point.
struct State {
    // high-rate stages
    float *dc_ff_ring;     // [dc_len][C]   MovingAverage window (feed-forward)
    float *dc_fb_ring;     // [dc_len][C]
    float *dc_sum0, *dc_sum1;
    float *agc_gain;
    float *win_ring;       // [win_ring][C] AGC output history feeding the matched filters
    uint32_t *ted_clock;   // ted_sample_clock, receiver.rs:87
    float *until_next_ted; // samples_until_next_ted, receiver.rs:88
    // timing loop
    float *ted_h0, *ted_h1, *ted_h2;
    float *period_avg, *period_inst;
    // squelch
    uint32_t *sq_data;     // CodeCorrelator::data
    float *sq_power;       // PowerTracker::power
    uint32_t *sq_phist;    // 32 newest power flags, bit 0 = newest
    uint32_t *sq_fill;     // samples in sample_history, saturating at 64
    int32_t *sq_clock;     // sample_clock: -1 = None
    uint64_t *sq_symbols;  // symbol_counter
    float *sq_hist;        // [64][C] sample_history ring, slot = (2*symbol_counter + k) & 63
    // equalizer
    float *eq_ffc, *eq_fbc;    // [nff][C], [nfb][C] coefficients
    float *eq_ffw, *eq_fbw;    // [nff][C], [nfb][C] windows, index 0 = oldest
    uint32_t *eq_word, *eq_count;
    // equalizer as of the last completed byte (restored if the byte clock is re-aligned)
    float *eq_snap_ffc, *eq_snap_fbc, *eq_snap_ffw, *eq_snap_fbw;
    // framer
    uint32_t *fr_word, *fr_count, *fr_invalid, *fr_len;
    uint8_t *fr_msg;       // [C][kBurstCap]
    uint32_t *flags;
    // transport wake-ups: symbol deadlines armed by bursts, sample deadline armed by the host
    uint64_t *tk_next;     // earliest pending symbol deadline (~0 = none)
    uint64_t *tk_last;     // last burst + MAX_HISTORY_DURATION (~0 = none / already reported)
    uint64_t *tk_ring;     // [kTickRing][C] burst + MAX_INTERBURST_SYMBOLS, oldest first
    uint32_t *tk_n;        // entries in tk_ring
    uint64_t *wake_sample; // force-EOM instant (receiver.rs:321-324), 0 = none; HOST-owned, read-only here
    uint64_t *wake_fired;  // the wake_sample value already reported (device-owned)
    // soft-symbol trace (optional)
    uint32_t *trace_n;     // [C]
    float *trace;          // [C][trace_cap][4]
    uint64_t *trace_idx;   // [C][trace_cap]
};

// Time-parallel chunks (same_kernels_pipe.hip).  A launch over n_chunks * in_channels
// state columns: column chunk * in_channels + cin demodulates input column cin from block
// chunk * stride_blocks of the call on.  Chunk 0 starts from the channel's real state, the others from
// a reset receiver `warm-up` samples before the range they own; every chunk but the last runs
// nominal_blocks blocks and then on until each of its lanes has been seen idle (NoCarrier), records
// that instant in handover[column] and leaves; the last one runs to the end of the input.
// n_chunks <= 1: an ordinary launch (all other fields ignored).
struct PipeChunks {
    uint32_t n_chunks;
    uint32_t in_channels;      // channels of the input rows (multiple of the workgroup width)
    uint32_t stride_blocks;    // blocks between the first rows of consecutive chunks
    uint32_t nominal_blocks;   // warm-up + own range, in blocks, of every chunk but the last
    uint64_t *handover;        // [n_chunks * in_channels] input sample counter of the hand-over, ~0 = never
    // Per-column geometry (channel-major input, boundaries chosen per channel at idle instants; null = the
    // uniform geometry above).  Column v reads its channel's samples from col_row0[v] on and may hand over
    // after col_nominal[v] blocks; a workgroup runs wg_blocks[workgroup] blocks at most.  Columns of chunk 0 and
    // of the last chunk share their row within a workgroup (they load / store real state).
    const uint32_t *col_row0;      // [n_chunks * in_channels] first sample, relative to the call's first; multiple of 4
    const uint32_t *col_nominal;   // [n_chunks * in_channels] blocks before a hand-over is allowed
    const uint32_t *wg_blocks;     // [workgroups]
    const uint32_t *col_perm;      // [n_chunks * in_channels] grid position -> state column (pieces of similar length share a
                                   // workgroup; chunk 0 and last-chunk columns keep workgroups of their own), null = identity
    float *hist_scratch;           // with col_perm: [kSquelchHist][n_chunks * in_channels] by GRID POSITION, or null.  A kernel that keeps
                                   // the squelch's sample history in global memory (same_kernels_sym.hip) works on this copy: a
                                   // wavefront's permuted columns are 64 different cache lines per access of the state array
    uint64_t in_samples;           // channel-major input: samples per channel (the pitch of a channel)
    uint32_t whole_samples;        // samples of the call that are whole blocks
};

// Geometry of one time-parallel call, shared by the device (which chunk's final state becomes the
// channel's state) and the host (which chunk's events are kept when): chunk k >= 1 owns the samples
// from own_start(k) on, and a chunk that hands over at input sample counter h is followed by the chunk
// that owns h.
struct ChunkGeom {
    uint64_t counter0;       // input sample counter of the call's first sample
    uint32_t n_chunks, block_len, stride_blocks, warmup_blocks;
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t owner_of(uint64_t h) const
    {
        const uint64_t blocks = (h - counter0) / block_len;
        if (blocks < warmup_blocks) return 0u;
        const uint64_t k = (blocks - warmup_blocks) / stride_blocks;
        return k >= n_chunks ? n_chunks - 1u : (uint32_t)k;
    }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint64_t own_start(uint32_t k) const
    { return k == 0u ? counter0 : counter0 + ((uint64_t)k * stride_blocks + warmup_blocks) * block_len; }
};
constexpr uint64_t kNoHandover = ~0ull;

// one State array for the column copies between the real and the time-parallel state blobs
struct StateArrayDesc { char *src; char *dst; uint32_t rows; uint32_t elem_words; };

// One link-layer event as the device records it.  32 bytes.
constexpr uint32_t kDevEventNone = 0xffffffffu;   // kind of a reserved but unused log slot (skipped by the host)
struct DevEvent {
    uint32_t channel;
    uint32_t kind;           // SAME_LINK_*, 8 = transport wake-up, kDevEventNone = empty slot
    uint64_t sample_counter;
    uint64_t symbol_count;
    uint32_t burst_len;      // true length (may exceed kBurstCap)
    uint32_t burst_slot;     // index into the burst pool, 0xffffffff = none
};

// Append-only output of one launch.
struct Output {
    DevEvent *events;
    uint32_t *n_events;      // atomic cursor
    uint32_t event_cap;
    uint8_t *bursts;         // [burst_cap][kBurstCap]
    uint32_t *n_bursts;      // atomic cursor
    uint32_t burst_cap;
    uint32_t *overflow;      // set non-zero when a cursor passed its capacity
};

}  // namespace same
