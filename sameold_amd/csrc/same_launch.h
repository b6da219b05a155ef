// same_launch.h -- host-callable launchers implemented in same_kernels.hip / same_synth.hip
#pragma once

#include <hip/hip_runtime.h>

#include "same_device.h"

namespace same {

hipError_t launch_demod(const Params &P, const State &S, const Output &O, const float4 *taps,
                        const float *x, uint32_t n_samples, uint64_t counter0, hipStream_t stream);
hipError_t launch_demod_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                            const int16_t *x, uint32_t n_samples, uint64_t counter0, hipStream_t stream);
size_t demod_lds_bytes(const Params &P);
// latency-optimised kernel for the standard rates (same_kernels_fast.hip); whole blocks of
// fast_block_len() samples, the generic kernel takes the rest of a call
// longest block that can hold at most one TED instant for this configuration (same_config.cpp)
uint32_t max_block_len(const Params &P);
bool fast_kernel_supported(const Params &P);
uint32_t fast_block_len(const Params &P);   // samples per block of the fast kernel variant for this batch
// four-stage wavefront pipeline (same_kernels_pipe.hip): up to 32 768 channels at 22.05 kHz
bool pipe_kernel_selected(const Params &P);
uint32_t pipe_kernel_stages(const Params &P);     // 0 (not selected) or non-zero
uint32_t pipe_block_len(const Params &P);         // samples per block of the pipeline at this rate
// relaxed: the FASTMATH build (relaxed arithmetic, same_relaxed_common.h); only where pipe_relaxed_supported(P)
hipError_t launch_demod_pipe(const Params &P, const State &S, const Output &O, const float4 *taps,
                             const float *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                             const PipeChunks &chunks = PipeChunks{}, bool relaxed = false);
hipError_t launch_demod_pipe_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                                 const int16_t *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                                 const PipeChunks &chunks = PipeChunks{}, bool relaxed = false);
bool pipe_relaxed_supported(const Params &P);
// symbol-paced pipeline (same_kernels_sym.hip): relaxed arithmetic, 22.05 kHz, 36-sample steps, whole groups of 64 state
// columns; takes time-parallel chunks like the pipeline
bool sym_kernel_supported(const Params &P);
uint32_t sym_block_len(const Params &P);
hipError_t launch_demod_sym(const Params &P, const State &S, const Output &O, const float4 *taps,
                            const float *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                            const PipeChunks &chunks = PipeChunks{});
hipError_t launch_demod_sym_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                                const int16_t *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                                const PipeChunks &chunks = PipeChunks{});
uint32_t pipe_workgroup_channels(const Params &P);   // channels per workgroup the pipeline would use for this batch
// relaxed-arithmetic throughput kernel (same_kernels_relaxed.hip): 22.05 kHz, one wavefront per 64 state columns,
// whole blocks of relaxed_block_len() samples; takes time-parallel chunks like the pipeline
bool relaxed_kernel_supported(const Params &P);
uint32_t relaxed_block_len(const Params &P);      // samples per block of the form relaxed_kernel_kind(P) picks for P.n_channels columns
uint32_t relaxed_kernel_kind(const Params &P);    // 0 solo (one wavefront per 64 columns), 1 duo (two) (same_kernels_relaxed.hip)
hipError_t launch_demod_relaxed(const Params &P, const State &S, const Output &O, const float4 *taps,
                                const float *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                                const PipeChunks &chunks = PipeChunks{});
hipError_t launch_demod_relaxed_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                                    const int16_t *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream,
                                    const PipeChunks &chunks = PipeChunks{});
hipError_t launch_demod_fast(const Params &P, const State &S, const Output &O, const float4 *taps,
                             const float *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream);
hipError_t launch_demod_fast_i16(const Params &P, const State &S, const Output &O, const float4 *taps,
                                 const int16_t *x, uint32_t n_blocks, uint64_t counter0, hipStream_t stream);
// zero the launch cursors (publish = 0) or copy them to host-mapped memory (publish = 1)
hipError_t launch_counters(uint32_t *dev, uint32_t *host_mapped, int publish, hipStream_t stream);
// the log ordered by state column: first [n_bins + 1] (exclusive offsets), sorted [cap] (the records, a column's in any
// order, their `channel` replaced by their index in the log), cnt [n_bins] scratch; counters[0] = events logged
uint32_t event_sort_extra_words(uint32_t n_bins);      // words `first` needs behind its n_bins + 1 (the scan's workgroup totals)
hipError_t launch_event_sort(const DevEvent *ev, const uint32_t *counters, uint32_t cap, uint32_t n_bins, uint32_t *cnt, uint32_t *first,
                             DevEvent *sorted, hipStream_t stream, bool cnt_is_zero = false);      // cnt_is_zero: the caller emptied cnt on this stream
// (first_col: columns before it are left alone -- the time-parallel launches copy the channels' own state over them next)
hipError_t launch_init_state(const Params &P, const State &S, int is_reset, hipStream_t stream, uint32_t first_col = 0);
// Column copies between state blobs of different widths: for every array of `desc` (device memory,
// n_desc entries) and every column col < n_cols, dst[row][col] = src[row][src_col ? src_col[col] : col + src_base].
hipError_t launch_copy_state_columns(const StateArrayDesc *desc, uint32_t n_desc, uint32_t src_channels,
                                     uint32_t dst_channels, const uint32_t *src_col, uint32_t n_cols, hipStream_t stream,
                                     uint32_t src_base = 0);   // src_col == nullptr: source column = col + src_base
// final_col[c] = the state column (chunk * in_channels + c) whose chunk ran to the end of the input
hipError_t launch_chunk_final_column(const uint64_t *handover, uint32_t in_channels, ChunkGeom g, uint32_t *final_col,
                                     hipStream_t stream);
// ... with per-channel boundaries: own_start [n_chunks][in_channels], relative to the call's first sample (counter0)
hipError_t launch_chunk_final_column_pc(const uint64_t *handover, const uint32_t *own_start, uint32_t in_channels, uint32_t n_chunks,
                                        uint64_t counter0, uint32_t *final_col, hipStream_t stream);
hipError_t launch_fill_u64(uint64_t *p, size_t n, uint64_t v, hipStream_t stream);
hipError_t launch_cast_i16_f32(const int16_t *in, float *out, size_t n, hipStream_t stream);      // out[i] = (float)in[i]
// The start of a time-parallel launch: blob[0 .. bytes) = fresh[0 .. bytes) (a template of freshly built receivers, 16-byte
// multiples), handover[0 .. n_cols) = kNoHandover, sort_cnt[0 .. n_cols) = 0 (may be null), counters[0 .. 3) = 0 (may be null)
hipError_t launch_tp_prologue(void *blob, const void *fresh, size_t bytes, uint64_t *handover, uint32_t *sort_cnt, uint32_t n_cols,
                              uint32_t *counters, hipStream_t stream);
// Per-channel chunk boundaries for a channel-major input (time-parallel mode): an energy scout over
// one 64-byte sector per 256-sample block, then per channel the idle instant nearest to every nominal boundary.
uint32_t tp_scout_block();      // samples per energy reading
struct TpPlan {
    uint32_t channels, n_chunks, block_len, warmup_samples, whole_samples;
    uint64_t in_samples;      // pitch of a channel in the input
    uint32_t scout_blocks;    // whole_samples / 256
};
// energy [channels][scout_blocks] (scratch), own_start [n_chunks][channels], row0 / nominal / perm [n_chunks * channels],
// wg_blocks [n_chunks * channels / 64]; perm_out / wg_blocks_out (same sizes, may be
// null; wg_blocks_out [2][workgroups]): the workgroups of perm / wg_blocks once more, longest first
hipError_t launch_tp_plan(const float *x, const TpPlan &g, float *energy, uint32_t *own_start, uint32_t *row0,
                          uint32_t *nominal, uint32_t *perm, uint32_t *wg_blocks, bool sorted, hipStream_t stream,
                          uint32_t *perm_out = nullptr, uint32_t *wg_blocks_out = nullptr, bool pairs = false);
hipError_t launch_transpose_f32(const float *in, float *out, uint32_t n_channels, uint32_t n_samples,
                                hipStream_t stream);
hipError_t launch_transpose_i16(const int16_t *in, int16_t *out, uint32_t n_channels, uint32_t n_samples,
                                hipStream_t stream);

// synthetic workload (same_synth.hip)
struct SynthParams {
    uint32_t n_channels;
    uint32_t input_rate;
    uint64_t seed;
    float noise_sigma;     // AWGN standard deviation relative to the carrier amplitude (0 = none)
    uint32_t flags;        // bit 0: integer (even) samples per symbol like the reference's test modulator
};
hipError_t launch_synth(const SynthParams &sp, float *x, size_t n_samples, hipStream_t stream);
struct TrialParams {
    uint32_t n_trials;     // trials in this buffer (one per channel)
    uint32_t first_trial;  // global id of trial 0 of this buffer
    uint32_t input_rate;
    uint32_t n_grid;       // Eb/N0 grid points; trial t uses point t mod n_grid
    uint64_t seed;
    float ebn0_db_lo, ebn0_db_step;
};
hipError_t launch_trials(const TrialParams &tp, float *x, size_t n_samples, hipStream_t stream);
// host mirror of the per-channel payload the generator transmits (header text)
uint32_t synth_payload(uint64_t seed, uint32_t channel, uint8_t *out, uint32_t cap);

}  // namespace same
