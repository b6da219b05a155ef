// same_config.h -- host-side mirror of SameReceiverBuilder and the constants derived
// from it in `impl From<&SameReceiverBuilder> for SameReceiver` (receiver.rs:502-560).
#pragma once

#include <cstdint>
#include <vector>

#include "same_device.h"

// The C ABI's opaque builder.  Field-for-field SameReceiverBuilder (rx/builder.rs:22-37)
// with EqualizerBuilder (:359-365) flattened into eq_*.
struct same_rx_builder {
    uint32_t input_rate;
    float dc_blocker_len;
    float agc_bandwidth;
    float agc_gain_limits[2];
    float timing_bandwidth_unlocked;
    float timing_bandwidth_locked;
    float timing_max_deviation;
    float squelch_power_open;
    float squelch_power_close;
    float squelch_bandwidth;
    uint32_t preamble_max_errors;
    bool equalizer;                 // Option<EqualizerBuilder>::is_some()
    uint32_t eq_nfeedforward, eq_nfeedback;
    float eq_relaxation, eq_regularization;
    uint32_t frame_prefix_max_errors;
    uint32_t frame_max_invalid_bytes;
};

namespace same {

// Rust f32::clamp / f32::min / f32::max on the host
float rs_clamp_h(float x, float mn, float mx);

void builder_defaults(same_rx_builder &b, uint32_t input_rate);

// Derive the per-batch constants.  Returns 0 or a SAME_E* code where the reference
// panics.  `taps` receives ntaps entries of (mark.re, mark.im, space.re, space.im) and, for an even tap count, ntaps / 2
// entries of the same taps centred (Re mark, Re space, Im mark, Im space: same_config.cpp) behind them.
int derive_params(const same_rx_builder &b, uint32_t n_channels, Params &P,
                  std::vector<float> &taps);

// largest B in {16,8,4,2,1} such that a TED cannot fire twice within B samples
uint32_t choose_block_len(const Params &P);

}  // namespace same
