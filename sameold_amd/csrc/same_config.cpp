// same_config.cpp -- SameReceiverBuilder mirror (C ABI) and derivation of the per-batch
// constants.  Host code; built with -ffp-contract=off so every f32 expression rounds
// once per operation, as rustc's output does.
//
// Citations: file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
#include "same_config.h"

#include <cmath>
#include <cstring>
#include <new>

#include "../../include/same_rx.h"

namespace same {

float rs_clamp_h(float x, float mn, float mx)
{
    if (x < mn) x = mn;
    if (x > mx) x = mx;
    return x;
}
static float rs_min_h(float a, float b) { return std::fmin(a, b); }   // f32::min ignores NaN
static float rs_max_h(float a, float b) { return std::fmax(a, b); }

static size_t as_usize(float f)
{
    // `f as usize`: truncating, saturating, NaN -> 0
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return (size_t)-1;
    return (size_t)f;
}

void builder_defaults(same_rx_builder &b, uint32_t input_rate)
{
    // rx/builder.rs:50-67, 369-376
    b.input_rate = input_rate;
    b.dc_blocker_len = 0.38f;
    b.agc_bandwidth = 0.01f;
    b.agc_gain_limits[0] = 0.0f;
    b.agc_gain_limits[1] = 1.0e6f;
    b.timing_bandwidth_unlocked = 0.125f;
    b.timing_bandwidth_locked = 0.05f;
    b.timing_max_deviation = 0.01f;
    b.squelch_power_open = 0.10f;
    b.squelch_power_close = 0.05f;
    b.squelch_bandwidth = 0.125f;
    b.preamble_max_errors = 2;
    b.equalizer = true;
    b.eq_nfeedforward = 6;
    b.eq_nfeedback = 4;
    b.eq_relaxation = 0.05f;
    b.eq_regularization = 1.0e-6f;
    b.frame_prefix_max_errors = 2;
    b.frame_max_invalid_bytes = 5;
}

// rx/waveform.rs:5-12
static const float FSK_MARK_HZ = 2083.3f;
static const float FSK_SPACE_HZ = 1562.5f;
static const float BAUD_HZ = 520.83f;
static const float PI_F32 = 3.14159274101257324f;
static const uint32_t PREAMBLE_SYNC_WORD = 0xababababu;   // rx/waveform.rs:26

// compute_loop_alphabeta rx/symsync.rs:329-337
static void loop_alphabeta(float bw, float &alpha, float &beta)
{
    float omega = (2.0f * PI_F32) * bw;
    float k0 = 2.0f;
    float k1 = std::exp(-omega);       // f32::exp -> expf
    float sh = std::sinh(omega);       // f32::sinh -> sinhf
    alpha = (k0 * k1) * sh;
    float t = sh + 1.0f;
    float u = k1 * t;
    float v = 1.0f - u;
    beta = k0 * v;
}

// cisoid_matched_filter rx/waveform.rs:54-64
static void cisoid(uint32_t points, float freq_fs, float *re, float *im)
{
    for (uint32_t it = 0; it < points; ++it) {
        float a = (2.0f * PI_F32) * freq_fs;
        float theta = a * (float)(points - 1 - it);
        float r = std::exp(0.0f);                  // Complex::exp: from_polar(re.exp(), im)
        float ere = r * std::cos(theta);
        float eim = r * std::sin(theta);
        float cre = ere, cim = -eim;               // conj()
        float sre = 2.0f * cre, sim = 2.0f * cim;  // 2.0f32 * Complex
        re[it] = sre / (float)points;              // Complex / f32
        im[it] = sim / (float)points;
    }
}

uint32_t max_block_len(const Params &P)
{
    // Shortest period the timing loop can command (rx/symsync.rs:219-244): period_inst =
    // period_avg + alpha*err + offset with period_avg >= period_min, |err| <= 1,
    // |offset| <= 0.5, and a negative result replaced by period_avg.  The larger
    // (unlocked) alpha bounds both modes.  After a TED instant the sample clock counts
    // 1, 2, ... and fires at the first count c with period - c < 0.5, so no second
    // instant can fall inside a block of B samples when B - 1 <= lower_bound - 0.5.
    float a = P.alpha_unlocked > P.alpha_locked ? P.alpha_unlocked : P.alpha_locked;
    float lb = P.period_min - a - 0.5f;
    if (!(lb > 0.0f)) lb = 0.0f;
    // one more sample of margin for the f32 rounding of (period - clock)
    const float maxb = std::floor(lb - 0.5f);
    return maxb >= 1.0f ? (uint32_t)maxb : 1u;
}

uint32_t choose_block_len(const Params &P)
{
    const uint32_t maxb = max_block_len(P);
    uint32_t B = 1;
    for (uint32_t cand : {16u, 8u, 4u, 2u}) {
        if (cand <= maxb) { B = cand; break; }
    }
    return B;
}

int derive_params(const same_rx_builder &b, uint32_t n_channels, Params &P, std::vector<float> &taps)
{
    std::memset(&P, 0, sizeof(P));
    P.n_channels = n_channels;
    P.input_rate = b.input_rate;
    const float sps = (float)b.input_rate / BAUD_HZ;                    // rx/waveform.rs:29-31

    // DC blocker: receiver.rs:509, rx/dcblock.rs:73-80
    size_t dc_len = as_usize(b.dc_blocker_len * sps);
    if (dc_len == 0) return SAME_EDCLEN;
    if (dc_len > (1u << 20)) return SAME_EINVAL;
    P.dc_len = (uint32_t)dc_len;
    P.dc_inv_len = 1.0f / (float)dc_len;
    P.dc_k = dc_len > 1 ? 1.0f : 0.0f;

    // AGC: receiver.rs:510-514, rx/agc.rs:49-57
    if (!(b.agc_gain_limits[0] <= b.agc_gain_limits[1])) return SAME_EAGCLIMITS;
    float t = b.agc_bandwidth * sps;
    float bw = t / (float)b.input_rate;
    P.agc_bw = rs_clamp_h(bw, 0.0f, 1.0f);
    P.agc_min = b.agc_gain_limits[0];
    P.agc_max = b.agc_gain_limits[1];
    P.agc_gain0 = rs_min_h(1.0f, b.agc_gain_limits[0]);

    // matched filters: rx/waveform.rs:39-44
    size_t ntaps = as_usize(std::floor(sps));
    if (ntaps < 1 || ntaps > (size_t)kMaxTaps) return SAME_ERATE;
    P.ntaps = (uint32_t)ntaps;
    std::vector<float> mre(ntaps), mim(ntaps), sre(ntaps), sim(ntaps);
    cisoid(P.ntaps, FSK_MARK_HZ / (float)b.input_rate, mre.data(), mim.data());
    cisoid(P.ntaps, FSK_SPACE_HZ / (float)b.input_rate, sre.data(), sim.data());
    taps.resize(4 * ntaps);
    for (size_t i = 0; i < ntaps; ++i) {
        taps[4 * i + 0] = mre[i]; taps[4 * i + 1] = mim[i];
        taps[4 * i + 2] = sre[i]; taps[4 * i + 3] = sim[i];
    }
    // The same taps once more, CENTRED, behind them (even tap counts; relaxed kernels only -- same_kernels_sym.hip).  The
    // filter is a cisoid h[i] = (2/N) e^{-j a (N-1-i)} (rx/waveform.rs:54-64) and only its output's magnitude is used
    // (rx/demod.rs:156-164), so it may be turned by any unit phasor: u[k] = h[k] e^{+j a (N-1)/2} = (2/N) e^{j a (k - (N-1)/2)}
    // has u[N-1-k] = conj(u[k]), and for a real window  sum_i w_i u_i = sum_{k<N/2} (w_k + w_{N-1-k}) Re u_k + j (w_k - w_{N-1-k}) Im u_k.
    // Entry k < N/2: (Re u_k mark, Re u_k space, Im u_k mark, Im u_k space); the reference's own f32 taps turned in f64.
    if (ntaps % 2 == 0) {
        taps.resize(4 * ntaps + 4 * (ntaps / 2));
        const float am = (2.0f * PI_F32) * (FSK_MARK_HZ / (float)b.input_rate), as = (2.0f * PI_F32) * (FSK_SPACE_HZ / (float)b.input_rate);
        const double hm = (double)am * 0.5 * (double)(ntaps - 1), hs = (double)as * 0.5 * (double)(ntaps - 1);
        for (size_t k = 0; k < ntaps / 2; ++k) {
            float *t = &taps[4 * ntaps + 4 * k];
            t[0] = (float)((double)mre[k] * std::cos(hm) - (double)mim[k] * std::sin(hm));
            t[1] = (float)((double)sre[k] * std::cos(hs) - (double)sim[k] * std::sin(hs));
            t[2] = (float)((double)mre[k] * std::sin(hm) + (double)mim[k] * std::cos(hm));
            t[3] = (float)((double)sre[k] * std::sin(hs) + (double)sim[k] * std::cos(hs));
        }
    }

    // timing loop: rx/symsync.rs:142-163
    loop_alphabeta(b.timing_bandwidth_unlocked, P.alpha_unlocked, P.beta_unlocked);
    loop_alphabeta(b.timing_bandwidth_locked, P.alpha_locked, P.beta_locked);
    P.samples_per_ted = sps / 2.0f;
    float dev = sps * rs_clamp_h(b.timing_max_deviation, 0.0f, 0.5f);
    P.period_min = P.samples_per_ted - dev;
    P.period_max = P.samples_per_ted + dev;

    // squelch: receiver.rs:517-523, rx/codesquelch.rs:190-212, 464-469
    P.sync_word = PREAMBLE_SYNC_WORD;
    P.sq_max_errors = b.preamble_max_errors;
    P.sq_power_open = b.squelch_power_open;
    P.sq_power_close = rs_min_h(b.squelch_power_close, b.squelch_power_open);
    P.sq_bw = rs_clamp_h(b.squelch_bandwidth, 0.0f, 1.0f);

    // equalizer: receiver.rs:524-534, 585-590
    if (b.equalizer) {
        P.eq_nff = b.eq_nfeedforward; P.eq_nfb = b.eq_nfeedback;
        P.eq_relaxation = b.eq_relaxation; P.eq_regularization = b.eq_regularization;
    } else {
        P.eq_nff = 1; P.eq_nfb = 1; P.eq_relaxation = 0.0f; P.eq_regularization = 1.0e-6f;
    }
    if (P.eq_nff < 1 || P.eq_nfb < 1 || P.eq_nff > (uint32_t)kMaxEqTaps || P.eq_nfb > (uint32_t)kMaxEqTaps)
        return SAME_EEQORDER;

    // framer: receiver.rs:535
    P.fr_max_prefix_errors = b.frame_prefix_max_errors;
    P.fr_max_invalid = b.frame_max_invalid_bytes;

    P.block_len = choose_block_len(P);
    uint32_t need = P.ntaps + P.block_len - 1;
    uint32_t ring = 1;
    while (ring < need) ring <<= 1;
    P.win_ring = ring;
    return SAME_OK;
}

}  // namespace same

// ------------------------------------------------------------------------------------
// C ABI: builder
// ------------------------------------------------------------------------------------
using same::rs_clamp_h;

extern "C" {

uint32_t same_rx_abi_version(void) { return SAME_RX_ABI_VERSION; }

same_rx_builder *same_rx_builder_new(uint32_t input_rate)
{
    same_rx_builder *b = new (std::nothrow) same_rx_builder;
    if (b) same::builder_defaults(*b, input_rate);
    return b;
}
same_rx_builder *same_rx_builder_default(void) { return same_rx_builder_new(22050); }
same_rx_builder *same_rx_builder_clone(const same_rx_builder *src)
{
    if (!src) return nullptr;
    same_rx_builder *b = new (std::nothrow) same_rx_builder;
    if (b) *b = *src;
    return b;
}
void same_rx_builder_free(same_rx_builder *b) { delete b; }

void same_rx_builder_with_dc_blocker_length(same_rx_builder *b, float len)
{ b->dc_blocker_len = same::rs_max_h(0.0f, len); }
void same_rx_builder_with_agc_bandwidth(same_rx_builder *b, float bw)
{ b->agc_bandwidth = rs_clamp_h(bw, 0.0f, 1.0f); }
void same_rx_builder_with_agc_gain_limits(same_rx_builder *b, float mn, float mx)
{ b->agc_gain_limits[0] = mn; b->agc_gain_limits[1] = mx; }
void same_rx_builder_with_timing_bandwidth(same_rx_builder *b, float unlocked, float locked)
{
    b->timing_bandwidth_unlocked = rs_clamp_h(unlocked, 0.0f, 1.0f);
    b->timing_bandwidth_locked = rs_clamp_h(locked, 0.0f, b->timing_bandwidth_unlocked);
}
void same_rx_builder_with_timing_max_deviation(same_rx_builder *b, float d)
{ b->timing_max_deviation = rs_clamp_h(d, 0.0f, 0.5f); }
void same_rx_builder_with_squelch_power(same_rx_builder *b, float open, float close)
{
    b->squelch_power_open = rs_clamp_h(open, 0.0f, 1.0f);
    b->squelch_power_close = same::rs_min_h(close, open);
}
void same_rx_builder_with_squelch_bandwidth(same_rx_builder *b, float bw) { b->squelch_bandwidth = bw; }
void same_rx_builder_with_preamble_max_errors(same_rx_builder *b, uint32_t e) { b->preamble_max_errors = e; }
void same_rx_builder_with_adaptive_equalizer(same_rx_builder *b, uint32_t nff, uint32_t nfb,
                                             float relaxation, float regularization)
{
    // EqualizerBuilder::with_filter_order / with_relaxation / with_regularization rx/builder.rs:393-425
    b->equalizer = true;
    b->eq_nfeedforward = nff > 1 ? nff : 1;
    b->eq_nfeedback = nfb < 1 ? 1 : (nfb > b->eq_nfeedforward ? b->eq_nfeedforward : nfb);
    b->eq_relaxation = rs_clamp_h(relaxation, 0.0f, 1.0f);
    b->eq_regularization = rs_clamp_h(regularization, 0.0f, 3.40282347e+38f);
}
void same_rx_builder_without_adaptive_equalizer(same_rx_builder *b) { b->equalizer = false; }
void same_rx_builder_with_frame_prefix_max_errors(same_rx_builder *b, uint32_t e)
{ b->frame_prefix_max_errors = e > 7 ? 7 : e; }
void same_rx_builder_with_frame_max_invalid(same_rx_builder *b, uint32_t n) { b->frame_max_invalid_bytes = n; }

uint32_t same_rx_builder_input_rate(const same_rx_builder *b) { return b->input_rate; }
float same_rx_builder_dc_blocker_length(const same_rx_builder *b) { return b->dc_blocker_len; }
float same_rx_builder_agc_bandwidth(const same_rx_builder *b) { return b->agc_bandwidth; }
void same_rx_builder_agc_gain_limits(const same_rx_builder *b, float out[2])
{ out[0] = b->agc_gain_limits[0]; out[1] = b->agc_gain_limits[1]; }
void same_rx_builder_timing_bandwidth(const same_rx_builder *b, float *unlocked, float *locked)
{ *unlocked = b->timing_bandwidth_unlocked; *locked = b->timing_bandwidth_locked; }
float same_rx_builder_timing_max_deviation(const same_rx_builder *b) { return b->timing_max_deviation; }
void same_rx_builder_squelch_power(const same_rx_builder *b, float *open, float *close)
{ *open = b->squelch_power_open; *close = b->squelch_power_close; }
float same_rx_builder_squelch_bandwidth(const same_rx_builder *b) { return b->squelch_bandwidth; }
uint32_t same_rx_builder_preamble_max_errors(const same_rx_builder *b) { return b->preamble_max_errors; }
int same_rx_builder_adaptive_equalizer(const same_rx_builder *b, uint32_t *nff, uint32_t *nfb,
                                       float *relaxation, float *regularization)
{
    if (!b->equalizer) return 0;
    if (nff) *nff = b->eq_nfeedforward;
    if (nfb) *nfb = b->eq_nfeedback;
    if (relaxation) *relaxation = b->eq_relaxation;
    if (regularization) *regularization = b->eq_regularization;
    return 1;
}
uint32_t same_rx_builder_frame_prefix_max_errors(const same_rx_builder *b) { return b->frame_prefix_max_errors; }
uint32_t same_rx_builder_frame_max_invalid(const same_rx_builder *b) { return b->frame_max_invalid_bytes; }

}  // extern "C"
