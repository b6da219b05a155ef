/*
 * same_oracle.h -- CPU restatement of sameold 0.6.0's receive chain.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the MI355X batched
 * demodulator in sameold_amd/.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product path never does.
 *
 * Parity status: PINNED at the text level against the reference's own golden
 * vectors (sample/{npt,two_and_two,long_message}.22050.s16le.{bin,txt}), the
 * in-process synthetic-burst tests (crates/sameold/src/receiver.rs:642-705) and
 * the per-block known answers of the reference's unit tests.
 * The reference itself is Rust and cannot be built in this image (no cargo/rustc,
 * dependencies not vendored), so there is no oracle/_ref; bit-level behaviour of
 * libm calls (cosf/sinf/expf/sinhf/hypotf) is that of glibc on x86-64, which is
 * what rustc's std binds to on x86_64-unknown-linux-gnu.
 *
 * All citations are file:line under /root/reference/crates/sameold/src/ unless
 * a crate is named.  "rx/" = receiver/.
 */
#ifndef SAME_ORACLE_H
#define SAME_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- configuration: mirrors SameReceiverBuilder (rx/builder.rs:22-67) ---- */
typedef struct so_config {
    uint32_t input_rate;
    float dc_blocker_len;            /* rx/builder.rs:53  default 0.38 */
    float agc_bandwidth;             /* :54  0.01 */
    float agc_gain_min, agc_gain_max;/* :55  [0, 1e6] */
    float timing_bw_unlocked;        /* :56  0.125 */
    float timing_bw_locked;          /* :57  0.05 */
    float timing_max_deviation;      /* :58  0.01 */
    float squelch_power_open;        /* :59  0.10 */
    float squelch_power_close;       /* :60  0.05 */
    float squelch_bandwidth;         /* :61  0.125 */
    uint32_t preamble_max_errors;    /* :62  2 */
    uint32_t eq_enabled;             /* :63  Some(default) */
    uint32_t eq_nff, eq_nfb;         /* rx/builder.rs:371-372  6, 4 */
    float eq_relaxation;             /* :373 0.05 */
    float eq_regularization;         /* :374 1e-6 */
    uint32_t frame_prefix_max_errors;/* :64  2 */
    uint32_t frame_max_invalid;      /* :65  5 */
} so_config;

void so_config_default(so_config *c, uint32_t input_rate);
/* setters with the reference's clamping (rx/builder.rs:95-279, 393-425) */
void so_config_with_dc_blocker_length(so_config *c, float len);
void so_config_with_agc_bandwidth(so_config *c, float bw);
void so_config_with_agc_gain_limits(so_config *c, float mn, float mx);
void so_config_with_timing_bandwidth(so_config *c, float unlocked, float locked);
void so_config_with_timing_max_deviation(so_config *c, float d);
void so_config_with_squelch_power(so_config *c, float open, float close);
void so_config_with_squelch_bandwidth(so_config *c, float bw);
void so_config_with_preamble_max_errors(so_config *c, uint32_t e);
void so_config_with_adaptive_equalizer(so_config *c, uint32_t nff, uint32_t nfb,
                                       float relaxation, float regularization);
void so_config_without_adaptive_equalizer(so_config *c);
void so_config_with_frame_prefix_max_errors(so_config *c, uint32_t e);
void so_config_with_frame_max_invalid(so_config *c, uint32_t n);
/* samedec's configuration: crates/samedec/src/main.rs:29-37 with cli.rs defaults */
void so_config_samedec(so_config *c, uint32_t input_rate);

/* ---- events: mirrors SameReceiverEvent (rx/output.rs:24-27, 231-261, 306-318) ---- */
enum {
    SO_LINK_NO_CARRIER = 0,
    SO_LINK_SEARCHING = 1,
    SO_LINK_READING = 2,
    SO_LINK_BURST = 3,
    SO_TRANSPORT_IDLE = 16,
    SO_TRANSPORT_ASSEMBLING = 17,
    SO_TRANSPORT_MSG_START = 18,  /* Message(Ok(StartOfMessage(hdr))) */
    SO_TRANSPORT_MSG_END = 19,    /* Message(Ok(EndOfMessage)) */
    SO_TRANSPORT_MSG_ERR = 20     /* Message(Err(e)); e in `aux` */
};
enum { SO_ERR_NOT_ASCII = 1, SO_ERR_UNRECOGNIZED_PREFIX = 2, SO_ERR_MALFORMED = 3 };

#define SO_EVENT_MAX_BYTES 288

typedef struct so_event {
    uint32_t kind;
    uint32_t len;            /* true byte length (may exceed SO_EVENT_MAX_BYTES) */
    uint64_t sample_counter; /* input_sample_counter at emission (receiver.rs:249-252) */
    uint64_t symbol_count;   /* squelch.symbol_count() at emission */
    uint32_t aux;            /* MSG_START: voting_byte_count; MSG_ERR: error code */
    uint32_t aux2;           /* MSG_START: parity_error_count */
    uint8_t bytes[SO_EVENT_MAX_BYTES];
} so_event;

typedef struct so_rx so_rx;

/* returns 0, or negative: -1 dc blocker length 0 (reference panics rx/dcblock.rs:74),
 * -2 agc min>max or NaN (f32::clamp panics), -3 allocation */
int so_rx_new(const so_config *cfg, so_rx **out);
void so_rx_free(so_rx *rx);
so_rx *so_rx_clone(const so_rx *rx);
void so_rx_reset(so_rx *rx);
uint32_t so_rx_input_rate(const so_rx *rx);
uint64_t so_rx_input_sample_counter(const so_rx *rx);
void so_rx_set_input_sample_counter(so_rx *rx, uint64_t v); /* receiver.rs:702 test hook */
int so_rx_force_eom_pending(const so_rx *rx, uint64_t *at);

/* SameReceiver::process (receiver.rs:233-274): returns 1 and fills *ev when an event
 * fires after consuming *consumed <= n samples; returns 0 when all n were consumed. */
int so_rx_process(so_rx *rx, const float *x, size_t n, size_t *consumed, so_event *ev);
/* same, input as int16 cast to f32 without scaling (crates/samedec/src/app.rs:112) */
int so_rx_process_i16(so_rx *rx, const int16_t *x, size_t n, size_t *consumed, so_event *ev);
/* convenience: run all of x, append events to ev[0..cap); returns number of events
 * produced (may exceed cap; extra ones are dropped). */
size_t so_rx_run(so_rx *rx, const float *x, size_t n, so_event *ev, size_t cap);
size_t so_rx_run_i16(so_rx *rx, const int16_t *x, size_t n, so_event *ev, size_t cap);
/* flush(): feed 4*input_rate zeros (receiver.rs:216-224), collecting events */
size_t so_rx_flush(so_rx *rx, so_event *ev, size_t cap);

/* soft-symbol trace: every SymbolEstimate (rx/symsync.rs:52-71) with the
 * input_sample_counter at which its TED fired. 4 floats + index per entry. */
typedef struct so_symbol_trace {
    uint64_t sample_counter;
    float zero, sym, err, samples_until_next_ted;
} so_symbol_trace;
void so_rx_set_trace(so_rx *rx, so_symbol_trace *buf, size_t cap);
size_t so_rx_trace_count(const so_rx *rx);

/* link-layer-only mode: skip the transport layer (no Assembler). Used to compare
 * against the GPU link path and as the CPU baseline for the hot path. */
void so_rx_set_link_only(so_rx *rx, int link_only);

/* threaded batch runner for the CPU baseline: C channels, time-major x[t*C + c],
 * link-only, returns total number of link events; nthreads >= 1 */
size_t so_batch_run_time_major(const so_config *cfg, const float *x, size_t n_channels,
                               size_t n_samples, int nthreads, so_event *ev, size_t cap);
/* x is [C][T] channel-major; workers optionally pinned to cpus[i]; the range is run `reps` times */
size_t so_batch_run_channel_major(const so_config *cfg, const float *x, size_t n_channels, size_t n_samples,
                                  int nthreads, const int *cpus, int reps, so_event *ev, size_t cap);

/* ---- derived constants, exposed for tests / cross-checks ---- */
typedef struct so_derived {
    float sps;               /* rx/waveform.rs:29-31 */
    uint32_t ntaps;          /* rx/waveform.rs:40 */
    uint32_t dc_len;         /* receiver.rs:509 */
    float agc_bw;            /* receiver.rs:511 + rx/agc.rs:51 */
    float agc_gain0;         /* rx/agc.rs:55 */
    float samples_per_ted;   /* rx/symsync.rs:146 */
    float period_min, period_max; /* rx/symsync.rs:150-151 */
    float alpha_unlocked, beta_unlocked, alpha_locked, beta_locked;
    float power_open, power_close, power_bw;
} so_derived;
void so_derive(const so_config *cfg, so_derived *d);
/* taps: out[4*i + {0,1,2,3}] = mark.re, mark.im, space.re, space.im */
void so_matched_filter_taps(uint32_t fs, float *out, uint32_t *ntaps);
void so_cisoid_matched_filter(uint32_t points, float freq_fs, float *re, float *im);
void so_compute_loop_alphabeta(float bw, float *alpha, float *beta);

/* ---- sub-blocks, exposed so tests can replay the reference's unit-test vectors ---- */
typedef struct so_dcblock so_dcblock;
so_dcblock *so_dcblock_new(uint32_t len);
void so_dcblock_free(so_dcblock *d);
float so_dcblock_filter(so_dcblock *d, float x);
/* moving average alone (rx/dcblock.rs:104-108): returns average, *delayed = window front */
typedef struct so_movavg so_movavg;
so_movavg *so_movavg_new(uint32_t len);
void so_movavg_free(so_movavg *m);
float so_movavg_filter(so_movavg *m, float x, float *delayed);

typedef struct so_agc { float bandwidth, min_gain, max_gain, gain; int locked; } so_agc;
void so_agc_init(so_agc *a, float bw, float mn, float mx);
void so_agc_reset(so_agc *a);
float so_agc_input(so_agc *a, float x);

/* multiply_accumulate (rx/filter.rs:363-377), real history x real coeff */
float so_mac_ff(const float *hist, size_t nh, const float *coeff, size_t nc);

typedef struct so_demod so_demod;
so_demod *so_demod_new(uint32_t fs);
void so_demod_free(so_demod *d);
void so_demod_push(so_demod *d, float x);
float so_demod_demod(const so_demod *d);
uint32_t so_demod_ntaps(const so_demod *d);

typedef struct so_ted { float h[3]; uint32_t counter; } so_ted;
void so_ted_reset(so_ted *t);
/* returns 1 when a SymbolEstimate is produced */
int so_ted_input(so_ted *t, float sample, float *zero, float *sym, float *err);
float so_zero_crossing_metric(const float v[3]);

typedef struct so_timing {
    float samples_per_ted, period_min, period_max, alpha, beta, period_avg, period_inst;
    so_ted ted;
} so_timing;
void so_timing_init(so_timing *t, float sps, float bw, float max_dev);
void so_timing_reset(so_timing *t);
void so_timing_set_bw(so_timing *t, float bw);
/* advance_loop (rx/symsync.rs:219-244); have_sym=0 -> None */
float so_timing_advance(so_timing *t, float offset, int have_sym, float sym_err);
/* input (rx/symsync.rs:198-201) */
float so_timing_input(so_timing *t, float sample, float offset, int *have_sym,
                      float *zero, float *sym, float *err);

enum { SO_SQ_NO_CARRIER = 0, SO_SQ_DROPPED = 1, SO_SQ_READING = 2, SO_SQ_READY = 3 };
typedef struct so_squelch {
    uint32_t max_errors; float power_open, power_close;
    uint32_t sync_to, data;           /* CodeCorrelator */
    float pt_bw, pt_power;            /* PowerTracker */
    float hist[64]; uint32_t hist_len, hist_head;   /* ArrayDeque<f32,64,Wrapping> */
    uint8_t phist[32]; uint32_t phist_len, phist_head;
    uint64_t symbol_counter;
    int sample_clock;                 /* -1 = None */
    int sync_lock;
} so_squelch;
void so_squelch_init(so_squelch *s, uint32_t sync_to, uint32_t max_err, float open,
                     float close, float bw);
void so_squelch_reset(so_squelch *s);
void so_squelch_end(so_squelch *s);
void so_squelch_lock(so_squelch *s, int lock);
/* returns SO_SQ_*; on READY fills out[16], *resync, *symbol_counter, *power */
int so_squelch_input(so_squelch *s, const float in[2], int *resync, float out[16],
                     uint64_t *symbol_counter, float *power);
uint32_t so_squelch_correlator_data(const so_squelch *s);
int so_squelch_is_sync(const so_squelch *s);
uint32_t so_code_search(uint32_t *data, uint32_t sync_to, float sym);
float so_power_track(float *power, float bw, float sym);

typedef struct so_equalizer so_equalizer;
so_equalizer *so_equalizer_new(uint32_t nff, uint32_t nfb, float relax, float reg,
                               int have_train, uint32_t train_to);
void so_equalizer_free(so_equalizer *e);
void so_equalizer_reset(so_equalizer *e);
void so_equalizer_enable(so_equalizer *e, int enable);
int so_equalizer_train(so_equalizer *e);   /* 0 ok, -1 no training sequence */
int so_equalizer_mode(const so_equalizer *e, uint32_t *word, uint32_t *count);
int so_equalizer_estimate_symbol(so_equalizer *e, const float in[2], float *err);
uint8_t so_equalizer_input(so_equalizer *e, const float in[16], float *err);
void so_nlms_update(float relax, float reg, float error, const float *window,
                    size_t n, float *coeff);

typedef struct so_framer so_framer;
so_framer *so_framer_new(uint32_t max_prefix_errors, uint32_t max_invalid);
void so_framer_free(so_framer *f);
void so_framer_reset(so_framer *f);
/* return SO_LINK_x; for BURST, burst and len point at framer-owned storage valid
 * until the next call */
int so_framer_input(so_framer *f, uint8_t data, uint64_t symbol_count, int restart,
                    const uint8_t **burst, size_t *len);
int so_framer_end(so_framer *f, const uint8_t **burst, size_t *len);
int so_framer_state(const so_framer *f);
uint32_t so_message_prefix_errors(uint32_t word);
int so_is_allowed_byte(uint8_t c);

/* transport layer pieces (rx/combiner.rs, rx/assembler.rs, sameplace message.rs:813-828) */
void so_bit_vote_detect(uint8_t b0, uint8_t b1, uint8_t *out, uint32_t *errs);
void so_bit_vote_correct(uint8_t b0, uint8_t b1, uint8_t b2, uint8_t *out, uint32_t *errs);
/* check_header: returns 0 and fills offsets, or SO_ERR_MALFORMED */
int so_check_header(const uint8_t *hdr, size_t n, size_t *offset_time, size_t *hdr_len);

/* estimate_message (rx/combiner.rs:154-203) over up to 3 bursts; outputs sized 268;
 * returns the estimated length */
uint32_t so_estimate_message(const uint8_t *const *bursts, const size_t *lens, uint32_t n,
                             uint8_t *bytes, uint8_t *nbursts, uint8_t *errs);
/* combine (rx/combiner.rs:32-80): 0 = None, 1 = Some(result in *ev) */
int so_combine(const uint8_t *const *bursts, const size_t *lens, uint32_t n, so_event *ev);

typedef struct so_assembler so_assembler;
so_assembler *so_assembler_new(void);
void so_assembler_free(so_assembler *a);
void so_assembler_reset(so_assembler *a);
/* both fill *ev (kind = SO_TRANSPORT_*) */
void so_assembler_assemble(so_assembler *a, const uint8_t *burst, size_t n,
                           uint64_t symbol_count, so_event *ev);
void so_assembler_idle(so_assembler *a, uint64_t symbol_count, so_event *ev);
uint64_t so_max_interburst_symbols(void);
uint64_t so_max_history_duration(void);

/* test-only AFSK modulator of the reference (rx/waveform.rs:73-104, 137-155):
 * bytes -> symbols (LSb first) -> continuous-phase AFSK, even integer samples/symbol.
 * Writes n_bytes*8*symlen floats to out (caller sizes via so_modulate_len). */
size_t so_modulate_len(size_t n_bytes, uint32_t fs, uint32_t *symlen);
void so_modulate_afsk_bytes(const uint8_t *bytes, size_t n_bytes, uint32_t fs, float *out);

#ifdef __cplusplus
}
#endif
#endif
