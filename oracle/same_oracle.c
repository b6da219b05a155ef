/*
 * same_oracle.c -- CPU restatement of sameold 0.6.0's receive chain (see same_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the parity oracle and the "port" CPU baseline.
 *
 * Build with: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math (see Makefile).
 * Every f32 operation is written as its own statement-level expression in the order
 * the Rust source evaluates it; rustc never contracts a*b+c into an FMA and has no
 * excess precision on x86-64, and neither does this file under the flags above.
 *
 * Citations are file:line under /root/reference/crates/sameold/src/ ("rx/" = receiver/).
 */
#define _GNU_SOURCE
#include "same_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Rust f32 helper semantics                                                   */
/* ------------------------------------------------------------------------- */

/* f32::clamp: `if x < min {min} else if x > max {max} else {x}` -- NaN passes through */
static inline float rs_clamp(float x, float mn, float mx)
{
    if (x < mn) x = mn;
    if (x > mx) x = mx;
    return x;
}
/* f32::min / f32::max: NaN-ignoring (IEEE minNum/maxNum) == fminf/fmaxf */
static inline float rs_min(float a, float b) { return fminf(a, b); }
static inline float rs_max(float a, float b) { return fmaxf(a, b); }
/* f32::signum: +1 for +0.0 and positives, -1 for -0.0 and negatives, NaN for NaN */
static inline float rs_signum(float x)
{
    if (x != x) return x;
    return copysignf(1.0f, x);
}
/* `f as usize`: truncate toward zero, saturating, NaN -> 0 */
static inline size_t rs_as_usize(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return (size_t)-1;
    return (size_t)f;
}
static inline uint32_t popcount32(uint32_t v) { return (uint32_t)__builtin_popcount(v); }

/* ------------------------------------------------------------------------- */
/* Builder (rx/builder.rs)                                                     */
/* ------------------------------------------------------------------------- */

void so_config_default(so_config *c, uint32_t input_rate)
{
    /* rx/builder.rs:50-67 and 369-376 */
    memset(c, 0, sizeof(*c));
    c->input_rate = input_rate;
    c->dc_blocker_len = 0.38f;
    c->agc_bandwidth = 0.01f;
    c->agc_gain_min = 0.0f;
    c->agc_gain_max = 1.0e6f;
    c->timing_bw_unlocked = 0.125f;
    c->timing_bw_locked = 0.05f;
    c->timing_max_deviation = 0.01f;
    c->squelch_power_open = 0.10f;
    c->squelch_power_close = 0.05f;
    c->squelch_bandwidth = 0.125f;
    c->preamble_max_errors = 2;
    c->eq_enabled = 1;
    c->eq_nff = 6;
    c->eq_nfb = 4;
    c->eq_relaxation = 0.05f;
    c->eq_regularization = 1.0e-6f;
    c->frame_prefix_max_errors = 2;
    c->frame_max_invalid = 5;
}
void so_config_with_dc_blocker_length(so_config *c, float len)
{ c->dc_blocker_len = rs_max(0.0f, len); }                      /* rx/builder.rs:95-98 */
void so_config_with_agc_bandwidth(so_config *c, float bw)
{ c->agc_bandwidth = rs_clamp(bw, 0.0f, 1.0f); }                /* :107-110 */
void so_config_with_agc_gain_limits(so_config *c, float mn, float mx)
{ c->agc_gain_min = mn; c->agc_gain_max = mx; }                 /* :120-123 (unvalidated) */
void so_config_with_timing_bandwidth(so_config *c, float unlocked, float locked)
{                                                               /* :139-143 */
    c->timing_bw_unlocked = rs_clamp(unlocked, 0.0f, 1.0f);
    c->timing_bw_locked = rs_clamp(locked, 0.0f, c->timing_bw_unlocked);
}
void so_config_with_timing_max_deviation(so_config *c, float d)
{ c->timing_max_deviation = rs_clamp(d, 0.0f, 0.5f); }          /* :155-158 */
void so_config_with_squelch_power(so_config *c, float open, float close)
{                                                               /* :172-176 */
    c->squelch_power_open = rs_clamp(open, 0.0f, 1.0f);
    c->squelch_power_close = rs_min(close, open);
}
void so_config_with_squelch_bandwidth(so_config *c, float bw)
{ c->squelch_bandwidth = bw; }                                  /* :187-190 */
void so_config_with_preamble_max_errors(so_config *c, uint32_t e)
{ c->preamble_max_errors = e; }                                 /* :204-207 */
void so_config_with_adaptive_equalizer(so_config *c, uint32_t nff, uint32_t nfb,
                                       float relaxation, float regularization)
{                                                               /* :222-225, 393-425 */
    c->eq_enabled = 1;
    c->eq_nff = nff > 1 ? nff : 1;
    c->eq_nfb = nfb < 1 ? 1 : (nfb > c->eq_nff ? c->eq_nff : nfb);
    c->eq_relaxation = rs_clamp(relaxation, 0.0f, 1.0f);
    c->eq_regularization = rs_clamp(regularization, 0.0f, 3.40282347e+38f);
}
void so_config_without_adaptive_equalizer(so_config *c) { c->eq_enabled = 0; } /* :231-234 */
void so_config_with_frame_prefix_max_errors(so_config *c, uint32_t e)
{ c->frame_prefix_max_errors = e > 7 ? 7 : e; }                 /* :247-250 */
void so_config_with_frame_max_invalid(so_config *c, uint32_t n)
{ c->frame_max_invalid = n; }                                   /* :276-279 */

void so_config_samedec(so_config *c, uint32_t input_rate)
{
    /* crates/samedec/src/main.rs:29-37 with the clap defaults of cli.rs:92-139 */
    so_config_default(c, input_rate);
    so_config_with_agc_gain_limits(c, 1.0f / 32767.0f, 1.0f / 200.0f);
    so_config_with_agc_bandwidth(c, 0.01f);
    so_config_with_dc_blocker_length(c, 0.38f);
    so_config_with_timing_bandwidth(c, 0.125f, 0.05f);
    so_config_with_timing_max_deviation(c, 0.01f);
    so_config_with_squelch_power(c, 0.10f, 0.05f);
    so_config_with_preamble_max_errors(c, 2);
}

/* ------------------------------------------------------------------------- */
/* Window (rx/filter.rs:218-323): fixed-length ring, zero initialised          */
/* ------------------------------------------------------------------------- */

typedef struct so_window {
    float *v;       /* ring storage */
    uint32_t len;
    uint32_t head;  /* index of the oldest element */
} so_window;

static int win_init(so_window *w, uint32_t len)
{
    w->v = (float *)calloc(len ? len : 1, sizeof(float));
    w->len = len;
    w->head = 0;
    return w->v ? 0 : -1;
}
static void win_free(so_window *w) { free(w->v); w->v = NULL; }
static void win_reset(so_window *w)
{
    for (uint32_t i = 0; i < w->len; ++i) w->v[i] = 0.0f;       /* rx/filter.rs:243-247 */
}
/* push_scalar (rx/filter.rs:284-288): returns the aged-off oldest sample */
static inline float win_push_scalar(so_window *w, float x)
{
    if (w->len == 0) return 0.0f; /* pop_front on empty -> unwrap_or(0); then push grows: not reachable here */
    float out = w->v[w->head];
    w->v[w->head] = x;
    w->head = (w->head + 1 == w->len) ? 0 : w->head + 1;
    return out;
}
/* push of a slice (rx/filter.rs:258-275): keep the right-most `len` of the input */
static void win_push(so_window *w, const float *in, size_t n)
{
    if (n > w->len) { in += n - w->len; n = w->len; }
    for (size_t i = 0; i < n; ++i) (void)win_push_scalar(w, in[i]);
}
static inline float win_front(const so_window *w) { return w->v[w->head]; }      /* oldest */
/* element i counted from the newest (0 = most recent) */
static inline float win_from_newest(const so_window *w, uint32_t i)
{
    uint32_t idx = w->head + w->len - 1 - i;
    if (idx >= w->len) idx -= w->len;
    return w->v[idx];
}
/* element i counted from the oldest */
static inline float win_from_oldest(const so_window *w, uint32_t i)
{
    uint32_t idx = w->head + i;
    if (idx >= w->len) idx -= w->len;
    return w->v[idx];
}

/* multiply_accumulate (rx/filter.rs:363-377): newest history sample times coeff[0]
 * first; `out += hi * co` is a rounded multiply followed by a rounded add. */
float so_mac_ff(const float *hist, size_t nh, const float *coeff, size_t nc)
{
    float out = 0.0f;
    size_t n = nh < nc ? nh : nc;
    for (size_t i = 0; i < n; ++i) {
        float p = hist[nh - 1 - i] * coeff[i];
        out += p;
    }
    return out;
}
static float win_mac(const so_window *w, const float *coeff, uint32_t nc)
{
    float out = 0.0f;
    uint32_t n = w->len < nc ? w->len : nc;
    for (uint32_t i = 0; i < n; ++i) {
        float p = win_from_newest(w, i) * coeff[i];
        out += p;
    }
    return out;
}

/* ------------------------------------------------------------------------- */
/* DC blocker (rx/dcblock.rs)                                                  */
/* ------------------------------------------------------------------------- */

struct so_movavg { so_window window; float inv_len, moving_sum; };

static int movavg_init(struct so_movavg *m, uint32_t len)
{
    /* rx/dcblock.rs:73-80 */
    m->inv_len = 1.0f / (float)len;
    m->moving_sum = 0.0f;
    return win_init(&m->window, len);
}
static inline float movavg_filter(struct so_movavg *m, float input, float *delayed)
{
    /* rx/dcblock.rs:104-108 */
    float aged = win_push_scalar(&m->window, input);
    float d = input - aged;
    m->moving_sum += d;
    *delayed = win_front(&m->window);
    return m->moving_sum * m->inv_len;
}
so_movavg *so_movavg_new(uint32_t len)
{
    if (len == 0) return NULL;
    so_movavg *m = (so_movavg *)calloc(1, sizeof(*m));
    if (m && movavg_init(m, len)) { free(m); m = NULL; }
    return m;
}
void so_movavg_free(so_movavg *m) { if (m) { win_free(&m->window); free(m); } }
float so_movavg_filter(so_movavg *m, float x, float *delayed) { return movavg_filter(m, x, delayed); }

struct so_dcblock { struct so_movavg ff, fb; };

static inline float dcblock_filter(struct so_dcblock *d, float input)
{
    /* rx/dcblock.rs:45-49 */
    float sig, unused;
    float ma0 = movavg_filter(&d->ff, input, &sig);
    float ma1 = movavg_filter(&d->fb, ma0, &unused);
    float k = (d->ff.window.len > 1) ? 1.0f : 0.0f;
    float t = k * ma1;
    return sig - t;
}
static void dcblock_reset(struct so_dcblock *d)
{
    win_reset(&d->ff.window); d->ff.moving_sum = 0.0f;           /* rx/dcblock.rs:85-88 */
    win_reset(&d->fb.window); d->fb.moving_sum = 0.0f;
}
so_dcblock *so_dcblock_new(uint32_t len)
{
    if (len == 0) return NULL;                                   /* assert!(len > 0) :74 */
    so_dcblock *d = (so_dcblock *)calloc(1, sizeof(*d));
    if (!d) return NULL;
    if (movavg_init(&d->ff, len) || movavg_init(&d->fb, len)) { so_dcblock_free(d); return NULL; }
    return d;
}
void so_dcblock_free(so_dcblock *d)
{
    if (!d) return;
    win_free(&d->ff.window); win_free(&d->fb.window); free(d);
}
float so_dcblock_filter(so_dcblock *d, float x) { return dcblock_filter(d, x); }

/* ------------------------------------------------------------------------- */
/* AGC (rx/agc.rs)                                                             */
/* ------------------------------------------------------------------------- */

void so_agc_init(so_agc *a, float bw, float mn, float mx)
{
    /* rx/agc.rs:49-57 */
    a->bandwidth = rs_clamp(bw, 0.0f, 1.0f);
    a->min_gain = mn;
    a->max_gain = mx;
    a->locked = 0;
    a->gain = rs_min(1.0f, mn);
}
void so_agc_reset(so_agc *a) { a->gain = 1.0f; a->locked = 0; }  /* rx/agc.rs:60-63 */
float so_agc_input(so_agc *a, float input)
{
    /* rx/agc.rs:72-77 */
    float out = input * a->gain;
    float k = a->locked ? 0.0f : 1.0f;
    float e = 1.0f - fabsf(out);
    float ke = k * e;
    float upd = ke * a->bandwidth;
    a->gain += upd;
    a->gain = rs_clamp(a->gain, a->min_gain, a->max_gain);
    return out;
}

/* ------------------------------------------------------------------------- */
/* Waveform + demod (rx/waveform.rs, rx/demod.rs)                              */
/* ------------------------------------------------------------------------- */

static const float FSK_MARK_HZ = 2083.3f;     /* rx/waveform.rs:6 */
static const float FSK_SPACE_HZ = 1562.5f;    /* :9 */
static const float BAUD_HZ = 520.83f;         /* :12 */
#define PREAMBLE 0xabu                        /* :19 */
#define PREAMBLE_SYNC_WORD 0xababababu        /* :26 */
static const float PI_F32 = 3.14159274101257324f; /* std::f32::consts::PI */

static float samples_per_symbol(uint32_t fs) { return (float)fs / BAUD_HZ; } /* :29-31 */

void so_cisoid_matched_filter(uint32_t points, float freq_fs, float *re, float *im)
{
    /* rx/waveform.rs:54-64.  o = Complex(0, 2*PI*freq*(N-1-i)); o = 2*conj(exp(o))/N.
     * num-complex 0.4.6 exp(): from_polar(re.exp(), im) = (r*cos, r*sin) with r = exp(0) = 1. */
    for (uint32_t it = 0; it < points; ++it) {
        float twopi = 2.0f * PI_F32;
        float a = twopi * freq_fs;
        float theta = a * (float)(points - 1 - it);
        float r = expf(0.0f);
        float ere = r * cosf(theta);
        float eim = r * sinf(theta);
        float cre = ere, cim = -eim;               /* conj */
        float sre = 2.0f * cre, sim = 2.0f * cim;  /* f32 * Complex */
        re[it] = sre / (float)points;              /* Complex / f32 */
        im[it] = sim / (float)points;
    }
}

void so_matched_filter_taps(uint32_t fs, float *out, uint32_t *ntaps_out)
{
    /* rx/waveform.rs:39-44 */
    uint32_t ntaps = (uint32_t)rs_as_usize(floorf(samples_per_symbol(fs)));
    if (ntaps_out) *ntaps_out = ntaps;
    if (!out) return;
    float *mre = (float *)malloc(sizeof(float) * 4 * (ntaps ? ntaps : 1));
    float *mim = mre + ntaps, *sre = mim + ntaps, *sim = sre + ntaps;
    so_cisoid_matched_filter(ntaps, FSK_MARK_HZ / (float)fs, mre, mim);
    so_cisoid_matched_filter(ntaps, FSK_SPACE_HZ / (float)fs, sre, sim);
    for (uint32_t i = 0; i < ntaps; ++i) {
        out[4 * i + 0] = mre[i]; out[4 * i + 1] = mim[i];
        out[4 * i + 2] = sre[i]; out[4 * i + 3] = sim[i];
    }
    free(mre);
}

struct so_demod { so_window window; uint32_t ntaps; float *taps; /* [4*ntaps] */ };

static int demod_init(struct so_demod *d, uint32_t fs)
{
    so_matched_filter_taps(fs, NULL, &d->ntaps);
    d->taps = (float *)malloc(sizeof(float) * 4 * (d->ntaps ? d->ntaps : 1));
    if (!d->taps) return -1;
    so_matched_filter_taps(fs, d->taps, NULL);
    return win_init(&d->window, d->ntaps);
}
static void demod_free_inner(struct so_demod *d) { free(d->taps); win_free(&d->window); }
static inline float demod_now(const struct so_demod *d)
{
    /* rx/demod.rs:156-164 + rx/filter.rs:363-377 with In=f32, Coeff=Complex<f32>:
     * f32 * Complex = (x*re, x*im); Complex += is componentwise. */
    float mre = 0.0f, mim = 0.0f, sre = 0.0f, sim = 0.0f;
    const so_window *w = &d->window;
    for (uint32_t i = 0; i < d->ntaps; ++i) {
        float x = win_from_newest(w, i);
        const float *h = d->taps + 4 * i;
        float p0 = x * h[0], p1 = x * h[1], p2 = x * h[2], p3 = x * h[3];
        mre += p0; mim += p1; sre += p2; sim += p3;
    }
    /* Complex::norm() = re.hypot(im) -> libm hypotf */
    float nm = hypotf(mre, mim);
    float ns = hypotf(sre, sim);
    float diff = nm - ns;
    return rs_clamp(diff, -1.0f, 1.0f);
}
so_demod *so_demod_new(uint32_t fs)
{
    so_demod *d = (so_demod *)calloc(1, sizeof(*d));
    if (d && demod_init(d, fs)) { free(d); d = NULL; }
    return d;
}
void so_demod_free(so_demod *d) { if (d) { demod_free_inner(d); free(d); } }
void so_demod_push(so_demod *d, float x) { (void)win_push_scalar(&d->window, x); }
float so_demod_demod(const so_demod *d) { return demod_now(d); }
uint32_t so_demod_ntaps(const so_demod *d) { return d->ntaps; }

/* ------------------------------------------------------------------------- */
/* Symbol timing (rx/symsync.rs)                                               */
/* ------------------------------------------------------------------------- */

void so_compute_loop_alphabeta(float bw, float *alpha, float *beta)
{
    /* rx/symsync.rs:329-337 */
    float omega = (2.0f * PI_F32) * bw;
    float k0 = 2.0f;
    float k1 = expf(-omega);
    float sh = sinhf(omega);
    *alpha = (k0 * k1) * sh;
    float t = sh + 1.0f;
    float u = k1 * t;
    float v = 1.0f - u;
    *beta = k0 * v;
}

float so_zero_crossing_metric(const float v[3])
{
    /* rx/symsync.rs:311-322 */
    float d = rs_signum(v[0]) - rs_signum(v[2]);
    return v[1] * d;
}
void so_ted_reset(so_ted *t)
{
    t->h[0] = t->h[1] = t->h[2] = 0.0f;                          /* rx/symsync.rs:265-271 */
    t->counter = 0;
}
int so_ted_input(so_ted *t, float sample, float *zero, float *sym, float *err)
{
    /* rx/symsync.rs:278-287 (ArrayDeque<_,3,Wrapping>::push_back drops the front) */
    t->h[0] = t->h[1]; t->h[1] = t->h[2]; t->h[2] = sample;
    t->counter = (t->counter + 1) % 2;
    if (t->counter == 1) {
        *err = so_zero_crossing_metric(t->h);
        *zero = t->h[1];
        *sym = t->h[2];
        return 1;
    }
    return 0;
}
void so_timing_init(so_timing *t, float sps, float bw, float max_dev)
{
    /* rx/symsync.rs:142-163 */
    so_compute_loop_alphabeta(bw, &t->alpha, &t->beta);
    t->samples_per_ted = sps / 2.0f;
    float dev = sps * rs_clamp(max_dev, 0.0f, 0.5f);
    t->period_avg = t->samples_per_ted;
    t->period_inst = t->samples_per_ted;
    t->period_min = t->period_avg - dev;
    t->period_max = t->period_avg + dev;
    so_ted_reset(&t->ted);
}
void so_timing_reset(so_timing *t)
{
    so_ted_reset(&t->ted);                                       /* rx/symsync.rs:166-170 */
    t->period_avg = t->samples_per_ted;
    t->period_inst = t->samples_per_ted;
}
void so_timing_set_bw(so_timing *t, float bw)
{ so_compute_loop_alphabeta(bw, &t->alpha, &t->beta); }          /* :176-180 */

float so_timing_advance(so_timing *t, float offset, int have_sym, float sym_err)
{
    /* rx/symsync.rs:219-244 */
    offset = rs_clamp(offset, -0.5f, 0.5f);
    if (have_sym) {
        float q = offset / t->samples_per_ted;
        float e0 = sym_err - q;
        float err = rs_clamp(e0, -1.0f, 1.0f);
        float bi = t->beta * err;
        t->period_avg += bi;
        t->period_avg = rs_clamp(t->period_avg, t->period_min, t->period_max);
        float ai = t->alpha * err;
        float s = t->period_avg + ai;
        t->period_inst = s + offset;
        if (t->period_inst < 0.0f) t->period_inst = t->period_avg;
    } else {
        t->period_inst += offset;
    }
    return t->period_inst;
}
float so_timing_input(so_timing *t, float sample, float offset, int *have_sym,
                      float *zero, float *sym, float *err)
{
    /* rx/symsync.rs:198-201 */
    *have_sym = so_ted_input(&t->ted, sample, zero, sym, err);
    return so_timing_advance(t, offset, *have_sym, *have_sym ? *err : 0.0f);
}

/* ------------------------------------------------------------------------- */
/* Code + power squelch (rx/codesquelch.rs)                                    */
/* ------------------------------------------------------------------------- */

uint32_t so_code_search(uint32_t *data, uint32_t sync_to, float sym)
{
    /* rx/codesquelch.rs:421-428, 441-445 */
    uint32_t bit = (sym >= 0.0f) ? 1u : 0u;
    *data = *data >> 1;
    *data |= bit << 31;
    return popcount32(sync_to ^ *data);
}
float so_power_track(float *power, float bw, float sym)
{
    /* rx/codesquelch.rs:483-488 */
    float pwr = sym * sym;
    float d = pwr - *power;
    float u = d * bw;
    *power += u;
    *power = rs_max(*power, 0.0f);
    return *power;
}
static void sq_hist_push(so_squelch *s, float v)
{
    /* ArrayDeque<f32,64,Wrapping>::push_back */
    if (s->hist_len < 64) {
        s->hist[(s->hist_head + s->hist_len) & 63] = v;
        s->hist_len++;
    } else {
        s->hist[s->hist_head] = v;
        s->hist_head = (s->hist_head + 1) & 63;
    }
}
static void sq_phist_push(so_squelch *s, uint8_t v)
{
    if (s->phist_len < 32) {
        s->phist[(s->phist_head + s->phist_len) & 31] = v;
        s->phist_len++;
    } else {
        s->phist[s->phist_head] = v;
        s->phist_head = (s->phist_head + 1) & 31;
    }
}
void so_squelch_end(so_squelch *s) { s->sync_lock = 0; s->sample_clock = -1; } /* :336-339 */
void so_squelch_reset(so_squelch *s)
{
    /* rx/codesquelch.rs:320-327 */
    so_squelch_end(s);
    s->data = 0;
    s->hist_len = 0; s->hist_head = 0;
    s->pt_power = 0.0f;
    s->phist_len = 0; s->phist_head = 0;
    s->symbol_counter = 0;
}
void so_squelch_init(so_squelch *s, uint32_t sync_to, uint32_t max_err, float open,
                     float close, float bw)
{
    /* rx/codesquelch.rs:190-212, 464-469 */
    memset(s, 0, sizeof(*s));
    s->max_errors = max_err;
    s->power_open = open;
    s->power_close = rs_min(close, open);
    s->sync_to = sync_to;
    s->pt_bw = rs_clamp(bw, 0.0f, 1.0f);
    so_squelch_reset(s);
}
void so_squelch_lock(so_squelch *s, int lock) { s->sync_lock = lock; }           /* :311 */
uint32_t so_squelch_correlator_data(const so_squelch *s) { return s->data; }
int so_squelch_is_sync(const so_squelch *s) { return s->sample_clock >= 0; }

int so_squelch_input(so_squelch *s, const float in[2], int *resync, float out[16],
                     uint64_t *symbol_counter, float *power)
{
    /* rx/codesquelch.rs:228-304 */
    sq_hist_push(s, in[0]);
    sq_hist_push(s, in[1]);
    uint32_t err = so_code_search(&s->data, s->sync_to, in[1]);
    float pwr = so_power_track(&s->pt_power, s->pt_bw, in[1]);
    sq_phist_push(s, (uint8_t)(pwr >= s->power_close));
    s->symbol_counter += 1;

    if (s->hist_len < 64) return SO_SQ_NO_CARRIER;

    int adjusted = 0;
    if (!s->sync_lock && err <= s->max_errors && pwr >= s->power_open) {
        if (s->sample_clock != 0) adjusted = 1;   /* None or Some(n != 0) */
        s->sample_clock = 0;
    } else if (s->sample_clock >= 0 && !s->phist[s->phist_head]) {
        /* power_history.front(): the oldest retained flag */
        so_squelch_end(s);
        return SO_SQ_DROPPED;
    }

    if (s->sample_clock < 0) return SO_SQ_NO_CARRIER;
    if (s->sample_clock == 0) {
        s->sample_clock = 1;
        for (uint32_t i = 0; i < 16; ++i) out[i] = s->hist[(s->hist_head + i) & 63];
        *symbol_counter = s->symbol_counter;
        *power = pwr;
        *resync = adjusted;
        return SO_SQ_READY;
    }
    s->sample_clock = (s->sample_clock + 1) % 8;
    return SO_SQ_READING;
}

/* ------------------------------------------------------------------------- */
/* Equalizer (rx/equalize.rs)                                                  */
/* ------------------------------------------------------------------------- */

enum { EQ_DISABLED = 0, EQ_FEEDBACK = 1, EQ_TRAINING = 2 };
struct so_equalizer {
    float relaxation, regularization;
    int have_train; uint32_t train_to;
    uint32_t nff, nfb;
    float *ff_coeff, *fb_coeff;
    so_window ff_wind, fb_wind;
    int mode; uint32_t train_word, train_count;
};

static void coeff_identity(float *c, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) c[i] = 0.0f;               /* rx/filter.rs:139-144 */
    c[0] = 1.0f;
}
so_equalizer *so_equalizer_new(uint32_t nff, uint32_t nfb, float relax, float reg,
                               int have_train, uint32_t train_to)
{
    /* rx/equalize.rs:127-153 */
    if (nff == 0 || nfb == 0) return NULL;  /* from_identity indexes [0]: panics */
    so_equalizer *e = (so_equalizer *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->relaxation = relax; e->regularization = reg;
    e->have_train = have_train; e->train_to = train_to;
    e->nff = nff; e->nfb = nfb;
    e->ff_coeff = (float *)malloc(sizeof(float) * nff);
    e->fb_coeff = (float *)malloc(sizeof(float) * nfb);
    if (!e->ff_coeff || !e->fb_coeff || win_init(&e->ff_wind, nff) || win_init(&e->fb_wind, nfb)) {
        so_equalizer_free(e); return NULL;
    }
    coeff_identity(e->ff_coeff, nff);
    coeff_identity(e->fb_coeff, nfb);
    e->mode = EQ_FEEDBACK;
    return e;
}
void so_equalizer_free(so_equalizer *e)
{
    if (!e) return;
    free(e->ff_coeff); free(e->fb_coeff); win_free(&e->ff_wind); win_free(&e->fb_wind); free(e);
}
static so_equalizer *equalizer_clone(const so_equalizer *s)
{
    so_equalizer *e = so_equalizer_new(s->nff, s->nfb, s->relaxation, s->regularization,
                                       s->have_train, s->train_to);
    if (!e) return NULL;
    memcpy(e->ff_coeff, s->ff_coeff, sizeof(float) * s->nff);
    memcpy(e->fb_coeff, s->fb_coeff, sizeof(float) * s->nfb);
    memcpy(e->ff_wind.v, s->ff_wind.v, sizeof(float) * s->nff); e->ff_wind.head = s->ff_wind.head;
    memcpy(e->fb_wind.v, s->fb_wind.v, sizeof(float) * s->nfb); e->fb_wind.head = s->fb_wind.head;
    e->mode = s->mode; e->train_word = s->train_word; e->train_count = s->train_count;
    return e;
}
void so_equalizer_reset(so_equalizer *e)
{
    /* rx/equalize.rs:191-196: mode is preserved */
    coeff_identity(e->ff_coeff, e->nff);
    coeff_identity(e->fb_coeff, e->nfb);
    win_reset(&e->ff_wind);
    win_reset(&e->fb_wind);
}
void so_equalizer_enable(so_equalizer *e, int enable)
{ e->mode = enable ? EQ_FEEDBACK : EQ_DISABLED; }               /* :203-208 */
int so_equalizer_train(so_equalizer *e)
{
    if (!e->have_train) return -1;                               /* :216-220 */
    e->mode = EQ_TRAINING; e->train_word = e->train_to; e->train_count = 0;
    return 0;
}
int so_equalizer_mode(const so_equalizer *e, uint32_t *word, uint32_t *count)
{
    if (word) *word = e->train_word;
    if (count) *count = e->train_count;
    return e->mode;
}

static void nlms_update_win(float relax, float reg, float error, const so_window *w,
                            float *coeff, uint32_t ncoeff)
{
    /* rx/equalize.rs:354-386.  gain = relax / (reg + sum_{oldest..newest} w^2);
     * coeff[i] += gain * error * w[newest - i]  == ((gain*error)*data) */
    float sumsq = 0.0f;
    for (uint32_t i = 0; i < w->len; ++i) {
        float v = win_from_oldest(w, i);
        float sq = v * v;
        sumsq += sq;
    }
    float den = reg + sumsq;
    float gain = relax / den;
    uint32_t n = ncoeff < w->len ? ncoeff : w->len;
    for (uint32_t i = 0; i < n; ++i) {
        float ge = gain * error;
        float p = ge * win_from_newest(w, i);
        coeff[i] += p;
    }
}
void so_nlms_update(float relax, float reg, float error, const float *window, size_t n,
                    float *coeff)
{
    so_window w; w.v = (float *)window; w.len = (uint32_t)n; w.head = 0;
    nlms_update_win(relax, reg, error, &w, coeff, (uint32_t)n);
}
static void eq_evolve(so_equalizer *e, float error)
{
    /* rx/equalize.rs:315-332 */
    nlms_update_win(e->relaxation, e->regularization, error, &e->ff_wind, e->ff_coeff, e->nff);
    nlms_update_win(e->relaxation, e->regularization, -error, &e->fb_wind, e->fb_coeff, e->nfb);
}
int so_equalizer_estimate_symbol(so_equalizer *e, const float in[2], float *err_out)
{
    /* rx/equalize.rs:249-308 */
    win_push(&e->ff_wind, in, 2);
    float ff = win_mac(&e->ff_wind, e->ff_coeff, e->nff);
    float fb = win_mac(&e->fb_wind, e->fb_coeff, e->nfb);
    float sym_val = ff - fb;
    float sym_est, err;
    switch (e->mode) {
    case EQ_DISABLED:
        sym_est = rs_signum(sym_val); err = 0.0f;
        break;
    case EQ_FEEDBACK:
        sym_est = rs_signum(sym_val);
        err = sym_est - sym_val;
        eq_evolve(e, err);
        break;
    default: { /* EQ_TRAINING */
        uint32_t sa = e->train_word, count = e->train_count;
        float b = (float)(sa & 1u);
        float tb = 2.0f * b;
        sym_est = tb - 1.0f;
        sa >>= 1;
        err = sym_est - sym_val;
        eq_evolve(e, err);
        count += 1;
        if (count >= 32) e->mode = EQ_FEEDBACK;
        else { e->train_word = sa; e->train_count = count; }
        break; }
    }
    float fbin[2] = { sym_est, 0.0f };
    win_push(&e->fb_wind, fbin, 2);
    *err_out = err;
    return sym_est >= 0.0f;
}
uint8_t so_equalizer_input(so_equalizer *e, const float in[16], float *err)
{
    /* rx/equalize.rs:173-186 */
    uint8_t byte = 0; float last = 0.0f;
    for (int b = 0; b < 8; ++b) {
        int bit = so_equalizer_estimate_symbol(e, in + 2 * b, &last);
        byte |= (uint8_t)(bit << b);
    }
    *err = last;
    return byte;
}

/* ------------------------------------------------------------------------- */
/* Framer (rx/framing.rs) and allowed bytes (rx/combiner.rs:105-137)           */
/* ------------------------------------------------------------------------- */

int so_is_allowed_byte(uint8_t c)
{
    return c == '-' || (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') ||
           (c >= 'a' && c <= 'z') || c == '/' || c == '?' || c == '(' || c == ')' ||
           c == '[' || c == ']' || c == '.' || c == '_' || c == ',' || c == '+' || c == ' ';
}
uint32_t so_message_prefix_errors(uint32_t inp)
{
    /* rx/framing.rs:235-243 */
    const uint32_t start = 0x5a435a43u; /* "ZCZC" big-endian */
    const uint32_t end = 0x4e4e4e4eu;   /* "NNNN" */
    uint32_t a = popcount32(inp ^ start), b = popcount32(inp ^ end);
    return a < b ? a : b;
}

enum { FR_IDLE = 0, FR_SEARCH = 1, FR_READ = 2 };
struct so_framer {
    int state;
    uint32_t word, count;          /* PrefixSearch(u32,u32) */
    uint8_t *msg; size_t len, cap; /* DataRead(Vec<u8>, u32) */
    uint32_t invalid;
    uint8_t *out; size_t out_len, out_cap; /* last emitted burst */
    uint32_t max_prefix_errors, max_invalid;
};
#define PREFIX_SEARCH_LEN 21u      /* rx/framing.rs:201 */

so_framer *so_framer_new(uint32_t max_prefix_errors, uint32_t max_invalid)
{
    so_framer *f = (so_framer *)calloc(1, sizeof(*f));
    if (!f) return NULL;
    f->max_prefix_errors = max_prefix_errors; f->max_invalid = max_invalid;
    f->cap = 512; f->msg = (uint8_t *)malloc(f->cap);
    f->out_cap = 512; f->out = (uint8_t *)malloc(f->out_cap);
    if (!f->msg || !f->out) { so_framer_free(f); return NULL; }
    return f;
}
void so_framer_free(so_framer *f) { if (f) { free(f->msg); free(f->out); free(f); } }
void so_framer_reset(so_framer *f) { f->state = FR_IDLE; }       /* rx/framing.rs:83-86 */
int so_framer_state(const so_framer *f)
{
    /* rx/framing.rs:191-197 */
    return f->state == FR_IDLE ? SO_LINK_NO_CARRIER
         : f->state == FR_SEARCH ? SO_LINK_SEARCHING : SO_LINK_READING;
}
int so_framer_end(so_framer *f, const uint8_t **burst, size_t *len)
{
    /* rx/framing.rs:174-186 */
    if (f->state == FR_READ) {
        if (f->out_cap < f->len) {
            f->out_cap = f->len * 2; f->out = (uint8_t *)realloc(f->out, f->out_cap);
        }
        memcpy(f->out, f->msg, f->len); f->out_len = f->len;
        f->state = FR_IDLE;
        if (burst) *burst = f->out;
        if (len) *len = f->out_len;
        return SO_LINK_BURST;
    }
    f->state = FR_IDLE;
    return SO_LINK_NO_CARRIER;
}
static void framer_msg_push(so_framer *f, uint8_t b)
{
    if (f->len == f->cap) { f->cap *= 2; f->msg = (uint8_t *)realloc(f->msg, f->cap); }
    f->msg[f->len++] = b;
}
int so_framer_input(so_framer *f, uint8_t data, uint64_t symbol_count, int restart,
                    const uint8_t **burst, size_t *len)
{
    /* rx/framing.rs:109-164 */
    if (restart) {
        const uint8_t *b0 = NULL; size_t l0 = 0;
        int out = so_framer_end(f, &b0, &l0);
        f->state = FR_SEARCH; f->word = 0; f->count = 0;
        (void)so_framer_input(f, data, symbol_count, 0, NULL, NULL);
        if (out == SO_LINK_BURST) {
            if (burst) *burst = b0;
            if (len) *len = l0;
            return out;
        }
        return SO_LINK_SEARCHING;
    }
    switch (f->state) {
    case FR_IDLE:
        return SO_LINK_NO_CARRIER;
    case FR_SEARCH:
        f->word = (f->word << 8) | (uint32_t)data;
        f->count += 1;
        if (so_message_prefix_errors(f->word) <= f->max_prefix_errors) {
            f->len = 0;
            framer_msg_push(f, (uint8_t)(f->word >> 24));
            framer_msg_push(f, (uint8_t)(f->word >> 16));
            framer_msg_push(f, (uint8_t)(f->word >> 8));
            framer_msg_push(f, (uint8_t)(f->word));
            f->invalid = 0;
            f->state = FR_READ;
        } else if (f->count > PREFIX_SEARCH_LEN) {
            f->state = FR_IDLE;
        }
        return so_framer_state(f);
    default: /* FR_READ */
        f->invalid += so_is_allowed_byte(data) ? 0u : 1u;
        if (f->invalid > f->max_invalid) return so_framer_end(f, burst, len);
        framer_msg_push(f, data);
        return so_framer_state(f);
    }
}

/* ------------------------------------------------------------------------- */
/* Transport layer: combiner, header check, assembler                          */
/* ------------------------------------------------------------------------- */

#define MAX_MESSAGE_LENGTH 268u                                   /* rx/assembler.rs:70 */
uint64_t so_max_interburst_symbols(void)
{
    /* rx/assembler.rs:85: ((1.05 * BAUD_HZ) + 17.0 * 8.0) as u64, f32 arithmetic */
    float a = 1.05f * BAUD_HZ;
    float b = 17.0f * 8.0f;
    float c = a + b;
    return (uint64_t)rs_as_usize(c);
}
uint64_t so_max_history_duration(void)
{
    return 2 * (so_max_interburst_symbols() + 8 * (uint64_t)MAX_MESSAGE_LENGTH); /* :92-93 */
}

void so_bit_vote_detect(uint8_t b0, uint8_t b1, uint8_t *out, uint32_t *errs)
{
    /* rx/combiner.rs:216-222 */
    uint8_t x = b0 ^ b1;
    *out = (uint8_t)(b0 & (uint8_t)~(uint8_t)(0xff * (x != 0)));
    *errs = popcount32(x);
}
void so_bit_vote_correct(uint8_t b0, uint8_t b1, uint8_t b2, uint8_t *out, uint32_t *errs)
{
    /* rx/combiner.rs:234-249 */
    uint8_t p0 = (uint8_t)~(b0 ^ b1), p1 = (uint8_t)~(b1 ^ b2), p2 = (uint8_t)~(b0 ^ b2);
    *out = (uint8_t)((b0 & p0) | (b2 & p1) | (b2 & p2));
    *errs = 8u - popcount32((uint8_t)(p0 & p1 & p2));
}

static int is_alpha(uint8_t c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
static int is_digit(uint8_t c) { return c >= '0' && c <= '9'; }

int so_check_header(const uint8_t *h, size_t n, size_t *offset_time, size_t *hdr_len)
{
    /* crates/sameplace/src/message.rs:813-828:
     * ^ZCZC-[[:alpha:]]{3}-[[:alpha:]]{3}(-[0-9]{6})+(\+[0-9]{4}-[0-9]{7}-.{3,8}-)
     * leftmost-first: the location group is greedy and group 2 must start with '+',
     * so no backtracking over the repetition count can succeed; .{3,8} is greedy. */
    size_t p = 0;
    if (n < 5 || memcmp(h, "ZCZC-", 5) != 0) return SO_ERR_MALFORMED;
    p = 5;
    for (int g = 0; g < 2; ++g) {
        for (int k = 0; k < 3; ++k, ++p) if (p >= n || !is_alpha(h[p])) return SO_ERR_MALFORMED;
        if (g == 0) { if (p >= n || h[p] != '-') return SO_ERR_MALFORMED; ++p; }
    }
    size_t nloc = 0;
    for (;;) {
        if (p + 7 > n || h[p] != '-') break;
        int ok = 1;
        for (int k = 1; k <= 6; ++k) if (!is_digit(h[p + k])) { ok = 0; break; }
        if (!ok) break;
        p += 7; ++nloc;
    }
    if (nloc == 0) return SO_ERR_MALFORMED;
    size_t g2 = p;
    if (p >= n || h[p] != '+') return SO_ERR_MALFORMED;
    ++p;
    for (int k = 0; k < 4; ++k, ++p) if (p >= n || !is_digit(h[p])) return SO_ERR_MALFORMED;
    if (p >= n || h[p] != '-') return SO_ERR_MALFORMED;
    ++p;
    for (int k = 0; k < 7; ++k, ++p) if (p >= n || !is_digit(h[p])) return SO_ERR_MALFORMED;
    if (p >= n || h[p] != '-') return SO_ERR_MALFORMED;
    ++p;
    for (int m = 8; m >= 3; --m) {
        if (p + (size_t)m >= n) continue;
        int ok = 1;
        for (int k = 0; k < m; ++k) if (h[p + k] == '\n') { ok = 0; break; }
        if (ok && h[p + m] == '-') {
            *offset_time = g2;
            *hdr_len = p + (size_t)m + 1;
            return 0;
        }
    }
    return SO_ERR_MALFORMED;
}

/* a parsed MessageResult */
typedef struct so_msg {
    int kind;                 /* SO_TRANSPORT_MSG_START / _END / _ERR */
    uint32_t err;             /* for _ERR */
    uint8_t text[MAX_MESSAGE_LENGTH]; uint32_t len;
    uint32_t offset_time, parity_errors, voting_bytes;
} so_msg;

/* Message::try_from((bytes, errs, counts)) crates/sameplace/src/message.rs:718-736, 184-230 */
static void message_try_from(const uint8_t *b, size_t n, const uint8_t *errs,
                             const uint8_t *counts, so_msg *out)
{
    memset(out, 0, sizeof(*out));
    for (size_t i = 0; i < n; ++i) if (b[i] & 0x80) { out->kind = SO_TRANSPORT_MSG_ERR; out->err = SO_ERR_NOT_ASCII; return; }
    if (n >= 5 && memcmp(b, "ZCZC-", 5) == 0) {
        size_t off, hl;
        if (so_check_header(b, n, &off, &hl)) { out->kind = SO_TRANSPORT_MSG_ERR; out->err = SO_ERR_MALFORMED; return; }
        out->kind = SO_TRANSPORT_MSG_START;
        out->len = (uint32_t)hl; memcpy(out->text, b, hl);
        out->offset_time = (uint32_t)off;
        for (size_t i = 0; i < hl && i < n; ++i) {
            out->parity_errors += errs[i];
            out->voting_bytes += counts[i] >= 3;
        }
    } else if (n >= 2 && b[0] == 'N' && b[1] == 'N') {
        out->kind = SO_TRANSPORT_MSG_END;
    } else {
        out->kind = SO_TRANSPORT_MSG_ERR; out->err = SO_ERR_UNRECOGNIZED_PREFIX;
    }
}

typedef struct so_burst { uint8_t b[MAX_MESSAGE_LENGTH]; uint32_t len; uint64_t deadline; } so_burst;

/* estimate_message rx/combiner.rs:154-203 */
static uint32_t estimate_message(const so_burst *hist, uint32_t nb, uint8_t *bytes,
                                 uint8_t *nbursts, uint8_t *errs)
{
    uint32_t out = 0;
    if (nb > 3) nb = 3;
    uint32_t pos[3] = {0, 0, 0};
    while (out < MAX_MESSAGE_LENGTH) {
        uint8_t cur[3]; uint32_t n = 0; int msb = 0;
        for (uint32_t k = 0; k < nb; ++k)
            if (pos[k] < hist[k].len) cur[n++] = hist[k].b[pos[k]++];
        for (uint32_t k = 0; k < n; ++k) { msb |= (cur[k] & 0x80) != 0; cur[k] &= 0x7f; }
        uint8_t est; uint32_t be;
        if (n == 0) break;
        else if (n == 1) { est = cur[0]; be = 0; }
        else if (n == 2) so_bit_vote_detect(cur[0], cur[1], &est, &be);
        else so_bit_vote_correct(cur[0], cur[1], cur[2], &est, &be);
        if (!so_is_allowed_byte(est)) break;
        bytes[out] = est; nbursts[out] = (uint8_t)n; errs[out] = (uint8_t)(be + (uint32_t)msb);
        ++out;
    }
    return out;
}

/* combine rx/combiner.rs:32-80; returns 0 for None, 1 with *res filled */
static int combine(const so_burst *hist, uint32_t nb, so_msg *res)
{
    uint8_t msg[MAX_MESSAGE_LENGTH], cnt[MAX_MESSAGE_LENGTH], errs[MAX_MESSAGE_LENGTH];
    uint32_t n = estimate_message(hist, nb, msg, cnt, errs);
    if (n == 0) return 0;
    uint32_t good = 0;
    while (good < n && cnt[good] >= 2) ++good;                   /* :264-273 */
    message_try_from(msg, good, errs, cnt, res);
    if (res->kind != SO_TRANSPORT_MSG_ERR) return 1;
    if (n >= 2 && msg[0] == 'N' && msg[1] == 'N') {              /* Fast EOM :251-258 */
        memset(res, 0, sizeof(*res)); res->kind = SO_TRANSPORT_MSG_END; return 1;
    }
    if (good == 0) return 0;
    return 1;
}

static uint32_t load_bursts(const uint8_t *const *bursts, const size_t *lens, uint32_t n, so_burst *h)
{
    if (n > 3) n = 3;
    for (uint32_t i = 0; i < n; ++i) {
        h[i].len = (uint32_t)(lens[i] < MAX_MESSAGE_LENGTH ? lens[i] : MAX_MESSAGE_LENGTH);
        memcpy(h[i].b, bursts[i], h[i].len);
        h[i].deadline = 0;
    }
    return n;
}
uint32_t so_estimate_message(const uint8_t *const *bursts, const size_t *lens, uint32_t n,
                             uint8_t *bytes, uint8_t *nbursts, uint8_t *errs)
{
    so_burst h[3];
    n = load_bursts(bursts, lens, n, h);
    return estimate_message(h, n, bytes, nbursts, errs);
}
static void msg_to_event(const so_msg *m, so_event *ev);
int so_combine(const uint8_t *const *bursts, const size_t *lens, uint32_t n, so_event *ev)
{
    so_burst h[3]; so_msg res;
    n = load_bursts(bursts, lens, n, h);
    if (!combine(h, n, &res)) return 0;
    msg_to_event(&res, ev);
    return 1;
}

struct so_assembler {
    so_burst hist[4]; uint32_t nhist;          /* VecDeque<TimedData<Burst>> */
    int pending; so_msg pend_msg; uint64_t pend_deadline;
    int have_prev; so_msg prev; uint64_t prev_deadline;
};
so_assembler *so_assembler_new(void) { return (so_assembler *)calloc(1, sizeof(so_assembler)); }
void so_assembler_free(so_assembler *a) { free(a); }
void so_assembler_reset(so_assembler *a) { memset(a, 0, sizeof(*a)); } /* rx/assembler.rs:127-131 */

static void prune_history(so_assembler *a, uint64_t now)
{
    /* rx/assembler.rs:362-368 */
    uint32_t w = 0;
    for (uint32_t i = 0; i < a->nhist; ++i)
        if (!(a->hist[i].deadline <= now)) { if (w != i) a->hist[w] = a->hist[i]; ++w; }
    a->nhist = w;
    while (a->nhist > 2) {
        memmove(&a->hist[0], &a->hist[1], sizeof(so_burst) * (a->nhist - 1));
        a->nhist--;
    }
}
static void pending_accept(so_assembler *a, const so_msg *msg, uint64_t now)
{
    /* rx/assembler.rs:294-328 */
    uint64_t deadline = (msg->kind == SO_TRANSPORT_MSG_END) ? now : now + so_max_interburst_symbols();
    if (a->pending) {
        const so_msg *old = &a->pend_msg;
        int replace;
        if (old->kind == SO_TRANSPORT_MSG_ERR) replace = 1;
        else if (old->kind == SO_TRANSPORT_MSG_END && msg->kind == SO_TRANSPORT_MSG_START) replace = 1;
        else if (old->kind == SO_TRANSPORT_MSG_START && msg->kind == SO_TRANSPORT_MSG_START)
            replace = msg->voting_bytes >= old->voting_bytes;
        else replace = 0;
        if (replace) { a->pend_msg = *msg; a->pend_deadline = deadline; }
    } else {
        a->pending = 1; a->pend_msg = *msg; a->pend_deadline = deadline;
    }
}
static void msg_to_event(const so_msg *m, so_event *ev)
{
    ev->kind = (uint32_t)m->kind;
    ev->len = 0; ev->aux = 0; ev->aux2 = 0;
    if (m->kind == SO_TRANSPORT_MSG_START) {
        ev->len = m->len; memcpy(ev->bytes, m->text, m->len);
        ev->aux = m->voting_bytes; ev->aux2 = m->parity_errors;
    } else if (m->kind == SO_TRANSPORT_MSG_ERR) {
        ev->aux = m->err;
    }
}
void so_assembler_idle(so_assembler *a, uint64_t now, so_event *ev)
{
    /* rx/assembler.rs:205-234 */
    prune_history(a, now);
    if (a->pending && a->pend_deadline <= now) {                 /* poll :336-345 */
        so_msg m = a->pend_msg;
        a->pending = 0;
        if (m.kind != SO_TRANSPORT_MSG_ERR) {
            a->have_prev = 1; a->prev = m; a->prev_deadline = now + so_max_history_duration();
        }
        msg_to_event(&m, ev);
        return;
    }
    ev->kind = a->nhist == 0 ? SO_TRANSPORT_IDLE : SO_TRANSPORT_ASSEMBLING;
    ev->len = 0; ev->aux = 0; ev->aux2 = 0;
}
void so_assembler_assemble(so_assembler *a, const uint8_t *burst, size_t n, uint64_t now,
                           so_event *ev)
{
    /* rx/assembler.rs:154-184 */
    if (n == 0) { so_assembler_idle(a, now, ev); return; }
    prune_history(a, now);
    if (a->have_prev && a->prev_deadline <= now) a->have_prev = 0; /* prune_previous :371-376 */
    so_burst *nb = &a->hist[a->nhist++];
    nb->len = (uint32_t)(n < MAX_MESSAGE_LENGTH ? n : MAX_MESSAGE_LENGTH);
    memcpy(nb->b, burst, nb->len);
    nb->deadline = now + so_max_history_duration();

    so_msg res;
    if (combine(a->hist, a->nhist, &res)) {
        /* deduplicate :245-265 */
        int keep = 1;
        if (res.kind != SO_TRANSPORT_MSG_ERR && a->have_prev) {
            /* Message::as_str(): header text, or "NNNN" for EndOfMessage */
            const so_msg *p = &a->prev;
            const char *ps = p->kind == SO_TRANSPORT_MSG_END ? "NNNN" : (const char *)p->text;
            size_t pl = p->kind == SO_TRANSPORT_MSG_END ? 4 : p->len;
            const char *rs = res.kind == SO_TRANSPORT_MSG_END ? "NNNN" : (const char *)res.text;
            size_t rl = res.kind == SO_TRANSPORT_MSG_END ? 4 : res.len;
            if (pl == rl && memcmp(ps, rs, pl) == 0) keep = 0;
        }
        if (keep) pending_accept(a, &res, now);
    }
    so_assembler_idle(a, now, ev);
}

/* ------------------------------------------------------------------------- */
/* The receiver (receiver.rs)                                                  */
/* ------------------------------------------------------------------------- */

struct so_rx {
    so_config cfg;
    struct so_dcblock dc;
    so_agc agc;
    struct so_demod demod;
    so_timing symsync;
    so_squelch squelch;
    so_equalizer *equalizer;
    so_framer *framer;
    so_assembler *assembler;
    float timing_bw_unlocked, timing_bw_locked;
    uint32_t input_rate;
    uint64_t input_sample_counter;
    /* link_state / transport_state as last reported (receiver.rs:84-85) */
    int link_kind; uint8_t *link_burst; size_t link_burst_len, link_burst_cap;
    so_event transport_state;
    so_event queue[2]; uint32_t qn;             /* event_queue */
    uint32_t ted_sample_clock;
    float samples_until_next_ted;
    int have_force_eom; uint64_t force_eom_at_sample;
    int link_only;
    so_symbol_trace *trace; size_t trace_cap, trace_n;
};

void so_derive(const so_config *cfg, so_derived *d)
{
    float sps = samples_per_symbol(cfg->input_rate);
    d->sps = sps;
    so_matched_filter_taps(cfg->input_rate, NULL, &d->ntaps);
    d->dc_len = (uint32_t)rs_as_usize(cfg->dc_blocker_len * sps);  /* receiver.rs:509 */
    float t = cfg->agc_bandwidth * sps;
    float bw = t / (float)cfg->input_rate;                         /* receiver.rs:511 */
    d->agc_bw = rs_clamp(bw, 0.0f, 1.0f);
    d->agc_gain0 = rs_min(1.0f, cfg->agc_gain_min);
    so_timing tl;
    so_timing_init(&tl, sps, cfg->timing_bw_unlocked, cfg->timing_max_deviation);
    d->samples_per_ted = tl.samples_per_ted;
    d->period_min = tl.period_min; d->period_max = tl.period_max;
    d->alpha_unlocked = tl.alpha; d->beta_unlocked = tl.beta;
    so_compute_loop_alphabeta(cfg->timing_bw_locked, &d->alpha_locked, &d->beta_locked);
    d->power_open = cfg->squelch_power_open;
    d->power_close = rs_min(cfg->squelch_power_close, cfg->squelch_power_open);
    d->power_bw = rs_clamp(cfg->squelch_bandwidth, 0.0f, 1.0f);
}

int so_rx_new(const so_config *cfg, so_rx **out)
{
    /* receiver.rs:502-560 */
    *out = NULL;
    so_derived d; so_derive(cfg, &d);
    if (d.dc_len == 0) return -1;
    if (!(cfg->agc_gain_min <= cfg->agc_gain_max)) return -2;
    so_rx *rx = (so_rx *)calloc(1, sizeof(*rx));
    if (!rx) return -3;
    rx->cfg = *cfg;
    if (movavg_init(&rx->dc.ff, d.dc_len) || movavg_init(&rx->dc.fb, d.dc_len)) goto fail;
    so_agc_init(&rx->agc, cfg->agc_bandwidth * d.sps / (float)cfg->input_rate,
                cfg->agc_gain_min, cfg->agc_gain_max);
    if (demod_init(&rx->demod, cfg->input_rate)) goto fail;
    so_timing_init(&rx->symsync, d.sps, cfg->timing_bw_unlocked, cfg->timing_max_deviation);
    so_squelch_init(&rx->squelch, PREAMBLE_SYNC_WORD, cfg->preamble_max_errors,
                    cfg->squelch_power_open, cfg->squelch_power_close, cfg->squelch_bandwidth);
    if (cfg->eq_enabled)
        rx->equalizer = so_equalizer_new(cfg->eq_nff, cfg->eq_nfb, cfg->eq_relaxation,
                                         cfg->eq_regularization, 1, PREAMBLE_SYNC_WORD);
    else /* disabled_equalizer() receiver.rs:585-590: order (1,1), relaxation 0, default reg */
        rx->equalizer = so_equalizer_new(1, 1, 0.0f, 1.0e-6f, 1, PREAMBLE_SYNC_WORD);
    rx->framer = so_framer_new(cfg->frame_prefix_max_errors, cfg->frame_max_invalid);
    rx->assembler = so_assembler_new();
    if (!rx->equalizer || !rx->framer || !rx->assembler) goto fail;
    rx->timing_bw_unlocked = cfg->timing_bw_unlocked;
    rx->timing_bw_locked = cfg->timing_bw_locked;
    rx->input_rate = cfg->input_rate;
    rx->link_kind = SO_LINK_NO_CARRIER;
    rx->transport_state.kind = SO_TRANSPORT_IDLE;
    rx->samples_until_next_ted = rx->symsync.samples_per_ted;
    *out = rx;
    return 0;
fail:
    so_rx_free(rx);
    return -3;
}
void so_rx_free(so_rx *rx)
{
    if (!rx) return;
    win_free(&rx->dc.ff.window); win_free(&rx->dc.fb.window);
    demod_free_inner(&rx->demod);
    so_equalizer_free(rx->equalizer); so_framer_free(rx->framer); so_assembler_free(rx->assembler);
    free(rx->link_burst);
    free(rx);
}
static int win_clone(so_window *dst, const so_window *src)
{
    if (win_init(dst, src->len)) return -1;
    memcpy(dst->v, src->v, sizeof(float) * src->len); dst->head = src->head;
    return 0;
}
so_rx *so_rx_clone(const so_rx *s)
{
    so_rx *rx = (so_rx *)malloc(sizeof(*rx));
    if (!rx) return NULL;
    *rx = *s;
    rx->trace = NULL; rx->trace_cap = rx->trace_n = 0;
    win_clone(&rx->dc.ff.window, &s->dc.ff.window);
    win_clone(&rx->dc.fb.window, &s->dc.fb.window);
    win_clone(&rx->demod.window, &s->demod.window);
    rx->demod.taps = (float *)malloc(sizeof(float) * 4 * (s->demod.ntaps ? s->demod.ntaps : 1));
    memcpy(rx->demod.taps, s->demod.taps, sizeof(float) * 4 * s->demod.ntaps);
    rx->equalizer = equalizer_clone(s->equalizer);
    rx->framer = so_framer_new(s->framer->max_prefix_errors, s->framer->max_invalid);
    rx->framer->state = s->framer->state; rx->framer->word = s->framer->word;
    rx->framer->count = s->framer->count; rx->framer->invalid = s->framer->invalid;
    for (size_t i = 0; i < s->framer->len; ++i) framer_msg_push(rx->framer, s->framer->msg[i]);
    rx->assembler = so_assembler_new(); *rx->assembler = *s->assembler;
    rx->link_burst = NULL; rx->link_burst_cap = 0;
    if (s->link_burst_len) {
        rx->link_burst = (uint8_t *)malloc(s->link_burst_len);
        memcpy(rx->link_burst, s->link_burst, s->link_burst_len); rx->link_burst_cap = s->link_burst_len;
    }
    return rx;
}
void so_rx_reset(so_rx *rx)
{
    /* receiver.rs:182-198 */
    dcblock_reset(&rx->dc);
    so_agc_reset(&rx->agc);
    win_reset(&rx->demod.window);
    so_timing_reset(&rx->symsync);
    so_squelch_reset(&rx->squelch);
    so_equalizer_reset(rx->equalizer);
    so_framer_reset(rx->framer);
    so_assembler_reset(rx->assembler);
    rx->input_sample_counter = 0;
    rx->link_kind = SO_LINK_NO_CARRIER; rx->link_burst_len = 0;
    memset(&rx->transport_state, 0, sizeof(rx->transport_state));
    rx->transport_state.kind = SO_TRANSPORT_IDLE;
    rx->qn = 0;
    rx->ted_sample_clock = 0;
    rx->samples_until_next_ted = rx->symsync.samples_per_ted;
    rx->have_force_eom = 0;
}
uint32_t so_rx_input_rate(const so_rx *rx) { return rx->input_rate; }
uint64_t so_rx_input_sample_counter(const so_rx *rx) { return rx->input_sample_counter; }
void so_rx_set_input_sample_counter(so_rx *rx, uint64_t v) { rx->input_sample_counter = v; }
int so_rx_force_eom_pending(const so_rx *rx, uint64_t *at)
{ if (at) *at = rx->force_eom_at_sample; return rx->have_force_eom; }
void so_rx_set_trace(so_rx *rx, so_symbol_trace *buf, size_t cap)
{ rx->trace = buf; rx->trace_cap = cap; rx->trace_n = 0; }
size_t so_rx_trace_count(const so_rx *rx) { return rx->trace_n; }
void so_rx_set_link_only(so_rx *rx, int link_only) { rx->link_only = link_only; }

/* receiver.rs:479-490 */
static void rx_end(so_rx *rx)
{
    rx->agc.locked = 0;
    so_squelch_end(&rx->squelch);
    so_equalizer_reset(rx->equalizer);
    so_timing_set_bw(&rx->symsync, rx->timing_bw_unlocked);
    so_timing_reset(&rx->symsync);
}

/* receiver.rs:407-474; returns SO_LINK_* and burst storage for BURST */
static int rx_symbol(so_rx *rx, const float data[2], const uint8_t **burst, size_t *blen)
{
    int resync = 0; float samples[16]; uint64_t symc = 0; float pwr = 0.0f;
    int st = so_squelch_input(&rx->squelch, data, &resync, samples, &symc, &pwr);
    switch (st) {
    case SO_SQ_NO_CARRIER:
        return so_framer_end(rx->framer, burst, blen);
    case SO_SQ_DROPPED:
        rx_end(rx);
        return so_framer_end(rx->framer, burst, blen);
    case SO_SQ_READING:
        return so_framer_state(rx->framer);
    default: break;
    }
    if (resync) {
        rx->agc.locked = 1;
        so_timing_set_bw(&rx->symsync, rx->timing_bw_locked);
        (void)so_equalizer_train(rx->equalizer);
    }
    float aerr;
    uint8_t byte = so_equalizer_input(rx->equalizer, samples, &aerr);
    int link = so_framer_input(rx->framer, byte, symc, resync, burst, blen);
    if (link == SO_LINK_READING) so_squelch_lock(&rx->squelch, 1);
    else if (link == SO_LINK_NO_CARRIER || link == SO_LINK_BURST) rx_end(rx);
    return link;
}

/* receiver.rs:343-395: returns -1 for None */
static inline int rx_high_rate(so_rx *rx, float input, const uint8_t **burst, size_t *blen)
{
    float sa = so_agc_input(&rx->agc, dcblock_filter(&rx->dc, input));
    (void)win_push_scalar(&rx->demod.window, sa);
    rx->ted_sample_clock += 1;
    rx->input_sample_counter += 1;
    float rem = rx->samples_until_next_ted - (float)rx->ted_sample_clock;
    if (rem <= 0.0f || fabsf(rem) < 0.5f) {
        rx->ted_sample_clock = 0;
        float sa_low = demod_now(&rx->demod);
        int have; float z, s, e;
        rx->samples_until_next_ted = so_timing_input(&rx->symsync, sa_low, rem, &have, &z, &s, &e);
        if (!have) return -1;
        if (rx->trace && rx->trace_n < rx->trace_cap) {
            so_symbol_trace *t = &rx->trace[rx->trace_n];
            t->sample_counter = rx->input_sample_counter;
            t->zero = z; t->sym = s; t->err = e;
            t->samples_until_next_ted = rx->samples_until_next_ted;
        }
        if (rx->trace) rx->trace_n++;
        float d[2] = { z, s };
        return rx_symbol(rx, d, burst, blen);
    }
    return -1;
}

static void fill_link_event(so_rx *rx, so_event *ev)
{
    memset(ev, 0, offsetof(so_event, bytes));
    ev->kind = (uint32_t)rx->link_kind;
    ev->sample_counter = rx->input_sample_counter;
    ev->symbol_count = rx->squelch.symbol_counter;
    if (rx->link_kind == SO_LINK_BURST) {
        ev->len = (uint32_t)rx->link_burst_len;
        size_t n = rx->link_burst_len < SO_EVENT_MAX_BYTES ? rx->link_burst_len : SO_EVENT_MAX_BYTES;
        memcpy(ev->bytes, rx->link_burst, n);
    }
}

static int transport_equal(const so_event *a, const so_event *b)
{
    if (a->kind != b->kind) return 0;
    if (a->kind == SO_TRANSPORT_MSG_START)
        return a->len == b->len && a->aux == b->aux && a->aux2 == b->aux2 &&
               memcmp(a->bytes, b->bytes, a->len) == 0;
    if (a->kind == SO_TRANSPORT_MSG_ERR) return a->aux == b->aux;
    return 1;
}

/* receiver.rs:291-333: returns 1 if a transport state was produced */
static int rx_transport(so_rx *rx, int link, const uint8_t *burst, size_t blen, so_event *ts)
{
    const uint64_t MAX_MESSAGE_DURATION_SECS = 135;              /* receiver.rs:496 */
    memset(ts, 0, offsetof(so_event, bytes));
    if (link == SO_LINK_BURST) {
        so_assembler_assemble(rx->assembler, burst, blen, rx->squelch.symbol_counter, ts);
    } else if (link == SO_LINK_NO_CARRIER && rx->have_force_eom &&
               rx->input_sample_counter > rx->force_eom_at_sample) {
        ts->kind = SO_TRANSPORT_MSG_END;
    } else if (link == SO_LINK_NO_CARRIER) {
        so_assembler_idle(rx->assembler, rx->squelch.symbol_counter, ts);
    } else {
        return 0;
    }
    if (ts->kind == SO_TRANSPORT_MSG_START) {
        rx->have_force_eom = 1;
        rx->force_eom_at_sample = rx->input_sample_counter +
                                  MAX_MESSAGE_DURATION_SECS * (uint64_t)rx->input_rate;
    } else if (ts->kind == SO_TRANSPORT_MSG_END) {
        rx->have_force_eom = 0;
    }
    return 1;
}

/* one sample through receiver.rs:243-270; returns 1 when an event was dequeued */
static inline int rx_step(so_rx *rx, float sample, so_event *ev)
{
    const uint8_t *burst = NULL; size_t blen = 0;
    int link = rx_high_rate(rx, sample, &burst, &blen);
    if (link < 0) return 0;
    int changed = link != rx->link_kind || link == SO_LINK_BURST;
    if (link == SO_LINK_BURST && rx->link_kind == SO_LINK_BURST)
        changed = blen != rx->link_burst_len || memcmp(burst, rx->link_burst, blen) != 0;
    if (changed) {
        rx->link_kind = link;
        rx->link_burst_len = 0;
        if (link == SO_LINK_BURST) {
            if (rx->link_burst_cap < blen) {
                rx->link_burst_cap = blen * 2 + 16;
                rx->link_burst = (uint8_t *)realloc(rx->link_burst, rx->link_burst_cap);
            }
            memcpy(rx->link_burst, burst, blen); rx->link_burst_len = blen;
        }
        fill_link_event(rx, &rx->queue[rx->qn++]);
    }
    if (!rx->link_only) {
        so_event ts;
        if (rx_transport(rx, link, burst, blen, &ts) && !transport_equal(&ts, &rx->transport_state)) {
            ts.sample_counter = rx->input_sample_counter;
            ts.symbol_count = rx->squelch.symbol_counter;
            rx->transport_state = ts;
            rx->queue[rx->qn++] = ts;
        }
    }
    if (rx->qn) {
        *ev = rx->queue[0];
        rx->queue[0] = rx->queue[1];
        rx->qn--;
        return 1;
    }
    return 0;
}

int so_rx_process(so_rx *rx, const float *x, size_t n, size_t *consumed, so_event *ev)
{
    /* receiver.rs:233-274 */
    *consumed = 0;
    if (rx->qn) { *ev = rx->queue[0]; rx->queue[0] = rx->queue[1]; rx->qn--; return 1; }
    for (size_t i = 0; i < n; ++i) {
        int got = rx_step(rx, x[i], ev);
        if (got) { *consumed = i + 1; return 1; }
    }
    *consumed = n;
    return 0;
}
int so_rx_process_i16(so_rx *rx, const int16_t *x, size_t n, size_t *consumed, so_event *ev)
{
    *consumed = 0;
    if (rx->qn) { *ev = rx->queue[0]; rx->queue[0] = rx->queue[1]; rx->qn--; return 1; }
    for (size_t i = 0; i < n; ++i) {
        int got = rx_step(rx, (float)x[i], ev);   /* `sa as f32`, crates/samedec/src/app.rs:112 */
        if (got) { *consumed = i + 1; return 1; }
    }
    *consumed = n;
    return 0;
}
size_t so_rx_run(so_rx *rx, const float *x, size_t n, so_event *ev, size_t cap)
{
    size_t count = 0, off = 0;
    so_event tmp;
    for (;;) {
        size_t used = 0;
        int got = so_rx_process(rx, x + off, n - off, &used, &tmp);
        off += used;
        if (!got) break;
        if (count < cap) ev[count] = tmp;
        ++count;
    }
    return count;
}
size_t so_rx_run_i16(so_rx *rx, const int16_t *x, size_t n, so_event *ev, size_t cap)
{
    size_t count = 0, off = 0;
    so_event tmp;
    for (;;) {
        size_t used = 0;
        int got = so_rx_process_i16(rx, x + off, n - off, &used, &tmp);
        off += used;
        if (!got) break;
        if (count < cap) ev[count] = tmp;
        ++count;
    }
    return count;
}
size_t so_rx_flush(so_rx *rx, so_event *ev, size_t cap)
{
    /* receiver.rs:216-224 feeds input_rate*4 zeros.  The reference stops at the first
     * Message; this helper keeps going so a caller can see every event, and the caller
     * picks the first message (tests do). */
    size_t n = (size_t)rx->input_rate * 4;
    float *z = (float *)calloc(n ? n : 1, sizeof(float));
    size_t c = so_rx_run(rx, z, n, ev, cap);
    free(z);
    return c;
}

/* ------------------------------------------------------------------------- */
/* Threaded batch runner (CPU baseline)                                        */
/* ------------------------------------------------------------------------- */

typedef struct batch_job {
    const so_config *cfg; const float *x; size_t C, T, c0, c1;
    so_event *ev; size_t cap; size_t *count; pthread_mutex_t *mu;
} batch_job;

static void *batch_worker(void *p)
{
    batch_job *j = (batch_job *)p;
    float *buf = (float *)malloc(sizeof(float) * (j->T ? j->T : 1));
    for (size_t c = j->c0; c < j->c1; ++c) {
        so_rx *rx = NULL;
        if (so_rx_new(j->cfg, &rx)) continue;
        so_rx_set_link_only(rx, 1);
        for (size_t t = 0; t < j->T; ++t) buf[t] = j->x[t * j->C + c];
        size_t off = 0; so_event tmp;
        for (;;) {
            size_t used = 0;
            int got = so_rx_process(rx, buf + off, j->T - off, &used, &tmp);
            off += used;
            if (!got) break;
            tmp.aux = (uint32_t)c;  /* channel index for link events */
            pthread_mutex_lock(j->mu);
            if (*j->count < j->cap) j->ev[*j->count] = tmp;
            (*j->count)++;
            pthread_mutex_unlock(j->mu);
        }
        so_rx_free(rx);
    }
    free(buf);
    return NULL;
}
size_t so_batch_run_time_major(const so_config *cfg, const float *x, size_t C, size_t T,
                               int nthreads, so_event *ev, size_t cap)
{
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > C) nthreads = (int)(C ? C : 1);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    batch_job *jobs = (batch_job *)malloc(sizeof(batch_job) * (size_t)nthreads);
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    size_t count = 0;
    for (int i = 0; i < nthreads; ++i) {
        jobs[i] = (batch_job){ cfg, x, C, T, C * (size_t)i / (size_t)nthreads,
                               C * (size_t)(i + 1) / (size_t)nthreads, ev, cap, &count, &mu };
        pthread_create(&th[i], NULL, batch_worker, &jobs[i]);
    }
    for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    free(th); free(jobs);
    return count;
}

/* Channel-major variant for an honest CPU baseline (bench.py): x is [C][T], every channel a
 * contiguous stream (what a host application feeding one SameReceiver per channel has), each
 * worker owns a contiguous channel range, optionally pinned to cpus[i] (one logical CPU per
 * physical core, chosen by the caller), and runs its range `reps` times with fresh receivers so
 * that the timed region is long enough to measure arithmetic rather than thread start-up.
 * Events of the first repetition are returned (aux = channel). */
typedef struct cm_job {
    const so_config *cfg; const float *x; size_t C, T, c0, c1; int reps; int cpu;
    so_event *ev; size_t cap; size_t *count; pthread_mutex_t *mu;
} cm_job;

static void *cm_worker(void *p)
{
    cm_job *j = (cm_job *)p;
    if (j->cpu >= 0) {
        cpu_set_t set; CPU_ZERO(&set); CPU_SET(j->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    for (int r = 0; r < j->reps; ++r) {
        for (size_t c = j->c0; c < j->c1; ++c) {
            so_rx *rx = NULL;
            if (so_rx_new(j->cfg, &rx)) continue;
            so_rx_set_link_only(rx, 1);
            const float *xs = j->x + c * j->T;
            size_t off = 0; so_event tmp;
            for (;;) {
                size_t used = 0;
                int got = so_rx_process(rx, xs + off, j->T - off, &used, &tmp);
                off += used;
                if (!got) break;
                if (r != 0) continue;
                tmp.aux = (uint32_t)c;
                pthread_mutex_lock(j->mu);
                if (*j->count < j->cap) j->ev[*j->count] = tmp;
                (*j->count)++;
                pthread_mutex_unlock(j->mu);
            }
            so_rx_free(rx);
        }
    }
    return NULL;
}
size_t so_batch_run_channel_major(const so_config *cfg, const float *x, size_t C, size_t T, int nthreads,
                                  const int *cpus, int reps, so_event *ev, size_t cap)
{
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > C) nthreads = (int)(C ? C : 1);
    if (reps < 1) reps = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    cm_job *jobs = (cm_job *)malloc(sizeof(cm_job) * (size_t)nthreads);
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    size_t count = 0;
    for (int i = 0; i < nthreads; ++i) {
        jobs[i] = (cm_job){ cfg, x, C, T, C * (size_t)i / (size_t)nthreads, C * (size_t)(i + 1) / (size_t)nthreads,
                            reps, cpus ? cpus[i] : -1, ev, cap, &count, &mu };
        pthread_create(&th[i], NULL, cm_worker, &jobs[i]);
    }
    for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    free(th); free(jobs);
    return count;
}

/* ------------------------------------------------------------------------- */
/* Test-only AFSK modulator of the reference (rx/waveform.rs:73-104, 137-155)   */
/* ------------------------------------------------------------------------- */

size_t so_modulate_len(size_t n_bytes, uint32_t fs, uint32_t *symlen_out)
{
    uint32_t symlen = (uint32_t)rs_as_usize(floorf(samples_per_symbol(fs)));
    if (symlen % 2 != 0) symlen += 1;
    if (symlen_out) *symlen_out = symlen;
    return n_bytes * 8 * (size_t)symlen;
}
void so_modulate_afsk_bytes(const uint8_t *bytes, size_t n_bytes, uint32_t fs, float *out)
{
    const float TWOPI = 2.0f * PI_F32;
    float mark = (TWOPI * FSK_MARK_HZ) / (float)fs;
    float space = (TWOPI * FSK_SPACE_HZ) / (float)fs;
    uint32_t symlen; size_t n = so_modulate_len(n_bytes, fs, &symlen);
    float phase = 0.0f;
    for (size_t it = 0; it < n; ++it) {
        size_t symi = it / symlen;
        int bit = (bytes[symi / 8] >> (symi % 8)) & 1;  /* bytes_to_samples(.., 1): LSb first */
        if (bit) phase += mark; else phase += space;
        if (phase > TWOPI) phase = -TWOPI + phase;
        out[it] = cosf(phase);
    }
}
