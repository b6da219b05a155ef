/*
 * same_rx.h -- C ABI of the MI355X batched SAME/EAS AFSK demodulator.
 *
 * This is the drop-in boundary for sameold's hot path: the per-sample receive chain
 * inside `sameold::SameReceiver` (DC block -> AGC -> mark/space matched filters ->
 * symbol-timing loop -> preamble code+power squelch -> DFE -> framer), run for
 * thousands of independent audio channels at once by hand-written gfx950 HIP kernels.
 *
 * The reference is pure Rust and has no FFI of its own; the
 * entry points below are what a `sameold-gpu` Rust shim would bind with `extern "C"`
 * (see INTEGRATION.md).  Each one names the reference item it replaces; citations are
 * file:line under crates/sameold/src/ ("rx/" = receiver/).
 *
 * Conventions: opaque handles; caller-owned input/output buffers; the library never
 * retains caller pointers past a call; return 0 on success or a negative SAME_E* code
 * where the reference would panic or where the GPU is unusable.  A handle is
 * single-writer; distinct handles are independent (as `&mut self` makes them in Rust).
 * No CPU fallback exists: without a usable gfx950 device every compute entry point
 * fails with SAME_ENODEVICE.
 */
#ifndef SAME_RX_H
#define SAME_RX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAME_RX_ABI_VERSION 2   /* 2: hip_stream NULL is the legacy default stream (SAME_STREAM_OWN selects the library's); SAME_BATCH_RELAXED; SAME_EKERNEL (an added error value) */

/* ------------------------------------------------------------------ errors */
enum {
    SAME_OK = 0,
    SAME_EINVAL = -1,        /* null/invalid argument */
    SAME_EDCLEN = -2,        /* DC blocker length rounds to 0: reference panics, rx/dcblock.rs:74 via receiver.rs:509 */
    SAME_EAGCLIMITS = -3,    /* agc min > max or NaN: f32::clamp panics, rx/agc.rs:75 */
    SAME_EEQORDER = -4,      /* equalizer order beyond the device limit (SAME_MAX_EQ_TAPS) */
    SAME_ENODEVICE = -5,     /* no usable HIP device / kernel image for this GPU */
    SAME_EHIP = -6,          /* a HIP runtime call failed (same_last_error() has text) */
    SAME_EOVERFLOW = -7,     /* event or burst pool overflow: results truncated */
    SAME_ENOMEM = -8,
    SAME_ERATE = -9,         /* input rate too low for the matched filters (ntaps < 1) */
    SAME_EKERNEL = -10       /* a kernel's wavefronts lost step with each other (an internal hand-over timed out): results of the call are void */
};
const char *same_last_error(void);
uint32_t same_rx_abi_version(void);
/* "SAME_SOURCE_HASH=<sha256>" of the sources the library was built from (sameold_amd/build.py
 * rebuilds when it differs from the tree's) */
const char *same_rx_source_hash(void);

/* ------------------------------------------------------------------ builder
 * Mirrors `SameReceiverBuilder` (rx/builder.rs:22-37): a plain copyable value with the
 * reference's defaults (:50-67) and setter-side clamping (:95-279).  EqualizerBuilder
 * (:359-425) is folded in as the four `eq_*` parameters. */
typedef struct same_rx_builder same_rx_builder;

same_rx_builder *same_rx_builder_new(uint32_t input_rate);            /* SameReceiverBuilder::new :50 */
same_rx_builder *same_rx_builder_default(void);                       /* Default :352-356 (22050 Hz) */
same_rx_builder *same_rx_builder_clone(const same_rx_builder *b);     /* #[derive(Clone, Copy)] :22 */
void same_rx_builder_free(same_rx_builder *b);

void same_rx_builder_with_dc_blocker_length(same_rx_builder *b, float len);                 /* :95-98 */
void same_rx_builder_with_agc_bandwidth(same_rx_builder *b, float bw);                      /* :107-110 */
void same_rx_builder_with_agc_gain_limits(same_rx_builder *b, float min, float max);        /* :120-123 */
void same_rx_builder_with_timing_bandwidth(same_rx_builder *b, float unlocked, float locked); /* :139-143 */
void same_rx_builder_with_timing_max_deviation(same_rx_builder *b, float max_dev);          /* :155-158 */
void same_rx_builder_with_squelch_power(same_rx_builder *b, float open, float close);       /* :172-176 */
void same_rx_builder_with_squelch_bandwidth(same_rx_builder *b, float bw);                  /* :187-190 */
void same_rx_builder_with_preamble_max_errors(same_rx_builder *b, uint32_t max_err);        /* :204-207 */
/* with_adaptive_equalizer(&EqualizerBuilder): with_filter_order/relaxation/regularization
 * clamps of :393-425 are applied here */
void same_rx_builder_with_adaptive_equalizer(same_rx_builder *b, uint32_t nfeedforward,
                                             uint32_t nfeedback, float relaxation,
                                             float regularization);                         /* :222-225 */
void same_rx_builder_without_adaptive_equalizer(same_rx_builder *b);                        /* :231-234 */
void same_rx_builder_with_frame_prefix_max_errors(same_rx_builder *b, uint32_t max_err);    /* :247-250 */
void same_rx_builder_with_frame_max_invalid(same_rx_builder *b, uint32_t max_invalid);      /* :276-279 */

uint32_t same_rx_builder_input_rate(const same_rx_builder *b);                              /* :282-285 */
float same_rx_builder_dc_blocker_length(const same_rx_builder *b);                          /* :288-291 */
float same_rx_builder_agc_bandwidth(const same_rx_builder *b);                              /* :294-297 */
void same_rx_builder_agc_gain_limits(const same_rx_builder *b, float out[2]);               /* :300-303 */
void same_rx_builder_timing_bandwidth(const same_rx_builder *b, float *unlocked, float *locked); /* :306-309 */
float same_rx_builder_timing_max_deviation(const same_rx_builder *b);                       /* :312-315 */
void same_rx_builder_squelch_power(const same_rx_builder *b, float *open, float *close);    /* :318-321 */
float same_rx_builder_squelch_bandwidth(const same_rx_builder *b);                          /* :324-327 */
uint32_t same_rx_builder_preamble_max_errors(const same_rx_builder *b);                     /* :330-333 */
/* adaptive_equalizer(): returns 0 for None, 1 for Some and fills the four outputs */
int same_rx_builder_adaptive_equalizer(const same_rx_builder *b, uint32_t *nfeedforward,
                                       uint32_t *nfeedback, float *relaxation,
                                       float *regularization);                              /* :336-339 */
uint32_t same_rx_builder_frame_prefix_max_errors(const same_rx_builder *b);                 /* :342-345 */
uint32_t same_rx_builder_frame_max_invalid(const same_rx_builder *b);                       /* :348-351 */

#define SAME_MAX_EQ_TAPS 16

/* ------------------------------------------------------------------ events
 * Mirrors `SameReceiverEvent` / `SameEventType` / `LinkState` / `TransportState`
 * (rx/output.rs:24-27, 166-180, 231-261, 306-318).  Burst bytes are copied into the
 * caller's event; nothing is borrowed. */
enum {
    SAME_LINK_NO_CARRIER = 0,      /* LinkState::NoCarrier */
    SAME_LINK_SEARCHING = 1,       /* LinkState::Searching */
    SAME_LINK_READING = 2,         /* LinkState::Reading */
    SAME_LINK_BURST = 3,           /* LinkState::Burst(Vec<u8>) */
    SAME_TRANSPORT_IDLE = 16,      /* TransportState::Idle */
    SAME_TRANSPORT_ASSEMBLING = 17,/* TransportState::Assembling */
    SAME_TRANSPORT_MSG_START = 18, /* Message(Ok(StartOfMessage(hdr))): bytes = header text */
    SAME_TRANSPORT_MSG_END = 19,   /* Message(Ok(EndOfMessage)) */
    SAME_TRANSPORT_MSG_ERR = 20    /* Message(Err(e)): aux = 1 NotAscii, 2 UnrecognizedPrefix, 3 Malformed */
};

#define SAME_EVENT_MAX_BYTES 288   /* >= MAX_MESSAGE_LENGTH 268 (rx/assembler.rs:70) */

typedef struct same_rx_event {
    uint32_t kind;             /* SAME_LINK_* / SAME_TRANSPORT_* */
    uint32_t channel;          /* channel index within the batch */
    uint64_t sample_counter;   /* input_sample_counter() at emission, rx/output.rs:121-123 */
    uint64_t symbol_count;     /* squelch symbol counter at emission (transport time base) */
    uint32_t len;              /* true burst/header length; bytes[] holds min(len, MAX) */
    uint32_t aux;              /* MSG_START: voting_byte_count; MSG_ERR: error code */
    uint32_t aux2;             /* MSG_START: parity_error_count */
    uint32_t reserved;
    uint8_t bytes[SAME_EVENT_MAX_BYTES];
} same_rx_event;

/* ------------------------------------------------------------------ batched receiver
 * `n_channels` independent `SameReceiver`s built from one builder
 * (`SameReceiverBuilder::build`, rx/builder.rs:73-76 / receiver.rs:502-560), resident on
 * one GPU.  All channels advance in lockstep: one call processes `n_samples` samples of
 * every channel through the whole link layer (receiver.rs:343-474).
 *
 * Input layout (f32 PCM, unscaled i16-range values as the reference expects,
 * lib.rs:78-81).  TIME_MAJOR is the native, coalesced layout:
 *   SAME_LAYOUT_TIME_MAJOR     x[t * n_channels + c]
 *   SAME_LAYOUT_CHANNEL_MAJOR  x[c * n_samples + t]   (transposed on the device first, slab by slab -- except in
 *                              time-parallel mode, where a 16-byte aligned f32 buffer of whole blocks is read in place)
 */
enum { SAME_LAYOUT_TIME_MAJOR = 0, SAME_LAYOUT_CHANNEL_MAJOR = 1 };

/* flags for same_batch_new */
enum {
    SAME_BATCH_LINK_ONLY = 1u << 0,   /* report link events only; skip the transport layer */
    SAME_BATCH_TRACE_SYMBOLS = 1u << 1,/* record every soft symbol (debug / parity tests) */
    SAME_BATCH_GENERIC_KERNEL = 1u << 2,/* always use the any-configuration kernel (tests) */
    /* Time-parallel ("fast") mode.  A long call on few channels is a few serial instruction streams on a
     * machine with a thousand SIMDs; with this flag a call is cut into K time chunks per channel that
     * run side by side (K * n_channels state columns: 32 768 fill the machine exactly once, which is the default).  Chunk 0
     * continues from the channel's state; every other chunk starts from a freshly built receiver a
     * warm-up (64 symbols by default) before the samples it owns, and a chunk keeps running past its
     * end until its channel has been seen idle (LinkState::NoCarrier), where the next chunk takes over.
     * What is approximated is the state a chunk starts from and -- where relaxed kernels exist: whole groups of 64
     * channels at 22.05 / 44.1 / 48 kHz -- the arithmetic inside the chunks, which is that of SAME_BATCH_RELAXED below (SAME_RELAXED=0 in the
     * environment keeps the chunks on the strict kernels).  At 22.05 kHz the mode wants a CHANNEL-MAJOR buffer: there
     * every channel is cut where it is quiet (at 44.1 / 48 kHz a channel-major call is transposed on the device first); a time-major buffer is cut at the same rows for all channels and its
     * chunks run on through whatever burst straddles a cut (an eighth slower at configs[1] and twice the HBM traffic,
     * DESIGN.md 6).
     * Contract (tests/test_time_parallel.py): burst bytes, their order and
     * the transport messages equal the reference's; link events are the same sequence with sample
     * counters within SAME_TP_EVENT_TOLERANCE_SYMBOLS symbols (Searching: anywhere inside the preamble);
     * soft symbols of an open squelch within 0.05 with equal sign; transport events are stamped from a
     * host-side symbol clock (within a few symbols of the reference's).  Calls too short to be cut, other
     * sample rates than 22.05 / 44.1 / 48 kHz, non-default equalizer orders and channel counts that are
     * not a multiple of 16 run as ordinary strict launches.  Not combinable with
     * SAME_BATCH_TRACE_SYMBOLS. */
    SAME_BATCH_TIME_PARALLEL = 1u << 3,
    /* Relaxed arithmetic ("fast mode" of the north star; same_kernels_sym.hip at 22.05, 44.1 and 48 kHz -- batches of any size in
     * whole groups of 64 channels, and inside time-parallel chunks --, same_kernels_relaxed.hip for 22.05 kHz batches that are not
     * whole groups of 64).  The reference's algorithm and every decision of it, with the rounding of the floating-point
     * expressions given up: matched filters as fused multiply-adds into four partial sums instead of one newest-first
     * chain (rx/filter.rs:363-377), |mark| and |space| as an f32 square root instead of hypot (rx/demod.rs:163), the AGC
     * update as gain * (1 - bw |x|) + bw (rx/agc.rs:72-77), reciprocals for the timing loop's and the equalizer's
     * divisions (the DC blocker keeps the reference's bits at every rate: divisions by 16 / 32 are exact, the 35-sample
     * averages of 48 kHz are computed operation for operation).  In same_kernels_sym.hip,
     * additionally, the stages of the receiver run as wavefronts one step apart (36 samples at 22.05 kHz, 72 at 44.1 / 48 kHz), and what one stage feeds back
     * to an earlier one arrives LATE by a fixed number of steps instead of at the sample of the symbol that caused it:
     *   - the lock at sync (AGC lock, locked loop bandwidth: receiver.rs:431-432) and what end() undoes (receiver.rs:479-490)
     *     take effect from the AGC's next block on -- a lock freezes the gain the AGC HAD at the symbol's sample, recomputed
     *     from the window's f32 samples, so the burst's soft symbols keep the reference's scale whatever the input's --
     *     and behind the timing loop's next symbol but one;
     *   - what the framer decides (squelch.lock(true) on Reading, end() on NoCarrier / Burst: receiver.rs:457-471) reaches
     *     the squelch and the equalizer one symbol late; a decision that a newer symbol has overtaken (the squelch lost or
     *     found sync in between) is dropped.
     * Precondition on the input: |x| * agc_bandwidth * samples_per_symbol / rate < 1 (|x| < 5.2e4 with the defaults at
     * 22.05 kHz).  Beyond it the REFERENCE's AGC overshoots zero and is clamped every other sample (rx/agc.rs:72-77); what
     * any arithmetic decodes from that limit cycle is a matter of rounding, and the two modes differ
     * (tests/test_sym_kernel.py::test_any_input_scale covers 0.5 .. 1e6).
     * The timing trajectory is chaotic in the last bit of those sums, so the contract is the
     * time-parallel mode's, whose chunks run this arithmetic as well (SAME_RELAXED=0 in the environment keeps them
     * strict): transmitted burst bytes and transport messages EQUAL, link events within
     * SAME_TP_EVENT_TOLERANCE_SYMBOLS symbols, soft symbols of an open squelch within 0.05 with equal sign
     * (tests/test_relaxed.py, tests/test_sym_kernel.py; directly against the oracle at all three rates).  A stream fed in
     * several calls meets the contract like one long call but is not bit-identical to it near a lock or an end(): at a call's end
     * everything in flight between the stages is applied at once (the samples behind the last whole step go to the strict
     * any-configuration kernel, which needs a canonical state) -- SAME_BATCH_CALL_INVARIANT below ties the launches to the
     * stream instead of to the calls and makes every call list come out the same.  22.05, 44.1 and 48 kHz with the default DC-blocker length,
     * default or disabled equalizer and a non-negative AGC floor (44.1 / 48 kHz: whole groups of 64 channels, any number
     * of them); any other configuration runs strict.  Strict mode (no flag) stays bit-exact. */
    SAME_BATCH_RELAXED = 1u << 4,
    /* Results that do not depend on how the stream is cut into calls, in every mode (round 6).  Strict launches have that
     * property by themselves.  A relaxed launch drains its pipeline at its end and applies the feedback in flight at once, a
     * time-parallel launch plans its cuts over what the call delivered: fed in different calls, a stream comes out within the
     * modes' contracts every time but not event for event the same.  With this flag the batch demodulates the stream in WINDOWS
     * of 18 432 samples (0.84 s at 22.05 kHz; 73 728 for a SAME_BATCH_TIME_PARALLEL batch, whose planner cuts every launch into
     * pieces; same_batch_set_call_window) that begin at fixed positions of the stream -- multiples of the window from the
     * batch's first sample, or from its last same_batch_flush / same_batch_reset -- whatever the calls look like: any list of
     * calls that delivers the same samples makes the same launches and therefore the same events, bit for bit.  Samples wait
     * in a device buffer until their window is whole (whole windows inside a call's buffer are demodulated where they lie): the
     * events of a window arrive when its last sample has; same_batch_flush demodulates what is waiting before its zeros have
     * filled the window, same_batch_sync does not.  Channel-major inputs take the transposing path (the per-channel cuts of
     * SAME_BATCH_TIME_PARALLEL on a channel-major input are planned per call).  same_batch_input_sample_counter counts the
     * samples accepted, waiting ones included.  Cost: what of a call does not lie in whole windows is copied once (a call
     * shorter than a window is copied whole: one more pass over the input), and launches are a window long. */
    SAME_BATCH_CALL_INVARIANT = 1u << 5
};
#define SAME_TP_EVENT_TOLERANCE_SYMBOLS 2

typedef struct same_batch same_batch;

/* build(): SAME_EDCLEN / SAME_EAGCLIMITS where the reference panics */
int same_batch_new(const same_rx_builder *b, uint32_t n_channels, int device,
                   uint32_t flags, same_batch **out);
void same_batch_free(same_batch *rx);
int same_batch_reset(same_batch *rx);                       /* SameReceiver::reset receiver.rs:182-198 */
uint32_t same_batch_input_rate(const same_batch *rx);       /* input_rate() :167-169 */
uint32_t same_batch_n_channels(const same_batch *rx);
uint64_t same_batch_input_sample_counter(const same_batch *rx); /* input_sample_counter() :175-177 */
int same_batch_device(const same_batch *rx);

/* The hot path.  `d_x` is a DEVICE pointer to n_samples * n_channels floats; `hip_stream`
 * is the hipStream_t the kernels are launched on.  Asynchronous: events become visible to
 * same_batch_poll_events after same_batch_sync.  Replaces the sample loop of
 * SameReceiver::process (receiver.rs:243-270) for every channel.
 *
 * Stream contract.  Every value of `hip_stream` other than SAME_STREAM_OWN is a real stream
 * handle -- NULL is HIP's legacy default stream, exactly as in hipLaunchKernelGGL -- and the
 * launch is ordered on it like any other work of the caller.  SAME_STREAM_OWN selects the
 * library's private non-blocking stream, which is NOT ordered after anything the caller has
 * queued elsewhere: whoever produced `d_x` on another stream calls same_batch_order_after(rx,
 * producer_stream) first (one event record + one stream wait, no host blocking).  In both
 * cases `d_x` must stay valid, and unmodified, until the launch has finished: until
 * same_batch_sync returns, or until the second-next process call on this handle returns
 * (at most two launches are in flight; a process call collects the launch before the
 * previous one).  A caching allocator must not be allowed to reuse the buffer earlier.
 * Successive launches of one batch are ordered among themselves whatever streams they are
 * given: a launch continues the state the previous one leaves, and waits for it. */
#define SAME_STREAM_OWN ((void *)(intptr_t)-1)
int same_batch_process_device(same_batch *rx, const float *d_x, size_t n_samples,
                              uint32_t layout, void *hip_stream);
/* SAME_BATCH_CALL_INVARIANT: the window length in samples (64 .. 4 194 304), before the batch's first sample (or right behind a
 * reset).  Two streams come out equal only if they are demodulated with equal windows. */
int same_batch_set_call_window(same_batch *rx, uint32_t samples);
/* make the library's own stream wait for everything queued on `producer_stream` so far
 * (a hipStream_t; NULL = the legacy default stream) */
int same_batch_order_after(same_batch *rx, void *producer_stream);
/* int16 PCM on the device, cast to f32 without scaling in the kernel
 * (crates/samedec/src/app.rs:112); 2 bytes/sample of HBM traffic */
int same_batch_process_device_i16(same_batch *rx, const int16_t *d_x, size_t n_samples,
                                  uint32_t layout, void *hip_stream);
/* host-buffer convenience: copies to the device, processes, synchronises */
int same_batch_process_host(same_batch *rx, const float *h_x, size_t n_samples, uint32_t layout);
int same_batch_process_host_i16(same_batch *rx, const int16_t *h_x, size_t n_samples, uint32_t layout);
/* SameReceiver::flush (receiver.rs:216-224): 4 * input_rate zero samples per channel.
 * Unlike the reference it does not stop at the first message; all events are reported. */
int same_batch_flush(same_batch *rx);
/* wait for all queued work of this handle */
int same_batch_sync(same_batch *rx);

/* Drain events already brought back to the host, ordered by (channel, sample_counter,
 * emission order) within each process call: per channel this is exactly the order
 * iter_events() yields them (receiver.rs:238-269).  Never blocks: a launch still running
 * contributes its events after same_batch_sync (or after the next process call, which
 * collects the previous launch while the new one runs).
 * Writes up to `cap` events, *n_out = number written, *n_left = events still queued. */
int same_batch_poll_events(same_batch *rx, same_rx_event *out, size_t cap, size_t *n_out,
                           size_t *n_left);
size_t same_batch_pending_events(same_batch *rx);
/* The same queue as an array inside the handle: *events points at *n queued events, valid until the
 * next call on this handle other than same_batch_pending_events / same_batch_drop_events;
 * same_batch_drop_events then removes the first n of them (n <= *n of the last peek).  For consumers that
 * scan the queue once and keep only the few events they care about (the bursts, say).  The handle queues
 * compact 48-byte records; the first peek after new events arrived builds the 328-byte same_rx_event array
 * (O(queue), on up to 8 threads), later peeks re-use it as long as events were only dropped.  The array is
 * not kept for the handle's life: once the queue has run empty it is released by the first call that ends the
 * view's life anyway (poll, peek, process, sync, reset) -- never by same_batch_drop_events itself, so
 * "peek, drop everything, finish reading the view" is safe. */
int same_batch_peek_events(same_batch *rx, const same_rx_event **events, size_t *n);
int same_batch_drop_events(same_batch *rx, size_t n);
/* The queued SAME_LINK_BURST events as fixed 304-byte records, in queue order: u32 channel +
 * first_channel, u64 sample_counter, u32 length (<= 288), 288 payload bytes zero-padded (the record
 * the multi-GPU gather moves, sameold_amd/distributed.py).  out == NULL: only counts.  Writes at most
 * `cap` records; *n_records = bursts queued.  The queue is left as it is. */
#define SAME_BURST_RECORD_BYTES 304
int same_batch_pack_bursts(same_batch *rx, uint32_t first_channel, uint8_t *out, size_t cap, size_t *n_records);

/* Time-parallel mode tuning (0 = keep the default): most chunks per channel (default: as many as fit
 * 32 768 state columns), fewest samples a chunk may own (default 4 x warm-up), warm-up samples
 * (default 64 symbols).  Takes effect from the next process call. */
int same_batch_time_parallel_config(same_batch *rx, uint32_t max_chunks, uint32_t min_own_samples,
                                    uint32_t warmup_samples);
/* chunks per channel the most recent process call was cut into (1 = it ran as one strict launch) */
uint32_t same_batch_time_parallel_chunks(const same_batch *rx);
/* 1 when that call's chunk boundaries were chosen per channel at idle instants: a channel-major f32 input of
 * whole blocks (n_samples a multiple of the kernel's block length and of 4) on a batch whose state columns fill
 * 64-channel workgroups.  Such a call reads the input where it lies (no transposition pass); its chunks seldom
 * run on, so it is the faster form of the mode. */
int same_batch_time_parallel_per_channel(const same_batch *rx);

/* soft-symbol trace (SAME_BATCH_TRACE_SYMBOLS): SymbolEstimate stream of one channel
 * (rx/symsync.rs:52-71) with the input sample counter of each TED instant */
typedef struct same_symbol_trace {
    uint64_t sample_counter;
    float zero, sym, err, samples_until_next_ted;
} same_symbol_trace;
int same_batch_read_trace(same_batch *rx, uint32_t channel, same_symbol_trace *out,
                          size_t cap, size_t *n_out);

/* device timing of the demodulation kernel(s) of the last process call, measured with
 * HIP events on the stream the kernel ran on (milliseconds); for bench.py's roofline */
int same_batch_last_kernel_ms(same_batch *rx, float *ms);
/* the demodulation kernel alone: equal to the above except for a time-parallel launch on a channel-major input, whose
 * figure above also covers the scout / planner / sort kernels (when they are not hidden under the previous launch) and
 * the state-column copies */
int same_batch_last_demod_kernel_ms(same_batch *rx, float *ms);
/* enable/disable that timing (off by default: two event records per call) */
void same_batch_set_kernel_timing(same_batch *rx, int enable);
/* name of the kernel variant the last call dispatched to (static string) */
const char *same_batch_kernel_name(const same_batch *rx);

/* ------------------------------------------------------------------ single receiver
 * `SameReceiver` with the reference's pull semantics, as a 1-channel batch.
 * same_rx_process mirrors iter_events()/process() (receiver.rs:119-130, 233-274):
 * returns 1 with *ev filled after consuming *consumed <= n samples, or 0 when all n
 * samples were consumed without an event.  The caller resumes at x + *consumed, exactly
 * as the Rust iterator leaves its source.  (The device runs ahead over the whole slice;
 * samples past *consumed must therefore be re-presented unchanged, which any iterator
 * adaptor does.) */
/* same_rx_flush mirrors flush() (receiver.rs:216-224): it feeds 4 s of zeros and returns at the
 * first Message.  The reference's iterator stops consuming zeros there; the device has already run
 * over all of them.  Further flushes are unaffected (they present zeros again, which is what the
 * device saw), and that is how samedec drains a file (crates/samedec/src/app.rs:118).  Real samples
 * are not: while the device is ahead over flush zeros, same_rx_process returns SAME_EINVAL instead
 * of silently skipping that many samples of the caller's audio; same_rx_reset (or flushing until
 * the zeros are used up) makes the handle usable again.  The same contract covers a caller that
 * abandons same_rx_process mid-slice: samples between *consumed and n must be re-presented
 * unchanged. */
typedef struct same_rx same_rx;
int same_rx_build(const same_rx_builder *b, int device, same_rx **out);  /* build() */
void same_rx_free(same_rx *rx);
int same_rx_process(same_rx *rx, const float *x, size_t n, size_t *consumed, same_rx_event *ev);
int same_rx_flush(same_rx *rx, same_rx_event *msg);      /* flush(): 1 if a message was produced */
int same_rx_reset(same_rx *rx);
uint32_t same_rx_input_rate(const same_rx *rx);
uint64_t same_rx_input_sample_counter(const same_rx *rx);

/* ------------------------------------------------------------------ helpers
 * device-side synthetic workload generator used by bench.py and the GPU tests: fills a
 * TIME_MAJOR f32 buffer with seeded continuous-phase AFSK bursts (see DESIGN.md
 * "Synthetic workload").  noise_sigma: AWGN std-dev relative to the carrier amplitude;
 * flags bit 0: even integer samples/symbol like the reference's test modulator. */
int same_synth_afsk_device(float *d_x, uint32_t n_channels, size_t n_samples,
                           uint32_t input_rate, uint64_t seed, float noise_sigma,
                           uint32_t flags, int device, void *hip_stream);
/* the header text the generator transmits on `channel` (host-side mirror) */
uint32_t same_synth_payload(uint64_t seed, uint32_t channel, uint8_t *out, uint32_t cap);
/* AWGN Monte-Carlo trials (BASELINE.json configs[4]): channel c of the TIME_MAJOR buffer is
 * trial first_trial + c: one burst (16 x 0xAB + same_synth_payload(seed, trial)) after a
 * ~0.1 s lead-in, in white Gaussian noise at Eb/N0 = ebn0_db_lo + (trial % n_grid) *
 * ebn0_db_step dB.  Noise: Philox4x32-10 keyed by the seed, counter = (sample block, trial),
 * Box-Muller in f32; sigma = A sqrt(sps / (4 Eb/N0)). */
int same_synth_trials_device(float *d_x, uint32_t n_trials, uint32_t first_trial, size_t n_samples,
                             uint32_t input_rate, uint64_t seed, float ebn0_db_lo, float ebn0_db_step,
                             uint32_t n_grid, int device, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SAME_RX_H */
