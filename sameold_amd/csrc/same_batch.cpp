// same_batch.cpp -- host side of the C ABI: owns the per-channel receiver state in HBM,
// launches the gfx950 kernels, and turns the device's append-only event log into the
// ordered event stream `SameReceiver::iter_events` yields (receiver.rs:119-130, 233-274).
//
// There is deliberately no CPU fallback here: if the HIP runtime or a gfx950 device is
// missing, every compute entry point fails with SAME_ENODEVICE / SAME_EHIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/same_rx.h"
#include "same_config.h"
#include "same_device.h"
#include "same_launch.h"
#include "same_transport.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(SAME_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                        __FILE__, __LINE__);                                               \
    } while (0)

}  // namespace

// ------------------------------------------------------------------------------------
// The host event queue.  A harvest of configs[1] brings ~110 000 link events per step; as 328-byte same_rx_event
// records that was 36 MB written twice per step (thread-local parts, then the queue) -- 2 ms of host time under a
// 2.7 ms launch.  The queue therefore holds 48-byte records, the burst / header bytes of the few events that carry any
// sit in a byte arena beside them, and a same_rx_event is only made when a consumer asks for one
// (same_batch_poll_events / _peek_events; same_batch_pack_bursts reads the records directly).
struct QEvent {
    uint32_t kind, channel;
    uint64_t sample_counter, symbol_count;
    uint32_t len, aux, aux2;
    uint32_t n_bytes;                // payload bytes kept (min(len, SAME_EVENT_MAX_BYTES)), 0 = none
    uint64_t payload;                // absolute offset of the payload in the arena
};
static_assert(sizeof(QEvent) == 48, "compact event record");
template <typename T>
struct PodQueue {                    // a plain growable array of PODs (no zero-fill on growth, appendable in slices)
    T *buf = nullptr;
    size_t n = 0, cap = 0;
    size_t base = 0;                 // how many elements have left the front since the handle was made: buf[i] is number base + i
    ~PodQueue() { std::free(buf); }
    PodQueue() = default;
    PodQueue(const PodQueue &) = delete;
    PodQueue &operator=(const PodQueue &) = delete;
    size_t size() const { return n; }
    T *data() { return buf; }
    const T *data() const { return buf; }
    void clear() { base += n; n = 0; }
    // drop the first `head` elements by moving the rest to the front; returns the new head (0)
    size_t compact(size_t head)
    {
        if (head == 0) return 0;
        if (head < n) std::memmove(buf, buf + head, (n - head) * sizeof(T));
        n -= head;
        base += head;
        return 0;
    }
    // room for `extra` more elements; returns where they go, or nullptr when out of memory
    T *grow(size_t extra)
    {
        if (n + extra > cap) {
            const size_t want = (n + extra) + (n + extra) / 2 + 64;
            void *p = std::realloc(buf, want * sizeof(T));
            if (!p) return nullptr;
            buf = static_cast<T *>(p);
            cap = want;
        }
        T *at = buf + n;
        n += extra;
        return at;
    }
};
using EventQueue = PodQueue<QEvent>;
using ByteArena = PodQueue<uint8_t>;     // payload bytes: element number = absolute offset
// SAME_HOST_PROF (measurement builds: SAME_EXTRA_DEFS=SAME_HOST_PROF=1): time-stamp-counter totals of the replay's parts,
// printed by same_debug_harvest_replay (one thread); nothing in the product build
#ifdef SAME_HOST_PROF
#include <x86intrin.h>
static uint64_t g_hp[8];
static const char *const g_hp_name[8] = {"range sort", "poll synthesis", "queue record + payload", "transport: burst", "transport: other", "after_event", "stitch scan", "-"};
struct HpScope { int k; uint64_t t0; explicit HpScope(int k_) : k(k_), t0(__rdtsc()) {} ~HpScope() { g_hp[k] += __rdtsc() - t0; } };
#define HP(k) HpScope hp_scope_##k(k)
#else
#define HP(k) do {} while (0)
#endif
struct HarvestPart {
    std::vector<QEvent> out;             // payload = offset into `bytes` until the part is appended to the queue
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> rearm;
    std::vector<uint32_t> bursts;        // indices into out
};
// a queue record as the event the ABI hands out
static void materialise(const QEvent &q, const uint8_t *arena, size_t arena_base, same_rx_event *ev)
{
    std::memset(ev, 0, sizeof(*ev));
    ev->kind = q.kind; ev->channel = q.channel; ev->sample_counter = q.sample_counter; ev->symbol_count = q.symbol_count;
    ev->len = q.len; ev->aux = q.aux; ev->aux2 = q.aux2;
    if (q.n_bytes) std::memcpy(ev->bytes, arena + (q.payload - arena_base), q.n_bytes);
}

using same::TickSynth;

// A few parked worker threads per batch for the harvest (replay, queue copy, burst packing): starting and joining 60
// std::threads per launch costs more than the work some of them do, and eight ranks share one host.
class WorkerPool {
public:
    ~WorkerPool()
    {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    // fn(0) .. fn(n - 1), the caller taking part; returns when all are done
    void run(size_t n, const std::function<void(size_t)> &fn)
    {
        if (n <= 1) { if (n) fn(0); return; }
        while (threads_.size() + 1 < n) threads_.emplace_back([this] { loop(); });
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn; next_ = 0; count_ = n; pending_ = n; ++generation_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void work()
    {
        for (;;) {
            size_t i;
            const std::function<void(size_t)> *fn;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!fn_ || next_ >= count_) return;
                i = next_++; fn = fn_;
            }
            (*fn)(i);
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            work();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t next_ = 0, count_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

struct same_batch {
    same_rx_builder builder{};
    same::Params P{};
    same::State S{};
    int device = 0;
    uint32_t flags = 0;
    uint64_t counter = 0;            // input_sample_counter (common to all channels)
    hipStream_t own_stream = nullptr;
    hipStream_t last_stream = nullptr;
    float4 *d_taps = nullptr;
    void *d_state_blob = nullptr;    // one allocation backing every State array
    size_t state_bytes = 0;
    // Output of a launch: event log + burst pool + cursors.  Two slots, so the host can
    // harvest launch k (copy back, order, transport layer) while launch k+1 runs.
    struct Slot {
        same::DevEvent *d_events = nullptr; uint32_t event_cap = 0;
        // the log's indices ordered by state column, made on the device behind the launch (launch_event_sort)
        uint32_t *d_sort = nullptr; size_t sort_words = 0;          // cnt [bins] | first [bins + 1] | the scan's workgroup totals
        uint32_t *h_sort = nullptr; size_t h_sort_words = 0;        // pinned: first [bins + 1]
        same::DevEvent *d_sorted = nullptr; uint32_t sorted_cap = 0; // the log's records in column order
        uint32_t sort_bins = 0;
        uint8_t *d_bursts = nullptr; uint32_t burst_cap = 0;
        uint32_t *d_counters = nullptr;  // [0] n_events [1] n_bursts [2] overflow
        uint32_t *h_counters = nullptr;  // pinned, host-mapped
        uint32_t *h_counters_dev = nullptr;  // device view of h_counters
        hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_done = nullptr, ev_planned = nullptr;
        hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;         // around the demodulation kernel alone (time-parallel launches bracket more with ev_start / ev_stop)
        bool have_k = false;
        bool timed = false;               // ev_start / ev_stop were recorded for this launch (latched when it was made)
        // pinned landing buffers of the read-back, grown on demand.  A copy into pageable memory
        // is staged by the runtime (blit kernel + host memcpy per chunk) and, queued beside the
        // next launch, holds that launch up for as long as the host is busy.
        void *h_events = nullptr; size_t h_events_bytes = 0;
        void *h_bursts = nullptr; size_t h_bursts_bytes = 0;
        bool in_flight = false;
        uint64_t seq = 0;                // launch order
        uint64_t end_counter = 0;        // input sample counter after this launch
        // time-parallel launch: geometry, the end of its last whole block, the hand-over instants
        bool chunked = false;
        same::ChunkGeom geom{};
        uint64_t end_blocks = 0;
        uint64_t *d_handover = nullptr, *h_handover = nullptr;
        size_t handover_cap = 0;
        // per-channel chunk boundaries (channel-major input): device arrays of the launch, host copies for the stitch
        bool per_channel = false;
        uint32_t *d_geom = nullptr;      // [own_start | row0 | nominal | perm] x columns, then wg_blocks
        uint32_t *h_geom = nullptr;      // pinned: own_start | row0
        size_t geom_cap = 0;
    } slot[2];
    // time-parallel mode (SAME_BATCH_TIME_PARALLEL)
    struct TimePar {
        bool enabled = false;
        uint32_t max_chunks = 0, min_own = 0, warmup = 0;     // same_batch_time_parallel_config (0 = default)
        uint32_t cap_columns = 0;                             // columns the wide state blob holds
        uint32_t carved_columns = 0;                          // columns Pv / Sv / the descriptor tables are laid out for
        int carved_kernel = -1;                               // ... and the kernel choice Pv's knobs were set for
        same::Params Pv{};
        same::State Sv{};
        void *blob = nullptr;
        void *blob_fresh = nullptr;                           // the same layout, every column a freshly built receiver: the template a launch's prologue copies
        size_t fresh_bytes = 0;                               // ... and how much of it a launch copies (the [rows][column] arrays; not the framer's rows)
        same::StateArrayDesc *d_desc_in = nullptr, *d_desc_out = nullptr;   // real -> wide, wide -> real
        uint32_t n_desc = 0;
        uint32_t *d_final_col = nullptr;
        float *d_energy = nullptr; size_t energy_cap = 0;   // scout scratch
        float *d_hist = nullptr; size_t hist_cap = 0;       // squelch histories by grid position (PipeChunks::hist_scratch)
        hipEvent_t ev_plan_prev = nullptr; bool plan_recorded = false;
        int knob_plan_stream = 0;                            // SAME_TP_PLAN_STREAM=0: planning kernels stay on the launch stream (A/B measurements)
        int sort_mode = -1;                                  // SAME_TP_SORT: -1 choose, 0 grid order, 1 pieces sorted by length into workgroups, 2 groups of 64 paired long with short
        uint32_t last_chunks = 1;
        bool last_per_channel = false;
        // which kernel runs the chunks of the call being planned: the wavefront pipeline (strict, or its FASTMATH build
        // when the batch is relaxed) or the one-wavefront relaxed kernel
        enum Kernel { kPipe = 0, kPipeRelaxed = 1, kWaveRelaxed = 2 } kernel = kPipe;
        int knob_kernel = 0;                                  // SAME_TP_KERNEL: 1 pipeline, 2 one-wavefront relaxed kernel, 0 choose
        int knob_prologue = 0;                                // SAME_TP_PROLOGUE=0: init kernel, fill and cursor reset as separate launches (A/B measurements)
        std::vector<int64_t> sym_off;        // per channel: reported symbol count - the device's
        std::vector<TickSynth> synth;
    } tp;
    uint64_t launch_seq = 0;
    hipStream_t copy_stream = nullptr;   // read-back of finished launches, beside the compute stream
    // SAME_BATCH_CALL_INVARIANT: the stream is demodulated in windows of kInvWindow samples that begin at fixed positions of the
    // STREAM (multiples of the window from the batch's first sample, or from its last flush / reset), whatever the calls that
    // deliver it look like: samples wait in `d_buf` (f32, time-major) until their window is whole
    struct Windowed {
        bool on = false;
        uint32_t window = 0;           // samples per window
        uint32_t fill = 0;             // samples of the current window that have arrived
        void *d_buf = nullptr; size_t buf_bytes = 0;
        hipEvent_t ev_buf = nullptr;   // behind the last operation on d_buf (a later call may come on another stream)
        hipStream_t buf_stream = nullptr; bool buf_used = false;
    } inv;
    // Time-parallel launches on the library's own stream: scout, planner and sort of call k + 1 run here, beside the tail of
    // launch k (whose short workgroups have left their CUs by then); the own stream waits for them (Slot::ev_planned)
    hipStream_t plan_stream = nullptr;
    float last_ms = 0.0f;
    bool overflowed = false;
    std::string record_path;         // SAME_RECORD_HARVEST (measurement aid of tools/host_step_probe.py: read once per batch, like every knob)
    bool kernel_fault = false;       // counters[2] bit 2: a wavefront pipeline's bounded hand-over wait ran out (same_kernels_sym.hip)
    bool use_fast = false;           // configuration has a latency-optimised kernel
    bool relaxed = false;            // relaxed arithmetic in time-parallel chunks (SAME_BATCH_TIME_PARALLEL or SAME_BATCH_RELAXED)
    bool last_plain_fm = false;      // the last ordinary launch ran the pipeline's FASTMATH build
    bool last_plain_wave = false;    // ... the one- / two-wavefront relaxed kernel
    bool last_fm_sym = false;        // ... or rather the symbol-paced pipeline (same_kernels_sym.hip)
    bool relaxed_plain = false;      // ... and in ordinary launches: the one-wavefront relaxed kernel runs whole blocks (SAME_BATCH_RELAXED)
    int knob_relaxed = 0;            // SAME_RELAXED: -1 never (time-parallel chunks keep the strict pipeline), +1 as if SAME_BATCH_RELAXED were set
    bool force_generic = false;      // SAME_BATCH_GENERIC_KERNEL (tests compare both kernels)
    bool debug = false;              // SAME_DEBUG: harvest statistics on stderr
    int host_threads = 0;            // SAME_HOST_THREADS: harvest threads (0 = choose)
    uint32_t sym_max_channels = 1u << 30;
    // staging for host / channel-major inputs
    void *d_stage = nullptr; size_t stage_bytes = 0;
    void *d_stage2 = nullptr; size_t stage2_bytes = 0;
    void *d_upload = nullptr; size_t upload_bytes = 0;      // host-buffer entry points: grow-only upload slab
    void *d_zero = nullptr; size_t zero_bytes = 0;          // flush: grow-only slab of zeros
    hipEvent_t ev_order = nullptr;                          // same_batch_order_after
    // kernel timing
    bool timing = false;
    bool have_timing = false;
    float last_demod_ms = 0.0f;         // the demodulation kernel alone (= last_ms unless the launch brackets planning kernels too)
    // ordered host-side event queue
    EventQueue queue;                   // events not yet polled: [queue_head, size)
    size_t queue_head = 0;
    ByteArena arena;                    // their payload bytes
    std::vector<same_rx_event> peeked;  // same_batch_peek_events: the queued events, materialised (328 bytes each) ...
    bool peeked_valid = false;          // ... and whether it still mirrors queue[peeked_head ..) (a harvest appends: invalid)
    bool peeked_release = false;        // the queue ran empty under same_batch_drop_events: the caller may still be reading the view; freed by the next call that ends its life
    size_t peeked_head = 0;
    std::vector<size_t> burst_seq;      // event numbers (EventQueue::base + index) of the queued SAME_LINK_BURST events, ascending
    size_t burst_seq_head = 0;          // entries before this one have been polled or dropped
    std::vector<HarvestPart> parts;     // per host thread, kept between harvests for their capacity
    WorkerPool workers;
    // transport layer, one assembler per channel (unless SAME_BATCH_LINK_ONLY)
    // (two arrays: one cache line per channel that every poll touches, and the burst bytes / message texts: same_transport.h)
    std::vector<same::TransportHot> thot;
    std::vector<same::TransportCold> tcold;
    same::TransportRef tr(uint32_t c) { return same::TransportRef(thot[c], tcold[c]); }
    uint64_t *h_wake = nullptr;      // host mirror of State::wake_sample (pinned, n_channels words, zero = unarmed)
};

static void release_stale_view(same_batch *rx);

namespace {

struct Carver {
    size_t off = 0;
    template <typename T> size_t take(size_t n)
    {
        off = (off + 255) & ~size_t(255);
        size_t o = off;
        off += n * sizeof(T);
        return o;
    }
};

// every State array that is laid out [rows][channel]: field, element type, rows
#define SAME_STATE_ARRAYS(X, P)                                                                            \
    X(dc_ff_ring, float, (P).dc_len) X(dc_fb_ring, float, (P).dc_len)                                        \
    X(dc_sum0, float, 1) X(dc_sum1, float, 1) X(agc_gain, float, 1)                                          \
    X(win_ring, float, (P).win_ring)                                                                         \
    X(ted_clock, uint32_t, 1) X(until_next_ted, float, 1)                                                    \
    X(ted_h0, float, 1) X(ted_h1, float, 1) X(ted_h2, float, 1)                                              \
    X(period_avg, float, 1) X(period_inst, float, 1)                                                         \
    X(sq_data, uint32_t, 1) X(sq_power, float, 1) X(sq_phist, uint32_t, 1)                                   \
    X(sq_fill, uint32_t, 1) X(sq_clock, int32_t, 1) X(sq_symbols, uint64_t, 1)                               \
    X(sq_hist, float, same::kSquelchHist)                                                                    \
    X(eq_ffc, float, (P).eq_nff) X(eq_fbc, float, (P).eq_nfb)                                                \
    X(eq_ffw, float, (P).eq_nff) X(eq_fbw, float, (P).eq_nfb)                                                \
    X(eq_word, uint32_t, 1) X(eq_count, uint32_t, 1)                                                         \
    X(eq_snap_ffc, float, (P).eq_nff) X(eq_snap_fbc, float, (P).eq_nfb)                                      \
    X(eq_snap_ffw, float, (P).eq_nff) X(eq_snap_fbw, float, (P).eq_nfb)                                      \
    X(fr_word, uint32_t, 1) X(fr_count, uint32_t, 1) X(fr_invalid, uint32_t, 1) X(fr_len, uint32_t, 1)       \
    X(flags, uint32_t, 1)                                                                                    \
    X(tk_next, uint64_t, 1) X(tk_last, uint64_t, 1) X(tk_ring, uint64_t, same::kTickRing)                    \
    X(tk_n, uint32_t, 1) X(wake_sample, uint64_t, 1) X(wake_fired, uint64_t, 1)

// lays out every State array inside one blob; with base == nullptr only sizes it
size_t carve_state(const same::Params &P, char *base, same::State &S)
{
    const size_t C = P.n_channels;
    Carver cv;
#define CARVE(field, type, count) \
    do { size_t o = cv.take<type>(count); if (base) S.field = reinterpret_cast<type *>(base + o); } while (0)
#define CARVE_ROWS(field, type, rows) CARVE(field, type, (size_t)(rows) * C);
    SAME_STATE_ARRAYS(CARVE_ROWS, P)
#undef CARVE_ROWS
    CARVE(fr_msg, uint8_t, (size_t)same::kBurstCap * C);      // [channel][kBurstCap]
    if (P.trace_cap) {
        CARVE(trace_n, uint32_t, C);
        CARVE(trace, float, (size_t)P.trace_cap * 4 * C);
        CARVE(trace_idx, uint64_t, (size_t)P.trace_cap * C);
    }
#undef CARVE
    return (cv.off + 255) & ~size_t(255);
}

// the same arrays as (source, destination) pairs for launch_copy_state_columns (no trace arrays: the
// time-parallel mode does not record one)
void list_state_arrays(const same::Params &P, const same::State &src, const same::State &dst,
                       std::vector<same::StateArrayDesc> &out)
{
    out.clear();
#define DESC_ROWS(field, type, rows)                                                                     \
    out.push_back(same::StateArrayDesc{reinterpret_cast<char *>(src.field), reinterpret_cast<char *>(dst.field), \
                                       (uint32_t)(rows), (uint32_t)(sizeof(type) / 4)});
    SAME_STATE_ARRAYS(DESC_ROWS, P)
#undef DESC_ROWS
    static_assert(same::kBurstCap % 4 == 0, "framer rows are copied in words");
    out.push_back(same::StateArrayDesc{reinterpret_cast<char *>(src.fr_msg), reinterpret_cast<char *>(dst.fr_msg), 1u,
                                       (uint32_t)(same::kBurstCap / 4)});
}

// every environment knob of the library, read once per batch
void read_knobs(same_batch *rx)
{
    auto num = [](const char *name, int unset) { const char *e = std::getenv(name); return e ? std::atoi(e) : unset; };
    auto tri = [](const char *name) { const char *e = std::getenv(name); return e ? (std::atoi(e) ? 1 : -1) : 0; };
    rx->P.knob_pipe = tri("SAME_PIPE");
    rx->P.knob_pipe_lanes = num("SAME_PIPE_LANES", 0);
    rx->P.knob_pipe_split = tri("SAME_PIPE_SPLIT");            // (test knob: tests/test_gpu_parity.py pits the two stage-2 forms against each other)
#ifdef SAME_PROFILE
    // measurement knobs: read by profile builds only (python -m sameold_amd.build with SAME_PROFILE set), never by the shipped library
    rx->P.knob_mirror = tri("SAME_MIRROR");
    rx->P.knob_pipe_share = tri("SAME_PIPE_SHARE");
    rx->P.knob_prio = num("SAME_PIPE_PRIO", 0);
#endif
    rx->P.knob_fast_dense = tri("SAME_FAST_DENSE");
    rx->P.knob_sym = tri("SAME_SYM");
    rx->sym_max_channels = (uint32_t)std::max(0, num("SAME_SYM_MAX", 1 << 30));     // (measurement knob: up to where an ordinary relaxed launch takes the symbol-paced pipeline; default: always)
    rx->debug = std::getenv("SAME_DEBUG") != nullptr;
    { const char *e = std::getenv("SAME_RECORD_HARVEST"); rx->record_path = e ? e : ""; }
    rx->host_threads = std::max(0, num("SAME_HOST_THREADS", 0));
    rx->tp.sort_mode = num("SAME_TP_SORT", -1);
    rx->tp.knob_plan_stream = tri("SAME_TP_PLAN_STREAM");
    rx->tp.knob_prologue = tri("SAME_TP_PROLOGUE");
    rx->knob_relaxed = tri("SAME_RELAXED");
    { const char *e = std::getenv("SAME_RELAXED_KERNEL"); rx->P.knob_relaxed_kernel = !e ? 0 : (std::strcmp(e, "solo") == 0 ? 1 : (std::strcmp(e, "duo") == 0 ? 2 : 0)); }
    { const char *e = std::getenv("SAME_TP_KERNEL"); rx->tp.knob_kernel = !e ? 0 : (std::strcmp(e, "wave") == 0 ? 2 : (std::strcmp(e, "pipe") == 0 ? 1 : 0)); }
}

int ensure_output(same_batch *rx, same_batch::Slot &sl, size_t n_samples, same::Output &O, size_t n_columns = 0)
{
    const size_t n_ch = n_columns ? n_columns : rx->P.n_channels;
    // Worst case per channel: an acquisition attempt (Searching ... NoCarrier) needs a
    // fresh byte sync, i.e. at least 32 symbols, so < 2 link events per 32 symbols; bursts
    // are rarer still.  Size generously: 4 events per 32 symbols + slack.
    const double symbols = (double)n_samples * 520.83 / (double)rx->P.input_rate;
    const size_t per_chan_events = (size_t)(symbols / 8.0) + 16;
    const size_t per_chan_bursts = (size_t)(symbols / 160.0) + 4;   // a burst is >= 20 bytes
    size_t ecap = std::min<size_t>(per_chan_events * n_ch, 0x7fffffffu / sizeof(same::DevEvent));
    size_t bcap = std::min<size_t>(per_chan_bursts * n_ch, 0x7fffffffu / same::kBurstCap);
    if (ecap > sl.event_cap) {
        if (sl.d_events) HIP_TRY(hipFree(sl.d_events));
        sl.d_events = nullptr; sl.event_cap = 0;
        HIP_TRY(hipMalloc((void **)&sl.d_events, ecap * sizeof(same::DevEvent)));
        sl.event_cap = (uint32_t)ecap;
    }
    if (bcap > sl.burst_cap) {
        if (sl.d_bursts) HIP_TRY(hipFree(sl.d_bursts));
        sl.d_bursts = nullptr; sl.burst_cap = 0;
        HIP_TRY(hipMalloc((void **)&sl.d_bursts, bcap * same::kBurstCap));
        sl.burst_cap = (uint32_t)bcap;
    }
    {
        const size_t need = 2 * n_ch + 1 + same::event_sort_extra_words((uint32_t)n_ch), need_h = n_ch + 1;
        if (sl.event_cap > sl.sorted_cap) {
            if (sl.d_sorted) HIP_TRY(hipFree(sl.d_sorted));
            sl.d_sorted = nullptr; sl.sorted_cap = 0;
            HIP_TRY(hipMalloc((void **)&sl.d_sorted, (size_t)sl.event_cap * sizeof(same::DevEvent)));
            sl.sorted_cap = sl.event_cap;
        }
        if (need > sl.sort_words) {
            if (sl.d_sort) HIP_TRY(hipFree(sl.d_sort));
            sl.d_sort = nullptr; sl.sort_words = 0;
            HIP_TRY(hipMalloc((void **)&sl.d_sort, need * sizeof(uint32_t)));
            sl.sort_words = need;
        }
        if (need_h > sl.h_sort_words) {
            if (sl.h_sort) HIP_TRY(hipHostFree(sl.h_sort));
            sl.h_sort = nullptr; sl.h_sort_words = 0;
            HIP_TRY(hipHostMalloc((void **)&sl.h_sort, need_h * sizeof(uint32_t), hipHostMallocDefault));
            sl.h_sort_words = need_h;
        }
        sl.sort_bins = (uint32_t)n_ch;
    }
    O.events = sl.d_events; O.event_cap = sl.event_cap;
    O.bursts = sl.d_bursts; O.burst_cap = sl.burst_cap;
    O.n_events = sl.d_counters; O.n_bursts = sl.d_counters + 1; O.overflow = sl.d_counters + 2;
    return SAME_OK;
}

int ensure_stage(void **p, size_t *have, size_t need)
{
    if (need <= *have) return SAME_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr; *have = 0;
    HIP_TRY(hipMalloc(p, need));
    *have = need;
    return SAME_OK;
}

// The relaxed-arithmetic pipeline of a launch over Pv.n_channels state columns: the symbol-paced one (36-sample steps,
// same_kernels_sym.hip; 72-sample steps at 44.1 / 48 kHz) where it is built, else the FASTMATH build of the strict pipeline
// the configuration as the pipeline's FASTMATH build takes it: 64-channel workgroups, the split form
same::Params fm_params(const same::Params &P)
{
    same::Params Pfm = P;
    Pfm.knob_pipe_lanes = 64; Pfm.knob_pipe_share = 1; Pfm.knob_pipe_split = 1; Pfm.knob_pipe = 1;
    return Pfm;
}
uint32_t fm_block_len(const same::Params &Pv) { return same::sym_kernel_supported(Pv) ? same::sym_block_len(Pv) : same::pipe_block_len(Pv); }
template <typename SampleT>
hipError_t launch_fm(const same::Params &Pv, const same::State &Sv, const same::Output &O, const float4 *taps, const SampleT *x,
                     uint32_t n_blocks, uint64_t counter0, hipStream_t stream, const same::PipeChunks &pc)
{
    if constexpr (sizeof(SampleT) == 4) {
        if (same::sym_kernel_supported(Pv)) return same::launch_demod_sym(Pv, Sv, O, taps, (const float *)x, n_blocks, counter0, stream, pc);
        return same::launch_demod_pipe(Pv, Sv, O, taps, (const float *)x, n_blocks, counter0, stream, pc, true);
    } else {
        if (same::sym_kernel_supported(Pv)) return same::launch_demod_sym_i16(Pv, Sv, O, taps, (const int16_t *)x, n_blocks, counter0, stream, pc);
        return same::launch_demod_pipe_i16(Pv, Sv, O, taps, (const int16_t *)x, n_blocks, counter0, stream, pc, true);
    }
}

// The host half of a harvest: the launch's ordered event log, burst pool, hand-over instants and chunk geometry are in the
// slot's host buffers; order each column's records, replay the channels (stitch + transport layer) on the worker threads and
// append to the queue.  Touches no device: same_debug_harvest_replay runs it on a recorded launch without one.
struct HarvestTimes { std::chrono::steady_clock::time_point sorted, replayed; uint32_t n_threads = 1; };
int harvest_host(same_batch *rx, same_batch::Slot &sl, uint32_t n_events, uint32_t n_bursts, std::vector<uint32_t> &rearm, HarvestTimes &times)
{
    const bool dbg = rx->debug;
    const same::DevEvent *evs = static_cast<const same::DevEvent *>(sl.h_events);
    const uint8_t *bursts = static_cast<const uint8_t *>(sl.h_bursts);
    const uint32_t n_ch = rx->P.n_channels;
    const uint32_t n_bins = sl.chunked ? sl.geom.n_chunks * n_ch : n_ch;     // state columns of the launch
    // Per column the device emits in time order (a lane takes its log slots one after the other); across lanes the
    // atomic cursor interleaves.  The device has counted the events per column and moved the records into column
    // ranges (launch_event_sort: `evs` is that ordered copy, a record's `channel` holding its index in the log); inside
    // a range they stand in the order the scatter's atomics landed, and the replay threads sort each range (a handful
    // of records) back into log order = time order before they walk it.
    // (slots a wavefront reserved but did not use carry kDevEventNone and were skipped there)
    std::vector<uint32_t> first(sl.h_sort, sl.h_sort + n_bins + 1u);
    same::DevEvent *evs_mut = static_cast<same::DevEvent *>(sl.h_events);
    const uint32_t n_real = first[n_bins];
    if (n_real > n_events) return fail(SAME_EHIP, "internal: event sort counted %u of %u events", n_real, n_events);
    // (what follows indexes the ordered copy directly)
    struct Identity { uint32_t operator[](uint32_t i) const { return i; } } order;
    if (dbg && sl.chunked && sl.per_channel) {
        // how the per-channel boundaries came out: chunk lengths (own range + warm-up) and run-ons, in samples
        const same::ChunkGeom &g = sl.geom;
        const uint32_t *own = sl.h_geom, *rows = sl.h_geom + n_bins;
        double len_sum = 0, run_sum = 0; uint64_t len_max = 0, run_max = 0, n_len = 0, n_run = 0, n_inf = 0, n_late = 0;
        uint32_t worst_c = 0;
        for (uint32_t k = 0; k + 1u < g.n_chunks; ++k)
            for (uint32_t c = 0; c < n_ch; ++c) {
                const uint64_t end = own[(size_t)(k + 1u) * n_ch + c], len = end - rows[(size_t)k * n_ch + c];
                len_sum += (double)len; if (len > len_max) { len_max = len; worst_c = c; } ++n_len;
                const uint64_t h = sl.h_handover[(size_t)k * n_ch + c];
                if (h == same::kNoHandover) { ++n_inf; continue; }
                const uint64_t run = h - g.counter0 > end ? h - g.counter0 - end : 0;
                run_sum += (double)run; run_max = std::max(run_max, run); ++n_run; n_late += run > 2u * g.block_len;
            }
        std::fprintf(stderr, "[same] longest piece on channel %u, its cuts:", worst_c);
        for (uint32_t k = 0; k < g.n_chunks; ++k) std::fprintf(stderr, " %u", own[(size_t)k * n_ch + worst_c]);
        std::fprintf(stderr, "\n");
        std::fprintf(stderr, "[same] per-channel chunks: length mean %.0f max %llu samples; run-on mean %.0f max %llu, %llu of %llu columns ran on "
                             "more than two blocks, %llu never handed over\n", len_sum / std::max<uint64_t>(n_len, 1), (unsigned long long)len_max,
                     run_sum / std::max<uint64_t>(n_run, 1), (unsigned long long)run_max, (unsigned long long)n_late, (unsigned long long)n_run,
                     (unsigned long long)n_inf);
    }
    // events per real channel, cumulative (a chunked launch spreads a channel over n_chunks columns)
    std::vector<uint32_t> chan_first;
    if (sl.chunked) {
        chan_first.assign(n_ch + 1u, 0u);
        for (uint32_t k = 0; k < sl.geom.n_chunks; ++k)
            for (uint32_t c = 0; c < n_ch; ++c) chan_first[c + 1u] += first[k * n_ch + c + 1u] - first[k * n_ch + c];
        for (uint32_t c = 0; c < n_ch; ++c) chan_first[c + 1u] += chan_first[c];
    }
    const std::vector<uint32_t> &cfirst = sl.chunked ? chan_first : first;
    const bool link_only = (rx->flags & SAME_BATCH_LINK_ONLY) != 0;
    const bool tp = rx->tp.enabled;
    times.sorted = std::chrono::steady_clock::now();

    // Channels are independent (one Transport each), so contiguous channel ranges are replayed
    // on separate host threads; each produces its slice of the output queue, in order.
    using Part = HarvestPart;
    const double sps = (double)rx->P.input_rate / 520.83;
    const uint64_t interburst = same::max_interburst_symbols(), history = same::max_history_duration();
    // one device event -> the link event of channel c (+ the transport event it causes).  `off`: what
    // the time-parallel mode adds to the device's symbol count (0 otherwise).
    // (transport events arrive as same_rx_event from the transport layer: kept as a record + payload)
    auto push_transport = [&](Part &part, const same_rx_event &tev, uint32_t c) {
        QEvent q{};
        q.kind = tev.kind; q.channel = c; q.sample_counter = tev.sample_counter; q.symbol_count = tev.symbol_count;
        q.len = tev.len; q.aux = tev.aux; q.aux2 = tev.aux2;
        q.n_bytes = std::min<uint32_t>(tev.len, SAME_EVENT_MAX_BYTES);
        if (tev.kind != SAME_TRANSPORT_MSG_START) q.n_bytes = 0;       // (only a header carries bytes)
        q.payload = part.bytes.size();
        if (q.n_bytes) part.bytes.insert(part.bytes.end(), tev.bytes, tev.bytes + q.n_bytes);
        part.out.push_back(q);
    };
    auto feed = [&](Part &part, same_rx_event &tev, const same::DevEvent &d, uint32_t c, int64_t off) {
        const uint64_t sym = (uint64_t)((int64_t)d.symbol_count + off);
        auto poll = [&](uint64_t psym, uint64_t pt) {
            if (rx->tr(c).on_link_event(same::kDevTick, pt, psym, nullptr, 0, rx->P.input_rate, &tev)) push_transport(part, tev, c);
        };
        if (tp && !link_only) {
            HP(1);
            rx->tp.synth[c].run_until(sym, d.sample_counter, sps, rx->tr(c).force_eom_at(), poll);
        }
        QEvent q{};
        const uint8_t *payload = nullptr;
        {
        HP(2);
        q.kind = d.kind; q.channel = c; q.sample_counter = d.sample_counter; q.symbol_count = sym;
        if (d.kind == SAME_LINK_BURST) {
            q.len = d.burst_len;
            if (d.burst_slot < n_bursts) {
                q.n_bytes = std::min<uint32_t>(d.burst_len, SAME_EVENT_MAX_BYTES);
                payload = bursts + (size_t)d.burst_slot * same::kBurstCap;
                q.payload = part.bytes.size();
                part.bytes.insert(part.bytes.end(), payload, payload + q.n_bytes);
            } else {
                q.len = 0;     // pool overflow: the burst bytes were lost (SAME_EOVERFLOW is reported)
            }
            part.bursts.push_back((uint32_t)part.out.size());
        }
        if (d.kind <= SAME_LINK_BURST) part.out.push_back(q);
        }
        if (!link_only) {
#ifdef SAME_HOST_PROF
            HpScope hp_tr(d.kind == SAME_LINK_BURST ? 3 : 4);
#endif
            if (rx->tr(c).on_link_event(d.kind, d.sample_counter, sym, payload, q.n_bytes, rx->P.input_rate, &tev)) push_transport(part, tev, c);
            if (!tp && rx->tr(c).force_eom_dirty()) part.rearm.push_back(c);
        }
        { HP(5); if (tp && d.kind <= SAME_LINK_BURST) rx->tp.synth[c].after_event(d.kind, sym, d.sample_counter, interburst, history); }
    };
    // Time-parallel launch: the events of channel c, stitched from its chunks.  Chunk `cur` is kept up to
    // its hand-over instant h (the end of the first block at or after its nominal end in which the
    // device saw it idle), then the chunk that owns h takes over -- from h if it is idle there too,
    // otherwise from its own next NoCarrier on (it was still busy with the tail of a burst it joined in
    // the middle, or with a duplicate of the burst the previous chunk has just delivered).
    auto stitch = [&](Part &part, same_rx_event &ev, uint32_t c) {
        const same::ChunkGeom &g = sl.geom;
        const uint64_t *hand = sl.h_handover;
        // per-channel boundaries: own_start[k][c] and row0[k][c] as the device planned them
        const uint32_t *own = sl.per_channel ? sl.h_geom : nullptr, *rows = sl.per_channel ? sl.h_geom + n_bins : nullptr;
        auto owner_of = [&](uint64_t h) -> uint32_t {
            if (!own) return g.owner_of(h);
            uint32_t k = 0;
            while (k + 1u < g.n_chunks && g.counter0 + own[(size_t)(k + 1u) * n_ch + c] <= h) ++k;
            return k;
        };
        TickSynth &ts = rx->tp.synth[c];
        int64_t off = rx->tp.sym_off[c];
        uint32_t cur = 0;
        uint64_t keep_after = 0;
        for (;;) {
            const uint32_t col = cur * n_ch + c;
            const uint64_t h = hand[col];
            const uint64_t upto = std::min(h, sl.end_blocks);
            bool rebased = cur == 0;
            for (uint32_t i = first[col]; i < first[col + 1u]; ++i) {
                const same::DevEvent &d = evs[order[i]];
                if (d.sample_counter <= keep_after) continue;
                if (d.sample_counter > upto) break;
                if (!rebased) {
                    // continue the channel's symbol clock: the last reported event plus the nominal symbol rate
                    const int64_t est = (int64_t)ts.a_sym + (int64_t)((double)(d.sample_counter - ts.a_t) / sps + 0.5);
                    off = est - (int64_t)d.symbol_count;
                    rebased = true;
                }
                feed(part, ev, d, c, off);
            }
            if (!rebased) {
                // nothing reported from this chunk: its counter started at 0 at its first row
                const uint64_t row0 = rows ? g.counter0 + rows[col] : g.counter0 + (uint64_t)cur * g.stride_blocks * g.block_len;
                const int64_t est_end = (int64_t)ts.a_sym + (int64_t)((double)(sl.end_blocks - ts.a_t) / sps + 0.5);
                off = est_end - (int64_t)((double)(sl.end_blocks - row0) / sps + 0.5);
            }
            if (h == same::kNoHandover) break;
            const uint32_t nxt = owner_of(h);
            if (nxt <= cur) break;
            // the next chunk's link state at h, and where it is first idle from there on
            const uint32_t ncol = nxt * n_ch + c;
            uint32_t st = 0, j = first[ncol];
            for (; j < first[ncol + 1u] && evs[order[j]].sample_counter <= h; ++j)
                if (evs[order[j]].kind <= SAME_LINK_BURST) st = evs[order[j]].kind;
            keep_after = h;
            if (st == SAME_LINK_READING || st == SAME_LINK_BURST) {
                // busy with a burst the previous chunk has already delivered (or with the garbage a chunk that
                // joined in the middle of one makes of it): its events count from its next NoCarrier on
                for (; j < first[ncol + 1u]; ++j)
                    if (evs[order[j]].kind == SAME_LINK_NO_CARRIER) { keep_after = evs[order[j]].sample_counter; break; }
            } else if (st == SAME_LINK_SEARCHING && ts.link == SAME_LINK_NO_CARRIER) {
                // it has byte sync where the previous chunk has none yet (the two acquire a preamble some
                // symbols apart): what follows -- Reading, Burst -- is real, so the channel is Searching from here
                same::DevEvent inj{};
                inj.channel = ncol; inj.kind = SAME_LINK_SEARCHING; inj.sample_counter = h;
                inj.symbol_count = ts.a_sym + (uint64_t)((double)(h > ts.a_t ? h - ts.a_t : 0) / sps + 0.5);
                inj.burst_slot = 0xffffffffu;
                feed(part, ev, inj, c, 0);
            }
            cur = nxt;
        }
        // the remainder of the call (less than a block), demodulated on the channel's real state afterwards,
        // logs under column c like chunk 0
        for (uint32_t i = first[c]; i < first[c + 1u]; ++i)
            if (evs[order[i]].sample_counter > sl.end_blocks) feed(part, ev, evs[order[i]], c, off);
        rx->tp.sym_off[c] = off;
    };
    auto run_range = [&](uint32_t c0, uint32_t c1, Part &part) {
        part.out.clear(); part.bytes.clear(); part.rearm.clear(); part.bursts.clear();
        part.out.reserve((size_t)(cfirst[c1] - cfirst[c0]) * 3 / 2 + 4);
        same_rx_event ev;                                  // (scratch for what the transport layer returns)
        std::memset(&ev, 0, sizeof(ev));
        for (uint32_t c = c0; c < c1; ++c) {
            // (the next channel's transport state: its hot line is in the cache as a rule -- 2 MB for 32 768 channels -- the burst
            // history and message texts are not: all of them when a burst is on its way there)
            if (!link_only && c + 1u < c1) {
                __builtin_prefetch(&rx->thot[c + 1u]);
                if (!sl.chunked) {
                    bool burst = false;
                    for (uint32_t k = first[c + 1u]; k < first[c + 2u]; ++k) burst |= evs[k].kind == SAME_LINK_BURST;
                    const char *nx = reinterpret_cast<const char *>(&rx->tcold[c + 1u]);
                    if (burst) for (size_t o = 0; o < sizeof(same::TransportCold); o += 64) __builtin_prefetch(nx + o);
                }
            }
            // this channel's column ranges back into log order (see above)
            { HP(0);
            for (uint32_t col = c; col < n_bins; col += n_ch)
                if (first[col + 1u] - first[col] > 1u)
                    std::sort(evs_mut + first[col], evs_mut + first[col + 1u],
                              [](const same::DevEvent &a, const same::DevEvent &b) { return a.channel < b.channel; });
            }
            if (sl.chunked) stitch(part, ev, c);
            else
                for (uint32_t k = first[c]; k < first[c + 1u]; ++k)
                    feed(part, ev, evs[order[k]], c, tp ? rx->tp.sym_off[c] : 0);
            if (tp && !link_only) {
                // polls due before the end of this launch (the next launch's events start after it)
                TickSynth &ts = rx->tp.synth[c];
                if (ts.link == SAME_LINK_NO_CARRIER && sl.end_counter > ts.a_t) {
                    const uint64_t sym_end = ts.a_sym + (uint64_t)((double)(sl.end_counter - ts.a_t) / sps);
                    auto poll = [&](uint64_t psym, uint64_t pt) {
                        if (rx->tr(c).on_link_event(same::kDevTick, pt, psym, nullptr, 0, rx->P.input_rate, &ev)) push_transport(part, ev, c);
                    };
                    ts.run_until(sym_end + 1u, sl.end_counter + 1u, sps, rx->tr(c).force_eom_at(), poll);
                }
            }
        }
    };
    uint32_t n_threads = 1;
    if (n_real >= 16384u && n_ch >= 64u) {
        const unsigned hw = std::thread::hardware_concurrency();
        // up to 32 replay threads, at most an eighth of the host's hardware threads (eight ranks share a node)
        n_threads = std::min<uint32_t>({32u, std::max(16u, hw / 8u), hw ? hw : 1u, n_ch / 32u});
        // (Measured against the container's cgroup CPU quota in round 5 -- the GPU boxes give 16 CPUs of 256 visible: capping the
        // pool at the quota made the headline step host-bound (replay 1.3-1.6 ms on 16 threads, 2.17 ms per step) where 32 threads
        // finish it in 0.8 ms and 2.09 ms per step; a burst of 32 threads for under a millisecond per 2 ms step stays inside a
        // 16-CPU quota on average and is not throttled.)
        if (rx->host_threads > 0) n_threads = (uint32_t)rx->host_threads;
    }
    if (rx->parts.size() < n_threads) rx->parts.resize(n_threads);
    std::vector<Part> &parts = rx->parts;
    for (Part &p : parts) { p.out.clear(); p.bytes.clear(); p.rearm.clear(); p.bursts.clear(); }
    if (n_threads == 1) {
        run_range(0, n_ch, parts[0]);
    } else {
        // split by event count, on channel boundaries
        std::vector<uint32_t> cut(n_threads + 1u, n_ch);
        cut[0] = 0;
        for (uint32_t t = 1; t < n_threads; ++t) {
            const uint32_t target = (uint32_t)((uint64_t)n_real * t / n_threads);
            cut[t] = (uint32_t)(std::lower_bound(cfirst.begin(), cfirst.end(), target) - cfirst.begin());
            cut[t] = std::min(std::max(cut[t], cut[t - 1u]), n_ch);
        }
        rx->workers.run(n_threads, [&](size_t t) { run_range(cut[t], cut[t + 1u], parts[t]); });
    }
    times.replayed = std::chrono::steady_clock::now();
    times.n_threads = n_threads;
    size_t total = 0, total_bytes = 0;
    for (const Part &p : parts) { total += p.out.size(); total_bytes += p.bytes.size(); }
    // a consumer that always polls less than is pending never drains the queue: reclaim the polled
    // prefix once it is at least as large as what is still waiting (amortised O(1) per event)
    rx->peeked_valid = false;                 // (the queue is about to move and grow: a materialised view of it is stale)
    release_stale_view(rx);
    if (rx->queue_head && rx->queue_head >= rx->queue.size() - rx->queue_head) {
        rx->queue_head = rx->queue.compact(rx->queue_head);
        // ... and the payload bytes in front of the first record that is still queued
        size_t keep_from = rx->arena.base + rx->arena.size();
        for (size_t i = 0; i < rx->queue.size(); ++i)
            if (rx->queue.data()[i].n_bytes) { keep_from = (size_t)rx->queue.data()[i].payload; break; }
        rx->arena.compact(keep_from - rx->arena.base);
    }
    QEvent *dst = rx->queue.grow(total);
    if (total && !dst) return fail(SAME_ENOMEM, "event queue");
    uint8_t *bdst = rx->arena.grow(total_bytes);
    if (total_bytes && !bdst) return fail(SAME_ENOMEM, "event payloads");
    {
        // every thread's slice lands at its prefix offset; the copies run side by side
        size_t at = 0, bat = 0;
        if (rx->burst_seq_head > 4096 && rx->burst_seq_head * 2 > rx->burst_seq.size()) {
            rx->burst_seq.erase(rx->burst_seq.begin(), rx->burst_seq.begin() + (std::ptrdiff_t)rx->burst_seq_head);
            rx->burst_seq_head = 0;
        }
        const size_t seq0 = rx->queue.base + (size_t)(dst - rx->queue.data());
        const size_t byte0 = rx->arena.base + (size_t)(bdst - rx->arena.data());
        struct Job { QEvent *q; uint8_t *b; Part *p; size_t byte_off; };
        std::vector<Job> jobs;
        for (size_t t = 0; t < parts.size(); ++t) {
            Part &p = parts[t];
            if (p.out.empty()) continue;
            for (uint32_t i : p.bursts) rx->burst_seq.push_back(seq0 + at + i);
            jobs.push_back(Job{dst + at, bdst + bat, &p, byte0 + bat});
            at += p.out.size(); bat += p.bytes.size();
        }
        rx->workers.run(jobs.size(), [&](size_t j) {
            const Job &job = jobs[j];
            for (QEvent &q : job.p->out) q.payload += job.byte_off;            // part-relative -> absolute
            std::memcpy(job.q, job.p->out.data(), job.p->out.size() * sizeof(QEvent));
            if (!job.p->bytes.empty()) std::memcpy(job.b, job.p->bytes.data(), job.p->bytes.size());
        });
    }
    for (Part &p : parts) rearm.insert(rearm.end(), p.rearm.begin(), p.rearm.end());
    return SAME_OK;
}

int record_harvest(same_batch *rx, same_batch::Slot &sl, uint32_t n_events, uint32_t n_bursts, const char *path);

// Collect the finished launch (device half): wait for it, copy its ordered log, burst pool and geometry back
int harvest_slot(same_batch *rx, same_batch::Slot &sl)
{
    if (!sl.in_flight) return SAME_OK;
    const bool dbg = rx->debug;
    auto t_begin = std::chrono::steady_clock::now();
    // wait for THIS launch only (its cursors have landed in pinned memory); a later launch
    // may still be running on the compute stream
    HIP_TRY(hipEventSynchronize(sl.ev_done));
    sl.in_flight = false;
    if (sl.timed) {
        HIP_TRY(hipEventElapsedTime(&rx->last_ms, sl.ev_start, sl.ev_stop));
        rx->last_demod_ms = rx->last_ms;
        if (sl.have_k) HIP_TRY(hipEventElapsedTime(&rx->last_demod_ms, sl.ev_k0, sl.ev_k1));
        rx->have_timing = true;
    }
    auto t_waited = std::chrono::steady_clock::now();
    const uint32_t n_events = std::min(sl.h_counters[0], sl.event_cap);
    if (dbg)
        std::fprintf(stderr, "[same] harvest: %u device events (%u bursts), cap %u/%u\n", sl.h_counters[0],
                     sl.h_counters[1], sl.event_cap, sl.burst_cap);
    const uint32_t n_bursts = std::min(sl.h_counters[1], sl.burst_cap);
    if (sl.h_counters[2] & 3u) rx->overflowed = true;
    if (sl.h_counters[2] & 4u) rx->kernel_fault = true;
    const size_t ev_bytes = (size_t)n_events * sizeof(same::DevEvent), bu_bytes = (size_t)n_bursts * same::kBurstCap;
    auto grow = [](void **p, size_t *have, size_t need) -> hipError_t {
        if (need <= *have) return hipSuccess;
        if (*p) { (void)hipHostFree(*p); *p = nullptr; *have = 0; }
        const size_t want = need + need / 2 + 4096;
        hipError_t e = hipHostMalloc(p, want, hipHostMallocDefault);
        if (e == hipSuccess) *have = want;
        return e;
    };
    HIP_TRY(grow(&sl.h_events, &sl.h_events_bytes, ev_bytes));
    HIP_TRY(grow(&sl.h_bursts, &sl.h_bursts_bytes, bu_bytes));
    // the event log first: it is what the sort below needs, and the sort runs while the burst pool (the larger copy) and
    // the chunk geometry are still on their way
    const uint32_t n_ch = rx->P.n_channels;
    const uint32_t n_bins = sl.chunked ? sl.geom.n_chunks * n_ch : n_ch;     // state columns of the launch
    if (n_bins != sl.sort_bins) return fail(SAME_EINVAL, "internal: event sort made for %u columns, launch has %u", sl.sort_bins, n_bins);
    // (the log arrives ordered by column, with the columns' offsets: the device did that behind the launch; at most
    // n_events records are real, the exact count is the last offset)
    HIP_TRY(hipMemcpyAsync(sl.h_sort, sl.d_sort + n_bins, ((size_t)n_bins + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, rx->copy_stream));
    if (n_events) HIP_TRY(hipMemcpyAsync(sl.h_events, sl.d_sorted, ev_bytes, hipMemcpyDeviceToHost, rx->copy_stream));
    HIP_TRY(hipStreamSynchronize(rx->copy_stream));
    if (n_bursts) HIP_TRY(hipMemcpyAsync(sl.h_bursts, sl.d_bursts, bu_bytes, hipMemcpyDeviceToHost, rx->copy_stream));
    if (sl.chunked) {
        HIP_TRY(hipMemcpyAsync(sl.h_handover, sl.d_handover, (size_t)n_bins * sizeof(uint64_t), hipMemcpyDeviceToHost, rx->copy_stream));
        if (sl.per_channel)
            HIP_TRY(hipMemcpyAsync(sl.h_geom, sl.d_geom, (size_t)2 * n_bins * sizeof(uint32_t), hipMemcpyDeviceToHost, rx->copy_stream));
    }
    if (n_bursts || sl.chunked) HIP_TRY(hipStreamSynchronize(rx->copy_stream));      // bursts, hand-overs, geometry
    auto t_copied = std::chrono::steady_clock::now();
    if (!rx->record_path.empty()) { const int rrc = record_harvest(rx, sl, n_events, n_bursts, rx->record_path.c_str()); if (rrc != SAME_OK) return rrc; }
    std::vector<uint32_t> rearm;     // channels whose forced-EOM instant changed
    HarvestTimes times;
    int rc = harvest_host(rx, sl, n_events, n_bursts, rearm, times);
    if (rc) return rc;
    // force_eom_at_sample (receiver.rs:321-328) lives on the host; tell the device when to
    // wake the transport layer for it.  Launches are capped well below the 135 s timeout,
    // so the instant is always armed before the device reaches it.
    if (!rearm.empty()) {
        // one upload of the whole wake table (a device word the kernel already cleared is
        // re-armed at worst to an instant in the past: one harmless extra poll)
        if (!rx->h_wake) {
            HIP_TRY(hipHostMalloc((void **)&rx->h_wake, (size_t)rx->P.n_channels * sizeof(uint64_t), hipHostMallocDefault));
            std::memset(rx->h_wake, 0, (size_t)rx->P.n_channels * sizeof(uint64_t));
        }
        for (uint32_t c : rearm) rx->h_wake[c] = rx->tr(c).force_eom_at();
        // the kernels only read this table (it is host-owned), so it may be updated while a
        // later launch runs; launches are capped at 45 s, two launches < the 135 s timeout
        HIP_TRY(hipMemcpyAsync(rx->S.wake_sample, rx->h_wake, (size_t)rx->P.n_channels * sizeof(uint64_t), hipMemcpyHostToDevice, rx->copy_stream));
        HIP_TRY(hipStreamSynchronize(rx->copy_stream));
    }
    if (dbg) {
        auto t_end = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[same] harvest timing: wait %.2f ms, copy %.2f ms, sort %.2f ms, replay on %u threads %.2f ms, "
                             "queue %.2f ms\n",
                     ms(t_begin, t_waited), ms(t_waited, t_copied), ms(t_copied, times.sorted), times.n_threads,
                     ms(times.sorted, times.replayed), ms(times.replayed, t_end));
    }
    return SAME_OK;
}

// harvest every launch still in flight, oldest first
int harvest(same_batch *rx)
{
    same_batch::Slot *order[2] = {&rx->slot[0], &rx->slot[1]};
    if (order[0]->seq > order[1]->seq) std::swap(order[0], order[1]);
    for (same_batch::Slot *sl : order) {
        int rc = harvest_slot(rx, *sl);
        if (rc) return rc;
    }
    return SAME_OK;
}

int harvest(same_batch *rx);

// ---- a harvest on file (tools/host_step_probe.py --ranks N: the host half of a step without a device) ----------------
// SAME_RECORD_HARVEST=<path>: the third harvest of a batch writes what the host half reads -- the ordered log, the columns'
// offsets, the burst pool, hand-over instants and chunk geometry -- to <path>.
struct HarvestFileHeader {
    uint64_t magic;                  // "SAMEHRV1"
    uint32_t n_channels, n_bins, n_events, n_bursts, chunked, per_channel, flags, input_rate, tp_enabled, pad;
    same::ChunkGeom geom;
    uint64_t end_blocks, end_counter;
};
constexpr uint64_t kHarvestMagic = 0x3156524845'4d4153ull;
int record_harvest(same_batch *rx, same_batch::Slot &sl, uint32_t n_events, uint32_t n_bursts, const char *path)
{
    if (sl.seq != 3u) return SAME_OK;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return fail(SAME_EINVAL, "SAME_RECORD_HARVEST: cannot write %s", path);
    const uint32_t n_ch = rx->P.n_channels, n_bins = sl.chunked ? sl.geom.n_chunks * n_ch : n_ch;
    HarvestFileHeader h{};
    h.magic = kHarvestMagic;
    h.n_channels = n_ch; h.n_bins = n_bins; h.n_events = n_events; h.n_bursts = n_bursts;
    h.chunked = sl.chunked; h.per_channel = sl.per_channel; h.flags = rx->flags; h.input_rate = rx->P.input_rate; h.tp_enabled = rx->tp.enabled;
    h.geom = sl.geom; h.end_blocks = sl.end_blocks; h.end_counter = sl.end_counter;
    bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && std::fwrite(sl.h_sort, sizeof(uint32_t), (size_t)n_bins + 1, f) == (size_t)n_bins + 1;
    ok = ok && (!n_events || std::fwrite(sl.h_events, sizeof(same::DevEvent), n_events, f) == n_events);
    ok = ok && (!n_bursts || std::fwrite(sl.h_bursts, same::kBurstCap, n_bursts, f) == n_bursts);
    if (sl.chunked) {
        ok = ok && std::fwrite(sl.h_handover, sizeof(uint64_t), n_bins, f) == n_bins;
        if (sl.per_channel) ok = ok && std::fwrite(sl.h_geom, sizeof(uint32_t), (size_t)2 * n_bins, f) == (size_t)2 * n_bins;
    }
    std::fclose(f);
    return ok ? SAME_OK : fail(SAME_EINVAL, "SAME_RECORD_HARVEST: short write to %s", path);
}

// How a call of n samples is cut into time-parallel chunks: fills geom / pc and returns the number of
// chunks, or 1 when the call runs as one strict launch (mode off, configuration without a pipeline kernel,
// call too short).
uint32_t plan_chunks(same_batch *rx, size_t n, same::ChunkGeom &geom, same::PipeChunks &pc, uint32_t column_cap = 32768u, bool channel_major = false)
{
    same_batch::TimePar &tp = rx->tp;
    if (!tp.enabled || !rx->use_fast || rx->force_generic) return 1;
    const uint32_t C = rx->P.n_channels;
    const double sps = (double)rx->P.input_rate / 520.83;
    auto fill = [&](uint32_t K, uint32_t fb) -> bool {
        const uint32_t warm = tp.warmup ? tp.warmup : (uint32_t)(64.0 * sps + 0.5);
        const uint32_t WB = (warm + fb - 1u) / fb;
        const uint64_t TB = n / fb;
        if (TB <= WB) return false;
        const uint64_t SB = (TB - WB) / K;
        const uint64_t min_own = tp.min_own ? tp.min_own : 4ull * WB * fb;
        if (SB == 0 || SB * fb < min_own) return false;
        geom.counter0 = rx->counter;
        geom.n_chunks = K; geom.block_len = fb; geom.stride_blocks = (uint32_t)SB; geom.warmup_blocks = WB;
        pc.n_chunks = K; pc.in_channels = C; pc.stride_blocks = (uint32_t)SB; pc.nominal_blocks = (uint32_t)SB + WB;
        pc.handover = nullptr;
        return true;
    };
    // Relaxed batches: the pipeline's FASTMATH build while the state columns fit the pipeline (whole 64-channel
    // workgroups), the one-wavefront relaxed kernel beyond (SAME_TP_KERNEL=pipe / wave overrides)
    const bool pipe_fm = rx->relaxed && tp.knob_kernel != 2 && C % same::kWave == 0u && C <= 16384u;
    // A batch that fills the machine by itself (more than 16 384 channels, i.e. more than one workgroup per CU before any
    // cut) gains nothing from a cut in time -- its ordinary relaxed launches run the symbol-paced pipeline at ~30 % of HBM,
    // 262 144 state columns on the one-wavefront kernel ran at 13 % (round 3's `scaled_long`): such time-major calls are not
    // cut, and same_batch_new has made them relaxed launches (32 768 ch x 10 s: 21 ms per call against 31.6; SAME_TP_KERNEL=wave
    // still cuts them).  A channel-major call keeps the cut: read where it lies by the one-wavefront kernel it is 31.6 ms,
    // transposed slab by slab first 38.6.
    if (rx->relaxed && !pipe_fm && !channel_major && tp.knob_kernel == 0 && C % same::kWave == 0u && same::sym_kernel_supported(fm_params(rx->P))) return 1;
    if (rx->relaxed && !pipe_fm && tp.knob_kernel != 1 && C % same::kWave == 0u && C <= 65536u && same::relaxed_kernel_supported(rx->P)) {
        // one wavefront per 64 state columns, any number of them
        // (up to 262 144 state columns, 16 pieces per channel unless the caller asks for more: a piece is a burst with its
        // margins at least, so more only sit empty)
        (void)column_cap;
        const uint32_t k_cap = std::min(63u, 262144u / C);
        const uint32_t k_max = tp.max_chunks ? std::min(tp.max_chunks, k_cap) : std::min(k_cap, 16u);
        same::Params Pk = rx->P;
        for (uint32_t K = k_max; K >= 2u; --K) {
            Pk.n_channels = K * C;                           // (the block length follows the form the column count selects)
            if (fill(K, same::relaxed_block_len(Pk))) { tp.kernel = same_batch::TimePar::kWaveRelaxed; return K; }
        }
        return 1;
    }
    if (C % 16u != 0u || C > 16384u) return 1;
    same::Params Pv = rx->P;
    Pv.ticks = 0; Pv.trace_cap = 0;
    if (pipe_fm) { Pv.knob_pipe_lanes = 64; Pv.knob_pipe_share = 1; Pv.knob_pipe_split = 1; Pv.knob_pipe = 1; }
    // state columns the pipeline takes at full speed: 32 768 at 22.05 kHz (two workgroups per CU), 16 384 at
    // 44.1 / 48 kHz (their window ring leaves room for one)
    // (column_cap 65 536: the channel-major path, whose workgroups are composed of pieces of similar length and may come
    // in two rounds)
    const uint32_t k_cap = (rx->P.ntaps == 42u ? column_cap : 16384u) / C;
    if (column_cap > 32768u) Pv.knob_pipe = 1;           // the pipeline kernel whatever the column count
    uint32_t k_max = tp.max_chunks ? std::min(tp.max_chunks, k_cap) : k_cap;
    for (uint32_t K = k_max; K >= 2u; --K) {
        Pv.n_channels = K * C;
        if (!same::pipe_kernel_selected(Pv) || C % same::pipe_workgroup_channels(Pv) != 0u) continue;
        if (pipe_fm && !same::pipe_relaxed_supported(Pv)) continue;
        const uint32_t fb = pipe_fm ? fm_block_len(Pv) : same::pipe_block_len(Pv);
        if (fill(K, fb)) { tp.kernel = pipe_fm ? same_batch::TimePar::kPipeRelaxed : same_batch::TimePar::kPipe; return K; }
    }
    return 1;
}

// the hand-over records of a time-parallel launch: one per state column, device + pinned host copy
int ensure_handover(same_batch::Slot &sl, uint32_t columns)
{
    if (sl.handover_cap >= columns) return SAME_OK;
    if (sl.d_handover) HIP_TRY(hipFree(sl.d_handover));
    if (sl.h_handover) HIP_TRY(hipHostFree(sl.h_handover));
    sl.d_handover = nullptr; sl.h_handover = nullptr; sl.handover_cap = 0;
    HIP_TRY(hipMalloc((void **)&sl.d_handover, (size_t)columns * sizeof(uint64_t)));
    HIP_TRY(hipHostMalloc((void **)&sl.h_handover, (size_t)columns * sizeof(uint64_t), hipHostMallocDefault));
    sl.handover_cap = columns;
    return SAME_OK;
}

// the wide state blob: `columns` state columns laid out like the channel state, plus the copy tables
int ensure_wide_state(same_batch *rx, uint32_t columns)
{
    same_batch::TimePar &tp = rx->tp;
    if (tp.carved_columns == columns && tp.carved_kernel == (int)tp.kernel) return SAME_OK;
    int rc = harvest(rx);                    // nothing in flight may still use the old layout
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    tp.Pv = rx->P;
    tp.Pv.n_channels = columns;
    tp.Pv.ticks = 0; tp.Pv.trace_cap = 0;
    if (columns > 32768u) tp.Pv.knob_pipe = 1;
    if (tp.kernel == same_batch::TimePar::kPipeRelaxed) { tp.Pv.knob_pipe_lanes = 64; tp.Pv.knob_pipe_share = 1; tp.Pv.knob_pipe_split = 1; tp.Pv.knob_pipe = 1; }
    if (columns > tp.cap_columns) {
        if (tp.blob) HIP_TRY(hipFree(tp.blob));
        if (tp.blob_fresh) HIP_TRY(hipFree(tp.blob_fresh));
        tp.blob = nullptr; tp.blob_fresh = nullptr; tp.cap_columns = 0;
        const size_t bytes = carve_state(tp.Pv, nullptr, tp.Sv);
        HIP_TRY(hipMalloc(&tp.blob, bytes));
        HIP_TRY(hipMemset(tp.blob, 0, bytes));
        HIP_TRY(hipMalloc(&tp.blob_fresh, bytes));
        tp.cap_columns = columns;
    }
    const size_t blob_bytes = carve_state(tp.Pv, (char *)tp.blob, tp.Sv);
    {
        // The template of a launch's fresh state columns (launch_tp_prologue): this layout with SameReceiver::from's state in
        // every column (receiver.rs:539-558), made once per layout by the kernel that used to run over the columns of every launch
        same::State Sf{};
        carve_state(tp.Pv, (char *)tp.blob_fresh, Sf);
        HIP_TRY(hipMemset(tp.blob_fresh, 0, blob_bytes));
        HIP_TRY(same::launch_init_state(tp.Pv, Sf, 0, nullptr, 0));
        HIP_TRY(hipDeviceSynchronize());
        tp.fresh_bytes = (size_t)(reinterpret_cast<char *>(tp.Sv.fr_msg) - reinterpret_cast<char *>(tp.blob));      // (carved in 256-byte steps)
    }
    std::vector<same::StateArrayDesc> in, out;
    list_state_arrays(rx->P, rx->S, tp.Sv, in);
    list_state_arrays(rx->P, tp.Sv, rx->S, out);
    tp.n_desc = (uint32_t)in.size();
    if (!tp.d_desc_in) {
        HIP_TRY(hipMalloc((void **)&tp.d_desc_in, in.size() * sizeof(same::StateArrayDesc)));
        HIP_TRY(hipMalloc((void **)&tp.d_desc_out, out.size() * sizeof(same::StateArrayDesc)));
        HIP_TRY(hipMalloc((void **)&tp.d_final_col, (size_t)rx->P.n_channels * sizeof(uint32_t)));
    }
    HIP_TRY(hipMemcpy(tp.d_desc_in, in.data(), in.size() * sizeof(same::StateArrayDesc), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(tp.d_desc_out, out.data(), out.size() * sizeof(same::StateArrayDesc), hipMemcpyHostToDevice));
    tp.carved_columns = columns;
    tp.carved_kernel = (int)tp.kernel;
    return SAME_OK;
}

template <typename SampleT>
int process_time_major_launches(same_batch *rx, const SampleT *d_x, size_t n_samples, hipStream_t stream)
{
    // keep one launch comfortably inside u32 sample indices and bounded output pools, and
    // well under the 135 s forced-EOM timeout the host arms one launch late (harvest)
    const size_t kMaxChunk = std::min<size_t>((size_t)1 << 22, (size_t)rx->P.input_rate * 45);
    size_t done = 0;
    while (done < n_samples) {
        const size_t n = std::min(kMaxChunk, n_samples - done);
        same_batch::Slot &sl = rx->slot[rx->launch_seq & 1];
        same_batch::Slot &prev = rx->slot[(rx->launch_seq & 1) ^ 1];
        int rc = harvest_slot(rx, sl);          // its buffers are about to be reused
        if (rc) return rc;
        // a launch continues the state the previous one leaves: on another stream than that one, wait for it
        if (prev.in_flight && rx->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, prev.ev_done, 0));
        same::ChunkGeom geom{};
        same::PipeChunks pc{};
        const uint32_t n_chunks = plan_chunks(rx, n, geom, pc);
        rx->tp.last_chunks = n_chunks;
        sl.have_k = false;
        bool sort_bins_empty = false;          // the launch's prologue kernel has emptied the event sort's bins
        same::Output O{};
        if (n_chunks > 1u) {
            // Time-parallel launch: every channel as n_chunks state columns side by side.
            const uint32_t C = rx->P.n_channels, columns = n_chunks * C;
            rc = ensure_wide_state(rx, columns);
            if (rc) return rc;
            const size_t fbk = geom.block_len;
            const size_t per_column = std::min<size_t>(n, 3 * (size_t)pc.nominal_blocks * fbk + 65536);
            rc = ensure_output(rx, sl, per_column, O, columns);
            if (rc) return rc;
            rc = ensure_handover(sl, columns);
            if (rc) return rc;
            pc.handover = sl.d_handover;
            same_batch::TimePar &tp = rx->tp;
            sl.timed = rx->timing; if (sl.timed) HIP_TRY(hipEventRecord(sl.ev_start, stream));
            // fresh receivers in every column (with the hand-over records, the event sort's bins and the launch cursors: one
            // kernel), then the channels' own state into chunk 0's columns
            HIP_TRY(same::launch_tp_prologue(tp.blob, tp.blob_fresh, tp.fresh_bytes, sl.d_handover, sl.d_sort, columns, sl.d_counters, stream));
            HIP_TRY(same::launch_copy_state_columns(tp.d_desc_in, tp.n_desc, C, columns, nullptr, C, stream));
            sort_bins_empty = sl.sort_bins == columns;
            const SampleT *xp = d_x + done * C;
            const uint32_t total_blocks = (uint32_t)(n / fbk);
            hipError_t e;
            if (tp.kernel == same_batch::TimePar::kPipeRelaxed)
                e = launch_fm(tp.Pv, tp.Sv, O, rx->d_taps, xp, total_blocks, rx->counter, stream, pc);
            else if constexpr (sizeof(SampleT) == 4)
                e = tp.kernel == same_batch::TimePar::kWaveRelaxed
                        ? same::launch_demod_relaxed(tp.Pv, tp.Sv, O, rx->d_taps, (const float *)xp, total_blocks, rx->counter, stream, pc)
                        : same::launch_demod_pipe(tp.Pv, tp.Sv, O, rx->d_taps, (const float *)xp, total_blocks, rx->counter, stream, pc, false);
            else
                e = tp.kernel == same_batch::TimePar::kWaveRelaxed
                        ? same::launch_demod_relaxed_i16(tp.Pv, tp.Sv, O, rx->d_taps, (const int16_t *)xp, total_blocks, rx->counter, stream, pc)
                        : same::launch_demod_pipe_i16(tp.Pv, tp.Sv, O, rx->d_taps, (const int16_t *)xp, total_blocks, rx->counter, stream, pc, false);
            if (e != hipSuccess) return fail(SAME_EHIP, "time-parallel demod kernel launch failed: %s", hipGetErrorString(e));
            // the channel's state afterwards is that of the chunk which ran to the end of the input
            HIP_TRY(same::launch_chunk_final_column(sl.d_handover, C, geom, tp.d_final_col, stream));
            HIP_TRY(same::launch_copy_state_columns(tp.d_desc_out, tp.n_desc, columns, C, tp.d_final_col, C, stream));
            const size_t n_whole = (size_t)total_blocks * fbk;
            if (n_whole < n) {
                // less than a block is left: the any-configuration kernel, on the channels' own state
                const SampleT *xr = xp + n_whole * C;
                if constexpr (sizeof(SampleT) == 4)
                    e = same::launch_demod(rx->P, rx->S, O, rx->d_taps, (const float *)xr, (uint32_t)(n - n_whole), rx->counter + n_whole, stream);
                else
                    e = same::launch_demod_i16(rx->P, rx->S, O, rx->d_taps, (const int16_t *)xr, (uint32_t)(n - n_whole), rx->counter + n_whole, stream);
                if (e != hipSuccess) return fail(SAME_EHIP, "demod kernel launch failed: %s", hipGetErrorString(e));
            }
            sl.chunked = true; sl.per_channel = false;
            sl.geom = geom;
            sl.end_blocks = rx->counter + n_whole;
            rx->tp.last_per_channel = false;
        } else {
        sl.chunked = false; sl.per_channel = false;
        rc = ensure_output(rx, sl, n, O);
        if (rc) return rc;
        HIP_TRY(same::launch_counters(sl.d_counters, sl.h_counters_dev, 0, stream));
        sl.timed = rx->timing; if (sl.timed) HIP_TRY(hipEventRecord(sl.ev_start, stream));
        const SampleT *xp = d_x + done * rx->P.n_channels;
        hipError_t e = hipSuccess;
        // whole blocks (16 or 18 samples) go to the latency-optimised kernel when the
        // configuration has one; the generic kernel takes the remainder (and every other
        // configuration)
        // SAME_BATCH_RELAXED on an ordinary launch of whole 64-channel groups: the symbol-paced pipeline at 22.05, 44.1 and 48 kHz
        // (any number of channels; SAME_SYM=0 puts the pipeline's FASTMATH build in its place -- round 5's relaxed kernel at 44.1 /
        // 48 kHz, up to 32 768 channels); the one- / two-wavefront relaxed kernel takes batches that are not whole groups of 64 and
        // whatever SAME_RELAXED_KERNEL=solo / duo sends it
        const same::Params Pfm = fm_params(rx->P);
        // (the symbol-paced pipeline takes any number of 64-channel workgroups: beyond two per CU they run in rounds -- and
        // it is the faster kernel at every channel count)
        // (at 44.1 / 48 kHz a CU holds one group of 64 columns: beyond 16 384 columns the workgroups run in rounds -- round 4 sent
        // such batches to the strict kernels whatever the flag said)
        const bool plain_fm = rx->relaxed_plain && rx->P.knob_relaxed_kernel == 0 && same::pipe_relaxed_supported(Pfm) &&
                              (rx->P.n_channels <= 32768u || (rx->P.n_channels <= rx->sym_max_channels && same::sym_kernel_supported(Pfm)) ||
                               !same::relaxed_kernel_supported(rx->P));
        // (measured, 2 s launches back to back with the transport layer on, the way a stream is fed: 49 152 channels 4.26 ms;
        // 98 304: 7.2 ms against the one-wavefront relaxed kernel's 12.3; 131 072: 9.45 against 15.35; 196 608: 13.95 against
        // 24.9; 262 144: 19.5 against 29.3 -- 30 % of HBM against 18-20 %.  Round 3's figures for the one-wavefront kernel, up to
        // 26 %, were single launches on an idle machine with the link layer only; sustained, its eight wavefronts per SIMD fall
        // back launch by launch: tools/big_sustained.py.  SAME_RELAXED_KERNEL=solo / duo still selects it.)
        const bool plain_wave = rx->relaxed_plain && !plain_fm && same::relaxed_kernel_supported(rx->P);
        rx->last_plain_fm = plain_fm; rx->last_plain_wave = plain_wave;
        const size_t fb = plain_fm ? fm_block_len(Pfm)
                                   : (plain_wave ? same::relaxed_block_len(rx->P) : (rx->use_fast ? same::fast_block_len(rx->P) : 16));
        size_t n_fast = (rx->use_fast && !rx->force_generic) ? (n / fb) * fb : 0;
        if (n_fast && plain_fm) {
            e = launch_fm(Pfm, rx->S, O, rx->d_taps, xp, (uint32_t)(n_fast / fb), rx->counter, stream, same::PipeChunks{});
            rx->last_fm_sym = same::sym_kernel_supported(Pfm);
            if (e != hipSuccess) return fail(SAME_EHIP, "relaxed pipeline launch failed: %s", hipGetErrorString(e));
        } else if (n_fast && plain_wave) {
            if constexpr (sizeof(SampleT) == 4)
                e = same::launch_demod_relaxed(rx->P, rx->S, O, rx->d_taps, (const float *)xp, (uint32_t)(n_fast / fb), rx->counter, stream);
            else
                e = same::launch_demod_relaxed_i16(rx->P, rx->S, O, rx->d_taps, (const int16_t *)xp, (uint32_t)(n_fast / fb), rx->counter, stream);
            if (e != hipSuccess) return fail(SAME_EHIP, "relaxed demod kernel launch failed: %s", hipGetErrorString(e));
        } else if (n_fast) {
            const bool pipe = same::pipe_kernel_selected(rx->P);
            if constexpr (sizeof(SampleT) == 4)
                e = pipe ? same::launch_demod_pipe(rx->P, rx->S, O, rx->d_taps, (const float *)xp, (uint32_t)(n_fast / fb), rx->counter, stream)
                         : same::launch_demod_fast(rx->P, rx->S, O, rx->d_taps, (const float *)xp, (uint32_t)(n_fast / fb), rx->counter, stream);
            else
                e = pipe ? same::launch_demod_pipe_i16(rx->P, rx->S, O, rx->d_taps, (const int16_t *)xp, (uint32_t)(n_fast / fb), rx->counter, stream)
                         : same::launch_demod_fast_i16(rx->P, rx->S, O, rx->d_taps, (const int16_t *)xp, (uint32_t)(n_fast / fb), rx->counter, stream);
            if (e != hipSuccess) return fail(SAME_EHIP, "fast demod kernel launch failed: %s", hipGetErrorString(e));
        }
        if (n_fast < n) {
            const SampleT *xr = xp + n_fast * rx->P.n_channels;
            if constexpr (sizeof(SampleT) == 4)
                e = same::launch_demod(rx->P, rx->S, O, rx->d_taps, (const float *)xr, (uint32_t)(n - n_fast), rx->counter + n_fast, stream);
            else
                e = same::launch_demod_i16(rx->P, rx->S, O, rx->d_taps, (const int16_t *)xr, (uint32_t)(n - n_fast), rx->counter + n_fast, stream);
            if (e != hipSuccess) return fail(SAME_EHIP, "demod kernel launch failed: %s", hipGetErrorString(e));
        }
        }
        if (sl.timed) HIP_TRY(hipEventRecord(sl.ev_stop, stream));
        HIP_TRY(same::launch_event_sort(sl.d_events, sl.d_counters, sl.event_cap, sl.sort_bins, sl.d_sort, sl.d_sort + sl.sort_bins,
                                        sl.d_sorted, stream, sort_bins_empty));
        HIP_TRY(same::launch_counters(sl.d_counters, sl.h_counters_dev, 1, stream));
        HIP_TRY(hipEventRecord(sl.ev_done, stream));
        sl.in_flight = true;
        sl.seq = ++rx->launch_seq;
        rx->last_stream = stream;
        rx->counter += n;
        sl.end_counter = rx->counter;
        done += n;
        // while this launch runs, bring in the previous one
        rc = harvest_slot(rx, prev);
        if (rc) return rc;
    }
    return SAME_OK;
}

// SAME_BATCH_CALL_INVARIANT (include/same_rx.h): launches cover whole windows of the stream, from fixed stream positions on.  What
// a launch computes is a function of the state it starts from and of the samples it covers -- the relaxed kernels drain their
// pipeline at a launch's end and apply the feedback in flight at once, the time-parallel mode plans its cuts per launch -- so a
// stream fed in different calls came out different near a lock or an end of burst wherever the calls' boundaries fell.  With the
// launches' boundaries tied to the stream instead of to the calls, it cannot: any list of calls that delivers the same samples
// makes the same launches.  The price is latency: the events of a window arrive when its last sample has (same_batch_flush
// demodulates what is waiting).  Whole windows that lie inside a call's buffer are launched where they lie; the pieces before
// and behind them are copied into the waiting buffer (f32; int16 pieces are cast the way the kernels cast them: unscaled).
// Window lengths (same_batch_set_call_window changes them): whole blocks of the kernels that take such launches (16, 18, 32, 36, 72
// samples).  Short windows keep the latency and the share of a call that has to be copied low (a call shorter than a window is
// copied whole: at 32 768 channels a 2-s call costs 4.4 ms with 73 728-sample windows, 3.5 with 18 432, 2.0 without the flag); a
// time-parallel batch needs long ones -- its planner cuts every LAUNCH into pieces of a burst's length and more.
constexpr uint32_t kInvWindow = 18432;                 // 0.84 s at 22.05 kHz, 0.38 s at 48 kHz
constexpr uint32_t kInvWindowTimeParallel = 73728;     // 3.3 s at 22.05 kHz, 1.5 s at 48 kHz
static int inv_buffer_wait(same_batch *rx, hipStream_t stream)
{
    same_batch::Windowed &iv = rx->inv;
    if (iv.buf_used && iv.buf_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, iv.ev_buf, 0));
    return SAME_OK;
}
static int inv_buffer_done(same_batch *rx, hipStream_t stream)
{
    same_batch::Windowed &iv = rx->inv;
    if (!iv.ev_buf) HIP_TRY(hipEventCreateWithFlags(&iv.ev_buf, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(iv.ev_buf, stream));
    iv.buf_stream = stream; iv.buf_used = true;
    return SAME_OK;
}
template <typename SampleT>
static int inv_append(same_batch *rx, const SampleT *d_x, size_t n, hipStream_t stream)
{
    // n samples (rows of n_channels) behind the ones already waiting
    same_batch::Windowed &iv = rx->inv;
    const size_t C = rx->P.n_channels;
    int rc = ensure_stage(&iv.d_buf, &iv.buf_bytes, (size_t)iv.window * C * sizeof(float));
    if (rc) return rc;
    rc = inv_buffer_wait(rx, stream);
    if (rc) return rc;
    float *dst = static_cast<float *>(iv.d_buf) + (size_t)iv.fill * C;
    if constexpr (sizeof(SampleT) == 4) {
        HIP_TRY(hipMemcpyAsync(dst, d_x, n * C * sizeof(float), hipMemcpyDeviceToDevice, stream));
    } else {
        hipError_t e = same::launch_cast_i16_f32((const int16_t *)d_x, dst, n * C, stream);
        if (e != hipSuccess) return fail(SAME_EHIP, "cast launch failed: %s", hipGetErrorString(e));
    }
    iv.fill += (uint32_t)n;
    return inv_buffer_done(rx, stream);
}
// demodulate what is waiting (a whole window, or -- flush -- whatever there is) and begin a new window
static int inv_launch_waiting(same_batch *rx, hipStream_t stream)
{
    same_batch::Windowed &iv = rx->inv;
    if (!iv.fill) return SAME_OK;
    int rc = inv_buffer_wait(rx, stream);
    if (rc) return rc;
    const uint32_t n = iv.fill;
    iv.fill = 0;
    rc = process_time_major_launches<float>(rx, static_cast<const float *>(iv.d_buf), n, stream);
    if (rc) return rc;
    return inv_buffer_done(rx, stream);
}
template <typename SampleT>
int process_time_major(same_batch *rx, const SampleT *d_x, size_t n_samples, hipStream_t stream)
{
    same_batch::Windowed &iv = rx->inv;
    if (!iv.on) return process_time_major_launches(rx, d_x, n_samples, stream);
    const size_t C = rx->P.n_channels;
    size_t pos = 0;
    if (iv.fill) {
        // complete the window that is waiting
        const size_t take = std::min<size_t>(iv.window - iv.fill, n_samples);
        int rc = inv_append(rx, d_x, take, stream);
        if (rc) return rc;
        pos = take;
        if (iv.fill < iv.window) return SAME_OK;
        rc = inv_launch_waiting(rx, stream);
        if (rc) return rc;
    }
    // whole windows where they lie
    while (n_samples - pos >= iv.window) {
        int rc = process_time_major_launches(rx, d_x + pos * C, iv.window, stream);
        if (rc) return rc;
        pos += iv.window;
    }
    if (pos < n_samples) return inv_append(rx, d_x + pos * C, n_samples - pos, stream);
    return SAME_OK;
}

// Time-parallel launch over a CHANNEL-MAJOR input with chunk boundaries chosen per channel at idle instants
// (DESIGN.md 4.6): an energy scout and the boundary planner run on the device ahead of the demodulation kernel, every
// state column then reads its own contiguous stream from its own first row.  Returns 1 when the call was handled, 0
// when it does not qualify (the caller then takes the transposing path), < 0 on error.
int process_channel_major_native(same_batch *rx, const float *d_x, size_t n, hipStream_t stream)
{
    same_batch::TimePar &tp = rx->tp;
    if (!tp.enabled || n > ((size_t)1 << 22)) return 0;
    // 22.05 kHz only: per-lane input streams are read by the sample stage of the 64-channel pipeline forms and by the
    // relaxed kernels; the 44.1 / 48 kHz pipeline reads its input on the DC wavefront, which takes time-major rows only
    // (such a call is transposed on the device and cut uniformly)
    if (rx->P.ntaps != 42u) return 0;
    same::ChunkGeom geom{};
    same::PipeChunks pc{};
    // A quarter more state columns than the machine holds at once (40 960: 10 pieces per channel at 4 096 channels) unless
    // the caller asks for a number of chunks: the launch is as long as its longest piece, a burst with its margins however
    // many pieces there are, but with 8 pieces the planner cannot give every long burst a piece of its own (longest piece
    // 39.9 k samples, with 10 or more 37.5 k), and beyond 10 the extra workgroups only add rounds.  Measured at 4 096
    // channels x 10 s, pieces sorted by length into workgroups: 8 pieces 3.84 ms (unsorted 3.83), 9 3.47, 10 3.45, 11 3.70,
    // 12 3.85, 16 4.2.
    // With the symbol-paced pipeline (round 4): exactly the columns the machine holds at once (32 768: 8 pieces per channel at
    // 4 096 channels, one round of workgroups in grid order).  Single launches on an idle machine favour 12 pieces (8: 2.16 ms,
    // 10: 2.14, 12: 2.03, 16: 2.21 -- the long pieces finish with a CU to themselves), but calls back to back, the way a stream
    // is fed, do not: 40 steps with 8 pieces 2.20 ms per step (kernel 2.05-2.08, launch to launch +-3 %), with 10: 2.38-2.42,
    // 12: 2.41-2.48 (kernel 2.26-2.35, launch to launch 2.06-3.1), 16: 2.46 (tools/headline_steady.py).
    const bool sym_cols = rx->relaxed && same::sym_kernel_supported(rx->P);
    const uint32_t dflt_cols = sym_cols ? 32768u : 40960u;
    const uint32_t want_cols = tp.max_chunks ? tp.max_chunks * rx->P.n_channels : dflt_cols;
    const uint32_t n_chunks = plan_chunks(rx, n, geom, pc, tp.max_chunks ? (want_cols > 32768u ? 65536u : 32768u) : dflt_cols, true);
    if (n_chunks < 2u) return 0;
    const uint32_t C = rx->P.n_channels, columns = n_chunks * C, fb = geom.block_len;
    const bool wave = tp.kernel == same_batch::TimePar::kWaveRelaxed;
    same::Params Pv = rx->P;
    Pv.n_channels = columns; Pv.ticks = 0; Pv.trace_cap = 0; Pv.knob_pipe = 1;
    if (tp.kernel == same_batch::TimePar::kPipeRelaxed) { Pv.knob_pipe_lanes = 64; Pv.knob_pipe_share = 1; Pv.knob_pipe_split = 1; }
    // 16-byte loads from every lane's stream (the scout's too: 16-byte aligned base and pitch; the relaxed kernel reads
    // 8 bytes at a time from even rows), full 64-column workgroups.  What is left of the call behind its last whole block
    // (less than a block) goes through the any-configuration kernel on the channels' own state afterwards.
    const size_t n_call = n;
    n -= n % fb;
    if (n_call % 4 != 0 || ((uintptr_t)d_x & 15u) != 0u || n < 64u * (size_t)same::tp_scout_block() || n / same::tp_scout_block() > 7000u || n_chunks > 63u || C % same::kWave != 0u) return 0;
    if (wave ? fb % 2 != 0 : (fb % 4 != 0 || same::pipe_workgroup_channels(Pv) != (uint32_t)same::kWave)) return 0;
    same_batch::Slot &sl = rx->slot[rx->launch_seq & 1];
    same_batch::Slot &prev = rx->slot[(rx->launch_seq & 1) ^ 1];
    int rc = harvest_slot(rx, sl);
    if (rc) return rc;
    // a launch continues the state the previous one leaves: on another stream than that one, wait for it
    if (prev.in_flight && rx->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, prev.ev_done, 0));
    rc = ensure_wide_state(rx, columns);
    if (rc) return rc;
    same::Output O{};
    rc = ensure_output(rx, sl, std::min<size_t>(n, 3 * (size_t)pc.nominal_blocks * fb + 65536), O, columns);
    if (rc) return rc;
    rc = ensure_handover(sl, columns);
    if (rc) return rc;
    if (sl.geom_cap < columns) {
        if (sl.d_geom) HIP_TRY(hipFree(sl.d_geom));
        if (sl.h_geom) HIP_TRY(hipHostFree(sl.h_geom));
        sl.d_geom = nullptr; sl.h_geom = nullptr; sl.geom_cap = 0;
        HIP_TRY(hipMalloc((void **)&sl.d_geom, ((size_t)5 * columns + 3 * (columns / same::kWave)) * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc((void **)&sl.h_geom, (size_t)2 * columns * sizeof(uint32_t), hipHostMallocDefault));
        sl.geom_cap = columns;
    }
    same::TpPlan plan{};
    plan.channels = C; plan.n_chunks = n_chunks; plan.block_len = fb;
    // Warm-up: a per-channel boundary lies in silence, so all a fresh receiver needs before the next burst begins is a
    // full squelch history (32 symbols; the preamble that follows is 128): 40 symbols unless the caller set one
    {
        const double sps = (double)rx->P.input_rate / 520.83;
        const uint32_t warm = tp.warmup ? tp.warmup : (uint32_t)(40.0 * sps + 0.5);
        plan.warmup_samples = std::min(geom.warmup_blocks, (warm + fb - 1u) / fb) * fb;
    }
    plan.whole_samples = (uint32_t)n; plan.in_samples = n_call;
    plan.scout_blocks = (uint32_t)(n / same::tp_scout_block());
    const size_t e_need = (size_t)C * plan.scout_blocks;
    if (tp.energy_cap < e_need) {
        if (tp.d_energy) HIP_TRY(hipFree(tp.d_energy));
        tp.d_energy = nullptr; tp.energy_cap = 0;
        HIP_TRY(hipMalloc((void **)&tp.d_energy, e_need * sizeof(float)));
        tp.energy_cap = e_need;
    }
    uint32_t *d_own = sl.d_geom, *d_row0 = sl.d_geom + columns, *d_nom = sl.d_geom + 2 * (size_t)columns,
             *d_perm = sl.d_geom + 3 * (size_t)columns, *d_wg = sl.d_geom + 4 * (size_t)columns,
             *d_wg2 = d_wg + columns / same::kWave, *d_perm2 = d_wg2 + 2 * (columns / same::kWave);      // (d_wg2: [2][workgroups])
    // pieces sorted by length into workgroups only when the workgroups come in more than one round (the long ones first;
    // SAME_TP_SORT=0 / 1 overrides).  Within one round neither the sorted order nor a long workgroup beside a short one on
    // every CU pays (round 2).
    // SAME_TP_SORT=2 (measurement knob): grid order inside the groups of 64 columns, the groups paired longest with shortest into
    // the symbol-paced pipeline's two-group workgroups, so that a long group runs the second part of its launch alone on its CU
    // (a step is ~12 % shorter there).  One launch by itself: demodulation kernel 1.746 -> 1.706 ms; launches back to back, the
    // way the bench and a stream step: 1.695 -> 1.732 (the shorter tail hides less of the next call's planning) -- not the default.
    const bool pair_groups = tp.sort_mode == 2 && tp.kernel == same_batch::TimePar::kPipeRelaxed && same::sym_kernel_supported(tp.Pv);
    const int sort_mode = pair_groups ? 0 : (tp.sort_mode >= 0 ? (tp.sort_mode ? 1 : 0) : (columns > 32768u ? 1 : 0));
    sl.timed = rx->timing; if (sl.timed) HIP_TRY(hipEventRecord(sl.ev_start, stream));
    // (sorted: the workgroups once more, longest first whatever their group -- those that wait for a free CU are then the short ones)
    const bool lpt = sort_mode != 0 || pair_groups;
    // On the library's own stream the planning kernels go to the plan stream: they need the input (ordered by
    // same_batch_order_after, which both streams honour) and this slot's geometry buffers (free since the harvest above),
    // not the previous launch -- so they run beside its tail instead of after it (~0.25 ms of small kernels per call at
    // configs[1]).  On a caller's stream everything stays in that stream's order.
    const bool side = stream == rx->own_stream && tp.knob_plan_stream >= 0;
    hipStream_t ps = side ? rx->plan_stream : stream;
    // (the energy map is one buffer for both slots: this call's scout after the previous call's planner, whichever
    // streams the two ran on)
    if (tp.plan_recorded) HIP_TRY(hipStreamWaitEvent(ps, tp.ev_plan_prev, 0));
    HIP_TRY(same::launch_tp_plan(d_x, plan, tp.d_energy, d_own, d_row0, d_nom, d_perm, d_wg, sort_mode != 0, ps,
                                 lpt ? d_perm2 : nullptr, lpt ? d_wg2 : nullptr, pair_groups));
    if (!tp.ev_plan_prev) HIP_TRY(hipEventCreateWithFlags(&tp.ev_plan_prev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(tp.ev_plan_prev, ps));
    tp.plan_recorded = true;
    if (side) {
        HIP_TRY(hipEventRecord(sl.ev_planned, ps));
        HIP_TRY(hipStreamWaitEvent(stream, sl.ev_planned, 0));
    }
    if (lpt) { d_perm = d_perm2; d_wg = d_wg2; }
    // fresh receivers in every column (with the hand-over records, the event sort's bins and the launch cursors: one kernel,
    // launch_tp_prologue), then the channels' own state into chunk 0's columns
    bool sort_bins_empty = false;
    if (tp.knob_prologue >= 0) {
        HIP_TRY(same::launch_tp_prologue(tp.blob, tp.blob_fresh, tp.fresh_bytes, sl.d_handover, sl.d_sort, columns, sl.d_counters, stream));
        sort_bins_empty = sl.sort_bins == columns;
    } else {
        // (SAME_TP_PROLOGUE=0, A/B measurements: the launches' start as rounds 2-5 made it)
        HIP_TRY(same::launch_counters(sl.d_counters, sl.h_counters_dev, 0, stream));
        HIP_TRY(same::launch_init_state(tp.Pv, tp.Sv, 0, stream, C));
        HIP_TRY(same::launch_fill_u64(sl.d_handover, columns, same::kNoHandover, stream));
    }
    HIP_TRY(same::launch_copy_state_columns(tp.d_desc_in, tp.n_desc, C, columns, nullptr, C, stream));
    pc.handover = sl.d_handover;
    pc.col_row0 = d_row0; pc.col_nominal = d_nom; pc.wg_blocks = d_wg; pc.col_perm = (sort_mode != 0 || pair_groups) ? d_perm : nullptr;
    pc.in_samples = n_call; pc.whole_samples = (uint32_t)n;
    pc.hist_scratch = nullptr;
    if (pc.col_perm && tp.kernel == same_batch::TimePar::kPipeRelaxed && same::sym_kernel_supported(tp.Pv)) {
        const size_t h_need = (size_t)columns * same::kSquelchHist;
        if (tp.hist_cap < h_need) {
            if (tp.d_hist) HIP_TRY(hipFree(tp.d_hist));
            tp.d_hist = nullptr; tp.hist_cap = 0;
            HIP_TRY(hipMalloc((void **)&tp.d_hist, h_need * sizeof(float)));
            tp.hist_cap = h_need;
        }
        pc.hist_scratch = tp.d_hist;
    }
    sl.have_k = sl.timed;
    if (sl.timed) HIP_TRY(hipEventRecord(sl.ev_k0, stream));
    hipError_t e = wave ? same::launch_demod_relaxed(tp.Pv, tp.Sv, O, rx->d_taps, d_x, (uint32_t)(n / fb), rx->counter, stream, pc)
                        : (tp.kernel == same_batch::TimePar::kPipeRelaxed ? launch_fm(tp.Pv, tp.Sv, O, rx->d_taps, d_x, (uint32_t)(n / fb), rx->counter, stream, pc)
                                                                          : same::launch_demod_pipe(tp.Pv, tp.Sv, O, rx->d_taps, d_x, (uint32_t)(n / fb), rx->counter, stream, pc, false));
    if (e != hipSuccess) return fail(SAME_EHIP, "time-parallel demod kernel launch failed: %s", hipGetErrorString(e));
    if (sl.timed) HIP_TRY(hipEventRecord(sl.ev_k1, stream));
    // The channels' state afterwards: that of the chunk the hand-over chain ends in, as the host's stitch follows it --
    // the last chunk as a rule (its columns end with the input); an earlier one where a burst ran on to the end of the
    // call without a hand-over (a forced cut on a channel that is never quiet: its events are the ones that are kept).
    HIP_TRY(same::launch_chunk_final_column_pc(sl.d_handover, d_own, C, n_chunks, rx->counter, tp.d_final_col, stream));
    HIP_TRY(same::launch_copy_state_columns(tp.d_desc_out, tp.n_desc, columns, C, tp.d_final_col, C, stream));
    if (n < n_call) {
        // less than a block is left: its [C][r] corner of the input transposed to rows, then the any-configuration kernel
        const size_t r = n_call - n;
        int rc2 = ensure_stage(&rx->d_stage, &rx->stage_bytes, r * C * sizeof(float));
        if (rc2) return rc2;
        rc2 = ensure_stage(&rx->d_stage2, &rx->stage2_bytes, r * C * sizeof(float));
        if (rc2) return rc2;
        HIP_TRY(hipMemcpy2DAsync(rx->d_stage, r * sizeof(float), d_x + n, n_call * sizeof(float), r * sizeof(float), C, hipMemcpyDeviceToDevice, stream));
        hipError_t e2 = same::launch_transpose_f32((const float *)rx->d_stage, (float *)rx->d_stage2, C, (uint32_t)r, stream);
        if (e2 == hipSuccess) e2 = same::launch_demod(rx->P, rx->S, O, rx->d_taps, (const float *)rx->d_stage2, (uint32_t)r, rx->counter + n, stream);
        if (e2 != hipSuccess) return fail(SAME_EHIP, "demod kernel launch failed: %s", hipGetErrorString(e2));
    }
    sl.chunked = true; sl.per_channel = true;
    sl.geom = geom;
    sl.end_blocks = rx->counter + n;
    if (sl.timed) HIP_TRY(hipEventRecord(sl.ev_stop, stream));
    HIP_TRY(same::launch_event_sort(sl.d_events, sl.d_counters, sl.event_cap, sl.sort_bins, sl.d_sort, sl.d_sort + sl.sort_bins,
                                    sl.d_sorted, stream, sort_bins_empty));
    HIP_TRY(same::launch_counters(sl.d_counters, sl.h_counters_dev, 1, stream));
    HIP_TRY(hipEventRecord(sl.ev_done, stream));
    sl.in_flight = true;
    sl.seq = ++rx->launch_seq;
    rx->last_stream = stream;
    rx->counter += n_call;
    sl.end_counter = rx->counter;
    tp.last_chunks = n_chunks; tp.last_per_channel = true;
    rc = harvest_slot(rx, prev);          // while this launch runs, bring in the previous one
    return rc ? rc : 1;
}

template <typename SampleT>
int process_device_any(same_batch *rx, const SampleT *d_x, size_t n_samples, uint32_t layout, void *hip_stream)
{
    if (!rx || (!d_x && n_samples)) return fail(SAME_EINVAL, "null argument");
    if (n_samples == 0) return SAME_OK;
    HIP_TRY(hipSetDevice(rx->device));
    // SAME_STREAM_OWN: the library's private stream; anything else (NULL included) is the caller's stream
    hipStream_t stream = hip_stream == SAME_STREAM_OWN ? rx->own_stream : (hipStream_t)hip_stream;
    if (layout == SAME_LAYOUT_TIME_MAJOR) return process_time_major(rx, d_x, n_samples, stream);
    if (layout != SAME_LAYOUT_CHANNEL_MAJOR) return fail(SAME_EINVAL, "unknown layout %u", layout);
    if constexpr (sizeof(SampleT) == 4) {
        // (its cuts are planned per CALL: a batch that is to be independent of its calls takes the transposing path and windows)
        const int done = rx->inv.on ? 0 : process_channel_major_native(rx, (const float *)d_x, n_samples, stream);
        if (done) return done < 0 ? done : SAME_OK;
    }
    // channel-major: transpose slabs of time through a staging buffer
    // (the staging buffers are the batch's own: a launch still in flight on ANOTHER stream may be reading its input out of
    // them -- successive launches are ordered whatever their streams, so this call's first staging write waits for it)
    for (same_batch::Slot &sl : rx->slot)
        if (sl.in_flight && rx->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, sl.ev_done, 0));
    // (slabs of 65 520 samples: whole blocks of every kernel -- 16, 18, 20, 36 and 42 samples -- so that only the call's own
    // tail goes through the any-configuration kernel, not one per slab)
    const size_t slab = std::min<size_t>(n_samples, (size_t)65520);
    int rc = ensure_stage(&rx->d_stage2, &rx->stage2_bytes, slab * rx->P.n_channels * sizeof(SampleT));
    if (rc) return rc;
    for (size_t t0 = 0; t0 < n_samples; t0 += slab) {
        const size_t n = std::min(slab, n_samples - t0);
        // stream order keeps the previous slab's kernel ahead of this slab's staging writes
        // rows of the source are n_samples long; transpose a [C][n] window starting at t0
        hipError_t e;
        // a strided window is expressed by offsetting the base and keeping the row pitch:
        // the kernel takes a dense [C][n] view, so copy row windows first
        // (2D copy keeps this simple and is only used on the non-native layout)
        rc = ensure_stage(&rx->d_stage, &rx->stage_bytes, n * rx->P.n_channels * sizeof(SampleT));
        if (rc) return rc;
        HIP_TRY(hipMemcpy2DAsync(rx->d_stage, n * sizeof(SampleT), d_x + t0, n_samples * sizeof(SampleT),
                                 n * sizeof(SampleT), rx->P.n_channels, hipMemcpyDeviceToDevice, stream));
        if constexpr (sizeof(SampleT) == 4)
            e = same::launch_transpose_f32((const float *)rx->d_stage, (float *)rx->d_stage2, rx->P.n_channels, (uint32_t)n, stream);
        else
            e = same::launch_transpose_i16((const int16_t *)rx->d_stage, (int16_t *)rx->d_stage2, rx->P.n_channels, (uint32_t)n, stream);
        if (e != hipSuccess) return fail(SAME_EHIP, "transpose launch failed: %s", hipGetErrorString(e));
        rc = process_time_major(rx, (const SampleT *)rx->d_stage2, n, stream);
        if (rc) return rc;
    }
    return SAME_OK;
}

template <typename SampleT>
int process_host_any(same_batch *rx, const SampleT *h_x, size_t n_samples, uint32_t layout)
{
    if (!rx || (!h_x && n_samples)) return fail(SAME_EINVAL, "null argument");
    if (n_samples == 0) return SAME_OK;
    HIP_TRY(hipSetDevice(rx->device));
    // upload in slabs so arbitrarily long host streams need bounded device memory
    const size_t C = rx->P.n_channels;
    const size_t slab = std::max<size_t>(1, std::min<size_t>(n_samples, ((size_t)256 << 20) / (C * sizeof(SampleT))));
    int rc = ensure_stage(&rx->d_upload, &rx->upload_bytes, slab * C * sizeof(SampleT));
    if (rc) return rc;
    void *d_in = rx->d_upload;
    for (size_t t0 = 0; t0 < n_samples && rc == SAME_OK; t0 += slab) {
        const size_t n = std::min(slab, n_samples - t0);
        hipError_t e;
        // (the previous slab's launch was collected below, so the buffer is free again)
        if (layout == SAME_LAYOUT_TIME_MAJOR) {
            e = hipMemcpy(d_in, h_x + t0 * C, n * C * sizeof(SampleT), hipMemcpyHostToDevice);
        } else {
            e = hipMemcpy2D(d_in, n * sizeof(SampleT), h_x + t0, n_samples * sizeof(SampleT),
                            n * sizeof(SampleT), C, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) { rc = fail(SAME_EHIP, "upload failed: %s", hipGetErrorString(e)); break; }
        rc = process_device_any(rx, (const SampleT *)d_in, n, layout, SAME_STREAM_OWN);
        if (rc == SAME_OK) rc = harvest(rx);
    }
    if (rc == SAME_OK && rx->kernel_fault) return fail(SAME_EKERNEL, "a demodulation kernel's wavefronts lost step (internal hand-over timed out)");
    if (rc == SAME_OK && rx->overflowed) return fail(SAME_EOVERFLOW, "event/burst pool overflow");
    return rc;
}

}  // namespace

// ------------------------------------------------------------------------------------
extern "C" {

const char *same_last_error(void) { return g_last_error.c_str(); }

#ifndef SAME_SOURCE_HASH
#define SAME_SOURCE_HASH "unknown"
#endif
// sha256 of the sources this binary was built from (sameold_amd/build.py compares it with the tree)
const char *same_rx_source_hash(void) { return "SAME_SOURCE_HASH=" SAME_SOURCE_HASH; }

int same_batch_new(const same_rx_builder *b, uint32_t n_channels, int device, uint32_t flags,
                   same_batch **out)
{
    if (!b || !out || n_channels == 0) return fail(SAME_EINVAL, "null builder/out or zero channels");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SAME_ENODEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(SAME_ENODEVICE, "device %d out of range (%d visible)", device, ndev);
    same_batch *rx = new (std::nothrow) same_batch;
    if (!rx) return fail(SAME_ENOMEM, "out of memory");
    rx->builder = *b;
    rx->device = device;
    rx->flags = flags;
    rx->inv.on = (flags & SAME_BATCH_CALL_INVARIANT) != 0;
    rx->inv.window = (flags & SAME_BATCH_TIME_PARALLEL) ? kInvWindowTimeParallel : kInvWindow;
    if (const char *e = std::getenv("SAME_INV_WINDOW")) { const long v = std::atol(e); if (v >= 64 && v <= (1 << 22)) rx->inv.window = (uint32_t)v; }      // (tests: short windows)
    std::vector<float> taps;
    int rc = same::derive_params(*b, n_channels, rx->P, taps);
    if (rc) { delete rx; return fail(rc, "builder rejected (code %d)", rc); }
    read_knobs(rx);
    rx->P.trace_cap = (flags & SAME_BATCH_TRACE_SYMBOLS) ? 4096u : 0u;
    if ((flags & SAME_BATCH_TIME_PARALLEL) && (flags & SAME_BATCH_TRACE_SYMBOLS)) {
        delete rx;
        return fail(SAME_EINVAL, "SAME_BATCH_TIME_PARALLEL cannot record a symbol trace");
    }
    rx->tp.enabled = (flags & SAME_BATCH_TIME_PARALLEL) != 0;
    // transport wake-ups come from the device in strict mode, from the host's symbol clock in time-parallel mode
    rx->P.ticks = ((flags & SAME_BATCH_LINK_ONLY) || rx->tp.enabled) ? 0u : 1u;
    rx->P.tick_interburst = (uint32_t)same::max_interburst_symbols();
    rx->P.tick_history = (uint32_t)same::max_history_duration();

    auto cleanup = [&](int code) { same_batch_free(rx); return code; };
#define TRY_OR_CLEAN(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return cleanup(fail(SAME_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)));    \
    } while (0)
    TRY_OR_CLEAN(hipSetDevice(device));
    hipDeviceProp_t prop;
    TRY_OR_CLEAN(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return cleanup(fail(SAME_ENODEVICE, "device %d is %s; this build carries gfx950 code only", device, prop.gcnArchName));
    rx->use_fast = same::fast_kernel_supported(rx->P);
    rx->force_generic = (flags & SAME_BATCH_GENERIC_KERNEL) != 0;
    // Relaxed arithmetic: asked for (SAME_BATCH_RELAXED), or implied by the time-parallel mode, whose contract is the
    // same one (SAME_RELAXED=0 keeps that mode on the strict pipeline kernel; =1 turns it on for any batch)
    rx->relaxed = (((flags & (SAME_BATCH_RELAXED | SAME_BATCH_TIME_PARALLEL)) != 0 && rx->knob_relaxed >= 0) || rx->knob_relaxed > 0) &&
                  rx->use_fast && !rx->force_generic &&
                  (same::relaxed_kernel_supported(rx->P) || (rx->P.n_channels % same::kWave == 0u && same::pipe_relaxed_supported(fm_params(rx->P))));
    // (a call of a time-parallel batch that is too short to be cut stays strict unless relaxed arithmetic was asked for)
    // (... nor a batch of more than 16 384 channels, which is never cut: plan_chunks)
    rx->relaxed_plain = rx->relaxed && ((flags & SAME_BATCH_RELAXED) != 0 || rx->knob_relaxed > 0 ||
                                        ((flags & SAME_BATCH_TIME_PARALLEL) != 0 && n_channels > 16384u && n_channels % same::kWave == 0u &&
                                         same::sym_kernel_supported(fm_params(rx->P))));
    if (same::demod_lds_bytes(rx->P) > 160 * 1024)
        return cleanup(fail(SAME_EINVAL, "configuration needs %zu bytes of LDS per wavefront (limit 160 KiB)", same::demod_lds_bytes(rx->P)));
    TRY_OR_CLEAN(hipStreamCreateWithFlags(&rx->own_stream, hipStreamNonBlocking));
    TRY_OR_CLEAN(hipStreamCreateWithFlags(&rx->copy_stream, hipStreamNonBlocking));
    TRY_OR_CLEAN(hipStreamCreateWithFlags(&rx->plan_stream, hipStreamNonBlocking));
    TRY_OR_CLEAN(hipEventCreateWithFlags(&rx->ev_order, hipEventDisableTiming));
    for (auto &sl : rx->slot) {
        TRY_OR_CLEAN(hipEventCreate(&sl.ev_start));
        TRY_OR_CLEAN(hipEventCreate(&sl.ev_stop));
        TRY_OR_CLEAN(hipEventCreate(&sl.ev_k0));
        TRY_OR_CLEAN(hipEventCreate(&sl.ev_k1));
        TRY_OR_CLEAN(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        TRY_OR_CLEAN(hipEventCreateWithFlags(&sl.ev_planned, hipEventDisableTiming));
        TRY_OR_CLEAN(hipMalloc((void **)&sl.d_counters, 4 * sizeof(uint32_t)));
        TRY_OR_CLEAN(hipHostMalloc((void **)&sl.h_counters, 4 * sizeof(uint32_t), hipHostMallocMapped));
        TRY_OR_CLEAN(hipHostGetDevicePointer((void **)&sl.h_counters_dev, sl.h_counters, 0));
    }
    TRY_OR_CLEAN(hipMalloc((void **)&rx->d_taps, taps.size() * sizeof(float)));
    TRY_OR_CLEAN(hipMemcpy(rx->d_taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice));
    rx->state_bytes = carve_state(rx->P, nullptr, rx->S);
    TRY_OR_CLEAN(hipMalloc(&rx->d_state_blob, rx->state_bytes));
    TRY_OR_CLEAN(hipMemset(rx->d_state_blob, 0, rx->state_bytes));
    carve_state(rx->P, (char *)rx->d_state_blob, rx->S);
    TRY_OR_CLEAN(same::launch_init_state(rx->P, rx->S, 0, rx->own_stream));
    TRY_OR_CLEAN(hipStreamSynchronize(rx->own_stream));
#undef TRY_OR_CLEAN
    if (!(flags & SAME_BATCH_LINK_ONLY)) { rx->thot.resize(n_channels); rx->tcold.resize(n_channels); }
    if (rx->tp.enabled) { rx->tp.sym_off.assign(n_channels, 0); rx->tp.synth.assign(n_channels, TickSynth{}); }
    *out = rx;
    return SAME_OK;
}

void same_batch_free(same_batch *rx)
{
    if (!rx) return;
    (void)hipSetDevice(rx->device);
    if (rx->last_stream) (void)hipStreamSynchronize(rx->last_stream);
    if (rx->d_taps) (void)hipFree(rx->d_taps);
    if (rx->d_state_blob) (void)hipFree(rx->d_state_blob);
    for (auto &sl : rx->slot) {
        if (sl.d_events) (void)hipFree(sl.d_events);
        if (sl.d_sort) (void)hipFree(sl.d_sort);
        if (sl.d_sorted) (void)hipFree(sl.d_sorted);
        if (sl.h_sort) (void)hipHostFree(sl.h_sort);
        if (sl.d_bursts) (void)hipFree(sl.d_bursts);
        if (sl.d_counters) (void)hipFree(sl.d_counters);
        if (sl.h_counters) (void)hipHostFree(sl.h_counters);
        if (sl.h_events) (void)hipHostFree(sl.h_events);
        if (sl.h_bursts) (void)hipHostFree(sl.h_bursts);
        if (sl.ev_start) (void)hipEventDestroy(sl.ev_start);
        if (sl.ev_stop) (void)hipEventDestroy(sl.ev_stop);
        if (sl.ev_k0) (void)hipEventDestroy(sl.ev_k0);
        if (sl.ev_k1) (void)hipEventDestroy(sl.ev_k1);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (sl.ev_planned) (void)hipEventDestroy(sl.ev_planned);
        if (sl.d_handover) (void)hipFree(sl.d_handover);
        if (sl.h_handover) (void)hipHostFree(sl.h_handover);
        if (sl.d_geom) (void)hipFree(sl.d_geom);
        if (sl.h_geom) (void)hipHostFree(sl.h_geom);
    }
    if (rx->inv.d_buf) (void)hipFree(rx->inv.d_buf);
    if (rx->inv.ev_buf) (void)hipEventDestroy(rx->inv.ev_buf);
    if (rx->tp.blob) (void)hipFree(rx->tp.blob);
    if (rx->tp.blob_fresh) (void)hipFree(rx->tp.blob_fresh);
    if (rx->tp.d_desc_in) (void)hipFree(rx->tp.d_desc_in);
    if (rx->tp.d_desc_out) (void)hipFree(rx->tp.d_desc_out);
    if (rx->tp.d_final_col) (void)hipFree(rx->tp.d_final_col);
    if (rx->tp.d_energy) (void)hipFree(rx->tp.d_energy);
    if (rx->tp.d_hist) (void)hipFree(rx->tp.d_hist);
    if (rx->tp.ev_plan_prev) (void)hipEventDestroy(rx->tp.ev_plan_prev);
    if (rx->copy_stream) (void)hipStreamDestroy(rx->copy_stream);
    if (rx->plan_stream) (void)hipStreamDestroy(rx->plan_stream);
    if (rx->h_wake) (void)hipHostFree(rx->h_wake);
    if (rx->d_stage) (void)hipFree(rx->d_stage);
    if (rx->d_stage2) (void)hipFree(rx->d_stage2);
    if (rx->d_upload) (void)hipFree(rx->d_upload);
    if (rx->d_zero) (void)hipFree(rx->d_zero);
    if (rx->ev_order) (void)hipEventDestroy(rx->ev_order);
    if (rx->own_stream) (void)hipStreamDestroy(rx->own_stream);
    delete rx;
}

int same_batch_reset(same_batch *rx)
{
    if (!rx) return fail(SAME_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(rx->device));
    int rc = harvest(rx);
    if (rc) return rc;
    hipError_t e = same::launch_init_state(rx->P, rx->S, 1, rx->own_stream);
    if (e != hipSuccess) return fail(SAME_EHIP, "reset launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(rx->own_stream));
    rx->counter = 0;
    rx->inv.fill = 0;                        // (samples that were waiting for their window are dropped with the rest of the state)
    rx->queue.clear(); rx->queue_head = 0;   // event_queue.clear() receiver.rs:194
    rx->arena.clear();
    rx->peeked_valid = false; rx->peeked_release = false; rx->peeked_head = 0;
    std::vector<same_rx_event>().swap(rx->peeked);
    rx->burst_seq.clear(); rx->burst_seq_head = 0;
    for (uint32_t c = 0; c < (uint32_t)rx->thot.size(); ++c) rx->tr(c).reset();
    for (auto &o : rx->tp.sym_off) o = 0;
    for (auto &t : rx->tp.synth) t.reset();
    if (rx->h_wake) std::memset(rx->h_wake, 0, (size_t)rx->P.n_channels * sizeof(uint64_t));
    rx->overflowed = false;
    rx->kernel_fault = false;
    return SAME_OK;
}

uint32_t same_batch_input_rate(const same_batch *rx) { return rx ? rx->P.input_rate : 0; }
uint32_t same_batch_n_channels(const same_batch *rx) { return rx ? rx->P.n_channels : 0; }
uint64_t same_batch_input_sample_counter(const same_batch *rx) { return rx ? rx->counter + rx->inv.fill : 0; }      // (samples accepted: demodulated or waiting for their window)
int same_batch_device(const same_batch *rx) { return rx ? rx->device : -1; }

int same_batch_process_device(same_batch *rx, const float *d_x, size_t n_samples, uint32_t layout, void *hip_stream)
{ return process_device_any<float>(rx, d_x, n_samples, layout, hip_stream); }
int same_batch_process_device_i16(same_batch *rx, const int16_t *d_x, size_t n_samples, uint32_t layout, void *hip_stream)
{ return process_device_any<int16_t>(rx, d_x, n_samples, layout, hip_stream); }
int same_batch_process_host(same_batch *rx, const float *h_x, size_t n_samples, uint32_t layout)
{ return process_host_any<float>(rx, h_x, n_samples, layout); }
int same_batch_process_host_i16(same_batch *rx, const int16_t *h_x, size_t n_samples, uint32_t layout)
{ return process_host_any<int16_t>(rx, h_x, n_samples, layout); }

int same_batch_flush(same_batch *rx)
{
    if (!rx) return fail(SAME_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(rx->device));
    // four seconds of zeros per channel (receiver.rs:216-224), generated on the device
    const size_t n = (size_t)rx->P.input_rate * 4;
    const size_t slab = std::max<size_t>(1, std::min<size_t>(n, ((size_t)256 << 20) / ((size_t)rx->P.n_channels * sizeof(float))));
    if (slab * rx->P.n_channels * sizeof(float) > rx->zero_bytes) {
        int rc0 = ensure_stage(&rx->d_zero, &rx->zero_bytes, slab * rx->P.n_channels * sizeof(float));
        if (rc0) return rc0;
        HIP_TRY(hipMemset(rx->d_zero, 0, rx->zero_bytes));       // once per growth: the kernels only read it
    }
    int rc = SAME_OK;
    for (size_t t0 = 0; t0 < n && rc == SAME_OK; t0 += slab) {
        rc = process_device_any<float>(rx, (const float *)rx->d_zero, std::min(slab, n - t0), SAME_LAYOUT_TIME_MAJOR, SAME_STREAM_OWN);
        if (rc == SAME_OK) rc = harvest(rx);
    }
    if (rc == SAME_OK && rx->inv.on) {
        // the end of the stream as far as the windows go: what is waiting is demodulated, the next window begins here
        rc = inv_launch_waiting(rx, rx->own_stream);
        if (rc == SAME_OK) rc = harvest(rx);
    }
    return rc;
}

int same_batch_set_call_window(same_batch *rx, uint32_t samples)
{
    if (!rx) return fail(SAME_EINVAL, "null handle");
    if (!rx->inv.on) return fail(SAME_EINVAL, "the batch was not made with SAME_BATCH_CALL_INVARIANT");
    if (samples < 64u || samples > (1u << 22)) return fail(SAME_EINVAL, "window of %u samples (64 .. 4 194 304)", samples);
    if (rx->counter || rx->inv.fill) return fail(SAME_EINVAL, "the window is set before the first sample (or right behind a reset)");
    rx->inv.window = samples;
    return SAME_OK;
}

int same_batch_order_after(same_batch *rx, void *producer_stream)
{
    if (!rx) return fail(SAME_EINVAL, "null handle");
    if (producer_stream == SAME_STREAM_OWN) return SAME_OK;
    HIP_TRY(hipSetDevice(rx->device));
    HIP_TRY(hipEventRecord(rx->ev_order, (hipStream_t)producer_stream));
    HIP_TRY(hipStreamWaitEvent(rx->own_stream, rx->ev_order, 0));
    HIP_TRY(hipStreamWaitEvent(rx->plan_stream, rx->ev_order, 0));      // (the planner reads the input too)
    return SAME_OK;
}

int same_batch_sync(same_batch *rx)
{
    if (!rx) return fail(SAME_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(rx->device));
    int rc = harvest(rx);
    if (rc) return rc;
    if (rx->kernel_fault) return fail(SAME_EKERNEL, "a demodulation kernel's wavefronts lost step (internal hand-over timed out)");
    if (rx->overflowed) return fail(SAME_EOVERFLOW, "event/burst pool overflow");
    return SAME_OK;
}

size_t same_batch_pending_events(same_batch *rx)
{
    // events already brought back to the host; does not wait for launches still in flight
    if (!rx) return 0;
    return rx->queue.size() - rx->queue_head;
}

// The view same_batch_peek_events handed out is tens of MB for a large queue and is not kept for the handle's life -- but the
// header promises it stays readable across same_batch_drop_events, so a drop that empties the queue only marks it; the next
// call that ends the view's life anyway (poll, peek, a harvest, reset) lets the memory go.
static void release_stale_view(same_batch *rx)
{
    if (!rx->peeked_release) return;
    rx->peeked_release = false;
    if (!rx->peeked_valid) std::vector<same_rx_event>().swap(rx->peeked);
}
// the queue is empty: nothing refers to the arena any more
static void queue_emptied(same_batch *rx, bool view_may_be_read = false)
{
    rx->queue.clear(); rx->queue_head = 0; rx->arena.clear();
    rx->peeked_valid = false; rx->peeked_head = 0;
    if (view_may_be_read) rx->peeked_release = true;
    else { rx->peeked_release = false; std::vector<same_rx_event>().swap(rx->peeked); }
}

int same_batch_poll_events(same_batch *rx, same_rx_event *out, size_t cap, size_t *n_out, size_t *n_left)
{
    if (!rx || (!out && cap)) return fail(SAME_EINVAL, "null argument");
    release_stale_view(rx);
    const size_t avail = rx->queue.size() - rx->queue_head;
    const size_t n = std::min(cap, avail);
    const QEvent *q = rx->queue.data() + rx->queue_head;
    const uint8_t *arena = rx->arena.data();
    const size_t abase = rx->arena.base;
    if (n >= 4096) {
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t n_threads = std::min<size_t>({8u, hw ? hw : 1u});
        rx->workers.run(n_threads, [&](size_t t) {
            for (size_t i = n * t / n_threads; i < n * (t + 1) / n_threads; ++i) materialise(q[i], arena, abase, out + i);
        });
    } else {
        for (size_t i = 0; i < n; ++i) materialise(q[i], arena, abase, out + i);
    }
    rx->queue_head += n;
    if (rx->queue_head == rx->queue.size()) queue_emptied(rx);
    if (n_out) *n_out = n;
    if (n_left) *n_left = rx->queue.size() - rx->queue_head;
    return SAME_OK;
}

int same_batch_peek_events(same_batch *rx, const same_rx_event **events, size_t *n)
{
    if (!rx || !events || !n) return fail(SAME_EINVAL, "null argument");
    // (the queue holds compact records: the view is made here, and stays valid as the header promises -- until the next
    // call on this handle other than same_batch_pending_events / same_batch_drop_events)
    release_stale_view(rx);
    const size_t avail = rx->queue.size() - rx->queue_head;
    if (rx->peeked_valid && rx->peeked_head == rx->queue_head && rx->peeked.size() == avail) {     // (nothing was queued or dropped since)
        *n = avail; *events = avail ? rx->peeked.data() : nullptr;
        return SAME_OK;
    }
    if (rx->peeked_valid && rx->peeked_head <= rx->queue_head && rx->queue_head - rx->peeked_head <= rx->peeked.size() &&
        rx->peeked.size() - (rx->queue_head - rx->peeked_head) == avail) {
        // only drops since the last peek (the usual consumer: peek, keep a few, drop): the view moves up, nothing is rebuilt
        rx->peeked.erase(rx->peeked.begin(), rx->peeked.begin() + (ptrdiff_t)(rx->queue_head - rx->peeked_head));
        rx->peeked_head = rx->queue_head;
        *n = avail; *events = avail ? rx->peeked.data() : nullptr;
        return SAME_OK;
    }
    try { rx->peeked.resize(avail); } catch (...) { return fail(SAME_ENOMEM, "event view"); }
    rx->peeked_valid = true; rx->peeked_head = rx->queue_head;
    const QEvent *q = rx->queue.data() + rx->queue_head;
    const uint8_t *arena = rx->arena.data();
    const size_t abase = rx->arena.base;
    same_rx_event *out = rx->peeked.data();
    if (avail >= 4096) {
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t n_threads = std::min<size_t>({8u, hw ? hw : 1u});
        rx->workers.run(n_threads, [&](size_t t) {
            for (size_t i = avail * t / n_threads; i < avail * (t + 1) / n_threads; ++i) materialise(q[i], arena, abase, out + i);
        });
    } else {
        for (size_t i = 0; i < avail; ++i) materialise(q[i], arena, abase, out + i);
    }
    *n = avail;
    *events = avail ? out : nullptr;
    return SAME_OK;
}

int same_batch_drop_events(same_batch *rx, size_t n)
{
    if (!rx) return fail(SAME_EINVAL, "null argument");
    if (n > rx->queue.size() - rx->queue_head) return fail(SAME_EINVAL, "more events than are queued");
    rx->queue_head += n;
    if (rx->queue_head == rx->queue.size()) queue_emptied(rx, /*view_may_be_read=*/true);
    return SAME_OK;
}

int same_batch_pack_bursts(same_batch *rx, uint32_t first_channel, uint8_t *out, size_t cap, size_t *n_records)
{
    if (!rx || !n_records) return fail(SAME_EINVAL, "null argument");
    static_assert(SAME_BURST_RECORD_BYTES == 16 + SAME_EVENT_MAX_BYTES, "record layout");
    // the queued bursts by the index the harvest keeps (no scan of the whole queue: bursts are a fifth of the events)
    const size_t first_seq = rx->queue.base + rx->queue_head;
    while (rx->burst_seq_head < rx->burst_seq.size() && rx->burst_seq[rx->burst_seq_head] < first_seq) ++rx->burst_seq_head;
    const size_t n_b = rx->burst_seq.size() - rx->burst_seq_head;
    const size_t *seq = rx->burst_seq.data() + rx->burst_seq_head;
    const QEvent *q = rx->queue.data();
    const size_t base = rx->queue.base;
    const uint8_t *arena = rx->arena.data();
    const size_t abase = rx->arena.base;
    *n_records = n_b;
    if (!out || !cap) return SAME_OK;
    const size_t n_copy = std::min(n_b, cap);
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_threads = n_copy >= 8192 ? std::min<size_t>({8u, hw ? hw : 1u}) : 1u;
    auto copy_slice = [&](size_t t) {
        for (size_t i = n_copy * t / n_threads; i < n_copy * (t + 1) / n_threads; ++i) {
            const QEvent &e = q[seq[i] - base];
            uint8_t *r = out + i * SAME_BURST_RECORD_BYTES;
            const uint32_t ch = e.channel + first_channel, len = e.n_bytes;
            std::memcpy(r, &ch, 4); std::memcpy(r + 4, &e.sample_counter, 8); std::memcpy(r + 12, &len, 4);
            if (len) std::memcpy(r + 16, arena + ((size_t)e.payload - abase), len);
            std::memset(r + 16 + len, 0, SAME_EVENT_MAX_BYTES - len);
        }
    };
    rx->workers.run(n_threads, copy_slice);
    return SAME_OK;
}

int same_batch_read_trace(same_batch *rx, uint32_t channel, same_symbol_trace *out, size_t cap, size_t *n_out)
{
    if (!rx || !n_out) return fail(SAME_EINVAL, "null argument");
    if (!rx->P.trace_cap) return fail(SAME_EINVAL, "batch was not created with SAME_BATCH_TRACE_SYMBOLS");
    if (channel >= rx->P.n_channels) return fail(SAME_EINVAL, "channel out of range");
    HIP_TRY(hipSetDevice(rx->device));
    int rc = harvest(rx);
    if (rc) return rc;
    uint32_t n = 0;
    HIP_TRY(hipMemcpy(&n, rx->S.trace_n + channel, sizeof(n), hipMemcpyDeviceToHost));
    n = std::min(n, rx->P.trace_cap);
    n = (uint32_t)std::min<size_t>(n, cap);
    std::vector<float> f((size_t)n * 4);
    std::vector<uint64_t> idx(n);
    if (n) {
        HIP_TRY(hipMemcpy(f.data(), rx->S.trace + (size_t)channel * rx->P.trace_cap * 4, f.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(idx.data(), rx->S.trace_idx + (size_t)channel * rx->P.trace_cap, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    for (uint32_t i = 0; i < n; ++i) {
        out[i].sample_counter = idx[i];
        out[i].zero = f[4 * i]; out[i].sym = f[4 * i + 1]; out[i].err = f[4 * i + 2];
        out[i].samples_until_next_ted = f[4 * i + 3];
    }
    *n_out = n;
    return SAME_OK;
}

void same_batch_set_kernel_timing(same_batch *rx, int enable) { if (rx) { rx->timing = enable != 0; rx->have_timing = false; } }
int same_batch_last_kernel_ms(same_batch *rx, float *ms)
{
    if (!rx || !ms) return fail(SAME_EINVAL, "null argument");
    // duration of the most recently collected launch; never waits for one still running
    if (!rx->have_timing) return fail(SAME_EINVAL, "no timed launch yet");
    *ms = rx->last_ms;
    return SAME_OK;
}
int same_batch_last_demod_kernel_ms(same_batch *rx, float *ms)
{
    if (!rx || !ms) return fail(SAME_EINVAL, "null argument");
    if (!rx->have_timing) return fail(SAME_EINVAL, "no timed launch yet");
    *ms = rx->last_demod_ms;
    return SAME_OK;
}
int same_batch_time_parallel_config(same_batch *rx, uint32_t max_chunks, uint32_t min_own_samples, uint32_t warmup_samples)
{
    if (!rx) return fail(SAME_EINVAL, "null handle");
    if (!rx->tp.enabled) return fail(SAME_EINVAL, "batch was not created with SAME_BATCH_TIME_PARALLEL");
    rx->tp.max_chunks = max_chunks; rx->tp.min_own = min_own_samples; rx->tp.warmup = warmup_samples;
    return SAME_OK;
}
uint32_t same_batch_time_parallel_chunks(const same_batch *rx) { return rx ? rx->tp.last_chunks : 0; }
int same_batch_time_parallel_per_channel(const same_batch *rx) { return rx && rx->tp.last_chunks > 1u && rx->tp.last_per_channel ? 1 : 0; }

const char *same_batch_kernel_name(const same_batch *rx)
{
    if (!rx) return "";
    if (rx->tp.last_chunks > 1u) {
        switch (rx->tp.kernel) {
        case same_batch::TimePar::kWaveRelaxed: return "demod_relaxed_kernel";
        case same_batch::TimePar::kPipeRelaxed: return same::sym_kernel_supported(rx->tp.Pv) ? "demod_sym_kernel" : "demod_pipe_kernel<fastmath>";
        default: return "demod_pipe_kernel";
        }
    }
    if (rx->relaxed_plain && rx->last_plain_fm) return rx->last_fm_sym ? "demod_sym_kernel" : "demod_pipe_kernel<fastmath>";
    if (rx->relaxed_plain && rx->last_plain_wave) return "demod_relaxed_kernel";
    if (rx->use_fast && !rx->force_generic) 
    {
        const uint32_t st = same::pipe_kernel_stages(rx->P);
        return st != 0u ? "demod_pipe_kernel" : "demod_fast_kernel";
    }
    switch (rx->P.block_len) {
    case 16: return "demod_kernel<B=16>";
    case 8: return "demod_kernel<B=8>";
    case 4: return "demod_kernel<B=4>";
    case 2: return "demod_kernel<B=2>";
    default: return "demod_kernel<B=1>";
    }
}

// ------------------------------------------------------------------------------------
// single receiver with the reference's pull semantics
// ------------------------------------------------------------------------------------

// The host half of a step on a recorded launch, `reps` times on `threads` harvest threads, without a device: wall time of
// each repetition in ms_out[reps] (the record's counters are moved on by the launch's length every time, so the transport
// layer sees a stream that continues).  Returns the number of queue records one repetition produced, or a negative error.
long same_debug_harvest_replay(const char *path, int threads, int reps, double *ms_out)
{
    if (!path || reps <= 0 || !ms_out) return -(long)fail(SAME_EINVAL, "null argument");
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return -(long)fail(SAME_EINVAL, "cannot read %s", path);
    HarvestFileHeader h{};
    bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && h.magic == kHarvestMagic && h.n_bins >= h.n_channels && h.n_channels > 0;
    std::vector<uint32_t> first, geom;
    std::vector<same::DevEvent> ev0, ev;
    std::vector<uint8_t> bursts;
    std::vector<uint64_t> hand0, hand;
    if (ok) {
        first.resize((size_t)h.n_bins + 1); ev0.resize(h.n_events); bursts.resize((size_t)h.n_bursts * same::kBurstCap);
        ok = std::fread(first.data(), sizeof(uint32_t), first.size(), f) == first.size();
        ok = ok && (!h.n_events || std::fread(ev0.data(), sizeof(same::DevEvent), h.n_events, f) == h.n_events);
        ok = ok && (!h.n_bursts || std::fread(bursts.data(), same::kBurstCap, h.n_bursts, f) == h.n_bursts);
        if (ok && h.chunked) {
            hand0.resize(h.n_bins);
            ok = std::fread(hand0.data(), sizeof(uint64_t), h.n_bins, f) == h.n_bins;
            if (ok && h.per_channel) { geom.resize((size_t)2 * h.n_bins); ok = std::fread(geom.data(), sizeof(uint32_t), geom.size(), f) == geom.size(); }
        }
    }
    std::fclose(f);
    if (ok) {
        // a record is data from a file: nothing in it is trusted before it has been checked against the sizes that were read
        ok = first[h.n_bins] <= h.n_events && h.n_bins % h.n_channels == 0;
        for (size_t i = 0; ok && i < (size_t)h.n_bins; ++i) ok = first[i] <= first[i + 1];
        for (size_t i = 0; ok && i < ev0.size(); ++i)
            ok = (ev0[i].burst_slot == 0xffffffffu || ev0[i].burst_slot < h.n_bursts) && (ev0[i].kind == same::kDevEventNone || ev0[i].kind <= 8u);
        if (ok && h.chunked) {
            ok = h.geom.n_chunks >= 1 && (uint64_t)h.geom.n_chunks * h.n_channels == h.n_bins && h.geom.block_len >= 1 && h.geom.stride_blocks >= 1;
            for (size_t i = 0; ok && h.per_channel && i < geom.size(); ++i) ok = geom[i] <= 0x7fffffffu;
        } else if (ok) {
            ok = h.n_bins == h.n_channels;
        }
    }
    if (!ok) return -(long)fail(SAME_EINVAL, "%s is not a (consistent) harvest record", path);
    same_batch *rx = new (std::nothrow) same_batch;
    if (!rx) return -(long)fail(SAME_ENOMEM, "out of memory");
    rx->P.n_channels = h.n_channels; rx->P.input_rate = h.input_rate; rx->flags = h.flags;
    if (std::getenv("SAME_REPLAY_LINK_ONLY")) rx->flags |= SAME_BATCH_LINK_ONLY;      // (what the transport layer's share is)
    rx->tp.enabled = h.tp_enabled != 0; rx->host_threads = threads;
    if (!(h.flags & SAME_BATCH_LINK_ONLY)) { rx->thot.resize(h.n_channels); rx->tcold.resize(h.n_channels); }
    if (rx->tp.enabled) { rx->tp.sym_off.assign(h.n_channels, 0); rx->tp.synth.assign(h.n_channels, TickSynth{}); }
    same_batch::Slot &sl = rx->slot[0];
    ev = ev0; hand = hand0;
    sl.h_sort = first.data(); sl.h_events = ev.data(); sl.h_bursts = bursts.data();
    sl.h_handover = hand.data(); sl.h_geom = geom.data();
    sl.chunked = h.chunked != 0; sl.per_channel = h.per_channel != 0; sl.sort_bins = h.n_bins;
    const uint64_t span = h.end_counter - h.geom.counter0;        // (an ordinary launch: geom is zero, the span the end counter -- fine for a stride)
    const double sps = (double)h.input_rate / 520.83;
    const uint64_t span_sym = (uint64_t)((double)span / sps);
    long produced = 0;
    int rc = SAME_OK;
    for (int r = 0; r < reps && rc == SAME_OK; ++r) {
        const uint64_t dt = span * (uint64_t)r, ds = span_sym * (uint64_t)r;
        for (size_t i = 0; i < ev0.size(); ++i) { ev[i] = ev0[i]; ev[i].sample_counter += dt; if (!sl.chunked) ev[i].symbol_count += ds; }
        for (size_t i = 0; i < hand0.size(); ++i) hand[i] = hand0[i] == same::kNoHandover ? hand0[i] : hand0[i] + dt;
        sl.geom = h.geom; sl.geom.counter0 += dt;
        sl.end_blocks = h.end_blocks + dt; sl.end_counter = h.end_counter + dt;
        std::vector<uint32_t> rearm;
        HarvestTimes times;
        const auto t0 = std::chrono::steady_clock::now();
        rc = harvest_host(rx, sl, h.n_events, h.n_bursts, rearm, times);
        ms_out[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        produced = (long)(rx->queue.size() - rx->queue_head);
        // the consumer's side: everything polled
        rx->queue_head = rx->queue.size();
        rx->burst_seq_head = rx->burst_seq.size();
    }
    sl.h_sort = nullptr; sl.h_events = nullptr; sl.h_bursts = nullptr; sl.h_handover = nullptr; sl.h_geom = nullptr;
#ifdef SAME_HOST_PROF
    { uint64_t tot = 0; for (int k = 0; k < 7; ++k) tot += g_hp[k];
      for (int k = 0; k < 7; ++k) std::fprintf(stderr, "[same prof] %-24s %8.3f Mclk per harvest (%4.1f %% of the instrumented parts)\n", g_hp_name[k], (double)g_hp[k] / reps / 1e6, tot ? 100.0 * (double)g_hp[k] / (double)tot : 0.0); }
#endif
    delete rx;
    return rc == SAME_OK ? produced : -(long)rc;
}

}  // extern "C"

struct same_rx {
    same_batch *batch = nullptr;
    uint64_t reported = 0;           // samples the caller has logically consumed
    uint64_t zeros_until = 0;        // the device ran over flush zeros up to this sample (0 = no such run-ahead)
    std::deque<same_rx_event> events;
};

static int rx_pump(same_rx *rx)
{
    same_rx_event buf[64];
    for (;;) {
        size_t n = 0, left = 0;
        int rc = same_batch_poll_events(rx->batch, buf, 64, &n, &left);
        if (rc) return rc;
        for (size_t i = 0; i < n; ++i) rx->events.push_back(buf[i]);
        if (!left) return SAME_OK;
    }
}

extern "C" {

int same_rx_build(const same_rx_builder *b, int device, same_rx **out)
{
    if (!out) return fail(SAME_EINVAL, "null out");
    *out = nullptr;
    same_rx *rx = new (std::nothrow) same_rx;
    if (!rx) return fail(SAME_ENOMEM, "out of memory");
    int rc = same_batch_new(b, 1, device, 0, &rx->batch);
    if (rc) { delete rx; return rc; }
    *out = rx;
    return SAME_OK;
}
void same_rx_free(same_rx *rx) { if (rx) { same_batch_free(rx->batch); delete rx; } }

static int rx_process(same_rx *rx, const float *x, size_t n, size_t *consumed, same_rx_event *ev);

int same_rx_process(same_rx *rx, const float *x, size_t n, size_t *consumed, same_rx_event *ev)
{
    if (!rx || !consumed || !ev || (!x && n)) return fail(SAME_EINVAL, "null argument");
    *consumed = 0;
    if (rx->zeros_until > rx->reported && n)
        return fail(SAME_EINVAL, "a flush returned at its first message with the device %llu zero samples ahead; "
                                 "reset the receiver (or flush again) before presenting audio",
                    (unsigned long long)(rx->zeros_until - rx->reported));
    return rx_process(rx, x, n, consumed, ev);
}

static int rx_process(same_rx *rx, const float *x, size_t n, size_t *consumed, same_rx_event *ev)
{
    *consumed = 0;
    // x[0] sits at absolute sample `reported`; the device may already be past it
    const uint64_t device_at = same_batch_input_sample_counter(rx->batch);
    const uint64_t end = rx->reported + n;
    if (end > device_at) {
        const size_t skip = (size_t)(device_at - rx->reported);
        int rc = same_batch_process_host(rx->batch, x + skip, n - skip, SAME_LAYOUT_TIME_MAJOR);
        if (rc) return rc;
        rc = rx_pump(rx);
        if (rc) return rc;
    }
    if (!rx->events.empty() && rx->events.front().sample_counter <= end) {
        *ev = rx->events.front();
        rx->events.pop_front();
        // queued events are returned before any new sample is consumed (receiver.rs:238-240)
        const uint64_t at = ev->sample_counter > rx->reported ? ev->sample_counter : rx->reported;
        *consumed = (size_t)(at - rx->reported);
        rx->reported = at;
        return 1;
    }
    *consumed = n;
    rx->reported = end;
    return 0;
}

int same_rx_flush(same_rx *rx, same_rx_event *msg)
{
    // flush() receiver.rs:216-224: four seconds of zeros, first Message(Ok(..)) wins
    if (!rx) return fail(SAME_EINVAL, "null handle");
    const size_t n = (size_t)same_batch_input_rate(rx->batch) * 4;
    std::vector<float> zeros(n, 0.0f);
    size_t off = 0;
    while (off < n) {
        size_t used = 0;
        same_rx_event ev;
        int got = rx_process(rx, zeros.data() + off, n - off, &used, &ev);
        if (got < 0) return got;
        off += used;
        if (!got) break;
        if (ev.kind == SAME_TRANSPORT_MSG_START || ev.kind == SAME_TRANSPORT_MSG_END) {
            if (msg) *msg = ev;
            // the device has consumed all n zeros, the caller only those up to this event
            rx->zeros_until = same_batch_input_sample_counter(rx->batch);
            return 1;
        }
    }
    return 0;
}

int same_rx_reset(same_rx *rx)
{
    if (!rx) return fail(SAME_EINVAL, "null handle");
    rx->events.clear();
    rx->reported = 0;
    rx->zeros_until = 0;
    return same_batch_reset(rx->batch);
}
uint32_t same_rx_input_rate(const same_rx *rx) { return rx ? same_batch_input_rate(rx->batch) : 0; }
uint64_t same_rx_input_sample_counter(const same_rx *rx) { return rx ? rx->reported : 0; }

int same_synth_afsk_device(float *d_x, uint32_t n_channels, size_t n_samples, uint32_t input_rate,
                           uint64_t seed, float noise_sigma, uint32_t flags, int device, void *hip_stream)
{
    if (!d_x || !n_channels) return fail(SAME_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SAME_ENODEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    same::SynthParams sp{n_channels, input_rate, seed, noise_sigma, flags};
    hipError_t e = same::launch_synth(sp, d_x, n_samples, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(SAME_EHIP, "synth launch failed: %s", hipGetErrorString(e));
    return SAME_OK;
}
uint32_t same_synth_payload(uint64_t seed, uint32_t channel, uint8_t *out, uint32_t cap)
{ return same::synth_payload(seed, channel, out, cap); }

int same_synth_trials_device(float *d_x, uint32_t n_trials, uint32_t first_trial, size_t n_samples,
                             uint32_t input_rate, uint64_t seed, float ebn0_db_lo, float ebn0_db_step,
                             uint32_t n_grid, int device, void *hip_stream)
{
    if (!d_x || !n_trials || !n_grid) return fail(SAME_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SAME_ENODEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    same::TrialParams tp{n_trials, first_trial, input_rate, n_grid, seed, ebn0_db_lo, ebn0_db_step};
    hipError_t e = same::launch_trials(tp, d_x, n_samples, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(SAME_EHIP, "trial generator launch failed: %s", hipGetErrorString(e));
    return SAME_OK;
}

}  // extern "C"
