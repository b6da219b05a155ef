// same_transport.h -- host-side transport layer: turns the link events of one channel
// into TransportState events (bursts -> 2-of-3 bit voting -> validated header text).
//
// This is SURVEY.md section 8f "next-1": rx/assembler.rs, rx/combiner.rs, rx/timeddata.rs,
// receiver.rs:291-333 and the header check of crates/sameplace/src/message.rs:813-828.
// It is symbol-rate integer/string work on a handful of bursts per message, so it runs
// on the host over the events the device reports.
//
// The reference polls the assembler on every symbol whose link state is NoCarrier or
// Burst (receiver.rs:292-315).  The device reports exactly those poll instants at which
// the assembler's answer can change (link events plus SAME_DEV_TICK wake-ups, see
// same_kernels.hip "transport wake-ups"); replaying the assembler at those instants
// reproduces the reference's transport events with identical sample counters.
#pragma once

#include <cstdint>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/same_rx.h"

namespace same {

constexpr uint32_t kDevTick = 8;                 // device-only event kind (never surfaced)
constexpr size_t kMaxMessageLength = 268;        // rx/assembler.rs:70

uint64_t max_interburst_symbols();               // rx/assembler.rs:85  (682)
uint64_t max_history_duration();                 // rx/assembler.rs:92-93 (5652)

// crates/sameplace/src/message.rs:813-828; returns true and fills offsets on a match
bool check_header(const uint8_t *hdr, size_t n, size_t *offset_time, size_t *hdr_len);
void bit_vote_detect(uint8_t b0, uint8_t b1, uint8_t *out, uint32_t *errs);   // rx/combiner.rs:216-222
void bit_vote_correct(uint8_t b0, uint8_t b1, uint8_t b2, uint8_t *out, uint32_t *errs); // :234-249
bool is_allowed_byte(uint8_t c);                 // rx/combiner.rs:105-137

// MessageResult = Result<Message, MessageDecodeErr>.  Fixed storage: this sits on the
// per-event path of every channel, so it must not touch the heap.
struct MessageResult {
    uint32_t kind = 0;               // SAME_TRANSPORT_MSG_START / _END / _ERR
    uint32_t err = 0;                // 1 NotAscii, 2 UnrecognizedPrefix, 3 Malformed
    uint32_t len = 0;                // header length for StartOfMessage
    uint32_t offset_time = 0, parity_errors = 0, voting_bytes = 0;
    char text[kMaxMessageLength + 4];
    // (copies move the header fields and the `len` bytes of text that are live, not the 272-byte buffer: a burst makes five of them)
    MessageResult() = default;
    MessageResult(const MessageResult &o) { *this = o; }
    MessageResult &operator=(const MessageResult &o)
    {
        kind = o.kind; err = o.err; len = o.len; offset_time = o.offset_time; parity_errors = o.parity_errors; voting_bytes = o.voting_bytes;
        if (o.len) std::memcpy(text, o.text, o.len <= sizeof(text) ? o.len : sizeof(text));
        return *this;
    }
    bool operator==(const MessageResult &o) const
    {
        return kind == o.kind && err == o.err && len == o.len && offset_time == o.offset_time &&
               parity_errors == o.parity_errors && voting_bytes == o.voting_bytes &&
               std::memcmp(text, o.text, len) == 0;
    }
    // Message::as_str(): header text, or "NNNN" for EndOfMessage (sameplace message.rs:105-110)
    bool same_text(const MessageResult &o) const
    {
        const char *a = kind == SAME_TRANSPORT_MSG_END ? "NNNN" : text;
        const char *b = o.kind == SAME_TRANSPORT_MSG_END ? "NNNN" : o.text;
        const uint32_t la = kind == SAME_TRANSPORT_MSG_END ? 4u : len, lb = o.kind == SAME_TRANSPORT_MSG_END ? 4u : o.len;
        return la == lb && std::memcmp(a, b, la) == 0;
    }
};

// one burst as the assembler keeps it (truncated to MAX_MESSAGE_LENGTH, rx/assembler.rs:163-169); its deadline lives beside
// the assembler's other scalars
struct BurstBuf {
    uint32_t len = 0; uint8_t data[kMaxMessageLength];
    BurstBuf() = default;
    BurstBuf(const BurstBuf &o) { *this = o; }
    BurstBuf &operator=(const BurstBuf &o) { len = o.len; if (o.len) std::memcpy(data, o.data, o.len <= sizeof(data) ? o.len : sizeof(data)); return *this; }
};

// combine() rx/combiner.rs:32-80 over up to three bursts; false = None
bool combine(const BurstBuf *bursts, uint32_t n, MessageResult *out);

// Assembler rx/assembler.rs:108-266 and the per-channel transport state of SameReceiver (receiver.rs:79, 85, 89, 291-333).
//
// Layout (round 5): a batch keeps the transport state of every channel and walks it once per launch -- four events per channel
// and launch at the configs[3] shard.  As one 1.8 KB object per channel (60 MB for 32 768 channels) every poll was a DRAM miss on
// a page of its own: 13.8 CPU-ms per launch, 10.8 of them here.  Now the state is TWO arrays: a HOT record of exactly one cache
// line per channel -- everything a poll that changes nothing reads or writes: state kind, flags, the forced-EOM instant, the last
// polled symbol, the burst history's and the pending message's deadlines -- 2 MB for 32 768 channels, which stays in the cache
// from launch to launch; and a COLD record (burst bytes, message texts) touched only when a burst arrives, a message is due or
// a message state ends.
struct alignas(64) TransportHot {      // (C++17 aligned new: std::vector honours it, so a record never straddles two lines)
    uint32_t state_kind = SAME_TRANSPORT_IDLE;
    uint8_t have_force_eom = 0, dirty = 0, have_polled = 0;          // Transport
    uint8_t nhist = 0, pending = 0, have_prev = 0;                   // Assembler: VecDeque<TimedData<Burst>> length, PendingResult, previous message
    uint8_t pad_[2] = {0, 0};
    uint64_t force_eom_at = 0;
    uint64_t last_polled_symbol = 0;
    uint64_t hist_deadline[3] = {0, 0, 0};                           // history[i] until hist_deadline[i], oldest first
    uint64_t pend_deadline = 0;
};
static_assert(sizeof(TransportHot) == 64 && alignof(TransportHot) == 64, "one cache line per channel");
struct TransportCold {
    uint64_t prev_deadline = 0;
    BurstBuf history[3];
    MessageResult pend, prev, state_msg;
};

// The operations on one channel's (hot, cold) pair
class TransportRef {
public:
    TransportRef(TransportHot &h, TransportCold &c) : h_(h), c_(c) {}
    void reset();
    // Feed one device event (link event or tick) of this channel, in order.  Returns true
    // and fills *out when the transport state changed (receiver.rs:256-265).
    bool on_link_event(uint32_t kind, uint64_t sample_counter, uint64_t symbol_count,
                       const uint8_t *bytes, uint32_t len, uint32_t input_rate, same_rx_event *out);
    // force_eom_at_sample (receiver.rs:89): 0 = None
    uint64_t force_eom_at() const { return h_.have_force_eom ? h_.force_eom_at : 0; }
    bool force_eom_dirty() { const bool d = h_.dirty != 0; h_.dirty = 0; return d; }

private:
    // Assembler: both return the TransportState kind and fill *msg for Message states
    uint32_t assemble(const uint8_t *burst, size_t n, uint64_t symbol_count, MessageResult *msg);
    uint32_t idle(uint64_t symbol_count, MessageResult *msg);
    void prune_history(uint64_t now);
    void accept(const MessageResult &m, uint64_t now);
    TransportHot &h_;
    TransportCold &c_;
};

// One channel's transport state as an object of its own (the single-channel receiver, tests)
class Transport {
public:
    void reset() { ref().reset(); }
    bool on_link_event(uint32_t kind, uint64_t sample_counter, uint64_t symbol_count,
                       const uint8_t *bytes, uint32_t len, uint32_t input_rate, same_rx_event *out)
    { return ref().on_link_event(kind, sample_counter, symbol_count, bytes, len, input_rate, out); }
    uint64_t force_eom_at() const { return hot_.have_force_eom ? hot_.force_eom_at : 0; }
    bool force_eom_dirty() { return ref().force_eom_dirty(); }
    void set_input_sample_counter_bias(int64_t) {}

private:
    TransportRef ref() { return TransportRef(hot_, cold_); }
    TransportHot hot_;
    TransportCold cold_;
};


// Time-parallel mode: the transport layer's poll instants, synthesised on the host.  The reference polls
// its assembler on every symbol whose link state is NoCarrier or Burst (receiver.rs:292-315); the answer
// can only change at a burst, at the first poll on or after a deadline a burst armed (burst + 682
// symbols, burst + 5 652 symbols: rx/assembler.rs:85, 92-93, 294-299), after the forced-EOM instant
// (receiver.rs:300-309), or at the poll following one of those.  In strict mode the device reports
// exactly those instants (SAME_DEV_TICK); chunks that start from a fresh receiver cannot know the
// deadlines earlier chunks armed, so here the host keeps them per channel, on the (rebased) symbol
// clock of the stitched event stream, and interpolates the sample counter of a poll from the last event.
struct TickSynth {
    uint32_t link = 0;                 // LinkState kind after the last event
    uint64_t a_sym = 0, a_t = 0;       // last event: symbol count (rebased) and input sample counter
    uint64_t dl[12];                   // pending symbol deadlines, ascending
    uint32_t n = 0;
    bool again = false;                // a deadline passed while the link was busy: poll once more after the next idle event
    void reset() { link = 0; a_sym = 0; a_t = 0; n = 0; again = false; }
    void add(uint64_t d)
    {
        if (n == 12) { for (uint32_t i = 1; i < n; ++i) dl[i - 1] = dl[i]; --n; }
        uint32_t i = n++;
        while (i > 0 && dl[i - 1] > d) { dl[i] = dl[i - 1]; --i; }
        dl[i] = d;
    }
    // polls for every deadline before symbol `sym_limit` (an event at sym_limit polls by itself)
    template <typename Poll> void run_until(uint64_t sym_limit, uint64_t t_limit, double sps, uint64_t force_eom_at, Poll &&poll)
    {
        auto at = [&](uint64_t sym) {
            const uint64_t t = a_t + (uint64_t)((double)(sym - a_sym) * sps);
            return t < t_limit ? t : (t_limit ? t_limit - 1 : 0);
        };
        if (force_eom_at && link == 0 && a_t <= force_eom_at && force_eom_at + 1 < t_limit) {
            const uint64_t sym = a_sym + (uint64_t)((double)(force_eom_at + 1 - a_t) / sps) + 1u;
            if (sym < sym_limit) add(sym);
        }
        while (n && dl[0] < sym_limit) {
            uint64_t d = dl[0];
            for (uint32_t i = 1; i < n; ++i) dl[i - 1] = dl[i];
            --n;
            if (link != 0) { again = true; continue; }       // busy: the next NoCarrier / Burst event is that poll
            if (d <= a_sym) d = a_sym + 1u;
            if (d >= sym_limit) { again = true; continue; }
            poll(d, at(d));
            if (d + 1u < sym_limit && !(n && dl[0] == d + 1u)) poll(d + 1u, at(d + 1u));
        }
    }
    void after_event(uint32_t kind, uint64_t sym, uint64_t t, uint64_t interburst, uint64_t history)
    {
        link = kind; a_sym = sym; a_t = t;
        if (kind == SAME_LINK_BURST) { add(sym + interburst); add(sym + history); again = true; }
        else if (kind == SAME_LINK_NO_CARRIER && again) { add(sym + 1u); again = false; }
    }
};


}  // namespace same
