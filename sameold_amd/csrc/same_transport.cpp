// same_transport.cpp -- see same_transport.h.  Citations: file:line under
// /root/reference/crates/sameold/src/ ("rx/" = receiver/) unless a crate is named.
#include "same_transport.h"

#include <algorithm>

namespace same {

uint64_t max_interburst_symbols()
{
    // ((1.05 * BAUD_HZ) + 17.0 * 8.0) as u64 in f32 arithmetic, rx/assembler.rs:85
    const float baud = 520.83f;
    float a = 1.05f * baud;
    float b = 17.0f * 8.0f;
    float c = a + b;
    return (uint64_t)c;
}
uint64_t max_history_duration() { return 2 * (max_interburst_symbols() + 8 * (uint64_t)kMaxMessageLength); }

// (a table: combine() asks for every byte of every burst of every channel)
struct AllowedTable {
    bool ok[256];
    constexpr AllowedTable() : ok()
    {
        for (int c = 0; c < 256; ++c) {
            bool a = (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z');
            switch (c) {
            case '-': case '/': case '?': case '(': case ')': case '[': case ']': case '.': case '_': case ',':
            case '+': case ' ':
                a = true; break;
            default: break;
            }
            ok[c] = a;
        }
    }
};
static constexpr AllowedTable kAllowed{};
bool is_allowed_byte(uint8_t c) { return kAllowed.ok[c]; }

void bit_vote_detect(uint8_t b0, uint8_t b1, uint8_t *out, uint32_t *errs)
{
    const uint8_t x = b0 ^ b1;
    *out = x ? 0 : b0;                                   // b0 & !(0xff * (xor != 0))
    *errs = (uint32_t)__builtin_popcount(x);
}
void bit_vote_correct(uint8_t b0, uint8_t b1, uint8_t b2, uint8_t *out, uint32_t *errs)
{
    const uint8_t p0 = (uint8_t)~(b0 ^ b1), p1 = (uint8_t)~(b1 ^ b2), p2 = (uint8_t)~(b0 ^ b2);
    *out = (uint8_t)((b0 & p0) | (b2 & p1) | (b2 & p2));
    *errs = 8u - (uint32_t)__builtin_popcount((uint8_t)(p0 & p1 & p2));
}

static bool alpha(uint8_t c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
static bool digit(uint8_t c) { return c >= '0' && c <= '9'; }

bool check_header(const uint8_t *h, size_t n, size_t *offset_time, size_t *hdr_len)
{
    // ^ZCZC-[[:alpha:]]{3}-[[:alpha:]]{3}(-[0-9]{6})+(\+[0-9]{4}-[0-9]{7}-.{3,8}-)
    // The location group is greedy and group 2 must begin with '+', so giving back a
    // repetition can never help; .{3,8} is greedy, so the longest callsign wins.
    auto at = [&](size_t i, uint8_t c) { return i < n && h[i] == c; };
    auto run = [&](size_t i, size_t k, bool (*pred)(uint8_t)) {
        for (size_t j = 0; j < k; ++j) if (i + j >= n || !pred(h[i + j])) return false;
        return true;
    };
    if (n < 5 || std::memcmp(h, "ZCZC-", 5) != 0) return false;
    size_t p = 5;
    if (!run(p, 3, alpha) || !at(p + 3, '-') || !run(p + 4, 3, alpha)) return false;
    p += 7;
    size_t nloc = 0;
    while (at(p, '-') && run(p + 1, 6, digit)) { p += 7; ++nloc; }
    if (!nloc) return false;
    const size_t g2 = p;
    if (!at(p, '+') || !run(p + 1, 4, digit) || !at(p + 5, '-') || !run(p + 6, 7, digit) || !at(p + 13, '-'))
        return false;
    p += 14;
    for (size_t m = 8; m >= 3; --m) {
        if (!at(p + m, '-')) continue;
        bool ok = true;
        for (size_t k = 0; k < m; ++k) if (h[p + k] == '\n') { ok = false; break; }
        if (ok) { *offset_time = g2; *hdr_len = p + m + 1; return true; }
    }
    return false;
}

// Message::try_from((bytes, errs, counts)) crates/sameplace/src/message.rs:718-736, 184-230
static void clear_result(MessageResult *m)
{ m->kind = 0; m->err = 0; m->len = 0; m->offset_time = 0; m->parity_errors = 0; m->voting_bytes = 0; }

static void parse_message(const uint8_t *b, size_t n, const uint8_t *errs, const uint8_t *counts, MessageResult *out)
{
    clear_result(out);
    {
        // (is_ascii, eight bytes at a time; combine()'s estimates are 7-bit by construction, the test is the reference's)
        size_t i = 0; uint64_t hi = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); hi |= w; }
        for (; i < n; ++i) hi |= b[i];
        if (hi & 0x8080808080808080ull) { out->kind = SAME_TRANSPORT_MSG_ERR; out->err = 1; return; }
    }
    if (n >= 5 && std::memcmp(b, "ZCZC-", 5) == 0) {
        size_t off = 0, hl = 0;
        if (!check_header(b, n, &off, &hl)) { out->kind = SAME_TRANSPORT_MSG_ERR; out->err = 3; return; }
        out->kind = SAME_TRANSPORT_MSG_START;
        out->len = (uint32_t)hl;
        std::memcpy(out->text, b, hl);
        out->offset_time = (uint32_t)off;
        {
            // the header's parity errors and voting bytes, eight positions at a time (an error count is at most 8 and a byte of
            // the running word holds 255: 24 words between two foldings; a count is 1, 2 or 3, "three votes" its bits 0 and 1 together)
            size_t i = 0; uint32_t pe = 0, vb = 0;
            while (i + 8 <= hl) {
                uint64_t se = 0, sv = 0; size_t k = 0;
                for (; k < 24 && i + 8 <= hl; ++k, i += 8) {
                    uint64_t e, c; std::memcpy(&e, errs + i, 8); std::memcpy(&c, counts + i, 8);
                    se += e; sv += c & (c >> 1) & 0x0101010101010101ull;
                }
                for (int q = 0; q < 8; ++q) { pe += (uint32_t)((se >> (8 * q)) & 0xffu); vb += (uint32_t)((sv >> (8 * q)) & 0xffu); }
            }
            for (; i < hl; ++i) { pe += errs[i]; vb += counts[i] >= 3 ? 1u : 0u; }
            out->parity_errors = pe; out->voting_bytes = vb;
        }
    } else if (n >= 2 && b[0] == 'N' && b[1] == 'N') {
        out->kind = SAME_TRANSPORT_MSG_END;
    } else {
        out->kind = SAME_TRANSPORT_MSG_ERR; out->err = 2;
    }
}

bool combine(const BurstBuf *bursts, uint32_t nbursts, MessageResult *res)
{
    // estimate_message rx/combiner.rs:154-203
    uint8_t msg[kMaxMessageLength], cnt[kMaxMessageLength], errs[kMaxMessageLength];
    const size_t nb = std::min<size_t>(nbursts, 3);
    size_t n = 0;
    // Every burst is read from its first byte on, so at position n exactly the bursts longer than n take part: first the
    // stretch all of them cover (the usual case: three bursts of one length), then the general walk.
    if (nb == 1) {
        // One burst (a message's first: one call in three).  Every byte of its estimate has one vote, so no prefix of it is
        // trusted (truncate_bytes_with_reference keeps nothing: rx/combiner.rs:264-273) and what is left of combine() is the
        // fast end-of-message (:251-258) on the first two estimated bytes -- 'N' is an allowed character, so the estimate reaches
        // them exactly when they are there -- or, failing that, "no message" (:75-79 with an empty prefix).
        const BurstBuf &a = bursts[0];
        if (a.len == 0 || !kAllowed.ok[a.data[0] & 0x7f]) return false;          // (estimate_message yields nothing: :176-180)
        if (a.len >= 2 && (a.data[0] & 0x7f) == 'N' && (a.data[1] & 0x7f) == 'N') { clear_result(res); res->kind = SAME_TRANSPORT_MSG_END; return true; }
        return false;
    }
    size_t common = kMaxMessageLength;
    for (size_t i = 0; i < nb; ++i) common = std::min<size_t>(common, bursts[i].len);
    bool stopped = false;
    // Eight bytes at a time where all bursts still have bytes: the bit votes of rx/combiner.rs:216-249 are bitwise, so they
    // hold for a word as for a byte (majority = (a & b) | (c & (a | b)); a position is in error where the bursts do not all
    // agree), the per-byte error counts are a byte-wise population count; only the allowed-character test looks at bytes.
    // (81 % of the transport layer's time was the byte loop below: profiles/r05_host_step_probe.txt.)
    if (nb >= 2) {
        constexpr uint64_t k7f = 0x7f7f7f7f7f7f7f7full, k01 = 0x0101010101010101ull;
        auto bytes_popcount = [](uint64_t x) {
            x = x - ((x >> 1) & 0x5555555555555555ull);
            x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
            return (x + (x >> 4)) & 0x0f0f0f0f0f0f0f0full;
        };
        for (; n + 8 <= common; n += 8) {
            uint64_t a, b, c = 0, est, dis, hi;
            std::memcpy(&a, bursts[0].data + n, 8); std::memcpy(&b, bursts[1].data + n, 8);
            if (nb == 3) {
                std::memcpy(&c, bursts[2].data + n, 8);
                hi = ((a | b | c) >> 7) & k01;
                a &= k7f; b &= k7f; c &= k7f;
                est = (a & b) | (c & (a | b));               // bit_vote_correct
                dis = (a ^ b) | (b ^ c);
            } else {
                hi = ((a | b) >> 7) & k01;
                a &= k7f; b &= k7f;
                dis = a ^ b;                                   // bit_vote_detect: the byte where the two agree, else 0
                const uint64_t nz = (((dis + k7f) | dis) >> 7) & k01;       // 1 in every byte that differs
                est = a & ~(nz * 0xffull);
            }
            const uint64_t er = bytes_popcount(dis) + hi;
            // the eight estimates and error counts as two word stores (a byte at or behind a character that is not allowed is
            // never read: n says where the estimate ends); the counts -- nb everywhere in this stretch -- are filled in below
            uint8_t e8[8];
            std::memcpy(e8, &est, 8);
            std::memcpy(msg + n, &est, 8); std::memcpy(errs + n, &er, 8);
            const bool all_ok = kAllowed.ok[e8[0]] & kAllowed.ok[e8[1]] & kAllowed.ok[e8[2]] & kAllowed.ok[e8[3]] &
                                kAllowed.ok[e8[4]] & kAllowed.ok[e8[5]] & kAllowed.ok[e8[6]] & kAllowed.ok[e8[7]];
            if (!all_ok) {
                int j = 0;
                while (kAllowed.ok[e8[j]]) ++j;
                n += (size_t)j; stopped = true; break;
            }
        }
        std::memset(cnt, (int)nb, n);
    }
    if (stopped) {
        // (a character that is not allowed ends the message: rx/combiner.rs:176-180)
    } else if (nb == 3) {
        const uint8_t *a = bursts[0].data, *b = bursts[1].data, *c = bursts[2].data;
        for (; n < common; ++n) {
            const uint8_t x = a[n], y = b[n], z = c[n];
            const uint8_t msb = (uint8_t)((x | y | z) >> 7);
            uint8_t est; uint32_t be;
            bit_vote_correct(x & 0x7f, y & 0x7f, z & 0x7f, &est, &be);
            if (!kAllowed.ok[est]) { stopped = true; break; }
            msg[n] = est; cnt[n] = 3; errs[n] = (uint8_t)(be + msb);
        }
    } else if (nb == 2) {
        // (the first and the second burst of a message are two calls in three: the same stretch for one and two bursts)
        const uint8_t *a = bursts[0].data, *b = bursts[1].data;
        for (; n < common; ++n) {
            const uint8_t x = a[n], y = b[n];
            const uint8_t msb = (uint8_t)((x | y) >> 7);
            uint8_t est; uint32_t be;
            bit_vote_detect(x & 0x7f, y & 0x7f, &est, &be);
            if (!kAllowed.ok[est]) { stopped = true; break; }
            msg[n] = est; cnt[n] = 2; errs[n] = (uint8_t)(be + msb);
        }
    } else if (nb == 1) {
        const uint8_t *a = bursts[0].data;
        for (; n < common; ++n) {
            const uint8_t x = a[n], est = (uint8_t)(x & 0x7f);
            if (!kAllowed.ok[est]) { stopped = true; break; }
            msg[n] = est; cnt[n] = 1; errs[n] = (uint8_t)(x >> 7);
        }
    }
    while (!stopped && n < kMaxMessageLength) {
        uint8_t cur[3]; uint32_t k = 0; bool msb = false;
        for (size_t i = 0; i < nb; ++i)
            if (n < bursts[i].len) cur[k++] = bursts[i].data[n];
        for (uint32_t i = 0; i < k; ++i) { msb |= (cur[i] & 0x80) != 0; cur[i] &= 0x7f; }
        if (k == 0) break;
        uint8_t est; uint32_t be = 0;
        if (k == 1) est = cur[0];
        else if (k == 2) bit_vote_detect(cur[0], cur[1], &est, &be);
        else bit_vote_correct(cur[0], cur[1], cur[2], &est, &be);
        if (!kAllowed.ok[est]) break;
        msg[n] = est; cnt[n] = (uint8_t)k; errs[n] = (uint8_t)(be + (msb ? 1u : 0u));
        ++n;
    }
    // combine rx/combiner.rs:32-80
    if (n == 0) return false;
    size_t good = 0;
    while (good < n && cnt[good] >= 2) ++good;           // truncate_bytes_with_reference :264-273
    parse_message(msg, good, errs, cnt, res);
    if (res->kind != SAME_TRANSPORT_MSG_ERR) return true;
    if (n >= 2 && msg[0] == 'N' && msg[1] == 'N') {      // Fast EOM :251-258
        clear_result(res);
        res->kind = SAME_TRANSPORT_MSG_END;
        return true;
    }
    return good != 0;
}

void TransportRef::prune_history(uint64_t now)
{
    // rx/assembler.rs:362-368: retain unexpired entries, then keep at most the two newest
    uint32_t w = 0;
    for (uint32_t i = 0; i < h_.nhist; ++i)
        if (!(h_.hist_deadline[i] <= now)) { if (w != i) { c_.history[w] = c_.history[i]; h_.hist_deadline[w] = h_.hist_deadline[i]; } ++w; }
    h_.nhist = (uint8_t)w;
    while (h_.nhist > 2) {
        for (uint32_t i = 1; i < h_.nhist; ++i) { c_.history[i - 1] = c_.history[i]; h_.hist_deadline[i - 1] = h_.hist_deadline[i]; }
        --h_.nhist;
    }
}
void TransportRef::accept(const MessageResult &m, uint64_t now)
{
    // PendingResult::accept rx/assembler.rs:294-328
    const uint64_t deadline = (m.kind == SAME_TRANSPORT_MSG_END) ? now : now + max_interburst_symbols();
    if (h_.pending) {
        bool replace;
        if (c_.pend.kind == SAME_TRANSPORT_MSG_ERR) replace = true;
        else if (c_.pend.kind == SAME_TRANSPORT_MSG_END && m.kind == SAME_TRANSPORT_MSG_START) replace = true;
        else if (c_.pend.kind == SAME_TRANSPORT_MSG_START && m.kind == SAME_TRANSPORT_MSG_START)
            replace = m.voting_bytes >= c_.pend.voting_bytes;
        else replace = false;
        if (replace) { c_.pend = m; h_.pend_deadline = deadline; }
    } else {
        h_.pending = 1; c_.pend = m; h_.pend_deadline = deadline;
    }
}
uint32_t TransportRef::idle(uint64_t now, MessageResult *msg)
{
    // rx/assembler.rs:205-234
    if (h_.nhist > 2 || (h_.nhist && h_.hist_deadline[0] <= now)) prune_history(now);   // deadlines are pushed in increasing order
    if (h_.pending && h_.pend_deadline <= now) {         // PendingResult::poll :336-345
        *msg = c_.pend;
        h_.pending = 0;
        if (msg->kind != SAME_TRANSPORT_MSG_ERR) {
            h_.have_prev = 1; c_.prev = *msg; c_.prev_deadline = now + max_history_duration();
        }
        return msg->kind;
    }
    return h_.nhist == 0 ? SAME_TRANSPORT_IDLE : SAME_TRANSPORT_ASSEMBLING;
}
uint32_t TransportRef::assemble(const uint8_t *burst, size_t n, uint64_t now, MessageResult *msg)
{
    // rx/assembler.rs:154-184
    if (n == 0) return idle(now, msg);
    prune_history(now);
    if (h_.have_prev && c_.prev_deadline <= now) h_.have_prev = 0;   // prune_previous :371-376
    h_.hist_deadline[h_.nhist] = now + max_history_duration();
    BurstBuf &t = c_.history[h_.nhist++];                           // at most 2 survive the prune
    t.len = (uint32_t)std::min(n, kMaxMessageLength);
    std::memcpy(t.data, burst, t.len);
    MessageResult res;
    if (combine(c_.history, h_.nhist, &res)) {
        // deduplicate :245-265: messages are duplicates when string-equal
        bool keep = true;
        if (res.kind != SAME_TRANSPORT_MSG_ERR && h_.have_prev && c_.prev.same_text(res)) keep = false;
        if (keep) accept(res, now);
    }
    return idle(now, msg);
}

void TransportRef::reset()
{
    h_.nhist = 0; h_.pending = 0; h_.have_prev = 0;
    h_.state_kind = SAME_TRANSPORT_IDLE; clear_result(&c_.state_msg);
    h_.have_force_eom = 0; h_.dirty = 1;
    h_.have_polled = 0; h_.last_polled_symbol = 0;
}

bool TransportRef::on_link_event(uint32_t kind, uint64_t sample_counter, uint64_t symbol_count,
                                 const uint8_t *bytes, uint32_t len, uint32_t input_rate, same_rx_event *out)
{
    // process_transportlayer receiver.rs:291-333
    const uint64_t kMaxMessageDurationSecs = 135;        // receiver.rs:496
    MessageResult msg;          // trivially constructed: no heap, text left uninitialised
    uint32_t st;
    // The reference polls once per symbol.  A wake-up tick that lands on the symbol of a link
    // event just handled (e.g. a deadline that expired while the link was Searching is served
    // by the NoCarrier transition itself) must not poll a second time.
    if (kind == kDevTick && h_.have_polled && symbol_count == h_.last_polled_symbol) return false;
    if (kind == SAME_LINK_BURST || kind == SAME_LINK_NO_CARRIER || kind == kDevTick) {
        h_.have_polled = 1; h_.last_polled_symbol = symbol_count;
    }
    if (kind == SAME_LINK_BURST) {
        st = assemble(bytes, len, symbol_count, &msg);
    } else if (kind == SAME_LINK_NO_CARRIER || kind == kDevTick) {
        if (h_.have_force_eom && sample_counter > h_.force_eom_at) {
            st = SAME_TRANSPORT_MSG_END; clear_result(&msg); msg.kind = st;
        } else {
            st = idle(symbol_count, &msg);
        }
    } else {
        return false;
    }
    if (st == SAME_TRANSPORT_MSG_START) {
        h_.have_force_eom = 1; h_.dirty = 1;
        h_.force_eom_at = sample_counter + kMaxMessageDurationSecs * (uint64_t)input_rate;
    } else if (st == SAME_TRANSPORT_MSG_END) {
        if (h_.have_force_eom) h_.dirty = 1;
        h_.have_force_eom = 0;
    }
    const bool is_msg = st >= SAME_TRANSPORT_MSG_START;
    // (the message text is compared, and the cold record touched, only between two message states)
    const bool same = (st == h_.state_kind) && (!is_msg || msg == c_.state_msg);
    if (same) return false;
    const bool was_msg = h_.state_kind >= SAME_TRANSPORT_MSG_START;
    h_.state_kind = st;
    if (is_msg) c_.state_msg = msg; else if (was_msg) clear_result(&c_.state_msg);
    std::memset(out, 0, sizeof(*out));          // payload bytes past `len` are zero, never stack contents
    out->kind = st;
    out->sample_counter = sample_counter;
    out->symbol_count = symbol_count;
    if (st == SAME_TRANSPORT_MSG_START) {
        out->len = msg.len;
        std::memcpy(out->bytes, msg.text, std::min<size_t>(msg.len, SAME_EVENT_MAX_BYTES));
        out->aux = msg.voting_bytes; out->aux2 = msg.parity_errors;
    } else if (st == SAME_TRANSPORT_MSG_ERR) {
        out->aux = msg.err;
    }
    return true;
}

}  // namespace same
