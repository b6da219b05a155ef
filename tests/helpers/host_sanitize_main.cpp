// Host-side C++ of the product (header text layer, transport layer, builder/config) under
// AddressSanitizer + UBSan on the CPU.  GPU sanitizers are not available on the pool; this covers
// the code that never touches the GPU.  Built and run by tests/test_host_sanitizers.py:
//   g++ -fsanitize=address,undefined  this file + same_place.cpp same_transport.cpp same_config.cpp
// Input (argv[1]): lines "kind sample_counter symbol_count hexbytes|-" = link events of one channel.
// Output: one line per transport event "kind sample_counter len text", then "OK".
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/same_place.h"
#include "../../include/same_rx.h"
#include "../../sameold_amd/csrc/same_transport.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

static void header_layer()
{
    // every prefix and a few mutilations of a long valid header: the parser must stay in bounds
    const std::string good = "ZCZC-WXR-SVR-012079-013019-013027-013075-013185-013173+0130-0462024-N0C4LL  -";
    same_header h;
    CHECK(same_message_parse(good.data(), good.size(), &h) == SAME_MSG_START);
    CHECK(same_header_location_count(&h) == 6);
    char buf[16];
    for (size_t i = 0; i < 8; ++i) (void)same_header_location(&h, i, buf, sizeof buf);      // incl. out of range
    for (size_t cap = 0; cap < 9; ++cap) { char small[9]; (void)same_header_callsign(&h, small, cap); }
    int phen = -1, sig = -1;
    same_header_event(&h, &phen, &sig);
    CHECK(phen == SAME_PHEN_SEVERE_THUNDERSTORM && sig == SAME_SIG_WARNING);
    for (int alt = 0; alt < 2; ++alt)
        for (size_t cap = 0; cap < 48; cap += 7) { char out[48]; (void)same_event_display(phen, sig, alt, out, cap); }
    int64_t t = 0;
    (void)same_header_issue_datetime(&h, 1600000000, &t);
    (void)same_header_purge_datetime(&h, 1600000000, &t);
    for (size_t n = 0; n <= good.size(); ++n) {
        std::string s = good.substr(0, n);
        (void)same_message_parse(s.data(), s.size(), &h);
        std::vector<uint8_t> errs(n, 1), counts(n / 2, 3);       // (shorter than the text on purpose)
        (void)same_header_new_with_error_info(s.data(), s.size(), errs.data(), errs.size(), counts.data(), counts.size(), &h);
    }
    for (size_t i = 0; i < good.size(); ++i) {
        std::string s = good; s[i] = (char)0xff; (void)same_message_parse(s.data(), s.size(), &h);
        s = good; s[i] = '-'; (void)same_message_parse(s.data(), s.size(), &h);
        s = good; s[i] = '+'; (void)same_message_parse(s.data(), s.size(), &h);
    }
    std::string longest = "ZCZC-EAS-RWT";
    for (int i = 0; i < 31; ++i) longest += "-012345";
    longest += "+0015-3652359-WXYZ/NWS-";
    CHECK(same_message_parse(longest.data(), longest.size(), &h) == SAME_MSG_START);
    CHECK(same_header_location_count(&h) == 31);
    std::string too_long = longest + std::string(400, 'A');
    (void)same_message_parse(too_long.data(), too_long.size(), &h);
    CHECK(same_message_parse("NNNN", 4, &h) == SAME_MSG_END);
    CHECK(same_message_parse("", 0, &h) < 0);
    for (int p = -2; p < SAME_PHEN_COUNT + 2; ++p) {
        (void)same_phenomenon_brief_str(p); (void)same_phenomenon_pattern_str(p);
        (void)same_phenomenon_is_national(p); (void)same_phenomenon_is_test(p); (void)same_phenomenon_is_weather(p);
        for (int s = -2; s < 8; ++s) { (void)same_event_is_test(p, s); (void)same_event_is_unrecognized(p, s); }
    }
    for (int o = -2; o < 8; ++o) { (void)same_originator_display_str(o); (void)same_originator_code_str(o); }
    for (int s = -2; s < 8; ++s) { (void)same_significance_display_str(s); (void)same_significance_code_str(s); }
    for (uint32_t day = 0; day < 370; day += 9) {
        int64_t issued = 0, purge = 0;
        if (same_calculate_issue_time(day, 23, 59, 2024, 366, &issued) == 0) (void)same_calculate_expire_time(issued, 99 * 3600 + 59 * 60, &purge);
    }
}

static void builder_layer()
{
    same_rx_builder *b = same_rx_builder_new(22050);
    same_rx_builder_with_timing_bandwidth(b, 0.1f, 0.5f);        // clamps locked <= unlocked
    same_rx_builder_with_squelch_power(b, 0.2f, 0.9f);
    same_rx_builder_with_frame_prefix_max_errors(b, 99);
    same_rx_builder_with_adaptive_equalizer(b, 64, 64, 0.2f, 1e-5f);
    float lim[2]; same_rx_builder_agc_gain_limits(b, lim);
    same_rx_builder *c = same_rx_builder_clone(b);
    same_rx_builder_free(b); same_rx_builder_free(c);
}

static int hexval(char c) { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; }

int main(int argc, char **argv)
{
    header_layer();
    builder_layer();
    if (argc > 1) {
        FILE *f = std::fopen(argv[1], "r");
        if (!f) { std::perror(argv[1]); return 2; }
        same::Transport tr;
        tr.reset();
        // argv[2] == "synth": the time-parallel mode's way -- only link events come from the device, the poll
        // instants of the transport layer are synthesised on the host (same::TickSynth), exactly as
        // same_batch.cpp's harvest does; argv[3] = sample counter at the end of the input
        const bool synth = argc > 2 && std::strcmp(argv[2], "synth") == 0;
        same::TickSynth ts;
        ts.reset();
        const double sps = 22050.0 / 520.83;
        auto print = [](const same_rx_event &ev) {
            std::printf("%u %" PRIu64 " %u ", ev.kind, ev.sample_counter, ev.len);
            std::fwrite(ev.bytes, 1, ev.len < SAME_EVENT_MAX_BYTES ? ev.len : SAME_EVENT_MAX_BYTES, stdout);
            std::printf("\n");
        };
        auto poll = [&](uint64_t psym, uint64_t pt) {
            same_rx_event tev;
            if (tr.on_link_event(same::kDevTick, pt, psym, nullptr, 0, 22050, &tev)) print(tev);
        };
        char hex[1024];
        unsigned kind; unsigned long long sc, sym;
        while (std::fscanf(f, "%u %llu %llu %1023s", &kind, &sc, &sym, hex) == 4) {
            std::vector<uint8_t> bytes;
            if (hex[0] != '-') for (size_t i = 0; hex[i] && hex[i + 1]; i += 2) bytes.push_back((uint8_t)(hexval(hex[i]) * 16 + hexval(hex[i + 1])));
            same_rx_event ev;
            if (synth) {
                if (kind >= 8) continue;                      // no device ticks in this mode
                ts.run_until(sym, sc, sps, tr.force_eom_at(), poll);
                if (tr.on_link_event(kind, sc, sym, bytes.data(), (uint32_t)bytes.size(), 22050, &ev)) print(ev);
                ts.after_event(kind, sym, sc, same::max_interburst_symbols(), same::max_history_duration());
                continue;
            }
            if (tr.on_link_event(kind, sc, sym, bytes.data(), (uint32_t)bytes.size(), 22050, &ev)) {
                std::printf("%u %" PRIu64 " %u ", ev.kind, ev.sample_counter, ev.len);
                std::fwrite(ev.bytes, 1, ev.len < SAME_EVENT_MAX_BYTES ? ev.len : SAME_EVENT_MAX_BYTES, stdout);
                std::printf("\n");
            }
            (void)tr.force_eom_at(); (void)tr.force_eom_dirty();
        }
        std::fclose(f);
        if (synth && argc > 3) {
            const uint64_t t_end = std::strtoull(argv[3], nullptr, 10);
            if (ts.link == SAME_LINK_NO_CARRIER && t_end > ts.a_t) {
                const uint64_t sym_end = ts.a_sym + (uint64_t)((double)(t_end - ts.a_t) / sps);
                ts.run_until(sym_end + 1u, t_end + 1u, sps, tr.force_eom_at(), poll);
            }
        }
    }
    if (fails) return 1;
    std::printf("OK\n");
    return 0;
}
