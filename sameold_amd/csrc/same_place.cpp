// same_place.cpp -- SAME header text layer behind include/same_place.h: what the `sameplace`
// crate does with a validated header string.  Host-side only (strings, two lookup tables and
// proleptic-Gregorian day arithmetic); citations are file:line under
// /root/reference/crates/sameplace/src/.
#include "../../include/same_place.h"

#include <cstring>

#include "same_transport.h"

namespace {

// ---- tables -----------------------------------------------------------------------------
// Phenomenon attributes (message/phenomenon.rs:75-374): brief name, full pattern ("%" stands
// for the significance word; nullptr = same as the brief name) and the national / test /
// weather properties.  Order = enum same_phenomenon.
enum : unsigned { kNat = 1, kTest = 2, kWx = 4 };
struct PhenInfo { const char *brief; const char *pattern; unsigned props; };
const PhenInfo kPhen[SAME_PHEN_COUNT] = {
    {"National Emergency", "National Emergency Message", kNat},
    {"National Information Center", nullptr, kNat},
    {"National Audible Test", nullptr, kNat | kTest},
    {"National Periodic Test", nullptr, kNat | kTest},
    {"National Silent Test", nullptr, kNat | kTest},
    {"Required Monthly Test", nullptr, kTest},
    {"Required Weekly Test", nullptr, kTest},
    {"Administrative Message", nullptr, 0},
    {"Avalanche", "Avalanche %", 0},
    {"Blizzard", "Blizzard %", kWx},
    {"Blue Alert", nullptr, 0},
    {"Child Abduction", "Child Abduction Emergency", 0},
    {"Civil Danger", "Civil Danger Warning", 0},
    {"Civil Emergency", "Civil Emergency Message", 0},
    {"Coastal Flood", "Coastal Flood %", kWx},
    {"Dust Storm", "Dust Storm %", kWx},
    {"Earthquake", "Earthquake Warning", 0},
    {"Evacuation", "Evacuation Immediate", 0},
    {"Extreme Wind", "Extreme Wind %", kWx},
    {"Fire", "Fire %", 0},
    {"Flash Flood", "Flash Flood %", kWx},
    {"Flash Freeze", "Flash Freeze %", kWx},
    {"Flood", "Flood %", kWx},
    {"Freeze", "Freeze %", kWx},
    {"Hazardous Materials", "Hazardous Materials Warning", 0},
    {"High Wind", "High Wind %", kWx},
    {"Hurricane", "Hurricane %", kWx},
    {"Hurricane Local Statement", nullptr, kWx},
    {"Law Enforcement Warning", nullptr, 0},
    {"Local Area Emergency", nullptr, 0},
    {"Network Message Notification", nullptr, 0},
    {"911 Telephone Outage", "911 Telephone Outage Emergency", 0},
    {"Nuclear Power Plant", "Nuclear Power Plant Warning", 0},
    {"Practice/Demo Warning", nullptr, 0},
    {"Radiological Hazard", "Radiological Hazard Warning", 0},
    {"Severe Thunderstorm", "Severe Thunderstorm %", kWx},
    {"Severe Weather", "Severe Weather %", kWx},
    {"Shelter In Place", "Shelter In Place Warning", 0},
    {"Snow Squall", "Snow Squall %", kWx},
    {"Special Marine", "Special Marine %", kWx},
    {"Special Weather Statement", nullptr, kWx},
    {"Storm Surge", "Storm Surge %", kWx},
    {"Tornado", "Tornado %", kWx},
    {"Tropical Storm", "Tropical Storm %", kWx},
    {"Tsunami", "Tsunami %", kWx},
    {"Volcano", "Volcano Warning", 0},
    {"Winter Storm", "Winter Storm %", kWx},
    {"Unrecognized", "Unrecognized %", 0},
};

// SignificanceLevel (message/significance.rs:75-123): code letter and display word.  Unknown
// displays as "Warning" on purpose (an unknown event is treated as the worst case).
const char *const kSigCode[6] = {"T", "S", "E", "A", "W", ""};
const char *const kSigDisplay[6] = {"Test", "Statement", "Emergency", "Watch", "Warning", "Warning"};

// Full three-character codes (eventcodes.rs:107-153)
struct Code3 { char code[4]; uint8_t phen, sig; };
const Code3 kCode3[] = {
    {"EAN", SAME_PHEN_NATIONAL_EMERGENCY, SAME_SIG_WARNING},
    {"NIC", SAME_PHEN_NATIONAL_INFORMATION_CENTER, SAME_SIG_STATEMENT},
    {"DMO", SAME_PHEN_PRACTICE_DEMO_WARNING, SAME_SIG_WARNING},
    {"NAT", SAME_PHEN_NATIONAL_AUDIBLE_TEST, SAME_SIG_TEST},
    {"NPT", SAME_PHEN_NATIONAL_PERIODIC_TEST, SAME_SIG_TEST},
    {"NST", SAME_PHEN_NATIONAL_SILENT_TEST, SAME_SIG_TEST},
    {"RMT", SAME_PHEN_REQUIRED_MONTHLY_TEST, SAME_SIG_TEST},
    {"RWT", SAME_PHEN_REQUIRED_WEEKLY_TEST, SAME_SIG_TEST},
    {"ADR", SAME_PHEN_ADMINISTRATIVE_MESSAGE, SAME_SIG_STATEMENT},
    {"BLU", SAME_PHEN_BLUE_ALERT, SAME_SIG_WARNING},
    {"CAE", SAME_PHEN_CHILD_ABDUCTION, SAME_SIG_EMERGENCY},
    {"CDW", SAME_PHEN_CIVIL_DANGER, SAME_SIG_WARNING},
    {"CEM", SAME_PHEN_CIVIL_EMERGENCY, SAME_SIG_WARNING},
    {"EQW", SAME_PHEN_EARTHQUAKE, SAME_SIG_WARNING},
    {"EVI", SAME_PHEN_EVACUATION, SAME_SIG_WARNING},
    {"FRW", SAME_PHEN_FIRE, SAME_SIG_WARNING},
    {"HMW", SAME_PHEN_HAZARDOUS_MATERIALS, SAME_SIG_WARNING},
    {"LAE", SAME_PHEN_LOCAL_AREA_EMERGENCY, SAME_SIG_EMERGENCY},
    {"LEW", SAME_PHEN_LAW_ENFORCEMENT_WARNING, SAME_SIG_WARNING},
    {"NMN", SAME_PHEN_NETWORK_MESSAGE_NOTIFICATION, SAME_SIG_STATEMENT},
    {"NUW", SAME_PHEN_NUCLEAR_POWER_PLANT, SAME_SIG_WARNING},
    {"RHW", SAME_PHEN_RADIOLOGICAL_HAZARD, SAME_SIG_WARNING},
    {"SPW", SAME_PHEN_SHELTER_IN_PLACE, SAME_SIG_WARNING},
    {"TOE", SAME_PHEN_TELEPHONE_OUTAGE, SAME_SIG_EMERGENCY},
    {"VOW", SAME_PHEN_VOLCANO, SAME_SIG_WARNING},
    {"HLS", SAME_PHEN_HURRICANE_LOCAL_STATEMENT, SAME_SIG_STATEMENT},
    {"SPS", SAME_PHEN_SPECIAL_WEATHER_STATEMENT, SAME_SIG_STATEMENT},
    {"SVR", SAME_PHEN_SEVERE_THUNDERSTORM, SAME_SIG_WARNING},
    {"SVS", SAME_PHEN_SEVERE_WEATHER, SAME_SIG_STATEMENT},
    {"TOR", SAME_PHEN_TORNADO, SAME_SIG_WARNING},
    {"FSW", SAME_PHEN_FLASH_FREEZE, SAME_SIG_WARNING},
};

// Two-character phenomenon + significance letter (eventcodes.rs:156-182)
struct Code2 { char code[3]; uint8_t phen; };
const Code2 kCode2[] = {
    {"AV", SAME_PHEN_AVALANCHE},       {"BZ", SAME_PHEN_BLIZZARD},       {"CF", SAME_PHEN_COASTAL_FLOOD},
    {"DS", SAME_PHEN_DUST_STORM},      {"EW", SAME_PHEN_EXTREME_WIND},   {"FF", SAME_PHEN_FLASH_FLOOD},
    {"FL", SAME_PHEN_FLOOD},           {"FZ", SAME_PHEN_FREEZE},         {"HU", SAME_PHEN_HURRICANE},
    {"HW", SAME_PHEN_HIGH_WIND},       {"SM", SAME_PHEN_SPECIAL_MARINE}, {"SQ", SAME_PHEN_SNOW_SQUALL},
    {"SS", SAME_PHEN_STORM_SURGE},     {"SV", SAME_PHEN_SEVERE_THUNDERSTORM}, {"TO", SAME_PHEN_TORNADO},
    {"TR", SAME_PHEN_TROPICAL_STORM},  {"TS", SAME_PHEN_TSUNAMI},        {"WS", SAME_PHEN_WINTER_STORM},
};

// Originator (message/originator.rs:47-87): SAME code and description
const char *const kOrgCode[6] = {"", "PEP", "CIV", "WXR", "WXR", "EAS"};
const char *const kOrgDisplay[6] = {"Unknown Originator", "Primary Entry Point System", "Civil authorities",
                                    "National Weather Service", "Environment Canada",
                                    "Broadcast station or cable system"};

// ---- helpers ----------------------------------------------------------------------------
constexpr size_t kOffOrg = 5, kOffEvt = 9, kOffArea = 13;            // message.rs:656-658
constexpr size_t kPlusValid = 1, kPlusIssue = 6, kPlusCall = 14;     // message.rs:659-661

size_t copy_out(const char *src, size_t n, char *out, size_t cap)
{
    if (out && cap) {
        const size_t m = n < cap ? n : cap;
        std::memcpy(out, src, m);
        if (n < cap) out[n] = '\0';
    }
    return n;
}

bool str_eq(const char *a, size_t n, const char *lit) { return std::strlen(lit) == n && std::memcmp(a, lit, n) == 0; }

uint32_t parse_digits(const char *p, size_t n)
{
    uint32_t v = 0;
    for (size_t i = 0; i < n; ++i) v = v * 10u + (uint32_t)(p[i] - '0');
    return v;
}

int clamp_phen(int p) { return (p < 0 || p >= SAME_PHEN_COUNT) ? SAME_PHEN_UNRECOGNIZED : p; }
int clamp_sig(int s) { return (s < 0 || s > SAME_SIG_UNKNOWN) ? SAME_SIG_UNKNOWN : s; }

int64_t floor_div(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
// days since 1970-01-01 of January 1st of `year` (proleptic Gregorian, any sign)
int64_t days_to_year_any(int64_t year)
{
    const int64_t y = year - 1;
    return y * 365 + floor_div(y, 4) - floor_div(y, 100) + floor_div(y, 400) - 719162;   // 719162 days from 0001-01-01 to 1970-01-01
}
bool is_leap(int64_t y) { return (y % 4 == 0 && y % 100 != 0) || y % 400 == 0; }

// (year, ordinal day) of a UNIX time, as chrono's Datelike::year()/ordinal() give for Utc
void year_ordinal(int64_t unix_seconds, int32_t *year, uint32_t *ordinal)
{
    const int64_t day = floor_div(unix_seconds, 86400);
    int64_t y = 1970 + floor_div(day * 400, 146097);
    while (days_to_year_any(y) > day) --y;
    while (days_to_year_any(y + 1) <= day) ++y;
    *year = (int32_t)y;
    *ordinal = (uint32_t)(day - days_to_year_any(y)) + 1u;
}

}  // namespace

extern "C" {

// ---- SignificanceLevel --------------------------------------------------------------------
int same_significance_from(const char *code, size_t n)
{
    // SignificanceLevel::from message/significance.rs:137-149
    if (n == 1) {
        switch (code[0]) {
        case 'T': return SAME_SIG_TEST;
        case 'S': return SAME_SIG_STATEMENT;
        case 'E': return SAME_SIG_EMERGENCY;
        case 'A': return SAME_SIG_WATCH;
        case 'W': return SAME_SIG_WARNING;
        default: break;
        }
    }
    return SAME_SIG_UNKNOWN;
}
const char *same_significance_display_str(int s) { return kSigDisplay[clamp_sig(s)]; }
const char *same_significance_code_str(int s) { return kSigCode[clamp_sig(s)]; }

// ---- Phenomenon -----------------------------------------------------------------------------
const char *same_phenomenon_brief_str(int p) { return kPhen[clamp_phen(p)].brief; }
const char *same_phenomenon_pattern_str(int p)
{
    const PhenInfo &i = kPhen[clamp_phen(p)];
    return i.pattern ? i.pattern : i.brief;            // as_full_pattern_str phenomenon.rs:473-476
}
int same_phenomenon_is_national(int p) { return (kPhen[clamp_phen(p)].props & kNat) != 0; }
int same_phenomenon_is_test(int p) { return (kPhen[clamp_phen(p)].props & kTest) != 0; }
int same_phenomenon_is_weather(int p) { return (kPhen[clamp_phen(p)].props & kWx) != 0; }

// ---- EventCode ------------------------------------------------------------------------------
void same_event_parse(const char *code, size_t n, int *phenomenon, int *significance)
{
    // parse_event eventcodes.rs:88-105: three characters, then two characters + significance
    // letter, then an unknown phenomenon with whatever significance the last letter gives
    int ph = SAME_PHEN_UNRECOGNIZED, sg = SAME_SIG_UNKNOWN;
    if (n == 3) {
        bool found = false;
        for (const Code3 &e : kCode3)
            if (std::memcmp(e.code, code, 3) == 0) { ph = e.phen; sg = e.sig; found = true; break; }
        if (!found) {
            sg = same_significance_from(code + 2, 1);
            for (const Code2 &e : kCode2)
                if (std::memcmp(e.code, code, 2) == 0) { ph = e.phen; break; }
        }
    }
    if (phenomenon) *phenomenon = ph;
    if (significance) *significance = sg;
}

size_t same_event_display(int phenomenon, int significance, int alternate, char *out, size_t cap)
{
    // Display for EventCode message/eventcode.rs:161-176
    const int p = clamp_phen(phenomenon), s = clamp_sig(significance);
    char buf[96];
    size_t n;
    if (alternate) {
        n = std::strlen(kPhen[p].brief);
        std::memcpy(buf, kPhen[p].brief, n);
    } else {
        const char *pat = same_phenomenon_pattern_str(p);
        n = std::strlen(pat);
        std::memcpy(buf, pat, n);
        if (n && buf[n - 1] == '%') {
            --n;
            const size_t m = std::strlen(kSigDisplay[s]);
            std::memcpy(buf + n, kSigDisplay[s], m);
            n += m;
        }
    }
    return copy_out(buf, n, out, cap);
}

int same_event_is_test(int phenomenon, int significance)
{ return clamp_sig(significance) == SAME_SIG_TEST || same_phenomenon_is_test(phenomenon); }
int same_event_is_unrecognized(int phenomenon, int significance)
{ return clamp_phen(phenomenon) == SAME_PHEN_UNRECOGNIZED || clamp_sig(significance) == SAME_SIG_UNKNOWN; }

// ---- Originator -----------------------------------------------------------------------------
int same_originator_from_org_and_call(const char *org, size_t n_org, const char *call, size_t n_call)
{
    // from_org_and_call message/originator.rs:91-102.  The string parser is strum's EnumString:
    // a variant without a `serialize` attribute answers to its own name.
    int o = SAME_ORG_UNKNOWN;
    if (str_eq(org, n_org, "PEP")) o = SAME_ORG_PRIMARY_ENTRY_POINT;
    else if (str_eq(org, n_org, "CIV")) o = SAME_ORG_CIVIL_AUTHORITY;
    else if (str_eq(org, n_org, "WXR")) o = SAME_ORG_NATIONAL_WEATHER_SERVICE;
    else if (str_eq(org, n_org, "EAS")) o = SAME_ORG_BROADCAST_STATION;
    else if (str_eq(org, n_org, "EnvironmentCanada")) o = SAME_ORG_ENVIRONMENT_CANADA;
    if (o == SAME_ORG_NATIONAL_WEATHER_SERVICE && n_call >= 3 && std::memcmp(call, "EC/", 3) == 0)
        o = SAME_ORG_ENVIRONMENT_CANADA;
    return o;
}
const char *same_originator_display_str(int o) { return kOrgDisplay[(o < 0 || o > 5) ? 0 : o]; }
const char *same_originator_code_str(int o) { return kOrgCode[(o < 0 || o > 5) ? 0 : o]; }

// ---- MessageHeader --------------------------------------------------------------------------
int same_header_new(const char *text, size_t n, same_header *hdr)
{
    // MessageHeader::new message.rs:181-199
    for (size_t i = 0; i < n; ++i)
        if ((unsigned char)text[i] >= 0x80) return SAME_MSG_ENOTASCII;
    size_t off = 0, len = 0;
    if (!same::check_header(reinterpret_cast<const uint8_t *>(text), n, &off, &len)) return SAME_MSG_EMALFORMED;
    if (len > SAME_HEADER_MAX) return SAME_MSG_EMALFORMED;   // cannot come out of the receiver
    if (hdr) {
        std::memset(hdr, 0, sizeof(*hdr));
        hdr->len = (uint32_t)len;
        hdr->offset_time = (uint32_t)off;
        std::memcpy(hdr->text, text, len);
    }
    return SAME_MSG_START;
}

int same_header_new_with_error_info(const char *text, size_t n, const uint8_t *error_counts, size_t n_err,
                                    const uint8_t *burst_counts, size_t n_burst, same_header *hdr)
{
    same_header h;
    const int rc = same_header_new(text, n, &h);
    if (rc < 0) return rc;
    // zip() stops at the shorter of (counts, truncated header)  message.rs:209-259
    for (size_t i = 0; i < n_err && i < h.len; ++i) h.parity_error_count += error_counts[i];
    for (size_t i = 0; i < n_burst && i < h.len; ++i) h.voting_byte_count += burst_counts[i] >= 3 ? 1u : 0u;
    if (hdr) *hdr = h;
    return rc;
}

int same_message_parse(const char *text, size_t n, same_header *hdr)
{
    // TryFrom<String> for Message message.rs:688-700
    if (n >= 5 && std::memcmp(text, "ZCZC-", 5) == 0) return same_header_new(text, n, hdr);
    if (n >= 2 && std::memcmp(text, "NN", 2) == 0) return SAME_MSG_END;
    return SAME_MSG_EPREFIX;
}

const char *same_message_as_str(int kind, const same_header *hdr)
{
    if (kind == SAME_MSG_START && hdr) return hdr->text;
    return kind == SAME_MSG_END ? "NNNN" : "";
}

size_t same_header_originator_str(const same_header *h, char out[4]) { return copy_out(h->text + kOffOrg, 3, out, 4); }
size_t same_header_event_str(const same_header *h, char out[4]) { return copy_out(h->text + kOffEvt, 3, out, 4); }

size_t same_header_callsign(const same_header *h, char *out, size_t cap)
{
    const size_t b = h->offset_time + kPlusCall, e = h->len - 1;      // message.rs:598-602
    return copy_out(h->text + b, e - b, out, cap);
}

int same_header_originator(const same_header *h)
{
    char call[16];
    const size_t nc = same_header_callsign(h, call, sizeof(call));
    return same_originator_from_org_and_call(h->text + kOffOrg, 3, call, nc);
}

void same_header_event(const same_header *h, int *phenomenon, int *significance)
{ same_event_parse(h->text + kOffEvt, 3, phenomenon, significance); }

size_t same_header_location_count(const same_header *h)
{
    // location_str().split('-') message.rs:388-390, 651-653
    size_t n = 1;
    for (size_t i = kOffArea; i < h->offset_time; ++i) n += h->text[i] == '-';
    return n;
}

size_t same_header_location(const same_header *h, size_t idx, char *out, size_t cap)
{
    size_t b = kOffArea, k = 0;
    for (size_t i = kOffArea; i <= h->offset_time; ++i) {
        if (i == h->offset_time || h->text[i] == '-') {
            if (k == idx) return copy_out(h->text + b, i - b, out, cap);
            ++k;
            b = i + 1;
        }
    }
    if (out && cap) out[0] = '\0';
    return 0;
}

void same_header_valid_duration_fields(const same_header *h, uint8_t *hours, uint8_t *minutes)
{
    const char *p = h->text + h->offset_time + kPlusValid;
    if (hours) *hours = (uint8_t)parse_digits(p, 2);
    if (minutes) *minutes = (uint8_t)parse_digits(p + 2, 2);
}

void same_header_issue_daytime_fields(const same_header *h, uint16_t *day, uint8_t *hour, uint8_t *minute)
{
    const char *p = h->text + h->offset_time + kPlusIssue;
    if (day) *day = (uint16_t)parse_digits(p, 3);
    if (hour) *hour = (uint8_t)parse_digits(p + 3, 2);
    if (minute) *minute = (uint8_t)parse_digits(p + 5, 2);
}

int same_header_is_national(const same_header *h)
{
    int ph = 0;
    same_header_event(h, &ph, nullptr);
    return h->offset_time - kOffArea == 6 && std::memcmp(h->text + kOffArea, "000000", 6) == 0 &&
           same_phenomenon_is_national(ph);
}

// ---- calendar -------------------------------------------------------------------------------
int same_calculate_issue_time(uint32_t day_of_year, uint32_t hour, uint32_t minute, int32_t rx_year,
                              uint32_t rx_day_of_year, int64_t *issued)
{
    // calculate_issue_time message.rs:836-862: a day-of-year more than 180 days away from the
    // receive date belongs to the neighbouring year
    const int32_t daydiff = (int32_t)rx_day_of_year - (int32_t)day_of_year;
    int64_t year = rx_year;
    if (daydiff >= 180) year = rx_year == INT32_MAX ? rx_year : rx_year + 1;
    else if (daydiff <= -180) year = rx_year == INT32_MIN ? rx_year : rx_year - 1;
    // yo_hms_to_utc :892-903: NaiveDate::from_yo_opt / and_hms_opt reject out-of-range fields
    if (day_of_year < 1 || day_of_year > (is_leap(year) ? 366u : 365u)) return SAME_EDATE;
    if (hour > 23 || minute > 59) return SAME_EDATE;
    if (year < -262143 || year > 262142) return SAME_EDATE;         // chrono's NaiveDate range
    if (issued) *issued = (days_to_year_any(year) + (int64_t)day_of_year - 1) * 86400 + hour * 3600 + minute * 60;
    return 0;
}

int same_calculate_expire_time(int64_t issued, int64_t valid_seconds, int64_t *purge)
{
    // calculate_expire_time message.rs:866-888: round to the nearest 15 minutes when the valid
    // time is at most one hour, otherwise to the nearest 30; chrono's duration_round sends an
    // exact tie upwards
    const int64_t t = issued + valid_seconds;
    const int64_t span = valid_seconds <= 3600 ? 900 : 1800;
    int64_t down = t % span;
    if (down < 0) down += span;
    int64_t r = t;
    if (down != 0) r = (span - down <= down) ? t + (span - down) : t - down;
    if (purge) *purge = r;
    return 0;
}

int same_header_issue_datetime(const same_header *h, int64_t received, int64_t *issued)
{
    uint16_t d; uint8_t hh, mm;
    same_header_issue_daytime_fields(h, &d, &hh, &mm);
    int32_t y; uint32_t o;
    year_ordinal(received, &y, &o);
    return same_calculate_issue_time(d, hh, mm, y, o, issued);
}

int same_header_purge_datetime(const same_header *h, int64_t received, int64_t *purge)
{
    int64_t issued = 0;
    const int rc = same_header_issue_datetime(h, received, &issued);
    if (rc) return rc;
    uint8_t hh, mm;
    same_header_valid_duration_fields(h, &hh, &mm);
    return same_calculate_expire_time(issued, (int64_t)hh * 3600 + (int64_t)mm * 60, purge);
}

int same_header_is_expired_at(const same_header *h, int64_t now)
{
    int64_t purge = 0;
    if (same_header_purge_datetime(h, now, &purge)) return 0;       // message.rs:561-567
    return purge < now;
}

}  // extern "C"
