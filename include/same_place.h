/* same_place.h -- C ABI of the SAME header text layer (SURVEY.md section 8f "next-4").
 *
 * Mirrors the public API of the `sameplace` crate that sits on top of the receiver's
 * transport layer: Message / MessageHeader (crates/sameplace/src/message.rs:62-660),
 * Originator (message/originator.rs), EventCode (message/eventcode.rs), Phenomenon
 * (message/phenomenon.rs), SignificanceLevel (message/significance.rs) and the event
 * codebook (eventcodes.rs).  It is host-side string, table and calendar work; nothing here
 * touches the GPU.  The reference panics nowhere on this path except `expect()` on fields the
 * header regex has already validated, so every function below is total.
 *
 * All strings are ASCII.  Functions that return text either return a pointer to a static
 * NUL-terminated string or copy into a caller buffer and return the length (no NUL needed
 * by the caller; one is written when it fits).
 */
#ifndef SAME_PLACE_H
#define SAME_PLACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Message::try_from / MessageHeader::new results (message.rs:62-98, 688-700) */
#define SAME_MSG_START 1        /* Message::StartOfMessage(header) */
#define SAME_MSG_END 2          /* Message::EndOfMessage */
#define SAME_MSG_EPREFIX (-1)   /* MessageDecodeErr::UnrecognizedPrefix */
#define SAME_MSG_ENOTASCII (-2) /* MessageDecodeErr::NotAscii */
#define SAME_MSG_EMALFORMED (-3)/* MessageDecodeErr::Malformed */
#define SAME_EDATE (-4)         /* InvalidDateErr (message.rs:149-151) */

#define SAME_HEADER_MAX 268     /* longest header the receiver can deliver (rx/assembler.rs:70) */

/* MessageHeader (message.rs:160-174): validated text, truncated to the end of the match */
typedef struct same_header {
    uint32_t len;               /* bytes of text */
    uint32_t offset_time;       /* index of the '+' that starts the time fields */
    uint32_t parity_error_count;
    uint32_t voting_byte_count;
    char text[SAME_HEADER_MAX + 4];   /* NUL-terminated */
} same_header;

/* Originator (message/originator.rs:47-87), declaration order */
enum same_originator {
    SAME_ORG_UNKNOWN = 0, SAME_ORG_PRIMARY_ENTRY_POINT, SAME_ORG_CIVIL_AUTHORITY,
    SAME_ORG_NATIONAL_WEATHER_SERVICE, SAME_ORG_ENVIRONMENT_CANADA, SAME_ORG_BROADCAST_STATION
};

/* SignificanceLevel (message/significance.rs:75-123), #[repr(u8)] order == SAMEDEC_SIG_NUM */
enum same_significance {
    SAME_SIG_TEST = 0, SAME_SIG_STATEMENT, SAME_SIG_EMERGENCY, SAME_SIG_WATCH, SAME_SIG_WARNING,
    SAME_SIG_UNKNOWN
};

/* Phenomenon (message/phenomenon.rs:75-374), declaration order */
enum same_phenomenon {
    SAME_PHEN_NATIONAL_EMERGENCY = 0, SAME_PHEN_NATIONAL_INFORMATION_CENTER,
    SAME_PHEN_NATIONAL_AUDIBLE_TEST, SAME_PHEN_NATIONAL_PERIODIC_TEST,
    SAME_PHEN_NATIONAL_SILENT_TEST, SAME_PHEN_REQUIRED_MONTHLY_TEST,
    SAME_PHEN_REQUIRED_WEEKLY_TEST, SAME_PHEN_ADMINISTRATIVE_MESSAGE, SAME_PHEN_AVALANCHE,
    SAME_PHEN_BLIZZARD, SAME_PHEN_BLUE_ALERT, SAME_PHEN_CHILD_ABDUCTION, SAME_PHEN_CIVIL_DANGER,
    SAME_PHEN_CIVIL_EMERGENCY, SAME_PHEN_COASTAL_FLOOD, SAME_PHEN_DUST_STORM,
    SAME_PHEN_EARTHQUAKE, SAME_PHEN_EVACUATION, SAME_PHEN_EXTREME_WIND, SAME_PHEN_FIRE,
    SAME_PHEN_FLASH_FLOOD, SAME_PHEN_FLASH_FREEZE, SAME_PHEN_FLOOD, SAME_PHEN_FREEZE,
    SAME_PHEN_HAZARDOUS_MATERIALS, SAME_PHEN_HIGH_WIND, SAME_PHEN_HURRICANE,
    SAME_PHEN_HURRICANE_LOCAL_STATEMENT, SAME_PHEN_LAW_ENFORCEMENT_WARNING,
    SAME_PHEN_LOCAL_AREA_EMERGENCY, SAME_PHEN_NETWORK_MESSAGE_NOTIFICATION,
    SAME_PHEN_TELEPHONE_OUTAGE, SAME_PHEN_NUCLEAR_POWER_PLANT, SAME_PHEN_PRACTICE_DEMO_WARNING,
    SAME_PHEN_RADIOLOGICAL_HAZARD, SAME_PHEN_SEVERE_THUNDERSTORM, SAME_PHEN_SEVERE_WEATHER,
    SAME_PHEN_SHELTER_IN_PLACE, SAME_PHEN_SNOW_SQUALL, SAME_PHEN_SPECIAL_MARINE,
    SAME_PHEN_SPECIAL_WEATHER_STATEMENT, SAME_PHEN_STORM_SURGE, SAME_PHEN_TORNADO,
    SAME_PHEN_TROPICAL_STORM, SAME_PHEN_TSUNAMI, SAME_PHEN_VOLCANO, SAME_PHEN_WINTER_STORM,
    SAME_PHEN_UNRECOGNIZED,
    SAME_PHEN_COUNT
};

/* ---- Message / MessageHeader ------------------------------------------------------ */

/* Message::try_from(String) (message.rs:688-700): "ZCZC-" -> header checks, "NN" -> EndOfMessage.
 * Returns SAME_MSG_START (with *hdr filled), SAME_MSG_END, or a negative SAME_MSG_E*. */
int same_message_parse(const char *text, size_t n, same_header *hdr);
/* Message::as_str (message.rs:105-110): header text or "NNNN" */
const char *same_message_as_str(int kind, const same_header *hdr);

/* MessageHeader::new (message.rs:181-199): ASCII check, check_header (:813-828), truncate */
int same_header_new(const char *text, size_t n, same_header *hdr);
/* MessageHeader::new_with_errors / new_with_error_info (message.rs:209-259): parity_error_count = sum of
 * error_counts over the header bytes, voting_byte_count = bytes with >= 3 bursts */
int same_header_new_with_error_info(const char *text, size_t n, const uint8_t *error_counts,
                                    size_t n_err, const uint8_t *burst_counts, size_t n_burst,
                                    same_header *hdr);

size_t same_header_originator_str(const same_header *hdr, char out[4]);   /* message.rs:301-303 */
int same_header_originator(const same_header *hdr);                       /* :281-283 */
size_t same_header_event_str(const same_header *hdr, char out[4]);        /* :368-370 */
void same_header_event(const same_header *hdr, int *phenomenon, int *significance);   /* :353-355 */
size_t same_header_location_count(const same_header *hdr);                /* :388-390 */
size_t same_header_location(const same_header *hdr, size_t i, char *out, size_t cap);
void same_header_valid_duration_fields(const same_header *hdr, uint8_t *hours, uint8_t *minutes); /* :461-469 */
void same_header_issue_daytime_fields(const same_header *hdr, uint16_t *ordinal_day,
                                      uint8_t *hour, uint8_t *minute);    /* :583-591 */
size_t same_header_callsign(const same_header *hdr, char *out, size_t cap);   /* :598-602 */
int same_header_is_national(const same_header *hdr);                      /* :639-641 */

/* Times are UTC UNIX seconds.  issue_datetime (message.rs:493-501), purge_datetime
 * (:539-544), is_expired_at (:561-567); 0 on success or SAME_EDATE. */
int same_header_issue_datetime(const same_header *hdr, int64_t received, int64_t *issued);
int same_header_purge_datetime(const same_header *hdr, int64_t received, int64_t *purge);
int same_header_is_expired_at(const same_header *hdr, int64_t now);
/* calculate_issue_time (message.rs:836-862) and calculate_expire_time (:866-888) */
int same_calculate_issue_time(uint32_t ordinal_day, uint32_t hour, uint32_t minute, int32_t rx_year,
                              uint32_t rx_ordinal_day, int64_t *issued);
int same_calculate_expire_time(int64_t issued, int64_t valid_seconds, int64_t *purge);

/* ---- Originator / EventCode / Phenomenon / SignificanceLevel ---------------------------- */

int same_originator_from_org_and_call(const char *org, size_t n_org, const char *call, size_t n_call);
const char *same_originator_display_str(int originator);   /* as_display_str */
const char *same_originator_code_str(int originator);      /* as_code_str */

/* EventCode::from (message/eventcode.rs:90-95) over parse_event (eventcodes.rs:88-105) */
void same_event_parse(const char *code, size_t n, int *phenomenon, int *significance);
/* Display for EventCode (message/eventcode.rs:161-176): alternate != 0 is the "{:#}" form */
size_t same_event_display(int phenomenon, int significance, int alternate, char *out, size_t cap);
int same_event_is_test(int phenomenon, int significance);          /* eventcode.rs:117-119 */
int same_event_is_unrecognized(int phenomenon, int significance);  /* eventcode.rs:129-132 */

const char *same_phenomenon_brief_str(int phenomenon);     /* as_brief_str */
const char *same_phenomenon_pattern_str(int phenomenon);   /* as_full_pattern_str ('%' = significance) */
int same_phenomenon_is_national(int phenomenon);
int same_phenomenon_is_test(int phenomenon);
int same_phenomenon_is_weather(int phenomenon);

int same_significance_from(const char *code, size_t n);    /* SignificanceLevel::from */
const char *same_significance_display_str(int significance);
const char *same_significance_code_str(int significance);

#ifdef __cplusplus
}
#endif
#endif
