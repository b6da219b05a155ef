"""Python mirror of the `sameplace` crate's API over the C ABI of include/same_place.h.

Names and argument meaning follow crates/sameplace/src/message.rs and message/*.rs so that the
tests read like the reference's own: `Message.try_from`, `MessageHeader.new`, `.originator()`,
`.event().phenomenon()`, `.issue_datetime(received)`, ...  All work is done by the shared
library (sameold_amd/csrc/same_place.cpp); nothing here re-implements it.
"""
from __future__ import annotations

import ctypes as C
import datetime as _dt
import enum
from typing import Iterator, Optional, Sequence, Tuple

from .receiver import SameError, load_library

MSG_START, MSG_END = 1, 2
EPREFIX, ENOTASCII, EMALFORMED, EDATE = -1, -2, -3, -4


class MessageDecodeErr(ValueError):
    """message.rs:86-98"""
    UnrecognizedPrefix, NotAscii, Malformed = EPREFIX, ENOTASCII, EMALFORMED

    def __init__(self, code: int):
        self.code = code
        text = {EPREFIX: "invalid SAME header: unrecognized prefix",
                ENOTASCII: "invalid SAME header: message contains non-ASCII characters",
                EMALFORMED: "invalid SAME header: message text does not match required pattern"}[code]
        super().__init__(text)

    def __eq__(self, other):
        return isinstance(other, MessageDecodeErr) and other.code == self.code

    def __hash__(self):
        return hash(self.code)


class InvalidDateErr(ValueError):
    """message.rs:149-151"""
    def __init__(self):
        super().__init__("message issuance time not valid for its receive time")


class _Header(C.Structure):
    _fields_ = [("len", C.c_uint32), ("offset_time", C.c_uint32), ("parity_error_count", C.c_uint32),
                ("voting_byte_count", C.c_uint32), ("text", C.c_char * 272)]


_declared = False


def _lib() -> C.CDLL:
    global _declared
    L = load_library()
    if _declared:
        return L
    H, cp, sz, i64 = C.POINTER(_Header), C.c_char_p, C.c_size_t, C.c_int64
    u8p = C.POINTER(C.c_uint8)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, list(args)

    sig("same_message_parse", C.c_int, cp, sz, H)
    sig("same_message_as_str", cp, C.c_int, H)
    sig("same_header_new", C.c_int, cp, sz, H)
    sig("same_header_new_with_error_info", C.c_int, cp, sz, u8p, sz, u8p, sz, H)
    sig("same_header_originator_str", sz, H, cp)
    sig("same_header_originator", C.c_int, H)
    sig("same_header_event_str", sz, H, cp)
    sig("same_header_event", None, H, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("same_header_location_count", sz, H)
    sig("same_header_location", sz, H, sz, cp, sz)
    sig("same_header_valid_duration_fields", None, H, u8p, u8p)
    sig("same_header_issue_daytime_fields", None, H, C.POINTER(C.c_uint16), u8p, u8p)
    sig("same_header_callsign", sz, H, cp, sz)
    sig("same_header_is_national", C.c_int, H)
    sig("same_header_issue_datetime", C.c_int, H, i64, C.POINTER(i64))
    sig("same_header_purge_datetime", C.c_int, H, i64, C.POINTER(i64))
    sig("same_header_is_expired_at", C.c_int, H, i64)
    sig("same_calculate_issue_time", C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(i64))
    sig("same_calculate_expire_time", C.c_int, i64, i64, C.POINTER(i64))
    sig("same_originator_from_org_and_call", C.c_int, cp, sz, cp, sz)
    sig("same_originator_display_str", cp, C.c_int)
    sig("same_originator_code_str", cp, C.c_int)
    sig("same_event_parse", None, cp, sz, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("same_event_display", sz, C.c_int, C.c_int, C.c_int, cp, sz)
    sig("same_event_is_test", C.c_int, C.c_int, C.c_int)
    sig("same_event_is_unrecognized", C.c_int, C.c_int, C.c_int)
    sig("same_phenomenon_brief_str", cp, C.c_int)
    sig("same_phenomenon_pattern_str", cp, C.c_int)
    sig("same_phenomenon_is_national", C.c_int, C.c_int)
    sig("same_phenomenon_is_test", C.c_int, C.c_int)
    sig("same_phenomenon_is_weather", C.c_int, C.c_int)
    sig("same_significance_from", C.c_int, cp, sz)
    sig("same_significance_display_str", cp, C.c_int)
    sig("same_significance_code_str", cp, C.c_int)
    _declared = True
    return L


def _b(s) -> bytes:
    return s if isinstance(s, (bytes, bytearray)) else str(s).encode("utf-8")


def _utc(ts: int) -> _dt.datetime:
    return _dt.datetime.fromtimestamp(ts, tz=_dt.timezone.utc)


def _ts(t) -> int:
    if isinstance(t, _dt.datetime):
        if t.tzinfo is None:
            t = t.replace(tzinfo=_dt.timezone.utc)
        return int(t.timestamp())
    return int(t)


# ------------------------------------------------------------------------------------------
class SignificanceLevel(enum.IntEnum):
    """message/significance.rs:75-123 (`repr(u8)` order)"""
    Test = 0
    Statement = 1
    Emergency = 2
    Watch = 3
    Warning = 4
    Unknown = 5

    @classmethod
    def from_code(cls, code: str) -> "SignificanceLevel":
        c = _b(code)
        return cls(_lib().same_significance_from(c, len(c)))

    def as_display_str(self) -> str:
        return _lib().same_significance_display_str(int(self)).decode()

    def as_code_str(self) -> str:
        return _lib().same_significance_code_str(int(self)).decode()

    def __str__(self):
        return self.as_display_str()


class Phenomenon(enum.IntEnum):
    """message/phenomenon.rs:75-374 (declaration order)"""
    NationalEmergency = 0
    NationalInformationCenter = 1
    NationalAudibleTest = 2
    NationalPeriodicTest = 3
    NationalSilentTest = 4
    RequiredMonthlyTest = 5
    RequiredWeeklyTest = 6
    AdministrativeMessage = 7
    Avalanche = 8
    Blizzard = 9
    BlueAlert = 10
    ChildAbduction = 11
    CivilDanger = 12
    CivilEmergency = 13
    CoastalFlood = 14
    DustStorm = 15
    Earthquake = 16
    Evacuation = 17
    ExtremeWind = 18
    Fire = 19
    FlashFlood = 20
    FlashFreeze = 21
    Flood = 22
    Freeze = 23
    HazardousMaterials = 24
    HighWind = 25
    Hurricane = 26
    HurricaneLocalStatement = 27
    LawEnforcementWarning = 28
    LocalAreaEmergency = 29
    NetworkMessageNotification = 30
    TelephoneOutage = 31
    NuclearPowerPlant = 32
    PracticeDemoWarning = 33
    RadiologicalHazard = 34
    SevereThunderstorm = 35
    SevereWeather = 36
    ShelterInPlace = 37
    SnowSquall = 38
    SpecialMarine = 39
    SpecialWeatherStatement = 40
    StormSurge = 41
    Tornado = 42
    TropicalStorm = 43
    Tsunami = 44
    Volcano = 45
    WinterStorm = 46
    Unrecognized = 47

    def as_brief_str(self) -> str:
        return _lib().same_phenomenon_brief_str(int(self)).decode()

    def as_full_pattern_str(self) -> str:
        return _lib().same_phenomenon_pattern_str(int(self)).decode()

    def is_national(self) -> bool:
        return bool(_lib().same_phenomenon_is_national(int(self)))

    def is_test(self) -> bool:
        return bool(_lib().same_phenomenon_is_test(int(self)))

    def is_weather(self) -> bool:
        return bool(_lib().same_phenomenon_is_weather(int(self)))

    def is_non_weather(self) -> bool:
        return not self.is_weather()

    def is_unrecognized(self) -> bool:
        return self is Phenomenon.Unrecognized

    def is_recognized(self) -> bool:
        return not self.is_unrecognized()

    def __str__(self):
        return self.as_brief_str()


class Originator(enum.IntEnum):
    """message/originator.rs:47-87"""
    Unknown = 0
    PrimaryEntryPoint = 1
    CivilAuthority = 2
    NationalWeatherService = 3
    EnvironmentCanada = 4
    BroadcastStation = 5

    @classmethod
    def from_org_and_call(cls, org: str, call: str) -> "Originator":
        o, c = _b(org), _b(call)
        return cls(_lib().same_originator_from_org_and_call(o, len(o), c, len(c)))

    def as_display_str(self) -> str:
        return _lib().same_originator_display_str(int(self)).decode()

    def as_code_str(self) -> str:
        return _lib().same_originator_code_str(int(self)).decode()

    def __str__(self):
        return self.as_display_str()


class EventCode:
    """message/eventcode.rs:71-197"""
    __slots__ = ("_ph", "_sg")

    def __init__(self, code: str = ""):
        c = _b(code)
        ph, sg = C.c_int(), C.c_int()
        _lib().same_event_parse(c, len(c), C.byref(ph), C.byref(sg))
        self._ph, self._sg = Phenomenon(ph.value), SignificanceLevel(sg.value)

    @classmethod
    def from_code(cls, code: str) -> "EventCode":
        return cls(code)

    def phenomenon(self) -> Phenomenon:
        return self._ph

    def significance(self) -> SignificanceLevel:
        return self._sg

    def is_test(self) -> bool:
        return bool(_lib().same_event_is_test(int(self._ph), int(self._sg)))

    def is_unrecognized(self) -> bool:
        return bool(_lib().same_event_is_unrecognized(int(self._ph), int(self._sg)))

    def _display(self, alternate: bool) -> str:
        buf = C.create_string_buffer(128)
        n = _lib().same_event_display(int(self._ph), int(self._sg), int(alternate), buf, len(buf))
        return buf.raw[:n].decode()

    def to_display_string(self) -> str:
        return self._display(False)

    def __str__(self):
        return self._display(False)

    def __format__(self, spec):
        """`format(evt, "#")` is Rust's `{:#}`: the phenomenon without its significance."""
        return self._display(spec == "#")

    def __eq__(self, other):
        return isinstance(other, EventCode) and (self._ph, self._sg) == (other._ph, other._sg)

    def __hash__(self):
        return hash((self._ph, self._sg))

    def __lt__(self, other):          # Ord for EventCode: by significance (eventcode.rs:178-182)
        return self._sg < other._sg

    def __repr__(self):
        return f"EventCode({self._ph.name}, {self._sg.name})"


# ------------------------------------------------------------------------------------------
class MessageHeader:
    """message.rs:160-660"""

    def __init__(self, raw: _Header):
        self._h = raw

    # ---- constructors ---------------------------------------------------------------------
    @classmethod
    def new(cls, message) -> "MessageHeader":
        m = _b(message)
        h = _Header()
        rc = _lib().same_header_new(m, len(m), C.byref(h))
        if rc < 0:
            raise MessageDecodeErr(rc)
        return cls(h)

    @classmethod
    def new_with_errors(cls, message, error_counts: Sequence[int]) -> "MessageHeader":
        return cls.new_with_error_info(message, error_counts, ())

    @classmethod
    def new_with_error_info(cls, message, error_counts: Sequence[int], burst_counts: Sequence[int]) -> "MessageHeader":
        m = _b(message)
        h = _Header()
        e = (C.c_uint8 * max(len(error_counts), 1))(*error_counts)
        b = (C.c_uint8 * max(len(burst_counts), 1))(*burst_counts)
        rc = _lib().same_header_new_with_error_info(m, len(m), e, len(error_counts), b, len(burst_counts), C.byref(h))
        if rc < 0:
            raise MessageDecodeErr(rc)
        return cls(h)

    try_from = new

    # ---- accessors ------------------------------------------------------------------------
    def message(self) -> str:
        return self._h.text[: self._h.len].decode("ascii")

    as_str = message

    def __str__(self):
        return self.message()

    def _p(self):
        return C.byref(self._h)

    def originator_str(self) -> str:
        buf = C.create_string_buffer(4)
        n = _lib().same_header_originator_str(self._p(), buf)
        return buf.raw[:n].decode()

    def originator(self) -> Originator:
        return Originator(_lib().same_header_originator(self._p()))

    def event_str(self) -> str:
        buf = C.create_string_buffer(4)
        n = _lib().same_header_event_str(self._p(), buf)
        return buf.raw[:n].decode()

    def event(self) -> EventCode:
        return EventCode(self.event_str())

    def location_str_iter(self) -> Iterator[str]:
        L = _lib()
        buf = C.create_string_buffer(300)
        for i in range(L.same_header_location_count(self._p())):
            n = L.same_header_location(self._p(), i, buf, len(buf))
            yield buf.raw[:n].decode()

    def valid_duration_fields(self) -> Tuple[int, int]:
        h, m = C.c_uint8(), C.c_uint8()
        _lib().same_header_valid_duration_fields(self._p(), C.byref(h), C.byref(m))
        return h.value, m.value

    def valid_duration(self) -> _dt.timedelta:
        h, m = self.valid_duration_fields()
        return _dt.timedelta(hours=h, minutes=m)

    def issue_daytime_fields(self) -> Tuple[int, int, int]:
        d, h, m = C.c_uint16(), C.c_uint8(), C.c_uint8()
        _lib().same_header_issue_daytime_fields(self._p(), C.byref(d), C.byref(h), C.byref(m))
        return d.value, h.value, m.value

    def issue_datetime(self, received) -> _dt.datetime:
        out = C.c_int64()
        if _lib().same_header_issue_datetime(self._p(), _ts(received), C.byref(out)):
            raise InvalidDateErr()
        return _utc(out.value)

    def purge_datetime(self, received) -> _dt.datetime:
        out = C.c_int64()
        if _lib().same_header_purge_datetime(self._p(), _ts(received), C.byref(out)):
            raise InvalidDateErr()
        return _utc(out.value)

    def is_expired_at(self, now) -> bool:
        return bool(_lib().same_header_is_expired_at(self._p(), _ts(now)))

    def callsign(self) -> str:
        buf = C.create_string_buffer(16)
        n = _lib().same_header_callsign(self._p(), buf, len(buf))
        return buf.raw[:n].decode()

    def parity_error_count(self) -> int:
        return int(self._h.parity_error_count)

    def voting_byte_count(self) -> int:
        return int(self._h.voting_byte_count)

    def is_national(self) -> bool:
        return bool(_lib().same_header_is_national(self._p()))

    def release(self) -> str:
        return self.message()

    def __eq__(self, other):
        return isinstance(other, MessageHeader) and (self.message(), self.parity_error_count(), self.voting_byte_count()) == \
            (other.message(), other.parity_error_count(), other.voting_byte_count())

    def __hash__(self):
        return hash(self.message())

    def __repr__(self):
        return f"MessageHeader({self.message()!r})"


class Message:
    """message.rs:62-145: `StartOfMessage(MessageHeader)` or `EndOfMessage`."""

    def __init__(self, header: Optional[MessageHeader]):
        self.header = header

    EndOfMessage: "Message"

    @classmethod
    def try_from(cls, text, error_counts: Optional[Sequence[int]] = None,
                 burst_counts: Optional[Sequence[int]] = None) -> "Message":
        try:
            m = _b(text)
            if isinstance(text, (bytes, bytearray)):
                bytes(text).decode("utf-8")
        except UnicodeDecodeError:
            raise MessageDecodeErr(ENOTASCII)
        h = _Header()
        rc = _lib().same_message_parse(m, len(m), C.byref(h))
        if rc < 0:
            raise MessageDecodeErr(rc)
        if rc == MSG_END:
            return cls(None)
        if error_counts is not None or burst_counts is not None:
            return cls(MessageHeader.new_with_error_info(m, error_counts or (), burst_counts or ()))
        return cls(MessageHeader(h))

    def is_start(self) -> bool:
        return self.header is not None

    def as_str(self) -> str:
        return self.header.message() if self.header else "NNNN"

    def parity_error_count(self) -> int:
        return self.header.parity_error_count() if self.header else 0

    def voting_byte_count(self) -> int:
        return self.header.voting_byte_count() if self.header else 0

    def __str__(self):
        return self.as_str()

    def __eq__(self, other):
        return isinstance(other, Message) and self.header == other.header

    def __hash__(self):
        return hash(self.as_str())

    def __repr__(self):
        return f"Message({self.as_str()!r})"


Message.EndOfMessage = Message(None)


def calculate_issue_time(message: Tuple[int, int, int], received: Tuple[int, int]) -> _dt.datetime:
    """message.rs:836-862; raises InvalidDateErr."""
    out = C.c_int64()
    if _lib().same_calculate_issue_time(message[0], message[1], message[2], received[0], received[1], C.byref(out)):
        raise InvalidDateErr()
    return _utc(out.value)


def calculate_expire_time(issued, purge: _dt.timedelta) -> _dt.datetime:
    """message.rs:866-888"""
    out = C.c_int64()
    if _lib().same_calculate_expire_time(_ts(issued), int(purge.total_seconds()), C.byref(out)):
        raise InvalidDateErr()
    return _utc(out.value)


__all__ = ["Message", "MessageHeader", "MessageDecodeErr", "InvalidDateErr", "EventCode", "Originator", "Phenomenon",
           "SignificanceLevel", "calculate_issue_time", "calculate_expire_time", "SameError"]
