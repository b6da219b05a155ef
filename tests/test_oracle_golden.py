"""Pins the oracle against the reference's own golden vectors.

1. sample/*.bin -> sample/*.txt at the text level (what sample/test.sh:19-56 checks),
2. the in-process synthetic tests of receiver.rs:642-705,
3. transport-layer vectors of rx/combiner.rs:280-441 and rx/assembler.rs:419-779,
4. the committed link-event fixtures (regression guard for the oracle itself).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, TEST_MESSAGE
from oracle import binding as ob

L = ob.lib()
SAMPLES = ["npt", "two_and_two", "long_message"]


def load_pcm(name):
    return np.fromfile(os.path.join(GOLDEN, f"{name}.22050.s16le.bin"), dtype="<i2")


def expected_lines(name):
    with open(os.path.join(GOLDEN, f"{name}.22050.s16le.txt")) as f:
        return [l for l in f.read().splitlines() if l != "+OK"]


@pytest.mark.parametrize("name", SAMPLES)
def test_sample_text(name):
    """samedec --rate 22050 --file X.bin prints exactly X.txt (minus the child's +OK)."""
    assert ob.samedec_lines(load_pcm(name)) == expected_lines(name)


def test_long_message_needs_flush():
    """long_message's header only appears during the EOF flush()."""
    rx = ob.Receiver(ob.samedec_config())
    kinds = [e.kind for e in rx.iter_events(load_pcm("long_message"))]
    assert ob.TRANSPORT_MSG_START not in kinds
    ev = rx.flush_first_message()
    assert ev is not None and ev.kind == ob.TRANSPORT_MSG_START
    assert ev.data().decode() == expected_lines("long_message")[0]


def make_test_burst(msg: bytes, num_bursts: int):
    """receiver.rs:611-639: 16 x 0xAB + payload, AFSK x16384, 1 s gaps, 2 s tail."""
    burst = ob.modulate_afsk(bytes([0xAB] * 16) + msg, 22050) * np.float32(16384.0)
    parts = [burst]
    for _ in range(1, num_bursts):
        parts += [np.zeros(22050, np.float32), burst]
    parts.append(np.zeros(2 * 22050, np.float32))
    return np.concatenate(parts)


def test_iter_events():
    """receiver.rs:642-675"""
    afsk = make_test_burst(TEST_MESSAGE.encode(), 1)
    cfg = ob.default_config(22050)
    L.so_config_with_timing_max_deviation(C.byref(cfg), 0.01)
    rx = ob.Receiver(cfg)
    evs = list(rx.iter_events(afsk))
    kinds = [e.kind for e in evs]
    assert kinds == [ob.LINK_SEARCHING, ob.LINK_READING, ob.LINK_BURST,
                     ob.TRANSPORT_ASSEMBLING, ob.LINK_NO_CARRIER]
    assert evs[2].data().startswith(TEST_MESSAGE.encode())


def test_top_level_receiver():
    """receiver.rs:677-705"""
    afsk = make_test_burst(TEST_MESSAGE.encode(), 3)
    cfg = ob.default_config(22050)
    L.so_config_with_timing_max_deviation(C.byref(cfg), 0.01)
    rx = ob.Receiver(cfg)
    first = None
    for ev in rx.iter_events(afsk):
        if ev.kind in (ob.TRANSPORT_MSG_START, ob.TRANSPORT_MSG_END):
            first = ev
            break
    assert first is not None and first.kind == ob.TRANSPORT_MSG_START
    assert first.data().decode() == TEST_MESSAGE
    at = rx.force_eom_at_sample()
    assert at is not None
    rx.input_sample_counter = at - 3 * rx.input_rate
    ev = rx.flush_first_message()
    assert ev is not None and ev.kind == ob.TRANSPORT_MSG_END


def test_lazy_consumption_and_chunking():
    """iter_events consumes only what it needs (receiver.rs:110-113); feeding the same
    stream in arbitrary chunks yields identical events."""
    pcm = load_pcm("npt").astype(np.float32)
    rx = ob.Receiver(ob.samedec_config())
    it = rx.iter_events(pcm)
    ev = next(it)
    assert rx.consumed == ev.sample_counter == rx.input_sample_counter
    whole = [e.as_tuple() for e in ob.Receiver(ob.samedec_config()).run(pcm)]
    rx2 = ob.Receiver(ob.samedec_config())
    chunked = []
    rng = np.random.default_rng(1)
    off = 0
    while off < len(pcm):
        n = int(rng.integers(1, 5000))
        chunked += [e.as_tuple() for e in rx2.run(pcm[off:off + n])]
        off += n
    assert chunked == whole


def test_reset_restores_initial_behaviour():
    pcm = load_pcm("npt")
    rx = ob.Receiver(ob.samedec_config())
    a = [e.as_tuple() for e in rx.run(pcm)]
    rx.reset()
    b = [e.as_tuple() for e in rx.run(pcm)]
    # AGC restarts at 1.0 after reset() but at min(1, min_gain) after new() (agc.rs:55,61):
    # the decoded bursts must agree even if the first acquisition instants differ
    assert [t[2] for t in a if t[0] == ob.LINK_BURST] == [t[2] for t in b if t[0] == ob.LINK_BURST]
    assert rx.input_sample_counter == len(pcm)


# ------------------------------------------------------------------ combiner.rs:280-441
def vote2(a, b):
    o, e = C.c_uint8(), C.c_uint32()
    L.so_bit_vote_detect(a, b, C.byref(o), C.byref(e))
    return o.value, e.value


def vote3(a, b, c):
    o, e = C.c_uint8(), C.c_uint32()
    L.so_bit_vote_correct(a, b, c, C.byref(o), C.byref(e))
    return o.value, e.value


def test_bit_votes():
    assert vote2(0xab, 0xab) == (0xab, 0) and vote2(0xff, 0xff) == (0xff, 0) and vote2(0, 0) == (0, 0)
    assert vote2(0x00, 0x01) == (0, 1) and vote2(0x02, 0x01) == (0, 2)
    assert vote2(0xff, 0xf0) == (0, 4) and vote2(0x0f, 0xf0) == (0, 8)
    assert vote3(0xab, 0xab, 0xab) == (0xab, 0) and vote3(0xff, 0xff, 0xff) == (0xff, 0)
    assert vote3(0xaa, 0xab, 0xab) == (0xab, 1) and vote3(0xa0, 0xa0, 0xaf) == (0xa0, 4)
    assert vote3(0x0f, 0xf0, 0xff) == (0xff, 8) and vote3(0x00, 0xf0, 0xff) == (0xf0, 8)
    assert vote3(0xaa, 0x55, 0xff) == (0xff, 8) and vote3(0xaa, 0x55, 0xa5) == (0xa5, 8)


def _arrs(bursts):
    n = len(bursts)
    ptrs = (C.c_char_p * n)(*bursts)
    lens = (C.c_size_t * n)(*[len(b) for b in bursts])
    return ptrs, lens, n


def estimate(bursts):
    ptrs, lens, n = _arrs(bursts)
    by, nb, er = C.create_string_buffer(268), C.create_string_buffer(268), C.create_string_buffer(268)
    k = L.so_estimate_message(ptrs, lens, n, by, nb, er)
    return by.raw[:k], list(nb.raw[:k]), list(er.raw[:k])


def combine(bursts):
    ptrs, lens, n = _arrs(bursts)
    ev = ob.Event()
    return ev if L.so_combine(ptrs, lens, n, C.byref(ev)) else None


def test_estimate_message():
    assert estimate([b""]) == (b"", [], [])
    assert estimate([b"@@", b""]) == (b"", [], [])
    assert estimate([b"HIHI", b"HI"]) == (b"HIHI", [2, 2, 1, 1], [0] * 4)
    assert estimate([b"TEST", b"TESZ", b""]) == (b"TES", [2, 2, 2], [0, 0, 0])
    assert estimate([b"NNNN", b"NNNN", b"ZCZC-"]) == (b"NNNN-", [3, 3, 3, 3, 1], [2, 3, 2, 3, 0])
    assert estimate([b"NNNN", b"NNNNB", b"ZC"]) == (b"NNNNB", [3, 3, 2, 2, 1], [2, 3, 0, 0, 0])
    assert estimate([bytes([0xce, ord("N")]), b"NN"]) == (b"NN", [2, 2], [1, 0])
    assert estimate([bytes([0xce, ord("N")]), b"NN", bytes([ord("N"), 0xce])]) == (b"NN", [3, 3], [1, 1])


def test_combine():
    MESSAGE = b"ZCZC-EAS-DMO-999000+0015-0011122-NOCALL00-"
    CORRUPT = b"ZKZK-EAS-DMO-999000+0015-0011122-NOCALL00-"
    assert combine([MESSAGE]) is None
    assert combine([b"NNZZ"]).kind == ob.TRANSPORT_MSG_END
    ev = combine([MESSAGE, MESSAGE[:16]])
    assert ev.kind == ob.TRANSPORT_MSG_ERR and ev.aux == 3  # Malformed
    ev = combine([b"NOPE", b"NOPE"])
    assert ev.kind == ob.TRANSPORT_MSG_ERR and ev.aux == 2  # UnrecognizedPrefix
    ev = combine([MESSAGE, MESSAGE])
    assert ev.kind == ob.TRANSPORT_MSG_START and ev.data() == MESSAGE and ev.aux == 0
    ev = combine([MESSAGE, MESSAGE, CORRUPT])
    assert ev.data() == MESSAGE and ev.aux == len(MESSAGE) and ev.aux2 == 2
    ev = combine([b"NNZZ", MESSAGE, MESSAGE])
    assert ev.data() == MESSAGE and ev.aux == 4


def test_check_header():
    def chk(s):
        a, b = C.c_size_t(), C.c_size_t()
        rc = L.so_check_header(s, len(s), C.byref(a), C.byref(b))
        return None if rc else (a.value, b.value)
    m = b"ZCZC-WXR-RWT-012345-567890-888990+0015-0321115-KLOX/NWS-"
    assert chk(m) == (m.index(b"+"), len(m))
    assert chk(m + b"garbage") == (m.index(b"+"), len(m))       # truncated to match end
    assert chk(b"ZCZC-WXR-RWT+0015-0321115-KLOX/NWS-") is None   # needs >= 1 location
    assert chk(b"ZCZC-WXR-RWT-012345+0015-0321115-KL-") is None  # callsign .{3,8}
    assert chk(b"ZCZC-WX1-RWT-012345+0015-0321115-KLOX-") is None
    long8 = b"ZCZC-EAS-DMO-999000+0015-0011122-NOCALL00-"
    assert chk(long8) == (long8.index(b"+"), len(long8))
    # greedy .{3,8}: the longest callsign followed by '-' wins
    two = b"ZCZC-EAS-DMO-999000+0015-0011122-ABC-DEF-"
    assert chk(two) == (two.index(b"+"), len(two))


# ------------------------------------------------------------------ assembler.rs:419-779
ONE_SECOND = 520
BURST_TIMEOUT = int(np.float32(1.31) * np.float32(520.83))
ALMOST_TIMEOUT = int(np.float32(1.2) * np.float32(520.83))
EOM = b"NNNN"
GOOD = b"ZCZC-EAS-DMO-999000+0015-0011122-NOCALL00-"
ERRS = b"ZCZK-EAS-DMF-999!00+0015-0011122-NOCALL00-KXYZ"
LONGEST = TEST_MESSAGE.encode()
I, A, MS, ME = ob.TRANSPORT_IDLE, ob.TRANSPORT_ASSEMBLING, ob.TRANSPORT_MSG_START, ob.TRANSPORT_MSG_END


def simulate(seq):
    asm = L.so_assembler_new()
    t = 0
    out = []
    for delay, data in seq:
        t += 8 * len(data) + delay
        if data:
            t += 16 * 8
        ev = ob.Event()
        L.so_assembler_assemble(asm, data, len(data), t, C.byref(ev))
        out.append((ev.kind, ev.data(), ev.aux))
    L.so_assembler_free(asm)
    return out


def test_assembler_deduplicate():
    out = simulate([(999 * ONE_SECOND, b""), (0, EOM), (ONE_SECOND, EOM), (ONE_SECOND, EOM), (12 * ONE_SECOND, EOM)])
    assert [k for k, *_ in out] == [I, ME, A, A, ME]


def test_assembler_normal_operation():
    out = simulate([(0, GOOD), (ONE_SECOND, b""), (0, GOOD), (ONE_SECOND, b""), (0, ERRS),
                    (BURST_TIMEOUT, b""), (15 * ONE_SECOND, EOM), (ONE_SECOND, EOM), (ONE_SECOND, EOM)])
    assert [k for k, *_ in out] == [A, A, A, A, A, MS, ME, A, A]
    assert out[5][1] == GOOD and out[5][2] == len(GOOD)


def test_assembler_very_long_message():
    out = simulate([(0, LONGEST), (ALMOST_TIMEOUT, b""), (0, LONGEST), (ALMOST_TIMEOUT, b""),
                    (0, LONGEST), (BURST_TIMEOUT, b"")])
    assert [k for k, *_ in out] == [A, A, A, A, A, MS]
    assert out[5][1] == LONGEST and out[5][2] == len(LONGEST)


def test_assembler_very_long_message_missing_middle():
    out = simulate([(0, LONGEST), (ALMOST_TIMEOUT, b""), (268 * 8, b""), (ALMOST_TIMEOUT, b""),
                    (0, LONGEST), (BURST_TIMEOUT, b"")])
    assert [k for k, *_ in out] == [A, A, A, A, A, MS]
    assert out[5][1] == LONGEST and out[5][2] == 0


def test_assembler_quickly_with_missing():
    out = simulate([(0, EOM), (ONE_SECOND, EOM), (ONE_SECOND, GOOD),
                    (int(np.float32(1.1) * np.float32(ONE_SECOND)), GOOD), (BURST_TIMEOUT, b""),
                    (ONE_SECOND, EOM), (ONE_SECOND, EOM)])
    assert [k for k, *_ in out] == [ME, A, A, A, MS, A, ME]
    assert out[4][2] == 4


# ------------------------------------------------------------------ committed fixtures
def test_link_event_fixtures():
    path = os.path.join(GOLDEN, "link_events.json")
    with open(path) as f:
        fix = json.load(f)
    for name in SAMPLES:
        rx = ob.Receiver(ob.samedec_config())
        got = [[e.kind, e.sample_counter, e.symbol_count, e.data().hex()] for e in rx.run(load_pcm(name))]
        assert got == fix[name]["events"], name
