"""The product's host-only C++ (header text layer, transport layer, builder) under ASan + UBSan.

GPU sanitizers are not available on the pool, and the library proper is a HIP build; the three
files below never touch the GPU and compile with plain g++, so they run here instrumented.  The
driver (tests/helpers/host_sanitize_main.cpp) walks the header parser over every prefix and a few
hundred mutilations of valid headers, the accessor tables out of range, and replays the golden
link events of the reference's recordings through the transport layer, whose output must be the
golden transport events."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden")
CSRC = os.path.join(ROOT, "sameold_amd", "csrc")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("g++ not found")
    out = str(tmp_path_factory.mktemp("sanitize") / "host_sanitize")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-Wall",
           os.path.join(ROOT, "tests", "helpers", "host_sanitize_main.cpp"),
           os.path.join(CSRC, "same_place.cpp"), os.path.join(CSRC, "same_transport.cpp"),
           os.path.join(CSRC, "same_config.cpp"), "-o", out, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


def run(driver, path=None, *more):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver] + ([path] if path else []) + list(more), capture_output=True, env=env, timeout=120)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    lines = r.stdout.split(b"\n")
    assert lines[-2] == b"OK"
    return lines[:-2]


def test_header_layer_and_builder_under_sanitizers(driver):
    run(driver)


@pytest.mark.parametrize("name", ["npt", "two_and_two", "long_message"])
def test_transport_replay_under_sanitizers(driver, tmp_path, name):
    with open(os.path.join(GOLDEN, "link_events.json")) as f:
        gold = json.load(f)[name]
    feed, want = [], []
    for kind, sample, symbol, hexbytes in gold["events"]:
        if kind < 16:
            feed.append(f"{kind} {sample} {symbol} {hexbytes or '-'}")
        else:
            # the reference produced this at a poll of its assembler: a link event of the same
            # symbol already fed, or a wake-up tick of the device (kind 8)
            feed.append(f"8 {sample} {symbol} -")
            want.append((kind, sample))
    p = tmp_path / "events.txt"
    p.write_text("\n".join(feed) + "\n")
    got = [(int(l.split(b" ")[0]), int(l.split(b" ")[1])) for l in run(driver, str(p))]
    assert got == want
    texts = [l.split(b" ", 3)[3].decode() for l in run(driver, str(p)) if l.startswith(b"18 ")]
    # (long_message's header is only delivered by the end-of-file flush, past these events)
    headers = [t for t in gold["lines"] if t.startswith("ZCZC")]
    assert len(texts) == sum(1 for k, _ in want if k == 18) and all(t in headers for t in texts)


@pytest.mark.parametrize("name", ["npt", "two_and_two", "long_message"])
def test_synthesised_transport_polls_reproduce_the_golden_transport_events(driver, tmp_path, name):
    """Time-parallel mode keeps the transport layer's poll instants on the host (same::TickSynth) instead of
    taking them from the device.  Fed the golden LINK events only, it must reproduce the golden TRANSPORT
    events of the reference's recordings: same kinds and header texts in the same order, sample counters
    within 6 symbols (the synthesised polls are interpolated from the last link event at the nominal
    symbol rate; the reference's free-running timing loop drifts by up to ~0.5 % between bursts, 3.3 symbols
    over the 682-symbol message hold of the npt recording)."""
    with open(os.path.join(GOLDEN, "link_events.json")) as f:
        gold = json.load(f)[name]
    feed = [f"{k} {t} {sym} {hx or '-'}" for k, t, sym, hx in gold["events"] if k < 16]
    want = [(k, t, bytes.fromhex(hx).decode() if hx else "") for k, t, sym, hx in gold["events"] if k >= 16]
    p = tmp_path / "events.txt"
    p.write_text("\n".join(feed) + "\n")
    t_end = max(t for _, t, _, _ in gold["events"]) + 2000
    out = run(driver, str(p), "synth", str(t_end))
    got = []
    for ln in out:
        kind, t, n, text = ln.split(b" ", 3)
        got.append((int(kind), int(t), text.decode() if int(kind) == 18 else ""))
    assert [(k, x) for k, _, x in got] == [(k, x) for k, _, x in want]
    tol = 6 * 22050 / 520.83
    assert all(abs(a[1] - b[1]) <= tol for a, b in zip(got, want)), [(a[1], b[1]) for a, b in zip(got, want)]


def test_combine_equals_the_byte_walk(tmp_path):
    """same::combine -- word-wide bit voting (eight bytes at a time), stretches specialised by burst count -- against the plain
    byte-by-byte walk of rx/combiner.rs:32-80, 154-203 on 200 000 random burst sets (bit errors, high bits, garbage bytes, ragged
    lengths), under ASan + UBSan."""
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("g++ not found")
    out = str(tmp_path / "combine_fuzz")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall",
           os.path.join(ROOT, "tests", "helpers", "combine_fuzz.cpp"), os.path.join(CSRC, "same_transport.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([out, "200000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cases equal" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
