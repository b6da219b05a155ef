"""Per-block known answers of the reference's unit tests, replayed against the oracle.

Each test names the reference test it re-expresses (file:line under
crates/sameold/src/receiver/).  Tolerances are the reference's (assert_approx_eq!
default 1e-6 unless it states another).
"""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import binding as ob

L = ob.lib()
f32 = C.c_float
F32P = C.POINTER(C.c_float)


def approx(a, b, tol=1e-6):
    return abs(float(a) - float(b)) < tol


def bytes_to_samples(data: bytes, nsps: int):
    """waveform.rs:137-155: LSb first, +-1 at the last of nsps samples per symbol."""
    out = []
    for byte in data:
        for i in range(8):
            out.extend([0.0] * (nsps - 1))
            out.append(1.0 if (byte >> i) & 1 else -1.0)
    return np.array(out, dtype=np.float32)


def bytes_to_symbols(data: bytes):
    return bytes_to_samples(data, 1)


# ---------------------------------------------------------------- dcblock.rs:118-173
def test_moving_average_simple():
    m = L.so_movavg_new(1)
    d = f32()
    assert approx(L.so_movavg_filter(m, 1.0, C.byref(d)), 1.0) and d.value == 1.0
    assert approx(L.so_movavg_filter(m, -10.0, C.byref(d)), -10.0) and d.value == -10.0
    L.so_movavg_free(m)
    m = L.so_movavg_new(2)
    assert approx(L.so_movavg_filter(m, 1.0, C.byref(d)), 0.5) and d.value == 0.0
    assert approx(L.so_movavg_filter(m, 2.0, C.byref(d)), 1.5) and d.value == 1.0
    L.so_movavg_free(m)


def test_moving_average_four():
    inp = [1.0, 2.0, -1.0, 3.0, 8.0]
    exp = [0.25, 0.75, 0.5, 1.25, 3.0]
    m = L.so_movavg_new(4)
    d = f32()
    for x, e in zip(inp, exp):
        assert approx(L.so_movavg_filter(m, x, C.byref(d)), e)
    assert d.value == 2.0
    L.so_movavg_free(m)


def test_dc_block_trivial():
    d = L.so_dcblock_new(1)
    assert L.so_dcblock_filter(d, 100.0) == 100.0
    assert L.so_dcblock_filter(d, -200.0) == -200.0
    L.so_dcblock_free(d)
    assert not L.so_dcblock_new(0)  # reference panics (dcblock.rs:74)


def test_dc_block():
    d = L.so_dcblock_new(31)
    clk = 1.0
    hist = []
    for _ in range(256):
        hist.append(L.so_dcblock_filter(d, 100.0 + clk))
        clk = -clk
    assert approx(hist[-2], 1.0, 1e-2) and approx(hist[-1], -1.0, 1e-2)
    L.so_dcblock_free(d)


# ---------------------------------------------------------------- agc.rs:105-125
def test_agc():
    a = ob.Agc()
    L.so_agc_init(C.byref(a), 0.05, 0.0, 1.0e6)
    assert a.gain == 0.0  # min(1.0, min_gain) agc.rs:55
    val = 0.0
    for _ in range(256):
        val = L.so_agc_input(C.byref(a), -2.0)
    assert approx(a.gain, 0.5) and approx(val, -1.0)
    L.so_agc_reset(C.byref(a))
    a.locked = 1
    for _ in range(16):
        val = L.so_agc_input(C.byref(a), -2.0)
    assert a.gain == 1.0 and approx(val, -2.0)


# ---------------------------------------------------------------- filter.rs:388-416
def mac(hist, coeff):
    h = np.array(hist, dtype=np.float32)
    c = np.array(coeff, dtype=np.float32)
    return L.so_mac_ff(h.ctypes.data_as(F32P), len(h), c.ctypes.data_as(F32P), len(c))


def test_multiply_accumulate():
    assert mac([], []) == 0.0
    assert mac([20.0, 1.0], [1.0]) == 1.0
    assert mac([1.0], [1.0, 20.0]) == 1.0
    assert approx(mac([20.0, 20.0], [1.0, -1.0]), 0.0)
    assert mac([10.0], [1.0, 0.0, 0.0, 0.0]) == 10.0  # identity filter :418-428


# ---------------------------------------------------------------- waveform.rs:162-187
def test_cisoid_matched_filter():
    exp_re = [-0.719973, -0.208581, 0.374184, 0.828910, 1.0]
    exp_im = [-0.694002, -0.978005, -0.927355, -0.559382, -0.0]
    re = (f32 * 5)()
    im = (f32 * 5)()
    L.so_cisoid_matched_filter(5, 0.0944807256, re, im)
    g = 2.0 / 5.0
    for i in range(5):
        assert math.hypot(re[i] - g * exp_re[i], im[i] - g * exp_im[i]) < 1e-4


def test_bytes_to_symbols():
    exp = [1, 1, -1, 1, -1, 1, -1, 1, 1, -1, -1, -1, -1, 1, -1, -1]
    assert list(bytes_to_symbols(bytes([0xAB, 0x21]))) == exp


def test_derived_constants():
    """SURVEY.md section 8 table, recomputed independently in numpy float32."""
    for fs, ntaps, dclen in [(22050, 42, 16), (48000, 92, 35), (44100, 84, 32), (11025, 21, 8), (8000, 15, 5)]:
        d = ob.derive(ob.default_config(fs))
        sps = np.float32(fs) / np.float32(520.83)
        assert d.sps == sps and d.ntaps == ntaps and d.dc_len == dclen
        assert d.samples_per_ted == sps / np.float32(2)
        assert d.agc_bw == np.float32(0.01) * sps / np.float32(fs)
    d = ob.derive(ob.default_config(22050))
    assert approx(d.alpha_unlocked, 0.79212046, 1e-6) and approx(d.beta_unlocked, 0.29600334, 1e-6)
    assert approx(d.alpha_locked, 0.46651193, 1e-6) and approx(d.beta_locked, 0.07268274, 1e-6)
    assert L.so_max_interburst_symbols() == 682 and L.so_max_history_duration() == 5652


# ---------------------------------------------------------------- demod.rs:197-227
def test_demod():
    syms = [1.0, -1.0, 1.0, -1.0, -1.0]
    fs = 11025
    bits = bytes([0b00101])  # LSb first: 1,0,1,0,0 then three more zero bits (unused)
    mod = ob.modulate_afsk(bits, fs)
    symlen = C.c_uint32()
    L.so_modulate_len(1, fs, C.byref(symlen))
    sps = symlen.value
    mod = mod[: 5 * sps]
    delay = sps // 2
    mod = np.concatenate([mod, np.zeros(delay, dtype=np.float32)])
    d = L.so_demod_new(fs)
    assert L.so_demod_ntaps(d) == 21
    for i in range(0, len(mod), delay):
        for v in mod[i:i + delay]:
            L.so_demod_push(d, float(v))
        sym = L.so_demod_demod(d)
        k = i // delay
        if k % 2 == 0:
            continue
        bit_index = (k - 1) // 2
        if bit_index < 5 and syms[bit_index] >= 0:
            assert sym >= 0.95
    L.so_demod_free(d)


# ---------------------------------------------------------------- symsync.rs:358-563
def zc(v):
    a = (f32 * 3)(*v)
    return L.so_zero_crossing_metric(a)


def test_zero_crossing_metric():
    assert approx(zc([1.0, 0.0, -1.0]), 0.0)
    assert approx(zc([-1.0, 0.0, 1.0]), 0.0)
    assert approx(zc([1.0, 1.0, 1.0]), 0.0)
    assert approx(zc([-1.0, -1.0, -1.0]), 0.0)
    assert approx(zc([0.8, 0.2, -0.8]), 0.4)
    assert approx(zc([0.8, -0.2, -0.8]), -0.4)
    # Rust signum(+0.0) = +1, signum(-0.0) = -1
    assert zc([0.0, 1.0, -0.0]) == 2.0


def alphabeta(bw):
    a, b = f32(), f32()
    L.so_compute_loop_alphabeta(bw, C.byref(a), C.byref(b))
    return a.value, b.value


def test_compute_loop_alphabeta():
    a, b = alphabeta(0.0)
    assert approx(a, 0.0) and approx(b, 0.0)
    a, b = alphabeta(0.5)
    assert approx(a, 0.99813, 1e-4) and approx(b, 0.91544, 1e-4)
    a, b = alphabeta(1.0)
    assert approx(a, 1.0, 1e-4) and approx(b, 0.99627, 1e-4)


def ted_input(t, x):
    z, s, e = f32(), f32(), f32()
    got = L.so_ted_input(C.byref(t), x, C.byref(z), C.byref(s), C.byref(e))
    return (z.value, s.value, e.value) if got else None


def test_zero_crossing_ted():
    t = ob.Ted()
    L.so_ted_reset(C.byref(t))
    assert ted_input(t, 0.8) is not None
    assert ted_input(t, 0.2) is None
    r = ted_input(t, -0.8)
    assert r[1] == np.float32(-0.8) and approx(r[2], 0.4)
    assert ted_input(t, 0.2) is None
    r = ted_input(t, 0.8)
    assert r[1] == np.float32(0.8) and approx(r[2], -0.4)


def test_timing_loop_advance():
    t = ob.Timing()
    L.so_timing_init(C.byref(t), 32.0, 0.25, 0.125)
    assert approx(t.period_inst, 16.0) and approx(t.period_max, 20.0)
    adv = lambda off, have, err=0.0: L.so_timing_advance(C.byref(t), off, have, err)
    assert approx(adv(0.0, 0), 16.0)
    assert approx(adv(0.5, 0), 16.5)
    assert approx(adv(-0.5, 0), 16.0)
    assert approx(adv(-0.5, 0), 15.5)
    L.so_timing_reset(C.byref(t))
    assert approx(t.period_inst, 16.0)
    assert approx(adv(0.0, 1, 0.0), 16.0)
    assert approx(adv(0.5, 1, 0.5 / 16.0), 16.5)
    assert approx(adv(-0.5, 1, -0.5 / 16.0), 15.5)


def timing_test(t, inp, start):
    offset = 0.0
    sa = start
    last = (0.0, 0.0, 0.0)
    L.so_timing_reset(C.byref(t))
    for _ in range(128):
        have = C.c_int()
        z, s, e = f32(), f32(), f32()
        skip = L.so_timing_input(C.byref(t), float(inp[sa]), offset, C.byref(have),
                                 C.byref(z), C.byref(s), C.byref(e))
        whole = float(np.round(np.float32(skip)))  # f32::round: half away from zero
        whole = math.floor(abs(skip) + 0.5) * (1 if skip >= 0 else -1)
        offset = float(np.float32(skip) - np.float32(whole))
        sa = (sa + int(whole)) % len(inp)
        if have.value:
            last = (z.value, s.value, e.value)
    return last


@pytest.mark.parametrize("bw,start", [(0.25, 16), (0.25, 15), (0.25, 0), (0.20, 16), (0.05, 3)])
def test_timing_loop_convergence(bw, start):
    n = 64
    inp = np.sin(np.float32(2 * np.pi) * np.arange(n, dtype=np.float32) / np.float32(n)).astype(np.float32)
    t = ob.Timing()
    L.so_timing_init(C.byref(t), 32.0, bw, 0.125)
    z, s, e = timing_test(t, inp, start)
    assert abs(s) > 0.99 and e < 1e-4


# ---------------------------------------------------------------- codesquelch.rs:499-667
SYNC = 0xABABABAB


def test_num_bit_errors_and_codecorr():
    data = C.c_uint32(0)
    syms = bytes_to_symbols(bytes([0xAB, 0xAB, 0xAB, 0xAB, 0x21]))
    out = [L.so_code_search(C.byref(data), SYNC, float(s)) for s in syms]
    for i, err in enumerate(out):
        assert (err == 0) if i == 31 else (err > 0)
    syms[19] = -syms[19]
    out = [L.so_code_search(C.byref(data), SYNC, float(s)) for s in syms]
    for i, err in enumerate(out):
        assert (err == 1) if i == 31 else (err >= 1)


def test_power_tracker():
    p = f32(0.0)
    L.so_power_track(C.byref(p), 1.0, 1.0)
    assert approx(L.so_power_track(C.byref(p), 0.5, -0.5), 0.625)
    p = f32(1.0)
    for _ in range(16):
        L.so_power_track(C.byref(p), 0.5, 1.0)
    assert approx(p.value, 1.0)


def squelch_new(max_err, open_, close, bw):
    s = ob.Squelch()
    L.so_squelch_init(C.byref(s), SYNC, max_err, open_, close, bw)
    return s


def squelch_in(s, two):
    a = (f32 * 2)(*two)
    out = (f32 * 16)()
    resync = C.c_int()
    symc = C.c_uint64()
    pwr = f32()
    st = L.so_squelch_input(C.byref(s), a, C.byref(resync), out, C.byref(symc), C.byref(pwr))
    return st, bool(resync.value), np.array(out[:], dtype=np.float32), symc.value


def test_simple_sync():
    insamp = bytes_to_samples(bytes([0xAB, 0xAB, 0xAB, 0xAB, 0x21]), 2)
    s = squelch_new(0, 0.0, 0.0, 0.1)
    assert not L.so_squelch_is_sync(C.byref(s))
    align = 0
    for chunk in range(len(insamp) // 2):
        st, resync, out, symc = squelch_in(s, insamp[2 * chunk: 2 * chunk + 2])
        if st == ob.SQ_READY:
            assert (not resync) or chunk == 31
            if chunk == 31:
                assert s.data == SYNC
            assert np.array_equal(out, insamp[align:align + 16])
            align += 16
            assert symc - 1 == chunk
    assert L.so_squelch_is_sync(C.byref(s)) and align == 32
    L.so_squelch_end(C.byref(s))
    assert not L.so_squelch_is_sync(C.byref(s))


def test_sync_with_error():
    insamp = bytes_to_samples(bytes([0xF0, 0x0B, 0xA9, 0xAB, 0xAB, 0xAB, 0x21]), 2)
    s = squelch_new(1, 0.0, 0.0, 0.1)
    align = 32
    for chunk in range(len(insamp) // 2):
        st, resync, out, _ = squelch_in(s, insamp[2 * chunk: 2 * chunk + 2])
        if st == ob.SQ_READY:
            assert (not resync) or chunk == 47
            assert np.array_equal(out, insamp[align:align + 16])
            align += 16
    assert L.so_squelch_is_sync(C.byref(s))


def test_sync_with_lots_of_errors():
    insamp = bytes_to_samples(bytes([0xAB, 0x0B, 0xA9, 0xAB, 0xAB, 0xAA, 0x21]), 2)
    s = squelch_new(3, 0.8, 0.1, 0.1)
    early = later = False
    align = 32
    for chunk in range(len(insamp) // 2):
        st, _, out, _ = squelch_in(s, insamp[2 * chunk: 2 * chunk + 2])
        if st == ob.SQ_READY:
            if chunk == 47:
                assert np.array_equal(out, insamp[align:align + 16])
                align += 16
                later = True
            else:
                early = True
    assert L.so_squelch_is_sync(C.byref(s)) and early and later


def test_power_detection():
    insamp = bytes_to_samples(bytes([0xF0, 0x0B, 0xA9, 0xAB, 0xAB, 0xAB, 0x21]), 2)
    s = squelch_new(1, 0.9, 0.5, 0.1)
    for chunk in range(len(insamp) // 2):
        squelch_in(s, insamp[2 * chunk: 2 * chunk + 2])
    assert L.so_squelch_is_sync(C.byref(s))
    seen = set()
    for _ in range(40):
        st, *_ = squelch_in(s, [0.0, 0.0])
        seen.add(st)
    assert {ob.SQ_READING, ob.SQ_DROPPED, ob.SQ_NO_CARRIER} <= seen
    assert not L.so_squelch_is_sync(C.byref(s))


# ---------------------------------------------------------------- equalize.rs:412-593
def est(e, two):
    a = (f32 * 2)(*two)
    err = f32()
    bit = L.so_equalizer_estimate_symbol(e, a, C.byref(err))
    return bool(bit), err.value


def test_estimate_symbol_simple():
    e = L.so_equalizer_new(8, 4, 0.2, 1.0e-5, 0, 0)
    L.so_equalizer_enable(e, 0)
    inp = [(0.0, 0.5), (0.0, -0.5)]
    out = [est(e, s) for s in inp]
    assert out[0][0] is True and approx(out[0][1], 0.0)
    assert out[1][0] is False and approx(out[1][1], 0.0)
    L.so_equalizer_enable(e, 1)
    out = [est(e, inp[i % 2]) for i in range(32)]
    assert abs(out[-1][1]) < 1e-5
    L.so_equalizer_free(e)


def test_nlms_evolve():
    proakis_b = np.array([0.407, 0.815, 0.407], dtype=np.float32)
    inp = [0.0, 1.0, 0.0, -1.0]
    ch = np.zeros(3, dtype=np.float32)   # oldest..newest
    inv_w = np.zeros(3, dtype=np.float32)
    inv_c = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    err = 0.0
    for k in range(128):
        s = inp[k % 4]
        ch = np.append(ch[1:], np.float32(s))
        ch_sample = L.so_mac_ff(ch.ctypes.data_as(F32P), 3, proakis_b.ctypes.data_as(F32P), 3)
        inv_w = np.append(inv_w[1:], np.float32(ch_sample))
        estv = L.so_mac_ff(inv_w.ctypes.data_as(F32P), 3, inv_c.ctypes.data_as(F32P), 3)
        err = s - estv
        L.so_nlms_update(0.10, 1.0e-6, err, inv_w.ctypes.data_as(F32P), 3, inv_c.ctypes.data_as(F32P))
    assert abs(err) < 1e-2


def test_estimate_symbol_channel():
    ch_c = np.array([0.8, -0.2], dtype=np.float32)
    inp = [(0.0, 1.0), (0.0, -1.0)]
    ch = np.zeros(2, dtype=np.float32)
    e = L.so_equalizer_new(8, 4, 0.2, 1.0e-5, 0, 0)

    def through(samples):
        nonlocal ch
        out = []
        for v in samples:
            ch = np.append(ch[1:], np.float32(v))
            out.append(L.so_mac_ff(ch.ctypes.data_as(F32P), 2, ch_c.ctypes.data_as(F32P), 2))
        return out

    last = (False, 0.0)
    for k in range(32):
        last = est(e, through(inp[k % 2]))
    assert abs(last[1]) < 1e-4
    for s in inp:
        last = est(e, through(s))
        assert last[0] == (s[1] >= 0) and abs(last[1]) < 1e-4
    L.so_equalizer_free(e)


def test_estimate_symbol_training():
    e = L.so_equalizer_new(8, 4, 0.2, 1.0e-5, 1, SYNC)
    assert L.so_equalizer_train(e) == 0
    w, c = C.c_uint32(), C.c_uint32()
    assert L.so_equalizer_mode(e, C.byref(w), C.byref(c)) == 2 and w.value == SYNC and c.value == 0
    chansig = bytes_to_samples(bytes([0x54, 0x54]), 2)
    for i in range(0, len(chansig), 2):
        est(e, chansig[i:i + 2])
    assert L.so_equalizer_mode(e, C.byref(w), C.byref(c)) == 2 and c.value == 16
    for i in range(0, len(chansig), 2):
        est(e, chansig[i:i + 2])
    assert L.so_equalizer_mode(e, None, None) == 1
    # trained on inverted data: the DFE flips bits (equalize.rs:558-563)
    assert est(e, (0.0, -1.0))[0] is True
    L.so_equalizer_reset(e)
    assert L.so_equalizer_train(e) == 0
    chansig = bytes_to_samples(bytes([0xAB] * 4), 2)
    for i in range(0, len(chansig), 2):
        est(e, chansig[i:i + 2])
    assert L.so_equalizer_mode(e, None, None) == 1
    assert est(e, (0.0, -1.0))[0] is False
    L.so_equalizer_free(e)
    e = L.so_equalizer_new(8, 4, 0.2, 1.0e-5, 0, 0)
    assert L.so_equalizer_train(e) == -1  # NoTrainingSequenceErr
    L.so_equalizer_free(e)


def test_equalizer_input():
    chansig = bytes_to_samples(bytes([0xAB, 0xBA]), 2)
    e = L.so_equalizer_new(8, 4, 0.2, 1.0e-5, 0, 0)
    err = f32()
    out = [L.so_equalizer_input(e, chansig[i:i + 16].ctypes.data_as(F32P), C.byref(err)) for i in (0, 16)]
    assert out == [0xAB, 0xBA]
    L.so_equalizer_free(e)


# ---------------------------------------------------------------- framing.rs:259-349
def framer_input(f, b, restart=False):
    p = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    st = L.so_framer_input(f, b, 0, int(restart), C.byref(p), C.byref(n))
    return st, (bytes(p[: n.value]) if st == ob.LINK_BURST else None)


def framer_end(f):
    p = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    st = L.so_framer_end(f, C.byref(p), C.byref(n))
    return st, (bytes(p[: n.value]) if st == ob.LINK_BURST else None)


def test_message_prefix_errors():
    be = lambda s: int.from_bytes(s, "big")
    assert L.so_message_prefix_errors(be(b"ZCZC")) == 0
    assert L.so_message_prefix_errors(be(b"NNNN")) == 0
    assert L.so_message_prefix_errors(be(bytes([171] * 4))) == 18
    assert L.so_message_prefix_errors(be(b"ZCZE")) == 2


def test_framer_prefix():
    f = L.so_framer_new(1, 10)
    gave_up = False
    for i in range(32):
        st, _ = framer_input(f, 0xAB, i == 0)
        if st == ob.LINK_NO_CARRIER:
            assert i >= 21
            gave_up = True
        else:
            assert st == ob.LINK_SEARCHING
    assert gave_up
    framer_input(f, 0xAB, True)
    framer_input(f, 0xAB, True)
    last = None
    for d in b"ZCZC":
        last, _ = framer_input(f, d)
        assert last in (ob.LINK_SEARCHING, ob.LINK_READING)
    assert last == ob.LINK_READING
    assert framer_end(f) == (ob.LINK_BURST, b"ZCZC")
    assert framer_end(f)[0] == ob.LINK_NO_CARRIER
    L.so_framer_free(f)


def test_framer_burst_process():
    msg = b"gArbAZgEZCZC-ORG-EEE-012345-567890+0000-0001122-NOCALL00-GARBAGE"
    permit = 10
    f = L.so_framer_new(2, permit)
    framer_input(f, 0xAB, True)
    for c in msg:
        st, _ = framer_input(f, c)
        assert st in (ob.LINK_SEARCHING, ob.LINK_READING)
    found = False
    for j in range(permit + 1):
        st, b = framer_input(f, 0xAB)
        if j >= permit:
            assert st == ob.LINK_BURST
            assert b.startswith(b"ZCZC-ORG-EEE-012345-567890+0000-0001122-NOCALL00-")
            found = True
        else:
            assert st in (ob.LINK_SEARCHING, ob.LINK_READING)
    assert found
    L.so_framer_free(f)


def test_is_allowed_byte():
    allowed = set(b"-+?()[]._,/ ") | set(range(ord("0"), ord("9") + 1)) | \
        set(range(ord("A"), ord("Z") + 1)) | set(range(ord("a"), ord("z") + 1))
    for c in range(256):
        assert bool(L.so_is_allowed_byte(c)) == (c in allowed)


# ---------------------------------------------------------------- builder.rs:95-279
def test_builder_clamping():
    c = ob.default_config(22050)
    L.so_config_with_timing_bandwidth(C.byref(c), 2.0, 3.0)
    assert c.timing_bw_unlocked == 1.0 and c.timing_bw_locked == 1.0
    L.so_config_with_timing_bandwidth(C.byref(c), 0.1, 0.5)
    assert c.timing_bw_locked == np.float32(0.1)
    L.so_config_with_squelch_power(C.byref(c), 2.0, 1.5)
    assert c.squelch_power_open == 1.0 and c.squelch_power_close == 1.5  # min(close, open) with raw open
    L.so_config_with_frame_prefix_max_errors(C.byref(c), 12)
    assert c.frame_prefix_max_errors == 7
    L.so_config_with_timing_max_deviation(C.byref(c), 0.9)
    assert c.timing_max_deviation == 0.5
    L.so_config_with_dc_blocker_length(C.byref(c), -1.0)
    assert c.dc_blocker_len == 0.0
    h = C.c_void_p()
    assert L.so_rx_new(C.byref(c), C.byref(h)) == -1  # reference panics: DCBlocker::new(0)
    c = ob.default_config(22050)
    L.so_config_with_adaptive_equalizer(C.byref(c), 0, 9, 2.0, -1.0)
    assert (c.eq_nff, c.eq_nfb, c.eq_relaxation, c.eq_regularization) == (1, 1, 1.0, 0.0)
    c = ob.default_config(22050)
    L.so_config_with_agc_gain_limits(C.byref(c), 2.0, 1.0)
    assert L.so_rx_new(C.byref(c), C.byref(h)) == -2  # f32::clamp(min > max) panics
