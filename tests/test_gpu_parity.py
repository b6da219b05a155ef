"""GPU parity: the gfx950 kernels, called through the C ABI, against the oracle.

Bar (DESIGN.md "Parity"): bit-exact.  Events (kind, input sample counter, burst bytes),
transport messages and the soft-symbol stream (tolerance 0.0: identical f32 bit patterns)
must equal the oracle's on the same inputs.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, TEST_MESSAGE
from helpers.oracle_compare import assert_every_channel_matches_oracle

pytestmark = pytest.mark.gpu

SOFT_SYMBOL_TOLERANCE = 0.0   # strict mode: same op order, same roundings => identical bits


@pytest.fixture(scope="module")
def sa():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sameold_amd import build as sbuild
    sbuild.build()
    import sameold_amd
    sameold_amd.load_library()
    return sameold_amd


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.lib()
    return binding


def load_pcm(name):
    return np.fromfile(os.path.join(GOLDEN, f"{name}.22050.s16le.bin"), dtype="<i2")


def events_by_channel(rx):
    out = {}
    for e in rx.poll_events():
        out.setdefault(e.channel, []).append(e.as_tuple())
    return out


def oracle_events(ob, cfg, x, link_only=False):
    return [e.as_tuple() for e in ob.Receiver(cfg, link_only=link_only).run(np.ascontiguousarray(x))]



# ------------------------------------------------------------------ golden recordings
@pytest.mark.parametrize("name", ["npt", "two_and_two", "long_message"])
def test_sample_recordings_match_oracle_and_text(sa, ob, name):
    pcm = load_pcm(name)
    with open(os.path.join(GOLDEN, "link_events.json")) as f:
        fix = json.load(f)[name]
    rx = sa.SameReceiverBuilder(22050).samedec().build_batch(1)
    rx.process_host(pcm.astype(np.float32))
    got = [[e.kind, e.sample_counter, e.symbol_count, e.data().hex()] for e in rx.poll_events()]
    assert got == fix["events"]
    # text level, with the EOF flush (crates/samedec/src/app.rs:118)
    lines = [bytes.fromhex(g[3]).decode() if g[0] == sa.TRANSPORT_MSG_START else "NNNN"
             for g in got if g[0] in (sa.TRANSPORT_MSG_START, sa.TRANSPORT_MSG_END)]
    rx.flush()
    lines += [e.message() for e in rx.poll_events() if e.message() is not None][:1]
    exp = [l for l in open(os.path.join(GOLDEN, f"{name}.22050.s16le.txt")).read().splitlines() if l != "+OK"]
    assert lines == exp


def test_i16_input_path(sa, ob):
    pcm = load_pcm("npt")
    rx = sa.SameReceiverBuilder(22050).samedec().build_batch(1)
    rx.process_host(pcm)            # int16 on the device, cast in the kernel
    got = [e.as_tuple() for e in rx.poll_events()]
    assert got == oracle_events(ob, ob.samedec_config(), pcm)


def test_single_receiver_iterator_semantics(sa, ob):
    """same_rx_process mirrors iter_events(): consumed counts and event order."""
    pcm = load_pcm("npt").astype(np.float32)
    rx = sa.SameReceiverBuilder(22050).samedec().build()
    ref = ob.Receiver(ob.samedec_config())
    it_ref = ref.iter_events(pcm)
    n = 0
    for ev in rx.iter_events(pcm):
        er = next(it_ref)
        assert ev.as_tuple() == er.as_tuple()
        assert rx.consumed == ref.consumed
        n += 1
    assert n > 10 and next(it_ref, None) is None
    assert rx.input_sample_counter() == len(pcm)
    msgs = list(sa.SameReceiverBuilder(22050).samedec().build().iter_messages(pcm))
    assert msgs == ["ZCZC-PEP-NPT-000000+0030-2771820-TEST    -"]


def test_flush_emits_long_message(sa):
    rx = sa.SameReceiverBuilder(22050).samedec().build()
    assert list(rx.iter_messages(load_pcm("long_message").astype(np.float32))) == []
    assert rx.flush() == TEST_MESSAGE
    assert rx.flush() is None


# ------------------------------------------------------------------ in-process synthetic (receiver.rs:642-705)
def make_test_burst(ob, msg, n):
    burst = ob.modulate_afsk(bytes([0xAB] * 16) + msg, 22050) * np.float32(16384.0)
    parts = [burst]
    for _ in range(1, n):
        parts += [np.zeros(22050, np.float32), burst]
    parts.append(np.zeros(2 * 22050, np.float32))
    return np.concatenate(parts)


def test_iter_events_reference_test(sa, ob):
    afsk = make_test_burst(ob, TEST_MESSAGE.encode(), 1)
    rx = sa.SameReceiverBuilder(22050).with_timing_max_deviation(0.01).build()
    evs = list(rx.iter_events(afsk))
    assert [e.kind for e in evs] == [sa.LINK_SEARCHING, sa.LINK_READING, sa.LINK_BURST,
                                     sa.TRANSPORT_ASSEMBLING, sa.LINK_NO_CARRIER]
    assert evs[2].data().startswith(TEST_MESSAGE.encode())


def test_top_level_receiver_reference_test(sa, ob):
    afsk = make_test_burst(ob, TEST_MESSAGE.encode(), 3)
    rx = sa.SameReceiverBuilder(22050).with_timing_max_deviation(0.01).build()
    assert next(rx.iter_messages(afsk)) == TEST_MESSAGE


# ------------------------------------------------------------------ many channels, chunking, layouts
def mixed_batch(sa, n_ch, n_samples, seed, rate=22050, noise=0.0, integer_symbols=False):
    x = sa.synth_afsk(n_ch, n_samples, rate, seed=seed, noise_sigma=noise,
                      integer_symbols=integer_symbols).cpu().numpy()
    return x


@pytest.mark.parametrize("n_ch,seconds,noise", [(64, 6.0, 0.0), (200, 4.0, 0.05), (130, 5.0, 0.4), (96, 12.0, 0.0)])
def test_synthetic_batch_bit_exact(sa, ob, n_ch, seconds, noise):
    n = int(22050 * seconds)
    x = mixed_batch(sa, n_ch, n, seed=11 + n_ch, noise=noise)
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    import torch
    rx.process_tensor(torch.from_numpy(x).cuda())
    rx.sync()
    got = events_by_channel(rx)
    cfg = ob.default_config(22050)
    n_bursts = 0
    for c in range(n_ch):
        ref = oracle_events(ob, cfg, x[:, c])
        assert got.get(c, []) == ref, f"channel {c}"
        n_bursts += sum(1 for t in ref if t[0] == sa.LINK_BURST)
    if noise < 0.1:
        assert n_bursts >= n_ch  # the workload really carries decodable bursts
        # clean channels decode exactly what was transmitted
        for c in range(min(n_ch, 16)):
            sent = sa.synth_payload(11 + n_ch, c)
            bursts = [t[2] for t in got[c] if t[0] == sa.LINK_BURST and t[2].startswith(b"ZCZC")]
            assert bursts and all(b.startswith(sent) for b in bursts)


def test_chunked_processing_equals_one_shot(sa, ob):
    n_ch, n = 96, 22050 * 4
    x = mixed_batch(sa, n_ch, n, seed=5)
    import torch
    xd = torch.from_numpy(x).cuda()
    one = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    one.process_tensor(xd); one.sync()
    a = events_by_channel(one)
    chunked = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    rng = np.random.default_rng(3)
    off = 0
    while off < n:
        k = int(rng.integers(1, 9000))
        chunked.process_tensor(xd[off:off + k].contiguous())
        off += k
    chunked.sync()
    assert events_by_channel(chunked) == a
    assert chunked.input_sample_counter() == n


@pytest.mark.parametrize("rate", [22050, 48000])
def test_tiny_chunks_through_the_pipeline(sa, ob, rate):
    """Calls of one, two and three blocks (and fractions of a block) exercise the pipeline's fill and
    drain: 64 channels at a standard rate run demod_pipe_kernel for every whole block (20 samples at
    22.05 kHz, 32 at 48 kHz, where stage 1's replay history also lives in a three-block LDS ring)."""
    n_ch, n = 64, rate * 2
    blk = 20 if rate == 22050 else 32
    x = mixed_batch(sa, n_ch, n, seed=21, rate=rate)
    import torch
    xd = torch.from_numpy(x).cuda()
    one = sa.SameReceiverBuilder(rate).build_batch(n_ch)
    assert one.kernel_name() == "demod_pipe_kernel"
    one.process_tensor(xd); one.sync()
    a = events_by_channel(one)
    chunked = sa.SameReceiverBuilder(rate).build_batch(n_ch)
    sizes = [1, blk - 1, blk, blk + 1, 2 * blk - 1, 2 * blk, 2 * blk + 1, 3 * blk, 5, 4 * blk, 5 * blk, 2000]
    off = i = 0
    while off < n:
        k = sizes[i % len(sizes)] if off < 12000 * blk // 18 else 7000
        chunked.process_tensor(xd[off:off + k].contiguous())
        off += k; i += 1
    chunked.sync()
    assert events_by_channel(chunked) == a
    for c in (0, 31, 63):
        assert a.get(c, []) == oracle_events(ob, ob.default_config(rate), x[:, c])


def test_empty_input_is_a_no_op(sa):
    """process() of no samples (the reference's iterator simply ends): no launch, no events, no state change."""
    import torch
    n_ch = 64
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    rx.process_host(np.zeros((0, n_ch), dtype=np.float32))
    rx.process_tensor(torch.zeros((0, n_ch), dtype=torch.float32, device="cuda"))
    rx.sync()
    assert rx.input_sample_counter() == 0 and len(rx.poll_events_np()) == 0
    x = mixed_batch(sa, n_ch, 22050, seed=3)
    rx.process_host(x); rx.process_host(x[:0]); rx.sync()
    ref = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    ref.process_host(x); ref.sync()
    assert rx.input_sample_counter() == 22050 and events_by_channel(rx) == events_by_channel(ref)


def test_channel_major_layout(sa, ob):
    n_ch, n = 70, 22050 * 3
    x = mixed_batch(sa, n_ch, n, seed=9)
    a = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    a.process_host(x)
    b = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    b.process_host(np.ascontiguousarray(x.T), layout=sa.LAYOUT_CHANNEL_MAJOR)
    assert events_by_channel(a) == events_by_channel(b)


def test_soft_symbols_bit_exact(sa, ob):
    pcm = load_pcm("npt").astype(np.float32)
    rx = sa.SameReceiverBuilder(22050).samedec().build_batch(1, trace_symbols=True)
    rx.process_host(pcm)
    tr = rx.read_trace(0)
    ref = ob.Receiver(ob.samedec_config(), link_only=True)
    ref.enable_trace(4096)
    ref.run(pcm)
    rt = ref.trace()
    n = min(len(tr), len(rt))
    assert n >= 3000
    assert np.array_equal(tr["sample_counter"][:n], rt["sample_counter"][:n])
    for f in ("zero", "sym", "err", "next"):
        d = np.abs(tr[f][:n].astype(np.float64) - rt[f][:n].astype(np.float64))
        assert d.max() <= SOFT_SYMBOL_TOLERANCE, f
    gold = np.load(os.path.join(GOLDEN, "soft_symbols_npt.npz"))["trace"]
    assert np.array_equal(tr["sym"][:len(gold)].view(np.uint32), gold["sym"].view(np.uint32))


# ------------------------------------------------------------------ other rates / configs
@pytest.mark.parametrize("rate", [48000, 44100, 11025, 8000])
def test_other_sample_rates(sa, ob, rate):
    n_ch, n = 64, rate * 4
    x = mixed_batch(sa, n_ch, n, seed=rate, rate=rate)
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch)
    rx.process_host(x)
    got = events_by_channel(rx)
    cfg = ob.default_config(rate)
    for c in range(n_ch):
        assert got.get(c, []) == oracle_events(ob, cfg, x[:, c]), f"rate {rate} channel {c}"


@pytest.mark.parametrize("variant", ["no_eq", "eq_8_3", "wide_timing", "dc1", "tight_squelch"])
def test_builder_variants(sa, ob, variant):
    n_ch, n = 64, 22050 * 4
    x = mixed_batch(sa, n_ch, n, seed=77, noise=0.1)
    b = sa.SameReceiverBuilder(22050)
    cfg = ob.default_config(22050)
    L = ob.lib()
    if variant == "no_eq":
        b.without_adaptive_equalizer(); L.so_config_without_adaptive_equalizer(C.byref(cfg))
    elif variant == "eq_8_3":
        b.with_adaptive_equalizer(8, 3, 0.2, 1e-5); L.so_config_with_adaptive_equalizer(C.byref(cfg), 8, 3, 0.2, 1e-5)
    elif variant == "wide_timing":
        b.with_timing_max_deviation(0.3).with_timing_bandwidth(0.5, 0.2)
        L.so_config_with_timing_max_deviation(C.byref(cfg), 0.3); L.so_config_with_timing_bandwidth(C.byref(cfg), 0.5, 0.2)
    elif variant == "dc1":
        b.with_dc_blocker_length(0.03); L.so_config_with_dc_blocker_length(C.byref(cfg), 0.03)
    elif variant == "tight_squelch":
        b.with_squelch_power(0.5, 0.3).with_preamble_max_errors(0).with_frame_max_invalid(0)
        L.so_config_with_squelch_power(C.byref(cfg), 0.5, 0.3); L.so_config_with_preamble_max_errors(C.byref(cfg), 0)
        L.so_config_with_frame_max_invalid(C.byref(cfg), 0)
    rx = b.build_batch(n_ch)
    rx.process_host(x)
    got = events_by_channel(rx)
    for c in range(n_ch):
        assert got.get(c, []) == oracle_events(ob, cfg, x[:, c]), f"{variant} channel {c}"


@pytest.mark.parametrize("seed", range(10))
def test_random_builder_configurations(sa, ob, seed):
    """Every builder setter at once, drawn at random (including rates no specialised kernel exists
    for): whatever kernel the dispatcher picks, every checked channel equals the oracle given the
    same calls."""
    rng = np.random.default_rng(1000 + seed)
    rate = int(rng.choice([22050, 22050, 48000, 44100, 16000, 32000, 11025]))
    b = sa.SameReceiverBuilder(rate)
    cfg = ob.default_config(rate)
    L = ob.lib()
    f32 = lambda v: float(np.float32(v))

    def both(name, *args):
        getattr(b, name)(*args)
        getattr(L, "so_config_" + name)(C.byref(cfg), *args)

    if rng.random() < 0.5: both("with_dc_blocker_length", f32(rng.uniform(0.05, 1.2)))
    if rng.random() < 0.5: both("with_agc_bandwidth", f32(rng.uniform(0.002, 0.05)))
    if rng.random() < 0.5: both("with_agc_gain_limits", f32(10 ** rng.uniform(-6, -4)), f32(10 ** rng.uniform(-3, 0)))
    if rng.random() < 0.5:
        u = f32(rng.uniform(0.05, 0.4)); both("with_timing_bandwidth", u, f32(u * rng.uniform(0.2, 1.5)))
    if rng.random() < 0.5: both("with_timing_max_deviation", f32(rng.choice([0.0, 0.005, 0.01, 0.03, 0.1, 0.25])))
    if rng.random() < 0.5:
        o = f32(rng.uniform(0.0, 0.6)); both("with_squelch_power", o, f32(o * rng.uniform(0.3, 1.3)))
    if rng.random() < 0.5: both("with_squelch_bandwidth", f32(rng.uniform(0.02, 0.3)))
    if rng.random() < 0.5: both("with_preamble_max_errors", int(rng.integers(0, 6)))
    r = rng.random()
    if r < 0.25: both("without_adaptive_equalizer")
    elif r < 0.6: both("with_adaptive_equalizer", int(rng.integers(1, 12)), int(rng.integers(1, 8)),
                       f32(rng.uniform(0.01, 0.3)), f32(10 ** rng.uniform(-7, -4)))
    if rng.random() < 0.5: both("with_frame_prefix_max_errors", int(rng.integers(0, 5)))
    if rng.random() < 0.5: both("with_frame_max_invalid", int(rng.integers(0, 6)))
    n_ch, n = 64, rate * 3
    x = mixed_batch(sa, n_ch, n, seed=500 + seed, rate=rate, noise=float(rng.choice([0.0, 0.05, 0.2])))
    rx = b.build_batch(n_ch)
    for off in range(0, n, 25013):
        rx.process_host(x[off:off + 25013])
    got = events_by_channel(rx)
    for c in range(0, n_ch, 6):
        assert got.get(c, []) == oracle_events(ob, cfg, x[:, c]), f"seed {seed} rate {rate} kernel {rx.kernel_name()} channel {c}"


@pytest.mark.parametrize("rate,dev,kernel", [(22050, 0.05, "demod_fast_kernel"), (22050, 0.01, "demod_pipe_kernel"),
                                             (48000, 0.2, "demod_kernel<B=16>"), (48000, 0.01, "demod_pipe_kernel"),
                                             (44100, 0.15, "demod_kernel<B=16>"), (44100, 0.01, "demod_pipe_kernel")])
def test_block_length_follows_the_timing_bound(sa, ob, rate, dev, kernel):
    """A block may hold at most one TED instant (the 22.05 kHz pipeline: two).  The 18-sample (22.05 kHz mirrored), 20-sample (pipelined)
    and 32-sample (48 / 44.1 kHz) variants are only dispatched when timing_max_deviation leaves room
    for them; with a wider deviation the dispatcher falls back to shorter blocks, and every
    choice matches the oracle."""
    n_ch, n = 64, rate * 3
    x = mixed_batch(sa, n_ch, n, seed=5, rate=rate, noise=0.05)
    b = sa.SameReceiverBuilder(rate).with_timing_max_deviation(dev)
    cfg = ob.default_config(rate)
    ob.lib().so_config_with_timing_max_deviation(C.byref(cfg), dev)
    rx = b.build_batch(n_ch)
    assert rx.kernel_name() == kernel
    rx.process_host(x)
    got = events_by_channel(rx)
    for c in range(0, n_ch, 5):
        assert got.get(c, []) == oracle_events(ob, cfg, x[:, c]), f"channel {c}"


def test_reset(sa, ob):
    pcm = load_pcm("npt").astype(np.float32)
    rx = sa.SameReceiverBuilder(22050).samedec().build_batch(1)
    rx.process_host(pcm); rx.poll_events()
    rx.reset()
    assert rx.input_sample_counter() == 0
    rx.process_host(pcm)
    got = [e.as_tuple() for e in rx.poll_events()]
    ref = ob.Receiver(ob.samedec_config()); ref.run(pcm); ref.reset()
    assert got == [e.as_tuple() for e in ref.run(pcm)]


def test_build_errors(sa):
    with pytest.raises(sa.SameError) as e:
        sa.SameReceiverBuilder(22050).with_dc_blocker_length(0.0).build_batch(4)
    assert e.value.code == -2
    with pytest.raises(sa.SameError) as e:
        sa.SameReceiverBuilder(22050).with_agc_gain_limits(2.0, 1.0).build_batch(4)
    assert e.value.code == -3
    with pytest.raises(sa.SameError) as e:
        sa.SameReceiverBuilder(22050).with_adaptive_equalizer(40, 4).build_batch(4)
    assert e.value.code == -4


def test_hypot_matches_glibc(sa, ob):
    """The device's (float)sqrt((double)re*re+(double)im*im) must equal glibc hypotf; exercised
    end-to-end by every parity test above, and here on a noise-only batch whose matched-filter
    outputs sweep many magnitudes."""
    n_ch, n = 64, 22050
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((n, n_ch)) * rng.uniform(1, 20000, n_ch)).astype(np.float32)
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch, link_only=True, trace_symbols=True)
    rx.process_host(x)
    for c in (0, 17, 63):
        ref = ob.Receiver(ob.default_config(22050), link_only=True)
        ref.enable_trace(4096); ref.run(np.ascontiguousarray(x[:, c]))
        rt, tr = ref.trace(), rx.read_trace(c)
        assert len(tr) == len(rt) > 400
        assert np.array_equal(tr["sym"].view(np.uint32), rt["sym"].view(np.uint32))
        assert np.array_equal(tr["sample_counter"], rt["sample_counter"])


# ------------------------------------------------------------------ fast vs generic kernel
PIPE_ENV = {"fast": "0", "pipe": "1"}      # SAME_PIPE: wavefront pipeline off / on (read when the batch is created)


@pytest.mark.parametrize("rate,variant", [(22050, "pipe"), (22050, "fast"), (48000, "pipe"), (48000, "fast"),
                                          (44100, "pipe"), (44100, "fast")])
def test_fast_kernel_equals_generic_kernel(sa, ob, rate, variant, monkeypatch):
    """Standard rates dispatch to the latency-optimised kernels (one wavefront per 64 channels,
    or the wavefront pipeline for small and medium batches); each must reproduce the
    any-configuration kernel (and therefore the oracle) bit for bit, including when chunk sizes
    are not whole blocks (remainder handled by the generic kernel)."""
    import torch
    monkeypatch.setenv("SAME_PIPE", PIPE_ENV[variant])    # read when the batch is created
    n_ch, n = 128, rate * 3 + 7
    x = mixed_batch(sa, n_ch, n, seed=rate + 1, rate=rate, noise=0.05)
    xd = torch.from_numpy(x).cuda()
    fast = sa.SameReceiverBuilder(rate).build_batch(n_ch, trace_symbols=True)
    assert fast.kernel_name() == f"demod_{variant}_kernel"
    gen = sa.SameReceiverBuilder(rate).build_batch(n_ch, generic_kernel=True, trace_symbols=True)
    assert gen.kernel_name().startswith("demod_kernel")
    rng = np.random.default_rng(rate)
    off = 0
    while off < n:
        k = int(rng.integers(1, 20000))
        fast.process_tensor(xd[off:off + k].contiguous())
        gen.process_tensor(xd[off:off + k].contiguous())
        off += k
    fast.sync(); gen.sync()
    a, b = events_by_channel(fast), events_by_channel(gen)
    assert a == b
    for c in (0, 5, 127):
        ta, tb = fast.read_trace(c), gen.read_trace(c)
        assert np.array_equal(ta["sample_counter"], tb["sample_counter"])
        assert np.array_equal(ta["sym"].view(np.uint32), tb["sym"].view(np.uint32))
    cfg = ob.default_config(rate)
    for c in range(0, n_ch, 9):
        assert a.get(c, []) == oracle_events(ob, cfg, x[:, c]), f"channel {c}"


@pytest.mark.parametrize("rate,n_ch,seconds", [(22050, 256, 12.0), (22050, 16448, 1.5), (48000, 256, 8.0),
                                               (44100, 192, 8.0)])
def test_pipeline_kernel_equals_single_wavefront_kernel(sa, monkeypatch, rate, n_ch, seconds):
    """The same batch through both variants of a rate, many bursts per channel (every AGC lock
    flip makes the pipeline's earlier stages replay a lane): identical events.  Above 16 384
    channels the 22.05 kHz pipeline kernel is the build with the halved register budget."""
    n = int(rate * seconds)
    x = sa.synth_afsk(n_ch, n, rate, seed=4242, noise_sigma=0.02)
    out = {}
    for variant in ("pipe", "fast"):
        monkeypatch.setenv("SAME_PIPE", PIPE_ENV[variant])
        rx = sa.SameReceiverBuilder(rate).build_batch(n_ch)
        assert rx.kernel_name() == f"demod_{variant}_kernel"
        for off in range(0, n, 50000):
            rx.process_tensor(x[off:off + 50000].contiguous())
        rx.sync()
        out[variant] = events_by_channel(rx)
    assert sum(len(v) for v in out["fast"].values()) > 2 * n_ch
    assert out["pipe"] == out["fast"]


@pytest.mark.parametrize("rate", [22050, 48000])
def test_pipeline_workgroup_widths(sa, ob, monkeypatch, rate):
    """Small batches spread over workgroups of 16 or 32 channels instead of 64 (more CUs, fewer
    divergent paths per wavefront).  Which channels share a wavefront must not matter."""
    n_ch, n = 192, rate * 5
    x = mixed_batch(sa, n_ch, n, seed=31 + rate, rate=rate, noise=0.05)
    out = {}
    for lanes in ("16", "32", "64"):
        monkeypatch.setenv("SAME_PIPE_LANES", lanes)         # read when the batch is created
        rx = sa.SameReceiverBuilder(rate).build_batch(n_ch)
        assert rx.kernel_name() == "demod_pipe_kernel"
        for off in range(0, n, 30001):
            rx.process_host(x[off:off + 30001])
        out[lanes] = events_by_channel(rx)
    assert out["16"] == out["64"] and out["32"] == out["64"]
    # 22.05 kHz: the narrow workgroups also split stage 2 over two wavefronts (mark / space filter)
    monkeypatch.setenv("SAME_PIPE_LANES", "16"); monkeypatch.setenv("SAME_PIPE_SPLIT", "0")
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch)
    rx.process_host(x)
    assert events_by_channel(rx) == out["64"]
    cfg = ob.default_config(rate)
    for c in range(0, n_ch, 23):
        assert out["16"].get(c, []) == oracle_events(ob, cfg, x[:, c]), f"channel {c}"


@pytest.mark.parametrize("n_ch,kernel", [(16, "demod_pipe_kernel"), (48, "demod_pipe_kernel"), (80, "demod_pipe_kernel"),
                                         (70, "demod_fast_kernel"), (4128, "demod_pipe_kernel"), (4112, "demod_fast_kernel")])
def test_channel_counts_around_the_workgroup_widths(sa, n_ch, kernel):
    """Batches that are whole 16- or 32-channel workgroups run the pipeline, others one wavefront per 64
    channels with a ragged tail; either way every channel equals the any-configuration kernel."""
    n = 22050 * 2
    x = sa.synth_afsk(n_ch, n, 22050, seed=77, noise_sigma=0.02)
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    assert rx.kernel_name() == kernel
    gen = sa.SameReceiverBuilder(22050).build_batch(n_ch, generic_kernel=True)
    rx.process_tensor(x); gen.process_tensor(x)
    rx.sync(); gen.sync()
    a, b = events_by_channel(rx), events_by_channel(gen)
    assert sum(len(v) for v in b.values()) >= n_ch and a == b


def test_dense_one_wavefront_build_equals_the_default(sa, ob, monkeypatch):
    """Batches of more than 1 024 wavefronts (beyond 65 536 channels) run the one-wavefront kernel in a build for two
    wavefronts per SIMD: half the registers, the squelch history in the HBM state array instead of LDS (fetched a
    block ahead).  Forced here on a small ragged batch fed in chunks: every event equals the default build's and
    the oracle's."""
    n_ch, n = 4112, 22050 * 3           # not a multiple of 16: the one-wavefront kernel, with a ragged last wavefront
    x = sa.synth_afsk(n_ch, n, 22050, seed=4112, noise_sigma=0.03)
    out = {}
    for dense in ("0", "1"):
        monkeypatch.setenv("SAME_FAST_DENSE", dense)
        rx = sa.SameReceiverBuilder(22050).build_batch(n_ch)
        assert rx.kernel_name() == "demod_fast_kernel"
        for off in range(0, n, 20011):
            rx.process_tensor(x[off:off + 20011].contiguous())
        rx.sync()
        out[dense] = events_by_channel(rx)
    assert sum(len(v) for v in out["0"].values()) > 2 * n_ch and out["1"] == out["0"]
    cfg = ob.default_config(22050)
    xh = x.cpu().numpy()
    for c in (0, 777, 4111):
        assert out["1"].get(c, []) == oracle_events(ob, cfg, xh[:, c]), f"channel {c}"


@pytest.mark.parametrize("amp", [300.0, 6000.0])
def test_two_instants_in_one_block(sa, ob, amp):
    """The 22.05 kHz pipeline's blocks are 20 samples, instants at least 19.45 apart: when the timing loop runs
    at its fastest a block holds two (spacing 19, the first one at the block's first sample).  Noise drives
    the loop to both ends of its range all the time: about five instant pairs 19 apart per channel-second,
    one in twenty of them inside one block -- some 80 in this batch -- and every instant, symbol and event
    still equals the one-wavefront kernel's (18-sample blocks, never two instants) and the oracle's."""
    import torch
    n_ch, n = 256, 22050 * 4
    rng = np.random.default_rng(int(amp))
    x = (rng.standard_normal((n, n_ch)) * amp).astype(np.float32)
    x[:, ::2] += mixed_batch(sa, n_ch, n, seed=91)[:, ::2]          # every other channel also carries bursts
    xd = torch.from_numpy(x).cuda()
    traced = list(range(1, n_ch, 8))
    out = {}
    for variant in ("pipe", "fast"):
        os.environ["SAME_PIPE"] = PIPE_ENV[variant]
        try:
            rx = sa.SameReceiverBuilder(22050).build_batch(n_ch, trace_symbols=True)
            assert rx.kernel_name() == ("demod_fast_kernel" if variant == "fast" else "demod_pipe_kernel")
            for off in range(0, n, 9973):
                rx.process_tensor(xd[off:off + 9973].contiguous())
            rx.sync()
            out[variant] = (events_by_channel(rx), [rx.read_trace(c, cap=8192) for c in traced])
        finally:
            os.environ.pop("SAME_PIPE", None)
    assert out["pipe"][0] == out["fast"][0]
    tight = 0
    for tp, tf in zip(out["pipe"][1], out["fast"][1]):
        assert len(tp) > 1500 and np.array_equal(tp["sample_counter"], tf["sample_counter"])
        for f in ("zero", "sym", "err", "next"):
            assert np.array_equal(tp[f].view(np.uint32), tf[f].view(np.uint32)), f
        tight += int(np.sum(np.diff(tp["sample_counter"].astype(np.int64)) <= 39))    # an instant pair 19 apart
    assert tight >= 60, "the input does not make the loop run fast often enough to test anything"
    cfg = ob.default_config(22050)
    for c in (1, 2, 40):
        assert out["pipe"][0].get(c, []) == oracle_events(ob, cfg, x[:, c]), f"channel {c}"


def test_pack_bursts_matches_the_python_packer(sa):
    """same_batch_pack_bursts (what bench.py gathers across ranks) == distributed.pack_burst_events on the
    same queue, single- and multi-threaded (the library splits queues of 32 768 events and more)."""
    from sameold_amd import distributed as sd
    for n_ch, secs in ((64, 4), (4096, 6)):
        x = sa.synth_afsk(n_ch, 22050 * secs, 22050, seed=5)
        rx = sa.SameReceiverBuilder(22050).build_batch(n_ch)
        rx.process_tensor(x); rx.sync()
        ev = rx.peek_events_np()
        want = sd.pack_burst_events(ev, first_channel=1000)
        got = rx.pack_bursts_np(first_channel=1000)
        assert len(ev) > (32768 if n_ch > 64 else 0) and got.shape == want.shape and len(got) > n_ch
        assert np.array_equal(got, want)
        assert len(rx.peek_events_np()) == len(ev)          # the queue is untouched
        # ... and into a buffer the caller keeps: a view of it, the same records; too small a buffer is refused
        buf = np.full((len(want) + 7, 304), 0xAA, dtype=np.uint8)
        view = rx.pack_bursts_np(first_channel=1000, out=buf)
        assert view.base is buf and np.array_equal(view, want) and (buf[len(want):] == 0xAA).all()
        with pytest.raises(ValueError):
            rx.pack_bursts_np(out=buf[:len(want) - 1])
        with pytest.raises(ValueError):
            rx.pack_bursts_np(out=np.empty((len(want), 300), dtype=np.uint8))


def test_peek_and_drop_events_equal_poll(sa):
    """same_batch_peek_events / same_batch_drop_events: the queue viewed in place, then released."""
    n_ch, n = 64, 22050 * 4
    x = mixed_batch(sa, n_ch, n, seed=12)
    a = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    b = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    a.process_host(x); b.process_host(x)
    a.sync(); b.sync()
    want = a.poll_events_np()
    view = b.peek_events_np()
    assert len(view) == len(want) > 0 and not view.flags.writeable
    assert view.tobytes() == want.tobytes()
    half = len(view) // 2
    b.drop_events(half)
    rest = b.peek_events_np()
    assert rest.tobytes() == want[half:].tobytes()
    with pytest.raises(sa.SameError):
        b.drop_events(len(rest) + 1)
    b.drop_events(len(rest))
    # the header's promise: the view outlives same_batch_drop_events, also the drop that empties the queue (round 5's
    # build released the array right there: a consumer that peeks, drops everything and then reads read freed memory)
    assert b.pending_events() == 0
    assert rest.tobytes() == want[half:].tobytes()
    assert len(b.peek_events_np()) == 0 and len(b.poll_events_np()) == 0


@pytest.mark.parametrize("rate", [22050, 48000])
def test_i16_input_through_the_pipeline_kernel(sa, ob, rate):
    """int16 samples cast in stage 1 of the pipelined kernel (64 channels => demod_pipe_kernel)."""
    n_ch, n = 64, rate * 4
    x = np.clip(np.rint(mixed_batch(sa, n_ch, n, seed=77, rate=rate)), -32768, 32767).astype(np.int16)
    rx = sa.SameReceiverBuilder(rate).samedec().build_batch(n_ch)
    assert rx.kernel_name() == "demod_pipe_kernel"
    rx.process_host(x)
    got = events_by_channel(rx)
    cfg = ob.samedec_config(rate)
    for c in range(0, n_ch, 7):
        assert got.get(c, []) == oracle_events(ob, cfg, x[:, c]), f"channel {c}"


def test_negative_zero_agc_bound_uses_exact_clamp(sa, ob):
    """v_med3_f32 is only used when no AGC bound is -0.0; the other path must still match."""
    n_ch, n = 64, 22050 * 2
    x = mixed_batch(sa, n_ch, n, seed=3)
    rx = sa.SameReceiverBuilder(22050).with_agc_gain_limits(-0.0, 1.0e6).build_batch(n_ch)
    rx.process_host(x)
    got = events_by_channel(rx)
    cfg = ob.default_config(22050)
    ob.lib().so_config_with_agc_gain_limits(C.byref(cfg), -0.0, 1.0e6)
    for c in range(n_ch):
        assert got.get(c, []) == oracle_events(ob, cfg, x[:, c]), f"channel {c}"


def test_byte_clock_realignment_mid_preamble(sa, ob):
    """A bit slip inside the preamble makes the squelch re-align its byte clock while it is
    still unlocked ("adjust byte sync", rx/codesquelch.rs:255-264).  The kernels equalize
    symbols ahead of the byte clock and must roll that work back exactly."""
    def bits_of(data):
        return [(b >> i) & 1 for b in data for i in range(8)]

    def modulate(bits, fs=22050):
        # continuous-phase AFSK with the reference test modulator's symbol length (42 @ 22.05k)
        by = bytearray()
        acc = 0
        for i, bit in enumerate(bits):
            acc |= bit << (i % 8)
            if i % 8 == 7:
                by.append(acc); acc = 0
        if len(bits) % 8:
            by.append(acc)
        wav = ob.modulate_afsk(bytes(by), fs)
        return wav[: len(bits) * 42]

    hdr = b"ZCZC-WXR-TOR-039173-039051+0030-1591829-KCLE/NWS-"
    chans = []
    for slip in ([1, 0, 1], [1], [0, 0, 1, 1, 0], [1, 1, 0, 1, 0, 1, 1]):
        bits = bits_of(bytes([0xAB] * 9)) + slip + bits_of(bytes([0xAB] * 9) + hdr)
        w = modulate(bits) * np.float32(12000.0)
        chans.append(np.concatenate([np.zeros(3000, np.float32), w, np.zeros(22050, np.float32)]))
    n = max(len(c) for c in chans)
    x = np.zeros((n, 64), np.float32)
    for i in range(64):
        c = chans[i % len(chans)]
        x[(i * 37) % 2000:(i * 37) % 2000 + len(c), i] = c[: n - (i * 37) % 2000]
    cfg = ob.default_config(22050)
    ref = [oracle_events(ob, cfg, x[:, c]) for c in range(64)]
    decoded = sum(1 for r in ref for t in r if t[0] == sa.LINK_BURST and t[2].startswith(hdr))
    assert decoded >= 48          # the slip is recovered from: the header still decodes
    for generic in (False, True):
        rx = sa.SameReceiverBuilder(22050).build_batch(64, generic_kernel=generic)
        rx.process_host(x)
        got = events_by_channel(rx)
        for c in range(64):
            assert got.get(c, []) == ref[c], f"generic={generic} channel {c}"


def test_forced_end_of_message_timeout(sa, ob):
    """receiver.rs:300-309: 135 s after a StartOfMessage with no EOM, the receiver emits
    EndOfMessage at the first idle symbol past the deadline.  The deadline lives on the host
    and is armed on the device as a wake-up; timing must equal the oracle's."""
    afsk = make_test_burst(ob, b"ZCZC-WXR-TOR-039173+0030-1591829-KCLE/NWS-", 3)
    x = np.concatenate([afsk, np.zeros(22050 * 140, np.float32)])
    ref = [e.as_tuple() for e in ob.Receiver(ob.default_config(22050)).run(x)]
    assert [t[0] for t in ref].count(sa.TRANSPORT_MSG_END) == 1
    rx = sa.SameReceiverBuilder(22050).build_batch(1)
    # several calls, so the wake-up is armed between launches as in streaming use
    for off in range(0, len(x), 22050 * 20):
        rx.process_host(x[off:off + 22050 * 20])
    got = [e.as_tuple() for e in rx.poll_events()]
    assert got == ref


def test_pipelined_calls_keep_event_order(sa, ob):
    """Two launches in flight (harvest of k overlaps launch k+1): events still come out in
    per-channel time order and equal the one-shot result."""
    import torch
    n_ch, n = 128, 22050 * 6
    x = mixed_batch(sa, n_ch, n, seed=21)
    xd = torch.from_numpy(x).cuda()
    one = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    one.process_tensor(xd); one.sync()
    want = events_by_channel(one)
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    got = {}
    step = 22050
    for off in range(0, n, step):
        rx.process_tensor(xd[off:off + step].contiguous())      # no sync between calls
        for e in rx.poll_events():
            got.setdefault(e.channel, []).append(e.as_tuple())
    rx.sync()
    for e in rx.poll_events():
        got.setdefault(e.channel, []).append(e.as_tuple())
    assert got == want


# ------------------------------------------------------------------ AWGN Monte-Carlo (configs[4])
def test_awgn_trials_bit_exact_and_scored(sa, ob):
    """One noisy burst per channel over a 0..14 dB Eb/N0 grid (Philox + Box-Muller on the device):
    the kernel's events equal the oracle's on the identical noisy samples at every SNR, and
    the sweep's bookkeeping sees detection rise with SNR."""
    from sameold_amd import montecarlo as mc
    n, grid, rate, seed = 240, 15, 22050, 99
    T = 2 * rate - (2 * rate) % 16
    x = mc.synth_trials(n, 1000, T, rate, seed, 0.0, 1.0, grid)
    xh = x.cpu().numpy()
    assert np.isfinite(xh).all()
    # the generator is counter-based: a sub-batch reproduces the same trials
    x2 = mc.synth_trials(16, 1000 + 32, T, rate, seed, 0.0, 1.0, grid).cpu().numpy()
    assert np.array_equal(x2, xh[:, 32:48])
    rx = sa.SameReceiverBuilder(rate).build_batch(n, link_only=True)
    rx.process_tensor(x)
    rx.sync()
    ev = rx.poll_events_np()
    cfg = ob.default_config(rate)
    for c in range(n):
        mine = ev[ev["channel"] == c]
        got = [(int(r["kind"]), int(r["sample_counter"]), r["bytes"][: min(int(r["len"]), 288)].tobytes()) for r in mine]
        assert got == oracle_events(ob, cfg, xh[:, c], link_only=True), f"trial {c} ({c % grid} dB)"
    tally = mc.new_tally(grid)
    payloads = [sa.synth_payload(seed, 1000 + c) for c in range(n)]
    mc.score_bursts(ev, payloads, 1000, n, grid, tally)
    rows = mc.summarise(tally, 0.0, 1.0)
    assert sum(r["trials"] for r in rows) == n
    # 16 trials per grid point: only the trend is asserted (non-coherent FSK: BER ~ exp(-Eb/2N0)/2)
    hi = [r for r in rows if r["ebn0_db"] >= 12.0]
    lo = [r for r in rows if r["ebn0_db"] <= 4.0]
    assert min(r["burst_detection_rate"] for r in hi) >= 0.8
    assert max(r["burst_detection_rate"] for r in lo) <= 0.2 and all(r["intact_header_rate"] == 0.0 for r in lo)
    assert sum(r["intact_header_rate"] for r in hi) > 2.0


# ------------------------------------------------------------------ full size (BASELINE.json configs[1])
def test_full_size_round_trip_4096_channels_10s(sa, ob):
    """configs[1] at full size (4 096 channels x 220 500 samples, the bench workload): every channel's
    decoded header equals the header it was sent (modulate -> demodulate round trip), the link
    layer delivered three bursts of it, and EVERY channel's link events (kind, sample counter, burst
    bytes) equal the oracle's."""
    import torch
    n_ch, n, seed = 4096, 220500, 20260000
    x = sa.synth_afsk(n_ch, n, 22050, seed=seed)
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    assert rx.kernel_name() == "demod_pipe_kernel"
    rx.process_tensor(x)
    rx.sync()
    ev = rx.poll_events_np()
    starts = ev[ev["kind"] == sa.TRANSPORT_MSG_START]
    assert len(np.unique(starts["channel"])) == n_ch
    first = {}
    for r in starts:
        first.setdefault(int(r["channel"]), r["bytes"][: int(r["len"])].tobytes())
    bursts = ev[ev["kind"] == sa.LINK_BURST]
    n_bursts = np.bincount(bursts["channel"], minlength=n_ch)
    for c in range(n_ch):
        want = sa.synth_payload(seed, c)
        assert first[c] == want, f"channel {c}"
        assert n_bursts[c] >= 3
    assert_every_channel_matches_oracle(ob, ob.default_config(22050), x, ev)


@pytest.mark.parametrize("n_ch,rate,seconds", [(32768, 22050, 2.6), (16384, 48000, 2.6)])
def test_full_width_batches_deliver_what_was_sent(sa, ob, n_ch, rate, seconds):
    """The per-GPU shard of configs[3] (32 768 channels) and configs[2] (16 384 channels at 48 kHz)
    at full width: every burst a channel delivers begins with the header it was sent, nearly all
    channels deliver one within the first 2.6 s, and every channel matches the oracle event for event."""
    n, seed = int(rate * seconds), 31337
    x = sa.synth_afsk(n_ch, n, rate, seed=seed)
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True)
    rx.process_tensor(x)
    rx.sync()
    ev = rx.poll_events_np()
    bursts = ev[ev["kind"] == sa.LINK_BURST]
    assert len(np.unique(bursts["channel"])) > 0.95 * n_ch
    payload = {}
    for r in bursts:
        c = int(r["channel"])
        want = payload.setdefault(c, sa.synth_payload(seed, c))
        assert r["bytes"][: len(want)].tobytes() == want, f"channel {c}"
    assert_every_channel_matches_oracle(ob, ob.default_config(rate), x, ev)


def test_configs2_full_length_16384_channels_48k_10s(sa, ob):
    """BASELINE.json configs[2] as stated: 16 384 channels at 48 kHz, 10 s (480 000 samples per channel,
    31 GB resident), every channel against the oracle."""
    n_ch, rate, seed = 16384, 48000, 4242
    n = rate * 10
    x = sa.synth_afsk(n_ch, n, rate, seed=seed)
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True)
    assert rx.kernel_name() == "demod_pipe_kernel"
    rx.process_tensor(x)
    rx.sync()
    ev = rx.poll_events_np()
    bursts = ev[ev["kind"] == sa.LINK_BURST]
    assert len(np.unique(bursts["channel"])) == n_ch
    assert assert_every_channel_matches_oracle(ob, ob.default_config(rate), x, ev) > 20 * n_ch


# ------------------------------------------------------------------ stream contract, flush run-ahead, small polls
def test_process_tensor_is_ordered_after_the_producer_stream(sa, ob):
    """The library's own stream must wait for whatever produced the input on torch's stream (round-1
    advisor finding): the input here is the last link of a chain of asynchronous torch kernels on a
    side stream, handed over without a synchronise, and the temporaries are dropped at once."""
    import torch
    n_ch, n = 64, 22050 * 5
    base = sa.synth_afsk(n_ch, n, 22050, seed=77)
    want = base.cpu().numpy()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch, link_only=True)
    with torch.cuda.stream(side):
        junk = torch.empty((8192, 8192), device="cuda")
        for _ in range(6):
            junk = junk @ junk.T * 0.0                   # keeps the side stream busy for a while
        y = base.double()
        for _ in range(4):
            y = y * 1.0 + junk[0, 0].double() * 0.0
        xd = y.float().contiguous()                       # produced behind all of the above
        rx.process_tensor(xd)                            # no synchronise in between
        del xd, y, junk                                  # the allocator may want the memory back right away
        scratch = torch.full((n, n_ch), 1.0e9, device="cuda")   # ... and would scribble over it if it got it
    rx.sync()
    del scratch
    got = events_by_channel(rx)
    cfg = ob.default_config(22050)
    for c in range(n_ch):
        assert got.get(c, []) == oracle_events(ob, cfg, want[:, c], link_only=True), f"channel {c}"


def test_successive_launches_are_ordered_whatever_their_streams(sa, ob):
    """A launch continues the state the previous one leaves.  Calls of one stream handed alternately to the library's own
    stream and to two streams of the caller, back to back without a synchronise: the library orders them itself (an event
    wait on the previous launch when the stream changes); every channel's events must equal the oracle's over the whole
    stream."""
    import torch
    n_ch, part = 2048, 22050
    x = sa.synth_afsk(n_ch, 6 * part, 22050, seed=4242)
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch, link_only=True)
    for i, st in enumerate([None, a.cuda_stream, b.cuda_stream, None, b.cuda_stream, a.cuda_stream]):
        rx.process_tensor(x[i * part:(i + 1) * part].contiguous(), stream=st)
    rx.sync()
    ev = rx.poll_events_np()
    ev = ev[np.lexsort((np.arange(len(ev)), ev["channel"]))]
    assert assert_every_channel_matches_oracle(ob, ob.default_config(22050), x, ev) > 4 * n_ch


def test_successive_channel_major_launches_are_ordered_whatever_their_streams(sa, ob):
    """The same with CHANNEL-MAJOR calls of a strict batch, which go through the batch's own staging buffers (copy + transpose,
    slab by slab): a launch still in flight on another stream reads its input out of those buffers, so the next call's
    first staging write must wait for it (round-3 advisor finding)."""
    import torch
    n_ch, part = 1024, 22050
    x = sa.synth_afsk(n_ch, 6 * part, 22050, seed=4343)
    parts = [x[i * part:(i + 1) * part].t().contiguous() for i in range(6)]
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch, link_only=True)
    for i, st in enumerate([None, a.cuda_stream, b.cuda_stream, None, b.cuda_stream, a.cuda_stream]):
        rx.process_tensor(parts[i], layout=sa.LAYOUT_CHANNEL_MAJOR, stream=st)
    rx.sync()
    ev = rx.poll_events_np()
    ev = ev[np.lexsort((np.arange(len(ev)), ev["channel"]))]
    assert assert_every_channel_matches_oracle(ob, ob.default_config(22050), x, ev) > 4 * n_ch


def test_audio_after_an_early_flush_is_refused_until_reset(sa):
    """flush() returns at its first message while the device has run over all 4 s of zeros; real audio
    presented next must not be silently skipped (round-1 advisor finding)."""
    pcm = load_pcm("long_message").astype(np.float32)
    rx = sa.SameReceiverBuilder(22050).samedec().build()
    assert list(rx.iter_messages(pcm)) == []
    assert rx.flush() == TEST_MESSAGE                      # returned early: zeros left over on the device
    with pytest.raises(sa.receiver.SameError):
        next(rx.iter_events(load_pcm("npt").astype(np.float32)), None)
    assert rx.flush() is None                             # flushing on is fine (zeros again) and uses them up
    rx.reset()
    assert list(rx.iter_messages(load_pcm("npt").astype(np.float32))) == ["ZCZC-PEP-NPT-000000+0030-2771820-TEST    -"]


def test_polling_less_than_is_pending_still_delivers_everything_in_order(sa, ob):
    """A consumer that never drains the queue (cap < pending on every poll) while launches keep
    appending: nothing lost, nothing reordered (the polled prefix is reclaimed inside the handle)."""
    import torch
    n_ch, n = 64, 22050 * 8
    x = sa.synth_afsk(n_ch, n, 22050, seed=5150)
    ref = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    ref.process_tensor(x); ref.sync()
    want = [e.as_tuple() + (e.channel,) for e in ref.poll_events()]
    rx = sa.SameReceiverBuilder(22050).build_batch(n_ch)
    got = []
    step = 22050
    for off in range(0, n, step):
        rx.process_tensor(x[off:off + step].contiguous())
        got += [e.as_tuple() + (e.channel,) for e in rx.poll_events(max_events=7)]
    rx.sync()
    while True:
        more = rx.poll_events(max_events=5)
        if not more:
            break
        got += [e.as_tuple() + (e.channel,) for e in more]
    # the reference handle saw one call, this one eight: per channel the streams are identical
    def by_channel(evs):
        out = {}
        for t in evs:
            out.setdefault(t[3], []).append(t[:3])
        return out
    assert by_channel(got) == by_channel(want)


def test_instant_behind_an_agc_lock_in_the_locks_own_block(sa, ob):
    """Round 5's find (one trial in 65 536 of configs[4], seed 2026, trial 15 354; tools/strict_bytes_repro.py): the symbol
    that acquires sync -- and locks the AGC at its sample (receiver.rs:431) -- is the FIRST of two TED instants in a 20-sample
    block of the wavefront pipeline.  Stage 1 replays the AGC from the lock's sample on and stage 2 redoes its current block,
    but the second instant shares the lock's block: its soft sample, filtered over the window as it stood before the replay,
    becomes the next symbol's first sample, sits in the squelch's history for 24 symbols, reaches the equalizer -- and two
    marginal bits of the burst's noise tail came out differently from the reference's.  Stage 2 now filters that instant
    again over the corrected window.  Every soft sample of every symbol, bit for bit against the oracle's trace, and the
    bursts byte for byte."""
    from sameold_amd import montecarlo as mc
    rate, seed, grid, T = 22050, 2026, 15, 44096
    x = mc.synth_trials(16, 15344, T, rate, seed, 0.0, 1.0, grid)
    xh = x.cpu().numpy()
    cfg = ob.default_config(rate)
    for kw, name in (({}, "demod_pipe_kernel"), ({"generic_kernel": True}, None)):
        rx = sa.SameReceiverBuilder(rate).build_batch(16, link_only=True, trace_symbols=True, **kw)
        rx.process_tensor(x)
        rx.sync()
        if name:
            assert rx.kernel_name() == name
        ev = rx.poll_events_np()
        for c in range(16):
            mine = ev[ev["channel"] == c]
            got = [(int(r["kind"]), int(r["sample_counter"]), r["bytes"][: min(int(r["len"]), 288)].tobytes()) for r in mine]
            assert got == oracle_events(ob, cfg, xh[:, c], link_only=True), f"trial {15344 + c}"
            ref = ob.Receiver(cfg, link_only=True)
            ref.enable_trace(4096)
            ref.run(np.ascontiguousarray(xh[:, c]))
            rt, tr = ref.trace(), rx.read_trace(c)
            assert len(tr) == len(rt) and np.array_equal(tr["sample_counter"], rt["sample_counter"])
            for f in ("zero", "sym"):
                assert np.array_equal(tr[f].view(np.uint32), rt[f].view(np.uint32)), f"trial {15344 + c}: soft sample `{f}` differs from the oracle's"


def test_awgn_batch_tally_equals_the_oracles(sa, ob):
    """configs[4]: the BER tally of a whole batch of AWGN trials equals the oracle's row for row, and every
    trial's events are equal (tests/helpers/ber_vs_oracle.py runs the same check on a 65 536-trial batch
    and files the rows under profiles/)."""
    from helpers import ber_vs_oracle
    res = ber_vs_oracle.run(trials=8192, seed=424242, first_trial=12345)
    assert res["events_equal"] and res["tally_rows_equal"]
    assert res["rows_gpu"] == res["rows_oracle"]
    assert sum(r["trials"] for r in res["rows_gpu"]) == 8192


@pytest.mark.parametrize("rate,kernel,trials,batches", [
    (22050, "pipe", 65536, 1),          # seed 2026, trials 0 .. 65 535: includes trial 15 354, the one round 5's replay bug showed in
    (48000, "pipe", 32768, 2),          # (at 44.1 / 48 kHz the pipeline takes at most 32 768 channels a launch)
    (44100, "pipe", 32768, 2),
    (22050, "fast", 65536, 1),          # demod_fast_kernel: every strict batch beyond 65 536 channels
    (22050, "generic", 65536, 1),       # the kernel of any configuration and of every call's tail
])
def test_awgn_65536_trials_per_strict_kernel_equal_the_oracle(sa, ob, rate, kernel, trials, batches):
    """The strict-mode bug of round 5 fired in ONE trial of 65 536 with every event counter equal: the noisy batches of this suite
    were an order of magnitude too small to see it.  Every strict kernel family, at every standard rate the wavefront pipeline
    runs at, against the oracle on 65 536 noisy trials (configs[4]'s generator, seed 2026): every link event -- kind, sample
    counter, burst bytes -- and the BER tally row for row.  (profiles/r06_ber_vs_oracle_*.json: the same over 262 144 each.)"""
    from helpers import ber_vs_oracle
    done = 0
    for b in range(batches):
        res = ber_vs_oracle.run(trials=trials, rate=rate, seed=2026, first_trial=b * trials, kernel=kernel)
        assert res["events_equal"] and res["tally_rows_equal"] and res["rows_gpu"] == res["rows_oracle"]
        assert res["strict_kernel"].startswith(ber_vs_oracle.KERNELS[kernel])
        done += sum(r["trials"] for r in res["rows_gpu"])
    assert done == 65536


@pytest.mark.parametrize("n_chunks", [3, 6, 7])
def test_pipelined_matched_filters_equal_the_chunk_loop_bit_for_bit(sa, n_chunks):
    """The software-pipelined relaxed matched filters (up to 21 LDS loads in flight, waits on a partial count) add the
    same products in the same order as the chunk-at-a-time loop, whose waits are all lgkmcnt(0): equal bit for bit on
    random windows at every ring position, 42 / 84 / 98 taps."""
    import torch
    L = sa.load_library()
    L.same_debug_filter_forms.restype = C.c_int
    L.same_debug_filter_forms.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    g = torch.Generator(device="cpu").manual_seed(900 + n_chunks)
    for scale in (1.0, 1.0e-3, 37.0):
        taps = (torch.rand(n_chunks * 14, 4, generator=g) * 2 - 1).mul(2.0 / (n_chunks * 14)).cuda().contiguous()
        win = ((torch.rand(160, 64, generator=g) * 2 - 1) * scale).cuda().contiguous()
        out = torch.full((160, 64, 6), float("nan"), device="cuda")
        assert L.same_debug_filter_forms(n_chunks, taps.data_ptr(), win.data_ptr(), out.data_ptr()) == 0
        o = out.cpu().numpy().view(np.uint32)
        assert np.isfinite(out.cpu().numpy()).all()
        assert np.array_equal(o[..., 0:2], o[..., 2:4]), "demod_pair_relaxed_chunks differs from demod_pair_relaxed"
        assert np.array_equal(o[..., 0:2], o[..., 4:6]), "demod_pair_relaxed_42 differs from demod_pair_relaxed"
        assert float(out[..., 0].abs().max()) > 0.0
