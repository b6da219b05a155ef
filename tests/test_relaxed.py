"""Relaxed arithmetic, SAME_BATCH_RELAXED (same_kernels_relaxed.hip; include/same_rx.h has the contract).

The reference's algorithm with the rounding of its floating-point expressions given up: matched filters as fused
multiply-adds into partial sums (rx/filter.rs:363-377 re-associated), |mark| and |space| as f32 square roots
(rx/demod.rs:163), the AGC update as gain * (1 - bw |x|) + bw (rx/agc.rs:72-77), reciprocals for the divisions of
timing loop and equalizer.  The timing trajectory is chaotic in the last bit of those sums (SURVEY.md section 7 hard
part 1), so against strict mode -- itself compared with the oracle on every channel, here as everywhere -- the
contract is the time-parallel mode's (tests/test_time_parallel.py::assert_contract):
  * bursts: their number and order per channel and every transmitted byte EQUAL; transport messages EQUAL;
  * link events of a delivered burst within TP_EVENT_TOLERANCE_SYMBOLS symbols;
  * soft symbols of an open squelch: instants within SOFT_INSTANT_TOLERANCE samples, values within
    SOFT_SYMBOL_TOLERANCE = 0.05 with equal sign (stated tolerance of the soft-symbol stream in this mode);
  * noisy input: bursts both deliver are equal, the number of unmatched bursts bounded, AWGN tallies statistically equal.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_time_parallel import (SOFT_INSTANT_TOLERANCE, SOFT_SYMBOL_TOLERANCE, assert_awgn_tallies_equal, assert_contract, split,
                                strict_events)

pytestmark = pytest.mark.gpu


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


@pytest.fixture(autouse=True, params=["solo", "duo"])
def wave_kernel(request, monkeypatch):
    """The relaxed kernels proper: one wavefront per 64 channels ("solo") or two ("duo": sample phase | instants).  Without
    SAME_RELAXED_KERNEL a relaxed batch of up to 32 768 channels runs the pipeline's FASTMATH build, which
    tests/test_time_parallel.py and test_relaxed_batches_on_the_pipeline cover.  Read when a batch is created."""
    monkeypatch.setenv("SAME_RELAXED_KERNEL", request.param)
    return request.param


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.lib()
    return binding


def wave_kernel_param(monkeypatch):
    """What the autouse fixture set for this run."""
    import os
    return os.environ.get("SAME_RELAXED_KERNEL")


def relaxed_events(sa, x, rate, builder=None, calls=None, **kw):
    n_ch = x.shape[1]
    rx = (builder or sa.SameReceiverBuilder(rate)).build_batch(n_ch, relaxed=True, **kw)
    off = 0
    for n in (calls or [x.shape[0]]):
        rx.process_tensor(x[off:off + n].contiguous())
        off += n
    assert off == x.shape[0]
    rx.sync()
    got = rx.poll_events_np()
    return rx, got[np.lexsort((np.arange(len(got)), got["channel"]))]


@pytest.mark.parametrize("n_ch,seconds,noise", [(256, 10.0, 0.0), (512, 8.0, 0.05), (130, 9.0, 0.0), (2048, 4.0, 0.02)])
def test_relaxed_meets_the_contract(sa, ob, n_ch, seconds, noise):
    """One wavefront per 64 channels (130: a partly filled one), clean and noisy, against strict mode -- which the same
    test holds against the oracle on every channel."""
    from helpers.oracle_compare import assert_every_channel_matches_oracle
    rate = 22050
    n = int(rate * seconds)
    x = sa.synth_afsk(n_ch, n, rate, seed=500 + n_ch, noise_sigma=noise)
    ref = strict_events(sa, x, rate)
    if n_ch <= 512:
        assert_every_channel_matches_oracle(ob, ob.default_config(rate), x, ref)
    rx, got = relaxed_events(sa, x, rate)
    assert rx.kernel_name() == "demod_relaxed_kernel"
    assert len(got[got["kind"] == 3]) >= 2 * n_ch * (seconds / 10.0) * 0.8
    worst = assert_contract(sa, got, ref, rate, n_ch, lambda c: sa.synth_payload(500 + n_ch, c), exact_bursts=(noise == 0.0), what="relaxed",
                            garbled_per_mille=(1 if noise > 0.0 else 0))
    print("event instants, worst difference in samples:", worst)


def test_relaxed_streaming_calls_and_input_forms(sa):
    """Calls of any length continue the stream (whole 42-sample blocks go to the relaxed kernel, what is left of a call
    to the any-configuration kernel on the same state); int16 input is cast in the kernel."""
    import torch
    rate, n_ch, n = 22050, 192, 22050 * 9
    x = torch.round(sa.synth_afsk(n_ch, n, rate, seed=61)).contiguous()
    ref = strict_events(sa, x, rate)
    pay = lambda c: sa.synth_payload(61, c)
    calls = [40000, 41, 1, 60017, 42 * 500, n - (40000 + 41 + 1 + 60017 + 42 * 500)]
    _, got = relaxed_events(sa, x, rate, calls=calls)
    assert_contract(sa, got, ref, rate, n_ch, pay, what="streaming")
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=True)
    rx.process_tensor(x.to(torch.int16)); rx.sync()
    assert_contract(sa, rx.poll_events_np(), ref, rate, n_ch, pay, what="i16")
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=True)
    rx.process_tensor(x.t().contiguous(), layout=sa.LAYOUT_CHANNEL_MAJOR); rx.sync()
    got = rx.poll_events_np()
    assert_contract(sa, got[np.lexsort((np.arange(len(got)), got["channel"]))], ref, rate, n_ch, pay, what="channel-major")


def soft_symbol_differences(full, rel, n_ch, rate, every=3):
    """Strict (full) against relaxed (rel) soft symbols while the squelch is open: per compared symbol the distance of the
    instants in samples and of the values, the number of sign differences, the number of bursts looked at."""
    sps = rate / 520.83
    ev = split(full.poll_events_np(), n_ch)
    all_dt, all_err, sign_flips, checked = [], [], 0, 0
    for c in range(0, n_ch, every):
        reading = ev[c][ev[c]["kind"] == 2]["sample_counter"]
        bursts = ev[c][ev[c]["kind"] == 3]["sample_counter"]
        ta, tb = full.read_trace(c, cap=4096), rel.read_trace(c, cap=4096)
        for t_read, t_burst in zip(reading, bursts):
            t_read, t_burst = int(t_read), int(t_burst)
            a = ta[(ta["sample_counter"] > t_read) & (ta["sample_counter"] < t_burst - int(5 * 8 * sps + 8))]
            b = tb[(tb["sample_counter"] > t_read - int(3 * sps)) & (tb["sample_counter"] < t_burst + int(3 * sps))]
            if len(a) < 50 or len(b) < 50:
                continue
            idx = np.clip(np.searchsorted(b["sample_counter"], a["sample_counter"]), 1, len(b) - 1)
            ta_ = a["sample_counter"].astype(np.int64)
            near = np.where(np.abs(b["sample_counter"][idx].astype(np.int64) - ta_)
                            < np.abs(b["sample_counter"][idx - 1].astype(np.int64) - ta_), idx, idx - 1)
            all_dt.append(np.abs(b["sample_counter"][near].astype(np.int64) - ta_))
            all_err.append(np.abs(b["sym"][near] - a["sym"]))
            sign_flips += int(np.sum(np.sign(b["sym"][near]) != np.sign(a["sym"])))
            checked += 1
    return np.concatenate(all_dt), np.concatenate(all_err), sign_flips, checked


def test_relaxed_soft_symbols_within_the_stated_tolerance(sa):
    """The soft-symbol stream of the relaxed kernel against strict mode's, symbol by symbol, while the squelch is open
    (Reading ... the end of the transmitted bytes): instants within SOFT_INSTANT_TOLERANCE samples, values within
    SOFT_SYMBOL_TOLERANCE, equal sign."""
    rate, n, n_ch = 22050, 22050 * 10, 64
    x = sa.synth_afsk(n_ch, n, rate, seed=99)
    full = sa.SameReceiverBuilder(rate).build_batch(n_ch, trace_symbols=True, link_only=True)
    full.process_tensor(x); full.sync()
    rel = sa.SameReceiverBuilder(rate).build_batch(n_ch, trace_symbols=True, link_only=True, relaxed=True)
    rel.process_tensor(x); rel.sync()
    assert rel.kernel_name() == "demod_relaxed_kernel"
    dt, err, sign_flips, checked = soft_symbol_differences(full, rel, n_ch, rate)
    assert checked >= 20
    stats = (f"{len(dt)} symbols of {checked} bursts: instants max {dt.max()} samples apart (mean {dt.mean():.2f}); "
             f"soft symbols max |diff| {err.max():.4f}, {np.mean(err <= SOFT_SYMBOL_TOLERANCE):.5f} within {SOFT_SYMBOL_TOLERANCE}, "
             f"{sign_flips} sign differences")
    print(stats)
    assert sign_flips == 0, stats
    assert dt.max() <= SOFT_INSTANT_TOLERANCE, stats
    assert err.max() <= SOFT_SYMBOL_TOLERANCE, stats


@pytest.mark.parametrize("name", ["npt", "two_and_two", "long_message"])
def test_relaxed_golden_recordings(sa, name):
    """The reference's recordings through the relaxed kernel (samedec's configuration, 64 copies with different
    lead-ins, the end-of-file flush): the decoded text equals the .txt -- headers bit-exact, as the north star asks."""
    import torch
    pcm = np.fromfile(os.path.join(GOLDEN, f"{name}.22050.s16le.bin"), dtype="<i2").astype(np.float32)
    n_ch = 64
    lead = [211 * c for c in range(n_ch)]
    n = len(pcm) + max(lead)
    x = np.zeros((n, n_ch), np.float32)
    for c in range(n_ch):
        x[lead[c]:lead[c] + len(pcm), c] = pcm
    exp = [ln for ln in open(os.path.join(GOLDEN, f"{name}.22050.s16le.txt")).read().splitlines() if ln != "+OK"]
    rx = sa.SameReceiverBuilder(22050).samedec().build_batch(n_ch, relaxed=True)
    rx.process_tensor(torch.from_numpy(x).cuda())
    rx.flush()
    rx.sync()
    assert rx.kernel_name() == "demod_relaxed_kernel"
    got = rx.poll_events_np()                       # two harvests (the recording, the flush): bring them into channel order
    ev = split(got[np.lexsort((np.arange(len(got)), got["channel"]))], n_ch)
    for c in range(n_ch):
        lines = [r["bytes"][: int(r["len"])].tobytes().decode() if r["kind"] == sa.TRANSPORT_MSG_START else "NNNN"
                 for r in ev[c] if r["kind"] in (sa.TRANSPORT_MSG_START, sa.TRANSPORT_MSG_END)]
        assert lines == exp, f"lead-in {lead[c]}: {lines}"


def test_relaxed_awgn_tally_statistically_equal(sa):
    """configs[4] through the relaxed kernel: detection, intact headers and bit errors per Eb/N0 grid point statistically
    equal to strict mode's; trial by trial where strict mode decodes every header (test_time_parallel has the reasoning)."""
    from sameold_amd import montecarlo as mc
    n, grid, rate, seed = 8192, 15, 22050, 31
    T = 2 * rate - (2 * rate) % 42
    x = mc.synth_trials(n, 0, T, rate, seed, 0.0, 1.0, grid)
    ref = strict_events(sa, x, rate, link_only=True)
    _, got = relaxed_events(sa, x, rate, link_only=True)
    payloads = [sa.synth_payload(seed, c) for c in range(n)]
    assert_awgn_tallies_equal(sa, got, ref, payloads, n, grid, strict_arithmetic=False, tight=True)


def test_relaxed_batches_on_the_pipeline(sa, monkeypatch):
    """The default for a relaxed batch of whole 64-channel groups up to 32 768 channels: the wavefront pipeline's FASTMATH
    build on an ordinary launch (no time-parallel cut).  Same contract."""
    monkeypatch.delenv("SAME_RELAXED_KERNEL", raising=False)
    rate, n_ch, n = 22050, 512, 22050 * 8
    for noise, seed in ((0.0, 71), (0.05, 72)):
        x = sa.synth_afsk(n_ch, n, rate, seed=seed, noise_sigma=noise)
        ref = strict_events(sa, x, rate)
        rx, got = relaxed_events(sa, x, rate, calls=[60000, 77, n - 60077])
        assert rx.kernel_name() == "demod_sym_kernel"             # (22.05 kHz; the other rates: test_relaxed_batches_on_the_pipeline_at_the_other_rates)
        assert_contract(sa, got, ref, rate, n_ch, lambda c: sa.synth_payload(seed, c), exact_bursts=(noise == 0.0), what="pipeline fastmath",
                        garbled_per_mille=(1 if noise > 0.0 else 0), t_end=n)


@pytest.mark.parametrize("sym", ["1", "0"])
@pytest.mark.parametrize("rate,n_ch", [(48000, 256), (44100, 128)])
def test_relaxed_batches_on_the_pipeline_at_the_other_rates(sa, monkeypatch, rate, n_ch, sym):
    """44.1 and 48 kHz (84 / 92 taps): since round 6 the symbol-paced pipeline in 72-sample steps, one group of 64 columns per CU
    (same_kernels_sym.hip: SymGeom<84> / <92>); SAME_SYM=0 still selects round 3's FASTMATH build of the 32- / 36-sample
    pipeline with the DC wavefront.  Same contract; state carried over calls that are not whole blocks."""
    monkeypatch.delenv("SAME_RELAXED_KERNEL", raising=False)
    monkeypatch.setenv("SAME_SYM", sym)
    kernel = "demod_sym_kernel" if sym == "1" else "demod_pipe_kernel<fastmath>"
    n = rate * 6
    for noise, seed in ((0.0, 81), (0.03, 82)):
        x = sa.synth_afsk(n_ch, n, rate, seed=seed, noise_sigma=noise)
        ref = strict_events(sa, x, rate)
        rx, got = relaxed_events(sa, x, rate, calls=[2 * rate + 11, 77, n - 2 * rate - 88])
        assert rx.kernel_name() == kernel
        assert_contract(sa, got, ref, rate, n_ch, lambda c: sa.synth_payload(seed, c), exact_bursts=(noise == 0.0), what=f"pipeline fastmath {rate}",
                        garbled_per_mille=(1 if noise > 0.0 else 0), t_end=n)
    # int16 input and a channel-major buffer (transposed on the device) give what f32 time-major gives
    import torch
    xi = torch.round(sa.synth_afsk(n_ch, n, rate, seed=84)).contiguous()
    ref = strict_events(sa, xi, rate)
    for what, feed in (("i16", lambda r: r.process_tensor(xi.to(torch.int16))),
                       ("channel-major", lambda r: r.process_tensor(xi.t().contiguous(), layout=sa.LAYOUT_CHANNEL_MAJOR))):
        rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=True)
        feed(rx); rx.sync()
        assert rx.kernel_name() == kernel
        got = rx.poll_events_np()
        assert_contract(sa, got[np.lexsort((np.arange(len(got)), got["channel"]))], ref, rate, n_ch, lambda c: sa.synth_payload(84, c), what=f"{what} {rate}", t_end=n)
    # its soft symbols
    x = sa.synth_afsk(64, n, rate, seed=83)
    full = sa.SameReceiverBuilder(rate).build_batch(64, trace_symbols=True, link_only=True)
    rel = sa.SameReceiverBuilder(rate).build_batch(64, trace_symbols=True, link_only=True, relaxed=True)
    for r in (full, rel):
        r.process_tensor(x); r.sync()
    assert rel.kernel_name() == kernel
    dt, err, sign_flips, checked = soft_symbol_differences(full, rel, 64, rate, every=5)
    stats = f"{len(dt)} symbols of {checked} bursts: instants max {dt.max()} samples apart, soft symbols max |diff| {err.max():.4f}, {sign_flips} sign differences"
    print(stats)
    assert checked >= 10 and sign_flips == 0, stats
    assert dt.max() <= SOFT_INSTANT_TOLERANCE * rate // 22050 + 1 and err.max() <= SOFT_SYMBOL_TOLERANCE, stats


def test_large_relaxed_batches_at_48_khz_stay_relaxed(sa, monkeypatch):
    """More channels than the relaxed pipeline holds at once (16 384 at 44.1 / 48 kHz: one group of 64 columns per CU): its workgroups
    run in rounds; the batch does not fall back to the strict kernels (round 4 did, whatever the flag said).  Same contract."""
    monkeypatch.delenv("SAME_RELAXED_KERNEL", raising=False)
    rate, n_ch = 48000, 40960
    n = rate * 3
    x = sa.synth_afsk(n_ch, n, rate, seed=91)
    ref = strict_events(sa, x, rate)
    rx, got = relaxed_events(sa, x, rate, calls=[rate + 7, n - rate - 7])
    assert rx.kernel_name() == "demod_sym_kernel"
    assert_contract(sa, got, ref, rate, n_ch, lambda c: sa.synth_payload(91, c), what="40 960 channels at 48 kHz", t_end=n)


def test_configurations_without_a_relaxed_kernel_run_strict(sa):
    """Other sample rates, other equalizer orders, a negative AGC floor: SAME_BATCH_RELAXED is accepted and the batch
    runs the strict kernels, bit for bit."""
    for rate, eq, agc in ((32000, None, None), (22050, (8, 3), None), (22050, None, (-1.0, 1.0e6))):
        b = sa.SameReceiverBuilder(rate)
        if eq:
            b.with_adaptive_equalizer(eq[0], eq[1], 0.05, 1.0e-5)
        if agc:
            b.with_agc_gain_limits(*agc)
        x = sa.synth_afsk(64, rate * 4, rate, seed=3)
        ref = strict_events(sa, x, rate, builder=b)
        rx, got = relaxed_events(sa, x, rate, builder=b)
        assert rx.kernel_name() != "demod_relaxed_kernel"
        assert np.array_equal(got["kind"], ref["kind"]) and np.array_equal(got["sample_counter"], ref["sample_counter"])
        assert np.array_equal(got["bytes"], ref["bytes"])


@pytest.mark.parametrize("layout", ["time_major", "channel_major"])
def test_time_parallel_chunks_on_the_relaxed_kernel(sa, monkeypatch, layout):
    """SAME_TP_KERNEL=wave: the chunks of a time-parallel call on the one-wavefront relaxed kernel (what batches too large
    for the pipeline get), uniform cut and per-channel boundaries; two calls, the second on carried state."""
    monkeypatch.setenv("SAME_TP_KERNEL", "wave")
    rate, n_ch = 22050, 256
    n = 22050 * 10
    n -= n % 1260
    x = sa.synth_afsk(n_ch, 2 * n, rate, seed=4242)
    ref = strict_events(sa, x, rate)
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, time_parallel=True)
    rx.time_parallel_config(max_chunks=8)
    for part in (x[:n], x[n:]):
        if layout == "channel_major":
            rx.process_tensor(part.t().contiguous(), layout=sa.LAYOUT_CHANNEL_MAJOR)
            assert rx.time_parallel_per_channel()
        else:
            rx.process_tensor(part.contiguous())
        assert rx.time_parallel_chunks() >= 7 and rx.kernel_name() == "demod_relaxed_kernel"
    rx.sync()
    got = rx.poll_events_np()
    got = got[np.lexsort((np.arange(len(got)), got["channel"]))]
    # (t_end: a message whose hold time runs out within a few symbols of the end of the input is reported by the mode
    # whose host-side symbol clock runs a few symbols fast and not yet by the other)
    assert_contract(sa, got, ref, rate, n_ch, lambda c: sa.synth_payload(4242, c), what=layout, t_end=2 * n)


@pytest.mark.parametrize("n_ch,kernel", [(131072, "solo"), (65536, "duo")])
def test_relaxed_large_batches_meet_the_contract(sa, ob, monkeypatch, n_ch, kernel):
    """The one- and two-wavefront relaxed kernels at the sizes they were built for: 131 072 channels x 2 s on the
    one-wavefront kernel (its two-per-SIMD build) and 65 536 channels on the two-wavefront kernel (SAME_RELAXED_KERNEL; by
    default such batches run the symbol-paced pipeline: tests/test_sym_kernel.py).  Every channel against strict mode; strict
    mode against the oracle on a 4 096-channel slab."""
    from helpers.oracle_compare import assert_every_channel_matches_oracle
    if wave_kernel_param(monkeypatch) != "solo":
        pytest.skip("one run is enough: the kernel is chosen below")
    monkeypatch.setenv("SAME_RELAXED_KERNEL", kernel)
    rate, seed = 22050, 780
    n = 2 * rate
    x = sa.synth_afsk(n_ch, n, rate, seed=seed)
    ref = strict_events(sa, x, rate)
    rx, got = relaxed_events(sa, x, rate)
    assert rx.kernel_name() == "demod_relaxed_kernel"
    assert_contract(sa, got, ref, rate, n_ch, lambda c: sa.synth_payload(seed, c), exact_bursts=True, what=f"{n_ch} channels", t_end=n)
    slab = slice(n_ch // 2, n_ch // 2 + 4096)
    sub = ref[(ref["channel"] >= slab.start) & (ref["channel"] < slab.stop)].copy()
    sub["channel"] -= slab.start
    assert assert_every_channel_matches_oracle(ob, ob.default_config(rate), x[:, slab].contiguous(), sub) > 4096
