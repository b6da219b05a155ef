"""The symbol-paced pipeline (sameold_amd/csrc/same_kernels_sym.hip): what a relaxed batch of whole 64-channel groups runs
up to 65 536 channels, and what every time-parallel chunk of a relaxed batch runs (tests/test_time_parallel.py covers
that side with every one of its tests).

Contract (include/same_rx.h, SAME_BATCH_RELAXED), against strict mode -- which tests/test_gpu_parity.py holds against the
oracle event for event:
  * every burst's transmitted bytes and every transport message EQUAL;
  * link events within SAME_TP_EVENT_TOLERANCE_SYMBOLS symbols;
  * soft symbols of an open squelch: instants within SOFT_INSTANT_TOLERANCE samples, values within SOFT_SYMBOL_TOLERANCE
    with equal sign;
  * the reference's recordings print their .txt (sample/*.txt).
The kernel is deterministic (the same call twice: the same events) and carries its whole state from call to call; a stream fed
in several calls meets the contract like one long call (not bit for bit: test_state_is_carried_from_call_to_call says why)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_time_parallel import SOFT_INSTANT_TOLERANCE, SOFT_SYMBOL_TOLERANCE, assert_contract, split, strict_events

pytestmark = pytest.mark.gpu

KERNEL = "demod_sym_kernel"


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


@pytest.fixture(autouse=True)
def default_kernels(monkeypatch):
    for k in ("SAME_RELAXED_KERNEL", "SAME_SYM", "SAME_RELAXED"):
        monkeypatch.delenv(k, raising=False)


def ordered(ev):
    return ev[np.lexsort((np.arange(len(ev)), ev["channel"]))]


def run(sa, x, rate=22050, calls=None, layout=None, **kw):
    import torch
    n_ch = x.shape[1]
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=True, **kw)
    keep = []
    off = 0
    for n in (calls or [x.shape[0]]):
        part = x[off:off + n]
        if layout == "cm":
            part = part.t().contiguous(); keep.append(part)
            rx.process_tensor(part, layout=sa.LAYOUT_CHANNEL_MAJOR)
        elif layout == "i16":
            part = part.round().to(torch.int16).contiguous(); keep.append(part)
            rx.process_tensor(part)
        else:
            rx.process_tensor(part.contiguous())
        off += n
    rx.sync()
    assert rx.kernel_name() == KERNEL
    return rx, ordered(rx.poll_events_np())


@pytest.mark.parametrize("n_ch,seconds,noise", [(64, 10.0, 0.0), (256, 10.0, 0.0), (512, 8.0, 0.05), (2048, 4.0, 0.02)])
def test_contract_against_strict_mode(sa, n_ch, seconds, noise):
    rate, seed = 22050, 500 + n_ch
    n = int(rate * seconds)
    x = sa.synth_afsk(n_ch, n, rate, seed=seed, noise_sigma=noise)
    ref = strict_events(sa, x, rate)
    _, got = run(sa, x)
    assert_contract(sa, got, ref, rate, n_ch, lambda c: sa.synth_payload(seed, c), exact_bursts=(noise == 0.0),
                    garbled_per_mille=(1 if noise > 0.0 else 0), what=f"{n_ch} channels", t_end=n)


def test_other_input_forms_and_repeatability(sa):
    """The same call twice gives the same events (nothing in the kernel depends on how its wavefronts interleave); int16
    samples give what the f32 call gives, event for event -- the kernel sees the same numbers; a channel-major buffer
    (transposed slab by slab: a relaxed batch that is not time-parallel) meets the contract."""
    rate, n_ch, seed = 22050, 128, 901
    x = sa.synth_afsk(n_ch, 22050 * 6, rate, seed=seed).round()
    _, a = run(sa, x)
    _, a2 = run(sa, x)
    _, b = run(sa, x, layout="i16")
    for other in (a2, b):
        assert np.array_equal(a["kind"], other["kind"]) and np.array_equal(a["sample_counter"], other["sample_counter"])
        assert np.array_equal(a["bytes"], other["bytes"])
    ref = strict_events(sa, x, rate)
    _, c = run(sa, x, layout="cm")
    assert_contract(sa, c, ref, rate, n_ch, lambda ch: sa.synth_payload(seed, ch), exact_bursts=True, what="channel-major", t_end=22050 * 6)


def test_state_is_carried_from_call_to_call(sa):
    """A stream fed in several calls -- whole 36-sample blocks (among them a call of a single block and cuts inside bursts),
    and calls with tails, which the strict any-configuration kernel takes -- meets the contract like one long call.  (Not
    bit for bit: what end() undoes is applied a fixed number of steps late, DESIGN.md 4.8, and a call boundary in that
    stretch moves it by a symbol; a cut anywhere else leaves exactly the state one long call has there.)"""
    rate, n_ch, seed = 22050, 192, 77
    n = 36 * 4000
    x = sa.synth_afsk(n_ch, n, rate, seed=seed)
    ref = strict_events(sa, x, rate)
    _, one = run(sa, x)
    for calls in ([36, 36 * 7, 36 * 1500, 36 * 1, 36 * 2491], [60000, 77, 35, 1, n - 60113]):
        _, many = run(sa, x, calls=calls)
        assert_contract(sa, many, ref, rate, n_ch, lambda c: sa.synth_payload(seed, c), exact_bursts=True, what=f"calls {calls}", t_end=n)
        same = sum(1 for a, b in zip(split(one, n_ch), split(many, n_ch))
                   if len(a) == len(b) and np.array_equal(a["sample_counter"], b["sample_counter"]) and np.array_equal(a["kind"], b["kind"]))
        print(f"calls {calls}: {same} of {n_ch} channels event for event as in one call")
        if all(k % 36 == 0 for k in calls):
            assert same >= n_ch * 9 // 10


def test_soft_symbols_within_the_stated_tolerance(sa):
    """The soft-symbol stream against strict mode's, symbol by symbol, while the squelch is open (Reading ... the end of the
    transmitted bytes)."""
    rate, n, n_ch = 22050, 22050 * 10, 64
    x = sa.synth_afsk(n_ch, n, rate, seed=99)
    full = sa.SameReceiverBuilder(rate).build_batch(n_ch, trace_symbols=True, link_only=True)
    full.process_tensor(x); full.sync()
    rel = sa.SameReceiverBuilder(rate).build_batch(n_ch, trace_symbols=True, link_only=True, relaxed=True)
    rel.process_tensor(x); rel.sync()
    assert rel.kernel_name() == KERNEL
    ev = split(full.poll_events_np(), n_ch)
    all_dt, all_err, sign_flips, checked = [], [], 0, 0
    for c in range(0, n_ch, 3):
        reading = ev[c][ev[c]["kind"] == 2]["sample_counter"]
        bursts = ev[c][ev[c]["kind"] == 3]["sample_counter"]
        ta, tb = full.read_trace(c, cap=4096), rel.read_trace(c, cap=4096)
        for t_read, t_burst in zip(reading, bursts):
            t_read, t_burst = int(t_read), int(t_burst)
            a = ta[(ta["sample_counter"] > t_read) & (ta["sample_counter"] < t_burst - 5 * 8 * 43)]
            b = tb[(tb["sample_counter"] > t_read - 64) & (tb["sample_counter"] < t_burst + 64)]
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
    assert checked >= 20
    dt, err = np.concatenate(all_dt), np.concatenate(all_err)
    stats = (f"{len(dt)} symbols of {checked} bursts: instants max {dt.max()} samples apart (mean {dt.mean():.2f}); "
             f"soft symbols max |diff| {err.max():.4f}, {sign_flips} sign differences")
    print(stats)
    assert sign_flips == 0, stats
    assert dt.max() <= SOFT_INSTANT_TOLERANCE, stats
    assert err.max() <= SOFT_SYMBOL_TOLERANCE, stats


@pytest.mark.parametrize("name", ["npt", "two_and_two", "long_message"])
def test_golden_recordings(sa, name):
    """The reference's recordings (samedec's configuration, 64 copies with different lead-ins, the end-of-file flush): the
    decoded text equals the .txt -- headers bit-exact, as the north star asks."""
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
    assert rx.kernel_name() == KERNEL
    rx.flush()
    rx.sync()
    ev = split(ordered(rx.poll_events_np()), n_ch)
    for c in range(n_ch):
        lines = [r["bytes"][: int(r["len"])].tobytes().decode() if r["kind"] == sa.TRANSPORT_MSG_START else "NNNN"
                 for r in ev[c] if r["kind"] in (sa.TRANSPORT_MSG_START, sa.TRANSPORT_MSG_END)]
        assert lines == exp, f"lead-in {lead[c]}: {lines}"


def test_disabled_equalizer_and_samedec_limits(sa):
    """The other builds of the kernel: the disabled equalizer (1 + 1 taps) and samedec's AGC limits."""
    rate, n_ch, seed = 22050, 128, 3131
    x = sa.synth_afsk(n_ch, 22050 * 8, rate, seed=seed)
    for make in (lambda: sa.SameReceiverBuilder(rate).without_adaptive_equalizer(), lambda: sa.SameReceiverBuilder(rate).samedec()):
        ref = strict_events(sa, x, rate, builder=make())
        rx = make().build_batch(n_ch, relaxed=True)
        rx.process_tensor(x); rx.sync()
        assert rx.kernel_name() == KERNEL
        assert_contract(sa, ordered(rx.poll_events_np()), ref, rate, n_ch, lambda c: sa.synth_payload(seed, c), exact_bursts=True)


@pytest.mark.parametrize("n_ch", [32768, 65536, 131072])
def test_large_batches(sa, n_ch):
    """The regimes the bench prices: the 32 768-channel shard of configs[3] (two workgroups per CU, everything resident) and
    65 536 / 131 072 channels (two / four rounds of workgroups; the latter is bench.py's `scaled_big` block).  Strict mode on the same input is the reference for every channel; strict
    mode itself against the oracle on a slab of them."""
    from oracle import binding as ob
    from helpers.oracle_compare import assert_every_channel_matches_oracle
    rate, seed = 22050, 20260000
    n = 2 * rate
    x = sa.synth_afsk(n_ch, n, rate, seed=seed)
    ref = strict_events(sa, x, rate)
    _, got = run(sa, x)
    assert_contract(sa, got, ref, rate, n_ch, lambda c: sa.synth_payload(seed, c), exact_bursts=True, what=f"{n_ch} channels", t_end=n)
    slab = slice(n_ch // 2, n_ch // 2 + 1024)
    sub = ref[(ref["channel"] >= slab.start) & (ref["channel"] < slab.stop)].copy()
    sub["channel"] -= slab.start
    assert assert_every_channel_matches_oracle(ob, ob.default_config(rate), x[:, slab], sub) > 1024
