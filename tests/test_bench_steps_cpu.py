"""bench.py's step loop (run_steps) on a stub receiver, no GPU: W untimed passes and the pre-heat before the timed region, EXACTLY K
process calls inside it, the untimed launch the first timed pass collects kept out of the kernel-time mean, and -- N ranks -- the
number of pre-heat passes agreed through the hook (a pass of an N-rank run holds a collective: every rank must make the same number)."""
import os
import sys
import time
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StubRx:
    """Two launches in flight: a process call queues launch k and collects launch k - 1 (its events become pending)."""
    def __init__(self, log):
        self.log, self.launched, self.collected, self.pending, self.kms = log, 0, 0, 0, 0.0
    def process_device_ptr(self, ptr, T, layout, stream):
        self.launched += 1
        self.log.append(("process", self.launched))
        self._collect(self.launched - 1)
    def _collect(self, upto):
        while self.collected < upto:
            self.collected += 1
            self.pending += 10
            self.kms = float(self.collected)          # launch k "took" k ms: the mean tells which launches were counted
    def sync(self):
        self.log.append(("sync", self.launched)); self._collect(self.launched)
    def pending_events(self): return self.pending
    def last_kernel_ms(self): return self.kms
    def last_demod_kernel_ms(self): return self.kms
    def peek_events_np(self):
        self.log.append(("peek", self.collected))
        ev = np.zeros(self.pending, dtype=[("sample_counter", "<u8"), ("kind", "<u4")]); ev["sample_counter"] = 1
        return ev
    def drop_events(self, n): self.pending -= n
    def input_sample_counter(self): return 0


@pytest.mark.parametrize("steps,warmup,preheat", [(20, 5, 30.0), (20, 3, 0.0), (4, 1, 0.0), (3, 0, 0.0), (5, 2, 10.0)])
def test_run_steps_times_exactly_k_passes(monkeypatch, steps, warmup, preheat):
    import torch
    import bench
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    log, agreed = [], []
    rx = StubRx(log)
    sa = types.SimpleNamespace(receiver=types.SimpleNamespace(EVENT_DTYPE=np.dtype([("sample_counter", "<u8"), ("kind", "<u4")])))
    x = types.SimpleNamespace(data_ptr=lambda: 0)
    monkeypatch.setattr(bench, "PREHEAT_MS", [preheat])
    def agree(v):
        agreed.append(v); return v
    monkeypatch.setattr(bench, "AGREE", [agree])
    gathered = []
    def barrier(): log.append(("barrier", rx.launched))
    el, kms, first_ev, nb, steady = bench.run_steps(sa, rx, x, 100, None, steps, warmup, lambda r: gathered.append(r.pending_events()) or 0, barrier)
    marks = [n for what, n in log if what == "barrier"]
    assert len(marks) == 2 and marks[1] - marks[0] == steps                      # exactly K launches between the two barriers
    untimed = marks[0]
    assert untimed >= warmup                                                      # W warm-up passes (+ the pre-heat's) before the first
    if preheat > 0.0:
        assert (untimed - warmup) % 8 == 0 and len(agreed) == (untimed - warmup) // 8    # pre-heat in agreed chunks of eight passes
    else:
        assert untimed == warmup and not agreed
    # the kernel times counted are those of the K timed launches (launch k "took" k ms), not the untimed one the first pass collected
    assert kms == pytest.approx(np.mean(np.arange(untimed + 1, untimed + steps + 1)))
    assert sum(1 for what, _ in log if what == "peek") == 1                       # only the first pass's events are materialised
    assert bench.run_steps.last_launches == untimed + steps
    assert rx.pending == 0 and rx.collected == rx.launched                        # everything collected and consumed inside the region


def test_preheat_pass_count_follows_the_slowest_rank(monkeypatch):
    """The ranks stop pre-heating together: the hook returns the maximum over the ranks, so a rank whose own clock says
    "enough" goes on while another's does not."""
    import torch
    import bench
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    rx = StubRx([])
    sa = types.SimpleNamespace(receiver=types.SimpleNamespace(EVENT_DTYPE=np.dtype([("sample_counter", "<u8"), ("kind", "<u4")])))
    x = types.SimpleNamespace(data_ptr=lambda: 0)
    calls = []
    def agree(v):
        calls.append(v)
        return 0.0 if len(calls) < 4 else 1e9          # "another rank" is not done for three rounds
    monkeypatch.setattr(bench, "PREHEAT_MS", [5.0])
    monkeypatch.setattr(bench, "AGREE", [agree])
    bench.run_steps(sa, rx, x, 100, None, 2, 1, lambda r: 0, lambda: None)
    assert len(calls) == 4 and bench.run_steps.last_launches == 1 + 4 * 8 + 2
