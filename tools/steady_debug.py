#!/usr/bin/env python3
"""Steady-state passes of the bench configuration: which channels differ between strict and time-parallel, and how."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
rate, n_ch, n, seed = 22050, 4096, 220500, 20260000
x = sa.synth_afsk(n_ch, n, rate, seed=seed)
xc = x.t().contiguous()
strict = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True)
tp = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, time_parallel=True)
for k in range(2):
    strict.process_tensor(x); strict.sync()
    tp.process_tensor(xc, layout=sa.LAYOUT_CHANNEL_MAJOR); tp.sync()
    ref, got = strict.poll_events_np(), tp.poll_events_np()
t0 = n
shown = 0
for c in range(n_ch):
    a = ref[(ref["channel"] == c)]; b = got[(got["channel"] == c)]
    ba = [(int(r["sample_counter"]) - t0, r["bytes"][:int(r["len"])].tobytes()) for r in a[a["kind"] == 3]]
    bb = [(int(r["sample_counter"]) - t0, r["bytes"][:int(r["len"])].tobytes()) for r in b[b["kind"] == 3]]
    pay = sa.synth_payload(seed, c)
    cut = lambda x_: x_[: (len(pay) if x_[:4] == pay[:4] else 4)]
    if len(ba) != len(bb) or any(cut(p[1]) != cut(q[1]) for p, q in zip(ba, bb)):
        shown += 1
        if shown > 6: continue
        print("== channel", c)
        for label, e in (("strict", a), ("tp", b)):
            print("  ", label, [(int(r["kind"]), int(r["sample_counter"]) - t0, r["bytes"][:min(int(r["len"]), 12)].tobytes() if r["kind"] == 3 else b"") for r in e])
print("channels differing:", shown)
