#!/usr/bin/env python3
"""Round 5: the 16-channel reproduction of a strict-pipeline / oracle difference (trial 15354 of configs[4]'s first batch, seed 2026:
bytes 80-81 of an 83-byte burst, decoded from the noise behind the carrier).  Prints the burst tail per kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
from sameold_amd import montecarlo as mc
from oracle import binding as ob
rate, seed, grid, n_samples = 22050, 2026, 15, 44096
lo = 15344
x = mc.synth_trials(16, lo, n_samples, rate, seed, 0.0, 1.0, grid)
xs = x.cpu().numpy()
ref = [e.as_tuple() for e in ob.Receiver(ob.default_config(rate)).run(np.ascontiguousarray(xs[:, 10]))]
print("oracle:", [(t[0], t[1], t[2][76:]) for t in ref if t[0] <= 3])
for kw in ({}, {"generic_kernel": True}):
    rx = sa.SameReceiverBuilder(rate).build_batch(16, link_only=True, trace_symbols=True, **kw)
    rx.process_tensor(x); rx.sync()
    ev = rx.poll_events_np()
    e = ev[(ev["channel"] == 10) & (ev["kind"] <= 3)]
    print(f"{os.environ.get('SAME_PIPE','-')}/{os.environ.get('SAME_PIPE_LANES','-')}/{os.environ.get('SAME_PIPE_SPLIT','-')} {kw} [{rx.kernel_name()}]:",
          [(int(r["kind"]), int(r["sample_counter"]), bytes(r["bytes"][76:int(r["len"])])) for r in e])
    tr = rx.read_trace(10)
    sel = np.flatnonzero((tr["sample_counter"] > 36400) & (tr["sample_counter"] < 37300))
    print("   symbols:", [(int(tr[i]["sample_counter"]), float(tr[i]["zero"]).hex()[:12], float(tr[i]["sym"]).hex()[:12]) for i in sel])
