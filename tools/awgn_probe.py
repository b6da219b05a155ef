#!/usr/bin/env python3
"""AWGN trials, strict against relaxed arithmetic, per Eb/N0 grid point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sameold_amd as sa
from sameold_amd import montecarlo as mc
n, grid, rate, seed = 8192, 15, 22050, 31
T = 2 * rate - (2 * rate) % 42
x = mc.synth_trials(n, 0, T, rate, seed, 0.0, 1.0, grid)
res = {}
for label, kw in (("strict", {}), ("relaxed", {"relaxed": True})):
    rx = sa.SameReceiverBuilder(rate).build_batch(n, link_only=True, **kw)
    rx.process_tensor(x); rx.sync()
    res[label] = rx.poll_events_np()
payloads = [sa.synth_payload(seed, c) for c in range(n)]
ta, tb = mc.new_tally(grid), mc.new_tally(grid)
mc.score_bursts(res["strict"], payloads, 0, n, grid, ta)
mc.score_bursts(res["relaxed"], payloads, 0, n, grid, tb)
def first_burst(ev):
    out = {}
    b = ev[ev["kind"] == 3]
    for r in b:
        c = int(r["channel"])
        if c not in out: out[c] = r["bytes"][:min(int(r["len"]), 288)].tobytes()
    return out
fa, fb = first_burst(res["strict"]), first_burst(res["relaxed"])
differ = np.zeros(grid, int)
for c in range(n):
    pay = payloads[c]
    a, b = fa.get(c), fb.get(c)
    ka = a[:len(pay)] if a else None; kb = b[:len(pay)] if b else None
    if ka != kb: differ[c % grid] += 1
print("Eb/N0  trials  differ  detected s/r   intact s/r   bit errors s/r")
for g in range(grid):
    print(f"{g:5d} {int(ta['trials'][g]):7d} {differ[g]:7d}  {int(ta['detected'][g]):5d}/{int(tb['detected'][g]):5d}  {int(ta['intact'][g]):5d}/{int(tb['intact'][g]):5d}  {int(ta['bit_errors'][g]):7d}/{int(tb['bit_errors'][g]):7d}")
