#!/usr/bin/env python3
"""Where does the relaxed kernel's decode of a recording differ from strict mode's?  python tools/relaxed_debug.py NAME LEAD"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
name = sys.argv[1]; leads = [int(a) for a in sys.argv[2:]]
pcm = np.fromfile(os.path.join("tests/golden", f"{name}.22050.s16le.bin"), dtype="<i2").astype(np.float32)
n_ch = 64
n = len(pcm) + max(leads)
x = np.zeros((n, n_ch), np.float32)
for c in range(n_ch):
    l = leads[c % len(leads)]
    x[l:l + len(pcm), c] = pcm
xt = torch.from_numpy(x).cuda()
out = {}
for label, kw in (("strict", {}), ("relaxed", {"relaxed": True})):
    rx = sa.SameReceiverBuilder(22050).samedec().build_batch(n_ch, trace_symbols=True, **kw)
    rx.process_tensor(xt); rx.flush(); rx.sync()
    ev = rx.poll_events_np()
    out[label] = (rx, ev)
for c in range(len(leads)):
    print("== lead", leads[c])
    for label in ("strict", "relaxed"):
        rx, ev = out[label]
        e = ev[ev["channel"] == c]
        for r in e:
            k = int(r["kind"])
            if k == 3 or k >= 18:
                print(f"  {label:8s} t={int(r['sample_counter']):8d} kind={k} len={int(r['len'])} {r['bytes'][:min(int(r['len']),60)].tobytes()!r}")
            elif k <= 2:
                print(f"  {label:8s} t={int(r['sample_counter']):8d} kind={k}")
    ta = out["strict"][0].read_trace(c, cap=4096); tb = out["relaxed"][0].read_trace(c, cap=4096)
    m = min(len(ta), len(tb))
    d = np.nonzero(ta["sample_counter"][:m] != tb["sample_counter"][:m])[0]
    print("  traces:", len(ta), len(tb), "first instant difference at symbol", (int(d[0]), int(ta["sample_counter"][d[0]]), int(tb["sample_counter"][d[0]])) if len(d) else None)
    big = np.nonzero(np.abs(ta["sym"][:m] - tb["sym"][:m]) > 0.05)[0]
    print("  soft symbols differing by more than 0.05 (same index):", len(big), big[:10])
