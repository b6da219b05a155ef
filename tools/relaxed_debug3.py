#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
def run(name, kw, show=False):
    pcm = np.fromfile(os.path.join("tests/golden", f"{name}.22050.s16le.bin"), dtype="<i2").astype(np.float32)
    n_ch = 64
    lead = [211 * c for c in range(n_ch)]
    n = len(pcm) + max(lead)
    x = np.zeros((n, n_ch), np.float32)
    for c in range(n_ch):
        x[lead[c]:lead[c] + len(pcm), c] = pcm
    rx = sa.SameReceiverBuilder(22050).samedec().build_batch(n_ch, **kw)
    rx.process_tensor(torch.from_numpy(x).cuda()); rx.flush(); rx.sync()
    ev = rx.poll_events_np()
    nmsg = [int(((ev["channel"] == c) & (ev["kind"] >= 18)).sum()) for c in range(n_ch)]
    bad = [c for c in range(n_ch) if nmsg[c] == 0]
    print(name, kw, rx.kernel_name(), "channels without a message:", bad)
    return ev, bad
run("npt", {"relaxed": True}); run("two_and_two", {"relaxed": True})
ev, bad = run("long_message", {"relaxed": True})
evs, _ = run("long_message", {})
for c in bad[:2]:
    for label, e in (("relaxed", ev), ("strict", evs)):
        for r in e[e["channel"] == c]:
            print(f"   {label} ch {c} t={int(r['sample_counter']):8d} kind={int(r['kind'])} len={int(r['len'])} sym={int(r['symbol_count'])} {r['bytes'][:min(int(r['len']),255)].tobytes()!r}")
