#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
from relaxed_probe import split, bursts

def parity(kernel, n_ch=256, secs=10.0, noise=0.0, seed=11):
    os.environ["SAME_RELAXED_KERNEL"] = kernel
    rate = 22050
    n = int(rate * secs); n -= n % 42
    x = sa.synth_afsk(n_ch, n, rate, seed=seed, noise_sigma=noise)
    out = {}
    for name, kw in (("strict", {}), ("relaxed", {"relaxed": True})):
        rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, **kw)
        rx.process_tensor(x); rx.sync()
        out[name] = split(rx.poll_events_np(), n_ch)
    nb = bad = 0; worst = 0
    for c in range(n_ch):
        pay = sa.synth_payload(seed, c)
        a, b = bursts(out["strict"][c]), bursts(out["relaxed"][c])
        nb += len(a)
        if len(a) != len(b): bad += 1; continue
        for (ta, ba), (tb, bb) in zip(a, b):
            k = len(pay) if ba[:4] == pay[:4] else 4
            worst = max(worst, abs(ta - tb))
            bad += ba[:k] != bb[:k]
    ma = [m for c in range(n_ch) for m in out["strict"][c][out["strict"][c]["kind"] >= 18]["bytes"].tobytes()]
    mb = [m for c in range(n_ch) for m in out["relaxed"][c][out["relaxed"][c]["kind"] >= 18]["bytes"].tobytes()]
    print(f"[{kernel}] noise {noise}: {nb} bursts, {bad} differing, burst instants at most {worst} samples apart, messages equal {ma == mb}", flush=True)

