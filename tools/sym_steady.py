#!/usr/bin/env python3
"""Round 4: kernel time of successive launches of the 32 768-channel relaxed batch -- one at a time with a sync in between, and
back to back the way bench.py steps (launch i+1 queued while launch i runs), link layer only and with the transport layer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sameold_amd as sa

rate, n_ch = 22050, 32768
n = 44100 - 44100 % 180
x = sa.synth_afsk(n_ch, n, rate, seed=20260000)
torch.cuda.synchronize()
modes = {'link': (True,), 'transport': (False,)}.get(sys.argv[1] if len(sys.argv) > 1 else '', (True, False))
for link_only in modes:
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=link_only, relaxed=True)
    rx.set_kernel_timing(True)
    one = []
    for r in range(12):
        rx.process_tensor(x); rx.sync(); one.append(rx.last_kernel_ms()); rx.poll_events_np()
    print(f"link_only={link_only} synced launches:", " ".join(f"{m:.3f}" for m in one), flush=True)
    b2b = []
    t0 = time.perf_counter()
    for r in range(12):
        rx.process_tensor(x); b2b.append(rx.last_kernel_ms()); rx.drop_events(rx.pending_events())
    rx.sync(); b2b.append(rx.last_kernel_ms())
    dt = (time.perf_counter() - t0) / 12 * 1e3
    print(f"link_only={link_only} back to back  :", " ".join(f"{m:.3f}" for m in b2b), f"| {dt:.3f} ms per step", flush=True)
