#!/usr/bin/env python3
"""Host-side time per bench step, split by call (the GPU kernel runs concurrently)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
from sameold_amd import distributed as sd
C, T = 4096, 220500
x = sa.synth_afsk(C, T, 22050, seed=20260000); torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C); rx.set_kernel_timing(True)
acc = {"process": 0.0, "poll": 0.0, "pack": 0.0}; n = 0
for k in range(12):
    t0 = time.perf_counter(); rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_TIME_MAJOR, 0)
    t1 = time.perf_counter(); ev = rx.poll_events_np()
    t2 = time.perf_counter(); recs = sd.pack_burst_events(ev, 0, zero_padded=True)
    t3 = time.perf_counter()
    if k >= 3:
        acc["process"] += t1 - t0; acc["poll"] += t2 - t1; acc["pack"] += t3 - t2; n += 1
rx.sync()
print({k: round(1e3 * v / n, 2) for k, v in acc.items()}, "ms per step; events per step", len(ev), "kernel ms", round(rx.last_kernel_ms(), 2))
