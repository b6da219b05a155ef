#!/usr/bin/env python3
"""Wall time per bench step against the kernel time, with more and more of the per-step host work
switched on: A launch + harvest only, B + draining the event queue into numpy, C + packing the bursts
(what bench.py does).  The GPU kernel runs concurrently with all of it; a step longer than the
kernel means the host (or a gap between launches) is on the critical path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
from sameold_amd import distributed as sd
C, T = 4096, 220500
x = sa.synth_afsk(C, T, 22050, seed=20260000); torch.cuda.synchronize()
for mode in "ABC":
    rx = sa.SameReceiverBuilder(22050).build_batch(C); rx.set_kernel_timing(True)
    acc = {"process": 0.0, "poll": 0.0, "pack": 0.0}; n = 0; kms = []
    for k in range(15):
        if k == 3:
            rx.sync(); torch.cuda.synchronize(); t_start = time.perf_counter()
        t0 = time.perf_counter(); rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_TIME_MAJOR, None)
        t1 = time.perf_counter()
        ev = rx.poll_events_np() if mode in "BC" else None
        t2 = time.perf_counter()
        if mode == "C": sd.pack_burst_events(ev, 0, zero_padded=True)
        t3 = time.perf_counter()
        if k >= 3:
            acc["process"] += t1 - t0; acc["poll"] += t2 - t1; acc["pack"] += t3 - t2; n += 1
            kms.append(rx.last_kernel_ms())
    rx.sync(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t_start) / n * 1e3
    print(mode, {k: round(1e3 * v / n, 2) for k, v in acc.items()}, f"ms host per step; wall {wall:.2f} ms per step; kernel {np.mean(kms):.2f} ms", flush=True)
    del rx
