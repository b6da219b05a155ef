#!/usr/bin/env python3
"""Kernel time of N launches enqueued back to back (GPU continuously busy) vs with an idle gap
after each.  usage: python tools/dvfs_probe.py <gap_ms> [channels]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa
gap = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = 220500 if C <= 8192 else 44100
x = sa.synth_afsk(C, T, 22050, seed=20260000)
torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, link_only=True)
rx.set_kernel_timing(True)
ms = []
t0 = time.perf_counter()
for k in range(14):
    rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_TIME_MAJOR, None)   # harvests launch k-1 (waits for it) after enqueueing k
    if k: ms.append(rx.last_kernel_ms())
    rx.poll_events_np()
    if gap: rx.sync(); time.sleep(gap * 1e-3)
rx.sync()
print(f"gap {gap} ms, {C} ch: kernel ms", " ".join(f"{m:.1f}" for m in ms), f" wall {1e3*(time.perf_counter()-t0)/14:.1f} ms/launch")
