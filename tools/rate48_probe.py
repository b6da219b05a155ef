#!/usr/bin/env python3
"""Round 4: configs[2] (16 384 channels at 48 kHz, 2 s per step) and its 44.1 kHz sibling: strict pipeline against its FASTMATH
build, kernel time back to back.   python tools/rate48_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
for rate in (48000, 44100):
    C, T = 16384, rate * 2
    x = sa.synth_afsk(C, T, rate, seed=778); torch.cuda.synchronize()
    for relaxed, link_only in ((False, True), (True, True), (False, False), (True, False)):
        rx = sa.SameReceiverBuilder(rate).build_batch(C, relaxed=relaxed, link_only=link_only); rx.set_kernel_timing(True)
        ms = []
        for k in range(11):
            rx.process_tensor(x)
            if k: ms.append(rx.last_kernel_ms())
            rx.drop_events(rx.pending_events())
        rx.sync(); ms.append(rx.last_kernel_ms())
        best = min(ms[2:])
        print(f"{rate} Hz {C} ch x {T}: relaxed={relaxed} link_only={link_only} [{rx.kernel_name()}] kernel ms min {best:.3f} mean {np.mean(ms[2:]):.3f} = {4*C*T/best/1e9/8*100:.2f} % of 8 TB/s", flush=True)
