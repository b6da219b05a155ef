#!/usr/bin/env python3
"""Round 5: configs[2]'s launch (16 384 channels x 96 000 samples at 48 kHz; or 44.1 kHz) in relaxed arithmetic, launch by launch
with the transport layer on.   python tools/rate48_once.py [48000|44100]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa
rate = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
n_ch, n = 16384, 2 * rate
x = sa.synth_afsk(n_ch, n, rate, seed=20260002); torch.cuda.synchronize()
for relaxed in (True, False):
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=relaxed); rx.set_kernel_timing(True)
    ms = []
    for r in range(8):
        rx.process_tensor(x); rx.sync(); ms.append(rx.last_kernel_ms()); ev = rx.poll_events_np()
    print(f"{rate} Hz relaxed={relaxed} [{rx.kernel_name()}]:", " ".join(f"{m:.3f}" for m in ms), f"| bursts {int((ev['kind']==3).sum())}", flush=True)
