#!/usr/bin/env python3
"""Relaxed launches of large batches: the symbol-paced pipeline against the one- / two-wavefront relaxed kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa
rate = 22050
n = 44100 - 44100 % 1260
for n_ch in (49152, 65536, 131072, 262144):
    x = sa.synth_afsk(n_ch, n, rate, seed=20260000); torch.cuda.synchronize()
    for sym in ("1", "0"):
        os.environ["SAME_SYM"] = sym
        rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, relaxed=True); rx.set_kernel_timing(True)
        best = 1e9
        for r in range(3):
            if r: rx.reset()
            rx.process_tensor(x); rx.sync(); best = min(best, rx.last_kernel_ms()); ev = rx.poll_events_np()
        print(f"SAME_SYM={sym} {n_ch} ch x {n} [{rx.kernel_name()}]: best {best:.3f} ms = {4*n_ch*n/best/1e9/8*100:.2f} % of 8 TB/s; bursts {int((ev['kind']==3).sum())}", flush=True)
        del rx
    del x
