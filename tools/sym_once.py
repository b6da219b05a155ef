#!/usr/bin/env python3
"""One relaxed launch per configuration, for rocprofv3 counter passes: python tools/sym_once.py [scaled|tp] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa
what = sys.argv[1] if len(sys.argv) > 1 else "scaled"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rate = 22050
if what == "scaled":
    n_ch, n = 32768, 44100 - 44100 % 180
    x = sa.synth_afsk(n_ch, n, rate, seed=20260000)
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, relaxed=True)
    lay = sa.LAYOUT_TIME_MAJOR
else:
    n_ch, n = 4096, 220500
    x = sa.synth_afsk(n_ch, n, rate, seed=20260000).t().contiguous()
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, time_parallel=True)
    lay = sa.LAYOUT_CHANNEL_MAJOR
torch.cuda.synchronize()
for r in range(reps):
    if r: rx.reset()
    rx.process_tensor(x, layout=lay); rx.sync()
    ev = rx.poll_events_np()
print(what, rx.kernel_name(), n_ch, n, len(ev), "events", flush=True)
