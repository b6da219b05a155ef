#!/usr/bin/env python3
"""Round 4: the headline launch (4 096 ch x 10 s, time-parallel, channel-major) back to back the way bench.py steps: kernel
time of every launch (whole launch | demodulation kernel alone) and the wall time per step.   python tools/headline_steady.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
C, T = 4096, 220500
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
x = sa.synth_afsk(C, T, 22050, seed=20260000).t().contiguous(); torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, time_parallel=True); rx.set_kernel_timing(True)
if os.environ.get("TP_CHUNKS"): rx.time_parallel_config(max_chunks=int(os.environ["TP_CHUNKS"]))
k, d = [], []
for i in range(steps + 5):
    if i == 5:
        rx.sync(); rx.drop_events(rx.pending_events()); torch.cuda.synchronize(); t0 = time.perf_counter()
    rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_CHANNEL_MAJOR, None)
    if i >= 5 and rx.pending_events():
        k.append(rx.last_kernel_ms()); d.append(rx.last_demod_kernel_ms())
    n = rx.pending_events()
    if n: rx.pack_bursts_np(0); rx.drop_events(n)
rx.sync(); wall = (time.perf_counter() - t0) / steps * 1e3
k, d = np.array(k), np.array(d)
print(f"pieces {rx.time_parallel_chunks()} SAME_TP_PLAN_STREAM={os.environ.get('SAME_TP_PLAN_STREAM','-')}: wall {wall:.3f} ms/step; launch mean {k.mean():.3f} min {k.min():.3f} max {k.max():.3f}; "
      f"demod alone mean {d.mean():.3f} min {d.min():.3f} p90 {np.percentile(d,90):.3f} max {d.max():.3f}", flush=True)
