#!/usr/bin/env python3
"""Round 4: the headline mode's step on the host.  Wall time per step against the kernel time over many steps, with the
library's own harvest split (SAME_DEBUG lines on stderr) and the consumer's share (burst packing).
    python tools/host_step_probe2.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
C, T = 4096, 220500
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
x = sa.synth_afsk(C, T, 22050, seed=20260000).t().contiguous(); torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, time_parallel=True); rx.set_kernel_timing(True)
acc = {"process": 0.0, "consume": 0.0}; n = 0; kms = []; walls = []
for k in range(steps + 5):
    if k == 5:
        rx.sync(); rx.drop_events(rx.pending_events()); torch.cuda.synchronize(); t_start = time.perf_counter()
    t0 = time.perf_counter(); rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_CHANNEL_MAJOR, None)
    t1 = time.perf_counter()
    n_ev = rx.pending_events()
    if n_ev:
        rec = rx.pack_bursts_np(0); rx.drop_events(n_ev)
    t2 = time.perf_counter()
    if k >= 5:
        acc["process"] += t1 - t0; acc["consume"] += t2 - t1; n += 1; walls.append(t2 - t0)
        kms.append(rx.last_kernel_ms())
rx.sync(); torch.cuda.synchronize()
wall = (time.perf_counter() - t_start) / n * 1e3
print({k: round(1e3 * v / n, 2) for k, v in acc.items()}, f"ms host per step; wall {wall:.2f} ms per step over {n} steps; kernel {np.mean(kms):.2f} ms; "
      f"per-step wall p50 {1e3*np.median(walls):.2f} p90 {1e3*np.percentile(walls, 90):.2f} max {1e3*np.max(walls):.2f}", flush=True)
