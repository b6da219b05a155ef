#!/usr/bin/env python3
"""Launch times of the headline mode after an idle GPU, with and without a busy GPU just before: is the slow start (2.1 -> 1.7 ms
over ~20 launches) the clocks?   python tools/ramp_probe.py [preheat_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
C, T = 4096, 220500
pre = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
x = sa.synth_afsk(C, T, 22050, seed=20260000).t().contiguous(); torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, time_parallel=True); rx.set_kernel_timing(True)
def run(n):
    k = []
    for i in range(n):
        rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_CHANNEL_MAJOR, None)
        m = rx.pending_events()
        if m: k.append(rx.last_demod_kernel_ms()); rx.pack_bursts_np(0); rx.drop_events(m)
    rx.sync(); m = rx.pending_events(); k.append(rx.last_demod_kernel_ms()); rx.drop_events(m)
    return k
run(3)
time.sleep(1.0)                      # the GPU idles, as between two modes of the bench
if pre > 0:
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < pre:
        (a @ a); torch.cuda.synchronize()
k = run(24)
print(f"preheat {pre:.0f} ms: demod alone per launch:", " ".join(f"{v:.3f}" for v in k))
