#!/usr/bin/env python3
"""The bench's timed region on the headline mode, pass by pass, with the library's harvest timing (SAME_DEBUG) between the marks:
where the wall time of the FIRST timed passes goes.   SAME_DEBUG=1 python tools/step_probe_passes.py [steps] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
C, T = 4096, 220500
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = sa.synth_afsk(C, T, 22050, seed=20260000).t().contiguous(); torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, time_parallel=True); rx.set_kernel_timing(True)
def consume():
    n = rx.pending_events()
    if n:
        rx.pack_bursts_np(0); rx.drop_events(n)
def mark(s): sys.stderr.write(f"---- {s}\n"); sys.stderr.flush()
for i in range(warmup):
    mark(f"warm-up pass {i}"); rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_CHANNEL_MAJOR, None); consume()
mark("drain"); rx.sync(); consume()
torch.cuda.synchronize()
t0 = time.perf_counter(); marks = []
for i in range(steps):
    mark(f"timed pass {i}"); a = time.perf_counter()
    rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_CHANNEL_MAJOR, None); b = time.perf_counter()
    consume(); c = time.perf_counter()
    sys.stderr.write(f"     process {1e3*(b-a):.2f} ms, consume {1e3*(c-b):.2f} ms, kernel {rx.last_kernel_ms():.3f}\n")
mark("drain"); rx.sync(); consume(); torch.cuda.synchronize()
print(f"{(time.perf_counter()-t0)/steps*1e3:.3f} ms/step over {steps}")
