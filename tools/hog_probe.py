#!/usr/bin/env python3
"""Does host memory traffic slow the demodulation kernel?  Runs the same launch repeatedly
(link-only, no harvest work in between), with N background threads streaming through host
memory.  usage: python tools/hog_probe.py <hog_threads> [channels] [seconds]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sameold_amd as sa

hogs = int(sys.argv[1]) if len(sys.argv) > 1 else 0
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(22050 * float(sys.argv[3])) if len(sys.argv) > 3 else 220500
x = sa.synth_afsk(C, T, 22050, seed=20260000)
torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, link_only=True)
rx.set_kernel_timing(True)
stop = False
def hog():
    a = np.ones(64 << 20, dtype=np.uint8); b = np.empty_like(a)
    while not stop:
        np.copyto(b, a)
ths = [threading.Thread(target=hog) for _ in range(hogs)]
for t in ths: t.start()
time.sleep(0.3)
ms = []
for k in range(8):
    rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_TIME_MAJOR, None)
    rx.sync()
    ms.append(rx.last_kernel_ms())
    rx.poll_events_np()
stop = True
for t in ths: t.join()
print(f"{hogs} host threads streaming memory: kernel ms", " ".join(f"{m:.1f}" for m in ms), f"[{rx.kernel_name()}]")
