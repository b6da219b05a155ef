#!/usr/bin/env python3
"""Back-to-back launches like bench.py (harvest of launch k overlapping launch k+1), printing the
kernel time and -- in a SAME_PROFILE build -- the shader-clock cycles per step of stage 3, to tell
a slower clock from more cycles.  usage: python tools/bimodal_probe.py [steps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
C, T = 4096, 220500
x = sa.synth_afsk(C, T, 22050, seed=20260000)
torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C)
rx.set_kernel_timing(True)
L = rx._L
ms = []
for k in range(steps):
    rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_TIME_MAJOR, None)
    ev = rx.poll_events_np()
    if len(ev):
        ms.append(rx.last_kernel_ms())
        if os.environ.get("PROBE_PACK"):
            from sameold_amd import distributed as sd
            sd.pack_burst_events(ev, 0)
rx.sync()
rx.poll_events_np()
print("kernel ms per launch:", " ".join(f"{m:.1f}" for m in ms))
if hasattr(L, "same_debug_profile_pipe"):
    buf = (ctypes.c_ulonglong * 9)()
    L.same_debug_profile_pipe(buf, 1)
    nstep = steps * (T // 18 + 2)
    tot = sum(buf[6:9]) / nstep
    print(f"stage 3: {tot:.0f} clk/step  -> at {sum(ms)/len(ms):.1f} ms per launch that is {tot * (T // 18 + 2) / (sum(ms)/len(ms)) / 1e6:.2f} GHz")
