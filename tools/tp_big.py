import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sameold_amd as sa
rate, n_ch = 22050, 32768
n = 220500 - 220500 % 1260
x = sa.synth_afsk(n_ch, n, rate, seed=779); torch.cuda.synchronize()
ref = None
for cm in (False, True):
    xin = x.t().contiguous() if cm else x
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, time_parallel=True); rx.set_kernel_timing(True)
    import time
    ms = []
    for k in range(4):
        t0 = time.perf_counter()
        rx.process_tensor(xin, layout=sa.LAYOUT_CHANNEL_MAJOR if cm else sa.LAYOUT_TIME_MAJOR); rx.sync()
        ms.append((time.perf_counter() - t0) * 1e3)
        ev = rx.poll_events_np()
    print(f"32768 ch x {n} time_parallel cm={cm} [{rx.kernel_name()}] chunks={rx.time_parallel_chunks()}: wall per call {' '.join(f'{m:.1f}' for m in ms)} ms = {4*n_ch*n/min(ms)/1e6/8000*100:.1f} % of 8 TB/s; bursts {int((ev['kind']==3).sum())}", flush=True)
    del rx, xin
