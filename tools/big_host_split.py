#!/usr/bin/env python3
"""Round 5: where a step of bench.py's scaled_big block (131 072 channels x 2 s, relaxed, transport layer on) spends its host time:
wall per step against the kernel, the library's harvest split (SAME_DEBUG=1 lines on stderr) and the consumer's share.
    SAME_DEBUG=1 python tools/big_host_split.py [channels] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
rate, n_ch = 22050, int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 44100
x = sa.synth_afsk(n_ch, n, rate, seed=780); torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=True); rx.set_kernel_timing(True)
buf = np.empty((n_ch * 2, 304), dtype=np.uint8) if os.environ.get('REUSE', '1') == '1' else None
acc = {"process": 0.0, "consume": 0.0}; kms = []
for k in range(steps + 2):
    if k == 2:
        rx.sync(); rx.drop_events(rx.pending_events()); torch.cuda.synchronize(); t_start = time.perf_counter()
    t0 = time.perf_counter(); rx.process_tensor(x); t1 = time.perf_counter()
    n_ev = rx.pending_events()
    if n_ev:
        rec = rx.pack_bursts_np(0, out=buf); rx.drop_events(n_ev)
    t2 = time.perf_counter()
    if k >= 2:
        acc["process"] += t1 - t0; acc["consume"] += t2 - t1; kms.append(rx.last_kernel_ms())
rx.sync(); torch.cuda.synchronize()
wall = (time.perf_counter() - t_start) / steps * 1e3
print({k: round(1e3 * v / steps, 2) for k, v in acc.items()}, f"ms host per step; wall {wall:.2f} ms per step over {steps} steps; kernel {np.mean(kms):.2f} ms [{rx.kernel_name()}]", flush=True)
