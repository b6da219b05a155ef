#!/usr/bin/env python3
"""Round 4: 131 072 channels x 2 s, relaxed, launches back to back the way bench.py's scaled_big block runs them: the one-wavefront
relaxed kernel (default) against the symbol-paced pipeline in rounds (SAME_SYM_MAX=262144), kernel time of every launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
rate, n_ch = 22050, int(sys.argv[1]) if len(sys.argv) > 1 else 131072
n = 44100
x = sa.synth_afsk(n_ch, n, rate, seed=780); torch.cuda.synchronize()
for sym_max in ("65536", "262144", "65536", "262144"):
    os.environ["SAME_SYM_MAX"] = sym_max
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=True); rx.set_kernel_timing(True)
    ms = []
    for k in range(9):
        rx.process_tensor(x)
        if k: ms.append(rx.last_kernel_ms())
        rx.drop_events(rx.pending_events())
    rx.sync(); ms.append(rx.last_kernel_ms())
    print(f"SAME_SYM_MAX={sym_max} {n_ch} ch [{rx.kernel_name()}]: " + " ".join(f"{m:.2f}" for m in ms) + f" | mean of the last 6: {np.mean(ms[-6:]):.2f} ms = {4*n_ch*n/np.mean(ms[-6:])/1e9/8*100:.1f} %", flush=True)
    del rx
