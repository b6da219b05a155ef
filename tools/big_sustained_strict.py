#!/usr/bin/env python3
"""Round 4: large strict batches, launches back to back with the transport layer on: the one-wavefront kernel (default beyond
32 768 channels) against the wavefront pipeline in rounds (SAME_PIPE=1), kernel time of every launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
rate, n_ch = 22050, int(sys.argv[1]) if len(sys.argv) > 1 else 131072
n = 44100
x = sa.synth_afsk(n_ch, n, rate, seed=780); torch.cuda.synchronize()
for pipe in ("", "1", "", "1"):
    if pipe: os.environ["SAME_PIPE"] = pipe
    else: os.environ.pop("SAME_PIPE", None)
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch); rx.set_kernel_timing(True)
    ms = []
    for k in range(9):
        rx.process_tensor(x)
        if k: ms.append(rx.last_kernel_ms())
        rx.drop_events(rx.pending_events())
    rx.sync(); ms.append(rx.last_kernel_ms())
    print(f"SAME_PIPE={pipe or '-'} {n_ch} ch [{rx.kernel_name()}]: " + " ".join(f"{m:.2f}" for m in ms) + f" | mean of the last 6: {np.mean(ms[-6:]):.2f} ms = {4*n_ch*n/np.mean(ms[-6:])/1e9/8*100:.1f} %", flush=True)
    del rx
