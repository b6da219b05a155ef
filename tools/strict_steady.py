#!/usr/bin/env python3
"""Round 4: configs[1] in strict mode, launches back to back: the workgroup widths of the wavefront pipeline (SAME_PIPE_LANES)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
C, T = 4096, 220500
x = sa.synth_afsk(C, T, 22050, seed=20260000); torch.cuda.synchronize()
for lanes in ("", "16", "32", "64"):
    if lanes: os.environ["SAME_PIPE_LANES"] = lanes
    else: os.environ.pop("SAME_PIPE_LANES", None)
    rx = sa.SameReceiverBuilder(22050).build_batch(C); rx.set_kernel_timing(True)
    ms = []
    for k in range(8):
        rx.process_tensor(x)
        if k: ms.append(rx.last_kernel_ms())
        rx.drop_events(rx.pending_events())
    rx.sync(); ms.append(rx.last_kernel_ms())
    print(f"SAME_PIPE_LANES={lanes or '-'} [{rx.kernel_name()}]: " + " ".join(f"{m:.2f}" for m in ms) + f" | mean of the last 5: {np.mean(ms[-5:]):.2f} ms", flush=True)
    del rx
