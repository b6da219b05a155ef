#!/usr/bin/env python3
"""Round 6: the `scaled_big.relaxed` launch (131 072 channels x 2 s, the symbol-paced pipeline in four rounds of
workgroups) by itself -- its HIP-event kernel time launch by launch, back to back the way bench.py steps, with or without the
131 072-channel STRICT launches bench.py runs before it.  Run plain and under `rocprofv3 --kernel-trace` to hold the two clocks
against each other (profiles/README.md).      python tools/big_once.py [strict_first] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa
strict_first = len(sys.argv) > 1 and sys.argv[1] == "strict_first"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rate, n_ch = 22050, 131072
n = 2 * rate
x = sa.synth_afsk(n_ch, n, rate, seed=780); torch.cuda.synchronize()
if strict_first:
    rs = sa.SameReceiverBuilder(rate).build_batch(n_ch); rs.set_kernel_timing(True)
    ms = []
    rs.process_tensor(x); rs.sync(); rs.drop_events(rs.pending_events())      # (warm-up: the first launch also builds the kernel's code object)
    for _ in range(6):
        rs.process_tensor(x); ms.append(rs.last_kernel_ms()); rs.drop_events(rs.pending_events())
    rs.sync(); ms.append(rs.last_kernel_ms())
    print(f"strict  [{rs.kernel_name()}] HIP-event kernel ms per launch:", " ".join(f"{m:.3f}" for m in ms[1:]), flush=True)
    del rs
rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=True); rx.set_kernel_timing(True)
ms = []
rx.process_tensor(x); rx.sync(); rx.drop_events(rx.pending_events())
t0 = time.perf_counter()
for _ in range(launches):
    rx.process_tensor(x); ms.append(rx.last_kernel_ms()); rx.drop_events(rx.pending_events())
rx.sync(); ms.append(rx.last_kernel_ms())
dt = (time.perf_counter() - t0) / launches * 1e3
print(f"relaxed [{rx.kernel_name()}] HIP-event kernel ms per launch:", " ".join(f"{m:.3f}" for m in ms[1:]), f"| {dt:.3f} ms per step", flush=True)
