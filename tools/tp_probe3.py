#!/usr/bin/env python3
"""Round 3: time-parallel launch of configs[1] with the kernel choices side by side.  python tools/tp_probe3.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sameold_amd as sa

def run(label, env, chunks=0, cm=True, n_ch=4096, secs=10.0, reps=4):
    for k in ("SAME_RELAXED", "SAME_TP_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rate = 22050
    n = int(rate * secs); n -= n % 420
    x = sa.synth_afsk(n_ch, n, rate, seed=20260000)
    if cm: x = x.t().contiguous()
    torch.cuda.synchronize()
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, time_parallel=True)
    if chunks: rx.time_parallel_config(max_chunks=chunks)
    rx.set_kernel_timing(True)
    best = 1e9
    for r in range(reps):
        if r: rx.reset()
        rx.process_tensor(x, layout=sa.LAYOUT_CHANNEL_MAJOR if cm else sa.LAYOUT_TIME_MAJOR); rx.sync()
        ms = rx.last_kernel_ms(); best = min(best, ms)
        ev = rx.poll_events_np()
    print(f"{label:34s} {n_ch} ch x {n} cm={cm} chunks={rx.time_parallel_chunks()} [{rx.kernel_name()}]: best {best:.3f} ms = {4*n_ch*n/best/1e9/8*100:.2f} % of 8 TB/s; bursts {int((ev['kind']==3).sum())}", flush=True)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
run("pipeline strict chunks", {"SAME_RELAXED": "0"}, reps=reps)
run("pipeline FASTMATH", {}, reps=reps)
run("pipeline FASTMATH, 12 chunks", {}, chunks=12, reps=reps)
run("pipeline FASTMATH, 16 chunks", {}, chunks=16, reps=reps)
run("one-wave relaxed kernel", {"SAME_TP_KERNEL": "wave"}, reps=reps)
run("pipeline FASTMATH time-major", {}, cm=False, reps=reps)
run("pipeline strict time-major", {"SAME_RELAXED": "0"}, cm=False, reps=reps)
