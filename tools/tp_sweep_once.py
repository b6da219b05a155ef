#!/usr/bin/env python3
"""Round 5: the headline launch (4 096 channels x 10 s, channel-major, time-parallel) for a list of pieces-per-channel settings;
SAME_TP_SORT=1 packs pieces of similar length into the same workgroups.   python tools/tp_sweep_once.py 8 10 12 16"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa
rate, n_ch, n = 22050, 4096, 220500
x = sa.synth_afsk(n_ch, n, rate, seed=20260000).t().contiguous(); torch.cuda.synchronize()
for k in [int(a) for a in sys.argv[1:]] or [8]:
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, time_parallel=True)
    if k: rx.time_parallel_config(max_chunks=k)
    rx.set_kernel_timing(True)
    best = (1e9, 0)
    for r in range(4):
        if r: rx.reset()
        rx.process_tensor(x, layout=sa.LAYOUT_CHANNEL_MAJOR); rx.sync()
        best = min(best, (rx.last_kernel_ms(), rx.last_demod_kernel_ms())); ev = rx.poll_events_np()
    print(f"SAME_TP_SORT={os.environ.get('SAME_TP_SORT','-')} pieces {k} -> {rx.time_parallel_chunks()}: kernels {best[0]:.3f} ms, demod {best[1]:.3f} ms, bursts {int((ev['kind']==3).sum())}", flush=True)
