#!/usr/bin/env python3
"""Round 5 debugging aid: one batch of configs[4]'s AWGN trials (65 536 channels) in strict mode against the oracle on one slab of
channels, burst bytes included; SAME_PIPE=0/1 picks the kernel.   python tools/strict_bytes_debug.py [first_channel] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import sameold_amd as sa
from sameold_amd import montecarlo as mc
from oracle import binding as ob
from helpers.oracle_compare import oracle_link_events
c0 = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
nslab = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
trials, rate, seed, grid = 65536, 22050, 2026, 15
n_samples = 44096
x = mc.synth_trials(trials, 0, n_samples, rate, seed, 0.0, 1.0, grid)
for rep in range(2):
    rx = sa.SameReceiverBuilder(rate).build_batch(trials, link_only=True)
    rx.process_tensor(x); rx.sync()
    ev = rx.poll_events_np()
    link = ev[ev["kind"] <= 3]
    first = np.searchsorted(link["channel"], np.arange(trials + 1))
    mine = link[first[c0]:first[c0 + nslab]]
    ref = oracle_link_events(ob, ob.default_config(rate), x[:, c0:c0 + nslab].contiguous().cpu().numpy())
    print(f"rep {rep} kernel {rx.kernel_name()}: {len(mine)} device events, oracle {len(ref)}")
    if len(mine) == len(ref):
        b = np.flatnonzero(mine["kind"] == 3)
        ln = np.minimum(mine["len"][b], 288)[:, None]
        diff = (mine["bytes"][b] != ref["bytes"][b]) & (np.arange(288)[None, :] < ln)
        for i in np.flatnonzero(diff.any(axis=1)):
            r, o = mine[b[i]], ref[b[i]]
            cols = np.flatnonzero(diff[i])
            print(f"  channel {int(r['channel'])} burst at {int(r['sample_counter'])} len {int(r['len'])}: bytes differ at {cols.tolist()[:20]}")
            print("    device:", bytes(r["bytes"][: int(r["len"])]))
            print("    oracle:", bytes(o["bytes"][: int(o["len"])]))

# the slab by itself, as batches of several sizes around the channel that differed (is it the kernel, or the launch in rounds?)
for lo, hi in ((12288, 16384), (15296, 15360), (15354 - 15354 % 16, 15354 - 15354 % 16 + 16), (0, 32768), (0, 49152)):
    xs = x[:, lo:hi].contiguous()
    rx = sa.SameReceiverBuilder(rate).build_batch(hi - lo, link_only=True)
    rx.process_tensor(xs); rx.sync()
    ev = rx.poll_events_np()
    e = ev[(ev["channel"] == 15354 - lo) & (ev["kind"] == 3)]
    print(f"batch of channels {lo}..{hi} [{rx.kernel_name()}]:", [bytes(r["bytes"][78:83]) for r in e])
