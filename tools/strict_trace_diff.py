#!/usr/bin/env python3
"""Round 5: the whole soft-symbol trace of the strict pipeline against the any-configuration kernel on the 16-channel reproduction
(tools/strict_bytes_repro.py): where do the two first differ, bit for bit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
from sameold_amd import montecarlo as mc
rate, seed, grid, n_samples = 22050, 2026, 15, 44096
lo = 15344
x = mc.synth_trials(16, lo, n_samples, rate, seed, 0.0, 1.0, grid)
tr = {}
for name, kw in (("pipe", {}), ("generic", {"generic_kernel": True})):
    rx = sa.SameReceiverBuilder(rate).build_batch(16, link_only=True, trace_symbols=True, **kw)
    rx.process_tensor(x); rx.sync()
    tr[name] = [rx.read_trace(c) for c in range(16)]
    print(name, rx.kernel_name())
for c in range(16):
    a, b = tr["pipe"][c], tr["generic"][c]
    n = min(len(a), len(b))
    same_t = a["sample_counter"][:n] == b["sample_counter"][:n]
    d = np.flatnonzero((a["zero"][:n].view(np.uint32) != b["zero"][:n].view(np.uint32)) | (a["sym"][:n].view(np.uint32) != b["sym"][:n].view(np.uint32)) | ~same_t)
    print(f"channel {c}: {len(a)} / {len(b)} symbols, {len(d)} differ" + (f"; first at symbol {d[0]} (sample {int(a['sample_counter'][d[0]])}): pipe {float(a['zero'][d[0]]).hex()} {float(a['sym'][d[0]]).hex()} generic {float(b['zero'][d[0]]).hex()} {float(b['sym'][d[0]]).hex()}; last at sample {int(a['sample_counter'][d[-1]])}" if len(d) else ""))
