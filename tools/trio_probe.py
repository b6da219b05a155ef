#!/usr/bin/env python3
"""trio relaxed kernel: parity smoke + timings against duo / solo / pipeline FASTMATH."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
from relaxed_probe import timeit
from duo_probe_lib import parity
for k in ("trio",):
    parity(k); parity(k, noise=0.05, seed=12)
for k in ("trio", "duo"):
    os.environ["SAME_RELAXED_KERNEL"] = k
    print("==", k, flush=True)
    for ch, secs in ((4096, 10.0), (32768, 2.0)):
        timeit(ch, secs, reps=2)
os.environ["SAME_TP_KERNEL"] = "wave"
for k in ("trio", "duo"):
    os.environ["SAME_RELAXED_KERNEL"] = k
    print("== time-parallel (channel-major) on", k, flush=True)
    timeit(4096, 10.0, tp=True, cm=True, chunks=8, reps=3)
os.environ.pop("SAME_TP_KERNEL"); os.environ.pop("SAME_RELAXED_KERNEL")
print("== time-parallel on the pipeline FASTMATH", flush=True)
timeit(4096, 10.0, tp=True, cm=True, chunks=8, reps=3)
