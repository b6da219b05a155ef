#!/usr/bin/env python3
"""A/B of an environment knob on ONE box, the way bench.py steps the headline mode (process + consume with the burst gather),
variants alternating:   python tools/bench_headline_ab.py KNOB=VALUE [reps] [steps]"""
import os, subprocess, sys
knob = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3; steps = sys.argv[3] if len(sys.argv) > 3 else "40"
here = os.path.dirname(os.path.abspath(__file__))
for r in range(reps):
    for env in ({}, dict([knob.split("=", 1)])):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(here, "headline_steady.py"), steps], env=e, capture_output=True, text=True).stdout
        print(("base   " if not env else knob.ljust(7)) + " " + out.strip().split("\n")[-1], flush=True)
