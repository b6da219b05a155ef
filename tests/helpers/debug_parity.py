#!/usr/bin/env python3
"""Compare GPU events with the oracle on the bench workload and print the first differences.
(Test tooling: it lives under tests/ because it runs the oracle.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sameold_amd as sa
from oracle import binding as ob

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
nchk = int(sys.argv[3]) if len(sys.argv) > 3 else 16
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 20260000
generic = len(sys.argv) > 5 and sys.argv[5] == "generic"
link_only = len(sys.argv) > 6 and sys.argv[6] == "link"
T = int(22050 * secs)
x = sa.synth_afsk(C, T, 22050, seed=seed)
rx = sa.SameReceiverBuilder(22050).build_batch(C, generic_kernel=generic, link_only=link_only)
rx.process_tensor(x); rx.sync()
ev = rx.poll_events_np()
xs = x[:, :nchk].contiguous().cpu().numpy()
bad = 0
for c in range(nchk):
    mine = ev[ev["channel"] == c]
    got = [(int(r["kind"]), int(r["sample_counter"]), r["bytes"][: min(int(r["len"]), 288)].tobytes()) for r in mine]
    ref = [e.as_tuple() for e in ob.Receiver(ob.default_config(22050), link_only=link_only).run(np.ascontiguousarray(xs[:, c]))]
    if got != ref:
        bad += 1
        print(f"channel {c}: {len(got)} vs {len(ref)} events")
        for i, (g, r) in enumerate(zip(got, ref)):
            if g != r:
                print("  first diff at", i, "\n   gpu", g, "\n   ref", r)
                for j in range(max(0, i - 2), min(len(got), i + 3)):
                    print("     gpu", got[j][:2], " ref", ref[j][:2] if j < len(ref) else None)
                break
        else:
            print("  prefix equal; tails:", got[len(ref):][:3], ref[len(got):][:3])
        if bad >= 3:
            break
print("kernel", rx.kernel_name(), "bad channels:", bad, "of", nchk)
