#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SAME_TP_KERNEL"] = sys.argv[1] if len(sys.argv) > 1 else "wave"
import numpy as np, torch
import sameold_amd as sa
rate, n_ch = 22050, 256
n = 22050 * 10; n -= n % 420
x = sa.synth_afsk(n_ch, 2 * n, rate, seed=4242)
rs = sa.SameReceiverBuilder(rate).build_batch(n_ch); rs.process_tensor(x); rs.sync(); ref = rs.poll_events_np()
rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, time_parallel=True)
rx.time_parallel_config(max_chunks=8)
for part in (x[:n], x[n:]):
    rx.process_tensor(part.contiguous())
rx.sync()
got = rx.poll_events_np()
print(rx.kernel_name(), rx.time_parallel_chunks())
bad = []
for c in range(n_ch):
    a = [(int(r["kind"]), r["bytes"][:int(r["len"])].tobytes()) for r in ref[(ref["channel"] == c) & (ref["kind"] >= 18)]]
    b = [(int(r["kind"]), r["bytes"][:int(r["len"])].tobytes()) for r in got[(got["channel"] == c) & (got["kind"] >= 18)]]
    if a != b: bad.append(c)
print("channels whose messages differ:", bad)
for c in bad[:2]:
    for label, ev in (("strict", ref), ("tp", got)):
        for r in ev[ev["channel"] == c]:
            print(f"   {label} ch {c} t={int(r['sample_counter']):8d} kind={int(r['kind'])} len={int(r['len'])} sym={int(r['symbol_count'])} {r['bytes'][:min(int(r['len']),30)].tobytes()!r}")
