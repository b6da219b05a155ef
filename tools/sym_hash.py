#!/usr/bin/env python3
"""Round 6: a digest of everything a relaxed launch delivers (kind, channel, sample counter, length, bytes of every event) on fixed
workloads at the three rates -- a change of the symbol-paced kernel that is meant to keep its arithmetic (fewer instructions, other
scheduling) must leave every digest as it was.   python tools/sym_hash.py [quick]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sameold_amd as sa

def digest(ev):
    ev = ev[np.lexsort((ev["sample_counter"], ev["channel"]))]
    h = hashlib.md5()
    for f in ("kind", "channel", "sample_counter", "len"):
        h.update(np.ascontiguousarray(ev[f]).tobytes())
    b = ev[ev["kind"] == 3]
    cols = np.arange(b["bytes"].shape[1])[None, :] < np.minimum(b["len"], b["bytes"].shape[1])[:, None]
    h.update(np.ascontiguousarray(np.where(cols, b["bytes"], 0)).tobytes())
    return h.hexdigest()[:16], len(ev), int((ev["kind"] == 3).sum())

cases = [(22050, 2048, 6.0, 0.0, False), (22050, 1024, 6.0, 0.05, False), (48000, 512, 4.0, 0.02, False), (44100, 256, 4.0, 0.0, False),
         (22050, 512, 10.0, 0.0, True)]
for rate, n_ch, secs, noise, tp in cases:
    n = int(rate * secs); n -= n % 360
    x = sa.synth_afsk(n_ch, n, rate, seed=4242 + n_ch, noise_sigma=noise)
    if tp:
        x = x.t().contiguous()
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=not tp, time_parallel=tp)
    rx.process_tensor(x, layout=sa.LAYOUT_CHANNEL_MAJOR if tp else sa.LAYOUT_TIME_MAJOR); rx.sync()
    d = digest(rx.poll_events_np())
    print(f"{rate} Hz {n_ch} ch x {n} noise {noise} tp={tp} [{rx.kernel_name()}]: {d[0]}  events {d[1]} bursts {d[2]}", flush=True)
