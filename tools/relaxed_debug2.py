#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
name = "long_message"
pcm = np.fromfile(os.path.join("tests/golden", f"{name}.22050.s16le.bin"), dtype="<i2").astype(np.float32)
n_ch = 64
lead = [211 * c for c in range(n_ch)]
n = len(pcm) + max(lead)
x = np.zeros((n, n_ch), np.float32)
for c in range(n_ch):
    x[lead[c]:lead[c] + len(pcm), c] = pcm
xt = torch.from_numpy(x).cuda()
for rep in range(2):
  for label, kw in (("strict", {}), ("relaxed", {"relaxed": True}), ("relaxed+link_only", {"relaxed": True, "link_only": True})):
    rx = sa.SameReceiverBuilder(22050).samedec().build_batch(n_ch, **kw)
    rx.process_tensor(xt); rx.flush(); rx.sync()
    ev = rx.poll_events_np()
    nmsg = [int(((ev["channel"] == c) & (ev["kind"] == 18)).sum()) for c in range(n_ch)]
    nb = [int(((ev["channel"] == c) & (ev["kind"] == 3)).sum()) for c in range(n_ch)]
    print(label, rx.kernel_name(), "channels without a message:", [c for c in range(n_ch) if nmsg[c] == 0][:20], "bursts!=3:", [(c, nb[c]) for c in range(n_ch) if nb[c] != 3])
    if label == "relaxed":
        for c in [c for c in range(n_ch) if nmsg[c] == 0][:2]:
            e = ev[ev["channel"] == c]
            for r in e:
                print(f"   ch {c} t={int(r['sample_counter']):8d} kind={int(r['kind'])} len={int(r['len'])} sym={int(r['symbol_count'])} {r['bytes'][:min(int(r['len']),24)].tobytes()!r}")
