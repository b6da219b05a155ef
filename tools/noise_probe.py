#!/usr/bin/env python3
"""Noisy synthetic channels: how often does a burst's transmitted payload differ from strict mode's, per mode?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
n_ch, secs, noise, seed = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), 3192
rate = 22050
n = int(rate * secs); n -= n % 420
x = sa.synth_afsk(n_ch, n, rate, seed=seed, noise_sigma=noise)
xc = x.t().contiguous()
def bursts(ev):
    out = [[] for _ in range(n_ch)]
    for r in ev[ev["kind"] == 3]:
        out[int(r["channel"])].append((int(r["sample_counter"]), r["bytes"][:min(int(r["len"]), 288)].tobytes()))
    return out
def run(label, env, kw, cm=False):
    for k in ("SAME_RELAXED", "SAME_TP_KERNEL"): os.environ.pop(k, None)
    os.environ.update(env)
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, **kw)
    rx.process_tensor(xc if cm else x, layout=sa.LAYOUT_CHANNEL_MAJOR if cm else sa.LAYOUT_TIME_MAJOR); rx.sync()
    return label, rx.kernel_name(), bursts(rx.poll_events_np())
ref = run("strict", {}, {})[2]
sps = rate / 520.83
def garbled(got):
    """bursts that carry a ZCZC / NNNN prefix (within 2 bit errors) but not the transmitted bytes"""
    bad = tot = 0
    for c in range(n_ch):
        pay = sa.synth_payload(seed, c)
        for t, b in got[c]:
            if t > n - 4 * sps * 8: continue
            tot += 1
            k = len(pay) if b[:2] == b"ZC" else 4
            want = pay[:k] if b[:2] == b"ZC" else b"NNNN"
            bad += b[:k] != want
    return bad, tot
print("strict: bursts whose transmitted bytes are wrong / all:", garbled(ref))
for label, env, kw, cm in (("relaxed one-wave", {}, {"relaxed": True}, False), ("TP strict chunks (cm)", {"SAME_RELAXED": "0"}, {"time_parallel": True}, True),
                           ("TP FASTMATH (cm)", {}, {"time_parallel": True}, True), ("TP FASTMATH (uniform)", {}, {"time_parallel": True}, False),
                           ("TP one-wave (cm)", {"SAME_TP_KERNEL": "wave"}, {"time_parallel": True}, True), ("TP one-wave (uniform)", {"SAME_TP_KERNEL": "wave"}, {"time_parallel": True}, False)):
    _, kn, got = run(label, env, kw, cm)
    nb = mism = unmatched = 0
    ex = []
    for c in range(n_ch):
        pay = sa.synth_payload(seed, c)
        a, b = ref[c], got[c]
        j = 0
        for (ta, ba) in a:
            if ta > n - 4 * sps * 8: continue
            nb += 1
            while j < len(b) and b[j][0] < ta - 4 * sps: j += 1
            if j < len(b) and abs(b[j][0] - ta) <= 4 * sps:
                k = len(pay) if ba[:4] == pay[:4] else 4
                if b[j][1][:k] != ba[:k]:
                    mism += 1
                    if len(ex) < 3: ex.append((c, ta, ba[:k], b[j][1][:k]))
                j += 1
            else:
                unmatched += 1
    print(f"{label:24s} [{kn}]: {nb} strict bursts, {mism} payload mismatches, {unmatched} unmatched; garbled vs transmitted {garbled(got)}", flush=True)
    for e in ex: print("     ", e)
