#!/usr/bin/env python3
"""Debug: the symbol-paced kernel over several calls (state carried) against one call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sameold_amd as sa

def split(ev, n_ch):
    ev = ev[np.lexsort((np.arange(len(ev)), ev["channel"]))]        # (several calls: each harvest is ordered by channel)
    first = np.searchsorted(ev["channel"], np.arange(n_ch + 1))
    return [ev[first[c]:first[c + 1]] for c in range(n_ch)]

rate, n_ch, seed = 22050, 64, 13
n = 22050 * 4; n -= n % 180
x = sa.synth_afsk(n_ch, n, rate, seed=seed)
def run(cuts, **kw):
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, **kw)
    for a, b in zip(cuts[:-1], cuts[1:]):
        rx.process_tensor(x[a:b].contiguous())
    rx.sync()
    return split(rx.poll_events_np(), n_ch), rx.kernel_name()
ref, k0 = run([0, n], relaxed=True)
print("one call:", k0, sum(len(e) for e in ref))
for name, cuts in (("two calls, whole blocks", [0, 36 * 100, n]), ("three calls, whole blocks", [0, 36 * 7, 36 * 500, n]),
                   ("tail of 5", [0, 36 * 100 + 5, n]), ("tail of 35", [0, 36 * 100 + 35, n]), ("short first call", [0, 36, n]), ("two blocks", [0, 72, n])):
    got, k = run(cuts, relaxed=True)
    bad = [c for c in range(n_ch) if not (np.array_equal(got[c]["kind"], ref[c]["kind"]) and np.array_equal(got[c]["sample_counter"], ref[c]["sample_counter"]))]
    print(f"{name:28s} [{k}] cuts {cuts}: {len(bad)} of {n_ch} channels differ from one call", flush=True)
    for c in bad[:2]:
        print("   ch", c, "ref", list(zip(ref[c]["kind"].tolist(), ref[c]["sample_counter"].tolist()))[:8])
        print("   ch", c, "got", list(zip(got[c]["kind"].tolist(), got[c]["sample_counter"].tolist()))[:8])

print("---- only the first 3600 samples, by mode")
def run1(kw, env):
    for k in ("SAME_SYM",): os.environ.pop(k, None)
    os.environ.update(env)
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, **kw)
    rx.process_tensor(x[:3600].contiguous()); rx.sync()
    ev = rx.poll_events_np()
    print(kw, env, rx.kernel_name(), [(int(e["channel"]), int(e["kind"]), int(e["sample_counter"])) for e in ev][:12], flush=True)
run1({}, {})
run1({"relaxed": True}, {"SAME_SYM": "0"})
run1({"relaxed": True}, {"SAME_SYM": "1"})
print("---- two calls by mode")
for kw, env in (({}, {}), ({"relaxed": True}, {"SAME_SYM": "0"}), ({"relaxed": True}, {"SAME_SYM": "1"})):
    for k in ("SAME_SYM",): os.environ.pop(k, None)
    os.environ.update(env)
    one, k1 = run([0, n], **kw)
    two, k2 = run([0, 36 * 100 * 5, n], **kw)
    bad = [c for c in range(n_ch) if not (np.array_equal(one[c]["kind"], two[c]["kind"]) and np.array_equal(one[c]["sample_counter"], two[c]["sample_counter"]))]
    print(kw, env, k1, k2, "channels differing between one call and two:", len(bad), flush=True)
    for c in bad[:2]:
        print("   ch", c, "one", list(zip(one[c]["kind"].tolist(), one[c]["sample_counter"].tolist()))[:8])
        print("   ch", c, "two", list(zip(two[c]["kind"].tolist(), two[c]["sample_counter"].tolist()))[:8])
