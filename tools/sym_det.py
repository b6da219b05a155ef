#!/usr/bin/env python3
"""Determinism probe: the same relaxed call several times, and its int16 / channel-major forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
rate, n_ch, seed = 22050, 128, 901
x = sa.synth_afsk(n_ch, 22050 * 6, rate, seed=seed).round()
def run(kind, calls=None):
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, relaxed=True)
    keep = []
    off = 0
    for n in (calls or [x.shape[0]]):
        part = x[off:off + n]
        if kind == "i16":
            part = part.to(torch.int16).contiguous(); keep.append(part); rx.process_tensor(part)
        elif kind == "cm":
            part = part.t().contiguous(); keep.append(part); rx.process_tensor(part, layout=sa.LAYOUT_CHANNEL_MAJOR)
        else:
            rx.process_tensor(part.contiguous())
        off += n
    rx.sync()
    ev = rx.poll_events_np()
    return ev[np.lexsort((np.arange(len(ev)), ev["channel"]))], rx.kernel_name()
ref, k = run("f32")
print(k, len(ref))
for name, kind, calls in (("f32 again", "f32", None), ("f32 again", "f32", None), ("i16", "i16", None), ("cm", "cm", None), ("f32 two calls 65520", "f32", [65520, x.shape[0] - 65520]),
                          ("f32 two calls 36*100", "f32", [3600, x.shape[0] - 3600])):
    got, k2 = run(kind, calls)
    same = len(got) == len(ref) and np.array_equal(got["kind"], ref["kind"]) and np.array_equal(got["sample_counter"], ref["sample_counter"])
    nd = int(np.sum(got["sample_counter"] != ref["sample_counter"])) if len(got) == len(ref) else -1
    print(f"{name:24s} [{k2}] equal {same}; differing counters {nd}", flush=True)
    if not same and nd > 0:
        i = np.flatnonzero(got["sample_counter"] != ref["sample_counter"])[:5]
        print("   ", [(int(ref["channel"][j]), int(ref["kind"][j]), int(ref["sample_counter"][j]), int(got["sample_counter"][j])) for j in i])
