#!/usr/bin/env python3
"""Time-parallel mode against strict mode on the same input: what differs, and how long each takes.
    python tools/tp_probe.py CHANNELS SECONDS [MAX_CHUNKS] [NOISE] [RATE]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sameold_amd as sa  # noqa: E402


def by_channel(ev, n_ch):
    first = np.searchsorted(ev["channel"], np.arange(n_ch + 1))
    return [ev[first[c]:first[c + 1]] for c in range(n_ch)]


def main():
    C = int(sys.argv[1]); secs = float(sys.argv[2])
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    noise = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    rate = int(sys.argv[5]) if len(sys.argv) > 5 else 22050
    T = int(rate * secs)
    x = sa.synth_afsk(C, T, rate, seed=20260000, noise_sigma=noise)
    torch.cuda.synchronize()
    res = {}
    cm = bool(os.environ.get("TP_CM"))          # feed the time-parallel batch a channel-major copy of the input
    xc = x.t().contiguous() if cm else None
    for mode in ("strict", "tp"):
        rx = sa.SameReceiverBuilder(rate).build_batch(C, time_parallel=(mode == "tp"))
        if mode == "tp" and K:
            rx.time_parallel_config(max_chunks=K)
        rx.set_kernel_timing(True)
        ms = []
        for rep in range(3):
            if rep:
                rx.reset()
            t0 = time.perf_counter()
            if mode == "tp" and cm:
                rx.process_tensor(xc, layout=sa.LAYOUT_CHANNEL_MAJOR)
            else:
                rx.process_tensor(x)
            rx.sync()
            wall = time.perf_counter() - t0
            ev = rx.poll_events_np()
            ms.append((rx.last_kernel_ms(), wall * 1e3))
        res[mode] = ev
        print(f"{mode}: chunks {rx.time_parallel_chunks() if mode == 'tp' else 1}{' per-channel boundaries' if mode == 'tp' and rx.time_parallel_per_channel() else ''}  kernel ms / wall ms per rep: "
              + "  ".join(f"{a:.2f}/{b:.1f}" for a, b in ms) + f"  events {len(ev)}  [{rx.kernel_name()}]", flush=True)
    a, b = by_channel(res["strict"], C), by_channel(res["tp"], C)
    sps = rate / 520.83
    n_burst_bad = n_seq_bad = n_msg_bad = 0
    dt = {1: [], 2: [], 3: [], 0: []}
    shown = 0
    for c in range(C):
        sa_, sb_ = a[c], b[c]
        la, lb = sa_[sa_["kind"] <= 3], sb_[sb_["kind"] <= 3]
        ba = [(r["bytes"][: int(r["len"])].tobytes()) for r in la[la["kind"] == 3]]
        bb = [(r["bytes"][: int(r["len"])].tobytes()) for r in lb[lb["kind"] == 3]]
        ma = [(int(r["kind"]), r["bytes"][: int(r["len"])].tobytes()) for r in sa_[sa_["kind"] >= 18]]
        mb = [(int(r["kind"]), r["bytes"][: int(r["len"])].tobytes()) for r in sb_[sb_["kind"] >= 18]]
        bad = False
        if ba != bb:
            n_burst_bad += 1; bad = True
        if ma != mb:
            n_msg_bad += 1; bad = True
        if list(la["kind"]) != list(lb["kind"]):
            n_seq_bad += 1; bad = True
        else:
            d = lb["sample_counter"].astype(np.int64) - la["sample_counter"].astype(np.int64)
            for k in (0, 1, 2, 3):
                dt[k] += list(d[la["kind"] == k])
        if len(ba) != len(bb) or ma != mb:
            print(f"=== channel {c}: {len(ba)} vs {len(bb)} bursts, messages {'==' if ma == mb else '!='}")
            print("  strict:", [(int(r['kind']), int(r['sample_counter']), r['bytes'][:min(int(r['len']), 12)].tobytes()) for r in sa_])
            print("  tp    :", [(int(r['kind']), int(r['sample_counter']), r['bytes'][:min(int(r['len']), 12)].tobytes()) for r in sb_])
        if (ma != mb or list(la["kind"]) != list(lb["kind"])) and shown < 3:
            shown += 1
            print(f"--- channel {c}: bursts {'==' if ba == bb else '!='}, messages {'==' if ma == mb else '!='}")
            print("  strict:", [(int(r['kind']), int(r['sample_counter'])) for r in sa_][:40])
            print("  tp    :", [(int(r['kind']), int(r['sample_counter'])) for r in sb_][:40])
    # what differs inside bursts: only bytes after the transmitted payload (decoded from the silence that
    # follows the carrier), or payload bytes too?
    tail_only = payload_bad = count_bad = 0
    for c in range(C):
        la, lb = a[c][a[c]["kind"] == 3], b[c][b[c]["kind"] == 3]
        if len(la) != len(lb):
            count_bad += 1
            continue
        pay = sa.synth_payload(20260000, c)
        for ra, rb in zip(la, lb):
            xa, xb = ra["bytes"][: int(ra["len"])].tobytes(), rb["bytes"][: int(rb["len"])].tobytes()
            if xa == xb:
                continue
            n = len(pay) if xa.startswith(pay[:4]) else 4      # header burst or NNNN burst
            if xa[:n] == xb[:n]:
                tail_only += 1
            else:
                payload_bad += 1
                if payload_bad <= 5:
                    print(f"   payload differs, channel {c} @ {int(ra['sample_counter'])}/{int(rb['sample_counter'])}: {xa!r} vs {xb!r}")
    print(f"burst differences: {tail_only} only after the payload, {payload_bad} inside it, {count_bad} channels with a different number of bursts")
    print(f"channels {C}: burst lists differ on {n_burst_bad}, message lists on {n_msg_bad}, link kind sequences on {n_seq_bad}")
    for k, name in ((1, "searching"), (2, "reading"), (3, "burst"), (0, "no_carrier")):
        d = np.array(dt[k])
        if len(d):
            print(f"  dt {name:10s}: n {len(d)}  exact {np.mean(d == 0):.4f}  |dt|<=2 {np.mean(np.abs(d) <= 2):.4f}  "
                  f"max |dt| {np.abs(d).max()} samples ({np.abs(d).max() / sps:.2f} symbols)  p99 {np.percentile(np.abs(d), 99):.0f}")
    # transport event times
    ta = res["strict"][res["strict"]["kind"] >= 16]; tb = res["tp"][res["tp"]["kind"] >= 16]
    if len(ta) == len(tb) and np.array_equal(ta["kind"], tb["kind"]) and np.array_equal(ta["channel"], tb["channel"]):
        d = tb["sample_counter"].astype(np.int64) - ta["sample_counter"].astype(np.int64)
        print(f"  transport events: {len(ta)} same kinds; max |dt| {np.abs(d).max() if len(d) else 0} samples, exact {np.mean(d == 0) if len(d) else 1:.4f}")
    else:
        print(f"  transport events: strict {len(ta)} vs tp {len(tb)} (kinds/channels differ)")


if __name__ == "__main__":
    main()
