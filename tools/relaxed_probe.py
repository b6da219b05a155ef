#!/usr/bin/env python3
"""Relaxed-arithmetic kernel: a quick look at parity against strict mode and at kernel time in the three regimes.
    python tools/relaxed_probe.py [parity|time|tp] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import sameold_amd as sa


def split(ev, n_ch):
    first = np.searchsorted(ev["channel"], np.arange(n_ch + 1))
    return [ev[first[c]:first[c + 1]] for c in range(n_ch)]


def bursts(e):
    b = e[e["kind"] == 3]
    return [(int(r["sample_counter"]), r["bytes"][: min(int(r["len"]), 288)].tobytes()) for r in b]


def parity(n_ch=256, secs=10.0, noise=0.0, seed=11):
    rate = 22050
    n = int(rate * secs); n -= n % 42
    x = sa.synth_afsk(n_ch, n, rate, seed=seed, noise_sigma=noise)
    out = {}
    for name, kw in (("strict", {}), ("relaxed", {"relaxed": True})):
        rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, **kw)
        rx.process_tensor(x); rx.sync()
        out[name] = split(rx.poll_events_np(), n_ch)
        print(name, rx.kernel_name(), sum(len(e) for e in out[name]), "events")
    nb = bad = 0; worst = 0
    for c in range(n_ch):
        pay = sa.synth_payload(seed, c)
        a, b = bursts(out["strict"][c]), bursts(out["relaxed"][c])
        nb += len(a)
        if len(a) != len(b):
            bad += 1
            if bad < 5: print("channel", c, "bursts", len(a), len(b))
            continue
        for (ta, ba), (tb, bb) in zip(a, b):
            k = len(pay) if ba[:4] == pay[:4] else 4
            worst = max(worst, abs(ta - tb))
            if ba[:k] != bb[:k]:
                bad += 1
                if bad < 5: print("channel", c, ba[:k], bb[:k])
    print(f"parity: {nb} bursts, {bad} differing, burst event instants at most {worst} samples apart")
    ma = [m for c in range(n_ch) for m in out["strict"][c][out["strict"][c]["kind"] >= 18]["bytes"].tobytes()]
    mb = [m for c in range(n_ch) for m in out["relaxed"][c][out["relaxed"][c]["kind"] >= 18]["bytes"].tobytes()]
    print("transport messages equal:", ma == mb)


def timeit(n_ch, secs, reps=3, relaxed=True, tp=False, cm=False, chunks=0):
    rate = 22050
    n = int(rate * secs); n -= n % 420
    x = sa.synth_afsk(n_ch, n, rate, seed=20260000)
    if cm:
        x = x.t().contiguous()
    torch.cuda.synchronize()
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=True, relaxed=relaxed, time_parallel=tp)
    if tp and chunks:
        rx.time_parallel_config(max_chunks=chunks)
    rx.set_kernel_timing(True)
    for r in range(reps):
        if r: rx.reset()
        t0 = time.perf_counter()
        rx.process_tensor(x, layout=sa.LAYOUT_CHANNEL_MAJOR if cm else sa.LAYOUT_TIME_MAJOR); rx.sync()
        dt = time.perf_counter() - t0
        ms = rx.last_kernel_ms()
        ev = rx.poll_events_np()
        print(f"{n_ch} ch x {n} [{rx.kernel_name()} tp={tp} cm={cm} chunks={rx.time_parallel_chunks()}]: kernel {ms:.3f} ms = {n_ch*n/ms/1e6:.1f} Gsample/s "
              f"= {4*n_ch*n/ms/1e9/8*100:.2f} % of 8 TB/s; wall {dt*1e3:.1f} ms; bursts {int((ev['kind']==3).sum())}", flush=True)
    profile_report(rx, reps)


def profile_report(rx, reps):
    if not hasattr(rx._L, "same_debug_profile_relaxed"):
        return
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    rx._L.same_debug_profile_relaxed(buf, 1)
    names = ["DC blocker (+ input wait)", "AGC + window push", "matched filters", "timing loop + symbol path", "replay / loop ends (duo: B posting + barrier wait)", "hand-over check (duo: A barrier wait + replay)"]
    nsb = max(int(buf[6]), 1)
    tot = sum(buf[:6])
    for n, v in zip(names, buf[:6]):
        print(f"  {n:28s} {v / nsb:9.1f} clk per sub-block  {100.0 * v / max(tot, 1):5.1f} %")
    print(f"  total {tot / nsb:.1f} clk per sub-block of 21 samples ({int(buf[6])} sub-blocks, {int(buf[7])} TED passes of wavefront 0)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if what == "parity":
        parity()
        parity(noise=0.05, seed=12)
    elif what == "time":
        for relaxed in (True, False):
            timeit(4096, 10.0, relaxed=relaxed)
            timeit(32768, 2.0, relaxed=relaxed)
            timeit(65536, 2.0, relaxed=relaxed)
            timeit(131072, 2.0, relaxed=relaxed)
    elif what == "prof":
        timeit(int(sys.argv[2]), float(sys.argv[3]), reps=2)
    elif what == "tp":
        for ch in (8, 16):
            timeit(4096, 10.0, tp=True, cm=True, chunks=ch)
        timeit(4096, 10.0, tp=True, cm=False)
        os.environ["SAME_RELAXED"] = "0"
        timeit(4096, 10.0, tp=True, cm=True, relaxed=False)
