#!/usr/bin/env python3
"""Round 4: the symbol-paced pipeline (same_kernels_sym.hip) beside the 20-sample FASTMATH pipeline (SAME_SYM=0):
parity against strict mode and kernel time.   python tools/sym_probe.py [parity|time|tp|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sameold_amd as sa


def split(ev, n_ch):
    ev = ev[np.lexsort((np.arange(len(ev)), ev["channel"]))]        # (several calls: each harvest is ordered by channel)
    first = np.searchsorted(ev["channel"], np.arange(n_ch + 1))
    return [ev[first[c]:first[c + 1]] for c in range(n_ch)]


def bursts(e):
    b = e[e["kind"] == 3]
    return [(int(r["sample_counter"]), r["bytes"][: min(int(r["len"]), 288)].tobytes()) for r in b]


def parity(n_ch=256, secs=10.0, noise=0.0, seed=11, chunked=False):
    rate = 22050
    n = int(rate * secs); n -= n % 180
    x = sa.synth_afsk(n_ch, n, rate, seed=seed, noise_sigma=noise)
    out = {}
    for name, kw in (("strict", {}), ("relaxed", {"relaxed": True})):
        rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, **kw)
        if chunked and name == "relaxed":
            cuts = [0, 36 * 7, 36 * 7 + 5000, n // 3 + 17, n // 2, n]
            for a, b in zip(cuts[:-1], cuts[1:]):
                rx.process_tensor(x[a:b].contiguous())
        else:
            rx.process_tensor(x)
        rx.sync()
        out[name] = split(rx.poll_events_np(), n_ch)
        print(name, rx.kernel_name(), sum(len(e) for e in out[name]), "events", flush=True)
    nb = bad = 0; worst = 0; kinds_bad = 0
    for c in range(n_ch):
        pay = sa.synth_payload(seed, c)
        a, b = bursts(out["strict"][c]), bursts(out["relaxed"][c])
        nb += len(a)
        la = out["strict"][c]; lb = out["relaxed"][c]
        ka = la[la["kind"] <= 3]["kind"].tolist(); kb = lb[lb["kind"] <= 3]["kind"].tolist()
        if ka != kb:
            kinds_bad += 1
            if kinds_bad < 4: print("channel", c, "link kinds differ", ka[:24], kb[:24])
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
    print(f"parity ({'chunked calls' if chunked else 'one call'}, noise {noise}): {nb} bursts, {bad} differing, {kinds_bad} channels with other link-event kinds, "
          f"burst instants at most {worst} samples apart", flush=True)
    ma = [out["strict"][c][out["strict"][c]["kind"] >= 18]["bytes"].tobytes() for c in range(n_ch)]
    mb = [out["relaxed"][c][out["relaxed"][c]["kind"] >= 18]["bytes"].tobytes() for c in range(n_ch)]
    print("transport messages equal:", ma == mb, flush=True)


def timeit(n_ch, secs, reps=3, relaxed=True, tp=False, cm=False, chunks=0, rate=22050):
    n = int(rate * secs); n -= n % 360
    x = sa.synth_afsk(n_ch, n, rate, seed=20260000)
    if cm:
        x = x.t().contiguous()
    torch.cuda.synchronize()
    rx = sa.SameReceiverBuilder(rate).build_batch(n_ch, link_only=not os.environ.get("SYM_TRANSPORT"), relaxed=relaxed, time_parallel=tp)
    if tp and chunks:
        rx.time_parallel_config(max_chunks=chunks)
    rx.set_kernel_timing(True)
    best = 1e9
    for r in range(reps):
        if r: rx.reset()
        rx.process_tensor(x, layout=sa.LAYOUT_CHANNEL_MAJOR if cm else sa.LAYOUT_TIME_MAJOR); rx.sync()
        ms = rx.last_kernel_ms()
        try:
            d = rx.last_demod_kernel_ms()
        except Exception:
            d = ms
        if ms < best: best, dm = ms, d
        ev = rx.poll_events_np()
    print(f"SAME_SYM={os.environ.get('SAME_SYM','-')} {n_ch} ch x {n} tp={tp} cm={cm} chunks={rx.time_parallel_chunks()} [{rx.kernel_name()}]: best {best:.3f} ms "
          f"(demod alone {dm:.3f}) = {4*n_ch*n/best/1e9/8*100:.2f} % of 8 TB/s; bursts {int((ev['kind']==3).sum())}", flush=True)
    return rx


def prof(rx):
    import ctypes
    L = rx._L
    if not hasattr(L, "same_debug_profile_sym"):
        return
    out = (ctypes.c_ulonglong * 32)()
    if L.same_debug_profile_sym(out, 1) != 0:
        return
    v = list(out)
    steps = max(v[19], 1)
    names = ["S agc", "T dc", "A flt A+events", "E flt B+ted", "Y1 squelch+eq", "Y2 framer"]
    for r in range(6):
        print(f"  {names[r]:16s} work {v[3*r]/steps:8.1f}  barrier wait {v[3*r+1]/steps:8.1f}  feedback {v[3*r+2]/steps:8.1f}  clk/step")
    print(f"  workgroup 0: {v[18]} launches, {steps} steps, {sum(v[0:3])/steps:.0f} clk per step;", end="")
    print(f"  E passes {v[20]}; E's filter {v[22]/max(v[20],1):.0f} clk/pass, waiting for A's {v[24]/max(v[20],1):.0f}, timing+post {v[23]/max(v[20],1):.0f} clk/pass", flush=True)


def timeline(rx):
    """Profile builds and SAME_SYM_TL builds: when each role of the reporting group began a step, got past its first wait and
    published, over 12 steps -- and which of the roles it waits for arrived last (the step's critical graph)."""
    import ctypes
    L = rx._L
    if not hasattr(L, "same_debug_profile_sym_trace"):
        return
    out = (ctypes.c_ulonglong * 288)()
    if L.same_debug_profile_sym_trace(out) != 0:
        return
    v = np.array(list(out), dtype=np.int64).reshape(6, 12, 4)
    t0 = v[:, :, 0][v[:, :, 0] > 0].min()
    v = (v - t0) & 0xffffffff
    names = ["S", "T", "A", "E", "Y1", "Y2"]
    print("  step: role begin / first wait over / (A: filter done, E: second wait over) / published, clk from the first mark")
    for k in range(12):
        print(f"  {600 + k:4d}: " + "  ".join(f"{names[r]} {v[r,k,0]:6d}/{v[r,k,1]:6d}" + (f"/{v[r,k,3]:6d}" if r in (2, 3) else "") + f"/{v[r,k,2]:6d}" for r in range(6)))
    first = {0: (1, 2, 3, 4, 5), 1: (2, 3), 2: (0, 3), 3: (0,), 4: (3, 5), 5: (4,)}
    second = {1: (5,), 2: (5,), 3: (2, 4, 5), 5: (0, 1, 2, 3)}
    print("  per role, averaged over steps 601..611: period; begin -> first wait over; the role that arrived last at the first wait (how often) and")
    print("  how long after its publish mark the wait was over; first wait over -> publish")
    for r in range(6):
        per = np.diff(v[r, :, 2]).mean()
        w = (v[r, 1:, 1] - v[r, 1:, 0]).mean()
        work = (v[r, 1:, 2] - v[r, 1:, 1]).mean()
        last, lag = {}, []
        for k in range(1, 12):
            cand = [(v[d, k - 1, 2], d) for d in first[r]]
            t, d = max(cand)
            own = v[r, k, 0]
            who = names[d] if t > own else "self"
            last[who] = last.get(who, 0) + 1
            lag.append(v[r, k, 1] - max(t, own))
        extra = ""
        if r in second:
            lag2 = []
            for k in range(1, 12):
                t, d = max((v[d, k - 1, 2], d) for d in second[r])
                lag2.append((v[r, k, 3] if r == 3 else v[r, k, 2]) - t)
            extra = f"; second wait's last role published {np.mean(lag2):.0f} before " + ("the wait was over" if r == 3 else "this role published")
        print(f"    {names[r]:2s} period {per:6.0f}  begin->wait over {w:5.0f}  last: {last}  lag {np.mean(lag):5.0f}  wait over->publish {work:5.0f}{extra}")


def marks(rx):
    import ctypes
    L = rx._L
    if not hasattr(L, "same_debug_profile_sym_marks"):
        return
    out = (ctypes.c_ulonglong * 15)()
    if L.same_debug_profile_sym_marks(out, 1) != 0:
        return
    v = list(out)
    names = ["other (mailbox, barrier, idle)", "squelch", "equalizer step", "byte/framer", "events", "-"]
    tot = sum(v[:6]) or 1
    for n, x in zip(names, v[:6]):
        print(f"  Y {n:32s} {x:14d} clk  {100.0*x/tot:5.1f} %")


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what == "marks":
    os.environ["SAME_SYM"] = "1"
    rx = timeit(32768, 2.0, reps=2); marks(rx)
    rx = timeit(4096, 10.0, tp=True, cm=True, reps=2); marks(rx)
if what in ("parity", "all"):
    parity()
    parity(noise=0.05, seed=12)
    parity(chunked=True, seed=13)
if what in ("time", "all"):
    for sym in ("1", "0"):
        os.environ["SAME_SYM"] = sym
        rx = timeit(32768, 2.0)
        if sym == "1": prof(rx)
        timeit(4096, 2.0)
if what in ("tp", "all"):
    for sym in ("1", "0"):
        os.environ["SAME_SYM"] = sym
        rx = timeit(4096, 10.0, tp=True, cm=True, reps=4)
        if sym == "1": prof(rx)
        timeit(4096, 10.0, tp=True, cm=False, reps=3)
if what == "timeline":
    os.environ["SAME_SYM"] = "1"
    rx = timeit(32768, 2.0, reps=2); timeline(rx)
    rx = timeit(4096, 10.0, tp=True, cm=True, reps=2); timeline(rx)
if what == "timeline48":
    os.environ["SAME_SYM"] = "1"
    for r in (48000, 44100):
        rx = timeit(16384, 2.0, reps=2, rate=r); timeline(rx)
if what == "quick48":
    os.environ["SAME_SYM"] = "1"
    for r in (48000, 44100):
        timeit(16384, 2.0, reps=5, rate=r)
if what == "quick":
    os.environ["SAME_SYM"] = "1"
    timeit(32768, 2.0, reps=5)
    timeit(4096, 10.0, tp=True, cm=True, reps=5)
if what == "sweep":
    os.environ["SAME_SYM"] = "1"
    for cm in (True, False):
        for k in (0, 6, 7, 8, 9, 10, 11, 12, 14, 16):
            timeit(4096, 10.0, tp=True, cm=cm, reps=4, chunks=k)
