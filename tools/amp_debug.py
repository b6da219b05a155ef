import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
rate, n_ch, seed = 22050, 128, 4100
n = 22050 * 6
names = {0: "NoCarrier", 1: "Searching", 2: "Reading", 3: "Burst"}
for amplitude, limits, ch in ((300.0, "samedec", 81),):
    x = sa.synth_afsk(n_ch, n, rate, seed=seed) * (amplitude / 30000.0)
    mk = lambda: (sa.SameReceiverBuilder(rate).samedec() if limits == "samedec" else sa.SameReceiverBuilder(rate))
    for relaxed in (False, True):
        rx = mk().build_batch(n_ch, relaxed=relaxed, link_only=True, trace_symbols=True)
        rx.process_tensor(x); rx.sync()
        ev = rx.poll_events_np()
        e = ev[ev["channel"] == ch]
        print(f"amp {amplitude} {limits} ch {ch} relaxed={relaxed} [{rx.kernel_name()}]:", [(names.get(int(k), int(k)), int(t)) for k, t in zip(e["kind"], e["sample_counter"])])
        tr = rx.symbol_trace(ch) if hasattr(rx, "symbol_trace") else None
        if tr is not None:
            idx, vals = tr
            sel = (idx > 95000) & (idx < 99000)
            print("   soft symbols 95k..99k:", np.round(vals[sel][:40, 1], 2).tolist())
