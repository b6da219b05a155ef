import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, sameold_amd as sa, bench
rate, C, part, calls = 22050, 4096, 110460, 16      # 16 calls of ~5 s (whole blocks of 20 and 42)
x = sa.synth_afsk(C, part * calls, rate, seed=31337)
out = []
for kw, lay in (({}, 0), ({"time_parallel": True}, 1)):
    rx = sa.SameReceiverBuilder(rate).build_batch(C, **kw)
    for i in range(calls):
        p = x[i * part:(i + 1) * part]
        rx.process_tensor(p.t().contiguous() if lay else p.contiguous(), layout=lay)
    rx.sync()
    ev = rx.poll_events_np()
    out.append(ev[np.lexsort((np.arange(len(ev)), ev["channel"]))])
    print(kw, rx.kernel_name(), len(ev), "events", int((ev["kind"] == 3).sum()), "bursts", flush=True)
ok, note = bench.tp_contract(sa, out[0], out[1], C, 31337, f"{calls} calls of one continuous stream", t_end=part * calls, rate=rate)
print(note)
sys.exit(0 if ok else 1)
