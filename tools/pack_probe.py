import numpy as np, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sameold_amd import receiver as R, distributed as sd
ev = np.zeros(108412, dtype=R.EVENT_DTYPE); ev["kind"][::5] = 3; ev["len"] = 60
def T(label, f, n=5):
    best = 1e9
    for _ in range(n):
        t = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t)
    print(f"{label:34s} {1e3*best:8.2f} ms"); return r
idx = np.flatnonzero(ev["kind"] == 3)
raw = ev.view(np.uint8).reshape(-1, ev.dtype.itemsize)
T("uint8 fancy take", lambda: raw[idx])
v = ev.view(np.dtype((np.void, ev.dtype.itemsize)))
T("void take", lambda: v[idx])
T("structured take", lambda: ev[idx])
T("copy 35 MB", lambda: ev.copy())
T("pack_burst_events zero_padded", lambda: sd.pack_burst_events(ev, 0, True))
T("pack_burst_events masked", lambda: sd.pack_burst_events(ev, 0, False))
