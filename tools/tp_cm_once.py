#!/usr/bin/env python3
"""One time-parallel pass over a channel-major input (per-channel chunk boundaries); SAME_DEBUG=1 prints how the
boundaries came out.   [TP_CHUNKS=K] [TP_WARMUP=samples] python tools/tp_cm_once.py CHANNELS SECONDS [REPS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa
C = int(sys.argv[1]); secs = float(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T = int(22050 * secs)
x = sa.synth_afsk(C, T, 22050, seed=20260000)
xc = x.t().contiguous()
del x
torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, link_only=True, time_parallel=True)
if os.environ.get("TP_CHUNKS"):
    rx.time_parallel_config(max_chunks=int(os.environ["TP_CHUNKS"]), warmup_samples=int(os.environ.get("TP_WARMUP", "0")))
rx.set_kernel_timing(True)
for r in range(reps):
    if r:
        rx.reset()
    t0 = time.perf_counter()
    rx.process_tensor(xc, layout=sa.LAYOUT_CHANNEL_MAJOR)
    rx.sync()
    print(f"rep {r}: kernel {rx.last_kernel_ms():.3f} ms (demodulation alone {rx.last_demod_kernel_ms():.3f})  wall {(time.perf_counter() - t0) * 1e3:.1f} ms  chunks {rx.time_parallel_chunks()} "
          f"per-channel {rx.time_parallel_per_channel()}  events {len(rx.poll_events_np())}", flush=True)

if hasattr(rx._L, "same_debug_profile_pipe"):
    import ctypes
    buf = (ctypes.c_ulonglong * 15)()
    rx._L.same_debug_profile_pipe(buf, 1)
    names = ["stage 1", "stage 2", "stage 3", "stage 4 (helper)"]
    tot = [buf[3 * r] + buf[3 * r + 1] + buf[3 * r + 2] for r in range(4)]
    for r in range(4):
        print(f"  {names[r]}: work {buf[3 * r] / max(tot[r], 1):.3f}  barrier wait {buf[3 * r + 1] / max(tot[r], 1):.3f}  feedback {buf[3 * r + 2] / max(tot[r], 1):.3f}  (fractions of workgroup 0's time; total {tot[r]} clk)")
