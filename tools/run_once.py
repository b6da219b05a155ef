#!/usr/bin/env python3
"""One timed pass of the hot path for profiling: python tools/run_once.py [channels] [seconds] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sameold_amd as sa

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rate = int(sys.argv[4]) if len(sys.argv) > 4 else 22050
T = int(rate * secs)
x = sa.synth_afsk(C, T, rate, seed=1, noise_sigma=float(os.environ.get('SAME_NOISE', '0')))
torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(rate).build_batch(C, link_only=not os.environ.get('SAME_TRANSPORT'),
                                             time_parallel=bool(os.environ.get('SAME_TP')), relaxed=bool(os.environ.get('SAME_RELAXED_MODE')))
rx.set_kernel_timing(True)
for r in range(reps):
    t0 = time.perf_counter()
    rx.process_tensor(x)
    rx.sync()
    dt = time.perf_counter() - t0
    ms = rx.last_kernel_ms()
    if r + 1 < reps and os.environ.get('SAME_TP'):
        rx.drop_events(len(rx.peek_events_np())); rx.reset()
    n = rx._L.same_batch_pending_events(rx._h)
    print(f"rep {r}: kernel {ms:.3f} ms  wall {dt*1e3:.3f} ms  {C*T/ms/1e3:.1f} Msamples/s (kernel)  "
          f"{4*C*T/ms/1e6:.2f} GB/s  events pending {n}  [{rx.kernel_name()}]", flush=True)

if hasattr(rx._L, "same_debug_profile"):
    import ctypes
    buf = (ctypes.c_ulonglong * 15)()
    rx._L.same_debug_profile(buf, 1)
    names = ["sample phase", "matched filter", "timing loop", "squelch", "equalizer step", "byte/framer", "events+ticks", "post/latch", "(one mark)"]
    tot = sum(buf)
    nblk = reps * (T // 16)
    for n, v in zip(names, buf):
        print(f"  {n:16s} {v:14d} clk  {100.0*v/max(tot,1):5.1f} %  {v/nblk:8.1f} clk/block")
    print(f"  total {tot} clk over {nblk} blocks = {tot/nblk:.1f} clk/block (each executed mark costs about the '(one mark)' figure, charged to the section after it)")

if hasattr(rx._L, "same_debug_profile_pipe") and "pipe" in rx.kernel_name():
    import ctypes
    buf = (ctypes.c_ulonglong * 15)()
    rx._L.same_debug_profile_pipe(buf, 1)
    nstep = reps * (T // {22050: 20, 48000: 32, 44100: 36}.get(rate, 32) + 3)
    if os.environ.get("SAME_P3_MARKS"):
        for name, v in zip(["other (mailbox, barrier, idle)", "squelch", "equalizer step", "byte/framer", "events+wake-ups", "-"], buf):
            print(f"  stage 3 {name:32s} {v/nstep:8.1f} clk/step (each mark costs ~340 clk, charged to the section after it)")
    else:
      for r, name in enumerate(["stage 1 (sample phase)", "stage 2 (filters + timing)", "stage 3 (symbol path)", "stage 4 (helper: filters, events)", "DC wave"]):
        w, b, f = buf[3 * r], buf[3 * r + 1], buf[3 * r + 2]
        if w + b + f == 0:
            continue
        print(f"  {name:28s} work {w/nstep:8.1f}  barrier wait {b/nstep:8.1f}  feedback {f/nstep:8.1f}  clk/step (total {(w+b+f)/nstep:8.1f})")
    if hasattr(rx._L, "same_debug_profile_hw"):
        hw = (ctypes.c_ulonglong * 8)()
        rx._L.same_debug_profile_hw(hw, 1)
        print("  HW_ID (SIMD) per role [stage1, stage2, stage3, stage4, DC wave]:",
              [f"{int(v) & 0xffffffff:#x} (simd {(int(v) >> 4) & 3}, wave slot {int(v) & 15})" for v in hw[:5]])
        print(f"  stage 2 polled stage 4 for the space magnitude {hw[5]/nstep:8.1f} clk/step; stage 4's space filter took {hw[6]/nstep:8.1f} clk/step")
        print(f"  second TED instants inside one block (all workgroups): {int(hw[7])}")

    if hasattr(rx._L, "same_debug_profile_s2"):
        s2 = (ctypes.c_ulonglong * 8)()
        rx._L.same_debug_profile_s2(s2, 1)
        names = ["mark filter + hypot", "polling stage 4", "combine + timing loop + next instant", "posting", "checkpoint + loop + barrier entry"]
        print("  stage 2 sections (clk/step): " + ", ".join(f"{n} {s2[i]/nstep:.0f}" for i, n in enumerate(names)))
