#!/usr/bin/env python3
"""How many CPUs the headline stream keeps busy (process CPU time over wall time), against the container's CFS quota, and the
throttling the kernel reports for the cgroup over the run (cpu.stat).   python tools/cpu_use_probe.py [steps]"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
def cpu_stat():
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try: return dict(l.split() for l in open(p).read().strip().splitlines())
        except OSError: pass
    return {}
C, T = 4096, 220500
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
x = sa.synth_afsk(C, T, 22050, seed=20260000).t().contiguous(); torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, time_parallel=True)
for i in range(20):
    rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_CHANNEL_MAJOR, None)
    n = rx.pending_events()
    if n: rx.pack_bursts_np(0); rx.drop_events(n)
rx.sync(); rx.drop_events(rx.pending_events())
s0 = cpu_stat(); r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
long_passes = []
for i in range(steps):
    a = time.perf_counter()
    rx.process_device_ptr(x.data_ptr(), T, sa.LAYOUT_CHANNEL_MAJOR, None)
    n = rx.pending_events()
    if n: rx.pack_bursts_np(0); rx.drop_events(n)
    b = time.perf_counter()
    if b - a > 3e-3: long_passes.append(round((b - a) * 1e3, 1))
rx.sync()
wall = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF); s1 = cpu_stat()
cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
print(f"{steps} steps: wall {wall/steps*1e3:.3f} ms/step, process CPU {cpu/steps*1e3:.2f} CPU-ms/step (user {1e3*(r1.ru_utime-r0.ru_utime)/steps:.2f} + system {1e3*(r1.ru_stime-r0.ru_stime)/steps:.2f}) = {cpu/wall:.1f} CPUs busy; "
      f"SAME_HOST_THREADS={os.environ.get('SAME_HOST_THREADS','-')}")
print(f"  passes longer than 3 ms: {len(long_passes)} of {steps}" + (f"; the longest: {sorted(long_passes)[-8:]}" if long_passes else ""))
for k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec"):
    if k in s0 and k in s1: print(f"  cgroup {k}: +{int(s1[k]) - int(s0[k])}")
