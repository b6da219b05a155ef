#!/usr/bin/env python3
"""The GPU timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: every kernel between two successive
demodulation launches with its duration and the idle gap before it.   python tools/trace_gaps.py <kernel_trace.csv> [demod name part]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else "demod_"
dem = [i for i, r in enumerate(rows) if key in r[2] and (r[1] - r[0]) > 500_000]
if len(dem) < 8:
    sys.exit(f"only {len(dem)} long demod launches found")
a, b = dem[len(dem) // 2], dem[len(dem) // 2 + 1]
t_prev_end = rows[a - 1][1] if a else rows[a][0]
print(f"step = launch {len(dem)//2} of {len(dem)}: from the start of one demodulation kernel to the start of the next")
tot_busy = tot_gap = 0
for i in range(a, b):
    s, e, n = rows[i]
    gap = s - max(r[1] for r in rows[max(0, i - 6):i]) if i else 0
    name = n.split("(")[0].split("<")[0][-40:]
    print(f"  gap {gap/1e3:8.1f} us | {(e-s)/1e3:9.1f} us  {name}")
    tot_busy += e - s; tot_gap += max(gap, 0)
gap = rows[b][0] - max(r[1] for r in rows[max(0, b - 6):b])
print(f"  gap {gap/1e3:8.1f} us | next demodulation kernel")
print(f"step {(rows[b][0]-rows[a][0])/1e6:.3f} ms: demod {(rows[a][1]-rows[a][0])/1e6:.3f} ms, other kernels {(tot_busy-(rows[a][1]-rows[a][0]))/1e6:.3f} ms, idle {(tot_gap+max(gap,0))/1e6:.3f} ms")
steps = [(rows[dem[i+1]][0]-rows[dem[i]][0])/1e6 for i in range(2, len(dem)-1)]
print("all steps (ms):", " ".join(f"{x:.2f}" for x in steps))
