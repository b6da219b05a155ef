#!/usr/bin/env python3
"""Round 5: which AGC history reproduces the strict pipeline's wrong first soft sample of the symbol at 3640 (trial 15354)?
f32 restatement of DC blocker -> AGC -> matched filters in numpy, the AGC lock applied from various samples on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sameold_amd as sa
from sameold_amd import montecarlo as mc
f32 = np.float32
rate, seed, grid, n_samples = 22050, 2026, 15, 44096
x = mc.synth_trials(16, 15344, n_samples, rate, seed, 0.0, 1.0, grid)[:, 10].cpu().numpy().astype(f32)[:3700]
# DC blocker, len 16
L = 16; inv = f32(1.0 / 16.0)
w0 = np.zeros(L, f32); w1 = np.zeros(L, f32); s0 = f32(0); s1 = f32(0); y = np.zeros(len(x), f32)
for n, v in enumerate(x):
    aged = w0[n % L]; w0[n % L] = v; s0 = f32(s0 + f32(v - aged)); ma0 = f32(s0 * inv); sig = w0[(n + 1) % L]
    aged1 = w1[n % L]; w1[n % L] = ma0; s1 = f32(s1 + f32(ma0 - aged1)); ma1 = f32(s1 * inv)
    y[n] = f32(sig - ma1)
bw = f32(f32(f32(0.01) * f32(22050 / 520.83)) / f32(22050)); gmin, gmax = f32(0.0), f32(1e6)
sps = f32(22050) / f32(520.83); nt = int(np.floor(sps))
def taps(freq):
    a = f32(f32(2.0) * f32(3.14159274101257324)) * f32(f32(freq) / f32(22050))
    re, im = np.zeros(nt, f32), np.zeros(nt, f32)
    for i in range(nt):
        th = f32(a * f32(nt - 1 - i)); re[i] = f32(f32(2.0) * f32(np.cos(th))) / f32(nt); im[i] = f32(f32(2.0) * f32(-np.sin(th))) / f32(nt)
    return re, im
mre, mim = taps(2083.3); sre, sim = taps(1562.5)
def agc(lock_from):
    g = f32(0.0); out = np.zeros(len(y), f32)
    for n, v in enumerate(y):
        o = f32(v * g); out[n] = o
        if n < lock_from:
            g = f32(g + f32(f32(f32(1.0) - abs(o)) * bw)); g = min(max(g, gmin), gmax)
    return out
def demod(w, n):
    acc = [f32(0)] * 4
    for i in range(nt):
        xv = w[n - i]
        for k, h in enumerate((mre, mim, sre, sim)):
            acc[k] = f32(acc[k] + f32(xv * h[i]))
    m = f32(np.sqrt(np.float64(acc[0]) ** 2 + np.float64(acc[1]) ** 2)); s = f32(np.sqrt(np.float64(acc[2]) ** 2 + np.float64(acc[3]) ** 2))
    return f32(min(max(f32(m - s), f32(-1)), f32(1)))
print("targets: generic 0x1.a156140000000p-1  pipe 0x1.c73bac0000000p-1")
for lock_from in (3601, 3600, 3602, 3620, 3621, 3622, 3640, 10**9):
    w = agc(lock_from)
    print(f"lock from sample {lock_from}:", {n: float(demod(w, n)).hex() for n in (3619, 3620, 3621, 3622)})
