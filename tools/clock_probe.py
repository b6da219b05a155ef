#!/usr/bin/env python3
"""Shader clock and power while the demodulation kernel runs (back the "~1.5 GHz with every CU
busy" figure, inferred from SQ_WAVE_CYCLES, with a reading taken during the launch).

    python tools/clock_probe.py CHANNELS SECONDS [time_parallel | relaxed]

A thread keeps launching the kernel back to back for ~4 s; the main thread samples the SMU's own figures
every 20 ms from sysfs (hwmon freq1_input = sclk in Hz, power1_average / power1_input in microwatts,
pp_dpm_sclk's starred level) and, once, `rocm-smi --showclocks --showpower` for cross-reference.
"""
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sameold_amd as sa  # noqa: E402


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input"):
            v = read(os.path.join(hw, name))
            if v is not None:
                out[os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(hw)))) + ":" + name] = v
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        v = read(f)
        if v:
            cur = [ln for ln in v.splitlines() if ln.endswith("*")]
            out[f.split("/")[4] + ":sclk_level"] = cur[0] if cur else v.replace("\n", " | ")
    return out


def main():
    C = int(sys.argv[1]); secs = float(sys.argv[2]); tp = len(sys.argv) > 3 and sys.argv[3] == "time_parallel"
    relaxed = len(sys.argv) > 3 and sys.argv[3] == "relaxed"
    T = int(22050 * secs)
    x = sa.synth_afsk(C, T, 22050, seed=1)
    rx = sa.SameReceiverBuilder(22050).build_batch(C, link_only=True, time_parallel=tp, relaxed=relaxed)
    rx.set_kernel_timing(True)
    rx.process_tensor(x); rx.sync(); rx.poll_events_np()
    print("idle:", sensors())
    stop = threading.Event()
    ms = []

    def work():
        while not stop.is_set():
            rx.process_tensor(x)
            rx.sync()
            ms.append(rx.last_kernel_ms())
            rx.drop_events(len(rx.peek_events_np()))

    th = threading.Thread(target=work)
    th.start()
    samples = []
    t0 = time.time()
    while time.time() - t0 < 4.0:
        samples.append(sensors())
        time.sleep(0.02)
    try:
        smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        smi = f"rocm-smi unavailable: {e}"
    stop.set(); th.join()
    keys = sorted({k for s in samples for k in s})
    print(f"{C} channels x {T} samples, {'time-parallel' if tp else ('relaxed' if relaxed else 'strict')} [{rx.kernel_name()}]: {len(ms)} launches, "
          f"kernel ms mean {sum(ms) / max(len(ms), 1):.3f}; {len(samples)} sensor samples while launches ran back to back")
    for k in keys:
        vals = [s[k] for s in samples if k in s]
        try:
            nums = [float(v) for v in vals]
            scale, unit = (1e-6, "MHz") if "freq" in k else ((1e-6, "W") if "power" in k else (1e-3, "C"))
            print(f"  {k}: min {min(nums) * scale:.0f} mean {sum(nums) / len(nums) * scale:.0f} max {max(nums) * scale:.0f} {unit}")
        except ValueError:
            print(f"  {k}: {sorted(set(vals))}")
    print(smi)


if __name__ == "__main__":
    main()
