"""AWGN Monte-Carlo over the batched demodulator (BASELINE.json configs[4], SURVEY.md 8d "Config 5").

One trial = one channel carrying a single burst (16 x 0xAB + a random valid header) in white
Gaussian noise at one point of an Eb/N0 grid.  Trials are generated on the device
(`same_synth_trials_device`: Philox4x32-10 + Box-Muller), demodulated by the same kernel as
every other workload, and scored on the host: burst-detection rate, bit error rate over the
bytes the framer delivered, and the rate of bursts that carry the whole header intact.

    python -m sameold_amd.montecarlo --trials 1048576 --batch 65536 --out profiles/r01_ber_sweep.json

The reference has no noise test of its own; "bit-matched" here means the device's events equal
the CPU oracle's on the identical noisy samples (tests/test_gpu_parity.py::test_awgn_trials).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import time
from typing import Dict

import numpy as np

from . import receiver as R

_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint8)


def synth_trials(n_trials: int, first_trial: int, n_samples: int, input_rate: int = 22050, seed: int = 1,
                 ebn0_db_lo: float = 0.0, ebn0_db_step: float = 1.0, n_grid: int = 15, device: int = 0):
    """[n_samples, n_trials] float32 CUDA tensor of noisy single-burst trials."""
    import torch
    L = R.load_library()
    fn = L.same_synth_trials_device
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint64, C.c_float, C.c_float,
                   C.c_uint32, C.c_int, C.c_void_p]
    x = torch.empty((n_samples, n_trials), dtype=torch.float32, device=f"cuda:{device}")
    stream = torch.cuda.current_stream(device).cuda_stream
    R._check(fn(C.c_void_p(x.data_ptr()), n_trials, first_trial, n_samples, input_rate, seed, ebn0_db_lo,
                ebn0_db_step, n_grid, device, C.c_void_p(stream)))
    return x


def new_tally(n_grid: int) -> Dict[str, np.ndarray]:
    z = lambda: np.zeros(n_grid, dtype=np.int64)
    return {"trials": z(), "detected": z(), "intact": z(), "bits": z(), "bit_errors": z(), "short_bytes": z()}


def score_bursts(events: np.ndarray, payloads, first_trial: int, n_trials: int, n_grid: int,
                 tally: Dict[str, np.ndarray]) -> None:
    """Add one batch to `tally`.  `events` is the EVENT_DTYPE array of the batch (any kinds),
    `payloads[c]` the bytes trial first_trial + c transmitted after its preamble.

    Per trial the FIRST burst counts: detected = a burst was delivered at all; bits / bit_errors
    = the bytes the burst and the header have in common (the framer starts the burst at the
    matched "ZCZC" prefix, rx/framing.rs:139-152); short_bytes = header bytes the burst ended
    before; intact = the burst begins with the whole header."""
    grid = (first_trial + np.arange(n_trials)) % n_grid
    np.add.at(tally["trials"], grid, 1)
    b = events[events["kind"] == R.LINK_BURST]
    if len(b) == 0:
        return
    order = np.lexsort((b["sample_counter"], b["channel"]))
    b = b[order]
    ch, first = np.unique(b["channel"], return_index=True)
    b = b[first]
    for rec in b:
        c = int(rec["channel"])
        tx = np.frombuffer(payloads[c], dtype=np.uint8)
        n_rx = min(int(rec["len"]), rec["bytes"].shape[0])
        n = min(n_rx, len(tx))
        rx = rec["bytes"][:n]
        errs = int(_POP8[rx ^ tx[:n]].sum())
        g = grid[c]
        tally["detected"][g] += 1
        tally["bits"][g] += 8 * n
        tally["bit_errors"][g] += errs
        tally["short_bytes"][g] += len(tx) - n
        tally["intact"][g] += int(errs == 0 and n == len(tx))


def summarise(tally: Dict[str, np.ndarray], ebn0_db_lo: float, ebn0_db_step: float):
    rows = []
    for g in range(len(tally["trials"])):
        t = int(tally["trials"][g])
        bits = int(tally["bits"][g])
        rows.append({
            "ebn0_db": ebn0_db_lo + g * ebn0_db_step,
            "trials": t,
            "burst_detection_rate": tally["detected"][g] / t if t else None,
            "intact_header_rate": tally["intact"][g] / t if t else None,
            "ber": tally["bit_errors"][g] / bits if bits else None,
            "bits_compared": bits,
            "bit_errors": int(tally["bit_errors"][g]),
            "header_bytes_cut_short": int(tally["short_bytes"][g]),
        })
    return rows


def ber_sweep(total_trials: int = 1 << 20, batch: int = 65536, input_rate: int = 22050, seconds: float = 2.0,
              seed: int = 2026, ebn0_db_lo: float = 0.0, ebn0_db_step: float = 1.0, n_grid: int = 15,
              device: int = 0, verbose: bool = False):
    """Run the sweep; returns (rows, timing dict)."""
    import torch
    n_samples = int(round(input_rate * seconds))
    n_samples -= n_samples % 16
    tally = new_tally(n_grid)
    t_gen = t_demod = t_score = 0.0
    kernel_ms = []
    done = 0
    while done < total_trials:
        n = min(batch, total_trials - done)
        t0 = time.perf_counter()
        x = synth_trials(n, done, n_samples, input_rate, seed, ebn0_db_lo, ebn0_db_step, n_grid, device)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # every trial meets a freshly built receiver (SameReceiverBuilder::build, not reset())
        rx = R.SameReceiverBuilder(input_rate).build_batch(n, device=device, link_only=True)
        rx.set_kernel_timing(True)
        rx.process_tensor(x)
        rx.sync()
        ev = rx.poll_events_np()
        kernel_ms.append(rx.last_kernel_ms())
        t2 = time.perf_counter()
        payloads = [R.synth_payload(seed, done + c) for c in range(n)]
        score_bursts(ev, payloads, done, n, n_grid, tally)
        t3 = time.perf_counter()
        t_gen += t1 - t0; t_demod += t2 - t1; t_score += t3 - t2
        done += n
        del x
        if verbose:
            print(f"  {done}/{total_trials} trials  (generate {t1 - t0:.2f} s, demodulate {t2 - t1:.2f} s, score {t3 - t2:.2f} s)",
                  flush=True)
    timing = {"generate_s": round(t_gen, 3), "demodulate_s": round(t_demod, 3), "score_s": round(t_score, 3),
              "demod_kernel_ms_per_batch": round(float(np.mean(kernel_ms)), 3), "batch": batch,
              "samples_per_trial": n_samples,
              "demod_kernel_Msamples_per_s": round(batch * n_samples / (float(np.mean(kernel_ms)) * 1e-3) / 1e6, 1)}
    return summarise(tally, ebn0_db_lo, ebn0_db_step), timing


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--trials", type=int, default=1 << 20)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--rate", type=int, default=22050)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--ebn0-lo", type=float, default=0.0)
    ap.add_argument("--ebn0-step", type=float, default=1.0)
    ap.add_argument("--grid", type=int, default=15)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows, timing = ber_sweep(a.trials, a.batch, a.rate, a.seconds, a.seed, a.ebn0_lo, a.ebn0_step, a.grid, verbose=True)
    print(f"{'Eb/N0 dB':>8} {'trials':>8} {'detect':>8} {'intact':>8} {'BER':>11}")
    for r in rows:
        ber = f"{r['ber']:.3e}" if r["ber"] is not None else "-"
        print(f"{r['ebn0_db']:8.1f} {r['trials']:8d} {r['burst_detection_rate']:8.4f} {r['intact_header_rate']:8.4f} {ber:>11}")
    print(json.dumps(timing))
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"workload": f"{a.trials} AWGN trials, one burst each, {a.rate} Hz, {a.seconds} s per trial, "
                                   f"Eb/N0 {a.ebn0_lo}..{a.ebn0_lo + (a.grid - 1) * a.ebn0_step} dB",
                       "seed": a.seed, "rows": rows, "timing": timing}, f, indent=1)


if __name__ == "__main__":
    main()
