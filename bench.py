#!/usr/bin/env python3
"""bench.py -- batched SAME AFSK demodulation throughput on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched under torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): 4 096 synthetic 22.05 kHz AFSK channels per GPU,
f32, `--seconds` of audio per step (default 10 s = 220 500 samples/channel, 3.6 GB),
generated on the device so the timed region starts with the input resident in HBM.
A "step" is one pass of the whole link layer (DC block -> AGC -> matched filters ->
timing loop -> squelch -> DFE -> framer) over that batch through the C ABI, including
the event-log read-back, ordering and the host-side transport layer.  Channels shard
across ranks with no data-path collective; each step ends with one RCCL gather of the
decoded bursts to rank 0 (weak scaling: per-GPU work is fixed).

roofline: algorithmic bytes = 4 B per input sample divided by the
demodulation kernel's duration, measured with HIP events on the stream the kernel runs on.
cpu_baseline: the oracle (scalar C port of the reference's Rust path; the reference
itself cannot be built in this image) on a bounded sample of the same input, all host
cores.  `scaled`: the same kernel on 32 768 channels (the per-GPU shard of
BASELINE.json configs[3]), `configs2_48k`: 16 384 channels at 48 kHz (configs[2]) -- both
reported beside, never instead of, the configs[1] value (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=4096, help="channels per GPU")
    ap.add_argument("--rate", type=int, default=22050)
    ap.add_argument("--seconds", type=float, default=10.0, help="audio per channel per step")
    ap.add_argument("--cpu-channels", type=int, default=1024, help="channels of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaled", action="store_true", help="skip the extra 32768-channel and 48 kHz measurements")
    ap.add_argument("--scaled-channels", type=int, default=32768)
    ap.add_argument("--check", type=int, default=16, help="channels of rank 0 verified against the oracle")
    ap.add_argument("--traffic", type=float, default=None, help="HBM bytes/launch from a PMC pass (profiles/)")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="minimum wall time of the CPU baseline run")
    ap.add_argument("--mode", choices=["auto", "strict", "time_parallel", "time_parallel_time_major"], default="auto",
                    help="which mode the headline value reports: auto = time-parallel when its parity contract holds on this "
                         "run's own first pass (payload bytes of every burst and every transport message equal to strict "
                         "mode's on every channel), strict otherwise; both are always measured and reported")
    ap.add_argument("--plumbing", action="store_true",
                    help="CPU-only check of the N-rank path (gloo, fabricated burst records, no kernel)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under
    torch.distributed.run as a CHILD process and pass its output and exit code on.  Decided before
    anything touches torch.cuda (a process that has initialised the GPU must never exec or be
    replaced; this one never initialises it at all)."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def plumbing(args):
    """The N-rank path without a GPU: gloo process group, every rank fabricates the burst records of
    its own channel shard, one gather per step to rank 0, max-over-ranks timing, rank 0 prints the
    JSON line.  No demodulation happens (value 0); tests/test_distributed_cpu.py runs this."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from sameold_amd import distributed as sd
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    C = args.channels
    first_ch = rank * C
    per_rank = 3 + rank
    recs = sd.pack_bursts([(first_ch + i, 1000 * rank + i, b"ZCZC-PLUMBING-%d-" % rank) for i in range(per_rank)])
    got = 0
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        r = sd.gather_records(recs, torch.device("cpu"))
        got = len(r) if r is not None else 0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "plumbing (no demodulation)", "value": 0.0, "unit": "Msamples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
                          "scaling": "weak", "data": "fabricated burst records",
                          "config": {"workload": "plumbing", "channels_per_gpu": C,
                                     "bursts_gathered_last_step": int(got),
                                     "first_channels": [r * C for r in range(world)]}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_steps(sa, rx, x, T, stream, steps, warmup, gather, barrier, layout=0):
    """W untimed + K timed passes.  Each pass launches one batch; the library collects the
    previous batch's event log (copy back, ordering, transport layer) while the new launch
    runs, so a pass consumes the events of the batch before it and the last batch is
    drained after the loop -- inside the timed region.  Returns (elapsed_s, mean kernel ms,
    events of the first pass, bursts gathered from the last drained batch)."""
    import numpy as np
    import torch
    kernel_ms = []
    first = []
    keep_first = [True]
    last_bursts = [0]

    def consume():
        ev = rx.peek_events_np()           # non-blocking: what the host already has, viewed in place
        if len(ev):
            kernel_ms.append(rx.last_kernel_ms())
            if keep_first[0]:
                first.append(ev.copy())
            last_bursts[0] = gather(rx)    # (copies the burst records out of the queue)
            rx.drop_events(len(ev))

    def one_pass():
        rx.process_device_ptr(x.data_ptr(), T, layout, stream)
        consume()

    def drain():
        rx.sync()
        consume()

    for _ in range(warmup):
        one_pass()
    drain()
    if first:
        keep_first[0] = False
    kernel_ms.clear()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for _ in range(steps):
        one_pass()
        marks.append(time.perf_counter())
    drain()
    marks.append(time.perf_counter())
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("SAME_BENCH_DEBUG"):
        sys.stderr.write("per-pass wall ms (last = drain): " + " ".join(f"{1e3 * (b - a):.2f}" for a, b in zip([t0] + marks[:-1], marks)) + "\n")
    first_ev = np.concatenate(first) if first else np.zeros(0, dtype=sa.receiver.EVENT_DTYPE)
    first_ev = first_ev[first_ev["sample_counter"] <= T]     # the very first pass only
    return elapsed, sum(kernel_ms) / max(len(kernel_ms), 1), first_ev, last_bursts[0]


def tp_contract(sa, first_strict, first_tp, C, seed):
    """The time-parallel mode's contract on this run's own first pass (fresh state in both modes): per
    channel the same number of bursts, every transmitted byte of every burst equal (the header, or NNNN;
    bytes decoded after the carrier stops are not compared), the same transport messages in the same
    order.  Returns (ok, note)."""
    import numpy as np

    def per_channel(ev, kind_lo, kind_hi):
        e = ev[(ev["kind"] >= kind_lo) & (ev["kind"] <= kind_hi)]
        first = np.searchsorted(e["channel"], np.arange(C + 1))
        return e, first

    bs, fs = per_channel(first_strict, 3, 3)
    bt, ft = per_channel(first_tp, 3, 3)
    ms, gs = per_channel(first_strict, 18, 20)
    mt, gt = per_channel(first_tp, 18, 20)
    bad_count = bad_payload = bad_msg = n_bursts = 0
    for c in range(C):
        a, b = bs[fs[c]:fs[c + 1]], bt[ft[c]:ft[c + 1]]
        n_bursts += len(a)
        if len(a) != len(b):
            bad_count += 1
            continue
        pay = sa.synth_payload(seed, c)
        for ra, rb in zip(a, b):
            xa = ra["bytes"][: int(ra["len"])].tobytes()
            n = len(pay) if xa[:4] == pay[:4] else 4
            if xa[:n] != rb["bytes"][:n].tobytes():
                bad_payload += 1
        ma, mb = ms[gs[c]:gs[c + 1]], mt[gt[c]:gt[c + 1]]
        if len(ma) != len(mb) or not np.array_equal(ma["kind"], mb["kind"]) or not np.array_equal(ma["bytes"], mb["bytes"]):
            bad_msg += 1
    ok = bad_count == 0 and bad_payload == 0 and bad_msg == 0 and n_bursts > 0
    return ok, (f"first pass vs strict mode, {C} channels, {n_bursts} bursts: {bad_count} channels with a different burst count, "
                f"{bad_payload} bursts with a different payload, {bad_msg} channels with different transport messages -> "
                f"{'OK' if ok else 'VIOLATED'}")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if args.plumbing:
        return plumbing(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start it as `python bench.py --gpus N` "
                         "(it launches the ranks itself) or under torch.distributed.run with a matching --nproc-per-node")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from sameold_amd import build as sbuild
    if rank == 0:
        sbuild.build()
    if distributed:
        dist.barrier()
    import sameold_amd as sa
    from sameold_amd import distributed as sd

    C = args.channels
    T = int(round(args.rate * args.seconds))
    first_ch = rank * C                       # weak scaling: every rank owns C channels
    x = sa.synth_afsk(C, T, args.rate, seed=20260000 + rank, device=local_rank)
    torch.cuda.synchronize()
    rx = sa.SameReceiverBuilder(args.rate).build_batch(C, device=local_rank)
    rx.set_kernel_timing(True)
    # None = the library's own non-blocking stream.  (The input was produced on torch's stream and
    # synchronised above; x stays alive for the whole run.  On the legacy null stream the runtime holds
    # a launch enqueued behind a running kernel until the next API call, which would serialise launch
    # k+1 with harvest k.)
    stream = None

    def gather(rx_):
        # the step's one collective: every rank's burst records to rank 0 (RCCL; packed by the library)
        recs = rx_.pack_bursts_np(first_ch)
        if not distributed:
            return len(recs)
        got = sd.gather_records(recs, dev)
        return len(got) if got is not None else 0

    def barrier():
        if distributed:
            dist.barrier()

    def max_over_ranks(v):
        if not distributed:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # strict mode (bit-exact), then the time-parallel mode on the same input with a second receiver
    elapsed, k_ms, first, n_bursts = run_steps(sa, rx, x, T, stream, args.steps, args.warmup, gather, barrier)
    elapsed = max_over_ranks(elapsed)
    # ... (a) on the same time-major buffer: one row offset per workgroup, uniform chunk boundaries, chunks run on until idle
    rx_tp = sa.SameReceiverBuilder(args.rate).build_batch(C, device=local_rank, time_parallel=True)
    rx_tp.set_kernel_timing(True)
    elapsed_tp, k_ms_tp, first_tp, n_bursts_tp = run_steps(sa, rx_tp, x, T, stream, args.steps, args.warmup, gather, barrier)
    elapsed_tp = max_over_ranks(elapsed_tp)
    tp_chunks = rx_tp.time_parallel_chunks()
    tp_ok, tp_note = tp_contract(sa, first, first_tp, C, 20260000 + rank)
    del rx_tp
    # ... (b) on a channel-major copy of it (every channel a contiguous stream, what a capture front end that delivers
    # per-channel buffers hands over): chunk boundaries per channel at idle instants, no run-on
    xc = x.t().contiguous()
    torch.cuda.synchronize()
    rx_cm = sa.SameReceiverBuilder(args.rate).build_batch(C, device=local_rank, time_parallel=True)
    rx_cm.set_kernel_timing(True)
    elapsed_cm, k_ms_cm, first_cm, n_bursts_cm = run_steps(sa, rx_cm, xc, T, stream, args.steps, args.warmup, gather, barrier,
                                                           layout=sa.LAYOUT_CHANNEL_MAJOR)
    elapsed_cm = max_over_ranks(elapsed_cm)
    cm_chunks, cm_per_channel = rx_cm.time_parallel_chunks(), rx_cm.time_parallel_per_channel()
    cm_ok, cm_note = tp_contract(sa, first, first_cm, C, 20260000 + rank)
    del rx_cm, xc
    torch.cuda.empty_cache()
    if distributed:
        t = torch.tensor([1.0 if tp_ok else 0.0, 1.0 if cm_ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        tp_ok, cm_ok = bool(t[0].item() > 0.5), bool(t[1].item() > 0.5)

    # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    # profiles/r01_fetch_calibration.txt); only valid for the workload it was collected on
    def pmc_traffic(kind):
        if args.traffic is not None:
            return args.traffic
        for name in ("r02_traffic.json", "r01_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    tj = json.load(f)
                for ent in (tj if isinstance(tj, list) else [tj]):
                    if ent.get("workload") == f"{C} ch x {T} samples" and ent.get("mode", "strict") == kind:
                        return ent["hbm_bytes_per_launch"]
            except Exception:
                pass
        return None

    def mode_block(kind, el, kms, nb, note):
        ach = 4.0 * C * T / (kms * 1e-3) / 1e9
        return {
            "value": round(C * T * world * args.steps / el / 1e6, 2), "unit": "Msamples/s",
            "ms_per_step": round(el / args.steps * 1e3, 3), "bursts_gathered_last_step": int(nb),
            "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(kind), "kernel_ms": round(kms, 4),
                         "algorithmic_bytes_per_launch": 4 * C * T, "note": note},
        }

    modes = {
        "strict": mode_block("strict", elapsed, k_ms, n_bursts,
                             "bit-exact; latency-bound serial streams: 256 workgroups (16 channels each) x 5 pipeline-stage "
                             "wavefronts (DESIGN.md 4.4, 4.4b)"),
        "time_parallel": mode_block("time_parallel", elapsed_cm, k_ms_cm, n_bursts_cm,
                                    f"channel-major input x[channel][t]; {cm_chunks} time chunks per channel = {cm_chunks * C} state columns through the "
                                    "same pipeline kernel, chunk boundaries per channel at idle instants (device-side energy scout + planner, "
                                    "inside kernel_ms), strict arithmetic per chunk (DESIGN.md 4.6)"),
        "time_parallel_time_major": mode_block("time_parallel_time_major", elapsed_tp, k_ms_tp, n_bursts_tp,
                                               f"time-major input; {tp_chunks} chunks per channel with uniform boundaries (one row offset per workgroup keeps "
                                               "the loads coalesced), chunks run on until idle (DESIGN.md 4.6); kernel_ms includes the state column copies"),
    }
    modes["time_parallel"].update(chunks=int(cm_chunks), per_channel_boundaries=bool(cm_per_channel), contract=cm_note,
                                  layout="channel-major x[channel][t]")
    modes["time_parallel_time_major"].update(chunks=int(tp_chunks), contract=tp_note, layout="time-major x[t][channel]")
    modes["strict"]["layout"] = "time-major x[t][channel]"
    if args.mode != "auto":
        headline = args.mode
    else:
        ok = [m for m, good, k in (("time_parallel", cm_ok, cm_chunks), ("time_parallel_time_major", tp_ok, tp_chunks)) if good and k > 1]
        headline = max(ok, key=lambda m: modes[m]["value"]) if ok else "strict"
    hb = modes[headline]
    out = {
        "metric": "Msamples/s demodulated (batched 22.05 kHz channels) + % HBM roofline, 1/8 GPU",
        "value": hb["value"],
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": hb["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{C} synthetic {args.rate / 1000:g} kHz AFSK channels per GPU, f32, "
                        f"{args.seconds:g} s ({T} samples) per channel per step (BASELINE.json configs[1])",
            "channels_per_gpu": C, "samples_per_channel": T, "input_rate": args.rate,
            "layout": modes[headline]["layout"], "kernel": rx.kernel_name(),
            "mode": headline,
            "parity": ("time-parallel: every burst's transmitted bytes and every transport message equal to strict mode's "
                       "(include/same_rx.h, tests/test_time_parallel.py); strict mode is bit-exact to the oracle"
                       if headline != "strict" else "bit-exact (strict op order)"),
            "bursts_gathered_last_step": hb["bursts_gathered_last_step"], "events_first_step_rank0": int(len(first)),
        },
        "roofline": hb["roofline"],
        "modes": modes,
    }

    if rank == 0 and world == 1:
        from oracle import binding as ob
        cfg = ob.default_config(args.rate)
        xs = None
        if args.check and args.warmup + args.steps > 0:
            # parity spot-check of this very run: the first pass started from fresh state
            chk = min(args.check, C)
            xs = x[:, :max(chk, min(args.cpu_channels, C))].contiguous().cpu().numpy()
            ok = True
            for c in range(chk):
                mine = first[first["channel"] == c]
                got = [(int(r["kind"]), int(r["sample_counter"]), r["bytes"][: min(int(r["len"]), 288)].tobytes()) for r in mine]
                ref = [e.as_tuple() for e in ob.Receiver(cfg).run(np.ascontiguousarray(xs[:, c]))]
                ok &= got == ref
            out["config"]["parity_check"] = f"{chk} channels of this run vs oracle: {'OK' if ok else 'MISMATCH'}"
            if not ok:
                out["config"]["parity"] = "MISMATCH"
        if not args.no_cpu_baseline:
            # One SameReceiver per channel over a CONTIGUOUS stream (the sample is transposed to
            # channel-major on the host first), one worker pinned per physical core, repeated until the
            # run is long enough to time arithmetic rather than thread start-up.
            cc = min(args.cpu_channels, C)
            if xs is None or xs.shape[1] < cc:
                xs = x[:, :cc].contiguous().cpu().numpy()
            xc = np.ascontiguousarray(xs[:, :cc].T)          # [channel][time]
            cores = ob.physical_cores()
            logical = len(os.sched_getaffinity(0))
            t1 = time.perf_counter()
            ob.batch_run_channel_major(cfg, xc, cpus=cores, reps=1)
            probe = time.perf_counter() - t1
            reps = max(1, int(np.ceil(args.cpu_seconds / max(probe, 1e-3))))
            t1 = time.perf_counter()
            ob.batch_run_channel_major(cfg, xc, cpus=cores, reps=reps)
            dt = time.perf_counter() - t1
            rate = cc * T * reps / dt / 1e6
            out["cpu_baseline"] = {
                "value": round(rate, 2), "unit": "Msamples/s", "cores": len(cores), "kind": "port",
                "per_core": round(rate / len(cores), 2), "logical_cpus": logical,
                "sample": f"first {cc} channels x {T} samples of the same synthetic input, channel-major on the host, "
                          f"{reps} repetition(s), link layer only, {dt:.2f} s wall on {len(cores)} threads pinned one per "
                          f"physical core (scalar C restatement of sameold 0.6.0; the Rust reference cannot be built in this image)",
            }
            del xc
        del xs
        if not args.no_scaled:
            # same kernel, the per-GPU shard of configs[3]: 32768 channels (2 s per step to bound memory)
            del x
            torch.cuda.empty_cache()
            Cs, Ts = args.scaled_channels, int(args.rate * 2)
            x2 = sa.synth_afsk(Cs, Ts, args.rate, seed=777, device=local_rank)
            rx2 = sa.SameReceiverBuilder(args.rate).build_batch(Cs, device=local_rank)
            rx2.set_kernel_timing(True)
            n2 = max(args.steps, 10)          # enough passes for launch k+1 to hide harvest k (first wait and last drain are inside the timed region)
            e2, k2, _, _ = run_steps(sa, rx2, x2, Ts, stream, n2, 2, lambda r: len(r.pack_bursts_np(0)), lambda: None)
            a2 = 4.0 * Cs * Ts / (k2 * 1e-3) / 1e9
            out["scaled"] = {
                "workload": f"{Cs} channels x {Ts} samples per step (per-GPU shard of BASELINE.json configs[3])",
                "value": round(Cs * Ts * n2 / e2 / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(k2, 4), "steps": n2,
                "ms_per_step": round(e2 / n2 * 1e3, 3),
                "roofline": {"bound": "hbm", "achieved": round(a2, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(a2 / HBM_PEAK_GBS, 5)},
            }
            # and configs[2]: 16384 channels at 48 kHz (92-tap filters, 32-sample blocks), 2 s per step
            del x2, rx2
            torch.cuda.empty_cache()
            C3, R3 = 16384, 48000
            T3 = R3 * 2
            x3 = sa.synth_afsk(C3, T3, R3, seed=778, device=local_rank)
            rx3 = sa.SameReceiverBuilder(R3).build_batch(C3, device=local_rank)
            rx3.set_kernel_timing(True)
            e3, k3, _, _ = run_steps(sa, rx3, x3, T3, stream, n2, 2, lambda r: len(r.pack_bursts_np(0)), lambda: None)
            a3 = 4.0 * C3 * T3 / (k3 * 1e-3) / 1e9
            out["configs2_48k"] = {
                "workload": f"{C3} channels x {T3} samples per step at {R3} Hz (BASELINE.json configs[2], 2 s of its 10 s)",
                "value": round(C3 * T3 * n2 / e3 / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(k3, 4), "steps": n2,
                "ms_per_step": round(e3 / n2 * 1e3, 3),
                "kernel": rx3.kernel_name(),
                "roofline": {"bound": "hbm", "achieved": round(a3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(a3 / HBM_PEAK_GBS, 5)},
            }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
