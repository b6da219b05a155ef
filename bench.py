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
cores the container's CPU quota covers.  `scaled`: 32 768 channels (the per-GPU shard of
BASELINE.json configs[3]) in strict and in relaxed arithmetic, `scaled_big`: 131 072 channels
likewise (strict: the one-wavefront kernel; relaxed: the symbol-paced pipeline in rounds), `configs2_48k`: 16 384 channels at 48 kHz
(configs[2]) in strict and in relaxed arithmetic, `scaled_long` (behind --scaled-long): the
32 768-channel shard with 10 s per step in time-parallel mode -- every relaxed block with its
contract check against the strict pass, all reported beside, never instead of, the configs[1]
value (rank 0, N = 1 only).  `--workload configs3` makes the
32 768-channel shard (2 s per step) the workload of every rank: the 8-GPU form of configs[3].
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["configs1", "configs3"], default="configs1",
                    help="configs1: 4 096 channels x 10 s per GPU (the metric's configuration); configs3: 32 768 channels x 2 s per GPU "
                         "(BASELINE.json configs[3]: 262 144 channels over 8 GPUs)")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default: the workload's)")
    ap.add_argument("--rate", type=int, default=22050)
    ap.add_argument("--seconds", type=float, default=None, help="audio per channel per step (default: the workload's)")
    ap.add_argument("--cpu-channels", type=int, default=1024, help="channels of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat-ms", type=float, default=350.0,
                    help="untimed passes of the same step for this long before every block's warm-up, so that the timed steps run at the "
                         "GPU's sustained clock instead of on its ramp from idle (0: none -- profiling passes, whose launch lists stay short)")
    ap.add_argument("--no-scaled", action="store_true", help="skip the extra 32768-channel and 48 kHz measurements")
    ap.add_argument("--scaled-channels", type=int, default=32768)
    ap.add_argument("--check", type=int, default=16, help="channels of rank 0 verified against the oracle")
    ap.add_argument("--traffic", type=float, default=None, help="HBM bytes/launch from a PMC pass (profiles/)")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="minimum wall time of the CPU baseline run")
    ap.add_argument("--scaled-long", action="store_true", help="also the 32768-channel x 10 s time-parallel block (29 GB of input): a full "
                    "machine gains nothing from a cut in time, the block documents that (DESIGN.md 4.7)")
    ap.add_argument("--no-scaled-long", action="store_true", help="(accepted for older command lines: the block is off unless --scaled-long)")
    ap.add_argument("--no-scaled-big", action="store_true", help="skip the 131072-channel block (23 GB of input)")
    ap.add_argument("--mode", choices=["auto", "strict", "time_parallel", "time_parallel_time_major", "time_parallel_strict_chunks", "relaxed"], default="auto",
                    help="which mode the headline value reports: auto = time-parallel when its parity contract holds on this "
                         "run's own first pass (payload bytes of every burst and every transport message equal to strict "
                         "mode's on every channel), strict otherwise; both are always measured and reported")
    ap.add_argument("--no-carried-state-check", action="store_true",
                    help="skip the three-call continuous-stream contract of every mode (profiling passes: keeps the launch list to the bench blocks)")
    ap.add_argument("--plumbing", action="store_true",
                    help="CPU-only check of the N-rank path (gloo, fabricated burst records, no kernel)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself")
    a = ap.parse_args()
    if a.channels is None:
        a.channels = 4096 if a.workload == "configs1" else 32768
    if a.seconds is None:
        a.seconds = 10.0 if a.workload == "configs1" else 2.0
    return a


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
    """The N-rank path without a GPU: gloo process group; every rank fabricates the link events of its own channel shard
    (Reading + Burst per transmission, the device's record layout), packs the bursts as the bench does (global channel
    numbers), one gather per step to rank 0, max-over-ranks timing; rank 0 checks what arrived record by record and prints
    the JSON line with the same `ranks` block as a real run.  `--workload configs3` sizes it like a real step of the
    262 144-channel workload (32 768 channels per rank, 3 bursts per 4 channels: ~7.5 MB of records per rank per step).
    No demodulation happens (value 0); tests/test_distributed_cpu.py runs this with 2 and with 8 ranks."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from sameold_amd import distributed as sd
    from sameold_amd.receiver import EVENT_DTYPE
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    C = args.channels
    first_ch = rank * C
    per_rank = (3 * C) // 4 if args.workload == "configs3" else 3 + rank

    def fabricate(r):
        n = (3 * C) // 4 if args.workload == "configs3" else 3 + r
        ev = np.zeros(2 * n, dtype=EVENT_DTYPE)
        ch = (np.arange(n, dtype=np.uint32) * 4) // 3 if args.workload == "configs3" else np.arange(n, dtype=np.uint32)
        ev["kind"][0::2] = 2; ev["kind"][1::2] = 3                       # Reading, then the Burst
        ev["channel"][0::2] = ch; ev["channel"][1::2] = ch
        ev["sample_counter"][0::2] = 1000 * r + np.arange(n); ev["sample_counter"][1::2] = 1000 * r + np.arange(n) + 500
        text = np.frombuffer(b"ZCZC-PLUMBING-%03d-" % r, dtype=np.uint8)
        ev["len"][1::2] = len(text)
        ev["bytes"][1::2, :len(text)] = text
        return ev, ch

    mine, _ = fabricate(rank)
    got = None
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        recs = sd.pack_burst_events(mine, first_ch, zero_padded=True)
        got = sd.gather_records(recs, torch.device("cpu"))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per = torch.tensor([float(rank), float(per_rank), float(len(recs)), elapsed], dtype=torch.float64)
    if world > 1:
        allr = [torch.zeros_like(per) for _ in range(world)]
        dist.all_gather(allr, per)
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    else:
        allr = [per]
    if rank == 0:
        # what arrived: every rank's records, in rank order, with global channel numbers and the payload untouched
        want = []
        for r in range(world):
            ev, ch = fabricate(r)
            want.append(sd.pack_burst_events(ev, r * C, zero_padded=True))
        want = np.concatenate(want)
        intact = got is not None and got.shape == want.shape and bool((got == want).all())
        print(json.dumps({"metric": "plumbing (no demodulation)", "value": 0.0, "unit": "Msamples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps + args.warmup, 1) * 1e3, 3),
                          "scaling": "weak", "data": "fabricated link events",
                          "config": {"workload": "plumbing" + (" at the size of configs[3]" if args.workload == "configs3" else ""), "channels_per_gpu": C,
                                     "bursts_gathered_last_step": int(len(got)) if got is not None else 0,
                                     "gathered_records_intact": intact,
                                     "record_bytes_per_rank_per_step": int(recs.nbytes),
                                     "first_channels": [r * C for r in range(world)]},
                          "ranks": {"ranks_seen": len(allr), "rank_ids": [int(t[0]) for t in allr],
                                    "bursts_per_rank": [int(t[2]) for t in allr],
                                    "first_channel_per_rank": [r * C for r in range(world)]}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def packed_bursts(rx, first_channel=0, _buf={}):
    """The gather of the single-rank blocks: the queued bursts packed into a buffer this process keeps (a fresh 40 MB
    array per step at 131 072 channels costs more in page faults than the copy).  Returns the number of bursts."""
    import numpy as np
    need = rx.pending_events()
    buf = _buf.get("a")
    if buf is None or buf.shape[0] < need:
        buf = _buf["a"] = np.empty((max(need, 1) * 5 // 4 + 16, 304), dtype=np.uint8)
    return len(rx.pack_bursts_np(first_channel, out=buf))


PREHEAT_MS = [350.0]      # --preheat-ms
AGREE = [lambda v: v]     # N ranks: the maximum of v over the ranks (main() sets it)


def run_steps(sa, rx, x, T, stream, steps, warmup, gather, barrier, layout=0, want_steady=False):
    """W untimed + K timed passes.  Each pass launches one batch; the library collects the
    previous batch's event log (copy back, ordering, transport layer) while the new launch
    runs, so a pass consumes the events of the batch before it and the last batch is
    drained after the loop -- inside the timed region.  Returns (elapsed_s, mean kernel ms,
    events of the first pass, bursts gathered from the last drained batch)."""
    import numpy as np
    import torch
    kernel_ms = []
    demod_ms = []
    first = []
    keep_first = [True]
    last_bursts = [0]

    def consume():
        # non-blocking: what the host already has.  The queue holds compact records; a consumer that only gathers the
        # bursts (the bench's job in every pass but the first) never has them made into 328-byte events.
        n_ev = rx.pending_events()
        if n_ev:
            kernel_ms.append(rx.last_kernel_ms())
            demod_ms.append(rx.last_demod_kernel_ms())
            if keep_first[0]:
                # (a harvest brings a whole launch: the first events to arrive are the first pass's, all of them -- no later
                # pass is materialised, whatever --warmup and --preheat-ms are)
                first.append(rx.peek_events_np().copy())
                keep_first[0] = False
            last_bursts[0] = gather(rx)    # (copies the burst records out of the queue)
            rx.drop_events(n_ev)

    n_launches = [0]

    def one_pass():
        rx.process_device_ptr(x.data_ptr(), T, layout, stream)
        n_launches[0] += 1
        consume()

    def drain():
        rx.sync()
        consume()

    # Warm-up: W untimed passes.  The FIRST is drained by itself: draining materialises its events for the contract checks (a
    # 35 MB view at configs[1]) -- milliseconds of host work in which the GPU idles -- and the library lets that view go at its
    # next harvest (2.9 ms of page-table work): both used to fall into the first timed passes, ~0.2 ms per step over 20 steps.
    # The others follow behind that (and behind the pre-heat), so the timed region starts on a busy GPU and a clean queue.
    #
    # Pre-heat (--preheat-ms, default 350; untimed, reported as `preheat_ms`): the SMU takes ~300 ms of load to bring an idle
    # MI355X to its sustained clock -- from idle the headline's demodulation kernel runs 1.85-2.04 ms and settles at 1.66-1.68 only
    # ~15 launches later, i.e. the 20 timed steps of a block that follows seconds of host-side contract checks measured the
    # ramp, not the stream (tools/ramp_probe.py: with 300 ms of load just before, the FIRST launch takes 1.63-1.70 ms).  So after
    # the first warm-up pass the same passes run untimed until the time is up; they are steps of the stream like any other
    # (their events are consumed), W more untimed passes than --warmup asks for by count, none inside the timed region.
    if warmup >= 1:
        one_pass()
        drain()
    if first:
        keep_first[0] = False
    if PREHEAT_MS[0] > 0.0:
        # (eight passes at a time, then the ranks agree on the time that has passed: a pass of an N-rank run holds a collective,
        # so every rank must make the same number of them)
        t_pre = time.perf_counter()
        while True:
            for _ in range(8):
                one_pass()
            if AGREE[0]((time.perf_counter() - t_pre) * 1e3) >= PREHEAT_MS[0]:
                break
    for _ in range(max(warmup - 1, 0)):
        one_pass()
    # No drain here: the last untimed launch is collected by the FIRST TIMED pass, the way every pass of a stream collects
    # the launch before it -- the timed region then holds K launches and K + 1 harvests (one more than it owes) and begins
    # on a GPU that has been idle for a synchronisation, not for a harvest (whose ~2 ms of idling cost the first timed
    # launches 5-15 %).  That launch's kernel time is not one of the K.
    n_untimed_in_flight = 1 if n_launches[0] > (1 if warmup >= 1 else 0) else 0
    kernel_ms.clear()
    demod_ms.clear()
    import gc
    gc.collect()
    gc.disable()             # (a generation-2 collection of this process's objects inside a 36 ms timed region is milliseconds)
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
    gc.enable()
    if os.environ.get("SAME_BENCH_DEBUG"):
        sys.stderr.write("per-pass wall ms (last = drain): " + " ".join(f"{1e3 * (b - a):.2f}" for a, b in zip([t0] + marks[:-1], marks)) + "\n")
        sys.stderr.write("kernel ms per launch: " + " ".join(f"{k:.3f}" for k in kernel_ms) + " | demod alone: " + " ".join(f"{k:.3f}" for k in demod_ms) + "\n")
    first_ev = np.concatenate(first) if first else np.zeros(0, dtype=sa.receiver.EVENT_DTYPE)
    first_ev = first_ev[first_ev["sample_counter"] <= T]     # the very first pass only
    del kernel_ms[:n_untimed_in_flight], demod_ms[:n_untimed_in_flight]      # (the untimed launch the first timed pass collected)
    k_mean = sum(kernel_ms) / max(len(kernel_ms), 1)
    run_steps.last_demod_ms = sum(demod_ms) / max(len(demod_ms), 1)      # (the demodulation kernel alone, for the roofline block's note)
    steady = None
    if want_steady:
        # one more pass, untimed, on the state the timed passes left: what a steady-state step delivers
        t_lo = rx.input_sample_counter()
        rx.process_device_ptr(x.data_ptr(), T, layout, stream)
        n_launches[0] += 1
        rx.sync()
        ev = rx.peek_events_np()
        steady = ev[(ev["sample_counter"] > t_lo)].copy()
        steady["sample_counter"] -= t_lo
        rx.drop_events(len(ev))
    run_steps.last_launches = n_launches[0]      # (warm-up, pre-heat, timed and the steady pass: tools/profile_collect.py sorts a trace's launches by block with it)
    return elapsed, k_mean, first_ev, last_bursts[0], steady


def tp_contract(sa, ev_strict, ev_tp, C, seed, what="first pass", t_end=None, rate=22050):
    """The time-parallel / relaxed contract on one pass of this run, against strict mode's same pass: per channel the
    same number of bursts, every transmitted byte of every burst equal (the header, or NNNN; bytes decoded after the
    carrier stops are not compared), the same transport messages in the same order.  t_end (a stream that goes on past the
    events compared -- several calls on carried state): bursts and messages within 3 symbols of the end may be reported
    by one mode now and by the other with the next call, and are left out.  Returns (ok, note)."""
    import numpy as np
    tol = 3.0 * rate / 520.83

    # (with t_end the first 64 symbols are left out as well, and bursts whose Reading is not among the events: a comparison
    # window that opens in the middle of a burst holds no transmission to compare)
    head = 64.0 * rate / 520.83

    def per_channel(ev, kind_lo, kind_hi, lo, hi):
        e = ev[(ev["kind"] >= kind_lo) & (ev["kind"] <= kind_hi)]
        if t_end is not None:
            e = e[(e["sample_counter"] > lo) & (e["sample_counter"] < t_end - hi)]
        first = np.searchsorted(e["channel"], np.arange(C + 1))
        return e, first

    def whole_bursts(ev):
        """the Burst events of a steady-state pass whose Reading lies in the same pass (a burst the previous pass's end cut in
        two is not a transmission: what is reported of it, and when, is decoded from a discontinuity)"""
        if t_end is None:
            return ev[ev["kind"] == 3]
        link = ev[ev["kind"] <= 3]
        kind, ch = link["kind"], link["channel"]
        keep = np.zeros(len(link), dtype=bool)
        reading_ch, have = -1, False
        for i in range(len(link)):
            if ch[i] != reading_ch:
                reading_ch, have = ch[i], False
            if kind[i] == 2:
                have = True
            elif kind[i] == 3:
                keep[i] = have
                have = False
        return link[keep]

    bs, fs = per_channel(whole_bursts(ev_strict), 3, 3, head, tol)
    bt, ft = per_channel(whole_bursts(ev_tp), 3, 3, head, tol)
    ms, gs = per_channel(ev_strict, 18, 20, head, 8 * tol)
    mt, gt = per_channel(ev_tp, 18, 20, head, 8 * tol)
    bad_count = bad_payload = bad_msg = n_bursts = 0
    for c in range(C):
        a, b = bs[fs[c]:fs[c + 1]], bt[ft[c]:ft[c + 1]]
        n_bursts += len(a)
        if len(a) != len(b):
            # a burst within the tolerance of the margin itself may fall on either side of it
            if t_end is not None and abs(len(a) - len(b)) == 1:
                longer = a if len(a) > len(b) else b
                edge = np.minimum(longer["sample_counter"] - head, t_end - tol - longer["sample_counter"]).min()
                if edge < 2 * tol:
                    continue
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
            if t_end is None:
                bad_msg += 1
            else:
                bad_msg += abs(len(ma) - len(mb)) > 1 or (len(ma) == len(mb))
    ok = bad_count == 0 and bad_payload == 0 and bad_msg == 0 and n_bursts > 0
    return ok, (f"{what} vs strict mode, {C} channels, {n_bursts} bursts: {bad_count} channels with a different burst count, "
                f"{bad_payload} bursts with a different payload, {bad_msg} channels with different transport messages -> "
                f"{'OK' if ok else 'VIOLATED'}")


def continuous_stream_contract(sa, C, T, rate, seed, device, env, layout, **kw):
    """State carried from call to call, checked where it means something: the bench's timed steps feed the same buffer again
    and again (a discontinuity at every step boundary, in the middle of a burst on many channels), so this check runs the
    mode and strict mode over ONE continuous stream of 3 T samples handed over in three calls of T -- bursts straddle the
    call boundaries and are delivered by the next call -- and holds the whole stream's bursts and messages against strict
    mode's, per channel.  Returns (ok, note)."""
    import numpy as np
    import torch
    x3 = sa.synth_afsk(C, 3 * T, rate, seed=seed, device=device)
    out = []
    for e, k, lay in ((dict(), dict(), 0), (env, kw, layout)):
        saved = {n: os.environ.get(n) for n in e}
        os.environ.update(e)
        try:
            r = sa.SameReceiverBuilder(rate).build_batch(C, device=device, **k)
        finally:
            for n, v in saved.items():
                if v is None:
                    os.environ.pop(n, None)
                else:
                    os.environ[n] = v
        for i in range(3):
            part = x3[i * T:(i + 1) * T]
            part = part.t().contiguous() if lay else part.contiguous()
            r.process_tensor(part, layout=lay)
        r.sync()
        ev = r.poll_events_np()
        out.append(ev[np.lexsort((np.arange(len(ev)), ev["channel"]))])
        del r
    del x3
    torch.cuda.empty_cache()
    return tp_contract(sa, out[0], out[1], C, seed, "continuous stream of 3 steps, state carried from call to call", t_end=3 * T, rate=rate)


def oracle_contract(sa, ob, cfg, x_cols, ev, chk, seed):
    """The headline mode's own first pass against the ORACLE on `chk` channels, in the form of its contract: strict mode
    -> every event equal (kind, sample counter, bytes); time-parallel / relaxed -> per channel the same bursts with equal
    transmitted bytes and the same transport messages."""
    import numpy as np
    exact = bursts_ok = True
    for c in range(chk):
        mine = ev[ev["channel"] == c]
        got = [(int(r["kind"]), int(r["sample_counter"]), r["bytes"][: min(int(r["len"]), 288)].tobytes()) for r in mine]
        ref = [e.as_tuple() for e in ob.Receiver(cfg).run(np.ascontiguousarray(x_cols[:, c]))]
        exact &= got == ref
        pay = sa.synth_payload(seed, c)
        gb = [g[2] for g in got if g[0] == 3]
        rb = [r[2] for r in ref if r[0] == 3]
        cut = lambda b: b[: (len(pay) if b[:4] == pay[:4] else 4)]
        bursts_ok &= [cut(b) for b in gb] == [cut(b) for b in rb]
        bursts_ok &= [(g[0], g[2]) for g in got if g[0] >= 18] == [(r[0], r[2]) for r in ref if r[0] >= 18]
    return exact, bursts_ok


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if args.plumbing:
        return plumbing(args)
    PREHEAT_MS[0] = max(0.0, float(args.preheat_ms))
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
    # None = the library's own non-blocking stream.  (The input was produced on torch's stream and
    # synchronised above; x stays alive for the whole run.  On the legacy null stream the runtime holds
    # a launch enqueued behind a running kernel until the next API call, which would serialise launch
    # k+1 with harvest k.)
    stream = None

    def gather(rx_):
        # the step's one collective: every rank's burst records to rank 0 (RCCL; packed by the library)
        if not distributed:
            return packed_bursts(rx_, first_ch)
        # (packed straight into the pinned buffer the gather sends from: no fresh 7 MB array and no second copy per step)
        recs = rx_.pack_bursts_np(first_ch, out=sd.pinned_send_rows(rx_.pending_events()) if dev.type == "cuda" else None)
        # (wait=False: the gathered records land in rank 0's pinned host buffer on a side stream -- inside the timed region, which
        # ends with a device-wide synchronisation -- instead of on rank 0's step, which every rank's next all_gather waits for)
        got = sd.gather_records(recs, dev, wait=False)
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

    AGREE[0] = max_over_ranks
    seed = 20260000 + rank

    def run_mode(env, layout=0, xin=None, **kw):
        """One receiver configuration over the same input: (elapsed, kernel ms, first pass, bursts gathered, steady pass, receiver facts)."""
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            r = sa.SameReceiverBuilder(args.rate).build_batch(C, device=local_rank, **kw)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        r.set_kernel_timing(True)
        el, kms, first_ev, nb, steady = run_steps(sa, r, x if xin is None else xin, T, stream, args.steps, args.warmup, gather, barrier,
                                                  layout=layout, want_steady=True)
        facts = {"kernel": r.kernel_name(), "chunks": int(r.time_parallel_chunks()), "per_channel": bool(r.time_parallel_per_channel()),
                 "demod_ms": run_steps.last_demod_ms, "launches": run_steps.last_launches}
        del r
        return max_over_ranks(el), kms, first_ev, nb, steady, facts

    # strict mode (bit-exact) first: every other mode of this run is held against its events
    elapsed, k_ms, first, n_bursts, steady, facts = run_mode({})
    runs = {"strict": (elapsed, k_ms, first, n_bursts, steady, facts)}
    kernel_name = facts["kernel"]
    tp_possible = args.rate == 22050 or args.rate in (44100, 48000)
    if tp_possible:
        # time-parallel mode (relaxed arithmetic inside the chunks unless SAME_RELAXED=0):
        # (a) on the same time-major buffer: one row offset per workgroup, uniform chunk boundaries, chunks run on until idle
        runs["time_parallel_time_major"] = run_mode({}, time_parallel=True)
        # (b) on a channel-major copy (every channel a contiguous stream, what a capture front end that delivers per-channel
        # buffers hands over): chunk boundaries per channel at idle instants, no run-on
        xc = x.t().contiguous()
        torch.cuda.synchronize()
        runs["time_parallel"] = run_mode({}, layout=sa.LAYOUT_CHANNEL_MAJOR, xin=xc, time_parallel=True)
        # (c) the same with strict arithmetic inside every chunk (round 2's form of the mode)
        runs["time_parallel_strict_chunks"] = run_mode({"SAME_RELAXED": "0"}, layout=sa.LAYOUT_CHANNEL_MAJOR, xin=xc, time_parallel=True)
        del xc
        torch.cuda.empty_cache()
    # relaxed arithmetic on an ordinary launch (no cut in time)
    runs["relaxed"] = run_mode({}, relaxed=True)

    mode_build = {"time_parallel_time_major": ({}, 0, {"time_parallel": True}), "time_parallel": ({}, 1, {"time_parallel": True}),
                  "time_parallel_strict_chunks": ({"SAME_RELAXED": "0"}, 1, {"time_parallel": True}), "relaxed": ({}, 0, {"relaxed": True})}
    contracts = {}
    for name, (el, kms, fe, nb, st, fc) in runs.items():
        if name == "strict":
            continue
        ok1, note1 = tp_contract(sa, first, fe, C, seed, "first pass", t_end=T, rate=args.rate)
        env_, lay_, kw_ = mode_build[name]
        ok2, note2 = (True, "skipped (--no-carried-state-check)") if args.no_carried_state_check else \
            continuous_stream_contract(sa, C, T, args.rate, seed + 1000, local_rank, env_, lay_, **kw_)
        contracts[name] = (ok1 and ok2, note1, note2)
    if distributed:
        names = sorted(contracts)
        t = torch.tensor([1.0 if contracts[n][0] else 0.0 for n in names], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        for i, n in enumerate(names):
            contracts[n] = (bool(t[i].item() > 0.5),) + contracts[n][1:]

    # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    # profiles/r01_fetch_calibration.txt); only valid for the workload it was collected on
    def pmc_traffic(kind):
        if args.traffic is not None:
            return args.traffic
        for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    tj = json.load(f)
                for ent in (tj if isinstance(tj, list) else [tj]):
                    if ent.get("workload") == f"{C} ch x {T} samples" and ent.get("mode", "strict") == kind:
                        return ent["hbm_bytes_per_launch"]
            except Exception:
                pass
        return None

    def pmc_issue(kind, workload=None):
        """What limits the kernel when it is not HBM (it is not): wavefront instructions per 64-column sample by kind and the fraction
        of a wavefront's cycles in which it issues a vector one, from the round's SQ counter pass over this same command (tools/profile_round.sh,
        profiles/r05_pmc_instruction_mix.txt); null when no pass has been filed."""
        for fname in ("r06_issue.json", "r05_issue.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    for ent in json.load(f):
                        if ent.get("workload") == (workload or f"{C} ch x {T} samples") and ent.get("mode") == kind:
                            return {"kind": "instruction_issue_per_wavefront", "valu_per_workgroup_sample": ent["valu_per_workgroup_sample"],
                                    "salu_per_workgroup_sample": ent.get("salu_per_workgroup_sample"), "lds_per_workgroup_sample": ent.get("lds_per_workgroup_sample"),
                                    "valu_issue_fraction_of_wave_cycles": ent["valu_issue_fraction_of_wave_cycles"],
                                    "note": "every role-wavefront of the symbol-paced pipeline is self-bound (profiles/r05_cycle_attribution.txt, light timeline): a launch is "
                                            "the longest role's instructions of ANY kind; one wavefront issues an instruction per ~2.6 ns alone on its SIMD, per ~3.6 ns "
                                            "with two neighbours (profiles/r06_ubench_simd.txt, DESIGN.md 8)",
                                    "budget_for_60pct_hbm": "<= ~36 instructions of all kinds per 64-column sample at the measured issue rate (58 in round 5's count); "
                                                            "the SMU shader clock under this kernel is 2.39 GHz (profiles/r05_clock_power.txt)",
                                    "source": ent["source"]}
            except Exception:
                pass
        return None

    notes = {
        "strict": "bit-exact; latency-bound serial streams (DESIGN.md 4.4, 4.4b)",
        "time_parallel": "channel-major input x[channel][t]; time chunks per channel = state columns side by side, chunk boundaries per channel at idle "
                         "instants (device-side energy scout + planner + sort: inside kernel_ms, which is what `achieved` is priced on, as far as they "
                         "are not hidden under the previous launch's tail on the plan stream; demod_kernel_alone_ms is the demodulation kernel by "
                         "itself, what rocprofv3 lists), 8 pieces per channel = one round of workgroups (10 with SAME_RELAXED=0), relaxed arithmetic inside the chunks (the symbol-paced pipeline, same_kernels_sym.hip; "
                         "DESIGN.md 4.6, 4.7)",
        "time_parallel_time_major": "time-major input; uniform chunk boundaries (one row offset per workgroup keeps the loads coalesced), chunks run on "
                                    "until idle, relaxed arithmetic inside the chunks; kernel_ms includes the state column copies",
        "time_parallel_strict_chunks": "as time_parallel with strict arithmetic inside every chunk (SAME_RELAXED=0: round 2's form of the mode)",
        "relaxed": "ordinary launch (no cut in time), relaxed arithmetic (SAME_BATCH_RELAXED): the symbol-paced pipeline (same_kernels_sym.hip)",
    }
    layouts = {"time_parallel": "channel-major x[channel][t]", "time_parallel_strict_chunks": "channel-major x[channel][t]"}
    modes = {}
    for name, (el, kms, fe, nb, st, fc) in runs.items():
        ach = 4.0 * C * T / (kms * 1e-3) / 1e9
        modes[name] = {
            "value": round(C * T * world * args.steps / el / 1e6, 2), "unit": "Msamples/s",
            "ms_per_step": round(el / args.steps * 1e3, 3), "bursts_gathered_last_step": int(nb),
            "bursts_pass_after_the_timed_ones_rank0": int((st["kind"] == 3).sum()),
            "kernel": fc["kernel"], "launches": fc["launches"], "layout": layouts.get(name, "time-major x[t][channel]"),
            "roofline": {"bound": "hbm", "binds_in_practice": "instruction_issue", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(name), "kernel_ms": round(kms, 4),
                         "demod_kernel_alone_ms": round(fc["demod_ms"], 4),
                         "algorithmic_bytes_per_launch": 4 * C * T,
                         # `bound` names the roofline `frac` is priced against (the north star's and BASELINE.json's metric: HBM);
                         # `binds_in_practice` / `limiter` are what the counters say limits the kernel -- no block of this line is HBM-bound
                         "limiter": pmc_issue(name), "note": notes[name]},
        }
        if name != "strict":
            modes[name].update(chunks=fc["chunks"], per_channel_boundaries=fc["per_channel"],
                               contract=contracts[name][1], contract_carried_state=contracts[name][2])
    if args.mode != "auto":
        headline = args.mode if args.mode in modes else "strict"
    else:
        # the fastest mode whose contract holds on this run's own first pass and on a continuous stream of three calls (every channel, every rank)
        ok = [m for m in modes if m != "strict" and contracts[m][0] and (runs[m][5]["chunks"] > 1 or m == "relaxed")]
        # (by kernel rate, not by whole-step value: two modes a few per cent apart in the kernel would otherwise swap places
        # with the host's jitter from run to run; `value` is then that mode's whole step)
        headline = max(ok, key=lambda m: modes[m]["roofline"]["achieved"]) if ok else "strict"
    hb = modes[headline]
    out = {
        "metric": "Msamples/s demodulated (batched 22.05 kHz channels) + % HBM roofline, 1/8 GPU",
        "value": hb["value"],
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        # (untimed passes of the same step before every block, so that the timed steps run at the sustained GPU clock: run_steps)
        "preheat_ms": args.preheat_ms,
        "ms_per_step": hb["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{C} synthetic {args.rate / 1000:g} kHz AFSK channels per GPU, f32, "
                        f"{args.seconds:g} s ({T} samples) per channel per step (BASELINE.json {'configs[1]' if args.workload == 'configs1' else 'configs[3]: 262 144 channels over 8 GPUs'})",
            "channels_per_gpu": C, "samples_per_channel": T, "input_rate": args.rate,
            "layout": hb["layout"], "kernel": hb["kernel"],
            "mode": headline,
            "parity": ("relaxed contract (include/same_rx.h; tests/test_time_parallel.py, tests/test_relaxed.py): every burst's transmitted bytes "
                       "and every transport message equal to strict mode's, link events within 2 symbols, soft symbols within 0.05; "
                       "strict mode is bit-exact to the oracle and stays in `modes`"
                       if headline != "strict" else "bit-exact (strict op order)"),
            "input_duty_cycle": "bursts with 1 s gaps (SURVEY.md 8d): the time-parallel cut needs idle instants; a channel that is never quiet "
                                "degrades to forced cuts that run on (tests/test_time_parallel.py)",
            "bursts_gathered_last_step": hb["bursts_gathered_last_step"], "events_first_step_rank0": int(len(first)),
        },
        "roofline": hb["roofline"],
        "modes": modes,
    }
    # what every rank saw (the driver's multi-GPU runs are checkable from rank 0's line alone)
    per_rank = torch.tensor([runs[headline][1], float(runs[headline][3]) if distributed else float(hb["bursts_gathered_last_step"]),
                             float((runs[headline][4]["kind"] == 3).sum()), float(int(os.environ.get("SAME_HOST_THREADS", "0") or 0))],
                            dtype=torch.float64, device=dev)
    if distributed:
        allr = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(allr, per_rank)
    else:
        allr = [per_rank]
    out["ranks"] = {"ranks_seen": len(allr), "kernel_ms_per_rank": [round(float(t[0]), 4) for t in allr],
                    "bursts_pass_after_the_timed_ones_per_rank": [int(t[2]) for t in allr],
                    "first_channel_per_rank": [r * C for r in range(world)],
                    "harvest_threads_per_rank": "min(32, max(16, hardware threads / 8)) unless SAME_HOST_THREADS is set (same_batch.cpp harvest_slot)"}

    if rank == 0 and world == 1:
        from oracle import binding as ob
        cfg = ob.default_config(args.rate)
        xs = None
        if args.check and args.warmup + args.steps > 0:
            # parity spot-checks of this very run (first passes start from fresh state): the strict receiver's events against
            # the oracle event for event, and the headline mode's own events against the oracle in the form of its contract
            chk = min(args.check, C)
            xs = x[:, :max(chk, min(args.cpu_channels, C))].contiguous().cpu().numpy()
            exact, _ = oracle_contract(sa, ob, cfg, xs, first, chk, seed)
            out["config"]["parity_check"] = f"strict receiver, {chk} channels of this run vs oracle, every event: {'OK' if exact else 'MISMATCH'}"
            if headline != "strict":
                _, good = oracle_contract(sa, ob, cfg, xs, runs[headline][2], chk, seed)
                out["config"]["parity_check_headline_mode"] = (f"{headline} receiver, {chk} channels of this run vs oracle, bursts' transmitted bytes and "
                                                                f"transport messages: {'OK' if good else 'MISMATCH'}")
                exact &= good
            if not exact:
                out["config"]["parity"] = "MISMATCH"
        if not args.no_cpu_baseline:
            # One SameReceiver per channel over a CONTIGUOUS stream (the sample is transposed to
            # channel-major on the host first), one worker pinned per physical core, repeated until the
            # run is long enough to time arithmetic rather than thread start-up.
            cc = min(args.cpu_channels, C)
            if xs is None or xs.shape[1] < cc:
                xs = x[:, :cc].contiguous().cpu().numpy()
            xcpu = np.ascontiguousarray(xs[:, :cc].T)          # [channel][time]
            cores_all = ob.physical_cores()
            logical = len(os.sched_getaffinity(0))
            # A container may be granted fewer CPUs than it can see (cgroup v2 cpu.max = quota period): 128 pinned threads
            # on a 16-CPU quota are throttled to 16 CPUs' worth of time -- round 3's "10 Msample/s per core against 46 for
            # the same code on one core".  One worker per physical core, as many as the quota covers.
            quota = None
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                quota = None if q == "max" else float(q) / float(per)
            except (OSError, ValueError):
                pass
            n_use = len(cores_all) if quota is None else max(1, min(len(cores_all), int(quota)))
            cores = cores_all[:n_use]
            # one core by itself first: what the code does when nothing competes with it
            x1 = xcpu[:8]
            t1 = time.perf_counter()
            ob.batch_run_channel_major(cfg, x1, cpus=cores[:1], reps=1)
            r1 = max(1, int(np.ceil(1.0 / max(time.perf_counter() - t1, 1e-3))))
            t1 = time.perf_counter()
            ob.batch_run_channel_major(cfg, x1, cpus=cores[:1], reps=r1)
            alone = x1.shape[0] * T * r1 / (time.perf_counter() - t1) / 1e6
            ob.batch_run_channel_major(cfg, xcpu, cpus=cores, reps=1)      # warm pass: threads, page faults, caches
            t1 = time.perf_counter()
            ob.batch_run_channel_major(cfg, xcpu, cpus=cores, reps=1)
            probe = time.perf_counter() - t1
            reps = max(1, int(np.ceil(1.15 * args.cpu_seconds / max(probe, 1e-3))))      # (a repetition inside one call is ~10 % faster than the probe call)
            t1 = time.perf_counter()
            ob.batch_run_channel_major(cfg, xcpu, cpus=cores, reps=reps)
            dt = time.perf_counter() - t1
            if dt < args.cpu_seconds:              # (the estimate fell short: once more, scaled up, and that run is the one reported)
                reps = int(np.ceil(reps * 1.1 * args.cpu_seconds / max(dt, 1e-3)))
                t1 = time.perf_counter()
                ob.batch_run_channel_major(cfg, xcpu, cpus=cores, reps=reps)
                dt = time.perf_counter() - t1
            rate = cc * T * reps / dt / 1e6
            out["cpu_baseline"] = {
                "value": round(rate, 2), "unit": "Msamples/s", "cores": len(cores), "kind": "port",
                "per_core_all": round(rate / len(cores), 2), "per_core_alone": round(alone, 2),
                "physical_cores_visible": len(cores_all), "logical_cpus": logical,
                "cgroup_cpu_quota": quota,
                "sample": f"first {cc} channels x {T} samples of the same synthetic input, channel-major on the host, "
                          f"{reps} repetition(s), link layer only, {dt:.2f} s wall on {len(cores)} threads pinned one per "
                          f"physical core (scalar C restatement of sameold 0.6.0; the Rust reference cannot be built in this image); "
                          + (f"the container's CPU quota is {quota:g} CPUs of the {len(cores_all)} physical cores it sees, so {len(cores)} "
                             f"workers were run -- more would only be throttled; " if quota is not None and quota < len(cores_all) else "")
                          + f"one worker alone on one core: {alone:.1f} Msample/s, each of the {len(cores)} together: {rate / len(cores):.1f}",
            }
            del xcpu
        del xs
        if not args.no_scaled and args.workload == "configs1":
            # the per-GPU shard of configs[3]: 32768 channels (2 s per step to bound memory), strict and relaxed
            del x
            torch.cuda.empty_cache()
            Cs, Ts = args.scaled_channels, int(args.rate * 2)
            x2 = sa.synth_afsk(Cs, Ts, args.rate, seed=777, device=local_rank)
            n2 = args.steps                   # every block of the line the same number of steps (round 5 doubled one block's: its ratio improved partly by the denominator)
            out["scaled"] = {"workload": f"{Cs} channels x {Ts} samples per step (per-GPU shard of BASELINE.json configs[3])"}
            ev_strict = None
            for label, kw in (("strict", {}), ("relaxed", {"relaxed": True})):
                rx2 = sa.SameReceiverBuilder(args.rate).build_batch(Cs, device=local_rank, **kw)
                rx2.set_kernel_timing(True)
                e2, k2, f2, _, _ = run_steps(sa, rx2, x2, Ts, stream, n2, 2, packed_bursts, lambda: None)
                a2 = 4.0 * Cs * Ts / (k2 * 1e-3) / 1e9
                blk = {"value": round(Cs * Ts * n2 / e2 / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(k2, 4), "steps": n2,
                       "ms_per_step": round(e2 / n2 * 1e3, 3), "kernel": rx2.kernel_name(),
                       "roofline": {"bound": "hbm", "binds_in_practice": "instruction_issue", "achieved": round(a2, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a2 / HBM_PEAK_GBS, 5),
                                    "limiter": pmc_issue(label, f"{Cs} ch x {Ts} samples")}}
                if label == "strict":
                    ev_strict = f2
                    out["scaled"].update(blk)                       # (the strict figures stay where round 1 and 2 put them)
                else:
                    blk["contract"] = tp_contract(sa, ev_strict, f2, Cs, 777, "first pass", t_end=Ts, rate=args.rate)[1]
                    out["scaled"]["relaxed"] = blk
                del rx2
            del x2, ev_strict
            torch.cuda.empty_cache()
            if not args.no_scaled_big:
                # four times the machine: 131072 channels x 2 s, strict (one-wavefront kernel) and relaxed (the symbol-paced pipeline in rounds), with the
                # relaxed pass held against the strict one (first pass, every channel)
                try:
                    Cb = 131072
                    xb = sa.synth_afsk(Cb, Ts, args.rate, seed=780, device=local_rank)
                    out["scaled_big"] = {"workload": f"{Cb} channels x {Ts} samples per step"}
                    ev_b = None
                    # (the last launch is drained inside the timed region, one whole harvest that nothing overlaps: at 131 072 channels that
                    # is ~7 ms -- a twentieth of this block's figure at the default 20 steps)
                    nb = args.steps
                    for label, kw in (("strict", {}), ("relaxed", {"relaxed": True})):
                        rxb = sa.SameReceiverBuilder(args.rate).build_batch(Cb, device=local_rank, **kw)
                        rxb.set_kernel_timing(True)
                        eb, kb, fb_, _, _ = run_steps(sa, rxb, xb, Ts, stream, nb, 1, packed_bursts, lambda: None)
                        ab = 4.0 * Cb * Ts / (kb * 1e-3) / 1e9
                        blk = {"value": round(Cb * Ts * nb / eb / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(kb, 4), "steps": nb,
                               "ms_per_step": round(eb / nb * 1e3, 3), "kernel": rxb.kernel_name(),
                               "roofline": {"bound": "hbm", "binds_in_practice": "instruction_issue", "achieved": round(ab, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ab / HBM_PEAK_GBS, 5)}}
                        if label == "strict":
                            ev_b = fb_
                            out["scaled_big"].update(blk)
                        else:
                            blk["contract"] = tp_contract(sa, ev_b, fb_, Cb, 780, "first pass", t_end=Ts, rate=args.rate)[1]
                            out["scaled_big"]["relaxed"] = blk
                        del rxb
                    del xb, ev_b
                except Exception as exc:      # (a smaller device memory: the block is extra evidence, never the headline)
                    out["scaled_big"] = {"skipped": repr(exc)[:200]}
                torch.cuda.empty_cache()
            if args.scaled_long:
                # the same shard with 10 s per step (29 GB resident): long enough for the time-parallel cut
                # -- 4 pieces per channel = 131 072 state columns on the relaxed kernel of same_kernels_relaxed.hip
                try:
                    Tl = int(args.rate * 10)
                    Tl -= Tl % 1260
                    x4 = sa.synth_afsk(Cs, Tl, args.rate, seed=779, device=local_rank)
                    x4c = x4.t().contiguous()
                    del x4
                    torch.cuda.synchronize()
                    rx4 = sa.SameReceiverBuilder(args.rate).build_batch(Cs, device=local_rank, time_parallel=True)
                    rx4.set_kernel_timing(True)
                    n4 = 3
                    e4, k4, f4, _, _ = run_steps(sa, rx4, x4c, Tl, stream, n4, 1, packed_bursts, lambda: None,
                                                 layout=sa.LAYOUT_CHANNEL_MAJOR)
                    a4 = 4.0 * Cs * Tl / (k4 * 1e-3) / 1e9
                    out["scaled_long"] = {
                        "workload": f"{Cs} channels x {Tl} samples per step, channel-major (per-GPU shard of configs[3] at 10 s per step), time-parallel",
                        "value": round(Cs * Tl * n4 / e4 / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(k4, 4), "steps": n4,
                        "ms_per_step": round(e4 / n4 * 1e3, 3), "kernel": rx4.kernel_name(), "chunks": int(rx4.time_parallel_chunks()),
                        "bursts_first_pass": int((f4["kind"] == 3).sum()),
                        "roofline": {"bound": "hbm", "binds_in_practice": "instruction_issue", "achieved": round(a4, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a4 / HBM_PEAK_GBS, 5)}}
                    del rx4, x4c
                except Exception as exc:      # (a smaller device memory: the block is extra evidence, never the headline)
                    out["scaled_long"] = {"skipped": repr(exc)[:200]}
                torch.cuda.empty_cache()
            # and configs[2]: 16384 channels at 48 kHz (92-tap filters; strict: the pipeline's 32-sample blocks, relaxed: the symbol-paced
            # pipeline's 72-sample steps, one group of 64 columns per CU -- round 6), 2 s per step
            C3, R3 = 16384, 48000
            T3 = R3 * 2
            x3 = sa.synth_afsk(C3, T3, R3, seed=778, device=local_rank)
            out["configs2_48k"] = {"workload": f"{C3} channels x {T3} samples per step at {R3} Hz (BASELINE.json configs[2], 2 s of its 10 s)"}
            ev3 = None
            for label, kw in (("strict", {}), ("relaxed", {"relaxed": True})):
                rx3 = sa.SameReceiverBuilder(R3).build_batch(C3, device=local_rank, **kw)
                rx3.set_kernel_timing(True)
                e3, k3, f3, _, _ = run_steps(sa, rx3, x3, T3, stream, n2, 2, packed_bursts, lambda: None)
                a3 = 4.0 * C3 * T3 / (k3 * 1e-3) / 1e9
                blk = {"value": round(C3 * T3 * n2 / e3 / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(k3, 4), "steps": n2,
                       "ms_per_step": round(e3 / n2 * 1e3, 3), "kernel": rx3.kernel_name(),
                       "roofline": {"bound": "hbm", "binds_in_practice": "instruction_issue", "achieved": round(a3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a3 / HBM_PEAK_GBS, 5)}}
                if label == "strict":
                    ev3 = f3
                    out["configs2_48k"].update(blk)
                else:
                    blk["contract"] = tp_contract(sa, ev3, f3, C3, 778, "first pass", t_end=T3, rate=R3)[1]
                    out["configs2_48k"]["relaxed"] = blk
                del rx3
            del x3, ev3
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
