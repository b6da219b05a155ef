#!/usr/bin/env python3
"""The host half of a step (ordering, stitch, transport layer, queue: harvest_host in same_batch.cpp) without a device, on
recorded launches, in N processes side by side -- what N ranks of one node do to each other's harvest.

    python tools/host_step_probe.py --record DIR          (needs the GPU: writes DIR/harvest_<shape>.bin.xz)
    python tools/host_step_probe.py --ranks 8 [--threads 16] [--reps 40] [--record-dir profiles/harvest]   (CPU only)

Shapes: `shard` = 32 768 channels x 2 s, relaxed arithmetic, transport layer on (the per-GPU shard of BASELINE.json configs[3]);
`headline` = 4 096 channels x 10 s, time-parallel, channel-major (configs[1]).  A record holds what the device hands
over for ONE launch: the ordered event log, the columns' offsets, the burst pool, hand-over instants and chunk geometry."""
import argparse, ctypes, lzma, multiprocessing as mp, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = ("shard", "headline")


def record(out_dir):
    import torch
    import sameold_amd as sa
    os.makedirs(out_dir, exist_ok=True)
    for shape in SHAPES:
        raw = os.path.join(out_dir, f"harvest_{shape}.bin")
        os.environ["SAME_RECORD_HARVEST"] = raw
        if shape == "shard":
            C, T = 32768, 44100 - 44100 % 180
            x = sa.synth_afsk(C, T, 22050, seed=777); torch.cuda.synchronize()
            rx = sa.SameReceiverBuilder(22050).build_batch(C, relaxed=True)
            for _ in range(5):
                rx.process_tensor(x); rx.drop_events(rx.pending_events())
        else:
            C, T = 4096, 220500
            x = sa.synth_afsk(C, T, 22050, seed=20260000).t().contiguous(); torch.cuda.synchronize()
            rx = sa.SameReceiverBuilder(22050).build_batch(C, time_parallel=True)
            for _ in range(5):
                rx.process_tensor(x, layout=sa.LAYOUT_CHANNEL_MAJOR); rx.drop_events(rx.pending_events())
        rx.sync(); del rx, x
        os.environ.pop("SAME_RECORD_HARVEST")
        data = open(raw, "rb").read()
        with lzma.open(raw + ".xz", "wb", preset=6) as f:
            f.write(data)
        os.remove(raw)
        print(f"{shape}: {len(data)/1e6:.1f} MB recorded, {os.path.getsize(raw + '.xz')/1e6:.2f} MB compressed -> {raw}.xz", flush=True)


def lib():
    v = os.environ.get("SAME_LIB_VARIANT")          # (a measurement build: sameold_amd/build.py)
    L = ctypes.CDLL(os.path.join(ROOT, "sameold_amd", f"libsame_rx.{v}.so" if v else "libsame_rx.so"))
    L.same_debug_harvest_replay.restype = ctypes.c_long
    L.same_debug_harvest_replay.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    return L


def rank_main(rank, path, threads, reps, barrier, q):
    L = lib()
    out = (ctypes.c_double * reps)()
    barrier.wait()
    t0 = time.perf_counter()
    n = L.same_debug_harvest_replay(path.encode(), threads, reps, out)
    q.put((rank, n, list(out), time.perf_counter() - t0))


def replay(shape, rec_dir, ranks, threads, reps):
    import numpy as np
    src = os.path.join(rec_dir, f"harvest_{shape}.bin.xz")
    if not os.path.exists(src):
        print(f"{shape}: no record at {src} (python tools/host_step_probe.py --record {rec_dir} on the GPU box)")
        return
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as tmp:
        tmp.write(lzma.open(src).read())
    try:
        barrier, q = mp.Barrier(ranks), mp.Queue()
        ps = [mp.Process(target=rank_main, args=(r, tmp.name, threads, reps, barrier, q)) for r in range(ranks)]
        for p in ps: p.start()
        res = sorted(q.get() for _ in ps)
        for p in ps: p.join()
    finally:
        os.unlink(tmp.name)
    if any(r[1] < 0 for r in res):
        print(shape, "replay failed:", [r[1] for r in res]); return
    ms = np.array([r[2][2:] for r in res])          # (the first two repetitions grow the queues)
    print(f"{shape}: {ranks} rank(s) x {threads} harvest thread(s), {reps} harvests each back to back, {res[0][1]} queue records per harvest")
    print(f"  per rank mean {ms.mean(axis=1).min():.2f} .. {ms.mean(axis=1).max():.2f} ms, p90 {np.percentile(ms, 90):.2f} ms, max {ms.max():.2f} ms"
          f"  (all ranks: mean {ms.mean():.2f} ms)")
    return ms.mean()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--record", metavar="DIR")
    ap.add_argument("--record-dir", default=os.path.join(ROOT, "profiles", "harvest"))
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--threads", type=int, default=0, help="harvest threads per rank (0: 1, 4, 16 and 32 in turn)")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    if args.record:
        record(args.record)
        sys.exit(0)
    n_cpu = len(os.sched_getaffinity(0))
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if a == "max" else float(a) / float(b)
    except Exception:
        pass
    print(f"host: {n_cpu} CPUs visible, cgroup quota {quota if quota else 'none'}")
    for shape in SHAPES:
        one = replay(shape, args.record_dir, 1, 1, args.reps)
        if one is None:
            continue
        print(f"  = {one:.1f} CPU-ms of host work per launch on one thread")
        for th in ([args.threads] if args.threads else [4, 16, 32]):
            replay(shape, args.record_dir, 1, th, args.reps)
            if args.ranks > 1:
                replay(shape, args.record_dir, args.ranks, th, args.reps)
