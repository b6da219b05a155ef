"""N > 1 path on CPU: channel sharding and the burst gather over gloo, world_size 2.
(The demodulation itself needs a GPU; what is distributed is only this plumbing.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sameold_amd import distributed as sd
    first, count = sd.shard_channels(10, rank, world)
    # rank r decodes (r + 1) * 2 bursts on its own channels; rank 1 also sends none in round 2
    mine = [(first + (i % count), 1000 * rank + i, bytes([65 + rank]) * (5 + i)) for i in range((rank + 1) * 2)]
    got = sd.gather_bursts(mine, torch.device("cpu"))
    got2 = sd.gather_bursts(mine if rank == 0 else [], torch.device("cpu"))
    # the vectorised record path bench.py uses
    recs = sd.gather_records(sd.pack_bursts(mine), torch.device("cpu"))
    if recs is not None:
        assert sd.unpack_bursts(recs) == got
    else:
        assert got is None
    dist.barrier()
    q.put((rank, first, count, got, got2))
    dist.destroy_process_group()


def test_shard_channels_partition():
    from sameold_amd import distributed as sd
    for n in (1, 7, 8, 4096, 262144, 1000003):
        for w in (1, 2, 3, 8):
            spans = [sd.shard_channels(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_pack_roundtrip():
    from sameold_amd import distributed as sd
    b = [(3, 12345678901, b"ZCZC-WXR-TOR-039173+0030-1591829-KCLE/NWS-\x00\x00"), (0, 0, b""), (2 ** 31, 2 ** 40, bytes(range(256)) + b"x" * 40)]
    r = sd.unpack_bursts(sd.pack_bursts(b))
    assert r[0] == b[0] and r[1] == b[1]
    assert r[2][:2] == b[2][:2] and r[2][2] == b[2][2][:288]


def test_gather_bursts_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, first, count, got, got2 = q.get(timeout=120)
        res[rank] = (first, count, got, got2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][:2] == (0, 5) and res[1][:2] == (5, 5)
    assert res[1][2] is None and res[1][3] is None
    got = res[0][2]
    assert len(got) == 2 + 4
    assert [g for g in got if g[2][:1] == b"A"] == [(0, 0, b"AAAAA"), (1, 1, b"AAAAAA")]
    assert [g[1] for g in got if g[2][:1] == b"B"] == [1000, 1001, 1002, 1003]
    assert len(res[0][3]) == 2   # a rank with nothing to send contributes nothing


def test_montecarlo_scoring_bookkeeping():
    """Host-side scoring of the AWGN sweep (sameold_amd/montecarlo.py): first burst per trial,
    bit errors over the common bytes, intact = whole header delivered."""
    import numpy as np
    from sameold_amd import montecarlo as mc
    from sameold_amd import receiver as R
    payloads = [b"ZCZC-AAA-BBB-123456+0015-1231234-STATION -"] * 4
    ev = np.zeros(5, dtype=R.EVENT_DTYPE)

    def put(i, ch, t, data, kind=R.LINK_BURST):
        ev[i]["kind"], ev[i]["channel"], ev[i]["sample_counter"], ev[i]["len"] = kind, ch, t, len(data)
        ev[i]["bytes"][: len(data)] = np.frombuffer(data, dtype=np.uint8)

    put(0, 0, 500, payloads[0] + b"\0\0\0")                    # intact, trailing bytes ignored
    put(1, 1, 700, b"ZCZC-AAA-BBB-123456+0015-1231234-STATIOO -")   # one byte wrong: 'N'^'O' = 1 bit
    put(2, 1, 300, payloads[0][:10])                           # earlier burst of trial 1 wins: cut short
    put(3, 2, 100, b"", kind=R.LINK_SEARCHING)                 # not a burst
    put(4, 3, 900, payloads[0])
    tally = mc.new_tally(2)
    mc.score_bursts(ev, payloads, 10, 4, 2, tally)            # trials 10..13 -> grid points 0,1,0,1
    assert tally["trials"].tolist() == [2, 2]
    assert tally["detected"].tolist() == [1, 2]
    assert tally["intact"].tolist() == [1, 1]
    assert tally["bits"].tolist() == [8 * len(payloads[0]), 8 * 10 + 8 * len(payloads[0])]
    assert tally["bit_errors"].tolist() == [0, 0]
    assert tally["short_bytes"].tolist() == [0, len(payloads[0]) - 10]
    rows = mc.summarise(tally, 3.0, 2.0)
    assert rows[1]["ebn0_db"] == 5.0 and rows[0]["burst_detection_rate"] == 0.5 and rows[1]["ber"] == 0.0


def test_vectorised_burst_packing_matches_the_record_layout():
    """pack_burst_events (what bench.py gathers) == pack_bursts on the same bursts."""
    import numpy as np
    from sameold_amd import distributed as sd
    from sameold_amd import receiver as R
    ev = np.zeros(4, dtype=R.EVENT_DTYPE)
    payloads = [b"ZCZC-AAA-BBB-123456+0015-1231234-STATION -", b"NNNN", b"x" * 300, b""]
    kinds = [R.LINK_BURST, R.LINK_BURST, R.LINK_BURST, R.LINK_SEARCHING]
    for i, (p, k) in enumerate(zip(payloads, kinds)):
        ev[i]["kind"], ev[i]["channel"], ev[i]["sample_counter"], ev[i]["len"] = k, 7 + i, 1000 * (i + 1) + (1 << 33), len(p)
        n = min(len(p), 288)
        ev[i]["bytes"][:n] = np.frombuffer(p[:n], dtype=np.uint8)
        ev[i]["bytes"][n:] = 0xEE                       # stale bytes past the burst must not leak
    got = sd.pack_burst_events(ev, first_channel=4096)
    want = sd.pack_bursts([(4096 + 7 + i, 1000 * (i + 1) + (1 << 33), payloads[i][:288]) for i in range(3)])
    assert got.shape == (3, sd.RECORD_BYTES) and np.array_equal(got, want)
    assert sd.unpack_bursts(got)[1] == (4096 + 8, 2000 + (1 << 33), b"NNNN")
    assert sd.gather_records(got, None) is got      # no process group: identity


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` must launch the ranks itself (round-1 finding: --gpus was parsed and
    ignored).  Plumbing mode: gloo, fabricated burst records, no kernel -- checks the world size rank 0
    reports and that the gathered burst count comes from both ranks (3 + 4)."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing", "--steps", "2",
                        "--warmup", "1", "--channels", "16"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["bursts_gathered_last_step"] == 7
    assert out["config"]["first_channels"] == [0, 16]


def test_bench_gpus_8_plumbing_runs_the_real_pack_and_gather_path():
    """The driver's 8-GPU line, as far as it can be checked without a GPU: `bench.py --gpus 8 --workload configs3 --plumbing`
    starts eight ranks, every one packs fabricated link events with the bench's own packing (global channel numbers) and the
    step's one gather brings them to rank 0, which compares what arrived with what every rank made, record for record, and
    prints the per-rank block a real run prints."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing", "--workload", "configs3", "--channels", "2048",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8
    assert out["config"]["gathered_records_intact"] is True
    assert out["config"]["bursts_gathered_last_step"] == 8 * (3 * 2048 // 4)
    assert out["ranks"]["ranks_seen"] == 8 and out["ranks"]["rank_ids"] == list(range(8))
    assert out["ranks"]["bursts_per_rank"] == [3 * 2048 // 4] * 8
    assert out["ranks"]["first_channel_per_rank"] == [r * 2048 for r in range(8)]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
