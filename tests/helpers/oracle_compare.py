"""Whole-batch comparison of device link events with the oracle (test infrastructure).

The oracle (oracle/same_oracle.c, a scalar C restatement of the reference) runs the same samples on
all host cores through `ob.batch_run_time_major`; the device's events must equal it event for event:
kind, input sample counter, burst length and burst bytes, on EVERY channel.
"""
import os

import numpy as np


def oracle_link_events(ob, cfg, xs, threads=None):
    """xs: host array [T, n] float32 time-major -> oracle link events as a numpy array ordered by
    (channel, time); field `aux` is the channel."""
    threads = threads or len(os.sched_getaffinity(0))
    cap = 1 << 16
    while True:
        n, evs = ob.batch_run_time_major(cfg, xs, threads, cap=cap)
        if n <= cap:
            break
        cap = int(n) + 1024
    ref = np.frombuffer(evs, dtype=ob._EVENT_NP, count=n).copy()
    return ref[np.argsort(ref["aux"], kind="stable")]          # workers append in time order per channel


def assert_every_channel_matches_oracle(ob, cfg, x, ev, slab=2048, collect=None):
    """`x`: device tensor [T, C]; `ev`: the polled device events (numpy EVENT_DTYPE, ordered by channel
    then time); transport events in `ev` are ignored.  Channels go to the host in slabs to bound
    memory.  `collect`: optional list that receives the oracle's events per slab (first channel, array).
    Returns the number of link events compared."""
    n_ch = x.shape[1]
    link = ev[ev["kind"] <= 3]
    assert np.all(np.diff(link["channel"].astype(np.int64)) >= 0), "events must be ordered by channel"
    first = np.searchsorted(link["channel"], np.arange(n_ch + 1))
    for c0 in range(0, n_ch, slab):
        c1 = min(n_ch, c0 + slab)
        xs = x[:, c0:c1].contiguous().cpu().numpy()
        ref = oracle_link_events(ob, cfg, xs)
        if collect is not None:
            collect.append((c0, ref))
        mine = link[first[c0]:first[c1]]
        assert len(mine) == len(ref), f"channels {c0}..{c1}: {len(mine)} device events, oracle {len(ref)}"
        bad = np.flatnonzero((mine["channel"] - c0 != ref["aux"]) | (mine["kind"] != ref["kind"])
                             | (mine["sample_counter"] != ref["sample_counter"]) | (mine["len"] != ref["len"]))
        assert len(bad) == 0, f"first mismatch at channel {int(mine['channel'][bad[0]])}: {mine[bad[0]]} vs {ref[bad[0]]}"
        b = np.flatnonzero(mine["kind"] == 3)
        if len(b):
            ln = np.minimum(mine["len"][b], 288)[:, None]
            cols = np.arange(288)[None, :]
            diff = (mine["bytes"][b] != ref["bytes"][b]) & (cols < ln)
            assert not diff.any(), f"burst bytes differ on channel {int(mine['channel'][b[np.flatnonzero(diff.any(axis=1))[0]]])}"
    return len(link)
