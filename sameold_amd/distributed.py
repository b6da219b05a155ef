"""Multi-GPU plumbing: channels shard embarrassingly across ranks (one process per GPU);
the only collective is a gather of the decoded bursts to rank 0 at the end of a batch.

The gather is two RCCL calls on fixed-size records (`backend="nccl"` is RCCL on ROCm): an
all_gather of per-rank counts, then a gather of records padded to the largest count.
Volume is a few hundred bytes per burst, so xGMI bandwidth is irrelevant; what matters is
that there is no collective on the demodulation path itself.  The same code runs over
gloo on CPU tensors, which is how the CPU test-suite covers it.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 304          # 4 channel + 8 sample_counter + 4 length + 288 burst bytes
_HDR = 16


def shard_channels(n_channels: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of channels owned by `rank`: [first, first + count)."""
    base, rem = divmod(n_channels, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def pack_bursts(bursts: Sequence[Tuple[int, int, bytes]]) -> np.ndarray:
    """(global_channel, sample_counter, bytes) -> uint8 [n, RECORD_BYTES]"""
    out = np.zeros((len(bursts), RECORD_BYTES), dtype=np.uint8)
    for i, (ch, ctr, data) in enumerate(bursts):
        out[i, 0:4] = np.frombuffer(np.uint32(ch).tobytes(), dtype=np.uint8)
        out[i, 4:12] = np.frombuffer(np.uint64(ctr).tobytes(), dtype=np.uint8)
        n = min(len(data), RECORD_BYTES - _HDR)
        out[i, 12:16] = np.frombuffer(np.uint32(n).tobytes(), dtype=np.uint8)
        out[i, _HDR:_HDR + n] = np.frombuffer(data[:n], dtype=np.uint8)
    return out


def unpack_bursts(recs: np.ndarray) -> List[Tuple[int, int, bytes]]:
    out = []
    for r in recs:
        ch = int(np.frombuffer(r[0:4].tobytes(), dtype=np.uint32)[0])
        ctr = int(np.frombuffer(r[4:12].tobytes(), dtype=np.uint64)[0])
        n = int(np.frombuffer(r[12:16].tobytes(), dtype=np.uint32)[0])
        out.append((ch, ctr, r[_HDR:_HDR + n].tobytes()))
    return out


def pack_burst_events(events: np.ndarray, first_channel: int = 0, zero_padded: bool = False) -> np.ndarray:
    """Vectorised pack_bursts for an EVENT_DTYPE array (sameold_amd.receiver): the SAME_LINK_BURST
    events of one rank -> uint8 [n, RECORD_BYTES] with global channel numbers.  `zero_padded`:
    the caller vouches that event bytes past `len` are zero (true for events polled from the
    library), which saves masking them."""
    # work on the raw bytes: fancy indexing of a structured array with a 288-byte sub-array field
    # goes element by element, a row take of a uint8 matrix is one memcpy per row
    dt = events.dtype
    raw = np.ascontiguousarray(events).view(np.uint8).reshape(-1, dt.itemsize)
    off = {name: dt.fields[name][1] for name in ("kind", "channel", "sample_counter", "len", "bytes")}
    kind = raw[:, off["kind"]: off["kind"] + 4].view(np.uint32).ravel()
    rows = raw[np.flatnonzero(kind == 3)]                # SAME_LINK_BURST
    out = np.empty((len(rows), RECORD_BYTES), dtype=np.uint8)
    if len(rows):
        ch = np.ascontiguousarray(rows[:, off["channel"]: off["channel"] + 4]).view(np.uint32).ravel()
        out[:, 0:4] = (ch + np.uint32(first_channel)).view(np.uint8).reshape(-1, 4)
        out[:, 4:12] = rows[:, off["sample_counter"]: off["sample_counter"] + 8]
        n = np.minimum(np.ascontiguousarray(rows[:, off["len"]: off["len"] + 4]).view(np.uint32).ravel(),
                       np.uint32(RECORD_BYTES - _HDR))
        out[:, 12:16] = n.view(np.uint8).reshape(-1, 4)
        out[:, _HDR:] = rows[:, off["bytes"]: off["bytes"] + RECORD_BYTES - _HDR]
        if not zero_padded:
            short = np.flatnonzero(n < RECORD_BYTES - _HDR)
            if len(short):
                cols = np.arange(RECORD_BYTES - _HDR, dtype=np.uint32)[None, :]
                body = out[short, _HDR:]
                body[cols >= n[short, None]] = 0
                out[short, _HDR:] = body
    return out


_pinned = {}


def _pinned_bytes(tag: str, n_rows: int) -> torch.Tensor:
    """A cached pinned host tensor of at least [n_rows, RECORD_BYTES] uint8.  Transfers between
    the device and pageable host memory are staged by the runtime and can hold up kernels queued
    beside them; pinned memory moves in one DMA."""
    t = _pinned.get(tag)
    if t is None or t.shape[0] < n_rows:
        rows = max(n_rows + n_rows // 2, 1024)
        t = torch.empty((rows, RECORD_BYTES), dtype=torch.uint8, pin_memory=True)
        _pinned[tag] = t
    return t


def pinned_send_rows(n_rows: int) -> np.ndarray:
    """A numpy view [>= n_rows, RECORD_BYTES] of the pinned host buffer `gather_records` stages its send from: a caller that
    packs its records straight into it (`SameBatchReceiver.pack_bursts_np(out=...)`) saves the copy into the stage -- 7 MB per
    step and rank at the bench's shapes, and a fresh array's page faults on top.  (Waits for the last send's host-to-device copy out
    of this buffer, which finished long ago as a rule.)"""
    ev = _send_done.get("ev")
    if ev is not None:
        ev.synchronize()
    return _pinned_bytes("send", n_rows).numpy()


_send_done: dict = {}


_device_rows: dict = {}


def _device_rows_buffer(tag, rows: int, device) -> torch.Tensor:
    """A cached device tensor of at least [rows, RECORD_BYTES] uint8 (grown by half when it is too small): the gather's send
    and receive buffers are not allocated and cleared again every step -- at configs[3]'s shape that was 8 x 7.5 MB of
    torch.zeros per step on rank 0.  Rows beyond a rank's count are never read, so nothing needs clearing."""
    key = (tag, str(device))
    t = _device_rows.get(key)
    if t is None or t.shape[0] < rows:
        t = torch.empty((max(rows + rows // 2, 1024), RECORD_BYTES), dtype=torch.uint8, device=device)
        _device_rows[key] = t
    return t


class LandedRecords:
    """What `gather_records(..., wait=False)` returns on the destination rank of a GPU gather: the records are on their way from
    the device's receive buffer to a pinned host buffer on a side stream.  `len()` / `.shape` are known at once (the counts came
    with the all_gather); `.numpy()` -- or `np.asarray(obj)` -- waits for the copy and returns the view (valid until the call
    after next: two landing buffers alternate)."""

    def __init__(self, land: torch.Tensor, total: int, done: "torch.cuda.Event"):
        self._land, self._total, self._done = land, total, done
        self.shape = (total, RECORD_BYTES)

    def __len__(self) -> int:
        return self._total

    def numpy(self) -> np.ndarray:
        self._done.synchronize()
        return self._land[: self._total].numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)


_side: dict = {}           # per device: the landing stream, the event of the last landing, which landing buffer is next


def gather_records(recs: np.ndarray, device: torch.device, dst: int = 0, wait: bool = True):
    """Gather packed records ([n, RECORD_BYTES] uint8 per rank) on rank `dst`: one all_gather of
    the counts (a single tensor, one host synchronisation), one gather of the records padded to the largest count, through
    cached device and pinned host buffers.  Returns the concatenation on `dst` (rank order; on a GPU a view of the cached
    landing buffer, valid until the next call), None elsewhere; the identity without a process group.

    `wait=False` (GPU, destination rank): the device-to-host landing of the gathered records -- 8 x 7 MB per step at the bench's
    shapes, over a millisecond of PCIe time that would otherwise sit on rank 0's step and, through the next all_gather, on every
    rank's -- goes to a side stream and a `LandedRecords` comes back at once; whoever needs the bytes waits for them there."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return recs
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = torch.device(device).type == "cuda"
    n = torch.tensor([len(recs)], dtype=torch.int64, device=device)
    all_n = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_n, n)
    counts = [int(v) for v in all_n.tolist()]                   # (the step's one device -> host synchronisation)
    m = max(max(counts), 1)
    buf = _device_rows_buffer("send", m, device)[:m]
    if len(recs):
        src = torch.from_numpy(np.ascontiguousarray(recs))
        if on_gpu:
            stage = _pinned_bytes("send", len(recs))[: len(recs)]
            if src.data_ptr() != stage.data_ptr():             # (packed in place by the caller: pinned_send_rows)
                stage.copy_(src)
            buf[: len(recs)].copy_(stage, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            _send_done["ev"] = ev
        else:
            buf[: len(recs)] = src
    out = None
    side = _side.setdefault(str(device), {"stream": None, "done": None, "flip": 0}) if (on_gpu and rank == dst) else None
    if rank == dst:
        big = _device_rows_buffer("recv", m * world, device)
        out = [big[r * m:(r + 1) * m] for r in range(world)]
        if side is not None and side["done"] is not None:
            torch.cuda.current_stream(device).wait_event(side["done"])       # the last landing has read the receive buffer
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    total = sum(counts)
    if on_gpu:
        if not wait:
            if side["stream"] is None:
                side["stream"] = torch.cuda.Stream(device)
            side["flip"] ^= 1
            land = _pinned_bytes("recv%d" % side["flip"], total)
            arrived = torch.cuda.Event()
            arrived.record(torch.cuda.current_stream(device))
            side["stream"].wait_event(arrived)
            with torch.cuda.stream(side["stream"]):
                at = 0
                for r in range(world):
                    land[at: at + counts[r]].copy_(out[r][: counts[r]], non_blocking=True)
                    at += counts[r]
                done = torch.cuda.Event()
                done.record(side["stream"])
            side["done"] = done
            return LandedRecords(land, total, done)
        land = _pinned_bytes("recv", total)
        at = 0
        for r in range(world):
            land[at: at + counts[r]].copy_(out[r][: counts[r]], non_blocking=True)
            at += counts[r]
        torch.cuda.current_stream(device).synchronize()
        return land[:total].numpy()            # a view of the cached landing buffer: valid until the next call
    return np.concatenate([out[r][: counts[r]].numpy() for r in range(world)], axis=0)


def gather_bursts(bursts: Sequence[Tuple[int, int, bytes]], device: torch.device,
                  dst: int = 0) -> Optional[List[Tuple[int, int, bytes]]]:
    """Gather every rank's bursts on rank `dst` (returns None on the other ranks).
    Without an initialised process group this is the identity."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(bursts)
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([len(bursts)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    buf = torch.zeros((m, RECORD_BYTES), dtype=torch.uint8, device=device)
    if len(bursts):
        buf[: len(bursts)] = torch.from_numpy(pack_bursts(bursts)).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    res: List[Tuple[int, int, bytes]] = []
    for r in range(world):
        res += unpack_bursts(out[r][: counts[r]].cpu().numpy())
    return res
