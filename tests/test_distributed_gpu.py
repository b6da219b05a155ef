"""The device path of the burst gather (pinned staging both ways) on one GPU: the collectives are
replaced by in-process stand-ins for a world of two identical ranks, everything else is the code
the multi-GPU bench runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gather_records_device_path(monkeypatch):
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sameold_amd import distributed as sd
    dev = torch.device("cuda", 0)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda: 2)
    monkeypatch.setattr(dist, "get_rank", lambda: 0)

    def fake_all_gather(out_list, t):
        for o in out_list:
            o.copy_(t)

    def fake_gather(t, gather_list=None, dst=0):
        for o in gather_list:
            o.copy_(t)

    def fake_all_gather_into_tensor(out, t):
        out.view(2, -1).copy_(t.view(1, -1).expand(2, -1))

    monkeypatch.setattr(dist, "all_gather", fake_all_gather)
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_all_gather_into_tensor)
    monkeypatch.setattr(dist, "gather", fake_gather)
    bursts = [(4096 + i, 1000 * i + (1 << 33), bytes([65 + i % 26]) * (5 + i % 200)) for i in range(3000)]
    recs = sd.pack_bursts(bursts)
    for _ in range(2):                                   # second call reuses the pinned buffers
        got = sd.gather_records(recs, dev)
        assert got.shape == (2 * len(recs), sd.RECORD_BYTES)
        assert np.array_equal(got[: len(recs)], recs) and np.array_equal(got[len(recs):], recs)
    assert sd.gather_records(recs[:0], dev).shape[0] == 0
    assert sd.unpack_bursts(sd.gather_records(recs[:7], dev))[:7] == bursts[:7]
    # wait=False (what the multi-GPU bench uses): the landing on a side stream; the count at once, the bytes on demand, two
    # landing buffers so that the result of one call survives the next
    a = sd.gather_records(recs, dev, wait=False)
    b = sd.gather_records(recs[:100], dev, wait=False)
    assert len(a) == 2 * len(recs) and a.shape == (2 * len(recs), sd.RECORD_BYTES) and len(b) == 200
    ga, gb = a.numpy(), np.asarray(b)
    assert np.array_equal(ga[: len(recs)], recs) and np.array_equal(ga[len(recs):], recs)
    assert np.array_equal(gb[:100], recs[:100]) and np.array_equal(gb[100:], recs[:100])
    assert len(sd.gather_records(recs[:0], dev, wait=False)) == 0
    # records packed straight into the pinned buffer the gather sends from (what bench.py's ranks do: no copy into the stage)
    for n in (len(recs), 500):
        view = sd.pinned_send_rows(n + 1000)
        assert view.shape[0] >= n + 1000 and view.shape[1] == sd.RECORD_BYTES
        view[:n] = recs[:n]
        got = sd.gather_records(view[:n], dev, wait=False).numpy()
        assert np.array_equal(got[:n], recs[:n]) and np.array_equal(got[n:], recs[:n])
