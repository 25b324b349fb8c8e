"""N>1 path on CPU: stream-range partition + per-step score gather, world_size 2 and 3 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openwakeword_amd import shard


def test_stream_range_partitions_exactly():
    for total in (0, 1, 7, 8, 65536, 1048576, 1000003):
        for world in (1, 2, 3, 4, 8):
            r = [shard.stream_range(k, world, total) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
            for sid in {0, total // 3, total - 1} if total else set():
                rank, local = shard.owner_of(sid, world, total)
                assert r[rank][0] + local == sid and r[rank][0] <= sid < r[rank][1]
    with pytest.raises(ValueError):
        shard.stream_range(2, 2, 10)
    with pytest.raises(ValueError):
        shard.owner_of(10, 2, 10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _block(lo, hi, n_labels, t):
    # a score block whose value encodes (global stream id, label, step): any mis-ordering shows
    sid = torch.arange(lo, hi, dtype=torch.float32)[:, None]
    return sid * 16 + torch.arange(n_labels, dtype=torch.float32)[None] + 0.001 * t


def _worker(rank, world, port, total, n_labels, steps, every, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = shard.ScoreGather(total, n_labels, torch.device("cpu"), every=every)
        lo, hi = shard.stream_range(rank, world, total)
        assert (g.lo, g.hi) == (lo, hi)
        ok = True
        delivered = []
        for t in range(steps):
            out = g.gather(_block(lo, hi, n_labels, t))
            flushes = (t + 1) % every == 0
            if rank == 0 and flushes:
                out = out[None] if every == 1 else out
                ok = ok and out.shape == (every, total, n_labels)
                delivered += [out[k].clone() for k in range(every)]
            else:
                ok = ok and out is None
        tail = g.flush()                                   # a partial batch (steps % every) is delivered on demand
        if rank == 0 and steps % every:
            ok = ok and tail is not None and tail.shape[0] == steps % every
            delivered += [tail[k].clone() for k in range(tail.shape[0])]
        else:
            ok = ok and tail is None
        if rank == 0:
            ok = ok and len(delivered) == steps
            for t, got in enumerate(delivered):
                ok = ok and torch.equal(got, _block(0, total, n_labels, t))
            ok = ok and g.collectives == (steps + every - 1) // every
        with pytest.raises(ValueError):
            g.gather(torch.zeros(hi - lo + 1, n_labels))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total,every", [(2, 64, 1), (2, 33, 1), (3, 10, 1), (2, 64, 4), (2, 33, 3)])
def test_score_gather_gloo(world, total, every):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, 3, 7, every, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


def test_single_process_gather_is_a_copy():
    g = shard.ScoreGather(5, 2, torch.device("cpu"))
    x = torch.arange(10, dtype=torch.float32).reshape(5, 2)
    assert torch.equal(g.gather(x), x)


def test_single_process_batched_gather():
    g = shard.ScoreGather(5, 2, torch.device("cpu"), every=3)
    xs = [torch.arange(10, dtype=torch.float32).reshape(5, 2) + t for t in range(4)]
    assert g.gather(xs[0]) is None and g.gather(xs[1]) is None
    out = g.gather(xs[2])
    assert out.shape == (3, 5, 2) and all(torch.equal(out[k], xs[k]) for k in range(3))
    assert g.gather(xs[3]) is None
    tail = g.flush()
    assert tail.shape == (1, 5, 2) and torch.equal(tail[0], xs[3]) and g.flush() is None
