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


def _worker(rank, world, port, total, n_labels, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = shard.ScoreGather(total, n_labels, torch.device("cpu"))
        lo, hi = shard.stream_range(rank, world, total)
        assert (g.lo, g.hi) == (lo, hi)
        ok = True
        for t in range(steps):
            # a score block whose value encodes (global stream id, label, step): any mis-ordering shows
            sid = torch.arange(lo, hi, dtype=torch.float32)[:, None]
            local = sid * 16 + torch.arange(n_labels, dtype=torch.float32)[None] + 0.001 * t
            out = g.gather(local)
            if rank == 0:
                want = torch.arange(total, dtype=torch.float32)[:, None] * 16 + torch.arange(n_labels, dtype=torch.float32)[None] + 0.001 * t
                ok = ok and torch.equal(out, want)
            else:
                ok = ok and out is None
        with pytest.raises(ValueError):
            g.gather(torch.zeros(hi - lo + 1, n_labels))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 64), (2, 33), (3, 10)])
def test_score_gather_gloo(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, 3, 4, q)) for r in range(world)]
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
