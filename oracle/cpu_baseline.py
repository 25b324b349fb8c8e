"""
TEST / BASELINE INFRASTRUCTURE ONLY -- times the reference's per-frame algorithm on the host cores.

One single-threaded worker process per core, each owning its own streams, exactly how the reference scales
out on a CPU (`openwakeword.utils.bulk_predict` forks one Model per core, utils.py:502-536, with the
inference sessions pinned to one thread, model.py:149-151).  The arithmetic is the torch-CPU port of
oracle/oww_oracle_torch.py because onnxruntime and the .onnx files do not exist offline.
Must be called BEFORE the parent process initialises HIP (workers are forked).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import time


def _worker(args):
    seed, budget_s, head_names, batch = args
    import torch
    torch.set_num_threads(1)
    from openwakeword_amd import weights as W
    from oracle.oww_oracle_torch import TorchCpuPort
    emb = W.synthetic_embedding(1234)
    heads = {n: W.synthetic_head(n, 1234) for n in head_names}
    port = TorchCpuPort(emb, heads, threads=1)
    pcm = W.synthetic_pcm(batch, 1760, seed=seed)
    mel_ring = torch.full((batch, 76, 32), 1.0)
    feat = torch.zeros(batch, max(h["T"] for h in heads.values()), 96)
    port.frame(pcm, mel_ring, feat)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        _, mel_ring, feat = port.frame(pcm, mel_ring, feat)
        n += batch
    return n, time.perf_counter() - t0


def run(head_names, budget_s: float = 12.0, max_workers: int | None = None, batch: int = 4) -> dict:
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    workers = max(1, min(cores, max_workers or cores))
    ctx = mp.get_context("fork")
    with ctx.Pool(workers) as pool:
        res = pool.map(_worker, [(1000 + i, budget_s, list(head_names), batch) for i in range(workers)])
    frames = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": round(frames / wall, 1), "unit": "frames/s", "cores": workers, "kind": "port",
            "per_core": round(frames / wall / workers, 1),
            "sample": f"{frames} frames in {wall:.1f} s: {workers} single-threaded processes x {batch} streams each, "
                      f"the reference's algorithm (257-bin DFT mel, FULL 76x32 window through the 20-layer CNN every frame, "
                      f"{len(head_names)} heads) as a torch-CPU/oneDNN port (oracle/oww_oracle_torch.py); "
                      "onnxruntime and the .onnx model files are not available offline"}
