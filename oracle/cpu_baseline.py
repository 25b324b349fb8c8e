"""
TEST / BASELINE INFRASTRUCTURE ONLY -- times the reference's per-frame algorithm on the host cores.

Two figures, as SURVEY §8d asks (the reference: `Model.predict` with onnxruntime pinned to one thread, model.py:149-151;
its only scale-out is one process per core, `utils.bulk_predict`, utils.py:502-536):
  (i)  ONE single-threaded process alone on the machine            -> the per-core figure
  (ii) one single-threaded process per PHYSICAL core (SMT siblings left idle: an oversubscribed run measured 48.6
       frames/s/core in round 1 against 273 frames/s for a lone process)  -> the per-host figure (`value`)
The arithmetic is the torch-CPU port of oracle/oww_oracle_torch.py because onnxruntime and the .onnx files do not exist
offline (kind = "port").  Must be called BEFORE the parent process initialises HIP (workers are forked).
"""
from __future__ import annotations

import glob
import multiprocessing as mp
import os
import time

SURVEY_PER_CORE_PROBE = 273.0       # single-thread probe of the same topology in the survey container


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


def _pinned_worker(args):
    cpu, rest = args[0], args[1:]
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})
        except (AttributeError, OSError):
            pass
    return _worker(rest)


def physical_cores():
    """One logical CPU per physical core among the CPUs this process may run on (first SMT sibling of each core)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    seen, firsts = set(), []
    for c in allowed:
        sib = None
        for f in glob.glob(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list"):
            try:
                sib = open(f).read().strip()
            except OSError:
                pass
        key = sib if sib else f"cpu{c}"
        if key not in seen:
            seen.add(key)
            firsts.append(c)
    return firsts, len(allowed)


def run(head_names, budget_s: float = 12.0, max_workers: int | None = None, batch: int = 4) -> dict:
    head_names = list(head_names)
    firsts, logical = physical_cores()
    # a container may SEE every CPU of the host and own a fraction of them (cgroup quota: the GPU boxes of this pool show 256 CPUs
    # and a quota of 16): more workers than that only share the quota -- round 2's "80 frames/s per core on 128 cores" was exactly
    # that, 16 CPUs' worth of work spread over 128 processes
    from .parity_sample import effective_cpus
    quota = effective_cpus()
    workers = max(1, min(len(firsts), quota, max_workers or len(firsts)))
    ctx = mp.get_context("fork")
    t_single = max(2.0, budget_s * 0.3)
    t_all = max(2.0, budget_s - t_single)
    with ctx.Pool(1) as pool:                       # (i) a lone process: nothing else runs on the host
        n1, w1 = pool.map(_pinned_worker, [(firsts[0], 999, t_single, head_names, batch)])[0]
    with ctx.Pool(workers) as pool:                 # (ii) one process per physical core
        res = pool.map(_pinned_worker, [(firsts[i], 1000 + i, t_all, head_names, batch) for i in range(workers)])
    frames = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": round(frames / wall, 1), "unit": "frames/s", "cores": workers, "kind": "port",
            "per_core": round(frames / wall / workers, 1),
            "single_process": {"value": round(n1 / w1, 1), "unit": "frames/s", "cores": 1,
                               "sample": f"{n1} frames in {w1:.1f} s, one single-threaded process alone on the host"},
            "host": {"logical_cpus": logical, "physical_cores": len(firsts), "cpu_quota": quota, "workers": workers,
                     "pinning": "one worker per physical core (first SMT sibling), SMT siblings idle; never more workers than the "
                                "container's CPU quota"},
            "survey_probe_per_core": SURVEY_PER_CORE_PROBE,
            "sample": f"{frames} frames in {wall:.1f} s: {workers} single-threaded processes (one per usable core: {len(firsts)} physical cores "
                      f"visible, CPU quota {quota}) x {batch} streams each, the reference's algorithm (257-bin DFT mel, FULL 76x32 window "
                      f"through the 20-layer CNN every frame, {len(head_names)} heads) as a torch-CPU/oneDNN port "
                      "(oracle/oww_oracle_torch.py); onnxruntime and the .onnx model files are not available offline"}
