#!/usr/bin/env python3
"""Throughput of bulk clip embedding (AudioFeatures.embed_clips /
compute_features_from_generator, /root/reference/openwakeword/utils.py:354-385, 542-601) on one MI355X, PCM and embeddings
resident in HBM, with the reference's algorithm timed beside it on the host cores (torch-CPU port of oracle/, one
single-threaded process per core: mel per clip, every 76-row window through the full CNN).  Prints one JSON line.

usage: python tools/bench_embed_clips.py [--clips 16384] [--seconds 2.0] [--reps 5] [--cpu-seconds 8]"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

STEP_MFLOP = 11.225      # executed by the incremental CNN per 8 mel rows
WINDOW_MFLOP = 83.912    # one full 76x32 window, what the reference evaluates per embedding


def _cpu_worker(args):
    seed, budget_s, n = args
    import torch
    torch.set_num_threads(1)
    from openwakeword_amd import weights as W
    from oracle.oww_oracle_torch import TorchCpuPort
    port = TorchCpuPort(W.synthetic_embedding(1234), {}, threads=1)
    clip = W.synthetic_pcm(1, n, seed=seed)
    done, t0 = 0, time.perf_counter()
    while True:
        spec = port.mel(clip)[0]
        wins = torch.stack([spec[i:i + 76] for i in range(0, spec.shape[0] - 75, 8)])
        port.embed(wins)
        done += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    return done, time.perf_counter() - t0


def cpu_baseline(n, budget_s):
    from oracle.parity_sample import effective_cpus
    cores = effective_cpus()               # (the affinity mask capped by the container's CPU quota)
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(50 + i, budget_s, n) for i in range(cores)])
    clips, wall = sum(r[0] for r in res), max(r[1] for r in res)
    return {"value": round(clips / wall, 2), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{clips} clips in {wall:.1f} s on {cores} single-threaded processes (full-window CNN per embedding, torch-CPU/oneDNN)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=16384)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    args = ap.parse_args()
    n = int(args.seconds * 16000)
    cpu = cpu_baseline(n, args.cpu_seconds) if args.cpu_seconds > 0 else None      # forks: before HIP is touched

    import torch
    from openwakeword_amd import weights as W
    from openwakeword_amd.engine import StreamEngine
    dev = torch.device("cuda", 0)
    B = args.clips
    frames = (n - 512) // 160 + 1
    n_out = (frames - 76) // 8 + 1
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    pcm = (torch.randn(B, n, device=dev, generator=gen) * 3000.0).round().clamp(-32768, 32767).to(torch.int16)
    out = torch.empty(B, n_out, 96, device=dev, dtype=torch.float32)
    eng = StreamEngine(B, {}, W.synthetic_embedding(1234))
    eng.embed_clips_device(pcm.data_ptr(), B, n, out.data_ptr())                   # warm-up (the call synchronises)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        eng.embed_clips_device(pcm.data_ptr(), B, n, out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    ok = bool(torch.isfinite(out).all().item())
    steps = 9 + n_out
    res = {"metric": "clip embeddings (AudioFeatures.embed_clips), clips/s on 1 GPU", "value": round(B / dt, 1), "unit": "clips/s",
           "n_gpus": 1, "clips": B, "clip_seconds": args.seconds, "windows_per_clip": n_out, "ms_per_call": round(dt * 1e3, 2),
           "audio_seconds_per_second": round(B * args.seconds / dt, 1),
           "embeddings_per_second": round(B * n_out / dt, 1),
           "executed_tflops": round(B * steps * STEP_MFLOP * 1e6 / dt / 1e12, 2),
           "reference_form_tflops": round(B * n_out * WINDOW_MFLOP * 1e6 / dt / 1e12, 2),
           "note": "incremental CNN: 9 warm-up steps + one step per window; 'reference_form' credits the full-window flops the "
                   "reference would execute for the same output (not a roofline claim)",
           "finite": ok, "data": "synthetic Gaussian int16 PCM (RMS 3000), synthetic weights seed 1234", "cpu_baseline": cpu}
    if cpu:
        res["x_cpu_host"] = round(res["value"] / cpu["value"], 1)
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
