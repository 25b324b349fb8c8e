#!/usr/bin/env python3
"""Probe: do two half-size engines on two HIP streams finish a frame of 131,072 streams sooner than one full-size engine?
(kernels of different HIP streams may run concurrently: the mel kernel's stalls and the partially filled last wave rounds of
the stage kernels could be filled by the other half's kernels).  Prints ms per frame for 1 x 131072, 2 x 65536, 4 x 32768."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

dev = torch.device("cuda", 0)
emb = W.synthetic_embedding(1234)
heads = {n: W.synthetic_head(n, 1234) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
TOTAL = 131072
for parts in (1, 2, 4, 1, 2):
    S = TOTAL // parts
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    engs = [StreamEngine(S, heads, emb, hip_stream=st.cuda_stream) for st in streams]
    pcm = [(torch.randn(S, 1280, device=dev) * 3000).round().clamp(-32768, 32767).to(torch.int16) for _ in range(parts)]
    sc = [torch.empty(S, engs[0].n_labels, device=dev) for _ in range(parts)]
    def step():
        for e, x, o in zip(engs, pcm, sc):
            e.step_device(x.data_ptr(), 1, o.data_ptr())
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 30
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"{parts} x {S}: {dt * 1e3:.3f} ms per frame of {TOTAL} streams = {TOTAL / dt / 1e6:.2f} M frames/s", flush=True)
    for e in engs:
        e.close()
    del engs, pcm, sc
