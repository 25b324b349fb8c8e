#!/usr/bin/env python3
"""Soak: the same 150-frame sequence through two fresh full-size engines must give bit-identical scores and feature rings
(no data race in the barrier-free wave-local sections, the LDS aliasing of the mel kernel or the double-buffered weight chunks)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

dev = torch.device("cuda", 0)
S, T = 131072, 150
emb = W.synthetic_embedding(1234)
heads = {n: W.synthetic_head(n, 1234) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
g = torch.Generator(device=dev); g.manual_seed(99)
pool = [(torch.randn(S, 1280, device=dev, generator=g) * a).round().clamp(-32768, 32767).to(torch.int16) for a in (3000.0, 200.0, 12000.0, 0.0, 3000.0)]
outs = []
for run in range(2):
    eng = StreamEngine(S, heads, emb)
    sc = torch.empty(S, eng.n_labels, device=dev)
    acc = torch.zeros(S, eng.n_labels, device=dev, dtype=torch.float64)
    for t in range(T):
        eng.step_device(pool[(t * 7) % len(pool)].data_ptr(), 1, sc.data_ptr())
        eng.sync()
        acc += sc.double() * (1 + t % 5)
    feats = [eng.get_features(s, 16) for s in (0, 1, 4097, 65535, 131071)]
    outs.append((acc.clone(), sc.clone(), feats))
    eng.close()
same = bool((outs[0][0] == outs[1][0]).all()) and bool((outs[0][1] == outs[1][1]).all()) and all((a == b).all() for a, b in zip(outs[0][2], outs[1][2]))
print("deterministic:", same, "| finite:", bool(torch.isfinite(outs[0][0]).all()), "| nonzero scores:", int((outs[0][1] > 0).sum()))
sys.exit(0 if same else 1)
