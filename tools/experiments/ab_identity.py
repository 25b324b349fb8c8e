#!/usr/bin/env python3
"""Scores + feature rings of one library build (OWW_LIB) over a fixed 60-frame sequence as one sha256 -- two builds whose kernels
differ only in instruction selection must print the same digest.  usage: OWW_LIB=... python tools/experiments/ab_identity.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

dev = torch.device("cuda", 0)
S, T = 4096, 60
emb = W.synthetic_embedding(1234)
heads = {n: W.synthetic_head(n, 1234) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
g = torch.Generator(device=dev); g.manual_seed(99)
pool = [(torch.randn(S, 1280, device=dev, generator=g) * a).round().clamp(-32768, 32767).to(torch.int16) for a in (3000.0, 200.0, 12000.0, 0.0)]
eng = StreamEngine(S, heads, emb)
sc = torch.empty(S, eng.n_labels, device=dev)
h = hashlib.sha256()
for t in range(T):
    eng.step_device(pool[(t * 3) % len(pool)].data_ptr(), 1, sc.data_ptr())
    eng.sync()
    h.update(sc.cpu().numpy().tobytes())
for s in (0, 1, 4095):
    h.update(eng.get_features(s, 16).tobytes() if hasattr(eng.get_features(s, 16), "tobytes") else eng.get_features(s, 16).cpu().numpy().tobytes())
eng.close()
print(os.environ.get("OWW_LIB", "default"), h.hexdigest())
