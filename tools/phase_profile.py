#!/usr/bin/env python3
"""Print per-phase shader-clock durations of one workgroup of each generic CNN stage kernel under full load.
usage: OWW_PROF_BLOCK=<wg index> python tools/phase_profile.py [streams]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
os.environ.setdefault("OWW_PROF_BLOCK", "5000")
heads = {n: W.synthetic_head(n) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
eng = StreamEngine(S, heads)
pcm = W.synthetic_pcm(4096, 1280, seed=1)
pcm = np.tile(pcm, (S // 4096 + 1, 1))[:S]
for _ in range(4):
    eng.step(pcm)
prof = eng.debug_profile()
names = ["in-loads", "barrier0", "conv a", "barrier1", "conv b", "barrier2", "conv c", "barrier3", "conv d", "barrier4", "pool/out"]
for si, st in enumerate("BCDE"):
    p = prof[si]
    waves = [w for w in range(16) if p[w, 0] != 0]
    print(f"stage {st}: {len(waves)} waves; durations in shader clocks per wave (mark k -> k+1)")
    t0 = min(p[w, 0] for w in waves)
    for k, nm in enumerate(names):
        print(f"  {nm:9s} " + " ".join(f"{int(p[w, k + 1] - p[w, k]):7d}" for w in waves))
    print(f"  {'total':9s} " + " ".join(f"{int(p[w, 11] - p[w, 0]):7d}" for w in waves) + f"   start skew {[int(p[w,0]-t0) for w in waves]}")
