#!/usr/bin/env python3
"""Fused vs separate: where does a bad stream first differ -- mel rows or CNN layers? (development aid)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

emb = W.synthetic_embedding(1234)
heads = {"alexa": W.synthetic_head("alexa", 1234)}
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6200
steps = 3
pcm = W.synthetic_pcm(S, 1280 * steps, seed=11)
res = {}
for mode in ("fused", "unfused"):
    if mode == "unfused":
        os.environ["OWW_NO_FUSE"] = "1"
    else:
        os.environ.pop("OWW_NO_FUSE", None)
    eng = StreamEngine(S, heads, emb, debug_layers=True)
    per = []
    for t in range(steps):
        eng.step(pcm[:, 1280 * t:1280 * (t + 1)])
        per.append((np.stack([eng.get_mel(s, 8) for s in range(S)]),
                    np.stack([eng.debug_layer(s, 0) for s in range(S)]),
                    np.stack([eng.debug_layer(s, 2) for s in range(S)])))
    res[mode] = per
    eng.close()
for t in range(steps):
    for name, k in (("mel", 0), ("conv0", 1), ("conv2", 2)):
        d = np.abs(res["fused"][t][k] - res["unfused"][t][k]).reshape(S, -1).max(axis=1)
        bad = np.nonzero(d > 1e-3)[0]
        print(f"step {t} {name}: max diff {d.max():.3e}, streams > 1e-3: {len(bad)} {bad[:12].tolist()}")
        if len(bad) and name == "mel":
            s = bad[0]
            dd = np.abs(res["fused"][t][0][s] - res["unfused"][t][0][s])
            print("   stream", s, "bad rows:", np.nonzero(dd.max(axis=1) > 1e-3)[0].tolist(), "row max diffs", np.round(dd.max(axis=1), 4).tolist())
            print("   fused row vals", np.round(res["fused"][t][0][s][np.argmax(dd.max(axis=1))][:8], 3), "unfused", np.round(res["unfused"][t][0][s][np.argmax(dd.max(axis=1))][:8], 3))
