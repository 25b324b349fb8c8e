#!/usr/bin/env python3
"""Per-probe error of the parity probe set on a small engine (development aid)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from oracle import parity_sample as PS
from openwakeword_amd.engine import StreamEngine

pcm = PS.probe_pcm()
ref = PS.oracle_reference()
emb, heads = PS._weights(PS.HEADS3)
for fam in (3, 1):
    for S in (64, 200):
        eng = StreamEngine(S, heads, emb, use_mfma=fam)
        eng.reset(None, ref["init_features"][-eng.feature_ring:])
        ids = np.arange(64) if S == 64 else PS.probe_stream_ids(S)
        err = np.zeros((64, 16))
        buf = np.zeros((S, 1280), np.int16)
        for t in range(16):
            buf[:] = 0
            buf[ids] = pcm[:, t * 1280:(t + 1) * 1280]
            got = eng.step(buf)
            err[:, t] = np.abs(got[ids] - ref["scores"][:, t]).max(axis=1)
        print(f"family {fam} S={S}: max err {err.max():.3e}")
        bad = np.argwhere(err > 1e-4)
        print("  bad probes:", sorted(set(bad[:, 0].tolist())), " first bad frames:", {int(p): int(bad[bad[:, 0] == p][:, 1].min()) for p in set(bad[:, 0].tolist())})
        print("  per-kind max err:", [f"{err[k::8].max():.1e}" for k in range(8)])
        eng.close()
