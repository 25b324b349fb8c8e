#!/usr/bin/env python3
"""Fused (mel + stage A) vs separate kernels on the same input: first mismatch by stream / step (development aid)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

emb = W.synthetic_embedding(1234)
heads = {n: W.synthetic_head(n, 1234) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
for S in [int(a) for a in sys.argv[1:]] or [64, 200, 5000, 40000]:
    dbg = S <= 5000
    pcm = W.synthetic_pcm(S, 1280 * 6, seed=11)
    res = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused":
            os.environ["OWW_NO_FUSE"] = "1"
        else:
            os.environ.pop("OWW_NO_FUSE", None)
        eng = StreamEngine(S, heads, emb, debug_layers=dbg)
        out, mels, feats = [], [], []
        for t in range(6):
            out.append(eng.step(pcm[:, 1280 * t:1280 * (t + 1)]).copy())
            if dbg:
                mels.append(np.stack([eng.get_mel(s, 8) for s in range(0, S, max(1, S // 50))]))
        feats = np.stack([eng.get_features(s, 6) for s in range(0, S, max(1, S // 200))])
        res[mode] = (np.stack(out), np.stack(mels) if dbg else None, feats)
        eng.close()
    a, b = res["fused"], res["unfused"]
    d = np.abs(a[0] - b[0])
    print(f"S={S}: scores max diff {d.max():.3e}; feature max diff {np.abs(a[2]-b[2]).max():.3e}")
    if d.max() > 1e-5:
        bad = np.argwhere(d > 1e-5)
        import collections
        it = collections.Counter((bad[:, 1] // 3072).tolist())
        print("   bad by iteration (stream // 3072):", sorted(it.items())[:20])
        print("   bad by step:", sorted(collections.Counter(bad[:, 0].tolist()).items()))
        w = collections.Counter((bad[:, 1] % 12).tolist())
        print("   bad by wave slot (stream % 12):", sorted(w.items()))
        bl = collections.Counter(((bad[:, 1] % 3072) // 12).tolist())
        print("   distinct blocks with bad streams:", len(bl), "of 256")
    if d.max() > 0:
        bad = np.argwhere(d > 0)
        print("   first mismatching (step, stream, label):", bad[:5].tolist(), " n bad streams:", len(set(bad[:, 1].tolist())), " of", S)
        bs = sorted(set(bad[:, 1].tolist()))
        print("   bad streams mod 12:", sorted(set(s % 12 for s in bs))[:12], " mod 3072:", sorted(set(s % 3072 for s in bs))[:10], " min/max:", bs[0], bs[-1])
    fd = np.abs(a[2] - b[2]).reshape(a[2].shape[0], -1).max(axis=1)
    step = max(1, S // 200)
    print("   streams with feature diff > 1e-4 (stream id):", (np.nonzero(fd > 1e-4)[0] * step)[:40].tolist())
    if dbg:
        md = np.abs(a[1] - b[1])
        print(f"   mel rows max diff {md.max():.3e}", np.argwhere(md > 0)[:3].tolist())
