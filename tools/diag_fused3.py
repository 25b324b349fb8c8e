#!/usr/bin/env python3
"""Which samples does a wrong fused mel row look like it was computed from? (development aid)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine
from oracle import oww_oracle as O

emb = W.synthetic_embedding(1234)
heads = {"alexa": W.synthetic_head("alexa", 1234)}
S = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
pcm = W.synthetic_pcm(S, 1280 * 2, seed=11)
res = {}
for mode in ("fused", "unfused"):
    if mode == "unfused":
        os.environ["OWW_NO_FUSE"] = "1"
    else:
        os.environ.pop("OWW_NO_FUSE", None)
    eng = StreamEngine(S, heads, emb, debug_layers=True)
    eng.step(pcm[:, :1280])
    eng.step(pcm[:, 1280:])
    res[mode] = np.stack([eng.get_mel(s, 8) for s in range(S)])
    eng.close()
d = np.abs(res["fused"] - res["unfused"])
bad = np.nonzero(d.reshape(S, -1).max(axis=1) > 1e-3)[0]
print("bad streams:", len(bad), bad[:10].tolist())
def mel_rows(x1760):         # oracle: dB rows of the 1760-sample virtual buffer, transformed
    db = O.mel_stage(x1760[None].astype(np.float32), np.float64)[0, 0]
    return db / 10 + 2
for s in bad[:4]:
    rows = np.nonzero(d[s].max(axis=1) > 1e-3)[0]
    print(f"stream {s}: bad rows {rows.tolist()}  (stream % 12 = {s % 12}, (s // 12) % 256 = {(s // 12) % 256}, s // 3072 = {s // 3072})")
    virt = np.concatenate([pcm[s, 800:1280], pcm[s, 1280:2560]]).astype(np.float64)      # tail ; chunk of step 1
    base = mel_rows(virt)
    print("   oracle vs unfused row7 max diff: %.2e ; vs fused: %.2e" % (np.abs(base[7] - res['unfused'][s, 7]).max(), np.abs(base[7] - res['fused'][s, 7]).max()))
    print("   fused - unfused row 7:", np.round(res["fused"][s, 7] - res["unfused"][s, 7], 4).tolist())
    # hypotheses about the 672-sample window of pass 3 (virtual [960, 1640))
    hyp = {}
    v = virt.copy(); v[960 + 512:960 + 680] = virt[640 + 512:640 + 680]; hyp["chunk2 stale from pass 2"] = v
    v = virt.copy(); v[960 + 512:960 + 680] = 0; hyp["chunk2 zero"] = v
    v = virt.copy(); v[960:960 + 512] = virt[640:640 + 512]; hyp["chunk1 stale from pass 2"] = v
    for sh in (8, 16, 64, 160, 320):
        v = virt.copy(); v[960 + 512:960 + 680] = np.roll(virt, -sh)[960 + 512:960 + 680]; hyp[f"chunk2 shifted +{sh}"] = v
    for k, v in hyp.items():
        r = mel_rows(v)
        print(f"   H[{k}]: |row7 - fused| = {np.abs(r[7] - res['fused'][s, 7]).max():.2e}")
