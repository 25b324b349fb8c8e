"""Which entry differs under OWW_GUARD_ALLOC=1 / 2 from the plain run (the plain run writes /tmp/diag_guard.npz first)."""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

B, n = 13, 16000
emb = W.synthetic_embedding(1234)
x = W.synthetic_pcm(B, n, seed=100 + B)
x[0, : n // 2] = 0
eng = StreamEngine(16, {"alexa": W.synthetic_head("alexa", 1234)}, emb)
r1 = eng.embed_clips(x)
spec = eng.mel_clips(x)
spec_b = eng.mel_clips(x)
mel1 = eng.mel(x)
n_win = (spec.shape[1] - 76) // 8 + 1
win = np.ascontiguousarray(spec[:, : 76 + 8 * (n_win - 1)] / 10.0 + 2.0, dtype=np.float32)
ref = eng.embed(win)
ref_b = eng.embed(win)
path = "/tmp/diag_guard.npz"
if not os.environ.get("OWW_GUARD_ALLOC", "0").strip("0"):
    np.savez(path, r1=r1, spec=spec, mel1=mel1, ref=ref, win=win)
    print("saved")
else:
    g = np.load(path)
    def rep(name, a, b):
        d = np.abs(a.astype(np.float64) - b)
        idx = np.argwhere(d > 0)
        print(f"{name}: shape {a.shape} differing {len(idx)} max {d.max():.3g}", "first", idx[:3].tolist(), "last", idx[-3:].tolist())
    rep("embed_clips", r1, g["r1"])
    rep("mel_clips", spec, g["spec"])
    rep("mel_clips again", spec_b, g["spec"])
    rep("mel", mel1, g["mel1"])
    rep("embed(own win)", ref, ref_b)
    ref_g = eng.embed(g["win"])
    rep("embed(plain win)", ref_g, g["ref"])
    rep("embed(plain win) again", eng.embed(g["win"]), g["ref"])
