"""Development aid: two identical engines, masked steps at a given participation: any difference is a race."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine
emb = W.synthetic_embedding(1234)
heads = {n: W.synthetic_head(n, 1234) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16384 + 96
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
pcm = W.synthetic_pcm(S, 1280 * 8, seed=72)
rng = np.random.default_rng(3)
a, b = StreamEngine(S, heads, emb), StreamEngine(S, heads, emb)
for t in range(8):
    x = np.ascontiguousarray(pcm[:, 1280 * t: 1280 * (t + 1)])
    on = (rng.random(S) < frac).astype(np.uint8)
    if t < 5:
        ra, rb = a.step(x), b.step(x)
    else:
        ra, rb = a.step_masked(x, on), b.step_masked(x, on)
    bad = np.nonzero((ra != rb).any(axis=1))[0]
    print(f"frame {t} ({'masked' if t >= 5 else 'plain'}): {len(bad)} streams differ", bad[:16].tolist(), (float(np.abs(ra - rb).max()) if len(bad) else 0.0), flush=True)
a.close(); b.close()
