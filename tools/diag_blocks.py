"""Development aid: single-launch step vs block-pipelined step (OWW_BLOCKS=3), frame by frame, printing where they differ."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine
emb = W.synthetic_embedding(1234)
heads = {n: W.synthetic_head(n, 1234) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
S = 16384 + 96
pcm = W.synthetic_pcm(S, 1280 * 7, seed=72)
os.environ.pop("OWW_BLOCKS", None)
one = StreamEngine(S, heads, emb)
os.environ["OWW_BLOCKS"] = sys.argv[1] if len(sys.argv) > 1 else "3"
three = StreamEngine(S, heads, emb)
MASK = len(sys.argv) > 2
on = (np.random.default_rng(3).random(S) < 0.8).astype(np.uint8)
for rep in range(3):
    one.reset(); three.reset()
    for t in range(7):
        x = np.ascontiguousarray(pcm[:, 1280 * t: 1280 * (t + 1)])
        if t == 5 and MASK:
            a, b = one.step_masked(x, on), three.step_masked(x, on)
        else:
            a, b = one.step(x), three.step(x)
        ea = np.stack([one.get_features(s, 1)[0] for s in (0, 1)])
        bad = np.nonzero((a != b).any(axis=1))[0]
        print(f"rep {rep} frame {t}: {len(bad)} streams differ", bad[:12].tolist(), (np.abs(a - b).max() if len(bad) else 0.0), flush=True)
one.close(); three.close()
