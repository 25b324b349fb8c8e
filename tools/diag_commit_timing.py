"""Development aid: phases of oww_commit (OWW_COMMIT_TIMING=1) for a few engine shapes.  python tools/diag_commit_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["OWW_COMMIT_TIMING"] = "1"
import numpy as np
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine
heads = {n: W.synthetic_head(n, 1) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
emb = W.synthetic_embedding(1)
StreamEngine(4, heads, emb).close()                      # (library load, first-use costs)
for S, fam, cal in ((6, 3, "default"), (6, 3, None), (4096, 3, "default"), (16480, 3, "default"), (6, 1, None), (131072, 3, "default")):
    t0 = time.perf_counter()
    e = StreamEngine(S, heads, emb, use_mfma=fam, calibration_pcm=cal)
    t1 = time.perf_counter()
    e.close()
    t2 = time.perf_counter()
    print(f"== S={S} family={fam} calibration={cal}: create {1e3 * (t1 - t0):.1f} ms, destroy {1e3 * (t2 - t1):.1f} ms", file=sys.stderr, flush=True)
