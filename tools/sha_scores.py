#!/usr/bin/env python3
"""sha256 over 60 steps of scores and the feature rings of a 4,096-stream engine (three heads + the default six) for the library named by
OWW_LIB: two builds that claim the same arithmetic must print the same digest.  usage: OWW_LIB=... python tools/sha_scores.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np                                           # noqa: E402
from openwakeword_amd import weights as W, _build            # noqa: E402
from openwakeword_amd.engine import StreamEngine             # noqa: E402

emb = W.synthetic_embedding(1234)
h = hashlib.sha256()
for names, S in ((("alexa", "hey_mycroft", "hey_jarvis"), 4096), (("alexa", "hey_mycroft", "hey_jarvis", "hey_rhasspy", "timer", "weather"), 70000)):
    heads = {n: W.synthetic_head(n, 1234) for n in names}
    e = StreamEngine(S, heads, emb)
    pcm = W.synthetic_pcm(S, 1280 * 4, seed=5)
    for t in range(60 if S < 10000 else 12):
        h.update(e.step(pcm[:, 1280 * (t % 4): 1280 * (t % 4 + 1)]).tobytes())
    for s in (0, 1, S // 2, S - 1):
        h.update(e.get_features(s, 16).tobytes())
    e.close()
print(_build.lib_path(), h.hexdigest())
