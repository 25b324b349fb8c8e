#!/usr/bin/env python3
"""Step time with a recurrent head (train.py's model_type "rnn": owk::heads_rnn_kernel, VALU) at a few batch sizes -- the record behind
DESIGN.md 5.12 ("fine for the tens of streams such a custom model would meet, not a large-batch path").  usage: python tools/rnn_timing.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np                                                     # noqa: E402
from openwakeword_amd import weights as W                              # noqa: E402
from openwakeword_amd.engine import StreamEngine                       # noqa: E402

emb = W.synthetic_embedding(1234)
heads = {"rnn": W.synthetic_head("rnn", 1234, kind="rnn", n_out=1)}
for S in (1, 64, 1024, 4096, 16384):
    e = StreamEngine(S, heads, emb)
    try:
        pcm = W.synthetic_pcm(S, 1280, seed=1)
        e.enable_timing(True)
        for _ in range(3):
            e.step(pcm)
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            e.step(pcm)
        dt = (time.perf_counter() - t0) / n
        kt = e.kernel_times()
        print(json.dumps({"streams": S, "ms_per_step_host_pcm": round(dt * 1e3, 3),
                          "heads_ms_per_launch": round(kt["heads"]["ms"] / max(kt["heads"]["launches"], 1), 4)}))
    finally:
        e.close()
