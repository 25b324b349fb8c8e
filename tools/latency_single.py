"""Per-call latency of the drop-in surface with ONE stream: Model.predict on 80 ms frames, Model.predict_clip on a 10 s clip, and the
bare C-ABI step (engine.step) for 1 / 16 / 256 streams.  python tools/latency_single.py"""
import sys
import time

import numpy as np
sys.path.insert(0, ".")
from openwakeword_amd import Model, weights as W
from openwakeword_amd.engine import StreamEngine

rng = np.random.default_rng(0)
m = Model(wakeword_models=["alexa"], weights="synthetic")
frames = (rng.standard_normal((1200, 1280)) * 3000).astype(np.int16)
for f in frames[:200]:
    m.predict(f)
t = []
for f in frames[200:]:
    t0 = time.perf_counter(); m.predict(f); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e6
print(f"Model.predict, 1 stream: median {np.median(t):.0f} us, p99 {np.percentile(t, 99):.0f} us")
clip = (rng.standard_normal(160000) * 3000).astype(np.int16)
m.predict_clip(clip)
t0 = time.perf_counter(); r = m.predict_clip(clip); dt = time.perf_counter() - t0
print(f"Model.predict_clip, 10 s clip ({len(r)} frames): {dt * 1e3:.1f} ms = {dt / len(r) * 1e6:.0f} us per frame")
for S in (1, 16, 256):
    for graph in (0, 1):
        eng = StreamEngine(S, {"alexa": W.synthetic_head("alexa", 1)}, W.synthetic_embedding(1))
        if graph:
            try:
                eng.use_graph(True)
            except Exception as e:       # noqa: BLE001
                print("use_graph:", e); eng.close(); continue
        pcm = (rng.standard_normal((S, 1280)) * 3000).astype(np.int16)
        out = np.empty((S, eng.n_labels), np.float32)
        for _ in range(200):
            eng.step(pcm, out)
        t = []
        for _ in range(1000):
            t0 = time.perf_counter(); eng.step(pcm, out); t.append(time.perf_counter() - t0)
        t = np.array(t) * 1e6
        print(f"engine.step, {S} streams, graph={graph}: median {np.median(t):.0f} us, p99 {np.percentile(t, 99):.0f} us")
        eng.close()
