import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import oww_oracle as O
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine, LAYER_NEW_SHAPES
emb = W.synthetic_embedding(1234)
heads = {n: W.synthetic_head(n, 1234) for n in ["alexa"]}
eng = StreamEngine(2, heads, emb, debug_layers=True)
models = []
for s in range(2):
    noise = W.synthetic_pcm(1, 64000, seed=100 + s, rms=600.0)[0]
    m = O.OracleModel(heads, emb, init_noise=noise); models.append(m)
    eng.reset([s], m.preprocessor.features[-eng.feature_ring:])
pcm = W.synthetic_pcm(2, 1280 * 12, seed=11)
for t in range(12):
    x = pcm[:, 1280 * t: 1280 * (t + 1)]
    eng.step(x)
    for s in range(2): models[s].predict(x[s])
np.set_printoptions(linewidth=250, precision=3, suppress=True)
s = 0
h = models[s].preprocessor.mel_rows[-76:].astype(np.float64)[None, :, :, None]
for li, (kh, kw, ci, co, relu_first, bn, pool) in enumerate(O.CNN_LAYERS):
    h = O._conv(h, emb["conv"][li].astype(np.float64))
    if relu_first: h = np.maximum(h, 0.0)
    if bn:
        sc, sh = O.bn_fold(*emb["bn"][li], dtype=np.float64); h = O._activation(h * sc + sh)
    got = eng.debug_layer(s, li); want = h[0, -LAYER_NEW_SHAPES[li][0]:]
    d = np.abs(got - want)
    print("layer", li, "max err", d.max(), "per-channel max:", d.max(axis=(0, 1))[:96].round(2))
    if li < 3: print("   per-f max:", d.max(axis=(0, 2)).round(2)); print("   per-row max:", d.max(axis=(1, 2)).round(2))
    if pool: h = O._pool(h, *pool)
    if li >= 4: break
