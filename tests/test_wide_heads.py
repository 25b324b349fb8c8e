"""The wide form of the fp16-split heads kernel (owwhip_hx.h: heads_wide_tail): nets of up to 128 hidden units and up to 8 outputs,
with or without LayerNorm, sigmoid or ReLU + softmax -- the released multiclass `timer` model (T = 34: 3264 -> 128 -> 128 -> 7,
docs/models/timers.md:9-27, train.py:152-165) and train.py's default width (layer_dim = 128, train.py:67) -- on the matrix pipe
instead of the VALU kernel.  Against the float64 head of the oracle (model.py:299-302, train.py:56-83).  pytest -m gpu"""
import numpy as np
import pytest

import cases
from oracle import oww_oracle as O
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

pytestmark = pytest.mark.gpu
TOL_SCORE = 1e-4

SHAPES = {
    "timer": dict(),                                                                       # multiclass, T = 34, 128 units, 7 classes, no LayerNorm
    "ln128": dict(kind="binary", T=16, hidden=128, n_out=1, layernorm=True),               # train.py's default width
    "multi100": dict(kind="multiclass", T=16, hidden=100, n_out=8, layernorm=True),        # padded units stay out of the LayerNorm statistics
    "multi72": dict(kind="multiclass", T=5, hidden=72, n_out=3, layernorm=False),
    "plain65": dict(kind="binary", T=34, hidden=65, n_out=1, layernorm=False),             # shares T with `timer`: a two-net launch
    "multi64": dict(kind="multiclass", T=16, hidden=64, n_out=2, layernorm=True),          # 64 units but two outputs: wide form, zero-padded
}


@pytest.fixture(scope="module")
def emb():
    return W.synthetic_embedding(cases.SEED_WEIGHTS)


@pytest.fixture(scope="module")
def wide_heads():
    return {n: W.synthetic_head(n, 51 + i, **kw) for i, (n, kw) in enumerate(SHAPES.items())}


def raw_want(models, s, heads):
    out = []
    for n, h in heads.items():
        out.extend(O.head_stage(models[s].preprocessor.get_features(h["T"]), h, np.float64)[0])
    return np.array(out)


@pytest.mark.parametrize("S", [1, 37, 700])          # one partly filled wave; several workgroups; (<= 512 workgroups: the deep weight ring)
def test_wide_heads_match_the_float64_oracle(emb, wide_heads, S):
    e = StreamEngine(S, wide_heads, emb)
    try:
        info = e.calibration_info()
        assert info["selftest_score_err"] < 1e-4                    # f16-split vs the exact family's VALU kernel over the probe set
        rng = np.random.default_rng(100 + S)
        for name, h in wide_heads.items():
            for scale in (1.5, 40.0, 0.01):                         # hidden vectors of any magnitude: each stream carries its own power of two
                ft = rng.normal(0, scale, (S, h["T"], 96)).astype(np.float32)
                got = e.head(name, ft)
                want = O.head_stage(ft, h, np.float64)
                np.testing.assert_allclose(got, want, rtol=0, atol=TOL_SCORE, err_msg=f"{name} S={S} scale={scale}")
                if h["kind"] == "multiclass":
                    np.testing.assert_allclose(got.sum(axis=1), 1.0, atol=1e-5)     # softmax rows
        assert not e.range_status()
    finally:
        e.close()


def test_wide_heads_streaming_and_masked_steps(emb, wide_heads):
    """Through the feature ring (T = 34 and T = 16 nets side by side), every raw score column against per-stream oracle models; then masked
    steps: a stream that sits steps out equals, bit for bit, a private sequence of its active chunks."""
    S, n_steps = 37, 8
    e = StreamEngine(S, wide_heads, emb)
    try:
        assert e.feature_ring == 34 and e.n_labels == sum(h["n_out"] for h in wide_heads.values())
        models = []
        for s in range(5):
            m = O.OracleModel(wide_heads, emb, init_noise=W.synthetic_pcm(1, 64000, seed=900 + s, rms=600.0)[0])
            models.append(m)
            e.reset([s], m.preprocessor.features[-e.feature_ring:])
        pcm = W.synthetic_pcm(S, 1280 * n_steps, seed=17)
        for t in range(n_steps):
            x = pcm[:, 1280 * t: 1280 * (t + 1)]
            got = e.step_raw(x)
            for s in range(5):
                models[s].predict(x[s])
                np.testing.assert_allclose(got[s], raw_want(models, s, wide_heads), rtol=0, atol=TOL_SCORE, err_msg=f"stream {s} step {t}")
    finally:
        e.close()
    a, b = StreamEngine(S, wide_heads, emb), StreamEngine(S, wide_heads, emb)
    try:
        rng = np.random.default_rng(9)
        on = rng.random((n_steps, S)) < 0.4
        masked = np.stack([a.step_masked(pcm[:, 1280 * t: 1280 * (t + 1)], on[t]) for t in range(n_steps)])
        counts = on.sum(0)
        packed = np.zeros((int(counts.max()), S, 1280), np.int16)
        for s in range(S):
            packed[:counts[s], s] = pcm[s].reshape(n_steps, 1280)[on[:, s]]
        plain = np.stack([b.step(p) for p in packed])
        for s in range(S):
            ts = np.nonzero(on[:, s])[0]
            np.testing.assert_array_equal(masked[ts, s], plain[:len(ts), s], err_msg=f"stream {s}")
    finally:
        a.close(); b.close()


def test_wide_heads_do_not_depend_on_the_batch_or_the_ring(emb):
    """Batch invariance: a stream's bits are the same in a 40-stream handle (deep weight ring, one workgroup per CU) and at its place in
    a 70,000-stream handle (two-slot ring, several workgroups per CU)."""
    heads = {"timer": W.synthetic_head("timer", cases.SEED_WEIGHTS), "alexa": W.synthetic_head("alexa", cases.SEED_WEIGHTS)}
    small, big = 40, 70000
    pcm = W.synthetic_pcm(small, 1280 * 7, seed=23)
    noise = W.synthetic_pcm(big, 1280, seed=24)
    ids = np.random.default_rng(5).choice(big, small, replace=False)
    outs = []
    for S in (small, big):
        e = StreamEngine(S, heads, emb)
        try:
            steps = []
            for t in range(7):
                if S == small:
                    x = pcm[:, 1280 * t: 1280 * (t + 1)]
                else:
                    x = noise.copy()
                    x[ids] = pcm[:, 1280 * t: 1280 * (t + 1)]
                r = e.step_raw(x)
                steps.append(r if S == small else r[ids])
            outs.append(np.stack(steps))
        finally:
            e.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    assert (outs[0][-1, :, :7].sum(axis=1) > 0.999).all()


def test_default_six_models_take_mfma_launches_only(emb, monkeypatch):
    """The reference's default Model() loads all six pretrained models (model.py:84-87): five 64-unit heads (six nets) and `timer`.  With
    the wide form none of them runs on the VALU kernel: same scores (1e-4) as a handle that is forced onto it (OWW_NO_WIDE_HEADS=1)."""
    names = ["alexa", "hey_mycroft", "hey_jarvis", "hey_rhasspy", "timer", "weather"]
    heads = {n: W.synthetic_head(n, cases.SEED_WEIGHTS) for n in names}
    S = 130
    pcm = W.synthetic_pcm(S, 1280 * 8, seed=31)

    def run():
        e = StreamEngine(S, heads, emb)
        try:
            e.enable_timing(True)
            out = np.stack([e.step(pcm[:, 1280 * t: 1280 * (t + 1)]) for t in range(8)])
            return out, e.kernel_times()
        finally:
            e.close()

    a, _ = run()
    monkeypatch.setenv("OWW_NO_WIDE_HEADS", "1")
    b, _ = run()
    monkeypatch.delenv("OWW_NO_WIDE_HEADS")
    assert a.shape == (8, S, 12)
    np.testing.assert_allclose(a, b, rtol=0, atol=TOL_SCORE)
    assert (a[5:] > 0).any()
