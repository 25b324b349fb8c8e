"""The reference's other model_type, "rnn" (train.py:85-98): a 2-layer bidirectional LSTM(64) over the T feature rows, Linear(128, n) on
the last step, Sigmoid | ReLU + softmax -- owk::heads_rnn_kernel, in every kernel family, against the float64 oracle
(oracle/oww_oracle.py::_rnn_head, itself held to torch.nn.LSTM in tests/test_oracle_golden.py).  pytest -m gpu"""
import numpy as np
import pytest

import cases
from oracle import oww_oracle as O
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

pytestmark = pytest.mark.gpu
TOL_SCORE = 1e-4


@pytest.fixture(scope="module")
def emb():
    return W.synthetic_embedding(cases.SEED_WEIGHTS)


@pytest.fixture(scope="module")
def rnn_heads():
    return {"rnn1": W.synthetic_head("rnn1", 61, kind="rnn", n_out=1),
            "rnn5": W.synthetic_head("rnn5", 62, kind="rnn", n_out=5, T=34),
            "alexa": W.synthetic_head("alexa", cases.SEED_WEIGHTS)}


@pytest.mark.parametrize("use_mfma", [3, 1, 0])
def test_rnn_head_stage_matches_the_float64_oracle(emb, rnn_heads, use_mfma):
    rng = np.random.default_rng(4)
    for S in (1, 5, 37):                                   # the last wave of two streams partly filled
        e = StreamEngine(S, rnn_heads, emb, use_mfma=use_mfma)
        try:
            assert e.feature_ring == 34 and e.n_labels == 7
            for name, h in rnn_heads.items():
                for scale in (0.3, 2.0):
                    ft = rng.normal(0, scale, (S, h["T"], 96)).astype(np.float32)
                    got = e.head(name, ft)
                    want = O.head_stage(ft, h, np.float64)
                    np.testing.assert_allclose(got, want, rtol=0, atol=TOL_SCORE, err_msg=f"{name} S={S}")
                    if name == "rnn1" and S == 37 and scale == 2.0:
                        assert np.ptp(want) > 0.05                  # the comparison is not vacuous
                    if name == "rnn5":
                        np.testing.assert_allclose(got.sum(axis=1), 1.0, atol=1e-5)
        finally:
            e.close()


def test_rnn_heads_streaming_post_processing_and_masked_steps(emb, rnn_heads):
    S, n_steps = 9, 8
    e = StreamEngine(S, rnn_heads, emb)
    try:
        models = []
        for s in range(4):
            m = O.OracleModel(rnn_heads, emb, init_noise=W.synthetic_pcm(1, 64000, seed=500 + s, rms=600.0)[0],
                              class_mapping={"rnn5": {str(i): f"rnn5#{i}" for i in range(5)}})
            models.append(m)
            e.reset([s], m.preprocessor.features[-e.feature_ring:])
        labels = ["rnn1"] + [f"rnn5#{i}" for i in range(5)] + ["alexa"]
        pcm = W.synthetic_pcm(S, 1280 * n_steps, seed=3)
        for t in range(n_steps):
            x = pcm[:, 1280 * t: 1280 * (t + 1)]
            got = e.step(x)
            for s in range(4):
                p = models[s].predict(x[s])
                np.testing.assert_allclose(got[s], [p[k] for k in labels], rtol=0, atol=TOL_SCORE, err_msg=f"stream {s} step {t}")
        assert (got[:4] > 0).any()
    finally:
        e.close()
    a, b = StreamEngine(S, rnn_heads, emb), StreamEngine(S, rnn_heads, emb)
    try:
        on = np.random.default_rng(2).random((n_steps, S)) < 0.5
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
