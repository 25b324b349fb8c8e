"""oracle/mini_ort.py -- the generic numpy ONNX evaluator behind the reference-on-real-files golden vectors -- against the one
authority available here for what a torch-exported graph means: the torch module it was exported from.  (CPU)"""
import os

import numpy as np
import pytest

import torch_export as TE
from openwakeword_amd import weights as W
from oracle import mini_ort

torch = pytest.importorskip("torch")


def _export_or_skip(fn):
    try:
        return fn()
    except Exception as e:                                  # noqa: BLE001 -- an exporter that cannot run here is not our failure
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")


@pytest.mark.parametrize("name,opset,ln", [("alexa", 17, None), ("alexa", 11, None), ("timer", 13, None), ("weather", 12, True)])
def test_head_graphs_evaluate_like_their_torch_modules(tmp_path, name, opset, ln):
    head = W.synthetic_head(name, 31, layernorm=ln)
    mod = TE.torch_head(head["net"], head["T"], head["n_out"])
    path = os.path.join(tmp_path, "h.onnx")
    _export_or_skip(lambda: TE.torch_export_head(mod, head["T"], path, opset))
    sess = mini_ort.InferenceSession(path)
    assert sess.get_inputs()[0].shape == [1, head["T"], 96] and sess.get_outputs()[0].shape == [1, head["n_out"]]    # model.py:156-157 reads these
    x = np.random.default_rng(1).normal(0, 2, (1, head["T"], 96)).astype(np.float32)
    got = sess.run(None, {sess.get_inputs()[0].name: x})[0]
    with torch.no_grad():
        want = mod(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


@pytest.mark.parametrize("opset,act", [(13, "leakyclamp"), (17, "maxmul")])
def test_embedding_graph_evaluates_like_its_torch_module(tmp_path, opset, act):
    emb = W.synthetic_embedding(32)
    mod = TE.torch_embedding(emb, act)
    path = os.path.join(tmp_path, "e.onnx")
    _export_or_skip(lambda: TE.export(mod, torch.rand(1, 76, 32, 1), path, opset, input_names=["input_1"], dynamic_axes={"input_1": {0: "b"}}))
    sess = mini_ort.InferenceSession(path)
    x = np.random.default_rng(2).normal(10, 1.5, (3, 76, 32, 1)).astype(np.float32)
    got = sess.run(None, {"input_1": x})[0]
    with torch.no_grad():
        want = mod(torch.from_numpy(x)).numpy()
    assert got.shape == want.shape == (3, 1, 1, 96)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())))


def test_melspectrogram_graph_evaluates_like_its_torch_module(tmp_path):
    mod = TE.torch_melspectrogram()
    path = os.path.join(tmp_path, "m.onnx")
    _export_or_skip(lambda: TE.export(mod, torch.rand(1, 1760) * 1000, path, 12, input_names=["input"],
                                      dynamic_axes={"input": {0: "batch", 1: "samples"}}))
    sess = mini_ort.InferenceSession(path)
    for n in (1760, 1280 * 3 + 480, 16000):                 # the dynamic sample axis is honoured
        x = np.random.default_rng(n).normal(0, 3000, (1, n)).astype(np.float32)
        got = sess.run(None, {"input": x})[0]
        with torch.no_grad():
            want = mod(torch.from_numpy(x)).numpy()
        assert got.shape == want.shape == (1, 1, (n - 512) // 160 + 1, 32)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)       # dB of int16-scale audio in fp32


@pytest.mark.parametrize("form", ["where", "if"])
def test_gated_graphs_evaluate_like_their_torch_modules(tmp_path, form):
    """Greater + Where, and an If whose then-branch is a whole network reading the OUTER graph's input."""
    head = W.synthetic_head("hey_jarvis", 33)
    mod = TE.torch_gated(head, form)
    path = os.path.join(tmp_path, "g.onnx")
    _export_or_skip(lambda: TE.torch_export_head(mod, head["T"], path, 13))
    sess = mini_ort.InferenceSession(path)
    rng = np.random.default_rng(4)
    arms = set()
    for _ in range(30):
        x = rng.normal(0, 2, (1, head["T"], 96)).astype(np.float32)
        got = sess.run(None, {sess.get_inputs()[0].name: x})[0]
        with torch.no_grad():
            want = mod(torch.from_numpy(x)).numpy()
            arms.add(bool(TE.torch_head(head["net"], head["T"], 1)(torch.from_numpy(x))[0, 0] > 0.5))
        np.testing.assert_allclose(got.reshape(want.shape), want, rtol=0, atol=2e-6)
    assert arms == {True, False}


def test_unknown_operators_and_wrong_feeds_are_errors(tmp_path):
    head = W.synthetic_head("alexa", 31)
    path = os.path.join(tmp_path, "h.onnx")
    _export_or_skip(lambda: TE.torch_export_head(TE.torch_head(head["net"], 16, 1), 16, path, 13))
    sess = mini_ort.InferenceSession(path)
    with pytest.raises(ValueError, match="feeds"):
        sess.run(None, {"nope": np.zeros((1, 16, 96), np.float32)})
    g = mini_ort.load(path)
    g["nodes"][0]["op"] = "Einsum"
    with pytest.raises(NotImplementedError, match="Einsum"):
        mini_ort.evaluate(g, {sess.get_inputs()[0].name: np.zeros((1, 16, 96), np.float32)})


def test_vad_standin_graph_evaluates_like_its_torch_module_and_the_oracle(tmp_path):
    """Two Conv1d STFT branches, Sqrt / Log, four strided Conv1d, two ONNX LSTM nodes with carried state, Gemm / MatMul decoder, mean
    over time -- the file the reference's VAD class (vad.py:60-130) is pointed at in tests/golden/make_golden_onnx.py."""
    from oracle import vad_standin as VS
    vad = W.synthetic_vad(34)
    mod = TE.torch_vad(vad)
    path = os.path.join(tmp_path, "silero_vad.onnx")
    _export_or_skip(lambda: TE.export_vad(vad, path))
    sess = mini_ort.InferenceSession(path)
    assert [i.name for i in sess.get_inputs()] == ["input", "sr", "h", "c"]
    rng = np.random.default_rng(5)
    h = np.zeros((2, 1, 64), np.float32)
    c = np.zeros((2, 1, 64), np.float32)
    ho, co = h.copy(), c.copy()
    for step in range(6):                                   # the state is carried from call to call (vad.py:121-124)
        x = (rng.normal(0, 0.05 * (1 + step), (1, 640))).astype(np.float32)
        out, h, c = sess.run(None, {"input": x, "sr": np.array(16000).astype("int64"), "h": h, "c": c})
        with torch.no_grad():
            want, hn, cn = mod(torch.from_numpy(x), torch.tensor(16000), torch.from_numpy(ho), torch.from_numpy(co))
        y, ho, co = VS.forward(vad, x, ho, co)
        np.testing.assert_allclose(out, want.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(out[:, 0], y, rtol=0, atol=2e-5)
        np.testing.assert_allclose(h, hn.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(h, ho, rtol=0, atol=2e-5)
        np.testing.assert_allclose(c, co, rtol=0, atol=2e-5)
