"""The stage-level seam of INTEGRATION.md section 1, executed as printed.

The two ctypes stubs a maintainer would paste next to the reference's `onnx` / `tflite` branches (utils.py:64-93,
model.py:133-159) are cut out of INTEGRATION.md between the `stub:` markers and exec'd here; the three closures they build
(melspec_model_predict, embedding_model_predict, model_prediction_function[name]) are plugged into the streaming state machine
of the reference (its restatement, which tests/test_oracle_golden.py pins to vectors produced by the reference's own code) and
all golden cases are replayed: every score must agree with the REFERENCE-produced vector to 1e-4."""
import functools
import os
import re
import types

import numpy as np
import pytest

import cases
from oracle import oww_oracle as O
from openwakeword_amd import _build, engine as E, weights as W

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TIMER_MAP = {"timer": {"1": "1_minute_timer", "2": "5_minute_timer", "3": "10_minute_timer",
                       "4": "20_minute_timer", "5": "30_minute_timer", "6": "1_hour_timer"}}


def _stub(tag):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- stub:%s -->\s*```python\n(.*?)```" % tag, text, re.S)
    assert m, f"INTEGRATION.md has no stub:{tag} block"
    return m.group(1)


def _bind(head_names):
    """exec the two stubs with the names they expect in scope; returns (mel closure, embed closure, {name: head closure})"""
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    heads = {n: W.synthetic_head(n, cases.SEED_WEIGHTS) for n in head_names}
    obj = types.SimpleNamespace(model_prediction_function={})
    scope = dict(inference_framework="hip", ncpu=4, self=obj, libowwhip_path=_build.lib_path(),
                 mel_blob=E.pack_mel_blob(), emb_blob=E.pack_embedding_blob(emb),
                 head_blobs=[E.pack_head_blob(heads[n]) for n in head_names], functools=functools)
    exec(_stub("features"), scope)
    body = "def _install(idx, T, n_out, mdl_name):\n" + _stub("heads")
    exec(body, scope)
    for idx, n in enumerate(head_names):
        scope["_install"](idx, int(heads[n]["T"]), int(heads[n]["n_out"]), n)
    return heads, emb, obj.melspec_model_predict, obj.embedding_model_predict, obj.model_prediction_function


def _model(head_names):
    heads, emb, mel_fn, embed_fn, head_fns = _bind(head_names)
    np.random.seed(cases.SEED_NP)
    return O.OracleModel(heads, emb, class_mapping=TIMER_MAP, head_fns=head_fns, mel_fn=mel_fn,
                         embed_fn=lambda x: embed_fn(x).squeeze())


@pytest.mark.parametrize("case", cases.CLIP_CASES, ids=[c[0] for c in cases.CLIP_CASES])
def test_integration_stub_replays_reference_vectors(golden, case):
    cid, head_names, clip, kw = case
    mdl = _model(head_names)
    # the feature ring the stub's closures seeded (embeddings of 4 s of random audio, utils.py:169)
    np.testing.assert_allclose(mdl.preprocessor.features, golden["init/feature_buffer"], rtol=0, atol=2e-4)
    preds = mdl.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(golden[f"{cid}/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    want = golden[f"{cid}/scores"]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    np.testing.assert_allclose(mdl.preprocessor.features, golden[f"{cid}/features"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(mdl.preprocessor.mel_rows[-16:], golden[f"{cid}/mel_tail"], rtol=0, atol=5e-4)


def test_stub_reset_then_second_clip(golden):
    mdl = _model(cases.HEADS_BINARY)
    mdl.predict_clip(golden["pcm/alexa_test"], chunk_size=1280)
    np.random.seed(cases.SEED_NP + 1)
    mdl.reset()
    preds = mdl.predict_clip(golden["pcm/hey_mycroft_test"], chunk_size=1280)
    labels = list(golden["c1280/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    np.testing.assert_allclose(got, golden["reset/scores"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("family", [3, 1])
def test_plain_c_host_gets_the_bits_the_python_layer_gets(tmp_path, family):
    """examples/c/owwhip_demo.c -- create / load / commit / step / destroy from C99, weights and PCM from files -- against the same
    calls made through ctypes: identical scores, bit for bit (binary, gated and multiclass heads; 40 streams leave a partly filled
    tile).  What a cgo / JNI / FFI binding of include/owwhip.h would do, minus the language."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    S, T = 40, 12
    names = ["alexa", "hey_jarvis", "timer"]
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    heads = {n: W.synthetic_head(n, cases.SEED_WEIGHTS) for n in names}
    E.pack_mel_blob().tofile(tmp_path / "mel.bin")
    E.pack_embedding_blob(emb).tofile(tmp_path / "embedding.bin")
    for i, n in enumerate(names):
        E.pack_head_blob(heads[n]).tofile(tmp_path / f"head_{i}.bin")
    pcm = (np.random.default_rng(3).standard_normal((T, S, 1280)) * 5000).astype(np.int16)
    pcm[:, 1] = 0
    pcm.tofile(tmp_path / "pcm.bin")
    exe = tmp_path / "owwhip_demo"
    libdir = os.path.dirname(_build.lib_path())
    r = subprocess.run([gcc, "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c", "owwhip_demo.c"), "-L", libdir, "-lowwhip", f"-Wl,-rpath,{libdir}", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe), str(tmp_path), str(S), str(T), str(family)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr + run.stdout
    assert run.stdout.count("step ") == T
    eng = E.StreamEngine(S, heads, emb, use_mfma=family, calibration_pcm=None)      # (the C program sets no calibration audio either)
    try:
        want = np.stack([eng.step(pcm[t]) for t in range(T)])
    finally:
        eng.close()
    got = np.fromfile(tmp_path / "scores.bin", dtype=np.float32).reshape(want.shape)
    assert want.shape == (T, S, 1 + 1 + 7) or want.shape[2] == eng.n_labels
    assert np.isfinite(got).all() and got[-1].max() > 0
    np.testing.assert_array_equal(got, want)
