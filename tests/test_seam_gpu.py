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
