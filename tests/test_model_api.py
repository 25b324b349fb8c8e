"""The reference's Python surface (openwakeword.Model) on the HIP library, checked against the golden vectors
that the REFERENCE's own Model / AudioFeatures code produced (tests/golden/make_golden.py) and, shaped after the
reference's tests (/root/reference/tests/test_models.py), chunk-size invariance, reset and error behaviour."""
import numpy as np
import pytest

import cases
from openwakeword_amd import weights as W
from oracle import oww_oracle as O

TOL_SCORE = 1e-4          # fp32 path; north_star budget is 1e-3


def _weights(names):
    return {"embedding": W.synthetic_embedding(cases.SEED_WEIGHTS),
            "heads": {n: W.synthetic_head(n, cases.SEED_WEIGHTS) for n in names}}


# ----------------------------------------------------------------------------- CPU: argument / error behaviour
def test_constructor_errors_match_reference():
    from openwakeword_amd import Model
    with pytest.raises(ValueError, match="Could not find pretrained model"):       # model.py:96-97
        Model(wakeword_models=["no_such_word"], weights="synthetic")
    # backend selector (model.py:112-141): "onnx" / "tflite" are handed to the reference package; it is not importable in this
    # environment (no onnxruntime), which surfaces as the reference's kind of error, never as a silent switch of backend
    with pytest.raises(ValueError, match="reference package"):
        Model(wakeword_models=["alexa"], inference_framework="onnx")
    with pytest.raises(ValueError, match="unknown inference_framework"):
        Model(wakeword_models=["alexa"], inference_framework="tensorrt")
    with pytest.raises(ValueError, match="does not exist"):                        # no silent synthetic weights
        Model(wakeword_models=["alexa"])


def test_inference_framework_passthrough_returns_the_reference_object(monkeypatch):
    """Where the reference package imports, inference_framework='onnx' / 'tflite' yield ITS Model (here: a stand-in module)."""
    import sys
    import types
    from openwakeword_amd import Model
    seen = {}

    class RefModel:
        def __init__(self, *a, **kw):
            seen.update(kw)

    fake = types.ModuleType("openwakeword")
    fake.Model = RefModel
    monkeypatch.setitem(sys.modules, "openwakeword", fake)
    m = Model(wakeword_models=["alexa"], inference_framework="tflite", vad_threshold=0.3, weights="synthetic", device=1)
    assert isinstance(m, RefModel) and not isinstance(m, Model)
    assert seen == {"wakeword_models": ["alexa"], "inference_framework": "tflite", "vad_threshold": 0.3}


def test_registry_mirrors_reference():
    import openwakeword_amd as oww
    assert list(oww.MODELS) == ["alexa", "hey_mycroft", "hey_jarvis", "hey_rhasspy", "timer", "weather"]
    assert oww.model_class_mappings["timer"]["6"] == "1_hour_timer"
    assert len(oww.get_pretrained_model_paths()) == 6


# ----------------------------------------------------------------------------- GPU: behaviour
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def mdl3():
    from openwakeword_amd import Model
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=list(cases.HEADS_BINARY), weights=_weights(cases.HEADS_BINARY))
    yield m
    m.close()


@gpu
@pytest.mark.parametrize("case", [c for c in cases.CLIP_CASES], ids=[c[0] for c in cases.CLIP_CASES])
def test_predict_clip_matches_reference_golden(golden, case):
    """Same clip, same weights, same np.random seed -> the scores the reference's own Model.predict_clip gave."""
    from openwakeword_amd import Model
    cid, head_names, clip, kw = case
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=list(head_names), weights=_weights(head_names))
    try:
        if cid == "c1280":
            np.testing.assert_allclose(m.preprocessor.get_features(41)[0], golden["init/feature_buffer"], rtol=0, atol=2e-4)
        preds = m.predict_clip(golden["pcm/" + clip], **kw)
        labels = list(golden[cid + "/labels"])
        assert sorted(preds[0].keys()) == labels
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        want = golden[cid + "/scores"]
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=TOL_SCORE)
        feats = golden[cid + "/features"]
        n = min(len(feats), 120)
        np.testing.assert_allclose(m.preprocessor.get_features(n)[0], feats[-n:], rtol=0, atol=2e-4)
        if cid == "c1280":
            np.random.seed(cases.SEED_NP + 1)
            m.reset()
            preds2 = m.predict_clip(golden["pcm/hey_mycroft_test"], chunk_size=1280)
            got2 = np.array([[float(p[k]) for k in labels] for p in preds2])
            np.testing.assert_allclose(got2, golden["reset/scores"], rtol=0, atol=TOL_SCORE)
    finally:
        m.close()


@gpu
def test_predict_argument_errors(mdl3):
    with pytest.raises(ValueError):                               # model.py:262-263
        mdl3.predict([0] * 1280)
    with pytest.raises(ValueError):                               # utils.py:195-197
        mdl3.predict(np.zeros(1280, np.float32))
    with pytest.raises(ValueError):                               # model.py:341-343
        mdl3.predict(np.zeros(1280, np.int16), patience={"alexa": 3})
    with pytest.raises(ValueError):                               # model.py:344-345
        mdl3.predict(np.zeros(1280, np.int16), patience={"alexa": 3}, threshold={"alexa": 0.5}, debounce_time=1.0)


@gpu
def test_chunk_size_invariance(golden):
    """Shaped after test_models.py:68-100: the clip maximum does not depend on how the audio was cut into calls,
    as long as the calls add up to the same 1280-sample device steps (sub-multiples of 1280 here).  Calls of
    2x1280 are NOT exactly invariant even in the reference when the weights are not the trained ones (one
    top_db clamp per call; the reference's own Model gives 0.95190 vs 0.95126 on this clip with these synthetic
    weights, golden c1280 vs c2560) -- that case is pinned by test_predict_clip_matches_reference_golden."""
    from openwakeword_amd import Model
    clip = golden["pcm/alexa_test"]
    finals = []
    for chunk in (1280, 640, 320):
        np.random.seed(3)
        m = Model(wakeword_models=["alexa"], weights="synthetic")
        try:
            preds = m.predict_clip(clip, chunk_size=chunk)
            finals.append(max(p["alexa"] for p in preds))
        finally:
            m.close()
    np.testing.assert_allclose(finals, finals[0], rtol=0, atol=1e-5)


@gpu
def test_timing_and_attributes(mdl3):
    out, timing = mdl3.predict(np.zeros(1280, np.int16), timing=True)
    assert set(out) == set(cases.HEADS_BINARY) and "preprocessor" in timing["models"]
    assert mdl3.model_inputs == {n: 16 for n in cases.HEADS_BINARY}
    assert mdl3.model_outputs == {n: 1 for n in cases.HEADS_BINARY}
    assert mdl3.preprocessor.get_features(16).shape == (1, 16, 96)
    assert all(len(v) <= 30 for v in mdl3.prediction_buffer.values())


@gpu
def test_batched_model_matches_single_stream_models(golden):
    from openwakeword_amd import BatchedModel, Model
    names = list(cases.HEADS_BINARY)
    w = _weights(names)
    S, steps = 5, 12
    pcm = W.synthetic_pcm(S, 1280 * steps, seed=5)
    pcm[0, : len(golden["pcm/alexa_test"])] = golden["pcm/alexa_test"][: 1280 * steps]
    bm = BatchedModel(S, names, weights=w)
    singles = []
    try:
        for s in range(S):
            np.random.seed(100 + s)
            m = Model(wakeword_models=names, weights=w)
            singles.append(m)
            bm.reset([s], m.preprocessor.get_features(16)[0])
        bm.set_postproc(debounce_time=0.3, threshold={"alexa": 0.4, "hey_mycroft": 0.5})
        for t in range(steps):
            x = pcm[:, 1280 * t: 1280 * (t + 1)]
            got = bm.predict_batch(x)
            for s in range(S):
                want = singles[s].predict(x[s], debounce_time=0.3, threshold={"alexa": 0.4, "hey_mycroft": 0.5})
                np.testing.assert_allclose(got[s], [want[k] for k in bm.labels], rtol=0, atol=TOL_SCORE)
    finally:
        bm.close()
        for m in singles:
            m.close()


@gpu
def test_embed_clips_matches_oracle(golden):
    """AudioFeatures.embed_clips (utils.py:354-385): per-clip mel clamp, 76-row windows every 8 rows."""
    from oracle import oww_oracle as O
    from openwakeword_amd import Model
    w = _weights(["alexa"])
    m = Model(wakeword_models=["alexa"], weights=w)
    try:
        clip = golden["pcm/hey_jane"]
        x = np.stack([clip[:16000], clip[8000:24000] // 4, np.zeros(16000, np.int16), W.synthetic_pcm(1, 16000, seed=9)[0]])
        got = m.preprocessor.embed_clips(x, batch_size=3)           # two device batches
        oracle = O.OracleAudioFeatures(w["embedding"], init_noise=np.zeros(64000, np.int16))
        want = np.stack([oracle.clip_embeddings(c) for c in x])
        assert got.shape == want.shape == (4, 3, 96)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
        # the object streams normally afterwards
        out = m.predict(np.zeros(1280, np.int16))
        assert set(out) == {"alexa"}
    finally:
        m.close()


@gpu
def test_stateless_feature_helpers_mirror_the_reference(golden):
    """The helpers training / data-preparation code calls on AudioFeatures directly (utils.py:180-352): same shapes, same
    clamp scope (one floor per _get_melspectrogram call, one per clip in _get_melspectrogram_batch), and none of them moves
    the streaming state."""
    from openwakeword_amd import Model
    w = _weights(["alexa"])
    m = Model(wakeword_models=["alexa"], weights=w)
    try:
        F = m.preprocessor
        oracle = O.OracleAudioFeatures(w["embedding"], init_noise=np.zeros(64000, np.int16))
        clip = golden["pcm/hey_jane"]
        x = np.stack([clip[:24123], clip[4000:28123] // 8, np.zeros(24123, np.int16)])
        # one clip, a list, a batch (ONE floor for the batch: the quiet clip is clamped against the loud one's maximum)
        one = F._get_melspectrogram(x[0])
        assert one.shape == (int(np.ceil(24123 / 160 - 3)), 32)
        np.testing.assert_allclose(one, oracle.melspectrogram(x[0]), rtol=0, atol=2e-4)
        np.testing.assert_array_equal(F._get_melspectrogram(list(x[0][:4000])), F._get_melspectrogram(x[0][:4000]))
        both = F._get_melspectrogram(x[:2], melspec_transform=lambda v: v)
        assert both.shape == (2, 148, 32)
        np.testing.assert_allclose(both, np.squeeze(oracle.mel_fn(x[:2].astype(np.float32))), rtol=0, atol=2e-3)
        with pytest.raises(ValueError, match="16-bit integers"):
            F._get_melspectrogram(x[0].astype(np.float32))
        # per-clip floors
        mb = F._get_melspectrogram_batch(x, batch_size=2)
        assert mb.shape == (3, 148, 32) and mb.dtype == np.float32
        for i in range(3):
            np.testing.assert_allclose(mb[i], oracle.melspectrogram(x[i]), rtol=0, atol=2e-4)
        # windows -> embeddings
        eb = F._get_embeddings_batch(mb[:, :, :, None])
        assert eb.shape == (3, (148 - 76) // 8 + 1, 96)
        want = np.stack([oracle.clip_embeddings(c) for c in x])
        np.testing.assert_allclose(eb, want, rtol=0, atol=2e-4)
        np.testing.assert_allclose(F.embed_clips(x[:, :24000]), F._get_embeddings_batch(F._get_melspectrogram_batch(x[:, :24000])), rtol=0, atol=2e-5)
        one_win = F._get_embeddings_from_melspec(mb[1, 8:84, :, None])
        assert one_win.shape == (96,)
        np.testing.assert_allclose(one_win, eb[1, 1], rtol=0, atol=2e-5)
        with pytest.raises(ValueError, match="at least 76 frames"):
            F._get_embeddings_batch(mb[:, :70])
        assert F._get_embeddings(x[0]).shape == F.get_embedding_shape(24123 / 16000)
    finally:
        m.close()
    # (embed_clips re-seeds the object like the reference's constructor does; the others left the ring alone)
    m = Model(wakeword_models=["alexa"], weights=w)
    try:
        F = m.preprocessor
        m.predict_clip(golden["pcm/hey_jane"][:12800])
        before = F.get_features(16).copy()
        F._get_melspectrogram(x[0]); F._get_melspectrogram_batch(x); F._get_embeddings_batch(mb); F._get_embeddings_from_melspec(mb[0, :76])
        np.testing.assert_array_equal(F.get_features(16), before)
    finally:
        m.close()


# ----------------------------------------------------------------------------- the rest of the reference's test list
# (/root/reference/tests/test_models.py: custom verifier 114-128, names with spaces 130-137, label mapping 139-149,
#  parent lookup 318-321, positive frames 323-330, VAD 259-285)
class _Verifier:
    """Stands in for the pickled scikit-learn pipeline of custom_verifier_model.py:95-113."""

    def predict_proba(self, feats):
        assert feats.shape == (1, 16, 96)
        v = float(np.tanh(np.abs(feats).mean()))
        return np.array([[1.0 - v, v]])


@gpu
@pytest.mark.parametrize("max_chunks", [1, 2, 32])
def test_call_longer_than_max_chunks_keeps_the_calls_clamp_floor(tmp_path, golden, max_chunks):
    """cases.ONNX_LONG on the HIP Model, files loaded by path: 5-chunk predict() calls whose loudest frame sits in the LAST (or first)
    chunk, through handles that hold 1, 2 or 32 chunks of mel rows.  The reference runs its melspectrogram graph once per call
    (utils.py:387-401), so the clamp floor is the call's; oww_step evaluates a call longer than max_chunks in slices behind one pass
    of the mel kernel that finds that maximum -- scores, the feature ring and the (clamped) mel rows of the last slice against the
    reference's own run on the same files, 1e-4; and the three handle sizes agree with each other to fp32 round-off."""
    pytest.importorskip("torch")
    import os
    import torch_export as TE
    from openwakeword_amd import Model
    cid, head_names, sizes = cases.ONNX_LONG
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    try:
        paths = TE.export_reference_files(str(tmp_path), cases.onnx_file_weights(), head_opsets=cases.ONNX_HEAD_OPSETS)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    x = cases.long_call_pcm(golden["pcm/alexa_test"])
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=[paths[n] for n in head_names], melspec_model_path=paths["melspectrogram"],
              embedding_model_path=paths["embedding_model"], max_chunks=max_chunks)
    try:
        labels = list(ref[f"{cid}/labels"])
        rows, o = [], 0
        for n in sizes:
            p = m.predict(x[o:o + n])
            o += n
            rows.append([float(p[k]) for k in labels])
            k_last = 5 if max_chunks >= 5 else (5 % max_chunks or max_chunks)          # chunks of a 5-chunk call's last slice
            if n == 6400 and o == 1280 * 6 + 6400 and k_last > 1:          # the call with the loud LAST chunk: its quiet rows sit on the call's floor
                # (a one-chunk slice leaves no rows to read back: oww_get_mel refuses, the fused front end keeps them in registers)
                got_mel = m._engine.get_mel(0, 8 * k_last)
                want_mel = ref[f"{cid}/mel_tail"]           # rows of the last 15 chunks; this call ends 9 chunks before the end
                want = want_mel[len(want_mel) - 8 * 9 - 8 * k_last: len(want_mel) - 8 * 9]
                np.testing.assert_allclose(got_mel, want, rtol=0, atol=5e-4)
        np.testing.assert_allclose(np.array(rows), ref[f"{cid}/scores"], rtol=0, atol=TOL_SCORE)
        feats = ref[f"{cid}/features"]
        n = min(len(feats), 120)
        np.testing.assert_allclose(m.preprocessor.get_features(n)[0], feats[-n:], rtol=0, atol=2e-4)
        assert m.preprocessor.accumulated_samples == 0
    finally:
        m.close()


def _flatten_features(x):                       # custom_verifier_model.py:91-92
    return [i.flatten() for i in x]


def _trained_verifier(seed=0):
    """The pipeline custom_verifier_model.train_verifier_model builds (custom_verifier_model.py:95-113), fitted on random data."""
    from sklearn.linear_model import LogisticRegression
    from sklearn.pipeline import make_pipeline
    from sklearn.preprocessing import FunctionTransformer, StandardScaler
    r = np.random.default_rng(seed)
    X = r.normal(0, 4.0, (60, 16, 96)).astype(np.float32)
    y = (X[:, :, :8].mean(axis=(1, 2)) > 0).astype(int)
    clf = LogisticRegression(random_state=0, max_iter=2000, C=0.001)
    pipe = make_pipeline(FunctionTransformer(_flatten_features), StandardScaler(), clf)
    pipe.fit(X, y)
    return pipe


def test_fold_verifier_equals_predict_proba():
    from openwakeword_amd.model import fold_verifier
    pipe = _trained_verifier()
    w, b = fold_verifier(pipe)
    X = np.random.default_rng(5).normal(0, 4.0, (7, 16, 96)).astype(np.float32)
    want = pipe.predict_proba(X)[:, -1]
    got = 1.0 / (1.0 + np.exp(-(X.reshape(7, -1).astype(np.float64) @ w.astype(np.float64) + b)))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        fold_verifier(object())


@gpu
def test_batched_custom_verifier_on_device(golden):
    """(f)4: the verifier of model.py:320-328 for every stream of a BatchedModel, as one dot product per stream on the device,
    against OracleModel with the scikit-learn pipeline wrapped around its head function."""
    from openwakeword_amd import BatchedModel
    pipe = _trained_verifier()
    names = ["alexa", "hey_mycroft"]
    w = _weights(names)
    thr = 0.1
    S, n_frames = 5, 12
    pcm = np.stack([np.resize(golden["pcm/" + k], n_frames * 1280) for k in ("alexa_test", "hey_mycroft_test", "hey_jane", "alexa_test", "hey_jane")])
    pcm[3] = np.roll(pcm[3], 4000); pcm[4] //= 3
    noise = W.synthetic_pcm(1, 64000, seed=3, rms=600.0)[0]

    def verified(head):
        def fn(x):
            sc = O.head_stage(x, head, np.float32)
            if sc[0, 0] >= thr:
                sc = np.array([[pipe.predict_proba(x)[0][-1]]], np.float32)
            return [sc]
        return fn

    bm = BatchedModel(S, names, weights=w)
    try:
        bm.set_custom_verifier("alexa", pipe, threshold=thr)
        with pytest.raises(ValueError, match="not matched"):
            bm.set_custom_verifier("no_such_model", pipe)
        models = [O.OracleModel(w["heads"], w["embedding"], init_noise=noise,
                                head_fns={"alexa": verified(w["heads"]["alexa"]),
                                          "hey_mycroft": (lambda x, _h=w["heads"]["hey_mycroft"]: [O.head_stage(x, _h, np.float32)])})
                  for _ in range(S)]
        bm.reset(None, models[0].preprocessor.features[-bm.engine.feature_ring:])
        n_rescored = 0
        for t in range(n_frames):
            x = pcm[:, t * 1280:(t + 1) * 1280]
            got = bm.predict_batch(x)
            for s, m in enumerate(models):
                pred = m.predict(x[s])
                np.testing.assert_allclose(got[s], [pred[k] for k in names], rtol=0, atol=1e-4, err_msg=f"stream {s} frame {t}")
            plain = np.array([O.head_stage(m.preprocessor.get_features(16), w["heads"]["alexa"], np.float32)[0, 0] for m in models])
            n_rescored += int((plain >= thr).sum())
        assert n_rescored > 0
        bm.set_custom_verifier("alexa", None)                    # removed: plain head scores again
        got = bm.predict_batch(pcm[:, :1280])
        for s, m in enumerate(models):
            m.head_fns["alexa"] = (lambda x, _h=w["heads"]["alexa"]: [O.head_stage(x, _h, np.float32)])
            pred = m.predict(pcm[s, :1280])
            np.testing.assert_allclose(got[s], [pred[k] for k in names], rtol=0, atol=1e-4)
    finally:
        bm.close()


@gpu
def test_custom_verifier_rescoring(tmp_path, golden):
    """model.py:320-328: a verifier re-scores frames whose base score reaches custom_verifier_threshold, from the same
    last-16-rows features the head saw; keys that match no model are an error (model.py:189-195)."""
    import pickle
    from openwakeword_amd import Model
    path = str(tmp_path / "verifier.pkl")
    pickle.dump(_Verifier(), open(path, "wb"))
    w = _weights(["alexa"])
    with pytest.raises(ValueError, match="not matched"):
        Model(wakeword_models=["alexa"], weights=w, custom_verifier_models={"nope": path})
    np.random.seed(cases.SEED_NP)
    plain = Model(wakeword_models=["alexa"], weights=w)
    np.random.seed(cases.SEED_NP)
    ver = Model(wakeword_models=["alexa"], weights=w, custom_verifier_models={"alexa": path}, custom_verifier_threshold=0.0)
    try:
        clip = golden["pcm/alexa_test"]
        for o in range(0, 1280 * 12, 1280):
            a = plain.predict(clip[o:o + 1280])["alexa"]
            b = ver.predict(clip[o:o + 1280])["alexa"]
            if len(plain.prediction_buffer["alexa"]) <= 5:
                assert a == 0.0 and b == 0.0                            # first-five zeroing comes after the verifier
            else:
                want = _Verifier().predict_proba(ver.preprocessor.get_features(16))[0][-1]
                assert b == pytest.approx(want, abs=1e-6) and b != a
    finally:
        plain.close()
        ver.close()


@gpu
def test_names_label_mapping_parent_lookup_and_positive_frames(tmp_path, golden):
    import wave
    from openwakeword_amd import Model
    w = _weights(["alexa", "hey_mycroft", "timer"])
    # names with spaces resolve like the reference's substring match and stay the keys the user passed (model.py:84-100)
    m = Model(wakeword_models=["alexa", "hey mycroft"], weights=w | {"heads": {}, "seed": cases.SEED_WEIGHTS})
    try:
        out = m.predict(np.random.randint(-1000, 1000, 1280).astype(np.int16))
        assert set(out) == {"alexa", "hey mycroft"}
    finally:
        m.close()
    # a caller-provided class mapping replaces the label (model.py:177-182), multiclass parents resolve (215-224)
    m = Model(wakeword_models=["alexa", "timer"], weights=w, class_mapping_dicts=[{"alexa": {"0": "positive"}}])
    try:
        assert m.class_mapping["alexa"] == {"alexa": {"0": "positive"}}     # the reference stores the whole entry (model.py:177-178)
        assert m.get_parent_model_from_label("1_minute_timer") == "timer"
        assert m.get_parent_model_from_label("alexa") == "alexa"
        assert m.get_parent_model_from_label("no such label") == ""
        out = m.predict(np.zeros(1280, np.int16))
        assert "alexa" in out and "1_hour_timer" in out and len(out) == 1 + 6      # class 0 of `timer` carries no label
        # _get_positive_prediction_frames (model.py:428-479): with threshold 0 every fed frame qualifies
        path = str(tmp_path / "clip.wav")
        with wave.open(path, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes(golden["pcm/hey_jane"][:1280 * 9].tobytes())
        m.reset()
        pos = m._get_positive_prediction_frames(path, threshold=0.0)
        assert pos["alexa"].shape == (8, 16, 96) and pos["1_hour_timer"].shape == (8, 34, 96)
        m.reset()
        assert m._get_positive_prediction_frames(path, threshold=2.0) == {}
        m.reset()
        aud = m._get_positive_prediction_frames(path, threshold=0.0, return_type="audio")
        assert aud == {}                                                # less than 4 s of context: nothing collected
    finally:
        m.close()
    with pytest.raises(ValueError, match="vad_session"):                # no VAD network available: refused, not ignored
        Model(wakeword_models=["alexa"], weights=w, vad_threshold=0.5)


@gpu
@pytest.mark.parametrize("case", cases.VAD_CASES, ids=[c[0] for c in cases.VAD_CASES])
def test_vad_gate_matches_reference_golden(golden, case):
    """Row I, the part that can be pinned: `Model(vad_threshold=..., vad_session=...)` against the reference's own Model with
    the same pseudo VAD network behind its VAD class (tests/golden/make_golden_vad.py): gated scores, the VAD ring, the
    ungated score ring, and reset() leaving the VAD alone."""
    import os
    from oracle.pseudo_vad import PseudoVadSession
    from openwakeword_amd import Model
    gv = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_vad.npz")))
    cid, head_names, clip, kw, thr = case
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=list(head_names), weights=_weights(head_names), vad_threshold=thr, vad_session=PseudoVadSession())
    try:
        preds = m.predict_clip(golden["pcm/" + clip], **kw)
        labels = list(gv[f"{cid}/labels"])
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        np.testing.assert_allclose(got, gv[f"{cid}/scores"], rtol=0, atol=TOL_SCORE)
        np.testing.assert_allclose(np.array(m.vad.prediction_buffer), gv[f"{cid}/vad"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.array([list(m.prediction_buffer[k]) for k in labels]), gv[f"{cid}/ring"], rtol=0, atol=TOL_SCORE)
        if cid == "vad03":
            np.random.seed(cases.SEED_NP + 1)
            m.reset()
            preds = m.predict_clip(golden["pcm/hey_mycroft_test"], chunk_size=1280)
            got = np.array([[float(p[k]) for k in labels] for p in preds])
            np.testing.assert_allclose(got, gv["vadreset/scores"], rtol=0, atol=TOL_SCORE)
            _, timing = m.predict(np.zeros(1280, np.int16), timing=True)
            assert "vad" in timing["models"]
    finally:
        m.close()


@gpu
def test_batched_vad_gate_matches_single_stream_models(golden):
    """The device-side gate (oww_push_vad / oww_set_vad_threshold) on S streams == S reference-style Models with the VAD
    wrapper, fed the same audio; the per-stream VAD scores come from the same pseudo network on the host."""
    from oracle.pseudo_vad import PseudoVadSession
    from openwakeword_amd import Model, BatchedModel, VAD
    names = ["alexa", "hey_mycroft"]
    w = _weights(names)
    clips = [golden["pcm/alexa_test"], golden["pcm/hey_jane"], golden["pcm/hey_mycroft_test"]]
    steps = 26
    S = len(clips)
    pcm = np.zeros((S, 1280 * steps), np.int16)
    for i, c in enumerate(clips):                                         # staggered starts, silence around the speech
        o = 1280 * (2 + 3 * i)
        n = min(len(c), pcm.shape[1] - o)
        pcm[i, o:o + n] = c[:n]
    singles, vads = [], []
    for s in range(S):
        np.random.seed(cases.SEED_NP + s)
        singles.append(Model(wakeword_models=names, weights=w, vad_threshold=0.5, vad_session=PseudoVadSession()))
        vads.append(VAD(session=PseudoVadSession()))
    bm = BatchedModel(S, names, weights=w)
    try:
        for s in range(S):                                                # same feature-ring seeds as the single models
            bm.reset([s], singles[s].preprocessor.get_features(bm.engine.feature_ring)[0])
        bm.set_vad_threshold(0.5)
        gated = 0
        for t in range(steps):
            x = np.ascontiguousarray(pcm[:, 1280 * t: 1280 * (t + 1)])
            for s in range(S):
                vads[s](x[s])
            bm.push_vad(np.array([v.prediction_buffer[-1] for v in vads], np.float32))
            got = bm.predict_batch(x)
            for s in range(S):
                want = singles[s].predict(x[s])
                np.testing.assert_allclose(got[s], [want[k] for k in bm.labels], rtol=0, atol=TOL_SCORE)
                gated += int(all(v == 0.0 for v in want.values()) and t >= 5)
        assert gated > 0
        bm.set_vad_threshold(0.0)                                         # gate off: scores flow again
        assert bm.predict_batch(np.ascontiguousarray(pcm[:, :1280])).shape == (S, 2)
    finally:
        bm.close()
        for m in singles:
            m.close()


@gpu
def test_models_loaded_by_path_from_files_written_by_pytorchs_exporter(tmp_path, golden):
    """/root/reference/tests/test_models.py:50-66 (load models by path) with files a user would actually have: wake-word heads,
    embedding network and melspectrogram graph written by torch.onnx.export (tests/test_onnx_ingest.py: the exporter the reference
    uses, train.py:144-165), named through the reference's own keyword arguments (wakeword_models=[paths], embedding_model_path,
    melspec_model_path: model.py:38-60, utils.py:38-44).  Scores must match the oracle on the SOURCE weights."""
    pytest.importorskip("torch")
    import torch_export as T
    from openwakeword_amd import Model
    w = _weights(["alexa", "timer"])
    paths = {}
    try:
        for n in ("alexa", "timer"):
            paths[n] = str(tmp_path / f"{n}_custom.onnx")
            T.torch_export_head(T.torch_head(w["heads"][n]["net"], w["heads"][n]["T"], w["heads"][n]["n_out"]), w["heads"][n]["T"], paths[n], 13)
        import io, warnings, torch
        from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
        keep = onnx_proto_utils._add_onnxscript_fn
        onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
        try:
            for name, module, x in (("embedding_model", T.torch_embedding(w["embedding"]), torch.rand(1, 76, 32, 1)),
                                    ("melspectrogram", T.torch_melspectrogram(), torch.rand(1, 1760) * 1000)):
                buf = io.BytesIO()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    torch.onnx.export(module, x, buf, opset_version=13, dynamo=False,
                                      dynamic_axes={"x": {0: "b", 1: "n"}} if name == "melspectrogram" else None,
                                      input_names=["x"] if name == "melspectrogram" else None)
                paths[name] = str(tmp_path / f"{name}.onnx")
                open(paths[name], "wb").write(buf.getvalue())
        finally:
            onnx_proto_utils._add_onnxscript_fn = keep
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    m = Model(wakeword_models=[paths["alexa"], paths["timer"]], embedding_model_path=paths["embedding_model"],
              melspec_model_path=paths["melspectrogram"], ncpu=2, device="gpu")
    try:
        assert set(m.models) == {"alexa_custom", "timer_custom"} and m.model_outputs == {"alexa_custom": 1, "timer_custom": 7}
        clip = golden["pcm/hey_jane"]
        got = m.predict_clip(clip)
    finally:
        m.close()
    # against a model built from the SOURCE weights (same seed for the random audio that seeds the feature ring, utils.py:169)
    m2 = Model(wakeword_models=[paths["alexa"], paths["timer"]], embedding_model_path=paths["embedding_model"], melspec_model_path=paths["melspectrogram"])
    try:
        src = Model(wakeword_models=["alexa_custom", "timer_custom"],
                    weights={"embedding": w["embedding"], "heads": {"alexa_custom": w["heads"]["alexa"], "timer_custom": w["heads"]["timer"]}})
        try:
            np.random.seed(5); m2.reset(); a = m2.predict_clip(clip)
            np.random.seed(5); src.reset(); b = src.predict_clip(clip)
        finally:
            src.close()
    finally:
        m2.close()
    assert len(a) == len(b) == len(got) and len(a) > 20
    for fa, fb in zip(a, b):
        assert set(fa) == set(fb)
        for k in fa:
            assert abs(fa[k] - fb[k]) <= 1e-4, (k, fa[k], fb[k])       # (the exporter folded BatchNorm into the convolutions: fp32 round-off)
    with pytest.raises(ValueError, match="tflite"):
        Model(wakeword_models=[paths["alexa"]], embedding_model_path=str(tmp_path / "embedding_model.tflite"))
    with pytest.raises(TypeError, match="unexpected keyword"):
        Model(wakeword_models=[paths["alexa"]], embedding_model_path=paths["embedding_model"], melspec_path="x")
    bad = str(tmp_path / "mel_hop128.onnx")
    T_mod = T.torch_melspectrogram(hop=128)
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    try:
        buf = io.BytesIO()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(T_mod, torch.rand(1, 1760), buf, opset_version=13, dynamo=False)
        open(bad, "wb").write(buf.getvalue())
    finally:
        onnx_proto_utils._add_onnxscript_fn = keep
    with pytest.raises(ValueError, match="stride"):          # a front end the analytic kernel does not compute is refused, not ignored
        Model(wakeword_models=[paths["alexa"]], embedding_model_path=paths["embedding_model"], melspec_model_path=bad)


@gpu
@pytest.mark.parametrize("case", cases.ONNX_FILE_CASES, ids=[c[0] for c in cases.ONNX_FILE_CASES])
def test_hip_model_on_exported_files_matches_the_reference_on_the_same_files(tmp_path, golden, case):
    """The drop-in claim on real FILES: heads, embedding network and melspectrogram graph written by PyTorch's exporter, loaded by path
    through the reference's own keyword arguments -- against what the reference's own code returned for the same files and clips
    (tests/golden/make_golden_onnx.py: `openwakeword.Model(..., inference_framework="onnx")` over a generic ONNX evaluator).  1e-4."""
    pytest.importorskip("torch")
    import os
    import torch_export as TE
    from openwakeword_amd import Model
    cid, head_names, clip, kw = case
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    try:
        paths = TE.export_reference_files(str(tmp_path), cases.onnx_file_weights(), head_opsets=cases.ONNX_HEAD_OPSETS)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=[paths[n] for n in head_names], melspec_model_path=paths["melspectrogram"],
              embedding_model_path=paths["embedding_model"])
    try:
        assert sorted(m.models) == sorted(head_names)
        preds = m.predict_clip(golden["pcm/" + clip], **kw)
        labels = list(ref[f"{cid}/labels"])
        assert sorted(preds[0].keys()) == labels
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        want = ref[f"{cid}/scores"]
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=TOL_SCORE)
        feats = ref[f"{cid}/features"]
        n = min(len(feats), 120)
        np.testing.assert_allclose(m.preprocessor.get_features(n)[0], feats[-n:], rtol=0, atol=2e-4)
    finally:
        m.close()


@gpu
def test_hip_model_on_exported_files_ragged_and_empty_calls(tmp_path, golden):
    """cases.ONNX_SEQUENCE: direct predict() calls of 0 ... 5000 samples, files loaded by path, against the reference's own scores."""
    pytest.importorskip("torch")
    import os
    import torch_export as TE
    from openwakeword_amd import Model
    cid, head_names, clip, sizes = cases.ONNX_SEQUENCE
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    try:
        paths = TE.export_reference_files(str(tmp_path), cases.onnx_file_weights(), head_opsets=cases.ONNX_HEAD_OPSETS)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=[paths[n] for n in head_names], melspec_model_path=paths["melspectrogram"],
              embedding_model_path=paths["embedding_model"])
    try:
        labels = list(ref[f"{cid}/labels"])
        rows, o = [], 0
        for n in sizes:
            p = m.predict(golden["pcm/" + clip][o:o + n])
            o += n
            assert sorted(p.keys()) == labels
            rows.append([float(p[k]) for k in labels])
        np.testing.assert_allclose(np.array(rows), ref[f"{cid}/scores"], rtol=0, atol=TOL_SCORE)
    finally:
        m.close()


@gpu
@pytest.mark.parametrize("case", cases.ONNX_VAD_CASES, ids=[c[0] for c in cases.ONNX_VAD_CASES])
def test_vad_file_host_session_and_device_network_match_the_reference_vad_class(tmp_path, golden, case):
    """Row I on a voice-activity FILE (the stand-in network written by PyTorch's exporter; tests/golden/make_golden_onnx.py ran the
    reference's own VAD class on it): (a) the HIP Model with a host session evaluating the file, (b) for 1280-sample chunks the
    network ON THE DEVICE -- `onnx_ingest.load_vad(file)` -> oww_load_vad -> vad_front / vad_lstm kernels + the gate inside the step
    (configs[4]) -- both against the reference's gated scores; decisions whose VAD window maximum lies within 1e-3 of the threshold
    are not compared for (b)."""
    pytest.importorskip("torch")
    import os
    import torch_export as TE
    from oracle import mini_ort
    from openwakeword_amd import Model, BatchedModel, onnx_ingest
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, clip, kw, thr = case
    path = str(tmp_path / "silero_vad.onnx")
    try:
        TE.export_vad(W.synthetic_vad(cases.ONNX_VAD_SEED), path)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    src = cases.onnx_file_weights()
    w = {"embedding": src["embedding"], "heads": {n: src["heads"][n] for n in head_names}}
    labels = list(ref[f"{cid}/labels"])
    want, vad_ring = ref[f"{cid}/scores"], ref[f"{cid}/vad"]
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=list(head_names), weights=w, vad_threshold=thr, vad_session=mini_ort.InferenceSession(path))
    try:
        seed_features = m.preprocessor.get_features(120)[0].copy()
        preds = m.predict_clip(golden["pcm/" + clip], **kw)
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        np.testing.assert_allclose(np.array(m.vad.prediction_buffer), vad_ring, rtol=0, atol=1e-6)
        np.testing.assert_allclose(got, want, rtol=0, atol=TOL_SCORE)
    finally:
        m.close()
    if kw.get("chunk_size") != 1280:
        return
    bm = BatchedModel(3, list(head_names), weights=w, vad_weights=onnx_ingest.load_vad(path), vad_threshold=thr)
    try:
        assert bm.labels == labels
        for s in range(3):
            bm.reset([s], seed_features[-bm.engine.feature_ring:])
        data = golden["pcm/" + clip]
        pad = kw.get("padding", 1)
        if pad:
            data = np.concatenate((np.zeros(16000 * pad, np.int16), data, np.zeros(16000 * pad, np.int16)))
        compared = skipped = 0
        for t, o in enumerate(range(0, len(data) - 1280, 1280)):
            x = np.ascontiguousarray(np.tile(data[o:o + 1280], (3, 1)))
            got = bm.predict_batch(x)
            np.testing.assert_allclose(bm.engine.get_vad()[0], vad_ring[t], rtol=0, atol=2e-5)       # the device network == the file
            window = vad_ring[max(0, t + 1 - 7): max(0, t + 1 - 4)]                                  # vad.py / model.py:366-381: ring[-7:-4]
            if len(window) and abs(window.max() - thr) < 1e-3:
                skipped += 1
                continue
            np.testing.assert_allclose(got[1], want[t], rtol=0, atol=TOL_SCORE, err_msg=f"frame {t}")
            compared += 1
        assert compared > 8 and (want > 0).any() and (want == 0).any()
    finally:
        bm.close()


@gpu
def test_custom_verifier_on_exported_files_host_hook_and_device_dot_product(tmp_path, golden):
    """(f)4 against the reference's own code: `Model(custom_verifier_models={name: pickle})` on the exporter-written files
    (cases.ONNX_VERIFIER, tests/golden/make_golden_onnx.py) -- (a) the HIP Model with the pickled pipeline as host hook, files by
    path; (b) BatchedModel with the pipeline folded into one dot product per stream on the device (oww_set_verifier)."""
    pytest.importorskip("torch")
    pytest.importorskip("sklearn")
    import os
    import torch_export as TE
    import verifier_fixture
    from openwakeword_amd import Model, BatchedModel
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, clip, kw, target, vthr = cases.ONNX_VERIFIER
    try:
        paths = TE.export_reference_files(str(tmp_path), cases.onnx_file_weights(), head_opsets=cases.ONNX_HEAD_OPSETS)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    pkl = verifier_fixture.write(str(tmp_path / "verifier.pkl"))
    labels = list(ref[f"{cid}/labels"])
    want = ref[f"{cid}/scores"]
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=[paths[n] for n in head_names], melspec_model_path=paths["melspectrogram"],
              embedding_model_path=paths["embedding_model"], custom_verifier_models={target: pkl}, custom_verifier_threshold=vthr)
    try:
        seed_features = m.preprocessor.get_features(120)[0].copy()
        preds = m.predict_clip(golden["pcm/" + clip], **kw)
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        np.testing.assert_allclose(got, want, rtol=0, atol=TOL_SCORE)
    finally:
        m.close()
    src = cases.onnx_file_weights()
    bm = BatchedModel(2, list(head_names), weights={"embedding": src["embedding"], "heads": {n: src["heads"][n] for n in head_names}})
    try:
        assert bm.labels == labels
        bm.set_custom_verifier(target, verifier_fixture.trained_verifier(), threshold=vthr)
        bm.reset(None, seed_features[-bm.engine.feature_ring:])
        data = np.concatenate((np.zeros(16000, np.int16), golden["pcm/" + clip], np.zeros(16000, np.int16)))
        for t, o in enumerate(range(0, len(data) - 1280, 1280)):
            got = bm.predict_batch(np.ascontiguousarray(np.tile(data[o:o + 1280], (2, 1))))
            np.testing.assert_allclose(got[1], want[t], rtol=0, atol=TOL_SCORE, err_msg=f"frame {t}")
    finally:
        bm.close()


@gpu
def test_mapping_parent_lookup_and_positive_frames_on_exported_files(tmp_path, golden):
    """class_mapping_dicts, get_parent_model_from_label and _get_positive_prediction_frames (model.py:176-182, 215-224, 428-479): the
    HIP Model on exporter-written files by path against the reference's own run on the same files (cases.ONNX_MAPPING)."""
    pytest.importorskip("torch")
    import torch_export as TE
    from openwakeword_amd import Model
    try:
        paths = TE.export_reference_files(str(tmp_path), cases.onnx_file_weights(), head_opsets=cases.ONNX_HEAD_OPSETS)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")

    import os
    import wave
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, clip, mapping, fthr = cases.ONNX_MAPPING
    wav = str(tmp_path / "clip.wav")
    with wave.open(wav, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes(golden["pcm/" + clip].tobytes())
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=[paths[n] for n in head_names], class_mapping_dicts=mapping,
              melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
    try:
        assert m.class_mapping["alexa_custom"] == {"alexa_custom": {"0": "positive"}}     # the reference stores the outer dict (model.py:176-177)
        preds = m.predict_clip(golden["pcm/" + clip], chunk_size=1280)
        labels = list(ref[f"{cid}/labels"])
        assert sorted(preds[0].keys()) == labels
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        np.testing.assert_allclose(got, ref[f"{cid}/scores"], rtol=0, atol=TOL_SCORE)
        assert [m.get_parent_model_from_label(k) for k in labels] == list(ref[f"{cid}/parents"])
        np.random.seed(cases.SEED_NP)
        m.reset()
        pos = m._get_positive_prediction_frames(wav, threshold=fthr, return_type="features")
        assert sorted(pos.keys()) == list(ref[f"{cid}/positive_labels"])
        for k, v in pos.items():
            want = ref[f"{cid}/positive/{k}"]
            assert v.shape == want.shape, k
            np.testing.assert_allclose(v, want, rtol=0, atol=2e-4)
        audio = m._get_positive_prediction_frames(wav, threshold=fthr, return_type="audio")
        assert all(a.shape[1] == 64000 for a in audio.values())
    finally:
        m.close()


@gpu
@pytest.mark.parametrize("case", cases.ONNX_SPEEX_CASES, ids=[c[0] for c in cases.ONNX_SPEEX_CASES])
def test_hip_model_with_speex_matches_the_reference_with_the_same_stand_in(tmp_path, golden, monkeypatch, case):
    """(f)4, the Speex hook on the HIP Model: files loaded by path, `speexdsp_ns` = oracle/fake_speex.py as in the reference's run
    (tests/golden/make_golden_onnx.py; the real package is not in this image).  Cleaned audio to the device front end (bit for bit what
    the reference's preprocessor buffered), raw audio to the voice-activity session (model.py:370), chunked by predict_clip."""
    pytest.importorskip("torch")
    import os
    import sys
    import torch_export as TE
    from oracle import fake_speex, mini_ort
    from openwakeword_amd import Model, model as M
    cid, head_names, clip, kw, thr = case
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    try:
        paths = TE.export_reference_files(str(tmp_path), cases.onnx_file_weights(), head_opsets=cases.ONNX_HEAD_OPSETS)
        vad_kw = {}
        if thr > 0:
            vpath = str(tmp_path / "silero_vad.onnx")
            TE.export_vad(W.synthetic_vad(cases.ONNX_VAD_SEED), vpath)
            vad_kw = dict(vad_threshold=thr, vad_session=mini_ort.InferenceSession(vpath))
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    monkeypatch.setitem(sys.modules, "speexdsp_ns", fake_speex.as_module())
    fake_speex.NoiseSuppression.instances.clear()
    fed = []
    real_call = M.AudioFeatures.__call__
    monkeypatch.setattr(M.AudioFeatures, "__call__", lambda self, x: (fed.append(np.array(x)), real_call(self, x))[1])
    np.random.seed(cases.SEED_NP)
    m = Model(wakeword_models=[paths[n] for n in head_names], melspec_model_path=paths["melspectrogram"],
              embedding_model_path=paths["embedding_model"], enable_speex_noise_suppression=True, **vad_kw)
    try:
        preds = m.predict_clip(golden["pcm/" + clip], **kw)
        labels = list(ref[f"{cid}/labels"])
        assert sorted(preds[0].keys()) == labels
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        np.testing.assert_allclose(got, ref[f"{cid}/scores"], rtol=0, atol=TOL_SCORE)
        feats = ref[f"{cid}/features"]
        n = min(len(feats), 120)
        np.testing.assert_allclose(m.preprocessor.get_features(n)[0], feats[-n:], rtol=0, atol=2e-4)
        np.testing.assert_array_equal(np.concatenate(fed)[-24000:-16000], ref[f"{cid}/raw_mid"])
        assert len(fake_speex.NoiseSuppression.instances) == 1
        if thr > 0:
            np.testing.assert_allclose(np.array(m.vad.prediction_buffer), ref[f"{cid}/vad"], rtol=0, atol=1e-6)
    finally:
        m.close()


@gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 13, 21, 34])
def test_random_call_sequences_match_the_oracle_model(seed):
    """tools/fuzz_model_vs_oracle.py: Model.predict on the HIP library against OracleModel over random call sequences (calls of
    0 ... 6000 samples, silence ... full scale, random head sets, patience / debounce, calls longer than max_chunks, a reset in
    mid-sequence); 150 seeds of it ran clean on the GPU in round 6 (profiles/r06_fuzz_model_vs_oracle.txt)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_model_vs_oracle", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_model_vs_oracle.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rec = fz.one_seed(seed)
    assert rec["worst"] <= fz.TOL and rec["scores"] > 0
    if seed == 1:          # one LONG sequence: 320 calls, every device ring (features 120, mel 970 rows, scores 30, VAD 125) wraps
        rec = fz.one_seed(2003, 320, [1280, 1280, 1280, 1280, 640, 2560, 1000, 0])
        assert rec["worst"] <= fz.TOL and rec["scores"] > 1000
