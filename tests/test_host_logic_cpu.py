"""The host shim of the drop-in surface (openwakeword_amd/model.py: Model, AudioFeatures) in the CPU tier: its engine calls are
served by tests/stub_engine.OracleEngine (the CPU oracle behind the engine interface), everything above them -- buffering in 1280-
sample units with carry-over, calls shorter / longer than a chunk, the first-five rule, patience and debounce, class mapping, parent
lookup, reset, argument errors -- is the product's own code, held to the vectors the REFERENCE's code produced
(tests/golden/make_golden.py).  The same cases run on the HIP engine in tests/test_model_api.py (GPU tier)."""
import numpy as np
import pytest

import cases
from openwakeword_amd import model as M, weights as W
from stub_engine import OracleEngine


@pytest.fixture()
def stub(monkeypatch):
    made = []

    def make_engine(n_streams, heads, embedding, use_mfma=None, **kw):
        made.append(OracleEngine(n_streams, heads, embedding, **kw))
        return made[-1]

    monkeypatch.setattr(M, "make_engine", make_engine)
    return made


def _weights(names):
    return {"embedding": W.synthetic_embedding(cases.SEED_WEIGHTS), "heads": {n: W.synthetic_head(n, cases.SEED_WEIGHTS) for n in names}}


@pytest.mark.parametrize("case", cases.CLIP_CASES, ids=[c[0] for c in cases.CLIP_CASES])
def test_host_shim_on_the_oracle_engine_matches_reference_golden(stub, golden, case):
    cid, head_names, clip, kw = case
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=list(head_names), weights=_weights(head_names))
    assert isinstance(m._engine, OracleEngine)
    if cid == "c1280":
        np.testing.assert_allclose(m.preprocessor.get_features(41)[0], golden["init/feature_buffer"], rtol=0, atol=2e-5)
    preds = m.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(golden[cid + "/labels"])
    assert sorted(preds[0].keys()) == labels
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    want = golden[cid + "/scores"]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    feats = golden[cid + "/features"]
    n = min(len(feats), 120)
    np.testing.assert_allclose(m.preprocessor.get_features(n)[0], feats[-n:], rtol=0, atol=2e-5)
    if cid == "c1280":
        np.random.seed(cases.SEED_NP + 1)
        m.reset()
        preds2 = m.predict_clip(golden["pcm/hey_mycroft_test"], chunk_size=1280)
        got2 = np.array([[float(p[k]) for k in labels] for p in preds2])
        np.testing.assert_allclose(got2, golden["reset/scores"], rtol=0, atol=2e-6)
    m.close()
    assert stub[-1].closed


def test_predict_argument_errors_and_short_calls(stub, golden):
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=["alexa"], weights=_weights(["alexa"]))
    with pytest.raises(ValueError):                               # model.py:262-263
        m.predict([0] * 1280)
    with pytest.raises(ValueError):                               # utils.py:195-197
        m.predict(np.zeros(1280, np.float32))
    with pytest.raises(ValueError):                               # model.py:341-343
        m.predict(np.zeros(1280, np.int16), patience={"alexa": 3})
    with pytest.raises(ValueError):                               # model.py:344-345
        m.predict(np.zeros(1280, np.int16), patience={"alexa": 3}, threshold={"alexa": 0.5}, debounce_time=1.0)
    # calls shorter than a chunk repeat the previous score and lose no sample (model.py:299-307, utils.py:413-430); ragged call
    # sizes around the chunk size; the first-five rule counts CALLS (model.py:331-333) -- call for call against the oracle's
    # restatement of the reference (itself golden-pinned for chunk sizes 400 ... 5120)
    from oracle import oww_oracle as O
    clip = golden["pcm/hey_jane"]
    w = _weights(["alexa"])
    for sizes in ([640] * 24, [400, 1280, 3000, 17, 1263, 2560, 5000, 1], [1279, 1, 1281, 2559]):
        np.random.seed(cases.SEED_NP)
        hip_side = M.Model(wakeword_models=["alexa"], weights=w)
        np.random.seed(cases.SEED_NP)
        ref = O.OracleModel(w["heads"], w["embedding"])
        o = 0
        for n in sizes:
            got, want = hip_side.predict(clip[o:o + n]), ref.predict(clip[o:o + n])
            assert set(got) == set(want) == {"alexa"}
            assert abs(got["alexa"] - want["alexa"]) < 2e-6, (sizes, o, n)
            o += n
    assert m.get_parent_model_from_label("alexa") == "alexa"
    _, clock = m.predict(np.zeros(1280, np.int16), timing=True)
    assert set(clock["models"]) == {"preprocessor", "alexa"}


def test_call_longer_than_max_chunks_keeps_the_calls_clamp_floor(stub, golden):
    """cases.ONNX_LONG: 5-chunk predict() calls through a Model whose engine holds ONE chunk of mel rows, against the reference's own
    run on the exporter-written files (one run of the melspectrogram graph per call = one clamp floor, utils.py:387-401): the whole
    call goes to the engine in one step (oww_step slices it behind a maximum pass; here the oracle engine), the stream carries on."""
    import os
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, sizes = cases.ONNX_LONG
    w = cases.onnx_file_weights()
    x = cases.long_call_pcm(golden["pcm/alexa_test"])
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=list(head_names), weights={"embedding": w["embedding"], "heads": {n: w["heads"][n] for n in head_names}},
                max_chunks=1)
    calls = []
    real = stub[-1].step_raw
    stub[-1].step_raw = lambda pcm: (calls.append(pcm.shape[1] // 1280), real(pcm))[1]
    labels = list(ref[f"{cid}/labels"])
    rows, o = [], 0
    for n in sizes:
        p = m.predict(x[o:o + n])
        o += n
        rows.append([float(p[k]) for k in labels])
    assert calls == [n // 1280 for n in sizes] and m.preprocessor.accumulated_samples == 0
    np.testing.assert_allclose(np.array(rows), ref[f"{cid}/scores"], rtol=0, atol=2e-5)
    feats = ref[f"{cid}/features"]
    np.testing.assert_allclose(m.preprocessor.get_features(len(feats))[0], feats, rtol=0, atol=2e-5)


@pytest.mark.parametrize("case", cases.VAD_CASES, ids=[c[0] for c in cases.VAD_CASES])
def test_vad_gate_of_the_host_shim_matches_reference_golden(stub, golden, case):
    """model.py:366-381 + vad.py as the host shim carries them (openwakeword_amd/vad.py, Model.predict): gated scores, the VAD ring,
    the ungated score ring, reset() leaving the VAD alone -- against the reference's own Model (tests/golden/make_golden_vad.py)."""
    import os
    from oracle.pseudo_vad import PseudoVadSession
    gv = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vad.npz")))
    cid, head_names, clip, kw, thr = case
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=list(head_names), weights=_weights(head_names), vad_threshold=thr, vad_session=PseudoVadSession())
    preds = m.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(gv[f"{cid}/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    np.testing.assert_allclose(got, gv[f"{cid}/scores"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.array(m.vad.prediction_buffer), gv[f"{cid}/vad"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.array([list(m.prediction_buffer[k]) for k in labels]), gv[f"{cid}/ring"], rtol=0, atol=2e-6)
    if cid == "vad03":
        np.random.seed(cases.SEED_NP + 1)
        m.reset()
        preds = m.predict_clip(golden["pcm/hey_mycroft_test"], chunk_size=1280)
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        np.testing.assert_allclose(got, gv["vadreset/scores"], rtol=0, atol=2e-6)


def test_custom_verifier_and_label_mapping_on_the_host(stub, golden, tmp_path):
    """model.py:320-328 (verifier re-scores frames the base model likes), 177-182 (class mapping), 215-224 (parent lookup)."""
    import pickle

    class Verifier:
        def predict_proba(self, feats):
            assert feats.shape == (1, 16, 96)
            return np.array([[0.75, 0.25]])

    path = tmp_path / "v.pkl"
    import test_host_logic_cpu as me          # (pickle needs an importable class)
    me.Verifier = Verifier
    Verifier.__module__, Verifier.__qualname__ = "test_host_logic_cpu", "Verifier"
    pickle.dump(Verifier(), open(path, "wb"))
    w = _weights(["alexa", "timer"])
    np.random.seed(cases.SEED_NP)
    plain = M.Model(wakeword_models=["alexa", "timer"], weights=w)
    np.random.seed(cases.SEED_NP)
    ver = M.Model(wakeword_models=["alexa", "timer"], weights=w, custom_verifier_models={"alexa": str(path)}, custom_verifier_threshold=0.3)
    clip = golden["pcm/hey_jane"]
    hit = False
    for o in range(0, len(clip) - 1280, 1280):
        a, b = plain.predict(clip[o:o + 1280]), ver.predict(clip[o:o + 1280])
        assert set(a) == set(b) and {"alexa", "1_minute_timer", "1_hour_timer"} <= set(a)          # timers through the class mapping
        if a["alexa"] >= 0.3:
            assert b["alexa"] == 0.25
            hit = True
        else:
            assert b["alexa"] == a["alexa"]
    assert hit
    assert ver.get_parent_model_from_label("1_hour_timer") == "timer" and ver.get_parent_model_from_label("alexa") == "alexa"
    with pytest.raises(ValueError, match="not matched"):
        M.Model(wakeword_models=["alexa"], weights=w, custom_verifier_models={"nope": str(path)})


def test_host_shim_matches_the_reference_on_ragged_and_empty_calls(stub, golden):
    """cases.ONNX_SEQUENCE (calls of 0 ... 5000 samples) against what the reference's own code returned on the exporter-written
    files (tests/golden/make_golden_onnx.py); here the host shim runs on the source weights over the oracle engine."""
    import os
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, clip, sizes = cases.ONNX_SEQUENCE
    w = cases.onnx_file_weights()
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=list(head_names), weights={"embedding": w["embedding"], "heads": {n: w["heads"][n] for n in head_names}})
    labels = list(ref[f"{cid}/labels"])
    rows, o = [], 0
    for n in sizes:
        p = m.predict(golden["pcm/" + clip][o:o + n])
        o += n
        assert sorted(p.keys()) == labels
        rows.append([float(p[k]) for k in labels])
    np.testing.assert_allclose(np.array(rows), ref[f"{cid}/scores"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("case", cases.ONNX_VAD_CASES, ids=[c[0] for c in cases.ONNX_VAD_CASES])
def test_host_vad_wrapper_on_an_exported_file_session(stub, golden, tmp_path, case):
    """The host VAD wrapper (openwakeword_amd/vad.py) driving a session that evaluates the exporter-written voice-activity FILE
    (oracle/mini_ort.py), against the reference's own VAD class on the same file (tests/golden/make_golden_onnx.py)."""
    pytest.importorskip("torch")
    import os
    import torch_export as TE
    from oracle import mini_ort
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, clip, kw, thr = case
    path = str(tmp_path / "silero_vad.onnx")
    try:
        TE.export_vad(W.synthetic_vad(cases.ONNX_VAD_SEED), path)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    w = cases.onnx_file_weights()
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=list(head_names), weights={"embedding": w["embedding"], "heads": {n: w["heads"][n] for n in head_names}},
                vad_threshold=thr, vad_session=mini_ort.InferenceSession(path))
    preds = m.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(ref[f"{cid}/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    np.testing.assert_allclose(np.array(m.vad.prediction_buffer), ref[f"{cid}/vad"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got, ref[f"{cid}/scores"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.array([list(m.prediction_buffer[k]) for k in labels]), ref[f"{cid}/ring"], rtol=0, atol=2e-5)


def test_custom_verifier_pickle_matches_the_reference_on_exported_files(stub, golden, tmp_path):
    """model.py:183-195 (loading) + 320-328 (re-scoring) with the pickled scikit-learn pipeline of tests/verifier_fixture.py, against
    the reference's own Model run with the same pickle on the exporter-written files (cases.ONNX_VERIFIER)."""
    pytest.importorskip("sklearn")
    import os
    import verifier_fixture
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, clip, kw, target, vthr = cases.ONNX_VERIFIER
    pkl = verifier_fixture.write(str(tmp_path / "verifier.pkl"))
    w = cases.onnx_file_weights()
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=list(head_names), weights={"embedding": w["embedding"], "heads": {n: w["heads"][n] for n in head_names}},
                custom_verifier_models={target: pkl}, custom_verifier_threshold=vthr)
    preds = m.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(ref[f"{cid}/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    want = ref[f"{cid}/scores"]
    assert (np.abs(want[:, 0] - ref["f1280j/scores"][:, 0]) > 1e-6).sum() > 10          # the verifier did re-score frames
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def test_mapping_parent_lookup_and_positive_frames_match_the_reference(stub, golden, tmp_path):
    """class_mapping_dicts, get_parent_model_from_label and _get_positive_prediction_frames (model.py:176-182, 215-224, 428-479)
    against the reference's own run on exported files (cases.ONNX_MAPPING); here the host shim on the oracle engine."""

    import os
    import wave
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, clip, mapping, fthr = cases.ONNX_MAPPING
    wav = str(tmp_path / "clip.wav")
    with wave.open(wav, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes(golden["pcm/" + clip].tobytes())
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=list(head_names), class_mapping_dicts=mapping,
                weights={"embedding": cases.onnx_file_weights()["embedding"], "heads": {n: cases.onnx_file_weights()["heads"][n] for n in head_names}})
    try:
        assert m.class_mapping["alexa_custom"] == {"alexa_custom": {"0": "positive"}}     # the reference stores the outer dict (model.py:176-177)
        preds = m.predict_clip(golden["pcm/" + clip], chunk_size=1280)
        labels = list(ref[f"{cid}/labels"])
        assert sorted(preds[0].keys()) == labels
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        np.testing.assert_allclose(got, ref[f"{cid}/scores"], rtol=0, atol=2e-5)
        assert [m.get_parent_model_from_label(k) for k in labels] == list(ref[f"{cid}/parents"])
        np.random.seed(cases.SEED_NP)
        m.reset()
        pos = m._get_positive_prediction_frames(wav, threshold=fthr, return_type="features")
        assert sorted(pos.keys()) == list(ref[f"{cid}/positive_labels"])
        for k, v in pos.items():
            want = ref[f"{cid}/positive/{k}"]
            assert v.shape == want.shape, k
            np.testing.assert_allclose(v, want, rtol=0, atol=1e-4)
        audio = m._get_positive_prediction_frames(wav, threshold=fthr, return_type="audio")
        assert all(a.shape[1] == 64000 for a in audio.values())
    finally:
        m.close()


@pytest.mark.parametrize("case", cases.ONNX_SPEEX_CASES, ids=[c[0] for c in cases.ONNX_SPEEX_CASES])
def test_speex_hook_of_the_host_shim_matches_the_reference_with_the_same_stand_in(stub, golden, tmp_path, monkeypatch, case):
    """Model(enable_speex_noise_suppression=True) (model.py:201-205, 272-273, 481-504; the reference's own test: tests/test_models.py:
    179-215).  `speexdsp_ns` is oracle/fake_speex.py on both sides (the package is not in this image): the reference's own Model ran
    with it on the exporter-written files (tests/golden/make_golden_onnx.py).  Pinned: ONE filter object per Model fed 160-sample
    frames in call order, the CLEANED audio is what the preprocessor buffers (bit for bit), the RAW call argument is what the
    voice-activity detector scores (model.py:370), predict_clip's chunking on top (chunk_size 1280 and 2560)."""
    pytest.importorskip("torch")
    import os
    import sys
    import torch_export as TE
    from oracle import fake_speex, mini_ort
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    cid, head_names, clip, kw, thr = case
    monkeypatch.setitem(sys.modules, "speexdsp_ns", fake_speex.as_module())
    fake_speex.NoiseSuppression.instances.clear()
    vad_kw = {}
    if thr > 0:
        path = str(tmp_path / "silero_vad.onnx")
        try:
            TE.export_vad(W.synthetic_vad(cases.ONNX_VAD_SEED), path)
        except Exception as e:                                  # noqa: BLE001
            pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
        vad_kw = dict(vad_threshold=thr, vad_session=mini_ort.InferenceSession(path))
    fed = []
    real_call = M.AudioFeatures.__call__
    monkeypatch.setattr(M.AudioFeatures, "__call__", lambda self, x: (fed.append(np.array(x)), real_call(self, x))[1])
    w = cases.onnx_file_weights()
    np.random.seed(cases.SEED_NP)
    m = M.Model(wakeword_models=list(head_names), weights={"embedding": w["embedding"], "heads": {n: w["heads"][n] for n in head_names}},
                enable_speex_noise_suppression=True, **vad_kw)
    assert type(m.speex_ns).__module__ == "oracle.fake_speex"
    preds = m.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(ref[f"{cid}/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    np.testing.assert_allclose(got, ref[f"{cid}/scores"], rtol=0, atol=2e-5)
    feats = ref[f"{cid}/features"]
    np.testing.assert_allclose(m.preprocessor.get_features(len(feats))[0], feats, rtol=0, atol=2e-5)
    np.testing.assert_array_equal(np.concatenate(fed)[-24000:-16000], ref[f"{cid}/raw_mid"])      # the preprocessor saw the cleaned audio
    assert len(fake_speex.NoiseSuppression.instances) == 1 and fake_speex.NoiseSuppression.instances[0].n_frames * 160 == sum(len(f) for f in fed)
    if thr > 0:
        np.testing.assert_allclose(np.array(m.vad.prediction_buffer), ref[f"{cid}/vad"], rtol=0, atol=1e-6)   # ... and the VAD the raw audio
        np.testing.assert_allclose(ref[f"{cid}/vad"], ref["fvad20/vad"], rtol=0, atol=0)        # (the reference's VAD ring with and without Speex: same clip)
    assert np.abs(ref[f"{cid}/scores"] - (ref["f1280/scores"] if cid == "fspeex" else 0)).max() > 0.05    # the filter matters
    m.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_call_sequences_through_the_host_shim_match_the_oracle_model(stub, seed):
    """tools/fuzz_model_vs_oracle.py with the host shim served by the oracle engine: the shim's alignment / carry-over, multi-chunk
    maxima, first-five zeroing, patience / debounce and reset must reproduce OracleModel (model.py:232-386 restated) EXACTLY."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_model_vs_oracle", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_model_vs_oracle.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rec = fz.one_seed(seed)
    assert rec["worst"] == 0.0 and rec["borderline"] == 0 and rec["scores"] > 0
    if seed == 1:          # and one long sequence with the debounce rule (the reference's ZeroDivisionError on frame-less calls included)
        rec = fz.one_seed(5, 300, [1280, 1280, 1280, 1280, 640, 2560, 1000, 0])
        assert rec["worst"] == 0.0 and rec["raised"] > 0
