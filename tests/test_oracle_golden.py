"""The oracle's restatement of the reference's streaming logic vs vectors produced by the REFERENCE's
own code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import cases
from oracle import oww_oracle as O
from openwakeword_amd import weights as W

TIMER_MAP = {"timer": {"1": "1_minute_timer", "2": "5_minute_timer", "3": "10_minute_timer",
                       "4": "20_minute_timer", "5": "30_minute_timer", "6": "1_hour_timer"}}


def build(head_names):
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    heads = {n: W.synthetic_head(n, cases.SEED_WEIGHTS) for n in head_names}
    np.random.seed(cases.SEED_NP)
    return O.OracleModel(heads, emb, class_mapping=TIMER_MAP)


@pytest.mark.parametrize("case", cases.CLIP_CASES, ids=[c[0] for c in cases.CLIP_CASES])
def test_streaming_restatement_matches_reference(golden, case):
    cid, head_names, clip, kw = case
    mdl = build(head_names)
    preds = mdl.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(golden[f"{cid}/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    want = golden[f"{cid}/scores"]
    assert got.shape == want.shape
    # identical stage math + identical control flow -> identical numbers (BLAS batching aside)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    np.testing.assert_allclose(mdl.preprocessor.features, golden[f"{cid}/features"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(mdl.preprocessor.mel_rows[-16:], golden[f"{cid}/mel_tail"], rtol=0, atol=1e-5)


def test_reset_then_second_clip(golden):
    mdl = build(cases.HEADS_BINARY)
    assert mdl.preprocessor.features.shape == (41, 96)            # SURVEY §3.3 structural pin
    np.testing.assert_allclose(mdl.preprocessor.features, golden["init/feature_buffer"], atol=2e-5)
    mdl.predict_clip(golden["pcm/alexa_test"], chunk_size=1280)
    np.random.seed(cases.SEED_NP + 1)
    mdl.reset()
    preds = mdl.predict_clip(golden["pcm/hey_mycroft_test"], chunk_size=1280)
    labels = list(golden["c1280/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    np.testing.assert_allclose(got, golden["reset/scores"], atol=2e-6)


def test_structural_pins(golden):
    # notebooks/converting_google_speech_embedding_model.ipynb:859 ; docs/models/alexa.md:26-29 ; timers.md:24-27
    assert O.cnn_param_count() == 332_088
    assert W.embedding_param_count(W.synthetic_embedding()) == 332_088
    assert W.head_param_count(W.synthetic_head("alexa")) == 102_849
    assert W.head_param_count(W.synthetic_head("timer")) == 435_335
    # SURVEY Appendix A self-consistency values of the mel recipe
    fb = O.mel_filterbank()
    assert (fb > 0).sum() == 229
    nz = np.nonzero(fb.sum(axis=1))[0]
    assert (nz[0], nz[-1]) == (2, 121)
    assert abs(fb.sum() - 1.023993) < 1e-5 and abs(fb.max() - 0.014234) < 1e-6
    assert abs((O.hann_window_padded() ** 2).sum() - 150.0) < 1e-9
    np.testing.assert_array_equal(fb, W.mel_filterbank())           # product table == oracle table
    # 5 rows on the first call, 8 afterwards
    assert O.n_mel_frames(1280) == 5 and O.n_mel_frames(1760) == 8 and O.n_mel_frames(1280 * 2 + 480) == 16


def test_stage_regression_vectors(golden):
    x = golden["stage/mel_in"].astype(np.float32)[None]
    np.testing.assert_allclose(O.mel_stage(x)[0, 0], golden["stage/mel_out"], atol=1e-4)
    # all-zero audio -> -100 dB everywhere -> mel value -8 after x/10+2 (Appendix A step 6)
    z = O.mel_transform(O.mel_stage(np.zeros((1, 1760), np.float32)))
    assert np.allclose(z, -8.0, atol=1e-5)
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    win = (golden["stage/mel_out"][:76] / 10 + 2).astype(np.float32)
    e = O.embedding_stage(win[None, :, :, None], emb).reshape(96)
    np.testing.assert_allclose(e, golden["stage/embed_out"], atol=2e-5)


def test_cnn_is_time_fully_convolutional():
    """SURVEY §8a-E probe: a (76+8K)-row strip gives the K+1 per-window outputs -> the incremental
    (streaming) evaluation used on the GPU is exact."""
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    r = np.random.default_rng(0)
    strip = r.normal(10, 1.5, (1, 76 + 8 * 3, 32, 1))
    full = O.embedding_stage(strip, emb, np.float64)[0, :, 0, :]
    assert full.shape == (4, 96)
    for k in range(4):
        one = O.embedding_stage(strip[:, 8 * k: 8 * k + 76], emb, np.float64).reshape(96)
        np.testing.assert_allclose(full[k], one, atol=1e-12)


# ---------------------------------------------------------------------------------------- row I: the VAD gate
@pytest.fixture(scope="module")
def golden_vad():
    import os
    return dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_vad.npz")))


@pytest.mark.parametrize("case", cases.VAD_CASES, ids=[c[0] for c in cases.VAD_CASES])
def test_vad_gate_restatement_matches_reference(golden, golden_vad, case):
    """The reference's own VAD class + gate (vad.py:54-130, model.py:366-381) ran on the pseudo network of
    oracle/pseudo_vad.py (tests/golden/make_golden_vad.py); the oracle's restatement must give the same gated scores, the same
    VAD ring and the same (ungated) score ring."""
    from oracle.pseudo_vad import PseudoVadSession
    cid, head_names, clip, kw, thr = case
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    heads = {n: W.synthetic_head(n, cases.SEED_WEIGHTS) for n in head_names}
    np.random.seed(cases.SEED_NP)
    mdl = O.OracleModel(heads, emb, vad_threshold=thr, vad_session=PseudoVadSession())
    preds = mdl.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(golden_vad[f"{cid}/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    np.testing.assert_allclose(got, golden_vad[f"{cid}/scores"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.array(mdl.vad.ring), golden_vad[f"{cid}/vad"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.array([list(mdl.prediction_buffer[k]) for k in labels]), golden_vad[f"{cid}/ring"], rtol=0, atol=2e-6)
    if cid == "vad03":
        np.random.seed(cases.SEED_NP + 1)
        mdl.reset()                                                   # leaves the VAD state and ring alone
        preds = mdl.predict_clip(golden["pcm/hey_mycroft_test"], chunk_size=1280)
        got = np.array([[float(p[k]) for k in labels] for p in preds])
        np.testing.assert_allclose(got, golden_vad["vadreset/scores"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(np.array(mdl.vad.ring), golden_vad["vadreset/vad"], rtol=0, atol=1e-6)


def test_vad_wrapper_of_the_package_matches_reference(golden, golden_vad):
    """openwakeword_amd.VAD (sub-framing, /32767 scaling, state carry, mean, 125-deep ring) on the calls predict_clip makes:
    depends on the audio only, so it is pinned on the CPU."""
    from oracle.pseudo_vad import PseudoVadSession
    from openwakeword_amd.vad import VAD, gate_value
    for cid, _heads, clip, kw, thr in cases.VAD_CASES:
        data = golden["pcm/" + clip]
        if kw.get("padding", 1):
            z = np.zeros(16000, np.int16)
            data = np.concatenate((z, data, z))
        v = VAD(session=PseudoVadSession())
        cs = kw["chunk_size"]
        gates = []
        for o in range(0, data.shape[0] - cs, cs):
            v(data[o:o + cs])
            gates.append(gate_value(v.prediction_buffer, thr))
        np.testing.assert_allclose(np.array(v.prediction_buffer), golden_vad[f"{cid}/vad"], rtol=0, atol=1e-6)
        # a gated frame is all zeros in the reference's output
        sc = golden_vad[f"{cid}/scores"]
        assert all((sc[i] == 0).all() for i, g in enumerate(gates) if g)
    with pytest.raises(ValueError, match="vad_session"):
        VAD()                                                          # no network, no onnxruntime: refused, not ignored


def test_torch_cpu_port_matches_numpy_oracle(golden):
    """oracle/oww_oracle_torch.py (the arithmetic bench.py times as `cpu_baseline`) against the numpy oracle on real audio: mel
    rows of a 1,760-sample streaming buffer, the embedding of a full 76-row window and all three head kinds."""
    torch = pytest.importorskip("torch")
    from oracle import oww_oracle_torch as OT
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    heads = {n: W.synthetic_head(n, cases.SEED_WEIGHTS) for n in ("alexa", "hey_jarvis", "timer")}
    port = OT.TorchCpuPort(emb, heads, threads=1)
    clip = golden["pcm/alexa_test"]
    pcm = clip[len(clip) // 4: len(clip) // 4 + 3 * 1760].reshape(3, 1760).astype(np.int16)
    rows = port.mel(pcm).numpy()
    want = np.stack([O.mel_transform(O.mel_stage(p.astype(np.float32))[0, 0]) for p in pcm])
    np.testing.assert_allclose(rows, want, rtol=0, atol=2e-4)
    win = np.random.default_rng(3).normal(9.0, 1.5, (2, 76, 32)).astype(np.float32)
    e = port.embed(torch.from_numpy(win)).numpy()
    np.testing.assert_allclose(e, O.embedding_stage(win[..., None], emb).reshape(2, 96), rtol=0, atol=2e-4)
    feats = np.random.default_rng(4).normal(0.0, 2.0, (2, 34, 96)).astype(np.float32)
    for name, h in heads.items():
        f = feats[:, -h["T"]:]
        np.testing.assert_allclose(port.head(name, torch.from_numpy(f)).numpy(), O.head_stage(f, h), rtol=0, atol=2e-5)


def test_mel_stage_matches_an_independent_stft(golden):
    """The oracle's log-mel (two dense DFT matmuls) against an independent formulation of the same recipe:
    torch.stft (n_fft 512, hop 160, periodic Hann(400) centred in the frame, center=False) -> |X|^2 -> the Slaney filterbank built
    here from the closed-form triangle definition -> 10 log10 -> top_db clamp over the call.  Pins framing, window placement and
    the bin <-> frequency mapping of the restatement to a library STFT (torchlibrosa's Spectrogram is a Conv1d form of the same)."""
    torch = pytest.importorskip("torch")
    clip = golden["pcm/hey_mycroft_test"].astype(np.float32)
    x = clip[len(clip) // 3: len(clip) // 3 + 1280 * 3 + 480]
    win = torch.hann_window(400, periodic=True, dtype=torch.float64)
    X = torch.stft(torch.from_numpy(x.astype(np.float64)), n_fft=512, hop_length=160, win_length=400, window=win, center=False,
                   return_complex=True)                                     # [257, F]; win_length < n_fft: the window is centre-padded
    power = (X.real ** 2 + X.imag ** 2).T.numpy()                           # [F, 257]
    # Slaney mel scale, slaney norm, 32 bands 60..3800 Hz, written from the definition (not the oracle's code path)
    def hz2mel(f):
        return np.where(f < 1000.0, f / (200.0 / 3.0), 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27.0))
    def mel2hz(m):
        return np.where(m < 15.0, m * (200.0 / 3.0), 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0)))
    edges = mel2hz(np.linspace(hz2mel(np.array(60.0)), hz2mel(np.array(3800.0)), 34))
    freqs = np.arange(257) * (16000.0 / 512.0)
    fb = np.zeros((257, 32))
    for i in range(32):
        lo, ce, hi = edges[i], edges[i + 1], edges[i + 2]
        tri = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)))
        fb[:, i] = tri * (2.0 / (hi - lo))
    db = 10.0 * np.log10(np.maximum(power @ fb, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    got = O.mel_stage(x[None], np.float64)[0, 0]
    assert got.shape == db.shape == ((len(x) - 512) // 160 + 1, 32)
    np.testing.assert_allclose(got, db, rtol=0, atol=1e-6)
    np.testing.assert_allclose(O.mel_stage(x[None], np.float32)[0, 0], db, rtol=0, atol=2e-3)     # fp32 DFT of int16-scale audio


def test_mel_recipe_matches_a_third_party_port_of_librosa(golden):
    """The [3P] part of the mel front end -- librosa's Slaney mel scale with slaney area normalisation, its periodic Hann window,
    power_to_db with amin 1e-10 / ref 1 / top_db 80 (what torchlibrosa's Spectrogram + LogmelFilterBank compute in the notebook's
    cell 15) -- against code that is neither the builder's nor the reference's: Hugging Face `transformers.audio_utils`
    (`mel_filter_bank(..., norm="slaney", mel_scale="slaney")`, `window_function`, `spectrogram(..., log_mel="dB", db_range=80)`),
    that library's own port of librosa's recipe.  Filter bank and window agree to float64 round-off (229 non-zero taps, FFT bins
    2 ... 121), the whole log-mel of real audio to 1e-6 dB."""
    au = pytest.importorskip("transformers.audio_utils")
    fb = au.mel_filter_bank(257, 32, 60.0, 3800.0, 16000, norm="slaney", mel_scale="slaney")
    np.testing.assert_allclose(O.mel_filterbank(np.float64), fb, rtol=0, atol=1e-15)
    assert int((fb != 0).sum()) == 229 and list(np.nonzero(fb.sum(1))[0][[0, -1]]) == [2, 121]
    win = np.zeros(512)
    win[56:456] = au.window_function(400, "hann", periodic=True)
    np.testing.assert_allclose(O.hann_window_padded(np.float64), win, rtol=0, atol=1e-15)
    for name in ("alexa_test", "hey_mycroft_test", "hey_jane"):
        clip = golden["pcm/" + name].astype(np.float64)
        x = clip[len(clip) // 3: len(clip) // 3 + 1280 * 3 + 480]
        want = au.spectrogram(x, win, frame_length=512, hop_length=160, fft_length=512, power=2.0, center=False, mel_filters=fb,
                              mel_floor=1e-10, log_mel="dB", reference=1.0, min_value=1e-10, db_range=80.0, dtype=np.float64).T
        got = O.mel_stage(x[None], np.float64)[0, 0]
        assert got.shape == want.shape == ((len(x) - 512) // 160 + 1, 32)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    # digital silence: every band at amin -> -100 dB, i.e. mel = -8 after the host transform
    z = au.spectrogram(np.zeros(1760), win, frame_length=512, hop_length=160, fft_length=512, power=2.0, center=False, mel_filters=fb,
                       mel_floor=1e-10, log_mel="dB", reference=1.0, min_value=1e-10, db_range=80.0, dtype=np.float64).T
    np.testing.assert_allclose(O.mel_stage(np.zeros((1, 1760)), np.float64)[0, 0], z, rtol=0, atol=0)
    assert float(O.mel_transform(z).max()) == -8.0


def test_cnn_and_head_stages_match_torch_nn_in_float64():
    """The oracle's embedding CNN and head restatements against an independent implementation of the same published architecture:
    torch.nn.Conv2d / BatchNorm2d(eps 1e-3) / leaky_relu(0.2) + clamp(-0.4) / MaxPool2d in NCHW, and nn.Linear / LayerNorm / ReLU /
    Sigmoid (| ReLU + softmax), both in float64 so that only the DEFINITIONS can differ.  The residual is the oracle's fp32 slope
    and floor constants (0.2, -0.4 as float32: 1.5e-8 relative).  Pins conv orientation and padding, BatchNorm folding, pooling
    windows, the LayerNorm epsilon and the Gemm orientation of the restatement to library kernels."""
    torch = pytest.importorskip("torch")
    import torch_export as TE
    for seed in (56, 1234):
        emb = W.synthetic_embedding(seed)
        net = TE.torch_embedding(emb).double()
        x = np.random.default_rng(seed).normal(10, 1.5, (3, 76 + 16, 32, 1))          # taller than one window: 3 outputs per item
        with torch.no_grad():
            ref = net(torch.from_numpy(x)).numpy()
        got = O.embedding_stage(x, emb, np.float64)
        assert got.shape == ref.shape == (3, 3, 1, 96)
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-7)
    for name, ln in (("alexa", None), ("timer", None), ("weather", True)):
        head = W.synthetic_head(name, 7, layernorm=ln)
        mod = TE.torch_head(head["net"], head["T"], head["n_out"]).double()
        f = np.random.default_rng(5).normal(0, 2, (4, head["T"], 96))
        with torch.no_grad():
            ref = mod(torch.from_numpy(f)).numpy()
        got = O.head_stage(f, head, np.float64)
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=0, atol=1e-12)


@pytest.fixture(scope="module")
def golden_files():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))


@pytest.mark.parametrize("case", cases.ONNX_FILE_CASES, ids=[c[0] for c in cases.ONNX_FILE_CASES])
def test_oracle_matches_the_reference_run_on_exported_model_files(golden, golden_files, case):
    """tests/golden/make_golden_onnx.py: the reference's own Model.predict_clip on heads / embedding / melspectrogram FILES written by
    PyTorch's exporter, evaluated by a generic ONNX interpreter -- no line of this repository's restatement in that loop.  The oracle
    on the SOURCE weights must give the same scores (the exporter folded BatchNorm into the convolutions in fp32: round-off)."""
    cid, head_names, clip, kw = case
    w = cases.onnx_file_weights()
    np.random.seed(cases.SEED_NP)
    mdl = O.OracleModel({n: w["heads"][n] for n in head_names}, w["embedding"])
    preds = mdl.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(golden_files[f"{cid}/labels"])
    assert sorted(preds[0].keys()) == labels
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    want = golden_files[f"{cid}/scores"]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    np.testing.assert_allclose(mdl.preprocessor.features, golden_files[f"{cid}/features"], rtol=0, atol=1e-4)
    if cid == "f1280":                                      # what the reference read from the files' input / output declarations
        assert list(golden_files["init/model_inputs"]) == [16, 16] and list(golden_files["init/model_outputs"]) == [1, 1]


def test_oracle_matches_the_reference_on_ragged_and_empty_calls(golden, golden_files):
    """Direct predict() calls of 0, 1, 17 ... 5000 samples (cases.ONNX_SEQUENCE) through the reference on exported files: calls that
    complete no chunk repeat the previous binary score and return zeros for a multiclass model (model.py:299-307)."""
    cid, head_names, clip, sizes = cases.ONNX_SEQUENCE
    w = cases.onnx_file_weights()
    np.random.seed(cases.SEED_NP)
    mdl = O.OracleModel({n: w["heads"][n] for n in head_names}, w["embedding"])
    labels = list(golden_files[f"{cid}/labels"])
    rows, o = [], 0
    for n in sizes:
        p = mdl.predict(golden["pcm/" + clip][o:o + n])
        o += n
        assert sorted(p.keys()) == labels
        rows.append([float(p[k]) for k in labels])
    np.testing.assert_allclose(np.array(rows), golden_files[f"{cid}/scores"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("case", cases.ONNX_VAD_CASES, ids=[c[0] for c in cases.ONNX_VAD_CASES])
def test_oracle_vad_matches_the_reference_vad_class_on_an_exported_file(golden, golden_files, case):
    """tests/golden/make_golden_onnx.py: the reference's Model(vad_threshold=...) whose own VAD class (vad.py:54-130) opens a
    voice-activity FILE -- the stand-in network written by PyTorch's exporter, evaluated by the generic ONNX interpreter.  The oracle
    (stand-in restatement + OracleVad + gate) must return the same gated scores, VAD ring and ungated score ring."""
    from oracle.vad_standin import StandinVadSession
    cid, head_names, clip, kw, thr = case
    w = cases.onnx_file_weights()
    np.random.seed(cases.SEED_NP)
    mdl = O.OracleModel({n: w["heads"][n] for n in head_names}, w["embedding"], vad_threshold=thr,
                        vad_session=StandinVadSession(W.synthetic_vad(cases.ONNX_VAD_SEED)))
    preds = mdl.predict_clip(golden["pcm/" + clip], **kw)
    labels = list(golden_files[f"{cid}/labels"])
    got = np.array([[float(p[k]) for k in labels] for p in preds])
    want, vad = golden_files[f"{cid}/scores"], golden_files[f"{cid}/vad"]
    np.testing.assert_allclose(np.array(mdl.vad.ring), vad, rtol=0, atol=2e-5)
    assert got.shape == want.shape and (want == 0).any() and (want > 0).any()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.array([list(mdl.prediction_buffer[k]) for k in labels]), golden_files[f"{cid}/ring"], rtol=0, atol=2e-5)


def test_oracle_bulk_features_match_the_reference_audiofeatures_on_exported_files(golden_files):
    """utils.py:180-385 as the reference itself ran them on the exporter-written melspectrogram / embedding files: embed_clips,
    _get_melspectrogram_batch (a floor per clip on the CPU path), _get_embeddings, _get_melspectrogram, get_embedding_shape."""
    w = cases.onnx_file_weights()
    F = O.OracleAudioFeatures(w["embedding"], init_noise=np.zeros(64000, np.int16))
    x = golden_files["embed/pcm"]
    want = golden_files["embed/embed_clips"]
    got = np.stack([F.clip_embeddings(c) for c in x])
    assert got.shape == want.shape == (4, 16, 96) and tuple(golden_files["embed/shape_2s"]) == (16, 96)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    np.testing.assert_allclose(np.stack([F.melspectrogram(c) for c in x]), golden_files["embed/melspec_batch"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(F.clip_embeddings(x[1]), golden_files["embed/get_embeddings"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(F.melspectrogram(x[0][:12345]), golden_files["embed/melspectrogram"], rtol=0, atol=2e-4)
