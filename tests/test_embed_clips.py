"""bulk clip embedding (`embed_clips`, `compute_features_from_generator`) on the device path.
Reference behaviour: /root/reference/openwakeword/utils.py:238-385, 542-601; data.py:856-892."""
import os

import numpy as np
import pytest

from openwakeword_amd import weights as W

gpu = pytest.mark.gpu


# ---------------------------------------------------------------- CPU: host-side arithmetic and file handling
def test_get_embedding_shape_arithmetic():
    from openwakeword_amd.model import AudioFeatures
    for n, want in ((16000, 3), (32000, 16), (12512, 1), (12511, 0), (400, 0), (24123, 10)):
        frames = (n - 512) // 160 + 1 if n >= 512 else 0
        assert AudioFeatures.get_embedding_shape(None, n / 16000) == (max((frames - 76) // 8 + 1, 0), 96)
        assert AudioFeatures.get_embedding_shape(None, n / 16000)[0] == want


def test_trim_mmap(tmp_path):
    from numpy.lib.format import open_memmap
    from openwakeword_amd.utils import trim_mmap
    path = str(tmp_path / "feats.npy")
    fp = open_memmap(path, mode="w+", dtype=np.float32, shape=(3000, 3, 96))
    fp[:2500] = np.random.default_rng(0).standard_normal((2500, 3, 96)).astype(np.float32)
    keep = np.array(fp[:2500])
    fp.flush()
    del fp
    assert trim_mmap(path) == 2500                       # data.py:867-873: rows after the last non-zero one go
    got = np.load(path)
    assert got.shape == (2500, 3, 96) and np.array_equal(got, keep)
    assert trim_mmap(path) == 2500                       # idempotent
    assert trim_mmap(path, 10) == 10 and np.array_equal(np.load(path), keep[:10])


# ---------------------------------------------------------------- GPU
def _oracle_clips(emb, x):
    from oracle import oww_oracle as O
    oracle = O.OracleAudioFeatures(emb, init_noise=np.zeros(64000, np.int16))
    return np.stack([np.asarray(oracle.clip_embeddings(c)).reshape(-1, 96) for c in x])


@gpu
@pytest.mark.parametrize("B,n", [(5, 24123), (13, 16000), (1, 12512), (3, 40000)])
def test_device_embed_clips_matches_oracle_and_window_path(B, n):
    """PCM -> embeddings without leaving the device == oracle == the stage-by-stage path (oww_mel_clips + oww_embed);
    ragged batch sizes, lengths that are not a multiple of the hop, the one-window minimum."""
    from openwakeword_amd.engine import StreamEngine
    emb = W.synthetic_embedding(1234)
    x = W.synthetic_pcm(B, n, seed=100 + B)
    x[0, : n // 2] = 0                                    # a clip with digital silence: the per-clip clamp floor matters
    eng = StreamEngine(16, {"alexa": W.synthetic_head("alexa", 1234)}, emb)
    try:
        got = eng.embed_clips(x)
        want = _oracle_clips(emb, x)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
        spec = eng.mel_clips(x) / 10.0 + 2.0
        n_win = (spec.shape[1] - 76) // 8 + 1
        ref = eng.embed(np.ascontiguousarray(spec[:, : 76 + 8 * (n_win - 1)], dtype=np.float32))
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
        # deterministic and independent of what the borrowed streams held before
        np.testing.assert_array_equal(got, eng.embed_clips(x))
    finally:
        eng.close()


@gpu
def test_embed_calls_leave_live_streams_untouched():
    """Since ABI 2 oww_embed / oww_embed_clips park the state of the streams they borrow (conv histories, feature ring, frame
    counters of ALL streams) and put it back: a stream that is in the middle of an utterance continues bit-for-bit as if the
    bulk call had not happened (they used to clobber streams [0, B))."""
    from openwakeword_amd.engine import StreamEngine
    emb = W.synthetic_embedding(1234)
    heads = {"alexa": W.synthetic_head("alexa", 1234), "hey_jarvis": W.synthetic_head("hey_jarvis", 1234)}
    S, n_frames = 11, 9
    pcm = W.synthetic_pcm(S, 1280 * n_frames, seed=21)
    clips = W.synthetic_pcm(7, 16000, seed=22)
    feats0 = np.random.default_rng(3).normal(0, 1, (16, 96)).astype(np.float32)
    runs = []
    for interrupt in (False, True):
        eng = StreamEngine(S, heads, emb)
        try:
            eng.reset(None, feats0)
            out = []
            for t in range(n_frames):
                out.append(eng.step(pcm[:, 1280 * t:1280 * (t + 1)]).copy())
                if interrupt and t in (2, 5):
                    e = eng.embed_clips(clips)
                    assert e.shape == (7, 3, 96) and np.isfinite(e).all()
                if interrupt and t == 6:
                    spec = eng.mel_clips(clips[:2]) / 10.0 + 2.0
                    eng.embed(np.ascontiguousarray(spec[:, :76], dtype=np.float32))
            runs.append((np.stack(out), np.stack([eng.get_features(s, 8) for s in range(S)])))
        finally:
            eng.close()
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])


def test_batched_model_never_pairs_real_heads_with_a_synthetic_embedding():
    """BatchedModel / bulk_predict resolve weights exactly like Model -- without the .onnx files and
    without an explicit request for synthetic weights they refuse instead of scoring with a random-init embedding."""
    from openwakeword_amd import BatchedModel
    from openwakeword_amd.model import resolve_embedding, resolve_weights
    with pytest.raises(ValueError, match="does not exist"):
        BatchedModel(4, ["alexa"])
    assert resolve_weights("synthetic")[0] == 1234 and resolve_weights(None) == (None, None, {})
    with pytest.raises(ValueError, match="does not exist"):
        resolve_embedding(None, None)
    with pytest.raises(ValueError):
        resolve_weights("random")


@gpu
def test_device_embed_clips_errors():
    from openwakeword_amd.engine import StreamEngine
    eng = StreamEngine(4, {"alexa": W.synthetic_head("alexa", 1234)}, W.synthetic_embedding(1234))
    try:
        with pytest.raises(ValueError):
            eng.embed_clips(np.zeros((2, 12511), np.int16))            # 75 frames: no window (utils.py:313-314)
        with pytest.raises(ValueError):
            eng.embed_clips(np.zeros((2, 16000), np.float32))
        with pytest.raises(ValueError):
            eng.embed_clips(np.zeros((eng.n_streams_padded + 1, 16000), np.int16))
    finally:
        eng.close()


@gpu
def test_compute_features_from_generator(tmp_path):
    """utils.py:542-601: generator of [batch, samples] int16 -> (N, windows, 96) float32 .npy, cut at n_total, trimmed
    when the generator runs dry; a batch larger than n_total is an error."""
    from openwakeword_amd.engine import StreamEngine
    from openwakeword_amd.utils import compute_features_from_generator
    emb = W.synthetic_embedding(1234)
    clips = W.synthetic_pcm(22, 32000, seed=5)

    def gen(bs):
        for lo in range(0, clips.shape[0], bs):
            yield clips[lo:lo + bs]

    eng = StreamEngine(8, {}, emb)                          # a handle without heads is enough for feature extraction
    try:
        want = np.concatenate([eng.embed_clips(clips[lo:lo + 8]) for lo in range(0, 22, 8)])
        path = str(tmp_path / "f.npy")
        assert compute_features_from_generator(gen(8), n_total=20, clip_duration=32000, output_file=path, engine=eng) == 20
        got = np.load(path)
        assert got.shape == (20, 16, 96) and got.dtype == np.float32
        np.testing.assert_array_equal(got, want[:20])
        # n_total larger than what the generator yields: file trimmed to the 22 rows written
        assert compute_features_from_generator(gen(6), n_total=40, clip_duration=32000, output_file=path, engine=eng) == 22
        np.testing.assert_array_equal(np.load(path), want)
        with pytest.raises(ValueError):
            compute_features_from_generator(gen(8), n_total=4, clip_duration=32000, output_file=path, engine=eng)
    finally:
        eng.close()
    # engine created on demand (synthetic weights: the model file is not in the checkout)
    assert compute_features_from_generator(gen(11), n_total=22, clip_duration=32000, output_file=path, weights="synthetic") == 22
    np.testing.assert_allclose(np.load(path), want, rtol=0, atol=1e-6)
    assert _oracle_clips(emb, clips[:2]).shape == (2, 16, 96)
    np.testing.assert_allclose(np.load(path)[:2], _oracle_clips(emb, clips[:2]), rtol=0, atol=2e-4)


@gpu
def test_bulk_predict_matches_predict_clip_per_file(tmp_path):
    """utils.bulk_predict (utils.py:466-536) with every file as one stream of a BatchedModel == Model.predict_clip on each
    file with a fresh Model (clips of different lengths, 80 ms and 160 ms calls, debounce)."""
    import wave
    from openwakeword_amd import Model
    from openwakeword_amd.utils import bulk_predict
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_streaming.npz"))
    files = []
    for name in ("alexa_test", "hey_mycroft_test", "hey_jane"):
        path = str(tmp_path / (name + ".wav"))
        with wave.open(path, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes(z["pcm/" + name].tobytes())
        files.append(path)
    names = ["alexa", "hey_mycroft"]
    w = {"embedding": W.synthetic_embedding(1234), "heads": {n: W.synthetic_head(n, 1234) for n in names}}
    for kw in (dict(chunk_size=1280), dict(chunk_size=2560, padding=0), dict(chunk_size=1280, debounce_time=0.3, threshold={"alexa": 0.4, "hey_mycroft": 0.5})):
        np.random.seed(11)
        got = bulk_predict(files, names, weights=w, **kw)
        assert list(got) == files
        for f in files:
            np.random.seed(11)
            m = Model(wakeword_models=names, weights=w)
            try:
                want = m.predict_clip(f, **kw)
            finally:
                m.close()
            assert len(got[f]) == len(want)
            a = np.array([[p[k] for k in names] for p in got[f]])
            b = np.array([[p[k] for k in names] for p in want])
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    # anything else goes file by file through a Model
    np.random.seed(11)
    other = bulk_predict(files[:1], names, weights=w, chunk_size=1024)
    assert len(other[files[0]]) == len(range(0, len(z["pcm/alexa_test"]) + 32000 - 1024, 1024))


@gpu
def test_bulk_feature_path_matches_the_reference_audiofeatures_on_exported_files(tmp_path):
    """The (f)1 row against the REFERENCE's own AudioFeatures (utils.py:180-385) run on melspectrogram / embedding FILES written by
    PyTorch's exporter (tests/golden/make_golden_onnx.py): the HIP side loads the same files by path and must return the reference's
    embed_clips / _get_melspectrogram_batch / _get_embeddings / _get_melspectrogram / get_embedding_shape results, and
    compute_features_from_generator must write the same rows."""
    pytest.importorskip("torch")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    import torch_export as TE
    from openwakeword_amd import Model
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_onnx_files.npz"))
    try:
        paths = TE.export_reference_files(str(tmp_path), cases.onnx_file_weights(), head_opsets=cases.ONNX_HEAD_OPSETS)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    m = Model(wakeword_models=[paths["alexa_custom"]], melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
    try:
        F = m.preprocessor
        x = ref["embed/pcm"]
        np.testing.assert_allclose(F.embed_clips(x, batch_size=3, ncpu=2), ref["embed/embed_clips"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(F._get_melspectrogram_batch(x, batch_size=2), ref["embed/melspec_batch"], rtol=0, atol=5e-4)
        np.testing.assert_allclose(F._get_embeddings(x[1]), ref["embed/get_embeddings"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(F._get_melspectrogram(x[0][:12345]), ref["embed/melspectrogram"], rtol=0, atol=5e-4)
        assert tuple(F.get_embedding_shape(2.0)) == tuple(ref["embed/shape_2s"])
    finally:
        m.close()
    from openwakeword_amd.utils import compute_features_from_generator
    out = str(tmp_path / "features.npy")
    assert compute_features_from_generator(iter([x[:2], x[2:]]), 4, 32000, out, weights=paths["embedding_model"]) == 4
    np.testing.assert_allclose(np.load(out), ref["embed/embed_clips"], rtol=0, atol=2e-4)
