#!/usr/bin/env python3
"""
Generate tests/golden/ref_onnx_files.npz: the REFERENCE's own Python (/root/reference/openwakeword: Model.predict_clip ->
AudioFeatures -> ort.InferenceSession(...).run) running on real model FILES, with nothing of this repository's restatement in the
loop.  The files are written by PyTorch's own ONNX exporter (tests/torch_export.py: the tool the reference exports its models with,
train.py:144-165) from the deterministic synthetic weights; `onnxruntime` is oracle/mini_ort.py, a generic numpy evaluator of the ONNX
operator semantics (held to the torch modules in tests/test_mini_ort.py) that knows nothing about these networks.

Run only in the build container (needs /root/reference):   python tests/golden/make_golden_onnx.py
Consumers regenerate the same files with the same exporter and compare: the oracle on the source weights (CPU,
tests/test_oracle_golden.py) and the HIP Model loading the files BY PATH (GPU, tests/test_model_api.py).

What these vectors add to ref_streaming.npz: there the three networks behind the reference's seams were oracle functions; here they
are graph files evaluated operator by operator -- the reference's file-loading, input / output discovery (model.py:153-159) and its
whole predict path on what a user would hand it.  What they still cannot pin: the weights and exporter idioms of the RELEASED files.
"""
import os
import sys
import tempfile
import wave

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(__file__))
REF = "/root/reference"

from oracle import mini_ort                           # noqa: E402
import cases                                          # noqa: E402
import torch_export as TE                             # noqa: E402


def main():
    sys.modules["onnxruntime"] = mini_ort.as_module()
    sys.path.insert(0, REF)
    import openwakeword                               # the real reference package
    assert os.path.realpath(openwakeword.__file__).startswith(REF)

    clips = {}
    for c in cases.CLIPS:
        with wave.open(os.path.join(REF, "tests", "data", c + ".wav"), "rb") as f:
            clips[c] = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16).copy()

    out = {}
    with tempfile.TemporaryDirectory() as d:
        paths = TE.export_reference_files(d, cases.onnx_file_weights(), head_opsets=cases.ONNX_HEAD_OPSETS)
        for cid, head_names, clip, kw in cases.ONNX_FILE_CASES:
            np.random.seed(cases.SEED_NP)
            mdl = openwakeword.Model(wakeword_models=[paths[n] for n in head_names], inference_framework="onnx",
                                     melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
            assert sorted(mdl.models) == sorted(head_names), mdl.models.keys()
            preds = mdl.predict_clip(clips[clip], **kw)
            labels = sorted(preds[0].keys())
            out[f"{cid}/labels"] = np.array(labels)
            out[f"{cid}/scores"] = np.array([[float(p[k]) for k in labels] for p in preds], dtype=np.float64)
            out[f"{cid}/features"] = mdl.preprocessor.feature_buffer.astype(np.float32)
            if cid == "f1280":
                out["init/model_inputs"] = np.array([mdl.model_inputs[n] for n in sorted(mdl.models)])
                out["init/model_outputs"] = np.array([mdl.model_outputs[n] for n in sorted(mdl.models)])
        # Model(vad_threshold > 0): the reference opens resources/models/silero_vad.onnx inside its own package (vad.py:60-67);
        # the evaluator is pointed at the exporter-written stand-in instead (the real file's graph is not available)
        from openwakeword_amd import weights as W
        vad_path = os.path.join(d, "silero_vad.onnx")
        TE.export_vad(W.synthetic_vad(cases.ONNX_VAD_SEED), vad_path)
        mini_ort.REDIRECT["silero_vad.onnx"] = vad_path
        for cid, head_names, clip, kw, thr in cases.ONNX_VAD_CASES:
            np.random.seed(cases.SEED_NP)
            mdl = openwakeword.Model(wakeword_models=[paths[n] for n in head_names], inference_framework="onnx", vad_threshold=thr,
                                     melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
            preds = mdl.predict_clip(clips[clip], **kw)
            labels = sorted(preds[0].keys())
            out[f"{cid}/labels"] = np.array(labels)
            out[f"{cid}/scores"] = np.array([[float(p[k]) for k in labels] for p in preds], dtype=np.float64)
            out[f"{cid}/vad"] = np.array(list(mdl.vad.prediction_buffer), dtype=np.float64)
            out[f"{cid}/ring"] = np.array([list(mdl.prediction_buffer[k]) for k in labels], dtype=np.float64)
        # Model(enable_speex_noise_suppression=True): `speexdsp_ns` is oracle/fake_speex.py (a deterministic stateful stand-in with the
        # package's interface -- the real one is not in this image); what is pinned is the reference's ORDER of operations around it
        from oracle import fake_speex
        sys.modules["speexdsp_ns"] = fake_speex.as_module()
        for cid, head_names, clip, kw, thr in cases.ONNX_SPEEX_CASES:
            np.random.seed(cases.SEED_NP)
            mdl = openwakeword.Model(wakeword_models=[paths[n] for n in head_names], inference_framework="onnx", vad_threshold=thr,
                                     enable_speex_noise_suppression=True,
                                     melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
            assert type(mdl.speex_ns).__module__ == "oracle.fake_speex"
            preds = mdl.predict_clip(clips[clip], **kw)
            labels = sorted(preds[0].keys())
            out[f"{cid}/labels"] = np.array(labels)
            out[f"{cid}/scores"] = np.array([[float(p[k]) for k in labels] for p in preds], dtype=np.float64)
            out[f"{cid}/features"] = mdl.preprocessor.feature_buffer.astype(np.float32)
            out[f"{cid}/raw_mid"] = np.array(list(mdl.preprocessor.raw_data_buffer)[-24000:-16000], dtype=np.int16)   # what the preprocessor was fed (before the trailing second of padding)
            if thr > 0:
                out[f"{cid}/vad"] = np.array(list(mdl.vad.prediction_buffer), dtype=np.float64)
        # a custom verifier model re-scoring the frames the base model likes (model.py:320-328)
        import verifier_fixture
        cid, head_names, clip, kw, target, vthr = cases.ONNX_VERIFIER
        pkl = verifier_fixture.write(os.path.join(d, "verifier.pkl"))
        np.random.seed(cases.SEED_NP)
        mdl = openwakeword.Model(wakeword_models=[paths[n] for n in head_names], inference_framework="onnx",
                                 custom_verifier_models={target: pkl}, custom_verifier_threshold=vthr,
                                 melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
        preds = mdl.predict_clip(clips[clip], **kw)
        labels = sorted(preds[0].keys())
        out[f"{cid}/labels"] = np.array(labels)
        out[f"{cid}/scores"] = np.array([[float(p[k]) for k in labels] for p in preds], dtype=np.float64)
        # label mapping given by the caller, parent lookup, false-positive mining from a WAV file
        cid, head_names, clip, mapping, fthr = cases.ONNX_MAPPING
        wav = os.path.join(d, "clip.wav")
        with wave.open(wav, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes(clips[clip].tobytes())
        np.random.seed(cases.SEED_NP)
        mdl = openwakeword.Model(wakeword_models=[paths[n] for n in head_names], inference_framework="onnx",
                                 class_mapping_dicts=mapping,
                                 melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
        preds = mdl.predict_clip(clips[clip], chunk_size=1280)
        labels = sorted(preds[0].keys())
        out[f"{cid}/labels"] = np.array(labels)
        out[f"{cid}/scores"] = np.array([[float(p[k]) for k in labels] for p in preds], dtype=np.float64)
        out[f"{cid}/parents"] = np.array([mdl.get_parent_model_from_label(k) for k in labels])
        np.random.seed(cases.SEED_NP)
        mdl.reset()
        pos = mdl._get_positive_prediction_frames(wav, threshold=fthr, return_type="features")
        out[f"{cid}/positive_labels"] = np.array(sorted(pos.keys()))
        for k, v in pos.items():
            out[f"{cid}/positive/{k}"] = np.asarray(v, np.float32)
        # the bulk feature path (utils.py:243-385, what compute_features_from_generator / the training scripts call)
        from openwakeword.utils import AudioFeatures
        F = AudioFeatures(melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"],
                          inference_framework="onnx", ncpu=2)
        j = clips["hey_jane"]
        x = np.stack([j[:32000], j[5000:37000] // 4, np.zeros(32000, np.int16), np.resize(clips["alexa_test"], 32000)])
        out["embed/pcm"] = x
        out["embed/embed_clips"] = F.embed_clips(x, batch_size=3, ncpu=2).astype(np.float32)
        out["embed/melspec_batch"] = F._get_melspectrogram_batch(x, batch_size=2, ncpu=1).astype(np.float32)
        out["embed/get_embeddings"] = np.asarray(F._get_embeddings(x[1]), np.float32)
        out["embed/melspectrogram"] = np.asarray(F._get_melspectrogram(x[0][:12345]), np.float32)
        out["embed/shape_2s"] = np.array(F.get_embedding_shape(2.0))
        cid, head_names, clip, sizes = cases.ONNX_SEQUENCE
        np.random.seed(cases.SEED_NP)
        mdl = openwakeword.Model(wakeword_models=[paths[n] for n in head_names], inference_framework="onnx",
                                 melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
        rows, o = [], 0
        for n in sizes:
            p = mdl.predict(clips[clip][o:o + n])
            o += n
            labels = sorted(p.keys())
            rows.append([float(p[k]) for k in labels])
        out[f"{cid}/labels"] = np.array(labels)
        out[f"{cid}/scores"] = np.array(rows, dtype=np.float64)
        # calls longer than a HIP handle's mel buffer: one run of the melspectrogram graph per call = one clamp floor per call
        cid, head_names, sizes = cases.ONNX_LONG
        x = cases.long_call_pcm(clips["alexa_test"])
        assert len(x) == sum(sizes)
        np.random.seed(cases.SEED_NP)
        mdl = openwakeword.Model(wakeword_models=[paths[n] for n in head_names], inference_framework="onnx",
                                 melspec_model_path=paths["melspectrogram"], embedding_model_path=paths["embedding_model"])
        rows, o = [], 0
        for n in sizes:
            p = mdl.predict(x[o:o + n])
            o += n
            labels = sorted(p.keys())
            rows.append([float(p[k]) for k in labels])
        out[f"{cid}/labels"] = np.array(labels)
        out[f"{cid}/scores"] = np.array(rows, dtype=np.float64)
        out[f"{cid}/features"] = mdl.preprocessor.feature_buffer.astype(np.float32)
        out[f"{cid}/mel_tail"] = mdl.preprocessor.melspectrogram_buffer[-8 * 15:].astype(np.float32)
    path = os.path.join(os.path.dirname(__file__), "ref_onnx_files.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for cid, *_ in cases.ONNX_FILE_CASES + cases.ONNX_VAD_CASES:
        s = out[f"{cid}/scores"]
        print(f"  {cid:10s} frames={s.shape[0]:3d} labels={list(out[cid + '/labels'])} max={s.max(axis=0).round(4)}")


if __name__ == "__main__":
    main()
