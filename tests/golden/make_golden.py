#!/usr/bin/env python3
"""
Generate tests/golden/ref_streaming.npz by running the REFERENCE's own Python
(/root/reference/openwakeword: utils.AudioFeatures, model.Model.predict / predict_clip / reset)
with `oracle.fake_ort` standing in for onnxruntime and `oracle.oww_oracle` supplying the stage math
at the reference's three backend seams (utils.py:87,93; model.py:137-138).

Run only in the build container (needs /root/reference):   python tests/golden/make_golden.py
The GPU box never reads /root/reference; it consumes the committed .npz.

What the vectors pin: the streaming / buffering / post-processing semantics of rows A,B,D,F,H of
SURVEY §8a (1280-alignment + remainder carry, 5-then-8 mel rows, ones-filled mel ring, 41-row random
feature ring, multi-chunk max, <1280 reuse, first-5 zeroing, patience, debounce, reset).  They do NOT
pin the stage math against the real .onnx graphs (absent) -- that part stays "parity unpinned".
"""
import os
import sys
import wave

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))
REF = "/root/reference"

from oracle import fake_ort, oww_oracle as O          # noqa: E402
from openwakeword_amd import weights as W             # noqa: E402
import cases                                          # noqa: E402


def main():
    sys.modules["onnxruntime"] = fake_ort.as_module()
    sys.path.insert(0, REF)
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    fake_ort.STAGES["mel"] = lambda x: O.mel_stage(x, np.float32)
    fake_ort.STAGES["embed"] = lambda x: O.embedding_stage(x, emb, np.float32)
    heads = {}
    for name in set(sum((c[1] for c in cases.CLIP_CASES), [])):
        h = W.synthetic_head(name, cases.SEED_WEIGHTS)
        heads[name] = h
        fake_ort.HEADS[name] = ((lambda x, _h=h: O.head_stage(x, _h, np.float32)), h["T"], h["n_out"])

    import openwakeword                                   # the real reference package
    assert os.path.realpath(openwakeword.__file__).startswith(REF)

    out = {}
    clips = {}
    for c in cases.CLIPS:
        with wave.open(os.path.join(REF, "tests", "data", c + ".wav"), "rb") as f:
            assert f.getframerate() == 16000 and f.getnchannels() == 1 and f.getsampwidth() == 2
            clips[c] = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16).copy()
        out["pcm/" + c] = clips[c]

    for cid, head_names, clip, kw in cases.CLIP_CASES:
        np.random.seed(cases.SEED_NP)
        mdl = openwakeword.Model(wakeword_models=list(head_names), inference_framework="onnx")
        if cid == "c1280":
            # structural pins quoted in SURVEY §3.3 / §8c
            assert mdl.preprocessor.feature_buffer.shape == (41, 96)
            assert mdl.preprocessor.melspectrogram_buffer.shape == (76, 32)
            out["init/feature_buffer"] = mdl.preprocessor.feature_buffer.astype(np.float32)
        preds = mdl.predict_clip(clips[clip], **kw)
        labels = sorted(preds[0].keys())
        out[f"{cid}/labels"] = np.array(labels)
        out[f"{cid}/scores"] = np.array([[float(p[k]) for k in labels] for p in preds], dtype=np.float64)
        out[f"{cid}/features"] = mdl.preprocessor.feature_buffer.astype(np.float32)
        out[f"{cid}/mel_tail"] = mdl.preprocessor.melspectrogram_buffer[-16:].astype(np.float32)
        if cid == "c1280":
            # reset + second clip on the same object (test_models.py:166,233-257 behaviour)
            np.random.seed(cases.SEED_NP + 1)
            mdl.reset()
            preds2 = mdl.predict_clip(clips["hey_mycroft_test"], chunk_size=1280)
            out["reset/scores"] = np.array([[float(p[k]) for k in labels] for p in preds2], dtype=np.float64)

    # stage-level vectors of the restatement itself (regression pins, weight independent for mel)
    x = clips["hey_mycroft_test"][:12560].astype(np.float32)[None]
    out["stage/mel_in"] = clips["hey_mycroft_test"][:12560]
    out["stage/mel_out"] = O.mel_stage(x, np.float32)[0, 0]
    win = (out["stage/mel_out"][:76] / 10 + 2).astype(np.float32)
    out["stage/embed_out"] = O.embedding_stage(win[None, :, :, None], emb, np.float32).reshape(96)

    path = os.path.join(os.path.dirname(__file__), "ref_streaming.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for cid, *_ in cases.CLIP_CASES:
        s = out[f"{cid}/scores"]
        print(f"  {cid:10s} frames={s.shape[0]:3d} labels={list(out[cid + '/labels'])} max={s.max(axis=0).round(4)}")


if __name__ == "__main__":
    main()
