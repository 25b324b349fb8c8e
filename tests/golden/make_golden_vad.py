#!/usr/bin/env python3
"""
Generate tests/golden/ref_vad.npz: the REFERENCE's own Model.predict_clip with `vad_threshold > 0`
(/root/reference/openwakeword/model.py:208-210, 366-381 and vad.py:54-130 executed unmodified), `oracle.fake_ort` standing in
for onnxruntime, the oracle's stage math behind the three model seams and `oracle.pseudo_vad.PseudoVadSession` behind the VAD
class's session (the real silero_vad.onnx is not available: the vectors pin the gate, the sub-framing, the state carry and the
reset behaviour -- not the voice-activity network).

Run only in the build container (needs /root/reference):   python tests/golden/make_golden_vad.py
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))
REF = "/root/reference"

from oracle import fake_ort, oww_oracle as O            # noqa: E402
from oracle.pseudo_vad import PseudoVadSession          # noqa: E402
from openwakeword_amd import weights as W               # noqa: E402
import cases                                            # noqa: E402


def main():
    sys.modules["onnxruntime"] = fake_ort.as_module()
    sys.path.insert(0, REF)
    emb = W.synthetic_embedding(cases.SEED_WEIGHTS)
    fake_ort.STAGES["mel"] = lambda x: O.mel_stage(x, np.float32)
    fake_ort.STAGES["embed"] = lambda x: O.embedding_stage(x, emb, np.float32)
    for name in set(sum((c[1] for c in cases.VAD_CASES), [])):
        h = W.synthetic_head(name, cases.SEED_WEIGHTS)
        fake_ort.HEADS[name] = ((lambda x, _h=h: O.head_stage(x, _h, np.float32)), h["T"], h["n_out"])

    import openwakeword
    assert os.path.realpath(openwakeword.__file__).startswith(REF)
    clips = dict(np.load(os.path.join(os.path.dirname(__file__), "ref_streaming.npz")))
    out = {}
    for cid, head_names, clip, kw, thr in cases.VAD_CASES:
        fake_ort.STAGES["vad"] = PseudoVadSession()
        np.random.seed(cases.SEED_NP)
        mdl = openwakeword.Model(wakeword_models=list(head_names), inference_framework="onnx", vad_threshold=thr)
        preds = mdl.predict_clip(clips["pcm/" + clip], **kw)
        labels = sorted(preds[0].keys())
        out[f"{cid}/labels"] = np.array(labels)
        out[f"{cid}/scores"] = np.array([[float(p[k]) for k in labels] for p in preds], dtype=np.float64)
        out[f"{cid}/vad"] = np.array(list(mdl.vad.prediction_buffer), dtype=np.float64)
        out[f"{cid}/ring"] = np.array([list(mdl.prediction_buffer[k]) for k in labels], dtype=np.float64)
        if cid == "vad03":
            # Model.reset() does not touch the VAD (model.py:226-230): the second clip starts with a full VAD ring
            np.random.seed(cases.SEED_NP + 1)
            mdl.reset()
            preds2 = mdl.predict_clip(clips["pcm/hey_mycroft_test"], chunk_size=1280)
            out["vadreset/scores"] = np.array([[float(p[k]) for k in labels] for p in preds2], dtype=np.float64)
            out["vadreset/vad"] = np.array(list(mdl.vad.prediction_buffer), dtype=np.float64)
    path = os.path.join(os.path.dirname(__file__), "ref_vad.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for cid, *_ in cases.VAD_CASES:
        s, v = out[f"{cid}/scores"], out[f"{cid}/vad"]
        print(f"  {cid:8s} frames={s.shape[0]:3d} nonzero={int((s != 0).sum()):3d} ring_nonzero={int((out[cid + '/ring'] != 0).sum()):3d} "
              f"vad min/max={v.min():.3f}/{v.max():.3f}")


if __name__ == "__main__":
    main()
