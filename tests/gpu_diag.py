#!/usr/bin/env python3
"""GPU bring-up diagnostic: per-stage / per-layer error report of the HIP path against the CPU oracle.
Not a test (the pytest -m gpu suite asserts the same quantities); prints a full table so that one
gpurun call localises any mismatch.   python tests/gpu_diag.py [--valu] > gpurun_out/diag.txt"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import oww_oracle as O                      # noqa: E402
from openwakeword_amd import weights as W               # noqa: E402
from openwakeword_amd.engine import StreamEngine, LAYER_NEW_SHAPES   # noqa: E402


def err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    i = np.unravel_index(np.argmax(d), d.shape) if d.size else ()
    return float(d.max()) if d.size else 0.0, i


def section(name):
    print(f"\n=== {name} " + "=" * max(0, 70 - len(name)), flush=True)


def main():
    use_mfma = "--valu" not in sys.argv
    emb = W.synthetic_embedding(1234)
    heads = {n: W.synthetic_head(n, 1234) for n in ["alexa", "hey_mycroft", "hey_jarvis"]}
    S = 6
    t0 = time.time()
    eng = StreamEngine(S, heads, emb, max_chunks=2, use_mfma=use_mfma, debug_layers=True)
    print(f"engine up in {time.time() - t0:.2f}s  mfma={use_mfma}  labels={eng.n_labels}")
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_streaming.npz"))

    # ------------------------------------------------------------------ mel stage (clip mode)
    section("mel stage (oww_mel) vs oracle f32 / f64 [dB]")
    r = np.random.default_rng(0)
    cases = {
        "noise1760": W.synthetic_pcm(3, 1760, seed=1),
        "noise12560": W.synthetic_pcm(2, 12560, seed=2),
        "mycroft12560": g["pcm/hey_mycroft_test"][:12560][None],
        "loud": (r.integers(-32768, 32767, (2, 4000))).astype(np.int16),
        "quiet": (r.integers(-3, 4, (1, 1760))).astype(np.int16),
        "zeros": np.zeros((1, 1760), np.int16),
        "odd_n": W.synthetic_pcm(2, 1999, seed=3),
    }
    for name, pcm in cases.items():
        try:
            got = eng.mel(pcm)
            w32 = O.mel_stage(pcm.astype(np.float32), np.float32)[:, 0]
            w64 = O.mel_stage(pcm.astype(np.float32), np.float64)[:, 0]
            e32, i32 = err(got, w32)
            e64, i64 = err(got, w64)
            eo, _ = err(w32, w64)
            print(f"{name:14s} shape={got.shape} |gpu-f32|={e32:.3e} |gpu-f64|={e64:.3e} at {i64} (oracle f32-f64 {eo:.3e}) range [{w64.min():.2f},{w64.max():.2f}]")
        except Exception:
            print(name, "FAILED"); traceback.print_exc(file=sys.stdout)

    # ------------------------------------------------------------------ embedding stage (full window via incremental kernels)
    section("embedding stage (oww_embed) vs oracle")
    try:
        mel = (O.mel_stage(g["pcm/hey_mycroft_test"][:12560 + 160 * 16].astype(np.float32)[None])[0, 0] / 10 + 2).astype(np.float32)
        rows = 76 + 8 * 2
        batch = np.stack([mel[:rows], mel[1:rows + 1], r.normal(10, 1.5, (rows, 32)).astype(np.float32)])
        got = eng.embed(batch)
        want = np.stack([np.stack([O.embedding_stage(b[8 * j: 8 * j + 76][None, :, :, None], emb).reshape(96) for j in range(3)]) for b in batch])
        want64 = np.stack([np.stack([O.embedding_stage(b[8 * j: 8 * j + 76][None, :, :, None], emb, np.float64).reshape(96) for j in range(3)]) for b in batch])
        for j in range(3):
            print(f"window {j}: |gpu-f32|={err(got[:, j], want[:, j])[0]:.3e} |gpu-f64|={err(got[:, j], want64[:, j])[0]:.3e}  (|emb|max {np.abs(want64).max():.2f})")
        eng.reset()
    except Exception:
        print("embed FAILED"); traceback.print_exc(file=sys.stdout)

    # ------------------------------------------------------------------ head stage
    section("head stage (oww_head) vs oracle")
    try:
        feats = r.normal(0, 2.0, (5, 16, 96)).astype(np.float32)
        for name, h in heads.items():
            got = eng.head(name, feats)
            want = O.head_stage(feats, h)
            print(f"{name:12s} |gpu-oracle|={err(got, want)[0]:.3e}  scores {np.round(got.ravel(), 4)}")
    except Exception:
        print("head FAILED"); traceback.print_exc(file=sys.stdout)

    # ------------------------------------------------------------------ streaming
    section("streaming: mel rows / embeddings / scores per step, layers at the last step")
    try:
        n_steps = 24
        pcm = W.synthetic_pcm(S, 1280 * n_steps, seed=11)
        pcm[1] = np.resize(g["pcm/alexa_test"], 1280 * n_steps)
        pcm[2] = np.resize(np.concatenate([np.zeros(4000, np.int16), g["pcm/hey_mycroft_test"]]), 1280 * n_steps)
        models = []
        for s in range(S):
            noise = W.synthetic_pcm(1, 64000, seed=100 + s, rms=600.0)[0]
            m = O.OracleModel(heads, emb, init_noise=noise)
            models.append(m)
            eng.reset([s], m.preprocessor.features[-eng.feature_ring:])
        worst = {"mel": 0.0, "emb": 0.0, "score": 0.0}
        for t in range(n_steps):
            x = pcm[:, 1280 * t: 1280 * (t + 1)]
            got = eng.step(x)
            line = []
            for s in range(S):
                want = models[s].predict(x[s])
                ws = np.array([want[k] for k in heads])
                em, _ = err(eng.get_mel(s, 8)[-(5 if t == 0 else 8):], models[s].preprocessor.mel_rows[-(5 if t == 0 else 8):])
                ee, _ = err(eng.get_features(s, 1)[0], models[s].preprocessor.features[-1])
                es, _ = err(got[s], ws)
                worst["mel"] = max(worst["mel"], em); worst["emb"] = max(worst["emb"], ee); worst["score"] = max(worst["score"], es)
                line.append(f"{em:.1e}/{ee:.1e}/{es:.1e}")
            print(f"step {t:2d} mel/emb/score err per stream: " + "  ".join(line), flush=True)
        print("worst:", worst)
        print("last scores gpu   :", np.round(got, 4).tolist())
        # layers at the last step, stream 0 and 1
        for s in (0, 1):
            win = models[s].preprocessor.mel_rows[-76:].astype(np.float32)
            _, layers = O.embedding_stage(win[None, :, :, None], emb, np.float32, return_layers=True)
            # oracle returns post-pool tensors for pooled layers; recompute pre-pool for comparison
            h = win[None, :, :, None]
            pre = []
            for li, (kh, kw, ci, co, relu_first, bn, pool) in enumerate(O.CNN_LAYERS):
                h = O._conv(h, emb["conv"][li].astype(np.float32))
                if relu_first:
                    h = np.maximum(h, np.float32(0))
                if bn:
                    sc, sh = O.bn_fold(*emb["bn"][li])
                    h = O._activation(h * sc + sh)
                pre.append(h)
                if pool:
                    h = O._pool(h, *pool)
            for li in range(20):
                rws = LAYER_NEW_SHAPES[li][0]
                got_l = eng.debug_layer(s, li)
                want_l = pre[li][0, -rws:]
                e, idx = err(got_l, want_l)
                print(f"stream {s} layer {li:2d} new-rows {got_l.shape}: max err {e:.3e} at {idx}  (|x|max {np.abs(want_l).max():.2f})")
    except Exception:
        print("streaming FAILED"); traceback.print_exc(file=sys.stdout)

    # ------------------------------------------------------------------ multi-chunk call
    section("multi-chunk call (n_chunks=2) vs oracle")
    try:
        eng.reset()
        noise = W.synthetic_pcm(1, 64000, seed=5, rms=600.0)[0]
        ms = [O.OracleModel(heads, emb, init_noise=noise) for _ in range(S)]
        eng.reset(None, ms[0].preprocessor.features[-eng.feature_ring:])
        pcm = W.synthetic_pcm(S, 2560 * 8, seed=12)
        w = 0.0
        for t in range(8):
            x = pcm[:, 2560 * t: 2560 * (t + 1)]
            got = eng.step(x)
            for s in range(S):
                want = ms[s].predict(x[s])
                w = max(w, err(got[s], np.array([want[k] for k in heads]))[0])
        print(f"worst score err over 8 double-chunk calls: {w:.3e}")
    except Exception:
        print("multi-chunk FAILED"); traceback.print_exc(file=sys.stdout)
    print("\nDONE")


if __name__ == "__main__":
    main()
