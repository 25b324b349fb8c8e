#!/usr/bin/env python3
"""Randomised differential run of the MANY-STREAM surface: openwakeword_amd.BatchedModel (device post-processing: first-five zeroing,
patience / debounce on the 30-deep score rings; masked steps; per-stream resets; the device voice-activity stand-in + gate) against one
oracle.OracleModel per stream (model.py:232-386 restated) over sequences long enough for every device ring to wrap.

Per seed: S streams with their own audio regime each, k chunks per call (fixed per seed: the debounce span depends on it), a random
participation mask in a third of the one-chunk steps (a masked-out stream makes no predict() call in the oracle), resets of random
stream subsets in mid-run (new feature ring handed over, score rings and frame counters cleared, VAD state kept: model.py:226-230).

Test infrastructure (imports oracle/): never part of the product path.
usage:  python tools/fuzz_batched_vs_oracle.py [first_seed] [n_seeds] [--steps=N]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np

from openwakeword_amd import model as M
from openwakeword_amd import weights as W
from oracle import oww_oracle as O
from oracle import vad_standin as V

TOL = 1e-4
TOL_VAD = 1e-3            # a VAD score this close to the gate's threshold may fall either side (tests/test_parity_scale.py)
HEADS = ["alexa", "hey_mycroft", "hey_jarvis", "hey_rhasspy", "timer", "weather"]


def one_seed(seed: int, steps: int = 150) -> dict:
    r = np.random.default_rng(seed)
    S = int(r.choice([3, 7, 12]))
    names = sorted(r.choice(HEADS, size=int(r.integers(1, 4)), replace=False).tolist())
    wseed = int(r.integers(1, 1 << 20))
    emb = W.synthetic_embedding(wseed)
    heads = {n: W.synthetic_head(n, wseed) for n in names}
    mapping = {n: M.model_class_mappings[n] for n in names if n in M.model_class_mappings}
    k = int(r.choice([1, 1, 1, 2, 3]))
    mode = int(r.integers(0, 3))
    kw = {}
    if mode == 1:
        kw = dict(patience={n: int(r.integers(1, 4)) for n in names}, threshold={n: float(r.choice([1e-4, 0.003, 0.03, 0.1, 0.3, 0.5])) for n in names})
    elif mode == 2:
        kw = dict(debounce_time=float(r.choice([0.25, 1.0])), threshold={n: float(r.choice([1e-4, 0.003, 0.03, 0.1, 0.3, 0.5])) for n in names})
    vad_thr = float(r.choice([0.0, 0.0, 0.35, 0.5]))
    vw = W.synthetic_vad(wseed) if vad_thr > 0 else None
    if vad_thr > 0:
        k = 1                                   # (the device network runs on one chunk's two sub-frames per step)
    bm = M.BatchedModel(S, names, weights={"embedding": emb, "heads": heads}, max_chunks=k, vad_weights=vw, vad_threshold=vad_thr)
    ring = bm.engine.feature_ring
    noise = np.zeros(64000, np.int16)

    def fresh(s):
        okw = dict(vad_threshold=vad_thr, vad_session=V.StandinVadSession(vw)) if vad_thr > 0 else {}
        m = O.OracleModel(heads, emb, class_mapping=mapping, init_noise=noise, **okw)
        f = r.normal(0.0, 1.0, (ring, 96)).astype(np.float32)
        m.preprocessor.features = f.copy()
        bm.reset([s], f)
        return m

    worst, near, n_scores, n_masked, n_resets, n_nonzero = 0.0, 0, 0, 0, 0, 0
    try:
        oras = [fresh(s) for s in range(S)]
        bm.set_postproc(chunk_samples=1280 * k, **kw)
        amps = r.choice([0.0, 60.0, 3000.0, 12000.0, 30000.0], size=S)
        reset_steps = set(int(v) for v in r.integers(10, steps, size=2))
        for t in range(steps):
            if t in reset_steps:
                ids = sorted(set(int(i) for i in r.integers(0, S, size=max(1, S // 3))))
                for s in ids:
                    f = r.normal(0.0, 1.0, (ring, 96)).astype(np.float32)
                    oras[s].reset()                                   # model.py:226-230: buffers + preprocessor; the VAD stays
                    oras[s].preprocessor.features = f.copy()
                    bm.reset([s], f)
                n_resets += len(ids)
            if r.random() < 0.1:
                amps = r.choice([0.0, 60.0, 3000.0, 12000.0, 30000.0], size=S)
            x = np.clip(np.round(r.normal(0.0, 1.0, (S, 1280 * k)) * amps[:, None]), -32768, 32767).astype(np.int16)
            if k == 1 and r.random() < 0.33:
                active = r.random(S) < 0.6
                got = bm.predict_active(x, active)
                n_masked += int((~active).sum())
            else:
                active = np.ones(S, bool)
                got = bm.predict_batch(x)
            for s in range(S):
                if not active[s]:
                    continue
                want = oras[s].predict(x[s], **kw)
                for c, lab in enumerate(bm.labels):
                    va, vb = float(got[s, c]), float(want[lab])
                    n_scores += 1
                    n_nonzero += int(vb != 0.0)
                    if abs(va - vb) <= TOL:
                        worst = max(worst, abs(va - vb))
                        continue
                    thr = kw.get("threshold", {}).get(oras[s].parent_of(lab))
                    hist = list(oras[s].prediction_buffer[lab])[-31:]
                    close = thr is not None and any(abs(float(v) - thr) <= TOL for v in hist + [va, vb])
                    if not close and vad_thr > 0:
                        close = any(abs(float(v) - vad_thr) <= TOL_VAD for v in list(oras[s].vad.ring)[-7:-4])
                    if close and (va == 0.0 or vb == 0.0):
                        near += 1
                        continue
                    raise AssertionError(f"seed {seed} step {t} stream {s} label {lab}: hip {va!r} oracle {vb!r} S={S} k={k} heads={names} "
                                         f"kw={kw} vad={vad_thr} active={bool(active[s])}")
    finally:
        bm.close()
    return dict(seed=seed, S=S, k=k, heads=names, steps=steps, scores=n_scores, nonzero=n_nonzero, worst=worst, borderline=near, mode=mode, vad=vad_thr,
                masked_out=n_masked, stream_resets=n_resets)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if args else 1
    count = int(args[1]) if len(args) > 1 else 6
    steps = 150
    for a in sys.argv[1:]:
        if a.startswith("--steps="):
            steps = int(a.split("=")[1])
    bad, worst, t0, total, near = 0, 0.0, time.time(), 0, 0
    for seed in range(first, first + count):
        try:
            rec = one_seed(seed, steps)
            worst = max(worst, rec["worst"]); total += rec["scores"]; near += rec["borderline"]
            print(rec, flush=True)
        except AssertionError as e:
            bad += 1
            print("FAILED", str(e)[:600], flush=True)
    print(f"fuzz_batched_vs_oracle seeds {first}..{first + count - 1} x {steps} steps: {count - bad} ok, {bad} failed, {total} scores compared, "
          f"{near} borderline decisions skipped, max |hip - oracle| {worst:.3g} (tolerance {TOL}), {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
