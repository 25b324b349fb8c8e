#!/usr/bin/env python3
"""Randomised differential run of the drop-in surface: openwakeword_amd.Model.predict (HIP library) against oracle.OracleModel
(the CPU restatement of /root/reference/openwakeword/model.py:232-386 + utils.py:387-463) over random CALL SEQUENCES -- calls of
0 ... 6000 samples (remainder carry-over, multi-chunk calls, calls longer than max_chunks), amplitude regimes from digital silence
to full scale, random head sets (binary, gated, multiclass), patience / threshold or debounce, a reset() in mid-sequence.

Test infrastructure (imports oracle/): never part of the product path.
usage:  python tools/fuzz_model_vs_oracle.py [first_seed] [n_seeds] [--stub] [--long=N_CALLS]
        --stub  serve the host shim from the oracle engine (tests/stub_engine.py): checks the host logic alone, no GPU needed."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from openwakeword_amd import model as M
from openwakeword_amd import weights as W
from oracle import oww_oracle as O

TOL = 1e-4          # scores (the GPU tier's tolerance against the oracle; north star 1e-3)
SIZES = [0, 1, 159, 160, 399, 400, 640, 1279, 1280, 1281, 1920, 2559, 2560, 2561, 3000, 3840, 5000, 6000]
HEADS = ["alexa", "hey_mycroft", "hey_jarvis", "hey_rhasspy", "timer", "weather"]


def one_seed(seed: int, n_calls: int = 0, sizes=None) -> dict:
    """n_calls > 0: a LONG sequence (past the 120-row feature ring, the 970-row mel ring = 121 steps, the 30-deep score rings and
    the 125-deep VAD ring of the reference: every device ring wraps at least once)."""
    r = np.random.default_rng(seed)
    sizes = SIZES if sizes is None else sizes
    names = sorted(r.choice(HEADS, size=int(r.integers(1, 4)), replace=False).tolist())
    wseed = int(r.integers(1, 1 << 20))
    emb = W.synthetic_embedding(wseed)
    heads = {n: W.synthetic_head(n, wseed) for n in names}
    mapping = {n: M.model_class_mappings[n] for n in names if n in M.model_class_mappings}
    max_chunks = int(r.choice([1, 2, 32]))
    mode = int(r.integers(0, 3))                     # 0 plain, 1 patience, 2 debounce
    kw = {}
    if mode == 1:
        kw = dict(patience={n: int(r.integers(1, 4)) for n in names}, threshold={n: float(r.choice([1e-4, 0.003, 0.03, 0.1, 0.3, 0.5])) for n in names})
    elif mode == 2:
        kw = dict(debounce_time=float(r.choice([0.25, 1.0])), threshold={n: float(r.choice([1e-4, 0.003, 0.03, 0.1, 0.3, 0.5])) for n in names})
    vad_thr = float(r.choice([0.0, 0.0, 0.3, 0.6]))     # > 0: the VAD gate (model.py:366-381) with the same pseudo network either side
    vkw_h, vkw_o = {}, {}
    if vad_thr > 0:
        from oracle.pseudo_vad import PseudoVadSession
        vkw_h = dict(vad_threshold=vad_thr, vad_session=PseudoVadSession())
        vkw_o = dict(vad_threshold=vad_thr, vad_session=PseudoVadSession())
    np.random.seed(seed)
    hip = M.Model(wakeword_models=names, weights={"embedding": emb, "heads": heads}, max_chunks=max_chunks, **vkw_h)
    np.random.seed(seed)
    ora = O.OracleModel(heads, emb, class_mapping=mapping, **vkw_o)
    n_calls = n_calls or int(r.integers(12, 30))
    reset_at = int(r.integers(4, n_calls)) if r.random() < 0.5 else -1
    worst, near, n_scores, n_raised = 0.0, 0, 0, 0
    try:
        for c in range(n_calls):
            if c == reset_at:
                np.random.seed(seed + 1)
                hip.reset()
                np.random.seed(seed + 1)
                ora.reset()
            n = int(r.choice(sizes))
            amp = float(r.choice([0.0, 1.0, 60.0, 3000.0, 12000.0, 40000.0]))
            x = np.clip(np.round(r.normal(0.0, 1.0, n) * amp), -32768, 32767).astype(np.int16)
            if r.random() < 0.15 and n:
                x[:] = np.where(np.arange(n) // int(r.integers(3, 80)) % 2, 20000, -20000)     # square wave
            # (the reference raises ZeroDivisionError from its debounce rule when a call prepares no frame while the label's last
            #  score is non-zero, model.py:355: error behaviour is part of the surface, so both sides must raise alike and carry on)
            res = []
            for mdl in (hip, ora):
                try:
                    res.append(mdl.predict(x, **kw))
                except (ZeroDivisionError, ValueError) as e:
                    res.append(type(e))
            a, b = res
            if isinstance(a, type) or isinstance(b, type):
                assert a is b, f"seed {seed} call {c} n={n}: hip {a!r} oracle {b!r} kw={kw}"
                n_raised += 1
                continue
            assert sorted(a) == sorted(b), (seed, c, sorted(a), sorted(b))
            for k in a:
                va, vb = float(a[k]), float(b[k])
                n_scores += 1
                if abs(va - vb) <= TOL:
                    worst = max(worst, abs(va - vb))
                    continue
                # a post-processing decision may flip when a score sits within TOL of its threshold: then one side is 0.0 exactly
                par = ora.parent_of(k)
                thr = kw.get("threshold", {}).get(par)
                hist_a = list(hip.prediction_buffer[k])[-31:]
                hist_b = list(ora.prediction_buffer[k])[-31:]
                close = thr is not None and any(abs(float(v) - thr) <= TOL for v in hist_a + hist_b)
                if not close and vad_thr > 0 and (va == 0.0 or vb == 0.0):
                    wa, wb = list(hip.vad.prediction_buffer)[-7:-4], list(ora.vad.ring)[-7:-4]
                    close = any(abs(float(v) - vad_thr) <= 1e-6 for v in wa + wb)
                if close and (va == 0.0 or vb == 0.0):
                    near += 1
                    # keep the two histories in step so that one borderline frame does not cascade
                    hip.prediction_buffer[k][-1] = ora.prediction_buffer[k][-1]
                    continue
                raise AssertionError(f"seed {seed} call {c} n={n} amp={amp} label {k}: hip {va!r} oracle {vb!r} heads={names} "
                                     f"max_chunks={max_chunks} kw={kw}")
    finally:
        hip.close()
    return dict(seed=seed, heads=names, calls=n_calls, scores=n_scores, worst=worst, borderline=near, mode=mode, max_chunks=max_chunks, vad=vad_thr, raised=n_raised)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if args else 1
    count = int(args[1]) if len(args) > 1 else 20
    long_calls = 0
    for a in sys.argv[1:]:
        if a.startswith("--long="):
            long_calls = int(a.split("=")[1])
    if "--stub" in sys.argv:
        from stub_engine import OracleEngine
        M.make_engine = lambda n_streams, heads, embedding, use_mfma=None, **kw: OracleEngine(n_streams, heads, embedding, **kw)
    bad, worst, t0, total = 0, 0.0, time.time(), 0
    for seed in range(first, first + count):
        try:
            rec = one_seed(seed, long_calls, [1280, 1280, 1280, 1280, 640, 2560, 1000, 0] if long_calls else None)
            worst = max(worst, rec["worst"])
            total += rec["scores"]
            print(rec, flush=True)
        except AssertionError as e:
            bad += 1
            print("FAILED", str(e)[:600], flush=True)
    print(f"fuzz_model_vs_oracle seeds {first}..{first + count - 1}: {count - bad} ok, {bad} failed, {total} scores compared, "
          f"max |hip - oracle| {worst:.3g} (tolerance {TOL}), {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
