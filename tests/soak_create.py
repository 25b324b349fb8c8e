#!/usr/bin/env python3
"""Create / calibrate / commit / step / destroy soak in ONE process (run by tests/test_create_soak.py; also usable by hand).

The pattern an intermittent GPU memory-access fault inside oww_commit was seen under: not fresh processes but
MANY engine life cycles with DIFFERENT weights in one process -- the nine weight regimes of tests/test_weight_regimes.py, engine
sizes 1 .. 16,480 streams, one to three heads, with and without the voice-activity network, both kernel families, speech / no /
caller calibration audio; then the same from two host threads at once.  The reference's counterpart: Model objects are constructed
and dropped freely (/root/reference/openwakeword/utils.py:502-536, tests/test_models.py throughout).

    python tests/soak_create.py [--cycles 500] [--threads 2] [--thread-cycles 60] [--seed 1]

OWW_GUARD_ALLOC=1|2 in the environment puts every device buffer of the library into its own mapping with unmapped granules
around it (csrc/owwhip.hip: guard_alloc).  Prints one line per cycle BEFORE the engine is created, so the log names the cycle a
fault happened in; the last line is `soak ok ...`."""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from openwakeword_amd import weights as W  # noqa: E402
from openwakeword_amd._lib import OwwRangeError  # noqa: E402
from openwakeword_amd.engine import StreamEngine, default_calibration_pcm  # noqa: E402
from test_weight_regimes import _heads_for, _regime  # noqa: E402

REGIMES = ["seed1", "seed2", "seed3", "hot", "cold", "conv_1e-3", "negative_bn", "tiny_embedding", "huge_embedding"]
SIZES = [1, 2, 6, 31, 32, 33, 40, 300, 1000, 4096, 16480]
EXTRA_HEADS = ["hey_mycroft", "weather", "timer"]          # binary / multiclass heads next to the regimes' alexa + hey_jarvis


PCM_POOL = (np.random.default_rng(4242).standard_normal((1024, 1280 * 3)) * 3000).astype(np.int16)


def make_sets():
    sets = []
    for name in REGIMES:
        emb, hseed = _regime(name)
        sets.append((name, emb, _heads_for(name, hseed), hseed))
    return sets


MAX_STREAMS = [131072]


def one_cycle(i, rng, sets, speech, log):
    name, emb, heads_all, hseed = sets[i % len(sets)]
    S = 131072 if rng.random() < 0.01 else int(SIZES[int(rng.integers(len(SIZES)))])     # (now and then a 9 GB handle: the allocator's large path)
    S = min(S, MAX_STREAMS[0])
    heads = dict(heads_all)
    if rng.random() < 0.3:
        heads.pop("hey_jarvis")
    if rng.random() < 0.4:
        extra = EXTRA_HEADS[int(rng.integers(len(EXTRA_HEADS)))]
        heads[extra] = W.synthetic_head(extra, hseed)
    fam = 3 if rng.random() < 0.85 else 1
    multiclass = any(h["kind"] == "multiclass" for h in heads.values())
    vad = W.synthetic_vad(hseed) if rng.random() < 0.25 and not multiclass else None
    cal_kind = ("default", None, "own")[int(rng.integers(3))]
    cal = cal_kind
    if cal_kind == "own":
        cal = np.concatenate([speech, speech[:, ::-1]], axis=1)[: int(rng.integers(1, 9))] if speech is not None else None
    kw = dict(vad=vad, vad_threshold=0.5) if vad is not None else {}
    log(f"{i}: regime={name} S={S} heads={list(heads)} family={fam} vad={vad is not None} calibration={cal_kind}")
    t_begin = time.perf_counter()
    try:
        eng = StreamEngine(S, heads, emb, use_mfma=fam, calibration_pcm=cal, **kw)
    except OwwRangeError:
        # a regime the split refuses is served by the exact family (what model.make_engine does)
        log(f"{i}: refused by the fp16-split family -> use_mfma=1")
        eng = StreamEngine(S, heads, emb, use_mfma=1, **kw)
    t_created = time.perf_counter()
    try:
        n = int(rng.integers(1, 4)) if S <= 16480 else 1
        r0, c0 = int(rng.integers(0, PCM_POOL.shape[0] - min(S, 512) + 1)), 1280 * int(rng.integers(0, 4 - n))
        base = PCM_POOL[r0:r0 + min(S, 512), c0:c0 + 1280 * n]      # (a fresh Gaussian batch per cycle would cost more than the engine's life cycle)
        pcm = np.concatenate([base] * ((S + base.shape[0] - 1) // base.shape[0]))[:S]      # (np.tile of a strided view costs 0.3 s here)
        for t in range(n):
            out = eng.step(pcm[:, 1280 * t:1280 * (t + 1)])
        assert np.isfinite(out).all(), f"cycle {i}: non-finite scores"
        assert eng.range_status() is False, f"cycle {i}: range flag raised"
    finally:
        eng.close()
    if i % 50 == 0:
        log(f"{i}: created in {1e3 * (t_created - t_begin):.0f} ms, stepped and destroyed in {1e3 * (time.perf_counter() - t_created):.0f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=500)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--thread-cycles", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-streams", type=int, default=131072, help="cap of the engine sizes (guarded allocations map every buffer on its own: "
                    "GB-sized handles make that mode slow without adding to what it checks)")
    a = ap.parse_args()
    MAX_STREAMS[0] = a.max_streams
    lock = threading.Lock()

    def log(msg):
        with lock:
            print(msg, flush=True)

    sets = make_sets()
    speech = default_calibration_pcm()
    t0 = time.perf_counter()
    rng = np.random.default_rng(a.seed)
    for i in range(a.cycles):
        one_cycle(i, rng, sets, speech, log)
    t1 = time.perf_counter()
    errors = []

    def worker(k):
        r = np.random.default_rng(1000 * a.seed + k)
        try:
            for i in range(a.thread_cycles):
                one_cycle(100000 * (k + 1) + i, r, sets, speech, log)
        except BaseException as e:            # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(a.threads if a.thread_cycles > 0 else 0)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        print("soak FAILED:", errors, flush=True)
        return 1
    t2 = time.perf_counter()
    print(f"soak ok: {a.cycles} cycles in {t1 - t0:.1f} s ({(t1 - t0) / max(a.cycles, 1) * 1e3:.0f} ms each), "
          f"{len(threads)} threads x {a.thread_cycles} cycles in {t2 - t1:.1f} s, guard={os.environ.get('OWW_GUARD_ALLOC', '0')}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
