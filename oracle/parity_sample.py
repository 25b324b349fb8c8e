"""
TEST INFRASTRUCTURE ONLY -- the oracle-sampled parity check at BASELINE scale (">= 1,024 random
(stream, step) pairs per config against the oracle").

A fixed set of 64 probe streams x 16 frames (distinct PCM: the reference's three fixture WAVs at several offsets,
silence, LSB noise, full-scale noise and square waves, Gaussian noise from RMS 30 to 12,000, the reference tests'
randint(-1000, 1000)) is run through `oracle.oww_oracle.OracleModel` -- the restatement of
/root/reference/openwakeword/model.py:232-386 + utils.py:409-452 -- on worker processes.  Callers place the probe
streams at random stream ids of a full-size engine (4,096 / 65,536 / 131,072 streams), fill every other stream
with background noise, and compare the engine's scores of the probe rows frame by frame:
    tests/test_parity_scale.py   (-m gpu, the three BASELINE configurations)
    bench.py                     (`parity: {n_pairs, max_abs_err}` next to the throughput line)
Only the checker lives here; nothing in the product path imports it.
"""
from __future__ import annotations

import multiprocessing as mp
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np

CHUNK = 1280
N_PROBE = 64
N_FRAMES = 16
SEED_WEIGHTS = 1234
SEED_INIT_NOISE = 3
PROBE_VERSION = 4           # bump when probe_pcm or the result layout changes (part of the cache file name)
HEADS3 = ("alexa", "hey_mycroft", "hey_jarvis")
HEADS6 = ("alexa", "hey_mycroft", "hey_jarvis", "hey_rhasspy", "timer", "weather")   # the reference's default Model(): model.py:84-87
_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_streaming.npz")


def probe_pcm(n_probe: int = N_PROBE, n_frames: int = N_FRAMES) -> np.ndarray:
    """int16 [n_probe, n_frames * 1280]: deterministic, every row different."""
    n = n_frames * CHUNK
    r = np.random.default_rng(0x5EED)
    z = np.load(_GOLDEN)
    wavs = [z["pcm/" + k] for k in ("alexa_test", "hey_mycroft_test", "hey_jane")]
    rows: List[np.ndarray] = []
    for i in range(n_probe):
        kind = i % 8
        if kind in (0, 1, 2):                       # fixture clips (tiled), a different phase each time
            w = wavs[kind]
            off = (i * 1777) % len(w)
            x = np.resize(np.roll(w, -off), n)
        elif kind == 3:
            amp = (30.0, 300.0, 3000.0, 12000.0)[(i // 8) % 4]
            x = np.clip(np.round(r.normal(0.0, amp, n)), -32768, 32767)
        elif kind == 4:
            x = r.integers(-1000, 1000, n)          # the reference tests' noise (tests/test_models.py:57)
        elif kind == 5:
            x = np.zeros(n) if i == 5 else r.integers(-3, 4, n)                     # silence (once) / LSB noise
        elif kind == 6:
            x = r.integers(-32768, 32768, n)        # full scale noise
        else:
            t = np.arange(n)
            hi, lo = (32767, -32768) if i // 8 < 4 else (12000, -12000)             # full-scale / loud square waves, four periods
            x = np.where((t // (8 << ((i // 8) % 4))) % 2, hi, lo)
        rows.append(np.asarray(x, dtype=np.int16))
    return np.stack(rows)


def _weights(head_names: Sequence[str]):
    from openwakeword_amd import weights as W
    return W.synthetic_embedding(SEED_WEIGHTS), {n: W.synthetic_head(n, SEED_WEIGHTS) for n in head_names}


def init_noise() -> np.ndarray:
    from openwakeword_amd import weights as W
    return W.synthetic_pcm(1, 64000, seed=SEED_INIT_NOISE, rms=600.0)[0]


VAD_THRESHOLD = 0.5        # the gate of BASELINE configs[4] (model.py:366-381) as bench.py / the tests configure it


def labels_of(head_names: Sequence[str]) -> List[str]:
    """Score columns in the engine's order: one per head, n_out per multiclass head (`timer`: 7, class 0 = the negative class included
    -- the columns are compared raw, before any class_mapping drops it: model.py:313-317)."""
    from openwakeword_amd import weights as W
    out: List[str] = []
    for n in head_names:
        n_out = W.HEAD_CATALOGUE[n][3]
        out.extend([n] if n_out == 1 else [f"{n}#{i}" for i in range(n_out)])
    return out


def _class_mapping(heads):
    return {n: {str(i): f"{n}#{i}" for i in range(int(h["n_out"]))} for n, h in heads.items() if int(h["n_out"]) > 1}


def _model(heads, emb, vad: bool):
    from oracle import oww_oracle as O
    if not vad:
        return O.OracleModel(heads, emb, init_noise=init_noise(), class_mapping=_class_mapping(heads))
    from oracle import vad_standin as V
    from openwakeword_amd import weights as W
    return O.OracleModel(heads, emb, init_noise=init_noise(), vad_threshold=VAD_THRESHOLD,
                         vad_session=V.StandinVadSession(W.synthetic_vad(SEED_WEIGHTS)))


def _worker(args) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    rows, head_names, vad = args
    emb, heads = _weights(head_names)
    proto = _model(heads, emb, vad)
    feats0 = np.array(proto.preprocessor.features, dtype=np.float32)
    n_frames = rows[0].shape[0] // CHUNK
    labels = labels_of(head_names)
    out = np.zeros((len(rows), n_frames, len(labels)), np.float64)
    last = np.zeros((len(rows), 16, 96), np.float32)
    gate = np.full((len(rows), n_frames), np.nan)                # max of the VAD ring's [-7:-4] window the gate compared (NaN: empty)
    for i, pcm in enumerate(rows):
        m = proto if i == 0 else _model(heads, emb, vad)
        for t in range(n_frames):
            p = m.predict(pcm[t * CHUNK:(t + 1) * CHUNK])
            out[i, t] = [p[k] for k in labels]
            if vad:
                window = list(m.vad.ring)[-7:-4]
                if window:
                    gate[i, t] = float(np.max(window))
        last[i] = np.asarray(m.preprocessor.features[-16:], dtype=np.float32)
    return out, feats0, last, gate


def effective_cpus() -> int:
    """CPUs this process can actually use: the affinity mask, capped by a cgroup CPU quota when one is set (a container may see 256
    CPUs and own eight of them -- forking 64 workers there is what makes a run crawl)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except Exception:
            continue
    return n


def _run_pool(pcm: np.ndarray, head_names: List[str], workers: int, vad: bool = False) -> Dict[str, np.ndarray]:
    n = pcm.shape[0]
    cores = effective_cpus()
    workers = max(1, min(workers or cores, n, 64))
    parts = [list(range(w, n, workers)) for w in range(workers)]
    jobs = [([pcm[i] for i in idx], head_names, vad) for idx in parts]
    if workers == 1:
        res = [_worker(jobs[0])]
    else:
        with mp.get_context("fork").Pool(workers) as pool:      # forked from a fresh interpreter (see oracle_reference)
            res = pool.map(_worker, jobs)
    scores = np.zeros((n, pcm.shape[1] // CHUNK, len(labels_of(head_names))), np.float64)
    feats = np.zeros((n, 16, 96), np.float32)
    gate = np.full((n, pcm.shape[1] // CHUNK), np.nan)
    for idx, (out, _, last, g) in zip(parts, res):
        scores[idx] = out
        feats[idx] = last
        gate[idx] = g
    return {"scores": scores, "init_features": res[0][1], "features": feats, "heads": np.array(head_names),
            "labels": np.array(labels_of(head_names)), "vad_window_max": gate}


def oracle_reference(n_probe: int = N_PROBE, n_frames: int = N_FRAMES, head_names: Sequence[str] = HEADS3,
                     workers: int = 0, timeout_s: float = 1500.0, vad: bool = False) -> Dict[str, np.ndarray]:
    """Oracle results for `probe_pcm(n_probe, n_frames)`: scores [n_probe, n_frames, n_labels] (float64; `labels`: labels_of), the feature-ring
    seed every stream starts from ([41, 96], oldest first) and each probe's last 16 feature rows.  With `vad` every model carries
    the voice-activity gate of BASELINE configs[4] (model.py:366-381; threshold VAD_THRESHOLD, the stand-in network with the seed-1234
    weights behind the reference's VAD wrapper) and `vad_window_max` [n_probe, n_frames] holds the value each gate decision compared
    with the threshold (NaN while the window is empty), so that a caller can skip decisions within rounding of it.

    Computed by a FRESH interpreter (`python -m oracle.parity_sample`) that fans the streams out over forked workers:
    the caller may already hold a HIP context / BLAS thread pools, neither of which survives a fork reliably, and a
    spawn pool needs an importable __main__.  The result is cached per (n_probe, n_frames, heads) under $TMPDIR."""
    import subprocess
    import sys
    import tempfile
    head_names = list(head_names)
    tag = f"oww_parity_{n_probe}x{n_frames}_{'-'.join(head_names)}_{SEED_WEIGHTS}_{SEED_INIT_NOISE}_v{PROBE_VERSION}{'_vad' if vad else ''}.npz"
    path = os.path.join(tempfile.gettempdir(), tag)
    if not os.path.exists(path):
        root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        tmp = path + f".{os.getpid()}.tmp.npz"
        # one BLAS thread per forked worker: the workers already cover the cores (an unbounded pool per worker oversubscribes the host)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        subprocess.run([sys.executable, "-m", "oracle.parity_sample", tmp, str(n_probe), str(n_frames), str(workers), str(int(vad))] + head_names,
                       cwd=root, check=True, timeout=timeout_s, env=env)
        os.replace(tmp, path)
    z = np.load(path)
    return {k: z[k] for k in z.files}


def probe_stream_ids(n_streams: int, n_probe: int = N_PROBE, seed: int = 7) -> np.ndarray:
    """Random distinct stream ids, always including the first and last stream and both sides of a 128-stream
    workgroup boundary (the heads kernel's tile) and of an 8-stream group boundary (stage E's tile)."""
    r = np.random.default_rng(seed + n_streams)
    fixed = [0, n_streams - 1]
    for b in (8, 128):
        k = int(r.integers(1, max(2, n_streams // b))) * b
        fixed += [min(k - 1, n_streams - 1), min(k, n_streams - 1)]
    fixed = list(dict.fromkeys(fixed))[:n_probe]
    pool = np.setdiff1d(np.arange(n_streams), np.array(fixed))
    rest = r.choice(pool, size=n_probe - len(fixed), replace=False) if n_probe > len(fixed) else np.empty(0, np.int64)
    ids = np.concatenate([np.array(fixed, dtype=np.int64), rest.astype(np.int64)])
    return ids[r.permutation(len(ids))]


if __name__ == "__main__":
    import sys
    out_path, n_probe_, n_frames_, workers_, vad_ = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5]))
    np.savez(out_path, **_run_pool(probe_pcm(n_probe_, n_frames_), list(sys.argv[6:]), workers_, vad_))
