"""Several handles alive in one process, stepped from different host threads at the same time (the reference's counterpart: one
openwakeword.Model per worker thread, model.py:60-197 keeps no module-level state).  Every handle owns its streams, device buffers
and error flag; the library keeps no process-wide device state -- so a handle's scores must not depend on what the others do."""
import threading

import numpy as np
import pytest

from openwakeword_amd import weights as W

pytestmark = pytest.mark.gpu

CASES = [  # (streams, heads, embedding seed, use_mfma, with voice-activity gate)
    (300, ["alexa", "hey_jarvis"], 3, 3, False),
    (4096, ["hey_mycroft"], 4, 3, False),
    (40, ["alexa", "weather", "timer"], 5, 1, False),
    (72, ["hey_jarvis"], 6, 3, True),
]
T = 24


def _make(case):
    from openwakeword_amd.engine import StreamEngine
    S, heads, seed, fam, vad = case
    hs = {n: W.synthetic_head(n, seed=seed + i) for i, n in enumerate(heads)}
    kw = dict(vad=W.synthetic_vad(seed=seed), vad_threshold=0.5) if vad else {}
    return StreamEngine(S, hs, W.synthetic_embedding(seed=seed), use_mfma=fam, **kw)


def _run(eng, pcm, out, errors, barrier=None):
    try:
        if barrier is not None:
            barrier.wait()
        for t in range(T):
            out[t] = eng.step(pcm[t])
    except Exception as e:               # noqa: BLE001 -- reported by the main thread
        errors.append(e)


def test_handles_stepped_from_concurrent_threads_match_their_solo_runs():
    rng = np.random.default_rng(17)
    pcms = [(rng.standard_normal((T, c[0], 1280)) * 3000).astype(np.int16) for c in CASES]
    solo = []
    for c, pcm in zip(CASES, pcms):      # each handle alone in the process
        eng = _make(c)
        out = np.empty((T, c[0], eng.n_labels), np.float32)
        errs = []
        _run(eng, pcm, out, errs)
        eng.close()
        assert not errs, errs
        solo.append(out)
    engines = [_make(c) for c in CASES]  # all alive together, one thread each, released at the same moment; three rounds
    try:
        for rnd in range(3):
            outs = [np.empty_like(s) for s in solo]
            errors = []
            barrier = threading.Barrier(len(CASES))
            threads = [threading.Thread(target=_run, args=(e, p, o, errors, barrier)) for e, p, o in zip(engines, pcms, outs)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            assert not errors, errors
            for c, got, want in zip(CASES, outs, solo):
                np.testing.assert_array_equal(got, want, err_msg=f"round {rnd}, handle {c}")
            for e, c in zip(engines, CASES):
                e.reset()
                if c[4]:
                    e.reset_vad()
    finally:
        for e in engines:
            e.close()


def test_error_text_is_per_thread():
    """oww_last_error is thread-local: a failing call in one thread does not overwrite what another thread reads."""
    from openwakeword_amd import _lib
    from openwakeword_amd.engine import StreamEngine
    eng = StreamEngine(4, {"alexa": W.synthetic_head("alexa", 1)}, W.synthetic_embedding(1))
    seen = {}

    def bad():
        try:
            eng.step(np.zeros((4, 1280 * 4097), np.int16))    # more chunks per call than the ABI takes (OWW_MAX_CALL_CHUNKS = 4096)
        except Exception as e:           # noqa: BLE001
            seen["bad"] = str(e)

    try:
        th = threading.Thread(target=bad)
        th.start()
        th.join()
        assert "bad" in seen and ": " in seen["bad"]
        text = seen["bad"].split(": ", 1)[1]
        assert text and text not in (_lib.load().oww_last_error() or b"").decode(errors="replace")
        out = eng.step(np.zeros((4, 1280), np.int16))         # and the handle is still usable
        assert np.isfinite(out).all()
    finally:
        eng.close()
