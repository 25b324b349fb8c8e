"""Host-side pieces of the fan-in server (openwakeword_amd/serve.py) that need no GPU: per-message resampling and the per-connection
sample queue that turns arbitrary message sizes into 1280-sample chunks (the remainder path of utils.py:413-430 for many clients)."""
import numpy as np
import pytest

serve = pytest.importorskip("openwakeword_amd.serve")


def test_to_16k_identity_and_lengths():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4800) * 3000).astype(np.int16)
    assert serve.to_16k(x, 16000) is x
    for rate, n_in in ((8000, 640), (32000, 2560), (48000, 3840), (44100, 4410), (22050, 2205)):
        y = serve.to_16k(x[:n_in], rate)
        assert y.dtype == np.int16
        assert abs(y.size - round(n_in * 16000 / rate)) <= 1, (rate, y.size)
    assert serve.to_16k(np.zeros(0, np.int16), 48000).size == 0


def test_to_16k_preserves_a_tone_and_clips():
    t = np.arange(4800) / 48000.0
    tone = (np.sin(2 * np.pi * 440.0 * t) * 12000).astype(np.int16)
    y = serve.to_16k(tone, 48000).astype(np.float64)
    ref = np.sin(2 * np.pi * 440.0 * np.arange(y.size) / 16000.0) * 12000
    assert np.abs(y[40:-40] - ref[40:-40]).max() < 150                    # (filter edge effects at the ends of a stateless message)
    loud = np.full(960, 32767, np.int16)
    assert serve.to_16k(loud, 8000).max() <= 32767                         # overshoot of the interpolation filter is clipped, not wrapped


def test_client_queue_rechunks_any_message_sizes():
    rng = np.random.default_rng(3)
    data = rng.integers(-30000, 30000, size=1280 * 7 + 333, dtype=np.int16)
    c = serve._Client(0, ws=None)
    i = 0
    sizes = [1, 77, 1279, 1280, 1281, 2560, 5000, 13]
    k = 0
    while i < data.size:
        n = sizes[k % len(sizes)]; k += 1
        c.push(data[i:i + n]); i += n
    assert c.n_pending == data.size
    out = np.empty(1280, np.int16)
    got = []
    while c.n_pending >= 1280:
        c.pop_chunk(out)
        got.append(out.copy())
    assert len(got) == 7 and c.n_pending == 333
    np.testing.assert_array_equal(np.concatenate(got), data[:1280 * 7])
    c.push(data[:0])                                                        # empty messages are ignored
    assert c.n_pending == 333


def test_slot_allocator_packs_cohorts_into_blocks():
    """Connections with the same cohort key share 32-stream blocks; a mask of "everybody whose chunk is due in this half of the
    period" then touches about half of the stream groups instead of nearly all of them."""
    rng = np.random.default_rng(5)
    S = 4096
    al = serve.SlotAllocator(S, group=32, distance=serve.cohort_distance)
    arrivals = rng.random(S) * 10.0                                         # connection times over 10 s, 80 ms messages
    keys = [serve.cohort_key(0.080, t) for t in arrivals]
    slots = np.array([al.alloc(k) for k in keys])
    assert sorted(slots.tolist()) == list(range(S)) and al.n_used == S
    with pytest.raises(IndexError):
        al.alloc(keys[0])
    phase = np.array([k[1] for k in keys])
    on = np.zeros(S, bool)
    on[slots[phase < serve.N_PHASE_BINS // 2]] = True                       # one pump round: the first half of the period is due
    groups8 = on.reshape(-1, 8).any(axis=1).mean()
    rnd = np.zeros(S, bool)
    rnd[rng.permutation(S)[: int(on.sum())]] = True
    assert 0.45 < on.mean() < 0.55
    assert groups8 < 0.60 and rnd.reshape(-1, 8).any(axis=1).mean() > 0.95   # packed: ~half the groups; random: nearly all
    # release / reuse: a freed slot goes back to its cohort's block, an emptied block can be claimed by any cohort
    k0 = al.key_of(int(slots[0]))
    al.release(int(slots[0]))
    assert al.alloc(k0) == slots[0]
    with pytest.raises(ValueError):
        al.release(int(slots[0])); al.release(int(slots[0]))
    al2 = serve.SlotAllocator(64, group=32)
    a = [al2.alloc("x") for _ in range(32)]
    b = al2.alloc("y")
    assert a == list(range(32)) and b == 32
    for s_ in a:
        al2.release(s_)
    assert al2.alloc("z") == 0 and al2.key_of(0) == "z"
    # every block taken by other cohorts: the nearest phase of the same period wins
    al3 = serve.SlotAllocator(64, group=32, distance=serve.cohort_distance)
    assert al3.alloc((80, 0)) == 0 and al3.alloc((80, 4)) == 32
    assert al3.alloc((80, 3)) == 33 and al3.alloc((80, 7)) == 1 and al3.alloc((93, 4)) in (2, 34)


def test_cohort_key_bins_by_period_and_phase():
    assert serve.cohort_key(0.080, 0.0) == (80, 0)
    assert serve.cohort_key(0.080, 0.079) == (80, 7)
    assert serve.cohort_key(0.080, 8.0 + 0.045) == (80, 4)                  # the same phase ten periods later
    assert serve.cohort_key(4096 / 48000.0, 0.0)[0] == 85                   # the reference client's 4096-sample buffers at 48 kHz


def test_workers_parent_forwards_signals_and_notices_a_dead_worker():
    """the `--workers N` parent must not leave orphans on SIGTERM and must not sit blocked on worker 0 while another
    worker has died.  `_supervise` with stand-in workers (sleepers): (a) one worker exits with code 3 -> the others are stopped, the
    parent returns 3; (b) SIGTERM to the parent -> every worker is gone when it returns 128 + 15."""
    import os
    import signal
    import subprocess
    import sys
    import threading
    import time
    flag = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"oww_supervise_{os.getpid()}")
    if os.path.exists(flag):
        os.remove(flag)
    # the first worker to start claims the flag file and dies after 0.3 s; the others sleep "forever"
    code = ("import os, sys, time\n"
            f"p = {flag!r}\n"
            "try:\n    fd = os.open(p, os.O_CREAT | os.O_EXCL | os.O_WRONLY); os.close(fd); time.sleep(0.3); sys.exit(3)\n"
            "except FileExistsError:\n    time.sleep(600)\n")
    t0 = time.monotonic()
    assert serve._supervise([sys.executable, "-c", code], 3, poll_s=0.05) == 3
    assert time.monotonic() - t0 < 20
    os.remove(flag)
    # (b) a signal to the parent reaches the children
    pids = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"oww_supervise_pids_{os.getpid()}")
    open(pids, "w").close()
    code = f"import os, time\nopen({pids!r}, 'a').write(str(os.getpid()) + '\\n')\ntime.sleep(600)\n"
    threading.Timer(1.0, lambda: os.kill(os.getpid(), signal.SIGTERM)).start()
    assert serve._supervise([sys.executable, "-c", code], 2, poll_s=0.05) == 128 + signal.SIGTERM
    for pid in [int(x) for x in open(pids).read().split()]:
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)
    os.remove(pids)
