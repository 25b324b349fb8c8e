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
