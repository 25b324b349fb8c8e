"""Rate conversion to 16 kHz (SURVEY 8f-3 "on-device resample"): the filter design on the CPU, oww_resample against its numpy
restatement on the GPU."""
import numpy as np
import pytest

from openwakeword_amd import resample as R
from openwakeword_amd import weights as W


@pytest.mark.parametrize("rate", [8000, 11025, 22050, 32000, 44100, 48000, 96000])
def test_design_unit_dc_gain_and_geometry(rate):
    p, q, taps = R.design(rate)
    assert p * 16000 == q * rate and np.gcd(p, q) == 1
    assert taps.dtype == np.float32 and taps.shape[0] == q and taps.shape[1] % 2 == 0
    np.testing.assert_allclose(taps.sum(1), 1.0, atol=1e-6)
    x = np.full(4 * p * 50, 1234, np.int16)
    y = R.apply_numpy(x, rate)
    assert y.size == R.n_out_for(x.size, rate) == x.size * q // p
    h = taps.shape[1]
    assert (y[h:-h] == 1234).all()                                          # a constant stays that constant away from the edges


@pytest.mark.parametrize("rate,f_pass,f_stop", [(48000, 3000.0, 9500.0), (44100, 5000.0, 9500.0), (8000, 1000.0, None), (32000, 6000.0, 10000.0)])
def test_numpy_restatement_is_band_limited(rate, f_pass, f_stop):
    n = rate // 5
    t = np.arange(n) / rate
    y = R.apply_numpy((np.sin(2 * np.pi * f_pass * t) * 10000).astype(np.int16), rate).astype(np.float64)
    ref = np.sin(2 * np.pi * f_pass * np.arange(y.size) / 16000.0) * 10000
    assert np.abs(y[100:-100] - ref[100:-100]).max() < 60
    if f_stop:
        z = R.apply_numpy((np.sin(2 * np.pi * f_stop * t) * 10000).astype(np.int16), rate).astype(np.float64)
        assert np.sqrt((z[100:-100] ** 2).mean()) < 40                      # > 45 dB down


def test_matches_scipy_polyphase_in_the_interior():
    from scipy.signal import resample_poly
    rng = np.random.default_rng(2)
    x = np.cumsum(rng.standard_normal(6000)) * 40                           # low-pass-ish signal
    x = np.clip(x, -30000, 30000).astype(np.int16)
    y = R.apply_numpy(x, 48000).astype(np.float64)
    z = resample_poly(x.astype(np.float64), 1, 3)
    assert np.abs(y[50:-50] - z[50:y.size - 50]).max() < 25


@pytest.mark.gpu
@pytest.mark.parametrize("rate,n_in", [(8000, 640), (48000, 3840), (44100, 3528), (22050, 1764), (32000, 2560), (11025, 1000)])
def test_device_resample_equals_numpy(rate, n_in):
    from openwakeword_amd.engine import StreamEngine
    rng = np.random.default_rng(rate)
    S = 37
    eng = StreamEngine(S, {"alexa": W.synthetic_head("alexa", seed=1)}, W.synthetic_embedding(seed=3))
    x = (rng.standard_normal((S, n_in)) * 6000).astype(np.int16)
    x[0] = 32767; x[1] = -32768; x[2] = 0                                   # saturation and silence rows
    got = eng.resample(x, rate)
    want = R.apply_numpy(x, rate)
    assert got.shape == want.shape == (S, n_in * 16000 // rate)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1  # fp32 FMA chain on the device, float64 on the host
    assert (got != want).mean() < 0.02
    eng.close()


@pytest.mark.gpu
def test_predict_batch_at_another_sample_rate():
    from openwakeword_amd.model import BatchedModel
    rng = np.random.default_rng(5)
    heads = {"alexa": W.synthetic_head("alexa", seed=1)}
    wts = {"heads": heads, "embedding": W.synthetic_embedding(seed=3)}
    a, b = BatchedModel(6, ["alexa"], weights=wts), BatchedModel(6, ["alexa"], weights=wts)
    for _ in range(8):
        x8 = (rng.standard_normal((6, 640)) * 5000).astype(np.int16)
        got = a.predict_batch(x8, sample_rate=8000)
        want = b.predict_batch(R.apply_numpy(x8, 8000))
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)            # (a +-1 LSB difference in a few samples is allowed above)
    with pytest.raises(ValueError):
        a.predict_batch(np.zeros((6, 100), np.int16), sample_rate=8000)     # 200 samples at 16 kHz: not a whole chunk
    a.close(); b.close()
