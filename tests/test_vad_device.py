"""Row I / K5: the voice-activity STAND-IN network on the device (openwakeword_amd/csrc/owwhip_vad.h) against its numpy
restatement (oracle/vad_standin.py), driven through the reference's own wrapper logic (oracle.oww_oracle.OracleVad =
vad.py:83-130; OracleModel's gate = model.py:366-381).  Architecture status: stand-in, Silero's graph is unavailable."""
import numpy as np
import pytest

from oracle import oww_oracle as O
from oracle import parity_sample as PS
from oracle import vad_standin as V
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

TOL_VAD = 1e-4
TOL_SCORE = 1e-4


# ------------------------------------------------------------------------------------------------ CPU: the restatement itself
def test_standin_blocks_match_independent_implementations():
    """STFT features against torch.stft, the LSTM cell against torch.nn.LSTM (pins gate order i, f, g, o and the
    (x ; h) row order), the strided convolution against torch.nn.functional.conv1d."""
    import torch
    r = np.random.default_rng(0)
    x = (np.clip(np.round(r.normal(0, 3000, (3, 640))), -32768, 32767) / 32767).astype(np.float32)
    got = V.stft_features(x, np.float64)
    st = torch.stft(torch.from_numpy(x).double(), n_fft=256, hop_length=64, win_length=256,
                    window=torch.from_numpy(V.hann_periodic()), center=False, return_complex=True)      # [B, 129, 7]
    want = np.log1p(50.0 * st.abs().numpy()[:, 1:129].transpose(0, 2, 1))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    w = W.synthetic_vad(7)
    a = r.normal(0, 1, (2, 7, 16))
    for stride in (1, 2):
        cw, cb = w["enc"][1]
        want = torch.nn.functional.conv1d(torch.from_numpy(a).permute(0, 2, 1), torch.from_numpy(cw.astype(np.float64)).permute(2, 1, 0),
                                          torch.from_numpy(cb.astype(np.float64)), stride=stride, padding=1).permute(0, 2, 1).numpy()
        np.testing.assert_allclose(V.conv1d_k3(a, cw.astype(np.float64), cb.astype(np.float64), stride), want, rtol=0, atol=1e-12)
    lw, lb = w["lstm"][0]
    lstm = torch.nn.LSTM(64, 64, num_layers=1).double()
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.from_numpy(lw[:64].T.astype(np.float64)))
        lstm.weight_hh_l0.copy_(torch.from_numpy(lw[64:].T.astype(np.float64)))
        lstm.bias_ih_l0.copy_(torch.from_numpy(lb.astype(np.float64)))
        lstm.bias_hh_l0.zero_()
    xx, h0, c0 = r.normal(0, 1, (1, 5, 64)), r.normal(0, 0.5, (1, 5, 64)), r.normal(0, 0.5, (1, 5, 64))
    _, (h1, c1) = lstm(torch.from_numpy(xx), (torch.from_numpy(h0), torch.from_numpy(c0)))
    gh, gc = V.lstm_cell(xx[0], h0[0], c0[0], lw.astype(np.float64), lb.astype(np.float64))
    np.testing.assert_allclose(gh, h1[0].detach().numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(gc, c1[0].detach().numpy(), rtol=0, atol=1e-12)


def test_standin_session_interface_and_state():
    w = W.synthetic_vad()
    ses = V.StandinVadSession(w)
    x = (W.synthetic_pcm(1, 1280, seed=5)[0] / 32767).astype(np.float32)
    h = np.zeros((2, 1, 64), np.float32)
    out, h1, c1 = ses.run(None, {"input": x[None, :640], "h": h, "c": h, "sr": np.array(16000)})
    assert out.shape == (1, 1) and h1.shape == (2, 1, 64) and 0 < out[0, 0] < 1 and np.abs(h1).max() > 0
    out2, _, _ = ses.run(None, {"input": x[None, 640:], "h": h1, "c": c1, "sr": np.array(16000)})
    out3, _, _ = ses.run(None, {"input": x[None, 640:], "h": h, "c": h, "sr": np.array(16000)})
    assert out2[0, 0] != out3[0, 0]                       # the recurrent state matters
    vad = O.OracleVad(ses)                                # vad.py:129-130: one ring entry per call = mean of the two sub-frames
    vad((x * 32767).round().astype(np.int16))
    np.testing.assert_allclose(vad.ring[-1], (out[0, 0] + out2[0, 0]) / 2, rtol=0, atol=1e-6)
    f32 = V.forward(w, x[None, :640], h, h, np.float32)[0]
    f64 = V.forward(w, x[None, :640], h, h, np.float64)[0]
    assert abs(float(f32[0]) - float(f64[0])) < 1e-5


# ------------------------------------------------------------------------------------------------ GPU
def _probe_rows(n, n_frames):
    return PS.probe_pcm(64, n_frames)[:n]


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams", [37, 300])
def test_device_vad_scores_match_the_oracle(n_streams):
    vw = W.synthetic_vad()
    emb, heads = PS._weights(("alexa",))
    n_frames = 10
    pcm = np.resize(_probe_rows(64, n_frames), (n_streams, n_frames * 1280)).copy()
    pcm[40:] = np.roll(pcm[40:], 333, axis=1) if n_streams > 40 else pcm[40:]
    eng = StreamEngine(n_streams, heads, emb, vad=vw)
    try:
        vads = [O.OracleVad(V.StandinVadSession(vw)) for _ in range(n_streams)]
        worst = 0.0
        for t in range(n_frames):
            x = pcm[:, t * 1280:(t + 1) * 1280]
            eng.step(x)
            got = eng.get_vad()
            for s, v in enumerate(vads):
                v(x[s])
            want = np.array([v.ring[-1] for v in vads])
            worst = max(worst, float(np.abs(got - want).max()))
            np.testing.assert_allclose(got, want, rtol=0, atol=TOL_VAD, err_msg=f"frame {t}")
            if t == 4:
                eng.reset()                                # Model.reset() leaves the VAD alone (model.py:226-230)
        assert np.ptp(want) > 0.2                          # loud and silent probes score differently
        # reset_vad: the listed streams restart from zero state, the others carry on
        eng.reset_vad([1, 5])
        for s in (1, 5):
            vads[s] = O.OracleVad(V.StandinVadSession(vw))
        x = pcm[:, :1280]
        eng.step(x)
        for s, v in enumerate(vads):
            v(x[s])
        np.testing.assert_allclose(eng.get_vad(), np.array([v.ring[-1] for v in vads]), rtol=0, atol=TOL_VAD)
        with pytest.raises(Exception):
            eng.push_vad(np.zeros(n_streams, np.float32))  # the handle computes its own scores
        assert eng.range_status() is False
    finally:
        eng.close()
    print(f"\nVAD stand-in, {n_streams} streams: max |device - oracle| = {worst:.2e}")


@pytest.mark.gpu
def test_fused_vad_gate_matches_oracle_model():
    """The whole of BASELINE configs[4] on a small batch: network + ring + gate inside the step, against OracleModel with
    the stand-in session behind its VAD (model.py:366-381)."""
    vw = W.synthetic_vad()
    emb, heads = PS._weights(PS.HEADS3)
    S, n_frames, thr = 12, 14, 0.5
    pcm = _probe_rows(S, n_frames)
    eng = StreamEngine(S, heads, emb, vad=vw, vad_threshold=thr)
    try:
        models = [O.OracleModel(heads, emb, init_noise=PS.init_noise(), vad_threshold=thr, vad_session=V.StandinVadSession(vw)) for _ in range(S)]
        eng.reset(None, models[0].preprocessor.features[-eng.feature_ring:])
        n_gated = n_open = 0
        for t in range(n_frames):
            x = pcm[:, t * 1280:(t + 1) * 1280]
            got = eng.step(x)
            for s, m in enumerate(models):
                pred = m.predict(x[s])
                want = np.array([pred[k] for k in heads])
                window = list(m.vad.ring)[-7:-4]
                if window and abs(max(window) - thr) < 1e-3:
                    continue                               # a gate decision within rounding of the threshold
                np.testing.assert_allclose(got[s], want, rtol=0, atol=TOL_SCORE, err_msg=f"stream {s} frame {t}")
                if t >= 6:
                    n_gated += int((want == 0).all())
                    n_open += int((want != 0).any())
        assert n_gated > 0 and n_open > 0                  # both branches of the gate were exercised
    finally:
        eng.close()


@pytest.mark.gpu
def test_vad_step_carries_one_chunk():
    emb, heads = PS._weights(("alexa",))
    eng = StreamEngine(2, heads, emb, vad=W.synthetic_vad(), max_chunks=2)
    try:
        with pytest.raises(Exception, match="one 1280-sample chunk"):
            eng.step(np.zeros((2, 2560), np.int16))
        eng.step(np.zeros((2, 1280), np.int16))
    finally:
        eng.close()
