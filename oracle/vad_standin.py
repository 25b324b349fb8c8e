"""
TEST INFRASTRUCTURE ONLY -- numpy restatement of the voice-activity STAND-IN network that libowwhip runs on the device.

Why a stand-in: the reference's VAD is Silero's `silero_vad.onnx` (/root/reference/openwakeword/vad.py:54-81), a release asset
whose graph is described nowhere in the reference checkout; neither the file nor onnxruntime exists here.
What the reference does fix is the network's INTERFACE and the code around it -- 640-sample sub-frames scaled by 1/32767,
recurrent state h, c of shape [2, 1, 64] carried from call to call, one score in (0, 1) per sub-frame, the mean over the
sub-frames of a predict() call appended to a 125-deep ring, the gate on ring[-7:-4] (vad.py:92-130, model.py:366-381).
SURVEY §7 step 9 therefore asks for a STRUCTURAL stand-in of the published Silero family shape with synthetic weights, so that
the fused cost of BASELINE configs[4] can be measured and the state handling / gate can be verified end to end:

    x[640] / 32767
      -> STFT as a strided convolution: 256-sample periodic-Hann frames, hop 64, no padding -> 7 frames x bins 1..128
         (re^2 + im^2) ^ 1/2, compressed as log(1 + 50 |X|)                                         [7, 128]
      -> encoder: four Conv1d(k = 3, zero padding 1) + ReLU over time: 128 -> 16 (stride 1), 16 -> 32 (stride 2),
         32 -> 32 (stride 2), 32 -> 64 (stride 1)                                                   [7,16] [4,32] [2,32] [2,64]
      -> 2-layer LSTM(64), PyTorch gate order i, f, g, o, over the 2 remaining time steps, (h, c) carried
      -> decoder: ReLU -> Linear(64 -> 1) -> sigmoid per time step, mean over the 2 steps = the sub-frame's score

`StandinVadSession` wraps it in the session interface the reference drives (`run(None, {'input','h','c','sr'}) -> [out, h, c]`,
vad.py:121-124), so the reference's own `openwakeword.vad.VAD` class can sit on top of it (tests/golden/make_golden_vad.py) and
`oracle.oww_oracle.OracleVad` / `OracleModel(vad_session=...)` use it unchanged.  Architecture status everywhere: "stand-in,
graph unavailable" -- scores carry no acoustic meaning with random weights.
"""
from __future__ import annotations

import numpy as np

N_FFT = 256
HOP = 64
SUB = 640
N_BINS = 128            # bins 1..128 of the 256-point DFT (DC dropped)
N_FRAMES = (SUB - N_FFT) // HOP + 1      # 7
MAG_GAIN = 50.0
ENC = ((128, 16, 1), (16, 32, 2), (32, 32, 2), (32, 64, 1))     # (cin, cout, stride), kernel 3, padding 1
HID = 64


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float64)


def stft_features(x: np.ndarray, dtype=np.float32) -> np.ndarray:
    """x [B, 640] (already / 32767) -> log(1 + 50 |STFT|) [B, 7, 128]."""
    x = np.asarray(x, dtype=np.float64)
    w = hann_periodic()
    frames = np.stack([x[:, HOP * t: HOP * t + N_FFT] * w for t in range(N_FRAMES)], axis=1)       # [B, 7, 256]
    spec = np.fft.rfft(frames, axis=-1)[..., 1:N_BINS + 1]
    return np.log1p(MAG_GAIN * np.abs(spec)).astype(dtype)


def conv1d_k3(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int) -> np.ndarray:
    """x [B, T, cin], w [3, cin, cout] (tap, in, out), zero padding 1: out[t'] = sum_d w[d] . x[stride t' + d - 1] + b."""
    B, T, _ = x.shape
    xp = np.concatenate([np.zeros_like(x[:, :1]), x, np.zeros_like(x[:, :1])], axis=1)
    To = (T - 1) // stride + 1
    out = np.zeros((B, To, w.shape[2]), dtype=x.dtype)
    for to in range(To):
        c = stride * to
        out[:, to] = xp[:, c] @ w[0] + xp[:, c + 1] @ w[1] + xp[:, c + 2] @ w[2] + b
    return out


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def lstm_cell(x, h, c, w, b):
    """PyTorch LSTM cell, w [128, 256] = rows (x ; h), columns (i | f | g | o), b [256] = b_ih + b_hh."""
    z = np.concatenate([x, h], axis=-1) @ w + b
    i, f, g, o = (z[..., k * HID:(k + 1) * HID] for k in range(4))
    c2 = _sigmoid(f) * c + _sigmoid(i) * np.tanh(g)
    return _sigmoid(o) * np.tanh(c2), c2


def forward(weights: dict, x: np.ndarray, h: np.ndarray, c: np.ndarray, dtype=np.float32):
    """One sub-frame: x [B, 640] float (samples / 32767), h / c [2, B, 64] -> (score [B], h', c')."""
    cast = lambda a: np.asarray(a, dtype=dtype)
    a = stft_features(x, dtype)
    for (w, b), (_, _, stride) in zip(weights["enc"], ENC):
        a = np.maximum(conv1d_k3(a, cast(w), cast(b), stride), 0).astype(dtype)
    h = cast(h).copy()
    c = cast(c).copy()
    wd, bd = weights["dec"]
    ys = []
    for t in range(a.shape[1]):
        inp = a[:, t]
        for layer, (w, b) in enumerate(weights["lstm"]):
            h[layer], c[layer] = lstm_cell(inp, h[layer], c[layer], cast(w), cast(b))
            inp = h[layer]
        ys.append(_sigmoid(np.maximum(inp, 0) @ cast(wd) + dtype(bd)))
    return np.mean(np.stack(ys, axis=0), axis=0).astype(dtype), h, c


class StandinVadSession:
    """The onnxruntime session interface of vad.py:121-124 around `forward`."""

    def __init__(self, weights: dict, dtype=np.float32):
        self.weights, self.dtype = weights, dtype

    def run(self, output_names, feeds):
        x = np.asarray(feeds["input"], dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == SUB, "the stand-in takes 640-sample sub-frames (VAD.__call__, vad.py:129)"
        assert int(feeds["sr"]) == 16000
        y, h, c = forward(self.weights, x, feeds["h"], feeds["c"], self.dtype)
        return [y.astype(np.float32)[:, None], h.astype(np.float32), c.astype(np.float32)]
