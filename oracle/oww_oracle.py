"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the openWakeWord streaming hot path.

This file is a from-scratch numpy restatement of the arithmetic that the reference
hands to onnxruntime / LiteRT plus the streaming control flow around it.  It exists
so that the HIP kernels in ``openwakeword_amd/csrc`` can be checked against an
independent implementation.  Nothing in the product package imports it: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may.

PARITY STATUS: the reference's own arithmetic lives in
un-vendored third-party code (onnxruntime>=1.10,<2 executing melspectrogram.onnx /
embedding_model.onnx / <wakeword>.onnx, release assets v0.5.1) that is absent from
/root/reference and from this machine, so the *stage math* below (mel, CNN, heads) is
"parity unpinned" against the RELEASED files -- it restates the published recipes cited
per function.  What IS pinned:
  * the streaming / control logic: the reference's own ``openwakeword.utils.AudioFeatures``
    and ``openwakeword.model.Model`` are executed (tests/golden/make_golden.py) with this
    file's stage math plugged in at the reference's inference-backend seam;
  * stage math + control logic together on real model FILES: the reference's own Model runs
    on heads / embedding / melspectrogram graphs written by PyTorch's ONNX exporter (the
    tool the reference exports with) over a generic ONNX evaluator (oracle/mini_ort.py),
    with nothing of this file in the loop (tests/golden/make_golden_onnx.py);
  * the stage math against library kernels in float64 (torch.stft, torch.nn).
``OracleAudioFeatures`` / ``OracleModel`` must reproduce all of these
(tests/test_oracle_golden.py).

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from collections import deque
from typing import Callable, Dict, Optional

import numpy as np

# --------------------------------------------------------------------------------------
# Constants of the front end (notebooks/converting_google_speech_embedding_model.ipynb
# cell 15: n_fft=512, hop=160, win=400, n_mels=32, fmin=60, fmax=3800, sr=16000)
# --------------------------------------------------------------------------------------
SR = 16000
N_FFT = 512
HOP = 160
WIN = 400
N_MELS = 32
FMIN = 60.0
FMAX = 3800.0
N_BINS = N_FFT // 2 + 1
TOP_DB = 80.0
AMIN = 1e-10
CHUNK = 1280            # one 80 ms step (openwakeword/utils.py:417-434)
MEL_WINDOW = 76         # rows per embedding window (utils.py:225,440)
MEL_HOP = 8             # window hop in mel rows (utils.py:229,438)
EMB_DIM = 96


# --------------------------------------------------------------------------------------
# Row C of SURVEY §8a: melspectrogram.onnx  (torchlibrosa Spectrogram + LogmelFilterBank)
# --------------------------------------------------------------------------------------
def hann_window_padded(dtype=np.float64) -> np.ndarray:
    """Periodic Hann(400) centred in a 512 frame (56 zeros either side).

    torchlibrosa.Spectrogram builds its conv kernels from
    ``librosa.filters.get_window('hann', win_length, fftbins=True)`` followed by
    ``librosa.util.pad_center(.., n_fft)`` [3P]; SURVEY Appendix A step 2.
    """
    w = np.zeros(N_FFT, dtype=np.float64)
    lead = (N_FFT - WIN) // 2
    for n in range(WIN):
        w[lead + n] = 0.5 - 0.5 * math.cos(2.0 * math.pi * n / WIN)
    return w.astype(dtype)


def dft_kernels(dtype=np.float32):
    """Windowed DFT matrices (real, imag), each [512, 257].

    torchlibrosa expresses the STFT as two Conv1d layers whose weights are
    real/imag parts of ``exp(-2*pi*i*n*k/N)`` times the window, stored as float32 [3P].
    """
    n = np.arange(N_FFT, dtype=np.float64)[:, None]
    k = np.arange(N_BINS, dtype=np.float64)[None, :]
    ang = -2.0 * np.pi * n * k / N_FFT
    w = hann_window_padded(np.float64)[:, None]
    return (np.cos(ang) * w).astype(dtype), (np.sin(ang) * w).astype(dtype)


def _hz_to_mel_slaney(f: float) -> float:
    f_sp = 200.0 / 3.0
    if f < 1000.0:
        return f / f_sp
    return 1000.0 / f_sp + math.log(f / 1000.0) / (math.log(6.4) / 27.0)


def _mel_to_hz_slaney(m: float) -> float:
    f_sp = 200.0 / 3.0
    m0 = 1000.0 / f_sp
    if m < m0:
        return m * f_sp
    return 1000.0 * math.exp((math.log(6.4) / 27.0) * (m - m0))


def mel_filterbank(dtype=np.float32) -> np.ndarray:
    """librosa.filters.mel(sr=16000, n_fft=512, n_mels=32, fmin=60, fmax=3800) transposed
    to [257, 32] (Slaney scale, slaney area normalisation; librosa defaults htk=False,
    norm='slaney' [3P]).  Written as explicit loops on purpose:
    the product builds the same table vectorised (openwakeword_amd/weights.py)."""
    lo, hi = _hz_to_mel_slaney(FMIN), _hz_to_mel_slaney(FMAX)
    edges = [_mel_to_hz_slaney(lo + (hi - lo) * i / (N_MELS + 1)) for i in range(N_MELS + 2)]
    fb = np.zeros((N_BINS, N_MELS), dtype=np.float64)
    for m in range(N_MELS):
        f0, f1, f2 = edges[m], edges[m + 1], edges[m + 2]
        norm = 2.0 / (f2 - f0)
        for k in range(N_BINS):
            f = k * (SR / 2.0) / (N_BINS - 1)
            up = (f - f0) / (f1 - f0)
            down = (f2 - f) / (f2 - f1)
            fb[k, m] = max(0.0, min(up, down)) * norm
    return fb.astype(dtype)


class MelTables:
    def __init__(self, dtype=np.float32):
        self.dtype = dtype
        self.re, self.im = dft_kernels(dtype)
        self.fb = mel_filterbank(dtype)


_MEL_CACHE: Dict[str, MelTables] = {}


def _tables(dtype) -> MelTables:
    key = np.dtype(dtype).name
    if key not in _MEL_CACHE:
        _MEL_CACHE[key] = MelTables(dtype)
    return _MEL_CACHE[key]


def n_mel_frames(n_samples: int) -> int:
    """center=False framing: floor((N-512)/160)+1 (ipynb cell 15; utils.py:270)."""
    if n_samples < N_FFT:
        return 0
    return (n_samples - N_FFT) // HOP + 1


def mel_stage(x: np.ndarray, dtype=np.float32) -> np.ndarray:
    """What ``melspec_model_predict`` returns[0] (utils.py:87,202): f32[B,N] -> [B,1,F,32] dB.

    The clamp floor ``max - 80 dB`` is taken over the WHOLE call (all batch items, frames
    and bins), exactly as the monkey-patched power_to_db does with ``log_spec.max()``
    (ipynb cell 15).  Input is int16-valued float with no +-1 scaling (utils.py:199).
    """
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[None, :]
    t = _tables(dtype)
    B, N = x.shape
    F = n_mel_frames(N)
    idx = (np.arange(F) * HOP)[:, None] + np.arange(N_FFT)[None, :]
    frames = x.astype(dtype)[:, idx]                       # [B,F,512]
    re = frames @ t.re
    im = frames @ t.im
    power = re * re + im * im                              # [B,F,257]
    mel = power @ t.fb                                     # [B,F,32]
    ten = np.asarray(10.0, dtype=dtype)
    db = ten * np.log(np.maximum(mel, np.asarray(AMIN, dtype=dtype))) / np.log(ten)
    if db.size:
        db = np.maximum(db, db.max() - np.asarray(TOP_DB, dtype=dtype))
    return db.astype(dtype)[:, None, :, :]


def mel_transform(spec: np.ndarray) -> np.ndarray:
    """Host-side ``x/10 + 2`` of utils.py:180,206."""
    return spec / 10 + 2


# --------------------------------------------------------------------------------------
# Row E: embedding_model.onnx -- Google speech_embedding CNN
# (notebooks/converting_google_speech_embedding_model.ipynb cell 18)
# --------------------------------------------------------------------------------------
# (kh, kw, cin, cout, relu_before_bn, bn_and_act, pool(time,freq) or None)
CNN_LAYERS = [
    (3, 3, 1, 24, True, True, None),
    (1, 3, 24, 24, False, True, None), (3, 1, 24, 24, False, True, (2, 2)),
    (1, 3, 24, 48, False, True, None), (3, 1, 48, 48, False, True, None),
    (1, 3, 48, 48, False, True, None), (3, 1, 48, 48, False, True, (1, 2)),
    (1, 3, 48, 72, False, True, None), (3, 1, 72, 72, False, True, None),
    (1, 3, 72, 72, False, True, None), (3, 1, 72, 72, False, True, (2, 2)),
    (1, 3, 72, 96, False, True, None), (3, 1, 96, 96, False, True, None),
    (1, 3, 96, 96, False, True, None), (3, 1, 96, 96, False, True, (1, 2)),
    (1, 3, 96, 96, False, True, None), (3, 1, 96, 96, False, True, None),
    (1, 3, 96, 96, False, True, None), (3, 1, 96, 96, False, True, (2, 2)),
    (3, 1, 96, 96, False, False, None),
]
BN_EPS = 1e-3            # tf.keras.layers.BatchNormalization default epsilon [3P]
LEAK = np.float32(0.20000000298023224)
FLOOR = np.float32(-0.4000000059604645)


def cnn_param_count() -> int:
    n = 0
    for kh, kw, ci, co, _, bn, _ in CNN_LAYERS:
        n += kh * kw * ci * co + (4 * co if bn else 0)
    return n


def _activation(x):
    """max(max(0.2*x, x), -0.4)  (ipynb cell 18: MyLeakyReLU(alpha=0.4) -> alpha*x/2, then tf.maximum)."""
    dt = x.dtype
    return np.maximum(np.maximum(x * dt.type(LEAK), x), dt.type(FLOOR))


def _conv(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """x[B,T,F,Cin], w[kh,kw,Cin,Cout] (Keras HWIO).  Time axis 'valid'; mel axis zero-padded
    by (kw-1)/2 either side (ZeroPadding2D((0,1)) for layer 0, padding='same' for 1x3)."""
    kh, kw, ci, co = w.shape
    B, T, F, _ = x.shape
    pf = (kw - 1) // 2
    if pf:
        x = np.pad(x, ((0, 0), (0, 0), (pf, pf), (0, 0)))
    To = T - kh + 1
    out = np.zeros((B, To, F, co), dtype=x.dtype)
    for dt in range(kh):
        for df in range(kw):
            out += x[:, dt:dt + To, df:df + F, :] @ w[dt, df]
    return out


def _pool(x: np.ndarray, pt: int, pf: int) -> np.ndarray:
    B, T, F, C = x.shape
    T2, F2 = T // pt, F // pf
    x = x[:, :T2 * pt, :F2 * pf, :].reshape(B, T2, pt, F2, pf, C)
    return x.max(axis=(2, 4))


def bn_fold(gamma, beta, mean, var, dtype=np.float32):
    """Inference BatchNorm as y = x*scale + shift (computed in float64, stored in `dtype`)."""
    g, b, m, v = (np.asarray(a, dtype=np.float64) for a in (gamma, beta, mean, var))
    scale = g / np.sqrt(v + BN_EPS)
    shift = b - m * scale
    return scale.astype(dtype), shift.astype(dtype)


def embedding_stage(x: np.ndarray, emb: dict, dtype=np.float32, return_layers: bool = False):
    """What ``embedding_model_predict`` computes before the squeeze (utils.py:93):
    f32[B,76,32,1] -> [B,1,1,96].  `emb` = {"conv": [20 x HWIO], "bn": [19 x (gamma,beta,mean,var)]}.
    Also works fully-convolutionally on taller inputs (76+8K rows -> K+1 outputs)."""
    h = np.asarray(x, dtype=dtype)
    if h.ndim == 3:
        h = h[..., None]
    layers = []
    for li, (kh, kw, ci, co, relu_first, bn, pool) in enumerate(CNN_LAYERS):
        h = _conv(h, emb["conv"][li].astype(dtype))
        if relu_first:
            h = np.maximum(h, dtype(0))
        if bn:
            scale, shift = bn_fold(*emb["bn"][li], dtype=dtype)
            h = _activation(h * scale + shift)
        if pool:
            h = _pool(h, *pool)
        if return_layers:
            layers.append(h)
    return (h, layers) if return_layers else h


# --------------------------------------------------------------------------------------
# Row G: <wakeword>.onnx -- classifier heads (openwakeword/train.py:56-83, docs/models/*.md)
# --------------------------------------------------------------------------------------
LN_EPS = 1e-5            # torch.nn.LayerNorm default [3P]


def _layernorm(x, g, b):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + x.dtype.type(LN_EPS)) * g + b


def net_blocks(net):
    """Hidden blocks of a head network as (w, b, ln): 'w2' / 'b2' / 'ln2' (None = no hidden block), then net['more'] = [{'w', 'b',
    'ln'}, ...] for train.py:73's n_blocks > 1."""
    if net.get("w2") is None:
        return []
    return [(net["w2"], net["b2"], net.get("ln2"))] + [(m["w"], m["b"], m.get("ln")) for m in net.get("more") or []]


def _mlp(x, net, dtype):
    """Flatten -> Linear -> [LN] -> ReLU -> n_blocks x (Linear -> [LN] -> ReLU) -> Linear  (train.py:56-83; n_blocks = 1 in the
    released models)."""
    h = x.reshape(x.shape[0], -1).astype(dtype)
    h = h @ net["w1"].astype(dtype) + net["b1"].astype(dtype)
    if net.get("ln1") is not None:
        h = _layernorm(h, net["ln1"][0].astype(dtype), net["ln1"][1].astype(dtype))
    h = np.maximum(h, dtype(0))
    for w, b, ln in net_blocks(net):
        h = h @ w.astype(dtype) + b.astype(dtype)
        if ln is not None:
            h = _layernorm(h, ln[0].astype(dtype), ln[1].astype(dtype))
        h = np.maximum(h, dtype(0))
    return h @ net["w3"].astype(dtype) + net["b3"].astype(dtype)


def head_stage(x: np.ndarray, head: dict, dtype=np.float32) -> np.ndarray:
    """What ``model_prediction_function[name]`` returns[0] (model.py:137-138): f32[B,T,96] -> [B,n_out].

    head["kind"]:
      "binary"  sigmoid(MLP)                                   (docs/models/alexa.md:9-24)
      "gated"   two binary nets; the second replaces the first's score where the first
                is > 0.5 (docs/models/hey_jarvis.md:9,38 -- routing formula itself is [3P])
      "multiclass"  softmax(relu(MLP))                         (train.py:152-165, docs/models/timers.md)
    """
    x = np.asarray(x, dtype=dtype)
    if x.ndim == 2:
        x = x[None]
    kind = head["kind"]
    if kind == "rnn":
        return _rnn_head(x, head, dtype)
    if kind == "multiclass":
        z = np.maximum(_mlp(x, head["net"], dtype), dtype(0))
        z = z - z.max(axis=1, keepdims=True)
        e = np.exp(z)
        return (e / e.sum(axis=1, keepdims=True)).astype(dtype)
    s = 1.0 / (1.0 + np.exp(-_mlp(x, head["net"], dtype)))
    if kind == "gated":
        s2 = 1.0 / (1.0 + np.exp(-_mlp(x, head["net2"], dtype)))
        s = np.where(s > 0.5, s2, s)
    return s.astype(dtype)


def _lstm_direction(x: np.ndarray, w: np.ndarray, b: np.ndarray, reverse: bool, dtype) -> np.ndarray:
    """One direction of one torch.nn.LSTM layer from a zero state: x [B, T, in] -> h [B, T, H]; w [in + H, 4H] rows (x ; h), columns
    (i | f | g | o), b = b_ih + b_hh.  (torch's equations: i, f, o = sigmoid, g = tanh; c' = f c + i g; h' = o tanh(c').)"""
    B, T, _ = x.shape
    H = w.shape[1] // 4
    w, b = w.astype(dtype), b.astype(dtype)
    h, c = np.zeros((B, H), dtype), np.zeros((B, H), dtype)
    out = np.zeros((B, T, H), dtype)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))        # noqa: E731
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        z = np.concatenate([x[:, t], h], axis=1) @ w + b
        i, f, g, o = sig(z[:, :H]), sig(z[:, H:2 * H]), np.tanh(z[:, 2 * H:3 * H]), sig(z[:, 3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        out[:, t] = h
    return out.astype(dtype)


def _rnn_head(x: np.ndarray, head: dict, dtype) -> np.ndarray:
    """train.py:85-98, model_type "rnn": `out, h = LSTM(96, 64, num_layers=2, bidirectional=True)(x)`, then
    `Sigmoid | ReLU (Linear(128, n_classes)(out[:, -1]))`; multiclass files carry the softmax train.py:152-165 wraps around the model."""
    a = x.astype(dtype)
    for layer in head["lstm"]:
        a = np.concatenate([_lstm_direction(a, layer[0][0], layer[0][1], False, dtype),
                            _lstm_direction(a, layer[1][0], layer[1][1], True, dtype)], axis=2)
    z = a[:, -1] @ head["w_out"].astype(dtype) + head["b_out"].astype(dtype)
    if int(head["n_out"]) == 1:
        return (1.0 / (1.0 + np.exp(-z))).astype(dtype)
    z = np.maximum(z, dtype(0))
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True)).astype(dtype)


# --------------------------------------------------------------------------------------
# Rows B, D, F: AudioFeatures streaming state machine (openwakeword/utils.py:163-178, 387-463)
# --------------------------------------------------------------------------------------
class OracleAudioFeatures:
    """Restatement of the reference's streaming front end with pluggable stage math.

    mel_fn(x f32[1,N]) -> [1,1,F,32] dB and embed_fn(x f32[B,76,32,1]) -> [B,96] are the two
    seams of utils.py:87,93; by default they are this file's numpy stages.
    """

    RAW_KEEP = 10 * SR          # utils.py:164
    MEL_KEEP = 10 * 97          # utils.py:166
    FEAT_KEEP = 120             # utils.py:170

    def __init__(self, emb: dict, mel_fn: Optional[Callable] = None,
                 embed_fn: Optional[Callable] = None, dtype=np.float32,
                 init_noise: Optional[np.ndarray] = None):
        self.dtype = dtype
        self.mel_fn = mel_fn or (lambda x: mel_stage(x, dtype))
        self.embed_fn = embed_fn or (lambda x: embedding_stage(x, emb, dtype).reshape(x.shape[0], -1).squeeze())
        self._init_noise = init_noise
        self.reset()

    # utils.py:172-178 (and the identical tail of __init__, 163-170)
    def reset(self):
        self.raw = np.zeros(0, dtype=np.int64)
        self.mel_rows = np.ones((MEL_WINDOW, N_MELS))
        self.pending = 0                               # "accumulated_samples"
        self.carry = np.empty(0)                       # "raw_data_remainder"
        noise = self._init_noise
        if noise is None:
            noise = np.random.randint(-1000, 1000, SR * 4).astype(np.int16)
        self.features = self.clip_embeddings(noise)

    # utils.py:180-208
    def melspectrogram(self, x) -> np.ndarray:
        if isinstance(x, list):
            x = np.array(x).astype(np.int16)
        if x.dtype != np.int16:
            raise ValueError("Input data must be 16-bit integers (i.e., 16-bit PCM audio)."
                             f"You provided {x.dtype} data.")
        if x.ndim < 2:
            x = x[None, :]
        spec = np.squeeze(self.mel_fn(x.astype(np.float32)))
        return mel_transform(spec)

    # utils.py:225-236
    def clip_embeddings(self, x: np.ndarray) -> np.ndarray:
        spec = self.melspectrogram(x)
        wins = [spec[i:i + MEL_WINDOW] for i in range(0, spec.shape[0], MEL_HOP)
                if spec[i:i + MEL_WINDOW].shape[0] == MEL_WINDOW]
        batch = np.asarray(wins)[..., None].astype(np.float32)
        return self.embed_fn(batch)

    def _append_raw(self, x):
        self.raw = np.concatenate((self.raw, np.asarray(x).astype(np.int64)))[-self.RAW_KEEP:]

    # utils.py:409-452
    def __call__(self, x: np.ndarray) -> int:
        done = 0
        if self.carry.shape[0]:
            x = np.concatenate((self.carry, x))
            self.carry = np.empty(0)
        total = self.pending + x.shape[0]
        if total >= CHUNK:
            extra = total % CHUNK
            if extra:
                self._append_raw(x[:-extra])
                self.pending += x.shape[0] - extra
                self.carry = x[-extra:]
            else:
                self._append_raw(x)
                self.pending += x.shape[0]
        else:
            self.pending += x.shape[0]
            self._append_raw(x)

        if self.pending >= CHUNK and self.pending % CHUNK == 0:
            if self.raw.shape[0] < 400:                       # utils.py:393-394
                raise ValueError("The number of input frames must be at least 400 samples @ 16khz (25 ms)!")
            tail = self.raw[-(self.pending + 3 * HOP):]       # utils.py:397
            new_rows = self.melspectrogram(tail.astype(np.int16))
            self.mel_rows = np.vstack((self.mel_rows, new_rows))[-self.MEL_KEEP:]
            for back in range(self.pending // CHUNK - 1, -1, -1):   # utils.py:437-443
                end = self.mel_rows.shape[0] - MEL_HOP * back
                win = self.mel_rows[end - MEL_WINDOW:end]
                if win.shape[0] == MEL_WINDOW:
                    e = self.embed_fn(win.astype(np.float32)[None, :, :, None])
                    self.features = np.vstack((self.features, e))
            done = self.pending
            self.pending = 0
        self.features = self.features[-self.FEAT_KEEP:]
        return done if done else self.pending

    # utils.py:454-460
    def get_features(self, n_feature_frames: int = 16, start_ndx: int = -1) -> np.ndarray:
        n = int(n_feature_frames)
        if start_ndx != -1:
            stop = start_ndx + n
            rows = self.features[start_ndx:] if stop == 0 else self.features[start_ndx:stop]
        else:
            rows = self.features[-n:]
        return rows[None].astype(np.float32)


# --------------------------------------------------------------------------------------
# Rows A, H: Model.predict / predict_clip / reset (openwakeword/model.py:226-426), without
# Speex and custom verifiers; row I: the VAD gate around a pluggable network
# --------------------------------------------------------------------------------------
class OracleVad:
    """openwakeword/vad.py:83-130 around any `session.run(None, {'input','h','c','sr'}) -> [out, h, c]`."""

    def __init__(self, session):
        self.session = session
        self.ring: deque = deque(maxlen=125)                       # vad.py:84
        self.h = np.zeros((2, 1, 64), np.float32)                  # vad.py:92-96
        self.c = np.zeros((2, 1, 64), np.float32)

    def __call__(self, x):                                         # vad.py:129-130: 640-sample sub-frames
        outs = []
        for i in range(0, x.shape[0], 640):                        # vad.py:115-125
            piece = (x[i:i + 640] / 32767).astype(np.float32)[None]
            out, self.h, self.c = self.session.run(None, {"input": piece, "h": self.h, "c": self.c,
                                                          "sr": np.array(16000).astype(np.int64)})
            outs.append(out[0][0])
        self.ring.append(np.mean(outs))                            # vad.py:127,130


class OracleModel:
    def __init__(self, heads: Dict[str, dict], emb: dict, dtype=np.float32,
                 class_mapping: Optional[Dict[str, Dict[str, str]]] = None,
                 head_fns: Optional[Dict[str, Callable]] = None, vad_threshold: float = 0.0, vad_session=None,
                 **feature_kwargs):
        self.vad_threshold = vad_threshold
        self.vad = OracleVad(vad_session) if vad_threshold > 0 else None      # model.py:208-210; survives reset()
        self.heads = heads
        self.dtype = dtype
        self.model_inputs = {k: int(h["T"]) for k, h in heads.items()}
        self.model_outputs = {k: int(h["n_out"]) for k, h in heads.items()}
        self.class_mapping = {}
        for k in heads:
            if class_mapping and k in class_mapping:
                self.class_mapping[k] = class_mapping[k]
            else:
                self.class_mapping[k] = {str(i): str(i) for i in range(self.model_outputs[k])}
        # same return structure as ort.InferenceSession.run: a list holding one [1,n_out] array
        self.head_fns = head_fns or {k: (lambda x, _h=h: [head_stage(x, _h, dtype)])
                                     for k, h in heads.items()}
        self.prediction_buffer: Dict[str, deque] = {}
        self.preprocessor = OracleAudioFeatures(emb, dtype=dtype, **feature_kwargs)

    def _ring(self, label):
        if label not in self.prediction_buffer:
            self.prediction_buffer[label] = deque(maxlen=30)      # model.py:198
        return self.prediction_buffer[label]

    def parent_of(self, label):                                    # model.py:215-224
        parent = ""
        for mdl, mapping in self.class_mapping.items():
            if label in mapping.values():
                parent = mdl
            elif label in self.class_mapping and label == mdl:
                parent = mdl
        return parent

    def reset(self):                                               # model.py:226-230
        self.prediction_buffer = {}
        self.preprocessor.reset()

    def predict(self, x, patience: dict = {}, threshold: dict = {}, debounce_time: float = 0.0):
        if not isinstance(x, np.ndarray):                          # model.py:262-263
            raise ValueError("The input audio data (x) must by a Numpy array, instead received "
                             f"an object of type {type(x)}.")
        n_ready = self.preprocessor(x)
        out = {}
        for name in self.heads:
            T = self.model_inputs[name]
            if n_ready > CHUNK:                                    # model.py:287-298
                group = []
                for back in range(n_ready // CHUNK - 1, -1, -1):
                    group.extend(self.head_fns[name](self.preprocessor.get_features(T, start_ndx=-T - back)))
                pred = np.array(group).max(axis=0)[None]
            elif n_ready == CHUNK:                                 # model.py:299-302
                pred = self.head_fns[name](self.preprocessor.get_features(T))
            else:                                                  # model.py:303-311
                if self.model_outputs[name] == 1:
                    ring = self._ring(name)
                    pred = [[[ring[-1] if len(ring) else 0]]]
                else:
                    n_cls = max(int(i) for i in self.class_mapping[name])
                    pred = [[[0] * (n_cls + 1)]]
            if self.model_outputs[name] == 1:                      # model.py:313-317
                out[name] = pred[0][0][0]
            else:
                for idx, cls in self.class_mapping[name].items():
                    out[cls] = pred[0][0][int(idx)]
            for cls in out:                                        # model.py:331-333
                if len(self._ring(cls)) < 5:
                    out[cls] = 0.0

        if patience != {} or debounce_time > 0:                    # model.py:340-359
            if threshold == {}:
                raise ValueError("Error! When using the `patience` argument, threshold "
                                 "values must be provided via the `threshold` argument!")
            if patience != {} and debounce_time > 0:
                raise ValueError("Error! The `patience` and `debounce_time` arguments cannot be used together!")
            for label in out:
                parent = self.parent_of(label)
                if out[label] != 0.0:
                    if parent in patience:
                        hist = np.array(self._ring(label))[-patience[parent]:]
                        if (hist >= threshold[parent]).sum() < patience[parent]:
                            out[label] = 0.0
                    elif debounce_time > 0 and parent in threshold:
                        n = int(np.ceil(debounce_time / (n_ready / 16000)))
                        hist = np.array(self._ring(label))[-n:]
                        if out[label] >= threshold[parent] and (hist >= threshold[parent]).sum() > 0:
                            out[label] = 0.0
        for label in out:                                          # model.py:362-363
            self._ring(label).append(out[label])
        if self.vad_threshold > 0:                                 # model.py:366-381, on the raw x, after the ring append
            self.vad(x)
            window = list(self.vad.ring)[-7:-4]                    # frames 0.4 .. 0.56 s before this one
            if (np.max(window) if len(window) > 0 else 0) < self.vad_threshold:
                for label in out:
                    out[label] = 0.0
        return out

    def predict_clip(self, clip: np.ndarray, padding: int = 1, chunk_size: int = 1280, **kw):
        data = clip                                                # model.py:408-418
        if padding:
            z = np.zeros(16000 * padding).astype(np.int16)
            data = np.concatenate((z, data, z))
        return [self.predict(data[i:i + chunk_size], **kw)         # model.py:421-426
                for i in range(0, data.shape[0] - chunk_size, chunk_size)]
