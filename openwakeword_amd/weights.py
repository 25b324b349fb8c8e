"""
Weight sets for the HIP streaming path: the three graph families the reference delegates to
onnxruntime (melspectrogram / embedding_model / <wakeword> heads, SURVEY §2 "resources/models").

No model file ships with the reference checkout (they are release assets fetched at run time,
/root/reference/openwakeword/__init__.py:8-51, utils.py:625-673) and none exists on the build or GPU
machines, so this module provides

  * the mel front-end tables, which are analytic (no learned values), and
  * deterministic synthetic weights of exactly the reference's shapes
    (notebooks/converting_google_speech_embedding_model.ipynb cell 18; openwakeword/train.py:56-83;
    docs/models/*.md), seeded so that every machine generates the same numbers.

Real weights enter through `openwakeword_amd.onnx_ingest` when `.onnx` files are available.
All arrays are host numpy; `openwakeword_amd.engine` packs and uploads them.
"""
from __future__ import annotations

import zlib
from typing import Optional

import numpy as np

SR = 16000
N_FFT = 512
HOP = 160
WIN = 400
N_MELS = 32
N_BINS = N_FFT // 2 + 1
MEL_TAPS = 16          # widest triangular filter touches 16 FFT bins

# Google speech_embedding CNN topology (ipynb cell 18): (kh, kw, cin, cout, pool_after)
CNN_TOPOLOGY = [
    (3, 3, 1, 24, None),
    (1, 3, 24, 24, None), (3, 1, 24, 24, (2, 2)),
    (1, 3, 24, 48, None), (3, 1, 48, 48, None),
    (1, 3, 48, 48, None), (3, 1, 48, 48, (1, 2)),
    (1, 3, 48, 72, None), (3, 1, 72, 72, None),
    (1, 3, 72, 72, None), (3, 1, 72, 72, (2, 2)),
    (1, 3, 72, 96, None), (3, 1, 96, 96, None),
    (1, 3, 96, 96, None), (3, 1, 96, 96, (1, 2)),
    (1, 3, 96, 96, None), (3, 1, 96, 96, None),
    (1, 3, 96, 96, None), (3, 1, 96, 96, (2, 2)),
    (3, 1, 96, 96, None),
]
BN_EPS = 1e-3
EMB_DIM = 96

# Pretrained head catalogue (shapes only): name -> (kind, T, hidden, n_out, layernorm)
# docs/models/alexa.md:9-24, hey_jarvis.md:9,38, timers.md:9-22; T from SURVEY §8a-F.
HEAD_CATALOGUE = {
    "alexa": ("binary", 16, 64, 1, True),
    "hey_mycroft": ("binary", 16, 64, 1, True),
    "hey_jarvis": ("gated", 16, 64, 1, True),
    "hey_rhasspy": ("binary", 16, 64, 1, True),
    "weather": ("binary", 16, 64, 1, True),
    "timer": ("multiclass", 34, 128, 7, False),
}


# ------------------------------------------------------------------------------- mel tables
def hann_window() -> np.ndarray:
    """Periodic Hann(400) as float32[400]; the kernel applies it at frame offsets 56..455
    (window centred in the 512-point frame)."""
    n = np.arange(WIN, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / WIN)).astype(np.float32)


def _mel_edges_hz() -> np.ndarray:
    f_sp = 200.0 / 3.0
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0

    def to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-9) / min_log_hz) / logstep, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), m * f_sp)

    return to_hz(np.linspace(to_mel(60.0), to_mel(3800.0), N_MELS + 2))


def mel_filterbank() -> np.ndarray:
    """Slaney-scale, area-normalised triangular filterbank, float32[257, 32] (Appendix A step 4)."""
    edges = _mel_edges_hz()
    freqs = np.linspace(0.0, SR / 2.0, N_BINS)
    ramps = edges[:, None] - freqs[None, :]
    diff = np.diff(edges)
    lower = -ramps[:-2] / diff[:-1, None]
    upper = ramps[2:] / diff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return w.T.astype(np.float32)


def mel_sparse_taps():
    """Compact form for the kernel: per mel bin the first FFT bin it touches and MEL_TAPS weights
    (zero padded).  Returns (start int32[32], taps float32[32,16], lo_bin, hi_bin)."""
    fb = mel_filterbank()
    start = np.zeros(N_MELS, dtype=np.int32)
    taps = np.zeros((N_MELS, MEL_TAPS), dtype=np.float32)
    lo, hi = N_BINS, 0
    for m in range(N_MELS):
        nz = np.nonzero(fb[:, m])[0]
        assert nz.size and nz[-1] - nz[0] + 1 <= MEL_TAPS
        start[m] = nz[0]
        taps[m, : nz[-1] - nz[0] + 1] = fb[nz[0]: nz[-1] + 1, m]
        lo, hi = min(lo, nz[0]), max(hi, nz[-1])
    return start, taps, int(lo), int(hi)


# ------------------------------------------------------------------------ synthetic weights
_LOGIT_CENTRE_SEED1234 = {"alexa": (10.05,), "hey_mycroft": (5.96,), "hey_jarvis": (-3.71, 4.21),
                          "hey_rhasspy": (-3.91,), "weather": (1.74,)}

def _rng(seed: int, tag: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(tag.encode())])


# E[a(x)^2] for x~N(0,1) through a(x)=max(max(0.2x,x),-0.4): keeps activations O(1) layer to layer
_ACT_GAIN = 0.52


def synthetic_embedding(seed: int = 1234) -> dict:
    """Random-init speech_embedding CNN: 20 bias-free convs (HWIO float32) + 19 BatchNorms
    (gamma, beta, moving_mean, moving_var); 332,088 parameters like the original (ipynb:859).
    The first kernel is mean-centred over (kh, kw) as the reference's CenterAround(0) constraint
    does (ipynb cell 18/19)."""
    conv, bn = [], []
    for li, (kh, kw, ci, co, _pool) in enumerate(CNN_TOPOLOGY):
        r = _rng(seed, f"conv{li}")
        fan_in = kh * kw * ci
        std = np.sqrt(1.0 / (fan_in * (_ACT_GAIN if li else 1.0)))
        w = r.normal(0.0, std, size=(kh, kw, ci, co))
        if li == 0:
            w = w - w.mean(axis=(0, 1), keepdims=True)
        conv.append(w.astype(np.float32))
        if li < len(CNN_TOPOLOGY) - 1:
            gamma = r.uniform(0.7, 1.3, co)
            beta = r.normal(0.0, 0.2, co)
            mean = r.normal(0.0, 0.2, co)
            var = r.uniform(0.6, 1.4, co)
            bn.append(tuple(a.astype(np.float32) for a in (gamma, beta, mean, var)))
    return {"conv": conv, "bn": bn}


def embedding_param_count(emb: dict) -> int:
    return int(sum(w.size for w in emb["conv"]) + sum(sum(a.size for a in t) for t in emb["bn"]))


def net_blocks(net: dict) -> list:
    """The hidden blocks of a head network in order, as (w [H, H], b [H], ln or None): block 0 under the keys every released model
    has ('w2', 'b2', 'ln2'; absent or None for a network without hidden block), further ones (train.py:73, n_blocks > 1) in
    net['more'] = [{'w', 'b', 'ln'}, ...]."""
    if net.get("w2") is None:
        if net.get("more"):
            raise ValueError("head network with further hidden blocks but no first one")
        return []
    return [(net["w2"], net["b2"], net.get("ln2"))] + [(m["w"], m["b"], m.get("ln")) for m in net.get("more") or []]


def _synthetic_mlp(r: np.random.Generator, n_in: int, hidden: int, n_out: int, layernorm: bool,
                   out_gain: float = 8.0, n_blocks: int = 1) -> dict:
    def lin(i, o, gain=1.0):
        return (r.normal(0.0, gain / np.sqrt(i), size=(i, o)).astype(np.float32),
                r.normal(0.0, 0.1, size=o).astype(np.float32))

    def ln(n):
        return (r.uniform(0.8, 1.2, n).astype(np.float32), r.normal(0.0, 0.1, n).astype(np.float32))

    w1, b1 = lin(n_in, hidden)
    w2, b2 = lin(hidden, hidden, 1.4)
    w3, b3 = lin(hidden, n_out, out_gain)
    net = {"w1": w1, "b1": b1, "ln1": ln(hidden) if layernorm else None,
           "w2": w2, "b2": b2, "ln2": ln(hidden) if layernorm else None,
           "w3": w3, "b3": b3}
    if n_blocks == 0:                                   # (drawn all the same, so that the other layers do not depend on n_blocks)
        net["w2"] = net["b2"] = net["ln2"] = None
    more = []
    for _ in range(max(0, n_blocks - 1)):
        w, b = lin(hidden, hidden, 1.4)
        more.append({"w": w, "b": b, "ln": ln(hidden) if layernorm else None})
    if more:
        net["more"] = more
    return net


def synthetic_head(name: str, seed: int = 1234, kind: Optional[str] = None, T: Optional[int] = None,
                   hidden: Optional[int] = None, n_out: Optional[int] = None,
                   layernorm: Optional[bool] = None, n_blocks: int = 1) -> dict:
    """Random-init wakeword head with the catalogue shape of `name` (102,849 params for the binary
    heads, docs/models/alexa.md:26-29).  Unknown names get the standard binary shape.  n_blocks: hidden blocks of
    train.py:73 (1 in the released models)."""
    base = name.split("_v0")[0]
    cat = HEAD_CATALOGUE.get(base, ("binary", 16, 64, 1, True))
    kind = kind or cat[0]
    T = T or cat[1]
    hidden = hidden or cat[2]
    n_out = n_out or cat[3]
    layernorm = cat[4] if layernorm is None else layernorm
    r = _rng(seed, f"head:{base}")
    if kind == "rnn":
        return synthetic_rnn_head(r, int(T), int(n_out))
    head = {"kind": kind, "T": int(T), "hidden": int(hidden), "n_out": int(n_out),
            "net": _synthetic_mlp(r, T * EMB_DIM, hidden, n_out, layernorm,
                                  out_gain=0.5 if kind == "multiclass" else 8.0, n_blocks=n_blocks)}
    if kind == "gated":
        head["net2"] = _synthetic_mlp(r, T * EMB_DIM, hidden, n_out, layernorm, n_blocks=n_blocks)
    # Conditioning of the synthetic data only: random heads sit far from 0.5 on the synthetic
    # embedding's (large, static) mean vector.  For the default seed the output biases are shifted by
    # minus the mean logit measured once on the three fixture clips, so that scores swing either
    # side of 0.5 and threshold / patience / debounce / gating logic is actually exercised.
    if seed == 1234 and base in _LOGIT_CENTRE_SEED1234 and kind != "multiclass":
        c = _LOGIT_CENTRE_SEED1234[base]
        head["net"]["b3"] = (head["net"]["b3"] + np.float32(c[0])).astype(np.float32)
        if kind == "gated":
            head["net2"]["b3"] = (head["net2"]["b3"] + np.float32(c[1])).astype(np.float32)
    return head


RNN_HID = 64         # train.py:88: nn.LSTM(input_shape[-1], 64, num_layers=2, bidirectional=True)


def synthetic_rnn_head(r: np.random.Generator, T: int = 16, n_out: int = 1) -> dict:
    """Random-init head of the reference's other model_type, "rnn" (train.py:85-98): a 2-layer bidirectional LSTM(64) over the T feature
    rows, Linear(128 -> n_out) on the LAST time step's output, Sigmoid (one class) or ReLU (+ the softmax train.py:152-165 adds at
    export).  'lstm': [layer][direction] = (w [in + 64, 256]: rows (x ; h), columns (i | f | g | o) -- torch's gate order --,
    b [256] = b_ih + b_hh), in = 96 for layer 0 and 128 for layer 1; 'w_out' [128, n_out], 'b_out' [n_out]."""
    H = RNN_HID
    lstm = []
    for layer in range(2):
        n_in = EMB_DIM if layer == 0 else 2 * H
        dirs = []
        for _d in range(2):
            # (the synthetic embeddings are of order 10: layer-0 input weights small enough that the gates are not all saturated)
            w = np.concatenate([r.normal(0, (0.35 if layer == 0 else 1.2) / np.sqrt(n_in), (n_in, 4 * H)),
                                r.normal(0, 0.8 / np.sqrt(H), (H, 4 * H))], axis=0).astype(np.float32)
            b = r.normal(0, 0.3, 4 * H).astype(np.float32)
            dirs.append((w, b))
        lstm.append(dirs)
    return {"kind": "rnn", "T": int(T), "hidden": H, "n_out": int(n_out), "lstm": lstm,
            "w_out": r.normal(0, 12.0 / np.sqrt(2 * H), (2 * H, n_out)).astype(np.float32),
            "b_out": r.normal(0, 0.3, n_out).astype(np.float32)}


def head_param_count(head: dict) -> int:
    if head["kind"] == "rnn":
        return sum(w.size + b.size for layer in head["lstm"] for w, b in layer) + head["w_out"].size + head["b_out"].size
    def cnt(net):
        n = 0
        for k, v in net.items():
            if v is None:
                continue
            if k == "more":
                n += sum(m["w"].size + m["b"].size + (sum(a.size for a in m["ln"]) if m.get("ln") is not None else 0) for m in v)
                continue
            n += sum(a.size for a in v) if isinstance(v, tuple) else v.size
        return n
    return cnt(head["net"]) + (cnt(head["net2"]) if "net2" in head else 0)


def bn_scale_shift(bn_tuple):
    """Inference BatchNorm as y = x*scale + shift, float32 (float64 intermediate)."""
    g, b, m, v = (np.asarray(a, dtype=np.float64) for a in bn_tuple)
    scale = g / np.sqrt(v + BN_EPS)
    return scale.astype(np.float32), (b - m * scale).astype(np.float32)


def synthetic_pcm(n_streams: int, n_samples: int, seed: int = 0xA11CE, rms: float = 3000.0) -> np.ndarray:
    """Gaussian noise int16[S, n] with the RMS of the reference's fixtures (SURVEY §8d "(n)")."""
    r = np.random.default_rng(seed)
    x = np.rint(r.normal(0.0, rms, size=(n_streams, n_samples)))
    return np.clip(x, -32768, 32767).astype(np.int16)


# ------------------------------------------------------------------------ voice-activity stand-in network
# Silero's silero_vad.onnx (vad.py:54-81) is a release asset whose graph is not in the reference checkout, so the device
# runs a STRUCTURAL stand-in with the interface the reference fixes (640-sample sub-frames / 32767, (h, c) [2, 1, 64]
# carried, one score per sub-frame): 256/64 STFT magnitudes of bins 1..128, log(1 + 50 |X|), four Conv1d(k=3)+ReLU
# (128->16 s1, 16->32 s2, 32->32 s2, 32->64 s1), 2-layer LSTM(64), ReLU -> Linear(64->1) -> sigmoid, mean over time.
# The numpy restatement the tests check it against is oracle/vad_standin.py.
VAD_N_FFT, VAD_HOP, VAD_SUB, VAD_BINS, VAD_HID = 256, 64, 640, 128, 64
VAD_MAG_GAIN = 50.0
VAD_ENC = ((128, 16, 1), (16, 32, 2), (32, 32, 2), (32, 64, 1))


def synthetic_vad(seed: int = 1234) -> dict:
    """Random-init weights of the stand-in: 'enc' [(w [3, cin, cout], b [cout])] x 4, 'lstm' [(w [128, 256] rows (x ; h),
    columns (i | f | g | o), b [256] = b_ih + b_hh)] x 2, 'dec' (w [64], b)."""
    r = _rng(seed, "vad")
    enc = []
    for cin, cout, _stride in VAD_ENC:
        enc.append((r.normal(0.0, np.sqrt(2.0 / (3 * cin)), size=(3, cin, cout)).astype(np.float32),
                    r.normal(0.0, 0.1, size=cout).astype(np.float32)))
    enc[0] = ((enc[0][0] * 0.25).astype(np.float32), enc[0][1])         # the log-magnitudes are O(4), not O(1)
    lstm = []
    for _layer in range(2):
        w = r.normal(0.0, 1.0 / np.sqrt(VAD_HID), size=(2 * VAD_HID, 4 * VAD_HID)).astype(np.float32)
        b = r.normal(0.0, 0.1, size=4 * VAD_HID).astype(np.float32)
        b[VAD_HID:2 * VAD_HID] += 1.0                                    # forget-gate bias, the usual initialisation
        lstm.append((w, b))
    dec = (r.normal(0.0, 1.5, size=VAD_HID).astype(np.float32), np.float32(-0.3))
    return {"enc": enc, "lstm": lstm, "dec": dec}


def vad_hann() -> np.ndarray:
    n = np.arange(VAD_N_FFT, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / VAD_N_FFT)).astype(np.float32)
