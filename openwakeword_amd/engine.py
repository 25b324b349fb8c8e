"""
StreamEngine: S concurrent audio streams advanced one 80 ms frame per call on one MI355X.

This is the batched form of the reference's per-object hot loop
(`openwakeword.Model.predict`, /root/reference/openwakeword/model.py:232-386, which drives
`utils.AudioFeatures.__call__`, utils.py:409-463): the reference keeps one Python object per stream and
calls onnxruntime 2+H times per frame; here every stream's state lives in HBM and one `step()` runs the
mel / embedding / head / post-processing kernels of libowwhip.so over all of them.

Thin by design: blob packing + ctypes calls.  All arithmetic happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from . import _lib, weights as W

CHUNK = 1280
EMB_DIM = 96
_KIND = {"binary": 0, "gated": 1, "multiclass": 2, "rnn": 3}
# rows x mel-width x channels that each CNN layer produces per 80 ms step (incremental form)
LAYER_NEW_SHAPES = ([(8, 32, 24)] * 3 + [(4, 16, 48)] * 4 + [(4, 8, 72)] * 4 + [(2, 4, 96)] * 4 +
                    [(2, 2, 96)] * 4 + [(1, 1, 96)])
KERNEL_CLASSES = ["mel", "stageA", "stageB", "stageC", "stageD", "stageE", "heads", "postproc", "vad_front", "vad_lstm"]


def pack_mel_blob() -> np.ndarray:
    start, taps, lo, hi = W.mel_sparse_taps()
    assert lo == 2 and hi == 121, "the mel kernel keeps FFT bins 2..121 only"
    return np.concatenate([W.hann_window().view(np.uint8), start.astype(np.int32).view(np.uint8),
                           taps.astype(np.float32).ravel().view(np.uint8)])


def pack_embedding_blob(emb: dict) -> np.ndarray:
    parts = []
    for li, w in enumerate(emb["conv"]):
        kh, kw, ci, co, _ = W.CNN_TOPOLOGY[li]
        w = np.ascontiguousarray(w, dtype=np.float32)
        if w.shape != (kh, kw, ci, co):
            raise ValueError(f"conv {li}: expected HWIO {(kh, kw, ci, co)}, got {w.shape}")
        parts.append(w.ravel())
        if li < len(emb["conv"]) - 1:
            scale, shift = W.bn_scale_shift(emb["bn"][li])
            parts += [scale, shift]
    return np.concatenate(parts).astype(np.float32)


def pack_head_blob(head: dict) -> np.ndarray:
    T, H, O = int(head["T"]), int(head["hidden"]), int(head["n_out"])
    if head["kind"] == "rnn":
        # train.py:85-98: hdr = {3, T, 64, n_out, 0, 0, 0, 0}; per layer, per direction: w [in + 64][256] (rows x ; h, columns i | f | g | o),
        # b [256]; then w_out [128][n_out], b_out [n_out]
        if H != W.RNN_HID or len(head["lstm"]) != 2 or any(len(layer) != 2 for layer in head["lstm"]):
            raise ValueError("an rnn head is a 2-layer bidirectional LSTM(64) (train.py:88)")
        parts = [np.array([3, T, H, O, 0, 0, 0, 0], dtype=np.int32).view(np.float32)]
        for li, layer in enumerate(head["lstm"]):
            n_in = EMB_DIM if li == 0 else 2 * H
            for w, b in layer:
                if w.shape != (n_in + H, 4 * H) or b.shape != (4 * H,):
                    raise ValueError("rnn head weight shapes do not match train.py:85-98")
                parts += [w.ravel(), b]
        if head["w_out"].shape != (2 * H, O) or head["b_out"].shape != (O,):
            raise ValueError("rnn head output layer shape")
        parts += [head["w_out"].ravel(), head["b_out"]]
        return np.concatenate([np.ascontiguousarray(p, dtype=np.float32).ravel() for p in parts])
    has_ln = head["net"].get("ln1") is not None
    n_blocks = len(W.net_blocks(head["net"]))             # train.py:73; hdr[5] counts the blocks beyond the released models' one
    hdr = np.array([_KIND[head["kind"]], T, H, O, int(has_ln), n_blocks - 1, 0, 0], dtype=np.int32)
    parts = [hdr.view(np.float32)]
    nets = [head["net"]] + ([head["net2"]] if head["kind"] == "gated" else [])
    for net in nets:
        blocks = W.net_blocks(net)
        if net["w1"].shape != (T * EMB_DIM, H) or net["w3"].shape != (H, O) or len(blocks) != n_blocks or \
                any(w.shape != (H, H) or b.shape != (H,) or (ln is not None) != has_ln for w, b, ln in blocks):
            raise ValueError("head weight shapes do not match its header")
        parts += [net["w1"].ravel(), net["b1"]]
        if has_ln:
            parts += list(net["ln1"])
        for w, b, ln in blocks:
            parts += [w.ravel(), b]
            if has_ln:
                parts += list(ln)
        parts += [net["w3"].ravel(), net["b3"]]
    return np.concatenate([np.ascontiguousarray(p, dtype=np.float32).ravel() for p in parts])


def pack_vad_blob(vad: dict) -> np.ndarray:
    """Blob of oww_load_vad for the voice-activity stand-in network (weights.synthetic_vad layout)."""
    hdr = np.array([1, W.VAD_N_FFT, W.VAD_HOP, W.VAD_BINS, W.VAD_HID, 0, 0, 0], dtype=np.int32)
    parts = [hdr.view(np.float32), np.array([W.VAD_MAG_GAIN], np.float32), W.vad_hann()]
    for (w, b), (cin, cout, _s) in zip(vad["enc"], W.VAD_ENC):
        if w.shape != (3, cin, cout) or b.shape != (cout,):
            raise ValueError(f"VAD encoder layer: expected w {(3, cin, cout)}, b {(cout,)}, got {w.shape}, {b.shape}")
        parts += [w.ravel(), b]
    for w, b in vad["lstm"]:
        if w.shape != (2 * W.VAD_HID, 4 * W.VAD_HID) or b.shape != (4 * W.VAD_HID,):
            raise ValueError("VAD LSTM layer: expected w [128, 256] (rows x ; h, columns i | f | g | o), b [256]")
        parts += [w.ravel(), b]
    wd, bd = vad["dec"]
    parts += [np.asarray(wd, np.float32).ravel(), np.array([bd], np.float32)]
    return np.concatenate([np.ascontiguousarray(p, dtype=np.float32).ravel() for p in parts])


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_CAL_CLIPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources", "calibration_clips.npz")


def default_calibration_pcm() -> Optional[np.ndarray]:
    """Speech for the commit-time calibration / self-test of the fp16-split kernels (oww_set_calibration): int16 [12, 16 * 1280].
    The three 16 kHz clips are the reference's own test fixtures (/root/reference/tests/data/{alexa_test,hey_mycroft_test,hey_jane}.wav,
    2.4 s of speech in all; shipped as resources/calibration_clips.npz), each tiled to 16 frames as recorded, peak-normalised to full
    scale, and 20 dB down, plus one half-period-shifted copy of each: the loudness range a microphone front end delivers, with the
    spectral and temporal structure noise and square waves do not have.  None when the resource file is absent."""
    if not os.path.exists(_CAL_CLIPS):
        return None
    z = np.load(_CAL_CLIPS)
    n = 16 * CHUNK
    rows = []
    for k in sorted(z.files):
        x = z[k].astype(np.float64)
        full = x * (32767.0 / max(1.0, np.abs(x).max()))
        for y in (x, full, 0.1 * x, np.roll(full, len(x) // 2)):
            rows.append(np.clip(np.rint(np.resize(y, n)), -32768, 32767).astype(np.int16))
    return np.stack(rows)


class StreamEngine:
    """One GPU, S streams.  `heads` maps model name -> head dict (see weights.synthetic_head).

    use_mfma selects the kernel family of the embedding CNN / heads (include/owwhip.h): 3 = register-resident kernels with
    every fp32 product evaluated as three f16 MFMAs (hi/lo operand split, fp32 accumulation; default, same tolerances as
    fp32), 1 = the same kernels on the exact-fp32 MFMA, 2 = LDS-tiled fp32 MFMA kernels, 0 = plain VALU."""

    def __init__(self, n_streams: int, heads: Dict[str, dict], embedding: Optional[dict] = None,
                 device: int = 0, max_chunks: int = 1, use_mfma: int = 3, debug_layers: bool = False,
                 feature_ring: int = 0, hip_stream: int = 0, vad: Optional[dict] = None, vad_threshold: float = 0.0,
                 calibration_pcm: Union[np.ndarray, str, None] = "default"):
        """`vad`: weights of the on-device voice-activity stand-in network (weights.synthetic_vad layout); when given, every
        step also runs it on the frame's two 640-sample sub-frames and gates the scores with `vad_threshold` (model.py:366-381).
        `calibration_pcm` (use_mfma = 3): audio of the deployment's domain, int16 [n, k * 1280], that oww_commit adds to its built-in
        probes when it calibrates the activation scales and holds the fp16-split kernels to the exact-fp32 ones; "default" = speech
        (default_calibration_pcm), None = the synthetic probes only."""
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self._inflight: List[np.ndarray] = []
        self.n_streams = int(n_streams)
        self.max_chunks = int(max_chunks)
        self.head_names = list(heads.keys())
        self.heads = heads
        self._embedding = embedding
        self._cfg = dict(device=int(device), use_mfma=int(use_mfma))
        cfg = _lib.Config(int(device), self.n_streams, self.max_chunks, int(feature_ring), int(use_mfma),
                          int(bool(debug_layers)), C.c_void_p(hip_stream) if hip_stream else None)
        _lib.check(self._lib.oww_create(C.byref(cfg), C.byref(self._h)))
        try:
            blob = pack_mel_blob()
            _lib.check(self._lib.oww_load_mel(self._h, _ptr(blob), blob.nbytes))
            blob = pack_embedding_blob(embedding if embedding is not None else W.synthetic_embedding())
            _lib.check(self._lib.oww_load_embedding(self._h, _ptr(blob), blob.nbytes))
            self.head_cols = {}
            col = 0
            for name, head in heads.items():
                blob = pack_head_blob(head)
                _lib.check(self._lib.oww_add_head(self._h, _ptr(blob), blob.nbytes))
                self.head_cols[name] = (col, col + int(head["n_out"]))
                col += int(head["n_out"])
            self.has_vad = vad is not None
            if vad is not None:
                blob = pack_vad_blob(vad)
                _lib.check(self._lib.oww_load_vad(self._h, _ptr(blob), blob.nbytes))
            if isinstance(calibration_pcm, str):
                calibration_pcm = default_calibration_pcm() if calibration_pcm == "default" else None
            if calibration_pcm is not None and int(use_mfma) == 3:
                cal = np.asarray(calibration_pcm)
                if cal.dtype.kind not in "iu":
                    # (float audio in [-1, 1] would be truncated to silence and the scale ladder calibrated on nothing)
                    raise ValueError(f"Input data must be 16-bit integers (i.e., 16-bit PCM audio). You provided {cal.dtype} data.")
                if cal.size and (int(cal.min()) < -32768 or int(cal.max()) > 32767):      # wider integer types: valid 16-bit PCM only
                    raise ValueError(f"calibration_pcm holds values outside the 16-bit PCM range ({int(cal.min())} .. {int(cal.max())})")
                cal = np.ascontiguousarray(cal, dtype=np.int16)
                if cal.ndim != 2 or cal.shape[1] < CHUNK:
                    raise ValueError("calibration_pcm must be int16 [n_streams, n_frames * 1280]")
                n_frames = cal.shape[1] // CHUNK
                cal = np.ascontiguousarray(cal[:, :n_frames * CHUNK])
                _lib.check(self._lib.oww_set_calibration(self._h, _ptr(cal), cal.shape[0], n_frames))
            _lib.check(self._lib.oww_commit(self._h))
            if vad_threshold:
                _lib.check(self._lib.oww_set_vad_threshold(self._h, float(vad_threshold)))
        except Exception:
            self.close()
            raise
        self.n_labels = self._lib.oww_n_labels(self._h)
        self.n_streams_padded = (self.n_streams + 31) // 32 * 32
        self.feature_ring = max([16] + [int(h["T"]) for h in heads.values()] + [int(feature_ring)])

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.oww_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state
    def reset(self, stream_ids: Optional[Sequence[int]] = None, init_features: Optional[np.ndarray] = None):
        ids = None
        n = 0
        if stream_ids is not None:
            ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
            n = ids.size
        feat = None
        if init_features is not None:
            feat = np.ascontiguousarray(init_features, dtype=np.float32)
            if feat.shape != (self.feature_ring, EMB_DIM):
                raise ValueError(f"init_features must be [{self.feature_ring}, 96] (oldest row first), got {feat.shape}")
        _lib.check(self._lib.oww_reset(self._h, _ptr(ids), n, _ptr(feat)))

    def set_postproc(self, patience: Optional[Sequence[int]] = None, threshold: Optional[Sequence[float]] = None,
                     debounce_frames: int = 0):
        pat = None if patience is None else np.ascontiguousarray(patience, dtype=np.int32)
        thr = None if threshold is None else np.ascontiguousarray(threshold, dtype=np.float32)
        for a in (pat, thr):
            if a is not None and a.size != self.n_labels:
                raise ValueError(f"need one value per label ({self.n_labels})")
        _lib.check(self._lib.oww_set_postproc(self._h, _ptr(pat), _ptr(thr), int(debounce_frames)))

    # ---- hot loop
    def step(self, pcm: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        """pcm int16 [S, 1280*k] (host) -> scores fp32 [S, n_labels] (host, blocking)."""
        pcm = np.ascontiguousarray(pcm)
        if pcm.dtype != np.int16:
            raise ValueError(f"Input data must be 16-bit integers (i.e., 16-bit PCM audio). You provided {pcm.dtype} data.")
        if pcm.ndim != 2 or pcm.shape[0] != self.n_streams or pcm.shape[1] % CHUNK or pcm.shape[1] == 0:
            raise ValueError(f"pcm must be [n_streams={self.n_streams}, 1280*k], got {pcm.shape}")
        k = pcm.shape[1] // CHUNK
        if out is None:
            out = np.empty((self.n_streams, self.n_labels), dtype=np.float32)
        _lib.check(self._lib.oww_step(self._h, _ptr(pcm), 0, k, _ptr(out), 0))
        return out

    def step_masked(self, pcm: np.ndarray, stream_on: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        """One 1280-sample step for the streams with stream_on != 0 only (oww_step_masked): the others keep every bit of their
        state and repeat their previous scores -- the batched form of a client that simply does not call predict() while it has
        no audio (examples/web/streaming_server.py:49-66)."""
        pcm = np.ascontiguousarray(pcm)
        if pcm.dtype != np.int16:
            raise ValueError(f"Input data must be 16-bit integers (i.e., 16-bit PCM audio). You provided {pcm.dtype} data.")
        if pcm.shape != (self.n_streams, CHUNK):
            raise ValueError(f"pcm must be [n_streams={self.n_streams}, 1280], got {pcm.shape}")
        on = np.ascontiguousarray(np.asarray(stream_on) != 0, dtype=np.uint8)
        if on.shape != (self.n_streams,):
            raise ValueError(f"stream_on must be [n_streams={self.n_streams}], got {on.shape}")
        if out is None:
            out = np.empty((self.n_streams, self.n_labels), dtype=np.float32)
        _lib.check(self._lib.oww_step_masked(self._h, _ptr(pcm), 0, _ptr(on), 0, _ptr(out), 0))
        return out

    def step_masked_device(self, pcm_dev_ptr: int, stream_on: np.ndarray, scores_dev_ptr: int = 0) -> None:
        """oww_step_masked on device-resident PCM / scores with the participation mask on the host (asynchronous); with at most
        half of the streams taking part only their groups are launched."""
        on = np.ascontiguousarray(stream_on, dtype=np.uint8)
        if on.shape != (self.n_streams,):
            raise ValueError(f"stream_on must be [{self.n_streams}]")
        _lib.check(self._lib.oww_step_masked(self._h, C.c_void_p(int(pcm_dev_ptr)), 1, _ptr(on), 0,
                                             C.c_void_p(int(scores_dev_ptr)) if scores_dev_ptr else None, 1))

    def resample(self, pcm: np.ndarray, sample_rate: int) -> np.ndarray:
        """int16 [S, n_in] at `sample_rate` Hz -> int16 [S, n_in * 16000 // sample_rate] at 16 kHz on the device (oww_resample with the
        filter bank of resample.design; stateless per call, like the per-message resampy call of examples/web/streaming_server.py:57-58)."""
        from . import resample as R
        pcm = np.ascontiguousarray(pcm)
        if pcm.dtype != np.int16 or pcm.ndim != 2 or pcm.shape[0] != self.n_streams:
            raise ValueError(f"pcm must be int16 [n_streams={self.n_streams}, n], got {pcm.dtype} {pcm.shape}")
        if int(sample_rate) == 16000:
            return pcm
        p, q, taps = R.design(int(sample_rate))
        n_out = (pcm.shape[1] * q) // p
        out = np.empty((self.n_streams, n_out), dtype=np.int16)
        _lib.check(self._lib.oww_resample(self._h, _ptr(pcm), 0, pcm.shape[1], p, q, _ptr(taps), taps.shape[1], _ptr(out), 0, n_out))
        return out

    def step_raw(self, pcm: np.ndarray) -> np.ndarray:
        """Like step(), but returns the head outputs before post-processing (model.py:313-317)."""
        pcm = np.ascontiguousarray(pcm)
        if pcm.dtype != np.int16:
            raise ValueError(f"Input data must be 16-bit integers (i.e., 16-bit PCM audio). You provided {pcm.dtype} data.")
        if pcm.ndim != 2 or pcm.shape[0] != self.n_streams or pcm.shape[1] % CHUNK or pcm.shape[1] == 0:
            raise ValueError(f"pcm must be [n_streams={self.n_streams}, 1280*k], got {pcm.shape}")
        _lib.check(self._lib.oww_step(self._h, _ptr(pcm), 0, pcm.shape[1] // CHUNK, None, 0))
        out = np.empty((self.n_streams, self.n_labels), dtype=np.float32)
        _lib.check(self._lib.oww_get_raw(self._h, _ptr(out)))
        return out

    def step_device(self, pcm_dev_ptr: int, n_chunks: int = 1, scores_dev_ptr: int = 0):
        """Asynchronous step on device pointers (e.g. torch tensors' data_ptr())."""
        _lib.check(self._lib.oww_step(self._h, C.c_void_p(pcm_dev_ptr), 1, int(n_chunks),
                                      C.c_void_p(scores_dev_ptr) if scores_dev_ptr else None, 1))

    # ---- host-fed pipeline: upload of step t+1 overlaps the kernels of step t (oww_submit / oww_collect) ----
    def submit(self, pcm: np.ndarray, stream_on: Optional[np.ndarray] = None) -> None:
        """Enqueue one step on host PCM int16 [S, 1280*k] and return at once.  At most two steps in flight; the array
        must stay alive and unmodified until the matching collect() (use `pinned_empty` buffers for an asynchronous
        upload).  Same results as step(); with `stream_on` ([S], one chunk) as step_masked()."""
        if not isinstance(pcm, np.ndarray) or pcm.dtype != np.int16 or not pcm.flags["C_CONTIGUOUS"]:
            raise ValueError("submit expects a C-contiguous int16 ndarray")
        if pcm.ndim != 2 or pcm.shape[0] != self.n_streams or pcm.shape[1] % CHUNK or pcm.shape[1] == 0:
            raise ValueError(f"pcm must be [n_streams={self.n_streams}, 1280*k], got {pcm.shape}")
        if stream_on is None:
            _lib.check(self._lib.oww_submit(self._h, _ptr(pcm), pcm.shape[1] // CHUNK))
        else:
            on = np.ascontiguousarray(np.asarray(stream_on) != 0, dtype=np.uint8)
            if on.shape != (self.n_streams,) or pcm.shape[1] != CHUNK:
                raise ValueError(f"a masked submit takes pcm [n_streams, 1280] and stream_on [n_streams], got {pcm.shape}, {on.shape}")
            _lib.check(self._lib.oww_submit_masked(self._h, _ptr(pcm), _ptr(on)))
        self._inflight.append(pcm)                      # keeps the buffer alive until collected

    def collect(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Scores fp32 [S, n_labels] of the oldest submitted step (blocks until they have arrived)."""
        if out is None:
            out = np.empty((self.n_streams, self.n_labels), dtype=np.float32)
        _lib.check(self._lib.oww_collect(self._h, _ptr(out)))
        if self._inflight:
            self._inflight.pop(0)
        return out

    def pinned_empty(self, shape, dtype=np.int16) -> np.ndarray:
        """A page-locked host array (hipHostMalloc) for PCM / score buffers; freed when the array is collected."""
        import weakref
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        _lib.check(self._lib.oww_host_alloc(C.byref(p), max(nbytes, 1)))
        buf = (C.c_byte * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        weakref.finalize(buf, self._lib.oww_host_free, C.c_void_p(p.value))
        return arr

    def calibration_info(self) -> Dict[str, object]:
        """What oww_commit measured on its probe set (fp16-split family): per-layer |activation| maxima, the power-of-two scale
        exponents (CNN layers and the heads' feature scale), the number of probe streams and the commit-time self-test errors."""
        absmax = np.zeros(20, np.float32)
        exps = np.zeros(21, np.int32)
        n = C.c_int32(0)
        st = np.zeros(3, np.float32)
        _lib.check(self._lib.oww_calibration_info(self._h, _ptr(absmax), _ptr(exps), C.byref(n), _ptr(st)))
        return {"absmax": absmax, "layer_exp": exps[:20].copy(), "feature_exp": int(exps[20]), "n_probe_streams": int(n.value),
                "selftest_embedding_err": float(st[0]), "selftest_embedding_max": float(st[1]), "selftest_score_err": float(st[2])}

    def self_test(self, n_frames: int = 24, pcm: Optional[np.ndarray] = None, tol: float = 1e-3) -> Dict[str, float]:
        """Deploy-time check of the default f16-split kernels against the exact-fp32 family on THIS engine's weights.

        The f16-split family carries activations as f16 hi/lo pairs; a network whose activations leave the f16 range
        (|x| > 65504) would not fail loudly -- max/ReLU stages swallow the resulting NaNs -- it would silently score
        differently.  This runs `n_frames` frames of 32 probe streams (silence, quiet and full-scale noise, full-scale
        square waves, or the rows of `pcm` [32, 1280*n]) through a scratch engine of this family and one with
        use_mfma=1 and compares raw scores and embeddings.  Returns the maxima; raises OwwError beyond `tol`
        (the north-star score tolerance).  Does not touch this engine's streams."""
        S = 32
        if pcm is None:
            r = np.random.default_rng(2024)
            pcm = np.zeros((S, CHUNK * n_frames), np.int16)
            for i in range(1, S):
                amp = (30, 300, 3000, 12000, 32767)[i % 5]
                if i % 3 == 0:
                    t = np.arange(CHUNK * n_frames)
                    pcm[i] = np.where((t // (8 << (i % 4))) % 2, amp, -amp)
                else:
                    pcm[i] = np.clip(np.round(r.normal(0.0, amp, CHUNK * n_frames)), -32768, 32767)
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        if pcm.shape[0] != S or pcm.shape[1] % CHUNK:
            raise ValueError("self_test expects pcm [32, 1280*n]")
        outs = []
        for fam in (self._cfg["use_mfma"], 1):
            eng = StreamEngine(S, self.heads, self._embedding, device=self._cfg["device"], use_mfma=fam,
                               feature_ring=self.feature_ring)
            try:
                raw = [eng.step_raw(pcm[:, CHUNK * t: CHUNK * (t + 1)]) for t in range(pcm.shape[1] // CHUNK)]
                feats = np.stack([eng.get_features(s, 16) for s in range(S)])
                outs.append((np.stack(raw), feats))
            finally:
                eng.close()
        res = {"max_abs_score_diff": float(np.max(np.abs(outs[0][0] - outs[1][0]))) if outs[0][0].size else 0.0,
               "max_abs_embedding_diff": float(np.max(np.abs(outs[0][1] - outs[1][1]))),
               "max_abs_embedding": float(np.max(np.abs(outs[1][1])))}
        bad = not np.isfinite(outs[0][1]).all() or res["max_abs_score_diff"] > tol or \
            res["max_abs_embedding_diff"] > tol * max(1.0, res["max_abs_embedding"])
        if bad:
            raise _lib.OwwError(f"kernel family {self._cfg['use_mfma']} disagrees with the exact-fp32 family on these weights "
                                f"({res}): create the engine with use_mfma=1")
        return res

    def set_verifier(self, label: int, w: Optional[np.ndarray], bias: float = 0.0, threshold: float = 0.1) -> None:
        """Custom verifier of score column `label` on the device: sigmoid(w . last-T-feature-rows + bias) replaces the head
        output wherever that is >= threshold (model.py:320-328).  w = None removes it."""
        if w is None:
            _lib.check(self._lib.oww_set_verifier(self._h, int(label), None, 0, 0.0, 0.0))
            return
        w = np.ascontiguousarray(w, dtype=np.float32).ravel()
        _lib.check(self._lib.oww_set_verifier(self._h, int(label), _ptr(w), int(w.size), float(bias), float(threshold)))

    def set_vad_threshold(self, threshold: float) -> None:
        """VAD gate of model.py:366-381 on the device (0 = off); the scores come from push_vad()."""
        _lib.check(self._lib.oww_set_vad_threshold(self._h, float(threshold)))

    def push_vad(self, vad_scores: np.ndarray) -> None:
        """One voice-activity score per stream for the frame about to be stepped (what VAD.__call__ appends, vad.py:129-130)."""
        v = np.ascontiguousarray(vad_scores, dtype=np.float32)
        if v.shape != (self.n_streams,):
            raise ValueError(f"need one VAD score per stream ({self.n_streams})")
        _lib.check(self._lib.oww_push_vad(self._h, _ptr(v), 0))

    def get_vad(self) -> np.ndarray:
        """Voice-activity scores [S] the on-device network pushed for the last step (what VAD.__call__ appended, vad.py:129-130)."""
        out = np.empty(self.n_streams, dtype=np.float32)
        _lib.check(self._lib.oww_get_vad(self._h, _ptr(out)))
        return out

    def reset_vad(self, stream_ids: Optional[Sequence[int]] = None) -> None:
        """Zero the VAD state (recurrent h, c and score ring) of the listed streams (None = all).  reset() leaves it alone, like
        Model.reset() leaves Model.vad alone (model.py:226-230); call this when a stream slot is handed to a new caller."""
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        _lib.check(self._lib.oww_reset_vad(self._h, _ptr(ids), 0 if ids is None else ids.size))

    def sync(self):
        _lib.check(self._lib.oww_sync(self._h))

    def range_status(self, clear: bool = False) -> bool:
        """True when the fp16-split kernels (use_mfma=3) have seen an activation beyond the f16 range since the flag was
        last cleared (sticky; every step / collect / sync raises `OwwRangeError` while it is up).  Waits for the stream."""
        rc = self._lib.oww_range_status(self._h, int(bool(clear)))
        if rc == _lib.ERANGE:
            return True
        _lib.check(rc)
        return False

    def range_where(self):
        """(first_stream, n_streams) of a wave that saw the out-of-range value behind OwwRangeError -- the offender is among these
        streams -- or (-1, 0) when the flag is down / the position is unknown (include/owwhip.h: oww_range_where)."""
        a, b = C.c_int32(-1), C.c_int32(0)
        _lib.check(self._lib.oww_range_where(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    @property
    def scores_dev_ptr(self) -> int:
        return int(self._lib.oww_scores_dev(self._h) or 0)

    # ---- stage-level entry points (the reference's per-stage closures)
    def mel(self, pcm: np.ndarray) -> np.ndarray:
        """melspec_model_predict (utils.py:87): int16 [B, n] -> dB [B, F, 32]."""
        pcm = np.ascontiguousarray(np.atleast_2d(pcm))
        if pcm.dtype != np.int16:
            raise ValueError(f"Input data must be 16-bit integers (i.e., 16-bit PCM audio). You provided {pcm.dtype} data.")
        B, n = pcm.shape
        out = np.empty((B, (n - 512) // 160 + 1, 32), dtype=np.float32)
        _lib.check(self._lib.oww_mel(self._h, _ptr(pcm), B, n, _ptr(out)))
        return out

    def mel_clips(self, pcm: np.ndarray) -> np.ndarray:
        """Like mel(), but every clip gets its own clamp floor (the reference's per-clip CPU path, utils.py:243-290)."""
        pcm = np.ascontiguousarray(np.atleast_2d(pcm))
        if pcm.dtype != np.int16:
            raise ValueError(f"Input data must be 16-bit integers (i.e., 16-bit PCM audio). You provided {pcm.dtype} data.")
        B, n = pcm.shape
        if B > self.n_streams_padded:
            raise ValueError(f"at most {self.n_streams_padded} clips per call (the handle's stream count)")
        out = np.empty((B, (n - 512) // 160 + 1, 32), dtype=np.float32)
        _lib.check(self._lib.oww_mel_clips(self._h, _ptr(pcm), B, n, _ptr(out)))
        return out

    def embed_clips(self, pcm: np.ndarray) -> np.ndarray:
        """AudioFeatures.embed_clips on the device (utils.py:354-385): int16 [B, n] -> [B, n_windows, 96], mel, window
        walk and CNN without leaving HBM.  Clobbers the streaming state of streams [0, B)."""
        pcm = np.ascontiguousarray(pcm)
        if pcm.dtype != np.int16 or pcm.ndim != 2:
            raise ValueError("embed_clips expects a 2-D int16 array [B, samples]")
        B, n = pcm.shape
        if B > self.n_streams_padded:
            raise ValueError(f"at most {self.n_streams_padded} clips per call (the handle's stream count)")
        F = (n - 512) // 160 + 1
        if n < 512 or F < 76:
            raise ValueError("clips are shorter than one 76-frame embedding window (12512 samples, 782 ms)")
        out = np.empty((B, (F - 76) // 8 + 1, EMB_DIM), dtype=np.float32)
        _lib.check(self._lib.oww_embed_clips(self._h, _ptr(pcm), 0, B, n, _ptr(out), 0))
        return out

    def embed_clips_device(self, pcm_ptr: int, B: int, n: int, out_ptr: int) -> None:
        """The same on device buffers (int16 [B, n] -> f32 [B, n_windows, 96]); pointers are raw device addresses."""
        _lib.check(self._lib.oww_embed_clips(self._h, C.c_void_p(pcm_ptr), 1, int(B), int(n), C.c_void_p(out_ptr), 1))

    def embed(self, mel_rows: np.ndarray) -> np.ndarray:
        """embedding_model_predict over sliding windows (utils.py:229-236): [B, 76+8j, 32] -> [B, j+1, 96].
        Clobbers the streaming state of streams [0, B)."""
        m = np.ascontiguousarray(mel_rows, dtype=np.float32)
        if m.ndim == 2:
            m = m[None]
        B, rows, _ = m.shape
        out = np.empty((B, (rows - 76) // 8 + 1, EMB_DIM), dtype=np.float32)
        _lib.check(self._lib.oww_embed(self._h, _ptr(m), B, rows, _ptr(out)))
        return out

    def head(self, head: Union[int, str], features: np.ndarray) -> np.ndarray:
        """model_prediction_function[name] (model.py:137-138): [B, T, 96] -> [B, n_out]."""
        idx = self.head_names.index(head) if isinstance(head, str) else int(head)
        f = np.ascontiguousarray(features, dtype=np.float32)
        if f.ndim == 2:
            f = f[None]
        hd = self.heads[self.head_names[idx]]
        if f.shape[1:] != (hd["T"], EMB_DIM):
            raise ValueError(f"features must be [B, {hd['T']}, 96], got {f.shape}")
        out = np.empty((f.shape[0], hd["n_out"]), dtype=np.float32)
        _lib.check(self._lib.oww_head(self._h, idx, _ptr(f), f.shape[0], _ptr(out)))
        return out

    # ---- introspection
    def get_features(self, sid: int, T: int = 16) -> np.ndarray:
        out = np.empty((T, EMB_DIM), dtype=np.float32)
        _lib.check(self._lib.oww_get_features(self._h, int(sid), int(T), _ptr(out)))
        return out

    def get_mel(self, sid: int, n_rows: int = 8) -> np.ndarray:
        out = np.empty((n_rows, 32), dtype=np.float32)
        _lib.check(self._lib.oww_get_mel(self._h, int(sid), _ptr(out), int(n_rows)))
        return out

    def debug_layer(self, sid: int, layer: int) -> np.ndarray:
        """New rows of CNN layer `layer` (0..19) for stream sid from the last chunk: [rows, F, C]."""
        r, f, c = LAYER_NEW_SHAPES[layer]
        out = np.empty(r * f * c, dtype=np.float32)
        _lib.check(self._lib.oww_debug_read(self._h, int(sid), int(layer), _ptr(out), out.size))
        return out.reshape(r, f, c)

    def debug_profile(self) -> np.ndarray:
        """Shader-clock phase stamps [stage B..E][wave][mark] of workgroup $OWW_PROF_BLOCK (development aid)."""
        out = np.zeros((4, 16, 16), dtype=np.int64)
        _lib.check(self._lib.oww_debug_profile(self._h, _ptr(out), out.size))
        return out

    def enable_timing(self, on: bool = True):
        _lib.check(self._lib.oww_enable_timing(self._h, int(on)))

    def kernel_times(self) -> Dict[str, Dict[str, float]]:
        k = len(KERNEL_CLASSES)
        ms = (C.c_double * k)()
        n = (C.c_int64 * k)()
        _lib.check(self._lib.oww_kernel_times(self._h, ms, n))
        return {KERNEL_CLASSES[i]: {"ms": ms[i], "launches": int(n[i])} for i in range(k)}

    # ---- multi-GPU delivery of the scores over RCCL without torch.distributed (include/owwhip.h: oww_comm_*)
    @staticmethod
    def comm_id() -> bytes:
        """Rank 0: the 128 bytes every rank passes to comm_init (ncclGetUniqueId); ship them by any side channel."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().oww_comm_id(buf))
        return buf.raw

    def comm_init(self, comm_id: bytes, rank: int, world: int) -> None:
        if len(comm_id) != 128:
            raise ValueError("comm_id must be the 128 bytes of StreamEngine.comm_id()")
        _lib.check(self._lib.oww_comm_init(self._h, C.c_char_p(comm_id), int(rank), int(world)))

    def gather_scores(self, out_dev_ptr: int, counts: Sequence[int]) -> None:
        """Enqueue the gather of every rank's [S_r, n_labels] scores to rank 0's device buffer `out_dev_ptr` (asynchronous)."""
        c = np.ascontiguousarray(counts, dtype=np.int32)
        _lib.check(self._lib.oww_gather_scores(self._h, C.c_void_p(int(out_dev_ptr)), _ptr(c)))

    def comm_count(self) -> int:
        """Ranks the handle's RCCL communicator itself reports (ncclCommCount)."""
        n = C.c_int32(0)
        _lib.check(self._lib.oww_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def comm_destroy(self) -> None:
        _lib.check(self._lib.oww_comm_destroy(self._h))

    def use_graph(self, on: bool = True):
        _lib.check(self._lib.oww_use_graph(self._h, int(on)))
