"""
TEST INFRASTRUCTURE ONLY -- a stand-in for the `speexdsp_ns` package (absent from this image; the reference imports it lazily at
/root/reference/openwakeword/model.py:201-205 and feeds it 160-sample frames at model.py:481-504).

Not SpeexDSP's algorithm: a deterministic, STATEFUL, non-linear per-frame filter with the same interface
(`NoiseSuppression.create(frame_size, sample_rate)`, `.process(bytes) -> bytes`), so that the ORDER of operations around it can be
pinned -- cleaned audio to the preprocessor (model.py:272-273), the RAW call argument to the voice-activity detector (model.py:370),
`predict_clip`'s chunking on top (model.py:421-426) -- by running the reference's own Model with it (tests/golden/make_golden_onnx.py)
and the host shim / the HIP Model with the same module injected into `sys.modules`.  Any mistake in which audio goes where, or a
filter object re-created per call, changes the numbers: the filter carries a noise-floor tracker and an IIR tail across frames.
"""
from __future__ import annotations

import types

import numpy as np


class NoiseSuppression:
    instances = []                        # every object ever created (tests read .n_frames / .n_bytes)

    def __init__(self, frame_size: int, sample_rate: int):
        self.frame_size, self.sample_rate = int(frame_size), int(sample_rate)
        self.floor = 0.0                  # running noise-floor estimate (mean square)
        self.tail = 0.0                   # last output sample (IIR state across frames)
        self.n_frames = 0
        self.n_bytes = 0
        NoiseSuppression.instances.append(self)

    @classmethod
    def create(cls, frame_size: int, sample_rate: int) -> "NoiseSuppression":
        return cls(frame_size, sample_rate)

    def process(self, frame: bytes) -> bytes:
        x = np.frombuffer(frame, dtype=np.int16).astype(np.float64)
        if x.shape[0] != self.frame_size:                          # SpeexDSP takes whole frames only (model.py:490-492)
            raise ValueError(f"frame of {x.shape[0]} samples, expected {self.frame_size}")
        e = float(np.mean(x * x))
        self.floor = e if self.n_frames == 0 else min(e, 0.98 * self.floor + 0.02 * e)
        gain = 0.35 + 0.65 * (e - self.floor + 1.0) / (e + 1.0)    # quiet frames (near the floor) are attenuated most
        y = np.empty_like(x)
        t = self.tail
        for i in range(x.shape[0]):
            t = gain * x[i] + 0.125 * t
            y[i] = t
        out = np.clip(np.rint(y), -32768, 32767).astype(np.int16)
        self.tail = float(out[-1])
        self.n_frames += 1
        self.n_bytes += len(frame)
        return out.tobytes()


def as_module() -> types.ModuleType:
    m = types.ModuleType("speexdsp_ns")
    m.NoiseSuppression = NoiseSuppression
    m.__stand_in__ = True
    return m


def clean(pcm: np.ndarray, frame_size: int = 160) -> np.ndarray:
    """The whole-signal equivalent of feeding `pcm` frame by frame through one fresh object (what model.py:481-504 does over the calls
    of one clip): used by tests to say what the preprocessor must have seen."""
    ns = NoiseSuppression(frame_size, 16000)
    NoiseSuppression.instances.pop()
    n = (len(pcm) // frame_size) * frame_size
    return np.frombuffer(b"".join(ns.process(pcm[o:o + frame_size].tobytes()) for o in range(0, n, frame_size)), np.int16)
