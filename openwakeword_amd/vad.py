"""
Voice-activity gate of the reference (`openwakeword.VAD`, /root/reference/openwakeword/vad.py:54-130) around a pluggable
network.

The reference runs Silero's `silero_vad.onnx` through onnxruntime: `input[1, n] f32 (samples / 32767), h[2,1,64], c[2,1,64],
sr` -> `out[1,1], h', c'`.  That file is a release asset whose graph is described nowhere in the reference checkout, and there
is no onnxruntime here.  Two ways to supply the network:
  * this class: the reference's wrapper (sub-framing, /32767, state carry, mean per call, 125-deep score ring) around ANY object
    with the session interface (`run(None, feeds) -> [out, h, c]`) -- an onnxruntime session where one exists; `Model` applies the
    gate of model.py:366-381 on top of it;
  * on the device: `csrc/owwhip_vad.h` runs a structural stand-in with the same interface and state inside every batched step
    (`oww_load_vad`, `StreamEngine(vad=...)`, BASELINE configs[4]); `onnx_ingest.load_vad` turns a `silero_vad.onnx` into its
    weights when the file's graph IS that architecture and refuses it otherwise (then the first way applies).
"""
from __future__ import annotations

import os
from collections import deque

import numpy as np


class VAD:
    def __init__(self, session=None, model_path: str = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources",
                                                                    "models", "silero_vad.onnx"), n_threads: int = 1):
        if session is None:
            try:
                import onnxruntime as ort                                  # noqa: F401  (absent in this image)
            except ImportError:
                raise ValueError("the VAD gate needs a voice-activity network: pass vad_session=<object with "
                                 "run(None, {'input','h','c','sr'}) -> [out, h, c]>; silero_vad.onnx is a release asset "
                                 "whose graph is not part of the reference checkout and onnxruntime is not installed")
            if not os.path.exists(model_path):
                raise ValueError(f"{model_path} does not exist (the reference downloads it at run time)")
            opts = ort.SessionOptions()
            opts.inter_op_num_threads = n_threads
            opts.intra_op_num_threads = n_threads
            session = ort.InferenceSession(model_path, sess_options=opts, providers=["CPUExecutionProvider"])
        self.model = session
        self.prediction_buffer: deque = deque(maxlen=125)                  # 10 s of 80 ms frames (vad.py:84)
        self.sample_rate = np.array(16000).astype(np.int64)
        self.reset_states()

    def reset_states(self, batch_size: int = 1):
        self._h = np.zeros((2, batch_size, 64), dtype=np.float32)
        self._c = np.zeros((2, batch_size, 64), dtype=np.float32)

    def predict(self, x: np.ndarray, frame_size: int = 480) -> float:
        """Mean network output over consecutive `frame_size`-sample pieces of x, recurrent state carried (vad.py:98-127)."""
        scores = []
        for o in range(0, x.shape[0], frame_size):
            piece = (x[o:o + frame_size] / 32767).astype(np.float32)
            out, self._h, self._c = self.model.run(None, {"input": piece[None, ], "h": self._h, "c": self._c,
                                                          "sr": self.sample_rate})
            scores.append(out[0][0])
        return np.mean(scores)

    def __call__(self, x: np.ndarray, frame_size: int = 160 * 4):
        self.prediction_buffer.append(self.predict(x, frame_size))


def gate_value(vad_ring, threshold: float) -> bool:
    """True when the scores of this call must be zeroed (model.py:375-381): the maximum VAD score of the frames 0.4 to 0.56 s
    back -- ring[-7:-4], an empty slice during the first four calls counts as 0 -- lies below the threshold."""
    frames = list(vad_ring)[-7:-4]
    return (np.max(frames) if len(frames) > 0 else 0) < threshold
