"""
TEST INFRASTRUCTURE ONLY -- a stand-in for the `onnxruntime` module.

The reference hands its arithmetic to `ort.InferenceSession(...).run` at three call sites
(/root/reference/openwakeword/utils.py:84-93, model.py:153-159).  onnxruntime and the `.onnx` release
assets do not exist on this machine, so the reference cannot be imported as is.  Installing this
module as ``sys.modules['onnxruntime']`` lets the reference's OWN streaming / post-processing code run
unmodified, with the stage math supplied by `oracle.oww_oracle` (or by any other callable, e.g. the HIP
stage entry points).  tests/golden/make_golden.py uses it to produce the golden vectors that pin
`OracleAudioFeatures` / `OracleModel`.

Dispatch is on the model file's basename:
    melspectrogram.onnx  -> STAGES['mel'](x f32[B,N])        -> [B,1,F,32]
    embedding_model.onnx -> STAGES['embed'](x f32[B,76,32,1]) -> [B,1,1,96]
    <name>_v0.1.onnx / <name>.onnx -> HEADS[name] = (fn(x f32[1,T,96]) -> [1,n_out], T, n_out)
    silero_vad.onnx      -> STAGES['vad'].run(None, {'input','h','c','sr'}) -> [out, h, c]   (oracle/pseudo_vad.py)
"""
from __future__ import annotations

import os
import types
from typing import Callable, Dict, Tuple

STAGES: Dict[str, Callable] = {}
HEADS: Dict[str, Tuple[Callable, int, int]] = {}


class SessionOptions:
    inter_op_num_threads = 1
    intra_op_num_threads = 1


class _IO:
    def __init__(self, name, shape):
        self.name = name
        self.shape = shape


def _head_key(base: str) -> str:
    stem = os.path.splitext(base)[0]
    if stem in HEADS:
        return stem
    short = stem.split("_v0")[0]
    if short in HEADS:
        return short
    raise FileNotFoundError(f"fake_ort: no head registered for {base!r}")


class InferenceSession:
    def __init__(self, path, sess_options=None, providers=None):
        base = os.path.basename(path)
        self._providers = list(providers or ["CPUExecutionProvider"])
        if base.startswith("melspectrogram"):
            self._fn, self._in, self._out = STAGES["mel"], _IO("input", ["batch", "samples"]), _IO("output", ["time", 1, "t", 32])
        elif base.startswith("silero_vad"):
            # vad.py:80-81,121-124: four named inputs, three outputs; the network behind it is whatever STAGES['vad'] holds
            self._vad = STAGES["vad"]
            self._fn, self._in, self._out = None, _IO("input", [1, "n"]), _IO("output", [1, 1])
        elif base.startswith("embedding_model"):
            self._fn, self._in, self._out = STAGES["embed"], _IO("input_1", ["unk", 76, 32, 1]), _IO("conv2d_19", ["unk", 1, 1, 96])
        else:
            fn, T, n_out = HEADS[_head_key(base)]
            self._fn, self._in, self._out = fn, _IO("onnx::Flatten_0", [1, T, 96]), _IO(base, [1, n_out])

    def run(self, output_names, feeds):
        if self._fn is None:
            return self._vad.run(output_names, feeds)
        (x,) = feeds.values()
        return [self._fn(x)]

    def get_inputs(self):
        return [self._in]

    def get_outputs(self):
        return [self._out]

    def get_providers(self):
        return self._providers


def as_module() -> types.ModuleType:
    m = types.ModuleType("onnxruntime")
    m.SessionOptions = SessionOptions
    m.InferenceSession = InferenceSession
    return m
