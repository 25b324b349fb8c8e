"""
openwakeword_amd -- the streaming hot path of dscripka/openWakeWord on AMD MI355X (gfx950).

    from openwakeword_amd import Model            # drop-in for openwakeword.Model(inference_framework="hip")
    from openwakeword_amd import BatchedModel     # S concurrent streams per GPU, scores [S, n_labels] per 80 ms
    from openwakeword_amd.engine import StreamEngine   # thin ctypes layer over libowwhip.so (include/owwhip.h)

All arithmetic runs in hand-written HIP kernels (openwakeword_amd/csrc); importing the package needs no GPU,
using it does -- there is no CPU fallback.
"""
from .model import (MODELS, FEATURE_MODELS, AudioFeatures, BatchedModel, Model, get_pretrained_model_paths,
                    model_class_mappings)
from .vad import VAD

__all__ = ["Model", "BatchedModel", "AudioFeatures", "VAD", "MODELS", "FEATURE_MODELS", "model_class_mappings",
           "get_pretrained_model_paths"]
__version__ = "0.1.0"
