"""
`Model` / `AudioFeatures`: the reference's Python surface for the streaming path, on the HIP library.

Mirrors `openwakeword.Model` (/root/reference/openwakeword/model.py:32-426) and the streaming half of
`openwakeword.utils.AudioFeatures` (utils.py:163-178, 387-463): same constructor keywords, `predict`,
`predict_clip`, `reset`, `prediction_buffer`, `models`, `model_inputs`, `model_outputs`, `class_mapping`,
`preprocessor.get_features`, same exceptions.  `inference_framework="hip"` is the value this package adds
to the reference's `"onnx"` / `"tflite"` selector (model.py:46,112,133).

One `Model` object = one audio stream, like the reference.  The arithmetic (mel, embedding CNN, heads) runs
in libowwhip.so on the GPU through a 1-stream `StreamEngine`; this file only keeps the reference's
host-side control flow: 1280-sample alignment with carry-over (utils.py:409-430), the "repeat the last
prediction" rule for short calls (model.py:299-307), custom verifier hook (model.py:320-328), first-5
zeroing / patience / debounce on the 30-deep `prediction_buffer` (model.py:330-363).  For many streams use
`BatchedModel`, which keeps that post-processing on the device as well.

There is no CPU fallback: constructing a Model without the HIP library or a GPU raises.
"""
from __future__ import annotations

import os
import pickle
import wave
from collections import defaultdict, deque
from functools import partial
from typing import Callable, DefaultDict, Dict, List, Optional, Sequence, Union

import numpy as np

from . import weights as W
from .engine import CHUNK, EMB_DIM, StreamEngine

# openwakeword/__init__.py:26-60 (names only; the files are release assets that are not in the checkout)
MODELS = {name: {"model_path": os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources", "models",
                                            f"{name}_v0.1.onnx")}
          for name in ("alexa", "hey_mycroft", "hey_jarvis", "hey_rhasspy", "timer", "weather")}
FEATURE_MODELS = {k: {"model_path": os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources", "models", f)}
                  for k, f in (("embedding", "embedding_model.onnx"), ("melspectrogram", "melspectrogram.onnx"))}
# openwakeword/__init__.py:62-71
model_class_mappings = {
    "timer": {"1": "1_minute_timer", "2": "5_minute_timer", "3": "10_minute_timer",
              "4": "20_minute_timer", "5": "30_minute_timer", "6": "1_hour_timer"}
}


def get_pretrained_model_paths(inference_framework: str = "hip") -> List[str]:
    return [MODELS[k]["model_path"] for k in MODELS]


def _load_head(path_or_name: str, synthetic_seed: Optional[int]):
    """Return (model name, head dict).  A path must be an .onnx file readable by onnx_ingest; a bare name is
    looked up among the pretrained files, or generated when synthetic weights were asked for."""
    if os.path.exists(path_or_name):
        if path_or_name.endswith(".tflite"):
            raise ValueError("The hip inference framework is selected, but tflite models were provided!")
        from . import onnx_ingest
        return os.path.splitext(os.path.basename(path_or_name))[0], onnx_ingest.load_head(path_or_name)
    key = path_or_name.replace(" ", "_")
    match = [p for p in get_pretrained_model_paths() if key in p.split(os.path.sep)[-1]]
    if not match:
        raise ValueError("Could not find pretrained model for model name '{}'".format(path_or_name))
    if os.path.exists(match[0]):
        from . import onnx_ingest
        return path_or_name, onnx_ingest.load_head(match[0])
    if synthetic_seed is None:
        raise ValueError(f"Pretrained model file {match[0]} does not exist (the reference downloads it at run "
                         "time, utils.py:625-673). Pass weights='synthetic' to use random-init weights of the "
                         "same architecture, or give the path of an .onnx file.")
    base = [k for k in MODELS if k in os.path.basename(match[0])][0]
    return path_or_name, W.synthetic_head(base, synthetic_seed)


def resolve_weights(weights: Union[str, dict, None]):
    """(seed, embedding or None, given heads) from the `weights` argument shared by Model, BatchedModel and utils.bulk_predict:
    'synthetic' -> seed 1234; a dict may carry 'embedding', 'heads', 'seed'; None -> real files only."""
    if weights is None:
        return None, None, {}
    if isinstance(weights, str):
        if weights != "synthetic":
            raise ValueError("weights must be 'synthetic', a dict {'embedding':..., 'heads':...} or None")
        return 1234, None, {}
    if isinstance(weights, dict):
        return weights.get("seed"), weights.get("embedding"), dict(weights.get("heads", {}))
    raise ValueError("weights must be 'synthetic', a dict {'embedding':..., 'heads':...} or None")


def _verify_melspectrogram(mel_path: str) -> None:
    """Hold a melspectrogram graph file to the analytic HIP front end (onnx_ingest.verify_melspectrogram).  A parameter FOUND to differ
    (window, hop, filter bank, amin, top_db ...) refuses.  A structure the verifier cannot follow (an exporter idiom it does not know)
    is not waved through: the file is then EVALUATED on probe audio (onnx_ingest.probe_melspectrogram) and must reproduce the recipe
    numerically, else it is refused too.  OWW_TRUST_MELSPECTROGRAM=1 is the only way past both checks."""
    if os.environ.get("OWW_TRUST_MELSPECTROGRAM") == "1":
        return
    from . import onnx_ingest
    hint = (".  The HIP front end computes the published recipe only; if this graph is known to be equivalent, set "
            "OWW_TRUST_MELSPECTROGRAM=1 and hold the result to tests/test_real_reference.py")
    try:
        onnx_ingest.verify_melspectrogram(mel_path)
    except onnx_ingest.GraphIdiomUnknown as e:
        try:
            onnx_ingest.probe_melspectrogram(mel_path)
        except ValueError as e2:
            raise ValueError(f"{e}; and the numeric check did not clear it either: {e2}{hint}") from e2
    except ValueError as e:
        raise ValueError(f"{e}{hint}") from e


def resolve_embedding(embedding: Optional[dict], seed: Optional[int], embedding_model_path: str = "",
                      melspec_model_path: str = "") -> dict:
    """The shared speech-embedding network (utils.py:90-93 loads embedding_model.onnx next to the wake-word models): the
    caller's weights, else the file the caller names (the reference's `embedding_model_path` / `melspec_model_path` keyword
    arguments, utils.py:38-44), else the real file in resources/models through onnx_ingest, else -- ONLY when synthetic weights
    were asked for -- the random-init network.  Never silently synthetic: real head files next to a random embedding would score
    nonsense."""
    for given in (embedding_model_path, melspec_model_path):
        if given and given.endswith(".tflite"):                         # utils.py:75-76, for this backend
            raise ValueError("The hip inference framework is selected, but tflite models were provided!")
        if given and not os.path.exists(given):
            raise ValueError(f"{given} does not exist")
    if embedding is not None and not embedding_model_path:
        if melspec_model_path:
            _verify_melspectrogram(melspec_model_path)
        return embedding
    path = embedding_model_path or FEATURE_MODELS["embedding"]["model_path"]
    if os.path.exists(path):
        from . import onnx_ingest
        # real model files: the log-mel front end of the HIP path is analytic, so the melspectrogram graph that sits next to the
        # embedding network is VERIFIED to be that recipe (or the construction fails loudly: onnx_ingest.verify_melspectrogram)
        mel_path = melspec_model_path or FEATURE_MODELS["melspectrogram"]["model_path"]
        if os.path.exists(mel_path):
            _verify_melspectrogram(mel_path)
        return onnx_ingest.load_embedding(path)
    if seed is not None:
        return W.synthetic_embedding(seed)
    raise ValueError(f"{path} does not exist; pass weights='synthetic' for random-init weights of the same architecture")


def fold_verifier(verifier):
    """(w, bias) of a custom verifier: the reference's scikit-learn pipeline FunctionTransformer(flatten) -> StandardScaler ->
    LogisticRegression (custom_verifier_model.py:95-113) folded into one affine map, or a ready (w, bias) pair."""
    if isinstance(verifier, (tuple, list)) and len(verifier) == 2:
        return np.asarray(verifier[0], np.float32).ravel(), float(verifier[1])
    steps = getattr(verifier, "steps", None)
    if not steps:
        raise ValueError("the device verifier needs the reference's pipeline (flatten -> StandardScaler -> LogisticRegression) or a (w, bias) pair")
    # exactly the reference's structure, nothing skipped in silence: [FunctionTransformer (the flatten)] [StandardScaler] classifier.
    # Any other transformer (PCA, a second scaler, ...) would fold into wrong scores: the caller must then keep the host-side hook
    # of the single-stream Model (model.py:320-328), which runs any pickled object.
    body = [st for _, st in steps]
    clf = body.pop()
    if body and type(body[0]).__name__ == "FunctionTransformer":
        body.pop(0)
    scaler = body.pop(0) if body and type(body[0]).__name__ == "StandardScaler" else None
    if body:
        raise ValueError("the device verifier folds only the reference's pipeline flatten -> StandardScaler -> LogisticRegression; "
                         f"unsupported steps: {[type(st).__name__ for st in body]} (use the host-side hook of openwakeword_amd.Model)")
    if not hasattr(clf, "coef_") or not hasattr(clf, "intercept_") or np.asarray(clf.coef_).shape[0] != 1:
        raise ValueError("the last step of the verifier pipeline must be a fitted binary LogisticRegression")
    coef = np.asarray(clf.coef_, np.float64)[0]
    b = float(np.asarray(clf.intercept_, np.float64)[0])
    if scaler is not None:
        # StandardScaler(with_mean=False) leaves mean_ = None, with_std=False leaves scale_ = None
        scale = np.ones_like(coef) if getattr(scaler, "scale_", None) is None else np.asarray(scaler.scale_, np.float64)
        mean = np.zeros_like(coef) if getattr(scaler, "mean_", None) is None else np.asarray(scaler.mean_, np.float64)
        if scale.shape != coef.shape or mean.shape != coef.shape:
            raise ValueError(f"verifier pipeline: scaler has {scale.size} features, the classifier {coef.size}")
        b -= float(np.sum(coef * mean / scale))
        coef = coef / scale
    return coef.astype(np.float32), b


class AudioFeatures:
    """Streaming half of openwakeword.utils.AudioFeatures on one device stream.

    `__call__(x)` buffers audio exactly like utils.py:409-452 and advances the device by whole 1280-sample
    chunks; `get_features(n)` reads the device feature ring (utils.py:454-460)."""

    feature_buffer_max_len = 120        # utils.py:170

    def __init__(self, engine: StreamEngine, stream: int = 0):
        self.engine = engine
        self.sid = int(stream)
        self._pending = np.empty(0, dtype=np.int16)     # samples appended to the raw buffer, not yet processed
        self.accumulated_samples = 0
        self.raw_data_remainder = np.empty(0, dtype=np.int16)
        self.last_scores: Optional[np.ndarray] = None
        self.reset()

    def _get_embeddings(self, x: np.ndarray) -> np.ndarray:
        """utils.py:210-241: mel of the whole clip, 76-row windows every 8 rows, one embedding each."""
        spec = self.engine.mel(np.asarray(x, dtype=np.int16)[None])[0] / 10.0 + 2.0       # utils.py:180,206
        n_win = (spec.shape[0] - 76) // 8 + 1
        if n_win < 1:
            return np.zeros((0, EMB_DIM), np.float32)
        out = self.engine.embed(spec[None, : 76 + 8 * (n_win - 1)].astype(np.float32))[0]
        return out

    # ---- the reference's stateless helpers (training / data-preparation code calls them directly), same shapes -------------
    def _get_melspectrogram(self, x, melspec_transform: Callable = lambda x: x / 10 + 2) -> np.ndarray:
        """utils.py:180-208: int16 samples `[n]` or `[B, n]` -> transformed mel rows `[F, 32]` / `[B, F, 32]`; one clamp floor
        for the whole call, like one run of the melspectrogram graph."""
        x = np.array(x).astype(np.int16) if isinstance(x, list) else np.asarray(x)
        if x.dtype != np.int16:
            raise ValueError("Input data must be 16-bit integers (i.e., 16-bit PCM audio)."
                             f"You provided {x.dtype} data.")
        x = x[None, ] if x.ndim < 2 else x
        return melspec_transform(np.squeeze(self.engine.mel(x)))

    def _batched(self, fn, x):
        cap = self.engine.n_streams_padded
        return np.concatenate([fn(x[o:o + cap]) for o in range(0, x.shape[0], cap)], axis=0)

    def _get_embeddings_from_melspec(self, melspec) -> np.ndarray:
        """utils.py:210-223: one 76-row window `[76, 32(, 1)]` (or a batch of them) -> its embedding, squeezed."""
        m = np.asarray(melspec, dtype=np.float32)
        if m.ndim >= 3 and m.shape[-1] == 1:
            m = m[..., 0]
        if m.ndim == 2:
            m = m[None, ]
        if m.ndim != 3 or m.shape[1:] != (76, 32):
            raise ValueError(f"expected melspectrogram windows of shape [76, 32, 1], got {np.asarray(melspec).shape}")
        return np.squeeze(self._batched(self.engine.embed, m))

    def _get_melspectrogram_batch(self, x, batch_size: int = 128, ncpu: int = 1) -> np.ndarray:
        """utils.py:243-290 (the CPU path: every clip is its own run of the melspectrogram graph, so every clip has its own
        clamp floor): int16 `[N, samples]` -> `[N, ceil(samples/160 - 3), 32]`, transformed (x/10 + 2)."""
        x = np.asarray(x)
        if x.ndim != 2 or x.dtype != np.int16:
            raise ValueError("Input data must be 16-bit integers (i.e., 16-bit PCM audio) of shape (N, samples)")
        return (self._batched(self.engine.mel_clips, x) / 10.0 + 2.0).astype(np.float32)

    def _get_embeddings_batch(self, x, batch_size: int = 128, ncpu: int = 1) -> np.ndarray:
        """utils.py:292-352: mel rows `[N, frames, 32(, 1)]` -> `[N, (frames - 76)//8 + 1, 96]`: 76-row windows every 8 rows;
        rows past the last whole window are ignored."""
        m = np.asarray(x, dtype=np.float32)
        if m.ndim == 4 and m.shape[-1] == 1:
            m = m[..., 0]
        if m.ndim != 3 or m.shape[2] != 32:
            raise ValueError(f"expected melspectrograms of shape (N, frames, 32), got {np.asarray(x).shape}")
        if m.shape[1] < 76:
            raise ValueError("Embedding model requires the input melspectrograms to have at least 76 frames")
        n_win = (m.shape[1] - 76) // 8 + 1
        return self._batched(self.engine.embed, np.ascontiguousarray(m[:, : 76 + 8 * (n_win - 1)]))

    def get_embedding_shape(self, audio_length: float, sr: int = 16000):
        """Shape of `_get_embeddings` for a clip of `audio_length` seconds (utils.py:238-241), from the frame arithmetic
        instead of a trial run: F = (n-512)//160 + 1 mel frames, (F-76)//8 + 1 windows."""
        n = int(audio_length * sr + 1e-6)           # (the float product of an exact sample count may land just below it)
        frames = (n - 512) // 160 + 1 if n >= 512 else 0
        return (max((frames - 76) // 8 + 1, 0), EMB_DIM)

    def embed_clips(self, x: np.ndarray, batch_size: int = 128, ncpu: int = 1) -> np.ndarray:
        """Embeddings of N equally long clips, `[N, samples] int16 -> [N, n_windows, 96]` (utils.py:354-385: mel per clip,
        76-row windows every 8 rows, the embedding model over all windows) -- one `oww_embed_clips` call per batch, PCM in,
        embeddings out, nothing else crosses the bus.  `ncpu` is accepted for signature compatibility; `batch_size` is
        capped by the handle's stream count.  Borrows the streaming state of the first streams: both the engine and this
        object are reset afterwards."""
        x = np.asarray(x)
        if x.ndim != 2 or x.dtype != np.int16:
            raise ValueError("embed_clips expects a 2-D int16 array [N, samples]")
        if x.shape[1] < 512 or (x.shape[1] - 512) // 160 + 1 < 76:          # utils.py:313-314
            raise ValueError("Embedding model requires the input melspectrograms to have at least 76 frames")
        cap = max(1, min(int(batch_size), self.engine.n_streams_padded))
        out = [self.engine.embed_clips(x[o:o + cap]) for o in range(0, x.shape[0], cap)]
        self.engine.reset()                 # borrowed streams back to the start-up state ...
        self.reset()                        # ... and this object's stream re-seeded like a fresh AudioFeatures
        return np.concatenate(out, axis=0) if out else np.zeros((0, 0, EMB_DIM), np.float32)

    def reset(self):
        """utils.py:172-178: the feature ring restarts from the embeddings of 4 s of random audio."""
        self._pending = np.empty(0, dtype=np.int16)
        self.accumulated_samples = 0
        self.raw_data_remainder = np.empty(0, dtype=np.int16)
        noise = np.random.randint(-1000, 1000, 16000 * 4).astype(np.int16)
        feats = self._get_embeddings(noise)                       # [41, 96]; clobbers stream state -> reset below
        ring = np.zeros((self.engine.feature_ring, EMB_DIM), np.float32)
        n = min(len(feats), ring.shape[0])
        ring[ring.shape[0] - n:] = feats[len(feats) - n:]
        self._n_features = n
        self.engine.reset([self.sid], ring)

    def __call__(self, x: np.ndarray) -> int:
        x = np.asarray(x)
        if x.dtype != np.int16:                                     # utils.py:195-197 (lists are cast, 194)
            raise ValueError("Input data must be 16-bit integers (i.e., 16-bit PCM audio)."
                             f"You provided {x.dtype} data.")
        # 1280-sample alignment (utils.py:413-430): samples beyond the last whole chunk wait in `raw_data_remainder`
        # -- but only once at least one whole chunk has accumulated; shorter input simply accumulates
        if self.raw_data_remainder.shape[0]:
            x = np.concatenate((self.raw_data_remainder, x))
            self.raw_data_remainder = np.empty(0, dtype=np.int16)
        total = self.accumulated_samples + x.shape[0]
        take = x.shape[0] - (total % CHUNK if total >= CHUNK else 0)
        self._pending = np.concatenate((self._pending, x[:take]))
        self.raw_data_remainder = x[take:]
        self.accumulated_samples += take

        n = self.accumulated_samples
        if n < CHUNK or n % CHUNK:
            return n                                             # nothing processed yet: the caller repeats its last scores
        k = n // CHUNK
        # one device step: mel over the k chunks (one clamp floor), k embeddings, heads per chunk, max over chunks.  A call
        # longer than max_chunks x 1280 samples (2.56 s by default; the reference takes any length, model.py:287-298) is
        # evaluated by the library in slices of max_chunks chunks that share the CALL's clamp floor (oww_step: one extra pass of
        # the mel kernel finds the call's maximum first), i.e. the reference's single run of the melspectrogram graph
        # (utils.py:387-401); calls beyond the ABI's limit (5.5 minutes in one predict()) are refused by it.
        self.last_scores = self.engine.step_raw(self._pending[None])[0]
        self._n_features = min(self._n_features + k, self.feature_buffer_max_len)
        self._pending = np.empty(0, dtype=np.int16)
        self.accumulated_samples = 0
        return n

    def get_features(self, n_feature_frames: int = 16, start_ndx: int = -1) -> np.ndarray:
        have = min(self._n_features, self.engine.feature_ring)
        buf = self.engine.get_features(self.sid, have)              # the reference's feature_buffer, oldest first
        if start_ndx != -1:                                         # utils.py:455-458
            end_ndx = start_ndx + int(n_feature_frames) if start_ndx + n_feature_frames != 0 else len(buf)
            return buf[start_ndx:end_ndx, :][None, ].astype(np.float32)
        return buf[int(-1 * n_feature_frames):, :][None, ].astype(np.float32)


def make_engine(n_streams: int, heads: dict, embedding: dict, use_mfma: Optional[int] = None, **kw) -> StreamEngine:
    """The engine behind Model / BatchedModel.  `use_mfma` None = the default fp16-split family (3), and -- when oww_commit refuses
    these weights for it (OWW_ERANGE: the commit-time comparison with the exact-fp32 kernels failed, i.e. the weights' range cannot be
    carried by f16 hi/lo pairs) -- the exact-fp32 family (1) instead, under a loud RuntimeWarning: a custom-trained model must never
    be unloadable.  An explicit value is taken as given (and an explicit 3 raises on such weights)."""
    from ._lib import OwwRangeError
    if use_mfma is not None:
        return StreamEngine(n_streams, heads, embedding, use_mfma=int(use_mfma), **kw)
    try:
        return StreamEngine(n_streams, heads, embedding, use_mfma=3, **kw)
    except OwwRangeError as e:
        import warnings
        warnings.warn(f"{e} -- falling back to the exact-fp32 kernel family (use_mfma=1): same results, about 2.5x slower (masked "
                      "steps -- predict_active, the fan-in server -- run full launches there)", RuntimeWarning)
        return StreamEngine(n_streams, heads, embedding, use_mfma=1, **kw)


class Model:
    """openwakeword.Model on the HIP library (one stream per object)."""

    def __new__(cls, *args, **kwargs):
        """`inference_framework="onnx"` / `"tflite"` (model.py:112-141) keep working where their runtimes exist: the call is
        handed to the reference package unchanged and ITS Model object is returned.  Where `openwakeword` (or the runtime it
        imports at module level, vad.py:48) is not importable this raises the reference's own kind of error (ValueError, cf.
        model.py:141) -- there is no silent fall-back to the HIP path or to anything else."""
        framework = kwargs.get("inference_framework", args[6] if len(args) > 6 else "hip")     # 7th positional parameter, as in the reference
        if framework == "hip":
            return super().__new__(cls)
        if framework not in ("onnx", "tflite"):
            raise ValueError(f"unknown inference_framework '{framework}': 'hip' (this package), or 'onnx' / 'tflite' "
                             "(handed to the reference package)")
        try:
            import openwakeword as reference_package
        except Exception as e:                       # ImportError of the package itself or of onnxruntime / tflite inside it
            raise ValueError(f"inference_framework='{framework}' is served by the reference package, which cannot be imported "
                             f"here ({type(e).__name__}: {e}); use inference_framework='hip'") from e
        passthrough = {k: v for k, v in kwargs.items() if k not in ("weights", "max_chunks", "vad_session", "use_mfma")
                       and not (k == "device" and not isinstance(v, str))}      # (the reference's own device is "cpu" / "gpu")
        return reference_package.Model(*args, **passthrough)

    def __init__(self, wakeword_models: List[str] = [], class_mapping_dicts: List[dict] = [],
                 enable_speex_noise_suppression: bool = False, vad_threshold: float = 0,
                 custom_verifier_models: dict = {}, custom_verifier_threshold: float = 0.1,
                 inference_framework: str = "hip", weights: Union[str, dict, None] = None, device: int = 0,
                 max_chunks: int = 32, wakeword_model_paths: Optional[List[str]] = None, vad_session=None,
                 use_mfma: Optional[int] = None, **kwargs):
        if wakeword_model_paths is not None:            # deprecated alias (model.py:37)
            wakeword_models = wakeword_model_paths
        seed, embedding, given_heads = resolve_weights(weights)
        wakeword_models = list(wakeword_models)
        if wakeword_models == [] and given_heads:
            wakeword_models = list(given_heads)
        if wakeword_models == []:
            wakeword_models = list(MODELS.keys())           # model.py:84-87: all pretrained models
        heads: Dict[str, dict] = {}
        for m in wakeword_models:
            if m in given_heads:
                heads[m] = given_heads[m]
            else:
                name, head = _load_head(m, seed)
                heads[name] = head
        # the keyword arguments the reference hands on to AudioFeatures (model.py:212, utils.py:38-44)
        unknown = sorted(set(kwargs) - {"melspec_model_path", "embedding_model_path", "sr", "ncpu"})
        if unknown:
            raise TypeError(f"AudioFeatures.__init__() got an unexpected keyword argument '{unknown[0]}'")
        if int(kwargs.get("sr", 16000)) != 16000:
            raise ValueError("the HIP path runs the reference's 16 kHz models; resample first (openwakeword_amd.resample)")
        if isinstance(device, str):                     # the reference's device is "cpu" / "gpu" (utils.py:43): this backend has one answer
            device = 0
        embedding = resolve_embedding(embedding, seed, kwargs.get("embedding_model_path", "") or "", kwargs.get("melspec_model_path", "") or "")

        self.models: Dict[str, dict] = heads
        self.model_inputs = {n: int(h["T"]) for n, h in heads.items()}          # model.py:156
        self.model_outputs = {n: int(h["n_out"]) for n, h in heads.items()}     # model.py:157
        self.class_mapping: Dict[str, dict] = {}
        for i, n in enumerate(heads):                                           # model.py:177-182
            if class_mapping_dicts and i < len(class_mapping_dicts) and class_mapping_dicts[i].get(n, None):
                self.class_mapping[n] = class_mapping_dicts[i]
            elif model_class_mappings.get(n, None):
                self.class_mapping[n] = model_class_mappings[n]
            else:
                self.class_mapping[n] = {str(j): str(j) for j in range(0, self.model_outputs[n])}

        self.custom_verifier_models: Dict[str, object] = {}
        self.custom_verifier_threshold = custom_verifier_threshold
        if isinstance(custom_verifier_models, dict):
            for n in heads:
                if custom_verifier_models.get(n, False):
                    self.custom_verifier_models[n] = pickle.load(open(custom_verifier_models[n], "rb"))
            if len(self.custom_verifier_models.keys()) < len(custom_verifier_models.keys()):     # model.py:189-195
                raise ValueError("Custom verifier models were provided, but some were not matched with a base model!"
                                 " Make sure that the keys provided in the `custom_verifier_models` dictionary argument"
                                 " exactly match that of the `.models` attribute of an instantiated openWakeWord Model object"
                                 " that has the same base models but doesn't have custom verifier models.")

        self.prediction_buffer: DefaultDict[str, deque] = defaultdict(partial(deque, maxlen=30))     # model.py:198
        if enable_speex_noise_suppression:                                       # model.py:201-205 (host library)
            from speexdsp_ns import NoiseSuppression
            self.speex_ns = NoiseSuppression.create(160, 16000)
        else:
            self.speex_ns = None
        self.vad_threshold = vad_threshold
        if vad_threshold > 0:                                                    # model.py:208-210
            from .vad import VAD
            self.vad = VAD(session=vad_session)         # raises ValueError without a network (silero_vad.onnx is not in the checkout)

        self._engine = make_engine(1, heads, embedding, use_mfma, device=device, max_chunks=max_chunks,
                                   feature_ring=AudioFeatures.feature_buffer_max_len)
        self._cols = self._engine.head_cols
        self.preprocessor = AudioFeatures(self._engine, 0)

    def close(self):
        self._engine.close()

    # ---- label bookkeeping ---------------------------------------------------------------------------------------
    def get_parent_model_from_label(self, label):
        """Model that owns a prediction label (model.py:215-224): a binary model's label is its own name, a multiclass
        model owns the values of its class mapping.  The last match wins, '' when nothing matches."""
        owner = ""
        for name, mapping in self.class_mapping.items():
            if label in mapping.values() or label == name:
                owner = name
        return owner

    def reset(self):
        """Model.reset (model.py:226-230): empty score history, fresh front-end state (new random feature-ring seed)."""
        self.prediction_buffer = defaultdict(partial(deque, maxlen=30))
        self.preprocessor.reset()

    def _suppress_noise_with_speex(self, x: np.ndarray, frame_size: int = 160):
        # host-side pre-filter exactly as in the reference (model.py:481-504): 10 ms frames through SpeexDSP
        pieces = (self.speex_ns.process(x[o:o + frame_size].tobytes()) for o in range(0, x.shape[0], frame_size))
        return np.frombuffer(b"".join(pieces), np.int16)

    # ---- one predict() call = the reference's model.py:232-386, split into its four phases ------------------------
    def _raw_outputs(self, name: str, n_prepared: int):
        """Per-model output vector of this call: device result when at least one 80 ms chunk was completed (already the
        maximum over chunks for longer calls, model.py:287-298), otherwise the previous prediction / zeros (299-307)."""
        lo, hi = self._cols[name]
        if n_prepared >= CHUNK:
            return self.preprocessor.last_scores[lo:hi]
        if self.model_outputs[name] == 1:
            history = self.prediction_buffer[name]
            return [history[-1] if len(history) else 0]
        return [0] * (max(int(k) for k in self.class_mapping[name]) + 1)

    def _apply_verifiers(self, scores: dict, name: str):
        # model.py:320-328: a pickled scikit-learn classifier re-scores frames the base model already likes
        for label, value in list(scores.items()):
            if value < self.custom_verifier_threshold:
                continue
            verifier = self.custom_verifier_models.get(self.get_parent_model_from_label(label), False)
            if verifier:
                scores[label] = verifier.predict_proba(self.preprocessor.get_features(self.model_inputs[name]))[0][-1]

    def _gate(self, scores: dict, n_prepared: int, patience: dict, threshold: dict, debounce_time: float):
        # model.py:340-359 -- the rules look at scores buffered BEFORE this call
        if not patience and not debounce_time > 0:
            return
        if threshold == {}:
            raise ValueError("Error! When using the `patience` argument, threshold "
                             "values must be provided via the `threshold` argument!")
        if patience != {} and debounce_time > 0:
            raise ValueError("Error! The `patience` and `debounce_time` arguments cannot be used together!")
        for label in scores:
            if scores[label] == 0.0:
                continue
            owner = self.get_parent_model_from_label(label)
            past = np.array(self.prediction_buffer[label])
            if owner in patience:
                need = patience[owner]
                if (past[-need:] >= threshold[owner]).sum() < need:
                    scores[label] = 0.0
            elif debounce_time > 0 and owner in threshold:
                span = int(np.ceil(debounce_time / (n_prepared / 16000)))
                if scores[label] >= threshold[owner] and (past[-span:] >= threshold[owner]).sum() > 0:
                    scores[label] = 0.0

    def predict(self, x: np.ndarray, patience: dict = {}, threshold: dict = {}, debounce_time: float = 0.0,
                timing: bool = False):
        if not isinstance(x, np.ndarray):
            raise ValueError(f"The input audio data (x) must by a Numpy array, instead received an object of type {type(x)}.")
        import time
        clock = {"models": {}}
        t0 = time.time()
        n_prepared = self.preprocessor(self._suppress_noise_with_speex(x) if self.speex_ns else x)
        clock["models"]["preprocessor"] = time.time() - t0          # mel, embedding AND every head run in this one device call

        scores: dict = {}
        for name in self.models:
            t1 = time.time()
            out = self._raw_outputs(name, n_prepared)
            if self.model_outputs[name] == 1:
                scores[name] = out[0]
            else:
                scores.update({cls: out[int(idx)] for idx, cls in self.class_mapping[name].items()})
            if self.custom_verifier_models:
                self._apply_verifiers(scores, name)
            for label in scores:                                    # model.py:331-333: warm-up, first five frames
                if len(self.prediction_buffer[label]) < 5:
                    scores[label] = 0.0
            clock["models"][name] = time.time() - t1
        self._gate(scores, n_prepared, patience, threshold, debounce_time)
        for label, value in scores.items():                         # model.py:362-363
            self.prediction_buffer[label].append(value)
        if self.vad_threshold > 0:                                  # model.py:366-381: after the ring append, on the raw x
            t2 = time.time()
            self.vad(x)
            clock["models"]["vad"] = time.time() - t2
            from .vad import gate_value
            if gate_value(self.vad.prediction_buffer, self.vad_threshold):
                for label in scores:
                    scores[label] = 0.0
        return (scores, clock) if timing else scores

    # ---- conveniences on top of predict() --------------------------------------------------------------------------
    @staticmethod
    def _read_wav(path: str) -> np.ndarray:
        with wave.open(path, mode="rb") as f:
            return np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)

    def predict_clip(self, clip: Union[str, np.ndarray], padding: int = 1, chunk_size=1280, **kwargs):
        """Stream a whole clip through predict() (model.py:388-426): `padding` seconds of silence either side, calls of
        `chunk_size` samples; like the reference the final partial chunk (and one full one) is not fed."""
        data = self._read_wav(clip) if isinstance(clip, str) else clip
        if padding:
            silence = np.zeros(16000 * padding, dtype=np.int16)
            data = np.concatenate((silence, data, silence))
        starts = range(0, data.shape[0] - chunk_size, chunk_size)
        return [self.predict(data[o:o + chunk_size], **kwargs) for o in starts]

    def _get_positive_prediction_frames(self, file: str, threshold: float = 0.5, return_type: str = "features", **kwargs):
        """Features (or 4 s of audio) behind every frame scoring >= threshold (model.py:428-479; false-positive mining)."""
        data = self._read_wav(file)
        found = defaultdict(list)
        for o in range(0, data.shape[0] - CHUNK, CHUNK):
            for label, value in self.predict(data[o:o + CHUNK], **kwargs).items():
                if value < threshold:
                    continue
                if return_type == "features":
                    found[label].append(self.preprocessor.get_features(self.model_inputs[self.get_parent_model_from_label(label)]))
                elif return_type == "audio":
                    context = data[max(0, o - 3 * 16000):o + 16000]
                    if len(context) == 4 * 16000:
                        found[label].append(context)
        return {label: np.vstack(v) for label, v in found.items()}


class BatchedModel:
    """S independent streams behind one object: `predict_batch(pcm[S, 1280*k])` -> fp32 scores [S, n_labels].

    This is the many-stream form the reference lacks (one Python object per stream, utils.py:502-536 forks
    processes for scale-out).  Post-processing (first-5 zeroing, patience, debounce, 30-deep score ring) runs
    on the device per stream; `labels` names the score columns in `Model.predict`'s key order."""

    def __init__(self, n_streams: int, wakeword_models: Sequence[str], weights: Union[str, dict, None] = None,
                 device: int = 0, max_chunks: int = 1, hip_stream: int = 0, vad_weights: Optional[dict] = None,
                 vad_threshold: float = 0.0, use_mfma: Optional[int] = None, calibration_pcm="default",
                 embedding_model_path: str = "", melspec_model_path: str = ""):
        # same weight resolution as Model: real .onnx files (heads AND the shared embedding network) unless synthetic
        # weights are asked for explicitly -- never a random-init embedding under real heads
        seed, emb, given = resolve_weights(weights)
        heads = {}
        for m in wakeword_models:
            if m in given:
                heads[m] = given[m]
            else:
                name, head = _load_head(m, seed)
                heads[name] = head
        emb = resolve_embedding(emb, seed, embedding_model_path, melspec_model_path)       # (as Model: utils.py:38-44)
        # vad_weights: the on-device voice-activity network (weights.synthetic_vad layout) -- BASELINE configs[4]: network + gate
        # fused into every step.  With a gate asked for and no weights given, a silero_vad.onnx next to the package is ingested
        # when its graph is the architecture the kernels implement (onnx_ingest.load_vad); a graph it refuses, or no file, leaves
        # the gate to externally pushed scores (push_vad) -- said loudly, never a silently different network.
        if vad_weights is None and vad_threshold > 0:
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources", "models", "silero_vad.onnx")
            if os.path.exists(path):
                from . import onnx_ingest
                try:
                    vad_weights = onnx_ingest.load_vad(path)
                except Exception as e:               # refused by name (ValueError) or unreadable in a way the reader did not foresee
                    import warnings
                    warnings.warn(f"{e} -- the VAD gate of this BatchedModel waits for push_vad() scores", RuntimeWarning)
        # calibration_pcm: audio of the deployment's domain for the commit-time calibration / self-test of the fp16-split kernels
        # (StreamEngine); "default" = speech shipped with the package
        self.engine = make_engine(n_streams, heads, emb, use_mfma, device=device, max_chunks=max_chunks, hip_stream=hip_stream,
                                  vad=vad_weights, vad_threshold=vad_threshold, calibration_pcm=calibration_pcm)
        self.labels: List[str] = []
        self._keep: List[int] = []
        col = 0
        for n, h in heads.items():
            if h["n_out"] == 1:
                self.labels.append(n)
                self._keep.append(col)
            else:
                for int_label, cls in model_class_mappings.get(n, {str(j): str(j) for j in range(h["n_out"])}).items():
                    self.labels.append(cls)
                    self._keep.append(col + int(int_label))
            col += h["n_out"]
        self._parent = {}
        col = 0
        for n, h in heads.items():
            for j in range(h["n_out"]):
                self._parent[col + j] = n
            col += h["n_out"]
        self.n_streams = n_streams

    def set_postproc(self, patience: dict = {}, threshold: dict = {}, debounce_time: float = 0.0, chunk_samples: int = 1280):
        """Same rules and errors as Model.predict's keyword arguments (model.py:340-359), applied to every stream."""
        if patience != {} or debounce_time > 0:
            if threshold == {}:
                raise ValueError("Error! When using the `patience` argument, threshold "
                                 "values must be provided via the `threshold` argument!")
            if patience != {} and debounce_time > 0:
                raise ValueError("Error! The `patience` and `debounce_time` arguments cannot be used together!")
        NL = self.engine.n_labels
        pat = [int(patience.get(self._parent[c], 0)) for c in range(NL)]
        thr = [float(threshold.get(self._parent[c], np.nan)) for c in range(NL)]
        frames = int(np.ceil(debounce_time / (chunk_samples / 16000))) if debounce_time > 0 else 0
        self.engine.set_postproc(pat, thr, frames)

    def reset(self, stream_ids: Optional[Sequence[int]] = None, init_features: Optional[np.ndarray] = None,
              reset_vad: bool = False):
        """Model.reset for the listed streams (None = all).  Like the reference (model.py:226-230) this leaves the VAD state
        alone; pass reset_vad=True when a stream slot is handed to a NEW caller, so that it does not inherit the previous
        caller's voice-activity history."""
        self.engine.reset(stream_ids, init_features)
        if reset_vad:
            self.engine.reset_vad(stream_ids)

    def predict_batch(self, pcm: np.ndarray, sample_rate: int = 16000) -> np.ndarray:
        """`sample_rate` != 16000: every stream's message is converted on the device first (engine.resample; e.g. 640 samples at 8 kHz,
        3840 at 48 kHz or 3528 at 44.1 kHz make one 1280-sample chunk)."""
        if not isinstance(pcm, np.ndarray):
            raise ValueError(f"The input audio data (x) must by a Numpy array, instead received an object of type {type(pcm)}.")
        if int(sample_rate) != 16000:
            pcm = self.engine.resample(pcm, sample_rate)
        return self.engine.step(pcm)[:, self._keep]

    def predict_active(self, pcm: np.ndarray, active) -> np.ndarray:
        """predict_batch for the streams flagged in `active` ([n_streams] bool) only; the others do not advance at all and
        repeat their previous scores (the reference's per-client behaviour: no audio, no predict() call)."""
        if not isinstance(pcm, np.ndarray):
            raise ValueError(f"The input audio data (x) must by a Numpy array, instead received an object of type {type(pcm)}.")
        return self.engine.step_masked(pcm, active)[:, self._keep]

    def set_custom_verifier(self, model_name: str, verifier, threshold: float = 0.1) -> None:
        """`Model(custom_verifier_models=..., custom_verifier_threshold=...)` for every stream, on the device (model.py:320-328).
        `verifier` is the scikit-learn pipeline custom_verifier_model.train_verifier_model returns (flatten -> StandardScaler ->
        LogisticRegression; custom_verifier_model.py:95-113), or an already folded (w [T*96], bias) pair, or None to remove it.
        Anything else -- an arbitrary object with predict_proba -- cannot run on the device: use the single-stream Model, which
        keeps the reference's host-side hook."""
        if model_name not in self._parent.values():
            raise ValueError("Custom verifier models were provided, but some were not matched with a base model!")
        cols = [c for c, n in self._parent.items() if n == model_name]
        if verifier is None:
            for c in cols:
                self.engine.set_verifier(c, None)
            return
        w, b = fold_verifier(verifier)
        for c in cols:
            self.engine.set_verifier(c, w, b, threshold)

    def set_vad_threshold(self, threshold: float) -> None:
        """The VAD gate of `Model(vad_threshold=...)` (model.py:366-381) for every stream; 0 switches it off."""
        self.engine.set_vad_threshold(threshold)

    def push_vad(self, vad_scores: np.ndarray) -> None:
        """Voice-activity score of each stream for the frame about to be predicted (the value `VAD.__call__` appends to its
        ring, vad.py:98-130); the network that produces it is the caller's (see openwakeword_amd.vad)."""
        self.engine.push_vad(vad_scores)

    def submit_batch(self, pcm: np.ndarray) -> None:
        """Pipelined `predict_batch` for host-fed serving: enqueue one step (upload on its own stream) and return; the
        scores come back from `collect_batch()` in submission order, at most two steps in flight.  Page-locked buffers
        (`engine.pinned_empty`) make the upload overlap the previous step's kernels."""
        if not isinstance(pcm, np.ndarray):
            raise ValueError(f"The input audio data (x) must by a Numpy array, instead received an object of type {type(pcm)}.")
        self.engine.submit(pcm)

    def collect_batch(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        return self.engine.collect(out)[:, self._keep]

    def close(self):
        self.engine.close()
