"""
Bulk feature extraction on the device: the `openwakeword.utils` functions that sit on `AudioFeatures.embed_clips`
(SURVEY.md section 8f rank 1).  `compute_features_from_generator` is the training pipeline's feature writer
(/root/reference/openwakeword/utils.py:542-601): clips from a generator -> embeddings -> one `.npy` file of shape
(N, n_windows, 96) float32 written through a memory map, trimmed to the rows actually produced
(/root/reference/openwakeword/data.py:856-892).
"""
from __future__ import annotations

import os
from typing import Iterator, Optional

import numpy as np
from numpy.lib.format import open_memmap

from .model import AudioFeatures  # noqa: F401  (re-export: the reference exposes it from this module)
from .engine import EMB_DIM, StreamEngine


def trim_mmap(mmap_path: str, n_rows: Optional[int] = None) -> int:
    """Drop the unused rows at the end of a feature file (data.py:856-892: the rows after the last one that is not all
    zero, or everything after `n_rows` when the writer knows its count).  The file is rewritten through a second map in
    blocks, then renamed over the original.  Returns the number of rows kept."""
    src = np.load(mmap_path, mmap_mode="r")
    if n_rows is None:
        n_rows = src.shape[0]
        while n_rows > 0 and not src[n_rows - 1].any():
            n_rows -= 1
    n_rows = int(min(max(n_rows, 0), src.shape[0]))
    if n_rows == src.shape[0]:
        return n_rows
    tmp = mmap_path[:-4] + ".trim.npy" if mmap_path.endswith(".npy") else mmap_path + ".trim"
    dst = open_memmap(tmp, mode="w+", dtype=src.dtype, shape=(n_rows,) + src.shape[1:])
    for lo in range(0, n_rows, 1024):
        dst[lo:min(lo + 1024, n_rows)] = src[lo:min(lo + 1024, n_rows)]
    dst.flush()
    del dst, src
    os.replace(tmp, mmap_path)
    return n_rows


def _embedding_weights(weights) -> dict:
    """The embedding model's parameters: the caller's ({"embedding": ...} as for `Model`, or the path of an embedding_model.onnx),
    else the .onnx file where the reference keeps it, else (opt-in, weights="synthetic") random-init ones."""
    from . import weights as W
    from .model import FEATURE_MODELS
    if isinstance(weights, dict):
        return weights["embedding"] if "embedding" in weights else weights
    path = FEATURE_MODELS["embedding"]["model_path"]
    if isinstance(weights, str) and weights != "synthetic":
        if not os.path.exists(weights):
            raise ValueError(f"{weights} does not exist")
        path = weights
    if os.path.exists(path):
        from . import onnx_ingest
        return onnx_ingest.load_embedding(path)
    if weights == "synthetic":
        return W.synthetic_embedding(1234)
    raise ValueError(f"{path} does not exist; pass weights='synthetic' for random-init weights of the same architecture")


def compute_features_from_generator(generator: Iterator[np.ndarray], n_total: int, clip_duration: int, output_file: str,
                                    device: str = "gpu", ncpu: int = 1, engine: Optional[StreamEngine] = None,
                                    weights=None) -> int:
    """utils.py:542-601 with the embedding work on the MI355X.

    `generator` yields int16 arrays [batch, clip_duration]; the first batch fixes the batch size (ValueError when it
    exceeds `n_total`, utils.py:581-583); rows beyond `n_total` are dropped; the file is trimmed to the rows written.
    `device` / `ncpu` exist for signature compatibility (there is no CPU path here).  `engine`: an existing
    `StreamEngine` to run on; otherwise one is created with as many streams as the batch.  `weights`: None = the embedding model
    file where the reference keeps it; the path of an embedding_model.onnx; a weight dict as for `Model`; 'synthetic' opts into
    random-init weights when the file is absent.  Returns the rows written."""
    n_cols = AudioFeatures.get_embedding_shape(None, clip_duration / 16000)
    if n_cols[0] < 1:
        raise ValueError("clips are shorter than one 76-frame embedding window (12512 samples, 782 ms)")
    fp = open_memmap(output_file, mode="w+", dtype=np.float32, shape=(int(n_total), n_cols[0], n_cols[1]))
    own = None
    rows = 0
    try:
        first = True
        for audio in generator:
            audio = np.asarray(audio)
            if first:
                first = False
                if audio.shape[0] > n_total:
                    raise ValueError(f"The value of 'n_total' ({n_total}) is less than the batch size ({audio.shape[0]})."
                                     " Please increase 'n_total' to be >= batch size.")
                if engine is None:
                    own = engine = StreamEngine(max(int(audio.shape[0]), 1), {}, _embedding_weights(weights))
            if rows >= n_total:
                break
            if audio.dtype != np.int16 or audio.ndim != 2:
                raise ValueError("the generator must yield 2-D int16 arrays [batch, samples]")
            cap = engine.n_streams_padded
            for lo in range(0, audio.shape[0], cap):
                feats = engine.embed_clips(audio[lo:lo + cap])[: n_total - rows]
                fp[rows:rows + feats.shape[0]] = feats
                rows += feats.shape[0]
                if rows >= n_total:
                    break
            fp.flush()
    finally:
        del fp
        if own is not None:
            own.close()
        elif engine is not None:
            engine.reset()
    trim_mmap(output_file, rows)
    return rows


def bulk_predict(file_paths, wakeword_models, prediction_function: str = "predict_clip", ncpu: int = 1,
                 inference_framework: str = "hip", **kwargs) -> dict:
    """`openwakeword.utils.bulk_predict` (utils.py:466-536) with the GPU as the pool of workers.

    The reference forks `ncpu` processes, each streaming its share of the files through one `Model`; here every file is one
    stream of a `BatchedModel` and all files advance together, 80 ms per device step (equivalent to the reference with one
    fresh process per file: no state is carried from one clip to the next).  `predict_clip` keyword arguments (`padding`,
    `chunk_size` -- a multiple of 1280 --, `patience`, `threshold`, `debounce_time`) and `Model` keyword arguments
    (`weights`, `device`) are taken from **kwargs like the reference filters them by name.  Other prediction functions, or
    chunk sizes that are not whole 80 ms frames, run file by file through a single `Model`.
    Returns {file path: value of the prediction function}, for `predict_clip` a list of {label: score} per chunk."""
    import wave
    from .model import BatchedModel, Model
    if inference_framework != "hip":
        raise ValueError("openwakeword_amd only provides inference_framework='hip'")
    file_paths = list(file_paths)
    model_kw = {k: v for k, v in kwargs.items() if k in ("weights", "device", "class_mapping_dicts", "vad_threshold", "vad_session",
                                                         "custom_verifier_models", "custom_verifier_threshold",
                                                         "enable_speex_noise_suppression")}
    chunk = int(kwargs.get("chunk_size", 1280))
    batched_ok = prediction_function == "predict_clip" and chunk % 1280 == 0 and chunk > 0 and \
        not (set(model_kw) - {"weights", "device"})
    if not batched_ok:
        mdl = Model(wakeword_models=wakeword_models, **model_kw)
        try:
            fn = getattr(mdl, prediction_function)
            names = fn.__code__.co_varnames
            takes_kw = bool(fn.__code__.co_flags & 0x08)
            out = {}
            for f in file_paths:
                out[f] = fn(f, **{k: v for k, v in kwargs.items() if (k in names or takes_kw) and k not in model_kw})
                mdl.reset()
            return out
        finally:
            mdl.close()

    padding = int(kwargs.get("padding", 1))
    clips = []
    for f in file_paths:
        with wave.open(f, mode="rb") as w:
            data = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        if padding:
            z = np.zeros(16000 * padding, dtype=np.int16)
            data = np.concatenate((z, data, z))
        clips.append(data)
    if not clips:
        return {}
    k = chunk // 1280
    n_calls = [len(range(0, c.shape[0] - chunk, chunk)) for c in clips]              # model.py:421-426
    S = len(clips)
    bm = BatchedModel(S, list(wakeword_models), weights=model_kw.get("weights"),
                      device=int(model_kw.get("device", 0)), max_chunks=k)
    try:
        seed_stream = AudioFeatures(bm.engine, 0)                                     # seeds a feature ring like a fresh Model
        init = seed_stream.get_features(bm.engine.feature_ring)[0]
        ring = np.zeros((bm.engine.feature_ring, EMB_DIM), np.float32)
        ring[ring.shape[0] - init.shape[0]:] = init
        bm.reset(None, ring)
        bm.set_postproc(kwargs.get("patience", {}), kwargs.get("threshold", {}), float(kwargs.get("debounce_time", 0.0)), chunk)
        results = [[] for _ in range(S)]
        pcm = np.zeros((S, chunk), np.int16)
        for t in range(max(n_calls)):
            pcm[:] = 0
            for s, c in enumerate(clips):
                if t < n_calls[s]:
                    pcm[s] = c[t * chunk:(t + 1) * chunk]
            scores = bm.predict_batch(pcm)
            for s in range(S):
                if t < n_calls[s]:
                    results[s].append({lab: scores[s, i] for i, lab in enumerate(bm.labels)})
        return {f: results[s] for s, f in enumerate(file_paths)}
    finally:
        bm.close()
