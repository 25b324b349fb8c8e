"""Test infrastructure: the engine interface the host shim (openwakeword_amd/model.py: AudioFeatures, Model) calls, served by the CPU
ORACLE instead of libowwhip.so -- so that the host logic (1280-alignment and carry-over, first-five zeroing, patience / debounce,
class mapping, verifier hook, VAD gate, chunk slicing) runs in the CPU test tier against the vectors the reference's own code
produced.  Never part of the product: the package refuses to run without the HIP library (tests/test_abi_cpu.py)."""
import numpy as np

from oracle import oww_oracle as O

CHUNK = 1280


class OracleEngine:
    """One stream.  step_raw / reset / get_features / mel / embed with the semantics of openwakeword_amd.engine.StreamEngine."""

    def __init__(self, n_streams, heads, embedding, max_chunks=32, feature_ring=120, **_):
        assert n_streams == 1
        self.heads, self.emb = heads, embedding
        self.n_streams, self.n_streams_padded = 1, 32
        self.max_chunks, self.feature_ring = int(max_chunks), int(feature_ring)
        self.head_cols, col = {}, 0
        for name, h in heads.items():
            self.head_cols[name] = (col, col + int(h["n_out"]))
            col += int(h["n_out"])
        self.n_labels = col
        self._oaf = O.OracleAudioFeatures(embedding, init_noise=np.zeros(64000, np.int16))
        self.closed = False

    # ---- stateless stage entries
    def mel(self, pcm):
        pcm = np.atleast_2d(pcm)
        if pcm.dtype != np.int16:
            raise ValueError(f"Input data must be 16-bit integers (i.e., 16-bit PCM audio). You provided {pcm.dtype} data.")
        return O.mel_stage(pcm.astype(np.float32))[:, 0]

    def mel_clips(self, pcm):
        return np.stack([self.mel(p[None])[0] for p in np.atleast_2d(pcm)])

    def embed(self, mel_rows):
        m = np.asarray(mel_rows, np.float32)
        m = m[None] if m.ndim == 2 else m
        return O.embedding_stage(m[..., None], self.emb)[:, :, 0, :]

    def embed_clips(self, pcm):
        return np.stack([self.embed((self.mel(p[None])[0] / 10.0 + 2.0)[: 76 + 8 * (((len(p) - 512) // 160 + 1 - 76) // 8)])[0] for p in pcm])

    # ---- streaming state
    def reset(self, stream_ids=None, init_features=None):
        self._oaf.reset()
        ring = np.zeros((self.feature_ring, 96), np.float32) if init_features is None else np.asarray(init_features, np.float32)
        self._oaf.features = ring.copy()

    def step_raw(self, pcm):
        pcm = np.asarray(pcm)
        assert pcm.shape[0] == 1 and pcm.shape[1] % CHUNK == 0        # (any number of chunks: oww_step slices longer calls itself)
        k = pcm.shape[1] // CHUNK
        assert self._oaf(pcm[0]) == pcm.shape[1]
        out = np.full(self.n_labels, -np.inf, np.float32)
        for back in range(k - 1, -1, -1):                                   # model.py:287-298: one evaluation per chunk, the maximum
            feats = self._oaf.features
            for name, h in self.heads.items():
                T = int(h["T"])
                f = feats[len(feats) - T - back: len(feats) - back][None].astype(np.float32)
                lo, hi = self.head_cols[name]
                out[lo:hi] = np.maximum(out[lo:hi], O.head_stage(f, h).reshape(-1))
        return out[None]

    def get_features(self, sid, T):
        return self._oaf.features[-int(T):].astype(np.float32) if T else np.zeros((0, 96), np.float32)

    def close(self):
        self.closed = True
