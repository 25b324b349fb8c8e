"""
TEST / BASELINE INFRASTRUCTURE ONLY -- torch-CPU (oneDNN) port of the reference's per-frame algorithm.

Same arithmetic as `oracle.oww_oracle` (which stays the parity oracle), but expressed with
torch.nn.functional.conv2d / matmul on CPU so that the `cpu_baseline` leg of bench.py times something
close to what a tuned CPU runtime (onnxruntime's MLAS, the reference's backend: utils.py:84-93) achieves,
instead of numpy's slow batched matmuls.  It executes the REFERENCE's algorithm, not the GPU's:
  * mel as a dense windowed-DFT matmul over 257 bins (what melspectrogram.onnx's two Conv1d do), per
    stream clamp;
  * the full 76x32 window through all 20 conv layers every frame (utils.py:437-443) -- no incremental reuse;
  * one MLP evaluation per head (model.py:299-302).
tests/test_oracle_golden.py::test_torch_cpu_port_matches_numpy_oracle checks it against the numpy oracle.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as TF

from . import oww_oracle as O


class TorchCpuPort:
    def __init__(self, emb: dict, heads: dict, threads: int | None = None):
        if threads:
            torch.set_num_threads(int(threads))
        self.threads = torch.get_num_threads()
        t = O._tables(np.float32)
        self.dft = torch.from_numpy(np.concatenate([t.re, t.im], axis=1).copy())        # [512, 514]
        self.fb = torch.from_numpy(t.fb.copy())                                           # [257, 32]
        self.convs = []
        for li, (kh, kw, ci, co, relu_first, bn, pool) in enumerate(O.CNN_LAYERS):
            w = torch.from_numpy(np.ascontiguousarray(emb["conv"][li].transpose(3, 2, 0, 1)))   # OIHW
            sc = sh = None
            if bn:
                s, b = O.bn_fold(*emb["bn"][li])
                sc, sh = torch.from_numpy(s).view(1, -1, 1, 1), torch.from_numpy(b).view(1, -1, 1, 1)
            self.convs.append((w, (0, (kw - 1) // 2), relu_first, sc, sh, pool))
        self.heads = heads
        self._ht = {k: self._head_tensors(h) for k, h in heads.items()}

    @staticmethod
    def _head_tensors(h):
        def net(n):
            tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
            d = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in n.items()}
            if n.get("ln1") is not None:
                d["ln1"] = tuple(tt(a) for a in n["ln1"])
            d["blocks"] = [(tt(w), tt(b), None if ln is None else tuple(tt(a) for a in ln)) for w, b, ln in O.net_blocks(n)]
            return d
        out = {"net": net(h["net"])}
        if "net2" in h:
            out["net2"] = net(h["net2"])
        return out

    @torch.no_grad()
    def mel(self, pcm: np.ndarray) -> torch.Tensor:
        """int16 [B, n] -> transformed mel rows [B, F, 32]; clamp floor per stream (streaming semantics)."""
        x = torch.from_numpy(pcm.astype(np.float32))
        fr = x.unfold(1, O.N_FFT, O.HOP)                                   # [B, F, 512]
        z = fr @ self.dft
        p = z[..., :257] ** 2 + z[..., 257:] ** 2
        m = p @ self.fb
        db = 10.0 * torch.log(torch.clamp(m, min=O.AMIN)) / np.log(10.0)
        db = torch.maximum(db, db.amax(dim=(1, 2), keepdim=True) - O.TOP_DB)
        return db / 10 + 2

    @torch.no_grad()
    def embed(self, windows: torch.Tensor) -> torch.Tensor:
        """[B, 76, 32] -> [B, 96]"""
        h = windows.unsqueeze(1)                                           # NCHW, H = time, W = mel
        for w, pad, relu_first, sc, sh, pool in self.convs:
            h = TF.conv2d(h, w, padding=pad)
            if relu_first:
                h = torch.relu(h)
            if sc is not None:
                h = h * sc + sh
                h = torch.maximum(torch.maximum(h * float(O.LEAK), h), torch.tensor(float(O.FLOOR)))
            if pool:
                h = TF.max_pool2d(h, pool)
        return h.reshape(h.shape[0], -1)

    @torch.no_grad()
    def head(self, name: str, feats: torch.Tensor) -> torch.Tensor:
        hd, t = self.heads[name], self._ht[name]

        def mlp(n):
            x = feats.reshape(feats.shape[0], -1) @ n["w1"] + n["b1"]
            if n.get("ln1") is not None:
                x = TF.layer_norm(x, (x.shape[1],), n["ln1"][0], n["ln1"][1], O.LN_EPS)
            for w, b, ln in n["blocks"]:
                x = torch.relu(x) @ w + b
                if ln is not None:
                    x = TF.layer_norm(x, (x.shape[1],), ln[0], ln[1], O.LN_EPS)
            return torch.relu(x) @ n["w3"] + n["b3"]

        if hd["kind"] == "multiclass":
            return torch.softmax(torch.relu(mlp(t["net"])), dim=1)
        s = torch.sigmoid(mlp(t["net"]))
        if hd["kind"] == "gated":
            s = torch.where(s > 0.5, torch.sigmoid(mlp(t["net2"])), s)
        return s

    @torch.no_grad()
    def frame(self, pcm1760: np.ndarray, mel_ring: torch.Tensor, feat_ring: torch.Tensor):
        """One reference frame for B streams: returns (scores [B, n_heads], new mel ring, new feature ring)."""
        rows = self.mel(pcm1760)                                           # [B, 8, 32]
        mel_ring = torch.cat([mel_ring[:, 8:], rows], dim=1)               # last 76 rows
        e = self.embed(mel_ring)
        feat_ring = torch.cat([feat_ring[:, 1:], e.unsqueeze(1)], dim=1)
        scores = torch.cat([self.head(k, feat_ring[:, -self.heads[k]["T"]:]) for k in self.heads], dim=1)
        return scores, mel_ring, feat_ring
