"""Test infrastructure: the reference's three networks as torch modules, written to `.onnx` by PyTorch's OWN TorchScript exporter --
the tool the reference exports its wake-word models with (train.py:144-165) and the melspectrogram graph came out of (notebook
cell 15).  Used by tests/test_onnx_ingest.py (the reader against a writer that is not ours), tests/test_oracle_golden.py (the
oracle's stage math against torch.nn in float64), tests/golden/make_golden_onnx.py (the reference's own code on these files) and the
by-path GPU tests.

The `onnx` package is not installed here; the exporter needs it only for a post-pass that splices onnx-script functions into the
finished ModelProto bytes (none occur in these models), so `export` replaces that post-pass by the identity while it runs."""
import io
import os
import sys
import warnings

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from openwakeword_amd import weights as W          # noqa: E402
from oracle import oww_oracle as O                # noqa: E402


def export(module, example, path, opset, **kw):
    """torch.onnx.export(module, example, path, opset_version=opset, **kw) with the TorchScript exporter."""
    import torch
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    keep = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    try:
        buf = io.BytesIO()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(module.eval(), example, buf, opset_version=opset, dynamo=False, **kw)
    finally:
        onnx_proto_utils._add_onnxscript_fn = keep
    with open(path, "wb") as f:
        f.write(buf.getvalue())


def torch_export_head(module, T, path, opset):
    """As train.py:144-165 calls it: a random [1, T, 96] example, one named output."""
    import torch
    export(module, torch.rand(T, 96)[None, ], path, opset, output_names=["out"])


def torch_head(net, T, n_out, n_blocks=None):
    """The architecture of train.py:56-83 (flatten, Linear + LayerNorm + ReLU, blocks of the same, Linear, Sigmoid | ReLU) with the
    given weights; multiclass models end in ReLU and are exported under a softmax wrapper (train.py:152-165)."""
    import torch
    import torch.nn as nn

    class Block(nn.Module):
        def __init__(self, w, b, ln):
            super().__init__()
            self.fc = nn.Linear(w.shape[0], w.shape[1])
            self.norm = nn.LayerNorm(w.shape[1]) if ln is not None else nn.Identity()      # (the catalogue's multiclass heads have none)
            self.act = nn.ReLU()
            with torch.no_grad():
                self.fc.weight.copy_(torch.from_numpy(w.T.copy())); self.fc.bias.copy_(torch.from_numpy(b))
                if ln is not None:
                    self.norm.weight.copy_(torch.from_numpy(ln[0])); self.norm.bias.copy_(torch.from_numpy(ln[1]))

        def forward(self, x):
            return self.act(self.norm(self.fc(x)))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.flatten = nn.Flatten()
            self.first = Block(net["w1"], net["b1"], net["ln1"])
            # (n_blocks = None: the network's own hidden blocks, weights.net_blocks; a number: that many copies of block 0)
            blocks = W.net_blocks(net) if n_blocks is None else [(net["w2"], net["b2"], net["ln2"])] * n_blocks
            self.blocks = nn.ModuleList([Block(w, b, ln) for w, b, ln in blocks])
            self.last = nn.Linear(net["w3"].shape[0], n_out)
            self.last_act = nn.Sigmoid() if n_out == 1 else nn.ReLU()
            with torch.no_grad():
                self.last.weight.copy_(torch.from_numpy(net["w3"].T.copy())); self.last.bias.copy_(torch.from_numpy(net["b3"]))

        def forward(self, x):
            x = self.first(self.flatten(x))
            for blk in self.blocks:
                x = blk(x)
            return self.last_act(self.last(x))

    if n_out == 1:
        return Net()

    class Wrapped(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Net()

        def forward(self, x):
            return torch.nn.functional.softmax(self.model(x), dim=1)

    return Wrapped()


def torch_rnn_head(head):
    """The reference's model_type "rnn" (train.py:85-98), with the given weights: `out, h = nn.LSTM(96, 64, num_layers=2,
    bidirectional=True, batch_first=True)(x)`; `Sigmoid | ReLU (Linear(128, n_classes)(out[:, -1]))`; a multiclass model is exported
    under the softmax wrapper of train.py:152-165."""
    import torch
    import torch.nn as nn
    n_out, H = int(head["n_out"]), W.RNN_HID

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.layer1 = nn.LSTM(W.EMB_DIM, H, num_layers=2, bidirectional=True, batch_first=True, dropout=0.0)
            self.layer2 = nn.Linear(H * 2, n_out)
            self.layer3 = nn.Sigmoid() if n_out == 1 else nn.ReLU()

        def forward(self, x):
            out, h = self.layer1(x)
            return self.layer3(self.layer2(out[:, -1]))

    m = Net()
    with torch.no_grad():
        for li in range(2):
            n_in = W.EMB_DIM if li == 0 else 2 * H
            for d in range(2):
                w, b = head["lstm"][li][d]
                sfx = f"_l{li}" + ("_reverse" if d else "")
                getattr(m.layer1, "weight_ih" + sfx).copy_(torch.from_numpy(np.ascontiguousarray(w[:n_in].T)))
                getattr(m.layer1, "weight_hh" + sfx).copy_(torch.from_numpy(np.ascontiguousarray(w[n_in:].T)))
                getattr(m.layer1, "bias_ih" + sfx).copy_(torch.from_numpy(b * np.float32(0.25)))       # (b = b_ih + b_hh: split unevenly on purpose)
                getattr(m.layer1, "bias_hh" + sfx).copy_(torch.from_numpy(b - b * np.float32(0.25)))
        m.layer2.weight.copy_(torch.from_numpy(np.ascontiguousarray(head["w_out"].T)))
        m.layer2.bias.copy_(torch.from_numpy(head["b_out"]))
    if n_out == 1:
        return m

    class M(nn.Module):                                 # train.py:152-165
        def __init__(self):
            super().__init__()
            self.model = m

        def forward(self, x):
            return torch.nn.functional.softmax(self.model(x), dim=1)

    return M()


def torch_embedding(emb, act="leakyclamp"):
    """The speech-embedding CNN (notebook cell 18, restated in oracle/oww_oracle.py: CNN_LAYERS) as a torch module in NCHW, input
    [B, 76, 32, 1] permuted once -- for PyTorch's exporter, which folds eval-mode BatchNorm into a preceding convolution."""
    import torch
    import torch.nn as nn

    class Act(nn.Module):
        def forward(self, x):
            if act == "leakyclamp":
                return torch.clamp(torch.nn.functional.leaky_relu(x, 0.2), min=-0.4)
            return torch.maximum(torch.maximum(x * 0.2, x), torch.tensor(-0.4))

    layers = []
    for li, (kh, kw, ci, co, relu_first, bn, pool) in enumerate(O.CNN_LAYERS):
        conv = nn.Conv2d(ci, co, (kh, kw), padding=(0, (kw - 1) // 2), bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(np.ascontiguousarray(emb["conv"][li].transpose(3, 2, 0, 1))))      # HWIO -> OIHW
        layers.append(conv)
        if relu_first:
            layers.append(nn.ReLU())
        if bn:
            g, b, m, v = emb["bn"][li]
            norm = nn.BatchNorm2d(co, eps=1e-3)
            with torch.no_grad():
                norm.weight.copy_(torch.from_numpy(g)); norm.bias.copy_(torch.from_numpy(b))
                norm.running_mean.copy_(torch.from_numpy(m)); norm.running_var.copy_(torch.from_numpy(v))
            layers += [norm, Act()]
        if pool:
            layers.append(nn.MaxPool2d(pool))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(*layers)

        def forward(self, x):                                   # [B, 76, 32, 1] -> [B, 1, 1, 96], the reference graph's interface
            return self.body(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)

    return Net().eval()


def torch_melspectrogram(top_db=80.0, hop=160):
    """A torch module with the structure of torchlibrosa's Spectrogram + LogmelFilterBank as the notebook patches them (cell 15):
    two Conv1d with window x cos / -sin kernels (center=False), real^2 + imag^2, matmul with melW, power_to_db with log_spec.max()."""
    import torch
    import torch.nn as nn
    n = np.arange(512, dtype=np.float64)
    win = np.zeros(512)
    win[56:456] = W.hann_window().astype(np.float64)
    ang = 2.0 * np.pi * np.outer(np.arange(257), n) / 512

    class Mel(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv_real = nn.Conv1d(1, 257, 512, stride=hop, bias=False)
            self.conv_imag = nn.Conv1d(1, 257, 512, stride=hop, bias=False)
            with torch.no_grad():
                self.conv_real.weight.copy_(torch.from_numpy((win * np.cos(ang))[:, None, :].astype(np.float32)))
                self.conv_imag.weight.copy_(torch.from_numpy((-win * np.sin(ang))[:, None, :].astype(np.float32)))
            self.melW = nn.Parameter(torch.from_numpy(W.mel_filterbank().astype(np.float32)), requires_grad=False)

        def forward(self, x):                                   # [B, samples] -> [B, 1, frames, 32]
            x = x[:, None, :]
            real = self.conv_real(x)[:, None, :, :].transpose(2, 3)
            imag = self.conv_imag(x)[:, None, :, :].transpose(2, 3)
            spec = real ** 2 + imag ** 2
            mel = torch.matmul(spec, self.melW)
            log_spec = 10.0 * torch.log10(torch.clamp(mel, min=1e-10, max=float("inf")))
            log_spec = log_spec - 10.0 * float(np.log10(max(1e-10, 1.0)))
            return torch.maximum(log_spec, log_spec.max() - top_db)       # (the patched power_to_db: the call's maximum)

    return Mel().eval()


def torch_gated(head, form="where"):
    """docs/models/hey_jarvis.md:38: a first network for every frame, a second (verifier) network whose score replaces the first's
    where the first exceeds 0.5, 'combined together prior to exporting' -- as torch.where, or as a scripted `if` (an ONNX If)."""
    import torch
    import torch.nn as nn
    first, second = torch_head(head["net"], head["T"], 1), torch_head(head["net2"], head["T"], 1)
    if form == "where":
        class Gated(nn.Module):
            def __init__(self):
                super().__init__()
                self.first, self.second = first, second

            def forward(self, x):
                a = self.first(x)
                return torch.where(a > 0.5, self.second(x), a)
        return Gated().eval()

    class GatedIf(nn.Module):
        def __init__(self):
            super().__init__()
            self.first, self.second = first, second

        def forward(self, x):
            a = self.first(x)
            if bool(a[0, 0] > 0.5):
                a = self.second(x)
            return a
    return torch.jit.script(GatedIf().eval())


def torch_vad(vad):
    """The voice-activity STAND-IN (oracle/vad_standin.py; the interface is Silero's, vad.py:92-130) as a torch module with the
    inputs the reference feeds -- input [B, 640] (samples / 32767), sr (int64), h, c [2, B, 64] -- and outputs (out [B, 1], hn, cn):
    |STFT| as two strided Conv1d (periodic Hann(256), hop 64, bins 1..128) -> log(1 + 50 |X|) -> four Conv1d(k=3, pad 1) + ReLU ->
    nn.LSTM(64, 64, num_layers=2) -> ReLU -> Linear(64, 1) -> Sigmoid -> mean over time."""
    import torch
    import torch.nn as nn
    from oracle import vad_standin as VS
    n = np.arange(VS.N_FFT, dtype=np.float64)
    ang = 2.0 * np.pi * np.outer(np.arange(1, VS.N_BINS + 1), n) / VS.N_FFT
    win = VS.hann_periodic()

    class Vad(nn.Module):
        def __init__(self):
            super().__init__()
            self.re = nn.Conv1d(1, VS.N_BINS, VS.N_FFT, stride=VS.HOP, bias=False)
            self.im = nn.Conv1d(1, VS.N_BINS, VS.N_FFT, stride=VS.HOP, bias=False)
            self.enc = nn.ModuleList([nn.Conv1d(ci, co, 3, stride=st, padding=1) for ci, co, st in VS.ENC])
            self.lstm = nn.LSTM(VS.HID, VS.HID, num_layers=2, batch_first=True)
            self.dec = nn.Linear(VS.HID, 1)
            with torch.no_grad():
                self.re.weight.copy_(torch.from_numpy((win * np.cos(ang))[:, None, :].astype(np.float32)))
                self.im.weight.copy_(torch.from_numpy((-win * np.sin(ang))[:, None, :].astype(np.float32)))
                for conv, (w, b) in zip(self.enc, vad["enc"]):
                    conv.weight.copy_(torch.from_numpy(np.ascontiguousarray(w.transpose(2, 1, 0))))      # [k, cin, cout] -> [cout, cin, k]
                    conv.bias.copy_(torch.from_numpy(b))
                H = VS.HID
                for layer, (w, b) in enumerate(vad["lstm"]):                  # rows (x ; h), columns i | f | g | o: torch's own order
                    getattr(self.lstm, f"weight_ih_l{layer}").copy_(torch.from_numpy(np.ascontiguousarray(w[:H].T)))
                    getattr(self.lstm, f"weight_hh_l{layer}").copy_(torch.from_numpy(np.ascontiguousarray(w[H:].T)))
                    getattr(self.lstm, f"bias_ih_l{layer}").copy_(torch.from_numpy(b))
                    getattr(self.lstm, f"bias_hh_l{layer}").zero_()
                self.dec.weight.copy_(torch.from_numpy(vad["dec"][0][None, :].copy()))
                self.dec.bias.fill_(float(vad["dec"][1]))

        def forward(self, x, sr, h, c):
            x = x * (sr == 16000).to(x.dtype)                  # (the rate input stays part of the graph, as in the real file)
            x = x[:, None, :]
            re, im = self.re(x), self.im(x)
            a = torch.log(1.0 + 50.0 * torch.sqrt(re * re + im * im))
            for conv in self.enc:
                a = torch.relu(conv(a))
            y, (hn, cn) = self.lstm(a.transpose(1, 2), (h, c))
            out = torch.sigmoid(self.dec(torch.relu(y))).mean(dim=1)
            return out, hn, cn

    return Vad().eval()


def export_vad(vad, path, opset=13):
    import torch
    export(torch_vad(vad), (torch.rand(1, 640) * 0.1, torch.tensor(16000), torch.zeros(2, 1, 64), torch.zeros(2, 1, 64)), path, opset,
           input_names=["input", "sr", "h", "c"], output_names=["output", "hn", "cn"])


def export_reference_files(directory, weights, head_opsets=None, embedding_opset=13, mel_opset=12):
    """Write melspectrogram.onnx, embedding_model.onnx and one <name>.onnx per head of `weights` = {"embedding", "heads"} (binary,
    multiclass and gated heads; file name = the head's key) into `directory`; returns {name: path}."""
    import torch
    head_opsets = head_opsets or {}
    paths = {}
    for name, head in weights["heads"].items():
        paths[name] = os.path.join(directory, f"{name}.onnx")
        if head["kind"] == "rnn":                           # train.py's model_type "rnn" (train.py:85-98)
            module = torch_rnn_head(head)
        elif head["kind"] == "gated":                       # hey_jarvis-style routing: "<name>_if" = the scripted branch (an ONNX If)
            module = torch_gated(head, "if" if name.endswith("_if") else "where")
        else:
            module = torch_head(head["net"], head["T"], head["n_out"])
        torch_export_head(module, head["T"], paths[name], head_opsets.get(name, 13))
    paths["embedding_model"] = os.path.join(directory, "embedding_model.onnx")
    export(torch_embedding(weights["embedding"]), torch.rand(1, 76, 32, 1), paths["embedding_model"], embedding_opset,
           input_names=["input_1"], dynamic_axes={"input_1": {0: "batch"}})
    paths["melspectrogram"] = os.path.join(directory, "melspectrogram.onnx")
    export(torch_melspectrogram(), torch.rand(1, 1760) * 1000, paths["melspectrogram"], mel_opset, input_names=["input"],
           dynamic_axes={"input": {0: "batch", 1: "samples"}})
    return paths
