"""
Read the reference's model files (`*.onnx`, /root/reference/openwakeword/__init__.py:8-51) into the numpy weight dicts
that `openwakeword_amd.engine` packs for the HIP library -- without the `onnx` package: a ~100-line protobuf
wire-format reader for the handful of ModelProto / GraphProto / NodeProto / TensorProto / AttributeProto fields needed.

Three graph families are recognised by structure, not by node names:

* wake-word heads  (`openwakeword/train.py:56-83`, `docs/models/*.md`):  [Flatten] -> Gemm/MatMul(T*96 -> H) [+Add]
  -> [LayerNormalization] -> Relu -> Gemm(H -> H) -> [LN] -> Relu -> Gemm(H -> n_out) -> Sigmoid | (Relu, Softmax).
  Two such chains in one file = the gated form of hey_jarvis (`docs/models/hey_jarvis.md:38`).
* embedding model (`notebooks/converting_google_speech_embedding_model.ipynb` cell 18): 20 Conv nodes (OIHW weights),
  19 BatchNormalization nodes -- or biases on the Conv nodes where an exporter folded the BatchNorm.
* melspectrogram: analytic in this package (the kernel cannot load another recipe); `verify_melspectrogram` checks that the file
  computes exactly that recipe -- window, hop, DFT, power, filter bank, 10 log10, amin, top_db -- or refuses.

No real model file exists in this environment, so the structural assumptions are pinned by tests/test_onnx_ingest.py:
two independent writers -- one in the plain idiom, one emitting the idioms tf2onnx (embedding model: NHWC <-> NCHW Transposes, Pad
nodes, BatchNorm folded into Conv bias or written as Mul / Add, the activation as Max(Mul(x, 0.2), x) -> Max(., -0.4) or as
LeakyRelu + Clip) and torch.onnx.export (heads: Gemm transB, opset-17 LayerNormalization or its decomposition, the hey_jarvis gate
as Greater + Where or as an If with subgraphs) produce -- must load to the same weights.  The loaders walk the DATAFLOW and check
every operator they pass against what the kernels compute; anything else raises ValueError naming the operator.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

from . import weights as W


# ------------------------------------------------------------------------------------------- protobuf wire format
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field number, wire type, value) of one message; value = int for varint/fixed, bytes for length-delimited."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _packed_ints(v, wt) -> List[int]:
    if wt == 0:
        return [v]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


_DTYPES = {1: np.float32, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _tensor(buf: bytes):
    dims: List[int] = []
    dtype, name, raw, floats, int64s = 1, "", None, [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_ints(v, wt)
        elif fno == 2:
            dtype = v
        elif fno == 4:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fno == 7:
            int64s += _packed_ints(v, wt)
        elif fno == 8:
            name = v.decode()
        elif fno == 9:
            raw = v
    if dtype not in _DTYPES:
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, dtype=_DTYPES[dtype]).copy()
    elif floats:
        arr = np.asarray(floats, dtype=np.float32)
    else:
        arr = np.asarray(int64s, dtype=np.int64)
    return name, arr.reshape([d for d in dims]) if dims else arr


def _attribute(buf: bytes):
    name, val = "", None
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = v.decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = v if v < (1 << 63) else v - (1 << 64)
        elif fno == 4:
            val = v.decode(errors="replace")
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 6:
            val = _graph(v)                                     # a subgraph (If then_branch / else_branch)
        elif fno == 7:
            val = (val or []) + (list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]])
        elif fno == 8:
            val = (val or []) + [x if x < (1 << 63) else x - (1 << 64) for x in _packed_ints(v, wt)]
    return name, val


def _graph(buf: bytes) -> dict:
    """One GraphProto (the model's, or an If branch): initializers (Constant nodes included), nodes, input / output names."""
    inits: Dict[str, np.ndarray] = {}
    nodes, inputs, outputs = [], [], []
    shapes: Dict[str, list] = {}
    for fno, wt, v in _fields(buf):
        if fno == 5:
            name, arr = _tensor(v)
            if arr is not None:
                inits[name] = arr
        elif fno in (11, 12):                                   # ValueInfoProto: name = field 1, type = field 2
            vname = ""
            for f2, _w2, v2 in _fields(v):
                if f2 == 1:
                    vname = v2.decode()
                    (inputs if fno == 11 else outputs).append(vname)
                elif f2 == 2:                                   # TypeProto.tensor_type (1) . shape (2) . dim (1) . dim_value (1)
                    dims = []
                    for f3, _w3, v3 in _fields(v2):
                        if f3 != 1:
                            continue
                        for f4, _w4, v4 in _fields(v3):
                            if f4 != 2:
                                continue
                            for f5, _w5, v5 in _fields(v4):
                                if f5 == 1:
                                    dv = [x for f6, _w6, x in _fields(v5) if f6 == 1]
                                    dims.append(int(dv[0]) if dv else None)        # (a symbolic dimension: None)
                    shapes[vname] = dims
        elif fno == 1:
            node = {"op": "", "name": "", "inputs": [], "outputs": [], "attrs": {}}
            for f2, w2, v2 in _fields(v):
                if f2 == 1:
                    node["inputs"].append(v2.decode())
                elif f2 == 2:
                    node["outputs"].append(v2.decode())
                elif f2 == 3:
                    node["name"] = v2.decode()
                elif f2 == 4:
                    node["op"] = v2.decode()
                elif f2 == 5:
                    k, a = _attribute(v2)
                    node["attrs"][k] = a
            if node["op"] == "Constant" and node["outputs"] and isinstance(node["attrs"].get("value"), np.ndarray):
                inits[node["outputs"][0]] = node["attrs"]["value"]
                continue
            nodes.append(node)
    return {"initializers": inits, "nodes": nodes, "inputs": [i for i in inputs if i not in inits], "outputs": outputs, "shapes": shapes}


def load_graph(path: str) -> dict:
    """{'initializers': {name: ndarray}, 'nodes': [{'op', 'name', 'inputs', 'outputs', 'attrs'}], 'inputs', 'outputs'} of an ONNX
    file.  Constant nodes become initializers; shape-only operators on constants (Transpose / Reshape / Identity / Cast / Squeeze /
    Unsqueeze of an initializer -- a weight stored HWIO behind a Transpose, a scalar behind a Cast) are folded away; subgraph
    attributes (If branches) are graphs of the same form under attrs[...]."""
    with open(path, "rb") as f:
        data = f.read()
    graph = None
    try:
        for fno, wt, v in _fields(data):
            if fno == 7 and wt == 2:
                graph = v
        g = _graph(graph) if graph is not None else None
    except (IndexError, struct.error, UnicodeDecodeError) as e:
        raise ValueError(f"{path}: not a readable ONNX ModelProto ({type(e).__name__}: {e})") from e
    if g is None:
        raise ValueError(f"{path}: no GraphProto found (not an ONNX ModelProto?)")
    _fold_constants(g)
    return g


def _fold_constants(g: dict) -> None:
    inits = g["initializers"]
    keep = []
    for n in g["nodes"]:
        for a in n["attrs"].values():
            if isinstance(a, dict) and "nodes" in a:
                a["initializers"] = dict(inits, **a["initializers"])     # a branch sees the outer scope's constants
                _fold_constants(a)
        x = inits.get(n["inputs"][0]) if n["inputs"] else None
        out = None
        if x is not None and all(i in inits or i == "" for i in n["inputs"]):
            if n["op"] == "Transpose":
                out = np.transpose(x, n["attrs"].get("perm") or list(range(x.ndim))[::-1])
            elif n["op"] in ("Identity", "Cast"):
                out = x if n["op"] == "Identity" else x.astype(_DTYPES.get(n["attrs"].get("to", 1), np.float32))
            elif n["op"] == "Reshape" and len(n["inputs"]) > 1:
                out = x.reshape([int(d) if d != 0 else x.shape[i] for i, d in enumerate(inits[n["inputs"][1]].tolist())])
            elif n["op"] in ("Squeeze", "Unsqueeze"):
                axes = n["attrs"].get("axes")
                if axes is None and len(n["inputs"]) > 1:
                    axes = inits[n["inputs"][1]].tolist()
                if n["op"] == "Squeeze":
                    out = np.squeeze(x, axis=tuple(axes) if axes else None)
                elif axes:
                    out = x
                    for ax in sorted(axes):
                        out = np.expand_dims(out, ax)
        if out is not None and n["outputs"]:
            inits[n["outputs"][0]] = np.ascontiguousarray(out)
        else:
            keep.append(n)
    g["nodes"] = keep


class GraphIdiomUnknown(ValueError):
    """A graph walker met a structure it does not know -- as opposed to a parameter it FOUND and that differs from what the kernels
    compute (a plain ValueError).  `verify_melspectrogram` callers may downgrade this one to a warning (model.resolve_embedding)."""


class _Flow:
    """Dataflow view of a graph: who consumes a tensor, who produced it.  The loaders walk the DATAFLOW (never node order or node
    names), interpret every operator they pass and refuse -- naming the operator and where it sits -- what they cannot interpret."""

    def __init__(self, g: dict, path: str):
        self.g, self.path, self.inits = g, path, g["initializers"]
        self.consumers: Dict[str, List[dict]] = {}
        self.producer: Dict[str, dict] = {}
        for n in g["nodes"]:
            for i in n["inputs"]:
                if i and i not in self.inits:
                    self.consumers.setdefault(i, []).append(n)
            for o in n["outputs"]:
                self.producer[o] = n

    def refuse(self, why: str, unknown: bool = False):
        """unknown = the walker lost the thread (a structure it does not know) as opposed to having FOUND a value that differs."""
        from collections import Counter
        raise (GraphIdiomUnknown if unknown else ValueError)(
            f"{self.path}: {why}; operators in the file: {dict(Counter(n['op'] for n in self.g['nodes']))}")

    def const(self, n: dict, skip: str = None):
        """The constant operand of a binary node (the input that is an initializer), or None."""
        c = [self.inits[i] for i in n["inputs"] if i in self.inits and i != skip]
        return c[0] if len(c) == 1 else None

    def data_input(self, n: dict) -> str:
        d = [i for i in n["inputs"] if i and i not in self.inits]
        return d[0] if d else ""

    def start(self) -> str:
        ins = [i for i in self.g["inputs"] if i in self.consumers]
        if not ins:                                              # files without ValueInfo: the tensor nobody produces
            ins = [t for t in self.consumers if t not in self.producer]
        if len(ins) != 1:
            self.refuse(f"expected one data input, found {ins}")
        return ins[0]


def _scalar(c) -> float:
    return float(np.asarray(c).reshape(-1)[0]) if c is not None and np.asarray(c).size == 1 else float("nan")


# ------------------------------------------------------------------------------------------- wake-word heads
LN_EPS = 1e-5        # torch.nn.LayerNorm's default (train.py:63-64): the constant the head kernels apply


def _walk_layernorm(fl: _Flow, cur: str, hidden: int, where: str):
    """LayerNormalization over the last axis starting at tensor `cur`: the opset-17 operator, or its decomposition
    (ReduceMean, Sub, Pow | Mul, ReduceMean, Add eps, Sqrt, Div, Mul gamma, Add beta).  Returns ((gamma, beta), output tensor),
    or (None, cur) when `cur` is not normalised."""
    cons = fl.consumers.get(cur, [])
    if len(cons) == 1 and cons[0]["op"] == "LayerNormalization":
        n = cons[0]
        eps = float(n["attrs"].get("epsilon", 1e-5))
        if abs(eps - LN_EPS) > 1e-9:
            fl.refuse(f"{where}: LayerNormalization epsilon {eps:g}, the head kernels apply {LN_EPS:g}")
        if int(n["attrs"].get("axis", -1)) not in (-1, 1):
            fl.refuse(f"{where}: LayerNormalization over axis {n['attrs'].get('axis')}")
        gm = fl.inits.get(n["inputs"][1]) if len(n["inputs"]) > 1 else None
        bt = fl.inits.get(n["inputs"][2]) if len(n["inputs"]) > 2 else None
        if gm is None or gm.shape != (hidden,):
            fl.refuse(f"{where}: LayerNormalization without a constant scale of {hidden} values")
        return (gm.astype(np.float32), np.zeros(hidden, np.float32) if bt is None else bt.astype(np.float32)), n["outputs"][0]
    if not any(c["op"] == "ReduceMean" for c in cons):
        return None, cur
    # decomposed form: collect the sub-graph between `cur` and the tensor the activation consumes
    seen, frontier, ops, gamma, beta, eps, out = set(), [cur], [], None, None, None, None
    while frontier:
        t = frontier.pop()
        for n in fl.consumers.get(t, []):
            if id(n) in seen:
                continue
            if n["op"] in ("Relu", "Gemm", "MatMul", "Sigmoid", "Softmax"):
                out = t
                continue
            seen.add(id(n))
            ops.append(n["op"])
            c = fl.const(n)
            if n["op"] == "Mul" and c is not None and c.shape == (hidden,):
                gamma = c
            elif n["op"] == "Add" and c is not None and c.shape == (hidden,):
                beta = c
            elif n["op"] == "Add" and c is not None and c.size == 1:
                eps = _scalar(c)
            elif n["op"] == "ReduceMean":
                axes = n["attrs"].get("axes") or ([] if len(n["inputs"]) < 2 else fl.inits[n["inputs"][1]].tolist())
                if [int(a) for a in axes] not in ([-1], [1]):
                    fl.refuse(f"{where}: ReduceMean over axes {axes} inside a LayerNorm")
            elif n["op"] == "Pow" and abs(_scalar(c) - 2.0) > 0:
                fl.refuse(f"{where}: Pow exponent {_scalar(c)} inside a LayerNorm")
            frontier += n["outputs"]
    want = {"ReduceMean": 2, "Sub": 1, "Sqrt": 1, "Div": 1}
    from collections import Counter
    cnt = Counter(ops)
    ok = all(cnt[k] == v for k, v in want.items()) and cnt["Pow"] + cnt["Mul"] in (1, 2) and cnt["Add"] in (1, 2) and \
        set(cnt) <= {"ReduceMean", "Sub", "Sqrt", "Div", "Pow", "Mul", "Add"}
    if not ok or eps is None or out is None:
        fl.refuse(f"{where}: operators {dict(cnt)} after a linear layer are not a LayerNorm over the last axis")
    if abs(eps - LN_EPS) > 1e-9:
        fl.refuse(f"{where}: LayerNorm epsilon {eps:g}, the head kernels apply {LN_EPS:g}")
    gm = np.ones(hidden, np.float32) if gamma is None else gamma.astype(np.float32)
    bt = np.zeros(hidden, np.float32) if beta is None else beta.astype(np.float32)
    return (gm, bt), out


def _walk_linear(fl: _Flow, n: dict, where: str):
    """(weight [in, out], bias [out], output tensor) of a Gemm, or of a MatMul and the Add that follows it."""
    if n["op"] == "Gemm":
        w = fl.inits.get(n["inputs"][1])
        if w is None or w.ndim != 2:
            fl.refuse(f"{where}: Gemm without a constant 2-D weight")
        at = n["attrs"]
        if float(at.get("alpha", 1.0)) != 1.0 or float(at.get("beta", 1.0)) != 1.0 or int(at.get("transA", 0)) != 0:
            fl.refuse(f"{where}: Gemm with alpha / beta / transA = {at.get('alpha', 1.0)} / {at.get('beta', 1.0)} / {at.get('transA', 0)}")
        w = w.T if int(at.get("transB", 0)) else w
        b = fl.inits.get(n["inputs"][2]) if len(n["inputs"]) > 2 and n["inputs"][2] else None
        if len(n["inputs"]) > 2 and n["inputs"][2] and b is None:
            fl.refuse(f"{where}: Gemm bias is not a constant")
        out = n["outputs"][0]
    else:
        w = fl.inits.get(n["inputs"][1])
        if w is None or w.ndim != 2:
            fl.refuse(f"{where}: MatMul without a constant 2-D right operand")
        b, out = None, n["outputs"][0]
        cons = fl.consumers.get(out, [])
        if len(cons) == 1 and cons[0]["op"] == "Add" and fl.const(cons[0]) is not None and fl.const(cons[0]).shape == (w.shape[1],):
            b, out = fl.const(cons[0]), cons[0]["outputs"][0]
    bias = np.zeros(w.shape[1], np.float32) if b is None else np.asarray(b, np.float32).reshape(-1)
    if bias.shape != (w.shape[1],):
        fl.refuse(f"{where}: bias of {bias.shape[0]} values on a layer with {w.shape[1]} outputs")
    return np.ascontiguousarray(w, np.float32), bias, out


MAX_HEAD_BLOCKS = 8     # OWW_MAX_HEAD_BLOCKS of include/owwhip.h


def _walk_net(fl: _Flow, first: dict, where: str):
    """One MLP of train.py:56-83 from its first linear node: Linear -> [LN] -> Relu, n_blocks times the same (train.py:73; one block
    in every released model), a last Linear, then the output activation(s).  A linear layer followed by [LN] + Relu + another linear
    layer is a hidden layer; the first one that is not is the output layer.  Returns (net dict in the layout of weights.net_blocks,
    tail operator names, last tensor)."""
    net, node, hidden_layers = {}, first, []
    while True:
        li = len(hidden_layers) + 1
        w, b, cur = _walk_linear(fl, node, f"{where} layer {li}")
        # hidden layer?  [LayerNorm] -> exactly one Relu -> exactly one linear layer
        nxt = None
        try_ln = _peek_hidden(fl, cur, w.shape[1], f"{where} layer {li}")
        if try_ln is not None:
            ln, relu_out = try_ln
            cons = fl.consumers.get(relu_out, [])
            if len(cons) == 1 and cons[0]["op"] in ("Gemm", "MatMul"):
                nxt = cons[0]
        if nxt is None:
            net["w3"], net["b3"] = w, b
            break
        hidden_layers.append((w, b, ln))
        if len(hidden_layers) > 1 + MAX_HEAD_BLOCKS:
            fl.refuse(f"{where}: more than {MAX_HEAD_BLOCKS} hidden blocks behind the first layer (train.py's Net with n_blocks > "
                      f"{MAX_HEAD_BLOCKS}); the head kernels take at most that many")
        node = nxt
    if not hidden_layers:
        fl.refuse(f"{where}: a single linear layer, no hidden layer (expected Linear -> [LN] -> Relu -> ... -> Linear)")
    if len({ln is None for _w, _b, ln in hidden_layers}) != 1:
        fl.refuse(f"{where}: LayerNorm after some hidden layers but not the others")
    (net["w1"], net["b1"], net["ln1"]), blocks = hidden_layers[0], hidden_layers[1:]
    net["w2"], net["b2"], net["ln2"] = blocks[0] if blocks else (None, None, None)
    if len(blocks) > 1:
        net["more"] = [{"w": w, "b": b, "ln": ln} for w, b, ln in blocks[1:]]
    hidden = net["w1"].shape[1]
    for k, (w, _b, _ln) in enumerate(blocks):
        if w.shape != (hidden, hidden):
            fl.refuse(f"{where}: hidden block {k} maps {w.shape[0]} -> {w.shape[1]} units, the first layer has {hidden} (train.py's "
                      "blocks are square)")
    if net["w3"].shape[0] != hidden:
        fl.refuse(f"{where}: the output layer reads {net['w3'].shape[0]} units, the hidden layers have {hidden}")
    tail = []
    while True:
        cons = fl.consumers.get(cur, [])
        if len(cons) == 1 and cons[0]["op"] in ("Relu", "Sigmoid", "Softmax", "Tanh", "LeakyRelu", "Elu", "Selu", "Gelu", "HardSigmoid", "PRelu", "Clip"):
            if cons[0]["op"] == "Softmax" and int(cons[0]["attrs"].get("axis", -1)) not in (-1, 1):
                fl.refuse(f"{where}: Softmax over axis {cons[0]['attrs'].get('axis')}")
            tail.append(cons[0]["op"])
            cur = cons[0]["outputs"][0]
        else:
            return net, tail, cur


def _peek_hidden(fl: _Flow, cur: str, width: int, where: str):
    """(ln, tensor behind the Relu) if tensor `cur` -- a linear layer's output -- goes through an optional LayerNorm and exactly one
    Relu, as a hidden layer of train.py:56-83 does; None if it does not (the output layer).  A LayerNorm that is there but is not the
    one the kernels apply is refused by _walk_layernorm, not skipped."""
    ln, out = _walk_layernorm(fl, cur, width, where)               # (None, cur) when `cur` is not normalised
    cons = fl.consumers.get(out, [])
    if len(cons) == 1 and cons[0]["op"] == "Relu":
        return ln, cons[0]["outputs"][0]
    if ln is not None:
        fl.refuse(f"{where}: expected exactly one Relu after the LayerNorm, found {[c['op'] for c in cons]}")
    return None


GATE_THRESHOLD = 0.5   # docs/models/hey_jarvis.md:38: the verifier network replaces the score where the first network is > 0.5


def _shape_only(fl: _Flow, t: str) -> str:
    """Follow Squeeze / Unsqueeze / Reshape / Identity / Flatten / Cast-free plumbing from tensor t to the tensor that is next USED."""
    while True:
        cons = fl.consumers.get(t, [])
        if len(cons) == 1 and cons[0]["op"] in ("Squeeze", "Unsqueeze", "Reshape", "Identity", "Flatten"):
            t = cons[0]["outputs"][0]
        else:
            return t


def load_head(path: str) -> dict:
    """A wake-word model file -> {'kind', 'T', 'hidden', 'n_out', 'net'[, 'net2']} (the layout of weights.synthetic_head).

    Walked by dataflow from the graph input: [Flatten | Reshape] -> network(s) -> output.  Two networks fed by the same features are
    the gated form of hey_jarvis; its routing must be the one the kernels implement -- `where(first > 0.5, second, first)`, written
    as Greater + Where or as Greater + If (second network inside the then-branch, the first score passed through by the
    else-branch) -- with that threshold and that branch order, or the file is refused."""
    g = load_graph(path)
    fl = _Flow(g, path)
    x = fl.start()
    feats = _shape_only(fl, x)
    cons = fl.consumers.get(feats, [])
    lin = [c for c in cons if c["op"] in ("Gemm", "MatMul")]
    ifs = [n for n in g["nodes"] if n["op"] == "If"]
    rec = sorted({n["op"] for n in g["nodes"] if n["op"] in ("LSTM", "GRU", "RNN")})
    if rec == ["LSTM"]:
        return _load_rnn_head(fl, g, x)                         # train.py:85-98: model_type = "rnn"
    if rec:
        fl.refuse(f"a recurrent network of {', '.join(rec)} nodes; of the recurrent forms only train.py's model_type 'rnn' (two "
                  "bidirectional LSTM(64) layers, train.py:85-98) has a kernel")
    if not lin:
        fl.refuse(f"no linear layer consumes the (flattened) input; found {[c['op'] for c in cons]}")
    if len(lin) > 2 or len(lin) + len(ifs) > 2:
        fl.refuse(f"{len(lin)} networks and {len(ifs)} If nodes read the input; one network, or two with a 0.5 gate, are supported")
    net, tail, s1 = _walk_net(fl, lin[0], "network 0")
    T, rem = divmod(net["w1"].shape[0], W.EMB_DIM)
    if rem:
        fl.refuse(f"first layer input {net['w1'].shape[0]} is not a multiple of {W.EMB_DIM}")
    hidden, n_out = net["w1"].shape[1], net["w3"].shape[1]
    net2 = None
    if len(lin) == 2 or ifs:
        if ifs:
            net2, s1, s_out = _gate_if(fl, ifs[0], feats, s1)
        else:
            net2, tail2, s2 = _walk_net(fl, lin[1], "network 1")
            if tail2 != tail:
                fl.refuse(f"the two networks of a gated model end differently: {tail} / {tail2}")
            s_out = _gate_where(fl, s1, s2)
        for k in ("w1", "w3"):
            if net2[k].shape != net[k].shape:
                fl.refuse(f"the two networks of a gated model differ in shape: {net[k].shape} / {net2[k].shape}")
        if len(W.net_blocks(net2)) != len(W.net_blocks(net)):
            fl.refuse(f"the two networks of a gated model differ in depth: {len(W.net_blocks(net))} / {len(W.net_blocks(net2))} hidden blocks")
        if (net2["ln1"] is None) != (net["ln1"] is None):
            fl.refuse("LayerNorm in one network of a gated model but not the other")
    # the activations the kernels apply are fixed (Linear -> [LN] -> ReLU per hidden layer, then Sigmoid, or ReLU + Softmax for a multiclass
    # model): the file must have exactly them
    if tail == ["Sigmoid"]:
        kind = "gated" if net2 is not None else "binary"
    elif tail == ["Relu", "Softmax"] and net2 is None:
        kind = "multiclass"
    else:
        # e.g. train.py's multiclass branch ends in a bare ReLU (train.py:81-83): no kernel applies that, so refuse
        fl.refuse(f"unsupported output activation {tail} (supported: Sigmoid, or Relu -> Softmax)")
    head = {"kind": kind, "T": int(T), "hidden": int(hidden), "n_out": int(n_out), "net": net}
    if net2 is not None:
        head["net2"] = net2
    return head


def _load_rnn_head(fl: _Flow, g: dict, x: str) -> dict:
    """train.py:85-98 (model_type "rnn") as torch.onnx.export writes it: input [B, T, 96] -> Transpose to time-major -> LSTM
    (bidirectional, hidden 64, zero initial state) -> Transpose + Reshape to [T, B, 128] -> LSTM (the same) -> back to batch-first ->
    Gather of the LAST time step -> Gemm(128 -> n_out) -> Sigmoid | Relu -> Softmax.  Returns the layout of weights.synthetic_rnn_head.
    Everything the kernel fixes is checked: two layers, both directions, hidden size, default activations, no peepholes / sequence
    lengths / input_forget, zero initial state, the last-step selection, the output activation."""
    nodes, inits = g["nodes"], g["initializers"]
    H = W.RNN_HID
    allowed = {"LSTM", "Transpose", "Reshape", "Shape", "Gather", "Unsqueeze", "Squeeze", "Concat", "Expand", "Slice", "Cast", "Identity",
               "ConstantOfShape", "Gemm", "MatMul", "Add", "Sigmoid", "Relu", "Softmax", "Flatten"}
    odd = sorted({n["op"] for n in nodes} - allowed)
    if odd:
        fl.refuse(f"a recurrent head with {odd}: not the graph of train.py's model_type 'rnn'")
    lstms = [n for n in nodes if n["op"] == "LSTM"]
    if len(lstms) != 2:
        fl.refuse(f"{len(lstms)} LSTM nodes, expected the two layers of nn.LSTM(96, 64, num_layers=2, bidirectional=True) (train.py:88)")

    def upstream(t: str) -> str:                                  # through shape-only operators to the tensor that carries the data
        seen = 0
        while t in fl.producer and fl.producer[t]["op"] in ("Transpose", "Reshape", "Squeeze", "Unsqueeze", "Identity", "Cast", "Flatten") and seen < 16:
            t = fl.producer[t]["inputs"][0]
            seen += 1
        return t

    if upstream(lstms[0]["inputs"][0]) != x:
        lstms.reverse()
    if upstream(lstms[0]["inputs"][0]) != x or upstream(lstms[1]["inputs"][0]) != lstms[0]["outputs"][0]:
        fl.refuse("the two LSTM layers are not chained input -> layer 0 -> Y -> layer 1")
    layers = []
    for li, n in enumerate(lstms):
        a = n["attrs"]
        n_in = W.EMB_DIM if li == 0 else 2 * H
        if (a.get("hidden_size") or 0) != H or (a.get("direction") or "forward") != "bidirectional":
            fl.refuse(f"LSTM layer {li}: hidden_size {a.get('hidden_size')} / direction {a.get('direction')}, expected {H} / bidirectional")
        if a.get("activations") or a.get("input_forget") or a.get("layout") or a.get("clip") is not None:
            fl.refuse(f"LSTM layer {li} sets activations / input_forget / layout / clip: only the defaults have a kernel")
        ins = n["inputs"] + [""] * (8 - len(n["inputs"]))
        if ins[4] or ins[7]:
            fl.refuse(f"LSTM layer {li} uses sequence_lens / peepholes")
        for k in (5, 6):                                          # initial_h / initial_c: absent, or zeros (Expand of a zero constant)
            t = ins[k]
            if not t:
                continue
            while t in fl.producer and fl.producer[t]["op"] in ("Expand", "Identity", "Reshape", "Cast"):
                t = fl.producer[t]["inputs"][0]
            v = inits.get(t)
            if v is None and t in fl.producer and fl.producer[t]["op"] == "ConstantOfShape":
                cv = fl.producer[t]["attrs"].get("value")
                v = np.zeros(1) if cv is None else np.asarray(cv)
            if v is None or np.any(np.asarray(v) != 0):
                fl.refuse(f"LSTM layer {li} starts from a non-zero (or computed) state; nn.LSTM called without (h0, c0) starts from zeros")
        try:
            wi, wr = inits[ins[1]], inits[ins[2]]                 # [2, 4H, in], [2, 4H, H]; ONNX gate order i, o, f, c
            bb = inits[ins[3]] if ins[3] else np.zeros((2, 8 * H), np.float32)
        except KeyError:
            fl.refuse(f"LSTM layer {li}: weights are not initializers")
        if wi.shape != (2, 4 * H, n_in) or wr.shape != (2, 4 * H, H) or bb.shape != (2, 8 * H):
            fl.refuse(f"LSTM layer {li} weights {tuple(wi.shape)} / {tuple(wr.shape)} / {tuple(bb.shape)}, expected (2, {4 * H}, {n_in}), (2, {4 * H}, {H}), (2, {8 * H})")
        order = [0, 2, 3, 1]                                      # ours (torch's): i | f | g | o  <-  ONNX blocks i(0) o(1) f(2) c(3)
        dirs = []
        for d in range(2):
            rows = np.concatenate([wi[d], wr[d]], axis=1)         # [4H, in + H]: columns x ; h
            rows = np.concatenate([rows[k * H:(k + 1) * H] for k in order], axis=0)
            bias = bb[d, :4 * H] + bb[d, 4 * H:]
            bias = np.concatenate([bias[k * H:(k + 1) * H] for k in order])
            dirs.append((np.ascontiguousarray(rows.T, np.float32), np.ascontiguousarray(bias, np.float32)))
        layers.append(dirs)
    # ---- out[:, -1] -> Linear -> activation
    lin = [n for n in nodes if n["op"] in ("Gemm", "MatMul") and any(i in inits and inits[i].ndim == 2 and 2 * H in inits[i].shape for i in n["inputs"])]
    if len(lin) != 1:
        fl.refuse(f"{len(lin)} linear layers of {2 * H} inputs behind the LSTM, expected one (train.py:90)")
    gsrc = fl.data_input(lin[0])
    sel = fl.producer.get(gsrc)
    while sel is not None and sel["op"] in ("Reshape", "Squeeze", "Unsqueeze", "Identity", "Flatten"):
        sel = fl.producer.get(sel["inputs"][0])
    shape_in = (g.get("shapes") or {}).get(x) or []
    T = shape_in[1] if len(shape_in) == 3 and shape_in[1] else None
    if sel is None or sel["op"] != "Gather" or upstream(sel["inputs"][0]) != lstms[1]["outputs"][0]:
        fl.refuse("the linear layer does not read ONE time step of the second LSTM layer's output (out[:, -1], train.py:94)")
    idx = inits.get(sel["inputs"][1])
    tr = fl.producer.get(sel["inputs"][0])
    batch_first = tr is not None and tr["op"] == "Transpose" and list(tr["attrs"].get("perm") or []) == [1, 0, 2]
    want_axis = 1 if batch_first else 0
    if idx is None or np.asarray(idx).size != 1 or int(sel["attrs"].get("axis", 0)) != want_axis or \
            int(np.asarray(idx).reshape(-1)[0]) not in ((-1,) if T is None else (-1, T - 1)):
        fl.refuse(f"the time step selected (Gather axis {sel['attrs'].get('axis', 0)}, index {None if idx is None else np.asarray(idx).reshape(-1).tolist()}) "
                  "is not the last one (out[:, -1])")
    w, b, cur = _walk_linear(fl, lin[0], "output layer")
    tail = []
    while True:
        cons = fl.consumers.get(cur, [])
        if len(cons) != 1 or cons[0]["op"] not in ("Sigmoid", "Relu", "Softmax"):
            break
        tail.append(cons[0]["op"])
        cur = cons[0]["outputs"][0]
    n_out = int(w.shape[1])
    if not ((tail == ["Sigmoid"] and n_out == 1) or (tail == ["Relu", "Softmax"] and n_out > 1)):
        fl.refuse(f"unsupported output activation {tail} for {n_out} output(s) (supported: Sigmoid for one class, Relu -> Softmax -- the "
                  "wrapper train.py:152-165 exports multiclass models under)")
    if T is None:
        fl.refuse("the input's time dimension is not declared in the file (the reference reads it from there: model.py:156)")
    if w.shape[0] != 2 * H or cur not in g["outputs"]:
        fl.refuse("the output layer does not end the graph")
    return {"kind": "rnn", "T": int(T), "hidden": H, "n_out": n_out, "lstm": layers,
            "w_out": np.ascontiguousarray(w, np.float32), "b_out": np.ascontiguousarray(b, np.float32)}


def _pick_zero(fl: _Flow, n: dict) -> bool:
    """Gather(x, 0): the element pick a scripted `if score[0, 0] > 0.5:` becomes (one Gather per index)."""
    if n["op"] != "Gather" or len(n["inputs"]) < 2 or n["inputs"][1] not in fl.inits:
        return False
    idx = np.asarray(fl.inits[n["inputs"][1]])
    return idx.size == 1 and int(idx.reshape(-1)[0]) == 0


def _gate_condition(fl: _Flow, s1: str, where: str) -> dict:
    """The comparison node that tests the first network's score against the gate threshold."""
    _CMP = ("Greater", "Less", "GreaterOrEqual", "LessOrEqual")
    cmps, seen, todo = [], set(), [s1]
    while todo:                      # through shape-only plumbing and element picks of the [1, 1] score (other consumers: Where / If arms)
        t = todo.pop()
        if t in seen:
            continue
        seen.add(t)
        for c in fl.consumers.get(t, []):
            if c["op"] in _CMP:
                cmps.append(c)
            elif c["op"] in ("Squeeze", "Unsqueeze", "Reshape", "Identity", "Flatten") or _pick_zero(fl, c):
                todo.append(c["outputs"][0])
    if not cmps:
        fl.refuse(f"{where}: two networks but no comparison of the first score with a threshold")
    n = cmps[0]
    thr = _scalar(fl.const(n))
    if n["op"] != "Greater" or n["inputs"][1] not in fl.inits:
        fl.refuse(f"{where}: the gate compares with {n['op']} (operands {n['inputs']}); the kernels implement `first > {GATE_THRESHOLD}`")
    if abs(thr - GATE_THRESHOLD) > 1e-7:
        fl.refuse(f"{where}: gate threshold {thr:g}; the kernels implement {GATE_THRESHOLD}")
    return n


def _gate_where(fl: _Flow, s1: str, s2: str) -> str:
    cmp_node = _gate_condition(fl, s1, "gate")
    cond = _shape_only(fl, cmp_node["outputs"][0])
    wh = [c for c in fl.consumers.get(cond, []) if c["op"] == "Where"]
    if len(wh) != 1:
        fl.refuse(f"gate: the comparison feeds {[c['op'] for c in fl.consumers.get(cond, [])]}, expected one Where")
    a, b = wh[0]["inputs"][1], wh[0]["inputs"][2]
    reach = lambda src, dst: dst == src or dst == _shape_only(fl, src)
    if not (reach(s2, a) and reach(s1, b)):
        fl.refuse("gate: Where(first > 0.5, X, Y) must select the SECOND network's score where the condition holds and the first "
                  "network's score elsewhere")
    return wh[0]["outputs"][0]


def _gate_if(fl: _Flow, node: dict, feats: str, s1: str):
    cmp_node = _gate_condition(fl, s1, "gate (If)")
    cond = cmp_node["outputs"][0]
    t = cond
    while t != node["inputs"][0]:                                   # Squeeze / ReduceMax / Reshape between the comparison and the If
        cons = fl.consumers.get(t, [])
        if len(cons) != 1 or (cons[0]["op"] not in ("Squeeze", "Reshape", "ReduceMax", "ReduceMin", "Identity") and
                              not (cons[0]["op"] == "Cast" and int(cons[0]["attrs"].get("to", 9)) == 9)):      # (Cast to bool only)
            fl.refuse(f"gate (If): the condition passes through {[c['op'] for c in cons]}")
        t = cons[0]["outputs"][0]
    then_g, else_g = node["attrs"].get("then_branch"), node["attrs"].get("else_branch")
    if not isinstance(then_g, dict) or not isinstance(else_g, dict):
        fl.refuse("gate (If): branches are not readable graphs")
    # else: the first network's score, passed through
    ef = _Flow(else_g, fl.path)
    src = else_g["outputs"][0] if else_g["outputs"] else ""
    while src in ef.producer and ef.producer[src]["op"] in ("Identity", "Squeeze", "Unsqueeze", "Reshape"):
        src = ef.data_input(ef.producer[src])
    if src != s1 and src != _shape_only(fl, s1) and fl.producer.get(src, {}).get("op") not in ("Squeeze", "Unsqueeze", "Reshape", "Identity"):
        fl.refuse(f"gate (If): the else-branch returns '{src}', not the first network's score")
    # then: the second network applied to the same features (captured from the outer scope)
    tf = _Flow(then_g, fl.path)
    inner = feats                    # ... or to its own flattening of the model input (what torch.jit.script + the exporter emit)
    x_outer = fl.start()
    own = [c for c in tf.consumers.get(x_outer, []) if c["op"] in ("Flatten", "Reshape")] if x_outer != feats else []
    if not tf.consumers.get(feats) and len(own) == 1:
        if own[0]["op"] == "Flatten" and int(own[0]["attrs"].get("axis", 1)) != 1:
            fl.refuse(f"gate (If): the then-branch flattens the input over axis {own[0]['attrs'].get('axis')}")
        inner = _shape_only(tf, own[0]["outputs"][0])
    lin = [c for c in tf.consumers.get(inner, []) if c["op"] in ("Gemm", "MatMul")]
    if len(lin) != 1:
        fl.refuse(f"gate (If): the then-branch does not apply one network to the features ({[c['op'] for c in tf.consumers.get(inner, [])]})")
    net2, tail2, _ = _walk_net(tf, lin[0], "network 1 (then-branch)")
    if tail2 != ["Sigmoid"]:
        fl.refuse(f"gate (If): the then-branch ends in {tail2}")
    return net2, s1, node["outputs"][0]


# ------------------------------------------------------------------------------------------- embedding CNN
LEAKY_ALPHA = 0.2      # a(x) = max(max(0.2 x, x), -0.4)  (converting_google_speech_embedding_model.ipynb:871-879)
ACT_FLOOR = -0.4


def _conv_pads(fl: _Flow, n: dict, kh: int, kw: int, where: str):
    at = n["attrs"]
    if any(int(v) != 1 for v in (at.get("strides") or [1, 1])) or any(int(v) != 1 for v in (at.get("dilations") or [1, 1])) or int(at.get("group", 1)) != 1:
        fl.refuse(f"{where}: strides / dilations / group = {at.get('strides')} / {at.get('dilations')} / {at.get('group', 1)}")
    ap = at.get("auto_pad", "NOTSET") or "NOTSET"
    if ap == "NOTSET":
        p = [int(v) for v in (at.get("pads") or [0, 0, 0, 0])]
    elif ap == "VALID":
        p = [0, 0, 0, 0]
    elif ap in ("SAME_UPPER", "SAME_LOWER"):
        p = [(kh - 1) // 2, (kw - 1) // 2, (kh - 1) // 2, (kw - 1) // 2]      # stride 1, odd kernels: symmetric
    else:
        fl.refuse(f"{where}: auto_pad {ap}")
    return p


def load_embedding(path: str) -> dict:
    """`embedding_model.onnx` -> {'conv': [HWIO] x 20, 'bn': [(gamma, beta, mean, var)] x 19} (weights.synthetic_embedding layout).

    The graph is walked by dataflow from its input and EVERY operator on the way is interpreted and checked against what the kernels
    compute (notebook cell 18: SURVEY 8a-E) -- the weights alone do not make the network:
      * layout: Transpose NHWC -> NCHW at the input (and back / Squeeze / Reshape at the output);
      * padding: zero padding of the MEL axis only, +-1 in front of the 3x3 and every 1x3 convolution, none on the time axis --
        as a Pad node, as Conv `pads`, or as auto_pad on a 1x3 kernel;
      * conv0 -> Relu -> BatchNorm; every other convolution but the last -> BatchNorm, where a BatchNorm is a BatchNormalization
        node, a per-channel Mul / Add pair, or already folded into the convolution (scaled weights + bias);
      * the activation after every BatchNorm: LeakyRelu(0.2) or Max(Mul(x, 0.2), x), then Clip(min=-0.4) or Max(., -0.4);
      * MaxPool 2x2 / 1x2 exactly where weights.CNN_TOPOLOGY has them (kernel = stride, no padding).
    Anything else is refused, naming the operator and the layer."""
    g = load_graph(path)
    fl = _Flow(g, path)
    cur = fl.start()
    topo = W.CNN_TOPOLOGY
    nchw = False                    # layout of `cur`: tf2onnx graphs start NHWC and transpose once; torch exports start NCHW
    pend = [0, 0, 0, 0]             # zero padding waiting for the next convolution: (time before, mel before, time after, mel after)
    conv, bn = [], []
    li = -1                         # layer whose convolution has been read
    st = dict(relu=False, bn=False, leaky=False, floor=False, pool=False)
    layout_known = False

    def layer_done(where):
        if li < 0:
            return
        kh, kw, ci, co, pool = topo[li]
        if li == len(topo) - 1:
            if any(st.values()):
                fl.refuse(f"{where}: operators after the last convolution ({[k for k, v in st.items() if v]})")
            return
        need = dict(relu=li == 0, bn=True, leaky=True, floor=True, pool=pool is not None)
        for k in ("relu", "bn", "leaky", "floor", "pool"):
            if st[k] != need[k]:
                fl.refuse(f"{where}: layer {li} {'lacks' if need[k] else 'has an unexpected'} {k} stage (found {[q for q, v in st.items() if v]})")

    while True:
        cons = fl.consumers.get(cur, [])
        where = f"after conv {li}" if li >= 0 else "at the input"
        if not cons:
            break
        # Max(Mul(x, 0.2), x): x has two consumers
        if len(cons) == 2 and {c["op"] for c in cons} == {"Mul", "Max"}:
            mul = next(c for c in cons if c["op"] == "Mul")
            mx = next(c for c in cons if c["op"] == "Max")
            if abs(_scalar(fl.const(mul)) - LEAKY_ALPHA) > 1e-6 or set(mx["inputs"]) != {cur, mul["outputs"][0]}:
                fl.refuse(f"{where}: Mul / Max pair is not max({LEAKY_ALPHA} x, x) (factor {_scalar(fl.const(mul))})")
            if st["leaky"] or st["floor"] or st["pool"] or not st["bn"]:
                fl.refuse(f"{where}: leaky stage out of order")
            st["leaky"] = True
            cur = mx["outputs"][0]
            continue
        if len(cons) != 1:
            fl.refuse(f"{where}: tensor '{cur}' feeds {[c['op'] for c in cons]}; the embedding network is a chain")
        n = cons[0]
        op = n["op"]
        if op == "Transpose":
            perm = [int(v) for v in (n["attrs"].get("perm") or [])]
            if perm == [0, 3, 1, 2] and not nchw:
                nchw = True
            elif perm == [0, 2, 3, 1] and nchw and li == len(topo) - 1:
                nchw = False
            else:
                fl.refuse(f"{where}: Transpose perm {perm}")
            layout_known = True
        elif op == "Pad":
            pads = n["attrs"].get("pads")
            if pads is None and len(n["inputs"]) > 1 and n["inputs"][1] in fl.inits:
                pads = fl.inits[n["inputs"][1]].tolist()
            val = n["attrs"].get("value", 0.0)
            if len(n["inputs"]) > 2 and n["inputs"][2] in fl.inits:
                val = _scalar(fl.inits[n["inputs"][2]])
            if pads is None or len(pads) != 8 or (n["attrs"].get("mode", "constant") or "constant") != "constant" or float(val or 0.0) != 0.0:
                fl.refuse(f"{where}: Pad with pads {pads}, mode {n['attrs'].get('mode')}, value {val}")
            pads = [int(v) for v in pads]
            hi, wi = (2, 3) if nchw else (1, 2)
            if any(pads[k] for k in range(8) if k % 4 not in (hi, wi)):
                fl.refuse(f"{where}: Pad touches the batch / channel axis: {pads}")
            pend = [pend[0] + pads[hi], pend[1] + pads[wi], pend[2] + pads[4 + hi], pend[3] + pads[4 + wi]]
        elif op == "Conv":
            layer_done(where)
            li += 1
            st = dict(relu=False, bn=False, leaky=False, floor=False, pool=False)
            if li >= len(topo):
                fl.refuse(f"more than {len(topo)} convolutions")
            if not nchw and layout_known is False and li == 0:
                nchw = True                                      # no Transpose seen: the file is NCHW from the start (torch export)
            kh, kw, ci, co, _pool = topo[li]
            w = fl.inits.get(n["inputs"][1])
            if w is None or w.ndim != 4:
                fl.refuse(f"conv {li}: no constant 4-D weight")
            if w.shape != (co, ci, kh, kw):
                fl.refuse(f"conv {li}: OIHW weight {tuple(w.shape)}, expected {(co, ci, kh, kw)}")
            p = _conv_pads(fl, n, kh, kw, f"conv {li}")
            tot = [p[k] + pend[k] for k in range(4)]
            want = [0, 1, 0, 1] if kw == 3 else [0, 0, 0, 0]
            if tot != want:
                fl.refuse(f"conv {li} ({kh}x{kw}): zero padding (time, mel, time, mel) = {tot}; the kernels pad the mel axis only: {want}")
            pend = [0, 0, 0, 0]
            bias = fl.inits.get(n["inputs"][2]) if len(n["inputs"]) > 2 and n["inputs"][2] else None
            conv.append(np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)), np.float32))            # OIHW -> HWIO
            if bias is not None and np.abs(bias).max() > 0:
                if li == 0:
                    fl.refuse("conv 0 carries a bias in front of its Relu; the kernel has no slot for it")
                if li == len(topo) - 1:
                    fl.refuse("the last convolution carries a bias; the kernel has no slot for it")
                one = np.ones(co, np.float32)                      # a BatchNorm folded into the convolution: y = conv + bias
                bn.append((one, np.asarray(bias, np.float32).reshape(co), np.zeros(co, np.float32), one - np.float32(W.BN_EPS)))
                st["bn"] = True
        elif op == "Relu":
            if li != 0 or st["relu"] or st["bn"]:
                fl.refuse(f"{where}: Relu (only conv 0 is followed by one, in front of its BatchNorm)")
            st["relu"] = True
        elif op == "BatchNormalization":
            if li < 0 or li == len(topo) - 1 or st["leaky"] or st["floor"] or st["pool"] or (li == 0 and not st["relu"]):
                fl.refuse(f"{where}: BatchNormalization out of order")
            p = [fl.inits.get(x) for x in n["inputs"][1:5]]
            co = topo[li][3]
            if len(p) != 4 or any(x is None or x.shape != (co,) for x in p):
                fl.refuse(f"{where}: BatchNormalization without four constant [{co}] parameters")
            eps = float(n["attrs"].get("epsilon", 1e-5))
            gm, bt, mu, var = (x.astype(np.float64) for x in p)
            if abs(eps - W.BN_EPS) > 1e-12:                       # re-expressed with this package's epsilon
                var = var + (eps - W.BN_EPS)
            new = (gm, bt, mu, var)
            if st["bn"]:                                           # a bias on the convolution AND a BatchNorm: BN(conv + b) = BN'(conv)
                _one, b0, _z, _v = bn.pop()
                new = (gm, bt, mu - b0.astype(np.float64), var)
            bn.append(tuple(a.astype(np.float32) for a in new))
            st["bn"] = True
        elif op in ("Mul", "Add") and fl.const(n) is not None and fl.const(n).size == topo[max(li, 0)][3] and li >= 0 and not st["leaky"]:
            # BatchNorm written as a per-channel affine map (Mul scale, Add shift)
            co = topo[li][3]
            c = np.asarray(fl.const(n), np.float64).reshape(-1)
            shp = list(np.asarray(fl.const(n)).shape)
            ch_axis = 1 if nchw else 3
            if len(shp) > 1 and (len(shp) != 4 or shp[ch_axis] != co):
                fl.refuse(f"{where}: per-channel {op} constant of shape {shp} in {'NCHW' if nchw else 'NHWC'} layout")
            if li == len(topo) - 1 or (li == 0 and not st["relu"]):
                fl.refuse(f"{where}: per-channel {op} out of order")
            if not st["bn"]:
                one = np.ones(co)
                bn.append((one.astype(np.float32), np.zeros(co, np.float32), np.zeros(co, np.float32), (one - W.BN_EPS).astype(np.float32)))
                st["bn"] = True
            gm, bt, mu, var = (a.astype(np.float64) for a in bn.pop())
            k = np.sqrt(var + W.BN_EPS)
            scale, shift = gm / k, bt - mu * gm / k
            scale, shift = (scale * c, shift * c) if op == "Mul" else (scale, shift + c)
            bn.append((scale.astype(np.float32), shift.astype(np.float32), np.zeros(co, np.float32), (np.ones(co) - W.BN_EPS).astype(np.float32)))
        elif op == "LeakyRelu":
            if abs(float(n["attrs"].get("alpha", 0.01)) - LEAKY_ALPHA) > 1e-6:
                fl.refuse(f"{where}: LeakyRelu alpha {n['attrs'].get('alpha', 0.01)}; the kernels apply {LEAKY_ALPHA}")
            if st["leaky"] or st["floor"] or st["pool"] or not st["bn"]:
                fl.refuse(f"{where}: LeakyRelu out of order")
            st["leaky"] = True
        elif op in ("Clip", "Max"):
            if op == "Clip":
                lo = n["attrs"].get("min")
                hi_ = n["attrs"].get("max")
                if lo is None and len(n["inputs"]) > 1 and n["inputs"][1] in fl.inits:
                    lo = _scalar(fl.inits[n["inputs"][1]])
                if hi_ is None and len(n["inputs"]) > 2 and n["inputs"][2] in fl.inits:
                    hi_ = _scalar(fl.inits[n["inputs"][2]])
                if hi_ is not None and float(hi_) < 3.0e38:
                    fl.refuse(f"{where}: Clip with an upper bound {hi_}")
            else:
                lo = _scalar(fl.const(n))
            if lo is None or abs(float(lo) - ACT_FLOOR) > 1e-6:
                fl.refuse(f"{where}: {op} with lower bound {lo}; the kernels apply {ACT_FLOOR}")
            if not st["leaky"] or st["floor"] or st["pool"]:
                fl.refuse(f"{where}: the {ACT_FLOOR} floor must follow the leaky stage")
            st["floor"] = True
        elif op == "MaxPool":
            pool = topo[li][4] if li >= 0 else None
            ks = [int(v) for v in (n["attrs"].get("kernel_shape") or [])]
            ss = [int(v) for v in (n["attrs"].get("strides") or ks)]
            pp = [int(v) for v in (n["attrs"].get("pads") or [0, 0, 0, 0])]
            if pool is None or ks != list(pool) or ss != list(pool) or any(pp) or int(n["attrs"].get("ceil_mode", 0)) or \
                    (n["attrs"].get("auto_pad", "NOTSET") or "NOTSET") not in ("NOTSET", "VALID") or not nchw:
                fl.refuse(f"{where}: MaxPool kernel {ks} strides {ss} pads {pp}; weights.CNN_TOPOLOGY has {pool} after layer {li}")
            if not st["floor"] or st["pool"]:
                fl.refuse(f"{where}: MaxPool before the activation")
            st["pool"] = True
        elif op in ("Squeeze", "Reshape", "Flatten", "Identity") and li == len(topo) - 1:
            pass
        else:
            fl.refuse(f"{where}: operator {op} is not part of the speech-embedding network the kernels implement")
        cur = n["outputs"][0]
    layer_done("at the output")
    if li != len(topo) - 1:
        fl.refuse(f"expected {len(topo)} convolutions on the path from the input, found {li + 1}")
    if any(pend):
        fl.refuse(f"a Pad node {pend} is not followed by a convolution")
    return {"conv": conv, "bn": bn}


def check_melspectrogram(path: str) -> dict:
    """Compare the file's mel filterbank ([257, 32] MatMul operand) with this package's analytic table (see verify_melspectrogram
    for the whole graph)."""
    g = load_graph(path)
    fb = [a for a in g["initializers"].values() if a.shape in ((W.N_BINS, W.N_MELS), (W.N_MELS, W.N_BINS))]
    if not fb:
        raise ValueError(f"{path}: no {W.N_BINS}x{W.N_MELS} filterbank initializer found")
    m = fb[0] if fb[0].shape == (W.N_BINS, W.N_MELS) else fb[0].T
    return {"filterbank_max_abs_diff": float(np.abs(m - W.mel_filterbank()).max()), "filterbank_max": float(np.abs(m).max())}


def verify_melspectrogram(path: str, tol: float = 1e-4) -> dict:
    """`melspectrogram.onnx` (utils.py:84-87; built by notebooks/converting_google_speech_embedding_model.ipynb cell 15 from
    torchlibrosa's Spectrogram + LogmelFilterBank with a patched power_to_db) CANNOT be loaded into the HIP front end -- that kernel is
    analytic (Hann(400) centred in 512-sample frames, hop 160, 512-point DFT, power, Slaney filter bank 60..3800 Hz x 32, 10 log10
    with amin 1e-10, clamp at the call's maximum - 80 dB).  What can be done is to VERIFY that the file computes exactly that, and to
    refuse to run next to a file that does not (a different window, hop, filter bank or top_db would score differently in silence):

      * two STFT convolutions with [257, 1, 512] kernels, stride 160, no padding, equal (up to sign) to window x cos / sin;
      * power 2 (no Sqrt), one MatMul against a [257, 32] matrix equal to weights.mel_filterbank();
      * a lower clamp of 1e-10 in front of one Log, constant factors behind it that multiply to 10 / ln 10, no reference offset;
      * a ReduceMax of the result, 80 subtracted from it, and a Max / Clip against that;
      * nothing else that does arithmetic.
    Returns the measured differences; raises ValueError naming what differs -- and its subclass GraphIdiomUnknown when nothing was found
    to differ but the walker met a structure it does not know (an unfolded constant, a second Log for a reference offset, a merged
    real / imaginary convolution ...): the caller may then go on under a warning, see model.resolve_embedding."""
    g = load_graph(path)
    fl = _Flow(g, path)
    nodes, inits = g["nodes"], g["initializers"]
    shape_ops = {"Unsqueeze", "Squeeze", "Transpose", "Reshape", "Identity", "Cast", "Shape", "Gather", "Concat", "Slice", "Flatten",
                 "ConstantOfShape", "Expand"}
    known = shape_ops | {"Conv", "Mul", "Pow", "Add", "Sub", "Div", "MatMul", "Clip", "Max", "Log", "ReduceMax", "Pad"}
    odd = sorted({n["op"] for n in nodes} - known)
    if odd:
        fl.refuse(f"melspectrogram graph contains {odd} (a Sqrt would mean a magnitude, not a power spectrogram); the HIP front end computes "
                  "a fixed recipe and cannot follow it", unknown="Sqrt" not in odd)
    for n in nodes:
        if n["op"] == "Pad":
            pads = n["attrs"].get("pads")
            if pads is None and len(n["inputs"]) > 1 and n["inputs"][1] in inits:
                pads = inits[n["inputs"][1]].tolist()
            if pads is None or any(int(v) for v in pads):
                fl.refuse(f"the waveform is padded ({pads}): the reference frames with center=False")
    # ---- STFT
    convs = [n for n in nodes if n["op"] == "Conv"]
    if len(convs) != 2:
        fl.refuse(f"{len(convs)} Conv nodes, expected the real and imaginary STFT convolutions", unknown=True)
    n_ = np.arange(W.N_FFT, dtype=np.float64)
    win = np.zeros(W.N_FFT)
    lo = (W.N_FFT - W.WIN) // 2
    win[lo:lo + W.WIN] = W.hann_window().astype(np.float64)
    ang = 2.0 * np.pi * np.outer(np.arange(W.N_BINS), n_) / W.N_FFT
    basis = {"cos": win[None, :] * np.cos(ang), "sin": win[None, :] * np.sin(ang)}
    found, worst = set(), 0.0
    for n in convs:
        w = inits.get(n["inputs"][1]) if len(n["inputs"]) > 1 else None
        if w is None or w.size != W.N_BINS * W.N_FFT or w.shape[0] != W.N_BINS:
            fl.refuse(f"STFT kernel of shape {None if w is None else tuple(w.shape)}, expected [{W.N_BINS}, 1, {W.N_FFT}]", unknown=w is None)
        k = np.asarray(w, np.float64).reshape(W.N_BINS, W.N_FFT)
        st = [int(v) for v in (n["attrs"].get("strides") or [1])]
        if max(st) != W.HOP or any(v not in (1, W.HOP) for v in st):
            fl.refuse(f"STFT stride {st}, the front end hops {W.HOP} samples")
        if any(int(v) for v in (n["attrs"].get("pads") or [0])) or (n["attrs"].get("auto_pad", "NOTSET") or "NOTSET") not in ("NOTSET", "VALID"):
            fl.refuse("the STFT convolutions pad their input (center=False expected)")
        d = {name: min(np.abs(k - b).max(), np.abs(k + b).max()) for name, b in basis.items()}
        name = min(d, key=d.get)
        if d[name] > tol:
            fl.refuse(f"an STFT kernel differs from Hann({W.WIN}) centred in {W.N_FFT} samples x cos / sin by {min(d.values()):.3g} "
                      "(another window, window length or FFT size)")
        found.add(name)
        worst = max(worst, d[name])
    if found != {"cos", "sin"}:
        fl.refuse("the two STFT kernels are not a cos / sin pair", unknown=True)
    # ---- power -> mel
    mm = [n for n in nodes if n["op"] == "MatMul"]
    fbs = [inits[i] for n in mm for i in n["inputs"] if i in inits and inits[i].shape in ((W.N_BINS, W.N_MELS), (W.N_MELS, W.N_BINS))]
    if len(mm) != 1 or len(fbs) != 1:
        fl.refuse(f"{len(mm)} MatMul nodes / {len(fbs)} [{W.N_BINS}, {W.N_MELS}] constants, expected one filter bank product", unknown=True)
    fb = fbs[0] if fbs[0].shape == (W.N_BINS, W.N_MELS) else fbs[0].T
    fb_diff = float(np.abs(fb.astype(np.float64) - W.mel_filterbank().astype(np.float64)).max())
    if fb_diff > 1e-6:
        fl.refuse(f"the mel filter bank differs from the Slaney bank 60..3800 Hz x {W.N_MELS} (slaney norm) by {fb_diff:.3g}")
    for n in nodes:
        if n["op"] == "Pow" and abs(_scalar(fl.const(n)) - 2.0) > 0:
            fl.refuse(f"Pow exponent {_scalar(fl.const(n))}: a power-2 spectrogram is expected")
    # ---- 10 log10(max(x, 1e-10)), no reference offset
    logs = [n for n in nodes if n["op"] == "Log"]
    if len(logs) != 1:
        fl.refuse(f"{len(logs)} Log nodes, expected one", unknown=True)
    src = fl.producer.get(logs[0]["inputs"][0])
    amin = None
    if src is not None and src["op"] == "Clip":
        amin = src["attrs"].get("min")
        if amin is None and len(src["inputs"]) > 1 and src["inputs"][1] in inits:
            amin = _scalar(inits[src["inputs"][1]])
    elif src is not None and src["op"] == "Max":
        amin = _scalar(fl.const(src))
    if amin is None or abs(float(amin) - 1e-10) > 1e-13:
        fl.refuse(f"the logarithm's input is clamped at {amin}, expected amin = 1e-10", unknown=amin is None)
    factor, offset, cur = 1.0, 0.0, logs[0]["outputs"][0]
    while True:
        cons = fl.consumers.get(cur, [])
        arith = [c for c in cons if c["op"] in ("Mul", "Div", "Sub", "Add") and fl.const(c) is not None and np.asarray(fl.const(c)).size == 1]
        if len(cons) != 1 or not arith:
            break
        c, v = arith[0], _scalar(fl.const(arith[0]))
        if c["op"] == "Mul":
            factor, offset = factor * v, offset * v
        elif c["op"] == "Div":
            if c["inputs"][0] != cur:
                break
            factor, offset = factor / v, offset / v
        elif c["op"] == "Add":
            offset += v
        else:
            if c["inputs"][0] != cur:
                break
            offset -= v
        cur = c["outputs"][0]
    if abs(factor - 10.0 / np.log(10.0)) > 1e-5 or abs(offset) > 1e-6:
        fl.refuse(f"the logarithm is scaled by {factor:.6g} and offset by {offset:.3g}; 10 log10 (x {10.0 / np.log(10.0):.6g}) with ref = 1 is expected")
    # ---- top_db: clamp at (max over the call) - 80
    rmax = [n for n in nodes if n["op"] == "ReduceMax"]
    if len(rmax) != 1:
        fl.refuse(f"{len(rmax)} ReduceMax nodes, expected the one of top_db", unknown=True)
    if rmax[0]["attrs"].get("axes") not in (None, []):
        fl.refuse(f"top_db reduces over axes {rmax[0]['attrs'].get('axes')}; the reference's patched power_to_db takes the maximum of the whole call")
    t = rmax[0]["outputs"][0]
    cons = fl.consumers.get(t, [])
    if len(cons) != 1 or cons[0]["op"] not in ("Sub", "Add") or fl.const(cons[0]) is None:
        fl.refuse("the call's maximum is not lowered by a constant top_db", unknown=True)
    top_db = _scalar(fl.const(cons[0])) * (1.0 if cons[0]["op"] == "Sub" else -1.0)
    if abs(top_db - 80.0) > 1e-6:
        fl.refuse(f"top_db = {top_db:g}; the front end clamps at the call's maximum - 80 dB")
    floor_t = cons[0]["outputs"][0]
    clampers = [c for c in fl.consumers.get(floor_t, []) if c["op"] in ("Max", "Clip")]
    if len(clampers) != 1 or cur not in clampers[0]["inputs"]:
        fl.refuse("the log-mel values are not clamped from below at (maximum - top_db)", unknown=True)
    return {"stft_kernel_max_abs_diff": worst, "filterbank_max_abs_diff": fb_diff, "log_factor": factor, "top_db": top_db}


# ---- numeric probe of a melspectrogram graph the walker cannot follow -------------------------------------------------------------
class _ProbeUnsupported(ValueError):
    pass


def _run_graph(g: dict, feeds: Dict[str, np.ndarray]) -> List[np.ndarray]:
    """Evaluate a (small, straight-line) fp32 graph with numpy: the operators a log-mel front end can be written with.  Load-time
    VERIFICATION only (probe_melspectrogram) -- nothing on the scoring path runs through it.  Raises _ProbeUnsupported on anything else."""
    env: Dict[str, np.ndarray] = dict(g["initializers"])
    env.update(feeds)

    def axes_of(n, idx=1):
        a = n["attrs"].get("axes")
        if a is None and len(n["inputs"]) > idx and n["inputs"][idx]:
            a = np.asarray(env[n["inputs"][idx]]).reshape(-1).tolist()
        return None if a is None else [int(v) for v in a]

    def conv(x, w, n):
        at = n["attrs"]
        if int(at.get("group", 1) or 1) != 1 or any(int(d) != 1 for d in (at.get("dilations") or [1])):
            raise _ProbeUnsupported("grouped / dilated Conv")
        if (at.get("auto_pad", "NOTSET") or "NOTSET") not in ("NOTSET", "VALID"):
            raise _ProbeUnsupported("Conv auto_pad")
        nd = x.ndim - 2
        st = [int(v) for v in (at.get("strides") or [1] * nd)]
        pads = [int(v) for v in (at.get("pads") or [0] * (2 * nd))]
        x = np.pad(x, [(0, 0), (0, 0)] + [(pads[i], pads[i + nd]) for i in range(nd)])
        win = np.lib.stride_tricks.sliding_window_view(x, w.shape[2:], axis=tuple(range(2, 2 + nd)))
        win = win[(slice(None), slice(None)) + tuple(slice(None, None, s_) for s_ in st)]       # [N, C, out..., k...]
        out = np.tensordot(win, w, axes=([1] + list(range(2 + nd, 2 + 2 * nd)), [1] + list(range(2, 2 + nd))))   # [N, out..., M]
        out = np.moveaxis(out, -1, 1)
        if len(n["inputs"]) > 2 and n["inputs"][2]:
            out = out + env[n["inputs"][2]].reshape((1, -1) + (1,) * nd)
        return out.astype(np.float32)

    binary = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply, "Div": np.divide, "Pow": np.power}
    unary = {"Log": np.log, "Sqrt": np.sqrt, "Abs": np.abs, "Neg": np.negative, "Exp": np.exp, "Relu": lambda v: np.maximum(v, 0), "Identity": lambda v: v}
    for n in g["nodes"]:
        op, at = n["op"], n["attrs"]
        x = [env[i] if i else None for i in n["inputs"]]
        with np.errstate(all="ignore"):
            if op in binary:
                y = binary[op](x[0], x[1])
                y = y.astype(np.result_type(x[0], x[1])) if isinstance(y, np.ndarray) else np.asarray(y)
            elif op in unary:
                y = unary[op](x[0])
            elif op in ("Max", "Min"):
                y = x[0]
                for o in x[1:]:
                    y = (np.maximum if op == "Max" else np.minimum)(y, o)
            elif op == "Clip":
                lo = at.get("min") if len(x) < 2 or x[1] is None else x[1]
                hi = at.get("max") if len(x) < 3 or x[2] is None else x[2]
                y = x[0]
                y = np.maximum(y, lo) if lo is not None else y
                y = np.minimum(y, hi) if hi is not None else y
            elif op == "Conv":
                y = conv(np.asarray(x[0], np.float32), np.asarray(x[1], np.float32), n)
            elif op == "MatMul":
                y = np.matmul(x[0], x[1])
            elif op in ("ReduceMax", "ReduceSum", "ReduceMean", "ReduceMin"):
                ax = axes_of(n)
                fn = {"ReduceMax": np.max, "ReduceSum": np.sum, "ReduceMean": np.mean, "ReduceMin": np.min}[op]
                y = fn(x[0], axis=None if ax is None else tuple(ax), keepdims=bool(at.get("keepdims", 1)))
            elif op == "Transpose":
                y = np.transpose(x[0], at.get("perm") or list(range(x[0].ndim))[::-1])
            elif op == "Reshape":
                shp = [int(d) for d in np.asarray(x[1]).reshape(-1)]
                y = x[0].reshape([x[0].shape[i] if d == 0 else d for i, d in enumerate(shp)])
            elif op == "Flatten":
                a = int(at.get("axis", 1))
                y = x[0].reshape(int(np.prod(x[0].shape[:a], dtype=np.int64)), -1)
            elif op == "Unsqueeze":
                y = x[0]
                for a in sorted(a_ if a_ >= 0 else a_ + x[0].ndim + len(axes_of(n)) for a_ in axes_of(n)):
                    y = np.expand_dims(y, a)
            elif op == "Squeeze":
                ax = axes_of(n)
                y = np.squeeze(x[0], axis=None if ax is None else tuple(ax))
            elif op == "Cast":
                y = np.asarray(x[0]).astype(_DTYPES.get(at.get("to", 1), np.float32))
            elif op == "Shape":
                y = np.asarray(x[0].shape, np.int64)
            elif op == "Gather":
                y = np.take(x[0], np.asarray(x[1], np.int64), axis=int(at.get("axis", 0)))
            elif op == "Concat":
                y = np.concatenate([np.atleast_1d(v) for v in x], axis=int(at.get("axis", 0)))
            elif op == "Slice":
                starts, ends = np.asarray(x[1]).reshape(-1), np.asarray(x[2]).reshape(-1)
                axes = np.asarray(x[3]).reshape(-1) if len(x) > 3 and x[3] is not None else np.arange(len(starts))
                steps = np.asarray(x[4]).reshape(-1) if len(x) > 4 and x[4] is not None else np.ones(len(starts), np.int64)
                sl = [slice(None)] * x[0].ndim
                for a, b, e, st_ in zip(axes, starts, ends, steps):
                    sl[int(a)] = slice(int(b), int(min(e, np.iinfo(np.int64).max)), int(st_))
                y = x[0][tuple(sl)]
            elif op == "Pad":
                pads = at.get("pads") if len(x) < 2 or x[1] is None else np.asarray(x[1]).reshape(-1).tolist()
                nd = x[0].ndim
                if (at.get("mode", "constant") or "constant") != "constant":
                    raise _ProbeUnsupported("Pad mode")
                y = np.pad(x[0], [(int(pads[i]), int(pads[i + nd])) for i in range(nd)])
            elif op == "Expand":
                y = x[0] * np.ones([int(d) for d in np.asarray(x[1]).reshape(-1)], x[0].dtype)
            elif op == "ConstantOfShape":
                v = at.get("value")
                y = np.full([int(d) for d in np.asarray(x[0]).reshape(-1)], v.reshape(-1)[0] if isinstance(v, np.ndarray) else 0.0,
                            v.dtype if isinstance(v, np.ndarray) else np.float32)
            else:
                raise _ProbeUnsupported(f"operator {op}")
        env[n["outputs"][0]] = np.asarray(y)
    return [env[o] for o in g["outputs"]]


def analytic_logmel_db(x: np.ndarray) -> np.ndarray:
    """The recipe of the HIP front end in numpy float64 (csrc/owwhip_fused.h, owwhip_kernels.h::mel_kernel): int16
    samples [n] as floats -> dB rows [F, 32] with the call's clamp floor."""
    x = np.asarray(x, np.float64)
    F = (len(x) - W.N_FFT) // W.HOP + 1
    win = np.zeros(W.N_FFT)
    lo = (W.N_FFT - W.WIN) // 2
    win[lo:lo + W.WIN] = W.hann_window().astype(np.float64)
    fr = np.lib.stride_tricks.sliding_window_view(x, W.N_FFT)[::W.HOP][:F] * win
    p = np.abs(np.fft.rfft(fr, axis=1)) ** 2
    db = 10.0 * np.log10(np.maximum(p @ W.mel_filterbank().astype(np.float64), 1e-10))
    return np.maximum(db, db.max() - 80.0)


def probe_melspectrogram(path: str, tol_db: float = 2e-2) -> dict:
    """Numeric check of a melspectrogram graph whose STRUCTURE verify_melspectrogram could not follow (GraphIdiomUnknown): the file is
    evaluated (numpy, _run_graph) on probe audio -- Gaussian noise at four levels, silence inside a loud call (the clamp floor), full
    scale square wave, a 1 kHz tone, several call lengths -- and must reproduce the analytic recipe to `tol_db` (0.02 dB = 2e-3 in the
    x/10 + 2 units the embedding network sees, far below what a different window / hop / bank / top_db / offset would move).  Returns
    the largest difference; raises ValueError when the file computes something else or uses an operator the evaluator does not have."""
    g = load_graph(path)
    fl = _Flow(g, path)
    name = fl.start()
    r = np.random.default_rng(0x0E1)
    t = np.arange(1280 * 3 + 480)
    probes = [np.clip(np.round(r.normal(0, a, n)), -32768, 32767) for a, n in ((30, 1760), (300, 1760), (3000, 3040), (12000, 4320))]
    loud = np.clip(np.round(r.normal(0, 9000, 4320)), -32768, 32767)
    loud[800:2600] = 0.0                                                       # silence inside a loud call: rows on the clamp floor
    probes += [loud, np.where((t // 8) % 2, 32767.0, -32768.0), np.round(8000 * np.sin(2 * np.pi * 1000 * t / 16000.0)), np.zeros(1760)]
    worst = 0.0
    for x in probes:
        try:
            out = _run_graph(g, {name: np.asarray(x, np.float32)[None, :]})[0]
        except _ProbeUnsupported as e:
            raise ValueError(f"{path}: cannot evaluate the melspectrogram graph for a numeric check ({e})") from e
        except Exception as e:                                                  # noqa: BLE001  (shape mismatch inside the evaluator etc.)
            raise ValueError(f"{path}: evaluating the melspectrogram graph failed ({type(e).__name__}: {e})") from e
        want = analytic_logmel_db(x)
        got = np.asarray(out, np.float64)
        if got.size != want.size:
            raise ValueError(f"{path}: the graph returns {got.shape} for {len(x)} samples, the front end produces {want.shape} rows")
        got = got.reshape(-1, W.N_MELS) if got.shape[-1] == W.N_MELS else np.squeeze(got)
        if got.shape != want.shape:
            raise ValueError(f"{path}: the graph returns {np.asarray(out).shape} for {len(x)} samples, the front end produces {want.shape} rows")
        d = float(np.abs(got - want).max())
        if not np.isfinite(d) or d > tol_db:
            raise ValueError(f"{path}: the graph's log-mel rows differ from the HIP front end's recipe by {d:.3g} dB on probe audio "
                             f"(tolerance {tol_db} dB): it computes a different front end")
        worst = max(worst, d)
    return {"probe_max_abs_diff_db": worst, "n_probes": len(probes)}


# ------------------------------------------------------------------------------------------- voice-activity network
def load_vad(path: str) -> dict:
    """`silero_vad.onnx` (/root/reference/openwakeword/vad.py:60-90; run at vad.py:98-130 with inputs input [1, n], h / c [2, 1, 64],
    sr) -> the weight dict of the on-device voice-activity network (`weights.synthetic_vad` layout, `engine.pack_vad_blob`),
    IF the file's graph is the architecture `csrc/owwhip_vad.h` implements:

        |STFT| (256-sample Hann frames, hop 64, bins 1..128; analytic in the kernel -- an STFT basis Conv in the file is only
        checked for its geometry) -> log(1 + gain |X|) -> 4 x [Conv1d(k=3, pad 1) + ReLU]: 128->16 (stride 1), 16->32 (2), 32->32 (2),
        32->64 (1) -> two ONNX `LSTM` nodes of hidden size 64 -> ReLU -> Linear / 1x1 Conv (64 -> 1) -> Sigmoid.

    Recognition is by structure (operator types, weight shapes, strides), never by node names.  Silero's real graph is not
    described anywhere in the reference and no copy of the file exists offline, so what this function does with
    the real release asset is unknown until one is seen: anything that is not exactly the structure above is REFUSED with a
    ValueError that lists what was found -- the caller then keeps the host path (`VAD(session=onnxruntime session)`,
    `oww_push_vad`) instead of a silently different network.  Any other failure while reading or recognising the file (a truncated
    protobuf, an LSTM node without its optional inputs, `If` subgraphs, attributes of an unexpected type ...) is a refusal too: it
    surfaces as the same ValueError, never as an IndexError / KeyError / struct.error from the middle of the reader."""
    try:
        return _load_vad(path)
    except ValueError:
        raise
    except Exception as e:
        raise ValueError(f"{path}: not the voice-activity architecture the HIP kernels implement (the graph could not be read as one: "
                         f"{type(e).__name__}: {e}).  Drive this network on the host instead (openwakeword_amd.VAD(session=...) / "
                         "oww_push_vad): the gate, the score ring and the state handling of vad.py stay the same") from e


def _load_vad(path: str) -> dict:
    g = load_graph(path)
    inits, nodes = g["initializers"], g["nodes"]
    ops = [n["op"] for n in nodes]

    def refuse(why: str):
        from collections import Counter
        raise ValueError(f"{path}: not the voice-activity architecture the HIP kernels implement ({why}); operators found: "
                         f"{dict(Counter(ops))}.  Drive this network on the host instead (openwakeword_amd.VAD(session=...) / "
                         "oww_push_vad): the gate, the score ring and the state handling of vad.py stay the same")

    convs = [n for n in nodes if n["op"] == "Conv" and len(n["inputs"]) >= 2 and n["inputs"][1] in inits]
    enc, basis = [], []
    for n in convs:
        w = inits[n["inputs"][1]]
        if w.ndim == 3 and w.shape[2] == 3:
            enc.append(n)
        elif w.ndim == 3 and w.shape[1] == 1:
            basis.append(n)                                   # an STFT written as a strided Conv1d over the waveform
    for n in basis:
        w = inits[n["inputs"][1]]
        hop = (n["attrs"].get("strides") or [0])[0]
        if w.shape[2] != W.VAD_N_FFT or hop != W.VAD_HOP:
            refuse(f"STFT basis of {w.shape[2]} samples / hop {hop}, kernels use {W.VAD_N_FFT} / {W.VAD_HOP}")
    if len(enc) != len(W.VAD_ENC):
        refuse(f"{len(enc)} kernel-3 Conv1d layers, expected {len(W.VAD_ENC)}")
    out_enc = []
    for n, (cin, cout, stride) in zip(enc, W.VAD_ENC):
        w = inits[n["inputs"][1]]                             # ONNX Conv1d weight: [cout, cin, k]
        st = (n["attrs"].get("strides") or [1])[0]
        pads = n["attrs"].get("pads") or [0, 0]
        if w.shape != (cout, cin, 3) or st != stride or list(pads) != [1, 1] or (n["attrs"].get("group") or 1) != 1:
            refuse(f"encoder Conv {tuple(w.shape)} stride {st} pads {pads}, expected {(cout, cin, 3)} stride {stride} pads [1, 1]")
        b = inits[n["inputs"][2]] if len(n["inputs"]) > 2 and n["inputs"][2] in inits else np.zeros(cout, np.float32)
        out_enc.append((np.ascontiguousarray(w.transpose(2, 1, 0), np.float32), np.asarray(b, np.float32).reshape(cout)))
    lstms = [n for n in nodes if n["op"] == "LSTM"]
    if len(lstms) != 2:
        refuse(f"{len(lstms)} LSTM nodes, expected 2 (one per layer)")
    out_lstm = []
    H = W.VAD_HID
    for n in lstms:
        if (n["attrs"].get("hidden_size") or 0) != H or (n["attrs"].get("direction") or "forward") != "forward":
            refuse(f"LSTM hidden_size {n['attrs'].get('hidden_size')} / direction {n['attrs'].get('direction')}, expected {H} / forward")
        try:
            wi, wr = inits[n["inputs"][1]], inits[n["inputs"][2]]            # [1, 4H, in], [1, 4H, H]; ONNX gate order i, o, f, c
            bb = inits[n["inputs"][3]] if len(n["inputs"]) > 3 and n["inputs"][3] in inits else np.zeros((1, 8 * H), np.float32)
        except KeyError:
            refuse("LSTM weights are not initializers")
        if wi.shape != (1, 4 * H, H) or wr.shape != (1, 4 * H, H) or bb.shape != (1, 8 * H):
            refuse(f"LSTM weights {tuple(wi.shape)} / {tuple(wr.shape)} / {tuple(bb.shape)}, expected (1, {4 * H}, {H}) x 2 and (1, {8 * H})")
        order = [0, 2, 3, 1]                                                # ours: i | f | g | o  <-  ONNX blocks i(0) o(1) f(2) c(3)
        rows = np.concatenate([wi[0], wr[0]], axis=1)                       # [4H, 2H]: columns x ; h
        rows = np.concatenate([rows[k * H:(k + 1) * H] for k in order], axis=0)
        bias = bb[0, :4 * H] + bb[0, 4 * H:]
        bias = np.concatenate([bias[k * H:(k + 1) * H] for k in order])
        out_lstm.append((np.ascontiguousarray(rows.T, np.float32), np.ascontiguousarray(bias, np.float32)))
    dec = None
    for n in nodes:
        if n["op"] in ("Gemm", "MatMul", "Conv") and len(n["inputs"]) >= 2 and n["inputs"][1] in inits and inits[n["inputs"][1]].size == H \
                and n not in enc:
            wd = np.asarray(inits[n["inputs"][1]], np.float32).reshape(H)
            bd = float(np.asarray(inits[n["inputs"][2]]).reshape(-1)[0]) if len(n["inputs"]) > 2 and n["inputs"][2] in inits else None
            if bd is None:                                    # nn.Linear as the exporter writes it for a 3-D input: MatMul, then Add(bias)
                adds = [m for m in nodes if m["op"] == "Add" and n["outputs"][0] in m["inputs"]]
                consts = [inits[i] for m in adds for i in m["inputs"] if i in inits and inits[i].size == 1]
                if len(adds) > 1 or len(consts) != len(adds):
                    refuse(f"the decoder's output feeds {[m['op'] for m in adds]} with non-scalar operands")
                bd = float(np.asarray(consts[0]).reshape(-1)[0]) if consts else 0.0
            dec = (wd, np.float32(bd))
    if dec is None or "Sigmoid" not in ops:
        refuse("no 64 -> 1 decoder followed by a Sigmoid")
    return {"enc": out_enc, "lstm": out_lstm, "dec": dec}
