"""
TEST INFRASTRUCTURE ONLY -- a small, generic ONNX evaluator in numpy behind the `onnxruntime` API the reference uses.

`oracle/fake_ort.py` lets the reference's own Python run with this repository's RESTATEMENT of the three networks at its backend
seams.  This module removes the restatement from that loop: it parses an `.onnx` file (own protobuf reader, nothing shared with
`openwakeword_amd.onnx_ingest`) and evaluates the graph operator by operator with the semantics of the ONNX operator specification,
knowing nothing about mel spectrograms, embedding networks or wake-word heads.  With it the reference's unmodified
`openwakeword.Model` (utils.py:84-93, model.py:153-159: `ort.InferenceSession(path).run(None, {name: x})`) runs on real model FILES --
here: files written by PyTorch's own exporter, the tool the reference exports its models with (train.py:144-165) --
`tests/golden/make_golden_onnx.py` stores what it returns, and the oracle (CPU) and the HIP `Model` loading the same files by path
(GPU) are held to those vectors.

Supported: the operators torch.onnx.export emits for these graphs (opsets 11..17).  float32 arithmetic, like the CPU provider.
"""
from __future__ import annotations

import struct
import types
from typing import Dict, List

import numpy as np


# ------------------------------------------------------------------------------------------------------------------- protobuf
def _varint(b: bytes, i: int):
    v, s = 0, 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, i
        s += 7


def _fields(b: bytes):
    """(field number, wire type, value) triples of one message; value = int (varint / fixed) or bytes (length-delimited)."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError(f"mini_ort: wire type {wt}")
        yield f, wt, v


def _sint64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_ints(wt, v) -> List[int]:
    if wt == 0:
        return [_sint64(v)]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_sint64(x))
    return out


_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


def _tensor(b: bytes):
    dims, dtype, raw, name = [], 1, None, ""
    floats, int32s, int64s, doubles = [], [], [], []
    for f, wt, v in _fields(b):
        if f == 1:
            dims += _packed_ints(wt, v)
        elif f == 2:
            dtype = v
        elif f == 4:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 5:
            int32s += _packed_ints(wt, v)
        elif f == 7:
            int64s += _packed_ints(wt, v)
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = v
        elif f == 10:
            doubles += list(struct.unpack(f"<{len(v) // 8}d", v)) if wt == 2 else [struct.unpack("<d", v)[0]]
        elif f in (13, 14) and v:
            raise ValueError("mini_ort: external tensor data is not supported")
    dt = _DTYPES[dtype]
    if raw is not None:
        a = np.frombuffer(raw, dtype=dt).copy()
    elif dtype == 1:
        a = np.array(floats, dtype=dt)
    elif dtype == 11:
        a = np.array(doubles, dtype=dt)
    elif dtype == 7:
        a = np.array(int64s, dtype=dt)
    else:
        a = np.array(int32s, dtype=dt)
    return name, a.reshape(dims)


def _attribute(b: bytes):
    name, out, ints, floats, strings = "", None, [], [], []
    seen = set()
    for f, wt, v in _fields(b):
        seen.add(f)
        if f == 1:
            name = v.decode()
        elif f == 2:
            out = struct.unpack("<f", v)[0]
        elif f == 3:
            out = _sint64(v)
        elif f == 4:
            out = v.decode(errors="replace")
        elif f == 5:
            out = _tensor(v)[1]
        elif f == 6:
            out = _graph(v)
        elif f == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8:
            ints += _packed_ints(wt, v)
        elif f == 9:
            strings.append(v.decode(errors="replace"))
    if 8 in seen:
        out = ints
    elif 7 in seen:
        out = floats
    elif 9 in seen:
        out = strings
    elif out is None:
        out = []                                            # an empty repeated field (e.g. axes = [])
    return name, out


def _value_info(b: bytes):
    name, shape = "", None
    for f, _, v in _fields(b):
        if f == 1:
            name = v.decode()
        elif f == 2:
            for f2, _, v2 in _fields(v):                    # TypeProto.tensor_type
                if f2 == 1:
                    for f3, _, v3 in _fields(v2):           # Tensor.shape
                        if f3 == 2:
                            shape = []
                            for f4, _, v4 in _fields(v3):   # TensorShapeProto.dim
                                if f4 == 1:
                                    d = None
                                    for f5, _, v5 in _fields(v4):
                                        d = _sint64(v5) if f5 == 1 else v5.decode()
                                    shape.append(d)
    return name, shape


def _node(b: bytes):
    n = {"inputs": [], "outputs": [], "op": "", "attrs": {}, "domain": ""}
    for f, _, v in _fields(b):
        if f == 1:
            n["inputs"].append(v.decode())
        elif f == 2:
            n["outputs"].append(v.decode())
        elif f == 4:
            n["op"] = v.decode()
        elif f == 5:
            k, a = _attribute(v)
            n["attrs"][k] = a
        elif f == 7:
            n["domain"] = v.decode()
    return n


def _graph(b: bytes):
    g = {"nodes": [], "init": {}, "inputs": [], "outputs": []}
    for f, _, v in _fields(b):
        if f == 1:
            g["nodes"].append(_node(v))
        elif f == 5:
            k, a = _tensor(v)
            g["init"][k] = a
        elif f == 11:
            g["inputs"].append(_value_info(v))
        elif f == 12:
            g["outputs"].append(_value_info(v))
    return g


def load(path: str):
    data = open(path, "rb").read()
    graph, opset = None, 0
    for f, _, v in _fields(data):
        if f == 7:
            graph = _graph(v)
        elif f == 8:
            dom, ver = "", 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    dom = v2.decode()
                elif f2 == 2:
                    ver = v2
            if dom in ("", "ai.onnx"):
                opset = ver
    if graph is None:
        raise ValueError(f"mini_ort: {path} holds no graph")
    graph["opset"] = opset
    return graph


# ------------------------------------------------------------------------------------------------------------------ operators
def _windows(x, kernel, strides, pads, fill):
    """x [N, C, *spatial] -> windows [N, C, *out, *kernel] (explicit padding first, value `fill`)."""
    nd = len(kernel)
    if any(pads):
        width = [(0, 0), (0, 0)] + [(pads[d], pads[nd + d]) for d in range(nd)]
        x = np.pad(x, width, constant_values=fill)
    w = np.lib.stride_tricks.sliding_window_view(x, tuple(kernel), axis=tuple(range(2, 2 + nd)))
    sl = (slice(None), slice(None)) + tuple(slice(None, None, s) for s in strides)
    return w[sl]


def _conv(x, w, b, a):
    nd = w.ndim - 2
    if a.get("group", 1) != 1 or any(d != 1 for d in a.get("dilations", [1] * nd)):
        raise NotImplementedError("mini_ort: grouped / dilated Conv")
    if a.get("auto_pad", "NOTSET") not in ("NOTSET", "VALID"):
        raise NotImplementedError("mini_ort: Conv auto_pad")
    win = _windows(x, w.shape[2:], a.get("strides", [1] * nd), a.get("pads", [0] * 2 * nd), 0.0)
    # win [N, C, *out, *k] x w [M, C, *k] -> [N, *out, M]
    y = np.tensordot(win, w, axes=([1] + list(range(2 + nd, 2 + 2 * nd)), [1] + list(range(2, 2 + nd))))
    y = np.moveaxis(y, -1, 1)
    if b is not None:
        y = y + b.reshape((1, -1) + (1,) * nd)
    return y.astype(x.dtype)


def _maxpool(x, a):
    k = a["kernel_shape"]
    nd = len(k)
    if a.get("ceil_mode", 0) or any(d != 1 for d in a.get("dilations", [1] * nd)):
        raise NotImplementedError("mini_ort: MaxPool ceil_mode / dilations")
    win = _windows(x, k, a.get("strides", [1] * nd), a.get("pads", [0] * 2 * nd), -np.inf)
    return win.max(axis=tuple(range(-nd, 0)))


def _lstm(x, a):
    """ONNX LSTM, default activations: X [T, B, in] (layout 0), W [D, 4H, in], R [D, 4H, H], B [D, 8H] (D = 1: forward | reverse, 2:
    bidirectional), gate order i o f c, optional sequence_lens (unsupported), initial_h / initial_c [D, B, H] -> Y [T, D, B, H] (the
    reverse direction's rows at the positions of their inputs), Y_h, Y_c [D, B, H]."""
    direction = a.get("direction", "forward")
    if direction not in ("forward", "reverse", "bidirectional") or a.get("layout", 0) != 0 or a.get("input_forget", 0) or a.get("activations"):
        raise NotImplementedError("mini_ort: LSTM direction / layout / input_forget / activations")
    X = x[0]
    H = int(a["hidden_size"])
    D = 2 if direction == "bidirectional" else 1
    if len(x) > 4 and x[4] is not None:
        raise NotImplementedError("mini_ort: LSTM sequence_lens")
    if len(x) > 7 and x[7] is not None:
        raise NotImplementedError("mini_ort: LSTM peepholes")
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    Y = np.zeros((X.shape[0], D, X.shape[1], H), X.dtype)
    Yh, Yc = [], []
    for d in range(D):
        Wm, R = x[1][d], x[2][d]
        Bv = x[3][d] if len(x) > 3 and x[3] is not None else np.zeros(8 * H, X.dtype)
        h = x[5][d] if len(x) > 5 and x[5] is not None else np.zeros((X.shape[1], H), X.dtype)
        c = x[6][d] if len(x) > 6 and x[6] is not None else np.zeros((X.shape[1], H), X.dtype)
        rev = direction == "reverse" or d == 1
        for t in (range(X.shape[0] - 1, -1, -1) if rev else range(X.shape[0])):
            z = X[t] @ Wm.T + h @ R.T + Bv[:4 * H] + Bv[4 * H:]
            i, o, f, g = (z[:, k * H:(k + 1) * H] for k in range(4))
            c = sig(f) * c + sig(i) * np.tanh(g)
            h = sig(o) * np.tanh(c)
            Y[t, d] = h
        Yh.append(h); Yc.append(c)
    return Y, np.stack(Yh).astype(X.dtype), np.stack(Yc).astype(X.dtype)


def _axes(node, inputs, opset_from_input: int, opset: int):
    if opset >= opset_from_input and len(inputs) > 1 and inputs[1] is not None:
        return [int(v) for v in np.asarray(inputs[1]).reshape(-1)]
    ax = node["attrs"].get("axes")
    return None if ax is None else [int(v) for v in ax]


def _reduce(fn, node, inputs, opset, from_input):
    ax = _axes(node, inputs, from_input, opset)
    keep = bool(node["attrs"].get("keepdims", 1))
    return fn(inputs[0], axis=None if not ax else tuple(ax), keepdims=keep)


def _run_node(n, x: List, opset: int):
    op, a = n["op"], n["attrs"]
    if op == "Conv":
        return _conv(x[0], x[1], x[2] if len(x) > 2 else None, a)
    if op == "MaxPool":
        return _maxpool(x[0], a)
    if op == "BatchNormalization":
        sh = (1, -1) + (1,) * (x[0].ndim - 2)
        return ((x[0] - x[3].reshape(sh)) / np.sqrt(x[4].reshape(sh) + np.float32(a.get("epsilon", 1e-5))) * x[1].reshape(sh) + x[2].reshape(sh)).astype(x[0].dtype)
    if op == "Gemm":
        A = x[0].T if a.get("transA", 0) else x[0]
        B = x[1].T if a.get("transB", 0) else x[1]
        y = np.float32(a.get("alpha", 1.0)) * (A @ B)
        if len(x) > 2 and x[2] is not None:
            y = y + np.float32(a.get("beta", 1.0)) * x[2]
        return y.astype(x[0].dtype)
    if op == "MatMul":
        return x[0] @ x[1]
    if op == "LayerNormalization":
        axis = a.get("axis", -1)
        axes = tuple(range(axis if axis >= 0 else x[0].ndim + axis, x[0].ndim))
        m = x[0].mean(axis=axes, keepdims=True)
        d = x[0] - m
        v = (d * d).mean(axis=axes, keepdims=True)
        y = d / np.sqrt(v + np.float32(a.get("epsilon", 1e-5))) * x[1]
        return (y + x[2] if len(x) > 2 and x[2] is not None else y).astype(x[0].dtype)
    if op == "ReduceMean":
        return _reduce(np.mean, n, x, opset, 18)
    if op == "ReduceMax":
        return _reduce(np.max, n, x, opset, 18)
    if op == "ReduceSum":
        return _reduce(np.sum, n, x, opset, 13)
    if op == "Softmax":
        axis = a.get("axis", -1 if opset >= 13 else 1)
        if opset < 13:                                      # coerced to 2-D around `axis`
            sh = x[0].shape
            flat = x[0].reshape(int(np.prod(sh[:axis])), -1)
            e = np.exp(flat - flat.max(axis=1, keepdims=True))
            return (e / e.sum(axis=1, keepdims=True)).reshape(sh)
        e = np.exp(x[0] - x[0].max(axis=axis, keepdims=True))
        return e / e.sum(axis=axis, keepdims=True)
    if op in ("Add", "Sub", "Mul", "Div", "Pow"):
        f = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply, "Div": np.divide, "Pow": np.power}[op]
        y = f(x[0], x[1])
        return y.astype(x[0].dtype) if np.issubdtype(np.asarray(x[0]).dtype, np.floating) else y
    if op == "Max":
        y = x[0]
        for o in x[1:]:
            y = np.maximum(y, o)
        return y
    if op == "Min":
        y = x[0]
        for o in x[1:]:
            y = np.minimum(y, o)
        return y
    if op == "Sqrt":
        return np.sqrt(x[0])
    if op == "Log":
        with np.errstate(divide="ignore"):
            return np.log(x[0])
    if op == "Exp":
        return np.exp(x[0])
    if op == "Neg":
        return -x[0]
    if op == "Abs":
        return np.abs(x[0])
    if op == "Tanh":
        return np.tanh(x[0])
    if op == "Relu":
        return np.maximum(x[0], 0)
    if op == "Sigmoid":
        return (1.0 / (1.0 + np.exp(-x[0]))).astype(x[0].dtype)
    if op == "LeakyRelu":
        return np.where(x[0] < 0, np.float32(a.get("alpha", 0.01)) * x[0], x[0]).astype(x[0].dtype)
    if op == "Clip":
        lo = x[1] if len(x) > 1 and x[1] is not None else a.get("min")
        hi = x[2] if len(x) > 2 and x[2] is not None else a.get("max")
        y = x[0]
        if lo is not None:
            y = np.maximum(y, np.asarray(lo, dtype=y.dtype))
        if hi is not None:
            y = np.minimum(y, np.asarray(hi, dtype=y.dtype))
        return y
    if op == "Transpose":
        return np.transpose(x[0], a.get("perm"))
    if op == "Reshape":
        shape = [int(v) for v in x[1]]
        shape = [x[0].shape[i] if v == 0 and not a.get("allowzero", 0) else v for i, v in enumerate(shape)]
        return x[0].reshape(shape)
    if op == "Flatten":
        axis = a.get("axis", 1)
        return x[0].reshape(int(np.prod(x[0].shape[:axis], dtype=np.int64)), -1)
    if op == "Unsqueeze":
        y = x[0]
        for ax in sorted(_axes(n, x, 13, opset)):
            y = np.expand_dims(y, ax)
        return y
    if op == "Squeeze":
        ax = _axes(n, x, 13, opset)
        return np.squeeze(x[0], axis=None if not ax else tuple(ax))
    if op == "Identity":
        return x[0]
    if op == "Cast":
        return np.asarray(x[0]).astype(_DTYPES[a["to"]])
    if op == "Constant":
        for k in ("value", "value_float", "value_int", "value_floats", "value_ints"):
            if k in a:
                v = a[k]
                return v if isinstance(v, np.ndarray) else np.asarray(v, dtype=np.float32 if "float" in k else np.int64)
        raise NotImplementedError("mini_ort: Constant without a value")
    if op == "Shape":
        return np.asarray(x[0].shape, dtype=np.int64)
    if op == "Gather":
        return np.take(x[0], np.asarray(x[1], dtype=np.int64), axis=a.get("axis", 0))
    if op == "Concat":
        return np.concatenate([np.atleast_1d(v) for v in x], axis=a.get("axis", 0))
    if op == "Slice":
        starts, ends = x[1], x[2]
        axes = x[3] if len(x) > 3 and x[3] is not None else range(len(starts))
        steps = x[4] if len(x) > 4 and x[4] is not None else [1] * len(starts)
        sl = [slice(None)] * x[0].ndim
        for s, e, ax, st in zip(starts, ends, axes, steps):
            sl[int(ax)] = slice(int(s), None if int(e) >= np.iinfo(np.int64).max // 2 else int(e), int(st))
        return x[0][tuple(sl)]
    if op == "ConstantOfShape":
        v = a.get("value")
        return np.full([int(d) for d in x[0]], v.reshape(-1)[0] if v is not None else np.float32(0))
    if op == "Expand":
        return x[0] * np.ones([int(d) for d in x[1]], dtype=x[0].dtype)
    if op == "LSTM":
        return _lstm(x, a)
    if op in ("Greater", "Less", "GreaterOrEqual", "LessOrEqual", "Equal"):
        f = {"Greater": np.greater, "Less": np.less, "GreaterOrEqual": np.greater_equal, "LessOrEqual": np.less_equal, "Equal": np.equal}[op]
        return f(x[0], x[1])
    if op == "Where":
        return np.where(x[0], x[1], x[2])
    if op == "Not":
        return np.logical_not(x[0])
    if op == "And":
        return np.logical_and(x[0], x[1])
    raise NotImplementedError(f"mini_ort: operator {op}")


def evaluate(graph, feeds: Dict[str, np.ndarray], outer: Dict[str, np.ndarray] = None, opset: int = None) -> List[np.ndarray]:
    env = dict(outer or {})          # (a subgraph sees the tensors of the enclosing scopes)
    env.update(graph["init"])
    env.update(feeds)
    opset = graph.get("opset", opset) if opset is None else opset
    for n in graph["nodes"]:
        if n["domain"] not in ("", "ai.onnx"):
            raise NotImplementedError(f"mini_ort: operator domain {n['domain']}")
        xs = [env[i] if i else None for i in n["inputs"]]
        if n["op"] == "If":
            cond = bool(np.asarray(xs[0]).reshape(-1)[0])
            y = tuple(evaluate(n["attrs"]["then_branch" if cond else "else_branch"], {}, env, opset))
        else:
            y = _run_node(n, xs, opset)
        ys = y if isinstance(y, tuple) else (y,)
        for name, v in zip(n["outputs"], ys):
            env[name] = np.asarray(v)
    return [env[name] for name, _ in graph["outputs"]]


# ------------------------------------------------------------------------------------------------------- the onnxruntime facade
class SessionOptions:
    inter_op_num_threads = 1
    intra_op_num_threads = 1


class _IO:
    def __init__(self, name, shape):
        self.name, self.shape = name, shape


REDIRECT: Dict[str, str] = {}       # basename -> actual file, for models the reference resolves inside its own (read-only) package


class InferenceSession:
    def __init__(self, path, sess_options=None, providers=None):
        import os
        path = REDIRECT.get(os.path.basename(path), path)
        self._g = load(path)
        self._providers = list(providers or ["CPUExecutionProvider"])
        init = self._g["init"]
        self._inputs = [_IO(n, s) for n, s in self._g["inputs"] if n not in init]
        self._outputs = [_IO(n, s) for n, s in self._g["outputs"]]

    def run(self, output_names, feeds):
        want = {i.name for i in self._inputs}
        if set(feeds) != want:
            raise ValueError(f"mini_ort: feeds {sorted(feeds)} do not match the graph inputs {sorted(want)}")
        outs = evaluate(self._g, {k: np.asarray(v) for k, v in feeds.items()})
        if output_names:
            names = [o.name for o in self._outputs]
            return [outs[names.index(n)] for n in output_names]
        return outs

    def get_inputs(self):
        return self._inputs

    def get_outputs(self):
        return self._outputs

    def get_providers(self):
        return self._providers


def as_module() -> types.ModuleType:
    m = types.ModuleType("onnxruntime")
    m.SessionOptions = SessionOptions
    m.InferenceSession = InferenceSession
    return m
