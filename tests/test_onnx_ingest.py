"""onnx_ingest: the reference's model-file format read without the `onnx` package.  No real .onnx file exists offline, so the
reader is pinned by a round trip: a minimal protobuf writer (below) emits ModelProto files with the reference's graph
structures (train.py:56-83 heads, the 20-conv embedding model), onnx_ingest reads them back, and the oracle must give the
same outputs from the re-read weights."""
import os
import struct

import numpy as np
import pytest

from oracle import oww_oracle as O
from openwakeword_amd import onnx_ingest, weights as W


# ---- minimal ONNX (protobuf) writer ---------------------------------------------------------------------------------
def _vi(x):
    x &= (1 << 64) - 1
    out = b""
    while True:
        b = x & 0x7F
        x >>= 7
        out += bytes([b | (0x80 if x else 0)])
        if not x:
            return out


def _ld(fno, payload):
    return _vi((fno << 3) | 2) + _vi(len(payload)) + payload


def _tensor(name, arr, raw=True):
    arr = np.ascontiguousarray(arr, np.float32)
    t = b"".join(_vi((1 << 3) | 0) + _vi(d) for d in arr.shape) + _vi((2 << 3) | 0) + _vi(1) + _ld(8, name.encode())
    return t + (_ld(9, arr.tobytes()) if raw else _ld(4, arr.tobytes()))


def _attr_i(name, v):
    return _ld(1, name.encode()) + _vi((3 << 3) | 0) + _vi(v) + _vi((20 << 3) | 0) + _vi(2)


def _attr_f(name, v):
    return _ld(1, name.encode()) + _vi((2 << 3) | 5) + struct.pack("<f", v) + _vi((20 << 3) | 0) + _vi(1)


def _node(op, inputs, outputs, attrs=()):
    return b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs) + \
        _ld(3, (op + "_" + outputs[0]).encode()) + _ld(4, op.encode()) + b"".join(_ld(5, a) for a in attrs)


def _model(nodes, inits):
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"g") + b"".join(_ld(5, t) for t in inits)
    return _vi((1 << 3) | 0) + _vi(8) + _ld(7, graph)


def write_head(path, head, layernorm_op=True, use_matmul=False, raw=True, tail=None, hidden_act="Relu", ln_eps=1e-5):
    nodes, inits, cur = [_node("Flatten", ["x"], ["f0"], [_attr_i("axis", 1)])], [], "f0"
    nets = [head["net"]] + ([head["net2"]] if head["kind"] == "gated" else [])
    for k, net in enumerate(nets):
        cur = "f0"
        for li in (1, 2, 3):
            w, b = net[f"w{li}"], net[f"b{li}"]
            if use_matmul:
                inits += [_tensor(f"n{k}w{li}", w, raw), _tensor(f"n{k}b{li}", b, raw)]
                nodes += [_node("MatMul", [cur, f"n{k}w{li}"], [f"n{k}m{li}"]), _node("Add", [f"n{k}m{li}", f"n{k}b{li}"], [f"n{k}l{li}"])]
            else:
                inits += [_tensor(f"n{k}w{li}", w.T, raw), _tensor(f"n{k}b{li}", b, raw)]       # torch Linear: [out, in], transB=1
                nodes.append(_node("Gemm", [cur, f"n{k}w{li}", f"n{k}b{li}"], [f"n{k}l{li}"], [_attr_i("transB", 1)]))
            cur = f"n{k}l{li}"
            if li < 3:
                ln = net.get(f"ln{li}")
                if ln is not None:
                    inits += [_tensor(f"n{k}g{li}", ln[0], raw), _tensor(f"n{k}be{li}", ln[1], raw)]
                    if layernorm_op:
                        nodes.append(_node("LayerNormalization", [cur, f"n{k}g{li}", f"n{k}be{li}"], [f"n{k}n{li}"], [_attr_f("epsilon", ln_eps)]))
                    else:                                           # the decomposition older opsets export (torch < opset 17)
                        t = f"n{k}L{li}"
                        inits += [_tensor(t + "two", np.array(2.0, np.float32), raw), _tensor(t + "eps", np.array(ln_eps, np.float32), raw)]
                        nodes += [_node("ReduceMean", [cur], [t + "m"], [_attr_ints("axes", [-1]), _attr_i("keepdims", 1)]),
                                  _node("Sub", [cur, t + "m"], [t + "d"]), _node("Pow", [t + "d", t + "two"], [t + "p"]),
                                  _node("ReduceMean", [t + "p"], [t + "v"], [_attr_ints("axes", [-1]), _attr_i("keepdims", 1)]),
                                  _node("Add", [t + "v", t + "eps"], [t + "ve"]), _node("Sqrt", [t + "ve"], [t + "s"]),
                                  _node("Div", [t + "d", t + "s"], [t + "q"]),
                                  _node("Mul", [t + "q", f"n{k}g{li}"], [f"n{k}q{li}"]), _node("Add", [f"n{k}q{li}", f"n{k}be{li}"], [f"n{k}n{li}"])]
                    cur = f"n{k}n{li}"
                nodes.append(_node(hidden_act, [cur], [f"n{k}r{li}"]))
                cur = f"n{k}r{li}"
        if tail is not None:
            for t_i, op in enumerate(tail):
                nodes.append(_node(op, [cur], [f"n{k}t{t_i}"]))
                cur = f"n{k}t{t_i}"
        elif head["kind"] == "multiclass":
            nodes += [_node("Relu", [cur], [f"n{k}rr"]), _node("Softmax", [f"n{k}rr"], [f"n{k}out"])]
        else:
            nodes.append(_node("Sigmoid", [cur], [f"n{k}out"]))
    if head["kind"] == "gated" and tail is None:               # docs/models/hey_jarvis.md:38: where(first > 0.5, second, first)
        inits.append(_tensor("half", np.array([0.5], np.float32), raw))
        nodes += [_node("Greater", ["n0out", "half"], ["gate"]), _node("Where", ["gate", "n1out", "n0out"], ["score"])]
    open(path, "wb").write(_model(nodes, inits))


def _attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_vi((8 << 3) | 0) + _vi(v) for v in vals) + _vi((20 << 3) | 0) + _vi(7)


def _attr_s(name, v):
    return _ld(1, name.encode()) + _ld(4, v.encode()) + _vi((20 << 3) | 0) + _vi(3)


def write_embedding(path, emb, fold_last_bns=0):
    """Plain idiom, NCHW from the input on (what torch.onnx.export of the notebook's network would give): Conv with `pads`,
    BatchNormalization nodes (or folded into the last few convolutions), LeakyRelu + Clip(min) attribute form, MaxPool."""
    nodes, inits, cur = [], [], "x"
    n = len(emb["conv"])
    for li, w in enumerate(emb["conv"]):
        kh, kw, _ci, _co, pool = W.CNN_TOPOLOGY[li]
        inits.append(_tensor(f"w{li}", np.transpose(w, (3, 2, 0, 1))))                       # HWIO -> OIHW
        folded = li < n - 1 and li >= n - 1 - fold_last_bns and li > 0
        pads = [_attr_ints("pads", [0, 1, 0, 1])] if kw == 3 else []
        if folded:
            scale, shift = W.bn_scale_shift(emb["bn"][li])
            inits[-1] = _tensor(f"w{li}", np.transpose(w * scale, (3, 2, 0, 1)))
            inits.append(_tensor(f"b{li}", shift))
            nodes.append(_node("Conv", [cur, f"w{li}", f"b{li}"], [f"c{li}"], pads))
        else:
            nodes.append(_node("Conv", [cur, f"w{li}"], [f"c{li}"], pads))
        cur = f"c{li}"
        if li == 0:
            nodes.append(_node("Relu", [cur], ["r0"]))
            cur = "r0"
        if li < n - 1 and not folded:
            for nm, arr in zip("gbmv", emb["bn"][li]):
                inits.append(_tensor(f"{nm}{li}", arr))
            nodes.append(_node("BatchNormalization", [cur, f"g{li}", f"b{li}", f"m{li}", f"v{li}"], [f"n{li}"], [_attr_f("epsilon", W.BN_EPS)]))
            cur = f"n{li}"
        if li < n - 1:
            nodes += [_node("LeakyRelu", [cur], [f"l{li}"], [_attr_f("alpha", 0.2)]), _node("Clip", [f"l{li}"], [f"a{li}"], [_attr_f("min", -0.4)])]
            cur = f"a{li}"
            if pool:
                nodes.append(_node("MaxPool", [cur], [f"p{li}"], [_attr_ints("kernel_shape", list(pool)), _attr_ints("strides", list(pool))]))
                cur = f"p{li}"
    open(path, "wb").write(_model(nodes, inits))


# ---- tests -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kw", [("alexa", {}), ("hey_jarvis", {}), ("timer", {}), ("weather", dict(layernorm_op=False, use_matmul=True, raw=False))])
def test_head_round_trip(tmp_path, name, kw):
    head = W.synthetic_head(name, 77)
    path = os.path.join(tmp_path, name + ".onnx")
    write_head(path, head, **kw)
    got = onnx_ingest.load_head(path)
    assert (got["kind"], got["T"], got["hidden"], got["n_out"]) == (head["kind"], head["T"], head["hidden"], head["n_out"])
    feats = np.random.default_rng(1).normal(0, 2, (5, head["T"], 96)).astype(np.float32)
    np.testing.assert_array_equal(O.head_stage(feats, got, np.float32), O.head_stage(feats, head, np.float32))


def test_head_activations_are_verified_not_assumed(tmp_path):
    """A graph whose activations differ from what the kernels apply must be refused, not scored differently from onnxruntime:
    train.py's multiclass branch ends in a bare ReLU (train.py:81-83); a Tanh hidden layer; a Softmax without its ReLU."""
    path = os.path.join(tmp_path, "odd.onnx")
    timer = W.synthetic_head("timer", 77)
    write_head(path, timer, tail=["Relu"])
    with pytest.raises(ValueError, match="unsupported output activation"):
        onnx_ingest.load_head(path)
    write_head(path, timer, tail=["Softmax"])
    with pytest.raises(ValueError, match="unsupported output activation"):
        onnx_ingest.load_head(path)
    write_head(path, W.synthetic_head("alexa", 77), hidden_act="Tanh")
    with pytest.raises(ValueError, match="exactly one Relu"):
        onnx_ingest.load_head(path)
    write_head(path, W.synthetic_head("alexa", 77), tail=["Relu", "Softmax"])          # n_out == 1 with a softmax tail loads as multiclass
    assert onnx_ingest.load_head(path)["kind"] == "multiclass"


def test_embedding_round_trip(tmp_path):
    emb = W.synthetic_embedding(55)
    x = np.random.default_rng(2).normal(10, 1.5, (2, 76, 32, 1)).astype(np.float32)
    want = O.embedding_stage(x, emb, np.float64)
    for fold in (0, 3):
        path = os.path.join(tmp_path, f"emb{fold}.onnx")
        write_embedding(path, emb, fold_last_bns=fold)
        got = onnx_ingest.load_embedding(path)
        assert W.embedding_param_count(got) == W.embedding_param_count(emb) == 332088
        np.testing.assert_allclose(O.embedding_stage(x, got, np.float64), want, rtol=0, atol=1e-5 if fold else 1e-9)  # folding rounds w*scale to fp32


def test_unrecognised_graphs_raise(tmp_path):
    path = os.path.join(tmp_path, "bad.onnx")
    open(path, "wb").write(_model([_node("Relu", ["x"], ["y"])], []))
    with pytest.raises(ValueError):
        onnx_ingest.load_head(path)
    with pytest.raises(ValueError):
        onnx_ingest.load_embedding(path)
    open(path, "wb").write(b"\x00\x01garbage")
    with pytest.raises(ValueError):
        onnx_ingest.load_graph(path)


def test_melspectrogram_check(tmp_path):
    path = os.path.join(tmp_path, "mel.onnx")
    open(path, "wb").write(_model([_node("MatMul", ["p", "fb"], ["m"])], [_tensor("fb", W.mel_filterbank())]))
    assert onnx_ingest.check_melspectrogram(path)["filterbank_max_abs_diff"] == 0.0


# ---- voice-activity network (onnx_ingest.load_vad): recognised by structure, anything else refused -------------------------
def write_vad(path, vad, lstm_hidden=64, strides=(1, 2, 2, 1), with_basis=True, basis_len=256):
    """The stand-in architecture of csrc/owwhip_vad.h as an ONNX graph: [STFT basis Conv] -> 4 x (Conv1d k=3 + Relu) -> 2 x LSTM
    (ONNX gate order i, o, f, c) -> Relu -> 1x1 Conv (64 -> 1) -> Sigmoid."""
    nodes, inits, cur = [], [], "input"
    if with_basis:
        inits.append(_tensor("basis", np.zeros((258, 1, basis_len), np.float32)))
        nodes.append(_node("Conv", [cur, "basis"], ["spec"], [_attr_ints("strides", [64])]))
        cur = "spec"
    for li, ((w, b), st) in enumerate(zip(vad["enc"], strides)):
        inits += [_tensor(f"ew{li}", np.transpose(w, (2, 1, 0))), _tensor(f"eb{li}", b)]       # [k, cin, cout] -> [cout, cin, k]
        nodes += [_node("Conv", [cur, f"ew{li}", f"eb{li}"], [f"e{li}"], [_attr_ints("strides", [st]), _attr_ints("pads", [1, 1])]),
                  _node("Relu", [f"e{li}"], [f"er{li}"])]
        cur = f"er{li}"
    H = 64
    inv = [0, 3, 1, 2]                                       # ONNX block k (i, o, f, c) <- ours (i, f, g, o)
    for li, (w, b) in enumerate(vad["lstm"]):
        rows = w.T                                           # [4H, 2H] in our gate order
        rows = np.concatenate([rows[k * H:(k + 1) * H] for k in inv], axis=0)
        bias = np.concatenate([b[k * H:(k + 1) * H] for k in inv])
        inits += [_tensor(f"lw{li}", rows[None, :, :H]), _tensor(f"lr{li}", rows[None, :, H:]),
                  _tensor(f"lb{li}", np.concatenate([bias, np.zeros(4 * H, np.float32)])[None])]
        nodes.append(_node("LSTM", [cur, f"lw{li}", f"lr{li}", f"lb{li}"], [f"l{li}"], [_attr_i("hidden_size", lstm_hidden), _attr_s("direction", "forward")]))
        cur = f"l{li}"
    wd, bd = vad["dec"]
    inits += [_tensor("dw", np.asarray(wd, np.float32).reshape(1, 64, 1)), _tensor("db", np.array([bd], np.float32))]
    nodes += [_node("Relu", [cur], ["dr"]), _node("Conv", ["dr", "dw", "db"], ["do"]), _node("Sigmoid", ["do"], ["out"])]
    open(path, "wb").write(_model(nodes, inits))


def test_vad_round_trip_and_refusals(tmp_path):
    from oracle import vad_standin as VS
    vad = W.synthetic_vad(31)
    path = os.path.join(tmp_path, "silero_vad.onnx")
    write_vad(path, vad)
    got = onnx_ingest.load_vad(path)
    for (w0, b0), (w1, b1) in zip(vad["enc"] + vad["lstm"], got["enc"] + got["lstm"]):
        np.testing.assert_array_equal(w0, w1)
        np.testing.assert_array_equal(b0, b1)
    np.testing.assert_array_equal(vad["dec"][0], got["dec"][0])
    assert float(vad["dec"][1]) == float(got["dec"][1])
    # ... and the re-read weights drive the restated network to the same scores
    x = (np.random.default_rng(5).normal(0, 3000, 1280)).astype(np.int16)
    a, b = VS.StandinVadSession(vad), VS.StandinVadSession(got)
    h = c = np.zeros((2, 1, 64), np.float32)
    feed = {"input": (x[:640] / 32767).astype(np.float32)[None], "h": h, "c": c, "sr": np.array(16000)}
    np.testing.assert_array_equal(a.run(None, feed)[0], b.run(None, feed)[0])
    # anything that is not exactly this architecture is refused, naming what was found and the host-side way out
    for kw, why in ((dict(strides=(1, 1, 2, 1)), "stride"), (dict(lstm_hidden=128), "hidden_size"), (dict(basis_len=512), "STFT basis")):
        write_vad(path, vad, **kw)
        with pytest.raises(ValueError, match=why):
            onnx_ingest.load_vad(path)
    open(path, "wb").write(_model([_node("Relu", ["input"], ["y"]), _node("Sigmoid", ["y"], ["out"])], []))
    with pytest.raises(ValueError, match="oww_push_vad"):
        onnx_ingest.load_vad(path)
    write_head(path, W.synthetic_head("alexa", 77))         # a wake-word head is not a VAD
    with pytest.raises(ValueError, match="operators found"):
        onnx_ingest.load_vad(path)


# =====================================================================================================================
# Second, independent writer: the idioms the REAL files are likely to use.  It shares nothing with the
# writer above -- its own protobuf encoder (packed dims, int64 tensors, ValueInfo inputs / outputs, opset imports, Constant
# nodes, graph-valued attributes) -- and emits
#   * tf2onnx-style embedding graphs (the notebook converts a Keras model): NHWC input, one Transpose to NCHW, explicit Pad
#     nodes (opset-11 form: pads as an int64 input) or Conv pads / auto_pad, BatchNorm folded into the Conv bias for layers
#     1-18 but a standalone BatchNormalization (or a per-channel Mul + Add) after conv0's Relu, the activation as
#     Max(Mul(x, 0.2), x) -> Max(., -0.4) or as LeakyRelu + Clip with tensor bounds, weights through Constant nodes or stored
#     HWIO behind a Transpose, a Transpose / Squeeze / Reshape tail;
#   * torch.onnx.export-style heads: Reshape or Flatten, Gemm with transB, opset-17 LayerNormalization AND its
#     decomposition (Mul(d, d) instead of Pow), the hey_jarvis gate as Greater + Where and as an If with subgraphs.
# Each variant must load to the same weights as the plain writer's file, or be refused naming what was found.
class PB:
    """Tiny protobuf message builder: PB().i(field, int).f(field, float).b(field, bytes | str | PB) ... .done()"""

    def __init__(self):
        self.buf = bytearray()

    @staticmethod
    def varint(x: int) -> bytes:
        x &= 0xFFFFFFFFFFFFFFFF
        out = bytearray()
        while x >= 0x80:
            out.append((x & 0x7F) | 0x80)
            x >>= 7
        out.append(x)
        return bytes(out)

    def i(self, field, x):
        self.buf += self.varint(field << 3) + self.varint(int(x))
        return self

    def f(self, field, x):
        self.buf += self.varint((field << 3) | 5) + struct.pack("<f", float(x))
        return self

    def b(self, field, payload):
        if isinstance(payload, PB):
            payload = payload.done()
        elif isinstance(payload, str):
            payload = payload.encode()
        self.buf += self.varint((field << 3) | 2) + self.varint(len(payload)) + bytes(payload)
        return self

    def done(self) -> bytes:
        return bytes(self.buf)


class G2:
    """Graph under construction (writer 2)."""
    FLOAT, INT64 = 1, 7

    def __init__(self, in_name="input_1", opset=13):
        self.nodes, self.inits, self.opset, self.n = [], [], opset, 0
        self.inputs, self.outputs = [in_name], []

    def name(self, stem="t"):
        self.n += 1
        return f"{stem}:{self.n}"

    @staticmethod
    def tensor(name, arr, as_float_data=False):
        arr = np.asarray(arr)
        t = PB()
        if arr.ndim:
            t.b(1, b"".join(PB.varint(d) for d in arr.shape))                 # packed dims
        if arr.dtype.kind == "i":
            t.i(2, G2.INT64).b(8, name).b(9, arr.astype("<i8").tobytes())
        elif as_float_data:
            t.i(2, G2.FLOAT).b(4, arr.astype("<f4").tobytes()).b(8, name)     # float_data, packed
        else:
            t.i(2, G2.FLOAT).b(8, name).b(9, arr.astype("<f4").tobytes())
        return t

    def const(self, arr, stem="const", via_node=False, as_float_data=False):
        nm = self.name(stem)
        if via_node:                                                            # tf2onnx leaves many constants as Constant nodes
            attr = PB().b(1, "value").b(5, self.tensor("", arr, as_float_data)).i(20, 4)
            self.nodes.append(PB().b(2, nm).b(3, self.name("Constant")).b(4, "Constant").b(5, attr))
        else:
            self.inits.append(self.tensor(nm, arr, as_float_data))
        return nm

    def op(self, op, inputs, n_out=1, out=None, **attrs):
        outs = [out] if out else [self.name(op) for _ in range(n_out)]
        nd = PB()
        for x in inputs:
            nd.b(1, x)
        for o in outs:
            nd.b(2, o)
        nd.b(3, self.name("node")).b(4, op)
        for k, v in attrs.items():
            a = PB().b(1, k)
            if isinstance(v, float):
                a.f(2, v).i(20, 1)
            elif isinstance(v, int):
                a.i(3, v).i(20, 2)
            elif isinstance(v, str):
                a.b(4, v).i(20, 3)
            elif isinstance(v, G2):
                a.b(6, v.graph()).i(20, 5)
            else:
                a.b(8, b"".join(PB.varint(int(q)) for q in v)).i(20, 7)        # packed ints
            nd.b(5, a)
        self.nodes.append(nd)
        return outs[0] if n_out == 1 else outs

    def graph(self) -> PB:
        g = PB()
        for nd in self.nodes:
            g.b(1, nd)
        g.b(2, "graph2")
        for t in self.inits:
            g.b(5, t)
        for nm in self.inputs:
            g.b(11, PB().b(1, nm))
        for nm in self.outputs:
            g.b(12, PB().b(1, nm))
        return g

    def save(self, path):
        m = PB().i(1, 8).b(2, "writer2").b(8, PB().b(1, "").i(2, self.opset)).b(7, self.graph())
        open(path, "wb").write(m.done())


def write_embedding_tf2onnx(path, emb, pad="node", act="maxmul", bn0="node", weights="const_node", tail="transpose_reshape",
                            leaky=0.2, floor=-0.4, time_pad=0, pool_override=None, extra_op=None):
    g = G2("input_1", opset=13)
    cur = g.op("Transpose", ["input_1"], perm=[0, 3, 1, 2])                                   # NHWC -> NCHW
    n = len(emb["conv"])
    for li, w in enumerate(emb["conv"]):
        kh, kw, _ci, co, pool = W.CNN_TOPOLOGY[li]
        attrs = dict(kernel_shape=[kh, kw], strides=[1, 1], dilations=[1, 1], group=1)
        if kw == 3 or time_pad:
            if pad == "node":                                                                 # ZeroPadding2D -> Pad (pads as an input)
                cur = g.op("Pad", [cur, g.const(np.array([0, 0, time_pad, 1 if kw == 3 else 0] * 2, np.int64), "pads")], mode="constant")
            elif pad == "attr":
                attrs["pads"] = [time_pad, 1 if kw == 3 else 0] * 2
            elif pad == "auto" and kh == 1:
                attrs["auto_pad"] = "SAME_UPPER"
            else:
                attrs["pads"] = [time_pad, 1 if kw == 3 else 0] * 2
        oihw = np.transpose(w, (3, 2, 0, 1))
        ins = [cur]
        fold = 0 < li < n - 1
        if fold:
            scale, shift = W.bn_scale_shift(emb["bn"][li])
            oihw = oihw * scale[:, None, None, None]
        if weights == "hwio_transpose":                                                       # stored HWIO, transposed in the graph
            ins.append(g.op("Transpose", [g.const(np.transpose(oihw, (2, 3, 1, 0)), "kernel")], perm=[3, 2, 0, 1]))
        else:
            ins.append(g.const(oihw, "kernel", via_node=(weights == "const_node" and li % 2 == 0)))
        if fold:
            ins.append(g.const(shift, "bias", as_float_data=True))
        cur = g.op("Conv", ins, **attrs)
        if li == n - 1:
            break
        if li == 0:
            cur = g.op("Relu", [cur])
            gm, bt, mu, var = emb["bn"][0]
            if bn0 == "node":
                cur = g.op("BatchNormalization", [cur] + [g.const(a, "bn") for a in (gm, bt, mu, var)], epsilon=float(W.BN_EPS))
            else:                                                                             # decomposed into a per-channel affine map
                scale, shift = W.bn_scale_shift(emb["bn"][0])
                cur = g.op("Mul", [cur, g.const(scale.reshape(1, co, 1, 1), "scale")])
                cur = g.op("Add", [cur, g.const(shift.reshape(1, co, 1, 1), "shift")])
        if extra_op and li == 5:
            cur = g.op(extra_op, [cur])
        if act == "maxmul" or (act == "mixed" and li % 2):
            m = g.op("Mul", [cur, g.const(np.array(leaky, np.float32), "alpha")])
            cur = g.op("Max", [m, cur])
            cur = g.op("Max", [cur, g.const(np.array(floor, np.float32), "floor")])
        else:
            cur = g.op("LeakyRelu", [cur], alpha=float(leaky))
            cur = g.op("Clip", [cur, g.const(np.array(floor, np.float32), "lo")])             # opset >= 11: bounds are inputs
        if pool:
            pk = pool_override if (pool_override and li == 2) else list(pool)
            cur = g.op("MaxPool", [cur], kernel_shape=pk, strides=pk)
    if tail == "transpose_reshape":
        cur = g.op("Transpose", [cur], perm=[0, 2, 3, 1])
        cur = g.op("Reshape", [cur, g.const(np.array([-1, 1, 1, 96], np.int64), "shape")], out="conv2d_19")
    else:
        cur = g.op("Squeeze", [cur], out="conv2d_19", axes=[2, 3])
    g.outputs = [cur]
    g.save(path)


def write_head_torch(path, head, ln="op17", gate="where", flatten="Reshape", thr=0.5, swap=False, ln_eps=1e-5, gemm_alpha=1.0):
    g = G2("onnx::Flatten_0", opset=17)
    T, hid = head["T"], head["hidden"]
    if flatten == "Reshape":
        feats = g.op("Reshape", ["onnx::Flatten_0", g.const(np.array([-1, T * 96], np.int64), "shape")])
    else:
        feats = g.op("Flatten", ["onnx::Flatten_0"], axis=1)

    def mlp(gr, net, x):
        cur = x
        for li in (1, 2, 3):
            w, b = net[f"w{li}"], net[f"b{li}"]
            cur = gr.op("Gemm", [cur, gr.const(w.T, f"fc{li}.weight"), gr.const(b, f"fc{li}.bias")], alpha=float(gemm_alpha), beta=1.0, transB=1)
            if li == 3:
                break
            lnp = net.get(f"ln{li}")
            if lnp is not None:
                gm, bt = gr.const(lnp[0], "ln.weight"), gr.const(lnp[1], "ln.bias")
                if ln == "op17":
                    cur = gr.op("LayerNormalization", [cur, gm, bt], axis=-1, epsilon=float(ln_eps))
                else:                                                       # opset < 17 export of nn.LayerNorm
                    mean = gr.op("ReduceMean", [cur], axes=[-1], keepdims=1)
                    d = gr.op("Sub", [cur, mean])
                    var = gr.op("ReduceMean", [gr.op("Mul", [d, d])], axes=[-1], keepdims=1)
                    std = gr.op("Sqrt", [gr.op("Add", [var, gr.const(np.array(ln_eps, np.float32), "eps")])])
                    cur = gr.op("Add", [gr.op("Mul", [gr.op("Div", [d, std]), gm]), bt])
            cur = gr.op("Relu", [cur])
        return gr.op("Sigmoid", [cur]) if head["kind"] != "multiclass" else gr.op("Softmax", [gr.op("Relu", [cur])], axis=-1)

    s1 = mlp(g, head["net"], feats)
    out = s1
    if head["kind"] == "gated":
        half = g.const(np.array(thr, np.float32), "thr")
        cond = g.op("Greater", [s1, half])
        if gate == "where":
            s2 = mlp(g, head["net2"], feats)
            out = g.op("Where", [cond, s1, s2] if swap else [cond, s2, s1])
        else:                                                               # torch.jit.script-style control flow: If with subgraphs
            flag = g.op("Squeeze", [cond])
            then_g, else_g = G2("", 17), G2("", 17)
            then_g.inputs = else_g.inputs = []
            then_g.n = 1000
            then_g.outputs = [mlp(then_g, head["net2"], feats)]             # `feats` is captured from the outer scope
            else_g.outputs = [else_g.op("Identity", [s1])]
            if swap:
                then_g, else_g = else_g, then_g
            out = g.op("If", [flag], then_branch=then_g, else_branch=else_g)
    g.outputs = [out]
    g.save(path)


def _same_embedding(a, b, atol=0.0):
    assert len(a["conv"]) == len(b["conv"]) == 20 and len(a["bn"]) == len(b["bn"]) == 19
    for li, (x, y) in enumerate(zip(a["conv"], b["conv"])):
        sa, sb = (W.bn_scale_shift(a["bn"][li])[0], W.bn_scale_shift(b["bn"][li])[0]) if li < 19 else (1.0, 1.0)
        np.testing.assert_allclose(x * sa, y * sb, rtol=0, atol=atol, err_msg=f"conv {li} (x BatchNorm scale)")
    for li in range(19):
        np.testing.assert_allclose(W.bn_scale_shift(a["bn"][li])[1], W.bn_scale_shift(b["bn"][li])[1], rtol=0, atol=atol, err_msg=f"shift {li}")


@pytest.mark.parametrize("kw", [
    dict(),                                                                  # Pad nodes, Max(Mul) activation, BN node after conv0, Constant nodes
    dict(pad="attr", act="leakyclip", bn0="affine", weights="init", tail="squeeze"),
    dict(pad="auto", act="mixed", weights="hwio_transpose"),
], ids=["pad-node_maxmul", "pads-attr_leakyclip_affine-bn0", "auto-pad_mixed_hwio"])
def test_embedding_tf2onnx_idioms_load_to_the_same_network(tmp_path, kw):
    emb = W.synthetic_embedding(55)
    plain, tf = os.path.join(tmp_path, "plain.onnx"), os.path.join(tmp_path, "tf2onnx.onnx")
    write_embedding(plain, emb)
    write_embedding_tf2onnx(tf, emb, **kw)
    a, b = onnx_ingest.load_embedding(plain), onnx_ingest.load_embedding(tf)
    _same_embedding(a, b, atol=2e-7)                                         # (folding rounds w * scale to fp32 once)
    x = np.random.default_rng(2).normal(10, 1.5, (2, 76, 32, 1)).astype(np.float32)
    np.testing.assert_allclose(O.embedding_stage(x, b, np.float64), O.embedding_stage(x, emb, np.float64), rtol=0, atol=1e-5)


@pytest.mark.parametrize("kw,why", [
    (dict(leaky=0.3), "0.3"),                                                # another slope
    (dict(act="leakyclip", leaky=0.01), "LeakyRelu alpha"),
    (dict(floor=-0.5), "lower bound -0.5"),
    (dict(time_pad=1), "pad the mel axis only"),                             # 'same' padding on the time axis too
    (dict(pool_override=[2, 1]), "MaxPool kernel"),
    (dict(extra_op="Tanh"), "operator Tanh"),
    (dict(extra_op="Relu"), "Relu"),
])
def test_embedding_graphs_that_compute_something_else_are_refused_by_name(tmp_path, kw, why):
    path = os.path.join(tmp_path, "odd.onnx")
    write_embedding_tf2onnx(path, W.synthetic_embedding(55), **kw)
    with pytest.raises(ValueError, match=why):
        onnx_ingest.load_embedding(path)


@pytest.mark.parametrize("name,kw", [
    ("alexa", dict(ln="op17")), ("alexa", dict(ln="decomposed", flatten="Flatten")), ("timer", dict()),
    ("hey_jarvis", dict(gate="where")), ("hey_jarvis", dict(gate="if", ln="decomposed")),
], ids=["ln-op17", "ln-decomposed", "multiclass", "gate-where", "gate-if"])
def test_head_torch_export_idioms_load_to_the_same_head(tmp_path, name, kw):
    head = W.synthetic_head(name, 77)
    plain, th = os.path.join(tmp_path, "plain.onnx"), os.path.join(tmp_path, "torch.onnx")
    write_head(plain, head)
    write_head_torch(th, head, **kw)
    a, b = onnx_ingest.load_head(plain), onnx_ingest.load_head(th)
    assert (a["kind"], a["T"], a["hidden"], a["n_out"]) == (b["kind"], b["T"], b["hidden"], b["n_out"]) == (head["kind"], head["T"], head["hidden"], head["n_out"])
    for net in ("net", "net2"):
        if net in head:
            for k, v in head[net].items():
                for x, y, z in ((v, a[net][k], b[net][k]),) if not isinstance(v, tuple) else zip(v, a[net][k], b[net][k]):
                    if x is not None:
                        np.testing.assert_array_equal(x, y)
                        np.testing.assert_array_equal(x, z)
    feats = np.random.default_rng(1).normal(0, 2, (5, head["T"], 96)).astype(np.float32)
    np.testing.assert_array_equal(O.head_stage(feats, b, np.float32), O.head_stage(feats, head, np.float32))


@pytest.mark.parametrize("kw,why", [
    (dict(gate="where", thr=0.7), "gate threshold 0.7"),
    (dict(gate="where", swap=True), "SECOND network"),
    (dict(gate="if", swap=True), "then-branch|else-branch"),
    (dict(ln="op17", ln_eps=1e-3), "epsilon 0.001"),
    (dict(ln="decomposed", ln_eps=1e-6), "epsilon 1e-06"),
    (dict(gemm_alpha=0.5), "alpha"),
])
def test_heads_that_compute_something_else_are_refused_by_name(tmp_path, kw, why):
    path = os.path.join(tmp_path, "odd.onnx")
    write_head_torch(path, W.synthetic_head("hey_jarvis", 77), **kw)
    with pytest.raises(ValueError, match=why):
        onnx_ingest.load_head(path)


def test_two_networks_without_a_gate_are_refused(tmp_path):
    path = os.path.join(tmp_path, "nogate.onnx")
    write_head(path, W.synthetic_head("hey_jarvis", 77), tail=["Sigmoid"])            # writer 1 emits the routing only for tail=None
    with pytest.raises(ValueError, match="no comparison"):
        onnx_ingest.load_head(path)


def test_vad_reader_turns_every_parsing_failure_into_a_refusal(tmp_path):
    """a Silero-like file with If subgraphs, or an LSTM node without its optional inputs, must come back as the
    ValueError that names the host-side way out -- not as an IndexError / KeyError from the middle of the reader."""
    path = os.path.join(tmp_path, "silero_vad.onnx")
    g = G2("input", 16)
    sr_ok = g.op("Equal", ["sr", g.const(np.array(16000, np.int64), "sr16k")])
    then_g, else_g = G2("", 16), G2("", 16)
    then_g.inputs = else_g.inputs = []
    then_g.outputs = [then_g.op("LSTM", ["input"], hidden_size=64)]                    # no W / R / B inputs at all
    else_g.outputs = [else_g.op("Identity", ["input"])]
    out = g.op("If", [sr_ok], then_branch=then_g, else_branch=else_g)
    g.op("LSTM", [out], hidden_size=64)
    g.op("LSTM", [out], hidden_size=64)
    for li, (cin, cout, _s) in enumerate(W.VAD_ENC):
        g.op("Conv", [out, g.const(np.zeros((cout, cin, 3), np.float32), "w")], strides=[_s], pads=[1, 1])
    g.outputs = [out]
    g.save(path)
    with pytest.raises(ValueError, match="oww_push_vad"):
        onnx_ingest.load_vad(path)
    open(path, "wb").write(b"\x3a\xff\xff\xff\x0f" + b"\x00" * 10)                     # a length prefix that runs past the end of the file
    with pytest.raises(ValueError):
        onnx_ingest.load_vad(path)


# ---- melspectrogram.onnx: the HIP front end is analytic, so the file is VERIFIED, not loaded ---------------------------------------
def write_melspectrogram(path, win_len=400, n_fft=512, hop=160, top_db=80.0, amin=1e-10, power=2, fb=None, log_factor=None,
                         reduce_axes=None, pad=0, idiom=None):
    """torch.onnx.export of torchlibrosa's Spectrogram + LogmelFilterBank as the notebook builds them (cell 15): Unsqueeze -> two
    Conv1d (window x cos / -sin, stride hop) -> squares -> Add -> MatMul(melW) -> Clip(amin) -> Log -> Div(ln 10) -> Mul(10) ->
    ReduceMax -> Sub(top_db) -> Max."""
    g = G2("input", opset=12)
    n = np.arange(n_fft, dtype=np.float64)
    win = np.zeros(n_fft)
    lo = (n_fft - win_len) // 2
    win[lo:lo + win_len] = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_len) / win_len)
    ang = 2.0 * np.pi * np.outer(np.arange(n_fft // 2 + 1), n) / n_fft
    x = g.op("Unsqueeze", ["input"], axes=[1])
    if pad:
        x = g.op("Pad", [x, g.const(np.array([0, 0, pad, 0, 0, pad], np.int64), "pads")], mode="reflect")
    re = g.op("Conv", [x, g.const((win * np.cos(ang))[:, None, :], "conv_real.weight")], strides=[hop], dilations=[1], group=1, kernel_shape=[n_fft], pads=[0, 0])
    im = g.op("Conv", [x, g.const((-win * np.sin(ang))[:, None, :], "conv_imag.weight")], strides=[hop], dilations=[1], group=1, kernel_shape=[n_fft], pads=[0, 0])
    two = g.const(np.array(2.0, np.float32), "two")
    p = g.op("Add", [g.op("Pow", [re, two]), g.op("Pow", [im, two])])
    if power == 1:
        p = g.op("Sqrt", [p])
    if idiom == "abs":                       # a harmless operator the verifier's whitelist does not hold (|power| = power)
        p = g.op("Abs", [p])
    p = g.op("Transpose", [g.op("Unsqueeze", [p], axes=[1])], perm=[0, 1, 3, 2])
    mel = g.op("MatMul", [p, g.const(W.mel_filterbank() if fb is None else fb, "melW")])
    lg = g.op("Log", [g.op("Clip", [mel, g.const(np.array(amin, np.float32), "amin")])])
    if log_factor is None:
        db = g.op("Mul", [g.op("Div", [lg, g.const(np.array(np.log(10.0), np.float32), "ln10")]), g.const(np.array(10.0, np.float32), "ten")])
    else:
        db = g.op("Mul", [lg, g.const(np.array(log_factor, np.float32), "k")])
    mx = g.op("ReduceMax", [db], keepdims=0) if reduce_axes is None else g.op("ReduceMax", [db], axes=reduce_axes, keepdims=1)
    out = g.op("Max", [db, g.op("Sub", [mx, g.const(np.array(top_db, np.float32), "top_db")])])
    g.outputs = [out]
    g.save(path)


def test_unknown_melspectrogram_idiom_is_cleared_numerically_or_refused(tmp_path, monkeypatch):
    """ADVICE r04 + r05: a rendering of the recipe the structural verifier cannot follow must not make the default real-weights path
    unloadable -- but must not be waved through either: on GraphIdiomUnknown the file is EVALUATED on probe audio
    (onnx_ingest.probe_melspectrogram) and accepted only when it reproduces the analytic front end; a graph of unknown structure that
    computes something else (a reference offset through a second Log, another normalisation) is refused, as is any parameter FOUND to
    differ.  OWW_TRUST_MELSPECTROGRAM=1 is the only way past."""
    import warnings
    from openwakeword_amd.model import resolve_embedding
    path = os.path.join(tmp_path, "melspectrogram.onnx")
    emb = W.synthetic_embedding(5)
    write_melspectrogram(path, idiom="abs")                  # |re|^2 + |im|^2: unknown to the walker, the same numbers
    with pytest.raises(onnx_ingest.GraphIdiomUnknown, match="Abs"):
        onnx_ingest.verify_melspectrogram(path)
    assert onnx_ingest.probe_melspectrogram(path)["probe_max_abs_diff_db"] < 2e-2
    with warnings.catch_warnings():
        warnings.simplefilter("error")                        # cleared numerically: loads without a warning
        assert resolve_embedding(emb, None, melspec_model_path=path) is emb
    write_melspectrogram(path, idiom="abs", log_factor=4.0)  # unknown structure AND another scale: the probe catches it
    with pytest.raises(ValueError, match="numeric check did not clear it"):
        resolve_embedding(emb, None, melspec_model_path=path)
    monkeypatch.setenv("OWW_TRUST_MELSPECTROGRAM", "1")
    assert resolve_embedding(emb, None, melspec_model_path=path) is emb
    monkeypatch.delenv("OWW_TRUST_MELSPECTROGRAM")
    write_melspectrogram(path, hop=128)
    with pytest.raises(ValueError, match="stride") as ei:
        resolve_embedding(emb, None, melspec_model_path=path)
    assert not isinstance(ei.value, onnx_ingest.GraphIdiomUnknown)
    write_melspectrogram(path, power=1)                       # a Sqrt is a different quantity, not an idiom
    with pytest.raises(ValueError, match="Sqrt") as ei:
        onnx_ingest.verify_melspectrogram(path)
    assert not isinstance(ei.value, onnx_ingest.GraphIdiomUnknown)


def test_probe_of_a_plain_melspectrogram_file_matches_the_analytic_recipe(tmp_path):
    path = os.path.join(tmp_path, "melspectrogram.onnx")
    write_melspectrogram(path)
    res = onnx_ingest.probe_melspectrogram(path)
    assert res["n_probes"] >= 8 and res["probe_max_abs_diff_db"] < 2e-2
    write_melspectrogram(path, top_db=60.0)
    with pytest.raises(ValueError, match="different front end"):
        onnx_ingest.probe_melspectrogram(path)


def test_melspectrogram_file_is_verified_against_the_analytic_front_end(tmp_path):
    path = os.path.join(tmp_path, "melspectrogram.onnx")
    write_melspectrogram(path)
    res = onnx_ingest.verify_melspectrogram(path)
    assert res["stft_kernel_max_abs_diff"] < 1e-6 and res["filterbank_max_abs_diff"] == 0.0
    assert abs(res["log_factor"] - 10.0 / np.log(10.0)) < 1e-6 and res["top_db"] == 80.0
    write_melspectrogram(path, log_factor=10.0 / np.log(10.0))             # the scaling as one constant
    onnx_ingest.verify_melspectrogram(path)
    htk = W.mel_filterbank().copy()
    htk[40:60] *= 1.05
    for kw, why in ((dict(win_len=512), "another window"), (dict(hop=128), "stride"), (dict(top_db=100.0), "top_db = 100"),
                    (dict(amin=1e-6), "amin"), (dict(power=1), "Sqrt"), (dict(fb=htk), "filter bank differs"),
                    (dict(log_factor=1.0), "scaled by 1"), (dict(reduce_axes=[3]), "axes"), (dict(pad=256), "padded")):
        write_melspectrogram(path, **kw)
        with pytest.raises(ValueError, match=why):
            onnx_ingest.verify_melspectrogram(path)
    # ... and the recipe the file is held to IS the one the oracle (and with it the kernels) compute: evaluate the verified graph's
    # formula with numpy -- frames by its stride, its two kernels, power, its filter bank, 10 log10, the clamp -- and compare
    x = (np.random.default_rng(3).normal(0, 3000, 1760)).astype(np.float64)
    n_ = np.arange(512)
    win = np.zeros(512); win[56:456] = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(400) / 400)
    ang = 2.0 * np.pi * np.outer(np.arange(257), n_) / 512
    frames = np.stack([x[160 * t:160 * t + 512] for t in range((1760 - 512) // 160 + 1)])
    power = (frames @ (win * np.cos(ang)).T) ** 2 + (frames @ (win * np.sin(ang)).T) ** 2
    db = 10.0 * np.log10(np.maximum(power @ W.mel_filterbank().astype(np.float64), 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    np.testing.assert_allclose(O.mel_stage(x[None].astype(np.float32), np.float64)[0, 0], db, rtol=0, atol=1e-6)


from torch_export import torch_export_head as _torch_export, torch_head as _torch_head, torch_embedding as _torch_embedding, torch_melspectrogram as _torch_melspectrogram, export as _export, torch_gated as _torch_gated  # noqa: E402


# ---------------------------------------------------------------------------------------------------------------------------------
# A THIRD writer that is not ours: PyTorch's own TorchScript ONNX exporter -- the one the reference exports its wake-word models
# with (train.py:144-165: torch.onnx.export(model, rand(input_shape)[None], path, output_names=[...]), multiclass models wrapped in
# a softmax).  The `onnx` package is not installed here; the exporter only needs it for a post-pass that splices onnx-script
# functions into the finished ModelProto bytes (none occur in these models), so the test replaces that post-pass by the identity.
@pytest.mark.parametrize("name,opset,ln", [("alexa", 17, None), ("alexa", 13, None), ("alexa", 11, None), ("hey_mycroft", 14, None),
                                           ("timer", 17, None), ("timer", 12, None), ("timer", 17, True), ("weather", 13, True)])
def test_heads_written_by_pytorchs_own_exporter_load_to_the_same_weights(tmp_path, name, opset, ln):
    """opset >= 17 emits LayerNormalization, older opsets the decomposed form (ReduceMean / Sub / Pow / Sqrt / Div); Linear becomes
    Gemm with transB; Flatten, Sigmoid / Relu + Softmax as exported.  Every weight must come back bit for bit, and the oracle must
    compute with the loaded head exactly what it computes with the source head -- and what torch computed (fp32 round-off)."""
    torch = pytest.importorskip("torch")
    head = W.synthetic_head(name, 91, layernorm=ln)
    assert head["kind"] in ("binary", "multiclass")
    module = _torch_head(head["net"], head["T"], head["n_out"])
    path = os.path.join(tmp_path, f"{name}_{opset}.onnx")
    try:
        _torch_export(module, head["T"], path, opset)
    except Exception as e:                                  # noqa: BLE001 -- an exporter that cannot run here is not our failure
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    got = onnx_ingest.load_head(path)
    assert (got["kind"], got["T"], got["hidden"], got["n_out"]) == (head["kind"], head["T"], head["hidden"], head["n_out"])
    for k, v in head["net"].items():
        for x, y in ((v, got["net"][k]),) if not isinstance(v, tuple) else zip(v, got["net"][k]):
            np.testing.assert_array_equal(x, y, err_msg=k)
    feats = np.random.default_rng(2).normal(0, 2, (6, head["T"], 96)).astype(np.float32)
    want = O.head_stage(feats, head, np.float32)
    np.testing.assert_array_equal(O.head_stage(feats, got, np.float32), want)
    with torch.no_grad():
        ref = module.eval()(torch.from_numpy(feats)).numpy()
    np.testing.assert_allclose(want.reshape(ref.shape), ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("name,n_blocks,opset,hidden", [("alexa", 2, 17, None), ("alexa", 3, 13, 32), ("alexa", 0, 17, None), ("timer", 2, 12, 48),
                                                       ("hey_mycroft", 0, 11, 32)])
def test_torch_exported_heads_of_any_depth_load_to_the_same_network(tmp_path, name, n_blocks, opset, hidden):
    """train.py:67-73: Net takes any n_blocks (hidden blocks Linear -> LayerNorm -> ReLU behind the first layer; 1 in the released
    models).  The reader must return every block, in order, bit for bit -- not drop one, and not mistake a hidden layer for the output
    layer -- and the oracle must compute with the loaded head what torch computed."""
    torch = pytest.importorskip("torch")
    head = W.synthetic_head(name, 92, n_blocks=n_blocks, hidden=hidden)
    assert len(W.net_blocks(head["net"])) == n_blocks
    module = _torch_head(head["net"], head["T"], head["n_out"])
    path = os.path.join(tmp_path, f"{name}_{n_blocks}_blocks.onnx")
    try:
        _torch_export(module, head["T"], path, opset)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    got = onnx_ingest.load_head(path)
    assert (got["kind"], got["T"], got["hidden"], got["n_out"]) == (head["kind"], head["T"], head["hidden"], head["n_out"])
    for k in ("w1", "b1", "w3", "b3"):
        np.testing.assert_array_equal(got["net"][k], head["net"][k], err_msg=k)
    want_blocks, got_blocks = W.net_blocks(head["net"]), W.net_blocks(got["net"])
    assert len(got_blocks) == n_blocks
    for bi, ((w, b, ln), (w2, b2, ln2)) in enumerate(zip(want_blocks, got_blocks)):
        np.testing.assert_array_equal(w, w2, err_msg=f"block {bi} weight")
        np.testing.assert_array_equal(b, b2, err_msg=f"block {bi} bias")
        assert (ln is None) == (ln2 is None)
        if ln is not None:
            np.testing.assert_array_equal(ln[0], ln2[0]); np.testing.assert_array_equal(ln[1], ln2[1])
    feats = np.random.default_rng(3).normal(0, 2, (6, head["T"], 96)).astype(np.float32)
    want = O.head_stage(feats, head, np.float32)
    np.testing.assert_array_equal(O.head_stage(feats, got, np.float32), want)
    with torch.no_grad():
        ref = module.eval()(torch.from_numpy(feats)).numpy()
    np.testing.assert_allclose(want.reshape(ref.shape), ref, rtol=0, atol=2e-6)
    # the C ABI blob carries the depth in its header (include/owwhip.h: hdr[5] = blocks beyond the first)
    from openwakeword_amd import engine
    blob = engine.pack_head_blob(got)
    assert blob[:8].view(np.int32)[5] == n_blocks - 1
    H, O_, T = head["hidden"], head["n_out"], head["T"]
    ln_n = 2 * H if head["net"]["ln1"] is not None else 0
    assert blob.size == 8 + T * 96 * H + H + ln_n + n_blocks * (H * H + H + ln_n) + H * O_ + O_


@pytest.mark.parametrize("n_out,opset", [(1, 11), (1, 17), (4, 13)])
def test_the_recurrent_model_type_written_by_pytorchs_exporter(tmp_path, n_out, opset):
    """train.py:85-98: model_type = 'rnn' -- nn.LSTM(96, 64, num_layers=2, bidirectional=True) over the feature rows, Linear(128, n) on
    out[:, -1], Sigmoid | ReLU (+ the softmax wrapper of train.py:152-165).  The file PyTorch's exporter writes for that module loads
    to the weights it was built from (ONNX gate order i o f c -> torch's i f g o, b_ih + b_hh summed), the oracle on the loaded head
    equals the torch module, and the blob the C ABI takes has the documented size."""
    torch = pytest.importorskip("torch")
    import torch_export as TE
    from openwakeword_amd import engine
    head = W.synthetic_head("anrnn", 9, kind="rnn", n_out=n_out, T=16)
    module = TE.torch_rnn_head(head)
    path = os.path.join(tmp_path, "rnn.onnx")
    try:
        TE.torch_export_head(module, 16, path, opset)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    got = onnx_ingest.load_head(path)
    assert (got["kind"], got["T"], got["hidden"], got["n_out"]) == ("rnn", 16, 64, n_out)
    for li in range(2):
        for d in range(2):
            np.testing.assert_array_equal(got["lstm"][li][d][0], head["lstm"][li][d][0])
            np.testing.assert_allclose(got["lstm"][li][d][1], head["lstm"][li][d][1], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(got["w_out"], head["w_out"])
    x = np.random.default_rng(1).normal(0, 3.0, (5, 16, 96)).astype(np.float32)
    with torch.no_grad():
        want = module(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(O.head_stage(x, got, np.float32), want, rtol=0, atol=2e-6)
    assert engine.pack_head_blob(got).size == 8 + 2 * (160 * 256 + 256) + 2 * (192 * 256 + 256) + 128 * n_out + n_out


def test_recurrent_heads_other_than_train_pys_are_refused_by_name(tmp_path):
    """What heads_rnn_kernel fixes is checked in the file: both directions, 64 hidden units, two layers, the LAST time step."""
    torch = pytest.importorskip("torch")

    class Rnn(torch.nn.Module):
        def __init__(self, hidden=64, layers=2, bi=True, step=-1):
            super().__init__()
            self.step = step
            self.layer1 = torch.nn.LSTM(96, hidden, num_layers=layers, bidirectional=bi, batch_first=True, dropout=0.0)
            self.layer2 = torch.nn.Linear(hidden * (2 if bi else 1), 1)
            self.layer3 = torch.nn.Sigmoid()

        def forward(self, x):
            out, _h = self.layer1(x)
            return self.layer3(self.layer2(out[:, self.step]))

    path = os.path.join(tmp_path, "rnn.onnx")
    for kw, why in ((dict(bi=False), "direction"), (dict(hidden=32), "hidden_size"), (dict(layers=1), "LSTM nodes"), (dict(step=0), "not the last one")):
        try:
            _torch_export(Rnn(**kw).eval(), 16, path, 13)
        except Exception as e:                                  # noqa: BLE001
            pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
        with pytest.raises(ValueError, match=why):
            onnx_ingest.load_head(path)

    class Gru(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layer1 = torch.nn.GRU(96, 64, batch_first=True)
            self.layer2 = torch.nn.Linear(64, 1)

        def forward(self, x):
            return torch.sigmoid(self.layer2(self.layer1(x)[0][:, -1]))

    _torch_export(Gru().eval(), 16, path, 13)
    with pytest.raises(ValueError, match="recurrent network of GRU"):
        onnx_ingest.load_head(path)


def test_a_head_deeper_than_the_kernels_take_is_refused_by_name(tmp_path):
    pytest.importorskip("torch")
    head = W.synthetic_head("alexa", 92, hidden=16)
    path = os.path.join(tmp_path, "ten_blocks.onnx")
    try:
        _torch_export(_torch_head(head["net"], head["T"], 1, n_blocks=onnx_ingest.MAX_HEAD_BLOCKS + 1), head["T"], path, 17)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    with pytest.raises(ValueError, match="hidden blocks"):
        onnx_ingest.load_head(path)


@pytest.mark.parametrize("opset,act", [(17, "leakyclamp"), (13, "leakyclamp"), (13, "maxmul")])
def test_embedding_written_by_pytorchs_own_exporter_loads_to_the_same_network(tmp_path, opset, act):
    """Not the exporter the real file came from (that was tf2onnx), but a writer that is not ours: Conv with the eval-mode BatchNorm
    folded in (scaled weights + bias), a standalone BatchNormalization behind conv0's Relu, LeakyRelu + Clip or Mul / Max chains,
    MaxPool, NCHW with one Transpose either side."""
    torch = pytest.importorskip("torch")
    emb = W.synthetic_embedding(56)
    module = _torch_embedding(emb, act)
    path = os.path.join(tmp_path, "embedding_torch.onnx")
    try:
        _export(module, torch.rand(1, 76, 32, 1), path, opset, input_names=["input_1"])
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    got = onnx_ingest.load_embedding(path)
    plain = os.path.join(tmp_path, "plain.onnx")
    write_embedding(plain, emb)
    _same_embedding(onnx_ingest.load_embedding(plain), got, atol=5e-7)
    x = np.random.default_rng(2).normal(10, 1.5, (2, 76, 32, 1)).astype(np.float32)
    want = O.embedding_stage(x, emb, np.float64)
    np.testing.assert_allclose(O.embedding_stage(x, got, np.float64), want, rtol=0, atol=2e-5)
    with torch.no_grad():
        ref = module(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(ref, want, rtol=0, atol=5e-4 * max(1.0, float(np.abs(want).max())))


def test_melspectrogram_written_by_pytorchs_own_exporter_is_verified(tmp_path):
    """The real melspectrogram.onnx came out of torch.onnx.export (notebook cell 15); so does this one.  The verifier must accept the
    exporter's rendering of the recipe and still refuse another hop or top_db written the same way."""
    torch = pytest.importorskip("torch")

    def export(module, path):
        _export(module, torch.rand(1, 1760) * 1000, path, 12, input_names=["input"], dynamic_axes={"input": {0: "batch", 1: "samples"}})

    path = os.path.join(tmp_path, "melspectrogram_torch.onnx")
    try:
        export(_torch_melspectrogram(), path)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    res = onnx_ingest.verify_melspectrogram(path)
    assert res["stft_kernel_max_abs_diff"] < 1e-6 and res["filterbank_max_abs_diff"] == 0.0 and res["top_db"] == 80.0
    assert onnx_ingest.probe_melspectrogram(path)["probe_max_abs_diff_db"] < 2e-2       # ... and the numeric probe agrees on the exporter's rendering
    # the module itself computes what the oracle computes
    x = (np.random.default_rng(4).normal(0, 3000, (2, 1760))).astype(np.float32)
    with torch.no_grad():
        ref = _torch_melspectrogram()(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(ref, O.mel_stage(x, np.float32), rtol=0, atol=2e-3)
    for kw, why in ((dict(top_db=100.0), "top_db"), (dict(hop=128), "stride")):
        export(_torch_melspectrogram(**kw), path)
        with pytest.raises(ValueError, match=why):
            onnx_ingest.verify_melspectrogram(path)
        with pytest.raises(ValueError, match="different front end|rows"):
            onnx_ingest.probe_melspectrogram(path)


@pytest.mark.parametrize("form,opset", [("where", 13), ("where", 17), ("if", 13)])
def test_gated_head_written_by_pytorchs_own_exporter(tmp_path, form, opset):
    """hey_jarvis-style routing as the real exporter renders it (Greater + Where; a scripted branch as an If with one subgraph per
    arm): kind 'gated', both networks bit for bit, and the oracle on the loaded head == the torch module on either side of 0.5."""
    torch = pytest.importorskip("torch")
    head = W.synthetic_head("hey_jarvis", 93)
    assert head["kind"] == "gated"
    module = _torch_gated(head, form)
    path = os.path.join(tmp_path, f"gated_{form}.onnx")
    try:
        _torch_export(module, head["T"], path, opset)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    got = onnx_ingest.load_head(path)
    assert got["kind"] == "gated" and (got["T"], got["hidden"], got["n_out"]) == (head["T"], head["hidden"], 1)
    for net in ("net", "net2"):
        for k, v in head[net].items():
            for x, y in ((v, got[net][k]),) if not isinstance(v, tuple) else zip(v, got[net][k]):
                np.testing.assert_array_equal(x, y, err_msg=f"{net}.{k}")
    rng = np.random.default_rng(3)
    feats = rng.normal(0, 2, (40, head["T"], 96)).astype(np.float32)
    want = O.head_stage(feats, got, np.float32).reshape(-1)
    with torch.no_grad():
        ref = np.array([float(module(torch.from_numpy(f[None]))[0, 0]) for f in feats])
    first = O.head_stage(feats, {"kind": "binary", "T": head["T"], "hidden": head["hidden"], "n_out": 1, "net": head["net"]}, np.float32).reshape(-1)
    assert (first > 0.5).any() and (first <= 0.5).any()     # both arms exercised
    np.testing.assert_allclose(want, ref, rtol=0, atol=2e-6)


def test_vad_standin_written_by_pytorchs_own_exporter_loads_to_the_same_weights(tmp_path):
    """nn.LSTM(num_layers=2) becomes two ONNX LSTM nodes (gate order i o f c, W / R / B split, initial states sliced from h, c), nn.Linear
    on a 3-D input MatMul + Add, the |STFT| two strided Conv1d: `load_vad` must return the source weights (LSTM biases as b_ih + b_hh)."""
    pytest.importorskip("torch")
    import torch_export as TE
    vad = W.synthetic_vad(35)
    path = os.path.join(tmp_path, "silero_vad.onnx")
    try:
        TE.export_vad(vad, path)
    except Exception as e:                                  # noqa: BLE001
        pytest.skip(f"torch.onnx.export is not usable in this environment: {type(e).__name__}: {e}")
    got = onnx_ingest.load_vad(path)
    for (w, b), (w2, b2) in zip(vad["enc"], got["enc"]):
        np.testing.assert_array_equal(w, w2)
        np.testing.assert_array_equal(b, b2)
    for (w, b), (w2, b2) in zip(vad["lstm"], got["lstm"]):
        np.testing.assert_array_equal(w, w2)
        np.testing.assert_array_equal(b, b2)
    np.testing.assert_array_equal(vad["dec"][0], got["dec"][0])
    assert float(got["dec"][1]) == float(vad["dec"][1])
