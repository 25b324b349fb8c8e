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


def write_head(path, head, layernorm_op=True, use_matmul=False, raw=True, tail=None, hidden_act="Relu"):
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
                        nodes.append(_node("LayerNormalization", [cur, f"n{k}g{li}", f"n{k}be{li}"], [f"n{k}n{li}"], [_attr_f("epsilon", 1e-5)]))
                    else:
                        nodes += [_node("Mul", [cur, f"n{k}g{li}"], [f"n{k}q{li}"]), _node("Add", [f"n{k}q{li}", f"n{k}be{li}"], [f"n{k}n{li}"])]
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
    open(path, "wb").write(_model(nodes, inits))


def write_embedding(path, emb, fold_last_bns=0):
    nodes, inits, cur = [], [], "x"
    n = len(emb["conv"])
    for li, w in enumerate(emb["conv"]):
        inits.append(_tensor(f"w{li}", np.transpose(w, (3, 2, 0, 1))))                       # HWIO -> OIHW
        folded = li < n - 1 and li >= n - 1 - fold_last_bns
        if folded:
            scale, shift = W.bn_scale_shift(emb["bn"][li])
            inits[-1] = _tensor(f"w{li}", np.transpose(w * scale, (3, 2, 0, 1)))
            inits.append(_tensor(f"b{li}", shift))
            nodes.append(_node("Conv", [cur, f"w{li}", f"b{li}"], [f"c{li}"]))
        else:
            nodes.append(_node("Conv", [cur, f"w{li}"], [f"c{li}"]))
        cur = f"c{li}"
        if li == 0:
            nodes.append(_node("Relu", [cur], ["r0"]))
            cur = "r0"
        if li < n - 1 and not folded:
            for nm, arr in zip("gbmv", emb["bn"][li]):
                inits.append(_tensor(f"{nm}{li}", arr))
            nodes.append(_node("BatchNormalization", [cur, f"g{li}", f"b{li}", f"m{li}", f"v{li}"], [f"n{li}"], [_attr_f("epsilon", W.BN_EPS)]))
            cur = f"n{li}"
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
def _attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_vi((8 << 3) | 0) + _vi(v) for v in vals) + _vi((20 << 3) | 0) + _vi(7)


def _attr_s(name, v):
    return _ld(1, name.encode()) + _ld(4, v.encode()) + _vi((20 << 3) | 0) + _vi(3)


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
