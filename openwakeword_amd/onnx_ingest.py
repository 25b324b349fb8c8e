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
* melspectrogram: analytic in this package; `check_melspectrogram` compares the file's filterbank with ours.

No real model file exists in this environment, so the structural assumptions are pinned only by the
round-trip test in tests/test_onnx_ingest.py (files written by a minimal protobuf writer); anything unrecognised raises
ValueError instead of guessing.
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
        elif fno == 7:
            val = (val or []) + (list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]])
        elif fno == 8:
            val = (val or []) + [x if x < (1 << 63) else x - (1 << 64) for x in _packed_ints(v, wt)]
    return name, val


def load_graph(path: str) -> dict:
    """{'initializers': {name: ndarray}, 'nodes': [{'op', 'name', 'inputs', 'outputs', 'attrs'}]} of an ONNX file."""
    data = open(path, "rb").read()
    graph = None
    for fno, wt, v in _fields(data):
        if fno == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError(f"{path}: no GraphProto found (not an ONNX ModelProto?)")
    inits: Dict[str, np.ndarray] = {}
    nodes = []
    for fno, wt, v in _fields(graph):
        if fno == 5:
            name, arr = _tensor(v)
            if arr is not None:
                inits[name] = arr
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
            if node["op"] == "Constant" and "value" in node["attrs"] and node["outputs"]:
                inits[node["outputs"][0]] = node["attrs"]["value"]
            nodes.append(node)
    return {"initializers": inits, "nodes": nodes}


# ------------------------------------------------------------------------------------------- wake-word heads
def _linear_layers(g: dict):
    """(weight [in, out], bias [out]) of every Gemm / MatMul(+Add) in graph order."""
    inits, out = g["initializers"], []
    nodes = g["nodes"]
    for i, n in enumerate(nodes):
        if n["op"] == "Gemm":
            w = inits.get(n["inputs"][1])
            if w is None or w.ndim != 2:
                continue
            w = w.T if n["attrs"].get("transB", 0) else w
            b = inits.get(n["inputs"][2]) if len(n["inputs"]) > 2 else None
            out.append((np.ascontiguousarray(w, np.float32), (np.zeros(w.shape[1], np.float32) if b is None else b.astype(np.float32)), i))
        elif n["op"] == "MatMul":
            w = inits.get(n["inputs"][1])
            if w is None or w.ndim != 2:
                continue
            b = None
            for m in nodes[i + 1:i + 3]:
                if m["op"] == "Add" and n["outputs"][0] in m["inputs"]:
                    other = [x for x in m["inputs"] if x != n["outputs"][0]]
                    b = inits.get(other[0]) if other else None
            out.append((np.ascontiguousarray(w, np.float32), (np.zeros(w.shape[1], np.float32) if b is None else b.astype(np.float32)), i))
    return out


def _layernorms(g: dict, hidden: int):
    """(gamma, beta) pairs in graph order: LayerNormalization nodes, or Mul / Add by [hidden] initializers after a Sqrt/Div."""
    inits, out = g["initializers"], []
    for n in g["nodes"]:
        if n["op"] == "LayerNormalization":
            gm = inits.get(n["inputs"][1])
            bt = inits.get(n["inputs"][2]) if len(n["inputs"]) > 2 else None
            if gm is not None:
                out.append((gm.astype(np.float32), np.zeros_like(gm, np.float32) if bt is None else bt.astype(np.float32)))
    if out:
        return out
    # decomposed form: ... -> Mul(x, gamma[hidden]) -> Add(., beta[hidden])
    nodes = g["nodes"]
    for i, n in enumerate(nodes):
        if n["op"] != "Mul":
            continue
        gm = [inits[x] for x in n["inputs"] if x in inits and inits[x].shape == (hidden,)]
        if not gm:
            continue
        bt = None
        for m in nodes[i + 1:i + 3]:
            if m["op"] == "Add" and n["outputs"][0] in m["inputs"]:
                cand = [inits[x] for x in m["inputs"] if x in inits and inits[x].shape == (hidden,)]
                bt = cand[0] if cand else None
        out.append((gm[0].astype(np.float32), np.zeros(hidden, np.float32) if bt is None else bt.astype(np.float32)))
    return out


def load_head(path: str) -> dict:
    g = load_graph(path)
    lin = _linear_layers(g)
    if len(lin) not in (3, 6):
        raise ValueError(f"{path}: expected 3 (or 6 for a gated model) linear layers, found {len(lin)}")
    n_nets = len(lin) // 3
    w1 = lin[0][0]
    if w1.shape[0] % W.EMB_DIM:
        raise ValueError(f"{path}: first layer input {w1.shape[0]} is not a multiple of {W.EMB_DIM}")
    T, hidden, n_out = w1.shape[0] // W.EMB_DIM, w1.shape[1], lin[2][0].shape[1]
    lns = _layernorms(g, hidden)
    if len(lns) not in (0, 2 * n_nets):
        raise ValueError(f"{path}: found {len(lns)} LayerNorm parameter pairs for {n_nets} network(s)")
    nets = []
    for k in range(n_nets):
        (a, ab, _), (b, bb, _), (c, cb, _) = lin[3 * k:3 * k + 3]
        if a.shape != (T * W.EMB_DIM, hidden) or b.shape != (hidden, hidden) or c.shape != (hidden, n_out):
            raise ValueError(f"{path}: network {k} has layer shapes {a.shape}, {b.shape}, {c.shape}")
        nets.append({"w1": a, "b1": ab, "ln1": lns[2 * k] if lns else None, "w2": b, "b2": bb,
                     "ln2": lns[2 * k + 1] if lns else None, "w3": c, "b3": cb})
    # the activations the kernels will apply are fixed (Linear -> [LN] -> ReLU twice, then Sigmoid, or ReLU + Softmax for a
    # multiclass model): check that the file really has them instead of inferring the kind from one op name
    acts = {"Relu", "Sigmoid", "Softmax", "Tanh", "LeakyRelu", "Elu", "Selu", "Gelu", "HardSigmoid", "PRelu", "Clip"}
    tails = []
    for k in range(n_nets):
        idx = [lin[3 * k + i][2] for i in range(3)]
        stop = lin[3 * k + 3][2] if k + 1 < n_nets else len(g["nodes"])
        for a, b in ((idx[0], idx[1]), (idx[1], idx[2])):
            between = [n["op"] for n in g["nodes"][a + 1:b] if n["op"] in acts]
            if between != ["Relu"]:
                raise ValueError(f"{path}: expected exactly one Relu between the linear layers of network {k}, found {between}")
        tails.append([n["op"] for n in g["nodes"][idx[2] + 1:stop] if n["op"] in acts])
    if any(t != tails[0] for t in tails):
        raise ValueError(f"{path}: the networks of a gated model end differently: {tails}")
    if tails[0] == ["Sigmoid"]:
        kind = "gated" if n_nets == 2 else "binary"
    elif tails[0] == ["Relu", "Softmax"] and n_nets == 1:
        kind = "multiclass"
    else:
        # e.g. train.py's multiclass branch ends in a bare ReLU (train.py:81-83): no kernel applies that, so refuse
        raise ValueError(f"{path}: unsupported output activation {tails[0]} (supported: Sigmoid, or Relu -> Softmax)")
    head = {"kind": kind, "T": int(T), "hidden": int(hidden), "n_out": int(n_out), "net": nets[0]}
    if n_nets == 2:
        head["net2"] = nets[1]
    return head


# ------------------------------------------------------------------------------------------- embedding CNN
def load_embedding(path: str) -> dict:
    g = load_graph(path)
    inits = g["initializers"]
    convs, bns = [], {}
    for i, n in enumerate(g["nodes"]):
        if n["op"] == "Conv":
            w = inits.get(n["inputs"][1])
            if w is None or w.ndim != 4:
                raise ValueError(f"{path}: Conv node {n['name']} has no 4-D weight initializer")
            bias = inits.get(n["inputs"][2]) if len(n["inputs"]) > 2 else None
            convs.append((np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)), np.float32), bias))        # OIHW -> HWIO
        elif n["op"] == "BatchNormalization":
            p = [inits.get(x) for x in n["inputs"][1:5]]
            if any(x is None for x in p):
                raise ValueError(f"{path}: BatchNormalization node {n['name']} without constant parameters")
            eps = float(n["attrs"].get("epsilon", 1e-5))
            gm, bt, mu, var = (x.astype(np.float64) for x in p)
            # re-express with this package's epsilon so that weights.bn_scale_shift reproduces the file's arithmetic
            if abs(eps - W.BN_EPS) > 1e-12:
                var = var + (eps - W.BN_EPS)
            bns[len(convs) - 1] = (gm.astype(np.float32), bt.astype(np.float32), mu.astype(np.float32), var.astype(np.float32))
    if len(convs) != len(W.CNN_TOPOLOGY):
        raise ValueError(f"{path}: expected {len(W.CNN_TOPOLOGY)} Conv nodes, found {len(convs)}")
    conv, bn = [], []
    for li, ((w, bias), (kh, kw, ci, co, _)) in enumerate(zip(convs, W.CNN_TOPOLOGY)):
        if w.shape != (kh, kw, ci, co):
            raise ValueError(f"{path}: Conv {li} has HWIO shape {w.shape}, expected {(kh, kw, ci, co)}")
        conv.append(w)
        if li == len(convs) - 1:
            if bias is not None and np.abs(bias).max() > 0:
                raise ValueError(f"{path}: the last convolution carries a bias; the kernel has no slot for it")
            continue
        if li in bns:
            if bias is not None and np.abs(bias).max() > 0:
                gm, bt, mu, var = bns[li]
                bns[li] = (gm, bt, mu - bias.astype(np.float32), var)          # BN(conv + b) == BN'(conv)
            bn.append(bns[li])
        elif bias is not None:
            one = np.ones(co, np.float32)                                       # folded BatchNorm: y = conv + bias
            bn.append((one, bias.astype(np.float32), np.zeros(co, np.float32), one - np.float32(W.BN_EPS)))
        else:
            raise ValueError(f"{path}: Conv {li} has neither a BatchNormalization nor a bias")
    return {"conv": conv, "bn": bn}


def check_melspectrogram(path: str) -> dict:
    """Compare the file's mel filterbank ([257, 32] MatMul operand) and DFT kernels with this package's analytic tables."""
    g = load_graph(path)
    fb = [a for a in g["initializers"].values() if a.shape in ((W.N_BINS, W.N_MELS), (W.N_MELS, W.N_BINS))]
    if not fb:
        raise ValueError(f"{path}: no {W.N_BINS}x{W.N_MELS} filterbank initializer found")
    m = fb[0] if fb[0].shape == (W.N_BINS, W.N_MELS) else fb[0].T
    return {"filterbank_max_abs_diff": float(np.abs(m - W.mel_filterbank()).max()), "filterbank_max": float(np.abs(m).max())}


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
            bd = float(np.asarray(inits[n["inputs"][2]]).reshape(-1)[0]) if len(n["inputs"]) > 2 and n["inputs"][2] in inits else 0.0
            dec = (wd, np.float32(bd))
    if dec is None or "Sigmoid" not in ops:
        refuse("no 64 -> 1 decoder followed by a Sigmoid")
    return {"enc": out_enc, "lstm": out_lstm, "dec": dec}
