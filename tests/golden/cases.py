"""Case list shared by make_golden.py (runs the REFERENCE) and the tests (run the oracle / HIP path)."""

SEED_WEIGHTS = 1234
SEED_NP = 7            # np.random.seed() before every Model construction (feature-ring init noise)
HEADS_BINARY = ["alexa", "hey_mycroft", "hey_jarvis"]
CLIPS = ["alexa_test", "hey_mycroft_test", "hey_jane"]

# (case id, heads, clip, predict_clip kwargs)
CLIP_CASES = [
    ("c1280", HEADS_BINARY, "alexa_test", dict(chunk_size=1280)),
    ("c1280m", HEADS_BINARY, "hey_mycroft_test", dict(chunk_size=1280)),
    ("c1280j", HEADS_BINARY, "hey_jane", dict(chunk_size=1280)),
    ("c2560", HEADS_BINARY, "alexa_test", dict(chunk_size=2560)),
    ("c1024", HEADS_BINARY, "alexa_test", dict(chunk_size=1024)),
    ("c2048", HEADS_BINARY, "alexa_test", dict(chunk_size=2048)),
    ("c400", HEADS_BINARY, "hey_mycroft_test", dict(chunk_size=400)),
    ("c3000", HEADS_BINARY, "hey_mycroft_test", dict(chunk_size=3000)),
    ("c5120np", HEADS_BINARY, "hey_jane", dict(chunk_size=5120, padding=0)),
    ("patience", HEADS_BINARY, "hey_jane",
     dict(chunk_size=1280, patience={"alexa": 2, "hey_jarvis": 3}, threshold={"alexa": 0.4, "hey_jarvis": 0.45})),
    ("debounce", HEADS_BINARY, "hey_jane",
     dict(chunk_size=1280, debounce_time=0.4, threshold={"alexa": 0.4, "hey_mycroft": 0.5, "hey_jarvis": 0.45})),
    ("timer", ["timer"], "hey_mycroft_test", dict(chunk_size=1280)),
    ("timer2560", ["timer"], "hey_mycroft_test", dict(chunk_size=2560)),
]

# VAD-gated cases (row I): (case id, heads, clip, predict_clip kwargs, vad_threshold).  The network behind the reference's VAD
# class is oracle/pseudo_vad.py (the real silero_vad.onnx is not available); chunk sizes are multiples of the 640-sample VAD frame.
VAD_CASES = [
    ("vad03", ["alexa", "hey_mycroft"], "alexa_test", dict(chunk_size=1280), 0.3),
    ("vad07", ["alexa", "hey_mycroft"], "hey_jane", dict(chunk_size=1280), 0.7),
    ("vad2560", ["alexa"], "hey_mycroft_test", dict(chunk_size=2560), 0.5),
    ("vad640", ["alexa"], "alexa_test", dict(chunk_size=640), 0.5),
    ("vadnp", ["hey_mycroft"], "hey_mycroft_test", dict(chunk_size=1280, padding=0), 0.5),
]

# The reference on real model FILES (tests/golden/make_golden_onnx.py): heads written by PyTorch's exporter under these names and
# opsets (13 and older: decomposed LayerNorm; 17: the fused operator), loaded BY PATH; multiclass = the catalogue's timer shape.
ONNX_HEADS = ["alexa_custom", "mycroft_custom", "timer_custom", "jarvis_custom", "jarvis_custom_if", "deep2_custom", "flat0_custom",
              "rnn_custom", "rnn3_custom"]
ONNX_HEAD_OPSETS = {"alexa_custom": 13, "mycroft_custom": 17, "timer_custom": 12, "jarvis_custom": 13, "jarvis_custom_if": 13,
                    "deep2_custom": 17, "flat0_custom": 13, "rnn_custom": 13, "rnn3_custom": 17}
ONNX_FILE_CASES = [
    ("f1280", ["alexa_custom", "mycroft_custom"], "alexa_test", dict(chunk_size=1280)),
    ("f1280j", ["alexa_custom", "mycroft_custom"], "hey_jane", dict(chunk_size=1280)),
    ("f2560", ["alexa_custom", "mycroft_custom"], "hey_mycroft_test", dict(chunk_size=2560)),
    ("f1024", ["alexa_custom"], "alexa_test", dict(chunk_size=1024)),
    ("ftimer", ["timer_custom"], "hey_mycroft_test", dict(chunk_size=1280)),
    ("fpat", ["alexa_custom", "mycroft_custom"], "hey_jane",
     dict(chunk_size=1280, patience={"alexa_custom": 2}, threshold={"alexa_custom": 0.4})),
    # hey_jarvis-style verifier routing (docs/models/hey_jarvis.md:38) as torch.where and as a scripted branch (an ONNX If)
    ("fjarvis", ["jarvis_custom", "alexa_custom"], "hey_jane", dict(chunk_size=1280)),
    ("fjarvisif", ["jarvis_custom_if"], "hey_jane", dict(chunk_size=1280)),
    # train.py:67-73: Net(n_blocks) -- two hidden blocks of 32 units (the training pipeline's width) and none at all
    ("fdeep", ["deep2_custom", "flat0_custom", "alexa_custom"], "hey_jane", dict(chunk_size=1280)),
    # train.py:85-98: model_type "rnn" -- 2-layer bidirectional LSTM(64) + Linear on the last step; one class (Sigmoid) and three (softmax wrapper)
    ("frnn", ["rnn_custom", "alexa_custom"], "hey_jane", dict(chunk_size=1280)),
    ("frnn3", ["rnn3_custom"], "alexa_test", dict(chunk_size=2560)),
]


def onnx_file_weights():
    """{"embedding", "heads"}: the synthetic weights behind the exported files (seed SEED_WEIGHTS)."""
    from openwakeword_amd import weights as W
    base = {"alexa_custom": "alexa", "mycroft_custom": "hey_mycroft", "timer_custom": "timer", "jarvis_custom": "hey_jarvis",
            "jarvis_custom_if": "hey_jarvis"}
    heads = {n: W.synthetic_head(b, SEED_WEIGHTS) for n, b in base.items()}
    heads["deep2_custom"] = W.synthetic_head("deep2", SEED_WEIGHTS, hidden=32, n_blocks=2)
    heads["flat0_custom"] = W.synthetic_head("flat0", SEED_WEIGHTS, n_blocks=0)
    heads["rnn_custom"] = W.synthetic_head("rnn", SEED_WEIGHTS, kind="rnn", n_out=1)
    heads["rnn3_custom"] = W.synthetic_head("rnn3", SEED_WEIGHTS, kind="rnn", n_out=3)
    return {"embedding": W.synthetic_embedding(SEED_WEIGHTS), "heads": heads}

# direct predict() calls of ragged sizes, empty calls included (model.py:232-386 on whatever the caller hands over)
ONNX_SEQUENCE = ("fseq", ["alexa_custom", "timer_custom"], "hey_jane",
                 [0, 1280, 0, 640, 0, 640, 400, 3000, 17, 1263, 2560, 5000, 1, 0, 1280, 1279, 1, 1281, 2559, 1280, 1280])

# the reference's VAD class (vad.py:54-130) on a voice-activity FILE: the stand-in network written by PyTorch's exporter (two ONNX
# LSTM nodes, state carried through the h / c inputs); (case id, heads, clip, predict_clip kwargs, vad_threshold)
ONNX_VAD_SEED = 77
ONNX_VAD_CASES = [
    ("fvad20", ["alexa_custom"], "hey_jane", dict(chunk_size=1280), 0.2),          # (this network's scores span 0.07 ... 0.36)
    ("fvad30", ["alexa_custom", "mycroft_custom"], "alexa_test", dict(chunk_size=2560), 0.3),
    ("fvad12", ["mycroft_custom"], "hey_mycroft_test", dict(chunk_size=1280, padding=0), 0.12),
]

# Model(custom_verifier_models={name: pickle}) (model.py:183-195, 320-328): the pickled scikit-learn pipeline of tests/verifier_fixture.py
ONNX_VERIFIER = ("fver", ["alexa_custom", "mycroft_custom"], "hey_jane", dict(chunk_size=1280), "alexa_custom", 0.3)

# class_mapping_dicts as the reference's own test passes them (tests/test_models.py:139-149; model.py:176-177 stores the OUTER dict, so
# the argument only ever matters for single-output models, whose mapping predict() never reads), parent lookup (215-224) and
# false-positive mining from a WAV file (428-479) on exported files
ONNX_MAPPING = ("fmap", ["timer_custom", "alexa_custom"], "hey_jane", [{}, {"alexa_custom": {"0": "positive"}}], 0.25)

# Model(enable_speex_noise_suppression=True) (model.py:201-205, 272-273, 481-504; the reference's own test: tests/test_models.py:179-215)
# with oracle/fake_speex.py standing in for the absent `speexdsp_ns` package: cleaned audio to the preprocessor, the RAW call argument
# to the voice-activity detector (model.py:370), predict_clip's chunking on top; (case id, heads, clip, predict_clip kwargs, vad_threshold)
ONNX_SPEEX_CASES = [
    ("fspeex", ["alexa_custom", "mycroft_custom"], "alexa_test", dict(chunk_size=1280), 0.0),
    ("fspeex2560", ["alexa_custom", "mycroft_custom"], "hey_jane", dict(chunk_size=2560), 0.0),
    ("fspeexvad", ["alexa_custom"], "hey_jane", dict(chunk_size=1280), 0.2),
]

# predict() calls LONGER than a handle's mel buffer (max_chunks): the reference runs melspectrogram.onnx once over the whole call
# (utils.py:387-401), so its clamp floor "maximum - 80 dB" is the call's -- a call whose loudest frame sits in its LAST chunk clamps
# the near-silent chunks before it.  (case id, heads, call sizes); the audio is long_call_pcm(alexa clip)
ONNX_LONG = ("flong", ["alexa_custom", "mycroft_custom"], [1280] * 6 + [6400, 1280, 6400, 2560, 1280])


def long_call_pcm(alexa_clip):
    """int16 audio for ONNX_LONG: six loud warm-up chunks; a 5-chunk call = 4 chunks of +-3 LSB noise, then the loudest 80 ms of the
    clip; a loud chunk; a 5-chunk call the other way round (loud first); a quiet 2-chunk call; a loud chunk."""
    import numpy as np
    r = np.random.default_rng(4242)
    c = np.asarray(alexa_clip, np.int16)
    c = np.resize(c, max(len(c), 1280 * 12))                              # (a short clip is tiled)
    i = int(np.argmax(np.abs(c.astype(np.int32))))
    lo = min(max(0, i - 640 - 1280 * 5), len(c) - 1280 * 10)
    loud = c[lo: lo + 1280 * 10].reshape(10, 1280)
    peak = loud[int(np.argmax(np.abs(loud.astype(np.int32)).max(axis=1)))]
    quiet = lambda n: r.integers(-3, 4, 1280 * n).astype(np.int16)      # noqa: E731
    parts = [loud[:6].reshape(-1), quiet(4), peak, loud[6], peak, quiet(4), quiet(2), loud[7]]
    return np.concatenate(parts)
