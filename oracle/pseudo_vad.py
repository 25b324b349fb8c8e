"""
TEST INFRASTRUCTURE ONLY -- a deterministic stand-in for the Silero VAD network.

`silero_vad.onnx` is a release asset that is not in the reference checkout and its graph is not described anywhere in the
reference, so the voice-activity NETWORK cannot be restated.  What can be pinned is everything
around it: the 640-sample sub-framing and /32767 scaling, the carried (h, c) state, the mean over sub-frames, the 125-deep
score ring, the gate on frames 0.4-0.56 s back and the fact that Model.reset() leaves the VAD alone
(/root/reference/openwakeword/vad.py:92-130, model.py:208-210, 366-381).  This session object has the interface the
reference drives (`run(None, {'input', 'h', 'c', 'sr'}) -> [out, h, c]`, vad.py:121-124) and a cheap energy-driven recurrence
inside; tests/golden/make_golden_vad.py installs it behind the reference's own VAD class, the tests hand the same object to
openwakeword_amd.VAD.
"""
import numpy as np


class PseudoVadSession:
    def run(self, output_names, feeds):
        x = np.asarray(feeds["input"], dtype=np.float64)
        h = np.asarray(feeds["h"], dtype=np.float32)
        c = np.asarray(feeds["c"], dtype=np.float32)
        assert x.ndim == 2 and h.shape == (2, x.shape[0], 64) and c.shape == h.shape and int(feeds["sr"]) == 16000
        e = np.log10(np.mean(x * x, axis=1) + 1e-10)                     # -10 (silence) .. 0 (full scale)
        drive = (1.2 * (e + 4.0)).astype(np.float32)
        hn = (0.6 * h + 0.4 * np.tanh(drive)[None, :, None]).astype(np.float32)
        cn = (c + 1.0).astype(np.float32)
        z = 2.0 * drive + 1.5 * hn[0, :, 0] - 0.01 * np.minimum(cn[0, :, 0], 50.0)
        out = (1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(np.float32)[:, None]
        return [out, hn, cn]
