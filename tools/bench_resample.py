#!/usr/bin/env python3
"""Device time of oww_resample for a whole batch (device pointers in and out): python tools/bench_resample.py [streams] [rate]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from openwakeword_amd import weights as W, resample as R, _lib
from openwakeword_amd.engine import StreamEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
rate = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
dev = torch.device("cuda", 0)
eng = StreamEngine(S, {"alexa": W.synthetic_head("alexa", 1234)}, W.synthetic_embedding(1234))
p, q, taps = R.design(rate)
n_in = 1280 * p // q
x = (torch.randn(S, n_in, device=dev) * 3000).to(torch.int16)
y = torch.empty(S, 1280, device=dev, dtype=torch.int16)
def run():
    _lib.check(eng._lib.oww_resample(eng._h, C.c_void_p(x.data_ptr()), 1, n_in, p, q, taps.ctypes.data_as(C.c_void_p), taps.shape[1],
                                     C.c_void_p(y.data_ptr()), 1, 1280))
for _ in range(3): run()
eng.sync(); t0 = time.perf_counter()
for _ in range(20): run()
eng.sync(); dt = (time.perf_counter() - t0) / 20
ref = R.apply_numpy(x[:4].cpu().numpy(), rate)
err = np.abs(y[:4].cpu().numpy().astype(np.int32) - ref.astype(np.int32)).max()
print({"streams": S, "rate_in": rate, "n_in": n_in, "taps": int(taps.shape[1]), "ms": round(dt * 1e3, 3),
       "GB_s": round((S * (n_in + 1280) * 2) / dt / 1e9, 1), "max_abs_diff_vs_numpy": int(err)})
