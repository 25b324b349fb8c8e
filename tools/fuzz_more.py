import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np
from openwakeword_amd import weights as W
import tests.test_gpu_parity as T
emb = W.synthetic_embedding(1234)
bad = 0
for seed in range(100, 160):
    try:
        T.test_fuzz_kernel_families_agree.__wrapped__(emb, seed) if hasattr(T.test_fuzz_kernel_families_agree, "__wrapped__") else T.test_fuzz_kernel_families_agree(emb, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:300])
print("fuzz seeds 100..159:", "all ok" if not bad else f"{bad} failed")
