"""Does torch still see the GPU when libowwhip.so (linked against the system ROCm's libamdhip64.so.7) was loaded BEFORE torch
(which bundles its own copy of the runtime)?  python tools/diag_torch_interop.py raw|raw_init|torch_first"""
import ctypes as C
import os
import sys
sys.path.insert(0, ".")
mode = sys.argv[1]
path = os.path.abspath("openwakeword_amd/libowwhip.so")


class Cfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_streams", C.c_int32), ("max_chunks", C.c_int32), ("feature_ring", C.c_int32),
                ("use_mfma", C.c_int32), ("debug_layers", C.c_int32), ("stream", C.c_void_p)]


def create(lib):
    h = C.c_void_p()
    cfg = Cfg(0, 4, 1, 0, 3, 0, None)
    lib.oww_last_error.restype = C.c_char_p
    rc = lib.oww_create(C.byref(cfg), C.byref(h))
    return rc, lib.oww_last_error()


if mode == "torch_first":
    import torch
    lib = C.CDLL(path)
    print("torch first: oww_create", create(lib), "| torch sees the GPU:", torch.cuda.is_available())
else:
    lib = C.CDLL(path)                      # no torch in the process yet: the system runtime gets loaded
    if mode == "raw_init":
        print("library first: oww_create", create(lib))
    import torch
    print(f"{mode}: torch imported afterwards sees the GPU:", torch.cuda.is_available())
    print("maps:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l}))
