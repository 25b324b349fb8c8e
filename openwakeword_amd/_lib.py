"""ctypes binding of libowwhip.so (C ABI in include/owwhip.h).  Fails loudly when the library is
missing -- there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

from . import _build

_lib = None
ABI_VERSION = 6          # OWW_ABI_VERSION of include/owwhip.h this binding was written against
ERANGE = -5


class OwwError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_streams", C.c_int32), ("max_chunks", C.c_int32),
                ("feature_ring", C.c_int32), ("use_mfma", C.c_int32), ("debug_layers", C.c_int32),
                ("stream", C.c_void_p)]


# every symbol include/owwhip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "oww_abi_version": (C.c_int, []),
    "oww_build_info": (C.c_char_p, []),
    "oww_last_error": (C.c_char_p, []),
    "oww_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "oww_destroy": (C.c_int, [_P]),
    "oww_load_mel": (C.c_int, [_P, _P, C.c_size_t]),
    "oww_load_embedding": (C.c_int, [_P, _P, C.c_size_t]),
    "oww_add_head": (C.c_int, [_P, _P, C.c_size_t]),
    "oww_commit": (C.c_int, [_P]),
    "oww_set_calibration": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "oww_calibration_info": (C.c_int, [_P, _P, _P, _P, _P]),
    "oww_n_labels": (C.c_int, [_P]),
    "oww_reset": (C.c_int, [_P, _P, C.c_int32, _P]),
    "oww_set_postproc": (C.c_int, [_P, _P, _P, C.c_int32]),
    "oww_step": (C.c_int, [_P, _P, C.c_int, C.c_int32, _P, C.c_int]),
    "oww_step_masked": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, C.c_int]),
    "oww_sync": (C.c_int, [_P]),
    "oww_range_status": (C.c_int, [_P, C.c_int]),
    "oww_range_where": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "oww_submit": (C.c_int, [_P, _P, C.c_int32]),
    "oww_submit_masked": (C.c_int, [_P, _P, _P]),
    "oww_collect": (C.c_int, [_P, _P]),
    "oww_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "oww_host_free": (C.c_int, [_P]),
    "oww_set_verifier": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_float, C.c_float]),
    "oww_set_vad_threshold": (C.c_int, [_P, C.c_float]),
    "oww_push_vad": (C.c_int, [_P, _P, C.c_int]),
    "oww_load_vad": (C.c_int, [_P, _P, C.c_size_t]),
    "oww_get_vad": (C.c_int, [_P, _P]),
    "oww_reset_vad": (C.c_int, [_P, _P, C.c_int32]),
    "oww_scores_dev": (_P, [_P]),
    "oww_get_raw": (C.c_int, [_P, _P]),
    "oww_resample": (C.c_int, [_P, _P, C.c_int, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int, C.c_int32]),
    "oww_mel": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "oww_mel_clips": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "oww_embed": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "oww_embed_clips": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32]),
    "oww_head": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P]),
    "oww_get_features": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "oww_get_mel": (C.c_int, [_P, C.c_int32, _P, C.c_int32]),
    "oww_debug_read": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32]),
    "oww_debug_profile": (C.c_int, [_P, _P, C.c_int32]),
    "oww_enable_timing": (C.c_int, [_P, C.c_int]),
    "oww_kernel_times": (C.c_int, [_P, _P, _P]),
    "oww_use_graph": (C.c_int, [_P, C.c_int]),
    "oww_comm_id": (C.c_int, [_P]),
    "oww_comm_init": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "oww_gather_scores": (C.c_int, [_P, _P, _P]),
    "oww_comm_count": (C.c_int, [_P, _P]),
    "oww_comm_destroy": (C.c_int, [_P]),
}


def load():
    """dlopen libowwhip.so once.  torch is imported first when available so that the HIP runtime
    torch bundles (soname libamdhip64.so.7) is the one and only runtime in the process."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.lib_path()
    if not os.path.exists(path):
        raise OwwError(f"{path} is missing: build it with `python -m openwakeword_amd._build` "
                       "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    if lib.oww_abi_version() != ABI_VERSION:
        raise OwwError("libowwhip.so ABI version mismatch")
    _lib = lib
    return lib


class OwwRangeError(OwwError):
    """OWW_ERANGE: an activation left the f16 range of the fp16-split kernels (sticky per handle)."""


def check(rc: int) -> int:
    if rc == ERANGE:
        raise OwwRangeError(f"libowwhip error {rc}: {load().oww_last_error().decode(errors='replace')}")
    if rc < 0:
        raise OwwError(f"libowwhip error {rc}: {load().oww_last_error().decode(errors='replace')}")
    return rc
