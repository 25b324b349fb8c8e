"""Build libowwhip.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch extension machinery:
the library is a plain C-ABI shared object (include/owwhip.h) loaded through ctypes."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "owwhip.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("owwhip_kernels.h", "owwhip_rr.h", "owwhip_hx.h", "owwhip_vad.h", "owwhip_fused.h")] + \
       [os.path.join(ROOT, "include", "owwhip.h")]
LIB = os.path.join(HERE, "libowwhip.so")


def lib_path() -> str:
    """OWW_LIB selects an alternative build of the same library (kernel-variant A/B runs only)."""
    return os.environ.get("OWW_LIB") or LIB


def source_hash(defines: tuple = (), extra_flags: str = "") -> str:
    """First 16 hex digits of the SHA-256 over what determines the library's kernels: the sources (csrc/ + include/owwhip.h), the -D
    defines of a variant build and any extra hipcc flags (OWW_HIPCC_FLAGS) -- compiled into the library (oww_build_info) and stored
    with every rocprofv3 counter summary under profiles/, so that bench.py can tell whether a committed PMC pass belongs to the
    kernels it is timing (an A/B build with OWH_* defines must not be priced with the default build's counters)."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    variant = " ".join(sorted(defines)) + "|" + " ".join(extra_flags.split())
    if variant != "|":                         # (the default build hashes exactly as before: sources only)
        h.update(b"\0variant\0" + variant.encode())
    return h.hexdigest()[:16]


def is_fresh() -> bool:
    return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)


def build(force: bool = False, verbose: bool = False, out: str | None = None, defines: tuple = ()) -> str:
    target = out or LIB
    if not force and out is None and is_fresh():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", SRC,
           "-I" + os.path.join(ROOT, "include"), "-o", target + ".tmp", "-Wall", "-Wno-unused-function"]
    cmd += ["-D" + d for d in defines]
    extra = os.environ.get("OWW_HIPCC_FLAGS", "")
    cmd += ['-DOWW_SRC_SHA16="' + source_hash(tuple(defines), extra) + '"']
    cmd += extra.split()          # (diagnostic builds only, e.g. -mllvm -amdgpu-waitcnt-forcezero)
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
