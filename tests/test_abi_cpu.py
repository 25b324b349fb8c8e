"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol include/owwhip.h declares,
and refuses to work without a GPU instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from openwakeword_amd import _build, _lib, engine, weights as W

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "owwhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(oww_[a-z_0-9]+)\s*\(", src)))


def test_library_is_built_and_exports_the_header():
    assert os.path.exists(_build.lib_path()), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/owwhip.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype"
    assert lib.oww_abi_version() == _lib.ABI_VERSION == 6


def test_blob_layouts():
    mel = engine.pack_mel_blob()
    assert mel.nbytes == (400 + 32 + 32 * 16) * 4
    emb = engine.pack_embedding_blob(W.synthetic_embedding())
    assert emb.size == 326_808 + 2 * (24 * 3 + 48 * 4 + 72 * 4 + 96 * 8)          # convs + folded BN scale/shift
    h = engine.pack_head_blob(W.synthetic_head("alexa"))
    assert h.size == 8 + 102_849
    hj = engine.pack_head_blob(W.synthetic_head("hey_jarvis"))
    assert hj.size == 8 + 2 * 102_849
    ht = engine.pack_head_blob(W.synthetic_head("timer"))
    assert ht.size == 8 + 435_335
    with pytest.raises(ValueError):
        bad = W.synthetic_embedding()
        bad["conv"][3] = bad["conv"][3][..., :40]
        engine.pack_embedding_blob(bad)


def test_argument_errors_surface_as_codes_not_crashes():
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.Config(0, 0, 1, 0, 1, 0, None)
    assert lib.oww_create(C.byref(cfg), C.byref(h)) == -1            # OWW_EINVAL: n_streams < 1
    assert b"n_streams" in lib.oww_last_error()
    assert lib.oww_step(None, None, 0, 1, None, 0) < 0
    assert lib.oww_destroy(None) == 0


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.OwwError):
        engine.StreamEngine(2, {"alexa": W.synthetic_head("alexa")})


def test_integration_stubs_compile_and_bind_only_declared_symbols():
    """INTEGRATION.md section 1 (executed on the GPU by tests/test_seam_gpu.py): here, without a GPU, the printed stubs must at
    least be valid Python, quote the header's ABI version, and call nothing the header does not declare."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"<!-- stub:(\w+) -->\s*```python\n(.*?)```", text, re.S)
    assert [t for t, _ in blocks] == ["features", "heads"]
    names = set(declared_symbols())
    for tag, body in blocks:
        src = body if tag == "features" else "def _install(idx, T, n_out, mdl_name):\n" + body
        compile(src, f"INTEGRATION.md:{tag}", "exec")
        for sym in re.findall(r"lib\.(oww_[a-z_0-9]+)", body):
            assert sym in names, f"INTEGRATION.md stub calls {sym}, which include/owwhip.h does not declare"
    assert f"lib.oww_abi_version() == {_lib.ABI_VERSION}" in text
    assert f"C ABI version {_lib.ABI_VERSION} " in text


def _gcc(args, cwd=None):
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    return subprocess.run([gcc] + args, cwd=cwd, capture_output=True, text=True)


def test_header_is_plain_c99_and_the_ctypes_struct_has_its_layout(tmp_path):
    """include/owwhip.h is what a C / cgo / JNI host compiles against: it must be valid C99 on its own, and the ctypes mirror the
    Python layer uses must describe the SAME struct (a mismatch would pass every Python test and break every C caller)."""
    src = tmp_path / "layout.c"
    fields = [f for f, _ in _lib.Config._fields_]
    src.write_text('#include <stdio.h>\n#include "owwhip.h"\nint main(void) {\n'
                   '  printf("sizeof %zu\\n", sizeof(oww_config));\n'
                   + "".join(f'  printf("{f} %zu\\n", offsetof(oww_config, {f}));\n' for f in fields)
                   + '  printf("abi %d chunk %d emb %d ring %d comm %d classes %d\\n", OWW_ABI_VERSION, OWW_CHUNK, OWW_EMB_DIM, OWW_SCORE_RING, '
                     'OWW_COMM_ID_BYTES, OWW_N_KERNEL_CLASSES);\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    r = _gcc(["-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    import subprocess
    out = dict(line.split(" ", 1) for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof"]) == C.sizeof(_lib.Config)
    for f in fields:
        assert int(out[f]) == getattr(_lib.Config, f).offset, f
    assert out["abi"].split() == ["6", "chunk", "1280", "emb", "96", "ring", "30", "comm", "128", "classes", "10"]
    assert _lib.ABI_VERSION == 6 and engine.CHUNK == 1280 and engine.EMB_DIM == 96


def test_c_consumer_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/c/owwhip_demo.c (the ABI driven from plain C) compiles warning-free against the header and links against the
    library; without a GPU it stops at oww_create with the library's error text, not with a crash or a fallback."""
    import subprocess
    import torch
    exe = tmp_path / "owwhip_demo"
    libdir = os.path.dirname(_build.lib_path())
    r = _gcc(["-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
              os.path.join(ROOT, "examples", "c", "owwhip_demo.c"), "-L", libdir, "-lowwhip", f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_seam_gpu.py runs the program for real")
    run = subprocess.run([str(exe), str(tmp_path), "4", "2"], capture_output=True, text=True)
    assert run.returncode == 1 and "oww_create" in run.stderr and "-> -2" in run.stderr
