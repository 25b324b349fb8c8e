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
    assert lib.oww_abi_version() == _lib.ABI_VERSION == 5


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
