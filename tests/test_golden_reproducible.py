"""Where the reference checkout is present (the build container; never the GPU box), the committed golden vectors must be what the
generators produce TODAY: tests/golden/make_golden_onnx.py is re-run into a scratch file and compared with tests/golden/
ref_onnx_files.npz.  Guards against vectors that were edited, or generated from other weights / another exporter than the tests use."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("OWW_REFERENCE_DIR", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "openwakeword")), reason="the reference checkout is not on this machine")
def test_reference_on_exported_files_vectors_are_reproducible(tmp_path):
    pytest.importorskip("torch")
    gen = os.path.join(ROOT, "tests", "golden", "make_golden_onnx.py")
    src = open(gen).read()
    assert 'os.path.join(os.path.dirname(__file__), "ref_onnx_files.npz")' in src
    scratch = tmp_path / "make_golden_onnx.py"                                   # same script, output redirected
    scratch.write_text(src.replace('os.path.join(os.path.dirname(__file__), "ref_onnx_files.npz")', repr(str(tmp_path / "out.npz")))
                          .replace('os.path.dirname(__file__)', repr(os.path.join(ROOT, "tests", "golden"))))
    r = subprocess.run([sys.executable, str(scratch)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    if r.returncode != 0 and "OnnxExporterError" in r.stderr:
        pytest.skip("torch.onnx.export is not usable in this environment")
    assert r.returncode == 0, r.stderr[-2000:]
    new, old = np.load(tmp_path / "out.npz"), np.load(os.path.join(ROOT, "tests", "golden", "ref_onnx_files.npz"))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        if old[k].dtype.kind in "US":
            assert list(new[k]) == list(old[k]), k
        else:
            np.testing.assert_allclose(new[k], old[k], rtol=0, atol=1e-4, err_msg=k)     # (fp32 BLAS summation order varies with the thread count)
