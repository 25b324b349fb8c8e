"""One process, many engine life cycles with different weights: the pattern an intermittent GPU memory-access fault inside oww_commit was
seen under in rounds 3-4.  tests/soak_create.py does the cycling (nine weight regimes, 1 .. 16,480
streams and now and then 131,072, one to three heads, both kernel families, with / without the voice-activity network, three
kinds of calibration audio, then two host threads creating handles at the same time); this test runs it as a subprocess -- a device
fault aborts the process, and OWW_GUARD_ALLOC is read once per process -- plainly and with every library buffer in its own guarded
mapping (pushed against the lower and the upper end).  The reference's counterpart: Model objects are constructed and dropped
freely (utils.py:502-536, tests/test_models.py throughout)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("guard,cycles,thread_cycles,max_streams", [("0", 500, 30, 131072), ("2", 500, 20, 4096), ("1", 40, 0, 4096)])
def test_create_calibrate_destroy_soak(guard, cycles, thread_cycles, max_streams, tmp_path):
    env = dict(os.environ, OWW_GUARD_ALLOC=guard)
    err_path = tmp_path / "stderr.txt"                 # (guard mode prints one line per allocation)
    with open(err_path, "w") as ef:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_create.py"), "--cycles", str(cycles),
                            "--thread-cycles", str(thread_cycles), "--max-streams", str(max_streams)], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=ef, text=True, timeout=900)
    tail = "\n".join(p.stdout.splitlines()[-6:])
    err_tail = ""
    if p.returncode != 0:
        with open(err_path) as ef:
            err_tail = "".join([l for l in ef.readlines() if "owwhip guard" not in l][-30:])
    assert p.returncode == 0, f"soak died (rc {p.returncode}) after:\n{tail}\n{err_tail}"
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("soak ok"), tail
    print("\n" + last)
