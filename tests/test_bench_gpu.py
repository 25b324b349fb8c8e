"""bench.py end to end on one GPU box (-m gpu): the JSON contract of the N = 1 line (parity block, roofline, extras
switched off for speed) and of the N = 2 line.  The N = 2 run uses the OWW_BENCH_ONE_GPU testing aid -- both ranks on
device 0, gloo instead of RCCL -- so it verifies the launch / sharding / gather-every-K / max-over-ranks code path that the
driver's multi-GPU run takes; RCCL itself stays unexercised on a 1-GPU box."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _last_json(text: str) -> dict:
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_single_gpu_line_carries_parity_and_roofline():
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--streams", "8192",
           "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["scores_valid"] and out["f16_range_flag"] is False
    assert out["parity"]["n_pairs"] == 1024 and out["parity"]["ok"] and out["parity"]["max_abs_err"] <= 1e-4
    assert out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1
    assert out["launches_per_step"] == 6 and out["config"]["rccl_ranks"] == 0 and out["config"]["gather"] == "none"
    assert abs(out["value"] - 8192 * 6 / (out["ms_per_step"] * 6e-3)) / out["value"] < 1e-3


def test_bare_two_rank_invocation_launches_itself():
    """The driver's N = 1 command is a bare `python3 bench.py --gpus 1`; if its N > 1 command is bare too, bench.py has to start its
    own ranks (VERDICT r03 missing 1).  Both ranks on device 0 over gloo (OWW_BENCH_ONE_GPU), as in the torchrun form below."""
    env = dict(os.environ, OWW_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--streams", "4096", "--cpu-seconds", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                             # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    # the N > 1 line is complete: rank 0 timed the CPU baseline before it joined the process group, and the
    # roofline comes from rank 0's per-kernel pass
    assert out["cpu_baseline"] is not None and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] in ("port", "reference")
    assert out["roofline"] is not None and out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1
    assert out["pre_rendezvous_s"] >= 0
    assert out["n_gpus"] == 2 and out["config"]["sharding"] == "stream-range x2" and out["config"]["rccl_ranks"] == 0
    assert out["config"]["gather"] == "dist" and out["parity"]["n_pairs"] == 2048 and out["parity"]["ok"]
    assert abs(out["value"] - 2 * 4096 * 5 / (out["ms_per_step"] * 5e-3)) / out["value"] < 1e-3


def test_c_abi_gather_inside_the_timed_region_world_of_one():
    """--gather c_abi: the scores of every timed step go through oww_gather_scores (grouped ncclSend / ncclRecv on the handle's
    stream); with one rank that is RCCL's send-to-self, and config.rccl_ranks is what ncclCommCount reports."""
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--streams", "8192", "--gather", "c_abi",
           "--no-cpu-baseline", "--no-extras", "--no-parity"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _last_json(r.stdout)
    assert out["config"]["gather"] == "c_abi" and out["config"]["rccl_ranks"] == 1 and "oww_gather_scores" in out["config"]["collective"]
    assert out["scores_valid"] and out["launches_per_step"] == 6


def test_two_rank_line_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OWW_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "7", "--warmup", "2", "--streams", "4096",
           "--gather-every", "3", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 7
    assert out["config"]["sharding"] == "stream-range x2" and "every 3 step" in out["config"]["collective"]
    assert abs(out["value"] - 2 * out["frames_per_sec_per_gpu"]) / out["value"] < 1e-6
    assert abs(out["value"] - 2 * 4096 * 7 / (out["ms_per_step"] * 7e-3)) / out["value"] < 1e-3
    assert out["parity"]["n_pairs"] == 2048 and out["parity"]["ok"]
    assert "sustained" not in out and "resident_1m" not in out and out["cpu_baseline"] is None
