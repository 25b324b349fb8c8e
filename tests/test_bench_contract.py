"""bench.py's constants agree with SURVEY §8d / DESIGN §5 (no GPU needed: the module is imported, main() is not run)."""
import importlib.util
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_flops_match_the_survey():
    b = _bench()
    # incremental CNN: 5,612,544 MAC per stream-step = 11.225 MFLOP, split over the five stage kernels
    assert sum(b.STAGE_FLOPS.values()) == 2 * 5_612_544
    from openwakeword_amd import weights as W
    heads = {n: W.synthetic_head(n) for n in ("alexa", "hey_mycroft", "hey_jarvis")}
    per_net = 2 * (16 * 96 * 64 + 64 * 64 + 64)
    assert b.head_flops(heads) == 4 * per_net                       # hey_jarvis carries two networks
    assert b.MEL_BYTES == 2560 + 1024
    # f16-split kernels: MFMAs per stream-step times 16x16x32 MACs cover the algorithmic MACs three times (plus padding)
    for k, n in b.HX_MFMAS.items():
        assert n * 8192 >= 3 * b.STAGE_FLOPS[k] // 2


def test_per_layer_macs():
    from openwakeword_amd import weights as W
    new_rows = [8, 8, 8, 4, 4, 4, 4, 4, 4, 4, 4, 2, 2, 2, 2, 2, 2, 2, 2, 1]
    width = [32, 32, 32, 16, 16, 16, 16, 8, 8, 8, 8, 4, 4, 4, 4, 2, 2, 2, 2, 1]
    macs = sum(r * f * kh * kw * ci * co for r, f, (kh, kw, ci, co, _) in zip(new_rows, width, W.CNN_TOPOLOGY))
    assert macs == 5_612_544


def test_bare_multi_gpu_invocation_relaunches_itself_under_torchrun(monkeypatch):
    """`python bench.py --gpus N` with no torchrun environment (VERDICT r03 missing 1) re-executes its own command line under
    torch.distributed.run on 127.0.0.1 at a free port; inside a torchrun environment it does not."""
    import sys
    import types
    import pytest
    b = _bench()
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env")
        return types.SimpleNamespace(returncode=0)
    monkeypatch.setattr(b.subprocess, "run", fake_run)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_composite_roofline_is_priced_only_from_a_counter_pass_of_the_loaded_build(tmp_path, monkeypatch):
    """a kernel change without a fresh PMC pass must not price new times with old instruction counts.  The library
    carries the hash of its sources (oww_build_info), tools/pmc.sh stores it with every counter summary, and bench.py takes the newest
    profiles/rNN_<kind>.json whose hash matches -- or says why there is no composite."""
    import json
    import types
    from openwakeword_amd import _build, _lib
    b = _bench()
    have = b.loaded_build()
    assert have == _build.source_hash() and len(have) == 16            # the in-tree library was built from the in-tree sources
    assert _lib.load().oww_build_info().decode().startswith("src=" + have)
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    args = types.SimpleNamespace(valu=False, lds_mfma=False, fp32=False)
    kern = {"SQ_INSTS_MFMA": 512.0 * 131072, "SQ_INSTS_VALU": 3147.0 * 131072, "SQ_INSTS_LDS": 682.0 * 131072,
            "SQ_LDS_IDX_ACTIVE": 3.6e8, "SQ_LDS_BANK_CONFLICT": 4.7e7}
    (prof / "r04_instr.json").write_text(json.dumps({"stageA_hx": kern}))                                   # no hash: an old round's pass
    (prof / "r05_instr.json").write_text(json.dumps({"stageA_hx": kern, "_csrc_sha16": "0123456789abcdef"}))   # another build's pass
    b._PROFILE_CACHE.clear()
    out = b.issue_bound("stageA", 131072, 1.5, args)
    assert set(out) == {"unavailable"} and have in out["unavailable"]
    (prof / "r06_instr.json").write_text(json.dumps({"stageA_hx": kern, "_csrc_sha16": have}))
    b._PROFILE_CACHE.clear()
    out = b.issue_bound("stageA", 131072, 1.5, args)
    assert out["wave_instructions_per_stream_step"] == {"valu": 2635.0, "mfma": 512.0, "lds": 682.0} and have in out["source"] and "r06_instr" in out["source"]
    assert b.pmc_traffic("stageA", 131072, args) is None                # no traffic pass of this build
