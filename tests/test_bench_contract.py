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
