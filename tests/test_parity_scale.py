"""Oracle-sampled parity at the BASELINE configurations: for 4,096 x 1 head
(configs[1]), 65,536 x 3 heads (configs[2]) and 131,072 x 3 heads (the per-GPU shard of configs[3]) 64 probe streams
with distinct audio sit at random stream ids of the full-size engine -- every other stream carries background noise --
and all 64 x 16 = 1,024 (stream, step) score vectors are compared with the CPU oracle (oracle/parity_sample.py).
Tolerances are the ones of tests/test_gpu_parity.py: scores 1e-4, embeddings 2e-4."""
import numpy as np
import pytest

from oracle import parity_sample as PS
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

TOL_SCORE = 1e-4
TOL_EMB = 2e-4


@pytest.fixture(scope="module")
def probe():
    pcm = PS.probe_pcm()
    ref = PS.oracle_reference()
    assert pcm.shape == (64, 16 * 1280) and len(np.unique(pcm, axis=0)) == 64
    assert ref["scores"].shape == (64, 16, 3)
    return pcm, ref


def test_probe_set_is_deterministic_and_varied():
    a, b = PS.probe_pcm(), PS.probe_pcm()
    assert np.array_equal(a, b)
    assert (a[5] == 0).all() and np.abs(a[6].astype(np.int32)).max() > 30000 and np.abs(a[7].astype(np.int32)).min() >= 32767
    ids = PS.probe_stream_ids(131072)
    assert len(set(ids.tolist())) == 64 and 0 in ids and 131071 in ids and ids.max() < 131072


def test_oracle_reference_worker_matches_direct_call():
    """(CPU) the pooled helper returns what a plain OracleModel loop returns."""
    from oracle import oww_oracle as O
    pcm = PS.probe_pcm(3, 6)
    ref = PS.oracle_reference(3, 6, ("alexa",), workers=2)
    emb, heads = PS._weights(("alexa",))
    m = O.OracleModel(heads, emb, init_noise=PS.init_noise())
    want = [m.predict(pcm[1, t * 1280:(t + 1) * 1280])["alexa"] for t in range(6)]
    np.testing.assert_allclose(ref["scores"][1, :, 0], want, rtol=0, atol=0)
    assert ref["init_features"].shape == (41, 96)


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams,head_names,family", [
    (4096, ("hey_jarvis",), 3),                    # BASELINE configs[1]
    (65536, PS.HEADS3, 3),                         # configs[2]
    (131072, PS.HEADS3, 3),                        # configs[3], one GPU's shard
    (131072, PS.HEADS3, 1),                        # the same on the exact-fp32 family
    (131072, PS.HEADS6, 3),                        # the reference's default Model(): six models, 12 score columns (model.py:84-87)
], ids=["c1_4096x1", "c2_65536x3", "c3_131072x3", "c3_131072x3_fp32", "default6_131072x6"])
def test_probe_streams_match_oracle_inside_a_full_size_batch(probe, n_streams, head_names, family):
    pcm, ref = probe
    if not set(head_names) <= set(PS.HEADS3):
        ref = PS.oracle_reference(head_names=head_names)
    ref_labels = [str(x) for x in ref["labels"]]
    cols = [ref_labels.index(l) for l in PS.labels_of(head_names)]
    emb, heads = PS._weights(head_names)
    ids = PS.probe_stream_ids(n_streams)
    r = np.random.default_rng(n_streams)
    background = np.clip(np.round(r.normal(0.0, 3000.0, (n_streams, 1280))), -32768, 32767).astype(np.int16)
    eng = StreamEngine(n_streams, heads, emb, use_mfma=family)
    worst = 0.0
    try:
        eng.reset(None, ref["init_features"][-eng.feature_ring:])
        buf = np.empty_like(background)
        for t in range(PS.N_FRAMES):
            np.copyto(buf, np.roll(background, 37 * t + 1, axis=1))
            buf[ids] = pcm[:, t * 1280:(t + 1) * 1280]
            got = eng.step(buf)
            assert np.isfinite(got).all() and got.min() >= 0.0 and got.max() <= 1.0
            want = ref["scores"][:, t][:, cols]
            worst = max(worst, float(np.abs(got[ids] - want).max()))
            np.testing.assert_allclose(got[ids], want, rtol=0, atol=TOL_SCORE, err_msg=f"frame {t}")
        # the probes' feature rings (last 16 embeddings) as well, for a third of them
        for k in range(0, PS.N_PROBE, 3):
            np.testing.assert_allclose(eng.get_features(int(ids[k]), 16), ref["features"][k], rtol=0, atol=TOL_EMB)
        assert eng.range_status() is False
        # scores of the probe set did not all collapse to one value (the comparison above is not vacuous)
        assert np.ptp(ref["scores"][:, 6:]) > 0.05
    finally:
        eng.close()
    print(f"\n{n_streams} streams x {len(head_names)} heads, family {family}: 1024 (stream, step) pairs, max |score - oracle| = {worst:.2e}")


@pytest.mark.gpu
def test_c4_131072x3_vad_probe_streams_match_gated_oracle():
    """BASELINE configs[4] at its per-GPU size: 131,072 streams x 3 heads with the voice-activity network and gate fused into every
    step.  The 64 probe streams x 16 frames are compared with OracleModel(vad_threshold=0.5, vad_session=StandinVadSession)
    -- the reference's gate (model.py:366-381) and VAD wrapper (vad.py:98-130) around the stand-in network; a gate decision whose
    compared value lies within 1e-3 of the threshold is skipped (as in test_fused_vad_gate_matches_oracle_model) and counted."""
    n_streams = 131072
    pcm = PS.probe_pcm()
    ref = PS.oracle_reference(vad=True)
    emb, heads = PS._weights(PS.HEADS3)
    ids = PS.probe_stream_ids(n_streams)
    r = np.random.default_rng(n_streams + 4)
    background = np.clip(np.round(r.normal(0.0, 3000.0, (n_streams, 1280))), -32768, 32767).astype(np.int16)
    eng = StreamEngine(n_streams, heads, emb, vad=W.synthetic_vad(PS.SEED_WEIGHTS), vad_threshold=PS.VAD_THRESHOLD)
    worst, skipped, n_gated, n_open = 0.0, 0, 0, 0
    try:
        eng.reset(None, ref["init_features"][-eng.feature_ring:])
        buf = np.empty_like(background)
        for t in range(PS.N_FRAMES):
            np.copyto(buf, np.roll(background, 37 * t + 1, axis=1))
            buf[ids] = pcm[:, t * 1280:(t + 1) * 1280]
            got = eng.step(buf)
            assert np.isfinite(got).all() and got.min() >= 0.0 and got.max() <= 1.0
            want = ref["scores"][:, t]
            g = ref["vad_window_max"][:, t]
            keep = ~(np.abs(g - PS.VAD_THRESHOLD) < 1e-3)            # NaN (empty window: gated by definition) compares False -> kept
            skipped += int((~keep).sum())
            worst = max(worst, float(np.abs(got[ids][keep] - want[keep]).max()))
            np.testing.assert_allclose(got[ids][keep], want[keep], rtol=0, atol=TOL_SCORE, err_msg=f"frame {t}")
            if t >= 6:
                n_gated += int((want[keep] == 0).all(axis=1).sum())
                n_open += int((want[keep] != 0).any(axis=1).sum())
        assert n_gated > 50 and n_open > 50 and skipped < 32         # both branches of the gate, and the skip rule is an exception
        assert eng.range_status() is False
    finally:
        eng.close()
    print(f"\nc4: {n_streams} streams x 3 heads + VAD gate: {1024 - skipped} (stream, step) pairs, max |score - oracle| = {worst:.2e}, "
          f"{skipped} decisions within 1e-3 of the threshold skipped, {n_gated} gated / {n_open} open after frame 6")


@pytest.mark.gpu
def test_scores_do_not_depend_on_the_batch_size_the_stream_sits_in():
    """A size-independent property at the full size: the same audio through stream s gives the same BITS whether s sits in an engine
    of 131,072 streams (two-slot weight rings, `__syncthreads`), 4,096 (three-slot rings with counted waits and bare barriers, deep
    heads ring) or 40 (partly filled tiles) -- same arithmetic in the same order in every launch shape."""
    import torch
    dev = torch.device("cuda", 0)
    emb, heads = PS._weights(PS.HEADS3)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    S0, T = 131072, 24
    pool = [(torch.randn(S0, 1280, device=dev, generator=g) * a).round().clamp(-32768, 32767).to(torch.int16) for a in (3000.0, 150.0, 14000.0)]
    got = {}
    for S in (S0, 4096, 40):
        eng = StreamEngine(S, heads, emb)
        try:
            sc = torch.empty(T, S, eng.n_labels, device=dev)
            pcm = [p[:S].contiguous() for p in pool]
            torch.cuda.synchronize()                     # (the engine runs on its own stream)
            for t in range(T):
                eng.step_device(pcm[t % 3].data_ptr(), 1, sc[t].data_ptr())
            eng.sync()
            got[S] = (sc.cpu().numpy(), np.stack([eng.get_features(s, 16) for s in (0, 39)]))
        finally:
            eng.close()
    assert np.isfinite(got[S0][0]).all() and got[S0][0][-1].max() > 0
    for S in (4096, 40):
        np.testing.assert_array_equal(got[S][0], got[S0][0][:, :S], err_msg=f"scores, engine of {S} streams")
        np.testing.assert_array_equal(got[S][1], got[S0][1], err_msg=f"feature rings, engine of {S} streams")
