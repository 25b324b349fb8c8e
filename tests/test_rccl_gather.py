"""oww_comm_* / oww_gather_scores: the C ABI's own RCCL exchange (ncclSend / ncclRecv, no torch.distributed).

gpurun boxes have one GPU, so what can be exercised here is a world of ONE: librccl is bound at run time, a communicator is
created, and the gather runs as a grouped send-to-self / receive-from-self on the handle's stream -- the same code path every
rank takes at any world size (rank 0 receives from each rank including itself).  RCCL refuses two ranks on one device, so a second
rank cannot be exercised on this pool; bench.py runs the same exchange at N > 1 as its `c_abi_gather` record (under a watchdog) when
the driver has a multi-GPU node.  Multi-GPU scaling itself stays unmeasured on this pool."""
import numpy as np
import pytest

from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

pytestmark = pytest.mark.gpu


def test_world_of_one_gathers_through_rccl():
    import torch
    heads = {n: W.synthetic_head(n, 1234) for n in ("alexa", "hey_jarvis")}
    S = 300
    eng = StreamEngine(S, heads, W.synthetic_embedding(1234))
    try:
        eng.comm_init(StreamEngine.comm_id(), 0, 1)
        assert eng.comm_count() == 1                          # ncclCommCount: what bench.py reports as config.rccl_ranks
        with pytest.raises(Exception):                       # a second communicator on the same handle is refused
            eng.comm_init(StreamEngine.comm_id(), 0, 1)
        pcm = W.synthetic_pcm(S, 1280 * 8, seed=5)
        out = torch.full((S, eng.n_labels), -1.0, device="cuda", dtype=torch.float32)
        for t in range(8):
            want = eng.step(pcm[:, 1280 * t:1280 * (t + 1)])
            eng.gather_scores(out.data_ptr(), [S])
            eng.sync()
            np.testing.assert_array_equal(out.cpu().numpy(), want)
        assert want.max() > 0.0
        with pytest.raises(Exception):                       # counts must describe this handle
            eng.gather_scores(out.data_ptr(), [S + 1])
        eng.comm_destroy()
        eng.comm_destroy()                                   # idempotent
    finally:
        eng.close()
