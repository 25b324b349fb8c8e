"""oww_step with a device PCM pointer that is only 2-byte aligned (e.g. a slice of a larger int16 tensor): the fused front end needs
16-byte rows, so such a call takes the separate mel kernel (scalar sample loads) -- same scores to fp32 round-off, not the same bits."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from openwakeword_amd import weights as W
from openwakeword_amd.engine import StreamEngine

S = 300
dev = torch.device("cuda", 0)
pcm = (torch.randn(8, S * 1280 + 8, device=dev) * 3000).round().clamp(-32768, 32767).to(torch.int16)
for off in (0, 1, 2, 3, 4):
    res = []
    for aligned in (True, False):
        eng = StreamEngine(S, {"alexa": W.synthetic_head("alexa", 1)}, W.synthetic_embedding(1))
        sc = torch.empty(S, 1, device=dev)
        out = []
        for t in range(8):
            x = pcm[t, off:off + S * 1280]
            if aligned:
                x = x.clone()
            torch.cuda.synchronize()
            assert aligned or x.data_ptr() % 16 == (pcm[t].data_ptr() + 2 * off) % 16
            eng.step_device(x.data_ptr(), 1, sc.data_ptr())
            eng.sync()
            out.append(sc.cpu().numpy().copy())
        eng.close()
        res.append(np.stack(out))
    print(f"offset {off} samples (pointer % 16 = {(2 * off) % 16}): identical to the aligned copy: {np.array_equal(res[0], res[1])}, "
          f"max |difference| {np.abs(res[0] - res[1]).max():.3g} (scores up to {res[0].max():.3f})")
