"""
Stream sharding across the GPUs of one node.

Streams are independent (the reference keeps one `Model` object per stream and its only scale-out is
process-level fan-out over clips, /root/reference/openwakeword/utils.py:502-536), so the multi-GPU form of the
hot path is a pure partition: rank r owns the contiguous global stream range `stream_range(r, world, total)`,
all weights are replicated, all per-stream state is local, and there is no data-path collective.  The one
exchange is the delivery of results: every step each rank's fp32 score block [S_r, n_labels] is gathered to
rank 0 (`ScoreGather`), over RCCL/xGMI on GPUs (torch.distributed backend "nccl") or gloo on CPU (tests).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def stream_range(rank: int, world: int, total: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of global stream ids owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request rank={rank} world={world} total={total}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(stream_id: int, world: int, total: int) -> Tuple[int, int]:
    """(rank, local index) of a global stream id under `stream_range`."""
    if not (0 <= stream_id < total):
        raise ValueError(f"stream id {stream_id} outside [0, {total})")
    base, extra = divmod(total, world)
    cut = extra * (base + 1)
    if stream_id < cut:
        return stream_id // (base + 1), stream_id % (base + 1)
    return extra + (stream_id - cut) // base, (stream_id - cut) % base


class ScoreGather:
    """Delivery of score blocks to rank 0: one collective per `every` steps.

    `gather(local)` takes this rank's [S_r, n_labels] fp32 tensor (device or CPU, matching the process group's
    backend).  With every == 1 (default) it returns the global [total, n_labels] tensor on rank 0 (None elsewhere),
    rows in global stream order.  With every == K > 1 the block is stashed on the device and every K-th call issues
    ONE K-times larger collective and returns [K, total, n_labels] on rank 0 (None on the other calls and ranks):
    the exchange is latency- not bandwidth-bound (1.5 MB per GPU and step at 131,072 x 3), so batching K steps
    amortises the launch + rendezvous cost; `flush()` delivers a partial batch.

    Even shards (total % world == 0) are received straight into the output buffer -- rank 0's receive list is
    `world` contiguous views of it, no unpack copies; uneven shards are padded to the largest one and unpacked."""

    def __init__(self, total_streams: int, n_labels: int, device: torch.device,
                 group: Optional[dist.ProcessGroup] = None, every: int = 1):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total, self.n_labels = int(total_streams), int(n_labels)
        self.every = max(1, int(every))
        self.ranges = [stream_range(r, self.world, self.total) for r in range(self.world)]
        self.lo, self.hi = self.ranges[self.rank]
        self.max_rows = max(hi - lo for lo, hi in self.ranges)
        self.even = all(hi - lo == self.max_rows for lo, hi in self.ranges)
        K = self.every
        self._send = torch.zeros(K, self.max_rows, n_labels, dtype=torch.float32, device=device)
        self._fill = 0
        self._recv: Optional[torch.Tensor] = None        # rank 0: [world, K, max_rows, n_labels], block r contiguous
        self._out: Optional[torch.Tensor] = None         # rank 0, uneven shards or K > 1: [K, total, n_labels]
        self.collectives = 0
        if self.rank == 0:
            self._recv = torch.empty(self.world, K, self.max_rows, n_labels, dtype=torch.float32, device=device)
            if not (self.even and K == 1):
                self._out = torch.empty(K, self.total, n_labels, dtype=torch.float32, device=device)

    @property
    def local_streams(self) -> int:
        return self.hi - self.lo

    def _exchange(self, k: int) -> Optional[torch.Tensor]:
        """One collective over the k (<= every) stashed steps."""
        self._fill = 0
        if self.world > 1:
            recv = [self._recv[r] for r in range(self.world)] if self.rank == 0 else None
            dist.gather(self._send, recv, dst=0, group=self.group)
            self.collectives += 1
        elif self.rank == 0:
            self._recv[0].copy_(self._send)
        if self.rank != 0:
            return None
        if self.even and self.every == 1:
            return self._recv.view(self.total, self.n_labels)            # received in place: blocks are already in stream order
        for r, (lo, hi) in enumerate(self.ranges):
            self._out[:k, lo:hi].copy_(self._recv[r, :k, : hi - lo])
        return self._out[0] if self.every == 1 else self._out[:k]

    def gather(self, local: torch.Tensor) -> Optional[torch.Tensor]:
        if local.shape != (self.local_streams, self.n_labels):
            raise ValueError(f"expected local scores {(self.local_streams, self.n_labels)}, got {tuple(local.shape)}")
        self._send[self._fill, : self.local_streams].copy_(local)
        self._fill += 1
        if self._fill < self.every:
            return None
        return self._exchange(self.every)

    def flush(self) -> Optional[torch.Tensor]:
        """Deliver the steps stashed since the last collective ([k, total, n_labels] on rank 0); None when there are none.
        Every rank must call it at the same point."""
        k = self._fill
        if k == 0:
            return None
        out = self._exchange(k)
        if out is not None and self.every == 1:
            out = out[None]
        return out
