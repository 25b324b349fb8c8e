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

from typing import List, Optional, Tuple

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
    """Per-step gather of score blocks to rank 0.

    `gather(local)` takes this rank's [S_r, n_labels] fp32 tensor (device or CPU, matching the process group's
    backend) and returns the global [total, n_labels] tensor on rank 0 (None elsewhere), rows in global stream
    order.  Blocks are padded to the largest shard so that one fixed-size collective is issued per step; the
    receive buffers are allocated once."""

    def __init__(self, total_streams: int, n_labels: int, device: torch.device,
                 group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total, self.n_labels = int(total_streams), int(n_labels)
        self.ranges = [stream_range(r, self.world, self.total) for r in range(self.world)]
        self.lo, self.hi = self.ranges[self.rank]
        self.max_rows = max(hi - lo for lo, hi in self.ranges)
        self._send = torch.zeros(self.max_rows, n_labels, dtype=torch.float32, device=device)
        self._recv: Optional[List[torch.Tensor]] = None
        self._out: Optional[torch.Tensor] = None
        if self.rank == 0:
            self._out = torch.empty(self.total, n_labels, dtype=torch.float32, device=device)
            if self.world > 1:
                self._recv = [torch.empty_like(self._send) for _ in range(self.world)]

    @property
    def local_streams(self) -> int:
        return self.hi - self.lo

    def gather(self, local: torch.Tensor) -> Optional[torch.Tensor]:
        if local.shape != (self.local_streams, self.n_labels):
            raise ValueError(f"expected local scores {(self.local_streams, self.n_labels)}, got {tuple(local.shape)}")
        if self.world == 1:
            self._out.copy_(local)
            return self._out
        even = all(hi - lo == self.max_rows for lo, hi in self.ranges)
        send = local if (even and local.is_contiguous()) else self._send
        if send is self._send:
            self._send[: self.local_streams].copy_(local)
        dist.gather(send, self._recv, dst=0, group=self.group)
        if self.rank != 0:
            return None
        for r, (lo, hi) in enumerate(self.ranges):
            self._out[lo:hi].copy_(self._recv[r][: hi - lo])
        return self._out
