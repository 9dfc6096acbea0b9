"""Multi-GPU sharding of the decomposition path: one process per GPU,
``torch.distributed`` with the nccl backend (= RCCL over xGMI on ROCm; gloo in the CPU
tests).  The path needs exactly one bulk exchange -- the sum of per-rank partial
covariance matrices -- plus a few scalars and the broadcast of finished factors.

Work is dealt round-robin by a deterministic index (calibration step, candidate rank or
layer number), every rank advances the data iterators identically, so the union of the
ranks' work is exactly the sequential reference stream.
"""

from __future__ import annotations

from typing import Any, Optional

import torch


class Shard:
    def __init__(self, group: Any = None, rank: int = 0, world: int = 1):
        self.group = group
        self.rank = rank
        self.world = world

    @classmethod
    def from_env(cls, group: Any = None) -> "Shard":
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return cls()
        world = dist.get_world_size(group)
        if world <= 1:
            return cls()
        return cls(group, dist.get_rank(group), world)

    @property
    def active(self) -> bool:
        return self.world > 1

    def mine(self, index: int) -> bool:
        """Round-robin ownership of a work item (calibration step, candidate, layer)."""
        return index % self.world == self.rank

    def owner(self, index: int) -> int:
        return index % self.world

    def owns(self, index: int) -> bool:
        return self.mine(index)

    def _global_rank(self, group_rank: int) -> int:
        import torch.distributed as dist

        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def all_reduce_small(self, t: torch.Tensor) -> None:
        import torch.distributed as dist

        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast_from_owner(self, t: Optional[torch.Tensor], index: int, shape, dtype, device) -> torch.Tensor:
        """The owner of `index` holds `t`; everyone returns a copy of it."""
        import torch.distributed as dist

        if not self.owns(index):
            t = torch.empty(tuple(shape), dtype=dtype, device=device)
        else:
            t = t.contiguous()
        dist.broadcast(t, src=self._global_rank(self.owner(index)), group=self.group)
        return t

    def broadcast_object(self, obj: Any, index: int) -> Any:
        import torch.distributed as dist

        box = [obj if self.owns(index) else None]
        dist.broadcast_object_list(box, src=self._global_rank(self.owner(index)), group=self.group)
        return box[0]
