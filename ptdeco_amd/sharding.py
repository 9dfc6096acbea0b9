"""Multi-GPU sharding of the decomposition path: one process per GPU,
``torch.distributed`` with the nccl backend (= RCCL over xGMI on ROCm; gloo in the CPU
tests).  The path needs exactly one bulk exchange -- the sum of per-rank partial
covariance matrices -- plus a few scalars and the broadcast of finished factors.

Work is dealt round-robin by a deterministic index (calibration step, candidate rank or
layer number), every rank advances the data iterators identically, so the union of the
ranks' work is exactly the sequential reference stream.

Traffic: only the layer's owner runs the eigendecomposition, so the partial
covariance sums are REDUCED to the owner (half the bytes a ring all-reduce moves per link) and
only the live lower triangle travels (half again); the owner sends back the top-k eigenvectors in
the weight dtype (n k elements instead of n^2 f64).

PTD_COV_COLLECTIVE=allreduce switches the covariance exchange to north_star's literal collective -- an all-reduce of the
packed lower triangle, every rank receives the sum, only the owner uses it -- so that the two forms can be compared on
a node (bench.py reports `cov_collective`); the default is "reduce" (to the owner).  Results are identical: the same
sums reach the same owners.
"""

from __future__ import annotations

from typing import Any, Optional

import torch


PACK_BLOCK = 512


def pack_lower(E: torch.Tensor, block: int = PACK_BLOCK) -> torch.Tensor:
    """The lower triangle of E (n x n, rows [i0, i1) contribute E[i0:i1, :i1]) as one flat buffer:
    n^2 / 2 + n block / 2 elements, no index tensors."""
    n = E.shape[0]
    return torch.cat([E[i0:min(n, i0 + block), :min(n, i0 + block)].reshape(-1) for i0 in range(0, n, block)])


def unpack_lower(packed: torch.Tensor, E: torch.Tensor, block: int = PACK_BLOCK) -> None:
    """Inverse of pack_lower: writes the slabs back into E (entries above the slabs are untouched)."""
    n, off = E.shape[0], 0
    for i0 in range(0, n, block):
        i1 = min(n, i0 + block)
        cnt = (i1 - i0) * i1
        E[i0:i1, :i1] = packed[off:off + cnt].view(i1 - i0, i1)
        off += cnt


class Shard:
    def __init__(self, group: Any = None, rank: int = 0, world: int = 1):
        import os

        self.group = group
        self.rank = rank
        self.world = world
        self.collective = os.environ.get("PTD_COV_COLLECTIVE", "reduce").lower()
        if self.collective not in ("reduce", "allreduce"):
            raise ValueError(f"PTD_COV_COLLECTIVE={self.collective!r}: expected reduce or allreduce")

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

    def reduce_lower_to_owner(self, E: torch.Tensor, index: int) -> None:
        """Sum the lower triangles of every rank's E on the owner of `index` (in place there; the other
        ranks' E is left as it was: they do not use it)."""
        import torch.distributed as dist

        packed = pack_lower(E)
        if self.collective == "allreduce":
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.reduce(packed, dst=self._global_rank(self.owner(index)), op=dist.ReduceOp.SUM, group=self.group)
        if self.owns(index):
            unpack_lower(packed, E)

    def reduce_lower_to_owner_async(self, E: torch.Tensor, index: int):
        """The same sum, started without waiting for it (``async_op``): the collective runs on the communicator's own
        stream while this rank goes on issuing work -- the owner of layer l starts its eigensolve as soon as l's sum
        has arrived, the sums of l + 1, l + 2, ... (other owners) travel meanwhile.  Returns the function that
        completes the exchange on the calling thread's current stream (the owner's E is final after it); every rank
        must start the collectives of a pass in the same order."""
        import torch.distributed as dist

        packed = pack_lower(E)
        if self.collective == "allreduce":
            work = dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.reduce(packed, dst=self._global_rank(self.owner(index)), op=dist.ReduceOp.SUM, group=self.group,
                               async_op=True)

        def complete() -> None:
            work.wait()
            if self.owns(index):
                unpack_lower(packed, E)
        return complete

    def reduce_small_to_owner_async(self, t: torch.Tensor, index: int):
        import torch.distributed as dist

        work = dist.reduce(t, dst=self._global_rank(self.owner(index)), op=dist.ReduceOp.SUM, group=self.group,
                           async_op=True)
        return work.wait

    def all_reduce_lower(self, E: torch.Tensor) -> None:
        """Sum of the lower triangles on every rank (a statistic several owners need)."""
        import torch.distributed as dist

        packed = pack_lower(E)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
        unpack_lower(packed, E)

    def reduce_small_to_owner(self, t: torch.Tensor, index: int) -> None:
        import torch.distributed as dist

        dist.reduce(t, dst=self._global_rank(self.owner(index)), op=dist.ReduceOp.SUM, group=self.group)

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
