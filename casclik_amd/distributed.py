"""Data-parallel sharding of an instance batch over the GPUs of one node.

Instances are independent (the reference solves one robot per call,
casclik/controllers/pseudo_inverse.py:512-556), so the batch is cut into
contiguous row blocks, one per rank, and each rank runs its own kernel launch:
there is NO collective on the data path.  The only exchange is an optional
all-gather of the joint-velocity rows when the caller wants the full batch on
every rank (RCCL over xGMI with backend "nccl"; gloo in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_rows, rank, world_size):
    """[lo, hi) of the contiguous block owned by ``rank``; the first
    ``n_rows % world_size`` ranks get one extra row."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(int(n_rows), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_rows(tensor, rank=None, world_size=None):
    """Rows of a replicated [B, ...] tensor that belong to this rank."""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    lo, hi = shard_bounds(tensor.shape[0], rank, world_size)
    return tensor[lo:hi]


def all_gather_rows(local, n_rows_total=None, group=None):
    """Concatenate the per-rank row blocks [b_r, n] into [B, n] on every rank.

    Equal shards use one ``all_gather_into_tensor`` (a single RCCL all-gather);
    uneven shards are padded to the largest block and trimmed."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    n_cols = local.shape[1:]
    if n_rows_total is None:
        cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(cnt, group=group)
        n_rows_total = int(cnt.item())
    sizes = [shard_bounds(n_rows_total, r, world)[1] - shard_bounds(n_rows_total, r, world)[0]
             for r in range(world)]
    if sizes[rank] != local.shape[0]:
        raise ValueError("rank %d holds %d rows, the contiguous partition expects %d"
                         % (rank, local.shape[0], sizes[rank]))
    biggest = max(sizes)
    if min(sizes) == biggest:
        out = torch.empty((n_rows_total,) + tuple(n_cols), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((biggest,) + tuple(n_cols), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


class ShardedController(object):
    """Runs a controller on this rank's shard of a batch.

    ``solve_batch(t, Q_local, input_var=Y_local, gather=False)`` returns the
    local velocities; with ``gather=True`` the full [B, n] velocity matrix (one
    all-gather per call, reported separately by bench.py)."""

    def __init__(self, controller, group=None):
        self.controller = controller
        self.group = group

    def solve_batch(self, time_var, robot_var, input_var=None, gather=False, n_rows_total=None):
        res = self.controller.solve_batch(time_var, robot_var, input_var=input_var)
        dq = res[0]
        if gather and dist.is_initialized():
            if not isinstance(dq, torch.Tensor):
                dq = torch.from_numpy(dq)
            if dq.is_cuda and dist.get_backend(self.group) == "gloo":
                dq = dq.cpu()           # (a gloo group exchanges host tensors; "nccl" = RCCL takes the device rows)
            return all_gather_rows(dq, n_rows_total, self.group)
        return dq
