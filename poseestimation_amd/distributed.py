"""Multi-GPU layer of the hot path: contiguous row-range sharding + ONE all-reduce of (sum, count).

The path shards trivially: every 3x3 block is independent, so rank g owns
rows [g*B/W, (g+1)*B/W) and runs K1/K4 on them with no data-path collective.  The only exchange
is the scalar metric: each rank reduces its shard to a (sum of angles, row count) pair on the
device (K4's fused reduction) and one all-reduce (RCCL over xGMI when the backend is "nccl")
sums the 16-byte pair.  The reference never does this: DataParallel gathers every output to GPU 0
and runs the head there (3D-Pose/main.py:58-60,154); its DDP variant leaves `reduce_loss`
(3D-Pose/main_DDP.py:56-60) uncalled.

Host logic here is backend-agnostic and is covered on CPU with gloo, world_size 2
(tests/test_distributed_gloo.py); the device kernels are injected by the caller.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Rows [lo, hi) owned by `rank`: contiguous, balanced to within one row, covering [0, total)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    if total_rows < 0:
        raise ValueError("total_rows must be >= 0")
    base, rem = divmod(total_rows, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allreduce_sum_count(sum_count: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """In-place SUM all-reduce of the (sum, count) pair; a no-op without an initialised group."""
    if sum_count.numel() != 2:
        raise ValueError("expected a 2-element (sum, count) tensor")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sum_count, op=dist.ReduceOp.SUM, group=group)
    return sum_count


def global_mean_angle_error(
    r_pred_shard: torch.Tensor,
    r_true_shard: torch.Tensor,
    group: Optional[dist.ProcessGroup] = None,
    local_sum_count: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
) -> torch.Tensor:
    """Mean geodesic angle (degrees, float64, 0-dim) over ALL ranks' rows.

    `local_sum_count(r1, r2) -> tensor([sum_deg, count], float64)` defaults to the K4 kernel with
    its fused device-side reduction; the result is identical on every rank.  Empty shards are
    allowed (count 0); an all-empty job returns NaN.
    """
    if local_sum_count is None:
        from .rotation_representation import angle_error_sum_count
        local_sum_count = angle_error_sum_count
    sc = local_sum_count(r_pred_shard, r_true_shard)
    if sc.dtype != torch.float64:
        sc = sc.double()
    sc = allreduce_sum_count(sc, group)
    return sc[0] / sc[1]


def project_shard(x_full_or_shard: torch.Tensor, total_rows: Optional[int] = None,
                  group: Optional[dist.ProcessGroup] = None,
                  project: Optional[Callable[[torch.Tensor], torch.Tensor]] = None) -> torch.Tensor:
    """Project this rank's rows.  If `total_rows` is given, `x_full_or_shard` is the full (B,9)
    batch and the rank's row range is sliced out of it; otherwise it already is the shard."""
    if project is None:
        from .rotation_representation import symmetric_orthogonalization
        project = symmetric_orthogonalization
    x = x_full_or_shard
    if total_rows is not None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        lo, hi = shard_range(total_rows, rank, world)
        x = x.reshape(-1, 9)[lo:hi]
    return project(x)
