"""world_size-2 gloo run (CPU) of the multi-GPU host logic: contiguous sharding + ONE all-reduce of
the (sum, count) pair.  The device kernel is replaced by the oracle here -- as the checker of the
HOST logic only; the GPU kernels themselves are tested in test_gpu_parity.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import so3_oracle as so
        from poseestimation_amd.distributed import global_mean_angle_error, shard_range
        rng = np.random.default_rng(123)                        # every rank builds the same full batch
        x = rng.standard_normal((total, 9)).astype(np.float32)
        t = so.symmetric_orthogonalization_np(rng.standard_normal((total, 9))).astype(np.float32)
        lo, hi = shard_range(total, rank, world)

        def local(r1, r2):                                       # stands in for K4's fused (sum, count)
            deg = so.angle_error_np(r1.numpy(), r2.numpy()) if len(r1) else np.zeros(0)
            return torch.tensor([deg.sum(), float(len(deg))], dtype=torch.float64)

        r_shard = torch.from_numpy(so.symmetric_orthogonalization_np(x[lo:hi]).astype(np.float32))
        mean = global_mean_angle_error(r_shard, torch.from_numpy(t[lo:hi]), local_sum_count=local)
        full = so.angle_error_np(so.symmetric_orthogonalization_np(x).astype(np.float32), t).mean()
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([mean.item(), full, lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [1001, 1])
def test_two_rank_mean_angle_allreduce(tmp_path, total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rank{r}.npy") for r in range(world)]
    assert res[0][0] == res[1][0]                               # identical on every rank
    assert abs(res[0][0] - res[0][1]) < 1e-9                    # equals the unsharded mean
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == total
