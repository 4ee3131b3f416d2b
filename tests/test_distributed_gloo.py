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


# ---- bench.py's N > 1 path without a GPU: the timing / reduction skeleton under two gloo ranks -------------------------
def _bench_worker(rank, world, port, out_dir, config5):
    import json
    import sys
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        rows, steps, warmup = (2_000 if config5 else 1_000), 6, 2
        x = bench.first_buffer(rank, rows)                     # rank r <-> seed r (configs[1] and configs[4])
        calls = []

        def step(i):                                           # a stub step on CPU tensors: rank 1 is the slow one
            calls.append(i)
            time.sleep(0.002 * (1 + rank))

        config = {"workload": "stub", "rows_per_gpu": rows, "global_rows": rows * world}
        line, times = bench.run_skeleton(rank, world, steps, warmup, rows, step, lambda: None, dist, None, config,
                                         extra_times=(lambda: 100.0 + rank,))
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
            json.dump({"line": line, "times": times, "calls": len(calls), "x0": x[0].tolist(), "own": None}, fh)
        if rank == 0:                                          # what main() does with the line: exactly one JSON line, rank 0 only
            with open(os.path.join(out_dir, "stdout.txt"), "a") as fh:
                fh.write(json.dumps(line) + "\n")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("config5", [False, True])
def test_bench_skeleton_with_two_ranks(tmp_path, config5):
    """bench.py's timed region, MAX-reduce of the clocks and headline (bench.run_skeleton: the code main() runs around its
    launches) with world_size 2 on CPU: value = global rows x steps / the SLOWEST rank's wall time, n_gpus = world, one
    line from rank 0, seeds by rank."""
    import json
    world = 2
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), config5), nprocs=world, join=True)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    rows, steps, warmup = (2_000 if config5 else 1_000), 6, 2
    assert res[1]["line"] is None and res[0]["line"] is not None
    lines = open(tmp_path / "stdout.txt").read().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == res[0]["line"]
    line = res[0]["line"]
    assert line["n_gpus"] == world and line["steps"] == steps and line["warmup"] == warmup and line["scaling"] == "weak"
    assert res[0]["calls"] == res[1]["calls"] == steps + warmup
    # both ranks hold the same job times: the MAX over ranks of the wall clock and of the extra (event) time
    assert res[0]["times"] == res[1]["times"] and res[0]["times"][1] == 101.0
    wall = res[0]["times"][0]
    assert wall >= steps * 0.004                               # the slow rank's 4 ms per step, not rank 0's 2 ms
    assert wall < steps * 0.004 * 3
    assert abs(line["value"] - rows * world * steps / wall) <= 1e-9 * line["value"]
    assert abs(line["ms_per_step"] - wall * 1e3 / steps) < 1e-12
    for r in range(world):                                     # rank r's first buffer is randn under seed r
        g = torch.Generator().manual_seed(r)
        assert res[r]["x0"] == torch.randn(rows, 9, generator=g)[0].tolist()


# ---- what rank 0 reports about every rank (bench.gather_rank_reports): one all-reduce of a (world, K) matrix ------------------
def _report_worker(rank, world, port, out_dir):
    import json
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        row = bench.rank_report_row(3 + rank, "gfx950:sramecc+:xnack-" if rank == 0 else "gfx942", 256 - 64 * rank, [0.015 + 0.001 * rank, 0.0149 + 0.002 * rank])
        reports = bench.gather_rank_reports(row, rank, world, dist, None, names=("ms_per_step", "ms_per_step_events"))
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
            json.dump({"reports": reports, "spread": bench.spread(reports, "ms_per_step_events"), "backend": dist.get_backend(),
                       "world": dist.get_world_size()}, fh)
    finally:
        dist.destroy_process_group()


def test_every_rank_is_reported_to_rank_0(tmp_path):
    """bench.py's `devices_seen`: per rank the device index, architecture, CU count and the rank's OWN clock readings, identical on
    every rank after ONE all-reduce; the spread names the slow rank.  (What a first 8-GPU run leaves behind to be read.)"""
    import json
    world = 2
    mp.spawn(_report_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert res[0] == res[1]
    r0, r1 = res[0]["reports"]
    assert r0 == {"rank": 0, "device_index": 3, "arch": "gfx950:sramecc+:xnack-", "cus": 256, "ms_per_step": 0.015, "ms_per_step_events": 0.0149}
    assert r1["rank"] == 1 and r1["device_index"] == 4 and r1["arch"] == "gfx942" and r1["cus"] == 192
    assert abs(r1["ms_per_step_events"] - 0.0169) < 1e-15
    assert res[0]["spread"] == {"min": 0.0149, "max": r1["ms_per_step_events"], "rank_of_min": 0, "rank_of_max": 1}
    assert res[0]["backend"] == "gloo" and res[0]["world"] == 2


def test_preflight_and_single_rank_report():
    import bench
    assert bench.preflight_device(0, 1) is None and bench.preflight_device(7, 8) is None
    why = bench.preflight_device(1, 1, "0")
    assert why.startswith("LOCAL_RANK=1 but this process sees 1 HIP device(s)") and "HIP_VISIBLE_DEVICES" in why and "\n" not in why
    assert "sees 0 HIP device(s)" in bench.preflight_device(0, 0)
    (only,) = bench.gather_rank_reports(bench.rank_report_row(0, "gfx950", 256, [1.5]), 0, 1, None, None, names=("ms_per_step",))
    assert only == {"rank": 0, "device_index": 0, "arch": "gfx950", "cus": 256, "ms_per_step": 1.5}
    assert len(bench.rank_report_row(0, "x" * 100, 1, [])) == 2 + bench.ARCH_BYTES          # long names are cut, the row keeps its width
