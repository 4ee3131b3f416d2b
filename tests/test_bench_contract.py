"""bench.py's one-line JSON contract and __graft_entry__.smoke(), on the GPU box (short runs)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_bench(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3", "--cpu-rows", "20000", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_bench_json_contract(mode):
    d = _run_bench(*(("--eager",) if mode == "eager" else ()))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "projections/s" and d["value"] > 1e7                      # BASELINE floor: >= 10 M proj/s
    assert abs(d["value"] - 1_000_000 * 30 / (d["ms_per_step"] * 30 * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.9 * 72e6 < r["traffic"] < 1.5 * 72e6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "projections/s" and "sample" in c
    assert abs(d["mean_angle_error_delta_vs_ref_deg"]) < 1e-4                      # the parity half of the metric
    # both clocks are reported: the host clock of the contract and the HIP events the roofline uses
    assert d["ms_per_step_events"] <= d["ms_per_step"] and abs(r["avg_launch_us"] - d["ms_per_step_events"] * 1e3) < 1e-6
    assert r["traffic"] is None or ("stored profile" in r["traffic_source"] and "profiles/r0" in r["traffic_source"])
    assert r["frac_events"] == r["frac"] and r["clock"].startswith("hip_events")
    # the fraction on the contract's clock beside the event one, and the per-launch spread (median, fastest of 20 eager launches)
    assert abs(r["frac_host_clock"] - 72.0 * 1_000_000 / (d["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-9 and r["frac_host_clock"] <= r["frac"]
    assert 0 < r["min_us"] <= r["median_us"] < 100.0
    c4 = d["secondary"]["config4_head_loss_backward_b512_bf16"]
    assert c4["us_per_step_empty_autograd_function_floor"] > 0 and "mirror_over_floor" not in c4
    mf = c4["mirror_minus_floor_us"]                     # mirror and floor in alternating blocks of one loop: the per-pair difference
    assert mf["pairs"] == 9 and mf["min"] <= mf["median"] <= mf["max"] and mf["iqr"] >= 0 and -20.0 < mf["median"] < 120.0
    # one rank: its device, its own clocks -- the same fields an 8-rank line carries per rank
    (seen,) = d["devices_seen"]
    assert seen["rank"] == 0 and seen["device_index"] == 0 and seen["arch"].startswith("gfx950") and seen["cus"] >= 64
    assert abs(seen["ms_per_step_events"] - d["ms_per_step_events"]) < 1e-6 * seen["ms_per_step_events"] and d["world_size_seen"] == 1      # (two reads of one pair of events)
    assert d["secondary"]["config1_head_b512_no_grad"]["us_per_call_host_clock"] > 0
    ov = d["secondary"]["config2_independent_batches_on_several_streams"]          # independent batches on 1 / 2 / 3 streams: throughput, labelled as such
    assert "error" not in ov and set(ov["us_per_launch_by_streams"]) == {"1", "2", "3"}
    assert 0 < ov["us_per_launch_by_streams"]["3"] <= 1.05 * ov["us_per_launch_by_streams"]["1"] and ov["frac_of_8TBps_by_streams"]["3"] < 1.0
    # the kernel is named by the library from the launch's own template arguments, not by a literal in bench.py
    assert r["kernel"].startswith("so3::k_rows<so3::OpProject<4, false>,") and r["kernel"].endswith(">")
    assert c4["mirror_path"] in ("cpp_node", "python")                               # which autograd node produced the mirror's figure
    assert c["cpu_model"] and c["cores"] <= c["host_cpu_count"]                      # every core the lease grants, and the CPU named
    if mode == "graph":
        pt = d["pre_timing"]
        assert pt["replays"] >= 1 and pt["ms"] <= 200.0
        assert pt["launches"] == 3 + 1 + 30 + pt["replays"] * 25                      # what ran untimed in front of the region, all of it


def test_bench_under_an_initialised_process_group_rccl_one_rank():
    """The N > 1 path of bench.py with one rank, in a fresh child process: RCCL comes up (backend "nccl" on the GPU),
    the K launches are captured into a hipGraph while the process group (and its watchdog thread) exists, and the
    (sum, count) pair goes through a real all-reduce.  Replaces 3D-Pose/main_DDP.py:39-60.  (Two and more ranks are the
    driver's to launch; host logic for world_size 2 is covered on CPU with gloo in tests/test_distributed_gloo.py.)"""
    env = dict(os.environ, SO3_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = _run_bench("--no-cpu-baseline", "--no-secondary", env=env)
    assert d["n_gpus"] == 1 and d["value"] > 1e7 and "hipGraph" in d["config"]["submission"]
    assert abs(d["mean_angle_error_delta_vs_ref_deg"]) < 1e-4                      # the all-reduced metric, vs the reference's number


def test_bench_with_two_ranks_sharing_the_device():
    """bench.py's N > 1 path end to end with TWO processes on the GPU box: both ranks on cuda:0 (RCCL refuses two ranks on one
    GPU, so the process group is gloo -- SO3_BENCH_BACKEND), each with its own buffers, its own hipGraph and its own timed
    region between the same two barriers; the clocks MAX-reduced, the (sum, count) pair all-reduced, ONE line from rank 0 with
    n_gpus = 2 and value = 2 x rows x steps / the slower rank's time.  (The ranks share the card, so the value is a bookkeeping
    check, not a throughput: what a real node gives is the driver's to measure.)"""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, SO3_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "3", "--rows", "500000"]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    lines0 = [l for l in outs[0][0].splitlines() if l.strip()]
    lines1 = [l for l in outs[1][0].splitlines() if l.strip()]
    assert len(lines0) == 1 and len(lines1) == 0                    # exactly one JSON line, from rank 0
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["scaling"] == "weak"
    assert d["config"]["rows_per_gpu"] == 500_000 and d["config"]["global_rows"] == 1_000_000 and d["config"]["parallelism"].startswith("dp2")
    assert abs(d["value"] - 1_000_000 * 30 / (d["ms_per_step"] * 30 * 1e-3)) / d["value"] < 1e-9
    assert d["ms_per_step_events"] <= d["ms_per_step"] * 1.001
    assert 120.0 < d["mean_angle_error_deg"] < 133.0               # the all-reduced metric over both ranks' rows (seeds 0 and 1)
    assert "cpu_baseline" not in d and set(d["secondary"]) == {"config5"}      # the CPU baseline and the other configs: rank 0 at N = 1 only


def test_bench_gpus_2_starts_its_own_ranks():
    """Exactly the driver's spelling -- `python bench.py --gpus 2 --steps 30 --warmup 3 --rows 500000` -- with NO rank variables in
    the environment: the entry is a launcher that starts two fresh ranks of itself before it touches HIP (bench.launch_ranks;
    3D-Pose/main_DDP.py:112-116 spawns its ranks the same way), relays rank 0's one line and returns 0.  On this one-GPU box both
    ranks are mapped onto cuda:0 (SO3_BENCH_SHARE_DEVICE) and the process group is gloo (RCCL refuses two ranks on one device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SO3_BENCH_BACKEND="gloo", SO3_BENCH_SHARE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "3", "--rows", "500000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["config"]["rows_per_gpu"] == 500_000 and d["config"]["global_rows"] == 1_000_000
    assert abs(d["value"] - 1_000_000 * 30 / (d["ms_per_step"] * 30 * 1e-3)) / d["value"] < 1e-9
    assert 120.0 < d["mean_angle_error_deg"] < 133.0
    # BASELINE configs[4] beside the headline, under this very command: twice the headline's rows per rank (2M at the default 1M), the same
    # skeleton, the evaluation's one all-reduce timed by itself, and the world size as the process group reports it
    c5 = d["secondary"]["config5"]
    assert c5["world_size_seen"] == 2 and c5["rows_per_gpu"] == 1_000_000 and c5["global_rows"] == 2_000_000 and c5["steps"] == 30
    assert abs(c5["value"] - c5["global_rows"] * 30 / (c5["ms_per_step"] * 30 * 1e-3)) / c5["value"] < 1e-9
    assert c5["ms_per_step_events"] <= c5["ms_per_step"] * 1.001 and c5["allreduce_us"] > 0 and c5["allreduce_backend"] == "gloo"
    assert c5["rows_counted_by_the_all_reduce"] == 2_000_000 and 120.0 < c5["mean_angle_error_deg"] < 133.0
    assert c5["hbm_bytes_resident_per_rank"] == 1_000_000 * 8 * 72 and c5["workload"].startswith("configs[4]")
    # what makes a first 8-GPU run readable after the fact: every rank's device, architecture, CU count and OWN clocks, the spread
    # over the ranks, the backend and the collective library's version
    assert [r["rank"] for r in d["devices_seen"]] == [0, 1] and all(r["device_index"] == 0 and r["arch"].startswith("gfx950") and r["cus"] >= 64
                                                                     for r in d["devices_seen"])
    sp = d["ms_per_step_events_by_rank"]
    assert 0 < sp["min"] <= sp["max"] and abs(sp["max"] - d["ms_per_step_events"]) < 1e-6 * sp["max"] and {sp["rank_of_min"], sp["rank_of_max"]} <= {0, 1}
    assert d["world_size_seen"] == 2 and d["allreduce_backend"] == "gloo" and d["collective_library"]["backend_in_use"] == "gloo" and "rccl_version" in d["collective_library"]
    assert [r["rank"] for r in c5["ranks"]] == [0, 1] and all(r["allreduce_us"] > 0 and r["ms_per_step_events"] > 0 for r in c5["ranks"])
    assert abs(c5["allreduce_us_by_rank"]["max"] - c5["allreduce_us"]) < 1e-6 * c5["allreduce_us"]


def test_a_rank_without_a_device_stops_the_job_with_one_line():
    """`python bench.py --gpus 2` on a box with ONE visible GPU (no SO3_BENCH_SHARE_DEVICE): rank 1's LOCAL_RANK names a device that does
    not exist.  It must say so in one line and exit non-zero BEFORE any process group exists, and the launcher must stop rank 0 and
    return non-zero promptly -- not after a rendezvous timeout, and without a JSON line on stdout."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SO3_BENCH_SHARE_DEVICE")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "3", "--rows", "100000"]
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and time.time() - t0 < 120.0
    assert "rank 1 of 2: LOCAL_RANK=1 but this process sees 1 HIP device(s)" in out.stderr, out.stderr[-1500:]
    assert not [l for l in out.stdout.splitlines() if l.strip()]


def test_bench_under_torch_distributed_run_as_the_driver_spells_it():
    """The driver's N > 1 command, word for word -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 --steps K --warmup W` -- on this one-GPU box: both ranks on cuda:0 (SO3_BENCH_SHARE_DEVICE) over gloo (RCCL
    refuses two ranks on one device).  The launcher's environment is authoritative (bench.py does not start ranks of its own under it), rank 0
    prints the one line, and the line carries both ranks."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SO3_BENCH_BACKEND="gloo", SO3_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--rows", "250000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2500:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["world_size_seen"] == 2
    assert [r["rank"] for r in d["devices_seen"]] == [0, 1] and d["config"]["global_rows"] == 500_000
    assert d["secondary"]["config5"]["world_size_seen"] == 2 and d["secondary"]["config5"]["rows_counted_by_the_all_reduce"] == 1_000_000


def test_config5_workload_string_at_the_configs_own_world():
    """At 8 ranks of 2M rows the leg's workload reads BASELINE.json configs[4] word for word (no GPU, no process group: the string only)."""
    import bench
    src = open(bench.__file__).read()
    assert "configs[4]: batch %dM sharded across %d MI355X (2M rows per GPU, seeds 0-%d), RCCL all-reduce of the mean angle error" in src
    assert ("configs[4]: batch %dM sharded across %d MI355X" % (2_000_000 * 8 // 1_000_000, 8)) == "configs[4]: batch 16M sharded across 8 MI355X"


def test_bench_config5_shape_on_one_rank():
    """--config 5 = BASELINE configs[4]: 2M rows per GPU (16M over 8), rank r seeded with r; one rank of it fits here."""
    d = _run_bench("--config", "5", "--no-cpu-baseline", "--no-secondary")
    assert d["config"]["rows_per_gpu"] == 2_000_000 and d["config"]["workload"].startswith("configs[4]")
    assert 120.0 < d["mean_angle_error_deg"] < 133.0                               # Haar-like pairs: 126.5 degrees


def test_smoke_entry_point():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
