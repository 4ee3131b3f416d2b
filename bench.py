#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SVD -> SO(3) hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N > 1: starts its N ranks itself, one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic Gaussian (rows, 9) float32 -> (rows, 3, 3) rotations,
rows = 1,000,000 per GPU.  A "step" is one K1 launch over one resident batch.  Inputs and outputs
rotate over 8 buffer pairs (576 MB per GPU) so the 256 MiB Infinity Cache cannot serve replays:
the reported number is an HBM number.  With N > 1 every rank owns its own rows (weak scaling, no
data-path collective); after the timed region the mean geodesic angle error is reduced on the
device per rank (K4) and summed with ONE all-reduce (RCCL), and rank 0's line gains
secondary.config5 = BASELINE configs[4] (2M rows per rank: 16M over 8), timed the same way.

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job projections/s.
Extra objects: "roofline" (HBM roofline of k_project_fwd from HIP events on the launch stream) and
"cpu_baseline" (the oracle's torch port -- the reference's ATen call sequence -- timed on this box's
host cores; N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import signal
import socket
import subprocess
import sys
import time


# ---- `python bench.py --gpus N` without an outer launcher ----------------------------------------------------------------
# The reference's only multi-process script starts its ranks itself (3D-Pose/main_DDP.py:112-116, mp.spawn(world_size=2)); so does
# this one.  With N > 1 and no WORLD_SIZE in the environment the process that was started is only a LAUNCHER: it starts N fresh
# children of this very script, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what
# torch.distributed.run would have set), relays rank 0's one JSON line, and exits non-zero as soon as any child does (ending the
# others).  It runs BEFORE torch is even imported: the launcher makes no HIP call of any kind, and children are started with
# subprocess (fork + exec of a process that never touched the GPU), never by replacing a process that did.
def _child_dies_with_parent():
    """preexec hook: SIGKILL to the child when the launcher dies (prctl(PR_SET_PDEATHSIG)), so that not even a SIGKILLed
    launcher leaves ranks behind."""
    try:
        import ctypes as ct
        ct.CDLL(None, use_errno=True).prctl(1, signal.SIGKILL, 0, 0, 0)
    except Exception:
        pass


def launch_ranks(n: int, argv, child_cmd=None, grace_s: float = 5.0, out=None) -> int:
    """Start n ranks of `child_cmd + argv` (default: this script under the same interpreter), wait for them, relay rank 0's
    stdout to `out` (default: this process' stdout) when all have ended well.  Returns the exit code: 0, or the first failing
    child's (negative signals as 128 + signal); on a failure -- or on SIGTERM / SIGINT to the launcher -- the remaining children
    get SIGTERM and, `grace_s` later, SIGKILL.  SO3_BENCH_SHARE_DEVICE=1 gives every child LOCAL_RANK 0 (the one-GPU test box)."""
    cmd = list(child_cmd) if child_cmd is not None else [sys.executable, os.path.abspath(__file__)]
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    share = os.environ.get("SO3_BENCH_SHARE_DEVICE") == "1"
    procs = []

    def end_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    got_signal = []

    def on_signal(signum, _frame):          # no exception out of the handler: the loop below sees the flag and takes the one
        got_signal.append(signum)           # shutdown path (SIGTERM to the ranks, grace_s for them to leave their process group, SIGKILL)

    old = {s: signal.signal(s, on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if share else str(r), WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SO3_BENCH_LAUNCHED="1")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # rank 0's stdout is the job's one line; the other ranks print nothing there (whatever a library writes goes to stderr)
            procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                          stderr=None, preexec_fn=_child_dies_with_parent))
        line = b""
        import select
        fd0 = procs[0].stdout
        while True:
            if fd0 is not None:                                  # drain rank 0's pipe while waiting (never block on a full pipe)
                ready, _, _ = select.select([fd0], [], [], 0.05)
                if ready:
                    chunk = os.read(fd0.fileno(), 65536)
                    if chunk:
                        line += chunk
                    else:
                        fd0 = None
            else:
                time.sleep(0.05)
            if got_signal:
                rc = 128 + got_signal[0]
                break
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0] if bad[0] > 0 else 128 - bad[0]
                break
            if all(c == 0 for c in codes) and fd0 is None:
                break
        if rc != 0:
            end_all(signal.SIGTERM)
            deadline = time.time() + grace_s
            while time.time() < deadline and any(p.poll() is None for p in procs):
                time.sleep(0.05)
            end_all(signal.SIGKILL)
            for p in procs:
                p.wait()
            print("[bench] %s; the ranks were stopped" % ("signal %d" % got_signal[0] if got_signal else "a rank ended with exit code %d" % rc), file=sys.stderr)
        else:
            dst = out if out is not None else sys.stdout
            dst.write(line.decode())
            dst.flush()
    finally:
        end_all(signal.SIGKILL)
        for s, h in old.items():
            signal.signal(s, h)
    return rc


def _self_launch_if_needed(argv) -> None:
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args(argv)
    if known.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(known.gpus, argv))


if __name__ == "__main__":
    _self_launch_if_needed(sys.argv[1:])          # before torch is imported: the launcher never touches HIP

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS_DEFAULT = 1_000_000
NBUF = 8
BYTES_PER_PROJECTION = 72          # 36 B read + 36 B written
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (default: 1M, config #2; 2M with --config 5)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="2: BASELINE configs[1], 1M rows per GPU (the headline).  5: configs[4], 16M rows sharded over 8 GPUs = 2M rows per GPU, "
                         "rank r generated with seed r, one RCCL all-reduce of (sum, count) for the mean angle error")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the brief timing of configs #3 and #4")
    ap.add_argument("--eager", action="store_true", help="launch every step from Python instead of replaying one hipGraph of K launches")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the CPU baseline sample")
    return ap.parse_args()


def host_cores() -> int:
    """CPU share of this process: min(os.cpu_count(), the affinity mask, the cgroup quota) -- every core the lease grants, as
    BASELINE.md section 3 asks (round 3 capped this at 16).  SO3_CPU_BASELINE_THREADS overrides it."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    forced = os.environ.get("SO3_CPU_BASELINE_THREADS")
    return max(1, int(forced)) if forced else max(1, n)


def cpu_baseline(rows: int):
    """The oracle's torch port (== the reference's ATen sequence) on the host cores, bounded sample."""
    from oracle import so3_oracle as so
    cores = host_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, 9, generator=g)
    so.symmetric_orthogonalization_torch(x[: max(1, rows // 50)])            # LAPACK warm-up
    best = float("inf")
    t_start = time.perf_counter()
    reps = 0
    while reps < 3 and (time.perf_counter() - t_start) < 25.0:
        t0 = time.perf_counter()
        so.symmetric_orthogonalization_torch(x)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    cpu_name = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": rows / best, "unit": "projections/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_name,
            "host_cpu_count": os.cpu_count(),
            "sample": f"{rows} rows of the same Gaussian workload, torch.linalg.svd-based restatement of "
                      f"rotation_representation.py:199-205 (oracle/so3_oracle.py), best of {reps}, {cpu_name}"}


# ---- the timing / reduction skeleton (no GPU in it: tests/test_distributed_gloo.py runs it with two gloo ranks and a stub step)
METRIC = "3x3 SVD->SO(3) projections/sec @ batch 1M"


def first_buffer(rank: int, rows: int) -> torch.Tensor:
    """Buffer 0 of rank r is config #2 / #5's generator: torch.manual_seed(r); randn(rows, 9) on the CPU."""
    g = torch.Generator().manual_seed(rank)
    return torch.randn(rows, 9, generator=g)


def timed_region(run, sync, barrier=None, now=time.perf_counter) -> float:
    """The contract's timed region on this rank: barrier + synchronize | clock | run() | synchronize | clock | barrier.
    Returns this rank's seconds.  The ranks start together (the opening barrier), so the job's time is the MAX over ranks
    (max_over_ranks below); the closing barrier's own latency is not a step and stays outside the clock."""
    if barrier is not None:
        barrier()
    sync()
    t0 = now()
    run()
    sync()
    wall = now() - t0
    if barrier is not None:
        barrier()
    return wall


def max_over_ranks(values, dist, device=None):
    """Element-wise MAX of a few per-rank floats over the job (one all-reduce); the values themselves without a group."""
    if dist is None or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()


# ---- what rank 0 says about the OTHER ranks (no GPU in it: tests/test_distributed_gloo.py drives it with two gloo ranks) ------------
# The driver's 8-GPU run is the first time more than one rank meets real devices and nobody watches it live: the line therefore carries,
# per rank, the device it bound to, that device's architecture and CU count, and its own clock readings -- a rank that landed on the wrong
# device, a partitioned (CPX / DPX) device or a slow one shows in the record.  One all-reduce of a (world, K) float64 matrix in which every
# rank fills its own row (the same machinery as the metric's (sum, count) pair; strings travel as bytes).
ARCH_BYTES = 24


def preflight_device(local_rank: int, device_count: int, visible=None):
    """None when LOCAL_RANK names a device this process can see, else the one-line reason the rank must stop with (before any
    process group exists: a rank that dies inside init_process_group leaves the others waiting for the rendezvous' timeout)."""
    if 0 <= local_rank < device_count:
        return None
    return ("LOCAL_RANK=%d but this process sees %d HIP device(s)%s: start at most one rank per visible GPU"
            % (local_rank, device_count, "" if not visible else " (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = %s)" % visible))


def rank_report_row(device_index: int, arch: str, cus: int, numbers) -> list:
    """One rank's row of the report matrix: device index, CU count, its own clock readings, the architecture name as bytes."""
    raw = arch.encode("ascii", "replace")[:ARCH_BYTES]
    return [float(device_index), float(cus)] + [float(v) for v in numbers] + [float(b) for b in raw] + [0.0] * (ARCH_BYTES - len(raw))


def gather_rank_reports(row, rank: int, world: int, dist, device=None, names=()) -> list:
    """Every rank's row on every rank (ONE SUM all-reduce; the rows themselves without a group) as dicts:
    {"rank", "device_index", "arch", "cus", <names...>}."""
    k = len(row)
    m = torch.zeros((world, k), dtype=torch.float64, device=device)
    m[rank] = torch.tensor(row, dtype=torch.float64, device=device)
    if dist is not None and world > 1:
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
    out = []
    for r, vals in enumerate(m.tolist()):
        nums = vals[2:k - ARCH_BYTES]
        arch = bytes(int(b) for b in vals[k - ARCH_BYTES:] if int(b) != 0).decode("ascii", "replace")
        d = {"rank": r, "device_index": int(vals[0]), "arch": arch, "cus": int(vals[1])}
        d.update({n: v for n, v in zip(names, nums)})
        out.append(d)
    return out


def spread(reports, key) -> dict:
    """min / max over the ranks of one reported number, and which ranks hold them: a slow rank shows."""
    vals = [r[key] for r in reports]
    lo, hi = min(vals), max(vals)
    return {"min": lo, "max": hi, "rank_of_min": vals.index(lo), "rank_of_max": vals.index(hi)}


def collective_library_version():
    """RCCL's version as torch reports it ("2.21.5"), or None (a CPU build, the gloo test path)."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        return None


def headline(rows_per_gpu: int, world: int, steps: int, warmup: int, wall_s: float, config: dict) -> dict:
    """The contract's fields: value = rows of ALL ranks x steps / the slowest rank's time (whole-job throughput, weak scaling)."""
    total_rows = rows_per_gpu * world
    return {"metric": METRIC, "value": total_rows * steps / wall_s, "unit": "projections/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": wall_s * 1e3 / steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config}


def run_skeleton(rank: int, world: int, steps: int, warmup: int, rows: int, step, sync, dist, device, config: dict, extra_times=(),
                 run=None, do_warmup=True, own=None):
    """Warm-up, the timed region, the MAX over ranks, the headline -- what main() does around the launches, callable with a
    stub `step(i)` on CPU tensors.  Returns (line or None, job times): rank 0 gets the dict it will print, the others None.
    extra_times: callables evaluated after the region on every rank (e.g. the HIP-event time), MAX-reduced with the wall time.
    own: a dict that receives this rank's own wall time ("wall_s") before the reduction (main() reports every rank's).
    run: what the region executes instead of `steps` calls of step (main(): one replay of the hipGraph that holds them)."""
    if do_warmup:
        for i in range(warmup):
            step(i)
    if run is None:
        def run():
            for i in range(steps):
                step(i)

    barrier = dist.barrier if dist is not None else None
    wall = timed_region(run, sync, barrier)
    if own is not None:
        own["wall_s"] = wall                    # this rank's own clock (the job's is the MAX over ranks below)
    times = max_over_ranks([wall] + [f() for f in extra_times], dist, device)
    line = headline(rows, world, steps, warmup, times[0], config) if rank == 0 else None
    return line, times


def secondary_configs(lib, dev):
    """The other BASELINE.json configs, timed briefly on rank 0 at N = 1 (reported beside the headline)."""
    P = ctypes.c_void_p
    st = P(torch.cuda.current_stream().cuda_stream)
    out = {}

    def timed(fn, iters, warm=3):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3          # us per call

    # config #3: 65 536 clouds x 1024 points, fused cross-covariance + projection (K5); two buffer sets (3.2 GB)
    b, n = 65536, 1024
    pc = [torch.rand(b, n, 3, device=dev) - 0.5 for _ in range(2)]
    qc = [torch.rand(b, n, 3, device=dev) - 0.5 for _ in range(2)]
    rk = torch.empty(b, 9, device=dev)
    us = timed(lambda i: lib.so3_kabsch_f32(P(pc[i % 2].data_ptr()), P(qc[i % 2].data_ptr()), P(rk.data_ptr()), None, b, n, st), 8, 2)
    bytes_ = b * (2 * n * 12 + 36)
    out["config3_kabsch_65536x1024"] = {"us_per_call": us, "clouds_per_s": b / us * 1e6, "achieved_GBps": bytes_ / us * 1e-3,
                                        "frac_of_8TBps": bytes_ / us * 1e-3 / HBM_PEAK_GBS, "bytes_per_cloud_algorithmic": 2 * n * 12 + 36}
    del pc, qc
    # config #4: B = 512, bf16 storage, head + Frobenius loss + backward in ONE C-ABI call (K3)
    b = 512
    x4 = torch.randn(b, 9, device=dev).bfloat16()
    t4 = torch.empty(b, 9, device=dev)
    lib.so3_project_fwd_f32(P(torch.randn(b, 9, device=dev).data_ptr()), P(t4.data_ptr()), None, b, st)
    r4 = torch.empty(b, 9, device=dev)
    d4 = torch.empty(b, 9, device=dev, dtype=torch.bfloat16)
    ls = torch.empty(1, dtype=torch.float64, device=dev)
    us = timed(lambda i: lib.so3_frob_fwd_bwd_v2_bf16(P(x4.data_ptr()), P(t4.data_ptr()), P(r4.data_ptr()), P(d4.data_ptr()), P(ls.data_ptr()), None, None, 0, b, st), 300, 10)
    out["config4_head_loss_backward_b512_bf16"] = {"us_per_fused_call": us, "note": "launch-latency-bound (9 KB); ONE launch (the one-workgroup kernel writes the loss itself)"}
    # the same step as a user calls it: through the Python mirror with autograd, and as a recorded hipGraph step
    from poseestimation_amd import rotation_representation as rr
    xg = x4.clone().requires_grad_(True)
    def mirror(_):
        loss, _r = rr.frobenius_head(xg, t4.view(b, 3, 3))
        loss.backward()
        xg.grad = None
    # torch's own floor in THIS process: an autograd.Function that launches nothing (it returns an empty scalar, its backward a
    # stored buffer) stepped the same way -- what any custom node costs here before it does any work of its own
    class _Floor(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, buf):
            ctx.buf = buf
            return x.new_empty(())

        @staticmethod
        def backward(ctx, g):
            return ctx.buf, None

    buf = torch.zeros_like(xg)

    def floor(_):
        _Floor.apply(xg, buf).backward()
        xg.grad = None

    def block(fn, n=300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    for i in range(200):                      # the autograd engine's device thread and the allocator's small pool settle over ~100 calls
        mirror(i)
        floor(i)
    # Mirror and floor in ALTERNATING 300-step blocks of one loop.  Either one alone is bimodal with where the autograd engine's device
    # thread is scheduled (~26 or ~66 us for the floor; docs/history/tools.tar.gz:tools/mirror_bisect.py), and measured in two separate loops the two
    # landed in their modes independently: rounds 4-5's ratio said which mode each had drawn.  Neighbouring blocks share the mode, so the
    # per-pair DIFFERENCE is what the mirror adds to an empty autograd.Function: median and interquartile range over the pairs.
    mblocks, fblocks, diffs = [], [], []
    for _ in range(9):
        m_us, f_us = block(mirror), block(floor)
        mblocks.append(m_us); fblocks.append(f_us); diffs.append(m_us - f_us)
    q = lambda v, f: float(np.quantile(np.asarray(v), f))
    # which autograd node served those steps: csrc/autograd_node.cpp's (no interpreter inside forward / backward) or the Python class
    probe, _ = rr.frobenius_head(xg, t4.view(b, 3, 3))
    c4 = out["config4_head_loss_backward_b512_bf16"]
    c4["mirror_path"] = "cpp_node" if "FrobeniusHeadNode" in probe.grad_fn.name() else "python"
    c4["us_per_step_python_mirror_autograd"] = q(mblocks, 0.5)
    c4["us_per_step_python_mirror_autograd_best_block"] = min(mblocks)
    c4["us_per_step_empty_autograd_function_floor"] = q(fblocks, 0.5)
    c4["mirror_minus_floor_us"] = {"median": q(diffs, 0.5), "iqr": q(diffs, 0.75) - q(diffs, 0.25), "min": min(diffs), "max": max(diffs),
                                   "pairs": len(diffs), "note": "alternating 300-step blocks of the mirror's step and of an autograd.Function that launches nothing; per-pair difference"}
    # the head alone, no autograd (evaluation loops), B = 512 float32, by the host clock
    x512 = torch.randn(b, 9, device=dev)
    for i in range(100):
        rr.symmetric_orthogonalization(x512)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2000):
        rr.symmetric_orthogonalization(x512)
    torch.cuda.synchronize()
    out["config1_head_b512_no_grad"] = {"us_per_call_host_clock": (time.perf_counter() - t0) / 2000 * 1e6,
                                        "note": "symmetric_orthogonalization(x) under no autograd, 2000 calls back to back, one synchronize at the end"}
    if os.environ.get("SO3_BENCH_PROFILE_MIRROR") == "1":
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for i in range(300):
            mirror(i)
        torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(16)
    step = rr.FrobeniusHeadStep(b, dtype=torch.bfloat16, device=dev)
    step.x.copy_(x4)
    step.r_true.copy_(t4.view(b, 3, 3))
    for i in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(500):
        step()
    torch.cuda.synchronize()
    out["config4_head_loss_backward_b512_bf16"]["us_per_step_recorded_graph"] = (time.perf_counter() - t0) / 500 * 1e6
    # config #2 again with ONE buffer pair (72 MB: resident in the 256 MiB Infinity Cache) -- labelled, never the headline
    xr = torch.randn(ROWS_DEFAULT, 9, device=dev)
    rr_ = torch.empty(ROWS_DEFAULT, 9, device=dev)
    us = timed(lambda i: lib.so3_project_fwd_f32(P(xr.data_ptr()), P(rr_.data_ptr()), None, ROWS_DEFAULT, st), 200, 20)
    out["config2_cache_resident_single_buffer_pair"] = {"us_per_call": us, "note": "same 36 MB in / 36 MB out replayed: served by the Infinity Cache, not HBM; eager launches"}
    del xr, rr_
    torch.cuda.empty_cache()
    # Independent batches on SEVERAL streams (a serving loop's shape, not a training step's): a launch of the persistent engine leaves wave
    # slots idle while its first loads are in flight and while the waves that hold one round more than the others finish (46 % of a 1M-row
    # launch's waves hold two rounds, the rest three); launches on other streams fill them.  One hipGraph of 240 launches over 8 rotating
    # buffer pairs per shape -- a chain on one stream, and forked over two and three -- replayed in turn; microseconds per launch = event time
    # / launches, which with more than one stream is a THROUGHPUT figure (a profiler's per-dispatch duration is longer: launches overlap).
    # The headline above stays the one-stream chain: its time per step IS the kernel's duration, which is what `roofline` prices.
    nl = 240
    xs_ = [torch.randn(ROWS_DEFAULT, 9, device=dev) for _ in range(NBUF)]
    rs_ = [torch.empty(ROWS_DEFAULT, 9, device=dev) for _ in range(NBUF)]

    def chain(nstreams):
        lanes = [torch.cuda.Stream() for _ in range(nstreams)]
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.stream(lanes[0]):
            with torch.cuda.graph(g, stream=lanes[0], capture_error_mode="thread_local"):
                for s_ in lanes[1:]:
                    s_.wait_stream(lanes[0])
                for i in range(nl):
                    if lib.so3_project_fwd_f32(P(xs_[i % NBUF].data_ptr()), P(rs_[i % NBUF].data_ptr()), None, ROWS_DEFAULT, P(lanes[i % nstreams].cuda_stream)) != 0:
                        raise RuntimeError("so3_project_fwd_f32 failed: %s" % lib.so3_last_error().decode())
                for s_ in lanes[1:]:
                    lanes[0].wait_stream(s_)
        return g, lanes[0]

    try:
        shapes = {k: chain(k) for k in (1, 2, 3)}
        best = {k: float("inf") for k in shapes}
        for _ in range(4):
            for k, (g, s0) in shapes.items():
                with torch.cuda.stream(s0):
                    g.replay()
                    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(s0)
                    g.replay(); g.replay()
                    b_.record(s0)
                torch.cuda.synchronize()
                best[k] = min(best[k], a.elapsed_time(b_) * 1e3 / (2 * nl))
        out["config2_independent_batches_on_several_streams"] = {
            "us_per_launch_by_streams": {str(k): v for k, v in best.items()},
            "frac_of_8TBps_by_streams": {str(k): BYTES_PER_PROJECTION * ROWS_DEFAULT / (v * 1e-6) / 1e9 / HBM_PEAK_GBS for k, v in best.items()},
            "projections_per_s_3_streams": ROWS_DEFAULT / (best[3] * 1e-6),
            "note": "throughput of a 240-launch hipGraph over 8 rotating buffer pairs, chained on one stream or forked over two / three; launches on "
                    "different streams overlap (the tail of one with the fill of the next), so the per-launch figure is event time / launches, not a kernel duration"}
        del shapes
    except Exception as exc:                   # a capture that the runtime refuses must not cost the headline its line
        out["config2_independent_batches_on_several_streams"] = {"error": repr(exc)}
    del xs_, rs_
    torch.cuda.empty_cache()
    return out


def config5_leg(lib, rr, dist, dev, rank: int, world: int, steps: int, warmup: int, rows: int = 2_000_000):
    """BASELINE.json configs[4] under the command the driver runs (`bench.py --gpus N`): the same timed skeleton at 2M rows per rank
    (16M over 8 ranks), rank r's first buffer from seed r, eight rotated buffer pairs (2M rows x 8 pairs x 72 B = 1.15 GB per rank);
    then the evaluation the config names -- every rank reduces its shard to (sum of angles, count) in ONE fused launch (K1+K4) and
    ONE all-reduce sums the 16-byte pair -- with that collective's own latency from events around 20 calls.  Collective: every rank
    calls it; rank 0 gets the dict.  (3D-Pose/main_DDP.py:39-42,112-116 is the reference's multi-process set-up; its reduce_loss,
    :56-60, is never called.)"""
    xs = [first_buffer(rank, rows).to(dev)]
    gen = torch.Generator(device=dev).manual_seed(5000 + rank)
    for _ in range(NBUF - 1):
        xs.append(torch.randn(rows, 9, device=dev, generator=gen))
    outs = [torch.empty(rows, 3, 3, device=dev) for _ in range(NBUF)]
    stream = torch.cuda.current_stream()
    st = ctypes.c_void_p(stream.cuda_stream)
    calls = [(ctypes.c_void_p(xs[i].data_ptr()), ctypes.c_void_p(outs[i].data_ptr())) for i in range(NBUF)]
    brows = ctypes.c_int64(rows)

    def step(i):
        a, b = calls[i % NBUF]
        if lib.so3_project_fwd_f32(a, b, None, brows, st) != 0:
            raise RuntimeError("so3_project_fwd_f32 failed: %s" % lib.so3_last_error().decode())

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def region():
        e0.record(stream)
        for i in range(steps):
            step(i)
        e1.record(stream)

    config = {"workload": "configs[4]: batch %dM sharded across %d MI355X (2M rows per GPU, seeds 0-%d), RCCL all-reduce of the mean angle error"
                          % (rows * world // 1_000_000, world, world - 1) if rows == 2_000_000 else "configs[4] at %d rows per GPU" % rows,
              "rows_per_gpu": rows, "global_rows": rows * world, "buffer_pairs_rotated": NBUF, "submission": "eager launches"}
    for i in range(max(warmup, NBUF)):
        step(i)
    e0.record(stream); e1.record(stream)
    own = {}
    line, times = run_skeleton(rank, world, steps, warmup, rows, step, torch.cuda.synchronize, dist, dev, config,
                               extra_times=(lambda: e0.elapsed_time(e1),), run=region, do_warmup=False, own=own)
    own_ev_ms = e0.elapsed_time(e1) / steps
    # the evaluation: one fused launch per rank, one all-reduce of (sum, count)
    gt = torch.Generator().manual_seed(1 + 1000 * rank)
    t_rot = rr.symmetric_orthogonalization(torch.randn(rows, 9, generator=gt).to(dev))
    sc = rr.head_angle_error(xs[0], t_rot, reduce="sum_count", check=False)
    pair = sc.clone()
    dist.all_reduce(pair, op=dist.ReduceOp.SUM)
    mean_angle = (pair[0] / pair[1]).item()
    count_seen = float(pair[1].item())
    probe = sc.clone()
    for _ in range(3):
        dist.all_reduce(probe, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    dist.barrier()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a0.record()
    for _ in range(20):
        dist.all_reduce(probe, op=dist.ReduceOp.SUM)
    a1.record()
    torch.cuda.synchronize()
    host_us = (time.perf_counter() - t0) / 20 * 1e6
    own_ar_us = a0.elapsed_time(a1) * 1e3 / 20
    ar = max_over_ranks([own_ar_us, host_us], dist, dev)
    props = torch.cuda.get_device_properties(dev)
    reports = gather_rank_reports(rank_report_row(dev.index, getattr(props, "gcnArchName", props.name), props.multi_processor_count,
                                                  [own["wall_s"] * 1e3 / steps, own_ev_ms, own_ar_us]),
                                  rank, world, dist, dev, names=("ms_per_step", "ms_per_step_events", "allreduce_us"))
    del xs, outs
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    return {"workload": config["workload"], "rows_per_gpu": rows, "global_rows": rows * world, "world_size_seen": dist.get_world_size(),
            "steps": steps, "ms_per_step": line["ms_per_step"], "ms_per_step_events": times[1] / steps, "value": line["value"], "unit": "projections/s",
            "frac_of_8TBps_per_gpu_events": BYTES_PER_PROJECTION * rows / (times[1] * 1e-3 / steps) / 1e9 / HBM_PEAK_GBS,
            "mean_angle_error_deg": mean_angle, "rows_counted_by_the_all_reduce": count_seen,
            "ms_per_step_events_by_rank": spread(reports, "ms_per_step_events"), "allreduce_us_by_rank": spread(reports, "allreduce_us"),
            "ranks": reports,
            "allreduce_us": ar[0], "allreduce_us_host_clock": ar[1], "allreduce_backend": dist.get_backend(),
            "allreduce_note": "one SUM all-reduce of a 16-byte (sum, count) device tensor; events on the launch stream around 20 calls, MAX over ranks",
            "hbm_bytes_resident_per_rank": rows * NBUF * BYTES_PER_PROJECTION}


def main():
    args = parse()
    # Exactly ONE line goes to stdout.  Native libraries write there too (RCCL prints a version banner when it comes up),
    # so file descriptor 1 is pointed at stderr for the whole run and the JSON line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before anything initialises HIP (dmabuf IPC only on this pool)
    if args.rows is None:
        args.rows = 2_000_000 if args.config == 5 else ROWS_DEFAULT
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SO3_BENCH_SHARE_DEVICE") == "1":      # the one-GPU test box: every rank on cuda:0, whichever launcher set LOCAL_RANK
        local_rank = 0
    if world != args.gpus:                  # under an outer launcher the environment is authoritative (python bench.py --gpus N with
        args.gpus = world                   # no WORLD_SIZE never gets here: _self_launch_if_needed started N ranks of its own)
    # device_count() does not initialise HIP: the check runs before anything can fail in a less readable way (set_device on a device
    # that does not exist; a rank that dies inside init_process_group while the others wait for the rendezvous)
    ndev = torch.cuda.device_count()
    if ndev == 0:
        sys.exit("[bench] rank %d of %d: bench.py needs an MI355X and this process sees no HIP device (no CPU fallback for the product path)" % (rank, world))
    why = preflight_device(local_rank, ndev, os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"))
    if why is not None:
        sys.exit("[bench] rank %d of %d: %s" % (rank, world, why))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("SO3_BENCH_FORCE_DIST") == "1":    # the env knob exercises the RCCL path with one rank
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("SO3_BENCH_BACKEND", "nccl")      # "gloo": two ranks on ONE device (RCCL refuses a shared GPU): the test of this path
        import datetime
        limit = datetime.timedelta(seconds=float(os.environ.get("SO3_BENCH_DIST_TIMEOUT_S", "300")))    # (default: 10 minutes of a dead job)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=limit)

    if world > 1:
        # N ranks on one host: each draws its first buffer with torch's CPU generator (config #2 / #5's recipe), and N processes with a thread per
        # hardware thread each would queue on each other for it
        torch.set_num_threads(max(1, host_cores() // world))
    from poseestimation_amd import _lib
    from poseestimation_amd import rotation_representation as rr
    from poseestimation_amd.distributed import allreduce_sum_count
    lib = _lib.load()                      # raises if the HIP extension is missing

    rows = args.rows
    # buffer 0 of rank r is config #2/#5's generator: torch.manual_seed(r); randn(rows, 9) on the CPU
    xs = [first_buffer(rank, rows).to(dev)]
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    for _ in range(NBUF - 1):
        xs.append(torch.randn(rows, 9, device=dev, generator=gen))
    outs = [torch.empty(rows, 3, 3, device=dev) for _ in range(NBUF)]
    stream = torch.cuda.current_stream()
    st = ctypes.c_void_p(stream.cuda_stream)
    calls = [(ctypes.c_void_p(xs[i].data_ptr()), ctypes.c_void_p(outs[i].data_ptr())) for i in range(NBUF)]
    fwd = lib.so3_project_fwd_f32
    brows = ctypes.c_int64(rows)

    def step(i):
        a, b = calls[i % NBUF]
        rc = fwd(a, b, None, brows, st)          # `st` is rebound to the capture stream while the graph is recorded
        if rc != 0:
            raise RuntimeError("so3_project_fwd_f32 failed: %s" % lib.so3_last_error().decode())

    for i in range(args.warmup):
        step(i)
    step(0)
    k1_kernel = lib.so3_last_kernel().decode()      # the instantiation the timed steps launch, as rocprofv3 --kernel-trace names it
    # The K timed steps are K kernel launches either way; by default they are captured once into a hipGraph
    # (the C ABI is enqueue-only, hence capturable) and replayed, so the 16-us kernels are not at the mercy
    # of Python's per-launch jitter.  --eager launches each step from the interpreter instead.
    graph = None
    pre = None
    if not args.eager:
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                # thread_local: the RCCL watchdog thread of a multi-rank job may query events while we capture
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    st = ctypes.c_void_p(side.cuda_stream)
                    for i in range(args.steps):
                        step(i)
            stream = side
            torch.cuda.synchronize()
            # Untimed replays: graph upload / first-touch effects, and the shader clock.  From idle the chip needs ~20 ms of
            # load before its clock is up (docs/history/tools.tar.gz:tools/k1_ramp.py: 20 us per launch falling to 16 over the first 20-30 ms), so a
            # fixed number of warm-up steps would time the ramp, not the kernel.  The timed graph is replayed once (upload,
            # first touch); then a SHORT graph of the same launches (<= 25 steps, so the ramp is sampled every ~0.4 ms) is
            # replayed until its time has stopped falling: the mean of the last 32 replays (10 ms) no longer beats the mean of
            # the 32 before it by 0.3 %, after at least 50 ms and at most 120 ms of load (the ramp is a staircase: one device
            # sat on a 16.2-us step at 25 ms and reached 14.9 us by 50 ms, docs/history/tools.tar.gz:tools/k1_ramp_fine.py).  (`sustained` below is the same graph after
            # 0.6 s: devices differ in whether that is faster -- clock still rising -- or slower -- power limit reached.)
            with torch.cuda.stream(side):
                graph.replay()
            torch.cuda.synchronize()
            nshort = min(args.steps, 25)
            warm = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(warm, stream=side, capture_error_mode="thread_local"):
                    for i in range(nshort):
                        step(i)
            torch.cuda.synchronize()
            pre = {"replays": 0, "ms": 0.0, "us_per_step_first": None, "us_per_step_last": None,
                   "policy": "a %d-step graph replayed until the mean of the last 32 replays no longer beats the mean of the 32 before "
                             "by 0.3 percent (50-120 ms of load)" % nshort}
            t_pre = time.perf_counter()
            hist = []
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            while True:
                with torch.cuda.stream(side):              # a graph replays on the CURRENT stream; eight replays per host
                    marks[0].record(side)                  # synchronisation keep the device loaded (one sync per replay left
                    for j in range(8):                     # it idle a tenth of the time)
                        warm.replay()
                        marks[j + 1].record(side)
                torch.cuda.synchronize()
                hist.extend(marks[j].elapsed_time(marks[j + 1]) for j in range(8))
                spent = time.perf_counter() - t_pre
                flat = len(hist) >= 64 and sum(hist[-32:]) >= 0.997 * sum(hist[-64:-32])
                if (flat and spent >= 0.050) or spent >= 0.120:
                    break
            pre["replays"] = len(hist)
            # every launch of the benchmark kernel that ran before the timed region: the W warm-up steps, one more for the kernel's
            # name, the timed graph once (upload, first touch), and the short graph's replays ("warmup" in the line is W alone)
            pre["launches"] = args.warmup + 1 + args.steps + len(hist) * nshort
            pre["us_per_step_first"] = hist[0] * 1e3 / nshort
            pre["us_per_step_last"] = sum(hist[-32:]) / len(hist[-32:]) * 1e3 / nshort
            pre["ms"] = (time.perf_counter() - t_pre) * 1e3
        except Exception as exc:               # submission mode only: the same kernels are then launched eagerly
            print(f"[bench] hipGraph capture failed ({exc!r}); falling back to eager launches", file=sys.stderr)
            graph = None
            args.eager = True
            stream = torch.cuda.current_stream()
            st = ctypes.c_void_p(stream.cuda_stream)
            torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    config = {"workload": ("configs[4]: batch 16M sharded across 8 MI355X (2M rows per GPU, seeds 0-7), RCCL all-reduce of the mean angle error"
                           if args.config == 5 else "configs[1]: batch 1M synthetic 3x3 Gaussian -> SO(3) projection, fp32, per GPU"),
              "rows_per_gpu": rows, "global_rows": rows * world, "buffer_pairs_rotated": NBUF,
              "parallelism": f"dp{world} (row shards, one all-reduce of (sum,count) for the metric)",
              "submission": "eager launches" if args.eager else "one hipGraph of K kernel launches, replayed"}

    def region():                             # the K steps, bracketed by HIP events on the launch stream
        e0.record(stream)
        if graph is not None:
            graph.replay()
        else:
            for i in range(args.steps):
                step(i)
        e1.record(stream)

    def synchronize():                        # the contract's synchronize, reached without a sleep: the host polls the closing event
        while not e1.query():                 # first (a blocked hipDeviceSynchronize wakes up 5-15 us after the device is done, which
            pass                              # is 2-5 % of a 20-step region and belongs to the scheduler, not to the kernel)
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):           # a graph replays on the CURRENT stream: entered before the clock starts
        e0.record(stream); e1.record(stream)  # (torch creates an event's handle at its first record: not inside the region either)
        own_clock = {}
        out, (wall, ev_ms) = run_skeleton(rank, world, args.steps, args.warmup, rows, step, synchronize, dist, dev, config,
                                          extra_times=(lambda: e0.elapsed_time(e1),), run=region, do_warmup=False, own=own_clock)
        # per-launch spread (median and min): a second, untimed pass of 20 eager launches with an event
        # between every two of them
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        st = ctypes.c_void_p(stream.cuda_stream)
        marks[0].record(stream)
        for i in range(20):
            step(i)
            marks[i + 1].record(stream)
        torch.cuda.synchronize()
        per_launch_us = sorted(marks[i].elapsed_time(marks[i + 1]) * 1e3 for i in range(20))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # every rank's device and its OWN clocks (wall / ev_ms above are the job's: MAX over ranks), gathered with one all-reduce
    props = torch.cuda.get_device_properties(dev)
    own_wall_ms, own_ev_ms = own_clock["wall_s"] * 1e3 / args.steps, e0.elapsed_time(e1) / args.steps
    reports = gather_rank_reports(rank_report_row(dev.index, getattr(props, "gcnArchName", props.name), props.multi_processor_count,
                                                  [own_wall_ms, own_ev_ms]),
                                  rank, world, dist, dev, names=("ms_per_step", "ms_per_step_events"))

    # parity metric: mean geodesic angle vs the decoy target (config #2), one all-reduce of (sum, count)
    gt = torch.Generator().manual_seed(1 + 1000 * rank)
    t_rot = rr.symmetric_orthogonalization(torch.randn(rows, 9, generator=gt).to(dev))
    step(0)
    sc = rr.angle_error_sum_count(outs[0], t_rot, check=False)
    allreduce_sum_count(sc)
    mean_angle = (sc[0] / sc[1]).item()
    delta = None
    golden = os.path.join(ROOT, "tests", "golden", "g6_stats_1m.npz")
    if rank == 0 and world == 1 and rows == ROWS_DEFAULT and os.path.exists(golden):
        delta = mean_angle - float(np.load(golden)["mean_angle_deg"])      # vs the reference's own number

    if rank == 0:
        per_launch_s = ev_ms * 1e-3 / args.steps
        achieved = BYTES_PER_PROJECTION * rows / per_launch_s / 1e9
        achieved_host = BYTES_PER_PROJECTION * rows / (wall / args.steps) / 1e9
        traffic, traffic_from = None, None
        pmc = os.path.join(ROOT, "profiles", "k1_pmc_traffic.json")
        if rows == ROWS_DEFAULT and os.path.exists(pmc):      # the stored figure belongs to the default row count only
            try:
                stored = json.load(open(pmc))
                traffic, traffic_from = stored.get("hbm_bytes_per_launch"), stored.get("source")
            except (OSError, ValueError):
                traffic = None
        out.update({
            "ms_per_step_events": ev_ms / args.steps,             # HIP events on the launch stream (ms_per_step: the host clock of the contract)
            "value_events": rows * world * args.steps / (ev_ms * 1e-3),
            "mean_angle_error_deg": mean_angle,
            "mean_angle_error_delta_vs_ref_deg": delta,
            "roofline": {"bound": "hbm", "kernel": k1_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # `achieved` / `frac` are the contract's: algorithmic bytes / the kernel's average launch duration by HIP events on the
                         # launch stream over the timed region (the figure a rocprofv3 trace of this command agrees with); under its own name too
                         "frac_events": achieved / HBM_PEAK_GBS, "clock": "hip_events_on_the_launch_stream",
                         # the same fraction on the contract's clock (barrier + synchronize around the K steps: ~10 us of submission
                         # and synchronisation latency are inside it, 3 % of a 20-step region)
                         "frac_host_clock": achieved_host / HBM_PEAK_GBS,
                         # one launch at a time (20 eager launches, an event between every two): median and fastest
                         "median_us": 0.5 * (per_launch_us[9] + per_launch_us[10]), "min_us": per_launch_us[0],
                         "traffic": traffic,
                         "traffic_source": ("stored profile profiles/k1_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                            "this workload, FETCH_SIZE doubled per the gfx950 caveat; written by tools/collect_profiles.sh from: %s); "
                                            "not measured in this run" % traffic_from) if traffic is not None else None,
                         "bytes_per_launch_algorithmic": BYTES_PER_PROJECTION * rows,
                         "avg_launch_us": per_launch_s * 1e6},
            "pre_timing": pre,
            # one entry per rank: the device it bound to and its own clocks (ms_per_step / ms_per_step_events above: MAX over ranks)
            "devices_seen": reports,
            "ms_per_step_events_by_rank": spread(reports, "ms_per_step_events"),
            "world_size_seen": dist.get_world_size() if dist is not None else 1,
            "allreduce_backend": dist.get_backend() if dist is not None else None,
            # (backend "nccl" IS RCCL on ROCm; the version is the library torch was built against, whichever backend this job uses)
            "collective_library": {"backend_in_use": dist.get_backend(), "rccl_version": collective_library_version()} if dist is not None else None,
        })
        if world == 1 and not args.no_secondary and graph is not None:
            # the same graph held for ~0.6 s: what the kernel sustains once the package sits at its power limit
            t_hold = time.perf_counter()
            h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nrep = max(1, 4000 // args.steps)
            with torch.cuda.stream(stream):                 # a graph replays on the CURRENT stream
                while time.perf_counter() - t_hold < 0.6:
                    for _ in range(max(1, 2000 // args.steps)):
                        graph.replay()
                    torch.cuda.synchronize()
                h0.record(stream)
                for _ in range(nrep):
                    graph.replay()
                h1.record(stream)
            torch.cuda.synchronize()
            sus_us = h0.elapsed_time(h1) * 1e3 / (nrep * args.steps)
            out["sustained"] = {"us_per_step": sus_us, "frac_of_8TBps": BYTES_PER_PROJECTION * rows / (sus_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "note": "same graph after 0.6 s of continuous replay (events over %d launches)" % (nrep * args.steps)}
        if world == 1 and not args.no_secondary:
            del xs, outs
            torch.cuda.empty_cache()
            out["secondary"] = secondary_configs(lib, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_rows)
    # N > 1: BASELINE configs[4] beside the headline (value stays config #2's weak scaling, so that N = 1 agrees with the 1-GPU record)
    c5 = None
    if dist is not None and world > 1 and not args.no_secondary and args.config == 2:
        c5 = config5_leg(lib, rr, dist, dev, rank, world, args.steps, args.warmup,
                         rows=2_000_000 if args.rows == ROWS_DEFAULT else 2 * args.rows)
    if rank == 0:
        if c5 is not None:
            out.setdefault("secondary", {})["config5"] = c5
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
