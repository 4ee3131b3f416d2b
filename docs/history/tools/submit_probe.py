#!/usr/bin/env python3
"""How the K timed launches of bench.py are best submitted when K is small (the driver runs K = 20): one hipGraph replay
with the events outside, the same graph with event-record nodes inside it, or plain stream launches.  Prints wall and
event microseconds per step for each, interleaved, after the shader clock is up.  usage: submit_probe.py [K] [repeats]"""
import ctypes, os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from poseestimation_amd import _lib

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 15
SPIN = len(sys.argv) > 3 and sys.argv[3] == "spin"
lib = _lib.load()
dev = torch.device("cuda", 0)
rows, NBUF = 1_000_000, 8
xs = [torch.randn(rows, 9, device=dev) for _ in range(NBUF)]
outs = [torch.empty(rows, 3, 3, device=dev) for _ in range(NBUF)]
side = torch.cuda.Stream()
st = ctypes.c_void_p(side.cuda_stream)
calls = [(ctypes.c_void_p(xs[i].data_ptr()), ctypes.c_void_p(outs[i].data_ptr())) for i in range(NBUF)]
fwd, brows = lib.so3_project_fwd_f32, ctypes.c_int64(rows)


def step(i):
    a, b = calls[i % NBUF]
    assert fwd(a, b, None, brows, st) == 0


def capture(with_events):
    g = torch.cuda.CUDAGraph()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            if with_events:
                ev[0].record(side)
            for i in range(K):
                step(i)
            if with_events:
                ev[1].record(side)
    return g, ev


torch.cuda.synchronize()
g_plain, _ = capture(False)
try:
    g_ev, ev_in = capture(True)
except Exception as exc:
    print("in-graph events: capture failed:", repr(exc)); g_ev = None
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    e0.record(side); e1.record(side)
torch.cuda.synchronize()


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        if mode == "graph":
            e0.record(side); g_plain.replay(); e1.record(side)
        elif mode == "graph+nodes":
            g_ev.replay()
        else:
            e0.record(side)
            for i in range(K):
                step(i)
            e1.record(side)
    if SPIN and mode != "graph+nodes":
        while not e1.query():
            pass
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6 / K
    try:
        ev = (ev_in[0].elapsed_time(ev_in[1]) if mode == "graph+nodes" else e0.elapsed_time(e1)) * 1e3 / K
    except Exception as exc:
        ev = float("nan"); print(mode, "elapsed_time failed:", repr(exc))
    return wall, ev


t = time.perf_counter()
with torch.cuda.stream(side):
    while time.perf_counter() - t < 0.05:           # clock up
        g_plain.replay()
torch.cuda.synchronize()
modes = ["graph", "eager"]
res = {m: [] for m in modes}
for r in range(REP):
    for m in modes:
        with torch.cuda.stream(side):
            for _ in range(3):
                g_plain.replay()                    # keep the clock where it is between samples
        res[m].append(run(m))
for m in modes:
    w = [x[0] for x in res[m]]; e = [x[1] for x in res[m]]
    print("%-12s K=%d  wall/step median %.2f min %.2f   events/step median %.2f min %.2f" % (m, K, statistics.median(w), min(w), statistics.median(e), min(e)))
with torch.cuda.stream(side):
    e0.record(side)
    for _ in range(50):
        g_plain.replay()
    e1.record(side)
torch.cuda.synchronize()
print("50 back-to-back replays of the %d-step graph: %.2f us/step" % (K, e0.elapsed_time(e1) * 1e3 / (50 * K)))
