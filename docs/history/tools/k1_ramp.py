#!/usr/bin/env python3
"""Clock ramp of K1: replay a hipGraph of 200 launches (1M rows, rotating buffers) back to back and print each
replay's time per launch, with rocm-smi power / sclk samples taken meanwhile.  usage: k1_ramp.py [replays] [idle_ms]"""
import ctypes, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
lib = _lib.load()
dev = torch.device("cuda:0")
rows, nb, per = 1_000_000, 8, 200
xs = [torch.randn(rows, 9, device=dev) for _ in range(nb)]
outs = [torch.empty(rows, 9, device=dev) for _ in range(nb)]
P = ctypes.c_void_p
s = torch.cuda.Stream()
samples, stop = [], False

def smi():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), out.strip().replace("\n", " | ")))
        except Exception as e:          # noqa: BLE001
            samples.append((time.time(), "smi failed: %r" % (e,)))
        time.sleep(0.2)

with torch.cuda.stream(s):
    st = P(s.cuda_stream)
    for i in range(10):
        lib.so3_project_fwd_f32(P(xs[i % nb].data_ptr()), P(outs[i % nb].data_ptr()), None, rows, st)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        for i in range(per):
            lib.so3_project_fwd_f32(P(xs[i % nb].data_ptr()), P(outs[i % nb].data_ptr()), None, rows, st)
    s.synchronize()
    th = threading.Thread(target=smi, daemon=True); th.start()
    time.sleep(1.0)                      # idle baseline samples
    t_begin = time.time()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for r in range(reps):
        ev[r][0].record(s); g.replay(); ev[r][1].record(s)
        if idle_ms: s.synchronize(); time.sleep(idle_ms * 1e-3)
    s.synchronize()
    t_end = time.time()
    # keep the device busy for 3 more seconds so that rocm-smi sees the loaded state
    t0 = time.time(); busy = []
    while time.time() - t0 < 3.0:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(20): g.replay()
        b.record(s); s.synchronize(); busy.append(a.elapsed_time(b) * 1e3 / (20 * per))
    stop = True
print("per-replay us/launch:", " ".join("%.2f" % (a.elapsed_time(b) * 1e3 / per) for a, b in ev))
print("busy phase us/launch (4000 launches each):", " ".join("%.2f" % v for v in busy))
print("timed replays ran from +%.2f s to +%.2f s" % (0.0, t_end - t_begin))
for t, line in samples:
    print("smi +%.2f s: %s" % (t - t_begin, line[:400]))
