#!/usr/bin/env python3
"""A/B of K1 / K2 / K3 / K1+K4 between builds of the library on ONE device (devices differ by 10 %): builds are interleaved, each
kernel timed as the best of AB_ROUNDS x 100 eager launches over rotating buffers.  Builds of either ABI (tools/_abi.py).
usage: ab_v2.py a.so b.so ...        env: AB_ROUNDS (4), AB_ONLY=K1,K2,...  AB_HARD=1 adds the hard-row families of k1_hard_rows.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _abi import Lib, P

dev = torch.device("cuda:0")
n, NB = 1_000_000, 6
x = [torch.randn(n, 9, device=dev) for _ in range(NB)]
g = [torch.randn(n, 9, device=dev) for _ in range(NB)]
r = [torch.empty(n, 9, device=dev) for _ in range(NB)]
dm = [torch.empty(n, 9, device=dev) for _ in range(NB)]
ls = torch.empty(1, dtype=torch.float64, device=dev)
lm = torch.empty((), device=dev)
st = P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr())
libs = [Lib(path) for path in sys.argv[1:]]
ws = torch.zeros(32768, dtype=torch.uint8, device=dev)
pool = torch.zeros(4096, 4, dtype=torch.float64, device=dev)
slot = lambda i: (P(pool[i % 4096].data_ptr()), P(pool[i % 4096].data_ptr() + 16))
rt = [torch.empty(n, 9, device=dev) for _ in range(2)]
for t_ in rt:
    libs[0].k1(p(torch.randn(n, 9, device=dev)), p(t_), n, st)


def timed(fn, iters=100):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


kernels = {
    "K1": lambda L: (lambda i: L.k1(p(x[i % NB]), p(r[i % NB]), n, st)),
    "K2": lambda L: (lambda i: L.k2(p(x[i % NB]), p(g[i % NB]), p(dm[i % NB]), n, st)),
    "K3 R+dM ws": lambda L: (lambda i: L.k3(p(x[i % NB]), p(rt[0]), p(r[i % NB]), p(dm[i % NB]), p(ls), p(lm), p(ws), n, st)),
    "K3 dM ws": lambda L: (lambda i: L.k3(p(x[i % NB]), p(rt[0]), None, p(dm[i % NB]), p(ls), p(lm), p(ws), n, st)),
    "K3 R+dM at": lambda L: (lambda i: L.k3(p(x[i % NB]), p(rt[0]), p(r[i % NB]), p(dm[i % NB]), p(ls), None, None, n, st)),
    "K3' ws": lambda L: (lambda i: L.k3p(p(x[i % NB]), p(g[i % NB]), p(dm[i % NB]), p(ls), p(lm), p(ws), n, st)),
    "K4 sum": lambda L: (lambda i: L.k4_sum(p(rt[i % 2]), p(rt[1 - i % 2]), slot(i)[0], slot(i)[1], n, st)),
    "K1+K4 sum": lambda L: (lambda i: L.k14_sum(p(x[i % NB]), p(rt[0]), slot(i)[0], slot(i)[1], n, st)),
    "K1+K4 f64": lambda L: (lambda i: L.k14_sum(p(x[i % NB]), p(rt[0]), slot(i)[0], slot(i)[1], n, st, exact=True)),
}
only = os.environ.get("AB_ONLY")
best = {(k, L.name): 1e9 for k in kernels for L in libs}
for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
    for k, mk in kernels.items():
        if only and k not in only.split(","):
            continue
        for L in libs:
            best[(k, L.name)] = min(best[(k, L.name)], timed(mk(L)))
for k in kernels:
    if best[(k, libs[0].name)] > 1e8:
        continue
    print("%-12s " % k + "   ".join("%s %6.2f" % (L.name, best[(k, L.name)]) for L in libs), flush=True)

if os.environ.get("AB_HARD") == "1":
    import importlib.util
    spec = importlib.util.spec_from_file_location("k1_hard_rows", os.path.join(os.path.dirname(os.path.abspath(__file__)), "k1_hard_rows.py"))
    hr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hr)
    gen = torch.Generator(device=dev).manual_seed(11)
    del g
    print("--- hard rows: us per 1M rows (x the same build's Gaussian time), K1 | K3 R+dM ---")
    base = {}
    for L in libs:
        base[L.name] = (min(timed(kernels["K1"](L)) for _ in range(3)), min(timed(kernels["K3 R+dM at"](L)) for _ in range(3)))
    print("%-34s " % "Gaussian" + "   ".join("%s %5.2f | %5.2f" % (L.name, *base[L.name]) for L in libs), flush=True)
    for fam in ("rank two", "near-reflection", "entries in {-1,0,1}", "generic ties", "rank one", "all zero", "1e5 * Gaussian"):
        for share in (0.01, 0.10, 1.0):
            xs = []
            for _ in range(3):
                xx = torch.randn(n, 9, device=dev, generator=gen)
                idx = torch.nonzero(torch.rand(n, device=dev, generator=gen) < share).flatten() if share < 1.0 else torch.arange(n, device=dev)
                xx[idx] = hr.family(fam, idx.numel(), dev, gen).reshape(-1, 9)
                xs.append(xx)
            line = "%-26s %5.0f %% " % (fam, share * 100)
            for L in libs:
                k1 = min(timed(lambda i: L.k1(p(xs[i % 3]), p(r[i % NB]), n, st), 60) for _ in range(2))
                k3 = min(timed(lambda i: L.k3(p(xs[i % 3]), p(rt[0]), p(r[i % NB]), p(dm[i % NB]), p(ls), None, None, n, st), 60) for _ in range(2))
                line += "   %s %5.2f (%.2fx) | %5.2f (%.2fx)" % (L.name, k1, k1 / base[L.name][0], k3, k3 / base[L.name][1])
            print(line, flush=True)
            del xs
