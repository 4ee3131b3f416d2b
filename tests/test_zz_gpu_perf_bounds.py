"""Timing bounds on the GPU box -- kept OUT of the parity files and named so that it sorts last: under `pytest -x` a throttled
or busy device can then only turn THIS file red, after every parity test has been reached.  (The bounds are wall-time ratios
between launches of one process on one device; DESIGN.md section 4 "Worst case" has the measured table they come from.)"""
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def rr():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from poseestimation_amd import _lib
    _lib.load()                                   # fail loudly if the HIP extension is missing
    from poseestimation_amd import rotation_representation
    return rotation_representation


def test_hard_rows_cost_is_bounded(rr):
    """The worst case of K1 / K3 is on record and bounded (tools/k1_hard_rows.py, profiles/r03_k1_hard_rows.txt): a batch whose
    rows are HARD for the fast path (ties, near-reflections, rank deficiency: the packed Jacobi body runs on top of the fast
    path for every round dense in them) costs at most 2.2 x a Gaussian batch for K1 and 1.7 x for K3 (measured: 1.5-1.9 and
    1.3-1.45; round 2: 1.9-2.0 already at 1 % hard rows), and rows that are merely far from unit scale cost nothing extra (they
    were hard in round 2: 1.6 x)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("k1_hard_rows", os.path.join(ROOT, "tools", "k1_hard_rows.py"))
    hr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hr)
    from poseestimation_amd import _lib
    lib = _lib.load()
    n, nb = 1_000_000, 3
    gen = torch.Generator(device=DEV).manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream
    out = [torch.empty(n, 9, device=DEV) for _ in range(nb)]
    dm = [torch.empty(n, 9, device=DEV) for _ in range(nb)]
    ls = torch.empty(1, dtype=torch.float64, device=DEV)
    rt = hr.haar(n, torch.device(DEV), gen).reshape(n, 9).contiguous()

    def timed(fn):
        best = float("inf")
        for _ in range(3):
            for i in range(5):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
        return best

    def both(xs):
        k1 = timed(lambda i: lib.so3_project_fwd_f32(xs[i % nb].data_ptr(), out[i % nb].data_ptr(), None, n, st))
        k3 = timed(lambda i: lib.so3_frob_fwd_bwd_f32(xs[i % nb].data_ptr(), rt.data_ptr(), out[i % nb].data_ptr(), dm[i % nb].data_ptr(),
                                                      ls.data_ptr(), n, st))
        return k1, k3

    g1, g3 = both([torch.randn(n, 9, device=DEV, generator=gen) for _ in range(nb)])
    report = {}
    for name, cap1, cap3 in (("near-reflection", 2.2, 1.7), ("entries in {-1,0,1}", 2.2, 1.7), ("generic ties", 2.2, 1.7), ("rank one", 2.2, 1.7),
                             ("1e5 * Gaussian", 1.15, 1.15), ("rank two", 1.15, 1.15)):
        xs = [hr.family(name, n, torch.device(DEV), gen).reshape(n, 9).contiguous() for _ in range(nb)]
        k1, k3 = both(xs)
        report[name] = (round(k1 / g1, 2), round(k3 / g3, 2))
        assert k1 <= cap1 * g1 and k3 <= cap3 * g3, (name, k1, g1, k3, g3, report)
        del xs
    # a batch with SOME hard rows (10 %) stays near the review's 1.6 x for K1: the engine queues them and runs the Jacobi path once
    # per wave instead of once per round that holds one (measured 1.29-1.61 over three devices; 1.6-1.9 before the queue)
    for name in ("near-reflection", "entries in {-1,0,1}", "generic ties", "rank one"):
        xs = []
        for _ in range(nb):
            x = torch.randn(n, 9, device=DEV, generator=gen)
            idx = torch.nonzero(torch.rand(n, device=DEV, generator=gen) < 0.10).flatten()
            x[idx] = hr.family(name, idx.numel(), torch.device(DEV), gen).reshape(-1, 9)
            xs.append(x)
        k1 = timed(lambda i: lib.so3_project_fwd_f32(xs[i % nb].data_ptr(), out[i % nb].data_ptr(), None, n, st))
        report[name + " 10 %"] = round(k1 / g1, 2)
        assert k1 <= 1.75 * g1, (name, k1, g1, report)
        del xs
