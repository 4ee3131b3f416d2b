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


# What is CLAIMED for hard rows (x the same build's Gaussian batch; README and DESIGN.md section 4 quote this table, which is the worst of
# the driver's round-5 box (GPUTEST_r05: ties at 1 % 1.33, near-reflections 1.22, rank one 1.23) and of round 6's own runs): reported in
# every run's tail.  What is ASSERTED: 1.25 x the claim for the hard families, and a cap of their own for the families that are NOT hard
# (zero rows, rows far from unit scale, rank two: 0.95-1.19 on record) -- a prescale branch that became twice as slow, or a hard-row path
# 40 % slower, turns the test red; only the band between a claim and its cap is an expected failure (a device that clocks K1 low).
CLAIMED = {  # share of hard rows: (K1, K3)
    0.01: (1.35, 1.2), 0.10: (1.6, 1.35), 1.0: (1.8, 1.5)}
CLAIMED_TIES_ALL = 2.05          # K1 on a whole batch of generic ties: both algorithms on every row
CLAIMED_EASY = (1.25, 1.2)       # zero rows (forward), rows far from unit scale (every round that holds one takes the prescale branch), rank two: not hard
CAP_EASY = (1.5, 1.5)            # asserted for those
CAP_OVER_CLAIM = 1.25            # asserted for the hard families: this times the claim


def test_hard_rows_cost_is_bounded(rr):
    """Rows that are HARD for the quaternion fast path (ties, near-reflections, rank deficiency) are parked -- inputs and row number, in a
    list the workgroup shares in LDS -- when they are few in their round, and redone one matrix per lane when the workgroup has
    streamed its share; a round dense in them runs the Jacobi path on the spot.  Ratios between launches of one process on one device
    (tools/ab_v2.py AB_HARD=1 is the same measurement across builds).  One JSON line with every ratio goes to the run's record
    (conftest.REPORT_LINES, and gpurun_out/hard_rows_table.json when that directory exists)."""
    import importlib.util
    import json
    import conftest
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
        k3 = timed(lambda i: lib.so3_frob_fwd_bwd_v2_f32(xs[i % nb].data_ptr(), rt.data_ptr(), out[i % nb].data_ptr(), dm[i % nb].data_ptr(),
                                                         ls.data_ptr(), None, None, 0, n, st))
        return k1, k3

    g1, g3 = both([torch.randn(n, 9, device=DEV, generator=gen) for _ in range(nb)])
    table, over_claim, over_gross = {}, [], []
    hard = ("near-reflection", "entries in {-1,0,1}", "generic ties", "rank one")
    for name in hard + ("all zero", "1e5 * Gaussian", "rank two"):
        for share in (0.01, 0.10, 1.0):
            xs = []
            for _ in range(nb):
                if share < 1.0:
                    x = torch.randn(n, 9, device=DEV, generator=gen)
                    idx = torch.nonzero(torch.rand(n, device=DEV, generator=gen) < share).flatten()
                    x[idx] = hr.family(name, idx.numel(), torch.device(DEV), gen).reshape(-1, 9)
                else:
                    x = hr.family(name, n, torch.device(DEV), gen).reshape(n, 9).contiguous()
                xs.append(x)
            k1, k3 = both(xs)
            del xs
            claim1, claim3 = CLAIMED[share] if name in hard else CLAIMED_EASY
            if name == "all zero":
                claim3 = CLAIMED[share][1]         # the BACKWARD of a zero row still goes through the Jacobi frames' floored denominators: hard for K3
            if name == "generic ties" and share == 1.0:
                claim1 = CLAIMED_TIES_ALL
            key = "%s %g %%" % (name, share * 100)
            table[key] = [round(k1 / g1, 2), round(k3 / g3, 2)]
            if k1 > claim1 * g1 or k3 > claim3 * g3:
                over_claim.append((key, table[key], (claim1, claim3)))
            cap1, cap3 = (CAP_OVER_CLAIM * claim1, CAP_OVER_CLAIM * claim3) if (name in hard or name == "all zero") else CAP_EASY
            if k1 > cap1 * g1 or k3 > cap3 * g3:
                over_gross.append((key, table[key], (round(cap1, 2), round(cap3, 2))))
    line = json.dumps({"hard_rows_x_gaussian_K1_K3": table, "gaussian_us_K1_K3": [round(g1, 2), round(g3, 2)],
                       "over_claimed_bar": [k for k, _, _ in over_claim]})
    conftest.REPORT_LINES.append(line)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        open(os.path.join(ROOT, "gpurun_out", "hard_rows_table.json"), "w").write(line + "\n")
    assert not over_gross, over_gross
    if over_claim:
        pytest.xfail("claimed bars exceeded on this device: %r" % (over_claim,))
