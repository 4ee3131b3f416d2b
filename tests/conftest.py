import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Tables a test wants in the run's record whatever its outcome (tests/test_zz_gpu_perf_bounds.py: measured ratios): one line each, written
# with the terminal summary, i.e. at the tail of `pytest -q`'s output.
REPORT_LINES = []


def pytest_terminal_summary(terminalreporter):
    for line in REPORT_LINES:
        terminalreporter.write_line(line)


@pytest.fixture(autouse=True)
def _seeded_rng():
    """Every test starts from the same RNG state: thresholds on random data must not depend on which tests ran before."""
    import torch
    torch.manual_seed(20240)
    yield


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def built_library():
    """Path of libso3proj.so.  On the build container hipcc (re)builds it; on the GPU box the prebuilt
    in-tree .so is used as shipped."""
    from poseestimation_amd import build
    try:
        return build.build_library()
    except RuntimeError:
        if os.path.exists(build.LIB):
            return build.LIB
        raise


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as co
    co.build()
    return co


def well_conditioned(s, det, rel_gap=1e-2):
    """Rows where R is well determined in float32: sigma1>0 and the relevant gap
    (s2+s3 without flip, s2-s3 with flip) is not small relative to s1."""
    s = np.asarray(s, np.float64)
    gap = np.where(np.asarray(det) < 0, s[:, 1] - s[:, 2], s[:, 1] + s[:, 2])
    return gap > rel_gap * np.maximum(s[:, 0], 1e-300)


def orth_err(r):
    r = np.asarray(r, np.float64).reshape(-1, 3, 3)
    return np.linalg.norm(np.einsum("bji,bjk->bik", r, r) - np.eye(3), axis=(1, 2))


# ---- G15: the metrics' gradients (tools/gen_golden.py g15_metric_gradients) -------------------------------------------
# name -> (eps of the clamp, unit of theta, divisor rule, upstream kind) of the six cases G15 holds per set and dtype
METRIC_GRAD_CASES = {
    "geo_mean": (1e-7, 1.0, "B", "one"),
    "geo_sum": (1e-7, 1.0, "1", "one"),
    "geo_none": (1e-7, 1.0, "1", "w"),
    "cgd": (0.0, 1.0, "1", "w"),
    "ang": (0.0, 180.0 / np.pi, "1", "w"),
    "ang_mean": (0.0, 180.0 / np.pi, "B", "one"),
}


def metric_grad_check(got1, got2, ref1, ref2, c, eps, u_c, base, label=""):
    """A gradient pair (dR1, dR2) against the reference's autograd (G15), row by row, at the tolerance the expression's conditioning
    allows: the slope -1/sqrt(1 - c^2) moves by c/(1 - c^2) (relative) per unit of c, and the arithmetic that produced the
    reference's c (float32 graph: u_c ~ a few 1e-7; float64: 1e-15) is not ours operation for operation; `base` = relative
    round-off of everything else.  Rows whose c is within u_c of the clamp's bounds may legitimately sit on either side of the mask
    (gradient 0 on one side): there either answer is accepted.  Returns the number of rows that were ambiguous."""
    got1, got2, ref1, ref2 = (np.asarray(v, np.float64).reshape(-1, 9) for v in (got1, got2, ref1, ref2))
    c = np.asarray(c, np.float64)
    lo, hi = -1.0 + eps, 1.0 - eps
    om = np.maximum(1.0 - c * c, 1e-300)
    rel = u_c * np.abs(c) / om + base
    edge = (np.abs(c - lo) <= 2 * u_c) | (np.abs(c - hi) <= 2 * u_c) | (om <= 4 * u_c)
    for got, ref in ((got1, ref1), (got2, ref2)):
        scale = np.abs(ref).max(axis=1)
        err = np.abs(got - ref).max(axis=1)
        ok = err <= rel * scale + 1e-30
        zero_side = (np.abs(got).max(axis=1) == 0) | (scale == 0)
        bad = ~(ok | (edge & zero_side) | (edge & (err <= 0.75 * np.maximum(scale, np.abs(got).max(axis=1)))))
        assert not bad.any(), (label, np.nonzero(bad)[0][:8], err[bad][:4], (rel * scale)[bad][:4], c[bad][:4])
    return int(edge.sum())
