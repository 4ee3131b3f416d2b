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
