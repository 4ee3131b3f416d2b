#!/usr/bin/env python3
"""The adversarial search on the fast path's certificate (tests/test_gpu_certificate_search.py) under several seeds and for any build of
the library: the worst accepted |dR| gap/s1 per seed and over all of them (the test asserts its bound for its own seeds only).
usage: search_seeds.py path/to/libso3proj.so seed [seed ...]      builds: tools/build_variant.sh name -DSO3_QUAT_CURV=... / -DSO3_QUAT_CONV=... / -DSO3_QUAT_CLOSE=..."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import test_gpu_certificate_search as cs
cs.BOUND = 1.0
worst = []
for seed in sys.argv[2:]:
    os.environ["SO3_SEARCH_SEED"] = seed
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cs.test_adversarial_search_finds_no_accepted_row_beyond_the_bound(int(seed))
    line = [l for l in buf.getvalue().split("\n") if "adversarial search" in l][0]
    w = float(line.split("gap/s1 = ")[1].split(" ")[0]); acc = line.split("rows, ")[1].split(" accepted")[0]
    worst.append(w)
    print(os.path.basename(sys.argv[1]), "seed", seed, "worst %.3g" % w, "accepted", acc, flush=True)
print(os.path.basename(sys.argv[1]), "max over seeds %.3g" % max(worst))
