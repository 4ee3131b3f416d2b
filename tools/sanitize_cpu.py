#!/usr/bin/env python3
"""The CPU-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU, no GPU sanitizer: host builds only).

    SO3_SANITIZE=1 LD_PRELOAD=$(clang -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tools/sanitize_cpu.py

What is built with -fsanitize=address,undefined -fno-sanitize-recover=all (clang, ONE runtime for all of them) and what is driven:
  * oracle/kernel_model.cpp -- the shipped device templates compiled for the host (csrc/so3_device.h, so3_rows.h): every adversarial family
    of tests/test_kernel_model.py through the fast path, the Jacobi path, the packed pair path, the backward and the float64 path; the
    park list's reservation protocol under replayed interleavings (the code round 4's advisor found a race in by reading);
  * oracle/so3_oracle.c -- the C oracle: forward, backward, angle metric, Kabsch on random and degenerate input, ragged sizes;
  * poseestimation_amd/csrc/fastcall.c -- the CPython METH_FASTCALL glue: every argument count, None, negative and 64-bit integers,
    bad arguments (the error paths), against a sanitized stub with the C ABI's twelve-integer call shape;
  * examples/c_abi_demo.c -- compiled and linked with the same flags (it needs a GPU to RUN, and GPU sanitizers are not available on
    this pool: the host half is checked as far as the compiler's instrumentation and -Wall -Werror go).
numpy only (no torch: importing a ROCm torch under a preloaded sanitizer runtime is a test of torch, not of this code).
Prints "SANITIZE OK" and exits 0; a finding aborts the process with the sanitizer's report (tests/test_sanitizers.py asserts on both)."""
import ctypes
import importlib.util
import os
import subprocess
import sys
import sysconfig
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
assert os.environ.get("SO3_SANITIZE") == "1", "run with SO3_SANITIZE=1 (see the docstring)"
from oracle import c_oracle, kernel_model as km          # noqa: E402

CLANG = km.clangxx().replace("clang++", "clang")


def haar(n, rng):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)


def families(n, rng):
    """tests/test_kernel_model.py's adversarial families (rotations drawn from quaternions here), plus non-finite rows."""
    a = rng.standard_normal((n, 3, 3))
    q1, q2 = haar(n, rng), haar(n, rng)
    with_s = lambda s: q1 @ (s[:, :, None] * q2)
    yield "gaussian", a
    for e in (1e-1, 1e-4, 1e-7):
        s = np.ones((n, 3)); s[:, 1] -= e * rng.random(n); s[:, 2] -= 2 * e * rng.random(n)
        yield "clustered %.0e" % e, with_s(s)
    for e in (1e-2, 1e-4, 1e-6):
        yield "graded %.0e" % e, with_s(np.stack((np.ones(n), np.full(n, e), np.full(n, e * e)), 1))
    yield "rotation + noise", q1 + 1e-3 * a
    yield "symmetric", a + a.transpose(0, 2, 1)
    yield "small integers", rng.integers(-3, 4, (n, 3, 3)).astype(np.float64)
    yield "outer products", rng.standard_normal((n, 3, 1)) @ rng.standard_normal((n, 1, 3))
    yield "integer outer products", (rng.integers(-3, 4, (n, 3, 1)) @ rng.integers(-3, 4, (n, 1, 3))).astype(np.float64)
    yield "nine equal entries", np.broadcast_to(rng.standard_normal((n, 1, 1)), (n, 3, 3)).copy()
    yield "rank two", np.concatenate((a[:, :2], a[:, :1] + a[:, 1:2]), 1)
    yield "entries in -1..1", rng.integers(-1, 2, (n, 3, 3)).astype(np.float64)
    yield "reflections", q1 * np.array([1.0, 1.0, -1.0])
    for e in (1e-3, 1e-1):
        s = np.ones((n, 3)); s[:, 1] -= e * rng.random(n); s[:, 2] = -(1 - e * rng.random(n))
        yield "near-reflections %.0e" % e, with_s(s)
    for sc in (1e18, 1e-18, 2e-5, 5e-5, 3e4, 1e5, 1e30, 1e-30):
        yield "scaled %.0e" % sc, sc * a
    yield "zeros", np.zeros((n, 3, 3))
    bad = a.copy(); bad[::3, 0, 0] = np.nan; bad[1::3, 1, 1] = np.inf; bad[2::3, 2, 2] = -np.inf
    yield "non-finite", bad
    yield "denormal", 1e-42 * a


def drive_kernel_model():
    rng = np.random.default_rng(2024)
    n = 1500
    rows = 0
    for name, m in families(n, rng):
        with np.errstate(all="ignore"):
            m32 = m.astype(np.float32).reshape(n, 9)
        g = rng.standard_normal((n, 9)).astype(np.float32)
        r = km.project(m32)
        rp = km.project(m32, packed=True)
        rq, hard = km.project_quat(m32)
        rj = km.project_jacobi(m32)
        d = km.project_bwd(m32, g)
        dj = km.project_bwd_jacobi(m32, g)
        r64 = km.project_f64(m.reshape(n, 9))
        d64 = km.project_bwd_f64(m.reshape(n, 9), g.astype(np.float64))
        # an odd row count: the packed path pairs the last row with itself
        km.project(m32[:n - 1], packed=True)
        finite = np.isfinite(m32).all(1)
        if name not in ("non-finite",):
            assert np.isfinite(r[finite]).all() and np.isfinite(rp[finite]).all() and np.isfinite(rj[finite]).all(), name
            assert np.isfinite(r64[np.isfinite(m.reshape(n, 9)).all(1)]).all(), name
        assert r.shape == rq.shape == rj.shape == (n, 3, 3) and d.shape[0] == dj.shape[0] == d64.shape[0] == n and hard.shape == (n,)
        rows += n
    for name in ("quat", "euler", "ortho5d", "expmap"):
        width = {"quat": 4, "euler": 3, "ortho5d": 5, "expmap": 3}[name]
        x = rng.standard_normal((777, width)).astype(np.float32)
        x[::50] = 0.0
        with np.errstate(all="ignore"):
            km.head(name, x)
            km.head_bwd(name, x, rng.standard_normal((777, 9)).astype(np.float32))
    # the park list's reservation protocol under replayed interleavings (tests/test_kernel_model.py: the advisor's case first)
    count, base = km.park_reserve_interleaved(500, 512, [13, 5, 6])
    assert count == 511 and list(base) == [-1, 500, 505]
    for trial in range(3000):
        cap = int(rng.choice([256, 512]))
        start = int(rng.integers(max(0, cap - 80), cap + 1))
        k = rng.integers(1, 33, int(rng.integers(1, 5))).astype(np.uint32)
        count, base = km.park_reserve_interleaved(start, cap, k, nested=bool(trial & 1))
        at = start
        for b, kk in sorted((int(b), int(kk)) for b, kk in zip(base, k) if b >= 0):
            assert b == at
            at += kk
        assert at == count <= cap
    km.fast_path_counters()
    return rows


def drive_c_oracle():
    rng = np.random.default_rng(7)
    done = 0
    for n in (0, 1, 2, 63, 64, 65, 1000):
        for scale in (1.0, 1e-20, 1e20):
            m = (scale * rng.standard_normal((n, 9)))
            r64, flip = c_oracle.project(m, want_flip=True)
            r32 = c_oracle.project(m.astype(np.float32))
            assert r64.shape == (n, 3, 3) and r32.dtype == np.float32 and flip.shape == (n,)
            g = rng.standard_normal((n, 9)).astype(np.float32)
            c_oracle.project_bwd(m.astype(np.float32), g)
            deg, bad = c_oracle.angle_error(r32, c_oracle.project(rng.standard_normal((n, 9)).astype(np.float32)))
            assert deg.shape == (n,) and not bad
            done += n
    special = np.stack([np.zeros(9), np.eye(3).reshape(9), np.diag([1.0, 1.0, -1.0]).reshape(9), np.outer([1, 2, 3], [4, 5, 6]).reshape(9).astype(float),
                        np.full(9, np.nan), np.full(9, np.inf), np.full(9, 1e-310)])
    with np.errstate(all="ignore"):
        c_oracle.project(special)
        c_oracle.project(special.astype(np.float32))
        c_oracle.project_bwd(special.astype(np.float32), np.ones_like(special, dtype=np.float32))
        c_oracle.angle_error(special.astype(np.float32), special[::-1].astype(np.float32))
    for b, npts in ((1, 1), (3, 7), (17, 65), (5, 1024)):
        p, q = rng.random((b, npts, 3)).astype(np.float32) - 0.5, rng.random((b, npts, 3)).astype(np.float32) - 0.5
        c_oracle.kabsch(p, q)
        c_oracle.kabsch(p, q, want_h=True)
    return done


def build_sanitized(src, out, extra=()):
    subprocess.check_call([CLANG, "-std=c11", "-O1", "-fPIC", "-Wall", "-Werror", *km.SAN_FLAGS, *extra, "-o", out, src])
    return out


def drive_fastcall(tmp):
    stub_c = os.path.join(tmp, "stub.c")
    with open(stub_c, "w") as fh:
        fh.write("#include <stdint.h>\n"
                 "static volatile intptr_t seen[12];\n"
                 "int stub12(intptr_t a, intptr_t b, intptr_t c, intptr_t d, intptr_t e, intptr_t f, intptr_t g, intptr_t h, intptr_t i, intptr_t j, intptr_t k, intptr_t l) {\n"
                 "    intptr_t v[12] = {a, b, c, d, e, f, g, h, i, j, k, l}; int nz = 0;\n"
                 "    for (int n = 0; n < 12; ++n) { seen[n] = v[n]; nz += v[n] != 0; }\n"
                 "    return nz; }\n"
                 "int stub3(const void *p, int64_t n, void *stream) { return (p != 0) + (n == -1) * 10 + (stream == 0) * 100; }\n"
                 "int stub0(void) { return 210; }\n"
                 "intptr_t seen_at(int n) { return seen[n]; }\n")
    stub = ctypes.CDLL(build_sanitized(stub_c, os.path.join(tmp, "libstub.so"), ["-shared"]))
    stub.seen_at.restype = ctypes.c_int64
    ext = os.path.join(tmp, "_so3fast" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    build_sanitized(os.path.join(ROOT, "poseestimation_amd", "csrc", "fastcall.c"), ext, ["-shared", "-I", sysconfig.get_paths()["include"]])
    spec = importlib.util.spec_from_file_location("_so3fast", ext)
    fast = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fast)
    addr = lambda name: ctypes.cast(getattr(stub, name), ctypes.c_void_p).value
    assert fast.call(addr("stub0")) == 210
    assert fast.call(addr("stub0"), 1, None, 2 ** 63 + 5, -1) == 210               # surplus integer arguments are ignored by the callee
    assert fast.call(addr("stub3"), 4096, -1, None) == 111
    for n in range(0, 13):
        args = [(1 << (5 * k + 3)) + k + 1 for k in range(n)]
        assert fast.call(addr("stub12"), *args) == n
        assert [stub.seen_at(k) for k in range(12)] == args + [0] * (12 - n)
    assert fast.call(addr("stub12"), -1, -2 ** 63, 2 ** 64 - 1, None, 0) == 3 and stub.seen_at(0) == -1 and stub.seen_at(1) == -2 ** 63 and stub.seen_at(2) == -1
    for bad in ((), (0,), (None,), (addr("stub0"),) + (1,) * 13, (addr("stub0"), 1.5), (addr("stub0"), "x"), ("address",)):
        try:
            fast.call(*bad)
        except (TypeError, ValueError):
            continue
        raise AssertionError("fastcall accepted %r" % (bad,))
    return 13


def build_demo(tmp):
    """examples/c_abi_demo.c with the sanitizers' instrumentation, linked against the shipped library (running it needs a GPU)."""
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime_api.h")) or not os.path.exists(os.path.join(ROOT, "poseestimation_amd", "libso3proj.so")):
        return "skipped (no ROCm headers or no built library)"
    libdir = os.path.join(ROOT, "poseestimation_amd")
    out = os.path.join(tmp, "c_abi_demo_san")
    subprocess.check_call([CLANG, "-std=c11", "-O1", "-Wall", "-Werror", *km.SAN_FLAGS, "-D__HIP_PLATFORM_AMD__", "-I" + hip_inc, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + libdir, "-lso3proj", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return "built"


def main():
    km.build()
    c_oracle.build()
    assert km.LIB.endswith("_san.so") and c_oracle._SO.endswith("_san.so")
    rows = drive_kernel_model()
    print("kernel_model.cpp: %d rows of %d families through every entry, 3000 park-list interleavings" % (rows, rows // 1500), flush=True)
    print("so3_oracle.c: %d rows, ragged sizes, degenerate and non-finite rows, Kabsch" % drive_c_oracle(), flush=True)
    with tempfile.TemporaryDirectory() as tmp:
        print("fastcall.c: argument counts 0..%d, None / negative / 64-bit integers, 7 refused calls" % (drive_fastcall(tmp) - 1), flush=True)
        print("c_abi_demo.c: %s" % build_demo(tmp), flush=True)
    print("SANITIZE OK")


if __name__ == "__main__":
    main()
