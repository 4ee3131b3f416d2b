"""The CPU-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (round-5 review, item 6): oracle/kernel_model.cpp (the
shipped device templates compiled for the host, park-list protocol included), oracle/so3_oracle.c, csrc/fastcall.c and the host half of
examples/c_abi_demo.c are built with -fsanitize=address,undefined -fno-sanitize-recover=all (SO3_SANITIZE=1) and driven by
tools/sanitize_cpu.py in a child process that has the sanitizer's shared runtime preloaded.  No GPU, no GPU sanitizer."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _runtime():
    from oracle import kernel_model
    cxx = kernel_model.clangxx()
    if cxx is None:
        return None, None
    cc = cxx.replace("clang++", "clang")
    rt = subprocess.run([cc, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return (cc, rt) if os.path.isabs(rt) and os.path.exists(rt) else (cc, None)


def test_cpu_builds_are_clean_under_asan_and_ubsan():
    cc, rt = _runtime()
    if rt is None:
        pytest.skip("clang or its AddressSanitizer runtime is not available")
    env = dict(os.environ, SO3_SANITIZE="1", LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize_cpu.py")], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("SANITIZE OK"), (out.stdout[-1500:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
    for line in ("kernel_model.cpp:", "so3_oracle.c:", "fastcall.c:", "c_abi_demo.c:"):
        assert line in out.stdout, out.stdout
    # the libraries that were driven are the instrumented ones: they import the sanitizer's entry points
    for lib in ("libso3model_san.so", "libso3oracle_san.so"):
        syms = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(ROOT, "oracle", lib)], capture_output=True, text=True).stdout
        assert "__asan_init" in syms and "__ubsan_handle" in syms, lib


def test_the_harness_reports_a_real_finding(tmp_path):
    """A deliberate heap overflow, built and preloaded the same way, stops the child with AddressSanitizer's report: the green run
    above is not green because the instrumentation is idle."""
    cc, rt = _runtime()
    if rt is None:
        pytest.skip("clang or its AddressSanitizer runtime is not available")
    from oracle import kernel_model
    src = tmp_path / "oob.c"
    src.write_text("#include <stdlib.h>\nint oob(int n) { int *p = malloc(4 * sizeof(int)); int v = p[n]; free(p); return v; }\n"
                   "int shift(int n) { return 1 << n; }\n")
    lib = tmp_path / "liboob.so"
    subprocess.check_call([cc, "-std=c11", "-O1", "-fPIC", "-shared", *kernel_model.SAN_FLAGS, "-o", str(lib), str(src)])
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    for call, needle in (("oob(4)", "heap-buffer-overflow"), ("shift(40)", "shift exponent 40 is too large")):
        code = "import ctypes; l = ctypes.CDLL(%r); l.%s" % (str(lib), call)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
        assert out.returncode != 0 and needle in out.stderr, (call, out.returncode, out.stderr[-800:])
