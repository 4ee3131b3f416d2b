"""CPU-side checks of the boundary: the C-ABI library loads and exports every declared symbol, the
Python mirror refuses to run without a HIP device (no CPU fallback, no oracle in the product)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "so3proj.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "static inline" not in text                  # round 4's one-round wrappers of the round-3 spellings are gone
    return sorted(set(re.findall(r"\b(so3_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_library):
    lib = ctypes.CDLL(built_library)
    names = header_symbols()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/so3proj.h but not exported"


def test_reducing_entry_points_exist_once(built_library):
    """One export per reducing entry point (workspace nullable, a flags word), the float64 ones included; the library exports exactly
    what the header declares, and none of the spellings of rounds 3 and 4."""
    import subprocess
    exported = subprocess.run(["nm", "-D", "--defined-only", built_library], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(so3_[a-z0-9_]+)\b", exported))
    assert exported == set(header_symbols())                      # nothing undeclared is exported either
    for name in ("so3_frob_fwd_bwd_v2_f32", "so3_frob_fwd_bwd_v2_bf16", "so3_frob_loss_v2_f32", "so3_angle_error_v2", "so3_project_angle_error_v2_f32",
                 "so3_angle_error_v2_f64", "so3_frob_loss_v2_f64"):
        assert name in exported
    for gone in ("so3_frob_fwd_bwd_f32", "so3_frob_fwd_bwd_ws_f32", "so3_frob_loss_f32", "so3_angle_error", "so3_angle_error_ws", "so3_angle_error_acc",
                 "so3_project_angle_error_f32", "so3_angle_error_f64", "so3_frob_loss_f64"):
        assert gone not in exported


def test_a_library_of_another_abi_version_is_refused(built_library, monkeypatch):
    """The binding checks so3_version() when it loads (advisor, round 4: argument lists had changed under unchanged names)."""
    from poseestimation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", 100)
    with pytest.raises(ImportError, match="ABI version 210"):
        _lib.load()
    monkeypatch.setattr(_lib, "ABI_VERSION", 210)
    assert _lib.load().so3_version() == 210


def test_binding_table_matches_header(built_library):
    from poseestimation_amd import _lib
    assert sorted(_lib.SYMBOLS) == header_symbols()
    lib = _lib.load()
    assert lib.so3_version() == 210 == _lib.ABI_VERSION
    assert lib.so3_last_error() == b""


def build_c_demo(built_library, out_dir):
    """examples/c_abi_demo.c compiled as C11 against include/so3proj.h and the shared library (plain gcc, no hipcc)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc or the ROCm headers are not available")
    exe = os.path.join(str(out_dir), "c_abi_demo")
    libdir = os.path.dirname(built_library)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"),
                           "-L" + libdir, "-lso3proj", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_header_is_plain_c_and_links(built_library, tmp_path):
    """The boundary is usable without C++, Python or torch: the demo compiles as C11 and links against the .so."""
    assert os.path.exists(build_c_demo(built_library, tmp_path))


@pytest.mark.gpu
def test_c_demo_runs_on_the_gpu(built_library, tmp_path):
    import subprocess
    exe = build_c_demo(built_library, tmp_path)
    for rows in ("100003", "64", "1"):
        res = subprocess.run([exe, rows], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


@pytest.mark.gpu
def test_last_kernel_names_the_instantiation_that_ran(built_library):
    """so3_last_kernel(): the streaming kernel's name as a profiler prints it, formed from the launch's template arguments."""
    from poseestimation_amd import _lib, rotation_representation as rr
    lib = _lib.load()
    x = torch.randn(4096, 9, device="cuda:0")
    rr.symmetric_orthogonalization(x)
    assert lib.so3_last_kernel() in (b"so3::k_rows<so3::OpProject<4, false>, 2, 3, 256, false>",)
    rr.symmetric_orthogonalization(x.bfloat16())
    assert lib.so3_last_kernel().startswith(b"so3::k_rows<so3::OpProject<2, false>,")
    rr.angle_error(rr.symmetric_orthogonalization(x), rr.symmetric_orthogonalization(x.flip(0)))
    assert lib.so3_last_kernel().startswith(b"so3::k_rows<so3::OpAngle<")


def test_argument_validation_without_gpu(built_library):
    """Bad arguments are rejected on the host before any launch (no GPU needed)."""
    from poseestimation_amd import _lib
    lib = _lib.load()
    assert lib.so3_project_fwd_f32(None, None, None, -1, None) == -1
    assert b"so3_project_fwd" in lib.so3_last_error()
    assert lib.so3_project_fwd_f32(None, None, None, 5, None) == -1          # null pointers with B > 0
    assert lib.so3_project_fwd_f32(None, None, None, 0, None) == 0           # B == 0 is a no-op
    assert lib.so3_project_bwd_f32(None, None, None, 3, None) == -1
    assert lib.so3_geodesic_f32(None, None, None, 0, None) == 0
    assert lib.so3_kabsch_f32(None, None, None, None, 0, 1024, None) == 0
    assert lib.so3_kabsch_f32(None, None, None, None, 4, -1, None) == -1


def test_cpu_tensor_is_refused_not_silently_computed():
    import poseestimation_amd as pa
    x = torch.randn(4, 9)
    for call in (lambda: pa.symmetric_orthogonalization(x),
                 lambda: pa.angle_error(torch.eye(3)[None], torch.eye(3)[None]),
                 lambda: pa.compute_geodesic_distance_from_two_matrices(torch.eye(3)[None], torch.eye(3)[None]),
                 lambda: pa.loss_frobenius(torch.eye(3)[None], torch.eye(3)[None]),
                 lambda: pa.frobenius_head(x, torch.eye(3).repeat(4, 1, 1)),
                 lambda: pa.kabsch_rotation(torch.randn(2, 8, 3), torch.randn(2, 8, 3))):
        with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
            call()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from poseestimation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libso3proj.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "poseestimation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "so3_oracle" not in text and "libso3oracle" not in text, f
                assert "linalg.svd" not in text and "torch.svd" not in text, f
                if f.endswith(".py"):          # no ATen arithmetic stands in for a kernel: float64 arguments have kernels of their own
                    for banned in ("torch.acos", ".acos(", "matrix_norm", "linalg.norm("):
                        assert banned not in text, (f, banned)


def test_dispatch_table_keys():
    import poseestimation_amd as pa
    dim, fn = pa.transform_output["SVD"]
    assert dim == 9 and fn is pa.symmetric_orthogonalization


def test_shard_range_partitions():
    from poseestimation_amd.distributed import shard_range
    for total in (0, 1, 7, 16, 1_000_003, 16_000_000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_fastcall_module_reaches_the_library_without_a_gpu():
    """poseestimation_amd/_so3fast (csrc/fastcall.c): the METH_FASTCALL entry the mirror uses for its enqueue-only calls.  No
    compute here: the version call, argument conversion (None = null pointer, ints above 2^63, negative ints) and an error
    status that the library raises before it touches the device."""
    import ctypes
    from poseestimation_amd import _lib, build
    if build.build_fastcall() is None:
        pytest.skip("no C compiler / Python.h: the mirror calls through ctypes")
    from poseestimation_amd import _so3fast
    lib = _lib.load()
    addr = lambda name: ctypes.cast(getattr(lib, name), ctypes.c_void_p).value
    assert _so3fast.call(addr("so3_version")) == lib.so3_version() == 210
    assert _so3fast.call(addr("so3_version"), 1, None, 2 ** 63 + 5, -1) == 210          # extra integer arguments are ignored
    assert _so3fast.call(addr("so3_scale_f32"), None, None, None, 5, None) != 0          # null pointers: SO3_ERR_INVALID
    assert b"so3_scale_f32" in lib.so3_last_error()
    assert _so3fast.call(addr("so3_project_fwd_f32"), None, None, None, 0, None) == 0     # B = 0: nothing to do, no launch
    with pytest.raises(TypeError):
        _so3fast.call()
    with pytest.raises(ValueError):
        _so3fast.call(0)
    with pytest.raises(TypeError):
        _so3fast.call(addr("so3_version"), 1.5)


def test_autograd_node_declines_what_it_does_not_cover():
    """poseestimation_amd/_so3node (csrc/autograd_node.cpp): frobenius_head's autograd node in C++.  Without a GPU only its
    refusals can be exercised: it returns None -- the mirror then takes the Python autograd.Function -- for CPU tensors, for
    dtypes, shapes and layouts outside its case, and before the C-ABI addresses are bound."""
    import torch
    from poseestimation_amd import build
    if build.build_autograd_node() is None:
        pytest.skip("no C++ compiler / torch headers: the mirror uses its Python autograd.Function")
    from poseestimation_amd import _so3node
    from poseestimation_amd import rotation_representation as rr
    x, t = torch.randn(8, 9), torch.randn(8, 3, 3)
    assert _so3node.frobenius_head(x, t, True, 0, 0) is None                       # not bound yet / CPU tensors
    assert rr._node() is _so3node                                                    # binds the addresses out of libso3proj.so
    for xi, ti in ((x, t), (x.half(), t), (x[:, :8], t), (x.t().contiguous().t(), t), (x, t.double()), (x, t[:4]), (torch.randn(9), t)):
        assert _so3node.frobenius_head(xi, ti, True, 0, 0) is None
    xg = x.clone().requires_grad_(True)
    assert _so3node.symmetric_orthogonalization(xg, 0) is None and _so3node.loss_frobenius(xg.view(8, 3, 3), t, 0, 0) is None   # CPU tensors
    with pytest.raises(RuntimeError, match="HIP device only"):                       # the mirror's own error for CPU tensors is unchanged
        rr.frobenius_head(x, t)
    with pytest.raises(RuntimeError, match="HIP device only"):
        rr.symmetric_orthogonalization(xg)
    with pytest.raises(RuntimeError, match="HIP device only"):
        rr.loss_frobenius(xg.view(8, 3, 3), t)



def test_a_missing_optional_helper_is_announced_once():
    """_so3fast / _so3node only remove host overhead, so their absence must not fail the import -- but it must not go unnoticed
    either (round 3: _so3node is compiled against the build machine's torch; another torch on the box that runs it silently moved
    config #4's step from the C++ nodes to the Python classes).  With _so3node hidden from the import system the package imports,
    says so on stderr once, and keeps _so3fast."""
    import subprocess
    import sys
    code = ("import sys, importlib.abc\n"
            "class Hide(importlib.abc.MetaPathFinder):\n"
            "    def find_spec(self, name, path, target=None):\n"
            "        if name.endswith('._so3node'):\n"
            "            raise ImportError('hidden by the test')\n"
            "sys.meta_path.insert(0, Hide())\n"
            "sys.path.insert(0, %r)\n"
            "import poseestimation_amd.rotation_representation as rr\n"
            "print('node', rr._so3node is None, 'fast', rr._so3fast is not None)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "node True fast True" in out.stdout
    assert out.stderr.count("optional helper _so3node is not available") == 1 and "hidden by the test" in out.stderr
    assert "_so3fast is not available" not in out.stderr


def test_graft_entry_build_runs_here(built_library):
    """The driver's "does it build" check, as the driver calls it (round 5 caught build() still asserting the previous ABI version)."""
    import __graft_entry__ as entry
    entry.build()
