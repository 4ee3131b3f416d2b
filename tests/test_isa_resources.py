"""The device code as compiled (`hipcc --cuda-device-only -S`, 25 s, no GPU): no kernel of the library may touch scratch memory, and the
streaming kernels must fit the waves per SIMD their launch shape assumes.  Round 4's hard-row commit put a 12-byte private segment (four
spilled VGPRs, in the one-round-in-five second eigenvector block) into the benchmark kernel and nothing noticed; this test would have."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not available")
    from poseestimation_amd import build
    out = str(tmp_path_factory.mktemp("isa") / "so3proj.s")
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    subprocess.run([HIPCC, *flags, "--cuda-device-only", "-S", "-o", out, os.path.join(build.CSRC, "so3proj.hip")], check=True, capture_output=True)
    text = open(out).read()
    names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M)
    nice = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")))
    table = {}
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
        body = m.group(2)
        get = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        table[nice[m.group(1)]] = {"scratch": get("private_segment_fixed_size"), "vgpr": get("next_free_vgpr"), "lds": get("group_segment_fixed_size")}
    assert not [n for n in re.findall(r"\.vgpr_spill_count:\s*(\d+)", text) if int(n) != 0]      # the code object's own notes agree
    assert len(table) > 100
    return table


def test_no_kernel_uses_scratch(kernels):
    bad = {k: v for k, v in kernels.items() if v["scratch"] != 0}
    assert not bad, "private segment (register spills or stack objects) in: %r" % bad


def test_streaming_kernels_fit_their_occupancy(kernels):
    """k_rows<Op, NPL, WPS, BLOCK>: amdgpu_waves_per_eu(WPS, WPS) asks for WPS waves per SIMD; 512 VGPRs per SIMD lane in units of 8."""
    seen = set()
    for name, v in kernels.items():
        m = re.match(r"void so3::k_rows<so3::(Op\w+)(?:<.*?>)?, (\d+), (\d+), (\d+), false>", name)
        if not m:
            continue
        op, wps, block = m.group(1), int(m.group(3)), int(m.group(4))
        seen.add(op)
        budget = (512 // wps) // 8 * 8
        assert v["vgpr"] <= budget, (name, v)
        assert v["lds"] * (4 * wps * 64 // block) <= 160 * 1024, (name, v)        # the workgroups that share a CU fit its LDS
    assert {"OpProject", "OpProjectBwd", "OpFrobHead", "OpProjectAngle", "OpAngle", "OpGeodesic"} <= seen


def test_benchmark_kernel_keeps_three_waves_per_simd(kernels):
    for flip in ("false", "true"):
        for eb in (4, 2):
            v = kernels["void so3::k_rows<so3::OpProject<%d, %s>, 2, 3, 256, false>(so3::OpProject<%d, %s>, long, unsigned long long*)" % (eb, flip, eb, flip)]
            assert v["scratch"] == 0 and v["vgpr"] <= 168, (eb, flip, v)
