"""The measurement tools are run by hand on the GPU box, where a syntax error costs a call: every script under tools/ at least parses here."""
import glob
import os
import py_compile
import subprocess

from conftest import ROOT


def test_every_python_tool_compiles(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) > 15
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_every_shell_tool_parses():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert len(files) >= 5
    for f in files:
        out = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert out.returncode == 0, (f, out.stderr)
