"""The driver's entry points in the orders a harness may call them.  build() loads libbtba.so; if that made /opt/rocm's
libamdhip64 the process's HIP runtime before torch brought its own copy (same soname), the first launch of smoke() failed with
hipErrorNoDevice -- _lib.lib() therefore imports torch first."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "smoke"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    assert "build ok" in out.stdout and "smoke: max |rot| diff" in out.stdout


def test_library_is_loaded_behind_torch():
    code = ("import sys; from bundletrack_amd import _lib; assert 'torch' not in sys.modules; _lib.lib(); "
            "assert 'torch' in sys.modules; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout + out.stderr)[-2000:]
