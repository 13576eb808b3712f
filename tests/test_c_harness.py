"""The C ABI consumed from plain C (SURVEY.md §4 iv): compile tests/c_harness/abi_harness.c with gcc against
include/softbody.h, link libsoftbody_mi355x.so, run it."""
import os
import subprocess

import pytest

from softbodyunity_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "abi_harness")
    libdir = os.path.dirname(native.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_harness", "abi_harness.c"), "-o", exe,
                           "-L", libdir, "-lsoftbody_mi355x", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_harness_host_side(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_harness_gpu(tmp_path):
    out = subprocess.run([_build(tmp_path), "gpu"], capture_output=True, text=True, env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr
