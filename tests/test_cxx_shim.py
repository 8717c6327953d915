"""The C++17 mirror of the reference classes (include/dpgo_hip.hpp) compiles with the host compiler,
links against libdpgo_hip.so, and re-runs the reference's known-answer tests through it."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(tmp_path):
    exe = os.path.join(str(tmp_path), "test_shim")
    libdir = os.path.join(ROOT, "dpgo_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cxx", "test_shim.cpp"), "-L", libdir, "-ldpgo_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cxx_shim_compiles_and_refuses_without_device(tmp_path):
    import dpgo_amd
    exe = _build(tmp_path)
    if dpgo_amd.device_count() > 0:
        pytest.skip("a GPU is present (covered by the gpu test)")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 77, p.stdout + p.stderr  # DPGO_ERR_HIP, no CPU fallback


@pytest.mark.gpu
def test_cxx_shim_reference_known_answers(tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "triangle" in p.stdout and "prior" in p.stdout and "project: ok" in p.stdout
    assert "rounding" in p.stdout and "robust: inlier" in p.stdout and "robust: outlier" in p.stdout
    assert "precond 1:" in p.stdout and "precond 2:" in p.stdout  # block-Jacobi and multilevel
