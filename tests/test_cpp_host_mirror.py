"""The C++ host mirror (ndarray-interp_amd/host/ndarray_interp.hpp) above the C ABI: compiled with g++,
`--host-only` checks the builder validation on CPU, the full run drives the device path the way the
reference's tests drive the crate."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
HDR = os.path.join(ROOT, "ndarray-interp_amd", "host", "ndarray_interp.hpp")
LIBDIR = os.path.join(ROOT, "ndarray-interp_amd")


def _build():
    newest = max(os.path.getmtime(p) for p in (SRC, HDR, os.path.join(ROOT, "include", "ndinterp.h")))
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < newest:
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", BIN, SRC, "-L", LIBDIR, "-lndinterp_hip",
                        "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True)
    return BIN


def test_cpp_mirror_host_logic(pkg):
    r = subprocess.run([_build(), "--host-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_device_path(pkg):
    r = subprocess.run([_build()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
