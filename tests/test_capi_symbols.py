"""CPU: the C-ABI library loads and exports every symbol include/ndinterp.h declares; the host-only
entry points (validation, monotonic scan) work without a GPU; compute entry points fail loudly (no
CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ndinterp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ndi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(pkg):
    names = _declared_symbols()
    assert len(names) >= 18
    lib = C.CDLL(pkg._capi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding covers exactly the header
    assert sorted(pkg._capi.SYMBOLS) == names


def test_rust_shim_binds_every_export():
    """rust/ndarray-interp-hip/src/hip_ffi.rs declares every symbol of the header (and nothing else); the types are
    compared in tests/test_rust_ffi_abi.py.  INTEGRATION.md quotes the files of the shim: they must exist."""
    text = open(os.path.join(ROOT, "rust", "ndarray-interp-hip", "src", "hip_ffi.rs")).read()
    rust = sorted(set(re.findall(r"pub fn (ndi_[a-z0-9_]+)\s*\(", text)))
    assert rust == _declared_symbols()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for rel in ("rust/ndarray-interp-hip/build.rs", "rust/patches/ndarray-interp-0.6.0-batched-hook.patch"):
        assert rel in doc and os.path.exists(os.path.join(ROOT, rel)), rel
    for f in ("lib", "hip_ffi", "strategies", "ring", "sharded"):
        assert os.path.exists(os.path.join(ROOT, "rust", "ndarray-interp-hip", "src", f + ".rs")), f
    patch = open(os.path.join(ROOT, "rust", "patches", "ndarray-interp-0.6.0-batched-hook.patch")).read()
    for target in ("src/interp1d/strategies/mod.rs", "src/interp2d/strategies/mod.rs", "src/interp1d/mod.rs",
                   "src/interp2d/mod.rs"):
        assert "+++ b/" + target in patch, target
    assert patch.count("+    fn interp_array_into<") == 2   # one defaulted method per finished-strategy trait


def test_version_and_error_string(pkg):
    lib = pkg._capi.lib()
    assert lib.ndi_version() == (0 << 16) | 5
    assert isinstance(pkg._capi.last_error(), str)


def test_host_validation_through_c_abi(pkg, refvec):
    lib = pkg._capi.lib()
    for case in refvec["builder1d_errors"]:
        x = np.array(case["x"], dtype=np.float64)
        st = lib.ndi_validate1d(pkg._capi.F64, x.ctypes.data, x.size, case["n_data"], pkg._capi.LINEAR)
        assert pkg._capi.STATUS_NAMES[st] == case["expect_status"], case["name"]
    for case in refvec["cubic_errors"]:
        if "n_data" not in case:
            continue
        x = np.array(case["x"], dtype=np.float32)
        st = lib.ndi_validate1d(pkg._capi.F32, x.ctypes.data, x.size, case["n_data"], pkg._capi.CUBIC_SPLINE)
        assert pkg._capi.STATUS_NAMES[st] == case["expect_status"], case["name"]
    for case in refvec["builder2d_errors"]:
        x = np.array(case["x"], dtype=np.float64); y = np.array(case["y"], dtype=np.float64)
        st = lib.ndi_validate2d(pkg._capi.F64, x.ctypes.data, x.size, y.ctypes.data, y.size, case["nx"], case["ny"])
        assert pkg._capi.STATUS_NAMES[st] == case["expect_status"], case["name"]


def test_no_device_means_loud_failure(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.DeviceError, match="no CPU fallback"):
        pkg.Interp1DBuilder.new(np.array([1.0, 2.0, 3.0])).build()
    with pytest.raises(pkg.DeviceError):
        pkg.get_lower_index(np.array([0.0, 1.0]), np.array([0.5]))


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under ndarray-interp_amd/ may reference it
    pkg_dir = os.path.join(ROOT, "ndarray-interp_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "import oracle" not in text and "oracle/" not in text, f


def test_argument_validation_needs_no_device(pkg):
    """BAD_ARG / builder errors that are decided before any device work behave the same with and without a GPU."""
    import ctypes
    lib = pkg._capi.lib()
    h = ctypes.c_void_p()
    assert lib.ndi_interp1d_create(None, ctypes.byref(h)) == pkg._capi.BAD_ARG
    d = pkg._capi.Interp1DDesc()
    d.dtype = 7
    assert lib.ndi_interp1d_create(ctypes.byref(d), ctypes.byref(h)) == pkg._capi.BAD_ARG
    assert "dtype" in pkg._capi.last_error()
    # validate=1 with host axes: the reference's builder errors come back without touching a device
    x = np.array([1.0, 2.0, 2.0]); y = np.array([1.0, 2.0, 3.0])
    d = pkg._capi.Interp1DDesc()
    d.dtype, d.strategy, d.n, d.lanes, d.x_len = pkg._capi.F64, pkg._capi.LINEAR, 3, 1, 3
    d.x, d.data, d.memspace, d.validate = x.ctypes.data, y.ctypes.data, pkg._capi.MEM_HOST, 1
    assert lib.ndi_interp1d_create(ctypes.byref(d), ctypes.byref(h)) == pkg._capi.MONOTONIC
    assert "strictly monotonic rising" in pkg._capi.last_error()
    d.x = None; d.n = 1; d.x_len = 1
    assert lib.ndi_interp1d_create(ctypes.byref(d), ctypes.byref(h)) == pkg._capi.NOT_ENOUGH_DATA
    assert lib.ndi_interp1d_eval(None, None, 0, None, 0, None, None) == pkg._capi.BAD_ARG
    assert lib.ndi_interp2d_eval(None, None, None, 0, None, 0, None, None) == pkg._capi.BAD_ARG
    assert lib.ndi_profile_read(None, 0) == pkg._capi.BAD_ARG


def test_header_is_plain_c99(pkg, tmp_path):
    """The boundary is a C ABI: the header must compile as strict C99 (no C++-isms, no torch/HIP types) and a
    C program using it must link against the library."""
    import subprocess
    exe = tmp_path / "c_abi_example"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                        os.path.join(ROOT, "examples", "c_abi_example.c"), "-I", os.path.join(ROOT, "include"),
                        "-L", os.path.join(ROOT, "ndarray-interp_amd"), "-lndinterp_hip",
                        "-Wl,-rpath," + os.path.join(ROOT, "ndarray-interp_amd"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    # (the library is reached through the package, i.e. after torch: a process that loads this library first and torch
    # second ends up with two HIP runtimes -- torch bundles its own -- and torch then sees no device)
    if pkg._capi.lib().ndi_device_count() == 0:
        assert run.returncode == 1 and "no CPU fallback" in run.stderr     # loud failure without a GPU
    else:
        vals = [float(v) for v in run.stdout.split()]
        exp = [0.5, 0.1851851851851852, 0.01851851851851853, -5.551115123125783e-17, 0.12962962962962965,
               0.40740740740740755, 0.8333333333333331, 1.407407407407407, 2.1296296296296293, 3.0]
        assert run.returncode == 0 and max(abs(a - b) for a, b in zip(vals, exp)) <= 2.220446049250313e-16


def test_sharded_example_is_plain_c99(pkg, tmp_path):
    """examples/c_abi_sharded.c: replicas by ndi_interp1d_clone + ndi_interp1d_eval_sharded from strict C99; loud
    failure without a GPU, the sharded result equal to the single-handle one (and the global first error) with one."""
    import ctypes
    import subprocess
    exe = tmp_path / "c_abi_sharded"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                        os.path.join(ROOT, "examples", "c_abi_sharded.c"), "-I", os.path.join(ROOT, "include"),
                        "-L", os.path.join(ROOT, "ndarray-interp_amd"), "-lndinterp_hip",
                        "-Wl,-rpath," + os.path.join(ROOT, "ndarray-interp_amd"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    if pkg._capi.lib().ndi_device_count() == 0:
        assert run.returncode == 1 and "no CPU fallback" in run.stderr
    else:
        assert run.returncode == 0, run.stdout + run.stderr
        assert "sharded == single-handle result; first error at flat index 5 (x = -1000000000 is not in range)" in run.stdout


@pytest.mark.gpu
def test_c_examples_on_the_device(pkg, tmp_path):
    """The same two C99 programs in the GPU suite: there they must run to completion against the device."""
    assert pkg.device_count() >= 1
    test_header_is_plain_c99(pkg, tmp_path)
    test_sharded_example_is_plain_c99(pkg, tmp_path)
