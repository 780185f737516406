"""The opt-in checked device build (`make debug` -> libndinterp_hip_dbg.so, -DNDI_BOUNDS; SURVEY 5 "sanitizers":
GPU AddressSanitizer is not available on this pool).  Every device-side index goes through NDI_CHK: a violation is
recorded and the index clamped, the host turns a recorded violation into NDI_HIP_ERROR.  Here: (1) the seeded fuzz
test, the short-row variants, the tile-grouped 2-D order and the ring run under the checked library with NO
violation and unchanged results; (2) the checker itself fires when it is given a wrong limit."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "ndarray-interp_amd")
DBG = os.path.join(PKG, "libndinterp_hip_dbg.so")


def _ensure_debug_library():
    srcs = [os.path.join(PKG, "csrc", f) for f in ("ndinterp_api.hip", "kernels.hpp", "host_logic.hpp", "common.hpp")]
    if not os.path.exists(DBG) or os.path.getmtime(DBG) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "debug"], check=True, capture_output=True)


def test_checked_library_exports_the_same_abi(pkg):
    """CPU: the checked build is the same library (same exported entry points), selected with NDI_LIB."""
    import ctypes as C
    _ensure_debug_library()
    dbg = C.CDLL(DBG)
    for name in pkg._capi.SYMBOLS:
        assert hasattr(dbg, name), name


@pytest.mark.gpu
def test_parity_suite_runs_clean_under_the_checked_library():
    _ensure_debug_library()
    env = dict(os.environ, NDI_LIB="libndinterp_hip_dbg.so")
    sel = ("test_fuzz_against_oracle or test_short_rows_every_variant_bit_exact or test_bilinear_tile_grouped_lds or "
           "test_cubic_eval_bit_exact or test_linear_eval_bit_exact or test_short_rows_first_error_and_extrapolation or "
           "test_short_rows_in_the_ring_and_strided_buffers or test_bilinear_bit_exact or test_extrapolation_modes")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_short_rows.py"), "-m", "gpu", "-x", "-q", "-k", sel],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1]


@pytest.mark.gpu
def test_round6_kernels_run_clean_under_the_checked_library():
    """Round 6's kernels under the checked build: the slope-record 2-D kernel (cell indices, strip slots), the wide spline
    build, the sliced-output store paths, the library-owned outputs and the large-batch AUTO fuzz (every plan that opens at
    1e5 .. 1e6 queries, random shapes) -- no violation, unchanged results."""
    _ensure_debug_library()
    env = dict(os.environ, NDI_LIB="libndinterp_hip_dbg.so")
    files = [os.path.join(ROOT, "tests", f) for f in ("test_gpu_slopes2d.py", "test_gpu_spline_wide.py", "test_gpu_output_alloc.py",
                                                      "test_gpu_auto_fuzz.py")]
    r = subprocess.run([sys.executable, "-m", "pytest"] + files + [os.path.join(ROOT, "tests", "test_gpu_lanes.py"), "-m", "gpu", "-x", "-q",
                        "-k", "slopes or wide_build or output or sliced_output or rows_after_error or auto_large_batches"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1]


@pytest.mark.gpu
def test_the_checker_fires_on_a_wrong_limit():
    """NDI_BOUNDS_SELFTEST=1 makes the checked library claim ONE interval to the evaluation kernels: interval indices
    >= 1 are recorded and clamped (no fault), and the call reports the first violation."""
    _ensure_debug_library()
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
import torch
rng = np.random.default_rng(0)
x = np.sort(rng.uniform(0, 1, 64)); y = rng.uniform(0, 1, (64, 1024)); q = rng.uniform(x[0], x[-1], 5000)
it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device="cuda:0")).x(torch.as_tensor(x, device="cuda:0")).strategy(pkg.CubicSpline.new()).build()
for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
    it.strategy.path = path
    try:
        it.interp_array(torch.as_tensor(q, device="cuda:0"))
    except pkg.DeviceError as e:
        assert "device bounds check failed" in str(e) and "code 1" in str(e) and "limit 1" in str(e), str(e)
        print("FIRED", path)
    else:
        raise SystemExit("the checker did not fire")
import os
del os.environ["NDI_BOUNDS_SELFTEST"]
it2 = pkg.Interp1DBuilder.new(y[:, :8]).x(x).strategy(pkg.CubicSpline.new()).build()   # and a clean call afterwards
it2.interp_array(torch.as_tensor(q, device="cuda:0"))
print("CLEAN")
""" % ROOT
    env = dict(os.environ, NDI_LIB="libndinterp_hip_dbg.so", NDI_BOUNDS_SELFTEST="1", NDI_TUNE_LIVE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("FIRED") == 2 and "CLEAN" in r.stdout
