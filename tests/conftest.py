"""pytest configuration: registers the `gpu` marker, puts the repo root on sys.path and
loads the product package (directory `ndarray-interp_amd/`, imported as `ndarray_interp_amd`)."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library re-reads its kernel-variant knobs on every evaluation (instead of once) when this is set at its first
# evaluation: tests/test_gpu_short_rows.py switches the variants inside one process
os.environ.setdefault("NDI_TUNE_LIVE", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_product_package():
    """The package directory carries a hyphen (fixed by the project layout), so it is loaded by path."""
    name = "ndarray_interp_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "ndarray-interp_amd")
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _ensure_native_library():
    """A fresh checkout has no built artefacts (they are git-ignored): build the product library once,
    exactly as __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU)."""
    import subprocess
    lib = os.path.join(ROOT, "ndarray-interp_amd", "libndinterp_hip.so")
    csrc = os.path.join(ROOT, "ndarray-interp_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in ("ndinterp_api.hip", "kernels.hpp", "host_logic.hpp", "common.hpp")]
    srcs.append(os.path.join(ROOT, "include", "ndinterp.h"))
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", csrc], check=True, capture_output=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_native_library()


def _f(v):
    return float(v) if isinstance(v, str) else v


def tofloat(seq):
    return [_f(v) for v in seq]


@pytest.fixture(scope="session")
def refvec():
    with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def scipy_golden():
    return np.load(os.path.join(GOLDEN, "scipy_cubic.npz"))


def rel_ok(got, ref, atol, rtol):
    """approx::assert_relative_eq! semantics (the form the reference's tests use,
    e.g. tests/cubic_spline_strat.rs:26): |d| <= atol  or  |d| <= rtol * max(|got|, |ref|)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = np.abs(got - ref)
    return (d <= atol) | (d <= rtol * np.maximum(np.abs(got), np.abs(ref)))


def assert_rel(got, ref, atol, rtol, what=""):
    ok = rel_ok(got, ref, atol, rtol)
    if not np.all(ok):
        got = np.asarray(got, dtype=np.float64)
        ref = np.asarray(ref, dtype=np.float64)
        bad = np.argwhere(~ok)
        i = tuple(bad[0])
        raise AssertionError(
            f"{what}: {bad.shape[0]} of {ok.size} elements off; first at {i}: got {got[i]!r} ref {ref[i]!r} "
            f"(atol={atol}, rtol={rtol})")


@pytest.fixture(scope="session")
def pkg():
    return load_product_package()
