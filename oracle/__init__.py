"""ctypes/numpy front-end of the CPU oracle (oracle/oracle.cpp).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg -- never from the product package
(``ndarray-interp_amd``), which has no CPU fallback.

Each function here maps 1:1 onto one ``oracle_*`` symbol; the reference
file:line each follows is cited in oracle.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

OK, NOT_ENOUGH_DATA, MONOTONIC, SHAPE, VALUE, OUT_OF_BOUNDS, NAN_QUERY = range(7)
STATUS_NAMES = ["OK", "NOT_ENOUGH_DATA", "MONOTONIC", "SHAPE", "VALUE", "OUT_OF_BOUNDS", "NAN_QUERY"]

MONO_NAMES = {0: "NotMonotonic", 1: "Rising{strict:true}", 2: "Rising{strict:false}",
              3: "Falling{strict:true}", 4: "Falling{strict:false}"}

BC_NOT_A_KNOT, BC_NATURAL, BC_CLAMPED, BC_FIRST_DERIV, BC_SECOND_DERIV = range(5)
EXTRAPOLATE_NO, EXTRAPOLATE_YES, EXTRAPOLATE_PERIODIC = range(3)


def build(force: bool = False) -> str:
    """Compile liboracle.so with oracle/Makefile (g++, -ffp-contract=off)."""
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """liboracle.so (portable x86-64-v3 build), or the library named by ORACLE_LIB (the sanitizer build in
    tests/test_oracle_sanitized.py)."""
    global _lib
    if _lib is None:
        override = os.environ.get("ORACLE_LIB")
        if override:
            _lib = C.CDLL(override)
        else:
            build()
            _lib = C.CDLL(_LIB_PATH)
    return _lib


def use_native() -> str:
    """Switches this process to liboracle_native.so: -O3 -march=native (the flags BASELINE.md states for the timed
    CPU baseline), compiled on the host it runs on -- a binary built for this container's CPU must not travel to
    another host.  Falls back to the portable build when the compile fails.  Returns the flags in use."""
    global _lib
    import platform
    import tempfile
    # a private directory (mode 0700, unique name): nothing else can replace the file between the compile and the
    # dlopen, and concurrent ranks / tests never overwrite a library another process has mapped
    tmp = tempfile.mkdtemp(prefix="liboracle_native_")
    out = os.path.join(tmp, f"liboracle_native_{platform.node()}.so")
    try:
        subprocess.run(["g++", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC",
                        "-fvisibility=hidden", "-pthread", "-shared", "-o", out, os.path.join(_HERE, "oracle.cpp")],
                       check=True, capture_output=True)
        _lib = C.CDLL(out)
        return "-O3 -march=native -ffp-contract=off"
    except Exception:
        lib()
        return "-O3 -march=x86-64-v3 -ffp-contract=off (native build failed)"
    finally:
        # the mapping stays valid after the unlink: nothing is left behind in /tmp, however many bench ranks and test
        # processes run on the box
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def _suf(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise TypeError(f"oracle supports float32/float64 only, got {dtype}")


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def get_lower_index(knots, queries) -> np.ndarray:
    """vector_extensions.rs:55-111; -1 marks a NaN query (reference panics)."""
    knots = np.ascontiguousarray(knots)
    dt = knots.dtype
    q = _c(queries, dt).ravel()
    out = np.empty(q.size, dtype=np.int64)
    getattr(lib(), f"oracle_get_lower_index_{_suf(dt)}")(
        _p(knots), C.c_size_t(knots.size), _p(q), C.c_size_t(q.size), _p(out))
    return out


def monotonic_prop(v) -> int:
    v = np.ascontiguousarray(v)
    return getattr(lib(), f"oracle_monotonic_prop_{_suf(v.dtype)}")(_p(v), C.c_size_t(v.size))


def validate1d(x, n_data: int, min_len: int) -> int:
    x = np.ascontiguousarray(x)
    return getattr(lib(), f"oracle_validate1d_{_suf(x.dtype)}")(
        _p(x), C.c_size_t(x.size), C.c_size_t(n_data), C.c_size_t(min_len))


def validate2d(x, y, nx: int, ny: int, min_len: int) -> int:
    x = np.ascontiguousarray(x)
    y = _c(y, x.dtype)
    return getattr(lib(), f"oracle_validate2d_{_suf(x.dtype)}")(
        _p(x), C.c_size_t(x.size), _p(y), C.c_size_t(y.size), C.c_size_t(nx), C.c_size_t(ny),
        C.c_size_t(min_len))


def _data2d(data, n):
    data = np.ascontiguousarray(data)
    L = int(np.prod(data.shape[1:], dtype=np.int64)) if data.ndim > 1 else 1
    return data.reshape(n, L), L


def interp1d_linear(x, data, queries, extrapolate=False, nthreads=1, out=None):
    """Returns (status, fail_idx, out[Q, L]) -- linear.rs:73-98 under interp1d/mod.rs:326-343."""
    x = np.ascontiguousarray(x)
    dt = x.dtype
    d2, L = _data2d(_c(data, dt), x.size)
    q = _c(queries, dt).ravel()
    if out is None:
        out = np.zeros((q.size, L), dtype=dt)
    fail = C.c_size_t(0)
    st = getattr(lib(), f"oracle_interp1d_linear_{_suf(dt)}")(
        _p(x), _p(d2), C.c_size_t(x.size), C.c_size_t(L), C.c_int(int(bool(extrapolate))),
        _p(q), C.c_size_t(q.size), _p(out), C.c_size_t(out.strides[0] // out.itemsize if out.ndim > 1 else L),
        C.c_int(nthreads), C.byref(fail))
    return st, fail.value, out


def cubic_build(x, data, periodic=False, left=(BC_NOT_A_KNOT, 0.0), right=(BC_NOT_A_KNOT, 0.0),
                per_lane=None):
    """Returns (status, a[n-1, L], b[n-1, L]) -- cubic_spline.rs:310-368, 409-721.

    ``per_lane``: optional (lkind[L], lval[L], rkind[L], rval[L]) for
    BoundaryCondition::Individual (cubic_spline.rs:370-403).
    """
    x = np.ascontiguousarray(x)
    dt = x.dtype
    n = x.size
    d2, L = _data2d(_c(data, dt), n)
    a = np.zeros((n - 1, L), dtype=dt)
    b = np.zeros((n - 1, L), dtype=dt)
    if per_lane is None:
        lk = np.array([left[0]], dtype=np.int32)
        lv = np.array([left[1]], dtype=np.float64)
        rk = np.array([right[0]], dtype=np.int32)
        rv = np.array([right[1]], dtype=np.float64)
        pl = 0
    else:
        lk, lv, rk, rv = per_lane
        lk = _c(lk, np.int32).ravel(); rk = _c(rk, np.int32).ravel()
        lv = _c(lv, np.float64).ravel(); rv = _c(rv, np.float64).ravel()
        assert lk.size == lv.size == rk.size == rv.size == L
        pl = 1
    st = getattr(lib(), f"oracle_cubic_build_{_suf(dt)}")(
        _p(x), _p(d2), C.c_size_t(n), C.c_size_t(L), C.c_int(int(bool(periodic))),
        _p(lk), _p(lv), _p(rk), _p(rv), C.c_int(pl), _p(a), _p(b))
    return st, a, b


def interp1d_cubic(x, data, a, b, queries, extrapolate=EXTRAPOLATE_NO, nthreads=1, out=None):
    """Returns (status, fail_idx, out[Q, L]) -- cubic_spline.rs:791-830."""
    x = np.ascontiguousarray(x)
    dt = x.dtype
    d2, L = _data2d(_c(data, dt), x.size)
    a = _c(a, dt).reshape(x.size - 1, L)
    b = _c(b, dt).reshape(x.size - 1, L)
    q = _c(queries, dt).ravel()
    if out is None:
        out = np.zeros((q.size, L), dtype=dt)
    fail = C.c_size_t(0)
    st = getattr(lib(), f"oracle_interp1d_cubic_{_suf(dt)}")(
        _p(x), _p(d2), _p(a), _p(b), C.c_size_t(x.size), C.c_size_t(L), C.c_int(int(extrapolate)),
        _p(q), C.c_size_t(q.size), _p(out), C.c_size_t(out.strides[0] // out.itemsize if out.ndim > 1 else L),
        C.c_int(nthreads), C.byref(fail))
    return st, fail.value, out


def interp2d_bilinear(x, y, data, qx, qy, extrapolate=False, nthreads=1, out=None):
    """Returns (status, fail_idx, fail_axis, out[Q, C]) -- bilinear.rs:64-99."""
    x = np.ascontiguousarray(x)
    dt = x.dtype
    y = _c(y, dt)
    data = _c(data, dt)
    nx, ny = x.size, y.size
    Cc = int(np.prod(data.shape[2:], dtype=np.int64)) if data.ndim > 2 else 1
    d3 = data.reshape(nx, ny, Cc)
    qx = _c(qx, dt).ravel()
    qy = _c(qy, dt).ravel()
    assert qx.size == qy.size
    if out is None:
        out = np.zeros((qx.size, Cc), dtype=dt)
    fail = C.c_size_t(0)
    axis = C.c_int(0)
    st = getattr(lib(), f"oracle_interp2d_bilinear_{_suf(dt)}")(
        _p(x), _p(y), _p(d3), C.c_size_t(nx), C.c_size_t(ny), C.c_size_t(Cc),
        C.c_int(int(bool(extrapolate))), _p(qx), _p(qy), C.c_size_t(qx.size), _p(out),
        C.c_size_t(out.strides[0] // out.itemsize if out.ndim > 1 else Cc), C.c_int(nthreads),
        C.byref(fail), C.byref(axis))
    return st, fail.value, axis.value, out
