"""Pins the CPU oracle (oracle/oracle.cpp) against
 (1) every literal vector the reference's own tests/doctests hold for the hot path
     (tests/golden/reference_vectors.json), at the reference's own tolerances, and
 (2) full-precision scipy.interpolate.CubicSpline goldens (tests/golden/scipy_cubic.npz), at the
     1e-10 bar of BASELINE.json.
CPU only."""
import numpy as np
import pytest

import oracle
from conftest import assert_rel, tofloat

EPS = np.finfo(np.float64).eps
DT = {"f64": np.float64, "f32": np.float32}


def _build_cubic(case, dt):
    x = np.array(case["x"], dtype=dt)
    data = np.array(case["data"], dtype=dt)
    if "per_lane" in case:
        pl = case["per_lane"]
        st, a, b = oracle.cubic_build(x, data, per_lane=(pl["lkind"], pl["lval"], pl["rkind"], pl["rval"]))
    else:
        st, a, b = oracle.cubic_build(x, data, periodic=case["periodic"], left=tuple(case["left"]),
                                      right=tuple(case["right"]))
    return st, x, data, a, b


def test_cubic_reference_vectors(refvec):
    for case in refvec["cubic"]:
        dt = DT[case["dtype"]]
        st, x, data, a, b = _build_cubic(case, dt)
        assert st == oracle.OK, case["name"]
        ext = oracle.EXTRAPOLATE_NO
        if case["extrapolate"]:
            ext = oracle.EXTRAPOLATE_PERIODIC if case["periodic"] else oracle.EXTRAPOLATE_YES
        st, fail, out = oracle.interp1d_cubic(x, data, a, b, np.array(case["q"], dtype=dt), ext)
        assert st == oracle.OK, case["name"]
        assert_rel(out, np.array(case["expect"]), case["atol"], case["rtol"], case["name"])


def test_cubic_tightest_doctest_bitexact(refvec):
    # cubic_spline.rs:62-82 -- the one spline vector the reference asserts at abs f64::EPSILON.
    case = refvec["cubic"][0]
    st, x, data, a, b = _build_cubic(case, np.float64)
    st, fail, out = oracle.interp1d_cubic(x, data, a, b, np.array(case["q"]), oracle.EXTRAPOLATE_NO)
    exp = np.array(case["expect"])
    assert np.max(np.abs(out - exp)) <= EPS
    # 9 of 10 are reproduced to the bit; the remaining one (a value ~ -5.6e-17) within EPS
    assert np.count_nonzero(out != exp) <= 1


def test_cubic_errors(refvec):
    for case in refvec["cubic_errors"]:
        if "n_data" in case:
            st = oracle.validate1d(np.array(case["x"]), case["n_data"], case["min_len"])
        else:
            st, _, _ = oracle.cubic_build(np.array(case["x"]), np.array(case["data"]), periodic=case["periodic"])
        assert oracle.STATUS_NAMES[st] == case["expect_status"], case["name"]
    for case in refvec["cubic_oob"]:
        x = np.array(case["x"]); data = np.array(case["data"])
        st, a, b = oracle.cubic_build(x, data)
        st, fail, _ = oracle.interp1d_cubic(x, data, a, b, np.array(case["q"]))
        assert st == oracle.OUT_OF_BOUNDS and fail == case["fail_idx"]


def test_linear_reference_vectors(refvec):
    for case in refvec["linear"]:
        x = np.array(case["x"]); data = np.array(case["data"])
        st, fail, out = oracle.interp1d_linear(x, data, np.array(case["q"]), case["extrapolate"])
        assert st == oracle.OK, case["name"]
        exp = np.array(case["expect"])
        if "atol" in case:
            assert np.max(np.abs(out - exp)) <= case["atol"], case["name"]
        else:  # the reference asserts these with assert_eq!
            assert np.array_equal(out, exp), case["name"]
    for case in refvec["linear_oob"]:
        st, fail, _ = oracle.interp1d_linear(np.array(case["x"]), np.array(case["data"]), np.array(case["q"]))
        assert st == oracle.OUT_OF_BOUNDS and fail == case["fail_idx"], case["name"]


def test_linear_first_error_semantics():
    # interp1d/mod.rs:326-343: stops at the first Err; rows before it stay written, later rows untouched.
    x = np.arange(4.0); data = np.arange(8.0).reshape(4, 2)
    out = np.full((5, 2), -7.0)
    st, fail, out = oracle.interp1d_linear(x, data, np.array([0.5, 1.5, 9.0, 2.5, -1.0]), out=out)
    assert st == oracle.OUT_OF_BOUNDS and fail == 2
    assert np.array_equal(out[:2], [[1.0, 2.0], [3.0, 4.0]])
    assert np.all(out[2:] == -7.0)


def test_builder_errors(refvec):
    for case in refvec["builder1d_errors"]:
        st = oracle.validate1d(np.array(case["x"]), case["n_data"], case["min_len"])
        assert oracle.STATUS_NAMES[st] == case["expect_status"], case["name"]
    for case in refvec["builder2d_errors"]:
        st = oracle.validate2d(np.array(case["x"]), np.array(case["y"]), case["nx"], case["ny"], 2)
        assert oracle.STATUS_NAMES[st] == case["expect_status"], case["name"]


def test_bilinear_reference_vectors(refvec):
    for case in refvec["bilinear"]:
        x = np.array(case["x"]); y = np.array(case["y"]); data = np.array(case["data"])
        st, fail, axis, out = oracle.interp2d_bilinear(x, y, data, np.array(case["qx"]), np.array(case["qy"]),
                                                       case["extrapolate"])
        assert st == oracle.OK, case["name"]
        exp = np.array(case["expect"])
        if "atol" in case:
            assert np.max(np.abs(out - exp)) <= case["atol"], case["name"]
        else:
            assert np.array_equal(out, exp), case["name"]
    for case in refvec["bilinear_oob"]:
        st, fail, axis, _ = oracle.interp2d_bilinear(
            np.array(case["x"]), np.array(case["y"]), np.array(case["data"]), np.array(case["qx"]),
            np.array(case["qy"]))
        assert st == oracle.OUT_OF_BOUNDS and fail == case["fail_idx"] and axis == case["fail_axis"], case["name"]


def test_bilinear_11x11_bitexact(refvec):
    # tests/interp2d.rs:85-238 is asserted at abs f64::EPSILON; the restatement reproduces it to the bit.
    case = [c for c in refvec["bilinear"] if c["name"] == "interpolate_array_11x11"][0]
    st, fail, axis, out = oracle.interp2d_bilinear(
        np.array(case["x"]), np.array(case["y"]), np.array(case["data"]), np.array(case["qx"]), np.array(case["qy"]))
    assert np.array_equal(out, np.array(case["expect"]))


def test_get_lower_index_tables(refvec):
    for case in refvec["get_lower_index"]:
        got = oracle.get_lower_index(np.array(case["knots"]), np.array(tofloat(case["q"])))
        assert got.tolist() == case["expect"], case["name"]


def test_get_lower_index_is_upper_bound_minus_one():
    # the net result is the unique i with k[i] <= x < k[i+1], clamped to [0, n-2].
    rng = np.random.default_rng(3)
    for n in (2, 3, 11, 100, 1024, 4096):
        for k in (np.linspace(0, 1, n), np.sort(rng.uniform(0, 1, n)), np.logspace(-3, 0, n)):
            k = np.unique(k)
            m = k.size
            q = np.concatenate([rng.uniform(k[0] - 0.1, k[-1] + 0.1, 2000), k, np.nextafter(k, -np.inf),
                                np.nextafter(k, np.inf)])
            got = oracle.get_lower_index(k, q)
            exp = np.clip(np.searchsorted(k, q, side="right") - 1, 0, m - 2)
            assert np.array_equal(got, exp)
            for dt in (np.float32,):
                k32 = np.unique(k.astype(dt)); q32 = q.astype(dt)
                got = oracle.get_lower_index(k32, q32)
                exp = np.clip(np.searchsorted(k32, q32, side="right") - 1, 0, k32.size - 2)
                assert np.array_equal(got, exp)


def test_monotonic_prop(refvec):
    for case in refvec["monotonic_prop"]:
        got = oracle.MONO_NAMES[oracle.monotonic_prop(np.array(case["v"], dtype=np.float64))]
        assert got == case["expect"], case
    assert oracle.MONO_NAMES[oracle.monotonic_prop(np.array([0.0, np.nan, 2.0]))] != "Rising{strict:true}"


def _dense_reference_system(x, y, left, right):
    """Second, independent restatement (dense numpy) of the non-periodic system of
    cubic_spline.rs:431-471, 597-670, solved by LU with pivoting instead of Thomas."""
    n = x.size
    A = np.zeros((n, n)); rhs = np.zeros_like(y)
    dx = np.diff(x)
    for i in range(1, n - 1):
        A[i, i - 1] = dx[i]; A[i, i] = 2 * (dx[i] + dx[i - 1]); A[i, i + 1] = dx[i - 1]
        rhs[i] = 3 * (dx[i] * (y[i] - y[i - 1]) / dx[i - 1] + dx[i - 1] * (y[i + 1] - y[i]) / dx[i])
    def spec(k, v):
        return (4, 0.0) if k == 1 else (3, 0.0) if k == 2 else (k, v)
    lk, lv = spec(*left); rk, rv = spec(*right)
    if lk == 0:
        d = x[2] - x[0]
        A[0, 0] = dx[1]; A[0, 1] = d
        rhs[0] = ((dx[0] + 2 * d) * dx[1] * (y[1] - y[0]) / dx[0] + dx[0] ** 2 * (y[2] - y[1]) / dx[1]) / d
    elif lk == 3:
        A[0, 0] = 1.0; rhs[0] = lv
    else:
        A[0, 0] = 2 * dx[0]; A[0, 1] = dx[0]; rhs[0] = 3 * (y[1] - y[0]) - lv * dx[0] ** 2 / 2
    if rk == 0:
        d = x[-1] - x[-3]
        A[-1, -1] = dx[-1]   # cubic_spline.rs:635 (scipy has dx[-2] here; see gen_scipy_golden.py)
        A[-1, -2] = d
        rhs[-1] = (dx[-1] ** 2 * (y[-2] - y[-3]) / dx[-2] + (2 * d + dx[-1]) * dx[-2] * (y[-1] - y[-2]) / dx[-1]) / d
    elif rk == 3:
        A[-1, -1] = 1.0; rhs[-1] = rv
    else:
        A[-1, -1] = 2 * dx[-1]; A[-1, -2] = dx[-1]; rhs[-1] = 3 * (y[-1] - y[-2]) + rv * dx[-1] ** 2 / 2
    k = np.linalg.solve(A, rhs)
    a = k[:-1] * dx[:, None] - (y[1:] - y[:-1])
    b = (y[1:] - y[:-1]) - k[1:] * dx[:, None]
    return a, b


@pytest.mark.parametrize("prefix", ["jit5", "log5", "jit64", "log64", "jit1024", "log1024"])
def test_cubic_coefficients_vs_dense_solve(scipy_golden, prefix):
    g = scipy_golden
    for name in [n for n in g["names"].tolist() if n.split("_")[0] == prefix and not n.endswith("_per")]:
        x = g[name + "/x"]; y = g[name + "/y"]
        per, lk, lv, rk, rv = g[name + "/bc"].tolist()
        st, a, b = oracle.cubic_build(x, y, left=(int(lk), lv), right=(int(rk), rv))
        a2, b2 = _dense_reference_system(x, y, (int(lk), lv), (int(rk), rv))
        scale = max(np.max(np.abs(a2)), np.max(np.abs(b2)))
        assert_rel(a, a2, 1e-10 * scale, 1e-10, name + " a")
        assert_rel(b, b2, 1e-10 * scale, 1e-10, name + " b")


@pytest.mark.parametrize("prefix", ["ref12"] + [g + str(n) for n in (4, 5, 64, 1024, 4096) for g in ("uni", "jit", "log")])
def test_cubic_vs_scipy(scipy_golden, prefix):
    g = scipy_golden
    names = [n for n in g["names"].tolist() if n.split("_")[0] == prefix]
    assert names
    for name in names:
        x = g[name + "/x"]; y = g[name + "/y"]; q = g[name + "/q"]; exp = g[name + "/expect"]
        per, lk, lv, rk, rv = g[name + "/bc"].tolist()
        if name.endswith("_nk") and prefix[:3] in ("jit", "log"):
            # reference deviation on non-uniform grids (cubic_spline.rs:635), pinned by
            # test_cubic_coefficients_vs_dense_solve instead
            continue
        st, a, b = oracle.cubic_build(x, y, periodic=bool(per), left=(int(lk), lv), right=(int(rk), rv))
        assert st == oracle.OK, name
        ext = oracle.EXTRAPOLATE_PERIODIC if per else oracle.EXTRAPOLATE_YES
        st, fail, out = oracle.interp1d_cubic(x, y, a, b, q, ext)
        assert st == oracle.OK, name
        scale = np.max(np.abs(y))
        # 1e-10 bar (BASELINE.json north_star); atol relative to the data scale because splines cross zero
        assert_rel(out, exp, 1e-10 * scale, 1e-10, name)


def test_cubic_f32_vs_f64_oracle():
    # f32 path of the oracle vs its own f64 path at the 1e-5 bar
    rng = np.random.default_rng(5)
    n, L = 257, 5
    x = np.linspace(0, 1, n) + rng.uniform(-0.2 / n, 0.2 / n, n); x.sort()
    y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], 500)
    st, a, b = oracle.cubic_build(x, y)
    _, _, o64 = oracle.interp1d_cubic(x, y, a, b, q)
    x32, y32, q32 = x.astype(np.float32), y.astype(np.float32), q.astype(np.float32)
    st, a32, b32 = oracle.cubic_build(x32, y32)
    _, _, o32 = oracle.interp1d_cubic(x32, y32, a32, b32, q32)
    # compare at the same (f32-rounded) inputs evaluated in f64
    st, a64, b64 = oracle.cubic_build(x32.astype(np.float64), y32.astype(np.float64))
    _, _, o64b = oracle.interp1d_cubic(x32.astype(np.float64), y32.astype(np.float64), a64, b64, q32.astype(np.float64))
    assert_rel(o32, o64b, 2e-4, 2e-4, "f32 vs f64")


def test_multithread_blocks_match_serial():
    rng = np.random.default_rng(11)
    n, L, Q = 64, 7, 1001
    x = np.sort(rng.uniform(0, 1, n)); y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
    st, a, b = oracle.cubic_build(x, y)
    _, _, o1 = oracle.interp1d_cubic(x, y, a, b, q, nthreads=1)
    _, _, o4 = oracle.interp1d_cubic(x, y, a, b, q, nthreads=4)
    assert np.array_equal(o1, o4)
    q[700] = 5.0
    st, fail, _ = oracle.interp1d_cubic(x, y, a, b, q, nthreads=4)
    assert st == oracle.OUT_OF_BOUNDS and fail == 700
