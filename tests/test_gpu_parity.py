"""GPU parity tests: the HIP path, called through the C ABI (via the host mirror), against the CPU oracle
on the same seeded inputs, against the reference's own golden vectors, and -- at BASELINE.json's full
sizes -- through size-independent properties.

Bar: the device code is compiled without FMA contraction and written in the reference's operation order,
so f32/f64 results are expected to be BIT-IDENTICAL to the oracle; the tests assert exact equality where
that holds and never accept more than the north-star tolerance (1e-10 rel f64, 1e-5 rel f32).
"""
import os

import numpy as np
import pytest

import oracle
from conftest import assert_rel, tofloat

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.float64): 1e-10, np.dtype(np.float32): 1e-5}
DT = {"f64": np.float64, "f32": np.float32}


def knots(kind, n, rng, dt):
    if kind == "lin":
        x = np.linspace(0.0, 1.0, n)
    elif kind == "rand":   # sorted-unique uniform, cf. benches/rand_extensions.rs:26-35
        x = np.unique(rng.uniform(0.0, 1.0, 4 * n).astype(dt))[:n].astype(np.float64)
        x = np.sort(x)
    elif kind == "jit":
        x = np.sort(np.linspace(0.0, 1.0, n) + rng.uniform(-0.2 / n, 0.2 / n, n))
    else:
        x = np.logspace(-2, 0, n)
    x = np.unique(x.astype(dt))
    assert x.size == n, (kind, n, x.size)
    return x


def check_equal(got, ref, what):
    """bit-identical expected; report the worst relative error if not"""
    got = np.asarray(got); ref = np.asarray(ref)
    if np.array_equal(got, ref, equal_nan=True):
        return
    tol = TOL[ref.dtype]
    scale = np.max(np.abs(ref)) if ref.size else 1.0
    assert_rel(got, ref, tol * scale, tol, what + " (not bit-identical, checking tolerance)")
    raise AssertionError(f"{what}: within tolerance but not bit-identical "
                         f"(max abs diff {np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64))):.3e})")


def blocked_build(n, L):
    """Shapes whose CubicSpline build takes the blocked Thomas sweeps by default (ndinterp_api.hip, build_spline):
    narrow trailing axes on many knots.  The one path that is not bit-identical to the reference order."""
    return n >= 2048 and L <= 256 and os.environ.get("NDI_SPLINE_BLOCKED", "-1") != "0"


def table_close(got, ref, dt, what):
    """north-star bar for a coefficient table: |got - ref| <= rtol * max|ref| (1e-10 f64, 1e-5 f32)"""
    got = np.asarray(got, dtype=np.float64).reshape(-1); ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    scale = np.max(np.abs(ref)) if ref.size else 1.0
    err = np.max(np.abs(got - ref)) if ref.size else 0.0
    assert err <= TOL[np.dtype(dt)] * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e}"


# ------------------------------------------------------------------------------------------------
# the reference's own golden vectors through the device path
# ------------------------------------------------------------------------------------------------
def _bc_from_case(pkg, case):
    S, R, B = pkg.SingleBoundary, pkg.RowBoundary, pkg.BoundaryCondition
    def single(kind, val):
        return {0: S.NotAKnot, 1: S.Natural, 2: S.Clamped}.get(kind) or \
            (S.FirstDeriv(val) if kind == 3 else S.SecondDeriv(val))
    if "per_lane" in case:
        pl = case["per_lane"]
        rows = [R.Mixed(single(lk, lv), single(rk, rv))
                for lk, lv, rk, rv in zip(pl["lkind"], pl["lval"], pl["rkind"], pl["rval"])]
        shape = (1,) + tuple(np.array(case["data"]).shape[1:])
        arr = np.empty(len(rows), dtype=object)
        arr[:] = rows
        return B.Individual(arr.reshape(shape))
    if case["periodic"]:
        return B.Periodic
    return {0: B.NotAKnot, 1: B.Natural, 2: B.Clamped}[case["left"][0]]


def test_reference_cubic_vectors(pkg, refvec):
    for case in refvec["cubic"]:
        dt = DT[case["dtype"]]
        x = np.array(case["x"], dtype=dt); data = np.array(case["data"], dtype=dt)
        strat = pkg.CubicSpline.new().extrapolate(case["extrapolate"]).boundary(_bc_from_case(pkg, case))
        interp = pkg.Interp1DBuilder.new(data).x(x).strategy(strat).build()
        res = interp.interp_array(np.array(case["q"], dtype=dt))
        assert_rel(res, np.array(case["expect"]), case["atol"], case["rtol"], case["name"])


def test_reference_linear_vectors(pkg, refvec):
    for case in refvec["linear"]:
        x = np.array(case["x"]); data = np.array(case["data"])
        if data.shape[1] == 1:
            data = data[:, 0]
        interp = pkg.Interp1DBuilder.new(data).x(x).strategy(pkg.Linear.new().extrapolate(case["extrapolate"])).build()
        q = np.array(case["q"]).reshape(case.get("q_shape", [len(case["q"])]))
        res = interp.interp_array(q)
        exp = np.array(case["expect"]).reshape(res.shape)
        if "atol" in case:
            assert np.max(np.abs(res - exp)) <= case["atol"], case["name"]
        else:
            assert np.array_equal(res, exp), case["name"]
        # single-point entries agree with the batch (interp1d/mod.rs:108-175)
        if data.ndim == 1:
            assert interp.interp_scalar(case["q"][0]) == res.reshape(-1)[0]
        else:
            assert np.array_equal(interp.interp(case["q"][0]), res.reshape(-1, data.shape[1])[0])
    for case in refvec["linear_oob"]:
        interp = pkg.Interp1DBuilder.new(np.array(case["data"])[:, 0]).x(np.array(case["x"])).build()
        with pytest.raises(pkg.InterpolateError.OutOfBounds, match="is not in range") as ei:
            interp.interp(case["q"][0])
        assert ei.value.index == case["fail_idx"]


def test_reference_bilinear_vectors(pkg, refvec):
    for case in refvec["bilinear"]:
        data = np.array(case["data"])
        interp = pkg.Interp2DBuilder.new(data).x(np.array(case["x"])).y(np.array(case["y"])).build()
        shape = case.get("q_shape", [len(case["qx"])])
        qx = np.array(case["qx"]).reshape(shape); qy = np.array(case["qy"]).reshape(shape)
        res = interp.interp_array(qx, qy)
        exp = np.array(case["expect"]).reshape(res.shape)
        if "atol" in case:
            assert np.max(np.abs(res - exp)) <= case["atol"], case["name"]
        else:
            assert np.array_equal(res, exp), case["name"]
    for case in refvec["bilinear_oob"]:
        interp = pkg.Interp2D.builder(np.array(case["data"])).build()
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            interp.interp(case["qx"][0], case["qy"][0])
        assert ei.value.axis == case["fail_axis"]
        assert str(ei.value).startswith("x = " if case["fail_axis"] == 0 else "y = ")


def test_bilinear_11x11_bit_exact(pkg, refvec):
    case = [c for c in refvec["bilinear"] if c["name"] == "interpolate_array_11x11"][0]
    interp = pkg.Interp2DBuilder.new(np.array(case["data"])).x(np.array(case["x"])).y(np.array(case["y"])).build()
    res = interp.interp_array(np.array(case["qx"]).reshape(11, 11), np.array(case["qy"]).reshape(11, 11))
    assert np.array_equal(res.reshape(-1), np.array(case["expect"]).reshape(-1))


def test_cubic_doctest_vector_eps(pkg, refvec):
    case = refvec["cubic"][0]   # cubic_spline.rs:62-82, asserted at abs f64::EPSILON
    interp = pkg.Interp1DBuilder.new(np.array(case["data"])[:, 0]).strategy(pkg.CubicSpline.new()) \
        .x(np.array(case["x"])).build()
    res = interp.interp_array(np.array(case["q"]))
    assert np.max(np.abs(res - np.array(case["expect"])[:, 0])) <= np.finfo(np.float64).eps


def test_custom_strategy_example(pkg, refvec):
    # examples/custom_strategy.rs: a user strategy that only implements the per-query hook
    class StepInterpolator(pkg.Interp1DStrategyBuilder, pkg.Interp1DStrategy):
        MINIMUM_DATA_LENGHT = 2

        def build(self, x, data):
            return self

        def interp_into(self, interpolator, target, x):
            idx = interpolator.get_index_left_of(x)
            x_left, d_left = interpolator.index_point(idx)
            x_right, d_right = interpolator.index_point(idx + 1)
            target[...] = d_left if (x_right - x_left) / 2.0 > (x - x_left) else d_right

    c = refvec["custom_strategy"]
    interp = pkg.Interp1D.builder(np.array(c["data"])).strategy(StepInterpolator()).build()
    assert np.array_equal(interp.interp_array(np.array(c["q"])), np.array(c["expect"]))


# ------------------------------------------------------------------------------------------------
# get_lower_index (vector_extensions.rs:55-111)
# ------------------------------------------------------------------------------------------------
def test_get_lower_index_reference_tables(pkg, refvec):
    for case in refvec["get_lower_index"]:
        got = pkg.get_lower_index(np.array(case["knots"]), np.array(tofloat(case["q"])))
        assert got.tolist() == case["expect"], case["name"]


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n", [2, 3, 64, 65, 100, 1024, 4096, 4097, 8192, 30000, 100000])
def test_get_lower_index_random(pkg, dt, n):
    rng = np.random.default_rng(n)
    for kind in ("lin", "rand", "log"):
        k = knots(kind, n, rng, dt)
        span = float(k[-1] - k[0])
        q = np.concatenate([rng.uniform(k[0] - 0.1 * span, k[-1] + 0.1 * span, 5000).astype(dt), k,
                            np.nextafter(k, dt(-np.inf)), np.nextafter(k, dt(np.inf)),
                            np.array([np.inf, -np.inf], dtype=dt)])
        got = pkg.get_lower_index(k, q)
        exp = np.clip(np.searchsorted(k, q, side="right") - 1, 0, n - 2)
        assert np.array_equal(got, exp), (kind, n)
        assert np.array_equal(got, oracle.get_lower_index(k, q))
    assert pkg.get_lower_index(k, np.array([np.nan], dtype=dt))[0] == -1


# ------------------------------------------------------------------------------------------------
# spline build (cubic_spline.rs:310-368, 409-721) vs oracle: coefficient tables bit-exact
# ------------------------------------------------------------------------------------------------
BCS = {
    "nk": (False, (0, 0.0), (0, 0.0)), "nat": (False, (1, 0.0), (1, 0.0)), "cl": (False, (2, 0.0), (2, 0.0)),
    "d1": (False, (3, -0.1), (3, -0.5)), "d2": (False, (4, -0.1), (4, -0.5)), "mix": (False, (0, 0.0), (3, 0.5)),
    "mix2": (False, (4, 0.3), (0, 0.0)), "per": (True, (0, 0.0), (0, 0.0)),
}


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n,L", [(3, 1), (3, 5), (4, 3), (5, 64), (12, 1), (257, 130), (1024, 7), (4096, 64)])
def test_spline_coefficients_bit_exact(pkg, dt, n, L):
    rng = np.random.default_rng(1000 * n + L)
    x = knots("jit" if n > 4 else "lin", n, rng, dt) if n > 3 else np.array([-1.0, 0.0, 3.0], dtype=dt)
    y = rng.uniform(0.0, 1.0, (n, L)).astype(dt)
    for name, (per, left, right) in BCS.items():
        yy = y.copy()
        if per:
            yy[-1] = yy[0]
        S, R, B = pkg.SingleBoundary, pkg.RowBoundary, pkg.BoundaryCondition
        def single(kind, val):
            return {0: S.NotAKnot, 1: S.Natural, 2: S.Clamped}.get(kind) or \
                (S.FirstDeriv(val) if kind == 3 else S.SecondDeriv(val))
        if per:
            bc = B.Periodic
        else:
            rows = np.empty((1,) + yy.shape[1:], dtype=object)
            for i in range(rows.size):
                rows.reshape(-1)[i] = R.Mixed(single(*left), single(*right))
            bc = B.Individual(rows)
        interp = pkg.Interp1DBuilder.new(yy).x(x).strategy(pkg.CubicSpline.new().boundary(bc)).build()
        a, b = interp.strategy.coefficients()
        st, ra, rb = oracle.cubic_build(x, yy, periodic=per, left=left, right=right)
        assert st == oracle.OK
        if blocked_build(n, L):   # the blocked sweeps: a few ulp of the largest entry, not bit-identical
            table_close(a, ra, dt, f"a[{name}] n={n} L={L} {np.dtype(dt)} (blocked build)")
            table_close(b, rb, dt, f"b[{name}] n={n} L={L} {np.dtype(dt)} (blocked build)")
            continue
        check_equal(a, ra, f"a[{name}] n={n} L={L} {np.dtype(dt)}")
        check_equal(b, rb, f"b[{name}] n={n} L={L} {np.dtype(dt)}")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n,shape", [(3, (7,)), (3, (70,)), (4, (5,)), (12, (3, 4)), (257, (130,)), (1024, (2, 3, 5))])
def test_individual_boundaries_bit_exact(pkg, dt, n, shape):
    """BoundaryCondition::Individual (cubic_spline.rs:332-347, 370-403): every trailing element has its own
    RowBoundary; the device solve must equal one scalar solve per column (what the oracle does)."""
    rng = np.random.default_rng(n * 31 + len(shape))
    L = int(np.prod(shape))
    x = knots("jit", n, rng, dt) if n > 3 else np.array([-1.0, 0.0, 3.0], dtype=dt)
    y = rng.uniform(0.0, 1.0, (n,) + shape).astype(dt)
    S, R, B = pkg.SingleBoundary, pkg.RowBoundary, pkg.BoundaryCondition
    lk = rng.integers(0, 5, L); rk = rng.integers(0, 5, L)
    if n == 3:
        lk[:3] = 0; rk[:3] = 0          # some NotAKnot/NotAKnot columns -> parabola rows (:569-596)
    lv = np.where(lk >= 3, rng.uniform(-1, 1, L), 0.0); rv = np.where(rk >= 3, rng.uniform(-1, 1, L), 0.0)
    def single(kind, val):
        return {0: S.NotAKnot, 1: S.Natural, 2: S.Clamped}.get(int(kind)) or \
            (S.FirstDeriv(val) if kind == 3 else S.SecondDeriv(val))
    rows = np.empty(L, dtype=object)
    for i in range(L):
        # exercise the shorthand rows too
        if lk[i] == rk[i] and lk[i] < 3:
            rows[i] = [R.NotAKnot, R.Natural, R.Clamped][lk[i]]
        else:
            rows[i] = R.Mixed(single(lk[i], lv[i]), single(rk[i], rv[i]))
    strat = pkg.CubicSpline.new().extrapolate(True).boundary(B.Individual(rows.reshape((1,) + shape)))
    interp = pkg.Interp1DBuilder.new(y).x(x).strategy(strat).build()
    a, b = interp.strategy.coefficients()
    st, ra, rb = oracle.cubic_build(x, y, per_lane=(lk, lv, rk, rv))
    assert st == oracle.OK
    check_equal(a, ra, f"a individual n={n}")
    check_equal(b, rb, f"b individual n={n}")
    q = rng.uniform(x[0] - 0.1, x[-1] + 0.1, 300).astype(dt)
    _, _, ref = oracle.interp1d_cubic(x, y, ra, rb, q, oracle.EXTRAPOLATE_YES)
    check_equal(interp.interp_array(q).reshape(300, L), ref, "eval individual")


def test_periodic_value_error(pkg):
    y = np.array([[0.5, 1.0], [0.0, 1.5], [0.5, 1.1]])
    with pytest.raises(pkg.BuilderError.ValueError, match="first and last value must be equal") as ei:
        pkg.Interp1DBuilder.new(y).strategy(pkg.CubicSpline.new().boundary(pkg.BoundaryCondition.Periodic)).build()
    # the text tests/cubic_spline_strat.rs:442-452 expects
    assert "First: [0.5, 1.0], shape=[2], strides=[1], layout=CFcf (0xf), const ndim=1, last: [0.5, 1.1]" in str(ei.value)
    with pytest.raises(pkg.BuilderError.ValueError, match="First: 1.0, last: 2.0"):
        pkg.Interp1DBuilder.new(np.array([1.0, 0.0, 2.0])).strategy(
            pkg.CubicSpline.new().boundary(pkg.BoundaryCondition.Periodic)).build()
    y4 = np.array([[0.5, 1.0], [0.0, 1.5], [0.2, 0.1], [0.5, 1.1]])
    with pytest.raises(pkg.BuilderError.ValueError):
        pkg.Interp1DBuilder.new(y4).strategy(pkg.CubicSpline.new().boundary(pkg.BoundaryCondition.Periodic)).build()


# ------------------------------------------------------------------------------------------------
# 1-D evaluation vs oracle: every kernel variant, both formulations
# ------------------------------------------------------------------------------------------------
SHAPES_1D = [  # (n, L, Q): covers flat VEC=1 / flat vector / rows U=1,2,4 and ragged tails
    (5, 1, 1000), (100, 1, 10000), (1024, 1, 10000), (64, 3, 777), (64, 4, 1000), (33, 6, 501), (17, 62, 300),
    (300, 512, 2000), (300, 514, 1500), (129, 1024, 3000), (100, 1026, 1000), (257, 2048, 4099), (64, 4096, 2500),
    (40, 6144, 700), (9, 8200, 300),
    # very long rows: 65536 / 65540 lanes = 16 whole 256 x 8-vector segments per row (f64; f32: 8) and a
    # ragged one -- gridDim.y walks the segments of eval_rows_kernel / eval_bucketed_kernel; 131072 lanes > the 64-segment
    # cap of gridDim.y for f64 U = 8 (the kernels stride over the segments)
    (7, 65536, 120), (6, 65540, 90), (5, 131072 + 64, 40),
]


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n,L,Q", SHAPES_1D)
def test_cubic_eval_bit_exact(pkg, dt, n, L, Q):
    rng = np.random.default_rng(n * 7919 + L)
    x = knots("rand", n, rng, dt)
    y = rng.uniform(0.0, 1.0, (n, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    q[:3] = [x[0], x[-1], x[n // 2]]
    interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED, pkg.PATH_AUTO):
        interp.strategy.path = path
        check_equal(interp.interp_array(q), ref, f"cubic n={n} L={L} path={path}")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n,Q", [(10000, 60000), (20000, 50000), (70000, 30000)])
def test_cubic_many_knots_grouping_variants(pkg, dt, n, Q):
    """Grouping back ends of the bucketed formulation: 10000 knots = block-local sort with a large LDS
    histogram next to the staged pyramid; 20000 = more intervals than the LDS histogram holds -> global-atomic
    histogram + placement; 70000 = more knots than the u16 bucket index covers and a pyramid too large for LDS
    (560 KB of f64 knots), searched in global memory."""
    rng = np.random.default_rng(n)
    L = 512 if dt == np.float64 else 1024
    x = knots("jit", n, rng, dt)
    y = rng.uniform(0.0, 1.0, (n, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    for path in (pkg.PATH_BUCKETED, pkg.PATH_GATHER):
        interp.strategy.path = path
        check_equal(interp.interp_array(q), ref, f"cubic n={n} path={path}")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n,L,Q", SHAPES_1D)
def test_linear_eval_bit_exact(pkg, dt, n, L, Q):
    rng = np.random.default_rng(n * 104729 + L)
    x = knots("rand", n, rng, dt)
    y = rng.uniform(0.0, 1.0, (n, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    interp = pkg.Interp1DBuilder.new(y).x(x).build()
    _, _, ref = oracle.interp1d_linear(x, y, q)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        interp.strategy.path = path
        check_equal(interp.interp_array(q), ref, f"linear n={n} L={L} path={path}")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_extrapolation_modes(pkg, dt):
    rng = np.random.default_rng(77)
    n, L, Q = 50, 260 * (2 if dt == np.float64 else 4), 3000
    x = knots("jit", n, rng, dt)
    y = rng.uniform(0.0, 1.0, (n, L)).astype(dt)
    span = x[-1] - x[0]
    q = rng.uniform(x[0] - 2.5 * span, x[-1] + 2.5 * span, Q).astype(dt)
    # Linear / CubicSpline extrapolate(true): end interval polynomial
    lin = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.Linear.new().extrapolate(True)).build()
    _, _, ref = oracle.interp1d_linear(x, y, q, True)
    check_equal(lin.interp_array(q), ref, "linear extrapolate")
    cub = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new().extrapolate(True)).build()
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q, oracle.EXTRAPOLATE_YES)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        cub.strategy.path = path
        check_equal(cub.interp_array(q), ref, "cubic extrapolate")
    # Periodic + extrapolate(true) wraps; Periodic + extrapolate(false) errors (cubic_spline.rs:763-769)
    yp = y.copy(); yp[-1] = yp[0]
    per = pkg.Interp1DBuilder.new(yp).x(x).strategy(
        pkg.CubicSpline.new().extrapolate(True).boundary(pkg.BoundaryCondition.Periodic)).build()
    st, a, b = oracle.cubic_build(x, yp, periodic=True)
    _, _, ref = oracle.interp1d_cubic(x, yp, a, b, q, oracle.EXTRAPOLATE_PERIODIC)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        per.strategy.path = path
        check_equal(per.interp_array(q), ref, "cubic periodic wrap")
    per_no = pkg.Interp1DBuilder.new(yp).x(x).strategy(
        pkg.CubicSpline.new().boundary(pkg.BoundaryCondition.Periodic)).build()
    with pytest.raises(pkg.InterpolateError.OutOfBounds):
        per_no.interp_array(q)


def test_first_error_semantics_and_messages(pkg):
    # interp1d/mod.rs:334-342: stop at the first Err; rows before it are written, later rows untouched
    rng = np.random.default_rng(5)
    n, L, Q = 20, 1024, 600
    x = np.arange(n, dtype=np.float64); y = rng.uniform(0, 1, (n, L))
    q = rng.uniform(0, n - 1, Q); q[317] = -0.1; q[500] = 99.0
    for strat, orc in ((pkg.Linear.new(), None), (pkg.CubicSpline.new(), "c")):
        interp = pkg.Interp1DBuilder.new(y).x(x).strategy(strat).build()
        for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
            interp.strategy.path = path
            buf = np.full((Q, L), -7.0)
            with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
                interp.interp_array_into(q, buf)
            assert ei.value.index == 317 and str(ei.value) == "x = -0.1 is not in range"
            if orc is None:
                _, _, ref = oracle.interp1d_linear(x, y, q[:317])
            else:
                st, a, b = oracle.cubic_build(x, y)
                _, _, ref = oracle.interp1d_cubic(x, y, a, b, q[:317])
            assert np.array_equal(buf[:317], ref)
            assert np.all(buf[317:] == -7.0)
    # NaN: not in range -> OutOfBounds without extrapolation; panic with extrapolation
    q2 = q.copy(); q2[317] = 1.0; q2[500] = 2.0; q2[44] = np.nan
    interp = pkg.Interp1DBuilder.new(y).x(x).build()
    with pytest.raises(pkg.InterpolateError.OutOfBounds, match="x = NaN is not in range") as ei:
        interp.interp_array(q2)
    assert ei.value.index == 44
    interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.Linear.new().extrapolate(True)).build()
    with pytest.raises(pkg.Panic, match="failed to convert NaN to usize"):
        interp.interp_array(q2)


def test_query_rank_and_buffer_shapes(pkg):
    # output shape = query shape ++ data.shape[1..] for any ranks (interp1d/mod.rs:508-537, 549-607)
    rng = np.random.default_rng(9)
    for rank in range(1, 8):       # interp1d/mod.rs:531-537 goes to rank 7 (IxDyn)
        data = rng.uniform(0, 1, (4,) * rank)
        interp = pkg.Interp1D.builder(data).build()
        res = interp.interp(2.2)
        assert res.ndim == rank - 1
        buf = np.zeros(res.shape); interp.interp_into(2.2, buf)
        assert np.array_equal(buf, res)
        query = np.array([[0.5, 1.0], [1.5, 2.0]])
        res = interp.interp_array(query)
        assert res.shape == (2, 2) + data.shape[1:]
        _, _, ref = oracle.interp1d_linear(np.arange(4.0), data, query)
        assert np.array_equal(res.reshape(4, -1), ref)
        buf = np.zeros(res.shape); interp.interp_array_into(query, buf)
        assert np.array_equal(buf, res)
    interp = pkg.Interp1D.builder(rng.uniform(0, 1, (4, 4))).build()
    with pytest.raises(pkg.Panic, match=r"expected: \[4\], got: \[3\]"):
        interp.interp_into(2.2, np.zeros(3))
    for bad in ((1, 4), (2, 3), (3, 4), (2, 5)):
        with pytest.raises(pkg.Panic):
            interp.interp_array_into(np.array([2.2, 2.4]), np.zeros(bad))
    # strided (non-contiguous) output view and negative-stride data view (tests/interp1d.rs:143-155)
    big = np.zeros((2, 8)); view = big[:, ::2]
    interp.interp_array_into(np.array([2.2, 2.4]), view)
    assert np.array_equal(view, interp.interp_array(np.array([2.2, 2.4])))
    a = np.arange(1.0, 11.0)
    rv = pkg.Interp1D.builder(a[::-1]).x(np.arange(-4.0, 6.0)).build()
    assert [rv.interp_scalar(v) for v in (-4.0, 5.0, 0.0, -3.5, 4.75)] == [10.0, 1.0, 6.0, 9.5, 1.25]


def test_device_tensors_strides_and_async(pkg):
    import torch
    rng = np.random.default_rng(21)
    n, L, Q = 100, 2048, 5000
    x = knots("rand", n, rng, np.float64); y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    dev = torch.device("cuda:0")
    interp = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
        .strategy(pkg.CubicSpline.new()).build()
    qd = torch.as_tensor(q, device=dev)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        interp.strategy.path = path
        out = interp.interp_array(qd)
        assert out.is_cuda and np.array_equal(out.cpu().numpy(), ref)
        # row stride > lanes through the raw hook
        wide = torch.full((Q, L + 64), -1.0, dtype=torch.float64, device=dev)
        interp.strategy.interp_array_into(interp, qd, wide[:, :L])
        w = wide.cpu().numpy()
        assert np.array_equal(w[:, :L], ref) and np.all(w[:, L:] == -1.0)
        # async launch + finish on the current stream
        out2 = torch.empty((Q, L), dtype=torch.float64, device=dev)
        interp.strategy.interp_array_into(interp, qd, out2, async_launch=True)
        interp.strategy.finish()
        assert np.array_equal(out2.cpu().numpy(), ref)
        bad = qd.clone(); bad[1234] = 7.0
        interp.strategy.interp_array_into(interp, bad, out2, async_launch=True)
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            interp.strategy.finish()
        assert ei.value.index == 1234
    # host queries / host output in chunks smaller than the batch is covered by the numpy tests


# ------------------------------------------------------------------------------------------------
# 2-D bilinear vs oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("nx,ny,C,Q", [(2, 2, 1, 500), (7, 5, 1, 4000), (33, 17, 3, 3000), (64, 48, 4, 5000),
                                        (50, 70, 16, 6000), (40, 30, 64, 5000), (9, 11, 130, 700),
                                        (6, 5, 1024, 300)])
def test_bilinear_bit_exact(pkg, dt, nx, ny, C, Q):
    rng = np.random.default_rng(nx * 131 + ny * 17 + C)
    x = knots("rand", nx, rng, dt) if nx > 2 else np.array([0.0, 1.0], dtype=dt)
    y = knots("jit", ny, rng, dt) if ny > 2 else np.array([-1.0, 1.0], dtype=dt)
    g = rng.uniform(0, 1, (nx, ny, C)).astype(dt)
    qx = rng.uniform(x[0], x[-1], Q).astype(dt); qy = rng.uniform(y[0], y[-1], Q).astype(dt)
    qx[:2] = [x[0], x[-1]]; qy[:2] = [y[-1], y[0]]
    interp = pkg.Interp2DBuilder.new(g).x(x).y(y).build()
    _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx, qy)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED, pkg.PATH_AUTO):   # BUCKETED = tile-grouped query order
        interp.strategy.path = path
        check_equal(interp.interp_array(qx, qy), ref, f"bilinear {nx}x{ny}x{C} path={path}")
    # extrapolation
    ex = pkg.Interp2DBuilder.new(g).x(x).y(y).strategy(pkg.Bilinear.new().extrapolate(True)).build()
    sx, sy = x[-1] - x[0], y[-1] - y[0]
    qx2 = rng.uniform(x[0] - sx, x[-1] + sx, Q).astype(dt); qy2 = rng.uniform(y[0] - sy, y[-1] + sy, Q).astype(dt)
    _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx2, qy2, True)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        ex.strategy.path = path
        check_equal(ex.interp_array(qx2, qy2), ref, f"bilinear extrapolate {nx}x{ny}x{C} path={path}")


def test_bilinear_errors(pkg):
    rng = np.random.default_rng(4)
    g = rng.uniform(0, 1, (6, 7, 8)); Q = 900
    interp = pkg.Interp2D.builder(g).build()
    qx = rng.uniform(0, 5, Q); qy = rng.uniform(0, 6, Q)
    qx[400] = 9.0; qy[400] = -1.0   # both out: x is reported (bilinear.rs:71-80)
    qy[650] = 11.0
    _, _, _, ref = oracle.interp2d_bilinear(np.arange(6.0), np.arange(7.0), g, qx[:400], qy[:400])
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        interp.strategy.path = path
        buf = np.full((Q, 8), -3.0)
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            interp.interp_array_into(qx, qy, buf)
        assert (ei.value.index, ei.value.axis, str(ei.value)) == (400, 0, "x = 9.0 is not in range")
        assert np.array_equal(buf[:400], ref) and np.all(buf[400:] == -3.0)
    qx[400] = 1.0
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        interp.interp_array(qx, qy)
    assert (ei.value.index, ei.value.axis) == (400, 1)
    with pytest.raises(pkg.Panic, match="do not match"):
        interp.interp_array(np.zeros(2), np.zeros(3))
    # data ranks 2..8 with a 2-D query (interp2d/mod.rs:521-589)
    for rank in range(2, 9):
        data = rng.uniform(0, 1, (4,) * rank)
        itr = pkg.Interp2D.builder(data).build()
        res = itr.interp(2.2, 1.1)
        assert res.ndim == rank - 2
        q2 = np.array([[0.5, 1.0], [1.5, 2.0]])
        res = itr.interp_array(q2, q2)
        assert res.shape == (2, 2) + data.shape[2:]
        _, _, _, ref = oracle.interp2d_bilinear(np.arange(4.0), np.arange(4.0), data, q2, q2)
        assert np.array_equal(res.reshape(4, -1), ref)
        buf = np.zeros(res.shape); itr.interp_array_into(q2, q2, buf)
        assert np.array_equal(buf, res)
    # N-d data and N-d queries (tests/interp2d.rs:241-265)
    data = rng.uniform(0, 1, (4, 4, 3, 2))
    it = pkg.Interp2D.builder(data).build()
    q = np.array([[0.5, 1.0], [1.5, 2.0]])
    res = it.interp_array(q, q)
    assert res.shape == (2, 2, 3, 2)
    _, _, _, ref = oracle.interp2d_bilinear(np.arange(4.0), np.arange(4.0), data, q, q)
    assert np.array_equal(res.reshape(4, 6), ref)


# ------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties + sampled oracle rows
# ------------------------------------------------------------------------------------------------
def test_full_size_c2_cubic(pkg):
    """configs[1]: 1D CubicSpline, 4096 knots x 4096 lanes f64, 1e6 queries, 1 GPU."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(42)
    n = L = 4096; Q = 1_000_000
    x = knots("rand", n, rng, np.float64)
    y = rng.uniform(0.0, 1.0, (n, L))
    interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
    q = rng.uniform(x[0], x[-1], Q)
    hit = rng.integers(0, n - 1, 2000)            # property 1: a query on a knot returns that data row exactly
    q[:2000] = x[hit]
    qd = torch.as_tensor(q, device=dev)
    out = torch.empty((Q, L), dtype=torch.float64, device=dev)
    interp.strategy.path = pkg.PATH_GATHER
    interp.interp_array_into(qd, out)
    yd = torch.as_tensor(y, device=dev)
    assert torch.equal(out[:2000], yd[torch.as_tensor(hit, device=dev)])
    # property 2: sampled rows against the oracle
    st, a, b = oracle.cubic_build(x, y)
    assert st == oracle.OK
    ca, cb = interp.strategy.coefficients()
    assert np.array_equal(ca, a) and np.array_equal(cb, b)
    pick = rng.integers(0, Q, 1500)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q[pick])
    assert np.array_equal(out[torch.as_tensor(pick, device=dev)].cpu().numpy(), ref)
    # property 3: the two formulations agree on every one of the 4.096e9 points
    s_gather = (out.sum(dtype=torch.float64).item(), out.view(torch.int64).sum().item())
    out.zero_()
    interp.strategy.path = pkg.PATH_BUCKETED
    interp.interp_array_into(qd, out)
    s_bucket = (out.sum(dtype=torch.float64).item(), out.view(torch.int64).sum().item())
    assert s_gather == s_bucket
    assert np.array_equal(out[torch.as_tensor(pick, device=dev)].cpu().numpy(), ref)


def test_full_size_c3_bilinear(pkg):
    """configs[2]: 2D Bilinear, 2048x2048 grid x 64 channels f32, 1e7 (x,y) queries, 1 GPU."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(96)
    nx = ny = 2048; C = 64; Q = 10_000_000
    x = np.arange(nx, dtype=np.float32); y = knots("jit", ny, rng, np.float32)
    g = rng.random((nx, ny, C), dtype=np.float32)
    interp = pkg.Interp2DBuilder.new(g).x(x).y(y).build()
    qx = rng.uniform(0, nx - 1, Q).astype(np.float32); qy = rng.uniform(y[0], y[-1], Q).astype(np.float32)
    hx = rng.integers(0, nx - 1, 1000); hy = rng.integers(0, ny - 1, 1000)
    qx[:1000] = x[hx]; qy[:1000] = y[hy]              # grid points reproduce the grid values exactly
    interp.strategy.path = pkg.PATH_GATHER
    out = interp.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    assert tuple(out.shape) == (Q, C)
    interp.strategy.path = pkg.PATH_BUCKETED      # tile-grouped order: identical results on all 6.4e8 points
    out_t = interp.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    assert torch.equal(out, out_t)
    del out_t
    assert np.array_equal(out[:1000].cpu().numpy(), g[hx, hy])
    pick = rng.integers(0, Q, 20000)
    _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx[pick], qy[pick])
    assert np.array_equal(out[torch.as_tensor(pick, device=dev)].cpu().numpy(), ref)
    # convexity: every value lies within the range of its four corners' channel values (here [0, 1))
    assert float(out.min()) >= -1e-6 and float(out.max()) <= 1.0 + 1e-6


# ------------------------------------------------------------------------------------------------
# host-buffer streaming mode and re-entrancy
# ------------------------------------------------------------------------------------------------
def test_host_output_is_streamed_in_chunks(pkg):
    """out_memspace = HOST: the batch goes through a 256 MiB device staging buffer in query chunks
    (8192 rows of 32 KiB here); first-error index and untouched rows must survive the chunking."""
    rng = np.random.default_rng(8)
    n, L, Q = 50, 4096, 20000           # 3 chunks
    x = knots("rand", n, rng, np.float64); y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
    interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    assert np.array_equal(interp.interp_array(q), ref)
    q[13000] = 7.0                       # inside the second chunk
    buf = np.full((Q, L), -2.0)
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        interp.interp_array_into(q, buf)
    assert ei.value.index == 13000 and ei.value.value == 7.0
    assert np.array_equal(buf[:13000], ref[:13000]) and np.all(buf[13000:] == -2.0)
    # 2-D
    g = rng.uniform(0, 1, (20, 30, 2048)); qx = rng.uniform(0, 19, 40000); qy = rng.uniform(0, 29, 40000)
    it = pkg.Interp2D.builder(g).build()
    _, _, _, ref2 = oracle.interp2d_bilinear(np.arange(20.0), np.arange(30.0), g, qx, qy)
    assert np.array_equal(it.interp_array(qx, qy), ref2)


def test_eval_is_reentrant_across_host_threads(pkg):
    """Query methods take &self and the reference's benches call one interpolator from many rayon workers
    (benches/bench_interp1d.rs:54-78): concurrent evaluations on one handle must not interfere."""
    import threading
    rng = np.random.default_rng(12)
    n, L = 200, 1024
    x = knots("rand", n, rng, np.float64); y = rng.uniform(0, 1, (n, L))
    interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(x, y)
    qs = [rng.uniform(x[0], x[-1], 3000 + 500 * i) for i in range(6)]
    refs = [oracle.interp1d_cubic(x, y, a, b, q)[2] for q in qs]
    outs = [None] * 6
    errs = []

    def work(i):
        try:
            for _ in range(5):
                outs[i] = interp.interp_array(qs[i])
                assert np.array_equal(outs[i], refs[i])
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_raw_c_abi_contract(pkg):
    """The C ABI used directly (as the Rust shim would): validate=1 create from host arrays, strided eval,
    BAD_ARG on a short row stride, coefficients copy-out, destroy."""
    import ctypes as C
    cap = pkg._capi
    lib = cap.lib()
    rng = np.random.default_rng(2)
    n, L, Q = 30, 8, 100
    x = np.sort(rng.uniform(0, 1, n)); y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
    d = cap.Interp1DDesc()
    d.dtype, d.strategy, d.extrapolate, d.device = cap.F64, cap.CUBIC_SPLINE, 0, 0
    d.n, d.lanes, d.x_len = n, L, n
    d.x, d.data, d.memspace, d.validate = x.ctypes.data, y.ctypes.data, cap.MEM_HOST, 1
    d.left = cap.Boundary(cap.BC_NATURAL, 0.0); d.right = cap.Boundary(cap.BC_FIRST_DERIV, 0.25)
    h = C.c_void_p()
    assert lib.ndi_interp1d_create(C.byref(d), C.byref(h)) == cap.OK
    a = np.empty((n - 1, L)); b = np.empty((n - 1, L))
    assert lib.ndi_interp1d_coefficients(h, a.ctypes.data, b.ctypes.data, cap.MEM_HOST) == cap.OK
    st, ra, rb = oracle.cubic_build(x, y, left=(1, 0.0), right=(3, 0.25))
    assert np.array_equal(a, ra) and np.array_equal(b, rb)
    out = np.full((Q, L + 3), -9.0)
    info = cap.OobInfo()
    opts = cap.EvalOpts()   # host queries, host output, default stream, AUTO
    assert lib.ndi_interp1d_eval(h, q.ctypes.data, Q, out.ctypes.data, L + 3, C.byref(opts), C.byref(info)) == cap.OK
    _, _, ref = oracle.interp1d_cubic(x, y, ra, rb, q)
    assert np.array_equal(out[:, :L], ref) and np.all(out[:, L:] == -9.0)
    assert lib.ndi_interp1d_eval(h, q.ctypes.data, Q, out.ctypes.data, L - 1, C.byref(opts), C.byref(info)) == cap.BAD_ARG
    assert lib.ndi_interp1d_eval(h, q.ctypes.data, 0, out.ctypes.data, L, C.byref(opts), C.byref(info)) == cap.OK
    q[5] = 3.0
    assert lib.ndi_interp1d_eval(h, q.ctypes.data, Q, out.ctypes.data, L + 3, C.byref(opts), C.byref(info)) == cap.OUT_OF_BOUNDS
    assert (info.index, info.value, info.axis, info.status) == (5, 3.0, 0, cap.OUT_OF_BOUNDS)
    assert "x = 3 is not in range" in cap.last_error()
    lib.ndi_interp1d_destroy(h)
    d.device = 99
    assert lib.ndi_interp1d_create(C.byref(d), C.byref(h)) == cap.BAD_ARG


def test_eval_can_be_captured_in_a_hip_graph(pkg):
    """Launch-bound repeated batches: after one warm-up call (scratch sized) the whole evaluation --
    status reset, locate, grouping, evaluation -- is stream-ordered work with no allocation or host sync, so it
    can be captured once in a HIP graph and replayed on new query values."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(31)
    n, L, Q = 64, 1024, 4096
    x = knots("rand", n, rng, np.float64); y = rng.uniform(0, 1, (n, L))
    interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(x, y)
    qd = torch.as_tensor(rng.uniform(x[0], x[-1], Q), device=dev)
    out = torch.zeros((Q, L), dtype=torch.float64, device=dev)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        interp.strategy.path = path
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            interp.strategy.interp_array_into(interp, qd, out, async_launch=True)   # warm-up on the capture stream
            interp.strategy.finish()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
        for rep in range(3):
            q_new = rng.uniform(x[0], x[-1], Q)
            qd.copy_(torch.as_tensor(q_new, device=dev))
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            _, _, ref = oracle.interp1d_cubic(x, y, a, b, q_new)
            assert np.array_equal(out.cpu().numpy(), ref)


# ------------------------------------------------------------------------------------------------
# seeded fuzz: random shapes / dtypes / strategies / modes / formulations against the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(8))
def test_fuzz_against_oracle(pkg, seed):
    rng = np.random.default_rng(9000 + seed)
    for _ in range(12):
        dt = rng.choice([np.float64, np.float32])
        kind = rng.choice(["linear", "cubic", "bilinear"])
        Q = int(rng.integers(1, 3000))
        path = int(rng.choice([pkg.PATH_AUTO, pkg.PATH_GATHER, pkg.PATH_BUCKETED]))
        if kind == "bilinear":
            nx, ny = int(rng.integers(2, 40)), int(rng.integers(2, 40))
            C = int(rng.choice([1, 2, 3, 4, 8, 12, 64, 100]))
            x = knots(rng.choice(["rand", "jit", "log"]), nx, rng, dt) if nx > 2 else np.array([0.0, 2.0], dtype=dt)
            y = knots(rng.choice(["rand", "jit", "log"]), ny, rng, dt) if ny > 2 else np.array([-1.0, 0.5], dtype=dt)
            g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
            ext = bool(rng.integers(0, 2))
            sx, sy = float(x[-1] - x[0]), float(y[-1] - y[0])
            m = 0.3 if ext else 0.0
            qx = rng.uniform(x[0] - m * sx, x[-1] + m * sx, Q).astype(dt)
            qy = rng.uniform(y[0] - m * sy, y[-1] + m * sy, Q).astype(dt)
            if not ext:
                qx = np.clip(qx, x[0], x[-1]); qy = np.clip(qy, y[0], y[-1])
            it = pkg.Interp2DBuilder.new(g).x(x).y(y).strategy(pkg.Bilinear.new().extrapolate(ext)).build()
            it.strategy.path = path
            _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx, qy, ext)
            check_equal(it.interp_array(qx, qy), ref, f"fuzz bilinear seed={seed} {nx}x{ny}x{C} ext={ext} path={path}")
            continue
        n = int(rng.integers(3 if kind == "cubic" else 2, 400))
        L = int(rng.choice([1, 2, 3, 5, 8, 64, 130, 512, 1024, 1030, 2048]))
        x = knots(rng.choice(["rand", "jit", "log", "lin"]), n, rng, dt) if n > 3 else np.arange(n).astype(dt)
        y = rng.uniform(-1, 1, (n, L)).astype(dt)
        span = float(x[-1] - x[0])
        if kind == "linear":
            ext = bool(rng.integers(0, 2))
            m = 0.5 if ext else 0.0
            q = rng.uniform(x[0] - m * span, x[-1] + m * span, Q).astype(dt)
            if not ext:
                q = np.clip(q, x[0], x[-1])
            it = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.Linear.new().extrapolate(ext)).build()
            it.strategy.path = path
            _, _, ref = oracle.interp1d_linear(x, y, q, ext)
            check_equal(it.interp_array(q), ref, f"fuzz linear seed={seed} n={n} L={L} ext={ext} path={path}")
            continue
        per = bool(rng.integers(0, 4) == 0) and n >= 3
        ext = bool(rng.integers(0, 2))
        lk, rk = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        lv, rv = float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))
        yy = y.copy()
        S, R, B = pkg.SingleBoundary, pkg.RowBoundary, pkg.BoundaryCondition
        def single(k, v):
            return {0: S.NotAKnot, 1: S.Natural, 2: S.Clamped}.get(k) or (S.FirstDeriv(v) if k == 3 else S.SecondDeriv(v))
        if per:
            yy[-1] = yy[0]
            bc = B.Periodic
        else:
            rows = np.empty((1, L), dtype=object)
            for i in range(L):
                rows[0, i] = R.Mixed(single(lk, lv), single(rk, rv))
            bc = B.Individual(rows)
        m = 0.5 if ext else 0.0
        q = rng.uniform(x[0] - m * span, x[-1] + m * span, Q).astype(dt)
        if not ext:
            q = np.clip(q, x[0], x[-1])
        it = pkg.Interp1DBuilder.new(yy).x(x).strategy(pkg.CubicSpline.new().extrapolate(ext).boundary(bc)).build()
        it.strategy.path = path
        st, a, b = oracle.cubic_build(x, yy, periodic=per, left=(lk, lv), right=(rk, rv))
        assert st == oracle.OK
        mode = oracle.EXTRAPOLATE_NO if not ext else (oracle.EXTRAPOLATE_PERIODIC if per else oracle.EXTRAPOLATE_YES)
        _, _, ref = oracle.interp1d_cubic(x, yy, a, b, q, mode)
        check_equal(it.interp_array(q), ref, f"fuzz cubic seed={seed} n={n} L={L} per={per} ext={ext} bc=({lk},{rk}) path={path}")


def test_infinite_queries_follow_ieee_like_the_cpu(pkg):
    """+-inf queries: out of range without extrapolation (OutOfBounds with the reference's text), and with
    extrapolation the end interval's polynomial evaluated at +-inf -- inf / NaN exactly where the CPU gets them."""
    rng = np.random.default_rng(3)
    n, L = 12, 512
    x = knots("jit", n, rng, np.float64); y = rng.uniform(-1, 1, (n, L))
    q = rng.uniform(x[0], x[-1], 64); q[7] = np.inf; q[20] = -np.inf
    lin = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.Linear.new().extrapolate(True)).build()
    _, _, ref = oracle.interp1d_linear(x, y, q, True)
    check_equal(lin.interp_array(q), ref, "linear +-inf")
    cub = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new().extrapolate(True)).build()
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q, oracle.EXTRAPOLATE_YES)
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        cub.strategy.path = path
        check_equal(cub.interp_array(q), ref, "cubic +-inf")
    strict = pkg.Interp1DBuilder.new(y).x(x).build()
    with pytest.raises(pkg.InterpolateError.OutOfBounds, match="x = inf is not in range") as ei:
        strict.interp_array(q)
    assert ei.value.index == 7


def test_empty_batches_and_size_limits(pkg):
    rng = np.random.default_rng(1)
    y = rng.uniform(0, 1, (5, 6))
    interp = pkg.Interp1D.builder(y).build()
    assert interp.interp_array(np.zeros((0,))).shape == (0, 6)
    assert interp.interp_array(np.zeros((3, 0))).shape == (3, 0, 6)
    cub = pkg.Interp1DBuilder.new(y).strategy(pkg.CubicSpline.new()).build()
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        cub.strategy.path = path
        assert cub.interp_array(np.zeros((0,))).shape == (0, 6)
    it2 = pkg.Interp2D.builder(rng.uniform(0, 1, (4, 5, 3))).build()
    assert it2.interp_array(np.zeros((0,)), np.zeros((0,))).shape == (0, 3)
    # one query, one lane
    s = pkg.Interp1D.builder(np.array([1.0, 3.0])).build()
    assert s.interp_array(np.array([0.25]))[0] == 1.5
    # a large axis: 64^3 + 1 knots (two-level pyramid read from global memory, blocks of 8192 knots)
    n = 64 ** 3 + 1
    big = pkg.Interp1DBuilder.new(np.arange(n, dtype=np.float32)).build()
    q = np.array([0.5, n - 1.5, 123456.25, 0.0, n - 1.0], dtype=np.float32)
    assert np.array_equal(big.interp_array(q), q)            # identity data on the index axis
    k = np.cumsum(np.random.default_rng(0).uniform(0.5, 1.5, n))
    qq = np.random.default_rng(1).uniform(k[0], k[-1], 20000)
    assert np.array_equal(pkg.get_lower_index(k, qq), np.clip(np.searchsorted(k, qq, side="right") - 1, 0, n - 2))


def test_small_lanes_device_buffers_keep_first_error_semantics(pkg):
    """<= 16 lanes with device-resident buffers run the fused kernel behind a range pre-check: rows at or after
    the first failing query must stay untouched in the caller's buffer, exactly as on the two-kernel path."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(17)
    for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        for L in (1, 2, 5):
            n, Q = 300, 50000
            x = knots("rand", n, rng, dt); y = rng.uniform(-1, 1, (n, L)).astype(dt)
            q = rng.uniform(x[0], x[-1], Q).astype(dt)
            st, a, b = oracle.cubic_build(x, y)
            _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
            it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
                .strategy(pkg.CubicSpline.new()).build()
            out = it.interp_array(torch.as_tensor(q, device=dev))
            check_equal(out.cpu().numpy().reshape(Q, L), ref, f"small lanes device L={L}")
            q[30000] = x[-1] + 1; q[41000] = x[0] - 1
            buf = torch.full((Q, L), -4.0, dtype=tdt, device=dev)
            with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
                it.interp_array_into(torch.as_tensor(q, device=dev), buf)
            assert ei.value.index == 30000
            h = buf.cpu().numpy()
            assert np.array_equal(h[:30000], ref[:30000]) and np.all(h[30000:] == -4.0)
    # periodic + extrapolate: an infinite query wraps to NaN -> panic-equivalent, as on the CPU
    x = np.arange(6.0); y = np.array([1.0, 2.0, 0.5, 3.0, 2.0, 1.0])
    per = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).strategy(
        pkg.CubicSpline.new().extrapolate(True).boundary(pkg.BoundaryCondition.Periodic)).build()
    with pytest.raises(pkg.Panic):
        per.interp_array(torch.as_tensor(np.array([0.5, np.inf, 2.0]), device=dev))
    # 2-D
    g = rng.uniform(0, 1, (20, 30, 2)); Q = 30000
    qx = rng.uniform(0, 19, Q); qy = rng.uniform(0, 29, Q)
    _, _, _, ref = oracle.interp2d_bilinear(np.arange(20.0), np.arange(30.0), g, qx, qy)
    it2 = pkg.Interp2D.builder(torch.as_tensor(g, device=dev)).build()
    out = it2.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    check_equal(out.cpu().numpy(), ref, "small lanes device 2-D")
    qy[12345] = 99.0; qx[20000] = -1.0
    buf = torch.full((Q, 2), -4.0, dtype=torch.float64, device=dev)
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        it2.interp_array_into(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), buf)
    assert (ei.value.index, ei.value.axis) == (12345, 1)
    h = buf.cpu().numpy()
    assert np.array_equal(h[:12345], ref[:12345]) and np.all(h[12345:] == -4.0)


def test_unaligned_device_buffers_and_two_streams(pkg):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(23)
    n, L, Q = 80, 1024, 3000
    x = knots("rand", n, rng, np.float64); y = rng.uniform(-1, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
        .strategy(pkg.CubicSpline.new()).build()
    qd = torch.as_tensor(q, device=dev)
    # output that starts 8 bytes into an allocation (not 16-byte aligned) with an odd row stride:
    # the library must fall back to scalar accesses, for both formulations
    big = torch.full((Q * (L + 1) + 8,), -1.0, dtype=torch.float64, device=dev)
    view = big[1:1 + Q * (L + 1)].view(Q, L + 1)[:, :L]
    assert view.data_ptr() % 16 == 8 and view.stride(0) == L + 1
    for path in (pkg.PATH_GATHER, pkg.PATH_BUCKETED):
        it.strategy.path = path
        big.fill_(-1.0)
        it.strategy.interp_array_into(it, qd, view)
        h = big.cpu().numpy()
        got = h[1:1 + Q * (L + 1)].reshape(Q, L + 1)
        assert np.array_equal(got[:, :L], ref) and np.all(got[:, L] == -1.0) and h[0] == -1.0
    # two streams, one handle, one host thread: each stream has its own scratch and status
    it.strategy.path = pkg.PATH_AUTO
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    q2 = rng.uniform(x[0], x[-1], Q); q2[77] = x[-1] + 1.0
    _, _, ref2 = oracle.interp1d_cubic(x, y, a, b, q2[:77])
    o1 = torch.zeros((Q, L), dtype=torch.float64, device=dev); o2 = torch.full((Q, L), -9.0, dtype=torch.float64, device=dev)
    q2d = torch.as_tensor(q2, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        it.strategy.interp_array_into(it, qd, o1, async_launch=True)
    with torch.cuda.stream(s2):
        it.strategy.interp_array_into(it, q2d, o2, async_launch=True)
    with torch.cuda.stream(s1):
        it.strategy.finish()                       # stream 1: clean batch
    with torch.cuda.stream(s2):
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.strategy.finish()                   # stream 2: its own first error
    assert ei.value.index == 77
    assert np.array_equal(o1.cpu().numpy(), ref)
    h2 = o2.cpu().numpy()
    assert np.array_equal(h2[:77], ref2) and np.all(h2[77:] == -9.0)


def test_full_size_c5_share_bilinear(pkg):
    """configs[4], one GPU's share: 2D Bilinear, 8192x8192 grid x 16 channels f32 (4 GiB, replicated per device),
    1.25e7 scattered queries (bilinear.rs:64-99).  Random knots on x, the default index axis on y, the pair-packed
    grid layout; grid-point hits exact, 20 000 sampled rows bit-exact against the oracle, convexity."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    nx = ny = 8192; C = 16; Q = 12_500_000
    gd = torch.rand((nx, ny, C), dtype=torch.float32, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    x = knots("jit", nx, rng, np.float32); y = np.arange(ny, dtype=np.float32)
    interp = pkg.Interp2DBuilder.new(gd).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    qx = rng.uniform(x[0], x[-1], Q).astype(np.float32); qy = rng.uniform(0, ny - 1, Q).astype(np.float32)
    hx = rng.integers(0, nx - 1, 1000); hy = rng.integers(0, ny - 1, 1000)
    qx[:1000] = x[hx]; qy[:1000] = y[hy]
    out = interp.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    assert tuple(out.shape) == (Q, C)
    hxd, hyd = torch.as_tensor(hx, device=dev), torch.as_tensor(hy, device=dev)
    assert torch.equal(out[:1000], gd[hxd, hyd])                       # grid points reproduce grid values
    pick = rng.integers(0, Q, 20000)
    g = gd.cpu().numpy()                                               # 4 GiB on the host for the oracle
    _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx[pick], qy[pick])
    assert np.array_equal(out[torch.as_tensor(pick, device=dev)].cpu().numpy(), ref)
    del g
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    # tile-grouped order on the same batch: identical on all 2e8 outputs
    interp.strategy.path = pkg.PATH_BUCKETED
    out_t = interp.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    assert torch.equal(out, out_t)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_bilinear_large_batch_knots_in_lds(pkg, dt):
    """Batches of >= 2^20 queries run the evaluation with both knot vectors staged in LDS (1024-thread workgroups,
    two items per thread): same bits as the oracle, incl. the first-error cut and the tile-grouped order."""
    import torch
    rng = np.random.default_rng(77)
    nx, ny, C, Q = 301, 200, 8, (1 << 20) + 12_345
    x = knots("rand", nx, rng, dt); y = knots("log", ny, rng, dt)
    g = rng.random((nx, ny, C)).astype(dt)
    interp = pkg.Interp2DBuilder.new(g).x(x).y(y).build()
    qx = rng.uniform(x[0], x[-1], Q).astype(dt); qy = rng.uniform(y[0], y[-1], Q).astype(dt)
    _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx, qy)
    out = interp.interp_array(torch.as_tensor(qx, device="cuda:0"), torch.as_tensor(qy, device="cuda:0"))
    assert np.array_equal(out.cpu().numpy(), ref)
    interp.strategy.path = pkg.PATH_BUCKETED
    out_t = interp.interp_array(torch.as_tensor(qx, device="cuda:0"), torch.as_tensor(qy, device="cuda:0"))
    assert torch.equal(out, out_t)
    interp.strategy.path = pkg.PATH_AUTO
    qy[1_000_000] = np.nan
    buf = torch.full((Q, C), -2.0, dtype=out.dtype, device="cuda:0")
    with pytest.raises(pkg.InterpolateError.OutOfBounds, match="^y = NaN") as ei:
        interp.interp_array_into(torch.as_tensor(qx, device="cuda:0"), torch.as_tensor(qy, device="cuda:0"), buf)
    assert ei.value.index == 1_000_000
    assert np.array_equal(buf[:1_000_000].cpu().numpy(), ref[:1_000_000]) and bool((buf[1_000_000:] == -2.0).all())


@pytest.mark.parametrize("dt,nx,ny,C", [
    (np.float32, 70, 50, 32),       # LV = 8
    (np.float32, 33, 17, 32),       # a single row / column of tiles
    (np.float32, 130, 257, 64),     # C3's row length, ragged last tiles in both directions
    (np.float32, 21, 2000, 20),     # LV = 5 does not divide the workgroup -> the gather order
    (np.float64, 100, 90, 32),      # LV = 16, f64 records ({qx, qy} beside the 16-byte record)
    (np.float64, 65, 66, 34),       # LV = 17 -> the gather order
    (np.float64, 40, 30, 3),        # odd trailing axis: no vector rows -> the gather order
    (np.float32, 300, 300, 16),     # <= 64 bytes per grid point: pair-packed layout -> the gather order
    (np.float32, 40, 40, 512),      # long rows: a smaller tile (2^ts + 1)^2 x 2 KiB must fit LDS
])
def test_bilinear_tile_grouped_lds(pkg, dt, nx, ny, C):
    """ndi_path BUCKETED for 2-D: queries grouped by tile, every tile staged once in LDS
    (eval_bilinear_tiles_kernel).  Bit-identical to the oracle and to the gather order, incl. grid-point hits, the
    last cells of both axes, extrapolation, and the first-error cut; shapes the tiled kernel does not cover take the
    gather order."""
    import torch
    rng = np.random.default_rng(nx * 7 + ny * 3 + C)
    vn = {np.float32: 4, np.float64: 2}[dt]
    tiled_ok = C * np.dtype(dt).itemsize > 64 and C % vn == 0 and 1024 % (C // vn) == 0
    expect = "bucketed" if tiled_ok else "gather"
    x = knots("rand", nx, rng, dt); y = knots("jit", ny, rng, dt)
    g = rng.uniform(0, 1, (nx, ny, C)).astype(dt)
    Q = 40_000
    qx = rng.uniform(x[0], x[-1], Q).astype(dt); qy = rng.uniform(y[0], y[-1], Q).astype(dt)
    gi = rng.integers(0, nx - 1, 500); gj = rng.integers(0, ny - 1, 500)
    qx[:500] = x[gi]; qy[:500] = y[gj]                                   # grid-point hits (left ends: t = 0 is exact)
    qx[500:503] = [x[-1], x[0], x[-1]]; qy[500:503] = [y[-1], y[-1], y[0]]  # corners: the last cells of both axes
    interp = pkg.Interp2DBuilder.new(g).x(x).y(y).build()
    _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx, qy)
    interp.strategy.path = pkg.PATH_BUCKETED
    got = interp.interp_array(qx, qy)
    assert pkg.profile_read(reset=False)["last_path"] == expect
    check_equal(got, ref, f"tiled {nx}x{ny}x{C}")
    assert np.array_equal(got[:500], g[gi, gj])
    # device-resident queries and output, strided output rows
    outd = torch.full((Q, 2 * C), -5.0, dtype=torch.float32 if dt == np.float32 else torch.float64, device="cuda:0")
    interp.strategy.interp_array_into(interp, torch.as_tensor(qx, device="cuda:0"), torch.as_tensor(qy, device="cuda:0"),
                                      outd[:, :C])
    assert np.array_equal(outd[:, :C].cpu().numpy(), ref) and bool((outd[:, C:] == -5.0).all())
    # extrapolation: queries outside the axes use the end cells
    ex = pkg.Interp2DBuilder.new(g).x(x).y(y).strategy(pkg.Bilinear.new().extrapolate(True)).build()
    ex.strategy.path = pkg.PATH_BUCKETED
    sx, sy = x[-1] - x[0], y[-1] - y[0]
    qx2 = rng.uniform(x[0] - sx, x[-1] + sx, Q).astype(dt); qy2 = rng.uniform(y[0] - sy, y[-1] + sy, Q).astype(dt)
    _, _, _, ref2 = oracle.interp2d_bilinear(x, y, g, qx2, qy2, True)
    check_equal(ex.interp_array(qx2, qy2), ref2, f"tiled extrapolate {nx}x{ny}x{C}")
    # first error: rows before it written, later rows untouched (interp2d/mod.rs:297-306)
    qy[31_000] = y[-1] + 1; qx[35_000] = x[0] - 1
    buf = np.full((Q, C), -3.0, dtype=dt)
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        interp.interp_array_into(qx, qy, buf)
    assert (ei.value.index, ei.value.axis) == (31_000, 1)
    assert np.array_equal(buf[:31_000], ref[:31_000]) and np.all(buf[31_000:] == -3.0)
    # skewed batch: every query in one cell (one tile holds everything; many workgroups stage the same tile)
    qx3 = rng.uniform(x[3], x[4], Q).astype(dt); qy3 = rng.uniform(y[5], y[6], Q).astype(dt)
    _, _, _, ref3 = oracle.interp2d_bilinear(x, y, g, qx3, qy3)
    check_equal(interp.interp_array(qx3, qy3), ref3, f"tiled skewed {nx}x{ny}x{C}")


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_bilinear_tiles_shared_divisor_window(pkg, dt):
    """The tile kernel divides by the per-query knot spacing with a shared reciprocal + exact-residual FMA
    corrections inside an exponent window and with the IEEE division outside it (kernels.hpp div_shared).  Grid
    values and knot spacings chosen to sit inside, at the edges of and far outside the window -- exact zeros (flat
    regions), denormal-sized and huge differences, mixed signs, a tiny and a huge knot spacing -- must give the
    oracle's bits, as the gather kernel (IEEE divisions throughout) does."""
    rng = np.random.default_rng(91)
    nx, ny, C = 70, 60, 32
    fin = np.finfo(dt)
    x = np.cumsum(rng.uniform(0.5, 1.5, nx)).astype(dt)
    y = np.cumsum(rng.uniform(0.5, 1.5, ny)).astype(dt)
    # knot spacings outside the divisor window: one tiny, one huge interval per axis
    x[20:] += dt(1e-30) - (x[20] - x[19]); x[40:] += dt(1e30 if dt == np.float32 else 1e250)
    y[10:] = y[10:] - (y[10] - y[9]) + dt(3e-25 if dt == np.float32 else 1e-200)
    x = np.unique(x); y = np.unique(y)
    nx, ny = x.size, y.size
    g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
    scales = np.array([1.0, fin.tiny * 4, fin.tiny / 8, 1e-22, 1e22, fin.max / 64, 0.0, 1.0], dtype=dt)
    g *= scales[rng.integers(0, scales.size, (nx, ny, 1))]          # per grid point: normal, tiny, denormal, huge, zero
    g[5:12, 5:12, :] = dt(0.75)                                     # a flat region: every difference is an exact zero
    g[30:33, 20:23, ::2] = -g[30:33, 20:23, ::2]
    Q = 60_000
    ix = rng.integers(0, nx - 1, Q); iy = rng.integers(0, ny - 1, Q)
    tx = rng.uniform(0, 1, Q).astype(dt); ty = rng.uniform(0, 1, Q).astype(dt)
    qx = (x[ix] + (x[ix + 1] - x[ix]) * tx).astype(dt); qy = (y[iy] + (y[iy + 1] - y[iy]) * ty).astype(dt)
    qx = np.clip(qx, x[0], x[-1]); qy = np.clip(qy, y[0], y[-1])
    qx[:2000] = x[ix[:2000]]; qy[2000:4000] = y[iy[2000:4000]]      # on knots: t = 0 exactly
    with np.errstate(all="ignore"):
        _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx, qy)
    interp = pkg.Interp2DBuilder.new(g).x(x).y(y).build()
    got = {}
    for name, path in (("gather", pkg.PATH_GATHER), ("tiles", pkg.PATH_BUCKETED)):
        interp.strategy.path = path
        got[name] = interp.interp_array(qx, qy)
    assert pkg.profile_read(reset=False)["last_path"] == "bucketed"
    for name, arr in got.items():
        assert np.array_equal(arr.view(np.uint32 if dt == np.float32 else np.uint64),
                              ref.view(np.uint32 if dt == np.float32 else np.uint64)) or \
            np.array_equal(arr, ref, equal_nan=True), name
    # bit for bit between the two device formulations, NaN payloads and zero signs included
    bits = np.uint32 if dt == np.float32 else np.uint64
    same = got["gather"].view(bits) == got["tiles"].view(bits)
    both_nan = np.isnan(got["gather"]) & np.isnan(got["tiles"])
    assert np.all(same | both_nan)
    assert np.isfinite(ref).mean() > 0.5                            # the case is not degenerate


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("C", [1, 2, 3, 5, 6, 7, 8])
def test_bilinear_small_channels_device_buffers(pkg, dt, C, capfd):
    """Scalar grids and rows of a few values that are not 16-byte vectors (the reference's 100 x 100 and 100 x 100 x 5,
    benches/bench_interp2d.rs) on device-resident buffers: one query per thread, both searches (bucket index from 4096
    queries, pyramid below) and the evaluation in one launch, the three divisions of a value through the query's two
    correctly rounded reciprocals -- the oracle's bits (bilinear.rs:64-99), incl. values far outside the divisor window,
    extrapolation, and the first-error cut."""
    import torch
    tdt = torch.float64 if dt == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(500 + C)
    for nx, ny, Q in ((100, 100, 100_000), (37, 300, 1000), (2, 2, 5000), (700, 45, 70_001), (90, 110, 600_000)):
        x = knots("rand", nx, rng, dt) if nx > 2 else np.array([0.0, 2.0], dtype=dt)
        y = knots("jit", ny, rng, dt) if ny > 2 else np.array([-1.0, 0.5], dtype=dt)
        g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
        g[0, 0] = 0.0; g[-1, -1] = np.finfo(dt).max / 4; g[nx // 2, ny // 2] = np.finfo(dt).tiny    # outside the window
        if nx > 4:
            g[3, :, 0] = g[4, :, 0]                                                                  # flat: zero numerators
        for ext in (False, True):
            m = 0.3 if ext else 0.0
            sx, sy = float(x[-1] - x[0]), float(y[-1] - y[0])
            qx = rng.uniform(x[0] - m * sx, x[-1] + m * sx, Q).astype(dt)
            qy = rng.uniform(y[0] - m * sy, y[-1] + m * sy, Q).astype(dt)
            if not ext:
                qx = np.clip(qx, x[0], x[-1]); qy = np.clip(qy, y[0], y[-1])
            qx[:3] = [x[0], x[-1], x[0]]; qy[:3] = [y[0], y[-1], y[-1]]
            it = pkg.Interp2DBuilder.new(torch.as_tensor(g if C > 1 else g[..., 0], device=dev)).x(torch.as_tensor(x, device=dev)) \
                .y(torch.as_tensor(y, device=dev)).strategy(pkg.Bilinear.new().extrapolate(ext)).build()
            _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx, qy, ext)
            out = torch.full((Q, C), -4.0, dtype=tdt, device=dev)
            capfd.readouterr()
            os.environ["NDI_TRACE_PLAN"] = "1"
            try:
                it.strategy.interp_array_into(it, torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), out)
            finally:
                del os.environ["NDI_TRACE_PLAN"]
            err = capfd.readouterr().err
            # (rows of up to 64 bytes on grids beyond LDS: the slope-record kernel, eval_slopes2d_kernel)
            took_query_order = "[ndi plan] fused2d" in err or "[ndi plan] slopes2d" in err
            # the query-order kernel from 65 536 queries (1-2 values per point: from 524 288), else the two-kernel / one-thread forms
            assert took_query_order == (Q >= (524_288 if C <= 2 else 65_536)), (nx, ny, C, Q)
            check_equal(out.cpu().numpy(), ref.reshape(Q, C), f"2-D small rows {nx}x{ny}x{C} Q={Q} ext={ext}")
            if not ext:   # first error: rows before it written, later rows untouched
                qx2 = qx.copy(); qx2[Q // 2] = x[-1] + 1
                out2 = torch.full((Q, C), -4.0, dtype=tdt, device=dev)
                with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
                    it.strategy.interp_array_into(it, torch.as_tensor(qx2, device=dev), torch.as_tensor(qy, device=dev), out2)
                assert (ei.value.index, ei.value.axis) == (Q // 2, 0)
                o2 = out2.cpu().numpy()
                assert np.array_equal(o2[: Q // 2], ref.reshape(Q, C)[: Q // 2]) and np.all(o2[Q // 2:] == -4.0)


def test_baseline_c1_exact_workload(pkg):
    """BASELINE configs[0] exactly as bench.py times it: 1-D Linear on the DEFAULT index axis (Interp1DBuilder::new's
    x = 0..n, interp1d/mod.rs:399-410), 1024 f64 knots, scalar data, 1e4 uniform queries in [0, n - 1] (the shape of
    benches/bench_interp1d.rs:33-37 with 1024 knots) -- host arrays in and out (the zero-copy small-batch path with the
    O(1) index guess inside the fused kernel), then the same batch with device buffers, both bit-equal to the oracle."""
    import torch
    n, nq = 1024, 10_000
    yv = np.random.default_rng(42).uniform(0, 1, n)
    q = np.random.default_rng(123).uniform(0, n - 1, nq)
    q[:3] = [0.0, n - 1.0, 511.0]
    x = np.arange(n, dtype=np.float64)
    ref = oracle.interp1d_linear(x, yv, q)[2][:, 0]
    interp = pkg.Interp1DBuilder.new(yv).build()                 # no .x(): the default axis; no .strategy(): Linear
    got = interp.interp_array(q)
    assert got.shape == (nq,) and got.dtype == np.float64
    check_equal(got, ref, "C1 host arrays")
    assert got[0] == yv[0] and got[1] == yv[-1] and got[2] == yv[511]
    # the bare C ABI call bench.py's secondary.c1 leg makes
    import ctypes
    cap = pkg._capi
    out = np.full(nq, -1.0); opts = cap.EvalOpts(); info = cap.OobInfo()
    assert cap.lib().ndi_interp1d_eval(interp.strategy._h, q.ctypes.data, nq, out.ctypes.data, 1, ctypes.byref(opts),
                                       ctypes.byref(info)) == 0
    check_equal(out, ref, "C1 C ABI host to host")
    # device buffers (tensor in, tensor out; and into a caller-owned device buffer)
    dev = torch.device("cuda:0")
    di = pkg.Interp1DBuilder.new(torch.as_tensor(yv, device=dev)).build()
    gd = di.interp_array(torch.as_tensor(q, device=dev))
    assert tuple(gd.shape) == (nq,)
    check_equal(gd.cpu().numpy(), ref, "C1 device buffers")
    buf = torch.full((nq,), -5.0, dtype=torch.float64, device=dev)
    di.interp_array_into(torch.as_tensor(q, device=dev), buf)
    check_equal(buf.cpu().numpy(), ref, "C1 device buffers, interp_array_into")
    # out of range on the index axis: the reference's message and the first-error cut
    qb = q.copy(); qb[7000] = n - 1 + 1e-9
    with pytest.raises(pkg.InterpolateError.OutOfBounds, match=r"^x = 1023\.0+1 is not in range") as ei:
        interp.interp_array(qb)
    assert ei.value.index == 7000
