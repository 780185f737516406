"""GPU parity of the BLOCKED Thomas sweeps (CubicSpline::build for narrow trailing axes on many knots,
cubic_spline.rs:409-721): the one path whose results are not bit-identical to the reference order -- the two sweeps
are evaluated block-wise with precomputed coefficient products and the back substitution as r'/mid' + (-up/mid') k
instead of (r' - up k)/mid'.  Bar = the north star's: 1e-10 (f64) / 1e-5 (f32) relative to the largest magnitude,
on the coefficient tables and on evaluated rows; with NDI_SPLINE_BLOCKED=0 -- or per handle with
CubicSpline.reference_order(True) = NDI_BUILD_REFERENCE_ORDER -- the same shapes are bit-exact again.
The contract include/ndinterp.h states is tighter than the global bar and LOCAL: every coefficient within
1e-12 (f64) / 1e-5 (f32) of the larger of the table magnitudes within 32 rows of it and the interval's |dy| -- asserted on
strongly non-uniform axes too (geometric over six decades, clustered gaps over six decades, data of varying scale)."""
import os

import numpy as np
import pytest

import oracle
from conftest import assert_rel
from test_gpu_parity import BCS, TOL, check_equal, knots, table_close

pytestmark = pytest.mark.gpu


class blocked:
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = os.environ.get("NDI_SPLINE_BLOCKED")
        if self.v is None:
            os.environ.pop("NDI_SPLINE_BLOCKED", None)
        else:
            os.environ["NDI_SPLINE_BLOCKED"] = str(self.v)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("NDI_SPLINE_BLOCKED", None)
        else:
            os.environ["NDI_SPLINE_BLOCKED"] = self.old


def _bc(pkg, L, left, right):
    S, R, B = pkg.SingleBoundary, pkg.RowBoundary, pkg.BoundaryCondition

    def single(kind, val):
        return {0: S.NotAKnot, 1: S.Natural, 2: S.Clamped}.get(kind) or (S.FirstDeriv(val) if kind == 3 else S.SecondDeriv(val))
    rows = np.empty((1, L), dtype=object)
    for i in range(L):
        rows[0, i] = R.Mixed(single(*left), single(*right))
    return B.Individual(rows)       # identical rows are one global boundary (interp1d.py)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("kind,n,L", [("rand", 2048, 1), ("jit", 4096, 8), ("rand", 5000, 3), ("log", 20_000, 5),
                                       ("jit", 100_000, 1), ("lin", 65_537, 2), ("jit", 3000, 256)])
def test_blocked_build_tables_and_rows_within_the_bar(pkg, dt, kind, n, L):
    rng = np.random.default_rng(n + L)
    x = knots(kind, n, rng, dt) if kind != "log" else np.unique(np.logspace(-2, 0, n).astype(dt))
    n = x.size
    y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], 20_000).astype(dt)
    q[:2] = [x[0], x[-1]]
    y0 = y
    for name, (per, left, right) in BCS.items():
        y = y0
        if per:                           # the condensed (n - 2) system through the same blocked sweeps
            y = y0.copy(); y[-1] = y[0]
        bc = pkg.BoundaryCondition.Periodic if per else _bc(pkg, L, left, right)
        st, ra, rb = oracle.cubic_build(x, y, periodic=per, left=left, right=right)
        assert st == oracle.OK
        _, _, ref = oracle.interp1d_cubic(x, y, ra, rb, q)
        with blocked(None):               # default: this shape takes the blocked sweeps
            it = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new().boundary(bc)).build()
        a, b = it.strategy.coefficients()
        table_close(a, ra, dt, f"a[{name}] {kind} n={n} L={L}")
        table_close(b, rb, dt, f"b[{name}] {kind} n={n} L={L}")
        got = np.asarray(it.interp_array(q), dtype=np.float64).reshape(ref.shape)
        # the reference's own assertion form (approx's assert_relative_eq!): absolute OR relative -- splines
        # on sorted-random knots overshoot by orders of magnitude between near-coincident knots
        tol = TOL[np.dtype(dt)]
        assert_rel(got, ref.astype(np.float64), tol * float(np.max(np.abs(y))), tol, f"rows [{name}] {kind} n={n} L={L}")
        assert np.array_equal(got[:2], y[[0, -1]].astype(np.float64).reshape(2, -1)), "knot hits return the data rows"
        with blocked(0):                  # the serial kernels on the same shape: bit-identical again
            it0 = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new().boundary(bc)).build()
        a0, b0 = it0.strategy.coefficients()
        check_equal(a0, ra, f"serial a[{name}] n={n} L={L}")
        check_equal(b0, rb, f"serial b[{name}] n={n} L={L}")


LOCAL_TOL = {np.dtype(np.float64): 1e-12, np.dtype(np.float32): 1e-5}


def table_close_local(got, ref, other_ref, y, dt, what):
    """|got - ref| <= tol * scale_i per entry, scale_i = max(|a|, |b| over rows i-32 .. i+32 of the same lane, |dy_i|)."""
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    mag = np.maximum(np.abs(ref), np.abs(np.asarray(other_ref, dtype=np.float64)))
    m = mag.shape[0]
    local = mag.copy()
    for sh in range(1, 33):                       # running maximum over +-32 rows
        local[sh:] = np.maximum(local[sh:], mag[:m - sh])
        local[:m - sh] = np.maximum(local[:m - sh], mag[sh:])
    dy = np.abs(np.diff(np.asarray(y, dtype=np.float64).reshape(m + 1, -1), axis=0))
    scale = np.maximum(local.reshape(m, -1), dy)
    err = np.abs(got.reshape(m, -1) - ref.reshape(m, -1))
    bad = err > LOCAL_TOL[np.dtype(dt)] * scale
    if np.any(bad):
        i = tuple(np.argwhere(bad)[0])
        worst = np.max(err / np.maximum(scale, 1e-300))
        raise AssertionError(f"{what}: {int(bad.sum())} entries beyond the local bound; first at {i}: err {err[i]:.3e} "
                             f"scale {scale[i]:.3e}; worst err / scale {worst:.3e}")


def _axis(kind, n, rng, dt):
    if kind == "geom":                            # six decades, every gap 1 + 14 / n times the previous one
        return np.unique(np.geomspace(1e-3, 1e3, n).astype(dt))
    if kind == "clustered":                       # gaps log-uniform over six decades: neighbours differ by up to 1e6
        return np.unique(np.cumsum(10.0 ** rng.uniform(-6, 0, n)).astype(dt))
    if kind == "alternating":                     # gaps alternating between decades -- 1, 1e-4, 1, 1e-2, ... -- on an
        g = np.ones(n)                            # axis long enough for the device-side elimination (n >= 32 768): the warm
        g[1::2] = 10.0 ** (-(np.arange(g[1::2].size) % 3 + 2.0))     # start of spline_eliminate_kernel against the host's chain
        g[::7] *= 50.0
        return np.unique(np.cumsum(g).astype(dt))
    return knots(kind, n, rng, dt)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("kind,n,L", [("geom", 4096, 4), ("clustered", 8192, 1), ("clustered", 3000, 64), ("geom", 50_000, 1),
                                       ("rand", 20_000, 2), ("jit", 4096, 8), ("alternating", 40_000, 2), ("clustered", 70_000, 1)])
def test_blocked_build_local_bound_on_non_uniform_axes(pkg, dt, kind, n, L):
    """a per-entry bound relative to the NEIGHBOURING magnitudes (not to the largest entry of the table), on
    axes whose gaps -- and therefore the table's magnitudes -- vary over many decades, with data whose scale varies
    along the axis as well; and the per-handle opt-out gives the reference's bits on the same shapes."""
    rng = np.random.default_rng(n * 3 + L)
    x = _axis(kind, n, rng, dt)
    n = x.size
    y = (rng.uniform(-1.0, 1.0, (n, L)) * 10.0 ** rng.uniform(-3, 3, (n, 1))).astype(dt)   # row scale varies over 6 decades
    if dt == np.float32 and kind == "clustered":
        y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)      # (f32: 1e6 x 1e6 dynamic range leaves no digits to compare)
    for name, (per, left, right) in BCS.items():
        yy = y
        if per:
            yy = y.copy(); yy[-1] = yy[0]
        bc = pkg.BoundaryCondition.Periodic if per else _bc(pkg, L, left, right)
        st, ra, rb = oracle.cubic_build(x, yy, periodic=per, left=left, right=right)
        assert st == oracle.OK
        with blocked(None):
            it = pkg.Interp1DBuilder.new(yy).x(x).strategy(pkg.CubicSpline.new().boundary(bc)).build()
            exact = pkg.Interp1DBuilder.new(yy).x(x).strategy(pkg.CubicSpline.new().boundary(bc).reference_order(True)).build()
        a, b = it.strategy.coefficients()
        table_close_local(a, ra, rb, yy, dt, f"a[{name}] {kind} n={n} L={L}")
        table_close_local(b, rb, ra, yy, dt, f"b[{name}] {kind} n={n} L={L}")
        a0, b0 = exact.strategy.coefficients()               # NDI_BUILD_REFERENCE_ORDER: the serial kernels, per handle
        check_equal(a0, ra, f"reference_order a[{name}] {kind} n={n} L={L}")
        check_equal(b0, rb, f"reference_order b[{name}] {kind} n={n} L={L}")
        q = rng.uniform(x[0], x[-1], 5000).astype(dt)
        _, _, ref = oracle.interp1d_cubic(x, yy, ra, rb, q)
        got = np.asarray(it.interp_array(q), dtype=np.float64).reshape(ref.shape)
        tol = TOL[np.dtype(dt)]
        assert_rel(got, ref.astype(np.float64), tol * float(np.max(np.abs(yy))), tol, f"rows [{name}] {kind} n={n} L={L}")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_blocked_build_forced_on_small_and_wide_shapes(pkg, dt):
    """NDI_SPLINE_BLOCKED=1: every block size / remainder combination (n just above a block multiple, one block,
    many lanes) against the oracle."""
    rng = np.random.default_rng(9)
    with blocked(1):
        for n, L in ((16, 1), (17, 4), (63, 2), (64, 3), (65, 1), (129, 70), (1000, 300), (4097, 2), (70_000, 1)):
            x = knots("jit", n, rng, dt)
            y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
            for left, right in (((0, 0.0), (0, 0.0)), ((3, 0.25), (4, -0.5)), ((1, 0.0), (2, 0.0))):
                st, ra, rb = oracle.cubic_build(x, y, left=left, right=right)
                it = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new().boundary(_bc(pkg, L, left, right))).build()
                a, b = it.strategy.coefficients()
                table_close(a, ra, dt, f"a n={n} L={L} {left} {right}")
                table_close(b, rb, dt, f"b n={n} L={L} {left} {right}")
            yp = y.copy(); yp[-1] = yp[0]
            st, ra, rb = oracle.cubic_build(x, yp, periodic=True)
            it = pkg.Interp1DBuilder.new(yp).x(x).strategy(pkg.CubicSpline.new().boundary(pkg.BoundaryCondition.Periodic)).build()
            a, b = it.strategy.coefficients()
            table_close(a, ra, dt, f"periodic a n={n} L={L}")
            table_close(b, rb, dt, f"periodic b n={n} L={L}")
            yp[-1] += 1.0                 # y[0] != y[n-1]: BuilderError::ValueError (cubic_spline.rs:501-507)
            with pytest.raises(pkg.BuilderError.ValueError):
                pkg.Interp1DBuilder.new(yp).x(x).strategy(pkg.CubicSpline.new().boundary(pkg.BoundaryCondition.Periodic)).build()
