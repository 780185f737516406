"""GPU parity of spline_build_wide_kernel (round 6): CubicSpline::build (cubic_spline.rs:310-368, 409-471, 597-721) for wide
trailing axes -- four waves per 64 lanes, the right-hand sides produced into LDS by three of them while the fourth runs
thomas (:678-721) and the a / b epilogue (:354-365) -- against the CPU oracle, BIT FOR BIT, for the three global boundary
conditions, both element types, row counts around the producer block size (32) and lane counts that are / are not
multiples of 64; AUTO (>= 1024 lanes) and forced (NDI_SPLINE_WIDE=1: narrower axes whose tables outgrow LDS); and the
serial kernels (NDI_SPLINE_WIDE=0) give the same bits.  BASELINE configs[1]'s own shape (4096 x 4096) is covered by
test_full_size_c2 (tests/test_gpu_parity.py), which builds through AUTO."""
import os

import numpy as np
import pytest

import oracle
from test_gpu_parity import check_equal, knots

pytestmark = pytest.mark.gpu

BCS = {"NotAKnot": ((0, 0.0), (0, 0.0)), "Natural": ((1, 0.0), (1, 0.0)), "Clamped": ((2, 0.0), (2, 0.0))}


class wide:
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        if self.value is not None:
            os.environ["NDI_SPLINE_WIDE"] = str(self.value)
        os.environ["NDI_SPLINE_BLOCKED"] = "0"          # (the blocked sweeps are another kernel family with its own tolerance)

    def __exit__(self, *a):
        os.environ.pop("NDI_SPLINE_WIDE", None)
        os.environ.pop("NDI_SPLINE_BLOCKED", None)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n,L,force", [(4, 40_000, None), (5, 30_000, None), (33, 5000, None), (34, 4099, None), (35, 4096, None),
                                        (66, 2500, None), (97, 1400, None), (1000, 1024, None), (4096, 1100, None),
                                        (2000, 70, 1), (700, 200, 1), (5000, 64, 1), (3000, 129, 1)])
def test_wide_build_bit_exact(pkg, dt, n, L, force):
    rng = np.random.default_rng(n * 7 + L)
    x = knots("jit" if n > 4 else "lin", n, rng, dt)
    y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
    for name, (left, right) in BCS.items():
        bc = getattr(pkg.BoundaryCondition, name)
        st, ra, rb = oracle.cubic_build(x, y, left=left, right=right)
        assert st == oracle.OK
        with wide(force):
            it = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new().boundary(bc)).build()
        a, b = it.strategy.coefficients()
        check_equal(a, ra, f"wide a[{name}] n={n} L={L} {np.dtype(dt)}")
        check_equal(b, rb, f"wide b[{name}] n={n} L={L} {np.dtype(dt)}")
        with wide(0):
            it0 = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new().boundary(bc)).build()
        a0, b0 = it0.strategy.coefficients()
        check_equal(a0, ra, f"serial a[{name}] n={n} L={L} {np.dtype(dt)}")
        check_equal(b0, rb, f"serial b[{name}] n={n} L={L} {np.dtype(dt)}")


def test_wide_build_values_outside_the_divisor_window(pkg):
    """Data spanning 1e-300 .. 1e300 and exact zeros: numerators of the back substitution leave the shared-divisor window
    and take the IEEE division -- same bits."""
    rng = np.random.default_rng(5)
    n, L = 200, 2048
    x = knots("rand", n, rng, np.float64)
    y = rng.uniform(-1.0, 1.0, (n, L)) * 10.0 ** rng.integers(-250, 250, (1, L)).astype(np.float64)
    y[:, :64] = 0.0
    y[50:60, 64:128] = 0.0
    st, ra, rb = oracle.cubic_build(x, y)
    with wide(None):
        it = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
    a, b = it.strategy.coefficients()
    check_equal(a, ra, "wide a, extreme magnitudes")
    check_equal(b, rb, "wide b, extreme magnitudes")
