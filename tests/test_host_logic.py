"""CPU: host-side logic of the mirror -- builder validation order and error variants, monotonic_prop,
buffer-shape arithmetic, sharding arithmetic.  No compute calls."""
import numpy as np
import pytest


def test_monotonic_prop_cases(pkg, refvec):
    for case in refvec["monotonic_prop"]:
        for dt in (np.float64, np.float32, np.int32):
            v = np.array(case["v"]).astype(dt) if dt != np.int32 or all(float(e).is_integer() for e in case["v"]) else None
            if v is None:
                continue
            got = repr(pkg.monotonic_prop(v)).replace(" ", "")
            assert got == case["expect"].replace(" ", ""), (case, dt)
    assert pkg.monotonic_prop(np.array([0.0, np.nan, 2.0])) != pkg.Monotonic.Rising(True)
    assert pkg.monotonic_prop(np.array([np.nan, 1.0, 2.0])) != pkg.Monotonic.Rising(True)
    # reversed view of a falling array is rising (vector_extensions.rs:379-384)
    assert pkg.monotonic_prop(np.array([5, 4, 3, 2, 1])[::-1]) == pkg.Monotonic.Rising(True)


def test_monotonic_c_abi_equals_generic(pkg):
    from ndarray_interp_amd.vector_extensions import _monotonic_generic
    rng = np.random.default_rng(0)
    for _ in range(300):
        n = int(rng.integers(0, 9))
        v = rng.integers(0, 4, n).astype(np.float64)
        assert pkg.monotonic_prop(v) == _monotonic_generic(v), v


def test_builder1d_errors_integer_data(pkg):
    # tests/interp1d.rs:123-140 uses i32 data: validation happens before any strategy/device work
    with pytest.raises(pkg.BuilderError.NotEnoughData, match="at least 2 data points"):
        pkg.Interp1DBuilder.new(np.array([1])).build()
    with pytest.raises(pkg.BuilderError.ShapeError, match="Got x: 3, data: 2"):
        pkg.Interp1DBuilder.new(np.array([1, 2])).x(np.array([1, 2, 3])).build()
    with pytest.raises(pkg.BuilderError.Monotonic):
        pkg.Interp1DBuilder.new(np.array([1, 2, 3])).x(np.array([1, 2, 2])).build()
    with pytest.raises(pkg.BuilderError.NotEnoughData, match="at least 3 data points"):
        pkg.Interp1D.builder(np.array([1.0, 2.0])).strategy(pkg.CubicSpline.new()).build()
    with pytest.raises(pkg.BuilderError.ShapeError, match="data dimension is 0"):
        pkg.Interp1DBuilder.new(np.array(1.0)).build()


def test_builder1d_check_order(pkg):
    # NotEnoughData before Monotonic before ShapeError (interp1d/mod.rs:454-471)
    with pytest.raises(pkg.BuilderError.NotEnoughData):
        pkg.Interp1DBuilder.new(np.array([1.0])).x(np.array([2.0, 1.0])).build()
    with pytest.raises(pkg.BuilderError.Monotonic):
        pkg.Interp1DBuilder.new(np.array([1.0, 2.0])).x(np.array([2.0, 1.0, 0.0])).build()


def test_builder2d_errors(pkg):
    B = pkg.Interp2D.builder
    for data in ([[1]], [[1, 2]], [[1], [2]]):
        with pytest.raises(pkg.BuilderError.NotEnoughData):
            B(np.array(data)).build()
    sq = np.array([[1, 2], [3, 4]])
    for ax in ([1], [1, 2, 3]):
        with pytest.raises(pkg.BuilderError.ShapeError, match="x-axis"):
            B(sq).x(np.array(ax)).build()
        with pytest.raises(pkg.BuilderError.ShapeError, match="y-axis"):
            B(sq).y(np.array(ax)).build()
    with pytest.raises(pkg.BuilderError.Monotonic, match="x-axis"):
        B(sq).x(np.array([2, 2])).build()
    with pytest.raises(pkg.BuilderError.Monotonic, match="y-axis"):
        B(sq).y(np.array([2, 2])).build()
    with pytest.raises(pkg.BuilderError.ShapeError, match="at least 2"):
        B(np.array([1.0, 2.0])).build()


def test_individual_boundary_shape_error(pkg):
    # tests/cubic_spline_strat.rs:413-440
    y = np.array([[0.5, 1.0], [0.0, 1.5], [3.0, 0.5]])
    R = pkg.RowBoundary
    with pytest.raises(pkg.BuilderError.ShapeError, match=r"Expected: \[1, 2\], got: \[1, 3\]"):
        pkg.Interp1DBuilder.new(y).strategy(pkg.CubicSpline.new().boundary(
            pkg.BoundaryCondition.Individual([[R.Natural, R.Clamped, R.NotAKnot]]))).build()
    with pytest.raises(pkg.BuilderError.ShapeError, match=r"Expected: \[1, 2\], got: \[2, 2\]"):
        pkg.Interp1DBuilder.new(y).strategy(pkg.CubicSpline.new().boundary(
            pkg.BoundaryCondition.Individual([[R.Natural, R.NotAKnot], [R.Natural, R.NotAKnot]]))).build()


def test_non_float_dtype_is_refused_by_device_strategies(pkg):
    with pytest.raises(TypeError, match="float32/float64"):
        pkg.Interp1DBuilder.new(np.array([1, 2, 3])).build()


def test_shard_bounds(pkg):
    sb = pkg.sharding.shard_bounds
    for nq in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 3, 8):
            blocks = [sb(nq, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == nq
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
