"""Seeded fuzz of what AUTO takes on LARGE device-resident batches: test_gpu_parity.py's fuzz stays under 3000 queries, where
every shape runs the small-batch kernels; the batch-size thresholds of the plans (query-per-lane kernels with the tables in
LDS, the slope-record kernel, tile grouping, interval grouping, the sorted rounds) only open at 1e5 .. 1e6 queries.  Here
random shapes / element types / strategies / extrapolation modes are evaluated at such sizes through `interp_array`,
`interp_array_into` (strided and sliced outputs included) and the opt-in "rows after the first failure unspecified" form,
each against the CPU oracle, bit for bit, with the first-error cut of interp1d/mod.rs:326-343 / interp2d/mod.rs:287-307 at a
random position.  The plan lines AUTO printed go into the assertion message, so a failure names the kernel."""
import os

import numpy as np
import pytest

import oracle
from test_gpu_parity import check_equal, knots

pytestmark = pytest.mark.gpu


def _tdt(dt):
    import torch
    return torch.float64 if dt == np.float64 else torch.float32


class traced:
    def __init__(self, capfd):
        self.capfd = capfd

    def __enter__(self):
        os.environ["NDI_TRACE_PLAN"] = "1"
        self.capfd.readouterr()
        return self

    def __exit__(self, *a):
        os.environ.pop("NDI_TRACE_PLAN", None)
        self.plans = [ln[11:60] for ln in self.capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]
        if os.environ.get("NDI_FUZZ_PLAN_LOG"):            # (which kernels a run covered: tools / profiles, not an assertion)
            with open(os.environ["NDI_FUZZ_PLAN_LOG"], "a") as f:
                f.write("".join(p + "\n" for p in self.plans))


def _budget_q(rng, L, lo=70_000, hi=1_500_000, points=1.6e7):
    return int(min(rng.integers(lo, hi), max(lo, points // L)))


def _one_1d(pkg, capfd, rng, dt, kind, wide=False):
    import torch
    dev = torch.device("cuda:0")
    n = int(rng.choice([3, 4, 7, 33, 100, 257, 1024, 3000]))
    L = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 12, 16, 24, 32, 64, 100, 128]))
    Q = _budget_q(rng, L)
    if wide:                                           # rows of 512 B .. 16 KiB: the interval-grouped forms' side of AUTO
        n = int(rng.choice([9, 33, 257, 1024, 2500]))
        L = int(rng.choice([128, 130, 256, 512, 1024, 2048]))
        Q = int(min(rng.integers(6 * n, 6 * n + 60_000), 3.2e7 // L))
    x = knots(str(rng.choice(["rand", "jit", "log", "lin"])), n, rng, dt) if n > 4 else np.arange(n).astype(dt)
    y = rng.uniform(-1, 1, (n, L)).astype(dt)
    ext = bool(rng.integers(0, 2))
    span = float(x[-1] - x[0])
    m = 0.4 if ext else 0.0
    q = rng.uniform(x[0] - m * span, x[-1] + m * span, Q).astype(dt)
    if not ext:
        q = np.clip(q, x[0], x[-1])
    q[:min(n, Q)] = x[:min(n, Q)]                      # knot hits, both ends included
    if kind == "linear":
        it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
            .strategy(pkg.Linear.new().extrapolate(ext)).build()
        ref = oracle.interp1d_linear(x, y, q, ext)[2].reshape(Q, L)
        per = False
    else:
        per = bool(rng.integers(0, 4) == 0)
        yy = y.copy()
        if per:
            yy[-1] = yy[0]
        bc = pkg.BoundaryCondition.Periodic if per else [pkg.BoundaryCondition.NotAKnot, pkg.BoundaryCondition.Natural,
                                                          pkg.BoundaryCondition.Clamped][int(rng.integers(0, 3))]
        it = pkg.Interp1DBuilder.new(torch.as_tensor(yy, device=dev)).x(torch.as_tensor(x, device=dev)) \
            .strategy(pkg.CubicSpline.new().extrapolate(ext).boundary(bc)).build()
        # the device build's own tables are the operands (their parity with the oracle's is test_spline_coefficients_bit_exact's
        # subject; the blocked build of many-knot narrow tables is within the bar, not bit-identical)
        a, b = it.strategy.coefficients()
        mode = oracle.EXTRAPOLATE_NO if not ext else (oracle.EXTRAPOLATE_PERIODIC if per else oracle.EXTRAPOLATE_YES)
        ref = oracle.interp1d_cubic(x, yy, a.reshape(n - 1, L), b.reshape(n - 1, L), q, mode)[2].reshape(Q, L)
    what = f"{kind} {np.dtype(dt).name} n={n} L={L} Q={Q} ext={ext} per={per}"
    qd = torch.as_tensor(q, device=dev)
    with traced(capfd) as t:
        got = it.interp_array(qd)
        wide = torch.full((Q + 1, L + 3), -7.0, dtype=_tdt(dt), device=dev)
        it.strategy.interp_array_into(it, qd, wide[1:, :L])   # row stride L + 3, base one row past an aligned address
    check_equal(got.cpu().numpy().reshape(Q, L), ref, f"{what} interp_array {t.plans}")
    check_equal(wide[1:, :L].cpu().numpy(), ref, f"{what} strided into {t.plans}")
    assert bool((wide[1:, L:] == -7.0).all()) and bool((wide[0] == -7.0).all()), f"{what}: wrote outside its rows {t.plans}"
    if ext:
        return
    # first-error cut at a random position, then a second, later failure that must not be the one reported
    bad = int(rng.integers(0, Q)); later = min(Q - 1, bad + int(rng.integers(1, 5000)))
    q2 = q.copy(); q2[bad] = x[-1] + dt(1.0); q2[later] = x[0] - dt(1.0)
    buf = torch.full((Q, L), -3.0, dtype=_tdt(dt), device=dev)
    with traced(capfd) as t:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array_into(torch.as_tensor(q2, device=dev), buf)
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ej:
            it.interp_array(torch.as_tensor(q2, device=dev))
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ek:
            it.interp_array_into(torch.as_tensor(q2, device=dev), torch.empty((Q, L), dtype=_tdt(dt), device=dev),
                                 rows_after_error_unspecified=True)
    assert ei.value.index == ej.value.index == ek.value.index == bad, f"{what}: first failure {ei.value.index}/{ej.value.index}/{ek.value.index} != {bad} {t.plans}"
    g = buf.cpu().numpy()
    assert np.array_equal(g[:bad], ref[:bad]) and np.all(g[bad:] == -3.0), f"{what}: rows around the first failure {t.plans}"


def _one_2d(pkg, capfd, rng, dt, tiles=False):
    import torch
    dev = torch.device("cuda:0")
    nx, ny = (int(v) for v in rng.choice([2, 3, 9, 40, 100, 160, 300, 700], 2))
    C = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64]))
    while nx * ny * C > 6_000_000:
        C = max(1, C // 2)
    Q = _budget_q(rng, C, points=1.2e7)
    if tiles:                                          # rows of 256 B and more on grids past 20 MiB, two queries per cell:
        nx, ny = (int(v) for v in rng.choice([290, 333, 420], 2))   # the tile-grouped form's side of AUTO
        C = int(rng.choice([64, 68, 96]) * 4 // np.dtype(dt).itemsize)
        Q = int(rng.integers(800_000, 1_000_000))
        while nx * ny * C * np.dtype(dt).itemsize < (21 << 20) or Q < 2 * nx * ny:
            nx, ny = min(nx, ny), min(nx, ny)
            C += 16
    x = knots(str(rng.choice(["rand", "jit", "log", "lin"])), nx, rng, dt) if nx > 2 else np.asarray([0.25, 1.5], dtype=dt)
    y = knots(str(rng.choice(["rand", "jit", "log", "lin"])), ny, rng, dt) if ny > 2 else np.asarray([-1.0, 3.0], dtype=dt)
    g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
    ext = bool(rng.integers(0, 2))
    sx, sy = float(x[-1] - x[0]), float(y[-1] - y[0])
    m = 0.3 if ext else 0.0
    qx = rng.uniform(x[0] - m * sx, x[-1] + m * sx, Q).astype(dt); qy = rng.uniform(y[0] - m * sy, y[-1] + m * sy, Q).astype(dt)
    if not ext:
        qx = np.clip(qx, x[0], x[-1]); qy = np.clip(qy, y[0], y[-1])
    k = min(nx, ny)
    qx[:k] = x[:k]; qy[:k] = y[:k]
    qx[k:k + 3] = [x[-1], x[0], x[-1]]; qy[k:k + 3] = [y[-1], y[-1], y[0]]
    ref = oracle.interp2d_bilinear(x, y, g, qx, qy, ext)[3].reshape(Q, C)
    it = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)) \
        .strategy(pkg.Bilinear.new().extrapolate(ext)).build()
    what = f"bilinear {np.dtype(dt).name} {nx}x{ny}x{C} Q={Q} ext={ext}"
    qxd, qyd = torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev)
    with traced(capfd) as t:
        got = it.interp_array(qxd, qyd)
        wide = torch.full((Q + 1, C + 1), -7.0, dtype=_tdt(dt), device=dev)
        it.strategy.interp_array_into(it, qxd, qyd, wide[1:, :C])
        again = it.interp_array(qxd, qyd)                  # (second call: a lazily built copy of the grid is now there)
    check_equal(got.cpu().numpy().reshape(Q, C), ref, f"{what} interp_array {t.plans}")
    check_equal(again.cpu().numpy().reshape(Q, C), ref, f"{what} interp_array again {t.plans}")
    check_equal(wide[1:, :C].cpu().numpy(), ref, f"{what} strided into {t.plans}")
    assert bool((wide[1:, C:] == -7.0).all()) and bool((wide[0] == -7.0).all()), f"{what}: wrote outside its rows {t.plans}"
    if ext:
        return
    bad = int(rng.integers(0, Q))
    axis = int(rng.integers(0, 2))
    qx2, qy2 = qx.copy(), qy.copy()
    (qx2 if axis == 0 else qy2)[bad] = (x if axis == 0 else y)[-1] + dt(1.0)
    later = min(Q - 1, bad + int(rng.integers(1, 5000)))
    if later != bad:
        qx2[later] = x[0] - dt(1.0)
    buf = torch.full((Q, C), -3.0, dtype=_tdt(dt), device=dev)
    with traced(capfd) as t:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array_into(torch.as_tensor(qx2, device=dev), torch.as_tensor(qy2, device=dev), buf)
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ej:
            it.interp_array(torch.as_tensor(qx2, device=dev), torch.as_tensor(qy2, device=dev))
    assert (ei.value.index, ei.value.axis) == (bad, axis) == (ej.value.index, ej.value.axis), \
        f"{what}: first failure {(ei.value.index, ei.value.axis)} / {(ej.value.index, ej.value.axis)} != {(bad, axis)} {t.plans}"
    gg = buf.cpu().numpy()
    assert np.array_equal(gg[:bad], ref[:bad]) and np.all(gg[bad:] == -3.0), f"{what}: rows around the first failure {t.plans}"


@pytest.mark.parametrize("seed", range(16))
def test_auto_large_batches_against_oracle(pkg, capfd, seed):
    rng = np.random.default_rng(61_000 + seed)
    for _ in range(10):
        dt = [np.float64, np.float32][int(rng.integers(0, 2))]
        kind = str(rng.choice(["linear", "cubic", "bilinear", "bilinear"]))
        if kind == "bilinear":
            _one_2d(pkg, capfd, rng, dt)
        else:
            _one_1d(pkg, capfd, rng, dt, kind)


@pytest.mark.parametrize("seed", range(4))
def test_auto_large_batches_wide_rows_and_tiles(pkg, capfd, seed):
    rng = np.random.default_rng(62_000 + seed)
    for _ in range(3):
        dt = [np.float64, np.float32][int(rng.integers(0, 2))]
        _one_1d(pkg, capfd, rng, dt, str(rng.choice(["linear", "cubic", "cubic"])), wide=True)
    _one_2d(pkg, capfd, rng, [np.float64, np.float32][seed % 2], tiles=True)
