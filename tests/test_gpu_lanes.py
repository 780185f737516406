"""GPU parity of the query-per-lane kernels with LDS-resident tables (round 5): eval_lanes_kernel (1-D: the reference's
own bench shapes -- scalar data and (100, 5) on 100 knots, benches/bench_interp1d.rs:12-47, 82-122) and
eval_lanes2d_kernel (2-D: grids that fit LDS -- the 100 x 100 scalar grid of benches/bench_interp2d.rs:12-18), against
the CPU oracle, bit for bit, through the C ABI.  The kernels are forced with NDI_LANES_KERNEL=1 / NDI_LANES2D_KERNEL=1
(conftest sets NDI_TUNE_LIVE, so the knobs are read per call) and the plan line the library prints under
NDI_TRACE_PLAN is asserted, so a silent fall-back to another kernel fails the test.  What is covered: both strategies and
element types, one / several values per row, every search (pyramid, O(1) guess, bucket index), the vector and the
one-query-per-lane scalar forms (unaligned / strided buffers), ragged tails, extrapolation, periodic wrap, NaN, the
first-error cut, and -- without forcing -- that AUTO takes them at the sizes the bench uses.
Linear = linear.rs:73-98, CubicSpline = cubic_spline.rs:791-830, Bilinear = bilinear.rs:64-99."""
import os

import numpy as np
import pytest

import oracle
from test_gpu_parity import check_equal, knots

pytestmark = pytest.mark.gpu


class forced:
    """Force (1) / forbid (0) the lanes kernels for the calls inside the block and capture the plan lines."""

    def __init__(self, capfd, value="1"):
        self.capfd, self.value = capfd, value

    def __enter__(self):
        for k in ("NDI_LANES_KERNEL", "NDI_LANES2D_KERNEL"):
            os.environ[k] = self.value
        os.environ["NDI_TRACE_PLAN"] = "1"
        self.capfd.readouterr()
        return self

    def __exit__(self, *a):
        for k in ("NDI_LANES_KERNEL", "NDI_LANES2D_KERNEL", "NDI_TRACE_PLAN"):
            os.environ.pop(k, None)
        self.plans = [ln for ln in self.capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]


def _tdt(dt):
    import torch
    return torch.float64 if dt == np.float64 else torch.float32


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("kind,n,L", [("rand", 100, 1), ("rand", 100, 5), ("lin", 100, 1), ("jit", 1024, 1), ("log", 333, 2),
                                      ("rand", 37, 3), ("rand", 200, 8), ("lin", 64, 7), ("rand", 3, 1), ("rand", 700, 4)])
def test_lanes_1d_bit_exact(pkg, capfd, dt, kind, n, L):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n * 31 + L)
    Q = 70_003                      # ragged: not a multiple of 64, of 4 or of 2
    x = knots(kind, n, rng, dt)
    y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    q[:4] = [x[0], x[-1], x[n // 2], np.nextafter(x[-1], x[0])]
    q[4:4 + n] = x                  # every knot itself (t = 0: the shared-divisor window's IEEE fall-back)
    st, a, b = oracle.cubic_build(x, y)
    assert st == oracle.OK
    ref = {"cubic": oracle.interp1d_cubic(x, y, a, b, q)[2].reshape(Q, L),
           "linear": oracle.interp1d_linear(x, y, q)[2].reshape(Q, L)}
    yd, xd = torch.as_tensor(y, device=dev), torch.as_tensor(x, device=dev)
    its = {"cubic": pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build(),
           "linear": pkg.Interp1DBuilder.new(yd).x(xd).build()}
    qd = torch.as_tensor(q, device=dev)
    for name, it in its.items():
        for path in (pkg.PATH_AUTO, pkg.PATH_GATHER):
            it.strategy.path = path
            with forced(capfd) as f:
                out = torch.full((Q, L), -9.0, dtype=_tdt(dt), device=dev)
                it.interp_array_into(qd, out)
                # queries one element off 16-byte alignment, output rows strided: the one-query-per-lane scalar form
                qo = torch.empty(Q + 1, dtype=_tdt(dt), device=dev)[1:]
                qo.copy_(qd)
                wide = torch.full((Q, L + 3), -9.0, dtype=_tdt(dt), device=dev)
                it.strategy.interp_array_into(it, qo, wide[:, :L])
            host = it.interp_array(q)      # host arrays in and out: the staged small-row path (PCIe-bound, not this kernel)
            assert len(f.plans) == 2 and all(" lanes L=" in p for p in f.plans), f.plans
            if L == 1:
                assert "qpl=%d" % (16 // np.dtype(dt).itemsize) in f.plans[0] and "qpl=1" in f.plans[1], f.plans
            check_equal(out.cpu().numpy(), ref[name], f"{name} lanes n={n} L={L}")
            check_equal(wide[:, :L].cpu().numpy(), ref[name], f"{name} lanes strided n={n} L={L}")
            assert bool((wide[:, L:] == -9.0).all())
            check_equal(np.asarray(host).reshape(Q, L), ref[name], f"{name} lanes host n={n} L={L}")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("L", [1, 5])
def test_lanes_1d_modes_and_first_error(pkg, capfd, dt, L):
    """Extrapolation with the end intervals, periodic wrap (cubic_spline.rs:805-809), NaN -> OutOfBounds without and
    NaN-query panic with extrapolation, and the first-error cut: rows before the first failing query are written, the rows
    at and after it keep their old contents (interp1d/mod.rs:334-342)."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(17 + L)
    n, Q = 100, 66_001
    x = knots("rand", n, rng, dt)
    y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
    span = x[-1] - x[0]
    q = rng.uniform(x[0] - span, x[-1] + span, Q).astype(dt)
    yd, xd = torch.as_tensor(y, device=dev), torch.as_tensor(x, device=dev)
    st, a, b = oracle.cubic_build(x, y)
    ex_c = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new().extrapolate(True)).build()
    ex_l = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.Linear.new().extrapolate(True)).build()
    with forced(capfd) as f:
        got_c = ex_c.interp_array(torch.as_tensor(q, device=dev)).cpu().numpy()
        got_l = ex_l.interp_array(torch.as_tensor(q, device=dev)).cpu().numpy()
    assert len(f.plans) == 2 and all(" lanes L=" in p for p in f.plans), f.plans
    check_equal(got_c.reshape(Q, L), oracle.interp1d_cubic(x, y, a, b, q, oracle.EXTRAPOLATE_YES)[2].reshape(Q, L), "cubic extrapolate")
    check_equal(got_l.reshape(Q, L), oracle.interp1d_linear(x, y, q, True)[2].reshape(Q, L), "linear extrapolate")
    # periodic
    yp = y.copy(); yp[-1] = yp[0]
    stp, ap, bp = oracle.cubic_build(x, yp, periodic=True)
    assert stp == oracle.OK
    per = pkg.Interp1DBuilder.new(torch.as_tensor(yp, device=dev)).x(xd) \
        .strategy(pkg.CubicSpline.new().boundary(pkg.BoundaryCondition.Periodic).extrapolate(True)).build()
    with forced(capfd) as f:
        got_p = per.interp_array(torch.as_tensor(q, device=dev)).cpu().numpy()
    assert len(f.plans) == 1 and " lanes L=" in f.plans[0], f.plans
    check_equal(got_p.reshape(Q, L), oracle.interp1d_cubic(x, yp, ap, bp, q, oracle.EXTRAPOLATE_PERIODIC)[2].reshape(Q, L),
                "cubic periodic")
    # first error / NaN
    noex = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
    qi = rng.uniform(x[0], x[-1], Q).astype(dt)
    ref = oracle.interp1d_cubic(x, y, a, b, qi)[2].reshape(Q, L)
    qi[40_001] = np.nan; qi[50_000] = x[-1] + 1
    out = torch.full((Q, L), -3.0, dtype=_tdt(dt), device=dev)
    with forced(capfd) as f:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            noex.interp_array_into(torch.as_tensor(qi, device=dev), out)
    assert ei.value.index == 40_001 and " lanes L=" in f.plans[0], f.plans
    got = out.cpu().numpy()
    assert np.array_equal(got[:40_001], ref[:40_001]) and np.all(got[40_001:] == -3.0) and "prepass=1" in f.plans[0]
    # interp_array (the output is the call's own and dropped on Err: NDI_EVAL_FRESH_OUTPUT): no range pre-pass, the
    # kernel's own test still reports the LOWEST failing query and its value
    with forced(capfd) as f:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            noex.interp_array(torch.as_tensor(qi, device=dev))
    assert ei.value.index == 40_001 and "prepass=0" in f.plans[0], f.plans
    with forced(capfd) as f:
        ok = noex.interp_array(torch.as_tensor(q[(q >= x[0]) & (q <= x[-1])], device=dev)).cpu().numpy()
    assert "prepass=0" in f.plans[0]
    qin = q[(q >= x[0]) & (q <= x[-1])]
    check_equal(ok.reshape(qin.size, L), oracle.interp1d_cubic(x, y, a, b, qin)[2].reshape(qin.size, L), "fresh output")
    with forced(capfd):
        with pytest.raises(pkg.Panic):
            ex_c.interp_array(torch.as_tensor(qi, device=dev))


def test_lanes_1d_auto_takes_the_bench_shapes(pkg, capfd):
    """Without any knob AUTO takes the lanes kernel for the reference's shapes at the bench's batch sizes and leaves
    batches too small to pay for the staging, rows over 64 bytes and table sets beyond LDS to the other kernels."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)

    def plan_of(n, L, Q, dt=np.float64):
        x = knots("rand", n, rng, dt)
        y = rng.uniform(0, 1, (n, L)).astype(dt)
        it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
            .strategy(pkg.CubicSpline.new().reference_order(True)).build()   # (4096 x 8 would take the blocked build)
        q = torch.as_tensor(rng.uniform(x[0], x[-1], Q).astype(dt), device=dev)
        os.environ["NDI_TRACE_PLAN"] = "1"
        capfd.readouterr()
        try:
            got = it.interp_array(q).cpu().numpy()
        finally:
            os.environ.pop("NDI_TRACE_PLAN", None)
        plans = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]
        st, a, b = oracle.cubic_build(x, y)
        check_equal(got.reshape(Q, L), oracle.interp1d_cubic(x, y, a, b, q.cpu().numpy())[2].reshape(Q, L), f"auto {n}x{L}")
        return " ".join(plans)
    assert " lanes L=1 " in plan_of(100, 1, 20_000_000)
    assert " lanes L=5 " in plan_of(100, 5, 10_000_000)
    assert " lanes " not in plan_of(100, 5, 70_000)          # too small to pay for a staging pass per workgroup
    assert " lanes " not in plan_of(100, 24, 3_000_000)      # 192-byte rows
    assert " lanes " not in plan_of(4096, 8, 3_000_000)      # 1 MiB of records: not an LDS-resident table set


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("kx,ky,nx,ny,C", [("rand", "jit", 100, 100, 1), ("lin", "lin", 100, 100, 1), ("rand", "rand", 40, 57, 5),
                                           ("log", "rand", 17, 300, 2), ("rand", "lin", 2, 2, 1), ("rand", "rand", 60, 33, 8)])
def test_lanes_2d_bit_exact(pkg, capfd, dt, kx, ky, nx, ny, C):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(nx * 13 + ny * 7 + C)
    Q = 70_003
    x = knots(kx, nx, rng, dt) if nx > 2 else np.asarray([0.25, 1.5], dtype=dt)
    y = knots(ky, ny, rng, dt) if ny > 2 else np.asarray([-1.0, 3.0], dtype=dt)
    g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
    qx = rng.uniform(x[0], x[-1], Q).astype(dt); qy = rng.uniform(y[0], y[-1], Q).astype(dt)
    k = min(nx, ny)
    qx[:k] = x[:k]; qy[:k] = y[:k]                     # grid-point hits
    qx[k:k + 3] = [x[-1], x[0], x[-1]]; qy[k:k + 3] = [y[-1], y[-1], y[0]]
    ref = oracle.interp2d_bilinear(x, y, g, qx, qy)[3].reshape(Q, C)
    it = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    qxd, qyd = torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev)
    with forced(capfd) as f:
        out = torch.full((Q, C), -9.0, dtype=_tdt(dt), device=dev)
        it.interp_array_into(qxd, qyd, out)
        qxo = torch.empty(Q + 1, dtype=_tdt(dt), device=dev)[1:]
        qxo.copy_(qxd)
        wide = torch.full((Q, C + 2), -9.0, dtype=_tdt(dt), device=dev)
        it.strategy.interp_array_into(it, qxo, qyd, wide[:, :C])
    host = it.interp_array(qx, qy)         # host arrays in and out: the staged small-row path
    assert len(f.plans) == 2 and all(" lanes2d L=" in p for p in f.plans), f.plans
    check_equal(out.cpu().numpy(), ref, f"lanes2d {nx}x{ny}x{C}")
    check_equal(wide[:, :C].cpu().numpy(), ref, f"lanes2d strided {nx}x{ny}x{C}")
    assert bool((wide[:, C:] == -9.0).all())
    check_equal(np.asarray(host).reshape(Q, C), ref, f"lanes2d host {nx}x{ny}x{C}")
    # extrapolation and the first-error cut (x before y for the same query: bilinear.rs:71-80)
    ex = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)) \
        .strategy(pkg.Bilinear.new().extrapolate(True)).build()
    sx, sy = x[-1] - x[0], y[-1] - y[0]
    qx2 = rng.uniform(x[0] - sx, x[-1] + sx, Q).astype(dt); qy2 = rng.uniform(y[0] - sy, y[-1] + sy, Q).astype(dt)
    with forced(capfd) as f:
        got2 = ex.interp_array(torch.as_tensor(qx2, device=dev), torch.as_tensor(qy2, device=dev)).cpu().numpy()
    assert " lanes2d L=" in f.plans[0], f.plans
    check_equal(got2.reshape(Q, C), oracle.interp2d_bilinear(x, y, g, qx2, qy2, True)[3].reshape(Q, C), "lanes2d extrapolate")
    qy[31_000] = y[-1] + 1; qx[31_000] = x[0] - 1; qx[35_000] = x[0] - 1
    buf = torch.full((Q, C), -3.0, dtype=_tdt(dt), device=dev)
    with forced(capfd) as f:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array_into(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), buf)
    assert (ei.value.index, ei.value.axis) == (31_000, 0) and " lanes2d L=" in f.plans[0], f.plans
    got = buf.cpu().numpy()
    assert np.array_equal(got[:31_000], ref[:31_000]) and np.all(got[31_000:] == -3.0)
    # interp_array: fresh output, no pre-pass; x is still reported before y for the same query, y alone when only y fails
    with forced(capfd) as f:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    assert (ei.value.index, ei.value.axis) == (31_000, 0) and "prepass=0" in f.plans[0], f.plans
    qx[31_000] = x[0]
    with forced(capfd):
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    assert (ei.value.index, ei.value.axis) == (31_000, 1)


def test_lanes_2d_auto_takes_the_bench_shape(pkg, capfd):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    x = np.cumsum(rng.uniform(0.5, 1.5, 100)); y = np.cumsum(rng.uniform(0.5, 1.5, 100))
    g = rng.uniform(0, 1, (100, 100))
    it = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    Q = 12_000_000
    qx = torch.as_tensor(rng.uniform(x[0], x[-1], Q), device=dev); qy = torch.as_tensor(rng.uniform(y[0], y[-1], Q), device=dev)
    os.environ["NDI_TRACE_PLAN"] = "1"
    capfd.readouterr()
    try:
        got = it.interp_array(qx, qy).cpu().numpy()
    finally:
        os.environ.pop("NDI_TRACE_PLAN", None)
    plans = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]
    assert any(" lanes2d L=1 qpl=2" in p for p in plans), plans
    sel = rng.choice(Q, 50_000, replace=False)
    ref = oracle.interp2d_bilinear(x, y, g.reshape(100, 100, 1), qx.cpu().numpy()[sel], qy.cpu().numpy()[sel])[3].reshape(-1)
    check_equal(got.reshape(-1)[sel], ref, "lanes2d auto 100x100")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_sliced_output_views_with_odd_lane_counts(pkg, capfd, dt):
    """`out[1:]` of a (Q + 1, 5) buffer is contiguous, T-aligned and NOT 16-byte aligned -- the C ABI accepts it;
    the kernels that store rows as 16-byte vectors (eval_lanes_kernel, eval_lanes2d_kernel, eval_slopes2d_kernel) must
    take their per-element store path for it.  1-D (100, 5) and 2-D 40 x 30 x 5 / 150 x 140 x 5, bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    Q, L = 70_001, 5
    x = knots("rand", 100, rng, dt)
    y = rng.uniform(-1, 1, (100, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(x, y)
    ref = oracle.interp1d_cubic(x, y, a, b, q)[2]
    big = torch.full((Q + 1, L), -5.0, dtype=_tdt(dt), device=dev)
    view = big[1:]
    assert view.data_ptr() % 16 != 0 and view.is_contiguous()
    with forced(capfd) as f:
        it.interp_array_into(torch.as_tensor(q, device=dev), view)
    assert " lanes L=5" in f.plans[0], f.plans
    check_equal(view.cpu().numpy(), ref, "1-D lanes into out[1:]")
    assert bool((big[0] == -5.0).all())
    for nx, ny, plan, env in ((40, 30, " lanes2d L=5", {}), (150, 140, " slopes2d L=5", {"NDI_SLOPES2D_KERNEL": "1"})):
        gx = knots("rand", nx, rng, dt); gy = knots("jit", ny, rng, dt)
        g = rng.uniform(-1, 1, (nx, ny, L)).astype(dt)
        qx = rng.uniform(gx[0], gx[-1], Q).astype(dt); qy = rng.uniform(gy[0], gy[-1], Q).astype(dt)
        it2 = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(gx, device=dev)).y(torch.as_tensor(gy, device=dev)).build()
        big = torch.full((Q + 1, L), -5.0, dtype=_tdt(dt), device=dev)
        os.environ.update(env)
        try:
            with forced(capfd) as f:
                it2.interp_array_into(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), big[1:])
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert plan in f.plans[0], f.plans
        check_equal(big[1:].cpu().numpy(), oracle.interp2d_bilinear(gx, gy, g, qx, qy)[3].reshape(Q, L), f"2-D {plan} into out[1:]")
        assert bool((big[0] == -5.0).all())
