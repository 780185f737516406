"""GPU parity of eval_slopes2d_kernel (round 6): short rows on grids that do not fit LDS through the SLOPE-RECORD copy of the
grid (slope_pack_kernel: {z, m = (z[xi+1] - z[xi]) / (x[xi+1] - x[xi])} per grid point -- linear.rs:33's division done once per
grid point, so a query costs one division per value instead of three and reads one contiguous run of 4 L values) -- the
reference's 100 x 100 x 5 bench grid (benches/bench_interp2d.rs:87-92) -- against the CPU oracle, bit for bit, through the
C ABI.  Bilinear = bilinear.rs:64-99.  The kernel is forced with NDI_SLOPES2D_KERNEL=1 (conftest sets NDI_TUNE_LIVE: the knob
is read per call; NDI_LANES2D_KERNEL=0 keeps the LDS-resident-grid kernel from taking the small grids first) and the plan
line is asserted.  Covered: both element types, both source layouts of the grid (plain: grids under 160 KB; pair-packed:
above), 1 to 8 (f64) / 12 (f32) values per grid point, odd counts included, ragged batches, strided output, extrapolation,
the first-error cut (x before y), interp_array's own range test (no pre-pass), and what AUTO takes on the bench shape."""
import os

import numpy as np
import pytest

import oracle
from test_gpu_parity import check_equal, knots

pytestmark = pytest.mark.gpu


class forced:
    def __init__(self, capfd, value="1"):
        self.capfd, self.value = capfd, value

    def __enter__(self):
        os.environ["NDI_SLOPES2D_KERNEL"] = self.value
        os.environ["NDI_LANES2D_KERNEL"] = "0"
        os.environ["NDI_TRACE_PLAN"] = "1"
        self.capfd.readouterr()
        return self

    def __exit__(self, *a):
        for k in ("NDI_SLOPES2D_KERNEL", "NDI_LANES2D_KERNEL", "NDI_TRACE_PLAN"):
            os.environ.pop(k, None)
        self.plans = [ln for ln in self.capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]


def _tdt(dt):
    import torch
    return torch.float64 if dt == np.float64 else torch.float32


SHAPES = [("rand", "rand", 100, 100, 5), ("rand", "jit", 40, 57, 5), ("lin", "lin", 64, 64, 8), ("log", "rand", 17, 300, 2),
          ("rand", "rand", 2, 2, 4), ("rand", "lin", 130, 97, 7), ("rand", "rand", 33, 41, 6), ("rand", "rand", 3, 90, 3),
          ("rand", "rand", 150, 160, 1), ("rand", "log", 210, 120, 7)]


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("kx,ky,nx,ny,C", SHAPES)
def test_slopes_2d_bit_exact(pkg, capfd, dt, kx, ky, nx, ny, C):
    import torch
    dev = torch.device("cuda:0")
    if dt == np.float32:
        C = {5: 5, 8: 12, 2: 10, 4: 4, 7: 9 if nx == 130 else 11, 6: 7, 3: 3, 1: 1}[C]      # 4 .. 48-byte rows
    rng = np.random.default_rng(nx * 13 + ny * 7 + C)
    Q = 70_003
    x = knots(kx, nx, rng, dt) if nx > 2 else np.asarray([0.25, 1.5], dtype=dt)
    y = knots(ky, ny, rng, dt) if ny > 2 else np.asarray([-1.0, 3.0], dtype=dt)
    g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
    qx = rng.uniform(x[0], x[-1], Q).astype(dt); qy = rng.uniform(y[0], y[-1], Q).astype(dt)
    k = min(nx, ny)
    qx[:k] = x[:k]; qy[:k] = y[:k]                     # grid-point hits
    qx[k:k + 3] = [x[-1], x[0], x[-1]]; qy[k:k + 3] = [y[-1], y[-1], y[0]]   # the last cell: the segment at the end of the grid
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
    assert len(f.plans) == 2 and all(" slopes2d L=" in p for p in f.plans), f.plans
    check_equal(out.cpu().numpy(), ref, f"slopes2d {nx}x{ny}x{C}")
    check_equal(wide[:, :C].cpu().numpy(), ref, f"slopes2d strided {nx}x{ny}x{C}")
    assert bool((wide[:, C:] == -9.0).all())
    # extrapolation
    ex = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)) \
        .strategy(pkg.Bilinear.new().extrapolate(True)).build()
    sx, sy = x[-1] - x[0], y[-1] - y[0]
    qx2 = rng.uniform(x[0] - sx, x[-1] + sx, Q).astype(dt); qy2 = rng.uniform(y[0] - sy, y[-1] + sy, Q).astype(dt)
    with forced(capfd) as f:
        got2 = ex.interp_array(torch.as_tensor(qx2, device=dev), torch.as_tensor(qy2, device=dev)).cpu().numpy()
    assert " slopes2d L=" in f.plans[0], f.plans
    check_equal(got2.reshape(Q, C), oracle.interp2d_bilinear(x, y, g, qx2, qy2, True)[3].reshape(Q, C), "slopes2d extrapolate")
    # the first-error cut (x before y for the same query: bilinear.rs:71-80)
    qy[31_000] = y[-1] + 1; qx[31_000] = x[0] - 1; qx[35_000] = x[0] - 1
    buf = torch.full((Q, C), -3.0, dtype=_tdt(dt), device=dev)
    with forced(capfd) as f:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array_into(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), buf)
    assert (ei.value.index, ei.value.axis) == (31_000, 0) and " slopes2d L=" in f.plans[0], f.plans
    got = buf.cpu().numpy()
    assert np.array_equal(got[:31_000], ref[:31_000]) and np.all(got[31_000:] == -3.0)
    with forced(capfd) as f:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    assert (ei.value.index, ei.value.axis) == (31_000, 0) and "prepass=0" in f.plans[0], f.plans
    qx[31_000] = x[0]
    with forced(capfd):
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
    assert (ei.value.index, ei.value.axis) == (31_000, 1)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_slopes_2d_auto_takes_the_bench_shape(pkg, capfd, dt):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    x = np.cumsum(rng.uniform(0.5, 1.5, 100)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, 100)).astype(dt)
    g = rng.uniform(0, 1, (100, 100, 5)).astype(dt)
    it = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    Q = 6_000_000
    qx = torch.as_tensor(rng.uniform(x[0], x[-1], Q).astype(dt), device=dev)
    qy = torch.as_tensor(rng.uniform(y[0], y[-1], Q).astype(dt), device=dev)
    os.environ["NDI_TRACE_PLAN"] = "1"
    capfd.readouterr()
    try:
        got = it.interp_array(qx, qy).cpu().numpy()
    finally:
        os.environ.pop("NDI_TRACE_PLAN", None)
    plans = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]
    assert any(" slopes2d L=5 " in p for p in plans), plans
    sel = rng.choice(Q, 50_000, replace=False)
    ref = oracle.interp2d_bilinear(x, y, g, qx.cpu().numpy()[sel], qy.cpu().numpy()[sel])[3].reshape(-1, 5)
    check_equal(got.reshape(Q, 5)[sel], ref, "slopes2d auto 100x100x5")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_slopes_2d_first_use_inside_the_ring_and_sharded(pkg, capfd, dt):
    """The slope-record copy is built lazily, on the stream that first needs it.  Here the handle's FIRST evaluation is a ring
    evaluation (Interp2D::interp_array through ndi_interp2d_eval_ring: search / pre-pass of chunk k + 1 on the side stream,
    evaluation on the caller's), chunks large enough for AUTO to take the slope-record kernel; then one sharded call over two
    replicas on the device, with a failing query (first-error over the whole batch: interp2d/mod.rs:297-306)."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(21)
    nx, ny, C, Q, chunk = 120, 90, 5, 800_000, 200_000
    x = knots("rand", nx, rng, dt); y = knots("jit", ny, rng, dt)
    g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
    qx = rng.uniform(x[0], x[-1], Q).astype(dt); qy = rng.uniform(y[0], y[-1], Q).astype(dt)
    ref = oracle.interp2d_bilinear(x, y, g, qx, qy)[3].reshape(Q, C)
    it = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    got = np.zeros_like(ref)

    def consumer(c, rows):
        got[c.q_begin:c.q_begin + c.q_count] = rows.cpu().numpy()
    os.environ["NDI_TRACE_PLAN"] = "1"
    capfd.readouterr()
    try:
        slots = [torch.empty((chunk, C), dtype=_tdt(dt), device=dev) for _ in range(2)]
        it.interp_array_ring(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), chunk, consumer, slots=slots)
    finally:
        os.environ.pop("NDI_TRACE_PLAN", None)
    plans = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]
    assert sum(" slopes2d L=5 " in p for p in plans) == 4, plans
    check_equal(got, ref, "slopes2d, first use inside the ring")
    reps = [it] + it.replicate([0])
    out = np.full_like(ref, -2.0)
    pkg.sharding.interp_array_sharded(reps, qx, qy, out=out)
    check_equal(out, ref, "slopes2d, sharded over two replicas")
    qx2 = qx.copy(); qx2[600_123] = x[-1] + 1
    out2 = np.full_like(ref, -2.0)
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        pkg.sharding.interp_array_sharded(reps, qx2, qy, out=out2)
    assert (ei.value.index, ei.value.axis) == (600_123, 0)
    assert np.array_equal(out2[:600_123], ref[:600_123]) and np.all(out2[600_123:] == -2.0)


def test_slopes_copy_is_released_by_trim_and_rebuilt(pkg, capfd):
    """ndi_interp2d_trim releases the lazily built slope-record copy (2 x the grid) when it is idle; the next batch rebuilds
    it and gives the same bits."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    nx, ny, C, Q = 400, 300, 4, 200_000
    x = knots("rand", nx, rng, np.float64); y = knots("rand", ny, rng, np.float64)
    g = rng.uniform(-1, 1, (nx, ny, C))
    qx = rng.uniform(x[0], x[-1], Q); qy = rng.uniform(y[0], y[-1], Q)
    it = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    ref = oracle.interp2d_bilinear(x, y, g, qx, qy)[3].reshape(Q, C)
    qxd, qyd = torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    with forced(capfd) as f:
        got = it.interp_array(qxd, qyd).cpu().numpy()
    assert " slopes2d L=4 " in f.plans[0], f.plans
    check_equal(got.reshape(Q, C), ref, "before trim")
    it.strategy.finish()
    torch.cuda.synchronize()
    held = free0 - torch.cuda.mem_get_info(dev)[0]
    it.strategy.trim()
    torch.cuda.synchronize()
    after = free0 - torch.cuda.mem_get_info(dev)[0]
    copy_bytes = (nx - 1) * ny * 2 * C * 8
    assert held - after >= copy_bytes // 2, (held, after, copy_bytes)       # the copy (7.7 MB here) is gone
    with forced(capfd) as f:
        got = it.interp_array(qxd, qyd).cpu().numpy()
    assert " slopes2d L=4 " in f.plans[0], f.plans
    check_equal(got.reshape(Q, C), ref, "after trim")
