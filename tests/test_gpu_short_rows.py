"""GPU parity of the short-row 1-D formulations (rows under 256 vectors; the reference's own data shapes -- scalar
data, (100, 5): benches/bench_interp1d.rs:82-122) against the CPU oracle, bit for bit, through the C ABI:

  * flat     locate_kernel + eval_flat_kernel (round 3's only form for these shapes)
  * fused    eval_fused_kernel: query order, search fused in; tables from L2 (plain / interval-packed) or from LDS
             ({y, a, b}, or {y, k} with a / b re-formed per item by the build's own operations)
  * grouped  eval_bucketed_short_kernel: grouped by interval, operand vectors in registers

The variants are selected with the library's tuning knobs (NDI_TUNE_LIVE is set by conftest, so one process can
switch them); `auto` is what ships.  Linear = linear.rs:73-98, CubicSpline = cubic_spline.rs:791-830."""
import os

import numpy as np
import pytest

import oracle
from test_gpu_parity import check_equal, knots

pytestmark = pytest.mark.gpu

KNOBS = ("NDI_SHORT_MODE", "NDI_FUSED_UNR", "NDI_FUSED_TB", "NDI_FUSED_LDS", "NDI_FUSED_WGS", "NDI_SHORT_CQ",
         "NDI_FUSED_PACK", "NDI_SHORT_ROWB")


class knobs:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k in KNOBS:
            os.environ.pop(k, None)
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k in KNOBS:
            os.environ.pop(k, None)


def variants(pkg):
    return [
        ("auto", pkg.PATH_AUTO, {}),
        ("gather", pkg.PATH_GATHER, {}),
        ("bucketed", pkg.PATH_BUCKETED, {}),
        ("flat", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=1)),
        ("fused_l2_u1", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_PACK=0, NDI_FUSED_UNR=1)),
        ("fused_l2_u4_tb512", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_PACK=0, NDI_FUSED_UNR=4,
                                                     NDI_FUSED_TB=512)),
        ("fused_pack", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_PACK=1)),
        ("fused_lds", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=1)),
        ("fused_lds_tb256_u4", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=1, NDI_FUSED_TB=256, NDI_FUSED_UNR=4)),
        # {y, k} in LDS, a / b re-formed per item (CubicSpline; Linear falls back to {y})
        ("fused_lds_yk", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=2)),
        ("fused_lds_yk_tb512_u1", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=2, NDI_FUSED_TB=512, NDI_FUSED_UNR=1)),
        ("grouped_cq16", pkg.PATH_BUCKETED, dict(NDI_SHORT_MODE=3, NDI_SHORT_CQ=16)),
        ("grouped_cq64", pkg.PATH_AUTO, dict(NDI_SHORT_MODE=3, NDI_SHORT_CQ=64)),
    ]


def _device_eval(pkg, interp, q, L, path, tdt, fill=None):
    import torch
    dev = torch.device("cuda:0")
    interp.strategy.path = path
    qd = torch.as_tensor(q, device=dev)
    out = torch.empty((q.size, L), dtype=tdt, device=dev) if fill is None else \
        torch.full((q.size, L), fill, dtype=tdt, device=dev)
    interp.interp_array_into(qd, out)
    return out


# (n, L): scalar data, the reference's (100, 5), unaligned and aligned short rows, both sides of every vector /
# group-size boundary, rows just under one workgroup pass
SHAPES = [(100, 1), (100, 5), (1024, 8), (300, 9), (64, 30), (1024, 32), (129, 64), (77, 128), (40, 254), (33, 500)]


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n,L", SHAPES)
def test_short_rows_every_variant_bit_exact(pkg, dt, n, L):
    import torch
    tdt = torch.float64 if dt == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n * 1009 + L)
    if dt == np.float32 and L == 500:
        L = 1000   # LV = 250 for both types
    Q = 150_001     # ragged last batch, more than 5 (n - 1) queries, above the bucket-index threshold
    x = knots("rand", n, rng, dt)
    y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    q[:4] = [x[0], x[-1], x[n // 2], np.nextafter(x[-1], x[0])]
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref_c = oracle.interp1d_cubic(x, y, a, b, q)
    _, _, ref_l = oracle.interp1d_linear(x, y, q)
    cub = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
        .strategy(pkg.CubicSpline.new()).build()
    lin = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).build()
    taken = set()
    for name, path, kn in variants(pkg):
        with knobs(**kn):
            pkg.profile_enable(True); pkg.profile_read(reset=True)
            got = _device_eval(pkg, cub, q, L, path, tdt).cpu().numpy()
            taken.add(pkg.profile_read(reset=True)["last_path"]); pkg.profile_enable(False)
            check_equal(got, ref_c.reshape(Q, L), f"cubic {name} n={n} L={L}")
            got = _device_eval(pkg, lin, q, L, path, tdt).cpu().numpy()
            check_equal(got, ref_l.reshape(Q, L), f"linear {name} n={n} L={L}")
    if L > 2 and L % (16 // np.dtype(dt).itemsize) == 0:   # 16-byte vector rows: the grouped form exists
        assert "bucketed" in taken and "gather" in taken, taken   # both formulations really ran


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_short_rows_small_batches_and_pyramid_search(pkg, dt):
    """Below 4096 queries the fused kernel searches with the wave-cooperative pyramid (no bucket index); even and
    uneven axes, n <= 64 (one pyramid level) and n > 64."""
    import torch
    tdt = torch.float64 if dt == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for kind, n, L, Q in (("lin", 100, 5, 1000), ("rand", 50, 12, 3000), ("log", 2000, 8, 4000), ("jit", 5000, 20, 70)):
        x = knots(kind, n, rng, dt)
        y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
        q = rng.uniform(x[0], x[-1], Q).astype(dt)
        st, a, b = oracle.cubic_build(x, y)
        _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
        os.environ["NDI_SPLINE_BLOCKED"] = "0"   # this test is about the evaluation kernels: bit-identical tables
        try:
            it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
                .strategy(pkg.CubicSpline.new()).build()
        finally:
            del os.environ["NDI_SPLINE_BLOCKED"]
        for name, path, kn in variants(pkg):
            with knobs(**kn):
                got = _device_eval(pkg, it, q, L, path, tdt).cpu().numpy()
                check_equal(got, ref.reshape(Q, L), f"{kind} n={n} L={L} {name}")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_short_rows_first_error_and_extrapolation(pkg, dt):
    """Rows before the first failing query are written, later rows untouched (interp1d/mod.rs:334-342), in every
    variant; extrapolation with the end interval, periodic wrap, NaN under extrapolation = panic-equivalent."""
    import torch
    tdt = torch.float64 if dt == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    for n, L in ((100, 5), (300, 16), (64, 192)):
        Q = 60_000
        x = knots("rand", n, rng, dt)
        y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
        y[-1] = y[0]
        q = rng.uniform(x[0], x[-1], Q).astype(dt)
        st, a, b = oracle.cubic_build(x, y)
        _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
        ref = ref.reshape(Q, L)
        qb = q.copy()
        qb[41_003] = x[-1] + 1; qb[52_000] = np.nan; qb[59_999] = x[0] - 1
        span = x[-1] - x[0]
        qe = rng.uniform(x[0] - 2.5 * span, x[-1] + 2.5 * span, Q).astype(dt)
        _, _, ref_e = oracle.interp1d_cubic(x, y, a, b, qe, oracle.EXTRAPOLATE_YES)
        _, _, ref_le = oracle.interp1d_linear(x, y, qe, True)
        stp, ap, bp = oracle.cubic_build(x, y, periodic=True)
        _, _, ref_p = oracle.interp1d_cubic(x, y, ap, bp, qe, oracle.EXTRAPOLATE_PERIODIC)
        yd, xd = torch.as_tensor(y, device=dev), torch.as_tensor(x, device=dev)
        cub = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
        cube = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new().extrapolate(True)).build()
        line = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.Linear.new().extrapolate(True)).build()
        per = pkg.Interp1DBuilder.new(yd).x(xd).strategy(
            pkg.CubicSpline.new().extrapolate(True).boundary(pkg.BoundaryCondition.Periodic)).build()
        for name, path, kn in variants(pkg):
            with knobs(**kn):
                cub.strategy.path = path
                buf = torch.full((Q, L), -4.0, dtype=tdt, device=dev)
                with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
                    cub.interp_array_into(torch.as_tensor(qb, device=dev), buf)
                assert ei.value.index == 41_003, name
                h = buf.cpu().numpy()
                assert np.array_equal(h[:41_003], ref[:41_003]), name
                assert np.all(h[41_003:] == -4.0), name
                check_equal(_device_eval(pkg, cube, qe, L, path, tdt).cpu().numpy(), ref_e.reshape(Q, L), f"extrap {name}")
                check_equal(_device_eval(pkg, line, qe, L, path, tdt).cpu().numpy(), ref_le.reshape(Q, L), f"lin extrap {name}")
                check_equal(_device_eval(pkg, per, qe, L, path, tdt).cpu().numpy(), ref_p.reshape(Q, L), f"periodic {name}")
                qn = qe.copy(); qn[777] = np.nan
                cube.strategy.path = path
                buf = torch.full((Q, L), -4.0, dtype=tdt, device=dev)
                with pytest.raises(pkg.Panic):
                    cube.interp_array_into(torch.as_tensor(qn, device=dev), buf)
                h = buf.cpu().numpy()
                assert np.array_equal(h[:777], ref_e.reshape(Q, L)[:777]) and np.all(h[777:] == -4.0), name


def test_reference_bench_shapes_at_1e7_queries(pkg):
    """The reference's own data shapes at Q = 1e7 on device buffers: scalar data and (100, 5) f64, both strategies
    (benches/bench_interp1d.rs:12-47, 82-122 with 1e7 instead of 1e4 queries).  Every output of the shipped path
    equals the two-kernel flat form's bit for bit; 20 000 sampled rows equal the oracle; knot hits return the data."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(42)
    Q = 10_000_000
    for shape in ((100,), (100, 5)):
        n = shape[0]
        L = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        y = rng.uniform(0.0, 1.0, shape)
        x = np.arange(n, dtype=np.float64)               # the default index axis of the benches
        q = rng.uniform(0.0, n - 1.0, Q)
        q[:n] = x                                         # every knot once
        sample = rng.choice(Q, 20_000, replace=False)
        sample[:n] = np.arange(n)
        yd = torch.as_tensor(y, device=dev)
        qd = torch.as_tensor(q, device=dev)
        st, a, b = oracle.cubic_build(x, y.reshape(n, L))
        for strat_name, strat in (("linear", pkg.Linear.new()), ("cubic", pkg.CubicSpline.new())):
            it = pkg.Interp1DBuilder.new(yd).strategy(strat).build()
            if strat_name == "linear":
                _, _, ref = oracle.interp1d_linear(x, y.reshape(n, L), q[sample])
            else:
                _, _, ref = oracle.interp1d_cubic(x, y.reshape(n, L), a, b, q[sample])
            with knobs():
                out = it.interp_array(qd)
            with knobs(NDI_SHORT_MODE=1):
                flat = it.interp_array(qd)
            assert torch.equal(out, flat), (shape, strat_name)
            got = out.reshape(Q, L)[torch.as_tensor(sample, device=dev)].cpu().numpy()
            check_equal(got, ref.reshape(-1, L), f"{shape} {strat_name} sampled rows")
            assert np.array_equal(got[:n], y.reshape(n, L)), "knot hits return the data rows"
            del out, flat
        torch.cuda.empty_cache()


def test_short_rows_in_the_ring_and_strided_buffers(pkg):
    """The same kernels behind the ring evaluation (chunks, side-stream preparation) and with a padded row stride."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    n, L, Q = 200, 24, 90_000
    x = knots("jit", n, rng, np.float64)
    y = rng.uniform(-1.0, 1.0, (n, L))
    q = rng.uniform(x[0], x[-1], Q)
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
        .strategy(pkg.CubicSpline.new()).build()
    for name, path, kn in variants(pkg):
        with knobs(**kn):
            it.strategy.path = path
            got = np.zeros((Q, L))
            ring = pkg.striped_ring(20_000, L, 2, np.float64, 0)

            def consume(c, rows):
                got[c.q_begin:c.q_begin + c.q_count] = rows.cpu().numpy()

            it.interp_array_ring(torch.as_tensor(q, device=dev), 20_000, consume, slots=ring)
            check_equal(got, ref.reshape(Q, L), f"ring {name}")
            wide = torch.full((Q, L + 8), -9.0, dtype=torch.float64, device=dev)
            it.strategy.interp_array_into(it, torch.as_tensor(q, device=dev), wide[:, :L])   # row stride L + 8
            h = wide.cpu().numpy()
            check_equal(h[:, :L], ref.reshape(Q, L), f"strided {name}")
            assert np.all(h[:, L:] == -9.0), name


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_yk_form_after_every_build_kernel(pkg, dt, capfd):
    """The {y, k} LDS form re-forms a / b from the derivatives every build kernel leaves (serial general / periodic /
    n == 3 closed forms / per-lane Individual / blocked sweeps, general and periodic): its rows must carry the bits of
    the a / b tables the same build wrote, whatever the boundary condition (cubic_spline.rs:354-365)."""
    import torch
    from test_gpu_parity import BCS
    from test_gpu_spline_blocked import _bc
    tdt = torch.float64 if dt == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    S, R, B = pkg.SingleBoundary, pkg.RowBoundary, pkg.BoundaryCondition
    cases = []
    for name, (per, left, right) in BCS.items():
        for n, L in ((3, 4), (4, 8), (300, 12), (2100, 3)):      # closed forms, serial kernels, blocked sweeps
            if n == 3 and name not in ("nk", "per"):
                continue
            cases.append((name, n, L, per, (lambda L=L, left=left, right=right: _bc(pkg, L, left, right)) if not per
                          else (lambda: B.Periodic)))
    def individual(L=6):    # every lane its own pair of end conditions
        rows = np.empty((1, L), dtype=object)
        kinds = [S.NotAKnot, S.Natural, S.Clamped, S.FirstDeriv(0.3), S.SecondDeriv(-0.2), S.NotAKnot]
        for i in range(L):
            rows[0, i] = R.Mixed(kinds[i % 6], kinds[(i + 2) % 6])
        return B.Individual(rows)
    cases.append(("individual", 200, 6, False, individual))
    for name, n, L, per, bc in cases:
        x = knots("jit" if n > 4 else "lin", n, rng, dt) if n > 3 else np.array([-1.0, 0.0, 3.0], dtype=dt)
        y = rng.uniform(-1.0, 1.0, (n, L)).astype(dt)
        if per:
            y[-1] = y[0]
        it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
            .strategy(pkg.CubicSpline.new().boundary(bc())).build()
        q = rng.uniform(x[0], x[-1], 20_001).astype(dt)
        q[:2] = [x[0], x[-1]]
        with knobs(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_PACK=0):
            ref = _device_eval(pkg, it, q, L, pkg.PATH_GATHER, tdt).cpu().numpy()
        capfd.readouterr()
        os.environ["NDI_TRACE_PLAN"] = "1"
        try:
            with knobs(NDI_SHORT_MODE=2, NDI_FUSED_LDS=2):
                got = _device_eval(pkg, it, q, L, pkg.PATH_GATHER, tdt).cpu().numpy()
        finally:
            del os.environ["NDI_TRACE_PLAN"]
        err = capfd.readouterr().err
        if L > 2:   # (1-2 lanes take the one-thread-per-query kernel)
            assert "tables=lds{y,k}" in err, (name, n, L, err)
        assert np.array_equal(got, ref, equal_nan=True), f"{name} n={n} L={L}: {{y, k}} rows differ from the table form"


@pytest.mark.parametrize("seed", range(12))
def test_device_buffer_fuzz(pkg, seed):
    """Seeded fuzz over what AUTO / GATHER / BUCKETED pick for device-resident batches: random knot counts (incl. axes
    too long for LDS: the global-memory bucket index), lanes from scalar data to 500, batch sizes from 1 to 400 000
    (both sides of the one-thread-per-query / query-order and of the grouped thresholds), both strategies,
    extrapolation and the periodic wrap -- every row against the oracle, bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4242 + seed)
    for _ in range(10):
        dt = rng.choice([np.float64, np.float32])
        tdt = torch.float64 if dt == np.float64 else torch.float32
        cubic = bool(rng.integers(0, 2))
        n = int(rng.choice([3 if cubic else 2, 5, 64, 65, 100, 777, 1024, 3000, 25_000, 70_000]))
        L = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 24, 32, 63, 64, 100, 128, 200, 256, 500]))
        if n >= 25_000:
            L = min(L, 8)
        Q = int(rng.choice([1, 63, 64, 65, 1000, 4095, 4096, 50_000, 400_000]))
        if Q * L > 40_000_000:
            Q = 40_000_000 // L
        path = int(rng.choice([pkg.PATH_AUTO, pkg.PATH_GATHER, pkg.PATH_BUCKETED]))
        x = knots(rng.choice(["rand", "jit", "log", "lin"]), n, rng, dt) if n > 3 else np.arange(n).astype(dt)
        y = rng.uniform(-1, 1, (n, L)).astype(dt)
        ext = bool(rng.integers(0, 2))
        per = cubic and n >= 3 and bool(rng.integers(0, 3) == 0)
        span = float(x[-1] - x[0])
        m = 0.4 if ext else 0.0
        q = rng.uniform(x[0] - m * span, x[-1] + m * span, Q).astype(dt)
        if not ext:
            q = np.clip(q, x[0], x[-1])
        if cubic:
            if per:
                y[-1] = y[0]
            os.environ["NDI_SPLINE_BLOCKED"] = "0"      # bit-identical tables: this test is about the evaluation
            try:
                strat = pkg.CubicSpline.new().extrapolate(ext)
                if per:
                    strat = strat.boundary(pkg.BoundaryCondition.Periodic)
                it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(strat).build()
            finally:
                del os.environ["NDI_SPLINE_BLOCKED"]
            st, a, b = oracle.cubic_build(x, y, periodic=per)
            assert st == oracle.OK
            mode = (2 if per else 1) if ext else 0
            _, _, ref = oracle.interp1d_cubic(x, y, a, b, q, mode)
        else:
            it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
                .strategy(pkg.Linear.new().extrapolate(ext)).build()
            _, _, ref = oracle.interp1d_linear(x, y, q, ext)
        got = _device_eval(pkg, it, q, L, path, tdt, fill=-9.0).cpu().numpy()
        check_equal(got, ref.reshape(Q, L), f"fuzz seed={seed} dt={np.dtype(dt).name} cubic={cubic} per={per} ext={ext} n={n} L={L} Q={Q} path={path}")


def test_auto_picks_the_measured_form(pkg, capfd):
    """What AUTO takes for the shapes of profiles/r04_short_rows_*: the plan trace of the query-order kernel (or its
    absence: grouped / one thread per query) for a large device-resident batch on 1024 knots."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    x = knots("rand", 1024, rng, np.float64)
    expect = [   # (dtype, lanes, cubic, fragment of the trace or None = not the query-order kernel)
        (np.float64, 8, True, "tables=lds{y,k}"), (np.float32, 8, True, "tables=lds{y,a,b}"),
        (np.float32, 16, True, "tables=lds{y,k}"), (np.float64, 32, True, "fused sorted"),   # 256-byte rows from L2: interval order per round
        (np.float64, 128, True, None),                      # 1 KiB rows, many queries per interval: grouped
        (np.float64, 8, False, "tables=lds{y,a,b}"), (np.float32, 32, False, "tables=memory"),
        (np.float64, 128, False, "tables=memory"),          # Linear: never grouped
        (np.float64, 1, True, "[ndi plan] lanes L=1 qpl=2"),   # scalar data at 1e7 queries: query per lane, tables in LDS
    ]
    for dt, L, cubic, frag in expect:
        tdt = torch.float64 if dt == np.float64 else torch.float32
        Q = 10_000_000 if L == 1 else 600_000_000 // (L * np.dtype(dt).itemsize)
        xs = np.unique(x.astype(dt))
        yd = torch.rand((xs.size, L), dtype=tdt, device=dev)
        strat = pkg.CubicSpline.new() if cubic else pkg.Linear.new()
        it = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(xs, device=dev)).strategy(strat).build()
        q = (torch.rand(Q, dtype=tdt, device=dev) * float(xs[-1] - xs[0]) * 0.999 + float(xs[0])).clamp(float(xs[0]), float(xs[-1]))
        out = torch.empty((Q, L), dtype=tdt, device=dev)
        capfd.readouterr()
        os.environ["NDI_TRACE_PLAN"] = "1"
        try:
            with knobs():
                it.interp_array_into(q, out)
        finally:
            del os.environ["NDI_TRACE_PLAN"]
        err = capfd.readouterr().err
        if frag is None:
            assert "[ndi plan] fused" not in err, (np.dtype(dt).name, L, cubic, err)
        else:
            assert frag in err, (np.dtype(dt).name, L, cubic, err)
        del out, q, it, yd


@pytest.mark.parametrize("dt,n,L", [(np.float64, 12000, 8), (np.float64, 16384, 5), (np.float32, 30000, 16), (np.float32, 9000, 3)])
def test_query_order_kernel_on_long_axes(pkg, dt, n, L, capfd):
    """Axes of more than half the LDS (up to ~18 000 f64 / 36 000 f32 knots) still take the query-order kernel -- one
    large workgroup per CU around the staged knots instead of the two-kernel flat form -- with the oracle's bits."""
    import torch
    tdt = torch.float64 if dt == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + L)
    x = knots("rand", n, rng, dt)
    y = rng.uniform(-1, 1, (n, L)).astype(dt)
    Q = 120_001
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    q[:3] = [x[0], x[-1], x[n // 3]]
    os.environ["NDI_SPLINE_BLOCKED"] = "0"
    try:
        cub = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
    finally:
        del os.environ["NDI_SPLINE_BLOCKED"]
    lin = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).build()
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref_c = oracle.interp1d_cubic(x, y, a, b, q)
    _, _, ref_l = oracle.interp1d_linear(x, y, q)
    for it, ref, name in ((cub, ref_c, "cubic"), (lin, ref_l, "linear")):
        capfd.readouterr()
        os.environ["NDI_TRACE_PLAN"] = "1"
        try:
            with knobs():
                got = _device_eval(pkg, it, q, L, pkg.PATH_AUTO, tdt).cpu().numpy()
        finally:
            del os.environ["NDI_TRACE_PLAN"]
        err = capfd.readouterr().err
        assert "[ndi plan] fused tables=memory" in err, (name, n, L, err)
        check_equal(got, ref.reshape(Q, L), f"long axis {name} n={n} L={L}")


@pytest.mark.parametrize("dt,n,L,kind", [(np.float64, 30_000, 1, "rand"), (np.float64, 100_000, 5, "jit"), (np.float32, 70_000, 8, "log"),
                                         (np.float64, 50_000, 2, "lin"), (np.float32, 200_000, 3, "rand")])
def test_query_order_kernel_with_knots_in_global_memory(pkg, dt, n, L, kind, capfd):
    """Axes too long for LDS: the query-order kernel leaves the knots in global memory and searches them through the
    u32 bucket index (evenly spaced axes: the reference's O(1) guess) -- one launch instead of locate + flat, the
    oracle's bits, first-error cut included."""
    import torch
    tdt = torch.float64 if dt == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + L)
    x = knots(kind, n, rng, dt) if kind != "log" else np.unique(np.logspace(-2, 0, n).astype(dt))
    n = x.size
    y = rng.uniform(-1, 1, (n, L)).astype(dt)
    Q = 90_001
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    q[:4] = [x[0], x[-1], x[n // 3], np.nextafter(x[-1], x[0])]
    os.environ["NDI_SPLINE_BLOCKED"] = "0"
    try:
        cub = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
    finally:
        del os.environ["NDI_SPLINE_BLOCKED"]
    lin = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).build()
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref_c = oracle.interp1d_cubic(x, y, a, b, q)
    _, _, ref_l = oracle.interp1d_linear(x, y, q)
    for it, ref, name in ((cub, ref_c, "cubic"), (lin, ref_l, "linear")):
        capfd.readouterr()
        os.environ["NDI_TRACE_PLAN"] = "1"
        try:
            with knobs():
                got = _device_eval(pkg, it, q, L, pkg.PATH_AUTO, tdt).cpu().numpy()
        finally:
            del os.environ["NDI_TRACE_PLAN"]
        assert "knots=global" in capfd.readouterr().err, (name, n, L)
        check_equal(got, ref.reshape(Q, L), f"global knots {name} n={n} L={L}")
    # first error: rows before it written, later rows untouched
    q2 = q.copy(); q2[40_000] = x[-1] + 1
    out = torch.full((Q, L), -2.0, dtype=tdt, device=dev)
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        lin.interp_array_into(torch.as_tensor(q2, device=dev), out)
    assert ei.value.index == 40_000
    o = out.cpu().numpy()
    assert np.array_equal(o[:40_000], ref_l.reshape(Q, L)[:40_000]) and np.all(o[40_000:] == -2.0)
