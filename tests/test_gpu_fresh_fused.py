"""`interp_array` semantics (NDI_EVAL_FRESH_OUTPUT: the output is the call's own and dropped on Err) through the
query-order kernels eval_fused_kernel (1-D, rows of 8 .. 128 lanes) and eval_fused2d_kernel (2-D): no range pre-pass
runs -- the kernel tests every query itself and records the lowest failing index (interp1d.rs:232-252,
interp2d/mod.rs:175-196 allocate the output and return Err on the first failing query).  Asserted: the plan line says
prepass=0 for interp_array and prepass=1 for interp_array_into on the same handle, both give the oracle's bits, the
error carries the reference's index (and axis: x before y), NaN queries under extrapolation fail as in the reference."""
import os

import numpy as np
import pytest

import oracle
from test_gpu_parity import check_equal, knots

pytestmark = pytest.mark.gpu


class traced:
    def __init__(self, capfd, **env):
        self.capfd, self.env = capfd, env

    def __enter__(self):
        os.environ["NDI_TRACE_PLAN"] = "1"
        for k, v in self.env.items():
            os.environ[k] = v
        self.capfd.readouterr()
        return self

    def __exit__(self, *a):
        os.environ.pop("NDI_TRACE_PLAN", None)
        for k in self.env:
            os.environ.pop(k, None)
        self.plans = [ln for ln in self.capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]


def _tdt(dt):
    import torch
    return torch.float64 if dt == np.float64 else torch.float32


@pytest.mark.parametrize("dt,n,L,strat", [(np.float64, 1024, 8, "cubic"), (np.float32, 1024, 8, "cubic"), (np.float64, 300, 32, "cubic"),
                                           (np.float32, 2000, 64, "linear"), (np.float64, 1024, 16, "linear")])
def test_fused_1d_interp_array_skips_the_prepass(pkg, capfd, dt, n, L, strat):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + L)
    Q = 300_007
    x = knots("rand", n, rng, dt)
    y = rng.uniform(-1, 1, (n, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    q[:3] = [x[0], x[-1], x[n // 2]]
    if strat == "cubic":
        st, a, b = oracle.cubic_build(x, y)
        ref = oracle.interp1d_cubic(x, y, a, b, q)[2].reshape(Q, L)
        s = pkg.CubicSpline.new()
    else:
        ref = oracle.interp1d_linear(x, y, q)[2].reshape(Q, L)
        s = pkg.Linear.new()
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(s).build()
    qd = torch.as_tensor(q, device=dev)
    with traced(capfd, NDI_LANES_KERNEL="0") as t:
        got = it.interp_array(qd)
        out = torch.full((Q, L), -2.0, dtype=_tdt(dt), device=dev)
        it.interp_array_into(qd, out)
    fused = [p for p in t.plans if " fused " in p]
    assert len(fused) == 2 and "prepass=0" in fused[0] and "prepass=1" in fused[1], t.plans
    check_equal(got.cpu().numpy().reshape(Q, L), ref, f"fresh fused {n}x{L}")
    check_equal(out.cpu().numpy(), ref, f"into fused {n}x{L}")
    # the first failing query: same index through both semantics
    q2 = q.copy()
    q2[200_000] = x[-1] + 1
    q2[123_457] = x[0] - 1
    with traced(capfd, NDI_LANES_KERNEL="0") as t:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as e1:
            it.interp_array(torch.as_tensor(q2, device=dev))
        buf = torch.full((Q, L), -2.0, dtype=_tdt(dt), device=dev)
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as e2:
            it.interp_array_into(torch.as_tensor(q2, device=dev), buf)
    assert e1.value.index == 123_457 and e2.value.index == 123_457
    assert any("prepass=0" in p for p in t.plans), t.plans
    g = buf.cpu().numpy()
    assert np.array_equal(g[:123_457], ref[:123_457]) and np.all(g[123_457:] == -2.0)
    # extrapolation: only NaN fails
    ex = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
        .strategy(s.extrapolate(True) if strat == "linear" else pkg.CubicSpline.new().extrapolate(True)).build()
    span = x[-1] - x[0]
    q3 = rng.uniform(x[0] - span, x[-1] + span, Q).astype(dt)
    with traced(capfd, NDI_LANES_KERNEL="0") as t:
        got3 = ex.interp_array(torch.as_tensor(q3, device=dev)).cpu().numpy().reshape(Q, L)
    assert any(" fused " in p and "prepass=0" in p for p in t.plans), t.plans
    if strat == "cubic":
        ref3 = oracle.interp1d_cubic(x, y, a, b, q3, oracle.EXTRAPOLATE_YES)[2].reshape(Q, L)
    else:
        ref3 = oracle.interp1d_linear(x, y, q3, True)[2].reshape(Q, L)
    check_equal(got3, ref3, "fresh fused extrapolate")


@pytest.mark.parametrize("dt,C", [(np.float64, 5), (np.float32, 16), (np.float64, 8)])
def test_fused_2d_interp_array_skips_the_prepass(pkg, capfd, dt, C):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(C)
    nx, ny, Q = 300, 257, 200_003
    x = knots("rand", nx, rng, dt); y = knots("jit", ny, rng, dt)
    g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
    qx = rng.uniform(x[0], x[-1], Q).astype(dt); qy = rng.uniform(y[0], y[-1], Q).astype(dt)
    ref = oracle.interp2d_bilinear(x, y, g, qx, qy)[3].reshape(Q, C)
    it = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    env = dict(NDI_SLOPES2D_KERNEL="0", NDI_LANES2D_KERNEL="0")
    with traced(capfd, **env) as t:
        got = it.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev))
        out = torch.full((Q, C), -2.0, dtype=_tdt(dt), device=dev)
        it.interp_array_into(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), out)
    fused = [p for p in t.plans if " fused2d " in p]
    assert len(fused) == 2 and "prepass=0" in fused[0] and "prepass=1" in fused[1], t.plans
    check_equal(got.cpu().numpy().reshape(Q, C), ref, "fresh fused2d")
    check_equal(out.cpu().numpy(), ref, "into fused2d")
    qx2 = qx.copy(); qy2 = qy.copy()
    qy2[90_000] = y[-1] + 1; qx2[90_000] = x[0] - 1; qy2[50_001] = y[0] - 1
    with traced(capfd, **env):
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as e1:
            it.interp_array(torch.as_tensor(qx2, device=dev), torch.as_tensor(qy2, device=dev))
    assert (e1.value.index, e1.value.axis) == (50_001, 1)
    qy2[50_001] = y[0]
    with traced(capfd, **env):
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as e1:
            it.interp_array(torch.as_tensor(qx2, device=dev), torch.as_tensor(qy2, device=dev))
    assert (e1.value.index, e1.value.axis) == (90_000, 0)       # x before y for the same query (bilinear.rs:71-80)
