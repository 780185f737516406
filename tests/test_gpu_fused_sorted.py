"""eval_fused_sorted_kernel (round 5): the query-order short-row kernel with the queries of a workgroup round ordered by
interval in LDS first, so that neighbouring items read the same operand rows (L1 instead of L2) -- rows of 64 bytes to a
few KiB whose tables do not fit LDS.  Forced with NDI_FUSED_SORTED=1 (conftest sets NDI_TUNE_LIVE), the plan line is
asserted; results against the CPU oracle bit for bit (Linear = linear.rs:73-98, CubicSpline = cubic_spline.rs:791-830):
both element types, vector and scalar rows (lane counts that are / are not whole 16-byte vectors), ragged batches that
end inside a workgroup round, strided output, extrapolation, the periodic wrap, the first-error cut, interp_array's own
range test."""
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
        os.environ["NDI_FUSED_SORTED"] = self.value
        os.environ["NDI_LANES_KERNEL"] = "0"
        os.environ["NDI_TRACE_PLAN"] = "1"
        self.capfd.readouterr()
        return self

    def __exit__(self, *a):
        for k in ("NDI_FUSED_SORTED", "NDI_LANES_KERNEL", "NDI_TRACE_PLAN"):
            os.environ.pop(k, None)
        self.plans = [ln for ln in self.capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]


def _tdt(dt):
    import torch
    return torch.float64 if dt == np.float64 else torch.float32


@pytest.mark.parametrize("dt,n,L,strat,kind", [
    (np.float64, 1024, 32, "cubic", "rand"), (np.float64, 300, 16, "cubic", "jit"), (np.float64, 4097, 8, "cubic", "rand"),
    (np.float32, 1024, 64, "cubic", "rand"), (np.float32, 777, 250, "cubic", "log"), (np.float64, 100, 9, "cubic", "lin"),
    (np.float64, 1024, 32, "linear", "rand"), (np.float32, 2000, 48, "linear", "rand"), (np.float64, 50, 128, "linear", "lin"),
])
def test_fused_sorted_bit_exact(pkg, capfd, dt, n, L, strat, kind):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n * 3 + L)
    Q = 23_456 if L >= 128 else 91_003          # ends inside a round of 4096 / 2048 queries
    x = knots(kind, n, rng, dt)
    y = rng.uniform(-1, 1, (n, L)).astype(dt)
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    q[:3] = [x[0], x[-1], x[n // 2]]
    q[3:3 + min(n, 500)] = x[:min(n, 500)]      # knots themselves
    if strat == "cubic":
        st, a, b = oracle.cubic_build(x, y)
        ref = oracle.interp1d_cubic(x, y, a, b, q)[2].reshape(Q, L)
        mk = lambda: pkg.CubicSpline.new().reference_order(True)   # (bit-identical tables also on 4097 knots)
    else:
        ref = oracle.interp1d_linear(x, y, q)[2].reshape(Q, L)
        mk = lambda: pkg.Linear.new()
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(mk()).build()
    qd = torch.as_tensor(q, device=dev)
    with forced(capfd) as f:
        out = torch.full((Q, L), -9.0, dtype=_tdt(dt), device=dev)
        it.interp_array_into(qd, out)
        wide = torch.full((Q, L + 4), -9.0, dtype=_tdt(dt), device=dev)
        it.strategy.interp_array_into(it, qd, wide[:, :L])
        fresh = it.interp_array(qd)
    sorted_plans = [p for p in f.plans if " fused sorted " in p]
    assert len(sorted_plans) == 3 and "prepass=0" in sorted_plans[2], f.plans
    check_equal(out.cpu().numpy(), ref, f"sorted {n}x{L}")
    check_equal(wide[:, :L].cpu().numpy(), ref, f"sorted strided {n}x{L}")
    assert bool((wide[:, L:] == -9.0).all())
    check_equal(fresh.cpu().numpy().reshape(Q, L), ref, f"sorted fresh {n}x{L}")
    # the first-error cut
    q2 = q.copy()
    q2[60_001 % Q] = x[-1] + 1
    q2[20_002 % Q] = x[0] - 1
    first = min(60_001 % Q, 20_002 % Q)
    buf = torch.full((Q, L), -3.0, dtype=_tdt(dt), device=dev)
    with forced(capfd) as f:
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.interp_array_into(torch.as_tensor(q2, device=dev), buf)
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as e2:
            it.interp_array(torch.as_tensor(q2, device=dev))
    assert ei.value.index == first and e2.value.index == first and any(" fused sorted " in p for p in f.plans), f.plans
    g = buf.cpu().numpy()
    assert np.array_equal(g[:first], ref[:first]) and np.all(g[first:] == -3.0)
    # extrapolation
    ex = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(mk().extrapolate(True)).build()
    span = x[-1] - x[0]
    q3 = rng.uniform(x[0] - span, x[-1] + span, Q).astype(dt)
    with forced(capfd) as f:
        got3 = ex.interp_array(torch.as_tensor(q3, device=dev)).cpu().numpy().reshape(Q, L)
    assert any(" fused sorted " in p for p in f.plans), f.plans
    if strat == "cubic":
        ref3 = oracle.interp1d_cubic(x, y, a, b, q3, oracle.EXTRAPOLATE_YES)[2].reshape(Q, L)
    else:
        ref3 = oracle.interp1d_linear(x, y, q3, True)[2].reshape(Q, L)
    check_equal(got3, ref3, "sorted extrapolate")


def test_fused_sorted_periodic(pkg, capfd):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(17)
    n, L, Q = 513, 24, 70_001
    x = knots("rand", n, rng, np.float64)
    y = rng.uniform(-1, 1, (n, L))
    y[-1] = y[0]
    st, a, b = oracle.cubic_build(x, y, periodic=True)
    assert st == oracle.OK
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
        .strategy(pkg.CubicSpline.new().boundary(pkg.BoundaryCondition.Periodic).extrapolate(True)).build()
    span = x[-1] - x[0]
    q = rng.uniform(x[0] - 3 * span, x[-1] + 3 * span, Q)
    ref = oracle.interp1d_cubic(x, y, a, b, q, oracle.EXTRAPOLATE_PERIODIC)[2].reshape(Q, L)
    with forced(capfd) as f:
        got = it.interp_array(torch.as_tensor(q, device=dev)).cpu().numpy().reshape(Q, L)
    assert any(" fused sorted " in p for p in f.plans), f.plans
    check_equal(got, ref, "sorted periodic")
