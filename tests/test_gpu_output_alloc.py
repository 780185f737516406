"""GPU tests of the round-6 boundary additions (include/ndinterp.h v0.5):
  * ndi_output_alloc / ndi_output_free -- the library-owned `Array::zeros` of Interp1D::interp_array (interp1d/mod.rs:204-209):
    zero-filled, placement-checked, usable as an evaluation target and as ring slots, refused pointers;
  * ndi_eval_opts validation -- unknown flag bits / non-zero `reserved` are NDI_BAD_ARG;
  * NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED -- interp_array_into's opt-in: rows before the first failing query are the
    reference's (interp1d/mod.rs:334-342), the error report is unchanged, no range pre-pass runs."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from test_gpu_parity import check_equal, knots

pytestmark = pytest.mark.gpu


def test_output_alloc_zeroed_and_freed(pkg):
    import torch
    cap = pkg._capi
    lib = cap.lib()
    for nbytes in (4096 + 8, 3 << 20, (1 << 30) + 4096):
        p = ctypes.c_void_p()
        info = cap.OutputInfo()
        assert lib.ndi_output_alloc(0, nbytes, 0, cap.OUTPUT_ZEROED, ctypes.byref(p), ctypes.byref(info)) == cap.OK, cap.last_error()
        assert p.value and info.tries >= 1 and info.fill_tbps > 0 and info.worst_fill_tbps <= info.fill_tbps
        if nbytes < (1 << 30):
            assert info.tries == 1
        assert lib.ndi_output_free(p) == cap.OK
        assert lib.ndi_output_free(p) == cap.BAD_ARG          # already gone
    # a freed buffer of >= 1 GiB is kept and handed out again for the same size (tries == 0), zeroed; trim releases it
    big = (1 << 30) + 4096
    p1 = ctypes.c_void_p(); info = cap.OutputInfo()
    assert lib.ndi_output_trim() == cap.OK
    assert lib.ndi_output_alloc(0, big, 1, cap.OUTPUT_ZEROED, ctypes.byref(p1), ctypes.byref(info)) == cap.OK and info.tries == 1
    import torch
    view = torch.as_tensor(type("V", (), {"__cuda_array_interface__": {"shape": (1024,), "typestr": "<f8", "data": (p1.value, False),
                                                                       "version": 2, "strides": None}})(), device="cuda:0")
    view.fill_(7.0)
    torch.cuda.synchronize()
    del view
    assert lib.ndi_output_free(p1) == cap.OK
    p2 = ctypes.c_void_p()
    assert lib.ndi_output_alloc(0, big, 1, cap.OUTPUT_ZEROED, ctypes.byref(p2), ctypes.byref(info)) == cap.OK
    assert info.tries == 0 and p2.value == p1.value
    view = torch.as_tensor(type("V", (), {"__cuda_array_interface__": {"shape": (1024,), "typestr": "<f8", "data": (p2.value, False),
                                                                       "version": 2, "strides": None}})(), device="cuda:0")
    assert bool((view == 0).all())
    del view
    # NDI_OUTPUT_UNINITIALIZED: the kept buffer comes back as it was left (no refill), with the fill rate it was kept with
    view = torch.as_tensor(type("V", (), {"__cuda_array_interface__": {"shape": (1024,), "typestr": "<f8", "data": (p2.value, False),
                                                                       "version": 2, "strides": None}})(), device="cuda:0")
    view.fill_(5.0)
    torch.cuda.synchronize()
    del view
    kept_rate = info.fill_tbps
    assert lib.ndi_output_free(p2) == cap.OK
    p2b = ctypes.c_void_p()
    assert lib.ndi_output_alloc(0, big, 1, cap.OUTPUT_UNINITIALIZED, ctypes.byref(p2b), ctypes.byref(info)) == cap.OK
    assert info.tries == 0 and p2b.value == p1.value and info.fill_tbps > 0 and info.alloc_ms < 5.0, (info.tries, info.alloc_ms, kept_rate)
    view = torch.as_tensor(type("V", (), {"__cuda_array_interface__": {"shape": (1024,), "typestr": "<f8", "data": (p2b.value, False),
                                                                       "version": 2, "strides": None}})(), device="cuda:0")
    assert bool((view == 5.0).all())
    del view
    assert lib.ndi_output_alloc(0, big, 1, 2, ctypes.byref(p2), ctypes.byref(info)) == cap.BAD_ARG      # unknown flag bit
    assert lib.ndi_output_free(p2b) == cap.OK and lib.ndi_output_trim() == cap.OK
    p3 = ctypes.c_void_p()
    assert lib.ndi_output_alloc(0, big, 1, cap.OUTPUT_ZEROED, ctypes.byref(p3), ctypes.byref(info)) == cap.OK and info.tries == 1
    assert lib.ndi_output_free(p3) == cap.OK and lib.ndi_output_trim() == cap.OK
    assert lib.ndi_output_free(ctypes.c_void_p(0x1000)) == cap.BAD_ARG
    assert lib.ndi_output_free(None) == cap.OK
    p = ctypes.c_void_p()
    assert lib.ndi_output_alloc(0, 0, 0, 0, ctypes.byref(p), None) == cap.BAD_ARG
    t = pkg.output_zeros((1000, 37), np.float32, 0)
    assert t.shape == (1000, 37) and t.dtype == torch.float32 and t.is_cuda and bool((t == 0).all())
    assert t.ndi_output_info["tries"] == 1
    te = pkg.output_empty((1000, 37), np.float32, 0)           # (contents unspecified)
    assert te.shape == (1000, 37) and te.ndi_output_info["tries"] == 1
    del te
    t2 = pkg.output_zeros((3, 5, 7), np.float64, 0)
    t2[1, 2, 3] = 4.0
    assert float(t2.sum()) == 4.0
    del t, t2


def test_interp_array_large_output_is_library_owned(pkg):
    """interp_array on a device query whose output is >= 1 GiB: the rows land in a buffer from ndi_output_alloc and equal
    the oracle's; the same rows through a ring whose slots are such buffers."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    n, L, Q = 300, 2048, 70_000                       # 70 000 x 2048 f64 = 1.15 GB
    x = knots("rand", n, rng, np.float64)
    y = rng.uniform(-1, 1, (n, L))
    q = rng.uniform(x[0], x[-1], Q)
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
    got = it.interp_array(torch.as_tensor(q, device=dev))
    assert hasattr(got, "ndi_output_info") and got.ndi_output_info["tries"] >= 1, "output did not come from ndi_output_alloc"
    st, a, b = oracle.cubic_build(x, y)
    sel = np.sort(rng.choice(Q, 3000, replace=False))
    ref = oracle.interp1d_cubic(x, y, a, b, q[sel])[2]
    check_equal(got[torch.as_tensor(sel, device=dev)].cpu().numpy(), ref, "interp_array into a library-owned output")
    del got
    # ring slots from the same allocator
    chunk = 9000
    slots = [pkg.output_empty((chunk, L), np.float64, 0) for _ in range(2)]
    rows = []

    def consumer(c, r):
        idx = np.arange(0, c.q_count, 97)
        rows.append((c.q_begin + idx, r[torch.as_tensor(idx, device=dev)].cpu().numpy()))
        return None
    it.interp_array_ring(torch.as_tensor(q, device=dev), chunk, consumer, slots=slots)
    gi = np.concatenate([r[0] for r in rows]); gv = np.concatenate([r[1] for r in rows])
    check_equal(gv, oracle.interp1d_cubic(x, y, a, b, q[gi])[2], "ring over library-owned slots")


def test_eval_opts_are_validated(pkg):
    import torch
    cap = pkg._capi
    dev = torch.device("cuda:0")
    it = pkg.Interp1DBuilder.new(np.linspace(0, 1, 50)).build()
    q = torch.rand(1000, dtype=torch.float64, device=dev) * 49
    out = torch.empty(1000, dtype=torch.float64, device=dev)
    info = cap.OobInfo()
    fn, h = cap.lib().ndi_interp1d_eval, it.strategy._h

    def call(flags, reserved):
        o = cap.EvalOpts()
        o.q_memspace = cap.MEM_DEVICE; o.out_memspace = cap.MEM_DEVICE
        o.flags = flags; o.reserved = reserved
        return fn(h, q.data_ptr(), 1000, out.data_ptr(), 1, ctypes.byref(o), ctypes.byref(info))
    assert call(0, 0) == cap.OK and call(1, 0) == cap.OK and call(2, 0) == cap.OK and call(3, 0) == cap.OK
    assert call(4, 0) == cap.BAD_ARG and "unknown bits" in cap.last_error()
    assert call(1 << 30, 0) == cap.BAD_ARG
    assert call(0, 7) == cap.BAD_ARG and "reserved" in cap.last_error()
    g = pkg.Interp2DBuilder.new(np.random.default_rng(0).uniform(0, 1, (9, 8))).build()
    o = cap.EvalOpts(); o.q_memspace = cap.MEM_DEVICE; o.out_memspace = cap.MEM_DEVICE; o.flags = 8
    qx = torch.rand(10, dtype=torch.float64, device=dev)
    assert cap.lib().ndi_interp2d_eval(g.strategy._h, qx.data_ptr(), qx.data_ptr(), 10, out.data_ptr(), 1, ctypes.byref(o),
                                       ctypes.byref(info)) == cap.BAD_ARG


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_rows_after_error_unspecified_opt_in(pkg, capfd, dt):
    """Scalar data, 100 knots, a batch the query-per-lane kernel takes: with the opt-in no range pre-pass runs (plan line),
    rows before the first failing query are the oracle's, the failure report is the reference's; without it later rows stay
    untouched."""
    import torch
    tdt = torch.float64 if dt == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    x = knots("rand", 100, rng, dt)
    y = rng.uniform(-1, 1, 100).astype(dt)
    Q = 4_000_000
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(x, y.reshape(-1, 1))
    ref = oracle.interp1d_cubic(x, y.reshape(-1, 1), a, b, q)[2].reshape(Q)
    qd = torch.as_tensor(q, device=dev)
    os.environ["NDI_TRACE_PLAN"] = "1"
    try:
        capfd.readouterr()
        out = torch.full((Q,), -7.0, dtype=tdt, device=dev)
        it.strategy.interp_array_into(it, qd, out.view(Q, 1), rows_after_error_unspecified=True)
        plans = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]
        assert plans and "prepass=0" in plans[0], plans
        check_equal(out.cpu().numpy(), ref, "opt-in, no failure")
        q2 = q.copy(); q2[1_234_567] = x[-1] + 1; q2[3_000_000] = x[0] - 1
        out2 = torch.full((Q,), -7.0, dtype=tdt, device=dev)
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.strategy.interp_array_into(it, torch.as_tensor(q2, device=dev), out2.view(Q, 1), rows_after_error_unspecified=True)
        assert ei.value.index == 1_234_567
        assert np.array_equal(out2.cpu().numpy()[:1_234_567], ref[:1_234_567])
        out3 = torch.full((Q,), -7.0, dtype=tdt, device=dev)
        capfd.readouterr()
        with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
            it.strategy.interp_array_into(it, torch.as_tensor(q2, device=dev), out3.view(Q, 1))
        plans = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("[ndi plan]")]
        assert ei.value.index == 1_234_567 and plans and "prepass=1" in plans[0], plans
        got3 = out3.cpu().numpy()
        assert np.array_equal(got3[:1_234_567], ref[:1_234_567]) and np.all(got3[1_234_567:] == -7.0)
    finally:
        os.environ.pop("NDI_TRACE_PLAN", None)
