"""GPU tests of the round-2 surface: the device-output ring (ndi_interp{1,2}d_eval_ring) incl. the C4 per-GPU
share at full size, the resident locator (ndi_locator_*), scratch-set reclaim, element-type checks, and several
devices driven from one process.  All through the C ABI (via the host mirror), checked against the CPU oracle."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import oracle
from test_gpu_parity import knots

pytestmark = pytest.mark.gpu


def _cubic(pkg, n, L, rng, dt=np.float64, kind="rand", device_data=False, **kw):
    import torch
    x = knots(kind, n, rng, dt)
    y = rng.uniform(0.0, 1.0, (n, L)).astype(dt)
    strat = pkg.CubicSpline.new().extrapolate(kw.get("extrapolate", False))
    yy = torch.as_tensor(y, device="cuda:0") if device_data else y
    xx = torch.as_tensor(x, device="cuda:0") if device_data else x
    interp = pkg.Interp1DBuilder.new(yy).x(xx).strategy(strat).build()
    st, a, b = oracle.cubic_build(x, y)
    assert st == oracle.OK
    return interp, x, y, a, b


# ------------------------------------------------------------------------------------------------
# ring evaluation
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("own_ring", [False, True])
def test_ring_chunks_equal_one_batch(pkg, dt, own_ring):
    """Every chunk handed to the consumer equals the matching rows of a single interp_array call (and the
    oracle), for caller-owned and library-owned rings, ragged last chunk included."""
    import torch
    rng = np.random.default_rng(11)
    interp, x, y, a, b = _cubic(pkg, 300, 1024, rng, dt)
    Q, chunk = 10_037, 2_048
    q = rng.uniform(x[0], x[-1], Q).astype(dt)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    qd = torch.as_tensor(q, device="cuda:0")
    tdt = torch.float64 if dt == np.float64 else torch.float32
    got = np.zeros_like(ref)
    seen = []

    if own_ring:
        def consumer(c, rows):
            assert rows is None
            # wrap the raw device pointer: copy the chunk out on the chunk's stream (stream-ordered before reuse)
            # the library-owned ring is one allocation with the slots interleaved row by row
            assert c.row_stride == 3 * 1024
            host = np.empty((c.q_count, 1024), dtype=dt)
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                             C.c_int, C.c_void_p]
            isz = host.itemsize
            assert hip.hipMemcpy2DAsync(host.ctypes.data, 1024 * isz, c.out, c.row_stride * isz, 1024 * isz,
                                        c.q_count, 2, c.stream) == 0
            assert hip.hipStreamSynchronize(C.c_void_p(c.stream)) == 0
            got[c.q_begin:c.q_begin + c.q_count] = host
            seen.append((c.index, c.q_begin, c.q_count, c.slot))
        interp.interp_array_ring(qd, chunk, consumer, n_slots=3)
        assert [s[3] for s in seen] == [k % 3 for k in range(len(seen))]
    else:
        ring = [torch.full((chunk, 1024), -1.0, dtype=tdt, device="cuda:0") for _ in range(2)]

        def consumer(c, rows):
            assert tuple(rows.shape) == (c.q_count, 1024) and rows.data_ptr() == c.out
            got[c.q_begin:c.q_begin + c.q_count] = rows.cpu().numpy()
            seen.append((c.index, c.q_begin, c.q_count, c.slot))
        interp.interp_array_ring(qd, chunk, consumer, slots=ring)
    assert [s[0] for s in seen] == list(range(5)) and [s[1] for s in seen] == [0, 2048, 4096, 6144, 8192]
    assert seen[-1][2] == Q - 4 * chunk
    assert np.array_equal(got, ref)
    # the recommended caller-owned layout: striped_ring (rows of a slot n_slots * lanes apart)
    sring = pkg.striped_ring(chunk, 1024, 3, dt, 0)
    assert sring[1].data_ptr() - sring[0].data_ptr() == 1024 * np.dtype(dt).itemsize and sring[0].stride(0) == 3 * 1024
    got3 = np.zeros_like(ref)

    def consumer3(c, rows):
        assert c.row_stride == 3 * 1024 and rows.data_ptr() == c.out and tuple(rows.shape) == (c.q_count, 1024)
        got3[c.q_begin:c.q_begin + c.q_count] = rows.cpu().numpy()
    interp.interp_array_ring(qd, chunk, consumer3, slots=sring)
    assert np.array_equal(got3, ref)
    # host queries are accepted as well (uploaded once), and both formulations agree
    interp.strategy.path = pkg.PATH_GATHER
    got2 = np.zeros_like(ref)
    ring = [torch.empty((chunk, 1024), dtype=tdt, device="cuda:0")]   # stream-ordered consumer: one slot is enough

    def consumer2(c, rows):
        got2[c.q_begin:c.q_begin + c.q_count] = rows.cpu().numpy()
    interp.interp_array_ring(q, chunk, consumer2, slots=ring)
    assert np.array_equal(got2, ref)


def test_ring_first_error_produces_exactly_the_rows_before_it(pkg):
    """Reference semantics (interp1d/mod.rs:334-342): rows before the first failing query exist, nothing after
    it is produced; the error names the lowest failing flat index and value."""
    import torch
    rng = np.random.default_rng(12)
    interp, x, y, a, b = _cubic(pkg, 100, 512, rng)
    Q, chunk = 9000, 1000
    q = rng.uniform(x[0], x[-1], Q)
    q[6500] = 7.25; q[3333] = -1.5; q[8999] = np.nan
    ring = [torch.empty((chunk, 512), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    produced = []

    def consumer(c, rows):
        produced.append((c.q_begin, c.q_count, rows.cpu().numpy()))
    with pytest.raises(pkg.InterpolateError.OutOfBounds, match=r"^x = -1\.5 is not in range") as ei:
        interp.interp_array_ring(torch.as_tensor(q, device="cuda:0"), chunk, consumer, slots=ring)
    assert ei.value.index == 3333
    assert [(p[0], p[1]) for p in produced] == [(0, 1000), (1000, 1000), (2000, 1000), (3000, 333)]
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q[:3333])
    assert np.array_equal(np.concatenate([p[2] for p in produced]), ref)
    # failure in the very first query: no chunk at all
    produced.clear()
    q[0] = 9.0
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        interp.interp_array_ring(q, chunk, consumer, slots=ring)
    assert ei.value.index == 0 and produced == []
    # extrapolating strategy: the only failure is a NaN query -> the reference's panic, at its index
    ex, x, y, a, b = _cubic(pkg, 100, 512, np.random.default_rng(12), extrapolate=True)
    produced.clear()
    q = rng.uniform(x[0] - 1, x[-1] + 1, Q); q[4100] = np.nan
    with pytest.raises(pkg.Panic, match="failed to convert NaN to usize") as ei:
        ex.interp_array_ring(q, chunk, consumer, slots=ring)
    assert ei.value.index == 4100 and sum(p[1] for p in produced) == 4100
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q[:4100], extrapolate=oracle.EXTRAPOLATE_YES)
    assert np.array_equal(np.concatenate([p[2] for p in produced]), ref)


def test_ring_consumer_exception_is_reraised(pkg):
    """An exception in a Python consumer does not unwind through the C frames and is not swallowed: the remaining
    chunks are produced without being consumed and the first exception is re-raised after the call."""
    rng = np.random.default_rng(15)
    interp, x, y, a, b = _cubic(pkg, 30, 256, rng)
    q = rng.uniform(x[0], x[-1], 5000)
    seen = []

    def consumer(c, rows):
        seen.append(c.index)
        if c.index == 1:
            raise ValueError("consumer failed on chunk 1")
    with pytest.raises(ValueError, match="consumer failed on chunk 1"):
        interp.interp_array_ring(q, 1000, consumer, n_slots=2)
    assert seen == [0, 1]
    assert np.array_equal(interp.interp_array(q), oracle.interp1d_cubic(x, y, a, b, q)[2])   # the handle is fine


def test_ring_consumer_on_its_own_stream(pkg):
    """A consumer that drains the slot on another stream returns an event; the library's stream waits for it
    before the slot is overwritten (2 slots, 9 chunks: every slot is reused four times)."""
    import torch
    rng = np.random.default_rng(13)
    interp, x, y, a, b = _cubic(pkg, 64, 2048, rng)
    Q, chunk = 9 * 4096, 4096
    q = rng.uniform(x[0], x[-1], Q)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    ring = [torch.empty((chunk, 2048), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    side = torch.cuda.Stream()
    total = torch.zeros((Q, 2048), dtype=torch.float64, device="cuda:0")

    def consumer(c, rows):
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())           # the chunk's kernels are enqueued on the current stream
        with torch.cuda.stream(side):
            side.wait_event(ready)
            for _ in range(20):                               # make the drain slower than the producer
                total[c.q_begin:c.q_begin + c.q_count].copy_(rows, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        return done
    interp.interp_array_ring(torch.as_tensor(q, device="cuda:0"), chunk, consumer, slots=ring)
    torch.cuda.synchronize()
    assert np.array_equal(total.cpu().numpy(), ref)


def test_ring_2d_bilinear(pkg):
    import torch
    rng = np.random.default_rng(14)
    nx, ny, Cn, Q, chunk = 70, 50, 16, 50_001, 8192
    g = rng.random((nx, ny, Cn), dtype=np.float32)
    x = knots("rand", nx, rng, np.float32); y = knots("jit", ny, rng, np.float32)
    interp = pkg.Interp2DBuilder.new(g).x(x).y(y).build()
    qx = rng.uniform(x[0], x[-1], Q).astype(np.float32); qy = rng.uniform(y[0], y[-1], Q).astype(np.float32)
    _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx, qy)
    got = np.zeros_like(ref)
    ring = [torch.empty((chunk, Cn), dtype=torch.float32, device="cuda:0") for _ in range(2)]

    def consumer(c, rows):
        got[c.q_begin:c.q_begin + c.q_count] = rows.cpu().numpy()
    interp.interp_array_ring(torch.as_tensor(qx, device="cuda:0"), torch.as_tensor(qy, device="cuda:0"), chunk,
                             consumer, slots=ring)
    assert np.array_equal(got, ref)
    # y failure before an x failure of a later query: y wins with its own index (bilinear.rs:71-80 per query)
    qy[30_000] = 99.0; qx[40_000] = -5.0
    count = []
    with pytest.raises(pkg.InterpolateError.OutOfBounds, match=r"^y = 99(\.0)? is not in range") as ei:
        interp.interp_array_ring(qx, qy, chunk, lambda c, rows: count.append(c.q_count), slots=ring)
    assert ei.value.index == 30_000 and sum(count) == 30_000


def test_ring_2d_tile_grouped_order_through_the_pipeline(pkg):
    """The tile-grouped 2-D order (tile histogram in locate2_kernel, LDS-tile evaluation) chunk by chunk through the
    two-stream ring pipeline: search + grouping of chunk k+1 on the side stream while chunk k is evaluated; both
    scratch sets are in use; bit-equal to the oracle; first-error cut."""
    import torch
    rng = np.random.default_rng(16)
    nx, ny, Cn, Q, chunk = 90, 75, 32, 70_001, 9000
    g = rng.random((nx, ny, Cn), dtype=np.float32)
    x = knots("rand", nx, rng, np.float32); y = knots("rand", ny, rng, np.float32)
    interp = pkg.Interp2DBuilder.new(g).x(x).y(y).build()
    interp.strategy.path = pkg.PATH_BUCKETED
    qx = rng.uniform(x[0], x[-1], Q).astype(np.float32); qy = rng.uniform(y[0], y[-1], Q).astype(np.float32)
    _, _, _, ref = oracle.interp2d_bilinear(x, y, g, qx, qy)
    got = np.zeros_like(ref)
    ring = pkg.striped_ring(chunk, Cn, 3, np.float32, 0)

    def consumer(c, rows):
        got[c.q_begin:c.q_begin + c.q_count] = rows.cpu().numpy()
    for _ in range(2):                                   # second pass: scratch sets and events are reused
        got[:] = 0
        interp.interp_array_ring(torch.as_tensor(qx, device="cuda:0"), torch.as_tensor(qy, device="cuda:0"), chunk,
                                 consumer, slots=ring)
        assert pkg.profile_read(reset=False)["last_path"] == "bucketed"
        assert np.array_equal(got, ref)
    qx[50_000] = x[-1] + 1
    count = []
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        interp.interp_array_ring(qx, qy, chunk, lambda c, rows: count.append(c.q_count), slots=ring)
    assert ei.value.index == 50_000 and sum(count) == 50_000


def test_c4_share_chunked(pkg):
    """configs[3], one GPU's share: 1D CubicSpline 4096 knots x 4096 lanes f64, 1.25e7 queries (409.6 GB of
    output) through a 2-slot device-output ring in chunks of 1e6.  Per chunk: the integer checksum of all
    4.096e9 outputs is equal between the gather and the bucketed formulation; sampled rows equal the oracle
    bit for bit; knot queries return the data row."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(44)
    n = L = 4096; Q = 12_500_000; chunk = 1_000_000
    x = knots("rand", n, rng, np.float64)
    y = rng.uniform(0.0, 1.0, (n, L))
    interp = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
        .strategy(pkg.CubicSpline.new()).build()
    q = np.concatenate([np.random.default_rng([44, c]).uniform(x[0], x[-1], min(chunk, Q - c * chunk))
                        for c in range(13)])
    hit = rng.integers(0, n - 1, 13 * 16).reshape(13, 16)
    for c in range(13):
        q[c * chunk:c * chunk + 16] = x[hit[c]]                 # 16 knot queries at the head of every chunk
    qd = torch.as_tensor(q, device=dev)
    yd = torch.as_tensor(y, device=dev)
    ring = [torch.empty((chunk, L), dtype=torch.float64, device=dev) for _ in range(2)]
    st, a, b = oracle.cubic_build(x, y)
    assert st == oracle.OK
    sums = {}
    for name, path in (("bucketed", pkg.PATH_BUCKETED), ("gather", pkg.PATH_GATHER)):
        interp.strategy.path = path
        sums[name] = []
        picks = []

        def consumer(c, rows, name=name, picks=picks):
            sums[name].append(rows.view(torch.int64).sum())         # stream-ordered device reduction
            if name == "bucketed":
                sel = np.sort(np.random.default_rng([7, c.index]).choice(c.q_count, 40, replace=False))
                picks.append((c.q_begin + sel, rows[torch.as_tensor(sel, device=dev)].clone()))
                assert torch.equal(rows[:16], yd[torch.as_tensor(hit[c.index], device=dev)])
        interp.interp_array_ring(qd, chunk, consumer, slots=ring)
        assert len(sums[name]) == 13
        if name == "bucketed":
            idx = np.concatenate([p[0] for p in picks])
            got = torch.cat([p[1] for p in picks]).cpu().numpy()
            _, _, ref = oracle.interp1d_cubic(x, y, a, b, q[idx])
            assert np.array_equal(got, ref)
    assert [int(s.item()) for s in sums["bucketed"]] == [int(s.item()) for s in sums["gather"]]


class _DeviceRows:
    """A raw device pointer with a row pitch (a chunk of a library-owned ring) as a torch tensor."""

    def __init__(self, ptr, rows, lanes, pitch_elems, dt):
        isz = np.dtype(dt).itemsize
        self.__cuda_array_interface__ = {"shape": (rows, lanes), "strides": (pitch_elems * isz, isz),
                                         "typestr": np.dtype(dt).str, "data": (ptr, False), "version": 2}


def test_target_default_geometry(pkg):
    """The benchmarked geometry itself (bench.py defaults; north-star Target): 4096 knots x 4096 f64 lanes, 1e7
    queries in 4 chunks of 2.5e6 through a 2-slot STRIPED ring of 2 x 81.9 GB -- caller-owned (`striped_ring()`)
    and library-owned.  Per chunk: the int64 checksum of all 1.024e10 outputs is equal between the gather and the
    bucketed formulation (grouped records, slice histograms and uint32 record indices at 2.5e6 queries per launch);
    256 sampled rows per chunk equal the oracle bit for bit; knot queries return the data row exactly
    (interp_array: src/interp1d/mod.rs:197-211)."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    dev = torch.device("cuda:0")
    free_b, _ = torch.cuda.mem_get_info(dev)
    n = L = 4096; Q = 10_000_000; chunk = 2_500_000; nchunks = 4
    need = 2 * chunk * L * 8 + (6 << 30)
    if free_b < need:
        pytest.skip(f"needs {need / 1e9:.0f} GB of free device memory, {free_b / 1e9:.0f} GB available")
    rng = np.random.default_rng(45)
    x = knots("rand", n, rng, np.float64)
    y = rng.uniform(0.0, 1.0, (n, L))
    yd = torch.as_tensor(y, device=dev)
    interp = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
    q = np.concatenate([np.random.default_rng([45, c]).uniform(x[0], x[-1], chunk) for c in range(nchunks)])
    hit = rng.integers(0, n - 1, nchunks * 16).reshape(nchunks, 16)
    for c in range(nchunks):
        q[c * chunk:c * chunk + 16] = x[hit[c]]                 # 16 knot queries at the head of every chunk
    q[Q - 1] = x[-1]                                            # and the right end of the axis as the very last query
    qd = torch.as_tensor(q, device=dev)
    st, a, b = oracle.cubic_build(x, y)
    assert st == oracle.OK
    sums, picks = {}, {}

    def run(name, path, slots):
        interp.strategy.path = path
        sums[name], picks[name] = [], []

        def consumer(c, rows):
            if rows is None:                                    # library-owned ring: wrap the raw chunk
                assert c.row_stride == 2 * L
                rows = torch.as_tensor(_DeviceRows(c.out, c.q_count, L, c.row_stride, np.float64), device=dev)
            assert c.q_count == chunk and c.row_stride == 2 * L and c.slot == c.index % 2
            sums[name].append(rows.view(torch.int64).sum())     # stream-ordered device reduction of the whole chunk
            sel = np.sort(np.random.default_rng([9, c.index]).choice(c.q_count, 256, replace=False))
            picks[name].append((c.q_begin + sel, rows[torch.as_tensor(sel, device=dev)].clone()))
            assert torch.equal(rows[:16], yd[torch.as_tensor(hit[c.index], device=dev)])
            if c.index == nchunks - 1:
                assert torch.equal(rows[chunk - 1], yd[n - 1])
        if slots is None:
            interp.interp_array_ring(qd, chunk, consumer, n_slots=2)
        else:
            interp.interp_array_ring(qd, chunk, consumer, slots=slots)
        assert len(sums[name]) == nchunks
        idx = np.concatenate([p[0] for p in picks[name]])
        got = torch.cat([p[1] for p in picks[name]]).cpu().numpy()
        _, _, ref = oracle.interp1d_cubic(x, y, a, b, q[idx])
        assert np.array_equal(got, ref), name
        return [int(s_.item()) for s_ in sums[name]]

    ring = pkg.striped_ring(chunk, L, 2, np.float64, 0)
    assert ring[0].stride(0) == 2 * L and ring[1].data_ptr() - ring[0].data_ptr() == L * 8
    s_b = run("bucketed/striped_ring()", pkg.PATH_BUCKETED, ring)
    assert pkg.profile_read(reset=False)["last_path"] == "bucketed"
    s_g = run("gather/striped_ring()", pkg.PATH_GATHER, ring)
    s_a = run("auto/striped_ring()", pkg.PATH_AUTO, ring)
    assert s_b == s_g == s_a
    del ring
    picks.clear(); sums.clear()
    torch.cuda.empty_cache()
    s_own = run("bucketed/library-owned", pkg.PATH_BUCKETED, None)
    assert s_own == s_b
    interp.strategy.trim()                                      # hands the 164 GB back
    interp.strategy.release()
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# resident locator
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_locator_matches_one_shot_search_and_oracle(pkg, dt):
    import torch
    rng = np.random.default_rng(21)
    for kind in ("lin", "rand", "jit", "log"):                  # the four grid families of bench_vector_extensions.rs
        for n in (2, 100, 4096, 70_000):
            k = knots(kind, n, rng, dt)
            loc = pkg.Locator(k)
            span = float(k[-1] - k[0])
            for Q in (1, 1000, 33_333):
                q = np.concatenate([rng.uniform(k[0] - 0.1 * span, k[-1] + 0.1 * span, Q).astype(dt), k[:50],
                                    np.array([np.inf, -np.inf, np.nan], dtype=dt)])
                got = loc.get_lower_index(q)
                exp = oracle.get_lower_index(k, q)
                assert np.array_equal(got, exp), (kind, n, Q)
                assert got[-1] == -1
                gd = loc.get_lower_index(torch.as_tensor(q, device="cuda:0"))
                assert gd.is_cuda and np.array_equal(gd.cpu().numpy(), exp)
            assert np.array_equal(loc.get_lower_index(q.reshape(-1, 1)).ravel(), exp)
            loc.release()
    # a locator built from device knots
    kd = torch.as_tensor(knots("rand", 1000, rng, dt), device="cuda:0")
    loc = pkg.Locator(kd)
    q = rng.uniform(0, 1, 5000).astype(dt)
    assert np.array_equal(loc.get_lower_index(torch.as_tensor(q, device="cuda:0")).cpu().numpy(),
                          oracle.get_lower_index(kd.cpu().numpy(), q))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n", [65, 4096, 30000])
def test_get_lower_index_jittered_axes(pkg, dt, n):
    """linspace + uniform(+-0.2/n) noise (benches/bench_vector_extensions.rs:36-40): the O(1) guess is tried
    first and is wrong for a fraction of the lanes."""
    rng = np.random.default_rng(n + 1)
    k = knots("jit", n, rng, dt)
    q = np.concatenate([rng.uniform(-0.1, 1.1, 20000).astype(dt), k, np.nextafter(k, dt(-np.inf)),
                        np.nextafter(k, dt(np.inf))])
    got = pkg.get_lower_index(k, q)
    assert np.array_equal(got, np.clip(np.searchsorted(k, q, side="right") - 1, 0, n - 2))
    assert np.array_equal(got, oracle.get_lower_index(k, q))


# ------------------------------------------------------------------------------------------------
# scratch reclaim, element types, lifetime
# ------------------------------------------------------------------------------------------------
def test_scratch_sets_are_bounded_under_thread_churn(pkg):
    """scratch is keyed by (stream, host thread); short-lived threads must not grow it without bound."""
    rng = np.random.default_rng(31)
    interp, x, y, a, b = _cubic(pkg, 50, 64, rng)
    q = rng.uniform(x[0], x[-1], 500)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    lib, h = pkg._capi.lib(), interp.strategy._h
    bad = []

    def work():
        if not np.array_equal(interp.interp_array(q), ref):
            bad.append(1)
    for _ in range(60):                                   # 60 distinct short-lived threads, one after the other
        t = threading.Thread(target=work); t.start(); t.join()
    assert not bad
    assert lib.ndi_interp1d_scratch_sets(h) <= 17         # 16 idle sets + the one in use
    interp.strategy.trim()
    assert lib.ndi_interp1d_scratch_sets(h) == 0
    assert np.array_equal(interp.interp_array(q), ref)    # and it keeps working after a trim


def test_mixed_element_types(pkg):
    """f32 device queries against f64 data get an f64 output (the data's element type), and a caller
    buffer of the wrong element type is refused instead of being overrun."""
    import torch
    rng = np.random.default_rng(32)
    interp, x, y, a, b = _cubic(pkg, 40, 256, rng)
    q32 = rng.uniform(x[0], x[-1], 3000).astype(np.float32)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q32.astype(np.float64))
    out = interp.interp_array(torch.as_tensor(q32, device="cuda:0"))
    assert out.dtype == torch.float64 and np.array_equal(out.cpu().numpy(), ref)
    with pytest.raises(TypeError, match="element type float32"):
        interp.interp_array_into(torch.as_tensor(q32, device="cuda:0"),
                                 torch.empty((3000, 256), dtype=torch.float32, device="cuda:0"))
    with pytest.raises(TypeError, match="element type float32"):
        interp.interp_array_into(q32, np.empty((3000, 256), dtype=np.float32))
    g = rng.random((9, 8, 4), dtype=np.float32)
    bi = pkg.Interp2DBuilder.new(g).build()
    qx = rng.uniform(0, 8, 100); qy = rng.uniform(0, 7, 100)            # f64 queries, f32 data
    _, _, _, ref2 = oracle.interp2d_bilinear(np.arange(9, dtype=np.float32), np.arange(8, dtype=np.float32), g,
                                            qx.astype(np.float32), qy.astype(np.float32))
    o2 = bi.interp_array(torch.as_tensor(qx, device="cuda:0"), torch.as_tensor(qy, device="cuda:0"))
    assert o2.dtype == torch.float32 and np.array_equal(o2.cpu().numpy(), ref2)
    with pytest.raises(TypeError, match="element type float64"):
        bi.interp_array_into(qx, qy, np.empty((100, 4)))


def test_async_launch_with_converted_host_queries_and_strided_error_rows(pkg):
    """an async batch whose host queries needed a dtype conversion keeps that copy alive until
    finish(); a strided host buffer keeps the rows after the first failing query untouched."""
    import gc
    import torch
    rng = np.random.default_rng(33)
    interp, x, y, a, b = _cubic(pkg, 40, 256, rng)
    q = rng.uniform(x[0] + 1e-3, x[-1] - 1e-3, 4000)
    q[2500] = 99.0
    out = torch.full((4000, 256), -1.0, dtype=torch.float64, device="cuda:0")
    interp.strategy.interp_array_into(interp, q.astype(np.float32), out,
                                      async_launch=True)   # f32 host queries: converted copy owned by the mirror
    gc.collect()
    with pytest.raises(pkg.InterpolateError.OutOfBounds) as ei:
        interp.strategy.finish()
    assert ei.value.index == 2500 and ei.value.value == 99.0
    # strided ArrayViewMut
    big = np.full((4000, 2, 256), -3.0)
    view = big[:, 1, :]
    with pytest.raises(pkg.InterpolateError.OutOfBounds):
        interp.interp_array_into(q, view)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q[:2500])
    assert np.array_equal(view[:2500], ref) and np.all(view[2500:] == -3.0) and np.all(big[:, 0, :] == -3.0)


# (several devices from one process: tests/test_gpu_sharded.py)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_bucket_index_edge_axes(pkg, dt):
    """The bucket index behind the search (kernels.hpp BucketIndex; used for >= 4096 queries on axes the O(1) formula
    guess does not resolve, n <= 65535): axes with almost every knot in one bucket, a tiny span on a large offset,
    the u16 limit, and queries on / next to every knot and bucket edge."""
    rng = np.random.default_rng(61)
    eps = np.finfo(dt).eps
    axes = {
        "clustered head": np.concatenate([np.linspace(0.0, 1e-6, 3000), np.linspace(1e-3, 1.0, 200)]),
        "clustered tail": np.concatenate([np.linspace(0.0, 1.0, 100), 1.0 + np.linspace(1e-7, 1e-5, 5000)]),
        "tiny span, big offset": 1000.0 + np.cumsum(rng.uniform(1, 3, 2000)) * 1000.0 * eps * 4,
        "two scales": np.unique(np.concatenate([rng.uniform(0, 1e-3, 4000), rng.uniform(0, 1e3, 4000)])),
        "n = 65535": np.cumsum(rng.uniform(0.5, 1.5, 65535)),
        "n = 65536 (no index)": np.cumsum(rng.uniform(0.5, 1.5, 65536)),
        "n = 65": np.cumsum(rng.uniform(0.5, 1.5, 65)),
    }
    for name, k in axes.items():
        k = np.unique(k.astype(dt))
        n = k.size
        span = float(k[-1] - k[0])
        # bucket edges of the index the library builds (m = smallest power of two >= 2n)
        m = 1
        while m < 2 * n:
            m *= 2
        edges = (k[0] + (np.arange(m + 1) / m) * span).astype(dt)
        q = np.concatenate([rng.uniform(k[0] - 0.01 * span, k[-1] + 0.01 * span, 20000).astype(dt), k,
                            np.nextafter(k, dt(-np.inf)), np.nextafter(k, dt(np.inf)), edges,
                            np.nextafter(edges, dt(-np.inf)), np.nextafter(edges, dt(np.inf)),
                            np.array([np.inf, -np.inf, np.nan, k[0], k[-1]], dtype=dt)])
        got = pkg.get_lower_index(k, q)
        exp = oracle.get_lower_index(k, q)
        assert np.array_equal(got, exp), name
        fin = ~np.isnan(q)
        assert np.array_equal(got[fin], np.clip(np.searchsorted(k, q[fin], side="right") - 1, 0, n - 2)), name
    # the same axes through a 2-D interpolator (both axes indexed in one launch) and the spline's locate + histogram
    import torch
    kx = np.unique(axes["two scales"].astype(dt)); ky = np.unique(axes["clustered head"].astype(dt))
    g = rng.random((kx.size, ky.size, 2)).astype(dt)
    bi = pkg.Interp2DBuilder.new(torch.as_tensor(g, device="cuda:0")).x(torch.as_tensor(kx, device="cuda:0")) \
        .y(torch.as_tensor(ky, device="cuda:0")).build()
    qx = rng.uniform(kx[0], kx[-1], 30000).astype(dt); qy = rng.uniform(ky[0], ky[-1], 30000).astype(dt)
    qy[:5000] = rng.uniform(0, 1e-6, 5000).astype(dt)          # into the cluster
    _, _, _, ref = oracle.interp2d_bilinear(kx, ky, g, qx, qy)
    assert np.array_equal(bi.interp_array(qx, qy), ref)
    y = rng.uniform(0, 1, (ky.size, 1024)).astype(dt)
    sp = pkg.Interp1DBuilder.new(y).x(ky).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(ky, y)
    qq = np.concatenate([rng.uniform(ky[0], ky[-1], 20000), rng.uniform(0, 1e-6, 20000)]).astype(dt)
    _, _, ref = oracle.interp1d_cubic(ky, y, a, b, qq)
    for path in (pkg.PATH_BUCKETED, pkg.PATH_GATHER):
        sp.strategy.path = path
        assert np.array_equal(sp.interp_array(qq), ref), path


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_two_level_grouping_skewed_batches_and_ring(pkg, capfd, dt):
    """The two-level 2-D grouping (tile row, then tile: coarse_scatter2d_kernel / fine_scatter2d_kernel, forced with
    NDI_GROUP_TWO_LEVEL=1 -- AUTO takes it only for batches of a million queries) on batches that are anything but evenly
    spread: every query in ONE tile, in one tile row, in one tile column, two far corners, a batch smaller than a
    workgroup; tile rows without a single query; then chunk by chunk through the two-stream ring pipeline (256-thread
    grouping workgroups beside the evaluation).  Bit-equal to the oracle (bilinear.rs:64-99) and to the gather order."""
    import torch
    rng = np.random.default_rng(77)
    nx, ny, Cn = 150, 131, 32 if dt == np.float32 else 16
    g = rng.uniform(-1, 1, (nx, ny, Cn)).astype(dt)
    x = knots("rand", nx, rng, dt); y = knots("rand", ny, rng, dt)
    interp = pkg.Interp2DBuilder.new(g).x(x).y(y).build()
    dev = torch.device("cuda:0")

    def batch(kind, Q):
        if kind == "one_tile":
            return rng.uniform(x[3], x[5], Q), rng.uniform(y[70], y[72], Q)
        if kind == "one_row":
            return rng.uniform(x[40], x[41], Q), rng.uniform(y[0], y[-1], Q)
        if kind == "one_col":
            return rng.uniform(x[0], x[-1], Q), rng.uniform(y[-3], y[-1], Q)
        if kind == "corners":
            s = rng.integers(0, 2, Q).astype(bool)
            return np.where(s, rng.uniform(x[0], x[1], Q), rng.uniform(x[-2], x[-1], Q)), \
                np.where(s, rng.uniform(y[0], y[1], Q), rng.uniform(y[-2], y[-1], Q))
        return rng.uniform(x[0], x[-1], Q), rng.uniform(y[0], y[-1], Q)

    os.environ["NDI_GROUP_TWO_LEVEL"] = "1"
    os.environ["NDI_TRACE_PLAN"] = "1"
    try:
        for kind, Q in (("one_tile", 50_001), ("one_row", 33_333), ("one_col", 40_000), ("corners", 70_003), ("uniform", 100),
                        ("uniform", 120_000)):
            qx, qy = batch(kind, Q)
            qx = qx.astype(dt); qy = qy.astype(dt)
            ref = oracle.interp2d_bilinear(x, y, g, qx, qy)[3].reshape(Q, Cn)
            interp.strategy.path = pkg.PATH_BUCKETED
            out = torch.full((Q, Cn), -5.0, dtype=torch.float32 if dt == np.float32 else torch.float64, device=dev)
            interp.interp_array_into(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), out)
            assert pkg.profile_read(reset=False)["last_path"] == "bucketed", kind
            assert np.array_equal(out.cpu().numpy(), ref), (kind, Q)
            assert "two-level grouping" in capfd.readouterr().err, kind      # (the plan line: no silent one-pass scatter)
        # the ring pipeline: search + two-level grouping of chunk k + 1 on the side stream while chunk k is evaluated
        Q, chunk = 90_001, 11_000
        qx, qy = batch("corners", Q)
        qx = qx.astype(dt); qy = qy.astype(dt)
        ref = oracle.interp2d_bilinear(x, y, g, qx, qy)[3].reshape(Q, Cn)
        got = np.zeros_like(ref)
        ring = pkg.striped_ring(chunk, Cn, 3, dt, 0)

        def consumer(c, rows):
            got[c.q_begin:c.q_begin + c.q_count] = rows.cpu().numpy()
        interp.interp_array_ring(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev), chunk, consumer, slots=ring)
        assert np.array_equal(got, ref)
        assert capfd.readouterr().err.count("two-level grouping") >= (Q + chunk - 1) // chunk
    finally:
        os.environ.pop("NDI_GROUP_TWO_LEVEL", None)
        os.environ.pop("NDI_TRACE_PLAN", None)
    interp.strategy.path = pkg.PATH_GATHER
