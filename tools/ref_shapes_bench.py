#!/usr/bin/env python3
"""tools/ref_shapes_bench.py -- the reference's own criterion bench shapes through the C ABI, beside the CPU port.

Shapes (no results are published by the reference, only the definitions):
  benches/bench_interp1d.rs:12-47, 82-122        100 knots (index axis) f64, scalar data and (100, 5) data, 1e4 uniform
                                                 queries in [0, 99]: one 1e4-query `interp_array`, and 2500 calls of 4
  benches/bench_interp1d_query_dim.rs:16-66      the same 1e4 queries as 2500 x (4,), 625 x (4,4), 125 x (5,4,4)
  benches/bench_interp2d.rs:12-18, 87-131        100 x 100 and 100 x 100 x 5 grids, 2500 calls of 4 (x, y) queries
  benches/bench_vector_extensions.rs:19-78       get_lower_index, 1000 queries in [-0.1, 1.1] on five 100-knot grids
Every GPU number is host arrays in, host array out through the bare C ABI (what a Rust caller of the shim pays:
H2D, kernels, D2H, one synchronisation per call).  The CPU column is oracle/ (the single-threaded port) on the
same arrays.  Tiny calls are launch-latency bound on the GPU -- the table shows where the device starts to pay.
Prints one JSON document.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_product_package  # noqa: E402

import oracle  # noqa: E402

pkg = load_product_package()
cap = pkg._capi
lib = cap.lib()


def best_us(fn, min_time=0.25, min_reps=3):
    fn()
    times = []
    t_end = time.perf_counter() + min_time
    while len(times) < min_reps or time.perf_counter() < t_end:
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return round(float(np.median(times)) * 1e6, 2)


def rand(n, lo, hi, seed):
    return np.random.default_rng(seed).uniform(lo, hi, n)


def calls_1d(interp, chunks, lanes):
    h = interp.strategy._h
    opts, info = cap.EvalOpts(), cap.OobInfo()
    outs = [np.zeros((c.size, lanes)) for c in chunks]
    args = [(h, c.ctypes.data, c.size, o.ctypes.data, lanes, C.byref(opts), C.byref(info)) for c, o in zip(chunks, outs)]

    def run():
        for a in args:
            assert lib.ndi_interp1d_eval(*a) == 0
    return run, outs


def calls_2d(interp, xs, ys, lanes):
    h = interp.strategy._h
    opts, info = cap.EvalOpts(), cap.OobInfo()
    outs = [np.zeros((c.size, lanes)) for c in xs]
    args = [(h, cx.ctypes.data, cy.ctypes.data, cx.size, o.ctypes.data, lanes, C.byref(opts), C.byref(info))
            for cx, cy, o in zip(xs, ys, outs)]

    def run():
        for a in args:
            assert lib.ndi_interp2d_eval(*a) == 0
    return run, outs


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def cpu_calls(kind, x, data, chunks, a=None, b=None, ys=None):
    """The CPU port with pre-built arguments (no numpy wrapper work inside the timed loop)."""
    ol = oracle.lib()
    fail, axis = C.c_size_t(0), C.c_int(0)
    if kind == "bilinear":
        nx, ny = x.size, x.size
        Cn = data.shape[2] if data.ndim > 2 else 1
        outs = [np.zeros((c.size, Cn)) for c in chunks]
        args = [(_p(x), _p(x), _p(data), C.c_size_t(nx), C.c_size_t(ny), C.c_size_t(Cn), C.c_int(0), _p(cx), _p(cy),
                 C.c_size_t(cx.size), _p(o), C.c_size_t(Cn), C.c_int(1), C.byref(fail), C.byref(axis))
                for cx, cy, o in zip(chunks, ys, outs)]
        fn = ol.oracle_interp2d_bilinear_f64
    else:
        d2 = data.reshape(x.size, -1)
        L = d2.shape[1]
        outs = [np.zeros((c.size, L)) for c in chunks]
        if kind == "linear":
            args = [(_p(x), _p(d2), C.c_size_t(x.size), C.c_size_t(L), C.c_int(0), _p(c), C.c_size_t(c.size), _p(o),
                     C.c_size_t(L), C.c_int(1), C.byref(fail)) for c, o in zip(chunks, outs)]
            fn = ol.oracle_interp1d_linear_f64
        else:
            args = [(_p(x), _p(d2), _p(a), _p(b), C.c_size_t(x.size), C.c_size_t(L), C.c_int(0), _p(c), C.c_size_t(c.size),
                     _p(o), C.c_size_t(L), C.c_int(1), C.byref(fail)) for c, o in zip(chunks, outs)]
            fn = ol.oracle_interp1d_cubic_f64

    def run():
        for t in args:
            fn(*t)
    return run, outs


def main():
    res = {"_about": __doc__.split("\n")[0], "unit": "microseconds per criterion iteration (all 1e4 queries, or 1000 for "
           "get_lower_index); median of repeated runs"}
    q = rand(10_000, 0.0, 99.0, 123)
    x100 = np.arange(100.0)

    # ---- 1-D, scalar data ------------------------------------------------------------------------------
    y = rand(100, 0.0, 1.0, 42)
    interp = pkg.Interp1DBuilder.new(y).build()
    rows = {}
    for name, shape in (("interp_array 1D-long (1 call x 10000)", (1, 10_000)), ("interp_array (2500 calls x 4)", (2500, 4)),
                        ("interp_array 2D-query (625 calls x 16)", (625, 16)), ("interp_array 3D-query (125 calls x 80)", (125, 80))):
        chunks = [np.ascontiguousarray(c) for c in q.reshape(shape)]
        run, outs = calls_1d(interp, chunks, 1)
        gpu = best_us(run)
        crun, couts = cpu_calls("linear", x100, y, chunks)
        cpu = best_us(crun)
        assert np.array_equal(np.concatenate(outs), np.concatenate(couts))
        rows[name] = {"gpu_c_abi_us": gpu, "cpu_port_us": cpu}
    res["bench_interp1d scalar data (100 knots)"] = rows

    # ---- 1-D, (100, 5) data ------------------------------------------------------------------------------
    y5 = rand(500, 0.0, 1.0, 69).reshape(100, 5)
    interp5 = pkg.Interp1DBuilder.new(y5).build()
    rows = {}
    for name, shape in (("interp_array (2500 calls x 4)", (2500, 4)), ("interp_array (1 call x 10000)", (1, 10_000))):
        chunks = [np.ascontiguousarray(c) for c in q.reshape(shape)]
        run, outs = calls_1d(interp5, chunks, 5)
        gpu = best_us(run)
        crun, couts = cpu_calls("linear", x100, y5, chunks)
        cpu = best_us(crun)
        assert np.array_equal(np.concatenate(outs), np.concatenate(couts))
        rows[name] = {"gpu_c_abi_us": gpu, "cpu_port_us": cpu}
    res["bench_interp1d (100, 5) data"] = rows

    # ---- cubic spline on the same shapes (the strategy the north star names) ----------------------------
    spl = pkg.Interp1DBuilder.new(y5).strategy(pkg.CubicSpline.new()).build()
    st, a, b = oracle.cubic_build(x100, y5)
    rows = {}
    for name, shape in (("interp_array (2500 calls x 4)", (2500, 4)), ("interp_array (1 call x 10000)", (1, 10_000))):
        chunks = [np.ascontiguousarray(c) for c in q.reshape(shape)]
        run, outs = calls_1d(spl, chunks, 5)
        gpu = best_us(run)
        crun, couts = cpu_calls("cubic", x100, y5, chunks, a=a, b=b)
        cpu = best_us(crun)
        assert np.array_equal(np.concatenate(outs), np.concatenate(couts))
        rows[name] = {"gpu_c_abi_us": gpu, "cpu_port_us": cpu}
    res["CubicSpline, (100, 5) data"] = rows

    # ---- 2-D ----------------------------------------------------------------------------------------------
    qx, qy = rand(10_000, 0.0, 99.0, 123), rand(10_000, 0.0, 99.0, 96)
    for label, g in (("bench_interp2d 100 x 100", rand(10_000, 0.0, 1.0, 42).reshape(100, 100)),
                     ("bench_interp2d 100 x 100 x 5", rand(50_000, 0.0, 1.0, 69).reshape(100, 100, 5))):
        lanes = 1 if g.ndim == 2 else 5
        bi = pkg.Interp2DBuilder.new(g).build()
        rows = {}
        for name, shape in (("interp_array (2500 calls x 4)", (2500, 4)), ("interp_array (1 call x 10000)", (1, 10_000))):
            xs = [np.ascontiguousarray(c) for c in qx.reshape(shape)]
            ys = [np.ascontiguousarray(c) for c in qy.reshape(shape)]
            run, outs = calls_2d(bi, xs, ys, lanes)
            gpu = best_us(run)
            crun, couts = cpu_calls("bilinear", x100, g, xs, ys=ys)
            cpu = best_us(crun)
            assert np.array_equal(np.concatenate(outs), np.concatenate(couts))
            rows[name] = {"gpu_c_abi_us": gpu, "cpu_port_us": cpu}
        res[label] = rows

    # ---- get_lower_index: five 100-knot grid families, 1000 queries (bench_vector_extensions.rs:19-78) ----
    rng = np.random.default_rng(42)
    bunched = np.unique(np.sort((np.linspace(0, 1, 20)[:, None] + rng.uniform(-0.001, 0.001, (20, 5))).ravel()))
    grids = {"Linspaced": np.linspace(0.0, 1.0, 100), "Uniform rng": np.unique(np.sort(rand(100, 0.0, 1.0, 42))),
             "Linspace bunched": bunched, "Linspace noisy": np.sort(np.linspace(0, 1, 100) + rand(100, -0.002, 0.002, 42)),
             "Logspaced": np.logspace(0.0, 8.0, 100, base=2.0)}
    q1k = rand(1000, -0.1, 1.1, 69)
    q1k_log = rand(1000, 0.95, 256.5, 69)
    rows = {}
    for name, k in grids.items():
        k = np.ascontiguousarray(k)
        qq = q1k_log if name == "Logspaced" else q1k
        out = np.empty(qq.size, dtype=np.int64)
        one_shot = best_us(lambda: lib.ndi_get_lower_index_batch(cap.F64, 0, k.ctypes.data, k.size, qq.ctypes.data, qq.size,
                                                                 out.ctypes.data, cap.MEM_HOST))
        h = C.c_void_p()
        assert lib.ndi_locator_create(cap.F64, 0, k.ctypes.data, k.size, cap.MEM_HOST, C.byref(h)) == 0
        resident = best_us(lambda: lib.ndi_locator_eval(h, qq.ctypes.data, qq.size, out.ctypes.data, cap.MEM_HOST, None))
        assert np.array_equal(out, oracle.get_lower_index(k, qq))
        lib.ndi_locator_destroy(h)
        rows[name] = {"knots": int(k.size), "gpu_one_shot_us": one_shot, "gpu_resident_locator_us": resident,
                      "cpu_port_us": best_us(lambda: oracle.get_lower_index(k, qq))}
    res["get_lower_index, 1000 queries, host arrays"] = rows

    # ---- get_lower_index throughput: 4096-knot grids, 1e6 / 1e7 device-resident queries ----------------------
    import torch
    rows = {}
    n = 4096
    fam = {"linspace": np.linspace(0.0, 1.0, n), "sorted uniform": np.unique(np.sort(rand(2 * n, 0.0, 1.0, 42)))[:n],
           "noisy linspace": np.sort(np.linspace(0, 1, n) + rand(n, -0.2 / n, 0.2 / n, 42)),
           "logspace": np.logspace(-2.0, 0.0, n)}
    for name, k in fam.items():
        k = np.ascontiguousarray(k)
        loc = pkg.Locator(k)
        row = {}
        for nq in (1_000_000, 10_000_000):
            qd = torch.as_tensor(rand(nq, k[0], k[-1], 7), device="cuda:0")
            od = torch.empty(nq, dtype=torch.int64, device="cuda:0")
            stream = torch.cuda.current_stream().cuda_stream
            call = lambda: lib.ndi_locator_eval(loc._h, qd.data_ptr(), nq, od.data_ptr(), cap.MEM_DEVICE, stream)  # noqa: E731
            us = best_us(call)
            row[f"{nq} queries"] = {"gpu_us": us, "Gqueries_s": round(nq / us / 1e3, 2)}
        sample = rand(200_000, k[0], k[-1], 7)
        cpu = best_us(lambda: oracle.get_lower_index(k, sample))
        row["cpu_port_Gqueries_s (200000 queries, 1 thread)"] = round(sample.size / cpu / 1e3, 4)
        assert np.array_equal(loc.get_lower_index(sample), oracle.get_lower_index(k, sample))
        rows[name] = row
    res["get_lower_index throughput, 4096 knots f64, device-resident queries -> int64 indices"] = rows
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
