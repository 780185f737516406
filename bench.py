#!/usr/bin/env python3
"""bench.py -- headline benchmark of the interp_array hot path on MI355X.

Metric (BASELINE.json): interp_array Mpoints/s (points = queries x lanes), 1-D CubicSpline f64, plus the
achieved fraction of the HBM roofline.

Default workload = the north-star **Target**: 4096 knots x 4096 f64 lanes, 1e7 queries per GPU.  The whole
output (327.7 GB) exceeds the 288 GB of HBM, so a "step" is one pass of `interp_array` over the batch through
the library's device-output ring (ndi_interp1d_eval_ring): 4 chunks of 2.5e6 *distinct* queries, each located,
grouped and evaluated into one of 2 ring slots of 81.9 GB (one allocation, slots interleaved row by row); nothing is
copied to the host.  Tables, queries and
the ring are resident in HBM when the timed region starts.  N>1: every rank runs the same per-GPU batch on its
own device (weak scaling), tables replicated, no collective on the data path; with N > 1 the per-GPU batch defaults to
C4's share (BASELINE configs[3]: 1.25e7 queries per GPU, 5 chunks) and every rank also runs C5's per-GPU share
(configs[4]) after the timed region -- the line carries both with per-rank clocks.

    python bench.py                                  # N=1, Target
    python bench.py --gpus 8 --steps 20 --warmup 5   # spawns 8 ranks itself (RCCL barrier / max-reduce only)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # external launcher: same
    python bench.py --workload c2|c3|c5|c2-linear|c2-f32|c1 ...                    # secondary measurements
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


def synth_target_queries(x, nq, chunk, rank):
    """Target / C4 share: nq queries in distinct chunks -- one PRNG stream per (rank, chunk), seeds following
    benches/bench_interp1d.rs:13-15 (123 for queries)."""
    parts = []
    for c, off in enumerate(range(0, nq, chunk)):
        m = min(chunk, nq - off)
        parts.append(np.random.default_rng([123, rank, c]).uniform(x[0], x[-1], m))
    return np.concatenate(parts)


def measure_traffic_pmc(args, kernel_substr, timeout_s=90):
    """HBM-side bytes per launch of the dominant kernel, MEASURED in this invocation: two short child runs of this
    very script under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes; counters are KiB, gfx950 FETCH_SIZE tallies 128-byte requests
    at 64 B: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024).  The parent must have released its device memory.  Returns
    (bytes_per_launch, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself running under a profiler"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-check",
             "--no-gather-leg", "--placement-probe", "0", "--no-secondary", "--no-pmc",
             "--knots", str(args.knots), "--lanes", str(args.lanes), "--queries", str(args.queries),
             "--chunk", str(args.chunk), "--ring-slots", str(args.ring_slots), "--path", args.path,
             "--ring-layout", args.ring_layout]
    means = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        progress(f"measuring HBM traffic: child run under rocprofv3 --pmc {counter}")
        d = tempfile.mkdtemp(prefix="ndi_pmc_", dir="/tmp")
        try:
            r = subprocess.run([prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                               timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                    if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]]
            if not vals:
                return None, f"no {counter} rows for {kernel_substr}"
            means[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as e:  # noqa: BLE001 -- a profiling failure must not fail the bench
            return None, f"{type(e).__name__}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = means["FETCH_SIZE"][0], means["WRITE_SIZE"][0]
    return int((2 * fetch + write) * 1024), {"FETCH_SIZE_KiB": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1),
                                             "launches_sampled": means["FETCH_SIZE"][1],
                                             "read_bytes": int(2 * fetch * 1024), "write_bytes": int(write * 1024)}


def progress(msg):
    """One line on stderr per stage after the timed region: a long default run keeps showing signs of life."""
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Cores this process may actually run on: the affinity mask, capped by a cgroup CPU quota when there is one
    (a GPU box hands each job a share of a large host: os.cpu_count() alone overstates it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline(x, y, q_all, budget_s=20.0):
    """The CPU port (oracle/, -ffp-contract=off) timed on this box's host cores on a bounded sample of the
    same workload: blocks of 2048 queries into a reused output block until ~budget_s of CPU work."""
    sys.path.insert(0, ROOT)
    import oracle
    flags = oracle.use_native()      # -O3 -march=native, compiled on this host
    n, lanes = y.shape
    t0 = time.perf_counter()
    st, a, b = oracle.cubic_build(x, y)
    build_s = time.perf_counter() - t0
    assert st == oracle.OK
    res = {}
    ncpu, usable = os.cpu_count() or 1, usable_cores()
    legs = [("1t", 1, budget_s / 2), ("all", usable, budget_s / 2)]     # 1 thread and all host cores
    if usable > 16:                                                     # this process may use (affinity / cgroup quota)
        legs.append(("share", 16, budget_s / 4))                        # 16 threads = one GPU's share of the box
    for label, threads, budget in legs:
        blk = 2048 if threads == 1 else 512 * threads
        out = np.zeros((blk, lanes))
        done, t_used, pos = 0, 0.0, 0
        while t_used < budget:
            if pos + blk > q_all.size:
                pos = 0
            t0 = time.perf_counter()
            s, _, _ = oracle.interp1d_cubic(x, y, a, b, q_all[pos:pos + blk], nthreads=threads, out=out)
            t_used += time.perf_counter() - t0
            assert s == oracle.OK
            done += blk
            pos += blk
        res[label] = (done * lanes / t_used / 1e6, done, threads)
    res["flags"] = flags
    res["host_cores"], res["usable_cores"] = ncpu, usable
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
    else:                        # C2 itself (one 1e6-query batch into one resident buffer), Linear f64 / CubicSpline f32 on its shape
        n = lanes = 4096; nq = args.queries
        x, yv, q = synth_c2(n, lanes, nq, rank)
        dt, tdt, strat = {"c2": (np.float64, torch.float64, pkg.CubicSpline.new()),
                          "c2-linear": (np.float64, torch.float64, None),
                          "c2-f32": (np.float32, torch.float32, pkg.CubicSpline.new())}[args.workload]
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


def bilinear_leg(pkg, torch, dev, nx, C, nq, steps=20, warmup=3, probe=False, rank=0):
    """One secondary 2-D leg: random-knot axes, uniform in-range queries, device-resident output; HIP-event kernel
    time from the library's profile, wall-clock end to end (search [+ grouping] + evaluation).  Measured with the
    formulation AUTO picks and, when that is the tile-grouped order, with the gather order as well (the gather
    order is the one SURVEY 8(d)'s bytes-per-point model describes)."""
    rng = np.random.default_rng(42)
    x = np.unique(rng.uniform(0, 1, 2 * nx).astype(np.float32))[:nx]
    y = np.unique(rng.uniform(0, 1, 2 * nx).astype(np.float32))[:nx]
    g = torch.rand((nx, nx, C), dtype=torch.float32, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
    interp = pkg.Interp2DBuilder.new(g).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    del g
    torch.cuda.empty_cache()
    sx, sy = (123, 96) if rank == 0 else ([123, rank], [96, rank])   # (every rank its own block of the scattered queries)
    qx = torch.as_tensor(np.random.default_rng(sx).uniform(x[0], x[-1], nq).astype(np.float32), device=dev)
    qy = torch.as_tensor(np.random.default_rng(sy).uniform(y[0], y[-1], nq).astype(np.float32), device=dev)
    out0 = torch.empty((nq, C), dtype=torch.float32, device=dev)
    cur = {"out": out0}
    step = lambda: interp.strategy.interp_array_into(interp, qx, qy, cur["out"], async_launch=True)
    alg = nq * C * 20 + nq * 8                                   # 5 x 4 B per point + the query pair

    import ctypes
    cap = pkg._capi

    def traced_kernel():
        """Which gather-order kernel the library takes for this batch: one call with NDI_TRACE_PLAN=1, the library's
        stderr line captured at file-descriptor level."""
        import tempfile
        sys.stderr.flush()
        saved = os.dup(2)
        with tempfile.TemporaryFile(mode="w+b") as tf:
            os.dup2(tf.fileno(), 2)
            os.environ["NDI_TRACE_PLAN"] = "1"
            try:
                step()
                interp.strategy.finish()
            finally:
                del os.environ["NDI_TRACE_PLAN"]
                os.dup2(saved, 2)
                os.close(saved)
            tf.seek(0)
            txt = tf.read().decode(errors="replace")
        return "eval_fused2d_kernel" if "[ndi plan] fused2d" in txt else "eval_bilinear_kernel"

    def measure(path, out=None):
        out = out0 if out is None else out
        cur["out"] = out
        interp.strategy.path = path
        for _ in range(warmup):
            step()
        interp.strategy.finish()
        gather_kernel = traced_kernel()
        # (a) per-stage kernel times: the library's HIP events (two event records per stage on the launch stream)
        pkg.profile_enable(True); pkg.profile_read(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el_prof = time.perf_counter() - t0        # wall time of the SAME loop the stage times come from (event recording on)
        interp.strategy.finish()
        prof = pkg.profile_read(reset=True); pkg.profile_enable(False)
        kms = prof["eval_ms"] / max(1, prof["eval_launches"])
        tiled = prof["last_path"] == "bucketed"
        # (b) end to end, event recording off, through the bare C ABI (what a compiled host pays per call) -- `steps`
        # stream-ordered async calls, one synchronisation.  host_ms_per_call is the time the host spends INSIDE the
        # call: when it stays below the GPU's time per step the stream never runs dry and a step costs what its
        # kernels cost.
        opts = cap.EvalOpts(); info = cap.OobInfo()
        opts.q_memspace = cap.MEM_DEVICE; opts.out_memspace = cap.MEM_DEVICE; opts.path = path; opts.async_launch = 1
        opts.stream = torch.cuda.current_stream(dev).cuda_stream   # the stream the mirror (and finish()) uses
        fn, h = cap.lib().ndi_interp2d_eval, interp.strategy._h
        cargs = (h, qx.data_ptr(), qy.data_ptr(), nq, out.data_ptr(), C, ctypes.byref(opts), ctypes.byref(info))
        for _ in range(warmup):
            assert fn(*cargs) == 0
        interp.strategy.finish()
        torch.cuda.synchronize()
        host = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            h0 = time.perf_counter()
            assert fn(*cargs) == 0
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        interp.strategy.finish()
        # (c) the same loop through the Python mirror (ctypes marshalling + the mirror's argument handling per call)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el_py = time.perf_counter() - t0
        interp.strategy.finish()
        gpu_ms = kms + prof["locate_ms"] / max(1, prof["locate_launches"]) + prof["group_ms"] / steps
        r = {"path": "tile-grouped" if tiled else "gather",
             "kernel": "eval_bilinear_tiles_kernel" if tiled else gather_kernel, "kernel_ms": round(kms, 4),
             "locate_ms": round(prof["locate_ms"] / max(1, prof["locate_launches"]), 4),
             "group_ms": round(prof["group_ms"] / steps, 4),
             "ms_per_step": round(el / steps * 1e3, 4), "Mpoints_s": round(nq * C * steps / el / 1e6, 1),
             "timed_through": "bare C ABI (ndi_interp2d_eval, async_launch), event recording off",
             "host_ms_per_call": round(host / steps * 1e3, 4),
             # the stage times are HIP-event intervals of a run WITH event recording (each interval carries the event
             # commands' own dispatch gaps), so they are compared with that run's own wall clock -- never negative -- and
             # the unprofiled step (ms_per_step) is what a caller pays
             "profiled_ms_per_step": round(el_prof / steps * 1e3, 4),
             "step_minus_kernels_ms": round(el_prof / steps * 1e3 - gpu_ms, 4),
             "python_mirror_ms_per_step": round(el_py / steps * 1e3, 4)}
        if tiled:   # every grid value once + output + records: the bytes this formulation has to move
            comp = nx * nx * C * 4 + nq * C * 4 + nq * 16
            r["compulsory_bytes_per_launch"] = comp
            r["frac"] = round(comp / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            r["bound"] = "arithmetic + tile staging, not HBM: DESIGN.md 4.4, profiles/r03_c3_grouped.md"
        else:
            r["algorithmic_bytes_per_launch"] = alg
            r["frac"] = round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        return r

    res = {"workload": f"2D Bilinear, {nx}x{nx} grid x {C} channels f32, {nq} queries, random knots"}
    res.update(measure(pkg.PATH_AUTO))
    if res["path"] != "gather":
        res["gather_order"] = measure(pkg.PATH_GATHER)
    if nq * C * 4 >= pkg.OUTPUT_OWNED_MIN_BYTES:
        # Interp2D::interp_array allocates its own output (interp2d/mod.rs:175-196); the mirror takes it from ndi_output_alloc
        # (placement look-and-retry).  The caller-buffer figures above stay the leg's primary ones -- the first
        # torch.empty buffer, whatever it is; this is the same batch into a library-owned buffer, as interp_array runs it.
        lib_out = pkg.output_empty((nq, C), np.float32, dev.index or 0)
        r = measure(pkg.PATH_AUTO, lib_out)
        res["library_owned_output"] = {k: r[k] for k in ("kernel_ms", "locate_ms", "group_ms", "ms_per_step", "Mpoints_s", "frac") if k in r}
        res["library_owned_output"]["alloc"] = lib_out.ndi_output_info
        del lib_out
        pkg.output_trim()
    if probe:      # context only (round 5): a stripped-down kernel with the same four-corner access mix -- random cells, no
        # searches, token arithmetic.  It is NOT an upper bound (its lane mapping and in-flight depth are not tuned, and the
        # product kernel has beaten it): no ratio is derived from it.  The guide's figure for this kind of access
        # (MI355X_MICROARCH.md, random rows fetched once from a table far beyond the Infinity Cache) is 5.5-5.8 TB/s.
        res["access_mix_probe_ms"] = round(interp.strategy.probe_ceiling(out0, reps=7), 4)
        res["access_mix_probe_note"] = "context, not a ceiling; guide: random rows fetched once 5.5-5.8 TB/s"
    interp.strategy.release()
    cur.clear()
    del interp, qx, qy, out0
    torch.cuda.empty_cache()
    return res


def short_rows_leg(pkg, torch, dev, out_bytes=4e9, steps=5):
    """1-D CubicSpline on rows shorter than one workgroup pass (the reference's own data shapes are of this kind:
    scalar data, (100, 5) -- benches/bench_interp1d.rs:82-122): 1024 random knots, 4 GB of device-resident output per
    call, the formulation AUTO picks (query order with the search fused in / grouped by interval), HIP-event stage
    times, output TB/s, and the long-row bucketed kernel (32 KiB rows) on the SAME buffer as the yardstick."""
    rng = np.random.default_rng(0)
    res = {"what": "1D CubicSpline, 1024 random knots, 4 GB of output per call, device buffers; out_TBps = output bytes / "
                   "wall time of a whole interp_array_into call (search [+ grouping] + evaluation); interp_array_* = the same "
                   "with interp_array's semantics (fresh output: no range pre-pass)", "shapes": []}
    for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        el = np.dtype(dt).itemsize
        x = np.unique(rng.uniform(0, 1, 2048).astype(dt))[:1024]
        xd = torch.as_tensor(x, device=dev)
        buf = torch.empty(int(out_bytes) // el, dtype=tdt, device=dev)
        for L in (32768 // el, 8, 32, 64, 128, 256):
            Q = int(out_bytes // (L * el))
            yd = torch.rand((xd.numel(), L), dtype=tdt, device=dev)
            interp = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
            qd = (torch.rand(Q, dtype=tdt, device=dev) * (xd[-1] - xd[0]) * 0.999 + xd[0]).clamp(xd[0], xd[-1])
            out = buf[: Q * L].view(Q, L)
            call = lambda: interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
            call(); interp.strategy.finish()
            pkg.profile_enable(True); pkg.profile_read(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                call()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps
            interp.strategy.finish()
            prof = pkg.profile_read(reset=True); pkg.profile_enable(False)
            # interp_array semantics (the output is the call's own: NDI_EVAL_FRESH_OUTPUT -- the query-order kernels
            # test the range themselves, no pre-pass; the grouped forms have none to drop)
            fcall = lambda: interp.strategy.interp_array_into(interp, qd, out, async_launch=True, fresh=True)
            fcall(); interp.strategy.finish()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fcall()
            torch.cuda.synchronize()
            fwall = (time.perf_counter() - t0) / steps
            interp.strategy.finish()
            res["shapes"].append({"dtype": np.dtype(dt).name, "lanes": L, "queries": Q, "path": prof["last_path"],
                                  "ms": round(wall * 1e3, 4), "out_TBps": round(Q * L * el / wall / 1e12, 3),
                                  "Gpoints_s": round(Q * L / wall / 1e9, 1),
                                  "eval_ms": round(prof["eval_ms"] / steps, 4),
                                  "locate_ms": round(prof["locate_ms"] / steps, 4),
                                  "group_ms": round(prof["group_ms"] / steps, 4),
                                  "interp_array_ms": round(fwall * 1e3, 4),
                                  "interp_array_out_TBps": round(Q * L * el / fwall / 1e12, 3)})
            interp.strategy.release()
            del interp, qd, yd, out
        del buf
        torch.cuda.empty_cache()
    for r in res["shapes"]:
        ref = next(s for s in res["shapes"] if s["dtype"] == r["dtype"] and s["lanes"] * np.dtype(r["dtype"]).itemsize == 32768)
        r["frac_of_long_row_rate"] = round(r["out_TBps"] / ref["out_TBps"], 3)
    # the reference's own bench shapes scaled up to 1e8 / 5e7 queries: scalar data on 100 and 1024 knots, (100, 5)
    res["reference_shapes"] = []
    for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        el = np.dtype(dt).itemsize
        for n, L, Q in ((100, 1, 100_000_000), (1024, 1, 100_000_000), (100, 5, 50_000_000)):
            x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n]
            xd = torch.as_tensor(x, device=dev)
            yd = torch.rand((xd.numel(), L), dtype=tdt, device=dev)
            interp = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
            qd = (torch.rand(Q, dtype=tdt, device=dev) * (xd[-1] - xd[0]) * 0.999 + xd[0]).clamp(xd[0], xd[-1])
            out = torch.empty((Q, L), dtype=tdt, device=dev)
            walls = {}
            for fresh in (True, False):     # interp_array semantics (the output is the call's own, dropped on Err:
                # NDI_EVAL_FRESH_OUTPUT, no range pre-pass; the allocation itself stays outside the timed loop) and
                # interp_array_into semantics (caller-owned buffer: rows at / after a failing query stay untouched)
                call = lambda: interp.strategy.interp_array_into(interp, qd, out, async_launch=True, fresh=fresh)
                call(); interp.strategy.finish()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    call()
                torch.cuda.synchronize()
                walls[fresh] = (time.perf_counter() - t0) / steps
                interp.strategy.finish()
            wall, fwall = walls[False], walls[True]
            # `ms` / `Gqueries_s` are interp_array_into (a caller-owned buffer; comparable with rounds 1-4); the
            # interp_array figures exclude the allocation of the output (the reference's includes its Array::zeros)
            res["reference_shapes"].append({"dtype": np.dtype(dt).name, "knots": int(xd.numel()), "lanes": L, "queries": Q,
                                            "ms": round(wall * 1e3, 4), "Gqueries_s": round(Q / wall / 1e9, 1),
                                            "out_TBps": round(Q * L * el / wall / 1e12, 3),
                                            "io_frac_of_peak": round(Q * (L + 1) * el / wall / 1e9 / HBM_PEAK_GBS, 4),
                                            "semantics": "interp_array_into (caller-owned buffer, range pre-pass)",
                                            "interp_array_excl_alloc_ms": round(fwall * 1e3, 4),
                                            "interp_array_excl_alloc_Gqueries_s": round(Q / fwall / 1e9, 1),
                                            "interp_array_excl_alloc_io_frac_of_peak": round(Q * (L + 1) * el / fwall / 1e9 / HBM_PEAK_GBS, 4)})
            interp.strategy.release()
            del interp, qd, yd, out
        torch.cuda.empty_cache()
    return res


def long_rows_leg(pkg, torch, dev, traffic_store, steps=3):
    """BASELINE configs[1] itself (`c2`: 1-D CubicSpline, 4096 knots x 4096 f64 lanes, 1e6 queries into ONE resident 32.8 GB
    buffer; cubic_spline.rs:791-830) and the same shape for the two long-row kernels the headline does not time: 1-D Linear
    f64 (linear.rs:73-98; SURVEY 8(d): 24 B per point) and CubicSpline f32.  AUTO takes the bucketed formulation
    (Q >= 5 (n - 1)): compulsory bytes = output + tables + records once; the SURVEY 8(d) gather-model ratio is printed next
    to it, never as `frac`.  Output buffers: `frac` / `kernel_ms` are those of the FIRST buffer the allocator hands out (no
    selection); three caller-style buffers (torch.empty) and three library-owned ones (ndi_output_alloc: what the mirrors'
    interp_array -- the reference's Array::zeros, interp1d/mod.rs:209 -- allocates: physical chunks spread over the device's
    memory) are all timed and listed.  PMC traffic: the stored figure of the same launch from profiles/traffic.json."""
    res = {}
    n = lanes = 4096
    nq = 1_000_000
    for key, dt, tdt, strat_name in (("c2", np.float64, torch.float64, "cubic"), ("c2_linear", np.float64, torch.float64, "linear"),
                                     ("c2_f32", np.float32, torch.float32, "cubic")):
        x, yv, q = synth_c2(n, lanes, nq, 0)
        x = np.unique(x.astype(dt)); yv = yv[:x.size].astype(dt)
        q = np.clip(q.astype(dt), x[0], x[-1])
        el = np.dtype(dt).itemsize
        b = pkg.Interp1DBuilder.new(torch.as_tensor(yv, device=dev)).x(torch.as_tensor(x, device=dev))
        interp = (b.strategy(pkg.CubicSpline.new()) if strat_name == "cubic" else b).build()
        qd = torch.as_tensor(q, device=dev)

        def time_into(out):
            step = lambda: interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
            step(); interp.strategy.finish()
            pkg.profile_enable(True); pkg.profile_read(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            wall_c = (time.perf_counter() - t0) / steps
            interp.strategy.finish()
            prof_c = pkg.profile_read(reset=True); pkg.profile_enable(False)
            return prof_c["eval_ms"] / max(1, prof_c["eval_launches"]), wall_c, prof_c

        bufs = [torch.empty((nq, lanes), dtype=tdt, device=dev) for _ in range(3)]
        runs = [time_into(o) for o in bufs]
        del bufs
        torch.cuda.empty_cache()
        owned, owned_info = [], []
        if hasattr(pkg, "output_empty"):          # library-owned outputs, one after the other (each freed before the next)
            for _ in range(3):
                o = pkg.output_empty((nq, lanes), dt, dev.index)
                owned_info.append(o.ndi_output_info)
                owned.append(time_into(o))
                del o
                pkg.output_trim()                 # (a kept buffer would be handed out again: every candidate is a fresh allocation here)
        # Interp1D::interp_array itself, call after call (interp1d/mod.rs:197-211): its allocation -- after the first call a
        # buffer ndi_output_free kept, handed out again without a refill (NDI_OUTPUT_UNINITIALIZED) -- plus the evaluation with
        # the kernel's own range test; the result is dropped before the next call, as a consumer that is done with it would
        ia_ms = ia_info = None
        if hasattr(pkg, "output_empty"):
            r = interp.interp_array(qd)
            torch.cuda.synchronize()
            ia_info = getattr(r, "ndi_output_info", None)
            del r
            t0 = time.perf_counter()
            for _ in range(steps):
                r = interp.interp_array(qd)
                del r
            torch.cuda.synchronize()
            ia_ms = (time.perf_counter() - t0) / steps * 1e3
            pkg.output_trim()
        kms, wall, prof = runs[0]                  # the FIRST allocation, no selection
        ntab = x.size + (2 * (x.size - 1) if strat_name == "cubic" else 0)
        comp = nq * lanes * el + ntab * lanes * el + nq * 16
        model = nq * lanes * (5 if strat_name == "cubic" else 3) * el + nq * el
        bucketed = prof["last_path"] == "bucketed"
        stored = (traffic_store or {}).get(f"{key}_bytes_per_launch")
        frac_of = lambda k: round(comp / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        res[key] = {"workload": f"1D {'CubicSpline' if strat_name == 'cubic' else 'Linear'}, {x.size} knots x {lanes} lanes "
                                f"{np.dtype(dt).name}, {nq} queries, one resident output buffer ({nq * lanes * el / 1e9:.1f} GB)"
                                + (" = BASELINE configs[1]" if key == "c2" else ""),
                    "path": prof["last_path"], "kernel": "eval_bucketed_kernel" if bucketed else "eval_rows_kernel",
                    "kernel_ms": round(kms, 4), "ms_per_step": round(wall * 1e3, 4),
                    "placement": "first of three caller-style buffers (torch.empty), no selection",
                    "kernel_ms_per_output_buffer": [round(r[0], 4) for r in runs],
                    "frac_per_output_buffer": [frac_of(r[0]) for r in runs],
                    "kernel_ms_library_owned_outputs": [round(r[0], 4) for r in owned],
                    "frac_library_owned_outputs": [frac_of(r[0]) for r in owned],
                    "library_owned_info": owned_info,
                    "Mpoints_s": round(nq * lanes / wall / 1e6, 1),
                    "interp_array_ms_per_call": round(ia_ms, 4) if ia_ms else None,
                    "interp_array_Mpoints_s": round(nq * lanes / (ia_ms * 1e-3) / 1e6, 1) if ia_ms else None,
                    "interp_array_note": "Interp1D::interp_array call after call, allocation included (a kept library-owned buffer, "
                                         "no refill), result dropped before the next call; first call's allocation: " + json.dumps(ia_info),
                    "compulsory_bytes_per_launch": int(comp), "frac": frac_of(kms),
                    "bytes_basis": "compulsory bytes per launch (output + tables + query records, once)",
                    "survey_8d_model_bytes": int(model),
                    "gather_model_ratio": round(model / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic": stored, "traffic_source": "stored (profiles/traffic.json, rocprofv3 --pmc of this launch)" if stored else None,
                    "traffic_frac": round(stored / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if stored else None,
                    "stages_ms_per_step": {k: round(prof[k + "_ms"] / steps, 4) for k in ("locate", "group", "eval")}}
        interp.strategy.release()
        del interp, qd
        torch.cuda.empty_cache()
    return res


def c2_variants_leg(pkg, torch, dev, steps=3):
    """SURVEY 8(d)'s secondary runs at BASELINE configs[1]'s shape (4096 knots x 4096 f64 lanes, 1e6 queries, one resident
    32.8 GB buffer): the four knot families of benches/bench_vector_extensions.rs:19-78 -- g1 linspace (the O(1) guess of
    vector_extensions.rs:70-90 is exact), g2 sorted-unique uniform (the headline's), g3 linspace + uniform(+-0.2/n) noise,
    g4 logspace -- and, on g2, SORTED queries (cache reuse in the gather order), each with both formulations: kernel time
    (library HIP events), the search pass, and the rate on the SURVEY 8(d) gather model (40 B per point) for the gather
    kernel / on compulsory bytes for the bucketed one."""
    n = lanes = 4096
    nq = 1_000_000
    rng = np.random.default_rng(42)
    fam = {"g1_linspace": np.linspace(0.0, 1.0, n),
           "g2_random": np.unique(rng.uniform(0.0, 1.0, 2 * n))[:n],
           "g3_jittered": np.sort(np.linspace(0.0, 1.0, n) + rng.uniform(-0.2 / n, 0.2 / n, n)),
           "g4_logspace": np.logspace(-3.0, 0.0, n)}
    yd = torch.rand((n, lanes), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
    out = torch.empty((nq, lanes), dtype=torch.float64, device=dev)
    table_bytes = (n + 2 * (n - 1)) * lanes * 8
    rows = []
    for name, x in fam.items():
        interp = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
        q = np.random.default_rng(123).uniform(x[0], x[-1], nq)
        for order in (("unsorted", "sorted") if name == "g2_random" else ("unsorted",)):
            qd = torch.as_tensor(np.sort(q) if order == "sorted" else q, device=dev)
            r = {"knots": name, "queries": order}
            for pname, path in (("bucketed", pkg.PATH_BUCKETED), ("gather", pkg.PATH_GATHER)):
                interp.strategy.path = path
                step = lambda: interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
                step(); interp.strategy.finish()
                pkg.profile_enable(True); pkg.profile_read(reset=True)
                for _ in range(steps):
                    step()
                interp.strategy.finish()
                p = pkg.profile_read(reset=True); pkg.profile_enable(False)
                kms = p["eval_ms"] / max(1, p["eval_launches"])
                nbytes = nq * lanes * 8 + nq * 16 + table_bytes if pname == "bucketed" else nq * lanes * 40 + nq * 8
                r[pname] = {"kernel_ms": round(kms, 4), "locate_ms": round(p["locate_ms"] / max(1, p["locate_launches"]), 4),
                            "group_ms": round(p["group_ms"] / steps, 4),
                            ("frac_compulsory" if pname == "bucketed" else "survey_8d_ratio"): round(nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            rows.append(r)
        interp.strategy.release()
        del interp
    del out, yd
    torch.cuda.empty_cache()
    return {"what": "BASELINE configs[1] shape, knot families g1-g4 and sorted queries (SURVEY 8(d) secondary runs)", "rows": rows}


def reference_shapes_2d_leg(pkg, torch, dev, steps=5):
    """The reference's 2-D bench shapes (benches/bench_interp2d.rs:12-18, 87-92: a 100 x 100 scalar grid and a
    100 x 100 x 5 grid) scaled to 5e7 / 2e7 queries on device buffers, f64 and f32: wall time of a whole
    interp_array_into call, Gqueries/s, and the query-in / row-out stream as a fraction of the HBM peak."""
    rng = np.random.default_rng(0)
    rows = []
    for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        el = np.dtype(dt).itemsize
        for nx, ny, C, Q in ((100, 100, 1, 50_000_000), (100, 100, 5, 20_000_000)):
            x = np.cumsum(rng.uniform(0.5, 1.5, nx)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, ny)).astype(dt)
            grid = torch.rand((nx, ny, C), dtype=tdt, device=dev)
            it = pkg.Interp2DBuilder.new(grid).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
            qx = torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])
            qy = torch.rand(Q, dtype=tdt, device=dev) * float(y[-1] - y[0]) * 0.999 + float(y[0])
            out = torch.empty((Q, C), dtype=tdt, device=dev)
            walls = {}
            for fresh in (True, False):     # interp_array / interp_array_into semantics: see short_rows_leg
                call = lambda: it.strategy.interp_array_into(it, qx, qy, out, async_launch=True, fresh=fresh)
                call(); it.strategy.finish()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    call()
                torch.cuda.synchronize()
                walls[fresh] = (time.perf_counter() - t0) / steps
                it.strategy.finish()
            wall, fwall = walls[False], walls[True]
            io = Q * (C + 2) * el
            rows.append({"dtype": np.dtype(dt).name, "grid": [nx, ny, C], "queries": Q, "ms": round(wall * 1e3, 4),
                         "Gqueries_s": round(Q / wall / 1e9, 1), "io_TBps": round(io / wall / 1e12, 3),
                         "io_frac_of_peak": round(io / wall / 1e9 / HBM_PEAK_GBS, 4),
                         "semantics": "interp_array_into (caller-owned buffer, range pre-pass)",
                         "interp_array_excl_alloc_ms": round(fwall * 1e3, 4),
                         "interp_array_excl_alloc_Gqueries_s": round(Q / fwall / 1e9, 1),
                         "interp_array_excl_alloc_io_frac_of_peak": round(io / fwall / 1e9 / HBM_PEAK_GBS, 4)})
            it.strategy.release()
            del it, qx, qy, out, grid
        torch.cuda.empty_cache()
    return rows


def host_path_leg(pkg, torch, dev, nq=100_000, reps=3):
    """The literal drop-in: host ndarrays in, host ndarray out (what a caller of the unpatched reference hands over), on
    BASELINE configs[1]'s tables -- 4096 knots x 4096 f64 lanes -- with 1e5 queries (3.3 GB of output per call; the
    full 1e6 queries would be 32.8 GB of pageable host memory per call).  The library streams the rows through a
    device staging buffer; the rate is the PCIe link's, not the kernel's -- reported here so that a reader of the
    bench line sees it, never part of `value`."""
    n = lanes = 4096
    x, yv, q = synth_c2(n, lanes, nq, 0)
    interp = pkg.Interp1DBuilder.new(torch.as_tensor(yv, device=dev)).x(torch.as_tensor(x, device=dev)) \
        .strategy(pkg.CubicSpline.new()).build()
    out = np.empty((nq, lanes))
    out[:] = 0.0                                     # touch the pages once, outside the timing
    interp.interp_array_into(q, out)
    t0 = time.perf_counter()
    for _ in range(reps):
        interp.interp_array_into(q, out)
    wall = (time.perf_counter() - t0) / reps
    interp.strategy.release()
    return {"workload": f"1D CubicSpline, {n} knots x {lanes} lanes f64, {nq} queries, host arrays in, host array out "
                        f"({nq * lanes * 8 / 1e9:.2f} GB of output per call, pageable memory)",
            "ms_per_call": round(wall * 1e3, 2), "output_GBps": round(nq * lanes * 8 / wall / 1e9, 1),
            "Mpoints_s": round(nq * lanes / wall / 1e6, 1),
            "bound": "PCIe (device -> host copy of every row); the HBM-resident rate is the headline's"}


def in_process_sharded_leg(args, pkg, torch, x, y, steps=3, budget_s=90.0):
    """When this ONE process sees several devices (the N = 1 run on a multi-GPU node): the Target batch per device
    through ndi_interp1d_eval_ring_sharded -- replicas by device-to-device copy, one library call per step, one host
    thread per device inside the library, no collective.  Reported after the timed region; never part of `value`."""
    devs = [int(d) for d in args.sharded_leg_devices.split(",")] if args.sharded_leg_devices else \
        list(range(pkg.device_count()))
    ndev = len(devs)
    if ndev < 2:
        return None
    t_leg = time.perf_counter()      # wall-clock guard: the whole leg (replicas, rings, steps) stays inside budget_s
    try:
        nq, chunk, lanes = args.queries, args.chunk, args.lanes
        # every device must have room for its share of rings and tables -- a device someone else is using is left alone
        per_dev = {d: devs.count(d) * (args.ring_slots * chunk * lanes * 8 + (2 << 30)) for d in set(devs)}
        for d, need in per_dev.items():
            free_b, _ = torch.cuda.mem_get_info(d)
            if free_b < need:
                return {"devices": devs, "skipped": f"device {d} has {free_b / 1e9:.0f} GB free, the leg needs {need / 1e9:.0f} GB"}
        first = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=f"cuda:{devs[0]}")).x(torch.as_tensor(x, device=f"cuda:{devs[0]}")) \
            .strategy(pkg.CubicSpline.new().device(devs[0])).build()
        reps = [first] + first.replicate(devs[1:])
        blocks = [torch.as_tensor(synth_target_queries(x, nq, chunk, 1000 + i), device=f"cuda:{d}") for i, d in enumerate(devs)]
        total = nq * ndev
        # the library splits the whole batch with ndi_shard_bounds: equal blocks of nq here
        assert all(pkg.sharding.shard_bounds(total, d, ndev) == (d * nq, (d + 1) * nq) for d in range(ndev))

        def step():
            pkg.sharding.interp_array_ring_sharded(reps, blocks, chunk_queries=chunk, consumer=None, n_slots=args.ring_slots)
        setup_s = time.perf_counter() - t_leg
        t0 = time.perf_counter()
        step()                        # first step: allocates the library-owned rings and the per-shard scratch
        for d in set(devs):
            torch.cuda.synchronize(d)
        first_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        step()                        # second step: warm
        for d in set(devs):
            torch.cuda.synchronize(d)
        warm_s = time.perf_counter() - t0
        left = budget_s - (time.perf_counter() - t_leg)
        n_timed = max(0, min(steps, int(left / max(warm_s, 1e-3)) - 1))
        el, note = warm_s, "the one warm step (wall-clock guard: no budget left for a timed loop)"
        if n_timed >= 1:
            t0 = time.perf_counter()
            for _ in range(n_timed):
                step()
            for d in set(devs):
                torch.cuda.synchronize(d)
            el, note = (time.perf_counter() - t0) / n_timed, f"mean of {n_timed} steps"
        res = {"devices": devs, "queries_per_device": nq, "ms_per_step": round(el * 1e3, 3), "timed": note,
               "Mpoints_s": round(total * lanes / el / 1e6, 1),
               "setup_s": round(setup_s, 2), "first_step_s": round(first_s, 2),
               "leg_wall_s": round(time.perf_counter() - t_leg, 2), "budget_s": budget_s,
               "what": "one process, one ndi_interp1d_eval_ring_sharded call per step over all visible devices"}
        for r in reps:
            r.strategy.release()
        del reps, blocks
        torch.cuda.empty_cache()
        return res
    except Exception as e:  # noqa: BLE001 -- an extra leg must not fail the bench
        return {"devices": devs, "error": f"{type(e).__name__}: {e}"}


def secondary_legs(pkg, torch, dev):
    """Short legs for the other BASELINE configs, run AFTER the timed Target region (never inside it): C3
    (configs[2]), C5's per-GPU share (configs[4]) and C1 (configs[0], the reference's CPU-runnable case, host arrays
    in and out through the C ABI next to the CPU port).  Mirrors benches/bench_interp2d.rs:12-131 and
    benches/bench_interp1d.rs:33-37."""
    import ctypes
    sys.path.insert(0, ROOT)
    import oracle
    try:
        traffic_store = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        traffic_store = None
    # the long tables first (they go to a line of their own, printed BEFORE the contract line), the BASELINE configs last
    detail = {"short_rows": short_rows_leg(pkg, torch, dev),
              "reference_shapes_2d": reference_shapes_2d_leg(pkg, torch, dev),
              "c2_knot_families_and_sorted_queries": c2_variants_leg(pkg, torch, dev)}
    sec = {"detail_line": "the stdout line before this one (prefix `detail `, then JSON) carries short_rows (+ reference_shapes), reference_shapes_2d and the knot-family / sorted-query runs"}
    sec["reference_shapes_summary"] = reference_summary(detail)
    sec["host_path"] = host_path_leg(pkg, torch, dev)
    sec.update(long_rows_leg(pkg, torch, dev, traffic_store))
    sec["c5_share"] = bilinear_leg(pkg, torch, dev, 8192, 16, 12_500_000, probe=True)
    sec["c3"] = bilinear_leg(pkg, torch, dev, 2048, 64, 10_000_000)
    n, nq = 1024, 10_000
    rng = np.random.default_rng(42)
    yv = rng.uniform(0, 1, n); q = np.random.default_rng(123).uniform(0, n - 1, nq)
    x = np.arange(n, dtype=np.float64)
    ref = oracle.interp1d_linear(x, yv, q)[2][:, 0]
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.interp1d_linear(x, yv, q)
    cpu_us = (time.perf_counter() - t0) / reps * 1e6
    interp = pkg.Interp1DBuilder.new(yv).build()
    cap = pkg._capi
    out = np.zeros(nq); opts = cap.EvalOpts(); info = cap.OobInfo()
    fn, h = cap.lib().ndi_interp1d_eval, interp.strategy._h
    cargs = (h, q.ctypes.data, nq, out.ctypes.data, 1, ctypes.byref(opts), ctypes.byref(info))
    for _ in range(30):
        assert fn(*cargs) == 0
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(*cargs)
    gpu_us = (time.perf_counter() - t0) / reps * 1e6
    sec["c1"] = {"workload": "1D Linear, 1024 f64 knots (index axis), scalar data, 1e4 queries, host arrays in and out",
                 "gpu_us": round(gpu_us, 1), "cpu_us": round(cpu_us, 1), "bit_exact": bool(np.array_equal(out, ref)),
                 "gpu_path": "C ABI host to host (zero-copy: memcpy into a pinned buffer, one fused search+evaluate launch that reads and writes it through its host mapping, one sync, memcpy out)",
                 "cpu_path": "oracle port, 1 thread"}
    return sec, detail


def reference_summary(detail):
    """The reference's own bench shapes (benches/bench_interp1d.rs:12-47, 82-122, benches/bench_interp2d.rs:12-18, 87-92) as
    [Gqueries/s of interp_array_into, of interp_array without its allocation, the latter's q-in / row-out stream as a
    fraction of the HBM peak] -- the compact copy of the detail line's tables that rides in the contract line."""
    out = {"columns": ["interp_array_into Gq/s", "interp_array (excl. alloc) Gq/s", "interp_array io frac of peak"]}
    pick = lambda r: [r["Gqueries_s"], r["interp_array_excl_alloc_Gqueries_s"], r["interp_array_excl_alloc_io_frac_of_peak"]]
    for r in detail["short_rows"]["reference_shapes"]:
        out[f"1d_{r['knots']}x{r['lanes']}_{r['dtype']}"] = pick(r)
    for r in detail["reference_shapes_2d"]:
        g = r["grid"]
        out[f"2d_{g[0]}x{g[1]}x{g[2]}_{r['dtype']}"] = pick(r)
    out["short_rows_out_TBps_by_lanes"] = {f"{r['dtype']}x{r['lanes']}": r["out_TBps"] for r in detail["short_rows"]["shapes"]}
    return out


def secondary_summary(sec):
    """Five-odd scalars per BASELINE config, small enough to ride inside `config` (the part of the line a truncating
    reader keeps) and, once more, at the very END of the line."""
    def pick(d, *ks):
        return {k: d.get(k) for k in ks if d.get(k) is not None}
    out = {}
    for k in ("c3", "c5_share"):
        if k in sec:
            out[k] = pick(sec[k], "ms_per_step", "kernel_ms", "locate_ms", "group_ms", "frac", "Mpoints_s", "path", "step_minus_kernels_ms")
            lo = sec[k].get("library_owned_output")
            if lo:      # the same batch into a buffer from ndi_output_alloc (what interp_array writes into)
                out[k]["library_owned_output"] = pick(lo, "ms_per_step", "kernel_ms", "frac")
    for k in ("c2", "c2_linear", "c2_f32"):
        if k in sec:
            out[k] = pick(sec[k], "ms_per_step", "kernel_ms", "frac", "kernel_ms_per_output_buffer", "kernel_ms_library_owned_outputs",
                          "frac_library_owned_outputs", "interp_array_ms_per_call")
    if "host_path" in sec:
        out["host_path_GBps"] = sec["host_path"].get("output_GBps")
    if "c1" in sec:
        out["c1_us"] = [sec["c1"].get("gpu_us"), sec["c1"].get("cpu_us")]
    rs = sec.get("reference_shapes_summary", {})
    for k in ("2d_100x100x5_float64", "2d_100x100x5_float32", "2d_100x100x1_float64", "1d_100x5_float64", "1d_100x1_float64"):
        if k in rs:
            out[k + "_Gq_s_into_and_array"] = rs[k][:2]
    return out


def gather_ranks(torch, dist, world, rank, ms_per_step, kernel_ms, my_dev, cdev):
    """Self-verification of an N > 1 line: every rank's own clock, kernel time and device, gathered after the timed
    region, so that a reader can see the backend saw `world` ranks, on which devices, and which rank was slowest."""
    per = torch.zeros((world, 2), dtype=torch.float64, device=cdev)
    per[rank, 0], per[rank, 1] = ms_per_step, kernel_ms
    dist.all_reduce(per, op=dist.ReduceOp.SUM)
    devs = [None] * world
    dist.all_gather_object(devs, my_dev)
    per = per.cpu().numpy()
    return {"world": dist.get_world_size(), "backend": dist.get_backend(),
            "per_rank_ms": [round(float(v), 4) for v in per[:, 0]],
            "per_rank_kernel_ms": [round(float(v), 4) for v in per[:, 1]],
            "slowest_rank": int(np.argmax(per[:, 0])), "devices": devs,
            "distinct_devices": len({(d["uuid"], d["ordinal"]) for d in devs})}


def self_launch(args):
    """`python bench.py --gpus N` without an external launcher: start N fresh rank processes -- before this
    process has made any GPU call -- with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, wait for them and
    return the worst exit code.  Rank 0 prints the JSON line (the children share our stdout)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(args.gpus)]
    # poll all ranks: the first one that fails takes its siblings down (they would otherwise sit in RCCL init or
    # a collective waiting for it) and its exit code is ours
    deadline = time.time() + 3000
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = abs(r) or 1
        if time.time() > deadline:
            rc = 124
        if live and rc == 0:
            time.sleep(0.05)
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def check_rows_against_oracle(x, y, q_rows, got_rows):
    """Outside the timed region: sampled output rows of the device path against the CPU oracle, bit for bit."""
    sys.path.insert(0, ROOT)
    import oracle
    st, a, b = oracle.cubic_build(x, y)
    assert st == oracle.OK
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q_rows)
    err = float(np.max(np.abs(ref - got_rows))) if got_rows.size else 0.0
    return {"rows": int(got_rows.shape[0]), "bit_exact": bool(np.array_equal(ref, got_rows)), "max_abs_err": err}


def workload_name(n, lanes, nq, world):
    """N = 1: the north-star Target (BASELINE configs[1]'s tables, 1e7 queries).  N > 1: BASELINE configs[3] -- C4, 1e8
    queries sharded over 8 GPUs -- as its per-GPU share of 1.25e7 queries on every rank (weak scaling: at N = 8 the job is
    C4 itself).  Other sizes (rehearsals, sweeps) say so."""
    if (n, lanes) == (4096, 4096) and world > 1 and nq == 12_500_000:
        return f"C4 (BASELINE configs[3]) per-GPU share x {world}" + (" = C4" if world == 8 else "")
    if (n, lanes) == (4096, 4096) and nq == 10_000_000:
        return "Target"
    if world > 1:
        return "C4-shaped rehearsal (sizes overridden)"
    return "Target-shaped (sizes overridden)"


def run_target(args, pkg, torch, dist, dev, rank, world):
    n, lanes, nq, chunk = args.knots, args.lanes, args.queries, args.chunk
    x, y, _ = synth_c2(n, lanes, 1, rank)
    q = synth_target_queries(x, nq, chunk, rank)
    if args.sorted_queries:
        q = np.sort(q)
    yd, xd = torch.as_tensor(y, device=dev), torch.as_tensor(x, device=dev)
    t0 = time.perf_counter()
    interp = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    paths = {"auto": pkg.PATH_AUTO, "gather": pkg.PATH_GATHER, "bucketed": pkg.PATH_BUCKETED}
    interp.strategy.path = paths[args.path]
    qd = torch.as_tensor(q, device=dev)
    # the ring: ONE allocation (the first of this process, taken as it comes) with the slots interleaved row by
    # row -- the layout ndarray-interp_amd recommends and a library-owned ring uses;
    # --ring-layout separate = one buffer per slot, rows contiguous (round 1's layout, for comparison)
    if args.ring_layout == "striped":
        ring = pkg.striped_ring(chunk, lanes, args.ring_slots, np.float64, dev.index)
    else:
        ring = [torch.empty((chunk, lanes), dtype=torch.float64, device=dev) for _ in range(args.ring_slots)]
    nchunks = (nq + chunk - 1) // chunk
    seen = {"chunks": 0, "rows": 0}

    def consumer(c, rows):        # a real consumer would enqueue its work on c.stream here
        seen["chunks"] += 1
        seen["rows"] += c.q_count
        return None

    def step():
        interp.interp_array_ring(qd, chunk, consumer, slots=ring)   # raises if any query failed (none may)

    use_dist = world > 1 or args.force_dist    # --force-dist: the collectives run on RCCL with world size 1

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    pkg.profile_enable(True)
    pkg.profile_read(reset=True)
    seen.update(chunks=0, rows=0)
    fence()
    step_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()                       # returns after the step's last kernel has completed (one sync inside)
        step_s.append(time.perf_counter() - ts)
    fence()
    elapsed = time.perf_counter() - t0
    prof = pkg.profile_read(reset=True)
    pkg.profile_enable(False)
    assert seen["chunks"] == nchunks * args.steps and seen["rows"] == nq * args.steps, seen

    my_elapsed = elapsed
    my_kernel_ms = prof["eval_ms"] / max(1, prof["eval_launches"])
    props = torch.cuda.get_device_properties(dev)
    my_dev = {"rank": rank, "ordinal": dev.index, "uuid": str(getattr(props, "uuid", "")), "name": props.name}
    cdev = dev if args.backend == "nccl" else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    ranks = None
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ranks = gather_ranks(torch, dist, world, rank, my_elapsed / args.steps * 1e3, my_kernel_ms, my_dev, cdev)
    elapsed = float(t.item())
    points_per_step = nq * lanes
    value = world * points_per_step * args.steps / elapsed / 1e6

    # N > 1 (or --c5-leg): BASELINE configs[4]'s per-GPU share on EVERY rank at the same time, after the timed region --
    # 8192 x 8192 x 16 f32 grid replicated per device, this rank's 1.25e7 of the 1e8 scattered queries, no collective on
    # the data path; per-rank clocks gathered like the headline's
    c5 = None
    if world > 1 or args.c5_leg:
        fence()
        mine = bilinear_leg(pkg, torch, dev, args.c5_leg_grid, 16, args.c5_leg_queries, steps=10, warmup=2, rank=rank)
        fence()
        per = [mine]
        if use_dist:
            per = [None] * world
            dist.all_gather_object(per, mine)
        if rank == 0:
            slow = max(p["ms_per_step"] for p in per)
            c5 = {"workload": f"C5 (BASELINE configs[4]) per-GPU share: 2D Bilinear, {args.c5_leg_grid}x{args.c5_leg_grid} grid x 16 "
                              f"channels f32 replicated per device, {args.c5_leg_queries} scattered queries per GPU "
                              f"({world} x {args.c5_leg_queries} in all), every rank at the same time",
                  "per_rank_ms": [p["ms_per_step"] for p in per], "per_rank_kernel_ms": [p["kernel_ms"] for p in per],
                  "kernel": per[0]["kernel"], "path": per[0]["path"],
                  "ms_per_step": slow, "Mpoints_s": round(world * args.c5_leg_queries * 16 / (slow * 1e-3) / 1e6, 1),
                  "frac_rank0": per[0].get("frac"), "scaling": "weak", "timed_through": per[0]["timed_through"]}
    if rank != 0:
        return

    # ---- everything below is outside the timed region (rank 0 only) -------------------------------------
    kernel_ms = prof["eval_ms"] / max(1, prof["eval_launches"])          # one launch = one chunk
    per_launch_q = nq / nchunks
    pts = per_launch_q * lanes
    table_bytes = (n + 2 * (n - 1)) * lanes * 8
    bucketed = prof["last_path"] == "bucketed"
    # bytes the kernel must move per launch: the output once + the queries' records; the bucketed formulation
    # reads every table row once per launch, the gather formulation 4 operand rows per query
    comp_bytes = pts * 8 + per_launch_q * 16 + (table_bytes if bucketed else pts * 32)
    alg_bytes = pts * 40 + per_launch_q * 8                                 # SURVEY 8(d) gather model
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and (n, lanes, chunk) == (4096, 4096, 2_500_000):
        try:
            traffic = json.load(open(tpath)).get(f"{prof['last_path']}_bytes_per_launch_chunk2500000")
        except Exception:
            traffic = None
    achieved = comp_bytes / (kernel_ms * 1e-3) / 1e9
    line = {
        "metric": "interp_array Mpoints/s (queries x lanes), 1D cubic f64",
        "value": round(value, 1), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        # `value` is the wall-clock MEAN over the timed steps (max over ranks); the median of rank 0's per-step
        # wall times is reported next to it (the mean is the conservative side)
        "ms_per_step_median": round(float(np.median(step_s)) * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload_name(n, lanes, nq, world) +
                               f": 1D CubicSpline NotAKnot, {n} knots x {lanes} lanes f64, {nq} queries per GPU "
                               f"in {nchunks} distinct chunks of {chunk} through a {args.ring_slots}-slot device-output "
                               "ring (sorted-unique uniform knots, unsorted uniform in-range queries; output "
                               f"{nq * lanes * 8 / 1e9:.1f} GB per step, never copied to the host)",
                   "path": prof["last_path"], "sharding": f"queries x{world}, tables replicated, no collective",
                   "output_ring": {"slots": args.ring_slots, "slot_bytes": chunk * lanes * 8,
                                   "layout": "one allocation, slots interleaved row by row (striped_ring)"
                                             if args.ring_layout == "striped" else "one buffer per slot",
                                   "placement": "first allocation of the process, no selection"}},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     # achieved / frac are the COMPUTED compulsory bytes of one launch divided by this run's kernel
                     # time.  `traffic` starts as the PMC figure of the same workload from the committed
                     # profiles/traffic.json and is REPLACED below by this invocation's own measurement (two child
                     # runs under rocprofv3 --pmc) unless --no-pmc / N > 1 / no profiler.
                     "traffic_source": "stored (profiles/traffic.json, rocprofv3 --pmc of this workload)" if traffic else None,
                     "bytes_basis": "compulsory bytes per launch (output + tables + query records, once)",
                     "kernel": "eval_bucketed_kernel" if bucketed else "eval_rows_kernel",
                     "kernel_ms": round(kernel_ms, 4), "launches": prof["eval_launches"],
                     "compulsory_bytes_per_launch": int(comp_bytes),
                     "stored_traffic_frac": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     # SURVEY 8(d)'s gather-model bytes / time / peak: > 1 on the bucketed path because that
                     # formulation removes the per-query table re-reads the model assumes -- not a hardware fraction
                     "gather_model_ratio": round(alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "stages_ms_per_step": {"locate": round(prof["locate_ms"] / args.steps, 4),
                               "group": round(prof["group_ms"] / args.steps, 4),
                               "eval": round(prof["eval_ms"] / args.steps, 4)},
        "build_ms": round(build_ms, 2),
    }
    if ranks is not None:
        line["ranks"] = ranks
    if c5 is not None:
        line["c5_share"] = c5

    # sanity: sampled rows of every chunk against the CPU oracle (one extra pass, rows gathered on the device)
    if not args.no_check:
        rng = np.random.default_rng(7)
        picked_q, picked_rows = [], []

        def checking_consumer(c, rows):
            sel = np.sort(rng.choice(c.q_count, size=min(max(48, 480 // nchunks), c.q_count), replace=False))
            picked_q.append(q[c.q_begin + sel])
            picked_rows.append(rows[torch.as_tensor(sel, device=dev)].cpu().numpy())
            return None
        interp.interp_array_ring(qd, chunk, checking_consumer, slots=ring)
        line["check"] = check_rows_against_oracle(x, y, np.concatenate(picked_q), np.concatenate(picked_rows))
        assert line["check"]["bit_exact"], line["check"]

    # the north-star's own formulation (per-query gather of 4 operand rows) on the same batch, for the
    # "fraction of the HBM-read roofline" it asks for: algorithmic bytes of SURVEY 8(d) / time / peak
    if not args.no_gather_leg and bucketed:
        interp.strategy.path = pkg.PATH_GATHER
        interp.interp_array_ring(qd[:2 * chunk], chunk, None, slots=ring)
        pkg.profile_enable(True); pkg.profile_read(reset=True)
        for _pass in range(2):
            interp.interp_array_ring(qd, chunk, None, slots=ring)
        gp = pkg.profile_read(reset=True); pkg.profile_enable(False)
        gms = gp["eval_ms"] / max(1, gp["eval_launches"])
        line["gather_formulation"] = {
            "kernel": "eval_rows_kernel", "kernel_ms": round(gms, 4),
            "algorithmic_GBs": round(alg_bytes / (gms * 1e-3) / 1e9, 1),
            # algorithmic bytes / time / peak: may exceed 1 (part of the 384 MiB table set is served on-die), so it
            # is a ratio of the SURVEY 8(d) model to the peak, not a hardware fraction; the read share below is
            "algorithmic_ratio_to_peak": round(alg_bytes / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "read_only_frac_of_peak": round(pts * 32 / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "Mpoints_s": round(pts / (gms * 1e-3) / 1e6, 1)}
        # SURVEY 8(d)'s formula applies to the GATHER kernel (4 operand rows in + 1 row out per query): both bases sit in
        # `roofline` -- `frac` (compulsory bytes, the shipped bucketed kernel) and these two (the gather kernel)
        line["roofline"]["survey_8d_frac"] = line["gather_formulation"]["algorithmic_ratio_to_peak"]
        line["roofline"]["survey_8d_read_frac"] = line["gather_formulation"]["read_only_frac_of_peak"]
        line["roofline"]["survey_8d_kernel"] = "eval_rows_kernel (ndi_path GATHER, the north star's literal formulation): " \
            "40 B per point / its own launch time / peak; > 1 means part of the 384 MiB table set is served on-die"
        interp.strategy.path = paths[args.path]

    # placement sensitivity of the output stream: the same chunk evaluated into every slot of the
    # ring in use, and into the slots of a second ring of the other layout; reported, never used for the headline
    if args.placement_probe > 0:
        def probe_ms(c):      # ms per 1e6 queries of one evaluation of c.shape[0] queries into c
            m = c.shape[0]
            interp.strategy.interp_array_into(interp, qd[:m], c, async_launch=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(2):
                interp.strategy.interp_array_into(interp, qd[:m], c, async_launch=True)
            e1.record(); e1.synchronize()
            return round(e0.elapsed_time(e1) / 2 / (m / 1e6), 3)
        ring_ms = [probe_ms(c) for c in ring]
        other_ms = []
        probe_q = min(chunk, 1_000_000)             # separate buffers: 1e6 queries = 32.8 GB each
        free_b, _ = torch.cuda.mem_get_info(dev)
        k = max(0, min(args.placement_probe, int((free_b - (8 << 30)) // (probe_q * lanes * 8))))
        if k:
            other = [torch.empty((probe_q, lanes), dtype=torch.float64, device=dev) for _ in range(k)]
            other_ms = [probe_ms(c) for c in other]
            del other
        interp.strategy.finish()
        line["placement"] = {"ring_layout": args.ring_layout, "ms_per_1e6_queries_ring_slots": ring_ms,
                             "ms_per_1e6_queries_separate_32.8GB_buffers": other_ms,
                             "spread_ring_slots": round(max(ring_ms) / min(ring_ms) - 1, 4),
                             "spread_separate": round(max(other_ms) / min(other_ms) - 1, 4) if other_ms else None}
        torch.cuda.empty_cache()

    if world == 1 and not args.no_secondary:
        interp.strategy.release()
        del ring, qd, interp, yd, xd
        torch.cuda.empty_cache()
        progress("secondary legs: C3, C5 share, C1")
        line["secondary"], detail = secondary_legs(pkg, torch, dev)
        line["config"]["secondary_summary"] = secondary_summary(line["secondary"])
        # The contract is ONE JSON line on stdout.  The long tables go out first, on a line that carries a `detail ` prefix and is
        # therefore NOT a JSON line to any parser -- the contract line stays the only JSON line and the last line of the output.
        print("detail " + json.dumps({"detail": "tables of the secondary legs (the contract line follows as the LAST line)", **detail}),
              flush=True)
        if pkg.device_count() >= 2 or args.sharded_leg_devices:
            progress("in-process sharded leg over the visible devices")
            leg = in_process_sharded_leg(args, pkg, torch, x, y)
            if leg:
                leg["speedup_vs_this_run_one_device"] = round(leg.get("Mpoints_s", 0.0) / value, 3) if "Mpoints_s" in leg else None
                line["in_process_sharded"] = leg

    if world == 1 and not args.no_pmc:
        # the parent holds no device memory any more (released before the secondary legs): two short child runs of
        # the same workload under rocprofv3 --pmc give this invocation's own HBM traffic per launch
        if "secondary" not in line:
            interp.strategy.release()
            del ring, qd, interp, yd, xd
            torch.cuda.empty_cache()
        measured, detail = measure_traffic_pmc(args, "eval_bucketed_kernel" if bucketed else "eval_rows_kernel")
        rf = line["roofline"]
        if measured:
            rf["traffic"] = measured
            rf["traffic_source"] = "measured by this run (child runs under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE)"
            rf["traffic_detail"] = detail
            rf["traffic_frac"] = round(measured / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            rf.pop("stored_traffic_frac", None)
        else:
            rf["traffic_measurement_failed"] = detail

    if world == 1 and not args.no_cpu_baseline:
        progress("CPU baseline (oracle port on the host cores, ~25 s)")
        res, build_s = cpu_baseline(x, y, q)
        v1, done1, _ = res["1t"]
        vall, doneall, threads = res["all"]
        line["cpu_baseline"] = {"value": round(v1, 1), "unit": "Mpoints/s", "cores": 1, "kind": "port",
                                "sample": f"{done1} queries x {lanes} lanes of the same workload (~10 s), oracle/ serial "
                                          "loop in blocks of 2048 queries (the reference is single-threaded)",
                                "host_cores": res["host_cores"], "usable_cores": res["usable_cores"],
                                "all_cores": {"value": round(vall, 1), "cores": threads,
                                              "sample": f"{doneall} queries (~10 s), contiguous query blocks per thread"},
                                "per_gpu_share": ({"value": round(res["share"][0], 1), "cores": 16,
                                                   "sample": f"{res['share'][1]} queries (~5 s)"} if "share" in res else None),
                                "build_s": round(build_s, 2), "compiler_flags": res["flags"]}
    # key order of the contract line: the long `secondary` object and, after it, a compact copy of its BASELINE-config
    # scalars come LAST, so that a reader who keeps only the tail of the output still sees C2 / C3 / C5
    if "secondary" in line:
        sec = line.pop("secondary")
        line["secondary"] = sec
        line["tail_summary"] = line["config"]["secondary_summary"]
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--knots", type=int, default=4096)
    ap.add_argument("--lanes", type=int, default=4096)
    ap.add_argument("--queries", type=int, default=None,
                    help="queries per GPU per step (target: 10000000; 12500000 = C4's per-GPU share; c2: 1000000)")
    ap.add_argument("--chunk", type=int, default=2_500_000,
                    help="target: queries per ring chunk (one launch of every kernel per chunk; larger chunks mean fewer "
                         "table reads per output byte: 1e6 / 2.5e6 / 5e6 -> 4.90 / 4.77 / 4.72 ms per 1e6 queries)")
    ap.add_argument("--ring-slots", type=int, default=2, help="target: device-output ring slots (chunk x lanes x 8 B each; "
                    "82 GB at the default chunk)")
    ap.add_argument("--path", choices=["auto", "gather", "bucketed"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the sampled-rows check against the CPU oracle")
    ap.add_argument("--no-gather-leg", action="store_true", help="skip the extra pass with the gather formulation")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C5-share / C1 legs after the timed region")
    ap.add_argument("--sharded-leg-devices", default=None,
                    help="rehearsal: device ordinals of the in-process sharded leg (default: every visible device when "
                         "there are at least two), e.g. 0,0 on a 1-GPU box")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two child runs under rocprofv3 --pmc that measure "
                    "this invocation's HBM traffic per launch (roofline.traffic then comes from profiles/traffic.json)")
    ap.add_argument("--ring-layout", choices=["striped", "separate"], default="striped",
                    help="target: striped = one allocation, slots interleaved row by row (recommended); separate = one "
                         "buffer per slot")
    ap.add_argument("--placement-probe", type=int, default=3,
                    help="after the timed region, evaluate one chunk into each ring slot and into this many further "
                         "allocations and report the spread (0 = skip); never used for the headline value")
    ap.add_argument("--even-axes", action="store_true", help="extra (c3/c5): default index axes 0..n instead of random knots")
    ap.add_argument("--sorted-queries", action="store_true", help="extra: sort the queries (cache reuse in the gather order)")
    ap.add_argument("--workload", choices=["target", "c2", "c3", "c5", "c2-linear", "c2-f32", "c1"], default="target",
                    help="target = headline (north-star Target; BASELINE configs[1] tables, 1e7 queries through the "
                         "ring); the others are secondary measurements for DESIGN.md")
    ap.add_argument("--c5-leg", action="store_true", help="N = 1: run the per-rank C5-share leg of the N > 1 line as well")
    ap.add_argument("--c5-leg-grid", type=int, default=8192, help="rehearsals: grid points per axis of the per-rank C5 leg")
    ap.add_argument("--c5-leg-queries", type=int, default=12_500_000, help="rehearsals: queries per rank of the C5 leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device-override", type=int, default=None, help="rehearsal only: put every rank on this GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: initialise torch.distributed anyway (backend nccl = RCCL, world size 1) so that "
                         "init_process_group(device_id=...), the barrier, the MAX all-reduce and all_gather_object of "
                         "the N > 1 path execute on a 1-GPU box")
    ap.add_argument("--launch-rehearsal", action="store_true",
                    help="no GPU work: only the launch / rendezvous / barrier / max-over-ranks skeleton (gloo), used by "
                         "the CPU tests to cover the N>1 self-launch")
    args = ap.parse_args()
    if args.queries is None:   # N = 1: the Target; N > 1: C4's per-GPU share (BASELINE configs[3]: 1e8 queries over 8 GPUs)
        args.queries = (12_500_000 if args.gpus > 1 else 10_000_000) if args.workload == "target" else 1_000_000

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))          # nothing in this process has touched the GPU

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device_override is not None:
        local_rank = args.device_override
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.launch_rehearsal:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            dist.barrier()
        if os.environ.get("NDI_BENCH_FAIL_RANK") == str(rank):       # test hook: this rank dies before the collective
            raise SystemExit(3)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        ranks = None
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ranks = gather_ranks(torch, dist, world, rank, 1.0 + rank, 0.5 + rank,
                                 {"rank": rank, "ordinal": local_rank, "uuid": f"rehearsal-{rank}", "name": "none"}, "cpu")
        c5 = None
        if world > 1:     # the per-rank C5-share leg's gather (all_gather_object of one dict per rank), with stand-in numbers
            per = [None] * world
            dist.all_gather_object(per, {"ms_per_step": 0.7 + 0.01 * rank, "kernel_ms": 0.6 + 0.01 * rank})
            c5 = {"per_rank_ms": [p["ms_per_step"] for p in per], "per_rank_kernel_ms": [p["kernel_ms"] for p in per],
                  "ms_per_step": max(p["ms_per_step"] for p in per)}
        if rank == 0:
            print(json.dumps({"rehearsal": True, "n_gpus": world, "max_over_ranks": float(t.item()),
                              "local_rank": local_rank, "ranks": ranks, "queries_per_gpu": args.queries, "c5_share": c5,
                              "workload": workload_name(args.knots, args.lanes, args.queries, world)}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:    # --force-dist without a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    pkg = load_package()
    if args.workload == "target":
        run_target(args, pkg, torch, dist, dev, rank, world)
    else:
        sys.path.insert(0, ROOT)
        extra_workload(args, pkg, torch, dev, rank, world)
    # success path only: a rank that raised above exits non-zero without entering another collective (its peers
    # may be inside a different one), and self_launch() takes the siblings down
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
