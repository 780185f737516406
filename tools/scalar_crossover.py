"""Crossover between the one-thread-per-query kernel and the query-order kernel for 1-2 lane rows.
    NDI_SMALL_MAXQ=<huge|0> python tools/scalar_crossover.py"""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    for n, L in ((100, 1), (1024, 1), (1024, 2), (8192, 1)):
        x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n]
        y = rng.uniform(0, 1, (x.size, L)).astype(dt)
        it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
        for Q in (10_000, 30_000, 100_000, 300_000, 1_000_000, 3_000_000, 10_000_000):
            q = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])).clamp(float(x[0]), float(x[-1]))
            out = torch.empty((Q, L), dtype=tdt, device=dev)
            for _ in range(3):
                it.strategy.interp_array_into(it, q, out, async_launch=True)
            torch.cuda.synchronize()
            reps = 50 if Q <= 1_000_000 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                it.strategy.interp_array_into(it, q, out, async_launch=True)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
            it.strategy.finish()
            print(json.dumps({"dtype": np.dtype(dt).name, "n": int(x.size), "lanes": L, "queries": Q, "us": round(us, 2),
                              "maxq": os.environ.get("NDI_SMALL_MAXQ", "default")}), flush=True)
