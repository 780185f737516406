"""Scalar / 2-lane data at large Q (the reference's "1D scalar interp_array" bench shape scaled up): the
one-thread-per-query kernel (eval_small_kernel) against the query-order kernel with the tables in LDS.
    NDI_SMALL_MODE=0|1 python tools/scalar_rows_probe.py"""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    for n, L, Q in ((100, 1, 100_000_000), (1024, 1, 100_000_000), (1024, 2, 100_000_000), (100, 5, 50_000_000), (100_000, 1, 50_000_000)):
        x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n]
        y = rng.uniform(0, 1, (x.size, L)).astype(dt)
        for strat_name in ("cubic", "linear"):
            strat = pkg.CubicSpline.new() if strat_name == "cubic" else pkg.Linear.new()
            it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(strat).build()
            q = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])).clamp(float(x[0]), float(x[-1]))
            out = torch.empty((Q, L), dtype=tdt, device=dev)
            it.strategy.interp_array_into(it, q, out, async_launch=True); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                it.strategy.interp_array_into(it, q, out, async_launch=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            it.strategy.finish()
            print(json.dumps({"dtype": np.dtype(dt).name, "n": int(x.size), "lanes": L, "queries": Q, "strategy": strat_name,
                              "ms": round(ms, 4), "Gqueries_s": round(Q / ms / 1e6, 1),
                              "out_TBps": round(Q * L * np.dtype(dt).itemsize / ms / 1e9, 3),
                              "small_mode": os.environ.get("NDI_SMALL_MODE", "1")}), flush=True)
            del out, q, it
