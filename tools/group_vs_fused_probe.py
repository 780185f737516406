"""Grouped (NDI_SHORT_MODE=3) against query order (NDI_SHORT_MODE=2) for short rows on long axes, where the tables
outgrow L2: 1 GB of output per call."""
import os, sys, time, json, numpy as np, torch
os.environ["NDI_TUNE_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
shapes = [(np.float32, 8192, 512), (np.float32, 16384, 512), (np.float64, 8192, 128), (np.float64, 16384, 256), (np.float32, 8192, 256),
          (np.float32, 8192, 128), (np.float64, 8192, 32), (np.float64, 16384, 32), (np.float64, 16384, 128), (np.float64, 16384, 8),
          (np.float64, 100_000, 8), (np.float64, 100_000, 32), (np.float64, 100_000, 128), (np.float32, 100_000, 32), (np.float64, 4096, 32),
          (np.float64, 4096, 64), (np.float64, 2048, 64)]
for strategy in (sys.argv[1:] or ["cubic", "linear"]):
    for dt, n, L in shapes:
        tdt = torch.float64 if dt == np.float64 else torch.float32
        el = np.dtype(dt).itemsize
        x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n]
        Q = int(min(1e9 // (L * el), 1e8))
        yd = torch.rand((x.size, L), dtype=tdt, device=dev)
        strat = pkg.CubicSpline.new() if strategy == "cubic" else pkg.Linear.new()
        it = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(x, device=dev)).strategy(strat).build()
        q = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])).clamp(float(x[0]), float(x[-1]))
        out = torch.empty((Q, L), dtype=tdt, device=dev)
        res = {}
        for name, mode, path in (("fused", "2", pkg.PATH_GATHER), ("grouped", "3", pkg.PATH_BUCKETED), ("auto", None, pkg.PATH_AUTO)):
            if mode is None: os.environ.pop("NDI_SHORT_MODE", None)
            else: os.environ["NDI_SHORT_MODE"] = mode
            it.strategy.path = path
            for _ in range(2):
                it.strategy.interp_array_into(it, q, out, async_launch=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                it.strategy.interp_array_into(it, q, out, async_launch=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 4 * 1e3
            it.strategy.finish()
            res[name] = round(Q * L * el / ms / 1e9, 2)
        os.environ.pop("NDI_SHORT_MODE", None)
        tab = x.size * L * el * (3 if strategy == "cubic" else 1)
        print(json.dumps({"strategy": strategy, "dtype": np.dtype(dt).name, "knots": int(x.size), "lanes": L, "row_bytes": L * el,
                          "table_MB": round(tab / 1e6, 1), "queries_per_interval": round(Q / x.size, 1), "out_TBps": res}), flush=True)
        it.strategy.release()
        del out, q, it, yd
        torch.cuda.empty_cache()
