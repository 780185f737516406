"""What AUTO delivers across the 1-D shape space (knots x lanes), 1 GB of output per call (at most 1e8 queries),
device buffers: output TB/s and Gqueries/s -- a map for spotting weak regions.   python tools/auto_map.py [cubic|linear]"""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
strategy = sys.argv[1] if len(sys.argv) > 1 else "cubic"
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    el = np.dtype(dt).itemsize
    for n in (100, 1024, 8192, 16384, 100_000):
        x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n]
        row = []
        for L in (1, 5, 8, 32, 128, 512, 4096):
            if x.size * L * el * 3 > 8e9:
                row.append("   -  "); continue
            Q = int(min(1e9 // (L * el), 1e8))
            yd = torch.rand((x.size, L), dtype=tdt, device=dev)
            strat = pkg.CubicSpline.new() if strategy == "cubic" else pkg.Linear.new()
            it = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(x, device=dev)).strategy(strat).build()
            q = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])).clamp(float(x[0]), float(x[-1]))
            out = torch.empty((Q, L), dtype=tdt, device=dev)
            for _ in range(2):
                it.strategy.interp_array_into(it, q, out, async_launch=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                it.strategy.interp_array_into(it, q, out, async_launch=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 4 * 1e3
            it.strategy.finish()
            row.append(f"{Q * L * el / ms / 1e9:5.2f}")
            it.strategy.release()
            del out, q, it, yd
            torch.cuda.empty_cache()
        print(f"{strategy} {np.dtype(dt).name} n={x.size:7d} | out TB/s at L = 1, 5, 8, 32, 128, 512, 4096: " + "  ".join(row), flush=True)
