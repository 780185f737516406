"""Short rows on axes with many knots (half the LDS and more): the query-order kernel with one workgroup per CU against
the two-kernel flat form.   NDI_FUSED_LONG_AXES=0|1 python tools/long_axis_probe.py"""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for dt, tdt, n, L in ((np.float64, torch.float64, 12000, 8), (np.float64, torch.float64, 16384, 32), (np.float64, torch.float64, 16384, 5),
                      (np.float32, torch.float32, 30000, 16), (np.float32, torch.float32, 20000, 64), (np.float64, torch.float64, 18000, 128)):
    el = np.dtype(dt).itemsize
    x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n]
    Q = int(2e9 // (L * el))
    yd = torch.rand((x.size, L), dtype=tdt, device=dev)
    for strat_name in ("cubic", "linear"):
        strat = pkg.CubicSpline.new() if strat_name == "cubic" else pkg.Linear.new()
        it = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(x, device=dev)).strategy(strat).build()
        q = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])).clamp(float(x[0]), float(x[-1]))
        out = torch.empty((Q, L), dtype=tdt, device=dev)
        for _ in range(2):
            it.strategy.interp_array_into(it, q, out, async_launch=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            it.strategy.interp_array_into(it, q, out, async_launch=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        it.strategy.finish()
        print(json.dumps({"dtype": np.dtype(dt).name, "knots": int(x.size), "lanes": L, "strategy": strat_name, "queries": Q,
                          "ms": round(ms, 4), "out_TBps": round(Q * L * el / ms / 1e9, 3),
                          "long_axes": os.environ.get("NDI_FUSED_LONG_AXES", "1")}), flush=True)
        del out, q, it
