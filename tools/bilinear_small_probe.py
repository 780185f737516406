"""2-D Bilinear on the reference's own bench shapes (100 x 100 scalar grid, 100 x 100 x 5) and a few larger scalar /
few-channel grids at large Q, device buffers: what the gather order gives where tiles are not eligible."""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    for nx, ny, C, Q in ((100, 100, 1, 50_000_000), (100, 100, 5, 20_000_000), (1000, 1000, 1, 50_000_000),
                         (4096, 4096, 1, 50_000_000), (1000, 1000, 4, 20_000_000), (1000, 1000, 16, 20_000_000)):
        x = np.cumsum(rng.uniform(0.5, 1.5, nx)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, ny)).astype(dt)
        grid = torch.rand((nx, ny, C) if C > 1 else (nx, ny), dtype=tdt, device=dev)
        it = pkg.Interp2DBuilder.new(grid).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
        qx = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0]))
        qy = (torch.rand(Q, dtype=tdt, device=dev) * float(y[-1] - y[0]) * 0.999 + float(y[0]))
        out = torch.empty((Q, C) if C > 1 else (Q,), dtype=tdt, device=dev)
        for _ in range(2):
            it.strategy.interp_array_into(it, qx, qy, out.view(Q, -1), async_launch=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            it.strategy.interp_array_into(it, qx, qy, out.view(Q, -1), async_launch=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        it.strategy.finish()
        print(json.dumps({"dtype": np.dtype(dt).name, "grid": [nx, ny, C], "queries": Q, "ms": round(ms, 4),
                          "Gqueries_s": round(Q / ms / 1e6, 1), "out_TBps": round(Q * C * np.dtype(dt).itemsize / ms / 1e9, 3)}), flush=True)
        del out, qx, qy, it, grid
