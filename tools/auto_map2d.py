"""What AUTO delivers across the 2-D shape space (grid side x channels), up to 1 GB of output per call (at most 3e7
queries), device buffers: Gqueries/s and algorithmic TB/s (5 x sizeof(T) per point) -- a map for spotting weak regions."""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for dt, tdt in ((np.float32, torch.float32), (np.float64, torch.float64)):
    el = np.dtype(dt).itemsize
    for side in (100, 1000, 2048, 4096, 8192):
        x = np.cumsum(rng.uniform(0.5, 1.5, side)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, side)).astype(dt)
        row = []
        for C in (1, 4, 5, 16, 64, 256):
            if side * side * C * el > 9e9:
                row.append("      -      "); continue
            Q = int(min(1e9 // (C * el), 3e7))
            grid = torch.rand((side, side, C), dtype=tdt, device=dev)
            it = pkg.Interp2DBuilder.new(grid).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
            qx = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0]))
            qy = (torch.rand(Q, dtype=tdt, device=dev) * float(y[-1] - y[0]) * 0.999 + float(y[0]))
            out = torch.empty((Q, C), dtype=tdt, device=dev)
            for _ in range(2):
                it.strategy.interp_array_into(it, qx, qy, out, async_launch=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                it.strategy.interp_array_into(it, qx, qy, out, async_launch=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 4 * 1e3
            it.strategy.finish()
            row.append(f"{Q / ms / 1e6:5.1f}/{Q * C * el * 5 / ms / 1e9:5.2f}")
            it.strategy.release()
            del out, qx, qy, it, grid
            torch.cuda.empty_cache()
        print(f"{np.dtype(dt).name} {side:5d}^2 | Gq/s / alg TB/s at C = 1, 4, 5, 16, 64, 256: " + "  ".join(row), flush=True)
