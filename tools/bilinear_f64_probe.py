"""2-D gather order on f64 / f32 grids of several channel counts: IEEE divisions vs the shared-divisor division
(NDI_BILINEAR_SDIV=0|1), gather path forced.  1 GiB grids, 1e7 queries."""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    el = np.dtype(dt).itemsize
    for C in (16, 32, 64, 128):
        nx = ny = int(np.sqrt((1 << 30) / (C * el)))
        Q = 10_000_000 if C * el <= 512 else 5_000_000
        x = np.cumsum(rng.uniform(0.5, 1.5, nx)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, ny)).astype(dt)
        grid = torch.rand((nx, ny, C), dtype=tdt, device=dev)
        it = pkg.Interp2DBuilder.new(grid).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
        it.strategy.path = pkg.PATH_GATHER
        qx = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0]))
        qy = (torch.rand(Q, dtype=tdt, device=dev) * float(y[-1] - y[0]) * 0.999 + float(y[0]))
        out = torch.empty((Q, C), dtype=tdt, device=dev)
        for _ in range(2):
            it.strategy.interp_array_into(it, qx, qy, out, async_launch=True)
        torch.cuda.synchronize()
        pkg.profile_enable(True); pkg.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(5):
            it.strategy.interp_array_into(it, qx, qy, out, async_launch=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        it.strategy.finish()
        prof = pkg.profile_read(reset=True); pkg.profile_enable(False)
        print(json.dumps({"dtype": np.dtype(dt).name, "grid": [nx, ny, C], "queries": Q, "ms": round(ms, 4), "eval_ms": round(prof["eval_ms"] / 5, 4),
                          "out_TBps": round(Q * C * el / ms / 1e9, 3), "alg_TBps": round(Q * C * el * 5 / ms / 1e9, 2),
                          "sdiv": os.environ.get("NDI_BILINEAR_SDIV", "default")}), flush=True)
        del out, qx, qy, it, grid
        torch.cuda.empty_cache()
