"""Sweep of row lengths: effective output rate of both formulations and of what AUTO picks (finds weak launch
shapes).  1024 random knots, CubicSpline, 4 GB of device-resident output per call; "out" = output bytes / wall time of
the whole interp_array_into call, "alg" = the 40 B/point (5 x sizeof T) gather model of SURVEY 8(d)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n = 1024
x = np.unique(rng.uniform(0, 1, 2 * n))[:n]
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    for L in (8, 32, 64, 128, 256, 512, 1024, 2048, 4096, 16384, 65536):
        el = np.dtype(dt).itemsize
        Q = int(min(4e9 // (L * el), 2**31 - 1))
        yd = torch.rand((n, L), dtype=tdt, device=dev)
        xd = torch.as_tensor(np.unique(x.astype(dt)), device=dev)
        yd = yd[: xd.numel()]
        interp = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
        qd = (torch.rand(Q, dtype=tdt, device=dev) * (xd[-1] - xd[0]) * 0.999 + xd[0]).clamp(xd[0], xd[-1])
        out = torch.empty((Q, L), dtype=tdt, device=dev)
        line = f"{np.dtype(dt).name} L={L:6d} Q={Q:10d}"
        for name, path in (("gather", pkg.PATH_GATHER), ("bucketed", pkg.PATH_BUCKETED), ("auto", pkg.PATH_AUTO)):
            interp.strategy.path = path
            interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t0) / 3
            interp.strategy.finish()
            line += f" | {name} {dtm*1e3:8.3f} ms {Q*L/dtm/1e9:7.1f} Gpt/s out {Q*L*el/dtm/1e12:5.2f} TB/s alg {Q*L*el*5/dtm/1e12:5.2f}"
        print(line, flush=True)
        del out, qd, interp, yd
        torch.cuda.empty_cache()
