"""Crossover of the two formulations in queries per interval (C2 tables: 4096 knots x 4096 lanes f64)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(42)
n = L = 4096
x = np.unique(rng.uniform(0, 1, 2 * n))[:n]
yd = torch.rand((n, L), dtype=torch.float64, device=dev)
interp = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
for Q in (2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144):
    qd = torch.as_tensor(rng.uniform(x[0], x[-1], Q), device=dev)
    out = torch.empty((Q, L), dtype=torch.float64, device=dev)
    line = f"Q={Q:7d} ({Q/(n-1):6.2f} per interval)"
    for name, path in (("gather", pkg.PATH_GATHER), ("bucketed", pkg.PATH_BUCKETED)):
        interp.strategy.path = path
        for _ in range(3):
            interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 20
        for _ in range(reps):
            interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
        torch.cuda.synchronize()
        line += f" | {name} {(time.perf_counter()-t0)/reps*1e3:8.3f} ms"
    print(line, flush=True)
