"""Spline build time by boundary kind at the C2 table size."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n = L = 4096
x = torch.as_tensor(np.unique(rng.uniform(0, 1, 2 * n))[:n], device=dev)
y = torch.rand((n, L), dtype=torch.float64, device=dev); y[-1] = y[0]
for name, bc in (("not-a-knot", pkg.BoundaryCondition.NotAKnot), ("periodic", pkg.BoundaryCondition.Periodic)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        it = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new().boundary(bc)).build()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        it.strategy.release()
    print(f"{name:12s} build (copy + plan + kernels) {dt*1e3:7.2f} ms")
