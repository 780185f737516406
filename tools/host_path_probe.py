"""PCIe-inclusive rate of the host-buffer mode (what a Rust caller with host ndarrays sees)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
rng = np.random.default_rng(0)
n, L = 4096, 4096
x = np.unique(rng.uniform(0, 1, 2 * n))[:n]; y = rng.uniform(0, 1, (n, L))
interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
for Q in (10_000, 100_000, 400_000):
    q = rng.uniform(x[0], x[-1], Q)
    out = np.zeros((Q, L))
    interp.interp_array_into(q, out)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        interp.interp_array_into(q, out)
    dt = (time.perf_counter() - t0) / reps
    print(f"host->host  Q={Q:7d}  {dt*1e3:9.2f} ms  {Q*L/dt/1e9:7.2f} Gpoints/s  out {Q*L*8/dt/1e9:6.2f} GB/s")
