"""Where a CubicSpline build's milliseconds go: NDI_BUILD_TIMING=1 makes ndi_interp1d_create print wall-clock
milestones on stderr; this prints the Python mirror's total next to them.
    NDI_BUILD_TIMING=1 python tools/build_timing.py [n lanes]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
rng = np.random.default_rng(0)
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(100, 5), (4096, 8), (1_000_000, 1)]
for n, L in shapes:
    x = np.cumsum(rng.uniform(0.5, 1.5, n)); y = rng.uniform(0, 1, (n, L))
    xd, yd = torch.as_tensor(x, device="cuda:0"), torch.as_tensor(y, device="cuda:0")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        it = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
        torch.cuda.synchronize()
        print(f"n={n} lanes={L}: python build() total {(time.perf_counter() - t0) * 1e3:.3f} ms", file=sys.stderr)
        it.strategy.release()
