"""Short trailing axes (the reference's own bench shapes: scalar data / 5 lanes): device-resident rate."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n in (100, 4096):
    x = np.unique(rng.uniform(0, 1, 2 * n))[:n]
    for L in (1, 2, 5, 8, 16):
        for kind in ("linear", "cubic"):
            Q = 20_000_000
            yd = torch.rand((n, L) if L > 1 else (n,), dtype=torch.float64, device=dev)
            b = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(x, device=dev))
            interp = (b.strategy(pkg.CubicSpline.new()) if kind == "cubic" else b).build()
            qd = torch.rand(Q, dtype=torch.float64, device=dev) * (x[-1] - x[0]) * 0.999 + x[0]
            out = torch.empty((Q, L), dtype=torch.float64, device=dev)
            pkg.profile_enable(True); pkg.profile_read(True)
            for _ in range(2):
                interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
            torch.cuda.synchronize()
            pkg.profile_read(True)
            t0 = time.perf_counter()
            for _ in range(5):
                interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            p = pkg.profile_read(True); pkg.profile_enable(False)
            print(f"n={n:5d} L={L:3d} {kind:6s} Q={Q}  {dt*1e3:7.3f} ms/step  {Q/dt/1e9:6.2f} Gquery/s  {Q*L/dt/1e9:7.2f} Gpt/s  (locate {p['locate_ms']/5:.3f} eval {p['eval_ms']/5:.3f} ms)", flush=True)
