"""Round 6 (AUTO guard finding): tile-grouped vs gather order on grids BELOW the 256 MiB AUTO limit, 64-channel f32 rows.
Prints one JSON line per (grid, queries per cell): ms of both forms."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
dev = torch.device("cuda:0")


def t(call, fin):
    for _ in range(2):
        call()
    fin()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    fin()
    return round(float(np.median(ts)), 4)


for C, dt, tdt in ((64, np.float32, torch.float32), (16, np.float32, torch.float32), (32, np.float64, torch.float64)):
    for n in (200, 300, 500, 700, 1000, 1400):
        rng = np.random.default_rng(n)
        x = np.cumsum(rng.uniform(0.5, 1.5, n)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, n)).astype(dt)
        it = pkg.Interp2DBuilder.new(torch.rand((n, n, C), dtype=tdt, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
        for qpc in (0.5, 1, 2, 4, 8, 16):
            Q = int(qpc * (n - 1) ** 2)
            if Q < 100_000 or Q > 30_000_000:
                continue
            qx = torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])
            qy = torch.rand(Q, dtype=tdt, device=dev) * float(y[-1] - y[0]) * 0.999 + float(y[0])
            out = torch.empty((Q, C), dtype=tdt, device=dev)
            call = lambda: it.strategy.interp_array_into(it, qx, qy, out, async_launch=True)
            r = {"grid": n, "C": C, "dtype": np.dtype(dt).name, "grid_MB": round(n * n * C * np.dtype(dt).itemsize / 1e6, 1), "qpc": qpc, "Q": Q}
            for name, path in (("gather", pkg.PATH_GATHER), ("tiles", pkg.PATH_BUCKETED), ("auto", pkg.PATH_AUTO)):
                it.strategy.path = path
                r[name] = t(call, it.strategy.finish)
            print(json.dumps(r), flush=True)
            del qx, qy, out
        it.strategy.release()
