"""One reference bench shape at large Q on device buffers, a few calls -- the unit tools/pmc_kernel.sh profiles.
    python tools/small_shape_once.py 1d:100:1:f64[:linear]      (knots : lanes : dtype)   1e8 / 5e7 queries
    python tools/small_shape_once.py 2d:100:100:5:f32           (nx : ny : channels : dtype)
Prints one JSON line (wall time per call, Gqueries/s, output TB/s)."""
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
spec = sys.argv[1].split(":")
reps = int(os.environ.get("REPS", "5"))
# FRESH=1: interp_array semantics (the output is the call's own: NDI_EVAL_FRESH_OUTPUT, no range pre-pass); 0: interp_array_into
fresh = os.environ.get("FRESH", "0") == "1"
rng = np.random.default_rng(0)
if spec[0] == "1d":
    n, L = int(spec[1]), int(spec[2])
    dt, tdt = (np.float64, torch.float64) if spec[3] == "f64" else (np.float32, torch.float32)
    linear = len(spec) > 4 and spec[4] == "linear"
    Q = int(os.environ.get("Q", 100_000_000 if L <= 2 else 50_000_000))
    x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n]
    y = rng.uniform(0, 1, (x.size, L)).astype(dt)
    strat = pkg.Linear.new() if linear else pkg.CubicSpline.new()
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(strat).build()
    q = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])).clamp(float(x[0]), float(x[-1]))
    out = torch.empty((Q, L), dtype=tdt, device=dev)
    call = lambda: it.strategy.interp_array_into(it, q, out, async_launch=True, fresh=fresh)
    C = L
else:
    nx, ny, C = int(spec[1]), int(spec[2]), int(spec[3])
    dt, tdt = (np.float64, torch.float64) if spec[4] == "f64" else (np.float32, torch.float32)
    Q = int(os.environ.get("Q", 50_000_000 if C <= 2 else 20_000_000))
    x = np.cumsum(rng.uniform(0.5, 1.5, nx)).astype(dt)
    y = np.cumsum(rng.uniform(0.5, 1.5, ny)).astype(dt)
    grid = torch.rand((nx, ny, C), dtype=tdt, device=dev)
    it = pkg.Interp2DBuilder.new(grid).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    qx = torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])
    qy = torch.rand(Q, dtype=tdt, device=dev) * float(y[-1] - y[0]) * 0.999 + float(y[0])
    qmode = os.environ.get("QMODE", "")       # experiments: const = every query in one cell (L1 hits), sorted = by x
    if qmode == "const":
        qx = qx * 0 + float(x[nx // 2] + 0.3 * (x[nx // 2 + 1] - x[nx // 2])); qy = qy * 0 + float(y[ny // 3] + 0.6 * (y[ny // 3 + 1] - y[ny // 3]))
    elif qmode == "sorted":
        qx, _ = torch.sort(qx)
    out = torch.empty((Q, C), dtype=tdt, device=dev)
    call = lambda: it.strategy.interp_array_into(it, qx, qy, out, async_launch=True, fresh=fresh)
call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    call()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
it.strategy.finish()
print(json.dumps({"shape": sys.argv[1], "fresh_output": fresh, "queries": Q, "ms": round(ms, 4), "Gqueries_s": round(Q / ms / 1e6, 1),
                  "out_TBps": round(Q * C * np.dtype(dt).itemsize / ms / 1e9, 3),
                  "io_TBps": round(Q * (C + (1 if spec[0] == "1d" else 2)) * np.dtype(dt).itemsize / ms / 1e9, 3)}), flush=True)
