"""Round 6: CubicSpline::build at BASELINE configs[1]'s shape (4096 knots x 4096 f64 lanes) -- ms inside ndi_interp1d_create
(device-resident arrays) with the wide kernel (AUTO) and with the serial kernels (NDI_SPLINE_WIDE=0), a few widths."""
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
os.environ["NDI_TUNE_LIVE"] = "1"
for n, L, dt in ((4096, 4096, torch.float64), (4096, 4096, torch.float32), (4096, 1024, torch.float64), (1024, 16384, torch.float64),
                 (512, 65536, torch.float32)):
    x = torch.cumsum(torch.rand(n, dtype=dt, device=dev) + 0.5, 0)
    y = torch.rand((n, L), dtype=dt, device=dev)
    r = {"n": n, "lanes": L, "dtype": str(dt)[6:]}
    for name, env in (("wide", None), ("serial", "0")):
        if env is None:
            os.environ.pop("NDI_SPLINE_WIDE", None)
        else:
            os.environ["NDI_SPLINE_WIDE"] = env
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            it = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            it.strategy.release()
        r[name + "_create_ms"] = round(float(np.median(ts)), 3)
    print(json.dumps(r), flush=True)
