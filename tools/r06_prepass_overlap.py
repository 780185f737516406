"""Short rows, interp_array_into semantics (range pre-pass + evaluation): one call against the same batch as K chunks
through ndi_interp1d_eval_ring with the chunks' slots cut out of the caller's output (chunk k + 1's pre-pass on the side
stream while chunk k is evaluated).  ms per 4 GB of output."""
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
rng = np.random.default_rng(0)
for dt, tdt, L in ((np.float64, torch.float64, 8), (np.float32, torch.float32, 8), (np.float64, torch.float64, 32), (np.float64, torch.float64, 5)):
    n = 1024
    Q = int(4e9 // (L * np.dtype(dt).itemsize))
    x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n]
    y = rng.uniform(0, 1, (x.size, L)).astype(dt)
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
    q = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])).clamp(float(x[0]), float(x[-1]))
    out = torch.empty((Q, L), dtype=tdt, device=dev)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    one = timed(lambda: (it.strategy.interp_array_into(it, q, out, async_launch=True), it.strategy.finish()))
    fresh = timed(lambda: (it.strategy.interp_array_into(it, q, out, async_launch=True, fresh=True), it.strategy.finish()))
    row = {"dtype": np.dtype(dt).name, "lanes": L, "queries": Q, "one_call_ms": round(one, 4), "fresh_ms": round(fresh, 4)}
    for K in (4, 8, 16):
        chunk = (Q + K - 1) // K
        slots = [out[k * chunk:(k + 1) * chunk] for k in range(K)]
        if slots[-1].shape[0] < chunk:        # (the ring wants slots of chunk rows: keep the test simple)
            continue
        row[f"ring_{K}_chunks_ms"] = round(timed(lambda: it.strategy.interp_array_ring(q, chunk, None, slots=slots)), 4)
    print(json.dumps(row), flush=True)
    it.strategy.release()
    del it, q, out
    torch.cuda.empty_cache()
