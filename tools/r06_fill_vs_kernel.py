"""Round 6: does the sequential zero-fill rate of an output buffer predict the evaluation kernel's time into it?  K buffers of
the C2 shape (f64: 32.8 GB, f32: 16.4 GB; all held at once), per buffer: ms of torch's zero_() (median of 3) and the kernel
time of 3 launches.  One JSON line per element type."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = bench.load_package()
dev = torch.device("cuda:0")
n = lanes = 4096
nq = 1_000_000
for dt, tdt, K in ((np.float32, torch.float32, 10), (np.float64, torch.float64, 6)):
    x, y, q = bench.synth_c2(n, lanes, nq, 0)
    x = np.unique(x.astype(dt)); y = y[:x.size].astype(dt); q = np.clip(q.astype(dt), x[0], x[-1])
    it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
    qd = torch.as_tensor(q, device=dev)
    bufs = [torch.empty((nq, lanes), dtype=tdt, device=dev) for _ in range(K)]
    rows = []
    for o in bufs:
        fills = []
        o.zero_(); torch.cuda.synchronize()
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o.zero_(); e1.record(); e1.synchronize()
            fills.append(e0.elapsed_time(e1))
        it.strategy.interp_array_into(it, qd, o, async_launch=True); it.strategy.finish()
        pkg.profile_enable(True); pkg.profile_read(reset=True)
        for _ in range(3):
            it.strategy.interp_array_into(it, qd, o, async_launch=True)
        it.strategy.finish()
        p = pkg.profile_read(reset=True); pkg.profile_enable(False)
        rows.append([round(float(np.median(fills)), 3), round(p["eval_ms"] / max(1, p["eval_launches"]), 4)])
    f = np.array(rows)
    print(json.dumps({"dtype": np.dtype(dt).name, "fill_ms_and_kernel_ms": rows,
                      "corr": round(float(np.corrcoef(f[:, 0], f[:, 1])[0, 1]), 3)}), flush=True)
    it.strategy.release()
    del bufs, qd, it
    torch.cuda.empty_cache()
