"""Round 6: is the buffer-dependent rate of the long-row kernel a TLB / page-fragment effect of its scattered write pattern?
(torch's sequential zero_() runs at the same 6.76 TB/s into every buffer, tools/r06_fill_vs_kernel.py.)  C2 shape, f64, four
32.8 GB buffers: the 1e6-query batch evaluated as K launches over K contiguous blocks of the query array (each launch then
writes a contiguous window of 32.8 / K GB), K = 1, 2, 4, 8, 16: total kernel ms per buffer."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = bench.load_package()
dev = torch.device("cuda:0")
n = lanes = 4096
nq = 1_000_000
x, y, q = bench.synth_c2(n, lanes, nq, 0)
it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
qd = torch.as_tensor(q, device=dev)
bufs = [torch.empty((nq, lanes), dtype=torch.float64, device=dev) for _ in range(4)]
for bi, o in enumerate(bufs):
    r = {"buffer": bi}
    for K in (1, 2, 4, 8, 16):
        step = nq // K

        def run():
            for k in range(K):
                it.strategy.interp_array_into(it, qd[k * step:(k + 1) * step], o[k * step:(k + 1) * step], async_launch=True)
        run(); it.strategy.finish()
        pkg.profile_enable(True); pkg.profile_read(reset=True)
        for _ in range(3):
            run()
        it.strategy.finish()
        p = pkg.profile_read(reset=True); pkg.profile_enable(False)
        r[f"K{K}_eval_ms"] = round(p["eval_ms"] / 3, 4)
        r[f"K{K}_all_ms"] = round((p["eval_ms"] + p["locate_ms"] + p["group_ms"]) / 3, 4)
    print(json.dumps(r), flush=True)
