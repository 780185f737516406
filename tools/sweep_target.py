#!/usr/bin/env python3
"""tools/sweep_target.py -- one process, one set of allocations: eval_bucketed_kernel time per 1e6 queries of the
Target workload for chunk sizes x chunks-per-workgroup runs (NDI_BUCKETED_RUN), into a striped ring of 164 GB."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = bench.load_package()
dev = torch.device("cuda:0")
n = L = 4096
x, y, _ = bench.synth_c2(n, L, 1, 0)
interp = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
    .strategy(pkg.CubicSpline.new()).build()
q = bench.synth_target_queries(x, 10_000_000, 1_000_000, 0)
qd = torch.as_tensor(q, device=dev)
base = torch.empty((5_000_000, L), dtype=torch.float64, device=dev)      # 164 GB, carved into striped slots
out = []
for chunk, slots in ((1_000_000, 5), (2_500_000, 2), (5_000_000, 1)):
    ring = [base.view(chunk, slots, L)[:, s, :] for s in range(slots)]
    for run in (1, 4, 8, 16, 32):
        os.environ["NDI_BUCKETED_RUN"] = str(run)
        interp.interp_array_ring(qd, chunk, None, slots=ring)
        pkg.profile_enable(True); pkg.profile_read(reset=True)
        for _ in range(2):
            interp.interp_array_ring(qd, chunk, None, slots=ring)
        p = pkg.profile_read(reset=True); pkg.profile_enable(False)
        ms = p["eval_ms"] / p["eval_launches"] / (chunk / 1e6)
        out.append({"chunk": chunk, "slots": slots, "run": run, "eval_ms_per_1e6_queries": round(ms, 4),
                    "locate_group_ms_per_1e6": round((p["locate_ms"] + p["group_ms"]) / 2 / 10, 4)})
        print(json.dumps(out[-1]), flush=True)
