"""C3 (2048^2 x 64 f32 grid, 1e7 queries, tile-grouped AUTO): does the tile kernel's time depend on WHICH output buffer it
writes (2.56 GB of scattered 128-byte half rows), as the long-row kernels' does (profiles/r06_tuning.md section 2)?
Four caller buffers (torch.empty, earlier ones kept alive so that each lands elsewhere) and three library-owned ones
(ndi_output_alloc, probe-and-retry), kernel time from the library's HIP events, one JSON line each."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
dev = torch.device("cuda:0")
nx, C, nq = 2048, 64, 10_000_000
rng = np.random.default_rng(42)
x = np.unique(rng.uniform(0, 1, 2 * nx).astype(np.float32))[:nx]
y = np.unique(rng.uniform(0, 1, 2 * nx).astype(np.float32))[:nx]
grid = torch.rand((nx, nx, C), dtype=torch.float32, device=dev)
it = pkg.Interp2DBuilder.new(grid).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
qx = torch.as_tensor(np.random.default_rng(123).uniform(x[0], x[-1], nq).astype(np.float32), device=dev)
qy = torch.as_tensor(np.random.default_rng(96).uniform(y[0], y[-1], nq).astype(np.float32), device=dev)


def measure(out, steps=10):
    step = lambda: it.strategy.interp_array_into(it, qx, qy, out, async_launch=True)
    for _ in range(3):
        step()
    it.strategy.finish()
    pkg.profile_enable(True); pkg.profile_read(reset=True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    it.strategy.finish()
    p = pkg.profile_read(reset=True); pkg.profile_enable(False)
    return round(p["eval_ms"] / max(1, p["eval_launches"]), 4), round(p["group_ms"] / steps, 4), round(p["locate_ms"] / max(1, p["locate_launches"]), 4)


keep = []
for i in range(4):
    out = torch.empty((nq, C), dtype=torch.float32, device=dev)
    keep.append(out)
    k, gms, l = measure(out)
    print(json.dumps({"buffer": f"torch.empty #{i}", "ptr": hex(out.data_ptr()), "kernel_ms": k, "group_ms": gms, "locate_ms": l}), flush=True)
for i in range(3):
    out = pkg.output_empty((nq, C), np.float32, 0)
    k, gms, l = measure(out)
    print(json.dumps({"buffer": f"library-owned #{i}", "ptr": hex(out.data_ptr()), "info": out.ndi_output_info, "kernel_ms": k, "group_ms": gms, "locate_ms": l}), flush=True)
    del out
    pkg.output_trim()
