"""Round 6: is the 10-25 % spread of the C2 legs between output buffers (rounds 1-5: "placement") a property of the buffer, or
of WHEN it is measured (clock / power-state ramp under sustained load)?  BASELINE configs[1] (4096 knots x 4096 f64 lanes,
1e6 queries), three 32.8 GB buffers A, B, C from torch.empty: kernel time (library HIP events) of 3 launches into each
  pass 1  A, B, C   right after the build (one warm-up launch each)
  pass 2  C, B, A   immediately afterwards
  pass 3  A, B, C   after SUSTAIN_MS of back-to-back launches
Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

pkg = g.load_package()
dev = torch.device("cuda:0")
n = lanes = 4096
nq = 1_000_000
x, y, q = bench.synth_c2(n, lanes, nq, 0)
it = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
qd = torch.as_tensor(q, device=dev)
bufs = {k: torch.empty((nq, lanes), dtype=torch.float64, device=dev) for k in "ABC"}


def kernel_ms(out, reps=3, warm=1):
    for _ in range(warm):
        it.strategy.interp_array_into(it, qd, out, async_launch=True)
    it.strategy.finish()
    pkg.profile_enable(True); pkg.profile_read(reset=True)
    for _ in range(reps):
        it.strategy.interp_array_into(it, qd, out, async_launch=True)
    it.strategy.finish()
    p = pkg.profile_read(reset=True); pkg.profile_enable(False)
    return round(p["eval_ms"] / max(1, p["eval_launches"]), 4)


res = {"pass1_ABC": [kernel_ms(bufs[k]) for k in "ABC"], "pass2_CBA": [kernel_ms(bufs[k]) for k in "CBA"]}
sustain = float(os.environ.get("SUSTAIN_MS", "500"))
t_end = time.perf_counter() + sustain * 1e-3
while time.perf_counter() < t_end:
    it.strategy.interp_array_into(it, qd, bufs["A"], async_launch=True)
    it.strategy.finish()
res["pass3_ABC_after_sustained_ms"] = sustain
res["pass3_ABC"] = [kernel_ms(bufs[k], warm=0) for k in "ABC"]
res["pass4_CBA"] = [kernel_ms(bufs[k], warm=0) for k in "CBA"]
time.sleep(2.0)
res["pass5_ABC_after_2s_idle"] = [kernel_ms(bufs[k], warm=0) for k in "ABC"]
print(json.dumps(res), flush=True)
