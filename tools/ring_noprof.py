import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = bench.load_package()
dev = torch.device("cuda:0")
x, y, _ = bench.synth_c2(4096, 4096, 1, 0)
q = bench.synth_target_queries(x, 10_000_000, 2_500_000, 0)
interp = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)).strategy(pkg.CubicSpline.new()).build()
qd = torch.as_tensor(q, device=dev)
ring = pkg.striped_ring(2_500_000, 4096, 2, np.float64, 0)
def step(): interp.interp_array_ring(qd, 2_500_000, None, slots=ring)
for prof in (False, True, False, True):
    pkg.profile_enable(prof); pkg.profile_read(reset=True)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20 * 1e3
    p = pkg.profile_read(reset=True)
    print(json.dumps({"profiling": prof, "ms_per_step": round(el, 4), "eval_ms_per_step": round(p["eval_ms"] / 20, 4)}))
