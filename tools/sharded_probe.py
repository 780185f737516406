"""One process, N replica handles, ONE call (ndi_interp1d_eval_sharded / _eval_ring_sharded) beside the single-handle
calls on the same total batch.  On a 1-GPU box all replicas share device 0, so the totals measure the overhead of the
sharded machinery (worker threads, per-shard range pre-pass, host barrier), not a speed-up; on a multi-GPU node pass
--devices 0,1,... to place one replica per device.  One JSON object per line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="0,0")
ap.add_argument("--queries", type=int, default=2_000_000)
ap.add_argument("--chunk", type=int, default=500_000)
args = ap.parse_args()
devs = [int(d) for d in args.devices.split(",")]
pkg = bench.load_package()
n = lanes = 4096
x, y, _ = bench.synth_c2(n, lanes, 1, 0)
q = np.random.default_rng(123).uniform(x[0], x[-1], args.queries)
reps = [pkg.Interp1DBuilder.new(torch.as_tensor(y, device=f"cuda:{d}")).x(torch.as_tensor(x, device=f"cuda:{d}"))
        .strategy(pkg.CubicSpline.new().device(d)).build() for d in devs]
N = len(reps)
bounds = [pkg.sharding.shard_bounds(args.queries, i, N) for i in range(N)]
blocks = [torch.as_tensor(q[lo:hi], device=f"cuda:{d}") for (lo, hi), d in zip(bounds, devs)]
qd0 = torch.as_tensor(q, device=f"cuda:{devs[0]}")


def timed(fn, reps_=5):
    fn()
    for d in set(devs):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for _ in range(reps_):
        fn()
    for d in set(devs):
        torch.cuda.synchronize(d)
    return (time.perf_counter() - t0) / reps_ * 1e3


pts = args.queries * lanes
# (1) ring evaluation: single handle, whole batch  vs  sharded ring, one block per replica
ms1 = timed(lambda: reps[0].interp_array_ring(qd0, args.chunk, None, n_slots=2))
msN = timed(lambda: pkg.sharding.interp_array_ring_sharded(reps, blocks, chunk_queries=args.chunk, consumer=None, n_slots=2))
print(json.dumps({"what": "ring", "devices": devs, "queries": args.queries, "chunk": args.chunk,
                  "single_handle_ms": round(ms1, 3), "sharded_ms": round(msN, 3),
                  "single_Gpoints_s": round(pts / ms1 / 1e6, 1), "sharded_Gpoints_s": round(pts / msN / 1e6, 1)}))
for r in reps:
    r.strategy.trim()
# (2) host arrays in and out (what a Rust caller with host ndarrays does): PCIe-bound
qh = q[:200_000]
out = np.empty((qh.size, lanes))
ms1 = timed(lambda: reps[0].interp_array_into(qh, out), 3)
msN = timed(lambda: pkg.sharding.interp_array_sharded(reps, qh, out=out), 3)
print(json.dumps({"what": "host arrays in and out", "devices": devs, "queries": qh.size,
                  "single_handle_ms": round(ms1, 2), "sharded_ms": round(msN, 2),
                  "single_GB_s_out": round(out.nbytes / ms1 / 1e6, 1), "sharded_GB_s_out": round(out.nbytes / msN / 1e6, 1)}))
