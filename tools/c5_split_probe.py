"""C5 share / C3: one launch of search + evaluation over the whole batch vs the same batch produced in K chunks
through the ring machinery into the caller's own buffer (search of chunk k+1 overlaps the evaluation of chunk k)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = bench.load_package()
dev = torch.device("cuda:0")
for name, nx, C, nq in (("c5", 8192, 16, 12_500_000), ("c3", 2048, 64, 10_000_000)):
    rng = np.random.default_rng(42)
    x = np.unique(rng.uniform(0, 1, 2 * nx).astype(np.float32))[:nx]
    y = np.unique(rng.uniform(0, 1, 2 * nx).astype(np.float32))[:nx]
    g = torch.rand((nx, nx, C), dtype=torch.float32, device=dev)
    it = pkg.Interp2DBuilder.new(g).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    it.strategy.path = pkg.PATH_GATHER
    del g
    qx = torch.as_tensor(np.random.default_rng(123).uniform(x[0], x[-1], nq).astype(np.float32), device=dev)
    qy = torch.as_tensor(np.random.default_rng(96).uniform(y[0], y[-1], nq).astype(np.float32), device=dev)
    out = torch.empty((nq, C), dtype=torch.float32, device=dev)
    def t(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    one = t(lambda: (it.strategy.interp_array_into(it, qx, qy, out, async_launch=True), it.strategy.finish()))
    ref = out.clone()
    res = {"workload": name, "one_launch_ms": round(one, 4)}
    for K in (2, 4, 8):
        chunk = (nq + K - 1) // K
        slots = [out[k * chunk:(k + 1) * chunk] for k in range(K)]
        if slots[-1].shape[0] != chunk:      # ring slots must hold a full chunk: pad by overlapping the tail
            continue
        ms = t(lambda: it.interp_array_ring(qx, qy, chunk, None, slots=slots))
        res[f"ring_{K}_chunks_ms"] = round(ms, 4)
        assert torch.equal(out, ref)
    print(json.dumps(res))
    it.strategy.release(); del it, qx, qy, out, ref
    torch.cuda.empty_cache()
