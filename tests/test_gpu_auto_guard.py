"""AUTO guard (round 6): on a fixed grid of shapes -- the reference's own bench shapes (benches/bench_interp1d.rs:12-47, 82-122,
benches/bench_interp2d.rs:12-18, 87-92), every BASELINE config's tables at a reduced batch, the 8 .. 256-lane rows of the
bench's short_rows table, three 2-D grid sizes -- the formulation AUTO picks is timed against every form that can be FORCED
(ndi_path, and the kernel-variant knobs conftest's NDI_TUNE_LIVE makes the library re-read per call) and the test fails when
AUTO is more than 15 % (+ 20 us of launch jitter) slower than the best of them: a new kernel that makes AUTO pick a slower
form for a neighbouring shape fails HERE instead of in a profile nobody re-runs.  Timing: median of 5 asynchronous calls
between two device synchronisations, after 2 warm-up calls (lazy table copies are built in the warm-up).
The interpolators are the reference's: Interp1D::interp_array_into (interp1d/mod.rs:272-343), Interp2D (interp2d/mod.rs:215-307)."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SLACK, JITTER_MS = 1.15, 0.020
SHORT1D = [{"NDI_SHORT_MODE": "1"}, {"NDI_SHORT_MODE": "2", "NDI_LANES_KERNEL": "0"}, {"NDI_SHORT_MODE": "3"},
           {"NDI_LANES_KERNEL": "0"}, {"NDI_LANES_KERNEL": "1"}, {"NDI_FUSED_SORTED": "0"}, {"NDI_FUSED_SORTED": "1"}]
FORMS2D = [{"NDI_LANES2D_KERNEL": "0"}, {"NDI_SLOPES2D_KERNEL": "0", "NDI_LANES2D_KERNEL": "0"}, {"NDI_SLOPES2D_KERNEL": "1"},
           {"NDI_GROUP_TWO_LEVEL": "0"}, {"NDI_TILE_SPLIT": "0"}]

# (name, strategy, dtype, n, lanes, queries)
GRID_1D = [
    ("ref scalar 100", "cubic", np.float64, 100, 1, 20_000_000), ("ref scalar 100 f32", "cubic", np.float32, 100, 1, 20_000_000),
    ("ref scalar linear", "linear", np.float64, 100, 1, 20_000_000), ("ref (100, 5)", "cubic", np.float64, 100, 5, 10_000_000),
    ("ref (100, 5) f32", "cubic", np.float32, 100, 5, 10_000_000), ("scalar 1024", "cubic", np.float64, 1024, 1, 20_000_000),
    ("C1 tables", "linear", np.float64, 1024, 1, 1_000_000),
    ("rows x8", "cubic", np.float64, 1024, 8, 4_000_000), ("rows x8 f32", "cubic", np.float32, 1024, 8, 8_000_000),
    ("rows x16", "cubic", np.float64, 1024, 16, 2_000_000), ("rows x32", "cubic", np.float64, 1024, 32, 1_000_000),
    ("rows x32 f32", "cubic", np.float32, 1024, 32, 2_000_000), ("rows x64", "cubic", np.float64, 1024, 64, 1_000_000),
    ("rows x128", "cubic", np.float64, 1024, 128, 500_000), ("rows x128 f32", "cubic", np.float32, 1024, 128, 1_000_000),
    ("rows x256", "cubic", np.float64, 1024, 256, 250_000), ("rows x256 linear", "linear", np.float64, 1024, 256, 250_000),
    ("long axis x8", "cubic", np.float64, 12_000, 8, 2_000_000), ("long axis x128 f32", "cubic", np.float32, 8192, 128, 500_000),
    ("C2 tables", "cubic", np.float64, 4096, 4096, 100_000), ("C2 tables small batch", "cubic", np.float64, 4096, 4096, 8_000),
    ("C2 linear", "linear", np.float64, 4096, 4096, 100_000), ("C2 f32", "cubic", np.float32, 4096, 4096, 100_000),
    ("x1024 lanes", "cubic", np.float64, 512, 1024, 100_000), ("scalar 1e5 knots", "cubic", np.float64, 100_000, 1, 5_000_000),
]
# (name, dtype, nx, ny, channels, queries)
GRID_2D = [
    ("ref 100x100 scalar", np.float64, 100, 100, 1, 20_000_000), ("ref 100x100 scalar f32", np.float32, 100, 100, 1, 20_000_000),
    ("ref 100x100x5", np.float64, 100, 100, 5, 8_000_000), ("ref 100x100x5 f32", np.float32, 100, 100, 5, 8_000_000),
    ("100x100x3", np.float64, 100, 100, 3, 8_000_000), ("100x100x8", np.float64, 100, 100, 8, 4_000_000),
    ("100x100x16 f32", np.float32, 100, 100, 16, 4_000_000), ("1000x1000x4 f32", np.float32, 1000, 1000, 4, 8_000_000),
    ("1000x1000 scalar", np.float64, 1000, 1000, 1, 8_000_000), ("300x300x64 f32", np.float32, 300, 300, 64, 1_000_000),
    ("C3 tables", np.float32, 2048, 2048, 64, 10_000_000), ("C3 tables small batch", np.float32, 2048, 2048, 64, 1_000_000),
    ("C5 tables", np.float32, 8192, 8192, 16, 4_000_000),
]


class knobs:
    def __init__(self, env):
        self.env = env

    def __enter__(self):
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k in self.env:
            os.environ.pop(k, None)


def _time(call, finish, torch):
    for _ in range(2):
        call()
    finish()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    finish()
    return float(np.median(ts))


def _judge(name, t_auto, forced, again=None):
    """`again(key)` re-times one form (key None: AUTO).  A verdict that would fail is confirmed first: AUTO and the best forced
    form are timed three more times each, alternating, and their minima compared -- one noisy sample must not fail the suite."""
    best_name, best = min(forced.items(), key=lambda kv: kv[1])
    if t_auto > SLACK * best + JITTER_MS and again is not None:
        for _ in range(3):
            t_auto = min(t_auto, again(None))
            forced[best_name] = min(forced[best_name], again(best_name))
        best_name, best = min(forced.items(), key=lambda kv: kv[1])
    assert t_auto <= SLACK * best + JITTER_MS, \
        f"{name}: AUTO {t_auto:.4f} ms is {t_auto / best:.2f} x the best forced form ({best_name}: {best:.4f} ms); all: " + \
        ", ".join(f"{k}={v:.4f}" for k, v in sorted(forced.items(), key=lambda kv: kv[1]))


@pytest.mark.parametrize("name,strat,dt,n,L,Q", GRID_1D, ids=[g[0] for g in GRID_1D])
def test_auto_is_within_15_percent_of_the_best_forced_form_1d(pkg, name, strat, dt, n, L, Q):
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dt == np.float64 else torch.float32
    rng = np.random.default_rng(n + L)
    x = np.unique(rng.uniform(0, 1, 2 * n).astype(dt))[:n] if name != "C1 tables" else np.arange(n, dtype=dt)
    yd = torch.rand((x.size, L), dtype=tdt, device=dev)
    b = pkg.Interp1DBuilder.new(yd).x(torch.as_tensor(x, device=dev))
    it = (b.strategy(pkg.CubicSpline.new()) if strat == "cubic" else b).build()
    qd = (torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])).clamp(float(x[0]), float(x[-1]))
    out = torch.empty((Q, L), dtype=tdt, device=dev)
    call = lambda: it.strategy.interp_array_into(it, qd, out, async_launch=True)
    fin = it.strategy.finish
    it.strategy.path = pkg.PATH_AUTO
    t_auto = _time(call, fin, torch)
    forced = {}
    for pname, path in (("gather", pkg.PATH_GATHER), ("bucketed", pkg.PATH_BUCKETED)):
        it.strategy.path = path
        forced[pname] = _time(call, fin, torch)
    it.strategy.path = pkg.PATH_AUTO
    if L * np.dtype(dt).itemsize < 4096:              # short rows: the kernel-variant knobs apply
        for env in SHORT1D:
            with knobs(env):
                forced[" ".join(f"{k[4:]}={v}" for k, v in env.items())] = _time(call, fin, torch)
    t_auto = min(t_auto, _time(call, fin, torch))     # (a second AUTO sample: the first one of a process pays clock ramp-up)
    envs = {" ".join(f"{k[4:]}={v}" for k, v in env.items()): env for env in SHORT1D}

    def again(key):
        if key is None:
            return _time(call, fin, torch)
        if key in ("gather", "bucketed"):
            it.strategy.path = pkg.PATH_GATHER if key == "gather" else pkg.PATH_BUCKETED
            try:
                return _time(call, fin, torch)
            finally:
                it.strategy.path = pkg.PATH_AUTO
        with knobs(envs[key]):
            return _time(call, fin, torch)
    try:
        _judge(name, t_auto, forced, again)
    finally:
        it.strategy.release()


@pytest.mark.parametrize("name,dt,nx,ny,C,Q", GRID_2D, ids=[g[0] for g in GRID_2D])
def test_auto_is_within_15_percent_of_the_best_forced_form_2d(pkg, name, dt, nx, ny, C, Q):
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dt == np.float64 else torch.float32
    rng = np.random.default_rng(nx + C)
    x = np.cumsum(rng.uniform(0.5, 1.5, nx)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, ny)).astype(dt)
    g = torch.rand((nx, ny, C) if C > 1 else (nx, ny), dtype=tdt, device=dev)
    it = pkg.Interp2DBuilder.new(g).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    del g
    qx = torch.rand(Q, dtype=tdt, device=dev) * float(x[-1] - x[0]) * 0.999 + float(x[0])
    qy = torch.rand(Q, dtype=tdt, device=dev) * float(y[-1] - y[0]) * 0.999 + float(y[0])
    out = torch.empty((Q, C), dtype=tdt, device=dev)
    call = lambda: it.strategy.interp_array_into(it, qx, qy, out, async_launch=True)
    fin = it.strategy.finish
    it.strategy.path = pkg.PATH_AUTO
    t_auto = _time(call, fin, torch)
    forced = {}
    for pname, path in (("gather", pkg.PATH_GATHER), ("bucketed", pkg.PATH_BUCKETED)):
        it.strategy.path = path
        forced[pname] = _time(call, fin, torch)
    it.strategy.path = pkg.PATH_AUTO
    for env in FORMS2D:
        with knobs(env):
            forced[" ".join(f"{k[4:]}={v}" for k, v in env.items())] = _time(call, fin, torch)
    t_auto = min(t_auto, _time(call, fin, torch))
    envs = {" ".join(f"{k[4:]}={v}" for k, v in env.items()): env for env in FORMS2D}

    def again(key):
        if key is None:
            return _time(call, fin, torch)
        if key in ("gather", "bucketed"):
            it.strategy.path = pkg.PATH_GATHER if key == "gather" else pkg.PATH_BUCKETED
            try:
                return _time(call, fin, torch)
            finally:
                it.strategy.path = pkg.PATH_AUTO
        with knobs(envs[key]):
            return _time(call, fin, torch)
    try:
        _judge(name, t_auto, forced, again)
    finally:
        it.strategy.release()
        torch.cuda.empty_cache()
