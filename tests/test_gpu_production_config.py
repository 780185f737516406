"""The PRODUCTION configuration of the library, which the rest of the suite never runs: conftest sets
NDI_TUNE_LIVE for the whole session so that one process can walk the kernel variants, which makes every evaluation re-read
its knobs -- in production they are read once into function-local statics.  Here a child process without ANY NDI_* variable
(no live knobs, the release library, no bounds checks) runs __graft_entry__.smoke() and a compact parity pass over every
1-D / 2-D formulation against the CPU oracle.  A second child packs every eligible small grid pair-wise (NDI_PAIR_PACK=1):
grids that fit LDS stay plain by default since round 5, so the pair-packed kernels on small grids are covered here.
One child at a time (the box allows few processes on the card)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
assert not [k for k in os.environ if k.startswith("NDI_") and k not in {allowed!r}], "the child must run without knobs"
import numpy as np, torch
sys.path.insert(0, {root!r})
import __graft_entry__ as g
import oracle
if {smoke!r}:
    g.smoke()
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
def eq(a, b, what):
    assert np.array_equal(np.asarray(a).reshape(np.asarray(b).shape), b), what
# 1-D: long rows (gather / bucketed), short rows (query order, grouped), scalar rows (one thread per query, lanes kernels)
for dt in (np.float64, np.float32):
    for n, L, Q in ((64, 4096, 3000), (300, 512, 4000), (1024, 8, 90_000), (100, 5, 300_000), (100, 1, 400_000), (100, 1, 5_000),
                    (2000, 64, 120_000)):
        x = np.unique(rng.uniform(0, 1, 4 * n).astype(dt))[:n]
        y = rng.uniform(-1, 1, (x.size, L)).astype(dt)
        q = rng.uniform(x[0], x[-1], Q).astype(dt)
        st, a, b = oracle.cubic_build(x, y)
        refc = oracle.interp1d_cubic(x, y, a, b, q)[2]
        refl = oracle.interp1d_linear(x, y, q)[2]
        yd, xd, qd = (torch.as_tensor(v, device=dev) for v in (y, x, q))
        cub = pkg.Interp1DBuilder.new(yd).x(xd).strategy(pkg.CubicSpline.new()).build()
        lin = pkg.Interp1DBuilder.new(yd).x(xd).build()
        for path in (pkg.PATH_AUTO, pkg.PATH_GATHER, pkg.PATH_BUCKETED):
            cub.strategy.path = path; lin.strategy.path = path
            eq(cub.interp_array(qd).cpu().numpy(), refc, ("cubic", dt, n, L, path))
            eq(lin.interp_array(qd).cpu().numpy(), refl, ("linear", dt, n, L, path))
        eq(cub.interp_array(q), refc, ("cubic host", dt, n, L))
# 2-D: gather, tile-grouped, query order, one thread per query, grid in LDS
for dt in (np.float64, np.float32):
    for nx, ny, C, Q in ((300, 280, 64, 200_000), (64, 48, 4, 150_000), (100, 100, 1, 300_000), (100, 100, 5, 100_000), (40, 30, 16, 3_000)):
        x = np.cumsum(rng.uniform(0.5, 1.5, nx)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, ny)).astype(dt)
        gr = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
        qx = rng.uniform(x[0], x[-1], Q).astype(dt); qy = rng.uniform(y[0], y[-1], Q).astype(dt)
        ref = oracle.interp2d_bilinear(x, y, gr, qx, qy)[3]
        it = pkg.Interp2DBuilder.new(torch.as_tensor(gr, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
        for path in (pkg.PATH_AUTO, pkg.PATH_GATHER, pkg.PATH_BUCKETED):
            it.strategy.path = path
            eq(it.interp_array(torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev)).cpu().numpy(), ref, ("bilinear", dt, nx, ny, C, path))
        eq(it.interp_array(qx, qy), ref, ("bilinear host", dt, nx, ny, C))
print("production-ok")
"""


def _run(extra_env, smoke):
    env = {k: v for k, v in os.environ.items() if not k.startswith("NDI_")}
    env.update(extra_env)
    code = CHILD.format(root=ROOT, allowed=set(extra_env), smoke=smoke)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "production-ok" in r.stdout, (extra_env, r.stdout[-2000:], r.stderr[-4000:])


def test_release_library_without_any_knob():
    _run({}, smoke=True)


def test_pair_packed_small_grids():
    _run({"NDI_PAIR_PACK": "1"}, smoke=False)
