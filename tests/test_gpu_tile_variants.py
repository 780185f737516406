"""The A/B variants of the 2-D tile-grouped order (eval_bilinear_tiles_kernel, bilinear.rs:64-99) that the default launch
never takes -- no staged x slopes (NDI_TILE_SLOPES=0: three divisions per channel), two 512-thread workgroups per CU
(NDI_TILE_WG=512), other tile sizes (NDI_TILE_TS), other chunk sizes, plain (xi, yi) arrays instead of cell words, the two-level grouping forced on or off -- must
give the gather order's bits on the same batch.  The knobs are read once per process, so every variant is one child
process (one at a time: the box allows few processes on the card)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
for dt, tdt, nx, ny, C, Q in ((np.float32, torch.float32, 500, 515, 64, 700_000), (np.float64, torch.float64, 200, 333, 32, 200_000),
                              (np.float32, torch.float32, 130, 97, 128, 90_000)):
    x = np.cumsum(rng.uniform(0.5, 1.5, nx)).astype(dt); y = np.cumsum(rng.uniform(0.5, 1.5, ny)).astype(dt)
    grid = torch.as_tensor(rng.uniform(-1, 1, (nx, ny, C)).astype(dt), device=dev)
    it = pkg.Interp2DBuilder.new(grid).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    qx = torch.as_tensor(rng.uniform(x[0], x[-1], Q).astype(dt), device=dev)
    qy = torch.as_tensor(rng.uniform(y[0], y[-1], Q).astype(dt), device=dev)
    qx[:3] = torch.as_tensor([x[0], x[-1], x[5]], device=dev); qy[:3] = torch.as_tensor([y[-1], y[0], y[7]], device=dev)
    outs = {{}}
    for name, path in (("gather", pkg.PATH_GATHER), ("tiled", pkg.PATH_BUCKETED)):
        it.strategy.path = path
        out = torch.full((Q, C), -7.0, dtype=tdt, device=dev)
        it.interp_array_into(qx, qy, out)
        outs[name] = out
        assert pkg.profile_read(reset=False)["last_path"] == ("gather" if name == "gather" else "bucketed"), name
    assert torch.equal(outs["gather"], outs["tiled"]), (np.dtype(dt).name, nx, ny, C)
print("variants-ok")
"""

VARIANTS = [
    {},                                   # what ships (the control of this harness): two half-row workgroups per CU
    {"NDI_TILE_SPLIT": "0"},              # one 1024-thread workgroup per CU staging whole rows (round 4's launch)
    {"NDI_TILE_SPLIT": "0", "NDI_GROUP_BLOCKS": "32"},
    {"NDI_TILE_SLOPES": "0"},
    {"NDI_TILE_WG": "512", "NDI_TILE_SLOPES": "0"},
    {"NDI_TILE_TS": "2"},
    {"NDI_TILE_TS": "3", "NDI_TILE_CHUNK": "1000"},
    {"NDI_TILE_CELLWORDS": "0", "NDI_TILE_CHUNK": "100000"},
    # two-level grouping (tile row, then tile: coarse_scatter2d_kernel + fine_scatter2d_kernel) forced on / off
    {"NDI_GROUP_TWO_LEVEL": "1"},
    {"NDI_GROUP_TWO_LEVEL": "0"},
    {"NDI_GROUP_TWO_LEVEL": "1", "NDI_TILE_TS": "2", "NDI_GROUP_FINE_THREADS": "256"},
    {"NDI_GROUP_TWO_LEVEL": "1", "NDI_TILE_CELLWORDS": "0", "NDI_TILE_TS": "3", "NDI_GROUP_BLOCKS": "32"},
    {"NDI_GROUP_TWO_LEVEL": "1", "NDI_TILE_SPLIT": "0", "NDI_TILE_CHUNK": "1000"},
    # both passes write their runs DIRECTLY from registers (one request per record) instead of sorting a round in LDS
    {"NDI_GROUP_TWO_LEVEL": "1", "NDI_GROUP_FINE_SORT": "0", "NDI_GROUP_COARSE_SORT": "0"},
    {"NDI_GROUP_TWO_LEVEL": "1", "NDI_GROUP_FINE_SORT": "0", "NDI_GROUP_COARSE_SORT": "1", "NDI_GROUP_BLOCKS": "32", "NDI_TILE_TS": "3"},
    {"NDI_GROUP_TWO_LEVEL": "1", "NDI_GROUP_FINE_SORT": "5", "NDI_GROUP_COARSE_SORT": "0"},
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "default")
def test_tile_variant_matches_gather_order(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "variants-ok" in r.stdout, (env, r.stdout[-2000:], r.stderr[-4000:])
