"""Randomised check of the two-level 2-D grouping (coarse_scatter2d_kernel / scan_bin_totals_kernel /
fine_scatter2d[_sorted]_kernel, csrc/kernels.hpp) behind the tile-grouped Bilinear order (bilinear.rs:64-99): random grid
shapes, channel counts, batch sizes from a few queries to several rounds per tile row, evenly spread and clustered query
distributions, both element types (f32 = the compact-record, LDS-sorted passes; f64 = the two-array, direct passes).  The
grouped result must equal the gather order's on the same batch bit for bit (every record reaches exactly one tile, no
record twice, none lost: a lost or doubled record shows as an untouched or a wrong row), and a sample of rows is compared
with the CPU oracle."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(12))
def test_two_level_grouping_fuzz(pkg, capfd, seed):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 + seed)
    dt, tdt = (np.float32, torch.float32) if seed % 3 else (np.float64, torch.float64)
    nx, ny = int(rng.integers(18, 420)), int(rng.integers(18, 420))
    C = int(rng.choice([32, 64, 128] if dt == np.float32 else [16, 32, 64]))     # (rows of 64 bytes are kept pair-packed: gather only)
    Q = int(rng.choice([7, 300, 5_000, 60_000, 250_000]))
    x = np.cumsum(rng.uniform(0.2, 2.0, nx)).astype(dt); y = np.cumsum(rng.uniform(0.2, 2.0, ny)).astype(dt)
    g = rng.uniform(-1, 1, (nx, ny, C)).astype(dt)
    kind = seed % 4
    if kind == 0:
        qx = rng.uniform(x[0], x[-1], Q); qy = rng.uniform(y[0], y[-1], Q)
    elif kind == 1:      # a few hot spots
        cx = rng.uniform(x[0], x[-1], 5); cy = rng.uniform(y[0], y[-1], 5)
        k = rng.integers(0, 5, Q)
        qx = np.clip(cx[k] + rng.normal(0, 0.01 * (x[-1] - x[0]), Q), x[0], x[-1])
        qy = np.clip(cy[k] + rng.normal(0, 0.01 * (y[-1] - y[0]), Q), y[0], y[-1])
    elif kind == 2:      # one thin stripe along y
        qx = rng.uniform(x[nx // 3], x[nx // 3 + 1], Q); qy = rng.uniform(y[0], y[-1], Q)
    else:                # sorted queries (long same-tile runs in query order)
        qx = np.sort(rng.uniform(x[0], x[-1], Q)); qy = rng.uniform(y[0], y[-1], Q)
    qx = qx.astype(dt); qy = qy.astype(dt)
    it = pkg.Interp2DBuilder.new(torch.as_tensor(g, device=dev)).x(torch.as_tensor(x, device=dev)).y(torch.as_tensor(y, device=dev)).build()
    qxd, qyd = torch.as_tensor(qx, device=dev), torch.as_tensor(qy, device=dev)
    outs = {}
    os.environ["NDI_GROUP_TWO_LEVEL"] = "1"
    os.environ["NDI_TRACE_PLAN"] = "1"
    capfd.readouterr()
    try:
        for name, path in (("gather", pkg.PATH_GATHER), ("tiled", pkg.PATH_BUCKETED)):
            it.strategy.path = path
            out = torch.full((Q, C), -7.0, dtype=tdt, device=dev)
            it.interp_array_into(qxd, qyd, out)
            outs[name] = out
            assert pkg.profile_read(reset=False)["last_path"] == ("gather" if name == "gather" else "bucketed"), (seed, name)
    finally:
        for k in ("NDI_GROUP_TWO_LEVEL", "NDI_TRACE_PLAN"):
            os.environ.pop(k, None)
    err = capfd.readouterr().err
    if Q >= 2:
        assert "two-level grouping" in err, err[-500:]
    assert torch.equal(outs["gather"], outs["tiled"]), (seed, nx, ny, C, Q, kind)
    sel = rng.choice(Q, min(Q, 2000), replace=False)
    ref = oracle.interp2d_bilinear(x, y, g, qx[sel], qy[sel])[3].reshape(-1, C)
    assert np.array_equal(outs["tiled"].cpu().numpy()[sel], ref), (seed, "oracle")
    it.strategy.path = pkg.PATH_GATHER
