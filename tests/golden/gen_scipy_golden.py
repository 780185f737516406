#!/usr/bin/env python3
"""Generate tests/golden/scipy_cubic.npz.

The reference's spline tests name scipy.interpolate.CubicSpline as the source
of their expected values (tests/cubic_spline_strat.rs:19,71,120,271,324,377 --
rounded there to 8 digits and asserted at max_relative=1e-3).  This script
regenerates such vectors at full f64 precision with scipy (an independent
third-party library, NOT the reference) on (i) the reference's own 12-point
data sets and (ii) seeded jittered / log-spaced grids up to n=4096, so the CPU
oracle can be pinned at the 1e-10 bar of BASELINE.json.

Known reference deviation (kept, the parity target is the reference): with a NotAKnot
boundary on the RIGHT end the reference sets a_mid[len-1] = dx_1 (cubic_spline.rs:635)
where scipy uses dx[-2]; the two coincide on uniform grids only.  The `*_nk` cases on the
non-uniform grids ("jit", "log") are therefore NOT used to pin the oracle (see
tests/test_oracle_golden.py::test_cubic_vs_scipy); the uniform-grid ones ("uni", "ref12") are.

Run:  python tests/golden/gen_scipy_golden.py      (scipy 1.15.3 was used)
"""
import os

import numpy as np
from scipy.interpolate import CubicSpline

NK, NAT, CL, D1, D2 = 0, 1, 2, 3, 4
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scipy_cubic.npz")


def bc_to_scipy(periodic, left, right, lanes):
    if periodic:
        return "periodic"

    def one(kind, val):
        val = np.full(lanes, val)
        if kind == NK:
            return "not-a-knot"
        if kind == NAT:
            return "natural"
        if kind == CL:
            return "clamped"
        if kind == D1:
            return (1, val)
        return (2, val)

    return (one(*left), one(*right))


def main():
    rng = np.random.default_rng(20240611)
    store = {}
    names = []

    def add(name, x, y, periodic, left, right, q):
        bc = bc_to_scipy(periodic, left, right, y.shape[1])
        if periodic:
            y = y.copy()
            y[-1] = y[0]
        cs = CubicSpline(x, y, axis=0, bc_type=bc, extrapolate="periodic" if periodic else True)
        exp = cs(q)
        store[name + "/x"] = x
        store[name + "/y"] = y
        store[name + "/q"] = q
        store[name + "/expect"] = exp
        store[name + "/bc"] = np.array([int(periodic), left[0], left[1], right[0], right[1]], dtype=np.float64)
        names.append(name)

    bcs = {
        "nk": (False, (NK, 0.0), (NK, 0.0)),
        "nat": (False, (NAT, 0.0), (NAT, 0.0)),
        "cl": (False, (CL, 0.0), (CL, 0.0)),
        "d1": (False, (D1, -0.1), (D1, -0.5)),
        "d2": (False, (D2, -0.1), (D2, -0.5)),
        "mix": (False, (NK, 0.0), (D1, 0.5)),
        "per": (True, (NK, 0.0), (NK, 0.0)),
    }
    # (i) the reference's own 12-point set (tests/cubic_spline_strat.rs:59) on x = 0..11
    y12 = np.array([1.0, 2.0, 2.5, 2.5, 3.0, 2.0, 1.0, -2.0, 3.0, 5.0, 6.3, 8.0])[:, None]
    x12 = np.arange(12.0)
    q12 = np.linspace(-3.0, 15.0, 181)
    for k, (p, l, r) in bcs.items():
        add(f"ref12_{k}", x12, y12, p, l, r, q12)
    # (ii) seeded grids
    for n in (4, 5, 64, 1024, 4096):
        L = 3
        base = np.linspace(0.0, 1.0, n)
        jit = base + rng.uniform(-0.2 / n, 0.2 / n, n)       # cf. bench_vector_extensions.rs:36-40
        jit.sort()
        logx = np.logspace(-2.0, 0.0, n)                      # cf. bench_vector_extensions.rs:69
        uni = np.linspace(-1.0, 2.0, n)
        for gname, x in (("uni", uni), ("jit", jit), ("log", logx)):
            y = rng.uniform(0.0, 1.0, (n, L))
            span = x[-1] - x[0]
            q = np.concatenate([
                rng.uniform(x[0], x[-1], 160),
                x[rng.integers(0, n, 16)],                    # exact knot hits
                np.array([x[0], x[-1]]),
                rng.uniform(x[0] - 0.02 * span / n, x[0], 8),       # slight extrapolation
                rng.uniform(x[-1], x[-1] + 0.02 * span / n, 8),
            ])
            for k, (p, l, r) in bcs.items():
                add(f"{gname}{n}_{k}", x, y, p, l, r, q)
    store["names"] = np.array(names)
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {len(names)} cases, {os.path.getsize(OUT)/1e6:.2f} MB")


if __name__ == "__main__":
    main()
