"""CubicSpline::build at BASELINE configs[1]'s shape, five times -- the unit tools/profile_r06.sh profiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
dev = torch.device("cuda:0")
n = L = 4096
x = torch.cumsum(torch.rand(n, dtype=torch.float64, device=dev) + 0.5, 0)
y = torch.rand((n, L), dtype=torch.float64, device=dev)
for _ in range(5):
    it = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
    torch.cuda.synchronize()
    it.strategy.release()
