import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(42)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
x = np.unique(rng.uniform(0, 1, 2 * n))[:n]
y = rng.uniform(0, 1, (n, L))
interp = pkg.Interp1DBuilder.new(y).x(x).strategy(pkg.CubicSpline.new()).build()
q = rng.uniform(x[0], x[-1], Q)
qd = torch.as_tensor(q, device=dev)
a = torch.empty((Q, L), dtype=torch.float64, device=dev)
b = torch.full((Q, L), -5.0, dtype=torch.float64, device=dev)
interp.strategy.path = pkg.PATH_GATHER
interp.interp_array_into(qd, a)
interp.strategy.path = pkg.PATH_BUCKETED
interp.interp_array_into(qd, b)
rows_bad = (a != b).any(dim=1)
print("bad rows", int(rows_bad.sum()), "of", Q)
untouched = (b == -5.0).all(dim=1)
print("untouched rows", int(untouched.sum()))
half0 = (a[:, :L // 2] != b[:, :L // 2]).any(dim=1); half1 = (a[:, L // 2:] != b[:, L // 2:]).any(dim=1)
print("bad in first half of lanes", int(half0.sum()), "second half", int(half1.sum()))
idx = torch.as_tensor(np.clip(np.searchsorted(x, q, side="right") - 1, 0, n - 2), device=dev)
bad_idx = idx[rows_bad]
print("distinct intervals among bad rows", int(torch.unique(bad_idx).numel()), "of", int(torch.unique(idx).numel()))
if int(rows_bad.sum()):
    bi = torch.nonzero(rows_bad)[:10, 0].cpu().numpy()
    print("first bad rows", bi, "their idx", idx[bi].cpu().numpy())
    cnt = torch.bincount(idx, minlength=n)
    print("bucket sizes of first bad idx", cnt[idx[bi]].cpu().numpy())
    good_idx = idx[~rows_bad]
    print("bucket size stats bad: mean", float(cnt[bad_idx].double().mean()), "good:", float(cnt[good_idx].double().mean()))
