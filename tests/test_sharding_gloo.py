"""CPU, world_size 2, gloo: the N>1 path.  Queries shard into contiguous blocks per rank with the tables
replicated and no data-path collective; the only cross-rank step is the MIN all-reduce that reproduces
the reference's first-error result.  The per-shard evaluator injected here is the CPU
oracle (tests may use it); on the GPUs the same helper wraps the device evaluation (bench.py)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["NDI_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NDI_ROOT"], "tests"))
import oracle
from conftest import load_product_package
pkg = load_product_package()
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(1)            # same inputs on every rank: tables are replicated
n, L, Q = 64, 9, 10007
x = np.sort(rng.uniform(0, 1, n)); y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
mode = os.environ["NDI_CASE"]
if mode == "err":
    q[7001] = 5.0; q[9000] = -3.0; q[200 + 5003] = 9.0      # failures in both shards; lowest is 5203
if mode == "nan":
    q[8000] = np.nan                                        # in rank 1's shard; the other rank has no failure
st, a, b = oracle.cubic_build(x, y)
out = np.full((Q, L), -1.0)
def evaluate(lo, hi):
    if mode == "boom" and rank == 1:
        raise RuntimeError("HIP_ERROR: device lost")         # a device failure on one rank only
    ex = oracle.EXTRAPOLATE_YES if mode == "nan" else oracle.EXTRAPOLATE_NO
    s, fail, _ = oracle.interp1d_cubic(x, y, a, b, q[lo:hi], extrapolate=ex, out=out[lo:hi])
    if s == oracle.OUT_OF_BOUNDS:
        raise pkg.InterpolateError.OutOfBounds("x = ? is not in range", index=fail, value=float(q[lo + fail]))
    if s == oracle.NAN_QUERY:
        raise pkg.Panic("not implemented: failed to convert NaN to usize", index=fail)
local, exc = pkg.sharding.eval_shard(evaluate, Q, rank, world)
first = pkg.sharding.first_error_across_ranks(local)
lo, hi = pkg.sharding.shard_bounds(Q, rank, world)
# the packaged protocol on the same inputs: every rank enters the all-reduce, the owner re-raises
outcome = "ok"
try:
    pkg.sharding.eval_sharded(evaluate, Q, rank, world)
except Exception as e:
    outcome = f"{type(e).__name__}:{getattr(e, 'index', None)}"
np.save(os.path.join(os.environ["NDI_OUT"], f"out{rank}.npy"), out[lo:hi])
with open(os.path.join(os.environ["NDI_OUT"], f"first{rank}.txt"), "w") as f:
    f.write(f"{first} {lo} {hi}")
with open(os.path.join(os.environ["NDI_OUT"], f"outcome{rank}.txt"), "w") as f:
    f.write(outcome)
dist.barrier()
dist.destroy_process_group()
'''


def _run(case):
    import oracle
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        open(script, "w").write(WORKER)
        env = dict(os.environ, NDI_ROOT=ROOT, NDI_OUT=tmp, NDI_CASE=case, MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(29500 + (os.getpid() % 2000)), WORLD_SIZE="2", OMP_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
                 for r in range(2)]
        for p in procs:
            assert p.wait(timeout=240) == 0
        firsts, parts = [], []
        global LAST_OUTCOMES
        LAST_OUTCOMES = []
        for r in range(2):
            first, lo, hi = map(int, open(os.path.join(tmp, f"first{r}.txt")).read().split())
            firsts.append(first)
            parts.append((lo, hi, np.load(os.path.join(tmp, f"out{r}.npy"))))
            op = os.path.join(tmp, f"outcome{r}.txt")
            LAST_OUTCOMES.append(open(op).read() if os.path.exists(op) else None)
        return firsts, parts


LAST_OUTCOMES = []


def _inputs():
    rng = np.random.default_rng(1)
    n, L, Q = 64, 9, 10007
    x = np.sort(rng.uniform(0, 1, n)); y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
    return x, y, q


def test_two_rank_shards_equal_single_process(pkg):
    import oracle
    firsts, parts = _run("ok")
    assert firsts == [pkg.sharding.NO_FAIL] * 2
    x, y, q = _inputs()
    st, a, b = oracle.cubic_build(x, y)
    _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
    assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == q.size
    assert np.array_equal(np.concatenate([p[2] for p in parts]), ref)


def test_two_rank_first_error_is_global_minimum(pkg):
    firsts, parts = _run("err")
    assert firsts == [5203, 5203]          # both ranks learn the reference's first failing index
    # rank 1 owns query 5203 and re-raises its OutOfBounds carrying the flat index of the whole batch (ADVICE r2:
    # every rank names the same index, the one the reference's serial loop would); rank 0 is told which query failed
    assert LAST_OUTCOMES == ["ShardFailed:5203", "OutOfBounds:5203"]


def test_two_rank_panic_and_device_failure_do_not_hang(pkg):
    """a NaN query while extrapolating (a panic in the reference) or a device failure on ONE rank must
    not leave the other rank blocked in the MIN all-reduce; the panic's index is the global first failure."""
    firsts, _ = _run("nan")
    assert firsts == [8000, 8000]
    assert LAST_OUTCOMES == ["ShardFailed:8000", "Panic:8000"]
    firsts, _ = _run("boom")
    assert firsts == [pkg.sharding.DEVICE_FAILED] * 2   # a device failure outranks every query index on all ranks
    assert LAST_OUTCOMES == ["ShardFailed:-1", "RuntimeError:None"]


GPU_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["NDI_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NDI_ROOT"], "tests"))
from conftest import load_product_package
pkg = load_product_package()
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")                     # rehearsal: both ranks share the one GPU of the box
rng = np.random.default_rng(1)
n, L, Q = 64, 1024, 20011
x = np.sort(rng.uniform(0, 1, n)); y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
if os.environ["NDI_CASE"] == "err":
    q[15001] = 5.0; q[19000] = -3.0; q[400 + 10005] = 9.0
interp = pkg.Interp1DBuilder.new(torch.as_tensor(y, device=dev)).x(torch.as_tensor(x, device=dev)) \
    .strategy(pkg.CubicSpline.new()).build()     # tables replicated per rank
qd = torch.as_tensor(q, device=dev)
lo, hi = pkg.sharding.shard_bounds(Q, rank, world)
out = torch.full((hi - lo, L), -1.0, dtype=torch.float64, device=dev)
def evaluate(a, b):
    interp.interp_array_into(qd[a:b], out)
local, exc = pkg.sharding.eval_shard(evaluate, Q, rank, world)
first = pkg.sharding.first_error_across_ranks(local)
np.save(os.path.join(os.environ["NDI_OUT"], f"out{rank}.npy"), out.cpu().numpy())
with open(os.path.join(os.environ["NDI_OUT"], f"first{rank}.txt"), "w") as f:
    f.write(f"{first} {lo} {hi}")
dist.barrier()
dist.destroy_process_group()
'''


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ok", "err"])
def test_two_ranks_through_the_device_path(pkg, case):
    """The same sharding helpers driving the real device evaluation (two processes, gloo, one GPU shared)."""
    import oracle
    global WORKER
    saved, WORKER = WORKER, GPU_WORKER
    try:
        firsts, parts = _run(case)
    finally:
        WORKER = saved
    rng = np.random.default_rng(1)
    n, L, Q = 64, 1024, 20011
    x = np.sort(rng.uniform(0, 1, n)); y = rng.uniform(0, 1, (n, L)); q = rng.uniform(x[0], x[-1], Q)
    st, a, b = oracle.cubic_build(x, y)
    if case == "ok":
        assert firsts == [pkg.sharding.NO_FAIL] * 2
        _, _, ref = oracle.interp1d_cubic(x, y, a, b, q)
        assert np.array_equal(np.concatenate([p[2] for p in parts]), ref)
    else:
        assert firsts == [10405, 10405]
        # rank 1's shard: rows before its local first error are written, later rows untouched
        lo1 = parts[1][0]
        _, _, ref = oracle.interp1d_cubic(x, y, a, b, q[lo1:10405])
        assert np.array_equal(parts[1][2][:10405 - lo1], ref) and np.all(parts[1][2][10405 - lo1:] == -1.0)
