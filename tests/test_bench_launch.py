"""bench.py's N>1 launch skeleton on the CPU (gloo, world size 2): `python bench.py --gpus 2` must start its own
rank processes before touching the GPU, and the same file must run under an external torch.distributed.run --
the two ways the driver may start it (benches/bench_interp1d.rs:49-79 is the reference's
multi-worker shape).  --launch-rehearsal skips the GPU work; everything else (env, rendezvous, barrier,
max-over-ranks, single JSON line from rank 0) is the code path of the real run."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-rehearsal"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["max_over_ranks"] == 2.0


def test_bench_runs_under_an_external_launcher():
    port = 29500 + (os.getpid() % 2000) + 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-rehearsal"],
                       capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr
    assert _one_json_line(r.stdout)["n_gpus"] == 2


def test_bench_single_rank_rehearsal():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-rehearsal"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert _one_json_line(r.stdout)["n_gpus"] == 1
