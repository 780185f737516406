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
    """The contract: exactly ONE JSON line on stdout, and it is the last line.  The secondary legs' long tables go out before
    it on a line prefixed `detail ` (JSON behind the prefix), which no JSON-lines parser takes for a result."""
    all_lines = [l for l in stdout.splitlines() if l.strip()]
    lines = [l for l in all_lines if l.startswith("{")]
    assert len(lines) == 1 and all_lines[-1] == lines[0], stdout[-2000:]
    for l in all_lines:
        if l.startswith("detail "):
            assert "detail" in json.loads(l[len("detail "):])
    return json.loads(lines[0])


def _run(args, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                          timeout=timeout, env=env)


def test_bench_self_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-rehearsal"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["max_over_ranks"] == 2.0
    # the N > 1 line verifies itself: world size, backend, every rank's clock and device
    rk = line["ranks"]
    assert rk["world"] == 2 and rk["backend"] == "gloo" and rk["per_rank_ms"] == [1.0, 2.0]
    assert rk["per_rank_kernel_ms"] == [0.5, 1.5] and rk["slowest_rank"] == 1 and rk["distinct_devices"] == 2
    assert [d["rank"] for d in rk["devices"]] == [0, 1]
    # N > 1 lands on a BASELINE config without any flag: C4's per-GPU share (configs[3]: 1e8 queries over 8 GPUs)
    assert line["queries_per_gpu"] == 12_500_000 and line["workload"].startswith("C4 (BASELINE configs[3]) per-GPU share x 2")


def test_bench_a_failing_rank_takes_the_others_down():
    """a rank that dies must not leave its siblings (and the driver) waiting in a collective."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["NDI_BENCH_FAIL_RANK"] = "1"
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-rehearsal"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and time.time() - t0 < 120


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
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["ranks"] is None
    assert line["queries_per_gpu"] == 10_000_000 and line["workload"] == "Target"


def test_bench_eight_rank_rehearsal_is_c4_with_eight_ranks():
    """The N = 8 launch the driver makes (rehearsed on the CPU over gloo: 8 real GPU ranks are the driver's to start, and a
    1-GPU box admits at most 6 processes on its card): the defaults land on BASELINE configs[3] -- "= C4" --, the line
    carries 8 ranks' clocks and devices, and the per-rank C5-share gather carries 8 entries."""
    r = _run(["--gpus", "8", "--launch-rehearsal"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 8 and line["queries_per_gpu"] == 12_500_000
    assert line["workload"] == "C4 (BASELINE configs[3]) per-GPU share x 8 = C4"
    rk = line["ranks"]
    assert rk["world"] == 8 and len(rk["per_rank_ms"]) == 8 and len(rk["devices"]) == 8 and rk["slowest_rank"] == 7
    assert sorted(d["rank"] for d in rk["devices"]) == list(range(8))
    c5 = line["c5_share"]
    assert len(c5["per_rank_ms"]) == 8 and len(c5["per_rank_kernel_ms"]) == 8 and c5["ms_per_step"] == max(c5["per_rank_ms"])


import pytest


@pytest.mark.gpu
def test_bench_six_ranks_on_one_gpu_line_carries_every_rank():
    """The real N > 1 code path with as many ranks as a 1-GPU box admits on its card (6; gloo, every rank on device 0,
    a small C4-shaped workload, children spawned before any GPU call): 6 ranks' clocks in `ranks`, 6 per-rank entries in
    the C5-share leg, max-over-ranks timing."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "6", "--backend", "gloo",
                        "--device-override", "0", "--knots", "512", "--lanes", "512", "--queries", "200000",
                        "--chunk", "50000", "--steps", "2", "--warmup", "1", "--placement-probe", "0", "--no-check",
                        "--no-gather-leg", "--no-pmc", "--c5-leg-grid", "512", "--c5-leg-queries", "100000"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _one_json_line(r.stdout)
    rk = line["ranks"]
    assert line["n_gpus"] == 6 and rk["world"] == 6 and len(rk["per_rank_ms"]) == 6 and rk["distinct_devices"] == 1
    c5 = line["c5_share"]
    assert len(c5["per_rank_ms"]) == 6 and len(c5["per_rank_kernel_ms"]) == 6 and all(v > 0 for v in c5["per_rank_ms"])
    assert line["ms_per_step"] >= max(rk["per_rank_ms"]) * 0.999 and line["scaling"] == "weak"


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_line_carries_ranks():
    """The real N = 2 code path on a 1-GPU box (gloo, both ranks on device 0, a small Target-shaped workload): the
    JSON line carries `ranks` with both ranks' clocks, kernel times and the device they ran on."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--device-override", "0", "--knots", "512", "--lanes", "1024", "--queries", "400000",
                        "--chunk", "100000", "--steps", "3", "--warmup", "1", "--placement-probe", "0",
                        "--no-gather-leg", "--no-pmc", "--c5-leg-grid", "1024", "--c5-leg-queries", "300000"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _one_json_line(r.stdout)
    rk = line["ranks"]
    assert line["n_gpus"] == 2 and rk["world"] == 2 and rk["backend"] == "gloo" and len(rk["per_rank_ms"]) == 2
    # the N > 1 line names its workloads after BASELINE configs[3] / [4] and carries the per-rank C5-share leg
    assert line["config"]["workload"].startswith("C4-shaped rehearsal")
    c5 = line["c5_share"]
    assert c5["workload"].startswith("C5 (BASELINE configs[4]) per-GPU share") and len(c5["per_rank_ms"]) == 2
    assert all(v > 0 for v in c5["per_rank_ms"] + c5["per_rank_kernel_ms"]) and c5["ms_per_step"] == max(c5["per_rank_ms"])
    assert all(v > 0 for v in rk["per_rank_ms"] + rk["per_rank_kernel_ms"])
    assert [d["ordinal"] for d in rk["devices"]] == [0, 0] and rk["distinct_devices"] == 1
    assert line["ms_per_step"] >= max(rk["per_rank_ms"]) * 0.999 and line["check"]["bit_exact"]


@pytest.mark.gpu
def test_bench_force_dist_runs_the_collectives_on_rccl_with_one_rank():
    """`--gpus 1 --force-dist`: init_process_group("nccl", device_id=...), the barrier, the MAX all-reduce and
    all_gather_object of the N > 1 path execute on RCCL with world size 1 -- environment / API breakage shows up on
    the 1-GPU box, before the driver's 8-GPU run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--knots", "512",
                        "--lanes", "1024", "--queries", "400000", "--chunk", "100000", "--steps", "3", "--warmup", "1",
                        "--placement-probe", "0", "--no-gather-leg", "--no-pmc", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _one_json_line(r.stdout)
    rk = line["ranks"]
    assert line["n_gpus"] == 1 and rk["world"] == 1 and rk["backend"] == "nccl" and rk["distinct_devices"] == 1
    assert rk["per_rank_ms"][0] > 0 and rk["devices"][0]["ordinal"] == 0 and line["check"]["bit_exact"]


@pytest.mark.gpu
def test_bench_in_process_sharded_leg():
    """The leg the N = 1 run adds when its one process sees several devices (one ndi_interp1d_eval_ring_sharded call
    per step over all of them), rehearsed on the 1-GPU box with two replicas on device 0 and a small workload."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--knots", "512", "--lanes", "1024", "--queries",
                        "400000", "--chunk", "100000", "--steps", "2", "--warmup", "1", "--placement-probe", "0",
                        "--no-gather-leg", "--no-pmc", "--no-cpu-baseline", "--sharded-leg-devices", "0,0"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _one_json_line(r.stdout)
    leg = line["in_process_sharded"]
    assert leg["devices"] == [0, 0] and "error" not in leg and leg["Mpoints_s"] > 0 and leg["queries_per_device"] == 400000
    assert set(line["secondary"]) == {"detail_line", "reference_shapes_summary", "host_path", "c2", "c2_linear", "c2_f32", "c5_share",
                                      "c3", "c1"}
    # the BASELINE configs ride in `config` (what a truncating reader keeps) and, compactly, at the very end of the line
    summ = line["config"]["secondary_summary"]
    assert {"c3", "c5_share", "c2", "c2_linear", "c2_f32"} <= set(summ) and summ["c3"]["ms_per_step"] > 0 and summ["c2"]["frac"] > 0
    assert list(line)[-1] == "tail_summary" and list(line)[-2] == "secondary"
    assert line["secondary"]["c3"]["step_minus_kernels_ms"] >= 0 and line["secondary"]["c5_share"]["step_minus_kernels_ms"] >= 0
    assert line["roofline"].get("survey_8d_frac") is None   # (--no-gather-leg: the 8(d) basis belongs to the gather kernel's leg)
    # the leg is bounded in wall-clock time and says what it timed
    assert leg["leg_wall_s"] <= leg["budget_s"] + 30 and "timed" in leg and leg["first_step_s"] > 0
