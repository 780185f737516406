cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.log 2>&1; echo pytest rc=$?; tail -4 gpurun_out/r02b/pytest.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r02b/target_stats2 --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 > gpurun_out/r02b/target_bench_under_rocprof.json 2>/dev/null; echo rc=$?
python tools/ref_shapes_bench.py > gpurun_out/r02b/reference_shapes.json 2>/dev/null; echo refshapes rc=$?
python bench.py --workload c5 --even-axes --steps 10 --warmup 3 2>/dev/null
python bench.py --workload c3 --even-axes --steps 10 --warmup 3 2>/dev/null
