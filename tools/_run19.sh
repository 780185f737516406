cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.log 2>&1; echo pytest rc=$?; tail -4 gpurun_out/r02b/pytest.log
python bench.py > gpurun_out/r02b/bench_default.json 2> gpurun_out/r02b/bench_default.err; echo bench rc=$?
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02b/target_stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02b/target_bench_under_rocprof.json 2>/dev/null; echo rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02b/target_fetch --output-format csv -- python3 $B > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02b/target_write --output-format csv -- python3 $B > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --stats -d gpurun_out/r02b/c5_stats --output-format csv -- python3 bench.py --workload c5 --steps 10 --warmup 3 > gpurun_out/r02b/c5_bench.json 2>/dev/null; echo rc=$?
rocprofv3 --kernel-trace --stats -d gpurun_out/r02b/c3_stats --output-format csv -- python3 bench.py --workload c3 --steps 10 --warmup 3 > gpurun_out/r02b/c3_bench.json 2>/dev/null; echo rc=$?
python bench.py --workload c5 --even-axes --steps 10 --warmup 3 > gpurun_out/r02b/c5_even.json 2>/dev/null
python bench.py --workload c2 --steps 10 --warmup 3 > gpurun_out/r02b/c2_single.json 2>/dev/null
python tools/ref_shapes_bench.py > gpurun_out/r02b/reference_shapes.json 2>/dev/null; echo refshapes rc=$?
