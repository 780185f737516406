cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
python -m pytest tests -m gpu -x -q > gpurun_out/r02c/pytest.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r02c/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r02c/bench_default.json 2> gpurun_out/r02c/bench_default.err; echo bench rc=$?
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02c/target_stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 > gpurun_out/r02c/target_bench_under_rocprof.json 2>/dev/null; echo rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02c/target_fetch --output-format csv -- python3 $B > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02c/target_write --output-format csv -- python3 $B > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02c/target_gather_fetch --output-format csv -- python3 $B --path gather > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02c/target_gather_write --output-format csv -- python3 $B --path gather > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --stats -d gpurun_out/r02c/gather_stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 --path gather > gpurun_out/r02c/gather_bench_under_rocprof.json 2>/dev/null; echo rc=$?
C5="bench.py --workload c5 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02c/c5_stats --output-format csv -- python3 bench.py --workload c5 --steps 10 --warmup 3 > gpurun_out/r02c/c5_bench.json 2>/dev/null; echo rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02c/c5_fetch --output-format csv -- python3 $C5 > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02c/c5_write --output-format csv -- python3 $C5 > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r02c/c5_tcc --output-format csv -- python3 $C5 > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --stats -d gpurun_out/r02c/c3_stats --output-format csv -- python3 bench.py --workload c3 --steps 10 --warmup 3 > gpurun_out/r02c/c3_bench.json 2>/dev/null; echo rc=$?
python bench.py --workload c5 --even-axes --steps 10 --warmup 3 > gpurun_out/r02c/c5_even.json 2>/dev/null
python bench.py --workload c3 --even-axes --steps 10 --warmup 3 > gpurun_out/r02c/c3_even.json 2>/dev/null
python bench.py --workload c2 --steps 10 --warmup 3 > gpurun_out/r02c/c2_single.json 2>/dev/null
python tools/ref_shapes_bench.py > gpurun_out/r02c/reference_shapes.json 2>/dev/null; echo refshapes rc=$?
python tools/sweep_target.py > gpurun_out/r02c/sweep_target.jsonl 2>/dev/null; echo sweep rc=$?
python bench.py --gpus 2 --backend gloo --device-override 0 --steps 3 --warmup 1 --placement-probe 0 --no-gather-leg --chunk 1000000 > gpurun_out/r02c/bench_2ranks_rehearsal.txt 2>/dev/null; echo bench2 rc=$?
