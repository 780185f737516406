set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python tools/ref_shapes_bench.py > gpurun_out/r02/reference_shapes.json 2> gpurun_out/r02/reference_shapes.err; echo refshapes rc=$?; tail -3 gpurun_out/r02/reference_shapes.err
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/target_stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02/target_bench_under_rocprof.json 2>/dev/null; echo rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02/target_fetch --output-format csv -- python3 $B > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02/target_write --output-format csv -- python3 $B > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02/target_gather_fetch --output-format csv -- python3 $B --path gather > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02/target_gather_write --output-format csv -- python3 $B --path gather > /dev/null 2>&1; echo rc=$?
C5="bench.py --workload c5 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/c5b_stats --output-format csv -- python3 bench.py --workload c5 --steps 10 --warmup 3 > gpurun_out/r02/c5b_bench.json 2>/dev/null; echo rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02/c5b_fetch --output-format csv -- python3 $C5 > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02/c5b_write --output-format csv -- python3 $C5 > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r02/c5b_tcc --output-format csv -- python3 $C5 > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/c3b_stats --output-format csv -- python3 bench.py --workload c3 --steps 10 --warmup 3 > gpurun_out/r02/c3b_bench.json 2>/dev/null; echo rc=$?
ls gpurun_out/r02/*/ | head -50
