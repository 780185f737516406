set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 300 ./tools/placement_probe 5 > gpurun_out/r02/placement_probe.jsonl 2>&1; echo placement rc=$?
timeout -k 10 200 ./tools/random_read_probe > gpurun_out/r02/random_read_probe.jsonl 2>&1; echo rr rc=$?
# C5: kernel stats + PMC passes (separate)
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/c5_stats --output-format csv -- python3 bench.py --workload c5 --even-axes --steps 5 --warmup 2 > gpurun_out/r02/c5_bench.json 2> gpurun_out/r02/c5_bench.err; echo c5stats rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02/c5_fetch --output-format csv -- python3 bench.py --workload c5 --even-axes --steps 3 --warmup 1 > /dev/null 2>&1; echo c5fetch rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02/c5_write --output-format csv -- python3 bench.py --workload c5 --even-axes --steps 3 --warmup 1 > /dev/null 2>&1; echo c5write rc=$?
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r02/c5_tcc --output-format csv -- python3 bench.py --workload c5 --even-axes --steps 3 --warmup 1 > /dev/null 2>&1; echo c5tcc rc=$?
find gpurun_out/r02 -name "*.csv" | head -30
