#!/bin/bash
# Round 5 C3: slices of the record scatter beyond one workgroup per CU; per-kernel stats and HBM counters of the step.
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
O=$R/gpurun_out/r05_c3_ab2.jsonl
: > $O
run() { echo "# $*" >> $O; env "$@" python bench.py --workload c3 --path bucketed --steps 30 --warmup 5 2>/dev/null | tail -1 >> $O; }
run NDI_GROUP_BLOCKS=256
run NDI_GROUP_BLOCKS=512
run NDI_GROUP_BLOCKS=1024
run NDI_GROUP_BLOCKS=256
cat $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r05_stats_c3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_stats_c3 -- python3 $R/bench.py --workload c3 --steps 10 --warmup 3 > $R/gpurun_out/r05_c3_bench_under_rocprof.json 2>/dev/null
cp "$(find /tmp/r05_stats_c3 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r05_c3_kernel_stats.csv
cut -c1-160 $R/gpurun_out/r05_c3_kernel_stats.csv | head -12
cd $R && bash tools/pmc_hbm_all.sh gpurun_out/r05_c3_hbm_all_kernels.txt bench.py --workload c3 --path bucketed --steps 3 --warmup 1 > /dev/null
cat gpurun_out/r05_c3_hbm_all_kernels.txt
