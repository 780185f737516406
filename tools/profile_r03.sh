#!/bin/bash
# Round-3 profile set: the default bench line, rocprofv3 per-kernel stats of the same workload, PMC traffic.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R && python bench.py --steps 20 --warmup 5 > $O/r03_bench_default.json 2> $O/r03_bench_default.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r03_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 --no-secondary --no-pmc > $O/r03_target_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/r03_stats -name "*kernel_stats.csv" | head -1) $O/r03_target_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/r03_pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/r03_pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 --no-secondary --no-pmc > /dev/null 2>&1
  python3 - "$(find /tmp/r03_pmc_$c -name '*counter_collection.csv' | head -1)" $c >> $O/r03_pmc_hbm_counters.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1:3]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c:
        acc[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f'{c},{k},launches={len(v)},mean_KiB={sum(v)/len(v):.1f}')
PY
done
rm -rf /tmp/r03_stats_sec
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_stats_sec -- python3 $R/bench.py --workload c3 --steps 10 --warmup 3 > $O/r03_c3_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/r03_stats_sec -name "*kernel_stats.csv" | head -1) $O/r03_c3_kernel_stats.csv
rm -rf /tmp/r03_stats_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_stats_c5 -- python3 $R/bench.py --workload c5 --steps 10 --warmup 3 > $O/r03_c5_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/r03_stats_c5 -name "*kernel_stats.csv" | head -1) $O/r03_c5_share_kernel_stats.csv
