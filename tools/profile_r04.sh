#!/bin/bash
# Round-4 profile set.  Everything lands in gpurun_out/ (scratch); the summaries worth keeping are copied to profiles/.
#   1. the default bench line (driver's flags)
#   2. rocprofv3 per-kernel stats of the Target workload, bucketed (shipped) and gather formulation
#   3. FETCH_SIZE / WRITE_SIZE of both formulations (separate --pmc passes)
#   4. one short-row shape (f64, 32 lanes, 4 GB of output): per-kernel stats + FETCH_SIZE / WRITE_SIZE
#   5. C3 / C5-share per-kernel stats
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T="--no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 --no-secondary --no-pmc"
cd $R && python bench.py --steps 20 --warmup 5 > $O/r04_bench_default.json 2> $O/r04_bench_default.err
cd /tmp && export TMPDIR=/tmp
for path in auto gather; do
  tag=$([ $path = auto ] && echo target || echo target_gather)
  steps=$([ $path = auto ] && echo 10 || echo 3)
  rm -rf /tmp/r04_stats_$path
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_stats_$path -- python3 $R/bench.py --steps $steps --warmup 1 $T --path $path > $O/r04_${tag}_bench_under_rocprof.json 2>/dev/null
  cp "$(find /tmp/r04_stats_$path -name '*kernel_stats.csv' | head -1)" $O/r04_${tag}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "/tmp/r04_pmc_${path}_${c}"
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "/tmp/r04_pmc_${path}_${c}" -- python3 $R/bench.py --steps 2 --warmup 1 $T --path $path > /dev/null 2>&1
    python3 - "$(find /tmp/r04_pmc_${path}_${c} -name '*counter_collection.csv' | head -1)" $c $path >> $O/r04_pmc_hbm_counters.txt <<'PY'
import csv, sys, collections
f, c, path = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c:
        acc[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f'{path},{c},{k},launches={len(v)},mean_KiB={sum(v)/len(v):.1f}')
PY
  done
done
# short rows: f64 x 32 lanes, the shipped (auto) formulation
S="$R/tools/short_rows_sweep.py --quick --lanes 32 --dtypes float64 --only auto --reps 5"
rm -rf /tmp/r04_stats_short
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_stats_short -- python3 $S > $O/r04_short_rows_f64_l32_under_rocprof.jsonl 2>/dev/null
cp "$(find /tmp/r04_stats_short -name '*kernel_stats.csv' | head -1)" $O/r04_short_rows_f64_l32_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "/tmp/r04_pmc_short_${c}"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "/tmp/r04_pmc_short_${c}" -- python3 $S > /dev/null 2>&1
  python3 - "$(find /tmp/r04_pmc_short_${c} -name '*counter_collection.csv' | head -1)" $c >> $O/r04_short_rows_f64_l32_pmc.txt <<'PY'
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
for w in c3 c5; do
  rm -rf /tmp/r04_stats_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 > $O/r04_${w}_bench_under_rocprof.json 2>/dev/null
  cp "$(find /tmp/r04_stats_$w -name '*kernel_stats.csv' | head -1)" $O/r04_${w}_kernel_stats.csv
done
echo profile_r04 done
