#!/bin/bash
# Round-6 profile set.  Everything lands in gpurun_out/ (scratch); the summaries worth keeping are copied to profiles/.
#   1. the default bench line (driver's flags)
#   2. rocprofv3 per-kernel stats of the Target workload, bucketed (shipped) and gather formulation
#   3. FETCH_SIZE / WRITE_SIZE of both formulations and of the C2-shaped Linear f64 / CubicSpline f32 launches
#      (separate --pmc passes; gfx950: read bytes = 2 x FETCH_SIZE)
#   4. C3 / C5-share per-kernel stats; HBM counters of every kernel of the C3 step
#   5. per-kernel stats of C2 itself (BASELINE configs[1]) and of the C2-shaped Linear f64 / CubicSpline f32 launches
#   6. CubicSpline::build at C2: per-kernel stats and FETCH / WRITE of spline_build_wide_kernel vs the serial kernels
# Progress lines go to stdout (a silent run is taken to be hung after 7 minutes).
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T="--no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 --no-secondary --no-pmc"
cd $R && python bench.py --steps 20 --warmup 5 > $O/r06_bench_default.json 2> $O/r06_bench_default.err
cd /tmp && export TMPDIR=/tmp
pmc() {   # pmc <tag> <counter> <out-file> <bench args...>
  local tag=$1 c=$2 out=$3; shift 3
  rm -rf "/tmp/r06_pmc_${tag}_${c}"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "/tmp/r06_pmc_${tag}_${c}" -- python3 $R/bench.py "$@" > /dev/null 2>&1
  python3 - "$(find /tmp/r06_pmc_${tag}_${c} -name '*counter_collection.csv' | head -1)" $c $tag >> $out <<'PY'
import csv, sys, collections
f, c, tag = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c:
        acc[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f'{tag},{c},{k},launches={len(v)},mean_KiB={sum(v)/len(v):.1f}')
PY
}
: > $O/r06_pmc_hbm_counters.txt
echo '[profile_r06] default bench done'
for path in auto gather; do
  tag=$([ $path = auto ] && echo target || echo target_gather)
  steps=$([ $path = auto ] && echo 10 || echo 3)
  rm -rf /tmp/r06_stats_$path
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_stats_$path -- python3 $R/bench.py --steps $steps --warmup 1 $T --path $path > $O/r06_${tag}_bench_under_rocprof.json 2>/dev/null
  cp "$(find /tmp/r06_stats_$path -name '*kernel_stats.csv' | head -1)" $O/r06_${tag}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do pmc $tag $c $O/r06_pmc_hbm_counters.txt --steps 2 --warmup 1 $T --path $path; done
  echo "[profile_r06] target $path done"
done
for w in c2 c2-linear c2-f32; do
  rm -rf /tmp/r06_stats_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_stats_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 > $O/r06_${w}_bench_under_rocprof.json 2>/dev/null
  cp "$(find /tmp/r06_stats_$w -name '*kernel_stats.csv' | head -1)" $O/r06_${w}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do pmc $w $c $O/r06_pmc_hbm_counters.txt --workload $w --steps 3 --warmup 1; done
done
echo '[profile_r06] c2 legs done'
for w in c3 c5; do
  rm -rf /tmp/r06_stats_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 > $O/r06_${w}_bench_under_rocprof.json 2>/dev/null
  cp "$(find /tmp/r06_stats_$w -name '*kernel_stats.csv' | head -1)" $O/r06_${w}_kernel_stats.csv
done
echo '[profile_r06] c3 c5 stats done'
cd $R && bash tools/pmc_hbm_all.sh gpurun_out/r06_c3_hbm_all_kernels.txt bench.py --workload c3 --path bucketed --steps 3 --warmup 1 > /dev/null
echo "[profile_r06] build kernels"
cd /tmp
for v in wide serial; do
  rm -rf /tmp/r06_stats_build_$v
  if [ $v = serial ]; then export NDI_SPLINE_WIDE=0; else unset NDI_SPLINE_WIDE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_stats_build_$v -- python3 $R/tools/r06_build_only.py > /dev/null 2>&1
  cp "$(find /tmp/r06_stats_build_$v -name '*kernel_stats.csv' | head -1)" $O/r06_build_${v}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/r06_pmc_build_${v}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/r06_pmc_build_${v}_$c -- python3 $R/tools/r06_build_only.py > /dev/null 2>&1
    python3 - "$(find /tmp/r06_pmc_build_${v}_$c -name '*counter_collection.csv' | head -1)" $c build_$v >> $O/r06_pmc_hbm_counters.txt <<'PY'
import csv, sys, collections
f, c, tag = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c and "spline" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f'{tag},{c},{k},launches={len(v)},mean_KiB={sum(v)/len(v):.1f}')
PY
  done
  echo "[profile_r06] build $v done"
done
unset NDI_SPLINE_WIDE
echo profile_r06 done
