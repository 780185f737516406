#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the C3 bilinear evaluation, gather order vs tile-grouped order (separate --pmc passes).
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
for path in gather bucketed; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "/tmp/pmc_${path}_${c}"
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${path}_$c -- python3 $R/bench.py --workload c3 --path $path --steps 3 --warmup 1 > /dev/null 2>&1
    f=$(find /tmp/pmc_${path}_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$path" "$c" <<'PY'
import csv, sys, collections
f, path, c = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c:
        acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "bilinear" in k or "scatter2d" in k or "locate2" in k:
        print(f'{{"path": "{path}", "counter": "{c}", "kernel": "{k}", "launches": {len(v)}, "mean": {sum(v)/len(v):.1f}}}')
PY
  done
done
