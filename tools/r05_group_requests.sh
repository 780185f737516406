#!/bin/bash
# L2 request counts (TCC_REQ / TCC write requests) of the two grouping passes, rounds sorted in LDS vs written directly
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r05_group_requests.txt
: > $O
for v in "NDI_GROUP_COARSE_SORT=1 NDI_GROUP_FINE_SORT=1" "NDI_GROUP_COARSE_SORT=0 NDI_GROUP_FINE_SORT=0"; do
  echo "## $v" >> $O
  for c in "TCC_REQ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
    d=/tmp/grq_$RANDOM; rm -rf $d
    env $v rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py --workload c3 --path bucketed --steps 3 --warmup 1 > /dev/null 2>&1
    python3 - "$(find $d -name '*counter_collection.csv' | head -1)" >> $O <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0]
        if "scatter2d" in k: acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()): print(f"  {c:22s} {sum(v)/len(v):14.1f}  n={len(v)}  {k}")
except Exception as e: print("  (no counters)", e)
PY
  done
done
cat $O
