#!/bin/bash
# per-kernel times of the C3 step (rocprofv3 --kernel-trace --stats), two-level grouping off / on
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for tl in ${TLS:-0 1}; do
  export NDI_GROUP_TWO_LEVEL=$tl
  rm -rf /tmp/prof_tl$tl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tl$tl -- python3 bench.py --workload c3 --path bucketed --steps 10 --warmup 2 > /dev/null 2>&1
  f=$(find /tmp/prof_tl$tl -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $tl <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("two_level=%s" % sys.argv[2])
for r in rows:
    n = r["Name"]
    if "ndi::" in n: print("  %-60s calls %4s avg %9.1f us" % (n.split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
