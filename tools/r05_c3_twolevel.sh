#!/bin/bash
# Round 5 C3 A/B: two-level grouping (tile row, then tile) vs the one-pass record scatter.  One JSON line per run,
# then a rocprofv3 kernel-trace of both for the per-kernel times.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_c3_twolevel.jsonl
: > $O
run() { echo "# $*" >> $O; env "$@" python bench.py --workload c3 --path bucketed --steps 30 --warmup 5 2>/dev/null | tail -1 >> $O; }
run NDI_GROUP_TWO_LEVEL=0
run NDI_GROUP_TWO_LEVEL=1
run NDI_GROUP_TWO_LEVEL=1 NDI_GROUP_FINE_THREADS=512
run NDI_GROUP_TWO_LEVEL=1 NDI_GROUP_FINE_THREADS=256
run NDI_GROUP_TWO_LEVEL=1 NDI_GROUP_BLOCKS=512
run NDI_GROUP_TWO_LEVEL=0
python - <<'PY'
import json
for l in open("gpurun_out/r05_c3_twolevel.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    try:
        d = json.loads(l)
        print({k: d.get(k) for k in ("ms_per_step",)}, {k: d["roofline"].get(k) for k in ("kernel_ms",)}, d.get("stages_ms_per_step"))
    except Exception as e: print("bad line", e, l[:200])
PY
export TMPDIR=/tmp
for tl in 0 1; do
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
