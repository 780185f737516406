#!/bin/bash
# Like tools/pmc_kernel.sh, with the counter groups given in GROUPS_FILE (one rocprofv3 --pmc pass per line).
#   GROUPS_FILE=tools/pmc_groups_tcp.txt bash tools/pmc_groups.sh <out-file> <kernel-substring> <script.py> [args...]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
OUT=$1; KERN=$2; SCRIPT=$3; shift 3
case "$SCRIPT" in /*) ;; *) SCRIPT="$R/$SCRIPT" ;; esac
GF=${GROUPS_FILE:-$R/tools/pmc_groups_tcp.txt}
case "$GF" in /*) ;; *) GF="$R/$GF" ;; esac
set -- "$SCRIPT" "$@"
cd /tmp && export TMPDIR=/tmp
: > "$OUT"
echo "# kernel ~ '$KERN' under: python3 $*" >> "$OUT"
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  d="/tmp/pmcg_${i}"
  rm -rf "$d"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -- python3 "$@" > /dev/null 2> "/tmp/pmcg_${i}.err"
  f=$(find "$d" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$KERN" >> "$OUT" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f'{k}: {sum(v)/len(v):.5g}   (n={len(v)})')
    if not acc:
        print("(no matching dispatches)")
except Exception as e:
    print("(no counters)", e)
PY
done < "$GF"
cat "$OUT"
