#!/bin/bash
# HBM traffic of EVERY kernel of a python command: FETCH_SIZE and WRITE_SIZE (KiB; gfx950: reads = 2 x FETCH_SIZE),
# one rocprofv3 --pmc pass each, averaged per kernel name.
#   bash tools/pmc_hbm_all.sh <out-file> <script.py> [args...]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
OUT=$1; SCRIPT=$2; shift 2
case "$SCRIPT" in /*) ;; *) SCRIPT="$R/$SCRIPT" ;; esac
case "$OUT" in /*) ;; *) OUT="$R/$OUT" ;; esac
set -- "$SCRIPT" "$@"
cd /tmp && export TMPDIR=/tmp
echo "# python3 $*" > "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  d="/tmp/pmch_$c"
  rm -rf "$d"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -- python3 "$@" > /dev/null 2> "/tmp/pmch_$c.err"
  f=$(find "$d" -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> "$OUT" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f'{c:11s} {sum(v)/len(v)/1024:12.2f} MiB  n={len(v):4d}  {k}')
PY
done
cat "$OUT"
