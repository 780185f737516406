#!/bin/bash
# Hardware counters of one kernel (name substring $2) under a python command: one rocprofv3 --pmc pass per counter
# group (SQ groups, then FETCH_SIZE / WRITE_SIZE and the L2 hit / miss counts), averages over the matching dispatches.
#   bash tools/pmc_kernel.sh <out-file> <kernel-substring> <script.py> [args...]
# (python3 sits directly after `--`: nothing re-execs under the profiler.)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
OUT=$1; KERN=$2; SCRIPT=$3; shift 3
case "$SCRIPT" in /*) ;; *) SCRIPT="$R/$SCRIPT" ;; esac
set -- "$SCRIPT" "$@"
cd /tmp && export TMPDIR=/tmp
: > "$OUT"
echo "# kernel ~ '$KERN' under: python3 $*" >> "$OUT"
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_LEVEL_LDS" \
         "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  d="/tmp/pmck_${i}"
  rm -rf "$d"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -- python3 "$@" > /dev/null 2> "/tmp/pmck_${i}.err"
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
done
cat "$OUT"
