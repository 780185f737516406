#!/bin/bash
# SQ counters of the C3 tile-grouped evaluation kernel (one group of counters per pass)
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_LEVEL_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf /tmp/pt_$i
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pt_$i -- python3 $R/bench.py --workload c3 --path bucketed --steps 2 --warmup 1 > /dev/null 2>&1
  f=$(find /tmp/pt_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "tiles_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f'{k}: {sum(v)/len(v):.4g}')
except Exception as e:
    print("(no counters)", e)
PY
done
