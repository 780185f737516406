#!/bin/bash
# Round 5 C3 A/B in one call: two-level grouping (tile row, then tile) vs the one-pass record scatter.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_c3_twolevel.jsonl
: > $O
run() { echo "# $*" >> $O; env "$@" python bench.py --workload c3 --path bucketed --steps 30 --warmup 5 2>/dev/null | tail -1 >> $O; }
run NDI_GROUP_TWO_LEVEL=0
run NDI_GROUP_TWO_LEVEL=1
run NDI_GROUP_TWO_LEVEL=0
run NDI_GROUP_TWO_LEVEL=1
python3 - <<'PY'
import json
for l in open("gpurun_out/r05_c3_twolevel.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    d = json.loads(l)
    print({k: d.get(k) for k in ("ms_per_step", "eval_kernel_ms", "stages_ms_per_step")})
PY
