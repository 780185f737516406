#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_c3_split4.jsonl
: > $O
run() { echo "# $*" >> $O; env "$@" NDI_TRACE_PLAN=1 python bench.py --workload c3 --path bucketed --steps 30 --warmup 5 2>gpurun_out/split4.err | tail -1 >> $O; grep "tiles ts" gpurun_out/split4.err | sort | uniq -c | head -2 >> $O; }
run NDI_TILE_SPLIT=2
run NDI_TILE_SPLIT=4
run NDI_TILE_SPLIT=2
run NDI_TILE_SPLIT=4
python3 - <<'PY'
import json
for l in open("gpurun_out/r05_c3_split4.jsonl"):
    if not l.startswith("{"): print(l.strip()); continue
    d = json.loads(l)
    print({k: d.get(k) for k in ("ms_per_step", "eval_kernel_ms", "stages_ms_per_step")})
PY
