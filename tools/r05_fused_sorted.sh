#!/bin/bash
# eval_fused_sorted_kernel (queries of a workgroup round ordered by interval in LDS) vs the query-order kernel:
# 1-D CubicSpline / Linear on 1024 knots, 4 GB of output per call
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_fused_sorted_rates.txt
: > $O
for s in 1d:1024:16:f64 1d:1024:32:f64 1d:1024:64:f64 1d:1024:128:f64 1d:1024:32:f32 1d:1024:64:f32 1d:1024:128:f32 1d:1024:32:f64:linear 1d:4096:32:f64 1d:256:32:f64; do
  L=$(echo $s | cut -d: -f3); dt=$(echo $s | cut -d: -f4); el=8; [ $dt = f32 ] && el=4
  Q=$((4000000000 / (L * el)))
for v in "NDI_FUSED_SORTED=0" "NDI_FUSED_SORTED=1" "NDI_FUSED_SORTED=0 FRESH=1" "NDI_FUSED_SORTED=1 FRESH=1"; do
  echo "# $v $s" >> $O
  env $v Q=$Q NDI_TRACE_PLAN=1 python3 tools/small_shape_once.py $s 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -2 >> $O
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05_fused_sorted_rates.txt"):
    l = l.strip()
    if l.startswith("#"): print(l, end=" -> ")
    elif "{" in l:
        d = json.loads(l[l.index("{"):]); print(d["ms"], d["out_TBps"])
PY
