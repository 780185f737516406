#!/bin/bash
# eval_slopes2d_kernel (slope records) vs the kernels AUTO took in round 5 on the reference's 100 x 100 x 5 grid and neighbours
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_slopes2d_rates.txt
: > $O
SHAPES=${SHAPES:-"2d:100:100:5:f64 2d:100:100:5:f32 2d:100:100:2:f64 2d:100:100:3:f64 2d:100:100:7:f64 2d:100:100:8:f64 2d:100:100:3:f32 2d:100:100:8:f32 2d:100:100:16:f32 2d:300:300:5:f64 2d:1000:1000:5:f64 2d:1000:1000:4:f32 2d:1000:1000:1:f64 2d:1000:1000:1:f32"}
for s in $SHAPES; do
for v in "NDI_SLOPES2D_KERNEL=0" "NDI_SLOPES2D_KERNEL=1" "NDI_SLOPES2D_KERNEL=1 FRESH=1" ${EXTRA:+"$EXTRA"}; do
  echo "# $v $s" >> $O
  env $v NDI_TRACE_PLAN=1 python3 tools/small_shape_once.py $s 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -2 >> $O
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_slopes2d_rates.txt"):
    l = l.strip()
    if l.startswith("#"): print(l, end=" -> ")
    elif "{" in l:
        d = json.loads(l[l.index("{"):]); print(d["ms"], d["Gqueries_s"])
    elif "plan" in l: print(l.split("]")[1][:40], end=" ")
PY
