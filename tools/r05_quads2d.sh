#!/bin/bash
# eval_quads2d_kernel (cell-quad layout) vs the kernels it replaces on the reference's 100 x 100 x 5 grid and neighbours
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_quads2d_rates.txt
: > $O
for s in 2d:100:100:5:f64 2d:100:100:5:f32 2d:100:100:8:f64 2d:300:300:5:f64 2d:100:100:3:f64 2d:100:100:16:f32 2d:100:100:2:f64 2d:1000:1000:5:f64 2d:1000:1000:4:f32; do
for v in "NDI_QUADS2D_KERNEL=0" "NDI_QUADS2D_KERNEL=1" "NDI_QUADS2D_KERNEL=1 NDI_QUADS2D_TB=128" "NDI_QUADS2D_KERNEL=1 NDI_QUADS2D_TB=512" "NDI_QUADS2D_KERNEL=1 FRESH=1"; do
  echo "# $v $s" >> $O
  env $v NDI_TRACE_PLAN=1 python3 tools/small_shape_once.py $s 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -2 >> $O
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05_quads2d_rates.txt"):
    l = l.strip()
    if l.startswith("#"): print(l, end=" -> ")
    elif "{" in l:
        d = json.loads(l[l.index("{"):]); print(d["ms"], d["Gqueries_s"])
PY
