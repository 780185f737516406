#!/bin/bash
# eval_staged2d_kernel vs the query-order kernel on the reference's 100 x 100 x 5 grid (and neighbours), 2e7 queries
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_staged2d_rates.txt
: > $O
for s in 2d:100:100:5:f64 2d:100:100:5:f32 2d:100:100:8:f64 2d:300:300:5:f64 2d:100:100:3:f64 2d:100:100:16:f32; do
for v in "NDI_STAGED2D_KERNEL=0" "NDI_STAGED2D_KERNEL=1" "NDI_STAGED2D_KERNEL=1 NDI_STAGED2D_TB=64" "NDI_STAGED2D_KERNEL=1 NDI_STAGED2D_TB=256" "NDI_STAGED2D_KERNEL=1 FRESH=1"; do
  echo "# $v $s" >> $O
  env $v NDI_TRACE_PLAN=1 python3 tools/small_shape_once.py $s 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -2 >> $O
done
done
cat $O
