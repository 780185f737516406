#!/bin/bash
# tuning build: eval_slopes2d_kernel with parts switched off (results meaningless) -- where the time goes
cd ${GRAFT_REPO_ROOT:-.}
export NDI_LIB=$PWD/ndarray-interp_amd/libndinterp_hip_tune.so NDI_SLOPES2D_KERNEL=1 FRESH=1
for s in ${SHAPES:-2d:100:100:5:f64 2d:100:100:5:f32}; do
for d in ${DEBUGS:-0 1 2 4 8 3 5 6 7 15}; do
  echo -n "debug=$d $s: "
  NDI_SLOPES2D_DEBUG=$d python3 tools/small_shape_once.py $s 2>&1 | grep -o '"ms": [0-9.]*'
done
done
