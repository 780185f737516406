#!/bin/bash
# SQ / LDS / TCP counters of eval_slopes2d_kernel on the reference's 100 x 100 x 5 grid (interp_array semantics)
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT
cd $R
export REPS=2 FRESH=1 NDI_SLOPES2D_KERNEL=1
TAG=${TAG:-r06_slopes2d}
for s in ${SHAPES:-2d:100:100:5:f64 2d:100:100:5:f32}; do
  n=$(echo $s | tr ':' '_')
  bash tools/pmc_kernel.sh $R/gpurun_out/${TAG}_counters_$n.txt eval_slopes2d tools/small_shape_once.py $s > /dev/null
done
echo done
