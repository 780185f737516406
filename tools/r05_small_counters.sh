#!/bin/bash
# Round 5: plan + rate + SQ/LDS/TCP counters of the reference's bench shapes at large Q (tools/small_shape_once.py).
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_small
mkdir -p $O
cd $R
for s in 1d:100:1:f64 1d:100:1:f32 1d:1024:1:f64 1d:100:5:f64 1d:100:5:f32 2d:100:100:1:f64 2d:100:100:1:f32 2d:100:100:5:f64 2d:100:100:5:f32; do
  NDI_TRACE_PLAN=1 python3 tools/small_shape_once.py $s 2>&1 | sort | uniq -c | sort -rn | head -4
done > $O/rates.txt 2>&1
cat $O/rates.txt
export REPS=2
bash tools/pmc_kernel.sh $O/c_1d_100_1_f64.txt eval_fused_kernel tools/small_shape_once.py 1d:100:1:f64 > /dev/null
bash tools/pmc_kernel.sh $O/c_1d_100_5_f64.txt eval_fused_kernel tools/small_shape_once.py 1d:100:5:f64 > /dev/null
bash tools/pmc_kernel.sh $O/c_2d_100_1_f64.txt eval_ tools/small_shape_once.py 2d:100:100:1:f64 > /dev/null
bash tools/pmc_kernel.sh $O/c_2d_100_5_f64.txt eval_ tools/small_shape_once.py 2d:100:100:5:f64 > /dev/null
echo done
