#!/bin/bash
# Round 5: SQ / LDS / TCP counters of the query-per-lane kernels on the reference's bench shapes and of the channel-split
# tile kernel at C3 (tools/pmc_kernel.sh: one rocprofv3 --pmc pass per counter group).
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_counters2
mkdir -p $O
cd $R
export REPS=2
bash tools/pmc_kernel.sh $O/scalar_1d_100_f64.txt eval_scalar_kernel tools/small_shape_once.py 1d:100:1:f64 > /dev/null
bash tools/pmc_kernel.sh $O/scalar_1d_100_f64_linear.txt eval_scalar_kernel tools/small_shape_once.py 1d:100:1:f64:linear > /dev/null
bash tools/pmc_kernel.sh $O/lanes_1d_100_5_f64.txt eval_lanes_kernel tools/small_shape_once.py 1d:100:5:f64 > /dev/null
bash tools/pmc_kernel.sh $O/scalar2d_100_f64.txt eval_scalar2d_kernel tools/small_shape_once.py 2d:100:100:1:f64 > /dev/null
bash tools/pmc_kernel.sh $O/c3_tiles_split.txt eval_bilinear_tiles_kernel bench.py --workload c3 --path bucketed --steps 3 --warmup 1 > /dev/null
echo done
