#!/bin/bash
# HBM bytes of every kernel of the C3 step (two-level grouping as shipped)
cd ${GRAFT_REPO_ROOT:-.}
bash tools/pmc_hbm_all.sh gpurun_out/r05_c3_hbm_twolevel.txt bench.py --workload c3 --path bucketed --steps 3 --warmup 1 > /dev/null
cat gpurun_out/r05_c3_hbm_twolevel.txt
