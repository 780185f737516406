#!/bin/bash
# tile kernel with / without its stores (tuning build: NDI_FUSED_DEBUG bit 2 = no stores; results meaningless)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for dbg in 0 4; do
  echo -n "NDI_FUSED_DEBUG=$dbg: "
  NDI_LIB=libndinterp_hip_tune.so NDI_FUSED_DEBUG=$dbg python bench.py --workload c3 --path bucketed --steps 30 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['eval_kernel_ms'])"
done
done
