#!/bin/bash
# Explicit store pacing in eval_bucketed_kernel (A/B builds libndinterp_hip_paceN.so: -DNDI_BUCK_PACE=N) vs the shipped
# library: kernel ms of the long-row legs, alternating.
cd ${GRAFT_REPO_ROOT:-.}
T="--no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 --no-secondary --no-pmc"
O=gpurun_out/r05_pace.txt
: > $O
for rep in 1 2; do
for lib in libndinterp_hip.so libndinterp_hip_pace4.so libndinterp_hip_pace8.so libndinterp_hip_pace12.so libndinterp_hip_pace20.so; do
  for w in c2-f32 c2-linear c2; do
    r=$(NDI_LIB=$lib python bench.py --workload $w --steps 10 --warmup 2 $T 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = [v for v in d.values() if isinstance(v, dict) and 'kernel_ms' in v]
print(d.get('eval_kernel_ms') or d.get('roofline', {}).get('kernel_ms') or (k[0]['kernel_ms'] if k else d))")
    echo "$lib $w kernel_ms=$r" | tee -a $O
  done
  r=$(NDI_LIB=$lib python bench.py --steps 10 --warmup 2 $T 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['ms_per_step'])")
  echo "$lib target kernel_ms,ms_per_step=$r" | tee -a $O
done
done
