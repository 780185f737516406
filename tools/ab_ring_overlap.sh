#!/bin/bash
# A/B of the ring pipeline (side-stream locate + group vs everything on one stream), alternating, same box.
for i in 1 2 3; do
  for ov in 1 0; do
    NDI_RING_OVERLAP=$ov python bench.py --steps 20 --warmup 3 --no-cpu-baseline --placement-probe 0 --no-gather-leg --no-check --no-secondary --no-pmc 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print(json.dumps({'overlap': $ov, 'ms_per_step': d['ms_per_step'], 'eval_ms_per_step': s['eval'], 'step_minus_eval': round(d['ms_per_step']-s['eval'],4), 'kernel_ms': d['roofline']['kernel_ms'], 'Mpoints_s': d['value']}))"
  done
done
