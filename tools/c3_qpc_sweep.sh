for q in 2000000 3000000 4000000 5000000 6000000 8000000; do
  for path in gather bucketed; do
    python bench.py --workload c3 --path $path --queries $q --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'queries':$q,'path':'$path','ms_per_step':d.get('ms_per_step'),'eval_kernel_ms':d.get('eval_kernel_ms'),'stages':d.get('stages_ms_per_step')}))"
  done
done
