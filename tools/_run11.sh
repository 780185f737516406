mkdir -p gpurun_out/r02
for rep in 1 2; do
for n in 2 4 6 7; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 --ring-slots $n 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print($n, d['value'], d['roofline']['kernel_ms'])"
done; done
