mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest15.log 2>&1; echo pytest rc=$?; tail -6 gpurun_out/r02/pytest15.log
for lut in 0 1; do
echo LUT=$lut
NDI_LOCATE_LUT=$lut python bench.py --workload c5 --steps 10 --warmup 3 2>/dev/null
NDI_LOCATE_LUT=$lut python bench.py --workload c3 --steps 10 --warmup 3 2>/dev/null
NDI_LOCATE_LUT=$lut python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-gather-leg --placement-probe 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('target', d['value'], d['roofline']['kernel_ms'], d['stages_ms_per_step'], d['check'])"
done
