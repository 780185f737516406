for cfg in "1000000 4" "2500000 2"; do
set -- $cfg
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-gather-leg --placement-probe 0 --chunk $1 --ring-slots $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('NT loads: chunk $1 slots $2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'], d['check'])"
done
