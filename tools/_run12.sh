mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest12.log 2>&1; echo pytest rc=$?; tail -4 gpurun_out/r02/pytest12.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; echo bench rc=$?
python bench.py --gpus 2 --backend gloo --device-override 0 --steps 3 --warmup 1 --placement-probe 0 --no-gather-leg --ring-slots 2 > gpurun_out/r02/bench_2ranks_rehearsal.json 2> gpurun_out/r02/bench_2ranks.err; echo bench2 rc=$?; tail -3 gpurun_out/r02/bench_2ranks.err
