set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest5.log 2>&1; echo rc=$?; tail -5 gpurun_out/r02/pytest5.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench5.json 2> gpurun_out/r02/bench5.err; echo bench rc=$?; tail -c 500 gpurun_out/r02/bench5.err
