set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest6.log 2>&1; echo rc=$?; tail -3 gpurun_out/r02/pytest6.log
for run in 1 2 4 8 16; do
NDI_BUCKETED_RUN=$run python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gather-leg --placement-probe 0 > gpurun_out/r02/bench6_run$run.json 2> gpurun_out/r02/bench6.err; echo bench rc=$?
done
