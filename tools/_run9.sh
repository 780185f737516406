set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q -k "bilinear or 2d or c3 or c5 or fuzz" > gpurun_out/r02/pytest9.log 2>&1; echo rc=$?; tail -3 gpurun_out/r02/pytest9.log
for k in 0 1; do
NDI_BILINEAR_KLDS=$k python bench.py --workload c5 --steps 10 --warmup 3 2>>gpurun_out/r02/c5.err
NDI_BILINEAR_KLDS=$k python bench.py --workload c3 --steps 10 --warmup 3 2>>gpurun_out/r02/c5.err
done
