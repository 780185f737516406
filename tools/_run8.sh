set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q -k "bilinear or 2d or c3 or c5 or fuzz or raw_c_abi" > gpurun_out/r02/pytest8.log 2>&1; echo rc=$?; tail -3 gpurun_out/r02/pytest8.log
python bench.py --workload c5 --even-axes --steps 10 --warmup 3 > gpurun_out/r02/c5_even.json 2>gpurun_out/r02/c5.err; echo rc=$?
python bench.py --workload c5 --steps 10 --warmup 3 > gpurun_out/r02/c5_rand.json 2>>gpurun_out/r02/c5.err; echo rc=$?
python bench.py --workload c3 --steps 10 --warmup 3 > gpurun_out/r02/c3_rand.json 2>>gpurun_out/r02/c5.err; echo rc=$?
python bench.py --workload c3 --even-axes --steps 10 --warmup 3 > gpurun_out/r02/c3_even.json 2>>gpurun_out/r02/c5.err; echo rc=$?
cat gpurun_out/r02/c5_even.json gpurun_out/r02/c5_rand.json gpurun_out/r02/c3_rand.json gpurun_out/r02/c3_even.json
