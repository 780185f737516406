set -x
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_ring_and_devices.py -m gpu -x -q -k "ring" > gpurun_out/r02/pytest_alloc.log 2>&1; echo rc=$?; tail -15 gpurun_out/r02/pytest_alloc.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r02/bench2.json 2> gpurun_out/r02/bench2.err; echo bench rc=$?; tail -c 1500 gpurun_out/r02/bench2.err
