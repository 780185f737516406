set -x
mkdir -p gpurun_out/r02
rocm-smi --showmemorypartition --showcomputepartition > gpurun_out/r02/partition.txt 2>&1
rocminfo 2>/dev/null | grep -E "Marketing|Pool|Size|Granule|Segment" | head -60 >> gpurun_out/r02/partition.txt
timeout -k 10 400 ./tools/placement_probe2 2 96 > gpurun_out/r02/placement_probe2_2g.jsonl 2>&1; echo rc=$?
timeout -k 10 400 ./tools/placement_probe2 1 128 > gpurun_out/r02/placement_probe2_1g.jsonl 2>&1; echo rc=$?
