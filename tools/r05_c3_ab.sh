#!/bin/bash
# Round 5 C3 A/B: channel-split tile kernel on / off, slices of the record scatter.  One JSON line per run.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_c3_ab.jsonl
: > $O
run() { echo "# $*" >> $O; env "$@" python bench.py --workload c3 --path bucketed --steps 30 --warmup 5 2>/dev/null | tail -1 >> $O; }
run NDI_TILE_SPLIT=1
run NDI_TILE_SPLIT=0
run NDI_TILE_SPLIT=1 NDI_GROUP_BLOCKS=128
run NDI_TILE_SPLIT=1 NDI_GROUP_BLOCKS=64
run NDI_TILE_SPLIT=1 NDI_TILE_CHUNK=4096
run NDI_TILE_SPLIT=1 NDI_TILE_CHUNK=16384
run NDI_TILE_SPLIT=1
cat $O
