#!/bin/bash
# C3 / C5-share: gather order vs tile-grouped order (tile histogram fused into the search, compact records, tiles
# staged in LDS), and the queries-per-cell crossover on the C3 grid.  One JSON object per line.
for wl in c3 c5; do
  for path in gather bucketed auto; do
    python bench.py --workload $wl --path $path --steps 20 --warmup 3 2>/dev/null | tail -1
  done
done
for q in 1000000 2000000 4000000 6000000 20000000 40000000; do
  for path in gather bucketed; do
    python bench.py --workload c3 --path $path --queries $q --steps 20 --warmup 3 2>/dev/null | tail -1
  done
done
