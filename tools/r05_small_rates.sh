#!/bin/bash
# Round 5: rate + plan of the reference's bench shapes at large Q (tools/small_shape_once.py), lanes kernels on / off.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_small_rates.txt
: > $O
for on in -1 fresh 0; do
for s in 1d:100:1:f64 1d:100:1:f32 1d:1024:1:f64 1d:100:1:f64:linear 1d:100:5:f64 1d:100:5:f32 1d:100:5:f64:linear 1d:100:2:f64 1d:100:8:f64 1d:100:8:f32 1d:100:16:f32 2d:100:100:1:f64 2d:100:100:1:f32 2d:100:100:5:f64 2d:100:100:5:f32 2d:64:64:2:f64; do
  echo "# lanes=$on $s" >> $O
  k=$on; f=0; if [ $on = fresh ]; then k=-1; f=1; fi
  FRESH=$f NDI_LANES_KERNEL=$k NDI_LANES2D_KERNEL=$k NDI_TRACE_PLAN=1 python3 tools/small_shape_once.py $s 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -3 >> $O
done
done
cat $O
