#!/bin/bash
# per-kernel times of the C3 step under a few settings of the fine pass (records per thread and round, threads)
cd ${GRAFT_REPO_ROOT:-.}
for v in "NDI_GROUP_FINE_R=4" "NDI_GROUP_FINE_R=2" "NDI_GROUP_FINE_R=8" "NDI_GROUP_FINE_R=8 NDI_GROUP_FINE_THREADS=512" "NDI_GROUP_FINE_R=2 NDI_GROUP_FINE_THREADS=512"; do
  echo "## $v"
  env $v TLS=1 bash tools/r05_c3_prof.sh 2>&1 | grep -v "reset_status\|two_level"
done
