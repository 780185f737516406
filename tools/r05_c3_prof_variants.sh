#!/bin/bash
# per-kernel times of the C3 step (rocprofv3 --kernel-trace --stats) under the A/B settings of the grouping passes:
#   bash tools/r05_c3_prof_variants.sh "NDI_GROUP_FINE_SORT=0" "NDI_GROUP_FINE_SORT=1" "NDI_GROUP_FINE_PARTS=8" ...
# (one setting per argument; default: sorted vs direct rounds of both passes)
cd ${GRAFT_REPO_ROOT:-.}
[ $# -eq 0 ] && set -- "NDI_GROUP_COARSE_SORT=1 NDI_GROUP_FINE_SORT=1" "NDI_GROUP_COARSE_SORT=0 NDI_GROUP_FINE_SORT=0"
for v in "$@"; do
  echo "## $v"
  env $v TLS=1 bash tools/r05_c3_prof.sh 2>&1 | grep -v "reset_status\|two_level"
done
