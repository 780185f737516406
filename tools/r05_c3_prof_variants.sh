#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in "NDI_GROUP_COARSE_SORT=0 NDI_GROUP_FINE_SORT=1" "NDI_GROUP_COARSE_SORT=1 NDI_GROUP_FINE_SORT=1" "NDI_GROUP_COARSE_SORT=0 NDI_GROUP_FINE_SORT=1" "NDI_GROUP_COARSE_SORT=1 NDI_GROUP_FINE_SORT=1"; do
  echo "## $v"
  env $v TLS=1 bash tools/r05_c3_prof.sh 2>&1 | grep -v "reset_status\|two_level"
done
