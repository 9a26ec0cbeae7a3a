#!/bin/bash
# round 6: the GPU suite on the build of the hour + the heaviest blocks in TWO parts (YCGE_SPLIT_TOP_LG=1), same call
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r6_suite_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r6_suite_pytest.log
for cfg in 4 3; do
  echo "== config $cfg"
  for spec in "32 2" "64 1" "128 1" "256 1" "512 1" "32 2"; do set -- $spec
    YCGE_SPLIT_TOP=$1 YCGE_SPLIT_TOP_LG=$2 timeout 200 python profiles/sync_ms.py $cfg 300 "split_top=$1 parts=2^$2" 2>&1 | tail -1; done
done
