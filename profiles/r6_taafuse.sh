#!/bin/bash
# round 6: TemporalBlendWithClamp inside the trace launch (ycge::TaaFuse) and the schedule built a frame ahead (Knobs::sync_defer), synchronous frames, same call
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_taafuse_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6_taafuse_pytest.log
for cfg in 4 3 2 1; do
  echo "== config $cfg"
  for spec in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do set -- $spec
    YCGE_TAA_FUSE=$1 YCGE_SYNC_DEFER=$2 timeout 200 python profiles/sync_ms.py $cfg 300 "fuse=$1 defer=$2" 2>&1 | tail -1; done
done
