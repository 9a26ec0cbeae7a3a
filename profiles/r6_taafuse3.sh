#!/bin/bash
# round 6: fused TAA, window staged in LDS: parity of the TAA tests, then synchronous frames fused / separate x schedule deferred / not
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r6_taafuse3_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6_taafuse3_pytest.log
for cfg in 4 3 2 1; do
  echo "== config $cfg"
  for spec in "0 0" "1 0" "1 1" "0 1" "0 0" "1 1"; do set -- $spec
    YCGE_TAA_FUSE=$1 YCGE_SYNC_DEFER=$2 timeout 200 python profiles/sync_ms.py $cfg 300 "fuse=$1 defer=$2" 2>&1 | tail -1; done
done
