#!/bin/bash
# round 6: where the time of the fused TAA goes (timing-only variants: written-through stores alone; + counters; everything); the pinning reproducer
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
L=$REPO/yetanotherconsolegameengine_amd/lib
for cfg in 4 2; do
  echo "== config $cfg"
  YCGE_TAA_FUSE=0 timeout 200 python profiles/sync_ms.py $cfg 200 "separate k_taa" 2>&1 | tail -1
  YCGE_LIB=$L/var_fuse_dbg1.so timeout 200 python profiles/sync_ms.py $cfg 200 "sc1 stores only (no TAA)" 2>&1 | tail -1
  YCGE_LIB=$L/var_fuse_dbg2.so timeout 200 python profiles/sync_ms.py $cfg 200 "sc1 stores + counters (no TAA)" 2>&1 | tail -1
  timeout 200 python profiles/sync_ms.py $cfg 200 "fused" 2>&1 | tail -1
done
echo "== pinfault reproducer"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 profiles/micro/pinfault.hip -o /tmp/pinfault 2>&1 | tail -2
for m in 0 1 2 3; do AMD_SERIALIZE_KERNEL=3 timeout 300 /tmp/pinfault $m 10000 2>&1 | tail -2; echo "mode $m rc=$?"; done
