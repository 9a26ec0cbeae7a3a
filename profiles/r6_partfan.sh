#!/bin/bash
# round 6: query fan-out INSIDE the wavefront for the parts of split blocks (trace_block MODE 3) against round 5's kernel (var_nopartfan.so), same call:
# parity of the timed kernels first, then the synchronous frame by the number of split blocks.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
L=$REPO/yetanotherconsolegameengine_amd/lib
timeout 1500 python -m pytest tests/test_gpu_timed_variants.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r6_partfan_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r6_partfan_pytest.log
for cfg in 4 3; do
  echo "== config $cfg"
  YCGE_LIB=$L/var_nopartfan.so timeout 200 python profiles/sync_ms.py $cfg 200 "round5-kernel split_top=32" 2>&1 | tail -1
  for st in 32 64 128 256 512 1024; do YCGE_SPLIT_TOP=$st timeout 200 python profiles/sync_ms.py $cfg 200 "partfan split_top=$st" 2>&1 | tail -1; done
  YCGE_LIB=$L/var_nopartfan.so timeout 200 python profiles/sync_ms.py $cfg 200 "round5-kernel split_top=32 (again)" 2>&1 | tail -1
  YCGE_SPLIT_TOP=128 YCGE_SPLIT_TOP_LG=3 timeout 200 python profiles/sync_ms.py $cfg 200 "partfan split_top=128 in 8 parts" 2>&1 | tail -1
  YCGE_SPLIT_TOP=128 YCGE_SPLIT_TOP_LG=4 timeout 200 python profiles/sync_ms.py $cfg 200 "partfan split_top=128 in 16 parts" 2>&1 | tail -1
done
timeout 300 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_mega_prof_partfan.txt; grep -E "trace_ms|span|slot time|>= 256|blocks split" gpurun_out/r6_mega_prof_partfan.txt
YCGE_SPLIT_TOP=256 timeout 300 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_mega_prof_partfan256.txt; grep -E "trace_ms|span|slot time|>= 256|blocks split" gpurun_out/r6_mega_prof_partfan256.txt
