#!/bin/bash
# Round 6's one-call experiments as modes of ONE script (no more r<N>_*.sh files):   bash profiles/experiments.sh <mode>
# Every mode is what ran in one gpurun call; results and readings are under profiles/r06/ (a_ ... f_).  Some modes load variant libraries that are
# not part of the product build (python profiles/build_variant.py <name> <-D flags>: nopartfan -DYCGE_PARTFAN=0 ... see the mode).
# Modes: partfan, partfan2, dbg, taafuse, taafuse2, taafuse3, suite, resolve, split2, costframes
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
L=$REPO/yetanotherconsolegameengine_amd/lib

# round 6: query fan-out INSIDE the wavefront for the parts of split blocks (trace_block MODE 3) against round 5's kernel (var_nopartfan.so), same call:
# parity of the timed kernels first, then the synchronous frame by the number of split blocks.
mode_partfan() {
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
}

# round 6: (1) the fault of the first partfan run, alone, under each library; (2) the exception barrier under allocation failure; (3) MODE 3 for parts only
# (var_partfan2.so: whole blocks through round 5's loop) against MODE 3 for everything (product) against round 5 (var_nopartfan.so); rank emulation of each.
mode_partfan2() {
  for lib in var_nopartfan.so libycge_hip.so var_partfan2.so; do
    echo "== $lib: test_one_call_drives_several_devices"
    YCGE_LIB=$L/$lib timeout 300 python -m pytest tests/test_gpu_timed_variants.py -m gpu -x -q -k "one_call_drives_several_devices" 2>&1 | grep -E "passed|failed|fault|Error" | head -5
  done
  echo "== abi barrier"; timeout 900 python -m pytest tests/test_gpu_abi_barrier.py -m gpu -x -q 2>&1 | tail -8
  for cfg in 4 3; do
    echo "== config $cfg"
    for spec in "var_nopartfan.so 32" "libycge_hip.so 32" "var_partfan2.so 32" "var_partfan2.so 64" "var_partfan2.so 128" "var_partfan2.so 256" "var_nopartfan.so 32"; do set -- $spec
      YCGE_LIB=$L/$1 YCGE_SPLIT_TOP=$2 timeout 200 python profiles/sync_ms.py $cfg 200 "$1 split_top=$2" 2>&1 | tail -1; done
  done
  echo "== a rank of 8 / 4, tile-resident ring of 4 (frame by frame) and batches of 4"
  for lib in var_nopartfan.so libycge_hip.so var_partfan2.so; do
    for w in 8 4; do echo -n "$lib world $w K=4: "; YCGE_LIB=$L/$lib timeout 300 python profiles/rank_flight.py 4 $w residentc 4 2>&1 | tail -1 | cut -c1-200; done
    echo -n "$lib world 8 slab two: "; YCGE_LIB=$L/$lib timeout 300 python profiles/rank_flight.py 4 8 two 2>&1 | tail -1 | cut -c1-200
  done
}

# round 6: the memory fault of the MODE 3 kernel in test_one_call_drives_several_devices, by variant and knob
mode_dbg() {
  T="tests/test_gpu_timed_variants.py -m gpu -x -q -k one_call_drives_several_devices"
  run() { echo "== $*"; ( for kv in "$@"; do export "$kv"; done; timeout 300 python -m pytest $T 2>&1 | grep -E "passed|failed|fault|Error|rror:" | head -4 ); }
  run A=product
  run YCGE_LIB=$L/var_pf_nofan.so
  run YCGE_LIB=$L/var_pf_zero.so
  run YCGE_NO_COOP=1
  run YCGE_SPLIT=0
  run YCGE_SPLIT=022220000
  run YCGE_SPLIT=033330000
  run AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3
  echo "== last launches before the fault (AMD_LOG_LEVEL=3)"
  AMD_LOG_LEVEL=3 timeout 300 python -m pytest $T > gpurun_out/r6_dbg_log.txt 2>&1; grep -E "ShaderName|fault" gpurun_out/r6_dbg_log.txt | tail -12 | cut -c1-260
  echo "== abi barrier + page-locked"; timeout 900 python -m pytest tests/test_gpu_abi_barrier.py -m gpu -x -q 2>&1 | tail -5
}

# round 6: TemporalBlendWithClamp inside the trace launch (ycge::TaaFuse) and the schedule built a frame ahead (Knobs::sync_defer), synchronous frames, same call
mode_taafuse() {
  timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_taafuse_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6_taafuse_pytest.log
  for cfg in 4 3 2 1; do
    echo "== config $cfg"
    for spec in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do set -- $spec
      YCGE_TAA_FUSE=$1 YCGE_SYNC_DEFER=$2 timeout 200 python profiles/sync_ms.py $cfg 300 "fuse=$1 defer=$2" 2>&1 | tail -1; done
  done
}

# round 6: where the time of the fused TAA goes (timing-only variants: written-through stores alone; + counters; everything); the pinning reproducer
mode_taafuse2() {
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
}

# round 6: fused TAA, window staged in LDS: parity of the TAA tests, then synchronous frames fused / separate x schedule deferred / not
mode_taafuse3() {
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r6_taafuse3_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6_taafuse3_pytest.log
  for cfg in 4 3 2 1; do
    echo "== config $cfg"
    for spec in "0 0" "1 0" "1 1" "0 1" "0 0" "1 1"; do set -- $spec
      YCGE_TAA_FUSE=$1 YCGE_SYNC_DEFER=$2 timeout 200 python profiles/sync_ms.py $cfg 300 "fuse=$1 defer=$2" 2>&1 | tail -1; done
  done
}

# round 6: the GPU suite on the build of the hour + the heaviest blocks in TWO parts (YCGE_SPLIT_TOP_LG=1), same call
mode_suite() {
  timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r6_suite_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r6_suite_pytest.log
  for cfg in 4 3; do
    echo "== config $cfg"
    for spec in "32 2" "64 1" "128 1" "256 1" "512 1" "32 2"; do set -- $spec
      YCGE_SPLIT_TOP=$1 YCGE_SPLIT_TOP_LG=$2 timeout 200 python profiles/sync_ms.py $cfg 300 "split_top=$1 parts=2^$2" 2>&1 | tail -1; done
  done
}

# round 6: the tile-resident resolve as ONE launch (k_resolve_tiles) against round 5's two; hardware queues per stream; a rank's period frame by frame
mode_resolve() {
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_resident" 2>&1 | tail -3
  run() { echo -n "$1 | world $2 K=$3: "; ( for kv in $1; do export "$kv"; done; timeout 300 python profiles/rank_flight.py 4 $2 residentc $3 2>&1 | tail -1 | cut -c75-210 ); }
  for w in 8 4; do
    run "A=one_launch" $w 4
    run "YCGE_RES_SPLIT_RESOLVE=1" $w 4
    run "GPU_MAX_HW_QUEUES=8" $w 4
    run "GPU_MAX_HW_QUEUES=8 YCGE_RES_SPLIT_RESOLVE=1" $w 4
    run "GPU_MAX_HW_QUEUES=8" $w 6
    run "GPU_MAX_HW_QUEUES=8" $w 8
    run "GPU_MAX_HW_QUEUES=16" $w 8
    run "A=one_launch" $w 6
    run "YCGE_RES_LOOP_COMM=slot" $w 4
    run "GPU_MAX_HW_QUEUES=8 YCGE_RES_LOOP_COMM=slot" $w 4
    run "GPU_MAX_HW_QUEUES=8 YCGE_RES_LOOP_COMM=slot" $w 8
  done
  echo "== ring 1 (latency 1: the synchronous form of a rank)"; run "A=one_launch" 8 2
}

# round 6: the new split default (two parts, count by block count) against round 5's (32 blocks in four), the spread of frame times with the SAME frame
# number every frame (same rays) against consecutive frames, the GPU suite on this build
mode_split2() {
  timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r6_split2_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6_split2_pytest.log
  for cfg in 4 3; do
    echo "== config $cfg"
    YCGE_SPLIT_TOP=32 YCGE_SPLIT_TOP_LG=2 timeout 200 python profiles/sync_ms.py $cfg 300 "round 5: 32 x 4 parts" 2>&1 | tail -1
    timeout 200 python profiles/sync_ms.py $cfg 300 "default" 2>&1 | tail -1
    YCGE_SPLIT_TOP=32 YCGE_SPLIT_TOP_LG=2 timeout 200 python profiles/sync_ms.py $cfg 300 "round 5: 32 x 4 parts" 2>&1 | tail -1
    timeout 200 python profiles/sync_ms.py $cfg 300 "default" 2>&1 | tail -1
    SAME_FRAME=40 timeout 200 python profiles/sync_ms.py $cfg 300 "default, frame 40 every time" 2>&1 | tail -1
    SAME_FRAME=41 timeout 200 python profiles/sync_ms.py $cfg 300 "default, frame 41 every time" 2>&1 | tail -1
    SAME_FRAME=42 timeout 200 python profiles/sync_ms.py $cfg 300 "default, frame 42 every time" 2>&1 | tail -1
  done
}

# round 6: frames of history behind a block's schedule cost (maximum over the last n frames; 4 in the product), synchronous frames, same call
mode_costframes() {
  for cfg in 4 3; do
    echo "== config $cfg"
    timeout 200 python profiles/sync_ms.py $cfg 300 "history 4 (product)" 2>&1 | tail -1
    for n in 2 3 6 8; do YCGE_LIB=$L/var_cost$n.so timeout 200 python profiles/sync_ms.py $cfg 300 "history $n" 2>&1 | tail -1; done
    timeout 200 python profiles/sync_ms.py $cfg 300 "history 4 (product)" 2>&1 | tail -1
  done
}

# round 6: the schedule's last class is cost ZERO alone (sky) - against round 5's classes (var_order5.so: -DYCGE_ORDER_ZERO_LAST=0); timing events only with statistics
mode_zerolast() {
  for cfg in 4 3; do
    echo "== config $cfg"
    for rep in 1 2; do
      YCGE_LIB=$L/var_order5.so timeout 200 python profiles/sync_ms.py $cfg 300 "round 5's order classes" 2>&1 | tail -1
      timeout 200 python profiles/sync_ms.py $cfg 300 "zero-cost blocks last" 2>&1 | tail -1
    done
    NO_STATS=1 timeout 200 python profiles/sync_ms.py $cfg 300 "zero-cost blocks last, no statistics asked for (no timing events)" 2>&1 | tail -1
    timeout 200 python profiles/sync_ms.py $cfg 300 "zero-cost blocks last" 2>&1 | tail -1
  done
  timeout 300 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_mega_prof_zerolast.txt; grep -E "trace_ms|span|slot time|>= 256|blocks split" gpurun_out/r6_mega_prof_zerolast.txt; sed -n 7,16p gpurun_out/r6_mega_prof_zerolast.txt
  timeout 600 python -m pytest tests/test_gpu_timed_variants.py -m gpu -x -q -k "steady or mesh or frames_in_flight" 2>&1 | tail -2
}

case "$1" in
  partfan|partfan2|dbg|taafuse|taafuse2|taafuse3|suite|resolve|split2|costframes|zerolast) mode_$1 ;;
  *) echo "usage: bash profiles/experiments.sh <partfan|partfan2|dbg|taafuse|taafuse2|taafuse3|suite|resolve|split2|costframes>"; exit 2 ;;
esac
