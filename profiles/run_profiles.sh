#!/bin/bash
# Profiles bench.py on the GPU box: one --kernel-trace --stats run, then separate --pmc passes
# (counters are never combined with tracing domains other than kernel-trace; see task notes).
# usage: profiles/run_profiles.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-post --no-flight --no-moving $*"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_BRANCH SQ_IFETCH" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -f csv -d $OUT/pmc$i -o pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i rc=$? ($set)"
done
# calibration of FETCH_SIZE / WRITE_SIZE on a copy of known size (same two passes)
if [ -x $REPO/profiles/micro/copycal ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_calf -o calf -- $REPO/profiles/micro/copycal > $OUT/copycal_f.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_calw -o calw -- $REPO/profiles/micro/copycal > $OUT/copycal_w.log 2>&1
fi
CFG=4; T01=""; prev=""; for a in "$@"; do if [ "$prev" = "--config" ]; then CFG=$a; fi; if [ "$prev" = "--t01" ]; then T01=$a; fi; prev=$a; done
# (a lit config-5 run - bench.py --t01 other than the survey's 0.25 - gets its own summary: pmc_config5_t050.json, what bench.py looks for)
TAG01=""; if [ -n "$T01" ] && [ "$T01" != "0.25" ]; then TAG01=$(python3 -c "print('_t%03d' % round(float('$T01') * 100))"); fi
python3 $REPO/profiles/summarize.py $OUT --json $OUT/pmc_config$CFG$TAG01.json --config $CFG --source "timeout 600 rocprofv3 --kernel-trace --stats + separate --pmc passes of: bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-post --no-flight --no-moving $* (tag $TAG)" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
