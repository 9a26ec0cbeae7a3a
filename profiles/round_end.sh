#!/bin/bash
# round-end measurement on the GPU box: full GPU test suite, bench of every GPU config WITH the CPU baseline beside it (config 5 at the survey's
# dark day phase AND at noon), the moving-camera headline, rocprofv3 kernel trace + PMC passes of config 4, config 5 dark and config 5 lit
# (every profiler run under its own timeout), rank emulations (slab form, tile-resident ring), in-flight A/B, post stage exact / waived,
# per-wavefront profile, cooperative-walk clocks.  Everything lands in gpurun_out/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-r06}; RND=${2:-r06}; PART=${3:-ab}          # part a: suite, profiles, benches; part b: emulations, A/Bs, forms (gpurun caps a call at one hour)
if [[ $PART == *a* ]]; then
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
# the profiles FIRST: bench.py prints the counter-derived roofline fields only from a summary of the running build (profiles/$RND/pmc_config*.json, source_hash)
bash profiles/run_profiles.sh ${TAG}_c4 > gpurun_out/prof_${TAG}_c4.log 2>&1; echo "profiles c4 rc=$?"; head -12 gpurun_out/prof_${TAG}_c4/summary.txt
bash profiles/run_profiles.sh ${TAG}_c5 --config 5 > gpurun_out/prof_${TAG}_c5.log 2>&1; echo "profiles c5 rc=$?"; head -12 gpurun_out/prof_${TAG}_c5/summary.txt
bash profiles/run_profiles.sh ${TAG}_c5lit --config 5 --t01 0.5 > gpurun_out/prof_${TAG}_c5lit.log 2>&1; echo "profiles c5 lit rc=$?"; head -12 gpurun_out/prof_${TAG}_c5lit/summary.txt
mkdir -p profiles/$RND; for f in gpurun_out/prof_${TAG}_c4/pmc_config4.json gpurun_out/prof_${TAG}_c5/pmc_config5.json gpurun_out/prof_${TAG}_c5lit/pmc_config5_t050.json; do [ -f $f ] && cp $f profiles/$RND/; done
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
for c in 1 2 3 5; do timeout 400 python bench.py --config $c --cpu-seconds 10 2>> gpurun_out/bench_$TAG.err > gpurun_out/bench_${TAG}_cfg$c.json; done
timeout 500 python bench.py --config 5 --t01 0.5 --cpu-seconds 10 2>> gpurun_out/bench_$TAG.err > gpurun_out/bench_${TAG}_cfg5_t050.json
timeout 300 python bench.py --camera orbit --no-cpu-baseline > gpurun_out/bench_${TAG}_orbit.json 2>> gpurun_out/bench_$TAG.err
for f in bench_$TAG bench_${TAG}_cfg1 bench_${TAG}_cfg2 bench_${TAG}_cfg3 bench_${TAG}_cfg5 bench_${TAG}_cfg5_t050 bench_${TAG}_orbit; do python - <<PY
import json
try:
    d = json.load(open("gpurun_out/$f.json"))
    fl = d.get("frames_in_flight") or {}
    print("$f", d["value"], "Mrays/s (by the reference's count", d.get("value_reference_ray_count"), ")", d["ms_per_step"], "ms/frame; trace", d.get("trace_ms"), "in flight", fl.get("ms_per_step"), fl.get("value"), "gate", fl.get("gate"), "with sdr", (fl.get("with_sdr") or {}).get("ms_per_step"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "x", d.get("gpu_over_cpu"), "moving", (d.get("moving_camera") or {}).get("trace_ms"), "post", (d.get("post_stage") or {}).get("post_ms"), (d.get("post_stage") or {}).get("frame_ms_with_sdr_readback"), "roofline frac", (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("$f failed", e)
PY
done
fi
if [[ $PART == *b* ]]; then
for cfg in 4 3; do echo "== rank emulation config $cfg"; timeout 600 python profiles/rank_times.py $cfg 2>&1 | grep -E "world"; done
echo "== a rank's period: slab form (two trace streams), tile-resident ring (loop driven from C)"
for w in 8 4 2; do timeout 300 python profiles/rank_flight.py 4 $w two 2>&1 | tail -1; done
for w in 8 4 2; do for k in 2 4; do timeout 300 python profiles/rank_flight.py 4 $w residentc $k 2>&1 | tail -1; done; done
timeout 300 python profiles/rank_flight.py 3 8 residentc 4 2>&1 | tail -1; timeout 300 python profiles/rank_flight.py 5 8 residentc 2 2>&1 | tail -1
echo "== ... n frames of a rank's tiles in ONE launch (ycge_trace_tiles_resident_batch; ring = three batches' sets)"
for spec in "8 6 2" "8 12 4" "8 15 5" "4 6 2" "4 12 4" "2 6 2" "2 12 4"; do set -- $spec; echo -n "batch $3: "; YCGE_RES_LOOP_BATCH=$3 timeout 300 python profiles/rank_flight.py 4 $1 residentc $2 2>&1 | tail -1; done
echo -n "batch 4: "; YCGE_RES_LOOP_BATCH=4 timeout 300 python profiles/rank_flight.py 3 8 residentc 12 2>&1 | tail -1
echo "== frames in flight against the synchronous call"; for cfg in 4 3 2 5; do timeout 300 python profiles/flight_ab.py $cfg 300 2>&1 | tail -2; done
echo "== post stage: exact and waived (config.atrous_inplace_exact)"; timeout 600 python profiles/post_waiver.py 2>&1 | grep -v amdgpu.ids
echo "== post stage bands"; for a in "4 1 270" "5 2 540"; do set -- $a; CFG=$1 SS=$2 NB=$3 timeout 300 python profiles/post_bands.py 2>&1 | grep -E "^post|launch span|chain:"; done
timeout 300 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/mega_prof_$TAG.txt; grep -E "trace_ms|span|slot time|>= 256" gpurun_out/mega_prof_$TAG.txt
echo "== voxel worlds: the walk tree against the scene tree, the light loop beside the trace, walk_phase (lit config 5, same call)"
for v in "-" "YCGE_NO_WALK_TREE=1" "YCGE_NO_LIGHTS_BESIDE=1" "-"; do
  ( if [ "$v" != "-" ]; then export "$v"; fi; echo -n "$v: "
    timeout 300 python bench.py --config 5 --t01 0.5 --steps 20 --warmup 3 --no-cpu-baseline --no-post 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms/frame trace', d['roofline']['mean_launch_ms'], 'moving', d['moving_camera']['trace_ms']['median'], 'in flight', d['frames_in_flight']['ms_per_step'])" )
done
if [ -f yetanotherconsolegameengine_amd/lib/var_voxstat.so ]; then YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_voxstat.so timeout 300 python profiles/vox_stats.py 0.5 2>&1 | grep -v amdgpu.ids; fi
if [ -f yetanotherconsolegameengine_amd/lib/var_coopstat.so ]; then YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_coopstat.so timeout 200 python profiles/coop_stats.py 4 2>&1 | grep -v amdgpu.ids; fi
if [ -f yetanotherconsolegameengine_amd/lib/var_batchstat.so ]; then YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_batchstat.so timeout 300 python profiles/batch_stats.py 4 2>&1 | grep -v amdgpu.ids; fi
echo "== round 6: the synchronous frame, p95 / p99 / max (config 4, 3); the all-gather behind the one call as a world of one against the plain frame"
for cfg in 4 3 2 1; do timeout 200 python profiles/sync_ms.py $cfg 300 "final build" 2>&1 | tail -1; NO_STATS=1 timeout 200 python profiles/sync_ms.py $cfg 300 "final build, no statistics asked for" 2>&1 | tail -1; done
timeout 300 python profiles/exchange_ms.py 4 2>&1 | grep "^config"
echo "== bench.py's one-process-per-GPU forms on one rank (torchrun + RCCL)"; bash profiles/forms.sh 2>&1 | grep -v "^\[" | cut -c1-330
fi
