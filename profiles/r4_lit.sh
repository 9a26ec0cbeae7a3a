#!/bin/bash
# round 4, first call: the timed voxel kernels against the oracle WITH a light on (noon / night / low sun), then config 5 benched at noon beside the survey's dark phase
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_timed_variants.py -m gpu -q -x -k "light_on or graze_culled or all_air" --durations=6 > gpurun_out/r4_lit_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r4_lit_pytest.log
for t in 0.5 0.25; do timeout 500 python bench.py --config 5 --t01 $t --cpu-seconds 8 --steps 100 > gpurun_out/r4_bench_cfg5_t$t.json 2> gpurun_out/r4_bench_cfg5_t$t.err; echo "bench t01=$t rc=$?"; python - <<PY
import json
d = json.load(open("gpurun_out/r4_bench_cfg5_t$t.json"))
print("t01=$t", d["value"], "Mrays/s traced;", d["value_reference_ray_count"], "by the reference's count;", d["ms_per_step"], "ms/frame; trace", d.get("trace_ms"), "rays", d["rays_per_frame"], "dark", d["rays_to_dark_lights_per_frame"], "cpu", (d.get("cpu_baseline") or {}).get("value"), "x", d.get("gpu_over_cpu"), "post", (d.get("post_stage") or {}).get("post_ms"))
PY
done
