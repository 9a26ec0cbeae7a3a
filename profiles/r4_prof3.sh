#!/bin/bash
# rocprofv3 kernel trace + PMC passes of configs 3 and 2 (the bench lines of those configs then carry counters too), and their bench lines again
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out profiles/r04
for c in 3 2; do
  bash profiles/run_profiles.sh r04z_c$c --config $c > gpurun_out/prof_r04z_c$c.log 2>&1; echo "profiles c$c rc=$?"; head -8 gpurun_out/prof_r04z_c$c/summary.txt
  [ -f gpurun_out/prof_r04z_c$c/pmc_config$c.json ] && cp gpurun_out/prof_r04z_c$c/pmc_config$c.json profiles/r04/
  timeout 400 python bench.py --config $c --cpu-seconds 10 2> gpurun_out/bench_r04z_cfg$c.err > gpurun_out/bench_r04z_cfg$c.json
  python -c "import json; d=json.load(open('gpurun_out/bench_r04z_cfg$c.json')); r=d['roofline']; print($c, d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), r.get('lanes_active'), r.get('valu_busy'))"
done
