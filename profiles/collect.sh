#!/bin/bash
# copies a round_end.sh run's outputs from gpurun_out/ into profiles/<round>/ (tracked): collect.sh r04z z r04
TAG=$1; L=$2; R=${3:-r04}; cd "$(dirname "$0")/.."; D=profiles/$R; mkdir -p $D
for f in gpurun_out/bench_${TAG}*.json; do cp $f $D/; done
cp gpurun_out/prof_${TAG}_c4/summary.txt $D/${L}_config4_summary.txt
cp gpurun_out/prof_${TAG}_c5/summary.txt $D/${L}_config5_summary.txt
cp gpurun_out/prof_${TAG}_c5lit/summary.txt $D/${L}_config5_t050_summary.txt
cp gpurun_out/prof_${TAG}_c4/pmc_config4.json $D/pmc_config4.json
cp gpurun_out/prof_${TAG}_c5/pmc_config5.json $D/pmc_config5.json
cp gpurun_out/prof_${TAG}_c5lit/pmc_config5_t050.json $D/pmc_config5_t050.json
cp gpurun_out/pytest_gpu_${TAG}.log $D/${L}_pytest_gpu.log
cp gpurun_out/mega_prof_${TAG}.txt $D/${L}_mega_prof_config4.txt
grep -E "source_hash" $D/pmc_config4.json | head -2
