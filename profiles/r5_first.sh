#!/bin/bash
# round 5, first call: the new tests (in-flight live textures, host buffers), the self-launch forms at world 1, the headline
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "live_texture or page_locked or tile_resident or frames_in_flight_say" > gpurun_out/r5_first_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5_first_pytest.log
bash profiles/forms.sh 2>&1 | grep -v "^\[" | cut -c1-900
timeout 600 python bench.py --steps 100 --cpu-seconds 5 > gpurun_out/r5_first_bench.json 2> gpurun_out/r5_first_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r5_first_bench.json
