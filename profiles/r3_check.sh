#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_check.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_check.log
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'), d['roofline']['kernel'])"; }
timeout 200 python bench.py --config 2 --steps 300 --no-cpu-baseline --no-post --no-moving 2>/dev/null | one cfg2
YCGE_NO_COOP=1 timeout 200 python bench.py --steps 200 --no-cpu-baseline --no-post --no-moving 2>/dev/null | one "cfg4 no-coop (4-wave nomesh instance)"
timeout 200 python bench.py --steps 200 --no-cpu-baseline --no-post --no-moving 2>/dev/null | one "cfg4 default"
timeout 200 python bench.py --config 5 --steps 50 --no-cpu-baseline --no-post --no-moving 2>/dev/null | one "cfg5"
