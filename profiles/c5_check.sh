#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 5', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame')"
bash profiles/c5_quick.sh | grep false
