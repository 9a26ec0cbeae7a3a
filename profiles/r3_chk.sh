#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
timeout 900 python -m pytest tests/test_gpu_timed_variants.py -x -q -m gpu 2>&1 | tail -2
for c in 4 3 2 5; do timeout 300 python profiles/flight_ab.py $c 300 2>&1 | tail -1; done
