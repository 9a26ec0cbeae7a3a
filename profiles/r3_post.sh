#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
python -m pytest tests -x -q -m gpu -k "post or atrous or denoise or exposure or tonemap or sdr" 2>&1 | tail -3
run() { echo "== config $1 ss $2 $4"; CFG=$1 SS=$2 NB=$3 timeout 300 python profiles/post_bands.py 2>&1 | grep -E "^post|launch span|chain:|clocks|^   "; }
run 5 1 270
run 4 1 270
run 3 1 180
run 5 2 540
export YCGE_POST_PROBE_BAND=100
run 5 1 270 prof
