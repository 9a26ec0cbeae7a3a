#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
run() { echo "== config $1 ss $2 $4"; CFG=$1 SS=$2 NB=$3 timeout 300 python profiles/post_bands.py 2>&1 | grep -E "^post|launch span|chain:|clocks|^   "; }
for b in 0 1 2 4; do
export YCGE_POST_BACKOFF=$b
run 5 1 270 backoff$b
run 5 2 540 backoff$b
done
