#!/bin/bash
# the persistent in-place A-trous on a 3840x2160 trace grid (config 5's): per-band timeline, and the knobs that could matter there
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
echo "== config 5, probing build"; CFG=5 NB=540 YCGE_POST_PROBE_BAND=${1:-200} timeout 400 python profiles/post_bands.py 2>&1 | grep -v Warning | head -12
echo "== config 5, production build"; CFG=5 NB=540 timeout 300 python profiles/post_bands.py 2>&1 | grep "^frame"
